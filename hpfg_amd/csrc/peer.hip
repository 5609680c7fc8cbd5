// Host side of the peer mailbox exchange (peer.h): allocation of the fine-grained mailbox, IPC handles, and the epoch-bump launch.
#include <string.h>
#include "common.h"

namespace {
__global__ void word_add_kernel(int32_t* w, int v) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *w = (int32_t)(((uint32_t)*w + (uint32_t)v) & 0x7fffffffu);
}

// ---- gradient all-reduce over the peer windows (include/hpfg_hip.h: HpfgPeerBuf) ------------------------------------------------------
constexpr int PB_FLAG_BYTES = 256;          // in_flag[r] at int 0..7, out_flag[r] at int 16..23
constexpr int PB_WG = 96;                   // workgroups of the polling kernels: few enough to share one GPU with another rank's kernels (tests)

__device__ __forceinline__ float* pb_inbox(const HpfgPeerBuf& pb, int owner, int src) {
  return reinterpret_cast<float*>(reinterpret_cast<char*>(pb.win[owner]) + PB_FLAG_BYTES) + (size_t)src * pb.stride;
}
__device__ __forceinline__ float* pb_result(const HpfgPeerBuf& pb, int owner) {
  return reinterpret_cast<float*>(reinterpret_cast<char*>(pb.win[owner]) + PB_FLAG_BYTES) + (size_t)pb.world * pb.stride;
}
__device__ __forceinline__ long pb_count(const HpfgPeerBuf& pb, int s) {          // valid floats of slice s
  const long c = pb.n - (long)s * pb.slice;
  return c < 0 ? 0 : (c > pb.slice ? pb.slice : c);
}

// set flag `which` (0: in, 16: out) of every peer to the epoch, then wait until every peer has set mine (bounded).  Called by all threads
// of all workgroups; on return the peers' stores that preceded their flag stores are visible to every thread of the workgroup.
__device__ inline void pb_handshake(const HpfgPeerBuf& pb, int which, int32_t ep) {
  const int tid = threadIdx.x;
  if (blockIdx.x == 0 && tid < pb.world && tid != pb.rank) {
    __atomic_thread_fence(__ATOMIC_RELEASE);          // system scope: everything this rank stored before this launch is performed first
    int32_t* f = reinterpret_cast<int32_t*>(pb.win[tid]) + which + pb.rank;
    __hip_atomic_store(f, ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (tid < pb.world && tid != pb.rank) {
    const int32_t* f = reinterpret_cast<const int32_t*>(pb.win[pb.rank]) + which + tid;
    long spins = 0;
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != ep) {
      if (spins < 64) __builtin_amdgcn_s_sleep(1);
      else __builtin_amdgcn_s_sleep(32);
      if (++spins > 8 * HPFG_PEER_MAX_SPINS) {      // (more patient than the mailbox polls: a rank may still be capturing its graph) every wave reaches an exit: report and go on
        if (pb.err) __hip_atomic_store(pb.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
  __syncthreads();
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
}

// (1) my copy of slice p -> rank p's inbox[my rank]; grid = (chunks, world).  One thread also counts the call (the epoch the two
// handshakes of this call carry; nothing in this launch reads it).
__global__ __launch_bounds__(256) void pb_push_kernel(HpfgPeerBuf pb, const float* __restrict__ buf) {
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *pb.epoch = (int32_t)(((uint32_t)*pb.epoch + 1u) & 0x7fffffffu);
  const int p = blockIdx.y;
  if (p == pb.rank) return;
  const long cnt = pb_count(pb, p);
  const float* src = buf + (size_t)p * pb.slice;
  float* dst = pb_inbox(pb, p, pb.rank);
  const long c4 = cnt >> 2;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < c4; i += (long)gridDim.x * 256)
    reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(src)[i];
  if (blockIdx.x == 0 && threadIdx.x < (cnt & 3)) dst[(c4 << 2) + threadIdx.x] = src[(c4 << 2) + threadIdx.x];
  __atomic_thread_fence(__ATOMIC_RELEASE);            // the remote stores are performed before this launch ends
}

// (2) every contribution to my slice has arrived: add them in rank order, store the sum into every rank's result
__global__ __launch_bounds__(256) void pb_reduce_kernel(HpfgPeerBuf pb, const float* __restrict__ buf) {
  const int32_t ep = *pb.epoch;
  pb_handshake(pb, 0, ep);
  const long cnt = pb_count(pb, pb.rank), c4 = cnt >> 2;
  const long base = (long)pb.rank * pb.slice;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < c4; i += (long)gridDim.x * 256) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < pb.world; ++r)
      v += r == pb.rank ? reinterpret_cast<const f32x4*>(buf + base)[i] : reinterpret_cast<const f32x4*>(pb_inbox(pb, pb.rank, r))[i];
    for (int p = 0; p < pb.world; ++p) reinterpret_cast<f32x4*>(pb_result(pb, p) + base)[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (cnt & 3)) {
    const long i = (c4 << 2) + threadIdx.x;
    float v = 0.f;
    for (int r = 0; r < pb.world; ++r) v += r == pb.rank ? buf[base + i] : pb_inbox(pb, pb.rank, r)[i];
    for (int p = 0; p < pb.world; ++p) pb_result(pb, p)[base + i] = v;
  }
  __atomic_thread_fence(__ATOMIC_RELEASE);
}

// (3) every reduced slice has arrived in my window: copy the buffer back
__global__ __launch_bounds__(256) void pb_gather_kernel(HpfgPeerBuf pb, float* __restrict__ buf) {
  const int32_t ep = *pb.epoch;
  pb_handshake(pb, 16, ep);
  const float* res = pb_result(pb, pb.rank);
  const long n4 = pb.n >> 2;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) reinterpret_cast<f32x4*>(buf)[i] = reinterpret_cast<const f32x4*>(res)[i];
  if (blockIdx.x == 0 && threadIdx.x < (pb.n & 3)) buf[(n4 << 2) + threadIdx.x] = res[(n4 << 2) + threadIdx.x];
}
}  // namespace

extern "C" long hpfg_peer_buf_slice(int world, long n) {
  if (world < 1 || n < 0) return -1;
  const long s = (n + world - 1) / world;
  return (s + 3) / 4 * 4;
}

extern "C" long hpfg_peer_buf_bytes(int world, long n) {
  const long s = hpfg_peer_buf_slice(world, n);
  return s < 0 ? -1 : PB_FLAG_BYTES + 2L * world * s * (long)sizeof(float);
}

extern "C" int hpfg_peer_allreduce_f32(const HpfgPeerBuf* pb, float* buf, void* stream) {
  HPFG_ARG_CHECK(pb && buf, "peer_allreduce: null pointer");
  HPFG_ARG_CHECK(pb->world >= 1 && pb->world <= HPFG_PEER_MAX_RANKS && pb->rank >= 0 && pb->rank < pb->world, "peer_allreduce: bad world / rank %d / %d",
                 pb->world, pb->rank);
  if (pb->world == 1 || pb->n == 0) return 0;
  HPFG_ARG_CHECK(pb->epoch && pb->n > 0 && pb->slice == hpfg_peer_buf_slice(pb->world, pb->n), "peer_allreduce: bad epoch / n / slice");
  // the window layout comes from the capacity, never from this call's n (consecutive calls of different sizes share the window without a handshake between them)
  HPFG_ARG_CHECK(pb->stride >= pb->slice && pb->stride % 4 == 0 && (long)pb->world * pb->slice <= (long)pb->world * pb->stride,
                 "peer_allreduce: stride %ld must be the capacity slice (a multiple of 4, >= this call's slice %ld)", (long)pb->stride, (long)pb->slice);
  HPFG_ARG_CHECK(((uintptr_t)buf & 15) == 0, "peer_allreduce: the buffer must be 16-byte aligned");
  for (int r = 0; r < pb->world; ++r) HPFG_ARG_CHECK(pb->win[r], "peer_allreduce: window of rank %d not mapped", r);
  hipStream_t st = (hipStream_t)stream;
  const long c4 = pb->slice / 4;
  const int chunks = (int)(c4 / 256 < 1 ? 1 : (c4 / 256 > 64 ? 64 : c4 / 256));
  hipLaunchKernelGGL(pb_push_kernel, dim3(chunks, pb->world), dim3(256), 0, st, *pb, (const float*)buf);
  hipLaunchKernelGGL(pb_reduce_kernel, dim3(PB_WG), dim3(256), 0, st, *pb, (const float*)buf);
  hipLaunchKernelGGL(pb_gather_kernel, dim3(PB_WG), dim3(256), 0, st, *pb, buf);
  return hpfg_launch_status("peer_allreduce_f32");
}

extern "C" int hpfg_peer_alloc(size_t bytes, void** ptr) {
  HPFG_ARG_CHECK(ptr && bytes > 0, "peer_alloc: bad args");
  // uncached (fine-grained) device memory: a peer's stores over xGMI and this GPU's polling loads meet in memory, not in a stale L2 line
  hipError_t e = hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    hpfg_set_error("peer_alloc: hipExtMallocWithFlags(%zu): %s", bytes, hipGetErrorString(e));
    return (int)e;
  }
  e = hipMemset(*ptr, 0, bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    hpfg_set_error("peer_alloc: memset: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

extern "C" int hpfg_peer_free(void* ptr) {
  hipError_t e = hipFree(ptr);
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int hpfg_peer_handle(void* ptr, unsigned char* handle64) {
  HPFG_ARG_CHECK(ptr && handle64, "peer_handle: bad args");
  static_assert(sizeof(hipIpcMemHandle_t) == HPFG_PEER_HANDLE_BYTES, "IPC handle size");
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, ptr);
  if (e != hipSuccess) {
    hpfg_set_error("peer_handle: hipIpcGetMemHandle: %s", hipGetErrorString(e));
    return (int)e;
  }
  memcpy(handle64, &h, sizeof(h));
  return 0;
}

extern "C" int hpfg_peer_open(const unsigned char* handle64, void** ptr) {
  HPFG_ARG_CHECK(ptr && handle64, "peer_open: bad args");
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof(h));
  hipError_t e = hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) {
    hpfg_set_error("peer_open: hipIpcOpenMemHandle: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

extern "C" int hpfg_peer_close(void* ptr) {
  hipError_t e = hipIpcCloseMemHandle(ptr);
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int hpfg_word_add(int32_t* word, int v, void* stream) {
  HPFG_ARG_CHECK(word, "word_add: null pointer");
  hipLaunchKernelGGL(word_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, word, v);
  return hpfg_launch_status("word_add_kernel");
}

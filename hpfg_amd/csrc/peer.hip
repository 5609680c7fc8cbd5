// Host side of the peer mailbox exchange (peer.h): allocation of the fine-grained mailbox, IPC handles, and the epoch-bump launch.
#include <string.h>
#include "common.h"

namespace {
__global__ void word_add_kernel(int32_t* w, int v) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *w = (int32_t)(((uint32_t)*w + (uint32_t)v) & 0x7fffffffu);
}
}  // namespace

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

// Peer mailbox exchange: the small all-reduces of the data-parallel global-batch mode (BatchNorm sums, loss sums) done by the kernels
// that produce the sums, over IPC-mapped peer memory (xGMI between the GPUs of a node) -- no host involvement, no collective launch,
// capturable into a hipGraph.  The reference (main.py:44: single card) has no counterpart; semantics = SUM all-reduce in rank order.
//
// Every rank owns a mailbox (fine-grained device memory, hpfg_peer_alloc) that all peers map (hipIpc).  A slot holds, for each of two
// parities and each source rank, `cap` fp64 payload values and `cap` 32-bit flags.  A value v of index i is exchanged by the ONE thread
// that owns it:
//   publish: for every peer p: store v into p's mailbox [slot][parity][my rank][i] (system scope), release fence, store the epoch into
//            the flag of i there;
//   gather:  in rank order r = 0..world-1: own value, or poll MY mailbox's flag [slot][parity][r][i] until it equals the epoch (bounded),
//            acquire, load the payload.  Every rank adds the same numbers in the same order: bit-identical sums everywhere.
// The epoch is a device word that counts the uses of the slot (consecutive integers, the same sequence on every rank, bumped by a
// stream-ordered launch BEFORE the exchanging kernel); parity = epoch & 1.  Reuse is safe without an acknowledgement: a rank can only be
// in use k+2 of a slot (the next write to parity k&1) after it has gathered every peer's use-(k+1) value, which a peer publishes only
// after it finished its own use k, i.e. after it read the use-k values.
// A poll that does not complete within the bound sets *err and carries on with what it has (no hang; the host checks the word).
#pragma once
#include "common.h"

__device__ __forceinline__ size_t hpfg_px_payload_off(const HpfgPeerX& px, int parity, int src) {
  // slot = [2 parities][world][cap] doubles, then [2][world][cap] uint32 flags
  return (size_t)px.slot * px.slot_bytes + ((size_t)(parity * px.world + src) * px.cap) * sizeof(double);
}
__device__ __forceinline__ size_t hpfg_px_flag_off(const HpfgPeerX& px, int parity, int src) {
  return (size_t)px.slot * px.slot_bytes + (size_t)2 * px.world * px.cap * sizeof(double) + ((size_t)(parity * px.world + src) * px.cap) * sizeof(uint32_t);
}

// SUM over ranks of the K values v[0..K) that THIS thread owns, payload indices idx[0..K); call from exactly one thread per value set.
template <int K>
__device__ inline void hpfg_peer_allreduce(const HpfgPeerX& px, const int (&idx)[K], double (&v)[K]) {
  if (px.world <= 1) return;
  const uint32_t ep = (uint32_t)*px.epoch;
  const int par = (int)(ep & 1u);
  for (int p = 0; p < px.world; ++p) {
    if (p == px.rank) continue;
    double* pay = reinterpret_cast<double*>(reinterpret_cast<char*>(px.mbox[p]) + hpfg_px_payload_off(px, par, px.rank));
#pragma unroll
    for (int k = 0; k < K; ++k) __hip_atomic_store(pay + idx[k], v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __atomic_thread_fence(__ATOMIC_RELEASE);          // system scope: the payload stores are performed before any flag store
  for (int p = 0; p < px.world; ++p) {
    if (p == px.rank) continue;
    uint32_t* fl = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(px.mbox[p]) + hpfg_px_flag_off(px, par, px.rank));
#pragma unroll
    for (int k = 0; k < K; ++k) __hip_atomic_store(fl + idx[k], ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  double t[K];
#pragma unroll
  for (int k = 0; k < K; ++k) t[k] = 0.0;
  char* mine = reinterpret_cast<char*>(px.mbox[px.rank]);
  for (int r = 0; r < px.world; ++r) {
    if (r == px.rank) {
#pragma unroll
      for (int k = 0; k < K; ++k) t[k] += v[k];
      continue;
    }
    const uint32_t* fl = reinterpret_cast<const uint32_t*>(mine + hpfg_px_flag_off(px, par, r));
    const double* pay = reinterpret_cast<const double*>(mine + hpfg_px_payload_off(px, par, r));
#pragma unroll
    for (int k = 0; k < K; ++k) {
      long spins = 0;
      while (__hip_atomic_load(fl + idx[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != ep) {
        if (spins < 64) __builtin_amdgcn_s_sleep(1);          // a peer in step arrives within microseconds: poll tightly first,
        else __builtin_amdgcn_s_sleep(32);                    // then about once a microsecond (the bound below is a few seconds)
        if (++spins > HPFG_PEER_MAX_SPINS) {          // every wave reaches an exit: report and go on
          if (px.err) __hip_atomic_store(px.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
      }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
#pragma unroll
    for (int k = 0; k < K; ++k) t[k] += __hip_atomic_load(pay + idx[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = t[k];
}

// dmf_xgmi.h — device side of the one-shot gradient exchange over xGMI (include/dmf.h "dmf_xgmi_*").
//
// Memory layout (per rank; both buffers are allocated UNCACHED by their owner, who is the only one that READS them):
//   data  : [region 2][parity 2][src rank world][cap] 8-byte words   inbox; region 0 = training exchange, 1 = dmf_xgmi_allreduce
//   flags : one status word (+ spare); the name is round 2's, whose protocol kept one arrival flag per block here
// A word is {low: the fp32 value, high: the sequence number it belongs to}, written by ONE 8-byte store.  Protocol for the
// element `idx` at sequence number `seq` (monotonic, identical on all ranks, >= 1) — a PUSH exchange with tagged words:
//   1. store {g, seq} into data[r][region][seq & 1][rank][idx] of EVERY peer r      (system-scope stores through the IPC mapping)
//   2. poll the words [peer][idx] of the OWN inbox until each carries `seq`         (wall-clock bounded)
//   3. add the world values in rank order (the own one from the register)
// The value and its "arrived" mark are the same store, so a reader that sees the mark has the value: no acknowledgement of the
// data before a flag may be raised (round 2's form waited a store round trip over the link, then sent the flag on a second
// trip), no per-block flags, no workgroup barrier — one link crossing per exchange instead of three.  (The low-latency form of
// NCCL / RCCL's ring protocols, applied to a one-shot all-to-all: 8 K words of 8 bytes per peer.)
// Reads only ever go to memory the reader allocated itself as uncached, so they can never be served from a stale cache
// line.  An IPC import does not carry the exporter's uncached attribute: the first (pull) form of this exchange read the
// peers' buffers through the imported mapping, those loads were cached in the READER's XCD L2, and the third round — the
// first one to reuse a parity slot — summed the slot's round-1 content for every element (gpurun_out/dp3r.log of round 1:
// all 1000 elements off by the known-answer pattern 3 * (i % 97)).  A peer can run at most one sequence number ahead (it
// needs this rank's words of `seq` to finish `seq`, and this rank sends those of `seq + 1` only after it has read `seq`), which
// is what the two parities are for: a word of parity p is overwritten by `seq + 2` only after its reader is past `seq`.
// Lanes never wait for other lanes or blocks of their own grid.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dmf {

constexpr int XGMI_MAX = 16;

struct XgmiDev {
  int world, rank;
  int64_t cap;
  int64_t timeout_ticks;            // wall_clock64 ticks (100 MHz)
  unsigned long long* data[XGMI_MAX];
  int32_t* flags[XGMI_MAX];         // flags[rank][0]: status (0 ok, 1 a wait timed out: sticky, later exchanges do not wait)
};

__device__ __forceinline__ unsigned long long xgmi_word(float v, int seq) {
  return ((unsigned long long)(unsigned)seq << 32) | (unsigned long long)__builtin_bit_cast(unsigned, v);
}

// Returns the rank-ordered sum for lanes with valid == true (the others return 0).  No barriers inside: lanes of a block
// may call it under a block-uniform or a divergent condition alike.
__device__ __forceinline__ float xgmi_exchange(const XgmiDev& x, int region, int seq, int64_t idx, bool valid, float g) {
  float s = 0.f;
  if (valid) {
    const int64_t base = ((int64_t)(region * 2 + (seq & 1))) * x.world * x.cap + idx;
    const unsigned long long w = xgmi_word(g, seq);
#pragma unroll
    for (int r = 0; r < XGMI_MAX; ++r)
      if (r < x.world && r != x.rank)
        __hip_atomic_store(x.data[r] + base + (int64_t)x.rank * x.cap, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long* inbox = x.data[x.rank] + base;
    int32_t* status = x.flags[x.rank];
    unsigned long long v[XGMI_MAX];
    bool wait = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0;
    const uint64_t t0 = wall_clock64();
    for (;;) {
      bool all = true;
#pragma unroll
      for (int r = 0; r < XGMI_MAX; ++r) {
        v[r] = (r < x.world && r != x.rank) ? __hip_atomic_load(inbox + (int64_t)r * x.cap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                            : w;
        all = all && (int)(v[r] >> 32) == seq;
      }
      if (all || !wait) break;
      __builtin_amdgcn_s_sleep(2);
      if ((int64_t)(wall_clock64() - t0) > x.timeout_ticks) {
        __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        wait = false;
      }
    }
#pragma unroll
    for (int r = 0; r < XGMI_MAX; ++r)
      if (r < x.world) s += __builtin_bit_cast(float, (unsigned)(v[r] & 0xffffffffull));
  }
  return s;
}

}  // namespace dmf

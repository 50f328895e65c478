// dmf_xgmi.h — device side of the one-shot gradient exchange over xGMI (include/dmf.h "dmf_xgmi_*").
//
// Memory layout (per rank; both buffers are allocated UNCACHED by their owner, who is the only one that READS them):
//   data  : [region 2][parity 2][src rank world][cap] float   inbox; region 0 = training exchange, 1 = dmf_xgmi_allreduce
//   flags : [region 2][src rank world][nblk] int32, then one status word
// Protocol for block `blk` at sequence number `seq` (monotonic, identical on all ranks) — a PUSH exchange:
//   1. write own values into data[r][region][seq&1][rank] of EVERY rank r      (system-scope stores through the IPC mapping)
//   2. one system-scope release per block, then flags[peer][region][rank][blk] = seq on every peer
//   3. wait until flags[rank][region][peer][blk] >= seq for every peer         (relaxed polls, bounded by a wall-clock timeout)
//   4. read the world values from the OWN inbox and add them in rank order
// Reads only ever go to memory the reader allocated itself as uncached, so they can never be served from a stale cache
// line.  An IPC import does not carry the exporter's uncached attribute: the first (pull) form of this exchange read the
// peers' buffers through the imported mapping, those loads were cached in the READER's XCD L2, and the third round — the
// first one to reuse a parity slot — summed the slot's round-1 content for every element (gpurun_out/dp3r.log of round 1:
// all 1000 elements off by the known-answer pattern 3 * (i % 97)).  Writes through the imported mapping are made
// visible by the system-scope release.  A peer can run at most one sequence number ahead (it needs this rank's flag to
// finish the next one), which is what the two parities are for.  Blocks never wait for other blocks of the same grid.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dmf {

constexpr int XGMI_MAX = 16;

struct XgmiDev {
  int world, rank;
  int64_t cap;
  int nblk;
  int64_t timeout_ticks;            // wall_clock64 ticks (100 MHz)
  float* data[XGMI_MAX];
  int32_t* flags[XGMI_MAX];
};

__host__ __device__ inline int xgmi_nblk(int64_t cap) { return (int)((cap + 15) / 16); }
__host__ __device__ inline int64_t xgmi_status_index(int world, int nblk) { return (int64_t)2 * world * nblk; }

// All 256 threads of the block call this (it contains block barriers).  Returns the rank-ordered sum for
// threads with valid == true.
__device__ __forceinline__ float xgmi_exchange(const XgmiDev& x, int region, int seq, int blk, int64_t idx,
                                               bool valid, float g) {
  const int tid = threadIdx.x;
  const int64_t base = ((int64_t)(region * 2 + (seq & 1))) * x.world * x.cap;
  if (valid) {
#pragma unroll
    for (int r = 0; r < XGMI_MAX; ++r)
      if (r < x.world)
        __hip_atomic_store(x.data[r] + base + (int64_t)x.rank * x.cap + idx, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // Cache maintenance is the expensive part of any cross-agent handshake on this part (a device-scope release /
  // acquire pair costs ~10 us when every thread does it, tools/gridbar.hip), so it is done ONCE per block: the value
  // stores above are write-through system-scope stores; EVERY storing wave waits for the acknowledgement of its own
  // stores (s_waitcnt vmcnt(0): the workgroup barrier alone is a bare s_barrier on gfx950 and does not wait for them —
  // without this wait the flag raised by wave 0 could overtake the data of waves 1..3); thread 0 then issues one
  // system-scope release and raises the flags with plain system-scope stores.  Waiting polls with relaxed loads; the
  // values are read from the reader's own UNCACHED inbox, so no acquire invalidation is needed on this side.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
#pragma unroll 1
    for (int r = 0; r < x.world; ++r)
      if (r != x.rank)
        __hip_atomic_store(x.flags[r] + ((int64_t)(region * x.world + x.rank)) * x.nblk + blk, seq, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (tid < x.world && tid != x.rank) {
    int32_t* mine = x.flags[x.rank] + ((int64_t)(region * x.world + tid)) * x.nblk + blk;
    int32_t* status = x.flags[x.rank] + xgmi_status_index(x.world, x.nblk);
    if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
      const uint64_t t0 = wall_clock64();
      while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
        __builtin_amdgcn_s_sleep(4);
        if ((int64_t)(wall_clock64() - t0) > x.timeout_ticks) {
          __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
      }
    }
  }
  __syncthreads();
  float s = 0.f;
  if (valid) {
    const float* inbox = x.data[x.rank] + base + idx;
    float v[XGMI_MAX];
#pragma unroll
    for (int r = 0; r < XGMI_MAX; ++r)
      v[r] = (r < x.world) ? __hip_atomic_load(inbox + (int64_t)r * x.cap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.f;
#pragma unroll
    for (int r = 0; r < XGMI_MAX; ++r)
      if (r < x.world) s += v[r];
  }
  return s;
}

}  // namespace dmf

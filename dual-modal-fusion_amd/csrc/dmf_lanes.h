// Lane-level helpers shared by the patch kernels: wavefront / 16-lane-row reductions by DPP and the gfx950
// row / half swaps, the LDS-only workgroup barrier, and the constant-address-space view of theta.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace dmf {

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global STORE
// (vmcnt counts stores on gfx950) because of its release fence; inside the patch loop the barriers only hand LDS data
// between waves, so the round trip of global stores must not sit on the critical path.
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// Stops the compiler from hoisting per-thread address arithmetic out of the patch loop (it then spills it).
#define OPAQUE(v) asm volatile("" : "+v"(v))

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// Row reductions by DPP (VALU modifiers, no LDS crossbar round trip): xor-1 / xor-2 by quad_perm, then
// row_half_mirror (i <-> 7-i) and row_mirror (i <-> 15-i) complete the butterfly inside 8 / 16 lanes.
#define DMF_DPP_ADD(v, CTRL) \
  ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (CTRL), 0xF, 0xF, true)))
__device__ __forceinline__ float sum8(float v) {          // over the 8 lanes sharing lane>>3
  v = DMF_DPP_ADD(v, 0xB1);     // quad_perm [1,0,3,2]
  v = DMF_DPP_ADD(v, 0x4E);     // quad_perm [2,3,0,1]
  v = DMF_DPP_ADD(v, 0x141);    // row_half_mirror
  return v;
}
__device__ __forceinline__ float sum16(float v) {         // over the 16 lanes sharing lane>>4
  v = sum8(v);
  v = DMF_DPP_ADD(v, 0x140);    // row_mirror
  return v;
}
__device__ __forceinline__ float swap_add16(float v) {    // v[row] + v[row ^ 1] in every lane (rows of 16 lanes)
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float swap_add32(float v) {    // v[half] + v[half ^ 1] in every lane (halves of 32 lanes)
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float wave_sum_dpp(float v) {  // all 64 lanes, result in every lane, no LDS crossbar
  return swap_add32(swap_add16(sum16(v)));
}
// over the 8 lanes sharing lane&7 (stride 8): xor-8 inside a row by row_ror:8, xor-16 / xor-32 by the gfx950
// row / half swaps (v_permlane16_swap, v_permlane32_swap)
__device__ __forceinline__ float sum_hi8(float v) {
  v = DMF_DPP_ADD(v, 0x128);    // row_ror:8
  {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
  }
  {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    v = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
  }
  return v;
}

// theta is never written during a launch: reading it through the constant address space lets the compiler use
// scalar loads (SGPR operands) wherever the index is wave-uniform.
typedef const float __attribute__((address_space(4))) cfloat;

}  // namespace dmf

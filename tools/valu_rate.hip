// micro-benchmark: vector-instruction issue cost on one CU of gfx950, by instruction kind and waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
// One workgroup of 4*k wavefronts (k per SIMD); every wave runs N independent-chain instructions of one kind between two
// s_memtime stamps.  Printed: cycles per instruction per WAVE, and per SIMD (= per wave / k): the issue slot price that
// bounds a kernel whose waves are all VALU-busy.  Used to budget the patch kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define REP 64
#define BODY8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

template <int KIND>
__global__ void rate_kernel(unsigned long long* out, float* sink, float seed) {
  float a[8], b[8];
  float2 p[8], q[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; b[i] = seed * 0.5f + i; p[i] = make_float2(a[i], b[i]); q[i] = make_float2(b[i], a[i]); }
  const float w = seed * 0.25f;
  const float2 w2 = make_float2(w, w);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < REP; ++it) {
    if (KIND == 0) {          // v_fma_f32, 8 independent chains
#define OP(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(w));
      BODY8(OP) BODY8(OP)
#undef OP
    } else if (KIND == 1) {   // v_pk_fma_f32
#define OP(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(q[i]), "v"(w2));
      BODY8(OP) BODY8(OP)
#undef OP
    } else if (KIND == 2) {   // v_mov_b32 dpp row_shr:1
#define OP(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a[i]) : "v"(b[i]));
      BODY8(OP) BODY8(OP)
#undef OP
    } else if (KIND == 3) {   // v_fmac_f32 with a DPP source
#define OP(i) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(b[i]), "v"(w));
      BODY8(OP) BODY8(OP)
#undef OP
    } else if (KIND == 4) {   // v_add_f32 dpp row_mirror (the reduction step)
#define OP(i) asm volatile("v_add_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a[i]) : "v"(b[i]));
      BODY8(OP) BODY8(OP)
#undef OP
    } else if (KIND == 5) {   // v_cndmask_b32 (vcc)
#define OP(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(b[i]), "v"(w) : "vcc");
      BODY8(OP) BODY8(OP)
#undef OP
    } else if (KIND == 6) {   // dependent v_fmac chain (latency)
#define OP(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[0]) : "v"(b[i]), "v"(w));
      BODY8(OP) BODY8(OP)
#undef OP
    } else if (KIND == 7) {   // v_cmp_lt_f32 + v_cndmask pair (ReLU gate)
#define OP(i) asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_cndmask_b32 %0, 0, %2, vcc" : "=v"(a[i]) : "v"(b[i]), "v"(w) : "vcc");
      BODY8(OP)
#undef OP
    } else if (KIND == 8) {   // v_pk_mul_f32
#define OP(i) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p[i]) : "v"(q[i]), "v"(w2));
      BODY8(OP) BODY8(OP)
#undef OP
    } else if (KIND == 10) {  // v_mov_b32 dpp wave_shr:1 (whole-wave shift by one lane)
#define OP(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a[i]) : "v"(b[i]));
      BODY8(OP) BODY8(OP)
#undef OP
    } else if (KIND == 11) {  // v_fmac_f32 dpp wave_shr:1
#define OP(i) asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(b[i]), "v"(w));
      BODY8(OP) BODY8(OP)
#undef OP
    } else if (KIND == 12) {  // ds_bpermute_b32 (8 in flight) + add
      int idx = ((threadIdx.x + 4) & 63) * 4;
      float r[8];
#define OP(i) r[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(idx, __builtin_bit_cast(int, b[i])));
      BODY8(OP)
#undef OP
#define OP(i) a[i] += r[i];
      BODY8(OP)
#undef OP
    } else if (KIND == 9) {   // ds_read_b128 broadcast-free (each lane its own 16 bytes), 8 in flight
      extern __shared__ float4 lds[];
      float4 r[8];
#define OP(i) r[i] = lds[(threadIdx.x + 64 * i) & 1023];
      BODY8(OP)
#undef OP
#define OP(i) a[i] += r[i].x + r[i].w;
      BODY8(OP)
#undef OP
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  sink[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int per_iter, unsigned long long* d_out, float* d_sink) {
  printf("%-34s", name);
  for (int k = 1; k <= 4; ++k) {
    const int waves = 4 * k;
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(1), dim3(64 * waves), 16384, 0, d_out, d_sink, 1.5f);
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(1), dim3(64 * waves), 16384, 0, d_out, d_sink, 1.5f);
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, d_out, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost);
    double mx = 0;
    for (int w = 0; w < waves; ++w) mx = h[w] > mx ? (double)h[w] : mx;
    const double per_wave = mx / (REP * (double)per_iter);
    printf("  k=%d: %6.2f /wave %5.2f /SIMD", k, per_wave, per_wave / k);
  }
  printf("\n");
}

int main() {
  unsigned long long* d_out; float* d_sink;
  hipMalloc(&d_out, 16 * sizeof(unsigned long long));
  hipMalloc(&d_sink, 1024 * sizeof(float));
  printf("cycles per instruction (slowest wave), k waves per SIMD on one CU\n");
  run<0>("v_fmac_f32", 16, d_out, d_sink);
  run<1>("v_pk_fma_f32", 16, d_out, d_sink);
  run<2>("v_mov_b32_dpp row_shr:1", 16, d_out, d_sink);
  run<3>("v_fmac_f32_dpp row_shr:1", 16, d_out, d_sink);
  run<4>("v_add_f32_dpp row_mirror", 16, d_out, d_sink);
  run<5>("v_cndmask_b32", 16, d_out, d_sink);
  run<6>("v_fmac_f32 dependent chain", 16, d_out, d_sink);
  run<7>("v_cmp + v_cndmask pair", 8, d_out, d_sink);
  run<8>("v_pk_mul_f32", 16, d_out, d_sink);
  run<9>("ds_read_b128 (8 in flight) + 2 add", 8, d_out, d_sink);
  run<10>("v_mov_b32_dpp wave_shr:1", 16, d_out, d_sink);
  run<11>("v_fmac_f32_dpp wave_shr:1", 16, d_out, d_sink);
  run<12>("ds_bpermute_b32 + add", 8, d_out, d_sink);
  return 0;
}

// dmf_attention.hip — cross-modal attention + head, forward (the `gmf.attention: 1` network; BASELINE configs[2]).
//
//   Ta' = Ta + bf16(O) bf16(Wo)^T,   O_h = softmax(bf16(Q_h / sqrt(dh)) bf16(K_h)^T) V_h,
//   Q = bf16(Ta) bf16(Wq)^T,  K = bf16(Tb) bf16(Wk)^T,  V = bf16(Tb) bf16(Wv)^T           (oracle/gmfnet_ref.py::attention)
// Tokens are the P*P pixels of a patch (121, padded to 128), E = heads x 32.  Every contraction — the three
// projections, Q K^T, P V and the output projection — runs on the matrix cores (`v_mfma_f32_16x16x32_bf16`, bf16
// operands, fp32 accumulate); softmax statistics are fp32, reduced inside the 16-lane MFMA column groups by DPP.
// Because the network reads the attended map only through the anchor pooling, the kernel never materialises Ta':
//   za[f] += sum_t w[t] * (bf16(O) bf16(Wo)^T)[t][f],  then fc1 / fc2 / argmax as in the fused patch kernel.
//
// One 512-thread workgroup per patch; wave w owns token rows 16w..16w+15 of every product.  All operands live in LDS
// as bf16 with k contiguous ("NT" form: C[m][n] = sum_k A[m][k] Bn[n][k]) and row strides padded so that the 16-byte
// fragment reads of a 16-lane group hit 64 distinct banks.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/dmf.h"
#include "dmf_shapes.h"

namespace dmf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bfs;   // bf16 storage

struct AttnArgs {
  const bfs* tokA;      // [B][128][64]
  const bfs* tokB;
  const float* zin;     // [B][2F] pooled features before attention
  const float* theta;
  const float* pool;    // [P*P]
  float* logits;        // [B][K]
  int32_t* pred;        // [B] or null
  int64_t oWq, oWk, oWv, oWo, oFc1w, oFc1b, oFc2w, oFc2b;
  int32_t B, K;
};

__device__ __forceinline__ bfs f2bf(float x) { return __builtin_bit_cast(bfs, (__bf16)x); }

#define ATT_DPP(v, CTRL, OP) \
  OP((v), __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (CTRL), 0xF, 0xF, true)))
__device__ __forceinline__ float row16_max(float v) {      // over the 16 lanes sharing lane>>4
  v = ATT_DPP(v, 0xB1, fmaxf); v = ATT_DPP(v, 0x4E, fmaxf); v = ATT_DPP(v, 0x141, fmaxf); v = ATT_DPP(v, 0x140, fmaxf);
  return v;
}
#define ATT_ADD(a, b) ((a) + (b))
__device__ __forceinline__ float row16_sum(float v) {
  v = ATT_DPP(v, 0xB1, ATT_ADD); v = ATT_DPP(v, 0x4E, ATT_ADD); v = ATT_DPP(v, 0x141, ATT_ADD); v = ATT_DPP(v, 0x140, ATT_ADD);
  return v;
}
__device__ __forceinline__ float xrow_sum(float v) {       // over the 4 lanes sharing lane&15 (the four 16-lane rows)
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

// fragment of a k-contiguous bf16 matrix M[row][k] (row stride RS halves): rows r0..r0+15, k0..k0+31
template <int RS>
__device__ __forceinline__ bf16x8 frag(const bfs* M, int r0, int k0, int lane) {
  return *reinterpret_cast<const bf16x8*>(M + (r0 + (lane & 15)) * RS + k0 + 8 * (lane >> 4));
}

template <class Sh, int E, int NH>
__global__ __launch_bounds__(512) void attn_head_kernel(const AttnArgs a) {
  constexpr int T = 128, FP = 64, DH = 32, NT = 512;
  constexpr int F = Sh::F, F2 = Sh::F2, H = Sh::H, P2 = Sh::P2;
  constexpr int FO = (F + 15) / 16 * 16;            // output-projection columns, padded to whole tiles
  constexpr int NFT = FO / 16;
  static_assert(E == NH * DH && F <= FP && P2 <= T && H == 64, "attention geometry");
  // padded row strides (halves): 16-byte reads of 16 consecutive rows must land on distinct bank quads
  constexpr int TS = 72, KS = 40, VS = 136, WS = 72, OS = 40, PS = 136, QS = 40;
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  bfs* sTa = reinterpret_cast<bfs*>(smraw);         // [T][TS]
  bfs* sTb = sTa + T * TS;
  bfs* sK = sTb + T * TS;                           // [T][KS]      K_h
  bfs* sVt = sK + T * KS;                           // [DH][VS]     V_h transposed
  bfs* sWq = sVt + DH * VS;                         // [DH][WS]     rows of Wq / Wk / Wv of this head
  bfs* sWk = sWq + DH * WS;
  bfs* sWv = sWk + DH * WS;
  bfs* sWo = sWv + DH * WS;                         // [FO][OS]     Wo[:, head]
  bfs* sP = sWo + FO * OS;                          // [8 waves][16][PS]
  bfs* sQ = sP + 8 * 16 * PS;                       // [8 waves][16][QS]  Qs strip, later the O_h strip
  float* sZ = reinterpret_cast<float*>(sQ + 8 * 16 * QS);   // [F2]
  float* sHd = sZ + ((F2 + 3) & ~3);                // [H]
  float* sLg = sHd + H;                             // [KMAX]
  float* sPw = sLg + KMAX;                          // [T] pooling weights, zero beyond P2
  float* sDz = sPw + T;                             // [8][FO]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* __restrict__ th = a.theta;
  const int K = a.K;
  const int m0 = wave * 16;                         // this wave's token rows
  const float scale = 0.17677669529663687f;         // 1/sqrt(32), as float32(1/math.sqrt(32))

  for (int i = tid; i < T; i += NT) sPw[i] = i < P2 ? a.pool[i] : 0.f;

  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    // ---- tokens -> LDS (16-byte pieces, rows re-strided to TS)
    for (int i = tid; i < 2 * T * 8; i += NT) {
      const int mp = i >> 10, rem = i & 1023, t = rem >> 3, pc = rem & 7;
      const uint4 v = *reinterpret_cast<const uint4*>((mp ? a.tokB : a.tokA) + ((size_t)b * T + t) * FP + pc * 8);
      *reinterpret_cast<uint4*>((mp ? sTb : sTa) + t * TS + pc * 8) = v;
    }
    if (tid < F2) sZ[tid] = a.zin[(size_t)b * F2 + tid];
    f32x4 accO[NFT];                                // (O Wo^T)[rows m0.., cols 16n..] accumulated over heads
#pragma unroll
    for (int n = 0; n < NFT; ++n) accO[n] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int h = 0; h < NH; ++h) {
      __syncthreads();                              // previous head's K/V/weights are no longer read; tokens landed
      // ---- this head's weight slices, fp32 -> bf16
      for (int i = tid; i < 3 * DH * FP; i += NT) {
        const int wsel = i / (DH * FP), rem = i - wsel * (DH * FP), d = rem / FP, f = rem - d * FP;
        const int64_t o = wsel == 0 ? a.oWq : (wsel == 1 ? a.oWk : a.oWv);
        const float v = f < F ? th[o + (int64_t)(h * DH + d) * F + f] : 0.f;
        (wsel == 0 ? sWq : (wsel == 1 ? sWk : sWv))[d * WS + f] = f2bf(v);
      }
      for (int i = tid; i < FO * DH; i += NT) {
        const int f = i / DH, d = i - f * DH;
        sWo[f * OS + d] = f2bf(f < F ? th[a.oWo + (int64_t)f * E + h * DH + d] : 0.f);
      }
      __syncthreads();
      // ---- projections of this wave's 16 tokens: Q_h, K_h, V_h  = tokens[16 x 64] x W_h^T[64 x 32]
      {
        f32x4 q[2], k[2], v[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) { q[n] = k[n] = v[n] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < FP / 32; ++ks) {
          const bf16x8 fa = frag<TS>(sTa, m0, 32 * ks, lane);
          const bf16x8 fb = frag<TS>(sTb, m0, 32 * ks, lane);
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            q[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, frag<WS>(sWq, 16 * n, 32 * ks, lane), q[n], 0, 0, 0);
            k[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, frag<WS>(sWk, 16 * n, 32 * ks, lane), k[n], 0, 0, 0);
            v[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, frag<WS>(sWv, 16 * n, 32 * ks, lane), v[n], 0, 0, 0);
          }
        }
        // C layout: lane holds rows 4*(lane>>4) + r, column lane&15 of each 16x16 tile
        const int col = lane & 15, rb = 4 * (lane >> 4);
        bfs* qrow = sQ + wave * 16 * QS;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            qrow[(rb + r) * QS + 16 * n + col] = f2bf(q[n][r] * scale);
            sK[(m0 + rb + r) * KS + 16 * n + col] = f2bf(k[n][r]);
          }
          // V transposed: 4 consecutive tokens of one feature -> one 8-byte store
          const uint2 pk = make_uint2((uint32_t)f2bf(v[n][0]) | ((uint32_t)f2bf(v[n][1]) << 16),
                                      (uint32_t)f2bf(v[n][2]) | ((uint32_t)f2bf(v[n][3]) << 16));
          *reinterpret_cast<uint2*>(sVt + (16 * n + col) * VS + m0 + rb) = pk;
        }
      }
      __syncthreads();                              // K_h, V_h of all 128 tokens are in LDS
      // ---- S = Qs K^T for this wave's 16 rows x 128 keys, softmax in registers
      {
        const bfs* qrow = sQ + wave * 16 * QS;
        const bf16x8 fq = frag<QS>(qrow, 0, 0, lane);
        f32x4 s[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq, frag<KS>(sK, 16 * j, 0, lane), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        const int col = lane & 15, rb = 4 * (lane >> 4);
        bfs* prow = sP + wave * 16 * PS;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float mx = -INFINITY;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (16 * j + col >= P2) s[j][r] = -INFINITY;       // padded keys
            mx = fmaxf(mx, s[j][r]);
          }
          mx = row16_max(mx);
          float sum = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) { s[j][r] = expf(s[j][r] - mx); sum += s[j][r]; }
          sum = row16_sum(sum);
#pragma unroll
          for (int j = 0; j < 8; ++j) prow[(rb + r) * PS + 16 * j + col] = f2bf(s[j][r] / sum);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();              // the P strip and the Q strip are private to this wave
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // ---- O_h = P V (16 x 32), then (O Wo^T) partial
      {
        const bfs* prow = sP + wave * 16 * PS;
        f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < T / 32; ++ks) {
          const bf16x8 fp = frag<PS>(prow, 0, 32 * ks, lane);
#pragma unroll
          for (int n = 0; n < 2; ++n)
            o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fp, frag<VS>(sVt, 16 * n, 32 * ks, lane), o[n], 0, 0, 0);
        }
        const int col = lane & 15, rb = 4 * (lane >> 4);
        bfs* orow = sQ + wave * 16 * QS;            // reuse the Q strip (its last reader was this wave's S product)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) orow[(rb + r) * QS + 16 * n + col] = f2bf(o[n][r]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const bf16x8 fo = frag<QS>(orow, 0, 0, lane);
#pragma unroll
        for (int n = 0; n < NFT; ++n)
          accO[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fo, frag<OS>(sWo, 16 * n, 0, lane), accO[n], 0, 0, 0);
      }
    }
    // ---- pooled correction: dz[f] = sum_t w[t] (O Wo^T)[t][f]
    {
      const int col = lane & 15, rb = 4 * (lane >> 4);
#pragma unroll
      for (int n = 0; n < NFT; ++n) {
        float p = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) p = fmaf(sPw[m0 + rb + r], accO[n][r], p);
        p = xrow_sum(p);                            // over the four row groups of this wave
        if (lane < 16) sDz[wave * FO + 16 * n + col] = p;
      }
    }
    __syncthreads();
    if (tid < F) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += sDz[w * FO + tid];
      sZ[tid] += s;
    }
    __syncthreads();
    // ---- head: fc1 + ReLU, fc2, argmax (thread <-> (row j, part p), column p + 8m)
    {
      const int j = tid >> 3, pp = tid & 7;
      float acc = 0.f;
      for (int i = pp; i < F2; i += 8) acc = fmaf(th[a.oFc1w + (int64_t)j * F2 + i], sZ[i], acc);
      acc = ATT_DPP(acc, 0xB1, ATT_ADD); acc = ATT_DPP(acc, 0x4E, ATT_ADD); acc = ATT_DPP(acc, 0x141, ATT_ADD);
      if (pp == 0) sHd[j] = fmaxf(acc + th[a.oFc1b + j], 0.f);
    }
    __syncthreads();
    {
      const int k = tid >> 3, pp = tid & 7;
      float acc = 0.f;
      if (k < K)
        for (int j = pp; j < H; j += 8) acc = fmaf(th[a.oFc2w + (int64_t)k * H + j], sHd[j], acc);
      acc = ATT_DPP(acc, 0xB1, ATT_ADD); acc = ATT_DPP(acc, 0x4E, ATT_ADD); acc = ATT_DPP(acc, 0x141, ATT_ADD);
      if (k < K && pp == 0) sLg[k] = acc + th[a.oFc2b + k];
    }
    __syncthreads();
    if (wave == 0) {
      const float v = lane < K ? sLg[lane] : -INFINITY;
      float mx = v;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      if (lane < K) a.logits[(size_t)b * K + lane] = v;
      if (a.pred != nullptr) {
        const unsigned long long bal = __ballot(v == mx);
        if (lane == 0) a.pred[b] = __ffsll((long long)bal) - 1;
      }
    }
    __syncthreads();
  }
}

template <class Sh, int E, int NH>
static size_t attn_lds_bytes() {
  constexpr int T = 128, DH = 32;
  constexpr int FO = (Sh::F + 15) / 16 * 16;
  size_t halves = 2 * T * 72 + T * 40 + DH * 136 + 3 * DH * 72 + FO * 40 + 8 * 16 * 136 + 8 * 16 * 40;
  size_t floats = ((Sh::F2 + 3) & ~3) + Sh::H + KMAX + T + 8 * FO;
  return halves * 2 + floats * 4;
}

using ShapeHSI = Shape<200, 1, 11, 1, 40, 10, 64>;
using ShapeTiny1 = Shape<8, 1, 5, 1, 40, 2, 64>;

template <class Sh>
static hipError_t launch_attn(const AttnArgs& a, hipStream_t st) {
  static bool done = false;
  const size_t bytes = attn_lds_bytes<Sh, 96, 3>();
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_head_kernel<Sh, 96, 3>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    done = true;
  }
  const int grid = a.B < 1024 ? a.B : 1024;
  hipLaunchKernelGGL((attn_head_kernel<Sh, 96, 3>), dim3(grid), dim3(512), bytes, st, a);
  return hipGetLastError();
}

int attn_shape_supported(const dmf_shape& s) {
  if (s.E != 96 || s.heads != 3 || s.F != 40 || s.H != 64) return 0;
  return (s.C == 200 && s.C2 == 1 && s.P == 11 && s.S == 1 && s.G == 10) || (s.C == 8 && s.C2 == 1 && s.P == 5 && s.S == 1 && s.G == 2);
}

hipError_t attn_dispatch(const dmf_shape& s, const AttnArgs& a, hipStream_t st) {
  if (s.C == 200) return launch_attn<ShapeHSI>(a, st);
  if (s.C == 8) return launch_attn<ShapeTiny1>(a, st);
  return hipErrorInvalidValue;
}

}  // namespace dmf

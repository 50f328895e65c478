// dmf_attention_train.hip — cross-modal attention + head, forward AND backward in one launch (training with
// `gmf.attention: 1`; BASELINE configs[2]).  Sits between two launches of the fused patch kernel:
//   patch_kernel<MODE_TOKENS>  conv stages -> bf16 token maps Ta, Tb [B][128][64] + pooled z before attention
//   attn_train_kernel (here)   attention + head forward, CE, head backward, attention backward:
//                              logits / loss, head vectors for the gradient reduce, attention weight gradients
//                              (per-workgroup slabs) and the DENSE maps dL/dYa, dL/dYb [B][F][P2]
//   patch_kernel<MODE_DENSE>   conv backward from those maps
//
// Arithmetic (oracle/gmfnet_ref.py::attention, gradients = torch autograd through it): every forward contraction has
// bf16 operands and fp32 accumulation; the roundings are straight-through, so every backward contraction multiplies a
// SAVED bf16 operand with an fp32 upstream gradient.  The matrix cores take bf16 only, so an fp32 gradient operand g is
// fed as hi = bf16(g), lo = bf16(g - hi) in two MFMAs (relative error ~2^-17).
//
// What the pooling makes cheap.  The network reads Ta' = Ta + bf16(O) bf16(Wo)^T only through z_a = sum_t w_t Ta'[t],
// so dTa'[t][f] = w_t dza[f] is rank one and, per head (u = bf16(Wo_h)^T dza, P = softmax, all [T x T] maps stay in
// registers):
//   dO = w (x) u                          dWo_h = dza (x) obar_h,    obar_h = sum_t w_t bf16(O_h)[t]
//   dV = c (x) u,  c = bf16(P)^T w        dWv_h = u (x) bbar,        bbar = bf16(Tb)^T c;     dTb += c (x) (bf16(Wv_h)^T u)
//   dS[t][j] = w_t P[t][j] (a_j - abar_t),  a = bf16(V) u,  abar = P a            (softmax backward, fp32)
//   dQs = dS bf16(K),  dK = dS^T bf16(Qs)                                         (MFMA, hi/lo)
//   dTa += scale dQs bf16(Wq_h),  dTb += dK bf16(Wk_h)                            (MFMA, hi/lo)
//   dWq_h = scale dQs^T bf16(Ta),  dWk_h = dK^T bf16(Tb)                          (MFMA, hi/lo, k = all 128 tokens)
//
// One 512-thread workgroup per patch, wave w owns queries (and, for dK, keys) 16w..16w+15.  Products are formed
// TRANSPOSED (C[m = key or feature][n = own token]) so that a result in the MFMA C layout — lane holds rows
// 4*(lane>>4)+r, column lane&15 — is directly the B operand of the next product (n = own token, k = the row index):
// S^T -> P^T -> O^T -> (O Wo^T)^T and dS^T -> dQs^T -> dTa^T never visit LDS.  Two C tiles i0, i1 give lane group g
// the k values {16 i0 + 4g + r} U {16 i1 + 4g + r}; the LDS operand is read with the same permutation (frag2).
// Attention weight gradients accumulate in registers over the workgroup's patches and leave once, as a slab.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/dmf.h"
#include "dmf_shapes.h"

namespace dmf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bfs;

struct AttnTrainArgs {   // must match dmf_capi.hip
  const bfs* tokA; const bfs* tokB;     // [B][128][64]
  const float* zin;                     // [B][2F]
  const float* theta; const float* pool;
  const int32_t* labels; const int32_t* cursor;   // labels[(*cursor) * B + b]   (cursor may be null)
  const float* dlogits;                 // used when labels == null: caller-supplied dL/dlogits [B][K]
  float loss_scale;
  float* logits; float* loss;           // [B][K], [B] (loss may be null)
  float* ws_z; float* ws_h; float* ws_dh; float* ws_dl;   // head vectors for the gradient reduce
  float* dYa; float* dYb;               // [B][F][P2]
  float* aslab;                         // [gridDim][4*E*F] attention weight gradients (Wq, Wk, Wv, Wo)
  int64_t oWq, oWk, oWv, oWo, oFc1w, oFc1b, oFc2w, oFc2b;
  int32_t B, K;
};

namespace at {

__device__ __forceinline__ bfs f2bf(float x) { return __builtin_bit_cast(bfs, (__bf16)x); }
__device__ __forceinline__ float bf2f(__bf16 x) { return (float)x; }

#define AT_DPP(v, CTRL, OP) \
  OP((v), __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (CTRL), 0xF, 0xF, true)))
#define AT_ADD(a, b) ((a) + (b))
__device__ __forceinline__ float row16_sum(float v) {      // over the 16 lanes sharing lane>>4 (the token index)
  v = AT_DPP(v, 0xB1, AT_ADD); v = AT_DPP(v, 0x4E, AT_ADD); v = AT_DPP(v, 0x141, AT_ADD); v = AT_DPP(v, 0x140, AT_ADD);
  return v;
}
__device__ __forceinline__ float swap16(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)(r[0] ^ r[1] ^ u));   // the value that is not mine
}
// reductions over the 4 lanes sharing lane&15 (the four 16-lane groups)
__device__ __forceinline__ float xrow_sum(float v) {
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
__device__ __forceinline__ float xrow_max(float v) {
  {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
  }
  {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    v = fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
  }
  return v;
}

// standard fragment of a k-contiguous bf16 matrix M[row][k]: rows r0..r0+15, k0 + 8*(lane>>4) .. +7
template <int RS>
__device__ __forceinline__ bf16x8 frag(const bfs* M, int r0, int k0, int lane) {
  return *reinterpret_cast<const bf16x8*>(M + (r0 + (lane & 15)) * RS + k0 + 8 * (lane >> 4));
}
// permuted fragment pairing with a register operand made of two C tiles: k = kA + 4g + {0..3}, kB + 4g + {0..3}
template <int RS>
__device__ __forceinline__ bf16x8 frag2(const bfs* M, int r0, int kA, int kB, int lane) {
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
  const bfs* row = M + (r0 + (lane & 15)) * RS + 4 * (lane >> 4);
  const bf16x4 a = *reinterpret_cast<const bf16x4*>(row + kA);
  const bf16x4 b = *reinterpret_cast<const bf16x4*>(row + kB);
  return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
// fragment of tokens[t][f] (k = f) gathered from the TRANSPOSED map MT[f][t]: row t = t0 + (lane&15)
template <int RS>
__device__ __forceinline__ bf16x8 frag_t(const bfs* MT, int t0, int k0, int lane) {
  const bfs* p = MT + (k0 + 8 * (lane >> 4)) * RS + t0 + (lane & 15);
  bf16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = __builtin_bit_cast(__bf16, p[i * RS]);
  return v;
}
// register operands from two C tiles
__device__ __forceinline__ bf16x8 pack_bf(const f32x4& a, const f32x4& b) {
  return bf16x8{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
}
__device__ __forceinline__ void split_hl(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const __bf16 ha = (__bf16)a[i], hb = (__bf16)b[i];
    hi[i] = ha; hi[4 + i] = hb;
    lo[i] = (__bf16)(a[i] - (float)ha); lo[4 + i] = (__bf16)(b[i] - (float)hb);
  }
}
#define AT_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16((A), (B), (C), 0, 0, 0)

}  // namespace at

template <class Sh, int E, int NH>
struct AttnTrainLds {
  static constexpr int T = 128, FP = 64, DH = 32, FO = 48;
  static constexpr int VS = 136, QS = 40, KS = 40, WS = 72, OS = 40;
  // halves
  static constexpr int oTaT = 0, oTbT = oTaT + FP * VS, oQ = oTbT + FP * VS, oK = oQ + T * QS, oQt = oK + T * KS,
                       oKt = oQt + DH * VS, oVt = oKt + DH * VS, oWq = oVt + DH * VS, oWk = oWq + DH * WS,
                       oWv = oWk + DH * WS, oWqT = oWv + DH * WS, oWkT = oWqT + FO * OS, oWo = oWkT + FO * OS,
                       oDhi = oWo + FO * OS, oDlo = oDhi + DH * VS, HALVES = oDlo + DH * VS;
  // floats (after the halves)
  static constexpr int fZ = 0, fHd = fZ + 80, fLg = fHd + 64, fDl = fLg + 64, fDh = fDl + 64, fDz = fDh + 64,
                       fPw = fDz + 80, fU = fPw + T, fOb = fU + E, fObw = fOb + E, fDzw = fObw + 8 * E,
                       fMx = fDzw + 8 * FO, fSm = fMx + T, fAb = fSm + T, fA = fAb + T, fCw = fA + T, fC = fCw + 8 * T,
                       fG = fC + T, fBb = fG + FO, FLOATS = fBb + FO;
  static constexpr size_t BYTES = (size_t)HALVES * 2 + (size_t)FLOATS * 4;
  static_assert(HALVES % 8 == 0, "float region stays 16-byte aligned");
};

template <class Sh, int E, int NH>
__global__ __launch_bounds__(512) void attn_train_kernel(const AttnTrainArgs a) {
  using namespace at;
  using L = AttnTrainLds<Sh, E, NH>;
  constexpr int T = L::T, FP = L::FP, DH = L::DH, FO = L::FO, NT = 512;
  constexpr int VS = L::VS, QS = L::QS, KS = L::KS, WS = L::WS, OS = L::OS;
  constexpr int F = Sh::F, F2 = Sh::F2, H = Sh::H, P2 = Sh::P2;
  static_assert(E == NH * DH && F == 40 && F2 == 80 && H == 64 && P2 <= T, "attention geometry");
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  bfs* sh = reinterpret_cast<bfs*>(smraw);
  bfs *sTaT = sh + L::oTaT, *sTbT = sh + L::oTbT, *sQ = sh + L::oQ, *sK = sh + L::oK, *sQt = sh + L::oQt, *sKt = sh + L::oKt,
      *sVt = sh + L::oVt, *sWq = sh + L::oWq, *sWk = sh + L::oWk, *sWv = sh + L::oWv, *sWqT = sh + L::oWqT,
      *sWkT = sh + L::oWkT, *sWo = sh + L::oWo, *sDhi = sh + L::oDhi, *sDlo = sh + L::oDlo;
  float* sf = reinterpret_cast<float*>(sh + L::HALVES);
  float *sZ = sf + L::fZ, *sHd = sf + L::fHd, *sLg = sf + L::fLg, *sDl = sf + L::fDl, *sDh = sf + L::fDh, *sDz = sf + L::fDz,
        *sPw = sf + L::fPw, *sU = sf + L::fU, *sOb = sf + L::fOb, *sObw = sf + L::fObw, *sDzw = sf + L::fDzw, *sMx = sf + L::fMx,
        *sSm = sf + L::fSm, *sAb = sf + L::fAb, *sA = sf + L::fA, *sCw = sf + L::fCw, *sC = sf + L::fC, *sG = sf + L::fG,
        *sBb = sf + L::fBb;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, col = lane & 15;
  const int m0 = wave * 16;                          // own queries (and own keys for dK)
  const int tq = m0 + col;                           // the token this lane's C columns belong to
  const float* __restrict__ th = a.theta;
  const int K = a.K;
  const float scale = 0.17677669529663687f;          // float32(1/sqrt(32))
  const int boff = a.cursor != nullptr ? a.cursor[0] * a.B : 0;

  for (int i = tid; i < T; i += NT) sPw[i] = i < P2 ? a.pool[i] : 0.f;

  // attention weight gradients, accumulated over this workgroup's patches
  f32x4 accWq[NH], accWk[NH];                        // waves 0..5: tile (d tile wave/3, f tile wave%3) of every head
  float accWv[NH][3], accWo[8];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    accWq[h] = accWk[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    accWv[h][0] = accWv[h][1] = accWv[h][2] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) accWo[i] = 0.f;
  const int wmt = wave / 3, wnt = wave % 3;          // dWq / dWk tile of this wave (waves 0..5)

  // stage one head's weights: W[d][f] for the projections; for the backward also W^T[f][d]
  auto stage_weights = [&](int h, bool bwd) {
    for (int i = tid; i < 3 * DH * FP; i += NT) {
      const int wsel = i / (DH * FP), rem = i - wsel * (DH * FP), d = rem / FP, f = rem - d * FP;
      const int64_t o = wsel == 0 ? a.oWq : (wsel == 1 ? a.oWk : a.oWv);
      const bfs v = f2bf(f < F ? th[o + (int64_t)(h * DH + d) * F + f] : 0.f);
      (wsel == 0 ? sWq : (wsel == 1 ? sWk : sWv))[d * WS + f] = v;
      if (bwd && wsel < 2 && f < FO) (wsel == 0 ? sWqT : sWkT)[f * OS + d] = v;
    }
    if (!bwd)
      for (int i = tid; i < FO * DH; i += NT) {
        const int f = i / DH, d = i - f * DH;
        sWo[f * OS + d] = f2bf(f < F ? th[a.oWo + (int64_t)f * E + h * DH + d] : 0.f);
      }
  };
  // projections of the wave's 16 tokens; writes Qs (both layouts), K (both layouts), V^T
  auto project = [&]() {
    f32x4 q[2], k[2], v[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) q[n] = k[n] = v[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < FP / 32; ++ks) {
      const bf16x8 fa = frag_t<VS>(sTaT, m0, 32 * ks, lane);
      const bf16x8 fb = frag_t<VS>(sTbT, m0, 32 * ks, lane);
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        q[n] = AT_MFMA(fa, frag<WS>(sWq, 16 * n, 32 * ks, lane), q[n]);
        k[n] = AT_MFMA(fb, frag<WS>(sWk, 16 * n, 32 * ks, lane), k[n]);
        v[n] = AT_MFMA(fb, frag<WS>(sWv, 16 * n, 32 * ks, lane), v[n]);
      }
    }
    // C layout: rows (tokens) m0 + 4g + r, column (feature) 16n + col
    const int rb = 4 * g;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      bfs qh[4], kh[4], vh[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        qh[r] = f2bf(q[n][r] * scale); kh[r] = f2bf(k[n][r]); vh[r] = f2bf(v[n][r]);
        sQ[(m0 + rb + r) * QS + 16 * n + col] = qh[r];
        sK[(m0 + rb + r) * KS + 16 * n + col] = kh[r];
      }
      const int o = (16 * n + col) * VS + m0 + rb;     // 4 consecutive tokens of one feature: one 8-byte store
      *reinterpret_cast<uint2*>(sQt + o) = make_uint2((uint32_t)qh[0] | ((uint32_t)qh[1] << 16), (uint32_t)qh[2] | ((uint32_t)qh[3] << 16));
      *reinterpret_cast<uint2*>(sKt + o) = make_uint2((uint32_t)kh[0] | ((uint32_t)kh[1] << 16), (uint32_t)kh[2] | ((uint32_t)kh[3] << 16));
      *reinterpret_cast<uint2*>(sVt + o) = make_uint2((uint32_t)vh[0] | ((uint32_t)vh[1] << 16), (uint32_t)vh[2] | ((uint32_t)vh[3] << 16));
    }
  };
  // P^T for the wave's queries: s[i][r] = P[t = tq][j = 16i + 4g + r]  (fp32, keys >= P2 masked)
  auto softmax_T = [&](f32x4 (&s)[8]) {
    const bf16x8 fq = frag<QS>(sQ, m0, 0, lane);
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = AT_MFMA(frag<KS>(sK, 16 * i, 0, lane), fq, (f32x4{0.f, 0.f, 0.f, 0.f}));
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (16 * i + 4 * g + r >= P2) s[i][r] = -INFINITY;
        mx = fmaxf(mx, s[i][r]);
      }
    mx = xrow_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) { s[i][r] = expf(s[i][r] - mx); sum += s[i][r]; }
    sum = xrow_sum(sum);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) s[i][r] = s[i][r] / sum;
    if (g == 0) { sMx[tq] = mx; sSm[tq] = sum; }
  };

  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    // ------------------------------------------------------------------ tokens -> LDS, transposed [f][t]
    __syncthreads();
    for (int i = tid; i < 2 * 8 * T; i += NT) {
      const int mp = i >> 10, rem = i & 1023, pc = rem >> 7, t = rem & 127;
      const uint4 v = *reinterpret_cast<const uint4*>((mp ? a.tokB : a.tokA) + ((size_t)b * T + t) * FP + pc * 8);
      bfs* dst = (mp ? sTbT : sTaT) + (pc * 8) * VS + t;
      dst[0 * VS] = (bfs)(v.x & 0xffff); dst[1 * VS] = (bfs)(v.x >> 16);
      dst[2 * VS] = (bfs)(v.y & 0xffff); dst[3 * VS] = (bfs)(v.y >> 16);
      dst[4 * VS] = (bfs)(v.z & 0xffff); dst[5 * VS] = (bfs)(v.z >> 16);
      dst[6 * VS] = (bfs)(v.w & 0xffff); dst[7 * VS] = (bfs)(v.w >> 16);
    }
    if (tid < F2) sZ[tid] = a.zin[(size_t)b * F2 + tid];

    // ================================================================== pass 1: forward
    f32x4 accO[3];                                   // (O Wo^T)^T[f = 16n + 4g + r][t = tq], summed over heads
#pragma unroll
    for (int n = 0; n < 3; ++n) accO[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float wq = sPw[tq];                        // (sPw was written before the first barrier of this iteration)
    for (int h = 0; h < NH; ++h) {
      __syncthreads();
      stage_weights(h, false);
      __syncthreads();
      project();
      __syncthreads();
      f32x4 s[8];
      softmax_T(s);
      f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};   // O^T[d = 16mt + 4g + r][t = tq]
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 pb = pack_bf(s[2 * ks], s[2 * ks + 1]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) o[mt] = AT_MFMA(frag2<VS>(sVt, 16 * mt, 32 * ks, 32 * ks + 16, lane), pb, o[mt]);
      }
      const bf16x8 ob = pack_bf(o[0], o[1]);
#pragma unroll
      for (int n = 0; n < 3; ++n) accO[n] = AT_MFMA(frag2<OS>(sWo, 16 * n, 0, 16, lane), ob, accO[n]);
      // obar_h[d] = sum_t w_t bf16(O)[t][d]: this wave's 16 tokens
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = row16_sum(wq * bf2f((__bf16)o[mt][r]));
          if (col == 0) sObw[wave * E + h * DH + 16 * mt + 4 * g + r] = v;
        }
    }
    // pooled correction za[f] += sum_t w_t (O Wo^T)[t][f]
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = row16_sum(wq * accO[n][r]);
        if (col == 0) sDzw[wave * FO + 16 * n + 4 * g + r] = v;
      }
    __syncthreads();
    if (tid < F) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += sDzw[w * FO + tid];
      sZ[tid] += s;
    } else if (tid >= 64 && tid < 64 + E) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += sObw[w * E + tid - 64];
      sOb[tid - 64] = s;
    }
    __syncthreads();
    // ------------------------------------------------------------------ head forward
    {
      const int j = tid >> 3, pp = tid & 7;
      float acc = 0.f;
      for (int i = pp; i < F2; i += 8) acc = fmaf(th[a.oFc1w + (int64_t)j * F2 + i], sZ[i], acc);
      acc = AT_DPP(acc, 0xB1, AT_ADD); acc = AT_DPP(acc, 0x4E, AT_ADD); acc = AT_DPP(acc, 0x141, AT_ADD);
      if (pp == 0) sHd[j] = fmaxf(acc + th[a.oFc1b + j], 0.f);
    }
    __syncthreads();
    {
      const int k = tid >> 3, pp = tid & 7;
      float acc = 0.f;
      if (k < K)
        for (int j = pp; j < H; j += 8) acc = fmaf(th[a.oFc2w + (int64_t)k * H + j], sHd[j], acc);
      acc = AT_DPP(acc, 0xB1, AT_ADD); acc = AT_DPP(acc, 0x4E, AT_ADD); acc = AT_DPP(acc, 0x141, AT_ADD);
      if (k < K && pp == 0) sLg[k] = acc + th[a.oFc2b + k];
    }
    __syncthreads();
    // ------------------------------------------------------------------ loss, dlogits, head backward
    if (wave == 0) {
      const float lg = lane < K ? sLg[lane] : -INFINITY;
      float dl = 0.f;
      if (a.labels != nullptr) {
        int label = a.labels[boff + b];
        label = label < 0 ? 0 : (label >= K ? K - 1 : label);
        float mx = lg;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        const float e = lane < K ? expf(lg - mx) : 0.f;
        float se = e;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
        dl = lane < K ? (e / se - (lane == label ? 1.f : 0.f)) * a.loss_scale : 0.f;
        const float lgt = __shfl(lg, label);
        if (a.loss != nullptr && lane == 0) a.loss[b] = (mx + logf(se)) - lgt;
      } else {
        dl = lane < K ? a.dlogits[(size_t)b * K + lane] : 0.f;
      }
      if (lane < K) a.logits[(size_t)b * K + lane] = lg;
      sDl[lane] = dl;
    }
    __syncthreads();
    if (tid < H) {
      float dh = 0.f;
      for (int k = 0; k < K; ++k) dh = fmaf(th[a.oFc2w + (int64_t)k * H + tid], sDl[k], dh);
      sDh[tid] = sHd[tid] > 0.f ? dh : 0.f;
    }
    __syncthreads();
    if (tid < F2) {
      float dz = 0.f;
      for (int j = 0; j < H; ++j) dz = fmaf(th[a.oFc1w + (int64_t)j * F2 + tid], sDh[j], dz);
      sDz[tid] = dz;
      a.ws_z[(size_t)b * F2 + tid] = sZ[tid];
    } else if (tid >= 128 && tid < 128 + H) {
      const int j = tid - 128;
      a.ws_h[(size_t)b * H + j] = sHd[j];
      a.ws_dh[(size_t)b * H + j] = sDh[j];
      a.ws_dl[(size_t)b * KMAX + j] = sDl[j];
    }
    __syncthreads();
    // u[e] = sum_f dza[f] bf16(Wo[f][e]);   dWo[f][e] += dza[f] * obar[e]
    if (tid < E) {
      float u = 0.f;
      for (int f = 0; f < F; ++f) u = fmaf(sDz[f], bf2f((__bf16)th[a.oWo + (int64_t)f * E + tid]), u);
      sU[tid] = u;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + NT * i;
      if (e < F * E) accWo[i] = fmaf(sDz[e / E], sOb[e % E], accWo[i]);
    }

    // ================================================================== pass 2: backward, head by head
    f32x4 accTa[3], accTb[3];                        // dTa^T / dTb^T [f = 16n + 4g + r][token tq]
#pragma unroll
    for (int n = 0; n < 3; ++n) accTa[n] = accTb[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      __syncthreads();
      stage_weights(h, true);
      __syncthreads();
      project();
      __syncthreads();
      // a_j = sum_d bf16(V)[j][d] u_h[d];  g_h[f] = sum_d u_h[d] bf16(Wv_h)[d][f]
      if (tid < T) {
        float s = 0.f;
#pragma unroll 8
        for (int d = 0; d < DH; ++d) s = fmaf(bf2f(__builtin_bit_cast(__bf16, sVt[d * VS + tid])), sU[h * DH + d], s);
        sA[tid] = s;
      } else if (tid < T + FO) {
        const int f = tid - T;
        float s = 0.f;
        for (int d = 0; d < DH; ++d) s = fmaf(sU[h * DH + d], bf2f(__builtin_bit_cast(__bf16, sWv[d * WS + f])), s);
        sG[f] = s;
      }
      f32x4 s[8];
      softmax_T(s);                                  // P[tq][j], j = 16i + 4g + r
      __syncthreads();                               // a_j visible
      {
        // abar_t = sum_j P[t][j] a_j ;  c_j (this wave's part) = sum_{t own} bf16(P)[t][j] w_t
        float ab = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) ab = fmaf(s[i][r], sA[16 * i + 4 * g + r], ab);
        ab = xrow_sum(ab);
        if (g == 0) sAb[tq] = ab;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float c = row16_sum(wq * bf2f((__bf16)s[i][r]));
            if (col == 0) sCw[wave * T + 16 * i + 4 * g + r] = c;
          }
        // dS^T in place: s[i][r] = w_t P (a_j - abar_t)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[i][r] = wq * s[i][r] * (sA[16 * i + 4 * g + r] - ab);
      }
      // dQs^T[d][t own] = sum_j K^T[d][j] dS[t][j]
      f32x4 dq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 hi, lo;
        split_hl(s[2 * ks], s[2 * ks + 1], hi, lo);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const bf16x8 fk = frag2<VS>(sKt, 16 * mt, 32 * ks, 32 * ks + 16, lane);
          dq[mt] = AT_MFMA(fk, hi, dq[mt]);
          dq[mt] = AT_MFMA(fk, lo, dq[mt]);
        }
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dq[mt][r] *= scale;
      {
        bf16x8 hi, lo;
        split_hl(dq[0], dq[1], hi, lo);
#pragma unroll
        for (int n = 0; n < 3; ++n) {
          const bf16x8 fw = frag2<OS>(sWqT, 16 * n, 0, 16, lane);
          accTa[n] = AT_MFMA(fw, hi, accTa[n]);
          accTa[n] = AT_MFMA(fw, lo, accTa[n]);
        }
        // dq^T[d][t] -> LDS (hi / lo) for dWq
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int o = (16 * mt + 4 * g + r) * VS + tq;
            const __bf16 vh = (__bf16)dq[mt][r];          // (scalars again: extracting hi[4*mt+r] from the packed
            sDhi[o] = __builtin_bit_cast(bfs, vh);         //  operand vector is folded to element 0 by the compiler)
            sDlo[o] = __builtin_bit_cast(bfs, (__bf16)(dq[mt][r] - (float)vh));
          }
      }
      __syncthreads();                               // sync1: dq, softmax statistics, abar, c parts
#ifdef DMF_ATT_DEBUG
      if (h == 0 && b == 0) {
        float* dbg = a.aslab + (size_t)(4 * E * F);
        for (int i = tid; i < DH * T; i += NT) {
          const int d = i / T, t = i % T;
          dbg[i] = bf2f(__builtin_bit_cast(__bf16, sDhi[d * VS + t]));
          dbg[DH * T + i] = bf2f(__builtin_bit_cast(__bf16, sTaT[d * VS + t]));
        }
        if (wave < 6) {
          f32x4 tmp = f32x4{0.f, 0.f, 0.f, 0.f};
          for (int ks = 0; ks < 4; ++ks)
            tmp = AT_MFMA(frag<VS>(sDhi, 16 * wmt, 32 * ks, lane), frag<VS>(sTaT, 16 * wnt, 32 * ks, lane), tmp);
          for (int r = 0; r < 4; ++r) dbg[2 * DH * T + (wave * 64 + lane) * 4 + r] = tmp[r];
        }
      }
#endif
      if (tid < T) {
        float c = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) c += sCw[w * T + tid];
        sC[tid] = c;
      }
      if (wave < 6) {                                // dWq_h tile: [d = 16 wmt ..][f = 16 wnt ..] over all 128 tokens
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 ft = frag<VS>(sTaT, 16 * wnt, 32 * ks, lane);
          accWq[h] = AT_MFMA(frag<VS>(sDhi, 16 * wmt, 32 * ks, lane), ft, accWq[h]);
          accWq[h] = AT_MFMA(frag<VS>(sDlo, 16 * wmt, 32 * ks, lane), ft, accWq[h]);
        }
      }
      // S for own KEYS: s[i][r] = S[t = 16i + 4g + r][j = tq]; dS the same way from the owners' statistics
      f32x4 dk[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      {
        const bf16x8 fk = frag<KS>(sK, m0, 0, lane);
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] = AT_MFMA(frag<QS>(sQ, 16 * i, 0, lane), fk, (f32x4{0.f, 0.f, 0.f, 0.f}));
        const float aj = sA[tq];
        const bool keyok = tq < P2;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int t = 16 * i + 4 * g + r;
            const float p = expf(s[i][r] - sMx[t]) / sSm[t];
            s[i][r] = keyok ? sPw[t] * p * (aj - sAb[t]) : 0.f;
          }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          bf16x8 hi, lo;
          split_hl(s[2 * ks], s[2 * ks + 1], hi, lo);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            const bf16x8 fq = frag2<VS>(sQt, 16 * mt, 32 * ks, 32 * ks + 16, lane);
            dk[mt] = AT_MFMA(fq, hi, dk[mt]);
            dk[mt] = AT_MFMA(fq, lo, dk[mt]);
          }
        }
      }
      bf16x8 khi, klo;
      split_hl(dk[0], dk[1], khi, klo);
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        const bf16x8 fw = frag2<OS>(sWkT, 16 * n, 0, 16, lane);
        accTb[n] = AT_MFMA(fw, khi, accTb[n]);
        accTb[n] = AT_MFMA(fw, klo, accTb[n]);
      }
      __syncthreads();                               // sync2: dWq products have read sD; c is complete
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = (16 * mt + 4 * g + r) * VS + tq;
          const __bf16 vh = (__bf16)dk[mt][r];
          sDhi[o] = __builtin_bit_cast(bfs, vh);
          sDlo[o] = __builtin_bit_cast(bfs, (__bf16)(dk[mt][r] - (float)vh));
        }
      if (tid < FO) {                                // bbar[f] = sum_j c_j bf16(Tb)[j][f]
        float s2 = 0.f;
        for (int j = 0; j < T; ++j) s2 = fmaf(sC[j], bf2f(__builtin_bit_cast(__bf16, sTbT[tid * VS + j])), s2);
        sBb[tid] = s2;
      }
      {                                              // dTb += c (x) g_h
        const float cj = sC[tq];
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) accTb[n][r] = fmaf(cj, sG[16 * n + 4 * g + r], accTb[n][r]);
      }
      __syncthreads();                               // sync3: dK in LDS, bbar ready
      if (wave < 6) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 ft = frag<VS>(sTbT, 16 * wnt, 32 * ks, lane);
          accWk[h] = AT_MFMA(frag<VS>(sDhi, 16 * wmt, 32 * ks, lane), ft, accWk[h]);
          accWk[h] = AT_MFMA(frag<VS>(sDlo, 16 * wmt, 32 * ks, lane), ft, accWk[h]);
        }
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {                  // dWv_h[d][f] += u_h[d] bbar[f]
        const int e = tid + NT * i;
        if (e < DH * F) accWv[h][i] = fmaf(sU[h * DH + e / F], sBb[e % F], accWv[h][i]);
      }
    }
    // ------------------------------------------------------------------ dense maps for the conv backward
    if (tq < P2) {
#pragma unroll
      for (int n = 0; n < 3; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * n + 4 * g + r;
          if (f < F) {
            a.dYa[((size_t)b * F + f) * P2 + tq] = accTa[n][r] + wq * sDz[f];
            a.dYb[((size_t)b * F + f) * P2 + tq] = accTb[n][r] + wq * sDz[F + f];
          }
        }
    }
  }
  // ==================================================================== attention weight gradients -> slab
  float* slab = a.aslab + (size_t)blockIdx.x * (4 * E * F);
  if (wave < 6) {
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int d = 16 * wmt + 4 * g + r, f = 16 * wnt + col;
        if (f < F) {
          slab[(size_t)(h * DH + d) * F + f] = accWq[h][r];
          slab[(size_t)E * F + (size_t)(h * DH + d) * F + f] = accWk[h][r];
        }
      }
  }
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int e = tid + NT * i;
      if (e < DH * F) slab[(size_t)2 * E * F + (size_t)h * DH * F + e] = accWv[h][i];
    }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int e = tid + NT * i;
    if (e < F * E) slab[(size_t)3 * E * F + e] = accWo[i];
  }
}

using ShapeHSI = Shape<200, 1, 11, 1, 40, 10, 64>;
using ShapeTiny1 = Shape<8, 1, 5, 1, 40, 2, 64>;

template <class Sh>
static hipError_t launch_attn_train(const AttnTrainArgs& a, int grid, hipStream_t st) {
  static bool done = false;
  constexpr size_t bytes = AttnTrainLds<Sh, 96, 3>::BYTES;
  static_assert(bytes <= 160 * 1024, "attention training kernel LDS");
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_train_kernel<Sh, 96, 3>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    done = true;
  }
  hipLaunchKernelGGL((attn_train_kernel<Sh, 96, 3>), dim3(grid), dim3(512), bytes, st, a);
  return hipGetLastError();
}

hipError_t attn_train_dispatch(const dmf_shape& s, const AttnTrainArgs& a, int grid, hipStream_t st) {
  if (s.C == 200) return launch_attn_train<ShapeHSI>(a, grid, st);
  if (s.C == 8) return launch_attn_train<ShapeTiny1>(a, grid, st);
  return hipErrorInvalidValue;
}

}  // namespace dmf

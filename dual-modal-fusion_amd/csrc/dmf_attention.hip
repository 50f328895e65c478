// dmf_attention.hip — cross-modal attention + head of the `gmf.attention: 1` network (BASELINE configs[2]):
// forward AND backward in one launch for training, and the same forward alone for inference (TRAIN = false).  Sits between two launches of the fused patch kernel:
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
#include <type_traits>

#include "dmf_kargs.h"

namespace dmf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bfs;

// Diagnostic build only (-DDMF_STAMPS, tools/attn_phase_profile.py): clock stamps of wave 0 along one patch.
// Accumulation of a weight gradient in the workgroup's slab row (global memory) over the patches the workgroup walks: the
// first patch stores, later ones add with a no-return float atomic.  Every element has ONE owning lane, so the sum order is
// the program order of that lane (deterministic) — the atomic is used because it needs no load: `*p = *p + v` stalls the
// wave for a global round trip in the middle of the backward, and every barrier behind it then waits for that wave.
__device__ __forceinline__ void slab_acc(float* p, bool first, float v) {
  if (first) *p = v;
  else __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)p, v);
}

#ifdef DMF_STAMPS
__device__ unsigned long long* g_astamps = nullptr;
#define ASTAMP() do { if (threadIdx.x == 0 && g_astamps != nullptr && sidx < 64) g_astamps[(size_t)blockIdx.x * 64 + sidx] = clock64(); ++sidx; } while (0)
#else
#define ASTAMP() do { } while (0)
#endif

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
// Two transposed reads (ds_read_b64_tr_b16): per 16-lane group a block of 4 rows x 16 columns, lane 4q+p of the group
// supplies the address of row q / columns 4p..4p+3 and lane i receives column i of the four rows.  EXEC must be all ones
// (every call site is workgroup uniform); t0 is a multiple of 16 and RS of 4 (8-byte aligned addresses).
template <int RS>
__device__ __forceinline__ bf16x8 frag_t(const bfs* MT, int t0, int k0, int lane) {
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  static_assert(RS % 4 == 0, "8-byte aligned rows");
  const int i = lane & 15;
  const bfs* p = MT + (k0 + 8 * (lane >> 4) + (i >> 2)) * RS + t0 + 4 * (i & 3);
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p + 4 * RS));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  return __builtin_bit_cast(bf16x8, (s16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]});
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
// two floats -> one dword of two bf16 (a in the low half): ONE v_cvt_pk_bf16_f32 (element-wise casts of a 2-vector are
// scalarised into two conversions); the halves of such a dword as floats again
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned pk2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2)); }
__device__ __forceinline__ float lo_f(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float hi_f(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ bfs lo_h(unsigned u) { return (bfs)(u & 0xffffu); }
__device__ __forceinline__ bfs hi_h(unsigned u) { return (bfs)(u >> 16); }
#define AT_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16((A), (B), (C), 0, 0, 0)
// softmax arithmetic: exp(x) = 2^(x * log2 e) on the transcendental unit and a reciprocal instead of 32 divisions per
// row; both are within ~1e-6 relative of expf / division, far below the bf16 rounding the probabilities get next
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
// exp(s - m) as 2^(s * log2 e + nm) with nm = -m * log2 e formed once per row: one fused multiply-add per element
// (pairs of them in one v_pk_fma_f32) instead of a subtraction and a multiplication
constexpr float LOG2E = 1.44269504088896341f;
__device__ __forceinline__ float fexp_nm(float s, float nm) { return __builtin_amdgcn_exp2f(__builtin_fmaf(s, LOG2E, nm)); }
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }

}  // namespace at

template <class Sh, int E, int NH>
struct AttnTrainLds {
  static constexpr int T = 128, FP = 64, DH = 32, FO = 48;
  // row strides in halves.  ds_read_b128 serves a wave in four NON-contiguous 16-lane groups ({0-3,12-15,20-27}, ...) over
  // 64 banks: a [row][k] image read as fragments (lane = row, 16-byte chunk lane>>4) is conflict free when the row stride
  // is an EVEN number of 16-byte slots that is not a multiple of 4 (QS, KS: 6; WS: 10) — 5 or 9 slots cost 2x per read
  static constexpr int VS = 136, QS = 48, KS = 48, WS = 80, OS = 40;
  // halves
  static constexpr int oTaT = 0, oTbT = oTaT + FP * VS, oQ = oTbT + FP * VS, oK = oQ + T * QS, oQt = oK + T * KS,
                       oKt = oQt + DH * VS, oVt = oKt + DH * VS, oWq = oVt + DH * VS, oWk = oWq + DH * WS,
                       oWv = oWk + DH * WS, oWqT = oWv + DH * WS, oWkT = oWqT + FO * OS, oWo = oWkT + FO * OS,
                       oDhi = oWo + FO * OS, oDlo = oDhi + DH * VS, HALVES = oDlo + DH * VS;
  // floats (after the halves)
  static constexpr int fZ = 0, fHd = fZ + 80, fLg = fHd + 64, fDl = fLg + 64, fDh = fDl + 64, fDz = fDh + 64,
                       fPw = fDz + 80, fU = fPw + T, fOb = fU + E, fObw = fOb + E, fDzw = fObw + 8 * E,
                       fSt = fDzw + 8 * FO, fA = fSt + 4 * T, fCw = fA + T, fC = fCw + 8 * T,
                       fG = fC + T, FLOATS = fG + FO;
  static constexpr int WPREP = 3 * DH * WS + 3 * FO * OS;      // one head's weights: Wq Wk Wv [DH][WS], WqT WkT Wo [FO][OS]
  static_assert(oWo + FO * OS - oWq == WPREP && WPREP % 8 == 0 && oWq % 8 == 0, "weights are one contiguous LDS block");
  static constexpr size_t BYTES = (size_t)HALVES * 2 + (size_t)FLOATS * 4;
  static_assert(HALVES % 8 == 0, "float region stays 16-byte aligned");
  static_assert(BYTES <= 160 * 1024, "one workgroup per CU");
};

// LDS of the forward-only kernel (TRAIN = false): what inference needs and nothing else — token maps, K as rows, V transposed,
// one head's Wq / Wk / Wv / Wo, the head's vectors — 77,888 bytes, so that TWO workgroups share a CU (4 waves per SIMD at its
// ~120 registers) and each covers the other's barrier waits and LDS round trips.  Q needs no image at all: formed as
// C[d][token] (operands swapped), the accumulators ARE the B operand of S^T = K Q^T for the wave's own queries; K is formed as
// C[d][token] too and stored as rows [token][d] eight bytes at a time (its row-fragment reads pair with that operand's k order:
// frag2).  Names the training code refers to exist with offset 0 and are never touched.
template <class Sh, int E, int NH>
struct AttnFwdLds {
  static constexpr int T = 128, FP = 64, DH = 32, FO = 48;
  static constexpr int VS = 136, QS = 48, KS = 48, WS = 80, OS = 40;
  static constexpr int oTaT = 0, oTbT = oTaT + FP * VS, oK = oTbT + FP * VS, oVt = oK + T * KS, oWq = oVt + DH * VS,
                       oWk = oWq + DH * WS, oWv = oWk + DH * WS, oWo = oWv + DH * WS, HALVES = oWo + FO * OS;
  static constexpr int oQ = 0, oQt = 0, oKt = 0, oWqT = 0, oWkT = 0, oDhi = 0, oDlo = 0;                 // (training only)
  static constexpr int fZ = 0, fHd = fZ + 80, fLg = fHd + 64, fPw = fLg + 64, fDzw = fPw + T, FLOATS = fDzw + 8 * FO;
  static constexpr int fDl = 0, fDh = 0, fDz = 0, fU = 0, fOb = 0, fObw = 0, fSt = 0, fA = 0, fCw = 0, fC = 0, fG = 0;   // (training only)
  static constexpr int WPREP = AttnTrainLds<Sh, E, NH>::WPREP;
  static constexpr size_t BYTES = (size_t)HALVES * 2 + (size_t)FLOATS * 4;
  static_assert(HALVES % 8 == 0 && oWq % 8 == 0 && oWo % 8 == 0, "16-byte aligned blocks");
  static_assert(2 * BYTES <= 160 * 1024, "two workgroups per CU");
};

// TRAIN = false: forward only (logits, argmax) — the inference / evaluation kernel of the attention network.
template <class Sh, int E, int NH, bool TRAIN>
__global__ __launch_bounds__(512) void attn_train_kernel(const AttnTrainArgs a) {
  using namespace at;
  using L = typename std::conditional<TRAIN, AttnTrainLds<Sh, E, NH>, AttnFwdLds<Sh, E, NH>>::type;
  constexpr int T = L::T, FP = L::FP, DH = L::DH, FO = L::FO, NT = 512;
  constexpr int VS = L::VS, QS = L::QS, KS = L::KS, WS = L::WS, OS = L::OS;
  constexpr int F = Sh::F, F2 = Sh::F2, H = Sh::H, P2 = Sh::P2;
  constexpr int RSD = (Sh::P + 3) & ~3;               // row stride of the dense gradient maps (16-byte aligned rows)
  static_assert(E == NH * DH && F == 40 && F2 == 80 && H == 64 && P2 <= T, "attention geometry");
  static_assert(H % 4 == 0 && F % 4 == 0 && 4 * F2 <= NT - 64 && 4 * E <= NT, "thread roles of the head backward");
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  // every LDS array is a constant offset from the dynamic-LDS symbol: no pointer variables that could be spilled as
  // generic 64-bit pointers (their accesses would become flat_* instead of ds_*)
#define AT_H(off) (reinterpret_cast<bfs*>(smraw) + (off))
#define AT_F(off) (reinterpret_cast<float*>(smraw + (size_t)L::HALVES * 2) + (off))
#define sTaT AT_H(L::oTaT)
#define sTbT AT_H(L::oTbT)
#define sQ AT_H(L::oQ)
#define sK AT_H(L::oK)
#define sQt AT_H(L::oQt)
#define sKt AT_H(L::oKt)
#define sVt AT_H(L::oVt)
#define sWq AT_H(L::oWq)
#define sWk AT_H(L::oWk)
#define sWv AT_H(L::oWv)
#define sWqT AT_H(L::oWqT)
#define sWkT AT_H(L::oWkT)
#define sWo AT_H(L::oWo)
#define sDhi AT_H(L::oDhi)
#define sDlo AT_H(L::oDlo)
#define sZ AT_F(L::fZ)
#define sHd AT_F(L::fHd)
#define sLg AT_F(L::fLg)
#define sDl AT_F(L::fDl)
#define sDh AT_F(L::fDh)
#define sDz AT_F(L::fDz)
#define sPw AT_F(L::fPw)
#define sU AT_F(L::fU)
#define sOb AT_F(L::fOb)
#define sObw AT_F(L::fObw)
#define sDzw AT_F(L::fDzw)
#define sSt AT_F(L::fSt)   /* per query t: {-row max * log2 e, 1 / row sum, abar_t, w_t} */
#define sA AT_F(L::fA)
#define sCw AT_F(L::fCw)
#define sC AT_F(L::fC)
#define sG AT_F(L::fG)

  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int m0 = wave * 16;                          // own queries (and own keys for dK)
  const float* __restrict__ th = a.theta;
  const int K = a.K;
  const float scale = 0.17677669529663687f;          // float32(1/sqrt(32))
  const int boff = a.cursor != nullptr ? a.cursor[0] * a.B : 0;

  for (int i = tid0; i < T; i += NT) sPw[i] = i < P2 ? a.pool[i] : 0.f;

  // attention weight gradients accumulate in this workgroup's slab in global memory (L2 resident); every element has
  // exactly one owning lane, which adds to it in program order: deterministic, no atomics
  float* __restrict__ slab = a.aslab + (size_t)blockIdx.x * (4 * E * F);
  const int wmt = wave / 3, wnt = wave % 3;          // dWq / dWk tile of this wave (waves 0..5): d tile, f tile
  // head weights this thread needs (forward and backward) are patch invariant: fetched once per workgroup
  float hw1[10], hw2[8], hwd[8], hwz[16], hwu[10], hb1, hb2;
  {
    const int j = tid0 >> 3, pp = tid0 & 7;                        // fc1: row j, columns pp + 8 m;  fc2: row k = j
#pragma unroll
    for (int m = 0; m < 10; ++m) hw1[m] = th[a.oFc1w + (int64_t)j * F2 + pp + 8 * m];
    hb1 = th[a.oFc1b + j];
#pragma unroll
    for (int m = 0; m < 8; ++m) hw2[m] = j < K ? th[a.oFc2w + (int64_t)j * H + pp + 8 * m] : 0.f;
    hb2 = j < K ? th[a.oFc2b + j] : 0.f;
    // the backward products keep the parts of one output in ADJACENT lanes (summed by DPP, no LDS round trip + barrier)
    const int jd = tid0 >> 3, pd = tid0 & 7;                       // dh: column jd, rows k = pd + 8 i
#pragma unroll
    for (int i = 0; i < 8; ++i) hwd[i] = (TRAIN && pd + 8 * i < K) ? th[a.oFc2w + (int64_t)(pd + 8 * i) * H + jd] : 0.f;
    const int iz = tid0 >> 2, pz = tid0 & 3;                       // dz: column iz (< F2), rows j = pz + 4 m
#pragma unroll
    for (int m = 0; m < 16; ++m) hwz[m] = (TRAIN && iz < F2) ? th[a.oFc1w + (int64_t)(pz + 4 * m) * F2 + iz] : 0.f;
    const int eu = tid0 >> 2, pu = tid0 & 3;                       // u: column eu (< E), rows f = pu + 4 m
#pragma unroll
    for (int m = 0; m < 10; ++m) hwu[m] = (TRAIN && eu < E) ? bf2f((__bf16)th[a.oWo + (int64_t)(pu + 4 * m) * E + eu]) : 0.f;
  }
  // first patch's tokens: 4 pieces of 16 bytes per thread, kept in registers and refilled one patch ahead
  uint4 tokr[4];
  auto fetch_tokens = [&](int bb, int tid) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = tid + NT * q;
      const int mp = i >> 10, rem = i & 1023, pc = rem >> 7, t = rem & 127;
      tokr[q] = *reinterpret_cast<const uint4*>((mp ? a.tokB : a.tokA) + ((size_t)bb * T + t) * FP + pc * 8);
    }
  };
  if ((int)blockIdx.x < a.B) fetch_tokens(blockIdx.x, tid0);

  bool first = true;
  for (int b = blockIdx.x; b < a.B; b += gridDim.x, first = false) {
    // per-thread roles are derived from an opaque copy of the thread id inside the loop: otherwise every index
    // computation below is loop invariant, gets hoisted in front of the loop and spilled (hundreds of registers)
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, g = lane >> 4, col = lane & 15;
    const int tq = m0 + col;                         // the token this lane's C columns belong to
    int sidx = 0; (void)sidx;
    ASTAMP();                                        // 0
    // stage one head's weights: W[d][f] for the projections; for the backward also W^T[f][d]
    auto stage_weights = [&](int h, bool bwd, int tid) {
      (void)bwd;
      const uint4* src = reinterpret_cast<const uint4*>(a.wprep + (size_t)h * L::WPREP);
      uint4* dst = reinterpret_cast<uint4*>(sWq);
      if constexpr (TRAIN) {
        for (int i = tid; i < L::WPREP / 8; i += NT) dst[i] = src[i];
      } else {                                         // [Wq | Wk | Wv] and, behind WqT / WkT in the global block, Wo
        constexpr int N1 = 3 * DH * WS / 8, N2 = FO * OS / 8, SK = (3 * DH * WS + 2 * FO * OS) / 8;
        for (int i = tid; i < N1 + N2; i += NT) dst[i] = src[i < N1 ? i : SK + (i - N1)];
      }
    };
    // projections of the wave's 16 tokens; writes Qs (both layouts), K (both layouts), V^T
    bf16x8 fq_own;                                     // forward-only kernel: bf16(Q scale)^T of the own queries as an MFMA operand
    auto project = [&](int lane, int g, int col, int hb) {         // hb >= 0: backward of head hb, also forms a_j
      if constexpr (!TRAIN) {
        // Q and K as C[d = 16n + 4g + r][token m0 + col] (weights as the A operand), V as C[token][d]: every LDS store of
        // the forward is an 8-byte store and Q never leaves the registers
        f32x4 qT[2], kT[2], v[2];
  #pragma unroll
        for (int n = 0; n < 2; ++n) qT[n] = kT[n] = v[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  #pragma unroll
        for (int ks = 0; ks < FP / 32; ++ks) {
          const bf16x8 fa = frag_t<VS>(sTaT, m0, 32 * ks, lane);
          const bf16x8 fb = frag_t<VS>(sTbT, m0, 32 * ks, lane);
  #pragma unroll
          for (int n = 0; n < 2; ++n) {
            qT[n] = AT_MFMA(frag<WS>(sWq, 16 * n, 32 * ks, lane), fa, qT[n]);
            kT[n] = AT_MFMA(frag<WS>(sWk, 16 * n, 32 * ks, lane), fb, kT[n]);
            v[n] = AT_MFMA(fb, frag<WS>(sWv, 16 * n, 32 * ks, lane), v[n]);
          }
        }
  #pragma unroll
        for (int n = 0; n < 2; ++n) {
          *reinterpret_cast<uint2*>(sK + (m0 + col) * KS + 16 * n + 4 * g) = make_uint2(pk2(kT[n][0], kT[n][1]), pk2(kT[n][2], kT[n][3]));
          *reinterpret_cast<uint2*>(sVt + (16 * n + col) * VS + m0 + 4 * g) = make_uint2(pk2(v[n][0], v[n][1]), pk2(v[n][2], v[n][3]));
        }
        const f32x4 q0 = qT[0] * scale, q1 = qT[1] * scale;
        fq_own = pack_bf(q0, q1);                      // k = d: {4g + r} and {16 + 4g + r}, the order frag2 reads K's rows in
        (void)hb;
        return;
      }
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
      float ap[4] = {0.f, 0.f, 0.f, 0.f};
  #pragma unroll
      for (int n = 0; n < 2; ++n) {
        const unsigned q01 = pk2(q[n][0] * scale, q[n][1] * scale), q23 = pk2(q[n][2] * scale, q[n][3] * scale);
        const unsigned k01 = pk2(k[n][0], k[n][1]), k23 = pk2(k[n][2], k[n][3]);
        bfs* pq = sQ + (m0 + rb) * QS + 16 * n + col;
        bfs* pk = sK + (m0 + rb) * KS + 16 * n + col;
        pq[0 * QS] = lo_h(q01); pq[1 * QS] = hi_h(q01); pq[2 * QS] = lo_h(q23); pq[3 * QS] = hi_h(q23);
        pk[0 * KS] = lo_h(k01); pk[1 * KS] = hi_h(k01); pk[2 * KS] = lo_h(k23); pk[3 * KS] = hi_h(k23);
        const int o = (16 * n + col) * VS + m0 + rb;     // 4 consecutive tokens of one feature: one 8-byte store
        const unsigned v01 = pk2(v[n][0], v[n][1]), v23 = pk2(v[n][2], v[n][3]);
        *reinterpret_cast<uint2*>(sQt + o) = make_uint2(q01, q23);
        *reinterpret_cast<uint2*>(sKt + o) = make_uint2(k01, k23);
        *reinterpret_cast<uint2*>(sVt + o) = make_uint2(v01, v23);
        if (hb >= 0) {                                   // a_j = sum_d bf16(V)[j][d] u_h[d]: this lane's features 16n + col
          const float ud = sU[hb * DH + 16 * n + col];
          ap[0] = fmaf(lo_f(v01), ud, ap[0]); ap[1] = fmaf(hi_f(v01), ud, ap[1]);
          ap[2] = fmaf(lo_f(v23), ud, ap[2]); ap[3] = fmaf(hi_f(v23), ud, ap[3]);
        }
      }
      if (hb >= 0)
  #pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float aj = row16_sum(ap[r]);             // over the 16 lanes (features) of the token m0 + 4g + r
          if (col == 0) sA[m0 + rb + r] = aj;
        }
    };
    // P^T for the wave's queries: s[i][r] = P[t = tq][j = 16i + 4g + r]  (fp32, keys >= P2 masked)
    auto softmax_T = [&](f32x4 (&s)[8], int lane, int g, float& mx_out, float& inv_out) {
      if constexpr (TRAIN) {
        const bf16x8 fq = frag<QS>(sQ, m0, 0, lane);
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] = AT_MFMA(frag<KS>(sK, 16 * i, 0, lane), fq, (f32x4{0.f, 0.f, 0.f, 0.f}));
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] = AT_MFMA(frag2<KS>(sK, 16 * i, 0, 16, lane), fq_own, (f32x4{0.f, 0.f, 0.f, 0.f}));
      }
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
      const float nm = -mx * LOG2E;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[i][r] = fexp_nm(s[i][r], nm); sum += s[i][r]; }
      sum = xrow_sum(sum);
      const float inv = frcp(sum);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[i][r] *= inv;
      mx_out = nm; inv_out = inv;                     // (the row statistic kept is nm = -max * log2 e)
    };
    // ------------------------------------------------------------------ tokens -> LDS, transposed [f][t]
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = tid + NT * q;
      const int mp = i >> 10, rem = i & 1023, pc = rem >> 7, t = rem & 127;
      const uint4 v = tokr[q];
      bfs* dst = (mp ? sTbT : sTaT) + (pc * 8) * VS + t;
      dst[0 * VS] = (bfs)(v.x & 0xffff); dst[1 * VS] = (bfs)(v.x >> 16);
      dst[2 * VS] = (bfs)(v.y & 0xffff); dst[3 * VS] = (bfs)(v.y >> 16);
      dst[4 * VS] = (bfs)(v.z & 0xffff); dst[5 * VS] = (bfs)(v.z >> 16);
      dst[6 * VS] = (bfs)(v.w & 0xffff); dst[7 * VS] = (bfs)(v.w >> 16);
    }
    if (b + (int)gridDim.x < a.B) fetch_tokens(b + gridDim.x, tid);      // lands while this patch is processed
    if (tid < F2) sZ[tid] = a.zin[(size_t)b * F2 + tid];

    // ================================================================== pass 1: forward
    f32x4 accO[3];                                   // (O Wo^T)^T[f = 16n + 4g + r][t = tq], summed over heads
#pragma unroll
    for (int n = 0; n < 3; ++n) accO[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float wq = sPw[tq];                        // (sPw was written before the first barrier of this iteration)
#pragma unroll 1
    for (int h = 0; h < NH; ++h) {
      int tidh = tid0;                               // (opaque again: keeps this head's index math inside the loop)
      asm volatile("" : "+v"(tidh));
      const int lane = tidh & 63, g = lane >> 4, col = lane & 15;
      __syncthreads();
      ASTAMP();                                      // p1: 1 + 4h
      stage_weights(h, false, tidh);
      __syncthreads();
      ASTAMP();
      project(lane, g, col, -1);
      __syncthreads();
      ASTAMP();
      f32x4 s[8];
      float mxq, invq;
      softmax_T(s, lane, g, mxq, invq);
      ASTAMP();
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
      if constexpr (TRAIN) {
        const u32x4 ow = __builtin_bit_cast(u32x4, ob);          // ob[4 mt + r] = bf16(o[mt][r]): word 2 mt + r / 2
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const unsigned w2 = ow[2 * mt + (r >> 1)];
            const float v = row16_sum(wq * ((r & 1) ? hi_f(w2) : lo_f(w2)));
            if (col == 0) sObw[wave * E + h * DH + 16 * mt + 4 * g + r] = v;
          }
      }
    }
    ASTAMP();                                        // 13
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
    } else if (TRAIN && tid >= 64 && tid < 64 + E) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += sObw[w * E + tid - 64];
      sOb[tid - 64] = s;
    }
    __syncthreads();
    // ------------------------------------------------------------------ head forward (weights in registers)
    {
      const int j = tid >> 3, pp = tid & 7;
      float acc = 0.f;
#pragma unroll
      for (int m = 0; m < 10; ++m) acc = fmaf(hw1[m], sZ[pp + 8 * m], acc);
      acc = AT_DPP(acc, 0xB1, AT_ADD); acc = AT_DPP(acc, 0x4E, AT_ADD); acc = AT_DPP(acc, 0x141, AT_ADD);
      if (pp == 0) sHd[j] = fmaxf(acc + hb1, 0.f);
    }
    __syncthreads();
    {
      const int k = tid >> 3, pp = tid & 7;
      float acc = 0.f;
#pragma unroll
      for (int m = 0; m < 8; ++m) acc = fmaf(hw2[m], sHd[pp + 8 * m], acc);
      acc = AT_DPP(acc, 0xB1, AT_ADD); acc = AT_DPP(acc, 0x4E, AT_ADD); acc = AT_DPP(acc, 0x141, AT_ADD);
      if (k < K && pp == 0) sLg[k] = acc + hb2;
    }
    __syncthreads();
    if constexpr (!TRAIN) {
      if (wave == 0) {
        const float v = lane < K ? sLg[lane] : -INFINITY;
        float mx = v;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if (lane < K) a.logits[(size_t)b * K + lane] = v;
        if (a.pred != nullptr) {
          const unsigned long long bal = __ballot(v == mx);
          if (lane == 0) a.pred[b] = __ffsll((long long)bal) - 1;       // first maximal index, as torch.max
        }
      }
      continue;                                      // (the next patch starts with a barrier)
    }
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
    // head backward.  Every matrix-vector product is split over the workgroup with its weights in registers; the parts
    // of one output sit in adjacent lanes and are summed by DPP in a fixed order: one barrier per product.
    {   // dh[j] = relu'(h[j]) sum_k W2[k][j] dl[k]                      thread <-> (j, part of 8): k = part + 8 i
      const int j = tid >> 3, part = tid & 7;
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { const int k = part + 8 * i; acc = fmaf(hwd[i], k < K ? sDl[k] : 0.f, acc); }
      acc = AT_DPP(acc, 0xB1, AT_ADD); acc = AT_DPP(acc, 0x4E, AT_ADD); acc = AT_DPP(acc, 0x141, AT_ADD);
      if (part == 0) sDh[j] = sHd[j] > 0.f ? acc : 0.f;
    }
    __syncthreads();
    {   // dz[i] = sum_j W1[j][i] dh[j]                                   thread <-> (i, part of 4): j = part + 4 m
      const int i = tid >> 2, part = tid & 3;
      float acc = 0.f;
#pragma unroll
      for (int m = 0; m < 16; ++m) acc = fmaf(hwz[m], sDh[part + 4 * m], acc);
      acc = AT_DPP(acc, 0xB1, AT_ADD); acc = AT_DPP(acc, 0x4E, AT_ADD);
      if (part == 0 && i < F2) {
        sDz[i] = acc;
        a.ws_z[hv_index(b, i, a.B)] = sZ[i];
      }
      if (tid >= NT - 64) {                          // (the last wave has no dz column)
        const int j = tid - (NT - 64);
        a.ws_h[hv_index(b, j, a.B)] = sHd[j];
        a.ws_dh[hv_index(b, j, a.B)] = sDh[j];
        a.ws_dl[hv_index(b, j, a.B)] = sDl[j];
      }
    }
    __syncthreads();
    {   // u[e] = sum_f dza[f] bf16(Wo[f][e])                             thread <-> (e, part of 4): f = part + 4 m
      const int e = tid >> 2, part = tid & 3;
      float acc = 0.f;
#pragma unroll
      for (int m = 0; m < 10; ++m) acc = fmaf(sDz[part + 4 * m], hwu[m], acc);
      acc = AT_DPP(acc, 0xB1, AT_ADD); acc = AT_DPP(acc, 0x4E, AT_ADD);
      if (part == 0 && e < E) sU[e] = acc;           // (read after the first barrier of pass 2)
    }
    // dWo[f][e] += dza[f] * obar[e]
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + NT * i;
      if (e < F * E) {
        const float v = sDz[e / E] * sOb[e % E];
        slab_acc(slab + (size_t)3 * E * F + e, first, v);
      }
    }

    ASTAMP();                                        // 14: head forward + backward done
    // ================================================================== pass 2: backward, head by head
    f32x4 accTa[3], accTb[3];                        // dTa^T / dTb^T [f = 16n + 4g + r][token tq]
#pragma unroll
    for (int n = 0; n < 3; ++n) accTa[n] = accTb[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int h = 0; h < NH; ++h) {
      int tid = tid0;
      asm volatile("" : "+v"(tid));
      const int lane = tid & 63, g = lane >> 4, col = lane & 15, tq = m0 + col;
      __syncthreads();
      ASTAMP();                                      // p2: 15 + 8h
      stage_weights(h, true, tid);
      __syncthreads();
      ASTAMP();
      project(lane, g, col, h);                      // (+ a_j = sum_d bf16(V)[j][d] u_h[d] of the own tokens)
      __syncthreads();
      ASTAMP();
      // g_h[f] = sum_d u_h[d] bf16(Wv_h)[d][f]      (read after sync2)
      if (tid >= NT - FO) {
        const int f = tid - (NT - FO);
        float s = 0.f;
        for (int d = 0; d < DH; ++d) s = fmaf(sU[h * DH + d], bf2f(__builtin_bit_cast(__bf16, sWv[d * WS + f])), s);
        sG[f] = s;
      }
      f32x4 s[8];
      float mxq, invq;
      softmax_T(s, lane, g, mxq, invq);              // P[tq][j], j = 16i + 4g + r
      ASTAMP();
      {
        // abar_t = sum_j P[t][j] a_j      (c = bf16(P)^T w is formed with the own-KEY orientation below: no lane reductions)
        float ab = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) ab = fmaf(s[i][r], sA[16 * i + 4 * g + r], ab);
        ab = xrow_sum(ab);
        if (g == 0) *reinterpret_cast<float4*>(sSt + 4 * tq) = make_float4(mxq, invq, ab, wq);
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
        // dq^T[d][t] -> LDS (hi / lo) for dWq: the halves of the operand words just formed (element 4 mt + r; read as
        // 32-bit words — extracting a __bf16 element of the packed vector was folded to element 0 by the compiler)
        const u32x4 hw = __builtin_bit_cast(u32x4, hi), lw = __builtin_bit_cast(u32x4, lo);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int o = (16 * mt + 4 * g + r) * VS + tq;
            const unsigned wh = hw[2 * mt + (r >> 1)], wl = lw[2 * mt + (r >> 1)];
            sDhi[o] = (r & 1) ? hi_h(wh) : lo_h(wh);
            sDlo[o] = (r & 1) ? hi_h(wl) : lo_h(wl);
          }
      }
      ASTAMP();
      __syncthreads();                               // sync1: dq, softmax statistics, abar, c parts
      ASTAMP();
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
      if (wave < 6) {                                // dWq_h tile: [d = 16 wmt ..][f = 16 wnt ..] over all 128 tokens
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 ft = frag<VS>(sTaT, 16 * wnt, 32 * ks, lane);
          acc = AT_MFMA(frag<VS>(sDhi, 16 * wmt, 32 * ks, lane), ft, acc);
          acc = AT_MFMA(frag<VS>(sDlo, 16 * wmt, 32 * ks, lane), ft, acc);
        }
        if (16 * wnt + col < F) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* p = slab + (size_t)(h * DH + 16 * wmt + 4 * g + r) * F + 16 * wnt + col;
            slab_acc(p, first, acc[r]);
          }
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
        float cpart = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            const float4 st0 = *reinterpret_cast<const float4*>(sSt + 4 * (16 * i + 4 * g + r));
            const float4 st1 = *reinterpret_cast<const float4*>(sSt + 4 * (16 * i + 4 * g + r + 1));
            const float p0 = fexp_nm(s[i][r], st0.x) * st0.y;                        // P[t][j], as softmax_T forms it
            const float p1 = fexp_nm(s[i][r + 1], st1.x) * st1.y;
            const unsigned pb = pk2(p0, p1);
            cpart = fmaf(lo_f(pb), st0.w, cpart);                                    // c_j += bf16(P)[t][j] w_t
            cpart = fmaf(hi_f(pb), st1.w, cpart);
            s[i][r] = keyok ? st0.w * p0 * (aj - st0.z) : 0.f;                       // w_t P[t][j] (a_j - abar_t)
            s[i][r + 1] = keyok ? st1.w * p1 * (aj - st1.z) : 0.f;
          }
        cpart = xrow_sum(cpart);                      // all 128 queries of this key: 32 in the lane, x 4 lane groups
        if (g == 0) sC[tq] = keyok ? cpart : 0.f;
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
      ASTAMP();
      __syncthreads();                               // sync2: dWq products have read sD; c is complete
      {
        const u32x4 hw = __builtin_bit_cast(u32x4, khi), lw = __builtin_bit_cast(u32x4, klo);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int o = (16 * mt + 4 * g + r) * VS + tq;
            const unsigned wh = hw[2 * mt + (r >> 1)], wl = lw[2 * mt + (r >> 1)];
            sDhi[o] = (r & 1) ? hi_h(wh) : lo_h(wh);
            sDlo[o] = (r & 1) ? hi_h(wl) : lo_h(wl);
          }
      }
      {                                              // bbar[f] = sum_j c_j bf16(Tb)[j][f]: thread <-> (f, 16 keys)
        const int f = tid & 63, part = tid >> 6;
        float s2 = 0.f;
        if (f < FO) {
          const bf16x8 t0 = *reinterpret_cast<const bf16x8*>(sTbT + f * VS + 16 * part);
          const bf16x8 t1 = *reinterpret_cast<const bf16x8*>(sTbT + f * VS + 16 * part + 8);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            s2 = fmaf(sC[16 * part + i], bf2f(t0[i]), s2);
            s2 = fmaf(sC[16 * part + 8 + i], bf2f(t1[i]), s2);
          }
        }
        sCw[part * T + f] = s2;                      // (the c partials in sCw were consumed before sync2)
      }
      {                                              // dTb += c (x) g_h
        const float cj = sC[tq];
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) accTb[n][r] = fmaf(cj, sG[16 * n + 4 * g + r], accTb[n][r]);
      }
      ASTAMP();
      __syncthreads();                               // sync3: dK in LDS, bbar partials ready
      if (wave < 6) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 ft = frag<VS>(sTbT, 16 * wnt, 32 * ks, lane);
          acc = AT_MFMA(frag<VS>(sDhi, 16 * wmt, 32 * ks, lane), ft, acc);
          acc = AT_MFMA(frag<VS>(sDlo, 16 * wmt, 32 * ks, lane), ft, acc);
        }
        if (16 * wnt + col < F) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* p = slab + (size_t)E * F + (size_t)(h * DH + 16 * wmt + 4 * g + r) * F + 16 * wnt + col;
            slab_acc(p, first, acc[r]);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {                  // dWv_h[d][f] += u_h[d] bbar[f]
        const int e = tid + NT * i;
        if (e < DH * F) {
          float* p = slab + (size_t)2 * E * F + (size_t)h * DH * F + e;
          const int f = e % F;
          float bb = 0.f;
#pragma unroll
          for (int q = 0; q < 8; ++q) bb += sCw[q * T + f];
          const float v = sU[h * DH + e / F] * bb;
          slab_acc(p, first, v);
        }
      }
    }
    ASTAMP();                                        // 39
    // ------------------------------------------------------------------ dense maps for the conv backward
    if (tq < P2) {
#pragma unroll
      for (int n = 0; n < 3; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * n + 4 * g + r;
          if (f < F) {
            const size_t o = ((size_t)b * F + f) * (Sh::P * RSD) + (tq / Sh::P) * RSD + (tq % Sh::P);   // [B][F][P][RSD]
            a.dYa[o] = accTa[n][r] + wq * sDz[f];
            a.dYb[o] = accTb[n][r] + wq * sDz[F + f];
          }
        }
    }
  }
}

#undef AT_H
#undef AT_F

// bf16 copies of the attention weights in the LDS layout of attn_train_kernel, one block per head:
//   [Wq | Wk | Wv : [DH][WS] rows d, k = f]  [WqT | WkT : [FO][OS] rows f, k = d]  [Wo_h : [FO][OS] rows f, k = d]
template <int E, int NH>
__global__ __launch_bounds__(256) void attn_prep_kernel(const float* __restrict__ th, int64_t oWq, int64_t oWk, int64_t oWv,
                                                        int64_t oWo, bfs* __restrict__ out) {
  using L = AttnTrainLds<Shape<8, 1, 5, 1, 40, 2, 64>, E, NH>;      // the layout does not depend on the patch shape
  constexpr int F = 40, DH = L::DH, WS = L::WS, FO = L::FO, OS = L::OS;
  const int h = blockIdx.y;
  bfs* o = out + (size_t)h * L::WPREP;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < L::WPREP; i += gridDim.x * 256) {
    float v = 0.f;
    if (i < 3 * DH * WS) {
      const int w = i / (DH * WS), rem = i - w * (DH * WS), d = rem / WS, f = rem - d * WS;
      if (f < F) v = th[(w == 0 ? oWq : (w == 1 ? oWk : oWv)) + (int64_t)(h * DH + d) * F + f];
    } else {
      const int j = i - 3 * DH * WS, w = j / (FO * OS), rem = j - w * (FO * OS), f = rem / OS, d = rem - f * OS;
      if (f < F && d < DH)
        v = w == 2 ? th[oWo + (int64_t)f * E + h * DH + d] : th[(w == 0 ? oWq : oWk) + (int64_t)(h * DH + d) * F + f];
    }
    o[i] = at::f2bf(v);
  }
}

using ShapeHSI = Shape<200, 1, 11, 1, 40, 10, 64>;
using ShapeTiny1 = Shape<8, 1, 5, 1, 40, 2, 64>;

#ifdef DMF_STAMPS
hipError_t set_attn_stamps(unsigned long long* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_astamps), &p, sizeof(p)); }
#endif

size_t attn_prep_bytes() { return (size_t)3 * AttnTrainLds<ShapeTiny1, 96, 3>::WPREP * 2; }

hipError_t attn_prep_launch(const float* theta, int64_t oWq, int64_t oWk, int64_t oWv, int64_t oWo, void* out, hipStream_t st) {
  hipLaunchKernelGGL((attn_prep_kernel<96, 3>), dim3(8, 3), dim3(256), 0, st, theta, oWq, oWk, oWv, oWo, static_cast<bfs*>(out));
  return hipGetLastError();
}

template <class Sh, bool TRAIN>
static hipError_t launch_attn_train(const AttnTrainArgs& a, int grid, hipStream_t st) {
  static LdsAttrOnce once;
  constexpr size_t bytes = std::conditional<TRAIN, AttnTrainLds<Sh, 96, 3>, AttnFwdLds<Sh, 96, 3>>::type::BYTES;
  static_assert(bytes <= 160 * 1024, "attention kernel LDS");
  hipError_t e = once.set(reinterpret_cast<const void*>(&attn_train_kernel<Sh, 96, 3, TRAIN>), (int)bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((attn_train_kernel<Sh, 96, 3, TRAIN>), dim3(grid), dim3(512), bytes, st, a);
  return hipGetLastError();
}

hipError_t attn_train_dispatch(const dmf_shape& s, const AttnTrainArgs& a, int grid, hipStream_t st) {
  if (s.C == 200) return launch_attn_train<ShapeHSI, true>(a, grid, st);
  if (s.C == 8) return launch_attn_train<ShapeTiny1, true>(a, grid, st);
  return hipErrorInvalidValue;
}

hipError_t attn_forward_dispatch(const dmf_shape& s, const AttnTrainArgs& a, int grid, hipStream_t st) {
  if (s.C == 200) return launch_attn_train<ShapeHSI, false>(a, grid, st);
  if (s.C == 8) return launch_attn_train<ShapeTiny1, false>(a, grid, st);
  return hipErrorInvalidValue;
}

int attn_shape_supported(const dmf_shape& s) {
  if (s.E != 96 || s.heads != 3 || s.F != 40 || s.H != 64) return 0;
  return (s.C == 200 && s.C2 == 1 && s.P == 11 && s.S == 1 && s.G == 10) || (s.C == 8 && s.C2 == 1 && s.P == 5 && s.S == 1 && s.G == 2);
}

}  // namespace dmf

// dmf_patch_kernel.hip — the hot path: ONE launch does, per patch, the dual-branch forward, the head,
// the softmax cross-entropy and the full backward (weight-gradient slabs), entirely on chip.
//
// Replaces (reference): `output = self.cur_model(data1, data2)`, `loss = self.loss(output, target.long())`,
// `loss.backward()`  — solver/mainsolver.py:52-54 — and, in MODE_FWD, the eval forward + argmax
// (mainsolver.py:109,139,169-170).  The arithmetic is the GMFNet stated in oracle/gmfnet_ref.py.
//
// Design (gfx950): one workgroup (10 wave64 for F = 40) owns one patch at a time.  The patch window
// (P*P pixels x C bands, pixel-major) is read from HBM/L2 ONCE with 16-byte coalesced loads issued back to
// back, and stays in LDS until the last weight gradient has consumed it; the two feature maps live in LDS
// channel-major with 16-byte-aligned rows, so every spatial stage loads whole rows with ds_read_b128 and works
// from registers.  The only global writes are logits/loss, four small per-patch head vectors and ONE gradient slab
// row per workgroup.  All reductions are fixed-order (DPP / ordered LDS sums, no float atomics): a step is bitwise
// reproducible.
//
//   P0  gather  X[pix][band]  (a scene row of the window is P*C contiguous floats), aux tile
//   P1  spec_a  grouped 1x1 and lift_b SxS conv as wave-tasks (channel block, 64-pixel half): lane <-> pixel,
//       weights wave-uniform in SGPRs (scalar loads through a constant-address-space view of theta)
//   P2  spat_a / spat_b depthwise 3x3: thread <-> (channel, row): three rows in registers, 99 FMAs, ReLU mask word,
//       anchor-Gaussian pooling reduced over the 16 row lanes by DPP
//   P3  head: fc1 / fc2 from register-resident weight rows, softmax-CE by wavefront shuffles, dlogits;
//       dh and dz re-use the same registers (transposed reduction by DPP / permlane swaps + ordered LDS sums)
//   P4  depthwise backward from the row masks (dY2 = mask * dz[f] * pool[pix]): dW/db (DPP row sums -> slab), then
//       dY1 rows in place, with the spec_a / lift_b bias and lift_b weight gradients
//   P5  spec_a weight gradient from the still-resident X tile
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "dmf_kargs.h"
#include "dmf_lanes.h"

namespace dmf {

// Diagnostic build only (-DDMF_STAMPS, tools/phase_profile.py): per-phase s_memtime stamps of wave 0, written to a
// buffer nothing else reads.  The shipped library contains no stamp.
#ifdef DMF_STAMPS
__device__ unsigned long long* g_stamps = nullptr;
#define STAMP(i) do { if (threadIdx.x == 0 && g_stamps != nullptr) g_stamps[(size_t)blockIdx.x * 16 + (i)] = clock64(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

template <class Sh>
struct Lds {
  static constexpr int cmax(int a, int b) { return a > b ? a : b; }
  // 16-byte band chunks a wavefront reads per spectral group (misaligned groups: lead + Cg <= Cg + 3 floats)
  static constexpr int QG = Sh::MISAL ? (Sh::Cg + 3 + 3) / 4 : Sh::Cg / 4;
  // misaligned groups: the part of a group's band columns that no neighbour's aligned chunks can touch (smallest
  // over the four leads); the spec_a gradient partials are packed into those private columns, row after row
  static constexpr int PW = Sh::MISAL ? ((Sh::Cg - 3) / 4) * 4 : Sh::Cg;
  static constexpr int NSLW = (64 / QG) < 16 ? (64 / QG) : 16;     // pixel slices of a wave's spec_a gradient
  // partial sums of that gradient: a wave whose group is its own (M == 4) recycles its private band slice of the X
  // tile; groups shared by several waves (M > 4) get a dedicated region
  static constexpr bool OWN_SLICE = (Sh::M == 4);
  static constexpr int GSCR = OWN_SLICE ? 0 : Sh::NB * NSLW * 4 * Sh::Cg;
  static constexpr int RS = (Sh::P + 3) & ~3;                      // row stride of the channel-major maps (floats)
  static constexpr int FSZ = Sh::P * RS;                           // one channel image
  static constexpr int AUXP = (Sh::PB * Sh::C2 + 3) & ~3;
  static constexpr int NWH = (Sh::H * 8 + 63) / 64;                // waves taking part in the head
  static constexpr int MS = Sh::P;                                 // row-mask stride per channel
  static constexpr int DWW = 2 * Sh::F * 12;                       // staged depthwise taps + bias: [branch][F][12]
  static constexpr int REST = 2 * Sh::F * FSZ + AUXP + Sh::P * RS + 2 * Sh::F * MS + Sh::F2 + 2 * Sh::H + 2 * KMAX +
                              NWH * cmax(Sh::F2, Sh::H) + GSCR + DWW;
  // X-tile row stride (floats).  (Cs/4) odd makes a 16-lane ds_read_b128 group with lane<->pixel hit 64 distinct
  // banks (MI355X_MICROARCH.md §LDS); the pad is dropped (2-way conflict) only where it would not fit in 160 KiB.
  static constexpr int CsPad = Sh::C + ((((Sh::C / 4) & 1) == 0) ? 4 : 0);
  static constexpr int Cs = (Sh::P2 * CsPad + REST <= 40960) ? CsPad : Sh::C;
  static constexpr int TOKSCR = (Sh::P2 <= 128) ? 2 * 128 * 72 / 2 : 0;      // MODE_TOKENS staging (floats)
  static constexpr int SCR = cmax(Sh::P2 * Cs, TOKSCR);                      // X tile (later: token staging)
  static_assert(!OWN_SLICE || NSLW * 4 <= Sh::P2, "own-slice scratch needs NSLW*4 pixel rows");
  static_assert(!Sh::MISAL || (OWN_SLICE && (NSLW * 4 * 4 * QG + PW - 1) / PW <= Sh::P2), "packed scratch fits the private columns");
  // offsets in floats
  static constexpr int oX = 0;
  static constexpr int oY1a = oX + SCR;                            // [F][P][RS]  spec_a output, later dY1a
  static constexpr int oY1b = oY1a + Sh::F * FSZ;
  static constexpr int oAux = oY1b + Sh::F * FSZ;
  static constexpr int oPool = oAux + AUXP;                        // [P][RS]
  static constexpr int oMaskA = oPool + Sh::P * RS;
  static constexpr int oMaskB = oMaskA + Sh::F * MS;
  static constexpr int oZ = oMaskB + Sh::F * MS;
  static constexpr int oH = oZ + Sh::F2;
  static constexpr int oDh = oH + Sh::H;
  static constexpr int oLg = oDh + Sh::H;
  static constexpr int oDl = oLg + KMAX;
  static constexpr int oTmp = oDl + KMAX;                          // [NWH][max(2F, H)] ordered partial sums
  static constexpr int TMPW = cmax(Sh::F2, Sh::H);
  static constexpr int oGscr = oTmp + NWH * TMPW;
  static constexpr int oDww = oGscr + GSCR;
  static constexpr int oW2 = oDww + DWW;                           // staged fc2.weight rows [W2ROWS][H] + bias [KMAX]
  static constexpr int W2ROWS_ = (40960 - (oW2 + KMAX)) / Sh::H;
  static constexpr int W2ROWS = W2ROWS_ > KMAX ? KMAX : (W2ROWS_ < 0 ? 0 : W2ROWS_);
  static constexpr int TOTAL = oW2 + KMAX + W2ROWS * Sh::H;
  // LDS-DMA gather of the window: 256-float (1 KiB) pieces of the padded LDS image, NXW pieces per wave
  static constexpr int XSZ = Sh::P2 * Cs;                           // the window image itself
  static constexpr int NCH = (XSZ + 255) / 256;
  static constexpr int NXW = (NCH + Sh::NW - 1) / Sh::NW;
  static constexpr int NAUXI = (Sh::PB * Sh::C2 + 63) / 64;
  static constexpr int BYTES = TOTAL * 4;
  static_assert(oW2 == SCR + REST, "LDS carve");
  static_assert(BYTES <= 160 * 1024, "LDS budget (160 KiB per CU on gfx950)");
  static_assert(SCR % 4 == 0 && FSZ % 4 == 0 && AUXP % 4 == 0, "16-byte aligned regions");
};

// ---------------------------------------------------------------------------------------- P0 loaders
// The window gather runs asynchronously (LDS-DMA), so that work which does not need the window — the whole
// aux-branch forward — runs under its latency.
// LDS-DMA form of the gather (mode 1): `global_load_lds_dwordx4` writes LDS directly — no staging registers and no
// ds_write pass, and the wave is free to compute while its pieces are in flight.  The LDS image is lane-linear per
// piece (wave-uniform base + 16 B x lane), so the piece's lanes are pointed at the SOURCE pixels / bands that belong
// at their LDS position; lanes that fall on row padding read a harmless in-range address.  Every wave issues exactly
// NXW pieces (a wave with fewer distinct pieces repeats its last one: same bytes) so that one counted
// `s_waitcnt vmcnt(NXW)` tells a wave that the loads it issued BEFORE the window (its aux tile) have landed.
template <class Sh>
__device__ __forceinline__ void x_gather_dma(const dmf_input& in, int b, float* sX, int wave, int lane) {
  using L = Lds<Sh>;
  const int x = in.xy[2 * (size_t)b], y = in.xy[2 * (size_t)b + 1];
  const float* __restrict__ base = in.sceneA + ((size_t)x * in.Wp + y) * Sh::C;
#pragma unroll
  for (int i = 0; i < L::NXW; ++i) {
    int k = wave + i * Sh::NW;
    k = k < L::NCH ? k : L::NCH - 1;
    const int o = k * 256 + 4 * lane;                       // float offset inside the LDS image
    const int pix = o / L::Cs, within = o - pix * L::Cs;
    const int pr = pix / Sh::P, pc = pix - pr * Sh::P;
    const bool data = o < L::XSZ && within < Sh::C;
    const float* src = base + (data ? (unsigned)((pr * in.Wp + pc) * Sh::C + within) : 0u);
    if (o < L::XSZ)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sX + k * 256), 16, 0, 0);
  }
}

template <class Sh>
__device__ __forceinline__ void aux_gather_dma(const dmf_input& in, int b, float* sAux, int lane) {
  using L = Lds<Sh>;
  constexpr int ROW = Sh::SP * Sh::C2, NE = Sh::PB * Sh::C2;
  const int x = in.xy[2 * (size_t)b], y = in.xy[2 * (size_t)b + 1];
  const float* __restrict__ base = in.sceneB + ((size_t)(Sh::S * x) * in.WpB + (size_t)Sh::S * y) * Sh::C2;
#pragma unroll
  for (int i = 0; i < L::NAUXI; ++i) {
    const int e = i * 64 + lane;
    const int ec = e < NE ? e : NE - 1;
    const int r = ec / ROW, rem = ec - r * ROW;
    const float* src = base + (unsigned)(r * in.WpB * Sh::C2 + rem);
    if (e < NE)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sAux + i * 64), 4, 0, 0);
  }
}

// materialised band-major patches (the reference dataloader's tensors): transposed on the way into LDS
template <class Sh>
__device__ __forceinline__ void x_load_patches(const dmf_input& in, int b, float* __restrict__ sX, int tid) {
  constexpr int Cs = Lds<Sh>::Cs;
  const float* __restrict__ src = in.a + (size_t)b * Sh::C * Sh::P2;
  constexpr int NE = Sh::C * Sh::P2;
  constexpr int UN = 8;
  for (int e0 = 0; e0 < NE; e0 += UN * Sh::NT) {
    float v[UN];
#pragma unroll
    for (int i = 0; i < UN; ++i) {
      const int e = e0 + i * Sh::NT + tid;
      v[i] = src[e < NE ? e : NE - 1];
    }
#pragma unroll
    for (int i = 0; i < UN; ++i) {
      const int e = e0 + i * Sh::NT + tid;
      if (e < NE) {
        const int c = e / Sh::P2, pix = e - c * Sh::P2;
        sX[pix * Cs + c] = v[i];
      }
    }
  }
}

// aux tile -> LDS as [aux pixel][C2].  `nthr` threads starting at `t` cooperate (a whole workgroup, or one wavefront
// when every wave keeps its own copy of a small tile so that no workgroup barrier is needed before lift_b).
template <class Sh, int NTHR>
__device__ __forceinline__ void load_aux_tile(const dmf_input& in, int b, float* __restrict__ sAux, int t) {
  constexpr int ROW = Sh::SP * Sh::C2;
  constexpr int NE = Sh::PB * Sh::C2;
  constexpr int NIT = (NE + NTHR - 1) / NTHR;
  float v[NIT];
  if (in.mode == 1) {
    const int x = in.xy[2 * (size_t)b], y = in.xy[2 * (size_t)b + 1];
    const float* __restrict__ src = in.sceneB + ((size_t)(Sh::S * x) * in.WpB + (size_t)Sh::S * y) * Sh::C2;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int e = i * NTHR + t;
      const int ec = e < NE ? e : NE - 1;
      const int r = ec / ROW, rem = ec - r * ROW;
      v[i] = src[(unsigned)(r * in.WpB * Sh::C2 + rem)];
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int e = i * NTHR + t;
      if (e < NE) sAux[e] = v[i];
    }
  } else {
    const float* __restrict__ src = in.b + (size_t)b * Sh::C2 * Sh::PB;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int e = i * NTHR + t;
      v[i] = src[e < NE ? e : NE - 1];
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int e = i * NTHR + t;
      if (e < NE) {
        const int k = e / Sh::PB, pixb = e - k * Sh::PB;
        sAux[pixb * Sh::C2 + k] = v[i];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------- spatial stages
// A thread owns row r of channel f.  Rows r-1, r, r+1 of the channel image are fetched as whole 16-byte-aligned rows;
// a row index outside the patch is clamped and its contribution is switched off through a 0/1 factor, so there is
// no per-element bounds logic.
template <class Sh>
__device__ __forceinline__ void load_row(const float* base, float row[Lds<Sh>::RS]) {
#pragma unroll
  for (int k = 0; k < Lds<Sh>::RS / 4; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(base + 4 * k);
    row[4 * k + 0] = v.x; row[4 * k + 1] = v.y; row[4 * k + 2] = v.z; row[4 * k + 3] = v.w;
  }
}

template <class Sh>
__device__ __forceinline__ void load_dww(const float* base, float w[12]) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(base + 4 * k);
    w[4 * k + 0] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
  }
}

// forward of one branch: y = bias + depthwise 3x3 (zero pad 1); ReLU mask word; pooled partial sum of the row
template <class Sh>
__device__ __forceinline__ void row_fwd(const float* sY, const float pw[Lds<Sh>::RS], int f, int r, const float w[9],
                                        float bias, uint32_t& mask, float& z, float* yout = nullptr) {
  using L = Lds<Sh>;
  const float m0 = r > 0 ? 1.f : 0.f, m2 = r < Sh::P - 1 ? 1.f : 0.f;
  const int r0 = r > 0 ? r - 1 : 0, r2 = r < Sh::P - 1 ? r + 1 : Sh::P - 1;
  float ya[3][L::RS];
  load_row<Sh>(sY + f * L::FSZ + r0 * L::RS, ya[0]);
  load_row<Sh>(sY + f * L::FSZ + r * L::RS, ya[1]);
  load_row<Sh>(sY + f * L::FSZ + r2 * L::RS, ya[2]);
  float we[9];
#pragma unroll
  for (int v = 0; v < 3; ++v) { we[v] = w[v] * m0; we[3 + v] = w[3 + v]; we[6 + v] = w[6 + v] * m2; }
  mask = 0u;
  z = 0.f;
#pragma unroll
  for (int c = 0; c < Sh::P; ++c) {
    float y = bias;
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        const int cc = c + v - 1;
        if (cc >= 0 && cc < Sh::P) y = fmaf(we[u * 3 + v], ya[u][cc], y);
      }
    if (y > 0.f) {
      mask |= (1u << c);
      z = fmaf(pw[c], y, z);
    }
    if (yout != nullptr) yout[c] = fmaxf(y, 0.f);
  }
}

// slab rows and head vectors are written once and next read by ANOTHER kernel: streaming (non-temporal) stores keep them
// from sitting dirty in L2 until the end-of-kernel write-back
__device__ __forceinline__ void slab_put(float* p, bool first, float v) {
  __builtin_nontemporal_store(first ? v : *p + v, p);
}

// x if bit c of the row mask is set, else +0: one signed 1-bit field extract (0 / -1) and one AND instead of the
// and + compare + select the ?: form compiles to
__device__ __forceinline__ float keep_if_bit(float x, uint32_t mask, int c) {
  const int t = __builtin_amdgcn_sbfe((int)mask, c, 1);
  return __builtin_bit_cast(float, __builtin_bit_cast(int, x) & t);
}

// backward pass 1 of one branch: dW[u][v] += dY2(r,c) * Y1(r+u-1, c+v-1), db += dY2(r,c),
// with dY2(r,c) = mask(r,c) ? dz * pool[r,c] : 0
//   (DENSE: dY2(r,c) = mask(r,c) ? dd[r*P + c] : 0, dd = this channel's dense gradient map in global memory)
template <class Sh, bool DENSE = false>
__device__ __forceinline__ void row_bwd_w(const float* sY, const float pw[Lds<Sh>::RS], uint32_t mr, int f, int r, float dz,
                                          float dw[9], float& db, const float* dd = nullptr) {
  using L = Lds<Sh>;
  const float m0 = r > 0 ? 1.f : 0.f, m2 = r < Sh::P - 1 ? 1.f : 0.f;
  const int r0 = r > 0 ? r - 1 : 0, r2 = r < Sh::P - 1 ? r + 1 : Sh::P - 1;
  float ya[3][L::RS];
  load_row<Sh>(sY + f * L::FSZ + r0 * L::RS, ya[0]);
  load_row<Sh>(sY + f * L::FSZ + r * L::RS, ya[1]);
  load_row<Sh>(sY + f * L::FSZ + r2 * L::RS, ya[2]);
#pragma unroll
  for (int k = 0; k < 9; ++k) dw[k] = 0.f;
  db = 0.f;
  float ddr[L::RS];
  if (DENSE) {
#pragma unroll
    for (int k = 0; k < L::RS / 4; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(dd + r * L::RS + 4 * k);
      ddr[4 * k] = v.x; ddr[4 * k + 1] = v.y; ddr[4 * k + 2] = v.z; ddr[4 * k + 3] = v.w;
    }
  }
#pragma unroll
  for (int c = 0; c < Sh::P; ++c) {
    const float d2 = keep_if_bit(DENSE ? ddr[c] : dz * pw[c], mr, c);
    db += d2;
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        const int cc = c + v - 1;
        if (cc >= 0 && cc < Sh::P) dw[u * 3 + v] = fmaf(d2, ya[u][cc], dw[u * 3 + v]);
      }
  }
#pragma unroll
  for (int v = 0; v < 3; ++v) { dw[v] *= m0; dw[6 + v] *= m2; }
}

// backward pass 2 of one branch (after the barrier: nobody reads a neighbour's Y1 any more): dY1 of this row, in
// place.   dY1(r,c) = (Y1(r,c) > 0) * sum_{u,v} W[u][v] * dY2(r-u+1, c-v+1);  returns the row in dy[] as well
template <class Sh, bool DENSE = false>
__device__ __forceinline__ void row_bwd_x(float* sY, const float* sPool, const uint32_t* sMk, int f, int r,
                                          const float w[9], float dz, float dy[Lds<Sh>::RS], const float* dd = nullptr) {
  using L = Lds<Sh>;
  float g[3][L::RS];       // g[u][c] = dY2 at (r+u-1, c); zero outside the patch
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int rr = r + u - 1;
    const bool in = rr >= 0 && rr < Sh::P;
    const uint32_t mm = in ? sMk[f * L::MS + rr] : 0u;
    float pw[L::RS];
    if (DENSE) {
#pragma unroll
      for (int k = 0; k < L::RS / 4; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(dd + (in ? rr : r) * L::RS + 4 * k);
        pw[4 * k] = v.x; pw[4 * k + 1] = v.y; pw[4 * k + 2] = v.z; pw[4 * k + 3] = v.w;
      }
    } else {
      load_row<Sh>(sPool + (in ? rr : r) * L::RS, pw);
    }
#pragma unroll
    for (int c = 0; c < Sh::P; ++c) g[u][c] = keep_if_bit(DENSE ? pw[c] : dz * pw[c], mm, c);
  }
  float y1[L::RS];
  float* row = sY + f * L::FSZ + r * L::RS;
  load_row<Sh>(row, y1);
#pragma unroll
  for (int c = 0; c < L::RS; ++c) dy[c] = 0.f;
#pragma unroll
  for (int c = 0; c < Sh::P; ++c) {
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int v = 0; v < 3; ++v) {   // W[u][v] pairs with dY2(r-u+1, c-v+1) = g[2-u][c-v+1]
        const int cc = c - v + 1;
        if (cc >= 0 && cc < Sh::P) s = fmaf(w[u * 3 + v], g[2 - u][cc], s);
      }
    dy[c] = (y1[c] > 0.f) ? s : 0.f;
  }
#pragma unroll
  for (int k = 0; k < L::RS / 4; ++k)
    *reinterpret_cast<float4*>(row + 4 * k) = make_float4(dy[4 * k], dy[4 * k + 1], dy[4 * k + 2], dy[4 * k + 3]);
}

// spec_a of one pixel for 4 channels of one group: aligned 16-byte chunks of the pixel's band row, slot 4*q4 + i holds
// band j = 4*q4 + i - LEAD of the group (slots outside [0, Cg) belong to neighbouring groups and are skipped); the
// bands are accumulated in ascending order whatever the lead
template <class Sh, int LEAD>
__device__ __forceinline__ void spec_fwd_chunks(const float* xrow, cfloat* wg, float acc[4]) {
#pragma unroll
  for (int q4 = 0; q4 < Lds<Sh>::QG; ++q4) {
    if (4 * q4 + 3 - LEAD < 0 || 4 * q4 - LEAD >= Sh::Cg) continue;
    const float4 xv = *reinterpret_cast<const float4*>(xrow + 4 * q4);
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = 4 * q4 + i - LEAD;
      if (j >= 0 && j < Sh::Cg) {
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = fmaf(wg[m * Sh::Cg + j], xs[i], acc[m]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------- the kernel
template <class Sh, int MODE>
__global__ __launch_bounds__(Sh::NT) void patch_kernel(const KArgs a) {
  using L = Lds<Sh>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sX = smem + L::oX;
  float* sY1a = smem + L::oY1a;
  float* sY1b = smem + L::oY1b;
  float* sAux = smem + L::oAux;
  float* sPool = smem + L::oPool;
  uint32_t* sMaskA = reinterpret_cast<uint32_t*>(smem + L::oMaskA);   // [F][P] row masks of spat_a's ReLU
  uint32_t* sMaskB = reinterpret_cast<uint32_t*>(smem + L::oMaskB);
  float* sZ = smem + L::oZ;
  float* sH = smem + L::oH;
  float* sDh = smem + L::oDh;
  float* sLg = smem + L::oLg;
  float* sDl = smem + L::oDl;
  float* sTmp = smem + L::oTmp;
  float* sGscr = smem + L::oGscr;
  constexpr int TMPW = L::TMPW;

  const int tid0 = threadIdx.x;
  const int lane = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const float* __restrict__ th = a.theta;
  cfloat* thc = (cfloat*)a.theta;
  const int K = a.K;
  const int B = a.in.B;
  // this workgroup's slab row, piece-major (dmf_shapes.h): element p lives at a.slab[slab_index(blockIdx.x, p, gridDim.x)]
  auto slab_at = [&](int p) -> float* { return a.slab + slab_index(blockIdx.x, p, gridDim.x); };

  STAMP(11);
  // Prologue.  The patch-invariant tables (pooling profile, depthwise taps, fc2 rows) are fetched into registers
  // first, the first patch's window gather is issued second, and only then do the tables go to LDS: their global-load
  // latency and the gather's overlap, and the final barrier waits on LDS traffic only (the gather stays in flight).
  constexpr int NPRE_P = (Sh::P * L::RS + Sh::NT - 1) / Sh::NT, NPRE_D = (2 * Sh::F * 12 + Sh::NT - 1) / Sh::NT;
  constexpr int NPRE_W = (L::W2ROWS * Sh::H + Sh::NT - 1) / Sh::NT;
  constexpr bool AUX_WAVE = (Sh::PB * Sh::C2 <= 256);        // small aux tile: every wave loads its own copy
  float* sDww = smem + L::oDww;
  float* sB2 = smem + L::oW2;                 // fc2.bias [KMAX]
  float* sW2 = sB2 + KMAX;                    // fc2.weight rows 0..min(K, W2ROWS)-1, [row][H]
  const int kst = K < L::W2ROWS ? K : L::W2ROWS;
  float preP[NPRE_P], preD[NPRE_D], preW[NPRE_W], preB = 0.f;
#pragma unroll
  for (int q = 0; q < NPRE_P; ++q) {          // pooling profile, rows padded to RS
    const int i = tid0 + q * Sh::NT, r = i / L::RS, c = i - r * L::RS;
    preP[q] = (i < Sh::P * L::RS && c < Sh::P) ? a.pool[r * Sh::P + c] : 0.f;
  }
#pragma unroll
  for (int q = 0; q < NPRE_D; ++q) {          // depthwise taps + bias of both branches: [branch][F][12]
    const int i = tid0 + q * Sh::NT;
    const int br = i / (Sh::F * 12), rem = i - br * (Sh::F * 12);
    const int f = rem / 12, k = rem - f * 12;
    float v = 0.f;
    if (i < 2 * Sh::F * 12) {
      if (k < 9) v = th[(unsigned)((br ? Sh::oB2w : Sh::oA2w) + f * 9 + k)];
      else if (k == 9) v = th[(unsigned)((br ? Sh::oB2b : Sh::oA2b) + f)];
    }
    preD[q] = v;
  }
#pragma unroll
  for (int q = 0; q < NPRE_W; ++q) {
    const int i = tid0 + q * Sh::NT;
    preW[q] = i < kst * Sh::H ? th[(unsigned)(Sh::oFc2w + i)] : 0.f;
  }
  if (tid0 < KMAX) preB = tid0 < K ? th[(unsigned)(Sh::oFc2w + K * Sh::H + tid0)] : 0.f;
  if ((MODE == MODE_TRAIN || MODE == MODE_DENSE) && a.adam_step != nullptr && blockIdx.x == 0 && tid0 == 0) *a.adam_step += 1;
  const int boff = (a.in.cursor != nullptr) ? a.in.cursor[0] * B : 0;   // epoch-plan offset of this batch
  STAMP(13);
  const bool pre_issued = AUX_WAVE && a.in.mode == 1 && (int)blockIdx.x < B;
  int label0 = 0;                             // (fetched before the gather: a younger load could only be waited for
  if (MODE == MODE_TRAIN && pre_issued) label0 = a.labels[boff + blockIdx.x];   //  together with the whole window)
  if (pre_issued) {
    aux_gather_dma<Sh>(a.in, boff + blockIdx.x, sAux, lane);
    x_gather_dma<Sh>(a.in, boff + blockIdx.x, sX, wave, lane);
  }
  STAMP(14);
#pragma unroll
  for (int q = 0; q < NPRE_P; ++q) { const int i = tid0 + q * Sh::NT; if (i < Sh::P * L::RS) sPool[i] = preP[q]; }
#pragma unroll
  for (int q = 0; q < NPRE_D; ++q) { const int i = tid0 + q * Sh::NT; if (i < 2 * Sh::F * 12) sDww[i] = preD[q]; }
#pragma unroll
  for (int q = 0; q < NPRE_W; ++q) { const int i = tid0 + q * Sh::NT; if (i < kst * Sh::H) sW2[i] = preW[q]; }
  if (tid0 < KMAX) sB2[tid0] = preB;
  LDS_BARRIER();
  STAMP(12);

  constexpr int N1 = (Sh::F2 + 7) / 8;
  constexpr int NPH = (Sh::P2 + 63) / 64;
  static_assert(Sh::H == 64, "the single-wave fc2 / softmax / dh step maps lane <-> hidden unit");

  bool first = true;
  for (int b = blockIdx.x; b < B; b += gridDim.x, first = false) {
    int tid = tid0;
    OPAQUE(tid);    // roles are re-derived from an opaque copy per patch (and again per backward phase) so their
                    // address arithmetic lives inside the phase that uses it instead of being hoisted and spilled
    // spatial stages: thread <-> (channel fS, row rS);  fc1: thread <-> (row jH, part pH), column i = pH + 8*m
    int fS = tid >> 4, rS = tid & 15;
    const bool spat = fS < Sh::F;
    const unsigned fSc = spat ? fS : Sh::F - 1;
    const int jH = tid >> 3, pH = tid & 7;
    const int pixl = lane;     // lane <-> pixel of a wave-task's 64-pixel half
    int label = 0;             // fetched now, needed in the head
    if (MODE == MODE_TRAIN) {
      label = (first && pre_issued) ? label0 : a.labels[boff + b];
      label = label < 0 ? 0 : (label >= K ? K - 1 : label);
      label = __builtin_amdgcn_readfirstlane(label);
    }
    // ------------------------------------------------------------------ P0: issue the window gather
    STAMP(0);
    if (a.in.mode == 1) {
      if (first && pre_issued) {
        // issued in the prologue
      } else {
      if (AUX_WAVE) {
        aux_gather_dma<Sh>(a.in, boff + b, sAux, lane);               // every wave fetches its own (identical) copy
      } else {
        load_aux_tile<Sh, Sh::NT>(a.in, boff + b, sAux, tid);
        __syncthreads();
      }
      x_gather_dma<Sh>(a.in, boff + b, sX, wave, lane);
      }
      // this wave's aux pieces were issued before its NXW window pieces: they have landed once at most NXW loads
      // are outstanding (vmcnt counts in issue order)
      if (AUX_WAVE) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L::NXW) : "memory");
    } else {
      load_aux_tile<Sh, Sh::NT>(a.in, boff + b, sAux, tid);
      x_load_patches<Sh>(a.in, boff + b, sX, tid);
      __syncthreads();
    }
    float wB[12];
    load_dww<Sh>(sDww + (Sh::F + fSc) * 12, wB);
    const float bB = wB[9];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ------------------------------------------------------------------ aux branch forward, under the gather latency
    // lift_b (SxS stride-S conv + ReLU): wave-task = (block of 4 channels, 64-pixel half), weights in SGPRs
    for (int task = wave; task < Sh::NB * NPH; task += Sh::NW) {
      const int fb = task % Sh::NB, ph = task / Sh::NB;
      const int pix = ph * 64 + pixl;
      const bool valid = pix < Sh::P2;
      const int pixc = valid ? pix : Sh::P2 - 1;
      const int pr = pixc / Sh::P, pc = pixc - pr * Sh::P;
      float ax[Sh::TB];
#pragma unroll
      for (int k = 0; k < Sh::C2; ++k)
#pragma unroll
        for (int u = 0; u < Sh::S; ++u)
#pragma unroll
          for (int v = 0; v < Sh::S; ++v)
            ax[(k * Sh::S + u) * Sh::S + v] = sAux[((Sh::S * pr + u) * Sh::SP + (Sh::S * pc + v)) * Sh::C2 + k];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int f = fb * 4 + m;
        float acc = thc[Sh::oB1b + f];
#pragma unroll
        for (int q = 0; q < Sh::TB; ++q) acc = fmaf(thc[Sh::oB1w + f * Sh::TB + q], ax[q], acc);
        if (valid) sY1b[f * L::FSZ + pr * L::RS + pc] = fmaxf(acc, 0.f);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // spat_b: depthwise 3x3 + ReLU + pooling on the wave's own 4 channels
    float zb = 0.f;
    float ytokA[Sh::P], ytokB[Sh::P];          // MODE_TOKENS: this thread's rows of the two feature maps
    if (spat && rS < Sh::P) {
      float pwrow[L::RS];
      load_row<Sh>(sPool + rS * L::RS, pwrow);
      uint32_t mkb;
      row_fwd<Sh>(sY1b, pwrow, fS, rS, wB, bB, mkb, zb, MODE == MODE_TOKENS ? ytokB : nullptr);
      sMaskB[fS * L::MS + rS] = mkb;
    }
    zb = sum16(zb);
    if (spat && rS == 0) sZ[Sh::F + fS] = zb;
    STAMP(1);

    // ------------------------------------------------------------------ the window must have landed (the compiler
    // drains vmcnt before the barrier; every wave has issued all its pieces above)
    __syncthreads();
    float wA[12];
    load_dww<Sh>(sDww + fSc * 12, wA);
    const float bA = wA[9];
    STAMP(2);

    // ------------------------------------------------------------------ P1: spec_a (grouped 1x1) + ReLU
    // wave-task = (block of 4 channels, 64-pixel half); the block's group supplies the bands; weights in SGPRs
    for (int task = wave; task < Sh::NB * NPH; task += Sh::NW) {
      const int blk = task % Sh::NB, ph = task / Sh::NB;
      const int g = (4 * blk) / Sh::M, mo = 4 * blk - g * Sh::M;     // group and first output inside it
      const int pix = ph * 64 + pixl;
      const bool valid = pix < Sh::P2;
      const int pixc = valid ? pix : Sh::P2 - 1;
      const int pr = pixc / Sh::P, pc = pixc - pr * Sh::P;
      cfloat* wg = thc + Sh::oA1w + (g * Sh::M + mo) * Sh::Cg;       // wave-uniform -> SGPRs
      float acc[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m] = thc[Sh::oA1b + g * Sh::M + mo + m];
      const int lead = Sh::MISAL ? ((g * Sh::Cg) & 3) : 0;          // wave-uniform
      const float* xrow = sX + pixc * L::Cs + g * Sh::Cg - lead;     // 16-byte aligned
      switch (lead) {
        case 0: spec_fwd_chunks<Sh, 0>(xrow, wg, acc); break;
        case 1: if constexpr (Sh::MISAL) spec_fwd_chunks<Sh, 1>(xrow, wg, acc); break;
        case 2: if constexpr (Sh::MISAL) spec_fwd_chunks<Sh, 2>(xrow, wg, acc); break;
        default: if constexpr (Sh::MISAL) spec_fwd_chunks<Sh, 3>(xrow, wg, acc); break;
      }
      if (valid) {
#pragma unroll
        for (int m = 0; m < 4; ++m) sY1a[(4 * blk + m) * L::FSZ + pr * L::RS + pc] = fmaxf(acc[m], 0.f);
      }
    }
    // head weights: issued here, used after spat_a, so their L2 latency hides under it
    float w1r[N1];
#pragma unroll
    for (int m = 0; m < N1; ++m) {
      const int i = pH + 8 * m;
      w1r[m] = (MODE != MODE_DENSE && jH < Sh::H && i < Sh::F2) ? th[(unsigned)(Sh::oFc1w + jH * Sh::F2 + i)] : 0.f;
    }
    const float b1H = (MODE != MODE_DENSE && jH < Sh::H) ? th[(unsigned)(Sh::oFc1b + jH)] : 0.f;
    // From here to the head every wave works on the 4 channels it has just produced (lanes = 4 channels x 16 rows):
    // no workgroup barrier, only the wave's own LDS ordering.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ------------------------------------------------------------------ P2: spat_a depthwise 3x3 + ReLU + pooling
    {
      float za = 0.f;
      if (spat && rS < Sh::P) {
        uint32_t mka;
        float pwrow[L::RS];
        load_row<Sh>(sPool + rS * L::RS, pwrow);
        row_fwd<Sh>(sY1a, pwrow, fS, rS, wA, bA, mka, za, MODE == MODE_TOKENS ? ytokA : nullptr);
        sMaskA[fS * L::MS + rS] = mka;
      }
      za = sum16(za);
      if (spat && rS == 0) sZ[fS] = za;
    }
    LDS_BARRIER();
    STAMP(3);
    if constexpr (MODE == MODE_TOKENS && Sh::P2 <= 128) {
      // every wave is past spec_a, so the window is dead: stage the two [128][64] bf16 token maps in its place
      // (row stride 72 halves keeps the 2-byte scatter off a single bank), then copy them out in 16-byte pieces
      constexpr int TS = 72;
      unsigned short* sTok = reinterpret_cast<unsigned short*>(sX);
      static_assert(2 * 128 * TS * 2 <= L::SCR * 4, "token staging fits in the window region");
      static_assert(L::NCH * 256 >= 0, "");
      for (int i = tid; i < 2 * 128 * TS / 2; i += Sh::NT) reinterpret_cast<uint32_t*>(sTok)[i] = 0u;
      LDS_BARRIER();
      if (spat && rS < Sh::P) {
#pragma unroll
        for (int c = 0; c < Sh::P; ++c) {
          const int t = rS * Sh::P + c;
          sTok[t * TS + fS] = __builtin_bit_cast(unsigned short, (__bf16)ytokA[c]);
          sTok[128 * TS + t * TS + fS] = __builtin_bit_cast(unsigned short, (__bf16)ytokB[c]);
        }
      }
      LDS_BARRIER();
      for (int i = tid; i < 2 * 128 * 8; i += Sh::NT) {          // 2 maps x 128 tokens x 8 pieces of 16 bytes
        const int mp = i >> 10, rem = i & 1023, t = rem >> 3, pc8 = rem & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(sTok + mp * 128 * TS + t * TS + pc8 * 8);
        unsigned short* dst = (mp ? a.tokB : a.tokA) + ((size_t)b * 128 + t) * 64 + pc8 * 8;
        *reinterpret_cast<uint4*>(dst) = v;
      }
      if (tid < Sh::F2) a.zout[(size_t)b * Sh::F2 + tid] = sZ[tid];
      LDS_BARRIER();
      continue;
    }

    // ------------------------------------------------------------------ P3: head
    if constexpr (MODE != MODE_DENSE) {
    {   // fc1 + ReLU, all waves: thread (jH, pH) holds W1[jH][pH + 8m]
      float acc = 0.f;
#pragma unroll
      for (int m = 0; m < N1; ++m) {
        const int i = pH + 8 * m;
        acc = fmaf(w1r[m], (i < Sh::F2) ? sZ[i] : 0.f, acc);
      }
      acc = sum8(acc);
      if (jH < Sh::H && pH == 0) sH[jH] = fmaxf(acc + b1H, 0.f);
    }
    LDS_BARRIER();
    {   // fc2, all waves: thread (k = jH, pH) sums W2[k][pH + 8m] h[pH + 8m]; rows staged in LDS (beyond: from L2)
      constexpr int N2 = (Sh::H + 7) / 8;
      float acc = 0.f;
      if (jH < K) {
        if (jH < kst) {
#pragma unroll
          for (int m = 0; m < N2; ++m) acc = fmaf(sW2[jH * Sh::H + pH + 8 * m], sH[pH + 8 * m], acc);
        } else {
#pragma unroll
          for (int m = 0; m < N2; ++m) acc = fmaf(th[(unsigned)(Sh::oFc2w + jH * Sh::H + pH + 8 * m)], sH[pH + 8 * m], acc);
        }
      }
      acc = sum8(acc);
      if (jH < K && pH == 0) sLg[jH] = acc + sB2[jH];
    }
    LDS_BARRIER();
    float lg = -INFINITY, loss_b = 0.f;     // wave 0: this patch's logit of class `lane`, its CE loss
    int pred_b = 0;
    if (wave == 0) {
      // softmax cross-entropy and dh in ONE wavefront (lane <-> class k, then lane <-> hidden unit j), no barriers
      lg = lane < K ? sLg[lane] : -INFINITY;
      const float mx = wave_max(lg);
      {
        const unsigned long long bal = __ballot(lg == mx);
        pred_b = __ffsll((long long)bal) - 1;                      // first maximal index, as torch.max
      }
      if (MODE != MODE_FWD) {
        float dl = 0.f;
        if (MODE == MODE_TRAIN) {
          const float e = lane < K ? expf(lg - mx) : 0.f;
          const float se = wave_sum_dpp(e);
          const float ls = a.loss_scale * (a.scaler != nullptr ? a.scaler[0] : 1.f);
          dl = lane < K ? (e / se - (lane == label ? 1.f : 0.f)) * ls : 0.f;
          const float lgt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lg), label));
          loss_b = (mx + logf(se)) - lgt;
        } else {
          dl = lane < K ? a.dlogits[(size_t)b * K + lane] : 0.f;
        }
        sDl[lane] = dl;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // dh[j] = relu'(h[j]) * sum_k W2[k][j] dl[k]
        float dh = 0.f;
        if (K <= kst) {
          for (int k = 0; k < K; ++k) dh = fmaf(sW2[k * Sh::H + lane], sDl[k], dh);
        } else {
          for (int k = 0; k < K; ++k) dh = fmaf(th[(unsigned)(Sh::oFc2w + k * Sh::H + lane)], sDl[k], dh);
        }
        sDh[lane] = sH[lane] > 0.f ? dh : 0.f;
      }
    }
    LDS_BARRIER();
    STAMP(4);
    // global results of the head leave after the barrier, off the critical path: wave 0 stores what it holds in
    // registers, the last wave copies the head vectors the gradient-reduce kernel needs from LDS
    if (wave == 0) {
      if ((MODE != MODE_BWD || a.logits != nullptr) && lane < K) a.logits[(size_t)b * K + lane] = lg;
      if (a.pred != nullptr && lane == 0) a.pred[b] = pred_b;
      if (MODE == MODE_TRAIN && lane == 0) a.loss[b] = loss_b;
    }
    if (MODE == MODE_FWD) continue;       // (the next patch's gather is fenced by its own barriers)
    if (wave == Sh::NW - 1) {
      const size_t hv = hv_index(b, lane, B);            // strip-major head vectors (dmf_shapes.h)
      a.ws_h[hv] = sH[lane];
      a.ws_dh[hv] = sDh[lane];
      a.ws_dl[hv] = sDl[lane];
      for (int i = lane; i < Sh::F2; i += 64) a.ws_z[hv_index(b, i, B)] = sZ[i];
    }
    {   // dz[i] = sum_j W1[j][i] dh[j]: 8 rows of a wave by DPP / permlane swaps (lane stride 8), waves in fixed order
      const float dh = (jH < Sh::H) ? sDh[jH] : 0.f;
#pragma unroll
      for (int m = 0; m < N1; ++m) {
        const float p = sum_hi8(w1r[m] * dh);
        const int i = pH + 8 * m;
        if (wave < L::NWH && (lane >> 3) == 0 && i < Sh::F2) sTmp[wave * TMPW + i] = p;
      }
    }
    LDS_BARRIER();
    STAMP(5);
    }   // MODE != MODE_DENSE

    // ------------------------------------------------------------------ P4: depthwise backward
    OPAQUE(tid);
    fS = tid >> 4; rS = tid & 15;
    float dza = 0.f, dzb = 0.f;
    constexpr bool DN = (MODE == MODE_DENSE);
    const float* dda = DN ? a.dYa + ((size_t)b * Sh::F + fSc) * (Sh::P * L::RS) : nullptr;
    const float* ddb = DN ? a.dYb + ((size_t)b * Sh::F + fSc) * (Sh::P * L::RS) : nullptr;
    if (!DN && spat) {   // dz[f] = ordered sum of the head waves' partials
#pragma unroll
      for (int w = 0; w < L::NWH; ++w) { dza += sTmp[w * TMPW + fS]; dzb += sTmp[w * TMPW + Sh::F + fS]; }
    }
    {
      const bool act = spat && rS < Sh::P;
      float dwa[9], dwb[9], dba = 0.f, dbb = 0.f;
      if (act) {
        float pw[L::RS];
        load_row<Sh>(sPool + rS * L::RS, pw);
        row_bwd_w<Sh, DN>(sY1a, pw, sMaskA[fS * L::MS + rS], fS, rS, dza, dwa, dba, dda);
        row_bwd_w<Sh, DN>(sY1b, pw, sMaskB[fS * L::MS + rS], fS, rS, dzb, dwb, dbb, ddb);
      } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) { dwa[k] = 0.f; dwb[k] = 0.f; }
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) { dwa[k] = sum16(dwa[k]); dwb[k] = sum16(dwb[k]); }
      dba = sum16(dba);
      dbb = sum16(dbb);
      if (spat && rS == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const int oa = Sh::oA2w + fS * 9 + k, ob = Sh::oB2w + fS * 9 + k;
          slab_put(slab_at(oa), first, dwa[k]);
          slab_put(slab_at(ob), first, dwb[k]);
        }
        slab_put(slab_at(Sh::oA2b + fS), first, dba);
        slab_put(slab_at(Sh::oB2b + fS), first, dbb);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();       // the wave's own window reads of Y1 are done; neighbours never read them
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    STAMP(6);
    OPAQUE(tid);
    fS = tid >> 4; rS = tid & 15;
    {   // dY1 rows overwrite Y1 in place; spec_a / lift_b bias and lift_b weight gradients ride along
      float ga = 0.f, gb = 0.f, dwl[Sh::TB];
#pragma unroll
      for (int q = 0; q < Sh::TB; ++q) dwl[q] = 0.f;
      if (spat && rS < Sh::P) {
        float dya[L::RS], dyb[L::RS];
        row_bwd_x<Sh, DN>(sY1a, sPool, sMaskA, fS, rS, wA, dza, dya, dda);
        row_bwd_x<Sh, DN>(sY1b, sPool, sMaskB, fS, rS, wB, dzb, dyb, ddb);
#pragma unroll
        for (int c = 0; c < Sh::P; ++c) {
          ga += dya[c];
          gb += dyb[c];
#pragma unroll
          for (int k = 0; k < Sh::C2; ++k)
#pragma unroll
            for (int u = 0; u < Sh::S; ++u)
#pragma unroll
              for (int v = 0; v < Sh::S; ++v)
                dwl[(k * Sh::S + u) * Sh::S + v] =
                    fmaf(dyb[c], sAux[((Sh::S * rS + u) * Sh::SP + (Sh::S * c + v)) * Sh::C2 + k], dwl[(k * Sh::S + u) * Sh::S + v]);
        }
      }
      ga = sum16(ga);
      gb = sum16(gb);
#pragma unroll
      for (int q = 0; q < Sh::TB; ++q) dwl[q] = sum16(dwl[q]);
      if (spat && rS == 0) {
        slab_put(slab_at(Sh::oA1b + fS), first, ga);
        slab_put(slab_at(Sh::oB1b + fS), first, gb);
#pragma unroll
        for (int q = 0; q < Sh::TB; ++q) {
          const int o = Sh::oB1w + fS * Sh::TB + q;
          slab_put(slab_at(o), first, dwl[q]);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    STAMP(7);

    // ------------------------------------------------------------------ P5: spec_a weight grad, per wave:
    //   dW[f][j] = sum_pix dY1a[f][pix] * X[pix][g*Cg + j]   for the wave's 4 channels f and its group's Cg bands.
    //   lane <-> (band chunk cc, pixel slice sl); slice partials go through LDS and are summed in fixed order.
    OPAQUE(tid);
    STAMP(8);
    if (wave < Sh::NB) {
      constexpr int QG = L::QG, NSLW = L::NSLW;
      const int ln = tid & 63;
      const int cc = ln % QG, sl = ln / QG;
      const int blk = wave;
      const int g = (4 * blk) / Sh::M;
      const int lead = Sh::MISAL ? ((g * Sh::Cg) & 3) : 0;          // slot 4*cc + i holds band 4*cc + i - lead of the group
      const int band0 = g * Sh::Cg - lead + 4 * cc;
      float acc[4][4];
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m][0] = acc[m][1] = acc[m][2] = acc[m][3] = 0.f;
      if (sl < NSLW) {
        const float* dyb = sY1a + (4 * blk) * L::FSZ;
        constexpr int DR = NSLW / Sh::P, DC = NSLW - DR * Sh::P;      // pix += NSLW  ->  row += DR (+1), col += DC (-P)
        int pr = sl / Sh::P, pc = sl - pr * Sh::P;
        const float* xp = sX + sl * L::Cs + band0;
        for (int pix = sl; pix < Sh::P2; pix += NSLW) {
          const float4 xv = *reinterpret_cast<const float4*>(xp);
          const float* dy = dyb + pr * L::RS + pc;
          xp += NSLW * L::Cs;
          pr += DR; pc += DC;
          if (pc >= Sh::P) { pc -= Sh::P; pr += 1; }
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const float d = dy[m * L::FSZ];
            acc[m][0] = fmaf(d, xv.x, acc[m][0]);
            acc[m][1] = fmaf(d, xv.y, acc[m][1]);
            acc[m][2] = fmaf(d, xv.z, acc[m][2]);
            acc[m][3] = fmaf(d, xv.w, acc[m][3]);
          }
        }
      }
      STAMP(9);
      // partial (sl, m, band) -> scratch.  OWN_SLICE: row (sl*4 + m) of this wave's private band columns of the X
      // tile (only this wave ever touches them, and it has finished reading them: same wave, program order).
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if constexpr (Sh::MISAL) {
        // packed scratch: partial (row = sl*4 + m, slot) -> element e = row * 4QG + slot of a [rows][PW] array that
        // lives in the group's private band columns (first aligned column at or after the group's first band)
        float* pbase = sX + g * Sh::Cg + ((4 - lead) & 3);
        constexpr int SW = 4 * QG;
        if (sl < NSLW) {
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const int e = (sl * 4 + m) * SW + 4 * cc;               // 4 consecutive slots never straddle a PW row: PW % 4 == 0
            *reinterpret_cast<float4*>(pbase + (e / L::PW) * L::Cs + (e % L::PW)) = make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int e0 = ln; e0 < 4 * Sh::Cg; e0 += 64) {
          const int m = e0 / Sh::Cg, j = e0 - m * Sh::Cg;
          float sacc = 0.f;
#pragma unroll
          for (int q = 0; q < NSLW; ++q) {
            const int e = (q * 4 + m) * SW + j + lead;
            sacc += pbase[(e / L::PW) * L::Cs + (e % L::PW)];
          }
          const int o = Sh::oA1w + (4 * blk + m) * Sh::Cg + j;
          slab_put(slab_at(o), first, sacc);
        }
      } else {
      float* scr = L::OWN_SLICE ? (sX + g * Sh::Cg) : (sGscr + blk * (NSLW * 4 * Sh::Cg));
      constexpr int SROW = L::OWN_SLICE ? L::Cs : Sh::Cg;
      if (sl < NSLW) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
          *reinterpret_cast<float4*>(scr + (sl * 4 + m) * SROW + 4 * cc) = make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (int e = ln; e < 4 * Sh::Cg; e += 64) {
        const int m = e / Sh::Cg, j = e - m * Sh::Cg;
        float sacc = 0.f;
#pragma unroll
        for (int q = 0; q < NSLW; ++q) sacc += scr[(q * 4 + m) * SROW + j];
        const int o = Sh::oA1w + (4 * blk + m) * Sh::Cg + j;
        slab_put(slab_at(o), first, sacc);
      }
      }   // aligned groups
    }
    LDS_BARRIER();
    STAMP(10);
  }
}

// ---------------------------------------------------------------------------------------- launch
template <class Sh>
static hipError_t launch_patch(int mode, const KArgs& a, hipStream_t st) {
  using L = Lds<Sh>;
  const int grid = a.in.B < MAX_BLOCKS ? a.in.B : MAX_BLOCKS;
  hipError_t e = hipSuccess;
  static LdsAttrOnce once[5];
  auto set_attr = [&](const void* fn, int m) { e = once[m].set(fn, L::BYTES); };
  if (grid <= 0) return hipSuccess;
  switch (mode) {
    case MODE_FWD:
      set_attr(reinterpret_cast<const void*>(&patch_kernel<Sh, MODE_FWD>), 0);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL((patch_kernel<Sh, MODE_FWD>), dim3(grid), dim3(Sh::NT), L::BYTES, st, a);
      break;
    case MODE_TRAIN:
      set_attr(reinterpret_cast<const void*>(&patch_kernel<Sh, MODE_TRAIN>), 1);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL((patch_kernel<Sh, MODE_TRAIN>), dim3(grid), dim3(Sh::NT), L::BYTES, st, a);
      break;
    case MODE_TOKENS:
      set_attr(reinterpret_cast<const void*>(&patch_kernel<Sh, MODE_TOKENS>), 3);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL((patch_kernel<Sh, MODE_TOKENS>), dim3(grid), dim3(Sh::NT), L::BYTES, st, a);
      break;
    case MODE_DENSE:
      if constexpr (Sh::P2 <= 128 && Sh::S == 1 && Sh::F == 40) {      // the shapes the attention kernel is built for
        set_attr(reinterpret_cast<const void*>(&patch_kernel<Sh, MODE_DENSE>), 4);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((patch_kernel<Sh, MODE_DENSE>), dim3(grid), dim3(Sh::NT), L::BYTES, st, a);
      } else {
        return hipErrorInvalidValue;
      }
      break;
    default:
      set_attr(reinterpret_cast<const void*>(&patch_kernel<Sh, MODE_BWD>), 2);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL((patch_kernel<Sh, MODE_BWD>), dim3(grid), dim3(Sh::NT), L::BYTES, st, a);
      break;
  }
  return hipGetLastError();
}

// Compiled instances.  (C, C2, P, S, F, G, H)
using ShapeHSI = Shape<200, 1, 11, 1, 40, 10, 64>;    // BASELINE configs 1-3: 200-band HSI + 1-band SAR/LiDAR, 11x11
using ShapeHSI32 = Shape<200, 1, 11, 1, 32, 8, 64>;    // the same data at gmf.width 32: 8 groups of 25 bands, 8 wavefronts (2 per SIMD)
using ShapeHSI224 = Shape<224, 3, 11, 1, 32, 8, 64>;  // BASELINE config 4: 224-band HSI + 3-band SAR (gmf.width 32)
using ShapePanMs = Shape<4, 1, 16, 4, 40, 1, 64>;     // the reference's own data: 4-band MS + PAN at 4x, patch 16
using ShapeTiny = Shape<8, 1, 5, 4, 40, 2, 64>;       // small test scene (tests/golden/g9_trajectory.npz)
using ShapeTiny1 = Shape<8, 1, 5, 1, 40, 2, 64>;      // small test scene, equal resolution
using ShapeQua = Shape<4, 1, 16, 1, 40, 1, 64>;       // stage 2 of the two-stage path: one 4-band stream + its band mean
using ShapeQuaTiny = Shape<4, 1, 5, 1, 40, 1, 64>;    // the same on the small test scene

template <class Sh>
static bool matches(const dmf_shape& s) {
  return s.C == Sh::C && s.C2 == Sh::C2 && s.P == Sh::P && s.S == Sh::S && s.F == Sh::F && s.G == Sh::G && s.H == Sh::H;
}

int patch_shape_supported(const dmf_shape& s) {
  if (s.K < 1 || s.K > KMAX) return 0;
  return matches<ShapeHSI>(s) || matches<ShapeHSI32>(s) || matches<ShapeHSI224>(s) || matches<ShapePanMs>(s) || matches<ShapeTiny>(s) ||
         matches<ShapeTiny1>(s) || matches<ShapeQua>(s) || matches<ShapeQuaTiny>(s);
}

hipError_t patch_dispatch(const dmf_shape& s, int mode, const KArgs& a, hipStream_t st) {
  if (matches<ShapeHSI>(s)) return launch_patch<ShapeHSI>(mode, a, st);
  if (matches<ShapeHSI32>(s)) return launch_patch<ShapeHSI32>(mode, a, st);
  if (matches<ShapeHSI224>(s)) return launch_patch<ShapeHSI224>(mode, a, st);
  if (matches<ShapePanMs>(s)) return launch_patch<ShapePanMs>(mode, a, st);
  if (matches<ShapeTiny>(s)) return launch_patch<ShapeTiny>(mode, a, st);
  if (matches<ShapeTiny1>(s)) return launch_patch<ShapeTiny1>(mode, a, st);
  if (matches<ShapeQua>(s)) return launch_patch<ShapeQua>(mode, a, st);
  if (matches<ShapeQuaTiny>(s)) return launch_patch<ShapeQuaTiny>(mode, a, st);
  return hipErrorInvalidValue;
}

#ifdef DMF_STAMPS
hipError_t set_stamps(unsigned long long* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof(p)); }
#endif

}  // namespace dmf

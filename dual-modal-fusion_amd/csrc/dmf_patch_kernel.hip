// dmf_patch_kernel.hip — the hot path: ONE launch does, per patch, the dual-branch forward, the head,
// the softmax cross-entropy and the full backward (weight-gradient slabs), entirely on chip.
//
// Replaces (reference): `output = self.cur_model(data1, data2)`, `loss = self.loss(output, target.long())`,
// `loss.backward()`  — solver/mainsolver.py:52-54 — and, in MODE_FWD, the eval forward + argmax
// (mainsolver.py:109,139,169-170).  The arithmetic is the GMFNet stated in oracle/gmfnet_ref.py.
//
// Design (gfx950): one 512-thread workgroup (8 wave64) owns one patch at a time; the whole patch window
// (P*P pixels x C bands, pixel-major) is staged once into LDS with 16-byte coalesced loads, every
// activation stays in LDS/registers, and the only HBM writes are logits/loss, a handful of per-patch head
// vectors and ONE gradient slab row per workgroup.  Reductions are fixed-order (no float atomics), so a
// step is bitwise reproducible.
//
//   P0  gather  X[pix][band]  (scene rows are P*C contiguous floats -> float4 loads), aux tile, pool profile
//   P1  spec_a  grouped 1x1:  wave-task = (group, 64-pixel half); lane <-> pixel, weights wave-uniform (SGPR)
//       lift_b  SxS stride-S conv
//   P2  spat_a / spat_b depthwise 3x3: thread <-> (channel,row), 3x3 window slides along the row in registers;
//       ReLU masks kept as one 32-bit word per (channel,row); anchor-Gaussian pooling partials
//   P3  head: fc1 / fc2 from LDS-staged weights, softmax-CE by wavefront shuffles, dlogits, dh, dz
//   P4  backward of the depthwise stages from the row masks (dY2 = mask * dz[f] * pool[pix]),
//       dW/db partials per row -> fixed-order row sum -> slab; dY1 rows overwrite Y1 in LDS
//   P5  lift_b / bias gradients; X tile re-staged (2nd read, L2-resident) -> spec_a weight gradient
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/dmf.h"
#include "dmf_shapes.h"

namespace dmf {

enum { MODE_FWD = 0, MODE_TRAIN = 1, MODE_BWD = 2 };

struct KArgs {
  dmf_input in;
  const float* theta;
  const float* pool;
  const int32_t* labels;
  const float* dlogits;
  float loss_scale;
  float* logits;
  float* loss;
  int32_t* pred;
  float* slab;   // [grid][SLAB]
  float* ws_z;   // [B][2F]
  float* ws_h;   // [B][H]
  float* ws_dh;  // [B][H]
  float* ws_dl;  // [B][KMAX]
  int32_t K;
};

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
__device__ __forceinline__ float sum8(float v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  return v;
}

template <class Sh>
struct Lds {
  static constexpr int cmax(int a, int b) { return a > b ? a : b; }
  static constexpr int MB = (Sh::M % 4 == 0) ? 4 : Sh::M;          // outputs handled per spec_a-gradient unit
  static constexpr int Q = Sh::C / 4;                              // 16-byte band chunks per pixel
  static constexpr int UNITS = Q * (Sh::M / MB);
  static constexpr int NSL = (Sh::NT / UNITS) < 32 ? (Sh::NT / UNITS) : 32;  // pixel slices of the spec_a gradient
  static constexpr int SCR = cmax(cmax(Sh::P2 * Sh::Cs, Sh::P * 2 * Sh::F * 10),
                                  cmax(Sh::H * Sh::F2 + KMAX * Sh::H, NSL * Sh::F * Sh::Cg));
  static constexpr int AUXP = (Sh::PB * Sh::C2 + 3) & ~3;
  static constexpr int P2P = (Sh::P2 + 3) & ~3;
  // offsets in floats
  static constexpr int oX = 0;                         // fwd: X tile; later: staged fc weights / scratch
  static constexpr int oY1a = oX + SCR;
  static constexpr int oY1b = oY1a + Sh::P2 * Sh::Fs;
  static constexpr int oAux = oY1b + Sh::P2 * Sh::Fs;
  static constexpr int oPool = oAux + AUXP;
  static constexpr int oMaskA = oPool + P2P;
  static constexpr int oMaskB = oMaskA + Sh::F * Sh::P;
  static constexpr int oZrow = oMaskB + Sh::F * Sh::P;  // [P][2F]   (also the dz partial scratch: needs P >= 4)
  static constexpr int oZ = oZrow + Sh::P * Sh::F2;
  static constexpr int oH = oZ + Sh::F2;
  static constexpr int oDh = oH + Sh::H;
  static constexpr int oDz = oDh + Sh::H;
  static constexpr int oLg = oDz + Sh::F2;
  static constexpr int oDl = oLg + KMAX;
  static constexpr int TOTAL = oDl + KMAX;
  static constexpr int BYTES = TOTAL * 4;
  static_assert(BYTES <= 160 * 1024, "LDS budget (160 KiB per CU on gfx950)");
  static_assert(Sh::P >= 4, "dz partial scratch reuses the [P][2F] row buffer");
};

// ---------------------------------------------------------------------------------------- P0 loaders
template <class Sh>
__device__ __forceinline__ void load_x_tile(const dmf_input& in, int b, float* sX, int tid) {
  if (in.mode == 1) {
    const int x = in.xy[2 * b], y = in.xy[2 * b + 1];
    constexpr int Q = Sh::C / 4;
    constexpr int NQ = Sh::P2 * Q;
    const float4* __restrict__ src = reinterpret_cast<const float4*>(in.sceneA);
#pragma unroll
    for (int q0 = 0; q0 < NQ; q0 += Sh::NT) {
      const int q = q0 + tid;
      if (q < NQ) {
        const int pix = q / Q, cc = q - pix * Q;
        const int pr = pix / Sh::P, pc = pix - pr * Sh::P;
        const size_t pixel = (size_t)(x + pr) * in.Wp + (y + pc);
        const float4 v = src[pixel * Q + cc];
        *reinterpret_cast<float4*>(sX + pix * Sh::Cs + 4 * cc) = v;
      }
    }
  } else {
    const float* __restrict__ src = in.a + (size_t)b * Sh::C * Sh::P2;
    for (int e = tid; e < Sh::C * Sh::P2; e += Sh::NT) {
      const int c = e / Sh::P2, pix = e - c * Sh::P2;
      sX[pix * Sh::Cs + c] = src[e];
    }
  }
}

template <class Sh>
__device__ __forceinline__ void load_aux_tile(const dmf_input& in, int b, float* sAux, int tid) {
  constexpr int ROW = Sh::SP * Sh::C2;
  if (in.mode == 1) {
    const int x = in.xy[2 * b], y = in.xy[2 * b + 1];
    for (int e = tid; e < Sh::PB * Sh::C2; e += Sh::NT) {
      const int r = e / ROW, rem = e - r * ROW;
      sAux[e] = in.sceneB[((size_t)(Sh::S * x + r) * in.WpB + (size_t)Sh::S * y) * Sh::C2 + rem];
    }
  } else {
    const float* __restrict__ src = in.b + (size_t)b * Sh::C2 * Sh::PB;
    for (int e = tid; e < Sh::PB * Sh::C2; e += Sh::NT) {
      const int k = e / Sh::PB, pixb = e - k * Sh::PB;
      sAux[pixb * Sh::C2 + k] = src[e];
    }
  }
}

// ---------------------------------------------------------------------------------------- P2 / P4 row walkers
template <class Sh>
__device__ __forceinline__ void load_col(const float* sY1, int f, int r, int c, float col[3]) {
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int rr = r + u - 1;
    col[u] = (rr >= 0 && rr < Sh::P) ? sY1[(rr * Sh::P + c) * Sh::Fs + f] : 0.f;
  }
}

// depthwise 3x3 (zero pad 1) + ReLU along one row; returns ReLU mask word and pooled partial sum
template <class Sh>
__device__ __forceinline__ void row_conv_fwd(const float* sY1, const float* sPool, int f, int r,
                                             const float w[9], float bias, uint32_t& mask, float& z) {
  float c0[3] = {0.f, 0.f, 0.f}, c1[3], c2[3];
  load_col<Sh>(sY1, f, r, 0, c1);
  mask = 0u;
  z = 0.f;
#pragma unroll
  for (int c = 0; c < Sh::P; ++c) {
    if (c + 1 < Sh::P) load_col<Sh>(sY1, f, r, c + 1, c2);
    else { c2[0] = c2[1] = c2[2] = 0.f; }
    float y = bias;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      y = fmaf(w[u * 3 + 0], c0[u], y);
      y = fmaf(w[u * 3 + 1], c1[u], y);
      y = fmaf(w[u * 3 + 2], c2[u], y);
    }
    if (y > 0.f) {
      mask |= (1u << c);
      z = fmaf(sPool[r * Sh::P + c], y, z);
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) { c0[u] = c1[u]; c1[u] = c2[u]; }
  }
}

// backward of the same row: dW[9], db partial sums of this row and dY1 of this row.
//   dY2(rr,cc) = mask(rr,cc) ? dzf * pool[rr,cc] : 0
//   dW[u][v]  += dY2(r,c) * Y1(r+u-1, c+v-1)
//   dY1(r,c)   = (Y1(r,c) > 0) * sum_{u,v} W[u][v] * dY2(r-u+1, c-v+1)
template <class Sh>
__device__ __forceinline__ void row_conv_bwd(const float* sY1, const uint32_t* sMask, const float* sPool,
                                             int f, int r, const float w[9], float dzf,
                                             float dw[9], float& db, float dy1[Sh::P]) {
  uint32_t mm[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int rr = r + u - 1;
    mm[u] = (rr >= 0 && rr < Sh::P) ? sMask[rr * Sh::F + f] : 0u;
  }
  auto gval = [&](int u, int c) -> float {   // dY2 at (r+u-1, c)
    const int rr = r + u - 1;
    return ((mm[u] >> c) & 1u) ? dzf * sPool[rr * Sh::P + c] : 0.f;
  };
  float y0[3] = {0.f, 0.f, 0.f}, y1[3], y2[3];
  float g0[3] = {0.f, 0.f, 0.f}, g1[3], g2[3];
  load_col<Sh>(sY1, f, r, 0, y1);
#pragma unroll
  for (int u = 0; u < 3; ++u) g1[u] = gval(u, 0);
#pragma unroll
  for (int k = 0; k < 9; ++k) dw[k] = 0.f;
  db = 0.f;
#pragma unroll
  for (int c = 0; c < Sh::P; ++c) {
    if (c + 1 < Sh::P) {
      load_col<Sh>(sY1, f, r, c + 1, y2);
#pragma unroll
      for (int u = 0; u < 3; ++u) g2[u] = gval(u, c + 1);
    } else {
#pragma unroll
      for (int u = 0; u < 3; ++u) { y2[u] = 0.f; g2[u] = 0.f; }
    }
    const float d2 = g1[1];
    db += d2;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      dw[u * 3 + 0] = fmaf(d2, y0[u], dw[u * 3 + 0]);
      dw[u * 3 + 1] = fmaf(d2, y1[u], dw[u * 3 + 1]);
      dw[u * 3 + 2] = fmaf(d2, y2[u], dw[u * 3 + 2]);
    }
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u) {   // W[u][v] pairs with dY2(r-u+1, c-v+1) = G[2-u][2-v]
      s = fmaf(w[u * 3 + 0], g2[2 - u], s);
      s = fmaf(w[u * 3 + 1], g1[2 - u], s);
      s = fmaf(w[u * 3 + 2], g0[2 - u], s);
    }
    dy1[c] = (y1[1] > 0.f) ? s : 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u) { y0[u] = y1[u]; y1[u] = y2[u]; g0[u] = g1[u]; g1[u] = g2[u]; }
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
  uint32_t* sMaskA = reinterpret_cast<uint32_t*>(smem + L::oMaskA);
  uint32_t* sMaskB = reinterpret_cast<uint32_t*>(smem + L::oMaskB);
  float* sZrow = smem + L::oZrow;
  float* sZ = smem + L::oZ;
  float* sH = smem + L::oH;
  float* sDh = smem + L::oDh;
  float* sDz = smem + L::oDz;
  float* sLg = smem + L::oLg;
  float* sDl = smem + L::oDl;
  float* sW1 = sX;                       // staged fc1.weight [H][2F]   (valid P2..P3)
  float* sW2 = sX + Sh::H * Sh::F2;      // staged fc2.weight [K][H]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* __restrict__ th = a.theta;
  const int K = a.K;
  const int B = a.in.B;
  float* __restrict__ slab = a.slab + (size_t)blockIdx.x * Sh::SLAB;

  for (int i = tid; i < Sh::P2; i += Sh::NT) sPool[i] = a.pool[i];

  bool first = true;
  for (int b = blockIdx.x; b < B; b += gridDim.x, first = false) {
    // ------------------------------------------------------------------ P0
    load_x_tile<Sh>(a.in, b, sX, tid);
    load_aux_tile<Sh>(a.in, b, sAux, tid);
    __syncthreads();

    // ------------------------------------------------------------------ P1: spec_a (grouped 1x1) + ReLU
    {
      constexpr int NPH = (Sh::P2 + 63) / 64;
      for (int task = wave; task < Sh::G * NPH; task += Sh::NW) {
        const int g = task / NPH, ph = task - g * NPH;
        const int pix = ph * 64 + lane;
        const bool valid = pix < Sh::P2;
        const int pixc = valid ? pix : Sh::P2 - 1;
        const float* __restrict__ wg = th + Sh::oA1w + g * Sh::M * Sh::Cg;
        float acc[Sh::M];
#pragma unroll
        for (int m = 0; m < Sh::M; ++m) acc[m] = th[Sh::oA1b + g * Sh::M + m];
        const float* xr = sX + pixc * Sh::Cs + g * Sh::Cg;
#pragma unroll
        for (int j4 = 0; j4 < Sh::Cg / 4; ++j4) {
          const float4 xv = *reinterpret_cast<const float4*>(xr + 4 * j4);
#pragma unroll
          for (int m = 0; m < Sh::M; ++m) {
            acc[m] = fmaf(wg[m * Sh::Cg + 4 * j4 + 0], xv.x, acc[m]);
            acc[m] = fmaf(wg[m * Sh::Cg + 4 * j4 + 1], xv.y, acc[m]);
            acc[m] = fmaf(wg[m * Sh::Cg + 4 * j4 + 2], xv.z, acc[m]);
            acc[m] = fmaf(wg[m * Sh::Cg + 4 * j4 + 3], xv.w, acc[m]);
          }
        }
        if (valid) {
#pragma unroll
          for (int m = 0; m < Sh::M; ++m) sY1a[pix * Sh::Fs + g * Sh::M + m] = fmaxf(acc[m], 0.f);
        }
      }
      // lift_b (SxS stride-S conv) + ReLU
      for (int e = tid; e < Sh::P2 * Sh::F; e += Sh::NT) {
        const int pix = e / Sh::F, f = e - pix * Sh::F;
        const int r = pix / Sh::P, c = pix - r * Sh::P;
        float acc = th[Sh::oB1b + f];
#pragma unroll
        for (int k = 0; k < Sh::C2; ++k)
#pragma unroll
          for (int u = 0; u < Sh::S; ++u)
#pragma unroll
            for (int v = 0; v < Sh::S; ++v)
              acc = fmaf(th[Sh::oB1w + f * Sh::TB + (k * Sh::S + u) * Sh::S + v],
                         sAux[((Sh::S * r + u) * Sh::SP + (Sh::S * c + v)) * Sh::C2 + k], acc);
        sY1b[pix * Sh::Fs + f] = fmaxf(acc, 0.f);
      }
    }
    __syncthreads();

    // ------------------------------------------------------------------ P2: depthwise 3x3 + ReLU + pooling
    for (int i = tid; i < Sh::H * Sh::F2; i += Sh::NT) sW1[i] = th[Sh::oFc1w + i];
    for (int i = tid; i < K * Sh::H; i += Sh::NT) sW2[i] = th[Sh::oFc2w + i];
    if (tid < Sh::F * Sh::P) {
      const int f = tid % Sh::F, r = tid / Sh::F;
      float w[9];
      uint32_t mk;
      float z;
#pragma unroll
      for (int k = 0; k < 9; ++k) w[k] = th[Sh::oA2w + f * 9 + k];
      row_conv_fwd<Sh>(sY1a, sPool, f, r, w, th[Sh::oA2b + f], mk, z);
      sMaskA[r * Sh::F + f] = mk;
      sZrow[r * Sh::F2 + f] = z;
#pragma unroll
      for (int k = 0; k < 9; ++k) w[k] = th[Sh::oB2w + f * 9 + k];
      row_conv_fwd<Sh>(sY1b, sPool, f, r, w, th[Sh::oB2b + f], mk, z);
      sMaskB[r * Sh::F + f] = mk;
      sZrow[r * Sh::F2 + Sh::F + f] = z;
    }
    __syncthreads();

    // ------------------------------------------------------------------ P3: head
    if (tid < Sh::F2) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < Sh::P; ++r) s += sZrow[r * Sh::F2 + tid];
      sZ[tid] = s;
    }
    __syncthreads();
    {
      const int j = tid >> 3, part = tid & 7;
      float acc = 0.f;
      if (j < Sh::H)
        for (int i = part; i < Sh::F2; i += 8) acc = fmaf(sW1[j * Sh::F2 + i], sZ[i], acc);
      acc = sum8(acc);
      if (j < Sh::H && part == 0) sH[j] = fmaxf(acc + th[Sh::oFc1b + j], 0.f);
    }
    __syncthreads();
    {
      const int k = tid >> 3, part = tid & 7;
      float acc = 0.f;
      if (k < K)
        for (int j = part; j < Sh::H; j += 8) acc = fmaf(sW2[k * Sh::H + j], sH[j], acc);
      acc = sum8(acc);
      if (k < K && part == 0) sLg[k] = acc + th[Sh::oFc2w + K * Sh::H + k];
    }
    __syncthreads();
    if (wave == 0) {   // softmax cross-entropy by wavefront shuffles (one wave64 covers K <= 64 logits)
      const float v = lane < K ? sLg[lane] : -INFINITY;
      const float mx = wave_max(v);
      if (MODE != MODE_BWD || a.logits != nullptr) {
        if (lane < K) a.logits[(size_t)b * K + lane] = v;
      }
      if (a.pred != nullptr) {
        const unsigned long long bal = __ballot(v == mx);
        if (lane == 0) a.pred[b] = __ffsll((long long)bal) - 1;   // first maximal index, as torch.max
      }
      if (MODE == MODE_TRAIN) {
        const float e = lane < K ? expf(v - mx) : 0.f;
        const float s = wave_sum(e);
        int t = a.labels[b];
        t = t < 0 ? 0 : (t >= K ? K - 1 : t);
        if (lane < K) sDl[lane] = (e / s - (lane == t ? 1.f : 0.f)) * a.loss_scale;
        if (lane == 0) a.loss[b] = (mx + logf(s)) - sLg[t];
      } else if (MODE == MODE_BWD) {
        if (lane < K) sDl[lane] = a.dlogits[(size_t)b * K + lane];
      }
    }
    if (MODE == MODE_FWD) {
      __syncthreads();
      continue;
    }
    __syncthreads();

    // head backward: dh, then dz (4 partial sums through LDS, fixed order)
    if (tid < Sh::H) {
      float acc = 0.f;
      const float hv = sH[tid];
      if (hv > 0.f)
        for (int k = 0; k < K; ++k) acc = fmaf(sW2[k * Sh::H + tid], sDl[k], acc);
      sDh[tid] = acc;
      a.ws_h[(size_t)b * Sh::H + tid] = hv;
      a.ws_dh[(size_t)b * Sh::H + tid] = acc;
    } else if (tid >= 64 && tid < 64 + KMAX) {
      const int k = tid - 64;
      a.ws_dl[(size_t)b * KMAX + k] = k < K ? sDl[k] : 0.f;
    } else if (tid >= 128 && tid < 128 + Sh::F2) {
      a.ws_z[(size_t)b * Sh::F2 + (tid - 128)] = sZ[tid - 128];
    }
    __syncthreads();
    if (tid < 4 * Sh::F2) {
      const int part = tid / Sh::F2, i = tid - part * Sh::F2;
      float acc = 0.f;
      for (int j = part; j < Sh::H; j += 4) acc = fmaf(sW1[j * Sh::F2 + i], sDh[j], acc);
      sZrow[part * Sh::F2 + i] = acc;
    }
    __syncthreads();
    if (tid < Sh::F2)
      sDz[tid] = (sZrow[tid] + sZrow[Sh::F2 + tid]) + (sZrow[2 * Sh::F2 + tid] + sZrow[3 * Sh::F2 + tid]);
    __syncthreads();

    // ------------------------------------------------------------------ P4: depthwise backward
    float dy1a[Sh::P], dy1b[Sh::P];
    float* sScr = sX;   // [P][2][F][10] row partials of (dW[9], db)
    if (tid < Sh::F * Sh::P) {
      const int f = tid % Sh::F, r = tid / Sh::F;
      float w[9], dw[9], db;
#pragma unroll
      for (int k = 0; k < 9; ++k) w[k] = th[Sh::oA2w + f * 9 + k];
      row_conv_bwd<Sh>(sY1a, sMaskA, sPool, f, r, w, sDz[f], dw, db, dy1a);
      float* dst = sScr + ((r * 2 + 0) * Sh::F + f) * 10;
#pragma unroll
      for (int k = 0; k < 9; ++k) dst[k] = dw[k];
      dst[9] = db;
#pragma unroll
      for (int k = 0; k < 9; ++k) w[k] = th[Sh::oB2w + f * 9 + k];
      row_conv_bwd<Sh>(sY1b, sMaskB, sPool, f, r, w, sDz[Sh::F + f], dw, db, dy1b);
      dst = sScr + ((r * 2 + 1) * Sh::F + f) * 10;
#pragma unroll
      for (int k = 0; k < 9; ++k) dst[k] = dw[k];
      dst[9] = db;
    }
    __syncthreads();
    if (tid < Sh::F * Sh::P) {   // dY1 rows overwrite Y1 (all window reads of Y1 are done)
      const int f = tid % Sh::F, r = tid / Sh::F;
#pragma unroll
      for (int c = 0; c < Sh::P; ++c) {
        sY1a[(r * Sh::P + c) * Sh::Fs + f] = dy1a[c];
        sY1b[(r * Sh::P + c) * Sh::Fs + f] = dy1b[c];
      }
    }
    for (int e = tid; e < 2 * Sh::F * 10; e += Sh::NT) {
      const int br = e / (Sh::F * 10), rem = e - br * (Sh::F * 10);
      const int f = rem / 10, j = rem - f * 10;
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < Sh::P; ++r) s += sScr[((r * 2 + br) * Sh::F + f) * 10 + j];
      const int off = (j < 9) ? ((br ? Sh::oB2w : Sh::oA2w) + f * 9 + j) : ((br ? Sh::oB2b : Sh::oA2b) + f);
      slab[off] = first ? s : slab[off] + s;
    }
    __syncthreads();

    // ------------------------------------------------------------------ P5: lift_b / bias grads, spec_a weight grad
    load_x_tile<Sh>(a.in, b, sX, tid);   // 2nd read of the window (the fwd copy was recycled as scratch)
    {
      constexpr int NQ = Sh::TB + 2;     // q < TB: lift tap, q == TB: lift bias, q == TB+1: spec_a bias
      for (int e = tid; e < Sh::F * NQ * 8; e += Sh::NT) {
        const int sl = e & 7, fq = e >> 3;
        const int f = fq / NQ, q = fq - f * NQ;
        float acc = 0.f;
        if (q < Sh::TB) {
          const int k = q / (Sh::S * Sh::S), uv = q - k * (Sh::S * Sh::S);
          const int u = uv / Sh::S, v = uv - u * Sh::S;
          for (int pix = sl; pix < Sh::P2; pix += 8) {
            const int r = pix / Sh::P, c = pix - r * Sh::P;
            acc = fmaf(sY1b[pix * Sh::Fs + f], sAux[((Sh::S * r + u) * Sh::SP + (Sh::S * c + v)) * Sh::C2 + k], acc);
          }
        } else {
          const float* sY = (q == Sh::TB) ? sY1b : sY1a;
          for (int pix = sl; pix < Sh::P2; pix += 8) acc += sY[pix * Sh::Fs + f];
        }
        acc = sum8(acc);
        if (sl == 0) {
          const int off = (q < Sh::TB) ? (Sh::oB1w + f * Sh::TB + q) : (q == Sh::TB ? Sh::oB1b + f : Sh::oA1b + f);
          slab[off] = first ? acc : slab[off] + acc;
        }
      }
    }
    __syncthreads();
    {
      constexpr int Q = L::Q, MB = L::MB, UNITS = L::UNITS, NSL = L::NSL;
      const int u = tid % UNITS, sl = tid / UNITS;
      const int cc = u % Q, mb = u / Q;
      const int g = (4 * cc) / Sh::Cg;
      float acc[MB][4];
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[m][0] = acc[m][1] = acc[m][2] = acc[m][3] = 0.f;
      if (sl < NSL) {
        for (int pix = sl; pix < Sh::P2; pix += NSL) {
          const float4 xv = *reinterpret_cast<const float4*>(sX + pix * Sh::Cs + 4 * cc);
          const float* dy = sY1a + pix * Sh::Fs + g * Sh::M + mb * MB;
#pragma unroll
          for (int m = 0; m < MB; ++m) {
            const float d = dy[m];
            acc[m][0] = fmaf(d, xv.x, acc[m][0]);
            acc[m][1] = fmaf(d, xv.y, acc[m][1]);
            acc[m][2] = fmaf(d, xv.z, acc[m][2]);
            acc[m][3] = fmaf(d, xv.w, acc[m][3]);
          }
        }
      }
      __syncthreads();   // X tile is dead: recycle it as [NSL][F*Cg] partials
      if (sl < NSL) {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          const int fo = g * Sh::M + mb * MB + m;
          float* dst = sX + sl * (Sh::F * Sh::Cg) + fo * Sh::Cg + (4 * cc - g * Sh::Cg);
          dst[0] = acc[m][0]; dst[1] = acc[m][1]; dst[2] = acc[m][2]; dst[3] = acc[m][3];
        }
      }
      __syncthreads();
      for (int e = tid; e < Sh::F * Sh::Cg; e += Sh::NT) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < NSL; ++q) s += sX[q * (Sh::F * Sh::Cg) + e];
        slab[Sh::oA1w + e] = first ? s : slab[Sh::oA1w + e] + s;
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------- launch
template <class Sh>
static hipError_t launch_patch(int mode, const KArgs& a, hipStream_t st) {
  using L = Lds<Sh>;
  const int grid = a.in.B < MAX_BLOCKS ? a.in.B : MAX_BLOCKS;
  hipError_t e = hipSuccess;
  static bool attr_done[3] = {false, false, false};
  auto set_attr = [&](const void* fn, int m) {
    if (!attr_done[m]) {
      e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, L::BYTES);
      attr_done[m] = (e == hipSuccess);
    }
  };
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
using ShapeHSI224 = Shape<224, 3, 11, 1, 40, 8, 64>;  // BASELINE config 4: 224-band HSI + 3-band SAR
using ShapePanMs = Shape<4, 1, 16, 4, 40, 1, 64>;     // the reference's own data: 4-band MS + PAN at 4x, patch 16
using ShapeTiny = Shape<8, 1, 5, 4, 40, 2, 64>;       // small test scene (tests/golden/g9_trajectory.npz)
using ShapeTiny1 = Shape<8, 1, 5, 1, 40, 2, 64>;      // small test scene, equal resolution

template <class Sh>
static bool matches(const dmf_shape& s) {
  return s.C == Sh::C && s.C2 == Sh::C2 && s.P == Sh::P && s.S == Sh::S && s.F == Sh::F && s.G == Sh::G && s.H == Sh::H;
}

int patch_shape_supported(const dmf_shape& s) {
  if (s.K < 1 || s.K > KMAX || s.attention != 0) return 0;
  return matches<ShapeHSI>(s) || matches<ShapeHSI224>(s) || matches<ShapePanMs>(s) || matches<ShapeTiny>(s) ||
         matches<ShapeTiny1>(s);
}

hipError_t patch_dispatch(const dmf_shape& s, int mode, const KArgs& a, hipStream_t st) {
  if (matches<ShapeHSI>(s)) return launch_patch<ShapeHSI>(mode, a, st);
  if (matches<ShapeHSI224>(s)) return launch_patch<ShapeHSI224>(mode, a, st);
  if (matches<ShapePanMs>(s)) return launch_patch<ShapePanMs>(mode, a, st);
  if (matches<ShapeTiny>(s)) return launch_patch<ShapeTiny>(mode, a, st);
  if (matches<ShapeTiny1>(s)) return launch_patch<ShapeTiny1>(mode, a, st);
  return hipErrorInvalidValue;
}

}  // namespace dmf

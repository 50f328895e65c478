// dmf_patch_v2.hip — the train / eval step of the late-fusion GMFNet as ONE launch, second design.
//
// Replaces (reference): `output = self.cur_model(data1, data2)`, `loss = self.loss(output, target.long())`,
// `loss.backward()` — solver/mainsolver.py:52-54 — and the eval forward + argmax (mainsolver.py:109,139,169-170).
// Arithmetic: oracle/gmfnet_ref.py.  Same inputs, outputs and workspace contract as dmf_patch_kernel.hip (which
// stays the generic kernel: S > 1, misaligned band groups, token / dense modes).
//
// What is different from the first design, and why (round-1 profile: one patch per CU is a latency chain):
//   * A wavefront owns 4 feature channels of BOTH branches end to end, lanes = 4 channels x 16 patch rows, and
//     every intermediate (spec_a rows, depthwise rows, masks, gradients) lives in registers: neighbouring patch rows
//     are neighbouring lanes, fetched by DPP row shifts.  No feature map is ever written to LDS, so the conv pipeline
//     has no LDS write -> wait -> read hops and no workgroup barrier.
//   * Each wave gathers ONLY its own band group of the window (Cg bands x P*P pixels, 80-byte runs) by LDS-DMA into a
//     private slice: no cross-wave dependency on the gather, the wave starts spec_a when ITS bytes have landed.
//   * The network is piecewise linear and channel f reaches the head only through the pooled scalars z_a[f], z_b[f].
//     So the whole conv backward is formed with a UNIT upstream gradient right behind the forward (while the rows are
//     in registers) and scaled by dL/dz[f] at the very end.  The aux branch (forward AND unit backward) therefore runs
//     under the window gather, and the head (fc1, fc2, softmax-CE, dh, dz) runs on a dedicated wavefront concurrently
//     with the primary branch's unit backward instead of in front of it.
//   * The aux rows are loaded global -> registers (never LDS): the compiler orders every LDS access behind ALL
//     outstanding LDS-DMA (vmcnt(0)), so any LDS read in the aux phase would serialise it behind the window gather.
//   * Prologue: the per-lane DMA source offsets do not depend on the pixel coordinates, so they are computed while
//     the coordinates are in flight; coordinates come through the scalar cache.
//
// Per patch: barrier 1 = "z complete" (conv waves -> head wave), barrier 2 = "dz complete" (head -> conv waves).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "dmf_kargs.h"
#include "dmf_lanes.h"

namespace dmf {

// Diagnostic build only (-DDMF_STAMPS, tools/phase_profile_v2.py): every wavefront keeps its clock stamps in scalar
// registers (no wait, no store, no branch inside the phases) and lane 0 dumps them right before the wave ends.
#ifdef DMF_STAMPS
__device__ unsigned long long* g_v2stamps = nullptr;     // [block][16 waves][16 stamps]
#define VSTAMP_DECL unsigned long long vst_[12] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull}
#define VSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(vst_[i])); __builtin_amdgcn_sched_barrier(0); } while (0)
#define VSTAMP_DUMP() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (lane == 0 && g_v2stamps != nullptr) { \
    _Pragma("unroll") for (int i_ = 0; i_ < 12; ++i_) g_v2stamps[((size_t)blockIdx.x * 16 + wave) * 16 + i_] = vst_[i_]; } } while (0)
#else
#define VSTAMP_DECL do { } while (0)
#define VSTAMP(i) do { } while (0)
#define VSTAMP_DUMP() do { } while (0)
#endif

typedef const int32_t __attribute__((address_space(4))) cint;

template <class Sh>
struct V2 {
  static constexpr int P = Sh::P, P2 = Sh::P2, Cg = Sh::Cg, C2 = Sh::C2, F = Sh::F, F2 = Sh::F2, H = Sh::H;
  static constexpr bool OK = Sh::S == 1 && C2 <= 4 && Sh::M % 4 == 0 && Cg % 4 == 0 && P <= 16 && H == 64 && F2 <= 128 && F % 4 == 0;
  static constexpr int NB = F / 4;                 // conv wavefronts (4 channels each)
  static constexpr int NW = NB + 1;                // + the head wavefront
  static constexpr int NT = NW * 64;
  // private window slice of a wave: [P rows][ROWQ 16-byte chunks]; a row holds P pixels x Cg bands (+ one pad chunk
  // where needed so that the row stride in chunks is odd: 16 row-lanes x ds_read_b128 then hit 64 distinct banks)
  static constexpr int QC = Cg / 4;
  static constexpr int ROWQ0 = P * QC;
  static constexpr int ROWQ = ROWQ0 | 1;
  static constexpr int ROWF = ROWQ * 4;
  static constexpr int SLQ = P * ROWQ;
  static constexpr int NPC = (SLQ + 63) / 64;      // 1-KiB LDS-DMA pieces per wave
  static constexpr int SLICE = NPC * 256;          // floats
  static constexpr int AR0 = P * C2;               // aux floats per patch row
  static constexpr int RSP = ((P + 3) / 4) * 4;    // row stride of the staged pooling profile
  static constexpr int W1TS = ((H / 4) | 1) * 4;   // row stride of the staged TRANSPOSED fc1.weight [2F][W1TS] (odd chunk count)
  static constexpr int W2S = ((H / 4) | 1) * 4;    // row stride of the staged fc2.weight
  static constexpr int oX = 0;
  static constexpr int oTh = oX + NB * SLICE;      // conv part of theta, parameter order
  static constexpr int oPool = oTh + Sh::SLAB;     // pooling profile [P][RSP]
  static constexpr int oW1T = oPool + P * RSP;     // fc1.weight transposed [2F][W1TS]  (training: dz on the conv waves)
  static constexpr int oZ = oW1T + F2 * W1TS;      // pooled features, double buffered (eval runs ahead of the head)
  static constexpr int oHv = oZ + 2 * 128;         // h [H]
  static constexpr int oDl = oHv + H;              // dlogits [KMAX]
  static constexpr int oDh = oDl + KMAX;           // dh [H]
  static constexpr int oBR = oDh + H;              // [F][16] unit gradients of the aux branch, parked while the primary branch runs
  static constexpr int oSlab = oBR + F * 16;       // [SLAB] this workgroup's weight-gradient slab row, accumulated over its patches
  static constexpr int oW2 = oSlab + Sh::SLAB;     // fc2.weight [K rounded up to 4][W2S]  (run-time K)
  static constexpr int FIXED = oW2;
  static int lds_bytes(int K) { return (FIXED + ((K + 3) & ~3) * W2S) * 4; }
};

// LDS reads the compiler's waitcnt pass cannot see.  It orders EVERY LDS access it knows of behind all outstanding
// LDS-DMA (s_waitcnt vmcnt(0)), which would put the aux phase's table reads behind the whole window gather; these go
// through inline asm, are waited for with hidden_wait(), and are only used on data written before the gather was issued.
__device__ __forceinline__ unsigned lds_addr(const float* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}
__device__ __forceinline__ float hidden_read(unsigned addr, int off) {
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(off) : "memory");
  return v;
}
__device__ __forceinline__ void hidden_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#define HIDDEN_USE(v) asm volatile("" : "+v"(v))

__device__ __forceinline__ float wave_max_dpp(float v) {  // all 64 lanes, result in every lane
#define DMF_DPP_MAX(v, CTRL) fmaxf((v), __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (v)), __builtin_bit_cast(int, (v)), (CTRL), 0xF, 0xF, false)))
  v = DMF_DPP_MAX(v, 0xB1);
  v = DMF_DPP_MAX(v, 0x4E);
  v = DMF_DPP_MAX(v, 0x141);
  v = DMF_DPP_MAX(v, 0x140);
#undef DMF_DPP_MAX
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

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }   // v_pk_fma_f32
__device__ __forceinline__ float relu_lim(float x, float lim) { return __builtin_amdgcn_fmed3f(x, 0.f, lim); }   // lim = +inf: ReLU; 0: 0

__device__ __forceinline__ float dpp_row_above(float v) {   // value held by lane-1 of the 16-lane row (patch row r-1); 0 at r = 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_row_below(float v) {   // lane+1 (patch row r+1); 0 at r = 15
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xF, 0xF, true));
}

// Depthwise 3x3 (zero pad 1) + ReLU + anchor pooling of one channel row, and — TRAIN — its backward for a UNIT
// gradient on the pooled scalar: dY2 = [y2 > 0] * pool.  A lane holds row r of its channel; rows r-1 / r+1 are the
// neighbouring lanes.  Lanes r >= P carry y1 = 0 and pool = 0, which is exactly the zero padding.
//   z  : pooled partial of this row (caller reduces over the 16 row lanes)
//   dw : unit dL/dW2[u][v] partial, db: unit dL/db2 partial, dy: unit dL/dY1 of this row (through Y1's ReLU)
template <int P, bool TR>
__device__ __forceinline__ void conv_row(const float (&y1c)[P], const float (&w)[9], float bias, const float (&pw)[P],
                                         float& z, float (&dw)[9], float& db, float (&dy)[P]) {
  float y1u[P], y1d[P], gq[P];
#pragma unroll
  for (int c = 0; c < P; ++c) { y1u[c] = dpp_row_above(y1c[c]); y1d[c] = dpp_row_below(y1c[c]); }
  z = 0.f;
#pragma unroll
  for (int c = 0; c < P; ++c) {
    float y = bias;
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const int cc = c + v - 1;
      if (cc >= 0 && cc < P) {
        y = fmaf(w[v], y1u[cc], y);
        y = fmaf(w[3 + v], y1c[cc], y);
        y = fmaf(w[6 + v], y1d[cc], y);
      }
    }
    gq[c] = y > 0.f ? pw[c] : 0.f;
    z = fmaf(gq[c], y, z);
  }
  if constexpr (TR) {
#pragma unroll
    for (int k = 0; k < 9; ++k) dw[k] = 0.f;
    db = 0.f;
#pragma unroll
    for (int c = 0; c < P; ++c) {
      db += gq[c];
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        const int cc = c + v - 1;
        if (cc >= 0 && cc < P) {
          dw[v] = fmaf(gq[c], y1u[cc], dw[v]);
          dw[3 + v] = fmaf(gq[c], y1c[cc], dw[3 + v]);
          dw[6 + v] = fmaf(gq[c], y1d[cc], dw[6 + v]);
        }
      }
    }
    // dY1(r,c) = [y1 > 0] * sum_{u,v} W[u][v] * dY2(r-u+1, c-v+1): u = 0 pairs with the row below, u = 2 with the row above
    float gu[P], gd[P];
#pragma unroll
    for (int c = 0; c < P; ++c) { gu[c] = dpp_row_above(gq[c]); gd[c] = dpp_row_below(gq[c]); }
#pragma unroll
    for (int c = 0; c < P; ++c) {
      float s = 0.f;
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        const int cc = c - v + 1;
        if (cc >= 0 && cc < P) {
          s = fmaf(w[v], gd[cc], s);
          s = fmaf(w[3 + v], gq[cc], s);
          s = fmaf(w[6 + v], gu[cc], s);
        }
      }
      dy[c] = y1c[c] > 0.f ? s : 0.f;
    }
  }
}

// The window gather: NPC 1-KiB pieces of LDS-DMA, lane l of piece i reads 16 bytes at base + off[i] into slice + 1024 i + 16 l.
// It goes through the BUFFER form (buffer_load_dwordx4 ... lds): with the FLAT-encoded global_load_lds the compiler's
// waitcnt pass marks a "pending flat" access and turns EVERY later vmcnt / lgkmcnt dependency into a wait for zero — the
// aux rows (loaded just before) would then wait for the whole window.  (Device pass only: the buffer-resource type does
// not exist in the host pass, which needs nothing but the kernel's stub.)
template <int NPC>
__device__ __forceinline__ void gather_slice(const float* base, float* slice, const int (&off)[NPC]) {
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
  for (int i = 0; i < NPC; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(slice + i * 256), 16, off[i], 0, 0, 0);
#endif
}

// INMODE: dmf_input.mode, compile time — with a run-time branch the waitcnt pass merges the two paths' states at the
// join and waits vmcnt(0) there, i.e. for the whole window, before the aux phase.
template <class Sh, int MODE, int INMODE>
__global__ __launch_bounds__(V2<Sh>::NT) void patch_v2_kernel(const KArgs a) {
  using V = V2<Sh>;
  constexpr int P = Sh::P, P2 = Sh::P2, Cg = Sh::Cg, C2 = Sh::C2, F = Sh::F, F2 = Sh::F2, H = Sh::H, QC = V::QC;
  constexpr bool TR = (MODE != MODE_FWD);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.K;
  const int B = a.in.B;
  const float* __restrict__ th = a.theta;
  float* sTh = smem + V::oTh;
  float* sPool = smem + V::oPool;
  float* sZ = smem + V::oZ;
  float* sDh = smem + V::oDh;
  float* sSlab = smem + V::oSlab;
  VSTAMP_DECL;
  VSTAMP(0);
  if (MODE == MODE_TRAIN && a.adam_step != nullptr && blockIdx.x == 0 && tid == 0) *a.adam_step += 1;
  const int boff = (a.in.cursor != nullptr) ? ((cint*)a.in.cursor)[0] * B : 0;     // epoch-plan offset of this batch
  // Patch-invariant tables -> LDS, one 16-byte piece (and one pooling weight) per thread: the conv part of theta in
  // parameter order, the pooling profile in 16-byte-aligned rows.  A lane's per-channel constants then come from LDS
  // (a handful of reads) instead of ~45 four-lane-wide global loads per wave queued in front of the gather.
  static_assert(Sh::SLAB / 4 <= V::NT && P * V::RSP <= V::NT, "one staging piece per thread");
  {
    float4 tv = make_float4(0.f, 0.f, 0.f, 0.f);
    float pv = 0.f;
    const int pr = tid / V::RSP, pc = tid - pr * V::RSP;
    if (tid < Sh::SLAB / 4) tv = *reinterpret_cast<const float4*>(th + 4 * (tid < Sh::NCONV / 4 ? tid : 0));
    if (tid < P * V::RSP && pc < P) pv = a.pool[pr * P + pc];
    if (tid < Sh::SLAB / 4) *reinterpret_cast<float4*>(sTh + 4 * tid) = tv;
    if (tid < P * V::RSP) sPool[tid] = pv;
    if (TR && tid < Sh::SLAB / 4) *reinterpret_cast<float4*>(sSlab + 4 * tid) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  static_assert(Sh::NCONV % 4 == 0, "conv parameters in whole 16-byte pieces");

  if (wave < V::NB) {
    // =============================================================================== conv wavefronts
    const int ch = lane >> 4, r = lane & 15;
    const int f = 4 * wave + ch;
    const bool act = r < P;
    const int rc = act ? r : P - 1;
    // Lanes of rows >= P are the zero padding below the patch: their Y1 is forced to 0 (ReLU limit 0 instead of +inf) and
    // their depthwise bias is hugely negative, so their ReLU gates are closed (pooling weight 0, no gradient) whatever
    // their (clamped-row) inputs are.
    const float lim = act ? INFINITY : 0.f;
    const int g = (4 * wave) / Sh::M;                         // band group of this wave's channels (wave-uniform)
    float* sX = smem + V::oX + wave * V::SLICE;
    const float* xr = sX + rc * V::ROWF;
    const unsigned aTh = lds_addr(sTh) + 4u * (unsigned)f;           // hidden-read bases (bytes)
    const unsigned aPool = lds_addr(sPool) + 4u * (unsigned)(rc * V::RSP);
    // first patch's coordinates: requested before the offset arithmetic below, which covers their latency
    int xn = 0, yn = 0;
    if (INMODE == 1 && (int)blockIdx.x < B) {
      xn = ((cint*)a.in.xy)[2 * (size_t)(boff + blockIdx.x)]; yn = ((cint*)a.in.xy)[2 * (size_t)(boff + blockIdx.x) + 1];
    }
    // ---- LDS-DMA source offsets of this lane (bytes from the patch's first pixel / this wave's first band);
    //      lanes that fall on pad chunks or beyond the slice read offset 0 (harmless, in range) into unused LDS
    int offx[V::NPC];
    {
      const int Wp = a.in.Wp;
#pragma unroll
      for (int i = 0; i < V::NPC; ++i) {
        const int n = 64 * i + lane;
        const int row = n / V::ROWQ, w = n - row * V::ROWQ;
        const int col = w / QC, q = w - col * QC;
        const bool ok = row < P && w < V::ROWQ0;
        offx[i] = ok ? ((row * Wp + col) * Sh::C + 4 * q) * 4 : 0;
      }
    }
    VSTAMP(1);

    int it = 0;
    for (int b = blockIdx.x; b < B; b += gridDim.x, ++it) {
      // ------------------------------------------------------------------ aux row -> registers, window -> LDS slice
      float ax[V::AR0];
      if constexpr (INMODE == 1) {
        const int x = xn, y = yn;
        {   // the next patch's coordinates, a whole patch ahead of their use
          const int bn = b + (int)gridDim.x < B ? b + (int)gridDim.x : b;
          xn = ((cint*)a.in.xy)[2 * (size_t)(boff + bn)]; yn = ((cint*)a.in.xy)[2 * (size_t)(boff + bn) + 1];
        }
        const float* __restrict__ srcB = a.in.sceneB + ((size_t)(x + rc) * a.in.WpB + y) * C2;
#pragma unroll
        for (int i = 0; i < V::AR0; ++i) ax[i] = srcB[i];
        // barrier 0 (first patch): the staged tables are complete — and every wave's aux-row loads are in the memory
        // pipeline AHEAD of every wave's window pieces (requests of one CU are served in issue order: an aux row issued
        // behind another wave's 10 KiB of gather would only arrive with the window)
        if (it == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // the scheduler must not sink the gather below the aux phase (nor hoist that phase above it): everything older
        // is issued, then the NPC pieces, then the aux phase runs under them
        __builtin_amdgcn_sched_barrier(0);
        gather_slice<V::NPC>(a.in.sceneA + ((size_t)x * a.in.Wp + y) * Sh::C + g * Cg, sX, offx);
        __builtin_amdgcn_sched_barrier(0);
      } else {
        // materialised band-major patches (the reference dataloader's tensors; test / drop-in path)
        const float* __restrict__ srcB = a.in.b + (size_t)(boff + b) * C2 * P2;
#pragma unroll
        for (int c = 0; c < P; ++c)
#pragma unroll
          for (int k = 0; k < C2; ++k) ax[c * C2 + k] = srcB[k * P2 + rc * P + c];
        const float* __restrict__ srcA = a.in.a + ((size_t)(boff + b) * Sh::C + g * Cg) * P2;
        for (int e = lane; e < Cg * P2; e += 64) {
          const int j = e / P2, pix = e - j * P2;
          const int pr = pix / P, pc = pix - pr * P;
          sX[pr * V::ROWF + pc * Cg + j] = srcA[e];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (it == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // barrier 0: the staged tables are complete
      }
      VSTAMP(2);

      // ------------------------------------------------------------------ aux branch, under the window gather
      float zb, dwb[9], dbb = 0.f, dwl[C2], dbl = 0.f;
      {
        float w2b[9], wl[C2], pw[P], b2b, bl;
        // (the channel-dependent part of an offset is in the base address, the rest is an instruction immediate)
#pragma unroll
        for (int k = 0; k < 9; ++k) w2b[k] = hidden_read(aTh + 4u * (unsigned)(f * 8), (Sh::oB2w + k) * 4);          // oB2w + 9f + k
        b2b = hidden_read(aTh, Sh::oB2b * 4);
        bl = hidden_read(aTh, Sh::oB1b * 4);
#pragma unroll
        for (int k = 0; k < C2; ++k) wl[k] = hidden_read(aTh + 4u * (unsigned)(f * (C2 - 1)), (Sh::oB1w + k) * 4);   // oB1w + C2 f + k
#pragma unroll
        for (int c = 0; c < P; ++c) pw[c] = hidden_read(aPool, c * 4);
        hidden_wait();
#pragma unroll
        for (int k = 0; k < 9; ++k) HIDDEN_USE(w2b[k]);
        HIDDEN_USE(b2b); HIDDEN_USE(bl);
#pragma unroll
        for (int k = 0; k < C2; ++k) HIDDEN_USE(wl[k]);
#pragma unroll
        for (int c = 0; c < P; ++c) HIDDEN_USE(pw[c]);

        float y1b[P], dyb[P];
#pragma unroll
        for (int c = 0; c < P; ++c) {
          float v = bl;
#pragma unroll
          for (int k = 0; k < C2; ++k) v = fmaf(wl[k], ax[c * C2 + k], v);
          y1b[c] = relu_lim(v, lim);
        }
        conv_row<P, TR>(y1b, w2b, act ? b2b : -1e30f, pw, zb, dwb, dbb, dyb);
        zb = sum16(zb);
        if constexpr (TR) {
#pragma unroll
          for (int k = 0; k < C2; ++k) dwl[k] = 0.f;
#pragma unroll
          for (int c = 0; c < P; ++c) {
            dbl += dyb[c];
#pragma unroll
            for (int k = 0; k < C2; ++k) dwl[k] = fmaf(dyb[c], ax[c * C2 + k], dwl[k]);
          }
#pragma unroll
          for (int k = 0; k < 9; ++k) dwb[k] = sum16(dwb[k]);
          dbb = sum16(dbb);
          dbl = sum16(dbl);
#pragma unroll
          for (int k = 0; k < C2; ++k) dwl[k] = sum16(dwl[k]);
        }
      }
      VSTAMP(3);
      // park the aux branch's unit gradients (the first LDS access the compiler sees behind the gather: it waits for the
      // wave's own LDS-DMA here, which the primary branch needs anyway)
      if constexpr (TR) {
        float* br = smem + V::oBR + f * 16;
        if (r == 0) {
#pragma unroll
          for (int k = 0; k < 9; ++k) br[k] = dwb[k];
          br[9] = dbb;
          br[10] = dbl;
#pragma unroll
          for (int k = 0; k < C2; ++k) br[11 + k] = dwl[k];
        }
      }

      // ------------------------------------------------------------------ primary branch
      float w2a[9], pw[P], b2a;
#pragma unroll
      for (int k = 0; k < 9; ++k) w2a[k] = sTh[Sh::oA2w + f * 9 + k];
      b2a = sTh[Sh::oA2b + f];
#pragma unroll
      for (int q = 0; q < V::RSP / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(sPool + rc * V::RSP + 4 * q);
        if (4 * q < P) pw[4 * q] = v.x;
        if (4 * q + 1 < P) pw[4 * q + 1] = v.y;
        if (4 * q + 2 < P) pw[4 * q + 2] = v.z;
        if (4 * q + 3 < P) pw[4 * q + 3] = v.w;
      }
      float y1a[P];
      {   // spec_a: even / odd bands of the group in the two halves of a packed accumulator
        v2f w1p[Cg / 2];
#pragma unroll
        for (int q = 0; q < QC; ++q) {
          const float4 v = *reinterpret_cast<const float4*>(sTh + Sh::oA1w + f * Cg + 4 * q);
          w1p[2 * q] = (v2f){v.x, v.y};
          w1p[2 * q + 1] = (v2f){v.z, v.w};
        }
        const float b1 = sTh[Sh::oA1b + f];
        v2f ap[P];
#pragma unroll
        for (int c = 0; c < P; ++c) ap[c] = (v2f){b1, 0.f};
#pragma unroll
        for (int c = 0; c < P; ++c)
#pragma unroll
          for (int q = 0; q < QC; ++q) {
            const float4 xv = *reinterpret_cast<const float4*>(xr + c * Cg + 4 * q);
            ap[c] = pk_fma(w1p[2 * q], (v2f){xv.x, xv.y}, ap[c]);
            ap[c] = pk_fma(w1p[2 * q + 1], (v2f){xv.z, xv.w}, ap[c]);
          }
#pragma unroll
        for (int c = 0; c < P; ++c) y1a[c] = relu_lim(ap[c].x + ap[c].y, lim);
      }
      VSTAMP(4);
      float za, dwa[9], dba = 0.f, dya[P];
      conv_row<P, TR>(y1a, w2a, act ? b2a : -1e30f, pw, za, dwa, dba, dya);
      za = sum16(za);
      {
        float* zbuf = sZ + (TR ? 0 : (it & 1) * 128);
        if (r == 0) { zbuf[f] = za; zbuf[F + f] = zb; }
      }
      LDS_BARRIER();                                     // barrier 1: z complete
      VSTAMP(5);
      if constexpr (!TR) continue;

      // ------------------------------------------------------------------ unit gradients of spec_a from the still-resident slice
      float acc[Cg], db1 = 0.f;
      {
        v2f gp[Cg / 2];
#pragma unroll
        for (int j = 0; j < Cg / 2; ++j) gp[j] = (v2f){0.f, 0.f};
#pragma unroll
        for (int c = 0; c < P; ++c) {
          db1 += dya[c];
          const v2f d2 = (v2f){dya[c], dya[c]};
#pragma unroll
          for (int q = 0; q < QC; ++q) {
            const float4 xv = *reinterpret_cast<const float4*>(xr + c * Cg + 4 * q);
            gp[2 * q] = pk_fma(d2, (v2f){xv.x, xv.y}, gp[2 * q]);
            gp[2 * q + 1] = pk_fma(d2, (v2f){xv.z, xv.w}, gp[2 * q + 1]);
          }
        }
#pragma unroll
        for (int j = 0; j < Cg / 2; ++j) { acc[2 * j] = gp[j].x; acc[2 * j + 1] = gp[j].y; }
      }
      VSTAMP(6);
#pragma unroll
      for (int j = 0; j < Cg; ++j) acc[j] = sum16(acc[j]);
#pragma unroll
      for (int k = 0; k < 9; ++k) dwa[k] = sum16(dwa[k]);
      dba = sum16(dba);
      db1 = sum16(db1);
      // this lane's slice of the two fc1 columns of its channel, fetched before the barrier: dz[i] = sum_j W1[j][i] dh[j]
      const float4 wta = *reinterpret_cast<const float4*>(smem + V::oW1T + f * V::W1TS + 4 * r);
      const float4 wtb = *reinterpret_cast<const float4*>(smem + V::oW1T + (F + f) * V::W1TS + 4 * r);
      float brv[16];                                    // the parked aux-branch unit gradients of this channel
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(smem + V::oBR + f * 16 + 4 * q);
        brv[4 * q] = v.x; brv[4 * q + 1] = v.y; brv[4 * q + 2] = v.z; brv[4 * q + 3] = v.w;
      }
      VSTAMP(7);
      LDS_BARRIER();                                     // barrier 2: dh complete
      VSTAMP(8);
      {
        const float4 dhv = *reinterpret_cast<const float4*>(sDh + 4 * r);
        float dza = fmaf(wta.x, dhv.x, fmaf(wta.y, dhv.y, fmaf(wta.z, dhv.z, wta.w * dhv.w)));
        float dzb = fmaf(wtb.x, dhv.x, fmaf(wtb.y, dhv.y, fmaf(wtb.z, dhv.z, wtb.w * dhv.w)));
        dza = sum16(dza);
        dzb = sum16(dzb);
        if (r == 0) {   // scale the unit gradients by dL/dz; one owner lane per slab element (first patch: plain stores)
          float* so = sSlab + Sh::oA1w + f * Cg;
          if (it == 0) {
#pragma unroll
            for (int q = 0; q < QC; ++q)
              *reinterpret_cast<float4*>(so + 4 * q) = make_float4(dza * acc[4 * q], dza * acc[4 * q + 1], dza * acc[4 * q + 2], dza * acc[4 * q + 3]);
            sSlab[Sh::oA1b + f] = dza * db1;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
              sSlab[Sh::oA2w + f * 9 + k] = dza * dwa[k];
              sSlab[Sh::oB2w + f * 9 + k] = dzb * brv[k];
            }
            sSlab[Sh::oA2b + f] = dza * dba;
            sSlab[Sh::oB2b + f] = dzb * brv[9];
            sSlab[Sh::oB1b + f] = dzb * brv[10];
#pragma unroll
            for (int k = 0; k < C2; ++k) sSlab[Sh::oB1w + f * C2 + k] = dzb * brv[11 + k];
          } else {
            float old[Cg + 22 + C2];
#pragma unroll
            for (int j = 0; j < Cg; ++j) old[j] = so[j];
            old[Cg] = sSlab[Sh::oA1b + f];
#pragma unroll
            for (int k = 0; k < 9; ++k) { old[Cg + 1 + k] = sSlab[Sh::oA2w + f * 9 + k]; old[Cg + 10 + k] = sSlab[Sh::oB2w + f * 9 + k]; }
            old[Cg + 19] = sSlab[Sh::oA2b + f];
            old[Cg + 20] = sSlab[Sh::oB2b + f];
            old[Cg + 21] = sSlab[Sh::oB1b + f];
#pragma unroll
            for (int k = 0; k < C2; ++k) old[Cg + 22 + k] = sSlab[Sh::oB1w + f * C2 + k];
#pragma unroll
            for (int j = 0; j < Cg; ++j) so[j] = fmaf(dza, acc[j], old[j]);
            sSlab[Sh::oA1b + f] = fmaf(dza, db1, old[Cg]);
#pragma unroll
            for (int k = 0; k < 9; ++k) {
              sSlab[Sh::oA2w + f * 9 + k] = fmaf(dza, dwa[k], old[Cg + 1 + k]);
              sSlab[Sh::oB2w + f * 9 + k] = fmaf(dzb, brv[k], old[Cg + 10 + k]);
            }
            sSlab[Sh::oA2b + f] = fmaf(dza, dba, old[Cg + 19]);
            sSlab[Sh::oB2b + f] = fmaf(dzb, brv[9], old[Cg + 20]);
            sSlab[Sh::oB1b + f] = fmaf(dzb, brv[10], old[Cg + 21]);
#pragma unroll
            for (int k = 0; k < C2; ++k) sSlab[Sh::oB1w + f * C2 + k] = fmaf(dzb, brv[11 + k], old[Cg + 22 + k]);
          }
        }
      }
      VSTAMP(9);
    }
  } else {
    // =============================================================================== head wavefront
    float* sW1T = smem + V::oW1T;
    float* sW2 = smem + V::oW2;
    float* sH = smem + V::oHv;
    float* sDl = smem + V::oDl;
    const int K4 = (K + 3) & ~3;
    __builtin_amdgcn_s_setprio(3);     // the youngest wave of its SIMD would otherwise only get the issue slots the conv waves leave
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // barrier 0 (staged tables; the conv waves wait for nothing else)
    // fc1.weight row `lane` stays in registers for fc1; training also stages it transposed for the conv waves' dz (needed
    // behind barrier 2).  fc2.weight goes to LDS as 16-byte pieces (rows up to K4 zero filled).
    float4 w1r[F2 / 4];
#pragma unroll
    for (int q = 0; q < F2 / 4; ++q) w1r[q] = *reinterpret_cast<const float4*>(th + Sh::oFc1w + lane * F2 + 4 * q);
    for (int i0 = 0; i0 < K4 / 4; i0 += 4) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = (i0 + u) * 64 + lane, k = idx >> 4;
        v[u] = (i0 + u < K4 / 4 && k < K) ? *reinterpret_cast<const float4*>(th + Sh::oFc2w + 4 * idx) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = (i0 + u) * 64 + lane, k = idx >> 4, c4 = idx & 15;
        if (i0 + u < K4 / 4) *reinterpret_cast<float4*>(sW2 + k * V::W2S + 4 * c4) = v[u];
      }
    }
    const float bh = th[Sh::oFc1b + lane];
    const float bk = lane < K ? th[Sh::oFc2w + K * H + lane] : 0.f;
    if constexpr (TR) {
#pragma unroll
      for (int q = 0; q < F2 / 4; ++q) {
        sW1T[(4 * q) * V::W1TS + lane] = w1r[q].x;
        sW1T[(4 * q + 1) * V::W1TS + lane] = w1r[q].y;
        sW1T[(4 * q + 2) * V::W1TS + lane] = w1r[q].z;
        sW1T[(4 * q + 3) * V::W1TS + lane] = w1r[q].w;
      }
    }
    VSTAMP(1);

    int it = 0;
    for (int b = blockIdx.x; b < B; b += gridDim.x, ++it) {
      int label = 0;
      float dlx = 0.f;
      if (MODE == MODE_TRAIN) {
        label = ((cint*)a.labels)[boff + b];
        label = label < 0 ? 0 : (label >= K ? K - 1 : label);
      }
      if (MODE == MODE_BWD) dlx = lane < K ? a.dlogits[(size_t)b * K + lane] : 0.f;
      LDS_BARRIER();                                     // barrier 1: z complete
      VSTAMP(5);
      const float* zbuf = sZ + (TR ? 0 : (it & 1) * 128);
      // fc1 + ReLU: lane j, its weight row in registers, z broadcast from LDS
      float h;
      {
        float s0 = bh, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int q = 0; q < F2 / 4; ++q) {
          const float4 zv = *reinterpret_cast<const float4*>(zbuf + 4 * q);
          s0 = fmaf(w1r[q].x, zv.x, s0); s1 = fmaf(w1r[q].y, zv.y, s1); s2 = fmaf(w1r[q].z, zv.z, s2); s3 = fmaf(w1r[q].w, zv.w, s3);
        }
        h = fmaxf((s0 + s1) + (s2 + s3), 0.f);
      }
      sH[lane] = h;
      float zo0 = 0.f, zo1 = 0.f;                         // this patch's pooled features, for the gradient reduce
      if constexpr (TR) {
        zo0 = zbuf[lane < F2 ? lane : F2 - 1];
        if (F2 > 64) zo1 = zbuf[64 + lane < F2 ? 64 + lane : F2 - 1];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // fc2: lane k (rows beyond K are zero rows or clamped; masked below)
      float lg;
      {
        const int kr = lane < K4 ? lane : K4 - 1;
        float s0 = bk, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int q = 0; q < H / 4; ++q) {
          const float4 wv = *reinterpret_cast<const float4*>(sW2 + kr * V::W2S + 4 * q);
          const float4 hv = *reinterpret_cast<const float4*>(sH + 4 * q);
          s0 = fmaf(wv.x, hv.x, s0); s1 = fmaf(wv.y, hv.y, s1); s2 = fmaf(wv.z, hv.z, s2); s3 = fmaf(wv.w, hv.w, s3);
        }
        lg = lane < K ? (s0 + s1) + (s2 + s3) : -INFINITY;
      }
      const float mx = wave_max_dpp(lg);
      const unsigned long long bal = __ballot(lg == mx);
      const int pred_b = __ffsll((long long)bal) - 1;              // first maximal index, as torch.max
      float loss_b = 0.f, dl = 0.f, dh = 0.f;
      if constexpr (TR) {
        if (MODE == MODE_TRAIN) {
          const float e = lane < K ? __expf(lg - mx) : 0.f;
          const float se = wave_sum_dpp(e);
          dl = lane < K ? (e / se - (lane == label ? 1.f : 0.f)) * a.loss_scale : 0.f;
          const float lgt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lg), __builtin_amdgcn_readfirstlane(label)));
          loss_b = (mx + __logf(se)) - lgt;
        } else {
          dl = dlx;
        }
        sDl[lane] = dl;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // dh[j] = relu'(h[j]) * sum_k W2[k][j] dl[k]: lane j
        {
          float s0 = 0.f, s1 = 0.f;
          for (int k = 0; k < K4; k += 4) {
            const float4 dv = *reinterpret_cast<const float4*>(sDl + k);
            s0 = fmaf(sW2[k * V::W2S + lane], dv.x, s0);
            s1 = fmaf(sW2[(k + 1) * V::W2S + lane], dv.y, s1);
            s0 = fmaf(sW2[(k + 2) * V::W2S + lane], dv.z, s0);
            s1 = fmaf(sW2[(k + 3) * V::W2S + lane], dv.w, s1);
          }
          dh = h > 0.f ? s0 + s1 : 0.f;
        }
        sDh[lane] = dh;
      }
      VSTAMP(7);
      if constexpr (TR) LDS_BARRIER();                   // barrier 2: dh complete
      VSTAMP(8);
      // global results leave after the barrier, off the conv waves' critical path
      if ((MODE != MODE_BWD || a.logits != nullptr) && lane < K) a.logits[(size_t)b * K + lane] = lg;
      if (a.pred != nullptr && lane == 0) a.pred[b] = pred_b;
      if constexpr (TR) {
        if (MODE == MODE_TRAIN && lane == 0) a.loss[b] = loss_b;
        a.ws_h[(size_t)b * H + lane] = h;
        a.ws_dh[(size_t)b * H + lane] = dh;
        a.ws_dl[(size_t)b * KMAX + lane] = dl;
        if (lane < F2) a.ws_z[(size_t)b * F2 + lane] = zo0;
        if (F2 > 64 && 64 + lane < F2) a.ws_z[(size_t)b * F2 + 64 + lane] = zo1;
      }
      VSTAMP(9);
    }
  }
  // the workgroup's slab row leaves in one coalesced pass (streaming stores: next read by the reduce kernel)
  if constexpr (TR) {
    LDS_BARRIER();
    float* __restrict__ slab = a.slab + (size_t)blockIdx.x * Sh::SLAB;
    for (int i = tid; i < Sh::SLAB / 4; i += V::NT) {
      const float4 v = *reinterpret_cast<const float4*>(sSlab + 4 * i);
      __builtin_nontemporal_store(v.x, slab + 4 * i);
      __builtin_nontemporal_store(v.y, slab + 4 * i + 1);
      __builtin_nontemporal_store(v.z, slab + 4 * i + 2);
      __builtin_nontemporal_store(v.w, slab + 4 * i + 3);
    }
  }
  VSTAMP(10);
  VSTAMP_DUMP();
}

// ---------------------------------------------------------------------------------------- launch
template <class Sh, int MODE, int INMODE>
static hipError_t launch_v2_inst(const KArgs& a, int grid, int bytes, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_v2_kernel<Sh, MODE, INMODE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL((patch_v2_kernel<Sh, MODE, INMODE>), dim3(grid), dim3(V2<Sh>::NT), bytes, st, a);
  return hipGetLastError();
}

template <class Sh>
static hipError_t launch_v2(int mode, const KArgs& a, hipStream_t st) {
  using V = V2<Sh>;
  const int grid = a.in.B < MAX_BLOCKS ? a.in.B : MAX_BLOCKS;
  if (grid <= 0) return hipSuccess;
  const int bytes = V::lds_bytes(a.K);
  if (bytes > 160 * 1024) return hipErrorInvalidValue;
  const bool gather = a.in.mode == 1;
  switch (mode) {
    case MODE_FWD:
      return gather ? launch_v2_inst<Sh, MODE_FWD, 1>(a, grid, bytes, st) : launch_v2_inst<Sh, MODE_FWD, 0>(a, grid, bytes, st);
    case MODE_TRAIN:
      return gather ? launch_v2_inst<Sh, MODE_TRAIN, 1>(a, grid, bytes, st) : launch_v2_inst<Sh, MODE_TRAIN, 0>(a, grid, bytes, st);
    case MODE_BWD:
      return gather ? launch_v2_inst<Sh, MODE_BWD, 1>(a, grid, bytes, st) : launch_v2_inst<Sh, MODE_BWD, 0>(a, grid, bytes, st);
    default:
      return hipErrorInvalidValue;
  }
}

// Compiled instances.  (C, C2, P, S, F, G, H)
using V2HSI = Shape<200, 1, 11, 1, 40, 10, 64>;      // BASELINE configs 1-2
using V2HSI224 = Shape<224, 3, 11, 1, 32, 8, 64>;    // BASELINE config 4
using V2Tiny1 = Shape<8, 1, 5, 1, 40, 2, 64>;        // small test scene, equal resolution
using V2Qua = Shape<4, 1, 16, 1, 40, 1, 64>;         // stage 2 of the two-stage path
using V2QuaTiny = Shape<4, 1, 5, 1, 40, 1, 64>;

template <class Sh>
static bool v2_matches(const dmf_shape& s) {
  static_assert(V2<Sh>::OK, "shape outside the v2 kernel's geometry");
  return s.C == Sh::C && s.C2 == Sh::C2 && s.P == Sh::P && s.S == Sh::S && s.F == Sh::F && s.G == Sh::G && s.H == Sh::H;
}

template <class Sh>
static bool v2_fits(const dmf_shape& s) { return v2_matches<Sh>(s) && V2<Sh>::lds_bytes(s.K) <= 160 * 1024; }

int patch_v2_supported(const dmf_shape& s, int mode) {
  if (mode != MODE_FWD && mode != MODE_TRAIN && mode != MODE_BWD) return 0;
  if (s.K < 1 || s.K > KMAX || s.attention) return 0;
  return v2_fits<V2HSI>(s) || v2_fits<V2HSI224>(s) || v2_fits<V2Tiny1>(s) || v2_fits<V2Qua>(s) || v2_fits<V2QuaTiny>(s);
}

hipError_t patch_v2_dispatch(const dmf_shape& s, int mode, const KArgs& a, hipStream_t st) {
  if (v2_matches<V2HSI>(s)) return launch_v2<V2HSI>(mode, a, st);
  if (v2_matches<V2HSI224>(s)) return launch_v2<V2HSI224>(mode, a, st);
  if (v2_matches<V2Tiny1>(s)) return launch_v2<V2Tiny1>(mode, a, st);
  if (v2_matches<V2Qua>(s)) return launch_v2<V2Qua>(mode, a, st);
  if (v2_matches<V2QuaTiny>(s)) return launch_v2<V2QuaTiny>(mode, a, st);
  return hipErrorInvalidValue;
}

#ifdef DMF_STAMPS
hipError_t set_v2_stamps(unsigned long long* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_v2stamps), &p, sizeof(p)); }
#endif

}  // namespace dmf

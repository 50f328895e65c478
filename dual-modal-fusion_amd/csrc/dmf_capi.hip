// dmf_capi.hip — extern "C" entry points of libdmf_hip.so (declared in include/dmf.h) and the small
// batch-level kernels around the fused patch kernel: slab/outer-product gradient reduction, fused Adam,
// on-device confusion matrix / label map, pan2ms.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dmf_kargs.h"
#include "dmf_lanes.h"
#include "dmf_xgmi.h"

namespace dmf {

static thread_local char g_err[512] = "";

static int fail(const char* fmt, const char* detail) {
  snprintf(g_err, sizeof(g_err), fmt, detail);
  return 1;
}
static int check(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return 1;
}

static Layout layout_of(const dmf_shape& s) {
  return make_layout(s.C, s.C2, s.P, s.S, s.F, s.G, s.H, s.K, s.attention, s.E);
}

// ------------------------------------------------------------------------------ gradient reduction (+Adam)
//   conv params  : grad[p] = sum_blk slab[blk][p]
//   fc1.weight   : grad = sum_b dh[b][j] * z[b][i]      fc1.bias: sum_b dh[b][j]
//   fc2.weight   : grad = sum_b dl[b][k] * h[b][j]      fc2.bias: sum_b dl[b][k]
struct ReduceArgs {
  const float* slab; const float* z; const float* h; const float* dh; const float* dl;
  int B, nblk, SLAB, NCONV, F2, H, K;
  int64_t oFc1w, oFc1b, oFc2w, oFc2b, n;
  float* grad;
  float* theta; float* m; float* v;   // Adam (theta == nullptr: reduce only)
  float lr, b1, b2, eps, bc1, bc2_sqrt;
  const int32_t* step_dev;            // optional device-side step count (overrides bc1 / bc2_sqrt)
  int32_t* cursor_dev;                // optional epoch-plan cursor to advance
  const float* loss; float* loss_hist;
  const float* aslab; int nablk, ASLAB; int64_t oAttn;   // attention weights: sum of the attention kernel's slabs
  XgmiDev x;                          // x.world > 1: exchange the gradient with the peer ranks before Adam
  float grad_scale; int seq_bias;
  float* scaler;                      // loss-scaler state (dmf_grad_reduce_scaled): grad <- sum / scaler[0], non-finite -> scaler[2]
  int dbg;                            // stamps build: probe switches (DMF_REDUCE_DBG)
};

// bias corrections from a device-resident step count, in double like torch's host-side scalars
__device__ __forceinline__ void bias_corrections(int step, float b1, float b2, float& bc1, float& bc2_sqrt) {
  bc1 = (float)(1.0 - pow((double)b1, (double)step));
  bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, (double)step));
}

__device__ __forceinline__ void adam_update(float* theta, float* m, float* v, int64_t p, float g,
                                            float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt) {
  // torch.optim.Adam single-tensor path: exp_avg.lerp_(grad, 1-b1); exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2);
  // denom = sqrt(v)/sqrt(bc2) + eps; p -= (lr/bc1) * m / denom
  const float mo = m[p], vo = v[p];
  const float mn = mo + (g - mo) * (1.f - b1);
  const float vn = vo * b2 + (1.f - b2) * g * g;
  m[p] = mn;
  v[p] = vn;
  const float denom = sqrtf(vn) / bc2_sqrt + eps;
  theta[p] -= (lr / bc1) * (mn / denom);
}

// ---- the reduce launch.  What bounds it (tools/reduce_phase_profile.py, stamps of round 3): ONE CU takes in only ~15 bytes
// per clock from the Infinity Cache (about 64 lines in flight x ~550 cycles), and the 2.2 MB the patch kernel left behind are
// nowhere else.  The first forms (16 or 64 parameters per block, 64-byte pieces of 7-KB rows, 64 KB fetched per block) spent
// 4.5 K of their 6.8 K cycles waiting for that; so the producers now lay their results out for THIS kernel (dmf_shapes.h):
//   * conv slabs piece-major: one block per 16 parameters reads rows x 64 contiguous bytes (16 KB at batch 256) — a lane
//     holds one 16-byte piece of up to 4 rows, rows are summed per lane, then over the 16 row lanes by DPP and the row /
//     half swaps, then over the 4 waves through LDS: a fixed order;
//   * fc1.weight / fc2.weight = dh^T z / dl^T h: one 8x8 output tile per block — 2 x 8 KB of contiguous strip-major head
//     vectors at batch 256 (a 16x16 tile needs 2 x 16 KB: twice the wait).  It still runs on the fp32 matrix cores
//     (v_mfma_f32_16x16x4_f32: bit for bit a k-ordered fmaf chain) with the two HALVES of the batch packed into one
//     instruction: rows 0-7 / columns 0-7 carry the first half, rows 8-15 / columns 8-15 the second, the two diagonal 8x8
//     blocks of the result are the two partial tiles (the off-diagonal blocks are discarded); wave w carries a quarter of each
//     half in two accumulators.  The tiles of the first column also sum their dh / dl strip: fc1.bias / fc2.bias;
//   * attention slabs keep the row-major form (64 parameters per block): 15.7 MB per step, bound by the chip, not the CU;
//   * ADAM's bias corrections: b^step by repeated squaring in double on two lanes of a fifth wave (beta1 / beta2 side by
//     side) beside the gradient loads, instead of two calls of the general pow() on one lane in front of the barrier.
// Block order: fc tiles first (the longest chains), then slabs, the bookkeeping block last.
typedef float f32x4_t __attribute__((ext_vector_type(4)));


__device__ __forceinline__ double powi_double(double b, int n) {   // b^n, n >= 0, by squaring (relative error ~ 2 log2(n) ulp)
  double r = 1.0;
  while (n > 0) {
    if (n & 1) r *= b;
    b *= b;
    n >>= 1;
  }
  return r;
}

// the fixed combine of 16 chunk partials ((0+8)+(4+12)) + ... (attention slabs)
__device__ __forceinline__ float tree16(float (&t)[16]) {
#pragma unroll
  for (int w = 8; w > 0; w >>= 1)
#pragma unroll
    for (int i = 0; i < w; ++i) t[i] += t[i + w];
  return t[0];
}

// One 8 x 8 tile of out[m][n] = sum_b U[b][m0 + m] * W[b][n0 + n] (U, W strip-major: hv_index; m0, n0 multiples of 8).
// MFMA row / column q < 8 works on the patches [0, Bh), q >= 8 on [Bh, B); wave w takes a quarter of each half's k-steps (4
// patches each), 8 steps per batch.  issue(): the 16 loads of one batch; consume(): its 8 MFMAs (two accumulators) and the
// running sum of the U operand (the bias gradient).  Rows >= mlim read as zero.  Threads 0..255.
struct FcTile {
  const float* up; const float* wp;
  int bbase, blim, kk, s1, sb_;
  bool mok;
  f32x4_t acc0, acc1;
  float bs;
  float av[8], bv[8];
  __device__ __forceinline__ int init(const float* U, int m0, int mlim, const float* W, int n0, int B) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q8 = lane & 7, half = (lane >> 3) & 1;
    kk = lane >> 4;
    mok = m0 + q8 < mlim;
    const int Bh = (((B + 1) >> 1) + 3) & ~3;                 // first half: a multiple of 4 patches
    bbase = half ? Bh : 0; blim = half ? B : min(Bh, B);
    up = U + (size_t)(m0 >> 3) * B * 8 + q8;                  // strip m0 / 8: element (b, m) at b * 8 + m
    wp = W + (size_t)(n0 >> 3) * B * 8 + q8;
    const int nk = Bh / 4, nkw = (nk + 3) / 4;                // k-steps per half: all, per wave
    const int s0 = w * nkw;
    s1 = min(nk, s0 + nkw);
    acc0 = (f32x4_t){0.f, 0.f, 0.f, 0.f}; acc1 = acc0; bs = 0.f;
    return s0;
  }
  // (loads are UNCONDITIONAL, from a clamped patch index, and masked in consume(): a load under a lane condition becomes a
  // branch around it, and the compiler then waits for the loads of one branch before it enters the next)
  __device__ __forceinline__ void issue(int sb) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int b = min(bbase + 4 * (sb + q) + kk, blim - 1);
      av[q] = up[(size_t)(b < 0 ? 0 : b) * 8];
      bv[q] = wp[(size_t)(b < 0 ? 0 : b) * 8];
    }
    sb_ = sb;
  }
  __device__ __forceinline__ void consume() {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const bool in = sb_ + q < s1 && bbase + 4 * (sb_ + q) + kk < blim;
      av[q] = (in && mok) ? av[q] : 0.f;
      bv[q] = in ? bv[q] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q + 1], bv[q + 1], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) bs += av[q];
  }
};

__device__ __forceinline__ void adam_apply(const ReduceArgs& a, int64_t p, float g, float th0, float m_0, float v_0, const float* bcs) {
  if (a.scaler != nullptr) {                         // unscale_ + the found_inf check of GradScaler, in the reduce
    g *= 1.f / a.scaler[0];
    if (!isfinite(g)) a.scaler[2] = 1.f;             // (every writer stores the same value)
  }
  if (a.grad != nullptr) a.grad[p] = g;
  if (a.theta != nullptr) {
    const float mn = m_0 + (g - m_0) * (1.f - a.b1);
    const float vn = v_0 * a.b2 + (1.f - a.b2) * g * g;
    a.m[p] = mn;
    a.v[p] = vn;
    a.theta[p] = th0 - (a.lr / bcs[0]) * (mn / (sqrtf(vn) / bcs[1] + a.eps));
  }
}

// Diagnostic build only (-DDMF_STAMPS, tools/reduce_phase_profile.py): clock stamps of every wave of the reduce launch in
// scalar registers, dumped by lane 0 right before the wave ends.  [block][5 waves][8]: 0 entry, 2 kernel arguments in registers, 1 role known, 3 partials
// written (loads landed), 4 behind the barrier, 5 stores issued, 6 end; 7 s_memrealtime at entry.
#ifdef DMF_STAMPS
__device__ unsigned long long* g_rstamps = nullptr;
#define RSTAMP_DECL unsigned long long rst_[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull}
#define RSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rst_[i]) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define RSTAMP_RT(i) do { asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rst_[i])); } while (0)
#define RSTAMP_DUMP() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); RSTAMP(6); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    if ((threadIdx.x & 63) == 0 && g_rstamps != nullptr) { _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) \
      g_rstamps[((size_t)blockIdx.x * 5 + (threadIdx.x >> 6)) * 8 + i_] = rst_[i_]; } } while (0)
#else
#define RSTAMP_DECL do { } while (0)
#define RSTAMP(i) do { } while (0)
#define RSTAMP_RT(i) do { } while (0)
#define RSTAMP_DUMP() do { } while (0)
#endif

// 320 threads: waves 0-3 reduce, wave 4 only forms ADAM's bias corrections (beside the other waves' gradient loads).
// The first nine arguments are everything a block needs to find its role and ISSUE its gradient loads; they are plain scalars
// in front of the argument struct so that the compiler's kernarg preload (-mllvm -amdgpu-kernarg-preload-count, build.py) puts
// them into scalar registers at wave launch: a kernel argument fetched by the wave itself arrives ~1.0 K cycles after wave
// entry (stamps), and every load of this kernel was waiting behind that.  w0 = nFc1 | t1n << 16, w1 = nFc2 | t2n << 16,
// w2 = nConv | nAttn << 16 (blocks per kind, tiles per row).
__global__ __launch_bounds__(320) void grad_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ dh,
                                                          const float* __restrict__ z, const float* __restrict__ dl,
                                                          const float* __restrict__ h, int w0, int w1, int w2, int B, const ReduceArgs a) {
  __shared__ float vbuf[4][256];        // tile partials of the four waves / [16 chunks][64] attention-slab partials / [4][16] piece partials
  __shared__ float bbuf[4][16];         // bias partials of the four waves
  __shared__ float bcs[2];
  const int tid = threadIdx.x;
  int blk = blockIdx.x;
  RSTAMP_DECL;
  RSTAMP_RT(7);
  RSTAMP(0);
  const int nFc1 = w0 & 0xffff, t1n = w0 >> 16, nFc2 = w1 & 0xffff, t2n = w1 >> 16, nConv = w2 & 0xffff, nAttn = w2 >> 16;
  const int nTotal = nFc1 + nFc2 + nConv + nAttn + 1;
  const int nblk = B < MAX_BLOCKS ? B : MAX_BLOCKS;  // slab rows: the patch kernel's grid
  if (blk == nTotal - 1) {                           // bookkeeping block
    const int cur = a.cursor_dev != nullptr ? *a.cursor_dev : 0;
    if (a.loss != nullptr && a.loss_hist != nullptr) {
      float s = 0.f;
      float* red = &vbuf[0][0];
      if (tid < 256) {
        for (int b = tid; b < a.B; b += 256) s += a.loss[b];
        red[tid] = s;
      }
      __syncthreads();
      for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
      }
      if (tid == 0) a.loss_hist[cur] = red[0] / (float)a.B;
    }
    if (tid == 0 && a.cursor_dev != nullptr) *a.cursor_dev = cur + 1;
    return;
  }
  int kind, sub;                                     // 0 fc1.weight tile, 1 fc2.weight tile, 2 conv slab piece, 3 attention slab
  if (blk < nFc1) { kind = 0; sub = blk; }
  else if ((blk -= nFc1) < nFc2) { kind = 1; sub = blk; }
  else if ((blk -= nFc2) < nConv) { kind = 2; sub = blk; }
  else { kind = 3; sub = blk - nConv; }
  // ---- the first batch of gradient loads goes out before anything else is looked at
  int m0 = 0, n0 = 0;
  FcTile ft;
  int sb = 0;
  float4 cv[4];
  const int lane = tid & 63, wv = tid >> 6, c4 = lane & 3, r16 = lane >> 2;
  const float* csrc = slab + (size_t)sub * nblk * 16 + 4 * c4;       // conv piece `sub` of every row: rows x 16 floats, contiguous
  if (kind < 2) {
    const int tn = kind == 0 ? t1n : t2n;
    const int mt = (int)((float)sub / (float)tn + 0.01f);            // (exact for the few hundred tiles there are)
    m0 = 8 * mt; n0 = 8 * (sub - mt * tn);
    if (tid < 256) {
      sb = kind == 0 ? ft.init(dh, m0, a.H, z, n0, B) : ft.init(dl, m0, a.K, h, n0, B);
      ft.issue(sb);
    }
  } else if (kind == 2 && tid < 256) {
    // lane = (16-byte quarter c4, row lane r16); wave wv, load i: rows (4 i + wv) * 16 + r16 — every wave-level load is 1 KiB
    // of contiguous bytes
#pragma unroll
    for (int i = 0; i < 4; ++i)                          // (unconditional, clamped row; masked where they are summed)
      cv[i] = *reinterpret_cast<const float4*>(csrc + (size_t)min((4 * i + wv) * 16 + r16, nblk - 1) * 16);
  }
  __builtin_amdgcn_sched_barrier(0);                 // (nothing that waits for these loads may move up here)
  RSTAMP(2);
  // ---- which parameter(s) this thread finishes (p, own; p2, own2: the bias of a first-column tile), and their ADAM state
  int64_t p = 0, p2 = 0;
  bool own = false, own2 = false;
  if (kind == 0) {
    const int j = m0 + ((tid >> 3) & 7), i = n0 + (tid & 7);
    own = tid < 64 && j < a.H && i < a.F2;
    p = a.oFc1w + (int64_t)j * a.F2 + i;
    own2 = n0 == 0 && tid < 8 && m0 + tid < a.H;
    p2 = a.oFc1b + m0 + tid;
  } else if (kind == 1) {
    const int k = m0 + ((tid >> 3) & 7), j = n0 + (tid & 7);
    own = tid < 64 && k < a.K && j < a.H;
    p = a.oFc2w + (int64_t)k * a.H + j;
    own2 = n0 == 0 && tid < 8 && m0 + tid < a.K;
    p2 = a.oFc2b + m0 + tid;
  } else if (kind == 2) {
    p = (int64_t)16 * sub + tid;
    own = tid < 16 && p < a.NCONV;
  } else {
    p = a.oAttn + (int64_t)64 * sub + tid;
    own = tid < 64 && 64 * sub + tid < a.ASLAB;
  }
  RSTAMP(1);
  float th0 = 0.f, m_0 = 0.f, v_0 = 0.f, th2 = 0.f, m_2 = 0.f, v_2 = 0.f;
  if (a.theta != nullptr) {
    if (own) { th0 = a.theta[p]; m_0 = a.m[p]; v_0 = a.v[p]; }
    if (own2) { th2 = a.theta[p2]; m_2 = a.m[p2]; v_2 = a.v[p2]; }
  }
  float* part = &vbuf[0][0];
  if (tid >= 256) {
    // bias corrections (lanes 0 / 1 of wave 4: beta1 / beta2 side by side), b^step by repeated squaring in double
    if (a.theta != nullptr && tid < 258) {
      float bc = tid == 256 ? a.bc1 : a.bc2_sqrt;
      if (a.step_dev != nullptr) {
        const double pw = powi_double((double)(tid == 256 ? a.b1 : a.b2), *a.step_dev);
        bc = tid == 256 ? (float)(1.0 - pw) : (float)sqrt(1.0 - pw);
      }
      bcs[tid - 256] = bc;
    }
  } else if (kind < 2) {
    ft.consume();
    for (sb += 8; sb < ft.s1; sb += 8) { ft.issue(sb); ft.consume(); }
    const f32x4_t v = ft.acc0 + ft.acc1;
#pragma unroll
    for (int r = 0; r < 4; ++r) vbuf[wv][(4 * ft.kk + r) * 16 + (lane & 15)] = v[r];   // C layout: row 4 (lane >> 4) + r, column lane & 15
    const float bias = swap_add32(swap_add16(ft.bs));                                  // over the four k lanes of a row
    if (lane < 16) bbuf[wv][lane] = bias;                                              // [wave][half * 8 + row]
  } else if (kind == 2) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i0 = 0;;) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool in = (4 * (i0 + i) + wv) * 16 + r16 < nblk;
        acc.x += in ? cv[i].x : 0.f; acc.y += in ? cv[i].y : 0.f; acc.z += in ? cv[i].z : 0.f; acc.w += in ? cv[i].w : 0.f;
      }
      i0 += 4;
      if ((4 * i0 + wv) * 16 >= nblk) break;                           // (more than 256 rows: not with today's MAX_BLOCKS)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        cv[i] = *reinterpret_cast<const float4*>(csrc + (size_t)min((4 * (i0 + i) + wv) * 16 + r16, nblk - 1) * 16);
    }
    // over the 16 row lanes (lane bits 2..5): xor 4 / xor 8 inside a 16-lane row by row rotations, then the row / half swaps
    float e[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float x = e[k];
      x = DMF_DPP_ADD(x, 0x124);     // row_ror:4
      x = DMF_DPP_ADD(x, 0x128);     // row_ror:8
      e[k] = swap_add32(swap_add16(x));
    }
    if (lane < 4) *reinterpret_cast<float4*>(part + wv * 16 + 4 * c4) = make_float4(e[0], e[1], e[2], e[3]);
  } else {
    const int pitch = a.ASLAB, nb = a.nablk;
    const int q4 = tid & 15, ch = tid >> 4;
    const bool in_row = 64 * sub + 4 * q4 < pitch;
    const float* sl = a.aslab + 64 * sub + 4 * q4;
    const int per = (nb + 15) / 16;
    const int lo = ch * per, hi = min(nb, lo + per);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b0 = lo; b0 < hi; b0 += 16) {
      float4 v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i)
        v[i] = (in_row && b0 + i < hi) ? *reinterpret_cast<const float4*>(sl + (size_t)(b0 + i) * pitch) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
    }
    *reinterpret_cast<float4*>(part + ch * 64 + 4 * q4) = acc;          // [16 chunks][64]
  }
  RSTAMP(3);
  __syncthreads();
  RSTAMP(4);
  float g = 0.f, g2 = 0.f;
  if (tid < 256) {
    if (kind < 2) {
      if (tid < 64) {                                  // the two diagonal blocks of every wave's result, in wave order
        const int e0 = (tid >> 3) * 16 + (tid & 7), e1 = e0 + 8 * 16 + 8;
        g = ((vbuf[0][e0] + vbuf[0][e1]) + (vbuf[1][e0] + vbuf[1][e1])) + ((vbuf[2][e0] + vbuf[2][e1]) + (vbuf[3][e0] + vbuf[3][e1]));
      }
      if (tid < 8) g2 = ((bbuf[0][tid] + bbuf[0][8 + tid]) + (bbuf[1][tid] + bbuf[1][8 + tid])) +
                        ((bbuf[2][tid] + bbuf[2][8 + tid]) + (bbuf[3][tid] + bbuf[3][8 + tid]));
    } else if (kind == 2) {
      if (tid < 16) g = (part[tid] + part[16 + tid]) + (part[32 + tid] + part[48 + tid]);
    } else if (tid < 64) {
      float t[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) t[i] = part[i * 64 + tid];
      g = tree16(t);
    }
  }
  if (a.x.world > 1) {                               // (the owning lanes exchange; no barriers inside)
    const int seq = *a.step_dev + a.seq_bias;
    g = xgmi_exchange(a.x, 0, seq, p, own, g) * a.grad_scale;
    if (kind < 2 && n0 == 0) g2 = xgmi_exchange(a.x, 0, seq, p2, own2, g2) * a.grad_scale;
  }
  if (own) adam_apply(a, p, g, th0, m_0, v_0, bcs);
  if (own2) adam_apply(a, p2, g2, th2, m_2, v_2, bcs);
  RSTAMP(5);
  RSTAMP_DUMP();
}

__global__ __launch_bounds__(256) void adam_kernel(float* theta, const float* grad, float* m, float* v, int64_t n,
                                                   float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                                   float grad_scale, const int32_t* step_dev, int32_t* cursor_dev) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (step_dev != nullptr) bias_corrections(*step_dev, b1, b2, bc1, bc2_sqrt);
  if (p < n) adam_update(theta, m, v, p, grad[p] * grad_scale, lr, b1, b2, eps, bc1, bc2_sqrt);
  if (cursor_dev != nullptr && p == 0) *cursor_dev += 1;
}

// ------------------------------------------------------------------------------ the reference's other two optimisers
// torch.optim.SGD(lr, momentum) (dampening 0, no Nesterov, no weight decay): buf = g on the first step, else m buf + g;
// p -= lr buf.  torch.optim.RMSprop(lr, alpha) (eps 1e-8, momentum 0, not centred): sq = alpha sq + (1 - alpha) g g;
// p -= lr g / (sqrt(sq) + eps).   (utils/utils.py:13-16)
__global__ __launch_bounds__(256) void sgd_kernel(float* theta, const float* grad, float* buf, int64_t n, float lr, float momentum,
                                                  float grad_scale, const int32_t* step_dev, int32_t step, int32_t* cursor_dev) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int st = step_dev != nullptr ? *step_dev : step;
  if (p < n) {
    const float g = grad[p] * grad_scale;
    float b = g;
    if (momentum != 0.f) { b = st <= 1 ? g : momentum * buf[p] + g; buf[p] = b; }
    theta[p] -= lr * b;
  }
  if (cursor_dev != nullptr && p == 0) *cursor_dev += 1;
}

__global__ __launch_bounds__(256) void rmsprop_kernel(float* theta, const float* grad, float* sq, int64_t n, float lr, float alpha,
                                                      float eps, float grad_scale, int32_t* cursor_dev) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p < n) {
    const float g = grad[p] * grad_scale;
    const float s = alpha * sq[p] + (1.f - alpha) * g * g;
    sq[p] = s;
    theta[p] -= lr * (g / (sqrtf(s) + eps));
  }
  if (cursor_dev != nullptr && p == 0) *cursor_dev += 1;
}

// ------------------------------------------------------------------------------ dynamic loss scaling (GradScaler's role)
// state: [0] scale  [1] growth tracker  [2] found_inf  [3] skipped steps  [4] ticket (int bits)
__global__ __launch_bounds__(256) void unscale_check_kernel(float* grad, int64_t n, float grad_scale, float* state) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const float g = grad[p] * (grad_scale / state[0]);
  grad[p] = g;
  if (!isfinite(g)) state[2] = 1.f;                 // (every writer stores the same value)
}

__global__ __launch_bounds__(256) void scaled_adam_kernel(float* theta, const float* grad, float* m, float* v, int64_t n,
                                                          float lr, float b1, float b2, float eps, float* state,
                                                          float growth, float backoff, int interval,
                                                          int32_t* step_dev, int32_t* cursor_dev) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool skip = state[2] != 0.f;
  if (!skip && p < n) {
    float bc1, bc2s;
    bias_corrections(*step_dev, b1, b2, bc1, bc2s);
    adam_update(theta, m, v, p, grad[p], lr, b1, b2, eps, bc1, bc2s);
  }
  // the last block to get here has seen every other block read found_inf and the step count: it closes the step
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    int* ticket = reinterpret_cast<int*>(state + 4);
    if (atomicAdd(ticket, 1) == (int)gridDim.x - 1) {
      *ticket = 0;
      if (skip) {
        state[0] *= backoff; state[1] = 0.f; state[3] += 1.f;
        *step_dev -= 1;                               // a skipped step does not count for the bias corrections
      } else {
        const float t = state[1] + 1.f;
        if (t >= (float)interval) { state[0] *= growth; state[1] = 0.f; }
        else state[1] = t;
      }
      state[2] = 0.f;
      if (cursor_dev != nullptr) *cursor_dev += 1;
    }
  }
}

// ------------------------------------------------------------------------------ eval helpers
__global__ __launch_bounds__(256) void confusion_kernel(const int32_t* pred, const int32_t* target, int B, int K,
                                                        unsigned long long* matrix) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < B) {
    const int p = pred[i], t = target[i];
    if (p >= 0 && p < K && t >= 0 && t < K) atomicAdd(&matrix[(size_t)p * K + t], 1ull);   // rows = prediction
  }
}

__global__ __launch_bounds__(256) void labelmap_kernel(const int32_t* pred, const int32_t* xy, int B, int W, int32_t* map) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < B) map[(size_t)xy[2 * i] * W + xy[2 * i + 1]] = pred[i];
}

// pan2ms (image_convert/IHS.py:14-19): p = 2x2 mean pool of pan; out[:, :, i] = p[i%2::2, i//2::2]
//   => out[h, w, i] = mean(pan[4h + 2(i%2) + {0,1}, 4w + 2(i//2) + {0,1}])
__global__ __launch_bounds__(256) void pan2ms_kernel(const double* pan, int pitch, int H, int W, double* out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)H * W * 4) return;
  const int i = (int)(e & 3);
  const int64_t hw = e >> 2;
  const int h = (int)(hw / W), w = (int)(hw - (int64_t)h * W);
  const int r = 4 * h + 2 * (i % 2), c = 4 * w + 2 * (i / 2);
  const double* p0 = pan + (size_t)r * pitch + c;
  // numpy.mean over a 2x2 block: running sum in row-major order, then / 4
  out[e] = (((p0[0] + p0[1]) + p0[pitch]) + p0[pitch + 1]) / 4.0;
}

}  // namespace dmf

using namespace dmf;

// the conv launches of the attention network (MODE_TOKENS, MODE_DENSE)
static hipError_t conv_dispatch(const dmf_shape& s, int mode, const KArgs& a, hipStream_t st) {
  if (!patch_v2_supported(s, mode)) return hipErrorInvalidValue;
  return patch_v2_dispatch(s, mode, a, st);
}

extern "C" {

int32_t dmf_version(void) { return DMF_VERSION; }
const char* dmf_last_error(void) { return g_err; }

int32_t dmf_shape_supported(const dmf_shape* s) {
  if (s == nullptr) return fail("%s", "null shape");
  if (s->attention ? (patch_v2_supported(*s, MODE_TOKENS) && attn_shape_supported(*s)) : patch_v2_supported(*s, MODE_TRAIN)) return 0;
  snprintf(g_err, sizeof(g_err),
           "no compiled kernel instance for C=%d C2=%d P=%d S=%d F=%d G=%d H=%d K=%d attention=%d; compiled (C/C2/P/S/F/G):%s "
           "(K <= 64, subject to 160 KiB of LDS at this K; attention: 200/1/11/1/40/10 and 8/1/5/1/40/2 with E = 96, 3 heads); "
           "`python dual-modal-fusion_amd/build.py --shapes FILE` adds rows",
           s->C, s->C2, s->P, s->S, s->F, s->G, s->H, s->K, s->attention, patch_v2_shape_list());
  return 1;
}

int32_t dmf_patch_variant(const dmf_shape* s, int32_t mode) {   // the same decision as run_patch / dmf_shape_supported
  if (s == nullptr) return 0;
  return patch_v2_supported(*s, mode) ? 2 : 0;
}

int32_t dmf_param_layout(const dmf_shape* s, int64_t offsets[17]) {
  if (s == nullptr || offsets == nullptr) return fail("%s", "null argument");
  const Layout L = layout_of(*s);
  for (int i = 0; i < 17; ++i) offsets[i] = L.off[i];
  return 0;
}

int64_t dmf_workspace_bytes(const dmf_shape* s, int32_t B) {
  if (s == nullptr || B < 0) return -1;
  const Layout L = layout_of(*s);
  return make_ws(L, B).total * 4;
}

static int run_patch(const dmf_shape* s, const dmf_input* in, int mode, const float* theta, const float* pool_w,
                     const int32_t* labels, const float* dlogits, float loss_scale, float* logits, float* loss,
                     int32_t* pred, void* workspace, int32_t* adam_step, void* stream, const float* scaler = nullptr) {
  if (s == nullptr || in == nullptr || theta == nullptr || pool_w == nullptr) return fail("%s", "null argument");
  if (dmf_shape_supported(s)) return 1;
  if (s->attention) return fail("%s", "attention network: use dmf_forward_attn / dmf_train_attn_fwd_bwd");
  if (in->half && dmf_half_supported(s)) return 1;
  if (in->B < 0) return fail("%s", "negative batch");
  if (in->B == 0) return 0;
  if (in->mode == 0 && (in->a == nullptr || in->b == nullptr)) return fail("%s", "mode 0 needs a and b");
  if (in->mode == 1 && (in->sceneA == nullptr || in->sceneB == nullptr || in->xy == nullptr || in->Wp <= 0 || in->WpB <= 0))
    return fail("%s", "mode 1 needs sceneA, sceneB, xy, Wp, WpB");
  if (in->mode != 0 && in->mode != 1) return fail("%s", "input mode must be 0 or 1");
  KArgs a{};
  a.in = *in;
  a.theta = theta;
  a.pool = pool_w;
  a.labels = labels;
  a.dlogits = dlogits;
  a.loss_scale = loss_scale;
  a.scaler = scaler;
  a.logits = logits;
  a.loss = loss;
  a.pred = pred;
  a.adam_step = adam_step;
  a.K = s->K;
  if (mode != MODE_FWD) {
    if (workspace == nullptr) return fail("%s", "null workspace");
    const Layout L = layout_of(*s);
    const WsLayout w = make_ws(L, in->B);
    float* ws = static_cast<float*>(workspace);
    a.slab = ws + w.slab;
    a.ws_z = ws + w.z;
    a.ws_h = ws + w.h;
    a.ws_dh = ws + w.dh;
    a.ws_dl = ws + w.dl;
  }
  // (S > 1: the aux patch image is fetched in 16-byte LDS-DMA pieces whose source addresses are only dword aligned when the
  // aux row pitch is not a multiple of 4 floats — the reference pads a 1024-wide PAN to 1087; the buffer loads take that:
  // tests/test_gpu_parity.py::test_aux_scene_pitch_not_a_multiple_of_four)
  if (!patch_v2_supported(*s, mode, in->half)) return fail("%s", "no compiled kernel instance for this shape / mode (dmf_shape_supported names the rows)");
  return check(patch_v2_dispatch(*s, mode, a, static_cast<hipStream_t>(stream)), "patch kernel launch");
}

int64_t dmf_attn_workspace_bytes(const dmf_shape* s, int32_t B) {
  if (s == nullptr || B < 0) return -1;
  // two bf16 token maps + pooled z per patch + the bf16 weight copies of every head
  return (int64_t)B * (2 * 128 * 64 * 2 + 2 * s->F * 4) + (int64_t)attn_prep_bytes();
}

int32_t dmf_forward_attn(const dmf_shape* s, const dmf_input* in, const float* theta, const float* pool_w,
                         void* workspace, float* logits, int32_t* pred, void* stream) {
  if (in != nullptr && in->B == 0) return 0;
  if (s == nullptr || in == nullptr || theta == nullptr || pool_w == nullptr || workspace == nullptr || logits == nullptr)
    return fail("%s", "null argument");
  if (!s->attention) return fail("%s", "dmf_forward_attn needs shape->attention == 1");
  if (in->half) return fail("%s", "fp16 scenes (dmf_input.half): late-fusion network only");
  if (dmf_shape_supported(s)) return 1;
  if (!attn_shape_supported(*s)) return fail("%s", "no compiled attention instance for this shape (E = 96, heads = 3, F = 40)");
  if (in->B <= 0) return in->B == 0 ? 0 : fail("%s", "negative batch");
  const Layout L = layout_of(*s);
  const size_t B = (size_t)in->B;
  unsigned short* tokA = static_cast<unsigned short*>(workspace);
  unsigned short* tokB = tokA + B * 128 * 64;
  float* z = reinterpret_cast<float*>(tokB + B * 128 * 64);
  unsigned short* wprep = reinterpret_cast<unsigned short*>(z + B * 2 * s->F);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (check(attn_prep_launch(theta, L.off[12], L.off[13], L.off[14], L.off[15], wprep, st), "attention weight prep launch")) return 1;
  KArgs a{};
  a.in = *in; a.theta = theta; a.pool = pool_w; a.K = s->K;
  a.tokA = tokA; a.tokB = tokB; a.zout = z;
  if (check(conv_dispatch(*s, MODE_TOKENS, a, st), "token kernel launch")) return 1;
  AttnTrainArgs t{};
  t.tokA = tokA; t.tokB = tokB; t.zin = z; t.theta = theta; t.pool = pool_w; t.logits = logits; t.pred = pred; t.wprep = wprep;
  t.oWq = L.off[12]; t.oWk = L.off[13]; t.oWv = L.off[14]; t.oWo = L.off[15];
  t.oFc1w = L.off[8]; t.oFc1b = L.off[9]; t.oFc2w = L.off[10]; t.oFc2b = L.off[11];
  t.B = in->B; t.K = s->K;
  const int grid = in->B < 2 * MAX_BLOCKS ? in->B : 2 * MAX_BLOCKS;
  return check(attn_forward_dispatch(*s, t, grid, st), "attention kernel launch");
}

int64_t dmf_attn_train_workspace_bytes(const dmf_shape* s, int32_t B) {
  if (s == nullptr || B < 0) return -1;
  // two bf16 token maps + pooled z + the two dense gradient maps [B][F][P][RS] + the bf16 weight copies of every head
  return (int64_t)B * (2 * 128 * 64 * 2 + 2 * s->F * 4 + 2 * (int64_t)s->F * s->P * ((s->P + 3) & ~3) * 4) + (int64_t)attn_prep_bytes();
}

int32_t dmf_train_attn_fwd_bwd(const dmf_shape* s, const dmf_input* in, const float* theta, const float* pool_w,
                               const int32_t* labels, const float* dlogits, float loss_scale, float* logits, float* loss,
                               void* workspace, void* attn_workspace, int32_t* adam_step_dev, void* stream) {
  if (in != nullptr && in->B == 0) return 0;
  if (s == nullptr || in == nullptr || theta == nullptr || pool_w == nullptr || workspace == nullptr ||
      attn_workspace == nullptr || logits == nullptr)
    return fail("%s", "null argument");
  if ((labels == nullptr) == (dlogits == nullptr)) return fail("%s", "give exactly one of labels / dlogits");
  if (!s->attention) return fail("%s", "dmf_train_attn_fwd_bwd needs shape->attention == 1");
  if (in->half) return fail("%s", "fp16 scenes (dmf_input.half): late-fusion network only");
  if (dmf_shape_supported(s)) return 1;
  if (!attn_shape_supported(*s)) return fail("%s", "no compiled attention instance for this shape (E = 96, heads = 3, F = 40)");
  if (in->B <= 0) return in->B == 0 ? 0 : fail("%s", "negative batch");
  const Layout L = layout_of(*s);
  const WsLayout w = make_ws(L, in->B);
  float* ws = static_cast<float*>(workspace);
  const size_t B = (size_t)in->B;
  unsigned short* tokA = static_cast<unsigned short*>(attn_workspace);
  unsigned short* tokB = tokA + B * 128 * 64;
  float* z = reinterpret_cast<float*>(tokB + B * 128 * 64);
  float* dYa = z + B * 2 * s->F;
  const size_t map = (size_t)s->F * s->P * ((s->P + 3) & ~3);       // one patch's dense gradient map, rows padded to 16 B
  float* dYb = dYa + B * map;
  unsigned short* wprep = reinterpret_cast<unsigned short*>(dYb + B * map);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (check(attn_prep_launch(theta, L.off[12], L.off[13], L.off[14], L.off[15], wprep, st), "attention weight prep launch")) return 1;
  KArgs a{};
  a.in = *in; a.theta = theta; a.pool = pool_w; a.K = s->K;
  a.tokA = tokA; a.tokB = tokB; a.zout = z;
  if (check(conv_dispatch(*s, MODE_TOKENS, a, st), "token kernel launch")) return 1;
  AttnTrainArgs t{};
  t.tokA = tokA; t.tokB = tokB; t.zin = z; t.theta = theta; t.pool = pool_w;
  t.labels = labels; t.cursor = in->cursor; t.dlogits = dlogits; t.loss_scale = loss_scale;
  t.logits = logits; t.loss = loss;
  t.ws_z = ws + w.z; t.ws_h = ws + w.h; t.ws_dh = ws + w.dh; t.ws_dl = ws + w.dl;
  t.dYa = dYa; t.dYb = dYb; t.aslab = ws + w.aslab; t.wprep = wprep;
  t.oWq = L.off[12]; t.oWk = L.off[13]; t.oWv = L.off[14]; t.oWo = L.off[15];
  t.oFc1w = L.off[8]; t.oFc1b = L.off[9]; t.oFc2w = L.off[10]; t.oFc2b = L.off[11];
  t.B = in->B; t.K = s->K;
  const int grid = in->B < MAX_BLOCKS ? in->B : MAX_BLOCKS;          // one slab per workgroup, as the conv kernel
  if (check(attn_train_dispatch(*s, t, grid, st), "attention training kernel launch")) return 1;
  KArgs d{};
  d.in = *in; d.theta = theta; d.pool = pool_w; d.K = s->K;
  d.slab = ws + w.slab; d.dYa = dYa; d.dYb = dYb; d.adam_step = adam_step_dev;
  return check(conv_dispatch(*s, MODE_DENSE, d, st), "dense conv backward launch");
}

int32_t dmf_forward(const dmf_shape* s, const dmf_input* in, const float* theta, const float* pool_w,
                    float* logits, int32_t* pred, void* stream) {
  if (in != nullptr && in->B == 0) return 0;          // an empty batch is a no-op (its tensors have null data pointers)
  if (logits == nullptr) return fail("%s", "null logits");
  if (s != nullptr && s->attention) return fail("%s", "attention network: use dmf_forward_attn");
  return run_patch(s, in, MODE_FWD, theta, pool_w, nullptr, nullptr, 0.f, logits, nullptr, pred, nullptr, nullptr, stream);
}

int32_t dmf_forward_ce(const dmf_shape* s, const dmf_input* in, const float* theta, const float* pool_w, const int32_t* labels,
                       float* logits, float* loss, int32_t* pred, void* stream) {
  if (in != nullptr && in->B == 0) return 0;
  if (logits == nullptr || labels == nullptr || loss == nullptr) return fail("%s", "null logits/labels/loss");
  if (s == nullptr || s->attention || !(in != nullptr && in->half ? patch_v2_supported(*s, MODE_FWD, 1) : patch_v2_supported(*s, MODE_FWD)))
    return fail("%s", "dmf_forward_ce: no evaluation kernel with a fused cross-entropy for this shape (use dmf_forward)");
  return run_patch(s, in, MODE_FWD, theta, pool_w, labels, nullptr, 0.f, logits, loss, pred, nullptr, nullptr, stream);
}

int32_t dmf_train_fwd_bwd(const dmf_shape* s, const dmf_input* in, const float* theta, const float* pool_w,
                          const int32_t* labels, float loss_scale, float* logits, float* loss, void* workspace,
                          int32_t* adam_step_dev, void* stream) {
  if (in != nullptr && in->B == 0) return 0;
  if (labels == nullptr || logits == nullptr || loss == nullptr) return fail("%s", "null labels/logits/loss");
  return run_patch(s, in, MODE_TRAIN, theta, pool_w, labels, nullptr, loss_scale, logits, loss, nullptr, workspace, adam_step_dev, stream);
}

int32_t dmf_train_fwd_bwd_scaled(const dmf_shape* s, const dmf_input* in, const float* theta, const float* pool_w,
                                 const int32_t* labels, float loss_scale, const float* scaler_state, float* logits,
                                 float* loss, void* workspace, int32_t* adam_step_dev, void* stream) {
  if (in != nullptr && in->B == 0) return 0;
  if (labels == nullptr || logits == nullptr || loss == nullptr || scaler_state == nullptr)
    return fail("%s", "null labels/logits/loss/scaler_state");
  return run_patch(s, in, MODE_TRAIN, theta, pool_w, labels, nullptr, loss_scale, logits, loss, nullptr, workspace,
                   adam_step_dev, stream, scaler_state);
}

int32_t dmf_scaler_init(float* state, float init_scale, void* stream) {
  if (state == nullptr || !(init_scale > 0.f)) return fail("%s", "scaler: null state or non-positive scale");
  const float h[DMF_SCALER_FLOATS] = {init_scale, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (check(hipMemcpyAsync(state, h, sizeof(h), hipMemcpyHostToDevice, st), "scaler init")) return 1;
  return check(hipStreamSynchronize(st), "scaler init");       // (h is a stack buffer)
}

int32_t dmf_unscale_adam(float* theta, float* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                         float eps, float grad_scale, float* scaler_state, float growth_factor, float backoff_factor,
                         int32_t growth_interval, int32_t unscaled, int32_t* adam_step_dev, int32_t* cursor_dev, void* stream) {
  if (theta == nullptr || grad == nullptr || m == nullptr || v == nullptr || scaler_state == nullptr || adam_step_dev == nullptr)
    return fail("%s", "null argument (dmf_unscale_adam needs the device step count)");
  if (n <= 0 || growth_interval < 1 || !(growth_factor >= 1.f) || !(backoff_factor > 0.f && backoff_factor <= 1.f))
    return fail("%s", "unscale_adam: bad n / growth_interval / factors");
  const unsigned nblk = (unsigned)((n + 255) / 256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!unscaled) {
    hipLaunchKernelGGL(unscale_check_kernel, dim3(nblk), dim3(256), 0, st, grad, n, grad_scale, scaler_state);
    if (check(hipGetLastError(), "unscale launch")) return 1;
  }
  hipLaunchKernelGGL(scaled_adam_kernel, dim3(nblk), dim3(256), 0, st, theta, grad, m, v, n, lr, beta1, beta2, eps,
                     scaler_state, growth_factor, backoff_factor, growth_interval, adam_step_dev, cursor_dev);
  return check(hipGetLastError(), "scaled adam launch");
}

int32_t dmf_half_supported(const dmf_shape* s) {
  if (s == nullptr) return fail("%s", "null shape");
  if (s->attention || !patch_v2_supported(*s, MODE_TRAIN, 1))
    return fail("no fp16-scene kernel for this shape (instances C/C2/P/S/F/G:%s)", patch_v2_half_shape_list());
  return 0;
}

int32_t dmf_unit_supported(const dmf_shape* s) {
  if (s == nullptr) return fail("%s", "null shape");
  if (s->attention || !patch_v2_supported(*s, MODE_UNIT))
    return fail("no unit-gradient kernel for this shape (instances C/C2/P/S/F/G:%s)", patch_v2_shape_list());
  return 0;
}

int32_t dmf_forward_unit(const dmf_shape* s, const dmf_input* in, const float* theta, const float* pool_w, float* logits,
                         void* workspace, int32_t* adam_step_dev, void* stream) {
  if (s == nullptr || in == nullptr || theta == nullptr || pool_w == nullptr || logits == nullptr || workspace == nullptr)
    return fail("%s", "null argument");
  if (dmf_unit_supported(s)) return 1;
  if (in->half && dmf_half_supported(s)) return 1;
  if (in->B < 0) return fail("%s", "negative batch");
  if (in->B == 0) return 0;
  if (in->mode == 0 && (in->a == nullptr || in->b == nullptr)) return fail("%s", "mode 0 needs a and b");
  if (in->mode == 1 && (in->sceneA == nullptr || in->sceneB == nullptr || in->xy == nullptr || in->Wp <= 0 || in->WpB <= 0))
    return fail("%s", "mode 1 needs sceneA, sceneB, xy, Wp, WpB");
  if (in->mode != 0 && in->mode != 1) return fail("%s", "input mode must be 0 or 1");
  const Layout L = layout_of(*s);
  const WsLayout w = make_ws(L, in->B);
  float* ws = static_cast<float*>(workspace);
  KArgs a{};
  a.in = *in;
  a.theta = theta;
  a.pool = pool_w;
  a.logits = logits;
  a.adam_step = adam_step_dev;
  a.K = s->K;
  a.slab = ws + w.unit;          // MODE_UNIT: one row per patch
  a.ws_z = ws + w.z;
  a.ws_h = ws + w.h;
  a.ws_dh = ws + w.dh;
  a.ws_dl = ws + w.dl;
  return check(patch_v2_dispatch(*s, MODE_UNIT, a, static_cast<hipStream_t>(stream)), "patch kernel (v2, unit) launch");
}

int32_t dmf_backward_unit(const dmf_shape* s, int32_t B, const float* theta, const float* dlogits, void* workspace,
                          void* stream) {
  if (s == nullptr || theta == nullptr || dlogits == nullptr || workspace == nullptr) return fail("%s", "null argument");
  if (dmf_unit_supported(s)) return 1;
  if (B < 0) return fail("%s", "negative batch");
  if (B == 0) return 0;
  const Layout L = layout_of(*s);
  const WsLayout w = make_ws(L, B);
  float* ws = static_cast<float*>(workspace);
  UnitBwdArgs a{theta, dlogits, ws + w.unit, ws + w.h, ws + w.dh, ws + w.dl, ws + w.slab, B, s->K};
  return check(patch_v2_unit_backward(*s, a, static_cast<hipStream_t>(stream)), "unit backward launch");
}

int32_t dmf_backward_dlogits(const dmf_shape* s, const dmf_input* in, const float* theta, const float* pool_w,
                             const float* dlogits, void* workspace, void* stream) {
  if (in != nullptr && in->B == 0) return 0;
  if (dlogits == nullptr) return fail("%s", "null dlogits");
  return run_patch(s, in, MODE_BWD, theta, pool_w, nullptr, dlogits, 1.f, nullptr, nullptr, nullptr, workspace, nullptr, stream);
}

static int fill_xgmi(const dmf_xgmi_comm* c, XgmiDev& x) {
  if (c->world < 2 || c->world > XGMI_MAX || c->rank < 0 || c->rank >= c->world) return fail("%s", "bad xgmi world/rank");
  if (c->capacity <= 0) return fail("%s", "bad xgmi capacity");
  x.world = c->world; x.rank = c->rank;
  x.cap = (c->capacity + 255) / 256 * 256;
  x.timeout_ticks = (int64_t)(c->timeout_ms > 0 ? c->timeout_ms : 20000) * 100000;   // wall_clock64 runs at 100 MHz
  for (int r = 0; r < c->world; ++r) {
    if (c->data[r] == nullptr || c->flags[r] == nullptr) return fail("%s", "xgmi peer buffer missing");
    x.data[r] = static_cast<unsigned long long*>(c->data[r]);
    x.flags[r] = static_cast<int32_t*>(c->flags[r]);
  }
  return 0;
}

static int run_reduce(const dmf_shape* s, int32_t B, const void* workspace, float* grad, float* theta, float* m,
                      float* v, float lr, float b1, float b2, float eps, int32_t step, const int32_t* step_dev,
                      int32_t* cursor_dev, const float* loss, float* loss_hist, void* stream,
                      const dmf_xgmi_comm* comm = nullptr, float grad_scale = 1.f, float* scaler = nullptr) {
  if (s == nullptr || workspace == nullptr) return fail("%s", "null argument");
  if (B <= 0) return fail("%s", "batch must be positive");
  const Layout L = layout_of(*s);
  const WsLayout w = make_ws(L, B);
  const float* ws = static_cast<const float*>(workspace);
  ReduceArgs a{};
  a.slab = ws + w.slab; a.z = ws + w.z; a.h = ws + w.h; a.dh = ws + w.dh; a.dl = ws + w.dl;
  a.B = B; a.nblk = B < MAX_BLOCKS ? B : MAX_BLOCKS; a.SLAB = L.SLAB; a.NCONV = L.NCONV; a.F2 = L.F2; a.H = L.H; a.K = L.K;
  a.oFc1w = L.off[8]; a.oFc1b = L.off[9]; a.oFc2w = L.off[10]; a.oFc2b = L.off[11]; a.n = L.n_params;
  a.aslab = ws + w.aslab; a.nablk = a.nblk; a.ASLAB = 4 * L.E * L.F; a.oAttn = L.attention ? L.off[12] : L.n_params;
  a.grad = grad; a.theta = theta; a.m = m; a.v = v;
  a.lr = lr; a.b1 = b1; a.b2 = b2; a.eps = eps;
  if (theta != nullptr) {
    if (m == nullptr || v == nullptr || (step < 1 && step_dev == nullptr)) return fail("%s", "Adam needs m, v and step >= 1");
    if (step_dev == nullptr) {
      a.bc1 = (float)(1.0 - pow((double)b1, (double)step));
      a.bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, (double)step));
    }
  }
  a.step_dev = step_dev; a.cursor_dev = cursor_dev; a.loss = loss; a.loss_hist = loss_hist;
  a.grad_scale = grad_scale;
  a.scaler = scaler;
#ifdef DMF_STAMPS
  { const char* e = getenv("DMF_REDUCE_DBG"); a.dbg = e != nullptr ? atoi(e) : 0; }
#endif
  if (comm != nullptr) {
    if (step_dev == nullptr) return fail("%s", "the xgmi exchange needs adam_step_dev");
    if (comm->capacity < L.n_params) return fail("%s", "xgmi communicator smaller than the parameter vector");
    if (fill_xgmi(comm, a.x)) return 1;
    a.seq_bias = comm->seq_bias;
  }
  if (L.H % 8 != 0 || L.F2 % 8 != 0) return fail("%s", "grad_reduce: hidden width and 2 x gmf.width must be multiples of 8");
  const int t1n = L.F2 / 8, t2n = L.H / 8;
  const int nFc1 = (L.H / 8) * t1n, nFc2 = ((L.K + 7) / 8) * t2n, nConv = (L.NCONV + 15) / 16;
  const int nAttn = L.attention ? (a.ASLAB + 63) / 64 : 0;
  if (nFc1 > 0xffff || nFc2 > 0xffff || nConv > 0xffff || nAttn > 0x7fff) return fail("%s", "grad_reduce: too many blocks of one kind");
  const int grid = nFc1 + nFc2 + nConv + nAttn + 1;   // + the bookkeeping block
  hipLaunchKernelGGL(grad_reduce_kernel, dim3(grid), dim3(320), 0, static_cast<hipStream_t>(stream), a.slab, a.dh, a.z, a.dl, a.h,
                     nFc1 | (t1n << 16), nFc2 | (t2n << 16), nConv | (nAttn << 16), B, a);
  return check(hipGetLastError(), "grad_reduce launch");
}

int32_t dmf_grad_reduce(const dmf_shape* s, int32_t B, const void* workspace, float* grad, void* stream) {
  if (grad == nullptr) return fail("%s", "null grad");
  return run_reduce(s, B, workspace, grad, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, 0.f, 0, nullptr, nullptr, nullptr, nullptr, stream);
}

int32_t dmf_grad_reduce_scaled(const dmf_shape* s, int32_t B, const void* workspace, float* grad, float* scaler_state,
                               int32_t* cursor_dev, const float* loss, float* loss_hist, void* stream) {
  if (grad == nullptr || scaler_state == nullptr) return fail("%s", "null grad / scaler_state");
  return run_reduce(s, B, workspace, grad, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, 0.f, 0, nullptr, cursor_dev, loss, loss_hist,
                    stream, nullptr, 1.f, scaler_state);
}

int32_t dmf_grad_reduce_adam(const dmf_shape* s, int32_t B, const void* workspace, float* theta, float* m, float* v,
                             float* grad, float lr, float beta1, float beta2, float eps, int32_t step,
                             const int32_t* adam_step_dev, int32_t* cursor_dev, const float* loss, float* loss_hist,
                             void* stream) {
  if (theta == nullptr) return fail("%s", "null theta");
  return run_reduce(s, B, workspace, grad, theta, m, v, lr, beta1, beta2, eps, step, adam_step_dev, cursor_dev, loss,
                    loss_hist, stream);
}

int32_t dmf_train_plan_steps(const dmf_shape* s, const dmf_input* in, float* theta, const float* pool_w, const int32_t* labels,
                             float loss_scale, float* logits, float* loss, void* workspace, float* m, float* v, float lr,
                             float beta1, float beta2, float eps, int32_t* adam_step_dev, int32_t* cursor_dev, float* loss_hist,
                             int32_t n_steps, void* stream) {
  if (s == nullptr || in == nullptr || theta == nullptr || labels == nullptr || logits == nullptr || loss == nullptr ||
      adam_step_dev == nullptr || cursor_dev == nullptr)
    return fail("%s", "null argument (dmf_train_plan_steps needs the device step count and cursor)");
  if (in->mode != 1 || in->cursor != nullptr) return fail("%s", "dmf_train_plan_steps: gather mode, no plan cursor (the batches are consecutive)");
  if (s->attention) return fail("%s", "dmf_train_plan_steps: late-fusion network only");
  if (n_steps < 0 || in->B <= 0) return fail("%s", "dmf_train_plan_steps: negative step count or empty batch");
  dmf_input ik = *in;
  for (int32_t k = 0; k < n_steps; ++k) {
    ik.xy = in->xy + (size_t)2 * in->B * k;
    if (run_patch(s, &ik, MODE_TRAIN, theta, pool_w, labels + (size_t)in->B * k, nullptr, loss_scale, logits, loss, nullptr, workspace,
                  adam_step_dev, stream))
      return 1;
    if (run_reduce(s, in->B, workspace, nullptr, theta, m, v, lr, beta1, beta2, eps, 0, adam_step_dev, cursor_dev, loss, loss_hist, stream))
      return 1;
  }
  return 0;
}

int32_t dmf_grad_reduce_xgmi_adam(const dmf_shape* s, int32_t B, const void* workspace, float* theta, float* m,
                                  float* v, const dmf_xgmi_comm* comm, float lr, float beta1, float beta2, float eps,
                                  float grad_scale, const int32_t* adam_step_dev, int32_t* cursor_dev,
                                  const float* loss, float* loss_hist, void* stream) {
  if (theta == nullptr || comm == nullptr) return fail("%s", "null theta/comm");
  return run_reduce(s, B, workspace, nullptr, theta, m, v, lr, beta1, beta2, eps, 0, adam_step_dev, cursor_dev, loss,
                    loss_hist, stream, comm, grad_scale);
}

// ------------------------------------------------------------------------------ xgmi buffers + small all-reduce
int32_t dmf_xgmi_sizes(int64_t capacity, int32_t world, int64_t* data_bytes, int64_t* flag_bytes) {
  if (capacity <= 0 || world < 1 || world > XGMI_MAX || data_bytes == nullptr || flag_bytes == nullptr)
    return fail("%s", "bad xgmi_sizes argument");
  const int64_t cap = (capacity + 255) / 256 * 256;
  *data_bytes = 4 * cap * world * (int64_t)sizeof(unsigned long long);   // inbox: [region 2][parity 2][src world][cap] tagged words
  *flag_bytes = 16 * (int64_t)sizeof(int32_t);                           // the status word (+ spare)
  return 0;
}

int32_t dmf_xgmi_alloc(int64_t bytes, void** ptr) {
  if (bytes <= 0 || ptr == nullptr) return fail("%s", "bad xgmi_alloc argument");
  void* p = nullptr;
  if (check(hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached), "hipExtMallocWithFlags(uncached)")) return 1;
  if (check(hipMemset(p, 0, (size_t)bytes), "hipMemset") || check(hipDeviceSynchronize(), "hipDeviceSynchronize")) {
    (void)hipFree(p);
    return 1;
  }
  *ptr = p;
  return 0;
}

int32_t dmf_xgmi_free(void* ptr) { return ptr == nullptr ? 0 : check(hipFree(ptr), "hipFree"); }

int32_t dmf_xgmi_export(void* ptr, uint8_t handle[64]) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "HIP IPC handle size");
  if (ptr == nullptr || handle == nullptr) return fail("%s", "null argument");
  hipIpcMemHandle_t h;
  if (check(hipIpcGetMemHandle(&h, ptr), "hipIpcGetMemHandle")) return 1;
  memcpy(handle, &h, 64);
  return 0;
}

int32_t dmf_xgmi_open(const uint8_t handle[64], void** ptr) {
  if (ptr == nullptr || handle == nullptr) return fail("%s", "null argument");
  hipIpcMemHandle_t h;
  memcpy(&h, handle, 64);
  return check(hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle");
}

int32_t dmf_xgmi_close(void* ptr) { return ptr == nullptr ? 0 : check(hipIpcCloseMemHandle(ptr), "hipIpcCloseMemHandle"); }

int32_t dmf_xgmi_status(const dmf_xgmi_comm* c, int32_t* status) {
  if (c == nullptr || status == nullptr) return fail("%s", "null argument");
  XgmiDev x{};
  if (fill_xgmi(c, x)) return 1;
  if (check(hipDeviceSynchronize(), "hipDeviceSynchronize")) return 1;
  return check(hipMemcpy(status, x.flags[x.rank], sizeof(int32_t), hipMemcpyDeviceToHost),
               "hipMemcpy(status)");
}

__global__ __launch_bounds__(256) void xgmi_allreduce_kernel(const XgmiDev x, float* buf, int64_t n, int seq) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool valid = i < n;
  const float s = xgmi_exchange(x, 1, seq, i, valid, valid ? buf[i] : 0.f);
  if (valid) buf[i] = s;
}

int32_t dmf_xgmi_allreduce(const dmf_xgmi_comm* c, float* buf, int64_t n, int32_t seq, void* stream) {
  if (c == nullptr || buf == nullptr) return fail("%s", "null argument");
  XgmiDev x{};
  if (fill_xgmi(c, x)) return 1;
  if (n <= 0 || n > x.cap || seq < 1) return fail("%s", "xgmi_allreduce: n must be in [1, capacity], seq >= 1");
  hipLaunchKernelGGL(xgmi_allreduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, buf, n, seq);
  return check(hipGetLastError(), "xgmi_allreduce launch");
}

int32_t dmf_adam_step(float* theta, const float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                      float beta2, float eps, int32_t step, float grad_scale, const int32_t* adam_step_dev,
                      int32_t* cursor_dev, void* stream) {
  if (theta == nullptr || grad == nullptr || m == nullptr || v == nullptr) return fail("%s", "null argument");
  if (n <= 0 || (step < 1 && adam_step_dev == nullptr)) return fail("%s", "n and step must be positive");
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)(step < 1 ? 1 : step)));
  const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)(step < 1 ? 1 : step)));
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     theta, grad, m, v, n, lr, beta1, beta2, eps, bc1, bc2s, grad_scale, adam_step_dev, cursor_dev);
  return check(hipGetLastError(), "adam launch");
}

int32_t dmf_sgd_step(float* theta, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum,
                     int32_t step, float grad_scale, const int32_t* step_dev, int32_t* cursor_dev, void* stream) {
  if (theta == nullptr || grad == nullptr || (momentum != 0.f && momentum_buf == nullptr)) return fail("%s", "null argument");
  if (n <= 0 || (step < 1 && step_dev == nullptr)) return fail("%s", "n and step must be positive");
  hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), theta, grad,
                     momentum_buf, n, lr, momentum, grad_scale, step_dev, step, cursor_dev);
  return check(hipGetLastError(), "sgd launch");
}

int32_t dmf_rmsprop_step(float* theta, const float* grad, float* square_avg, int64_t n, float lr, float alpha, float eps,
                         float grad_scale, int32_t* cursor_dev, void* stream) {
  if (theta == nullptr || grad == nullptr || square_avg == nullptr) return fail("%s", "null argument");
  if (n <= 0) return fail("%s", "n must be positive");
  hipLaunchKernelGGL(rmsprop_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), theta,
                     grad, square_avg, n, lr, alpha, eps, grad_scale, cursor_dev);
  return check(hipGetLastError(), "rmsprop launch");
}

int32_t dmf_qua_loss_scaled(const float* logits, int32_t bs, int32_t K, const int32_t* labels, const int32_t* cursor,
                            const dmf_qua_params* prm, float grad_scale, const float* scaler_state, float* loss,
                            float* loss_hist, float* dlogits, void* stream) {
  if (logits == nullptr || labels == nullptr || prm == nullptr) return fail("%s", "null argument");
  if (bs <= 0 || K < 2 || K > KMAX) return fail("%s", "qua_loss: bs must be positive and 2 <= K <= DMF_KMAX");
  QuaArgs a{logits, bs, K, labels, cursor, prm->alpha, prm->beta, prm->gamma, prm->epsilon, prm->tao, grad_scale,
            loss, loss_hist, dlogits, scaler_state};
  return check(launch_qua_loss(a, static_cast<hipStream_t>(stream)), "qua_loss launch");
}

int32_t dmf_qua_loss(const float* logits, int32_t bs, int32_t K, const int32_t* labels, const int32_t* cursor,
                     const dmf_qua_params* prm, float grad_scale, float* loss, float* loss_hist, float* dlogits,
                     void* stream) {
  return dmf_qua_loss_scaled(logits, bs, K, labels, cursor, prm, grad_scale, nullptr, loss, loss_hist, dlogits, stream);
}

int32_t dmf_pair_argmax(const float* logits, int32_t bs, int32_t K, int32_t* pred, void* stream) {
  if (logits == nullptr || pred == nullptr) return fail("%s", "null argument");
  if (bs <= 0) return 0;
  if (K < 1) return fail("%s", "pair_argmax: K must be positive");
  return check(launch_pair_argmax(logits, bs, K, pred, static_cast<hipStream_t>(stream)), "pair_argmax launch");
}

int32_t dmf_band_mean(const float* x, int32_t layout, int64_t n_img, int64_t n_pix, int32_t C, float* out, void* stream) {
  if (x == nullptr || out == nullptr) return fail("%s", "null argument");
  if ((layout != 0 && layout != 1) || n_img <= 0 || n_pix <= 0 || C <= 0 || (layout == 0 && n_img != 1))
    return fail("%s", "bad band_mean geometry");
  return check(launch_band_mean(x, layout, n_img, n_pix, C, out, static_cast<hipStream_t>(stream)), "band_mean launch");
}

int32_t dmf_confusion_accum(const int32_t* pred, const int32_t* target, int32_t B, int32_t K, int64_t* matrix, void* stream) {
  if (pred == nullptr || target == nullptr || matrix == nullptr) return fail("%s", "null argument");
  if (B <= 0) return 0;
  hipLaunchKernelGGL(confusion_kernel, dim3((B + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), pred,
                     target, B, K, reinterpret_cast<unsigned long long*>(matrix));
  return check(hipGetLastError(), "confusion launch");
}

int32_t dmf_labelmap_write(const int32_t* pred, const int32_t* xy, int32_t B, int32_t W, int32_t* map, void* stream) {
  if (pred == nullptr || xy == nullptr || map == nullptr) return fail("%s", "null argument");
  if (B <= 0) return 0;
  hipLaunchKernelGGL(labelmap_kernel, dim3((B + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), pred, xy,
                     B, W, map);
  return check(hipGetLastError(), "labelmap launch");
}

int32_t dmf_pan2ms(const double* pan, int32_t pitch, int32_t H, int32_t W, double* out, void* stream) {
  if (pan == nullptr || out == nullptr) return fail("%s", "null argument");
  if (H <= 0 || W <= 0 || pitch < 4 * W) return fail("%s", "bad pan2ms geometry");
  const int64_t n = (int64_t)H * W * 4;
  hipLaunchKernelGGL(pan2ms_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     pan, pitch, H, W, out);
  return check(hipGetLastError(), "pan2ms launch");
}

#ifdef DMF_STAMPS
int32_t dmf_debug_set_reduce_stamps(void* p) { return check(hipMemcpyToSymbol(HIP_SYMBOL(dmf::g_rstamps), &p, sizeof(p)), "set_reduce_stamps"); }
int32_t dmf_debug_set_attn_stamps(void* p) { return check(dmf::set_attn_stamps(static_cast<unsigned long long*>(p)), "set_attn_stamps"); }
int32_t dmf_debug_set_v2_stamps(void* p) { return check(dmf::set_v2_stamps(static_cast<unsigned long long*>(p)), "set_v2_stamps"); }
#endif

}  // extern "C"

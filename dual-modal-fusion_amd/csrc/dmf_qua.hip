// dmf_qua.hip — batch-level kernels of the two-stage path's stage 2 (solver/tostagesolver.py:259-346):
//   qua_loss_kernel   : `qua_loss.forward` (train/loss_function.py:15-76) value + d loss / d logits for the four
//                       stacked streams [4*bs, K]
//   pair_argmax_kernel: `(output[:bs] + output[bs:2*bs]).softmax(-1).max(1)` (tostagesolver.py:337,366,378)
//   band_mean_kernel  : per-pixel mean over bands, the auxiliary input of the single-stream net (oracle/gmfnet_ref.py)
//
// The loss couples all samples of a batch through six batch-mean KL terms, the sign of two of their differences and
// a mean over all probabilities, so it cannot live inside the per-patch kernel.  It is tiny (4*bs*K logits), so one
// 1024-thread workgroup does three passes with fixed-order tree reductions in between:
//   pass 1  the six KL sums  A1=D(q>p) A2=D(r>p) A3=D(s>p) B1=D(p>q) B2=D(r>q) B3=D(s>q)  and the class term l4
//           D(x>y) = 1/bs * sum_ik y (log y - log(x + eps))            (F.kl_div(log(x+eps), y, 'batchmean'))
//   pass 2  l3 = mean_ik exp(-|A3/p|) + exp(-|B3/q|)  and  dl3/dA3, dl3/dB3
//   pass 3  d loss / d probabilities of the four streams, pushed through the four softmaxes
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dmf_kargs.h"

namespace dmf {

constexpr int QT = 1024;

// sum of N per-thread values over the 1024-thread block, result in every thread: butterfly inside each wavefront,
// then the 16 wave sums in fixed order through LDS (two barriers for all N values; the order never depends on timing)
template <int N>
__device__ __forceinline__ void block_sum(float (&v)[N], float* red) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    float x = v[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    if (lane == 0) red[wave * N + j] = x;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < N; ++j) {
    float x = 0.f;
#pragma unroll
    for (int w = 0; w < QT / 64; ++w) x += red[w * N + j];
    v[j] = x;
  }
  __syncthreads();
}

__device__ __forceinline__ float xlogx(float y) { return y > 0.f ? y * logf(y) : 0.f; }
__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// The batch is walked in tiles of TS samples whose 4*TS softmax rows live in LDS (sP[stream][sample][k]); within a
// tile all 1024 threads work: one softmax row each, then thread <-> (sample, quarter) for the sums and
// thread <-> (sample, stream) for the gradient rows.  A batch that fits one tile (bs <= TS, e.g. 256 x 12 logits)
// computes its rows once; larger batches recompute them per sweep.
// The kernel is ONE workgroup and bound by its transcendental calls (precise expf / logf, ~25 instructions each), so
// every logarithm is taken ONCE, by the thread that makes the row, and kept beside the probabilities:
//   sE[stream][sample][k] = log(prob + eps)   (all four streams; the KL terms and their gradients)
//   sX[stream][sample][k] = prob > 0 ? log(prob) : 0   (streams p, q: x log x and its derivative)
// — the same calls on the same arguments as the formulas' own order, so the results are bit-identical to taking them
// in place (32 K -> 17 K calls per sample at K = 12).
// KC > 0: K <= KC, every loop over the classes is unrolled to KC (guarded by k < K) — with run-time trip counts each
// iteration waits for its own LDS / global access, and one 1024-thread workgroup has nothing else to hide that behind.
// KC = 0: any K, run-time loops.
#define DMF_KLOOP(k, k0) _Pragma("unroll") for (int k = k0; k < (KC ? KC : K); ++k) if (KC == 0 || k < K)
template <int KC>
__global__ __launch_bounds__(QT) void qua_loss_kernel(const QuaArgs a, const int TS) {
  extern __shared__ __attribute__((aligned(16))) float sdyn[];
  float* red = sdyn;                    // [QT]
  float* sP = sdyn + QT;                // [4][TS][K]
  float* sE = sP + (size_t)4 * TS * a.K;   // [4][TS][K]  log(prob + eps)
  float* sX = sE + (size_t)4 * TS * a.K;   // [2][TS][K]  prob > 0 ? log(prob) : 0   (streams p, q)
  const int tid = threadIdx.x, bs = a.bs, K = a.K;
  const int cur = a.cursor != nullptr ? *a.cursor : 0;
  const int32_t* lab = a.labels + (size_t)cur * bs;
  const float inv_n = 1.f / (float)bs;
  const float e1 = expf(-1.f);
  const float lsum = 1.f + (float)(K - 1) * e1;          // softmax of a one-hot row (loss_function.py:52)
  const float l_hit = 1.f / lsum, l_miss = e1 / lsum;
  const float ll_hit = logf(l_hit), ll_miss = logf(l_miss);
  const int ntile = (bs + TS - 1) / TS;
  const bool one_tile = ntile == 1;

  // softmax rows of tile `t0..t0+nt` -> sP
  auto fill = [&](int t0, int nt) {
    for (int row = tid; row < 4 * nt; row += QT) {
      const int st = row / nt, i = row - st * nt;
      const float* x = a.logits + ((size_t)st * bs + t0 + i) * K;
      float mx = x[0];
      DMF_KLOOP(k, 1) mx = fmaxf(mx, x[k]);
      float sum = 0.f;
      float* o = sP + ((size_t)st * TS + i) * K;
      float* oe = sE + ((size_t)st * TS + i) * K;
      DMF_KLOOP(k, 0) { const float e = expf(x[k] - mx); o[k] = e; sum += e; }      // (parked, scaled below)
      const float inv = 1.f / sum;
      DMF_KLOOP(k, 0) {
        const float y = o[k] * inv;
        o[k] = y;
        oe[k] = logf(y + a.eps);
        if (st < 2) sX[((size_t)st * TS + i) * K + k] = y > 0.f ? logf(y) : 0.f;
      }
    }
    __syncthreads();
  };

  // ---- sweep 1: the six KL sums and the class term.  thread <-> (sample i, quarter j):
  //      j = 0: A1 = D(q>p), A2 = D(r>p)   j = 1: A3 = D(s>p), l4   j = 2: B1 = D(p>q), B2 = D(r>q)   j = 3: B3 = D(s>q)
  float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < ntile; ++t) {
    const int t0 = t * TS, nt = min(TS, bs - t0);
    if (t > 0) __syncthreads();
    fill(t0, nt);
    for (int w = tid; w < 4 * nt; w += QT) {
      const int j = w / nt, i = w - j * nt;
      const float* p = sP + (size_t)i * K;
      const float* q = p + (size_t)TS * K;
      const float* pe = sE + (size_t)i * K;            // log(. + eps) of the four streams
      const float* qe = pe + (size_t)TS * K;
      const float* re = qe + (size_t)TS * K;
      const float* se = re + (size_t)TS * K;
      const float* px = sX + (size_t)i * K;            // log p, log q (0 where the probability is 0)
      const float* qx = px + (size_t)TS * K;
      if (j == 0) {
        DMF_KLOOP(k, 0) { const float pl = p[k] * px[k]; acc[0] += pl - p[k] * qe[k]; acc[1] += pl - p[k] * re[k]; }
      } else if (j == 1) {
        float zmx = -INFINITY;
        DMF_KLOOP(k, 0) zmx = fmaxf(zmx, p[k] + q[k]);
        float zs = 0.f;
        DMF_KLOOP(k, 0) zs += expf(p[k] + q[k] - zmx);
        const float lzs = logf(zs);
        const int tg = lab[t0 + i];
        DMF_KLOOP(k, 0) {
          acc[2] += p[k] * px[k] - p[k] * se[k];
          const bool hit = k == tg;
          acc[6] += (hit ? l_hit : l_miss) * ((hit ? ll_hit : ll_miss) - ((p[k] + q[k] - zmx) - lzs));   // log softmax(p + q)
        }
      } else if (j == 2) {
        DMF_KLOOP(k, 0) { const float ql = q[k] * qx[k]; acc[3] += ql - q[k] * pe[k]; acc[4] += ql - q[k] * re[k]; }
      } else {
        DMF_KLOOP(k, 0) acc[5] += q[k] * qx[k] - q[k] * se[k];
      }
    }
  }
  block_sum(acc, red);
  const float A1 = acc[0] * inv_n, A2 = acc[1] * inv_n, A3 = acc[2] * inv_n;
  const float B1 = acc[3] * inv_n, B2 = acc[4] * inv_n, B3 = acc[5] * inv_n, l4 = acc[6] * inv_n;
  const float d1 = A3 - A2 + a.tao, d2 = B3 - B2 + a.tao;
  const float l12 = (A1 + A2 + fabsf(d1)) + (B1 + B2 + fabsf(d2));

  // ---- sweep 2: l3 = mean_ik exp(-|A3/p|) + exp(-|B3/q|) and dl3/dA3, dl3/dB3.  thread <-> (sample, p or q)
  float l3 = 0.f, GA = 0.f, GB = 0.f;
  const float inv_nk = 1.f / ((float)bs * (float)K);
  if (a.beta != 0.f) {
    float b3[3] = {0.f, 0.f, 0.f};
    for (int t = 0; t < ntile; ++t) {
      const int t0 = t * TS, nt = min(TS, bs - t0);
      if (!one_tile) { __syncthreads(); fill(t0, nt); }
      for (int w = tid; w < 2 * nt; w += QT) {
        const int j = w / nt, i = w - j * nt;
        const float* y = sP + ((size_t)j * TS + i) * K;
        const float c = j == 0 ? A3 : B3;
        float f = 0.f, gsum = 0.f;
        DMF_KLOOP(k, 0) {
          const float u = c / y[k], e = expf(-fabsf(u));
          f += e;
          gsum += (y[k] > 0.f) ? -sgn(u) * e / y[k] : 0.f;
        }
        b3[0] += f;
        b3[1 + j] += gsum;
      }
    }
    block_sum(b3, red);
    l3 = b3[0] * inv_nk; GA = b3[1] * inv_nk; GB = b3[2] * inv_nk;
  }
  const float loss = (a.alpha != 0.f ? a.alpha * l12 : 0.f) + a.beta * l3 + a.gamma * l4;
  if (tid == 0) {
    if (a.loss != nullptr) a.loss[0] = loss;
    if (a.loss_hist != nullptr) a.loss_hist[cur] = loss;
  }
  if (a.dlogits == nullptr) return;

  // ---- sweep 3: d loss / d probabilities of the four streams, through the four softmaxes.
  //      thread <-> (sample, stream): first the row's inner product sum_k dProb_k prob_k, then the gradient row
  const float s1 = sgn(d1), s2 = sgn(d2);
  const float al = a.alpha;
  const float cA1 = al, cA2 = al * (1.f - s1), cA3 = al * s1 + a.beta * GA;
  const float cB1 = al, cB2 = al * (1.f - s2), cB3 = al * s2 + a.beta * GB;
  for (int t = 0; t < ntile; ++t) {
    const int t0 = t * TS, nt = min(TS, bs - t0);
    if (!one_tile) { __syncthreads(); fill(t0, nt); }
    for (int w = tid; w < 4 * nt; w += QT) {
      const int st = w / nt, i = w - st * nt;
      const float* p = sP + (size_t)i * K;
      const float* q = p + (size_t)TS * K;
      const float* r = q + (size_t)TS * K;
      const float* s = r + (size_t)TS * K;
      const float* pe = sE + (size_t)i * K;
      const float* qe = pe + (size_t)TS * K;
      const float* re = qe + (size_t)TS * K;
      const float* se = re + (size_t)TS * K;
      const float* px = sX + (size_t)i * K;
      const float* qx = px + (size_t)TS * K;
      float zmx = 0.f, zinv = 0.f;
      const int tg = lab[t0 + i];
      if (st < 2) {
        zmx = -INFINITY;
        DMF_KLOOP(k, 0) zmx = fmaxf(zmx, p[k] + q[k]);
        float zs = 0.f;
        DMF_KLOOP(k, 0) zs += expf(p[k] + q[k] - zmx);
        zinv = 1.f / zs;
      }
      // dProb of this thread's row at class k
      auto dprob = [&](int k) -> float {
        const float pk = p[k], qk = q[k], rk = r[k], sk = s[k];
        if (st == 2) return -inv_n * (cA2 * pk + cB2 * qk) / (rk + a.eps);
        if (st == 3) return -inv_n * (cA3 * pk + cB3 * qk) / (sk + a.eps);
        const float l = (k == tg) ? l_hit : l_miss;
        const float dz = a.gamma * inv_n * (expf(pk + qk - zmx) * zinv - l);     // sum_k l == 1 up to rounding
        const float lr = re[k], ls = se[k];
        if (st == 0) {
          const float lg1 = (pk > 0.f) ? px[k] + 1.f : 0.f;
          float d = inv_n * (cA1 * (lg1 - qe[k]) + cA2 * (lg1 - lr) + cA3 * (lg1 - ls)) - inv_n * cB1 * qk / (pk + a.eps) + dz;
          if (a.beta != 0.f && pk > 0.f) { const float u = A3 / pk; d += a.beta * inv_nk * sgn(u) * expf(-fabsf(u)) * A3 / (pk * pk); }
          return d;
        }
        const float lg1 = (qk > 0.f) ? qx[k] + 1.f : 0.f;
        float d = inv_n * (cB1 * (lg1 - pe[k]) + cB2 * (lg1 - lr) + cB3 * (lg1 - ls)) - inv_n * cA1 * pk / (qk + a.eps) + dz;
        if (a.beta != 0.f && qk > 0.f) { const float u = B3 / qk; d += a.beta * inv_nk * sgn(u) * expf(-fabsf(u)) * B3 / (qk * qk); }
        return d;
      };
      const float* y = sP + ((size_t)st * TS + i) * K;
      float* g = a.dlogits + ((size_t)st * bs + t0 + i) * K;
      float ip = 0.f;
      float dk[KC ? KC : 1];                               // the row of dProb: registers (KC > 0) or parked in its output
      DMF_KLOOP(k, 0) { const float d = dprob(k); if (KC) dk[KC ? k : 0] = d; else g[k] = d; ip += d * y[k]; }
      const float gs = a.grad_scale * (a.scaler != nullptr ? a.scaler[0] : 1.f);
      DMF_KLOOP(k, 0) g[k] = gs * y[k] * ((KC ? dk[KC ? k : 0] : g[k]) - ip);
    }
  }
}
#undef DMF_KLOOP

// ---------------------------------------------------------------------------------------------------------------------------
// The same loss, element-parallel, for K <= 16 and bs <= 16 * QE_MAXG: 16 lanes per sample (lane k of the group holds class k
// of all four streams), 16 samples per 256-thread workgroup, ONE workgroup per CU's worth of samples instead of one
// workgroup for the whole batch.  The one-workgroup kernel above is bound by ~50 K precise expf / logf / divisions on ONE
// CU (33 us at bs = 256, K = 12); here a thread makes 5 exponentials and 6 logarithms.  The batch sums are grid-wide, so the
// three sweeps are three launches that hand their partial sums over in a small device buffer, summed in workgroup order
// (fixed order: the result does not depend on scheduling).  Each sweep recomputes the softmax rows from the logits.
// The buffer is a static of the library: calls on DIFFERENT streams must not overlap (the engines use one stream).
constexpr int QE_MAXG = 256;
__device__ float g_qua_part[QE_MAXG * 8];      // sweep 1: A1 A2 A3 B1 B2 B3 l4 (per workgroup)
__device__ float g_qua_part2[QE_MAXG * 4];     // sweep 2: l3 GA GB

#define DMF_DPP_OP(OP, v, CTRL) OP((v), __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (v)), __builtin_bit_cast(int, (v)), (CTRL), 0xF, 0xF, false)))
__device__ __forceinline__ float row16_sum(float v) {       // over the 16 lanes of a DPP row, result in all of them
#define ADDF(a, b) ((a) + (b))
  v = DMF_DPP_OP(ADDF, v, 0xB1); v = DMF_DPP_OP(ADDF, v, 0x4E); v = DMF_DPP_OP(ADDF, v, 0x141); v = DMF_DPP_OP(ADDF, v, 0x140);
#undef ADDF
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = DMF_DPP_OP(fmaxf, v, 0xB1); v = DMF_DPP_OP(fmaxf, v, 0x4E); v = DMF_DPP_OP(fmaxf, v, 0x141); v = DMF_DPP_OP(fmaxf, v, 0x140);
  return v;
}
#undef DMF_DPP_OP

struct QeRows { float y[4], le[4], lx[2]; bool on; int i, k; };     // probabilities, log(. + eps), log p / log q (0 at p = 0)

// softmax rows of sample i at class k for the four streams
__device__ __forceinline__ QeRows qe_rows(const QuaArgs& a) {
  QeRows r;
  const int t = blockIdx.x * 256 + threadIdx.x;
  r.i = t >> 4; r.k = t & 15;
  r.on = r.i < a.bs && r.k < a.K;
  const int ic = r.i < a.bs ? r.i : a.bs - 1;
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const float x = r.k < a.K ? a.logits[((size_t)st * a.bs + ic) * a.K + r.k] : -INFINITY;
    const float mx = row16_max(x);
    const float e = r.k < a.K ? expf(x - mx) : 0.f;
    const float inv = 1.f / row16_sum(e);
    r.y[st] = e * inv;
    r.le[st] = logf(r.y[st] + a.eps);
    if (st < 2) r.lx[st] = r.y[st] > 0.f ? logf(r.y[st]) : 0.f;
  }
  return r;
}

// sum of N per-thread values over the 256-thread workgroup -> out[0..N) (thread 0), fixed order
template <int N>
__device__ __forceinline__ void qe_block_sum(float (&v)[N], float* red, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    float x = v[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    if (lane == 0) red[wave * N + j] = x;
  }
  __syncthreads();
  if (threadIdx.x < N) out[threadIdx.x] = ((red[threadIdx.x] + red[N + threadIdx.x]) + red[2 * N + threadIdx.x]) + red[3 * N + threadIdx.x];
}

// workgroup-ordered sum of column j of a partial buffer (every thread gets it; G <= QE_MAXG)
__device__ __forceinline__ void qe_totals(const float* part, int stride, int n, int G, float* sh) {
  if ((int)threadIdx.x < n) {
    float s = 0.f;
    for (int g = 0; g < G; ++g) s += part[g * stride + threadIdx.x];
    sh[threadIdx.x] = s;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void qua_e1_kernel(const QuaArgs a) {
  __shared__ float red[4 * 7];
  const QeRows r = qe_rows(a);
  const int cur = a.cursor != nullptr ? *a.cursor : 0;
  const float e1 = expf(-1.f), lsum = 1.f + (float)(a.K - 1) * e1;
  const float l_hit = 1.f / lsum, l_miss = e1 / lsum, ll_hit = logf(l_hit), ll_miss = logf(l_miss);
  const float p = r.y[0], q = r.y[1];
  // log softmax(p + q) of the row
  const float z = r.k < a.K ? p + q : -INFINITY;
  const float zmx = row16_max(z);
  const float zs = row16_sum(r.k < a.K ? expf(z - zmx) : 0.f);
  const float lzs = logf(zs);
  float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (r.on) {
    const float pl = p * r.lx[0], ql = q * r.lx[1];
    acc[0] = pl - p * r.le[1]; acc[1] = pl - p * r.le[2]; acc[2] = pl - p * r.le[3];
    acc[3] = ql - q * r.le[0]; acc[4] = ql - q * r.le[2]; acc[5] = ql - q * r.le[3];
    const bool hit = r.k == a.labels[(size_t)cur * a.bs + r.i];
    acc[6] = (hit ? l_hit : l_miss) * ((hit ? ll_hit : ll_miss) - ((z - zmx) - lzs));
  }
  qe_block_sum(acc, red, g_qua_part + blockIdx.x * 8);
}

struct QeScalars { float A1, A2, A3, B1, B2, B3, l4, d1, d2, l12; };
__device__ __forceinline__ QeScalars qe_scalars(const QuaArgs& a, const float* t) {
  QeScalars s;
  const float inv_n = 1.f / (float)a.bs;
  s.A1 = t[0] * inv_n; s.A2 = t[1] * inv_n; s.A3 = t[2] * inv_n; s.B1 = t[3] * inv_n; s.B2 = t[4] * inv_n; s.B3 = t[5] * inv_n;
  s.l4 = t[6] * inv_n;
  s.d1 = s.A3 - s.A2 + a.tao; s.d2 = s.B3 - s.B2 + a.tao;
  s.l12 = (s.A1 + s.A2 + fabsf(s.d1)) + (s.B1 + s.B2 + fabsf(s.d2));
  return s;
}

__global__ __launch_bounds__(256) void qua_e2_kernel(const QuaArgs a, const int G) {
  __shared__ float red[4 * 3], tot[8];
  qe_totals(g_qua_part, 8, 7, G, tot);
  const QeScalars sc = qe_scalars(a, tot);
  const QeRows r = qe_rows(a);
  float b3[3] = {0.f, 0.f, 0.f};
  if (r.on) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float y = r.y[j], c = j == 0 ? sc.A3 : sc.B3;
      const float u = c / y, e = expf(-fabsf(u));
      b3[0] += e;
      b3[1 + j] = (y > 0.f) ? -sgn(u) * e / y : 0.f;
    }
  }
  qe_block_sum(b3, red, g_qua_part2 + blockIdx.x * 4);
}

__global__ __launch_bounds__(256) void qua_e3_kernel(const QuaArgs a, const int G) {      // (G = workgroups of sweeps 1 and 2)
  __shared__ float tot[8], tot2[4];
  qe_totals(g_qua_part, 8, 7, G, tot);
  if (a.beta != 0.f) qe_totals(g_qua_part2, 4, 3, G, tot2);
  const QeScalars sc = qe_scalars(a, tot);
  const int bs = a.bs, K = a.K;
  const int cur = a.cursor != nullptr ? *a.cursor : 0;
  const float inv_n = 1.f / (float)bs, inv_nk = 1.f / ((float)bs * (float)K);
  float l3 = 0.f, GA = 0.f, GB = 0.f;
  if (a.beta != 0.f) { l3 = tot2[0] * inv_nk; GA = tot2[1] * inv_nk; GB = tot2[2] * inv_nk; }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const float loss = (a.alpha != 0.f ? a.alpha * sc.l12 : 0.f) + a.beta * l3 + a.gamma * sc.l4;
    if (a.loss != nullptr) a.loss[0] = loss;
    if (a.loss_hist != nullptr) a.loss_hist[cur] = loss;
  }
  if (a.dlogits == nullptr) return;
  const QeRows r = qe_rows(a);
  const float e1 = expf(-1.f), lsum = 1.f + (float)(K - 1) * e1;
  const float l_hit = 1.f / lsum, l_miss = e1 / lsum;
  const float s1 = sgn(sc.d1), s2 = sgn(sc.d2), al = a.alpha;
  const float cA1 = al, cA2 = al * (1.f - s1), cA3 = al * s1 + a.beta * GA;
  const float cB1 = al, cB2 = al * (1.f - s2), cB3 = al * s2 + a.beta * GB;
  const float pk = r.y[0], qk = r.y[1], rk = r.y[2], sk = r.y[3];
  const float z = r.k < K ? pk + qk : -INFINITY;
  const float zmx = row16_max(z);
  const float ez = r.k < K ? expf(z - zmx) : 0.f;
  const float zinv = 1.f / row16_sum(ez);
  float d[4] = {0.f, 0.f, 0.f, 0.f};
  if (r.on) {
    const bool hit = r.k == a.labels[(size_t)cur * bs + r.i];
    const float dz = a.gamma * inv_n * (ez * zinv - (hit ? l_hit : l_miss));
    const float lr = r.le[2], ls = r.le[3];
    {
      const float lg1 = (pk > 0.f) ? r.lx[0] + 1.f : 0.f;
      float v = inv_n * (cA1 * (lg1 - r.le[1]) + cA2 * (lg1 - lr) + cA3 * (lg1 - ls)) - inv_n * cB1 * qk / (pk + a.eps) + dz;
      if (a.beta != 0.f && pk > 0.f) { const float u = sc.A3 / pk; v += a.beta * inv_nk * sgn(u) * expf(-fabsf(u)) * sc.A3 / (pk * pk); }
      d[0] = v;
    }
    {
      const float lg1 = (qk > 0.f) ? r.lx[1] + 1.f : 0.f;
      float v = inv_n * (cB1 * (lg1 - r.le[0]) + cB2 * (lg1 - lr) + cB3 * (lg1 - ls)) - inv_n * cA1 * pk / (qk + a.eps) + dz;
      if (a.beta != 0.f && qk > 0.f) { const float u = sc.B3 / qk; v += a.beta * inv_nk * sgn(u) * expf(-fabsf(u)) * sc.B3 / (qk * qk); }
      d[1] = v;
    }
    d[2] = -inv_n * (cA2 * pk + cB2 * qk) / (rk + a.eps);
    d[3] = -inv_n * (cA3 * pk + cB3 * qk) / (sk + a.eps);
  }
  const float gs = a.grad_scale * (a.scaler != nullptr ? a.scaler[0] : 1.f);
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const float ip = row16_sum(d[st] * r.y[st]);          // (lanes beyond K, samples beyond bs: d = 0)
    if (r.on) a.dlogits[((size_t)st * bs + r.i) * K + r.k] = gs * r.y[st] * (d[st] - ip);
  }
}


hipError_t launch_qua_loss(const QuaArgs& a, hipStream_t st) {
  if (a.K <= 16 && a.bs <= 16 * QE_MAXG) {          // element-parallel form: three launches over bs / 16 workgroups
    const int G = (a.bs + 15) / 16;
    hipLaunchKernelGGL(qua_e1_kernel, dim3(G), dim3(256), 0, st, a);
    if (a.beta != 0.f) hipLaunchKernelGGL(qua_e2_kernel, dim3(G), dim3(256), 0, st, a, G);
    hipLaunchKernelGGL(qua_e3_kernel, dim3(a.dlogits != nullptr ? G : 1), dim3(256), 0, st, a, G);
    return hipGetLastError();
  }
  // tile: as many samples as fit ~150 KB of rows (per sample 4 K probabilities + 4 K + 2 K logarithms), at most 256
  // (4 rows x 256 = one row per thread)
  int ts = (150 * 1024 / 4) / (10 * a.K);
  ts = ts > 256 ? 256 : (ts < 1 ? 1 : ts);
  if (ts > a.bs) ts = a.bs;
  const size_t bytes = (size_t)(QT + 10 * ts * a.K) * sizeof(float);
  static LdsAttrOnce once16, once0;
  hipError_t e = once16.set(reinterpret_cast<const void*>(&qua_loss_kernel<16>), 160 * 1024);
  if (e == hipSuccess) e = once0.set(reinterpret_cast<const void*>(&qua_loss_kernel<0>), 160 * 1024);
  if (e != hipSuccess) return e;
  if (a.K <= 16) hipLaunchKernelGGL(qua_loss_kernel<16>, dim3(1), dim3(QT), bytes, st, a, ts);
  else hipLaunchKernelGGL(qua_loss_kernel<0>, dim3(1), dim3(QT), bytes, st, a, ts);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void pair_argmax_kernel(const float* logits, int bs, int K, int32_t* pred) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= bs) return;
  const float* x = logits + (size_t)i * K;
  const float* y = x + (size_t)bs * K;
  float mx = x[0] + y[0];
  for (int k = 1; k < K; ++k) mx = fmaxf(mx, x[k] + y[k]);
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += expf(x[k] + y[k] - mx);
  const float inv = 1.f / s;
  float best = -1.f; int arg = 0;
  for (int k = 0; k < K; ++k) {
    const float v = expf(x[k] + y[k] - mx) * inv;        // first maximum wins, like torch.max(1)
    if (v > best) { best = v; arg = k; }
  }
  pred[i] = arg;
}

hipError_t launch_pair_argmax(const float* logits, int bs, int K, int32_t* pred, hipStream_t st) {
  hipLaunchKernelGGL(pair_argmax_kernel, dim3((bs + 255) / 256), dim3(256), 0, st, logits, bs, K, pred);
  return hipGetLastError();
}

// out[img][pix] = (((x0 + x1) + x2) + ...) / C — the summation order oracle/gmfnet_ref.py::band_mean spells out.
// layout 0: pixel-major scene [n_pix, C] (n_img = 1);  layout 1: band-major patches [n_img, C, n_pix]
__global__ __launch_bounds__(256) void band_mean_kernel(const float* x, int layout, int64_t n_img, int64_t n_pix, int C,
                                                        float* out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_img * n_pix) return;
  const int64_t img = i / n_pix, pix = i - img * n_pix;
  const float* px = layout == 0 ? x + pix * C : x + img * C * n_pix + pix;
  const int64_t bstride = layout == 0 ? 1 : n_pix;
  float s = px[0];
  for (int c = 1; c < C; ++c) s += px[(int64_t)c * bstride];
  out[i] = s / (float)C;
}

hipError_t launch_band_mean(const float* x, int layout, int64_t n_img, int64_t n_pix, int C, float* out, hipStream_t st) {
  hipLaunchKernelGGL(band_mean_kernel, dim3((unsigned)((n_img * n_pix + 255) / 256)), dim3(256), 0, st, x, layout, n_img,
                     n_pix, C, out);
  return hipGetLastError();
}

}  // namespace dmf

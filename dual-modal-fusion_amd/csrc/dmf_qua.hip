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
// thread <-> (sample, stream) for the gradient rows.  A batch that fits one tile (bs <= TS, e.g. 256 x 17 logits)
// computes its probabilities once; larger batches recompute them per sweep.
__global__ __launch_bounds__(QT) void qua_loss_kernel(const QuaArgs a, const int TS) {
  extern __shared__ __attribute__((aligned(16))) float sdyn[];
  float* red = sdyn;                    // [QT]
  float* sP = sdyn + QT;                // [4][TS][K]
  const int tid = threadIdx.x, bs = a.bs, K = a.K;
  const int cur = a.cursor != nullptr ? *a.cursor : 0;
  const int32_t* lab = a.labels + (size_t)cur * bs;
  const float inv_n = 1.f / (float)bs;
  const float e1 = expf(-1.f);
  const float lsum = 1.f + (float)(K - 1) * e1;          // softmax of a one-hot row (loss_function.py:52)
  const float l_hit = 1.f / lsum, l_miss = e1 / lsum;
  const int ntile = (bs + TS - 1) / TS;
  const bool one_tile = ntile == 1;

  // softmax rows of tile `t0..t0+nt` -> sP
  auto fill = [&](int t0, int nt) {
    for (int row = tid; row < 4 * nt; row += QT) {
      const int st = row / nt, i = row - st * nt;
      const float* x = a.logits + ((size_t)st * bs + t0 + i) * K;
      float mx = x[0];
      for (int k = 1; k < K; ++k) mx = fmaxf(mx, x[k]);
      float sum = 0.f;
      for (int k = 0; k < K; ++k) sum += expf(x[k] - mx);
      const float inv = 1.f / sum;
      float* o = sP + ((size_t)st * TS + i) * K;
      for (int k = 0; k < K; ++k) o[k] = expf(x[k] - mx) * inv;
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
      const float* r = q + (size_t)TS * K;
      const float* s = r + (size_t)TS * K;
      if (j == 0) {
        for (int k = 0; k < K; ++k) { const float pl = xlogx(p[k]); acc[0] += pl - p[k] * logf(q[k] + a.eps); acc[1] += pl - p[k] * logf(r[k] + a.eps); }
      } else if (j == 1) {
        float zmx = -INFINITY;
        for (int k = 0; k < K; ++k) zmx = fmaxf(zmx, p[k] + q[k]);
        float zs = 0.f;
        for (int k = 0; k < K; ++k) zs += expf(p[k] + q[k] - zmx);
        const float lzs = logf(zs);
        const int tg = lab[t0 + i];
        for (int k = 0; k < K; ++k) {
          acc[2] += xlogx(p[k]) - p[k] * logf(s[k] + a.eps);
          const float l = (k == tg) ? l_hit : l_miss;
          acc[6] += l * (logf(l) - ((p[k] + q[k] - zmx) - lzs));        // log softmax(p + q)
        }
      } else if (j == 2) {
        for (int k = 0; k < K; ++k) { const float ql = xlogx(q[k]); acc[3] += ql - q[k] * logf(p[k] + a.eps); acc[4] += ql - q[k] * logf(r[k] + a.eps); }
      } else {
        for (int k = 0; k < K; ++k) acc[5] += xlogx(q[k]) - q[k] * logf(s[k] + a.eps);
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
        for (int k = 0; k < K; ++k) {
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
      float zmx = 0.f, zinv = 0.f;
      const int tg = lab[t0 + i];
      if (st < 2) {
        zmx = -INFINITY;
        for (int k = 0; k < K; ++k) zmx = fmaxf(zmx, p[k] + q[k]);
        float zs = 0.f;
        for (int k = 0; k < K; ++k) zs += expf(p[k] + q[k] - zmx);
        zinv = 1.f / zs;
      }
      // dProb of this thread's row at class k
      auto dprob = [&](int k) -> float {
        const float pk = p[k], qk = q[k], rk = r[k], sk = s[k];
        if (st == 2) return -inv_n * (cA2 * pk + cB2 * qk) / (rk + a.eps);
        if (st == 3) return -inv_n * (cA3 * pk + cB3 * qk) / (sk + a.eps);
        const float l = (k == tg) ? l_hit : l_miss;
        const float dz = a.gamma * inv_n * (expf(pk + qk - zmx) * zinv - l);     // sum_k l == 1 up to rounding
        const float lr = logf(rk + a.eps), ls = logf(sk + a.eps);
        if (st == 0) {
          const float lg1 = (pk > 0.f) ? logf(pk) + 1.f : 0.f;
          float d = inv_n * (cA1 * (lg1 - logf(qk + a.eps)) + cA2 * (lg1 - lr) + cA3 * (lg1 - ls)) - inv_n * cB1 * qk / (pk + a.eps) + dz;
          if (a.beta != 0.f && pk > 0.f) { const float u = A3 / pk; d += a.beta * inv_nk * sgn(u) * expf(-fabsf(u)) * A3 / (pk * pk); }
          return d;
        }
        const float lg1 = (qk > 0.f) ? logf(qk) + 1.f : 0.f;
        float d = inv_n * (cB1 * (lg1 - logf(pk + a.eps)) + cB2 * (lg1 - lr) + cB3 * (lg1 - ls)) - inv_n * cA1 * pk / (qk + a.eps) + dz;
        if (a.beta != 0.f && qk > 0.f) { const float u = B3 / qk; d += a.beta * inv_nk * sgn(u) * expf(-fabsf(u)) * B3 / (qk * qk); }
        return d;
      };
      const float* y = sP + ((size_t)st * TS + i) * K;
      float* g = a.dlogits + ((size_t)st * bs + t0 + i) * K;
      float ip = 0.f;
      for (int k = 0; k < K; ++k) { const float d = dprob(k); g[k] = d; ip += d * y[k]; }     // (row parked in its output)
      const float gs = a.grad_scale * (a.scaler != nullptr ? a.scaler[0] : 1.f);
      for (int k = 0; k < K; ++k) g[k] = gs * y[k] * (g[k] - ip);
    }
  }
}

hipError_t launch_qua_loss(const QuaArgs& a, hipStream_t st) {
  // tile: as many samples as fit ~128 KB of probabilities, at most 256 (4 rows x 256 = one row per thread)
  int ts = (128 * 1024 / 4) / (4 * a.K);
  ts = ts > 256 ? 256 : (ts < 1 ? 1 : ts);
  if (ts > a.bs) ts = a.bs;
  const size_t bytes = (size_t)(QT + 4 * ts * a.K) * sizeof(float);
  static bool done = false;
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qua_loss_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return e;
    done = true;
  }
  hipLaunchKernelGGL(qua_loss_kernel, dim3(1), dim3(QT), bytes, st, a, ts);
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

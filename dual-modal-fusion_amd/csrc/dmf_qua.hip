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

namespace dmf {

struct QuaArgs {
  const float* logits; int bs, K;
  const int32_t* labels; const int32_t* cursor;
  float alpha, beta, gamma, eps, tao, grad_scale;
  float* loss; float* loss_hist; float* dlogits;
};

constexpr int QT = 1024;

template <int N>
__device__ __forceinline__ void block_sum(float (&v)[N], float* red) {   // fixed tree; result in every thread
  const int tid = threadIdx.x;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    red[tid] = v[j];
    __syncthreads();
    for (int w = QT / 2; w > 0; w >>= 1) {
      if (tid < w) red[tid] += red[tid + w];
      __syncthreads();
    }
    v[j] = red[0];
    __syncthreads();
  }
}

struct Row { float mx, inv; };   // softmax row statistics: p_k = exp(x_k - mx) * inv

__device__ __forceinline__ Row row_stats(const float* x, int K) {
  float mx = x[0];
  for (int k = 1; k < K; ++k) mx = fmaxf(mx, x[k]);
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += expf(x[k] - mx);
  return Row{mx, 1.f / s};
}
__device__ __forceinline__ float prob(const float* x, int k, const Row& r) { return expf(x[k] - r.mx) * r.inv; }
__device__ __forceinline__ float xlogx(float y) { return y > 0.f ? y * logf(y) : 0.f; }
__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

__global__ __launch_bounds__(QT) void qua_loss_kernel(const QuaArgs a) {
  __shared__ float red[QT];
  const int tid = threadIdx.x, bs = a.bs, K = a.K;
  const int cur = a.cursor != nullptr ? *a.cursor : 0;
  const int32_t* lab = a.labels + (size_t)cur * bs;
  const float inv_n = 1.f / (float)bs;
  const float e1 = expf(-1.f);
  const float lsum = 1.f + (float)(K - 1) * e1;          // softmax of a one-hot row (loss_function.py:52)
  const float l_hit = 1.f / lsum, l_miss = e1 / lsum;

  // ---- pass 1
  float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < bs; i += QT) {
    const float* xp = a.logits + (size_t)i * K;
    const float* xq = xp + (size_t)bs * K;
    const float* xr = xq + (size_t)bs * K;
    const float* xs = xr + (size_t)bs * K;
    const Row rp = row_stats(xp, K), rq = row_stats(xq, K), rr = row_stats(xr, K), rs = row_stats(xs, K);
    float zmx = -INFINITY;
    for (int k = 0; k < K; ++k) zmx = fmaxf(zmx, prob(xp, k, rp) + prob(xq, k, rq));
    float zs = 0.f;
    for (int k = 0; k < K; ++k) zs += expf(prob(xp, k, rp) + prob(xq, k, rq) - zmx);
    const float lzs = logf(zs);
    const int t = lab[i];
    for (int k = 0; k < K; ++k) {
      const float p = prob(xp, k, rp), q = prob(xq, k, rq), r = prob(xr, k, rr), s = prob(xs, k, rs);
      const float lq = logf(q + a.eps), lp = logf(p + a.eps), lr = logf(r + a.eps), ls = logf(s + a.eps);
      const float plp = xlogx(p), qlq = xlogx(q);
      acc[0] += plp - p * lq;  acc[1] += plp - p * lr;  acc[2] += plp - p * ls;
      acc[3] += qlq - q * lp;  acc[4] += qlq - q * lr;  acc[5] += qlq - q * ls;
      const float l = (k == t) ? l_hit : l_miss;
      const float logm = (p + q - zmx) - lzs;              // log softmax(p + q)
      acc[6] += l * (logf(l) - logm);
    }
  }
  block_sum(acc, red);
  const float A1 = acc[0] * inv_n, A2 = acc[1] * inv_n, A3 = acc[2] * inv_n;
  const float B1 = acc[3] * inv_n, B2 = acc[4] * inv_n, B3 = acc[5] * inv_n, l4 = acc[6] * inv_n;
  const float d1 = A3 - A2 + a.tao, d2 = B3 - B2 + a.tao;
  const float l12 = (A1 + A2 + fabsf(d1)) + (B1 + B2 + fabsf(d2));

  // ---- pass 2
  float l3 = 0.f, GA = 0.f, GB = 0.f;
  if (a.beta != 0.f) {
    float b3[3] = {0.f, 0.f, 0.f};
    for (int i = tid; i < bs; i += QT) {
      const float* xp = a.logits + (size_t)i * K;
      const float* xq = xp + (size_t)bs * K;
      const Row rp = row_stats(xp, K), rq = row_stats(xq, K);
      for (int k = 0; k < K; ++k) {
        const float p = prob(xp, k, rp), q = prob(xq, k, rq);
        const float ua = A3 / p, ub = B3 / q;
        const float fa = expf(-fabsf(ua)), fb = expf(-fabsf(ub));
        b3[0] += fa + fb;
        b3[1] += (p > 0.f) ? -sgn(ua) * fa / p : 0.f;
        b3[2] += (q > 0.f) ? -sgn(ub) * fb / q : 0.f;
      }
    }
    block_sum(b3, red);
    const float inv_nk = 1.f / ((float)bs * (float)K);
    l3 = b3[0] * inv_nk; GA = b3[1] * inv_nk; GB = b3[2] * inv_nk;
  }
  const float loss = (a.alpha != 0.f ? a.alpha * l12 : 0.f) + a.beta * l3 + a.gamma * l4;
  if (tid == 0) {
    if (a.loss != nullptr) a.loss[0] = loss;
    if (a.loss_hist != nullptr) a.loss_hist[cur] = loss;
  }
  if (a.dlogits == nullptr) return;

  // ---- pass 3
  const float s1 = sgn(d1), s2 = sgn(d2);
  const float al = a.alpha;
  const float cA1 = al, cA2 = al * (1.f - s1), cA3 = al * s1 + a.beta * GA;
  const float cB1 = al, cB2 = al * (1.f - s2), cB3 = al * s2 + a.beta * GB;
  const float inv_nk = 1.f / ((float)bs * (float)K);
  for (int i = tid; i < bs; i += QT) {
    const float* xp = a.logits + (size_t)i * K;
    const float* xq = xp + (size_t)bs * K;
    const float* xr = xq + (size_t)bs * K;
    const float* xs = xr + (size_t)bs * K;
    float* gp = a.dlogits + (size_t)i * K;
    float* gq = gp + (size_t)bs * K;
    float* gr = gq + (size_t)bs * K;
    float* gs = gr + (size_t)bs * K;
    const Row rp = row_stats(xp, K), rq = row_stats(xq, K), rr = row_stats(xr, K), rs = row_stats(xs, K);
    float zmx = -INFINITY;
    for (int k = 0; k < K; ++k) zmx = fmaxf(zmx, prob(xp, k, rp) + prob(xq, k, rq));
    float zs = 0.f;
    for (int k = 0; k < K; ++k) zs += expf(prob(xp, k, rp) + prob(xq, k, rq) - zmx);
    const float zinv = 1.f / zs;
    const int t = lab[i];
    // two sweeps over k: first the inner products  sum_k dProb_k * prob_k  of the four rows, then the gradients
    float ip = 0.f, iq = 0.f, ir = 0.f, is = 0.f;
#pragma unroll 1
    for (int sweep = 0; sweep < 2; ++sweep) {
      for (int k = 0; k < K; ++k) {
        const float p = prob(xp, k, rp), q = prob(xq, k, rq), r = prob(xr, k, rr), s = prob(xs, k, rs);
        const float l = (k == t) ? l_hit : l_miss;
        const float m = expf(p + q - zmx) * zinv;
        const float dz = a.gamma * inv_n * (m - l);          // sum_k l == 1 up to rounding
        const float logp1 = (p > 0.f) ? logf(p) + 1.f : 0.f, logq1 = (q > 0.f) ? logf(q) + 1.f : 0.f;
        float dP = inv_n * (cA1 * (logp1 - logf(q + a.eps)) + cA2 * (logp1 - logf(r + a.eps)) + cA3 * (logp1 - logf(s + a.eps)))
                   - inv_n * cB1 * q / (p + a.eps) + dz;
        float dQ = inv_n * (cB1 * (logq1 - logf(p + a.eps)) + cB2 * (logq1 - logf(r + a.eps)) + cB3 * (logq1 - logf(s + a.eps)))
                   - inv_n * cA1 * p / (q + a.eps) + dz;
        if (a.beta != 0.f) {
          const float ua = A3 / p, ub = B3 / q;
          if (p > 0.f) dP += a.beta * inv_nk * sgn(ua) * expf(-fabsf(ua)) * A3 / (p * p);
          if (q > 0.f) dQ += a.beta * inv_nk * sgn(ub) * expf(-fabsf(ub)) * B3 / (q * q);
        }
        const float dR = -inv_n * (cA2 * p + cB2 * q) / (r + a.eps);
        const float dS = -inv_n * (cA3 * p + cB3 * q) / (s + a.eps);
        if (sweep == 0) {
          ip += dP * p; iq += dQ * q; ir += dR * r; is += dS * s;
        } else {
          gp[k] = a.grad_scale * p * (dP - ip);
          gq[k] = a.grad_scale * q * (dQ - iq);
          gr[k] = a.grad_scale * r * (dR - ir);
          gs[k] = a.grad_scale * s * (dS - is);
        }
      }
    }
  }
}

hipError_t launch_qua_loss(const QuaArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(qua_loss_kernel, dim3(1), dim3(QT), 0, st, a);
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

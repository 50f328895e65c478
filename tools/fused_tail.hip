// What would folding the gradient reduce into the patch kernel cost?  256 workgroups (one per CU, 576 threads like the patch
// kernel): each writes its 8-KB slab row with sc1 stores, waits for them, adds to a counter, polls until all rows are there,
// then reads 32 parameters x 256 rows with sc1 loads (the hand-off form of MI355X_MICROARCH.md "Valid forms", third row of its
// table) and writes 32 sums.  Prints the median in-kernel time of the tail (realtime clock, 100 MHz) and checks the sums.
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/fused_tail tools/fused_tail.hip && tools/bin/fused_tail
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

constexpr int SLAB = 2016, NT = 576, GRID = 256;

__global__ __launch_bounds__(NT) void tail_kernel(float* slab, float* out, unsigned* counter, unsigned long long* times, int spin_ns) {
  const int tid = threadIdx.x, g = blockIdx.x;
  // pretend patch work of slightly different length per workgroup
  unsigned long long t0;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  if (tid < SLAB / 4) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 v = {(float)g, (float)(tid & 7), 1.f, 0.5f};
    float* p = slab + (size_t)g * SLAB + 4 * tid;
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ unsigned target;
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned tg = (old / GRID + 1) * GRID;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < tg) __builtin_amdgcn_s_sleep(2);
    target = tg;
  }
  __syncthreads();
  unsigned long long t1;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  // 32 parameters of this workgroup: thread (p = tid & 31, chunk = tid >> 5 of 16) sums 16 rows
  __shared__ float part[16][33];
  if (tid < 512) {
    const int p = tid & 31, ch = tid >> 5;
    const int col = g * 32 + p;
    float acc = 0.f;
    if (col < SLAB) {
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = __hip_atomic_load(slab + (size_t)(ch * 16 + i) * SLAB + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc += v[i];
    }
    part[ch][p] = acc;
  }
  __syncthreads();
  if (tid < 32 && g * 32 + tid < SLAB) {
    float s = 0.f;
    for (int c = 0; c < 16; ++c) s += part[c][tid];
    out[g * 32 + tid] = s;
  }
  unsigned long long t2;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2));
  if (tid == 0) { times[3 * g] = t0; times[3 * g + 1] = t1; times[3 * g + 2] = t2; }
}

int main() {
  float *slab, *out; unsigned* counter; unsigned long long* times;
  hipMalloc(&slab, (size_t)GRID * SLAB * 4); hipMalloc(&out, SLAB * 4); hipMalloc(&counter, 4); hipMalloc(&times, GRID * 3 * 8);
  hipMemset(counter, 0, 4);
  std::vector<unsigned long long> h(GRID * 3);
  std::vector<float> ho(SLAB);
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL(tail_kernel, dim3(GRID), dim3(NT), 0, 0, slab, out, counter, times, 0);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), times, GRID * 3 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(ho.data(), out, SLAB * 4, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull, tmax = 0, e1 = 0;
    std::vector<double> sync, red;
    for (int g = 0; g < GRID; ++g) {
      tmin = std::min(tmin, h[3 * g]); tmax = std::max(tmax, h[3 * g + 2]); e1 = std::max(e1, h[3 * g + 1]);
      sync.push_back((h[3 * g + 1] - h[3 * g]) * 0.01); red.push_back((h[3 * g + 2] - h[3 * g + 1]) * 0.01);
    }
    std::sort(sync.begin(), sync.end()); std::sort(red.begin(), red.end());
    int bad = 0;
    for (int c = 0; c < SLAB; ++c) {
      const int e = c & 3, t = c >> 2;
      const float want = e == 0 ? 255.f * 256.f / 2.f : e == 1 ? 256.f * (t & 7) : e == 2 ? 256.f : 128.f;
      bad += ho[c] != want;
    }
    printf("rep %d: store+wait+arrive+poll median %.2f us (max %.2f), reduce 32 KB median %.2f us (max %.2f), whole %.2f us, wrong sums %d\n",
           rep, sync[GRID / 2], sync.back(), red[GRID / 2], red.back(), (tmax - tmin) * 0.01, bad);
  }
  return 0;
}

// What does one resident wavefront per SIMD cost the field multiplier, and does instruction-level parallelism inside
// a wavefront (two or three independent products interleaved by the compiler) buy it back?  Grid = waves_per_simd x
// (CUs x 4) single-wavefront workgroups; every lane runs `iters` dependent rounds of 1, 2 or 3 independent products.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../bellman_amd/csrc/ff.cuh"
using namespace bh;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int CHAINS, bool CALL>
__global__ __launch_bounds__(64) void k_fp(fp_t *out, fp_t seed, int iters) {
  fp_t a[CHAINS], b[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; c++) { a[c] = seed; b[c] = seed; a[c].l[0] += threadIdx.x + c; b[c].l[1] += (blockIdx.x & 0xff) + 7 * c; }
  for (int it = 0; it < iters; it++) {
    fp_t r[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) {
      if (CALL) r[c] = fp_mul_call(a[c], b[c]); else fe_mul<FpParams, false>(r[c], a[c], b[c]);
    }
#pragma unroll
    for (int c = 0; c < CHAINS; c++) { b[c] = a[c]; a[c] = r[c]; }
  }
  fp_t s = a[0];
#pragma unroll
  for (int c = 1; c < CHAINS; c++) for (int i = 0; i < 12; i++) s.l[i] ^= a[c].l[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// Fp2 product: Karatsuba on three out-of-line products (the curve code's form) vs the three products inline in
// one body (the scheduler may interleave them)
template <bool INLINE3>
__global__ __launch_bounds__(64) void k_fp2(fp2_t *out, fp_t seed, int iters) {
  fp2_t a, b;
  a.c0 = seed; a.c1 = seed; b.c0 = seed; b.c1 = seed;
  a.c0.l[0] += threadIdx.x; a.c1.l[1] += 3; b.c0.l[2] += blockIdx.x & 0xff; b.c1.l[3] += 5;
  for (int it = 0; it < iters; it++) {
    fp2_t r;
    if (INLINE3) Fp2Ops::mul_tail(r, a, b); else Fp2Ops::mul(r, a, b);
    b = a; a = r;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
template <class F> static float timeit(F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
  const int simds = p.multiProcessorCount * 4;
  void *buf; CHK(hipMalloc(&buf, (size_t)simds * 8 * 64 * 96));
  fp_t sp; for (int i = 0; i < 12; i++) sp.l[i] = 0x01234567u * (i + 1); sp.l[11] &= 0x0fffffff;
  const int iters = 2048;
  printf("%d SIMDs; Fp products per lane per launch: chains x %d; G products/s = lanes x chains x iters / t\n", simds, iters);
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int blocks = simds * wps;
    const double lanes = (double)blocks * 64;
    float ms;
#define RUN(NAME, K, MULS) ms = timeit([&] { hipLaunchKernelGGL(K, dim3(blocks), dim3(64), 0, 0, (decltype(K##_out))buf, sp, iters); }); \
    printf("waves/SIMD %d  %-34s %7.3f ms  %7.2f G Fp products/s\n", wps, NAME, ms, lanes * (MULS) * iters / ms / 1e6);
    fp_t *k1_out = nullptr; (void)k1_out;
    ms = timeit([&] { hipLaunchKernelGGL((k_fp<1, true>), dim3(blocks), dim3(64), 0, 0, (fp_t *)buf, sp, iters); });
    printf("waves/SIMD %d  %-34s %7.3f ms  %7.2f G Fp products/s\n", wps, "1 chain, out-of-line product", ms, lanes * 1 * iters / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_fp<1, false>), dim3(blocks), dim3(64), 0, 0, (fp_t *)buf, sp, iters); });
    printf("waves/SIMD %d  %-34s %7.3f ms  %7.2f G Fp products/s\n", wps, "1 chain, inline", ms, lanes * 1 * iters / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_fp<2, false>), dim3(blocks), dim3(64), 0, 0, (fp_t *)buf, sp, iters); });
    printf("waves/SIMD %d  %-34s %7.3f ms  %7.2f G Fp products/s\n", wps, "2 chains, inline", ms, lanes * 2 * iters / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_fp<3, false>), dim3(blocks), dim3(64), 0, 0, (fp_t *)buf, sp, iters); });
    printf("waves/SIMD %d  %-34s %7.3f ms  %7.2f G Fp products/s\n", wps, "3 chains, inline", ms, lanes * 3 * iters / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_fp2<false>), dim3(blocks), dim3(64), 0, 0, (fp2_t *)buf, sp, iters); });
    printf("waves/SIMD %d  %-34s %7.3f ms  %7.2f G Fp products/s\n", wps, "Fp2 product, 3 calls", ms, lanes * 3 * iters / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_fp2<true>), dim3(blocks), dim3(64), 0, 0, (fp2_t *)buf, sp, iters); });
    printf("waves/SIMD %d  %-34s %7.3f ms  %7.2f G Fp products/s\n", wps, "Fp2 product, 3 inline", ms, lanes * 3 * iters / ms / 1e6);
  }
  return 0;
}

// Integer / fp64 instruction throughput probe for gfx950 (decides the limb representation of
// the field multiplier).  Each kernel runs ITER iterations of 8 independent dependency chains
// per lane; 256 CUs x 8 blocks x 256 threads.  Prints G-ops/s per instruction kind and the
// measured throughput of the library's 12-limb / 8-limb Montgomery products.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../bellman_amd/csrc/ff.cuh"
using namespace bh;
#define ITER 4096
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_mad_u64_u32(uint64_t *out, uint32_t a, uint32_t b) {
  uint64_t acc[8];
  for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i;
  uint32_t x = a + threadIdx.x, y = b;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = (uint64_t)x * (uint32_t)(y + i) + acc[i];
    x += (uint32_t)acc[0];
  }
  uint64_t s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mul_lo(uint32_t *out, uint32_t a) {
  uint32_t acc[8];
  for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i + a;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = acc[i] * (a + i);
  }
  uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mul_hi(uint32_t *out, uint32_t a) {
  uint32_t acc[8];
  for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i + a;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = __umulhi(acc[i], a + i) + 0x9e3779b9u;
  }
  uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mad_u24(uint32_t *out, uint32_t a) {
  uint32_t acc[8];
  for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i + a;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = __umul24(acc[i], a + i) + acc[i];
  }
  uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_add_u32(uint32_t *out, uint32_t a) {
  uint32_t acc[8];
  for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i + a;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = (acc[i] + a) ^ (acc[i] >> 3);
  }
  uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma_f64(double *out, double a) {
  double acc[8];
  for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = __builtin_fma(acc[i], a, 1.0 + i);
  }
  double s = 0; for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma_f32(float *out, float a) {
  float acc[8];
  for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = __builtin_fmaf(acc[i], a, 1.0f + i);
  }
  float s = 0; for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class P>
__global__ void k_fe_mul(Fe<P> *out, Fe<P> seed, int iters) {
  Fe<P> a = seed, b = seed;
  a.l[0] += threadIdx.x; b.l[1] += blockIdx.x & 0xff;
  for (int it = 0; it < iters; it++) { Fe<P> r; fe_mul(r, a, b); b = a; a = r; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void k_fp_mul_call(fp_t *out, fp_t seed, int iters) {
  fp_t a = seed, b = seed;
  a.l[0] += threadIdx.x; b.l[1] += blockIdx.x & 0xff;
  for (int it = 0; it < iters; it++) { fp_t r = fp_mul_call(a, b); b = a; a = r; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

template <class F> static float timeit(F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  int dev = 0; hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, dev));
  printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  const int blocks = p.multiProcessorCount * 8, threads = 256;
  void *buf; CHK(hipMalloc(&buf, (size_t)blocks * threads * 64));
  const double lanes = (double)blocks * threads;
  float ms;
  ms = timeit([&] { hipLaunchKernelGGL(k_mad_u64_u32, dim3(blocks), dim3(threads), 0, 0, (uint64_t *)buf, 12345u, 678u); });
  printf("v_mad_u64_u32   %8.1f Gop/s  (%.3f ms)\n", lanes * ITER * 8 / ms / 1e6, ms);
  ms = timeit([&] { hipLaunchKernelGGL(k_mul_lo, dim3(blocks), dim3(threads), 0, 0, (uint32_t *)buf, 12345u); });
  printf("v_mul_lo_u32    %8.1f Gop/s  (%.3f ms)\n", lanes * ITER * 8 / ms / 1e6, ms);
  ms = timeit([&] { hipLaunchKernelGGL(k_mul_hi, dim3(blocks), dim3(threads), 0, 0, (uint32_t *)buf, 12345u); });
  printf("v_mul_hi_u32+add%8.1f Gop/s  (%.3f ms)\n", lanes * ITER * 8 / ms / 1e6, ms);
  ms = timeit([&] { hipLaunchKernelGGL(k_mad_u24, dim3(blocks), dim3(threads), 0, 0, (uint32_t *)buf, 12345u); });
  printf("v_mad_u32_u24(+and) %6.1f Gop/s  (%.3f ms)\n", lanes * ITER * 8 / ms / 1e6, ms);
  ms = timeit([&] { hipLaunchKernelGGL(k_add_u32, dim3(blocks), dim3(threads), 0, 0, (uint32_t *)buf, 12345u); });
  printf("add+xor+shift (3 ops) %6.1f Giter/s  (%.3f ms)\n", lanes * ITER * 8 / ms / 1e6, ms);
  ms = timeit([&] { hipLaunchKernelGGL(k_fma_f64, dim3(blocks), dim3(threads), 0, 0, (double *)buf, 1.0000001); });
  printf("v_fma_f64       %8.1f Gop/s  (%.3f ms)\n", lanes * ITER * 8 / ms / 1e6, ms);
  ms = timeit([&] { hipLaunchKernelGGL(k_fma_f32, dim3(blocks), dim3(threads), 0, 0, (float *)buf, 1.0000001f); });
  printf("v_fma_f32       %8.1f Gop/s  (%.3f ms)\n", lanes * ITER * 8 / ms / 1e6, ms);
  fp_t sp; for (int i = 0; i < 12; i++) sp.l[i] = 0x01234567u * (i + 1); sp.l[11] &= 0x0fffffff;
  fr_t sr; for (int i = 0; i < 8; i++) sr.l[i] = 0x01234567u * (i + 1); sr.l[7] &= 0x3fffffff;
  const int it2 = 512;
  ms = timeit([&] { hipLaunchKernelGGL(k_fe_mul<FpParams>, dim3(blocks), dim3(threads), 0, 0, (fp_t *)buf, sp, it2); });
  printf("Fp mul (inline) %8.2f Gmul/s  (%.3f ms)\n", lanes * it2 / ms / 1e6, ms);
  ms = timeit([&] { hipLaunchKernelGGL(k_fp_mul_call, dim3(blocks), dim3(threads), 0, 0, (fp_t *)buf, sp, it2); });
  printf("Fp mul (call)   %8.2f Gmul/s  (%.3f ms)\n", lanes * it2 / ms / 1e6, ms);
  ms = timeit([&] { hipLaunchKernelGGL(k_fe_mul<FrParams>, dim3(blocks), dim3(threads), 0, 0, (fr_t *)buf, sr, it2); });
  printf("Fr mul (inline) %8.2f Gmul/s  (%.3f ms)\n", lanes * it2 / ms / 1e6, ms);
  // HBM copy bandwidth for reference
  size_t nb = (size_t)1 << 30; void *a, *b2; CHK(hipMalloc(&a, nb)); CHK(hipMalloc(&b2, nb));
  ms = timeit([&] { hipMemcpyAsync(b2, a, nb, hipMemcpyDeviceToDevice, 0); });
  printf("D2D copy 1 GiB  %8.1f GB/s (r+w)  (%.3f ms)\n", 2.0 * nb / ms / 1e6, ms);
  return 0;
}

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
// The same product on operands that are ALREADY in 13 limbs of 30 bits and stay so (Montgomery radix 2^390): no
// slicing of the inputs, no repacking of the output - what a curve layer in limb form would call.
struct L13 { u32 l[13]; };
__device__ __forceinline__ void mul_limbs(L13 &r, const L13 &a, const L13 &b) {
  typedef Radix30<FpParams> R;
  constexpr int L = 13;
  u64 c[2 * L];
#pragma unroll
  for (int k = 0; k < 2 * L; k++) c[k] = 0;
#pragma unroll
  for (int i = 0; i < L; i++) {
#pragma unroll
    for (int j = 0; j < L; j++) c[i + j] += (u64)a.l[i] * b.l[j];
  }
  u32 m[L];
  u64 carry = 0;
#pragma unroll
  for (int k = 0; k < L; k++) {
    u64 t = (c[k] & R::MASK) + carry;
#pragma unroll
    for (int i = 0; i < k; i++) t += (u64)m[i] * R::mod(k - i);
    m[k] = ((u32)t * R::INV) & R::MASK;
    t += (u64)m[k] * R::mod(0);
    carry = (t >> 30) + (c[k] >> 30);
  }
#pragma unroll
  for (int k = L; k < 2 * L; k++) {
    u64 t = (c[k] & R::MASK) + carry;
#pragma unroll
    for (int i = k - L + 1; i < L; i++) t += (u64)m[i] * R::mod(k - i);
    r.l[k - L] = (u32)t & R::MASK;
    carry = (t >> 30) + (c[k] >> 30);
  }
}
typedef u32 u32x13 __attribute__((ext_vector_type(13)));
__device__ __attribute__((noinline)) static u32x13 mul_limbs_call(u32x13 a, u32x13 b) {
  L13 x, y, r;
#pragma unroll
  for (int i = 0; i < 13; i++) { x.l[i] = a[i]; y.l[i] = b[i]; }
  mul_limbs(r, x, y);
  u32x13 o;
#pragma unroll
  for (int i = 0; i < 13; i++) o[i] = r.l[i];
  return o;
}
template <bool CALL>
__global__ __launch_bounds__(64) void k_limbs(L13 *out, fp_t seed, int iters) {
  L13 a, b;
#pragma unroll
  for (int i = 0; i < 13; i++) { a.l[i] = (seed.l[i % 12] + threadIdx.x * 977u + i) & 0x3fffffffu; b.l[i] = (seed.l[(i + 5) % 12] + blockIdx.x + 13u * i) & 0x3fffffffu; }
  a.l[12] &= 0xfffff; b.l[12] &= 0xfffff;
  for (int it = 0; it < iters; it++) {
    L13 r;
    if (CALL) {
      u32x13 x, y;
#pragma unroll
      for (int i = 0; i < 13; i++) { x[i] = a.l[i]; y[i] = b.l[i]; }
      const u32x13 o = mul_limbs_call(x, y);
#pragma unroll
      for (int i = 0; i < 13; i++) r.l[i] = o[i];
    } else {
      mul_limbs(r, a, b);
    }
    b = a; a = r;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
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
    ms = timeit([&] { hipLaunchKernelGGL((k_limbs<true>), dim3(blocks), dim3(64), 0, 0, (L13 *)buf, sp, iters); });
    printf("waves/SIMD %d  %-34s %7.3f ms  %7.2f G Fp products/s\n", wps, "limb form, out-of-line product", ms, lanes * 1 * iters / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_limbs<false>), dim3(blocks), dim3(64), 0, 0, (L13 *)buf, sp, iters); });
    printf("waves/SIMD %d  %-34s %7.3f ms  %7.2f G Fp products/s\n", wps, "limb form, inline", ms, lanes * 1 * iters / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_fp2<false>), dim3(blocks), dim3(64), 0, 0, (fp2_t *)buf, sp, iters); });
    printf("waves/SIMD %d  %-34s %7.3f ms  %7.2f G Fp products/s\n", wps, "Fp2 product, 3 calls", ms, lanes * 3 * iters / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_fp2<true>), dim3(blocks), dim3(64), 0, 0, (fp2_t *)buf, sp, iters); });
    printf("waves/SIMD %d  %-34s %7.3f ms  %7.2f G Fp products/s\n", wps, "Fp2 product, 3 inline", ms, lanes * 3 * iters / ms / 1e6);
  }
  return 0;
}

/* A plain-C caller of include/bellman_hip.h (C11, gcc; no C++ runtime in this translation unit): what the reference-side FFI
 * would bind.  Without a GPU it checks the boundary that needs none (version string, the plan query, the
 * host-side group operations, the error code of a context request); with one (argument "gpu") it runs a 4096-term G1
 * multiexp and an FFT round trip through the C ABI and checks them against size-independent identities:
 *   sum_i s_i [t_i]G == [sum_i s_i t_i]G for small t_i, s_i (64-bit arithmetic), icoset_fft(coset_fft(x)) == x.
 * Exit code 0 = all checks passed.  Built and run by tests/test_abi_cpu.py / tests/test_gpu_c_abi.py. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bellman_hip.h"

static const uint64_t G1_GEN[12] = { /* BLS12-381 G1 generator, Montgomery limbs (x | y) */
    0x5CB38790FD530C16ull, 0x7817FC679976FFF5ull, 0x154F95C7143BA1C1ull, 0xF0AE6ACDF3D0E747ull,
    0xEDCE6ECC21DBF440ull, 0x120177419E0BFB75ull, 0xBAAC93D50CE72271ull, 0x8C22631A7918FD8Eull,
    0xDD595F13570725CEull, 0x51AC582950405194ull, 0x0E1C8C3FAD0059C0ull, 0x0BBC3EFC5008A26Aull};

#define CHECK(cond)                                                         \
  do {                                                                      \
    if (!(cond)) {                                                          \
      fprintf(stderr, "abi_smoke: check failed at line %d: %s\n", __LINE__, #cond); \
      return 1;                                                             \
    }                                                                       \
  } while (0)

static int host_checks(void) {
  CHECK(strstr(bh_version(), "gfx950") != NULL);
  unsigned plan[9];
  CHECK(bh_msm_plan_info((size_t)1 << 20, BH_G1, 0, plan) == BH_OK);
  CHECK(plan[0] == 16 && plan[1] == 16 && plan[2] == (1u << 15));   /* c, windows, buckets per window at 2^20 terms */
  CHECK(bh_msm_plan_info(1, 7, 0, plan) == BH_ERR_INVALID_ARG);
  /* [2]G + [3]G == [5]G with the host-side group operations */
  uint64_t k[4] = {2, 0, 0, 0}, p2[12], p3[12], p5[12], sum[12];
  bh_point_mul(BH_G1, p2, G1_GEN, k);
  k[0] = 3; bh_point_mul(BH_G1, p3, G1_GEN, k);
  k[0] = 5; bh_point_mul(BH_G1, p5, G1_GEN, k);
  bh_point_add(BH_G1, sum, p2, p3, 1);
  CHECK(memcmp(sum, p5, sizeof sum) == 0);
  return 0;
}

static int gpu_checks(void) {
  bh_ctx *ctx = NULL;
  CHECK(bh_ctx_create(0, &ctx) == BH_OK && ctx);
  enum { N = 4096 };
  /* bases [t_i]G made by the library on the device, t_i = i + 1 */
  uint64_t *t = calloc((size_t)N * 4, 8), *s = calloc((size_t)N * 4, 8);
  CHECK(t && s);
  unsigned __int128 dot = 0;
  for (int i = 0; i < N; i++) {
    t[4 * i] = (uint64_t)i + 1;
    s[4 * i] = (uint64_t)(i * 2654435761u) >> 8;   /* 24-bit scalars: zeros, ones and repeats included */
    if (i % 7 == 0) s[4 * i] = (uint64_t)(i & 1);
    dot += (unsigned __int128)t[4 * i] * s[4 * i];
  }
  void *dt = NULL, *dbases = NULL;
  CHECK(bh_dev_alloc(ctx, (size_t)N * 32, &dt) == BH_OK && bh_dev_alloc(ctx, (size_t)N * 96, &dbases) == BH_OK);
  CHECK(bh_dev_upload(ctx, dt, t, (size_t)N * 32) == BH_OK);
  CHECK(bh_fixed_base_mul_dev(ctx, BH_G1, G1_GEN, dt, N, 0, dbases, NULL) == BH_OK);
  bh_bases *bases = NULL;
  CHECK(bh_bases_copy_dev(ctx, BH_G1, dbases, N, &bases) == BH_OK);
  bh_msm_job *job = NULL;
  CHECK(bh_msm_async(ctx, bases, 0, s, N, 0, NULL, 0, &job) == BH_OK);
  uint64_t got[12], want[12], stats[8];
  float ms[4];
  CHECK(bh_msm_wait_stats(job, got, ms, stats) == BH_OK);
  uint64_t kk[4] = {(uint64_t)dot, (uint64_t)(dot >> 64), 0, 0};
  bh_point_mul(BH_G1, want, G1_GEN, kk);
  CHECK(memcmp(got, want, sizeof got) == 0);
  CHECK(stats[0] >= N && stats[1] <= stats[0] && stats[2] <= stats[0] - stats[1]);
  /* EOF semantics through the C ABI: one base too few (src/multiexp.rs:55-61) */
  CHECK(bh_msm_async(ctx, bases, 1, s, N, 0, NULL, 0, &job) == BH_OK);
  CHECK(bh_msm_wait(job, got) == BH_ERR_UNEXPECTED_EOF);
  /* FFT round trip on the device vector (Montgomery in, Montgomery out: any 32-byte words below q round-trip) */
  uint64_t *back = calloc((size_t)N * 4, 8);
  CHECK(back);
  CHECK(bh_dev_upload(ctx, dt, s, (size_t)N * 32) == BH_OK);
  CHECK(bh_fft_fr_dev(ctx, dt, 12, 2, NULL) == BH_OK && bh_fft_fr_dev(ctx, dt, 12, 3, NULL) == BH_OK);
  CHECK(bh_ctx_synchronize(ctx) == BH_OK);
  CHECK(bh_dev_download(ctx, back, dt, (size_t)N * 32) == BH_OK);
  CHECK(memcmp(back, s, (size_t)N * 32) == 0);
  CHECK(bh_fft_fr_dev(ctx, dt, 32, 0, NULL) == BH_ERR_DEGREE_TOO_LARGE);
  bh_bases_release(ctx, bases);
  bh_dev_free(ctx, dt);
  bh_dev_free(ctx, dbases);
  bh_ctx_destroy(ctx);
  free(t); free(s); free(back);
  return 0;
}

int main(int argc, char **argv) {
  if (host_checks()) return 1;
  if (argc > 1 && strcmp(argv[1], "gpu") == 0) {
    if (gpu_checks()) return 1;
    puts("abi_smoke: host + gpu checks passed");
  } else {
    bh_ctx *ctx = NULL;
    const int rc = bh_ctx_create(0, &ctx);
    if (rc == BH_OK) bh_ctx_destroy(ctx);
    else if (rc != BH_ERR_NO_DEVICE) { fprintf(stderr, "abi_smoke: bh_ctx_create -> %d\n", rc); return 1; }
    puts("abi_smoke: host checks passed");
  }
  return 0;
}

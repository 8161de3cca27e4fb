/* TEST INFRASTRUCTURE ONLY - CPU oracle (C restatement) of bellman's Groth16 hot path
 * over BLS12-381.  May be loaded ONLY by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  The product (bellman_amd/) never links it.
 *
 * Restates:  /root/reference/src/multiexp.rs   (via curve_tmpl.h)
 *            /root/reference/src/domain.rs     (serial_fft, parallel_fft, best_fft, domain ops)
 *            /root/reference/src/multicore.rs  (log2_floor, scope chunking)
 * Parity: BLS12-381 golden vectors do not exist in the reference ("parity unpinned"
 * by reference vectors); this file is validated against oracle/pyref (which IS pinned
 * by the reference's toy-field KAT) in tests/test_oracle_c_vs_pyref.py.
 *
 * Build: make -C oracle/c   ->  oracle/_build/liboracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>
#include "field.h"

/* ---------------- G1 over Fp ---------------- */
static inline void fp_mul_b3(fp_t *r, const fp_t *a) { /* 3b = 12 */
  fp_t t2, t4, t8; fp_dbl(&t2, a); fp_dbl(&t4, &t2); fp_dbl(&t8, &t4); fp_add(r, &t8, &t4);
}
#define CV(n) g1_##n
#define FE fp_t
#define FE_add fp_add
#define FE_sub fp_sub
#define FE_mul fp_mul
#define FE_sqr fp_sqr
#define FE_neg fp_neg
#define FE_inv fp_inv
#define FE_is_zero fp_is_zero
#define FE_eq fp_eq
#define FE_zero fp_zero
#define FE_one fp_one
#define FE_MUL_B3 fp_mul_b3
#include "curve_tmpl.h"
#undef CV
#undef FE
#undef FE_add
#undef FE_sub
#undef FE_mul
#undef FE_sqr
#undef FE_neg
#undef FE_inv
#undef FE_is_zero
#undef FE_eq
#undef FE_zero
#undef FE_one
#undef FE_MUL_B3

/* ---------------- G2 over Fp2 ---------------- */
static inline void fp2_mul_b3(fp2_t *r, const fp2_t *a) { /* 3b = 12(1+u) */
  fp2_t t; fp_sub(&t.c0, &a->c0, &a->c1); fp_add(&t.c1, &a->c0, &a->c1);
  fp_mul_b3(&r->c0, &t.c0); fp_mul_b3(&r->c1, &t.c1);
}
#define CV(n) g2_##n
#define FE fp2_t
#define FE_add fp2_add
#define FE_sub fp2_sub
#define FE_mul fp2_mul
#define FE_sqr fp2_sqr
#define FE_neg fp2_neg
#define FE_inv fp2_inv
#define FE_is_zero fp2_is_zero
#define FE_eq fp2_eq
#define FE_zero fp2_zero
#define FE_one fp2_one
#define FE_MUL_B3 fp2_mul_b3
#include "curve_tmpl.h"
#undef CV

/* ============================ exported: fields ============================ */
void orc_fr_to_mont(uint64_t *out, const uint64_t *in, size_t n) {
  for (size_t i = 0; i < n; i++) fr_to_mont((fr_t *)(out + 4 * i), (const fr_t *)(in + 4 * i));
}
void orc_fr_from_mont(uint64_t *out, const uint64_t *in, size_t n) {
  for (size_t i = 0; i < n; i++) fr_from_mont((fr_t *)(out + 4 * i), (const fr_t *)(in + 4 * i));
}
void orc_fp_to_mont(uint64_t *out, const uint64_t *in, size_t n) {
  for (size_t i = 0; i < n; i++) fp_to_mont((fp_t *)(out + 6 * i), (const fp_t *)(in + 6 * i));
}
void orc_fp_from_mont(uint64_t *out, const uint64_t *in, size_t n) {
  for (size_t i = 0; i < n; i++) fp_from_mont((fp_t *)(out + 6 * i), (const fp_t *)(in + 6 * i));
}
void orc_fr_mul(uint64_t *r, const uint64_t *a, const uint64_t *b) { fr_mul((fr_t *)r, (const fr_t *)a, (const fr_t *)b); }
void orc_fp_mul(uint64_t *r, const uint64_t *a, const uint64_t *b) { fp_mul((fp_t *)r, (const fp_t *)a, (const fp_t *)b); }
void orc_fr_inv(uint64_t *r, const uint64_t *a) { fr_inv((fr_t *)r, (const fr_t *)a); }

/* ============================ exported: curves ============================
 * Affine points are Montgomery-form coordinates; identity = all-zero record.
 * G1 record: x[6] y[6] (96 B).  G2 record: x.c0[6] x.c1[6] y.c0[6] y.c1[6] (192 B). */
void orc_g1_add(uint64_t *r, const uint64_t *a, const uint64_t *b) {
  g1_proj_t p, q; g1_from_affine(&p, (const g1_aff_t *)a); g1_from_affine(&q, (const g1_aff_t *)b);
  g1_add(&p, &p, &q); g1_to_affine((g1_aff_t *)r, &p);
}
void orc_g2_add(uint64_t *r, const uint64_t *a, const uint64_t *b) {
  g2_proj_t p, q; g2_from_affine(&p, (const g2_aff_t *)a); g2_from_affine(&q, (const g2_aff_t *)b);
  g2_add(&p, &p, &q); g2_to_affine((g2_aff_t *)r, &p);
}
void orc_g1_mul(uint64_t *r, const uint64_t *a, const uint64_t *k /*canonical*/) {
  g1_proj_t p; g1_from_affine(&p, (const g1_aff_t *)a); g1_mul(&p, &p, k); g1_to_affine((g1_aff_t *)r, &p);
}
void orc_g2_mul(uint64_t *r, const uint64_t *a, const uint64_t *k) {
  g2_proj_t p; g2_from_affine(&p, (const g2_aff_t *)a); g2_mul(&p, &p, k); g2_to_affine((g2_aff_t *)r, &p);
}
int orc_g1_on_curve(const uint64_t *a) {
  const g1_aff_t *p = (const g1_aff_t *)a;
  if (g1_aff_is_identity(p)) return 1;
  fp_t l, r, b = {{4, 0, 0, 0, 0, 0}}; fp_to_mont(&b, &b);
  fp_sqr(&l, &p->y); fp_sqr(&r, &p->x); fp_mul(&r, &r, &p->x); fp_add(&r, &r, &b);
  return fp_eq(&l, &r);
}
int orc_g2_on_curve(const uint64_t *a) {
  const g2_aff_t *p = (const g2_aff_t *)a;
  if (g2_aff_is_identity(p)) return 1;
  fp2_t l, r, b; fp_t four = {{4, 0, 0, 0, 0, 0}}; fp_to_mont(&four, &four); b.c0 = four; b.c1 = four;
  fp2_sqr(&l, &p->y); fp2_sqr(&r, &p->x); fp2_mul(&r, &r, &p->x); fp2_add(&r, &r, &b);
  return fp2_eq(&l, &r);
}

/* P_i = [a + i*b]G by incremental addition + batch normalisation: the synthetic
 * base generator of SURVEY.md §8d (distinct, prime-order, never identity for small a,b). */
void orc_g1_gen_bases(uint64_t *out, size_t n, const uint64_t *gen_aff, const uint64_t *a, const uint64_t *b) {
  g1_proj_t g, cur, step; g1_from_affine(&g, (const g1_aff_t *)gen_aff);
  g1_mul(&cur, &g, a); g1_mul(&step, &g, b);
  g1_proj_t *tmp = (g1_proj_t *)malloc((n ? n : 1) * sizeof(g1_proj_t));
  for (size_t i = 0; i < n; i++) { tmp[i] = cur; g1_add(&cur, &cur, &step); }
  g1_batch_to_affine((g1_aff_t *)out, tmp, n); free(tmp);
}
void orc_g2_gen_bases(uint64_t *out, size_t n, const uint64_t *gen_aff, const uint64_t *a, const uint64_t *b) {
  g2_proj_t g, cur, step; g2_from_affine(&g, (const g2_aff_t *)gen_aff);
  g2_mul(&cur, &g, a); g2_mul(&step, &g, b);
  g2_proj_t *tmp = (g2_proj_t *)malloc((n ? n : 1) * sizeof(g2_proj_t));
  for (size_t i = 0; i < n; i++) { tmp[i] = cur; g2_add(&cur, &cur, &step); }
  g2_batch_to_affine((g2_aff_t *)out, tmp, n); free(tmp);
}

/* window-size rule, multiexp.rs:318-322 */
unsigned orc_window_size(size_t n) {
  if (n < 32) return 3;
  return (unsigned)ceil(log((double)(uint32_t)n));
}

/* multiexp(): multiexp.rs:305-332.  c == 0 -> reference rule.  out = affine. */
int orc_multiexp_g1(const uint64_t *bases, size_t nbases, size_t offset, const uint64_t *density,
                    const uint64_t *scalars, size_t n, unsigned c, int threads, uint64_t *out_aff) {
  if (!c) c = orc_window_size(n);
  if (threads <= 0) threads = omp_get_max_threads();
  g1_proj_t r; int rc = g1_multiexp((const g1_aff_t *)bases, nbases, offset, density, scalars, n, c, threads, &r);
  if (rc == 0) g1_to_affine((g1_aff_t *)out_aff, &r);
  return rc;
}
int orc_multiexp_g2(const uint64_t *bases, size_t nbases, size_t offset, const uint64_t *density,
                    const uint64_t *scalars, size_t n, unsigned c, int threads, uint64_t *out_aff) {
  if (!c) c = orc_window_size(n);
  if (threads <= 0) threads = omp_get_max_threads();
  g2_proj_t r; int rc = g2_multiexp((const g2_aff_t *)bases, nbases, offset, density, scalars, n, c, threads, &r);
  if (rc == 0) g2_to_affine((g2_aff_t *)out_aff, &r);
  return rc;
}
void orc_naive_multiexp_g1(const uint64_t *bases, const uint64_t *scalars, size_t n, int threads, uint64_t *out_aff) {
  if (threads <= 0) threads = omp_get_max_threads();
  g1_proj_t r; g1_naive_multiexp((const g1_aff_t *)bases, scalars, n, threads, &r); g1_to_affine((g1_aff_t *)out_aff, &r);
}
void orc_naive_multiexp_g2(const uint64_t *bases, const uint64_t *scalars, size_t n, int threads, uint64_t *out_aff) {
  if (threads <= 0) threads = omp_get_max_threads();
  g2_proj_t r; g2_naive_multiexp((const g2_aff_t *)bases, scalars, n, threads, &r); g2_to_affine((g2_aff_t *)out_aff, &r);
}

/* ============================ domain.rs ============================ */
static unsigned log2_floor(unsigned num) { /* multicore.rs:120-130 */
  unsigned p = 0; while ((1u << (p + 1)) <= num) p++; return p;
}
static uint32_t bitreverse(uint32_t n, uint32_t l) { /* domain.rs:273-280 */
  uint32_t r = 0; for (uint32_t i = 0; i < l; i++) { r = (r << 1) | (n & 1); n >>= 1; } return r;
}
static void serial_fft(fr_t *a, const fr_t *omega, uint32_t log_n) { /* domain.rs:272-314 */
  uint32_t n = 1u << log_n;
  for (uint32_t k = 0; k < n; k++) {
    uint32_t rk = bitreverse(k, log_n);
    if (k < rk) { fr_t t = a[rk]; a[rk] = a[k]; a[k] = t; }
  }
  uint32_t m = 1;
  for (uint32_t s = 0; s < log_n; s++) {
    fr_t w_m; fr_pow_u64(&w_m, omega, n / (2 * m));
    for (uint32_t k = 0; k < n; k += 2 * m) {
      fr_t w; fr_one(&w);
      for (uint32_t j = 0; j < m; j++) {
        fr_t t, tmp; fr_mul(&t, &a[k + j + m], &w);
        fr_sub(&tmp, &a[k + j], &t); a[k + j + m] = tmp;
        fr_add(&a[k + j], &a[k + j], &t);
        fr_mul(&w, &w, &w_m);
      }
    }
    m *= 2;
  }
}
static void parallel_fft(fr_t *a, const fr_t *omega, uint32_t log_n, uint32_t log_cpus, int threads) { /* domain.rs:316-372 */
  uint32_t num_cpus = 1u << log_cpus, log_new_n = log_n - log_cpus;
  size_t sub = (size_t)1 << log_new_n, n = (size_t)1 << log_n;
  fr_t *tmp = (fr_t *)calloc((size_t)num_cpus * sub, sizeof(fr_t));
  fr_t new_omega; fr_pow_u64(&new_omega, omega, num_cpus);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
  for (uint32_t j = 0; j < num_cpus; j++) {
    fr_t *t = tmp + (size_t)j * sub;
    fr_t omega_j, omega_step, elt;
    fr_pow_u64(&omega_j, omega, j);
    fr_pow_u64(&omega_step, omega, (uint64_t)j << log_new_n);
    fr_one(&elt);
    for (size_t i = 0; i < sub; i++) {
      for (uint32_t s = 0; s < num_cpus; s++) {
        size_t idx = (i + ((size_t)s << log_new_n)) % n;
        fr_t x; fr_mul(&x, &a[idx], &elt); fr_add(&t[i], &t[i], &x);
        fr_mul(&elt, &elt, &omega_step);
      }
      fr_mul(&elt, &elt, &omega_j);
    }
    serial_fft(t, &new_omega, log_new_n);
  }
  size_t mask = ((size_t)1 << log_cpus) - 1;
#pragma omp parallel for schedule(static) num_threads(threads)
  for (size_t idx = 0; idx < n; idx++) a[idx] = tmp[(idx & mask) * sub + (idx >> log_cpus)];
  free(tmp);
}
static void best_fft(fr_t *a, const fr_t *omega, uint32_t log_n, int threads) { /* domain.rs:261-269 */
  uint32_t log_cpus = log2_floor((unsigned)threads);
  if (log_n <= log_cpus) serial_fft(a, omega, log_n); else parallel_fft(a, omega, log_n, log_cpus, threads);
}
static void domain_omega(fr_t *omega, uint32_t exp) { /* domain.rs:62-66 */
  /* ROOT_OF_UNITY = 7^((q-1)/2^32), canonical value checked in tests */
  static const uint64_t ROU[4] = {0x3829971f439f0d2bULL, 0xb63683508c2280b9ULL, 0xd09b681922c813b4ULL, 0x16a2a19edfe81f20ULL};
  fr_t w; memcpy(w.l, ROU, sizeof ROU); fr_to_mont(&w, &w);
  for (uint32_t i = exp; i < 32; i++) fr_sqr(&w, &w);
  *omega = w;
}
static void scale_all(fr_t *a, size_t n, const fr_t *k, int threads) {
#pragma omp parallel for schedule(static) num_threads(threads)
  for (size_t i = 0; i < n; i++) fr_mul(&a[i], &a[i], k);
}
static void distribute_powers(fr_t *a, size_t n, const fr_t *g, int threads) { /* domain.rs:101-113 */
  size_t chunk = n < (size_t)threads ? 1 : n / (size_t)threads;   /* multicore.rs:83-88 */
  size_t nchunks = (n + chunk - 1) / chunk;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
  for (size_t i = 0; i < nchunks; i++) {
    fr_t u; fr_pow_u64(&u, g, (uint64_t)(i * chunk));
    size_t end = (i + 1) * chunk; if (end > n) end = n;
    for (size_t k = i * chunk; k < end; k++) { fr_mul(&a[k], &a[k], &u); fr_mul(&u, &u, g); }
  }
}
static void fr_from_u64(fr_t *r, uint64_t v) { fr_t c = {{v, 0, 0, 0}}; fr_to_mont(r, &c); }

/* mode: 0 fft, 1 ifft, 2 coset_fft, 3 icoset_fft  (domain.rs:81-125).
 * data: 2^log_n Montgomery-form Fr, in place.  threads = rayon pool size analogue. */
void orc_fft(uint64_t *data, uint32_t log_n, int mode, int threads) {
  if (threads <= 0) threads = omp_get_max_threads();
  fr_t *a = (fr_t *)data; size_t n = (size_t)1 << log_n;
  fr_t omega, omegainv, g, ginv, minv;
  domain_omega(&omega, log_n); fr_inv(&omegainv, &omega);
  fr_from_u64(&g, 7); fr_inv(&ginv, &g);
  fr_from_u64(&minv, (uint64_t)n); fr_inv(&minv, &minv);
  switch (mode) {
    case 0: best_fft(a, &omega, log_n, threads); break;
    case 1: best_fft(a, &omegainv, log_n, threads); scale_all(a, n, &minv, threads); break;
    case 2: distribute_powers(a, n, &g, threads); best_fft(a, &omega, log_n, threads); break;
    case 3: best_fft(a, &omegainv, log_n, threads); scale_all(a, n, &minv, threads);
            distribute_powers(a, n, &ginv, threads); break;
  }
}
void orc_serial_fft(uint64_t *data, uint32_t log_n) { fr_t w; domain_omega(&w, log_n); serial_fft((fr_t *)data, &w, log_n); }
void orc_parallel_fft(uint64_t *data, uint32_t log_n, uint32_t log_cpus, int threads) {
  fr_t w; domain_omega(&w, log_n); parallel_fft((fr_t *)data, &w, log_n, log_cpus, threads > 0 ? threads : 1);
}
void orc_distribute_powers(uint64_t *data, size_t n, const uint64_t *g_mont, int threads) {
  distribute_powers((fr_t *)data, n, (const fr_t *)g_mont, threads > 0 ? threads : omp_get_max_threads());
}
void orc_mul_assign(uint64_t *a, const uint64_t *b, size_t n) { /* domain.rs:154-170 */
  for (size_t i = 0; i < n; i++) fr_mul((fr_t *)(a + 4 * i), (const fr_t *)(a + 4 * i), (const fr_t *)(b + 4 * i));
}
void orc_sub_assign(uint64_t *a, const uint64_t *b, size_t n) { /* domain.rs:173-189 */
  for (size_t i = 0; i < n; i++) fr_sub((fr_t *)(a + 4 * i), (const fr_t *)(a + 4 * i), (const fr_t *)(b + 4 * i));
}
void orc_divide_by_z_on_coset(uint64_t *a, uint32_t log_n, int threads) { /* domain.rs:129-151 */
  if (threads <= 0) threads = omp_get_max_threads();
  size_t n = (size_t)1 << log_n;
  fr_t g, z, one; fr_from_u64(&g, 7); fr_pow_u64(&z, &g, (uint64_t)n); fr_one(&one); fr_sub(&z, &z, &one); fr_inv(&z, &z);
  scale_all((fr_t *)a, n, &z, threads);
}
/* prover.rs:221-240: a,b,c = evaluations padded to 2^log_n (Montgomery); result (m entries, last
 * one to be dropped by the caller) left in a.  b, c are clobbered. */
void orc_h_coeffs(uint64_t *a, uint64_t *b, uint64_t *c, uint32_t log_n, int threads) {
  size_t n = (size_t)1 << log_n;
  orc_fft(a, log_n, 1, threads); orc_fft(a, log_n, 2, threads);
  orc_fft(b, log_n, 1, threads); orc_fft(b, log_n, 2, threads);
  orc_fft(c, log_n, 1, threads); orc_fft(c, log_n, 2, threads);
  orc_mul_assign(a, b, n); orc_sub_assign(a, c, n);
  orc_divide_by_z_on_coset(a, log_n, threads);
  orc_fft(a, log_n, 3, threads);
}
int orc_max_threads(void) { return omp_get_max_threads(); }
/* Test helper (not a reference function): sum_i a[i]*b[i] mod q over CANONICAL inputs, canonical output.
 * Used for the size-independent MSM property  sum_i s_i [t_i]G = [sum_i s_i t_i]G  at sizes where the
 * restated multiexp itself takes minutes. */
void orc_fr_dot(uint64_t *out, const uint64_t *a, const uint64_t *b, size_t n, int threads) {
  if (threads <= 0) threads = omp_get_max_threads();
  fr_t *part = (fr_t *)calloc((size_t)threads, sizeof(fr_t));
#pragma omp parallel num_threads(threads)
  {
    fr_t acc; memset(&acc, 0, sizeof acc);
    const int t = omp_get_thread_num(), nt = omp_get_num_threads();
    const size_t lo = n * (size_t)t / (size_t)nt, hi = n * (size_t)(t + 1) / (size_t)nt;
    for (size_t i = lo; i < hi; i++) {
      fr_t x, y, z;
      fr_to_mont(&x, (const fr_t *)(a + 4 * i)); fr_to_mont(&y, (const fr_t *)(b + 4 * i));
      fr_mul(&z, &x, &y); fr_add(&acc, &acc, &z);
    }
    part[t] = acc;
  }
  fr_t tot; memset(&tot, 0, sizeof tot);
  for (int t = 0; t < threads; t++) fr_add(&tot, &tot, &part[t]);
  fr_from_mont((fr_t *)out, &tot);
  free(part);
}

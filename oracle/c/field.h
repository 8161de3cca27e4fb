/* TEST INFRASTRUCTURE ONLY - CPU oracle for bellman's Groth16 hot path.
 *
 * BLS12-381 field arithmetic, 64-bit limbs, Montgomery form, unsigned __int128.
 * Restates the published arithmetic of the third-party crate `bls12_381 0.8.0`
 * (pinned at /root/reference/Cargo.lock:105-108; source not vendored): Fr is
 * 4x64 Montgomery (R = 2^256), Fp is 6x64 Montgomery (R = 2^384).  Reached from
 * the reference through trait calls at src/domain.rs:250-258 (Fr mul/add/sub),
 * src/multiexp.rs:39,273-274,299 (point ops -> Fp / Fp2 ops).
 *
 * Parity pinning for BLS12-381: UNPINNED by reference golden vectors (the
 * reference has none); pinned by oracle/pyref (big-int, affine formulas), the
 * compressed-generator KAT and algebraic properties.  See tests/.
 */
#ifndef ORACLE_FIELD_H
#define ORACLE_FIELD_H
#include <stdint.h>
#include <string.h>

typedef unsigned __int128 u128;

typedef struct { uint64_t l[4]; } fr_t;
typedef struct { uint64_t l[6]; } fp_t;
typedef struct { fp_t c0, c1; } fp2_t;

/* ---- moduli and Montgomery constants (verified in tests) ---- */
static const uint64_t FR_MOD[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL,
                                   0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
static const uint64_t FR_INV = 0xfffffffeffffffffULL; /* -q^-1 mod 2^64 */
static const uint64_t FR_R[4] = {0x00000001fffffffeULL, 0x5884b7fa00034802ULL,
                                 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL};
static const uint64_t FR_R2[4] = {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL,
                                  0x05d314967254398fULL, 0x0748d9d99f59ff11ULL};

static const uint64_t FP_MOD[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL,
                                   0x6730d2a0f6b0f624ULL, 0x64774b84f38512bfULL,
                                   0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const uint64_t FP_INV = 0x89f3fffcfffcfffdULL; /* -p^-1 mod 2^64 */
static const uint64_t FP_R[6] = {0x760900000002fffdULL, 0xebf4000bc40c0002ULL,
                                 0x5f48985753c758baULL, 0x77ce585370525745ULL,
                                 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL};
static const uint64_t FP_R2[6] = {0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL,
                                  0x8de5476c4c95b6d5ULL, 0x67eb88a9939d83c0ULL,
                                  0x9a793e85b519952dULL, 0x11988fe592cae3aaULL};

/* ---- generic N-limb helpers (N is a compile-time constant at each call) ---- */
#define DEF_FIELD(NAME, T, N, MOD, INV)                                              \
  static inline int NAME##_is_zero(const T *a) {                                     \
    uint64_t o = 0;                                                                  \
    for (int i = 0; i < N; i++) o |= a->l[i];                                        \
    return o == 0;                                                                   \
  }                                                                                  \
  static inline int NAME##_eq(const T *a, const T *b) {                              \
    uint64_t o = 0;                                                                  \
    for (int i = 0; i < N; i++) o |= a->l[i] ^ b->l[i];                              \
    return o == 0;                                                                   \
  }                                                                                  \
  static inline int NAME##_geq_mod(const uint64_t *a) {                              \
    for (int i = N - 1; i >= 0; i--) {                                               \
      if (a[i] > MOD[i]) return 1;                                                   \
      if (a[i] < MOD[i]) return 0;                                                   \
    }                                                                                \
    return 1;                                                                        \
  }                                                                                  \
  static inline void NAME##_sub_mod_raw(uint64_t *a) {                               \
    u128 br = 0;                                                                     \
    for (int i = 0; i < N; i++) {                                                    \
      u128 d = (u128)a[i] - MOD[i] - (uint64_t)br;                                   \
      a[i] = (uint64_t)d;                                                            \
      br = (d >> 64) & 1;                                                            \
    }                                                                                \
  }                                                                                  \
  static inline void NAME##_add(T *r, const T *a, const T *b) {                      \
    u128 c = 0;                                                                      \
    uint64_t t[N];                                                                   \
    for (int i = 0; i < N; i++) {                                                    \
      c += (u128)a->l[i] + b->l[i];                                                  \
      t[i] = (uint64_t)c;                                                            \
      c >>= 64;                                                                      \
    }                                                                                \
    if (c || NAME##_geq_mod(t)) NAME##_sub_mod_raw(t);                               \
    memcpy(r->l, t, sizeof t);                                                       \
  }                                                                                  \
  static inline void NAME##_sub(T *r, const T *a, const T *b) {                      \
    uint64_t t[N];                                                                   \
    u128 br = 0;                                                                     \
    for (int i = 0; i < N; i++) {                                                    \
      u128 d = (u128)a->l[i] - b->l[i] - (uint64_t)br;                               \
      t[i] = (uint64_t)d;                                                            \
      br = (d >> 64) & 1;                                                            \
    }                                                                                \
    if (br) {                                                                        \
      u128 c = 0;                                                                    \
      for (int i = 0; i < N; i++) {                                                  \
        c += (u128)t[i] + MOD[i];                                                    \
        t[i] = (uint64_t)c;                                                          \
        c >>= 64;                                                                    \
      }                                                                              \
    }                                                                                \
    memcpy(r->l, t, sizeof t);                                                       \
  }                                                                                  \
  static inline void NAME##_neg(T *r, const T *a) {                                  \
    T z;                                                                             \
    memset(&z, 0, sizeof z);                                                         \
    NAME##_sub(r, &z, a);                                                            \
  }                                                                                  \
  static inline void NAME##_dbl(T *r, const T *a) { NAME##_add(r, a, a); }           \
  /* CIOS Montgomery product: r = a*b*R^-1 mod MOD */                                \
  static inline void NAME##_mul(T *r, const T *a, const T *b) {                      \
    uint64_t t[N + 2];                                                               \
    memset(t, 0, sizeof t);                                                          \
    for (int i = 0; i < N; i++) {                                                    \
      u128 c = 0;                                                                    \
      for (int j = 0; j < N; j++) {                                                  \
        c += (u128)a->l[j] * b->l[i] + t[j];                                         \
        t[j] = (uint64_t)c;                                                          \
        c >>= 64;                                                                    \
      }                                                                              \
      c += t[N];                                                                     \
      t[N] = (uint64_t)c;                                                            \
      t[N + 1] = (uint64_t)(c >> 64);                                                \
      uint64_t m = t[0] * INV;                                                       \
      c = ((u128)m * MOD[0] + t[0]) >> 64;                                           \
      for (int j = 1; j < N; j++) {                                                  \
        c += (u128)m * MOD[j] + t[j];                                                \
        t[j - 1] = (uint64_t)c;                                                      \
        c >>= 64;                                                                    \
      }                                                                              \
      c += t[N];                                                                     \
      t[N - 1] = (uint64_t)c;                                                        \
      t[N] = t[N + 1] + (uint64_t)(c >> 64);                                         \
    }                                                                                \
    if (t[N] || NAME##_geq_mod(t)) NAME##_sub_mod_raw(t);                            \
    memcpy(r->l, t, N * sizeof(uint64_t));                                           \
  }                                                                                  \
  static inline void NAME##_sqr(T *r, const T *a) { NAME##_mul(r, a, a); }

DEF_FIELD(fr, fr_t, 4, FR_MOD, FR_INV)
DEF_FIELD(fp, fp_t, 6, FP_MOD, FP_INV)

static inline void fr_one(fr_t *r) { memcpy(r->l, FR_R, sizeof FR_R); }
static inline void fp_one(fp_t *r) { memcpy(r->l, FP_R, sizeof FP_R); }
static inline void fr_zero(fr_t *r) { memset(r, 0, sizeof *r); }
static inline void fp_zero(fp_t *r) { memset(r, 0, sizeof *r); }

static inline void fr_to_mont(fr_t *r, const fr_t *canon) {
  fr_t r2; memcpy(r2.l, FR_R2, sizeof FR_R2); fr_mul(r, canon, &r2);
}
static inline void fr_from_mont(fr_t *r, const fr_t *m) {
  fr_t one = {{1, 0, 0, 0}}; fr_mul(r, m, &one);
}
static inline void fp_to_mont(fp_t *r, const fp_t *canon) {
  fp_t r2; memcpy(r2.l, FP_R2, sizeof FP_R2); fp_mul(r, canon, &r2);
}
static inline void fp_from_mont(fp_t *r, const fp_t *m) {
  fp_t one = {{1, 0, 0, 0, 0, 0}}; fp_mul(r, m, &one);
}

/* r = a^e, e given as little-endian 64-bit limbs (pow_vartime, domain.rs:105 etc.) */
static inline void fr_pow(fr_t *r, const fr_t *a, const uint64_t *e, int nlimbs) {
  fr_t acc; fr_one(&acc);
  for (int i = nlimbs * 64 - 1; i >= 0; i--) {
    fr_sqr(&acc, &acc);
    if ((e[i / 64] >> (i % 64)) & 1) fr_mul(&acc, &acc, a);
  }
  *r = acc;
}
static inline void fr_pow_u64(fr_t *r, const fr_t *a, uint64_t e) { fr_pow(r, a, &e, 1); }
static inline void fr_inv(fr_t *r, const fr_t *a) { /* a^(q-2) */
  uint64_t e[4]; memcpy(e, FR_MOD, sizeof e); e[0] -= 2; fr_pow(r, a, e, 4);
}
static inline void fp_inv(fp_t *r, const fp_t *a) { /* a^(p-2) */
  uint64_t e[6]; memcpy(e, FP_MOD, sizeof e); e[0] -= 2;
  fp_t acc; fp_one(&acc);
  for (int i = 383; i >= 0; i--) {
    fp_sqr(&acc, &acc);
    if ((e[i / 64] >> (i % 64)) & 1) fp_mul(&acc, &acc, a);
  }
  *r = acc;
}

/* ---- Fp2 = Fp[u]/(u^2+1) ---- */
static inline int fp2_is_zero(const fp2_t *a) { return fp_is_zero(&a->c0) && fp_is_zero(&a->c1); }
static inline int fp2_eq(const fp2_t *a, const fp2_t *b) { return fp_eq(&a->c0, &b->c0) && fp_eq(&a->c1, &b->c1); }
static inline void fp2_zero(fp2_t *r) { memset(r, 0, sizeof *r); }
static inline void fp2_one(fp2_t *r) { fp_one(&r->c0); fp_zero(&r->c1); }
static inline void fp2_add(fp2_t *r, const fp2_t *a, const fp2_t *b) { fp_add(&r->c0, &a->c0, &b->c0); fp_add(&r->c1, &a->c1, &b->c1); }
static inline void fp2_sub(fp2_t *r, const fp2_t *a, const fp2_t *b) { fp_sub(&r->c0, &a->c0, &b->c0); fp_sub(&r->c1, &a->c1, &b->c1); }
static inline void fp2_neg(fp2_t *r, const fp2_t *a) { fp_neg(&r->c0, &a->c0); fp_neg(&r->c1, &a->c1); }
static inline void fp2_dbl(fp2_t *r, const fp2_t *a) { fp2_add(r, a, a); }
static inline void fp2_mul(fp2_t *r, const fp2_t *a, const fp2_t *b) {
  fp_t t0, t1, t2, t3;
  fp_mul(&t0, &a->c0, &b->c0);
  fp_mul(&t1, &a->c1, &b->c1);
  fp_add(&t2, &a->c0, &a->c1);
  fp_add(&t3, &b->c0, &b->c1);
  fp_mul(&t2, &t2, &t3);      /* (a0+a1)(b0+b1) */
  fp_sub(&t2, &t2, &t0);
  fp_sub(&r->c1, &t2, &t1);   /* a0b1 + a1b0 */
  fp_sub(&r->c0, &t0, &t1);   /* a0b0 - a1b1 */
}
static inline void fp2_sqr(fp2_t *r, const fp2_t *a) { fp2_mul(r, a, a); }
static inline void fp2_inv(fp2_t *r, const fp2_t *a) {
  fp_t n, t;
  fp_sqr(&n, &a->c0); fp_sqr(&t, &a->c1); fp_add(&n, &n, &t); fp_inv(&n, &n);
  fp_mul(&r->c0, &a->c0, &n);
  fp_mul(&t, &a->c1, &n); fp_neg(&r->c1, &t);
}
#endif

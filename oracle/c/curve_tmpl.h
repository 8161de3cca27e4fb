/* TEST INFRASTRUCTURE ONLY.
 * Short-Weierstrass a=0 curve template in homogeneous projective coordinates with
 * the complete formulas of Renes-Costello-Batina (eprint 2015/1060, Alg. 7/8/9),
 * the formulas `bls12_381 0.8.0` (Cargo.lock:105-108) is documented to use for
 * G1Projective/G2Projective.  Instantiate by defining before inclusion:
 *   CV(name)  - symbol prefixer          FE   - coordinate field type
 *   FE_xxx    - add sub mul sqr neg inv dbl is_zero eq zero one
 *   FE_MUL_B3 - r = 3*b * a
 * The reference reaches these through `AddAssign<&Affine>` (multiexp.rs:39),
 * `AddAssign<&G>` (:273-274), `double()` (:299), `to_affine` (prover.rs:356-360).
 * Results are only ever compared as group elements (affine), multiexp.rs:377.
 */

typedef struct { FE x, y; } CV(aff_t);        /* identity encoded as (0,0) - not on curve */
typedef struct { FE x, y, z; } CV(proj_t);    /* identity = (0:1:0) */

static inline int CV(aff_is_identity)(const CV(aff_t) *p) { return FE_is_zero(&p->x) && FE_is_zero(&p->y); }
static inline void CV(proj_identity)(CV(proj_t) *p) { FE_zero(&p->x); FE_one(&p->y); FE_zero(&p->z); }
static inline int CV(proj_is_identity)(const CV(proj_t) *p) { return FE_is_zero(&p->z); }
static inline void CV(from_affine)(CV(proj_t) *r, const CV(aff_t) *p) {
  if (CV(aff_is_identity)(p)) { CV(proj_identity)(r); return; }
  r->x = p->x; r->y = p->y; FE_one(&r->z);
}

/* Alg. 7: complete addition */
static inline void CV(add)(CV(proj_t) *r, const CV(proj_t) *p, const CV(proj_t) *q) {
  FE t0, t1, t2, t3, t4, x3, y3, z3;
  FE_mul(&t0, &p->x, &q->x); FE_mul(&t1, &p->y, &q->y); FE_mul(&t2, &p->z, &q->z);
  FE_add(&t3, &p->x, &p->y); FE_add(&t4, &q->x, &q->y); FE_mul(&t3, &t3, &t4);
  FE_add(&t4, &t0, &t1); FE_sub(&t3, &t3, &t4); FE_add(&t4, &p->y, &p->z);
  FE_add(&x3, &q->y, &q->z); FE_mul(&t4, &t4, &x3); FE_add(&x3, &t1, &t2);
  FE_sub(&t4, &t4, &x3); FE_add(&x3, &p->x, &p->z); FE_add(&y3, &q->x, &q->z);
  FE_mul(&x3, &x3, &y3); FE_add(&y3, &t0, &t2); FE_sub(&y3, &x3, &y3);
  FE_add(&x3, &t0, &t0); FE_add(&t0, &x3, &t0); FE_MUL_B3(&t2, &t2);
  FE_add(&z3, &t1, &t2); FE_sub(&t1, &t1, &t2); FE_MUL_B3(&y3, &y3);
  FE_mul(&x3, &t4, &y3); FE_mul(&t2, &t3, &t1); FE_sub(&x3, &t2, &x3);
  FE_mul(&y3, &y3, &t0); FE_mul(&t1, &t1, &z3); FE_add(&y3, &t1, &y3);
  FE_mul(&t0, &t0, &t3); FE_mul(&z3, &z3, &t4); FE_add(&z3, &z3, &t0);
  r->x = x3; r->y = y3; r->z = z3;
}

/* Alg. 8: complete mixed addition (q affine, q != identity) */
static inline void CV(add_mixed)(CV(proj_t) *r, const CV(proj_t) *p, const CV(aff_t) *q) {
  if (CV(aff_is_identity)(q)) { *r = *p; return; }
  FE t0, t1, t2, t3, t4, x3, y3, z3;
  FE_mul(&t0, &p->x, &q->x); FE_mul(&t1, &p->y, &q->y); FE_add(&t3, &q->x, &q->y);
  FE_add(&t4, &p->x, &p->y); FE_mul(&t3, &t3, &t4); FE_add(&t4, &t0, &t1);
  FE_sub(&t3, &t3, &t4); FE_mul(&t4, &q->y, &p->z); FE_add(&t4, &t4, &p->y);
  FE_mul(&y3, &q->x, &p->z); FE_add(&y3, &y3, &p->x); FE_add(&x3, &t0, &t0);
  FE_add(&t0, &x3, &t0); FE_MUL_B3(&t2, &p->z); FE_add(&z3, &t1, &t2);
  FE_sub(&t1, &t1, &t2); FE_MUL_B3(&y3, &y3); FE_mul(&x3, &t4, &y3);
  FE_mul(&t2, &t3, &t1); FE_sub(&x3, &t2, &x3); FE_mul(&y3, &y3, &t0);
  FE_mul(&t1, &t1, &z3); FE_add(&y3, &t1, &y3); FE_mul(&t0, &t0, &t3);
  FE_mul(&z3, &z3, &t4); FE_add(&z3, &z3, &t0);
  r->x = x3; r->y = y3; r->z = z3;
}

/* Alg. 9: doubling */
static inline void CV(double)(CV(proj_t) *r, const CV(proj_t) *p) {
  FE t0, t1, t2, x3, y3, z3;
  FE_sqr(&t0, &p->y); FE_add(&z3, &t0, &t0); FE_add(&z3, &z3, &z3);
  FE_add(&z3, &z3, &z3); FE_mul(&t1, &p->y, &p->z); FE_sqr(&t2, &p->z);
  FE_MUL_B3(&t2, &t2); FE_mul(&x3, &t2, &z3); FE_add(&y3, &t0, &t2);
  FE_mul(&z3, &t1, &z3); FE_add(&t1, &t2, &t2); FE_add(&t2, &t1, &t2);
  FE_sub(&t0, &t0, &t2); FE_mul(&y3, &t0, &y3); FE_add(&y3, &x3, &y3);
  FE_mul(&t1, &p->x, &p->y); FE_mul(&x3, &t0, &t1); FE_add(&x3, &x3, &x3);
  r->x = x3; r->y = y3; r->z = z3;
}

static inline void CV(to_affine)(CV(aff_t) *r, const CV(proj_t) *p) {
  if (CV(proj_is_identity)(p)) { FE_zero(&r->x); FE_zero(&r->y); return; }
  FE zi; FE_inv(&zi, &p->z);
  FE_mul(&r->x, &p->x, &zi); FE_mul(&r->y, &p->y, &zi);
}

/* [k]p, k = 4x64 canonical little-endian; MSB-first double-and-add */
static inline void CV(mul)(CV(proj_t) *r, const CV(proj_t) *p, const uint64_t k[4]) {
  CV(proj_t) acc; CV(proj_identity)(&acc);
  for (int i = 255; i >= 0; i--) {
    CV(double)(&acc, &acc);
    if ((k[i / 64] >> (i % 64)) & 1) CV(add)(&acc, &acc, p);
  }
  *r = acc;
}

/* Montgomery's trick batch normalisation (group::Curve::batch_normalize) */
static void CV(batch_to_affine)(CV(aff_t) *out, const CV(proj_t) *in, size_t n) {
  if (n == 0) return;
  FE *pref = (FE *)malloc(n * sizeof(FE));
  FE acc; FE_one(&acc);
  for (size_t i = 0; i < n; i++) {
    pref[i] = acc;
    if (!FE_is_zero(&in[i].z)) FE_mul(&acc, &acc, &in[i].z);
  }
  FE inv; FE_inv(&inv, &acc);
  for (size_t i = n; i-- > 0;) {
    if (FE_is_zero(&in[i].z)) { FE_zero(&out[i].x); FE_zero(&out[i].y); continue; }
    FE zi; FE_mul(&zi, &inv, &pref[i]);
    FE_mul(&inv, &inv, &in[i].z);
    FE_mul(&out[i].x, &in[i].x, &zi); FE_mul(&out[i].y, &in[i].y, &zi);
  }
  free(pref);
}

/* ------------------------------------------------------------------------------
 * multiexp_inner restated: /root/reference/src/multiexp.rs:210-301.
 * bases: affine array (compacted), nbases entries, cursor starts at `offset`
 *        (the `(Arc<Vec<G>>, usize)` Source, multiexp.rs:45-86).
 * density: NULL = FullDensity, else LSB0 bitmap over scalar indices (:117-131).
 * scalars: n canonical little-endian 4x64 (Exponent::Bits, :172-182); the Zero /
 *          One classification (:174-177) is redone here from the value.
 * Returns 0 ok, 1 UnexpectedIdentity, 2 UnexpectedEof (error of the HIGHEST
 * window wins, :295-300; within a window the first failing index).
 * One OpenMP task per window == rayon's into_par_iter over windows (:288-293).
 */
static int CV(multiexp)(const CV(aff_t) *bases, size_t nbases, size_t offset,
                        const uint64_t *density, const uint64_t *scalars, size_t n,
                        unsigned c, int threads, CV(proj_t) *out) {
  unsigned nwin = (255 + c - 1) / c;            /* (0..NUM_BITS).step_by(c) */
  CV(proj_t) *parts = (CV(proj_t) *)malloc(nwin * sizeof(CV(proj_t)));
  int *errs = (int *)calloc(nwin, sizeof(int));
  /* classification, multiexp.rs:172-182 */
  uint8_t *kind = (uint8_t *)malloc(n ? n : 1);
  for (size_t i = 0; i < n; i++) {
    const uint64_t *s = scalars + 4 * i;
    if ((s[0] | s[1] | s[2] | s[3]) == 0) kind[i] = 0;
    else if (s[0] == 1 && (s[1] | s[2] | s[3]) == 0) kind[i] = 1;
    else kind[i] = 2;
  }
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
  for (unsigned chunk = 0; chunk < nwin; chunk++) {
    CV(proj_t) acc; CV(proj_identity)(&acc);
    size_t nb = ((size_t)1 << c) - 1;
    CV(proj_t) *buckets = (CV(proj_t) *)malloc(nb * sizeof(CV(proj_t)));
    for (size_t b = 0; b < nb; b++) CV(proj_identity)(&buckets[b]);
    size_t pos = offset;
    int err = 0;
    unsigned lo = chunk * c;
    for (size_t i = 0; i < n && !err; i++) {
      if (density && !((density[i >> 6] >> (i & 63)) & 1)) continue;
      /* both next() and skip() check EOF first (multiexp.rs:55-61,74-80) */
      if (nbases <= pos) { err = 2; break; }
      uint64_t digit = 0;
      int consume = 0, to_acc = 0;
      if (kind[i] == 1) { if (chunk == 0) { consume = 1; to_acc = 1; } }
      else if (kind[i] == 2) {
        const uint64_t *s = scalars + 4 * i;
        /* Exponent::chunks (:195-203): c-bit LSB-first slices of the 256-bit repr */
        unsigned w = lo / 64, sh = lo % 64;
        digit = s[w] >> sh;
        if (sh + c > 64 && w + 1 < 4) digit |= s[w + 1] << (64 - sh);
        digit &= (((uint64_t)1 << c) - 1);
        if (lo + c > 256) digit &= (((uint64_t)1 << (256 - lo)) - 1);
        if (digit != 0) consume = 1;
      }
      if (consume) {
        if (CV(aff_is_identity)(&bases[pos])) { err = 1; break; }   /* :63-65 */
        if (to_acc) CV(add_mixed)(&acc, &acc, &bases[pos]);
        else CV(add_mixed)(&buckets[digit - 1], &buckets[digit - 1], &bases[pos]);
      }
      pos++;
    }
    if (!err) {                                 /* summation by parts :271-275 */
      CV(proj_t) running; CV(proj_identity)(&running);
      for (size_t b = nb; b-- > 0;) {
        CV(add)(&running, &running, &buckets[b]);
        CV(add)(&acc, &acc, &running);
      }
    }
    free(buckets);
    parts[chunk] = acc; errs[chunk] = err;
  }
  int rc = 0;
  CV(proj_t) acc; CV(proj_identity)(&acc);
  for (unsigned w = nwin; w-- > 0;) {           /* :295-300 */
    if (errs[w]) { rc = errs[w]; break; }
    for (unsigned k = 0; k < c; k++) CV(double)(&acc, &acc);
    CV(add)(&acc, &acc, &parts[w]);
  }
  free(parts); free(errs); free(kind);
  if (rc == 0) *out = acc;
  return rc;
}

/* naive Σ s_i·P_i - the reference test's own checker (multiexp.rs:337-350) */
static void CV(naive_multiexp)(const CV(aff_t) *bases, const uint64_t *scalars, size_t n,
                               int threads, CV(proj_t) *out) {
  CV(proj_t) total; CV(proj_identity)(&total);
#pragma omp parallel num_threads(threads)
  {
    CV(proj_t) loc; CV(proj_identity)(&loc);
#pragma omp for schedule(static)
    for (size_t i = 0; i < n; i++) {
      CV(proj_t) p, t; CV(from_affine)(&p, &bases[i]);
      CV(mul)(&t, &p, scalars + 4 * i);
      CV(add)(&loc, &loc, &t);
    }
#pragma omp critical
    CV(add)(&total, &total, &loc);
  }
  *out = total;
}

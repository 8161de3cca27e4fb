// Host side of groth16::create_proof over the C ABI (see groth16.hpp for the reference map).
#include "groth16.hpp"
#include "host_fp.hpp"

#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <memory>
#include <unordered_map>

#define BH_TRACE(...) do { if (getenv("BH_DEBUG")) { fprintf(stderr, "[groth16 %.2f ms] ", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count()); fprintf(stderr, __VA_ARGS__); fputc(10, stderr); fflush(stderr); } } while (0)

namespace bellman {
namespace {
typedef unsigned __int128 u128;
const uint64_t FR_MOD[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
const uint64_t FR_INV = 0xfffffffeffffffffULL;
const uint64_t FR_R[4] = {0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL};
const uint64_t FR_R2[4] = {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL};

inline bool geq_mod(const uint64_t *a) {
  for (int i = 3; i >= 0; i--) {
    if (a[i] > FR_MOD[i]) return true;
    if (a[i] < FR_MOD[i]) return false;
  }
  return true;
}
inline void sub_mod(uint64_t *a) {
  u128 br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a[i] - FR_MOD[i] - (uint64_t)br;
    a[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
}
// 4x64 CIOS Montgomery product, fully unrolled (synthesis is the serial part of create_proof)
__attribute__((always_inline)) inline void mont_mul(uint64_t *r, const uint64_t *a, const uint64_t *b) {
  uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
#define BH_ROW(bi)                                                                        \
  {                                                                                       \
    u128 c = (u128)a[0] * (bi) + t0; t0 = (uint64_t)c; c >>= 64;                          \
    c += (u128)a[1] * (bi) + t1; t1 = (uint64_t)c; c >>= 64;                              \
    c += (u128)a[2] * (bi) + t2; t2 = (uint64_t)c; c >>= 64;                              \
    c += (u128)a[3] * (bi) + t3; t3 = (uint64_t)c; c >>= 64;                              \
    c += t4; t4 = (uint64_t)c; const uint64_t t5 = (uint64_t)(c >> 64);                   \
    const uint64_t m = t0 * FR_INV;                                                       \
    c = ((u128)m * FR_MOD[0] + t0) >> 64;                                                 \
    c += (u128)m * FR_MOD[1] + t1; t0 = (uint64_t)c; c >>= 64;                            \
    c += (u128)m * FR_MOD[2] + t2; t1 = (uint64_t)c; c >>= 64;                            \
    c += (u128)m * FR_MOD[3] + t3; t2 = (uint64_t)c; c >>= 64;                            \
    c += t4; t3 = (uint64_t)c; t4 = t5 + (uint64_t)(c >> 64);                             \
  }
  BH_ROW(b[0]) BH_ROW(b[1]) BH_ROW(b[2]) BH_ROW(b[3])
#undef BH_ROW
  uint64_t t[4] = {t0, t1, t2, t3};
  if (t4 || geq_mod(t)) sub_mod(t);
  r[0] = t[0]; r[1] = t[1]; r[2] = t[2]; r[3] = t[3];
}
}  // namespace

Fr Fr::zero() { Fr r; memset(r.l, 0, sizeof r.l); return r; }
Fr Fr::one() { Fr r; memcpy(r.l, FR_R, sizeof FR_R); return r; }
Fr Fr::from_u64(uint64_t v) {
  uint64_t c[4] = {v, 0, 0, 0};
  Fr r;
  mont_mul(r.l, c, FR_R2);
  return r;
}
Fr Fr::operator+(const Fr &o) const {
  Fr r;
  u128 c = 0;
  for (int i = 0; i < 4; i++) {
    c += (u128)l[i] + o.l[i];
    r.l[i] = (uint64_t)c;
    c >>= 64;
  }
  if (geq_mod(r.l)) sub_mod(r.l);
  return r;
}
Fr Fr::operator-(const Fr &o) const {
  Fr r;
  u128 br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)l[i] - o.l[i] - (uint64_t)br;
    r.l[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
  if (br) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (u128)r.l[i] + FR_MOD[i];
      r.l[i] = (uint64_t)c;
      c >>= 64;
    }
  }
  return r;
}
Fr Fr::operator*(const Fr &o) const { Fr r; mont_mul(r.l, l, o.l); return r; }
Fr Fr::neg() const { return Fr::zero() - *this; }
void Fr::to_canonical(uint64_t out[4]) const {
  const uint64_t one[4] = {1, 0, 0, 0};
  mont_mul(out, l, one);
}
// little-endian 512-bit integer -> Fr (ff's wide reduction behind Field::random): lo + hi * 2^256 mod q
Fr Fr::from_u512(const uint64_t limbs[8]) {
  Fr lo, hi, r2;
  memcpy(lo.l, limbs, 32);
  memcpy(hi.l, limbs + 4, 32);
  memcpy(r2.l, FR_R2, 32);
  // mont_mul(x, R^2) = x * R mod q for any 256-bit x: the Montgomery form of x mod q
  const Fr lo_m = lo * r2, hi_m = hi * r2;
  return lo_m + hi_m * r2;   // Montgomery form of 2^256 is R * R = R^2
}
Fr Fr::pow_vartime(uint64_t e) const {
  Fr acc = Fr::one();
  for (int i = 63; i >= 0; i--) {
    acc = acc * acc;
    if ((e >> i) & 1) acc = acc * *this;
  }
  return acc;
}
Fr Fr::invert() const {   // a^(q-2)
  uint64_t e[4] = {FR_MOD[0] - 2, FR_MOD[1], FR_MOD[2], FR_MOD[3]};   // no borrow: the low limb ends in ...00000001 + 0xffffffff00000000
  Fr acc = Fr::one();
  for (int i = 255; i >= 0; i--) {
    acc = acc * acc;
    if ((e[i >> 6] >> (i & 63)) & 1) acc = acc * *this;
  }
  return acc;
}
}  // namespace bellman

namespace groth16 {
using namespace bellman;


bool G1Affine::is_identity() const { uint64_t o = 0; for (uint64_t x : v) o |= x; return o == 0; }
bool G2Affine::is_identity() const { uint64_t o = 0; for (uint64_t x : v) o |= x; return o == 0; }

static void check(int rc) {
  switch (rc) {
    case BH_OK: return;
    case BH_ERR_UNEXPECTED_IDENTITY: throw SynthesisError(rc, "UnexpectedIdentity");
    case BH_ERR_UNEXPECTED_EOF: throw SynthesisError(rc, "IoError(UnexpectedEof): expected more bases from source");
    case BH_ERR_DEGREE_TOO_LARGE: throw SynthesisError(rc, "PolynomialDegreeTooLarge");
    default: throw std::runtime_error("bellman_hip: HIP/runtime failure (no CPU fallback)");
  }
}

namespace {
struct DevBuf {
  bh_ctx *ctx;
  void *p = nullptr;
  DevBuf(bh_ctx *c, size_t bytes) : ctx(c) { check(bh_dev_alloc(ctx, bytes, &p)); }
  ~DevBuf() { if (p) bh_dev_free(ctx, p); }
  DevBuf(const DevBuf &) = delete;
};
}  // namespace

Parameters::Parameters(bh_ctx *c, const VerifyingKey &k, const G1Affine *hq, size_t nh, const G1Affine *lq, size_t nl,
                       const G1Affine *aq, size_t na, const G1Affine *b1, size_t nb1, const G2Affine *b2, size_t nb2)
    : ctx(c), vk(k) {
  check(bh_bases_register(ctx, BH_G1, hq, nh, 96, -1, &h));
  check(bh_bases_register(ctx, BH_G1, lq, nl, 96, -1, &l));
  check(bh_bases_register(ctx, BH_G1, aq, na, 96, -1, &a));
  check(bh_bases_register(ctx, BH_G1, b1, nb1, 96, -1, &b_g1));
  check(bh_bases_register(ctx, BH_G2, b2, nb2, 192, -1, &b_g2));
}
// ---- groth16/src/lib.rs:159-215 (VerifyingKey::read) + :289-398 (Parameters::read) --------------------
namespace {
struct ByteReader {
  const unsigned char *p;
  size_t len, pos = 0;
  size_t remaining() const { return len - pos; }
};
[[noreturn]] void throw_io(int rc, int group) {
  switch (rc) {
    case BH_ERR_UNEXPECTED_EOF: throw bellman::IoError(rc, "failed to fill whole buffer");
    case BH_ERR_INVALID_POINT: throw bellman::IoError(rc, group == BH_G1 ? "invalid G1" : "invalid G2");
    case BH_ERR_POINT_AT_INFINITY: throw bellman::IoError(rc, "point at infinity");
    default: check(rc); throw std::runtime_error("unreachable");
  }
}
// reads `count` points the way the reference's read loop does: every complete point is validated in
// stream order first; running out of bytes is reported only if all points before the cut are fine
bh_bases *read_points(bh_ctx *ctx, ByteReader &rd, int group, size_t count, unsigned flags) {
  const size_t rec = group == BH_G1 ? 96 : 192;
  const size_t complete = rd.remaining() / rec < count ? rd.remaining() / rec : count;
  bh_bases *b = nullptr;
  const int rc = bh_bases_read_uncompressed(ctx, group, rd.p + rd.pos, complete, flags, &b, nullptr);
  if (rc != BH_OK) throw_io(rc, group);
  if (complete < count) { bh_bases_release(ctx, b); throw_io(BH_ERR_UNEXPECTED_EOF, group); }
  rd.pos += count * rec;
  return b;
}
size_t read_u32_be(ByteReader &rd) {
  if (rd.remaining() < 4) throw_io(BH_ERR_UNEXPECTED_EOF, BH_G1);
  const unsigned char *q = rd.p + rd.pos;
  rd.pos += 4;
  return ((size_t)q[0] << 24) | ((size_t)q[1] << 16) | ((size_t)q[2] << 8) | (size_t)q[3];
}
struct BasesGuard {   // releases what was read so far if a later section throws
  bh_ctx *ctx;
  std::vector<bh_bases *> v;
  ~BasesGuard() { for (bh_bases *b : v) bh_bases_release(ctx, b); }
  bh_bases *keep(bh_bases *b) { v.push_back(b); return b; }
};
}  // namespace

Parameters::Parameters(bh_ctx *c, const void *bytes, size_t len, bool checked) : ctx(c) {
  ByteReader rd{(const unsigned char *)bytes, len};
  BasesGuard guard{ctx, {}};
  // verifying key: always from_uncompressed (lib.rs:160-185); identity allowed except in ic (:199-207)
  auto vk_point = [&](int group, void *out) {
    bh_bases *b = guard.keep(read_points(ctx, rd, group, 1, BH_POINTS_CHECKED));
    check(bh_bases_download(ctx, b, 0, 1, out));
  };
  vk_point(BH_G1, &vk.alpha_g1); vk_point(BH_G1, &vk.beta_g1); vk_point(BH_G2, &vk.beta_g2);
  vk_point(BH_G2, &vk.gamma_g2); vk_point(BH_G1, &vk.delta_g1); vk_point(BH_G2, &vk.delta_g2);
  {
    const size_t n_ic = read_u32_be(rd);
    bh_bases *ic = guard.keep(read_points(ctx, rd, BH_G1, n_ic, BH_POINTS_CHECKED | BH_POINTS_FORBID_IDENTITY));
    vk.ic.resize(n_ic);
    check(bh_bases_download(ctx, ic, 0, n_ic, vk.ic.data()));
  }
  const unsigned qflags = (checked ? BH_POINTS_CHECKED : 0u) | BH_POINTS_FORBID_IDENTITY;   // lib.rs:294-315
  bh_bases *hq = guard.keep(read_points(ctx, rd, BH_G1, read_u32_be(rd), qflags));
  bh_bases *lq = guard.keep(read_points(ctx, rd, BH_G1, read_u32_be(rd), qflags));
  bh_bases *aq = guard.keep(read_points(ctx, rd, BH_G1, read_u32_be(rd), qflags));
  bh_bases *b1 = guard.keep(read_points(ctx, rd, BH_G1, read_u32_be(rd), qflags));
  bh_bases *b2 = guard.keep(read_points(ctx, rd, BH_G2, read_u32_be(rd), qflags));
  h = hq; l = lq; a = aq; b_g1 = b1; b_g2 = b2;
  // the five query vectors now belong to this object; the verifying-key scratch handles are released
  std::vector<bh_bases *> scratch;
  for (bh_bases *b : guard.v)
    if (b != h && b != l && b != a && b != b_g1 && b != b_g2) scratch.push_back(b);
  guard.v.swap(scratch);
}

// ---- groth16/src/generator.rs:163-510 ------------------------------------------------------------------
namespace {
template <class A>
std::vector<A> download_points(bh_ctx *ctx, const void *dev, size_t n) {
  std::vector<A> v(n);
  if (n) check(bh_dev_download(ctx, v.data(), dev, n * sizeof(A)));
  return v;
}
template <class A>
void drop_identities(std::vector<A> &v) {   // generator.rs:491-505
  size_t k = 0;
  for (size_t i = 0; i < v.size(); i++)
    if (!v[i].is_identity()) v[k++] = v[i];
  v.resize(k);
}
}  // namespace

Parameters::Parameters(bh_ctx *c, R1cs &r1cs, const G1Affine &g1, const G2Affine &g2, const Fr &alpha, const Fr &beta,
                       const Fr &gamma, const Fr &delta, const Fr &tau)
    : ctx(c) {
  if (gamma.is_zero() || delta.is_zero())   // generator.rs:227-243
    throw SynthesisError(BH_ERR_UNEXPECTED_IDENTITY, "UnexpectedIdentity");
  const Fr gamma_inv = gamma.invert(), delta_inv = delta.invert();
  const size_t n_cons = r1cs.num_constraints, n_in = r1cs.num_inputs, n_vars = r1cs.num_inputs + r1cs.num_aux;
  uint32_t log_m = 0;
  size_t m = 1;
  while (m < n_cons) {   // EvaluationDomain::from_coeffs, generator.rs:204-205
    m *= 2;
    log_m++;
    if (log_m >= 32) throw SynthesisError(BH_ERR_DEGREE_TOO_LARGE, "PolynomialDegreeTooLarge");
  }
  const Fr one = Fr::one();
  // h query: g1^(tau^i * t(tau) / delta), i < m - 1                                      generator.rs:247-296
  const Fr coeff = (tau.pow_vartime(m) - one) * delta_inv;
  DevBuf d_tau(ctx, m * 32), d_hs(ctx, m * 32), d_h(ctx, m * 96);
  check(bh_fr_powers_dev(ctx, d_tau.p, m, &tau, &one, nullptr));
  check(bh_fr_powers_dev(ctx, d_hs.p, m - 1, &tau, &coeff, nullptr));
  check(bh_fixed_base_mul_dev(ctx, BH_G1, &g1, d_hs.p, m - 1, BH_SCALARS_MONT, d_h.p, nullptr));
  // Lagrange coefficients of tau, then the QAP polynomials of every variable at tau       :299-387
  check(bh_fft_fr_dev(ctx, d_tau.p, log_m, BH_IFFT, nullptr));
  DevBuf d_at(ctx, n_vars * 32 + 32), d_bt(ctx, n_vars * 32 + 32), d_ct(ctx, n_vars * 32 + 32), d_e(ctx, n_vars * 32 + 32);
  check(bh_r1cs_eval_transposed_dev(ctx, r1cs.handle, d_tau.p, d_at.p, d_bt.p, d_ct.p, nullptr));
  check(bh_fr_qap_ext_dev(ctx, d_e.p, d_at.p, d_bt.p, d_ct.p, n_in, n_vars, &alpha, &beta, &gamma_inv, &delta_inv, nullptr));
  // a = g1^at, b = g1^bt / g2^bt, ext = g1^e (a zero scalar gives the identity, :389-397)   :389-409
  DevBuf d_a(ctx, n_vars * 96 + 96), d_b1(ctx, n_vars * 96 + 96), d_b2(ctx, n_vars * 192 + 192), d_ext(ctx, n_vars * 96 + 96);
  check(bh_fixed_base_mul_dev(ctx, BH_G1, &g1, d_at.p, n_vars, BH_SCALARS_MONT, d_a.p, nullptr));
  check(bh_fixed_base_mul_dev(ctx, BH_G1, &g1, d_bt.p, n_vars, BH_SCALARS_MONT, d_b1.p, nullptr));
  check(bh_fixed_base_mul_dev(ctx, BH_G2, &g2, d_bt.p, n_vars, BH_SCALARS_MONT, d_b2.p, nullptr));
  check(bh_fixed_base_mul_dev(ctx, BH_G1, &g1, d_e.p, n_vars, BH_SCALARS_MONT, d_ext.p, nullptr));
  check(bh_ctx_synchronize(ctx));
  std::vector<G1Affine> av = download_points<G1Affine>(ctx, d_a.p, n_vars), b1v = download_points<G1Affine>(ctx, d_b1.p, n_vars),
                        ext = download_points<G1Affine>(ctx, d_ext.p, n_vars);
  std::vector<G2Affine> b2v = download_points<G2Affine>(ctx, d_b2.p, n_vars);
  for (size_t i = n_in; i < n_vars; i++)                       // :464-470
    if (ext[i].is_identity()) throw SynthesisError(BH_ERR_UNCONSTRAINED_VARIABLE, "UnconstrainedVariable");
  vk.ic.assign(ext.begin(), ext.begin() + n_in);
  auto scalar_mul1 = [](const G1Affine &p, const Fr &k) { uint64_t kc[4]; k.to_canonical(kc); G1Affine r; bh_point_mul(BH_G1, &r, &p, kc); return r; };
  auto scalar_mul2 = [](const G2Affine &p, const Fr &k) { uint64_t kc[4]; k.to_canonical(kc); G2Affine r; bh_point_mul(BH_G2, &r, &p, kc); return r; };
  vk.alpha_g1 = scalar_mul1(g1, alpha); vk.beta_g1 = scalar_mul1(g1, beta); vk.beta_g2 = scalar_mul2(g2, beta);   // :475-484
  vk.gamma_g2 = scalar_mul2(g2, gamma); vk.delta_g1 = scalar_mul1(g1, delta); vk.delta_g2 = scalar_mul2(g2, delta);
  drop_identities(av); drop_identities(b1v); drop_identities(b2v);
  BasesGuard guard{ctx, {}};
  bh_bases *hq = nullptr, *lq = nullptr, *aq = nullptr, *b1q = nullptr, *b2q = nullptr;
  check(bh_bases_copy_dev(ctx, BH_G1, d_h.p, m - 1, &hq)); guard.keep(hq);
  check(bh_bases_register(ctx, BH_G1, ext.data() + n_in, n_vars - n_in, 96, -1, &lq)); guard.keep(lq);
  check(bh_bases_register(ctx, BH_G1, av.data(), av.size(), 96, -1, &aq)); guard.keep(aq);
  check(bh_bases_register(ctx, BH_G1, b1v.data(), b1v.size(), 96, -1, &b1q)); guard.keep(b1q);
  check(bh_bases_register(ctx, BH_G2, b2v.data(), b2v.size(), 192, -1, &b2q)); guard.keep(b2q);
  h = hq; l = lq; a = aq; b_g1 = b1q; b_g2 = b2q;
  guard.v.clear();
}

// ---- groth16/src/lib.rs:143-156 + :258-287 (VerifyingKey::write, Parameters::write) -------------------
namespace {
void fp_to_be48(unsigned char *out, const uint64_t mont[6], bool *lex_largest, bool *is_zero);
void put_g1(std::vector<unsigned char> &o, const G1Affine &p) {
  const size_t at = o.size();
  o.resize(at + 96, 0);
  if (p.is_identity()) { o[at] = 0x40; return; }
  bool x, y;
  fp_to_be48(&o[at], p.v, &x, &y);
  fp_to_be48(&o[at + 48], p.v + 6, &x, &y);
}
void put_g2(std::vector<unsigned char> &o, const G2Affine &p) {
  const size_t at = o.size();
  o.resize(at + 192, 0);
  if (p.is_identity()) { o[at] = 0x40; return; }
  bool x, y;
  fp_to_be48(&o[at], p.v + 6, &x, &y);          // x.c1
  fp_to_be48(&o[at + 48], p.v, &x, &y);         // x.c0
  fp_to_be48(&o[at + 96], p.v + 18, &x, &y);    // y.c1
  fp_to_be48(&o[at + 144], p.v + 12, &x, &y);   // y.c0
}
void put_u32(std::vector<unsigned char> &o, size_t v) {
  for (int s = 24; s >= 0; s -= 8) o.push_back((unsigned char)(v >> s));
}
}  // namespace

std::vector<unsigned char> Parameters::write() const {
  std::vector<unsigned char> o;
  put_g1(o, vk.alpha_g1); put_g1(o, vk.beta_g1); put_g2(o, vk.beta_g2); put_g2(o, vk.gamma_g2);
  put_g1(o, vk.delta_g1); put_g2(o, vk.delta_g2);
  put_u32(o, vk.ic.size());
  for (const G1Affine &p : vk.ic) put_g1(o, p);
  const bh_bases *qs[5] = {h, l, a, b_g1, b_g2};
  for (int q = 0; q < 5; q++) {
    const size_t n = bh_bases_len(qs[q]);
    put_u32(o, n);
    if (q < 4) {
      std::vector<G1Affine> v(n);
      check(bh_bases_download(ctx, qs[q], 0, n, v.data()));
      o.reserve(o.size() + n * 96);
      for (const G1Affine &p : v) put_g1(o, p);
    } else {
      std::vector<G2Affine> v(n);
      check(bh_bases_download(ctx, qs[q], 0, n, v.data()));
      o.reserve(o.size() + n * 192);
      for (const G2Affine &p : v) put_g2(o, p);
    }
  }
  return o;
}

// ---- groth16/src/lib.rs:38-46 (Proof::write): Zcash compressed encoding ------------------------------
namespace {
void fp_to_be48(unsigned char *out, const uint64_t mont[6], bool *lex_largest, bool *is_zero) {
  bh::hfp_t a, one_raw, c;
  memcpy(a.l, mont, 48);
  memset(one_raw.l, 0, 48);
  one_raw.l[0] = 1;
  bh::hostfp::mul(c, a, one_raw);   // Montgomery -> canonical
  for (int i = 0; i < 6; i++)
    for (int b = 0; b < 8; b++) out[47 - (8 * i + b)] = (unsigned char)(c.l[i] >> (8 * b));
  // y > (p - 1) / 2   <=>   2y > p - 1   <=>   2y >= p + 1 > p  (p odd)
  uint64_t d[7];
  uint64_t carry = 0;
  for (int i = 0; i < 6; i++) { d[i] = (c.l[i] << 1) | carry; carry = c.l[i] >> 63; }
  d[6] = carry;
  bool gt = d[6] != 0;
  if (!gt) {
    gt = false;
    for (int i = 5; i >= 0; i--) {
      if (d[i] != bh::hostfp::MOD[i]) { gt = d[i] > bh::hostfp::MOD[i]; break; }
    }
  }
  *lex_largest = gt;
  *is_zero = (c.l[0] | c.l[1] | c.l[2] | c.l[3] | c.l[4] | c.l[5]) == 0;
}
}  // namespace

void Proof::write(unsigned char out[192]) const {
  auto g1 = [](const G1Affine &p, unsigned char *o) {
    if (p.is_identity()) { memset(o, 0, 48); o[0] = 0xC0; return; }
    bool ly, zy, lx, zx;
    unsigned char ybuf[48];
    fp_to_be48(o, p.v, &lx, &zx);
    fp_to_be48(ybuf, p.v + 6, &ly, &zy);
    o[0] |= 0x80 | (ly ? 0x20 : 0);
  };
  g1(a, out);
  if (b.is_identity()) { memset(out + 48, 0, 96); out[48] = 0xC0; }
  else {
    // x.c1 | x.c0 ; sort flag from y: compare c1 first, c0 only when c1 = 0
    bool l0, z0, l1, z1, t0, t1;
    unsigned char y0[48], y1[48];
    fp_to_be48(out + 48 + 48, b.v, &t0, &t1);        // x.c0 second
    fp_to_be48(out + 48, b.v + 6, &t0, &t1);         // x.c1 first
    fp_to_be48(y0, b.v + 12, &l0, &z0);
    fp_to_be48(y1, b.v + 18, &l1, &z1);
    const bool largest = z1 ? l0 : l1;
    out[48] |= 0x80 | (largest ? 0x20 : 0);
  }
  g1(c, out + 144);
}

Parameters::~Parameters() {
  bh_bases_release(ctx, h); bh_bases_release(ctx, l); bh_bases_release(ctx, a);
  bh_bases_release(ctx, b_g1); bh_bases_release(ctx, b_g2);
}

// ---- prover.rs:19-55 ------------------------------------------------------------------------------
static Fr eval(const LinearCombination &lc, DensityTracker *input_density, DensityTracker *aux_density,
               const std::vector<Fr> &input_assignment, const std::vector<Fr> &aux_assignment) {
  Fr acc = Fr::zero();
  const Fr one = Fr::one();
  for (size_t t = 0; t < lc.size(); t++) {
    const Variable &var = lc[t].first;
    const Fr &coeff = lc[t].second;
    if (coeff.is_zero()) continue;          // zero coefficients count for neither value nor density (:31)
    Fr tmp;
    if (var.kind == Index::Input) {
      tmp = input_assignment[var.idx];
      if (input_density) input_density->inc(var.idx);
    } else {
      tmp = aux_assignment[var.idx];
      if (aux_density) aux_density->inc(var.idx);
    }
    if (tmp == one) tmp = coeff;            // 1 * coeff (the ubiquitous `(c, CS::one())` terms)
    else if (coeff != one) tmp = tmp * coeff;
    acc = acc + tmp;
  }
  return acc;
}

// ---- prover.rs:73-162 -----------------------------------------------------------------------------
Variable ProvingAssignment::alloc(ValueFn f) {
  aux_assignment.push_back(f());
  a_aux_density.add_element();
  b_aux_density.add_element();
  return Variable::new_unchecked(Index::Aux, aux_assignment.size() - 1);
}
Variable ProvingAssignment::alloc_input(ValueFn f) {
  input_assignment.push_back(f());
  b_input_density.add_element();
  return Variable::new_unchecked(Index::Input, input_assignment.size() - 1);
}
void ProvingAssignment::enforce(LcFn fa, LcFn fb, LcFn fc) {
  const LinearCombination la = fa(LinearCombination::zero()), lb = fb(LinearCombination::zero()),
                          lc = fc(LinearCombination::zero());
  // inputs have full density in the A query; there is no C query (prover.rs:119-141)
  a.push_back(eval(la, nullptr, &a_aux_density, input_assignment, aux_assignment));
  b.push_back(eval(lb, &b_input_density, &b_aux_density, input_assignment, aux_assignment));
  c.push_back(eval(lc, nullptr, nullptr, input_assignment, aux_assignment));
}

namespace {
struct ProofStream {   // uploads + h block of one proof; independent of other proofs in flight
  bh_ctx *ctx;
  void *st = nullptr;
  explicit ProofStream(bh_ctx *c) : ctx(c) { check(bh_stream_create(ctx, &st)); }
  ~ProofStream() { if (st) { (void)bh_stream_synchronize(ctx, st); (void)bh_stream_destroy(ctx, st); } }
  ProofStream(const ProofStream &) = delete;
};
// Every issued multiexp owns device buffers and reads ours: if anything throws between issue and
// wait, the jobs still in flight are drained before the DevBufs they read are released.
struct JobSet {
  std::vector<bh_msm_job **> slots;
  void track(bh_msm_job **j) { slots.push_back(j); }
  int wait(bh_msm_job *&j, void *out) {
    bh_msm_job *job = j;
    j = nullptr;
    return bh_msm_wait(job, out);
  }
  ~JobSet() {
    unsigned char sink[192];
    for (bh_msm_job **s : slots)
      if (*s) { (void)bh_msm_wait(*s, sink); *s = nullptr; }
  }
};
double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
template <class A> A add_pts(int group, const A &x, const A &y) { A r; bh_point_add(group, &r, &x, &y, 1); return r; }
template <class A> A mul_pt(int group, const A &x, const Fr &k) {
  uint64_t kc[4];
  k.to_canonical(kc);
  A r;
  bh_point_mul(group, &r, &x, kc);
  return r;
}
}  // namespace

// ---- prover.rs:217-360 ----------------------------------------------------------------------------
namespace {
// What differs between the two ways of getting the constraint evaluations into HBM
struct AssignmentSource {
  const Fr *inputs; size_t n_in;
  const Fr *aux; size_t n_aux;
  size_t n_cons;
  // host path (prove_assignment): evaluations and density bitmaps computed during synthesis
  const ProvingAssignment *host = nullptr;
  // device path (prove_witness): matrices and densities already resident
  const R1cs *r1cs = nullptr;
};
}  // namespace

namespace {
// The slice of an n-term multiexp that part `k` of `parts` computes when one proof is spread over
// several GPUs: contiguous in the scalar index, cut at multiples of 64 so that the
// density bitmap of the slice starts on a word.  parts = 1 gives [0, n).
struct Slice { size_t lo, hi; };
Slice slice_of(size_t n, size_t part, size_t parts) {
  auto cut = [&](size_t k) {
    if (k >= parts) return n;
    const size_t c = (size_t)((unsigned __int128)n * k / parts) & ~size_t(63);
    return c < n ? c : n;
  };
  return Slice{cut(part), cut(part + 1)};
}
size_t popcount_prefix(const uint64_t *words, size_t bits) {   // bits is a multiple of 64
  size_t t = 0;
  for (size_t w = 0; w < bits / 64; w++) t += (size_t)__builtin_popcountll(words[w]);
  return t;
}
}  // namespace

// prover.rs:217-318 + the waits of :339-354: the eight multiexp results (of this part's slices)
static void msm_sums(const AssignmentSource &src, Parameters &params, size_t part, size_t parts, MsmSums &out,
                     ProveTimings *tm) {
  bh_ctx *ctx = params.ctx;
  const double t0 = now_ms();
  const size_t n_cons = src.n_cons;
  // EvaluationDomain::from_coeffs (domain.rs:47-79)
  uint32_t log_m = 0;
  size_t m = 1;
  while (m < n_cons) {
    m *= 2;
    log_m++;
    if (log_m >= 32) throw SynthesisError(BH_ERR_DEGREE_TOO_LARGE, "PolynomialDegreeTooLarge");
  }
  // assignments: uploaded once, shared by seven multiexps (prover.rs:248-318)
  const size_t n_in = src.n_in, n_aux = src.n_aux;
  DevBuf d_in(ctx, n_in * 32 + 32), d_aux(ctx, n_aux * 32 + 32);
  ProofStream ps(ctx);
  check(bh_dev_upload_on(ctx, d_in.p, src.inputs, n_in * 32, ps.st));
  if (n_aux) check(bh_dev_upload_on(ctx, d_aux.p, src.aux, n_aux * 32, ps.st));
  std::unique_ptr<DevBuf> dens_buf[3];
  const uint64_t *dens_a_aux = nullptr, *dens_b_in = nullptr, *dens_b_aux = nullptr;
  const uint64_t *hw_a_aux = nullptr, *hw_b_in = nullptr, *hw_b_aux = nullptr;   // the same bitmaps on the host
  size_t b_in_total = 0;
  if (src.host) {
    auto upload_density = [&](const DensityTracker &d, std::unique_ptr<DevBuf> &buf) {
      const size_t nw = (d.get_query_size() + 63) / 64;
      buf.reset(new DevBuf(ctx, nw * 8 + 8));
      if (nw) check(bh_dev_upload_on(ctx, buf->p, d.words(), nw * 8, ps.st));
      return (const uint64_t *)buf->p;
    };
    dens_a_aux = upload_density(src.host->a_aux_density, dens_buf[0]);
    dens_b_in = upload_density(src.host->b_input_density, dens_buf[1]);
    dens_b_aux = upload_density(src.host->b_aux_density, dens_buf[2]);
    b_in_total = src.host->b_input_density.get_total_density();
    hw_a_aux = src.host->a_aux_density.words(); hw_b_in = src.host->b_input_density.words();
    hw_b_aux = src.host->b_aux_density.words();
  } else {
    check(bh_r1cs_density(src.r1cs->handle, 0, &dens_a_aux, &hw_a_aux, nullptr));
    check(bh_r1cs_density(src.r1cs->handle, 1, &dens_b_in, &hw_b_in, &b_in_total));
    check(bh_r1cs_density(src.r1cs->handle, 2, &dens_b_aux, &hw_b_aux, nullptr));
  }

  BH_TRACE("prove_core start: uploads queued");
  check(bh_stream_synchronize(ctx, ps.st));   // the multiexp jobs run on their own streams
  BH_TRACE("assignment resident");
  bh_msm_job *l_job = nullptr, *a_in_job = nullptr, *a_aux_job = nullptr, *b1_in_job = nullptr, *b1_aux_job = nullptr,
             *b2_in_job = nullptr, *b2_aux_job = nullptr, *h_job = nullptr;
  // h-block buffers are declared here so that `jobs` (declared after every buffer a job reads) is
  // destroyed first and drains whatever is still in flight if an exception unwinds this frame
  DevBuf da(ctx, m * 32), db(ctx, m * 32), dc(ctx, m * 32);
  JobSet jobs;
  for (bh_msm_job **j : {&l_job, &a_in_job, &a_aux_job, &b1_in_job, &b1_aux_job, &b2_in_job, &b2_aux_job, &h_job}) jobs.track(j);
  // one multiexp over this part's slice of the scalars: `skip` advances by the number of bases the
  // skipped scalars would have consumed (all of them without a density map, the set bits with one)
  auto issue = [&](bh_bases *bases, size_t skip, const void *scalars, size_t n, const uint64_t *dens_dev,
                   const uint64_t *dens_host, bh_msm_job **job) {
    const Slice sl = slice_of(n, part, parts);
    const size_t base_skip = skip + (dens_dev ? popcount_prefix(dens_host, sl.lo) : sl.lo);
    check(bh_msm_async_dev(ctx, bases, base_skip, (const char *)scalars + sl.lo * 32, sl.hi - sl.lo, BH_SCALARS_MONT,
                           dens_dev ? dens_dev + sl.lo / 64 : nullptr, dens_dev ? sl.hi - sl.lo : 0, job));
  };
  issue(params.l, 0, d_aux.p, n_aux, nullptr, nullptr, &l_job);
  // get_a(num_inputs, _) -> ((a,0),(a,num_inputs))            groth16/src/lib.rs:451-457
  issue(params.a, 0, d_in.p, n_in, nullptr, nullptr, &a_in_job);
  issue(params.a, n_in, d_aux.p, n_aux, dens_a_aux, hw_a_aux, &a_aux_job);
  // get_b_g1/g2(b_input_density_total, _) -> ((b,0),(b,total))   groth16/src/lib.rs:459-473
  issue(params.b_g1, 0, d_in.p, n_in, dens_b_in, hw_b_in, &b1_in_job);
  issue(params.b_g1, b_in_total, d_aux.p, n_aux, dens_b_aux, hw_b_aux, &b1_aux_job);
  issue(params.b_g2, 0, d_in.p, n_in, dens_b_in, hw_b_in, &b2_in_job);
  issue(params.b_g2, b_in_total, d_aux.p, n_aux, dens_b_aux, hw_b_aux, &b2_aux_job);

  // The seven multiexps above only need the assignments, so they are already running on their own
  // streams while the h block below uploads a/b/c and runs its FFTs (the reference issues h first,
  // prover.rs:221-245; the order of issue is unobservable, the order of waits is kept).
  // h block (prover.rs:221-245): a, b, c stay in HBM; the quotient's coefficients are consumed by
  // the H multiexp straight from device memory (no host round trip, no serial Fr -> Exponent pass).
  if (src.host) {
    // EvaluationDomain::from_coeffs pads with zeros (domain.rs:68): the padding is written on the device
    const std::vector<Fr> *ev[3] = {&src.host->a, &src.host->b, &src.host->c};
    void *dst[3] = {da.p, db.p, dc.p};
    for (int i = 0; i < 3; i++) {
      if (m > n_cons) check(bh_dev_zero_on(ctx, (char *)dst[i] + n_cons * 32, (m - n_cons) * 32, ps.st));
      check(bh_dev_upload_on(ctx, dst[i], ev[i]->data(), n_cons * 32, ps.st));
    }
  } else {
    // a = A.w, b = B.w, c = C.w straight into the FFT buffers (prover.rs:19-55,105-145 on the device)
    check(bh_r1cs_eval_dev(ctx, src.r1cs->handle, d_in.p, d_aux.p, da.p, db.p, dc.p, log_m, ps.st));
  }
  BH_TRACE("7 multiexps issued; n_cons=%zu m=%zu a/b/c queued", n_cons, m);
  check(bh_h_poly_fr_dev(ctx, da.p, db.p, dc.p, log_m, ps.st));   // synchronises ps.st before returning
  BH_TRACE("h poly done");
  const double t1 = now_ms();
  issue(params.h, 0, da.p, m - 1, nullptr, nullptr, &h_job);   // a.len() - 1, :238-244

  BH_TRACE("all msm issued n_in=%zu n_aux=%zu", n_in, n_aux);
  // every job must be waited on (it owns device resources), even when an earlier one fails
  int rcs[8];
  // prover.rs:339-354 waits in this order: a_inputs, a_aux, b_g1_inputs, b_g1_aux, b_g2_inputs, b_g2_aux, h, l
  rcs[0] = jobs.wait(a_in_job, &out.a_in);
  rcs[1] = jobs.wait(a_aux_job, &out.a_aux);
  rcs[2] = jobs.wait(b1_in_job, &out.b1_in);
  rcs[3] = jobs.wait(b1_aux_job, &out.b1_aux);
  rcs[4] = jobs.wait(b2_in_job, &out.b2_in);
  rcs[5] = jobs.wait(b2_aux_job, &out.b2_aux);
  rcs[6] = jobs.wait(h_job, &out.h);
  rcs[7] = jobs.wait(l_job, &out.l);
  const double t2 = now_ms();
  BH_TRACE("waits done rc=%d %d %d %d %d %d %d %d", rcs[0], rcs[1], rcs[2], rcs[3], rcs[4], rcs[5], rcs[6], rcs[7]);
  if (params.vk.delta_g1.is_identity() || params.vk.delta_g2.is_identity())   // subversion check, prover.rs:320-324
    throw SynthesisError(BH_ERR_UNEXPECTED_IDENTITY, "UnexpectedIdentity");
  for (int i = 0; i < 8; i++) check(rcs[i]);                      // first failing `?` in wait order
  if (tm) {
    tm->h_poly_ms = (float)(t1 - t0);
    tm->msm_ms = (float)(t2 - t1);
    tm->total_ms = (float)(now_ms() - t0);
  }
}

// prover.rs:326-360 from the eight multiexp results
Proof assemble_proof(const Parameters &params, const MsmSums &m, const Fr &r, const Fr &s) {
  const VerifyingKey &vk = params.vk;
  if (vk.delta_g1.is_identity() || vk.delta_g2.is_identity())   // subversion check, prover.rs:320-324
    throw SynthesisError(BH_ERR_UNEXPECTED_IDENTITY, "UnexpectedIdentity");
  G1Affine g_a = add_pts(BH_G1, mul_pt(BH_G1, vk.delta_g1, r), vk.alpha_g1);   // :326-327
  G2Affine g_b = add_pts(BH_G2, mul_pt(BH_G2, vk.delta_g2, s), vk.beta_g2);    // :328-329
  const Fr rs = r * s;
  G1Affine g_c = mul_pt(BH_G1, vk.delta_g1, rs);                                // :331-338
  g_c = add_pts(BH_G1, g_c, mul_pt(BH_G1, vk.alpha_g1, s));
  g_c = add_pts(BH_G1, g_c, mul_pt(BH_G1, vk.beta_g1, r));
  G1Affine a_answer = add_pts(BH_G1, m.a_in, m.a_aux);                          // :339-343
  g_a = add_pts(BH_G1, g_a, a_answer);
  a_answer = mul_pt(BH_G1, a_answer, s);
  g_c = add_pts(BH_G1, g_c, a_answer);
  G1Affine b1_answer = add_pts(BH_G1, m.b1_in, m.b1_aux);                       // :345-354
  G2Affine b2_answer = add_pts(BH_G2, m.b2_in, m.b2_aux);
  g_b = add_pts(BH_G2, g_b, b2_answer);
  b1_answer = mul_pt(BH_G1, b1_answer, r);
  g_c = add_pts(BH_G1, g_c, b1_answer);
  g_c = add_pts(BH_G1, g_c, m.h);
  g_c = add_pts(BH_G1, g_c, m.l);
  Proof p;
  p.a = g_a; p.b = g_b; p.c = g_c;
  return p;
}

void MsmSums::add(const MsmSums &o) {
  a_in = add_pts(BH_G1, a_in, o.a_in); a_aux = add_pts(BH_G1, a_aux, o.a_aux);
  b1_in = add_pts(BH_G1, b1_in, o.b1_in); b1_aux = add_pts(BH_G1, b1_aux, o.b1_aux);
  b2_in = add_pts(BH_G2, b2_in, o.b2_in); b2_aux = add_pts(BH_G2, b2_aux, o.b2_aux);
  h = add_pts(BH_G1, h, o.h); l = add_pts(BH_G1, l, o.l);
}

static Proof prove_core(const AssignmentSource &src, Parameters &params, const Fr &r, const Fr &s, ProveTimings *tm) {
  const double t0 = now_ms();
  MsmSums sums;
  msm_sums(src, params, 0, 1, sums, tm);
  Proof p = assemble_proof(params, sums, r, s);
  if (tm) tm->total_ms = (float)(now_ms() - t0);
  return p;
}

MsmSums prove_witness_part(const R1cs &r1cs, Parameters &params, const Fr *inputs, size_t n_inputs, const Fr *aux, size_t n_aux,
                           size_t part, size_t parts, ProveTimings *tm) {
  if (n_inputs != r1cs.num_inputs || n_aux != r1cs.num_aux || parts == 0 || part >= parts)
    throw std::invalid_argument("witness does not have the shape of the captured circuit / bad part");
  AssignmentSource src;
  src.inputs = inputs; src.n_in = n_inputs;
  src.aux = aux; src.n_aux = n_aux;
  src.n_cons = r1cs.num_constraints;
  src.r1cs = &r1cs;
  MsmSums sums;
  msm_sums(src, params, part, parts, sums, tm);
  return sums;
}

Proof prove_assignment(ProvingAssignment &prover, Parameters &params, const Fr &r, const Fr &s, ProveTimings *tm) {
  AssignmentSource src;
  src.inputs = prover.input_assignment.data(); src.n_in = prover.input_assignment.size();
  src.aux = prover.aux_assignment.data(); src.n_aux = prover.aux_assignment.size();
  src.n_cons = prover.a.size();
  src.host = &prover;
  return prove_core(src, params, r, s, tm);
}

Proof prove_witness(const R1cs &r1cs, Parameters &params, const Fr *inputs, size_t n_inputs, const Fr *aux, size_t n_aux,
                    const Fr &r, const Fr &s, ProveTimings *tm) {
  if (n_inputs != r1cs.num_inputs || n_aux != r1cs.num_aux)
    throw std::invalid_argument("witness does not have the shape of the captured circuit");
  AssignmentSource src;
  src.inputs = inputs; src.n_in = n_inputs;
  src.aux = aux; src.n_aux = n_aux;
  src.n_cons = r1cs.num_constraints;
  src.r1cs = &r1cs;
  return prove_core(src, params, r, s, tm);
}

// ---- structure capture (generator.rs:43-131 KeypairAssembly, plus the input rows of prover.rs:208-215)
namespace {
struct FrHash {
  size_t operator()(const Fr &f) const {
    uint64_t h = f.l[0] * 0x9E3779B97F4A7C15ULL;
    h ^= f.l[1] + 0xBF58476D1CE4E5B9ULL + (h << 6) + (h >> 2);
    h ^= f.l[2] + 0x94D049BB133111EBULL + (h << 6) + (h >> 2);
    h ^= f.l[3] + (h << 6) + (h >> 2);
    return (size_t)h;
  }
};
class ShapeAssembly : public ConstraintSystem {
 public:
  size_t num_inputs = 0, num_aux = 0;
  struct Term { Index kind; uint32_t idx, coeff; };
  std::vector<uint32_t> row_ptr[3];
  std::vector<Term> terms[3];
  std::vector<Fr> coeffs;
  std::unordered_map<Fr, uint32_t, FrHash> coeff_index;
  ShapeAssembly() {
    coeffs.push_back(Fr::one());
    coeff_index.emplace(Fr::one(), 0);
    for (auto &rp : row_ptr) rp.push_back(0);
  }
  Variable alloc(ValueFn) override { return Variable::new_unchecked(Index::Aux, num_aux++); }
  Variable alloc_input(ValueFn) override { return Variable::new_unchecked(Index::Input, num_inputs++); }
  void enforce(LcFn fa, LcFn fb, LcFn fc) override {
    const LinearCombination lcs[3] = {fa(LinearCombination::zero()), fb(LinearCombination::zero()),
                                      fc(LinearCombination::zero())};
    for (int m = 0; m < 3; m++) {
      for (size_t i = 0; i < lcs[m].size(); i++) {
        const Variable &v = lcs[m][i].first;
        const Fr &k = lcs[m][i].second;
        if (k.is_zero()) continue;   // prover.rs:31: no value, no density
        auto it = coeff_index.find(k);
        uint32_t ci;
        if (it == coeff_index.end()) {
          ci = (uint32_t)coeffs.size();
          coeffs.push_back(k);
          coeff_index.emplace(k, ci);
        } else {
          ci = it->second;
        }
        terms[m].push_back(Term{v.kind, (uint32_t)v.idx, ci});
      }
      row_ptr[m].push_back((uint32_t)terms[m].size());
    }
  }
};
}  // namespace

R1cs::R1cs(Circuit &shape_of, bh_ctx *ctx) {
  ShapeAssembly cs;
  cs.alloc_input([] { return Fr::one(); });
  shape_of.synthesize(cs);
  for (size_t i = 0; i < cs.num_inputs; i++) {
    cs.enforce([i](LinearCombination lc) { return lc + Variable::new_unchecked(Index::Input, i); },
               [](LinearCombination lc) { return lc; }, [](LinearCombination lc) { return lc; });
  }
  num_inputs = cs.num_inputs; num_aux = cs.num_aux; num_constraints = cs.row_ptr[0].size() - 1;
  std::vector<uint32_t> var[3], coeff[3];
  bh_csr abc[3];
  for (int m = 0; m < 3; m++) {
    var[m].reserve(cs.terms[m].size()); coeff[m].reserve(cs.terms[m].size());
    for (const auto &t : cs.terms[m]) {
      var[m].push_back(t.kind == Index::Input ? t.idx : (uint32_t)(num_inputs + t.idx));   // inputs first, then aux
      coeff[m].push_back(t.coeff);
    }
    abc[m] = bh_csr{cs.row_ptr[m].data(), var[m].data(), coeff[m].data()};
  }
  check(bh_r1cs_create(ctx, num_inputs, num_aux, num_constraints, abc, cs.coeffs.data(), cs.coeffs.size(), &handle));
}
R1cs::R1cs(bh_r1cs *existing) : handle(existing) {
  check(bh_r1cs_shape(existing, &num_inputs, &num_aux, &num_constraints));
}
R1cs::~R1cs() { bh_r1cs_release(handle); }

Variable WitnessAssignment::alloc(ValueFn f) {
  aux_assignment.push_back(f());
  return Variable::new_unchecked(Index::Aux, aux_assignment.size() - 1);
}
Variable WitnessAssignment::alloc_input(ValueFn f) {
  input_assignment.push_back(f());
  return Variable::new_unchecked(Index::Input, input_assignment.size() - 1);
}

Proof create_proof(Circuit &circuit, const R1cs &r1cs, Parameters &params, const Fr &r, const Fr &s, ProveTimings *tm) {
  const double t0 = now_ms();
  WitnessAssignment w;
  w.input_assignment.reserve(r1cs.num_inputs);
  w.aux_assignment.reserve(r1cs.num_aux);
  w.alloc_input([] { return Fr::one(); });
  circuit.synthesize(w);
  const double t1 = now_ms();
  ProveTimings local;
  Proof p = prove_witness(r1cs, params, w.input_assignment.data(), w.input_assignment.size(), w.aux_assignment.data(),
                          w.aux_assignment.size(), r, s, &local);
  if (tm) {
    *tm = local;
    tm->synthesis_ms = (float)(t1 - t0);
    tm->total_ms = (float)(now_ms() - t0);
  }
  return p;
}

// ---- prover.rs:182-215 ----------------------------------------------------------------------------
Proof create_proof(Circuit &circuit, Parameters &params, const Fr &r, const Fr &s, ProveTimings *tm) {
  const double t0 = now_ms();
  ProvingAssignment prover;
  prover.alloc_input([] { return Fr::one(); });
  circuit.synthesize(prover);
  for (size_t i = 0; i < prover.input_assignment.size(); i++) {
    prover.enforce([i](LinearCombination lc) { return lc + Variable::new_unchecked(Index::Input, i); },
                   [](LinearCombination lc) { return lc; }, [](LinearCombination lc) { return lc; });
  }
  const double t1 = now_ms();
  BH_TRACE("synthesised: %zu constraints", prover.a.size());
  ProveTimings local;
  Proof p = prove_assignment(prover, params, r, s, &local);
  if (tm) {
    *tm = local;
    tm->synthesis_ms = (float)(t1 - t0);
    tm->total_ms = (float)(now_ms() - t0);
  }
  return p;
}

// ---------------------------------------------------------------------------------------------------
// demo circuits, written against the mirror exactly like bellman user code
// ---------------------------------------------------------------------------------------------------
// MiMCDemo: /root/reference/groth16/tests/common/mod.rs:37-129 (LongsightF322p3)
class MiMCDemo : public Circuit {
 public:
  Fr xl, xr;
  const Fr *constants;
  size_t rounds;
  void synthesize(ConstraintSystem &cs) override {
    Fr xl_value = xl, xr_value = xr;
    Variable xlv = cs.alloc([&] { return xl_value; });
    Variable xrv = cs.alloc([&] { return xr_value; });
    for (size_t i = 0; i < rounds; i++) {
      const Fr ci = constants[i];
      const Fr t0 = xl_value + ci;
      const Fr tmp_value = t0 * t0;
      Variable tmp = cs.alloc([&] { return tmp_value; });
      cs.enforce([&](LinearCombination lc) { return lc + xlv + std::make_pair(ci, ConstraintSystem::one()); },
                 [&](LinearCombination lc) { return lc + xlv + std::make_pair(ci, ConstraintSystem::one()); },
                 [&](LinearCombination lc) { return lc + tmp; });
      const Fr new_xl_value = t0 * tmp_value + xr_value;
      Variable new_xl = (i == rounds - 1) ? cs.alloc_input([&] { return new_xl_value; })
                                          : cs.alloc([&] { return new_xl_value; });
      cs.enforce([&](LinearCombination lc) { return lc + tmp; },
                 [&](LinearCombination lc) { return lc + xlv + std::make_pair(ci, ConstraintSystem::one()); },
                 [&](LinearCombination lc) { return lc + new_xl - xrv; });
      xrv = xlv; xr_value = xl_value;
      xlv = new_xl; xl_value = new_xl_value;
    }
  }
};

// Synthetic multiplicative chain: M rounds
//   even i: (x_i + k_i) * (x_i + k'_i) = x_{i+1}       (x_i in the A and B queries)
//   odd  i: (x_i + k_i + 0*x_0) * (k'_i)  = x_{i+1}    (x_i only in A; a zero-coefficient term, prover.rs:31)
// and finally x_M * 1 = out (public input).  Constants from SplitMix64(seed).
class ChainCircuit : public Circuit {
 public:
  uint64_t seed;
  size_t rounds;
  Fr x0;
  static uint64_t splitmix(uint64_t &st) {
    uint64_t z = (st += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
  }
  void synthesize(ConstraintSystem &cs) override {
    uint64_t st = seed;
    Fr x_value = x0;
    Variable x = cs.alloc([&] { return x_value; });
    const Variable first = x;
    for (size_t i = 0; i < rounds; i++) {
      const Fr k = Fr::from_u64(splitmix(st)), k2 = Fr::from_u64(splitmix(st) | 1);
      const Fr lhs = x_value + k;
      const Fr rhs = (i & 1) ? k2 : (x_value + k2);
      const Fr next_value = lhs * rhs;
      Variable next = cs.alloc([&] { return next_value; });
      if (i & 1) {
        cs.enforce([&](LinearCombination lc) { return lc + x + std::make_pair(k, ConstraintSystem::one()) + std::make_pair(Fr::zero(), first); },
                   [&](LinearCombination lc) { return lc + std::make_pair(k2, ConstraintSystem::one()); },
                   [&](LinearCombination lc) { return lc + next; });
      } else {
        cs.enforce([&](LinearCombination lc) { return lc + x + std::make_pair(k, ConstraintSystem::one()); },
                   [&](LinearCombination lc) { return lc + x + std::make_pair(k2, ConstraintSystem::one()); },
                   [&](LinearCombination lc) { return lc + next; });
      }
      x = next;
      x_value = next_value;
    }
    Variable out = cs.alloc_input([&] { return x_value; });
    cs.enforce([&](LinearCombination lc) { return lc + x; }, [&](LinearCombination lc) { return lc + ConstraintSystem::one(); },
               [&](LinearCombination lc) { return lc + out; });
  }
};

}  // namespace groth16

// ---------------------------------------------------------------------------------------------------
// C entry points (declared in include/bellman_hip.h)
// ---------------------------------------------------------------------------------------------------
struct bh_params {
  groth16::Parameters *p;
};

static int run_guarded_sums(const std::function<groth16::MsmSums()> &f, void *sums_out) {
  try {
    groth16::MsmSums m = f();
    memcpy(sums_out, &m, sizeof m);
    return BH_OK;
  } catch (const bellman::SynthesisError &e) { return e.code;
  } catch (const std::invalid_argument &) { return BH_ERR_INVALID_ARG;
  } catch (...) { return BH_ERR_HIP; }
}

// a non-owning groth16::R1cs over a handle that belongs to the C caller
struct R1csView {
  groth16::R1cs r;
  explicit R1csView(const bh_r1cs *h) : r(const_cast<bh_r1cs *>(h)) {}
  ~R1csView() { r.handle = nullptr; }
};

template <class F>
static int with_demo_circuit(int circuit_kind, size_t size, uint64_t seed, const void *witness, const void *constants, F &&f) {
  using namespace groth16;
  if (circuit_kind == 0) {   // MiMC: witness = xl | xr, constants = `size` round constants (Montgomery Fr)
    MiMCDemo c;
    c.xl = Fr::zero(); c.xr = Fr::zero();
    if (witness) { memcpy(&c.xl, witness, 32); memcpy(&c.xr, (const char *)witness + 32, 32); }
    c.constants = (const Fr *)constants;
    c.rounds = size;
    return f(c);
  }
  if (circuit_kind == 1) {   // chain: witness = x0, `size` rounds
    ChainCircuit c;
    c.seed = seed; c.rounds = size;
    c.x0 = Fr::zero();
    if (witness) memcpy(&c.x0, witness, 32);
    return f(c);
  }
  return BH_ERR_INVALID_ARG;
}

extern "C" {

int bh_groth16_params_create(bh_ctx *ctx, const void *alpha_g1, const void *beta_g1, const void *beta_g2,
                             const void *delta_g1, const void *delta_g2, const void *h, size_t nh, const void *l,
                             size_t nl, const void *a, size_t na, const void *b_g1, size_t nb1, const void *b_g2,
                             size_t nb2, bh_params **out) {
  try {
    groth16::VerifyingKey vk;
    memcpy(&vk.alpha_g1, alpha_g1, 96); memcpy(&vk.beta_g1, beta_g1, 96); memcpy(&vk.beta_g2, beta_g2, 192);
    memcpy(&vk.delta_g1, delta_g1, 96); memcpy(&vk.delta_g2, delta_g2, 192);
    *out = new bh_params{new groth16::Parameters(ctx, vk, (const groth16::G1Affine *)h, nh, (const groth16::G1Affine *)l, nl,
                                                 (const groth16::G1Affine *)a, na, (const groth16::G1Affine *)b_g1, nb1,
                                                 (const groth16::G2Affine *)b_g2, nb2)};
    return BH_OK;
  } catch (const bellman::SynthesisError &e) { return e.code;
  } catch (...) { return BH_ERR_HIP; }
}
int bh_groth16_params_read(bh_ctx *ctx, const void *bytes, size_t len, int checked, bh_params **out) {
  if (!ctx || !out || (len && !bytes)) return BH_ERR_INVALID_ARG;
  try {
    *out = new bh_params{new groth16::Parameters(ctx, bytes, len, checked != 0)};
    return BH_OK;
  } catch (const bellman::IoError &e) { return e.code;
  } catch (const bellman::SynthesisError &e) { return e.code;
  } catch (...) { return BH_ERR_HIP; }
}
int bh_groth16_generate(bh_ctx *ctx, bh_r1cs *r1cs, const void *g1, const void *g2, const void *alpha, const void *beta,
                        const void *gamma, const void *delta, const void *tau, bh_params **out) {
  if (!ctx || !r1cs || !out) return BH_ERR_INVALID_ARG;
  using namespace groth16;
  try {
    R1csView view(r1cs);
    G1Affine p1; G2Affine p2;
    Fr f[5];
    memcpy(&p1, g1, 96); memcpy(&p2, g2, 192);
    const void *src[5] = {alpha, beta, gamma, delta, tau};
    for (int i = 0; i < 5; i++) memcpy(&f[i], src[i], 32);
    *out = new bh_params{new Parameters(ctx, view.r, p1, p2, f[0], f[1], f[2], f[3], f[4])};
    return BH_OK;
  } catch (const bellman::SynthesisError &e) { return e.code;
  } catch (...) { return BH_ERR_HIP; }
}
int bh_groth16_params_write(const bh_params *p, void *buf, size_t cap, size_t *len) {
  if (!p || !len) return BH_ERR_INVALID_ARG;
  try {
    const groth16::Parameters &P = *p->p;
    size_t need = 864 + 4 + P.vk.ic.size() * 96 + 5 * 4;
    const bh_bases *qs[5] = {P.h, P.l, P.a, P.b_g1, P.b_g2};
    for (int q = 0; q < 5; q++) need += bh_bases_len(qs[q]) * (q < 4 ? 96 : 192);
    *len = need;
    if (!buf || cap < need) return buf ? BH_ERR_INVALID_ARG : BH_OK;   // size query when buf == NULL
    std::vector<unsigned char> o = P.write();
    memcpy(buf, o.data(), o.size());
    return BH_OK;
  } catch (const bellman::SynthesisError &e) { return e.code;
  } catch (...) { return BH_ERR_HIP; }
}
int bh_groth16_params_vk_ext(const bh_params *p, void *gamma_g2, void *ic_out, size_t ic_cap, size_t *n_ic) {
  if (!p) return BH_ERR_INVALID_ARG;
  const groth16::VerifyingKey &vk = p->p->vk;
  if (gamma_g2) memcpy(gamma_g2, &vk.gamma_g2, 192);
  if (n_ic) *n_ic = vk.ic.size();
  if (ic_out) {
    if (ic_cap < vk.ic.size()) return BH_ERR_INVALID_ARG;
    if (!vk.ic.empty()) memcpy(ic_out, vk.ic.data(), vk.ic.size() * 96);
  }
  return BH_OK;
}
int bh_groth16_params_query(const bh_params *p, int which, const bh_bases **bases, size_t *len) {
  if (!p || which < 0 || which > 4) return BH_ERR_INVALID_ARG;
  const bh_bases *q[5] = {p->p->h, p->p->l, p->p->a, p->p->b_g1, p->p->b_g2};
  if (bases) *bases = q[which];
  if (len) *len = bh_bases_len(q[which]);
  return BH_OK;
}
int bh_groth16_params_vk(const bh_params *p, void *alpha_g1, void *beta_g1, void *beta_g2, void *delta_g1, void *delta_g2) {
  if (!p) return BH_ERR_INVALID_ARG;
  const groth16::VerifyingKey &vk = p->p->vk;
  if (alpha_g1) memcpy(alpha_g1, &vk.alpha_g1, 96);
  if (beta_g1) memcpy(beta_g1, &vk.beta_g1, 96);
  if (beta_g2) memcpy(beta_g2, &vk.beta_g2, 192);
  if (delta_g1) memcpy(delta_g1, &vk.delta_g1, 96);
  if (delta_g2) memcpy(delta_g2, &vk.delta_g2, 192);
  return BH_OK;
}
double bh_test_synthesis_ms(int circuit_kind, size_t size, uint64_t seed, int mode) {
  // host-only timing of circuit synthesis (no device involved): mode 0 = ProvingAssignment (the
  // reference's structure: every linear combination evaluated on the host), 1 = WitnessAssignment
  using namespace groth16;
  std::vector<Fr> constants(circuit_kind == 0 ? size : 0, Fr::from_u64(7));
  Fr wit[2] = {Fr::from_u64(123456789), Fr::from_u64(987654321)};
  double ms = -1.0;
  with_demo_circuit(circuit_kind, size, seed, wit, constants.data(), [&](bellman::Circuit &c) -> int {
    const auto t0 = std::chrono::steady_clock::now();
    if (mode == 0) {
      ProvingAssignment pa;
      pa.alloc_input([] { return Fr::one(); });
      c.synthesize(pa);
    } else {
      WitnessAssignment w;
      w.alloc_input([] { return Fr::one(); });
      c.synthesize(w);
    }
    ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
  });
  return ms;
}
void bh_test_fr_from_u512_host(void *r, const void *limbs8) {
  uint64_t w[8];
  memcpy(w, limbs8, 64);
  const bellman::Fr f = bellman::Fr::from_u512(w);
  memcpy(r, &f, 32);
}
void bh_proof_write(const void *proof_affine, void *out192) {
  groth16::Proof p;
  memcpy(&p.a, proof_affine, 96);
  memcpy(&p.b, (const char *)proof_affine + 96, 192);
  memcpy(&p.c, (const char *)proof_affine + 288, 96);
  p.write((unsigned char *)out192);
}
void bh_groth16_params_release(bh_params *p) {
  if (!p) return;
  delete p->p;
  delete p;
}

static int run_guarded(const std::function<groth16::Proof()> &f, void *proof_out) {
  try {
    groth16::Proof p = f();
    memcpy(proof_out, &p.a, 96);
    memcpy((char *)proof_out + 96, &p.b, 192);
    memcpy((char *)proof_out + 288, &p.c, 96);
    return BH_OK;
  } catch (const bellman::SynthesisError &e) { return e.code;
  } catch (const std::invalid_argument &) { return BH_ERR_INVALID_ARG;
  } catch (...) { return BH_ERR_HIP; }
}

int bh_groth16_prove_assignment(bh_params *params, const void *a_evals, const void *b_evals, const void *c_evals,
                                size_t n_constraints, const void *input_assignment, size_t n_inputs,
                                const void *aux_assignment, size_t n_aux, const uint64_t *a_aux_density,
                                const uint64_t *b_input_density, const uint64_t *b_aux_density, const void *r,
                                const void *s, void *proof_out, float *timings4) {
  using namespace groth16;
  ProvingAssignment pa;
  auto fill = [](std::vector<Fr> &v, const void *src, size_t n) { v.resize(n); if (n) memcpy(v.data(), src, n * 32); };
  fill(pa.a, a_evals, n_constraints); fill(pa.b, b_evals, n_constraints); fill(pa.c, c_evals, n_constraints);
  fill(pa.input_assignment, input_assignment, n_inputs); fill(pa.aux_assignment, aux_assignment, n_aux);
  auto fill_d = [](bellman::DensityTracker &d, const uint64_t *w, size_t n) {
    for (size_t i = 0; i < n; i++) { d.add_element(); if ((w[i >> 6] >> (i & 63)) & 1) d.inc(i); }
  };
  fill_d(pa.a_aux_density, a_aux_density, n_aux);
  fill_d(pa.b_input_density, b_input_density, n_inputs);
  fill_d(pa.b_aux_density, b_aux_density, n_aux);
  Fr rr, ss;
  memcpy(&rr, r, 32); memcpy(&ss, s, 32);
  ProveTimings tm = {0, 0, 0, 0};
  int rc = run_guarded([&] { return prove_assignment(pa, *params->p, rr, ss, &tm); }, proof_out);
  if (timings4) { timings4[0] = tm.synthesis_ms; timings4[1] = tm.h_poly_ms; timings4[2] = tm.msm_ms; timings4[3] = tm.total_ms; }
  return rc;
}

int bh_groth16_prove_witness(bh_params *params, const bh_r1cs *r1cs, const void *input_assignment, size_t n_inputs,
                             const void *aux_assignment, size_t n_aux, const void *r, const void *s, void *proof_out,
                             float *timings4) {
  using namespace groth16;
  if (!params || !r1cs) return BH_ERR_INVALID_ARG;
  R1csView view(r1cs);
  std::vector<Fr> in(n_inputs), aux(n_aux);   // caller records may be unaligned
  if (n_inputs) memcpy(in.data(), input_assignment, n_inputs * 32);
  if (n_aux) memcpy(aux.data(), aux_assignment, n_aux * 32);
  Fr rr, ss;
  memcpy(&rr, r, 32); memcpy(&ss, s, 32);
  ProveTimings tm = {0, 0, 0, 0};
  int rc = run_guarded([&] { return prove_witness(view.r, *params->p, in.data(), n_inputs, aux.data(), n_aux, rr, ss, &tm); },
                       proof_out);
  if (timings4) { timings4[0] = tm.synthesis_ms; timings4[1] = tm.h_poly_ms; timings4[2] = tm.msm_ms; timings4[3] = tm.total_ms; }
  return rc;
}

int bh_groth16_demo_r1cs(bh_ctx *ctx, int circuit_kind, size_t size, uint64_t seed, const void *constants, bh_r1cs **out) {
  if (!ctx || !out) return BH_ERR_INVALID_ARG;
  return with_demo_circuit(circuit_kind, size, seed, nullptr, constants, [&](bellman::Circuit &c) -> int {
    try {
      groth16::R1cs r(c, ctx);
      *out = r.handle;
      r.handle = nullptr;   // ownership moves to the caller (bh_r1cs_release)
      return BH_OK;
    } catch (const bellman::SynthesisError &e) { return e.code;
    } catch (...) { return BH_ERR_HIP; }
  });
}

int bh_groth16_prove_demo_r1cs(bh_params *params, const bh_r1cs *r1cs, int circuit_kind, size_t size, uint64_t seed,
                               const void *witness, const void *constants, const void *r, const void *s, void *proof_out,
                               float *timings4) {
  using namespace groth16;
  if (!params || !r1cs) return BH_ERR_INVALID_ARG;
  R1csView view(r1cs);
  Fr rr, ss;
  memcpy(&rr, r, 32); memcpy(&ss, s, 32);
  ProveTimings tm = {0, 0, 0, 0};
  int rc = with_demo_circuit(circuit_kind, size, seed, witness, constants, [&](bellman::Circuit &c) -> int {
    return run_guarded([&] { return create_proof(c, view.r, *params->p, rr, ss, &tm); }, proof_out);
  });
  if (timings4) { timings4[0] = tm.synthesis_ms; timings4[1] = tm.h_poly_ms; timings4[2] = tm.msm_ms; timings4[3] = tm.total_ms; }
  return rc;
}

int bh_groth16_prove_witness_part(bh_params *params, const bh_r1cs *r1cs, const void *input_assignment, size_t n_inputs,
                                  const void *aux_assignment, size_t n_aux, size_t part, size_t parts, void *sums_out,
                                  float *timings4) {
  using namespace groth16;
  if (!params || !r1cs || !sums_out) return BH_ERR_INVALID_ARG;
  R1csView view(r1cs);
  std::vector<Fr> in(n_inputs), aux(n_aux);
  if (n_inputs) memcpy(in.data(), input_assignment, n_inputs * 32);
  if (n_aux) memcpy(aux.data(), aux_assignment, n_aux * 32);
  ProveTimings tm = {0, 0, 0, 0};
  int rc = run_guarded_sums([&] { return prove_witness_part(view.r, *params->p, in.data(), n_inputs, aux.data(), n_aux, part, parts, &tm); },
                            sums_out);
  if (timings4) { timings4[0] = tm.synthesis_ms; timings4[1] = tm.h_poly_ms; timings4[2] = tm.msm_ms; timings4[3] = tm.total_ms; }
  return rc;
}

int bh_groth16_prove_demo_r1cs_part(bh_params *params, const bh_r1cs *r1cs, int circuit_kind, size_t size, uint64_t seed,
                                    const void *witness, const void *constants, size_t part, size_t parts, void *sums_out,
                                    float *timings4) {
  using namespace groth16;
  if (!params || !r1cs || !sums_out) return BH_ERR_INVALID_ARG;
  R1csView view(r1cs);
  ProveTimings tm = {0, 0, 0, 0};
  int rc = with_demo_circuit(circuit_kind, size, seed, witness, constants, [&](bellman::Circuit &c) -> int {
    return run_guarded_sums([&] {
      const double t0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
      WitnessAssignment w;
      w.input_assignment.reserve(view.r.num_inputs);
      w.aux_assignment.reserve(view.r.num_aux);
      w.alloc_input([] { return Fr::one(); });
      c.synthesize(w);
      const double t1 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
      MsmSums m = prove_witness_part(view.r, *params->p, w.input_assignment.data(), w.input_assignment.size(),
                                     w.aux_assignment.data(), w.aux_assignment.size(), part, parts, &tm);
      tm.synthesis_ms = (float)(t1 - t0);
      tm.total_ms += tm.synthesis_ms;
      return m;
    }, sums_out);
  });
  if (timings4) { timings4[0] = tm.synthesis_ms; timings4[1] = tm.h_poly_ms; timings4[2] = tm.msm_ms; timings4[3] = tm.total_ms; }
  return rc;
}

void bh_groth16_sums_add(void *acc, const void *other) {
  groth16::MsmSums a, b;
  memcpy(&a, acc, sizeof a);
  memcpy(&b, other, sizeof b);
  a.add(b);
  memcpy(acc, &a, sizeof a);
}

int bh_groth16_assemble(bh_params *params, const void *sums, const void *r, const void *s, void *proof_out) {
  using namespace groth16;
  if (!params || !sums) return BH_ERR_INVALID_ARG;
  MsmSums m;
  memcpy(&m, sums, sizeof m);
  Fr rr, ss;
  memcpy(&rr, r, 32); memcpy(&ss, s, 32);
  return run_guarded([&] { return assemble_proof(*params->p, m, rr, ss); }, proof_out);
}

int bh_groth16_prove_demo(bh_params *params, int circuit_kind, size_t size, uint64_t seed, const void *witness,
                          const void *constants, const void *r, const void *s, void *proof_out, float *timings4) {
  using namespace groth16;
  Fr rr, ss;
  memcpy(&rr, r, 32); memcpy(&ss, s, 32);
  ProveTimings tm = {0, 0, 0, 0};
  int rc;
  if (circuit_kind == 0) {   // MiMC: witness = xl | xr, constants = `size` round constants (Montgomery Fr)
    MiMCDemo c;
    memcpy(&c.xl, witness, 32); memcpy(&c.xr, (const char *)witness + 32, 32);
    c.constants = (const Fr *)constants;
    c.rounds = size;
    rc = run_guarded([&] { return create_proof(c, *params->p, rr, ss, &tm); }, proof_out);
  } else if (circuit_kind == 1) {   // chain: witness = x0, `size` rounds
    ChainCircuit c;
    c.seed = seed; c.rounds = size;
    memcpy(&c.x0, witness, 32);
    rc = run_guarded([&] { return create_proof(c, *params->p, rr, ss, &tm); }, proof_out);
  } else {
    return BH_ERR_INVALID_ARG;
  }
  if (timings4) { timings4[0] = tm.synthesis_ms; timings4[1] = tm.h_poly_ms; timings4[2] = tm.msm_ms; timings4[3] = tm.total_ms; }
  return rc;
}

}  // extern "C"

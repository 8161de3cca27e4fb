// groth16::Parameters (device-resident CRS): construction from host vectors, Parameters::read, the
// device parameter generator, Parameters::write, and Proof::write.  Reference map in groth16.hpp.
#include <string.h>

#include <vector>

#include "groth16_internal.hpp"
#include "host_fp.hpp"

namespace groth16 {
using namespace bellman;
using namespace detail;

bool G1Affine::is_identity() const { uint64_t o = 0; for (uint64_t x : v) o |= x; return o == 0; }
bool G2Affine::is_identity() const { uint64_t o = 0; for (uint64_t x : v) o |= x; return o == 0; }

Parameters::Parameters(bh_ctx *c, const VerifyingKey &k, const G1Affine *hq, size_t nh, const G1Affine *lq, size_t nl,
                       const G1Affine *aq, size_t na, const G1Affine *b1, size_t nb1, const G2Affine *b2, size_t nb2)
    : ctx(c), vk(k) {
  h = l = a = b_g1 = b_g2 = nullptr;
  try {
    check(bh_bases_register(ctx, BH_G1, hq, nh, 96, -1, &h));
    check(bh_bases_register(ctx, BH_G1, lq, nl, 96, -1, &l));
    check(bh_bases_register(ctx, BH_G1, aq, na, 96, -1, &a));
    check(bh_bases_register(ctx, BH_G1, b1, nb1, 96, -1, &b_g1));
    check(bh_bases_register(ctx, BH_G2, b2, nb2, 192, -1, &b_g2));
  } catch (...) {   // the destructor does not run for a half-built object: release what was registered
    for (bh_bases *b : {h, l, a, b_g1, b_g2}) bh_bases_release(ctx, b);
    throw;
  }
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

size_t Parameters::serialized_size() const {
  size_t need = 3 * 96 + 3 * 192 + 4 + vk.ic.size() * 96;   // alpha_g1, beta_g1, beta_g2, gamma_g2, delta_g1, delta_g2, |ic|, ic
  const bh_bases *qs[5] = {h, l, a, b_g1, b_g2};
  for (int q = 0; q < 5; q++) need += 4 + bh_bases_len(qs[q]) * (q < 4 ? 96 : 192);
  return need;
}
// The five queries are encoded on the device and copied straight into `dst` (bh_bases_write_uncompressed): round 3's host
// loop over 2.6 M points took 1.05 s of the 1.09 s this call needed for a 2^20-constraint CRS; the growing vector it
// went through until late in round 4 (three reallocation copies, 450 MB of zero fill, one more copy into the C caller's
// buffer) was most of the 0.36 s that were left.
void Parameters::write_into(unsigned char *dst, size_t cap) const {
  if (cap < serialized_size()) throw std::runtime_error("Parameters::write_into: buffer too small");
  std::vector<unsigned char> head;   // the verifying key: a few hundred bytes, encoded on the host
  put_g1(head, vk.alpha_g1); put_g1(head, vk.beta_g1); put_g2(head, vk.beta_g2); put_g2(head, vk.gamma_g2);
  put_g1(head, vk.delta_g1); put_g2(head, vk.delta_g2);
  put_u32(head, vk.ic.size());
  for (const G1Affine &p : vk.ic) put_g1(head, p);
  memcpy(dst, head.data(), head.size());
  size_t at = head.size();
  const bh_bases *qs[5] = {h, l, a, b_g1, b_g2};
  for (int q = 0; q < 5; q++) {
    const size_t n = bh_bases_len(qs[q]), rec = q < 4 ? 96 : 192;
    for (int sft = 24; sft >= 0; sft -= 8) dst[at++] = (unsigned char)(n >> sft);
    if (n) check(bh_bases_write_uncompressed(ctx, qs[q], 0, n, dst + at));
    at += n * rec;
  }
}
std::vector<unsigned char> Parameters::write() const {
  std::vector<unsigned char> o(serialized_size());
  write_into(o.data(), o.size());
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

}  // namespace groth16

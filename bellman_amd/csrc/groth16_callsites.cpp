// The sequences of C-ABI calls that bellman's `create_proof` issues once the reference tree carries
// shim/patches/bellman-hip.patch - transcribed in C++ so that the drop-in as patched can be run, checked against
// bh_groth16_prove_assignment and timed in this image (it has no Rust toolchain; the Rust files are in shim/).
//
//   mode 1  "prover.rs patched" - `issue_on_device` of the patched groth16/src/prover.rs:
//             bh_scalars_register x2 (input / aux assignment, Montgomery as they are in memory; shared by the
//             multiexps that use them, prover.rs:267,279,285,300,306,316,318), bh_msm_async_scalars x7,
//             bh_h_poly_fr_scalars (prover.rs:221-242 in one call, coefficients stay in HBM),
//             bh_msm_async_scalars (H), bh_msm_wait x8 in the order of prover.rs:339-354, bh_scalars_release x3.
//   mode 0  "multiexp.rs + domain.rs patched only" - the UNCHANGED call sites of prover.rs:221-318 running
//             through the patched `EvaluationDomain` and `multiexp`: from_coeffs (host padding), 7 x bh_fft_fr on
//             host vectors, mul_assign / sub_assign / divide_by_z_on_coset on the host (bellman's worker.scope
//             loops: one chunk per host thread), the serial Fr -> Exponent passes (prover.rs:241-261), and
//             8 x bh_msm_async with canonical host scalars - each multiexp re-reads its `Arc<Vec<Exponent>>`
//             (src/hip.rs exponent_words) and uploads it again.
//   mode 2  [r5] "multiexp.rs + domain.rs patched only", second form of that patch level (shim/patches, feature `hip`):
//             the call sites of prover.rs:221-318 are again UNCHANGED, but
//               * an `EvaluationDomain` keeps its vector in HBM between calls: the first transform uploads `coeffs`,
//                 ifft / coset_fft / mul_assign / sub_assign / divide_by_z_on_coset / icoset_fft run on the device vector
//                 (bh_fft_fr_dev, bh_fr_*_dev), `into_coeffs` / `as_ref` download it - 3 uploads + 1 download per proof
//                 instead of 7 round trips, no host pointwise passes;
//               * `Exponent::from(&Scalar)` (src/multiexp.rs:172-184, in the patched file) keeps the element as it is in
//                 memory (`Exponent::Raw`: classification only, no Montgomery reduction) - the serial maps of
//                 prover.rs:241-261 become copies - and the patched `multiexp` gathers an `Arc<Vec<Exponent>>` into
//                 contiguous words on the worker's threads ONCE per Arc (a cache keyed by the Arc's pointer, as for base
//                 vectors), registers it (bh_scalars_register, Montgomery) and issues bh_msm_async_scalars: the aux
//                 assignment is uploaded once for its four multiexps.
// All end with the unchanged tail of create_proof (prover.rs:320-360), here the mirror's assemble_proof.
#include <string.h>

#include <memory>
#include <thread>
#include <vector>

#include "../../include/bellman_hip_test.h"
#include "groth16_internal.hpp"

namespace groth16 {
using namespace bellman;
using namespace detail;

namespace {
struct CallSiteInputs {
  const Fr *a, *b, *c; size_t n_cons;
  const Fr *inputs; size_t n_in;
  const Fr *aux; size_t n_aux;
  const uint64_t *a_aux_density, *b_input_density, *b_aux_density;
};
size_t popcount_bits(const uint64_t *w, size_t n) {
  size_t t = 0;
  for (size_t i = 0; i < n / 64; i++) t += (size_t)__builtin_popcountll(w[i]);
  if (n & 63) t += (size_t)__builtin_popcountll(w[n / 64] & ((uint64_t(1) << (n & 63)) - 1));
  return t;
}
// every issued job is waited on even when an earlier wait fails or something throws (the Rust MsmJob's Drop)
struct Jobs {
  bh_msm_job *j[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  ~Jobs() {
    unsigned char sink[192];
    for (bh_msm_job *&x : j) if (x) { (void)bh_msm_wait(x, sink); x = nullptr; }
  }
  int wait(int i, void *out) { bh_msm_job *x = j[i]; j[i] = nullptr; return bh_msm_wait(x, out); }
};
struct ScalarsHandle {
  bh_scalars *s = nullptr;
  ScalarsHandle() = default;
  ScalarsHandle(ScalarsHandle &&o) noexcept : s(o.s) { o.s = nullptr; }
  ScalarsHandle(const ScalarsHandle &) = delete;
  ScalarsHandle &operator=(const ScalarsHandle &) = delete;
  ~ScalarsHandle() { if (s) bh_scalars_release(s); }
};
// wait order of prover.rs:339-354: a_inputs, a_aux, b_g1_inputs, b_g1_aux, b_g2_inputs, b_g2_aux, h, l
enum { A_IN, A_AUX, B1_IN, B1_AUX, B2_IN, B2_AUX, H, L };
MsmSums wait_all(Jobs &jobs, const Parameters &p) {
  // prover.rs:320-324 sits between the last multiexp call and the first wait()
  if (p.vk.delta_g1.is_identity() || p.vk.delta_g2.is_identity())
    throw SynthesisError(BH_ERR_UNEXPECTED_IDENTITY, "UnexpectedIdentity");
  MsmSums m;
  int rcs[8];
  rcs[0] = jobs.wait(A_IN, &m.a_in); rcs[1] = jobs.wait(A_AUX, &m.a_aux);
  rcs[2] = jobs.wait(B1_IN, &m.b1_in); rcs[3] = jobs.wait(B1_AUX, &m.b1_aux);
  rcs[4] = jobs.wait(B2_IN, &m.b2_in); rcs[5] = jobs.wait(B2_AUX, &m.b2_aux);
  rcs[6] = jobs.wait(H, &m.h); rcs[7] = jobs.wait(L, &m.l);
  for (int rc : rcs) check(rc);   // the first failing `?` in wait order
  return m;
}

// ---- mode 1 ----------------------------------------------------------------------------------------------------------
MsmSums issue_patched(Parameters &p, const CallSiteInputs &in) {
  bh_ctx *ctx = p.ctx;
  // declared before `jobs`: the jobs are waited on (Jobs' destructor) before the vectors they read are released
  ScalarsHandle inputs, aux, h;
  Jobs jobs;
  check(bh_scalars_register(ctx, in.inputs, in.n_in, BH_SCALARS_MONT, &inputs.s));
  check(bh_scalars_register(ctx, in.aux, in.n_aux, BH_SCALARS_MONT, &aux.s));
  const size_t b_in_total = popcount_bits(in.b_input_density, in.n_in);
  // the G2 multiexp first: the longest job; the bucket accumulations run on the device in issue order and its
  // reduction tail then overlaps the G1 accumulations (csrc/common.hpp, the accumulation chain)
  check(bh_msm_async_scalars(ctx, p.b_g2, b_in_total, aux.s, 0, in.n_aux, in.b_aux_density, in.n_aux, nullptr, &jobs.j[B2_AUX]));
  check(bh_msm_async_scalars(ctx, p.l, 0, aux.s, 0, in.n_aux, nullptr, 0, nullptr, &jobs.j[L]));
  check(bh_msm_async_scalars(ctx, p.a, 0, inputs.s, 0, in.n_in, nullptr, 0, nullptr, &jobs.j[A_IN]));
  check(bh_msm_async_scalars(ctx, p.a, in.n_in, aux.s, 0, in.n_aux, in.a_aux_density, in.n_aux, nullptr, &jobs.j[A_AUX]));
  check(bh_msm_async_scalars(ctx, p.b_g1, 0, inputs.s, 0, in.n_in, in.b_input_density, in.n_in, nullptr, &jobs.j[B1_IN]));
  check(bh_msm_async_scalars(ctx, p.b_g1, b_in_total, aux.s, 0, in.n_aux, in.b_aux_density, in.n_aux, nullptr, &jobs.j[B1_AUX]));
  check(bh_msm_async_scalars(ctx, p.b_g2, 0, inputs.s, 0, in.n_in, in.b_input_density, in.n_in, nullptr, &jobs.j[B2_IN]));
  check(bh_h_poly_fr_scalars(ctx, in.a, in.b, in.c, in.n_cons, &h.s));
  check(bh_msm_async_scalars(ctx, p.h, 0, h.s, 0, bh_scalars_len(h.s), nullptr, 0, nullptr, &jobs.j[H]));
  return wait_all(jobs, p);
}

// ---- mode 0 ----------------------------------------------------------------------------------------------------------
// worker.scope(len, |scope, chunk| ...) of src/multicore.rs:78-91: one chunk per host thread
template <class F>
void scope(size_t n, F &&body) {
  unsigned threads = std::thread::hardware_concurrency();
  if (threads == 0) threads = 1;
  const size_t chunk = n < threads ? 1 : n / threads;   // multicore.rs:83-87
  std::vector<std::thread> pool;
  for (size_t lo = 0; lo < n; lo += chunk) {
    const size_t hi = lo + chunk < n ? lo + chunk : n;
    pool.emplace_back([=, &body] { body(lo, hi); });
  }
  for (std::thread &t : pool) t.join();
}
// `.map(|s| s.into())` of prover.rs:241-261 (serial in the reference: "TODO: parallelize"), then the
// exponent_words pass of the patched multiexp (src/hip.rs) happens per multiexp call below
std::vector<uint64_t> to_exponents(const Fr *v, size_t n) {
  std::vector<uint64_t> out(n * 4);
  for (size_t i = 0; i < n; i++) v[i].to_canonical(&out[4 * i]);
  return out;
}
MsmSums issue_unpatched_prover(Parameters &p, const CallSiteInputs &in) {
  bh_ctx *ctx = p.ctx;
  // EvaluationDomain::from_coeffs (domain.rs:47-79)
  uint32_t exp = 0;
  size_t m = 1;
  while (m < in.n_cons) {
    m *= 2;
    exp++;
    if (exp >= 32) throw SynthesisError(BH_ERR_DEGREE_TOO_LARGE, "PolynomialDegreeTooLarge");
  }
  std::vector<Fr> a(m, Fr::zero()), b(m, Fr::zero()), c(m, Fr::zero());
  if (in.n_cons) {
    memcpy(a.data(), in.a, in.n_cons * 32);
    memcpy(b.data(), in.b, in.n_cons * 32);
    memcpy(c.data(), in.c, in.n_cons * 32);
  }
  // host buffers handed to bh_msm_async stay valid until the job has been waited on (the Rust closure keeps its
  // `words` alive the same way): declared before `jobs`, destroyed after it
  std::vector<uint64_t> h_exps, in_exps, aux_exps;
  std::vector<std::vector<uint64_t>> call_words;
  call_words.reserve(8);
  Jobs jobs;
  // prover.rs:222-230
  for (std::vector<Fr> *v : {&a, &b, &c}) {
    check(bh_fft_fr(ctx, v->data(), exp, BH_IFFT));
    check(bh_fft_fr(ctx, v->data(), exp, BH_COSET_FFT));
  }
  // :232-236 on the host
  scope(m, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) a[i] = a[i] * b[i]; });
  std::vector<Fr>().swap(b);
  scope(m, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) a[i] = a[i] - c[i]; });
  std::vector<Fr>().swap(c);
  {
    const Fr zinv = (Fr::from_u64(7).pow_vartime((uint64_t)m) - Fr::one()).invert();   // domain.rs:129-151
    scope(m, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) a[i] = a[i] * zinv; });
  }
  check(bh_fft_fr(ctx, a.data(), exp, BH_ICOSET_FFT));
  a.resize(m - 1);                                   // :238-239
  h_exps = to_exponents(a.data(), a.size());         // :241-242
  auto multiexp = [&](bh_bases *bases, size_t skip, const std::vector<uint64_t> &exps, const uint64_t *density, int slot) {
    call_words.emplace_back(exps);                   // exponent_words: a fresh Vec<[u64; 4]> per call
    const std::vector<uint64_t> &words = call_words.back();
    const size_t n = words.size() / 4;
    check(bh_msm_async(ctx, bases, skip, words.data(), n, BH_SCALARS_CANONICAL, density, density ? n : 0, &jobs.j[slot]));
  };
  multiexp(p.h, 0, h_exps, nullptr, H);              // :244
  in_exps = to_exponents(in.inputs, in.n_in);        // :247-261
  aux_exps = to_exponents(in.aux, in.n_aux);
  const size_t b_in_total = popcount_bits(in.b_input_density, in.n_in);
  multiexp(p.l, 0, aux_exps, nullptr, L);                               // :263-268
  multiexp(p.a, 0, in_exps, nullptr, A_IN);                             // :275-280
  multiexp(p.a, in.n_in, aux_exps, in.a_aux_density, A_AUX);            // :281-286
  multiexp(p.b_g1, 0, in_exps, in.b_input_density, B1_IN);              // :296-301
  multiexp(p.b_g1, b_in_total, aux_exps, in.b_aux_density, B1_AUX);     // :302-307
  multiexp(p.b_g2, 0, in_exps, in.b_input_density, B2_IN);              // :312-317
  multiexp(p.b_g2, b_in_total, aux_exps, in.b_aux_density, B2_AUX);     // :318
  return wait_all(jobs, p);
}
// ---- mode 2 ----------------------------------------------------------------------------------------------------------
// the patched EvaluationDomain: `coeffs` on the host, its device copy while the last operation ran there.  In Rust the
// domain OWNS the vector it was built from (`from_coeffs(prover.a)` moves it, pads it in place) and `into_coeffs` hands the
// same allocation back; this harness only borrows the caller's arrays, so the host vector is a view until somebody needs it
// as a vector: the upload reads the caller's memory (the padding is zeroed on the device) and the download lands in a
// buffer that is recycled across proofs - neither a copy nor a fresh 32 MiB of page faults that the Rust code would not have.
struct DeviceDomain {
  bh_ctx *ctx;
  const Fr *src;
  size_t n, m;
  uint32_t exp;
  void *dev = nullptr;
  DeviceDomain(bh_ctx *c, const Fr *v, size_t n_, size_t m_, uint32_t e) : ctx(c), src(v), n(n_), m(m_), exp(e) {}
  DeviceDomain(const DeviceDomain &) = delete;
  // (the Rust DeviceVec's Drop: the operations on the vector were only enqueued - wait before the block returns to the pool)
  ~DeviceDomain() { if (dev) { (void)bh_ctx_synchronize(ctx); (void)bh_dev_free(ctx, dev); } }
  void to_device() {
    if (dev) return;
    check(bh_dev_alloc(ctx, m * 32, &dev));
    if (n < m) check(bh_dev_zero(ctx, (char *)dev + n * 32, (m - n) * 32));
    if (n) check(bh_dev_upload(ctx, dev, src, n * 32));
  }
  void transform(int mode) { to_device(); check(bh_fft_fr_dev(ctx, dev, exp, mode, nullptr)); }
  void mul_assign(DeviceDomain &o) { to_device(); o.to_device(); check(bh_fr_mul_assign_dev(ctx, dev, o.dev, m, nullptr)); }
  void sub_assign(DeviceDomain &o) { to_device(); o.to_device(); check(bh_fr_sub_assign_dev(ctx, dev, o.dev, m, nullptr)); }
  void divide_by_z_on_coset() { to_device(); check(bh_fr_divide_by_z_on_coset_dev(ctx, dev, exp, nullptr)); }
  // domain.rs:42-45: the download (bh_dev_download synchronises the context stream) into the vector the domain owns
  std::vector<Fr> &into_coeffs() {
    static thread_local std::vector<Fr> owned;
    to_device();
    owned.resize(m);
    check(bh_dev_download(ctx, owned.data(), dev, m * 32));
    return owned;
  }
};
// `.map(|s| s.into()).collect()` with Exponent::Raw: zero / one are still classified (multiexp.rs:174-177), nothing is
// converted; a fresh vector per call like the Rust `collect`
std::vector<Fr> to_raw_exponents(const Fr *v, size_t n) {
  std::vector<Fr> out;
  out.reserve(n);
  const Fr one = Fr::one();
  size_t special = 0;
  for (size_t i = 0; i < n; i++) {
    special += v[i].is_zero() || v[i] == one;   // (the tag a Rust enum would store)
    out.push_back(v[i]);
  }
  (void)special;
  return out;
}
MsmSums issue_unpatched_prover_resident(Parameters &p, const CallSiteInputs &in) {
  bh_ctx *ctx = p.ctx;
  uint32_t exp = 0;
  size_t m = 1;
  while (m < in.n_cons) {
    m *= 2;
    exp++;
    if (exp >= 32) throw SynthesisError(BH_ERR_DEGREE_TOO_LARGE, "PolynomialDegreeTooLarge");
  }
  // the Arc<Vec<Exponent>> cache of the patched multiexp: one registration per distinct vector, alive until the jobs end
  std::vector<std::pair<const Fr *, ScalarsHandle>> registered;
  registered.reserve(4);
  std::vector<Fr> h_exps, in_exps, aux_exps;
  Jobs jobs;
  {
    DeviceDomain a(ctx, in.a, in.n_cons, m, exp);
    std::unique_ptr<DeviceDomain> b(new DeviceDomain(ctx, in.b, in.n_cons, m, exp)), c(new DeviceDomain(ctx, in.c, in.n_cons, m, exp));
    for (DeviceDomain *v : {&a, b.get(), c.get()}) {   // prover.rs:222-230
      v->transform(BH_IFFT);
      v->transform(BH_COSET_FFT);
    }
    a.mul_assign(*b);                         // :232
    b.reset();                                // :233 drop(b)
    a.sub_assign(*c);                         // :234
    c.reset();                                // :235 drop(c)
    a.divide_by_z_on_coset();                 // :236
    a.transform(BH_ICOSET_FFT);               // :237
    std::vector<Fr> &av = a.into_coeffs();
    h_exps = to_raw_exponents(av.data(), m - 1);       // :238-242 (truncate, then the map)
  }
  auto multiexp = [&](bh_bases *bases, size_t skip, const std::vector<Fr> &exps, const uint64_t *density, int slot) {
    bh_scalars *sc = nullptr;
    for (auto &r : registered) if (r.first == exps.data()) sc = r.second.s;
    if (!sc) {
      // the gather into contiguous words (a Vec<Exponent> has a tag per element; rayon's par_iter in the patch - here one
      // plain pass, which is slower than that, not faster), then one upload
      std::vector<Fr> words;
      words.reserve(exps.size());
      words.insert(words.end(), exps.begin(), exps.end());
      registered.emplace_back(exps.data(), ScalarsHandle());
      check(bh_scalars_register(ctx, words.data(), words.size(), BH_SCALARS_MONT, &registered.back().second.s));
      sc = registered.back().second.s;
    }
    const size_t n = exps.size();
    check(bh_msm_async_scalars(ctx, bases, skip, sc, 0, n, density, density ? n : 0, nullptr, &jobs.j[slot]));
  };
  multiexp(p.h, 0, h_exps, nullptr, H);              // :244
  in_exps = to_raw_exponents(in.inputs, in.n_in);    // :247-261
  aux_exps = to_raw_exponents(in.aux, in.n_aux);
  const size_t b_in_total = popcount_bits(in.b_input_density, in.n_in);
  multiexp(p.l, 0, aux_exps, nullptr, L);                               // :263-268
  multiexp(p.a, 0, in_exps, nullptr, A_IN);                             // :275-280
  multiexp(p.a, in.n_in, aux_exps, in.a_aux_density, A_AUX);            // :281-286
  multiexp(p.b_g1, 0, in_exps, in.b_input_density, B1_IN);              // :296-301
  multiexp(p.b_g1, b_in_total, aux_exps, in.b_aux_density, B1_AUX);     // :302-307
  multiexp(p.b_g2, 0, in_exps, in.b_input_density, B2_IN);              // :312-317
  multiexp(p.b_g2, b_in_total, aux_exps, in.b_aux_density, B2_AUX);     // :318
  return wait_all(jobs, p);
}
}  // namespace

static Proof prove_via_call_sites(Parameters &params, const CallSiteInputs &in, const Fr &r, const Fr &s, int mode, float *ms2) {
  const double t0 = now_ms();
  const MsmSums sums = mode == 1 ? issue_patched(params, in) : mode == 2 ? issue_unpatched_prover_resident(params, in)
                                                                          : issue_unpatched_prover(params, in);
  const double t1 = now_ms();
  Proof p = assemble_proof(params, sums, r, s);   // prover.rs:320-360
  if (ms2) { ms2[0] = (float)(t1 - t0); ms2[1] = (float)(now_ms() - t0); }
  return p;
}
}  // namespace groth16

extern "C" int bh_test_groth16_prove_via_call_sites(bh_params *params, int mode, const void *a_evals, const void *b_evals,
                                                    const void *c_evals, size_t n_constraints, const void *input_assignment,
                                                    size_t n_inputs, const void *aux_assignment, size_t n_aux,
                                                    const uint64_t *a_aux_density, const uint64_t *b_input_density,
                                                    const uint64_t *b_aux_density, const void *r, const void *s,
                                                    void *proof_out, float *ms2) {
  using namespace groth16;
  if (!params || !r || !s || !proof_out || mode < 0 || mode > 2) return BH_ERR_INVALID_ARG;
  if ((n_constraints && (!a_evals || !b_evals || !c_evals)) || (n_inputs && (!input_assignment || !b_input_density)) ||
      (n_aux && (!aux_assignment || !a_aux_density || !b_aux_density)))
    return BH_ERR_INVALID_ARG;
  try {
    CallSiteInputs in{(const Fr *)a_evals, (const Fr *)b_evals, (const Fr *)c_evals, n_constraints,
                      (const Fr *)input_assignment, n_inputs, (const Fr *)aux_assignment, n_aux,
                      a_aux_density, b_input_density, b_aux_density};
    Fr rr, ss;
    memcpy(&rr, r, 32); memcpy(&ss, s, 32);
    const Proof p = prove_via_call_sites(*params->p, in, rr, ss, mode, ms2);
    memcpy(proof_out, &p.a, 96);
    memcpy((char *)proof_out + 96, &p.b, 192);
    memcpy((char *)proof_out + 288, &p.c, 96);
    return BH_OK;
  } catch (const bellman::SynthesisError &e) { return e.code;
  } catch (const std::invalid_argument &) { return BH_ERR_INVALID_ARG;
  } catch (...) { return BH_ERR_HIP; }
}

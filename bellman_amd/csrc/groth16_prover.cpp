// groth16::create_proof and friends: linear-combination evaluation during synthesis, the eight
// multiexps + h block on the device, proof assembly, the R1CS capture and the witness-only path.
// Reference map in groth16.hpp.
#include <string.h>

#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include <exception>
#include <functional>

#include "groth16_internal.hpp"

namespace groth16 {
using namespace bellman;
using namespace detail;

// ---- prover.rs:19-55 ------------------------------------------------------------------------------
static inline Fr eval(const LinearCombination &lc, DensityTracker *input_density, DensityTracker *aux_density,
                      const std::vector<Fr> &input_assignment, const std::vector<Fr> &aux_assignment) {
  Fr acc = Fr::zero();
  const Fr one = Fr::one();
  const size_t n = lc.size();
  for (size_t t = 0; t < n; t++) {
    const Variable &var = lc[t].first;
    const Fr &coeff = lc[t].second;
    if (coeff.is_zero()) continue;          // zero coefficients count for neither value nor density (:31)
    const Fr *value;
    if (var.kind() == Index::Input) {
      value = &input_assignment[var.idx()];
      if (input_density) input_density->inc(var.idx());
    } else {
      value = &aux_assignment[var.idx()];
      if (aux_density) aux_density->inc(var.idx());
    }
    // most terms carry the coefficient one (`lc + x`), then the ubiquitous `(c, CS::one())` terms: value one
    if (coeff == one) acc = acc + *value;
    else if (*value == one) acc = acc + coeff;
    else acc = acc + *value * coeff;
  }
  return acc;
}

// ---- recycled assignments ----------------------------------------------------------------------------------------
// A 2^20-constraint ProvingAssignment is five vectors of 32 MiB.  Grown by push_back from empty and freed after every
// proof they cost reallocation copies and ~40 000 first-touch page faults per proof - a third of the synthesis time
// (profiles/archive/r3_host_synthesis.txt).  Finished assignments are cleared (capacity kept) and handed to the next proof.
namespace {
template <class T>
class Recycler {
 public:
  std::unique_ptr<T> get() {
    {
      std::lock_guard<std::mutex> g(mu_);
      if (!free_.empty()) {
        std::unique_ptr<T> p = std::move(free_.back());
        free_.pop_back();
        return p;
      }
    }
    return std::unique_ptr<T>(new T());
  }
  void put(std::unique_ptr<T> p) {
    if (!p) return;
    reset(*p);
    std::lock_guard<std::mutex> g(mu_);
    if (free_.size() < 16) free_.push_back(std::move(p));   // at most 16 idle assignments are kept
  }

 private:
  static void reset(ProvingAssignment &a) {
    a.a.clear(); a.b.clear(); a.c.clear(); a.input_assignment.clear(); a.aux_assignment.clear();
    a.a_aux_density.clear(); a.b_input_density.clear(); a.b_aux_density.clear();
  }
  static void reset(WitnessAssignment &w) { w.input_assignment.clear(); w.aux_assignment.clear(); }
  std::mutex mu_;
  std::vector<std::unique_ptr<T>> free_;
};
Recycler<ProvingAssignment> g_assignments;
Recycler<WitnessAssignment> g_witnesses;
template <class T>
struct Recycled {   // returns the object to its pool on scope exit, exceptions included
  Recycler<T> &pool;
  std::unique_ptr<T> p;
  explicit Recycled(Recycler<T> &r) : pool(r), p(r.get()) {}
  ~Recycled() { pool.put(std::move(p)); }
  T &operator*() { return *p; }
};
}  // namespace
AsyncProof::~AsyncProof() {
  if (worker.joinable()) worker.join();
  g_assignments.put(std::move(assignment));
  g_witnesses.put(std::move(witness));
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
void ProvingAssignment::enforce(const LcFn &fa, const LcFn &fb, const LcFn &fc) {
  // inputs have full density in the A query; there is no C query (prover.rs:119-141).  The closures get combinations
  // that evaluate each term as it is added (groth16.hpp); a closure that returns some other, stored combination is
  // evaluated the classic way.  CONTRACT of the evaluating form (the reference's `eval` walks only the RETURNED
  // combination, prover.rs:19-55): a closure returns a combination derived linearly from its argument - terms added to a
  // copy that is then discarded would still count in the value and in the density maps (the accumulator lives in the sink,
  // groth16.hpp LcSink::acc) - or a stored combination built without touching the argument.  Every closure of the
  // reference's own circuits and gadgets has the first shape (`|lc| lc + a + (c, b)`).
  const std::vector<Fr> *in = &input_assignment, *ax = &aux_assignment;
  const LcSink sa{in, ax, nullptr, &a_aux_density}, sb{in, ax, &b_input_density, &b_aux_density}, sc{in, ax, nullptr, nullptr};
  // (the value goes from the combination's accumulator into the vector's new slot limb by limb: handed on as an Fr it is
  //  copied with 16-byte loads from the 8-byte stores of the last addition, which the store buffer cannot forward -
  //  a tenth of the synthesis time of a 2^20-constraint circuit, tools/host_profile.py)
  auto run = [&](const LcFn &f, const LcSink &sink, std::vector<Fr> &out) {
    const LinearCombination r = f(LinearCombination::evaluating(&sink));
    // a closure compiled with BELLMAN_HIP_CHECK_CLOSURES counted every term it added to any copy of its argument: the
    // returned combination must carry exactly those (prover.rs:19-55 evaluates the returned combination and nothing else)
    if (sink.pushed && (!r.is_evaluating() || r.size() != sink.pushed))
      throw std::invalid_argument("enforce: the closure added terms to a copy of its argument that it did not return");
    out.emplace_back();
    if (r.is_evaluating()) r.value_into(out.back());
    else out.back() = eval(r, sink.input_density, sink.aux_density, input_assignment, aux_assignment);
  };
  run(fa, sa, a);
  run(fb, sb, b);
  run(fc, sc, c);
}

namespace {
struct ProofStream {   // uploads + h block of one proof; independent of other proofs in flight
  bh_ctx *ctx;
  void *st = nullptr;
  // the h block is a short dependent chain on the proof's critical path (the H multiexp waits for it).
  // BELLMAN_HIP_H_PRIORITY=1 puts it on a high-priority stream; measured: no gain for one proof and 8 % less throughput
  // with twelve proofs in flight (profiles/archive/r3_call3_oversub.txt), so it is off by default
  explicit ProofStream(bh_ctx *c) : ctx(c) {
    static const bool high = [] { const char *e = getenv("BELLMAN_HIP_H_PRIORITY"); return e && *e == '1'; }();
    check(bh_stream_create_priority(ctx, high ? 1 : 0, &st));
  }
  ~ProofStream() { if (st) { (void)bh_stream_synchronize(ctx, st); (void)bh_stream_destroy(ctx, st); } }
  ProofStream(const ProofStream &) = delete;
};
// Declared after every DevBuf the proof stream writes (and so destroyed before them): if an exception
// unwinds the frame, queued uploads / memsets / evaluations finish before their targets return to the
// shared pool, where another proof's thread could pick them up.
struct StreamDrain {
  ProofStream &ps;
  ~StreamDrain() { if (ps.st) (void)bh_stream_synchronize(ps.ctx, ps.st); }
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
  // host path (prove_assignment): evaluations and density bitmaps computed during synthesis - plain views, so that the
  // C entry point hands the caller's arrays through without copying them
  bool host = false;
  const Fr *a = nullptr, *b = nullptr, *c = nullptr;
  const uint64_t *a_aux_density = nullptr, *b_input_density = nullptr, *b_aux_density = nullptr;   // LSB0 words
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

// test hook (host only): the slice of an n-term multiexp part `part` of `parts` computes
void proof_slice_for_tests(size_t n, size_t part, size_t parts, size_t *lo, size_t *hi) {
  const Slice sl = slice_of(n, part, parts);
  *lo = sl.lo;
  *hi = sl.hi;
}

// prover.rs:217-318 + the waits of :339-354: the eight multiexp results (of this part's slices)
// `while_running` (optional) is host work that needs no multiexp result: it runs after the last job has been
// issued and before the first wait, i.e. while the GPU is busy
static void msm_sums(const AssignmentSource &src, Parameters &params, size_t part, size_t parts, MsmSums &out,
                     ProveTimings *tm, const std::function<void()> *while_running = nullptr) {
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
    auto upload_density = [&](const uint64_t *words, size_t bits, std::unique_ptr<DevBuf> &buf) {
      const size_t nw = (bits + 63) / 64;
      buf.reset(new DevBuf(ctx, nw * 8 + 8));
      if (nw) check(bh_dev_upload_on(ctx, buf->p, words, nw * 8, ps.st));
      return (const uint64_t *)buf->p;
    };
    dens_a_aux = upload_density(src.a_aux_density, n_aux, dens_buf[0]);
    dens_b_in = upload_density(src.b_input_density, n_in, dens_buf[1]);
    dens_b_aux = upload_density(src.b_aux_density, n_aux, dens_buf[2]);
    for (size_t w = 0; w < (n_in + 63) / 64; w++) {   // get_total_density (multiexp.rs:154-156)
      uint64_t x = src.b_input_density[w];
      if (w == n_in / 64 && (n_in & 63)) x &= (uint64_t(1) << (n_in & 63)) - 1;
      b_in_total += (size_t)__builtin_popcountll(x);
    }
    hw_a_aux = src.a_aux_density; hw_b_in = src.b_input_density; hw_b_aux = src.b_aux_density;
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
  StreamDrain drain{ps};
  JobSet jobs;
  for (bh_msm_job **j : {&l_job, &a_in_job, &a_aux_job, &b1_in_job, &b1_aux_job, &b2_in_job, &b2_aux_job, &h_job}) jobs.track(j);
  // one multiexp over this part's slice of the scalars: `skip` advances by the number of bases the
  // skipped scalars would have consumed (all of them without a density map, the set bits with one)
  DevBuf dscratch(ctx, log_m > 11 ? m * 32 : 32);   // FFT ping-pong vector of the h block
  StreamDrain drain2{ps};                            // (drains before dscratch is released)
  // Optional two-phase issue (BH_MSM_HOLD / bh_msm_start, BELLMAN_HIP_PROOF_HOLD=1): first every job's digit + sort
  // stage, then the bucket accumulations in chain order.  Measured (profiles/archive/r3_call6_hold_ab.txt): no gain for one proof
  // - the accumulations are as slow with nothing beside them, 6.35 ms for G2 against 5.46 in a warm back-to-back loop:
  // the chip has just left idle clocks after the host's 51 ms of witness generation - and 12 % less throughput with
  // twelve proofs in flight (the sorts of one proof no longer fill the gaps of another).  Off by default.
  static const bool hold_env = [] { const char *e = getenv("BELLMAN_HIP_PROOF_HOLD"); return e && *e == '1'; }();
  const bool hold_jobs = hold_env && log_m > 16;
  auto issue = [&](bh_bases *bases, size_t skip, const void *scalars, size_t n, const uint64_t *dens_dev,
                   const uint64_t *dens_host, bh_msm_job **job, const void *scalars_host = nullptr) {
    const Slice sl = slice_of(n, part, parts);
    const size_t base_skip = skip + (dens_dev ? popcount_prefix(dens_host, sl.lo) : sl.lo);
    if (scalars_host && sl.hi - sl.lo <= 8) {
      // the `inputs` multiexps have one or two terms: handed over with their host scalars, the library
      // answers them on the host instead of running a kernel pipeline per term
      check(bh_msm_async(ctx, bases, base_skip, (const char *)scalars_host + sl.lo * 32, sl.hi - sl.lo, BH_SCALARS_MONT,
                         dens_host ? dens_host + sl.lo / 64 : nullptr, dens_host ? sl.hi - sl.lo : 0, job));
      BH_TRACE("  multiexp of %zu terms answered on the host", sl.hi - sl.lo);
      return;
    }
    const bh_msm_opts held = {0, 0, hold_jobs ? BH_MSM_HOLD : 0u};
    check(bh_msm_async_dev_opts(ctx, bases, base_skip, (const char *)scalars + sl.lo * 32, sl.hi - sl.lo, BH_SCALARS_MONT,
                                dens_dev ? dens_dev + sl.lo / 64 : nullptr, dens_dev ? sl.hi - sl.lo : 0, &held, job));
    BH_TRACE("  multiexp of %zu terms issued", sl.hi - sl.lo);
  };
  auto issue_seven = [&](bool longest_first) {
    // get_b_g1/g2(b_input_density_total, _) -> ((b,0),(b,total))   groth16/src/lib.rs:459-473
    // (small proofs: the G2 multiexp is the longest job, and the host needs ~0.2 ms to enqueue each job's ~20 launches:
    //  what is issued last finishes last)
    if (longest_first) issue(params.b_g2, b_in_total, d_aux.p, n_aux, dens_b_aux, hw_b_aux, &b2_aux_job);
    issue(params.l, 0, d_aux.p, n_aux, nullptr, nullptr, &l_job);
    // get_a(num_inputs, _) -> ((a,0),(a,num_inputs))            groth16/src/lib.rs:451-457
    issue(params.a, 0, d_in.p, n_in, nullptr, nullptr, &a_in_job, src.inputs);
    issue(params.a, n_in, d_aux.p, n_aux, dens_a_aux, hw_a_aux, &a_aux_job);
    issue(params.b_g1, 0, d_in.p, n_in, dens_b_in, hw_b_in, &b1_in_job, src.inputs);
    issue(params.b_g1, b_in_total, d_aux.p, n_aux, dens_b_aux, hw_b_aux, &b1_aux_job);
    issue(params.b_g2, 0, d_in.p, n_in, dens_b_in, hw_b_in, &b2_in_job, src.inputs);
    if (!longest_first) issue(params.b_g2, b_in_total, d_aux.p, n_aux, dens_b_aux, hw_b_aux, &b2_aux_job);
  };
  // h block (prover.rs:221-245), enqueue only: a, b, c stay in HBM; the quotient's coefficients are consumed by the
  // H multiexp straight from device memory (no host round trip, no serial Fr -> Exponent pass)
  auto enqueue_h_block = [&] {
    if (src.host) {
      // EvaluationDomain::from_coeffs pads with zeros (domain.rs:68): the padding is written on the device
      const Fr *ev[3] = {src.a, src.b, src.c};
      void *dst[3] = {da.p, db.p, dc.p};
      for (int i = 0; i < 3; i++) {
        if (m > n_cons) check(bh_dev_zero_on(ctx, (char *)dst[i] + n_cons * 32, (m - n_cons) * 32, ps.st));
        check(bh_dev_upload_on(ctx, dst[i], ev[i], n_cons * 32, ps.st));
      }
    } else {
      // a = A.w, b = B.w, c = C.w straight into the FFT buffers (prover.rs:19-55,105-145 on the device)
      check(bh_r1cs_eval_dev(ctx, src.r1cs->handle, d_in.p, d_aux.p, da.p, db.p, dc.p, log_m, ps.st));
    }
    check(bh_h_poly_fr_dev_on(ctx, da.p, db.p, dc.p, dscratch.p, log_m, ps.st));
  };
  // The seven multiexps that only need the assignments and the h block are independent: the order of issue is
  // unobservable (the order of waits is kept, prover.rs:339-354).  Large proofs issue the multiexps first, so the
  // GPU is busy while the host stages a/b/c; small proofs (a few launches of latency-bound kernels each) put the h
  // block's dozen kernels at the head of the hardware queues instead of behind ~100 multiexp launches.
  // A small proof is bound by the HOST: every job is ~20 kernel launches, ~0.2 ms of API time, and the H multiexp can
  // only be issued once the h block has run (profiles/archive/r2_call21_mimc_timeline.txt: five jobs issued one after the other,
  // the last at 0.97 ms of a 1.62 ms GPU span).  So the seven assignment multiexps are enqueued by a helper thread while
  // this one enqueues the h block, waits for it and issues H.
  double t1;
  if (log_m <= 16) {
    std::exception_ptr helper_error;
    std::thread helper([&] {
      try { issue_seven(true); } catch (...) { helper_error = std::current_exception(); }
    });
    try {
      enqueue_h_block();
      BH_TRACE("h block enqueued");
      check(bh_stream_synchronize(ctx, ps.st));
      t1 = now_ms();
      issue(params.h, 0, da.p, m - 1, nullptr, nullptr, &h_job);   // a.len() - 1, :238-244
    } catch (...) {
      helper.join();
      throw;
    }
    helper.join();
    if (helper_error) std::rethrow_exception(helper_error);
  } else {
    // The H multiexp is ordered after the h block by an event (bh_msm_async_dev_after): the host never waits between
    // them.  With the constraints evaluated on the device the h block is enqueued FIRST - nothing on the host delays it,
    // and started late it queues behind the multiexps' long kernels, the H multiexp then runs alone at the end
    // (profiles/archive/r3_call2_proof_timeline.txt).  With host evaluations the multiexps go first: the GPU works while the
    // host stages a, b, c (96 MiB of pageable memory at 2^20).
    auto issue_h = [&] {
      const Slice sl = slice_of(m - 1, part, parts);   // a.len() - 1, :238-244
      const bh_msm_opts held = {0, 0, hold_jobs ? BH_MSM_HOLD : 0u};
      check(bh_msm_async_dev_after(ctx, params.h, sl.lo, (const char *)da.p + sl.lo * 32, sl.hi - sl.lo, BH_SCALARS_MONT, nullptr,
                                   0, &held, ps.st, &h_job));
    };
    // Issue order = order of the bucket accumulations on the device (the accumulation chain, common.hpp): the G2
    // multiexp first - the longest job, its reduction tail then runs beside the G1 accumulations - and H last, by which
    // time the h block (enqueued first when nothing on the host delays it) has long finished beside the others.
    if (!src.host) {
      enqueue_h_block();
      // the accumulations wait for the h block (their digit / sort stages run beside it): behind the G2 accumulation's
      // 5 ms workgroups an FFT pass waits for a free SIMD, and the h block took 17 ms (profiles/archive/r3_call4_proof_timeline.txt)
      check(bh_ctx_accumulations_after(ctx, ps.st));
    }
    issue_seven(true);
    if (src.host) enqueue_h_block();
    issue_h();
    if (hold_jobs)   // second phase, in chain order
      for (bh_msm_job *j : {b2_aux_job, l_job, a_aux_job, b1_aux_job, h_job, a_in_job, b1_in_job, b2_in_job}) check(bh_msm_start(j));
    BH_TRACE("7 multiexps + h block + H issued; n_cons=%zu m=%zu", n_cons, m);
    t1 = now_ms();
  }

  BH_TRACE("all msm issued n_in=%zu n_aux=%zu", n_in, n_aux);
  if (while_running) (*while_running)();
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
// prover.rs:320-338: the part of the proof elements that depends on the verifying key and r, s only
struct ProofBlinding {
  G1Affine g_a, g_c;
  G2Affine g_b;
};
static ProofBlinding blind_terms(const Parameters &params, const Fr &r, const Fr &s) {
  const VerifyingKey &vk = params.vk;
  if (vk.delta_g1.is_identity() || vk.delta_g2.is_identity())   // subversion check, prover.rs:320-324
    throw SynthesisError(BH_ERR_UNEXPECTED_IDENTITY, "UnexpectedIdentity");
  ProofBlinding b;
  b.g_a = add_pts(BH_G1, mul_pt(BH_G1, vk.delta_g1, r), vk.alpha_g1);   // :326-327
  b.g_b = add_pts(BH_G2, mul_pt(BH_G2, vk.delta_g2, s), vk.beta_g2);    // :328-329
  const Fr rs = r * s;
  b.g_c = mul_pt(BH_G1, vk.delta_g1, rs);                                // :331-338
  b.g_c = add_pts(BH_G1, b.g_c, mul_pt(BH_G1, vk.alpha_g1, s));
  b.g_c = add_pts(BH_G1, b.g_c, mul_pt(BH_G1, vk.beta_g1, r));
  return b;
}
// prover.rs:339-360: fold the multiexp results in
static Proof finish_proof(const ProofBlinding &b, const MsmSums &m, const Fr &r, const Fr &s) {
  // g_a = blind + a_answer; g_b = blind + b_g2_answer; g_c = blind + [s] a_answer + [r] b_g1_answer + h + l with
  // a_answer = a_inputs + a_aux etc. - each as ONE host linear combination (shared doubling chain, one inversion)
  const uint64_t one[4] = {1, 0, 0, 0};
  uint64_t rc[4], sc[4];
  r.to_canonical(rc);
  s.to_canonical(sc);
  Proof p;
  {
    const G1Affine pts[3] = {b.g_a, m.a_in, m.a_aux};                                              // :339-343
    bh_point_lincomb(BH_G1, &p.a, pts, nullptr, 3);
  }
  {
    const G2Affine pts[3] = {b.g_b, m.b2_in, m.b2_aux};                                            // :347-350
    bh_point_lincomb(BH_G2, &p.b, pts, nullptr, 3);
  }
  {
    const G1Affine pts[7] = {b.g_c, m.a_in, m.a_aux, m.b1_in, m.b1_aux, m.h, m.l};                 // :342, 351-354
    uint64_t ks[7][4];
    const uint64_t *src[7] = {one, sc, sc, rc, rc, one, one};
    for (int i = 0; i < 7; i++) memcpy(ks[i], src[i], 32);
    bh_point_lincomb(BH_G1, &p.c, pts, ks, 7);
  }
  return p;
}
Proof assemble_proof(const Parameters &params, const MsmSums &m, const Fr &r, const Fr &s) {
  return finish_proof(blind_terms(params, r, s), m, r, s);
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
  // the five scalar multiplications that involve only the verifying key, r and s (prover.rs:326-338; ~0.9 ms of host
  // arithmetic) run on their own host thread from the start, beside the enqueueing of the multiexps and the GPU
  // work: for a small proof they are as long as everything else together.  A failure there (identity delta) is
  // reported after the jobs drain.
  ProofBlinding blind;
  std::exception_ptr blind_err;
  std::thread blind_thread([&] {
    try { blind = blind_terms(params, r, s); } catch (...) { blind_err = std::current_exception(); }
  });
  std::exception_ptr msm_err;
  try {
    msm_sums(src, params, 0, 1, sums, tm, nullptr);
  } catch (...) {
    msm_err = std::current_exception();   // every job has been drained by now (JobSet)
  }
  blind_thread.join();
  // prover.rs:320-324 returns UnexpectedIdentity for an identity delta BEFORE any multiexp is waited on: that error
  // takes precedence over whatever a multiexp reports (e.g. UnexpectedEof from a short query)
  if (blind_err) std::rethrow_exception(blind_err);
  if (msm_err) std::rethrow_exception(msm_err);
  Proof p = finish_proof(blind, sums, r, s);
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
  AssignmentView v;
  v.a = prover.a.data(); v.b = prover.b.data(); v.c = prover.c.data(); v.n_constraints = prover.a.size();
  v.input_assignment = prover.input_assignment.data(); v.n_inputs = prover.input_assignment.size();
  v.aux_assignment = prover.aux_assignment.data(); v.n_aux = prover.aux_assignment.size();
  v.a_aux_density = prover.a_aux_density.words(); v.b_input_density = prover.b_input_density.words();
  v.b_aux_density = prover.b_aux_density.words();
  return prove_assignment(v, params, r, s, tm);
}
Proof prove_assignment(const AssignmentView &v, Parameters &params, const Fr &r, const Fr &s, ProveTimings *tm) {
  AssignmentSource src;
  src.inputs = v.input_assignment; src.n_in = v.n_inputs;
  src.aux = v.aux_assignment; src.n_aux = v.n_aux;
  src.n_cons = v.n_constraints;
  src.host = true;
  src.a = v.a; src.b = v.b; src.c = v.c;
  src.a_aux_density = v.a_aux_density; src.b_input_density = v.b_input_density; src.b_aux_density = v.b_aux_density;
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
  // the matrices in the layout bh_r1cs_create takes (CSR: row_ptr, variable, coefficient-table index), written once.  A
  // variable is stored as its index with AUX_BIT for an aux variable until the capture ends: only then is the number
  // of inputs known, and `finish` turns every entry into its column (inputs first, then aux) in place.
  static constexpr uint32_t AUX_BIT = 0x80000000u;
  std::vector<uint32_t> row_ptr[3], var[3], coeff[3];
  static uint32_t pack(const Variable &v) { return (uint32_t)v.idx() | (v.kind() == Index::Aux ? AUX_BIT : 0u); }
  void finish() {
    const uint32_t ni = (uint32_t)num_inputs;
    for (auto &vm : var)
      for (uint32_t &x : vm) x = (x & AUX_BIT) ? ni + (x & ~AUX_BIT) : x;
  }
  std::vector<Fr> coeffs;
  // Coefficients are shared through a direct-mapped cache of table indices, not an exact map: a circuit's constants
  // (round constants, powers of two) repeat and hit; a circuit whose coefficients are all different (the synthetic
  // chain: 2 new ones per constraint) pays one compare and one append per term instead of a node allocation and a
  // rehash - the one-time capture of 2^20 constraints 1.32 -> 0.42 s in the build container (bh_test_capture_check).  A collision only stores a
  // constant twice (the table may hold duplicates: terms carry an index, csrc/r1cs.hip).
  // A cache entry is (32 hash bits | table index): the stored coefficient is only looked at when the hash bits agree - for a
  // circuit of distinct coefficients every probe would otherwise be a cache miss into a table of tens of megabytes.
  static constexpr size_t CACHE_SLOTS = size_t(1) << 16;
  std::vector<uint64_t> coeff_cache;
  const Fr one_ = Fr::one();
  ShapeAssembly() : coeff_cache(CACHE_SLOTS, 0) {   // 0 = the constant one (never looked up): an empty slot
    coeffs.push_back(Fr::one());
    for (auto &rp : row_ptr) rp.push_back(0);
  }
  Variable alloc(ValueFn) override { return Variable::new_unchecked(Index::Aux, num_aux++); }
  Variable alloc_input(ValueFn) override { return Variable::new_unchecked(Index::Input, num_inputs++); }
  void add_term(int m, const Variable &v, const Fr &k) {
    if (k.is_zero()) return;     // prover.rs:31: no value, no density
    if (k == one_) {             // the usual `lc + x` term: coefficient table entry 0, no hash lookup
      var[m].push_back(pack(v)); coeff[m].push_back(0);
      return;
    }
    const uint64_t h = FrHash()(k);
    uint64_t &entry = coeff_cache[h & (CACHE_SLOTS - 1)];
    const uint64_t tag = h & 0xffffffff00000000ULL;
    uint32_t slot = (uint32_t)entry;
    if (slot == 0 || (entry & 0xffffffff00000000ULL) != tag || !(coeffs[slot] == k)) {
      slot = (uint32_t)coeffs.size();
      entry = tag | slot;
      coeffs.push_back(k);
    }
    var[m].push_back(pack(v)); coeff[m].push_back(slot);
  }
  struct Hooked { ShapeAssembly *cs; int m; };
  static void hook(void *self, Variable v, const Fr &k) {
    Hooked *h = static_cast<Hooked *>(self);
    h->cs->add_term(h->m, v, k);
  }
  void enforce(const LcFn &fa, const LcFn &fb, const LcFn &fc) override {
    // the closures get combinations whose terms go straight into the matrices (LcSink::hook); a closure that returns
    // some other, stored combination is walked the classic way
    const LcFn *fs[3] = {&fa, &fb, &fc};
    for (int m = 0; m < 3; m++) {
      Hooked h{this, m};
      LcSink sink{nullptr, nullptr, nullptr, nullptr, &ShapeAssembly::hook, &h};
      const LinearCombination r = (*fs[m])(LinearCombination::evaluating(&sink));
      if (sink.pushed && (!r.is_evaluating() || r.size() != sink.pushed))   // (checking builds: see ProvingAssignment::enforce)
        throw std::invalid_argument("enforce: the closure added terms to a copy of its argument that it did not return");
      if (!r.is_evaluating())
        for (size_t i = 0; i < r.size(); i++) add_term(m, r[i].first, r[i].second);
      row_ptr[m].push_back((uint32_t)var[m].size());
    }
  }
};
}  // namespace

// the circuit's shape with the input rows of prover.rs:208-215 appended
static void capture_shape(Circuit &shape_of, ShapeAssembly &cs) {
  cs.alloc_input([] { return Fr::one(); });
  shape_of.synthesize(cs);
  for (size_t i = 0; i < cs.num_inputs; i++) {
    cs.enforce([i](LinearCombination lc) { return lc + Variable::new_unchecked(Index::Input, i); },
               [](LinearCombination lc) { return lc; }, [](LinearCombination lc) { return lc; });
  }
  if (cs.num_inputs + cs.num_aux >= ShapeAssembly::AUX_BIT)   // columns are 32-bit on the device (csrc/r1cs.hip)
    throw std::invalid_argument("R1CS capture: more than 2^31 variables");
  cs.finish();
}
// host-only self check (bh_test_capture_check): the captured matrices times the assignment a ProvingAssignment
// computes for the same circuit must give that assignment's a, b, c rows.  out4 = [constraints, terms, coefficients in
// the table, rows that differ]; returns the capture time in ms.
double capture_check_for_tests(Circuit &shape_of, Circuit &proved, size_t out4[4]) {
  const double t0 = now_ms();
  ShapeAssembly cs;
  capture_shape(shape_of, cs);
  const double ms = now_ms() - t0;
  ProvingAssignment pa;
  pa.alloc_input([] { return Fr::one(); });
  proved.synthesize(pa);
  for (size_t i = 0; i < pa.input_assignment.size(); i++) {
    pa.enforce([i](LinearCombination lc) { return lc + Variable::new_unchecked(Index::Input, i); },
               [](LinearCombination lc) { return lc; }, [](LinearCombination lc) { return lc; });
  }
  const size_t rows = cs.row_ptr[0].size() - 1;
  size_t bad = (rows != pa.a.size() || cs.num_inputs != pa.input_assignment.size() || cs.num_aux != pa.aux_assignment.size()) ? 1 : 0;
  const std::vector<Fr> *want[3] = {&pa.a, &pa.b, &pa.c};
  for (int m = 0; m < 3 && !bad; m++) {
    for (size_t r = 0; r < rows; r++) {
      Fr acc = Fr::zero();
      for (uint32_t t = cs.row_ptr[m][r]; t < cs.row_ptr[m][r + 1]; t++) {
        const uint32_t col = cs.var[m][t];   // (after finish: inputs first, then aux)
        const Fr &v = col < cs.num_inputs ? pa.input_assignment[col] : pa.aux_assignment[col - cs.num_inputs];
        acc = acc + cs.coeffs[cs.coeff[m][t]] * v;
      }
      if (!(acc == (*want[m])[r])) bad++;
    }
  }
  out4[0] = rows; out4[1] = cs.var[0].size() + cs.var[1].size() + cs.var[2].size(); out4[2] = cs.coeffs.size(); out4[3] = bad;
  return ms;
}

R1cs::R1cs(Circuit &shape_of, bh_ctx *ctx) {
  ShapeAssembly cs;
  capture_shape(shape_of, cs);
  num_inputs = cs.num_inputs; num_aux = cs.num_aux; num_constraints = cs.row_ptr[0].size() - 1;
  bh_csr abc[3];
  for (int m = 0; m < 3; m++) abc[m] = bh_csr{cs.row_ptr[m].data(), cs.var[m].data(), cs.coeff[m].data()};
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
  Recycled<WitnessAssignment> wr(g_witnesses);
  WitnessAssignment &w = *wr;
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

// ---- one caller, proofs back to back (groth16.hpp) ---------------------------------------------------------------
Proof AsyncProof::wait(ProveTimings *tm) {
  if (worker.joinable()) worker.join();
  if (tm) *tm = timings;
  if (error) std::rethrow_exception(error);
  return proof;
}
std::unique_ptr<AsyncProof> create_proof_async(Circuit &circuit, const R1cs *r1cs, Parameters &params, const Fr &r, const Fr &s) {
  std::unique_ptr<AsyncProof> job(new AsyncProof());
  const double t0 = now_ms();
  if (r1cs) {
    job->witness = g_witnesses.get();
    WitnessAssignment &w = *job->witness;
    w.input_assignment.reserve(r1cs->num_inputs);
    w.aux_assignment.reserve(r1cs->num_aux);
    w.alloc_input([] { return Fr::one(); });
    circuit.synthesize(w);
  } else {
    job->assignment = g_assignments.get();
    ProvingAssignment &pa = *job->assignment;
    pa.alloc_input([] { return Fr::one(); });
    circuit.synthesize(pa);
    for (size_t i = 0; i < pa.input_assignment.size(); i++) {
      pa.enforce([i](LinearCombination lc) { return lc + Variable::new_unchecked(Index::Input, i); },
                 [](LinearCombination lc) { return lc; }, [](LinearCombination lc) { return lc; });
    }
  }
  const float synth_ms = (float)(now_ms() - t0);
  AsyncProof *j = job.get();
  Parameters *pp = &params;
  job->worker = std::thread([j, r1cs, pp, r, s, synth_ms] {
    try {
      ProveTimings local = {0, 0, 0, 0};
      if (r1cs) {
        const WitnessAssignment &w = *j->witness;
        j->proof = prove_witness(*r1cs, *pp, w.input_assignment.data(), w.input_assignment.size(), w.aux_assignment.data(),
                                 w.aux_assignment.size(), r, s, &local);
      } else {
        j->proof = prove_assignment(*j->assignment, *pp, r, s, &local);
      }
      j->timings = local;
      j->timings.synthesis_ms = synth_ms;
      j->timings.total_ms = local.total_ms + synth_ms;
    } catch (...) {
      j->error = std::current_exception();
    }
  });
  return job;
}
std::unique_ptr<AsyncProof> prove_assignment_async(const AssignmentView &v, Parameters &params, const Fr &r, const Fr &s) {
  std::unique_ptr<AsyncProof> job(new AsyncProof());
  AsyncProof *j = job.get();
  Parameters *pp = &params;
  job->worker = std::thread([j, v, pp, r, s] {
    try {
      ProveTimings local = {0, 0, 0, 0};
      j->proof = prove_assignment(v, *pp, r, s, &local);
      j->timings = local;
    } catch (...) {
      j->error = std::current_exception();
    }
  });
  return job;
}
std::unique_ptr<AsyncProof> prove_witness_async(const R1cs &r1cs, Parameters &params, const void *input_assignment, size_t n_inputs,
                                                const void *aux_assignment, size_t n_aux, const Fr &r, const Fr &s) {
  std::unique_ptr<AsyncProof> job(new AsyncProof());
  job->witness = g_witnesses.get();
  WitnessAssignment &w = *job->witness;
  w.input_assignment.resize(n_inputs);
  w.aux_assignment.resize(n_aux);
  if (n_inputs) memcpy((void *)w.input_assignment.data(), input_assignment, n_inputs * sizeof(Fr));
  if (n_aux) memcpy((void *)w.aux_assignment.data(), aux_assignment, n_aux * sizeof(Fr));
  AsyncProof *j = job.get();
  const R1cs *rp = &r1cs;
  Parameters *pp = &params;
  job->worker = std::thread([j, rp, pp, r, s] {
    try {
      ProveTimings local = {0, 0, 0, 0};
      const WitnessAssignment &w = *j->witness;
      j->proof = prove_witness(*rp, *pp, w.input_assignment.data(), w.input_assignment.size(), w.aux_assignment.data(),
                               w.aux_assignment.size(), r, s, &local);
      j->timings = local;
    } catch (...) {
      j->error = std::current_exception();
    }
  });
  return job;
}
void ProofPipeline::retire_oldest() {
  std::unique_ptr<AsyncProof> j = std::move(inflight_.front());
  inflight_.pop_front();
  Done d;
  d.tm = ProveTimings{0, 0, 0, 0};
  try { d.proof = j->wait(&d.tm); } catch (...) { d.error = std::current_exception(); }
  done_.push_back(std::move(d));
}
void ProofPipeline::submit(Circuit &circuit, const Fr &r, const Fr &s) {
  while (inflight_.size() >= depth_) retire_oldest();
  inflight_.push_back(create_proof_async(circuit, r1cs_, params_, r, s));   // synthesis runs here, beside the proofs in flight
}
Proof ProofPipeline::next(ProveTimings *tm) {
  if (done_.empty()) {
    if (inflight_.empty()) throw std::logic_error("ProofPipeline::next without a submitted proof");
    retire_oldest();
  }
  Done d = std::move(done_.front());
  done_.pop_front();
  if (tm) *tm = d.tm;
  if (d.error) std::rethrow_exception(d.error);
  return d.proof;
}

// ---- prover.rs:182-215 ----------------------------------------------------------------------------
Proof create_proof(Circuit &circuit, Parameters &params, const Fr &r, const Fr &s, ProveTimings *tm) {
  const double t0 = now_ms();
  Recycled<ProvingAssignment> pr(g_assignments);
  ProvingAssignment &prover = *pr;
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

}  // namespace groth16

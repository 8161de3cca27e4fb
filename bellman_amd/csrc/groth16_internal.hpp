// Internals shared by the groth16_*.cpp translation units (not part of the public mirror, groth16.hpp).
#pragma once
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <functional>
#include <memory>
#include <stdexcept>

#include "groth16.hpp"

#define BH_TRACE(...) do { if (getenv("BH_DEBUG")) { fprintf(stderr, "[groth16 %.2f ms] ", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count()); fprintf(stderr, __VA_ARGS__); fputc(10, stderr); fflush(stderr); } } while (0)

namespace groth16 {
namespace detail {

// C-ABI return code -> the exception the mirror throws (src/lib.rs:303-319); HIP failures are never
// turned into a CPU path
inline void check(int rc) {
  switch (rc) {
    case BH_OK: return;
    case BH_ERR_UNEXPECTED_IDENTITY: throw bellman::SynthesisError(rc, "UnexpectedIdentity");
    case BH_ERR_UNEXPECTED_EOF: throw bellman::SynthesisError(rc, "IoError(UnexpectedEof): expected more bases from source");
    case BH_ERR_DEGREE_TOO_LARGE: throw bellman::SynthesisError(rc, "PolynomialDegreeTooLarge");
    default: throw std::runtime_error("bellman_hip: HIP/runtime failure (no CPU fallback)");
  }
}

struct DevBuf {   // a device buffer from the context's pool, returned on scope exit
  bh_ctx *ctx;
  void *p = nullptr;
  DevBuf(bh_ctx *c, size_t bytes) : ctx(c) { check(bh_dev_alloc(ctx, bytes, &p)); }
  ~DevBuf() { if (p) bh_dev_free(ctx, p); }
  DevBuf(const DevBuf &) = delete;
};

inline double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace detail
void proof_slice_for_tests(size_t n, size_t part, size_t parts, size_t *lo, size_t *hi);   // groth16_prover.cpp
double capture_check_for_tests(bellman::Circuit &shape_of, bellman::Circuit &proved, size_t out4[4]);   // groth16_prover.cpp
}  // namespace groth16

// ---- pieces shared by the C entry points of the product (groth16_capi.cpp) and of the test library (demo_circuits.cpp)
struct bh_params {
  groth16::Parameters *p;
};

inline int run_guarded_sums(const std::function<groth16::MsmSums()> &f, void *sums_out) {
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

inline int run_guarded(const std::function<groth16::Proof()> &f, void *proof_out) {
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

struct bh_proof_job {
  // members are destroyed in reverse order: the job (whose destructor joins the helper thread) before the view of the
  // constraint matrices that thread reads
  std::unique_ptr<R1csView> view;
  std::unique_ptr<groth16::AsyncProof> job;
  int early_rc = BH_OK;
};

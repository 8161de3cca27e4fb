// Internals shared by the groth16_*.cpp translation units (not part of the public mirror, groth16.hpp).
#pragma once
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
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

// R1CS matrices resident in HBM + the evaluation of all constraints' linear combinations on the
// device.  The reference evaluates the three LinearCombinations of every
// constraint on one host thread while the circuit is synthesised
// (/root/reference/groth16/src/prover.rs:19-55 `eval`, :105-145 `enforce`).  The matrices do not
// depend on the witness, so they are captured once per circuit (like the CRS) and a proof only needs
//   a[i] = <A_i, w>,  b[i] = <B_i, w>,  c[i] = <C_i, w>      (w = input_assignment | aux_assignment)
// which is one sparse matrix-vector product per matrix: one lane per (matrix, constraint) row.
// The query densities (prover.rs:31-44: a variable counts only through a non-zero coefficient) are a
// function of the matrices alone and are computed here once.
#include <string.h>

#include <mutex>
#include <vector>

#include "common.hpp"

using namespace bh;

struct bh_r1cs {
  bh_ctx *ctx = nullptr;
  size_t n_inputs = 0, n_aux = 0, n_constraints = 0;
  u32 *row_ptr[3] = {nullptr, nullptr, nullptr};    // [n_constraints + 1]
  uint2 *terms[3] = {nullptr, nullptr, nullptr};    // (variable, coefficient index)
  fr_t *coeffs = nullptr;                           // Montgomery; index 0 is always 1
  u64 *dens[3] = {nullptr, nullptr, nullptr};       // a_aux, b_input, b_aux (LSB0 words, device)
  size_t dens_total[3] = {0, 0, 0};
  std::vector<u64> dens_host[3];
  // host copy of the matrices + the transposed (variable-major) device copy the parameter generator
  // uses (generator.rs:43-131 stores exactly that: per variable, (coeff, constraint) lists); built on
  // first use
  uint2 *long_rows = nullptr;                        // (matrix, row) of rows with more than LONG_ROW terms
  u32 n_long = 0;
  uint2 *t_long_rows = nullptr;                      // the same for the transposed matrices
  u32 t_n_long = 0;
  std::vector<u32> h_row_ptr[3], h_var[3], h_coeff[3];
  std::mutex t_mu;
  bool t_ready = false;
  u32 *t_row_ptr[3] = {nullptr, nullptr, nullptr};   // [n_inputs + n_aux + 1]
  uint2 *t_terms[3] = {nullptr, nullptr, nullptr};   // (constraint, coefficient index)
};

namespace {

struct R1csEvalArgs {
  const u32 *row_ptr[3];
  const uint2 *terms[3];
  fr_t *out[3];
  const fr_t *coeffs, *inputs, *aux;
  u32 n_inputs;
  u64 n_constraints, m;   // rows >= n_constraints are the zero padding of from_coeffs (domain.rs:68)
  const uint2 *long_rows; // rows the lane-per-row kernel leaves to r1cs_long_rows_kernel
};
// A row with more terms than this is summed by a whole workgroup instead of one lane: the constant
// ONE typically appears in every constraint, so its row of a transposed matrix has ~n terms.
constexpr u32 LONG_ROW = 1024;

__device__ __forceinline__ fr_t ld_fr16(const fr_t *p) {
  const uint4 *q = reinterpret_cast<const uint4 *>(p);
  uint4 lo = q[0], hi = q[1];
  fr_t r;
  r.l[0] = lo.x; r.l[1] = lo.y; r.l[2] = lo.z; r.l[3] = lo.w;
  r.l[4] = hi.x; r.l[5] = hi.y; r.l[6] = hi.z; r.l[7] = hi.w;
  return r;
}
__device__ __forceinline__ void st_fr16(fr_t *p, const fr_t &r) {
  uint4 *q = reinterpret_cast<uint4 *>(p);
  q[0] = make_uint4(r.l[0], r.l[1], r.l[2], r.l[3]);
  q[1] = make_uint4(r.l[4], r.l[5], r.l[6], r.l[7]);
}

__global__ void __launch_bounds__(256) r1cs_eval_kernel(R1csEvalArgs a) {
  const u64 total = 3 * a.m;
  for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (u64)gridDim.x * blockDim.x) {
    const u32 mat = (u32)(g / a.m);
    const u64 row = g - (u64)mat * a.m;
    fr_t acc;
    fe_zero(acc);
    if (row < a.n_constraints) {
      const u32 lo = a.row_ptr[mat][row], hi = a.row_ptr[mat][row + 1];
      if (hi - lo > LONG_ROW) continue;   // written by r1cs_long_rows_kernel
      for (u32 t = lo; t < hi; t++) {
        const uint2 term = a.terms[mat][t];
        fr_t w = ld_fr16(term.x < a.n_inputs ? a.inputs + term.x : a.aux + (term.x - a.n_inputs));
        if (term.y != 0) {           // coefficient 1 is by far the most common one (prover.rs:47-52)
          const fr_t k = ld_fr16(a.coeffs + term.y);
          fe_mul(w, w, k);
        }
        fe_add(acc, acc, w);
      }
    }
    st_fr16(a.out[mat] + row, acc);
  }
}

__global__ void __launch_bounds__(256) qap_ext_kernel(fr_t *e, const fr_t *at, const fr_t *bt, const fr_t *ct, u64 n_inputs,
                                                      u64 n, fr_t alpha, fr_t beta, fr_t gamma_inv, fr_t delta_inv) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
    fr_t a = ld_fr16(at + i), b = ld_fr16(bt + i), c = ld_fr16(ct + i);
    fe_mul(a, a, beta);
    fe_mul(b, b, alpha);
    fe_add(a, a, b);
    fe_add(a, a, c);
    if (i < n_inputs) fe_mul(a, a, gamma_inv); else fe_mul(a, a, delta_inv);
    st_fr16(e + i, a);
  }
}

__global__ void __launch_bounds__(256) r1cs_long_rows_kernel(R1csEvalArgs a) {
  __shared__ fr_t part[256];
  const uint2 job = a.long_rows[blockIdx.x];
  const u32 mat = job.x, row = job.y;
  const u32 lo = a.row_ptr[mat][row], hi = a.row_ptr[mat][row + 1];
  fr_t acc;
  fe_zero(acc);
  for (u32 t = lo + threadIdx.x; t < hi; t += 256) {
    const uint2 term = a.terms[mat][t];
    fr_t w = ld_fr16(term.x < a.n_inputs ? a.inputs + term.x : a.aux + (term.x - a.n_inputs));
    if (term.y != 0) {
      const fr_t k = ld_fr16(a.coeffs + term.y);
      fe_mul(w, w, k);
    }
    fe_add(acc, acc, w);
  }
  part[threadIdx.x] = acc;
  for (u32 off = 128; off >= 1; off >>= 1) {
    __syncthreads();
    if (threadIdx.x < off) {
      fr_t x = part[threadIdx.x], y = part[threadIdx.x + off];
      fe_add(x, x, y);
      part[threadIdx.x] = x;
    }
  }
  if (threadIdx.x == 0) st_fr16(a.out[mat] + row, part[0]);
}

static std::vector<uint2> find_long_rows(const std::vector<u32> *row_ptr_of_3) {
  std::vector<uint2> v;
  for (u32 m = 0; m < 3; m++)
    for (size_t r = 0; r + 1 < row_ptr_of_3[m].size(); r++)
      if (row_ptr_of_3[m][r + 1] - row_ptr_of_3[m][r] > LONG_ROW) v.push_back(make_uint2(m, (u32)r));
  return v;
}

static int launch_eval(bh_ctx *ctx, R1csEvalArgs &a, u32 n_long, hipStream_t st) {
  const u64 blocks = (3 * a.m + 255) / 256, cap = (u64)ctx->c.num_cus * 16;
  hipLaunchKernelGGL(r1cs_eval_kernel, dim3((u32)(blocks < cap ? blocks : cap)), dim3(256), 0, st, a);
  BH_HIP_CHECK(hipGetLastError());
  if (n_long) {
    hipLaunchKernelGGL(r1cs_long_rows_kernel, dim3(n_long), dim3(256), 0, st, a);
    BH_HIP_CHECK(hipGetLastError());
  }
  return BH_OK;
}

template <class T>
int upload_vec(bh_ctx *ctx, T **dst, const T *src, size_t n) {
  *dst = (T *)ctx->c.pool.acquire((n ? n : 1) * sizeof(T));
  if (!*dst) return BH_ERR_HIP;
  if (n) BH_HIP_CHECK(hipMemcpyAsync(*dst, src, n * sizeof(T), hipMemcpyHostToDevice, ctx->c.stream));
  return BH_OK;
}

}  // namespace

extern "C" {

int bh_r1cs_create(bh_ctx *ctx, size_t n_inputs, size_t n_aux, size_t n_constraints, const bh_csr abc[3],
                   const void *coeffs, size_t n_coeffs, bh_r1cs **out) {
  if (!ctx || !out || !abc || !coeffs || n_coeffs == 0 || n_inputs + n_aux >= (size_t(1) << 32) ||
      n_constraints >= (size_t(1) << 32))
    return BH_ERR_INVALID_ARG;
  fr_t one;
  fe_one(one);
  if (memcmp(coeffs, &one, 32) != 0) return BH_ERR_INVALID_ARG;   // slot 0 is the constant 1
  for (int m = 0; m < 3; m++) {
    if (!abc[m].row_ptr || abc[m].row_ptr[0] != 0) return BH_ERR_INVALID_ARG;
    const u32 nnz = abc[m].row_ptr[n_constraints];
    if (nnz && (!abc[m].var || !abc[m].coeff)) return BH_ERR_INVALID_ARG;
    for (size_t i = 0; i < n_constraints; i++)
      if (abc[m].row_ptr[i] > abc[m].row_ptr[i + 1]) return BH_ERR_INVALID_ARG;
    for (u32 t = 0; t < nnz; t++)
      if (abc[m].var[t] >= n_inputs + n_aux || abc[m].coeff[t] >= n_coeffs) return BH_ERR_INVALID_ARG;
  }
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  bh_r1cs *r = new bh_r1cs;
  r->ctx = ctx;
  r->n_inputs = n_inputs; r->n_aux = n_aux; r->n_constraints = n_constraints;
  // densities: prover.rs:119-141 (A: aux only, inputs are always fully dense; B: inputs and aux; C: none),
  // zero coefficients do not count (prover.rs:31)
  const fr_t *cf = (const fr_t *)coeffs;
  std::vector<uint8_t> coeff_is_zero(n_coeffs);
  for (size_t i = 0; i < n_coeffs; i++) {
    fr_t c;
    memcpy(&c, cf + i, 32);
    coeff_is_zero[i] = fe_is_zero(c);
  }
  r->dens_host[0].assign((n_aux + 63) / 64 + 1, 0);
  r->dens_host[1].assign((n_inputs + 63) / 64 + 1, 0);
  r->dens_host[2].assign((n_aux + 63) / 64 + 1, 0);
  auto mark = [&](int which, size_t idx) { r->dens_host[which][idx >> 6] |= u64(1) << (idx & 63); };
  for (int m = 0; m < 2; m++) {
    const u32 nnz = abc[m].row_ptr[n_constraints];
    for (u32 t = 0; t < nnz; t++) {
      if (coeff_is_zero[abc[m].coeff[t]]) continue;
      const u32 v = abc[m].var[t];
      if (v >= n_inputs) mark(m == 0 ? 0 : 2, v - n_inputs);
      else if (m == 1) mark(1, v);
    }
  }
  int rc = BH_OK;
  for (int d = 0; d < 3 && rc == BH_OK; d++) {
    for (u64 w : r->dens_host[d]) r->dens_total[d] += (size_t)__builtin_popcountll(w);
    rc = upload_vec(ctx, &r->dens[d], r->dens_host[d].data(), r->dens_host[d].size());
  }
  for (int m = 0; m < 3 && rc == BH_OK; m++) {
    const u32 nnz = abc[m].row_ptr[n_constraints];
    std::vector<uint2> terms(nnz);
    for (u32 t = 0; t < nnz; t++) terms[t] = make_uint2(abc[m].var[t], abc[m].coeff[t]);
    r->h_row_ptr[m].assign(abc[m].row_ptr, abc[m].row_ptr + n_constraints + 1);
    r->h_var[m].assign(abc[m].var, abc[m].var + nnz);
    r->h_coeff[m].assign(abc[m].coeff, abc[m].coeff + nnz);
    rc = upload_vec(ctx, &r->row_ptr[m], abc[m].row_ptr, n_constraints + 1);
    if (rc == BH_OK) rc = upload_vec(ctx, &r->terms[m], terms.data(), (size_t)nnz);
    if (rc == BH_OK && hipStreamSynchronize(ctx->c.stream) != hipSuccess) rc = BH_ERR_HIP;   // `terms` is a local
  }
  if (rc == BH_OK) {
    const std::vector<uint2> lr = find_long_rows(r->h_row_ptr);
    r->n_long = (u32)lr.size();
    rc = upload_vec(ctx, &r->long_rows, lr.data(), lr.size());
    if (rc == BH_OK && hipStreamSynchronize(ctx->c.stream) != hipSuccess) rc = BH_ERR_HIP;   // `lr` is a local
  }
  if (rc == BH_OK) rc = upload_vec(ctx, &r->coeffs, cf, n_coeffs);
  if (rc == BH_OK && hipStreamSynchronize(ctx->c.stream) != hipSuccess) rc = BH_ERR_HIP;
  if (rc != BH_OK) { bh_r1cs_release(r); return rc; }
  *out = r;
  return BH_OK;
}

void bh_r1cs_release(bh_r1cs *r) {
  if (!r) return;
  for (int m = 0; m < 3; m++) {
    r->ctx->c.pool.release(r->row_ptr[m]);
    r->ctx->c.pool.release(r->terms[m]);
    r->ctx->c.pool.release(r->dens[m]);
    r->ctx->c.pool.release(r->t_row_ptr[m]);
    r->ctx->c.pool.release(r->t_terms[m]);
  }
  r->ctx->c.pool.release(r->long_rows);
  r->ctx->c.pool.release(r->t_long_rows);
  r->ctx->c.pool.release(r->coeffs);
  delete r;
}

int bh_r1cs_shape(const bh_r1cs *r, size_t *n_inputs, size_t *n_aux, size_t *n_constraints) {
  if (!r) return BH_ERR_INVALID_ARG;
  if (n_inputs) *n_inputs = r->n_inputs;
  if (n_aux) *n_aux = r->n_aux;
  if (n_constraints) *n_constraints = r->n_constraints;
  return BH_OK;
}

int bh_r1cs_density(const bh_r1cs *r, int which, const uint64_t **dev_words, const uint64_t **host_words,
                    size_t *total) {
  if (!r || which < 0 || which > 2) return BH_ERR_INVALID_ARG;
  if (dev_words) *dev_words = r->dens[which];
  if (host_words) *host_words = r->dens_host[which].data();
  if (total) *total = r->dens_total[which];
  return BH_OK;
}

int bh_r1cs_eval_dev(bh_ctx *ctx, const bh_r1cs *r, const void *inputs_dev, const void *aux_dev, void *a_dev,
                     void *b_dev, void *c_dev, uint32_t log_m, void *stream) {
  if (!ctx || !r || log_m >= 32 || (size_t(1) << log_m) < r->n_constraints) return BH_ERR_INVALID_ARG;
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  R1csEvalArgs a;
  for (int m = 0; m < 3; m++) { a.row_ptr[m] = r->row_ptr[m]; a.terms[m] = r->terms[m]; }
  a.out[0] = (fr_t *)a_dev; a.out[1] = (fr_t *)b_dev; a.out[2] = (fr_t *)c_dev;
  a.coeffs = r->coeffs; a.inputs = (const fr_t *)inputs_dev; a.aux = (const fr_t *)aux_dev;
  a.n_inputs = (u32)r->n_inputs;
  a.n_constraints = r->n_constraints;
  a.m = u64(1) << log_m;
  a.long_rows = r->long_rows;
  return launch_eval(ctx, a, r->n_long, stream ? (hipStream_t)stream : ctx->c.stream);
}

// QAP polynomials at tau (generator.rs:369-387 eval_at_tau for every variable): at[v] = sum over the
// constraints j that use v in A of coeff * L_j(tau); the same sparse product as bh_r1cs_eval_dev on
// the transposed matrices, with the Lagrange coefficients in the role of the witness.
int bh_r1cs_eval_transposed_dev(bh_ctx *ctx, bh_r1cs *r, const void *lagrange_dev, void *at_dev, void *bt_dev,
                                void *ct_dev, void *stream) {
  if (!ctx || !r) return BH_ERR_INVALID_ARG;
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  const size_t n_vars = r->n_inputs + r->n_aux;
  {
    std::lock_guard<std::mutex> g(r->t_mu);
    if (!r->t_ready) {
      std::vector<u32> t_rp[3];
      for (int m = 0; m < 3; m++) {
        const size_t nnz = r->h_var[m].size();
        std::vector<u32> &rp = t_rp[m];
        rp.assign(n_vars + 1, 0);
        for (size_t t = 0; t < nnz; t++) rp[r->h_var[m][t] + 1]++;
        for (size_t v = 0; v < n_vars; v++) rp[v + 1] += rp[v];
        std::vector<uint2> terms(nnz);
        std::vector<u32> cursor(rp.begin(), rp.end() - 1);
        for (size_t row = 0; row < r->n_constraints; row++)
          for (u32 t = r->h_row_ptr[m][row]; t < r->h_row_ptr[m][row + 1]; t++)
            terms[cursor[r->h_var[m][t]]++] = make_uint2((u32)row, r->h_coeff[m][t]);
        int rc = upload_vec(ctx, &r->t_row_ptr[m], rp.data(), rp.size());
        if (rc == BH_OK) rc = upload_vec(ctx, &r->t_terms[m], terms.data(), nnz);
        if (rc == BH_OK && hipStreamSynchronize(ctx->c.stream) != hipSuccess) rc = BH_ERR_HIP;
        if (rc != BH_OK) return rc;
      }
      const std::vector<uint2> lr = find_long_rows(t_rp);
      r->t_n_long = (u32)lr.size();
      int rc = upload_vec(ctx, &r->t_long_rows, lr.data(), lr.size());
      if (rc == BH_OK && hipStreamSynchronize(ctx->c.stream) != hipSuccess) rc = BH_ERR_HIP;
      if (rc != BH_OK) return rc;
      r->t_ready = true;
    }
  }
  R1csEvalArgs a;
  for (int m = 0; m < 3; m++) { a.row_ptr[m] = r->t_row_ptr[m]; a.terms[m] = r->t_terms[m]; }
  a.out[0] = (fr_t *)at_dev; a.out[1] = (fr_t *)bt_dev; a.out[2] = (fr_t *)ct_dev;
  a.coeffs = r->coeffs; a.inputs = (const fr_t *)lagrange_dev; a.aux = (const fr_t *)lagrange_dev;
  a.n_inputs = 0;                       // every "variable" of the product is a constraint index
  a.n_constraints = n_vars;
  a.m = n_vars;
  if (!n_vars) return BH_OK;
  a.long_rows = r->t_long_rows;
  return launch_eval(ctx, a, r->t_n_long, stream ? (hipStream_t)stream : ctx->c.stream);
}

// generator.rs:400-407: e[v] = (at[v]*beta + bt[v]*alpha + ct[v]) * inv, inv = 1/gamma for the public
// inputs (ic) and 1/delta for the auxiliary variables (l)
int bh_fr_qap_ext_dev(bh_ctx *ctx, void *e_dev, const void *at_dev, const void *bt_dev, const void *ct_dev,
                      size_t n_inputs, size_t n_vars, const void *alpha, const void *beta, const void *gamma_inv,
                      const void *delta_inv, void *stream) {
  if (!ctx) return BH_ERR_INVALID_ARG;
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  if (!n_vars) return BH_OK;
  fr_t al, be, gi, di;
  memcpy(&al, alpha, 32); memcpy(&be, beta, 32); memcpy(&gi, gamma_inv, 32); memcpy(&di, delta_inv, 32);
  const u64 blocks = (n_vars + 255) / 256, cap = (u64)ctx->c.num_cus * 16;
  hipStream_t st = stream ? (hipStream_t)stream : ctx->c.stream;
  hipLaunchKernelGGL(qap_ext_kernel, dim3((u32)(blocks < cap ? blocks : cap)), dim3(256), 0, st, (fr_t *)e_dev,
                     (const fr_t *)at_dev, (const fr_t *)bt_dev, (const fr_t *)ct_dev, (u64)n_inputs, (u64)n_vars, al, be,
                     gi, di);
  BH_HIP_CHECK(hipGetLastError());
  return BH_OK;
}

}  // extern "C"

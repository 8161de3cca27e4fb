// Radix-2 number-theoretic FFT over BLS12-381 Fr for gfx950.
//
// Replaces bellman's EvaluationDomain transforms (src/domain.rs):
//   fft/ifft/coset_fft/icoset_fft (:81-125)  ->  ntt_run()
//   best_fft/serial_fft/parallel_fft (:261-372): any exact NTT produces the same canonical
//   field elements (Appendix A item 14 of SURVEY.md), so the CPU's log_cpus split is replaced
//   by a decomposition that fits the chip.
//   distribute_powers (:101-113), mul_assign (:154-170), sub_assign (:173-189),
//   divide_by_z_on_coset (:139-151)          ->  the element-wise kernels at the bottom,
//   plus the fused (a*b-c)*zinv pass used by create_proof's h block (prover.rs:232-236).
//
// Structure (MI355X-first): n = R_0 * R_1 * ... * R_{L-1} (R_p = 2^r_p <= 256, L <= 4).
// Pass p transforms, for every already-fixed prefix, the R_p-point sub-FFT along stride M_p
// entirely inside LDS (tile = R_p rows x C columns of 32-byte elements, 32 KiB), multiplies
// by the inter-pass twiddle w_N^(j'*k) and writes back to the same positions (pass 0 moves the
// data into a scratch vector, middle passes work in place there).  The last pass has no twiddle
// and scatters the digit-reversed result to its natural position back in the caller's vector,
// tiled over the FIRST digit so stores are contiguous 128-256-byte runs.
// Inside a tile the sub-FFT is a radix-2 DIT over LDS with the tile's twiddles (w_R^i)
// staged in LDS.  Work per pass: one 32-B read + one 32-B write per element.
#include "common.hpp"

namespace bh {

constexpr int NTT_THREADS = 256;
constexpr int NTT_LOG_TILE = 10;       // 1024 Fr = 32 KiB of LDS per workgroup (4 workgroups / CU)

struct NttPass {
  const fr_t *in;
  fr_t *out;
  const fr_t *tw;     // w_n^i for i < n
  const fr_t *pre;    // multiply input element i by pre[i]   (pass 0 only; may be null)
  const fr_t *post;   // multiply output element k by post[k] (last pass only; may be null)
  fr_t post_const;    // ... or by this constant when has_post_const
  u32 has_post_const;
  u32 log_n;
  u32 s;              // bits consumed by earlier passes
  u32 r;              // radix bits of this pass
  u32 log_c;          // log2(columns per tile)
  u32 inverse;        // use w^-e = w^(n-e)
  u32 is_last;
  u32 r0;             // radix bits of pass 0 (for the last pass' column dimension)
  u32 L;              // number of passes
  u32 rmid[2];        // radix bits of the middle passes (1 .. L-2)
};

__device__ __forceinline__ fr_t ld_fr(const fr_t *p) {
  fr_t v;
  const uint4 *q = reinterpret_cast<const uint4 *>(p);
  uint4 a = q[0], b = q[1];
  v.l[0] = a.x; v.l[1] = a.y; v.l[2] = a.z; v.l[3] = a.w;
  v.l[4] = b.x; v.l[5] = b.y; v.l[6] = b.z; v.l[7] = b.w;
  return v;
}
__device__ __forceinline__ void st_fr(fr_t *p, const fr_t &v) {
  uint4 *q = reinterpret_cast<uint4 *>(p);
  q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

__global__ __launch_bounds__(NTT_THREADS) void ntt_pass_kernel(NttPass a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const u32 R = 1u << a.r, C = 1u << a.log_c;
  fr_t *tile = reinterpret_cast<fr_t *>(smem);   // [C][R]
  fr_t *twl = tile + (size_t)R * C;              // [R/2] : w_R^i
  const u32 tid = threadIdx.x;
  const u32 n_mask = (a.log_n >= 32) ? 0xffffffffu : ((1u << a.log_n) - 1);
  const u64 n = (u64)1 << a.log_n;

  // ---- where is this tile? ---------------------------------------------------------------
  // non-last: element(row j, col c) at base + j*M + c            (M = n >> (s + r))
  // last:     element(row j, col c) at base + c*colstride + j    (rows contiguous)
  u64 base, in_row_stride, in_col_stride;
  u32 jp0 = 0;        // first column's j' (non-last passes)
  u64 out_base = 0, out_row_stride = 1, out_col_stride = 1;
  const u64 t = blockIdx.x;
  if (!a.is_last) {
    const u32 logM = a.log_n - a.s - a.r;
    const u64 tiles_per_block = ((u64)1 << logM) >> a.log_c;
    const u64 kprefix = t / tiles_per_block;
    jp0 = (u32)((t % tiles_per_block) << a.log_c);
    base = (kprefix << (a.log_n - a.s)) + jp0;
    in_row_stride = (u64)1 << logM;
    in_col_stride = 1;
    out_base = base; out_row_stride = in_row_stride; out_col_stride = 1;
  } else if (a.L == 1) {
    base = 0; in_row_stride = 1; in_col_stride = 0;
    out_base = 0; out_row_stride = 1; out_col_stride = 0;
  } else {
    // columns = C consecutive values of the FIRST digit k0; middle digits fixed by `mid`
    const u32 groups = (1u << a.r0) >> a.log_c;          // tiles per middle combination
    const u64 mid = t / groups;
    const u32 k00 = (u32)(t % groups) << a.log_c;
    const u64 M0 = n >> a.r0;
    base = (u64)k00 * M0 + (mid << a.r);
    in_row_stride = 1; in_col_stride = M0;
    // output index = k0 + R0*(k1 + R1*(k2 ...)) ; storage order of `mid` is (k1, k2) MSB first
    u64 rev = 0;
    if (a.L == 3) rev = mid;
    else if (a.L == 4) { const u64 k1 = mid >> a.rmid[1], k2 = mid & ((1u << a.rmid[1]) - 1); rev = k1 + (k2 << a.rmid[0]); }
    out_base = k00 + (rev << a.r0);
    out_row_stride = (u64)1 << (a.log_n - a.r);  // last digit is the most significant
    out_col_stride = 1;
  }

  // ---- stage this tile's twiddles w_R^i = w_n^(i * n/R) into LDS ---------------------------
  for (u32 i = tid; i < (R >> 1); i += NTT_THREADS) {
    u32 e = i << (a.log_n - a.r);
    if (a.inverse) e = (u32)((n - e) & n_mask);
    twl[i] = ld_fr(a.tw + e);
  }
  // ---- load (bit-reversed rows), optional pre-multiplication ------------------------------
  const u32 total = R << a.log_c;
  for (u32 e = tid; e < total; e += NTT_THREADS) {
    u32 row, col;
    if (!a.is_last) { row = e >> a.log_c; col = e & (C - 1); }   // consecutive lanes -> consecutive columns
    else { col = e >> a.r; row = e & (R - 1); }                  // consecutive lanes -> consecutive rows
    const u64 g = base + row * in_row_stride + col * in_col_stride;
    fr_t v = ld_fr(a.in + g);
    if (a.pre) { fr_t p = ld_fr(a.pre + g); fe_mul(v, v, p); }
    const u32 rrow = a.r ? (__brev(row) >> (32 - a.r)) : 0;
    tile[(size_t)col * R + rrow] = v;
  }
  // ---- radix-2 DIT stages in LDS ------------------------------------------------------------
  const u32 half = R >> 1;
  for (u32 s = 0; s < a.r; s++) {
    __syncthreads();
    const u32 m = 1u << s;
    for (u32 b = tid; b < (half << a.log_c); b += NTT_THREADS) {
      const u32 col = b >> (a.r - 1), bb = b & (half - 1);
      const u32 j = bb & (m - 1), k = bb >> s;
      const u32 r1 = (k << (s + 1)) | j, r2 = r1 + m;
      fr_t *p1 = tile + (size_t)col * R + r1, *p2 = tile + (size_t)col * R + r2;
      fr_t x = *p1, y = *p2, w = twl[j << (a.r - 1 - s)];
      fe_mul(y, y, w);
      fr_t u, v;
      fe_add(u, x, y);
      fe_sub(v, x, y);
      *p1 = u; *p2 = v;
    }
  }
  __syncthreads();
  // ---- store: inter-pass twiddle (non-last) or final scaling + digit-reversed position ------
  for (u32 e = tid; e < total; e += NTT_THREADS) {
    const u32 row = e >> a.log_c, col = e & (C - 1);   // consecutive lanes -> consecutive columns
    fr_t v = tile[(size_t)col * R + row];
    const u64 g = out_base + row * out_row_stride + col * out_col_stride;
    if (!a.is_last) {
      u32 ex = (u32)((((u64)(jp0 + col) * row) << a.s) & n_mask);
      if (a.inverse) ex = (u32)((n - ex) & n_mask);
      fr_t w = ld_fr(a.tw + ex);
      fe_mul(v, v, w);
    } else if (a.post) {
      fr_t w = ld_fr(a.post + g);
      fe_mul(v, v, w);
    } else if (a.has_post_const) {
      fe_mul(v, v, a.post_const);
    }
    st_fr(a.out + g, v);
  }
}

// out[i] = scale * g^i  (i < n).  Each thread owns 16 consecutive powers: g^(16 t) from the
// table g^(2^k) (k < 32) then 15 multiplications by g.
struct PowTable {
  fr_t p2[32];  // g^(2^k)
  fr_t scale;
};
__global__ void gen_powers_kernel(fr_t *out, u64 n, PowTable tab, int mul_into) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 i0 = t * 16;
  if (i0 >= n) return;
  fr_t acc = tab.scale;
  for (int k = 0; k < 32; k++)
    if ((i0 >> k) & 1) fe_mul(acc, acc, tab.p2[k]);
  for (u64 i = i0; i < n && i < i0 + 16; i++) {
    if (mul_into) { fr_t v = ld_fr(out + i); fe_mul(v, v, acc); st_fr(out + i, v); }
    else st_fr(out + i, acc);
    fe_mul(acc, acc, tab.p2[0]);
  }
}

// ---- element-wise domain ops -----------------------------------------------------------------
__global__ void fr_mul_assign_kernel(fr_t *a, const fr_t *b, u64 n) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
    fr_t x = ld_fr(a + i), y = ld_fr(b + i);
    fe_mul(x, x, y);
    st_fr(a + i, x);
  }
}
__global__ void fr_sub_assign_kernel(fr_t *a, const fr_t *b, u64 n) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
    fr_t x = ld_fr(a + i), y = ld_fr(b + i);
    fe_sub(x, x, y);
    st_fr(a + i, x);
  }
}
__global__ void fr_scale_kernel(fr_t *a, fr_t k, u64 n) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
    fr_t x = ld_fr(a + i);
    fe_mul(x, x, k);
    st_fr(a + i, x);
  }
}
// a = (a*b - c) * zinv : mul_assign + sub_assign + divide_by_z_on_coset in one pass
__global__ void fr_quotient_kernel(fr_t *a, const fr_t *b, const fr_t *c, fr_t zinv, u64 n) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
    fr_t x = ld_fr(a + i), y = ld_fr(b + i), z = ld_fr(c + i);
    fe_mul(x, x, y);
    fe_sub(x, x, z);
    fe_mul(x, x, zinv);
    st_fr(a + i, x);
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static fr_t fr_from_u64_host(u64 v) {
  fr_t c;
  fe_zero(c);
  c.l[0] = (u32)v;
  c.l[1] = (u32)(v >> 32);
  fr_t m;
  fe_to_mont(m, c);
  return m;
}
static fr_t fr_root_of_unity_host() {  // 7^((q-1)/2^32), canonical limbs (ff::PrimeField::ROOT_OF_UNITY)
  static const u32 rou[8] = {0x439f0d2bu, 0x3829971fu, 0x8c2280b9u, 0xb6368350u,
                             0x22c813b4u, 0xd09b6819u, 0xdfe81f20u, 0x16a2a19eu};
  fr_t c, m;
  for (int i = 0; i < 8; i++) c.l[i] = rou[i];
  fe_to_mont(m, c);
  return m;
}
fr_t fr_domain_omega_host(uint32_t log_n) {  // domain.rs:62-66
  fr_t w = fr_root_of_unity_host();
  for (uint32_t i = log_n; i < 32; i++) fe_sqr(w, w);
  return w;
}
static fr_t fr_pow_u64_host(const fr_t &a, u64 e) {
  u32 el[2] = {(u32)e, (u32)(e >> 32)};
  fr_t r;
  fe_pow(r, a, el, 2);
  return r;
}

static int launch_gen_powers(fr_t *out, u64 n, const fr_t &g, const fr_t &scale, int mul_into, hipStream_t st) {
  PowTable tab;
  tab.p2[0] = g;
  for (int k = 1; k < 32; k++) fe_sqr(tab.p2[k], tab.p2[k - 1]);
  tab.scale = scale;
  u64 threads = (n + 15) / 16;
  u32 blocks = (u32)((threads + 255) / 256);
  hipLaunchKernelGGL(gen_powers_kernel, dim3(blocks), dim3(256), 0, st, out, n, tab, mul_into);
  BH_HIP_CHECK(hipGetLastError());
  return BH_OK;
}

static int get_tables(Context &c, uint32_t log_n, bool need_coset, bool need_icoset, hipStream_t st, FftTables *out) {
  std::lock_guard<std::mutex> g(c.fft_mu);
  FftTables &t = c.fft_tables[log_n];
  const u64 n = (u64)1 << log_n;
  fr_t one;
  fe_one(one);
  bool generated = false;
  if (!t.tw) {
    generated = true;
    BH_HIP_CHECK(hipMalloc((void **)&t.tw, n * sizeof(fr_t)));
    int rc = launch_gen_powers(t.tw, n, fr_domain_omega_host(log_n), one, 0, st);
    if (rc) return rc;
    fr_t nn = fr_from_u64_host(n);
    fe_inv(t.minv, nn);
  }
  if (need_coset && !t.coset) {
    generated = true;
    BH_HIP_CHECK(hipMalloc((void **)&t.coset, n * sizeof(fr_t)));
    int rc = launch_gen_powers(t.coset, n, fr_from_u64_host(7), one, 0, st);  // MULTIPLICATIVE_GENERATOR
    if (rc) return rc;
  }
  if (need_icoset && !t.icoset) {
    generated = true;
    BH_HIP_CHECK(hipMalloc((void **)&t.icoset, n * sizeof(fr_t)));
    fr_t ginv;
    fe_inv(ginv, fr_from_u64_host(7));
    int rc = launch_gen_powers(t.icoset, n, ginv, t.minv, 0, st);
    if (rc) return rc;
  }
  // tables are generated on `st`; later users may be on other streams
  if (generated) BH_HIP_CHECK(hipStreamSynchronize(st));
  *out = t;
  return BH_OK;
}

// split log_n into <= 4 passes of <= 8 bits, as evenly as possible (largest first)
static void plan_passes(uint32_t log_n, uint32_t *r, uint32_t *L) {
  if (log_n <= (uint32_t)NTT_LOG_TILE) { *L = 1; r[0] = log_n; return; }
  uint32_t l = (log_n + 7) / 8;
  *L = l;
  uint32_t q = log_n / l, rem = log_n % l;
  for (uint32_t i = 0; i < l; i++) r[i] = q + (i < rem ? 1 : 0);
}

// data: device, 2^log_n Montgomery Fr, in place.  scratch: device, same size (ping-pong for the
// digit-reversing last pass; may be null when log_n <= NTT_LOG_TILE).
int ntt_run(Context &c, fr_t *data, fr_t *scratch, uint32_t log_n, int mode, hipStream_t st) {
  if (log_n >= 32) return BH_ERR_DEGREE_TOO_LARGE;
  const bool inverse = (mode == BH_IFFT || mode == BH_ICOSET_FFT);
  FftTables tab;
  int rc = get_tables(c, log_n, mode == BH_COSET_FFT, mode == BH_ICOSET_FFT, st, &tab);
  if (rc) return rc;
  uint32_t r[4] = {0, 0, 0, 0}, L = 1;
  plan_passes(log_n, r, &L);
  uint32_t s = 0;
  for (uint32_t p = 0; p < L; p++) {
    NttPass a;
    const bool last = (p == L - 1);
    // ping-pong without a copy: pass 0 data -> scratch, middle passes in place in scratch,
    // last pass scratch -> data (digit-reversing scatter)
    a.in = (p == 0) ? data : scratch;
    a.out = (last) ? data : scratch;
    a.tw = tab.tw;
    a.pre = (p == 0 && mode == BH_COSET_FFT) ? tab.coset : nullptr;
    a.post = (last && mode == BH_ICOSET_FFT) ? tab.icoset : nullptr;
    a.post_const = tab.minv;
    a.has_post_const = (last && mode == BH_IFFT) ? 1 : 0;
    a.log_n = log_n;
    a.s = s;
    a.r = r[p];
    a.inverse = inverse ? 1 : 0;
    a.is_last = last ? 1 : 0;
    a.r0 = r[0];
    a.L = L;
    a.rmid[0] = r[1];
    a.rmid[1] = r[2];
    // tile columns: as many as fit 2048 elements, bounded by the extent of the column dimension
    uint32_t log_c = NTT_LOG_TILE - r[p];
    if (L == 1) log_c = 0;
    else if (last) { if (log_c > r[0]) log_c = r[0]; }
    else { uint32_t logM = log_n - s - r[p]; if (log_c > logM) log_c = logM; }
    a.log_c = log_c;
    const u64 tiles = ((u64)1 << log_n) >> (r[p] + log_c);
    const size_t lds = ((size_t)(1u << (r[p] + log_c)) + (size_t)(1u << r[p]) / 2 + 1) * sizeof(fr_t);
    hipLaunchKernelGGL(ntt_pass_kernel, dim3((u32)tiles), dim3(NTT_THREADS), lds, st, a);
    BH_HIP_CHECK(hipGetLastError());
    s += r[p];
  }
  return BH_OK;
}

static u32 ew_blocks(Context &c, u64 n) {
  u64 b = (n + 255) / 256;
  u64 cap = (u64)c.num_cus * 8;
  return (u32)(b < cap ? (b ? b : 1) : cap);
}

int fr_mul_assign(Context &c, fr_t *a, const fr_t *b, u64 n, hipStream_t st) {
  hipLaunchKernelGGL(fr_mul_assign_kernel, dim3(ew_blocks(c, n)), dim3(256), 0, st, a, b, n);
  BH_HIP_CHECK(hipGetLastError());
  return BH_OK;
}
int fr_sub_assign(Context &c, fr_t *a, const fr_t *b, u64 n, hipStream_t st) {
  hipLaunchKernelGGL(fr_sub_assign_kernel, dim3(ew_blocks(c, n)), dim3(256), 0, st, a, b, n);
  BH_HIP_CHECK(hipGetLastError());
  return BH_OK;
}
static fr_t zinv_host(uint32_t log_n) {  // domain.rs:129-140: (7^m - 1)^-1
  fr_t g = fr_from_u64_host(7), z = fr_pow_u64_host(g, (u64)1 << log_n), one, zi;
  fe_one(one);
  fe_sub(z, z, one);
  fe_inv(zi, z);
  return zi;
}
int fr_divide_by_z(Context &c, fr_t *a, uint32_t log_n, hipStream_t st) {
  u64 n = (u64)1 << log_n;
  hipLaunchKernelGGL(fr_scale_kernel, dim3(ew_blocks(c, n)), dim3(256), 0, st, a, zinv_host(log_n), n);
  BH_HIP_CHECK(hipGetLastError());
  return BH_OK;
}
int fr_distribute_powers(Context &c, fr_t *a, u64 n, const fr_t &g, hipStream_t st) {
  fr_t one;
  fe_one(one);
  return launch_gen_powers(a, n, g, one, 1, st);
}
// out[i] = scale * g^i (generator.rs:249-263 powers of tau, with the h-query factor folded in)
int fr_gen_powers(Context &c, fr_t *out, u64 n, const fr_t &g, const fr_t &scale, hipStream_t st) {
  (void)c;
  if (!n) return BH_OK;
  return launch_gen_powers(out, n, g, scale, 0, st);
}
// prover.rs:221-240 on device-resident, already padded a,b,c; result (m entries) in a
int h_poly_dev(Context &c, fr_t *a, fr_t *b, fr_t *cc, fr_t *scratch, uint32_t log_n, hipStream_t st) {
  int rc;
  fr_t *v[3] = {a, b, cc};
  for (int i = 0; i < 3; i++) {
    if ((rc = ntt_run(c, v[i], scratch, log_n, BH_IFFT, st))) return rc;
    if ((rc = ntt_run(c, v[i], scratch, log_n, BH_COSET_FFT, st))) return rc;
  }
  u64 n = (u64)1 << log_n;
  hipLaunchKernelGGL(fr_quotient_kernel, dim3(ew_blocks(c, n)), dim3(256), 0, st, a, b, cc, zinv_host(log_n), n);
  BH_HIP_CHECK(hipGetLastError());
  return ntt_run(c, a, scratch, log_n, BH_ICOSET_FFT, st);
}

}  // namespace bh

// A C++ user of the bellman mirror (bellman_amd/csrc/groth16.hpp): proves knowledge of x with
// x^3 + x + 5 = out - the classic R1CS example - exactly as one would write it against bellman:
//   struct CubicDemo: Circuit { synthesize(cs) { alloc / alloc_input / enforce } }
//   create_proof(circuit, params, r, s)
// CRS points and r, s come from a binary file written by the test (tests/test_gpu_groth16.py);
// the proof (a | b | c affine records, 384 bytes) is written to stdout as hex.
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../../bellman_amd/csrc/groth16.hpp"

using namespace bellman;

struct CubicDemo : Circuit {
  Fr x;
  void synthesize(ConstraintSystem &cs) override {
    const Fr x2v = x * x, x3v = x2v * x, outv = x3v + x + Fr::from_u64(5);
    Variable xv = cs.alloc([&] { return x; });
    Variable x2 = cs.alloc([&] { return x2v; });
    Variable x3 = cs.alloc([&] { return x3v; });
    Variable out = cs.alloc_input([&] { return outv; });
    cs.enforce([&](LinearCombination lc) { return lc + xv; }, [&](LinearCombination lc) { return lc + xv; },
               [&](LinearCombination lc) { return lc + x2; });
    cs.enforce([&](LinearCombination lc) { return lc + x2; }, [&](LinearCombination lc) { return lc + xv; },
               [&](LinearCombination lc) { return lc + x3; });
    // (x3 + x + 5) * 1 = out
    cs.enforce([&](LinearCombination lc) { return lc + x3 + xv + std::make_pair(Fr::from_u64(5), ConstraintSystem::one()); },
               [&](LinearCombination lc) { return lc + ConstraintSystem::one(); },
               [&](LinearCombination lc) { return lc + out; });
  }
};

template <class T> static std::vector<T> read_vec(FILE *f) {
  uint64_t n = 0;
  if (fread(&n, 8, 1, f) != 1) exit(3);
  std::vector<T> v(n);
  if (n && fread(v.data(), sizeof(T), n, f) != n) exit(3);
  return v;
}

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  groth16::VerifyingKey vk;
  if (fread(&vk.alpha_g1, 96, 1, f) != 1 || fread(&vk.beta_g1, 96, 1, f) != 1 || fread(&vk.beta_g2, 192, 1, f) != 1 ||
      fread(&vk.delta_g1, 96, 1, f) != 1 || fread(&vk.delta_g2, 192, 1, f) != 1) return 3;
  auto h = read_vec<groth16::G1Affine>(f), l = read_vec<groth16::G1Affine>(f), a = read_vec<groth16::G1Affine>(f),
       b1 = read_vec<groth16::G1Affine>(f);
  auto b2 = read_vec<groth16::G2Affine>(f);
  Fr x, r, s;
  if (fread(&x, 32, 1, f) != 1 || fread(&r, 32, 1, f) != 1 || fread(&s, 32, 1, f) != 1) return 3;
  fclose(f);

  bh_ctx *ctx = nullptr;
  if (bh_ctx_create(0, &ctx) != BH_OK) { fprintf(stderr, "no gfx950 device (no CPU fallback)\n"); return 4; }
  int rc = 0;
  try {
    groth16::Parameters params(ctx, vk, h.data(), h.size(), l.data(), l.size(), a.data(), a.size(), b1.data(), b1.size(),
                               b2.data(), b2.size());
    CubicDemo circuit;
    circuit.x = x;
    groth16::Proof p = groth16::create_proof(circuit, params, r, s);
    const unsigned char *bytes = reinterpret_cast<const unsigned char *>(&p);
    static_assert(sizeof(groth16::Proof) == 384, "a | b | c");
    for (size_t i = 0; i < sizeof p; i++) printf("%02x", bytes[i]);
    printf("\n");
    // the same proof with the circuit's matrices resident on the device: captured from a witness-free
    // instance of the circuit, then only the value closures run per proof
    CubicDemo shape;
    shape.x = Fr::zero();
    groth16::R1cs r1cs(shape, ctx);
    groth16::Proof p2 = groth16::create_proof(circuit, r1cs, params, r, s);
    if (memcmp(&p, &p2, sizeof p) != 0) { fprintf(stderr, "R1cs path produced a different proof\n"); rc = 5; }
    // create_random_proof (prover.rs:164-180): r, s from the caller's generator; two calls differ
    uint64_t state = 88172645463325252ULL;
    auto rng = [&state] { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return state; };
    groth16::Proof q1 = groth16::create_random_proof(circuit, params, rng), q2 = groth16::create_random_proof(circuit, params, rng);
    if (memcmp(&q1, &q2, sizeof q1) == 0 || memcmp(&q1, &p, sizeof p) == 0) { fprintf(stderr, "create_random_proof is not random\n"); rc = 6; }
    // one caller, proofs back to back (groth16::ProofPipeline): synthesis of proof k+1 beside the device part of proof k;
    // every proof must equal create_proof's for the same x, r, s
    {
      groth16::ProofPipeline pipe(params, &r1cs, 2);
      groth16::ProofPipeline pipe_host(params, nullptr, 2);
      std::vector<groth16::Proof> want;
      for (int i = 0; i < 5; i++) {
        CubicDemo c;
        c.x = x + Fr::from_u64((uint64_t)i);
        want.push_back(groth16::create_proof(c, params, r + Fr::from_u64((uint64_t)i), s));
        pipe.submit(c, r + Fr::from_u64((uint64_t)i), s);
        pipe_host.submit(c, r + Fr::from_u64((uint64_t)i), s);
      }
      for (int i = 0; i < 5; i++) {
        groth16::Proof a1 = pipe.next(), a2 = pipe_host.next();
        if (memcmp(&a1, &want[i], sizeof a1) != 0 || memcmp(&a2, &want[i], sizeof a2) != 0) {
          fprintf(stderr, "pipelined proof %d differs\n", i);
          rc = 7;
        }
      }
      if (pipe.pending() || pipe_host.pending()) rc = 8;
    }
  } catch (const SynthesisError &e) {
    fprintf(stderr, "SynthesisError %d: %s\n", e.code, e.what());
    rc = 10 + e.code;
  }
  bh_ctx_destroy(ctx);
  return rc;
}

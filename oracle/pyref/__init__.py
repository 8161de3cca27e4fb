"""Pure-Python big-int oracle for bellman's Groth16 proving hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product (`bellman_amd/`) may import
this package; only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may.  It restates, function by function, the algorithms of

  /root/reference/src/multiexp.rs          -> oracle.pyref.multiexp
  /root/reference/src/domain.rs            -> oracle.pyref.domain
  /root/reference/src/multicore.rs         -> oracle.pyref.multicore
  /root/reference/groth16/src/prover.rs    -> oracle.pyref.prover
  /root/reference/groth16/src/generator.rs -> oracle.pyref.generator (fixture producer)

generically over an "engine" so that the SAME code runs over the reference's
toy `DummyEngine` (F_64513, groth16/src/tests/dummy_engine.rs) and BLS12-381.

Parity pinning: the toy-engine path is pinned by the reference's only
known-answer test, `test_xordemo` (groth16/src/tests/mod.rs:91-373).  For
BLS12-381 the reference holds NO golden vectors (every BLS test draws from
thread_rng and checks a property) and the arithmetic crate `bls12_381 0.8.0`
(Cargo.lock:105-108) is absent, so BLS parity is pinned only through
(a) the public curve parameters + the Zcash compressed-generator KAT and
(b) the algebraic properties the reference's own tests check.
"""

"""Cached answers of the CPU oracle for the largest seeded GPU parity cases (the 2^26-term multiexp and the
2^22 / 2^24 proofs cost about a minute of oracle time each on the driver's box, every run, for inputs that never change).

`oracle_answer(key, inputs, compute)` returns the oracle's result for a case:
  * default: the record stored in tests/golden/scale_oracle.json under `key`, provided the digest of the case's INPUTS
    (a strided sample of every input array, sha256) equals the stored one - otherwise, and for unknown keys, `compute()`
    runs the oracle as before;
  * BELLMAN_GOLDEN_REGEN=1: always runs the oracle, compares with a stored record if there is one (a mismatch fails the
    test: the oracle or the inputs changed) and writes the records of the run to gpurun_out/golden/scale_oracle.json,
    from where tools/r6/gpu_golden.sh copies them into tests/golden/ (inputs are generated on the device, so the file is
    produced on the GPU box:  BELLMAN_GOLDEN_REGEN=1 python -m pytest tests/test_gpu_scale.py tests/test_gpu_boolean.py -m gpu).
What is stored is DATA: key, input digest, the oracle's output words - never code."""

import hashlib
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "tests", "golden", "scale_oracle.json")
REGEN_DIR = os.path.join(ROOT, "gpurun_out", "golden")
REGEN = os.environ.get("BELLMAN_GOLDEN_REGEN") == "1"
_SAMPLE_BYTES = 32 << 20


def _digest(arrays):
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str((a.dtype.str, a.shape)).encode())
        flat = a.reshape(-1).view(np.uint8)
        if flat.size <= _SAMPLE_BYTES:
            h.update(flat.tobytes())
        else:   # rows at a fixed stride + the last rows: a changed generator or seed moves every row
            rows = a.reshape(a.shape[0], -1)
            stride = max(1, (rows.shape[0] * rows.shape[1] * rows.itemsize) // _SAMPLE_BYTES)
            h.update(np.ascontiguousarray(rows[::stride]).tobytes())
            h.update(np.ascontiguousarray(rows[-64:]).tobytes())
    return h.hexdigest()


def _load(path):
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return {}


def oracle_answer(key, inputs, compute):
    """-> (list of numpy uint64 arrays, "cached" | "oracle").  `inputs`: the arrays the oracle would be given;
    `compute()` -> list of numpy uint64 arrays."""
    digest = _digest(inputs)
    stored = _load(PATH).get(key)
    if stored is not None and stored.get("inputs_sha256") == digest and not REGEN:
        return [np.array([int(w, 16) for w in rec], dtype=np.uint64) for rec in stored["outputs"]], "cached"
    out = [np.ascontiguousarray(o, dtype=np.uint64).reshape(-1) for o in compute()]
    if REGEN:
        rec = {"inputs_sha256": digest, "outputs": [["%016x" % int(w) for w in o] for o in out]}
        if stored is not None and stored.get("inputs_sha256") == digest:
            assert stored["outputs"] == rec["outputs"], "oracle answer for %s differs from the stored one" % key
        os.makedirs(REGEN_DIR, exist_ok=True)
        p = os.path.join(REGEN_DIR, "scale_oracle.json")
        cur = _load(p)
        cur[key] = rec
        with open(p, "w") as f:
            json.dump(cur, f, indent=0, sort_keys=True)
    return out, "oracle"

"""Index arithmetic of the bucket reduction (bellman_amd/csrc/msm_ec.cuh, section 5), restated over plain integers.

The device computes  sum_d (d + 1) * B[d]  over 2^(c-1) buckets without a serial running sum: d = hi * 2^l + lo, row sums over
lo, column sums over hi, then per-bit sums of those vectors and a plain total; the host finishes with
sum_p 2^p U[p] + T (msm_host_tail).  Round 6 changed two pieces of index arithmetic this file pins with integers in place of curve
points (a point is an element of an abelian group; Z is one):
  * SumDesc::splits - the two-stage row / column sums: every output is cut into `splits` pieces of count / splits consecutive
    elements, one worker adds up a piece (out[group * splits + piece]), a second SUM_STRIDED job folds the pieces;
  * nth_with_bit    - SUM_BITS walks the j-th index with bit k set, j < count / 2, instead of striding over all indices.
Everything is mirrored statement by statement from the kernels (k2_partial_sum / msm_sum_kernel), so a change there shows here."""
import random

import pytest


def nth_with_bit(j, k):   # msm_ec.cuh nth_with_bit
    return ((j >> k) << (k + 1)) | (1 << k) | (j & ((1 << k) - 1))


def partial_sum(d, data, g, sub, G):
    """what worker `sub` of the G workers of output g adds up (k2_partial_sum)"""
    gg, ln = g // d["splits"], d["count"] // d["splits"]
    k0 = (g % d["splits"]) * ln
    outer, in_idx = gg // d["inner"], gg % d["inner"]
    base = outer << d["group_shift"]
    acc = 0
    if d["mode"] == "strided":
        for k in range(k0 + sub, k0 + ln, G):
            acc += data[base + in_idx * d["istride"] + k * d["stride"]]
    else:   # bits: in_idx = bit position
        for j in range(sub, d["count"] >> 1, G):
            acc += data[base + nth_with_bit(j, in_idx)]
    return acc


def run_job(d, data, G):
    return [sum(partial_sum(d, data, g, sub, G) for sub in range(G)) for g in range(d["groups"])]


@pytest.mark.parametrize("k", range(0, 11))
def test_nth_with_bit_enumerates_exactly_the_selected_half(k):
    count = 1 << 11
    got = [nth_with_bit(j, k) for j in range(count >> 1)]
    assert got == [i for i in range(count) if (i >> k) & 1]


@pytest.mark.parametrize("c,W,two_len", [(16, 3, 16), (20, 1, 16), (13, 2, 4), (9, 1, 64), (20, 1, 8)])
def test_two_stage_sums_and_bit_sums_give_the_weighted_bucket_sum(c, W, two_len):
    rnd = random.Random(c * 100 + W)
    cb = c - 1
    lo_bits, hi_bits = cb // 2, cb - cb // 2
    Lw, H = 1 << lo_bits, 1 << hi_bits
    NB = W << cb
    pts = [rnd.randrange(-5, 6) if rnd.random() < 0.7 else 0 for _ in range(NB)]
    want = [sum((d + 1) * pts[(w << cb) + d] for d in range(1 << cb)) for w in range(W)]
    # rows (sum over lo, contiguous) and columns (sum over hi, stride Lw): msm_enqueue dr / dc
    dr = dict(mode="strided", groups=W * H, count=Lw, inner=H, stride=1, istride=Lw, group_shift=cb, splits=1)
    dc = dict(dr, groups=W * Lw, count=H, inner=Lw, stride=Lw, istride=1)
    rows1, cols1 = run_job(dr, pts, 8), run_job(dc, pts, 4)
    # the same in two stages: pieces of two_len elements, one worker each, then a fold of the pieces
    len_r, len_c = min(two_len, Lw), min(two_len, H)
    Sr, Sc = Lw // len_r, H // len_c
    r1 = dict(dr, splits=Sr, groups=dr["groups"] * Sr)
    c1 = dict(dc, splits=Sc, groups=dc["groups"] * Sc)
    part_r, part_c = run_job(r1, pts, 1), run_job(c1, pts, 1)
    r2 = dict(mode="strided", groups=dr["groups"], count=Sr, inner=dr["groups"], stride=1, istride=Sr, group_shift=0, splits=1)
    c2 = dict(r2, groups=dc["groups"], count=Sc, inner=dc["groups"], istride=Sc)
    rows, cols = run_job(r2, part_r, max(1, Sr // 2)), run_job(c2, part_c, max(1, Sc // 2))
    assert rows == rows1 and cols == cols1
    # bit sums: U[w][p], p < lo_bits from the column sums, p >= lo_bits from the row sums; T[w] = plain total
    bl = dict(mode="bits", groups=W * lo_bits, count=Lw, inner=max(1, lo_bits), stride=1, istride=0, group_shift=lo_bits, splits=1)
    bh = dict(bl, groups=W * hi_bits, count=H, inner=max(1, hi_bits), group_shift=hi_bits)
    u_lo, u_hi = run_job(bl, cols, 32), run_job(bh, rows, 32)
    for w in range(W):
        total = sum(cols[w * Lw:(w + 1) * Lw])
        assert total == sum(rows[w * H:(w + 1) * H])
        acc = total                                              # the "+1" of the bucket weights
        for p in range(lo_bits):
            acc += (1 << p) * u_lo[w * lo_bits + p]
        for p in range(hi_bits):
            acc += (1 << (lo_bits + p)) * u_hi[w * hi_bits + p]
        assert acc == want[w]

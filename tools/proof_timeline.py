"""Kernel timeline of the LAST create_proof in a rocprofv3 --kernel-trace csv (tools/gpu_r4_final.sh keeps the tail of the
trace of `profile_suite.py proof 20 3 1` as proof_trace.csv): span, union-busy time, per-kernel totals, start/end of every
kernel of 0.15 ms or more.   python tools/proof_timeline.py <kernel_trace.csv> [title]"""
import csv
import re
import sys


def short(name):
    name = name.replace("void bh::", "").replace("bh::", "")
    name = re.sub(r"\(.*", "", name)
    return name[:64]


def main():
    rows = [r for r in csv.DictReader(open(sys.argv[1])) if r.get("Kind") == "KERNEL_DISPATCH"]
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]) for r in rows)
    # proofs are separated by the host's witness generation (tens of ms without a kernel): take what follows the last gap > 10 ms
    cut = 0
    for i in range(1, len(ev)):
        if ev[i][0] - max(e[1] for e in ev[:i]) > 10_000_000:
            cut = i
    ev = ev[cut:]
    t0 = ev[0][0]
    span = (max(e[1] for e in ev) - t0) / 1e6
    busy, cur_s, cur_e = 0, None, None
    for s, e, _, _ in ev:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(sys.argv[2] if len(sys.argv) > 2 else "kernel trace of ONE 2^20-constraint proof (R1CS resident)")
    print("span ms %.3f kernels %d" % (span, len(ev)))
    print("union busy %.3f ms" % (busy / 1e6))
    tot = {}
    for s, e, n, _ in ev:
        c, d = tot.get(n, (0, 0))
        tot[n] = (c + 1, d + e - s)
    for n, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:14]:
        print("  %-64s %3d %8.3f ms" % (n, c, d / 1e6))
    for s, e, n, q in ev:
        if e - s >= 150_000:
            print("  %7.3f  %7.3f q=%s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, q, n))


main()

#!/usr/bin/env python3
"""Attributes the samples of tools/host_profile.cpp to functions:  python tools/host_profile.py <prefix> [top N] [--hot PCT]
(--hot: also list the instructions that hold at least PCT % of the samples, with the five instructions before them)."""
import bisect, collections, re, subprocess, sys

prefix = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else 15
hot = float(sys.argv[sys.argv.index("--hot") + 1]) if "--hot" in sys.argv else None
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
mods = []   # (lo, hi, file offset, path)
for line in open(prefix + ".maps"):
    f = line.split()
    if len(f) >= 6 and "x" in f[1]:
        lo, hi = (int(x, 16) for x in f[0].split("-"))
        mods.append((lo, hi, int(f[2], 16), f[5]))
symtab = {}
def symbols(path):
    if path not in symtab:
        out = subprocess.run(["nm", "-C", "-n", "--defined-only", path], capture_output=True, text=True).stdout
        syms = sorted((int(p[0], 16), p[2]) for p in (l.split(" ", 2) for l in out.splitlines()) if len(p) == 3 and p[1] in "tTwW")
        # the first executable LOAD segment's (vaddr - offset): shared objects here are linked with vaddr == offset + const
        ph = subprocess.run(["readelf", "-lW", path], capture_output=True, text=True).stdout
        delta = 0
        for l in ph.splitlines():
            m = re.match(r"\s*LOAD\s+0x([0-9a-f]+)\s+0x([0-9a-f]+)\s+0x[0-9a-f]+\s+0x[0-9a-f]+\s+0x[0-9a-f]+\s+(.*?)\s+0x[0-9a-f]+\s*$", l)
            if m and "E" in m.group(3):
                delta = int(m.group(2), 16) - int(m.group(1), 16)
                break
        symtab[path] = ([a for a, _ in syms], syms, delta)
    return symtab[path]
cnt = collections.Counter()
per_mod_pc = collections.defaultdict(collections.Counter)
pcs = [int(l, 16) for l in open(prefix + ".samples")]
for pc in pcs:
    for lo, hi, off, path in mods:
        if lo <= pc < hi:
            addrs, syms, delta = symbols(path) if path.startswith("/") else ([], [], 0)
            va = pc - lo + off + delta
            i = bisect.bisect_right(addrs, va) - 1
            name = syms[i][1] if i >= 0 else "?"
            cnt[(path.rsplit("/", 1)[-1], name)] += 1
            per_mod_pc[path][va] += 1
            break
    else:
        cnt[("?", hex(pc))] += 1
tot = len(pcs)
for (mod, name), c in cnt.most_common(top):
    print(f"{100.0 * c / tot:5.1f}%  {mod:26s} {name[:140]}")
if hot is not None:
    for path, pcc in per_mod_pc.items():
        if not path.startswith("/") or "bellman" not in path: continue
        out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", "-C", path], capture_output=True, text=True).stdout
        lines = out.splitlines()
        func = ""
        for k, line in enumerate(lines):
            m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
            if m: func = m.group(1); continue
            m = re.match(r"\s+([0-9a-f]+):\s+(.*)", line)
            if m and 100.0 * pcc.get(int(m.group(1), 16), 0) / tot >= hot:
                print(f"--- {100.0 * pcc[int(m.group(1), 16)] / tot:.1f}% in {func[:120]}")
                for l in lines[max(0, k - 5):k + 1]: print("     " + l.strip()[:120])

"""Compile one .hip file of bellman_amd/csrc for gfx950 and print the register / scratch / LDS / occupancy
figures clang reports for every kernel (-Rpass-analysis=kernel-resource-usage).  Works without a GPU.
Usage: python tools/kernel_resources.py msm_g2.hip [extra hipcc flags]"""
import os, re, subprocess, sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bellman_amd", "csrc")


def main():
    src = sys.argv[1]
    os.makedirs("/tmp/kr", exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/tmp/kr/out.o",
           "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
    out = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
    cur = None
    rows = []
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = {"name": re.sub(r"\(.*", "", name).replace("void ", "").replace("bh::", "")[:70]}
            rows.append(cur)
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    if not rows:
        print(out[-3000:])
    print("%-70s %5s %5s %7s %6s %6s %4s" % ("kernel", "VGPR", "AGPR", "scratch", "vspill", "LDS", "occ"))
    for r in rows:
        print("%-70s %5s %5s %7s %6s %6s %4s" % (r["name"], r.get("vgpr"), r.get("agpr"), r.get("scratch"), r.get("vspill"), r.get("lds"), r.get("occ")))


main()

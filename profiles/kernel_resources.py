#!/usr/bin/env python3
"""Per-kernel register / LDS / spill table of every shipped HIP kernel (hipcc -Rpass-analysis=kernel-resource-usage on each
csrc/*.hip, gfx950).  Usage: python profiles/kernel_resources.py [> profiles/rNN_kernel_resources.txt]
Exit code 1 if any kernel spills (scratch > 0)."""
import glob
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "videoloop3d_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
         "-Rpass-analysis=kernel-resource-usage"]


def one(src):
    out = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", src, "-o", "/dev/null"], capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    return os.path.basename(src), rows


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.splitlines()


def main():
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    with ThreadPoolExecutor(8) as ex:
        res = list(ex.map(one, srcs))
    nk = nspill = 0
    for tu, rows in res:
        names = demangle([r["name"] for r in rows])
        print(f"== {tu}: {len(rows)} kernels")
        for r, n in zip(rows, names):
            n = re.sub(r"\(anonymous namespace\)::", "", n)
            n = re.sub(r"\(.*$", "", n)
            scratch = int(r.get("ScratchSize [bytes/lane]", 0))
            nk += 1
            nspill += scratch > 0
            print(f"  {n[:88]:88s} vgpr {r.get('VGPRs'):>3s} agpr {r.get('AGPRs'):>3s} sgpr {r.get('TotalSGPRs'):>3s} "
                  f"scratch {scratch:>4d} lds {r.get('LDS Size [bytes/block]'):>6s} occ {r.get('Occupancy [waves/SIMD]')}")
    print(f"total {nk} kernels, {nspill} with scratch")
    return 1 if nspill else 0


if __name__ == "__main__":
    sys.exit(main())

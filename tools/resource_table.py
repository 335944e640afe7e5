#!/usr/bin/env python3
"""Condense `make -C jet-pbrt_amd/csrc asm`'s resource_usage.txt (hipcc -Rpass-analysis=kernel-resource-usage) into one
line per kernel of this library: VGPRs, AGPRs, SGPRs, scratch, waves/SIMD, static LDS.  Library kernels (rocprim) are
dropped.  CPU only:  python tools/resource_table.py [resource_usage.txt] > profiles/rNN_resource_usage.txt"""
import re
import subprocess
import sys
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    try:
        out = subprocess.run(["c++filt"] + names, stdout=subprocess.PIPE, text=True).stdout.split("\n")
        return [o if o else n for o, n in zip(out, names)]
    except Exception:
        return names


def short(n):
    n = re.sub(r"\(jp::SceneView.*|\((?:unsigned|int|float|HIP|Queues|JpBsdf|Wide).*", "", n)   # drop the argument list, keep template arguments
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "jet-pbrt_amd", "csrc", "resource_usage.txt")
    rows, cur = [], None
    for line in open(path, errors="replace"):
        m = re.search(r"^(\S+?:\d+):\d+: remark: (?:Function )?Name: (\S+)", line)
        if m:
            cur = {"src": m.group(1), "name": m.group(2)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[^\]]+\])?): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    rows = [r for r in rows if "rocprim" not in r["src"] and "hipcub" not in r["src"]]
    names = demangle([r["name"] for r in rows])
    print("%-44s %5s %5s %5s %8s %10s %8s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "waves/SIMD", "LDS(B)"))
    for r, n in zip(rows, names):
        print("%-44s %5d %5d %5d %8d %10d %8d" % (short(n)[:44], r.get("VGPRs", -1), r.get("AGPRs", 0), r.get("TotalSGPRs", -1),
                                                  r.get("ScratchSize [bytes/lane]", 0), r.get("Occupancy [waves/SIMD]", -1), r.get("LDS Size [bytes/block]", 0)))


if __name__ == "__main__":
    main()

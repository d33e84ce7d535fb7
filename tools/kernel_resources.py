#!/usr/bin/env python3
"""Registers, LDS, scratch and spills of every kernel in the shipped libscanfuse.so, read from the code objects' metadata notes (no GPU needed):
what the compiler gave each kernel, and the occupancy that follows (gfx950: 512 VGPRs per SIMD lane in blocks of 8, at most 8 waves per SIMD; 160 KB LDS
per CU).  rocPRIM's kernels (the radix sorts of the weld and the cleaning filters) are counted, not listed.

    python tools/kernel_resources.py [path to libscanfuse.so]  >  profiles/rNN_kernel_resources.txt
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(lib):
    d = tempfile.mkdtemp(prefix="sf_res_")
    try:
        c = os.path.join(d, "lib.so")
        shutil.copy(lib, c)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", c], capture_output=True, text=True, cwd=d)   # writes the bundles next to its input
        rows = []
        for f in sorted(glob.glob(os.path.join(d, "*gfx950*"))):
            t = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f], capture_output=True, text=True).stdout
            for blk in re.split(r"\n\s*- \.agpr_count:", t)[1:]:
                blk = ".agpr_count:" + blk

                def g(k, blk=blk):
                    m = re.search(r"\.%s:\s*(\S+)" % k, blk)
                    return m.group(1) if m else "0"
                rows.append({"name": g("name"), "vgpr": int(g("vgpr_count")), "agpr": int(g("agpr_count")), "sgpr": int(g("sgpr_count")), "scratch": int(g("private_segment_fixed_size")),
                             "lds": int(g("group_segment_fixed_size")), "wg": int(g("max_flat_workgroup_size")), "vspill": int(g("vgpr_spill_count")), "sspill": int(g("sgpr_spill_count"))})
        return rows
    finally:
        shutil.rmtree(d, ignore_errors=True)


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"\(anonymous namespace\)::", "", o) for o in out]


def short(name):
    name = re.sub(r"^void ", "", name)
    depth, cut = 0, len(name)
    for i, ch in enumerate(name):   # cut the argument list, keep the template arguments
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    return name[:cut]


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "scannet_amd", "libscanfuse.so")
    rows = kernels(lib)
    for r, n in zip(rows, demangle([r["name"] for r in rows])):
        r["dem"] = n
    own = [r for r in rows if "rocprim" not in r["dem"] and "hipcub" not in r["dem"]]
    print("# %s: %d kernels in the code objects, %d of this repository (the rest: rocPRIM instantiations)" % (os.path.basename(lib), len(rows), len(own)))
    print("# waves/SIMD = min(8, 512 // ceil8(vgpr + agpr)); workgroups/CU by LDS = 163840 // lds; scratch = bytes of private memory per lane (0 = everything in registers)")
    print("%-78s %5s %5s %7s %8s %6s %7s %7s %10s" % ("kernel", "vgpr", "sgpr", "lds B", "scratch", "max wg", "v-spill", "s-spill", "waves/SIMD"))
    for r in sorted(own, key=lambda r: short(r["dem"])):
        v = r["vgpr"] + r["agpr"]
        waves = min(8, 512 // max(8, -(-v // 8) * 8))
        by_lds = (163840 // r["lds"]) * max(1, r["wg"] // 64) / 4.0 if r["lds"] else 8
        print("%-78s %5d %5d %7d %8d %6d %7d %7d %6d%s" % (short(r["dem"])[:78], r["vgpr"], r["sgpr"], r["lds"], r["scratch"], r["wg"], r["vspill"], r["sspill"], waves,
                                                         "  (LDS: %.1f)" % by_lds if by_lds < waves else ""))
    sc = [r for r in own if r["scratch"] or r["vspill"]]
    print("# kernels of this repository with private memory: %s" % (", ".join("%s %d B" % (short(r["dem"]), r["scratch"]) for r in sc) or "none"))
    return 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""What does the vector ALU of a gfx950 SIMD have to do for one frame of the batched integrate kernel?

Disassembles the device code of scannet_amd/libscanfuse.so, takes the per-frame loop body of k_integrate<1, COLOR, true, 2, false> (the kernels bench.py
times), and prices every VALU instruction with the issue cost tools/gpu/valu_peak.hip measured on the MI355X (profiles/r05_valu_issue_table.txt):

  fast   2.25 cycles per wave-instruction and SIMD: v_fma / v_fmac / v_mul / v_add / v_sub_f32, v_mov_b32, v_add / v_sub_u32, v_and / v_or / v_xor_b32,
         v_lshrrev_b32, v_ashrrev_i32 -- when every source is a vector register, an inline constant or a literal
  slow   4.25 cycles: the same instructions with a scalar-register source; every v_pk_*_f32; conversions, v_floor / v_trunc, v_min / v_max / v_med3,
         compares, v_cndmask, v_lshlrev_b32, SDWA / DPP forms, 24-bit and 32-bit multiplies, every three-operand integer instruction
         (v_add_lshl / v_lshl_add / v_add3 / v_and_or / v_bfe / v_bfi / v_alignbit / v_perm / v_lerp_u8 / v_lshl_add_u64), v_readlane-class moves
  trans  8.3 cycles: v_rcp / v_rsq / v_sqrt / v_exp / v_log_f32

and prints (JSON) the loop's instruction count, its mean cost per instruction with every instruction priced as if nothing overlapped (`cycles_serial`),
and the floor if the fp32 fast class ran entirely beside the slow classes of other waves (`cycles_overlapped`: max(all x 2.25, slow pipe)) -- the two
ends between which the hardware's schedule lies.  bench.py multiplies SQ_INSTS_VALU by `cycles_serial` per instruction for roofline.frac.

  python tools/valu_cost_model.py [--so scannet_amd/libscanfuse.so] [--kernel 'k_integrateILi1ELi2ELb1ELi2ELb0']
"""
import argparse
import glob
import json
import os
import re
import shutil
import subprocess
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
FAST = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b64"}
TRANS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_rcp_iflag_f32", "v_sin_f32", "v_cos_f32"}
C_FAST, C_SLOW, C_TRANS = 2.25, 4.25, 8.3


def disassemble(so):
    d = tempfile.mkdtemp(prefix="sf_dis_")
    try:
        c = os.path.join(d, os.path.basename(so))
        shutil.copy(so, c)
        subprocess.run([OBJDUMP, "--offloading", c], capture_output=True, text=True, cwd=d)
        out = []
        for b in sorted(glob.glob(c + ".*gfx950*")):
            out.append(subprocess.run([OBJDUMP, "-d", b], capture_output=True, text=True).stdout)
        return "\n".join(out)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def kernel_body(text, pat):
    m = re.search(r"^[0-9a-f]+ <([^>]*%s[^>]*)>:\n(.*?)s_endpgm" % re.escape(pat), text, re.S | re.M)
    if not m:
        raise SystemExit("kernel %s not found" % pat)
    lines = []
    for ln in m.group(2).splitlines():
        ln = ln.split("//")[0].strip()
        if ln and not ln.startswith("<"):
            lines.append(ln)
    return m.group(1), lines


def frame_loop(lines):
    """The frame loop = the innermost backward branch that encloses the gathers (buffer_load) -- found by the label comments objdump leaves out, so: the
    span between the first s_ff1 / s_bcnt-style frame pick (s_ff1_i32_b32 = __builtin_ctz of the frame mask) and the last branch back to it."""
    first = next((i for i, l in enumerate(lines) if l.startswith("s_ff1_i32_b32")), None)
    if first is None:
        return lines
    last = max(i for i, l in enumerate(lines) if l.startswith(("s_cbranch", "s_branch")))
    tile_store = next((i for i, l in enumerate(lines) if i > first and l.startswith("global_store_dwordx4")), last)
    return lines[first:min(last, tile_store)]


def price(lines):
    n = {"fast": 0, "slow": 0, "trans": 0}
    by = {}
    for l in lines:
        op = l.split()[0]
        if not op.startswith("v_"):
            continue
        base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
        operands = l[len(op):]
        srcs = operands.split(",")[1:] if "," in operands else []
        scalar_src = any(re.match(r"\s*(s\d+|s\[|vcc|exec|ttmp|m0)", s) for s in srcs)
        if base in TRANS:
            k = "trans"
        elif base in FAST and not scalar_src and not op.endswith(("_sdwa", "_dpp")):
            k = "fast"
        else:
            k = "slow"
        n[k] += 1
        by[op] = by.get(op, 0) + 1
    total = sum(n.values())
    serial = n["fast"] * C_FAST + n["slow"] * C_SLOW + n["trans"] * C_TRANS
    # the fp32 fast class issues beside the slow classes of other waves (valu_peak: alternations and clumps run at the fast rate), the integer fast class does not
    fp_fast = sum(v for k, v in by.items() if re.sub(r"_(e32|e64)$", "", k) in ("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32"))
    fp_fast = min(fp_fast, n["fast"])
    other_pipe = (n["fast"] - fp_fast) * C_FAST + n["slow"] * C_SLOW + n["trans"] * C_TRANS
    overlapped = max(total * C_FAST, other_pipe)
    return {"valu_instructions": total, "by_class": n, "cycles_serial": round(serial, 1), "cycles_serial_per_instruction": round(serial / max(total, 1), 3),
            "cycles_overlapped_floor": round(overlapped, 1), "cycles_overlapped_per_instruction": round(overlapped / max(total, 1), 3),
            "top_opcodes": dict(sorted(by.items(), key=lambda kv: -kv[1])[:24])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--so", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scannet_amd", "libscanfuse.so"))
    ap.add_argument("--kernel", action="append", default=[])
    a = ap.parse_args()
    text = disassemble(a.so)
    out = {"costs_cycles_per_wave_instruction_and_simd": {"fast": C_FAST, "slow": C_SLOW, "trans": C_TRANS}, "source": "tools/gpu/valu_peak.hip on MI355X: profiles/r05_valu_issue_table.txt",
           "kernels": {}}
    for pat in (a.kernel or ["k_integrateILi1ELi2ELb1ELi2ELb0ELi4E", "k_integrateILi1ELi0ELb1ELi2ELb0ELi4E"]):
        name, lines = kernel_body(text, pat)
        out["kernels"][pat] = {"whole_kernel": price(lines), "frame_loop": price(frame_loop(lines))}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

"""tools/check_division.c: the CPU-side brute-force check of the hand-expanded, correctly rounded divisions the kernels use (k_integrate: 1/z by
two Newton steps, n/m through a table reciprocal and one correction; k_alloc: a/b through a reciprocal and two residual corrections).  The
device-side counterpart on the real v_rcp_f32 is sf_selftest_division (tests/test_gpu_tsdf.py)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read().replace("\n", " ")
    except OSError:
        return False


@pytest.mark.skipif(shutil.which("gcc") is None or not _has_fma(), reason="needs gcc and a CPU with fused multiply-add")
def test_cpu_brute_force_of_the_expanded_divisions(tmp_path):
    exe = str(tmp_path / "check_division")
    subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tools", "check_division.c"), "-lm"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 3 and all(" 0 mismatches" in ln or "0 differences; exact seed -> wrong result 0 times" in ln for ln in lines), r.stdout

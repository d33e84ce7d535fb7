"""bench.py on a machine without a GPU: it must refuse loudly (there is no CPU fallback to measure), and its help must name the four
configurations of BASELINE.json it can run."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="this check is for machines without a GPU")
def test_bench_refuses_without_a_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1"], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]      # and prints no result line


def test_bench_help_names_the_configs():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0
    for word in ("--config", "4mm", "1mm", "scans", "partition", "--gpus", "--steps", "--warmup", "--host-stage", "--repeats", "--scene", "--noise",
                 "--exchange", "--share-gpu", "--no-e2e", "--cpu-frames", "--depth-only", "--no-e2e-rgbd", "--no-prefix-check"):
        assert word in r.stdout, word


def test_repeats_rule_is_a_function_of_the_arguments_alone():
    """Every rank of a multi-GPU run must take the same number of repeats without talking to the others: the rule only reads K and the config."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.repeats_for(20) == 1667 and b.repeats_for(5514) == 7 and b.repeats_for(10 ** 7) == 3 and b.repeats_for(1) == 2000
    assert b.repeats_for(192, "1mm") == 6 and b.repeats_for(20, "1mm") == 50

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


def test_valu_cost_model_prices_the_shipped_kernels():
    """roofline.frac of the batched integrate kernel = SQ_INSTS_VALU x the mean issue cost of the kernel's frame loop / SIMD cycles; the cost comes from
    tools/valu_cost_model.py: the shipped library disassembled, the frame loop of the timed kernel priced by instruction class (issue costs measured on
    the MI355X: profiles/r05_valu_issue_table.txt).  Here, without a GPU: the disassembler is there, both timed kernels are found by name, the loop has
    the size of a frame's work, and the price lies between the all-simple and the all-slow class."""
    import importlib.util
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no llvm-objdump")
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    for depth_only in (False, True):
        m = b.valu_cost_model(depth_only)
        assert m is not None, "the kernel bench.py times was not found in libscanfuse.so (renamed template arguments?)"
        n = m["by_class"]
        assert 200 < m["valu_instructions"] < 500 and n["fast"] + n["slow"] + n["trans"] == m["valu_instructions"] and n["trans"] == 8   # 8 reciprocals: one per voxel of the lane
        assert 2.25 < m["cycles_overlapped_per_instruction"] <= m["cycles_serial_per_instruction"] < 4.4
        assert m["costs_cycles"] == {"fast": 2.25, "slow": 4.25, "trans": 8.3}
    assert b.valu_cost_model(False)["valu_instructions"] > b.valu_cost_model(True)["valu_instructions"]      # colour costs instructions


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


def _blob(path):
    import json
    raw = open(path).read()
    try:
        return json.loads(raw)
    except ValueError:
        return json.loads([ln for ln in raw.splitlines() if ln.startswith('{"')][-1])


def test_the_line_bench_prints_is_compact_and_parseable():
    """Round 5's record was one 24 KB JSON line; the driver keeps an 8 KB tail of stdout and could not parse it (BENCH_r05.json: parsed null).  The line is
    now built by compact_line(): the contract's keys, roofline and cpu_baseline as numbers, < 4 KB; the full measurement goes to bench_detail.json.
    Checked here on the full blobs of earlier rounds (every configuration bench.py runs)."""
    import glob
    import json
    b = _load_bench()
    blobs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[2-5]_bench_*.json")))
    assert len(blobs) >= 10
    for path in blobs:
        out = _blob(path)
        if "metric" not in out:
            continue
        line = b.compact_line(out, "bench_detail.json")
        assert len(line) < 4096 and "\n" not in line, (path, len(line))
        c = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
            assert k in c, (path, k)
        assert c["value"] == out["value"] and c["ms_per_step"] == out["ms_per_step"] and isinstance(c["config"]["workload"], str) and c["config"]["workload"]
        if isinstance(out.get("roofline"), dict):
            for k in ("bound", "frac", "achieved", "peak", "unit", "traffic"):
                assert k in c["roofline"], (path, k)
        if isinstance(out.get("cpu_baseline"), dict):
            for k in ("value", "unit", "cores", "kind", "sample"):
                assert k in c["cpu_baseline"], (path, k)


def test_compact_line_sheds_optional_groups_before_it_exceeds_the_limit():
    import json
    b = _load_bench()
    out = _blob(os.path.join(ROOT, "profiles", "r05_bench_4mm_driver_args.json"))
    out["config"]["workload"] = "w" * 5000
    out["cpu_baseline"]["sample"] = "s" * 5000
    out["end_to_end"]["error"] = "e" * 9000
    line = b.compact_line(out, "d")
    c = json.loads(line)
    assert len(line) < 4096 and c["value"] == out["value"] and c["roofline"]["frac"] is not None and c["cpu_baseline"]["value"] == out["cpu_baseline"]["value"]

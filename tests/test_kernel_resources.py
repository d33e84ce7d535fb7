"""What the compiler gave the hot-path kernels, read from the shipped code objects without a GPU (tools/kernel_resources.py): a flag or a source change
that pushes a fusion kernel into private memory (scratch) or over the LDS of a CU shows up here, before it shows up as a slower pass on the MI355X."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def rows():
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("no llvm-readelf")
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    rows = kr.kernels(os.path.join(ROOT, "scannet_amd", "libscanfuse.so"))
    for r, n in zip(rows, kr.demangle([r["name"] for r in rows])):
        r["short"] = kr.short(n)
    return rows


def test_fusion_kernels_live_in_registers(rows):
    hot = [r for r in rows if r["short"].split("<")[0] in ("k_integrate", "k_alloc_ray", "k_alloc", "k_prepass", "k_compactify", "k_inflate_tokens", "k_inflate_copy",
                                                          "k_mc_count", "k_mc_emit", "k_gather_where", "k_import", "k_jpeg_reconstruct")]
    names = {r["short"].split("<")[0] for r in hot}
    assert {"k_integrate", "k_alloc_ray", "k_prepass", "k_compactify", "k_inflate_tokens", "k_inflate_copy"} <= names, names
    for r in hot:
        assert r["scratch"] == 0 and r["vspill"] == 0, (r["short"], r["scratch"], r["vspill"])
        assert r["lds"] <= 160 * 1024, (r["short"], r["lds"])                       # one CU's LDS on gfx950
    integ = [r for r in hot if r["short"].startswith("k_integrate<")]
    assert len(integ) >= 20 and max(r["vgpr"] + r["agpr"] for r in integ) <= 96      # at least 5 waves per SIMD (512 // 96) for every variant
    timed = [r for r in integ if r["short"] in ("k_integrate<1, 2, true, 2, false, 4, true>", "k_integrate<1, 0, true, 2, false, 4, true>")]
    assert len(timed) == 2                                                           # the two kernels bench.py times (RGB-D, depth only; x-row lane layout)


def test_kernels_with_private_memory_are_the_known_ones(rows):
    own = [r for r in rows if "rocprim" not in r["short"] and "hipcub" not in r["short"]]
    assert len(own) >= 100
    assert {r["short"] for r in own if r["scratch"]} <= {"k_priority", "k_jpeg_huff", "k_pa_raster", "k_pa_raster_big"}   # DESIGN section 9 item 3 names the one on the hot path

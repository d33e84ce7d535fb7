"""Sanitizer jobs for the host code (SURVEY section 5 "Race detection / sanitizers"), run as part of the CPU suite:
  * tools/sanitize.py   AddressSanitizer + UndefinedBehaviorSanitizer build of the host translation units, the host tests against it
  * tools/tsan/run.sh   ThreadSanitizer run of sf_fuse_run (decode pool, pinned ring, copy streams, reaper thread) against an asynchronous
                        fake HIP runtime: real stream threads, events, "device memory" = host memory, device passes stubbed by checksums."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_asan_ubsan_host_suite():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sanitize.py")], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0 and "sanitize: clean" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_tsan_fuse_run_pipeline():
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "tsan", "run.sh"), "200", "8"], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0 and "tsan: clean" in r.stdout and r.stdout.count("checksum ok") == 3, (r.stdout[-3000:], r.stderr[-3000:])


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_tsan_host_thread_pools():
    """tools/tsan/run_host.sh: the thread pools that have no GPU in them -- the streaming .sens writer (producer against the background writer through a
    two-frame queue), the image export pool with its in-order progress, the mesh merge over key ranges and the PLY writer's slices -- under ThreadSanitizer."""
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "tsan", "run_host.sh")], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0 and "tsan: clean" in r.stdout and "merged" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_codec_mutation_fuzz_under_asan():
    """tools/fuzz_codecs.py: truncated, bit-flipped, spliced and length-poked JPEG / zlib / PNG / PLY / Occipital inputs through the
    ASan + UBSan build: every call returns, none trips a sanitizer (a short run here; the tool takes an iteration count)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_codecs.py"), "600", "11"], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0 and "fuzz: no sanitizer report" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])

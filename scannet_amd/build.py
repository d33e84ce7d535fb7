"""Builds libscanfuse.so in-tree with hipcc for gfx950 (no JIT cache: the .so travels with the snapshot)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libscanfuse.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the TSDF / Segmentator arithmetic is specified operation by operation;
# only explicit fmaf() may fuse.  HIP's default correctly-rounded fp32 divide / sqrt is relied upon.
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-Wall", "-Wno-unused-result",
          "-I" + os.path.join(ROOT, "include")] + os.environ.get("SCANFUSE_BUILD_FLAGS", "").split()   # e.g. -DSF_MEASURE_ABLATE for tools/gpu/r03_ablate.sh


# fuser.hip / calib.hip: the voxel pairs of the integrate kernels are two plain fp32 operations each (fuser_internal.h: packed fp32 buys no issue rate on
# gfx950); the SLP vectoriser would pack them again
PER_FILE = {} if "-DSF_PACKED_PAIRS" in COMMON else {"fuser.hip": ["-fno-slp-vectorize"], "calib.hip": ["-fno-slp-vectorize"]}


def sources():
    src = sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")) + glob.glob(os.path.join(HERE, "csrc", "*.cpp")))
    return [s for s in src if not os.path.basename(s).startswith("tool_")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    src = sources()
    hdr = glob.glob(os.path.join(HERE, "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    objdir = os.path.join(HERE, "_build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for s in src:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdr):
            cmd = [HIPCC, "--offload-arch=gfx950", "-c", s, "-o", o] + COMMON + PER_FILE.get(os.path.basename(s), [])
            if s.endswith(".cpp"):
                cmd = [HIPCC, "-x", "hip", "--offload-arch=gfx950", "-c", s, "-o", o] + COMMON
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (s, out))
        if verbose and out.strip():
            print(out)
    linked = False
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-o", LIB] + objs + ["-lpthread"]
        out = subprocess.run(cmd, capture_output=True, text=True)
        if out.returncode != 0:
            raise RuntimeError("link failed:\n" + out.stdout + out.stderr)
        linked = True
    build_tools(force, verbose)
    _record(src, [s for s, _ in procs], linked)
    return LIB


RECORD = os.path.join(HERE, "_build", "build_record.json")


def _record(src, compiled, linked):
    """What this call of build() did, next to the objects (it travels to the GPU box with them): which sources were compiled now, whether the library was
    linked now, where, with which compiler -- so that a reader of a test log can tell a library built by THIS run from one that shipped with the snapshot
    (Weak 12).  A call that finds everything fresh appends to the history and leaves the library alone."""
    import hashlib
    import json
    import socket
    import time
    try:
        ver = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout.splitlines()[0]
    except Exception:
        ver = "?"
    try:
        old = json.load(open(RECORD))
    except Exception:
        old = {}
    h = hashlib.sha256(open(LIB, "rb").read()).hexdigest()[:16] if os.path.exists(LIB) else None
    entry = {"when": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "host": socket.gethostname(), "sources": len(src),
             "compiled_now": [os.path.basename(s) for s in compiled], "linked_now": bool(linked), "hipcc": ver, "library_sha256_16": h}
    if compiled or linked or not old.get("last_build"):
        old["last_build"] = entry
    old["last_call"] = entry
    try:
        json.dump(old, open(RECORD, "w"), indent=1)
    except OSError:
        pass


def build_record():
    """The record _record() keeps (None when the library was never built by build())."""
    import json
    try:
        return json.load(open(RECORD))
    except Exception:
        return None


def build_tools(force=False, verbose=False):
    """Drop-in command-line tools (tool_*.cpp): plain C++ hosts linked against libscanfuse.so."""
    bindir = os.path.join(ROOT, "bin")
    for s in sorted(glob.glob(os.path.join(HERE, "csrc", "tool_*.cpp"))):
        name = os.path.basename(s)[len("tool_"):-len(".cpp")]
        exe = os.path.join(bindir, name)
        if force or _stale(exe, [s, LIB]):
            os.makedirs(bindir, exist_ok=True)
            cmd = ["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), s, "-o", exe, "-L" + HERE, "-lscanfuse",
                   "-Wl,-rpath," + HERE, "-Wl,-rpath,/opt/rocm/lib", "-lpthread"]
            out = subprocess.run(cmd, capture_output=True, text=True)
            if out.returncode != 0:
                raise RuntimeError("tool build failed (%s):\n%s" % (name, out.stdout + out.stderr))


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

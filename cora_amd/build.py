"""Builds cora_amd/lib/libcora_hip.so with hipcc for gfx950 (no GPU needed).

Sources: csrc/format_build.cpp (host format builder), csrc/kernels.hip (CDNA4
kernels), csrc/capi.hip (C ABI, include/cora_hip.h) and csrc/host/*.cpp (the
C++ host mirroring the reference's CORA::Problem / solveCORA interface)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcora_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = os.environ.get("CORA_EXTRA_HIPCC_FLAGS", "").split() + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(os.path.dirname(HERE), "include"), "-I" + CSRC]


def sources():
    srcs = [os.path.join(CSRC, f) for f in ("format_build.cpp", "trisolve_build.cpp", "kernels.hip", "capi.hip")]
    srcs += sorted(glob.glob(os.path.join(CSRC, "host", "*.cpp")))
    return srcs


def _deps():
    return sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "host", "*.h")) + \
        [os.path.join(os.path.dirname(HERE), "include", "cora_hip.h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in _deps())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for s in sources():
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        newest = max(os.path.getmtime(p) for p in _deps() if p.endswith(".h") or p == s)
        if not force and os.path.exists(o) and os.path.getmtime(o) > newest:
            continue
        cmd = [HIPCC] + FLAGS + (["-x", "hip"] if s.endswith(".hip") else []) + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on " + s)
        if verbose and out:
            sys.stderr.write(out.decode())
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

"""Builds cora_amd/lib/libcora_hip.so with hipcc for gfx950 (no GPU needed).

Sources: csrc/format_build.cpp (host format builder), csrc/kernels.hip (CDNA4
kernels), csrc/capi.hip (C ABI, include/cora_hip.h), csrc/p2p.hip (device-side
collectives over peer-mapped mailboxes) and csrc/host/*.cpp (the
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


# kernels.hip is compiled as several translation units in parallel (CORA_TU / CORA_LDG, see the head of the file)
KERNEL_PARTS = [("spmm_g0", 1, 1), ("spmm_g1", 1, 2), ("spmm_g2", 1, 4), ("spmm_g3", 1, 8), ("spmm_g4", 1, 16),
                ("spmm_g5", 1, 32), ("rows", 2, 63), ("tri_g0", 4, 1), ("tri_g1", 4, 2), ("tri_g2", 4, 4),
                ("tri_g3", 4, 8), ("tri_g4", 4, 16), ("tri_g5", 4, 32)]


def sources():
    srcs = [os.path.join(CSRC, f) for f in ("format_build.cpp", "trisolve_build.cpp", "kernels.hip", "capi.hip", "p2p.hip")]
    srcs += sorted(glob.glob(os.path.join(CSRC, "host", "*.cpp")))
    return srcs


def units():
    """(source, object name, extra flags) of every translation unit."""
    out = []
    for s in sources():
        if os.path.basename(s) == "kernels.hip":
            for name, tu, ldg in KERNEL_PARTS:
                out.append((s, "kernels_%s.o" % name, ["-DCORA_TU=%d" % tu, "-DCORA_LDG=%d" % ldg]))
        else:
            out.append((s, os.path.basename(s) + ".o", []))
    return out


def _deps():
    return sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "host", "*.h")) + \
        glob.glob(os.path.join(CSRC, "capi", "*.inc")) + glob.glob(os.path.join(CSRC, "kernels", "*.inc")) + \
        [os.path.join(os.path.dirname(HERE), "include", "cora_hip.h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in _deps())


def build(force=False, verbose=False):
    """Compiles what is stale and links.  CORA_REBUILD_UNITS=<prefix,prefix> forces the objects whose names start
    with one of the prefixes (variant builds of one kernel group: tools/variant.sh).  Lab: CORA_VARIANT=<name> builds
    lib/variants/<name>/libcora_hip.so instead -- the units of CORA_REBUILD_UNITS compiled with CORA_EXTRA_HIPCC_FLAGS,
    every other object taken from the main build -- which capi.load() picks up under CORA_LIB_VARIANT=<name>."""
    variant = os.environ.get("CORA_VARIANT")
    if not variant and not force and not needs_build() and not os.environ.get("CORA_REBUILD_UNITS"):
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    only = [u for u in os.environ.get("CORA_REBUILD_UNITS", "").split(",") if u]
    lib_out = LIB
    vdir = None
    if variant:
        vdir = os.path.join(LIBDIR, "variants", variant)
        os.makedirs(os.path.join(vdir, "obj"), exist_ok=True)
        lib_out = os.path.join(vdir, "libcora_hip.so")
    objs = []
    jobs = []
    for s, oname, extra in units():
        o = os.path.join(objdir, oname)
        if variant:
            if any(oname.startswith(u) for u in only):
                o = os.path.join(vdir, "obj", oname)
                objs.append(o)
                jobs.append((s, [HIPCC] + FLAGS + extra + (["-x", "hip"] if s.endswith(".hip") else []) + ["-c", s, "-o", o]))
            else:
                objs.append(o)   # the main build's object
            continue
        objs.append(o)
        newest = max(os.path.getmtime(p) for p in _deps() if p.endswith(".h") or p == s or
                     (p.endswith(".inc") and os.path.basename(os.path.dirname(p)) + ".hip" == os.path.basename(s)))
        forced = force or any(oname.startswith(u) for u in only)
        if not forced and os.path.exists(o) and os.path.getmtime(o) > newest:
            continue
        jobs.append((s, [HIPCC] + FLAGS + extra + (["-x", "hip"] if s.endswith(".hip") else []) + ["-c", s, "-o", o]))
    # the kernel units are the long poles: start them first; at most MAXJOBS compilers at a time.  Every compiler's
    # output is drained by subprocess.run (a child blocked on a full pipe would never exit).
    jobs.sort(key=lambda j: 0 if j[0].endswith("kernels.hip") else 1)
    maxjobs = int(os.environ.get("CORA_BUILD_JOBS", str(max(2, (os.cpu_count() or 4)))))

    def run(job):
        s, cmd = job
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        return s, r.returncode, r.stdout.decode(errors="replace")
    failed = None
    with ThreadPoolExecutor(max_workers=maxjobs) as pool:
        for s, rc, out in pool.map(run, jobs):
            if rc != 0 and failed is None:
                failed = s
                sys.stderr.write(out)
            elif verbose and out:
                sys.stderr.write(out)
    if failed:
        raise RuntimeError("hipcc failed on " + failed)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_out] + objs
    subprocess.check_call(cmd)
    return lib_out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

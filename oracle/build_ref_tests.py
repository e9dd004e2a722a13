"""Compiles the REFERENCE'S OWN test translation units, unmodified and from where they lie under /root/reference/tests,
against this repository's include/CORA/*.h and libcora_hip.so -- the drop-in boundary of the C++ host ("drops into
examples/ and tests/").  Outputs go to oracle/_ref/ only (git-ignored; they travel to the GPU box like any built
artefact, the reference's sources never do).  Catch2 and the reference's Eigen-based test helper are replaced by the
stand-ins under tests/drop_in/shim/ (test infrastructure written for this repository, see the headers there).

    python oracle/build_ref_tests.py        # in the build container (needs /root/reference)

tests/test_gpu_reference_tests.py runs the binaries on the GPU against the committed golden fixtures."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TESTS = "/root/reference/tests"
OUT = os.path.join(ROOT, "oracle", "_ref")
SHIM = os.path.join(ROOT, "tests", "drop_in", "shim")
# the reference's test files that use nothing but the Problem / solveCORA / parser API
UNITS = ["test_optimizer_helpers", "test_cora", "test_parse_pyfg", "test_certification", "test_geometry", "test_construct_problem",
         "test"]   # tests/test.cpp: getBlockCholeskyFactorization / blockCholeskySolve against a dense inverse


def available():
    return all(os.path.exists(os.path.join(REF_TESTS, u + ".cpp")) for u in UNITS)


def build(verbose=False):
    """Returns the list of binaries built (empty when the reference tree is not mounted)."""
    if not available():
        return []
    sys.path.insert(0, ROOT)
    from cora_amd import build as b
    lib = b.build()
    os.makedirs(OUT, exist_ok=True)
    built, errors = [], []
    for u in UNITS:
        out = os.path.join(OUT, "ref_" + u)
        src = os.path.join(REF_TESTS, u + ".cpp")
        deps = [src, lib, os.path.join(SHIM, "test_utils.h"), os.path.join(SHIM, "test_utils_shim.cpp"),
                os.path.join(SHIM, "catch_main.cpp"), os.path.join(SHIM, "catch2", "catch_test_macros.hpp"),
                os.path.join(SHIM, "catch2", "generators", "catch_generators.hpp")]
        if os.path.exists(out) and all(os.path.getmtime(out) > os.path.getmtime(d) for d in deps):
            built.append(out)
            continue
        cmd = [b.HIPCC, "-O1", "-std=c++17", "-I" + SHIM, "-I" + os.path.join(ROOT, "include"), src,
               os.path.join(SHIM, "test_utils_shim.cpp"), os.path.join(SHIM, "catch_main.cpp"),
               "-L" + os.path.dirname(lib), "-lcora_hip", "-Wl,-rpath,$ORIGIN/../../cora_amd/lib", "-o", out]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:  # the other units are still built; a stale binary of this one must not pass for a fresh one
            if os.path.exists(out):
                os.remove(out)
            errors.append("the reference's %s.cpp does not compile against include/CORA:\n%s" % (u, r.stdout[-4000:]))
            continue
        built.append(out)
    if errors:
        raise RuntimeError("\n".join(errors))
    return built


if __name__ == "__main__":
    print("\n".join(build(verbose=True)) or "reference tree not mounted: nothing built")

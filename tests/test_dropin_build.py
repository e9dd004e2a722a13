"""Drop-in boundary of the C++ host (north star: "keeping the CORA::Problem / solveCORA() C++ API ... so it
drops into examples/"): sources written against the reference's headers compile with -I<repo>/include and link
libcora_hip.so.  No GPU needed: compile + link only (running them is tests/test_gpu_datasets.py's business)."""
import os
import shutil
import subprocess

import pytest

from cora_amd import build as _build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_MAIN = "/root/reference/examples/main.cpp"   # only in the build container; never copied


def _compile(src, out, extra=()):
    lib = _build.build()
    cmd = [_build.HIPCC, "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include")] + list(extra) + [src,
           "-L" + os.path.dirname(lib), "-lcora_hip", "-Wl,-rpath," + os.path.dirname(lib), "-o", out]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert os.path.exists(out)


@pytest.mark.skipif(shutil.which(_build.HIPCC) is None and not os.path.exists(_build.HIPCC), reason="no hipcc")
def test_own_example_compiles_and_links(tmp_path):
    """The repository's own examples/main.cpp (the flow of the reference's example with timing and output options)."""
    _compile(os.path.join(ROOT, "examples", "main.cpp"), str(tmp_path / "cora_main"),
             ["-I" + os.path.join(ROOT, "cora_amd", "csrc", "host")])


@pytest.mark.skipif(shutil.which(_build.HIPCC) is None and not os.path.exists(_build.HIPCC), reason="no hipcc")
def test_symbol_and_eigen_style_surface(tmp_path):
    """include/CORA/Symbol.h:33-89 of the reference (conversions, Key comparisons, symIndex / symChar / symbol, namespace
    shorthand) and the element-wise sparse interface / comma initialisers / inverse of CORA_types.h: compiled against
    <CORA/...> and RUN (host code only)."""
    exe = str(tmp_path / "symbol_surface")
    _compile(os.path.join(ROOT, "tests", "drop_in", "symbol_surface.cpp"), exe)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=60)
    assert r.returncode == 0, r.stdout[-2000:]


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="the reference tree is only mounted in the build container")
def test_reference_examples_main_compiles_unmodified(tmp_path):
    """The reference's own examples/main.cpp, from where it lies, unmodified."""
    _compile(REF_MAIN, str(tmp_path / "ref_main"))

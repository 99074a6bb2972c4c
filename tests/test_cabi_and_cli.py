"""C-ABI library loads and exports what include/quack_hip.h declares; the CLI
reproduces the reference's command-line behaviour (no input files needed)."""
import ctypes
import os
import re
import subprocess

import pytest

import cases
from conftest import has_gpu

ROOT = cases.ROOT
QUACK = os.path.join(ROOT, "quack_amd", "host", "quack")


def declared_symbols(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qkh?_[a-z_0-9]+)\s*\(", text)))


def test_hip_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(os.path.join(ROOT, "quack_amd", "libquack_hip.so"))
    syms = declared_symbols("quack_hip.h")
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "libquack_hip.so lacks %s" % s
    lib.qk_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.qk_version()


def test_host_library_exports_its_header():
    lib = ctypes.CDLL(os.path.join(ROOT, "quack_amd", "libquack_host.so"))
    text = open(os.path.join(ROOT, "quack_amd", "host", "quack_host.h")).read()
    for s in sorted(set(re.findall(r"\b(qkh_[a-z_0-9]+)\s*\(", text))):
        assert hasattr(lib, s), s


def test_code_object_is_gfx950_only(tmp_path):
    # llvm-objdump --offloading unbundles the code objects next to its input: give it a link in tmp_path
    link = tmp_path / "libquack_hip.so"
    os.symlink(os.path.join(ROOT, "quack_amd", "libquack_hip.so"), link)
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", str(link)],
                         capture_output=True, text=True, cwd=tmp_path)
    archs = set(re.findall(r"gfx[0-9a-f]+", out.stdout))
    assert archs == {"gfx950"}, archs


@pytest.mark.skipif(has_gpu(), reason="checks the no-device error path")
def test_no_device_is_a_loud_error_not_a_fallback():
    import quack_amd
    with pytest.raises(quack_amd.HipUnavailable):
        quack_amd.Accumulator(0)
    with pytest.raises(quack_amd.HipUnavailable):
        quack_amd.read_fastq(cases.inp("kat.fq"))
    r = subprocess.run([QUACK, "-u", cases.inp("kat.fq")], capture_output=True, timeout=240)
    assert r.returncode == 1 and r.stdout == b"" and b"no HIP device" in r.stderr


def test_product_does_not_reference_the_oracle():
    """the product path must not include, link or call anything under oracle/"""
    for base, _, files in os.walk(os.path.join(ROOT, "quack_amd")):
        for f in files:
            if f.endswith((".c", ".h", ".hip", ".py", "Makefile")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle_" not in text and "liboracle" not in text, os.path.join(base, f)
    for lib in ("libquack_hip.so", "libquack_host.so"):
        out = subprocess.run(["ldd", os.path.join(ROOT, "quack_amd", lib)], capture_output=True, text=True).stdout
        assert "oracle" not in out


@pytest.mark.parametrize("name,argv", cases.load("cli_cases.tsv"), ids=[c[0] for c in cases.load("cli_cases.tsv")])
def test_cli_without_inputs(name, argv):
    g = os.path.join(cases.G, "cli", name)
    r = subprocess.run([QUACK] + argv, capture_output=True, cwd=cases.inp(""), timeout=240)
    assert r.stdout == open(g + ".out", "rb").read()
    assert r.stderr == open(g + ".err", "rb").read()
    assert r.returncode == int(open(g + ".rc").read())


def test_missing_input_gives_no_partial_svg():
    r = subprocess.run([QUACK, "-u", "/nonexistent/reads.fq.gz"], capture_output=True, timeout=240)
    assert r.returncode != 0 and r.stdout == b"" and r.stderr != b""

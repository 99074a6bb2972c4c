"""libquack_dropin.so: the reference's own seam as linkable symbols
(include/quack_dropin.h; quack.c:154 read_adapters, quack.c:180 read_fastq).

CPU tier: the library exports exactly those two symbols, and a caller that
re-declares the reference's structs itself (tests/c/dropin_caller.c) reproduces
the reference binary's SVGs when dropin.c + the C host run on the C-ABI test
double.  GPU tier (-m gpu): the same caller linked against the real
libquack_dropin.so on the MI355X.
"""
import os
import subprocess

import pytest

import cases

ROOT = cases.ROOT
HOST = os.path.join(ROOT, "quack_amd", "host")
CALLER = os.path.join(ROOT, "tests", "c", "dropin_caller.c")
CASES = [(n, a) for n, a in cases.load() if n in (
    "uniform100", "uniform100_adapters", "uniform100_named", "adapter100", "ragged100_adapters", "ragged100_gz2",
    "paired", "paired_adapters_named", "long40_adapters", "kat_adapters", "truncated100", "len500")]


def short_flags(argv):
    m = {"--unpaired": "-u", "--forward": "-1", "--reverse": "-2", "--adapters": "-a", "--name": "-n"}
    return [m.get(a, a) for a in argv]


def test_the_library_exports_exactly_the_reference_seam():
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "quack_amd", "libquack_dropin.so")],
                         capture_output=True, text=True, check=True).stdout
    syms = sorted(l.split()[-1] for l in out.splitlines() if " T " in l)
    assert syms == ["read_adapters", "read_fastq"]


def test_the_struct_images_are_the_references(tmp_path):
    """sizeof/offsetof of include/quack_dropin.h == quack.c:134-146 [776 B/position probed in SURVEY §8a]"""
    src = tmp_path / "s.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "quack_dropin.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(base_information), '
                   'offsetof(base_information, content), offsetof(base_information, length_count), '
                   'offsetof(base_information, kmer_count), sizeof(sequence_data), '
                   'offsetof(sequence_data, original_max_length), offsetof(sequence_data, number_of_sequences));}')
    exe = tmp_path / "s"
    subprocess.check_call(["gcc", "-I" + os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    assert subprocess.check_output([str(exe)], text=True).split() == ["776", "728", "760", "768", "32", "16", "24"]


@pytest.fixture(scope="module")
def caller_double(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("dropin") / "dropin_caller_double")
    src = [CALLER] + [os.path.join(HOST, f) for f in ("dropin.c", "pipeline.c", "reader.c", "source.c", "inflate_fast.c", "crc32_fold.c",
                                                      "pinflate.c", "render.c", "cli.c")] + \
          [os.path.join(ROOT, "tests", "c", "cabi_double.c"), os.path.join(ROOT, "oracle", "quack_oracle.c")]
    subprocess.check_call(["gcc", "-O1", "-g", "-std=c11", "-D_DEFAULT_SOURCE", "-D_POSIX_C_SOURCE=200809L", "-pthread",
                           "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I" + os.path.join(ROOT, "include"), "-I" + HOST, "-I" + os.path.join(ROOT, "oracle"),
                           "-o", exe] + src + ["-lz", "-lm"])
    return exe


@pytest.mark.parametrize("name,argv", CASES, ids=[c[0] for c in CASES])
def test_reference_side_caller_on_the_test_double(caller_double, name, argv):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1")
    if name.startswith("long40"):
        env["QK_DOUBLE_SLOT_BYTES"] = "45000"
    r = subprocess.run([caller_double] + short_flags(argv), capture_output=True, cwd=cases.inp(""), env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout == cases.golden_svg(name)


@pytest.fixture(scope="module")
def caller_gpu(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("dropin") / "dropin_caller")
    lib = os.path.join(ROOT, "quack_amd")
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-o", exe, CALLER, "-L" + lib, "-lquack_dropin", "-lquack_host",
                           "-lquack_hip", "-Wl,-rpath," + lib])
    return exe


@pytest.mark.gpu
@pytest.mark.parametrize("name,argv", CASES, ids=[c[0] for c in CASES])
def test_reference_side_caller_on_the_gpu(caller_gpu, name, argv):
    r = subprocess.run([caller_gpu] + short_flags(argv), capture_output=True, cwd=cases.inp(""), timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout == cases.golden_svg(name)


@pytest.mark.gpu
def test_unreadable_input_exits_loudly(caller_gpu):
    r = subprocess.run([caller_gpu, "-u", "/nonexistent.fq"], capture_output=True, timeout=120)
    assert r.returncode == 1 and r.stdout == b"" and b"cannot accumulate" in r.stderr

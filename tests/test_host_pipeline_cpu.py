"""CPU tier: the whole C host — cli.c, pipeline.c, reader.c, source.c, render.c —
end to end, linked against a TEST DOUBLE of the C-ABI (tests/c/cabi_double.c:
the calls the host makes, answered by the oracle, with 70 kB batch slots).
Checks the host's own logic (batching and slot alternation, fixed / ragged /
cache-line-aligned commits, sharding over several accumulators, the paired
threads, exit codes) against the reference's SVG bytes without a GPU.  The
product itself has no CPU path; on a GPU box tests/test_gpu_parity.py runs the
same cases through the real library."""
import os
import subprocess

import numpy as np
import pytest

import cases

HOST = os.path.join(cases.ROOT, "quack_amd", "host")
SRC = [os.path.join(HOST, f) for f in ("main.c", "cli.c", "pipeline.c", "reader.c", "source.c", "inflate_fast.c",
                                       "pinflate.c", "render.c")] + \
      [os.path.join(cases.ROOT, "tests", "c", "cabi_double.c"), os.path.join(cases.ROOT, "oracle", "quack_oracle.c")]


@pytest.fixture(scope="module")
def quack_double(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("dbl") / "quack_double")
    subprocess.check_call(
        ["gcc", "-O1", "-g", "-std=c11", "-D_DEFAULT_SOURCE", "-D_POSIX_C_SOURCE=200809L", "-pthread",
         "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
         "-I" + os.path.join(cases.ROOT, "include"), "-I" + HOST, "-I" + os.path.join(cases.ROOT, "oracle"),
         "-o", exe] + SRC + ["-lz", "-lm"])
    return exe


def run(exe, argv, **env):
    return subprocess.run([exe] + argv, capture_output=True, cwd=os.path.join(cases.G, "inputs"),
                          env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", **env), timeout=300)


@pytest.mark.parametrize("name,argv", cases.load(), ids=[c[0] for c in cases.load()])
def test_whole_host_reproduces_the_reference_svg(quack_double, name, argv):
    want = cases.golden_svg(name)
    for env in ({}, {"QUACK_DEVICES": "0,1,2"}, {"QK_DOUBLE_SLOT_BYTES": "4000000"}):
        if name.startswith("long40") and not env.get("QK_DOUBLE_SLOT_BYTES") and "QUACK_DEVICES" not in env:
            env = dict(env, QK_DOUBLE_SLOT_BYTES="45000")      # still several batches, but every read fits
        r = run(quack_double, argv, **env)
        assert r.returncode == 0, r.stderr[-3000:]
        assert r.stdout == want, (name, env)
        assert r.stderr == cases.golden_err(name)


def test_long_reads_switch_to_cache_line_batches(quack_double, tmp_path):
    """the pipeline pads long-read batches (QK_BATCH_ALIGNED128) from the second batch on"""
    g = np.random.default_rng(4)
    fq = tmp_path / "long.fq"
    with open(fq, "wb") as f:
        for r in range(300):
            n = int(g.integers(900, 5000))
            f.write(b"@r%d\n" % r + g.choice(np.frombuffer(b"ACGT", np.uint8), n).tobytes() + b"\n+\n" +
                    g.integers(35, 70, n).astype(np.uint8).tobytes() + b"\n")
    a = run(quack_double, ["-u", str(fq)], QK_DOUBLE_VERBOSE="1")
    b = run(quack_double, ["-u", str(fq)], QK_DOUBLE_VERBOSE="1", QUACK_NO_ALIGN="1")
    assert a.returncode == 0 and b.returncode == 0 and a.stdout == b.stdout and len(a.stdout) > 1000
    stats = lambda r: [int(t) for t in r.stderr.decode().split("[double]")[1].split() if t.isdigit()]
    commits, gapped, aligned, strided = stats(a)
    assert commits > 5 and gapped == aligned == commits - 1 and strided == 0, a.stderr     # all but the first batch
    assert stats(b)[1:] == [0, 0, 0]


def write_fastq(path, lens, g):
    with open(path, "wb") as f:
        for r, n in enumerate(lens):
            f.write(b"@r%d\n" % r + g.choice(np.frombuffer(b"ACGTN", np.uint8), n).tobytes() + b"\n+\n" +
                    g.integers(35, 70, n).astype(np.uint8).tobytes() + b"\n")


def test_trimmed_short_reads_switch_to_strided_batches(quack_double, tmp_path):
    """reads of nearly one length (trimmed Illumina): from the second batch on the pipeline lays them out at a
    fixed stride (qk_accum_commit_strided); a longer read widens the stride, a long one ends the mode; the
    SVG is the packed pipeline's (QUACK_NO_STRIDE=1) byte for byte"""
    g = np.random.default_rng(9)
    lens = np.where(g.random(6000) < 0.7, 150, g.integers(120, 150, 6000))
    lens[3000] = 170          # longer than the stride of 152: the stride grows to 172
    lens[4500] = 900          # not a short read any more: back to packed batches (and then to strided again)
    fq = tmp_path / "trimmed.fq"
    write_fastq(fq, lens, g)
    stats = lambda r: [int(t) for t in r.stderr.decode().split("[double]")[1].split() if t.isdigit()]
    for extra in ({}, {"QUACK_DEVICES": "0,1"}):
        extra = dict(extra, QK_DOUBLE_SLOT_BYTES="20000")
        a = run(quack_double, ["-u", str(fq), "-a", "adapters.fa"], QK_DOUBLE_VERBOSE="1", **extra)
        b = run(quack_double, ["-u", str(fq), "-a", "adapters.fa"], QK_DOUBLE_VERBOSE="1", QUACK_NO_STRIDE="1", **extra)
        assert a.returncode == 0 and b.returncode == 0, a.stderr[-2000:]
        assert a.stdout == b.stdout and len(a.stdout) > 1000
        commits, gapped, aligned, strided = stats(a)
        assert gapped == 0 and strided >= commits - 6 and strided > 10, a.stderr
        assert stats(b)[3] == 0


def test_equal_length_reads_stay_fixed_length_batches(quack_double, tmp_path):
    g = np.random.default_rng(10)
    fq = tmp_path / "fixed.fq"
    write_fastq(fq, [100] * 3000, g)
    a = run(quack_double, ["-u", str(fq)], QK_DOUBLE_VERBOSE="1")
    assert a.returncode == 0
    assert [int(t) for t in a.stderr.decode().split("[double]")[1].split() if t.isdigit()][3] == 0


def test_a_read_larger_than_a_slot_is_an_error_not_a_truncation(quack_double):
    r = run(quack_double, ["-u", "long40.fq.gz"], QK_DOUBLE_SLOT_BYTES="3000")
    assert r.returncode == 1 and r.stdout == b"" and b"exceeds the batch size" in r.stderr

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
SRC = [os.path.join(HOST, f) for f in ("main.c", "cli.c", "pipeline.c", "reader.c", "source.c", "inflate_fast.c", "crc32_fold.c",
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


def run(exe, argv, early=False, **env):
    """early=False: no reads are parsed while the accumulators start (QUACK_NO_EARLY), so that every batch
    goes through the pinned slots whose handling these tests are about"""
    base = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", QUACK_FULL_TEARDOWN="1")
    if not early:
        base["QUACK_NO_EARLY"] = "1"
    return subprocess.run([exe] + argv, capture_output=True, cwd=os.path.join(cases.G, "inputs"),
                          env=dict(base, **env), timeout=300)


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


def test_worker_process(quack_double):
    """by default the CLI accumulates in a worker process forked before the HIP runtime starts (the 0.13 s a GPU
    process takes to exit are then not the caller's): same bytes as in one process, for a single file, a pair (two
    tables through the pipe), long reads (a 31 MB table), a file without reads; and a worker that dies is reported —
    message on stderr, nothing on stdout, exit code 1"""
    base = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", QUACK_NO_EARLY="1")
    cwd = os.path.join(cases.G, "inputs")
    for name in ("uniform100_adapters", "paired_adapters_named", "long40", "badcrc800"):
        argv = dict(cases.load())[name]
        r = subprocess.run([quack_double] + argv, capture_output=True, cwd=cwd, env=base, timeout=300)
        one = subprocess.run([quack_double] + argv, capture_output=True, cwd=cwd, env=dict(base, QUACK_NO_FORK="1"), timeout=300)
        assert r.returncode == 0 and one.returncode == 0, (name, r.stderr[-2000:])
        assert r.stdout == one.stdout == cases.golden_svg(name) and r.stderr == one.stderr == cases.golden_err(name), name
    r = subprocess.run([quack_double, "-u", "/dev/null"], capture_output=True, cwd=cwd, env=base, timeout=60)
    assert r.returncode == 1 and r.stdout == b"" and b"no sequence data" in r.stderr
    r = subprocess.run([quack_double, "-u", "no_such_file.fq"], capture_output=True, cwd=cwd, env=base, timeout=60)
    assert r.returncode == 1 and r.stdout == b"" and b"cannot open" in r.stderr
    r = subprocess.run([quack_double, "-u", "uniform100.fq"], capture_output=True, cwd=cwd, env=dict(base, QUACK_TEST_WORKER_DIES="1"), timeout=60)
    assert r.returncode == 1 and r.stdout == b"" and b"killed by signal 6" in r.stderr, r.stderr


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
    commits, gapped, aligned, strided = stats(a)[:4]
    assert commits > 5 and gapped == aligned == commits - 1 and strided == 0, a.stderr     # all but the first batch
    assert stats(b)[1:4] == [0, 0, 0]


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
        commits, gapped, aligned, strided = stats(a)[:4]
        assert gapped == 0 and strided >= commits - 6 and strided > 10, a.stderr
        assert stats(a)[7] == strided       # every strided batch with 0xFF behind its reads (the double checks every pad byte)
        assert stats(b)[3] == 0


def test_equal_length_reads_stay_fixed_length_batches(quack_double, tmp_path):
    g = np.random.default_rng(10)
    fq = tmp_path / "fixed.fq"
    write_fastq(fq, [100] * 3000, g)
    a = run(quack_double, ["-u", str(fq)], QK_DOUBLE_VERBOSE="1")
    assert a.returncode == 0
    assert [int(t) for t in a.stderr.decode().split("[double]")[1].split() if t.isdigit()][3] == 0


def test_uniform_reads_of_an_odd_length_are_laid_out_at_a_padded_stride(quack_double, tmp_path):
    """round 4: uniform reads whose length is not a multiple of 4 (150 bp) with the adapter scan: from the second batch on
    (or from the first, for batches parsed while the accumulators start up) the pipeline writes them 152 bytes apart and
    commits them with qk_accum_commit_padded; a batch that is not uniform goes on as a strided one, very ragged reads end
    the mode; without adapters (the library does not ask for it) and with QUACK_NO_STRIDE nothing is padded; the SVG is
    the packed pipeline's byte for byte"""
    g = np.random.default_rng(21)
    stats = lambda r: [int(t) for t in r.stderr.decode().split("[double]")[1].split() if t.isdigit()]
    fq = tmp_path / "u150.fq"
    write_fastq(fq, [150] * 4000, g)
    for extra in ({}, {"QUACK_DEVICES": "0,1,2"}):
        extra = dict(extra, QK_DOUBLE_SLOT_BYTES="20000", QK_DOUBLE_VERBOSE="1")
        a = run(quack_double, ["-u", str(fq), "-a", "adapters.fa"], **extra)
        b = run(quack_double, ["-u", str(fq), "-a", "adapters.fa"], QUACK_NO_STRIDE="1", **extra)
        c = run(quack_double, ["-u", str(fq)], **extra)
        d = run(quack_double, ["-u", str(fq)], QK_DOUBLE_PAD_ALWAYS="1", **extra)
        e = run(quack_double, ["-u", str(fq), "-a", "adapters.fa"], early=True, QK_DOUBLE_CREATE_DELAY_MS="300",
                **dict(extra, QK_DOUBLE_SLOT_BYTES="4000000"))   # (slots that hold an early batch: it is re-laid into one)
        for r in (a, b, c, d, e):
            assert r.returncode == 0, r.stderr[-2000:]
        assert a.stdout == b.stdout == e.stdout and c.stdout == d.stdout and len(a.stdout) > 1000
        n_acc = len(extra.get("QUACK_DEVICES", "0").split(","))
        assert stats(a)[6] >= stats(a)[0] - n_acc and stats(a)[3] == 0, a.stderr        # all but the first batch (of every accumulator's turn)
        assert stats(b)[6] == 0 and stats(c)[6] == 0 and stats(d)[6] > 5
        assert sum(int(l.split("padded")[1].split()[0]) for l in e.stderr.decode().splitlines() if l.startswith("[double]")) > 0, e.stderr   # (any accumulator)
    # 150s, then a stretch trimmed to 120-149 (strided batches at the same stride), then mostly very short reads — a mean below 30 % of
    # the stride: packed again (round 5: with adapters the stride holds down to that fill; rounds 2-4 left it below 75 %)
    lens = np.concatenate([[150] * 1500, np.where(g.random(1500) < 0.6, 150, g.integers(120, 150, 1500)),
                           np.where(g.random(6000) < 0.08, 150, g.integers(5, 31, 6000)), [150] * 1500])
    fq2 = tmp_path / "mixed.fq"
    write_fastq(fq2, lens, g)
    a = run(quack_double, ["-u", str(fq2), "-a", "adapters.fa"], QK_DOUBLE_SLOT_BYTES="20000", QK_DOUBLE_VERBOSE="1")
    b = run(quack_double, ["-u", str(fq2), "-a", "adapters.fa"], QK_DOUBLE_SLOT_BYTES="20000", QK_DOUBLE_VERBOSE="1", QUACK_NO_STRIDE="1")
    assert a.returncode == 0 and b.returncode == 0 and a.stdout == b.stdout and len(a.stdout) > 1000
    commits, gapped, aligned, strided, copied, resized, padded = stats(a)[:7]
    assert padded > 15 and strided > 5 and commits - padded - strided > 5, a.stderr


def test_reads_parsed_while_the_accumulators_start_up(quack_double, tmp_path):
    """a slow HIP start-up (here: the double sleeps in qk_accum_create): the tokenizer fills heap batches
    meanwhile, the thread that made the accumulators carries them into the slots while the tokenizer parses on (round 5); same SVG, also when the process exits without teardown
    (the CLI's default) and when a read is too long for an early batch"""
    g = np.random.default_rng(12)
    fq = tmp_path / "early.fq"
    write_fastq(fq, g.integers(30, 200, 30000), g)
    stats = lambda r: [int(t) for t in r.stderr.decode().split("[double]")[1].split() if t.isdigit()]
    want = run(quack_double, ["-u", str(fq), "-a", "adapters.fa"], QK_DOUBLE_VERBOSE="1")
    assert want.returncode == 0 and stats(want)[4] == 0
    # (QUACK_EARLY_BYTES=20000: ~170 early batches for a pool of 48 buffers — the tokenizer waits for the other thread and reuses them)
    for extra in ({}, {"QUACK_DEVICES": "0,1,2"}, {"QUACK_EARLY_BYTES": "20000"}, {"QUACK_EARLY_BYTES": "20000", "QUACK_DEVICES": "0,1"}):
        got = run(quack_double, ["-u", str(fq), "-a", "adapters.fa"], early=True, QK_DOUBLE_CREATE_DELAY_MS="300",
                  QK_DOUBLE_VERBOSE="1", **extra)
        assert got.returncode == 0, got.stderr[-2000:]
        assert got.stdout == want.stdout
        copied = sum(int(l.split("copied")[1].split()[0]) for l in got.stderr.decode().splitlines() if l.startswith("[double]"))
        if "QUACK_EARLY_BYTES" in extra:             # (small early batches fit a slot of the double: copied into it and committed)
            assert sum(int(l.split("commits")[1].split()[0]) for l in got.stderr.decode().splitlines() if l.startswith("[double]")) > 150, got.stderr
        else:
            assert copied >= 1, got.stderr           # early batches went through the copying feed
    env = dict(os.environ, QK_DOUBLE_CREATE_DELAY_MS="200")
    fast = subprocess.run([quack_double, "-u", str(fq), "-a", "adapters.fa"], capture_output=True,
                          cwd=os.path.join(cases.G, "inputs"), env=env, timeout=300)
    assert fast.returncode == 0 and fast.stdout == want.stdout
    # a read longer than an early batch (shrunk to 10 kB here) waits, parked, for a pinned slot
    big = tmp_path / "big.fq"
    with open(big, "wb") as f:
        f.write(open(fq, "rb").read()[:200000].rsplit(b"\n@", 1)[0] + b"\n")
        f.write(b"@big\n" + b"A" * 20000 + b"\n+\n" + b"I" * 20000 + b"\n")
        f.write(open(fq, "rb").read())
    a = run(quack_double, ["-u", str(big)], early=True, QK_DOUBLE_CREATE_DELAY_MS="300", QUACK_EARLY_BYTES="10000",
            QK_DOUBLE_VERBOSE="1", QUACK_VERBOSE="1")
    b = run(quack_double, ["-u", str(big)])
    assert a.returncode == 0 and b.returncode == 0 and a.stdout == b.stdout and len(a.stdout) > 1000
    import re
    n_early = int(re.search(rb"(\d+) early batches", a.stderr).group(1))
    commits = int(a.stderr.decode().split("[double]")[1].split()[1])
    assert 1 <= n_early < commits, a.stderr            # early batches (they fit a slot here: copied into it), then the rest


def test_early_batch_queue_under_tsan(tmp_path):
    """Round 5: the queue of early (heap) batches between the tokenizer and the thread that creates the accumulators and then
    carries them into the slots — ThreadSanitizer over the whole host with the C-ABI double: a slow start-up, the default early
    batches, small ones (the pool of buffers goes round), several accumulators, a gzip file through the decoder threads"""
    exe = str(tmp_path / "quack_double_tsan")
    subprocess.check_call(
        ["gcc", "-O1", "-g", "-std=c11", "-D_DEFAULT_SOURCE", "-D_POSIX_C_SOURCE=200809L", "-pthread",
         "-fsanitize=thread", "-fno-omit-frame-pointer",
         "-I" + os.path.join(cases.ROOT, "include"), "-I" + HOST, "-I" + os.path.join(cases.ROOT, "oracle"),
         "-o", exe] + SRC + ["-lz", "-lm"])
    g = np.random.default_rng(13)
    fq = tmp_path / "early.fq"
    write_fastq(fq, g.integers(30, 200, 20000), g)
    import gzip
    gz = tmp_path / "early.fq.gz"
    with open(fq, "rb") as f, gzip.open(gz, "wb", compresslevel=6) as o:
        o.write(f.read())
    base = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1", QUACK_FULL_TEARDOWN="1", QK_DOUBLE_CREATE_DELAY_MS="200")
    outs = []
    for extra in ({}, {"QUACK_EARLY_BYTES": "20000"}, {"QUACK_EARLY_BYTES": "20000", "QUACK_DEVICES": "0,1"},
                  {"QUACK_NO_EARLY": "1"}):
        for path in (fq, gz):
            r = subprocess.run([exe, "-u", str(path), "-a", "adapters.fa"], capture_output=True, cwd=os.path.join(cases.G, "inputs"),
                               env=dict(base, QUACK_PGZIP_CHUNK_KB="64", QUACK_THREADS="4", **extra), timeout=600)
            assert r.returncode == 0 and b"ThreadSanitizer" not in r.stderr, (extra, r.stderr[-3000:])
            outs.append(r.stdout)
    assert len(outs[0]) > 1000
    # (the SVG names its input: the plain file's runs agree among themselves, and so do the .gz file's)
    assert all(o == outs[0] for o in outs[0::2]) and all(o == outs[1] for o in outs[1::2])


def test_uncompressed_files_are_read_by_a_worker_pool(quack_double, tmp_path):
    """plain files of two blocks (8 MiB) and more: a dispatcher hands 4 MiB file ranges to workers that pread() them
    and index their lines; same SVG as the one-thread read(2) producer, also when records straddle every block edge"""
    g = np.random.default_rng(21)
    fq = tmp_path / "plain.fq"
    write_fastq(fq, g.integers(40, 160, 70000), g)
    assert os.path.getsize(fq) > 3 * (4 << 20)
    want = run(quack_double, ["-u", str(fq), "-a", "adapters.fa"], QUACK_NO_PLAIN_POOL="1", QK_DOUBLE_SLOT_BYTES="3000000")
    assert want.returncode == 0 and len(want.stdout) > 1000
    for env in ({}, {"QUACK_THREADS": "2"}, {"QUACK_DEVICES": "0,1"}):
        got = run(quack_double, ["-u", str(fq), "-a", "adapters.fa"], QK_DOUBLE_SLOT_BYTES="3000000", QUACK_VERBOSE="1", **env)
        assert got.returncode == 0, got.stderr[-2000:]
        assert got.stdout == want.stdout, env


def test_a_read_larger_than_a_slot_gets_bigger_slots(quack_double):
    """a batch holds whole reads: when one does not fit, the accumulator's slots grow (qk_accum_resize_slots) —
    the reference has no length limit (quack.c:194-198) — and if they cannot, that is an error, not a truncation"""
    want = cases.golden_svg("long40")
    r = run(quack_double, ["-u", "long40.fq.gz"], QK_DOUBLE_SLOT_BYTES="3000", QK_DOUBLE_VERBOSE="1")
    assert r.returncode == 0 and r.stdout == want, r.stderr[-2000:]
    assert int(r.stderr.decode().split("resized")[1].split()[0]) >= 1
    r = run(quack_double, ["-u", "long40.fq.gz"], QK_DOUBLE_SLOT_BYTES="3000", QUACK_DEVICES="0,1")
    assert r.returncode == 0 and r.stdout == want
    r = run(quack_double, ["-u", "long40.fq.gz"], QK_DOUBLE_SLOT_BYTES="3000", QK_DOUBLE_NO_RESIZE="1")
    assert r.returncode == 1 and r.stdout == b"" and b"does not fit a batch" in r.stderr


# ---------------------------------------------------------------- damaged gzip trailers (CRC-32 / ISIZE)
PRODUCERS = {"pgzip": dict(QUACK_PGZIP_CHUNK_KB="4", QUACK_THREADS="4"), "inflate_fast": dict(QUACK_NO_PGZIP="1"),
             "zlib": dict(QUACK_ZLIB="1")}


@pytest.mark.parametrize("producer", sorted(PRODUCERS))
@pytest.mark.parametrize("name", ["badcrc800", "badlen800", "badcrc_first_of_two"])
def test_a_member_whose_trailer_does_not_match_ends_the_stream_like_gzread(quack_double, name, producer):
    """every producer stops where the reference (zlib's gzread under kseq's 16 KiB reads) stops: at the last
    multiple of 16384 before the end of the damaged member — the goldens are the reference binary's output"""
    argv = dict(cases.load())[name]
    r = run(quack_double, argv, **PRODUCERS[producer])
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout == cases.golden_svg(name)


def bgzf_bytes(data, member=20000):
    import struct
    import zlib
    out = b""
    for a in list(range(0, len(data), member)) + [len(data)]:      # (+ the empty end-of-file member)
        piece = data[a:a + member]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(piece) + c.flush()
        out += (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(body) + 25) + body +
                struct.pack("<II", zlib.crc32(piece), len(piece)))
    return out


def member_starts(blob):
    return [i for i in range(len(blob) - 3) if blob[i:i + 4] in (b"\x1f\x8b\x08\x04", b"\x1f\x8b\x08\x00") and
            (i == 0 or True)]


@pytest.mark.parametrize("kind", ["bgzf", "members"])
def test_damaged_member_in_the_middle_of_many(quack_double, tmp_path, kind):
    """many members, one of them with a flipped CRC bit, at every slice / block geometry: the own producers
    (BGZF workers, speculative multi-threaded inflate, serial decoder) deliver exactly what zlib delivers —
    also when the cut lies in the chunk BEFORE the one that holds the damaged trailer (the 16 KiB hold-back)"""
    import gzip
    import io
    g = np.random.default_rng(31)
    fq = tmp_path / "many.fq"
    write_fastq(fq, g.integers(60, 120, 4000), g)
    text = open(fq, "rb").read()
    for victim in (3, 9, 17):
        if kind == "bgzf":
            blob = bytearray(bgzf_bytes(text, member=16000 + 123 * victim))
        else:
            buf = io.BytesIO()
            step = 5000 + 777 * victim
            for a in range(0, len(text), step):
                with gzip.GzipFile(fileobj=buf, mode="wb", mtime=0) as z:
                    z.write(text[a:a + step])
            blob = bytearray(buf.getvalue())
        # flip a bit in the CRC field of member `victim`: 8 bytes before the next member's magic
        starts = [i for i in range(len(blob) - 10) if blob[i:i + 3] == b"\x1f\x8b\x08" and blob[i + 8] in (0, 2) and blob[i + 9] in (0xff, 3)]
        assert len(starts) > victim + 1, len(starts)
        blob[starts[victim + 1] - 7] ^= 0x04
        path = tmp_path / ("damaged_%s_%d.fq.gz" % (kind, victim))
        path.write_bytes(bytes(blob))
        want = run(quack_double, ["-u", str(path)], QUACK_ZLIB="1")
        assert want.returncode == 0 and len(want.stdout) > 1000
        for env in (dict(QUACK_PGZIP_CHUNK_KB="4", QUACK_THREADS="5"), dict(QUACK_PGZIP_CHUNK_KB="16", QUACK_THREADS="2"),
                    dict(QUACK_NO_PGZIP="1", QUACK_NO_BGZF="1"), dict(QUACK_THREADS="3")):
            got = run(quack_double, ["-u", str(path)], **env)
            assert got.returncode == 0, got.stderr[-1000:]
            assert got.stdout == want.stdout, (kind, victim, env)


def test_pipeline_threads_under_tsan(tmp_path):
    """ThreadSanitizer build of the whole host on the test double: the accumulator-creator thread beside the
    tokenizer, the copy threads and the reaper of the early batches, the two mate threads of a pair, several
    accumulators per file"""
    exe = str(tmp_path / "quack_double_tsan")
    subprocess.check_call(
        ["gcc", "-O1", "-g", "-std=c11", "-D_DEFAULT_SOURCE", "-D_POSIX_C_SOURCE=200809L", "-pthread", "-fsanitize=thread",
         "-I" + os.path.join(cases.ROOT, "include"), "-I" + HOST, "-I" + os.path.join(cases.ROOT, "oracle"),
         "-o", exe] + SRC + ["-lz", "-lm"])
    base = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:exitcode=66", QUACK_FULL_TEARDOWN="1", QK_DOUBLE_CREATE_DELAY_MS="150")
    g = np.random.default_rng(22)
    plain = tmp_path / "plain9.fq"
    write_fastq(plain, g.integers(100, 200, 30000), g)      # > 8 MiB: the plain-file worker pool
    want_plain = subprocess.run([exe, "-u", str(plain)], capture_output=True, env=dict(base, QUACK_NO_PLAIN_POOL="1"), timeout=600)
    got_plain = subprocess.run([exe, "-u", str(plain)], capture_output=True, env=base, timeout=600)
    assert want_plain.returncode == 0 and got_plain.returncode == 0, got_plain.stderr[-4000:]
    assert got_plain.stdout == want_plain.stdout and b"ThreadSanitizer" not in got_plain.stderr
    for name, env in (("paired_adapters_named", {}), ("ragged100_adapters", {"QUACK_DEVICES": "0,1,2", "QUACK_EARLY_BYTES": "3000"}),
                      ("long40", {"QK_DOUBLE_SLOT_BYTES": "3000"}), ("uniform100_gz", {"QUACK_THREADS": "4"})):
        argv = dict(cases.load())[name]
        r = subprocess.run([exe] + argv, capture_output=True, cwd=os.path.join(cases.G, "inputs"), env=dict(base, **env), timeout=600)
        assert r.returncode == 0, (name, r.stderr[-4000:])
        assert r.stdout == cases.golden_svg(name) and b"ThreadSanitizer" not in r.stderr

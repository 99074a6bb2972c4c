"""Product tokenizer/batcher (quack_amd/host/reader.c) vs the oracle's
tokenizer, on the fixtures and on generated hostile text."""
import ctypes
import gzip
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import cases
import oracle_binding as ob
from quack_amd import _capi


def product_tokenize(path, cap_bytes=1 << 20, cap_reads=1 << 16):
    H = _capi.host()
    r = H.qkh_reader_open(os.fsencode(path))
    assert r
    out, batches = [], 0
    seq = np.zeros(cap_bytes + 16, np.uint8)
    qual = np.zeros(cap_bytes + 16, np.uint8)
    off = np.zeros(cap_reads + 1, np.uint64)
    try:
        while not H.qkh_reader_done(r):
            total, uni = ctypes.c_uint64(), ctypes.c_uint32()
            n = H.qkh_reader_fill(r, seq.ctypes.data, qual.ctypes.data, off.ctypes.data, cap_bytes, cap_reads,
                                  ctypes.byref(total), ctypes.byref(uni))
            assert n >= 0, n
            assert int(off[n]) == total.value
            lens = np.diff(off[:n + 1].astype(np.int64))
            if n and (lens == lens[0]).all() and lens[0] > 0:
                assert uni.value == lens[0]
            else:
                assert uni.value == 0
            for i in range(n):
                a, b = int(off[i]), int(off[i + 1])
                out.append((seq[a:b].tobytes(), qual[a:b].tobytes()))
            batches += 1
    finally:
        H.qkh_reader_close(r)
    return out, batches


def oracle_tokenize(path):
    recs, _ = ob.tokenize(path)
    return [(s, q if q is not None else b"\0" * len(s)) for s, q in recs]


FILES = sorted(f for f in os.listdir(cases.inp("")) if ".f" in f)


@pytest.mark.parametrize("fname", FILES)
def test_fixtures(fname):
    got, _ = product_tokenize(cases.inp(fname))
    assert got == oracle_tokenize(cases.inp(fname))


@pytest.mark.parametrize("cap_bytes,cap_reads", [(200, 1000), (90, 1000), (100000, 3), (4096, 7)])
def test_small_batches_park_and_resume(cap_bytes, cap_reads):
    """records that do not fit the rest of a batch open the next one"""
    want = oracle_tokenize(cases.inp("ragged100.fq"))
    got, batches = product_tokenize(cases.inp("ragged100.fq"), cap_bytes, cap_reads)
    assert got == want and batches > 1


def test_read_longer_than_batch_is_an_error(tmp_path):
    H = _capi.host()
    r = H.qkh_reader_open(os.fsencode(cases.inp("uniform100.fq")))
    seq = np.zeros(64, np.uint8)
    off = np.zeros(8, np.uint64)
    total, uni = ctypes.c_uint64(), ctypes.c_uint32()
    n = H.qkh_reader_fill(r, seq.ctypes.data, seq.ctypes.data, off.ctypes.data, 30, 4, ctypes.byref(total),
                          ctypes.byref(uni))
    H.qkh_reader_close(r)
    assert n == -4


line = st.text(alphabet="ACGTNacgtn@+>I5#! \r\t", min_size=0, max_size=12)


@settings(max_examples=300, deadline=None)
@given(st.lists(line, min_size=0, max_size=30), st.sampled_from(["\n", "\r\n"]), st.booleans())
def test_hostile_text(tmp_path_factory, lines, nl, final_newline):
    text = nl.join(lines) + (nl if final_newline else "")
    p = tmp_path_factory.mktemp("h") / "x.fq"
    p.write_bytes(text.encode())
    got, _ = product_tokenize(str(p))
    assert got == oracle_tokenize(str(p))


@settings(max_examples=60, deadline=None)
@given(st.lists(st.tuples(st.integers(0, 70), st.booleans()), min_size=1, max_size=40), st.integers(0, 2 ** 32))
def test_wellformed_records_gz_multimember(tmp_path_factory, shape, seed):
    rng = np.random.default_rng(seed)
    recs = []
    for n, multiline in shape:
        s = "".join(rng.choice(list("ACGTN"), n)) if n else ""
        q = "".join(chr(33 + int(v)) for v in rng.integers(0, 60, n))
        if multiline and n > 4:
            recs.append("@r\n%s\n%s\n+\n%s\n%s\n" % (s[:n // 2], s[n // 2:], q[:3], q[3:]))
        else:
            recs.append("@r x\n%s\n+\n%s\n" % (s, q))
    data = "".join(recs).encode()
    p = tmp_path_factory.mktemp("g") / "x.fq.gz"
    with open(p, "wb") as f:
        half = len(data) // 2
        f.write(gzip.compress(data[:half]) + gzip.compress(data[half:]))
    got, _ = product_tokenize(str(p), cap_bytes=300, cap_reads=9)
    assert got == oracle_tokenize(str(p))
    assert len(got) == len(shape)


def hostile_big(seed, n=20000):
    """a large FASTQ (several source blocks, line index in use) with everything the general parser handles sprinkled into
    long runs of plain four-line records"""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        ln = int(rng.integers(1, 300))
        s = "".join(rng.choice(list("ACGTNacgt"), ln))
        q = "".join(chr(33 + int(v)) for v in rng.integers(0, 60, ln))
        kind = rng.integers(0, 400)
        if kind == 0 and ln > 6:
            out.append("@r%d\n%s\n%s\n+\n%s\n%s\n" % (i, s[:ln // 2], s[ln // 2:], q[:5], q[5:]))     # multi-line
        elif kind == 1:
            out.append("@r%d\r\n%s\r\n+\r\n%s\r\n" % (i, s, q))                                      # CRLF
        elif kind == 2:
            out.append("\n\n@r%d x y\n%s\n+r%d again\n%s\n" % (i, s, i, q))                             # blank lines, text on both header lines
        elif kind == 3:
            out.append(">f%d\n%s\n" % (i, s))                                                             # a FASTA record in between
        elif kind == 4 and ln > 1:
            out.append("@r%d\n%s\n+\n@%s\n" % (i, s, q[1:]))                                             # a quality line that starts with '@'
        else:
            out.append("@r%d\n%s\n+\n%s\n" % (i, s, q))
    return "".join(out).encode()


@pytest.mark.parametrize("form", ["plain", "gz"])
def test_large_hostile_files(tmp_path, form):
    """round 4: 20,000 records with anomalies between long runs of plain ones, plain and as a two-member .gz, at three batch
    geometries; the second file ends at a malformed record two thirds in (quack.c:193: the stream stops there)"""
    for seed in (1, 2):
        data = hostile_big(seed)
        if seed == 2:
            cut = data.index(b"@r13000\n")
            data = data[:cut] + b"@bad\nACGTACGT\n+\nIIII\n" + data[cut:]
        p = tmp_path / ("h%d.fq%s" % (seed, ".gz" if form == "gz" else ""))
        if form == "gz":
            with open(p, "wb") as f:
                third = len(data) // 3
                f.write(gzip.compress(data[:third], 1) + gzip.compress(data[third:], 1))
        else:
            p.write_bytes(data)
        want = oracle_tokenize(str(p))
        assert len(want) > 12000
        for cap_bytes, cap_reads in ((1 << 20, 1 << 16), (70000, 300), (1 << 20, 1000)):
            got, _ = product_tokenize(str(p), cap_bytes, cap_reads)
            assert got == want, (seed, cap_bytes, cap_reads)


def test_adapter_table_matches_oracle():
    import quack_amd
    for f in ("adapters.fa", "adapters.fa.gz", "kat_adapter.fa"):
        bits = quack_amd.read_adapters(cases.inp(f))
        np.testing.assert_array_equal(bits, ob.kmers_to_bitset(ob.kmers_from_file(cases.inp(f))))
    assert int(np.unpackbits(quack_amd.read_adapters(cases.inp("kat_adapter.fa")).view(np.uint8)).sum()) == 3

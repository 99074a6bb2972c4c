"""Host feed: quack_amd/host/inflate_fast.c must decode exactly what zlib
decodes — every block type, strategy, window size, multi-member files, and any
output chunking (matches cut by the end of an output block)."""
import ctypes
import gzip
import io
import os
import random
import zlib

import numpy as np
import pytest

from quack_amd import _capi
import cases
from test_reader_differential import product_tokenize

L = _capi.host()
L.qkh_inflate_read.restype = ctypes.c_long
L.qkh_inflate_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t]
L.qkh_inflate_init.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]


def inflate(comp, expect_len, chunk):
    z = ctypes.create_string_buffer(1 << 18)          # > sizeof(qkh_inflate)
    cbuf = ctypes.create_string_buffer(comp, len(comp))
    L.qkh_inflate_init(z, cbuf, len(comp))
    cap = expect_len + 70000
    out = ctypes.create_string_buffer(cap)
    base, pos, n = ctypes.addressof(out), 0, 0
    while True:
        n = L.qkh_inflate_read(z, base + pos, min(chunk, cap - pos), min(pos, 32768))
        if n <= 0:
            break
        pos += n
    return out.raw[:pos], n


def gz(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=31, memlevel=8):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, memlevel, strategy)
    return c.compress(data) + c.flush()


rng = random.Random(1)


def fastq(n, length):
    out = []
    for i in range(n):
        s = "".join(rng.choice("ACGT") for _ in range(length))
        q = "".join(chr(33 + rng.randint(2, 41)) for _ in range(length))
        out.append("@r%d\n%s\n+\n%s\n" % (i, s, q))
    return "".join(out).encode()


DATA = {
    "empty": b"", "one": b"A", "short": b"hello world\n" * 3,
    "fastq": fastq(800, 150), "runs": (b"I" * 500 + b"\n") * 400, "zeros": bytes(100000),
    "random": bytes(rng.getrandbits(8) for _ in range(60000)),
    "lowent": bytes(rng.choice(b"AB") for _ in range(100000)),
    "text": b"the quick brown fox jumps over the lazy dog. " * 3000,
    "period3": b"abc" * 30000, "period7": b"abcdefg" * 15000,
}
STRATEGIES = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED]


@pytest.mark.parametrize("name", sorted(DATA))
def test_against_zlib_all_block_types_and_chunkings(name):
    data = DATA[name]
    for level in (0, 1, 6, 9):
        for strat in STRATEGIES:
            for memlevel in (1, 8):
                comp = gz(data, level, strat, 31, memlevel)
                for chunk in (1 << 22, 4096, 259, 7, 1):
                    if chunk < 259 and len(data) > 20000:
                        continue
                    got, last = inflate(comp, len(data), chunk)
                    assert last == 0 and got == data, (name, level, strat, memlevel, chunk)


def test_multi_member_header_fields_and_trailing_garbage():
    d1, d2 = DATA["fastq"][:60000], DATA["text"][:30000]
    b = io.BytesIO()
    with gzip.GzipFile(filename="some_name.fq", fileobj=b, mode="wb", mtime=5) as g:
        g.write(d1)
    multi = b.getvalue() + gz(d2, 9) + gz(b"") + gz(d1, 1)
    for chunk in (1 << 22, 1000, 3):
        got, last = inflate(multi, 2 * len(d1) + len(d2), chunk)
        assert last == 0 and got == d1 + d2 + d1
    got, _ = inflate(multi + b"\0\0\0garbage", 2 * len(d1) + len(d2), 1 << 22)
    assert got == d1 + d2 + d1                              # like zlib: trailing garbage ignored
    assert inflate(b"not gzip at all", 100, 4096) == (b"", -1)


@pytest.mark.parametrize("wbits", [25, 27, 29])
def test_small_windows(wbits):
    comp = gz(DATA["text"], 6, zlib.Z_DEFAULT_STRATEGY, wbits)
    assert inflate(comp, len(DATA["text"]), 1 << 20)[0] == DATA["text"]


def test_corrupt_and_truncated_streams_never_crash():
    data = DATA["fastq"]
    comp = bytearray(gz(data, 6))
    for _ in range(300):
        c = bytearray(comp)
        p = rng.randrange(10, len(c))
        c[p] ^= 1 << rng.randrange(8)
        inflate(bytes(c), len(data), 1 << 16)               # any result, no crash, bounded output
    for cut in (5, 18, 19, 100, len(comp) // 2, len(comp) - 4):
        got, _ = inflate(bytes(comp[:cut]), len(data), 1 << 16)
        ref = zlib.decompressobj(31).decompress(bytes(comp[:cut]))
        assert got[:len(ref)] == ref[:len(got)]             # a prefix of the truth


def test_reader_gives_the_same_records_through_zlib_fallback(monkeypatch):
    """QUACK_ZLIB=1 routes gzip through zlib's gzread; records must not change"""
    for f in ("uniform100.fq.gz", "ragged100_2member.fq.gz", "long40.fq.gz"):
        fast, _ = product_tokenize(cases.inp(f))
        monkeypatch.setenv("QUACK_ZLIB", "1")
        slow, _ = product_tokenize(cases.inp(f))
        monkeypatch.delenv("QUACK_ZLIB")
        assert fast == slow and len(fast) > 0


# ------------------------------------------------------------------ BGZF
def bgzf(data, block=65280, eof=True, level=6):
    out = bytearray()
    chunks = [data[i:i + block] for i in range(0, len(data), block)] + ([b""] if eof else [])
    for c in chunks:
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        cd = co.compress(c) + co.flush()
        bsize = 18 + len(cd) + 8 - 1
        out += bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 66, 67, 2, 0, bsize & 255, bsize >> 8])
        out += cd + zlib.crc32(c).to_bytes(4, "little") + len(c).to_bytes(4, "little")
    return bytes(out)


def source_kind(path):
    H = _capi.host()
    H.qkh_source_open.restype = ctypes.c_void_p
    H.qkh_source_open.argtypes = [ctypes.c_char_p]
    H.qkh_source_kind.restype = ctypes.c_char_p
    H.qkh_source_kind.argtypes = [ctypes.c_void_p]
    H.qkh_source_close.argtypes = [ctypes.c_void_p]
    s = H.qkh_source_open(os.fsencode(path))
    k = H.qkh_source_kind(s).decode()
    H.qkh_source_close(s)
    return k


@pytest.mark.parametrize("block,threads", [(65280, "4"), (700, "3"), (65280, "1"), (5000, "16")])
def test_bgzf_parallel_inflate_gives_the_same_records(tmp_path, monkeypatch, block, threads):
    """a BGZF file is decoded by a worker pool, in order; the records must be
    those of the same text as ordinary gzip"""
    monkeypatch.setenv("QUACK_THREADS", threads)
    text = fastq(30000, 150)                    # ~9.4 MB: several 4 MiB ring blocks
    p_b, p_g = tmp_path / "x.fq.bgz", tmp_path / "x.fq.gz"
    p_b.write_bytes(bgzf(text, block))
    p_g.write_bytes(gz(text))
    assert source_kind(str(p_b)).startswith("bgzf x")
    assert source_kind(str(p_g)) == ("inflate_fast" if threads == "1" else "pgzip x" + threads)
    a, _ = product_tokenize(str(p_b), 1 << 22, 1 << 16)
    b, _ = product_tokenize(str(p_g), 1 << 22, 1 << 16)
    assert len(a) == 30000 and a == b


def test_bgzf_followed_by_ordinary_gzip_and_truncation(tmp_path):
    t1, t2 = fastq(3000, 100), fastq(2000, 80)
    p = tmp_path / "mixed.gz"
    p.write_bytes(bgzf(t1, 4000, eof=False) + gz(t2))
    want, _ = product_tokenize_text(tmp_path, t1 + t2)
    got, _ = product_tokenize(str(p))
    assert got == want and len(got) == 5000
    # a BGZF file cut in the middle of a member: everything before it is delivered
    whole = bgzf(t1, 4000)
    p.write_bytes(whole[:len(whole) // 2])
    got, _ = product_tokenize(str(p))
    assert 0 < len(got) < 3000 and got == want[:len(got)]


def product_tokenize_text(tmp_path, text):
    q = tmp_path / "plain.fq"
    q.write_bytes(text)
    return product_tokenize(str(q))


# ------------------------------------------------ ordinary gzip, several threads
def source_bytes(path, final_kind=False):
    """the whole decompressed stream as the tokenizer would see it"""
    H = _capi.host()
    H.qkh_source_open.restype = ctypes.c_void_p
    H.qkh_source_open.argtypes = [ctypes.c_char_p]
    H.qkh_source_next.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
    H.qkh_source_kind.restype = ctypes.c_char_p
    H.qkh_source_kind.argtypes = [ctypes.c_void_p]
    H.qkh_source_close.argtypes = [ctypes.c_void_p]
    s = H.qkh_source_open(os.fsencode(path))
    assert s
    kind = H.qkh_source_kind(s).decode()
    d, n, out = ctypes.c_void_p(), ctypes.c_size_t(), []
    while H.qkh_source_next(s, ctypes.byref(d), ctypes.byref(n)):
        out.append(ctypes.string_at(d.value, n.value))
    if final_kind:
        kind = H.qkh_source_kind(s).decode()
    H.qkh_source_close(s)
    return kind, b"".join(out)


def np_fastq(n, length, seed):
    g = np.random.default_rng(seed)
    seq = np.frombuffer(b"ACGT", np.uint8)[g.integers(0, 4, (n, length))]
    # qualities with plateaus, like real instruments: long matches and dist-1 runs
    q = (33 + np.repeat(g.integers(2, 41, (n, length // 10 + 1)), 10, axis=1)[:, :length]).astype(np.uint8)
    rows = []
    for i in range(n):
        rows.append(b"@read%d some/description\n" % i + seq[i].tobytes() + b"\n+\n" + q[i].tobytes() + b"\n")
    return b"".join(rows)


BIG = {}


def big(name):
    if not BIG:
        g = np.random.default_rng(5)
        BIG["fastq"] = np_fastq(20000, 150, 1)                          # 6.7 MB
        BIG["ragged"] = b"".join(np_fastq(1, int(l), int(l)) for l in g.integers(1, 4000, 1200))
        BIG["zeros"] = bytes(30 << 20)                                   # ratio ~1000: output buffers must grow
        BIG["random"] = g.integers(0, 256, 3 << 20, dtype=np.uint8).tobytes()   # stored blocks only
        BIG["mixed"] = BIG["fastq"][:2 << 20] + BIG["random"][:1 << 20] + BIG["fastq"][2 << 20:4 << 20] + bytes(1 << 20)
    return BIG[name]


@pytest.mark.parametrize("name,level,strategy", [
    ("fastq", 6, zlib.Z_DEFAULT_STRATEGY), ("fastq", 1, zlib.Z_DEFAULT_STRATEGY), ("fastq", 9, zlib.Z_DEFAULT_STRATEGY),
    ("fastq", 6, zlib.Z_FIXED), ("fastq", 6, zlib.Z_HUFFMAN_ONLY), ("ragged", 6, zlib.Z_DEFAULT_STRATEGY),
    ("zeros", 6, zlib.Z_DEFAULT_STRATEGY), ("random", 6, zlib.Z_DEFAULT_STRATEGY), ("mixed", 6, zlib.Z_DEFAULT_STRATEGY),
    ("mixed", 0, zlib.Z_DEFAULT_STRATEGY)])
def test_parallel_gzip_delivers_exactly_the_stream(tmp_path, monkeypatch, capfd, name, level, strategy):
    data = big(name)
    p = tmp_path / "x.gz"
    p.write_bytes(gz(data, level, strategy))
    for threads, chunk_kb in (("3", "16"), ("8", "4"), ("2", "300")):
        if p.stat().st_size < 2 * int(chunk_kb) * 1024:
            continue
        monkeypatch.setenv("QUACK_THREADS", threads)
        monkeypatch.setenv("QUACK_PGZIP_CHUNK_KB", chunk_kb)
        monkeypatch.setenv("QUACK_VERBOSE", "1")
        kind, got = source_bytes(str(p))
        assert kind == "pgzip x" + threads
        assert got == data, (name, level, threads, chunk_kb)
        err = capfd.readouterr().err
        kept, redone = [int(t) for t in err.replace(",", " ").split() if t.isdigit()][-2:]
        if name in ("fastq", "ragged") and strategy == zlib.Z_DEFAULT_STRATEGY:
            # the speculation really is what ran (slices smaller than a DEFLATE
            # block often hold no block start at all and are redone, empty)
            assert kept >= (2 * redone if chunk_kb == "300" else 10), err
        if strategy == zlib.Z_FIXED or name == "random":
            assert kept == 0, err                  # nothing to recognise: all in order, still right


def test_parallel_gzip_members_garbage_and_damage(tmp_path, monkeypatch):
    """whatever the file, the bytes delivered are those of the one-thread decoder"""
    monkeypatch.setenv("QUACK_THREADS", "4")
    monkeypatch.setenv("QUACK_PGZIP_CHUNK_KB", "8")
    fq = big("fastq")
    p = tmp_path / "x.gz"

    def both():
        kind, par = source_bytes(str(p))
        monkeypatch.setenv("QUACK_NO_PGZIP", "1")
        kind1, ser = source_bytes(str(p))
        monkeypatch.delenv("QUACK_NO_PGZIP")
        assert kind1 == "inflate_fast" and (kind == "pgzip x4" or p.stat().st_size < 16384)
        return par, ser

    # many members of very different sizes (members shorter than the window too)
    g = np.random.default_rng(9)
    cuts = sorted(set([0, len(fq)] + [int(c) for c in g.integers(0, len(fq), 40)] + [100000 + i * 900 for i in range(30)]))
    members = [gz(fq[a:b], int(g.integers(1, 10))) for a, b in zip(cuts, cuts[1:])]
    p.write_bytes(b"".join(members))
    par, ser = both()
    assert par == fq and ser == fq
    # trailing garbage after the last member is ignored
    p.write_bytes(b"".join(members) + b"\0" * 5000 + b"junk")
    par, ser = both()
    assert par == fq and ser == fq
    # truncated and damaged files: a prefix of the truth, the same prefix
    whole = gz(fq, 6)
    for cut in (len(whole) // 3, len(whole) - 5, 9000):
        p.write_bytes(whole[:cut])
        par, ser = both()
        assert par == ser and fq.startswith(par) and (cut < 10000 or len(par) > 0)
    r = random.Random(4)
    for _ in range(25):
        c = bytearray(whole)
        for _ in range(r.randrange(1, 4)):
            c[r.randrange(20, len(c))] ^= 1 << r.randrange(8)
        p.write_bytes(bytes(c))
        par, ser = both()
        assert par == ser
    # a wrong ISIZE or CRC-32 in the middle of the file ends the stream where zlib's gzread ends it for the
    # reference (16 KiB reads: the call that meets the bad trailer returns nothing, see source.c)
    for first, field in ((3 << 20, -2), ((3 << 20) + 5000, -2), ((3 << 20) + 5000, -7), (700, -6)):
        m1, m2 = gz(fq[:first], 6), gz(fq[first:], 6)
        bad = bytearray(m1)
        bad[field] ^= 0x10
        p.write_bytes(bytes(bad) + m2)
        par, ser = both()
        monkeypatch.setenv("QUACK_ZLIB", "1")
        kind, ref = source_bytes(str(p))
        monkeypatch.delenv("QUACK_ZLIB")
        assert kind == "zlib" and par == ser == ref
        assert fq.startswith(ref) and len(ref) % 16384 == 0 and first - 32768 < len(ref) <= first


def test_parallel_gzip_through_the_tokenizer(tmp_path, monkeypatch):
    monkeypatch.setenv("QUACK_THREADS", "5")
    monkeypatch.setenv("QUACK_PGZIP_CHUNK_KB", "64")
    text = big("fastq")
    p = tmp_path / "x.fq.gz"
    p.write_bytes(gz(text, 6))
    a, _ = product_tokenize(str(p), 1 << 22, 1 << 16)
    want, _ = product_tokenize_text(tmp_path, text)
    assert len(a) == 20000 and a == want


def test_crc32_fold_is_zlibs_crc32():
    """the carry-less-multiply CRC-32 of the host feed (crc32_fold.c) against zlib on random lengths,
    alignments and start values"""
    H = _capi.host()
    H.qkh_crc32.restype = ctypes.c_uint32
    H.qkh_crc32.argtypes = [ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t]
    rng = random.Random(11)
    blob = bytes(rng.getrandbits(8) for _ in range(70000))
    for _ in range(3000):
        a = rng.randrange(0, 100)
        n = rng.choice([rng.randrange(0, 300), rng.randrange(0, 5000), rng.randrange(0, len(blob) - a)])
        start = rng.choice([0, rng.getrandbits(32)])
        assert H.qkh_crc32(start, blob[a:a + n], n) == zlib.crc32(blob[a:a + n], start)


def test_a_slice_that_outgrows_the_parallel_decoder_is_streamed_by_the_serial_one(tmp_path, monkeypatch):
    """the in-order fallback of the multi-threaded decoder had no output bound (1 MiB of zeros
    inflates to 1 GiB in one slot).  Past the bound the decoder's state is handed to the one-thread ring
    producer; the bytes are the same (bound lowered to 1 MiB here, the file has a 24 MiB run in one block)"""
    monkeypatch.setenv("QUACK_THREADS", "4")
    monkeypatch.setenv("QUACK_PGZIP_CHUNK_KB", "4")
    monkeypatch.setenv("QUACK_PGZIP_MAX_SLICE_MB", "1")
    fq = big("fastq")[:2 << 20]
    text = fq[:1 << 20] + b"@long\n" + b"A" * (24 << 20) + b"\n+\n" + b"I" * (24 << 20) + b"\n" + fq[1 << 20:]
    p = tmp_path / "run.fq.gz"
    p.write_bytes(gz(text[:30 << 20], 9) + gz(text[30 << 20:], 6))   # (two members: the CRC chain crosses the hand-over)
    kind, par = source_bytes(str(p), final_kind=True)
    assert kind == "pgzip -> inflate_fast"          # the hand-over happened
    monkeypatch.setenv("QUACK_NO_PGZIP", "1")
    kind1, ser = source_bytes(str(p))
    assert kind1 == "inflate_fast"
    assert par == ser == text
    # ... and a damaged trailer behind the hand-over still ends the stream where zlib ends it
    monkeypatch.delenv("QUACK_NO_PGZIP")
    blob = bytearray(p.read_bytes())
    blob[-6] ^= 0x08
    p.write_bytes(bytes(blob))
    _, par = source_bytes(str(p))
    monkeypatch.setenv("QUACK_ZLIB", "1")
    _, ref = source_bytes(str(p))
    assert par == ref and len(ref) < len(text)


def test_long_runs_of_empty_members_do_not_end_the_stream(tmp_path, monkeypatch):
    """zlib's gzread — the reference's reader — walks through any number of empty members; the decoders log at
    most 64 trailers per call, and a call that only met trailers returns 0 without being the end (ADVICE round 2:
    the callers took that 0 for the end of the stream)"""
    a, b, c = fastq(300, 150), fastq(200, 100), fastq(50, 75)
    empty = gz(b"")
    for run in (63, 64, 65, 200):
        p = tmp_path / ("e%d.fq.gz" % run)
        p.write_bytes(empty * run + gz(a) + empty * run + gz(b) + empty * (2 * run + 1) + gz(c) + empty * run)
        want = a + b + c
        assert gzip.open(str(p)).read() == want
        for env in ({"QUACK_NO_PGZIP": "1"}, {"QUACK_THREADS": "4", "QUACK_PGZIP_CHUNK_KB": "8"}, {"QUACK_ZLIB": "1"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            kind, got = source_bytes(str(p))
            for k in env:
                monkeypatch.delenv(k)
            assert got == want, (run, kind, len(got), len(want))
        # the same as BGZF blocks (htslib writes an empty block at the end of every file: `cat` of many small files)
        pb = tmp_path / ("e%d.bgzf.gz" % run)
        eof_block = bgzf(b"", eof=False) if False else bgzf(b"")   # one empty block
        pb.write_bytes(eof_block * run + bgzf(a, eof=False) + eof_block * (run + 3) + bgzf(b, block=1000) + eof_block * run)
        monkeypatch.setenv("QUACK_THREADS", "3")
        kind, got = source_bytes(str(pb))
        monkeypatch.delenv("QUACK_THREADS")
        assert kind.startswith("bgzf") and got == a + b, (run, kind, len(got))

"""ctypes binding of oracle/_build/liboracle.so — the CHECKER used by tests.
(Test infrastructure: nothing under quack_amd/ imports this.)"""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))

ROWS = 97
KMER_TABLE = 1 << 20


class _Table(ctypes.Structure):
    _fields_ = [("bases", ctypes.POINTER(ctypes.c_uint64)), ("max_length", ctypes.c_uint64),
                ("number_of_sequences", ctypes.c_uint64), ("capacity", ctypes.c_uint64)]


_lib.oracle_read_fastq.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.POINTER(_Table)]
_lib.oracle_read_adapters.argtypes = [ctypes.c_char_p, ctypes.c_void_p]
_lib.oracle_adapter_insert.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
_lib.oracle_adapter_insert.restype = None
_lib.oracle_accumulate_batch.argtypes = [ctypes.POINTER(_Table), ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p]
_lib.oracle_base_code.argtypes = [ctypes.c_ubyte]
_lib.oracle_qual_bin.argtypes = [ctypes.c_ubyte]
_lib.oracle_reader_open.restype = ctypes.c_void_p
_lib.oracle_reader_open.argtypes = [ctypes.c_char_p]
_lib.oracle_reader_next.restype = ctypes.c_long
_lib.oracle_reader_next.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p),
                                    ctypes.POINTER(ctypes.c_void_p)]
_lib.oracle_reader_close.argtypes = [ctypes.c_void_p]
_lib.oracle_table_init.argtypes = [ctypes.POINTER(_Table)]
_lib.oracle_table_free.argtypes = [ctypes.POINTER(_Table)]


def _take(t):
    n = t.max_length * ROWS
    arr = np.ctypeslib.as_array(t.bases, shape=(n,)).copy() if n else np.zeros(0, np.uint64)
    res = (arr.reshape(-1, ROWS), int(t.number_of_sequences))
    _lib.oracle_table_free(ctypes.byref(t))
    return res


def kmers_from_file(path):
    k = np.zeros(KMER_TABLE, dtype=np.uint8)
    assert _lib.oracle_read_adapters(os.fsencode(path), k.ctypes.data) == 0
    return k


def kmers_from_seqs(seqs):
    k = np.zeros(KMER_TABLE, dtype=np.uint8)
    for s in seqs:
        b = s if isinstance(s, bytes) else s.encode()
        _lib.oracle_adapter_insert(k.ctypes.data, b, len(b))
    return k


def kmers_to_bitset(k):
    """byte table (oracle) -> uint32[32768] bitset (C-ABI format)"""
    return np.packbits(k.astype(np.uint8), bitorder="little").view(np.uint32).copy()


def read_fastq(path, kmers=None):
    t = _Table()
    _lib.oracle_table_init(ctypes.byref(t))
    rc = _lib.oracle_read_fastq(os.fsencode(path), kmers.ctypes.data if kmers is not None else None,
                                ctypes.byref(t))
    assert rc == 0, "oracle cannot read %s" % path
    return _take(t)


def accumulate_batch(seq, qual, offsets=None, read_len=0, kmers=None):
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    qual = np.ascontiguousarray(qual, dtype=np.uint8)
    t = _Table()
    _lib.oracle_table_init(ctypes.byref(t))
    if offsets is not None:
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        op = offsets.ctypes.data
    else:
        n = len(seq) // read_len if read_len else 0
        op = None
    rc = _lib.oracle_accumulate_batch(ctypes.byref(t), seq.ctypes.data, qual.ctypes.data, op, n, read_len,
                                      kmers.ctypes.data if kmers is not None else None)
    assert rc == 0
    return _take(t)


def accumulate_batch_threads(seq, qual, offsets=None, read_len=0, kmers=None, threads=None):
    """accumulate_batch with the reads cut into contiguous shares, one oracle table per thread (ctypes releases
    the GIL), tables summed: the same integers (every update is a commutative ++, quack.c:202-220), in a
    fraction of the time on a many-core host — what lets the full BASELINE sizes be checked exactly"""
    import os
    import threading
    T = threads or max(1, min(32, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    n = (len(offsets) - 1) if offsets is not None else (len(seq) // read_len if read_len else 0)
    T = max(1, min(T, n))
    cuts = [n * i // T for i in range(T + 1)]
    parts = [None] * T

    def work(i):
        lo, hi = cuts[i], cuts[i + 1]
        if offsets is None:
            parts[i] = accumulate_batch(seq[lo * read_len:hi * read_len], qual[lo * read_len:hi * read_len], read_len=read_len, kmers=kmers)
        else:
            a, e = int(offsets[lo]), int(offsets[hi])
            parts[i] = accumulate_batch(seq[a:e], qual[a:e], offsets[lo:hi + 1] - offsets[lo], kmers=kmers)

    th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    ml = max(p[0].shape[0] for p in parts)
    total = np.zeros((ml, ROWS), dtype=np.uint64)
    for b, _ in parts:
        total[:b.shape[0]] += b
    return total, sum(p[1] for p in parts)


def base_code(c):
    return _lib.oracle_base_code(c)


def qual_bin(b):
    return _lib.oracle_qual_bin(b)


def tokenize(path):
    """[(seq_bytes, qual_bytes_or_None)], final_status  via the oracle tokenizer"""
    r = _lib.oracle_reader_open(os.fsencode(path))
    assert r
    out = []
    s, q = ctypes.c_void_p(), ctypes.c_void_p()
    while True:
        l = _lib.oracle_reader_next(r, ctypes.byref(s), ctypes.byref(q))
        if l < 0:
            break
        out.append((ctypes.string_at(s, l), ctypes.string_at(q, l) if q else None))
    _lib.oracle_reader_close(r)
    return out, l

"""C-ABI error convention (include/quack_hip.h): misuse returns a negative code
and a message, never crashes, and leaves the accumulator usable."""
import ctypes

import numpy as np
import pytest

import oracle_binding as ob
import synth
import quack_amd
from quack_amd import _capi

pytestmark = pytest.mark.gpu
L = _capi.hip()


def err():
    return L.qk_last_error().decode()


def test_bad_arguments_and_call_order():
    h = ctypes.c_void_p()
    assert L.qk_accum_create(ctypes.byref(h), 99, None, 0) == -1 and "out of range" in err()
    assert L.qk_accum_create(None, 0, None, 0) == -1
    assert L.qk_accum_create(ctypes.byref(h), 0, None, 0) == 0
    try:
        assert L.qk_accum_commit(h, 1, 8, 0, 8) == -6 and "no batch acquired" in err()      # QK_ESTATE
        s, q = ctypes.POINTER(ctypes.c_uint8)(), ctypes.POINTER(ctypes.c_uint8)()
        o = ctypes.POINTER(ctypes.c_uint64)()
        cb, cr = ctypes.c_uint64(), ctypes.c_uint64()
        assert L.qk_accum_acquire(h, ctypes.byref(s), ctypes.byref(q), ctypes.byref(o), ctypes.byref(cb), ctypes.byref(cr)) == 0
        assert L.qk_accum_acquire(h, ctypes.byref(s), ctypes.byref(q), ctypes.byref(o), ctypes.byref(cb), ctypes.byref(cr)) == -6
        assert L.qk_accum_commit(h, 1, cb.value + 1, 0, 8) == -1 and "capacity" in err()
        # the slot is still held after a rejected commit: a correct one goes through
        o[0], o[1], o[2] = 0, 5, 4                                                             # not monotonic
        assert L.qk_accum_commit(h, 2, 4, 1, 0) == -1 and "monotonic" in err()
        seq = np.frombuffer(b"ACGTACGTAC", np.uint8)
        ctypes.memmove(s, seq.ctypes.data, 10)
        ctypes.memmove(q, np.full(10, 70, np.uint8).ctypes.data, 10)
        o[0], o[1], o[2] = 0, 4, 10
        assert L.qk_accum_commit(h, 2, 10, 1, 0) == 0
        ml, nr = ctypes.c_uint64(), ctypes.c_uint64()
        out = np.zeros(3 * 97, np.uint64)
        assert L.qk_accum_finish(h, out.ctypes.data, 3, ctypes.byref(ml), ctypes.byref(nr)) == -1 and "positions" in err()
        out = np.zeros(6 * 97, np.uint64)
        assert L.qk_accum_finish(h, out.ctypes.data, 6, ctypes.byref(ml), ctypes.byref(nr)) == 0
        assert (ml.value, nr.value) == (6, 2)
        want, _ = ob.accumulate_batch(seq, np.full(10, 70, np.uint8), np.array([0, 4, 10], np.uint64))
        np.testing.assert_array_equal(out.reshape(6, 97), want)
        assert L.qk_accum_import_table(h, out.ctypes.data, 1 << 20, None) == -1 and "reserve" in err()
        assert L.qk_accum_configure(h, 333, 3, 0, 0) == 0
        assert L.qk_accum_submit_fixed(h, seq.ctypes.data, seq.ctypes.data, 5, 2) == -1 and "unsupported threads" in err()
    finally:
        L.qk_accum_destroy(h)
    L.qk_accum_destroy(None)                                                                  # no-op


def test_python_mirror_validates_batches():
    with quack_amd.Accumulator(0) as acc:
        with pytest.raises(ValueError):
            acc.submit(np.zeros(10, np.uint8), np.zeros(9, np.uint8), np.array([0, 10], np.uint64))
        with pytest.raises(ValueError):
            acc.submit_fixed(np.zeros(10, np.uint8), np.zeros(10, np.uint8), 3)
        with pytest.raises(quack_amd.HipUnavailable):
            acc.submit(np.zeros(10, np.uint8), np.zeros(10, np.uint8), np.array([0, 7, 5, 10], np.uint64))
        seq, qual, off = synth.ragged(500, 1, 50, seed=3)
        acc.submit(seq, qual, off)                       # still usable afterwards
        sd = acc.finish()
    want = ob.accumulate_batch(seq, qual, off)
    assert sd.number_of_sequences == want[1] and np.array_equal(sd.bases, want[0])


def test_read_longer_than_a_batch_slot_grows_the_slots(monkeypatch, tmp_path):
    """(round 1 reported it as an error; a batch holds whole reads, so the slots now grow:
    qk_accum_resize_slots, tests/test_gpu_parity.py has the oracle comparison at small sizes)"""
    monkeypatch.setenv("QUACK_HIP_BATCH_MB", "1")
    p = tmp_path / "huge.fq"
    n = (1 << 20) + 100
    p.write_bytes(b"@r\n" + b"A" * n + b"\n+\n" + b"I" * n + b"\n")
    sd = quack_amd.read_fastq(str(p))
    assert sd.max_length == n and sd.number_of_sequences == 1 and sd.bases[:, 91].sum() == n

"""Python mirror of the reference's function-level interface, on the C-ABI."""
import ctypes
import os
import tempfile

import numpy as np

from . import _capi
from ._capi import QK_BATCH_ALIGNED128, QK_BATCH_NEUTRAL_PADS, QK_KMER_TABLE_WORDS, QK_N_ROWS, QK_TAIL_SLACK


class HipUnavailable(RuntimeError):
    """The HIP path could not run (no device, failed call).  Never caught and
    replaced by a CPU computation anywhere in this package."""


def _check(rc, lib=None):
    if rc != 0:
        raise HipUnavailable("quack_hip error %d: %s" % (rc, (lib or _capi.hip()).qk_last_error().decode()))


def device_count():
    n = ctypes.c_int(0)
    rc = _capi.hip().qk_device_count(ctypes.byref(n))
    return n.value if rc == 0 else 0


class SequenceData:
    """sequence_data of quack.c:141-146: bases[max_length] of 97 u64 counters
    (91 scores, content A/T/C/G, length_count, kmer_count)."""

    def __init__(self, bases, number_of_sequences):
        self.bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, QK_N_ROWS)
        self.max_length = int(self.bases.shape[0])
        self.number_of_sequences = int(number_of_sequences)

    @property
    def scores(self):
        return self.bases[:, :91]

    @property
    def content(self):
        return self.bases[:, 91:95]

    @property
    def length_count(self):
        return self.bases[:, 95]

    @property
    def kmer_count(self):
        return self.bases[:, 96]


def read_adapters(adapters_file):
    """quack.c:154-178 -> 2^20-bit table as uint32[32768] (bit i <=> kmers[i])."""
    bits = np.zeros(QK_KMER_TABLE_WORDS, dtype=np.uint32)
    rc = _capi.host().qkh_read_adapters(os.fsencode(adapters_file), bits.ctypes.data)
    if rc != 0:
        raise OSError("cannot read adapters file %s" % adapters_file)
    return bits


def read_fastq(fastq_file, kmers=None, devices=(0,)):
    """quack.c:180-228 on the GPU(s): tokenise on the host, accumulate in HIP."""
    H = _capi.host()
    devs = (ctypes.c_int * len(devices))(*devices)
    out = ctypes.c_void_p()
    max_len = ctypes.c_uint64()
    n_reads = ctypes.c_uint64()
    kp = kmers.ctypes.data if kmers is not None else None
    rc = H.qkh_accumulate_file(os.fsencode(fastq_file), kp, devs, len(devices),
                               ctypes.byref(out), ctypes.byref(max_len), ctypes.byref(n_reads))
    if rc != 0:
        raise HipUnavailable(H.qkh_last_error().decode())
    try:
        n = max_len.value * QK_N_ROWS
        arr = np.ctypeslib.as_array(ctypes.cast(out, ctypes.POINTER(ctypes.c_uint64)), shape=(n,)).copy() \
            if n else np.zeros(0, dtype=np.uint64)
    finally:
        _capi.libc().free(out)
    return SequenceData(arr, n_reads.value)


def render_svg(forward, reverse=None, name=None, adapters=False):
    """transform() + draw() + the <svg> envelope (quack.c:230-293, 295-856,
    879-925) -> (svg_bytes, stderr_bytes).  Inputs are not modified."""
    H = _capi.host()
    C = _capi.libc()
    f = forward.bases.copy()
    r = reverse.bases.copy() if reverse is not None else None
    with tempfile.TemporaryDirectory() as d:
        po, pe = os.path.join(d, "o"), os.path.join(d, "e")
        fo, fe = C.fopen(po.encode(), b"wb"), C.fopen(pe.encode(), b"wb")
        try:
            rc = H.qkh_render_document(
                fo, fe, name.encode() if name is not None else None, 1 if adapters else 0,
                f.ctypes.data, forward.max_length, forward.number_of_sequences,
                r.ctypes.data if r is not None else None,
                reverse.max_length if reverse is not None else 0,
                reverse.number_of_sequences if reverse is not None else 0)
        finally:
            C.fclose(fo)
            C.fclose(fe)
        if rc != 0:
            raise ValueError("nothing to draw (max_length == 0)")
        with open(po, "rb") as a, open(pe, "rb") as b:
            return a.read(), b.read()


class Accumulator:
    """One qk_accum (include/quack_hip.h): the state of a read_fastq() call on
    one GPU.  Host batches are numpy arrays; device batches are anything with
    a data_ptr() (torch tensors on the accumulator's device)."""

    def __init__(self, device=0, kmers=None, max_len_hint=0, _lib=None, experiment=None):
        """experiment: use the -DQK_EXPERIMENT build (libquack_hip_exp.so: QUACK_HIP_TUNE, configure()); by default that is the
        case exactly when QUACK_HIP_TUNE is set — the product library reads no such switch"""
        self._h = ctypes.c_void_p()
        if experiment is None:
            experiment = bool(os.environ.get("QUACK_HIP_TUNE"))
        # (_lib: a second build bound with _capi.bind_hip, developer tools only)
        self._L = _lib if _lib is not None else (_capi.hip_exp() if experiment else _capi.hip())
        kp = kmers.ctypes.data if kmers is not None else None
        self._kmers = kmers  # keep alive during create
        self._ck(self._L.qk_accum_create(ctypes.byref(self._h), device, kp, max_len_hint))
        self.device = device

    def _ck(self, rc):
        _check(rc, self._L)   # (the error text lives in the library that returned rc)

    def close(self):
        if self._h:
            self._L.qk_accum_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def configure(self, threads=0, unroll=0, tile=0, wgs_per_cu=0):
        self._ck(self._L.qk_accum_configure(self._h, threads, unroll, tile, wgs_per_cu))

    # -- host-resident batches (copied through the pinned double buffer) ----
    def submit(self, seq, qual, offsets):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        qual = np.ascontiguousarray(qual, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        if n < 0 or len(seq) != len(qual) or (n >= 0 and int(offsets[-1]) != len(seq)):
            raise ValueError("inconsistent batch arrays")
        self._ck(self._L.qk_accum_submit(self._h, seq.ctypes.data, qual.ctypes.data, offsets.ctypes.data, n))

    def submit_fixed(self, seq, qual, read_len):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        qual = np.ascontiguousarray(qual, dtype=np.uint8)
        if read_len <= 0 or len(seq) % read_len or len(seq) != len(qual):
            raise ValueError("inconsistent fixed-length batch")
        self._ck(self._L.qk_accum_submit_fixed(self._h, seq.ctypes.data, qual.ctypes.data, read_len,
                                             len(seq) // read_len))

    def submit_strided(self, seq, qual, lengths, stride):
        """fixed stride, own lengths (qk_accum_submit_strided): read r is seq[r*stride : r*stride + lengths[r]]"""
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        qual = np.ascontiguousarray(qual, dtype=np.uint8)
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
        if stride <= 0 or len(seq) != len(lengths) * stride or len(seq) != len(qual):
            raise ValueError("inconsistent strided batch")
        self._ck(self._L.qk_accum_submit_strided(self._h, seq.ctypes.data, qual.ctypes.data, lengths.ctypes.data,
                                               stride, len(lengths)))

    def submit_gapped(self, seq, qual, starts, lengths, aligned=False):
        """one gapped batch through a pinned slot (qk_accum_acquire / slot_lengths /
        commit_gapped): read r is seq[starts[r] : starts[r] + lengths[r]]"""
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        qual = np.ascontiguousarray(qual, dtype=np.uint8)
        starts = np.ascontiguousarray(starts, dtype=np.uint64)
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
        n, extent = len(starts), len(seq)
        hs, hq = ctypes.POINTER(ctypes.c_uint8)(), ctypes.POINTER(ctypes.c_uint8)()
        ho, hl = ctypes.POINTER(ctypes.c_uint64)(), ctypes.POINTER(ctypes.c_uint32)()
        capb, capr = ctypes.c_uint64(), ctypes.c_uint64()
        self._ck(self._L.qk_accum_acquire(self._h, ctypes.byref(hs), ctypes.byref(hq), ctypes.byref(ho),
                                        ctypes.byref(capb), ctypes.byref(capr)))
        if extent > capb.value or n > capr.value:
            self._L.qk_accum_commit(self._h, 0, 0, 0, 0)
            raise ValueError("batch larger than a pinned slot")
        self._ck(self._L.qk_accum_slot_lengths(self._h, ctypes.byref(hl)))
        ctypes.memmove(hs, seq.ctypes.data, extent)
        ctypes.memmove(hq, qual.ctypes.data, extent)
        ctypes.memmove(ho, starts.ctypes.data, 8 * n)
        ctypes.memmove(hl, lengths.ctypes.data, 4 * n)
        rc = self._L.qk_accum_commit_gapped(self._h, n, extent, QK_BATCH_ALIGNED128 if aligned else 0)
        if rc:
            self._L.qk_accum_commit(self._h, 0, 0, 0, 0)   # give the slot back
            self._ck(rc)

    # -- device-resident batches (bench / torch plumbing) --------------------
    def submit_device_gapped(self, d_seq, d_qual, d_starts, d_lengths, n_reads, extent_bytes, max_len,
                             aligned=False, stream=None):
        """starts: int64/uint64[n] device tensor, lengths: int32/uint32[n]; see
        qk_accum_submit_device_gapped"""
        self._ck(self._L.qk_accum_submit_device_gapped(
            self._h, d_seq.data_ptr(), d_qual.data_ptr(), d_starts.data_ptr(), d_lengths.data_ptr(),
            n_reads, extent_bytes, max_len, QK_BATCH_ALIGNED128 if aligned else 0, stream))

    def submit_device_strided(self, d_seq, d_qual, d_lengths, n_reads, stride, max_len, stream=None, neutral_pads=False):
        """read r at [r*stride, r*stride + lengths[r]); lengths: int32/uint32[n] device tensor; stride % 4 == 0;
        neutral_pads: the caller promises 0xFF behind every read's last base (QK_BATCH_NEUTRAL_PADS: the kernel without tail
        masks); see qk_accum_submit_device_strided"""
        self._ck(self._L.qk_accum_submit_device_strided_flags(
            self._h, d_seq.data_ptr(), d_qual.data_ptr(), d_lengths.data_ptr() if d_lengths is not None else None,
            n_reads, stride, max_len, QK_BATCH_NEUTRAL_PADS if neutral_pads else 0, stream))

    def submit_device_padded(self, d_seq, d_qual, n_reads, read_len, stride, stream=None):
        """fixed-length reads of read_len bases, read r at r*stride (stride % 4 == 0, >= read_len): the layout the host
        feed gives uniform reads whose length is not a multiple of 4 (qk_accum_submit_device_strided, lengths NULL)"""
        self.submit_device_strided(d_seq, d_qual, None, n_reads, stride, read_len, stream)

    def padded_stride(self, read_len):
        """the stride the library wants for uniform reads of read_len on this accumulator (0: packed)"""
        s = ctypes.c_uint32()
        self._ck(self._L.qk_accum_padded_stride(self._h, read_len, ctypes.byref(s)))
        return s.value

    def commit_padded(self, n_reads, read_len, stride):
        """enqueue the acquired slot as a padded fixed-length batch (read r written at r*stride)"""
        self._ck(self._L.qk_accum_commit_padded(self._h, n_reads, read_len, stride))

    def submit_device(self, d_seq, d_qual, d_offsets, n_reads, total_bytes, max_len, stream=None):
        """d_* expose data_ptr(); buffers need QK_TAIL_SLACK readable bytes
        after total_bytes.  Enqueues only."""
        self._ck(self._L.qk_accum_submit_device(
            self._h, d_seq.data_ptr(), d_qual.data_ptr(),
            d_offsets.data_ptr() if d_offsets is not None else None,
            n_reads, total_bytes, max_len, stream))

    def sync(self):
        self._ck(self._L.qk_accum_sync(self._h))

    def stats(self):
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        self._ck(self._L.qk_accum_stats(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def table_words(self):
        n = ctypes.c_uint64()
        self._ck(self._L.qk_accum_table_words(self._h, ctypes.byref(n)))
        return n.value

    def reserve(self, max_len):
        self._ck(self._L.qk_accum_reserve(self._h, max_len))

    def export_table(self, d_dst, stream=None):
        self._ck(self._L.qk_accum_export_table(self._h, d_dst.data_ptr(), stream))

    def import_table(self, d_src, max_len, stream=None):
        self._ck(self._L.qk_accum_import_table(self._h, d_src.data_ptr(), max_len, stream))

    def timing(self, on=True):
        """on: False / True, or N > 1 = HIP events around every Nth batch only"""
        self._ck(self._L.qk_accum_timing_enable(self._h, int(on)))

    def timing_read(self):
        ms, n = ctypes.c_double(), ctypes.c_uint64()
        self._ck(self._L.qk_accum_timing_read(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def timing_read_batch(self):
        """(histogram-kernel ms, all-kernels-of-the-batch ms, launches) since timing(True)"""
        ms, bms, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_uint64()
        self._ck(self._L.qk_accum_timing_read_batch(self._h, ctypes.byref(ms), ctypes.byref(bms), ctypes.byref(n)))
        return ms.value, bms.value, n.value

    def timing_read_range(self):
        """(shortest, longest) timed histogram-kernel launch in ms"""
        lo, hi = ctypes.c_double(), ctypes.c_double()
        self._ck(self._L.qk_accum_timing_read_range(self._h, ctypes.byref(lo), ctypes.byref(hi)))
        return lo.value, hi.value

    # -- the pinned double buffer itself (qk_accum_acquire / qk_accum_commit): what the C host feed drives
    def acquire(self):
        """-> (seq, qual, offsets) numpy views of the next pinned batch slot (capacity: len(seq) bytes,
        len(offsets) - 1 reads); valid until commit()"""
        hs, hq = ctypes.POINTER(ctypes.c_uint8)(), ctypes.POINTER(ctypes.c_uint8)()
        ho = ctypes.POINTER(ctypes.c_uint64)()
        capb, capr = ctypes.c_uint64(), ctypes.c_uint64()
        self._ck(self._L.qk_accum_acquire(self._h, ctypes.byref(hs), ctypes.byref(hq), ctypes.byref(ho),
                                        ctypes.byref(capb), ctypes.byref(capr)))
        return (np.ctypeslib.as_array(hs, shape=(capb.value,)), np.ctypeslib.as_array(hq, shape=(capb.value,)),
                np.ctypeslib.as_array(ho, shape=(capr.value + 1,)))

    def commit(self, n_reads, total_bytes, read_len=0):
        """enqueue H2D + kernels of the acquired slot; read_len > 0: fixed-length batch (offsets unused)"""
        self._ck(self._L.qk_accum_commit(self._h, n_reads, total_bytes, 0 if read_len else 1, read_len))

    def finish(self):
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        self._ck(self._L.qk_accum_finish(self._h, None, 0, ctypes.byref(a), ctypes.byref(b)))
        out = np.zeros(a.value * QK_N_ROWS, dtype=np.uint64)
        if a.value:
            self._ck(self._L.qk_accum_finish(self._h, out.ctypes.data, a.value, ctypes.byref(a), ctypes.byref(b)))
        return SequenceData(out, b.value)


def allreduce(accumulators):
    """qk_accum_allreduce: sum the tables of several accumulators in one process
    (same-device shards by an add kernel, distinct devices by one RCCL
    all-reduce); afterwards each holds the global table."""
    arr = (ctypes.c_void_p * len(accumulators))(*[a._h for a in accumulators])
    L = accumulators[0]._L if accumulators else _capi.hip()   # (accumulators of one library)
    _check(L.qk_accum_allreduce(arr, len(accumulators)), L)


def pad_for_device(arr):
    """numpy uint8 array + the tail slack the kernels may read past the end."""
    out = np.zeros(len(arr) + QK_TAIL_SLACK, dtype=np.uint8)
    out[:len(arr)] = arr
    return out

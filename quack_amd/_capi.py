"""ctypes bindings of the two native libraries.

libquack_hip.so   include/quack_hip.h  — the C-ABI of the HIP accumulation path
libquack_host.so  quack_amd/host/      — tokenizer, adapters, transform, draw, CLI

There is no Python or CPU implementation of the accumulation path behind
these bindings: if the libraries are missing, importing fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))

QK_N_ROWS = 97
QK_N_SCORES = 91
QK_ROW_CONTENT = 91
QK_ROW_LENGTH = 95
QK_ROW_KMER = 96
QK_KMER_TABLE_WORDS = (1 << 20) // 32
QK_TAIL_SLACK = 16
QK_BATCH_ALIGNED128 = 1
QK_BATCH_NEUTRAL_PADS = 2
QK_ENODEV = -2


class NativeLibraryMissing(RuntimeError):
    pass


def _load(name):
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        raise NativeLibraryMissing(
            "%s not built: run `make` at the repository root (or __graft_entry__.build())" % path)
    return ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)


_hip = None
_host = None

c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_u32p = ctypes.POINTER(ctypes.c_uint32)
c_u64p = ctypes.POINTER(ctypes.c_uint64)
c_vp = ctypes.c_void_p


def hip():
    """libquack_hip.so with argtypes set (every symbol of include/quack_hip.h)."""
    global _hip
    if _hip is not None:
        return _hip
    _hip = bind_hip(_load("libquack_hip.so"))
    return _hip


_hip_exp = None


def hip_exp():
    """libquack_hip_exp.so — the -DQK_EXPERIMENT build of the same source (QUACK_HIP_TUNE switches, kernel variants of launch
    geometries the planner does not pick) — loaded beside the product's library with its own symbols first.  Tools and the
    parity tests that cross-check those geometries only; nothing of the product links it."""
    global _hip_exp
    if _hip_exp is None:
        # the product's library first: it is loaded RTLD_GLOBAL, and with it the HIP runtime it links — a process whose FIRST HIP user is a
        # library loaded RTLD_LOCAL ends up with two runtimes once torch brings its own copy ("No HIP GPUs are available" from torch)
        hip()
        path = os.path.join(_HERE, "libquack_hip_exp.so")
        if not os.path.exists(path):
            raise NativeLibraryMissing("%s not built: run `make exp` at the repository root" % path)
        _hip_exp = bind_hip(ctypes.CDLL(path, mode=os.RTLD_LOCAL | os.RTLD_DEEPBIND))
    return _hip_exp


def bind_hip(L):
    """argtypes of every symbol of include/quack_hip.h on a loaded library (hip() does this for the product's own;
    tools/ab_inproc.py binds a second build beside it)"""
    L.qk_last_error.restype = ctypes.c_char_p
    L.qk_version.restype = ctypes.c_char_p
    L.qk_device_count.argtypes = [ctypes.POINTER(ctypes.c_int)]
    L.qk_accum_create.argtypes = [ctypes.POINTER(c_vp), ctypes.c_int, c_vp, ctypes.c_uint64]
    L.qk_accum_destroy.argtypes = [c_vp]
    L.qk_accum_destroy.restype = None
    L.qk_accum_acquire.argtypes = [c_vp, ctypes.POINTER(c_u8p), ctypes.POINTER(c_u8p),
                                   ctypes.POINTER(c_u64p), c_u64p, c_u64p]
    L.qk_accum_resize_slots.argtypes = [c_vp, ctypes.c_uint64]
    L.qk_accum_commit.argtypes = [c_vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32]
    L.qk_accum_submit.argtypes = [c_vp, c_vp, c_vp, c_vp, ctypes.c_uint64]
    L.qk_accum_submit_fixed.argtypes = [c_vp, c_vp, c_vp, ctypes.c_uint32, ctypes.c_uint64]
    L.qk_accum_submit_device.argtypes = [c_vp, c_vp, c_vp, c_vp, ctypes.c_uint64, ctypes.c_uint64,
                                         ctypes.c_uint32, c_vp]
    L.qk_accum_submit_device_gapped.argtypes = [c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_uint64, ctypes.c_uint64,
                                                ctypes.c_uint32, ctypes.c_uint32, c_vp]
    L.qk_accum_submit_device_strided.argtypes = [c_vp, c_vp, c_vp, c_vp, ctypes.c_uint64, ctypes.c_uint32,
                                                 ctypes.c_uint32, c_vp]
    L.qk_accum_commit_strided.argtypes = [c_vp, ctypes.c_uint64, ctypes.c_uint32]
    L.qk_accum_submit_device_strided_flags.argtypes = [c_vp, c_vp, c_vp, c_vp, ctypes.c_uint64, ctypes.c_uint32,
                                                       ctypes.c_uint32, ctypes.c_uint32, c_vp]
    L.qk_accum_commit_strided_flags.argtypes = [c_vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32]
    L.qk_accum_padded_stride.argtypes = [c_vp, ctypes.c_uint32, c_u32p]
    L.qk_accum_commit_padded.argtypes = [c_vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32]
    L.qk_accum_submit_strided.argtypes = [c_vp, c_vp, c_vp, c_vp, ctypes.c_uint32, ctypes.c_uint64]
    L.qk_accum_slot_lengths.argtypes = [c_vp, ctypes.POINTER(ctypes.POINTER(ctypes.c_uint32))]
    L.qk_accum_commit_gapped.argtypes = [c_vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32]
    L.qk_accum_sync.argtypes = [c_vp]
    L.qk_accum_stats.argtypes = [c_vp, c_u64p, c_u64p]
    L.qk_accum_table_words.argtypes = [c_vp, c_u64p]
    L.qk_accum_reserve.argtypes = [c_vp, ctypes.c_uint64]
    L.qk_accum_export_table.argtypes = [c_vp, c_vp, c_vp]
    L.qk_accum_import_table.argtypes = [c_vp, c_vp, ctypes.c_uint64, c_vp]
    L.qk_accum_allreduce.argtypes = [ctypes.POINTER(c_vp), ctypes.c_int]
    L.qk_accum_finish.argtypes = [c_vp, c_vp, ctypes.c_uint64, c_u64p, c_u64p]
    L.qk_accum_timing_enable.argtypes = [c_vp, ctypes.c_int]
    L.qk_accum_timing_read.argtypes = [c_vp, ctypes.POINTER(ctypes.c_double), c_u64p]
    L.qk_accum_timing_read_batch.argtypes = [c_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), c_u64p]
    L.qk_accum_timing_read_range.argtypes = [c_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    L.qk_accum_configure.argtypes = [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    return L


def host():
    """libquack_host.so (needs libquack_hip.so next to it)."""
    global _host
    if _host is not None:
        return _host
    hip()
    L = _load("libquack_host.so")
    L.qkh_last_error.restype = ctypes.c_char_p
    L.qkh_reader_open.restype = c_vp
    L.qkh_reader_open.argtypes = [ctypes.c_char_p]
    L.qkh_reader_close.argtypes = [c_vp]
    L.qkh_reader_close.restype = None
    L.qkh_reader_done.argtypes = [c_vp]
    L.qkh_reader_fill.restype = ctypes.c_int64
    L.qkh_reader_fill.argtypes = [c_vp, c_vp, c_vp, c_vp, ctypes.c_uint64, ctypes.c_uint64,
                                  c_u64p, c_u32p]
    L.qkh_adapter_insert.argtypes = [c_vp, c_vp, ctypes.c_uint64]
    L.qkh_adapter_insert.restype = None
    L.qkh_read_adapters.argtypes = [ctypes.c_char_p, c_vp]
    L.qkh_accumulate_file.argtypes = [ctypes.c_char_p, c_vp, ctypes.POINTER(ctypes.c_int), ctypes.c_int,
                                      ctypes.POINTER(c_vp), c_u64p, c_u64p]
    L.qkh_render_document.argtypes = [c_vp, c_vp, ctypes.c_char_p, ctypes.c_int,
                                      c_vp, ctypes.c_uint64, ctypes.c_uint64,
                                      c_vp, ctypes.c_uint64, ctypes.c_uint64]
    L.qkh_main.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p)]
    _host = L
    return L


def libc():
    L = ctypes.CDLL(None)
    L.fopen.restype = c_vp
    L.fopen.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    L.fclose.argtypes = [c_vp]
    L.free.argtypes = [c_vp]
    L.free.restype = None
    return L

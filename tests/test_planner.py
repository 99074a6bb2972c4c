"""Launch planner (qk_shim.hip: make_plan) — host logic, no GPU needed: the
invariants the kernels rely on, over a sweep of batch shapes."""
import ctypes
import itertools

import pytest

from quack_amd import _capi

L = _capi.hip()
L.qk_debug_plan.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_uint32,
                            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
KEYS = ("n_tiles tile_pos ch rw unroll pipe reads_per_slice n_slices n_blocks lds halo fused dynamic aligned "
        "replicas row_dwords").split()


def plan(n_reads, max_len, ragged=False, adapters=False, bucket_log2=10, gapped=False, aligned=False, n_cu=256):
    out = (ctypes.c_uint64 * 16)()
    rc = L.qk_debug_plan(n_reads, max_len, ragged, adapters, bucket_log2, gapped, aligned, n_cu, out)
    if rc:
        raise RuntimeError(L.qk_last_error().decode())
    return dict(zip(KEYS, out))


LENGTHS = [1, 7, 8, 9, 36, 40, 50, 56, 57, 64, 72, 76, 80, 81, 88, 100, 150, 151, 250, 300, 304, 448, 449, 512, 576,
           577, 1000, 4096, 10500, 20000, 100000, 3_000_000]


@pytest.mark.parametrize("ragged,adapters", list(itertools.product([False, True], [False, True])))
def test_plans_fit_the_hardware(ragged, adapters):
    for max_len, n_reads, blog in itertools.product(LENGTHS, [1, 1000, 10_000_000, 3_000_000_000 // 150], [0, 6, 10]):
        if n_reads * max_len > 1 << 36:
            continue
        p = plan(n_reads, max_len, ragged, adapters, blog)
        ctx = (max_len, n_reads, ragged, adapters, blog, p)
        assert p["lds"] <= 160 * 1024, ctx                                  # one workgroup's LDS
        assert p["tile_pos"] == 8 * p["ch"] and p["n_tiles"] * p["tile_pos"] >= max_len, ctx
        assert (p["n_tiles"] - 1) * p["tile_pos"] < max_len, ctx            # no empty tile
        w16 = bool(p["aligned"] & 2)          # 16 positions per lane: a lane owns two adjacent chunks
        lanes = 1024 // 64 * (63 if w16 else 62) if p["fused"] else 1024
        assert not w16 or (p["ch"] % 2 == 0 and p["aligned"] & 1), ctx
        assert p["rw"] >= 1 and p["rw"] * (p["ch"] // (2 if w16 else 1) + p["halo"]) <= lanes, ctx
        assert p["halo"] == ((1 if w16 else 2) if p["fused"] and p["n_tiles"] > 1 else 0), ctx
        step = p["rw"] * p["unroll"]
        assert p["reads_per_slice"] % step == 0 and p["reads_per_slice"] + step <= 65535, ctx   # u16 LDS counters
        assert p["n_slices"] * p["reads_per_slice"] >= n_reads, ctx          # every read belongs to a slice
        assert p["reads_per_slice"] * max_len <= 0x7FFFFFFF or max_len > 0x7FFFFFFF // step, ctx  # 32-bit offsets
        assert p["row_dwords"] % 32 == 0 and p["row_dwords"] >= 4 * p["replicas"] * p["ch"], ctx  # bank == column
        assert p["dynamic"] == (p["n_tiles"] > 1), ctx
        assert p["n_blocks"] >= 1 and (p["dynamic"] or p["n_blocks"] == p["n_slices"]), ctx
        assert (p["unroll"], p["pipe"]) == ((4, 1) if ragged else ((1, 2) if w16 else ((2, 2) if adapters else (1, 2)))), ctx
        # fixed-length reads of a multiple of 4 bases take the dword-aligned variant, 16 positions per lane
        # ... and with the adapter scan fused in, 16 positions per lane
        pairs_pay = max_len >= 64 or (max_len + 15) // 16 * 16 == (max_len + 7) // 8 * 8
        assert p["aligned"] == ((3 if adapters and pairs_pay else 1) if (not ragged and max_len % 4 == 0) else 0), ctx


def test_cache_line_plans():
    for max_len in (600, 1000, 5000, 20000, 100000):
        for adapters in (False, True):
            p = plan(100000, max_len, ragged=True, adapters=adapters, gapped=True, aligned=True)
            assert p["aligned"] == 3 and p["tile_pos"] % 128 == 0 and p["n_tiles"] > 1 and p["lds"] <= 160 * 1024, (max_len, p)
            assert (p["unroll"], p["pipe"]) == (1, 2) and p["rw"] * (p["ch"] // 2 + p["halo"]) <= (1008 if adapters else 1024), (max_len, p)
    p = plan(100000, 150, ragged=True, gapped=True, aligned=True)       # one tile: nothing to align
    assert not p["aligned"] and p["n_tiles"] == 1
    with pytest.raises(RuntimeError):
        plan(10, 0xFFFFFFF0, ragged=True)                                # a read that no slice can address


def test_rows_of_several_reads():
    """round 4 (choose_group + the plan of a row, qk_debug_group): fixed-length reads with the adapter scan are taken as rows
    of several reads where that fills the 16-position lanes better; whatever is chosen, a row fits one tile beside the adapter
    tables, runs 16 positions per lane, and a step's reads fit half the smallest first-hit ring"""
    L.qk_debug_group.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64)]

    def group(n, read_len, stride, blog=9):
        out = (ctypes.c_uint64 * 8)()
        assert L.qk_debug_group(n, read_len, stride, blog, out) == 0, L.qk_last_error()
        return dict(zip("group row tile_pos rw unroll w16 bucket_log2 lds".split(), out))

    want = {36: 4, 50: 4, 76: 4, 100: 3, 125: 1, 150: 2, 151: 2, 200: 1, 250: 1, 300: 1}
    for read_len, g in want.items():
        p = group(10_000_000, read_len, (read_len + 3) & ~3)
        assert p["group"] == g, (read_len, p)
    for read_len in list(range(11, 330)) + [400, 448, 500]:
        for blog in (0, 6, 9, 10):
            stride = (read_len + 3) & ~3
            p = group(1_000_000, read_len, stride, blog)
            assert p["lds"] <= 160 * 1024, (read_len, blog, p)
            assert p["row"] == (p["group"] - 1) * stride + read_len
            if p["group"] > 1:
                assert p["w16"] == 1 and p["tile_pos"] >= p["row"] and p["tile_pos"] % 16 == 0 and p["tile_pos"] <= 352, (read_len, blog, p)
                assert p["rw"] * p["unroll"] * p["group"] <= 1024, (read_len, blog, p)
                assert p["bucket_log2"] == blog or blog == 0, (read_len, blog, p)      # never at the price of the exact LDS table
                lanes1 = -(-read_len // 16) * 16 / read_len
                lanesg = p["tile_pos"] / (p["group"] * read_len)
                assert lanesg < lanes1 * 0.985, (read_len, p)                          # ... and only where it pays


def test_experiment_switches_live_in_the_experiment_build_only(monkeypatch):
    """Round 5: the product library reads no switch that selects another launch geometry — QUACK_HIP_TUNE is parsed by the
    -DQK_EXPERIMENT build alone (libquack_hip_exp.so), which the Python mirror loads exactly when the variable is set; the old
    per-switch variables are gone from both."""
    X = _capi.hip_exp()
    for lib in (L, X):
        lib.qk_debug_group.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64)]

    def group(lib):
        out = (ctypes.c_uint64 * 8)()
        assert lib.qk_debug_group(1_000_000, 150, 152, 9, out) == 0
        return out[0]

    def threads(lib):
        out = (ctypes.c_uint64 * 16)()
        assert lib.qk_debug_plan(1_000_000, 150, 0, 0, 0, 0, 0, 256, out) == 0
        return dict(zip(KEYS, out))["n_blocks"], dict(zip(KEYS, out))["unroll"]

    assert group(L) == 2 and group(X) == 2            # the planner's own choice: rows of two 150 bp reads
    base = threads(L)
    monkeypatch.setenv("QUACK_HIP_TUNE", "group=1,unroll=4")
    assert group(L) == 2 and group(X) == 1            # only the experiment build listens
    assert threads(L) == base and threads(X)[1] == 4
    monkeypatch.setenv("QUACK_HIP_GROUP", "1")        # (round 4's variable: nobody listens any more)
    monkeypatch.delenv("QUACK_HIP_TUNE")
    assert group(L) == 2 and group(X) == 2
    # ... and the strings of the product library hold no experiment switch
    import re
    blob = open(_capi.hip()._name, "rb").read()
    names = set(re.findall(rb"QUACK_[A-Z0-9_]+", blob))
    allowed = {b"QUACK_VERBOSE", b"QUACK_HIP_BATCH_MB", b"QUACK_HIP_BATCH_KB", b"QUACK_HIP_CHECK_PADS", b"QUACK_HIP_RCCL_ALWAYS",
               b"QUACK_HIP_UNFUSED_ADAPTERS", b"QUACK_HIP_TUNE"}   # (QUACK_HIP_TUNE: named in an error message, never read)
    extra = {n for n in names if n not in allowed and not n.startswith(b"QUACK_HIP_NO_")}
    assert not extra, extra


def test_wide_lds_layout_pieces_never_overlap():
    """Round 5: the fixed-length adapter kernel's LDS map (qk::wide_plan, shared by planner and kernel) — counter planes 64 KiB
    apart, the 9-mer filter in the first gap, queue + letter rows + bucket table in the second, the first-hit ring in the unused
    column pairs of the last plane or behind it.  For every shape: the pieces lie inside the dynamic segment asked for, inside
    160 KiB, and do not overlap one another or a plane; a shape that does not fit says so (bytes == 0)."""
    L.qk_debug_wide.argtypes = [ctypes.c_uint32] * 4 + [ctypes.POINTER(ctypes.c_uint64)]
    PLANE, STRIDE, FILTER, REST, TOP, QUEUE = 8192, 16384, 8192, 24576, 32768, 4096     # dwords (qk_kernels.hip.h: kWide*)
    fits = 0
    for ch in range(2, 60, 2):
        for rep in (1, 2, 3, 4, 6):
            for blog in (0, 6, 8, 9, 10):
                for fh in (2048, 4096, 8192, 16384):
                    out = (ctypes.c_uint64 * 8)()
                    assert L.qk_debug_wide(ch, rep, blog, fh, out) == 0
                    planes, b_at, r_at, nbytes, spare, rest = (int(x) for x in out[:6])
                    cols = 4 * rep * ch
                    assert planes == (cols + 63) // 64
                    if nbytes == 0:
                        continue
                    fits += 1
                    assert planes <= 3 and nbytes <= 160 * 1024 and nbytes % 4 == 0
                    end = nbytes // 4
                    pieces = [(p * STRIDE, p * STRIDE + PLANE, "plane %d" % p) for p in range(planes)]
                    pieces.append((FILTER, FILTER + 8192, "filter"))
                    pieces.append((REST, REST + rest, "queue + rows"))
                    assert rest >= QUEUE + 6 * 8 * ch
                    if blog:
                        pieces.append((b_at, b_at + (4 << blog), "buckets"))
                    if r_at:
                        pieces.append((r_at, r_at + fh, "ring"))
                    else:
                        assert fh <= spare and spare in (2048, 4096)
                        # the ring's words: the last `spare / 256` column pairs of every row of the last plane — beyond the pairs in use
                        free_pairs = planes * 32 - cols // 2
                        assert spare // 256 <= free_pairs
                    for a, e, what in pieces:
                        assert 0 <= a < e <= end, (ch, rep, blog, fh, what, a, e, end)
                    pieces.sort()
                    for (a0, e0, w0), (a1, e1, w1) in zip(pieces, pieces[1:]):
                        assert e0 <= a1, (ch, rep, blog, fh, w0, w1)
    assert fits > 500

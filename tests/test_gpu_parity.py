"""Parity of the HIP path (through the C-ABI) with the oracle.  Needs an
MI355X: run with `pytest -m gpu`.  Integer work: the bar is bit-exact."""
import os
import subprocess

import numpy as np
import pytest

import cases
import oracle_binding as ob
import synth
import quack_amd
from quack_amd.api import pad_for_device

pytestmark = pytest.mark.gpu


def hip_table(seq, qual, offsets=None, read_len=0, kmers_bits=None, chunks=1, **cfg):
    # (a launch geometry of the caller's choosing — configure() — is the experiment build's business; so is QUACK_HIP_TUNE)
    with quack_amd.Accumulator(0, kmers_bits, experiment=True if cfg else None) as acc:
        if cfg:
            acc.configure(**cfg)
        if offsets is None:
            n = len(seq) // read_len
            cut = [n * i // chunks for i in range(chunks + 1)]
            for a, b in zip(cut, cut[1:]):
                acc.submit_fixed(seq[a * read_len:b * read_len], qual[a * read_len:b * read_len], read_len)
        else:
            n = len(offsets) - 1
            cut = [n * i // chunks for i in range(chunks + 1)]
            for a, b in zip(cut, cut[1:]):
                lo, hi = int(offsets[a]), int(offsets[b])
                acc.submit(seq[lo:hi], qual[lo:hi], offsets[a:b + 1] - offsets[a])
        sd = acc.finish()
    return sd.bases, sd.number_of_sequences


def assert_same(got, want):
    gb, gn = got
    wb, wn = want
    assert gn == wn
    assert gb.shape == wb.shape
    if not np.array_equal(gb, wb):
        pos, row = np.argwhere(gb != wb)[0]
        raise AssertionError("first mismatch at position %d row %d: hip %d oracle %d (%d cells differ)"
                             % (pos, row, gb[pos, row], wb[pos, row], (gb != wb).sum()))


# ---------------------------------------------------------------- seeded batches
@pytest.mark.parametrize("n,L", [(20000, 150), (5000, 300), (70000, 36), (3000, 8), (3000, 9),
                                 (1, 150), (1000, 1), (999, 151), (2000, 250), (300, 1000)])
def test_fixed_length(n, L):
    seq, qual = synth.fixed(n, L, seed=n + L)
    assert_same(hip_table(seq, qual, read_len=L), ob.accumulate_batch(seq, qual, read_len=L))


@pytest.mark.parametrize("n,lo,hi", [(30000, 1, 150), (20000, 20, 151), (50000, 1, 12), (400, 1000, 20000),
                                     (64, 5000, 5000), (5000, 0, 40)])
def test_ragged(n, lo, hi):
    seq, qual, off = synth.ragged(n, lo, hi, seed=n + hi, q_lo=1, q_hi=60, alphabet=b"ACGTNacgtnRYKM")
    assert_same(hip_table(seq, qual, off), ob.accumulate_batch(seq, qual, off))


@pytest.mark.parametrize("L", [300, 150, 40])
def test_adapters_fixed(L):
    ads = synth.synthetic_adapters()
    seq, qual = synth.fixed(20000, L, seed=3)
    seq = synth.splice_adapters(seq, L, ads, seed=4)
    k = ob.kmers_from_seqs(ads)
    assert_same(hip_table(seq, qual, read_len=L, kmers_bits=ob.kmers_to_bitset(k)),
                ob.accumulate_batch(seq, qual, read_len=L, kmers=k))
    assert ob.accumulate_batch(seq, qual, read_len=L, kmers=k)[0][:, 96].sum() > 100  # hits happened


@pytest.mark.parametrize("adapters", [False, True], ids=["plain", "adapters"])
def test_sixteen_positions_per_lane_every_shape(adapters, monkeypatch):
    monkeypatch.setenv("QUACK_HIP_TUNE", "w16_always=1")      # (the planner picks it by itself only with the adapter scan)
    """fixed-length reads of a multiple of 4 bases run with 16 positions per lane (hist_kernel W16: two adjacent
    chunks per lane, one dwordx4 per array, odd chunk counts rounded up to whole pairs, one feeder lane per wave,
    one halo lane in front of a tile, per-lane candidate entries of 16 windows): every read length 4..128, lengths
    around the one-tile limit with and without the adapter tables, two and more tiles, adapters at every offset of
    a lane's 16 positions; and the same reads with 8 positions per lane (QUACK_HIP_NO_W16) must agree too"""
    ads = synth.synthetic_adapters()
    k = ob.kmers_from_seqs(ads) if adapters else None
    bits = ob.kmers_to_bitset(k) if adapters else None
    rng = np.random.default_rng(16)
    for L in list(range(4, 132, 4)) + [148, 152, 156, 160, 296, 300, 304, 308, 444, 448, 452, 572, 576, 580, 600, 1024, 1500]:
        n = 1500 if L <= 160 else 400
        seq, qual = synth.fixed(n, L, seed=L, q_lo=0, q_hi=60)
        seq = seq.copy().reshape(n, L)
        if adapters and L >= 14:
            for r in range(0, n, 3):                     # an adapter at a sweep of offsets (every position of a lane's 16)
                ad = np.frombuffer(ads[r % len(ads)], np.uint8)
                at = (r // 3) % max(1, L - 10)
                m = min(len(ad), L - at)
                seq[r, at:at + m] = ad[:m]
        seq = seq.reshape(-1)
        want = ob.accumulate_batch(seq, qual, read_len=L, kmers=k)
        assert_same(hip_table(seq, qual, read_len=L, kmers_bits=bits, chunks=1 + L % 3), want)
    assert not adapters or want[0][:, 96].sum() > 0


def padded_layout(seq, qual, n, L, stride, seed=11):
    """the reads of a packed fixed-length batch laid out `stride` bytes apart; the pad bytes are valid-looking letters
    and scores that must never be counted"""
    rng = np.random.default_rng(seed)
    s2 = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (n, stride))].copy()
    q2 = (33 + rng.integers(0, 60, (n, stride))).astype(np.uint8)
    s2[:, :L] = seq.reshape(n, L)
    q2[:, :L] = qual.reshape(n, L)
    return s2.reshape(-1), q2.reshape(-1)


@pytest.mark.parametrize("adapters", [False, True], ids=["plain", "adapters"])
def test_padded_fixed_length_every_length(adapters, monkeypatch):
    """round 4: uniform reads whose length is NOT a multiple of 4 (150, 250, 125, 50 bp) laid out at a stride rounded up
    to 4 run the dword-aligned kernels — with the adapter scan the 16-positions-per-lane one.  Every read length 4..160
    (and around the tile limits), adapters at every offset of a lane's 16 positions, through three routes into ONE
    accumulator: the copying host feed (qk_accum_submit_fixed pads on the way into the pinned slot), the device API
    with a padded layout (qk_accum_submit_device_strided, lengths NULL; garbage in the pad bytes), and the packed
    device API (the 12-byte-window kernels): 3 x the oracle's table."""
    import torch
    overridden = [e for e in os.environ if e.startswith("QUACK_HIP_")]      # (tools/stress_gpu.sh runs this file under overrides)
    if not adapters:
        monkeypatch.setenv("QUACK_HIP_TUNE", "pad_always=1")   # (the feeds pad by themselves only with the adapter scan)
    ads = synth.synthetic_adapters()
    k = ob.kmers_from_seqs(ads) if adapters else None
    bits = ob.kmers_to_bitset(k) if adapters else None
    hits = 0
    for L in list(range(4, 161)) + [249, 250, 251, 299, 301, 445, 446, 447, 449, 573, 575, 577, 601, 1001]:
        n = 1200 if L <= 160 else 300
        seq, qual = synth.fixed(n, L, seed=L, q_lo=0, q_hi=60)
        seq = seq.copy().reshape(n, L)
        if adapters and L >= 14:
            for r in range(0, n, 3):
                ad = np.frombuffer(ads[r % len(ads)], np.uint8)
                at = (r // 3) % max(1, L - 10)
                m = min(len(ad), L - at)
                seq[r, at:at + m] = ad[:m]
        seq = seq.reshape(-1)
        want = ob.accumulate_batch(seq, qual, read_len=L, kmers=k)
        hits += int(want[0][:, 96].sum())
        stride = (L + 3) & ~3
        s2, q2 = padded_layout(seq, qual, n, L, stride)
        with quack_amd.Accumulator(0, bits) as acc:
            if not overridden:
                assert acc.padded_stride(L) == (stride if L % 4 and L >= 16 or (L % 4 and not adapters) else 0)
            acc.submit_fixed(seq, qual, L)
            d_s, d_q = torch.from_numpy(pad_for_device(s2)).cuda(), torch.from_numpy(pad_for_device(q2)).cuda()
            acc.submit_device_padded(d_s, d_q, n, L, stride)
            p_s, p_q = torch.from_numpy(pad_for_device(seq)).cuda(), torch.from_numpy(pad_for_device(qual)).cuda()
            acc.submit_device(p_s, p_q, None, n, n * L, L)
            sd = acc.finish()
        assert sd.number_of_sequences == 3 * want[1]
        try:
            assert_same((sd.bases, want[1]), (3 * want[0], want[1]))
        except AssertionError as e:
            raise AssertionError("read length %d: %s" % (L, e))
    assert not adapters or hits > 1000


@pytest.mark.parametrize("forced", [None, "1", "2", "3", "4", "5", "8"])
def test_grouped_rows_every_group_size(forced, monkeypatch):
    """round 4: with the adapter scan a ROW of the fixed-length kernel is several consecutive reads (two 150 bp reads 152
    bytes apart fill 19 lanes of 16 positions where one fills 10 with 6 % idle; three of 100 bp; four of 36 bp): the flush
    folds the column groups, the candidate check splits a lane's windows between the two reads it may straddle, the first-hit
    ring holds a word per read.  Every group size that fits (QUACK_HIP_GROUP) and the planner's own choice, read counts that
    leave a remainder, adapters at every offset of every read of a row (hits at position 9, at the last base, across the
    seam between two reads — which must not count), packed and padded strides; and QUACK_HIP_NO_GROUP agrees"""
    import torch
    if forced:
        monkeypatch.setenv("QUACK_HIP_TUNE", "group=" + str(forced))
    ads = synth.synthetic_adapters()
    k = ob.kmers_from_seqs(ads)
    bits = ob.kmers_to_bitset(k)
    hits = 0
    for L in (11, 16, 17, 20, 31, 36, 40, 50, 64, 75, 76, 100, 101, 125, 148, 149, 150, 151, 152, 160, 250, 300):
        n = 3001 if L <= 160 else 803               # (odd: a remainder for every group size but 1)
        seq, qual = synth.fixed(n, L, seed=1000 + L, q_lo=0, q_hi=60)
        seq = seq.copy().reshape(n, L)
        for r in range(n):                          # an adapter in two reads out of three, every offset, also cut off by the read's end
            if r % 3 == 2:
                continue
            ad = np.frombuffer(ads[r % len(ads)], np.uint8)
            at = (r // 3 * 7 + r) % max(1, L - 9) if r % 3 == 0 else max(0, L - 10 - (r // 3) % 12)
            m = min(len(ad), L - at)
            seq[r, at:at + m] = ad[:m]
        # ... a false candidate right in front of a true adapter (nine bases of an adapter and a wrong tenth: passes the 9-mer
        # filter, is in no table), so that the lane of the true first hit stands behind a lane that confirms nothing
        if L >= 60:
            for r in range(2, n, 3):
                ad = np.frombuffer(ads[(r + 2) % len(ads)], np.uint8)
                at = (r * 5) % (L - 45)
                seq[r, at:at + 9] = ad[:9]
                seq[r, at + 9] = ord("A") if ad[9] != ord("A") else ord("C")
                gap = 10 + (r // 3) % 20
                m = min(len(ad), L - at - gap)
                seq[r, at + gap:at + gap + m] = ad[:m]
        # ... and adapter 10-mers that only exist ACROSS the seam of two neighbouring reads (never a hit)
        for r in range(1, n, 5):
            ad = np.frombuffer(ads[(r + 1) % len(ads)], np.uint8)
            cut = 1 + r % 9
            if L > 30:
                seq[r - 1, L - cut:] = ad[:cut]
                seq[r, :10 - cut] = ad[cut:10]
        seq = seq.reshape(-1)
        want = ob.accumulate_batch(seq, qual, read_len=L, kmers=k)
        hits += int(want[0][:, 96].sum())
        stride = (L + 3) & ~3
        s2, q2 = padded_layout(seq, qual, n, L, stride, seed=L)
        for env in ({}, {"QUACK_HIP_NO_GROUP": "1"}, {"QUACK_HIP_TUNE": "small_ring=1"}) if forced is None else ({},):
            for kk, v in env.items():
                monkeypatch.setenv(kk, v)
            with quack_amd.Accumulator(0, bits) as acc:
                d_s, d_q = torch.from_numpy(pad_for_device(s2)).cuda(), torch.from_numpy(pad_for_device(q2)).cuda()
                acc.submit_device_padded(d_s, d_q, n, L, stride)
                acc.submit_fixed(seq, qual, L)
                sd = acc.finish()
            for kk in env:
                monkeypatch.delenv(kk)
            try:
                assert_same((sd.bases, sd.number_of_sequences), (2 * want[0], 2 * n))
            except AssertionError as e:
                raise AssertionError("read length %d, %s: %s" % (L, env, e))
    assert hits > 5000


def test_padded_batches_through_the_pinned_slots_and_under_overrides(monkeypatch):
    """qk_accum_acquire / qk_accum_commit_padded (what the host feed drives), several batches with table growth in
    between; and the padded form under tuning overrides, where the 12-byte-window kernels take the stride as it is"""
    ads = synth.synthetic_adapters()
    k = ob.kmers_from_seqs(ads)
    bits = ob.kmers_to_bitset(k)
    total, n_all = None, 0
    with quack_amd.Accumulator(0, bits) as acc:
        for L in (50, 150, 151, 250):
            n = 4000
            seq, qual = synth.fixed(n, L, seed=L + 1)
            seq = synth.splice_adapters(seq, L, ads, seed=L + 2, fraction=0.5)
            want = ob.accumulate_batch(seq, qual, read_len=L, kmers=k)
            stride = acc.padded_stride(L) or (L + 3) & ~3      # (0 under a tuning override: the explicit form still works)
            s2, q2 = padded_layout(seq, qual, n, L, stride)
            hs, hq, _ = acc.acquire()
            hs[:n * stride] = s2
            hq[:n * stride] = q2
            acc.commit_padded(n, L, stride)
            grown = np.zeros((max(L, 0 if total is None else total.shape[0]), 97), np.uint64)
            if total is not None:
                grown[:total.shape[0]] += total
            grown[:L] += want[0]
            total, n_all = grown, n_all + n
        sd = acc.finish()
    assert_same((sd.bases, sd.number_of_sequences), (total, n_all))
    seq, qual = synth.fixed(6000, 150, seed=9)
    seq = synth.splice_adapters(seq, 150, ads, seed=10)
    want = ob.accumulate_batch(seq, qual, read_len=150, kmers=k)
    s2, q2 = padded_layout(seq, qual, 6000, 150, 152)
    import torch
    d_s, d_q = torch.from_numpy(pad_for_device(s2)).cuda(), torch.from_numpy(pad_for_device(q2)).cuda()
    for cfg, env in ((dict(threads=512), {}), (dict(unroll=2), {}), ({}, {"QUACK_HIP_TUNE": "pipe=1"}), ({}, {"QUACK_HIP_NO_ALIGN4": "1"}),
                     ({}, {"QUACK_HIP_NO_W16": "1"}), ({}, {"QUACK_HIP_UNFUSED_ADAPTERS": "1"}), ({}, {"QUACK_HIP_TUNE": "separate_count=1"}),
                     ({}, {"QUACK_HIP_NO_PAD": "1"}), (dict(tile=64), {})):
        for kk, v in env.items():
            monkeypatch.setenv(kk, v)
        with quack_amd.Accumulator(0, bits, experiment=True if cfg else None) as acc:
            if cfg:
                acc.configure(**cfg)
            acc.submit_device_padded(d_s, d_q, 6000, 150, 152)
            acc.submit_fixed(seq, qual, 150)
            sd = acc.finish()
        assert_same((sd.bases, 6000), (2 * want[0], 6000))
        for kk in env:
            monkeypatch.delenv(kk)
    with quack_amd.Accumulator(0) as acc:
        with pytest.raises(quack_amd.HipUnavailable):
            acc.submit_device_padded(d_s, d_q, 10, 150, 154)    # stride % 4
        with pytest.raises(quack_amd.HipUnavailable):
            acc.submit_device_padded(d_s, d_q, 10, 153, 152)    # read > stride


def test_sixteen_positions_per_lane_on_and_off(monkeypatch):
    import torch
    ads = synth.synthetic_adapters()
    k = ob.kmers_from_seqs(ads)
    bits = ob.kmers_to_bitset(k)
    seq, qual = synth.fixed(30000, 300, seed=5)
    seq = synth.splice_adapters(seq, 300, ads, seed=6)
    want = ob.accumulate_batch(seq, qual, read_len=300, kmers=k)
    for env in ({}, {"QUACK_HIP_NO_W16": "1"}):
        for kk, v in env.items():
            monkeypatch.setenv(kk, v)
        assert_same(hip_table(seq, qual, read_len=300, kmers_bits=bits), want)
        for kk in env:
            monkeypatch.delenv(kk)
    # without the adapter scan fixed-length reads keep 8 positions per lane (faster: memory-bound at 70 VGPRs); the
    # 16-position build of that shape exists and is forced here
    monkeypatch.setenv("QUACK_HIP_TUNE", "w16_always=1")
    for L in (4, 36, 100, 300, 580):
        s2, q2 = synth.fixed(3000, L, seed=L)
        assert_same(hip_table(s2, q2, read_len=L), ob.accumulate_batch(s2, q2, read_len=L))


def test_adapters_every_chunk_count_of_short_reads():
    """1..14 chunks per read: the replicated-column layouts next to the resident
    adapter tables (LDS budget), fixed and ragged"""
    ads = synth.synthetic_adapters()
    k = ob.kmers_from_seqs(ads)
    bits = ob.kmers_to_bitset(k)
    for L in (7, 12, 20, 31, 40, 44, 52, 60, 71, 80, 85, 100, 112):
        seq, qual = synth.fixed(3000, L, seed=L)
        seq = synth.splice_adapters(seq, L, ads, seed=L + 1, fraction=0.5)
        assert_same(hip_table(seq, qual, read_len=L, kmers_bits=bits), ob.accumulate_batch(seq, qual, read_len=L, kmers=k))
        off = np.arange(0, 3000 * L + 1, L, dtype=np.uint64)
        off[1:-1] -= np.random.default_rng(L).integers(0, 3, 2999).astype(np.uint64)   # ragged by a base or two
        off = np.sort(off)
        assert_same(hip_table(seq, qual, off, kmers_bits=bits), ob.accumulate_batch(seq, qual, off, kmers=k))


@pytest.mark.parametrize("replicas", [None, "1", "3"])
def test_every_tile_width_and_counter_replica_count(monkeypatch, replicas):
    """1..64 chunks per read: every layout of the quality counters (sets of columns, byte rotation per read
    row, replicas summed by the flush — qk::hist_replicas), with the planner's replica count and with fewer"""
    if replicas:
        monkeypatch.setenv("QUACK_HIP_TUNE", "replicas=" + str(replicas))
    for L in [3, 8, 15, 16, 17, 24, 33, 47, 56, 64, 79, 90, 101, 125, 149, 176, 211, 250, 255, 256, 301, 390, 448, 509]:
        n = 2500 if L < 200 else 900
        seq, qual = synth.fixed(n, L, seed=L, q_lo=1, q_hi=90)
        assert_same(hip_table(seq, qual, read_len=L), ob.accumulate_batch(seq, qual, read_len=L))


def test_many_reads_just_longer_than_a_tile():
    """two or three tiles and few distinct lengths: length_count must not funnel
    through a handful of global addresses (it took 11 ms per 3 Gbases), and the
    unsorted multi-tile path still has users"""
    for n, lo, hi in ((60000, 590, 600), (30000, 1100, 1101), (20000, 1, 2100)):
        seq, qual, off = synth.ragged(n, lo, hi, seed=hi)
        assert_same(hip_table(seq, qual, off), ob.accumulate_batch(seq, qual, off))


@pytest.mark.parametrize("n_adapters,length", [(400, 100), (6000, 200)])
def test_large_adapter_sets_take_the_global_table(n_adapters, length):
    """tens of thousands (up to most) of the 2^20 10-mers set: the exact LDS bucket
    table no longer fits, filter hits are confirmed in the global bit set, and
    nearly every window passes the filters"""
    rng = np.random.default_rng(n_adapters)
    ads = [bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, length)]) for _ in range(n_adapters)]
    k = ob.kmers_from_seqs(ads)
    bits = ob.kmers_to_bitset(k)
    assert int(np.asarray(k).sum()) > 30000
    seq, qual = synth.fixed(20000, 150, seed=7)
    assert_same(hip_table(seq, qual, read_len=150, kmers_bits=bits), ob.accumulate_batch(seq, qual, read_len=150, kmers=k))
    seq, qual, off = synth.ragged(400, 500, 6000, seed=8)
    assert_same(hip_table(seq, qual, off, kmers_bits=bits), ob.accumulate_batch(seq, qual, off, kmers=k))


def test_adapters_ragged_and_long():
    ads = synth.synthetic_adapters()
    k = ob.kmers_from_seqs(ads)
    for n, lo, hi in [(20000, 1, 120), (300, 1000, 20000)]:
        seq, qual, off = synth.ragged(n, lo, hi, seed=hi)
        rng = np.random.default_rng(5)
        for r in rng.integers(0, n, n // 3):           # splice adapters anywhere, also near the end
            a, b = int(off[r]), int(off[r + 1])
            ad = np.frombuffer(ads[int(rng.integers(0, len(ads)))], np.uint8)
            at = a + int(rng.integers(0, max(1, b - a)))
            m = min(len(ad), b - at)
            seq[at:at + m] = ad[:m]
        assert_same(hip_table(seq, qual, off, kmers_bits=ob.kmers_to_bitset(k)),
                    ob.accumulate_batch(seq, qual, off, kmers=k))


@pytest.mark.parametrize("shape", ["fixed300", "ragged150", "ragged_long", "fixed_long"])
def test_fused_and_separate_adapter_paths_agree(shape, monkeypatch):
    """the adapter scan fused into the histogram pass (default) and the separate
    scan kernels (QUACK_HIP_UNFUSED_ADAPTERS=1) are two implementations of
    quack.c:206-217; both must equal the oracle"""
    ads = synth.synthetic_adapters()
    k = ob.kmers_from_seqs(ads)
    bits = ob.kmers_to_bitset(k)
    rng = np.random.default_rng(8)
    if shape == "fixed300":
        seq, qual = synth.fixed(8000, 300, seed=81)
        seq = synth.splice_adapters(seq, 300, ads, seed=82)
        off, L = None, 300
    elif shape == "fixed_long":
        seq, qual = synth.fixed(300, 3000, seed=83)
        seq = synth.splice_adapters(seq, 3000, ads, seed=84, fraction=0.8)
        off, L = None, 3000
    else:
        lo, hi, n = (1, 150, 20000) if shape == "ragged150" else (500, 9000, 600)
        seq, qual, off = synth.ragged(n, lo, hi, seed=85)
        L = 0
        for r in rng.integers(0, n, n // 2):          # adapters anywhere, also across tile seams
            a, b = int(off[r]), int(off[r + 1])
            ad = np.frombuffer(ads[int(rng.integers(0, len(ads)))], np.uint8)
            at = a + int(rng.integers(0, max(1, b - a)))
            m = min(len(ad), b - at)
            seq[at:at + m] = ad[:m]
    want = ob.accumulate_batch(seq, qual, off, read_len=L, kmers=k)
    assert_same(hip_table(seq, qual, off, read_len=L, kmers_bits=bits), want)
    monkeypatch.setenv("QUACK_HIP_UNFUSED_ADAPTERS", "1")
    assert_same(hip_table(seq, qual, off, read_len=L, kmers_bits=bits), want)
    monkeypatch.setenv("QUACK_HIP_TUNE", "tile=64")             # many tiles, halo lanes at every seam
    monkeypatch.delenv("QUACK_HIP_UNFUSED_ADAPTERS")
    assert_same(hip_table(seq, qual, off, read_len=L, kmers_bits=bits), want)


def test_adapter_hit_positions_edge_cases():
    """first window un-inserted, hit in the seed window, hit ending on the last
    base (not counted), l == 10, l < 10 — SURVEY §8a rows a6/a7"""
    k = ob.kmers_from_seqs(["ACGTTGCAAGGCT"])
    reads = [b"ACGTTGCAAGAAAA", b"CGTTGCAAGGAA", b"AACGTTGCAAGGAA", b"AACGTTGCAAGG", b"CGTTGCAAGG", b"NGT",
             b"CGTTGCAAGGCTAAAAAAAAAAAAAAAAAAAACGTTGCAAGG"]
    seq = np.frombuffer(b"".join(reads), np.uint8)
    qual = np.full(len(seq), ord("5"), np.uint8)
    off = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint64)
    got = hip_table(seq, qual, off, kmers_bits=ob.kmers_to_bitset(k))
    assert_same(got, ob.accumulate_batch(seq, qual, off, kmers=k))
    assert got[0][10, 96] == 2 and got[0][12, 96] == 1


def test_every_byte_value_is_defined_and_identical():
    """bytes outside the reference's defined domain (UB there) follow the rules
    fixed in oracle/quack_oracle.c — identically on the GPU"""
    rng = np.random.default_rng(11)
    lens = rng.integers(1, 200, 20000)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    seq = rng.integers(0, 256, int(off[-1])).astype(np.uint8)
    qual = rng.integers(0, 256, int(off[-1])).astype(np.uint8)
    k = ob.kmers_from_seqs(synth.synthetic_adapters())
    assert_same(hip_table(seq, qual, off), ob.accumulate_batch(seq, qual, off))
    assert_same(hip_table(seq, qual, off, kmers_bits=ob.kmers_to_bitset(k)),
                ob.accumulate_batch(seq, qual, off, kmers=k))


@pytest.mark.parametrize("L", [300, 150, 100, 64])
@pytest.mark.parametrize("alphabet", ["all256", "rare_iupac", "acgtn_mixed_case"])
def test_base_codes_by_table_lookup_are_exact_on_every_byte(L, alphabet):
    """The fixed-length adapter kernel (16 positions per lane) takes a base's 2-bit code from bits 3..1 of the byte by
    table lookup (round 5) and falls back, a wave and a step at a time, to the exact five-bit rule when a byte is met for
    which the two differ: the other letters of A..T (quack.c:148-150 maps B, D, E, F, R, S ... to A) and, outside the
    reference's domain, the oracle's `c & 31` rule.  All 256 byte values (every step takes the exact path), ordinary reads
    with one such byte in ~2000 (most steps take the lookup, some the exact path, inside one launch), and A C G T N in
    either case (the lookup alone) — adapters spliced in, so that content counters AND first hits are compared."""
    rng = np.random.default_rng(1000 + L)
    n = 30001
    ads = synth.synthetic_adapters()
    seq, qual = synth.fixed(n, L, seed=L)
    if alphabet == "all256":
        seq = rng.integers(0, 256, n * L).astype(np.uint8)
    elif alphabet == "rare_iupac":
        odd = np.frombuffer(b"BDEFHIJKLMOPQRSUVWXYbdefhijklmopqrsuvwxy@[`{\x00\xff\x7f0123456789", np.uint8)
        at = np.flatnonzero(rng.random(n * L) < 1 / 2000)
        seq = seq.copy()
        seq[at] = odd[rng.integers(0, len(odd), len(at))]
    else:
        seq = np.frombuffer(b"ACGTNacgtn", np.uint8)[rng.integers(0, 10, n * L)]
    seq = synth.splice_adapters(seq, L, ads, seed=5)
    if alphabet == "rare_iupac":   # ... and some of them INSIDE an adapter occurrence: the window must stop matching
        at = np.flatnonzero(rng.random(n * L) < 1 / 5000)
        seq[at] = np.frombuffer(b"RSBDVWU", np.uint8)[rng.integers(0, 7, len(at))]
    k = ob.kmers_from_seqs(ads)
    want = ob.accumulate_batch(seq, qual, read_len=L, kmers=k)
    assert_same(hip_table(seq, qual, read_len=L, kmers_bits=ob.kmers_to_bitset(k)), want)
    assert want[0][:, 96].sum() > 100


def test_four_level_quality_worst_case_contention():
    rng = np.random.default_rng(9)
    n, L = 40000, 150
    seq, _ = synth.fixed(n, L, seed=9)
    qual = (33 + np.array([2, 12, 23, 37], np.uint8)[rng.choice(4, n * L, p=[.03, .05, .12, .80])]).astype(np.uint8)
    assert_same(hip_table(seq, qual, read_len=L), ob.accumulate_batch(seq, qual, read_len=L))


# ------------------------------------------------------- accumulation mechanics
def test_many_submits_and_table_growth():
    """short reads first, longer later (realloc path, quack.c:194-198);
    fixed and ragged batches mixed into one accumulator"""
    parts = [synth.ragged(5000, 1, 30, seed=1), synth.ragged(5000, 10, 140, seed=2),
             synth.ragged(500, 200, 3000, seed=3)]
    f_seq, f_qual = synth.fixed(4000, 100, seed=4)
    with quack_amd.Accumulator(0, None, max_len_hint=16) as acc:
        for s, q, o in parts:
            acc.submit(s, q, o)
        acc.submit_fixed(f_seq, f_qual, 100)
        sd = acc.finish()
    seq = np.concatenate([p[0] for p in parts] + [f_seq])
    qual = np.concatenate([p[1] for p in parts] + [f_qual])
    lens = np.concatenate([np.diff(p[2].astype(np.int64)) for p in parts] + [np.full(4000, 100)])
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    assert_same((sd.bases, sd.number_of_sequences), ob.accumulate_batch(seq, qual, off))


# ------------------------------------------------------------ gapped batches
def gapped(seq, qual, off, align, seed):
    """same reads, but every read starts on an `align`-byte boundary (align=0: after
    a random gap of 0..300 bytes); the gaps hold bytes that would count if touched"""
    g = np.random.default_rng(seed)
    lens = np.diff(off.astype(np.int64))
    starts, pos = np.zeros(len(lens), np.int64), 0
    for r, l in enumerate(lens):
        pos = (pos + align - 1) // align * align if align else pos + int(g.integers(0, 301))
        starts[r] = pos
        pos += int(l)
    gs = g.choice(np.frombuffer(b"ACGT", np.uint8), pos).astype(np.uint8)
    gq = g.integers(35, 70, pos).astype(np.uint8)
    for r, l in enumerate(lens):
        gs[starts[r]:starts[r] + l] = seq[int(off[r]):int(off[r + 1])]
        gq[starts[r]:starts[r] + l] = qual[int(off[r]):int(off[r + 1])]
    return gs, gq, starts.astype(np.uint64), lens.astype(np.uint32)


def hip_gapped(gs, gq, starts, lens, aligned, bits, device):
    import torch
    with quack_amd.Accumulator(0, bits) as acc:
        if device:
            d = [torch.from_numpy(pad_for_device(gs)).cuda(), torch.from_numpy(pad_for_device(gq)).cuda(),
                 torch.from_numpy(starts.astype(np.int64)).cuda(), torch.from_numpy(lens.astype(np.int32)).cuda()]
            acc.submit_device_gapped(d[0], d[1], d[2], d[3], len(lens), len(gs), int(lens.max()), aligned=aligned)
        else:
            acc.submit_gapped(gs, gq, starts, lens, aligned=aligned)
        sd = acc.finish()
    return sd.bases, sd.number_of_sequences


@pytest.mark.parametrize("shape", ["long", "long_adapters", "short", "mixed_adapters"])
def test_gapped_and_cache_line_aligned_batches(shape):
    """reads separated by gaps (QK_BATCH_ALIGNED128: on cache lines — the layout
    the host feed gives long reads) count exactly like the packed batch"""
    ads = synth.synthetic_adapters()
    k = ob.kmers_from_seqs(ads) if "adapters" in shape else None
    bits = ob.kmers_to_bitset(k) if k is not None else None
    n, lo, hi = {"long": (600, 1000, 20000), "long_adapters": (400, 500, 9000), "short": (20000, 1, 150),
                 "mixed_adapters": (3000, 1, 2500)}[shape]
    seq, qual, off = synth.ragged(n, lo, hi, seed=hi)
    if k is not None:
        rng = np.random.default_rng(3)
        for r in rng.integers(0, n, n // 2):
            a, b = int(off[r]), int(off[r + 1])
            ad = np.frombuffer(ads[int(rng.integers(0, len(ads)))], np.uint8)
            at = a + int(rng.integers(0, max(1, b - a)))
            m = min(len(ad), b - at)
            seq[at:at + m] = ad[:m]
    want = ob.accumulate_batch(seq, qual, off, kmers=k)
    for align, aligned in ((128, True), (128, False), (0, False), (8, False)):
        gs, gq, starts, lens = gapped(seq, qual, off, align, seed=align)
        for device in (True, False):
            assert_same(hip_gapped(gs, gq, starts, lens, aligned, bits, device), want)


def test_gapped_edge_cases():
    """empty batch, a single read, zero-length reads between long ones, reads that end
    exactly on a tile / cache-line boundary"""
    import torch
    with quack_amd.Accumulator(0) as acc:                                  # nothing at all
        z8, z64, z32 = torch.zeros(16, dtype=torch.uint8).cuda(), torch.zeros(1, dtype=torch.int64).cuda(), torch.zeros(1, dtype=torch.int32).cuda()
        acc.submit_device_gapped(z8, z8, z64, z32, 0, 0, 0, aligned=True)
        acc.submit_gapped(np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(0, np.uint64), np.zeros(0, np.uint32), aligned=True)
        sd = acc.finish()
        assert sd.number_of_sequences == 0 and sd.bases.shape[0] == 0
    lens = np.array([0, 512, 0, 1024, 1, 128, 4096, 0, 511, 513, 2000, 0], dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    g = np.random.default_rng(8)
    seq = g.choice(np.frombuffer(b"ACGT", np.uint8), int(off[-1])).astype(np.uint8)
    qual = g.integers(33, 75, int(off[-1])).astype(np.uint8)
    want = ob.accumulate_batch(seq, qual, off)
    for align, aligned in ((128, True), (0, False)):
        gs, gq, starts, ln = gapped(seq, qual, off, align, seed=2)
        for device in (True, False):
            assert_same(hip_gapped(gs, gq, starts, ln, aligned, None, device), want)
    one = off[3:5] - off[3]
    assert_same(hip_gapped(seq[int(off[3]):int(off[4])], qual[int(off[3]):int(off[4])], np.zeros(1, np.uint64),
                           np.array([1024], np.uint32), True, None, True),
                ob.accumulate_batch(seq[int(off[3]):int(off[4])], qual[int(off[3]):int(off[4])], one.astype(np.uint64)))


def test_a_false_alignment_promise_is_detected():
    seq, qual, off = synth.ragged(300, 2000, 9000, seed=77)
    gs, gq, starts, lens = gapped(seq, qual, off, 0, seed=1)          # random gaps: not on cache lines
    with pytest.raises(quack_amd.HipUnavailable, match="128"):
        hip_gapped(gs, gq, starts, lens, True, None, device=True)       # found by the kernel, reported at finish
    with pytest.raises(quack_amd.HipUnavailable, match="128"):
        hip_gapped(gs, gq, starts, lens, True, None, device=False)      # found by the host check at commit


def test_two_host_threads_with_different_geometries():
    """the two mates of a pair are accumulated by two host threads (cli.c); their reads
    may differ in length, i.e. in kernel variant and LDS size — per-function state of
    the runtime (the dynamic-LDS ceiling) must not depend on who launched last"""
    import threading
    ads = synth.synthetic_adapters()
    k = ob.kmers_from_seqs(ads)
    jobs = [dict(L=300, n=3000, bits=ob.kmers_to_bitset(k), k=k), dict(L=36, n=20000, bits=None, k=None),
            dict(L=150, n=6000, bits=None, k=None), dict(L=72, n=9000, bits=ob.kmers_to_bitset(k), k=k)]
    out, err = {}, []

    def work(i, j):
        try:
            with quack_amd.Accumulator(0, j["bits"]) as acc:
                for rep in range(25):
                    seq, qual = synth.fixed(j["n"], j["L"], seed=100 * i + rep)
                    acc.submit_fixed(seq, qual, j["L"])
                sd = acc.finish()
            out[i] = (sd.bases, sd.number_of_sequences)
        except Exception as e:            # noqa: BLE001 - reported below
            err.append((i, e))

    for pair in ((0, 1), (2, 3)):       # two at a time, like the CLI's forward / reverse threads
        th = [threading.Thread(target=work, args=(i, jobs[i])) for i in pair]
        [t.start() for t in th]
        [t.join(timeout=150) for t in th]
        assert not any(t.is_alive() for t in th), "a host thread is stuck"
    assert not err, err
    for i, j in enumerate(jobs):
        seq = np.concatenate([synth.fixed(j["n"], j["L"], seed=100 * i + rep)[0] for rep in range(25)])
        qual = np.concatenate([synth.fixed(j["n"], j["L"], seed=100 * i + rep)[1] for rep in range(25)])
        assert_same(out[i], ob.accumulate_batch(seq, qual, read_len=j["L"], kmers=j["k"]))


def test_submits_from_several_streams_into_one_accumulator():
    """device-resident batches enqueued from three streams, un-synchronised, with
    pinned-slot batches in between: launches of one accumulator share its work
    queues and first-hit scratch, so the shim orders them itself"""
    import torch
    adapters = synth.synthetic_adapters()
    k = ob.kmers_from_seqs(adapters)
    parts = []
    for i in range(24):
        lo, hi = [(1, 150), (1000, 9000), (100, 700)][i % 3]
        seq, qual, off = synth.ragged([4000, 300, 1500][i % 3], lo, hi, seed=100 + i)
        rng = np.random.default_rng(i)
        for r in rng.integers(0, len(off) - 1, (len(off) - 1) // 3):     # adapters anywhere in a third of the reads
            a, b = int(off[r]), int(off[r + 1])
            ad = np.frombuffer(adapters[int(rng.integers(0, len(adapters)))], np.uint8)
            at = a + int(rng.integers(0, max(1, b - a)))
            m = min(len(ad), b - at)
            seq[at:at + m] = ad[:m]
        parts.append((seq, qual, off))
    streams = [torch.cuda.Stream(device=0) for _ in range(3)]
    keep = []
    with quack_amd.Accumulator(0, ob.kmers_to_bitset(k)) as acc:
        for i, (seq, qual, off) in enumerate(parts):
            if i % 4 == 3:
                acc.submit(seq, qual, off)          # through the pinned slots and their streams
                continue
            st = streams[i % 3]
            with torch.cuda.stream(st):
                d = [torch.from_numpy(pad_for_device(seq)).to("cuda:0", non_blocking=True),
                     torch.from_numpy(pad_for_device(qual)).to("cuda:0", non_blocking=True),
                     torch.from_numpy(off.astype(np.int64)).to("cuda:0", non_blocking=True)]
            keep.append(d)
            max_len = int(np.diff(off.astype(np.int64)).max())
            acc.submit_device(d[0], d[1], d[2], len(off) - 1, len(seq), max_len, stream=st.cuda_stream)
        sd = acc.finish()
    seq = np.concatenate([p[0] for p in parts])
    qual = np.concatenate([p[1] for p in parts])
    lens = np.concatenate([np.diff(p[2].astype(np.int64)) for p in parts])
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    assert_same((sd.bases, sd.number_of_sequences), ob.accumulate_batch(seq, qual, off, kmers=k))


def test_pinned_pipeline_many_small_batches(monkeypatch):
    monkeypatch.setenv("QUACK_HIP_BATCH_MB", "1")
    seq, qual, off = synth.ragged(60000, 50, 150, seed=21)       # ~6 MB -> >= 6 slot turnovers
    assert_same(hip_table(seq, qual, off), ob.accumulate_batch(seq, qual, off))
    seq, qual = synth.fixed(50000, 150, seed=22)
    assert_same(hip_table(seq, qual, read_len=150), ob.accumulate_batch(seq, qual, read_len=150))


def test_host_feed_pads_long_reads_onto_cache_lines(tmp_path, monkeypatch):
    """whole-file path on long reads: from the second batch on the tokenizer writes every
    read at a 128-byte boundary (QK_BATCH_ALIGNED128); same counters as the oracle, as
    the unpadded feed, and through three accumulators"""
    monkeypatch.setenv("QUACK_HIP_BATCH_MB", "2")
    seq, qual, off = synth.ragged(900, 800, 12000, seed=5)         # ~5.8 MB of reads: several batches
    fq = tmp_path / "long.fq"
    with open(fq, "wb") as f:
        for r in range(len(off) - 1):
            a, b = int(off[r]), int(off[r + 1])
            f.write(b"@r%d\n" % r + seq[a:b].tobytes() + b"\n+\n" + qual[a:b].tobytes() + b"\n")
    ads = synth.synthetic_adapters()
    k = ob.kmers_from_seqs(ads)
    for kmers, bits in ((None, None), (k, ob.kmers_to_bitset(k))):
        want = ob.accumulate_batch(seq, qual, off, kmers=kmers)
        sd = quack_amd.read_fastq(str(fq), bits)
        assert_same((sd.bases, sd.number_of_sequences), want)
        sd3 = quack_amd.read_fastq(str(fq), bits, devices=(0, 0, 0))
        assert_same((sd3.bases, sd3.number_of_sequences), want)
        monkeypatch.setenv("QUACK_NO_ALIGN", "1")
        sd = quack_amd.read_fastq(str(fq), bits)
        monkeypatch.delenv("QUACK_NO_ALIGN")
        assert_same((sd.bases, sd.number_of_sequences), want)


@pytest.mark.parametrize("cfg", [dict(threads=512, unroll=2, tile=96, wgs_per_cu=4),
                                 dict(threads=256, unroll=4, tile=64, wgs_per_cu=8),
                                 dict(threads=1024, unroll=1, tile=304, wgs_per_cu=1),
                                 dict(threads=1024, unroll=4, tile=8, wgs_per_cu=2)])
def test_launch_configurations_are_equivalent(cfg):
    seq, qual, off = synth.ragged(20000, 1, 320, seed=31)
    assert_same(hip_table(seq, qual, off, **cfg), ob.accumulate_batch(seq, qual, off))
    seq, qual = synth.fixed(20000, 150, seed=32)
    assert_same(hip_table(seq, qual, read_len=150, **cfg), ob.accumulate_batch(seq, qual, read_len=150))


@pytest.mark.parametrize("unroll,pipe", [("1", "2"), ("2", "2"), ("4", "2"), ("1", "1"), ("4", "1")])
def test_software_pipelined_variants_are_equivalent(monkeypatch, unroll, pipe):
    """QUACK_HIP_PIPE=2: the next step's loads are issued before the current
    step is consumed.  Automatic for fixed-length batches; forced here through
    every path (ragged staging, several tiles, fused adapters, short reads)"""
    monkeypatch.setenv("QUACK_HIP_TUNE", "unroll=%s,pipe=%s" % (unroll, pipe))
    ads = synth.synthetic_adapters()
    k = ob.kmers_from_seqs(ads)
    bits = ob.kmers_to_bitset(k)
    for L, n in ((150, 30011), (36, 50000), (300, 9000), (5, 1000)):
        seq, qual = synth.fixed(n, L, seed=L)
        assert_same(hip_table(seq, qual, read_len=L), ob.accumulate_batch(seq, qual, read_len=L))
        seq = synth.splice_adapters(seq, L, ads, seed=L + 1)
        assert_same(hip_table(seq, qual, read_len=L, kmers_bits=bits), ob.accumulate_batch(seq, qual, read_len=L, kmers=k))
    for n, lo, hi in ((30000, 1, 150), (300, 1000, 9000), (2500, 1, 40)):
        seq, qual, off = synth.ragged(n, lo, hi, seed=hi)
        assert_same(hip_table(seq, qual, off), ob.accumulate_batch(seq, qual, off))
        assert_same(hip_table(seq, qual, off, kmers_bits=bits), ob.accumulate_batch(seq, qual, off, kmers=k))


def test_reads_longer_than_a_slot_make_the_slots_grow(monkeypatch, tmp_path):
    """64 KiB slots and reads of up to 300 kb: the copying feed and the host feed both move to bigger
    slots (qk_accum_resize_slots) instead of failing — the reference has no length limit"""
    monkeypatch.setenv("QUACK_HIP_BATCH_KB", "64")
    seq, qual, off = synth.ragged(60, 100, 300000, seed=77)
    want = ob.accumulate_batch(seq, qual, off)
    assert_same(hip_table(seq, qual, off), want)
    L = 70000
    fs, fq = synth.fixed(5, L, seed=5)
    assert_same(hip_table(fs, fq, read_len=L), ob.accumulate_batch(fs, fq, read_len=L))
    path = tmp_path / "long.fq"
    with open(path, "wb") as f:
        for i in range(len(off) - 1):
            a, b = int(off[i]), int(off[i + 1])
            f.write(b"@r%d\n" % i + seq[a:b].tobytes() + b"\n+\n" + qual[a:b].tobytes() + b"\n")
    sd = quack_amd.read_fastq(str(path))
    assert_same((sd.bases, sd.number_of_sequences), want)


def test_timing_hooks_sample_every_nth_batch():
    """qk_accum_timing_enable(acc, N): HIP events around every Nth batch only; the tables are not affected"""
    seq, qual = synth.fixed(5000, 100, seed=8)       # (one batch per submit also with 1 MiB slots: tools/stress_gpu.sh)
    want = ob.accumulate_batch(np.tile(seq, 9), np.tile(qual, 9), read_len=100)
    with quack_amd.Accumulator(0) as acc:
        acc.timing(4)
        for _ in range(9):
            acc.submit_fixed(seq, qual, 100)
        acc.sync()
        hist_ms, batch_ms, launches = acc.timing_read_batch()
        assert launches == 3 and 0 < hist_ms <= batch_ms + 1e-6      # batches 0, 4, 8
        acc.timing(True)
        acc.submit_fixed(seq, qual, 100)
        acc.sync()
        assert acc.timing_read_batch()[2] == 1
        acc.timing(False)
        sd = acc.finish()
    want10 = ob.accumulate_batch(np.tile(seq, 10), np.tile(qual, 10), read_len=100)
    assert_same((sd.bases, sd.number_of_sequences), want10)
    del want


def test_empty_inputs():
    with quack_amd.Accumulator(0) as acc:
        acc.submit(np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(1, np.uint64))
        sd = acc.finish()
    assert sd.max_length == 0 and sd.number_of_sequences == 0
    # zero-length reads count as sequences only (quack.c:219 is undefined there)
    seq, qual, off = synth.ragged(100, 0, 0, seed=1)
    got = hip_table(seq, qual, off)
    assert got[1] == 100 and got[0].shape[0] == 0


def test_linearity_and_determinism():
    seq, qual, off = synth.ragged(40000, 1, 200, seed=41)
    whole = hip_table(seq, qual, off)
    again = hip_table(seq, qual, off)
    split = hip_table(seq, qual, off, chunks=7)
    assert_same(again, whole)
    assert_same(split, whole)


# ------------------------------------------------------------ reference goldens
QUACK = os.path.join(cases.ROOT, "quack_amd", "host", "quack")


@pytest.mark.parametrize("name,argv", cases.load(), ids=[c[0] for c in cases.load()])
def test_cli_is_a_drop_in(name, argv):
    """the C CLI on the GPU writes the reference binary's bytes"""
    r = subprocess.run([QUACK] + argv, capture_output=True, cwd=cases.inp(""), timeout=240)
    assert r.returncode == 0, r.stderr
    assert r.stderr == cases.golden_err(name)
    assert r.stdout == cases.golden_svg(name)


@pytest.mark.parametrize("fname,adapters", [("uniform100.fq.gz", None), ("adapter100.fq", "adapters.fa"),
                                            ("ragged100_2member.fq.gz", "adapters.fa.gz"),
                                            ("long40.fq.gz", "adapters.fa"), ("multiline100.fq", None),
                                            ("one_base.fq", None), ("kat.fq", "kat_adapter.fa")])
def test_read_fastq_mirror(fname, adapters):
    bits = quack_amd.read_adapters(cases.inp(adapters)) if adapters else None
    k = ob.kmers_from_file(cases.inp(adapters)) if adapters else None
    sd = quack_amd.read_fastq(cases.inp(fname), bits)
    assert_same((sd.bases, sd.number_of_sequences), ob.read_fastq(cases.inp(fname), k))


def test_sharded_accumulators_in_one_process():
    """batches dealt round-robin to three accumulators (one device here), merged
    by qk_accum_allreduce: equals one accumulator that saw every batch"""
    from quack_amd import api
    seq, qual, off = synth.ragged(30000, 1, 260, seed=61)
    k = ob.kmers_from_seqs(synth.synthetic_adapters())
    want = ob.accumulate_batch(seq, qual, off, kmers=k)
    accs = [quack_amd.Accumulator(0, ob.kmers_to_bitset(k)) for _ in range(3)]
    try:
        n = len(off) - 1
        cut = [n * i // 10 for i in range(11)]
        for b, (a, e) in enumerate(zip(cut, cut[1:])):
            lo, hi = int(off[a]), int(off[e])
            accs[b % 3].submit(seq[lo:hi], qual[lo:hi], off[a:e + 1] - off[a])
        api.allreduce(accs)
        for acc in accs:                       # every shard now holds the global table
            sd = acc.finish()
            assert_same((sd.bases, sd.number_of_sequences), want)
    finally:
        for acc in accs:
            acc.close()


def test_shards_with_different_longest_reads():
    """shards whose tables grew to different lengths (one saw only short
    reads, one a 3 kb read after doubling a few times) must agree on one geometry"""
    import torch
    from quack_amd import api, distributed as qd
    parts = [synth.ragged(4000, 1, 40, seed=71), synth.ragged(300, 100, 3000, seed=72),
             synth.ragged(2000, 150, 150, seed=73)]
    accs = [quack_amd.Accumulator(0, None, max_len_hint=8) for _ in parts]
    try:
        for acc, (s, q, o) in zip(accs, parts):
            acc.submit(s[:int(o[len(o) // 2])], q[:int(o[len(o) // 2])], o[:len(o) // 2 + 1])   # grow in two steps
            h = len(o) // 2
            acc.submit(s[int(o[h]):], q[int(o[h]):], o[h:] - o[h])
        assert len({a.table_words() for a in accs}) > 1
        # the torch.distributed path's per-rank steps, on one device
        words = max(a.table_words() for a in accs)
        tl = (words - 1) // 97
        bufs = []
        for a in accs:
            a.reserve(tl)
            assert a.table_words() == words
            b = torch.empty(words, dtype=torch.int64, device="cuda:0")
            a.export_table(b)
            bufs.append(b)
        total = sum(bufs)
        api.allreduce(accs)                      # in-process path
        seq = np.concatenate([p[0] for p in parts])
        qual = np.concatenate([p[1] for p in parts])
        lens = np.concatenate([np.diff(p[2].astype(np.int64)) for p in parts])
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        want = ob.accumulate_batch(seq, qual, off)
        sd = accs[0].finish()
        assert_same((sd.bases, sd.number_of_sequences), want)
        got, n_all = qd.bases_from_planar(total.cpu(), tl, want[0].shape[0])
        assert n_all == want[1]
        np.testing.assert_array_equal(got, want[0])
    finally:
        for a in accs:
            a.close()


@pytest.mark.parametrize("name", ["ragged100_adapters", "paired_adapters_named", "long40", "uniform100"])
def test_cli_sharded_over_three_accumulators(name):
    """QUACK_DEVICES=0,0,0: the CLI's multi-device path on one GPU"""
    argv = dict(cases.load())[name]
    r = subprocess.run([QUACK] + argv, capture_output=True, cwd=cases.inp(""),
                       env=dict(os.environ, QUACK_DEVICES="0,0,0", QUACK_HIP_BATCH_MB="1"), timeout=240)
    assert r.returncode == 0, r.stderr
    assert r.stdout == cases.golden_svg(name)


def test_device_table_roundtrip_and_single_rank_allreduce():
    """export -> all-reduce (world of one, RCCL) -> import leaves the table intact"""
    import torch
    import torch.distributed as dist
    from quack_amd import distributed as qd
    seq, qual, off = synth.ragged(20000, 1, 150, seed=51)
    want = ob.accumulate_batch(seq, qual, off)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    import socket
    with socket.socket() as sk:               # a free port, not a fixed one
        sk.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        with quack_amd.Accumulator(0) as acc:
            acc.submit(seq, qual, off)
            qd.allreduce_accumulator(acc)
            sd = acc.finish()
        assert_same((sd.bases, sd.number_of_sequences), want)
    finally:
        dist.destroy_process_group()


# ---------------------------------------------------------------- strided batches (fixed stride, own lengths)
def strided_from_ragged(seq, qual, off, stride, fill=None):
    """the reads of a packed batch laid out at a fixed stride; the bytes behind a read's last base are
    garbage on purpose (letters and scores that must never be counted)"""
    lens = np.diff(off.astype(np.int64))
    n = len(lens)
    rng = np.random.default_rng(5)
    s2 = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n * stride)].copy() if fill is None else np.full(n * stride, fill, np.uint8)
    q2 = (33 + rng.integers(0, 60, n * stride)).astype(np.uint8) if fill is None else np.full(n * stride, fill, np.uint8)
    idx = (np.arange(n, dtype=np.int64) * stride).repeat(lens) + (np.arange(int(off[-1]), dtype=np.int64) - off[:-1].astype(np.int64).repeat(lens))
    s2[idx] = seq
    q2[idx] = qual
    return s2, q2, lens.astype(np.uint32)


@pytest.mark.parametrize("n,lo,hi,stride,adapters", [(30000, 120, 150, 152, False), (30000, 120, 150, 152, True),
                                                     (20000, 0, 151, 152, False), (50000, 1, 36, 36, True),
                                                     (5000, 250, 300, 300, True), (3000, 500, 1000, 1000, False),
                                                     (2000, 400, 1200, 1200, True), (1, 7, 7, 8, False)])
def test_strided_batches(n, lo, hi, stride, adapters):
    """trimmed-Illumina form: equals the oracle on the same reads packed; host feed and device feed"""
    import torch
    ads = synth.synthetic_adapters()
    k = ob.kmers_from_seqs(ads) if adapters else None
    bits = ob.kmers_to_bitset(k) if adapters else None
    seq, qual, off = synth.ragged(n, lo, hi, seed=n + hi, q_lo=1, q_hi=60, alphabet=b"ACGTNacgt")
    seq = seq.copy()
    rng = np.random.default_rng(3)
    if adapters:
        for r in rng.integers(0, n, n // 4):            # an adapter somewhere in a quarter of the reads
            a, e = int(off[r]), int(off[r + 1])
            ad = np.frombuffer(ads[r % len(ads)], np.uint8)
            if e - a > 12:
                at = a + int(rng.integers(0, e - a - 10))
                m = min(len(ad), e - at)
                seq[at:at + m] = ad[:m]
    want = ob.accumulate_batch(seq, qual, off, kmers=k)
    s2, q2, lens = strided_from_ragged(seq, qual, off, stride)
    with quack_amd.Accumulator(0, bits) as acc:
        acc.submit_strided(s2, q2, lens, stride)                       # pinned slots
        d_s, d_q = torch.from_numpy(pad_for_device(s2)).cuda(), torch.from_numpy(pad_for_device(q2)).cuda()
        d_l = torch.from_numpy(lens.astype(np.int32)).cuda()
        acc.submit_device_strided(d_s, d_q, d_l, n, stride, int(lens.max()))
        # round 4: 0xFF behind every read and the producer's promise of it (QK_BATCH_NEUTRAL_PADS): the kernel without tail
        # masks, content[A] from the lengths counted in the loop; device-resident and through a pinned slot
        s3, q3, _ = strided_from_ragged(seq, qual, off, stride, fill=0xFF)
        d_s3, d_q3 = torch.from_numpy(pad_for_device(s3)).cuda(), torch.from_numpy(pad_for_device(q3)).cuda()
        acc.submit_device_strided(d_s3, d_q3, d_l, n, stride, int(lens.max()), neutral_pads=True)
        sd = acc.finish()
    assert sd.number_of_sequences == 3 * want[1]
    assert_same((sd.bases, want[1]), (3 * want[0], want[1]))


@pytest.mark.parametrize("stride,lo,hi", [(16, 0, 16), (20, 9, 19), (36, 0, 36), (52, 40, 50), (64, 0, 64), (76, 30, 75), (100, 60, 100), (152, 120, 150), (152, 0, 152), (252, 200, 250),
                                          (300, 280, 300), (352, 11, 352)])
def test_strided_rows_with_adapters_sixteen_positions_per_lane(stride, lo, hi, monkeypatch):
    """Round 5: trimmed reads WITH the adapter scan (quack.c:206-217 runs on every read whatever its length) — a strided batch
    whose pads are 0xFF takes the 16-positions-per-lane kernel, rows of several reads where that fills the lanes: every group
    size, reads of every length from empty to the whole stride (70 % of them the longest), an odd read count (the reads that do
    not fill a last row are a launch of their own), adapters spliced in up to the last base; device-resident and through the
    pinned slots; against the oracle on the same reads packed."""
    import torch
    rng = np.random.default_rng(stride * 1000 + lo)
    n = 40001
    ads = synth.synthetic_adapters()
    k = ob.kmers_from_seqs(ads)
    bits = ob.kmers_to_bitset(k)
    lens = rng.integers(lo, hi + 1, n)
    lens[rng.random(n) < 0.7] = hi
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    total = int(off[-1])
    seq = np.frombuffer(b"ACGTNacgt", np.uint8)[rng.integers(0, 9, total)].copy()
    qual = (33 + rng.integers(1, 61, total)).astype(np.uint8)
    for r in rng.integers(0, n, n // 3):
        a, e = int(off[r]), int(off[r + 1])
        if e - a > 12:
            ad = np.frombuffer(ads[r % len(ads)], np.uint8)
            at = a + int(rng.integers(0, e - a))
            m = min(len(ad), e - at)
            seq[at:at + m] = ad[:m]
    want = ob.accumulate_batch(seq, qual, off, kmers=k)
    assert want[0][:, 96].sum() > 100
    s3, q3, l3 = strided_from_ragged(seq, qual, off, stride, fill=0xFF)
    d_s3, d_q3 = torch.from_numpy(pad_for_device(s3)).cuda(), torch.from_numpy(pad_for_device(q3)).cuda()
    d_l = torch.from_numpy(l3.astype(np.int32)).cuda()
    torch.cuda.synchronize()
    for env in ({}, {"QUACK_HIP_TUNE": "group=1"}, {"QUACK_HIP_TUNE": "group=2"}, {"QUACK_HIP_TUNE": "group=3"}, {"QUACK_HIP_TUNE": "group=4"}, {"QUACK_HIP_NO_W16": "1"}):
        for kk, v in env.items():
            monkeypatch.setenv(kk, v)
        with quack_amd.Accumulator(0, bits) as acc:
            acc.submit_device_strided(d_s3, d_q3, d_l, n, stride, int(l3.max()), neutral_pads=True)
            acc.submit_strided(s3, q3, l3, stride)       # (pinned slots: the library writes the pads itself)
            sd = acc.finish()
        for kk in env:
            monkeypatch.delenv(kk)
        assert sd.number_of_sequences == 2 * want[1], env
        np.testing.assert_array_equal(sd.bases, 2 * want[0], err_msg=str(env))


def test_neutral_pads_promise_is_checked_where_the_host_has_the_bytes():
    """qk_accum_commit_strided_flags(QK_BATCH_NEUTRAL_PADS) looks at the first and the last pad byte of every read"""
    import ctypes
    from quack_amd import _capi
    n, stride = 1000, 152
    lens = np.full(n, 140, np.uint32)
    with quack_amd.Accumulator(0) as acc:
        for spoil in (None, 140, 151):
            hs, hq, _ = acc.acquire()
            hl = ctypes.POINTER(ctypes.c_uint32)()
            assert acc._L.qk_accum_slot_lengths(acc._h, ctypes.byref(hl)) == 0
            np.ctypeslib.as_array(hl, shape=(n,))[:] = lens
            hs[:n * stride] = 0xFF
            hq[:n * stride] = 0xFF
            hs[:n * stride].reshape(n, stride)[:, :140] = ord("C")
            hq[:n * stride].reshape(n, stride)[:, :140] = ord("5")
            if spoil is not None:
                hq[500 * stride + spoil] = ord("I")
            rc = acc._L.qk_accum_commit_strided_flags(acc._h, n, stride, _capi.QK_BATCH_NEUTRAL_PADS)
            if spoil is None:
                assert rc == 0
            else:
                assert rc != 0 and b"0xFF" in acc._L.qk_last_error()
                acc._L.qk_accum_commit(acc._h, 0, 0, 0, 0)      # give the slot back
        sd = acc.finish()
    assert sd.number_of_sequences == n and sd.bases[:140, 20].sum() == 140 * n and sd.bases[:140, 93].sum() == 140 * n
    assert sd.bases[139, 95] == n and sd.bases.shape[0] == 140


def test_neutral_pads_promise_on_a_device_batch_can_be_verified(monkeypatch):
    """a device-resident batch is taken at its word; with QUACK_HIP_CHECK_PADS=1 the shim checks every pad byte first and a
    violation fails the next sync (and without the check a broken promise is counted — which is what the check is for)"""
    import torch
    n, stride = 5000, 152
    rng = np.random.default_rng(8)
    lens = rng.integers(100, 151, n)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(off[-1]))]
    qual = (33 + rng.integers(0, 42, int(off[-1]))).astype(np.uint8)
    want = ob.accumulate_batch(seq, qual, off)
    s3, q3, l3 = strided_from_ragged(seq, qual, off, stride, fill=0xFF)
    d_l = torch.from_numpy(l3.astype(np.int32)).cuda()
    monkeypatch.setenv("QUACK_HIP_CHECK_PADS", "1")
    with quack_amd.Accumulator(0) as acc:
        acc.submit_device_strided(torch.from_numpy(pad_for_device(s3)).cuda(), torch.from_numpy(pad_for_device(q3)).cuda(), d_l, n, stride, 150, neutral_pads=True)
        sd = acc.finish()
    assert_same((sd.bases, sd.number_of_sequences), want)
    q3 = q3.copy()
    q3[3000 * stride + 151] = ord("I")          # one pad byte that is not 0xFF
    with quack_amd.Accumulator(0) as acc:
        acc.submit_device_strided(torch.from_numpy(pad_for_device(s3)).cuda(), torch.from_numpy(pad_for_device(q3)).cuda(), d_l, n, stride, 150, neutral_pads=True)
        with pytest.raises(quack_amd.HipUnavailable, match="0xFF"):
            acc.sync()


def test_strided_rejects_bad_geometry():
    with quack_amd.Accumulator(0) as acc:
        with pytest.raises(quack_amd.HipUnavailable):
            acc.submit_strided(np.zeros(30, np.uint8), np.zeros(30, np.uint8), np.array([3, 3, 3], np.uint32), 10)   # stride % 4
        with pytest.raises(quack_amd.HipUnavailable):
            acc.submit_strided(np.zeros(24, np.uint8), np.zeros(24, np.uint8), np.array([3, 9, 3], np.uint32), 8)    # read > stride


@pytest.mark.parametrize("cfg,env", [(dict(threads=512), {}), (dict(unroll=2), {}), (dict(threads=256, unroll=2, tile=64), {}),
                                     ({}, {"QUACK_HIP_TUNE": "pipe=1"}), ({}, {"QUACK_HIP_NO_ALIGN4": "1"}),
                                     ({}, {"QUACK_HIP_TUNE": "adapt_pd=3"}), ({}, {"QUACK_HIP_UNFUSED_ADAPTERS": "1"})])
def test_strided_batches_under_tuning_overrides(cfg, env, monkeypatch):
    """the strided kernel variant exists for the planner's own geometry; under an override the same reads run as
    gapped batches (starts written on the device) — round 2 failed with QK_EINVAL here, and so did the CLI on
    any trimmed FASTQ with such a knob set"""
    import torch
    for kk, v in env.items():
        monkeypatch.setenv(kk, v)
    ads = synth.synthetic_adapters()
    k = ob.kmers_from_seqs(ads)
    n, stride = 20000, 152
    seq, qual, off = synth.ragged(n, 100, 150, seed=77, q_lo=1, q_hi=60, alphabet=b"ACGTNacgt")
    seq = seq.copy()
    rng = np.random.default_rng(5)
    for r in rng.integers(0, n, n // 4):                # an adapter somewhere in a quarter of the reads
        a, e = int(off[r]), int(off[r + 1])
        ad = np.frombuffer(ads[r % len(ads)], np.uint8)
        at = a + int(rng.integers(0, e - a - 10))
        m = min(len(ad), e - at)
        seq[at:at + m] = ad[:m]
    s2, q2, lens = strided_from_ragged(seq, qual, off, stride)
    for kmers, bits in ((None, None), (k, ob.kmers_to_bitset(k))):
        want = ob.accumulate_batch(seq, qual, off, kmers=kmers)
        with quack_amd.Accumulator(0, bits, experiment=True if cfg else None) as acc:
            if cfg:
                acc.configure(**cfg)
            acc.submit_strided(s2, q2, lens, stride)
            d_s, d_q = torch.from_numpy(pad_for_device(s2)).cuda(), torch.from_numpy(pad_for_device(q2)).cuda()
            d_l = torch.from_numpy(lens.astype(np.int32)).cuda()
            acc.submit_device_strided(d_s, d_q, d_l, n, stride, int(lens.max()))
            sd = acc.finish()
        assert sd.number_of_sequences == 2 * want[1]
        assert_same((sd.bases, want[1]), (2 * want[0], want[1]))


def test_strided_fallback_in_several_chunks(monkeypatch):
    """a strided batch restated as gapped ones is cut where a gapped batch would pass 2 GiB; here at 3000 reads"""
    import torch
    monkeypatch.setenv("QUACK_HIP_TUNE", "strided_chunk_reads=3000")
    ads = synth.synthetic_adapters()
    k = ob.kmers_from_seqs(ads)
    n, stride = 10000, 152
    seq, qual, off = synth.ragged(n, 90, 150, seed=78, q_lo=1, q_hi=60)
    s2, q2, lens = strided_from_ragged(seq, qual, off, stride)
    want = ob.accumulate_batch(seq, qual, off, kmers=k)
    with quack_amd.Accumulator(0, ob.kmers_to_bitset(k), experiment=True) as acc:
        acc.configure(unroll=2)
        d_s, d_q = torch.from_numpy(pad_for_device(s2)).cuda(), torch.from_numpy(pad_for_device(q2)).cuda()
        acc.submit_device_strided(d_s, d_q, torch.from_numpy(lens.astype(np.int32)).cuda(), n, stride, int(lens.max()))
        sd = acc.finish()
    assert_same((sd.bases, sd.number_of_sequences), want)


def test_device_side_length_beyond_the_declared_maximum_is_reported():
    """lengths[] of a device-resident strided batch cannot be vetted by the host: a read longer than the caller
    declared must fail the next sync, and must not write outside the table"""
    import torch
    n, stride = 5000, 152
    s2 = np.full(n * stride, ord("A"), np.uint8)
    q2 = np.full(n * stride, 70, np.uint8)
    lens = np.full(n, 100, np.int32)
    lens[1234] = 140          # within the stride, beyond max_len = 100
    with quack_amd.Accumulator(0, max_len_hint=100) as acc:
        d_s, d_q = torch.from_numpy(pad_for_device(s2)).cuda(), torch.from_numpy(pad_for_device(q2)).cuda()
        acc.submit_device_strided(d_s, d_q, torch.from_numpy(lens).cuda(), n, stride, 100)
        with pytest.raises(quack_amd.HipUnavailable, match="longer than"):
            acc.sync()
    lens[1234] = 100000       # beyond everything
    with quack_amd.Accumulator(0, max_len_hint=100) as acc:
        d_s, d_q = torch.from_numpy(pad_for_device(s2)).cuda(), torch.from_numpy(pad_for_device(q2)).cuda()
        acc.submit_device_strided(d_s, d_q, torch.from_numpy(lens).cuda(), n, stride, 100)
        with pytest.raises(quack_amd.HipUnavailable, match="longer than"):
            acc.sync()


def test_host_feed_lays_trimmed_reads_out_at_a_fixed_stride(tmp_path, monkeypatch):
    """whole-file path on short reads of nearly one length: from the second batch on the tokenizer writes
    them at a fixed stride (qk_accum_commit_strided); same counters as the oracle, as the packed feed
    (QUACK_NO_STRIDE=1), and through three accumulators; with a read that widens the stride and one that
    ends the mode"""
    monkeypatch.setenv("QUACK_HIP_BATCH_MB", "1")
    rng = np.random.default_rng(12)
    n = 60000
    lens = np.where(rng.random(n) < 0.7, 150, rng.integers(100, 150, n))
    lens[20000], lens[40000], lens[50000] = 163, 2000, 0
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    total = int(off[-1])
    seq = np.frombuffer(b"ACGTNacgt", np.uint8)[rng.integers(0, 9, total)]
    qual = (33 + rng.integers(0, 42, total)).astype(np.uint8)
    fq = tmp_path / "trimmed.fq"
    with open(fq, "wb") as f:
        for r in range(n):
            a, b = int(off[r]), int(off[r + 1])
            f.write(b"@r%d\n" % r + seq[a:b].tobytes() + b"\n+\n" + qual[a:b].tobytes() + b"\n")
    ads = synth.synthetic_adapters()
    k = ob.kmers_from_seqs(ads)
    for kmers, bits in ((None, None), (k, ob.kmers_to_bitset(k))):
        want = ob.read_fastq(str(fq), kmers)
        assert want[1] == n
        sd = quack_amd.read_fastq(str(fq), bits)
        assert_same((sd.bases, sd.number_of_sequences), want)
        sd3 = quack_amd.read_fastq(str(fq), bits, devices=(0, 0, 0))
        assert_same((sd3.bases, sd3.number_of_sequences), want)
        monkeypatch.setenv("QUACK_NO_STRIDE", "1")
        sd = quack_amd.read_fastq(str(fq), bits)
        monkeypatch.delenv("QUACK_NO_STRIDE")
        assert_same((sd.bases, sd.number_of_sequences), want)


def test_host_feed_lays_uniform_150bp_reads_out_at_a_padded_stride(tmp_path, monkeypatch):
    """whole-file path on the metric's own input with the adapter scan (round 4): uniform 150 bp reads — from the second batch
    on, and for the batches parsed while the accumulators start, the tokenizer writes them 152 bytes apart and commits padded
    fixed-length batches (rows of two reads, 16 positions per lane); a stretch of trimmed reads in the middle goes on as strided
    batches with 0xFF pads, a stretch of very ragged ones as packed batches.  Same counters as the oracle's read_fastq, as the
    packed feed (QUACK_NO_STRIDE=1), through three accumulators, as a .gz, and byte-identical SVG from the CLI"""
    monkeypatch.setenv("QUACK_HIP_BATCH_MB", "1")
    rng = np.random.default_rng(15)
    lens = np.concatenate([np.full(30000, 150), np.where(rng.random(12000) < 0.6, 150, rng.integers(125, 150, 12000)),
                           rng.integers(30, 151, 8000), np.full(20001, 150)])
    n = len(lens)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    total = int(off[-1])
    ads = synth.synthetic_adapters()
    seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, total)].copy()
    qual = (33 + rng.integers(2, 42, total)).astype(np.uint8)
    for r in range(0, n, 4):
        a, e = int(off[r]), int(off[r + 1])
        ad = np.frombuffer(ads[r % len(ads)], np.uint8)
        at = a + int(rng.integers(0, e - a - 11))
        m = min(len(ad), e - at)
        seq[at:at + m] = ad[:m]
    fq = tmp_path / "u150.fq"
    with open(fq, "wb") as f:
        for r in range(n):
            a, b = int(off[r]), int(off[r + 1])
            f.write(b"@r%d\n" % r + seq[a:b].tobytes() + b"\n+\n" + qual[a:b].tobytes() + b"\n")
    fa = tmp_path / "ads.fa"
    with open(fa, "wb") as f:
        for i, a in enumerate(ads):
            f.write(b">a%d\n" % i + a + b"\n")
    k = ob.kmers_from_file(str(fa))
    bits = quack_amd.read_adapters(str(fa))
    assert np.array_equal(bits, ob.kmers_to_bitset(k))
    want = ob.read_fastq(str(fq), k)
    assert want[1] == n and want[0][:, 96].sum() > 10000
    subprocess.check_call(["gzip", "-k", "-1", str(fq)])
    for env in ({}, {"QUACK_NO_STRIDE": "1"}, {"QUACK_NO_EARLY": "1"}, {"QUACK_HIP_NO_NEUTRAL": "1"}):
        for kk, v in env.items():
            monkeypatch.setenv(kk, v)
        sd = quack_amd.read_fastq(str(fq), bits)
        assert_same((sd.bases, sd.number_of_sequences), want)
        for kk in env:
            monkeypatch.delenv(kk)
    sd3 = quack_amd.read_fastq(str(fq) + ".gz", bits, devices=(0, 0, 0))
    assert_same((sd3.bases, sd3.number_of_sequences), want)
    quack = os.path.join(cases.ROOT, "quack_amd", "host", "quack")
    svgs = [subprocess.run([quack, "-u", str(fq), "-a", str(fa)], capture_output=True, env=dict(os.environ, **e), timeout=300)
            for e in ({}, {"QUACK_NO_STRIDE": "1", "QUACK_HIP_UNFUSED_ADAPTERS": "1"})]
    assert svgs[0].returncode == 0 and svgs[0].stdout == svgs[1].stdout and svgs[0].stdout.startswith(b"<svg")


# ---------------------------------------------------------------- long ragged reads: reach sort + static split
@pytest.mark.parametrize("n,lo,hi,adapters", [(50, 1000, 20000, False), (300, 5000, 5000, False), (3000, 0, 9000, True),
                                              (700, 2049, 2049, False), (40000, 1500, 3000, False), (257, 511, 40000, True),
                                              (300000, 600, 2600, False)])
def test_long_ragged_reads_static_split(n, lo, hi, adapters):
    """4..64 tiles: the reads are sorted by reach and the read-tiles are cut into one share per workgroup —
    fewer reads than workgroups, equal lengths (every tile the same work), zero-length reads in between,
    a length one past a tile boundary, more reads than one item may hold, more reads than the pre-pass has
    threads, packed and on cache lines"""
    import torch
    ads = synth.synthetic_adapters()
    k = ob.kmers_from_seqs(ads) if adapters else None
    bits = ob.kmers_to_bitset(k) if adapters else None
    seq, qual, off = synth.ragged(n, lo, hi, seed=n + hi, q_lo=1, q_hi=60, alphabet=b"ACGTNacgt")
    seq = seq.copy()
    if adapters:
        rng = np.random.default_rng(8)
        for r in rng.integers(0, n, n // 3):
            a, e = int(off[r]), int(off[r + 1])
            if e - a > 100:
                ad = np.frombuffer(ads[r % len(ads)], np.uint8)
                at = a + int(rng.integers(0, e - a - 50))
                seq[at:at + len(ad)] = ad[:max(0, min(len(ad), e - at))]
    want = ob.accumulate_batch(seq, qual, off, kmers=k)
    assert_same(hip_table(seq, qual, off, kmers_bits=bits), want)                  # packed, through the pinned slots
    lens = np.diff(off.astype(np.int64))
    starts = np.concatenate([[0], np.cumsum((lens + 127) // 128 * 128)])[:-1].astype(np.uint64)
    extent = int(starts[-1] + lens[-1]) if n else 0
    s2 = np.full(extent, ord("T"), np.uint8)
    q2 = np.full(extent, 70, np.uint8)
    for r in range(n):
        a, l = int(starts[r]), int(lens[r])
        s2[a:a + l] = seq[int(off[r]):int(off[r]) + l]
        q2[a:a + l] = qual[int(off[r]):int(off[r]) + l]
    with quack_amd.Accumulator(0, bits) as acc:                                    # on cache lines, device-resident, twice
        d_s, d_q = torch.from_numpy(pad_for_device(s2)).cuda(), torch.from_numpy(pad_for_device(q2)).cuda()
        d_st, d_l = torch.from_numpy(starts.astype(np.int64)).cuda(), torch.from_numpy(lens.astype(np.int32)).cuda()
        for _ in range(2):
            acc.submit_device_gapped(d_s, d_q, d_st, d_l, n, extent, int(lens.max()), aligned=True)
        sd = acc.finish()
    assert sd.number_of_sequences == 2 * want[1]
    assert_same((sd.bases, want[1]), (2 * want[0], want[1]))

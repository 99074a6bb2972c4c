"""BASELINE.json configurations at full size on one MI355X, device-resident
(generated on the GPU), checked through size-independent properties and exactly against the oracle —
every read of configs 2, 3, 5, of config 4's per-GPU share and of the trimmed-reads
workload (the oracle manages ~0.25 Gbases/s per core; it runs on up to 32 threads,
one table each, summed)."""
import numpy as np
import pytest

import oracle_binding as ob
import synth
import quack_amd

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def device_fixed(n, L, seed, q_lo=2, q_hi=41):
    g = torch.Generator(device="cuda").manual_seed(seed)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
    seq = torch.zeros(n * L + 16, dtype=torch.uint8, device="cuda")
    qual = torch.zeros(n * L + 16, dtype=torch.uint8, device="cuda")
    step = 1 << 28
    for a in range(0, n * L, step):
        b = min(n * L, a + step)
        seq[a:b] = lut[torch.randint(0, 4, (b - a,), generator=g, device="cuda")]
        qual[a:b] = (33 + torch.randint(q_lo, q_hi + 1, (b - a,), generator=g, device="cuda")).to(torch.uint8)
    return seq, qual


def run_device(seq, qual, off, n, total, max_len, bits=None, passes=1):
    torch.cuda.synchronize()     # (the batch was made on torch's stream; the accumulator's own stream waits for nobody)
    with quack_amd.Accumulator(0, bits) as acc:
        for _ in range(passes):
            acc.submit_device(seq, qual, off, n, total, max_len)
        return acc.finish()


def test_config2_10M_x_150_exact_and_properties():
    n, L = 10_000_000, 150
    seq, qual = device_fixed(n, L, seed=2)
    sd = run_device(seq, qual, None, n, n * L, L)
    b = sd.bases.astype(np.int64)
    assert sd.number_of_sequences == n and sd.max_length == L
    assert (b[:, :91].sum(axis=1) == n).all()          # every base has one valid score
    assert (b[:, 91:95].sum(axis=1) == n).all()        # ... and one content bin
    assert b[:, 95].sum() == n and b[L - 1, 95] == n   # one length per read
    assert b[10, 96] == n and b[:, 96].sum() == n      # kmers == NULL: bases[10].kmer_count++ (quack.c:215)
    assert b[:, :2].sum() == 0 and b[:, 42:91].sum() == 0   # Q in [2,41] only
    # idempotence/linearity: a second pass doubles every counter
    sd2 = run_device(seq, qual, None, n, n * L, L, passes=2)
    np.testing.assert_array_equal(sd2.bases, 2 * sd.bases)
    # exact, against the oracle on the same bytes
    want, wn = ob.accumulate_batch_threads(seq[:n * L].cpu().numpy(), qual[:n * L].cpu().numpy(), read_len=L)
    assert wn == n
    np.testing.assert_array_equal(sd.bases, want)


def test_config3_10M_x_300_adapters_exact_on_every_read():
    """the bench's own config-3 batch (bench.make_batch: 25 % of the reads carry a spliced adapter, seed 3): properties,
    and every counter against the oracle over ALL 10M reads (round 2 checked 2M and took the rest by additivity)"""
    import bench
    w = dict(bench.WORKLOADS["cfg3"])
    n, L = w["n"], w["L"]
    bits, ads = bench.synthetic_adapter_bits(np)
    k = ob.kmers_from_seqs([bytes(a) for a in ads])
    assert np.array_equal(ob.kmers_to_bitset(k), bits)          # the product's read_adapters rule == the oracle's
    b = bench.make_batch(torch, np, w, seed=3, device=torch.device("cuda", 0), ads=ads)
    assert 0.2 * n < b["spliced"] < 0.3 * n
    sd = run_device(b["seq"], b["qual"], None, n, n * L, L, bits)
    t = sd.bases.astype(np.int64)
    assert sd.number_of_sequences == n and sd.max_length == L
    assert (t[:, :91].sum(axis=1) == n).all() and (t[:, 91:95].sum(axis=1) == n).all()
    assert t[L - 1, 95] == n
    assert 0.2 * n < t[:, 96].sum() <= n and t[:10, 96].sum() == 0     # at most one first hit per read, never before 10
    want, wn = ob.accumulate_batch_threads(b["seq"][:n * L].cpu().numpy(), b["qual"][:n * L].cpu().numpy(), read_len=L, kmers=k)
    assert wn == n
    np.testing.assert_array_equal(sd.bases, want)


def test_config3_at_150bp_padded_exact_on_every_read(monkeypatch):
    """the metric's own read length on config 3's path (round 4): the bench's cfg3_150 batch — 10M x 150 + adapters, 25 % of
    the reads spliced, reads 152 bytes apart as the host feed lays them out — every counter against the oracle over ALL
    reads; and the same reads as single-read rows (QUACK_HIP_NO_GROUP) and with the smallest first-hit ring"""
    import bench
    w = dict(bench.WORKLOADS["cfg3_150"])
    n, L, S = w["n"], w["L"], w["pad"]
    bits, ads = bench.synthetic_adapter_bits(np)
    k = ob.kmers_from_seqs([bytes(a) for a in ads])
    b = bench.make_batch(torch, np, w, seed=5, device=torch.device("cuda", 0), ads=ads)
    assert 0.2 * n < b["spliced"] < 0.3 * n and b["pad"] == S
    hs, hq, off, m, bases = bench.host_sample(np, b, w, 1 << 62)
    assert off is None and m == n and bases == n * L
    want, wn = ob.accumulate_batch_threads(hs, hq, read_len=L, kmers=k)
    assert wn == n and 0.2 * n < want[:, 96].sum() <= n
    torch.cuda.synchronize()
    for env in ({}, {"QUACK_HIP_NO_GROUP": "1"}, {"QUACK_HIP_TUNE": "small_ring=1"}):
        for kk, v in env.items():
            monkeypatch.setenv(kk, v)
        with quack_amd.Accumulator(0, bits) as acc:
            acc.submit_device_padded(b["seq"], b["qual"], n, L, S)
            sd = acc.finish()
        for kk in env:
            monkeypatch.delenv(kk)
        assert sd.number_of_sequences == n and sd.max_length == L, env
        np.testing.assert_array_equal(sd.bases, want, err_msg=str(env))


@pytest.mark.parametrize("workload", ["trimmed", "trimmed_adapters"])
def test_trimmed_10M_strided_exact_on_every_read(workload):
    """the bench's trimmed-reads batches (70 % full length, the rest 120-149, stride 152 + lengths[], 0xFF behind every read;
    `trimmed_adapters`: the adapter table loaded, a quarter of the reads with a spliced adapter) against the oracle on the same
    reads packed — submitted the way bench.py submits them (QK_BATCH_NEUTRAL_PADS: the kernel variants the bench line times)
    AND without the promise (the variants that mask the tails)"""
    import bench
    w = dict(bench.WORKLOADS[workload])
    bits, ads = bench.synthetic_adapter_bits(np) if w["adapters"] else (None, None)
    b = bench.make_batch(torch, np, w, seed=7, device=torch.device("cuda", 0), ads=ads)
    torch.cuda.synchronize()
    hs, hq, off, m, bases = bench.host_sample(np, b, w, 1 << 62)
    assert m == b["n"] and bases == b["total"]
    want, wn = ob.accumulate_batch_threads(hs, hq, off, kmers=ob.kmers_from_seqs([bytes(a) for a in ads]) if ads is not None else None)
    assert wn == b["n"]
    for neutral in (True, False):
        with quack_amd.Accumulator(0, bits) as acc:
            acc.submit_device_strided(b["seq"], b["qual"], b["d_len"], b["n"], b["stride"], b["max_len"], neutral_pads=neutral)
            sd = acc.finish()
        assert sd.number_of_sequences == wn, neutral
        np.testing.assert_array_equal(sd.bases, want, err_msg="neutral_pads=%s" % neutral)
    if w["adapters"]:
        assert want[:, 96].sum() > 1_000_000


def test_config4_full_per_gpu_share_each_mate_exact():
    """config 4 (paired 2 x 50M x 150 bp over 8 GPUs) as far as one GPU goes: its full per-GPU share — 2 x 6.25M
    reads, R2 qualities in [2,30], seeds 4 / 5 — through the bench's path: two accumulators fed from ONE stream,
    three steps; each mate against the oracle (x3: the counters are linear in the passes).  quack.c:911-921."""
    import bench
    w = dict(bench.WORKLOADS["cfg4"])
    n, L = w["n"], w["L"]
    dev = torch.device("cuda", 0)
    f = bench.make_batch(torch, np, w, seed=4, device=dev)
    r = bench.make_batch(torch, np, w, seed=5, device=dev, q_hi_override=30)
    side = torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    with quack_amd.Accumulator(0, None, max_len_hint=L) as fa, quack_amd.Accumulator(0, None, max_len_hint=L) as ra:
        for _ in range(3):
            fa.submit_device(f["seq"], f["qual"], None, n, n * L, L, stream=side.cuda_stream)
            ra.submit_device(r["seq"], r["qual"], None, n, n * L, L, stream=side.cuda_stream)
        fsd, rsd = fa.finish(), ra.finish()
    for sd, b in ((fsd, f), (rsd, r)):
        want, wn = ob.accumulate_batch_threads(b["seq"][:n * L].cpu().numpy(), b["qual"][:n * L].cpu().numpy(), read_len=L)
        assert wn == n and sd.number_of_sequences == 3 * n
        np.testing.assert_array_equal(sd.bases, 3 * want)
    assert rsd.bases[:, 31:91].sum() == 0 and fsd.bases[:, 31:42].sum() > 0      # R2's qualities stop at Q30


def test_config5_ragged_1kb_to_20kb_exact():
    rng = np.random.default_rng(6)
    n = 143_000
    lens = rng.integers(1000, 20001, n)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    total = int(off[-1])
    g = torch.Generator(device="cuda").manual_seed(6)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
    seq = torch.zeros(total + 16, dtype=torch.uint8, device="cuda")
    qual = torch.zeros(total + 16, dtype=torch.uint8, device="cuda")
    seq[:total] = lut[torch.randint(0, 4, (total,), generator=g, device="cuda")]
    qual[:total] = (33 + torch.randint(1, 61, (total,), generator=g, device="cuda")).to(torch.uint8)
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    sd = run_device(seq, qual, d_off, n, total, int(lens.max()))
    assert sd.number_of_sequences == n and sd.max_length == int(lens.max())
    b = sd.bases.astype(np.int64)
    cover = n - np.searchsorted(np.sort(lens), np.arange(sd.max_length), side="right")   # reads longer than pos
    assert (b[:, :91].sum(axis=1) == cover).all() and (b[:, 91:95].sum(axis=1) == cover).all()
    assert (b[:, 95] == np.bincount(lens - 1, minlength=sd.max_length)).all()
    want, _ = ob.accumulate_batch_threads(seq[:total].cpu().numpy(), qual[:total].cpu().numpy(), off)
    np.testing.assert_array_equal(sd.bases, want)


def test_config4_shape_two_independent_accumulators():
    """paired = two independent accumulations (quack.c:911-921); sharding the
    batches of each mate over accumulators and summing equals one pass"""
    n, L = 2_000_000, 150
    f_seq, f_qual = device_fixed(n, L, seed=4)
    r_seq, r_qual = device_fixed(n, L, seed=5, q_hi=30)
    whole = [run_device(s, q, None, n, n * L, L) for s, q in ((f_seq, f_qual), (r_seq, r_qual))]
    assert not np.array_equal(whole[0].bases, whole[1].bases)
    shards = 8
    per = n // shards
    for (s, q), w in zip(((f_seq, f_qual), (r_seq, r_qual)), whole):
        acc = np.zeros_like(w.bases)
        for k in range(shards):                      # what 8 ranks would each compute
            a = k * per * L
            part = run_device(s[a:], q[a:], None, per, per * L, L)
            acc += part.bases
        np.testing.assert_array_equal(acc, w.bases)  # what the all-reduce(SUM) yields

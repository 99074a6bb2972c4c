"""Worst-case inputs at full size, and the guards that ordinary batches never reach ("missing" 4).
quack.c:199-205 on constant quality / constant base — NovaSeq's four quality bins, poly-G tails — is where the kernels'
narrow counters are at their limits: u16 pairs in LDS (at most 65,535 reads per workgroup between flushes:
qk::kMaxReadsPerSlice), SWAR byte counters in registers (spilled every <= 255 events), same-address LDS atomics, and the
32-bit side table the histogram kernels flush into (folded into the u64 table before 2^32 reads could have gone in:
qk_shim.hip enqueue_batch).  Every cell is checked — against a closed form where the batch has one, against the oracle on
ALL reads otherwise — also with the side table switched off (QUACK_HIP_NO_TABLE32)."""
import numpy as np
import pytest

import oracle_binding as ob
import quack_amd

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

Q = lambda c: ord(c) - 33


def constant_batch(n, L, base=b"G", qual=b"I"):
    seq = torch.full((n * L + 16,), base[0], dtype=torch.uint8, device="cuda")
    ql = torch.full((n * L + 16,), qual[0], dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()     # (torch fills on ITS stream; the accumulator launches on its own, which waits for nobody)
    return seq, ql


def closed_form(n, L, score, content_row, adapters):
    want = np.zeros((L, 97), np.uint64)
    want[:, score] = n
    want[:, 91 + content_row] = n
    want[L - 1, 95] = n
    if not adapters and L > 10:
        want[10, 96] = n          # kmers == NULL: i stays 10 (quack.c:210-217)
    return want


def test_more_than_2_to_the_32_reads_into_one_accumulator():
    """440 submits of a 10M-read constant batch: 4.4e9 reads > 2^32 through ONE accumulator — the 32-bit side table must
    have been folded into the u64 table on the way (it takes one count per read and position)"""
    n, L, passes = 10_000_000, 150, 440
    seq, qual = constant_batch(n, L)
    with quack_amd.Accumulator(0, None, max_len_hint=L) as acc:
        for _ in range(passes):
            acc.submit_device(seq, qual, None, n, n * L, L)
        sd = acc.finish()
    assert sd.number_of_sequences == n * passes > 1 << 32
    np.testing.assert_array_equal(sd.bases, closed_form(n * passes, L, Q("I"), 3, False))


@pytest.mark.parametrize("table32", [True, False], ids=["table32", "u64-flush"])
@pytest.mark.parametrize("adapters", [False, True], ids=["plain", "adapters"])
def test_17M_identical_reads(adapters, table32, monkeypatch):
    """17M x 150 of one letter and one score: every read of a workgroup's share hits the same LDS counters, every
    workgroup runs more than one round of 65,535 reads, every SWAR byte counter spills at its limit; with the adapter
    tables loaded the poly-G windows are candidates of nothing (and with rows of two reads, round 4)"""
    if not table32:
        monkeypatch.setenv("QUACK_HIP_NO_TABLE32", "1")
    import synth
    n, L = 17_000_000, 150
    seq, qual = constant_batch(n, L)
    ads = synth.synthetic_adapters()
    k = ob.kmers_from_seqs(ads)
    assert k[int("3" * 10, 4)] == 0          # GGGGGGGGGG is no adapter 10-mer
    with quack_amd.Accumulator(0, ob.kmers_to_bitset(k) if adapters else None, max_len_hint=L) as acc:
        acc.submit_device(seq, qual, None, n, n * L, L)
        if adapters:                         # ... and as the host feed lays 150 bp reads out when adapters are loaded
            s2 = torch.full((n * 152 + 16,), ord("G"), dtype=torch.uint8, device="cuda")
            q2 = torch.full((n * 152 + 16,), ord("I"), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            acc.submit_device_padded(s2, q2, n, L, 152)
        sd = acc.finish()
    m = 2 if adapters else 1
    assert sd.number_of_sequences == m * n
    np.testing.assert_array_equal(sd.bases, m * closed_form(n, L, Q("I"), 3, adapters))


@pytest.mark.parametrize("adapters", [False, True], ids=["plain", "adapters"])
def test_17M_reads_of_T_and_of_C(adapters):
    """Round 5: the letter counters of T and C share one LDS word, 16 bits each (hist_body: spill).  17M reads of nothing but T,
    then of nothing but C: every workgroup takes a column's T count to its limit between two flushes — a carry would show up in
    C — and the other way round; fixed length, padded to 152 with the adapter tables loaded (no 10-mer of T or C is an adapter's)"""
    import synth
    n, L = 17_000_000, 150
    ads = synth.synthetic_adapters()
    k = ob.kmers_from_seqs(ads)
    assert k[int("1" * 10, 4)] == 0 and k[int("2" * 10, 4)] == 0     # TTTTTTTTTT, CCCCCCCCCC (A,T,C,G = 0,1,2,3: quack.c:150)
    want = np.zeros((L, 97), np.uint64)
    with quack_amd.Accumulator(0, ob.kmers_to_bitset(k) if adapters else None, max_len_hint=L) as acc:
        for letter, row in ((b"T", 1), (b"C", 2)):
            if adapters:
                seq = torch.full((n * 152 + 16,), letter[0], dtype=torch.uint8, device="cuda")
                ql = torch.full((n * 152 + 16,), ord("I"), dtype=torch.uint8, device="cuda")
                torch.cuda.synchronize()
                acc.submit_device_padded(seq, ql, n, L, 152)
            else:
                seq, ql = constant_batch(n, L, base=letter)
                acc.submit_device(seq, ql, None, n, n * L, L)
            acc.sync()
            del seq, ql
            want += closed_form(n, L, Q("I"), row, adapters)
        sd = acc.finish()
    assert sd.number_of_sequences == 2 * n
    np.testing.assert_array_equal(sd.bases, want)


def test_poly_g_tails_10M():
    """NovaSeq's signature: a random prefix, then G at Q2 ('#') to the end of the read"""
    n, L = 10_000_000, 150
    g = torch.Generator(device="cuda").manual_seed(21)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
    seq = torch.zeros(n * L + 16, dtype=torch.uint8, device="cuda")
    qual = torch.zeros(n * L + 16, dtype=torch.uint8, device="cuda")
    cut = torch.randint(30, L + 1, (n,), generator=g, device="cuda")
    pos = torch.arange(L, device="cuda")
    step = 1_000_000
    for a in range(0, n, step):
        tail = pos[None, :] >= cut[a:a + step, None]
        s = lut[torch.randint(0, 4, (step, L), generator=g, device="cuda")]
        q = (33 + torch.randint(20, 41, (step, L), generator=g, device="cuda")).to(torch.uint8)
        s[tail] = ord("G")
        q[tail] = ord("#")
        seq[a * L:(a + step) * L] = s.reshape(-1)
        qual[a * L:(a + step) * L] = q.reshape(-1)
    torch.cuda.synchronize()
    with quack_amd.Accumulator(0, None, max_len_hint=L) as acc:
        acc.submit_device(seq, qual, None, n, n * L, L)
        sd = acc.finish()
    want, wn = ob.accumulate_batch_threads(seq[:n * L].cpu().numpy(), qual[:n * L].cpu().numpy(), read_len=L)
    assert wn == n == sd.number_of_sequences
    np.testing.assert_array_equal(sd.bases, want)
    assert sd.bases[L - 1, Q("#")] > 0.9 * n * (L - 30) / (L - 29)      # nearly every read ends in its tail


@pytest.mark.parametrize("table32", [True, False], ids=["table32", "u64-flush"])
def test_four_level_novaseq_qualities_10M(table32, monkeypatch):
    if not table32:
        monkeypatch.setenv("QUACK_HIP_NO_TABLE32", "1")
    import bench
    w = dict(bench.WORKLOADS["cfg2"])
    b = bench.make_batch(torch, np, w, seed=31, device=torch.device("cuda", 0), quality="novaseq4")
    n, L = w["n"], w["L"]
    torch.cuda.synchronize()
    with quack_amd.Accumulator(0, None, max_len_hint=L) as acc:
        acc.submit_device(b["seq"], b["qual"], None, n, n * L, L)
        sd = acc.finish()
    want, wn = ob.accumulate_batch_threads(b["seq"][:n * L].cpu().numpy(), b["qual"][:n * L].cpu().numpy(), read_len=L)
    assert wn == n == sd.number_of_sequences
    np.testing.assert_array_equal(sd.bases, want)
    assert set(np.flatnonzero(sd.bases[:, :91].sum(axis=0))) == {2, 12, 23, 37}


def test_every_read_starts_with_an_adapter_10M():
    """every read carries an adapter from its first base on: every lane of the first positions holds candidates in every
    step, every read's first hit is the adapter's second window -> bases[11].kmer_count == n (quack.c:206-217); packed (12-byte windows)
    and padded to 152 (16 positions per lane, rows of two reads)"""
    import bench
    n, L = 10_000_000, 150
    bits, ads = bench.synthetic_adapter_bits(np)
    k = ob.kmers_from_seqs([bytes(a) for a in ads])
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(41)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    width = max(len(a) for a in ads)
    tab = torch.zeros((len(ads), width), dtype=torch.uint8, device=dev)
    alen = torch.tensor([len(a) for a in ads], device=dev)
    for i, a in enumerate(ads):
        tab[i, :len(a)] = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    seq = torch.zeros(n * L + 16, dtype=torch.uint8, device=dev)
    qual = torch.zeros(n * L + 16, dtype=torch.uint8, device=dev)
    step = 1_000_000
    col = torch.arange(width, device=dev)
    for a in range(0, n, step):
        s = lut[torch.randint(0, 4, (step, L), generator=g, device=dev)]
        which = torch.randint(0, len(ads), (step,), generator=g, device=dev)
        head = tab[which]                                     # (step, width)
        keep = col[None, :] < alen[which][:, None]
        s[:, :width] = torch.where(keep, head, s[:, :width])
        seq[a * L:(a + step) * L] = s.reshape(-1)
        qual[a * L:(a + step) * L] = (33 + torch.randint(2, 42, (step * L,), generator=g, device=dev)).to(torch.uint8)
    s2 = torch.zeros(n * 152 + 16, dtype=torch.uint8, device=dev)
    q2 = torch.zeros(n * 152 + 16, dtype=torch.uint8, device=dev)
    s2[:n * 152].view(n, 152)[:, :L] = seq[:n * L].view(n, L)
    q2[:n * 152].view(n, 152)[:, :L] = qual[:n * L].view(n, L)
    torch.cuda.synchronize()
    with quack_amd.Accumulator(0, bits, max_len_hint=L) as acc:
        acc.submit_device(seq, qual, None, n, n * L, L)
        acc.submit_device_padded(s2, q2, n, L, 152)
        sd = acc.finish()
    # (read_adapters never inserts an adapter's FIRST window, quack.c:162-171: the first hit is the window ending at 10, i = 11)
    assert sd.number_of_sequences == 2 * n and sd.bases[11, 96] == 2 * n and sd.bases[:, 96].sum() == 2 * n
    want, wn = ob.accumulate_batch_threads(seq[:n * L].cpu().numpy(), qual[:n * L].cpu().numpy(), read_len=L, kmers=k)
    assert wn == n
    np.testing.assert_array_equal(sd.bases, 2 * want)

"""Run ON THE GPU BOX: random batches through the C-ABI against the oracle for a given time.
   python tools/fuzz_gpu.py [--seconds 120] [--seed 1]
Every round draws a batch family (fixed length / packed ragged / long reads on cache lines / fixed stride + lengths),
a shape, an alphabet, whether adapters are loaded and spliced in, and how the batch reaches the device (host arrays in
1-3 submits, or device-resident); compares every counter with oracle/quack_oracle.c.  Prints the seed of a failing
round (re-run with --seed S --rounds 1) and exits 1.  Developer tool: tests/ holds the fixed cases."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np


def fuzz(seconds, seed, rounds=1 << 30, log=print):
    """-> (rounds done, None) or (rounds done, description of the first mismatch); tests/test_gpu_fuzz.py runs a
    seeded, time-boxed stretch of it inside the -m gpu suite"""
    import types
    a = types.SimpleNamespace(seconds=seconds, seed=seed, rounds=rounds)
    import torch
    import oracle_binding as ob
    import synth
    import quack_amd
    from quack_amd.api import pad_for_device
    ads = synth.synthetic_adapters()
    kmers = ob.kmers_from_seqs(ads)
    bits = ob.kmers_to_bitset(kmers)
    t_end = time.time() + a.seconds
    done = 0
    seed = a.seed
    while time.time() < t_end and done < a.rounds:
        rng = np.random.default_rng(seed)
        family = rng.choice(["fixed", "ragged", "long", "strided"], p=[0.4, 0.2, 0.2, 0.2])
        adapters = bool(rng.integers(0, 2))
        # (the fifth: ordinary reads with an IUPAC letter / odd byte in ~1500 — the table-lookup codes of round 5 take their exact
        #  path for the waves that meet one and the lookup for the rest, inside one launch)
        alpha_i = int(rng.integers(0, 5))
        alphabet = [b"ACGT", b"ACGTN", b"ACGTNacgtn", bytes(range(256)), b"ACGTN"][alpha_i]
        alpha = np.frombuffer(alphabet, np.uint8)
        if family == "fixed":
            L = int(rng.choice([int(rng.integers(1, 40)), int(rng.integers(40, 320)), int(rng.integers(320, 700)), 4 * int(rng.integers(16, 150))]))
            n = int(rng.integers(1, max(2, min(150000, 20_000_000 // L))))
            lens = np.full(n, L)
        elif family == "ragged":
            lo = int(rng.integers(0, 50)); hi = lo + int(rng.integers(1, 600))
            n = int(rng.integers(1, 60000))
            lens = rng.integers(lo, hi + 1, n)
        elif family == "long":
            lo = int(rng.integers(0, 3000)); hi = lo + int(rng.integers(600, 40000))
            n = int(rng.integers(1, max(2, min(4000, 30_000_000 // hi))))
            lens = rng.integers(lo, hi + 1, n)
        else:
            L = int(rng.integers(20, 260))
            n = int(rng.integers(1, 120000))
            lens = np.where(rng.random(n) < 0.7, L, rng.integers(max(1, L - 40), L + 1, n))
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        total = int(off[-1])
        seq = alpha[rng.integers(0, len(alpha), total)]
        if alpha_i == 4 and total:
            odd = np.frombuffer(b"BDEFHIJKLMOPQRSUVWXYbdefhijklmopqrsuvwxy@[`{\x00\xff\x7f0123456789", np.uint8)
            at = np.flatnonzero(rng.random(total) < 1 / 1500)
            seq = seq.copy()
            seq[at] = odd[rng.integers(0, len(odd), len(at))]
        qual = (rng.integers(0, 256, total) if rng.random() < 0.2 else 33 + rng.integers(0, 94, total)).astype(np.uint8)
        if adapters and total:
            for r in rng.integers(0, n, max(1, n // 3)):
                s, e = int(off[r]), int(off[r + 1])
                if e - s > 12:
                    ad = np.frombuffer(ads[int(rng.integers(0, len(ads)))], np.uint8)
                    at = s + int(rng.integers(0, e - s - 1))
                    m = min(len(ad), e - at)
                    seq[at:at + m] = ad[:m]
        k = kmers if adapters else None
        kb = bits if adapters else None
        fixed = family == "fixed"
        want = ob.accumulate_batch(seq, qual, None if fixed else off, read_len=int(lens[0]) if fixed else 0, kmers=k)
        how = "?"
        with quack_amd.Accumulator(0, kb) as acc:
            route = int(rng.integers(0, 3))
            if family == "fixed":
                L = int(lens[0])
                if route == 2 and (L & 3):
                    stride = (L + 3) & ~3
                    how = "submit_device_padded stride %d" % stride
                    s2 = np.full((n, stride), ord("C"), np.uint8)
                    q2 = np.full((n, stride), 50, np.uint8)
                    s2[:, :L], q2[:, :L] = seq.reshape(n, L), qual.reshape(n, L)
                    d_s, d_q = torch.from_numpy(pad_for_device(s2.reshape(-1))).cuda(), torch.from_numpy(pad_for_device(q2.reshape(-1))).cuda()
                    acc.submit_device_padded(d_s, d_q, n, L, stride)
                elif route == 0 or (L & 3):
                    how = "submit_fixed x%d" % (1 + route)
                    cut = [n * i // (1 + route) for i in range(2 + route)]
                    for x, y in zip(cut, cut[1:]):
                        acc.submit_fixed(seq[x * L:y * L], qual[x * L:y * L], L)
                else:
                    how = "submit_device (fixed)"
                    d_s, d_q = torch.from_numpy(pad_for_device(seq)).cuda(), torch.from_numpy(pad_for_device(qual)).cuda()
                    acc.submit_device(d_s, d_q, None, n, total, L)
            elif family == "ragged":
                how = "submit x%d" % (1 + route)
                cut = [n * i // (1 + route) for i in range(2 + route)]
                for x, y in zip(cut, cut[1:]):
                    lo_, hi_ = int(off[x]), int(off[y])
                    acc.submit(seq[lo_:hi_], qual[lo_:hi_], off[x:y + 1] - off[x])
            elif family == "long":
                if route == 0:
                    how = "submit (packed)"
                    acc.submit(seq, qual, off)
                else:
                    how = "submit_device_gapped aligned128"
                    starts = np.concatenate([[0], np.cumsum((lens + 127) // 128 * 128)])[:-1].astype(np.uint64)
                    extent = int(starts[-1] + lens[-1])
                    s2, q2 = np.full(extent, ord("T"), np.uint8), np.full(extent, 70, np.uint8)
                    idx = np.repeat(starts.astype(np.int64) - off[:-1].astype(np.int64), lens) + np.arange(total)
                    s2[idx], q2[idx] = seq, qual
                    d_s, d_q = torch.from_numpy(pad_for_device(s2)).cuda(), torch.from_numpy(pad_for_device(q2)).cuda()
                    d_st, d_l = torch.from_numpy(starts.astype(np.int64)).cuda(), torch.from_numpy(lens.astype(np.int32)).cuda()
                    acc.submit_device_gapped(d_s, d_q, d_st, d_l, n, extent, int(lens.max()), aligned=True)
            else:
                stride = (L + 3) // 4 * 4 + 4 * int(rng.integers(0, 2))
                how = "submit_strided stride %d" % stride
                s2, q2 = np.full(n * stride, ord("G"), np.uint8), np.full(n * stride, 40, np.uint8)
                idx = np.repeat(np.arange(n, dtype=np.int64) * stride - off[:-1].astype(np.int64), lens) + np.arange(total)
                s2[idx], q2[idx] = seq, qual
                if route == 2:     # 0xFF behind every read, promised (QK_BATCH_NEUTRAL_PADS), device-resident
                    how = "submit_device_strided neutral pads, stride %d" % stride
                    s2[:], q2[:] = 0xFF, 0xFF
                    s2[idx], q2[idx] = seq, qual
                    d_s, d_q = torch.from_numpy(pad_for_device(s2)).cuda(), torch.from_numpy(pad_for_device(q2)).cuda()
                    d_l = torch.from_numpy(lens.astype(np.int32)).cuda()
                    acc.submit_device_strided(d_s, d_q, d_l, n, stride, int(lens.max()), neutral_pads=True)
                else:
                    acc.submit_strided(s2, q2, lens.astype(np.uint32), stride)
            sd = acc.finish()
        ok = sd.number_of_sequences == want[1] and sd.bases.shape == want[0].shape and np.array_equal(sd.bases, want[0])
        if not ok:
            msg = "MISMATCH seed %d: %s n=%d lens %d..%d adapters=%s alphabet=%d via %s" % (seed, family, n, lens.min(), lens.max(), adapters, len(alphabet), how)
            if sd.bases.shape == want[0].shape:
                pos, row = np.argwhere(sd.bases != want[0])[0]
                msg += "; first at position %d row %d: hip %d oracle %d (%d cells)" % (pos, row, sd.bases[pos, row], want[0][pos, row], (sd.bases != want[0]).sum())
            return done, msg
        done += 1
        seed += 1
        if done % 20 == 0:
            log("%d rounds ok (last: %s n=%d via %s)" % (done, family, n, how))
    return done, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=1 << 30)
    a = ap.parse_args()
    done, bad = fuzz(a.seconds, a.seed, a.rounds, log=lambda m: print(m, flush=True))
    if bad:
        print(bad, flush=True)
        sys.exit(1)
    print("fuzz: %d rounds, all equal to the oracle (seeds %d..%d)" % (done, a.seed, a.seed + done - 1))


if __name__ == "__main__":
    main()

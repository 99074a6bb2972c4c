"""Seeded synthetic batches in the C-ABI layout (numpy, host side)."""
import numpy as np

ACGT = np.frombuffer(b"ACGT", np.uint8)


def fixed(n_reads, read_len, seed, q_lo=2, q_hi=41):
    rng = np.random.default_rng(seed)
    seq = ACGT[rng.integers(0, 4, n_reads * read_len, dtype=np.uint8)]
    qual = (33 + rng.integers(q_lo, q_hi + 1, n_reads * read_len, dtype=np.uint8)).astype(np.uint8)
    return seq, qual


def ragged(n_reads, lo, hi, seed, q_lo=2, q_hi=41, alphabet=b"ACGT"):
    rng = np.random.default_rng(seed)
    lens = rng.integers(lo, hi + 1, n_reads)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    total = int(off[-1])
    alpha = np.frombuffer(alphabet, np.uint8)
    seq = alpha[rng.integers(0, len(alpha), total)]
    qual = (33 + rng.integers(q_lo, q_hi + 1, total)).astype(np.uint8)
    return seq, qual, off


def splice_adapters(seq, read_len, adapters, seed, fraction=0.25):
    """config 3: a quarter of the reads get one adapter at a uniform offset,
    truncated at the read end"""
    rng = np.random.default_rng(seed)
    seq = seq.copy().reshape(-1, read_len)
    n = seq.shape[0]
    pick = np.flatnonzero(rng.random(n) < fraction)
    for r in pick:
        ad = np.frombuffer(adapters[rng.integers(0, len(adapters))], np.uint8)
        at = int(rng.integers(0, read_len))
        m = min(len(ad), read_len - at)
        seq[r, at:at + m] = ad[:m]
    return seq.reshape(-1)


def synthetic_adapters(seed=3, n=24):
    rng = np.random.default_rng(seed)
    return [bytes(ACGT[rng.integers(0, 4, int(rng.integers(30, 61)))]) for _ in range(n)]

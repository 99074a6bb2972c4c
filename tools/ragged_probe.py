"""How fast the packed ragged path runs heavily trimmed reads (round 5 probe): 10M reads of U[lo, hi] bases, with and without the adapter
table, device-resident; kernel time from the library's HIP events.   python tools/ragged_probe.py [lo hi]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, quack_amd
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (30, 150)
n = 10_000_000
rng = np.random.default_rng(1)
lens = rng.integers(lo, hi + 1, n)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
total = int(off[-1])
g = torch.Generator(device="cuda").manual_seed(1)
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
seq = torch.zeros(total + 16, dtype=torch.uint8, device="cuda"); qual = torch.zeros(total + 16, dtype=torch.uint8, device="cuda")
seq[:total] = lut[torch.randint(0, 4, (total,), generator=g, device="cuda")]
qual[:total] = (33 + torch.randint(2, 42, (total,), generator=g, device="cuda")).to(torch.uint8)
d_off = torch.from_numpy(off).cuda()
bits, ads = bench.synthetic_adapter_bits(np)
torch.cuda.synchronize()
for name, kb in (("no adapters", None), ("adapters", bits)):
    with quack_amd.Accumulator(0, kb, max_len_hint=hi) as acc:
        for _ in range(30):
            acc.submit_device(seq, qual, d_off, n, total, hi)
        acc.sync(); acc.timing(1)
        for _ in range(30):
            acc.submit_device(seq, qual, d_off, n, total, hi)
        acc.sync()
        ms, bms, l = acc.timing_read_batch()
        alg = 2.0 * total + 8.0 * n
        print("packed  U[%d,%d] 10M reads, %-11s: kernel %.4f ms, batch %.4f ms, frac %.3f of 8 TB/s (%.2f Gbases, fill %.2f of the longest read)" % (
            lo, hi, name, ms / l, bms / l, alg / (ms / l * 1e-3) / 8e12, total / 1e9, total / (n * hi)))
# the same reads at a fixed stride with 0xFF pads (what the host feed gives "one length, some of them trimmed")
stride = (hi + 3) & ~3
s2 = torch.full((n * stride + 16,), 255, dtype=torch.uint8, device="cuda")
q2 = torch.full((n * stride + 16,), 255, dtype=torch.uint8, device="cuda")
d_len = torch.from_numpy(lens.astype(np.int32)).cuda()
for a in range(0, n, 1 << 20):
    e = min(n, a + (1 << 20))
    ll = d_len[a:e].long()
    idx = (torch.arange(a, e, device="cuda") * stride).repeat_interleave(ll) + (torch.arange(int(off[a]), int(off[e]), device="cuda") - d_off[a:e].repeat_interleave(ll))
    s2[idx] = seq[int(off[a]):int(off[e])]
    q2[idx] = qual[int(off[a]):int(off[e])]
torch.cuda.synchronize()
for name, kb in (("no adapters", None), ("adapters", bits)):
    with quack_amd.Accumulator(0, kb, max_len_hint=hi) as acc:
        for _ in range(30):
            acc.submit_device_strided(s2, q2, d_len, n, stride, hi, neutral_pads=True)
        acc.sync(); acc.timing(1)
        for _ in range(30):
            acc.submit_device_strided(s2, q2, d_len, n, stride, hi, neutral_pads=True)
        acc.sync()
        ms, bms, l = acc.timing_read_batch()
        alg = 2.0 * total + 4.0 * n
        print("strided U[%d,%d] 10M reads, %-11s: kernel %.4f ms, batch %.4f ms, frac %.3f of 8 TB/s (stride %d, 0xFF pads)" % (lo, hi, name, ms / l, bms / l, alg / (ms / l * 1e-3) / 8e12, stride))

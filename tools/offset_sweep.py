"""Does the long-read kernel's time depend on where the quality array lies relative to the sequence array?
(bench runs of one box were bimodal, 0.557 / 0.587 ms, with nothing changed but the allocations.)
Config-5-shaped batch in ONE buffer, the quality bytes at seq + Q0 + off for a list of offsets."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import quack_amd

n = 143000
rng = np.random.default_rng(6)
lens = rng.integers(1000, 20001, n)
starts = np.concatenate([[0], np.cumsum((lens + 127) // 128 * 128)])[:-1]
extent = int(starts[-1] + lens[-1])
Q0 = (extent + (1 << 21)) // (1 << 21) * (1 << 21) + (1 << 21)
buf = torch.empty(2 * Q0 + (1 << 26), dtype=torch.uint8, device="cuda")
print("buffer at %#x, extent %d, Q0 %#x" % (buf.data_ptr(), extent, Q0), flush=True)
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
buf[:Q0] = lut[torch.randint(0, 4, (Q0,), device="cuda")]
d_st = torch.from_numpy(starts.astype(np.int64)).cuda()
d_l = torch.from_numpy(lens.astype(np.int32)).cuda()
offs = [int(x) for x in sys.argv[1:]] or [0, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536, 1 << 18, 1 << 20, (1 << 20) + 4096, 3 << 19]
for rep in range(2):
    for off in offs:
        q = buf[Q0 + off:Q0 + off + Q0]
        q[:] = (33 + torch.randint(0, 42, (Q0,), device="cuda")).to(torch.uint8)
        seq = buf[:Q0]
        with quack_amd.Accumulator(0, None, max_len_hint=20000) as acc:
            for _ in range(60):
                acc.submit_device_gapped(seq, q, d_st, d_l, n, extent, int(lens.max()), aligned=True)
            acc.sync()
            acc.timing(1)
            for _ in range(40):
                acc.submit_device_gapped(seq, q, d_st, d_l, n, extent, int(lens.max()), aligned=True)
            acc.sync()
            k, b, l = acc.timing_read_batch()
            lo, hi = acc.timing_read_range()
            acc.finish()
        print("off %9d (%#9x): kernel %.4f ms (%.4f..%.4f) batch %.4f" % (off, off, k, lo, hi, b), flush=True)

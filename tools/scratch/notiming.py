import time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import quack_amd
n, L = 10_000_000, 150
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(2)
seq = torch.randint(0, 4, (n * L + 16,), device=dev, dtype=torch.uint8, generator=g)
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
seq = lut[seq.long()]
qual = (torch.randint(2, 42, (n * L + 16,), device=dev, dtype=torch.uint8, generator=g) + 33)
for timing in (False, True, False, True):
    with quack_amd.Accumulator(0, None, max_len_hint=L) as acc:
        for _ in range(50):
            acc.submit_device(seq, qual, None, n, n * L, L)
        acc.sync(); torch.cuda.synchronize()
        acc.timing(timing)
        t0 = time.perf_counter()
        for _ in range(200):
            acc.submit_device(seq, qual, None, n, n * L, L)
        acc.sync(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 200 * 1e3
        k = acc.timing_read_batch() if timing else (0, 0, 1)
        print("timing", timing, "ms/step %.4f" % dt, "kernel %.4f" % (k[0] / max(k[2], 1)))

# per-launch fixed cost of the histogram kernel: tiny batches, the kernel cut short at several points (QK_DBG_STOP build)
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import quack_amd
L = 150
dev = torch.device("cuda", 0)
for n in (1024, 60000, 1000000):
    seq = torch.full((n * L + 16,), 65, device=dev, dtype=torch.uint8)
    qual = torch.full((n * L + 16,), 70, device=dev, dtype=torch.uint8)
    with quack_amd.Accumulator(0, None, max_len_hint=L) as acc:
        for _ in range(20):
            acc.submit_device(seq, qual, None, n, n * L, L)
        acc.sync()
        acc.timing(1)
        for _ in range(200):
            acc.submit_device(seq, qual, None, n, n * L, L)
        acc.sync()
        k = acc.timing_read_batch()
        print("stop", os.environ.get("QK_DBG_STOP", "0"), "n", n, "kernel %.2f us" % (1e3 * k[0] / k[2]))

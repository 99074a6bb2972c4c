import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, torch, quack_amd, synth
from quack_amd.api import pad_for_device
seq, qual = synth.fixed(2000, 152, seed=1)
d_s, d_q = torch.from_numpy(pad_for_device(seq)).cuda(), torch.from_numpy(pad_for_device(qual)).cuda()
accs = [quack_amd.Accumulator(0, None) for _ in range(3)]
side = torch.cuda.Stream()
t0 = time.time()
for it in range(300):
    for i, a in enumerate(accs):
        a.timing(1 if it % 3 else 2)
        for k in range(40):
            a.submit_device(d_s, d_q, None, 2000, len(seq), 152, stream=(side.cuda_stream if (i == 1) else None))
        a.submit_fixed(seq, qual, 152)
    for a in accs:
        a.sync()
        ms, bms, n = a.timing_read_batch()
        assert n > 0 and ms > 0, (ms, n)
    if it % 50 == 0:
        print("round", it, "%.1f s" % (time.time() - t0), flush=True)
for a in accs:
    a.finish(); a.close()
print("ext stress ok: %d timed launches in %.1f s" % (300 * 3 * 41, time.time() - t0))

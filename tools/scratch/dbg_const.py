import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_binding as ob, synth, quack_amd
ads = synth.synthetic_adapters(); k = ob.kmers_from_seqs(ads); bits = ob.kmers_to_bitset(k)
L = 150
def run(n, mode, env):
    for kk, v in env.items(): os.environ[kk] = v
    want = np.zeros((L, 97), np.uint64); want[:, 40] = n; want[:, 94] = n; want[L-1, 95] = n
    with quack_amd.Accumulator(0, bits, max_len_hint=L) as acc:
        if mode == "packed":
            s = torch.full((n*L+16,), ord("G"), dtype=torch.uint8, device="cuda"); q = torch.full((n*L+16,), ord("I"), dtype=torch.uint8, device="cuda")
            acc.submit_device(s, q, None, n, n*L, L)
        else:
            s = torch.full((n*152+16,), ord("G"), dtype=torch.uint8, device="cuda"); q = torch.full((n*152+16,), ord("I"), dtype=torch.uint8, device="cuda")
            acc.submit_device_padded(s, q, n, L, 152)
        sd = acc.finish()
    for kk in env: del os.environ[kk]
    d = sd.bases.astype(np.int64) - want.astype(np.int64)
    bad = np.argwhere(d != 0)
    print(n, mode, env, "reads", sd.number_of_sequences, "bad cells", len(bad), "first", [(int(p), int(r), int(d[p, r])) for p, r in bad[:6]], flush=True)
for n in (100_000, 1_000_000, 10_000_000, 17_000_000):
    for mode in ("packed", "padded"):
        for env in ({}, {"QUACK_HIP_NO_GROUP": "1"}, {"QUACK_HIP_SMALL_RING": "1"}):
            if mode == "packed" and env: continue
            run(n, mode, env)

"""One rank of the N>1 path on HIP tables (started by tests/test_gpu_multi.py,
one process per rank): accumulate this rank's share of a seeded batch on the
GPU, exchange the tables with torch.distributed (nccl = RCCL over xGMI on
distinct devices; gloo + host staging when several ranks share one device),
and on rank 0 compare the merged table with the oracle over ALL reads.

    multi_worker.py <backend:nccl|gloo> <devices: same|distinct> <out.json>
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    backend, devices, out_path = sys.argv[1:4]
    import torch
    import torch.distributed as dist
    import quack_amd
    from quack_amd import distributed as qd
    import oracle_binding as ob
    import synth

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = 0 if devices == "same" else int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group("gloo")
    ads = synth.synthetic_adapters()
    k = ob.kmers_from_seqs(ads)
    # ranks see different longest reads, so the tables have different lengths before the exchange
    seq, qual, off = synth.ragged(24000, 1, 260, seed=81, alphabet=b"ACGTNacgt")
    seq = seq.copy()
    for r in range(0, 24000, 7):           # adapters in some reads
        a, e = int(off[r]), int(off[r + 1])
        ad = np.frombuffer(ads[r % len(ads)], np.uint8)
        m = min(len(ad), e - a)
        seq[a:a + m] = ad[:m]
    n = len(off) - 1
    cut = [n * i // (3 * world) for i in range(3 * world + 1)]   # batches dealt round-robin to the ranks
    with quack_amd.Accumulator(local, ob.kmers_to_bitset(k), max_len_hint=8) as acc:
        for b, (a, e) in enumerate(zip(cut, cut[1:])):
            if b % world != rank:
                continue
            lo, hi = int(off[a]), int(off[e])
            acc.submit(seq[lo:hi], qual[lo:hi], off[a:e + 1] - off[a])
        # a second accumulator travels in the same exchange (the two mates of a pair do): every rank gives it
        # the first 500 reads without adapters -> world x that table
        with quack_amd.Accumulator(local, None, max_len_hint=8) as other:
            o500 = int(off[500])
            other.submit(seq[:o500], qual[:o500], off[:501])
            qd.allreduce_accumulators([acc, other], via_host=backend == "gloo")
            osd = other.finish()
        sd = acc.finish()
    ok = True
    if rank == 0:
        want, nseq = ob.accumulate_batch(seq, qual, off, kmers=k)
        ok = sd.number_of_sequences == nseq and np.array_equal(sd.bases, want)
        o500 = int(off[500])
        owant, on = ob.accumulate_batch(seq[:o500], qual[:o500], off[:501])
        ok = ok and osd.number_of_sequences == world * on and np.array_equal(osd.bases, world * owant)
        json.dump({"ok": bool(ok), "world": dist.get_world_size(), "backend": backend,
                   "reads": int(sd.number_of_sequences), "max_length": int(sd.bases.shape[0]),
                   "kmer_hits": int(want[:, 96].sum())}, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()

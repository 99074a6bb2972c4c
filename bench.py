#!/usr/bin/env python3
"""bench.py — headline benchmark of the accumulation path (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path (histogram kernel[s]) over one batch of
synthetic reads that is already resident in HBM.  Workload at N=1 = the
configuration the metric is quoted on: 10M-read synthetic 150 bp FASTQ, no
adapters (BASELINE.json configs[1]); every rank holds its own 10M-read batch
(weak scaling; the path shards by read batch with no data-path collective),
and the job ends with the single RCCL all-reduce of the integer tables, which
is inside the timed region.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     dominant kernel, algorithmic bytes (2 B/base [+8 B/read ragged])
               / average launch duration from HIP events on the launch stream
  cpu_baseline the oracle (CPU restatement, kind "port") on one host core over
               the same batch — a reported baseline, not the target
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import quack_amd  # noqa: E402  (fails loudly when the native libraries are missing)
from quack_amd import distributed as qd  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)

WORKLOADS = {
    # name: (reads, read_len, ragged (lo, hi), adapters, BASELINE.json config)
    "cfg2": dict(n=10_000_000, L=150, ragged=None, adapters=False,
                 label="10M-read synthetic 150 bp FASTQ, no adapters (BASELINE.json configs[1])"),
    "cfg3": dict(n=10_000_000, L=300, ragged=None, adapters=True,
                 label="10M-read synthetic 300 bp FASTQ + adapter FASTA (configs[2])"),
    # long reads live in HBM the way the host feed lays them out (pipeline.c): every read starts on a
    # 128-byte cache line (QK_BATCH_ALIGNED128); cfg5packed = the same reads without the padding
    "cfg5": dict(n=143_000, L=20000, ragged=(1000, 20000), adapters=False, aligned=True,
                 label="PacBio-style ragged 1-20 kb synthetic FASTQ, reads on 128-B lines as the host feed lays them out (configs[4])"),
    "cfg5packed": dict(n=143_000, L=20000, ragged=(1000, 20000), adapters=False,
                 label="PacBio-style ragged 1-20 kb synthetic FASTQ, packed (configs[4])"),
    # configs[3]: paired 2 x 50M x 150 bp over 8 GPUs -> per GPU 2 x 6.25M reads; the two mates are two
    # independent accumulations (quack.c:911-921); R2 qualities skewed lower
    "cfg4": dict(n=6_250_000, L=150, ragged=None, adapters=False, paired=True,
                 label="paired 2x50M 150 bp sharded over 8 GPUs: per-GPU share 2 x 6.25M reads (configs[3])"),
}


def make_batch(w, seed, device, quality="uniform", q_hi_override=None):
    g = torch.Generator(device=device).manual_seed(seed)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    if w["ragged"]:
        rng = np.random.default_rng(seed)
        lens = rng.integers(w["ragged"][0], w["ragged"][1] + 1, w["n"])
        d_len = None
        if w.get("aligned"):
            starts = np.concatenate([[0], np.cumsum((lens + 127) // 128 * 128)]).astype(np.int64)
            extent = int(starts[-2] + lens[-1])
            d_off = torch.from_numpy(starts[:-1].copy()).to(device)
            d_len = torch.from_numpy(lens.astype(np.int32)).to(device)
        else:
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            extent = int(off[-1])
            d_off = torch.from_numpy(off).to(device)
        total, max_len = int(lens.sum()), int(lens.max())
        q_lo, q_hi = 1, 60
    else:
        total, max_len, d_off, d_len = w["n"] * w["L"], w["L"], None, None
        extent = total
        q_lo, q_hi = 2, (q_hi_override or 41)
    seq = torch.zeros(extent + 16, dtype=torch.uint8, device=device)
    qual = torch.zeros(extent + 16, dtype=torch.uint8, device=device)
    step = 1 << 28
    for a in range(0, extent, step):   # (the padding between aligned reads holds letters and scores too: never counted)
        b = min(extent, a + step)
        seq[a:b] = lut[torch.randint(0, 4, (b - a,), generator=g, device=device)]
        if quality == "novaseq4":
            levels = torch.tensor([33 + 2, 33 + 12, 33 + 23, 33 + 37], dtype=torch.uint8, device=device)
            u = torch.rand(b - a, generator=g, device=device)
            idx = (u > 0.03).long() + (u > 0.08).long() + (u > 0.20).long()
            qual[a:b] = levels[idx]
        else:
            qual[a:b] = (33 + torch.randint(q_lo, q_hi + 1, (b - a,), generator=g, device=device)).to(torch.uint8)
    return seq, qual, d_off, total, max_len, d_len, extent


def synthetic_adapter_bits(seed=3):
    """config 3's adapter FASTA: 24 records of 30-60 nt -> 2^20-bit table via
    the product's read_adapters rule (quack_amd.host qkh_adapter_insert)"""
    import ctypes
    from quack_amd import _capi
    rng = np.random.default_rng(seed)
    bits = np.zeros(_capi.QK_KMER_TABLE_WORDS, dtype=np.uint32)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    ads = []
    for _ in range(24):
        s = acgt[rng.integers(0, 4, int(rng.integers(30, 61)))].copy()
        ads.append(s)
        _capi.host().qkh_adapter_insert(bits.ctypes.data, s.ctypes.data, len(s))
    return bits, ads


def cpu_baseline(seq, qual, d_off, n, total, w, ads, d_len=None):
    """oracle on one host core over (a bounded sample of) the same batch"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    budget_bases = 3_000_000_000     # ~12 s at ~0.25 Gbases/s
    kmers = ob.kmers_from_seqs([bytes(a) for a in ads]) if ads is not None else None
    if d_off is None:
        m = min(n, max(1, budget_bases // w["L"]))
        hs, hq = seq[:m * w["L"]].cpu().numpy(), qual[:m * w["L"]].cpu().numpy()
        t0 = time.perf_counter()
        ob.accumulate_batch(hs, hq, read_len=w["L"], kmers=kmers)
        dt = time.perf_counter() - t0
        bases, sample = m * w["L"], "%d of %d reads x %d bp (same bytes as the GPU batch)" % (m, n, w["L"])
    else:
        if d_len is not None:   # gapped on the device: the oracle takes the same reads packed
            starts, lens = d_off.cpu().numpy().astype(np.int64), d_len.cpu().numpy().astype(np.int64)
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
            m = int(min(n, np.searchsorted(off, budget_bases)))
            end = int(starts[m - 1] + lens[m - 1])
            gs, gq = seq[:end].cpu().numpy(), qual[:end].cpu().numpy()
            keep = np.zeros(end, dtype=bool)
            for a, l in zip(starts[:m], lens[:m]):
                keep[a:a + l] = True
            hs, hq = gs[keep], gq[keep]
        else:
            off = d_off.cpu().numpy().astype(np.uint64)
            m = int(min(n, np.searchsorted(off, budget_bases)))
            hs, hq = seq[:int(off[m])].cpu().numpy(), qual[:int(off[m])].cpu().numpy()
        t0 = time.perf_counter()
        ob.accumulate_batch(hs, hq, off[:m + 1], kmers=kmers)
        dt = time.perf_counter() - t0
        bases, sample = int(off[m]), "%d of %d ragged reads (same bytes as the GPU batch)" % (m, n)
    return {"value": bases / dt, "unit": "bases/s", "cores": 1, "kind": "port", "sample": sample,
            "seconds": round(dt, 3), "host": "oracle/quack_oracle.c, single thread (quack is single-threaded)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    # rehearsal on a one-GPU box: several ranks share one device and the table
    # exchange goes through gloo (the driver's runs use the defaults: nccl = RCCL)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--device", type=int, default=None, help="force this device for every rank")
    ap.add_argument("--reads", type=int, default=None, help="override reads per GPU (rehearsals)")
    ap.add_argument("--quality", default="uniform", choices=["uniform", "novaseq4"],
                    help="novaseq4: Q in {2,12,23,37} with 3/5/12/80 %% (stress for same-bin LDS atomics)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))
    if args.device is not None:
        local = args.device
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    w = dict(WORKLOADS[args.workload])
    if args.reads:
        w["n"] = args.reads
    bits, ads = synthetic_adapter_bits() if w["adapters"] else (None, None)
    seq, qual, d_off, total, max_len, d_len, extent = make_batch(w, seed=2 + rank, device=device, quality=args.quality)
    n = w["n"]
    alg_bytes = 2.0 * total + ((12.0 if d_len is not None else 8.0) * n if d_off is not None else 0.0)

    acc = quack_amd.Accumulator(local, bits, max_len_hint=max_len)
    mate = None
    if w.get("paired"):
        # the reverse mate: its own batch and its own accumulator
        seq2, qual2, _, _, _, _, _ = make_batch(w, seed=1000 + rank, device=device, quality=args.quality, q_hi_override=30)
        mate = quack_amd.Accumulator(local, bits, max_len_hint=max_len)

    # paired: both mates on ONE stream, so that every launch has the GPU to itself and its
    # HIP-event duration means something (on separate streams the two kernels would overlap)
    side = torch.cuda.Stream(device) if mate is not None else None   # (the default stream's handle is NULL)
    shared_stream = side.cuda_stream if side is not None else None

    def step():
        if d_len is not None:
            acc.submit_device_gapped(seq, qual, d_off, d_len, n, extent, max_len, aligned=True, stream=shared_stream)
        else:
            acc.submit_device(seq, qual, d_off, n, total, max_len, stream=shared_stream)
        if mate is not None:
            mate.submit_device(seq2, qual2, None, n, total, max_len, stream=shared_stream)

    def fence():
        acc.sync()
        if mate is not None:
            mate.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    if world > 1:
        # warm the exchange too (communicator, collective kernels) — on a throwaway
        # accumulator, so that the measured tables stay the sum of exactly W+K steps
        with quack_amd.Accumulator(local, None, max_len_hint=max_len) as tmp:
            qd.allreduce_accumulator(tmp, via_host=args.backend == "gloo")
    fence()
    acc.timing(True)
    if mate is not None:
        mate.timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        qd.allreduce_accumulator(acc, via_host=args.backend == "gloo")   # the path's single exchange (RCCL over xGMI)
        if mate is not None:
            qd.allreduce_accumulator(mate, via_host=args.backend == "gloo")
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms, launches = acc.timing_read()
    mates = 2 if mate is not None else 1
    if mate is not None:
        ms2, l2 = mate.timing_read()
        kernel_ms, launches = kernel_ms + ms2, launches + l2

    # sanity: the counters must add up (every base carries one score and one content bin)
    sd = acc.finish()
    # after the all-reduce every rank holds the sum over ranks (equal batch sizes for fixed-length workloads)
    got = int(sd.bases[:, 91:95].sum())
    if d_off is None:
        expect = (args.warmup + args.steps) * total * world
        if got != expect:
            raise SystemExit("counter check failed: content sum %d != %d" % (got, expect))
    elif world == 1 and got != (args.warmup + args.steps) * total:
        raise SystemExit("counter check failed: content sum %d" % got)

    if rank == 0:
        kernel_s = kernel_ms * 1e-3 / max(launches, 1)
        achieved = alg_bytes / kernel_s / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tf):
            traffic = json.load(open(tf)).get(args.workload)
        out = {
            "metric": "bases/sec on synthetic 150 bp FASTQ; achieved HBM GB/s vs peak",
            "value": world * args.steps * total * mates / elapsed,
            "unit": "bases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": w["label"] + ("" if args.quality == "uniform" else " [quality: %s]" % args.quality),
                       "reads_per_gpu": n * mates, "bases_per_gpu_per_step": total * mates,
                       "resident": "HBM", "parallelism": "batch-sharded x%d, one all-reduce of u64 tables" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "qk::hist_kernel", "kernel_ms": kernel_s * 1e3,
                         "algorithmic_bytes_per_launch": alg_bytes, "launches_timed": launches},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(seq, qual, d_off, n, total, w, ads, d_len)
        print(json.dumps(out), flush=True)
    acc.close()
    if mate is not None:
        sd2 = mate.finish()
        if int(sd2.bases[:, 91:95].sum()) != (args.warmup + args.steps) * total * world:
            raise SystemExit("counter check failed for the reverse mate")
        mate.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

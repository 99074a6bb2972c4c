#!/usr/bin/env python3
"""bench.py — headline benchmark of the accumulation path (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

N>1 without a launcher (WORLD_SIZE unset): this process starts N fresh child
ranks itself — before it imports torch or touches the GPU — relays rank 0's
JSON line and exits non-zero if any rank does.  Under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the
ranks come from the launcher (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).

A "step" is one pass of the hot path (histogram kernel[s]) over one batch of
synthetic reads that is already resident in HBM.

  N = 1   the configuration the metric is quoted on: 10M-read synthetic 150 bp
          FASTQ, no adapters (BASELINE.json configs[1]); afterwards, outside
          the headline's timed region, the same run times configs[2] (10M x
          300 bp + adapters, 25 % of the reads carrying a spliced adapter) and
          configs[4] (ragged 1-20 kb) for a few steps each -> "also".
  N > 1   configs[3]'s per-GPU share: paired 2 x 50M x 150 bp over 8 GPUs =
          2 x 6.25M reads per GPU, two independent accumulators (forward /
          reverse mate, quack.c:911-921); weak scaling — every rank holds its
          own share, there is no data-path collective, and the job ends with
          ONE RCCL all-reduce of each mate's integer table, inside the timed
          region.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     dominant kernel, algorithmic bytes (2 B/base [+8 B/read ragged,
               +12 gapped]) / average launch duration from HIP events on the
               launch stream; `batch_ms` = all kernels of a step
  cpu_baseline the oracle (CPU restatement, kind "port") on one host core over
               the same bytes; cpu_baseline_threads = the same on every host
               core the process may use — reported baselines, not the target
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_CEILING_GBS = 6290.0   # the guide's measured copy ceiling
TIMING_EVERY = 8
METRIC = "bases/sec on synthetic 150 bp FASTQ; achieved HBM GB/s vs peak"
TRAFFIC_SOURCE = "profiles/hbm_traffic.json (rocprofv3 --pmc passes, builder-run; not measured in this run)"

WORKLOADS = {
    # name: (reads, read_len, ragged (lo, hi), adapters, BASELINE.json config)
    "cfg2": dict(n=10_000_000, L=150, ragged=None, adapters=False,
                 label="10M-read synthetic 150 bp FASTQ, no adapters (BASELINE.json configs[1])"),
    "cfg3": dict(n=10_000_000, L=300, ragged=None, adapters=True, splice=0.25,
                 label="10M-read synthetic 300 bp FASTQ + adapter FASTA, 25% of reads with a spliced adapter (configs[2])"),
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
    # trimmed Illumina: 150 bp reads, most of them full length, the rest cut back to 120..149
    "trimmed": dict(n=10_000_000, L=150, ragged=(120, 150), adapters=False, full=0.7, stride=152,
                    label="10M-read synthetic trimmed 150 bp FASTQ (70% full length, rest 120-149), fixed stride 152 + "
                          "per-read lengths as the host feed lays such reads out"),
    "trimmedpacked": dict(n=10_000_000, L=150, ragged=(120, 150), adapters=False, full=0.7,
                          label="10M-read synthetic trimmed 150 bp FASTQ (70% full length, rest 120-149), packed ragged"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="auto", choices=["auto"] + sorted(WORKLOADS),
                    help="auto: cfg2 at N=1 (the headline), cfg4's per-GPU share at N>1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="N=1: skip the cfg3 / cfg5 lines")
    ap.add_argument("--also-steps", type=int, default=50)
    # rehearsal on a one-GPU box: several ranks share one device and the table
    # exchange goes through gloo (the driver's runs use the defaults: nccl = RCCL)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--device", type=int, default=None, help="force this device for every rank")
    ap.add_argument("--reads", type=int, default=None, help="override reads per GPU (rehearsals)")
    ap.add_argument("--read-len", type=int, default=None,
                    help="override the read length of a fixed-length workload (kernel exploration; the line says so)")
    ap.add_argument("--quality", default="uniform", choices=["uniform", "novaseq4"],
                    help="novaseq4: Q in {2,12,23,37} with 3/5/12/80 %% (stress for same-bin LDS atomics)")
    return ap.parse_args()


# --------------------------------------------------------------------------
# N>1 without a launcher: start the ranks ourselves.  Runs BEFORE torch /
# quack_amd are imported: the parent never initialises HIP, every rank is a
# fresh child process (never an exec of a process that touched the GPU).
def self_launch(args):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rank 0 writes the JSON line to our stdout; the other ranks have nothing to say there
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    pending = set(range(args.gpus))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                sys.stderr.write("bench.py: rank %d exited with %d; stopping the other ranks\n" % (r, code))
                for o in pending:          # a rank died: the others would wait in the collective for ever
                    procs[o].terminate()
        time.sleep(0.05)
    return rc


# --------------------------------------------------------------------------
def make_batch(torch, np, w, seed, device, quality="uniform", q_hi_override=None, ads=None):
    g = torch.Generator(device=device).manual_seed(seed)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    if w["ragged"]:
        rng = np.random.default_rng(seed)
        lens = rng.integers(w["ragged"][0], w["ragged"][1] + 1, w["n"])
        if w.get("full"):
            lens[rng.random(w["n"]) < w["full"]] = w["ragged"][1]
        d_len = None
        if w.get("stride"):
            extent = w["n"] * w["stride"]
            d_off = None
            d_len = torch.from_numpy(lens.astype(np.int32)).to(device)
        elif w.get("aligned"):
            starts = np.concatenate([[0], np.cumsum((lens + 127) // 128 * 128)]).astype(np.int64)
            extent = int(starts[-2] + lens[-1])
            d_off = torch.from_numpy(starts[:-1].copy()).to(device)
            d_len = torch.from_numpy(lens.astype(np.int32)).to(device)
        else:
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            extent = int(off[-1])
            d_off = torch.from_numpy(off).to(device)
        total, max_len = int(lens.sum()), int(lens.max())
        q_lo, q_hi = (1, 60) if w["L"] > 1000 else (2, 41)
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
    spliced = 0
    if ads is not None and w.get("splice") and d_off is None:
        # a quarter of the reads get one adapter at a uniform offset, truncated at the
        # read end — first-hit, hit-at-the-end and no-hit paths are all in the timed region
        L, n = w["L"], w["n"]
        pick = torch.nonzero(torch.rand(n, generator=g, device=device) < w["splice"]).flatten()
        which = torch.randint(0, len(ads), (len(pick),), generator=g, device=device)
        at = torch.randint(0, L, (len(pick),), generator=g, device=device)
        width = max(len(a) for a in ads)
        tab = torch.zeros((len(ads), width), dtype=torch.uint8, device=device)
        alen = torch.tensor([len(a) for a in ads], device=device)
        for i, a in enumerate(ads):
            tab[i, :len(a)] = torch.from_numpy(np.ascontiguousarray(a)).to(device)
        for j in range(width):
            ok = (at + j < L) & (j < alen[which])
            seq[(pick * L + at + j)[ok]] = tab[which[ok], j]
        spliced = int(len(pick))
    return dict(seq=seq, qual=qual, d_off=d_off, d_len=d_len, total=total, max_len=max_len, extent=extent,
                n=w["n"], spliced=spliced, stride=w.get("stride") if w["ragged"] else None)


def synthetic_adapter_bits(np, seed=3):
    """config 3's adapter FASTA: 24 records of 30-60 nt -> 2^20-bit table via
    the product's read_adapters rule (quack_amd.host qkh_adapter_insert)"""
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


def alg_bytes_of(b):
    if b.get("stride"):
        return 2.0 * b["total"] + 4.0 * b["n"]          # strided: 4 B/read of lengths
    return 2.0 * b["total"] + ((12.0 if b["d_len"] is not None else 8.0) * b["n"] if b["d_off"] is not None else 0.0)


def host_sample(np, b, w, budget_bases):
    """the first reads of the GPU batch, packed, on the host: (seq, qual, offsets or None, reads, bases)"""
    n = b["n"]
    if b.get("stride"):   # strided on the device: the oracle takes the same reads packed
        st = b["stride"]
        lens = b["d_len"].cpu().numpy().astype(np.int64)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        m = int(min(n, max(1, np.searchsorted(off, budget_bases))))
        gs, gq = b["seq"][:m * st].cpu().numpy().reshape(m, st), b["qual"][:m * st].cpu().numpy().reshape(m, st)
        keep = np.arange(st)[None, :] < lens[:m, None]
        return gs[keep], gq[keep], off[:m + 1], m, int(off[m])
    if b["d_off"] is None:
        m = min(n, max(1, budget_bases // w["L"]))
        return b["seq"][:m * w["L"]].cpu().numpy(), b["qual"][:m * w["L"]].cpu().numpy(), None, m, m * w["L"]
    if b["d_len"] is not None:   # gapped on the device: the oracle takes the same reads packed
        starts, lens = b["d_off"].cpu().numpy().astype(np.int64), b["d_len"].cpu().numpy().astype(np.int64)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        m = int(min(n, max(1, np.searchsorted(off, budget_bases))))
        end = int(starts[m - 1] + lens[m - 1])
        gs, gq = b["seq"][:end].cpu().numpy(), b["qual"][:end].cpu().numpy()
        keep = np.zeros(end, dtype=bool)
        for a, l in zip(starts[:m], lens[:m]):
            keep[a:a + l] = True
        return gs[keep], gq[keep], off[:m + 1], m, int(off[m])
    off = b["d_off"].cpu().numpy().astype(np.uint64)
    m = int(min(n, max(1, np.searchsorted(off, budget_bases))))
    return b["seq"][:int(off[m])].cpu().numpy(), b["qual"][:int(off[m])].cpu().numpy(), off[:m + 1], m, int(off[m])


def cpu_baselines(np, b, w, ads, threads=True, budget=3_000_000_000):
    """the oracle over (a bounded sample of) the same bytes: one core, then every core we may use"""
    import threading
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    kmers = ob.kmers_from_seqs([bytes(a) for a in ads]) if ads is not None else None
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    T = max(1, cores)
    # one thread: ~3 Gbases (~12 s at 0.25 Gbases/s); T threads: up to the whole batch, twice
    hs, hq, off, m, bases = host_sample(np, b, w, budget if not threads else 1 << 62)
    L = w["L"]
    m1 = m
    if threads:   # the single-thread leg takes a prefix of the sample
        if off is None:
            m1 = min(m, max(1, budget // L))
        else:
            m1 = int(min(m, max(1, np.searchsorted(off, budget))))

    def run(lo, hi):
        if off is None:
            ob.accumulate_batch(hs[lo * L:hi * L], hq[lo * L:hi * L], read_len=L, kmers=kmers)
        else:
            a, e = int(off[lo]), int(off[hi])
            ob.accumulate_batch(hs[a:e], hq[a:e], (off[lo:hi + 1] - off[lo]), kmers=kmers)

    t0 = time.perf_counter()
    run(0, m1)
    dt = time.perf_counter() - t0
    b1 = m1 * L if off is None else int(off[m1])
    what = "%d of %d reads" % (m1, b["n"]) + (" x %d bp" % L if off is None else " (ragged)")
    one = {"value": b1 / dt, "unit": "bases/s", "cores": 1, "kind": "port",
           "sample": what + " (same bytes as the GPU batch)", "seconds": round(dt, 3),
           "host": "oracle/quack_oracle.c, single thread (quack is single-threaded)"}
    if not threads:
        return one, None
    # N threads: the sample cut into T contiguous shares, one oracle table per thread (ctypes releases the
    # GIL); the merge of T small tables is not timed (microseconds)
    passes = 2
    cuts = [m * i // T for i in range(T + 1)]
    t0 = time.perf_counter()
    for _ in range(passes):
        th = [threading.Thread(target=run, args=(cuts[i], cuts[i + 1])) for i in range(T) if cuts[i + 1] > cuts[i]]
        for t in th:
            t.start()
        for t in th:
            t.join()
    dt = time.perf_counter() - t0
    many = {"value": passes * bases / dt, "unit": "bases/s", "cores": T, "nproc": os.cpu_count(), "kind": "port",
            "sample": "%d passes over %d of %d reads, cut into %d contiguous shares" % (passes, m, b["n"], T),
            "seconds": round(dt, 3), "host": "oracle/quack_oracle.c, one table per thread"}
    return one, many


def time_workload(torch, quack_amd, w, b, local, bits, steps, warmup, stream=None):
    """W + K device-resident passes of one workload on a fresh accumulator -> (roofline dict, table sums)"""
    acc = quack_amd.Accumulator(local, bits, max_len_hint=b["max_len"])

    def step():
        if b.get("stride"):
            acc.submit_device_strided(b["seq"], b["qual"], b["d_len"], b["n"], b["stride"], b["max_len"], stream=stream)
        elif b["d_len"] is not None:
            acc.submit_device_gapped(b["seq"], b["qual"], b["d_off"], b["d_len"], b["n"], b["extent"], b["max_len"],
                                     aligned=True, stream=stream)
        else:
            acc.submit_device(b["seq"], b["qual"], b["d_off"], b["n"], b["total"], b["max_len"], stream=stream)

    for _ in range(warmup):
        step()
    acc.sync()
    torch.cuda.synchronize()
    acc.timing(min(TIMING_EVERY, max(1, steps // 5)))   # (at least five timed launches)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    acc.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    hist_ms, batch_ms, launches = acc.timing_read_batch()
    sd = acc.finish()
    acc.close()
    got = int(sd.bases[:, 91:95].sum())
    if got != (warmup + steps) * b["total"]:
        raise SystemExit("counter check failed (%s): content sum %d != %d" % (w["label"], got, (warmup + steps) * b["total"]))
    return elapsed, hist_ms / max(launches, 1), batch_ms / max(launches, 1), launches, sd


def roofline_of(alg_bytes, kernel_ms, batch_ms, launches, traffic):
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    whole = alg_bytes / (batch_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": TRAFFIC_SOURCE if traffic else None,
            "kernel": "qk::hist_kernel", "kernel_ms": kernel_ms,
            "batch_ms": batch_ms, "frac_whole_batch": whole / HBM_PEAK_GBS,
            # ("also quote vs the 6.29 TB/s measured-copy ceiling" of MI355X_MICROARCH.md)
            "frac_of_measured_copy_ceiling": achieved / HBM_COPY_CEILING_GBS,
            "batch_kernels": "every kernel of a step on the launch stream (reach pre-pass, first-hit reset, hist_kernel, adapter count)",
            "algorithmic_bytes_per_launch": alg_bytes, "launches_timed": launches}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    # stdout carries exactly ONE JSON line: anything a library prints there (RCCL logs its
    # version banner and NCCL_DEBUG output to stdout) goes to stderr instead
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import quack_amd   # fails loudly when the native libraries are missing
    from quack_amd import distributed as qd

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.device is not None:
        local = args.device
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    name = args.workload if args.workload != "auto" else ("cfg2" if world == 1 else "cfg4")
    w = dict(WORKLOADS[name])
    if args.reads:
        w["n"] = args.reads
    if args.read_len and not w["ragged"]:
        w["L"] = args.read_len
        w["label"] += " [read length overridden: %d]" % args.read_len
    bits, ads = synthetic_adapter_bits(np) if w["adapters"] else (None, None)
    b = make_batch(torch, np, w, seed=2 + rank, device=device, quality=args.quality, ads=ads)
    seq, qual, d_off, d_len = b["seq"], b["qual"], b["d_off"], b["d_len"]
    total, max_len, extent, n = b["total"], b["max_len"], b["extent"], b["n"]
    alg_bytes = alg_bytes_of(b)

    acc = quack_amd.Accumulator(local, bits, max_len_hint=max_len)
    mate = None
    if w.get("paired"):
        # the reverse mate: its own batch and its own accumulator
        b2 = make_batch(torch, np, w, seed=1000 + rank, device=device, quality=args.quality, q_hi_override=30)
        seq2, qual2 = b2["seq"], b2["qual"]
        mate = quack_amd.Accumulator(local, bits, max_len_hint=max_len)

    # paired: both mates on ONE stream, so that every launch has the GPU to itself and its
    # HIP-event duration means something (on separate streams the two kernels would overlap)
    side = torch.cuda.Stream(device) if mate is not None else None   # (the default stream's handle is NULL)
    shared_stream = side.cuda_stream if side is not None else None

    def step():
        if b.get("stride"):
            acc.submit_device_strided(seq, qual, d_len, n, b["stride"], max_len, stream=shared_stream)
        elif d_len is not None:
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

    via_host = args.backend == "gloo"
    for _ in range(args.warmup):
        step()
    if world > 1:
        # warm the exchange too (communicator, collective kernels) — on a throwaway
        # accumulator, so that the measured tables stay the sum of exactly W+K steps
        with quack_amd.Accumulator(local, None, max_len_hint=max_len) as tmp:
            qd.allreduce_accumulator(tmp, via_host=via_host)
    fence()
    # HIP events around every 8th batch: the events themselves take ~10 us of stream time per batch (2 % of a
    # config-2 step), and `value` is this loop's wall clock
    acc.timing(TIMING_EVERY)
    if mate is not None:
        mate.timing(TIMING_EVERY)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        qd.allreduce_accumulator(acc, via_host=via_host)   # the path's single exchange (RCCL over xGMI)
        if mate is not None:
            qd.allreduce_accumulator(mate, via_host=via_host)
    fence()
    elapsed = time.perf_counter() - t0
    my_elapsed = elapsed
    kernel_ms, batch_ms, launches = acc.timing_read_batch()
    mates = 2 if mate is not None else 1
    if mate is not None:
        ms2, bms2, l2 = mate.timing_read_batch()
        kernel_ms, batch_ms, launches = kernel_ms + ms2, batch_ms + bms2, launches + l2
    per_rank = None
    if world > 1:
        cd = device if args.backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=cd)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([kernel_ms / max(launches, 1), my_elapsed / args.steps * 1e3, float(local)],
                            dtype=torch.float64, device=cd)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": i, "device": int(x[2].item()), "kernel_ms": round(float(x[0].item()), 4),
                     "ms_per_step": round(float(x[1].item()), 4)} for i, x in enumerate(allr)]

    # sanity: the counters must add up (every base carries one score and one content bin)
    sd = acc.finish()
    # after the all-reduce every rank holds the sum over ranks (equal batch sizes for fixed-length workloads)
    got = int(sd.bases[:, 91:95].sum())
    if d_off is None and not b.get("stride"):
        expect = (args.warmup + args.steps) * total * world
        if got != expect:
            raise SystemExit("counter check failed: content sum %d != %d" % (got, expect))
        if sd.number_of_sequences != (args.warmup + args.steps) * n * world:
            raise SystemExit("counter check failed: %d sequences" % sd.number_of_sequences)
    elif world == 1 and got != (args.warmup + args.steps) * total:
        raise SystemExit("counter check failed: content sum %d" % got)
    acc.close()
    if mate is not None:
        sd2 = mate.finish()
        if int(sd2.bases[:, 91:95].sum()) != (args.warmup + args.steps) * total * world:
            raise SystemExit("counter check failed for the reverse mate")
        mate.close()

    if rank == 0:
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        traffic_tab = json.load(open(tf)) if os.path.exists(tf) else {}
        kms, bms = kernel_ms / max(launches, 1), batch_ms / max(launches, 1)
        out = {
            "metric": METRIC,
            "value": world * args.steps * total * mates / elapsed,
            "unit": "bases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": w["label"] + ("" if args.quality == "uniform" else " [quality: %s]" % args.quality),
                       "reads_per_gpu": n * mates, "bases_per_gpu_per_step": total * mates,
                       "resident": "HBM", "parallelism": "batch-sharded x%d, one all-reduce of u64 tables%s" % (
                           world, " per mate" if mate is not None else "")},
            "roofline": roofline_of(alg_bytes, kms, bms, launches, traffic_tab.get(name)),
        }
        if world > 1:
            out["ranks"] = {"world_size": dist.get_world_size(), "backend": "rccl" if args.backend == "nccl" else "gloo (rehearsal)",
                            "exchange": "%d x all-reduce(SUM, u64) of %d words, inside the timed region" % (
                                mates, 97 * ((max_len + 63) // 64 * 64) + 1),
                            "per_rank": per_rank}
        if world == 1 and args.workload == "auto" and not args.no_also:
            also = {}
            for nm in ("cfg3", "cfg5", "trimmed"):
                w2 = dict(WORKLOADS[nm])
                bits2, ads2 = synthetic_adapter_bits(np) if w2["adapters"] else (None, None)
                bb = make_batch(torch, np, w2, seed={"cfg3": 3, "cfg5": 6, "trimmed": 7}[nm], device=device, ads=ads2)
                el, kms2, bms2, l2, _ = time_workload(torch, quack_amd, w2, bb, local, bits2, args.also_steps, 15)
                entry = {"workload": w2["label"], "steps": args.also_steps, "value": args.also_steps * bb["total"] / el,
                         "unit": "bases/s", "ms_per_step": el / args.also_steps * 1e3,
                         "roofline": roofline_of(alg_bytes_of(bb), kms2, bms2, l2, traffic_tab.get(nm))}
                if nm == "cfg3":
                    entry["reads_with_spliced_adapter"] = bb["spliced"]
                if not args.no_cpu_baseline:
                    entry["cpu_baseline"], _ = cpu_baselines(np, bb, w2, ads2, threads=False, budget=1_500_000_000)
                also[nm] = entry
                del bb
                torch.cuda.empty_cache()
            out["also"] = also
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], out["cpu_baseline_threads"] = cpu_baselines(np, b, w, ads)
        print(json.dumps(out), file=json_out, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

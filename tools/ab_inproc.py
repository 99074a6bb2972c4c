"""Two (or more) environment settings of the shim compared INSIDE one process, alternating every few launches —
the boxes drift between two states for seconds at a time (profiles/r03_ab_prepass.log), which process-level
alternation (env_abc.sh) only averages out over many rounds.  Works for every knob the shim reads per submit.
   python tools/ab_inproc.py [--block 10] [--rounds 30] <workload> - "QUACK_HIP_X=1" "QUACK_HIP_X=2 QUACK_HIP_Y=3"      (- = no variables)
Prints, per setting: mean / median of the per-launch histogram-kernel time and of the whole batch (HIP events)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("--block", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=30)
    ap.add_argument("--warm", type=int, default=80)
    ap.add_argument("settings", nargs="*")
    a = ap.parse_args()
    settings = [("" if s == "-" else s) for s in a.settings] or ["", ""]
    import numpy as np
    import torch
    import bench
    import quack_amd
    w = dict(bench.WORKLOADS[a.workload])
    ads_bits, ads = bench.synthetic_adapter_bits(np) if w["adapters"] else (None, None)
    b = bench.make_batch(torch, np, w, seed={"cfg3": 3, "cfg5": 6, "trimmed": 7}.get(a.workload, 2), device="cuda:0", quality="uniform", ads=ads)
    job = bench.Job.__new__(bench.Job)
    keys = sorted({kv.split("=")[0] for s in settings for kv in s.split()})
    res = {s: ([], []) for s in settings}
    with quack_amd.Accumulator(0, ads_bits, max_len_hint=b["max_len"]) as acc:
        for _ in range(a.warm):
            bench.Job.submit(job, acc, b, None)
        acc.sync()
        for r in range(a.rounds):
            for s in (settings if r % 2 == 0 else settings[::-1]):
                for k in keys:
                    os.environ.pop(k, None)
                for kv in s.split():
                    k, v = kv.split("=")
                    os.environ[k] = v
                acc.timing(1)
                for _ in range(a.block):
                    bench.Job.submit(job, acc, b, None)
                acc.sync()
                k_ms, b_ms, n = acc.timing_read_batch()
                res[s][0].append(k_ms / n)
                res[s][1].append(b_ms / n)
        acc.finish()
    for s in settings:
        k, bb = sorted(res[s][0]), sorted(res[s][1])
        print("%-44s blocks %3d x %d  kernel mean %.4f median %.4f min %.4f | batch mean %.4f median %.4f ms"
              % (s or "(default)", len(k), a.block, sum(k) / len(k), k[len(k) // 2], k[0], sum(bb) / len(bb), bb[len(bb) // 2]))


if __name__ == "__main__":
    main()

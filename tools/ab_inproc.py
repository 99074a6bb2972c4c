"""Two (or more) environment settings of the shim compared INSIDE one process, alternating every few launches —
the boxes drift between two states for seconds at a time (profiles/r03_ab_prepass.log), which process-level
alternation (env_abc.sh) only averages out over many rounds.  One accumulator per setting, created and fed under
that setting's variables; LIB=<path> as a "variable" sends the setting's launches through another build of the shim.
   python tools/ab_inproc.py [--block 10] [--rounds 30] <workload> - "QUACK_HIP_X=1" "LIB=tools/exp/e2.so QUACK_HIP_Y=3"   (- = nothing set)
Prints, per setting: mean / median of the per-launch histogram-kernel time and of the whole batch (HIP events), and
checks that every setting ended with the same counters."""
import argparse
import ctypes
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
    ap.add_argument("--shared", action="store_true", help="ONE accumulator for all settings (knobs read per submit only): "
                    "where an accumulator's buffers land is worth up to 2 %% on some boxes")
    ap.add_argument("--dummy-streams", type=int, default=0, help="streams created (and kept) before the first accumulator")
    ap.add_argument("--dummy-mb", type=int, default=0, help="device memory allocated (and kept) before the first accumulator")
    ap.add_argument("--splice", type=float, default=None, help="share of the reads that carry a spliced adapter (adapter workloads)")
    ap.add_argument("--read-len", type=int, default=None)
    ap.add_argument("--reads", type=int, default=None)
    ap.add_argument("settings", nargs="*")
    a = ap.parse_args()
    settings = [("" if s == "-" else s) for s in a.settings] or ["", ""]
    import numpy as np
    import torch
    import bench
    import quack_amd
    w = dict(bench.WORKLOADS[a.workload])
    if a.splice is not None:
        w["splice"] = a.splice
    if a.reads:
        w["n"] = a.reads
    if a.read_len:
        w["L"] = a.read_len
        if w.get("pad"):
            w["pad"] = (a.read_len + 3) & ~3
    ads_bits, ads = bench.synthetic_adapter_bits(np) if w["adapters"] else (None, None)
    b = bench.make_batch(torch, np, w, seed={"cfg3": 3, "cfg5": 6, "trimmed": 7}.get(a.workload, 2), device="cuda:0", quality="uniform", ads=ads)
    job = bench.Job.__new__(bench.Job)
    keys = sorted({kv.split("=")[0] for s in settings for kv in s.split()} - {"LIB", "MODE"})
    libs = {}

    def enter(s):
        """the environment of setting s; -> the library its launches go through (LIB=<path>: another build of the shim,
        loaded beside the product's with its own symbols first)"""
        for k in keys:
            os.environ.pop(k, None)
        lib = None
        mode = 0
        for kv in s.split():
            k, v = kv.split("=", 1)
            if k == "MODE":   # ablation builds (-DQK_ABLATION): 1 loads only, 2 quality only, 3 bases only
                mode = int(v)
            elif k == "LIB":
                if v not in libs:
                    from quack_amd import _capi
                    libs[v] = _capi.bind_hip(ctypes.CDLL(os.path.abspath(v), mode=os.RTLD_LOCAL | os.RTLD_DEEPBIND))
                lib = libs[v]
            else:
                os.environ[k] = v
        if lib is not None and hasattr(lib, "qk_debug_set_mode"):
            lib.qk_debug_set_mode(mode)
        return lib

    res = {s: ([], []) for s in settings}
    accs = {}
    dummies = [torch.cuda.Stream() for _ in range(a.dummy_streams)]
    ballast = torch.empty(a.dummy_mb << 20, dtype=torch.uint8, device="cuda") if a.dummy_mb else None
    for s in settings:   # one accumulator per setting, created under it (some knobs are read at creation)
        lib = enter(s)
        if a.shared and accs:
            if lib is not None:
                raise SystemExit("--shared cannot mix libraries")
            accs[s] = accs[settings[0]]
            continue
        accs[s] = quack_amd.Accumulator(0, ads_bits, max_len_hint=b["max_len"], _lib=lib)
    for s in settings:
        enter(s)
        for _ in range(a.warm // len(settings) + 1):
            bench.Job.submit(job, accs[s], b, None)
        accs[s].sync()
    for r in range(a.rounds):
        for s in (settings if r % 2 == 0 else settings[::-1]):
            enter(s)
            acc = accs[s]
            acc.timing(1)
            for _ in range(a.block):
                bench.Job.submit(job, acc, b, None)
            acc.sync()
            k_ms, b_ms, n = acc.timing_read_batch()
            res[s][0].append(k_ms / n)
            res[s][1].append(b_ms / n)
    tables = {}
    for s in settings:
        enter(s)
        if a.shared and s != settings[0]:
            continue
        sd = accs[s].finish()
        tables[s] = (sd.bases.copy(), sd.number_of_sequences)
        accs[s].close()
    if a.shared:
        settings_checked = settings[:1]
    else:
        settings_checked = settings
    first = tables[settings[0]]
    for s in settings_checked[1:]:   # every setting saw the same batches the same number of times
        same = tables[s][1] == first[1] and tables[s][0].shape == first[0].shape and bool((tables[s][0] == first[0]).all())
        if not same:
            print("!! the counters of %r differ from those of %r" % (s, settings[0]))
    for s in settings:
        k, bb = sorted(res[s][0]), sorted(res[s][1])
        print("%-44s blocks %3d x %d  kernel mean %.4f median %.4f min %.4f | batch mean %.4f median %.4f ms"
              % (s or "(default)", len(k), a.block, sum(k) / len(k), k[len(k) // 2], k[0], sum(bb) / len(bb), bb[len(bb) // 2]))


if __name__ == "__main__":
    main()

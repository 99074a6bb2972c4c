#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the small files
committed under profiles/:  <tag>_kernel_stats.csv, <tag>_pmc.json,
<tag>_summary.md, and the per-workload entry of profiles/hbm_traffic.json that
bench.py reports as roofline.traffic.

Usage: tools/summarize_profile.py <tag> <workload> <algorithmic bytes per launch> <bases per launch> [kernel]

HBM traffic per launch = 2 * FETCH_SIZE + WRITE_SIZE   (KB -> bytes)
  - FETCH_SIZE / WRITE_SIZE come from separate --pmc passes (TCC slots);
  - gfx950 tallies the 128-B requests of a coalesced stream at 64 B, so
    FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section);
  - the guide calibrates that factor for 16 B/lane loads only; this kernel
    loads 8 B/lane, so the factor is checked against a loads-only build of the
    same kernel that reads every byte exactly once (pmc_calib).
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(files):
    """gpurun merges the files of several calls into one directory: keep the latest run's"""
    files = sorted(files, key=os.path.getmtime)
    return files[-1:]


def counters(d, kern):
    out = collections.defaultdict(list)
    for f in newest(glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                out[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}, {k: len(v) for k, v in out.items()}


def main():
    tag, workload, alg_bytes, bases = sys.argv[1], sys.argv[2], float(sys.argv[3]), float(sys.argv[4])
    kern = sys.argv[5] if len(sys.argv) > 5 else "hist_kernel"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    rows = []
    for f in newest(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))):
        for r in csv.DictReader(open(f)):
            name = r["Name"]
            r["Name"] = name if len(name) < 100 else name[:60] + "...<%d chars>" % len(name)
            rows.append(r)
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    k = [r for r in rows if kern in r["Name"]][0]
    pmc, cnt = {}, {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_tcc", "pmc_tcc2"):
        a, b = counters(os.path.join(src, sub), kern)
        pmc.update(a)
        cnt.update(b)
    # SQ passes: instruction mix per wave and 8-position chunk (a lane owns 8 positions of a read per step),
    # VALU / LDS-issue load of the SIMDs, LDS bank conflicts
    sq = {}
    for sub in ("sq1", "sq2", "sq3"):
        a, _ = counters(os.path.join(src, sub), kern)
        sq.update(a)
    sq_derived = {}
    if sq.get("SQ_INSTS_VALU"):
        wave_chunks = bases / 8.0 / 64.0
        # SQ_BUSY_CYCLES sums the 32 shader engines (measured: 32 x kernel cycles); a wave64 VALU
        # instruction occupies its SIMD for 4 cycles; 256 CUs x 4 SIMDs
        cycles = sq.get("SQ_BUSY_CYCLES", 0) / 32.0
        sq_derived = {
            "valu_insts_per_wave_chunk": sq["SQ_INSTS_VALU"] / wave_chunks,
            "salu_insts_per_wave_chunk": sq.get("SQ_INSTS_SALU", 0) / wave_chunks,
            "lds_insts_per_wave_chunk": sq.get("SQ_INSTS_LDS", 0) / wave_chunks,
            "vmem_rd_insts_per_wave_chunk": sq.get("SQ_INSTS_VMEM_RD", 0) / wave_chunks,
            "kernel_cycles": cycles,
            "clock_GHz_implied": cycles / float(k["AverageNs"]) if cycles else None,
            "valu_busy_share_of_simd_cycles": (sq["SQ_INSTS_VALU"] * 4.0 / 1024.0 / cycles) if cycles else None,
            "lds_bank_conflict_share": (sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_LDS_IDX_ACTIVE"]) if sq.get("SQ_LDS_IDX_ACTIVE") else None,
            "wait_inst_any_over_wave_cycles": (sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"]) if sq.get("SQ_WAVE_CYCLES") and sq.get("SQ_WAIT_INST_ANY") else None,
        }
    # the whole step on the GPU's own clock: the period between the ENDS of consecutive histogram-kernel dispatches in the kernel
    # trace (every kernel of a step lies between them, whichever stream it ran on), and what else ran in a step
    step = {}
    for f in newest(glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))):
        disp = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda x: x[1])
        ends = [e for s_, e, n in disp if kern in n]
        if len(ends) > 60:
            per = sorted(b_ - a_ for a_, b_ in zip(ends[50:-1], ends[51:]))      # (past the warm-up)
            own = sorted(e - s_ for s_, e, n in disp if kern in n)[len(ends) // 10:]
            others = collections.defaultdict(list)
            for s_, e, n in disp:
                if kern not in n and "qk::" in n:     # (the product's own kernels; torch's batch-making kernels precede the loop)
                    others[n.split("(")[0][-60:]].append(e - s_)
            step = {"dispatches": len(ends), "period_ns_median": per[len(per) // 2], "period_ns_mean": sum(per) / len(per),
                    "hist_kernel_ns_median": own[len(own) // 2],
                    "other_kernels_per_step": {n: {"calls": len(v), "avg_ns": sum(v) / len(v), "min_ns": min(v)} for n, v in others.items()}}
    calib, _ = counters(os.path.join(src, "pmc_calib"), kern)
    fetch_b = pmc.get("FETCH_SIZE", 0) * 1024
    write_b = pmc.get("WRITE_SIZE", 0) * 1024
    traffic = 2 * fetch_b + write_b
    info = {"tag": tag, "workload": workload, "kernel": k["Name"], "calls": int(k["Calls"]),
            "avg_ns": float(k["AverageNs"]), "min_ns": float(k["MinNs"]), "max_ns": float(k["MaxNs"]),
            "algorithmic_bytes_per_launch": alg_bytes, "pmc_per_launch": pmc, "pmc_samples": cnt,
            "hbm_bytes_per_launch": traffic, "traffic_over_algorithmic": traffic / alg_bytes,
            "achieved_GBs_from_trace": alg_bytes / float(k["AverageNs"]),
            "sq_per_launch": sq, "sq_derived": sq_derived, "step_from_kernel_trace": step}
    if calib:
        info["calibration_loads_only_FETCH_SIZE_KB"] = calib.get("FETCH_SIZE")
        info["calibration_factor_for_8B_per_lane"] = alg_bytes / (calib["FETCH_SIZE"] * 1024)
    json.dump(info, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1)
    tf = os.path.join(dst, "hbm_traffic.json")
    t = json.load(open(tf)) if os.path.exists(tf) else {}
    t[workload] = traffic
    json.dump(t, open(tf, "w"), indent=1)
    with open(os.path.join(dst, tag + "_summary.md"), "w") as f:
        f.write("# rocprofv3 summary %s (%s)\n\n" % (tag, workload))
        f.write("Command: `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline%s` (200 steps + 50 warm-up; PMC passes: `--steps 20 --warmup 3`)\n\n"
                % ("" if workload == "cfg2" else " --workload " + workload))
        f.write("| kernel | calls | avg µs | min µs | max µs | algorithmic GB/s |\n|---|---|---|---|---|---|\n")
        f.write("| `%s` | %d | %.1f | %.1f | %.1f | %.0f |\n\n" % (k["Name"], info["calls"], info["avg_ns"] / 1e3,
                info["min_ns"] / 1e3, info["max_ns"] / 1e3, info["achieved_GBs_from_trace"]))
        f.write("PMC (separate passes, per launch): FETCH_SIZE %.0f KB, WRITE_SIZE %.0f KB, TCC_EA0_RDREQ %.0f "
                "(32B: %.0f), TCC_HIT %.0f, TCC_MISS %.0f, TCC_EA0_ATOMIC %.0f\n\n"
                % (pmc.get("FETCH_SIZE", 0), pmc.get("WRITE_SIZE", 0), pmc.get("TCC_EA0_RDREQ_sum", 0),
                   pmc.get("TCC_EA0_RDREQ_32B_sum", 0), pmc.get("TCC_HIT_sum", 0), pmc.get("TCC_MISS_sum", 0),
                   pmc.get("TCC_EA0_ATOMIC_sum", 0)))
        f.write("HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE = %.3f GB = %.3f x algorithmic (%.3f GB)\n"
                % (traffic / 1e9, traffic / alg_bytes, alg_bytes / 1e9))
        if sq_derived:
            d = sq_derived
            f.write("\nSQ counters (separate passes): %.1f VALU, %.1f SALU, %.1f LDS, %.2f VMEM-read instructions per wave and "
                    "8-position chunk; VALU instructions x 4 cycles = %.0f %% of the SIMD cycles — an upper bound: SQ_ACTIVE_INST_VALU counts "
                    "instructions, and half of the integer ones issue in 2 (tools/instr_rate.hip) — (%.0f kernel cycles = %.2f GHz); LDS bank-conflict "
                    "share %.0f %%; waves waiting to issue %.0f %% of their cycles\n"
                    % (d["valu_insts_per_wave_chunk"], d["salu_insts_per_wave_chunk"], d["lds_insts_per_wave_chunk"],
                       d["vmem_rd_insts_per_wave_chunk"], 100 * (d["valu_busy_share_of_simd_cycles"] or 0), d["kernel_cycles"],
                       d["clock_GHz_implied"] or 0, 100 * (d["lds_bank_conflict_share"] or 0),
                       100 * (d["wait_inst_any_over_wave_cycles"] or 0)))
        if step:
            f.write("\nWhole step from the kernel trace's begin / end timestamps: the period between the ends of consecutive histogram-kernel dispatches is "
                    "%.1f us (median; mean %.1f) for a kernel of %.1f us (median) — algorithmic bytes / period = %.0f GB/s = %.3f of 8 TB/s for the whole step.  "
                    "Other kernels of a step (their trace durations are NOT additive: on the side stream they wait behind the histogram kernel): %s\n"
                    % (step["period_ns_median"] / 1e3, step["period_ns_mean"] / 1e3, step["hist_kernel_ns_median"] / 1e3, alg_bytes / step["period_ns_median"],
                       alg_bytes / step["period_ns_median"] / 8000.0,
                       "; ".join("%s x%d avg %.1f us (min %.1f)" % (n, v["calls"], v["avg_ns"] / 1e3, v["min_ns"] / 1e3) for n, v in step["other_kernels_per_step"].items()) or "none"))
        if calib:
            f.write("\nCalibration (loads-only build, every byte read once, same 8 B/lane pattern): FETCH_SIZE %.0f KB "
                    "=> factor %.3f (the guide's 2.0 is for 16 B/lane)\n"
                    % (calib["FETCH_SIZE"], info["calibration_factor_for_8B_per_lane"]))
    print(open(os.path.join(dst, tag + "_summary.md")).read())


if __name__ == "__main__":
    main()

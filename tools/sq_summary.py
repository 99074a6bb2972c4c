#!/usr/bin/env python3
"""Condense the SQ_* counter passes of tools/pmc_probe.sh (gpurun_out/pmc_<tag>/p1, p2)
into one JSON object per kernel: VALU / LDS busy shares, instructions per wave
and per 8-byte chunk, LDS bank-conflict share.

    tools/sq_summary.py <tag> <chunks_per_launch> [kernel-substring] > profiles/<name>.json

chunks_per_launch = (bases per launch) / 8: the kernels own 8 positions per lane and step.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, chunks = sys.argv[1], float(sys.argv[2])
    kern = sys.argv[3] if len(sys.argv) > 3 else "hist_kernel"
    vals = collections.defaultdict(list)
    meta = {}
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_" + tag, "p*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {"vgprs": int(r["VGPR_Count"]), "sgprs": int(r["SGPR_Count"]), "lds_bytes": int(r["LDS_Block_Size"]),
                        "scratch": int(r["Scratch_Size"]), "grid": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"])}
    # the first launch of a run is the correctness pass on cold clocks: drop it when there are more
    c = {k: (sum(v[1:]) / len(v[1:]) if len(v) > 1 else v[0]) for k, v in vals.items()}
    out = {"tag": tag, "kernel": kern, "launches_averaged": {k: max(1, len(v) - 1) for k, v in vals.items()}, **meta, "counters": c}
    wave_chunks = chunks / 64.0
    d = {}
    if "SQ_INSTS_VALU" in c:
        d["valu_insts_per_wave_chunk"] = c["SQ_INSTS_VALU"] / wave_chunks
    if "SQ_INSTS_LDS" in c:
        d["lds_insts_per_wave_chunk"] = c["SQ_INSTS_LDS"] / wave_chunks
    if "SQ_INSTS_VMEM_RD" in c:
        d["vmem_rd_insts_per_wave_chunk"] = c["SQ_INSTS_VMEM_RD"] / wave_chunks
    if "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"]:
        # SQ_ACTIVE_INST_* count cycles (x4 per SIMD group on this part: MI355X_MICROARCH.md); shares of the busy cycles
        for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"):
            if k in c:
                d[k.lower() + "_per_busy_cycle"] = c[k] / c["SQ_BUSY_CYCLES"]
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_share"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    out["derived"] = d
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()

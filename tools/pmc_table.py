#!/usr/bin/env python3
"""rocprofv3 --pmc passes (tools/pmc_sam.sh: directories a, b, c under <dir>) -> one markdown table per kernel for profiles/.

usage: pmc_table.py <dir with a/ b/ c/> <out.md> [kernel substrings ...]
Columns (per launch, averaged over the launches of the pass):
  us          kernel duration in the counter pass (End - Start timestamp; counter passes serialise launches)
  MFMA busy   SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz): matrix-pipe occupancy against the NOMINAL clock
              (= achieved fraction of the dense peak, padding included; the counter is 32 cycles per 32x32x16 MFMA)
  VALU/MFMA   SQ_INSTS_VALU / SQ_INSTS_MFMA (wave instructions; SQ_INSTS_VALU includes the MFMAs)
  wait        SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES: share of wave-resident time spent waiting on any instruction dependency
  LDS confl   SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE: LDS-array cycles lost to bank conflicts
"""
import collections
import csv
import glob
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kname import canon


def clean(name):
    return canon(name)[:90]           # demangled, parameter list cut: the spelling bench.py / cvmi_last_kernel() use


def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    dur = collections.defaultdict(list)
    seen = set()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = clean(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (r["Dispatch_Id"], f)
            if key not in seen:
                seen.add(key)
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return acc, dur


def main():
    root, out = sys.argv[1], sys.argv[2]
    subs = sys.argv[3:] or ["attn_", "tok_linear", "hiera_mlp", "gemm256", "gemm_glds", "igemm"]
    a, da = load(root + "/a")
    b, _ = load(root + "/b")
    rows = []
    for k in a:
        if not any(s in k for s in subs):
            continue
        n = len(da[k])
        us = sum(da[k]) / n
        c = a[k]
        mf, busy, valu = c["SQ_INSTS_MFMA"] / n, c["SQ_VALU_MFMA_BUSY_CYCLES"] / n, c["SQ_INSTS_VALU"] / n
        wait = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"] if c["SQ_WAVE_CYCLES"] else 0
        cb = b.get(k, {})
        confl = cb.get("SQ_LDS_BANK_CONFLICT", 0) / cb["SQ_LDS_IDX_ACTIVE"] if cb.get("SQ_LDS_IDX_ACTIVE") else 0
        rows.append((us * n, k, n, us, mf, busy / (1024 * us * 1e-6 * 2.4e9), valu / mf if mf else 0, wait, confl))
    rows.sort(reverse=True)
    with open(out, "w") as f:
        f.write("# SQ counters per kernel: one eager SAM 2.1 Hiera-L B=16 forward (tools/pmc_sam.sh; rocprofv3 --pmc, separate passes a / b)\n\n")
        f.write(__doc__.split("Columns")[1].join(["Columns", ""]) + "\n")
        f.write("| kernel | launches | us / launch | MFMA insts / launch | MFMA busy (of 2.4 GHz peak) | VALU / MFMA | wait | LDS conflict |\n|---|---:|---:|---:|---:|---:|---:|---:|\n")
        for _, k, n, us, mf, util, vm, wait, confl in rows:
            f.write(f"| `{k}` | {n} | {us:.1f} | {mf:.3g} | {util:.3f} | {vm:.1f} | {wait:.2f} | {confl:.3f} |\n")
    print("wrote", out, len(rows), "kernels")


if __name__ == "__main__":
    main()

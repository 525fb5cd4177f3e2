#!/usr/bin/env python3
"""HBM traffic PER LAUNCH of the SAM 2.1 pass, from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/one_step.py sam2l.

usage: traffic_sam.py <fetch dir> <write dir> <passes profiled> <out.json> [note]
gfx950 corrections as MI355X_MICROARCH.md (HBM) prescribes: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read -> doubled;
WRITE_SIZE is exact for 16-byte-per-lane stores; both counters are in KiB; the two counters cannot share a pass (TCC slots).
One entry per (kernel, grid size): launches per pass, average duration in the counter pass and fetch / write / total bytes per launch.
bench.py looks an entry up by the kernel name cvmi_last_kernel() reports and the launches per pass."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kname import canon


def load(d, counter):
    acc = collections.defaultdict(lambda: [0.0, 0, 0.0])          # (kernel, grid) -> [KiB, dispatches, us]
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = (canon(r["Kernel_Name"]), int(r.get("Grid_Size", 0) or 0))
            a = acc[k]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
            a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return acc


def main():
    fetch_dir, write_dir, passes, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    note = " ".join(sys.argv[5:])
    fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    kernels = collections.defaultdict(list)
    tot = 0.0
    for key in sorted(set(fe) | set(wr)):
        f, w = fe.get(key, [0.0, 0, 0.0]), wr.get(key, [0.0, 0, 0.0])
        n = max(f[1], w[1])
        if n == 0:
            continue
        fb, wb = f[0] * 1024 * 2 / n, w[0] * 1024 / n
        us = (f[2] + w[2]) / max(1, f[1] + w[1])
        kernels[key[0]].append({"grid": key[1], "launches_per_pass": n // passes, "us_in_counter_pass": round(us, 2),
                                "fetch_bytes_x2": int(fb), "write_bytes": int(wb), "hbm_bytes": int(fb + wb),
                                "gbs_in_counter_pass": round((fb + wb) / max(us, 1e-9) / 1e3, 1)})
        tot += (fb + wb) * n / passes
    res = {"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/one_step.py sam2l (eager launches); FETCH_SIZE "
                     "doubled (gfx950: 64 B counted per 128-B request), WRITE_SIZE as reported; KiB -> bytes; per launch = counter sum / dispatches",
           "passes_profiled": passes, "hbm_bytes_per_pass": int(tot), "note": note, "kernels": kernels}
    json.dump(res, open(out, "w"), indent=1)
    rows = sorted(((e["hbm_bytes"] * e["launches_per_pass"], k, e) for k, es in kernels.items() for e in es), key=lambda r: -r[0])
    print(f"HBM bytes per pass {tot / 1e9:.2f} GB")
    for b, k, e in rows[:25]:
        print(f"{b / 1e6:9.1f} MB/pass  {e['launches_per_pass']:4d} x {e['hbm_bytes'] / 1e6:8.1f} MB  {e['us_in_counter_pass']:8.1f} us  {e['gbs_in_counter_pass']:7.1f} GB/s  {k}")


if __name__ == "__main__":
    main()

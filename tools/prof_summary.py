#!/usr/bin/env python3
"""Condense a rocprofv3 `--kernel-trace --stats --output-format csv` directory into a small markdown
table for profiles/ (kernel, calls, total ms, avg us, %)."""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kname import canon


def main(src, out, note=""):
    files = glob.glob(f"{src}/**/*kernel_stats.csv", recursive=True)
    assert files, f"no kernel_stats.csv under {src}"
    rows = list(csv.DictReader(open(files[0])))
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary\n\n{note}\n\n")
        f.write("| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
        for r in rows:
            name = canon(r["Name"]).replace("|", "/")          # demangled, parameter list cut: the spelling cvmi_last_kernel() / bench.py use
            if len(name) > 110:
                name = name[:107] + "..."
            f.write(f"| `{name}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |\n")
    print("wrote", out)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))

#!/usr/bin/env python
"""Condense rocprofv3 CSV output (kernel_stats / counter_collection) into a small text summary that is
committed under profiles/.  Usage: python profiles/summarize.py <rocprof out dir> <summary.txt> [label]"""
import csv
import glob
import os
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    label = sys.argv[3] if len(sys.argv) > 3 else ""
    lines = [f"# rocprofv3 summary {label}".rstrip(), f"# source dir: {src}"]
    for f in sorted(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)):
        lines.append(f"## kernel stats ({os.path.basename(f)}): name, calls, avg_us, total_ms, pct")
        rows = list(csv.DictReader(open(f)))
        for r in rows[:40]:
            name = r["Name"]
            name = name if len(name) < 90 else name[:87] + "..."
            lines.append(f"{name} | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['TotalDurationNs'])/1e6:.3f} | {r['Percentage']}")
    for f in sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)):
        lines.append(f"## counters ({os.path.basename(f)}): kernel, counter, launches, mean value per launch")
        acc = {}
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:60], r["Counter_Name"])
            a = acc.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        for (kn, cn), (n, tot) in sorted(acc.items()):
            if kn.startswith("void vc_") or kn.startswith("vc_"):
                lines.append(f"{kn} | {cn} | {n} | {tot/n:.1f}")
    open(dst, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()

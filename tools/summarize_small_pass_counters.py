#!/usr/bin/env python3
"""Condense the rocprofv3 passes of tools/collect_small_pass_counters.sh into <out>/small_pass_counters.csv:
variant, kernel, counter, dispatches, mean per dispatch — plus the kernel-trace averages (variant, kernel, calls,
average ns).  Values are as rocprofv3 reports them."""
import collections
import csv
import glob
import os
import sys


def main():
    out, variants = sys.argv[1], sys.argv[2:]
    rows = []
    for variant in variants:
        values = collections.defaultdict(list)
        for path in glob.glob(os.path.join(out, variant + "_*", "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                values[(row["Kernel_Name"], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (kernel, counter), vals in sorted(values.items()):
            rows.append([variant, kernel[:80], counter, len(vals), f"{sum(vals) / len(vals):.1f}"])
        for path in glob.glob(os.path.join(out, variant + "_trace", "**", "*kernel_stats.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                rows.append([variant, row["Name"][:80], "trace_average_ns", row["Calls"], row["AverageNs"]])
    with open(os.path.join(out, "small_pass_counters.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["variant", "kernel", "counter", "dispatches", "mean_per_dispatch"])
        w.writerows(rows)
    for r in rows:
        if "score_sed" in r[1]:
            print(",".join(str(x) for x in r))


if __name__ == "__main__":
    main()

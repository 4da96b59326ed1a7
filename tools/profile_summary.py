"""Aggregate the rocprofv3 CSVs of tools/profile_round.sh into kernel_stats.csv / pmc_counters.csv (+ the traffic file
bench.py reads).  usage: python tools/profile_summary.py gpurun_out/<dir>"""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
# ---- kernel stats: per-kernel calls / total / average duration from the kernel trace
rows = defaultdict(list)
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
tot = sum(sum(v) for v in rows.values()) or 1.0
with open(os.path.join(out, "kernel_stats.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "calls", "total_ms", "average_ms", "min_ms", "max_ms", "percent"])
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k, len(v), f"{sum(v):.4f}", f"{sum(v) / len(v):.4f}", f"{min(v):.4f}", f"{max(v):.4f}", f"{100 * sum(v) / tot:.2f}"])
# ---- PMC counters: mean per launch per (kernel, counter)
acc = defaultdict(list)
for p in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(p):
        continue
    for f in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(os.path.basename(p), r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(os.path.join(out, "pmc_counters.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["pass", "kernel", "counter", "launches", "mean_per_launch"])
    for (p, k, c), v in sorted(acc.items()):
        # one row per dispatch per counter instance: sum the instances of a dispatch (dimension rows), average over dispatches
        w.writerow([p, k, c, len(v), sum(v) / len(v)])
scan = [k for k in rows if "scan_kernel" in k]
if scan:
    main = max(scan, key=lambda k: sum(rows[k]))
    fetch = [sum(v) / len(v) for (p, k, c), v in acc.items() if k == main and c == "FETCH_SIZE"]
    write = [sum(v) / len(v) for (p, k, c), v in acc.items() if k == main and c == "WRITE_SIZE"]
    print("dominant kernel:", main, "avg ms", sum(rows[main]) / len(rows[main]), "FETCH_SIZE KB", fetch, "WRITE_SIZE KB", write)

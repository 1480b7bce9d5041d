"""Per-(kernel, grid) duration summary from a rocprofv3 --kernel-trace CSV.  bench.py launches the fused kernel at
two batch sizes (25 000 = 196 workgroups, the graded workload; 65 536 = 256 persistent workgroups, the north-star
point), which rocprofv3's own --stats table averages together; this keeps them apart.

    python tools/kernel_stats_by_grid.py KERNEL_TRACE.csv > profiles/rNN_kernel_stats_by_grid.csv
"""
import collections
import csv
import sys

rows = collections.defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows[(r["Kernel_Name"], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))].append(
            int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
w = csv.writer(sys.stdout)
w.writerow(["Name", "Workgroups", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
for (name, wg), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    w.writerow([name, wg, len(v), sum(v), round(sum(v) / len(v), 1), min(v), max(v)])

"""Per-(kernel, workgroups, counter) mean of rocprofv3 --pmc counter_collection CSVs (one or more passes).

    python tools/pmc_by_grid.py PASS1_counter_collection.csv [PASS2 ...] > profiles/rNN_pmc_by_grid.csv

FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE under-reports by 2x (MI355X_MICROARCH.md, HBM section): the
column bytes_corrected applies that."""
import collections
import csv
import sys

acc = collections.defaultdict(list)
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            wg = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
            acc[(r["Kernel_Name"], wg, r["Counter_Name"])].append(float(r["Counter_Value"]))
w = csv.writer(sys.stdout)
w.writerow(["Kernel", "Workgroups", "Counter", "Launches", "Mean", "bytes_corrected"])
for (k, wg, c), v in sorted(acc.items()):
    m = sum(v) / len(v)
    b = ""
    if c == "FETCH_SIZE":
        b = int(2.0 * m * 1024)
    elif c == "WRITE_SIZE":
        b = int(m * 1024)
    w.writerow([k.split("(")[0], wg, c, len(v), round(m, 3), b])

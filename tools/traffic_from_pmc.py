"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) into
profiles/traffic_latest.json for the fused SIREN kernel.  Units and the gfx950 correction follow
/opt/skills/guides/MI355X_MICROARCH.md (counter values in KB; FETCH_SIZE under-reports by 2x on gfx950, calibrated
on reduce_slabs_real_kernel whose reads are known exactly).

    python tools/traffic_from_pmc.py FETCH.csv WRITE.csv [round]
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mean_counter(path, counter, kernel_substr, grid=None):
    vals = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter and kernel_substr in row["Kernel_Name"]:
                if grid is None or int(row["Grid_Size"]) == grid:
                    vals.append(float(row["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main():
    fetch_csv, write_csv = sys.argv[1], sys.argv[2]
    rnd = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    k = "inr_mlp_kernel<8, 4, 1, 1, 2>"
    grid = 196 * 256  # B = 25000 -> 196 workgroups of 256 threads
    f, nf = mean_counter(fetch_csv, "FETCH_SIZE", k, grid)
    w, nw = mean_counter(write_csv, "WRITE_SIZE", k, grid)
    rf, _ = mean_counter(fetch_csv, "FETCH_SIZE", "reduce_slabs_real_kernel", 1288 * 256)
    gf, _ = mean_counter(fetch_csv, "FETCH_SIZE", "dw_gemm_kernel<128, 4>", 245 * 256)
    gw, _ = mean_counter(write_csv, "WRITE_SIZE", "dw_gemm_kernel<128, 4>", 245 * 256)
    # what the reduction reads at B = 25000: 49 GEMM slabs of the four 256-row layers, 196 fused slabs of the last
    # layer (514 floats) and of the loss word
    known = (49 * 328704 + 196 * 514 + 196) * 4
    out = {
        "kernel": "inr_mlp_kernel<8,GAUSS,SIN,FUSED>", "workload": "SIREN 5x256 gauss-512, B = 25000 (196 workgroups)",
        "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "launches_averaged": [nf, nw], "fetch_correction": 2.0,
        "bytes_per_launch": int((2.0 * f + w) * 1024),
        "dw_gemm_kernel": {"FETCH_SIZE_KB": gf, "WRITE_SIZE_KB": gw,
                           "bytes_per_launch": int((2.0 * gf + gw) * 1024) if gf is not None and gw is not None else None},
        "calibration": {"kernel": "reduce_slabs_real_kernel", "FETCH_SIZE_KB": rf, "known_read_bytes": known},
        "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes ({os.path.basename(fetch_csv)}, "
                  f"{os.path.basename(write_csv)})", "round": rnd}
    with open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w") as fp:
        json.dump(out, fp, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

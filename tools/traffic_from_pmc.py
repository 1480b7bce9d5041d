"""profiles/traffic_latest.json from a round's per-(kernel, workgroups) PMC summary (tools/pmc_round.sh -> pmc_by_grid.py):
HBM bytes per launch of the fused kernels and the weight-gradient GEMM at the two benchmark batch sizes.  Units and the
gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (counter values in KB; FETCH_SIZE x 2), calibrated on
reduce_slabs_real_kernel, whose reads are known exactly.

    python tools/traffic_from_pmc.py profiles/r04_pmc_by_grid.csv 4
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path):
    t = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            t[(r["Kernel"], int(r["Workgroups"]), r["Counter"])] = (float(r["Mean"]), int(r["Launches"]),
                                                                    int(r["bytes_corrected"] or 0))
    return t


def entry(t, kernel, wg, label=None):
    f, w = t.get((kernel, wg, "FETCH_SIZE")), t.get((kernel, wg, "WRITE_SIZE"))
    if f is None or w is None:
        return None
    return {"kernel": label or kernel, "workgroups": wg, "FETCH_SIZE_KB": f[0], "WRITE_SIZE_KB": w[0],
            "launches_averaged": [f[1], w[1]], "read_bytes": f[2], "write_bytes": w[2], "bytes_per_launch": f[2] + w[2]}


def main():
    path, rnd = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t = load(path)
    rs = entry(t, "void inr::inr_mlp_rs_kernel<7, 1>", 256, "inr_mlp_rs_kernel<7,SIN>")
    out = dict(rs or {})
    out.update({
        "round": rnd, "fetch_correction": 2.0,
        "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/pmc_round.sh -> {os.path.basename(path)}); "
                  "counter values in KB, FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section)",
        "workload": "SIREN 5x256 gauss-512, B = 25000 (256 workgroups, tiles of 7 / 6 column blocks of 16 coordinates)",
        "batch_65536": entry(t, "void inr::inr_mlp_kernel<8, 4, 1, 1, 2>", 256, "inr_mlp_kernel<8,GAUSS,SIN,FUSED>"),
        "dw_gemm_kernel": entry(t, "void inr::dw_gemm_kernel<128, 2, 4>", 250,
                                "dw_gemm_kernel<128,2,4> (B = 25000: 128 x 256 tiles, 25 chunks)"),
        "calibration": {"kernel": "reduce_slabs_real_kernel", "measured": entry(t, "inr::reduce_slabs_real_kernel", 1288),
                        # B = 25000: 25 GEMM chunk slabs of the four 256-row layers (328 704 floats), 256 workgroup slabs of
                        # the last layer (514 floats) and of the loss word
                        "known_read_bytes": (25 * 328704 + 256 * 514 + 256) * 4},
    })
    with open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w") as fp:
        json.dump(out, fp, indent=1)
    print(json.dumps(out)[:600])


if __name__ == "__main__":
    main()

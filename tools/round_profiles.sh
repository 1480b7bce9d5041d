#!/bin/bash
# Everything a round commits under profiles/ (run on the GPU box from the repo root): tools/round_profiles.sh r03
# Each step writes under gpurun_out/; a failing step stops the script (no GPU step behind a failed one).
set -o pipefail
R=${1:-r03}
export TMPDIR=/tmp
mkdir -p gpurun_out
[ -f mri-implicit-neural-representations_amd/lib/libinr_mi355x_dbg.so ] || { echo "build the diagnostic library first (here, not on the GPU box): tools/build_dbg_rs.sh inr_siren_bf16_m0 inr_siren_bf16_m1 inr_siren_bf16_m2 inr_mlp_nb8 inr_dw_gemm inr_dw_gemm_bf16"; exit 1; }
echo "== bench"; timeout -k 10 400 python bench.py > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err || { tail -5 gpurun_out/${R}_bench.err; exit 1; }
echo "== bench, the command the driver runs"; timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${R}_bench_driver_command.json 2> gpurun_out/${R}_bench_driver_command.err || { tail -5 gpurun_out/${R}_bench_driver_command.err; exit 1; }
echo "== kernel trace + PMC of the bench command"; timeout -k 10 900 bash tools/pmc_round.sh $R > gpurun_out/${R}_pmc_round.log 2>&1 || { tail -5 gpurun_out/${R}_pmc_round.log; exit 1; }
echo "== bench_models"; timeout -k 10 400 python tools/bench_models.py > gpurun_out/${R}_bench_models.json 2> gpurun_out/${R}_bench_models.err || { tail -5 gpurun_out/${R}_bench_models.err; exit 1; }
echo "== bf16 kernel durations by batch"; timeout -k 10 300 bash tools/prof_bf16.sh ${R}_bf16 65536 25000 > gpurun_out/${R}_prof_bf16.log 2>&1 || { tail -5 gpurun_out/${R}_prof_bf16.log; exit 1; }
echo "== bf16 PMC"; timeout -k 10 600 bash tools/pmc_bf16.sh ${R}_bf16 65536 > gpurun_out/${R}_pmc_bf16.log 2>&1 || { tail -5 gpurun_out/${R}_pmc_bf16.log; exit 1; }
# (afterwards, in the repository: cp gpurun_out/${R}_pmc_by_grid.csv profiles/ && python tools/traffic_from_pmc.py profiles/${R}_pmc_by_grid.csv)
echo "== stamps (phases, in-kernel clock) and launch-level stamps (entry / exit of every wave)"
for a in "65536 bf16" "25000 bf16" "65536 f32"; do
  set -- $a
  timeout -k 10 200 python3 tools/stamps.py $1 $2 > gpurun_out/${R}_stamps_$2_$1.txt 2>&1 || { tail -3 gpurun_out/${R}_stamps_$2_$1.txt; exit 1; }
done
timeout -k 10 200 python3 tools/stamps_rs.py 25000 > gpurun_out/${R}_stamps_rs_25000.txt 2>&1 || { tail -3 gpurun_out/${R}_stamps_rs_25000.txt; exit 1; }
for a in "25000 f32" "65536 f32" "25000 bf16" "65536 bf16"; do
  set -- $a
  timeout -k 10 200 python3 tools/stamps_launch.py $1 $2 > gpurun_out/${R}_launch_$2_$1.txt 2>&1 || { tail -3 gpurun_out/${R}_launch_$2_$1.txt; exit 1; }
done
echo "== rounding oracle distances"; timeout -k 10 300 python3 tests/debug_bf16_oracle.py 127 4133 32845 > gpurun_out/${R}_bf16_oracle_distances.txt 2>&1 || { tail -3 gpurun_out/${R}_bf16_oracle_distances.txt; exit 1; }
echo "== probes"
for p in overlap_probe sinf16_probe stream_probe; do  # (binaries are not in the history: built where missing)
  [ -x tools/probes/$p ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value tools/probes/$p.hip -o tools/probes/$p || exit 1
done
timeout -k 10 60 tools/probes/overlap_probe > gpurun_out/${R}_probe_overlap.txt 2>&1 && timeout -k 10 60 tools/probes/sinf16_probe > gpurun_out/${R}_probe_sinf16.txt 2>&1 && timeout -k 10 60 tools/probes/stream_probe > gpurun_out/${R}_probe_stream_run.txt 2>&1

echo "== kernel trace of the secondary workloads"
(cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${R}_models_kt -o m -- python3 $GRAFT_REPO_ROOT/tools/bench_models.py > $GRAFT_REPO_ROOT/gpurun_out/${R}_models_kt.log 2>&1) || { tail -5 gpurun_out/${R}_models_kt.log; exit 1; }
python3 tools/kernel_stats_by_grid.py "$(find gpurun_out/${R}_models_kt -name '*kernel_trace.csv' | head -1)" > gpurun_out/${R}_kernel_stats_models_by_grid.csv && rm -rf gpurun_out/${R}_models_kt
echo done

#!/bin/bash
# Diagnostic: an experiment build of the library -- the bf16 translation units recompiled with extra flags, linked with the
# shipped objects of everything else -- as lib/libinr_exp_NAME.so (loaded through INR_LIB_PATH by tools/exp_libs.sh).
# usage: tools/build_exp.sh NAME "-DFLAG ..."
set -e
NAME=$1; FLAGS=$2
cd "$(dirname "$0")/../mri-implicit-neural-representations_amd/csrc"
mkdir -p ../build_exp_$NAME
for tu in inr_siren_bf16_m0 inr_siren_bf16_m1 inr_siren_bf16_m2 inr_dw_gemm_bf16; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function $FLAGS -DINR_ONLY_NH=3 -c $tu.hip -o ../build_exp_$NAME/$tu.o &
done
wait
# no experiment object reaches a GPU with a register of an in-flight load touched (the fault of round 3: a knock-out had
# removed the waits of inline-assembly loads but not the loads)
for o in ../build_exp_$NAME/*.o; do
  python3 ../../tools/check_inflight_regs.py $o kernel || { echo "check_inflight_regs: $o fails -- not linked"; rm -rf ../build_exp_$NAME; exit 1; }
done
objs=$(ls ../build/*.o | grep -v "inr_siren_bf16_m\|inr_dw_gemm_bf16.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libinr_exp_$NAME.so $objs ../build_exp_$NAME/*.o
rm -rf ../build_exp_$NAME
echo built lib/libinr_exp_$NAME.so

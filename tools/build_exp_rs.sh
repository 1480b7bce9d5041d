#!/bin/bash
# Diagnostic: lib/libinr_exp_NAME.so = the shipped objects with the row-split kernels (n7, n8) and the API rebuilt with
# phase stamps and extra flags.  usage: tools/build_exp_rs.sh NAME "-DFLAG ..."   (run: INR_LIB_PATH=... tools/stamps_rs.py)
set -e
NAME=$1; FLAGS=$2
cd "$(dirname "$0")/../mri-implicit-neural-representations_amd/csrc"
mkdir -p ../build_exp_$NAME
for tu in inr_api inr_mlp_rs_n1 inr_mlp_rs_n7; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -DINR_STAMPS $FLAGS -c $tu.hip -o ../build_exp_$NAME/$tu.o &
done
wait
objs=$(ls ../build/*.o | grep -v "inr_api.o\|inr_mlp_rs_n1.o\|inr_mlp_rs_n7.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libinr_exp_$NAME.so $objs ../build_exp_$NAME/*.o
rm -rf ../build_exp_$NAME
echo built lib/libinr_exp_$NAME.so

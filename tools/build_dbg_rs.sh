#!/bin/bash
# Diagnostic: lib/libinr_mi355x_dbg.so with phase stamps (-DINR_STAMPS) in the API, the row-split kernels and any further
# translation units named on the command line (e.g. inr_siren_bf16_m0 inr_siren_bf16_m1 inr_siren_bf16_m2 inr_mlp_nb8),
# linked with the shipped objects of everything else (a full `make dbg` takes several minutes).
# Used by tools/stamps_rs.py and tools/stamps.py.
set -e
cd "$(dirname "$0")/../mri-implicit-neural-representations_amd/csrc"
mkdir -p ../build_dbg_rs
TUS="inr_api inr_mlp_rs_n6 inr_mlp_rs_n7 $*"
for tu in $TUS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -DINR_STAMPS -c $tu.hip -o ../build_dbg_rs/$tu.o &
done
wait
objs=$(ls ../build/*.o)
for tu in $TUS; do objs=$(echo "$objs" | grep -v "/$tu.o"); done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libinr_mi355x_dbg.so $objs ../build_dbg_rs/*.o
rm -rf ../build_dbg_rs
echo built lib/libinr_mi355x_dbg.so

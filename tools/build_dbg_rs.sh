#!/bin/bash
# Diagnostic: lib/libinr_mi355x_dbg.so with phase stamps (-DINR_STAMPS) in the row-split kernels and the API only, linked
# with the shipped objects of everything else (a full `make dbg` takes several minutes).  Used by tools/stamps_rs.py.
set -e
cd "$(dirname "$0")/../mri-implicit-neural-representations_amd/csrc"
mkdir -p ../build_dbg_rs
for tu in inr_api inr_mlp_rs_n6 inr_mlp_rs_n7; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -DINR_STAMPS -c $tu.hip -o ../build_dbg_rs/$tu.o &
done
wait
objs=$(ls ../build/*.o | grep -v "inr_api.o\|inr_mlp_rs_n6.o\|inr_mlp_rs_n7.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libinr_mi355x_dbg.so $objs ../build_dbg_rs/*.o
rm -rf ../build_dbg_rs
echo built lib/libinr_mi355x_dbg.so

for B in 16384 32768 49152 65536 98304 131072; do echo "B=$B"; bash tools/exp_libs.sh $B base 2>&1 | grep "siren_bf16\|dw_gemm\|reduce"; done

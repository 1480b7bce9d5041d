// WIRE with 65..128 complex hidden features (network_width 128 -> 90 complex): 256 interleaved rows
#define INR_NB 8
#define INR_DWG_STATIC 1  // 256 rows: dW of the hidden-width layers by inr_dw_gemm.hip, always
#define INR_NW 4
#define INR_FAMILY_WIRE 1
#define INR_LAUNCH_NAME launch_wire_nb8
#include "inr_mlp_inst.h"

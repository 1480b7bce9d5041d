// hidden width 256 (the graded SIREN 5x256 / 4x256 shapes), 4 waves = 128-coordinate tiles
#define INR_NB 8
#define INR_DWG_STATIC 1  // 256 rows: dW of the hidden-width layers by inr_dw_gemm.hip, always
#define INR_NW 4
#define INR_LAUNCH_NAME launch_mlp_nb8
#include "inr_mlp_inst.h"

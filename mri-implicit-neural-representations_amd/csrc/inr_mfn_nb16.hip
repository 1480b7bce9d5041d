// MFN, hidden width 512 (BASELINE config 4: MultiscaleKFourier 8x512).  2 waves = 64-coordinate tiles,
// 2 x (16+1) row blocks x 36 floats = 157 KB of LDS.
#define INR_NB 16
#define INR_NW 2
#define INR_LAUNCH_NAME launch_mfn_nb16
#include "inr_mfn_inst.h"

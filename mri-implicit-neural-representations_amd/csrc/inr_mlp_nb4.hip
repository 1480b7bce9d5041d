// hidden widths 65..128, 4 waves = 128-coordinate tiles
#define INR_NB 4
#define INR_NW 4
#define INR_LAUNCH_NAME launch_mlp_nb4
#include "inr_mlp_inst.h"

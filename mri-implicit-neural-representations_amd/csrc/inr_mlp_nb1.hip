// hidden width 32 (golden-vector / test shapes), 4 waves = 128-coordinate tiles
#define INR_NB 1
#define INR_NW 4
#define INR_LAUNCH_NAME launch_mlp_nb1
#include "inr_mlp_inst.h"

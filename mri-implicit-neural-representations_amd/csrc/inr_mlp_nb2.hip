// hidden widths 33..64, 4 waves = 128-coordinate tiles
#define INR_NB 2
#define INR_NW 4
#define INR_LAUNCH_NAME launch_mlp_nb2
#include "inr_mlp_inst.h"

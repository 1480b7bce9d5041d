// hidden widths 257..512.  2 waves = 64-coordinate tiles: 2 x 512 rows x 36 floats = 147 KB of LDS.
#define INR_NB 16
#define INR_NW 2
#define INR_LAUNCH_NAME launch_mlp_nb16
#include "inr_mlp_inst.h"

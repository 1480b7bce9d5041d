// WIRE2D with 129..256 complex hidden features (network_width 256 is not reduced, wire2d.py:76): 2 waves x 512 rows x 36 floats = 147 KB
#define INR_NB 16
#define INR_NW 2
#define INR_FAMILY_WIRE2D 1
#define INR_LAUNCH_NAME launch_wire2d_nb16
#include "inr_mlp_inst.h"

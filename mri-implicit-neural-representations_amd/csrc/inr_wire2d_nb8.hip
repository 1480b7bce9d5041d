// WIRE2D with 65..128 complex hidden features
#define INR_NB 8
#define INR_NW 4
#define INR_FAMILY_WIRE2D 1
#define INR_LAUNCH_NAME launch_wire2d_nb8
#include "inr_mlp_inst.h"

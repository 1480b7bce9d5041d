// WIRE2D with up to 16 complex hidden features (64 interleaved rows incl. padding)
#define INR_NB 2
#define INR_NW 4
#define INR_FAMILY_WIRE2D 1
#define INR_LAUNCH_NAME launch_wire2d_nb2
#include "inr_mlp_inst.h"

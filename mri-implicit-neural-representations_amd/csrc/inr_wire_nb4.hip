// WIRE with 17..64 complex hidden features (network_width <= 90): 128 interleaved rows
#define INR_NB 4
#define INR_NW 4
#define INR_FAMILY_WIRE 1
#define INR_LAUNCH_NAME launch_wire_nb4
#include "inr_mlp_inst.h"

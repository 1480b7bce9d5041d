// WIRE, 32 complex hidden features = 64 interleaved real rows (golden-vector / test shapes)
#define INR_NB 2
#define INR_NW 4
#define INR_FAMILY_WIRE 1
#define INR_LAUNCH_NAME launch_wire_nb2
#include "inr_mlp_inst.h"

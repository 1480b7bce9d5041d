// hidden width 32 (golden-vector / test shapes)
#define INR_NB 1
#define INR_LAUNCH_NAME launch_mlp_nb1
#include "inr_mlp_inst.h"

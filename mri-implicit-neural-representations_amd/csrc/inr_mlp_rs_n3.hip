// row-split fused step (inr_mlp_rs_impl.h), tiles of 3 column blocks of 16 coordinates
#define INR_RS_NCB 3
#include "inr_mlp_rs_inst.h"

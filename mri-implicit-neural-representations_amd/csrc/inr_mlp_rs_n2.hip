// row-split fused step (inr_mlp_rs_impl.h), tiles of 2 column blocks of 16 coordinates
#define INR_RS_NCB 2
#include "inr_mlp_rs_inst.h"

// row-split fused step (inr_mlp_rs_impl.h), tiles of 7 column blocks of 16 coordinates
#define INR_RS_NCB 7
#include "inr_mlp_rs_inst.h"

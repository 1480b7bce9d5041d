// row-split fused step (inr_mlp_rs_impl.h), tiles of 1 column block of 16 coordinates
#define INR_RS_NCB 1
#include "inr_mlp_rs_inst.h"

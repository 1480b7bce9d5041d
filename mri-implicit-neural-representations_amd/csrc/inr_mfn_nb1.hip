// MFN, hidden width 32 (golden-vector / test shapes)
#define INR_NB 1
#define INR_NW 4
#define INR_LAUNCH_NAME launch_mfn_nb1
#include "inr_mfn_inst.h"

// MFN, hidden width 256.  4 waves x (8+1) row blocks: 33-float rows keep the images at 152 KB.
#define INR_LDS_LD 33
#define INR_NB 8
#define INR_NW 4
#define INR_LAUNCH_NAME launch_mfn_nb8
#include "inr_mfn_inst.h"

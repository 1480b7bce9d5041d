// MFN, hidden widths 33..128.  4 waves x (4+1) row blocks x 36 floats = 92 KB of LDS.
#define INR_NB 4
#define INR_NW 4
#define INR_LAUNCH_NAME launch_mfn_nb4
#include "inr_mfn_inst.h"

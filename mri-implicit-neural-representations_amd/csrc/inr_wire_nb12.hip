// WIRE width 256 -> int(256/sqrt(2)) = 181 complex hidden features = 362 interleaved real rows,
// padded to 384 = 12 blocks.  12 blocks x 33 floats x 32 rows per wave image = 50.7 KB, so a
// workgroup is 3 waves (96-coordinate tiles, 152 KB of LDS).
#define INR_LDS_LD 33  // 3 waves x 384 rows x 33 floats = 152 KB (36 would not fit)
#define INR_NB 12
#define INR_NW 3
#define INR_FAMILY_WIRE 1
#define INR_LAUNCH_NAME launch_wire_nb12
#include "inr_mlp_inst.h"

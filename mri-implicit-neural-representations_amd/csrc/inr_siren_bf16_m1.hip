#define INR_BF16_MODE 1
#include "inr_siren_bf16.hip"

#define INR_BF16_MODE 2
#include "inr_siren_bf16.hip"

#define INR_BF16_MODE 0
#include "inr_siren_bf16.hip"

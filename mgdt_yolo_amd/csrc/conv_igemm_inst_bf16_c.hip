#include "conv_igemm_kernel.h"
MGDT_IGEMM_INSTANTIATE(bf16, 6)
MGDT_IGEMM_INSTANTIATE(bf16, 8)

#include "conv_igemm_kernel.h"
MGDT_IGEMM_INSTANTIATE(bf16, 4)
MGDT_IGEMM_INSTANTIATE(bf16, 5)

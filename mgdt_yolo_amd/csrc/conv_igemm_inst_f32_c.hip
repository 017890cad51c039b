#include "conv_igemm_kernel.h"
MGDT_IGEMM_INSTANTIATE(float, 6)
MGDT_IGEMM_INSTANTIATE(float, 8)

#include "conv_igemm_kernel.h"
MGDT_IGEMM_INSTANTIATE(float, 1)
MGDT_IGEMM_INSTANTIATE(float, 2)
MGDT_IGEMM_INSTANTIATE(float, 3)

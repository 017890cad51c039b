#include "conv_igemm_kernel.h"
MGDT_IGEMM_INSTANTIATE(float, 4)
MGDT_IGEMM_INSTANTIATE(float, 5)

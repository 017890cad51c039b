#include "conv_igemm_kernel.h"
template int launch_igemm<float, 1, 4>(const ConvArgs&, int, int, int, size_t, hipStream_t);

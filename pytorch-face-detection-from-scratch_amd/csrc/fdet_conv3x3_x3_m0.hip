// bf16x3 3x3 conv kernels, epilogue mode EPI_GENERIC (see fdet_conv3x3_x3_kernel.inc)
#define X3_MODE EPI_GENERIC
#define X3_MODE_ID 0
#include "fdet_conv3x3_x3_configs.h"
#include "fdet_conv3x3_x3_kernel.inc"

// bf16x3 3x3 conv kernels, epilogue mode EPI_DGRAD_ADD (see fdet_conv3x3_x3_kernel.inc)
#define X3_MODE EPI_DGRAD_ADD
#define X3_MODE_ID 5
#include "fdet_conv3x3_x3_configs.h"
#include "fdet_conv3x3_x3_kernel.inc"

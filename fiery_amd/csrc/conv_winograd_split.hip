// The split form of the Winograd convolution (conv_winograd.hip, "SPLIT form"): the same kernel source with its K loop on the
// bf16 matrix cores and every fp32 operand as three bf16 terms - fp32 accuracy at six short MFMAs per sixteen channels instead of
// eight long ones.  A translation unit of its own so that it compiles beside the fp32 form.
#define FIERY_WINOGRAD_SPLIT 1
#include "conv_winograd.hip"

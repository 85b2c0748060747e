// conv_b3_gelu.hip — the GELU-carrying instantiations of the bf16x3 convolution kernels (see conv_b3_kernels.h).
#include "conv_b3_kernels.h"

void pg_b3_dispatch_gelu(const B3Args& a, const B3Launch& l, hipStream_t st) { b3_dispatch<true>(a, l, st); }

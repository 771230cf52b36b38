// Fast arithmetic build of the trace kernels: compiled with -ffp-contract=fast
// (v_fma_f64 contraction); parity with the oracle is by tolerance, not bitwise.
#define RTOW_SUFFIX fast
#define RTOW_FAST_MATH 1
#include "rtow_trace_body.h"

// binary32 build of the trace kernels (RTOW_F32): rays, small-primitive hit tests and shading in
// binary32 on binary32 records; the always-test large primitives, every sphere met by the STREAM
// and BVH kernels, and the pixel sums stay binary64 (see `real` in rtow_trace_body.h).  Parity
// with the binary64 builds is by tolerance, stated in tests/test_gpu_f32.py.
#define RTOW_SUFFIX f32
#define RTOW_FAST_MATH 1
#define RTOW_REAL_F32 1
#include "rtow_trace_body.h"

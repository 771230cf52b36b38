// Strict arithmetic build of the trace kernels: compiled with -ffp-contract=off so
// every f64 operation rounds exactly like the CPU oracle's (bit-identical images).
#define RTOW_SUFFIX strict
#include "rtow_trace_body.h"

#include "rt_shade.h"
/* the BSSRDF tails whose probe chain the tracer rounds settled: no traversal in this kernel (rt_shade.h) */
void rt_launch_shade_tail(unsigned blocks, hipStream_t s, const ShadeArgs &a) { rt_launch_shade_t<SSS_TAIL>(blocks, s, a); }

#include "rt_shade.h"
/* scenes with a subsurface material: every vertex except the BSSRDF tails, which are left to rt_shade_sss.hip */
void rt_launch_shade_defer(unsigned blocks, hipStream_t s, const ShadeArgs &a) { rt_launch_shade_t<SSS_DEFER>(blocks, s, a); }
/* the deferred vertices replayed up to their probe segment: chain records for the tracer rounds (rt_shade.h) */
void rt_launch_shade_begin(unsigned blocks, hipStream_t s, const ShadeArgs &a) { rt_launch_shade_t<SSS_BEGIN>(blocks, s, a); }

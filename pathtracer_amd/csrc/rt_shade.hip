#include "rt_shade.h"
void rt_launch_shade(unsigned blocks, hipStream_t s, const ShadeArgs &a) { rt_launch_shade_t<false>(blocks, s, a); }

#include "rt_shade.h"
void rt_launch_shade_sss(unsigned blocks, hipStream_t s, const ShadeArgs &a) { rt_launch_shade_t<true>(blocks, s, a); }

#include "rt_shade.h"
/* the tail kernel: only the vertices k_shade<SSS_DEFER> (rt_shade_defer.hip) left unfinished */
void rt_launch_shade_sss(unsigned blocks, hipStream_t s, const ShadeArgs &a) { rt_launch_shade_t<SSS_INPLACE>(blocks, s, a); }

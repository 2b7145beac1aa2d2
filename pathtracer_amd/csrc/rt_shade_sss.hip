/* The tail kernel lives on occupancy (rt_shade.h): three waves per SIMD measured better than two in spite of the spills
 * (configs[4] shade 159 -> 140 ms), which takes 168 VGPRs and at most 53 KB of LDS per block - 16 stack entries per lane
 * in LDS instead of 24 (deeper entries go to the scratch part of the stack, dev_isect.h), with two lobes and the list: 52 KB. */
#define SSS_WAVES 3
#define LDS_STACK 16
#include "rt_shade.h"
/* the tail kernel: only the vertices k_shade<SSS_DEFER> (rt_shade_defer.hip) left unfinished */
void rt_launch_shade_sss(unsigned blocks, hipStream_t s, const ShadeArgs &a) { rt_launch_shade_t<SSS_INPLACE>(blocks, s, a); }

#include "rt_finish.h"
void rt_launch_finish(unsigned blocks, hipStream_t s, const FinishArgs &a) { rt_launch_finish_t<false>(blocks, s, a); }

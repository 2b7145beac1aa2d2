#include "rt_finish.h"
void rt_launch_finish_sss(unsigned blocks, hipStream_t s, const FinishArgs &a) { rt_launch_finish_t<true>(blocks, s, a); }

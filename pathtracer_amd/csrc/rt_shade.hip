#include "rt_shade.h"
#ifdef SH_STATS
/* diagnostic build only (tools/gpu_shadestats.py): read and clear the section statistics of k_shade<false> */
extern "C" int ptr_debug_shade_stats(unsigned long long *out, int n) {
    unsigned long long tmp[SH_N] = {};
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(tmp, HIP_SYMBOL(g_sh_stats), sizeof(tmp)) != hipSuccess) return 1;
    for (int i = 0; i < n && i < SH_N; i++) out[i] = tmp[i];
    unsigned long long zero[SH_N] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_sh_stats), zero, sizeof(zero)) != hipSuccess) return 1;
    return 0;
}
#endif
void rt_launch_shade(unsigned blocks, hipStream_t s, const ShadeArgs &a) { rt_launch_shade_t<SSS_NONE>(blocks, s, a); }

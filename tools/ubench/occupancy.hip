// occupancy of the HT kernels as the runtime computes it (workgroups of 64 lanes per CU) for a given dynamic LDS size
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include "../../include/htj2k_amd.h"
#include "../../ffmpeg-ht_amd/csrc/j2k_plan.h"
#include "../../ffmpeg-ht_amd/csrc/ht_cxtvlc_rows.h"
#include "../../ffmpeg-ht_amd/csrc/ht_kernels.hpp"
using namespace htj2k;
template <class K> static void show(const char *name, K k, size_t lds)
{
    int n = -1;
    hipFuncAttributes a;
    hipFuncGetAttributes(&a, (const void *)k);
    if (lds > 48 * 1024) hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 64, lds);
    printf("%-24s lds %6zu B  vgpr %3d  static lds %5zu  -> %d workgroups per CU (%s)\n", name, lds, a.numRegs, (size_t)a.sharedSizeBytes, n, hipGetErrorString(e));
}
int main(int argc, char **argv)
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs, %zu B LDS per workgroup max, %zu B per CU\n", p.gcnArchName, p.multiProcessorCount, p.sharedMemPerBlock, (size_t)p.maxSharedMemoryPerMultiProcessor);
    show("k_ht_vlc<true>", k_ht_vlc<true>, HT_VLC_LDS_NARROW);
    show("k_ht_vlc<true> 12 KB", k_ht_vlc<true>, 12160);
    show("k_ht_vlc<false> qw32", k_ht_vlc<false>, ht_vlc_lds_bytes(32));
    show("k_ht_decode_pair 5.8K", k_ht_decode_pair, 2 * (725 + 4) * 4);
    show("k_ht_unstuff 3 KB", k_ht_unstuff, 3072);
    return 0;
}

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
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, a.maxThreadsPerBlock < 256 ? 64 : 256, lds);
    printf("%-24s lds %6zu B  vgpr %3d  static lds %5zu  -> %d workgroups per CU (%s)\n", name, lds, a.numRegs, (size_t)a.sharedSizeBytes, n, hipGetErrorString(e));
}
int main(int argc, char **argv)
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs, %zu B LDS per workgroup max, %zu B per CU\n", p.gcnArchName, p.multiProcessorCount, p.sharedMemPerBlock, (size_t)p.maxSharedMemoryPerMultiProcessor);
    show("k_ht_vlc2 (4 waves)", k_ht_vlc2, HT_VLC2_LDS);
    show("k_ht_vlc qw 64", k_ht_vlc, ht_vlc_lds_bytes(64));
    show("k_ht_decode_pair 5.8K", k_ht_decode_pair, 2 * (725 + 4) * 4);
    show("k_ht_unstuff 3 KB", k_ht_unstuff, 3072);
    show("k_ht_unstuff_g<32> 6 KB", k_ht_unstuff_g<32>, 2 * 3072);
    show("k_ht_unstuff_g<16> 4 KB", k_ht_unstuff_g<16>, 4 * 1024);
    return 0;
}

// membw.hip -- calibration of HBM copy bandwidth for the access shapes the IDWT kernels use.
//   linear    : float4 grid-stride copy (what the microarchitecture guide quotes)
//   rowwalk   : one wave per (column strip, row strip) walking down rows of a W x H plane, VEC dwords per lane,
//               NS source planes read per row, one plane written (the streaming IDWT's shape without arithmetic)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_linear(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

template <int VEC, int NS, int WPB>
__global__ void __launch_bounds__(64 * WPB)
k_rowwalk(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int W, int H, int th, size_t plane)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = (blockIdx.x * WPB + wv) * 64 * VEC + lane * VEC;
    const int y0 = blockIdx.y * th, y1 = min(y0 + th, H);
    if (x >= W) return;
    const uint32_t *s = src + (size_t)blockIdx.z * NS * plane + x;
    uint32_t *d = dst + (size_t)blockIdx.z * plane + x;
    typedef uint32_t vec __attribute__((ext_vector_type(VEC)));
    typedef vec vec_a4 __attribute__((aligned(4)));
    vec cur[NS], nxt[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) cur[k] = *(const vec_a4 *)(s + k * plane + (size_t)y0 * W);
    for (int y = y0; y < y1; y++) {
        const int yn = min(y + 1, y1 - 1);
#pragma unroll
        for (int k = 0; k < NS; k++) nxt[k] = *(const vec_a4 *)(s + k * plane + (size_t)yn * W);
        vec acc = cur[0];
#pragma unroll
        for (int k = 1; k < NS; k++) acc += cur[k];
        *(vec_a4 *)(d + (size_t)y * W) = acc;
#pragma unroll
        for (int k = 0; k < NS; k++) cur[k] = nxt[k];
    }
}

template <int VEC, int NS, int WPB>
static void run_rowwalk(const uint32_t *src, uint32_t *dst, int W, int H, int th, int nz, const char *name)
{
    const size_t plane = (size_t)W * H;
    dim3 g((W + 64 * VEC * WPB - 1) / (64 * VEC * WPB), (H + th - 1) / th, nz);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_rowwalk<VEC, NS, WPB>), g, dim3(64 * WPB), 0, 0, src, dst, W, H, th, plane);
    CK(hipEventRecord(e0));
    const int it = 10;
    for (int i = 0; i < it; i++) hipLaunchKernelGGL((k_rowwalk<VEC, NS, WPB>), g, dim3(64 * WPB), 0, 0, src, dst, W, H, th, plane);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
    const double by = (double)plane * 4 * nz * (NS + 1);
    printf("rowwalk %-10s VEC %d NS %2d WPB %d th %4d nz %3d: %8.1f us  %6.0f GB/s\n", name, VEC, NS, WPB, th, nz, ms * 1e3, by / ms / 1e6);
    fflush(stdout);
}

int main()
{
    const int W = 3840, H = 2160;
    const size_t plane = (size_t)W * H;
    const size_t total = plane * 4 * 96;                       // 3.2 GB source
    uint32_t *src, *dst;
    CK(hipMalloc(&src, total)); CK(hipMalloc(&dst, plane * 4 * 96));
    CK(hipMemset(src, 1, total)); CK(hipMemset(dst, 0, plane * 4 * 96));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t mb : { 256ul, 796ul, 1592ul }) {
        const size_t n = mb * 1000000 / 16;
        for (int blocks : { 2048, 8192, 65536 }) {
            for (int i = 0; i < 2; i++) hipLaunchKernelGGL(k_linear, dim3(blocks), dim3(256), 0, 0, (const uint4 *)src, (uint4 *)dst, n);
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k_linear, dim3(blocks), dim3(256), 0, 0, (const uint4 *)src, (uint4 *)dst, n);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
            printf("linear copy %5zu MB read + same written, %6d blocks: %8.1f us  %6.0f GB/s\n", mb, blocks, ms * 1e3, 2.0 * n * 16 / ms / 1e6);
            fflush(stdout);
        }
    }
    // plain IDWT shape: 1 read stream per written stream, 24 planes
    for (int th : { 16, 64, 2160 }) {
        run_rowwalk<2, 1, 1>(src, dst, W, H, th, 24, "1r1w v2");
        run_rowwalk<4, 1, 1>(src, dst, W, H, th, 24, "1r1w v4");
        run_rowwalk<4, 1, 4>(src, dst, W, H, th, 24, "1r1w v4w4");
    }
    // fused shape: 3 (x4 bands -> modelled as 12 streams of quarter planes is not needed: bytes are what matter) read planes per written plane
    for (int th : { 16, 32, 64 }) {
        run_rowwalk<2, 3, 1>(src, dst, W, H, th, 8, "3r1w v2");
        run_rowwalk<4, 3, 1>(src, dst, W, H, th, 8, "3r1w v4");
        run_rowwalk<2, 4, 1>(src, dst, W, H, th, 24, "4r1w v2");
        run_rowwalk<4, 4, 1>(src, dst, W, H, th, 24, "4r1w v4");
    }
    return 0;
}

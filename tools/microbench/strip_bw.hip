// Access-pattern microbenchmark for the streaming IDWT level kernels (dwt_stream.hpp): one wave walks down a strip of
// 64 * CPL 16-bit columns, per row pair it loads 4 x (CPL / 2) dwords per lane from four sub-band rows and stores 2 x CPL
// 16-bit samples per lane.  CPL = 4 is k_idwt_stream_ll16 as shipped; CPL = 8 / 16 are the wider variants.
// build: hipcc --offload-arch=gfx950 -O3 -o strip_bw strip_bw.hip ; run: ./strip_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <int CPL> struct Vec;
template <> struct Vec<4> { typedef uint32_t L; typedef uint2 S; };
template <> struct Vec<8> { typedef uint2 L; typedef uint4 S; };
template <int CPL>
__global__ void __launch_bounds__(64) k_strip(const uint16_t *__restrict__ band, const uint16_t *__restrict__ ll, uint16_t *__restrict__ out,
                                              int W, int H, int th, int nplanes, int BS, size_t bplane)
{
    typedef typename Vec<CPL>::L L; typedef typename Vec<CPL>::S S;
    const int strips_x = (W + 64 * CPL - 1) / (64 * CPL), strips_y = (H + th - 1) / th;
    int id = blockIdx.x;
    const int bx = id % strips_x; id /= strips_x;
    const int by = id % strips_y; const int pl = id / strips_y;
    if (pl >= nplanes) return;
    const int x = bx * 64 * CPL + threadIdx.x * CPL;
    if (x + CPL > W) return;
    const size_t plane = (size_t)W * H;
    const uint16_t *b = band + pl * bplane, *l = ll + pl * (plane / 4);   /* sub-bands: rows BS samples apart in a plane of bplane samples */
    uint16_t *o = out + pl * plane;
    const int hw = W / 2, hh = H / 2;
    const int y0 = by * th, y1 = min(y0 + th, H);
    uint32_t acc = 0;
    for (int y = y0; y < y1; y += 2) {
        const int r = y >> 1;
        // LL (quarter plane), HL (right half of top rows), LH / HH (bottom half)
        const L a = *(const L *)(l + (size_t)r * hw + x / 2);
        const L h = *(const L *)(b + (size_t)r * BS + hw + x / 2);
        const L c = *(const L *)(b + (size_t)(hh + r) * BS + x / 2);
        const L d = *(const L *)(b + (size_t)(hh + r) * BS + hw + x / 2);
        S s0, s1;
        if constexpr (CPL == 4) { s0 = make_uint2(a + acc, h); s1 = make_uint2(c, d + acc); acc += a ^ d; }
        else { s0 = make_uint4(a.x + acc, a.y, h.x, h.y); s1 = make_uint4(c.x, c.y, d.x, d.y + acc); acc += a.x ^ d.y; }
        *(S *)(o + (size_t)y * W + x) = s0;
        *(S *)(o + (size_t)(y + 1) * W + x) = s1;
    }
}
template <int CPL> void run(const char *name, int W, int H, int nplanes, int th, uint16_t *band, uint16_t *ll, uint16_t *out, int BS = 0, size_t bplane = 0)
{
    if (!BS) { BS = W; bplane = (size_t)W * H; }
    const int strips_x = (W + 64 * CPL - 1) / (64 * CPL), strips_y = (H + th - 1) / th;
    const int grid = strips_x * strips_y * nplanes;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_strip<CPL>, dim3(grid), dim3(64), 0, 0, band, ll, out, W, H, th, nplanes, BS, bplane);
    hipEventRecord(e0);
    const int R = 10;
    for (int i = 0; i < R; i++) hipLaunchKernelGGL(k_strip<CPL>, dim3(grid), dim3(64), 0, 0, band, ll, out, W, H, th, nplanes, BS, bplane);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= R;
    const double bytes = (double)W * H * nplanes * 2 * 2;   // every sample read once and written once, 2 B each
    printf("%-10s W %5d H %5d planes %4d th %3d stride %5d grid %7d: %8.1f us  %6.2f TB/s\n", name, W, H, nplanes, th, BS, grid, ms * 1e3, bytes / ms / 1e9);
}
// The shipped geometry: a wave covers columns x0 - 4 .. x0 + 251 (4 per lane), lanes 0 and 63 only feed their neighbours'
// lifting, lanes 1 .. TW / 4 store.  TW = 256: what a wave that fetched its two halo samples separately could do -- every
// lane stores, strips start on line boundaries.  RGB: three 16-bit components in, 12 bytes of rgb24 per lane and row out.
template <bool RGB>
__global__ void __launch_bounds__(64) k_halo(const uint16_t *__restrict__ band, const uint16_t *__restrict__ ll, uint8_t *__restrict__ out,
                                             int W, int H, int th, int nplanes, int TW)
{
    const int strips_x = (W + TW - 1) / TW, strips_y = (H + th - 1) / th;
    int id = blockIdx.x;
    const int bx = id % strips_x; id /= strips_x;
    const int by = id % strips_y; const int pl = id / strips_y;
    if (pl >= nplanes) return;
    const int halo = TW == 256 ? 0 : 4;
    const int x0 = bx * TW, x = x0 - halo + (int)threadIdx.x * 4;
    const int xc = min(max(x, 0), W - 4);
    const bool st = x >= x0 && x + 4 <= min(x0 + TW, W);
    const size_t plane = (size_t)W * H;
    const int hw = W / 2, hh = H / 2;
    const int y0 = by * th, y1 = min(y0 + th, H);
    constexpr int NC = RGB ? 3 : 1;
    uint32_t acc = 0;
    for (int y = y0; y < y1; y += 2) {
        const int r = y >> 1;
        uint32_t a[NC], h[NC], c[NC], d[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) {
            const uint16_t *b = band + (pl * NC + k) * plane, *l = ll + (pl * NC + k) * (plane / 4);
            a[k] = *(const uint32_t *)(l + (size_t)r * hw + xc / 2);
            h[k] = *(const uint32_t *)(b + (size_t)r * W + hw + xc / 2);
            c[k] = *(const uint32_t *)(b + (size_t)(hh + r) * W + xc / 2);
            d[k] = *(const uint32_t *)(b + (size_t)(hh + r) * W + hw + xc / 2);
        }
        if (RGB) {
            typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
            typedef u32x3 u32x3_a4 __attribute__((aligned(4)));
            u32x3 s0, s1;
            s0.x = a[0] + acc; s0.y = h[1]; s0.z = a[2]; s1.x = c[0]; s1.y = d[1] + acc; s1.z = c[2] ^ d[0] ^ h[0] ^ h[2] ^ a[1] ^ c[1] ^ d[2];
            acc += a[0] ^ d[2];
            uint8_t *o = out + (size_t)pl * plane * 3;
            if (st) { *(u32x3_a4 *)(o + ((size_t)y * W + x) * 3) = s0; *(u32x3_a4 *)(o + ((size_t)(y + 1) * W + x) * 3) = s1; }
        } else {
            uint16_t *o = (uint16_t *)out + pl * plane;
            const uint2 s0 = make_uint2(a[0] + acc, h[0]), s1 = make_uint2(c[0], d[0] + acc);
            acc += a[0] ^ d[0];
            if (st) { *(uint2 *)(o + (size_t)y * W + x) = s0; *(uint2 *)(o + (size_t)(y + 1) * W + x) = s1; }
        }
    }
}
template <bool RGB> void run_halo(int W, int H, int nplanes, int th, int TW, uint16_t *band, uint16_t *ll, uint16_t *out)
{
    const int grid = ((W + TW - 1) / TW) * ((H + th - 1) / th) * nplanes;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_halo<RGB>, dim3(grid), dim3(64), 0, 0, band, ll, (uint8_t *)out, W, H, th, nplanes, TW);
    (void)hipEventRecord(e0);
    const int R = 10;
    for (int i = 0; i < R; i++) hipLaunchKernelGGL(k_halo<RGB>, dim3(grid), dim3(64), 0, 0, band, ll, (uint8_t *)out, W, H, th, nplanes, TW);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= R;
    const double bytes = (double)W * H * nplanes * (RGB ? 3 * 2 + 3 : 2 + 2);
    printf("%-6s W %5d H %5d planes %4d th %3d TW %3d grid %7d: %8.1f us  %6.2f TB/s\n", RGB ? "rgb24" : "ll16", W, H, nplanes, th, TW, grid, ms * 1e3, bytes / ms / 1e9);
}
int main()
{
    const size_t maxs = (size_t)3840 * 2160 * 144;
    uint16_t *band, *ll, *out;
    hipMalloc(&band, maxs * 2); hipMalloc(&ll, maxs / 2); hipMalloc(&out, maxs * 2);
    hipMemset(band, 1, maxs * 2); hipMemset(ll, 2, maxs / 2);
    for (int th : { 16, 32, 68 }) {
        run<4>("cpl4", 1920, 1080, 144, th, band, ll, out);
        run<8>("cpl8", 1920, 1080, 144, th, band, ll, out);
    }
    /* level 4 of a 5-level 3840 x 2160 plane as the decoder has it: the sub-bands sit in the top left quarter of the full-size plane */
    run<4>("cpl4-L4", 1920, 1080, 144, 16, band, ll, out, 3840, (size_t)3840 * 2160);
    run<8>("cpl8-L4", 1920, 1080, 144, 16, band, ll, out, 3840, (size_t)3840 * 2160);
    run<4>("cpl4-L3", 960, 540, 144, 16, band, ll, out, 3840, (size_t)3840 * 2160);
    run<4>("cpl4", 3840, 2160, 144, 16, band, ll, out);
    run<8>("cpl8", 3840, 2160, 144, 16, band, ll, out);
    run<4>("cpl4", 960, 540, 144, 16, band, ll, out);
    run<8>("cpl8", 960, 540, 144, 16, band, ll, out);
    for (int tw : { 244, 224, 256 }) run_halo<false>(1920, 1080, 144, 16, tw, band, ll, out);
    for (int tw : { 244, 224, 256 }) run_halo<false>(960, 540, 144, 16, tw, band, ll, out);
    for (int tw : { 244, 224, 192, 256 }) run_halo<true>(3840, 2160, 48, 16, tw, band, ll, out);
    hipMemcpy(out, band, maxs * 2, hipMemcpyDeviceToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 5; i++) hipMemcpy(out, band, maxs * 2, hipMemcpyDeviceToDevice);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("hipMemcpy D2D %.2f TB/s (read + write)\n", maxs * 2 * 2 * 5 / ms / 1e9);
    return 0;
}

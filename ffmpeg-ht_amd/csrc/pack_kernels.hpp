/*
 * pack_kernels.hpp -- tail of jpeg2000_decode_tile() (libavcodec/jpeg2000dec.c:2368-2395) as
 * one fused pass over the reconstructed planes of a tile:
 *   mct_decode()          jpeg2000dec.c:2183-2209 -> rct_int / ict_float / ict_int,
 *                         libavcodec/jpeg2000dsp.c:43-91 (x86: jpeg2000dsp.asm:37-163)
 *   write_frame_8 / _16   jpeg2000dec.c:2301-2364: (lrintf) + DC level shift, clip to
 *                         [0, 2^cbps - 1], << (precision - cbps), planar or packed store
 * The reference makes two more passes over memory for this (MCT in place, then the
 * store); here every plane sample is read once and every output sample written once.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "j2k_plan.h"

namespace htj2k {

struct PackComp {
    const uint32_t *src;     /* reconstructed plane (int32 or float bits), row stride w */
    int32_t w, h;
    int32_t transform, cbps;
    int32_t out_plane, out_x, out_y;
    int32_t pix_step, pix_off;
};

struct OutPlanes {
    uint8_t *ptr[4];
    int32_t  linesize[4];    /* bytes */
    int32_t  width[4];       /* pixels */
    int32_t  height[4];
};

struct PackTile {
    PackComp c[4];
    OutPlanes out;           /* the frame this tile belongs to */
    int32_t ncomp;
    int32_t mct;             /* components 0..2 go through the inverse MCT of c[0].transform */
    int32_t out_bytes;       /* 1: write_frame_8, 2: write_frame_16 */
    int32_t precision;       /* write_frame's `precision` argument */
    int32_t maxw, maxh;      /* largest component extent: the launch grid */
    int32_t fused;           /* this run's final IDWT level writes the frame itself: k_mct_pack skips the tile */
};

/* The per-position arithmetic and the stores are shared by k_mct_pack (planes in HBM) and by
 * the fused final IDWT level (dwt_stream.hpp: samples still in registers).  A caller holds 4
 * horizontally adjacent sample positions of up to 4 components in v[component][position]. */
__device__ __forceinline__ int pack_value(const PackTile &T, const PackComp &C, int val)
{
    val += 1 << (C.cbps - 1);
    val = min(max(val, 0), (1 << C.cbps) - 1);            /* av_clip */
    return val << (T.precision - C.cbps);
}

/* inverse MCT of components 0..2 (raw plane bits in, integers out) */
__device__ __forceinline__ void pack_mct(int tr, int (&v)[4][4])
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t s0 = (uint32_t)v[0][i], s1 = (uint32_t)v[1][i], s2 = (uint32_t)v[2][i];
        if (tr == J2K_DWT53) {                          /* rct_int, jpeg2000dsp.c:78-91 */
            const uint32_t i1 = s0 - (uint32_t)((int32_t)(s2 + s1) >> 2);
            v[0][i] = (int32_t)(i1 + s2); v[1][i] = (int32_t)i1; v[2][i] = (int32_t)(i1 + s1);
        } else if (tr == J2K_DWT97) {                   /* ict_float, :43-59, then lrintf (jpeg2000dec.c:2340) */
            const float f0 = __uint_as_float(s0), f1 = __uint_as_float(s1), f2 = __uint_as_float(s2);
            const float i0f = f0 + (1.402f * f2);
            const float i1f = f0 - (0.34413f * f1) - (0.71414f * f2);
            const float i2f = f0 + (1.772f * f1);
            v[0][i] = __float2int_rn(i0f); v[1][i] = __float2int_rn(i1f); v[2][i] = __float2int_rn(i2f);
        } else {                                        /* ict_int, :61-76 */
            const int32_t a0 = (int32_t)s0, a1 = (int32_t)s1, a2 = (int32_t)s2;
            v[0][i] = a0 + a2 + ((int)((26345U * (uint32_t)a2) + (1 << 15)) >> 16);
            v[1][i] = a0 - ((int)((22553U * (uint32_t)a1) + (1 << 15)) >> 16)
                         - ((int)((46802U * (uint32_t)a2) + (1 << 15)) >> 16);
            v[2][i] = a0 + (2 * a1) + ((int)((-14942U * (uint32_t)a1) + (1 << 15)) >> 16);
        }
    }
}

/* component c: lrintf of a float plane that did not go through the MCT, DC shift, clip, << */
__device__ __forceinline__ void pack_convert(const PackTile &T, int c, int (&v)[4])
{
    const PackComp &C = T.c[c];
    const bool fl = C.transform == J2K_DWT97 && !(T.mct && c < 3);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int raw = fl ? __float2int_rn(__int_as_float(v[i])) : v[i];
        v[i] = pack_value(T, C, raw);
    }
}

/* packed pixel formats: all components share plane, geometry and a pixel pitch of `step` samples;
 * the n leading positions of v[][], starting at component-plane column x0 of row y */
__device__ __forceinline__ void pack_store_packed(const PackTile &T, int x0, int y, int n, const int (&v)[4][4])
{
    const OutPlanes &O = T.out;
    const PackComp &C0 = T.c[0];
    const int ncomp = T.ncomp, step = C0.pix_step, pl = C0.out_plane;
    const int px = C0.out_x + x0, py = C0.out_y + y;
    if (py < 0 || py >= O.height[pl] || px < 0) return;
    const int nvalid = min(n, O.width[pl] - px);
    if (nvalid <= 0) return;
    uint8_t *dst = O.ptr[pl] + (size_t)py * O.linesize[pl] + (size_t)px * step * T.out_bytes;
    const int nbytes = nvalid * step * T.out_bytes;
    uint32_t wbuf[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };       /* up to 4 px * 4 comps * 2 bytes */
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (c >= step) continue;
            const int val = (c < ncomp) ? v[c][i] : 0;
            const int k = i * step + T.c[c < ncomp ? c : 0].pix_off;   /* pix_off == component index for packed formats */
            if (T.out_bytes == 1) wbuf[k >> 2] |= (uint32_t)(val & 0xFF) << ((k & 3) * 8);
            else wbuf[k >> 1] |= (uint32_t)(val & 0xFFFF) << ((k & 1) * 16);
        }
    if ((((uintptr_t)dst) & 3) == 0 && (nbytes & 3) == 0) {
        uint32_t *d32 = (uint32_t *)dst;
        typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef u32x3 u32x3_a4 __attribute__((aligned(4)));
        typedef u32x4 u32x4_a4 __attribute__((aligned(4)));
        if (nbytes == 12) {                                  /* rgb24, 4 pixels: one 12-byte store */
            u32x3 q; q.x = wbuf[0]; q.y = wbuf[1]; q.z = wbuf[2];
            *(u32x3_a4 *)d32 = q;
        } else if (nbytes == 16) {
            u32x4 q; q.x = wbuf[0]; q.y = wbuf[1]; q.z = wbuf[2]; q.w = wbuf[3];
            *(u32x4_a4 *)d32 = q;
        } else {
            for (int k = 0; k < (nbytes >> 2); k++) d32[k] = wbuf[k];
        }
    } else {
        for (int k = 0; k < nbytes; k++) dst[k] = (uint8_t)(wbuf[k >> 2] >> ((k & 3) * 8));
    }
}

/* planar formats: component c on its own plane */
__device__ __forceinline__ void pack_store_planar(const PackTile &T, int c, int x0, int y, int n, const int (&v)[4])
{
    const OutPlanes &O = T.out;
    const PackComp &C = T.c[c];
    const int pl = C.out_plane;
    const int px = C.out_x + x0, py = C.out_y + y;
    if (py < 0 || py >= O.height[pl] || px < 0) return;
    const int nvalid = min(n, O.width[pl] - px);
    if (nvalid <= 0) return;
    uint8_t *dst = O.ptr[pl] + (size_t)py * O.linesize[pl] + (size_t)(px * C.pix_step + C.pix_off) * T.out_bytes;
    if (T.out_bytes == 1) {
        if (nvalid == 4 && (((uintptr_t)dst) & 3) == 0)
            *(uint32_t *)dst = (uint32_t)(v[0] & 0xFF) | ((uint32_t)(v[1] & 0xFF) << 8) |
                               ((uint32_t)(v[2] & 0xFF) << 16) | ((uint32_t)(v[3] & 0xFF) << 24);
        else
            for (int i = 0; i < nvalid; i++) dst[i] = (uint8_t)v[i];
    } else {
        if (nvalid == 4 && (((uintptr_t)dst) & 7) == 0) {
            uint2 q;
            q.x = (uint32_t)(v[0] & 0xFFFF) | ((uint32_t)(v[1] & 0xFFFF) << 16);
            q.y = (uint32_t)(v[2] & 0xFFFF) | ((uint32_t)(v[3] & 0xFFFF) << 16);
            *(uint2 *)dst = q;
        } else
            for (int i = 0; i < nvalid; i++) ((uint16_t *)dst)[i] = (uint16_t)v[i];
    }
}

/* One thread = 4 horizontally adjacent sample positions of every component of a tile:
 * 16-byte plane loads, the inverse MCT in registers, and the packed / planar output row
 * assembled into whole dwords (rgb24: 12 bytes per thread) instead of byte stores. */
__global__ void __launch_bounds__(256)
k_mct_pack(const PackTile *__restrict__ tiles)
{
    const PackTile &T = tiles[blockIdx.z];
    if (T.fused) return;                                   /* written by the fused final IDWT level */
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    if (x0 >= T.maxw || y >= T.maxh) return;
    const int ncomp = T.ncomp;
    int v[4][4];
    int cnt[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        cnt[c] = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) v[c][i] = 0;
        if (c >= ncomp) continue;
        const PackComp &C = T.c[c];
        if (x0 >= C.w || y >= C.h) continue;
        const int n = min(4, C.w - x0);
        cnt[c] = n;
        const uint32_t *p = C.src + (size_t)y * C.w + x0;
        uint32_t r[4] = { 0, 0, 0, 0 };
        if (n == 4 && ((((uintptr_t)p) & 15) == 0)) {
            const uint4 q = *(const uint4 *)p;
            r[0] = q.x; r[1] = q.y; r[2] = q.z; r[3] = q.w;
        } else {
            for (int i = 0; i < n; i++) r[i] = p[i];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) v[c][i] = (int)r[i];
    }
    if (T.mct && cnt[0]) pack_mct(T.c[0].transform, v);
#pragma unroll
    for (int c = 0; c < 4; c++)
        if (cnt[c]) pack_convert(T, c, v[c]);
    if (T.c[0].pix_step > 1) {
        if (cnt[0]) pack_store_packed(T, x0, y, cnt[0], v);
    } else {
#pragma unroll
        for (int c = 0; c < 4; c++)
            if (cnt[c]) pack_store_planar(T, c, x0, y, cnt[c], v[c]);
    }
}

/* Jpeg2000DSPContext.mct_decode[] alone (unit parity against jpeg2000dsp.c / checkasm) */
__global__ void __launch_bounds__(256)
k_mct_only(uint32_t *p0, uint32_t *p1, uint32_t *p2, int n, int type)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t s0 = p0[i], s1 = p1[i], s2 = p2[i];
    if (type == J2K_DWT53) {
        const uint32_t i1 = s0 - (uint32_t)((int32_t)(s2 + s1) >> 2);
        p0[i] = i1 + s2; p1[i] = i1; p2[i] = i1 + s1;
    } else if (type == J2K_DWT97) {
        const float f0 = __uint_as_float(s0), f1 = __uint_as_float(s1), f2 = __uint_as_float(s2);
        p0[i] = __float_as_uint(f0 + (1.402f * f2));
        p1[i] = __float_as_uint(f0 - (0.34413f * f1) - (0.71414f * f2));
        p2[i] = __float_as_uint(f0 + (1.772f * f1));
    } else {
        const int32_t a0 = (int32_t)s0, a1 = (int32_t)s1, a2 = (int32_t)s2;
        p0[i] = (uint32_t)(a0 + a2 + ((int)((26345U * (uint32_t)a2) + (1 << 15)) >> 16));
        p1[i] = (uint32_t)(a0 - ((int)((22553U * (uint32_t)a1) + (1 << 15)) >> 16)
                              - ((int)((46802U * (uint32_t)a2) + (1 << 15)) >> 16));
        p2[i] = (uint32_t)(a0 + (2 * a1) + ((int)((-14942U * (uint32_t)a1) + (1 << 15)) >> 16));
    }
}

}  // namespace htj2k

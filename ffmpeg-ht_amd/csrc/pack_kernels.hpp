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
};

__device__ __forceinline__ void pack_store(const OutPlanes &O, const PackTile &T, const PackComp &C,
                                           int x, int y, int val)
{
    val += 1 << (C.cbps - 1);
    val = min(max(val, 0), (1 << C.cbps) - 1);            /* av_clip */
    val <<= (T.precision - C.cbps);
    const int px = C.out_x + x, py = C.out_y + y;
    if (px < 0 || py < 0 || px >= O.width[C.out_plane] || py >= O.height[C.out_plane])
        return;                                            /* never write outside the caller's picture */
    uint8_t *line = O.ptr[C.out_plane] + (size_t)py * O.linesize[C.out_plane];
    if (T.out_bytes == 1) line[px * C.pix_step + C.pix_off] = (uint8_t)val;
    else ((uint16_t *)line)[px * C.pix_step + C.pix_off] = (uint16_t)val;
}

__global__ void __launch_bounds__(256)
k_mct_pack(const PackTile *__restrict__ tiles)
{
    const PackTile T = tiles[blockIdx.z];
    const OutPlanes &O = T.out;
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= T.maxw || y >= T.maxh) return;
    int first_plain = 0;
    if (T.mct) {
        first_plain = 3;
        const PackComp &C0 = T.c[0];
        if (x < C0.w && y < C0.h) {
            const size_t o = (size_t)y * C0.w + x;
            const uint32_t s0 = T.c[0].src[o], s1 = T.c[1].src[o], s2 = T.c[2].src[o];
            int v0, v1, v2;
            if (C0.transform == J2K_DWT53) {                /* rct_int, jpeg2000dsp.c:78-91 */
                const uint32_t i1 = s0 - (uint32_t)((int32_t)(s2 + s1) >> 2);
                v0 = (int32_t)(i1 + s2); v1 = (int32_t)i1; v2 = (int32_t)(i1 + s1);
            } else if (C0.transform == J2K_DWT97) {         /* ict_float, :43-59 */
                const float f0 = __uint_as_float(s0), f1 = __uint_as_float(s1), f2 = __uint_as_float(s2);
                const float i0f = f0 + (1.402f * f2);
                const float i1f = f0 - (0.34413f * f1) - (0.71414f * f2);
                const float i2f = f0 + (1.772f * f1);
                v0 = __float2int_rn(i0f); v1 = __float2int_rn(i1f); v2 = __float2int_rn(i2f);   /* lrintf */
            } else {                                        /* ict_int, :61-76 */
                const int32_t a0 = (int32_t)s0, a1 = (int32_t)s1, a2 = (int32_t)s2;
                v0 = a0 + a2 + ((int)((26345U * (uint32_t)a2) + (1 << 15)) >> 16);
                v1 = a0 - ((int)((22553U * (uint32_t)a1) + (1 << 15)) >> 16)
                        - ((int)((46802U * (uint32_t)a2) + (1 << 15)) >> 16);
                v2 = a0 + (2 * a1) + ((int)((-14942U * (uint32_t)a1) + (1 << 15)) >> 16);
            }
            pack_store(O, T, T.c[0], x, y, v0);
            pack_store(O, T, T.c[1], x, y, v1);
            pack_store(O, T, T.c[2], x, y, v2);
        }
    }
    for (int c = first_plain; c < T.ncomp; c++) {
        const PackComp &C = T.c[c];
        if (x >= C.w || y >= C.h) continue;
        const uint32_t s = C.src[(size_t)y * C.w + x];
        const int v = C.transform == J2K_DWT97 ? __float2int_rn(__uint_as_float(s)) : (int32_t)s;
        pack_store(O, T, C, x, y, v);
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

/*
 * dwt_kernels.hpp -- inverse DWT (ff_dwt_decode, libavcodec/jpeg2000dwt.c:601-620) for gfx950.
 *
 * Per decomposition level the reference runs a horizontal 1-D synthesis over every row of
 * the level's region, then a vertical one over every column, in place, one line at a time
 * through a line buffer (dwt_decode53 :327-374, dwt_decode97_float :403-451,
 * dwt_decode97_int :483-537).  Lifting with symmetric extension is evaluated here in
 * closed form per output sample: an output at absolute position a depends on the
 * de-interleaved coefficients at a-2..a+2 (5/3) or a-4..a+4 (9/7), fetched through a
 * whole-sample reflection of the position into [i0, i1) -- bit-identical to the
 * reference's sequential extend53/extend97 (:49-75) for every length >= 2 (SURVEY
 * Appendix B); 1-sample lines are the special cases of :313-317, :380-386, :457-463.
 *
 * Arithmetic notes that are part of parity:
 *   5/3      unsigned wrap-around adds, arithmetic shift of the int-cast sum (:321-324)
 *   9/7 f    p -= C * (a + b): separately rounded add, multiply, subtract; this file is
 *            compiled with -ffp-contract=off (the reference object has no FMA)
 *   9/7 int  16.16 constants, int64 products, +32768 >> 16 (:467-480)
 *
 * Kernel families (htj2k_set_int "idwt_mode"):
 *   3  k_idwt_stream / k_idwt_stream_pack (dwt_stream.hpp, default): register-streaming, no
 *      LDS; the final level fused with the inverse MCT and the frame store
 *   1  k_idwt_tile: the four sub-band tiles staged through LDS, lifting in LDS: what the
 *      streaming kernels hand lines of a single sample to
 *   0  k_idwt_h / k_idwt_v: any geometry, one closed-form output per thread, row pass into a
 *      scratch plane then column pass back (2 reads + 2 writes per level): planes with more
 *      levels than their size allows, and the definition the others are tested against
 * All are bit-identical.  (Round 1's second LDS family, k_idwt_tile2, was an A/B reference only and is gone.)
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "j2k_plan.h"

namespace htj2k {

/* geometry of one level of one plane */
struct DwtLevel {
    uint32_t plane_off;     /* sample offset of the tile-component plane */
    int32_t  stride;        /* full plane width */
    int32_t  lh, lv;        /* linelen[lev][0], [1] */
    int32_t  mh, mv;        /* mod[lev][0], [1] */
    int32_t  last;          /* 1 for the final level of a 9/7-int plane: apply (x + 128) >> 8 */
};

#define F_ALPHA 1.586134342059924f
#define F_BETA  0.052980118572961f
#define F_GAMMA 0.882911075530934f
#define F_DELTA 0.443506852043971f
#define F_K     1.230174104914001f
#define F_X     0.812893066115961f
#define I_ALPHA_PRIME 38413ll
#define I_BETA         3472ll
#define I_GAMMA       57862ll
#define I_DELTA       29066ll
#define I_K           80621ll
#define I_X           53274ll

/* one line: absolute positions [i0, i1), lows at the even positions stored first */
struct LineMap {
    int i0, i1, fe, fo, nl;
    __device__ __forceinline__ LineMap(int m, int len)
    {
        i0 = m; i1 = m + len;
        fe = i0 + (i0 & 1);
        fo = i0 + 1 - (i0 & 1);
        nl = ((i1 + 1) >> 1) - ((i0 + 1) >> 1);
    }
    /* whole-sample symmetric reflection into [i0, i1) (len >= 2) */
    __device__ __forceinline__ int refl(int a) const
    {
        const int n = i1 - i0, p = 2 * (n - 1);
        int r = (a - i0) % p;
        if (r < 0) r += p;
        if (r >= n) r = p - r;
        return i0 + r;
    }
    /* storage index (within the line) of the coefficient that sits at absolute position a */
    __device__ __forceinline__ int idx(int a) const
    {
        a = refl(a);
        return (a & 1) ? nl + ((a - fo) >> 1) : ((a - fe) >> 1);
    }
};

/* ---- closed-form 1-D synthesis of the sample at absolute position a ----
 * F fetch(int storage_index) returns the coefficient as raw 32 bits */
template <class F>
__device__ __forceinline__ uint32_t synth53(const LineMap &L, int a, F fetch)
{
    if (L.i1 <= L.i0 + 1) {                               /* jpeg2000dwt.c:313-317 */
        uint32_t v = fetch(0);
        return L.i0 == 1 ? (uint32_t)((int)v >> 1) : v;
    }
    auto even = [&](int e) -> uint32_t {                  /* p[2i] -= (int)(p[2i-1] + p[2i+1] + 2) >> 2 */
        return fetch(L.idx(e)) - (uint32_t)((int)(fetch(L.idx(e - 1)) + fetch(L.idx(e + 1)) + 2u) >> 2);
    };
    if (!(a & 1)) return even(a);
    return fetch(L.idx(a)) + (uint32_t)((int)(even(a - 1) + even(a + 1)) >> 1);
}

template <class F>
__device__ __forceinline__ float synth97f(const LineMap &L, int a, F fetch)
{
    if (L.i1 <= L.i0 + 1) {                               /* :380-386 */
        float v = __uint_as_float(fetch(0));
        return L.i0 == 1 ? v * (F_K / 2) : v * F_X;
    }
    auto c  = [&](int p) -> float { return __uint_as_float(fetch(L.idx(p))); };
    auto e1 = [&](int p) -> float { return c(p) - F_DELTA * (c(p - 1) + c(p + 1)); };        /* delta, even p */
    auto o2 = [&](int p) -> float { return c(p) - F_GAMMA * (e1(p - 1) + e1(p + 1)); };      /* gamma, odd p  */
    auto e3 = [&](int p) -> float { return e1(p) + F_BETA * (o2(p - 1) + o2(p + 1)); };      /* beta,  even p */
    if (!(a & 1)) return e3(a);
    return o2(a) + F_ALPHA * (e3(a - 1) + e3(a + 1));                                         /* alpha, odd p  */
}

template <class F>
__device__ __forceinline__ int32_t synth97i(const LineMap &L, int a, F fetch)
{
    if (L.i1 <= L.i0 + 1) {                               /* :457-463 */
        int32_t v = (int32_t)fetch(0);
        return L.i0 == 1 ? (int32_t)((v * I_K + (1 << 16)) >> 17) : (int32_t)((v * I_X + (1 << 15)) >> 16);
    }
    auto c  = [&](int p) -> int32_t { return (int32_t)fetch(L.idx(p)); };
    auto e1 = [&](int p) -> int32_t { return c(p) - (int32_t)((I_DELTA * (c(p - 1) + (int64_t)c(p + 1)) + (1 << 15)) >> 16); };
    auto o2 = [&](int p) -> int32_t { return c(p) - (int32_t)((I_GAMMA * (e1(p - 1) + (int64_t)e1(p + 1)) + (1 << 15)) >> 16); };
    auto e3 = [&](int p) -> int32_t { return e1(p) + (int32_t)((I_BETA * (o2(p - 1) + (int64_t)o2(p + 1)) + (1 << 15)) >> 16); };
    if (!(a & 1)) return e3(a);
    {
        const int64_t sum = e3(a - 1) + (int64_t)e3(a + 1);
        int32_t v = o2(a);
        v += (int32_t)sum;
        v += (int32_t)((I_ALPHA_PRIME * sum + (1 << 15)) >> 16);
        return v;
    }
}

template <int TYPE, class F>
__device__ __forceinline__ uint32_t synth(const LineMap &L, int a, F fetch)
{
    if (TYPE == J2K_DWT53) return synth53(L, a, fetch);
    if (TYPE == J2K_DWT97) return __float_as_uint(synth97f(L, a, fetch));
    return (uint32_t)synth97i(L, a, fetch);
}

/* ---- generic kernels: grid.z indexes the DwtLevel table (one entry per plane) ---- */
template <int TYPE>
__global__ void __launch_bounds__(256)
k_idwt_h(const DwtLevel *__restrict__ lv, const uint32_t *__restrict__ src, uint32_t *__restrict__ dst)
{
    const DwtLevel g = lv[blockIdx.z];
    const int k = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (k >= g.lh || y >= g.lv) return;
    const uint32_t *row = src + g.plane_off + (size_t)y * g.stride;
    LineMap L(g.mh, g.lh);
    dst[g.plane_off + (size_t)y * g.stride + k] = synth<TYPE>(L, g.mh + k, [&](int i) { return row[i]; });
}

template <int TYPE>
__global__ void __launch_bounds__(256)
k_idwt_v(const DwtLevel *__restrict__ lv, const uint32_t *__restrict__ src, uint32_t *__restrict__ dst)
{
    const DwtLevel g = lv[blockIdx.z];
    const int x = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
    if (x >= g.lh || k >= g.lv) return;
    const uint32_t *col = src + g.plane_off + x;
    const int stride = g.stride;
    LineMap L(g.mv, g.lv);
    uint32_t v = synth<TYPE>(L, g.mv + k, [&](int i) { return col[(size_t)i * stride]; });
    if (TYPE == J2K_DWT97_INT && g.last)
        v = (uint32_t)((int32_t)((int32_t)v + 128) >> 8);        /* jpeg2000dwt.c:534-536 */
    dst[g.plane_off + (size_t)k * stride + x] = v;
}

/* 9/7-int planes that never reach a vertical pass (no levels) still are not rescaled by the
 * reference either (ff_dwt_decode returns early, :603): nothing to do. */

/* ================================================================== fused tile kernels
 * Output tile TW x TH at (x0, y0) of the level's lh x lv region.  The needed coefficients are
 * staged in LDS in *interleaved* order (low/high de-interleaving undone while loading, so
 * the lifting steps work on plain neighbouring positions), with HALO extra positions on
 * every side filled through the symmetric reflection.  Then: horizontal lifting on every
 * staged row (in LDS), vertical lifting on every staged column (in LDS), and the TW x TH
 * centre is written out in full coalesced rows.
 *
 * LL comes from `ll` (the previous level's output, row stride ll_stride) and HL/LH/HH from
 * `band` (the dequantised coefficient plane, Mallat layout, row stride g.stride); `out` is
 * never one of the inputs of the same launch, so tiles are independent. */
struct DwtTileArgs {
    DwtLevel g;
    uint32_t ll_off;  int32_t ll_stride;     /* where the LL quadrant of this level lives */
    uint32_t out_off; int32_t out_stride;
};

template <int TYPE, int TW, int TH>
__global__ void __launch_bounds__(256)
k_idwt_tile(const DwtTileArgs *__restrict__ args, const uint32_t *__restrict__ ll_base,
            const uint32_t *__restrict__ band_base, uint32_t *__restrict__ out_base)
{
    constexpr int HALO = (TYPE == J2K_DWT53) ? 2 : 4;
    constexpr int SW = TW + 2 * HALO, SH = TH + 2 * HALO;
    constexpr int PITCH = SW + 1;                         /* odd pitch: column walks hit distinct banks */
    __shared__ uint32_t tile[SH * PITCH];

    const DwtTileArgs A = args[blockIdx.z];
    const DwtLevel g = A.g;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    if (x0 >= g.lh || y0 >= g.lv) return;
    const LineMap LX(g.mh, g.lh), LY(g.mv, g.lv);
    const int tid = threadIdx.x;
    const uint32_t *ll = ll_base + A.ll_off, *band = band_base + g.plane_off;
    const bool one_x = g.lh == 1, one_y = g.lv == 1;

    /* ---- stage the coefficients, interleaved, with reflected halo ---- */
    for (int i = tid; i < SH * SW; i += 256) {
        const int sy = i / SW, sx = i - sy * SW;
        /* absolute positions of this staged cell */
        int ax = g.mh + x0 - HALO + sx, ay = g.mv + y0 - HALO + sy;
        uint32_t v = 0;
        /* cells further than the line ends + HALO are never used */
        if (ax >= g.mh - HALO && ax < g.mh + g.lh + HALO && ay >= g.mv - HALO && ay < g.mv + g.lv + HALO) {
            const int ix = one_x ? 0 : LX.idx(ax), iy = one_y ? 0 : LY.idx(ay);
            const bool lowx = ix < LX.nl, lowy = iy < LY.nl;   /* (a 1-sample line is low iff its origin is even: nl says so) */
            if (lowx && lowy) v = ll[(size_t)iy * A.ll_stride + ix];
            else              v = band[(size_t)iy * g.stride + ix];
        }
        tile[sy * PITCH + sx] = v;
    }
    __syncthreads();

    /* ---- lifting in LDS: each step updates all cells of one parity at once (a step only
     * reads the other parity, so there is no hazard inside a step); `margin` cells at both
     * ends of a line are not updated, so after the last step exactly the TW x TH centre
     * is valid ---- */
    auto step_all = [&](bool horizontal, int parity_odd, int margin, int kind) {
        /* kind: 0 = 5/3 even, 1 = 5/3 odd, 2..5 = 9/7 delta, gamma, beta, alpha */
        const int n_line = horizontal ? SH : SW;          /* number of lines */
        const int n_cell = horizontal ? SW : SH;          /* cells per line */
        const int a_first = horizontal ? (g.mh + x0 - HALO) : (g.mv + y0 - HALO);
        const int first = margin + (((a_first + margin) & 1) != parity_odd);
        const int per_line = (n_cell - margin - first + 1) >> 1;   /* cells first, first+2, ... < n_cell - margin */
        if (per_line <= 0) return;
        for (int i = tid; i < n_line * per_line; i += 256) {
            const int line = i / per_line, k = first + 2 * (i - line * per_line);
            if (k >= n_cell - margin) continue;
            const int c = horizontal ? line * PITCH + k : k * PITCH + line;
            const int st = horizontal ? 1 : PITCH;
            if (TYPE == J2K_DWT53) {
                if (kind == 0) tile[c] -= (uint32_t)((int)(tile[c - st] + tile[c + st] + 2u) >> 2);
                else           tile[c] += (uint32_t)((int)(tile[c - st] + tile[c + st]) >> 1);
            } else if (TYPE == J2K_DWT97) {
                float *t = (float *)tile;
                const float s = t[c - st] + t[c + st];
                if (kind == 2)      t[c] -= F_DELTA * s;
                else if (kind == 3) t[c] -= F_GAMMA * s;
                else if (kind == 4) t[c] += F_BETA * s;
                else                t[c] += F_ALPHA * s;
            } else {
                int32_t *t = (int32_t *)tile;
                const int64_t s = t[c - st] + (int64_t)t[c + st];
                if (kind == 2)      t[c] -= (int32_t)((I_DELTA * s + (1 << 15)) >> 16);
                else if (kind == 3) t[c] -= (int32_t)((I_GAMMA * s + (1 << 15)) >> 16);
                else if (kind == 4) t[c] += (int32_t)((I_BETA * s + (1 << 15)) >> 16);
                else { t[c] += (int32_t)s; t[c] += (int32_t)((I_ALPHA_PRIME * s + (1 << 15)) >> 16); }
            }
        }
    };
    auto scale_single = [&](bool horizontal) {
        /* the line has one sample: every staged cell along that axis is that sample */
        const int m = horizontal ? g.mh : g.mv;
        for (int i = tid; i < SH * SW; i += 256) {
            const int sy = i / SW, sx = i - sy * SW;
            uint32_t v = tile[sy * PITCH + sx];
            if (TYPE == J2K_DWT53) { if (m == 1) v = (uint32_t)((int)v >> 1); }
            else if (TYPE == J2K_DWT97) { float f = __uint_as_float(v); f = m == 1 ? f * (F_K / 2) : f * F_X; v = __float_as_uint(f); }
            else { int32_t s = (int32_t)v; s = m == 1 ? (int32_t)((s * I_K + (1 << 16)) >> 17) : (int32_t)((s * I_X + (1 << 15)) >> 16); v = (uint32_t)s; }
            tile[sy * PITCH + sx] = v;
        }
    };

    for (int pass = 0; pass < 2; pass++) {
        const bool horizontal = pass == 0;
        if (horizontal ? one_x : one_y) {
            scale_single(horizontal);
            __syncthreads();
            continue;
        }
        if (TYPE == J2K_DWT53) {
            step_all(horizontal, 0, 1, 0); __syncthreads();
            step_all(horizontal, 1, 2, 1); __syncthreads();
        } else {
            step_all(horizontal, 0, 1, 2); __syncthreads();
            step_all(horizontal, 1, 2, 3); __syncthreads();
            step_all(horizontal, 0, 3, 4); __syncthreads();
            step_all(horizontal, 1, 4, 5); __syncthreads();
        }
    }

    /* ---- write the centre ---- */
    uint32_t *out = out_base + A.out_off;
    for (int i = tid; i < TH * TW; i += 256) {
        const int ty = i / TW, tx = i - ty * TW;
        const int x = x0 + tx, y = y0 + ty;
        if (x >= g.lh || y >= g.lv) continue;
        uint32_t v = tile[(ty + HALO) * PITCH + tx + HALO];
        if (TYPE == J2K_DWT97_INT && g.last)
            v = (uint32_t)((int32_t)((int32_t)v + 128) >> 8);
        out[(size_t)y * A.out_stride + x] = v;
    }
}


/* lane-to-lane moves of the streaming kernels (dwt_stream.hpp) */
__device__ __forceinline__ uint32_t dpp_from_left(uint32_t v)    /* lane i <- lane i-1, lane 0 <- 0 */
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);     /* wave_shr:1, bound_ctrl: no `old` value to set up */
}
__device__ __forceinline__ uint32_t dpp_from_right(uint32_t v)   /* lane i <- lane i+1, lane 63 <- 0 */
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);     /* wave_shl:1 */
}

}  // namespace htj2k

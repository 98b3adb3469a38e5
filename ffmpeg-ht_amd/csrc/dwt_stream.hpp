/*
 * dwt_stream.hpp -- one level of the inverse DWT (dwt_decode53 / dwt_decode97_float /
 * dwt_decode97_int, libavcodec/jpeg2000dwt.c:327-537) as a register-streaming kernel, and the
 * same kernel with the tail of jpeg2000_decode_tile() (mct_decode + write_frame_8/16,
 * libavcodec/jpeg2000dec.c:2183-2209, :2301-2364) fused behind the final level.
 *
 * One wave owns a strip of 256 absolute column positions (4 per lane: even, odd, even, odd)
 * and walks down the rows of its strip two at a time.  Per step it
 *   loads      the vertical-low row (LL | HL) and the vertical-high row (LH | HH) of the step,
 *              two 8-byte loads per lane and row, issued one step ahead of their use
 *   horizontal lifts both rows in registers; the only values that cross lanes are the odd
 *              sample on the left and the even sample on the right: two DPP wave shifts per
 *              lifting step
 *   vertical   advances a pipelined lifting state (2 rows of history for 5/3, 4 for 9/7) and
 *              gets two finished output rows
 *   stores     them: 16 bytes per lane and row (plain level), or -- final level -- inverse MCT,
 *              rounding, DC shift, clip and the packed / planar frame store (rgb24: 12 bytes
 *              per lane and row) with the samples still in registers.
 * No LDS, no barrier: every coefficient is read once (plus the strip halo: lanes 0 and 63, and
 * HALO rows above and below the strip) and every output written once.  The final level of an
 * RGB frame moves 4 + 1 bytes per sample instead of the 4 + 4 (IDWT) + 4 + 1 (MCT/pack
 * kernel) of the unfused pipeline.
 *
 * Boundaries: as in k_idwt_tile2, positions outside the line are fetched through the
 * whole-sample symmetric reflection (LineMap::idx), which is bit-identical to the reference's
 * sequential extend53/extend97 for lines of >= 2 samples; the lifting then runs on the extended
 * signal and results more than HALO positions away from the valid span are discarded.
 * Lines of a single sample are left to k_idwt_tile.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dwt_kernels.hpp"
#include "pack_kernels.hpp"

namespace htj2k {

#define STREAM_TW 244          /* output columns per wave: lanes 1..61 whole + slack for an odd origin; multiple of 4 */

/* elementwise lifting steps on raw 32-bit samples: s1/s3 update an even sample from its odd
 * neighbours a, b; s2/s4 an odd sample from its even neighbours */
template <int TYPE> struct LiftOps;
template <> struct LiftOps<J2K_DWT53> {
    static constexpr int HALO = 2, DELAY = 0;
    static __device__ __forceinline__ uint32_t s1(uint32_t c, uint32_t a, uint32_t b) { return c - (uint32_t)((int)(a + b + 2u) >> 2); }   /* :321-322 */
    static __device__ __forceinline__ uint32_t s2(uint32_t c, uint32_t a, uint32_t b) { return c + (uint32_t)((int)(a + b) >> 1); }        /* :323-324 */
    static __device__ __forceinline__ uint32_t s3(uint32_t c, uint32_t, uint32_t) { return c; }
    static __device__ __forceinline__ uint32_t s4(uint32_t c, uint32_t, uint32_t) { return c; }
};
template <> struct LiftOps<J2K_DWT97> {
    static constexpr int HALO = 4, DELAY = 2;
    static __device__ __forceinline__ float F(uint32_t u) { return __uint_as_float(u); }
    static __device__ __forceinline__ uint32_t s1(uint32_t c, uint32_t a, uint32_t b) { return __float_as_uint(F(c) - F_DELTA * (F(a) + F(b))); }   /* :390-391 */
    static __device__ __forceinline__ uint32_t s2(uint32_t c, uint32_t a, uint32_t b) { return __float_as_uint(F(c) - F_GAMMA * (F(a) + F(b))); }   /* :393-394 */
    static __device__ __forceinline__ uint32_t s3(uint32_t c, uint32_t a, uint32_t b) { return __float_as_uint(F(c) + F_BETA  * (F(a) + F(b))); }   /* :396-397 */
    static __device__ __forceinline__ uint32_t s4(uint32_t c, uint32_t a, uint32_t b) { return __float_as_uint(F(c) + F_ALPHA * (F(a) + F(b))); }   /* :399-400 */
};
template <> struct LiftOps<J2K_DWT97_INT> {
    static constexpr int HALO = 4, DELAY = 2;
    static __device__ __forceinline__ int64_t S(uint32_t a, uint32_t b) { return (int32_t)a + (int64_t)(int32_t)b; }
    static __device__ __forceinline__ uint32_t s1(uint32_t c, uint32_t a, uint32_t b) { return (uint32_t)((int32_t)c - (int32_t)((I_DELTA * S(a, b) + (1 << 15)) >> 16)); }   /* :467-468 */
    static __device__ __forceinline__ uint32_t s2(uint32_t c, uint32_t a, uint32_t b) { return (uint32_t)((int32_t)c - (int32_t)((I_GAMMA * S(a, b) + (1 << 15)) >> 16)); }   /* :470-471 */
    static __device__ __forceinline__ uint32_t s3(uint32_t c, uint32_t a, uint32_t b) { return (uint32_t)((int32_t)c + (int32_t)((I_BETA  * S(a, b) + (1 << 15)) >> 16)); }   /* :473-474 */
    static __device__ __forceinline__ uint32_t s4(uint32_t c, uint32_t a, uint32_t b)                                                                                         /* :476-480 */
    {
        const int64_t sum = S(a, b);
        int32_t o = (int32_t)c;
        o += (int32_t)sum;
        o += (int32_t)((I_ALPHA_PRIME * sum + (1 << 15)) >> 16);
        return (uint32_t)o;
    }
};

/* horizontal synthesis of the lane's (even, odd, even, odd) quadruple; called with all 64 lanes
 * active.  Afterwards lanes 1..62 hold finished samples. */
template <int TYPE>
__device__ __forceinline__ void stream_hlift(uint32_t (&v)[4])
{
    using O = LiftOps<TYPE>;
    uint32_t lo = dpp_from_left(v[3]);
    v[0] = O::s1(v[0], lo, v[1]);
    v[2] = O::s1(v[2], v[1], v[3]);
    uint32_t re = dpp_from_right(v[0]);
    v[1] = O::s2(v[1], v[0], v[2]);
    v[3] = O::s2(v[3], v[2], re);
    if (TYPE != J2K_DWT53) {
        lo = dpp_from_left(v[3]);
        v[0] = O::s3(v[0], lo, v[1]);
        v[2] = O::s3(v[2], v[1], v[3]);
        re = dpp_from_right(v[0]);
        v[1] = O::s4(v[1], v[0], v[2]);
        v[3] = O::s4(v[3], v[2], re);
    }
}

/* the components one wave reconstructs in lockstep share the level geometry g of a[0] */
struct DwtFusedArgs {
    DwtTileArgs a[4];
    int32_t ncomp;            /* 1, 3 or 4 */
    int32_t pack_tile;        /* index of the PackTile the group belongs to */
    int32_t comp0;            /* first component of that tile held by the group */
    int32_t pad;
};

template <int TYPE, int NC, bool FUSED>
__device__ __forceinline__ void
idwt_stream_body(const DwtTileArgs (&A)[NC], const uint32_t *__restrict__ ll_base, const uint32_t *__restrict__ band_base,
                 uint32_t *__restrict__ out_base, const PackTile *__restrict__ T, int comp0, int th)
{
    using O = LiftOps<TYPE>;
    constexpr int HALO = O::HALO, DELAY = O::DELAY;
    const DwtLevel g = A[0].g;
    const int x0 = blockIdx.x * STREAM_TW, y0 = blockIdx.y * th;
    if (x0 >= g.lh || y0 >= g.lv) return;
    const LineMap LX(g.mh, g.lh), LY(g.mv, g.lv);
    const int lane = threadIdx.x;

    /* ---- columns of this lane ---- */
    const int ax0 = (g.mh + x0 - 4) & ~1;                 /* even; lane 1 starts at or one before the strip */
    const int pe0 = ax0 + 4 * lane;
    const bool interior = ax0 >= g.mh && ax0 + 256 <= g.mh + g.lh;      /* wave-uniform: no reflection, all loaded */
    int col[4];
    bool use[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int p = pe0 + k;
        use[k] = p >= g.mh - HALO - 2 && p < g.mh + g.lh + HALO + 2;
        col[k] = use[k] ? LX.idx(p) : 0;
    }
    const uint32_t *llp[NC], *bandp[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        llp[c] = ll_base + A[c].ll_off;
        bandp[c] = band_base + A[c].g.plane_off;
    }

    /* ---- rows of this strip ---- */
    const int a_first = g.mv + y0, a_last = g.mv + min(y0 + th, g.lv) - 1;
    const int s_first = (a_first - HALO) & ~1;
    const int s_last = (a_last + DELAY + 1) & ~1;

    auto load_rows = [&](int ye, uint32_t (&Lr)[NC][4], uint32_t (&Hr)[NC][4]) {
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int ay = ye + r;
            const bool rowok = ay >= g.mv - HALO - 2 && ay < g.mv + g.lv + HALO + 2;
            int iy = 0;
            if (rowok)
                iy = (ay >= LY.i0 && ay < LY.i1) ? ((ay & 1) ? LY.nl + ((ay - LY.fo) >> 1) : ((ay - LY.fe) >> 1)) : LY.idx(ay);
#pragma unroll
            for (int c = 0; c < NC; c++) {
                uint32_t (&dstv)[4] = r ? Hr[c] : Lr[c];
                /* even absolute rows are vertical-low rows (reflection keeps the parity): their
                 * low-horizontal half is the previous level's output */
                const uint32_t *orow = bandp[c] + (size_t)iy * A[c].g.stride;
                const uint32_t *erow = r ? orow : llp[c] + (size_t)iy * A[c].ll_stride;
                if (!rowok) {
                    dstv[0] = dstv[1] = dstv[2] = dstv[3] = 0;
                } else if (interior) {
                    const uint2 e = *(const uint2 *)(erow + col[0]);
                    const uint2 o = *(const uint2 *)(orow + col[1]);
                    dstv[0] = e.x; dstv[1] = o.x; dstv[2] = e.y; dstv[3] = o.y;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) dstv[k] = use[k] ? ((k & 1) ? orow : erow)[col[k]] : 0u;
                }
            }
        }
    };

    /* vertical lifting state per component and column: the previous high row and the
     * unfinished rows above it */
    uint32_t Hp[NC][4], Sa[NC][4], Sb[NC][4], Sc[NC][4];
#pragma unroll
    for (int c = 0; c < NC; c++)
#pragma unroll
        for (int k = 0; k < 4; k++) { Hp[c][k] = 0; Sa[c][k] = 0; Sb[c][k] = 0; Sc[c][k] = 0; }

    /* output columns of this lane */
    const int xa = pe0 - g.mh;
    const int x_hi = min(x0 + STREAM_TW, g.lh);
    int ia = 0, ib = 0;                                   /* valid positions of the quadruple: [ia, ib) */
    if (lane >= 1 && lane <= 62) {
        ia = max(0, x0 - xa);
        ib = min(4, x_hi - xa);
        if (ib < ia) ib = ia;
    }
    const bool full = ia == 0 && ib == 4;

    auto emit = [&](int row_abs, uint32_t (&val)[NC][4]) {
        const int y = row_abs - g.mv;
        if (row_abs < a_first || row_abs > a_last) return;                 /* wave-uniform */
        if (ib <= ia) return;
        if (TYPE == J2K_DWT97_INT && g.last) {
#pragma unroll
            for (int c = 0; c < NC; c++)
#pragma unroll
                for (int k = 0; k < 4; k++) val[c][k] = (uint32_t)((int32_t)((int32_t)val[c][k] + 128) >> 8);   /* :534-536 */
        }
        if (!FUSED) {
#pragma unroll
            for (int c = 0; c < NC; c++) {
                uint32_t *p = out_base + A[c].out_off + (size_t)y * A[c].out_stride + xa;
                if (full) {
                    *(uint4 *)p = make_uint4(val[c][0], val[c][1], val[c][2], val[c][3]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (k >= ia && k < ib) p[k] = val[c][k];
                }
            }
        } else {
            int v[4][4];
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int k = 0; k < 4; k++) v[c][k] = c < NC ? (int)val[c < NC ? c : 0][k] : 0;
            int xs = xa, n = ib;
            if (ia > 0) {                                  /* odd strip origin: drop the leading positions */
#pragma unroll
                for (int s = 1; s < 4; s++)
                    if (ia == s) {
#pragma unroll
                        for (int c = 0; c < 4; c++)
#pragma unroll
                            for (int k = 0; k + s < 4; k++) v[c][k] = v[c][k + s];
                    }
                xs = xa + ia; n = ib - ia;
            }
            /* the group holds components comp0 .. comp0 + NC - 1 of the tile; the MCT triple is
             * always a group of its own starting at component 0 */
            if (T->mct && comp0 == 0 && NC >= 3) pack_mct(T->c[0].transform, v);
            if (T->c[0].pix_step > 1) {
#pragma unroll
                for (int c = 0; c < NC; c++) pack_convert(*T, c, v[c]);
                pack_store_packed(*T, xs, y, n, v);
            } else {
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    pack_convert(*T, comp0 + c, v[c]);
                    pack_store_planar(*T, comp0 + c, xs, y, n, v[c]);
                }
            }
        }
    };

    uint32_t Lc[NC][4], Hc[NC][4], Ln[NC][4], Hn[NC][4];
    load_rows(s_first, Lc, Hc);
    for (int ye = s_first; ye <= s_last; ye += 2) {
        if (ye + 2 <= s_last) load_rows(ye + 2, Ln, Hn);
        uint32_t r_odd[NC][4], r_even[NC][4];
#pragma unroll
        for (int c = 0; c < NC; c++) {
            stream_hlift<TYPE>(Lc[c]);
            stream_hlift<TYPE>(Hc[c]);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (TYPE == J2K_DWT53) {
                    const uint32_t e = O::s1(Lc[c][k], Hp[c][k], Hc[c][k]);        /* row ye     */
                    const uint32_t o = O::s2(Hp[c][k], Sa[c][k], e);               /* row ye - 1 */
                    Hp[c][k] = Hc[c][k]; Sa[c][k] = e;
                    r_odd[c][k] = o; r_even[c][k] = e;
                } else {
                    const uint32_t e1 = O::s1(Lc[c][k], Hp[c][k], Hc[c][k]);       /* row ye,     after delta */
                    const uint32_t o2 = O::s2(Hp[c][k], Sa[c][k], e1);             /* row ye - 1, after gamma */
                    const uint32_t e3 = O::s3(Sa[c][k], Sb[c][k], o2);             /* row ye - 2, finished    */
                    const uint32_t o4 = O::s4(Sb[c][k], Sc[c][k], e3);             /* row ye - 3, finished    */
                    Hp[c][k] = Hc[c][k]; Sa[c][k] = e1; Sb[c][k] = o2; Sc[c][k] = e3;
                    r_odd[c][k] = o4; r_even[c][k] = e3;
                }
            }
        }
        emit(ye - DELAY - 1, r_odd);
        emit(ye - DELAY, r_even);
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int k = 0; k < 4; k++) { Lc[c][k] = Ln[c][k]; Hc[c][k] = Hn[c][k]; }
    }
}

/* plain level: grid.z indexes the DwtTileArgs table (one plane each), blockDim = one wave */
template <int TYPE>
__global__ void __launch_bounds__(64)
k_idwt_stream(const DwtTileArgs *__restrict__ args, const uint32_t *__restrict__ ll_base,
              const uint32_t *__restrict__ band_base, uint32_t *__restrict__ out_base, int th)
{
    const DwtTileArgs A[1] = { args[blockIdx.z] };
    idwt_stream_body<TYPE, 1, false>(A, ll_base, band_base, out_base, nullptr, 0, th);
}

/* final level + inverse MCT + frame store: grid.z indexes the DwtFusedArgs table */
template <int TYPE, int NC>
__global__ void __launch_bounds__(64)
k_idwt_stream_pack(const DwtFusedArgs *__restrict__ args, const uint32_t *__restrict__ ll_base,
                   const uint32_t *__restrict__ band_base, const PackTile *__restrict__ tiles, int th)
{
    const DwtFusedArgs &F = args[blockIdx.z];
    DwtTileArgs A[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) A[c] = F.a[c];
    idwt_stream_body<TYPE, NC, true>(A, ll_base, band_base, nullptr, tiles + F.pack_tile, F.comp0, th);
}

}  // namespace htj2k

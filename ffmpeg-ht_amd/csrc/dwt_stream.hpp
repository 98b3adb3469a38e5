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
 * Where every coefficient of a job provably fits 16 bits (reversible 5/3, M_b <= 15 in every band:
 * C16 / LL16 below, decided in htj2k_device.hip) the sub-bands the block decoder wrote are read as
 * 16-bit pairs and widened in registers: 2.5 + 1 bytes per sample at the final level.
 *
 * Round 3: a workgroup may hold up to eight such waves, which then take neighbouring strips and walk the same rows in step
 * (stream_strip; what that does for the DRAM pages is worth 5-10 % of a launch); the 5/3 levels of 8-bit pictures run on
 * pairs of 16-bit samples (PK: v_pk_* lifting, RCT and clip, half the instructions) where the host has proven that no
 * intermediate leaves 16 bits; loads and stores take their row base from scalar registers (sgpr_ptr); and the first three
 * levels of a plane are one launch with the LL bands in between held in LDS (k_idwt_stream_ll16_x3).
 *
 * Boundaries: positions outside the line are fetched through the
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

#define STREAM_TW 244          /* most output columns a wave can own: lanes 1..61 whole + slack for an odd origin; multiple of 4.
                                * The host picks the strip width per launch (StreamGrid::tw <= STREAM_TW): a width whose
                                * row of output bytes is a multiple of 64 keeps the stores of neighbouring strips out of
                                * each other's lines where the plane is wide enough to pay for the idle lanes. */

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
 * active.  Afterwards lanes 1..62 hold finished samples.
 * mirror_l / mirror_r (fast path): this lane is the first / last one inside a line that starts
 * on an even position / ends on an odd one.  The symmetric extension makes the odd sample left
 * of the line the mirror image of the lane's own first odd sample, and the even sample right of
 * it the image of its own last even sample -- after every lifting step, since the steps are
 * symmetric -- so the neighbour lane need not hold reflected data. */
template <int TYPE>
__device__ __forceinline__ void stream_hlift(uint32_t (&v)[4], bool mirror_l = false, bool mirror_r = false)
{
    using O = LiftOps<TYPE>;
    uint32_t lo = dpp_from_left(v[3]);
    if (mirror_l) lo = v[1];
    v[0] = O::s1(v[0], lo, v[1]);
    v[2] = O::s1(v[2], v[1], v[3]);
    uint32_t re = dpp_from_right(v[0]);
    if (mirror_r) re = v[2];
    v[1] = O::s2(v[1], v[0], v[2]);
    v[3] = O::s2(v[3], v[2], re);
    if (TYPE != J2K_DWT53) {
        lo = dpp_from_left(v[3]);
        if (mirror_l) lo = v[1];
        v[0] = O::s3(v[0], lo, v[1]);
        v[2] = O::s3(v[2], v[1], v[3]);
        re = dpp_from_right(v[0]);
        if (mirror_r) re = v[2];
        v[1] = O::s4(v[1], v[0], v[2]);
        v[3] = O::s4(v[3], v[2], re);
    }
}

/* ---- 5/3 lifting on pairs of 16-bit samples (PK, below) ----
 * A dword of a 16-bit band holds two neighbouring samples of that band: of a lane's four positions e0 o0 e1 o1 the register
 * E = (e0, e1) is the dword it loaded from the low band and O = (o0, o1) the one from the high band.  v_pk_* instructions lift
 * both pairs at once; the vertical lifting, the RCT and the clip run on the same pairs, so a step takes half the
 * arithmetic instructions of the 32-bit form and no widening.  Sums wrap at 16 bits: the host starts these kernels only where
 * interval arithmetic over the bands' M_b and the checked range of the LL band shows that no intermediate can
 * (htj2k_device.hip, pk16_bounds). */
/* a wave-uniform pointer, pinned to scalar registers and opaque to the optimiser: `sgpr_ptr(row) + lane_offset` then is a
 * load with a scalar base and a 32-bit vector offset instead of a 64-bit vector add per load (left alone, the compiler hoists
 * `base + lane_offset` out of the row loop as a 64-bit vector value and adds the row offset to it with v_lshl_add_u64) */
typedef __attribute__((address_space(1))) const char g_cchar;
typedef __attribute__((address_space(1))) const uint32_t g_cu32;
__device__ __forceinline__ g_cchar *sgpr_ptr(const void *p)               /* p: global memory */
{
    uint64_t v = (uint64_t)p;
    asm("" : "+s"(v));
    return (g_cchar *)v;
}

typedef unsigned short pk_u16 __attribute__((ext_vector_type(2)));
typedef short pk_i16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk_u16 pk_from(uint32_t v) { return __builtin_bit_cast(pk_u16, v); }
__device__ __forceinline__ uint32_t pk_bits(pk_u16 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ pk_u16 pk_sar(pk_u16 v, int n) { return __builtin_bit_cast(pk_u16, __builtin_bit_cast(pk_i16, v) >> (short)n); }
__device__ __forceinline__ uint32_t pk_s1(uint32_t c, uint32_t a, uint32_t b)   /* jpeg2000dwt.c:321-322 on both halves */
{
    const pk_u16 two = { 2, 2 };
    return pk_bits(pk_from(c) - pk_sar(pk_from(a) + pk_from(b) + two, 2));
}
__device__ __forceinline__ uint32_t pk_s2(uint32_t c, uint32_t a, uint32_t b)   /* :323-324 */
{
    return pk_bits(pk_from(c) + pk_sar(pk_from(a) + pk_from(b), 1));
}
/* E = (e0, e1), O = (o0, o1) of every lane; the odd sample left of e0 and the even sample right of o1 come from the
 * neighbouring lanes (or, at the ends of the line, from the lane itself: the mirror rule of stream_hlift) */
__device__ __forceinline__ void stream_hlift_pk(uint32_t &E, uint32_t &O, uint32_t sel_l, uint32_t sel_r)
{
    /* sel_l / sel_r: the lane's byte selectors for v_perm -- 0x05040302 takes (neighbour's high half, own low half), at the
     * end of a line 0x05040504 / 0x03020302 repeat the lane's own sample */
    E = pk_s1(E, __builtin_amdgcn_perm(O, dpp_from_left(O), sel_l), O);  /* (left o1, o0) + (o0, o1) */
    O = pk_s2(O, E, __builtin_amdgcn_perm(dpp_from_right(E), E, sel_r)); /* (e0, e1) + (e1, right e0) */
}

/* the components one wave reconstructs in lockstep share the level geometry g of a[0] */
struct DwtFusedArgs {
    DwtTileArgs a[4];
    int32_t ncomp;            /* 1, 3 or 4 */
    int32_t pack_tile;        /* index of the PackTile the group belongs to */
    int32_t comp0;            /* first component of that tile held by the group */
    int32_t pad;
};

/* FAST (chosen per wave, see idwt_stream_body): the line starts on an even position and its
 * length is a multiple of 4, so every lane is either wholly inside the line or wholly outside;
 * loads are unconditional 8-byte loads (columns of outside lanes are clamped, the two values
 * that cross the line ends come from the mirror rule of stream_hlift), stores are one
 * predicated 16-byte (plain) or 12-byte (rgb24) store per row, and nothing in the loop body
 * has a data-dependent trip count: the compiler can then keep the next step's loads in
 * flight across the current step (s_waitcnt vmcnt(N > 0)).  FAST && FUSED is the rgb24 case:
 * three 8-bit components on one packed plane. */
/* C16: the sub-bands the block decoder wrote are 16-bit samples (the coefficient buffer read as int16_t with the same
 * element offsets; reversible 5/3 planes whose bands all have M_b <= 15, see htj2k_device.hip); LL16: so is the LL band,
 * i.e. this is the first level.  Both only on the FAST path (aligned sample pairs). */
/* OUTK: what the FAST && FUSED store writes per lane and row (four pixel columns): 0 rgb24 (three 8-bit components
 * interleaved, 12 bytes), 1 rgb48 (three 16-bit components interleaved, 24 bytes), 2 one 8-bit plane (4 bytes),
 * 3 one 16-bit plane (8 bytes) */
/* !FUSED with OUTK == 16: the level's output -- the next level's LL band -- is written as 16-bit samples too (same
 * element offsets, the buffer read as int16_t), and *ovf is set when a sample does not fit: an LL band of a real
 * picture stays in the picture's range, but nothing bounds what crafted or corrupt coefficients add up to.  The host
 * then runs the transform again with 32-bit LL bands (htj2k_device.hip, job_settle). */
/* PK (FAST, 16-bit sub-bands and LL band; a plain level that stores a 16-bit LL band, or the fused final level with 8-bit
 * output of 8-bit components, OUTK 0 or 2): the whole step on pairs of 16-bit samples, see stream_hlift_pk */
/* LLG: the 16-bit LL band is in global memory (everywhere but in k_idwt_stream_ll16_x3, whose LL inputs may be LDS windows) */
template <int TYPE, int NC, bool FUSED, bool FAST, bool C16 = false, bool LL16 = false, int OUTK = 0, bool PK = false, bool LLG = FUSED>
__device__ __forceinline__ void
idwt_stream_impl(const DwtTileArgs (&A)[NC], const uint32_t *__restrict__ ll_base, const uint32_t *__restrict__ band_base,
                 uint32_t *__restrict__ out_base, const PackTile *__restrict__ T, int comp0, int th, int tw, int bx, int by,
                 int *__restrict__ ovf = nullptr, int ovf_bits = 16, int y0_rows = -1, int dir_in = 0, int ll_row0 = 0, int out_row0 = 0)
{
    /* ll_row0 / out_row0 (FAST 16-bit paths only): the LL input / the output is a window whose first row is row ll_row0 /
     * out_row0 of the band (k_idwt_stream_ll16_x3's LDS windows) */
    using O = LiftOps<TYPE>;
    static_assert(!PK || (TYPE == J2K_DWT53 && FAST && C16 && LL16 && (FUSED ? (OUTK == 0 || OUTK == 2) : OUTK == 16)),
                  "PK: 16-bit levels of 16-bit jobs and their 8-bit fused fast stores");
    constexpr int HALO = O::HALO, DELAY = O::DELAY;
    const DwtLevel g = A[0].g;
    /* the strip: columns [bx tw, bx tw + tw), rows [by th, by th + th) -- or, for callers that cut a level their own way
     * (k_idwt_stream_ll16_x3), rows [y0_rows, y0_rows + th) walked in direction dir_in */
    const int x0 = bx * tw, y0 = y0_rows >= 0 ? y0_rows : by * th;
    if (x0 >= g.lh || y0 >= g.lv) return;
    const LineMap LX(g.mh, g.lh), LY(g.mv, g.lv);
    const int lane = threadIdx.x & 63;                   /* (k_idwt_stream_ll16_x3 runs four waves per workgroup) */

    /* ---- columns of this lane ---- */
    const int ax0 = (g.mh + x0 - 4) & ~1;                 /* even; lane 1 starts at or one before the strip */
    const int pe0 = ax0 + 4 * lane;
    const bool interior = ax0 >= g.mh && ax0 + 256 <= g.mh + g.lh;      /* wave-uniform: no reflection, 8-byte loads */
    /* storage columns; positions outside the line go through the reflection, so every load
     * address is valid and no load is predicated (results further than HALO + 2 outside the
     * line are never used) */
    int col[4];
    bool mirror_l = false, mirror_r = false;
    if (FAST) {
        const int q = (pe0 - g.mh) >> 1;                  /* pair index of (e0, o0); g.mh is even */
        const int qc = min(max(q, 0), LX.nl - 2);         /* lanes outside the line: any valid address */
        col[0] = qc; col[2] = qc + 1; col[1] = LX.nl + qc; col[3] = col[1] + 1;
        mirror_l = pe0 == g.mh;
        mirror_r = pe0 + 4 == g.mh + g.lh;
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) col[k] = LX.idx(pe0 + k);
    }
    const uint32_t pk_sel_l = mirror_l ? 0x05040504u : 0x05040302u, pk_sel_r = mirror_r ? 0x03020302u : 0x05040302u;
    const uint32_t *llp[NC], *bandp[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        llp[c] = ll_base + A[c].ll_off;
        bandp[c] = band_base + A[c].g.plane_off;
    }

    /* ---- rows of this strip ---- */
    const int a_first = g.mv + y0, a_last = g.mv + min(y0 + th, g.lv) - 1;
    /* Odd strips walk UP their rows, even strips down.  The lifting is symmetric, so the upward walk is
     * the same recurrence with the high row of a step taken below instead of above its low row.  Two
     * vertically adjacent strips -- which start together: consecutive strip numbers on one XCD -- then
     * read the HALO rows they share at the same moment (both at their start, or both at their end)
     * and the second reader hits in L2 instead of fetching them again. */
    const int dir = dir_in ? dir_in : ((by & 1) ? -1 : 1);
    const int s_first = dir > 0 ? (a_first - HALO) & ~1 : (a_last + HALO + 1) & ~1;
    const int s_last = dir > 0 ? (a_last + DELAY + 1) & ~1 : (a_first - DELAY) & ~1;

    /* The loads of a step: unconditional and the same number on every pass of the loop, so
     * that the compiler can wait for exactly the older step's loads (s_waitcnt vmcnt(N)) while
     * the newer step's are in flight. */
    auto load_rows = [&](int ye, uint32_t (&Lr)[NC][4], uint32_t (&Hr)[NC][4]) {
        int iy[2];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int ay = (dir > 0 ? min(ye, s_last) : max(ye, s_last)) + r * dir;   /* the one prefetch past the strip re-reads its last rows */
            iy[r] = (ay >= LY.i0 && ay < LY.i1) ? ((ay & 1) ? LY.nl + ((ay - LY.fo) >> 1) : ((ay - LY.fe) >> 1)) : LY.idx(ay);
            /* wave-uniform, but the reflection's division runs on the vector unit: say so, and the row pointers below stay in
             * scalar registers */
            iy[r] = __builtin_amdgcn_readfirstlane(iy[r]);
        }
        /* even absolute rows are vertical-low rows (the reflection keeps the parity): their
         * low-horizontal half is the previous level's output */
        if (C16) {
            static_assert(!C16 || FAST, "16-bit sub-bands: fast path only");
            /* The buffers carry the dwords as loaded -- two 16-bit samples each, in Lr[1], Hr[0], Hr[1] (and Lr[0] under
             * LL16) -- and widen() makes samples of them when the step that uses them begins.  Widened here, the values
             * that cross the loop's back edge are results of the loads, and the compiler waits for every load in flight
             * (s_waitcnt vmcnt(0)) at the bottom of the loop: the next step's loads were not ahead of anything. */
#pragma unroll
            for (int c = 0; c < NC; c++) {
                /* a wave-uniform row pointer plus an unsigned 32-bit lane offset: the load takes its base from scalar
                 * registers and the address costs no vector instruction */
                const uint16_t *b16 = (const uint16_t *)band_base + A[c].g.plane_off;
                g_cchar *brow0 = sgpr_ptr(b16 + (size_t)iy[0] * A[c].g.stride), *brow1 = sgpr_ptr(b16 + (size_t)iy[1] * A[c].g.stride);
                uint32_t cb0 = (uint32_t)col[0] * 2u, cb1 = (uint32_t)col[1] * 2u;         /* col[] are even here: aligned pairs */
                /* (instruction selection works block by block: the zero-extension of the offset has to be seen in the block
                 * of the load, so the offsets are re-made "here" for the compiler) */
                asm volatile("" : "+v"(cb0), "+v"(cb1));
                Lr[c][1] = *(g_cu32 *)(brow0 + cb1);
                Hr[c][0] = *(g_cu32 *)(brow1 + cb0);
                Hr[c][1] = *(g_cu32 *)(brow1 + cb1);
                if (LL16 && LLG) {
                    g_cchar *lrow = sgpr_ptr((const uint16_t *)ll_base + A[c].ll_off + (size_t)(iy[0] - ll_row0) * A[c].ll_stride);
                    Lr[c][0] = *(g_cu32 *)(lrow + cb0);
                } else if (LL16) {                             /* k_idwt_stream_ll16_x3: global memory or an LDS window */
                    const uint16_t *lrow = (const uint16_t *)ll_base + A[c].ll_off + (size_t)(iy[0] - ll_row0) * A[c].ll_stride;
                    Lr[c][0] = *(const uint32_t *)(lrow + col[0]);
                } else {
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    typedef __attribute__((address_space(1))) const u32x2 g_cu2;
                    g_cchar *lrow = sgpr_ptr(llp[c] + (size_t)iy[0] * A[c].ll_stride);
                    uint32_t cl = (uint32_t)col[0] * 4u;
                    asm volatile("" : "+v"(cl));
                    const u32x2 le = *(g_cu2 *)(lrow + cl);
                    Lr[c][0] = le.x; Lr[c][2] = le.y;
                }
            }
        } else if (FAST) {                                     /* (as above: scalar row bases, 32-bit lane offsets) */
#pragma unroll
            for (int c = 0; c < NC; c++) {
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                typedef __attribute__((address_space(1))) const u32x2 g_cu2;
                g_cchar *lrow = sgpr_ptr(llp[c] + (size_t)iy[0] * A[c].ll_stride);
                g_cchar *brow0 = sgpr_ptr(bandp[c] + (size_t)iy[0] * A[c].g.stride), *brow1 = sgpr_ptr(bandp[c] + (size_t)iy[1] * A[c].g.stride);
                uint32_t cb0 = (uint32_t)col[0] * 4u, cb1 = (uint32_t)col[1] * 4u;
                asm volatile("" : "+v"(cb0), "+v"(cb1));
                const u32x2 le = *(g_cu2 *)(lrow + cb0), lo = *(g_cu2 *)(brow0 + cb1);
                const u32x2 he = *(g_cu2 *)(brow1 + cb0), ho = *(g_cu2 *)(brow1 + cb1);
                Lr[c][0] = le.x; Lr[c][1] = lo.x; Lr[c][2] = le.y; Lr[c][3] = lo.y;
                Hr[c][0] = he.x; Hr[c][1] = ho.x; Hr[c][2] = he.y; Hr[c][3] = ho.y;
            }
        } else if (interior) {
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const uint32_t *lrow = llp[c] + (size_t)iy[0] * A[c].ll_stride;
                const uint32_t *brow0 = bandp[c] + (size_t)iy[0] * A[c].g.stride;
                const uint32_t *brow1 = bandp[c] + (size_t)iy[1] * A[c].g.stride;
                const uint2 le = *(const uint2 *)(lrow + col[0]), lo = *(const uint2 *)(brow0 + col[1]);
                const uint2 he = *(const uint2 *)(brow1 + col[0]), ho = *(const uint2 *)(brow1 + col[1]);
                Lr[c][0] = le.x; Lr[c][1] = lo.x; Lr[c][2] = le.y; Lr[c][3] = lo.y;
                Hr[c][0] = he.x; Hr[c][1] = ho.x; Hr[c][2] = he.y; Hr[c][3] = ho.y;
            }
        } else {
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const uint32_t *lrow = llp[c] + (size_t)iy[0] * A[c].ll_stride;
                const uint32_t *brow0 = bandp[c] + (size_t)iy[0] * A[c].g.stride;
                const uint32_t *brow1 = bandp[c] + (size_t)iy[1] * A[c].g.stride;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    Lr[c][k] = ((k & 1) ? brow0 : lrow)[col[k]];
                    Hr[c][k] = brow1[col[k]];
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
    const int x_hi = min(x0 + tw, g.lh);
    int ia = 0, ib = 0;                                   /* valid positions of the quadruple: [ia, ib) */
    if (lane >= 1 && lane <= 62) {
        ia = max(0, x0 - xa);
        ib = min(4, x_hi - xa);
        if (ib < ia) ib = ia;
    }
    const bool full = ia == 0 && ib == 4;

    /* fast fused path: frame geometry and per-component conversion constants in scalars */
    uint8_t *f_dst = nullptr, *f_row0 = nullptr;             /* f_row0 + f_xoff: the same address as a wave-uniform row and a lane offset */
    uint32_t f_xoff = 0;
    const bool lane_okq = lane >= 1 && xa + 4 <= x_hi;       /* (= lane_ok below) */
    int f_ls = 0, f_dc[3] = { 0, 0, 0 }, f_hi[3] = { 0, 0, 0 }, f_sh[3] = { 0, 0, 0 };
    bool f_mct = false;
    if (FAST && FUSED) {
        const PackComp &C0 = T->c[OUTK >= 2 ? comp0 : 0];
        f_ls = T->out.linesize[C0.out_plane];
        f_dst = T->out.ptr[C0.out_plane] + (size_t)C0.out_y * f_ls + (size_t)(C0.out_x + xa) * (OUTK == 0 ? 3 : OUTK == 1 ? 6 : OUTK == 2 ? 1 : 2);
        f_row0 = T->out.ptr[C0.out_plane] + (size_t)C0.out_y * f_ls;
        f_xoff = (uint32_t)(lane_okq ? (C0.out_x + xa) * (OUTK == 0 ? 3 : OUTK == 1 ? 6 : OUTK == 2 ? 1 : 2) : 0);
        f_mct = OUTK < 2 && T->mct != 0;
#pragma unroll
        for (int c = 0; c < (OUTK >= 2 ? 1 : 3); c++) {
            const PackComp &C = T->c[OUTK >= 2 ? comp0 : c];
            f_dc[c] = 1 << (C.cbps - 1);
            f_hi[c] = (1 << C.cbps) - 1;
            f_sh[c] = T->precision - C.cbps;
        }
    }
    const bool lane_ok = lane >= 1 && xa + 4 <= x_hi;     /* FAST: the quadruple is wholly inside the strip */

    constexpr int NW = OUTK == 0 ? 3 : OUTK == 1 ? 6 : OUTK == 2 ? 1 : 2;   /* dwords a lane stores per row */
    uint32_t wk[2][6] = { { 0, 0, 0, 0, 0, 0 }, { 0, 0, 0, 0, 0, 0 } };   /* fast fused path: the two rows' store data ... */
    uintptr_t ak[2] = { 0, 0 };                              /* ... and addresses of the last step */
    uint32_t ovf_acc = 0;                                    /* OUTK == 16: a stored sample did not fit 16 bits */
    auto emit = [&](int row_abs, uint32_t (&val)[NC][4], int slot) {
        const int y = row_abs - g.mv;
        if (row_abs < a_first || row_abs > a_last) return;                 /* wave-uniform */
        if (FAST) {
            if (TYPE == J2K_DWT97_INT && g.last) {
#pragma unroll
                for (int c = 0; c < NC; c++)
#pragma unroll
                    for (int k = 0; k < 4; k++) val[c][k] = (uint32_t)((int32_t)((int32_t)val[c][k] + 128) >> 8);   /* :534-536 */
            }
            if (!FUSED && OUTK == 16 && PK) {
                /* val[c][0] = (x0, x2), val[c][1] = (x1, x3): two byte permutes put them in order; x + hb has no bit at or above
                 * ovf_bits in either half exactly when both fit ovf_bits bits */
                const uint32_t hb2 = (1u << (ovf_bits - 1)) * 0x10001u, hm2 = ((0xFFFFu << ovf_bits) & 0xFFFFu) * 0x10001u;
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    uint16_t *p = (uint16_t *)out_base + A[c].out_off + (size_t)(y - out_row0) * A[c].out_stride + xa;
                    const uint32_t t = pk_bits(pk_from(val[c][0]) + pk_from(hb2)) | pk_bits(pk_from(val[c][1]) + pk_from(hb2));
                    if (lane_ok) {
                        ovf_acc |= t & hm2;
                        *(uint2 *)p = make_uint2(__builtin_amdgcn_perm(val[c][1], val[c][0], 0x05040100), __builtin_amdgcn_perm(val[c][1], val[c][0], 0x07060302));
                    }
                }
            } else if (!FUSED && OUTK == 16) {
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    uint16_t *p = (uint16_t *)out_base + A[c].out_off + (size_t)(y - out_row0) * A[c].out_stride + xa;
                    /* v + 0x8000 < 0x10000 exactly for v in [-32768, 32767] (ovf_bits = 16; fewer bits: the tests' way to
                     * make ordinary frames take the overflow path) */
                    const uint32_t hb = 1u << (ovf_bits - 1);
                    const uint32_t t = (val[c][0] + hb) | (val[c][1] + hb) | (val[c][2] + hb) | (val[c][3] + hb);
                    if (lane_ok) {
                        ovf_acc |= t >> ovf_bits;
                        *(uint2 *)p = make_uint2((val[c][0] & 0xFFFFu) | (val[c][1] << 16), (val[c][2] & 0xFFFFu) | (val[c][3] << 16));
                    }
                }
            } else if (!FUSED) {
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    uint32_t *p = out_base + A[c].out_off + (size_t)y * A[c].out_stride + xa;
                    if (lane_ok) *(uint4 *)p = make_uint4(val[c][0], val[c][1], val[c][2], val[c][3]);
                }
            } else if (PK) {
                /* val[c][0] = (x0, x2), val[c][1] = (x1, x3) of component c.  Components of 8 bits (the host's condition): DC
                 * shift and clip are "clamp to [-128, 127], flip bit 7 of the byte", and the flip is done on the packed dwords. */
                typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
                typedef u32x3 u32x3_a4 __attribute__((aligned(4)));
                typedef __attribute__((address_space(1))) u32x3_a4 g_u32x3;
                typedef __attribute__((address_space(1))) uint32_t g_u32;
                const pk_i16 lo = { -128, -128 }, hi = { 127, 127 };
                auto clamp8 = [&](uint32_t v) {
                    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(pk_i16, v), lo), hi));
                };
                g_cchar *const rowp = sgpr_ptr(f_row0 + (size_t)y * f_ls);
                uint32_t xo = f_xoff;
                asm volatile("" : "+v"(xo));                             /* (see load_rows) */
                if (OUTK == 0) {
                    uint32_t e[3], o[3];
                    if (f_mct) {                                       /* rct_int, jpeg2000dsp.c:78-91, on both pairs */
                        const pk_u16 y0 = pk_from(val[0][0]), y1 = pk_from(val[0][1]);
                        const pk_u16 b0 = pk_from(val[NC > 1 ? 1 : 0][0]), b1 = pk_from(val[NC > 1 ? 1 : 0][1]);
                        const pk_u16 r0 = pk_from(val[NC > 2 ? 2 : 0][0]), r1 = pk_from(val[NC > 2 ? 2 : 0][1]);
                        const pk_u16 g0 = y0 - pk_sar(r0 + b0, 2), g1 = y1 - pk_sar(r1 + b1, 2);
                        /* the host's bounds cover r + b and g; g + r and g + b may leave 16 bits and saturate instead, which
                         * the clip below cannot tell from the true sum */
                        auto adds = [](pk_u16 a, pk_u16 b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(pk_i16, a), __builtin_bit_cast(pk_i16, b))); };
                        e[0] = adds(g0, r0); e[1] = pk_bits(g0); e[2] = adds(g0, b0);
                        o[0] = adds(g1, r1); o[1] = pk_bits(g1); o[2] = adds(g1, b1);
                    } else {
#pragma unroll
                        for (int c = 0; c < 3; c++) { e[c] = val[c < NC ? c : 0][0]; o[c] = val[c < NC ? c : 0][1]; }
                    }
#pragma unroll
                    for (int c = 0; c < 3; c++) { e[c] = clamp8(e[c]); o[c] = clamp8(o[c]); }
                    /* bytes: e[c] = (c0 . c2 .), o[c] = (c1 . c3 .) -> r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3 */
                    const uint32_t A0 = __builtin_amdgcn_perm(e[1], e[0], 0x06020400);    /* r0 g0 r2 g2 */
                    const uint32_t B0 = __builtin_amdgcn_perm(o[0], e[2], 0x06020400);    /* b0 r1 b2 r3 */
                    const uint32_t C0 = __builtin_amdgcn_perm(o[2], o[1], 0x06020400);    /* g1 b1 g3 b3 */
                    u32x3 w;
                    w.x = __builtin_amdgcn_perm(B0, A0, 0x05040100) ^ 0x80808080u;       /* A.0 A.1 B.0 B.1 */
                    w.y = __builtin_amdgcn_perm(A0, C0, 0x07060100) ^ 0x80808080u;       /* C.0 C.1 A.2 A.3 */
                    w.z = __builtin_amdgcn_perm(C0, B0, 0x07060302) ^ 0x80808080u;       /* B.2 B.3 C.2 C.3 */
                    if (lane_ok) *(g_u32x3 *)(rowp + xo) = w;
                    wk[slot][0] = w.x; wk[slot][1] = w.y; wk[slot][2] = w.z;
                } else {
                    const uint32_t e = clamp8(val[0][0]), o = clamp8(val[0][1]);
                    const uint32_t w = __builtin_amdgcn_perm(o, e, 0x06020400) ^ 0x80808080u;   /* x0 x1 x2 x3 */
                    if (lane_ok) *(g_u32 *)(rowp + xo) = w;
                    wk[slot][0] = w;
                }
            } else {
                int v[4][4];
#pragma unroll
                for (int c = 0; c < 4; c++)
#pragma unroll
                    for (int k = 0; k < 4; k++) v[c][k] = c < NC ? (int)val[c < NC ? c : 0][k] : 0;
                constexpr int NO = OUTK >= 2 ? 1 : 3;               /* components this store writes */
                if (f_mct) {
                    pack_mct(TYPE, v);
                } else if (TYPE == J2K_DWT97) {
#pragma unroll
                    for (int c = 0; c < NO; c++)
#pragma unroll
                        for (int k = 0; k < 4; k++) v[c][k] = __float2int_rn(__int_as_float(v[c][k]));
                }
#pragma unroll
                for (int c = 0; c < NO; c++)
#pragma unroll
                    for (int k = 0; k < 4; k++) v[c][k] = min(max(v[c][k] + f_dc[c], 0), f_hi[c]) << f_sh[c];
                /* the frame pointer comes out of a descriptor: say that it is global memory, or the
                 * store is a FLAT one (slower, and the compiler then waits vmcnt(0) everywhere) */
                typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
                typedef u32x3 u32x3_a4 __attribute__((aligned(4)));
                typedef __attribute__((address_space(1))) u32x3_a4 g_u32x3;
                const uintptr_t addr = (uintptr_t)(f_dst + (size_t)y * f_ls);
                if (OUTK == 0) {
                    u32x3 w;
                    w.x = (uint32_t)v[0][0] | ((uint32_t)v[1][0] << 8) | ((uint32_t)v[2][0] << 16) | ((uint32_t)v[0][1] << 24);
                    w.y = (uint32_t)v[1][1] | ((uint32_t)v[2][1] << 8) | ((uint32_t)v[0][2] << 16) | ((uint32_t)v[1][2] << 24);
                    w.z = (uint32_t)v[2][2] | ((uint32_t)v[0][3] << 8) | ((uint32_t)v[1][3] << 16) | ((uint32_t)v[2][3] << 24);
                    if (lane_ok) *(g_u32x3 *)addr = w;
                    wk[slot][0] = w.x; wk[slot][1] = w.y; wk[slot][2] = w.z;
                } else if (OUTK == 1) {                              /* r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3, 16 bits each */
                    u32x3 w0, w1;
                    w0.x = (uint32_t)v[0][0] | ((uint32_t)v[1][0] << 16); w0.y = (uint32_t)v[2][0] | ((uint32_t)v[0][1] << 16);
                    w0.z = (uint32_t)v[1][1] | ((uint32_t)v[2][1] << 16); w1.x = (uint32_t)v[0][2] | ((uint32_t)v[1][2] << 16);
                    w1.y = (uint32_t)v[2][2] | ((uint32_t)v[0][3] << 16); w1.z = (uint32_t)v[1][3] | ((uint32_t)v[2][3] << 16);
                    if (lane_ok) { *(g_u32x3 *)addr = w0; *(g_u32x3 *)(addr + 12) = w1; }
                    wk[slot][0] = w0.x; wk[slot][1] = w0.y; wk[slot][2] = w0.z; wk[slot][3] = w1.x; wk[slot][4] = w1.y; wk[slot][5] = w1.z;
                } else if (OUTK == 2) {
                    typedef __attribute__((address_space(1))) uint32_t g_u32;
                    const uint32_t w = (uint32_t)v[0][0] | ((uint32_t)v[0][1] << 8) | ((uint32_t)v[0][2] << 16) | ((uint32_t)v[0][3] << 24);
                    if (lane_ok) *(g_u32 *)addr = w;
                    wk[slot][0] = w;
                } else {
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    typedef u32x2 u32x2_a4 __attribute__((aligned(4)));
                    typedef __attribute__((address_space(1))) u32x2_a4 g_u32x2;
                    u32x2 w;
                    w.x = (uint32_t)v[0][0] | ((uint32_t)v[0][1] << 16); w.y = (uint32_t)v[0][2] | ((uint32_t)v[0][3] << 16);
                    if (lane_ok) *(g_u32x2 *)addr = w;
                    wk[slot][0] = w.x; wk[slot][1] = w.y;
                }
                ak[slot] = addr;
            }
            return;
        }
        /* (no early return for lanes without output: the values they hold feed their neighbours' lifting by DPP, and a
         * divergent region in front of the next step invites the compiler to sink those moves into it) */
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

    auto step = [&](int ye, uint32_t (&Lc)[NC][4], uint32_t (&Hc)[NC][4]) {
        uint32_t r_odd[NC][4], r_even[NC][4];
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if (PK) {                                          /* the dwords of load_rows are the pairs */
                stream_hlift_pk(Lc[c][0], Lc[c][1], pk_sel_l, pk_sel_r);
                stream_hlift_pk(Hc[c][0], Hc[c][1], pk_sel_l, pk_sel_r);
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const uint32_t e = pk_s1(Lc[c][k], Hp[c][k], Hc[c][k]);        /* row ye     */
                    const uint32_t o = pk_s2(Hp[c][k], Sa[c][k], e);               /* row ye - 1 */
                    Hp[c][k] = Hc[c][k]; Sa[c][k] = e;
                    r_odd[c][k] = o; r_even[c][k] = e;
                }
                r_odd[c][2] = r_odd[c][3] = r_even[c][2] = r_even[c][3] = 0;
                continue;
            }
            if (C16) {                                         /* the dwords of load_rows -> samples */
                auto lo16 = [](uint32_t v) { return (uint32_t)(int32_t)(int16_t)(v & 0xFFFFu); };
                auto hi16 = [](uint32_t v) { return (uint32_t)((int32_t)v >> 16); };
                const uint32_t lo = Lc[c][1], he = Hc[c][0], ho = Hc[c][1];
                if (LL16) { const uint32_t le = Lc[c][0]; Lc[c][0] = lo16(le); Lc[c][2] = hi16(le); }
                Lc[c][1] = lo16(lo); Lc[c][3] = hi16(lo);
                Hc[c][0] = lo16(he); Hc[c][1] = lo16(ho); Hc[c][2] = hi16(he); Hc[c][3] = hi16(ho);
            }
            stream_hlift<TYPE>(Lc[c], mirror_l, mirror_r);
            stream_hlift<TYPE>(Hc[c], mirror_l, mirror_r);
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
        emit(ye - dir * (DELAY + 1), r_odd, 0);
        emit(ye - dir * DELAY, r_even, 1);
    };

    /* two register buffers, loop unrolled by two: the loads of step s + 1 are in flight while
     * step s computes and stores */
    uint32_t LA[NC][4], HA[NC][4], LB[NC][4], HB[NC][4];
    load_rows(s_first, LA, HA);
    /* an empty asm that "rewrites" the older buffer and clobbers memory: the newer buffer's loads
     * cannot sink below it and the older buffer's arithmetic cannot rise above it, so the loads
     * really are issued a whole step ahead (the instruction scheduler otherwise interleaves them
     * with the arithmetic they were meant to overlap) */
    auto pin = [&](uint32_t (&Lr)[NC][4], uint32_t (&Hr)[NC][4]) {
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if (C16 && LL16) asm volatile("" : "+v"(Lr[c][0]), "+v"(Lr[c][1]), "+v"(Hr[c][0]), "+v"(Hr[c][1]) : : "memory");
            else if (C16) asm volatile("" : "+v"(Lr[c][0]), "+v"(Lr[c][1]), "+v"(Lr[c][2]), "+v"(Hr[c][0]), "+v"(Hr[c][1]) : : "memory");
            else asm volatile("" : "+v"(Lr[c][0]), "+v"(Lr[c][1]), "+v"(Lr[c][2]), "+v"(Lr[c][3]),
                                   "+v"(Hr[c][0]), "+v"(Hr[c][1]), "+v"(Hr[c][2]), "+v"(Hr[c][3]) : : "memory");
        }
        /* the registers of the last step's store data stay allocated until the next loads are
         * out: rewriting them earlier costs a wait for those stores (and, vmcnt being in order,
         * for every load issued before them) */
        if (FAST && FUSED) {
#pragma unroll
            for (int i = 0; i < NW; i++) asm volatile("" : : "v"(wk[0][i]), "v"(wk[1][i]));
            asm volatile("" : : "v"(ak[0]), "v"(ak[1]));
        }
    };
    if (NC == 1) {
        /* One component per wave: a step's loads are 1 KB per wave, and 8 waves per SIMD with one step in flight each
         * are 8 MB on the whole chip -- at ~1.6 us of loaded HBM latency that caps the kernel near 5 TB/s, which is
         * where the plain levels sat (4.1-4.4 TB/s against 5.2 for the three-component final level and 5.4 for
         * tools/microbench/strip_bw.hip, the same access pattern with every load of a strip in flight).  So: three
         * buffers, loads two steps ahead. */
        uint32_t LC[NC][4], HC[NC][4];
        load_rows(s_first + 2 * dir, LB, HB);
        for (int ye = s_first;; ye += 6 * dir) {
            load_rows(ye + 4 * dir, LC, HC);
            pin(LA, HA);
            step(ye, LA, HA);
            if (dir * (ye + 2 * dir - s_last) > 0) break;
            load_rows(ye + 6 * dir, LA, HA);
            pin(LB, HB);
            step(ye + 2 * dir, LB, HB);
            if (dir * (ye + 4 * dir - s_last) > 0) break;
            load_rows(ye + 8 * dir, LB, HB);
            pin(LC, HC);
            step(ye + 4 * dir, LC, HC);
            if (dir * (ye + 6 * dir - s_last) > 0) break;
        }
    } else
    for (int ye = s_first;; ye += 4 * dir) {
        load_rows(ye + 2 * dir, LB, HB);
        pin(LA, HA);
        step(ye, LA, HA);
        if (dir * (ye + 2 * dir - s_last) > 0) break;
        load_rows(ye + 4 * dir, LA, HA);
        pin(LB, HB);
        step(ye + 2 * dir, LB, HB);
        if (dir * (ye + 4 * dir - s_last) > 0) break;
    }
    if (!FUSED && OUTK == 16 && ovf_acc) atomicOr(ovf, 1);
}

/* Which waves may take the fast path: conditions on the level geometry and, for the fused
 * kernel, on the frame it writes.  The host evaluates the same predicates per launch: when every
 * entry of a launch qualifies it starts the FASTONLY kernels, which do not carry the general
 * path's registers (126 instead of 172 VGPRs for 5/3 rgb24: 4 instead of 2 waves per SIMD). */
__host__ __device__ inline bool stream_fast_geom(const DwtLevel &g)
{
    return !(g.mh & 1) && !(g.lh & 3) && g.lh >= 8 && g.lv >= 2;
}
__host__ __device__ inline bool stream_fast_rgb24(const PackTile &T, const DwtLevel &g, int ncomp_group, int comp0)
{
    const PackComp &C0 = T.c[0];
    const OutPlanes &O = T.out;
    const int pl = C0.out_plane;
    return ncomp_group == 3 && T.ncomp == 3 && T.out_bytes == 1 && C0.pix_step == 3 && comp0 == 0 &&
           T.c[0].pix_off == 0 && T.c[1].pix_off == 1 && T.c[2].pix_off == 2 &&
           T.precision == 8 && T.c[0].cbps <= 8 && T.c[1].cbps <= 8 && T.c[2].cbps <= 8 &&
           C0.out_x >= 0 && !(C0.out_x & 3) && C0.out_y >= 0 &&
           C0.out_x + g.lh <= O.width[pl] && C0.out_y + g.lv <= O.height[pl] &&
           !(O.linesize[pl] & 3) && !(((uintptr_t)O.ptr[pl]) & 3);
}

/* which fast store a component group qualifies for: 0 rgb24, 1 rgb48, 2 one 8-bit plane, 3 one 16-bit plane; -1 none
 * (general path) */
__host__ __device__ inline int stream_fast_outk(const PackTile &T, const DwtLevel &g, int ncomp_group, int comp0)
{
    if (stream_fast_rgb24(T, g, ncomp_group, comp0)) return 0;
    const OutPlanes &O = T.out;
    if (ncomp_group == 3 && T.ncomp == 3 && comp0 == 0 && T.out_bytes == 2) {
        const PackComp &C0 = T.c[0];
        const int pl = C0.out_plane;
        if (C0.pix_step == 3 && T.c[0].pix_off == 0 && T.c[1].pix_off == 1 && T.c[2].pix_off == 2 &&
            T.precision <= 16 && T.c[0].cbps <= T.precision && T.c[1].cbps <= T.precision && T.c[2].cbps <= T.precision &&
            C0.out_x >= 0 && !(C0.out_x & 3) && C0.out_y >= 0 && C0.out_x + g.lh <= O.width[pl] && C0.out_y + g.lv <= O.height[pl] &&
            !(O.linesize[pl] & 3) && !(((uintptr_t)O.ptr[pl]) & 3))
            return 1;
    }
    if (ncomp_group == 1 && T.out_bytes == 1) {
        const PackComp &C = T.c[comp0];
        const int pl = C.out_plane;
        if (C.pix_step == 1 && C.pix_off == 0 && T.precision == 8 && C.cbps <= 8 &&
            C.out_x >= 0 && !(C.out_x & 3) && C.out_y >= 0 && C.out_x + g.lh <= O.width[pl] && C.out_y + g.lv <= O.height[pl] &&
            !(O.linesize[pl] & 3) && !(((uintptr_t)O.ptr[pl]) & 3))
            return 2;
    }
    if (ncomp_group == 1 && T.out_bytes == 2) {
        const PackComp &C = T.c[comp0];
        const int pl = C.out_plane;
        if (C.pix_step == 1 && C.pix_off == 0 && T.precision <= 16 && C.cbps <= T.precision &&
            C.out_x >= 0 && !(C.out_x & 3) && C.out_y >= 0 && C.out_x + g.lh <= O.width[pl] && C.out_y + g.lv <= O.height[pl] &&
            !(O.linesize[pl] & 3) && !(((uintptr_t)O.ptr[pl]) & 3))
            return 3;
    }
    return -1;
}

template <int TYPE, int NC, bool FUSED, bool FASTONLY, bool C16 = false, bool LL16 = false, int OUTK = 0, bool PK = false>
__device__ __forceinline__ void
idwt_stream_body(const DwtTileArgs (&A)[NC], const uint32_t *__restrict__ ll_base, const uint32_t *__restrict__ band_base,
                 uint32_t *__restrict__ out_base, const PackTile *__restrict__ T, int comp0, int th, int tw, int bx, int by)
{
    static_assert(!C16 || FASTONLY, "16-bit sub-bands: FASTONLY launches only");
    if (FASTONLY) {
        idwt_stream_impl<TYPE, NC, FUSED, true, C16, LL16, OUTK, PK>(A, ll_base, band_base, out_base, T, comp0, th, tw, bx, by);
    } else {                                               /* per-wave choice; all conditions are wave-uniform */
        bool fast = stream_fast_geom(A[0].g);
        if (FUSED) fast = fast && stream_fast_rgb24(*T, A[0].g, NC, comp0);
        if (fast) idwt_stream_impl<TYPE, NC, FUSED, true>(A, ll_base, band_base, out_base, T, comp0, th, tw, bx, by);
        else idwt_stream_impl<TYPE, NC, FUSED, false>(A, ll_base, band_base, out_base, T, comp0, th, tw, bx, by);
    }
}

/* XCD-aware order of the strips.  The launch is a 1-D grid of 8 * per_xcd workgroups, which the
 * hardware deals out to the 8 XCDs round-robin (workgroup b -> XCD b % 8).  Strip number
 * (b % 8) * per_xcd + b / 8 then gives every XCD one contiguous run of strips in (x fastest, y,
 * table entry) order: column strips that share cache lines at their edges (a wave loads 512 bytes
 * per sub-band row of which 488 are its own, and 244-sample strips are not line-aligned) and row
 * strips that share HALO rows meet in the same L2 instead of fetching those lines once per XCD. */
struct StreamGrid { int gx, gy, nstrips, per_xcd, tw; };
__device__ __forceinline__ bool stream_strip(const StreamGrid &G, int &bx, int &by, int &bz)
{
    /* workgroups of several waves (blockDim.x / 64): the waves of one take neighbouring strips -- G.per_xcd is a multiple
     * of that count -- and so walk down neighbouring column ranges of the same rows together, on one CU */
    const int wpb = (int)blockDim.x >> 6, b = blockIdx.x;
    const int s = (b & 7) * G.per_xcd + (b >> 3) * wpb + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);   /* (wave-uniform: keep it scalar) */
    if (s >= G.nstrips) return false;
    bx = s % G.gx;
    const int t = s / G.gx;
    by = t % G.gy;
    bz = t / G.gy;
    return true;
}

/* plain level: one DwtTileArgs table entry per plane, blockDim = one wave */
/* FASTONLY kernels may be launched with up to eight waves per workgroup (stream_strip) */
template <int TYPE, bool FASTONLY, bool C16 = false, bool LL16 = false>
__global__ void __launch_bounds__(FASTONLY ? 512 : 64)
k_idwt_stream(const DwtTileArgs *__restrict__ args, const uint32_t *__restrict__ ll_base,
              const uint32_t *__restrict__ band_base, uint32_t *__restrict__ out_base, int th, StreamGrid G)
{
    int bx, by, bz;
    if (!stream_strip(G, bx, by, bz)) return;
    const DwtTileArgs A[1] = { args[bz] };
    idwt_stream_body<TYPE, 1, false, FASTONLY, C16, LL16>(A, ll_base, band_base, out_base, nullptr, 0, th, G.tw, bx, by);
}

/* plain 5/3 level of a job with 16-bit sub-bands whose LL bands are 16-bit as well, in and out */
template <bool PK>
__global__ void __launch_bounds__(512)
k_idwt_stream_ll16(const DwtTileArgs *__restrict__ args, const uint32_t *__restrict__ ll_base,
                   const uint32_t *__restrict__ band_base, uint32_t *__restrict__ out_base, int th, StreamGrid G,
                   int *__restrict__ ovf, int ovf_bits)
{
    int bx, by, bz;
    if (!stream_strip(G, bx, by, bz)) return;
    const DwtTileArgs A[1] = { args[bz] };
    idwt_stream_impl<J2K_DWT53, 1, false, true, true, true, 16, PK, true>(A, ll_base, band_base, out_base, nullptr, 0, th, G.tw, bx, by, ovf, ovf_bits);
}

/* ================================================================== three levels in one launch
 * The first three levels of a 4K plane are 240 x 135, 480 x 270 and 960 x 540 samples: launches of a few thousand waves that
 * end before the chip is full (1.3 / 2.9 / 4.0 TB/s where the big levels reach 4.8), and each writes an LL band only for the
 * next to read it back.  Here one workgroup (4 waves) reconstructs a band of TH rows of the THIRD level's output of one plane:
 * the rows of the first level it needs (through two levels of lifting: TH / 4 + a few) into LDS, from those the rows of the
 * second level (TH / 2 + a few) into LDS, from those its own rows to memory.  The two intermediate LL bands never leave the
 * CU; neighbouring workgroups recompute the few rows they share.  Every level is idwt_stream_impl as it stands -- the same
 * code, the same arithmetic, the same 16-bit range check (`ovf`) on what a level stores -- with its LL input and / or its
 * output pointing into LDS (generic pointers: the loads and stores become FLAT ones).  For jobs with 16-bit sub-bands and LL
 * bands whose planes all start at the origin; everything else runs the levels one launch each. */
struct X3Rows { int lo0, n0, lo1, n1; };          /* first-level rows [lo0, lo0 + n0) and second-level rows [lo1, lo1 + n1) a band needs */
__host__ __device__ inline X3Rows x3_rows(int r0, int r1, int lv0, int lv1)
{
    /* a strip of rows [a, b) of a level reads its LL input at the even absolute rows from (a - 2) & ~1 to (b + 1) & ~1,
     * reflected into the level: LL rows (a - 2) >> 1 .. (b + 1) >> 1, clamped */
    X3Rows R;
    R.lo1 = (r0 - 2) >> 1; if (R.lo1 < 0) R.lo1 = 0;
    int hi1 = (r1 + 1) >> 1; if (hi1 > lv1 - 1) hi1 = lv1 - 1;
    R.n1 = hi1 - R.lo1 + 1;
    R.lo0 = (R.lo1 - 2) >> 1; if (R.lo0 < 0) R.lo0 = 0;
    int hi0 = (hi1 + 2) >> 1; if (hi0 > lv0 - 1) hi0 = lv0 - 1;
    R.n0 = hi0 - R.lo0 + 1;
    return R;
}
/* rows of LDS the two windows take for bands of `th` third-level rows */
__host__ __device__ inline int x3_win0_rows(int th) { return th / 4 + 6; }
__host__ __device__ inline int x3_win1_rows(int th) { return th / 2 + 5; }

template <bool PK>
__global__ void __launch_bounds__(256)
k_idwt_stream_ll16_x3(const DwtTileArgs *__restrict__ args0, const DwtTileArgs *__restrict__ args1, const DwtTileArgs *__restrict__ args2,
                      const uint32_t *__restrict__ coef, uint32_t *__restrict__ out_base, int th, int nbands, int nplanes, int max_lh0, int max_lh1,
                      int *__restrict__ ovf, int ovf_bits)
{
    extern __shared__ __align__(16) uint16_t x3_lds[];
    /* as stream_strip(): every XCD gets a contiguous run of (plane, band) pairs, so that bands which share rows share an L2 */
    const int total = nplanes * nbands, per_xcd = (total + 7) / 8;
    const int sidx = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (sidx >= total) return;                               /* (whole workgroup) */
    const int plane = sidx / nbands, band = sidx % nbands;
    const int wv = threadIdx.x >> 6;
    DwtTileArgs A0[1] = { args0[plane] }, A1[1] = { args1[plane] }, A2[1] = { args2[plane] };
    const int r0 = band * th, r1 = min(r0 + th, A2[0].g.lv);
    if (r0 >= A2[0].g.lv) return;                           /* (whole workgroup) */
    const X3Rows R = x3_rows(r0, r1, A0[0].g.lv, A1[0].g.lv);
    uint16_t *w0 = x3_lds, *w1 = x3_lds + (size_t)x3_win0_rows(th) * max_lh0;
    /* the windows as planes whose first row is row lo0 / lo1 of their band: offsets 0, stride = the level's width */
    uint32_t *out0 = (uint32_t *)w0, *out1 = (uint32_t *)w1;
    A0[0].out_off = 0; A0[0].out_stride = A0[0].g.lh;
    A1[0].ll_off = 0;  A1[0].ll_stride = A0[0].g.lh;
    A1[0].out_off = 0; A1[0].out_stride = A1[0].g.lh;
    A2[0].ll_off = 0;  A2[0].ll_stride = A1[0].g.lh;
    constexpr int TW = STREAM_TW;
    auto level = [&](const DwtTileArgs (&A)[1], const uint32_t *ll, uint32_t *out, int lo, int n, int ll_row0, int out_row0) {
        constexpr bool LL16 = true;                          /* every LL input here is 16-bit: the coefficient buffer's, or a window */
        const int ncol = (A[0].g.lh + TW - 1) / TW;
        int nrow = max(1, 4 / ncol);                         /* row parts, so that the four waves all have a strip */
        int part = ((n + nrow - 1) / nrow + 1) & ~1;         /* an even number of rows each */
        if (part < 2) part = 2;
        nrow = (n + part - 1) / part;
        for (int i = wv; i < ncol * nrow; i += 4) {
            const int bx = i % ncol, pr = i / ncol;
            const int y0 = lo + pr * part, rows = min(part, lo + n - y0);
            idwt_stream_impl<J2K_DWT53, 1, false, true, true, LL16, 16, PK>(A, ll, coef, out, nullptr, 0, rows, TW, bx, 0, ovf, ovf_bits, y0, 1, ll_row0, out_row0);
        }
    };
    level(A0, coef, out0, R.lo0, R.n0, 0, R.lo0);
    __syncthreads();
    level(A1, (const uint32_t *)out0, out1, R.lo1, R.n1, R.lo0, R.lo1);
    __syncthreads();
    level(A2, (const uint32_t *)out1, out_base, r0, r1 - r0, R.lo1, 0);
}

/* final level + inverse MCT + frame store: one DwtFusedArgs table entry per component group */
template <int TYPE, int NC, bool FASTONLY, bool C16 = false, bool LL16 = false, int OUTK = 0, bool PK = false>
__global__ void __launch_bounds__(FASTONLY && TYPE != J2K_DWT97_INT ? 512 : 64)       /* (9/7 fixed point, three components: 232-280 VGPRs) */
k_idwt_stream_pack(const DwtFusedArgs *__restrict__ args, const uint32_t *__restrict__ ll_base,
                   const uint32_t *__restrict__ band_base, const PackTile *__restrict__ tiles, int th, StreamGrid G)
{
    int bx, by, bz;
    if (!stream_strip(G, bx, by, bz)) return;
    const DwtFusedArgs &F = args[bz];
    DwtTileArgs A[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) A[c] = F.a[c];
    idwt_stream_body<TYPE, NC, true, FASTONLY, C16, LL16, OUTK, PK>(A, ll_base, band_base, nullptr, tiles + F.pack_tile, F.comp0, th, G.tw, bx, by);
}

}  // namespace htj2k

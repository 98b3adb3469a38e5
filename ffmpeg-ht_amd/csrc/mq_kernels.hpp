/*
 * mq_kernels.hpp -- Part-1 (EBCOT / MQ) block decoder on the device: decode_cblk() + dequantisation for the
 * codeblocks tile_codeblocks() does not hand to the HT decoder (libavcodec/jpeg2000dec.c:2264-2273).
 *
 *   reference                                         here
 *   MQ decoder, mqcdec.c:30-111, mqc.c:32-79          MqLane::step / bytein / init, mq_rows[]
 *   context labels, jpeg2000.c:91-170                 mq_sig_label / mq_sgn_label -> LDS look-up tables
 *   decode_sigpass/refpass/clnpass, jpeg2000dec.c:1872-1991   the three stripe loops of k_mq_decode
 *   decode_cblk, jpeg2000dec.c:1993-2089              the pass loop, terminations, ROI shift
 *   dequantization_*, jpeg2000dec.c:2098-2181         ht_dequant() at the final store
 *
 * The MQ decoder is one serial chain per codeblock, so the parallelism is across blocks: ONE LANE PER BLOCK,
 * 64 blocks per wave, all lanes walking the stripe-oriented scan in lockstep (the host groups blocks of equal
 * size).  What makes that affordable:
 *   - no per-sample flag words: significance, sign, "visited" and "refined" live as 64-bit ROW MASKS (block
 *     chunk of <= 64 columns), six rows of a stripe in registers; the eight neighbours of a sample are three 3-bit fields
 *     of three row masks, and the column position is wave-uniform (scalar shifts);
 *   - no per-sample magnitudes either: every bit-plane k leaves one row mask per row ("bit k of the samples of
 *     this row"); the samples are assembled once, at the end, together with the half-bit the reference keeps
 *     below the last coded plane, then ROI-shifted, dequantised and stored;
 *   - the row masks of the 64 blocks are interleaved by lane in a scratch buffer, so every load/store of a row
 *     is one 512-byte coalesced access;
 *   - each lane's code bytes come from a 128-byte LDS window refilled for the whole wave when the furthest lane
 *     gets within 40 bytes of its end (a decision consumes at most 3 bytes, a column at most 11 decisions).
 * Rows wider than 64 columns (legal: up to 1024 x 4) are walked in 64-column chunks; the two columns next to a chunk
 * come along as one border bit per row.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "j2k_plan.h"
#include "ht_kernels.hpp"

namespace htj2k {

struct MqWave {                 /* one wave = up to 64 consecutive blocks of the Part-1 table */
    uint32_t soff;              /* scratch offset in 512-byte units (one row mask of 64 lanes) */
    uint16_t hmax, wmax;        /* largest block of the wave */
    uint16_t pmax;              /* most coding passes */
    uint16_t rows;              /* row slots per plane: round4(hmax) + 2 */
    uint16_t chunks;            /* 64-column chunks per row: ceil(wmax / 64) */
    uint16_t pad;
};

#define MQ_WIN_BYTES   96
#define MQ_WIN_PITCH   25       /* dwords per lane */
#define MQ_WIN_MARGIN  40
#define MQ_CX_UNI 17
#define MQ_CX_RL  18
/* LDS: [code-byte windows + context states | aliased by the plane rows of the final assembly] + look-up tables.
 * mq_lds_area(planes) is the size of the first, shared part: what limits the waves per CU */
#define MQ_LDS_TABLES (94 * 4 + 256 * 4 + 256)
__host__ __device__ inline uint32_t mq_lds_area(uint32_t nplanes)
{
    const uint32_t decode = 64 * MQ_WIN_PITCH * 4 + 19 * 64, assemble = nplanes * 64 * 8;
    return ((decode > assemble ? decode : assemble) + 15u) & ~15u;
}

/* T.800 Table C.2: Qe, next state after an MPS, after an LPS, MPS switch */
__device__ static const uint16_t mq_rows[47][4] = {
    { 0x5601,  1,  1, 1 }, { 0x3401,  2,  6, 0 }, { 0x1801,  3,  9, 0 }, { 0x0ac1,  4, 12, 0 }, { 0x0521,  5, 29, 0 },
    { 0x0221, 38, 33, 0 }, { 0x5601,  7,  6, 1 }, { 0x5401,  8, 14, 0 }, { 0x4801,  9, 14, 0 }, { 0x3801, 10, 14, 0 },
    { 0x3001, 11, 17, 0 }, { 0x2401, 12, 18, 0 }, { 0x1c01, 13, 20, 0 }, { 0x1601, 29, 21, 0 }, { 0x5601, 15, 14, 1 },
    { 0x5401, 16, 14, 0 }, { 0x5101, 17, 15, 0 }, { 0x4801, 18, 16, 0 }, { 0x3801, 19, 17, 0 }, { 0x3401, 20, 18, 0 },
    { 0x3001, 21, 19, 0 }, { 0x2801, 22, 19, 0 }, { 0x2401, 23, 20, 0 }, { 0x2201, 24, 21, 0 }, { 0x1c01, 25, 22, 0 },
    { 0x1801, 26, 23, 0 }, { 0x1601, 27, 24, 0 }, { 0x1401, 28, 25, 0 }, { 0x1201, 29, 26, 0 }, { 0x1101, 30, 27, 0 },
    { 0x0ac1, 31, 28, 0 }, { 0x09c1, 32, 29, 0 }, { 0x08a1, 33, 30, 0 }, { 0x0521, 34, 31, 0 }, { 0x0441, 35, 32, 0 },
    { 0x02a1, 36, 33, 0 }, { 0x0221, 37, 34, 0 }, { 0x0141, 38, 35, 0 }, { 0x0111, 39, 36, 0 }, { 0x0085, 40, 37, 0 },
    { 0x0049, 41, 38, 0 }, { 0x0025, 42, 39, 0 }, { 0x0015, 43, 40, 0 }, { 0x0009, 44, 41, 0 }, { 0x0005, 45, 42, 0 },
    { 0x0001, 45, 43, 0 }, { 0x5601, 46, 46, 0 },
};

/* significance context from the counts of significant horizontal / vertical / diagonal neighbours
 * (getsigctxno, jpeg2000.c:91-138; bandpos 0 LL, 1 HL, 2 LH, 3 HH) */
__device__ __forceinline__ uint32_t mq_sig_label(int h, int v, int d, int bandpos)
{
    if (bandpos < 3) {
        if (bandpos == 1) { const int t = h; h = v; v = t; }
        if (h == 2) return 8;
        if (h == 1) return v >= 1 ? 7 : d >= 1 ? 6 : 5;
        if (v == 2) return 4;
        if (v == 1) return 3;
        return d >= 2 ? 2 : d;
    }
    if (d >= 3) return 8;
    if (d == 2) return h + v >= 1 ? 7 : 6;
    if (d == 1) return h + v >= 2 ? 5 : h + v == 1 ? 4 : 3;
    return h + v >= 2 ? 2 : h + v;
}

/* sign context and xor bit (getsgnctxno, jpeg2000.c:140-158): contribution of a neighbour pair is
 * clamp(sum of +1 per positive, -1 per negative significant neighbour) */
__device__ __forceinline__ uint32_t mq_sgn_label(int hc, int vc)
{
    const int lab = hc == 0 ? (vc == 0 ? 9 : 10) : (hc * vc > 0 ? 13 : vc == 0 ? 12 : 11);
    const int x_or = hc < 0 || (hc == 0 && vc < 0);
    return (uint32_t)lab | (uint32_t)(x_or << 7);
}

__device__ __forceinline__ int mq_needs_termination(int style, int passno)   /* jpeg2000.h:302-317 */
{
    if (style & 0x01) {
        const int type = passno % 3;
        passno /= 3;
        if (type == 0 && passno > 2) return 2;
        if (type == 2 && passno > 2) return 1;
        if (style & 0x04) return passno > 2 ? 2 : 1;
    }
    return (style & 0x04) ? 1 : 0;
}

/* three neighbouring bits of a row mask around column x: bit 0 = column x - 1, bit 1 = x, bit 2 = x + 1 */
__device__ __forceinline__ uint32_t mq_g3(uint64_t m, int x)
{
    return (x ? (uint32_t)(m >> (x - 1)) : (uint32_t)m << 1) & 7u;
}
/* the same for one 64-column chunk of a wider row: `lb` / `rb` are the bits of the columns just outside the chunk */
__device__ __forceinline__ uint32_t mq_g3(uint64_t m, int x, uint32_t lb, uint32_t rb)
{
    return ((x ? (uint32_t)(m >> (x - 1)) : ((uint32_t)m << 1 | lb)) & 7u) | (x == 63 ? rb << 2 : 0u);
}

struct MqLane {
    uint32_t a, c;              /* mqc->a, mqc->c */
    uint32_t bp;                /* mqc->bp as an offset into the block's bytes */
    uint32_t cur;               /* *mqc->bp */
    uint32_t wb;                /* first byte of the LDS window */
    bool raw;
};

/*
 * One lane per codeblock.  blocks: the Part-1 table (first block of wave g at waves[g] order: 64 * g).
 */
/* (64, 4): 128 VGPRs, 47 / 70 of them spilled to scratch.  (64, 3) compiles without spills (152 / 168 VGPRs) and is SLOWER:
 * 5.24 against 6.17 Gpixel/s on bench.py --part1 80 (round 3, gpurun_out/r03/bench_p1_lb3.log) -- the lockstep decoder step
 * waits on LDS and scratch either way, and a fourth wave per SIMD hides more of it than the spills cost. */
template <bool WIDE>                                                 /* WIDE: some block of the launch has more than 64 columns */
__global__ void __launch_bounds__(64, 4)
k_mq_decode(const J2kBlock *__restrict__ blocks, int nblocks, const uint8_t *__restrict__ bytes,
            uint32_t *__restrict__ coef, int *__restrict__ status, const MqWave *__restrict__ waves,
            uint64_t *__restrict__ scratch, uint32_t lds_area)
{
    extern __shared__ __align__(16) uint8_t mq_lds[];
    uint32_t *win    = (uint32_t *)mq_lds;                          /* [64][MQ_WIN_PITCH] */
    uint8_t  *cx     = mq_lds + 64 * MQ_WIN_PITCH * 4;              /* [19][64] */
    uint64_t *vrow   = (uint64_t *)mq_lds;                          /* [planes][64]: the planes of one row, final assembly
                                                                     * (the windows and contexts are dead by then) */
    uint32_t *mqtab  = (uint32_t *)(mq_lds + lds_area);             /* [94]: qe | nmps << 16 | nlps << 24 */
    uint32_t *siglut = mqtab + 94;                                  /* [256]: one byte per bandpos */
    uint8_t  *sgnlut = (uint8_t *)(siglut + 256);                   /* [256]: label | xorbit << 7 */

    const int lane = threadIdx.x;
    const MqWave W = waves[blockIdx.x];
    const int hmax = W.hmax, wmax = W.wmax, pmax = W.pmax, R = W.rows, WC = WIDE ? W.chunks : 1;
    uint64_t *S = scratch + ((size_t)W.soff << 6) + lane;           /* chunk c of row slot s of plane p: S[SL(p, s, c)] */
#define SL(p, s, c) ((size_t)(((p) * R + (s)) * WC + (c)) << 6)
    const int bi = blockIdx.x * 64 + lane;
    const bool have = bi < nblocks;

    /* ---- look-up tables ---- */
    for (int i = lane; i < 94; i += 64) {
        const int st = i >> 1, mps = i & 1;
        const uint32_t nm = 2u * mq_rows[st][1] + mps;
        const uint32_t nl = 2u * mq_rows[st][2] + (mq_rows[st][3] ? 1 - mps : mps);
        mqtab[i] = mq_rows[st][0] | nm << 16 | nl << 24;
    }
    for (int i = lane; i < 256; i += 64) {
        /* significance: index = NW N NE (bits 0-2) | SW S SE (3-5) | W (6) | E (7) */
        const int h = ((i >> 6) & 1) + ((i >> 7) & 1), v = ((i >> 1) & 1) + ((i >> 4) & 1);
        const int d = (i & 1) + ((i >> 2) & 1) + ((i >> 3) & 1) + ((i >> 5) & 1);
        siglut[i] = mq_sig_label(h, v, d, 0) | mq_sig_label(h, v, d, 1) << 8 | mq_sig_label(h, v, d, 2) << 16 |
                    mq_sig_label(h, v, d, 3) << 24;
        /* sign: index = significant N S W E (bits 0-3) | negative N S W E (bits 4-7) */
        auto contrib = [&](int k) { return ((i >> k) & 1) ? (((i >> (4 + k)) & 1) ? -1 : 1) : 0; };
        const int vc = max(-1, min(1, contrib(0) + contrib(1))), hc = max(-1, min(1, contrib(2) + contrib(3)));
        sgnlut[i] = (uint8_t)mq_sgn_label(hc, vc);
    }

    /* ---- per-lane block ---- */
    J2kBlock b;
    if (have) b = blocks[bi];
    else { b.w = b.h = 0; b.npasses = 0; b.lcup = 0; b.lref = 0; b.data_off = 0; b.M_b = 1; b.zbp = 0; b.roi_shift = 0; b.flags = 0;
           b.plane_off = 0; b.stride = 0; b.f_step = 0.f; b.i_step = 0; b.tcomp = 0; }
    const int w = b.w, h = b.h, npasses = b.npasses, M_b = b.M_b;
    const uint8_t *data = bytes + b.data_off;
    const J2kPart1Trailer *tr = (const J2kPart1Trailer *)(data + J2K_P1_TRAILER_OFF(b.lcup));
    const uint16_t *starts = (const uint16_t *)((const uint8_t *)tr + 4);   /* cblk->data_start[1 ..] */
    int style = 0, bandpos = 0, nterm = 0;
    if (have) { style = tr->style; bandpos = tr->bandpos; nterm = tr->nterm; }
    const bool vsc = (style & 0x08) != 0;
    const int bpno0 = (int)b.zbp - 1 + 31 - M_b - 1 - (int)b.roi_shift;     /* decode_cblk, jpeg2000dec.c:1997 */
    bool alive = have && npasses > 0;
    int err = 0, nexec = 0, term_cnt = 0;

    /* zero the state planes: significance, sign, visited, refined */
    for (int s = 0; s < 4 * R * WC; s++) S[(size_t)s << 6] = 0;

    MqLane m;
    m.a = 0x8000; m.c = 0; m.bp = 0; m.cur = 0; m.wb = 0; m.raw = false;

    auto refill = [&]() {                                            /* window <- bytes [bp & ~3, +128) of every lane */
        m.wb = m.bp & ~3u;
        const uint32_t *src = (const uint32_t *)(data + m.wb);
#pragma unroll
        for (int q = 0; q < MQ_WIN_BYTES / 16; q++) {
            uint4 t;
            __builtin_memcpy(&t, src + 4 * q, 16);
            win[lane * MQ_WIN_PITCH + 4 * q + 0] = t.x; win[lane * MQ_WIN_PITCH + 4 * q + 1] = t.y;
            win[lane * MQ_WIN_PITCH + 4 * q + 2] = t.z; win[lane * MQ_WIN_PITCH + 4 * q + 3] = t.w;
        }
    };
    auto byte_at = [&](uint32_t off) -> uint32_t {                   /* off within [wb, wb + 128) */
        return ((const uint8_t *)win)[lane * (MQ_WIN_PITCH * 4) + (off - m.wb)];
    };
    auto bytein = [&]() {                                            /* mqcdec.c:30-43 */
        const uint32_t b1 = byte_at(m.bp + 1);
        const bool ff = m.cur == 0xff, marker = ff && b1 > 0x8f;
        m.c += marker ? 1u : ff ? 0xfe02u - (b1 << 9) : 0xff01u - (b1 << 8);
        if (!marker) { m.bp++; m.cur = b1; }
    };
    auto mq_init = [&](bool raw) {                                   /* ff_mqc_initdec, mqcdec.c:73-83; window at bp */
        m.raw = raw;
        m.cur = byte_at(m.bp);
        m.c = (m.cur ^ 0xffu) << 16;
        bytein();
        m.c <<= 7;
        m.a = 0x8000;
    };
    auto reset_contexts = [&]() {                                    /* ff_mqc_init_contexts, mqc.c:73-79 */
        for (int k = 0; k < 19; k++) cx[k * 64 + lane] = k == 0 ? 8 : k == MQ_CX_UNI ? 92 : k == MQ_CX_RL ? 6 : 0;
    };
    /* one decision for the lanes with `pred` (ff_mqc_decode, mqcdec.c:94-111; exchange :45-71) */
    auto step = [&](bool pred, uint32_t ctx) -> uint32_t {
        uint32_t d = 0;
        if (pred) {
            if (m.raw) {                                             /* mqc_decode_bypass, :85-92 */
                d = (m.c & 0x40000000u) ? 0u : 1u;
                if (!(m.c & 0xff)) { m.c -= 0x100; bytein(); }
                m.c += m.c;
            } else {
                const uint32_t st = cx[ctx * 64 + lane];
                const uint32_t t = mqtab[st], qe = t & 0xffffu;
                m.a -= qe;
                const bool lps = (m.c >> 16) >= m.a;
                if (!lps && (m.a & 0x8000)) {
                    d = st & 1;
                } else {
                    const bool small = m.a < qe;
                    if (lps) { m.c -= m.a << 16; m.a = qe; }
                    const bool sw = lps ? !small : small;
                    d = (st & 1) ^ (sw ? 1u : 0u);
                    cx[ctx * 64 + lane] = (uint8_t)(sw ? t >> 24 : (t >> 16) & 0xff);
                    int n = __clz((int)m.a) - 16;                    /* RENORMD: shifts until bit 15 of a is set */
                    do {
                        if (!(m.c & 0xff)) { m.c -= 0x100; bytein(); }
                        const int avail = 8 - (__ffs((int)(m.c & 0xff)) - 1);   /* shifts before the next byte is due */
                        const int s = min(n, avail);
                        m.a <<= s; m.c <<= s; n -= s;
                    } while (n);
                }
            }
        }
        return d;
    };

    __syncthreads();
    reset_contexts();
    refill();
    if (alive) {
        if (b.lcup == 0) alive = false; else mq_init(false);
    }

    for (int i = 0; i < pmax; i++) {
        const int type = (i + 2) % 3;                                /* 2 cleanup, 0 significance, 1 refinement */
        const int k = (i + 2) / 3;                                   /* bit-plane index below the block's first */
        bool act = alive && i < npasses;
        if (act) {
            const int bpno = bpno0 - k;
            if (bpno < 0 || bpno > 29) { alive = false; act = false; err = 1; }      /* "bpno became invalid" */
        }
        if (!__ballot(act)) break;
        if (act) nexec = i + 1;
        const int PV = 4 + k;                                        /* this bit-plane's value rows */

        for (int y0 = 0; y0 < hmax; y0 += 4)
        for (int ch = 0; ch < WC; ch++) {                            /* rows wider than 64 columns: chunk by chunk */
            if (__ballot(act && m.bp + 1 + MQ_WIN_MARGIN > m.wb + MQ_WIN_BYTES)) refill();
            uint64_t sg[6], sn[6], vis[4], ref[4], val[4];
#pragma unroll
            for (int q = 0; q < 6; q++) { sg[q] = S[SL(0, y0 + q, ch)]; sn[q] = S[SL(1, y0 + q, ch)]; }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                vis[q] = (type != 0 && k > 0) ? S[SL(2, y0 + 1 + q, ch)] : 0;
                ref[q] = type == 1 ? S[SL(3, y0 + 1 + q, ch)] : 0;
                val[q] = (type == 1 || (type == 2 && k > 0)) ? S[SL(PV, y0 + 1 + q, ch)] : 0;
            }
            /* significance and sign of the columns next to the chunk, one bit per row (bit q = row y0 - 1 + q): the
             * chunk on the left has been through this pass already, the one on the right has not */
            uint32_t lbg = 0, rbg = 0, lbs = 0, rbs = 0;
            if (WIDE && WC > 1) {
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    if (ch > 0)      { lbg |= (uint32_t)(S[SL(0, y0 + q, ch - 1)] >> 63) << q; lbs |= (uint32_t)(S[SL(1, y0 + q, ch - 1)] >> 63) << q; }
                    if (ch + 1 < WC) { rbg |= ((uint32_t)S[SL(0, y0 + q, ch + 1)] & 1u) << q; rbs |= ((uint32_t)S[SL(1, y0 + q, ch + 1)] & 1u) << q; }
                }
            }
            const bool full = y0 + 3 < h;
            const int xw = min(64, wmax - 64 * ch);

            for (int x = 0; x < xw; x++) {
                if (__ballot(act && m.bp + 1 + MQ_WIN_MARGIN > m.wb + MQ_WIN_BYTES)) refill();
                const uint64_t bit = 1ull << x;
                const bool inx = act && 64 * ch + x < w;
                auto field = [&](int r, uint32_t &n3, uint32_t &c3, uint32_t &s3) {
                    if (WIDE) {
                        n3 = mq_g3(sg[r], x, (lbg >> r) & 1, (rbg >> r) & 1);
                        c3 = mq_g3(sg[r + 1], x, (lbg >> (r + 1)) & 1, (rbg >> (r + 1)) & 1);
                        s3 = (vsc && r == 3) ? 0u : mq_g3(sg[r + 2], x, (lbg >> (r + 2)) & 1, (rbg >> (r + 2)) & 1);
                    } else {
                        n3 = mq_g3(sg[r], x); c3 = mq_g3(sg[r + 1], x);
                        s3 = (vsc && r == 3) ? 0u : mq_g3(sg[r + 2], x);
                    }
                };
                /* decode the sign of (x, r) and make it significant (set_significance, jpeg2000.c:172-195) */
                auto sign_and_set = [&](int r, bool on, bool always_xor) {
                    if (!__ballot(on)) return;
                    uint32_t n3, c3, s3;
                    field(r, n3, c3, s3);
                    const uint32_t sc = WIDE ? mq_g3(sn[r + 1], x, (lbs >> (r + 1)) & 1, (rbs >> (r + 1)) & 1) : mq_g3(sn[r + 1], x);
                    const uint32_t sN = (n3 >> 1) & 1, sS = (s3 >> 1) & 1, sW = c3 & 1, sE = (c3 >> 2) & 1;
                    const uint32_t J = sN | sS << 1 | sW << 2 | sE << 3 |
                                       (sN & (uint32_t)(sn[r] >> x)) << 4 | (sS & (uint32_t)(sn[r + 2] >> x) & 1) << 5 |
                                       (sW & sc) << 6 | (sE & (sc >> 2)) << 7;
                    const uint32_t t = sgnlut[J & 0xff];
                    uint32_t sbit = step(on, t & 0x1f);
                    if (always_xor || !m.raw) sbit ^= t >> 7;
                    if (on) {
                        sg[r + 1] |= bit;
                        if (sbit) sn[r + 1] |= bit;
                        val[r] |= bit;
                    }
                };

                if (type == 0) {                                     /* decode_sigpass, jpeg2000dec.c:1872-1905 */
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        uint32_t n3, c3, s3;
                        field(r, n3, c3, s3);
                        const bool cond = inx && y0 + r < h && !(c3 & 2) && (n3 | s3 | (c3 & 5));
                        if (!__ballot(cond)) continue;
                        const uint32_t I = n3 | s3 << 3 | (c3 & 1) << 6 | (c3 >> 2) << 7;
                        const uint32_t d = step(cond, (siglut[I] >> (8 * bandpos)) & 0xff);
                        sign_and_set(r, cond && d, false);
                        if (cond) vis[r] |= bit;
                    }
                } else if (type == 1) {                              /* decode_refpass, :1907-1932 */
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const bool cond = inx && y0 + r < h && (sg[r + 1] & bit) && !(vis[r] & bit);
                        if (!__ballot(cond)) continue;
                        uint32_t n3, c3, s3;
                        field(r, n3, c3, s3);
                        const uint32_t ctx = (ref[r] & bit) ? 16u : (n3 | s3 | (c3 & 5)) ? 15u : 14u;
                        const uint32_t d = step(cond, ctx);
                        if (cond) { ref[r] |= bit; if (d) val[r] |= bit; }
                    }
                } else {                                             /* decode_clnpass, :1934-1991 */
                    bool quiet = inx && full;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        uint32_t n3, c3, s3;
                        field(r, n3, c3, s3);
                        if ((n3 | c3 | s3) || (vis[r] & bit)) quiet = false;
                    }
                    int runlen = 0;
                    bool skip = false;
                    if (__ballot(quiet)) {
                        const uint32_t d = step(quiet, MQ_CX_RL);
                        skip = quiet && !d;
                        const bool go = quiet && d;
                        if (__ballot(go)) {
                            const uint32_t u1 = step(go, MQ_CX_UNI);
                            const uint32_t u0 = step(go, MQ_CX_UNI);
                            if (go) runlen = (int)(u1 << 1 | u0);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const bool on = inx && !skip && y0 + r < h && r >= runlen;
                        const bool first = on && quiet && r == runlen;
                        const bool code = on && !first && !(sg[r + 1] & bit) && !(vis[r] & bit);
                        uint32_t d = 0;
                        if (__ballot(code)) {
                            uint32_t n3, c3, s3;
                            field(r, n3, c3, s3);
                            const uint32_t I = n3 | s3 << 3 | (c3 & 1) << 6 | (c3 >> 2) << 7;
                            d = step(code, (siglut[I] >> (8 * bandpos)) & 0xff);
                        }
                        sign_and_set(r, first || (code && d), true);
                    }
                }
            }
            /* rows of this stripe back to the planes */
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (type != 1) { S[SL(0, y0 + 1 + q, ch)] = sg[q + 1]; S[SL(1, y0 + 1 + q, ch)] = sn[q + 1]; }
                if (type == 0) S[SL(2, y0 + 1 + q, ch)] = vis[q];
                if (type == 1) S[SL(3, y0 + 1 + q, ch)] = ref[q];
                S[SL(PV, y0 + 1 + q, ch)] = val[q];
            }
        }
        if (type == 2 && __ballot(act && (style & 0x20))) {          /* segmentation symbol: read, not checked (:1980-1990) */
            const bool on = act && (style & 0x20);
            if (__ballot(on && m.bp + 1 + MQ_WIN_MARGIN > m.wb + MQ_WIN_BYTES)) refill();
            for (int q = 0; q < 4; q++) step(on, MQ_CX_UNI);
        }
        if (act && (style & 0x02)) reset_contexts();                 /* JPEG2000_CBLK_RESET, :2036-2037 */
        /* a terminated segment ends here: restart the decoder on the next one (:2039-2053) */
        bool restart = false;
        int coder = 0;
        if (act && i + 1 < npasses && (coder = mq_needs_termination(style, i)) != 0) {
            if (term_cnt >= nterm) { alive = false; err = 1; }       /* "Missing needed termination" */
            else { term_cnt++; m.bp = starts[term_cnt - 1]; restart = true; }
        }
        if (__ballot(restart)) {
            refill();
            if (restart) mq_init(coder == 2);
        }
    }

    /* ---- assemble, ROI shift, dequantise, store (jpeg2000dec.c:2071-2086, 2098-2181) ---- */
    const int klast = nexec > 0 ? (nexec + 1) / 3 : -1;              /* last bit-plane index touched */
    const bool refran = klast > 0 && nexec >= 3 * klast;             /* its refinement pass ran: every significant sample was coded there */
    const int nplanes = min((pmax + 1) / 3 + 1, 32);                 /* bpno stays within 0..29: at most 30 planes are ever coded */
    float fscale = b.f_step;
    fscale /= (float)(1 << (31 - M_b));
    const int transform = b.flags & 3, roi_shift = err ? 0 : b.roi_shift;
    uint32_t *dst = coef + b.plane_off;
    __syncthreads();                                                 /* vrow takes over the window / context area */
    for (int y = 0; y < hmax; y++)
    for (int ch = 0; ch < WC; ch++) {
        const uint64_t sgr = S[SL(0, y + 1, ch)], snr = S[SL(1, y + 1, ch)];
        __builtin_amdgcn_wave_barrier();
        for (int k = 0; k < nplanes; k++)
            vrow[k * 64 + lane] = k <= klast ? S[SL(4 + k, y + 1, ch)] : 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint64_t vl = klast >= 0 ? vrow[klast * 64 + lane] : 0;
        for (int x = 0; x < min(64, wmax - 64 * ch); x++) {
            uint32_t mag = 0;
            for (int k = 0; k < nplanes; k++) {
                const uint32_t bitk = (uint32_t)(vrow[k * 64 + lane] >> x) & 1u;
                mag |= bitk << ((bpno0 - k + 1) & 31);
            }
            if ((sgr >> x) & 1) {
                const int kl = refran || ((vl >> x) & 1) ? klast : klast - 1;
                mag |= 1u << ((bpno0 - kl) & 31);
            }
            const uint32_t smag = (mag & 0x7FFFFFFFu) | ((uint32_t)(snr >> x) & 1u) << 31;
            if (have && 64 * ch + x < w && y < h && npasses > 0)
                dst[(size_t)y * b.stride + 64 * ch + x] = ht_dequant(smag, transform, M_b, roi_shift, fscale, b.i_step);
        }
    }
    if (have && err) status[bi] = 1;
#undef SL
}

}  // namespace htj2k

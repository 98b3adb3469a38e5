/*
 * ht_kernels.hpp -- HT block decoder for gfx950.
 *
 * What it computes is ff_jpeg2000_decode_htj2k() (libavcodec/jpeg2000htdec.c:1188-1336)
 * followed by dequantization_int / _float / _int_97 (libavcodec/jpeg2000dec.c:2098-2181)
 * written straight into the tile-component plane at the block's Mallat position
 * (jpeg2000dec.c:2279-2287).  How it computes it is not the reference's byte-at-a-time
 * bit buffers.  The work of a block splits by its dependence structure:
 *
 *   un-stuffing (16, 32 or 64 lanes per block)   k_ht_unstuff_g / k_ht_unstuff (VLC, SigProp, MagRef bytes) and the
 *                       head of the MagSgn kernels, eight bytes per lane: the bytes that carry 7 bits are
 *                       flagged by SWAR tests (after a 0xFF, jpeg2000htdec.c:207-221; a 0x7F-low byte below a
 *                       >0x8F byte for the backward streams, :145-201), their spare bits squeezed out, then a
 *                       prefix sum over the block's lanes -> bit offset of the lane's chunk, ds_or into
 *                       32-bit words.
 *   serial chains (one LANE per block, 64 blocks per wave)
 *                       k_ht_vlc2 (k_ht_vlc for blocks wider than 64 columns): MEL + CxtVLC + U-VLC of every quad (:632-973) -- "codeword
 *                       length -> next codeword position -> context" -- and k_ht_refine:
 *                       SigProp / MagRef (:1016-1185) -- "which sample takes the next bit depends
 *                       on the significance the previous bits made".  Neither parallelises
 *                       inside a block, both do across blocks.
 *   MagSgn (64 lanes per block)   k_ht_decode: lanes = sample columns of a quad row: exponent
 *                       predictor kappa from the row above (:855-885), m_n, wave prefix sum ->
 *                       MagSgn bit offsets, extraction from the LDS bit array, mu/E (:395-427),
 *                       refinement bits, dequantisation and two coalesced row stores.
 *                       k_ht_decode_pair: the same for two blocks per wave with one lane per quad
 *                       (the quad-level work is not done twice), for jobs whose sub-bands are
 *                       stored as 16-bit samples (htj2k_device.hip: coef16).
 *                       k_ht_decode_multi: a lane per quad and two or four blocks per wave for
 *                       jobs with 32-bit sub-bands (all three dequantisers, refinement masks).
 *
 * k_ht_decode<false> is the first correct version (everything in one kernel, the serial stages on
 * lane 0), kept as a fallback and A/B reference; its LDS layout is HtLds.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "j2k_plan.h"

namespace htj2k {

struct HtLds {
    uint32_t off_ms, ms_words;        /* un-stuffed MagSgn bits */
    uint32_t off_vlc, vlc_words;      /* un-stuffed VLC bits (in read order) */
    uint32_t off_suf, suf_bytes;      /* raw MEL+VLC suffix bytes (MEL reads them serially) */
    uint32_t off_qinfo, max_qw;       /* 2 x max_qw packed quad symbols (current / previous row) */
    uint32_t off_E;                   /* 2 x (2*max_qw + 8) exponent bytes */
    uint32_t off_bm, bm_words;        /* 4 bitmaps of bm_words each (sigma, ref, sign, magref), 0 if unused */
    uint32_t total;
};

#define HT_ERR_INVALID 1   /* the reference returns AVERROR_INVALIDDATA; block left zero */
#define HT_REF_STRIDE 64   /* k_ht_refine's masks: 64-bit words between consecutive masks of a block (lane-interleaved per wave) */
#define HT_MEL_SYMS  1344  /* MEL symbols a block can consume: <= 1024 quads + <= 256 first-row pairs, rounded up */
#define HT_MEL_WORDS (HT_MEL_SYMS / 32 + 2)
#define HT_UVLC_ENTRIES (5 * 64)
#define HT_VSTAGE_PITCH 17                  /* dwords per lane: a ring of 16 stream words; odd, so that the lanes' reads fall into distinct banks */
#define HT_VSTAGE_BYTES (64 * HT_VSTAGE_PITCH * 4)

/* quad symbols of a block in d_qsym: rows padded to an even number of quads (k_ht_vlc emits two
 * per pass of its loop), the block rounded up to 16 symbols (its lanes flush 32-byte chunks) */
/* One quad symbol is 16 bits: bits 0-7 four 2-bit fields, field n = rho_n + e_k,n + e_1,n of sample n (e_1 is only
 * ever set where e_k is, e_k where rho is: 0 = not significant, 1 = significant, 2 = + known exponent bound, 3 = + known
 * MSB), bits 8-15 the U-VLC value u (at most 2 + 5 + 31 + 4 * 15).  (Until round 2 a dword: rho | e_k << 4 | e_1 << 8 |
 * u << 16 -- the symbol array was 1.24 GB written and read again per 48 4K frames.) */
typedef uint16_t ht_sym_t;
/* words of the un-stuffed arrays of one block: [0, ht_nsw) VLC resp. MEL, [ht_nsw, ht_nsw + ht_nsp) SigProp
 * resp. MagRef (blocks with refinement passes); both fit the block's byte region (J2K_BLOCK_PAD) */
__host__ __device__ inline uint32_t ht_nsw(uint32_t Scup) { return (Scup * 8 + 31) / 32 + 2; }
__host__ __device__ inline uint32_t ht_nsp(uint32_t Lref) { return (Lref * 8 + 31) / 32 + 3; }

__host__ __device__ inline uint32_t ht_sym_pack_fields(uint32_t rho, uint32_t ek, uint32_t e1)
{
    uint32_t pk = 0;
    for (int n = 0; n < 4; n++) pk |= (((rho >> n) & 1) + ((ek >> n) & 1) + ((e1 >> n) & 1)) << (2 * n);
    return pk;
}
/* symbol -> rho | e_k << 4 | e_1 << 8 | u << 16 (the paths that are not bench-critical keep that form) */
__device__ __forceinline__ uint32_t ht_sym_unpack(uint32_t s)
{
    const uint32_t pk = s & 0xFF, lo = pk & 0x55, hi = (pk >> 1) & 0x55;
    uint32_t t = (lo | hi) | (hi << 8) | ((lo & hi) << 16);             /* >= 1, >= 2, == 3 at the even bits of bytes 0, 1, 2 */
    t = (t | (t >> 1)) & 0x333333u;
    t = (t | (t >> 2)) & 0x0F0F0Fu;
    return (t & 0xF) | ((t >> 4) & 0xF0) | ((t >> 8) & 0xF00) | ((s >> 8) << 16);
}
__host__ __device__ inline uint32_t ht_qsym_pitch(uint32_t w) { return (((w + 1) >> 1) + 1) & ~1u; }
__host__ __device__ inline uint32_t ht_qsym_words(uint32_t w, uint32_t h) { return (ht_qsym_pitch(w) * ((h + 1) >> 1) + 15) & ~15u; }

/* inclusive prefix sum over the 64 lanes with DPP: Hillis-Steele inside each row of 16 lanes
 * (row_shr 1,2,4,8), then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2-3
 * (gfx9 DPP controls; lanes without a source keep `old` = 0) */
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane)
{
    (void)lane;
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return (uint32_t)x;
}
__device__ __forceinline__ uint32_t wave_last(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }

/* same-wave LDS hand-off: the LDS executes a wave's instructions in order, only the compiler
 * must not reorder across this point (a one-wave workgroup needs no s_barrier and, above all,
 * no vmcnt(0) wait for the row stores that __syncthreads() would add) */
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

/* the MagSgn bit array of a block in LDS: `p` is 16-byte aligned and its region a multiple of 16 bytes (the host rounds
 * ms_words up), so the clearing runs 16 bytes per lane; past the end of the stream the array is all ones
 * (jpeg2000htdec.c:207-221) -- only the words from the one that holds bit `total` on need touching */
__device__ __forceinline__ void ht_ms_clear(uint32_t *p, uint32_t nwords, int lane)
{
    for (uint32_t i = 4u * lane; i < nwords; i += 256) *(uint4 *)(p + i) = make_uint4(0u, 0u, 0u, 0u);
}
__device__ __forceinline__ void ht_ms_ones_tail(uint32_t *p, uint32_t total, uint32_t last_word, int lane)
{
    for (uint32_t i = (total >> 5) + lane; i <= last_word; i += 64) {
        if (i * 32 >= total) p[i] = 0xFFFFFFFFu;
        else p[i] |= 0xFFFFFFFFu << (total & 31);
    }
}

/* dequantise one sign-magnitude sample (bit 31 sign, magnitude LSB at 31 - M_b) */
__device__ __forceinline__ uint32_t ht_dequant(uint32_t smag, int transform, int M_b, int roi_shift,
                                               float fscale, int i_step)
{
    uint32_t mag = smag & 0x7FFFFFFFu;
    const bool neg = (smag >> 31) != 0;
    if (roi_shift) {                                   /* jpeg2000htdec.c:1326-1328 */
        const uint32_t mask = 0xFFFFFFFFu >> (M_b + 1);
        if ((mag & ~mask) == 0) mag <<= roi_shift;
    }
    if (transform == J2K_DWT53) {                      /* dequantization_int */
        int v = (int)(mag >> (31 - M_b));
        if (neg) v = -v;
        if (i_step != 32768) {
            long long t = (long long)v * i_step;       /* (val * (int64_t)i_stepsize) / 65536, truncating */
            v = (int)(t < 0 ? -((-t) >> 16) : (t >> 16));
        }
        return (uint32_t)v;
    } else if (transform == J2K_DWT97) {               /* dequantization_float */
        int v = neg ? -(int)mag : (int)mag;
        return __float_as_uint((float)v * fscale);
    } else {                                           /* dequantization_int_97 */
        int v = neg ? -(int)mag : (int)mag;
        v = (v + 32) >> 6;
        long long t = (long long)v * i_step;
        return (uint32_t)(int)((t + (1 << 15)) >> 16);
    }
}

/* serial decoder state of lane 0 */
struct HtSerial {
    /* VLC: 64-bit window over the un-stuffed LDS words */
    uint64_t vbuf; int vbits; uint32_t vword;
    const uint32_t *vlc; uint32_t vlc_words;
    /* MEL (jpeg2000htdec.c:429-440, 462-495) */
    const uint8_t *suf; uint32_t mel_pos, mel_len; uint32_t mel_tmp; int mel_bits;
    int mel_k, mel_run, mel_one;

    __device__ __forceinline__ void vfill()
    {
        if (vbits <= 32) {
            uint32_t wv = vword < vlc_words ? vlc[vword] : 0u;
            vword++;
            vbuf |= (uint64_t)wv << vbits;
            vbits += 32;
        }
    }
    __device__ __forceinline__ uint32_t vpeek(int n) { vfill(); return (uint32_t)vbuf & ((1u << n) - 1); }
    __device__ __forceinline__ void vdrop(int n) { vbuf >>= n; vbits -= n; }
    __device__ __forceinline__ uint32_t vget(int n) { uint32_t v = vpeek(n); vdrop(n); return v; }

    __device__ __forceinline__ int mel_bit()
    {
        if (mel_bits == 0) {
            const bool cond = mel_pos < mel_len;
            mel_bits = (mel_tmp == 0xFF) ? 7 : 8;
            mel_tmp = cond ? suf[mel_pos] : 0xFFu;
            mel_pos += cond;
        }
        mel_bits--;
        return (mel_tmp >> mel_bits) & 1;
    }
    __device__ __forceinline__ int mel_sym()
    {
        static const uint8_t MEL_E[13] = { 0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 4, 5 };
        if (mel_run == 0 && mel_one == 0) {
            int eval = (0x5433222111000ull >> (4 * mel_k)) & 0xF;   /* MEL_E packed in nibbles */
            (void)MEL_E;
            if (mel_bit()) {
                mel_run = 1 << eval;
                mel_k = mel_k < 12 ? mel_k + 1 : 12;
            } else {
                mel_run = 0;
                while (eval > 0) { mel_run = 2 * mel_run + mel_bit(); eval--; }
                mel_run &= 0xFF;                                  /* uint8_t run in the reference */
                mel_k = mel_k > 0 ? mel_k - 1 : 0;
                mel_one = 1;
            }
        }
        if (mel_run > 0) { mel_run--; return 0; }
        mel_one = 0;
        return 1;
    }
    /* U-VLC prefix / suffix / extension, jpeg2000htdec.c:338-388 */
    __device__ __forceinline__ int upfx()
    {
        uint32_t b = vpeek(3);
        /* value {5,1,2,1,3,1,2,1}, drop {3,1,2,1,3,1,2,1} packed in nibbles, index = 3 peeked bits */
        int val  = (0x12131215u >> (4 * b)) & 0xF;
        int drop = (0x12131213u >> (4 * b)) & 0xF;
        vdrop(drop);
        return val;
    }
    __device__ __forceinline__ int usfx(int pfx)
    {
        if (pfx < 3) return 0;
        if (pfx == 3) return (int)vget(1);
        return (int)vget(5);
    }
    __device__ __forceinline__ int uext(int sfx) { return sfx >= 28 ? (int)vget(4) : 0; }
};

__device__ __forceinline__ void ht_zero_window(uint32_t *dst, int w, int h, int stride, int lane)
{
    for (int y = 0; y < h; y++)
        for (int x = lane; x < w; x += 64)
            dst[(size_t)y * stride + x] = 0u;
}

__device__ __forceinline__ void ht_zero_window16(uint16_t *dst, int w, int h, int stride, int lane)
{
    for (int y = 0; y < h; y++)
        for (int x = lane; x < w; x += 64)
            dst[(size_t)y * stride + x] = 0;
}

__device__ __forceinline__ int bm_get(const uint32_t *bm, int idx) { return (bm[idx >> 5] >> (idx & 31)) & 1; }

/* ---- MagSgn fast path: blocks of at most 64 sample columns, no ROI shift ----
 * lanes = sample columns; each lane owns the two samples of its column in the current quad row.
 * Everything the row needs from its neighbours moves by DPP: the bottom-row exponents of the
 * row above (kappa, jpeg2000htdec.c:855-885) through two wave shifts and a quad swap, the MagSgn
 * bit offsets through the DPP prefix sum.  Both samples of a lane are cut out of one 96-bit LDS
 * window.  The quad symbols of the next row are prefetched while the current one is processed. */
__device__ __forceinline__ uint32_t ht_dpp_swap_pair(uint32_t v)   /* lane <-> lane ^ 1 */
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);          /* quad_perm [1,0,3,2] */
}
__device__ __forceinline__ uint32_t ht_dpp_left(uint32_t v)        /* lane i <- lane i-1, lane 0 <- 0 */
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t ht_dpp_right(uint32_t v)       /* lane i <- lane i+1, lane 63 <- 0 */
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false);
}

/* ---- un-stuffing, eight bytes per lane and pass ----
 * A lane holds 64 raw stream bits `hi:lo` in read order (LSB first) and a flag at the top bit of every byte that
 * carries 7 bits instead of 8.  Stuffed bytes are rare (one byte in 256 is an 0xFF), so instead of rebuilding the
 * chunk byte by byte the flagged bits are squeezed out one at a time, top down, in two wave-level loops that run once
 * or not at all on most passes; the rest -- bit counts, wave prefix sum, three LDS ORs -- is per pass.  (Round 2
 * merged the eight bytes with per-byte shifts, ~110 VALU instructions per pass; this is ~50.)
 *   OR_SEM = true : MagSgn (jpeg2000htdec.c:207-221): the byte behind an 0xFF is ORed in whole and the stream
 *                   advances by 7 bits, so its top bit lands on the next byte's bit 0 (zero in a conforming stream);
 *   OR_SEM = false: the flagged byte's top bit is dropped (SigProp :1016-1131, the backward VLC / MagRef streams
 *                   :145-201).
 * `nv` = stream bytes in this lane (0..8; bytes past them are zero and carry no flag).  `out` must be zero up to the
 * word after the last stream bit. */
template <bool OR_SEM>
__device__ __forceinline__ void ht_squeeze_place(uint32_t lo, uint32_t hi, uint32_t fl, uint32_t fh, int nv,
                                                 uint32_t *out, int lane, uint32_t &base)
{
    const uint32_t tot = 8u * (uint32_t)nv - (uint32_t)__builtin_popcount(fl) - (uint32_t)__builtin_popcount(fh);
    while (__ballot(fh != 0) != 0) {                     /* flags in bytes 4..7, the highest first */
        if (fh) {
            const uint32_t k = 31u - (uint32_t)__builtin_clz(fh), bit = 1u << k, m = bit - 1u;
            uint32_t nh = (hi & m) | ((hi >> 1) & ~m);
            if (OR_SEM) nh |= hi & bit;
            hi = nh;
            fh &= m;
        }
    }
    while (__ballot(fl != 0) != 0) {                     /* ... then bytes 0..3: the upper word moves down with them */
        if (fl) {
            const uint32_t k = 31u - (uint32_t)__builtin_clz(fl), bit = 1u << k, m = bit - 1u;
            uint32_t nl = (lo & m) | (__builtin_amdgcn_alignbit(hi, lo, 1) & ~m);
            if (OR_SEM) nl |= lo & bit;
            lo = nl;
            hi >>= 1;
            fl &= m;
        }
    }
    const uint32_t incl = wave_incl_scan_u32(tot, lane);
    const uint32_t off = base + incl - tot;
    if (nv > 0) {
        const uint32_t sh = off & 31;
        const uint64_t t = (uint64_t)lo << sh, u = (uint64_t)hi << sh;
        uint32_t *o = out + (off >> 5);
        atomicOr(o, (uint32_t)t);
        atomicOr(o + 1, (uint32_t)(t >> 32) | (uint32_t)u);
        if (sh + tot + (OR_SEM ? 1u : 0u) > 64) atomicOr(o + 2, (uint32_t)(u >> 32));   /* OR_SEM: bit `tot` may be set */
    }
    base += wave_last(incl);
}

/* forward streams: `dq` = the lane's eight bytes (dwords 2 (p0 + lane), 2 (p0 + lane) + 1 of the stream), `nbytes` the
 * stream's length, `carry` the last byte of the pass before (0 at the start).  A byte that follows an 0xFF is flagged: one
 * SWAR test per dword on the dword shifted up by a byte with the byte before it in front (x == 0xFF exactly when bit 7
 * is set and the low seven bits carry into it when 1 is added). */
template <bool OR_SEM>
__device__ __forceinline__ void ht_unstuff_fwd_step8(uint2 dq, uint32_t p0, uint32_t nbytes, uint32_t *out, int lane,
                                                     uint32_t &base, uint32_t &carry)
{
    uint32_t lo = dq.x, hi = dq.y;
    int nv = 8;
    if ((p0 + 64) * 8 > nbytes) {                        /* wave-uniform: the pass that holds the end of the stream */
        nv = min(max((int)nbytes - (int)((p0 + lane) * 8), 0), 8);
        const int nl = min(nv, 4), nh = nv - nl;
        lo = nl == 4 ? lo : (nl ? lo & (0xFFFFFFFFu >> (32 - 8 * nl)) : 0u);
        hi = nh == 4 ? hi : (nh ? hi & (0xFFFFFFFFu >> (32 - 8 * nh)) : 0u);
    }
    uint32_t prev = ht_dpp_left(hi >> 24);
    if (lane == 0) prev = carry;
    carry = (uint32_t)__builtin_amdgcn_readlane((int)(hi >> 24), 63);   /* the last byte of this pass */
    const uint32_t Pl = (lo << 8) | prev, Ph = __builtin_amdgcn_alignbyte(hi, lo, 3);   /* the bytes in front of b0..b3, b4..b7 */
    uint32_t fl = ((Pl & 0x7F7F7F7Fu) + 0x01010101u) & Pl & 0x80808080u;
    uint32_t fh = ((Ph & 0x7F7F7F7Fu) + 0x01010101u) & Ph & 0x80808080u;
    if (nv < 8) {                                        /* bytes past the stream carry no flag */
        const int nl = min(nv, 4), nh = nv - nl;
        fl = nl == 4 ? fl : (nl ? fl & (0xFFFFFFFFu >> (32 - 8 * nl)) : 0u);
        fh = nh == 4 ? fh : (nh ? fh & (0xFFFFFFFFu >> (32 - 8 * nh)) : 0u);
    }
    ht_squeeze_place<OR_SEM>(lo, hi, fl, fh, nv, out, lane, base);
}

/* MagSgn bytes -> bit array (the name the kernels use) */
__device__ __forceinline__ void ht_unstuff_magsgn_step8(uint2 dq, uint32_t p0, uint32_t Pcup, uint32_t *ms, int lane,
                                                        uint32_t &base, uint32_t &carry)
{
    ht_unstuff_fwd_step8<true>(dq, p0, Pcup, ms, lane, base, carry);
}

__device__ __forceinline__ uint32_t ht_unstuff_magsgn(const uint8_t *__restrict__ D, uint32_t Pcup, uint32_t *ms, int lane)
{
    const uint2 *Dq = (const uint2 *)D;
    uint32_t base = 0, carry = 0;
    uint2 pv[4];                                         /* the first 2 KB are requested at once: one memory round trip */
#pragma unroll
    for (int jx = 0; jx < 4; jx++) {
        const uint32_t pi = 64 * jx + lane;
        pv[jx] = pi * 8 < Pcup ? Dq[pi] : make_uint2(0u, 0u);
    }
#pragma unroll
    for (int jx = 0; jx < 4; jx++)
        if (512u * jx < Pcup) ht_unstuff_magsgn_step8(pv[jx], 64 * jx, Pcup, ms, lane, base, carry);
    for (uint32_t p0 = 256; p0 * 8 < Pcup; p0 += 64) {
        const uint32_t pi = p0 + lane;
        ht_unstuff_magsgn_step8(pi * 8 < Pcup ? Dq[pi] : make_uint2(0u, 0u), p0, Pcup, ms, lane, base, carry);
    }
    return base;
}

/* C16: the plane is written as 16-bit samples (`dst` then points at int16_t elements; 5/3 blocks with M_b <= 15
 * only: every dequantised value fits, see htj2k_device.hip) */
template <int TRANSFORM, bool REFINE, bool C16 = false>
__device__ __forceinline__ int ht_magsgn_rows_narrow(const ht_sym_t *__restrict__ qglob, const uint32_t *ms,
                                                     uint32_t *__restrict__ dst, int lane, int w, int h,
                                                     int stride, int pLSB, int maxbp, int M_b, float fscale, int i_step,
                                                     uint32_t last_wi, uint32_t *__restrict__ sink,
                                                     const uint64_t *__restrict__ rb, int z_blk)
{
    const int qw = (w + 1) >> 1, qh = (h + 1) >> 1;
    const int col = lane;
    const bool act = col < 2 * qw;                         /* lanes past the block read qi = 0: no bits, exponent 0 */
    const bool st_ok = col < w;
    const int q = col >> 1, sh = (col & 1) * 2;
    const int dshift = 31 - M_b;
    const uint32_t half = 1u << ((pLSB - 1) & 31);
    /* 5/3 shortcut (block-uniform): valid when the magnitude LSB sits at or above the dequantiser's (always, in a
     * conforming stream) and at least one magnitude bit survives the 31-bit mask */
    const bool direct53 = pLSB >= dshift && pLSB >= 1 && pLSB <= 30;
    const uint32_t keep53 = (uint32_t)(31 - pLSB) & 31u, up53 = (uint32_t)(pLSB - dshift) & 31u, hb53 = (half & 0x7FFFFFFFu) >> (dshift & 31);
    uint32_t ms_pos = 0, Eb = 0;
    int err = 0;
    const int qwp = (int)ht_qsym_pitch((uint32_t)w);       /* k_ht_vlc pads the symbol rows to an even quad count */
    const ht_sym_t *qp = qglob + q;
    uint32_t qi_next = act ? *qp : 0u;
    uint32_t *prow = dst + col;                            /* this lane's column, row 2 * row */
    uint16_t *prow16 = (uint16_t *)dst + col;              /* the same under C16 */
    for (int row = 0; row < qh; row++) {
        const uint32_t qi = qi_next;
        qp += qwp;
        if (row + 1 < qh) qi_next = act ? *qp : 0u;
        const uint32_t rho = (qi | (qi >> 1)) & 0x55, uq = qi >> 8;   /* significance at the even bits: only "more than one?" is asked */
        /* this lane's samples: top = field sh, bottom = field sh + 1 */
        const uint32_t f_t = (qi >> (2 * sh)) & 3, f_b = (qi >> (2 * sh + 2)) & 3;
        const uint32_t s_t = min(f_t, 1u), s_b = min(f_b, 1u);
        const uint32_t k_t = f_t >> 1, k_b = f_b >> 1;
        const uint32_t e_t = (f_t + 1) >> 2, e_b = (f_b + 1) >> 2;
        int kappa = 1;
        if (row > 0) {
            /* own-lane neighbours cover columns c-1, c+1; together with the pair partner's
             * (c^1)-1, (c^1)+1 that is exactly 2q-1 .. 2q+2 -- no lane-dependent select, so no
             * DPP ends up under a divergent EXEC mask */
            const uint32_t nbm = max(ht_dpp_left(Eb), ht_dpp_right(Eb));
            const uint32_t pm = max(Eb, ht_dpp_swap_pair(Eb));
            const int me = (int)max(pm, max(nbm, ht_dpp_swap_pair(nbm)));
            kappa = (rho & (rho - 1)) ? max(me - 1, 1) : 1;           /* gamma: more than one significant sample */
        }
        const int U = kappa + (int)uq;
        if (act && U > maxbp) err = 1;
        /* m = s * U - k as one 24-bit multiply-add; k is only ever set where s is (the CxtVLC tables'
         * e_k is a subset of rho), so m >= 0 */
        const int m_t = __mul24((int)s_t, U) - (int)k_t, m_b = __mul24((int)s_b, U) - (int)k_b;
        const uint32_t nt = (uint32_t)max(m_t, 0), nb = (uint32_t)max(m_b, 0);
        const uint32_t incl = wave_incl_scan_u32(nt + nb, lane);
        const uint32_t pos = ms_pos + incl - nt - nb;
        ms_pos += wave_last(incl);
        /* 96-bit window; words last_wi .. last_wi + 2 are the all-ones continuation of the stream (:207-221) */
        const uint32_t wi = min(pos >> 5, last_wi), bs = pos & 31;
        const uint32_t w0 = ms[wi], w1 = ms[wi + 1], w2 = ms[wi + 2];
        /* v_alignbit takes the shift mod 32, v_bfe a width of 0..31 */
        uint32_t vt = __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(w1, w0, bs), 0u, nt);
        const uint32_t sb = bs + nt;                       /* 0 .. 62 */
        const bool hiw = sb >= 32;
        uint32_t vb = __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(hiw ? w2 : w1, hiw ? w1 : w0, sb), 0u, nb);
        /* m == 0: no bits were read and the sample is zero whatever e_1 says (:395-427) */
        vt += e_t << nt;
        vb += e_b << nb;
        Eb = m_b != 0 ? (uint32_t)(32 - __clz((int)(vb | 1))) : 0u;
        uint32_t mu_t = 0, mu_b = 0;
        if (REFINE || TRANSFORM != J2K_DWT53 || !direct53) {
            mu_t = ((((vt >> 1) + 1) << pLSB) | half) | (vt << 31);
            mu_b = ((((vb >> 1) + 1) << pLSB) | half) | (vb << 31);
            mu_t = m_t != 0 ? mu_t : 0u;
            mu_b = m_b != 0 ? mu_b : 0u;
        }
        const bool two = 2 * row + 1 < h;                  /* odd heights: the outside half of the last quad row is discarded (:976-1007) */
        if (REFINE) {
            /* the SigProp / MagRef decisions of k_ht_refine, three 64-bit masks per sample row (newly
             * significant, its sign, MagRef bit), applied before the dequantisation: both passes work
             * on bit-plane pLSB - 1 (jpeg2000htdec.c:1309-1315, :1066-1100, :1160-1185) */
            const int qq = (pLSB - 1) & 31;
            const uint64_t *r = rb + (size_t)6 * row * HT_REF_STRIDE;   /* rows 2 * row and 2 * row + 1 (layout: k_ht_refine) */
            const uint64_t Rt = r[0], Gt = r[HT_REF_STRIDE], Qt = r[2 * HT_REF_STRIDE];
            uint64_t Rb = 0, Gb = 0, Qb = 0;
            if (two) { Rb = r[3 * HT_REF_STRIDE]; Gb = r[4 * HT_REF_STRIDE]; Qb = r[5 * HT_REF_STRIDE]; }
            auto refine = [&](uint32_t v, uint32_t sig, uint64_t R, uint64_t G, uint64_t Q) -> uint32_t {
                const uint32_t nsig = (uint32_t)(R >> col) & 1u, sgn = (uint32_t)(G >> col) & 1u, mrb = (uint32_t)(Q >> col) & 1u;
                if (nsig) v |= (1u << qq) | (1u << ((qq - 1) & 31)) | (sgn << 31);
                if (z_blk > 2 && sig) { v &= (0xFFFFFFFEu | mrb) << qq; v |= 1u << ((qq - 1) & 31); }
                return v;
            };
            const uint32_t o_t = ht_dequant(refine(mu_t, s_t, Rt, Gt, Qt), TRANSFORM, M_b, 0, fscale, i_step);
            const uint32_t o_b = ht_dequant(refine(mu_b, s_b, Rb, Gb, Qb), TRANSFORM, M_b, 0, fscale, i_step);
            *(st_ok ? prow : sink) = o_t;
            *((st_ok && two) ? prow + stride : sink) = o_b;
        } else {
            uint32_t o_t, o_b;
            if (TRANSFORM == J2K_DWT53) {
                int r_t, r_b;
                if (direct53) {
                    /* ((((v >> 1) + 1) << pLSB | half) & 0x7FFFFFFF) >> dshift without building mu: the
                     * magnitude keeps its low 31 - pLSB bits and moves up by pLSB - dshift >= 0 */
                    const int sg_t = -(int)(vt & 1), sg_b = -(int)(vb & 1);
                    r_t = (int)((__builtin_amdgcn_ubfe((vt >> 1) + 1, 0u, keep53) << up53) | hb53);
                    r_b = (int)((__builtin_amdgcn_ubfe((vb >> 1) + 1, 0u, keep53) << up53) | hb53);
                    r_t = (r_t ^ sg_t) - sg_t;
                    r_b = (r_b ^ sg_b) - sg_b;
                    r_t = m_t != 0 ? r_t : 0;
                    r_b = m_b != 0 ? r_b : 0;
                } else {
                    const int sg_t = (int)mu_t >> 31, sg_b = (int)mu_b >> 31;
                    r_t = (int)((mu_t & 0x7FFFFFFFu) >> dshift); r_b = (int)((mu_b & 0x7FFFFFFFu) >> dshift);
                    r_t = (r_t ^ sg_t) - sg_t;
                    r_b = (r_b ^ sg_b) - sg_b;
                }
                if (i_step != 32768) {                      /* wave-uniform; reversible bands have step 1.0 */
                    const long long a = (long long)r_t * i_step, b2 = (long long)r_b * i_step;
                    r_t = (int)(a < 0 ? -((-a) >> 16) : (a >> 16));
                    r_b = (int)(b2 < 0 ? -((-b2) >> 16) : (b2 >> 16));
                }
                o_t = (uint32_t)r_t; o_b = (uint32_t)r_b;
            } else if (TRANSFORM == J2K_DWT97) {
                /* (float)(-x) * s == -((float)x * s), and a zero magnitude never carries a sign */
                o_t = __float_as_uint((float)(mu_t & 0x7FFFFFFFu) * fscale) | (mu_t & 0x80000000u);
                o_b = __float_as_uint((float)(mu_b & 0x7FFFFFFFu) * fscale) | (mu_b & 0x80000000u);
            } else {
                o_t = ht_dequant(mu_t, TRANSFORM, M_b, 0, fscale, i_step);
                o_b = ht_dequant(mu_b, TRANSFORM, M_b, 0, fscale, i_step);
            }
            /* exactly two stores per row, no branch around them: lanes (and the odd last row) that
             * have nothing to write aim at a scratch dword.  With a fixed number of stores behind
             * the prefetch of the next row's symbols the wait at the loop top is vmcnt(2), not
             * vmcnt(0) -- the rows of a block no longer wait for each other's stores to land. */
            if (C16) {
                *(st_ok ? prow16 : (uint16_t *)sink) = (uint16_t)o_t;
                *((st_ok && two) ? prow16 + stride : (uint16_t *)sink) = (uint16_t)o_b;
            } else {
                *(st_ok ? prow : sink) = o_t;
                *((st_ok && two) ? prow + stride : sink) = o_b;
            }
        }
        prow += 2 * stride;
        prow16 += 2 * stride;
    }
    return __any(err) ? HT_ERR_INVALID : 0;
}

/* ================================================================== k_ht_decode_pair
 * MagSgn + dequantisation of TWO codeblocks per wavefront, one LANE PER QUAD (lanes 0-31: block 2 g, lanes 32-63:
 * block 2 g + 1 of the table).  In k_ht_decode<true> a lane owns one sample column, so the two lanes of a quad both
 * do the quad's work (symbol fields, kappa from the row above, U, the error test: about a quarter of the kernel's
 * instructions, and the kernel is bound by instruction issue).  Here that work is done once per quad and a lane
 * cuts its quad's four samples out of one 128-bit LDS window.  Only for what the bench-critical case needs -- jobs
 * with 16-bit sub-bands (htj2k_device.hip: reversible 5/3, cleanup pass only, M_b <= 15, step 1, no ROI shift) whose
 * blocks all have an even width of at most 64 columns -- everything else goes through k_ht_decode<true>.
 * Same arithmetic as ht_magsgn_rows_narrow (jpeg2000htdec.c:855-885 kappa, :395-427 mu/E, jpeg2000dec.c:2120-2151). */
__device__ __forceinline__ uint32_t half_incl_scan_u32(uint32_t v)   /* inclusive prefix sum inside each 32-lane half */
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);          /* row_bcast15 into rows 1 and 3 */
    return (uint32_t)x;
}

__global__ void __launch_bounds__(64)
k_ht_decode_pair(const J2kBlock *__restrict__ blocks, int nblocks, const uint8_t *__restrict__ bytes,
                 uint32_t *__restrict__ coef, int *__restrict__ status, uint32_t ms_words,
                 const ht_sym_t *__restrict__ qsym, const uint32_t *__restrict__ qoff, uint32_t *__restrict__ sink)
{
    extern __shared__ __align__(16) uint8_t smem[];
    uint32_t *ms_all = (uint32_t *)smem;                     /* [2][ms_words + 4] */
    const int lane = threadIdx.x, hf = lane >> 5, q = lane & 31;
    const uint32_t mspitch = ms_words + 4;
    bool ok_h[2] = { false, false };
    uint32_t lastwi_h[2] = { 0, 0 }, Pcup_h[2] = { 0, 0 };
    const uint32_t *Dw_h[2] = { nullptr, nullptr };
    uint2 pv[2][4];                                          /* the first 2 KB of each block's MagSgn bytes */

    /* ---- this lane's quad symbols of the first 32 quad rows, requested before anything else and kept in 16 registers
     * (rows 2 k | 2 k + 1 << 16).  With one 2-byte load per row inside the row loop, the wait for row r + 1's symbol is
     * -- the memory counter being in order -- also a wait for the row stores issued before that load: every row paid a
     * store round trip (0.35 of the kernel's 1.24 ms).  Now the loop of a block of up to 64 rows holds no load at all and
     * never waits for its stores.  Rows past the block's last are read from whatever follows in the symbol array
     * (padded by the host) and are masked off below; lanes past the block's width read column 0. ---- */
    const int bi = min(2 * (int)blockIdx.x + hf, nblocks - 1);
    uint32_t sy[16];
    {
        const uint32_t wh = *(const uint32_t *)&blocks[bi].w;      /* w | h << 16 */
        const int w_l = (int)(wh & 0xFFFFu), qw_l = (w_l + 1) >> 1;
        const uint32_t qwp_l = ht_qsym_pitch((uint32_t)w_l);
        const ht_sym_t *qp0 = qsym + qoff[bi] + (q < qw_l ? q : 0);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t a = qp0[0], c = qp0[qwp_l];
            sy[k] = a | (c << 16);
            qp0 += 2 * qwp_l;
        }
        if (q >= qw_l) {
#pragma unroll
            for (int k = 0; k < 16; k++) sy[k] = 0;
        }
    }

    /* ---- per block, whole wave: checks, zero-fill of blocks without passes; the MagSgn bytes of both blocks are
     * requested before either is worked on (a wave's start is a chain of dependent loads otherwise) ---- */
#pragma unroll
    for (int hb = 0; hb < 2; hb++) {
#pragma unroll
        for (int jx = 0; jx < 4; jx++) pv[hb][jx] = make_uint2(0u, 0u);
        const int bidx = 2 * (int)blockIdx.x + hb;
        if (bidx >= nblocks) continue;
        const J2kBlock b = blocks[bidx];
        uint16_t *dst16 = (uint16_t *)coef + b.plane_off;
        if (b.npasses == 0) {
            ht_zero_window16(dst16, b.w, b.h, b.stride, lane);
            continue;
        }
        const int rem = b.npasses % 3, num_plhd = rem ? b.npasses - rem : b.npasses - 3;
        const int S_blk = (num_plhd / 3 + b.zbp) & 0xFF, maxbp = S_blk + 1;
        const uint32_t Lcup = b.lcup;
        const uint8_t *D = bytes + b.data_off;
        Dw_h[hb] = (const uint32_t *)D;
#pragma unroll
        for (int jx = 0; jx < 4; jx++) {                     /* before Scup says where the MagSgn bytes end: Pcup <= Lcup */
            const uint32_t pi = 64 * jx + lane;
            if (pi * 8 < Lcup) pv[hb][jx] = ((const uint2 *)Dw_h[hb])[pi];   /* (the pool is padded: the pair may end past Lcup) */
        }
        int err = 0;
        uint32_t Scup = 0, Pcup = 0;
        if (Lcup < 2) err = HT_ERR_INVALID;
        if (!err) {
            Scup = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((uint32_t)D[Lcup - 1] << 4) + (D[Lcup - 2] & 0x0F)));
            if (Scup < 2 || Scup > Lcup || Scup > 4079) err = HT_ERR_INVALID;
            Pcup = Lcup - Scup;
        }
        if (!err && maxbp >= 32) err = HT_ERR_INVALID;
        if (!err && (Pcup * 8 + 31) / 32 + 3 > ms_words) err = HT_ERR_INVALID;
        if (err) {
            ht_zero_window16(dst16, b.w, b.h, b.stride, lane);
            if (lane == 0) status[bidx] = err;
            continue;
        }
        ok_h[hb] = true;
        Pcup_h[hb] = Pcup;
    }
#pragma unroll
    for (int hb = 0; hb < 2; hb++) {
        if (!ok_h[hb]) continue;
        const uint32_t Pcup = Pcup_h[hb];
        uint32_t *ms = ms_all + hb * mspitch;
        const uint32_t nms = (Pcup * 8 + 31) / 32 + 2;
        ht_ms_clear(ms, nms + 2, lane);
        __syncthreads();
        uint32_t ms_total = 0, carry = 0;
#pragma unroll
        for (int jx = 0; jx < 4; jx++)
            if (512u * jx < Pcup) ht_unstuff_magsgn_step8(pv[hb][jx], 64 * jx, Pcup, ms, lane, ms_total, carry);
        for (uint32_t p0 = 256; p0 * 8 < Pcup; p0 += 64) {
            const uint32_t pi = p0 + lane;
            ht_unstuff_magsgn_step8(pi * 8 < Pcup ? ((const uint2 *)Dw_h[hb])[pi] : make_uint2(0u, 0u), p0, Pcup, ms, lane, ms_total, carry);
        }
        __syncthreads();
        ht_ms_ones_tail(ms, ms_total, nms + 1, lane);        /* past the end the MagSgn stream is all ones (:207-221) */
        lastwi_h[hb] = nms - 2;                              /* words last_wi .. last_wi + 3 exist and are ones past the end */
    }
    __syncthreads();
    if (!ok_h[0] && !ok_h[1]) return;

    /* ---- both blocks in lockstep: this lane's block ---- */
    const J2kBlock b = blocks[bi];
    const bool ok = hf ? ok_h[1] : ok_h[0];
    const int w = b.w, h = b.h, stride = b.stride, M_b = b.M_b;
    const int qw = (w + 1) >> 1, qh = ok ? (h + 1) >> 1 : 0;
    const int rem = b.npasses % 3, num_plhd = rem ? b.npasses - rem : b.npasses - 3;
    const int S_blk = (num_plhd / 3 + b.zbp) & 0xFF;
    const int pLSB = (30 - S_blk) & 0xFF, maxbp = S_blk + 1, dshift = 31 - M_b;
    const uint32_t halfbit = 1u << ((pLSB - 1) & 31);
    const uint32_t *ms = ms_all + hf * mspitch;
    const uint32_t last_wi = hf ? lastwi_h[1] : lastwi_h[0];
    const bool act = ok && q < qw;
    const int qwp = (int)ht_qsym_pitch((uint32_t)w);
    const ht_sym_t *qp = qsym + qoff[bi] + q;
    uint16_t *prow = (uint16_t *)coef + b.plane_off + 2 * q;     /* this quad's two columns, row 2 * row */
    const int rows = max(__builtin_amdgcn_readlane(qh, 0), __builtin_amdgcn_readlane(qh, 32));
    uint32_t qi_next = (act && qh > 0) ? *qp : 0u;
    uint32_t E1p = 0, E3p = 0, ms_pos = 0;
    int err = 0;
    /* The common case, taken when both blocks of the wave allow it: the magnitude LSB sits at or above the dequantiser's
     * (pLSB >= 31 - M_b: every conforming stream) and a sample has at most 16 magnitude bits (maxbp <= 16; a larger U is
     * an error that zeroes the block).  A quad then takes at most 64 MagSgn bits: one 96-bit LDS window per quad instead
     * of two words per sample; the four bit counts, their prefix sums and the significance masks are bytes of one
     * register (SWAR); the dequantiser is (t + 1) << (pLSB - dshift) | half bit, exact under those conditions.
     * 180 -> ~125 VALU instructions per quad row in a kernel that is bound by instruction issue. */
    const bool fast_blk = !ok || (pLSB >= dshift && pLSB >= 1 && pLSB <= 30 && maxbp <= 16);
    if (__ballot(!fast_blk) == 0) {
        const uint32_t up = (uint32_t)(pLSB - dshift) & 31u, hb = (halfbit & 0x7FFFFFFFu) >> (dshift & 31);
        /* SIMPLE (wave-uniform): both blocks have the same, even, height -- nearly every wave, the block table is sorted
         * by size.  Then which lanes store is fixed for the whole loop: lanes outside their block aim at the scratch
         * line from the start (stride 0), and every row issues exactly two stores behind the prefetch of the next row's
         * symbols, so that the wait at the loop top is vmcnt(2), not vmcnt(0) (see ht_magsgn_rows_narrow). */
        /* Per-lane constants of the loop.  The exponent predictor only ever asks for the LARGEST exponent of four samples
         * of the row above (this quad's two bottom samples, the left quad's bottom-right, the right quad's bottom-left,
         * jpeg2000htdec.c:855-885), and max(32 - clz(a), 32 - clz(b)) = 32 - clz(a | b): so the row keeps the two bottom
         * values themselves, packed (bottom-left | bottom-right << 16, each ORed with its significance bit so that a
         * significant zero counts as exponent 1), the neighbours' halves are ORed in through two DPP moves, and ONE
         * count-leading-zeros gives the maximum.  kappa = gamma ? max(E - 1, 1) : 1 = 1 + (gamma ? max(E - 2, 0) : 0).
         * A quad outside the block reads symbol 0 and contributes nothing; nothing crosses the boundary between the two blocks (lanes 31 | 32). */
        const uint32_t LmHi = q == 0 ? 0u : 0xFFFF0000u, RmLo = q == 31 ? 0u : 0x0000FFFFu;
        /* dequantisation of two samples per instruction: ((v >> 1) + 1) << up | hb = (v >> 1) * 2^up + (2^up | hb), as hb < 2^up */
        const uint32_t mul2 = ((1u << up) & 0xFFFFu) * 0x00010001u, add2 = (((1u << up) | hb) & 0xFFFFu) * 0x00010001u;
        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
        /* PRE: the wave's blocks have at most 32 quad rows and their symbols sit in sy[] */
        auto fast_rows = [&](auto simple_tag, auto pre_tag) {
        constexpr bool SIMPLE = decltype(simple_tag)::value, PRE = decltype(pre_tag)::value;
        uint16_t *pt = (SIMPLE && !act) ? (uint16_t *)sink : prow;
        const int st = (SIMPLE && !act) ? 0 : stride;
        uint32_t Zb = 0;                                                    /* the row above: bottom-left' | bottom-right' << 16 */
        uint32_t Umax = 0;                                                  /* SIMPLE: the error test (U > maxbp) once, after the loop */
        const uint32_t minus2 = 0xFFFEFFFEu;
        /* one quad row; `sym` holds the lane's symbol in its low (HI = false) or high 16 bits */
        auto quad_row = [&](int row, uint32_t sym, auto hi_tag) {
            constexpr bool HI = decltype(hi_tag)::value;
            const bool arow = SIMPLE ? act : (act && row < qh);
            /* field n of the symbol -> bits 0-1 of byte n (ORs of shifted copies, not one multiply: the copies overlap and a
             * product would carry); R / K / X1: significant / exponent bound / MSB known, bit 0 of byte n */
            const uint32_t pk = HI ? (sym >> 16) & 0xFFu : sym & 0xFFu, uq = HI ? sym >> 24 : (sym >> 8) & 0xFFu;
            const uint32_t pk2 = pk | (pk << 12);
            const uint32_t F = (pk2 | (pk2 << 6)) & 0x03030303u, Fh = F >> 1;
            const uint32_t R = (F | Fh) & 0x01010101u, K = Fh & 0x01010101u, X1 = F & Fh;
            /* kappa + u.  Row 0: Zb = 0, the maximum is "1", kappa = 1 */
            uint32_t Z = (ht_dpp_left(Zb) & LmHi) | Zb;
            Z = (ht_dpp_right(Zb) & RmLo) | Z;
            const uint32_t Zf = ((Z >> 16) | Z | 1u) & 0xFFFFu;
            const int em2 = max(30 - (int)__builtin_clz(Zf), 0);            /* max(E - 2, 0) */
            const bool gamma = __builtin_amdgcn_sad_u8(R, 0u, 0u) > 1u;     /* more than one significant sample */
            const uint32_t U = (gamma ? (uint32_t)em2 : 0u) + 1u + uq;
            if (SIMPLE) Umax = max(Umax, U);                                /* (lanes outside their block: symbol 0, U = 1) */
            else if (arow && (int)U > maxbp) err = 1;
            uint32_t Rs = R << 8;
            asm("" : "+v"(Rs));                                             /* (or the compiler makes it R * 255: v_mul_lo_u32 is quarter rate) */
            const uint32_t Rm = Rs - R;                                     /* 0xFF in the bytes of significant samples */
            const uint32_t U4 = __builtin_amdgcn_perm(U, U, 0x00000000u);   /* U in all four bytes (U < 256) */
            const uint32_t N = (U4 - K) & Rm;                               /* m_n = sigma_n * U - k_n, one byte each (:883-888) */
            const uint32_t tot = __builtin_amdgcn_sad_u8(N, 0u, 0u);         /* sum of the four bytes */
            const uint32_t incl = half_incl_scan_u32(tot);
            const uint32_t pos = ms_pos + incl - tot;
            ms_pos += (uint32_t)__builtin_amdgcn_ds_swizzle((int)incl, 31 << 5);   /* lane 31 of this half: one LDS-pipe op, no readlane + select */
            const uint32_t wi = min(pos >> 5, last_wi + 1);
            const uint32_t w0 = ms[wi], w1 = ms[wi + 1], w2 = ms[wi + 2];
            uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, pos), hi = __builtin_amdgcn_alignbit(w2, w1, pos);
            /* the bit counts as shift amounts and field widths: the instructions look at the low five bits only (m_n <= 16 here) */
            const uint32_t N1 = N >> 8, N2 = N >> 16, N3 = N >> 24;
            const uint32_t v0 = __builtin_amdgcn_ubfe(lo, 0u, N);
            lo = __builtin_amdgcn_alignbit(hi, lo, N); hi = __builtin_amdgcn_alignbit(0u, hi, N);
            const uint32_t v1 = __builtin_amdgcn_ubfe(lo, 0u, N1);
            lo = __builtin_amdgcn_alignbit(hi, lo, N1); hi = __builtin_amdgcn_alignbit(0u, hi, N1);
            const uint32_t v2 = __builtin_amdgcn_ubfe(lo, 0u, N2);
            lo = __builtin_amdgcn_alignbit(hi, lo, N2);
            const uint32_t v3 = __builtin_amdgcn_ubfe(lo, 0u, N3);
            /* pairs: top = samples 0 | 2 << 16, bottom = 1 | 3 << 16 (one byte permute each: v < 2^16); the known MSB goes in at
             * bit m_n (where it is known the bound took a bit away, m_n <= 15: the 16-bit shift sees the low four bits of each
             * half of N resp. N >> 8).  The bottom pair's masks are the 16-bit halves of R and X1 shifted down a byte. */
            uint32_t xt, xb, sgB, X1b;
            asm("v_pk_lshrrev_b16 %0, 8, %1 op_sel_hi:[0,1]" : "=v"(sgB) : "v"(R));   /* (the constant's low half for both lanes) */
            asm("v_pk_lshrrev_b16 %0, 8, %1 op_sel_hi:[0,1]" : "=v"(X1b) : "v"(X1));
            asm("v_pk_lshlrev_b16 %0, %1, %2" : "=v"(xt) : "v"(N), "v"(X1 & 0x00010001u));
            asm("v_pk_lshlrev_b16 %0, %1, %2" : "=v"(xb) : "v"(N1), "v"(X1b));
            const uint32_t Pt = __builtin_amdgcn_perm(v2, v0, 0x05040100u) | xt, Pb = __builtin_amdgcn_perm(v3, v1, 0x05040100u) | xb;
            const uint32_t sgT = R & 0x00010001u;
            Zb = Pb | sgB;                                                  /* feeds the next row */
            /* mu (:407-427) -> dequantization_int, two samples per instruction: v < 2^16 (at most 16 magnitude bits, the
             * known MSB only where the bound took one away), the result < 2^M_b <= 2^15 -- and whatever a block that is
             * about to be rejected overflows stays inside its own half.  The sign (bit 0 of v) and "not significant" are
             * one multiplier: sig - 2 (v & 1) = +1, -1 or 0 */
            auto samples = [&](uint32_t P, uint32_t sig) -> uint32_t {
                const u16x2 r = (__builtin_bit_cast(u16x2, P) >> (u16x2){ 1, 1 }) * __builtin_bit_cast(u16x2, mul2) + __builtin_bit_cast(u16x2, add2);
                uint32_t m;
                asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(m) : "v"(P & 0x00010001u), "v"(minus2), "v"(sig));
                return __builtin_bit_cast(uint32_t, (u16x2)(r * __builtin_bit_cast(u16x2, m)));
            };
            const uint32_t top = samples(Pt, sgT), bot = samples(Pb, sgB);
            if (SIMPLE) {
                *(uint32_t *)pt = top;
                *(uint32_t *)(pt + st) = bot;
            } else {
                const bool two = 2 * row + 1 < h;
                *(arow ? (uint32_t *)pt : sink) = top;
                *((arow && two) ? (uint32_t *)(pt + st) : sink) = bot;
            }
            pt += 2 * st;
        };
        if (PRE) {                                                          /* two rows per register, no load in the loop */
            for (int k = 0; 2 * k < rows; k++) {
                uint32_t s2 = sy[k];                                        /* (uniform index) */
                if (!SIMPLE) {                                              /* rows past this lane's block; a rejected block */
                    s2 = (act && 2 * k < qh) ? s2 : 0u;
                    s2 = (2 * k + 1 < qh) ? s2 : s2 & 0xFFFFu;
                }
                quad_row(2 * k, s2, std::false_type{});
                if (2 * k + 1 < rows) quad_row(2 * k + 1, s2, std::true_type{});
            }
        } else {
            for (int row = 0; row < rows; row++) {
                const uint32_t qi = qi_next;
                qp += qwp;
                qi_next = (act && row + 1 < (SIMPLE ? rows : qh)) ? *qp : 0u;
                quad_row(row, qi, std::false_type{});
            }
        }
        if (SIMPLE && act && (int)Umax > maxbp) err = 1;
        };
        const int qh0 = __builtin_amdgcn_readlane(qh, 0), qh1 = __builtin_amdgcn_readlane(qh, 32);
        const bool simple = qh0 == qh1 && __ballot(ok && (h & 1)) == 0;
        if (rows <= 32) {
            if (simple) fast_rows(std::true_type{}, std::true_type{}); else fast_rows(std::false_type{}, std::true_type{});
        } else {
            if (simple) fast_rows(std::true_type{}, std::false_type{}); else fast_rows(std::false_type{}, std::false_type{});
        }
    } else
    for (int row = 0; row < rows; row++) {
        const bool arow = act && row < qh;
        const uint32_t qi = qi_next;
        qp += qwp;
        qi_next = (act && row + 1 < qh) ? *qp : 0u;
        const uint32_t qx = ht_sym_unpack(qi);
        const uint32_t rho = qx & 0xF, ek = (qx >> 4) & 0xF, e1 = (qx >> 8) & 0xF, uq = (qx >> 16) & 0xFF;
        int kappa = 1;
        if (row > 0) {
            /* exponents of the row above at columns 2q-1 .. 2q+2: the neighbours' come by DPP, nothing crosses the
             * boundary between the two blocks (lanes 31 | 32) */
            uint32_t l = ht_dpp_left(E3p), r = ht_dpp_right(E1p);
            l = q == 0 ? 0u : l;
            r = q == 31 ? 0u : r;
            const int me = (int)max(max(E1p, E3p), max(l, r));
            kappa = (rho & (rho - 1)) ? max(me - 1, 1) : 1;
        }
        const int U = kappa + (int)uq;
        if (arow && U > maxbp) err = 1;
        const int m0 = __mul24((int)(rho & 1), U) - (int)(ek & 1), m1 = __mul24((int)((rho >> 1) & 1), U) - (int)((ek >> 1) & 1);
        const int m2 = __mul24((int)((rho >> 2) & 1), U) - (int)((ek >> 2) & 1), m3 = __mul24((int)((rho >> 3) & 1), U) - (int)((ek >> 3) & 1);
        const uint32_t n0 = (uint32_t)max(m0, 0), n1 = (uint32_t)max(m1, 0), n2 = (uint32_t)max(m2, 0), n3 = (uint32_t)max(m3, 0);
        const uint32_t tot = n0 + n1 + n2 + n3;
        const uint32_t incl = half_incl_scan_u32(tot);
        const uint32_t pos = ms_pos + incl - tot;
        ms_pos += (uint32_t)__builtin_amdgcn_ds_swizzle((int)incl, 31 << 5);
        /* every sample cuts its bits out of the two LDS words they start in (no register window: selecting the word
         * pair of a sample by a per-lane index makes the compiler put the window into scratch memory) */
        auto cut = [&](uint32_t p, uint32_t n) -> uint32_t {
            const uint32_t i = min(p >> 5, last_wi + 2);
            return __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(ms[i + 1], ms[i], p), 0u, n);
        };
        const uint32_t p1 = pos + n0, p2 = p1 + n1, p3 = p2 + n2;
        uint32_t v0 = cut(pos, n0), v1 = cut(p1, n1), v2 = cut(p2, n2), v3 = cut(p3, n3);
        v0 += (e1 & 1) << n0; v1 += ((e1 >> 1) & 1) << n1; v2 += ((e1 >> 2) & 1) << n2; v3 += ((e1 >> 3) & 1) << n3;
        E1p = m1 != 0 ? (uint32_t)(32 - __clz((int)(v1 | 1))) : 0u;      /* bottom-left and bottom-right feed the next row */
        E3p = m3 != 0 ? (uint32_t)(32 - __clz((int)(v3 | 1))) : 0u;
        auto sample = [&](uint32_t v, int m) -> uint32_t {               /* mu (:407-427) -> dequantization_int, 16 bits */
            const uint32_t mu = ((((v >> 1) + 1) << pLSB) | halfbit) & 0x7FFFFFFFu;
            int r = (int)(mu >> dshift);
            const int sg = -(int)(v & 1);
            r = (r ^ sg) - sg;
            return (uint32_t)(m != 0 ? r : 0) & 0xFFFFu;
        };
        const uint32_t top = sample(v0, m0) | sample(v2, m2) << 16, bot = sample(v1, m1) | sample(v3, m3) << 16;
        const bool two = 2 * row + 1 < h;
        /* exactly two stores per row, as in ht_magsgn_rows_narrow */
        *(arow ? (uint32_t *)prow : sink) = top;
        *((arow && two) ? (uint32_t *)(prow + stride) : sink) = bot;
        prow += 2 * stride;
    }
    /* a block whose U ran past maxbp is rejected as a whole (:862-868): zero it, whole wave per block */
    const bool e0 = __ballot(err && hf == 0) != 0, e1b = __ballot(err && hf == 1) != 0;
#pragma unroll
    for (int hb = 0; hb < 2; hb++) {
        if (!(hb ? e1b : e0)) continue;
        const int bidx = 2 * (int)blockIdx.x + hb;
        const J2kBlock bb = blocks[bidx];
        __syncthreads();
        ht_zero_window16((uint16_t *)coef + bb.plane_off, bb.w, bb.h, bb.stride, lane);
        if (lane == 0) status[bidx] = HT_ERR_INVALID;
    }
}

/* ================================================================== k_ht_decode_multi
 * k_ht_decode_pair's arrangement -- a LANE PER QUAD, several codeblocks per wavefront -- for the jobs that keep 32-bit
 * sub-bands: NB = 2 blocks of up to 64 columns (lanes 0-31 / 32-63) or NB = 4 blocks of up to 32 columns (16 lanes
 * each; cb = 32 streams left half the lanes of the column-per-lane kernel idle), any of the three dequantisers
 * (jpeg2000dec.c:2098-2181), any width and height.  For jobs whose HT blocks are all cleanup-only, at most 64 columns
 * wide, without ROI shift and of one transform (htj2k_device.hip: multi_ok); everything else stays with
 * k_ht_decode<true>.  Same arithmetic as ht_magsgn_rows_narrow (jpeg2000htdec.c:855-885 kappa, :395-427 mu / E). */
template <int LPB>
__device__ __forceinline__ uint32_t seg_incl_scan_u32(uint32_t v)    /* inclusive prefix sum inside each run of LPB lanes */
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    if (LPB == 32) x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);      /* row_bcast15 into rows 1 and 3 */
    return (uint32_t)x;
}
template <int LPB>
__device__ __forceinline__ uint32_t seg_last(uint32_t v)             /* the value of the last lane of this lane's run */
{
    /* ds_swizzle, bit mode: lane' = (lane & and) | or inside each half of the wave */
    return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, LPB == 32 ? (31 << 5) : (0x10 | (15 << 5)));
}

/* REFINE: some blocks of the job carry SigProp / MagRef passes; their decisions come from k_ht_refine as three 64-bit
 * masks per sample row (newly significant, its sign, MagRef bit) and are applied to mu before the dequantisation, as
 * in ht_magsgn_rows_narrow (jpeg2000htdec.c:1309-1315, :1066-1100, :1160-1185) */
/* LDS of a REFINE launch behind the MagSgn bit arrays, per block: the un-stuffed MagRef bits (mr_words words), the SigProp
 * masks of k_ht_refine (newly significant, sign: 2 masks of MW words per sample row, rg_rows rows); then once per wave two
 * 256-byte tables: symbol fields -> the quad's four significance bits, and (4 significance bits, 4 stream bits) -> the
 * stream bits dealt out to the significant positions */
__host__ __device__ inline size_t ht_multi_refine_lds(int nb, uint32_t mr_words, uint32_t rg_rows)
{
    const uint32_t mw = nb == 4 ? 1u : 2u;
    return (size_t)nb * (mr_words + 2 * mw * rg_rows) * 4 + 512;
}

template <int NB, int TRANSFORM, bool REFINE>
__global__ void __launch_bounds__(64)
k_ht_decode_multi(const J2kBlock *__restrict__ blocks, int nblocks, const uint8_t *__restrict__ bytes,
                  uint32_t *__restrict__ coef, int *__restrict__ status, uint32_t ms_words,
                  const ht_sym_t *__restrict__ qsym, const uint32_t *__restrict__ qoff, uint32_t *__restrict__ sink,
                  const uint64_t *__restrict__ refbits, const uint32_t *__restrict__ roff,
                  const uint32_t *__restrict__ mel_u, uint32_t mr_words, uint32_t rg_rows)
{
    constexpr int LPB = 64 / NB, PF = 8 / NB;                /* lanes per block; 512-byte pieces of a block's bytes requested up front */
    constexpr int MW = NB == 4 ? 1 : 2;                      /* words of a SigProp row mask: blocks of at most 32 / 64 columns */
    extern __shared__ __align__(16) uint8_t smem[];
    uint32_t *ms_all = (uint32_t *)smem;                     /* [NB][ms_words + 4] */
    const int lane = threadIdx.x, seg = lane / LPB, q = lane % LPB;
    const uint32_t mspitch = ms_words + 4;
    uint32_t *mr_all = ms_all + NB * mspitch;                /* REFINE: [NB][mr_words] */
    uint32_t *rg_all = mr_all + NB * mr_words;               /* REFINE: [NB][rg_rows][R, G][MW] */
    uint8_t *lut_rho = (uint8_t *)(rg_all + NB * 2 * MW * rg_rows), *lut_dep = lut_rho + 256;
    bool ok_s[NB];
    uint32_t lastwi_s[NB], Pcup_s[NB], Scup_s[NB];
    const uint32_t *Dw_s[NB];
    uint2 pv[NB][PF];

    /* ---- this lane's quad symbols of the first 32 quad rows, requested before anything else (see k_ht_decode_pair) ---- */
    const int bi = min(NB * (int)blockIdx.x + seg, nblocks - 1);
    uint32_t sy[16];
    {
        const uint32_t wh = *(const uint32_t *)&blocks[bi].w;      /* w | h << 16 */
        const int w_l = (int)(wh & 0xFFFFu), qw_l = (w_l + 1) >> 1;
        const uint32_t qwp_l = ht_qsym_pitch((uint32_t)w_l);
        const ht_sym_t *qp0 = qsym + qoff[bi] + (q < qw_l ? q : 0);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t a = qp0[0], c = qp0[qwp_l];
            sy[k] = a | (c << 16);
            qp0 += 2 * qwp_l;
        }
        if (q >= qw_l) {
#pragma unroll
            for (int k = 0; k < 16; k++) sy[k] = 0;
        }
    }
    if (REFINE) {
        /* field byte of a symbol -> rho of the quad; (rho-like mask, four stream bits) -> the bits at the mask's set positions */
        for (int i = lane; i < 256; i += 64) {
            const uint32_t x = ((uint32_t)i | ((uint32_t)i >> 1)) & 0x55u;
            lut_rho[i] = (uint8_t)((x & 1u) | ((x >> 1) & 2u) | ((x >> 2) & 4u) | ((x >> 3) & 8u));
            uint32_t m = (uint32_t)i >> 4, bts = (uint32_t)i & 15u, d = 0;
            for (int n = 0; n < 4; n++)
                if ((m >> n) & 1u) { d |= (bts & 1u) << n; bts >>= 1; }
            lut_dep[i] = (uint8_t)d;
        }
    }

    /* ---- per block, whole wave: checks, zero-fill of blocks without passes; the MagSgn bytes of all blocks are
     * requested before any is worked on ---- */
#pragma unroll
    for (int hb = 0; hb < NB; hb++) {
        ok_s[hb] = false; lastwi_s[hb] = 0; Pcup_s[hb] = 0; Scup_s[hb] = 0; Dw_s[hb] = nullptr;
#pragma unroll
        for (int jx = 0; jx < PF; jx++) pv[hb][jx] = make_uint2(0u, 0u);
        const int bidx = NB * (int)blockIdx.x + hb;
        if (bidx >= nblocks) continue;
        const J2kBlock b = blocks[bidx];
        uint32_t *dst = coef + b.plane_off;
        if (b.npasses == 0) {
            ht_zero_window(dst, b.w, b.h, b.stride, lane);
            continue;
        }
        const int rem = b.npasses % 3, num_plhd = rem ? b.npasses - rem : b.npasses - 3;
        const int S_blk = (num_plhd / 3 + b.zbp) & 0xFF, maxbp = S_blk + 1;
        const uint32_t Lcup = b.lcup;
        const uint8_t *D = bytes + b.data_off;
        Dw_s[hb] = (const uint32_t *)D;
#pragma unroll
        for (int jx = 0; jx < PF; jx++) {                    /* before Scup says where the MagSgn bytes end: Pcup <= Lcup */
            const uint32_t pi = 64 * jx + lane;
            if (pi * 8 < Lcup) pv[hb][jx] = ((const uint2 *)Dw_s[hb])[pi];
        }
        int err = 0;
        uint32_t Scup = 0, Pcup = 0;
        if (Lcup < 2) err = HT_ERR_INVALID;
        if (!err) {
            Scup = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((uint32_t)D[Lcup - 1] << 4) + (D[Lcup - 2] & 0x0F)));
            if (Scup < 2 || Scup > Lcup || Scup > 4079) err = HT_ERR_INVALID;
            Pcup = Lcup - Scup;
        }
        if (!err && maxbp >= 32) err = HT_ERR_INVALID;
        if (!err && ((Pcup * 8 + 31) / 32 + 3 > ms_words || ((b.w + 1) >> 1) > LPB)) err = HT_ERR_INVALID;
        if (!err && REFINE && b.npasses - num_plhd > 1 && (b.h > rg_rows || (b.lref && ht_nsp(b.lref) > mr_words))) err = HT_ERR_INVALID;   /* (host-sized) */
        if (err) {
            ht_zero_window(dst, b.w, b.h, b.stride, lane);
            if (lane == 0) status[bidx] = err;
            continue;
        }
        ok_s[hb] = true;
        Pcup_s[hb] = Pcup;
        Scup_s[hb] = Scup;
    }
    bool any_ok = false;
#pragma unroll
    for (int hb = 0; hb < NB; hb++) {
        if (!ok_s[hb]) continue;
        any_ok = true;
        const uint32_t Pcup = Pcup_s[hb];
        uint32_t *ms = ms_all + hb * mspitch;
        const uint32_t nms = (Pcup * 8 + 31) / 32 + 2;
        ht_ms_clear(ms, nms + 2, lane);
        __syncthreads();
        uint32_t ms_total = 0, carry = 0;
#pragma unroll
        for (int jx = 0; jx < PF; jx++)
            if (512u * jx < Pcup) ht_unstuff_magsgn_step8(pv[hb][jx], 64 * jx, Pcup, ms, lane, ms_total, carry);
        for (uint32_t p0 = 64 * PF; p0 * 8 < Pcup; p0 += 64) {
            const uint32_t pi = p0 + lane;
            ht_unstuff_magsgn_step8(pi * 8 < Pcup ? ((const uint2 *)Dw_s[hb])[pi] : make_uint2(0u, 0u), p0, Pcup, ms, lane, ms_total, carry);
        }
        __syncthreads();
        ht_ms_ones_tail(ms, ms_total, nms + 1, lane);        /* past the end the MagSgn stream is all ones (:207-221) */
        lastwi_s[hb] = nms - 2;                              /* words last_wi .. last_wi + 3 exist and are ones past the end */
    }
    __syncthreads();
    if (!any_ok) return;

    /* ---- all blocks in lockstep: this lane's block ---- */
    const J2kBlock b = blocks[bi];
    bool ok = false;
    uint32_t last_wi = 0;
#pragma unroll
    for (int hb = 0; hb < NB; hb++)
        if (seg == hb) { ok = ok_s[hb]; last_wi = lastwi_s[hb]; }
    const int w = b.w, h = b.h, stride = b.stride, M_b = b.M_b;
    const int qw = (w + 1) >> 1, qh = ok ? (h + 1) >> 1 : 0;
    const int rem = b.npasses % 3, num_plhd = rem ? b.npasses - rem : b.npasses - 3;
    const int S_blk = (num_plhd / 3 + b.zbp) & 0xFF;
    const int pLSB = (30 - S_blk) & 0xFF, maxbp = S_blk + 1, dshift = 31 - M_b;
    const uint32_t halfbit = 1u << ((pLSB - 1) & 31);
    float fscale = b.f_step;
    fscale /= (float)(1 << (31 - b.M_b));                    /* jpeg2000dec.c:2104-2106 */
    const int i_step = b.i_step;
    const uint32_t *ms = ms_all + seg * mspitch;
    const bool act = ok && q < qw;
    const bool c2 = 2 * q + 1 < w;                           /* the quad's right column is inside the block */
    const int qwp = (int)ht_qsym_pitch((uint32_t)w);
    const ht_sym_t *qp = qsym + qoff[bi] + q;
    uint32_t *prow = coef + b.plane_off + 2 * q;             /* this quad's two columns, row 2 * row */
    const int z_blk = b.npasses - num_plhd;
    const bool refined = REFINE && ok && z_blk > 1;          /* this lane's block carries SigProp (and with z_blk > 2 MagRef) decisions */
    uint32_t *mr = mr_all + seg * mr_words, *rg = rg_all + seg * 2 * MW * rg_rows;
    if (REFINE) {
        /* the blocks' MagRef bits (k_ht_unstuff: zero past either end) and SigProp masks (k_ht_refine) into LDS, every block by
         * its own lanes, all loads of the wave in flight together: the row loop then holds no load whose wait would also be
         * a wait for the stores in front of it */
        uint32_t Scup = 0;
#pragma unroll
        for (int hb = 0; hb < NB; hb++)
            if (seg == hb) Scup = Scup_s[hb];
        const uint32_t nsp = (refined && b.lref) ? ht_nsp(b.lref) : 0u;
        const uint32_t *src = mel_u + (b.data_off >> 2) + ht_nsw(Scup);
        for (uint32_t i = q; i < mr_words; i += LPB) mr[i] = i < nsp ? src[i] : 0u;
        const uint32_t *rsrc = (const uint32_t *)(refbits + roff[bi]);
        const uint32_t nrg = refined ? 2u * MW * (uint32_t)h : 0u;
        for (uint32_t i = q; i < nrg; i += LPB) {
            const uint32_t y = i / (2 * MW), k = (i / MW) & 1u, wd = i % MW;
            rg[i] = rsrc[((size_t)(3 * y + k) * HT_REF_STRIDE) * 2 + wd];
        }
        __syncthreads();
    }
    int rows = 0;
#pragma unroll
    for (int hb = 0; hb < NB; hb++) rows = max(rows, __builtin_amdgcn_readlane(qh, hb * LPB));
    /* odd widths: the last quad of a row has one column -- such waves store dwords, the others pairs */
    const bool pairs = __ballot(act && !c2) == 0;
    const bool pre = rows <= 32;                             /* (REFINE launches: always, the host admits blocks of at most 64 rows) */
    uint32_t qi_next = (!pre && act && qh > 0) ? *qp : 0u;
    uint32_t E1p = 0, E3p = 0, ms_pos = 0, mr_pos = 0, dB = 0;
    int err = 0;
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef u32x2 u32x2_a4 __attribute__((aligned(4)));
    /* one quad row; `qi` the lane's symbol, `qn` the symbol of the row below (REFINE: the MagRef bits of a stripe of four
     * sample rows are dealt out when its upper quad row is decoded) */
    /* refinement constants of this lane's block (bit-plane qq = pLSB - 1) */
    const uint32_t qq_ = (uint32_t)(pLSB - 1) & 31u;
    const uint32_t ref_new = (1u << qq_) | (1u << ((qq_ - 1) & 31u));
    const uint32_t ref_keep = z_blk > 2 ? 0xFFFFFFFEu << qq_ : 0xFFFFFFFFu, ref_half = z_blk > 2 ? 1u << ((qq_ - 1) & 31u) : 0u;
    auto quad_row = [&](int row, uint32_t qi, uint32_t qn) {
        const bool arow = act && row < qh;
        /* field n of the symbol -> bits 0-1 of byte n; R / K / X1: significant / exponent bound / MSB known, bit 0 of byte n */
        const uint32_t pk = qi & 0xFF, pk2 = pk | (pk << 12);
        const uint32_t F = (pk2 | (pk2 << 6)) & 0x03030303u, Fh = F >> 1, uq = (qi >> 8) & 0xFFu;
        const uint32_t R = (F | Fh) & 0x01010101u, K = Fh & 0x01010101u, X1 = F & Fh;
        int kappa = 1;
        if (row > 0) {
            uint32_t l = ht_dpp_left(E3p), r = ht_dpp_right(E1p);
            l = q == 0 ? 0u : l;
            r = q == LPB - 1 ? 0u : r;
            const int me = (int)max(max(E1p, E3p), max(l, r));
            kappa = (R & (R - 1)) ? max(me - 1, 1) : 1;
        }
        const uint32_t U = (uint32_t)kappa + uq;
        if (arow && (int)U > maxbp) err = 1;
        uint32_t Rs = R << 8;
        asm("" : "+v"(Rs));
        const uint32_t Rm = Rs - R;                                     /* 0xFF in the bytes of significant samples */
        const uint32_t U4 = __builtin_amdgcn_perm(U, U, 0x00000000u);   /* U in all four bytes (U < 256) */
        const uint32_t N = (U4 - K) & Rm;                               /* m_n = sigma_n * U - k_n, one byte each (:883-888) */
        const uint32_t tot = __builtin_amdgcn_sad_u8(N, 0u, 0u);
        const uint32_t incl = seg_incl_scan_u32<LPB>(tot);
        const uint32_t p0 = ms_pos + incl - tot;
        ms_pos += seg_last<LPB>(incl);
        const uint32_t n0 = N & 0xFF, n1 = (N >> 8) & 0xFF, n2 = (N >> 16) & 0xFF, n3 = N >> 24;
        const uint32_t p1 = p0 + n0, p2 = p1 + n1, p3 = p2 + n2;
        auto cut = [&](uint32_t p, uint32_t n) -> uint32_t {            /* n <= 31 stream bits from position p */
            const uint32_t i = min(p >> 5, last_wi + 2);
            return __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(ms[i + 1], ms[i], p), 0u, n);
        };
        uint32_t v0 = cut(p0, n0), v1 = cut(p1, n1), v2 = cut(p2, n2), v3 = cut(p3, n3);
        v0 += (X1 & 1) << n0; v1 += ((X1 >> 8) & 1) << n1; v2 += ((X1 >> 16) & 1) << n2; v3 += (X1 >> 24) << n3;
        const int s0m = -(int)(R & 1), s1m = -(int)((R >> 8) & 1), s2m = -(int)((R >> 16) & 1), s3m = -(int)(R >> 24);
        E1p = (uint32_t)(32 - __clz((int)(v1 | 1))) & (uint32_t)s1m;     /* bottom-left and bottom-right feed the next row */
        E3p = (uint32_t)(32 - __clz((int)(v3 | 1))) & (uint32_t)s3m;
        /* the refinement decisions of this quad's samples: bit 0 / 1 = column 2 q / 2 q + 1.  SigProp (newly significant, sign)
         * from k_ht_refine's masks in LDS; MagRef here: the pass visits a stripe of four sample rows column by column
         * (:1137-1185), one bit per sample the cleanup pass made significant, so a lane's eight samples of the stripe take
         * CONSECUTIVE bits, and where they start is a prefix sum of the significant counts over the lanes of the block */
        uint32_t Rt = 0, Gt = 0, Qt = 0, Rb = 0, Gb = 0, Qb = 0;
        if (REFINE) {
            if (!(row & 1)) {
                uint32_t rt = lut_rho[qi & 0xFFu], rbm = lut_rho[qn & 0xFFu];
                if (2 * row + 1 >= h) rt &= 5u;                            /* odd heights: the lower sample row of the last quad row is outside */
                if (2 * row + 3 >= h) rbm &= 5u;
                if (!(refined && z_blk > 2 && row < qh)) rt = 0;
                if (!(refined && z_blk > 2 && row + 1 < qh)) rbm = 0;
                /* stream order inside the lane: column 2 q rows 0..3, then column 2 q + 1 (quad samples: 0 1 | 2 3, row = n & 1) */
                const uint32_t c0 = (rt & 3u) | ((rbm & 3u) << 2), c1 = (rt >> 2) | (rbm & 0xCu);
                const uint32_t cnt = (uint32_t)__builtin_popcount(c0 | (c1 << 4));
                const uint32_t inc = seg_incl_scan_u32<LPB>(cnt);
                const uint32_t mp = mr_pos + inc - cnt;
                mr_pos += seg_last<LPB>(inc);
                const uint32_t wi = min(mp >> 5, mr_words - 2);
                const uint32_t bits = __builtin_amdgcn_alignbit(mr[wi + 1], mr[wi], mp);
                const uint32_t d0 = lut_dep[(c0 << 4) | (bits & 15u)];
                const uint32_t d1 = lut_dep[(c1 << 4) | ((bits >> __builtin_popcount(c0)) & 15u)];
                Qt = (d0 & 1u) | ((d1 & 1u) << 1);                         /* row 0 of the stripe */
                Qb = ((d0 >> 1) & 1u) | (d1 & 2u);                         /* row 1 */
                dB = (d0 >> 2) | ((d1 >> 2) << 2);                         /* rows 2, 3: the quad row below */
            } else {
                Qt = (dB & 1u) | ((dB >> 1) & 2u);
                Qb = ((dB >> 1) & 1u) | ((dB >> 2) & 2u);
            }
            if (refined && row < qh) {
                const uint32_t *r = rg + (size_t)(2 * row) * 2 * MW;
                if (MW == 1) {
                    Rt = r[0] >> (2 * q); Gt = r[1] >> (2 * q);
                    if (2 * row + 1 < h) { Rb = r[2] >> (2 * q); Gb = r[3] >> (2 * q); }
                } else {
                    Rt = (uint32_t)((((uint64_t)r[1] << 32) | r[0]) >> (2 * q)); Gt = (uint32_t)((((uint64_t)r[3] << 32) | r[2]) >> (2 * q));
                    if (2 * row + 1 < h) { Rb = (uint32_t)((((uint64_t)r[5] << 32) | r[4]) >> (2 * q)); Gb = (uint32_t)((((uint64_t)r[7] << 32) | r[6]) >> (2 * q)); }
                }
            }
        }
        auto sample = [&](uint32_t v, int sm, uint32_t nsig, uint32_t sgn, uint32_t mrb) -> uint32_t {   /* mu (:407-427) -> dequantisation */
            uint32_t mu = (((((v >> 1) + 1u) << pLSB) | halfbit) | (v << 31)) & (uint32_t)sm;
            if (REFINE) {
                /* SigProp made the sample significant: magnitude bit qq, the half bit below it, its sign (:1066-1100); MagRef
                 * (blocks with the third pass, samples the cleanup pass made significant): bit qq replaced by the pass's bit,
                 * the half bit moves down (:1160-1185).  Both as masks, no branch */
                mu |= ((sgn << 31) | ref_new) & (uint32_t)-(int)(nsig & 1u);
                const uint32_t keep = ((mrb & 1u) << qq_) | ref_keep | ~(uint32_t)sm;      /* all ones where MagRef does not apply */
                mu = (mu & keep) | (ref_half & (uint32_t)sm);
                if (TRANSFORM == J2K_DWT97) return __float_as_uint((float)(mu & 0x7FFFFFFFu) * fscale) | (mu & 0x80000000u);
                return ht_dequant(mu, TRANSFORM, M_b, 0, fscale, i_step);
            }
            if (TRANSFORM == J2K_DWT53) {
                const int sg = (int)mu >> 31;
                int r = (int)((mu & 0x7FFFFFFFu) >> dshift);
                r = (r ^ sg) - sg;
                if (i_step != 32768) {                                   /* block-uniform; reversible bands have step 1.0 */
                    const long long a = (long long)r * i_step;
                    r = (int)(a < 0 ? -((-a) >> 16) : (a >> 16));
                }
                return (uint32_t)r;
            } else if (TRANSFORM == J2K_DWT97) {
                /* (float)(-x) * s == -((float)x * s), and a zero magnitude never carries a sign */
                return __float_as_uint((float)(mu & 0x7FFFFFFFu) * fscale) | (mu & 0x80000000u);
            } else {
                return ht_dequant(mu, TRANSFORM, M_b, 0, fscale, i_step);
            }
        };
        const uint32_t o0 = sample(v0, s0m, Rt, Gt & 1, Qt);
        const uint32_t o1 = sample(v1, s1m, Rb, Gb & 1, Qb);
        const uint32_t o2 = sample(v2, s2m, Rt >> 1, (Gt >> 1) & 1, Qt >> 1);
        const uint32_t o3 = sample(v3, s3m, Rb >> 1, (Gb >> 1) & 1, Qb >> 1);
        const bool two = 2 * row + 1 < h;
        /* lanes and rows with nothing to write aim at a scratch line */
        if (pairs) {
            u32x2 t, bt;
            t.x = o0; t.y = o2; bt.x = o1; bt.y = o3;
            *(u32x2_a4 *)(arow ? prow : sink) = t;
            *(u32x2_a4 *)((arow && two) ? prow + stride : sink) = bt;
        } else {
            *(arow ? prow : sink) = o0;
            *((arow && c2) ? prow + 1 : sink) = o2;
            *((arow && two) ? prow + stride : sink) = o1;
            *((arow && two && c2) ? prow + stride + 1 : sink) = o3;
        }
        prow += 2 * stride;
    };
    if (pre) {                                               /* two rows per register, no load in the loop */
        for (int k = 0; 2 * k < rows; k++) {
            uint32_t s2 = sy[k];                             /* (uniform index) */
            s2 = (act && 2 * k < qh) ? s2 : 0u;              /* rows past this lane's block; a rejected block */
            s2 = (2 * k + 1 < qh) ? s2 : s2 & 0xFFFFu;
            quad_row(2 * k, s2 & 0xFFFFu, s2 >> 16);
            if (2 * k + 1 < rows) quad_row(2 * k + 1, s2 >> 16, 0u);
        }
    } else {
        for (int row = 0; row < rows; row++) {
            const uint32_t qi = qi_next;
            qp += qwp;
            qi_next = (act && row + 1 < qh) ? *qp : 0u;
            quad_row(row, qi, qi_next);
        }
    }
    /* a block whose U ran past maxbp is rejected as a whole (:862-868): zero it, whole wave per block */
#pragma unroll
    for (int hb = 0; hb < NB; hb++) {
        if (__ballot(err && seg == hb) == 0) continue;
        const int bidx = NB * (int)blockIdx.x + hb;
        const J2kBlock bb = blocks[bidx];
        __syncthreads();
        ht_zero_window(coef + bb.plane_off, bb.w, bb.h, bb.stride, lane);
        if (lane == 0) status[bidx] = HT_ERR_INVALID;
    }
}

/* EXTERNAL_VLC = false: the whole block in this kernel (stage 1 on lane 0).
 * EXTERNAL_VLC = true : stage 1 was done by k_ht_vlc (one LANE per codeblock, 64 serial decodes
 *                       per wavefront); the packed quad symbols come from `qsym` (qoff[b] is the
 *                       index of block b's first quad) and only MagSgn/refinement run here. */
template <bool EXTERNAL_VLC>
__global__ void __launch_bounds__(64)
k_ht_decode(const J2kBlock *__restrict__ blocks, int nblocks, const uint8_t *__restrict__ bytes,
            uint32_t *__restrict__ coef, const uint16_t *__restrict__ g_tables,
            int *__restrict__ status, HtLds L, const ht_sym_t *__restrict__ qsym, const uint32_t *__restrict__ qoff,
            uint32_t *__restrict__ sink, const uint64_t *__restrict__ refbits, const uint32_t *__restrict__ roff,
            int coef16 = 0)
{
    extern __shared__ __align__(16) uint8_t smem[];
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= nblocks) return;
    const J2kBlock b = blocks[blockIdx.x];
    const int w = b.w, h = b.h, stride = b.stride;
    const int qw = (w + 1) >> 1, qh = (h + 1) >> 1;
    const int transform = b.flags & 3;
    /* coef16 (host-checked: every block of the job is a reversible 5/3 cleanup-only block of at most 64 columns
     * with M_b <= 15 and no ROI shift): the planes are written as int16_t, same element offsets */
    uint16_t *dst16 = (uint16_t *)coef + b.plane_off;
    uint32_t *dst = coef16 ? (uint32_t *)dst16 : coef + b.plane_off;

    if (b.npasses == 0) {                              /* not coded: the reference plane is calloc'ed */
        if (coef16) ht_zero_window16(dst16, w, h, stride, lane); else ht_zero_window(dst, w, h, stride, lane);
        return;
    }
    /* pass bookkeeping, jpeg2000htdec.c:1240-1264 */
    const int rem = b.npasses % 3;
    const int num_plhd = rem ? b.npasses - rem : b.npasses - 3;
    const int p0 = num_plhd / 3;
    const int z_blk = b.npasses - num_plhd;
    const uint32_t Lcup = b.lcup, Lref = b.lref;
    const uint8_t *D = bytes + b.data_off;
    const int S_blk = (p0 + b.zbp) & 0xFF;
    const int pLSB = (30 - S_blk) & 0xFF;
    const int maxbp = S_blk + 1;                       /* (S_blk - 1) + 2, :605,:1263 */
    int err = 0;
    uint32_t Scup = 0, Pcup = 0;
    if (Lcup < 2) err = HT_ERR_INVALID;                /* :1252 */
    if (!err) {
        Scup = ((uint32_t)D[Lcup - 1] << 4) + (D[Lcup - 2] & 0x0F);
        if (Scup < 2 || Scup > Lcup || Scup > 4079) err = HT_ERR_INVALID;   /* :1268 */
        Pcup = Lcup - Scup;
    }
    if (!err && maxbp >= 32) err = HT_ERR_INVALID;     /* :617 */
    /* LDS capacity is sized by the host from the same fields; never index past it */
    if (!err && ((Pcup * 8 + 31) / 32 + 3 > L.ms_words || (uint32_t)qw > L.max_qw ||
                 (!EXTERNAL_VLC && ((Scup * 8 + 31) / 32 + 2 > L.vlc_words || Scup > L.suf_bytes))))
        err = HT_ERR_INVALID;
    if (err) {
        if (coef16) ht_zero_window16(dst16, w, h, stride, lane); else ht_zero_window(dst, w, h, stride, lane);
        if (lane == 0) status[blockIdx.x] = err;
        return;
    }

    uint16_t *tbl  = (uint16_t *)smem;
    uint32_t *ms   = (uint32_t *)(smem + L.off_ms);
    uint32_t *vlcw = (uint32_t *)(smem + L.off_vlc);
    uint8_t  *suf  = smem + L.off_suf;
    uint32_t *qinfo = (uint32_t *)(smem + L.off_qinfo);
    uint8_t  *Earr = smem + L.off_E;
    uint32_t *bm   = (uint32_t *)(smem + L.off_bm);
    const int Estride = 2 * (int)L.max_qw + 8;
    const uint32_t nms = (Pcup * 8 + 31) / 32 + 2, nvl = (Scup * 8 + 31) / 32 + 2;

    /* ---- stage 0: tables, zero, un-stuff ---- */
    if (!EXTERNAL_VLC) {
        for (int i = lane; i < 1024; i += 64)
            ((uint32_t *)tbl)[i] = ((const uint32_t *)g_tables)[i];
        for (uint32_t i = lane; i < nvl; i += 64) vlcw[i] = 0;
    }
    const bool fast = EXTERNAL_VLC && 2 * qw <= 64 && b.roi_shift == 0;   /* ht_magsgn_rows_narrow: no LDS exponent rows */
    for (uint32_t i = lane; i <= nms; i += 64) ms[i] = 0;
    if (!fast) {
        for (int i = lane; i < 2 * Estride; i += 64) Earr[i] = 0;
        for (uint32_t i = lane; i < 2 * L.max_qw; i += 64) qinfo[i] = 0;
    }
    if (z_blk > 1 && !fast)
        for (uint32_t i = lane; i < 4 * L.bm_words; i += 64) bm[i] = 0;
    __syncthreads();

    /* MagSgn: un-stuffed here in both modes, straight into LDS */
    const uint32_t ms_total = ht_unstuff_magsgn(D, Pcup, ms, lane);
    if (!EXTERNAL_VLC) {   /* VLC: backward from Dcup[Lcup-2]; Dcup[Lcup-1] counts as 0xFF and the low nibble of
         * Dcup[Lcup-2] as 0xF (:1277-1278); a byte with 7 LSBs set below a byte > 0x8F loses its MSB */
        uint32_t base = 0;
        const uint32_t nv = Scup - 1;                 /* bytes Lcup-2 .. Pcup */
        for (uint32_t k0 = 0; k0 < nv; k0 += 64) {
            const uint32_t k = k0 + lane;
            const bool act = k < nv;
            uint32_t v = 0, above = 0xFF;
            if (act) {
                const uint32_t j = Lcup - 2 - k;
                v = D[j];
                if (k == 0) v |= 0x0F;
                else { above = D[j + 1]; if (k == 1) above |= 0x0F; }
            }
            const uint32_t nb = act ? ((above > 0x8F && (v & 0x7F) == 0x7F) ? 7u : 8u) : 0u;
            v &= (1u << nb) - 1;
            const uint32_t incl = wave_incl_scan_u32(nb, lane);
            const uint32_t off = base + incl - nb;
            if (act) {
                atomicOr(&vlcw[off >> 5], v << (off & 31));
                if ((off & 31) > 24) atomicOr(&vlcw[(off >> 5) + 1], v >> (32 - (off & 31)));
            }
            base += wave_last(incl);
        }
    }
    for (uint32_t i = lane; !EXTERNAL_VLC && i < Scup; i += 64) {       /* MEL reads the (patched) suffix bytes */
        uint32_t v = D[Pcup + i];
        if (Pcup + i == Lcup - 1) v = 0xFF;
        else if (Pcup + i == Lcup - 2) v |= 0x0F;
        suf[i] = (uint8_t)v;
    }
    __syncthreads();
    for (uint32_t i = lane; i <= nms; i += 64) {                        /* past the end the MagSgn stream is all ones */
        if (i * 32 >= ms_total) ms[i] = 0xFFFFFFFFu;
        else if (i * 32 + 32 > ms_total) ms[i] |= 0xFFFFFFFFu << (ms_total & 31);
    }
    __syncthreads();

    HtSerial S;
    S.vbuf = 0; S.vbits = 0; S.vword = 0; S.vlc = vlcw; S.vlc_words = nvl;
    S.suf = suf; S.mel_pos = 0; S.mel_len = Scup; S.mel_tmp = 0; S.mel_bits = 0;
    S.mel_k = 0; S.mel_run = 0; S.mel_one = 0;
    if (!EXTERNAL_VLC && lane == 0) S.vdrop(0), S.vfill(), S.vdrop(4);   /* jpeg2000_init_vlc drops the Scup nibble, :283-295 */
    const ht_sym_t *qglob = EXTERNAL_VLC ? qsym + qoff[blockIdx.x] : nullptr;

    float fscale = b.f_step;
    fscale /= (float)(1 << (31 - b.M_b));              /* jpeg2000dec.c:2104-2106 */
    const int i_step = b.i_step, M_b = b.M_b, roi_shift = b.roi_shift;

    uint32_t ms_pos = 0;                               /* bit position in the un-stuffed MagSgn stream */
    int ctx_run = 0;                                   /* first-row context carried along the row */
    const int bmW = w + 2;                             /* bitmap row pitch (1-cell border) */

    const uint64_t *rbits = (fast && z_blk > 1 && refbits) ? refbits + roff[blockIdx.x] : nullptr;
    if (fast) {
        if (z_blk > 1) {
            if (transform == J2K_DWT53) err = ht_magsgn_rows_narrow<J2K_DWT53, true>(qglob, ms, dst, lane, w, h, stride, pLSB, maxbp, M_b, fscale, i_step, nms - 2, sink, rbits, z_blk);
            else if (transform == J2K_DWT97) err = ht_magsgn_rows_narrow<J2K_DWT97, true>(qglob, ms, dst, lane, w, h, stride, pLSB, maxbp, M_b, fscale, i_step, nms - 2, sink, rbits, z_blk);
            else err = ht_magsgn_rows_narrow<J2K_DWT97_INT, true>(qglob, ms, dst, lane, w, h, stride, pLSB, maxbp, M_b, fscale, i_step, nms - 2, sink, rbits, z_blk);
        } else if (EXTERNAL_VLC && coef16) {
            err = ht_magsgn_rows_narrow<J2K_DWT53, false, true>(qglob, ms, dst, lane, w, h, stride, pLSB, maxbp, M_b, fscale, i_step, nms - 2, sink, rbits, z_blk);
        } else {
            if (transform == J2K_DWT53) err = ht_magsgn_rows_narrow<J2K_DWT53, false>(qglob, ms, dst, lane, w, h, stride, pLSB, maxbp, M_b, fscale, i_step, nms - 2, sink, rbits, z_blk);
            else if (transform == J2K_DWT97) err = ht_magsgn_rows_narrow<J2K_DWT97, false>(qglob, ms, dst, lane, w, h, stride, pLSB, maxbp, M_b, fscale, i_step, nms - 2, sink, rbits, z_blk);
            else err = ht_magsgn_rows_narrow<J2K_DWT97_INT, false>(qglob, ms, dst, lane, w, h, stride, pLSB, maxbp, M_b, fscale, i_step, nms - 2, sink, rbits, z_blk);
        }
    }
    for (int row = 0; row < qh && !err && !fast; row++) {
        uint32_t *qcur = qinfo + (row & 1) * L.max_qw, *qprev = qinfo + ((row & 1) ^ 1) * L.max_qw;
        uint8_t *Ecur = Earr + (row & 1) * Estride + 4, *Eprev = Earr + ((row & 1) ^ 1) * Estride + 4;

        /* ---- stage 1: serial quad-row decode on lane 0 ---- */
        if (EXTERNAL_VLC) {
            /* symbols are read straight from global in stage 2 */
        } else if (lane == 0) {
            const uint16_t *table = tbl + (row ? 1024 : 0);
            int rho_left = 0;
            for (int qx = 0; qx < qw; qx += 2) {
                const int npair = (qx + 1 < qw) ? 2 : 1;
                int rho[2] = { 0, 0 }, uoff[2] = { 0, 0 }, ek[2] = { 0, 0 }, e1[2] = { 0, 0 }, u[2] = { 0, 0 };
                for (int k = 0; k < npair; k++) {
                    const int q = qx + k;
                    int ctx;
                    if (row == 0) {
                        ctx = ctx_run;
                    } else {
                        const int ra  = qprev[q] & 0xF;
                        const int ral = q > 0 ? (qprev[q - 1] & 0xF) : 0;
                        const int rar = q + 1 < qw ? (qprev[q + 1] & 0xF) : 0;
                        const int rl  = q > 0 ? rho_left : 0;
                        ctx = (((ra >> 1) | (ral >> 3)) & 1) | ((((rl >> 2) | (rl >> 3)) & 1) << 1) |
                              ((((ra >> 3) | (rar >> 1)) & 1) << 2);
                    }
                    if (ctx != 0 || S.mel_sym() != 0) {
                        const uint32_t e = table[(ctx << 7) | S.vpeek(7)];
                        S.vdrop((e >> 1) & 7);
                        uoff[k] = e & 1; rho[k] = (e >> 4) & 0xF; ek[k] = (e >> 8) & 0xF; e1[k] = (e >> 12) & 0xF;
                    }
                    rho_left = rho[k];
                    if (row == 0)
                        ctx_run = ((rho[k] | (rho[k] >> 1)) & 1) | (((rho[k] >> 2) & 1) << 1) | (((rho[k] >> 3) & 1) << 2);
                }
                if (npair == 2 && uoff[0] && uoff[1]) {
                    if (row == 0) {
                        if (S.mel_sym()) {
                            const int p1 = S.upfx(), p2 = S.upfx();
                            const int s1 = S.usfx(p1), s2 = S.usfx(p2);
                            const int x1 = S.uext(s1), x2 = S.uext(s2);
                            u[0] = 2 + p1 + s1 + 4 * x1;
                            u[1] = 2 + p2 + s2 + 4 * x2;
                        } else {
                            const int p1 = S.upfx();
                            if (p1 > 2) {
                                u[1] = (int)S.vget(1) + 1;
                                const int s1 = S.usfx(p1), x1 = S.uext(s1);
                                u[0] = p1 + s1 + 4 * x1;
                            } else {
                                const int p2 = S.upfx();
                                const int s1 = S.usfx(p1), s2 = S.usfx(p2);
                                const int x1 = S.uext(s1), x2 = S.uext(s2);
                                u[0] = p1 + s1 + 4 * x1;
                                u[1] = p2 + s2 + 4 * x2;
                            }
                        }
                    } else {
                        const int p1 = S.upfx(), p2 = S.upfx();
                        const int s1 = S.usfx(p1), s2 = S.usfx(p2);
                        const int x1 = S.uext(s1), x2 = S.uext(s2);
                        u[0] = p1 + s1 + 4 * x1;
                        u[1] = p2 + s2 + 4 * x2;
                    }
                } else {
                    for (int k = 0; k < npair; k++)
                        if (uoff[k]) {
                            const int p = S.upfx(), s = S.usfx(p), x = S.uext(s);
                            u[k] = p + s + 4 * x;
                        }
                }
                for (int k = 0; k < npair; k++)
                    qcur[qx + k] = (uint32_t)rho[k] | ((uint32_t)ek[k] << 4) | ((uint32_t)e1[k] << 8) | ((uint32_t)u[k] << 16);
            }
        }
        if (!EXTERNAL_VLC) __syncthreads();

        /* ---- stage 2: MagSgn of the quad row, lanes = sample columns ---- */
        const int ncols = 2 * qw;
        int row_err = 0;
        for (int c0 = 0; c0 < ncols; c0 += 64) {
            const int col = c0 + lane;
            const bool act = col < ncols;
            const int q = col >> 1;
            const uint32_t qi = act ? (EXTERNAL_VLC ? ht_sym_unpack(qglob[row * (int)ht_qsym_pitch((uint32_t)w) + q]) : qcur[q]) : 0;
            const int rho = qi & 0xF, ekq = (qi >> 4) & 0xF, e1q = (qi >> 8) & 0xF, uq = (qi >> 16) & 0xFF;
            int kappa = 1;
            if (row > 0 && act) {                      /* :855-885; Eprev[-1] and Eprev[ncols] are 0 */
                const int x2 = 2 * q;
                int me = max(max((int)Eprev[x2 - 1], (int)Eprev[x2]), max((int)Eprev[x2 + 1], (int)Eprev[x2 + 2]));
                const int gamma = (rho & (rho - 1)) != 0;           /* more than one significant sample */
                kappa = max(1, gamma * (me - 1));
            }
            const int U = kappa + uq;
            if (act && U > maxbp) row_err = 1;         /* :715,756,889,961 */
            const int sh = (col & 1) * 2;              /* samples 0,1 (left column) or 2,3 (right column) */
            const int s_t = (rho >> sh) & 1, s_b = (rho >> (sh + 1)) & 1;
            int m_t = act ? s_t * U - ((ekq >> sh) & 1) : 0;
            int m_b = act ? s_b * U - ((ekq >> (sh + 1)) & 1) : 0;
            /* a negative m (e_k outside rho: not in the Annex C tables) reads no bits */
            const uint32_t nb = (uint32_t)(max(m_t, 0) + max(m_b, 0));
            const uint32_t incl = wave_incl_scan_u32(nb, lane);
            uint32_t pos = ms_pos + incl - nb;
            uint32_t smag[2] = { 0, 0 };
            int Ebot = 0;
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int m = r ? m_b : m_t;
                const int e1b = (e1q >> (sh + r)) & 1;
                if (m != 0) {
                    uint32_t v = 0;
                    if (m > 0) {
                        const uint32_t wi = min(pos >> 5, nms - 1), wj = min((pos >> 5) + 1, nms - 1);
                        const uint64_t two = ((uint64_t)ms[wj] << 32) | ms[wi];
                        v = (uint32_t)(two >> (pos & 31)) & (uint32_t)((1ull << m) - 1);
                        v += (uint32_t)e1b << m;
                        pos += m;
                    }
                    const int E = 32 - __clz((int)(v | 1));
                    uint32_t mu = (v >> 1) + 1;
                    mu <<= pLSB;
                    mu |= 1u << ((pLSB - 1) & 31);
                    mu |= (v & 1) << 31;
                    smag[r] = mu;
                    if (r) Ebot = E;
                }
            }
            if (act) Ecur[col] = (uint8_t)Ebot;
            ms_pos += wave_last(incl);

            /* raster positions of this lane's two samples; odd sizes: the outside half of the
             * border quads is discarded (:976-1007) */
            const int y0 = 2 * row;
            if (act && col < w) {
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const int y = y0 + r;
                    if (y >= h) continue;
                    if (z_blk > 1) {
                        dst[(size_t)y * stride + col] = smag[r];          /* raw, finished in stage 3 */
                        if (((rho >> (sh + r)) & 1))
                            atomicOr(&bm[((y + 1) * bmW + col + 1) >> 5], 1u << (((y + 1) * bmW + col + 1) & 31));
                    } else {
                        dst[(size_t)y * stride + col] = ht_dequant(smag[r], transform, M_b, roi_shift, fscale, i_step);
                    }
                }
            }
        }
        if (__any(row_err)) err = HT_ERR_INVALID;
        if (EXTERNAL_VLC) wave_lds_fence(); else __syncthreads();
    }

    if (err) {
        __syncthreads();
        if (coef16) ht_zero_window16(dst16, w, h, stride, lane); else ht_zero_window(dst, w, h, stride, lane);
        if (lane == 0) status[blockIdx.x] = err;
        return;
    }
    if (z_blk <= 1 || fast) return;                    /* the fast path applied k_ht_refine's decisions in its row loop */

    /* ---- stage 3: refinement passes on bitmaps (bit index = (y+1)*(w+2) + x+1) ---- */
    uint32_t *bm_sig = bm, *bm_ref = bm + L.bm_words, *bm_sgn = bm + 2 * L.bm_words, *bm_mr = bm + 3 * L.bm_words;
    const uint8_t *Dref = D + Lcup;
    const int causal = b.flags & J2K_CBLK_VSC;
    __syncthreads();
    if (lane == 0) {
        /* SigProp, jpeg2000htdec.c:1016-1131: forward LSB-first reader, 7 bits after 0xFF, zeros past Lref */
        uint32_t pos = 0, tmp = 0, last = 0; int bits = 0;
        auto rdbit = [&]() -> int {
            if (bits == 0) {
                bits = (last == 0xFF) ? 7 : 8;
                if (pos < Lref) { tmp = Dref[pos]; pos++; } else tmp = 0;
                last = tmp;
            }
            int bit = tmp & 1; tmp >>= 1; bits--;
            return bit;
        };
        for (int i0 = 0; i0 < h; i0 += 4) {
            const int gh = min(4, h - i0);
            for (int j0 = 0; j0 < w; j0 += 4) {
                const int gw = min(4, w - j0);
                for (int j = j0; j < j0 + gw; j++)
                    for (int i = i0; i < i0 + gh; i++) {
                        const int c = (i + 1) * bmW + (j + 1);
                        if (bm_get(bm_sig, c)) continue;
                        const int below_ok = !causal || (i != i0 + gh - 1);
                        int mbr = bm_get(bm_sig, c - bmW - 1) | bm_get(bm_sig, c - bmW) | bm_get(bm_sig, c - bmW + 1) |
                                  bm_get(bm_sig, c - 1) | bm_get(bm_sig, c + 1) |
                                  bm_get(bm_ref, c - bmW - 1) | bm_get(bm_ref, c - bmW) | bm_get(bm_ref, c - bmW + 1) |
                                  bm_get(bm_ref, c - 1) | bm_get(bm_ref, c + 1);
                        if (below_ok)
                            mbr |= bm_get(bm_sig, c + bmW - 1) | bm_get(bm_sig, c + bmW) | bm_get(bm_sig, c + bmW + 1) |
                                   bm_get(bm_ref, c + bmW - 1) | bm_get(bm_ref, c + bmW) | bm_get(bm_ref, c + bmW + 1);
                        if (mbr && rdbit())
                            bm_ref[c >> 5] |= 1u << (c & 31);
                    }
                for (int j = j0; j < j0 + gw; j++)
                    for (int i = i0; i < i0 + gh; i++) {
                        const int c = (i + 1) * bmW + (j + 1);
                        if (bm_get(bm_ref, c) && rdbit())
                            bm_sgn[c >> 5] |= 1u << (c & 31);
                    }
            }
        }
        if (z_blk > 2) {
            /* MagRef, :1137-1185: backward reader over Dref with the VLC un-stuffing rule; the byte
             * after the segment counts as 0xFF (:1260), bytes below Dref[0] read as zero bits */
            int rpos = (int)Lref - 1; uint32_t above = 0xFF, cur = 0; int nb = 0;
            auto rdback = [&]() -> int {
                if (nb == 0) {
                    if (rpos >= 0) {
                        cur = Dref[rpos];
                        nb = (above > 0x8F && (cur & 0x7F) == 0x7F) ? 7 : 8;
                        above = cur;
                        rpos--;
                    } else { cur = 0; nb = 8; }
                }
                int bit = cur & 1; cur >>= 1; nb--;
                return bit;
            };
            for (int i0 = 0; i0 < h; i0 += 4)
                for (int j = 0; j < w; j++)
                    for (int i = i0; i < min(i0 + 4, h); i++) {
                        const int c = (i + 1) * bmW + (j + 1);
                        if (bm_get(bm_sig, c) && rdback())
                            bm_mr[c >> 5] |= 1u << (c & 31);
                    }
        }
    }
    __syncthreads();
    {
        const int qq = (pLSB - 1) & 31;                /* both passes are called with pLSB - 1, :1309-1315 */
        for (int y = 0; y < h; y++)
            for (int x = lane; x < w; x += 64) {
                const int c = (y + 1) * bmW + (x + 1);
                uint32_t v = dst[(size_t)y * stride + x];
                if (bm_get(bm_ref, c)) {
                    v |= 1u << qq;
                    v |= 1u << ((qq - 1) & 31);
                    v |= (uint32_t)bm_get(bm_sgn, c) << 31;
                }
                if (z_blk > 2 && bm_get(bm_sig, c)) {
                    v &= (0xFFFFFFFEu | (uint32_t)bm_get(bm_mr, c)) << qq;
                    v |= 1u << ((qq - 1) & 31);
                }
                dst[(size_t)y * stride + x] = ht_dequant(v, transform, M_b, roi_shift, fscale, i_step);
            }
    }
}


/* ================================================================== split pipeline
 * k_ht_unstuff  (wave per block)  removes the bit stuffing of the VLC byte stream of the cleanup
 *               segment in parallel (four bytes per lane: per-byte bit counts, wave prefix sum,
 *               ds_or) and writes it as a plain bit array: LSB-first in read order padded with
 *               zeros (jpeg2000htdec.c:145-201, first 4 bits = the Scup nibble).  (MagSgn is
 *               un-stuffed by k_ht_decode itself, straight into LDS; the MEL stream -- short,
 *               but of unknown length -- is read from the raw bytes by k_ht_vlc, :429-440.)
 * k_ht_vlc      (LANE per block)  the serial MEL / CxtVLC / U-VLC chain (:632-973); 64 blocks
 *               per wavefront.  With stuffing gone a refill is "append the next dword", a MEL
 *               read is a shift, and the first-row / other-row / paired / unpaired U-VLC cases
 *               are one branch-free sequence -- divergent lanes cost every path, so the loop
 *               body is kept small rather than fast-pathed.
 * k_ht_decode<true> (wave per block)  MagSgn + dequantisation + refinement passes.
 * The host sorts the block table by size so that the 64 lanes of a k_ht_vlc wave run similar
 * trip counts. */
/* Backward byte stream (VLC, MagRef): bytes top[0], top[-1], ... top[-(n-1)] in read order, eight per lane and
 * pass.  A byte whose 7 LSBs are set loses its MSB when the byte read before it is > 0x8F
 * (jpeg2000htdec.c:145-201); the byte "before" the first one is 0xFF (carry starts as 0x80: "was > 0x8F").
 * `first_or` is ORed into the first byte (the VLC stream's Dcup[Lcup-2] counts with its low nibble set,
 * :1277-1278).  `dq` = the eight memory bytes top[-k-7 .. -k], k = k0 + 8 lane: byte-swapped they are the lane's
 * bytes in read order.  `out` (LDS) is zeroed. */
__device__ __forceinline__ void ht_unstuff_backward_step8(uint2 dq, uint32_t k0, uint32_t n, uint32_t first_or,
                                                          uint32_t *out, int lane, uint32_t &base, uint32_t &carry)
{
    const uint32_t k = k0 + 8 * lane;
    uint32_t lo = __builtin_bswap32(dq.y), hi = __builtin_bswap32(dq.x);
    if (k == 0) lo |= first_or;
    int nv = 8;
    if (k0 + 512 > n) {                                  /* wave-uniform: the pass that holds the end of the stream */
        nv = min(max((int)n - (int)k, 0), 8);
        const int nl = min(nv, 4), nh = nv - nl;
        lo = nl == 4 ? lo : (nl ? lo & (0xFFFFFFFFu >> (32 - 8 * nl)) : 0u);
        hi = nh == 4 ? hi : (nh ? hi & (0xFFFFFFFFu >> (32 - 8 * nh)) : 0u);
    }
    /* A: byte > 0x8F (top bit set and the low seven >= 0x10); B: low seven bits all set -- flags at bit 7 of each byte */
    const uint32_t ll = lo & 0x7F7F7F7Fu, lh = hi & 0x7F7F7F7Fu;
    const uint32_t Al = (ll + 0x70707070u) & lo & 0x80808080u, Ah = (lh + 0x70707070u) & hi & 0x80808080u;
    const uint32_t Bl = (ll + 0x01010101u) & 0x80808080u, Bh = (lh + 0x01010101u) & 0x80808080u;
    uint32_t prevA = ht_dpp_left(Ah >> 24);              /* the flag of the byte read just before this lane's first */
    if (lane == 0) prevA = carry;
    carry = (uint32_t)__builtin_amdgcn_readlane((int)(Ah >> 24), 63);
    uint32_t fl = Bl & ((Al << 8) | prevA), fh = Bh & __builtin_amdgcn_alignbyte(Ah, Al, 3);
    if (nv < 8) {                                        /* bytes past the stream are zero: B is clear there already */
        const int nl = min(nv, 4), nh = nv - nl;
        fl = nl == 4 ? fl : (nl ? fl & (0xFFFFFFFFu >> (32 - 8 * nl)) : 0u);
        fh = nh == 4 ? fh : (nh ? fh & (0xFFFFFFFFu >> (32 - 8 * nh)) : 0u);
    }
    ht_squeeze_place<false>(lo, hi, fl, fh, nv, out, lane, base);
}

__device__ __forceinline__ uint32_t ht_unstuff_backward(const uint8_t *__restrict__ top, uint32_t n, uint32_t first_or,
                                                        uint32_t *out, int lane)
{
    uint32_t base = 0, carry = 0x80;
    for (uint32_t k0 = 0; k0 < n; k0 += 512) {
        const uint32_t k = k0 + 8 * lane;
        uint2 dq = make_uint2(0u, 0u);
        if (k < n) __builtin_memcpy(&dq, top - k - 7, 8);      /* up to 7 bytes in front of the stream: block bytes or pad */
        ht_unstuff_backward_step8(dq, k0, n, first_or, out, lane, base, carry);
    }
    return base;
}

/* Forward LSB-first stream whose bytes after 0xFF carry 7 bits, the MSB dropped (SigProp,
 * jpeg2000htdec.c:1016-1131): bytes src[0..n), any alignment.  `out` (LDS) is zeroed. */
__device__ __forceinline__ uint32_t ht_unstuff_forward7(const uint8_t *__restrict__ src, uint32_t n, uint32_t *out, int lane)
{
    uint32_t base = 0, carry = 0;
    for (uint32_t p0 = 0; p0 * 8 < n; p0 += 64) {
        const uint32_t i = (p0 + lane) * 8;
        uint2 dq = make_uint2(0u, 0u);
        if (i < n) __builtin_memcpy(&dq, src + i, 8);           /* up to 7 bytes behind the stream: block bytes or pad */
        ht_unstuff_fwd_step8<false>(dq, p0, n, out, lane, base, carry);
    }
    return base;
}


/* ---- the same un-stuffing with SEVERAL blocks per wave (k_ht_unstuff_g<LPB>: LPB lanes per block, 64 / LPB blocks) ----
 * A wave per block spends its fixed costs -- set-up, zeroing and copying loops, a prefix sum and three LDS ORs per pass of
 * 512 bytes -- on a VLC stream of a few hundred bytes (64 x 64 blocks) or a few dozen (32 x 32) that fills a fraction of
 * one pass.  Here 16 (or 32) lanes take a block: a pass covers 128 (256) contiguous bytes of each of four (two) blocks, the
 * prefix sum stops at the group (the first four, five, of the six DPP steps), the carries from one pass to the next and
 * the group's bit total travel by ds_bpermute from the group's last lane.  Accesses stay contiguous per group (a lane per
 * block does not pay: DESIGN.md section 3.1).  Blocks are sorted by size, so the groups of a wave end together. */
template <int LPB>
__device__ __forceinline__ uint32_t grp_incl_scan_u32(uint32_t v)
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    if (LPB >= 32) x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    if (LPB >= 64) x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return (uint32_t)x;
}
template <int LPB>
__device__ __forceinline__ uint32_t grp_last(uint32_t v, int lane)      /* the value the last lane of this lane's group holds */
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute((lane | (LPB - 1)) << 2, (int)v);
}
template <bool OR_SEM, int LPB>
__device__ __forceinline__ void ht_squeeze_place_g(uint32_t lo, uint32_t hi, uint32_t fl, uint32_t fh, int nv,
                                                   uint32_t *out, int lane, uint32_t &base)
{
    const uint32_t tot = 8u * (uint32_t)nv - (uint32_t)__builtin_popcount(fl) - (uint32_t)__builtin_popcount(fh);
    while (__ballot(fh != 0) != 0) {
        if (fh) {
            const uint32_t k = 31u - (uint32_t)__builtin_clz(fh), bit = 1u << k, m = bit - 1u;
            uint32_t nh = (hi & m) | ((hi >> 1) & ~m);
            if (OR_SEM) nh |= hi & bit;
            hi = nh;
            fh &= m;
        }
    }
    while (__ballot(fl != 0) != 0) {
        if (fl) {
            const uint32_t k = 31u - (uint32_t)__builtin_clz(fl), bit = 1u << k, m = bit - 1u;
            uint32_t nl = (lo & m) | (__builtin_amdgcn_alignbit(hi, lo, 1) & ~m);
            if (OR_SEM) nl |= lo & bit;
            lo = nl;
            hi >>= 1;
            fl &= m;
        }
    }
    const uint32_t incl = grp_incl_scan_u32<LPB>(tot);
    const uint32_t off = base + incl - tot;
    if (nv > 0) {
        const uint32_t sh = off & 31;
        const uint64_t t = (uint64_t)lo << sh, u = (uint64_t)hi << sh;
        uint32_t *o = out + (off >> 5);
        atomicOr(o, (uint32_t)t);
        atomicOr(o + 1, (uint32_t)(t >> 32) | (uint32_t)u);
        if (sh + tot + (OR_SEM ? 1u : 0u) > 64) atomicOr(o + 2, (uint32_t)(u >> 32));
    }
    base += grp_last<LPB>(incl, lane);
}
__device__ __forceinline__ uint32_t ht_first_bytes(uint32_t v, int n) { return n >= 4 ? v : (n > 0 ? v & (0xFFFFFFFFu >> (32 - 8 * n)) : 0u); }
/* eight bytes of a backward stream per lane: `dq` = the memory bytes top[-k-7 .. -k] of the lane's block, k = k0 + 8 gl
 * (gl = lane within the group), `n` = the stream's length (0: the group has nothing to do), `carry` = "the byte before this
 * pass was > 0x8F" at bit 7, per group */
template <int LPB>
__device__ __forceinline__ void ht_unstuff_backward_step8_g(uint2 dq, uint32_t k0, uint32_t n, uint32_t first_or,
                                                            uint32_t *out, int lane, uint32_t &base, uint32_t &carry)
{
    const int gl = lane & (LPB - 1);
    const uint32_t k = k0 + 8 * gl;
    const int nv = min(max((int)n - (int)k, 0), 8);
    uint32_t lo = __builtin_bswap32(dq.y), hi = __builtin_bswap32(dq.x);
    if (k == 0) lo |= first_or;
    lo = ht_first_bytes(lo, nv); hi = ht_first_bytes(hi, nv - 4);
    const uint32_t ll = lo & 0x7F7F7F7Fu, lh = hi & 0x7F7F7F7Fu;
    const uint32_t Al = (ll + 0x70707070u) & lo & 0x80808080u, Ah = (lh + 0x70707070u) & hi & 0x80808080u;
    const uint32_t Bl = (ll + 0x01010101u) & 0x80808080u, Bh = (lh + 0x01010101u) & 0x80808080u;
    uint32_t prevA = ht_dpp_left(Ah >> 24);
    if (gl == 0) prevA = carry;
    carry = grp_last<LPB>(Ah >> 24, lane);
    const uint32_t fl = ht_first_bytes(Bl & ((Al << 8) | prevA), nv), fh = ht_first_bytes(Bh & __builtin_amdgcn_alignbyte(Ah, Al, 3), nv - 4);
    ht_squeeze_place_g<false, LPB>(lo, hi, fl, fh, nv, out, lane, base);
}
/* ... of a forward stream whose bytes behind an 0xFF lose their top bit (SigProp): dq = bytes src[i .. i + 8), i = p0 + 8 gl */
template <int LPB>
__device__ __forceinline__ void ht_unstuff_fwd7_step8_g(uint2 dq, uint32_t p0, uint32_t n, uint32_t *out, int lane,
                                                        uint32_t &base, uint32_t &carry)
{
    const int gl = lane & (LPB - 1);
    const int nv = min(max((int)n - (int)(p0 + 8 * gl), 0), 8);
    const uint32_t lo = ht_first_bytes(dq.x, nv), hi = ht_first_bytes(dq.y, nv - 4);
    uint32_t prev = ht_dpp_left(hi >> 24);
    if (gl == 0) prev = carry;
    carry = grp_last<LPB>(hi >> 24, lane);
    const uint32_t Pl = (lo << 8) | prev, Ph = __builtin_amdgcn_alignbyte(hi, lo, 3);
    const uint32_t fl = ht_first_bytes(((Pl & 0x7F7F7F7Fu) + 0x01010101u) & Pl & 0x80808080u, nv);
    const uint32_t fh = ht_first_bytes(((Ph & 0x7F7F7F7Fu) + 0x01010101u) & Ph & 0x80808080u, nv - 4);
    ht_squeeze_place_g<false, LPB>(lo, hi, fl, fh, nv, out, lane, base);
}

template <int LPB>
__global__ void __launch_bounds__(64)
k_ht_unstuff_g(const J2kBlock *__restrict__ blocks, int nblocks, const uint8_t *__restrict__ bytes,
               uint32_t *__restrict__ vlc_u, uint32_t *__restrict__ mel_u, uint32_t lds_words)
{
    extern __shared__ __align__(16) uint32_t sw[];
    constexpr int G = 64 / LPB;
    const int lane = threadIdx.x, gl = lane & (LPB - 1);
    const int bi = (int)blockIdx.x * G + lane / LPB;
    J2kBlock b;
    memset(&b, 0, sizeof(b));
    if (bi < nblocks) b = blocks[bi];
    const uint8_t *D = bytes + b.data_off;
    const uint32_t Lcup = b.lcup;
    /* (the blocks k_ht_unstuff leaves alone; nothing returns early: the wave's groups go through the barriers together) */
    bool ok = bi < nblocks && b.npasses != 0 && Lcup >= 2;
    uint2 pv[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const long long o = (long long)b.data_off + Lcup - 9 - (8 * LPB * j + 8 * gl);
        pv[j] = make_uint2(0u, 0u);
        if (ok && o >= 0) __builtin_memcpy(&pv[j], bytes + o, 8);
    }
    uint32_t Scup = 0;
    if (ok) Scup = ((uint32_t)D[Lcup - 1] << 4) + (D[Lcup - 2] & 0x0F);
    ok = ok && !(Scup < 2 || Scup > Lcup || Scup > 4079);
    const uint32_t nsw = ht_nsw(Scup);
    ok = ok && 2 * nsw <= lds_words;
    uint32_t *vlO = vlc_u + (b.data_off >> 2), *meO = mel_u + (b.data_off >> 2);
    uint32_t *sv = sw + (size_t)(lane / LPB) * lds_words;

    if (ok) for (uint32_t i = gl; i < nsw; i += LPB) sv[i] = 0;
    __syncthreads();
    {
        const uint8_t *top = D + Lcup - 2;
        const uint32_t n = ok ? Scup - 1 : 0u;
        uint32_t base = 0, carry = 0x80;
#pragma unroll
        for (int j = 0; j < 2; j++)
            if (__ballot((uint32_t)(8 * LPB * j) < n) != 0) ht_unstuff_backward_step8_g<LPB>(pv[j], 8 * LPB * j, n, 0x0F, sv, lane, base, carry);
        for (uint32_t k0 = 16 * LPB; __ballot(k0 < n) != 0; k0 += 8 * LPB) {
            const uint32_t k = k0 + 8 * gl;
            uint2 dq = make_uint2(0u, 0u);
            if (k < n) __builtin_memcpy(&dq, top - k - 7, 8);
            ht_unstuff_backward_step8_g<LPB>(dq, k0, n, 0x0F, sv, lane, base, carry);
        }
    }
    __syncthreads();
    if (ok) for (uint32_t i = gl; i < nsw; i += LPB) vlO[i] = sv[i];

    const int rem = b.npasses % 3, plhd = rem ? b.npasses - rem : b.npasses - 3;
    const uint32_t Lref = b.lref, nsp = ht_nsp(Lref);
    const bool ref = ok && b.npasses - plhd > 1 && Lref > 0 && 2 * nsp <= lds_words;
    if (__ballot(ref) == 0) return;                      /* (the whole wave) */
    uint32_t *ss = sv, *sr = sv + nsp;
    __syncthreads();
    if (ref) for (uint32_t i = gl; i < 2 * nsp; i += LPB) sv[i] = 0;
    __syncthreads();
    {
        const uint32_t n = ref ? Lref : 0u;
        uint32_t base = 0, carry = 0;
        for (uint32_t p0 = 0; __ballot(p0 < n) != 0; p0 += 8 * LPB) {
            const uint32_t i = p0 + 8 * gl;
            uint2 dq = make_uint2(0u, 0u);
            if (i < n) __builtin_memcpy(&dq, D + Lcup + i, 8);
            ht_unstuff_fwd7_step8_g<LPB>(dq, p0, n, ss, lane, base, carry);
        }
        base = 0; carry = 0x80;
        const uint8_t *top = D + Lcup + Lref - 1;
        for (uint32_t k0 = 0; __ballot(k0 < n) != 0; k0 += 8 * LPB) {
            const uint32_t k = k0 + 8 * gl;
            uint2 dq = make_uint2(0u, 0u);
            if (k < n) __builtin_memcpy(&dq, top - k - 7, 8);
            ht_unstuff_backward_step8_g<LPB>(dq, k0, n, 0, sr, lane, base, carry);
        }
    }
    __syncthreads();
    if (ref) for (uint32_t i = gl; i < nsp; i += LPB) {
        vlO[nsw + i] = ss[i];
        meO[nsw + i] = sr[i];
    }
}

__global__ void __launch_bounds__(64)
k_ht_unstuff(const J2kBlock *__restrict__ blocks, int nblocks, const uint8_t *__restrict__ bytes,
             uint32_t *__restrict__ vlc_u, uint32_t *__restrict__ mel_u, uint32_t lds_words)
{
    extern __shared__ __align__(16) uint32_t sw[];
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= nblocks) return;
    const J2kBlock b = blocks[blockIdx.x];
    if (b.npasses == 0 || b.lcup < 2) return;
    const uint8_t *D = bytes + b.data_off;
    const uint32_t Lcup = b.lcup;
    /* A wave lives as long as its chain of dependent global loads.  The first 1024 bytes of the VLC stream (read
     * backward from Dcup[Lcup-2]) are fetched before Scup -- which says how many of them count -- is known: the
     * addresses depend on Lcup only and stay inside the byte pool (blocks in front, 16 bytes of pad at its start). */
    uint2 pv[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const long long o = (long long)b.data_off + Lcup - 9 - (512 * j + 8 * lane);
        pv[j] = make_uint2(0u, 0u);
        if (o >= 0) __builtin_memcpy(&pv[j], bytes + o, 8);
    }
    const uint32_t Scup = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((uint32_t)D[Lcup - 1] << 4) + (D[Lcup - 2] & 0x0F)));
    if (Scup < 2 || Scup > Lcup || Scup > 4079) return;
    const uint32_t nsw = ht_nsw(Scup);
    if (2 * nsw > lds_words) return;                     /* host sized the LDS from the same fields */
    uint32_t *vlO = vlc_u + (b.data_off >> 2), *meO = mel_u + (b.data_off >> 2);
    uint32_t *sv = sw;

    for (uint32_t i = lane; i < nsw; i += 64) sw[i] = 0;
    __syncthreads();

    /* ---- VLC: backward from Dcup[Lcup-2] ---- */
    {
        const uint8_t *top = D + Lcup - 2;
        const uint32_t n = Scup - 1;
        uint32_t base = 0, carry = 0x80;
#pragma unroll
        for (int j = 0; j < 2; j++)
            if (512u * j < n) ht_unstuff_backward_step8(pv[j], 512 * j, n, 0x0F, sv, lane, base, carry);
        for (uint32_t k0 = 1024; k0 < n; k0 += 512) {
            const uint32_t k = k0 + 8 * lane;
            uint2 dq = make_uint2(0u, 0u);
            if (k < n) __builtin_memcpy(&dq, top - k - 7, 8);
            ht_unstuff_backward_step8(dq, k0, n, 0x0F, sv, lane, base, carry);
        }
    }

    /* (the MEL stream, forward from Dcup[Pcup], is read from the raw bytes by k_ht_vlc: it is a few dozen bytes
     * in most blocks, but where it ends is not signalled, and un-stuffing all Scup bytes for it was half of this
     * kernel's work) */
    __syncthreads();
    for (uint32_t i = lane; i < nsw; i += 64) vlO[i] = sv[i];

    /* ---- refinement segment (blocks with more than the cleanup pass): SigProp forward, MagRef backward
     * over Dref = Dcup + Lcup (:1016-1185, :1260); zero bits past either end ---- */
    const int rem = b.npasses % 3, plhd = rem ? b.npasses - rem : b.npasses - 3;
    if (b.npasses - plhd > 1 && b.lref > 0) {
        const uint32_t Lref = b.lref, nsp = ht_nsp(Lref);
        if (2 * nsp > lds_words) return;
        uint32_t *ss = sw, *sr = sw + nsp;
        __syncthreads();
        for (uint32_t i = lane; i < 2 * nsp; i += 64) sw[i] = 0;
        __syncthreads();
        ht_unstuff_forward7(D + Lcup, Lref, ss, lane);
        ht_unstuff_backward(D + Lcup + Lref - 1, Lref, 0, sr, lane);
        __syncthreads();
        for (uint32_t i = lane; i < nsp; i += 64) {
            vlO[nsw + i] = ss[i];
            meO[nsw + i] = sr[i];
        }
    }
}

/* ================================================================== k_ht_refine
 * SigProp and MagRef decisions (jpeg2000htdec.c:1016-1185) of the blocks that carry refinement
 * passes, one LANE per block.  Both passes are serial chains through the block -- which sample
 * takes the next stream bit depends on the significance the previous bits created -- so, as with
 * the VLC chain, the parallelism is across blocks: 64 chains per wavefront instead of one chain
 * on lane 0 of a wavefront that idles its other 63 lanes.
 * Significance is kept as one mask per sample row (MT: 32 bits when no block of the launch is wider than 32 columns,
 * else 64; wider blocks stay with k_ht_decode's own serial pass): the 3x3 neighbourhood test is three shifts and an OR.
 * Input bits: the un-stuffed SigProp / MagRef arrays of k_ht_unstuff (zero bits past either end), read 16 bytes at a
 * time with the next 16 in flight (a lane reads its own block, so every load of the wave is 64 scattered requests: with
 * one word per load the kernel fetched 2.6 GB for 0.2 GB of stream, profiles/r02_configs_C3_C4_pmc_hbm.csv).
 * Output: three masks per row -- newly significant, its sign, MagRef bit -- which the MagSgn kernels apply while they
 * dequantise.  They are stored LANE-INTERLEAVED per wavefront of this kernel: mask k of row y of the block on lane l
 * sits at refbits[roff[block] + (3 y + k) * 64], roff[block] = base of the wave + l (htj2k_device.hip: ref_layout), so
 * that a store of the wave is 512 contiguous bytes.  ref_list[i] = block index. */
/* MAGREF = false: the MagSgn kernel deals out the MagRef bits itself (k_ht_decode_multi: the pass is a prefix sum there) and
 * the third mask of a row stays zero */
/* (six waves per SIMD where the state fits 80 registers without spilling -- 32-bit masks, SigProp only: the C3 case, 0.72 ->
 * 0.50 ms with both -- four otherwise; the kernel waits on scattered loads, so waves in flight are what it lives on) */
template <typename MT, bool MAGREF>
__global__ void __launch_bounds__(64, (sizeof(MT) == 4 && !MAGREF) ? 6 : 4)
k_ht_refine(const J2kBlock *__restrict__ blocks, const uint32_t *__restrict__ ref_list, int nref,
            const uint8_t *__restrict__ bytes, const ht_sym_t *__restrict__ qsym, const uint32_t *__restrict__ qoff,
            const uint32_t *__restrict__ vlc_u, const uint32_t *__restrict__ mel_u,
            uint64_t *__restrict__ refbits, const uint32_t *__restrict__ roff)
{
    constexpr int MB = (int)sizeof(MT) * 8;
    const int li = blockIdx.x * 64 + threadIdx.x;
    if (li >= nref) return;
    const uint32_t bi = ref_list[li];
    const J2kBlock b = blocks[bi];
    const int w = b.w, h = b.h, qw = (w + 1) >> 1, qh = (h + 1) >> 1;
    uint64_t *out = refbits + roff[bi];
    for (int y = 0; y < 3 * h; y++) out[(size_t)y * HT_REF_STRIDE] = 0;   /* also what an invalid block leaves behind */
    if (b.npasses == 0 || b.lcup < 2 || w > MB) return;
    const uint8_t *D = bytes + b.data_off;
    const uint32_t Scup = ((uint32_t)D[b.lcup - 1] << 4) + (D[b.lcup - 2] & 0x0F);
    if (Scup < 2 || Scup > b.lcup || Scup > 4079) return;
    const int rem = b.npasses % 3, plhd = rem ? b.npasses - rem : b.npasses - 3;
    const int z_blk = b.npasses - plhd;
    if (z_blk <= 1) return;
    const bool causal = (b.flags & J2K_CBLK_VSC) != 0;
    const ht_sym_t *qs = qsym + qoff[bi];
    const uint32_t nsw = ht_nsw(Scup), nsp = b.lref ? ht_nsp(b.lref) : 0;
    const uint32_t *spw = vlc_u + (b.data_off >> 2) + nsw, *mrw = mel_u + (b.data_off >> 2) + nsw;
    const MT wmask = w >= MB ? (MT)~(MT)0 : (MT)(((MT)1 << w) - 1);

    /* LSB-first readers over the un-stuffed arrays: a 64-bit buffer topped up a word at a time out of four words in
     * registers, the next four already requested.  Past the array: zero bits (a 16-byte piece may end up to three words
     * behind the array: in the next block's region or the buffer's slack, masked off here). */
    struct Bits {
        const uint32_t *p; uint32_t n, idx; uint64_t buf; int cnt; uint32_t q0, q1, q2, q3, left; uint4 nx;
        __device__ __forceinline__ uint4 fetch(uint32_t i) const
        {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (i < n) {
                __builtin_memcpy(&v, p + i, 16);
                if (i + 1 >= n) v.y = 0;
                if (i + 2 >= n) v.z = 0;
                if (i + 3 >= n) v.w = 0;
            }
            return v;
        }
        __device__ __forceinline__ void init(const uint32_t *q, uint32_t nw)
        {
            p = q; n = nw; buf = 0; cnt = 0;
            const uint4 v = fetch(0);
            q0 = v.x; q1 = v.y; q2 = v.z; q3 = v.w; left = 4;
            nx = fetch(4); idx = 8;
        }
        __device__ __forceinline__ void top_up()                     /* call with cnt <= 32 */
        {
            buf |= (uint64_t)q0 << cnt;
            cnt += 32;
            q0 = q1; q1 = q2; q2 = q3;
            if (--left == 0) {
                q0 = nx.x; q1 = nx.y; q2 = nx.z; q3 = nx.w; left = 4;
                nx = fetch(idx); idx += 4;
            }
        }
        __device__ __forceinline__ uint32_t get()
        {
            const uint32_t bit = (uint32_t)buf & 1u;
            buf >>= 1; cnt--;
            return bit;
        }
    } sp, mr;
    sp.init(spw, nsp);
    if (MAGREF) mr.init(mrw, nsp);

    /* significance rows of quad row qy: the symbols of a row are read as dwords (two quads; rows are padded to an even
     * number of quads and start on a dword), four at a time where the row pitch allows 16-byte loads */
    const uint32_t qpitch = ht_qsym_pitch((uint32_t)w);
    auto two_quads = [](uint32_t s2, int q, MT &top, MT &bot) {
        const uint32_t r2 = (s2 | (s2 >> 1)) & 0x00550055u;           /* significance of sample n of quad q / q + 1 at bit 2 n / 16 + 2 n */
        top |= (MT)((r2 & 1u) | ((r2 >> 3) & 2u) | ((r2 >> 14) & 4u) | ((r2 >> 17) & 8u)) << (2 * q);            /* samples 0, 2 */
        bot |= (MT)(((r2 >> 2) & 1u) | ((r2 >> 5) & 2u) | ((r2 >> 16) & 4u) | ((r2 >> 19) & 8u)) << (2 * q);     /* samples 1, 3 */
    };
    auto quad_rows = [&](int qy, MT &top, MT &bot) {
        top = 0; bot = 0;
        if (qy >= qh) return;
        const uint32_t *row = (const uint32_t *)(qs + (size_t)qy * qpitch);
        if ((qpitch & 7) == 0) {
            for (int q = 0; q < qw; q += 8) {
                const uint4 s8 = *(const uint4 *)(row + (q >> 1));
                two_quads(s8.x, q, top, bot);
                two_quads(s8.y, q + 2, top, bot);
                two_quads(s8.z, q + 4, top, bot);
                two_quads(s8.w, q + 6, top, bot);
            }
        } else {
            for (int q = 0; q < qw; q += 2) two_quads(row[q >> 1], q, top, bot);
        }
        top &= wmask; bot &= wmask;
        if (2 * qy + 1 >= h) bot = 0;
    };

    MT a_top, a_bot;
    quad_rows(0, a_top, a_bot);
    MT Mup = 0;                                                        /* significance of the row above the stripe, after its SigProp */
    for (int i0 = 0; i0 < h; i0 += 4) {
        MT b_top, b_bot, c_top, c_bot;
        quad_rows(i0 / 2 + 1, b_top, b_bot);
        quad_rows(i0 / 2 + 2, c_top, c_bot);
        const MT S0 = a_top, S1 = a_bot, S2 = b_top, S3 = b_bot, S4 = c_top;
        MT R0 = 0, R1 = 0, R2 = 0, R3 = 0, G0 = 0, G1 = 0, G2 = 0, G3 = 0, Q0 = 0, Q1 = 0, Q2 = 0, Q3 = 0;
        const int gh = min(4, h - i0);
        /* SigProp (:1016-1100).  A sample is a candidate when the cleanup pass left it insignificant and one of its eight
         * neighbours is significant.  The neighbours split into what the cleanup pass decided -- per row one mask
         * T_i, made once per stripe: the rows above and below smeared by a column either way, the row itself shifted --
         * and what this pass has decided so far, which at column j is only column j - 1 (all rows: `rprev`) and the rows
         * above in column j itself (`rcur`): two 4-bit values.  The masks are shifted along with the column, so that a
         * sample costs a handful of 32-bit operations instead of three variable 64-bit shifts. */
        {
            auto smear = [](MT x) -> MT { return x | (MT)(x << 1) | (x >> 1); };
            const bool dn0 = !causal || gh != 1, dn1 = !causal || gh != 2, dn2 = !causal || gh != 3, dn3 = !causal || gh != 4;
            MT s0 = S0, s1 = S1, s2 = S2, s3 = S3;
            MT T0 = smear(Mup) | (MT)(S0 << 1) | (S0 >> 1) | (dn0 ? smear(S1) : (MT)0);
            MT T1 = smear(S0) | (MT)(S1 << 1) | (S1 >> 1) | (dn1 ? smear(S2) : (MT)0);
            MT T2 = smear(S1) | (MT)(S2 << 1) | (S2 >> 1) | (dn2 ? smear(S3) : (MT)0);
            MT T3 = smear(S2) | (MT)(S3 << 1) | (S3 >> 1) | (dn3 ? smear(S4) : (MT)0);
            const uint32_t m0 = dn0 ? 7u : 3u, m1 = dn1 ? 7u : 3u, m2 = dn2 ? 7u : 3u, m3 = dn3 ? 7u : 3u;
            uint32_t rprev = 0;
            for (int j0 = 0; j0 < w; j0 += 4) {
                uint32_t rc4[4] = { 0, 0, 0, 0 };
                if (sp.cnt <= 32) sp.top_up();                         /* a 4x4 group takes at most 32 bits */
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    const int j = j0 + jj;
                    if (j < w) {
                        const uint32_t RP = rprev << 1;                 /* bit k + 1: row k of column j - 1 */
                        uint32_t rcur = 0;
                        if (gh > 0 && !((uint32_t)s0 & 1u) && ((((uint32_t)T0 & 1u) | (RP & m0)) != 0) && sp.get()) rcur |= 1u;
                        if (gh > 1 && !((uint32_t)s1 & 1u) && ((((uint32_t)T1 & 1u) | ((RP >> 1) & m1) | (rcur & 1u)) != 0) && sp.get()) rcur |= 2u;
                        if (gh > 2 && !((uint32_t)s2 & 1u) && ((((uint32_t)T2 & 1u) | ((RP >> 2) & m2) | (rcur & 2u)) != 0) && sp.get()) rcur |= 4u;
                        if (gh > 3 && !((uint32_t)s3 & 1u) && ((((uint32_t)T3 & 1u) | ((RP >> 3) & m3) | (rcur & 4u)) != 0) && sp.get()) rcur |= 8u;
                        R0 |= (MT)(rcur & 1u) << j; R1 |= (MT)((rcur >> 1) & 1u) << j;
                        R2 |= (MT)((rcur >> 2) & 1u) << j; R3 |= (MT)(rcur >> 3) << j;
                        rc4[jj] = rcur;
                        rprev = rcur;
                        s0 >>= 1; s1 >>= 1; s2 >>= 1; s3 >>= 1;
                        T0 >>= 1; T1 >>= 1; T2 >>= 1; T3 >>= 1;
                    }
                }
                /* signs of the samples this group made significant, same order (:1085-1100) */
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    const int j = j0 + jj;
                    const uint32_t r = rc4[jj];
                    if (j < w && r) {
                        if (r & 1u) G0 |= (MT)sp.get() << j;
                        if (r & 2u) G1 |= (MT)sp.get() << j;
                        if (r & 4u) G2 |= (MT)sp.get() << j;
                        if (r & 8u) G3 |= (MT)sp.get() << j;
                    }
                }
            }
        }
        if (MAGREF && z_blk > 2) {                                     /* MagRef: one bit per sample the cleanup pass made significant */
            for (int j = 0; j < w; j++) {
                if (mr.cnt <= 32) mr.top_up();
                if ((S0 >> j) & 1) Q0 |= (MT)mr.get() << j;
                if ((S1 >> j) & 1) Q1 |= (MT)mr.get() << j;
                if ((S2 >> j) & 1) Q2 |= (MT)mr.get() << j;
                if ((S3 >> j) & 1) Q3 |= (MT)mr.get() << j;
            }
        }
        uint64_t *o = out + (size_t)3 * i0 * HT_REF_STRIDE;            /* a store of the wave: the 64 lanes' masks, 512 contiguous bytes */
        o[0] = R0; o[HT_REF_STRIDE] = G0; o[2 * HT_REF_STRIDE] = Q0;
        if (gh > 1) { o[3 * HT_REF_STRIDE] = R1; o[4 * HT_REF_STRIDE] = G1; o[5 * HT_REF_STRIDE] = Q1; }
        if (gh > 2) { o[6 * HT_REF_STRIDE] = R2; o[7 * HT_REF_STRIDE] = G2; o[8 * HT_REF_STRIDE] = Q2; }
        if (gh > 3) { o[9 * HT_REF_STRIDE] = R3; o[10 * HT_REF_STRIDE] = G3; o[11 * HT_REF_STRIDE] = Q3; }
        Mup = S3 | R3;
        a_top = c_top; a_bot = c_bot;
    }
}

/* U-VLC prefix table (T.814 7.3.6 / jpeg2000htdec.c:338-352, 666-712): index = mode * 64 + 6 stream
 * bits; mode 0 no offset, 1 quad 0 only, 2 quad 1 only, 3 both, 4 both in the first quad row
 * after a MEL 0 (if prefix 1 > 2 the second value is a single bit + 1).
 * entry: p1 | p2 << 3 | prefix bits << 6 | suffix-1 bits << 9 | suffix-2 bits << 12 */
__host__ __device__ inline uint16_t ht_uvlc_entry(int mode, uint32_t v)
{
    const int pv[8] = { 5, 1, 2, 1, 3, 1, 2, 1 }, pl[8] = { 3, 1, 2, 1, 3, 1, 2, 1 };
    int p1 = 0, p2 = 0, lp = 0;
    if (mode == 1 || mode == 3 || mode == 4) { p1 = pv[v & 7]; lp = pl[v & 7]; }
    if (mode == 4 && p1 > 2) { p2 = (int)((v >> lp) & 1) + 1; lp += 1; }
    else if (mode >= 2) { const uint32_t w = v >> lp; p2 = pv[w & 7]; lp += pl[w & 7]; }
    const int s1 = p1 < 3 ? 0 : (p1 == 3 ? 1 : 5);
    const int s2 = (mode == 4 && p1 > 2) ? 0 : (p2 < 3 ? 0 : (p2 == 3 ? 1 : 5));
    return (uint16_t)(p1 | (p2 << 3) | (lp << 6) | (s1 << 9) | (s2 << 12));
}

/* LDS of k_ht_vlc beyond the two decode tables: staged VLC words, the output stage (16 quad
 * symbols per lane, pitch 17) with the three words of flush bookkeeping per lane, and the
 * significance bytes of the row above */
#define HT_VLC_OUT_PITCH 9
#define HT_VLC_OUT_BYTES (64 * (HT_VLC_OUT_PITCH + 3) * 4)
__host__ __device__ inline size_t ht_vlc_lds_bytes(uint32_t max_qw)
{
    return 4096 + 1024 + HT_VSTAGE_BYTES + HT_VLC_OUT_BYTES + (size_t)((((max_qw + 3) >> 2) | 1) << 2) * 64;
}
/* k_ht_vlc: launches with a block wider than 64 columns (legal up to 1024 x 4); one wave per workgroup, the significance
 * patterns of the row above in LDS byte rows.  Everything else runs k_ht_vlc2 below, which shares four waves' tables. */
#define HT_VLC_NARROW_WAVES 4                  /* waves per workgroup of k_ht_vlc2 */
__global__ void __launch_bounds__(64)
k_ht_vlc(const J2kBlock *__restrict__ blocks, int nblocks, const uint8_t *__restrict__ bytes,
         const uint16_t *__restrict__ g_tables, ht_sym_t *__restrict__ qsym,
         const uint32_t *__restrict__ qoff, uint32_t max_qw,
         const uint32_t *__restrict__ vlc_u, const uint32_t *__restrict__ mel_u, uint32_t *__restrict__ sink)
{
    extern __shared__ __align__(16) uint8_t smem[];
    uint16_t *tbl = (uint16_t *)smem;
    uint16_t *utbl = (uint16_t *)(smem + 4096);          /* HT_UVLC_ENTRIES entries */
    constexpr int CAD = 8;                               /* passes per flush of the output stage */
    constexpr int VPITCH = HT_VSTAGE_PITCH, OPITCH = HT_VLC_OUT_PITCH;
    const int lane = threadIdx.x & 63;
    uint32_t *vstage = (uint32_t *)(smem + 4096 + 1024);
    /* output stage: the two quad symbols of a pass go to LDS; every 8 passes the wave writes the 64
     * lanes' 64-byte chunks out together, 4 lanes per chunk with 16-byte stores.  A lane storing
     * its own two dwords per pass made 128 separate line requests per pass and wave (every lane
     * writes into a different block's symbol array): that address traffic, not arithmetic, was
     * 40 % of this kernel's time. */
    uint32_t *ostage = (uint32_t *)(smem + 4096 + 1024 + HT_VSTAGE_BYTES);
    uint32_t *obase_lo = ostage + 64 * HT_VLC_OUT_PITCH, *obase_hi = obase_lo + 64, *onit = obase_hi + 64;
    /* significance patterns of the row above, one byte per quad, [lane][quad]; the pitch in
     * dwords is odd so the 64 lanes hit distinct banks */
    uint8_t *rho_rows = smem + 4096 + 1024 + HT_VSTAGE_BYTES + HT_VLC_OUT_BYTES;
    const int pitch = (int)((((max_qw + 3) >> 2) | 1) << 2);
    const int bi = blockIdx.x * blockDim.x + threadIdx.x;
    /* the decode tables with the symbol's four fields where e_k and e_1 were: u_off | len << 1 | rho << 4 | fields << 8 */
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) {
        const uint32_t e = g_tables[i];
        tbl[i] = (uint16_t)((e & 0xFF) | (ht_sym_pack_fields((e >> 4) & 0xF, (e >> 8) & 0xF, (e >> 12) & 0xF) << 8));
    }
    for (int i = threadIdx.x; i < HT_UVLC_ENTRIES; i += blockDim.x)
        utbl[i] = ht_uvlc_entry(i >> 6, (uint32_t)(i & 63));

    int qw = 0, qh = 0;
    uint32_t doff = 0;
    ht_sym_t *qout = qsym;
    const uint8_t *mraw = bytes + 16;                    /* Dcup + Pcup: the MEL bytes; lanes without a block read the pool's front pad */
    int mlim = 0;                                        /* Scup */
    if (bi < nblocks) {
        const J2kBlock b = blocks[bi];
        bool ok = b.npasses != 0 && b.lcup >= 2;
        uint32_t Scup = 0;
        if (ok) {
            const uint8_t *D = bytes + b.data_off;
            Scup = ((uint32_t)D[b.lcup - 1] << 4) + (D[b.lcup - 2] & 0x0F);
            ok = !(Scup < 2 || Scup > b.lcup || Scup > 4079);
        }
        if (ok && ((b.w + 1u) >> 1) <= max_qw) {
            qw = (b.w + 1) >> 1;
            qh = (b.h + 1) >> 1;
            qout = qsym + qoff[bi];
            doff = b.data_off >> 2;
            mraw = bytes + b.data_off + (b.lcup - Scup);
            mlim = (int)Scup;
        }
    }
    /* one flat loop, the same number of passes for every lane of the wave: lane-local (row, qx)
     * walk the block's quad pairs; n_it passes produce output */
    const int ppr = (qw + 1) >> 1;                        /* passes per quad row */
    const int n_it = qh * ppr;
    int max_it = n_it;
#pragma unroll
    for (int o = 32; o; o >>= 1) max_it = max(max_it, __shfl_xor(max_it, o));
    const uint32_t my_lo = (uint32_t)(uintptr_t)qout, my_hi = (uint32_t)((uintptr_t)qout >> 32);
    obase_lo[lane] = my_lo;
    obase_hi[lane] = my_hi;
    onit[lane] = (uint32_t)n_it;
    __syncthreads();

    const uint32_t *vsrc = vlc_u + doff;
    uint8_t *myrho = rho_rows + lane * pitch;
    uint32_t vpos = 4;                                   /* the first 4 VLC bits are the Scup nibble (:283-295) */
    uint32_t *vst = vstage + lane * VPITCH;
    uint32_t *ost = ostage + lane * OPITCH;
    /* VLC words: a ring of 16 per lane in LDS (vst[w & 15] = stream word w).  Every second pass a lane whose ring holds
     * fewer than 8 words beyond its position asks for the next 8 (32 bytes: a quarter of a line -- with 16 the line had
     * left the L2 again before the lane came back for its next piece, and FETCH_SIZE was six times the stream); they are
     * written to the ring two passes later.  Two passes take at most 76 bits, so a lane that did not ask still has 5
     * words at the next check (a pass reads 3 from a position at most 2 words further); the 8 slots written are at
     * least one word behind the position (it asked with fewer than 8 ahead), and at most 15 words are ever ahead.
     * Every stream byte is requested once: 24-dword windows re-requested every 8 passes asked for each byte ~7 times,
     * and L2 requests, not arithmetic, bounded this kernel. */
    uint4 nxv = make_uint4(0u, 0u, 0u, 0u), nxw = make_uint4(0u, 0u, 0u, 0u);
    uint32_t hw = 16, pw = 0;                            /* words requested so far / word index of nxv (nxw: the four after it) */
    bool pend = false;
    /* MEL (jpeg2000htdec.c:462-495): decoded symbols are buffered, LSB = next symbol; the
     * adaptive run-length state machine only runs in a rarely taken refill path that decodes
     * up to six codewords (>= 6, typically >= 32 symbols).  The stream is read from the raw bytes Dcup[Pcup ...]
     * (:429-440: MSB first, a byte behind 0xFF has 7 bits, Dcup[Lcup-1] counts as 0xFF and Dcup[Lcup-2] with its low
     * nibble set, 0xFF for ever behind the segment).  `mrb` is the position in raw bits: bit 0 of a byte behind 0xFF
     * is never pointed at. */
    uint64_t msyms = 0; int mcnt = 0; uint32_t mrb = 0; int mel_k = 0;
    uint32_t pfA, pfB, pfC;                              /* raw bytes -1 .. 10 around the next refill's position */
    __builtin_memcpy(&pfA, mraw - 1, 4);
    __builtin_memcpy(&pfB, mraw + 3, 4);
    __builtin_memcpy(&pfC, mraw + 7, 4);
    if (qh > 0) {
#pragma unroll
        for (int jx = 0; jx < 4; jx++) {
            uint4 q;
            __builtin_memcpy(&q, vsrc + 4 * jx, 16);
            vst[4 * jx] = q.x; vst[4 * jx + 1] = q.y; vst[4 * jx + 2] = q.z; vst[4 * jx + 3] = q.w;
        }
    }
    /* the wave writes window `win` (passes CAD win .. CAD win + CAD - 1: 2 CAD symbols = 32 bytes) of all lanes: 2 lanes
     * per chunk */
    auto flush = [&](int win) {
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int c = 32 * p + (lane >> 1), part = lane & 1;
            const uint32_t *src = ostage + c * OPITCH + 4 * part;
            const uint4 v = make_uint4(src[0], src[1], src[2], src[3]);
            /* the symbol array and pass count of the lane that owns chunk c */
            const uint32_t c_lo = obase_lo[c], c_hi = obase_hi[c], c_nit = onit[c];
            uint32_t *dst = (uint32_t *)(((uintptr_t)c_hi << 32) | c_lo) + (size_t)win * CAD + 4 * part;
            /* chunks of lanes that are done (or never had a block) go to a scratch line: always two
             * stores, so that the wait for the staged VLC words can be counted (vmcnt) */
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(1))) u32x4 g_u32x4;    /* a global, not a FLAT, store */
            u32x4 vv; vv.x = v.x; vv.y = v.y; vv.z = v.z; vv.w = v.w;
            *(g_u32x4 *)(uintptr_t)((uint32_t)(win * CAD) < c_nit ? dst : sink + 4 * part) = vv;
        }
    };

    int ctx_run = 0, row = 0, qx = 0;
    int rho_left = 0, ral = 0, ra_next = 0;
    for (int t = 0; t < max_it; t++) {
        const bool active = t < n_it;
        const uint16_t *table = tbl + (row ? 1024 : 0);
        const bool row0 = row == 0;
        const bool pair = qx + 1 < qw;
        if ((t & 1) == 0) {
            if (pend) {
                uint32_t *r = vst + (pw & 15);
                r[0] = nxv.x; r[1] = nxv.y; r[2] = nxv.z; r[3] = nxv.w;
                r = vst + ((pw + 4) & 15);
                r[0] = nxw.x; r[1] = nxw.y; r[2] = nxw.z; r[3] = nxw.w;
            }
            pend = (int)(hw - (vpos >> 5)) < 8;
            if (pend) {
                __builtin_memcpy(&nxv, vsrc + hw, 16);
                __builtin_memcpy(&nxw, vsrc + hw + 4, 16);
                pw = hw;
                hw += 8;
            }
            if ((t & (CAD - 1)) == 0 && t) flush(t / CAD - 1);   /* behind the load: nothing waits for these stores */
        }
        /* A pair uses at most 3 MEL symbols.  When one lane is short, every lane with room for more takes the refill
         * with it: the wave executes the refill block for all of them anyway, and a lane that has just been topped up
         * to 32 or more symbols will not be the one that asks for the next ten passes.  (Content with many all-zero
         * neighbourhoods -- smooth pictures -- uses MEL symbols on most quads: with every lane refilling on its own
         * schedule some lane was short on nearly every pass, and the 390 instructions of this block ran every pass:
         * k_ht_vlc 1.61 ms against 0.81 on noisy frames.) */
        if (__ballot(mcnt < 3) != 0 && mcnt <= 32) {
            /* >= 42 MEL bits from mrb, first bit in the MSB; six codewords need <= 36.  The twelve bytes were
             * requested at the end of the previous refill (mrb only moves here): with 64 lanes nearly every pass
             * has some lane refilling, and a fresh load would cost the whole wave a memory round trip each time */
            const int mb = (int)(mrb >> 3);
            const uint32_t o = mrb & 7;
            const uint32_t a0 = mb ? pfA : (pfA & ~0xFFu);              /* the first byte of the stream has 8 bits */
            const uint32_t anyff = ((~a0 - 0x01010101u) & a0 & 0x80808080u) | ((~pfB - 0x01010101u) & pfB & 0x80808080u);
            const bool slow = anyff != 0 || mb + 10 > mlim;             /* a 0xFF in bytes -1 .. 6, or the segment's end near */
            uint64_t mw;
            uint32_t n7 = 0;                                            /* bit j: byte j of the window has 7 bits */
            if (!slow) {
                const uint32_t hi = __builtin_amdgcn_perm(pfB, pfA, 0x01020304u), lo = __builtin_amdgcn_perm(pfC, pfB, 0x01020304u);
                mw = (((uint64_t)hi << 32) | lo) << o;
            } else {
                uint32_t prev = 0, total = 0;
                mw = 0;
#pragma unroll
                for (int j = -1; j < 7; j++) {
                    const int p = mb + j;
                    const uint32_t raw = ((j < 3 ? pfA >> (8 * (j + 1)) : pfB >> (8 * (j - 3)))) & 0xFF;
                    const uint32_t v = p >= mlim - 1 ? 0xFFu : (p == mlim - 2 ? raw | 0x0Fu : raw);
                    if (j >= 0) {
                        const uint32_t n = prev == 0xFF ? 7u : 8u;
                        mw = (mw << n) | (v & ((1u << n) - 1));
                        total += n;
                        n7 |= (8u - n) << j;
                    }
                    prev = p < 0 ? 0u : v;
                }
                mw <<= (64 - total) + (o - (n7 & 1));
            }
            uint32_t used_all = 0;
#pragma unroll
            for (int cw = 0; cw < 6; cw++) {
                const int eval = (int)((0x5433222111000ull >> (4 * mel_k)) & 0xF);
                const int b = (int)(mw >> 63);
                const int run = b ? (1 << eval) : (eval ? (int)((mw << 1) >> (64 - eval)) : 0);
                const int nsy = run + (b ? 0 : 1);
                if (mcnt + nsy <= 64) {                  /* otherwise leave the codeword for the next refill */
                    if (!b) msyms |= 1ull << ((mcnt + run) & 63);
                    mcnt += nsy;
                    const int used = b ? 1 : 1 + eval;
                    mw <<= used; used_all += used;
                    mel_k = b ? (mel_k < 12 ? mel_k + 1 : 12) : (mel_k > 0 ? mel_k - 1 : 0);
                }
            }
            if (!slow) {
                mrb += used_all;                                        /* lands in bytes 0 .. 5: all of 8 bits */
            } else {
                /* byte L of the window is the first one with bits left: cum = bits up to the end of byte j */
                uint32_t L = 0, before = 0, cum = 8 - o;
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    const bool ge = used_all >= cum;
                    L += ge;
                    before = ge ? cum : before;
                    cum += 8 - ((n7 >> (j + 1)) & 1);
                }
                mrb = L ? (uint32_t)(mb + (int)L) * 8 + ((n7 >> L) & 1) + (used_all - before) : mrb + used_all;
            }
            const uint8_t *pm = mraw + min((int)(mrb >> 3), mlim) - 1;
            __builtin_memcpy(&pfA, pm, 4);
            __builtin_memcpy(&pfB, pm + 4, 4);
            __builtin_memcpy(&pfC, pm + 8, 4);
        }
        uint64_t vwin;
        {
            const uint32_t kw = (vpos >> 5) & 15, sh = vpos & 31;
            const uint32_t a0 = vst[kw], a1 = vst[(kw + 1) & 15], a2 = vst[(kw + 2) & 15];
            vwin = ((uint64_t)__builtin_amdgcn_alignbit(a2, a1, sh) << 32) | __builtin_amdgcn_alignbit(a1, a0, sh);
        }
        uint32_t m = (uint32_t)msyms, mused = 0;         /* next MEL symbols, LSB first */
        uint32_t a = (uint32_t)vwin, aused = 0;         /* the two codewords need <= 14 bits */
        int rho[2], uoff[2];
        uint32_t pk[2];                                  /* the four 2-bit fields of the symbol */
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const bool en = active && (k == 0 || pair);
            const int q = qx + k;
            const int ra = ra_next;
            const int rar = (row && en && q + 1 < qw) ? (int)myrho[q + 1] : 0;
            const int ctx = row0 ? ctx_run
                                 : ((((ra >> 1) | (ral >> 3)) & 1) | ((((rho_left >> 2) | (rho_left >> 3)) & 1) << 1) |
                                    ((((ra >> 3) | (rar >> 1)) & 1) << 2));
            const bool mq = en && ctx == 0;
            const int msym = (int)(m & 1);
            m >>= mq ? 1 : 0; mused += mq ? 1 : 0;
            const bool dec = en && (ctx != 0 || msym != 0);
            const uint32_t e = dec ? table[(ctx << 7) | (a & 0x7F)] : 0u;
            const uint32_t len = (e >> 1) & 7;
            a >>= len; aused += len;
            uoff[k] = e & 1; rho[k] = (e >> 4) & 0xF; pk[k] = e >> 8;
            if (en) {
                rho_left = rho[k];
                ral = ra;
                ra_next = rar;
                ctx_run = ((rho[k] | (rho[k] >> 1)) & 1) | (((rho[k] >> 2) & 1) << 1) | (((rho[k] >> 3) & 1) << 2);   /* first row only */
            }
        }
        /* U-VLC (jpeg2000htdec.c:338-388, 666-712, 828-854) for both quads, branch-free, on a
         * fresh 32-bit window (<= 24 bits): decode order pfx1 pfx2 sfx1 sfx2 ext1 ext2; first
         * row with both offsets set: one MEL symbol, 1 => both u get +2, 0 and pfx1 > 2 => u2
         * is a single bit + 1 */
        uint32_t u32w = (uint32_t)(vwin >> aused), uused;
        const bool both = uoff[0] && uoff[1];
        const bool mq2 = row0 && both;
        const int mel2 = (int)(m & 1);
        mused += mq2 ? 1 : 0;
        const int mode = (mq2 && !mel2) ? 4 : (uoff[0] | (uoff[1] << 1));
        const uint32_t ue = utbl[(mode << 6) | (u32w & 63)];
        const int p1 = ue & 7, p2 = (ue >> 3) & 7;
        uint32_t d = (ue >> 6) & 7;
        u32w >>= d; uused = d;
        d = (ue >> 9) & 7;
        const int s1 = (int)(u32w & ((1u << d) - 1)); u32w >>= d; uused += d;
        d = (ue >> 12) & 7;
        const int s2 = (int)(u32w & ((1u << d) - 1)); u32w >>= d; uused += d;
        d = s1 >= 28 ? 4u : 0u;
        const int x1 = (int)(u32w & ((1u << d) - 1)); u32w >>= d; uused += d;
        d = s2 >= 28 ? 4u : 0u;
        const int x2 = (int)(u32w & ((1u << d) - 1)); uused += d;
        const int bias = (mq2 && mel2) ? 2 : 0;
        const int u1 = uoff[0] ? bias + p1 + s1 + 4 * x1 : 0;
        const int u2 = uoff[1] ? bias + p2 + s2 + 4 * x2 : 0;
        if (active) {
            vpos += aused + uused;
            msyms >>= mused; mcnt -= (int)mused;
            myrho[qx] = (uint8_t)rho[0];
            if (pair) myrho[qx + 1] = (uint8_t)rho[1];
        }
        ost[t & (CAD - 1)] = pk[0] | ((uint32_t)u1 << 8) | (pk[1] << 16) | ((uint32_t)u2 << 24);
        /* next quad pair of this lane's block */
        qx += 2;
        if (active && qx >= qw) {
            qx = 0; row++;
            rho_left = 0; ral = 0;
            ra_next = (int)myrho[0];                     /* above quad 0 of the new row */
        }
    }
    if (max_it > 0) flush((max_it - 1) / CAD);
}


/* ================================================================== k_ht_vlc2
 * k_ht_vlc for launches without a block wider than 64 columns (nearly every stream), with everything a pass computes
 * rearranged around what costs instructions -- the kernel is bound by VALU issue (SQ counters: profiles/r03_ht_sq.csv):
 *   contexts   A context of a quad below the first row looks at three things (jpeg2000htdec.c:725-760): c0 = the
 *              sample above-left or above, c2 = above-right, and the quad to the left.  c0 and c2 of ALL quads of a row
 *              are made at once when the row above ends: that row's (rho bit 1, rho bit 3) pairs were shifted into a
 *              64-bit register M as they were decoded, and with I = M in quad order
 *                  D = ((I | I << 1) & 0x55...) | ((I | I >> 1) & 0xAA...)
 *              holds (c0, c2) of quad q at bits 2 q, 2 q + 1.  A pass takes its four bits off the bottom of D.
 *   tables     The CxtVLC table of the other rows is stored under the index  codeword bits | c0 << 7 | c2 << 8 | left << 9,
 *              so the byte address is  bits << 1 | (D & 3) << 8 | left << 10  and "context 0" is "address < 0x100".  An
 *              entry is  u_off | len << 1 | rho bit 1 << 4 | rho bit 3 << 5 | (rho bit 2 or 3) << 6 | fields << 8:
 *              what the NEXT quads ask of this one is in the entry, rho itself is not (the first row, which needs all of
 *              it for its running context, gets it back from the fields).
 *   MEL        A quad in context 0 whose MEL symbol is 0 decodes nothing: the entry is multiplied by (symbol | not context 0).
 *   U-VLC      32-bit entries laid out so that they serve as operands as they are: prefix length in bits 0-4 (the offset of
 *              the first suffix), the offset of the second suffix in bits 11-15, both prefixes at the byte positions of
 *              the two u of the output word; suffix extensions (u > 32) in a branch that is taken when a lane has one.
 *   stream     a ring of 16 words per lane with its first two words mirrored behind it: the three words of a window are
 *              consecutive.
 * MAIN: every lane of the wave is inside its block and past the first quad row, and every block has an even number of
 * quads per row -- all but the first and the last few passes of nearly every wave (the block table is sorted by size).
 * Then nothing is predicated: both quads of the pair exist, the contexts come from the row above, the first-row U-VLC rule
 * (:666-712) cannot apply.  The general form of the same pass runs the ragged start and end.  LDS: 4 KB tables + 1 KB U-VLC + 4 waves x 64 x (18 + 9) words =
 * exactly a fifth of a CU's 160 KB. */
#define HT_VLC2_VPITCH 18
#define HT_VLC2_LDS (4096 + 1024 + HT_VLC_NARROW_WAVES * 64 * (HT_VLC2_VPITCH + HT_VLC_OUT_PITCH) * 4)

/* ht_uvlc_entry's fields as lp | d1 << 5 | p1 << 8 | (lp + d1) << 11 | d2 << 16 | (lp + d1 + d2) << 19 | p2 << 24 */
__host__ __device__ inline uint32_t ht_uvlc_entry2_from(uint32_t p1, uint32_t p2, uint32_t lp, uint32_t d1, uint32_t d2)
{
    return lp | (d1 << 5) | (p1 << 8) | ((lp + d1) << 11) | (d2 << 16) | ((lp + d1 + d2) << 19) | (p2 << 24);
}
__host__ __device__ inline uint32_t ht_uvlc_entry2(int mode, uint32_t v)
{
    const uint32_t e = ht_uvlc_entry(mode, v);
    return ht_uvlc_entry2_from(e & 7, (e >> 3) & 7, (e >> 6) & 7, (e >> 9) & 7, (e >> 12) & 7);
}
/* mode 4 (both quads of a first-row pair have an offset and the MEL symbol was 0, :666-712) without a table: the prefix
 * of quad 1 as ever, and if it is > 2 the second value is a single bit + 1 */
__device__ __forceinline__ uint32_t ht_uvlc_entry2_mode4(uint32_t v)
{
    const uint32_t b = v & 7;
    const uint32_t p1 = (0x12131215u >> (4 * b)) & 0xF;                 /* value {5,1,2,1,3,1,2,1}, length {3,1,2,1,3,1,2,1} */
    uint32_t lp = (0x12131213u >> (4 * b)) & 0xF, p2;
    if (p1 > 2) { p2 = ((v >> lp) & 1) + 1; lp += 1; }
    else { const uint32_t b2 = (v >> lp) & 7; p2 = (0x12131215u >> (4 * b2)) & 0xF; lp += (0x12131213u >> (4 * b2)) & 0xF; }
    const uint32_t d1 = p1 < 3 ? 0u : (p1 == 3 ? 1u : 5u);
    const uint32_t d2 = p1 > 2 ? 0u : (p2 < 3 ? 0u : (p2 == 3 ? 1u : 5u));
    return ht_uvlc_entry2_from(p1, p2, lp, d1, d2);
}

__global__ void __launch_bounds__(64 * HT_VLC_NARROW_WAVES)
k_ht_vlc2(const J2kBlock *__restrict__ blocks, int nblocks, const uint8_t *__restrict__ bytes,
          const uint16_t *__restrict__ g_tables, ht_sym_t *__restrict__ qsym,
          const uint32_t *__restrict__ qoff,
          const uint32_t *__restrict__ vlc_u, uint32_t *__restrict__ sink)
{
    extern __shared__ __align__(16) uint8_t smem[];
    uint16_t *tbl = (uint16_t *)smem;                    /* [0, 1024): first row, index ctx << 7 | bits; [1024, 2048): the other rows */
    uint32_t *utbl = (uint32_t *)(smem + 4096);          /* modes 0-3, 64 entries each */
    constexpr int CAD = 8;                               /* passes per flush of the output stage */
    constexpr int VPITCH = HT_VLC2_VPITCH, OPITCH = HT_VLC_OUT_PITCH;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t *vstage = (uint32_t *)(smem + 4096 + 1024) + wv * 64 * (VPITCH + OPITCH);
    uint32_t *ostage = vstage + 64 * VPITCH;
    const int bi = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) {
        const uint32_t e = g_tables[i], rho = (e >> 4) & 0xF;
        const uint32_t ne = (e & 0xF) | (((rho >> 1) & 1) << 4) | (((rho >> 3) & 1) << 5) | ((((rho >> 2) | (rho >> 3)) & 1) << 6) |
                            (ht_sym_pack_fields(rho, (e >> 8) & 0xF, (e >> 12) & 0xF) << 8);
        const int ctx = (i >> 7) & 7, bits = i & 127;
        const int dst = i < 1024 ? i : 1024 + (bits | ((ctx & 1) << 7) | (((ctx >> 2) & 1) << 8) | (((ctx >> 1) & 1) << 9));
        tbl[dst] = (uint16_t)ne;
    }
    for (int i = threadIdx.x; i < 4 * 64; i += blockDim.x)
        utbl[i] = ht_uvlc_entry2(i >> 6, (uint32_t)(i & 63));

    int qw = 0, qh = 0;
    uint32_t doff = 0;
    ht_sym_t *qout = qsym;
    const uint8_t *mraw = bytes + 16;                    /* Dcup + Pcup: the MEL bytes; lanes without a block read the pool's front pad */
    int mlim = 0;                                        /* Scup */
    if (bi < nblocks) {
        const J2kBlock b = blocks[bi];
        bool ok = b.npasses != 0 && b.lcup >= 2;
        uint32_t Scup = 0;
        if (ok) {
            const uint8_t *D = bytes + b.data_off;
            Scup = ((uint32_t)D[b.lcup - 1] << 4) + (D[b.lcup - 2] & 0x0F);
            ok = !(Scup < 2 || Scup > b.lcup || Scup > 4079);
        }
        if (ok && ((b.w + 1u) >> 1) <= 32u) {
            qw = (b.w + 1) >> 1;
            qh = (b.h + 1) >> 1;
            qout = qsym + qoff[bi];
            doff = b.data_off >> 2;
            mraw = bytes + b.data_off + (b.lcup - Scup);
            mlim = (int)Scup;
        }
    }
    const int ppr = (qw + 1) >> 1;                        /* passes per quad row */
    const int n_it = qh * ppr;
    int max_it = n_it;
#pragma unroll
    for (int o = 32; o; o >>= 1) max_it = max(max_it, __shfl_xor(max_it, o));
    const uint32_t my_lo = (uint32_t)(uintptr_t)qout, my_hi = (uint32_t)((uintptr_t)qout >> 32);
    __syncthreads();

    const uint32_t *vsrc = vlc_u + doff;
    uint32_t vpos = 4;                                   /* the first 4 VLC bits are the Scup nibble (:283-295) */
    uint32_t *vst = vstage + lane * VPITCH;
    uint32_t *ost = ostage + lane * OPITCH;
    /* VLC words: see k_ht_vlc.  vst[w & 15] = stream word w; vst[16], vst[17] repeat vst[0], vst[1] */
    uint4 nxv = make_uint4(0u, 0u, 0u, 0u), nxw = make_uint4(0u, 0u, 0u, 0u);
    uint32_t hw = 16, pw = 0;                            /* words requested so far / word index of nxv (nxw: the four after it) */
    bool pend = false;
    uint64_t msyms = 0; int mcnt = 0; uint32_t mrb = 0; int mel_k = 0;     /* MEL: see k_ht_vlc */
    uint32_t pfA, pfB, pfC;                              /* raw bytes -1 .. 10 around the next refill's position */
    __builtin_memcpy(&pfA, mraw - 1, 4);
    __builtin_memcpy(&pfB, mraw + 3, 4);
    __builtin_memcpy(&pfC, mraw + 7, 4);
    if (qh > 0) {
#pragma unroll
        for (int jx = 0; jx < 4; jx++) {
            uint4 q;
            __builtin_memcpy(&q, vsrc + 4 * jx, 16);
            vst[4 * jx] = q.x; vst[4 * jx + 1] = q.y; vst[4 * jx + 2] = q.z; vst[4 * jx + 3] = q.w;
            if (jx == 0) { vst[16] = q.x; vst[17] = q.y; }
        }
    }
    auto flush = [&](int win) {
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int c = 32 * p + (lane >> 1), part = lane & 1;
            const uint32_t *src = ostage + c * OPITCH + 4 * part;
            const uint4 v = make_uint4(src[0], src[1], src[2], src[3]);
            /* the symbol array and pass count of the lane that owns chunk c */
            const uint32_t c_lo = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * c, (int)my_lo);
            const uint32_t c_hi = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * c, (int)my_hi);
            const uint32_t c_nit = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * c, n_it);
            uint32_t *dst = (uint32_t *)(((uintptr_t)c_hi << 32) | c_lo) + (size_t)win * CAD + 4 * part;
            /* chunks of lanes that are done (or never had a block) go to a scratch line: always two stores */
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(1))) u32x4 g_u32x4;    /* a global, not a FLAT, store */
            u32x4 vv; vv.x = v.x; vv.y = v.y; vv.z = v.z; vv.w = v.w;
            *(g_u32x4 *)(uintptr_t)((uint32_t)(win * CAD) < c_nit ? dst : sink + 4 * part) = vv;
        }
    };

    int row = 0, qx = 0;
    uint32_t ctx_run = 0;                                /* first row: the context the quad just decoded leaves for the next */
    uint32_t lf10 = 0;                                   /* other rows: "the quad to the left has a significant sample on its right side", at bit 10 */
    uint32_t Dlo = 0, Dhi = 0, Mlo = 0, Mhi = 0;         /* D: (c0, c2) of the quads from qx on, lowest first; M: (rho1, rho3) of this row's quads so far, at the top */
    const uint8_t *tb1 = (const uint8_t *)(tbl + 1024);
    auto pass = [&](int t, auto main_tag) {
        constexpr bool MAIN = decltype(main_tag)::value;
        const bool active = MAIN || t < n_it;
        const bool row0 = !MAIN && row == 0;
        const bool pair = MAIN || qx + 1 < qw;
        if ((t & 1) == 0) {
            if (pend) {
                uint32_t *r = vst + (pw & 15);
                r[0] = nxv.x; r[1] = nxv.y; r[2] = nxv.z; r[3] = nxv.w;
                if (!(pw & 15)) { vst[16] = nxv.x; vst[17] = nxv.y; }
                r = vst + ((pw + 4) & 15);
                r[0] = nxw.x; r[1] = nxw.y; r[2] = nxw.z; r[3] = nxw.w;
            }
            pend = (int)(hw - (vpos >> 5)) < 8;
            if (pend) {
                __builtin_memcpy(&nxv, vsrc + hw, 16);
                __builtin_memcpy(&nxw, vsrc + hw + 4, 16);
                pw = hw;
                hw += 8;
            }
            if ((t & (CAD - 1)) == 0 && t) flush(t / CAD - 1);   /* behind the load: nothing waits for these stores */
        }
        if (__ballot(mcnt < 3) != 0 && mcnt <= 32) {
            /* >= 42 MEL bits from mrb, first bit in the MSB; six codewords need <= 36.  The twelve bytes were
             * requested at the end of the previous refill (mrb only moves here): with 64 lanes nearly every pass
             * has some lane refilling, and a fresh load would cost the whole wave a memory round trip each time */
            const int mb = (int)(mrb >> 3);
            const uint32_t o = mrb & 7;
            const uint32_t a0 = mb ? pfA : (pfA & ~0xFFu);              /* the first byte of the stream has 8 bits */
            const uint32_t anyff = ((~a0 - 0x01010101u) & a0 & 0x80808080u) | ((~pfB - 0x01010101u) & pfB & 0x80808080u);
            const bool slow = anyff != 0 || mb + 10 > mlim;             /* a 0xFF in bytes -1 .. 6, or the segment's end near */
            uint64_t mw;
            uint32_t n7 = 0;                                            /* bit j: byte j of the window has 7 bits */
            if (!slow) {
                const uint32_t hi = __builtin_amdgcn_perm(pfB, pfA, 0x01020304u), lo = __builtin_amdgcn_perm(pfC, pfB, 0x01020304u);
                mw = (((uint64_t)hi << 32) | lo) << o;
            } else {
                uint32_t prev = 0, total = 0;
                mw = 0;
#pragma unroll
                for (int j = -1; j < 7; j++) {
                    const int p = mb + j;
                    const uint32_t raw = ((j < 3 ? pfA >> (8 * (j + 1)) : pfB >> (8 * (j - 3)))) & 0xFF;
                    const uint32_t v = p >= mlim - 1 ? 0xFFu : (p == mlim - 2 ? raw | 0x0Fu : raw);
                    if (j >= 0) {
                        const uint32_t n = prev == 0xFF ? 7u : 8u;
                        mw = (mw << n) | (v & ((1u << n) - 1));
                        total += n;
                        n7 |= (8u - n) << j;
                    }
                    prev = p < 0 ? 0u : v;
                }
                mw <<= (64 - total) + (o - (n7 & 1));
            }
            uint32_t used_all = 0;
#pragma unroll
            for (int cw = 0; cw < 6; cw++) {
                const int eval = (int)((0x5433222111000ull >> (4 * mel_k)) & 0xF);
                const int b = (int)(mw >> 63);
                const int run = b ? (1 << eval) : (eval ? (int)((mw << 1) >> (64 - eval)) : 0);
                const int nsy = run + (b ? 0 : 1);
                if (mcnt + nsy <= 64) {                  /* otherwise leave the codeword for the next refill */
                    if (!b) msyms |= 1ull << ((mcnt + run) & 63);
                    mcnt += nsy;
                    const int used = b ? 1 : 1 + eval;
                    mw <<= used; used_all += used;
                    mel_k = b ? (mel_k < 12 ? mel_k + 1 : 12) : (mel_k > 0 ? mel_k - 1 : 0);
                }
            }
            if (!slow) {
                mrb += used_all;                                        /* lands in bytes 0 .. 5: all of 8 bits */
            } else {
                /* byte L of the window is the first one with bits left: cum = bits up to the end of byte j */
                uint32_t L = 0, before = 0, cum = 8 - o;
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    const bool ge = used_all >= cum;
                    L += ge;
                    before = ge ? cum : before;
                    cum += 8 - ((n7 >> (j + 1)) & 1);
                }
                mrb = L ? (uint32_t)(mb + (int)L) * 8 + ((n7 >> L) & 1) + (used_all - before) : mrb + used_all;
            }
            const uint8_t *pm = mraw + min((int)(mrb >> 3), mlim) - 1;
            __builtin_memcpy(&pfA, pm, 4);
            __builtin_memcpy(&pfB, pm + 4, 4);
            __builtin_memcpy(&pfC, pm + 8, 4);
        }
        /* 64 stream bits from vpos on */
        uint32_t lo, hi;
        {
            const uint32_t *wp = (const uint32_t *)((const uint8_t *)vst + ((vpos >> 3) & 60u));
            const uint32_t a0 = wp[0], a1 = wp[1], a2 = wp[2];
            lo = __builtin_amdgcn_alignbit(a1, a0, vpos);
            hi = __builtin_amdgcn_alignbit(a2, a1, vpos);
        }
        uint32_t m = (uint32_t)msyms;                    /* next MEL symbols, LSB first */
        uint32_t e0, e1, mused;
        {
            /* quad 0 */
            const bool en0 = active;
            uint32_t ad = row0 ? ((lo & 0x7Fu) << 1) | (ctx_run << 8)
                               : ((lo & 0x7Fu) << 1) | ((Dlo & 3u) << 8) | lf10;
            const uint32_t mq = (en0 && ad < 0x100u) ? 1u : 0u;
            const uint32_t et = *(const uint16_t *)((row0 ? (const uint8_t *)tbl : tb1) + ad);
            e0 = __umul24(et, (m | ~mq) & (MAIN ? 1u : (uint32_t)en0));          /* context 0 and MEL symbol 0: nothing coded */
            m >>= mq; mused = mq;
            if (row0) {
                const uint32_t f = e0 >> 8, tt = (f | (f >> 1)) & 0x55u;          /* rho of the quad at bits 0, 2, 4, 6 */
                if (en0) ctx_run = ((tt | (tt >> 2)) & 1u) | ((tt >> 3) & 2u) | ((tt >> 4) & 4u);
            }
            /* quad 1 */
            const bool en1 = active && pair;
            const uint32_t len0 = (e0 >> 1) & 7u;
            const uint32_t bits1 = __builtin_amdgcn_ubfe(lo, len0, 7u) << 1;
            ad = row0 ? bits1 | (ctx_run << 8) : bits1 | ((Dlo & 0xCu) << 6) | ((e0 & 0x40u) << 4);
            const uint32_t mq1 = (en1 && ad < 0x100u) ? 1u : 0u;
            const uint32_t et1 = *(const uint16_t *)((row0 ? (const uint8_t *)tbl : tb1) + ad);
            e1 = __umul24(et1, (m | ~mq1) & (MAIN ? 1u : (uint32_t)en1));
            m >>= mq1; mused += mq1;
            if (row0) {
                const uint32_t f = e1 >> 8, tt = (f | (f >> 1)) & 0x55u;
                if (en1) ctx_run = ((tt | (tt >> 2)) & 1u) | ((tt >> 3) & 2u) | ((tt >> 4) & 4u);
            }
            if (MAIN || en1) lf10 = (e1 & 0x40u) << 4;
            else if (en0) lf10 = (e0 & 0x40u) << 4;
            /* this row's (rho bit 1, rho bit 3) pairs go into M from the top, two bits per quad decoded */
            const uint32_t X = ((e0 >> 4) & 3u) | (((e1 >> 4) & 3u) << 2);
            const uint32_t sh = MAIN ? 4u : 2u * ((uint32_t)en0 + (uint32_t)en1);
            Mlo = __builtin_amdgcn_alignbit(Mhi, Mlo, sh);
            Mhi = __builtin_amdgcn_alignbit(X, Mhi, sh);
            Dlo = __builtin_amdgcn_alignbit(Dhi, Dlo, 4u);
            Dhi >>= 4;
        }
        /* U-VLC (jpeg2000htdec.c:338-388, 666-712, 828-854) of both quads: decode order pfx1 pfx2 sfx1 sfx2 ext1 ext2.  First
         * row with both offsets set: one MEL symbol, 1 => both u get +2, 0 and pfx1 > 2 => u2 is a single bit + 1 */
        const uint32_t aused = ((e0 >> 1) & 7u) + ((e1 >> 1) & 7u);
        const uint32_t w = __builtin_amdgcn_alignbit(hi, lo, aused);       /* <= 14 bits in: 32 bits from there, <= 24 are used */
        uint32_t ue = utbl[(w & 63u) | ((e0 & 1u) << 6) | ((e1 & 1u) << 7)];
        uint32_t bias = 0;
        if (!MAIN) {
            const bool mq2 = row0 && (e0 & e1 & 1u);
            if (__ballot(mq2) != 0) {
                const uint32_t mel2 = m & 1u;
                if (mq2) {
                    mused += 1;
                    if (mel2) bias = 0x02000200u; else ue = ht_uvlc_entry2_mode4(w & 63u);
                }
            }
        }
        const uint32_t s1 = __builtin_amdgcn_ubfe(w, ue, __builtin_amdgcn_ubfe(ue, 5u, 3u));
        const uint32_t s2 = __builtin_amdgcn_ubfe(w, ue >> 11, __builtin_amdgcn_ubfe(ue, 16u, 3u));
        uint32_t uused = __builtin_amdgcn_ubfe(ue, 19u, 5u);
        /* fields of quad 0 | u1 << 8 | fields of quad 1 << 16 | u2 << 24 */
        uint32_t out = __builtin_amdgcn_perm(e1, e0, 0x0C050C01u) + (ue & 0x07000700u) + (s1 << 8) + (s2 << 24) + bias;
        if (__ballot(s1 >= 28u || s2 >= 28u) != 0) {                        /* suffix extensions: u > 32 */
            const uint32_t d1x = s1 >= 28u ? 4u : 0u, d2x = s2 >= 28u ? 4u : 0u;
            const uint32_t x1 = __builtin_amdgcn_ubfe(w, uused, d1x), x2 = __builtin_amdgcn_ubfe(w, uused + d1x, d2x);
            out += (x1 << 10) + (x2 << 26);
            uused += d1x + d2x;
        }
        if (active) {
            vpos += aused + uused;
            msyms >>= mused; mcnt -= (int)mused;
        }
        ost[t & (CAD - 1)] = out;
        /* next quad pair of this lane's block */
        qx += 2;
        if (active && qx >= qw) {
            /* the row is done: M holds its 2 qw bits at the top; the next row's (c0, c2) pairs from them */
            const uint64_t I = (((uint64_t)Mhi << 32) | Mlo) >> (64 - 2 * qw);
            const uint64_t Dn = ((I | (I << 1)) & 0x5555555555555555ull) | ((I | (I >> 1)) & 0xAAAAAAAAAAAAAAAAull);
            Dlo = (uint32_t)Dn; Dhi = (uint32_t)(Dn >> 32);
            Mlo = 0; Mhi = 0;
            qx = 0; row++;
            lf10 = 0; ctx_run = 0;
        }
    };
    /* [0, t_a): some lane is still in its first row; [t_a, t_b): the MAIN form; [t_b, max_it): some lane is done */
    int t_a = ppr, t_b = n_it, odd = qw & 1;
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        t_a = max(t_a, __shfl_xor(t_a, o));
        t_b = min(t_b, __shfl_xor(t_b, o));
        odd |= __shfl_xor(odd, o);
    }
    if (odd || t_b < t_a) t_a = t_b = max_it;            /* no MAIN stretch */
    int t = 0;
    for (; t < t_a; t++) pass(t, std::false_type{});
    for (; t < t_b; t++) pass(t, std::true_type{});
    for (; t < max_it; t++) pass(t, std::false_type{});
    if (max_it > 0) flush((max_it - 1) / CAD);
}

}  // namespace htj2k

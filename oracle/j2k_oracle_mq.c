/*
 * j2k_oracle_mq.c -- TEST INFRASTRUCTURE ONLY (part of libj2k_oracle.so, see j2k_oracle.c).
 *
 * CPU restatement of the reference's Part-1 (EBCOT / MQ) block decoder, the path
 * tile_codeblocks() takes for codeblocks without JPEG2000_CTSY_HTJ2K_F:
 *   MQ arithmetic decoder      libavcodec/mqcdec.c:30-111, tables libavcodec/mqc.c:32-71
 *   context labels             libavcodec/jpeg2000.c:91-170, jpeg2000.h:268-290
 *   significance bookkeeping   libavcodec/jpeg2000.c:172-195
 *   the three coding passes    libavcodec/jpeg2000dec.c:1872-1991
 *   decode_cblk()              libavcodec/jpeg2000dec.c:1993-2089
 *   needs_termination()        libavcodec/jpeg2000.h:302-317
 *
 * Parity pinning: no golden vector of the reference covers this path offline (its FATE
 * references need the rsync'd sample suite).  It is pinned by OpenJPEG 2.5.4 (third party,
 * via Pillow), both ways: streams OpenJPEG *encodes* must decode to the source image, and
 * streams the test-vector factory encodes (all mode switches) must decode identically in
 * OpenJPEG and here.  Lossless results are unique, so agreement there is exact.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "j2k_oracle_plan.h"

#define ORC_EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------ MQ decoder */
/* Qe and the two transition tables, indexed by 2 * state + mps exactly as the reference
 * stores them (mqc.c:32-71): generated here from the 47 rows of T.800 Table C.2 */
typedef struct MqRow { uint16_t qe; uint8_t nmps, nlps, sw; } MqRow;
static const MqRow mq_rows[47] = {
    { 0x5601,  1,  1, 1 }, { 0x3401,  2,  6, 0 }, { 0x1801,  3,  9, 0 }, { 0x0ac1,  4, 12, 0 },
    { 0x0521,  5, 29, 0 }, { 0x0221, 38, 33, 0 }, { 0x5601,  7,  6, 1 }, { 0x5401,  8, 14, 0 },
    { 0x4801,  9, 14, 0 }, { 0x3801, 10, 14, 0 }, { 0x3001, 11, 17, 0 }, { 0x2401, 12, 18, 0 },
    { 0x1c01, 13, 20, 0 }, { 0x1601, 29, 21, 0 }, { 0x5601, 15, 14, 1 }, { 0x5401, 16, 14, 0 },
    { 0x5101, 17, 15, 0 }, { 0x4801, 18, 16, 0 }, { 0x3801, 19, 17, 0 }, { 0x3401, 20, 18, 0 },
    { 0x3001, 21, 19, 0 }, { 0x2801, 22, 19, 0 }, { 0x2401, 23, 20, 0 }, { 0x2201, 24, 21, 0 },
    { 0x1c01, 25, 22, 0 }, { 0x1801, 26, 23, 0 }, { 0x1601, 27, 24, 0 }, { 0x1401, 28, 25, 0 },
    { 0x1201, 29, 26, 0 }, { 0x1101, 30, 27, 0 }, { 0x0ac1, 31, 28, 0 }, { 0x09c1, 32, 29, 0 },
    { 0x08a1, 33, 30, 0 }, { 0x0521, 34, 31, 0 }, { 0x0441, 35, 32, 0 }, { 0x02a1, 36, 33, 0 },
    { 0x0221, 37, 34, 0 }, { 0x0141, 38, 35, 0 }, { 0x0111, 39, 36, 0 }, { 0x0085, 40, 37, 0 },
    { 0x0049, 41, 38, 0 }, { 0x0025, 42, 39, 0 }, { 0x0015, 43, 40, 0 }, { 0x0009, 44, 41, 0 },
    { 0x0005, 45, 42, 0 }, { 0x0001, 45, 43, 0 }, { 0x5601, 46, 46, 0 },
};
static uint16_t mq_qe[94];
static uint8_t  mq_nmps[94], mq_nlps[94];
static uint8_t  lut_sigctx[256][4], lut_sgnctx[16][16], lut_xorbit[16][16];
static int      luts_ready;

#define CX_UNI 17
#define CX_RL  18

/* flag bits of jpeg2000.h:85-107 */
#define F_SIG_N  0x0001
#define F_SIG_E  0x0002
#define F_SIG_W  0x0004
#define F_SIG_S  0x0008
#define F_SIG_NE 0x0010
#define F_SIG_NW 0x0020
#define F_SIG_SE 0x0040
#define F_SIG_SW 0x0080
#define F_SIG_NB 0x00ff
#define F_SGN_N  0x0100
#define F_SGN_S  0x0200
#define F_SGN_W  0x0400
#define F_SGN_E  0x0800
#define F_VIS    0x1000
#define F_SIG    0x2000
#define F_REF    0x4000
#define F_SOUTH  (F_SIG_S | F_SIG_SW | F_SIG_SE | F_SGN_S)

#define CBLK_BYPASS  0x01
#define CBLK_RESET   0x02
#define CBLK_TERMALL 0x04
#define CBLK_VSC     0x08
#define CBLK_SEGSYM  0x20

/* getsigctxno(), jpeg2000.c:91-138 */
static int sig_label(int flag, int bandno)
{
    int h = !!(flag & F_SIG_E) + !!(flag & F_SIG_W);
    int v = !!(flag & F_SIG_N) + !!(flag & F_SIG_S);
    int d = !!(flag & F_SIG_NE) + !!(flag & F_SIG_NW) + !!(flag & F_SIG_SE) + !!(flag & F_SIG_SW);
    if (bandno < 3) {
        if (bandno == 1) { int t = h; h = v; v = t; }
        if (h == 2) return 8;
        if (h == 1) return v >= 1 ? 7 : d >= 1 ? 6 : 5;
        if (v == 2) return 4;
        if (v == 1) return 3;
        if (d >= 2) return 2;
        return d == 1;
    }
    if (d >= 3) return 8;
    if (d == 2) return h + v >= 1 ? 7 : 6;
    if (d == 1) return h + v >= 2 ? 5 : h + v == 1 ? 4 : 3;
    return h + v >= 2 ? 2 : h + v == 1;
}

/* getsgnctxno(), jpeg2000.c:140-158 */
static int sgn_label(int flag, uint8_t *xorbit)
{
    static const int contrib[3][3] = { { 0, -1, 1 }, { -1, -1, 0 }, { 1, 0, 1 } };
    static const int label[3][3]   = { { 13, 12, 11 }, { 10, 9, 10 }, { 11, 12, 13 } };
    static const int xbit[3][3]    = { { 1, 1, 1 }, { 1, 0, 0 }, { 0, 0, 0 } };
    int hc = contrib[flag & F_SIG_E ? flag & F_SGN_E ? 1 : 2 : 0][flag & F_SIG_W ? flag & F_SGN_W ? 1 : 2 : 0] + 1;
    int vc = contrib[flag & F_SIG_S ? flag & F_SGN_S ? 1 : 2 : 0][flag & F_SIG_N ? flag & F_SGN_N ? 1 : 2 : 0] + 1;
    *xorbit = (uint8_t)xbit[hc][vc];
    return label[hc][vc];
}

static void luts_build(void)
{
    int i, j;
    if (luts_ready) return;
    for (i = 0; i < 47; i++) {
        /* entry 2i: mps = 0, entry 2i+1: mps = 1; an LPS on a "switch" row flips the mps */
        for (j = 0; j < 2; j++) {
            mq_qe[2 * i + j]   = mq_rows[i].qe;
            mq_nmps[2 * i + j] = (uint8_t)(2 * mq_rows[i].nmps + j);
            mq_nlps[2 * i + j] = (uint8_t)(2 * mq_rows[i].nlps + (mq_rows[i].sw ? 1 - j : j));
        }
    }
    for (i = 0; i < 256; i++)
        for (j = 0; j < 4; j++)
            lut_sigctx[i][j] = (uint8_t)sig_label(i, j);
    for (i = 0; i < 16; i++)
        for (j = 0; j < 16; j++)
            lut_sgnctx[i][j] = (uint8_t)sgn_label(i + (j << 8), &lut_xorbit[i][j]);
    luts_ready = 1;
}

ORC_EXPORT void orc_mq_tables(uint16_t *qe, uint8_t *nmps, uint8_t *nlps)
{
    luts_build();
    memcpy(qe, mq_qe, sizeof(mq_qe)); memcpy(nmps, mq_nmps, 94); memcpy(nlps, mq_nlps, 94);
}

typedef struct Mq {
    const uint8_t *bp;
    uint32_t a, c;
    uint8_t cx[19];
    int raw;
} Mq;

static void mq_reset_contexts(Mq *m)               /* ff_mqc_init_contexts, mqc.c:73-79 */
{
    memset(m->cx, 0, sizeof(m->cx));
    m->cx[CX_UNI] = 2 * 46;
    m->cx[CX_RL]  = 2 * 3;
    m->cx[0]      = 2 * 4;
}

static void mq_bytein(Mq *m)                        /* mqcdec.c:30-43 */
{
    if (*m->bp == 0xff) {
        if (m->bp[1] > 0x8f) {
            m->c++;
        } else {
            m->bp++;
            m->c += 2 + 0xfe00 - ((uint32_t)*m->bp << 9);
        }
    } else {
        m->bp++;
        m->c += 1 + 0xff00 - ((uint32_t)*m->bp << 8);
    }
}

static void mq_init(Mq *m, const uint8_t *bp, int raw, int reset)   /* ff_mqc_initdec, mqcdec.c:73-83 */
{
    m->raw = raw;
    if (reset)
        mq_reset_contexts(m);
    m->bp = bp;
    m->c  = (uint32_t)(*m->bp ^ 0xff) << 16;
    mq_bytein(m);
    m->c <<= 7;
    m->a  = 0x8000;
}

static int mq_renorm(Mq *m, uint8_t *cx, int lps)  /* exchange(), mqcdec.c:45-71 */
{
    int d;
    if ((m->a < mq_qe[*cx]) ^ (!lps)) {
        if (lps) m->a = mq_qe[*cx];
        d = *cx & 1;
        *cx = mq_nmps[*cx];
    } else {
        if (lps) m->a = mq_qe[*cx];
        d = 1 - (*cx & 1);
        *cx = mq_nlps[*cx];
    }
    do {
        if (!(m->c & 0xff)) {
            m->c -= 0x100;
            mq_bytein(m);
        }
        m->a += m->a;
        m->c += m->c;
    } while (!(m->a & 0x8000));
    return d;
}

static int mq_decode(Mq *m, int ctx)                /* ff_mqc_decode + mqc_decode_bypass, mqcdec.c:85-111 */
{
    uint8_t *cx = m->cx + ctx;
    if (m->raw) {
        int bit = !(m->c & 0x40000000);
        if (!(m->c & 0xff)) {
            m->c -= 0x100;
            mq_bytein(m);
        }
        m->c += m->c;
        return bit;
    }
    m->a -= mq_qe[*cx];
    if ((m->c >> 16) < m->a) {
        if (m->a & 0x8000)
            return *cx & 1;
        return mq_renorm(m, cx, 0);
    }
    m->c -= m->a << 16;
    return mq_renorm(m, cx, 1);
}

/* ------------------------------------------------------------------ coding passes */
typedef struct T1 {
    int32_t  *data;          /* w x h, stride = dstride */
    uint16_t *flags;         /* (w + 2) x (h + 2), stride = fstride; sample (x, y) lives at (x + 1, y + 1) */
    int dstride, fstride;
    Mq mq;
} T1;

static void set_significant(T1 *t, int x, int y, int negative)   /* jpeg2000.c:172-195 */
{
    uint16_t *f = t->flags + (y + 1) * t->fstride + (x + 1);
    f[0] |= F_SIG;
    f[1]               |= negative ? F_SIG_W | F_SGN_W : F_SIG_W;
    f[-1]              |= negative ? F_SIG_E | F_SGN_E : F_SIG_E;
    f[t->fstride]      |= negative ? F_SIG_N | F_SGN_N : F_SIG_N;
    f[-t->fstride]     |= negative ? F_SIG_S | F_SGN_S : F_SIG_S;
    f[t->fstride + 1]  |= F_SIG_NW;
    f[t->fstride - 1]  |= F_SIG_NE;
    f[-t->fstride + 1] |= F_SIG_SW;
    f[-t->fstride - 1] |= F_SIG_SE;
}

static int sign_ctx(int flag, int *xorbit)
{
    *xorbit = lut_xorbit[flag & 15][(flag >> 8) & 15];
    return lut_sgnctx[flag & 15][(flag >> 8) & 15];
}

static void pass_sig(T1 *t, int w, int h, int bpno, int bandno, int vsc)   /* decode_sigpass, jpeg2000dec.c:1872-1905 */
{
    const int32_t mask = (int32_t)(3u << (bpno - 1));
    int y0, x, y;
    for (y0 = 0; y0 < h; y0 += 4)
        for (x = 0; x < w; x++)
            for (y = y0; y < h && y < y0 + 4; y++) {
                uint16_t *f = t->flags + (y + 1) * t->fstride + x + 1;
                int keep = (vsc && y == y0 + 3) ? ~F_SOUTH : -1;
                if ((*f & F_SIG_NB & keep) && !(*f & (F_SIG | F_VIS))) {
                    if (mq_decode(&t->mq, lut_sigctx[*f & keep & 255][bandno])) {
                        int xorbit, ctx = sign_ctx(*f & keep, &xorbit);
                        int32_t *d = t->data + y * t->dstride + x;
                        uint32_t s = (uint32_t)mq_decode(&t->mq, ctx);
                        if (!t->mq.raw)
                            s ^= (uint32_t)xorbit;
                        *d |= (int32_t)(s << 31);
                        *d |= mask;
                        set_significant(t, x, y, *d & INT32_MIN);
                    }
                    *f |= F_VIS;
                }
            }
}

static void pass_ref(T1 *t, int w, int h, int bpno, int vsc)      /* decode_refpass, jpeg2000dec.c:1907-1932 */
{
    const int32_t phalf = (int32_t)(1u << (bpno - 1));
    int y0, x, y;
    for (y0 = 0; y0 < h; y0 += 4)
        for (x = 0; x < w; x++)
            for (y = y0; y < h && y < y0 + 4; y++) {
                uint16_t *f = t->flags + (y + 1) * t->fstride + x + 1;
                if ((*f & (F_SIG | F_VIS)) == F_SIG) {
                    int keep = (vsc && y == y0 + 3) ? ~F_SOUTH : -1;
                    int fl = *f & keep;
                    int ctx = (fl & F_REF) ? 16 : (fl & 255) ? 15 : 14;    /* refctxno_lut, jpeg2000.h:273-280 */
                    int32_t *d = t->data + y * t->dstride + x;
                    *d |= phalf;
                    if (mq_decode(&t->mq, ctx))
                        *d |= (int32_t)((uint32_t)phalf << 1);
                    else
                        *d &= ~(int32_t)((uint32_t)phalf << 1);
                    *f |= F_REF;
                }
            }
}

/* decode_clnpass, jpeg2000dec.c:1934-1991.  Returns 1 when the segmentation symbol was wrong (logged only). */
static int pass_cln(T1 *t, int w, int h, int bpno, int bandno, int segsym, int vsc)
{
    const int32_t mask = (int32_t)(3u << (bpno - 1));
    int y0, x, y, runlen, dec;
    for (y0 = 0; y0 < h; y0 += 4)
        for (x = 0; x < w; x++) {
            const uint16_t *c = t->flags + (y0 + 1) * t->fstride + x + 1;
            const int quiet = F_SIG_NB | F_VIS | F_SIG;
            int keep4 = vsc ? ~F_SOUTH : -1;
            if (y0 + 3 < h &&
                !((c[0] & quiet) || (c[t->fstride] & quiet) || (c[2 * t->fstride] & quiet) ||
                  (c[3 * t->fstride] & quiet & keep4))) {
                if (!mq_decode(&t->mq, CX_RL))
                    continue;
                runlen = mq_decode(&t->mq, CX_UNI);
                runlen = (runlen << 1) | mq_decode(&t->mq, CX_UNI);
                dec = 1;
            } else {
                runlen = 0;
                dec = 0;
            }
            for (y = y0 + runlen; y < y0 + 4 && y < h; y++) {
                uint16_t *f = t->flags + (y + 1) * t->fstride + x + 1;
                int keep = (vsc && y == y0 + 3) ? ~F_SOUTH : -1;
                if (!dec && !(*f & (F_SIG | F_VIS)))
                    dec = mq_decode(&t->mq, lut_sigctx[*f & keep & 255][bandno]);
                if (dec) {
                    int xorbit, ctx = sign_ctx(*f & keep, &xorbit);
                    int32_t *d = t->data + y * t->dstride + x;
                    *d |= (int32_t)((uint32_t)(mq_decode(&t->mq, ctx) ^ xorbit) << 31);
                    *d |= mask;
                    set_significant(t, x, y, *d & INT32_MIN);
                }
                dec = 0;
                *f &= (uint16_t)~F_VIS;
            }
        }
    if (segsym) {
        int val = mq_decode(&t->mq, CX_UNI);
        val = (val << 1) + mq_decode(&t->mq, CX_UNI);
        val = (val << 1) + mq_decode(&t->mq, CX_UNI);
        val = (val << 1) + mq_decode(&t->mq, CX_UNI);
        return val != 0xa;
    }
    return 0;
}

static int needs_termination(int style, int passno)   /* jpeg2000.h:302-317 */
{
    if (style & CBLK_BYPASS) {
        int type = passno % 3;
        passno /= 3;
        if (type == 0 && passno > 2)
            return 2;
        if (type == 2 && passno > 2)
            return 1;
        if (style & CBLK_TERMALL)
            return passno > 2 ? 2 : 1;
    }
    if (style & CBLK_TERMALL)
        return 1;
    return 0;
}

/* decode_cblk, jpeg2000dec.c:1993-2089.
 * data: the block's bytes as the reference accumulates them (segments of all packets back to back, 0xFF 0xFF
 * after every terminated segment, jpeg2000dec.c:1508-1516); data[length], data[length + 1] must be writable.
 * data_start[1..nterm]: where the segment after the k-th termination starts.
 * out (stride out_stride) receives t1->data: sign-magnitude, binary point at 31 - M_b.
 * Returns 1 coded, 0 nothing to do, < 0 error -- in which case `out` holds the passes decoded so far, which is
 * what the reference goes on to dequantise (jpeg2000dec.c:2275-2290: `if (ret) coded = 1`). */
ORC_EXPORT int orc_mq_decode_block(uint8_t *data, int length, int npasses, int nonzerobits, int width, int height,
                                   int M_b, int roi_shift, int style, int bandpos, int nterm, const uint16_t *data_start,
                                   int32_t *out, int out_stride)
{
    T1 t;
    int passno = npasses, pass_t = 2, bpno = nonzerobits - 1 + 31 - M_b - 1 - roi_shift;
    int pass_cnt = 0, term_cnt = 0, coder, ret = 1, x, y;
    const int vsc = style & CBLK_VSC;

    luts_build();
    for (y = 0; y < height; y++)
        memset(out + (size_t)y * out_stride, 0, (size_t)width * sizeof(*out));
    if (!length)
        return 0;
    t.data = out; t.dstride = out_stride;
    t.fstride = width + 2;
    t.flags = (uint16_t *)calloc((size_t)(width + 2) * (height + 2), sizeof(uint16_t));
    if (!t.flags)
        return HTJ2K_ERR_ENOMEM;

    data[length] = 0xff;
    data[length + 1] = 0xff;
    mq_init(&t.mq, data, 0, 1);

    while (passno--) {
        if (bpno < 0 || bpno > 29) {
            ret = HTJ2K_ERR_INVALIDDATA;           /* "bpno became invalid" */
            goto done;
        }
        switch (pass_t) {
        case 0: pass_sig(&t, width, height, bpno + 1, bandpos, vsc); break;
        case 1: pass_ref(&t, width, height, bpno + 1, vsc); break;
        case 2: pass_cln(&t, width, height, bpno + 1, bandpos, style & CBLK_SEGSYM, vsc); break;
        }
        if (style & CBLK_RESET)
            mq_reset_contexts(&t.mq);
        if (passno && (coder = needs_termination(style, pass_cnt))) {
            if (term_cnt >= nterm) {
                ret = HTJ2K_ERR_INVALIDDATA;       /* "Missing needed termination" */
                goto done;
            }
            mq_init(&t.mq, data + data_start[++term_cnt], coder == 2, 0);
        }
        if (++pass_t == 3) {
            bpno--;
            pass_t = 0;
        }
        pass_cnt++;
    }
    /* jpeg2000dec.c:2071-2086: the ROI up-shift of samples below the ROI threshold */
    if (roi_shift) {
        const uint32_t below = UINT32_MAX >> (M_b + 1);
        for (y = 0; y < height; y++)
            for (x = 0; x < width; x++) {
                int32_t v = out[x + (size_t)y * out_stride], sign = v & INT32_MIN;
                v &= INT32_MAX;
                if (((uint32_t)v & ~below) == 0)
                    v = (int32_t)((uint32_t)v << roi_shift);
                out[x + (size_t)y * out_stride] = v | sign;
            }
    }
done:
    free(t.flags);
    return ret;
}

/*
 * j2k_syntax.c -- codestream syntax of the host front-end: JP2 box wrapper, marker
 * segments of the main and tile-part headers, choice of the output sample layout.
 *
 * Behaviour follows the reference decoder -- what it accepts, what it rejects and with
 * which AVERROR, which bytes it looks at (libavcodec/jpeg2000dec.c:197-1014 marker
 * segments, :2425-2637 header loop, :2658-2805 JP2 boxes, :133-193,330-420 pixel
 * formats) -- the structure does not: marker segments are dispatched through one rule
 * table, every handler works on the resolved per-component parameter sets of j2k_host.h,
 * tile-part bodies are only located (no byte of them is read here), and TLM / PLT
 * lengths are kept (the reference reads and drops them, :901-956).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "j2k_host.h"

/* marker codes, T.800 Table A.2 / T.814 A.1 */
enum {
    MK_SOC = 0xFF4F, MK_CAP = 0xFF50, MK_SIZ = 0xFF51, MK_COD = 0xFF52, MK_COC = 0xFF53,
    MK_TLM = 0xFF55, MK_PLM = 0xFF57, MK_PLT = 0xFF58, MK_CPF = 0xFF59, MK_QCD = 0xFF5C,
    MK_QCC = 0xFF5D, MK_RGN = 0xFF5E, MK_POC = 0xFF5F, MK_PPM = 0xFF60, MK_PPT = 0xFF61,
    MK_CRG = 0xFF63, MK_COM = 0xFF64, MK_SOT = 0xFF90, MK_SOD = 0xFF93, MK_EOC = 0xFFD9
};

void cs_log(J2kParser *ps, int level, const char *fmt, ...)
{
    char line[400];
    va_list ap;
    if (!ps->log)
        return;
    va_start(ap, fmt);
    vsnprintf(line, sizeof line, fmt, ap);
    va_end(ap);
    ps->log(ps->log_opaque, level, line);
}

/* ------------------------------------------------------------------ bump allocator */
void *pool_get(Pool *a, size_t n, int zeroed)
{
    Slab *s;
    n = (n + 15) & ~(size_t)15;
    for (s = a->cur; s; s = s->next) {
        if (s != a->cur)
            s->used = 0;
        if (s->cap - s->used >= n) {
            void *r = (uint8_t *)(s + 1) + s->used;
            s->used += n;
            a->cur = s;
            if (zeroed)
                memset(r, 0, n);
            return r;
        }
    }
    {
        const size_t cap = n > ((size_t)1 << 20) ? n : ((size_t)1 << 20);
        Slab *fresh = (Slab *)malloc(sizeof(Slab) + cap), *tail = a->cur;
        if (!fresh)
            return NULL;
        fresh->next = NULL; fresh->cap = cap; fresh->used = n;
        while (tail && tail->next)
            tail = tail->next;
        if (tail) tail->next = fresh; else a->first = fresh;
        a->cur = fresh;
        if (zeroed)
            memset(fresh + 1, 0, n);
        return fresh + 1;
    }
}

void pool_rewind(Pool *a)
{
    a->cur = a->first;
    if (a->cur)
        a->cur->used = 0;
}

void pool_destroy(Pool *a)
{
    Slab *s = a->first;
    while (s) {
        Slab *n = s->next;
        free(s);
        s = n;
    }
    a->first = a->cur = NULL;
}

int cs_sig_append(J2kParser *ps, const void *data, size_t n)
{
    if (ps->sig_len + n > ps->sig_cap) {
        size_t cap = ps->sig_cap ? ps->sig_cap * 2 : 1024;
        uint8_t *nb;
        while (cap < ps->sig_len + n)
            cap *= 2;
        nb = (uint8_t *)realloc(ps->sig, cap);
        if (!nb)
            return HTJ2K_ERR_ENOMEM;
        ps->sig = nb;
        ps->sig_cap = cap;
    }
    memcpy(ps->sig + ps->sig_len, data, n);
    ps->sig_len += n;
    return 0;
}

/* ------------------------------------------------------------------ sample layouts
 * What the glue maps 1:1 onto AV_PIX_FMT_* (libavutil/pixdesc.c entries of the formats
 * jpeg2000dec.c:170-193 lists). */
#define PACKED(name, nc, bits, bytes)             { name, nc, 0, 0, 0, 0, { bits, (nc) > 1 ? bits : 0, (nc) > 2 ? bits : 0, (nc) > 3 ? bits : 0 }, 1, bytes }
#define PLANAR(name, nc, cw, ch, bits, bytes)     { name, nc, cw, ch, 1, 0, { bits, bits, bits, (nc) > 3 ? bits : 0 }, nc, bytes }
static const J2kPixDesc layout_table[HTJ2K_PIX_NB] = {
    [HTJ2K_PIX_PAL8]       = { "pal8", 1, 0, 0, 0, 1, { 8, 0, 0, 0 }, 2, 1 },
    [HTJ2K_PIX_RGB24]      = PACKED("rgb24", 3, 8, 1),
    [HTJ2K_PIX_RGBA]       = PACKED("rgba", 4, 8, 1),
    [HTJ2K_PIX_RGB48]      = PACKED("rgb48le", 3, 16, 2),
    [HTJ2K_PIX_RGBA64]     = PACKED("rgba64le", 4, 16, 2),
    [HTJ2K_PIX_GRAY8]      = PACKED("gray", 1, 8, 1),
    [HTJ2K_PIX_YA8]        = PACKED("ya8", 2, 8, 1),
    [HTJ2K_PIX_GRAY16]     = PACKED("gray16le", 1, 16, 2),
    [HTJ2K_PIX_YA16]       = PACKED("ya16le", 2, 16, 2),
    [HTJ2K_PIX_XYZ12]      = PACKED("xyz12le", 3, 12, 2),
    [HTJ2K_PIX_YUV410P]    = PLANAR("yuv410p", 3, 2, 2, 8, 1),
    [HTJ2K_PIX_YUV411P]    = PLANAR("yuv411p", 3, 2, 0, 8, 1),
    [HTJ2K_PIX_YUVA420P]   = PLANAR("yuva420p", 4, 1, 1, 8, 1),
    [HTJ2K_PIX_YUV420P]    = PLANAR("yuv420p", 3, 1, 1, 8, 1),
    [HTJ2K_PIX_YUV422P]    = PLANAR("yuv422p", 3, 1, 0, 8, 1),
    [HTJ2K_PIX_YUVA422P]   = PLANAR("yuva422p", 4, 1, 0, 8, 1),
    [HTJ2K_PIX_YUV440P]    = PLANAR("yuv440p", 3, 0, 1, 8, 1),
    [HTJ2K_PIX_YUV444P]    = PLANAR("yuv444p", 3, 0, 0, 8, 1),
    [HTJ2K_PIX_YUVA444P]   = PLANAR("yuva444p", 4, 0, 0, 8, 1),
    [HTJ2K_PIX_YUV420P9]   = PLANAR("yuv420p9le", 3, 1, 1, 9, 2),
    [HTJ2K_PIX_YUV422P9]   = PLANAR("yuv422p9le", 3, 1, 0, 9, 2),
    [HTJ2K_PIX_YUV444P9]   = PLANAR("yuv444p9le", 3, 0, 0, 9, 2),
    [HTJ2K_PIX_YUVA420P9]  = PLANAR("yuva420p9le", 4, 1, 1, 9, 2),
    [HTJ2K_PIX_YUVA422P9]  = PLANAR("yuva422p9le", 4, 1, 0, 9, 2),
    [HTJ2K_PIX_YUVA444P9]  = PLANAR("yuva444p9le", 4, 0, 0, 9, 2),
    [HTJ2K_PIX_YUV420P10]  = PLANAR("yuv420p10le", 3, 1, 1, 10, 2),
    [HTJ2K_PIX_YUV422P10]  = PLANAR("yuv422p10le", 3, 1, 0, 10, 2),
    [HTJ2K_PIX_YUV444P10]  = PLANAR("yuv444p10le", 3, 0, 0, 10, 2),
    [HTJ2K_PIX_YUVA420P10] = PLANAR("yuva420p10le", 4, 1, 1, 10, 2),
    [HTJ2K_PIX_YUVA422P10] = PLANAR("yuva422p10le", 4, 1, 0, 10, 2),
    [HTJ2K_PIX_YUVA444P10] = PLANAR("yuva444p10le", 4, 0, 0, 10, 2),
    [HTJ2K_PIX_YUV420P12]  = PLANAR("yuv420p12le", 3, 1, 1, 12, 2),
    [HTJ2K_PIX_YUV422P12]  = PLANAR("yuv422p12le", 3, 1, 0, 12, 2),
    [HTJ2K_PIX_YUV444P12]  = PLANAR("yuv444p12le", 3, 0, 0, 12, 2),
    [HTJ2K_PIX_YUV420P14]  = PLANAR("yuv420p14le", 3, 1, 1, 14, 2),
    [HTJ2K_PIX_YUV422P14]  = PLANAR("yuv422p14le", 3, 1, 0, 14, 2),
    [HTJ2K_PIX_YUV444P14]  = PLANAR("yuv444p14le", 3, 0, 0, 14, 2),
    [HTJ2K_PIX_YUV420P16]  = PLANAR("yuv420p16le", 3, 1, 1, 16, 2),
    [HTJ2K_PIX_YUV422P16]  = PLANAR("yuv422p16le", 3, 1, 0, 16, 2),
    [HTJ2K_PIX_YUV444P16]  = PLANAR("yuv444p16le", 3, 0, 0, 16, 2),
    [HTJ2K_PIX_YUVA420P16] = PLANAR("yuva420p16le", 4, 1, 1, 16, 2),
    [HTJ2K_PIX_YUVA422P16] = PLANAR("yuva422p16le", 4, 1, 0, 16, 2),
    [HTJ2K_PIX_YUVA444P16] = PLANAR("yuva444p16le", 4, 0, 0, 16, 2),
};

const J2kPixDesc *j2k_pix_desc(int pix_fmt)
{
    return pix_fmt >= 0 && pix_fmt < HTJ2K_PIX_NB ? &layout_table[pix_fmt] : NULL;
}

/* Does a layout fit the components of the SIZ segment?  (pix_fmt_match, jpeg2000dec.c:133-166:
 * same component count, every channel at least as deep as the deepest component, components 0
 * and 3 at full resolution, components 1 and 2 subsampled exactly like the layout's chroma,
 * palette layouts only for palettised files.)  sub[c] = log2 of (XRsiz, YRsiz), two bits each. */
static int layout_fits(int fmt, int ncomp, int bits, const uint8_t sub[J2K_MAX_COMPS][2], int palettised)
{
    const J2kPixDesc *d = j2k_pix_desc(fmt);
    int c;
    if (!d || d->nb_components != ncomp || d->pal != palettised)
        return 0;
    for (c = 0; c < ncomp; c++) {
        const int chroma = c == 1 || c == 2;
        if (d->depth[c] < bits)
            return 0;
        if (sub[c][0] != (chroma ? d->log2_chroma_w : 0) || sub[c][1] != (chroma ? d->log2_chroma_h : 0))
            return 0;
    }
    return 1;
}

/* the reference searches its candidate lists front to back (jpeg2000dec.c:170-193, 354-372);
 * a class is a contiguous range of the enum, XYZ12 goes in front for digital-cinema profiles */
static int pick_layout(J2kParser *ps, const uint8_t sub[J2K_MAX_COMPS][2])
{
    int lo = 0, hi = HTJ2K_PIX_NB - 1, f, try_xyz_first = 0, with_xyz = 1;
    if (ps->rsiz == 3 || ps->rsiz == 4) {                 /* AV_PROFILE_JPEG2000_DCINEMA_2K / _4K */
        lo = HTJ2K_PIX_YUV410P; hi = HTJ2K_PIX_YUVA444P16; try_xyz_first = 1; with_xyz = 0;
    } else if (ps->colour_space == 16) {                  /* sRGB */
        lo = HTJ2K_PIX_PAL8; hi = HTJ2K_PIX_RGBA64; with_xyz = 0;
    } else if (ps->colour_space == 17) {                  /* greyscale */
        lo = HTJ2K_PIX_GRAY8; hi = HTJ2K_PIX_YA16; with_xyz = 0;
    } else if (ps->colour_space == 18) {                  /* sYCC */
        lo = HTJ2K_PIX_YUV410P; hi = HTJ2K_PIX_YUVA444P16; with_xyz = 0;
    }
    if (ps->opts.req_pix_fmt != HTJ2K_PIX_NONE &&
        layout_fits(ps->opts.req_pix_fmt, ps->ncomp, ps->precision, sub, ps->palettised))
        return ps->opts.req_pix_fmt;
    if (try_xyz_first && layout_fits(HTJ2K_PIX_XYZ12, ps->ncomp, ps->precision, sub, ps->palettised))
        return HTJ2K_PIX_XYZ12;
    for (f = lo; f <= hi; f++)
        if (f != HTJ2K_PIX_XYZ12 && layout_fits(f, ps->ncomp, ps->precision, sub, ps->palettised))
            return f;
    if (with_xyz && layout_fits(HTJ2K_PIX_XYZ12, ps->ncomp, ps->precision, sub, ps->palettised))
        return HTJ2K_PIX_XYZ12;
    return HTJ2K_PIX_NONE;
}

/* the shapes the reference still accepts when no listed layout fits (jpeg2000dec.c:374-414) */
static int pick_layout_by_shape(J2kParser *ps)
{
    const int *dx = ps->sub_x, *dy = ps->sub_y, n = ps->ncomp, bits = ps->precision;
    int same01 = n >= 2 && dx[0] == dx[1] && dy[0] == dy[1];
    if (n == 4 && dx[0] == 1 && dy[0] == 1 && dx[1] == 1 && dy[1] == 1 && dx[2] == dx[3] && dy[2] == dy[3]) {
        if (bits == 8 && dx[2] == 2 && dy[2] == 2 && !ps->palettised) {
            int c;
            for (c = 0; c < 4; c++)
                ps->cdef[c] = c;
            return HTJ2K_PIX_YUVA420P;
        }
        return HTJ2K_PIX_NONE;
    }
    if (n == 3 && bits == 8 && same01 && dx[0] == dx[2] && dy[0] == dy[2]) return HTJ2K_PIX_RGB24;
    if (n == 2 && bits == 8 && same01)  return HTJ2K_PIX_YA8;
    if (n == 2 && bits == 16 && same01) return HTJ2K_PIX_YA16;
    if (n == 1 && bits == 8)            return HTJ2K_PIX_GRAY8;
    if (n == 1 && bits == 12)           return HTJ2K_PIX_GRAY16;
    return HTJ2K_PIX_NONE;
}

/* av_image_check_size2() without a pixel format (libavutil/imgutils.c:289-316) */
int cs_picture_size_ok(uint32_t w, uint32_t h, int64_t max_pixels)
{
    const int64_t row = 8 * (int64_t)w + 1024;
    if (!w || !h || w > INT32_MAX || h > INT32_MAX || row >= INT_MAX || (uint64_t)row * (h + (uint64_t)128) >= INT_MAX)
        return 0;
    return !(max_pixels < INT64_MAX && (int64_t)w * h > max_pixels);
}


void cs_fill_info(const J2kParser *ps, htj2k_info *info)
{
    const J2kPixDesc *d = j2k_pix_desc(ps->pix_fmt);
    int p;
    memset(info, 0, sizeof *info);
    info->width = ps->out_w;
    info->height = ps->out_h;
    info->pix_fmt = ps->pix_fmt;
    info->bits_per_raw_sample = ps->precision;
    info->profile = ps->rsiz;
    info->lossless = ps->lossless;
    info->sar_num = ps->sar_num;
    info->sar_den = ps->sar_den;
    info->ncomponents = ps->ncomp;
    info->is_ht = ps->is_ht;
    info->has_palette = ps->pix_fmt == HTJ2K_PIX_PAL8;
    if (!d)
        return;
    info->nplanes = d->nplanes;
    for (p = 0; p < d->nplanes; p++) {
        const int chroma = p == 1 || p == 2;
        if (d->pal && p == 1) {                /* AVFrame.data[1] of pal8: 256 entries of 0xAARRGGBB */
            info->plane_width[p] = 256;
            info->plane_height[p] = 1;
            info->plane_bytes_per_sample[p] = 4;
            continue;
        }
        info->plane_width[p]  = d->planar && chroma ? cdiv_pow2(ps->out_w, d->log2_chroma_w) : ps->out_w;
        info->plane_height[p] = d->planar && chroma ? cdiv_pow2(ps->out_h, d->log2_chroma_h) : ps->out_h;
        info->plane_bytes_per_sample[p] = d->bytes * (d->planar ? 1 : d->nb_components);
    }
}

/* ------------------------------------------------------------------ marker segments
 * A handler reads from the scanner's cursor and may run past its segment, as the
 * reference's get_* functions do; the dispatcher compares what was consumed with Lxxx
 * afterwards (jpeg2000dec.c:2626-2633).  `lseg` is the length field. */
typedef int (*SegHandler)(J2kParser *ps, Cur *c, int lseg);

/* parameter sets a segment of the current header writes to; a tile whose own headers carry any is
 * remembered as such (the geometry signature then holds its parameters, j2k_tier2.c) */
static CompCoding *coding_target(J2kParser *ps)
{
    if (ps->cur_tile < 0) return ps->cod;
    ps->tile[ps->cur_tile].own_params |= TILE_OWN_PARAMS;
    return ps->tile[ps->cur_tile].cod;
}
static CompQuant *quant_target(J2kParser *ps)
{
    if (ps->cur_tile < 0) return ps->q;
    ps->tile[ps->cur_tile].own_params |= TILE_OWN_PARAMS;
    return ps->tile[ps->cur_tile].q;
}
static uint8_t *seen_target(J2kParser *ps) { return ps->cur_tile < 0 ? ps->seen : ps->tile[ps->cur_tile].seen; }

static int seg_siz(J2kParser *ps, Cur *c, int lseg)
{
    uint8_t sub[J2K_MAX_COMPS][2];
    uint32_t ntiles;
    int i, n, rw, rh;
    (void)lseg;
    if (cur_left(c) < 36) {
        cs_log(ps, LOGL_ERROR, "SIZ segment cut short\n");
        return HTJ2K_ERR_INVALIDDATA;
    }
    ps->rsiz   = (int)ld_be16(c->p);
    ps->xsiz   = (int32_t)ld_be32(c->p + 2);   ps->ysiz   = (int32_t)ld_be32(c->p + 6);
    ps->xosiz  = (int32_t)ld_be32(c->p + 10);  ps->yosiz  = (int32_t)ld_be32(c->p + 14);
    ps->xtsiz  = (int32_t)ld_be32(c->p + 18);  ps->ytsiz  = (int32_t)ld_be32(c->p + 22);
    ps->xtosiz = (int32_t)ld_be32(c->p + 26);  ps->ytosiz = (int32_t)ld_be32(c->p + 30);
    n = (int)ld_be16(c->p + 34);
    c->p += 36;

    if (!cs_picture_size_ok((uint32_t)ps->xsiz, (uint32_t)ps->ysiz, pixel_budget(ps))) {
        cs_log(ps, LOGL_ERROR, "reference grid %ux%u is more than this decoder handles\n", (unsigned)ps->xsiz, (unsigned)ps->ysiz);
        return HTJ2K_ERR_PATCHWELCOME;
    }
    if (n == 0) {
        cs_log(ps, LOGL_ERROR, "SIZ declares no components\n");
        return HTJ2K_ERR_INVALIDDATA;
    }
    if (n > J2K_MAX_COMPS) {
        cs_log(ps, LOGL_ERROR, "%d components: at most %d are supported\n", n, J2K_MAX_COMPS);
        return HTJ2K_ERR_PATCHWELCOME;
    }
    if (ps->xtosiz < 0 || ps->ytosiz < 0 || ps->xosiz < ps->xtosiz || ps->yosiz < ps->ytosiz ||
        (int64_t)ps->xtsiz + ps->xtosiz <= ps->xosiz || (int64_t)ps->ytsiz + ps->ytosiz <= ps->yosiz) {
        cs_log(ps, LOGL_ERROR, "tile grid origin does not cover the image origin\n");
        return HTJ2K_ERR_INVALIDDATA;
    }
    if (ps->xosiz >= ps->xsiz || ps->yosiz >= ps->ysiz) {
        cs_log(ps, LOGL_ERROR, "image origin lies outside the reference grid\n");
        return HTJ2K_ERR_INVALIDDATA;
    }
    if (ps->reduce && (ps->xosiz || ps->yosiz)) {
        cs_log(ps, LOGL_ERROR, "lowres decoding of an image with an origin offset is not supported\n");
        return HTJ2K_ERR_PATCHWELCOME;
    }
    ps->ncomp = n;
    if (ps->xtsiz <= 0 || ps->ytsiz <= 0) {
        cs_log(ps, LOGL_ERROR, "tile size %dx%d\n", ps->xtsiz, ps->ytsiz);
        return HTJ2K_ERR_INVALIDDATA;
    }
    if (cur_left(c) < 3 * n) {
        cs_log(ps, LOGL_ERROR, "SIZ segment cut short in its component list\n");
        return HTJ2K_ERR_INVALIDDATA;
    }
    for (i = 0; i < n; i++, c->p += 3) {
        const int dx = c->p[1], dy = c->p[2];
        ps->depth[i] = (uint8_t)((c->p[0] & 0x7F) + 1);
        ps->is_signed[i] = c->p[0] >> 7;
        if (ps->depth[i] > ps->precision)
            ps->precision = ps->depth[i];
        ps->sub_x[i] = dx;
        ps->sub_y[i] = dy;
        if ((dx != 1 && dx != 2 && dx != 4) || (dy != 1 && dy != 2 && dy != 4)) {
            cs_log(ps, LOGL_ERROR, "component %d: sub-sampling %dx%d is not supported\n", i, dx, dy);
            return HTJ2K_ERR_INVALIDDATA;
        }
        sub[i][0] = (uint8_t)(dx >> 1);
        sub[i][1] = (uint8_t)(dy >> 1);
    }

    ps->tiles_x = (uint32_t)cdiv(ps->xsiz - ps->xtosiz, ps->xtsiz);
    ps->tiles_y = (uint32_t)cdiv(ps->ysiz - ps->ytosiz, ps->ytsiz);
    /* every tile costs at least SOT + SOD = 14 bytes of input */
    if ((uint64_t)ps->tiles_x * ps->tiles_y > INT_MAX / sizeof(TileHdr) ||
        (int64_t)ps->tiles_x * ps->tiles_y * 14 > (int64_t)(c->end - c->base)) {
        ps->tiles_x = ps->tiles_y = 0;
        return HTJ2K_ERR_EINVAL;
    }
    ntiles = ps->tiles_x * ps->tiles_y;
    ps->tile = (TileHdr *)pool_get(&ps->frame, (size_t)ntiles * sizeof(TileHdr), 1);
    if (!ps->tile) {
        ps->tiles_x = ps->tiles_y = 0;
        return HTJ2K_ERR_ENOMEM;
    }
    ps->have_siz = 1;

    /* picture size = the largest component after `lowres` (jpeg2000dec.c:308-326) */
    rw = cdiv_pow2(ps->xsiz - ps->xosiz, ps->reduce);
    rh = cdiv_pow2(ps->ysiz - ps->yosiz, ps->reduce);
    ps->out_w = ps->out_h = 0;
    for (i = 0; i < n; i++) {
        ps->out_w = max32(ps->out_w, cdiv(rw, ps->sub_x[i]));
        ps->out_h = max32(ps->out_h, cdiv(rh, ps->sub_y[i]));
    }
    if (!cs_picture_size_ok((uint32_t)ps->out_w, (uint32_t)ps->out_h, pixel_budget(ps)))
        return HTJ2K_ERR_EINVAL;

    ps->pix_fmt = pick_layout(ps, sub);
    if (ps->pix_fmt == HTJ2K_PIX_NONE)
        ps->pix_fmt = pick_layout_by_shape(ps);
    if (ps->pix_fmt == HTJ2K_PIX_NONE) {
        cs_log(ps, LOGL_ERROR, "no output layout for Rsiz %d, colour space %d, %d components of %d bits\n",
               ps->rsiz, ps->colour_space, n, ps->precision);
        return HTJ2K_ERR_PATCHWELCOME;
    }
    return 0;
}

/* CAP: only the Part-15 word matters (jpeg2000dec.c:424-489, T.814 A.3) */
static int seg_cap(J2kParser *ps, Cur *c, int lseg)
{
    uint32_t pcap, ccap15 = 0;
    int part;
    (void)lseg;
    if (cur_left(c) < 6) {
        cs_log(ps, LOGL_ERROR, "CAP segment cut short\n");
        return HTJ2K_ERR_INVALIDDATA;
    }
    pcap = ld_be32(c->p);
    c->p += 4;
    for (part = 1; part <= 32; part++)
        if (pcap & (0x80000000u >> (part - 1))) {
            const uint32_t w = cur_u16(c);
            if (part == 15)
                ccap15 = w;
        }
    ps->is_ht = (pcap >> 17) & 1;
    if (!ps->is_ht)
        return 0;
    cs_log(ps, LOGL_INFO, "HT block coder announced (Ccap15 %04x)\n", (unsigned)ccap15);
    ps->ht_kind = (uint8_t)(ccap15 >> 14);                    /* 0 HTONLY, 1 HTDECLARED, 3 MIXED */
    if (ps->ht_kind == 2) {
        cs_log(ps, LOGL_ERROR, "Ccap15 bits 14-15 hold a reserved value\n");
        return HTJ2K_ERR_EINVAL;
    }
    if (ccap15 & 0x2000) {
        cs_log(ps, LOGL_ERROR, "multiple HT sets per block are not supported\n");
        return HTJ2K_ERR_PATCHWELCOME;
    }
    ps->ht_rgn_ok = (ccap15 >> 12) & 1;
    ps->ht_hetero = (ccap15 >> 11) & 1;
    ps->ht_irrev  = (ccap15 >> 5) & 1;
    {
        const uint32_t p = ccap15 & 31;
        ps->ht_magbits = (uint8_t)(p == 0 ? 8 : p < 20 ? p + 8 : p < 31 ? 4 * (p - 19) + 27 : 74);
    }
    if (ps->ht_magbits > 31) {
        cs_log(ps, LOGL_ERROR, "magnitude bound B = %d exceeds 31 bits\n", ps->ht_magbits);
        return HTJ2K_ERR_PATCHWELCOME;
    }
    return 0;
}

/* SPcod / SPcoc (jpeg2000dec.c:492-568) into one component's parameter set */
static int read_coding_params(J2kParser *ps, Cur *c, CompCoding *k)
{
    int r;
    if (cur_left(c) < 5) {
        cs_log(ps, LOGL_ERROR, "coding style parameters cut short\n");
        return HTJ2K_ERR_INVALIDDATA;
    }
    k->nres = (uint8_t)(c->p[0] + 1);
    if (c->p[0] + 1 >= CS_MAX_RES) {
        cs_log(ps, LOGL_ERROR, "%d decomposition levels\n", c->p[0]);
        c->p += 1;
        return HTJ2K_ERR_INVALIDDATA;
    }
    if (k->nres <= ps->reduce) {
        cs_log(ps, LOGL_ERROR, "lowres %d asks for more than the %d levels of this stream\n", ps->reduce, k->nres - 1);
        c->p += 1;
        return HTJ2K_ERR_EINVAL;
    }
    k->nres_dec = (uint8_t)(k->nres - ps->reduce);
    k->cbw = (uint8_t)((c->p[1] & 15) + 2);
    k->cbh = (uint8_t)((c->p[2] & 15) + 2);
    if (k->cbw > 10 || k->cbh > 10 || k->cbw + k->cbh > 12) {
        cs_log(ps, LOGL_ERROR, "code-block size 2^%d x 2^%d\n", k->cbw, k->cbh);
        c->p += 3;
        return HTJ2K_ERR_INVALIDDATA;
    }
    k->cb_style = c->p[3];
    if (k->cb_style && !(k->cb_style & (CBS_HT | CBS_HT_MIXED)))
        cs_log(ps, LOGL_WARNING, "Part-1 mode switches %02x\n", k->cb_style);
    k->wavelet = c->p[4];
    if (k->wavelet == J2K_DWT97 && ps->opts.bitexact)
        k->wavelet = J2K_DWT97_INT;
    else if (k->wavelet == J2K_DWT53)
        ps->lossless = 1;
    c->p += 5;
    if (!(k->scod & SCOD_PRECINCTS)) {
        memset(k->ppx, 15, sizeof k->ppx);
        memset(k->ppy, 15, sizeof k->ppy);
        return 0;
    }
    for (r = 0; r < k->nres; r++) {
        const uint32_t v = cur_u8(c);
        k->ppx[r] = v & 15;
        k->ppy[r] = (uint8_t)(v >> 4);
        if (r && (!k->ppx[r] || !k->ppy[r])) {
            cs_log(ps, LOGL_ERROR, "precinct size 2^%d x 2^%d at resolution %d\n", k->ppx[r], k->ppy[r], r);
            return HTJ2K_ERR_INVALIDDATA;
        }
    }
    return 0;
}

static int seg_cod(J2kParser *ps, Cur *c, int lseg)
{
    CompCoding fresh, *dst = coding_target(ps);
    const uint8_t *seen = seen_target(ps);
    int i, r;
    (void)lseg;
    if (cur_left(c) < 5) {
        cs_log(ps, LOGL_ERROR, "COD segment cut short\n");
        return HTJ2K_ERR_INVALIDDATA;
    }
    memset(&fresh, 0, sizeof fresh);
    fresh.scod   = c->p[0];
    fresh.order  = c->p[1];
    fresh.layers = c->p[3];                  /* low byte of the layer count, see CompCoding */
    fresh.mct    = c->p[4];
    c->p += 5;
    if (fresh.mct && ps->ncomp < 3) {
        cs_log(ps, LOGL_ERROR, "component transform signalled for %d components\n", ps->ncomp);
        return HTJ2K_ERR_INVALIDDATA;
    }
    if ((r = read_coding_params(ps, c, &fresh)) < 0)
        return r;
    fresh.defined = 1;
    for (i = 0; i < ps->ncomp; i++)
        if (!(seen[i] & SEEN_COC))
            dst[i] = fresh;
    return 0;
}

static int seg_coc(J2kParser *ps, Cur *c, int lseg)
{
    CompCoding *k;
    int comp, r;
    (void)lseg;
    if (cur_left(c) < 2) {
        cs_log(ps, LOGL_ERROR, "COC segment cut short\n");
        return HTJ2K_ERR_INVALIDDATA;
    }
    comp = c->p[0];
    if (comp >= ps->ncomp) {
        c->p += 1;
        cs_log(ps, LOGL_ERROR, "COC for component %d of %d\n", comp, ps->ncomp);
        return HTJ2K_ERR_INVALIDDATA;
    }
    k = coding_target(ps) + comp;
    k->scod = (uint8_t)(c->p[1] | (k->scod & (SCOD_SOP | SCOD_EPH)));   /* SOP / EPH come from the COD only */
    c->p += 2;
    if ((r = read_coding_params(ps, c, k)) < 0)
        return r;
    seen_target(ps)[comp] |= SEEN_COC;
    k->defined = 1;
    return 0;
}

/* RGN: the up-shift of the region of interest (jpeg2000dec.c:643-673) */
static int seg_rgn(J2kParser *ps, Cur *c, int lseg)
{
    uint32_t comp, shift;
    int r = 0;
    (void)lseg;
    comp = cur_u8(c);
    if (cur_u8(c) != 0) {
        cs_log(ps, LOGL_ERROR, "RGN style is not 'implicit'\n");
        r = HTJ2K_ERR_INVALIDDATA;
    } else if ((int)comp >= ps->ncomp) {
        r = HTJ2K_ERR_INVALIDDATA;
    } else if (ps->cur_tile >= 0 && ps->tile[ps->cur_tile].cur_part != 0) {
        r = HTJ2K_ERR_INVALIDDATA;
    } else if ((shift = cur_u8(c)) > 30) {
        r = HTJ2K_ERR_PATCHWELCOME;
    } else if (ps->cur_tile < 0) {
        ps->roi[comp] = (uint8_t)shift;
    } else {
        ps->tile[ps->cur_tile].roi[comp] = (uint8_t)shift;
        ps->tile[ps->cur_tile].own_params |= TILE_OWN_PARAMS;
    }
    if (ps->is_ht && !ps->ht_rgn_ok) {
        cs_log(ps, LOGL_ERROR, "RGN segment in a codestream of the RGNFREE set\n");
        return HTJ2K_ERR_INVALIDDATA;
    }
    return r;
}

/* Sqcd/SPqcd resp. Sqcc/SPqcc (jpeg2000dec.c:676-718); `payload` = bytes of the segment from Sqcx on */
static int read_quant_params(Cur *c, int payload, CompQuant *q)
{
    int i, n;
    if (cur_left(c) < 1)
        return HTJ2K_ERR_INVALIDDATA;
    q->guard = c->p[0] >> 5;
    q->style = c->p[0] & 31;
    c->p += 1;
    if (q->style == 0) {                                   /* reversible: exponents only */
        n = payload - 1;
        if (cur_left(c) < n || n > CS_MAX_BANDS)
            return HTJ2K_ERR_INVALIDDATA;
        for (i = 0; i < n; i++)
            q->expn[i] = *c->p++ >> 3;
    } else if (q->style == 1) {                            /* derived from the LL band's pair */
        uint32_t v;
        if (cur_left(c) < 2)
            return HTJ2K_ERR_INVALIDDATA;
        v = ld_be16(c->p);
        c->p += 2;
        for (i = 0; i < CS_MAX_BANDS; i++) {
            const int e = (int)(v >> 11) - (i ? (i - 1) / 3 : 0);
            q->expn[i] = (uint8_t)(e > 0 ? e : 0);
            q->mant[i] = v & 0x7FF;
        }
    } else {                                               /* one (exponent, mantissa) pair per band */
        n = (payload - 1) >> 1;
        if (cur_left(c) < 2 * n || n > CS_MAX_BANDS)
            return HTJ2K_ERR_INVALIDDATA;
        for (i = 0; i < n; i++, c->p += 2) {
            q->expn[i] = c->p[0] >> 3;
            q->mant[i] = ld_be16(c->p) & 0x7FF;
        }
    }
    return 0;
}

static int seg_qcd(J2kParser *ps, Cur *c, int lseg)
{
    CompQuant fresh, *dst = quant_target(ps);
    const uint8_t *seen = seen_target(ps);
    int i, r;
    memset(&fresh, 0, sizeof fresh);
    if ((r = read_quant_params(c, lseg - 2, &fresh)) < 0)
        return r;
    for (i = 0; i < ps->ncomp; i++)
        if (!(seen[i] & SEEN_QCC))
            dst[i] = fresh;
    return 0;
}

static int seg_qcc(J2kParser *ps, Cur *c, int lseg)
{
    int comp;
    if (cur_left(c) < 1)
        return HTJ2K_ERR_INVALIDDATA;
    comp = *c->p++;
    if (comp >= ps->ncomp) {
        cs_log(ps, LOGL_ERROR, "QCC for component %d of %d\n", comp, ps->ncomp);
        return HTJ2K_ERR_INVALIDDATA;
    }
    seen_target(ps)[comp] |= SEEN_QCC;
    return read_quant_params(c, lseg - 3, quant_target(ps) + comp);
}

/* POC (jpeg2000dec.c:760-818): a tile-part POC replaces progressions inherited from the
 * main header and extends the tile's own */
static int seg_poc(J2kParser *ps, Cur *c, int lseg)
{
    PocList add, *dst = ps->cur_tile < 0 ? &ps->poc : &ps->tile[ps->cur_tile].poc;
    const int entry = 7;                                   /* component indices are bytes below 257 components */
    int i;
    memset(&add, 0, sizeof add);
    if (cur_left(c) < 5 || lseg < 2 + entry) {
        cs_log(ps, LOGL_ERROR, "POC segment cut short\n");
        return HTJ2K_ERR_INVALIDDATA;
    }
    add.n = (lseg - 2) / entry;
    if (add.n > CS_MAX_POC) {
        cs_log(ps, LOGL_ERROR, "%d progression changes in one POC\n", add.n);
        return HTJ2K_ERR_PATCHWELCOME;
    }
    for (i = 0; i < add.n; i++) {
        PocVolume *v = &add.v[i];
        v->rs    = (uint8_t)cur_u8(c);
        v->cs    = (uint16_t)cur_u8(c);
        v->lye   = (uint16_t)cur_u16(c);
        v->re    = (uint8_t)cur_u8(c);
        v->ce    = (uint16_t)cur_u8(c);
        v->order = (uint8_t)cur_u8(c);
        if (!v->ce)
            v->ce = 256;
        if (v->ce > ps->ncomp)
            v->ce = (uint16_t)ps->ncomp;
        if (v->rs >= v->re || v->re > 33 || v->cs >= v->ce || !v->lye) {
            cs_log(ps, LOGL_ERROR, "POC volume %d is empty or out of range (R %d..%d, C %d..%d, L ..%d)\n",
                   i, v->rs, v->re, v->cs, v->ce, v->lye);
            return HTJ2K_ERR_INVALIDDATA;
        }
    }
    if (!dst->n || dst->inherited) {
        *dst = add;
    } else {
        if (dst->n + add.n > CS_MAX_POC) {
            cs_log(ps, LOGL_ERROR, "more than %d progression changes in a tile\n", CS_MAX_POC);
            return HTJ2K_ERR_INVALIDDATA;
        }
        memcpy(dst->v + dst->n, add.v, (size_t)add.n * sizeof add.v[0]);
        dst->n += add.n;
    }
    dst->inherited = 0;
    return 0;
}

/* SOT (jpeg2000dec.c:822-873): where the tile-part ends; the first tile-part of a tile starts from
 * the main header's parameters */
static int seg_sot(J2kParser *ps, Cur *c, int lseg)
{
    uint32_t isot, psot, tpsot, rest;
    TileHdr *t;
    if (!ps->in_tile_hdr) {
        ps->in_tile_hdr = 1;
        if (ps->has_ppm)
            ps->ppm_cur = cur_make(ps->ppm, (size_t)ps->ppm_size);
    }
    if (cur_left(c) < 8)
        return HTJ2K_ERR_INVALIDDATA;
    isot = ld_be16(c->p);
    ps->cur_tile = 0;
    if (isot >= ps->tiles_x * ps->tiles_y) {
        c->p += 2;
        return HTJ2K_ERR_INVALIDDATA;
    }
    ps->cur_tile = (int)isot;
    psot  = ld_be32(c->p + 2);
    tpsot = c->p[6];                                       /* c->p[7], TNsot, is not needed */
    c->p += 8;
    rest = (uint32_t)(cur_left(c) + lseg);                 /* from the SOT marker's length field + 2 to the end */
    if (!psot)
        psot = rest;                                       /* "until the end of the codestream" */
    if (psot > rest) {
        cs_log(ps, LOGL_ERROR, "tile-part length %u runs past the end of the data\n", (unsigned)psot);
        return HTJ2K_ERR_INVALIDDATA;
    }
    if (tpsot >= CS_MAX_TPARTS) {
        cs_log(ps, LOGL_ERROR, "tile-part index %u: at most %d per tile\n", (unsigned)tpsot, CS_MAX_TPARTS);
        return HTJ2K_ERR_PATCHWELCOME;
    }
    t = &ps->tile[isot];
    t->cur_part = (uint16_t)tpsot;
    t->part[tpsot].limit = c->p + psot - lseg - 2;
    if (!tpsot) {
        memcpy(t->cod, ps->cod, (size_t)ps->ncomp * sizeof t->cod[0]);
        memcpy(t->q, ps->q, (size_t)ps->ncomp * sizeof t->q[0]);
        t->poc = ps->poc;
        t->poc.inherited = 1;
        t->own_params |= TILE_HAS_DEFAULTS;
    }
    return 0;
}

static int seg_crg(J2kParser *ps, Cur *c, int lseg)
{
    if (ps->ncomp * 4 != lseg - 2) {
        cs_log(ps, LOGL_ERROR, "CRG segment does not hold one offset pair per component\n");
        return HTJ2K_ERR_INVALIDDATA;
    }
    cur_skip(c, (uint32_t)(lseg - 2));
    return 0;
}

static int seg_cpf(J2kParser *ps, Cur *c, int lseg)
{
    (void)ps;
    if (cur_left(c) < lseg - 2)
        return HTJ2K_ERR_INVALIDDATA;
    cur_skip(c, (uint32_t)(lseg - 2));
    return 0;
}

static int seg_skip(J2kParser *ps, Cur *c, int lseg)
{
    (void)ps;
    cur_skip(c, (uint32_t)(lseg - 2));
    return 0;
}

/* TLM (jpeg2000dec.c:901-936 reads and forgets it): Ztlm, Stlm, then (Ttlm, Ptlm) records */
static int seg_tlm(J2kParser *ps, Cur *c, int lseg)
{
    uint32_t stlm, st, sp, records, i;
    cur_u8(c);
    stlm = cur_u8(c);
    st = (stlm >> 4) & 3;
    sp = (stlm >> 6) & 1;
    if (st == 3) {
        cs_log(ps, LOGL_ERROR, "TLM with the reserved index size\n");
        return HTJ2K_ERR_INVALIDDATA;
    }
    records = (uint32_t)((lseg - 4) / (int)((sp + 1) * 2 + st)) & 0xFF;    /* the reference counts them in a byte */
    for (i = 0; i < records; i++) {
        if (st == 1) cur_u8(c);
        else if (st == 2) cur_u16(c);
        if (sp) cur_u32(c);
        else cur_u16(c);
    }
    return 0;
}

/* PLT (jpeg2000dec.c:938-956): packet lengths as 7-bit groups; the last group of the segment must
 * end a length */
static int seg_plt(J2kParser *ps, Cur *c, int lseg)
{
    TileHdr *t = ps->cur_tile >= 0 ? &ps->tile[ps->cur_tile] : NULL;
    uint32_t last = 0;
    int i;
    if (lseg < 4)
        return HTJ2K_ERR_INVALIDDATA;
    cur_u8(c);                                             /* Zplt: the segments are taken in the order they come */
    if (t && t->plt_open)
        t->plt_bad = 1;                                    /* a length does not continue across segments */
    for (i = 0; i < lseg - 3; i++) {
        last = cur_u8(c);
        if (!t || t->plt_bad)
            continue;
        if (t->plt_acc >> 25) {                            /* more than 32 bits */
            t->plt_bad = 1;
            continue;
        }
        t->plt_acc = (t->plt_acc << 7) | (last & 0x7F);
        t->plt_open = (uint8_t)(last >> 7);
        if (!t->plt_open) {
            if (t->nplt == t->plt_cap) {
                const uint32_t nc = t->plt_cap ? 2 * t->plt_cap : 1024;
                uint32_t *nb = (uint32_t *)pool_get(&ps->frame, (size_t)nc * sizeof *nb, 0);
                if (!nb) {
                    t->plt_bad = 1;
                    continue;
                }
                if (t->nplt)
                    memcpy(nb, t->plt, (size_t)t->nplt * sizeof *nb);
                t->plt = nb;
                t->plt_cap = nc;
            }
            t->plt[t->nplt++] = t->plt_acc;
            t->plt_acc = 0;
        }
    }
    return (last & 0x80) ? HTJ2K_ERR_INVALIDDATA : 0;
}

/* PPM / PPT: packet headers moved out of the tile-parts (jpeg2000dec.c:958-1014) */
static int grow_packed(J2kParser *ps, Cur *c, uint8_t **buf, int *size, int n)
{
    uint8_t *nb = (uint8_t *)pool_get(&ps->frame, (size_t)*size + (size_t)n + 8, 1);
    const int have = min32(n, cur_left(c));
    if (!nb)
        return HTJ2K_ERR_ENOMEM;
    if (*size)
        memcpy(nb, *buf, (size_t)*size);
    memcpy(nb + *size, c->p, (size_t)have);
    c->p += have;
    *buf = nb;
    *size += n;
    return 0;
}

static int seg_ppm(J2kParser *ps, Cur *c, int lseg)
{
    if (ps->in_tile_hdr) {
        cs_log(ps, LOGL_ERROR, "PPM segment outside the main header\n");
        return HTJ2K_ERR_INVALIDDATA;
    }
    if (lseg < 3) {
        cs_log(ps, LOGL_ERROR, "PPM segment without an index\n");
        return HTJ2K_ERR_INVALIDDATA;
    }
    cur_u8(c);
    ps->has_ppm = 1;
    memset(&ps->ppm_cur, 0, sizeof ps->ppm_cur);
    return grow_packed(ps, c, &ps->ppm, &ps->ppm_size, lseg - 3);
}

static int seg_ppt(J2kParser *ps, Cur *c, int lseg)
{
    TileHdr *t;
    if (ps->has_ppm) {
        cs_log(ps, LOGL_ERROR, "PPT segment in a codestream that has PPM segments\n");
        return HTJ2K_ERR_INVALIDDATA;
    }
    if (ps->is_ht && !ps->ht_hetero) {
        cs_log(ps, LOGL_ERROR, "PPT segment in a codestream of the HOMOGENEOUS set\n");
        return HTJ2K_ERR_INVALIDDATA;
    }
    if (lseg < 3) {
        cs_log(ps, LOGL_ERROR, "PPT segment without an index\n");
        return HTJ2K_ERR_INVALIDDATA;
    }
    if (ps->cur_tile < 0)
        return HTJ2K_ERR_INVALIDDATA;
    t = &ps->tile[ps->cur_tile];
    if (t->cur_part != 0) {
        cs_log(ps, LOGL_ERROR, "PPT segment behind the first tile-part of its tile\n");
        return HTJ2K_ERR_INVALIDDATA;
    }
    t->has_ppt = 1;
    cur_u8(c);
    memset(&t->ppt_cur, 0, sizeof t->ppt_cur);
    return grow_packed(ps, c, &t->ppt, &t->ppt_size, lseg - 3);
}

/* what may not appear in a tile-part header of a HOMOGENEOUS HT codestream (jpeg2000dec.c:2494-2560) */
#define RULE_FIXED_IN_HOMOGENEOUS 1
#define RULE_NEEDS_SIZ            2

static const struct SegRule { uint16_t code; uint8_t flags; const char *name; SegHandler fn; } seg_rules[] = {
    { MK_SIZ, 0,                         "SIZ", seg_siz },
    { MK_CAP, RULE_NEEDS_SIZ,            "CAP", seg_cap },
    { MK_COD, RULE_FIXED_IN_HOMOGENEOUS, "COD", seg_cod },
    { MK_COC, RULE_FIXED_IN_HOMOGENEOUS, "COC", seg_coc },
    { MK_QCD, RULE_FIXED_IN_HOMOGENEOUS, "QCD", seg_qcd },
    { MK_QCC, RULE_FIXED_IN_HOMOGENEOUS, "QCC", seg_qcc },
    { MK_RGN, RULE_FIXED_IN_HOMOGENEOUS, "RGN", seg_rgn },
    { MK_POC, RULE_FIXED_IN_HOMOGENEOUS, "POC", seg_poc },
    { MK_SOT, 0,                         "SOT", seg_sot },
    { MK_TLM, 0,                         "TLM", seg_tlm },
    { MK_PLT, 0,                         "PLT", seg_plt },
    { MK_PPM, 0,                         "PPM", seg_ppm },
    { MK_PPT, 0,                         "PPT", seg_ppt },
    { MK_CRG, 0,                         "CRG", seg_crg },
    { MK_CPF, 0,                         "CPF", seg_cpf },
    { MK_PLM, 0,                         "PLM", seg_skip },
    { MK_COM, 0,                         "COM", seg_skip },
};

/* SOD: the rest of the tile-part is packet data; remember where, read none of it
 * (jpeg2000dec.c:2455-2490) */
static int enter_tile_part_body(J2kParser *ps)
{
    Cur *c = &ps->cs;
    TileHdr *t;
    TilePartSpan *tp;
    if (!ps->tile) {
        cs_log(ps, LOGL_ERROR, "SOD without SIZ\n");
        return HTJ2K_ERR_INVALIDDATA;
    }
    if (ps->cur_tile < 0) {
        cs_log(ps, LOGL_ERROR, "SOD without SOT\n");
        return HTJ2K_ERR_INVALIDDATA;
    }
    t = &ps->tile[ps->cur_tile];
    tp = &t->part[t->cur_part];
    if (tp->limit < c->p) {
        cs_log(ps, LOGL_ERROR, "tile-part ends in front of its own SOD\n");
        return HTJ2K_ERR_INVALIDDATA;
    }
    if (ps->has_ppm) {                       /* Nppm, then this tile-part's packet headers */
        const uint32_t n = cur_u32(&ps->ppm_cur);
        if ((uint32_t)cur_left(&ps->ppm_cur) < n)
            return HTJ2K_ERR_INVALIDDATA;
        tp->hdr = cur_make(ps->ppm_cur.p, n);
        cur_skip(&ps->ppm_cur, n);
    }
    if (t->has_ppt && t->cur_part == 0)
        t->ppt_cur = cur_make(t->ppt, (size_t)t->ppt_size);
    tp->body = cur_make(c->p, (size_t)(tp->limit - c->p));
    c->p = tp->limit;
    return 0;
}

int cs_scan_headers(J2kParser *ps)
{
    Cur *c = &ps->cs;
    for (;;) {
        const struct SegRule *rule = NULL;
        uint32_t code;
        int lseg, at, r = 0;
        size_t k;

        if (cur_left(c) < 2) {
            cs_log(ps, LOGL_ERROR, "codestream ends without EOC\n");
            return 0;
        }
        code = ld_be16(c->p);
        c->p += 2;
        at = cur_pos(c);
        if (code >= 0xFF30 && code <= 0xFF3F)              /* markers without a segment */
            continue;
        if (code == MK_SOD) {
            if ((r = enter_tile_part_body(ps)) < 0)
                return r;
            continue;
        }
        if (code == MK_EOC)
            return 0;

        lseg = (int)cur_u16(c);
        if (lseg < 2 || cur_left(c) < lseg - 2) {
            if (ps->opts.strict) {
                cs_log(ps, LOGL_ERROR, "marker %04x: segment length %d, %d bytes left\n", (unsigned)code, lseg, cur_left(c));
                return HTJ2K_ERR_INVALIDDATA;
            }
            cs_log(ps, LOGL_WARNING, "codestream ends inside marker segment %04x\n", (unsigned)code);
            return 0;
        }
        for (k = 0; k < sizeof seg_rules / sizeof seg_rules[0]; k++)
            if (seg_rules[k].code == code)
                rule = &seg_rules[k];
        if (!rule) {
            cs_log(ps, LOGL_ERROR, "marker %04x at %#x is not supported: skipped\n", (unsigned)code, at - 2);
            cur_skip(c, (uint32_t)(lseg - 2));
        } else {
            if (code == MK_SIZ && ps->ncomp) {
                cs_log(ps, LOGL_ERROR, "second SIZ segment\n");
                return HTJ2K_ERR_INVALIDDATA;
            }
            if ((rule->flags & RULE_NEEDS_SIZ) && !ps->ncomp) {
                cs_log(ps, LOGL_ERROR, "%s segment in front of SIZ\n", rule->name);
                return HTJ2K_ERR_INVALIDDATA;
            }
            if ((rule->flags & RULE_FIXED_IN_HOMOGENEOUS) && ps->in_tile_hdr && ps->is_ht && !ps->ht_hetero) {
                cs_log(ps, LOGL_ERROR, "%s segment in a tile-part header of a HOMOGENEOUS codestream\n", rule->name);
                return HTJ2K_ERR_INVALIDDATA;
            }
            r = rule->fn(ps, c, lseg);
            if (code == MK_SIZ && !ps->tile)
                ps->tiles_x = ps->tiles_y = 0;
        }
        if (r || cur_pos(c) - at != lseg) {
            cs_log(ps, LOGL_ERROR, "marker segment %04x: %s\n", (unsigned)code, r ? "rejected" : "length does not match its contents");
            return r ? r : -1;
        }
    }
}

/* ------------------------------------------------------------------ JP2 file wrapper
 * Boxes in front of the codestream (jpeg2000dec.c:2658-2805): colour space, palette, channel
 * definitions and capture / display resolution are picked up from the `jp2h` super-box. */
#define BOX(a, b, c, d) (((uint32_t)(a) << 24) | ((uint32_t)(b) << 16) | ((uint32_t)(c) << 8) | (uint32_t)(d))

/* pixel aspect ratio = (hnum * vden * 10^hexp) : (vnum * hden * 10^vexp) brought into 31 bits
 * (av_reduce, libavutil/rational.c:35-81, for arguments that real files carry) */
static void store_aspect(J2kParser *ps, double n, double d)
{
    int64_t a = (int64_t)n, b = (int64_t)d, g, r, p0 = 0, p1 = 1, q0 = 1, q1 = 0;
    if (a <= 0 || b <= 0)
        return;
    for (g = a, r = b; r; ) { const int64_t t = g % r; g = r; r = t; }
    a /= g;
    b /= g;
    if (a > INT32_MAX || b > INT32_MAX) {                  /* best convergent of the continued fraction that fits */
        int64_t x = a, y = b;
        while (y) {
            const int64_t quo = x / y, rem = x - quo * y, p2 = quo * p1 + p0, q2 = quo * q1 + q0;
            if (p2 > INT32_MAX || q2 > INT32_MAX)
                break;
            p0 = p1; q0 = q1; p1 = p2; q1 = q2;
            x = y; y = rem;
        }
        a = p1;
        b = q1;
    }
    ps->sar_den = (int)a;
    ps->sar_num = (int)b;
}

static void box_colr(J2kParser *ps, Cur *c, uint32_t payload)
{
    if (payload >= 7 && c->p[0] == 1)                      /* enumerated colour space */
        ps->colour_space = (int)ld_be32(c->p + 3);
}

static void box_pclr(J2kParser *ps, Cur *c, uint32_t payload)
{
    int entries, i, k, bits[3], need = 0;
    if (payload < 6)
        return;
    entries = (int)ld_be16(c->p);
    for (k = 0; k < 3; k++) {
        bits[k] = (c->p[3 + k] & 0x7F) + 1;
        need += ((bits[k] + 7) >> 3) * entries;
    }
    if (entries > 256 || c->p[2] != 3 || bits[0] > 16 || bits[1] > 16 || bits[2] > 16 || payload < (uint32_t)need) {
        cs_log(ps, LOGL_ERROR, "palette box of an unsupported shape: ignored\n");
        return;
    }
    c->p += 6;
    ps->palettised = 1;
    for (i = 0; i < entries; i++) {
        uint32_t rgb = 0;
        for (k = 0; k < 3; k++) {
            uint32_t v;
            if (bits[k] <= 8) {
                v = cur_u8(c) << (8 - bits[k]);
                v |= v >> bits[k];
            } else {
                v = cur_u16(c) >> (bits[k] - 8);
            }
            rgb = (rgb << 8) | v;
        }
        ps->palette[i] = 0xFF000000u | rgb;
    }
}

static void box_cdef(J2kParser *ps, Cur *c, uint32_t payload)
{
    int n;
    if (payload < 2)
        return;
    for (n = (int)cur_u16(c); n > 0; n--) {
        const uint32_t chan = cur_u16(c);
        uint32_t assoc;
        cur_u16(c);                                         /* channel type */
        assoc = cur_u16(c);
        if (chan < J2K_MAX_COMPS && assoc < J2K_MAX_COMPS)
            ps->cdef[chan] = (int)assoc;
    }
}

static void box_res(J2kParser *ps, Cur *c, uint32_t payload)
{
    int64_t vn, vd, hn, hd, ve, he;
    uint32_t kind;
    if (payload < 18)
        return;
    kind = ld_be32(c->p + 4);
    if (kind != BOX('r', 'e', 's', 'c') && kind != BOX('r', 'e', 's', 'd'))
        return;
    vn = ld_be16(c->p + 8);  vd = ld_be16(c->p + 10);
    hn = ld_be16(c->p + 12); hd = ld_be16(c->p + 14);
    ve = c->p[16]; he = c->p[17];
    if (!vn || !vd || !hn || !hd) {
        cs_log(ps, LOGL_WARNING, "resolution box with a zero term: ignored\n");
        return;
    }
    if (ve > he) { ve -= he; he = 0; }
    else         { he -= ve; ve = 0; }
    if ((double)INT64_MAX / (double)(hn * vd) > pow(10, (double)he) &&
        (double)INT64_MAX / (double)(vn * hd) > pow(10, (double)ve))
        store_aspect(ps, (double)(hn * vd) * pow(10, (double)he), (double)(vn * hd) * pow(10, (double)ve));
}

static const struct { uint32_t type; void (*fn)(J2kParser *, Cur *, uint32_t); } header_boxes[] = {
    { BOX('c', 'o', 'l', 'r'), box_colr }, { BOX('p', 'c', 'l', 'r'), box_pclr },
    { BOX('c', 'd', 'e', 'f'), box_cdef }, { BOX('r', 'e', 's', ' '), box_res },
};

/* returns 1 with the cursor behind the header of the contiguous-codestream box, 0 when there is
 * none within reach, < 0 for a box length that cannot be (which the caller treats like 1, as the
 * reference does: jpeg2000dec.c:2846) */
static int find_codestream_box(J2kParser *ps)
{
    Cur *c = &ps->cs;
    int others = 10;                                       /* boxes other than jp2h the search walks past */
    while (others && cur_left(c) >= 8) {
        uint32_t size = ld_be32(c->p), type = ld_be32(c->p + 4), end;
        c->p += 8;
        if (size == 1) {                                   /* XLBox */
            if (cur_u32(c)) {
                cs_log(ps, LOGL_ERROR, "box larger than 4 GB\n");
                return 0;
            }
            size = cur_u32(c);
            if (size < 16 || (int64_t)cur_pos(c) + size - 16 > INT_MAX)
                return HTJ2K_ERR_INVALIDDATA;
            end = (uint32_t)cur_pos(c) + size - 16;
        } else {
            if (size < 8 || (int64_t)cur_pos(c) + size - 8 > INT_MAX)
                return HTJ2K_ERR_INVALIDDATA;
            end = (uint32_t)cur_pos(c) + size - 8;
        }
        if (type == BOX('j', 'p', '2', 'c'))
            return 1;
        if ((uint32_t)cur_left(c) < size || end < size)
            return 0;
        if (type == BOX('j', 'p', '2', 'h') && size >= 16) {
            uint32_t sub_end;
            do {
                uint32_t sub_size, sub_type;
                size_t k;
                if (cur_left(c) < 8)
                    break;
                sub_size = ld_be32(c->p);
                sub_type = ld_be32(c->p + 4);
                c->p += 8;
                sub_end = (uint32_t)cur_pos(c) + sub_size - 8;
                if (sub_size < 8 || sub_end > end || sub_end < sub_size)
                    break;
                if (sub_type == BOX('j', 'p', '2', 'c'))
                    return 1;
                for (k = 0; k < sizeof header_boxes / sizeof header_boxes[0]; k++)
                    if (header_boxes[k].type == sub_type)
                        header_boxes[k].fn(ps, c, sub_size - 8);
                cur_goto(c, sub_end);
            } while (end - sub_end >= 8);
        } else {
            others--;
        }
        cur_goto(c, end);
    }
    return 0;
}

int cs_locate_codestream(J2kParser *ps)
{
    Cur *c = &ps->cs;
    if (cur_left(c) < 2)
        return HTJ2K_ERR_INVALIDDATA;
    if (cur_left(c) >= 12 && ld_be32(c->p) == 12 && ld_be32(c->p + 4) == BOX('j', 'P', ' ', ' ') &&
        ld_be32(c->p + 8) == 0x0D0A870Au) {
        c->p += 12;
        if (!find_codestream_box(ps)) {
            cs_log(ps, LOGL_ERROR, "JP2 file without a codestream box\n");
            return HTJ2K_ERR_INVALIDDATA;
        }
    }
    while (cur_left(c) >= 3 && ld_be16(c->p) != MK_SOC)    /* tolerate junk in front of SOC */
        c->p++;
    if (cur_left(c) < 2 || ld_be16(c->p) != MK_SOC) {
        cs_log(ps, LOGL_ERROR, "no SOC marker\n");
        return HTJ2K_ERR_INVALIDDATA;
    }
    c->p += 2;
    return 0;
}

/*
 * j2k_tier2.c -- geometry and Tier-2 of the host front-end.
 *
 * What the reference computes while it builds and walks its Tile -> Component -> ResLevel ->
 * Band -> Precinct -> Cblk tree (libavcodec/jpeg2000.c:214-577 geometry and step sizes,
 * jpeg2000dec.c:1016-1070 tile set-up, :1073-1869 packet headers and packet order) is
 * computed here in closed form into the flat tables of j2k_host.h:
 *
 *   t2_build_geometry()       every rectangle (resolution, sub-band, precinct, code-block) from
 *                             the tile-component rectangle by shifts; the block table of the
 *                             plan is laid out once and reused while the headers stay the same
 *   PacketWalk                the five progression orders as one table-driven loop nest
 *   BitWin                    packet-header bits through a 64-bit window: stuffing is removed
 *                             when bytes enter the window, a field is one shift
 *   tag trees                 two flat node arrays per precinct-band, parents by index arithmetic
 *
 * The numbers are the reference's (same rectangles, same order of blocks, same bits read for
 * every field, same errors); T.800 Annex B defines them, the reference's deviations from it
 * (zero-sized blocks of precincts that miss a band still take part in the packet headers; the
 * position tests of CPRL differ from those of RPCL / PCRL) are kept and marked.
 */
#include <math.h>
#include <stdlib.h>
#include "j2k_host.h"

/* ================================================================== geometry */

/* 2^(bits - exponent) * (1 + mantissa / 2^11), then the synthesis gain of the 9/7 filter bank folded in
 * (init_band_stepsize, jpeg2000.c:214-272).  The reference evaluates this in float with double
 * intermediates; the roundings are part of the result, so each product below is rounded where the
 * reference's assignment rounds. */
static void band_step(J2kParser *ps, BandGeom *b, const CompCoding *k, const CompQuant *q, int gband, int orient, int r, int bits)
{
    static const float LIFT_X = 0.812893066115961f, LIFT_K = 1.230174104914001f;
    float step;
    if (q->style == 0) {
        step = 1.0f;
    } else if (q->style <= 2) {
        union { uint32_t u; float f; } pw;
        pw.u = (uint32_t)((uint8_t)bits - q->expn[gband] + 127) << 23;
        step = pw.f;
        step = (float)(step * (q->mant[gband] / 2048.0 + 1.0));
    } else {
        step = 0.0f;
        cs_log(ps, LOGL_ERROR, "quantisation style %d is not defined\n", q->style);
    }
    if (k->wavelet != J2K_DWT53) {
        int one_lowpass = 0;
        if (orient == 1 || orient == 2) {           /* HL, LH: one high-pass filter */
            step *= LIFT_X * 2;
            one_lowpass = 1;
        } else if (orient == 3) {                   /* HH */
            step *= LIFT_X * LIFT_X * 4;
        }
        step = (float)(step * pow(LIFT_K, 2 * (k->nres_dec - r) + one_lowpass - 2));
    }
    if (step > (float)(INT_MAX >> 15)) {
        step = 0.0f;
        cs_log(ps, LOGL_ERROR, "step size does not fit 16.15 fixed point\n");
    }
    b->fstep = step;
    b->istep = (int32_t)floorf(step * 32768.0f);
}

static uint32_t tag_tree_nodes(int w, int h)
{
    uint64_t n = 1;
    while (w > 1 || h > 1) {
        n += (uint64_t)w * h;
        w = (w + 1) >> 1;
        h = (h + 1) >> 1;
    }
    return n >= INT32_MAX ? 0 : (uint32_t)n;
}

typedef struct GeomBuild {
    J2kParser *ps;
    GeomCache *g;
    /* growing tables */
    PrecBand *pb; uint32_t npb, pb_cap;
    J2kBlock *rows; uint32_t *row_blk, *row_tc; uint16_t *row_aux; uint32_t nrows, rows_cap;
    uint32_t nblk, nnodes, nprec;
} GeomBuild;

static int grow(void **tab, uint32_t *cap, uint32_t need, size_t elem)
{
    if (need > *cap) {
        uint32_t nc = *cap ? *cap : 256;
        void *nt;
        while (nc < need)
            nc = nc < (1u << 30) ? nc * 2 : need;
        nt = realloc(*tab, (size_t)nc * elem);
        if (!nt)
            return HTJ2K_ERR_ENOMEM;
        *tab = nt;
        *cap = nc;
    }
    return 0;
}

/* resolution r of a tile-component: rectangles, precinct grid, the precinct-bands with their block grids */
static int build_resolution(GeomBuild *gb, TcGeom *tc, const CompCoding *k, const CompQuant *q, int r, int bits, int *gband)
{
    J2kParser *ps = gb->ps;
    ResGeom *rg = &tc->res[r];
    const int down = k->nres - 1 - r;               /* halvings between the component and this resolution */
    int b, np;
    rg->x0 = cdiv_pow2(tc->ox0, down); rg->x1 = cdiv_pow2(tc->ox1, down);
    rg->y0 = cdiv_pow2(tc->oy0, down); rg->y1 = cdiv_pow2(tc->oy1, down);
    rg->ppx = k->ppx[r];
    rg->ppy = k->ppy[r];
    rg->nbands = r ? 3 : 1;
    rg->npx = rg->x1 == rg->x0 ? 0 : cdiv_pow2(rg->x1, rg->ppx) - (rg->x0 >> rg->ppx);
    rg->npy = rg->y1 == rg->y0 ? 0 : cdiv_pow2(rg->y1, rg->ppy) - (rg->y0 >> rg->ppy);
    /* the reference sizes an array of 56-byte precinct nodes here and gives up beyond max_pixels (jpeg2000.c:541-545) */
    if ((uint64_t)rg->npx * (uint64_t)rg->npy * rg->nbands > (uint64_t)pixel_budget(ps) / 56)
        return HTJ2K_ERR_ENOMEM;
    if ((uint64_t)rg->npx * (uint64_t)rg->npy > INT_MAX)
        return HTJ2K_ERR_ENOMEM;
    np = rg->npx * rg->npy;
    rg->pb0 = gb->npb;
    rg->lay0 = gb->nprec;
    gb->nprec += (uint32_t)np;
    if (grow((void **)&gb->pb, &gb->pb_cap, gb->npb + (uint32_t)np * rg->nbands, sizeof(PrecBand)) < 0)
        return HTJ2K_ERR_ENOMEM;

    for (b = 0; b < rg->nbands; b++, (*gband)++) {
        BandGeom *bg = &rg->band[b];
        const int orient = b + (r > 0);             /* 0 LL, 1 HL, 2 LH, 3 HH */
        int p;
        band_step(ps, bg, k, q, *gband, orient, r, bits);
        if (!r) {
            bg->x0 = rg->x0; bg->x1 = rg->x1; bg->y0 = rg->y0; bg->y1 = rg->y1;
            bg->bppx = rg->ppx;
            bg->bppy = rg->ppy;
        } else {
            /* the high-pass bands start half a sample period later (T.800 B-15) */
            const int64_t half = (int64_t)1 << down, hx = (orient & 1) ? half : 0, hy = (orient & 2) ? half : 0;
            bg->x0 = cdiv_pow2((int32_t)(tc->ox0 - hx), down + 1); bg->x1 = cdiv_pow2((int32_t)(tc->ox1 - hx), down + 1);
            bg->y0 = cdiv_pow2((int32_t)(tc->oy0 - hy), down + 1); bg->y1 = cdiv_pow2((int32_t)(tc->oy1 - hy), down + 1);
            bg->bppx = (uint8_t)(rg->ppx - 1);
            bg->bppy = (uint8_t)(rg->ppy - 1);
        }
        bg->cbw = (uint8_t)min32(k->cbw, bg->bppx);
        bg->cbh = (uint8_t)min32(k->cbh, bg->bppy);

        for (p = 0; p < np; p++) {
            PrecBand *pb = &gb->pb[rg->pb0 + (uint32_t)b * (uint32_t)np + (uint32_t)p];
            const int32_t gx = ((rg->x0 >> rg->ppx) + p % rg->npx) * (1 << bg->bppx);
            const int32_t gy = ((rg->y0 >> rg->ppy) + p / rg->npx) * (1 << bg->bppy);
            const int32_t px0 = max32(gx, bg->x0), px1 = min32(gx + (1 << bg->bppx), bg->x1);
            const int32_t py0 = max32(gy, bg->y0), py1 = min32(gy + (1 << bg->bppy), bg->y1);
            uint32_t tn;
            /* (a precinct that misses the band still gets the blocks its empty rectangle straddles: reference behaviour) */
            pb->ncw = cdiv_pow2(px1, bg->cbw) - (px0 >> bg->cbw);
            pb->nch = cdiv_pow2(py1, bg->cbh) - (py0 >> bg->cbh);
            if ((uint64_t)pb->ncw * (uint64_t)pb->nch > INT_MAX)
                return HTJ2K_ERR_ENOMEM;
            tn = tag_tree_nodes(pb->ncw, pb->nch);
            if (!tn)
                return HTJ2K_ERR_ENOMEM;
            pb->blk0 = gb->nblk;
            pb->node0 = gb->nnodes;
            pb->ntree = tn;
            if ((uint64_t)gb->nblk + (uint64_t)pb->ncw * pb->nch > (1u << 27) || (uint64_t)gb->nnodes + 2 * (uint64_t)tn > (1u << 29))
                return HTJ2K_ERR_ENOMEM;            /* resource limit of this implementation: 128 M blocks per frame */
            gb->nblk += (uint32_t)(pb->ncw * pb->nch);
            gb->nnodes += 2 * tn;
        }
    }
    gb->npb += (uint32_t)np * rg->nbands;
    return 0;
}

/* tile set-up (init_tile, jpeg2000dec.c:1016-1070, and the checks of ff_jpeg2000_init_component, jpeg2000.c:469-577) */
static int build_tile(GeomBuild *gb, int tileno)
{
    J2kParser *ps = gb->ps;
    TileHdr *t = &ps->tile[tileno];
    const int tx = tileno % (int)ps->tiles_x, ty = tileno / (int)ps->tiles_x;
    int c;
#define GRID_CLIP(v, lo, hi) ((v) < (lo) ? (lo) : (v) > (hi) ? (hi) : (int32_t)(v))
    t->x0 = GRID_CLIP((int64_t)tx * ps->xtsiz + ps->xtosiz, ps->xosiz, ps->xsiz);
    t->x1 = GRID_CLIP((int64_t)(tx + 1) * ps->xtsiz + ps->xtosiz, ps->xosiz, ps->xsiz);
    t->y0 = GRID_CLIP((int64_t)ty * ps->ytsiz + ps->ytosiz, ps->yosiz, ps->ysiz);
    t->y1 = GRID_CLIP((int64_t)(ty + 1) * ps->ytsiz + ps->ytosiz, ps->yosiz, ps->ysiz);
#undef GRID_CLIP
    for (c = 0; c < ps->ncomp; c++) {
        TcGeom *tc = &gb->g->tc[tileno * ps->ncomp + c];
        const CompCoding *k = &t->cod[c];
        int r, gband = 0, ret;
        tc->ox0 = cdiv(t->x0, ps->sub_x[c]); tc->ox1 = cdiv(t->x1, ps->sub_x[c]);
        tc->oy0 = cdiv(t->y0, ps->sub_y[c]); tc->oy1 = cdiv(t->y1, ps->sub_y[c]);
        tc->x0 = cdiv_pow2(tc->ox0, ps->reduce); tc->x1 = cdiv_pow2(tc->ox1, ps->reduce);
        tc->y0 = cdiv_pow2(tc->oy0, ps->reduce); tc->y1 = cdiv_pow2(tc->oy1, ps->reduce);
        if (!t->roi[c])
            t->roi[c] = ps->roi[c];
        if (!k->defined)
            return HTJ2K_ERR_INVALIDDATA;                   /* no COD reached this tile-component */
        if (ps->is_ht && !ps->ht_irrev && k->wavelet == J2K_DWT97) {
            cs_log(ps, LOGL_ERROR, "irreversible transform in a codestream of the HTREV set\n");
            return HTJ2K_ERR_INVALIDDATA;
        }
        if (ps->is_ht && ps->ht_kind != 0 && ps->ht_kind != (k->cb_style >> 6)) {
            cs_log(ps, LOGL_ERROR, "code-block style %02x contradicts Ccap15 bits 14-15\n", k->cb_style);
            return HTJ2K_ERR_INVALIDDATA;
        }
        if (!cs_picture_size_ok((uint32_t)(tc->x1 - tc->x0), (uint32_t)(tc->y1 - tc->y0), INT64_MAX))
            return HTJ2K_ERR_INVALIDDATA;
        if (tc->x1 - tc->x0 > 32768 || tc->y1 - tc->y0 > 32768) {
            cs_log(ps, LOGL_ERROR, "tile-component larger than 32768 samples in one direction\n");
            return HTJ2K_ERR_PATCHWELCOME;
        }
        tc->res = (ResGeom *)pool_get(&gb->g->pool, (size_t)k->nres * sizeof(ResGeom), 1);
        if (!tc->res)
            return HTJ2K_ERR_ENOMEM;
        for (r = 0; r < k->nres; r++)
            if ((ret = build_resolution(gb, tc, k, &t->q[c], r, ps->depth[c], &gband)) < 0)
                return ret;
    }
    return 0;
}

/* the component transform is undone only over three components of one shape (mct_decode, jpeg2000dec.c:2183-2197) */
static int tile_mct_usable(J2kParser *ps, int tileno)
{
    const TileHdr *t = &ps->tile[tileno];
    const TcGeom *tc = &ps->geo.tc[tileno * ps->ncomp];
    int c;
    if (!t->cod[0].mct || ps->ncomp < 3)
        return 0;
    for (c = 1; c < 3; c++) {
        if (t->cod[c].wavelet != t->cod[0].wavelet) {
            cs_log(ps, LOGL_ERROR, "component transform skipped: the components use different wavelets\n");
            return 0;
        }
        if (tc[c].x0 != tc[0].x0 || tc[c].x1 != tc[0].x1 || tc[c].y0 != tc[0].y0 || tc[c].y1 != tc[0].y1) {
            cs_log(ps, LOGL_ERROR, "component transform skipped: the components differ in size\n");
            return 0;
        }
    }
    return 1;
}

/* The rows of the plan's block table, in the order tile_codeblocks() visits blocks
 * (jpeg2000dec.c:2219-2289), with everything that does not depend on the packets; also the
 * tile-component table with write_frame's placement (jpeg2000dec.c:2301-2395). */
static int layout_rows(GeomBuild *gb)
{
    J2kParser *ps = gb->ps;
    GeomCache *g = gb->g;
    const J2kPixDesc *pd = j2k_pix_desc(ps->pix_fmt);
    const int planar = pd->planar, interleave = planar ? 1 : pd->nb_components;
    int cdef[J2K_MAX_COMPS], tileno, c, have_cdef = 1;
    size_t nsamples = 0;

    /* channel definitions default to "in order, an even count ends with alpha" (jpeg2000dec.c:2883-2892) */
    memcpy(cdef, ps->cdef, sizeof cdef);
    for (c = 0; c < ps->ncomp; c++)
        if (cdef[c] < 0)
            have_cdef = 0;
    if (!have_cdef) {
        for (c = 0; c < ps->ncomp; c++)
            cdef[c] = c + 1;
        if (!(ps->ncomp & 1))
            cdef[ps->ncomp - 1] = 0;
    }

    g->tcd = (J2kTileComp *)pool_get(&g->pool, (size_t)g->ntiles * ps->ncomp * sizeof(J2kTileComp), 1);
    if (!g->tcd)
        return HTJ2K_ERR_ENOMEM;
    for (tileno = 0; tileno < g->ntiles; tileno++) {
        const TileHdr *t = &ps->tile[tileno];
        const int mct = tile_mct_usable(ps, tileno);
        for (c = 0; c < ps->ncomp; c++) {
            const int tci = tileno * ps->ncomp + c;
            const TcGeom *tc = &g->tc[tci];
            const CompCoding *k = &t->cod[c];
            const CompQuant *q = &t->q[c];
            J2kTileComp *d = &g->tcd[tci];
            const int32_t ix = cdiv(ps->xosiz, ps->sub_x[c]), iy = cdiv(ps->yosiz, ps->sub_y[c]);
            int32_t bx[2] = { tc->x0, tc->x1 }, by[2] = { tc->y0, tc->y1 };
            int r, lev, gband = 0;

            d->comp = c; d->tile = tileno;
            d->x0 = tc->x0; d->x1 = tc->x1; d->y0 = tc->y0; d->y1 = tc->y1;
            d->w = tc->x1 - tc->x0;
            d->h = tc->y1 - tc->y0;
            d->transform = k->wavelet;
            d->ndeclevels = k->nres_dec - 1;
            /* line lengths and origin parities of the synthesis levels, finest first (ff_jpeg2000_dwt_init, jpeg2000dwt.c:554-560) */
            for (lev = d->ndeclevels - 1; lev >= 0; lev--) {
                if (lev < J2K_MAX_DWTLEV) {
                    d->linelen[lev][0] = bx[1] - bx[0]; d->mod[lev][0] = bx[0] & 1;
                    d->linelen[lev][1] = by[1] - by[0]; d->mod[lev][1] = by[0] & 1;
                }
                bx[0] = (bx[0] + 1) >> 1; bx[1] = (bx[1] + 1) >> 1;
                by[0] = (by[0] + 1) >> 1; by[1] = (by[1] + 1) >> 1;
            }
            if (nsamples + (size_t)d->w * d->h > 0xFFFFFFF0u) {
                g->static_err = HTJ2K_ERR_PATCHWELCOME;     /* sample offsets are 32-bit */
                goto done;
            }
            d->plane_off = (uint32_t)nsamples;
            nsamples += ((size_t)d->w * d->h + 63) & ~(size_t)63;
            d->cbps = ps->depth[c];
            d->mct = mct && c < 3;
            d->out_plane = planar ? (cdef[c] ? cdef[c] - 1 : ps->ncomp - 1) : 0;
            d->out_x = tc->x0 - ix;
            d->out_y = tc->y0 - iy;
            d->out_w = tc->x1 - ix - d->out_x;
            d->out_h = tc->y1 - iy - d->out_y;
            d->pix_step = interleave;
            d->pix_off = planar ? 0 : c;

            for (r = 0; r < k->nres_dec; r++) {
                const ResGeom *rg = &tc->res[r];
                const int np = rg->npx * rg->npy;
                int b, p;
                for (b = 0; b < rg->nbands; b++, gband++) {
                    const BandGeom *bg = &rg->band[b];
                    const int orient = b + (r > 0), M_b = q->expn[gband] + q->guard - 1;
                    /* where the band sits in the Mallat layout of the plane (jpeg2000.c:365-376) */
                    const int32_t shift_x = (orient & 1) ? tc->res[r - 1].x1 - tc->res[r - 1].x0 : 0;
                    const int32_t shift_y = (orient & 2) ? tc->res[r - 1].y1 - tc->res[r - 1].y0 : 0;
                    float scale97 = 0.0f;
                    if (bg->x0 == bg->x1 || bg->y0 == bg->y1)
                        continue;
                    if ((k->cb_style & CBS_HT) && M_b >= 31) {
                        cs_log(ps, LOGL_ERROR, "HT code-blocks with %d magnitude bits\n", M_b);
                        g->static_err = HTJ2K_ERR_PATCHWELCOME;
                        goto done;
                    }
                    if (k->wavelet == J2K_DWT97_INT) {      /* dequantization_int_97's scale (jpeg2000dec.c:2159-2168) */
                        scale97 = bg->fstep;
                        scale97 /= (float)(int32_t)(1u << ((31 - M_b) & 31));   /* (M_b outside 0..30 only in broken streams) */
                        scale97 *= 64.0f;
                        scale97 *= (float)(1 << 24);
                    }
                    for (p = 0; p < np; p++) {
                        const PrecBand *pb = &g->pb[rg->pb0 + (uint32_t)b * (uint32_t)np + (uint32_t)p];
                        const int32_t gx = ((rg->x0 >> rg->ppx) + p % rg->npx) * (1 << bg->bppx);
                        const int32_t gy = ((rg->y0 >> rg->ppy) + p / rg->npx) * (1 << bg->bppy);
                        const int32_t px0 = max32(gx, bg->x0), px1 = min32(gx + (1 << bg->bppx), bg->x1);
                        const int32_t py0 = max32(gy, bg->y0), py1 = min32(gy + (1 << bg->bppy), bg->y1);
                        const int32_t ax = (px0 >> bg->cbw) << bg->cbw, ay = (py0 >> bg->cbh) << bg->cbh;   /* block grid anchor */
                        int i, j;
                        for (j = 0; j < pb->nch; j++)
                            for (i = 0; i < pb->ncw; i++) {
                                const int32_t cx0 = ax + (i << bg->cbw), cy0 = ay + (j << bg->cbh);
                                const int32_t x0 = max32(cx0, px0), x1 = min32(cx0 + (1 << bg->cbw), px1);
                                const int32_t y0 = max32(cy0, py0), y1 = min32(cy0 + (1 << bg->cbh), py1);
                                const int32_t bw = x1 - x0, bh = y1 - y0;
                                const int32_t wx = x0 + shift_x - bg->x0, wy = y0 + shift_y - bg->y0;   /* window in the plane */
                                J2kBlock *row;
                                if (bw <= 0 || bh <= 0)
                                    continue;
                                if (wx < 0 || wy < 0 || wx + bw > d->w || wy + bh > d->h ||
                                    bw > 1024 || bh > 1024 || bw * bh > 4096) {
                                    g->static_err = HTJ2K_ERR_INVALIDDATA;   /* the reference would write outside the plane
                                                                              * resp. assert (jpeg2000htdec.c:1230-1231) */
                                    goto done;
                                }
                                if (gb->nrows == gb->rows_cap) {     /* the per-row tables grow in step */
                                    uint32_t c1 = gb->rows_cap, c2 = gb->rows_cap, c3 = gb->rows_cap, c4 = gb->rows_cap;
                                    if (grow((void **)&gb->rows, &c1, gb->nrows + 1, sizeof(J2kBlock)) < 0 ||
                                        grow((void **)&gb->row_blk, &c2, gb->nrows + 1, sizeof(uint32_t)) < 0 ||
                                        grow((void **)&gb->row_aux, &c3, gb->nrows + 1, sizeof(uint16_t)) < 0 ||
                                        grow((void **)&gb->row_tc, &c4, gb->nrows + 1, sizeof(uint32_t)) < 0)
                                        return HTJ2K_ERR_ENOMEM;
                                    gb->rows_cap = c1;
                                }
                                row = &gb->rows[gb->nrows];
                                memset(row, 0, sizeof *row);
                                row->plane_off = d->plane_off + (uint32_t)wy * (uint32_t)d->w + (uint32_t)wx;
                                row->w = (uint16_t)bw;
                                row->h = (uint16_t)bh;
                                row->stride = (uint16_t)d->w;
                                row->M_b = (uint8_t)M_b;
                                row->flags = (uint8_t)(k->wavelet & 3);
                                row->roi_shift = t->roi[c];
                                row->tcomp = (uint8_t)tci;
                                row->f_step = bg->fstep;
                                row->i_step = k->wavelet == J2K_DWT97_INT ? (int32_t)(scale97 + 0.5) : bg->istep;
                                gb->row_aux[gb->nrows] = (uint16_t)(k->cb_style << 8 | orient);
                                gb->row_tc[gb->nrows] = (uint32_t)tci;
                                gb->row_blk[gb->nrows++] = pb->blk0 + (uint32_t)(j * pb->ncw + i);
                            }
                    }
                }
            }
        }
    }
done:
    g->nsamples = nsamples;
    return 0;
}

/* everything the tables depend on, as bytes: equal bytes, equal tables */
static int build_signature(J2kParser *ps)
{
    struct {
        int32_t siz[8]; int32_t ncomp, reduce, pix_fmt, cdef[J2K_MAX_COMPS], sub[2 * J2K_MAX_COMPS];
        uint8_t depth[J2K_MAX_COMPS], roi[J2K_MAX_COMPS], ht[4]; int64_t budget; uint32_t tiles[2];
    } head;
    const uint32_t ntiles = ps->tiles_x * ps->tiles_y;
    uint32_t t;
    int c, r;
    memset(&head, 0, sizeof head);
    head.siz[0] = ps->xsiz; head.siz[1] = ps->ysiz; head.siz[2] = ps->xosiz; head.siz[3] = ps->yosiz;
    head.siz[4] = ps->xtsiz; head.siz[5] = ps->ytsiz; head.siz[6] = ps->xtosiz; head.siz[7] = ps->ytosiz;
    head.ncomp = ps->ncomp; head.reduce = ps->reduce; head.pix_fmt = ps->pix_fmt;
    for (c = 0; c < J2K_MAX_COMPS; c++) {
        head.cdef[c] = ps->cdef[c];
        head.sub[2 * c] = ps->sub_x[c]; head.sub[2 * c + 1] = ps->sub_y[c];
        head.depth[c] = ps->depth[c]; head.roi[c] = ps->roi[c];
    }
    head.ht[0] = ps->is_ht; head.ht[1] = ps->ht_kind; head.ht[2] = ps->ht_irrev;
    head.budget = pixel_budget(ps);
    head.tiles[0] = ps->tiles_x; head.tiles[1] = ps->tiles_y;
    ps->sig_len = 0;
    if ((r = cs_sig_append(ps, &head, sizeof head)) < 0 ||
        (r = cs_sig_append(ps, ps->cod, sizeof ps->cod)) < 0 || (r = cs_sig_append(ps, ps->q, sizeof ps->q)) < 0)
        return r;
    for (t = 0; t < ntiles; t++) {
        const TileHdr *th = &ps->tile[t];
        if ((r = cs_sig_append(ps, &th->own_params, 1)) < 0)
            return r;
        if (th->own_params & TILE_OWN_PARAMS)
            if ((r = cs_sig_append(ps, th->cod, sizeof th->cod)) < 0 || (r = cs_sig_append(ps, th->q, sizeof th->q)) < 0 ||
                (r = cs_sig_append(ps, th->roi, sizeof th->roi)) < 0)
                return r;
    }
    return 0;
}

int t2_build_geometry(J2kParser *ps)
{
    GeomCache *g = &ps->geo;
    GeomBuild gb;
    int r, tileno;

    if ((r = build_signature(ps)) < 0)
        return r;
    if (g->valid && g->sig_len == ps->sig_len && !memcmp(g->sig, ps->sig, ps->sig_len)) {
        /* same headers as the frame before: only the tile rectangles and ROI defaults live in per-frame state */
        for (tileno = 0; tileno < g->ntiles; tileno++) {
            TileHdr *t = &ps->tile[tileno];
            int c;
            for (c = 0; c < ps->ncomp; c++)
                if (!t->roi[c])
                    t->roi[c] = ps->roi[c];
            t->x0 = g->tile_rect[4 * tileno]; t->x1 = g->tile_rect[4 * tileno + 1];
            t->y0 = g->tile_rect[4 * tileno + 2]; t->y1 = g->tile_rect[4 * tileno + 3];
        }
        return 0;
    }

    g->valid = 0;
    pool_rewind(&g->pool);
    free(g->pb); free(g->rows); free(g->row_blk); free(g->row_aux); free(g->row_tc);
    g->pb = NULL; g->rows = NULL; g->row_blk = NULL; g->row_aux = NULL; g->row_tc = NULL;
    g->nrows = 0;
    g->ntiles = (int)(ps->tiles_x * ps->tiles_y);
    g->ncomp = ps->ncomp;
    g->static_err = 0;
    g->tc = (TcGeom *)pool_get(&g->pool, (size_t)g->ntiles * ps->ncomp * sizeof(TcGeom), 1);
    g->tile_err = (int *)pool_get(&g->pool, (size_t)g->ntiles * sizeof(int), 1);
    g->tile_rect = (int32_t *)pool_get(&g->pool, (size_t)g->ntiles * 4 * sizeof(int32_t), 1);
    if (!g->tc || !g->tile_err || !g->tile_rect)
        return HTJ2K_ERR_ENOMEM;
    memset(&gb, 0, sizeof gb);
    gb.ps = ps;
    gb.g = g;
    for (tileno = 0; tileno < g->ntiles; tileno++) {
        g->tile_err[tileno] = build_tile(&gb, tileno);
        g->tile_rect[4 * tileno] = ps->tile[tileno].x0; g->tile_rect[4 * tileno + 1] = ps->tile[tileno].x1;
        g->tile_rect[4 * tileno + 2] = ps->tile[tileno].y0; g->tile_rect[4 * tileno + 3] = ps->tile[tileno].y1;
        if (g->tile_err[tileno] < 0) {
            /* the packets of the tiles in front are still read before this surfaces; nothing behind is reached */
            for (r = tileno + 1; r < g->ntiles; r++)
                g->tile_err[r] = g->tile_err[tileno];
            break;
        }
    }
    g->pb = gb.pb; g->npb = gb.npb;
    g->nblk = gb.nblk; g->nnodes = gb.nnodes; g->nprec = gb.nprec;
    g->first_bad_tile = tileno < g->ntiles ? tileno : -1;
    if (g->first_bad_tile < 0) {
        r = layout_rows(&gb);
        g->rows = gb.rows; g->row_blk = gb.row_blk; g->row_aux = gb.row_aux; g->row_tc = gb.row_tc; g->nrows = gb.nrows;
        if (r < 0)
            return r;
    }
    /* remember what this was built from */
    if (g->sig_cap < ps->sig_len) {
        uint8_t *nb = (uint8_t *)realloc(g->sig, ps->sig_len);
        if (!nb)
            return HTJ2K_ERR_ENOMEM;
        g->sig = nb;
        g->sig_cap = ps->sig_len;
    }
    memcpy(g->sig, ps->sig, ps->sig_len);
    g->sig_len = ps->sig_len;
    g->valid = 1;
    return 0;
}

/* ================================================================== packet headers */

/* MSB-first bit window over a packet header.  A byte that follows 0xFF carries seven bits
 * (its top bit is a stuffing bit, T.800 B.10.1); beyond the end of the data the window supplies
 * zeros, as the reference's reader does (get_bits, jpeg2000dec.c:70-83). */
typedef struct BitWin {
    uint64_t acc;               /* unread bits, left-aligned */
    int n;                      /* how many */
    int fed_past_end;           /* zero bits supplied beyond `end` */
    const uint8_t *next, *end, *first;
    uint32_t prev;              /* the byte loaded last (0 at the start: the first byte is a full one) */
} BitWin;

static void bw_open(BitWin *w, const Cur *c)
{
    w->acc = 0; w->n = 0; w->fed_past_end = 0;
    w->next = w->first = c->p;
    w->end = c->end;
    w->prev = 0;
}

static inline void bw_fill(BitWin *w)
{
    while (w->n <= 56) {
        if (w->next < w->end) {
            const uint32_t b = *w->next++;
            const int width = w->prev == 0xFF ? 7 : 8;
            w->acc |= (uint64_t)(b & (0xFFu >> (8 - width))) << (64 - width - w->n);
            w->n += width;
            w->prev = b;
        } else {
            w->n += 8;
            w->fed_past_end += 8;
        }
    }
}

static inline uint32_t bw_take(BitWin *w, int k)       /* 0 <= k <= 32 */
{
    uint32_t v;
    if (!k)
        return 0;
    if (w->n < k)
        bw_fill(w);
    v = (uint32_t)(w->acc >> (64 - k));
    w->acc <<= k;
    w->n -= k;
    return v;
}

/* Where the data continues behind the header: behind the byte that holds the last bit read, and
 * behind one more if that byte is 0xFF (its successor would only carry a stuffing bit first):
 * jpeg2000_flush, jpeg2000dec.c:85-90. */
static const uint8_t *bw_close(const BitWin *w)
{
    const uint8_t *q = w->next;
    int unread = w->n - w->fed_past_end;               /* of the bits that came from real bytes */
    if (unread < 0)
        return w->end;                                 /* reading ran off the end */
    while (q > w->first) {
        const int width = (q - 1 > w->first && q[-2] == 0xFF) ? 7 : 8;
        if (unread < width)
            break;
        unread -= width;
        q--;
    }
    if (q == w->first)                                 /* nothing was read (cannot happen: a header has at least one bit) */
        return q;
    if (q[-1] == 0xFF && q < w->end)
        q++;
    return q;
}

/* Tag tree (T.800 B.10.2) over an ncw x nch grid as a flat array: level 0 = the leaves in raster
 * order, each further level half the size, the root last.  A node is its running value in the low
 * byte and a "value is final" flag in bit 8.  Decoding against a threshold reads bits exactly as
 * tag_tree_decode does (jpeg2000dec.c:93-131): from the first unresolved ancestor down, one 0 bit
 * per increment, a 1 bit fixes a node, and the descent stops as soon as the value reaches the
 * threshold. */
#define NODE_FINAL 0x100

static int tagtree_read(BitWin *w, uint16_t *tree, int ncw, int nch, int x, int y, int threshold)
{
    uint32_t path[32];
    int depth = 0, lw = ncw, lh = nch, value, d;
    uint32_t off = 0;
    /* the chain leaf -> root */
    for (;;) {
        path[depth++] = off + (uint32_t)(y * lw + x);
        if (lw <= 1 && lh <= 1)
            break;
        off += (uint32_t)(lw * lh);
        lw = (lw + 1) >> 1; lh = (lh + 1) >> 1;
        x >>= 1; y >>= 1;
    }
    /* the part of it that is still open: up to (not including) the first final node */
    for (d = 0; d < depth && !(tree[path[d]] & NODE_FINAL); d++)
        ;
    value = d < depth ? (tree[path[d]] & 0xFF) : (tree[path[depth - 1]] & 0xFF);
    while (value < threshold && d > 0) {
        uint16_t *node = &tree[path[--d]];
        if (value < (*node & 0xFF))
            value = *node & 0xFF;
        while (value < threshold) {
            if (bw_take(w, 1)) {
                *node |= NODE_FINAL;
                break;
            }
            value++;
        }
        *node = (uint16_t)((*node & NODE_FINAL) | (value & 0xFF));
    }
    return value;
}

/* number of coding passes (T.800 Table B.4) */
static int read_pass_count(BitWin *w)
{
    uint32_t v;
    if (!bw_take(w, 1)) return 1;
    if (!bw_take(w, 1)) return 2;
    if ((v = bw_take(w, 2)) != 3) return 3 + (int)v;
    if ((v = bw_take(w, 5)) != 31) return 6 + (int)v;
    return 37 + (int)bw_take(w, 7);
}

/* does Part-1 pass `pass` end a terminated codeword segment under these mode switches
 * (needs_termination, jpeg2000.h:302-317: every pass with TERMALL; with BYPASS the raw passes and
 * the cleanup pass in front of them from the fourth bit-plane on) */
static int pass_terminates(int style, int pass)
{
    if (style & CBS_TERMALL)
        return 1;
    return (style & CBS_BYPASS) && pass / 3 > 2 && pass % 3 != 1;
}

typedef struct Contribution { uint32_t blk; uint32_t len; uint32_t nterm; uint32_t first; } Contribution;

typedef struct PacketCtx {
    J2kParser *ps;
    TileHdr *tile;
    int tileno;
    int part;                                  /* tile-part the packets currently come from */
    /* lengths signalled by the header being read, in body order */
    uint32_t *lens; uint32_t nlens, lens_cap;
    Contribution *con; uint32_t ncon, con_cap;
    /* a worker of the parallel reader (read_tile_parallel): the packet sits at *at, its end goes to end_p, nothing is said
     * (anything worth a message makes the tile `anomalous` and the frame is parsed again the sequential way, which
     * then says it), and chained pieces come out of the worker's own slice of the table */
    const Cur *at;
    const uint8_t *end_p;
    int quiet, anomaly;
    uint32_t arena_next, arena_limit;
} PacketCtx;

#define PKT_LOG(pc, level, ...) do { if ((pc)->quiet) (pc)->anomaly = 1; else cs_log((pc)->ps, level, __VA_ARGS__); } while (0)

static int note_length(PacketCtx *pc, uint32_t len)
{
    if (pc->nlens == pc->lens_cap) {
        const uint32_t nc = pc->lens_cap ? pc->lens_cap * 2 : 1024;
        uint32_t *nl = (uint32_t *)realloc(pc->lens, (size_t)nc * sizeof *nl);
        if (!nl)
            return HTJ2K_ERR_ENOMEM;
        pc->lens = nl;
        pc->lens_cap = nc;
    }
    pc->lens[pc->nlens++] = len;
    return 0;
}

/* one more piece of a block's byte string; adjacent unterminated pieces become one */
static int add_piece(PacketCtx *pc, BlkState *s, uint32_t src, uint32_t len, int term)
{
    J2kParser *ps = pc->ps;
    SegNode *last = s->last ? &ps->segs[s->last - 1] : NULL;
    uint32_t idx;
    if (!(s->flags & BS_HAS_BYTES)) {
        s->flags |= BS_HAS_BYTES;
        s->first_src = src; s->first_len = len; s->first_term = (uint8_t)term;
        return 0;
    }
    if (last ? (!last->term && last->src + last->len == src) : (!s->first_term && s->first_src + s->first_len == src)) {
        if (last) { last->len += len; last->term = (uint32_t)term; }
        else      { s->first_len += len; s->first_term = (uint8_t)term; }
        return 0;
    }
    if (pc->arena_limit) {                                 /* (the table does not move while the workers run) */
        if (pc->arena_next == pc->arena_limit) {
            pc->anomaly = 1;
            return HTJ2K_ERR_BUG;
        }
        idx = pc->arena_next++;
    } else {
        if (ps->nsegs == ps->segs_cap) {
            const uint32_t nc = ps->segs_cap ? ps->segs_cap * 2 : 4096;
            SegNode *ns = (SegNode *)realloc(ps->segs, (size_t)nc * sizeof *ns);
            if (!ns)
                return HTJ2K_ERR_ENOMEM;
            ps->segs = ns;
            ps->segs_cap = nc;
            last = s->last ? &ps->segs[s->last - 1] : NULL;
        }
        idx = ps->nsegs++;
    }
    ps->segs[idx].src = src; ps->segs[idx].len = len;
    ps->segs[idx].term = (uint32_t)term; ps->segs[idx].next = 0;
    if (last) last->next = idx + 1; else s->more = idx + 1;
    s->last = idx + 1;
    return 0;
}

/* ---- which stream a packet's header and body come from (select_header / select_stream,
 *      jpeg2000dec.c:1099-1134): tile-part bodies in TPsot order, headers from there too unless
 *      PPM / PPT moved them ---- */
static Cur body_stream(PacketCtx *pc, const CompCoding *k)
{
    Cur c = pc->tile->part[pc->part].body;
    while (!cur_left(&c) && pc->part < CS_MAX_TPARTS - 1)
        c = pc->tile->part[++pc->part].body;
    if (k->scod & SCOD_SOP) {
        if (cur_peek32(&c) == 0xFF910004u)
            cur_skip(&c, 6);
        else
            PKT_LOG(pc, LOGL_ERROR, "no SOP marker in front of a packet (%08x)\n", (unsigned)cur_peek32(&c));
    }
    return c;
}

static Cur header_stream(PacketCtx *pc, const CompCoding *k)
{
    J2kParser *ps = pc->ps;
    if (ps->has_ppm) {
        Cur c = pc->tile->part[pc->part].hdr;
        if (!cur_left(&c)) {
            PKT_LOG(pc, LOGL_WARNING, "the packed packet headers of tile-part %d are used up\n", pc->part);
            if (pc->part < CS_MAX_TPARTS - 1)
                c = pc->tile->part[++pc->part].body;       /* (the body, not the headers: reference behaviour) */
        }
        return c;
    }
    if (pc->tile->has_ppt)
        return pc->tile->ppt_cur;
    return body_stream(pc, k);
}

/* after the header: EPH, hand the header stream back, switch to the body stream */
static Cur header_done(PacketCtx *pc, const CompCoding *k, Cur c)
{
    J2kParser *ps = pc->ps;
    if (k->scod & SCOD_EPH) {
        if (cur_peek16(&c) == 0xFF92)
            cur_skip(&c, 2);
        else
            PKT_LOG(pc, LOGL_ERROR, "no EPH marker behind a packet header (%08x)\n", (unsigned)cur_peek32(&c));
    }
    if (ps->has_ppm) {
        pc->tile->part[pc->part].hdr = c;
        c = body_stream(pc, k);
    } else if (pc->tile->has_ppt) {
        pc->tile->ppt_cur = c;
        c = body_stream(pc, k);
    }
    return c;
}

/* ---- lengths of the codeword segments a packet adds to a block (T.800 B.10.7, T.814 B.4;
 *      jpeg2000dec.c:1240-1430).  `fresh` passes arrive; every segment length is a field of
 *      lblock + floor(log2(passes in the segment)) bits.  Returns < 0 on failure. ---- */
static int read_segment_lengths(PacketCtx *pc, BitWin *w, BlkState *s, int fresh)
{
    int seg, nbits, follow = 0, bypass = 0, left, r;
    uint32_t bytes;

    if (s->flags & BS_PLACEHOLD) {
        /* An HT block that has shown no cleanup segment yet.  Its passes come in sets of three; if the
         * fresh passes reach into a set, the first segment is that set's cleanup pass, otherwise they are
         * placeholders (length 0) -- or, in a MIXED stream, this is a Part-1 block after all. */
        const int beyond = (s->npasses + fresh - 1) % 3;
        int limit = 2;
        seg = fresh - beyond;
        nbits = s->lblock;
        if (seg < 1) {
            seg = fresh;
            for (; limit <= seg; limit += limit)
                nbits++;
            bytes = bw_take(w, nbits);
            if (bytes) {
                if (s->style & CBS_HT_MIXED) {
                    s->flags &= (uint8_t)~BS_PLACEHOLD;
                    s->style &= (uint8_t)~CBS_HT;
                } else {
                    PKT_LOG(pc, LOGL_WARNING, "HT block: bytes signalled for placeholder passes\n");
                }
            }
        } else {
            for (; limit <= seg; limit += limit)
                nbits++;
            bytes = bw_take(w, nbits);
            if (bytes) {
                const int looks_ht = !(s->style & CBS_HT_MIXED) ||
                                     (s->lblock > 3 && bytes > 1 && !(bytes >> (nbits - 1)));
                s->flags &= (uint8_t)~BS_PLACEHOLD;
                if (looks_ht) {
                    if (bytes < 2)
                        PKT_LOG(pc, LOGL_WARNING, "HT block: cleanup segment of %u byte\n", (unsigned)bytes);
                    follow = 2;
                    s->lcup = bytes;
                } else {
                    /* a Part-1 block of a MIXED stream: all fresh passes are one segment, whose length field
                     * is as much longer as the pass count demands */
                    s->style &= (uint8_t)~CBS_HT;
                    seg = fresh;
                    for (; limit <= seg; limit += limit)
                        bytes = (bytes << 1) | bw_take(w, 1);
                }
            } else {
                /* no bytes for a cleanup pass: placeholder passes, all fresh ones; a longer field decides */
                seg = fresh;
                if (limit <= seg) {
                    do {
                        bytes = (bytes << 1) | bw_take(w, 1);
                        limit += limit;
                    } while (limit <= seg);
                    if (bytes) {
                        if (s->style & CBS_HT_MIXED) {
                            s->style &= (uint8_t)~CBS_HT;
                            s->flags &= (uint8_t)~BS_PLACEHOLD;
                        } else {
                            PKT_LOG(pc, LOGL_WARNING, "HT block: bytes signalled for placeholder passes\n");
                        }
                    }
                }
            }
        }
    } else if (s->style & CBS_HT) {
        /* an HT block past its cleanup pass: refinement passes, SigProp + MagRef share a segment */
        const int in_set = s->npasses % 3;
        nbits = 0;
        if (!in_set) {
            seg = 1; follow = 2;
        } else {
            seg = fresh > 1 ? 3 - in_set : 1;
            follow = 1;
            nbits = ilog2u((uint32_t)seg);
        }
        bytes = bw_take(w, nbits + s->lblock);
        s->lref += bytes;
    } else if (!(s->style & (CBS_TERMALL | CBS_BYPASS))) {
        seg = fresh;
        bytes = bw_take(w, s->lblock + ilog2u((uint32_t)(uint8_t)fresh));
    } else if (s->style & CBS_TERMALL) {
        seg = 1; follow = 1;
        bytes = bw_take(w, s->lblock);
    } else {
        /* BYPASS: the first ten passes are one MQ segment, then raw (2 passes) and MQ (1) alternate */
        bypass = 10;
        nbits = 0;
        if (s->npasses < bypass) {
            seg = min32(bypass - s->npasses, fresh);
            while ((2 << nbits) <= seg)
                nbits++;
            follow = 2;
        } else if ((s->npasses - bypass) % 3 < 2) {
            seg = fresh > 1 ? 2 - (s->npasses - bypass) % 3 : 1;
            nbits = ilog2u((uint32_t)seg);
            follow = 1;
        } else {
            seg = 1; follow = 2;
        }
        bytes = bw_take(w, nbits + s->lblock);
    }
    s->npasses = (uint8_t)(s->npasses + seg);
    if ((r = note_length(pc, bytes)) < 0)
        return r;

    left = fresh - (uint8_t)seg;
    if ((s->style & CBS_HT) && !(s->flags & BS_PLACEHOLD)) {
        while (left > 0) {                                 /* HT refinement segments */
            seg = left > 1 ? follow : 1;
            follow = 3 - follow;
            bytes = bw_take(w, s->lblock + ilog2u((uint32_t)seg));
            left -= seg;
            s->lref += bytes;
            s->npasses = (uint8_t)(s->npasses + seg);
            if ((r = note_length(pc, bytes)) < 0)
                return r;
        }
    } else {
        while (left > 0) {
            if (bypass) {
                seg = left > 1 ? follow : 1;
                follow = 3 - follow;
                nbits = s->lblock + ilog2u((uint32_t)seg);
            } else {
                if (!(s->style & CBS_TERMALL))
                    PKT_LOG(pc, LOGL_WARNING, "packet header: more passes than its one segment can hold\n");
                seg = 1;
                nbits = s->lblock;
            }
            bytes = bw_take(w, nbits);
            left -= seg;
            s->npasses = (uint8_t)(s->npasses + seg);
            if ((r = note_length(pc, bytes)) < 0)
                return r;
        }
    }
    return 0;
}

/* one packet: (component, resolution, precinct, layer).  jpeg2000_decode_packet, jpeg2000dec.c:1136-1542 */
static int read_packet(PacketCtx *pc, int comp, int r, int prec, int layer)
{
    J2kParser *ps = pc->ps;
    GeomCache *g = &ps->geo;
    const TcGeom *tc = &g->tc[pc->tileno * ps->ncomp + comp];
    const ResGeom *rg = &tc->res[r];
    const CompCoding *k = &pc->tile->cod[comp];
    const CompQuant *q = &pc->tile->q[comp];
    const uint8_t *expn = q->expn + (r ? 3 * (r - 1) + 1 : 0);
    const int np = rg->npx * rg->npy;
    uint8_t *done = &ps->layers_done[rg->lay0 + (uint32_t)prec];
    BitWin w;
    Cur c;
    int b, ret;
    uint32_t i;

    if (pc->at) {                                          /* parallel reader: the list was made with the revisits left out */
        c = *pc->at;
        if (k->scod & SCOD_SOP) {
            if (cur_peek32(&c) == 0xFF910004u)
                cur_skip(&c, 6);
            else
                pc->anomaly = 1;
        }
    } else {
        if (layer < *done)
            return 0;                                      /* a progression change revisits the packet */
        *done = (uint8_t)(layer + 1);
        c = header_stream(pc, k);
    }
    bw_open(&w, &c);
    pc->nlens = pc->ncon = 0;
    if (bw_take(&w, 1)) {                                  /* the packet is not empty */
        for (b = 0; b < rg->nbands; b++) {
            const BandGeom *bg = &rg->band[b];
            const PrecBand *pb = &g->pb[rg->pb0 + (uint32_t)b * (uint32_t)np + (uint32_t)prec];
            uint16_t *incl_tree = ps->nodes + pb->node0, *zbp_tree;
            const int nblk = pb->ncw * pb->nch;
            int n, bx, by;
            if (bg->x0 == bg->x1 || bg->y0 == bg->y1)
                continue;
            zbp_tree = incl_tree + pb->ntree;
            for (n = 0, bx = 0, by = 0; n < nblk; n++, bx++) {
                BlkState *s = &ps->blk[pb->blk0 + (uint32_t)n];
                int included, fresh, grow_by, p;
                Contribution *cn;
                if (bx == pb->ncw) {
                    bx = 0;
                    by++;
                }

                if (!(s->flags & BS_INCLUDED)) {
                    /* first inclusion is tag-tree coded against the layer index */
                    s->style = k->cb_style;
                    if (s->style >= CBS_HT)
                        s->flags |= BS_PLACEHOLD;
                    if (layer > 0)
                        tagtree_read(&w, incl_tree, pb->ncw, pb->nch, bx, by, 1);
                    included = tagtree_read(&w, incl_tree, pb->ncw, pb->nch, bx, by, layer + 1) == layer;
                    if (included) {
                        const int zbp = tagtree_read(&w, zbp_tree, pb->ncw, pb->nch, bx, by, 100);
                        const int nzb = expn[b] + q->guard - 1 - (zbp - pc->tile->roi[0]);
                        if (nzb < 0 || nzb > 30) {
                            PKT_LOG(pc, LOGL_ERROR, "code-block with %d magnitude bit-planes\n", nzb);
                            return HTJ2K_ERR_INVALIDDATA;
                        }
                        s->flags |= BS_INCLUDED;
                        s->nzb = (uint8_t)nzb;
                        s->zbp = (uint8_t)zbp;
                        s->lblock = 3;
                    }
                } else {
                    included = (int)bw_take(&w, 1);
                }
                if (!included)
                    continue;

                fresh = read_pass_count(&w);
                if (s->npasses + fresh >= CS_MAX_PASSES) {
                    PKT_LOG(pc, LOGL_ERROR, "code-block with %d coding passes\n", s->npasses + fresh);
                    return HTJ2K_ERR_PATCHWELCOME;
                }
                for (grow_by = 0; bw_take(&w, 1); grow_by++)   /* Lblock increment: a comma code */
                    ;
                if (s->lblock + grow_by + ilog2u((uint32_t)fresh) > 16) {
                    PKT_LOG(pc, LOGL_ERROR, "code-block segment length field wider than 16 bits\n");
                    return HTJ2K_ERR_PATCHWELCOME;
                }
                s->lblock = (uint8_t)(s->lblock + grow_by);

                if (pc->ncon == pc->con_cap) {
                    const uint32_t nc = pc->con_cap ? pc->con_cap * 2 : 1024;
                    Contribution *ncn = (Contribution *)realloc(pc->con, (size_t)nc * sizeof *ncn);
                    if (!ncn)
                        return HTJ2K_ERR_ENOMEM;
                    pc->con = ncn;
                    pc->con_cap = nc;
                }
                cn = &pc->con[pc->ncon++];
                cn->blk = pb->blk0 + (uint32_t)n;
                cn->first = pc->nlens;
                cn->nterm = 0;
                if (!(s->style & CBS_HT))                  /* Part-1: which of the fresh passes end a terminated segment */
                    for (p = 0; p < fresh; p++)
                        cn->nterm += (uint32_t)pass_terminates(k->cb_style, s->npasses + p);
                if ((ret = read_segment_lengths(pc, &w, s, fresh)) < 0)
                    return ret;
                cn->len = pc->nlens - cn->first;
            }
        }
    }
    c.p = bw_close(&w);
    if (pc->at && w.fed_past_end)
        pc->anomaly = 1;                                   /* the header ran past the end the list gives the packet */
    c = header_done(pc, k, c);

    /* the body: the signalled bytes of every contributing block, in header order */
    for (i = 0; i < pc->ncon; i++) {
        const Contribution *cn = &pc->con[i];
        BlkState *s = &ps->blk[cn->blk];
        uint32_t todo_term = cn->nterm, j;
        for (j = 0; j < cn->len; j++) {
            const uint32_t len = pc->lens[cn->first + j];
            /* a block's bytes are counted in 16 bits (Jpeg2000Cblk.length, jpeg2000.h:188) */
            if ((uint32_t)cur_left(&c) < len || s->length + len > 65535u || (todo_term && s->length + len + 2 > 65535u)) {
                PKT_LOG(pc, LOGL_ERROR, "code-block bytes: %u so far, %u more signalled, %d left in the tile-part\n",
                       (unsigned)s->length, (unsigned)len, cur_left(&c));
                return HTJ2K_ERR_INVALIDDATA;
            }
            if (len || todo_term) {
                /* the plan will want Scup, the last two bytes of an HT block's cleanup segment: have them on their way */
                if (!j && len > 1 && (s->style & CBS_HT) && !(s->flags & BS_HAS_BYTES))
                    __builtin_prefetch(c.p + len - 1);
                if ((ret = add_piece(pc, s, (uint32_t)(c.p - ps->pkt), len, todo_term != 0)) < 0)
                    return ret;
            }
            c.p += len;
            s->length += len;
            if (todo_term) {                               /* 0xFF 0xFF will sit behind a terminated Part-1 segment */
                todo_term--;
                s->nterm++;
                s->length += 2;
            }
        }
    }
    if (pc->at) {
        pc->end_p = c.p;
        return 0;
    }
    pc->tile->part[pc->part].body = c;
    ps->g = c;
    return 0;
}

/* ================================================================== packet order
 * T.800 B.12: a progression volume (resolutions RS..RE, components CS..CE, layers 0..LYE) is
 * walked in one of five nestings.  One recursive walker runs whatever nesting the table gives it. */
enum { AX_LAYER, AX_RES, AX_COMP, AX_PREC, AX_Y, AX_X, AX_END };
static const uint8_t nesting[5][6] = {
    { AX_LAYER, AX_RES, AX_COMP, AX_PREC, AX_END },        /* LRCP */
    { AX_RES, AX_LAYER, AX_COMP, AX_PREC, AX_END },        /* RLCP */
    { AX_RES, AX_Y, AX_X, AX_COMP, AX_LAYER, AX_END },     /* RPCL */
    { AX_Y, AX_X, AX_COMP, AX_RES, AX_LAYER, AX_END },     /* PCRL */
    { AX_COMP, AX_Y, AX_X, AX_RES, AX_LAYER, AX_END },     /* CPRL */
};

typedef struct PacketWalk {
    PacketCtx *pc;
    const PocVolume *vol;
    const uint8_t *axes;
    int order;
    int layer, res, comp, prec, x, y;
    int step_x, step_y;                                    /* position progressions: grid pitch on the reference grid */
    int hit;                                               /* RPCL: some position of this resolution started a precinct */
    /* list mode (parallel reader): the packets are noted in order instead of being read */
    struct PktRef *list; uint32_t nlist, list_cap;
    int list_mode;
} PacketWalk;

typedef struct PktRef { uint32_t prec; uint16_t comp; uint8_t res, layer; } PktRef;

static int visit_packet(PacketWalk *pw, int comp, int r, int prec, int layer)
{
    J2kParser *ps = pw->pc->ps;
    const ResGeom *rg;
    uint8_t *done;
    if (!pw->list_mode)
        return read_packet(pw->pc, comp, r, prec, layer);
    rg = &ps->geo.tc[pw->pc->tileno * ps->ncomp + comp].res[r];
    done = &ps->layers_done[rg->lay0 + (uint32_t)prec];
    if (layer < *done)
        return 0;                                          /* as read_packet: a progression change revisits the packet */
    *done = (uint8_t)(layer + 1);
    if (pw->nlist == pw->list_cap) {
        const uint32_t nc = pw->list_cap ? 2 * pw->list_cap : 4096;
        PktRef *nl = (PktRef *)realloc(pw->list, (size_t)nc * sizeof *nl);
        if (!nl)
            return HTJ2K_ERR_ENOMEM;
        pw->list = nl;
        pw->list_cap = nc;
    }
    pw->list[pw->nlist].prec = (uint32_t)prec; pw->list[pw->nlist].comp = (uint16_t)comp;
    pw->list[pw->nlist].res = (uint8_t)r; pw->list[pw->nlist].layer = (uint8_t)layer;
    pw->nlist++;
    return 0;
}

/* position progressions: the precinct of (comp, res) that starts at reference-grid position (x, y), or -1
 * (jpeg2000dec.c:1701-1745, 1784-1821 for RPCL / PCRL; :1642-1690 for CPRL, whose tests differ) */
static int precinct_at(PacketWalk *pw)
{
    J2kParser *ps = pw->pc->ps;
    const TileHdr *t = pw->pc->tile;
    const TcGeom *tc = &ps->geo.tc[pw->pc->tileno * ps->ncomp + pw->comp];
    const ResGeom *rg = &tc->res[pw->res];
    const int down = (uint8_t)(t->cod[pw->comp].nres - 1 - pw->res);
    const int dx = ps->sub_x[pw->comp], dy = ps->sub_y[pw->comp];
    uint32_t px, py;
    if (pw->order == 4) {
        const int xc = pw->x / dx, yc = pw->y / dy;
        if (yc % ((int64_t)1 << (rg->ppy + down)) && pw->y != t->y0) return -1;
        if (xc % ((int64_t)1 << (rg->ppx + down)) && pw->x != t->x0) return -1;
        px = (uint32_t)(cdiv_pow2(xc, down) >> rg->ppx);
        py = (uint32_t)(cdiv_pow2(yc, down) >> rg->ppy);
    } else {
        const int32_t rx0 = cdiv(t->x0, (int64_t)dx << down), ry0 = cdiv(t->y0, (int64_t)dy << down);
        if (!((uint64_t)pw->y % ((uint64_t)dy << (rg->ppy + down)) == 0 ||
              (pw->y == t->y0 && ((uint64_t)((int64_t)ry0 << down) % ((uint64_t)1 << (down + rg->ppy))))))
            return -1;
        if (!((uint64_t)pw->x % ((uint64_t)dx << (rg->ppx + down)) == 0 ||
              (pw->x == t->x0 && ((uint64_t)((int64_t)rx0 << down) % ((uint64_t)1 << (down + rg->ppx))))))
            return -1;
        px = (uint32_t)(cdiv(pw->x, (int64_t)dx << down) >> rg->ppx);
        py = (uint32_t)(cdiv(pw->y, (int64_t)dy << down) >> rg->ppy);
        pw->hit = 1;
    }
    px -= (uint32_t)(cdiv_pow2(tc->ox0, down) >> rg->ppx);
    py -= (uint32_t)(cdiv_pow2(tc->oy0, down) >> rg->ppy);
    if (px >= (uint32_t)rg->npx || py >= (uint32_t)rg->npy) {
        PKT_LOG(pw->pc, LOGL_WARNING, "precinct (%u, %u) outside the %d x %d grid of its resolution\n", (unsigned)px, (unsigned)py, rg->npx, rg->npy);
        return -1;
    }
    return (int)(px + (uint32_t)rg->npx * py);
}

/* pitch of the position grid for the resolutions / components inside the current loops: the smallest
 * precinct pitch among them, as a power of two on the reference grid */
static int position_steps(PacketWalk *pw, int comp_lo, int comp_hi, int res_lo, int res_hi_or_neg, int start, int *sx, int *sy)
{
    const TileHdr *t = pw->pc->tile;
    int c, r, ex = start, ey = start;
    for (c = comp_lo; c < comp_hi; c++) {
        const CompCoding *k = &t->cod[c];
        const int hi = res_hi_or_neg < 0 ? res_lo + 1 : min32(k->nres, res_hi_or_neg);
        for (r = res_lo; r < hi; r++) {
            if (r >= k->nres)
                continue;
            ex = min32(ex, k->ppx[r] + (uint8_t)(k->nres - 1 - r));
            ey = min32(ey, k->ppy[r] + (uint8_t)(k->nres - 1 - r));
        }
    }
    *sx = ex;
    *sy = ey;
    return 0;
}

static int walk(PacketWalk *pw, int depth)
{
    PacketCtx *pc = pw->pc;
    J2kParser *ps = pc->ps;
    const TileHdr *t = pc->tile;
    const PocVolume *v = pw->vol;
    const int axis = pw->axes[depth], positional = pw->order >= 2;
    int ret, lo, hi;

    switch (axis) {
    case AX_END: {
        int prec = pw->prec;
        if (positional && (prec = precinct_at(pw)) < 0)
            return 0;
        return visit_packet(pw, pw->comp, pw->res, prec, pw->layer);
    }
    case AX_LAYER:
        if (positional) {
            /* innermost: the precinct is resolved once, then all its layers follow */
            const int prec = precinct_at(pw);
            if (prec < 0)
                return 0;
            for (pw->layer = 0; pw->layer < v->lye; pw->layer++)
                if ((ret = visit_packet(pw, pw->comp, pw->res, prec, pw->layer)) < 0)
                    return ret;
            return 0;
        }
        for (pw->layer = 0; pw->layer < v->lye; pw->layer++)
            if ((ret = walk(pw, depth + 1)) < 0)
                return ret;
        return 0;
    case AX_RES:
        if (pw->order <= 2) {
            /* outer resolution loop over all components: stops behind the deepest component (and, in RPCL,
             * behind the first resolution at which no position started a precinct) */
            int c, deepest = 0;
            for (c = v->cs; c < v->ce; c++)
                deepest = max32(deepest, t->cod[c].nres);
            for (pw->res = v->rs; pw->res < v->re; pw->res++) {
                if (pw->order == 2) {
                    pw->hit = 0;
                    position_steps(pw, v->cs, v->ce, pw->res, -1, 30, &pw->step_x, &pw->step_y);
                    pw->step_x = 1 << pw->step_x;
                    pw->step_y = 1 << pw->step_y;
                } else if (pw->res >= deepest) {
                    break;
                }
                if ((ret = walk(pw, depth + 1)) < 0)
                    return ret;
                if (pw->order == 2 && !pw->hit)
                    break;
            }
            return 0;
        }
        hi = min32(t->cod[pw->comp].nres, v->re);
        for (pw->res = v->rs; pw->res < hi; pw->res++)
            if ((ret = walk(pw, depth + 1)) < 0)
                return ret;
        return 0;
    case AX_COMP:
        for (pw->comp = v->cs; pw->comp < v->ce; pw->comp++) {
            if (pw->order == 4) {
                if (v->rs >= min32(t->cod[pw->comp].nres, v->re))
                    continue;
                position_steps(pw, pw->comp, pw->comp + 1, v->rs, v->re, 32, &pw->step_x, &pw->step_y);
                if (pw->step_x >= 31 || pw->step_y >= 31) {
                    PKT_LOG(pw->pc, LOGL_ERROR, "CPRL: precinct pitch beyond 2^30\n");
                    return HTJ2K_ERR_PATCHWELCOME;
                }
                pw->step_x = 1 << pw->step_x;
                pw->step_y = 1 << pw->step_y;
            } else if (pw->order <= 2 && pw->res >= t->cod[pw->comp].nres) {
                continue;                                   /* this component has fewer resolutions */
            }
            if ((ret = walk(pw, depth + 1)) < 0)
                return ret;
        }
        return 0;
    case AX_PREC: {
        const ResGeom *rg = &ps->geo.tc[pc->tileno * ps->ncomp + pw->comp].res[pw->res];
        hi = rg->npx * rg->npy;
        for (pw->prec = 0; pw->prec < hi; pw->prec++)
            if ((ret = walk(pw, depth + 1)) < 0)
                return ret;
        return 0;
    }
    case AX_Y:
        if (pw->order == 3) {
            position_steps(pw, v->cs, v->ce, v->rs, v->re, 32, &pw->step_x, &pw->step_y);
            if (pw->step_x >= 31 || pw->step_y >= 31) {
                PKT_LOG(pw->pc, LOGL_ERROR, "PCRL: precinct pitch beyond 2^30\n");
                return HTJ2K_ERR_PATCHWELCOME;
            }
            pw->step_x = 1 << pw->step_x;
            pw->step_y = 1 << pw->step_y;
        }
        lo = t->y0;
        for (pw->y = lo; pw->y < t->y1; pw->y = (pw->y / pw->step_y + 1) * pw->step_y)
            if ((ret = walk(pw, depth + 1)) < 0)
                return ret;
        return 0;
    case AX_X:
        for (pw->x = t->x0; pw->x < t->x1; pw->x = (pw->x / pw->step_x + 1) * pw->step_x)
            if ((ret = walk(pw, depth + 1)) < 0)
                return ret;
        return 0;
    }
    return HTJ2K_ERR_BUG;
}

/* ================================================================== the packets of a tile on several threads
 * A PLT marker segment lists the length of every packet of its tile-part (T.800 A.7.3).  The reference reads and
 * drops the list (jpeg2000dec.c:901-956); a packet header can only be found by reading the one before it, so its packet
 * reader is one chain per tile.  With the list, where every packet starts is known beforehand, and with one quality
 * layer no packet needs what another left behind (inclusion and zero-bit-plane trees, Lblock and pass counts are per
 * code-block, a code-block belongs to one precinct, a precinct has one packet per layer): the packets of the tile are
 * dealt out to threads in contiguous runs of about equal bytes.
 * The sequential reader stays the definition of what a stream means.  The parallel one reads a tile only if the list is
 * complete and plausible, the packets are where it says, every packet ends where the list says it ends, and nothing
 * came up that the sequential reader would have had something to say about; otherwise the frame is parsed again
 * from the start without it (T2_AGAIN_SEQUENTIAL) -- same plan, same messages, same errors as ever. */
#include <pthread.h>

#define PAR_MAX_THREADS 16

struct PktPool;
typedef struct PoolSlot { struct PktPool *pool; int tid; } PoolSlot;

typedef struct PktPool {
    pthread_t th[PAR_MAX_THREADS - 1];
    PoolSlot slot[PAR_MAX_THREADS - 1];
    int nth;                                               /* helpers besides the calling thread */
    pthread_mutex_t m;
    pthread_cond_t go, done;
    uint64_t gen;
    int pending, quit, active;                             /* helpers 1 .. active - 1 take part in the current run */
    void (*fn)(void *, int);
    void *arg;
    /* kept between frames */
    PktRef *list; uint32_t list_cap;
    Cur *starts; uint32_t starts_cap;
    uint32_t *lens[PAR_MAX_THREADS]; uint32_t lens_cap[PAR_MAX_THREADS];
    Contribution *con[PAR_MAX_THREADS]; uint32_t con_cap[PAR_MAX_THREADS];
} PktPool;

static void *pool_helper(void *v)
{
    PoolSlot *me = (PoolSlot *)v;
    PktPool *p = me->pool;
    uint64_t seen = 0;
    for (;;) {
        void (*fn)(void *, int);
        void *arg;
        pthread_mutex_lock(&p->m);
        while (p->gen == seen && !p->quit)
            pthread_cond_wait(&p->go, &p->m);
        if (p->quit) {
            pthread_mutex_unlock(&p->m);
            return NULL;
        }
        seen = p->gen;
        fn = p->fn; arg = p->arg;
        if (me->tid >= p->active) {                        /* not needed this time */
            pthread_mutex_unlock(&p->m);
            continue;
        }
        pthread_mutex_unlock(&p->m);
        fn(arg, me->tid);
        pthread_mutex_lock(&p->m);
        if (--p->pending == 0)
            pthread_cond_signal(&p->done);
        pthread_mutex_unlock(&p->m);
    }
}

static PktPool *pool_get_threads(J2kParser *ps, int want)     /* want: threads including the caller */
{
    PktPool *p = ps->pool;
    if (!p) {
        p = (PktPool *)calloc(1, sizeof *p);
        if (!p)
            return NULL;
        pthread_mutex_init(&p->m, NULL);
        pthread_cond_init(&p->go, NULL);
        pthread_cond_init(&p->done, NULL);
        ps->pool = p;
    }
    while (p->nth < want - 1 && p->nth < PAR_MAX_THREADS - 1) {
        p->slot[p->nth].pool = p;
        p->slot[p->nth].tid = p->nth + 1;
        if (pthread_create(&p->th[p->nth], NULL, pool_helper, &p->slot[p->nth]))
            break;
        p->nth++;
    }
    return p;
}

static void pool_run(PktPool *p, void (*fn)(void *, int), void *arg, int nworkers)
{
    pthread_mutex_lock(&p->m);
    p->fn = fn; p->arg = arg;
    p->active = nworkers;
    p->pending = nworkers - 1;
    p->gen++;
    pthread_cond_broadcast(&p->go);
    pthread_mutex_unlock(&p->m);
    fn(arg, 0);
    pthread_mutex_lock(&p->m);
    while (p->pending)
        pthread_cond_wait(&p->done, &p->m);
    pthread_mutex_unlock(&p->m);
}

void t2_pool_free(J2kParser *ps)
{
    PktPool *p = ps->pool;
    int i;
    if (!p)
        return;
    pthread_mutex_lock(&p->m);
    p->quit = 1;
    pthread_cond_broadcast(&p->go);
    pthread_mutex_unlock(&p->m);
    for (i = 0; i < p->nth; i++)
        pthread_join(p->th[i], NULL);
    pthread_mutex_destroy(&p->m);
    pthread_cond_destroy(&p->go);
    pthread_cond_destroy(&p->done);
    for (i = 0; i < PAR_MAX_THREADS; i++) {
        free(p->lens[i]);
        free(p->con[i]);
    }
    free(p->list);
    free(p->starts);
    free(p);
    ps->pool = NULL;
}

typedef struct ParJob {
    PktPool *pool;
    const PktRef *list;
    const Cur *starts;
    uint32_t first[PAR_MAX_THREADS + 1];                   /* worker w reads packets first[w] .. first[w + 1] - 1 */
    int nworkers;
    PacketCtx pc[PAR_MAX_THREADS];
    int failed[PAR_MAX_THREADS];
} ParJob;

static void par_worker(void *v, int tid)
{
    ParJob *j = (ParJob *)v;
    PacketCtx *pc;
    uint32_t i;
    if (tid >= j->nworkers)
        return;
    pc = &j->pc[tid];
    for (i = j->first[tid]; i < j->first[tid + 1]; i++) {
        const PktRef *e = &j->list[i];
        pc->at = &j->starts[i];
        if (read_packet(pc, e->comp, e->res, (int)e->prec, e->layer) < 0 || pc->anomaly || pc->end_p != j->starts[i].end) {
            j->failed[tid] = 1;
            break;
        }
    }
    j->pool->lens[tid] = pc->lens; j->pool->lens_cap[tid] = pc->lens_cap;
    j->pool->con[tid] = pc->con; j->pool->con_cap[tid] = pc->con_cap;
}

static int tile_reads_in_parallel(const J2kParser *ps, const TileHdr *t)
{
    return ps->packet_threads > 1 && !ps->seq_only && !ps->has_ppm && !t->has_ppt &&
           t->nplt >= 16 && !t->plt_bad && !t->plt_open && t->cod[0].layers == 1;
}

/* `list`: the tile's packets in order (the walker in list mode).  0, T2_AGAIN_SEQUENTIAL, or < 0 (out of memory) */
static int read_tile_parallel(J2kParser *ps, TileHdr *t, int tileno, PktPool *pool, uint32_t n)
{
    ParJob *job;
    Cur bodies[CS_MAX_TPARTS], c;
    uint64_t total = 0, acc = 0;
    uint32_t i, each, need;
    int part = 0, w, nworkers, failed = 0;

    if (n != t->nplt)
        return T2_AGAIN_SEQUENTIAL;
    if (pool->starts_cap < n) {
        Cur *ns = (Cur *)realloc(pool->starts, (size_t)n * sizeof *ns);
        if (!ns)
            return HTJ2K_ERR_ENOMEM;
        pool->starts = ns;
        pool->starts_cap = n;
    }
    /* where the packets start: the tile-part bodies one after the other, as body_stream() walks them */
    for (i = 0; i < CS_MAX_TPARTS; i++)
        bodies[i] = t->part[i].body;
    c = bodies[0];
    for (i = 0; i < n; i++) {
        while (!cur_left(&c) && part < CS_MAX_TPARTS - 1)
            c = bodies[++part];
        if (t->plt[i] > (uint32_t)cur_left(&c))
            return T2_AGAIN_SEQUENTIAL;
        pool->starts[i].p = c.p; pool->starts[i].end = c.p + t->plt[i]; pool->starts[i].base = c.base;
        c.p += t->plt[i];
        bodies[part] = c;
        total += t->plt[i];
    }

    job = (ParJob *)calloc(1, sizeof *job);
    if (!job)
        return HTJ2K_ERR_ENOMEM;
    nworkers = pool->nth + 1;
    if ((uint32_t)nworkers > n / 8)
        nworkers = (int)(n / 8) ? (int)(n / 8) : 1;
    job->pool = pool; job->list = pool->list; job->starts = pool->starts; job->nworkers = nworkers;
    /* runs of about equal bytes: a run starts at the packet whose middle is past the run's share */
    for (i = 0, w = 0; i < n && w < nworkers; i++) {
        if ((2 * acc + t->plt[i]) * (uint64_t)nworkers >= 2 * total * (uint64_t)w)
            job->first[w++] = i;
        acc += t->plt[i];
    }
    while (w <= nworkers)
        job->first[w++] = n;
    job->first[nworkers] = n;
    /* every worker chains code-block pieces out of a slice of the table of its own; the table does not move meanwhile */
    each = 4096 + 2 * (ps->geo.nblk / (uint32_t)nworkers);
    need = ps->nsegs + (uint32_t)nworkers * each;
    if (ps->segs_cap < need) {
        SegNode *ns = (SegNode *)realloc(ps->segs, (size_t)need * sizeof *ns);
        if (!ns) {
            free(job);
            return HTJ2K_ERR_ENOMEM;
        }
        ps->segs = ns;
        ps->segs_cap = need;
    }
    for (w = 0; w < nworkers; w++) {
        PacketCtx *pc = &job->pc[w];
        pc->ps = ps; pc->tile = t; pc->tileno = tileno;
        pc->quiet = 1;
        pc->arena_next = ps->nsegs + (uint32_t)w * each;
        pc->arena_limit = pc->arena_next + each;
        pc->lens = pool->lens[w]; pc->lens_cap = pool->lens_cap[w];
        pc->con = pool->con[w]; pc->con_cap = pool->con_cap[w];
    }
    pool_run(pool, par_worker, job, nworkers);
    for (w = 0; w < nworkers; w++)
        failed |= job->failed[w];
    free(job);
    if (failed)
        return T2_AGAIN_SEQUENTIAL;
    ps->nsegs = need;
    for (i = 0; i < CS_MAX_TPARTS; i++)
        t->part[i].body = bodies[i];
    ps->g = c;
    ps->parallel_tiles++;
    return 0;
}

static int walk_tile(J2kParser *ps, TileHdr *t, PacketWalk *pw)
{
    PocVolume whole;
    int i, ret = HTJ2K_ERR_BUG;
    if (t->poc.n) {
        for (i = 0; i < t->poc.n; i++) {
            PocVolume v = t->poc.v[i];
            v.lye = (uint16_t)min32(v.lye, t->cod[0].layers);
            v.ce = (uint16_t)min32(v.ce, ps->ncomp);
            pw->vol = &v;
            pw->order = v.order;
            ret = 0;
            if (v.order <= 4) {
                pw->axes = nesting[v.order];
                ret = walk(pw, 0);
            }
            if (ret < 0)
                break;
        }
    } else {
        whole.rs = 0; whole.cs = 0; whole.lye = t->cod[0].layers; whole.re = 33;
        whole.ce = (uint16_t)ps->ncomp; whole.order = t->cod[0].order;
        pw->vol = &whole;
        pw->order = whole.order;
        ret = 0;
        if (whole.order <= 4) {
            pw->axes = nesting[whole.order];
            ret = walk(pw, 0);
        }
    }
    return ret;
}

/* all packets of a tile (jpeg2000_decode_packets, jpeg2000dec.c:1835-1869) */
int t2_read_tile_packets(J2kParser *ps, int tileno)
{
    TileHdr *t = &ps->tile[tileno];
    PacketCtx pc;
    PacketWalk pw;
    int ret;

    if (ps->geo.tile_err[tileno] < 0)
        return ps->geo.tile_err[tileno];
    memset(&pc, 0, sizeof pc);
    pc.ps = ps; pc.tile = t; pc.tileno = tileno;
    memset(&pw, 0, sizeof pw);
    pw.pc = &pc;

    if (tile_reads_in_parallel(ps, t)) {
        PktPool *pool = pool_get_threads(ps, ps->packet_threads);
        if (pool && pool->nth > 0) {
            pc.quiet = 1;
            pw.list_mode = 1;
            pw.list = pool->list; pw.list_cap = pool->list_cap;
            ret = walk_tile(ps, t, &pw);
            pool->list = pw.list; pool->list_cap = pw.list_cap;
            if (ret == HTJ2K_ERR_ENOMEM)
                return ret;
            if (ret < 0 || pc.anomaly)
                return T2_AGAIN_SEQUENTIAL;                /* (the list-making has marked precincts as read) */
            ret = read_tile_parallel(ps, t, tileno, pool, pw.nlist);
            if (ret)
                return ret;
            cur_skip(&ps->g, 2);                           /* EOC, or the next tile-part's SOT: not looked at */
            return 0;
        }
    }

    pc.lens = ps->scratch_lens; pc.lens_cap = ps->scratch_lens_cap;
    pc.con = (Contribution *)ps->scratch_con; pc.con_cap = ps->scratch_con_cap;
    ret = walk_tile(ps, t, &pw);
    ps->scratch_lens = pc.lens; ps->scratch_lens_cap = pc.lens_cap;
    ps->scratch_con = pc.con; ps->scratch_con_cap = pc.con_cap;
    if (ret < 0)
        return ret;
    cur_skip(&ps->g, 2);                                   /* EOC, or the next tile-part's SOT: not looked at */
    return ret;
}

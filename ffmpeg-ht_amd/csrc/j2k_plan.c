/*
 * j2k_plan.c -- the host front-end's public face: one packet in, one J2kPlan out.
 *
 *   j2k_parse()   locate the codestream, read every header (j2k_syntax.c), lay out or re-use
 *                 the geometry tables, read the packets of every tile (j2k_tier2.c), then
 *                 fill in the block table row by row and gather the code-block bytes
 *
 * This is the work jpeg2000_decode_frame() does before it fans out over tiles
 * (libavcodec/jpeg2000dec.c:2825-2894), plus the bookkeeping tile_codeblocks() does per block
 * (:2212-2289: which decoder, M_b, where the block lands).  What the device needs from a
 * code-block is one 32-byte row (j2k_plan.h) and its bytes at a 16-byte aligned offset.
 */
#include <stdlib.h>
#include "j2k_host.h"

J2kParser *j2k_parser_new(void)
{
    J2kParser *ps = (J2kParser *)calloc(1, sizeof(J2kParser));
    if (ps)
        ps->gather_on_host = 1;
    return ps;
}

void j2k_parser_free(J2kParser *ps)
{
    if (!ps)
        return;
    t2_pool_free(ps);
    pool_destroy(&ps->frame);
    pool_destroy(&ps->geo.pool);
    free(ps->geo.sig); free(ps->geo.pb); free(ps->geo.rows); free(ps->geo.row_blk); free(ps->geo.row_aux); free(ps->geo.row_tc);
    free(ps->sig); free(ps->segs); free(ps->scratch_lens); free(ps->scratch_con); free(ps->gseg); free(ps->glit);
    free(ps);
}

void j2k_parser_set_log(J2kParser *ps, j2k_log_fn fn, void *opaque)
{
    ps->log = fn;
    ps->log_opaque = opaque;
}

void j2k_parser_set_bytes_alloc(J2kParser *ps, j2k_bytes_alloc_fn fn, void *opaque)
{
    ps->bytes_alloc = fn;
    ps->bytes_alloc_opaque = opaque;
}

/* everything that belongs to one frame goes back to zero; allocations, the geometry cache and the
 * caller's hooks stay (the reference clears its context the same way, jpeg2000dec.c:2397-2423) */
static void begin_frame(J2kParser *ps)
{
    const size_t keep_from = offsetof(J2kParser, pkt);
    const size_t keep_to = offsetof(J2kParser, segs);
    memset((uint8_t *)ps + keep_from, 0, keep_to - keep_from);
    ps->nsegs = 0;
    ps->sig_len = 0;
    memset(&ps->plan, 0, sizeof ps->plan);
    pool_rewind(&ps->frame);
}

void j2k_parser_set_gather(J2kParser *ps, int on_host) { ps->gather_on_host = on_host != 0; }
void j2k_parser_set_packet_threads(J2kParser *ps, int n) { ps->packet_threads = n < 1 ? 1 : (n > 16 ? 16 : n); }
void j2k_parser_parallel_stats(const J2kParser *ps, uint32_t *tiles, uint32_t *retries)
{
    if (tiles) *tiles = ps->parallel_tiles;
    if (retries) *retries = ps->parallel_retries;
}

/* the reference implementation of the gather: k_gather (htj2k_device.hip) does the same on the device.  The pieces
 * are in pool order, so one pass writes every byte of the pool once: pieces, and zeros in the gaps between them. */
void j2k_plan_gather(const J2kPlan *pl, uint8_t *dst)
{
    size_t pos = 0;
    uint32_t i;
    for (i = 0; i < pl->nsegs; i++) {
        const J2kSeg *g = &pl->segs[i];
        if (g->dst > pos)
            memset(dst + pos, 0, g->dst - pos);
        memcpy(dst + g->dst, ((g->flags & J2K_SEG_LIT) ? pl->lit : pl->pkt) + g->src, g->len);
        pos = (size_t)g->dst + g->len;
        if (g->flags & J2K_SEG_TERM) {
            dst[pos] = dst[pos + 1] = 0xFF;
            pos += 2;
        }
    }
    memset(dst + pos, 0, pl->nbytes + 64 - pos);
}

/* byte `off` of a block's byte string, straight from the packet */
static uint8_t block_byte(const J2kParser *ps, const BlkState *s, uint32_t off)
{
    uint32_t next = s->more;
    if (off < s->first_len)
        return ps->pkt[s->first_src + off];
    off -= s->first_len;
    while (next) {
        const SegNode *n = &ps->segs[next - 1];
        if (off < n->len)
            return ps->pkt[n->src + off];
        off -= n->len;
        next = n->next;
    }
    return 0;
}

static size_t block_region(const BlkState *s)
{
    return (s->style & CBS_HT) ? J2K_BLOCK_REGION(s->length) : J2K_P1_REGION(s->length, s->nterm);
}


static int put_seg(J2kParser *ps, uint32_t src, uint32_t dst, uint32_t len, uint32_t flags)
{
    J2kPlan *pl = &ps->plan;
    if (pl->nsegs == ps->gseg_cap) {
        const uint32_t nc = ps->gseg_cap ? ps->gseg_cap * 2 : 8192;
        J2kSeg *ns = (J2kSeg *)realloc(ps->gseg, (size_t)nc * sizeof *ns);
        if (!ns)
            return HTJ2K_ERR_ENOMEM;
        ps->gseg = ns;
        ps->gseg_cap = nc;
    }
    ps->gseg[pl->nsegs].src = src; ps->gseg[pl->nsegs].dst = dst;
    ps->gseg[pl->nsegs].len = len; ps->gseg[pl->nsegs].flags = flags;
    pl->nsegs++;
    return 0;
}

static uint8_t *put_lit(J2kParser *ps, uint32_t n, uint32_t *at)
{
    J2kPlan *pl = &ps->plan;
    if (pl->nlit + n > ps->glit_cap) {
        uint32_t nc = ps->glit_cap ? ps->glit_cap * 2 : 4096;
        uint8_t *nl;
        while (nc < pl->nlit + n)
            nc *= 2;
        nl = (uint8_t *)realloc(ps->glit, nc);
        if (!nl)
            return NULL;
        ps->glit = nl;
        ps->glit_cap = nc;
    }
    *at = pl->nlit;
    pl->nlit += n;
    return ps->glit + *at;
}

/* the gather table of one block: its pieces in pool order from pool offset `at`; for a Part-1 block also the
 * closing 0xFF 0xFF and the trailer.  Returns < 0 when memory runs out. */
static int block_segments(J2kParser *ps, const BlkState *s, uint32_t at, int part1, uint32_t aux)
{
    uint32_t next = s->more, pos = at, k = 0, lit_at = 0;
    uint8_t *tr = NULL;
    if (part1) {
        /* J2kPart1Trailer: style, bandpos, nterm, start[nterm] */
        if (!(tr = put_lit(ps, 4 + 2 * (uint32_t)s->nterm, &lit_at)))
            return HTJ2K_ERR_ENOMEM;
        tr[0] = (uint8_t)((aux >> 8) & 0x3F);
        tr[1] = (uint8_t)(aux & 0xFF);
        tr[2] = (uint8_t)(s->nterm & 0xFF);
        tr[3] = (uint8_t)(s->nterm >> 8);
    }
    if (s->flags & BS_HAS_BYTES) {
        const int term = part1 && s->first_term;
        if (put_seg(ps, s->first_src, pos, s->first_len, term ? J2K_SEG_TERM : 0) < 0)
            return HTJ2K_ERR_ENOMEM;
        pos += s->first_len + (term ? 2 : 0);
        if (term) { tr[4 + 2 * k] = (uint8_t)(pos - at); tr[5 + 2 * k] = (uint8_t)((pos - at) >> 8); k++; }
        while (next) {
            const SegNode *n = &ps->segs[next - 1];
            const int t2 = part1 && n->term;
            if (put_seg(ps, n->src, pos, n->len, t2 ? J2K_SEG_TERM : 0) < 0)
                return HTJ2K_ERR_ENOMEM;
            pos += n->len + (t2 ? 2 : 0);
            if (t2) { tr[4 + 2 * k] = (uint8_t)(pos - at); tr[5 + 2 * k] = (uint8_t)((pos - at) >> 8); k++; }
            next = n->next;
        }
    }
    if (part1) {
        if (put_seg(ps, 0, pos, 2, J2K_SEG_LIT) < 0 ||         /* lit[0..1] = 0xFF 0xFF */
            put_seg(ps, lit_at, at + (uint32_t)J2K_P1_TRAILER_OFF(s->length), 4 + 2 * (uint32_t)s->nterm, J2K_SEG_LIT) < 0)
            return HTJ2K_ERR_ENOMEM;
    }
    return 0;
}

/* the dynamic half of the block table + the gather table (+ the byte pool itself when gathering on the host) */
static int assemble_plan(J2kParser *ps)
{
    const GeomCache *g = &ps->geo;
    J2kPlan *pl = &ps->plan;
    size_t at = 0;
    uint32_t i, lit0;

    if (g->static_err)
        return g->static_err;
    pl->ntiles = g->ntiles;
    pl->ntilecomps = g->ntiles * ps->ncomp;
    pl->tilecomps = (J2kTileComp *)pool_get(&ps->frame, (size_t)pl->ntilecomps * sizeof(J2kTileComp), 0);
    pl->blocks = (J2kBlock *)pool_get(&ps->frame, (size_t)(g->nrows ? g->nrows : 1) * sizeof(J2kBlock), 0);
    pl->blk_seg0 = (uint32_t *)pool_get(&ps->frame, ((size_t)g->nrows + 1) * sizeof(uint32_t), 0);
    if (!pl->tilecomps || !pl->blocks || !pl->blk_seg0)
        return HTJ2K_ERR_ENOMEM;
    memcpy(pl->tilecomps, g->tcd, (size_t)pl->ntilecomps * sizeof(J2kTileComp));
    if (g->nrows)
        memcpy(pl->blocks, g->rows, (size_t)g->nrows * sizeof(J2kBlock));
    pl->max_scup = 2;
    pl->max_qw = 1;
    pl->nsegs = pl->nlit = 0;
    {
        uint8_t *ff = put_lit(ps, 2, &lit0);
        if (!ff)
            return HTJ2K_ERR_ENOMEM;
        ff[0] = ff[1] = 0xFF;
    }

    for (i = 0; i < g->nrows; i++) {
        const BlkState *s = &ps->blk[g->row_blk[i]];
        J2kBlock *b = &pl->blocks[i];
        J2kTileComp *tc = &pl->tilecomps[g->row_tc[i]];
        /* tile_codeblocks() picks the block decoder by the HT bit of the block's modes (jpeg2000dec.c:2264-2273);
         * a Part-1 block without bytes decodes to nothing (decode_cblk, :2008-2009), as an HT block without passes */
        const int part1 = !(s->style & CBS_HT) && s->length > 0;

        if (at + block_region(s) > 0xFFFFFF00u)
            return HTJ2K_ERR_PATCHWELCOME;                  /* pool offsets are 32-bit */
        pl->blk_seg0[i] = pl->nsegs;
        b->data_off = (uint32_t)at;
        b->flags |= s->style & J2K_CBLK_VSC;
        if (block_segments(ps, s, (uint32_t)at, part1, g->row_aux[i]) < 0)
            return HTJ2K_ERR_ENOMEM;
        at += block_region(s);
        if (part1) {
            /* the bytes as decode_cblk() sees them: segments back to back, 0xFF 0xFF behind every terminated one
             * and behind the last byte (jpeg2000dec.c:1508-1516, 2012-2013); then the trailer (J2kPart1Trailer) */
            b->flags |= J2K_BLK_PART1;
            b->npasses = s->npasses;
            b->lcup = (uint16_t)s->length;
            b->lref = s->nterm;
            b->zbp = s->nzb;
            tc->coded = 1;
            pl->have_part1 = 1;
            continue;
        }
        b->npasses = (s->style & CBS_HT) ? s->npasses : 0;
        b->zbp = s->zbp;
        if (s->lcup + s->lref <= s->length) {
            b->lcup = (uint16_t)s->lcup;
            b->lref = (uint16_t)s->lref;
        }   /* else: lengths that contradict the byte count -- the block goes to the device as empty-but-coded and is rejected there */
        if (s->npasses) {
            /* sizing figures for the kernels' LDS windows: longest MagSgn and VLC/MEL parts among the valid cleanup
             * segments (Scup sits in the last two bytes of the cleanup segment, jpeg2000htdec.c:1252-1273), widest block
             * in quads, largest refinement bitmap */
            const uint32_t quads_w = ((uint32_t)b->w + 1) >> 1;
            const int in_set = s->npasses % 3, placeholders = in_set ? s->npasses - in_set : s->npasses - 3;
            tc->coded = 1;
            if (b->lcup > pl->max_lcup) pl->max_lcup = b->lcup;
            if (b->lref > pl->max_lref) pl->max_lref = b->lref;
            if (quads_w > pl->max_qw) pl->max_qw = quads_w;
            if (b->lcup >= 2) {
                const uint32_t scup = ((uint32_t)block_byte(ps, s, b->lcup - 1u) << 4) | (block_byte(ps, s, b->lcup - 2u) & 15);
                if (scup >= 2 && scup <= b->lcup && scup <= 4079) {
                    if (scup > pl->max_scup) pl->max_scup = scup;
                    if (b->lcup - scup > pl->max_pcup) pl->max_pcup = b->lcup - scup;
                }
            }
            if (s->npasses - placeholders > 1) {
                const uint32_t words = ((uint32_t)(b->w + 2) * (uint32_t)(b->h + 2) + 31) / 32 + 1;
                if (words > pl->max_bm_words) pl->max_bm_words = words;
            }
        }
    }
    pl->blk_seg0[g->nrows] = pl->nsegs;
    pl->segs = ps->gseg;
    pl->lit = ps->glit;
    pl->pkt = ps->pkt;
    pl->pkt_size = ps->pkt_size;
    pl->nblocks = (int32_t)g->nrows;
    pl->nbytes = at;
    pl->nsamples = g->nsamples;
    pl->precision = ps->precision;
    /* 8-bit frames for up to 8 bits; 16-bit ones carry the samples at the top for the formats that are always
     * full-range, otherwise at the component's own precision (jpeg2000_decode_tile, jpeg2000dec.c:2383-2392) */
    pl->out_bytes = ps->precision <= 8 ? 1 : 2;
    pl->out_shift_precision = ps->precision <= 8 ? 8 :
        (ps->pix_fmt == HTJ2K_PIX_XYZ12 || ps->pix_fmt == HTJ2K_PIX_RGB48 || ps->pix_fmt == HTJ2K_PIX_RGBA64 ||
         ps->pix_fmt == HTJ2K_PIX_GRAY16) ? 16 : ps->precision;
    memcpy(pl->palette, ps->palette, sizeof pl->palette);
    if (ps->gather_on_host) {
        /* the device layer may hand out pinned memory for the pool */
        pl->bytes = ps->bytes_alloc ? (uint8_t *)ps->bytes_alloc(ps->bytes_alloc_opaque, at + 64)
                                    : (uint8_t *)pool_get(&ps->frame, at + 64, 0);
        if (!pl->bytes)
            return HTJ2K_ERR_ENOMEM;
        j2k_plan_gather(pl, pl->bytes);
    }
    return 0;
}

static int parse_frame(J2kParser *ps, const uint8_t *pkt, int size, const htj2k_opts *opts, int headers_only, const J2kPlan **plan);

int j2k_parse(J2kParser *ps, const uint8_t *pkt, int size, const htj2k_opts *opts, int headers_only, const J2kPlan **plan)
{
    int r = parse_frame(ps, pkt, size, opts, headers_only, plan);
    if (r == T2_AGAIN_SEQUENTIAL) {                        /* the parallel packet reader gave up: the frame again, without it */
        ps->parallel_retries++;
        ps->seq_only = 1;
        r = parse_frame(ps, pkt, size, opts, headers_only, plan);
        ps->seq_only = 0;
    }
    return r;
}

static int parse_frame(J2kParser *ps, const uint8_t *pkt, int size, const htj2k_opts *opts, int headers_only, const J2kPlan **plan)
{
    int r, tileno;

    begin_frame(ps);
    if (plan)
        *plan = NULL;
    if (opts)
        ps->opts = *opts;
    else
        ps->opts.req_pix_fmt = HTJ2K_PIX_NONE;
    ps->reduce = ps->opts.reduction_factor;
    if (ps->reduce < 0 || ps->reduce >= CS_MAX_RES)
        return HTJ2K_ERR_EINVAL;
    ps->pkt = pkt;
    ps->pkt_size = size > 0 ? size : 0;
    ps->cs = cur_make(pkt, (size_t)ps->pkt_size);
    ps->cur_tile = -1;
    ps->pix_fmt = HTJ2K_PIX_NONE;
    memset(ps->cdef, -1, sizeof ps->cdef);

    if ((r = cs_locate_codestream(ps)) < 0)
        return r;
    if ((r = cs_scan_headers(ps)) != 0)
        return r;
    if (!ps->tile || ps->pix_fmt == HTJ2K_PIX_NONE) {
        cs_log(ps, LOGL_ERROR, "no usable SIZ segment\n");
        return HTJ2K_ERR_INVALIDDATA;
    }
    cs_fill_info(ps, &ps->plan.info);
    if (headers_only) {                     /* skip_frame >= AVDISCARD_ALL: jpeg2000dec.c:2871-2874 */
        ps->plan.bytes_consumed = size;
        if (plan)
            *plan = &ps->plan;
        return 0;
    }

    ps->g = ps->cs;
    if ((r = t2_build_geometry(ps)) < 0)
        return r;
    ps->blk = (BlkState *)pool_get(&ps->frame, (size_t)(ps->geo.nblk ? ps->geo.nblk : 1) * sizeof(BlkState), 1);
    ps->nodes = (uint16_t *)pool_get(&ps->frame, (size_t)(ps->geo.nnodes ? ps->geo.nnodes : 1) * sizeof(uint16_t), 1);
    ps->layers_done = (uint8_t *)pool_get(&ps->frame, (size_t)(ps->geo.nprec ? ps->geo.nprec : 1), 1);
    if (!ps->blk || !ps->nodes || !ps->layers_done)
        return HTJ2K_ERR_ENOMEM;
    for (tileno = 0; tileno < ps->geo.ntiles; tileno++)
        if ((r = t2_read_tile_packets(ps, tileno)) != 0)
            return r;
    ps->plan.bytes_consumed = cur_pos(&ps->g);
    if ((r = assemble_plan(ps)) < 0)
        return r;
    if (plan)
        *plan = &ps->plan;
    return 0;
}

/*
 * j2k_plan.h -- the "decode plan": everything the host parser distils from one
 * codestream for the device.  Produced by the host front-end
 * (j2k_syntax.c, j2k_tier2.c, j2k_plan.c), consumed by the HIP layer (htj2k_device.hip).
 * (The CPU oracle has its own parser and its own frozen copy of these types,
 * oracle/j2k_oracle_plan.h; tests/native/plan_diff.c compares the two.)
 *
 * Reference types this is distilled from: Jpeg2000Cblk (libavcodec/jpeg2000.h:183-205),
 * Jpeg2000Band (:217-223), Jpeg2000Component (:233-241), DWTContext
 * (libavcodec/jpeg2000dwt.h:44-52), Jpeg2000DecoderContext (jpeg2000dec.h:73-123).
 */
#ifndef J2K_PLAN_H
#define J2K_PLAN_H

#include <stddef.h>
#include <stdint.h>
#include "../../include/htj2k_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

#define J2K_MAX_COMPS   4
#define J2K_MAX_DWTLEV  32      /* FF_DWT_MAX_DECLVLS, jpeg2000dwt.h:30 */

/* enum DWTType, jpeg2000dwt.h:36-41 (COD transform byte: 0 = 9/7, 1 = 5/3) */
#define J2K_DWT97      0
#define J2K_DWT53      1
#define J2K_DWT97_INT  2

#define J2K_CBLK_VSC   0x08     /* JPEG2000_CBLK_VSC, jpeg2000.h:113 */
#define J2K_BLK_PART1  0x04     /* J2kBlock.flags: a Part-1 (MQ-coded) block, decode_cblk() instead of the HT decoder
                                 * (jpeg2000dec.c:2264-2273); see J2kPart1Trailer for what the fields then mean */

/* One codeblock: 32 bytes, read by one wavefront.  Blocks with npasses == 0 are
 * kept in the table so that the device zero-fills their window (the reference
 * gets that from av_calloc of the plane, jpeg2000.c:499-511). */
typedef struct J2kBlock {
    uint32_t data_off;   /* offset of Dcup in J2kPlan.bytes; Dref follows at +lcup.  >= 8 bytes of pad follow */
    uint32_t plane_off;  /* sample offset of the block's (x,y) inside the frame coefficient buffer */
    uint16_t lcup;       /* cblk->pass_lengths[0]  (jpeg2000htdec.c:1249) */
    uint16_t lref;       /* cblk->pass_lengths[1]  (jpeg2000htdec.c:1250) */
    uint16_t w, h;       /* cblk->coord spans (jpeg2000dec.c:2266-2267) */
    uint16_t stride;     /* row stride of the tile-component plane, in samples */
    uint8_t  npasses;    /* cblk->npasses */
    uint8_t  zbp;        /* cblk->zbp from the zero-bit-plane tag tree (jpeg2000dec.c:1185,1194) */
    uint8_t  M_b;        /* expn[subband] + nguardbits - 1 (jpeg2000dec.c:2238) */
    uint8_t  flags;      /* bit 3: J2K_CBLK_VSC; bits 0-1: transform of the component (J2K_DWT*) */
    uint8_t  roi_shift;  /* comp->roi_shift (jpeg2000dec.c:2268) */
    uint8_t  tcomp;      /* tile-component index modulo 256: informational, nothing indexes with it */
    float    f_step;     /* band->f_stepsize (jpeg2000.c:243-264); for 9/7-int: the rounded int scale as float bits unused */
    int32_t  i_step;     /* band->i_stepsize (jpeg2000.c:271); for J2K_DWT97_INT: the (int)(fscale+0.5) scale of jpeg2000dec.c:2164-2168 */
} J2kBlock;

/* Part-1 blocks (flags & J2K_BLK_PART1) reuse the descriptor:
 *   lcup    = cblk->length: all segments back to back, 0xFF 0xFF after every terminated one
 *             (jpeg2000dec.c:1508-1516); two more bytes of 0xFF follow (decode_cblk writes them, :2012-2013)
 *   lref    = cblk->nb_terminations
 *   zbp     = cblk->nonzerobits (jpeg2000dec.c:1186-1193)
 *   npasses = cblk->npasses
 * and a trailer sits in the byte pool at data_off + J2K_P1_TRAILER_OFF(lcup). */
typedef struct J2kPart1Trailer {
    uint8_t  style;      /* codsty->cblk_style: BYPASS 0x01, RESET 0x02, TERMALL 0x04, VSC 0x08, PREDTERM 0x10, SEGSYM 0x20 */
    uint8_t  bandpos;    /* bandno + (reslevelno > 0): 0 LL, 1 HL, 2 LH, 3 HH (jpeg2000dec.c:2240) */
    uint16_t nterm;      /* = lref */
    uint16_t start[2];   /* cblk->data_start[1 .. nterm] (really nterm entries) */
} J2kPart1Trailer;
#define J2K_P1_TRAILER_OFF(len)      ((((size_t)(len)) + 2 + 3) & ~(size_t)3)
#define J2K_P1_REGION(len, nterm)    ((J2K_P1_TRAILER_OFF(len) + 4 + 2 * (size_t)(nterm) + J2K_BLOCK_PAD + 15) & ~(size_t)15)

/* One tile-component = one coefficient plane + its DWT geometry */
typedef struct J2kTileComp {
    int32_t comp, tile;
    int32_t x0, x1, y0, y1;      /* comp->coord (after reduction_factor), jpeg2000dec.c:1047-1050 */
    int32_t w, h;                /* plane dims = x1-x0, y1-y0 */
    int32_t transform;           /* J2K_DWT* */
    int32_t ndeclevels;          /* nreslevels2decode - 1 (jpeg2000.c:483-485) */
    int32_t linelen[J2K_MAX_DWTLEV][2]; /* ff_jpeg2000_dwt_init, jpeg2000dwt.c:554-560 */
    uint8_t mod[J2K_MAX_DWTLEV][2];
    int32_t coded;               /* any contributing block: run the IDWT (jpeg2000dec.c:2224,2276,2294) */
    uint32_t plane_off;          /* sample offset of the plane in the frame coefficient buffer */
    /* write_frame_8/16 placement (jpeg2000dec.c:2312-2358) */
    int32_t cbps;                /* s->cbps[compno] */
    int32_t out_plane;           /* picture->data[] index */
    int32_t out_x, out_y;        /* first pixel written in that plane */
    int32_t out_w, out_h;        /* pixels written per row / rows: (w - x), (h - y) of the macro */
    int32_t pix_step;            /* pixelsize: samples between consecutive pixels of this component */
    int32_t pix_off;             /* compno * !planar */
    int32_t mct;                 /* 1 if this tile's codsty[0].mct and comp < 3 and mct_decode() accepts it */
} J2kTileComp;

/* One piece of a code-block's byte string: `len` bytes at offset `src` of the packet (or of J2kPlan.lit with
 * J2K_SEG_LIT) go to offset `dst` of the byte pool; with J2K_SEG_TERM the bytes 0xFF 0xFF follow them (a terminated
 * Part-1 segment, jpeg2000dec.c:1510-1516).  The pieces of block i are segs[blk_seg0[i] .. blk_seg0[i + 1]), in pool
 * order.  With this table the byte pool can be put together anywhere -- by j2k_plan_gather() on the host, or by
 * k_gather on the device from the uploaded packet (the host then never touches the code-block bytes). */
typedef struct J2kSeg { uint32_t src, dst, len, flags; } J2kSeg;
#define J2K_SEG_TERM 1
#define J2K_SEG_LIT  2

typedef struct J2kPlan {
    htj2k_info info;
    int32_t bytes_consumed;      /* bytestream2_tell at return (jpeg2000dec.c:2903) */
    int32_t precision;           /* s->precision */
    int32_t out_bytes;           /* 1: write_frame_8, 2: write_frame_16 (jpeg2000dec.c:2383-2392) */
    int32_t out_shift_precision; /* "precision" argument of write_frame (8, 16 or s->precision) */
    int32_t ntiles;
    int32_t ntilecomps;
    J2kTileComp *tilecomps;
    int32_t nblocks;
    J2kBlock *blocks;
    uint8_t *bytes;              /* concatenated codeblock bytes; NULL when the parser does not gather (j2k_parser_set_gather) */
    size_t   nbytes;
    const uint8_t *pkt;          /* the packet the plan was made from (borrowed) */
    int32_t  pkt_size;
    J2kSeg  *segs;               /* gather table: how `bytes` is put together from `pkt` and `lit` */
    uint32_t nsegs;
    uint32_t *blk_seg0;          /* nblocks + 1 entries */
    uint8_t *lit;                /* bytes that are not in the packet: 0xFF 0xFF, Part-1 trailers */
    uint32_t nlit;
    size_t   nsamples;           /* total samples of all planes (frame coefficient buffer size) */
    uint32_t max_lcup, max_lref; /* sizing of the per-wave LDS windows */
    /* over the coded blocks with a valid cleanup segment: longest MagSgn (Pcup) and VLC/MEL (Scup) parts,
     * widest block in quads, largest refinement bitmap ((w+2)*(h+2) bits in words, blocks with > 1 pass).
     * Computed while the block bytes are hot in cache, so that the device layer never has to touch the
     * byte pool again (jpeg2000htdec.c:1252-1273 for the Scup rules) */
    uint32_t max_pcup, max_scup, max_qw, max_bm_words;
    int32_t  have_part1;         /* some blocks are Part-1 (MQ) coded: J2K_BLK_PART1 */
    uint32_t palette[256];
} J2kPlan;

/* bytes kept free behind every block's Dcup||Dref in J2kPlan.bytes (regions are 16-byte aligned): the device
 * kernels read whole dwords up to the end of a segment, and the un-stuffed copies of a block's VLC/MEL and
 * SigProp/MagRef streams (k_ht_unstuff) share one region of the same size: (Scup + Lref) / 4 + 7 words at most */
#define J2K_BLOCK_PAD 32
#define J2K_BLOCK_REGION(len) ((((size_t)(len)) + J2K_BLOCK_PAD + 15) & ~(size_t)15)

typedef struct J2kParser J2kParser;   /* reusable arena; not thread-safe */

typedef void (*j2k_log_fn)(void *opaque, int level, const char *msg);

J2kParser *j2k_parser_new(void);
void       j2k_parser_free(J2kParser *p);
void       j2k_parser_set_log(J2kParser *p, j2k_log_fn fn, void *opaque);
/* where J2kPlan.bytes of the next plans lives: fn(opaque, n) returns a buffer of >= n bytes that stays
 * valid until its next call (the device layer hands out pinned host memory); NULL fn = the parser's arena */
typedef void *(*j2k_bytes_alloc_fn)(void *opaque, size_t n);
void       j2k_parser_set_bytes_alloc(J2kParser *p, j2k_bytes_alloc_fn fn, void *opaque);

/* on (default): j2k_parse fills J2kPlan.bytes; off: only the gather table is made and no code-block byte is read
 * except the two that hold Scup */
void       j2k_parser_set_gather(J2kParser *p, int on_host);
/* n > 1: the packets of a tile are read by n threads where the stream allows it -- a PLT packet-length list that covers
 * the tile, no PPM / PPT, one quality layer; everything else, and every frame in which anything at all disagrees with the
 * list, is read by the calling thread as before (same plan, same messages).  For callers that decode one frame at a time;
 * batches are better served by one frame per thread (htj2k_job_parse_batch). */
void       j2k_parser_set_packet_threads(J2kParser *p, int n);
/* tiles the parallel reader has read / frames it gave up on, since the parser was made */
void       j2k_parser_parallel_stats(const J2kParser *p, uint32_t *tiles, uint32_t *retries);
/* the byte pool of a plan, nbytes + 64 bytes, written to dst (what the parser does itself when gathering is on) */
void       j2k_plan_gather(const J2kPlan *plan, uint8_t *dst);

/* Full parse: markers + all packets.  `headers_only` stops after the main
 * header (info valid, no blocks), like skip_frame >= AVDISCARD_ALL
 * (jpeg2000dec.c:2871-2874).  The plan is owned by the parser and valid until
 * the next call.  Returns 0 or a negative HTJ2K_ERR_*. */
int j2k_parse(J2kParser *p, const uint8_t *pkt, int size, const htj2k_opts *opts,
              int headers_only, const J2kPlan **plan);

/* pix-fmt facts shared by parser, device layer and oracle */
typedef struct J2kPixDesc {
    const char *name;
    uint8_t nb_components;
    uint8_t log2_chroma_w, log2_chroma_h;
    uint8_t planar;     /* AV_PIX_FMT_FLAG_PLANAR */
    uint8_t pal;        /* AV_PIX_FMT_FLAG_PAL */
    uint8_t depth[4];
    uint8_t nplanes;
    uint8_t bytes;      /* bytes per sample */
} J2kPixDesc;
const J2kPixDesc *j2k_pix_desc(int pix_fmt);

#ifdef __cplusplus
}
#endif
#endif

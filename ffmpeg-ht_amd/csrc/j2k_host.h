/*
 * j2k_host.h -- internals of the host front-end (not installed): what the three
 * translation units of the codestream parser share.
 *
 *   j2k_syntax.c  marker segments (table-driven), JP2 boxes, output format
 *   j2k_tier2.c   closed-form geometry, packet sequencing, packet headers
 *   j2k_plan.c    J2kPlan assembly, byte gathering, public entry points
 *
 * Data model.  The reference builds a pointer tree Tile -> Component -> ResLevel -> Band ->
 * Precinct -> Cblk per frame (libavcodec/jpeg2000.c:469-577, freed jpeg2000dec.c:2397-2423).
 * Here a frame is four flat tables:
 *
 *   Geometry (static for a given set of header bytes, kept across frames -- GeomCache):
 *     TcGeom[tile * ncomp + comp]       resolution rectangles, band rectangles, step sizes
 *     PrecBand[...]                     per (resolution, band, precinct): block grid, first block,
 *                                       first tag-tree node
 *     J2kBlock skeleton rows            position / size / M_b / step of every block that
 *                                       tile_codeblocks() (jpeg2000dec.c:2212-2299) would visit
 *   Per-frame state (zeroed per frame):
 *     BlkState[...]                     one per code-block in packet order
 *     tag-tree nodes, layers_done[...]  packet-header decoding state
 *
 * All tables are indexed, not linked; code-blocks are numbered in the order
 * tile_codeblocks() walks them, so the block table of the plan is a linear pass.
 */
#ifndef J2K_HOST_H
#define J2K_HOST_H

#include <limits.h>
#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include "j2k_plan.h"

#define CS_MAX_RES      34          /* resolutions per component: NL + 1 <= 33 is accepted (jpeg2000.h:82) */
#define CS_MAX_BANDS    (3 * 33)    /* sub-bands a QCD/QCC can describe */
#define CS_MAX_PASSES   100         /* coding passes per block (jpeg2000.h:84) */
#define CS_MAX_POC      32          /* progression changes per tile (jpeg2000dec.h:41) */
#define CS_MAX_TPARTS   32          /* tile-parts per tile (jpeg2000dec.h:63) */

/* Scod / Scoc bits and code-block style bits (T.800 Table A.13, A.19; T.814 A.3) */
#define SCOD_PRECINCTS  0x01
#define SCOD_SOP        0x02
#define SCOD_EPH        0x04
#define CBS_BYPASS      0x01
#define CBS_TERMALL     0x04
#define CBS_HT          0x40        /* HT code-blocks possible */
#define CBS_HT_MIXED    0x80        /* HT and Part-1 blocks may both occur */

enum { LOGL_ERROR = 16, LOGL_WARNING = 24, LOGL_INFO = 32, LOGL_DEBUG = 48 };

/* ------------------------------------------------------------------ byte cursor
 * `base` is what positions are reported against; reads past `end` yield zero and leave
 * the cursor at `end` (the reference's checked bytestream2 readers behave that way and
 * several of its decisions depend on it, libavcodec/bytestream.h:150-206). */
typedef struct Cur { const uint8_t *p, *end, *base; } Cur;

static inline Cur      cur_make(const uint8_t *b, size_t n) { Cur c = { b, b + n, b }; return c; }
static inline int      cur_left(const Cur *c) { return (int)(c->end - c->p); }
static inline int      cur_pos(const Cur *c) { return (int)(c->p - c->base); }
static inline void     cur_skip(Cur *c, uint32_t n) { c->p += (n < (uint32_t)cur_left(c)) ? n : (uint32_t)cur_left(c); }
static inline void     cur_goto(Cur *c, int64_t off)
{
    const int64_t sz = c->end - c->base;
    c->p = c->base + (off < 0 ? 0 : (off > sz ? sz : off));
}
static inline uint32_t ld_be16(const uint8_t *p) { return ((uint32_t)p[0] << 8) | p[1]; }
static inline uint32_t ld_be32(const uint8_t *p) { return (ld_be16(p) << 16) | ld_be16(p + 2); }
static inline uint32_t cur_u8(Cur *c)  { if (cur_left(c) < 1) { c->p = c->end; return 0; } return *c->p++; }
static inline uint32_t cur_u16(Cur *c) { if (cur_left(c) < 2) { c->p = c->end; return 0; } c->p += 2; return ld_be16(c->p - 2); }
static inline uint32_t cur_u32(Cur *c) { if (cur_left(c) < 4) { c->p = c->end; return 0; } c->p += 4; return ld_be32(c->p - 4); }
static inline uint32_t cur_peek16(const Cur *c) { return cur_left(c) < 2 ? 0 : ld_be16(c->p); }
static inline uint32_t cur_peek32(const Cur *c) { return cur_left(c) < 4 ? 0 : ld_be32(c->p); }

/* ------------------------------------------------------------------ bump allocator */
typedef struct Slab { struct Slab *next; size_t cap, used; } Slab;
typedef struct Pool { Slab *first, *cur; } Pool;
void *pool_get(Pool *a, size_t n, int zeroed);     /* 16-byte aligned; NULL when the system is out of memory */
void  pool_rewind(Pool *a);
void  pool_destroy(Pool *a);

/* ------------------------------------------------------------------ coding parameters
 * One resolved set per component: what COD/COC (jpeg2000dec.c:492-641) and QCD/QCC (:676-758)
 * leave behind after the inheritance rules (main header -> first tile-part header,
 * COC/QCC beat COD/QCD of the same header). */
typedef struct CompCoding {
    uint8_t nres;               /* NL + 1 */
    uint8_t nres_dec;           /* resolutions left after `reduction_factor` */
    uint8_t cbw, cbh;           /* log2 nominal code-block size */
    uint8_t cb_style;           /* SPcod/SPcoc byte 8: mode switches, bits 6-7 the HT kind */
    uint8_t wavelet;            /* J2K_DWT53 / J2K_DWT97 / J2K_DWT97_INT */
    uint8_t scod;               /* SCOD_* */
    uint8_t order;              /* progression order of the COD it inherited from */
    uint8_t layers;             /* the reference keeps the 16-bit SGcod count in a uint8_t (jpeg2000.h:152) */
    uint8_t mct;
    uint8_t defined;
    uint8_t ppx[CS_MAX_RES], ppy[CS_MAX_RES];   /* log2 precinct size per resolution (15 = maximal) */
} CompCoding;

typedef struct CompQuant {
    uint8_t  style;             /* 0 none, 1 scalar derived, 2 scalar expounded */
    uint8_t  guard;
    uint8_t  expn[CS_MAX_BANDS];
    uint16_t mant[CS_MAX_BANDS];
} CompQuant;

typedef struct PocVolume { uint16_t lye, cs, ce; uint8_t rs, re, order; } PocVolume;
typedef struct PocList { PocVolume v[CS_MAX_POC]; int n; int inherited; } PocList;

#define SEEN_COC 1
#define SEEN_QCC 2
#define TILE_HAS_DEFAULTS 1     /* a first tile-part header copied the main header's parameters */
#define TILE_OWN_PARAMS   2     /* COD / COC / QCD / QCC / RGN segments in the tile's own headers */

typedef struct TilePartSpan { const uint8_t *limit; Cur hdr, body; } TilePartSpan;

typedef struct TileHdr {
    CompCoding cod[J2K_MAX_COMPS];
    CompQuant  q[J2K_MAX_COMPS];
    uint8_t    seen[J2K_MAX_COMPS];
    uint8_t    roi[J2K_MAX_COMPS];
    PocList    poc;
    TilePartSpan part[CS_MAX_TPARTS];
    uint16_t   cur_part;                   /* TPsot of the tile-part header being read */
    uint8_t    has_ppt;
    uint8_t    own_params;                 /* TILE_* */
    uint8_t   *ppt; int ppt_size; Cur ppt_cur;
    int32_t    x0, x1, y0, y1;             /* on the reference grid */
    /* packet lengths of the tile's PLT segments, in the order they appear (the reference reads and drops them,
     * jpeg2000dec.c:901-956; here they let several threads read the packets of a tile, j2k_tier2.c) */
    uint32_t  *plt; uint32_t nplt, plt_cap;
    uint32_t   plt_acc;                    /* the length being assembled from 7-bit pieces */
    uint8_t    plt_open, plt_bad;          /* a length is unfinished at the end of a segment / the list is unusable */
} TileHdr;

/* ------------------------------------------------------------------ geometry tables */
typedef struct BandGeom {
    int32_t x0, y0, x1, y1;     /* sub-band rectangle (band->coord, jpeg2000.c:413-441) */
    uint8_t cbw, cbh;           /* log2 code-block size after the precinct limit */
    uint8_t bppx, bppy;         /* log2 precinct size in band coordinates */
    float   fstep;              /* band->f_stepsize */
    int32_t istep;              /* band->i_stepsize */
} BandGeom;

typedef struct PrecBand {       /* one precinct of one band */
    int32_t  ncw, nch;          /* code-block grid */
    uint32_t blk0;              /* first BlkState */
    uint32_t node0;             /* first tag-tree node: inclusion tree, then zero-bit-plane tree */
    uint32_t ntree;             /* nodes of one tree */
} PrecBand;

typedef struct ResGeom {
    int32_t  x0, y0, x1, y1;    /* resolution rectangle */
    int32_t  npx, npy;          /* precinct grid */
    uint8_t  ppx, ppy;
    uint8_t  nbands;
    BandGeom band[3];
    uint32_t pb0;               /* PrecBand index of (band 0, precinct 0); band b, precinct p -> pb0 + b * np + p */
    uint32_t lay0;              /* layers_done index of precinct 0 */
} ResGeom;

typedef struct TcGeom {
    int32_t ox0, oy0, ox1, oy1; /* tile-component rectangle before `reduction_factor` (comp->coord_o) */
    int32_t x0, y0, x1, y1;     /* after it (comp->coord) */
    ResGeom *res;               /* nres entries */
} TcGeom;

/* dynamic state of one code-block while the packets are read */
typedef struct BlkState {
    uint32_t first_src;         /* offset in the packet of the first contribution */
    uint32_t first_len;
    uint32_t more;              /* index + 1 of the second contribution in the frame's SegNode list, 0 = none */
    uint32_t last;              /* index + 1 of the last node */
    uint32_t length;            /* cblk->length: all bytes so far, plus two per terminated Part-1 segment */
    uint32_t lcup, lref;        /* HT: cleanup / refinement segment bytes (cblk->pass_lengths) */
    uint16_t nterm;             /* Part-1: terminated segments */
    uint8_t  npasses;
    uint8_t  lblock;
    uint8_t  flags;             /* BS_* */
    uint8_t  style;             /* cblk->modes */
    uint8_t  zbp;               /* zero-bit-plane tag-tree value */
    uint8_t  nzb;               /* cblk->nonzerobits */
    uint8_t  first_term;        /* the first contribution ends a terminated segment */
} BlkState;
#define BS_INCLUDED   1
#define BS_PLACEHOLD  2         /* an HT block that has not shown a cleanup segment yet (HT_PLHD_ON) */
#define BS_HAS_BYTES  4         /* first_src / first_len are set */

typedef struct SegNode { uint32_t src, len; uint32_t next; uint32_t term; } SegNode;

typedef struct GeomCache {
    Pool     pool;
    uint8_t *sig; size_t sig_len, sig_cap;     /* the header bytes + options the tables were built from */
    int      valid;
    int      ntiles, ncomp;
    TcGeom  *tc;                               /* ntiles * ncomp */
    PrecBand *pb; uint32_t npb;
    uint32_t nblk;                             /* BlkState entries */
    uint32_t nnodes;                           /* tag-tree nodes */
    uint32_t nprec;                            /* layers_done entries */
    J2kBlock *rows; uint32_t *row_blk; uint32_t nrows;   /* skeleton of the plan's block table + BlkState of each row */
    uint16_t *row_aux;                         /* per row: code-block style of the component << 8 | band orientation */
    uint32_t *row_tc;                          /* per row: index of its tile-component */
    J2kTileComp *tcd;                          /* skeleton of the plan's tile-component table */
    size_t   nsamples;
    int     *tile_err;                         /* per tile: what setting it up fails with (init_tile), 0 = fine */
    int32_t *tile_rect;                        /* per tile: x0, x1, y0, y1 on the reference grid */
    int      first_bad_tile;                   /* -1: every tile is fine and the rows are laid out */
    int      static_err;                       /* what assembling the plan will fail with (0 = nothing) */
} GeomCache;

struct J2kParser {
    Pool frame;                                /* per-frame tables */
    GeomCache geo;
    j2k_log_fn log; void *log_opaque;
    j2k_bytes_alloc_fn bytes_alloc; void *bytes_alloc_opaque;
    htj2k_opts opts;
    int gather_on_host;                        /* 1 (default): J2kPlan.bytes is filled by the parser */
    int packet_threads;                        /* > 1: tiles with a complete PLT list and one layer have their packets read by
                                                * this many threads (t2_read_tile_packets) */
    struct PktPool *pool;                      /* ... which live here between frames */
    int seq_only;                              /* set while a frame is parsed again after the parallel reader gave up */
    uint32_t parallel_tiles, parallel_retries; /* statistics: tiles read in parallel, frames parsed again */

    /* ---- codestream-level state of the frame being parsed ---- */
    const uint8_t *pkt; int pkt_size;
    Cur cs;                                    /* the marker scanner's position */
    int32_t xsiz, ysiz, xosiz, yosiz, xtsiz, ytsiz, xtosiz, ytosiz;
    int ncomp;
    uint8_t depth[J2K_MAX_COMPS], is_signed[J2K_MAX_COMPS];
    int sub_x[J2K_MAX_COMPS], sub_y[J2K_MAX_COMPS];
    int precision;
    int rsiz;
    uint32_t tiles_x, tiles_y;
    int have_siz;
    int reduce;                                /* opts.reduction_factor */
    /* CAP (Part 15) */
    uint8_t is_ht, ht_kind, ht_rgn_ok, ht_hetero, ht_irrev, ht_magbits;
    /* main-header defaults */
    CompCoding cod[J2K_MAX_COMPS];
    CompQuant  q[J2K_MAX_COMPS];
    uint8_t    seen[J2K_MAX_COMPS];
    uint8_t    roi[J2K_MAX_COMPS];
    PocList    poc;
    uint8_t    has_ppm; uint8_t *ppm; int ppm_size; Cur ppm_cur;
    int        cur_tile;                       /* Isot of the tile-part header being read, -1 in the main header */
    int        in_tile_hdr;
    TileHdr   *tile;
    /* JP2 wrapper */
    int colour_space; int palettised; uint32_t palette[256]; int cdef[J2K_MAX_COMPS];
    int sar_num, sar_den;
    /* output format */
    int pix_fmt, lossless, out_w, out_h;
    /* ---- packet reading ---- */
    Cur g;                                     /* the stream the last packet left off in (bytes_consumed, jpeg2000dec.c:2903) */
    BlkState *blk; uint16_t *nodes; uint8_t *layers_done;
    SegNode *segs; uint32_t nsegs, segs_cap;
    uint32_t *scratch_lens; uint32_t scratch_lens_cap;      /* per-packet lists, kept between packets */
    void *scratch_con; uint32_t scratch_con_cap;
    J2kSeg *gseg; uint32_t gseg_cap;                        /* the plan's gather table and literal bytes */
    uint8_t *glit; uint32_t glit_cap;
    /* geometry signature under construction */
    uint8_t *sig; size_t sig_len, sig_cap;
    J2kPlan plan;
};

/* j2k_syntax.c */
void cs_log(J2kParser *ps, int level, const char *fmt, ...) __attribute__((format(printf, 3, 4)));
int  cs_locate_codestream(J2kParser *ps);      /* JP2 boxes or raw: leaves ps->cs behind SOC */
int  cs_scan_headers(J2kParser *ps);           /* main header, every tile-part header; bodies are only located */
void cs_fill_info(const J2kParser *ps, htj2k_info *info);
int  cs_sig_append(J2kParser *ps, const void *data, size_t n);
int  cs_picture_size_ok(uint32_t w, uint32_t h, int64_t max_pixels);
static inline int64_t pixel_budget(const J2kParser *ps) { return ps->opts.max_pixels > 0 ? ps->opts.max_pixels : INT_MAX; }

/* j2k_tier2.c */
int  t2_build_geometry(J2kParser *ps);         /* fills ps->geo from the resolved tile headers (or finds it cached) */
int  t2_read_tile_packets(J2kParser *ps, int tileno);   /* < 0: error; T2_AGAIN_SEQUENTIAL: parse the frame again with seq_only */
#define T2_AGAIN_SEQUENTIAL 0x7fff0001
void t2_pool_free(J2kParser *ps);

/* small arithmetic shared by all */
static inline int32_t cdiv_pow2(int32_t a, int s) { return (int32_t)-((-(int64_t)a) >> s); }
static inline int32_t cdiv(int32_t a, int64_t b) { return (int32_t)((a + b - 1) / b); }
static inline int     ilog2u(uint32_t v) { return v ? 31 - __builtin_clz(v) : 0; }
static inline int32_t min32(int32_t a, int32_t b) { return a < b ? a : b; }
static inline int32_t max32(int32_t a, int32_t b) { return a > b ? a : b; }

#endif

/* plan_diff.c -- test harness: the product's host parser (ffmpeg-ht_amd/csrc/j2k_syntax.c, j2k_tier2.c,
 * j2k_plan.c, compiled in) against the oracle's parser (oracle/libj2k_oracle.so, loaded with dlopen), on the
 * same packets: same return code, and for accepted packets the same plan -- stream facts, tile-component
 * table, block table, byte pool, LDS sizing figures.
 *
 *   plan_diff [-m N] [-s SEED] [-t THREADS] [-v] ORACLE_SO FILE...
 *      -t: the product's parsers read the packets of tiles with a usable PLT list on THREADS threads
 *          (j2k_parser_set_packet_threads); the plans must not change, and "parallel tiles" says how many tiles took that way
 *      every FILE is parsed as it is with reduction_factor 0..2 and bitexact 0/1, then N mutations of it
 *      (truncation, byte and bit damage, header-only damage) with random options.
 *   A FILE named *.list holds one path per line.
 * Prints "plan_diff: <parses> parses, <accepted> accepted, <differences> differences"; exit status 1 when
 * anything differed. */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "j2k_plan.h"

#define J2kBlock OrcBlock
#define J2kPart1Trailer OrcPart1Trailer
#define J2kTileComp OrcTileComp
#define J2kPlan OrcPlan
#define J2kParser OrcParser
#define J2kPixDesc OrcPixDesc
#define j2k_log_fn orc_log_fn
#define j2k_bytes_alloc_fn orc_bytes_alloc_fn
#include "../../oracle/j2k_oracle_plan.h"
#undef J2kBlock
#undef J2kPart1Trailer
#undef J2kTileComp
#undef J2kPlan
#undef J2kParser
#undef J2kPixDesc
#undef j2k_log_fn
#undef j2k_bytes_alloc_fn

static OrcParser *(*o_new)(void);
static void (*o_free)(OrcParser *);
static int (*o_parse)(OrcParser *, const uint8_t *, int, const htj2k_opts *, int, const OrcPlan **);

static uint32_t rs = 12345;
static uint32_t rnd(void) { rs = rs * 1664525u + 1013904223u; return rs >> 8; }
static int verbose;
static long n_parse, n_ok, n_diff;
static J2kParser *lazy;

#define DIFF(...) do { if (n_diff < 20 || verbose) { printf("DIFF %s: ", what); printf(__VA_ARGS__); printf("\n"); } n_diff++; return; } while (0)

static void compare(const char *what, J2kParser *mine, OrcParser *theirs, const uint8_t *pkt, int size, const htj2k_opts *o, int headers_only)
{
    const J2kPlan *a = NULL;
    const OrcPlan *b = NULL;
    int ra = j2k_parse(mine, pkt, size, o, headers_only, &a);
    int rb = o_parse(theirs, pkt, size, o, headers_only, &b);
    int i;
    n_parse++;
    if (ra != rb)
        DIFF("return %d, oracle %d (size %d, lowres %d, bitexact %d)", ra, rb, size, o->reduction_factor, o->bitexact);
    if (ra < 0)
        return;
    n_ok++;
    if (memcmp(&a->info, &b->info, sizeof a->info))
        DIFF("stream info differs (%dx%d fmt %d / %dx%d fmt %d)", a->info.width, a->info.height, a->info.pix_fmt, b->info.width, b->info.height, b->info.pix_fmt);
    if (a->bytes_consumed != b->bytes_consumed)
        DIFF("bytes_consumed %d / %d", a->bytes_consumed, b->bytes_consumed);
    if (headers_only)
        return;
    if (a->precision != b->precision || a->out_bytes != b->out_bytes || a->out_shift_precision != b->out_shift_precision)
        DIFF("output precision");
    if (a->ntiles != b->ntiles || a->ntilecomps != b->ntilecomps)
        DIFF("tile counts %d %d / %d %d", a->ntiles, a->ntilecomps, b->ntiles, b->ntilecomps);
    if (sizeof(J2kTileComp) != sizeof(OrcTileComp) || sizeof(J2kBlock) != sizeof(OrcBlock))
        DIFF("descriptor sizes");
    for (i = 0; i < a->ntilecomps; i++)
        if (memcmp(&a->tilecomps[i], &b->tilecomps[i], sizeof(J2kTileComp))) {
            const J2kTileComp *x = &a->tilecomps[i]; const OrcTileComp *y = &b->tilecomps[i];
            DIFF("tile-component %d: %dx%d lev %d coded %d off %u out %d,%d %dx%d mct %d / %dx%d lev %d coded %d off %u out %d,%d %dx%d mct %d", i,
                 x->w, x->h, x->ndeclevels, x->coded, x->plane_off, x->out_x, x->out_y, x->out_w, x->out_h, x->mct,
                 y->w, y->h, y->ndeclevels, y->coded, y->plane_off, y->out_x, y->out_y, y->out_w, y->out_h, y->mct);
        }
    if (a->nblocks != b->nblocks)
        DIFF("%d blocks, oracle %d", a->nblocks, b->nblocks);
    for (i = 0; i < a->nblocks; i++)
        if (memcmp(&a->blocks[i], &b->blocks[i], sizeof(J2kBlock))) {
            const J2kBlock *x = &a->blocks[i]; const OrcBlock *y = &b->blocks[i];
            DIFF("block %d of %d: off %u plane %u lcup %u lref %u %ux%u stride %u np %u zbp %u Mb %u fl %02x roi %u tc %u f %g i %d / "
                 "off %u plane %u lcup %u lref %u %ux%u stride %u np %u zbp %u Mb %u fl %02x roi %u tc %u f %g i %d", i, a->nblocks,
                 x->data_off, x->plane_off, x->lcup, x->lref, x->w, x->h, x->stride, x->npasses, x->zbp, x->M_b, x->flags, x->roi_shift, x->tcomp, x->f_step, x->i_step,
                 y->data_off, y->plane_off, y->lcup, y->lref, y->w, y->h, y->stride, y->npasses, y->zbp, y->M_b, y->flags, y->roi_shift, y->tcomp, y->f_step, y->i_step);
        }
    if (a->nbytes != b->nbytes || a->nsamples != b->nsamples)
        DIFF("pool %zu bytes %zu samples / %zu %zu", a->nbytes, a->nsamples, b->nbytes, b->nsamples);
    if (memcmp(a->bytes, b->bytes, a->nbytes + 64)) {
        size_t k = 0;
        while (a->bytes[k] == b->bytes[k]) k++;
        DIFF("byte pool differs at %zu of %zu (%02x / %02x)", k, a->nbytes, a->bytes[k], b->bytes[k]);
    }
    {   /* the same packet through a parser that does not gather: same tables, and the gather table must rebuild the pool */
        const J2kPlan *c = NULL;
        uint8_t *pool;
        int rc = j2k_parse(lazy, pkt, size, o, 0, &c);
        if (rc != ra || !c || c->bytes || c->nblocks != a->nblocks || c->nbytes != a->nbytes ||
            memcmp(c->blocks, a->blocks, (size_t)a->nblocks * sizeof(J2kBlock)) || c->max_scup != a->max_scup || c->max_pcup != a->max_pcup)
            DIFF("non-gathering parse differs from the gathering one (%d)", rc);
        if ((int)c->blk_seg0[c->nblocks] != (int)c->nsegs)
            DIFF("gather table: %u segments, index ends at %u", c->nsegs, c->blk_seg0[c->nblocks]);
        pool = malloc(c->nbytes + 64);
        memset(pool, 0xA5, c->nbytes + 64);
        j2k_plan_gather(c, pool);
        if (memcmp(pool, b->bytes, c->nbytes + 64)) {
            free(pool);
            DIFF("pool rebuilt from the gather table differs");
        }
        free(pool);
    }
    if (a->max_lcup != b->max_lcup || a->max_lref != b->max_lref || a->max_pcup != b->max_pcup || a->max_scup != b->max_scup ||
        a->max_qw != b->max_qw || a->max_bm_words != b->max_bm_words || a->have_part1 != b->have_part1)
        DIFF("sizing figures lcup %u lref %u pcup %u scup %u qw %u bm %u p1 %d / %u %u %u %u %u %u %d",
             a->max_lcup, a->max_lref, a->max_pcup, a->max_scup, a->max_qw, a->max_bm_words, a->have_part1,
             b->max_lcup, b->max_lref, b->max_pcup, b->max_scup, b->max_qw, b->max_bm_words, b->have_part1);
    if (memcmp(a->palette, b->palette, sizeof a->palette))
        DIFF("palette");
}

static void one_file(const char *path, J2kParser *mine, OrcParser *theirs, int iters)
{
    FILE *f = fopen(path, "rb");
    long n, it;
    uint8_t *b;
    htj2k_opts o;
    char what[600];
    if (!f) { printf("cannot open %s\n", path); n_diff++; return; }
    fseek(f, 0, SEEK_END); n = ftell(f); fseek(f, 0, SEEK_SET);
    b = malloc(n + 64);
    if (fread(b, 1, n, f) != (size_t)n) { fclose(f); free(b); return; }
    fclose(f);
    memset(b + n, 0, 64);
    memset(&o, 0, sizeof o);
    o.req_pix_fmt = -1;
    for (it = 0; it < 6; it++) {
        o.reduction_factor = it % 3;
        o.bitexact = it / 3;
        snprintf(what, sizeof what, "%s lowres=%d bitexact=%d", path, o.reduction_factor, o.bitexact);
        compare(what, mine, theirs, b, (int)n, &o, 0);
    }
    o.reduction_factor = 0; o.bitexact = 0;
    snprintf(what, sizeof what, "%s headers-only", path);
    compare(what, mine, theirs, b, (int)n, &o, 1);
    o.strict = 1;
    snprintf(what, sizeof what, "%s strict", path);
    compare(what, mine, theirs, b, (int)n, &o, 0);
    o.strict = 0;
    for (it = 0; it < iters; it++) {
        long m = n;
        int mode = rnd() % 6;
        uint32_t seed_at = rs;
        uint8_t *c;
        if (mode == 0) m = rnd() % (n + 1);
        c = malloc(m + 64);
        memcpy(c, b, m);
        memset(c + m, 0, 64);
        if (mode >= 1 && m > 0) {
            int k = 1 + rnd() % (mode == 4 ? 64 : 4), i;
            for (i = 0; i < k; i++) {
                long pos = (mode == 2 || mode == 5) ? rnd() % (m < 200 ? m : 200) : rnd() % m;
                if (mode == 3 || mode == 5) c[pos] ^= 1 << (rnd() % 8); else c[pos] = rnd();
            }
        }
        o.reduction_factor = (it % 7 == 0) ? rnd() % 4 : 0;
        o.bitexact = (it % 5 == 0);
        o.strict = (it % 11 == 0);
        snprintf(what, sizeof what, "%s mutation %ld (seed state %u, mode %d)", path, it, seed_at, mode);
        compare(what, mine, theirs, c, (int)m, &o, 0);
        free(c);
    }
    free(b);
}

int main(int argc, char **argv)
{
    int iters = 0, a = 1, threads = 0;
    uint32_t ptiles = 0, pretries = 0, lt = 0, lr = 0;
    void *h;
    J2kParser *mine;
    OrcParser *theirs;
    while (a < argc && argv[a][0] == '-') {
        if (!strcmp(argv[a], "-m") && a + 1 < argc) { iters = atoi(argv[a + 1]); a += 2; }
        else if (!strcmp(argv[a], "-s") && a + 1 < argc) { rs = (uint32_t)strtoul(argv[a + 1], NULL, 0); a += 2; }
        else if (!strcmp(argv[a], "-t") && a + 1 < argc) { threads = atoi(argv[a + 1]); a += 2; }
        else if (!strcmp(argv[a], "-v")) { verbose = 1; a++; }
        else break;
    }
    if (a >= argc) { fprintf(stderr, "usage: plan_diff [-m N] [-s SEED] [-v] ORACLE_SO FILE...\n"); return 2; }
    h = dlopen(argv[a++], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "%s\n", dlerror()); return 2; }
    o_new = (OrcParser *(*)(void))dlsym(h, "orc_parser_new");
    o_free = (void (*)(OrcParser *))dlsym(h, "orc_parser_free");
    o_parse = (int (*)(OrcParser *, const uint8_t *, int, const htj2k_opts *, int, const OrcPlan **))dlsym(h, "orc_parse");
    if (!o_new || !o_free || !o_parse) { fprintf(stderr, "oracle parser symbols missing\n"); return 2; }
    mine = j2k_parser_new();
    lazy = j2k_parser_new();
    j2k_parser_set_gather(lazy, 0);
    if (threads > 1) {
        j2k_parser_set_packet_threads(mine, threads);
        j2k_parser_set_packet_threads(lazy, threads);
    }
    theirs = o_new();
    for (; a < argc; a++) {
        size_t L = strlen(argv[a]);
        if (L > 5 && !strcmp(argv[a] + L - 5, ".list")) {
            FILE *lf = fopen(argv[a], "r");
            char line[1024];
            while (lf && fgets(line, sizeof line, lf)) {
                line[strcspn(line, "\r\n")] = 0;
                if (line[0]) one_file(line, mine, theirs, iters);
            }
            if (lf) fclose(lf);
        } else {
            one_file(argv[a], mine, theirs, iters);
        }
    }
    printf("plan_diff: %ld parses, %ld accepted, %ld differences\n", n_parse, n_ok, n_diff);
    j2k_parser_parallel_stats(mine, &ptiles, &pretries);
    j2k_parser_parallel_stats(lazy, &lt, &lr);
    printf("plan_diff: parallel tiles %u, sequential retries %u\n", ptiles + lt, pretries + lr);
    j2k_parser_free(mine);
    j2k_parser_free(lazy);
    o_free(theirs);
    return n_diff ? 1 : 0;
}

/* htj2k_enc.h -- test-vector factory (see htj2k_enc.c).  Test tooling only. */
#ifndef HTJ2K_ENC_H
#define HTJ2K_ENC_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct htj2k_enc_params {
    int width, height;          /* image area (Xsiz - XOsiz, Ysiz - YOsiz) */
    int x_off, y_off;           /* XOsiz, YOsiz */
    int tile_w, tile_h;         /* 0 = one tile */
    int tx_off, ty_off;         /* XTOsiz, YTOsiz */
    int ncomp;
    int depth[4], sgnd[4], dx[4], dy[4];
    int nlevels;                /* decomposition levels NL */
    int cb_w_log2, cb_h_log2;   /* 2..10, sum <= 12 */
    int transform;              /* 1 = reversible 5/3, 0 = irreversible 9/7 (COD value) */
    int mct;
    int guard_bits;             /* 0 = 2 */
    int prog_order;             /* 0 LRCP 1 RLCP 2 RPCL 3 PCRL 4 CPRL */
    int nprec;                  /* 0 = maximal precincts, else entries in prec_*_log2 (last one repeats) */
    int prec_w_log2[34], prec_h_log2[34];
    double qstep;               /* 9/7: base step size relative to the sample range */
    int expn_bias;              /* 5/3: added to every exponent (more headroom) */
    int passes;                 /* 1 cleanup; 2 +SigProp; 3 +SigProp+MagRef */
    int placeholder_sets;       /* p0: 3*p0 placeholder passes signalled before the cleanup pass */
    int cblk_style;             /* only 0x08 (vertically causal) is honoured */
    int sop, eph;
    int force_include;          /* code all-zero blocks too */
    int never_empty_packets;
    int psot_zero;              /* last tile-part: Psot = 0 */
    int rsiz;
    int cap_extra_bits;         /* OR'ed into Ccap15 bits 11..15 (test error paths) */
    const char *comment;
    int part1;                  /* 1: Part-1 (MQ-coded) blocks, no CAP marker; cblk_style then honours BYPASS 0x01,
                                 * RESET 0x02, TERMALL 0x04, VSC 0x08, SEGSYM 0x20 */
    int p1_drop_passes;         /* Part-1: leave out the last N coding passes of every block (lossy truncation) */
    int mixed;                  /* 1: MIXED stream (SPcod bits 6-7 = 3, Ccap15 bits 14-15 = 3): HT and Part-1 blocks in a
                                 * checkerboard; of cblk_style only VSC is honoured */
} htj2k_enc_params;

/* comps[c]: int32 samples of component c, row-major, ceil(X1/dx)-ceil(X0/dx) wide.
 * Returns 0 and a malloc'ed codestream (free with htj2k_enc_free), or <0:
 * -4 = a band needs more magnitude bits than M_b (raise guard_bits / expn_bias). */
int  htj2k_encode(const htj2k_enc_params *P, const int32_t *const comps[4], uint8_t **out, size_t *out_len);
void htj2k_enc_free(uint8_t *p);
int  htj2k_encode_block(const int32_t *vals, int w, int h, int passes, int causal,
                        uint8_t **out, int *lcup, int *lref, int *max_U);
int  htj2k_encode_block_p1(const int32_t *vals, int w, int h, int band, int style, int drop_passes,
                           uint8_t **out, int *kbits, int *npasses, int *nseg, int *seglen, int *segpasses);
#ifdef __cplusplus
}
#endif
#endif

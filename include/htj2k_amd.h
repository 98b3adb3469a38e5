/*
 * htj2k_amd.h -- C ABI of the MI355X-native HTJ2K decode path.
 *
 * This is the drop-in boundary for FFmpeg's JPEG 2000 decoder plugin
 * (`const FFCodec ff_jpeg2000_decoder`, libavcodec/jpeg2000dec.c:2926-2939):
 *
 *   FFCodec.init   (jpeg2000_decode_init,  jpeg2000dec.c:2807)  -> htj2k_open()
 *   FFCodec.cb.decode (jpeg2000_decode_frame, jpeg2000dec.c:2825) -> htj2k_probe() + htj2k_decode()
 *   FFCodec.close  (none in the reference; device state persists here)  -> htj2k_close()
 *
 * Everything from `tile_codeblocks()` down (jpeg2000dec.c:2212-2299: HT block
 * decode, dequantisation, inverse DWT) plus `mct_decode()` (:2183) and
 * `write_frame_8/16()` (:2301-2364) runs as HIP kernels on gfx950.  Marker and
 * Tier-2 packet parsing (jpeg2000dec.c:197-1869) stays on the host in C.
 *
 * Plain C types only: no FFmpeg, no torch, no HIP types cross this boundary.
 * The FFmpeg-side binding a maintainer adds is shown in INTEGRATION.md.
 */
#ifndef HTJ2K_AMD_H
#define HTJ2K_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes: numerically identical to FFmpeg's AVERROR values
 *      (libavutil/error.h:41,61,64) so the glue can return them unchanged ---- */
#define HTJ2K_ERR_INVALIDDATA   (-0x41444E49) /* AVERROR_INVALIDDATA  = -MKTAG('I','N','D','A') */
#define HTJ2K_ERR_PATCHWELCOME  (-0x45574150) /* AVERROR_PATCHWELCOME = -MKTAG('P','A','W','E') */
#define HTJ2K_ERR_BUG           (-0x21475542) /* AVERROR_BUG          = -MKTAG('B','U','G','!') */
#define HTJ2K_ERR_EXTERNAL      (-0x20545845) /* AVERROR_EXTERNAL     = -MKTAG('E','X','T',' ') : HIP runtime failure */
#define HTJ2K_ERR_EAGAIN        (-11)         /* AVERROR(EAGAIN): pipeline: send more / receive first */
#define HTJ2K_ERR_ENOMEM        (-12)         /* AVERROR(ENOMEM) */
#define HTJ2K_ERR_EINVAL        (-22)         /* AVERROR(EINVAL) */
#define HTJ2K_ERR_ENOSYS        (-38)         /* AVERROR(ENOSYS): no usable gfx950 device */

/* ---- output sample layouts the reference can pick in get_siz()
 *      (jpeg2000dec.c:170-193,330-420).  The glue maps them 1:1 to AV_PIX_FMT_*. ---- */
enum htj2k_pix_fmt {
    HTJ2K_PIX_NONE = -1,
    HTJ2K_PIX_PAL8 = 0, HTJ2K_PIX_RGB24, HTJ2K_PIX_RGBA, HTJ2K_PIX_RGB48, HTJ2K_PIX_RGBA64,
    HTJ2K_PIX_GRAY8, HTJ2K_PIX_YA8, HTJ2K_PIX_GRAY16, HTJ2K_PIX_YA16,
    HTJ2K_PIX_YUV410P, HTJ2K_PIX_YUV411P, HTJ2K_PIX_YUVA420P,
    HTJ2K_PIX_YUV420P, HTJ2K_PIX_YUV422P, HTJ2K_PIX_YUVA422P,
    HTJ2K_PIX_YUV440P, HTJ2K_PIX_YUV444P, HTJ2K_PIX_YUVA444P,
    HTJ2K_PIX_YUV420P9, HTJ2K_PIX_YUV422P9, HTJ2K_PIX_YUV444P9,
    HTJ2K_PIX_YUVA420P9, HTJ2K_PIX_YUVA422P9, HTJ2K_PIX_YUVA444P9,
    HTJ2K_PIX_YUV420P10, HTJ2K_PIX_YUV422P10, HTJ2K_PIX_YUV444P10,
    HTJ2K_PIX_YUVA420P10, HTJ2K_PIX_YUVA422P10, HTJ2K_PIX_YUVA444P10,
    HTJ2K_PIX_YUV420P12, HTJ2K_PIX_YUV422P12, HTJ2K_PIX_YUV444P12,
    HTJ2K_PIX_YUV420P14, HTJ2K_PIX_YUV422P14, HTJ2K_PIX_YUV444P14,
    HTJ2K_PIX_YUV420P16, HTJ2K_PIX_YUV422P16, HTJ2K_PIX_YUV444P16,
    HTJ2K_PIX_YUVA420P16, HTJ2K_PIX_YUVA422P16, HTJ2K_PIX_YUVA444P16,
    HTJ2K_PIX_XYZ12,
    HTJ2K_PIX_NB
};

/* options: mirror of what jpeg2000dec.c reads from AVCodecContext / its AVOption
 * (jpeg2000dec.c:224,543,2488,2811-2817,2913-2917) plus device selection */
typedef struct htj2k_opts {
    int bitexact;          /* AV_CODEC_FLAG_BITEXACT: 9/7 float -> 9/7 fixed point (jpeg2000dec.c:543) */
    int reduction_factor;  /* private option "lowres" (jpeg2000dec.c:2913-2917) */
    int64_t max_pixels;    /* avctx->max_pixels (jpeg2000dec.c:224); 0 = INT_MAX */
    int strict;            /* strict_std_compliance >= FF_COMPLIANCE_STRICT (jpeg2000dec.c:2488) */
    int device_id;         /* HIP device ordinal */
    int frames_in_flight;  /* device-side pipeline depth: the `depth` of htj2k_pipe_open when that is called with depth 0
                            * (batches in flight, each on a HIP stream of its own); 0 = default (3) */
    int req_pix_fmt;       /* avctx->pix_fmt preset by the caller, or HTJ2K_PIX_NONE (jpeg2000dec.c:354) */
} htj2k_opts;

/* what jpeg2000_read_main_headers()/get_siz() write back into AVCodecContext
 * (jpeg2000dec.c:213,326,330-420,546,2867) */
typedef struct htj2k_info {
    int width, height;         /* ff_set_dimensions() arguments (jpeg2000dec.c:326) */
    int pix_fmt;               /* enum htj2k_pix_fmt */
    int bits_per_raw_sample;   /* s->precision (jpeg2000dec.c:420) */
    int profile;               /* Rsiz (jpeg2000dec.c:213) */
    int lossless;              /* FF_CODEC_PROPERTY_LOSSLESS (jpeg2000dec.c:546) */
    int sar_num, sar_den;      /* JP2 'res ' box (jpeg2000dec.c:2762-2795,2867) */
    int ncomponents;
    int is_ht;                 /* CAP marker announced Part 15 (jpeg2000dec.c:437) */
    int nplanes;               /* planes of pix_fmt */
    int plane_width[4];        /* in samples */
    int plane_height[4];
    int plane_bytes_per_sample[4]; /* bytes per sample * samples per pixel in that plane (row = this * plane_width) */
    int has_palette;           /* pal8 (JP2 pclr box): plane 1 is the palette, 256 native-endian 0xAARRGGBB entries (jpeg2000dec.c:2900-2901) */
} htj2k_info;

/* mirror of AVFrame.data/linesize (libavutil/frame.h:410,434): caller-owned system memory */
typedef struct htj2k_frame {
    uint8_t *data[4];
    int      linesize[4];
    int      width, height;    /* filled by htj2k_decode */
    int      pix_fmt;
} htj2k_frame;

/* per-call statistics (all optional) */
typedef struct htj2k_stats {
    int   n_codeblocks;        /* codeblocks dispatched to the device */
    int   n_block_errors;      /* blocks the HT decoder rejected (left zero, frame still returned;
                                  jpeg2000dec.c:2275-2278, jpeg2000htdec.c:1305-1306) */
    float ms_parse;            /* host marker + Tier-2 parse */
    float ms_h2d, ms_kernels, ms_d2h;
    float ms_ht, ms_idwt, ms_pack; /* device time per stage (hipEvent) */
} htj2k_stats;

typedef struct htj2k_ctx htj2k_ctx;

typedef void (*htj2k_log_fn)(void *opaque, int level, const char *msg);

/* FFCodec.init equivalent.  Fails with HTJ2K_ERR_ENOSYS when no gfx950 device /
 * HIP runtime is usable: there is NO CPU fallback in this library. */
int  htj2k_open(const htj2k_opts *opts, htj2k_ctx **out);
/* FFCodec.close equivalent */
void htj2k_close(htj2k_ctx *ctx);
void htj2k_set_log(htj2k_ctx *ctx, htj2k_log_fn fn, void *opaque);

/* Parse the main header only and report what get_siz()/get_cod() would set on the
 * AVCodecContext, so the glue can call ff_thread_get_buffer() before decoding
 * (jpeg2000dec.c:2864-2878).  Returns 0 or a negative HTJ2K_ERR_*. */
int  htj2k_probe(htj2k_ctx *ctx, const uint8_t *pkt, int pkt_size, htj2k_info *info);

/* One packet (= one codestream or JP2 file) -> one frame in caller memory.
 * Returns bytes consumed (>= 0) like FFCodec.cb.decode (codec_internal.h:188-192),
 * or a negative HTJ2K_ERR_*.  The packet is only read; nothing is retained. */
int  htj2k_decode(htj2k_ctx *ctx, const uint8_t *pkt, int pkt_size,
                  htj2k_frame *frame, htj2k_stats *stats);

/* ---- staged interface (what htj2k_decode does internally), used by the frame
 *      pipeline and by bench.py to time the device-resident hot path ---- */
typedef struct htj2k_job htj2k_job;  /* one parsed frame: descriptors + device buffers */

/* host: markers + Tier-2 -> per-codeblock descriptor table (no device work) */
int  htj2k_job_parse(htj2k_ctx *ctx, const uint8_t *pkt, int pkt_size, htj2k_job **job);
/* same for a batch of independent frames (the reference's frame-thread axis,
 * libavcodec/pthread_frame.c:856-889): the descriptor tables are concatenated so that each
 * device stage of the whole batch is ONE launch */
int  htj2k_job_parse_batch(htj2k_ctx *ctx, const uint8_t *const *pkts, const int *pkt_sizes, int nframes,
                           htj2k_job **job);
/* as htj2k_job_parse_batch; pinned[i] != 0 says that packet i lies in page-locked memory (htj2k_host_alloc) and stays
 * valid and unchanged until htj2k_job_upload's transfers are done (htj2k_job_wait): the H2D copy then starts from the
 * packet itself and the staging copy is left out.  pinned == NULL: none is. */
int  htj2k_job_parse_batch_ex(htj2k_ctx *ctx, const uint8_t *const *pkts, const int *pkt_sizes, int nframes,
                              const uint8_t *pinned, htj2k_job **job);
int  htj2k_job_num_frames(const htj2k_job *job);
/* host cost of the last htj2k_job_parse(_batch), averaged over its frames (each frame is timed on the thread that worked
 * on it): `ms_parse` the marker + Tier-2 parse (no code-block byte is read with "device_gather", the default), `ms_stage`
 * the copy of the packet into pinned memory that the H2D transfer starts from (for a single large packet, whose copy runs
 * on helper threads under the parse: the part of it the parse did not cover) */
int  htj2k_job_host_ms(const htj2k_job *job, float *ms_parse, float *ms_stage);
int  htj2k_job_frame_info(const htj2k_job *job, int frame, htj2k_info *info);
int  htj2k_job_download_frame(htj2k_ctx *ctx, htj2k_job *job, int frame, htj2k_frame *out);
/* per-launch device time (ms) and algorithmic bytes (2 * 4 * lh * lv per plane and level, one
 * read + one write of every sample) of the IDWT kernels of the last run; returns the number of
 * launches.  htj2k_job_idwt_hbm_bytes() gives, for the same launches, the bytes the kernel has
 * to move through HBM at least: the same figure for a plain level, 4 * lh * lv + the frame bytes
 * written for a final level that is fused with the MCT / pack stage. */
int  htj2k_job_idwt_launches(htj2k_ctx *ctx, htj2k_job *job, float *ms, double *bytes, int cap);
/* 1 when the last htj2k_job_run kept the sub-bands as 16-bit samples between the block decoder and the inverse DWT
 * (exact: reversible 5/3 jobs whose every band has M_b <= 15, rgb24 output, all levels of even geometry; knob
 * "coef16", default on).  The reference holds them as int32 (comp->i_data, jpeg2000.c:499-511). */
int  htj2k_job_coef16(const htj2k_job *job);
/* codeblocks per wavefront in the MagSgn kernel of the last HT stage run: 1 (k_ht_decode: a lane per sample column), 2
 * (k_ht_decode_pair, or k_ht_decode_multi with blocks of up to 64 columns), 4 (k_ht_decode_multi, blocks of up to 32
 * columns); 0 before the first run.  Which kernel applies is decided per job: DESIGN.md section 3.1. */
int  htj2k_job_ht_blocks_per_wave(const htj2k_job *job);
/* 0: the last run held the LL bands between the IDWT levels as int32 (as the reference, jpeg2000dwt.c:539-581);
 * 1: as 16-bit samples (knob "ll16", jobs with 16-bit sub-bands only); 2: it did, a sample of an LL band did not fit
 * -- only crafted or corrupt coefficients do that -- and htj2k_job_wait / _download ran the transform again with
 * int32 LL bands before handing out the frames */
int  htj2k_job_ll16(const htj2k_job *job);
/* 0: no launch of the last run's final IDWT level ran on pairs of 16-bit samples; otherwise the number of bits the LL
 * bands of the job had to fit for that (10..16; knob "idwt_pk").  The final 5/3 level of 8-bit pictures (rgb24 or 8-bit
 * planes out, 16-bit sub-bands and LL band in) is computed with packed 16-bit instructions -- lifting, inverse RCT
 * (jpeg2000dsp.c:78-91) and clip -- where the bands' M_b and the checked range of the LL band prove that no intermediate
 * leaves 16 bits; the result is the reference's int32 arithmetic exactly. */
int  htj2k_job_idwt_packed(const htj2k_job *job);
int  htj2k_job_idwt_hbm_bytes(htj2k_ctx *ctx, htj2k_job *job, double *bytes, int cap);
/* H2D: compressed codeblock bytes + descriptors (async on the job's stream) */
int  htj2k_job_upload(htj2k_ctx *ctx, htj2k_job *job);
/* device: HT block decode + dequant -> IDWT -> MCT/level shift/clip/pack (async) */
int  htj2k_job_run(htj2k_ctx *ctx, htj2k_job *job);
/* D2H into caller planes, then waits for the job's stream */
int  htj2k_job_download(htj2k_ctx *ctx, htj2k_job *job, htj2k_frame *frame);
int  htj2k_job_wait(htj2k_ctx *ctx, htj2k_job *job);
int  htj2k_job_info(const htj2k_job *job, htj2k_info *info);
int  htj2k_job_bytes_consumed(const htj2k_job *job);
void htj2k_job_free(htj2k_ctx *ctx, htj2k_job *job);
/* debugging / parity hooks: copy a tile-component's coefficient plane (int32 or
 * float, after the last stage that ran) back to the host */
int  htj2k_job_num_tilecomps(const htj2k_job *job);
int  htj2k_job_tilecomp_dims(const htj2k_job *job, int tc, int *w, int *h, int *is_float);
int  htj2k_job_read_plane(htj2k_ctx *ctx, htj2k_job *job, int tc, void *dst, size_t dst_bytes);
/* run only some stages (bit 0 = HT+dequant, bit 1 = IDWT, bit 2 = MCT+pack) */
int  htj2k_job_run_stages(htj2k_ctx *ctx, htj2k_job *job, int stage_mask);
/* device-side event timing of the last run of each stage, ms */
int  htj2k_job_stage_ms(htj2k_ctx *ctx, htj2k_job *job, float *ms_ht, float *ms_idwt, float *ms_pack);

/* ---- kernel-level entry points (unit parity tests call these through the C ABI) ---- */
/* ff_dwt_decode() (jpeg2000dwt.c:601) on a host plane: uploads, runs the IDWT
 * kernels, downloads.  border = {{x0,x1},{y0,y1}} as ff_jpeg2000_dwt_init
 * (jpeg2000dwt.c:539); type: 0 = 9/7 float, 1 = 5/3, 2 = 9/7 fixed. */
int  htj2k_idwt_plane(htj2k_ctx *ctx, void *plane, const int border[2][2],
                      int decomp_levels, int type);
/* same, device-resident and timed: runs `iters` back-to-back transforms of
 * `nplanes` identical-geometry planes and returns mean ms per iteration */
int  htj2k_idwt_bench(htj2k_ctx *ctx, int w, int h, int decomp_levels, int type,
                      int nplanes, int iters, float *ms_per_iter);
/* calibration for the roofline figures (no counterpart in the reference): GB/s, read + written, of a kernel that only
 * copies `mbytes` MB of device memory per launch (16-byte elements, grid-stride; best of three launch shapes) on the
 * context's device -- the ceiling a bandwidth-bound kernel can be held against on this particular box */
int  htj2k_copy_bench(htj2k_ctx *ctx, int mbytes, int iters, float *gbps);
/* The host-side proof behind knob "idwt_pk" (no GPU needed; tests/test_pk16_bounds.py checks it against a simulation
 * of the kernel's wrapping 16-bit arithmetic): with |LL| <= ll, |HL| <= hl, |LH| <= lh, |HH| <= hh on one level of the
 * inverse 5/3 transform (jpeg2000dwt.c:309-385), the largest magnitude any output sample can have, or -1 when some
 * intermediate sum of the horizontal or vertical lifting could leave 16 bits. */
long htj2k_pk16_lift_bound(long ll, long hl, long lh, long hh);
/* ... and for a group of `nc` components, b[c] = { ll, hl, lh, hh }, followed by the inverse RCT when `rct` != 0
 * (jpeg2000dsp.c:78-91; the two final sums of the RCT saturate and are not bounded): 1 = exact in 16 bits, 0 = not */
int  htj2k_pk16_bounds(const long (*b)[4], int nc, int rct);
/* Jpeg2000DSPContext.mct_decode[type] (jpeg2000dsp.c:43-91) on host planes */
int  htj2k_mct_planes(htj2k_ctx *ctx, void *p0, void *p1, void *p2, int csize, int type);

/* ff_jpeg2000_decode_htj2k() + dequantisation (jpeg2000htdec.c:1188, jpeg2000dec.c:2098-2181)
 * on a caller-built table of `nblocks` 32-byte codeblock descriptors (layout: struct J2kBlock
 * in ffmpeg-ht_amd/csrc/j2k_plan.h) over a byte pool, into `coef` (nsamples 32-bit samples).
 * status[i] != 0: block i was rejected and left zero. */
int  htj2k_ht_blocks(htj2k_ctx *ctx, const void *blocks, int nblocks, const uint8_t *bytes, size_t nbytes,
                     void *coef, size_t nsamples, int *status);
/* decode_cblk() + dequantisation (jpeg2000dec.c:1993-2089, 2098-2181) on a table of Part-1 (MQ-coded) blocks:
 * descriptors with J2K_BLK_PART1 set, bytes laid out as in j2k_plan.h (segments back to back, 0xFF 0xFF behind
 * terminated ones and behind the last byte, J2kPart1Trailer behind that).  status[i] != 0: decode_cblk() failed
 * part-way ("bpno became invalid", "Missing needed termination"); `coef` then holds the passes decoded up to there,
 * which is what the reference dequantises (jpeg2000dec.c:2275-2290). */
int  htj2k_mq_blocks(htj2k_ctx *ctx, const void *blocks, int nblocks, const uint8_t *bytes, size_t nbytes,
                     void *coef, size_t nsamples, int *status);
/* codeblocks the HT decoder rejected in the job's last run (they are left zero) */
int  htj2k_job_block_errors(htj2k_ctx *ctx, htj2k_job *job);
int  htj2k_job_num_blocks(const htj2k_job *job);
/* device addresses of the decoded planes of frame `frame` of the job (data[] = device pointers), for callers that
 * keep frames on the GPU; valid until the job is parsed again */
int  htj2k_job_device_frame(htj2k_ctx *ctx, htj2k_job *job, int frame, htj2k_frame *out);
/* device address of an output plane, for callers that keep decoded frames on the GPU.  Call htj2k_job_wait first: the
 * planes of a run are final only after it (htj2k_job_device_frame waits by itself) */
void *htj2k_job_device_plane(htj2k_job *job, int plane, int *linesize);
/* tuning / test knobs:
 *   "idwt_mode"   0 generic closed-form kernels, 1 LDS tile kernel, 3 register-streaming kernel (default)
 *   "fuse_pack"   1 (default): with idwt_mode 3, a run that covers both the IDWT and the pack stage
 *                 lets the final IDWT level do the inverse MCT and write the frame
 *   "ht_mode"     1 (default) k_ht_unstuff + k_ht_vlc + k_ht_decode<true>, 0 single kernel
 *   "packet_threads" 1 (default) .. 16: a frame that is parsed on its own (htj2k_decode, jobs of one frame) has the packets
 *                 of every tile with a complete PLT packet-length list, one quality layer and no PPM / PPT read by this many
 *                 threads; the plan is the same, and any disagreement between the list and the packets sends the frame through
 *                 the sequential reader (csrc/j2k_tier2.c: read_tile_parallel)
 *   "ht_multi"    1 (default): jobs with 32-bit sub-bands whose HT blocks all are cleanup-only, at most 64 columns wide,
 *                 without ROI shift and of one transform decode 2 or 4 blocks per wavefront (k_ht_decode_multi); 0: one
 *                 block per wavefront, a lane per sample column
 *   "coef16"      1 (default): jobs that qualify keep the sub-bands as 16-bit samples (htj2k_job_coef16)
 *   "ht_pair"     1 (default): such jobs decode MagSgn with k_ht_decode_pair (two blocks per wave, a lane per quad)
 *   "idwt_x3"     1 (default): jobs with 16-bit LL bands run the first three 5/3 levels of every plane as one launch, the LL
 *                 bands in between in LDS (k_idwt_stream_ll16_x3); 0: one launch per level
 *   "idwt_pk"     1 (default): final 5/3 levels of 8-bit pictures on pairs of 16-bit samples where that is exact
 *                 (htj2k_job_idwt_packed); 0: 32-bit arithmetic throughout
 *   "ll16"        1 (default): such jobs also hold the LL bands between the IDWT levels as 16-bit samples, with a check
 *                 on the device and a second run with int32 LL bands should one not fit (htj2k_job_ll16)
 *   "device_gather"  1 (default): packets are uploaded as they are and the byte pool of the job is put together by
 *                 k_gather on the device from the parser's gather table; 0: the parser copies the code-block bytes
 *                 into the pool on the host
 *   "parse_threads"  host threads that parse the frames of a batch (0 = min(cores, 16))
 *   "bitexact", "reduction_factor"   as the AVCodecContext flag / the decoder's `lowres` option
 * Environment variables (experiments and tests; read by htj2k_open unless noted):
 *   HTJ2K_HT=fused|split             "ht_mode" 0 | 1
 *   HTJ2K_IDWT=generic|tile|stream   "idwt_mode" 0 | 1 | 3
 *   HTJ2K_LL16=0|1, HTJ2K_FUSE=0|1, HTJ2K_PK=0|1   "ll16", "fuse_pack", "idwt_pk"
 *   HTJ2K_MULTI_LDS=bytes            (per upload) most LDS a wave of k_ht_decode_multi may take for its blocks' MagSgn bits
 *                                    (default 12288): above it the job decodes one block per wave
 *   HTJ2K_UNSTUFF_G=1|2|4            (per launch) codeblocks per wavefront in the un-stuffing kernel (default 4 for jobs without blocks wider
 *                                    than 32 columns, else 2)
 *   HTJ2K_STRIP=rows                 (per launch) rows per wave of the streaming IDWT kernels (default 8 or 16 by launch size)
 *   HTJ2K_TW16 / HTJ2K_TW32 / HTJ2K_TWF=columns   (per launch) output columns per wave of the streaming IDWT for 16-bit LL
 *                                    bands / 32-bit LL bands / the fused final level (64 .. 244; default 224 or 244 by row length)
 *   HTJ2K_WPB=waves                  (first launch) waves per workgroup of the 16-bit streaming IDWT kernels, 1 .. 8 (default: the
 *                                    strips of a row, at most 8)
 *   HTJ2K_PK_LDS=bytes               (first launch) dynamic LDS the packed final-level kernel is launched with -- an occupancy limit,
 *                                    the kernel uses none (default: two workgroups of eight waves per CU)
 *   HTJ2K_X3_TH=rows                 (per launch) rows of the third level one workgroup of k_idwt_stream_ll16_x3 reconstructs (default 24)
 *   HTJ2K_POISON=1                   fresh device buffers start as 0xA5 bytes (tools/gpu_random_configs.py) */
int  htj2k_set_int(htj2k_ctx *ctx, const char *name, int value);

/* Pinned (page-locked) host memory for frame planes: a D2H copy into it runs at PCIe rate (about 5x a
 * copy into pageable memory).  An integration wraps it in its buffer pool, e.g. av_buffer_create()
 * with htj2k_host_free as the free callback.  Plain pageable planes work everywhere, only slower. */
void *htj2k_host_alloc(htj2k_ctx *ctx, size_t size);
void  htj2k_host_free(htj2k_ctx *ctx, void *ptr);
/* copies `size` bytes of a device plane handed out by htj2k_pipe_receive_device / htj2k_job_device_frame to host memory
 * (a consumer that keeps frames on the GPU and wants one on the host after all); synchronous */
int   htj2k_device_to_host(htj2k_ctx *ctx, void *dst, const void *device_src, size_t size);

/* ---- asynchronous pipeline: packets in, frames out, in order (csrc/htj2k_pipe.cpp) ----
 * The throughput path for a stream of frames.  It takes the place of FFmpeg's frame threads
 * (libavcodec/pthread_frame.c:856-889: N decoder contexts, one packet each) and maps onto
 * FFCodec.cb.receive_frame: `depth` device jobs of `batch` frames are kept in flight, so that host
 * parsing (on several threads, see "parse_threads"), PCIe transfers and kernels overlap.
 *   htj2k_pipe_send     queues a copy of the packet; HTJ2K_ERR_EAGAIN when `depth` batches are
 *                       waiting to be received
 *   htj2k_pipe_flush    starts the partly filled batch (end of stream, or latency matters)
 *   htj2k_pipe_info     what the next frame will be (blocks until its batch is decoded);
 *                       HTJ2K_ERR_EAGAIN when nothing is in flight
 *   htj2k_pipe_receive  copies the next frame into the caller's planes; a packet that failed
 *                       returns its own error, the other frames of its batch are still delivered
 *   htj2k_pipe_skip     drops the next frame
 * One producer/consumer thread at a time may use a pipe (like an AVCodecContext). */
typedef struct htj2k_pipe htj2k_pipe;
/* batch: 1 .. 256 frames per device job; depth: 1 .. 16 jobs in flight, 0 = htj2k_opts.frames_in_flight */
int  htj2k_pipe_open(htj2k_ctx *ctx, int batch, int depth, htj2k_pipe **pipe);
int  htj2k_pipe_send(htj2k_pipe *pipe, const uint8_t *pkt, int size);
/* as htj2k_pipe_send without the copy: `pkt` (with its 64 bytes of input padding) stays valid until
 * release(opaque) is called -- when the packet's frame has been received or skipped, or on close */
int  htj2k_pipe_send_ref(htj2k_pipe *pipe, const uint8_t *pkt, int size, void (*release)(void *opaque), void *opaque);
int  htj2k_pipe_flush(htj2k_pipe *pipe);
int  htj2k_pipe_info(htj2k_pipe *pipe, htj2k_info *info);
int  htj2k_pipe_receive(htj2k_pipe *pipe, htj2k_frame *out);
/* as htj2k_pipe_receive without the copy: `out->data[]` are the device pointers of the decoded planes; they stay
 * valid until the pipe has handed out all frames of `depth - 1` further batches.  The pipe keeps 2 depth - 1 jobs for
 * this: `depth` batches in flight and the depth - 1 most recent ones whose frames are out */
int  htj2k_pipe_receive_device(htj2k_pipe *pipe, htj2k_frame *out);
/* the same for consumers that keep frames for as long as they like (reference-counted frames: the AV_PIX_FMT_HIP
 * hand-out of glue/jpeg2000_hip_hw.c): the planes stay valid until htj2k_pipe_release_device(token) -- callable from
 * any thread, e.g. an AVBuffer free callback; a token releases its own frame, once.  Until all frames of a batch are
 * released its job is not reused; the other jobs go on (a new batch takes any free job, frames still come out in the
 * order their packets went in).  A consumer that sits on frames of 2 depth - 1 batches starves the pipe (htj2k_pipe_send
 * answers HTJ2K_ERR_EAGAIN and nothing is in flight): size `depth` for the frames the consumer holds, as
 * extra_hw_frames does for hardware decoders. */
int  htj2k_pipe_receive_device_ref(htj2k_pipe *pipe, htj2k_frame *out, uint64_t *token);
int  htj2k_pipe_release_device(htj2k_pipe *pipe, uint64_t token);
int  htj2k_pipe_skip(htj2k_pipe *pipe);
/* Stops the workers, drops what is queued and frees the pipe -- unless frames handed out by
 * htj2k_pipe_receive_device_ref are still out: then the jobs they live in, the pipe and the context (the pipe holds a
 * reference: an htj2k_close by the caller does not free it yet) stay until the last htj2k_pipe_release_device, the only
 * call the handle is still good for.  Frames may so outlive the decoder that made them. */
void htj2k_pipe_close(htj2k_pipe *pipe);

/* ---- framing: cutting a byte stream of back-to-back frames into packets ----------------------------
 * The reference's AVCodecParser for this codec, `ff_jpeg2000_parser` (libavcodec/jpeg2000_parser.c:213-218,
 * used by av_parser_parse2() for raw .j2k / .jp2 sequences and image pipes).  No device involved.
 *   htj2k_splitter_find_end  = find_frame_end() (jpeg2000_parser.c:92-184): scans `size` more bytes and returns
 *                              the offset, relative to `buf`, of the first byte of the NEXT frame -- negative
 *                              (down to -11) when that byte arrived with an earlier call -- or
 *                              HTJ2K_SPLIT_END_NOT_FOUND.
 *   htj2k_splitter_parse     = jpeg2000_parse() + ff_combine_frame() (jpeg2000_parser.c:186-211,
 *                              parser.c:203-288): returns the number of bytes of `buf` consumed; when a frame is
 *                              complete *frame / *frame_size describe it (valid until the next call, followed by
 *                              zeroed input padding unless it points into `buf`), else they are NULL / 0.
 *                              Call with size 0 at the end of the input to flush the last frame. */
#define HTJ2K_SPLIT_END_NOT_FOUND (-100)      /* END_NOT_FOUND, libavcodec/parser.h:40 */
typedef struct htj2k_splitter htj2k_splitter;
int  htj2k_splitter_open(htj2k_splitter **sp);
int  htj2k_splitter_find_end(htj2k_splitter *sp, const uint8_t *buf, int size);
int  htj2k_splitter_parse(htj2k_splitter *sp, const uint8_t *buf, int size, const uint8_t **frame, int *frame_size);
void htj2k_splitter_close(htj2k_splitter *sp);

/* ---- MXF: JPEG 2000 picture essence out of a file held in memory ------------------------------------
 * The KLV layer of the reference's demuxer (klv_read_packet, libavformat/mxfdec.c:432-504, and the essence
 * branch of mxf_read_packet, :4034-4160); no header metadata is read.  Starting at *pos, finds the next
 * generic-container picture element that carries JPEG 2000 (SMPTE 422M: key ...0d.01.03.01.15.nn.08.nn
 * frame-wrapped, ...15.nn.09.nn clip-wrapped; libavformat/mxfenc.c:216-217) and advances *pos behind it.
 * Returns 1 with *out filled in (pointing into `buf`), 0 at the end of the buffer, < 0 on a malformed length.
 * A frame-wrapped element is one packet for htj2k_decode / htj2k_pipe_send; a clip-wrapped one holds all
 * codestreams back to back and is cut apart by htj2k_splitter_*. */
#define HTJ2K_MXF_FRAME_WRAPPED 1
#define HTJ2K_MXF_CLIP_WRAPPED  2
typedef struct htj2k_mxf_essence {
    const uint8_t *data;
    size_t   size;                /* shorter than the KLV length when the file is truncated */
    size_t   klv_offset;          /* of the element's key (AVPacket.pos) */
    uint32_t track_number;        /* key bytes 12..15: matches the track's TrackNumber (SMPTE 379M 7.3) */
    int      wrapping;
} htj2k_mxf_essence;
int  htj2k_mxf_next_essence(const uint8_t *buf, size_t size, size_t *pos, htj2k_mxf_essence *out);

const char *htj2k_version(void);
/* name of the device the context is bound to, e.g. "gfx950" */
const char *htj2k_device_name(htj2k_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* HTJ2K_AMD_H */

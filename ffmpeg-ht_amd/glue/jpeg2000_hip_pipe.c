/*
 * jpeg2000_hip_pipe.c -- FFmpeg-side binding of the MI355X HTJ2K decode library, throughput form.
 *
 * Same plugin as glue/jpeg2000_hip.c (an FFCodec for AV_CODEC_ID_JPEG2000 beside `ff_jpeg2000_decoder`,
 * libavcodec/jpeg2000dec.c:2926-2939), but with the decoupled callback
 *     FFCodec.cb.receive_frame                     libavcodec/codec_internal.h:200-208
 * instead of FF_CODEC_DECODE_CB: the decoder pulls packets with ff_decode_get_packet() (decode.h:58-64) and hands
 * frames out when they are ready.  Behind it sits the library's asynchronous pipeline (htj2k_pipe_*): `depth` device
 * jobs of `batch` frames in flight, so that host parsing (several threads), PCIe transfers and kernels of different
 * batches overlap.  It takes the place of FFmpeg's frame threads (libavcodec/pthread_frame.c:856-889: N contexts, one
 * packet each), so AV_CODEC_CAP_FRAME_THREADS is NOT set; latency is (batch * depth) frames unless the caller drains.
 * avcodec_send_packet() / avcodec_receive_frame() are unchanged (libavcodec/decode.c:620-760).
 *
 * Needs the FFmpeg build tree (config.h): not compiled in this repository.  Build line: INTEGRATION.md section 2.
 */
#include "libavutil/avassert.h"
#include "libavutil/imgutils.h"
#include "libavutil/opt.h"
#include "libavutil/pixdesc.h"
#include "avcodec.h"
#include "codec_internal.h"
#include "decode.h"

#include <htj2k_amd.h>

typedef struct Jpeg2000HipPipeContext {
    const AVClass *class;
    htj2k_ctx *ctx;
    htj2k_pipe *pipe;
    AVPacket *held;                /* a packet the pipe had no room for: goes in first on the next call */
    int draining;                  /* ff_decode_get_packet() has returned AVERROR_EOF */
    int reduction_factor;          /* private option "lowres", as jpeg2000dec.c:2913-2917 */
    int device, batch, depth;
} Jpeg2000HipPipeContext;

/* enum htj2k_pix_fmt -> AVPixelFormat (the candidate lists of jpeg2000dec.c:170-193), shared with jpeg2000_hip.c */
extern const enum AVPixelFormat ff_jpeg2000_hip_pix_map[HTJ2K_PIX_NB];

static void log_cb(void *opaque, int level, const char *msg) { av_log(opaque, level, "%s", msg); }

static void release_packet(void *opaque)
{
    AVPacket *pkt = opaque;
    av_packet_free(&pkt);
}

static int open_pipe(AVCodecContext *avctx)
{
    Jpeg2000HipPipeContext *s = avctx->priv_data;
    return htj2k_pipe_open(s->ctx, s->batch, s->depth, &s->pipe);
}

static av_cold int hip_pipe_init(AVCodecContext *avctx)
{
    Jpeg2000HipPipeContext *s = avctx->priv_data;
    htj2k_opts o = { 0 };
    int ret;

    if (!s->reduction_factor && avctx->lowres < 34)                           /* jpeg2000dec.c:2811-2817 */
        s->reduction_factor = avctx->lowres;
    if (avctx->lowres != s->reduction_factor && avctx->lowres)
        return AVERROR(EINVAL);
    o.bitexact         = !!(avctx->flags & AV_CODEC_FLAG_BITEXACT);            /* jpeg2000dec.c:543 */
    o.reduction_factor = s->reduction_factor;
    o.max_pixels       = avctx->max_pixels;                                    /* jpeg2000dec.c:224 */
    o.strict           = avctx->strict_std_compliance >= FF_COMPLIANCE_STRICT; /* jpeg2000dec.c:2488 */
    o.device_id        = s->device;
    o.req_pix_fmt      = HTJ2K_PIX_NONE;
    for (int i = 0; i < HTJ2K_PIX_NB; i++)                                     /* jpeg2000dec.c:354 */
        if (ff_jpeg2000_hip_pix_map[i] == avctx->pix_fmt)
            o.req_pix_fmt = i;
    if ((ret = htj2k_open(&o, &s->ctx)) < 0)        /* AVERROR(ENOSYS) without a gfx950 device: no CPU fallback */
        return ret;
    htj2k_set_log(s->ctx, log_cb, avctx);
    return open_pipe(avctx);
}

/* what jpeg2000_read_main_headers() / get_siz() leave on the context (jpeg2000dec.c:213,326,330-420,546,2867) */
static int apply_info(AVCodecContext *avctx, const htj2k_info *info)
{
    int ret;
    avctx->profile = info->profile;
    if ((ret = ff_set_dimensions(avctx, info->width << avctx->lowres, info->height << avctx->lowres)) < 0)
        return ret;
    avctx->pix_fmt = ff_jpeg2000_hip_pix_map[info->pix_fmt];
    avctx->bits_per_raw_sample = info->bits_per_raw_sample;
    if (info->lossless)
        avctx->properties |= FF_CODEC_PROPERTY_LOSSLESS;
    if (info->sar_num && info->sar_den)
        avctx->sample_aspect_ratio = (AVRational){ info->sar_num, info->sar_den };
    return 0;
}

/* feed the pipe until it is full or the input runs dry.  Packets go in by reference: the AVPacket lives until its
 * frame has been handed out (htj2k_pipe_send_ref calls release_packet), nothing is copied on this thread. */
static int feed(AVCodecContext *avctx)
{
    Jpeg2000HipPipeContext *s = avctx->priv_data;
    for (;;) {
        AVPacket *pkt = s->held;
        int ret;
        s->held = NULL;
        if (!pkt) {
            if (s->draining)
                return 0;
            if (!(pkt = av_packet_alloc()))
                return AVERROR(ENOMEM);
            ret = ff_decode_get_packet(avctx, pkt);
            if (ret < 0) {
                av_packet_free(&pkt);
                if (ret == AVERROR_EOF) {
                    s->draining = 1;
                    htj2k_pipe_flush(s->pipe);                /* start the partly filled batch */
                    return 0;
                }
                return ret == AVERROR(EAGAIN) ? 0 : ret;
            }
            if (!pkt->size) {                                 /* an empty flush packet */
                av_packet_free(&pkt);
                continue;
            }
        }
        ret = htj2k_pipe_send_ref(s->pipe, pkt->data, pkt->size, release_packet, pkt);
        if (ret == HTJ2K_ERR_EAGAIN) {                        /* `depth` batches pending: a frame has to go out first */
            s->held = pkt;
            return 0;
        }
        if (ret < 0) {
            av_packet_free(&pkt);
            return ret;
        }
    }
}

static int hip_pipe_receive_frame(AVCodecContext *avctx, AVFrame *frame)
{
    Jpeg2000HipPipeContext *s = avctx->priv_data;
    htj2k_info info;
    htj2k_frame out = { { 0 } };
    int ret;

    if ((ret = feed(avctx)) < 0)
        return ret;
    ret = htj2k_pipe_info(s->pipe, &info);                     /* blocks until the next frame's batch is decoded */
    if (ret == HTJ2K_ERR_EAGAIN) {                             /* nothing in flight */
        if (s->draining)
            return AVERROR_EOF;
        htj2k_pipe_flush(s->pipe);                             /* low-latency callers: do not sit on a partial batch for ever */
        ret = htj2k_pipe_info(s->pipe, &info);
        if (ret == HTJ2K_ERR_EAGAIN)
            return AVERROR(EAGAIN);
    }
    if (ret < 0) {                                             /* this packet failed; the rest of its batch still comes */
        htj2k_pipe_skip(s->pipe);
        return ret;
    }
    if ((ret = apply_info(avctx, &info)) < 0 || (ret = ff_get_buffer(avctx, frame, 0)) < 0) {
        htj2k_pipe_skip(s->pipe);
        return ret;
    }
    /* AVFrame.data[1] of a PAL8 frame is the palette: the library writes its 256 entries there as plane 1
     * (jpeg2000dec.c:2900-2901) */
    for (int p = 0; p < info.nplanes && p < 4; p++) {
        out.data[p] = frame->data[p];
        out.linesize[p] = frame->linesize[p];
    }
    if ((ret = htj2k_pipe_receive(s->pipe, &out)) < 0) {
        av_frame_unref(frame);
        return ret;
    }
    frame->pict_type = AV_PICTURE_TYPE_I;
    frame->flags |= AV_FRAME_FLAG_KEY;
    return 0;
}

/* seeking: everything in flight is dropped (FFCodec.flush, codec_internal.h:240-244) */
static void hip_pipe_flush(AVCodecContext *avctx)
{
    Jpeg2000HipPipeContext *s = avctx->priv_data;
    av_packet_free(&s->held);
    s->draining = 0;
    htj2k_pipe_close(s->pipe);                                 /* waits for the jobs in flight, releases every packet */
    s->pipe = NULL;
    if (open_pipe(avctx) < 0)
        av_log(avctx, AV_LOG_ERROR, "could not restart the decode pipeline\n");
}

static av_cold int hip_pipe_close(AVCodecContext *avctx)
{
    Jpeg2000HipPipeContext *s = avctx->priv_data;
    av_packet_free(&s->held);
    if (s->pipe)
        htj2k_pipe_close(s->pipe);
    htj2k_close(s->ctx);
    s->pipe = NULL;
    s->ctx = NULL;
    return 0;
}

#define OFFSET(x) offsetof(Jpeg2000HipPipeContext, x)
#define VD AV_OPT_FLAG_VIDEO_PARAM | AV_OPT_FLAG_DECODING_PARAM
static const AVOption options[] = {
    { "lowres", "Lower the decoding resolution by a power of two", OFFSET(reduction_factor), AV_OPT_TYPE_INT, { .i64 = 0 }, 0, 33, VD },
    { "device", "HIP device ordinal", OFFSET(device), AV_OPT_TYPE_INT, { .i64 = 0 }, 0, 63, VD },
    { "batch", "frames per device job", OFFSET(batch), AV_OPT_TYPE_INT, { .i64 = 8 }, 1, 256, VD },
    { "depth", "device jobs in flight", OFFSET(depth), AV_OPT_TYPE_INT, { .i64 = 3 }, 1, 16, VD },
    { NULL },
};

static const AVClass jpeg2000_hip_pipe_class = {
    .class_name = "jpeg2000_hip_pipe",
    .item_name  = av_default_item_name,
    .option     = options,
    .version    = LIBAVUTIL_VERSION_INT,
};

const FFCodec ff_jpeg2000_hip_pipe_decoder = {
    .p.name           = "jpeg2000_hip_pipe",
    CODEC_LONG_NAME("JPEG 2000 / HTJ2K (AMD MI355X, HIP, pipelined)"),
    .p.type           = AVMEDIA_TYPE_VIDEO,
    .p.id             = AV_CODEC_ID_JPEG2000,
    .p.capabilities   = AV_CODEC_CAP_DELAY | AV_CODEC_CAP_DR1,
    .priv_data_size   = sizeof(Jpeg2000HipPipeContext),
    .init             = hip_pipe_init,
    .close            = hip_pipe_close,
    .flush            = hip_pipe_flush,
    FF_CODEC_RECEIVE_FRAME_CB(hip_pipe_receive_frame),
    .p.priv_class     = &jpeg2000_hip_pipe_class,
    .p.max_lowres     = 5,
    .caps_internal    = FF_CODEC_CAP_SKIP_FRAME_FILL_PARAM | FF_CODEC_CAP_INIT_CLEANUP,
};

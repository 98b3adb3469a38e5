/*
 * jpeg2000_hip.c -- FFmpeg-side binding of the MI355X HTJ2K decode library.
 *
 * Drop this file into libavcodec/ of the reference tree.  It registers a
 * second decoder for AV_CODEC_ID_JPEG2000, `jpeg2000_hip`, with the same plugin shape as
 * `ff_jpeg2000_decoder` (libavcodec/jpeg2000dec.c:2926-2939): FFCodec.init / FF_CODEC_DECODE_CB /
 * FFCodec.close forward to htj2k_open / htj2k_probe + htj2k_decode / htj2k_close.
 * avcodec_send_packet / avcodec_receive_frame are unchanged.
 *
 * It needs the FFmpeg build tree (config.h), so it is not compiled in this repository; the
 * library itself has no FFmpeg dependency.
 */
#include "libavutil/avassert.h"
#include "libavutil/imgutils.h"
#include "libavutil/opt.h"
#include "libavutil/pixdesc.h"
#include "avcodec.h"
#include "codec_internal.h"
#include "decode.h"
#include "thread.h"

#include <htj2k_amd.h>

typedef struct Jpeg2000HipContext {
    const AVClass *class;
    htj2k_ctx *ctx;
    int reduction_factor;          /* private option "lowres", as jpeg2000dec.c:2913-2917 */
    int device;
} Jpeg2000HipContext;

/* enum htj2k_pix_fmt -> AVPixelFormat: the candidate lists of jpeg2000dec.c:170-193 */
/* (shared with glue/jpeg2000_hip_pipe.c and glue/jpeg2000_hip_hw.c) */
const enum AVPixelFormat ff_jpeg2000_hip_pix_map[HTJ2K_PIX_NB] = {
    [HTJ2K_PIX_PAL8] = AV_PIX_FMT_PAL8, [HTJ2K_PIX_RGB24] = AV_PIX_FMT_RGB24, [HTJ2K_PIX_RGBA] = AV_PIX_FMT_RGBA,
    [HTJ2K_PIX_RGB48] = AV_PIX_FMT_RGB48, [HTJ2K_PIX_RGBA64] = AV_PIX_FMT_RGBA64,
    [HTJ2K_PIX_GRAY8] = AV_PIX_FMT_GRAY8, [HTJ2K_PIX_YA8] = AV_PIX_FMT_GRAY8A, [HTJ2K_PIX_GRAY16] = AV_PIX_FMT_GRAY16,
    [HTJ2K_PIX_YA16] = AV_PIX_FMT_YA16,
    [HTJ2K_PIX_YUV410P] = AV_PIX_FMT_YUV410P, [HTJ2K_PIX_YUV411P] = AV_PIX_FMT_YUV411P, [HTJ2K_PIX_YUVA420P] = AV_PIX_FMT_YUVA420P,
    [HTJ2K_PIX_YUV420P] = AV_PIX_FMT_YUV420P, [HTJ2K_PIX_YUV422P] = AV_PIX_FMT_YUV422P, [HTJ2K_PIX_YUVA422P] = AV_PIX_FMT_YUVA422P,
    [HTJ2K_PIX_YUV440P] = AV_PIX_FMT_YUV440P, [HTJ2K_PIX_YUV444P] = AV_PIX_FMT_YUV444P, [HTJ2K_PIX_YUVA444P] = AV_PIX_FMT_YUVA444P,
    [HTJ2K_PIX_YUV420P9] = AV_PIX_FMT_YUV420P9, [HTJ2K_PIX_YUV422P9] = AV_PIX_FMT_YUV422P9, [HTJ2K_PIX_YUV444P9] = AV_PIX_FMT_YUV444P9,
    [HTJ2K_PIX_YUVA420P9] = AV_PIX_FMT_YUVA420P9, [HTJ2K_PIX_YUVA422P9] = AV_PIX_FMT_YUVA422P9, [HTJ2K_PIX_YUVA444P9] = AV_PIX_FMT_YUVA444P9,
    [HTJ2K_PIX_YUV420P10] = AV_PIX_FMT_YUV420P10, [HTJ2K_PIX_YUV422P10] = AV_PIX_FMT_YUV422P10, [HTJ2K_PIX_YUV444P10] = AV_PIX_FMT_YUV444P10,
    [HTJ2K_PIX_YUVA420P10] = AV_PIX_FMT_YUVA420P10, [HTJ2K_PIX_YUVA422P10] = AV_PIX_FMT_YUVA422P10, [HTJ2K_PIX_YUVA444P10] = AV_PIX_FMT_YUVA444P10,
    [HTJ2K_PIX_YUV420P12] = AV_PIX_FMT_YUV420P12, [HTJ2K_PIX_YUV422P12] = AV_PIX_FMT_YUV422P12, [HTJ2K_PIX_YUV444P12] = AV_PIX_FMT_YUV444P12,
    [HTJ2K_PIX_YUV420P14] = AV_PIX_FMT_YUV420P14, [HTJ2K_PIX_YUV422P14] = AV_PIX_FMT_YUV422P14, [HTJ2K_PIX_YUV444P14] = AV_PIX_FMT_YUV444P14,
    [HTJ2K_PIX_YUV420P16] = AV_PIX_FMT_YUV420P16, [HTJ2K_PIX_YUV422P16] = AV_PIX_FMT_YUV422P16, [HTJ2K_PIX_YUV444P16] = AV_PIX_FMT_YUV444P16,
    [HTJ2K_PIX_YUVA420P16] = AV_PIX_FMT_YUVA420P16, [HTJ2K_PIX_YUVA422P16] = AV_PIX_FMT_YUVA422P16, [HTJ2K_PIX_YUVA444P16] = AV_PIX_FMT_YUVA444P16,
    [HTJ2K_PIX_XYZ12] = AV_PIX_FMT_XYZ12,
};

static int to_htj2k_pix(enum AVPixelFormat f)
{
    for (int i = 0; i < HTJ2K_PIX_NB; i++)
        if (ff_jpeg2000_hip_pix_map[i] == f)
            return i;
    return HTJ2K_PIX_NONE;
}

static void log_cb(void *opaque, int level, const char *msg)
{
    av_log(opaque, level, "%s", msg);      /* levels are AV_LOG_* values already */
}

static av_cold int jpeg2000_hip_init(AVCodecContext *avctx)
{
    Jpeg2000HipContext *s = avctx->priv_data;
    htj2k_opts o = { 0 };
    int ret;

    /* lowres handling of jpeg2000_decode_init(), jpeg2000dec.c:2811-2817 */
    if (!s->reduction_factor && avctx->lowres < 34)
        s->reduction_factor = avctx->lowres;
    if (avctx->lowres != s->reduction_factor && avctx->lowres)
        return AVERROR(EINVAL);

    o.bitexact         = !!(avctx->flags & AV_CODEC_FLAG_BITEXACT);            /* jpeg2000dec.c:543 */
    o.reduction_factor = s->reduction_factor;
    o.max_pixels       = avctx->max_pixels;                                    /* jpeg2000dec.c:224 */
    o.strict           = avctx->strict_std_compliance >= FF_COMPLIANCE_STRICT; /* jpeg2000dec.c:2488 */
    o.device_id        = s->device;
    o.req_pix_fmt      = to_htj2k_pix(avctx->pix_fmt);                         /* jpeg2000dec.c:354 */
    ret = htj2k_open(&o, &s->ctx);          /* fails (ENOSYS) without a gfx950 device: no CPU fallback */
    if (ret < 0)
        return ret;
    htj2k_set_log(s->ctx, log_cb, avctx);
    return 0;
}

static int jpeg2000_hip_decode_frame(AVCodecContext *avctx, AVFrame *picture, int *got_frame, AVPacket *avpkt)
{
    Jpeg2000HipContext *s = avctx->priv_data;
    htj2k_info info;
    htj2k_frame fr = { { 0 } };
    int ret;

    /* what jpeg2000_read_main_headers()/get_siz() set on the context, jpeg2000dec.c:213,326,330-420,546 */
    if ((ret = htj2k_probe(s->ctx, avpkt->data, avpkt->size, &info)) < 0)
        return ret;
    avctx->profile = info.profile;
    if ((ret = ff_set_dimensions(avctx, info.width << avctx->lowres, info.height << avctx->lowres)) < 0)
        return ret;
    avctx->pix_fmt = ff_jpeg2000_hip_pix_map[info.pix_fmt];
    avctx->bits_per_raw_sample = info.bits_per_raw_sample;
    if (info.lossless)
        avctx->properties |= FF_CODEC_PROPERTY_LOSSLESS;
    if (info.sar_num && info.sar_den)
        avctx->sample_aspect_ratio = (AVRational){ info.sar_num, info.sar_den };   /* jpeg2000dec.c:2867 */
    if (avctx->skip_frame >= AVDISCARD_ALL)                                         /* jpeg2000dec.c:2871 */
        return avpkt->size;

    if ((ret = ff_thread_get_buffer(avctx, picture, 0)) < 0)                        /* jpeg2000dec.c:2877 */
        return ret;
    for (int p = 0; p < 4; p++) {
        fr.data[p]     = picture->data[p];
        fr.linesize[p] = picture->linesize[p];
    }
    ret = htj2k_decode(s->ctx, avpkt->data, avpkt->size, &fr, NULL);
    if (ret < 0)
        return ret;
    *got_frame = 1;
    return ret;                                   /* bytes consumed, jpeg2000dec.c:2903 */
}

static av_cold int jpeg2000_hip_close(AVCodecContext *avctx)
{
    Jpeg2000HipContext *s = avctx->priv_data;
    htj2k_close(s->ctx);
    s->ctx = NULL;
    return 0;
}

#define OFFSET(x) offsetof(Jpeg2000HipContext, x)
#define VD AV_OPT_FLAG_VIDEO_PARAM | AV_OPT_FLAG_DECODING_PARAM
static const AVOption options[] = {
    { "lowres", "Lower the decoding resolution by a power of two", OFFSET(reduction_factor), AV_OPT_TYPE_INT, { .i64 = 0 }, 0, 33, VD },
    { "device", "HIP device ordinal", OFFSET(device), AV_OPT_TYPE_INT, { .i64 = 0 }, 0, 63, VD },
    { NULL },
};

static const AVClass jpeg2000_hip_class = {
    .class_name = "jpeg2000_hip",
    .item_name  = av_default_item_name,
    .option     = options,
    .version    = LIBAVUTIL_VERSION_INT,
};

const FFCodec ff_jpeg2000_hip_decoder = {
    .p.name           = "jpeg2000_hip",
    CODEC_LONG_NAME("JPEG 2000 / HTJ2K (AMD MI355X, HIP)"),
    .p.type           = AVMEDIA_TYPE_VIDEO,
    .p.id             = AV_CODEC_ID_JPEG2000,
    /* frames are independent: one context per frame thread, each with its own device streams
     * (jpeg2000dec.c:2931; pthread_frame.c:856-889) */
    .p.capabilities   = AV_CODEC_CAP_FRAME_THREADS | AV_CODEC_CAP_DR1,
    .priv_data_size   = sizeof(Jpeg2000HipContext),
    .init             = jpeg2000_hip_init,
    .close            = jpeg2000_hip_close,
    FF_CODEC_DECODE_CB(jpeg2000_hip_decode_frame),
    .p.priv_class     = &jpeg2000_hip_class,
    .p.max_lowres     = 5,
    .caps_internal    = FF_CODEC_CAP_SKIP_FRAME_FILL_PARAM | FF_CODEC_CAP_INIT_CLEANUP,
};

/*
 * jpeg2000_hip_hw.c -- FFmpeg-side binding of the MI355X HTJ2K decode library, device-resident frames
 * (SURVEY 8(f) rank 2): decoded frames stay in HBM and leave the decoder as AV_PIX_FMT_HIP frames whose data[] are
 * device pointers; a GPU filter chain consumes them there, av_hwframe_transfer_data() downloads one when asked.
 *
 * Two registrations, both on top of htj2k_pipe_receive_device_ref() / htj2k_pipe_release_device():
 *
 *   ff_jpeg2000_hip_hw_decoder   an FFCodec with the decoupled callback FFCodec.cb.receive_frame
 *                                (libavcodec/codec_internal.h:200-208) whose output format is the hardware one,
 *                                the way the *_cuvid / *_qsv wrapper decoders work (cuviddec.c:1100-1140):
 *                                .hw_configs advertises AV_PIX_FMT_HIP on AV_HWDEVICE_TYPE_HIP with
 *                                AV_CODEC_HW_CONFIG_METHOD_HW_DEVICE_CTX | ..._INTERNAL (hwconfig.h:42-53, 68-69).
 *   ff_jpeg2000_hip_hwaccel      the FFHWAccel form (libavcodec/hwaccel_internal.h:34-164, model nvdec_mjpeg.c:30-90)
 *                                for the native jpeg2000 decoder.  jpeg2000dec.c has no hwaccel call sites; the
 *                                four that are needed are listed at the end of this file.  As with MJPEG the whole
 *                                packet is handed to start_frame (the codestream syntax is parsed again inside the
 *                                library: the native decoder's tile tree is of no use to the device).
 *
 * libavutil needs a device type for this, which the tree does not have (libavutil/hwcontext.h:27-41):
 * glue/hwcontext_hip.c is that type (AV_HWDEVICE_TYPE_HIP, AV_PIX_FMT_HIP), modelled on hwcontext_cuda.c.
 *
 * Needs the FFmpeg build tree (config.h): not compiled in this repository.
 */
#include "libavutil/buffer.h"
#include "libavutil/hwcontext.h"
#include "libavutil/hwcontext_hip.h"          /* glue/hwcontext_hip.c: AVHIPDeviceContext { int device; } */
#include "libavutil/opt.h"
#include "libavutil/pixdesc.h"
#include "avcodec.h"
#include "codec_internal.h"
#include "decode.h"
#include "hwaccel_internal.h"
#include "hwconfig.h"

#include <htj2k_amd.h>

extern const enum AVPixelFormat ff_jpeg2000_hip_pix_map[HTJ2K_PIX_NB];       /* glue/jpeg2000_hip.c */

typedef struct HipHwContext {
    const AVClass *class;
    htj2k_ctx *ctx;
    htj2k_pipe *pipe;
    AVPacket *held;
    int draining, reduction_factor, batch, depth;
    AVBufferRef *frames_ref;        /* AVHWFramesContext of the frames handed out: sw_format, width, height */
} HipHwContext;

/* one per frame in the consumer's hands: gives the planes back to the pipe when the last reference goes */
typedef struct HipFrameRef { htj2k_pipe *pipe; uint64_t token; } HipFrameRef;

static void frame_released(void *opaque, uint8_t *data)
{
    HipFrameRef *r = opaque;
    /* any thread; the batch's job may be reused after the last one.  Also after hw_close: a pipe that is closed with
     * frames out stays behind (with its jobs -- the planes -- and a reference on the context) until the last of them
     * is released here (include/htj2k_amd.h, htj2k_pipe_close) */
    htj2k_pipe_release_device(r->pipe, r->token);
    av_free(r);
}

static void log_cb(void *opaque, int level, const char *msg) { av_log(opaque, level, "%s", msg); }
static void release_packet(void *opaque) { AVPacket *p = opaque; av_packet_free(&p); }

/* the AVHWFramesContext describing what comes out: no pool of its own -- the planes belong to the pipeline's jobs */
static int make_frames_ctx(AVCodecContext *avctx, const htj2k_info *info)
{
    HipHwContext *s = avctx->priv_data;
    AVHWFramesContext *fc;
    int ret;
    if (s->frames_ref) {
        fc = (AVHWFramesContext *)s->frames_ref->data;
        if (fc->width == info->width && fc->height == info->height && fc->sw_format == ff_jpeg2000_hip_pix_map[info->pix_fmt])
            return 0;
        av_buffer_unref(&s->frames_ref);
    }
    if (!avctx->hw_device_ctx)
        return AVERROR(EINVAL);
    if (!(s->frames_ref = av_hwframe_ctx_alloc(avctx->hw_device_ctx)))
        return AVERROR(ENOMEM);
    fc = (AVHWFramesContext *)s->frames_ref->data;
    fc->format = AV_PIX_FMT_HIP;
    fc->sw_format = ff_jpeg2000_hip_pix_map[info->pix_fmt];
    fc->width = info->width;
    fc->height = info->height;
    fc->initial_pool_size = 0;                              /* hwcontext_hip.c: frames_init creates no pool then */
    if ((ret = av_hwframe_ctx_init(s->frames_ref)) < 0)
        av_buffer_unref(&s->frames_ref);
    return ret;
}

static av_cold int hw_init(AVCodecContext *avctx)
{
    HipHwContext *s = avctx->priv_data;
    htj2k_opts o = { 0 };
    int ret;
    if (!avctx->hw_device_ctx) {
        av_log(avctx, AV_LOG_ERROR, "jpeg2000_hip_hw needs a HIP device (-init_hw_device hip=...)\n");
        return AVERROR(EINVAL);
    }
    o.device_id        = ((AVHIPDeviceContext *)((AVHWDeviceContext *)avctx->hw_device_ctx->data)->hwctx)->device;
    o.bitexact         = !!(avctx->flags & AV_CODEC_FLAG_BITEXACT);
    o.reduction_factor = s->reduction_factor ? s->reduction_factor : avctx->lowres;
    o.max_pixels       = avctx->max_pixels;
    o.strict           = avctx->strict_std_compliance >= FF_COMPLIANCE_STRICT;
    o.req_pix_fmt      = HTJ2K_PIX_NONE;
    if ((ret = htj2k_open(&o, &s->ctx)) < 0)
        return ret;
    htj2k_set_log(s->ctx, log_cb, avctx);
    avctx->pix_fmt = AV_PIX_FMT_HIP;
    /* depth counts the batches whose frames the consumer may hold on to, on top of the ones in flight */
    return htj2k_pipe_open(s->ctx, s->batch, s->depth + (avctx->extra_hw_frames > 0 ? (avctx->extra_hw_frames + s->batch - 1) / s->batch : 0),
                           &s->pipe);
}

static int hw_receive_frame(AVCodecContext *avctx, AVFrame *frame)
{
    HipHwContext *s = avctx->priv_data;
    htj2k_info info;
    htj2k_frame out = { { 0 } };
    HipFrameRef *ref;
    int ret;

    for (;;) {                                               /* keep the pipeline full (as glue/jpeg2000_hip_pipe.c: feed()) */
        AVPacket *pkt = s->held;
        s->held = NULL;
        if (!pkt) {
            if (s->draining || !(pkt = av_packet_alloc()))
                break;
            ret = ff_decode_get_packet(avctx, pkt);
            if (ret < 0) {
                av_packet_free(&pkt);
                if (ret == AVERROR_EOF) { s->draining = 1; htj2k_pipe_flush(s->pipe); }
                else if (ret != AVERROR(EAGAIN)) return ret;
                break;
            }
        }
        ret = htj2k_pipe_send_ref(s->pipe, pkt->data, pkt->size, release_packet, pkt);
        if (ret == HTJ2K_ERR_EAGAIN) { s->held = pkt; break; }
        if (ret < 0) { av_packet_free(&pkt); return ret; }
    }
    ret = htj2k_pipe_info(s->pipe, &info);
    if (ret == HTJ2K_ERR_EAGAIN)
        return s->draining ? AVERROR_EOF : AVERROR(EAGAIN);
    if (ret < 0) { htj2k_pipe_skip(s->pipe); return ret; }

    avctx->profile = info.profile;
    if ((ret = ff_set_dimensions(avctx, info.width, info.height)) < 0 || (ret = make_frames_ctx(avctx, &info)) < 0) {
        htj2k_pipe_skip(s->pipe);
        return ret;
    }
    avctx->sw_pix_fmt = ff_jpeg2000_hip_pix_map[info.pix_fmt];
    avctx->bits_per_raw_sample = info.bits_per_raw_sample;

    if (!(ref = av_mallocz(sizeof(*ref)))) { htj2k_pipe_skip(s->pipe); return AVERROR(ENOMEM); }
    if ((ret = htj2k_pipe_receive_device_ref(s->pipe, &out, &ref->token)) < 0) { av_free(ref); return ret; }
    ref->pipe = s->pipe;
    frame->buf[0] = av_buffer_create((uint8_t *)out.data[0], 0, frame_released, ref, AV_BUFFER_FLAG_READONLY);
    if (!frame->buf[0]) { htj2k_pipe_release_device(s->pipe, ref->token); av_free(ref); return AVERROR(ENOMEM); }
    frame->hw_frames_ctx = av_buffer_ref(s->frames_ref);
    frame->format = AV_PIX_FMT_HIP;
    frame->width = info.width;
    frame->height = info.height;
    for (int p = 0; p < info.nplanes && p < 4; p++) {         /* device pointers and pitches; PAL8: plane 1 = the palette */
        frame->data[p] = out.data[p];
        frame->linesize[p] = out.linesize[p];
    }
    frame->pict_type = AV_PICTURE_TYPE_I;
    frame->flags |= AV_FRAME_FLAG_KEY;
    return ff_decode_frame_props(avctx, frame);
}

static void hw_flush(AVCodecContext *avctx)
{
    HipHwContext *s = avctx->priv_data;
    av_packet_free(&s->held);
    s->draining = 0;
    /* frames still in the consumer's hands pin their batches; the pipe cannot simply be closed under them.  Drain what
     * is in flight instead: results are dropped, held frames stay valid */
    htj2k_pipe_flush(s->pipe);
    while (htj2k_pipe_skip(s->pipe) >= 0)
        ;
}

static av_cold int hw_close(AVCodecContext *avctx)
{
    HipHwContext *s = avctx->priv_data;
    av_packet_free(&s->held);
    av_buffer_unref(&s->frames_ref);
    /* AV_PIX_FMT_HIP frames may outlive the decoder (ordinary API use: avcodec_free_context with frames still referenced):
     * htj2k_pipe_close defers freeing the jobs those frames' planes belong to, and the context they were made on, until
     * frame_released() has given the last one back */
    if (s->pipe) htj2k_pipe_close(s->pipe);
    htj2k_close(s->ctx);
    return 0;
}

#define OFFSET(x) offsetof(HipHwContext, x)
#define VD AV_OPT_FLAG_VIDEO_PARAM | AV_OPT_FLAG_DECODING_PARAM
static const AVOption options[] = {
    { "lowres", "Lower the decoding resolution by a power of two", OFFSET(reduction_factor), AV_OPT_TYPE_INT, { .i64 = 0 }, 0, 33, VD },
    { "batch", "frames per device job", OFFSET(batch), AV_OPT_TYPE_INT, { .i64 = 8 }, 1, 256, VD },
    { "depth", "device jobs in flight", OFFSET(depth), AV_OPT_TYPE_INT, { .i64 = 3 }, 1, 16, VD },
    { NULL },
};
static const AVClass hip_hw_class = { .class_name = "jpeg2000_hip_hw", .item_name = av_default_item_name, .option = options, .version = LIBAVUTIL_VERSION_INT };

static const AVCodecHWConfigInternal *const hip_hw_configs[] = {
    &(const AVCodecHWConfigInternal) {
        .public = { .pix_fmt = AV_PIX_FMT_HIP,
                    .methods = AV_CODEC_HW_CONFIG_METHOD_HW_DEVICE_CTX | AV_CODEC_HW_CONFIG_METHOD_INTERNAL,
                    .device_type = AV_HWDEVICE_TYPE_HIP },
        .hwaccel = NULL,
    },
    NULL
};

const FFCodec ff_jpeg2000_hip_hw_decoder = {
    .p.name           = "jpeg2000_hip_hw",
    CODEC_LONG_NAME("JPEG 2000 / HTJ2K (AMD MI355X, HIP, frames in device memory)"),
    .p.type           = AVMEDIA_TYPE_VIDEO,
    .p.id             = AV_CODEC_ID_JPEG2000,
    .p.capabilities   = AV_CODEC_CAP_DELAY | AV_CODEC_CAP_HARDWARE | AV_CODEC_CAP_AVOID_PROBING,
    .priv_data_size   = sizeof(HipHwContext),
    .init             = hw_init,
    .close            = hw_close,
    .flush            = hw_flush,
    FF_CODEC_RECEIVE_FRAME_CB(hw_receive_frame),
    .p.priv_class     = &hip_hw_class,
    .p.pix_fmts       = (const enum AVPixelFormat[]){ AV_PIX_FMT_HIP, AV_PIX_FMT_NONE },
    .hw_configs       = hip_hw_configs,
    .p.wrapper_name   = "hip",
    .caps_internal    = FF_CODEC_CAP_INIT_CLEANUP,
};

/* ------------------------------------------------------------------ FFHWAccel form --------------------------------
 * hwaccel_internal.h:34-164.  Synchronous: one packet per start_frame, the frame's device planes are ready at
 * end_frame (one job per AVCodecContext; frame threads give the overlap, pthread_frame.c:856-889). */
typedef struct HipAccelContext { htj2k_ctx *ctx; htj2k_job *job; } HipAccelContext;

static int accel_init(AVCodecContext *avctx)
{
    HipAccelContext *a = avctx->internal->hwaccel_priv_data;
    htj2k_opts o = { 0 };
    o.device_id        = ((AVHIPDeviceContext *)((AVHWDeviceContext *)avctx->hw_device_ctx->data)->hwctx)->device;
    o.bitexact         = !!(avctx->flags & AV_CODEC_FLAG_BITEXACT);
    o.reduction_factor = avctx->lowres;
    o.req_pix_fmt      = HTJ2K_PIX_NONE;
    return htj2k_open(&o, &a->ctx);
}

static int accel_uninit(AVCodecContext *avctx)
{
    HipAccelContext *a = avctx->internal->hwaccel_priv_data;
    if (a->job) htj2k_job_free(a->ctx, a->job);
    htj2k_close(a->ctx);
    return 0;
}

static int accel_start_frame(AVCodecContext *avctx, const uint8_t *buf, uint32_t size)
{
    HipAccelContext *a = avctx->internal->hwaccel_priv_data;
    int ret = htj2k_job_parse(a->ctx, buf, (int)size, &a->job);      /* host: markers + Tier-2 -> descriptors */
    if (ret >= 0) ret = htj2k_job_upload(a->ctx, a->job);            /* H2D of the packet, k_gather */
    if (ret >= 0) ret = htj2k_job_run(a->ctx, a->job);               /* block decode, IDWT, MCT + frame store: asynchronous */
    return ret;
}

static int accel_decode_slice(AVCodecContext *avctx, const uint8_t *buf, uint32_t size) { return 0; }

/* the picture the native decoder got from ff_thread_get_buffer() is an AV_PIX_FMT_HIP frame out of the user's
 * hw_frames_ctx pool (hwcontext_hip.c: hipMalloc'ed planes): copy device to device.  (A zero-copy variant hands the
 * job's own planes out as in hw_receive_frame above, at the price of one job per frame in flight.) */
static int accel_end_frame(AVCodecContext *avctx)
{
    HipAccelContext *a = avctx->internal->hwaccel_priv_data;
    AVFrame *pic = ((Jpeg2000DecoderContext *)avctx->priv_data)->picture;    /* added next to s->avctx, see below */
    htj2k_frame dev;
    int ret = htj2k_job_device_frame(a->ctx, a->job, 0, &dev);                /* waits for the job's stream */
    if (ret < 0) return ret;
    return ff_hip_copy_planes(avctx->hw_frames_ctx, pic, dev.data, dev.linesize);    /* hwcontext_hip.c: hipMemcpy2D D2D */
}

static int accel_frame_params(AVCodecContext *avctx, AVBufferRef *hw_frames_ctx)
{
    AVHWFramesContext *fc = (AVHWFramesContext *)hw_frames_ctx->data;
    fc->format = AV_PIX_FMT_HIP;
    fc->sw_format = avctx->sw_pix_fmt;
    fc->width = avctx->coded_width;
    fc->height = avctx->coded_height;
    fc->initial_pool_size = 2 + avctx->extra_hw_frames;      /* intra only: the frame being decoded + one being consumed */
    return 0;
}

const FFHWAccel ff_jpeg2000_hip_hwaccel = {
    .p.name         = "jpeg2000_hip",
    .p.type         = AVMEDIA_TYPE_VIDEO,
    .p.id           = AV_CODEC_ID_JPEG2000,
    .p.pix_fmt      = AV_PIX_FMT_HIP,
    .start_frame    = accel_start_frame,
    .decode_slice   = accel_decode_slice,
    .end_frame      = accel_end_frame,
    .frame_params   = accel_frame_params,
    .init           = accel_init,
    .uninit         = accel_uninit,
    .priv_data_size = sizeof(HipAccelContext),
};

/* Call sites to add to libavcodec/jpeg2000dec.c for the FFHWAccel form (the file has none today):
 *   1. get_siz(), after the software format is chosen (jpeg2000dec.c:330-420): offer the hardware format,
 *          enum AVPixelFormat fmts[] = { AV_PIX_FMT_HIP, s->avctx->pix_fmt, AV_PIX_FMT_NONE };
 *          s->avctx->sw_pix_fmt = s->avctx->pix_fmt;  s->avctx->pix_fmt = ff_get_format(s->avctx, fmts);
 *      as mjpegdec.c:731-749 does.
 *   2. jpeg2000_decode_frame(), after ff_thread_get_buffer() (:2877): with avctx->hwaccel set,
 *          s->picture = picture;
 *          ret = hwaccel->start_frame(avctx, avpkt->data, avpkt->size);   if (ret >= 0) ret = hwaccel->end_frame(avctx);
 *          *got_frame = ret >= 0;  goto end;
 *      i.e. jpeg2000_read_bitstream_packets() and the execute2() fan-out (:2880-2894) are skipped (mjpegdec.c:804-811, 2555).
 *   3. ff_jpeg2000_decoder (:2926-2939): .hw_configs = (const AVCodecHWConfigInternal *const []) { HWACCEL_HIP(jpeg2000), NULL }
 *      with  #define HWACCEL_HIP(codec) HW_CONFIG_HWACCEL(1, 1, 0, HIP, HIP, ff_ ## codec ## _hip_hwaccel)  in hwconfig.h:68-84.
 *   4. hwaccels.h: extern const struct FFHWAccel ff_jpeg2000_hip_hwaccel;  configure: jpeg2000_hip_hwaccel_deps="libhtj2k_amd".
 */

/*
 * hwcontext_hip.c -- AV_HWDEVICE_TYPE_HIP for libavutil: what glue/jpeg2000_hip_hw.c needs from the hardware-context
 * layer, modelled on libavutil/hwcontext_cuda.c (device_create :440-520, frames_init :160-220, transfer_data :250-330).
 * The HIP runtime is reached through the decode library (htj2k_host_alloc / htj2k_device_to_host) and four runtime
 * calls (hipSetDevice, hipMalloc, hipFree, hipMemcpy2D).
 *
 * Additions elsewhere in libavutil:
 *   hwcontext.h:27-41        AV_HWDEVICE_TYPE_HIP                    (new enumerator, at the end)
 *   pixfmt.h (after :260)    AV_PIX_FMT_HIP                           "data[i] = HIP device pointers, linesize[i] = pitches"
 *   pixdesc.c                [AV_PIX_FMT_HIP] = { .name = "hip", .flags = AV_PIX_FMT_FLAG_HWACCEL }
 *   hwcontext.c:30-70        &ff_hwcontext_type_hip in hw_table[], "hip" in hw_type_names[]
 *   hwcontext_hip.h          typedef struct AVHIPDeviceContext { int device; } AVHIPDeviceContext;
 *                            int ff_hip_copy_planes(AVBufferRef *frames, AVFrame *dst, uint8_t *const src[4], const int pitch[4]);
 *
 * Needs the FFmpeg build tree: not compiled in this repository.
 */
#include <hip/hip_runtime_api.h>

#include "buffer.h"
#include "hwcontext.h"
#include "hwcontext_internal.h"
#include "hwcontext_hip.h"
#include "imgutils.h"
#include "mem.h"
#include "pixdesc.h"

static const enum AVPixelFormat supported_sw[] = {          /* what get_siz() can pick (jpeg2000dec.c:170-193) minus pal8 */
    AV_PIX_FMT_RGB24, AV_PIX_FMT_RGBA, AV_PIX_FMT_RGB48, AV_PIX_FMT_RGBA64, AV_PIX_FMT_GRAY8, AV_PIX_FMT_GRAY16, AV_PIX_FMT_YA8,
    AV_PIX_FMT_YA16, AV_PIX_FMT_YUV420P, AV_PIX_FMT_YUV422P, AV_PIX_FMT_YUV444P, AV_PIX_FMT_YUV420P10, AV_PIX_FMT_YUV422P10,
    AV_PIX_FMT_YUV444P10, AV_PIX_FMT_YUV420P12, AV_PIX_FMT_YUV422P12, AV_PIX_FMT_YUV444P12, AV_PIX_FMT_YUV444P16, AV_PIX_FMT_XYZ12,
};

static int hip_device_create(AVHWDeviceContext *ctx, const char *device, AVDictionary *opts, int flags)
{
    AVHIPDeviceContext *h = ctx->hwctx;
    int n = 0;
    h->device = device ? atoi(device) : 0;
    if (hipGetDeviceCount(&n) != hipSuccess || h->device < 0 || h->device >= n)
        return AVERROR(ENODEV);
    return 0;
}

static int hip_frames_get_constraints(AVHWDeviceContext *ctx, const void *hwconfig, AVHWFramesConstraints *c)
{
    const int n = FF_ARRAY_ELEMS(supported_sw);
    if (!(c->valid_sw_formats = av_malloc_array(n + 1, sizeof(*c->valid_sw_formats))) ||
        !(c->valid_hw_formats = av_malloc_array(2, sizeof(*c->valid_hw_formats))))
        return AVERROR(ENOMEM);
    memcpy(c->valid_sw_formats, supported_sw, sizeof(supported_sw));
    c->valid_sw_formats[n] = AV_PIX_FMT_NONE;
    c->valid_hw_formats[0] = AV_PIX_FMT_HIP;
    c->valid_hw_formats[1] = AV_PIX_FMT_NONE;
    return 0;
}

static void hip_buffer_free(void *opaque, uint8_t *data) { (void)hipFree(data); }

static AVBufferRef *hip_pool_alloc(void *opaque, size_t size)
{
    AVHWFramesContext *fc = opaque;
    void *p = NULL;
    AVBufferRef *ref;
    if (hipSetDevice(((AVHIPDeviceContext *)fc->device_ctx->hwctx)->device) != hipSuccess || hipMalloc(&p, size) != hipSuccess)
        return NULL;
    if (!(ref = av_buffer_create(p, size, hip_buffer_free, fc, 0)))
        (void)hipFree(p);
    return ref;
}

/* initial_pool_size == 0 and no user pool: the frames context only DESCRIBES frames whose memory somebody else owns
 * (the decode pipeline's jobs, jpeg2000_hip_hw.c); otherwise one hipMalloc'ed buffer per frame, planes back to back */
static int hip_frames_init(AVHWFramesContext *fc)
{
    if (!fc->pool && fc->initial_pool_size > 0) {
        const int size = av_image_get_buffer_size(fc->sw_format, fc->width, fc->height, 256);
        if (size < 0)
            return size;
        ffhwframesctx(fc)->pool_internal = av_buffer_pool_init2(size, fc, hip_pool_alloc, NULL);
        if (!ffhwframesctx(fc)->pool_internal)
            return AVERROR(ENOMEM);
    }
    return 0;
}

static int hip_get_buffer(AVHWFramesContext *fc, AVFrame *frame)
{
    int ret;
    if (!fc->pool)
        return AVERROR(ENOSYS);                               /* describe-only context: frames come from the decoder */
    if (!(frame->buf[0] = av_buffer_pool_get(fc->pool)))
        return AVERROR(ENOMEM);
    if ((ret = av_image_fill_arrays(frame->data, frame->linesize, frame->buf[0]->data, fc->sw_format, fc->width, fc->height, 256)) < 0)
        return ret;
    frame->format = AV_PIX_FMT_HIP;
    frame->width = fc->width;
    frame->height = fc->height;
    return 0;
}

static int hip_transfer_get_formats(AVHWFramesContext *fc, enum AVHWFrameTransferDirection dir, enum AVPixelFormat **formats)
{
    enum AVPixelFormat *f = av_malloc_array(2, sizeof(*f));
    if (!f)
        return AVERROR(ENOMEM);
    f[0] = fc->sw_format;
    f[1] = AV_PIX_FMT_NONE;
    *formats = f;
    return 0;
}

static int copy_planes(AVHWFramesContext *fc, uint8_t *const dst[4], const int dst_ls[4], uint8_t *const src[4], const int src_ls[4],
                       enum hipMemcpyKind kind)
{
    const AVPixFmtDescriptor *d = av_pix_fmt_desc_get(fc->sw_format);
    if (hipSetDevice(((AVHIPDeviceContext *)fc->device_ctx->hwctx)->device) != hipSuccess)
        return AVERROR_EXTERNAL;
    for (int p = 0; p < 4 && src[p] && dst[p]; p++) {
        const int h = p == 1 || p == 2 ? AV_CEIL_RSHIFT(fc->height, d->log2_chroma_h) : fc->height;
        const int bytes = av_image_get_linesize(fc->sw_format, fc->width, p);
        if (bytes < 0 || hipMemcpy2D(dst[p], dst_ls[p], src[p], src_ls[p], bytes, h, kind) != hipSuccess)
            return AVERROR_EXTERNAL;
    }
    return 0;
}

static int hip_transfer_data_from(AVHWFramesContext *fc, AVFrame *dst, const AVFrame *src)
{
    return copy_planes(fc, dst->data, dst->linesize, (uint8_t *const *)src->data, src->linesize, hipMemcpyDeviceToHost);
}

static int hip_transfer_data_to(AVHWFramesContext *fc, AVFrame *dst, const AVFrame *src)
{
    return copy_planes(fc, dst->data, dst->linesize, (uint8_t *const *)src->data, src->linesize, hipMemcpyHostToDevice);
}

int ff_hip_copy_planes(AVBufferRef *frames, AVFrame *dst, uint8_t *const src[4], const int pitch[4])
{
    return copy_planes((AVHWFramesContext *)frames->data, dst->data, dst->linesize, src, pitch, hipMemcpyDeviceToDevice);
}

const HWContextType ff_hwcontext_type_hip = {
    .type                   = AV_HWDEVICE_TYPE_HIP,
    .name                   = "HIP",
    .device_hwctx_size      = sizeof(AVHIPDeviceContext),
    .device_create          = hip_device_create,
    .frames_get_constraints = hip_frames_get_constraints,
    .frames_init            = hip_frames_init,
    .frames_get_buffer      = hip_get_buffer,
    .transfer_get_formats   = hip_transfer_get_formats,
    .transfer_data_to       = hip_transfer_data_to,
    .transfer_data_from     = hip_transfer_data_from,
    .pix_fmts               = (const enum AVPixelFormat[]){ AV_PIX_FMT_HIP, AV_PIX_FMT_NONE },
};

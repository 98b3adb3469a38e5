/*
 * htj2k_decode.c -- the C ABI of include/htj2k_amd.h used from plain C, the way an FFmpeg-side glue would
 * (INTEGRATION.md): open, probe, allocate planes, decode, close.
 *
 *   htj2k_decode in.j2c [out.raw]            one codestream / JP2 file -> raw planes (plane after plane, tight rows)
 *   htj2k_decode -p N in.j2c [out.raw]       the same packet N times through the asynchronous pipeline
 *                                            (htj2k_pipe_*), frames written are those of the last round
 *   htj2k_decode -s in.j2k [out.raw]         a sequence of back-to-back codestreams / JP2 files: cut into packets by
 *                                            htj2k_splitter_* (64 KB reads), every frame decoded and written
 *   htj2k_decode -x in.mxf [out.raw]         the JPEG 2000 picture elements of an MXF file (htj2k_mxf_next_essence;
 *                                            first picture track; frame-wrapped, or clip-wrapped through the
 *                                            splitter), decoded and written
 *
 * build: make examples   (cc examples/htj2k_decode.c -Iinclude -Lffmpeg-ht_amd -lhtj2k_amd)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "htj2k_amd.h"

static void log_cb(void *opaque, int level, const char *msg)
{
    (void)opaque;
    if (level <= 24)                                     /* AV_LOG_WARNING and worse */
        fprintf(stderr, "[htj2k %d] %s", level, msg);
}

static double now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + t.tv_nsec * 1e-9;
}

static int alloc_planes(const htj2k_info *info, htj2k_frame *fr)
{
    memset(fr, 0, sizeof(*fr));
    for (int p = 0; p < info->nplanes; p++) {
        fr->linesize[p] = info->plane_width[p] * info->plane_bytes_per_sample[p];
        fr->data[p] = malloc((size_t)fr->linesize[p] * info->plane_height[p]);
        if (!fr->data[p]) return -1;
    }
    return 0;
}

/* -s: the file is a sequence of frames; what av_parser_parse2() + avcodec_send_packet() do in the reference */
static int decode_sequence(const char *in, const char *outname)
{
    FILE *f = fopen(in, "rb"), *o = outname ? fopen(outname, "wb") : NULL;
    if (!f || (outname && !o)) { perror(f ? outname : in); return 2; }
    htj2k_opts opts;
    memset(&opts, 0, sizeof(opts));
    opts.req_pix_fmt = HTJ2K_PIX_NONE;
    htj2k_ctx *ctx = NULL;
    htj2k_splitter *sp = NULL;
    int r = htj2k_open(&opts, &ctx);
    if (r < 0 || (r = htj2k_splitter_open(&sp)) < 0) { fprintf(stderr, "open: %d\n", r); return 1; }
    htj2k_set_log(ctx, log_cb, NULL);
    static uint8_t chunk[65536 + 64];
    int nframes = 0, eof = 0;
    while (!eof) {
        int n = (int)fread(chunk, 1, 65536, f), pos = 0;
        eof = n == 0;                                        /* a final call with size 0 flushes the last frame */
        memset(chunk + n, 0, 64);
        do {
            const uint8_t *frame = NULL;
            int fsize = 0;
            int used = htj2k_splitter_parse(sp, chunk + pos, n - pos, &frame, &fsize);
            if (used < 0) { fprintf(stderr, "htj2k_splitter_parse: %d\n", used); return 1; }
            pos += used;
            if (frame && fsize > 0) {
                htj2k_info info;
                htj2k_frame fr;
                if ((r = htj2k_probe(ctx, frame, fsize, &info)) < 0 || alloc_planes(&info, &fr) < 0 ||
                    (r = htj2k_decode(ctx, frame, fsize, &fr, NULL)) < 0) { fprintf(stderr, "frame %d: %d\n", nframes, r); return 1; }
                for (int p = 0; o && p < info.nplanes; p++)
                    fwrite(fr.data[p], 1, (size_t)fr.linesize[p] * info.plane_height[p], o);
                for (int p = 0; p < 4; p++) free(fr.data[p]);
                printf("frame %d: %d bytes, %dx%d pix_fmt %d\n", nframes++, fsize, info.width, info.height, info.pix_fmt);
            } else if (used == 0) {
                break;
            }
        } while (pos < n);
    }
    if (o) fclose(o);
    fclose(f);
    htj2k_splitter_close(sp);
    htj2k_close(ctx);
    return 0;
}

/* -x: what the reference's mxf demuxer hands to the decoder packet by packet (libavformat/mxfdec.c:4034-4160) */
static int decode_mxf(const char *in, const char *outname)
{
    FILE *f = fopen(in, "rb"), *o = outname ? fopen(outname, "wb") : NULL;
    if (!f || (outname && !o)) { perror(f ? outname : in); return 2; }
    fseek(f, 0, SEEK_END);
    long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *file = calloc(1, (size_t)size + 64), *pkt = NULL;
    if (!file || fread(file, 1, (size_t)size, f) != (size_t)size) { fprintf(stderr, "read failed\n"); return 2; }
    fclose(f);
    htj2k_opts opts;
    memset(&opts, 0, sizeof(opts));
    opts.req_pix_fmt = HTJ2K_PIX_NONE;
    htj2k_ctx *ctx = NULL;
    int r = htj2k_open(&opts, &ctx), nframes = 0;
    if (r < 0) { fprintf(stderr, "htj2k_open: %d\n", r); return 1; }
    htj2k_set_log(ctx, log_cb, NULL);
    size_t pos = 0;
    uint32_t track = 0;
    htj2k_mxf_essence es;
    htj2k_splitter *sp = NULL;
    while ((r = htj2k_mxf_next_essence(file, (size_t)size, &pos, &es)) == 1) {
        if (!track) track = es.track_number;
        if (es.track_number != track) continue;              /* one picture track */
        /* a frame-wrapped element is one packet; a clip-wrapped one holds all codestreams back to back and is cut
         * apart by the splitter (two passes of the loop below: the element, then the flush) */
        const int clip = es.wrapping == HTJ2K_MXF_CLIP_WRAPPED;
        if (clip && !sp && (r = htj2k_splitter_open(&sp)) < 0) return 1;
        size_t off = 0;
        for (int flush = 0; flush < 2; flush++) {
            for (;;) {
                const uint8_t *frame = es.data;
                int fsize = (int)es.size;
                if (clip) {
                    const size_t left = flush ? 0 : es.size - off;
                    const int chunk = left > (1u << 30) ? (1 << 30) : (int)left;
                    int used = htj2k_splitter_parse(sp, es.data + off, chunk, &frame, &fsize);
                    if (used < 0) { fprintf(stderr, "htj2k_splitter_parse: %d\n", used); return 1; }
                    off += (size_t)used;
                    if (!frame || fsize <= 0) { if (used == 0 || flush) break; else continue; }
                }
                htj2k_info info;
                htj2k_frame fr;
                pkt = realloc(pkt, (size_t)fsize + 64);      /* the packet with its input padding, as av_get_packet() */
                if (!pkt) return 1;
                memcpy(pkt, frame, (size_t)fsize);
                memset(pkt + fsize, 0, 64);
                if ((r = htj2k_probe(ctx, pkt, fsize, &info)) < 0 || alloc_planes(&info, &fr) < 0 ||
                    (r = htj2k_decode(ctx, pkt, fsize, &fr, NULL)) < 0) { fprintf(stderr, "frame %d: %d\n", nframes, r); return 1; }
                for (int p = 0; o && p < info.nplanes; p++)
                    fwrite(fr.data[p], 1, (size_t)fr.linesize[p] * info.plane_height[p], o);
                for (int p = 0; p < 4; p++) free(fr.data[p]);
                printf("frame %d: %d bytes (element at %zu), %dx%d pix_fmt %d\n", nframes++, fsize, es.klv_offset, info.width, info.height, info.pix_fmt);
                if (!clip || flush) break;
            }
            if (!clip) break;
        }
    }
    htj2k_splitter_close(sp);
    if (r < 0) { fprintf(stderr, "htj2k_mxf_next_essence: %d\n", r); return 1; }
    if (o) fclose(o);
    free(pkt);
    free(file);
    htj2k_close(ctx);
    return 0;
}

int main(int argc, char **argv)
{
    int npipe = 0, a = 1;
    if (argc > 2 && !strcmp(argv[1], "-x")) return decode_mxf(argv[2], argc > 3 ? argv[3] : NULL);
    if (argc > 2 && !strcmp(argv[1], "-s")) return decode_sequence(argv[2], argc > 3 ? argv[3] : NULL);
    if (argc > 2 && !strcmp(argv[1], "-p")) { npipe = atoi(argv[2]); a = 3; }
    if (argc <= a) { fprintf(stderr, "usage: %s [-p N | -s | -x] in.j2c [out.raw]\n", argv[0]); return 2; }
    FILE *f = fopen(argv[a], "rb");
    if (!f) { perror(argv[a]); return 2; }
    fseek(f, 0, SEEK_END);
    long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *pkt = calloc(1, (size_t)size + 64);                 /* AV_INPUT_BUFFER_PADDING_SIZE */
    if (!pkt || fread(pkt, 1, (size_t)size, f) != (size_t)size) { fprintf(stderr, "read error\n"); return 2; }
    fclose(f);

    htj2k_opts opts;
    memset(&opts, 0, sizeof(opts));
    opts.req_pix_fmt = HTJ2K_PIX_NONE;
    htj2k_ctx *ctx = NULL;
    int r = htj2k_open(&opts, &ctx);
    if (r < 0) { fprintf(stderr, "htj2k_open: %d (no gfx950 device? there is no CPU fallback)\n", r); return 1; }
    htj2k_set_log(ctx, log_cb, NULL);

    htj2k_info info;
    if ((r = htj2k_probe(ctx, pkt, (int)size, &info)) < 0) { fprintf(stderr, "htj2k_probe: %d\n", r); return 1; }
    htj2k_frame fr;
    if (alloc_planes(&info, &fr) < 0) return 1;
    printf("%s: %dx%d pix_fmt %d, %d bits, %d component(s), %s, HT %d, device %s\n", argv[a], info.width, info.height,
           info.pix_fmt, info.bits_per_raw_sample, info.ncomponents, info.lossless ? "lossless" : "lossy", info.is_ht,
           htj2k_device_name(ctx));

    if (!npipe) {
        htj2k_stats st;
        memset(&st, 0, sizeof(st));
        double t0 = now();
        r = htj2k_decode(ctx, pkt, (int)size, &fr, &st);
        double t1 = now();
        if (r < 0) { fprintf(stderr, "htj2k_decode: %d\n", r); return 1; }
        printf("decoded %d bytes in %.2f ms: %d codeblocks (%d rejected), parse %.2f ms, device HT %.3f + IDWT %.3f + pack %.3f ms\n",
               r, (t1 - t0) * 1e3, st.n_codeblocks, st.n_block_errors, st.ms_parse, st.ms_ht, st.ms_idwt, st.ms_pack);
    } else {
        htj2k_pipe *pipe = NULL;
        if ((r = htj2k_pipe_open(ctx, 8, 3, &pipe)) < 0) { fprintf(stderr, "htj2k_pipe_open: %d\n", r); return 1; }
        int sent = 0, got = 0;
        double t0 = now();
        while (got < npipe) {
            while (sent < npipe && (r = htj2k_pipe_send_ref(pipe, pkt, (int)size, NULL, NULL)) == 0) sent++;
            if (r < 0 && r != HTJ2K_ERR_EAGAIN) { fprintf(stderr, "htj2k_pipe_send_ref: %d\n", r); return 1; }
            if (sent == npipe) htj2k_pipe_flush(pipe);
            if ((r = htj2k_pipe_receive(pipe, &fr)) < 0) { fprintf(stderr, "htj2k_pipe_receive: %d\n", r); return 1; }
            got++;
        }
        double t1 = now();
        printf("pipeline: %d frames in %.1f ms = %.1f Mpixel/s\n", got, (t1 - t0) * 1e3,
               (double)got * info.width * info.height / (t1 - t0) / 1e6);
        htj2k_pipe_close(pipe);
    }
    if (argc > a + 1) {
        FILE *o = fopen(argv[a + 1], "wb");
        if (!o) { perror(argv[a + 1]); return 2; }
        for (int p = 0; p < info.nplanes; p++)
            fwrite(fr.data[p], 1, (size_t)fr.linesize[p] * info.plane_height[p], o);
        fclose(o);
    }
    for (int p = 0; p < 4; p++) free(fr.data[p]);
    free(pkt);
    htj2k_close(ctx);
    return 0;
}

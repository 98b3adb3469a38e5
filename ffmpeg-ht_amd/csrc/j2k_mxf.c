/*
 * j2k_mxf.c -- pulling JPEG 2000 picture essence out of an MXF file held in memory ("MXF J2K
 * essence in the harness").  This is the KLV layer of the reference's demuxer and nothing more:
 *
 *   next_klv                  klv_read_packet() + mxf_read_sync_klv() + klv_decode_ber_length()
 *                                                                      libavformat/mxfdec.c:432-504
 *   htj2k_mxf_next_essence    the essence branch of mxf_read_packet()  libavformat/mxfdec.c:4034-4160
 *                             element key of SMPTE 422M essence        libavformat/mxfenc.c:216-217
 *                             frame / clip wrapping (J2KWrap)          libavformat/mxfdec.c:1617, 1752-1760
 *
 * Header metadata (tracks, descriptors, index tables) is not read: elements are recognised by their key -- the
 * generic-container essence prefix, item type 0x15 (GC picture), element type 0x08 (frame-wrapped JPEG 2000) or
 * 0x09 (clip-wrapped) -- and the track number (key bytes 12..15) is handed to the caller for stream selection.
 * A frame-wrapped element is one packet; a clip-wrapped element holds all codestreams back to back and goes
 * through htj2k_splitter_*.  Encrypted triplets (mxfdec.c:4054) are skipped.
 */
#include <string.h>
#include "../../include/htj2k_amd.h"

static const uint8_t essence_prefix[12] = { 0x06, 0x0e, 0x2b, 0x34, 0x01, 0x02, 0x01, 0x01, 0x0d, 0x01, 0x03, 0x01 };

/* one KLV triplet from *pos on: resynchronises on 06 0E 2B 34 like the reference, so run-in bytes and damaged
 * stretches are stepped over.  1 = key / value / length filled in, 0 = end of buffer, <0 = error */
static int next_klv(const uint8_t *buf, size_t size, size_t *pos, const uint8_t **key, size_t *voff, uint64_t *vlen)
{
    size_t p = *pos, lenpos;
    uint64_t len;
    while (p + 4 <= size && !(buf[p] == 0x06 && buf[p + 1] == 0x0e && buf[p + 2] == 0x2b && buf[p + 3] == 0x34))
        p++;
    if (p + 17 > size) {                 /* no room for a key and one length byte */
        *pos = size;
        return 0;
    }
    *key = buf + p;
    lenpos = p + 16;
    len = buf[lenpos++];
    if (len & 0x80) {                    /* BER long form: at most 8 length bytes (SMPTE 379M 5.3.4) */
        int n = (int)(len & 0x7f);
        if (n > 8)
            return HTJ2K_ERR_INVALIDDATA;
        if (lenpos + (size_t)n > size) {
            *pos = size;
            return 0;
        }
        len = 0;
        while (n--)
            len = len << 8 | buf[lenpos++];
        if (len > (uint64_t)INT64_MAX)
            return HTJ2K_ERR_INVALIDDATA;
    }
    *voff = lenpos;
    *vlen = len;
    return 1;
}

int htj2k_mxf_next_essence(const uint8_t *buf, size_t size, size_t *pos, htj2k_mxf_essence *out)
{
    if (!buf || !pos || !out || *pos > size)
        return HTJ2K_ERR_EINVAL;
    memset(out, 0, sizeof(*out));
    for (;;) {
        const uint8_t *key = NULL;
        size_t voff = 0;
        uint64_t vlen = 0, avail;
        int r = next_klv(buf, size, pos, &key, &voff, &vlen);
        if (r <= 0)
            return r;
        avail = (uint64_t)(size - voff);
        *pos = vlen < avail ? voff + (size_t)vlen : size;
        if (memcmp(key, essence_prefix, sizeof(essence_prefix)) || key[12] != 0x15 || (key[14] != 0x08 && key[14] != 0x09))
            continue;                    /* partition packs, primer, metadata sets, fill, index, sound, data ... */
        out->data = buf + voff;
        out->size = vlen < avail ? (size_t)vlen : (size_t)avail;       /* a truncated file yields a short packet */
        out->klv_offset = (size_t)(key - buf);
        out->track_number = (uint32_t)key[12] << 24 | (uint32_t)key[13] << 16 | (uint32_t)key[14] << 8 | key[15];
        out->wrapping = key[14] == 0x09 ? HTJ2K_MXF_CLIP_WRAPPED : HTJ2K_MXF_FRAME_WRAPPED;
        return 1;
    }
}

/*
 * j2k_split.c -- cutting a byte stream of back-to-back JPEG 2000 frames (raw codestreams or JP2 files) into
 * packets: what the reference's AVCodecParser for this codec does for av_parser_parse2()
 * (libavcodec/jpeg2000_parser.c:92-211 with ff_combine_frame, libavcodec/parser.c:203-288).
 *
 * The reference finds frame ends with a byte-at-a-time scanner over the last eight bytes.  This one walks the
 * syntax instead: inside a codestream it goes from marker to marker (a marker segment is passed over by its length
 * field, a tile-part by Psot, so that nothing inside them is ever looked at), inside a JP2 file from box to box
 * (every box by its LBox / XLBox; the contiguous-codestream box is walked as a codestream).  The cuts are the
 * reference's:
 *   a raw codestream ends behind its EOC;
 *   a JP2 file ends where the next signature box begins, or where a bare SOC follows its codestream.
 * Input may arrive in pieces of any size; whatever is needed across a piece boundary (a marker, a length field,
 * a box header: at most 16 bytes) is held in the splitter.
 */
#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/htj2k_amd.h"

#define INPUT_PAD 64                    /* zero bytes behind a frame that is handed out from the splitter's own buffer */

enum {
    AT_GAP,                             /* between frames: looking for SOC or a JP2 signature box */
    AT_MARKER,                          /* codestream: the two bytes of a marker code come next */
    AT_SEG_LEN,                         /* ... the length field of a marker segment */
    AT_SOT,                             /* ... Lsot Isot Psot TPsot TNsot */
    AT_FIND_EOC,                        /* ... packet data of unknown length: the next FF D9 ends the codestream */
    AT_BOX,                             /* JP2 file: LBox TBox come next */
    AT_BOX_XL,                          /* ... XLBox */
    AT_BOX_SIG,                         /* ... the four content bytes of what may be the next file's signature box */
    AT_PASS                             /* `pass` bytes to step over, then `then` */
};
enum { K_NONE, K_RAW, K_JP2 };

static const uint8_t signature[12] = { 0, 0, 0, 12, 'j', 'P', ' ', ' ', 0x0D, 0x0A, 0x87, 0x0A };

struct htj2k_splitter {
    int at, then, kind;
    uint64_t pass;
    uint8_t held[16];
    int nheld;
    uint64_t pos;                       /* bytes of the current frame behind the scanner */
    uint64_t box_end;                   /* JP2: where the codestream box ends (0 = not known: it is the file's last box) */
    int in_box;                         /* JP2: walking the codestream of the jp2c box */
    int codestream_done;                /* JP2: its EOC has gone by */
    int seen_ff;                        /* AT_FIND_EOC: the previous byte was FF */
    /* frame assembly */
    uint8_t *acc;
    size_t acc_len, acc_cap;
    int handed_out;                     /* the last call returned `acc`: start afresh on the next one */
    uint8_t carry[12];
    int ncarry;                         /* first bytes of the next frame that arrived in front of the cut's detection */
};

int htj2k_splitter_open(htj2k_splitter **out)
{
    htj2k_splitter *s;
    if (!out)
        return HTJ2K_ERR_EINVAL;
    s = (htj2k_splitter *)calloc(1, sizeof *s);
    if (!s)
        return HTJ2K_ERR_ENOMEM;
    *out = s;
    return 0;
}

void htj2k_splitter_close(htj2k_splitter *s)
{
    if (!s)
        return;
    free(s->acc);
    free(s);
}

/* back between frames; `known` bytes of the frame that follows have been seen already (a cut detected late) */
static void to_gap(htj2k_splitter *s, const uint8_t *seen, int known)
{
    s->at = AT_GAP;
    s->kind = K_NONE;
    s->nheld = known;
    if (known)
        memcpy(s->held, seen, (size_t)known);
    s->pos = (uint64_t)known;
    s->pass = 0;
    s->box_end = 0;
    s->in_box = s->codestream_done = s->seen_ff = 0;
}

static void pass_over(htj2k_splitter *s, uint64_t n, int then)
{
    s->nheld = 0;
    if (n) {
        s->pass = n;
        s->then = then;
        s->at = AT_PASS;
    } else {
        s->at = then;
    }
}

/* the codestream of a JP2 file is over (EOC, or its box ran out): on to the boxes behind it */
static void leave_box_codestream(htj2k_splitter *s)
{
    s->codestream_done = 1;
    s->in_box = 0;
    pass_over(s, s->box_end > s->pos ? s->box_end - s->pos : 0, AT_BOX);
}

int htj2k_splitter_find_end(htj2k_splitter *s, const uint8_t *buf, int size)
{
    int i = 0;
    if (!s || (!buf && size) || size < 0)
        return HTJ2K_ERR_EINVAL;
    if (!size)
        return 0;                                          /* as find_frame_end (jpeg2000_parser.c:100-102) */
    while (i < size) {
        const uint8_t b = buf[i];
        switch (s->at) {
        case AT_PASS: {
            const uint64_t room = (uint64_t)(size - i), n = s->pass < room ? s->pass : room;
            i += (int)n;
            s->pos += n;
            s->pass -= n;
            if (!s->pass)
                s->at = s->then;
            continue;
        }
        case AT_FIND_EOC: {
            /* packet data holds no FF D9 (a byte behind FF stays below 0x90 there), so the first one is EOC */
            if (s->seen_ff && b == 0xD9) {
                i++; s->pos++;
                s->seen_ff = 0;
                if (s->kind == K_RAW) {
                    to_gap(s, NULL, 0);
                    return i;
                }
                leave_box_codestream(s);
                continue;
            }
            if (b != 0xFF) {
                const uint8_t *ff = (const uint8_t *)memchr(buf + i, 0xFF, (size_t)(size - i));
                const int n = ff ? (int)(ff - (buf + i)) : size - i;
                s->seen_ff = 0;
                i += n; s->pos += (uint64_t)n;
                continue;
            }
            s->seen_ff = 1;
            i++; s->pos++;
            continue;
        }
        default:
            break;
        }
        /* the remaining states collect a few bytes first */
        s->held[s->nheld++] = b;
        i++; s->pos++;
        switch (s->at) {
        case AT_GAP:
            /* anything may sit between frames; a frame starts at SOC or at a signature box */
            if (s->nheld >= 2 && s->held[s->nheld - 2] == 0xFF && s->held[s->nheld - 1] == 0x4F) {
                s->kind = K_RAW;
                s->nheld = 0;
                s->at = AT_MARKER;
            } else if (s->nheld == 12 && !memcmp(s->held, signature, 12)) {
                s->kind = K_JP2;
                s->nheld = 0;
                s->at = AT_BOX;
            } else if (s->nheld == 12) {
                memmove(s->held, s->held + 1, 11);
                s->nheld = 11;
            }
            break;
        case AT_MARKER: {
            uint32_t code;
            if (s->nheld < 2)
                break;
            code = (uint32_t)s->held[0] << 8 | s->held[1];
            s->nheld = 0;
            if (code == 0xFFD9) {                          /* EOC */
                if (s->kind == K_RAW) {
                    to_gap(s, NULL, 0);
                    return i;
                }
                leave_box_codestream(s);
            } else if (code == 0xFF90) {                   /* SOT */
                s->at = AT_SOT;
            } else if (code == 0xFF93) {                   /* SOD of a tile-part whose length is not known */
                s->at = AT_FIND_EOC;
            } else if (code == 0xFF4F || code == 0xFF92 || (code >= 0xFF30 && code <= 0xFF3F)) {
                /* SOC, EPH, reserved markers: no segment */
            } else if (code < 0xFF00) {
                s->at = AT_FIND_EOC;                       /* not a marker: lost in damaged data, EOC is the only landmark left */
            } else {
                s->at = AT_SEG_LEN;
            }
            break;
        }
        case AT_SEG_LEN:
            if (s->nheld == 2) {
                const uint32_t len = (uint32_t)s->held[0] << 8 | s->held[1];
                pass_over(s, len > 2 ? len - 2 : 0, AT_MARKER);
            }
            break;
        case AT_SOT:
            if (s->nheld == 10) {
                const uint32_t psot = (uint32_t)s->held[4] << 24 | (uint32_t)s->held[5] << 16 | (uint32_t)s->held[6] << 8 | s->held[7];
                if (psot >= 14) {
                    pass_over(s, psot - 12, AT_MARKER);    /* Psot counts from the SOT marker: 12 bytes of it are behind us */
                } else {
                    s->nheld = 0;                          /* 0: up to EOC; walk the tile-part header to its SOD first */
                    s->at = AT_MARKER;
                }
            }
            break;
        case AT_BOX:
            if (s->nheld == 2 && s->codestream_done && s->held[0] == 0xFF && s->held[1] == 0x4F) {
                const int cut = i - 2;                     /* a bare codestream follows the file */
                to_gap(s, s->held, cut < 0 ? -cut : 0);    /* (the caller goes on from the cut, or from buf[0] if it lies in front) */
                return cut;
            }
            if (s->nheld == 8) {
                const uint32_t lbox = (uint32_t)s->held[0] << 24 | (uint32_t)s->held[1] << 16 | (uint32_t)s->held[2] << 8 | s->held[3];
                const int is_codestream = !memcmp(s->held + 4, "jp2c", 4);
                if (lbox == 12 && !memcmp(s->held, signature, 8)) {
                    s->at = AT_BOX_SIG;
                } else if (lbox == 1) {
                    s->at = AT_BOX_XL;
                } else if (is_codestream) {
                    s->box_end = lbox >= 8 ? s->pos + lbox - 8 : 0;
                    s->in_box = 1;
                    s->nheld = 0;
                    s->at = AT_MARKER;
                } else if (lbox == 0) {
                    pass_over(s, UINT64_MAX, AT_BOX);      /* a box that runs to the end of the file */
                } else {
                    pass_over(s, lbox > 8 ? lbox - 8 : 0, AT_BOX);
                }
            }
            break;
        case AT_BOX_XL:
            if (s->nheld == 16) {
                uint64_t xl = 0;
                int k;
                for (k = 8; k < 16; k++)
                    xl = xl << 8 | s->held[k];
                if (!memcmp(s->held + 4, "jp2c", 4)) {
                    s->box_end = xl >= 16 ? s->pos + xl - 16 : 0;
                    s->in_box = 1;
                    s->nheld = 0;
                    s->at = AT_MARKER;
                } else {
                    pass_over(s, xl > 16 ? xl - 16 : 0, AT_BOX);
                }
            }
            break;
        case AT_BOX_SIG:
            if (s->nheld == 12) {
                if (!memcmp(s->held, signature, 12)) {     /* the next file */
                    const int cut = i - 12;
                    to_gap(s, signature, cut < 0 ? -cut : 0);
                    return cut;
                }
                pass_over(s, 0, AT_BOX);                   /* some other 12-byte box */
            }
            break;
        }
    }
    return HTJ2K_SPLIT_END_NOT_FOUND;
}

static int acc_room(htj2k_splitter *s, size_t need)
{
    if (need > s->acc_cap) {
        size_t cap = s->acc_cap ? s->acc_cap : 65536;
        uint8_t *nb;
        while (cap < need)
            cap *= 2;
        nb = (uint8_t *)realloc(s->acc, cap);
        if (!nb)
            return HTJ2K_ERR_ENOMEM;
        s->acc = nb;
        s->acc_cap = cap;
    }
    return 0;
}

int htj2k_splitter_parse(htj2k_splitter *s, const uint8_t *buf, int size, const uint8_t **frame, int *frame_size)
{
    int cut, r;
    if (!s || !frame || !frame_size || (!buf && size) || size < 0)
        return HTJ2K_ERR_EINVAL;
    *frame = NULL;
    *frame_size = 0;
    if (s->handed_out) {                                   /* the frame returned last time is the caller's no longer */
        s->handed_out = 0;
        s->acc_len = 0;
        if (s->ncarry) {
            if ((r = acc_room(s, (size_t)s->ncarry)) < 0)
                return r;
            memcpy(s->acc, s->carry, (size_t)s->ncarry);
            s->acc_len = (size_t)s->ncarry;
            s->ncarry = 0;
        }
    }
    if (!size) {                                           /* end of the input: what has been collected is the last frame */
        if (!s->acc_len)
            return 0;
        if ((r = acc_room(s, s->acc_len + INPUT_PAD)) < 0)
            return r;
        memset(s->acc + s->acc_len, 0, INPUT_PAD);
        *frame = s->acc;
        *frame_size = (int)s->acc_len;
        s->handed_out = 1;
        to_gap(s, NULL, 0);
        return 0;
    }
    cut = htj2k_splitter_find_end(s, buf, size);
    if (cut == HTJ2K_SPLIT_END_NOT_FOUND) {
        if ((r = acc_room(s, s->acc_len + (size_t)size + INPUT_PAD)) < 0)
            return r;
        memcpy(s->acc + s->acc_len, buf, (size_t)size);
        s->acc_len += (size_t)size;
        return size;
    }
    if (cut >= 0 && !s->acc_len) {                         /* the whole frame lies in the caller's buffer */
        *frame = buf;
        *frame_size = cut;
        return cut;
    }
    if (cut >= 0) {
        if ((r = acc_room(s, s->acc_len + (size_t)cut + INPUT_PAD)) < 0)
            return r;
        memcpy(s->acc + s->acc_len, buf, (size_t)cut);
        s->acc_len += (size_t)cut;
    } else {
        /* the cut lies in earlier input: the last -cut bytes collected open the next frame; this call consumes
         * nothing, the scanner holds those bytes and takes buf again from its start */
        if ((size_t)-cut > s->acc_len)
            return HTJ2K_ERR_BUG;
        s->ncarry = -cut;
        memcpy(s->carry, s->acc + s->acc_len + cut, (size_t)-cut);
        s->acc_len -= (size_t)-cut;
        if ((r = acc_room(s, s->acc_len + INPUT_PAD)) < 0)
            return r;
    }
    memset(s->acc + s->acc_len, 0, INPUT_PAD);
    *frame = s->acc;
    *frame_size = (int)s->acc_len;
    s->handed_out = 1;
    return cut > 0 ? cut : 0;
}

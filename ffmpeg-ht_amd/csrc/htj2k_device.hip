/*
 * htj2k_device.hip -- the C ABI of include/htj2k_amd.h: device management, the frame
 * pipeline (parse -> H2D -> HT decode -> IDWT -> MCT/pack -> D2H) and the kernel launches.
 *
 * Mirrors the plugin surface of `ff_jpeg2000_decoder` (libavcodec/jpeg2000dec.c:2926-2939):
 * htj2k_open = FFCodec.init, htj2k_decode = FFCodec.cb.decode (jpeg2000_decode_frame,
 * :2825-2908), htj2k_close = FFCodec.close.  There is no CPU fallback anywhere in this
 * library: without a usable HIP device htj2k_open fails with HTJ2K_ERR_ENOSYS.
 */
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/htj2k_amd.h"
#include "j2k_plan.h"
#include "ht_cxtvlc_rows.h"
#include "ht_kernels.hpp"
#include "dwt_kernels.hpp"
#include "pack_kernels.hpp"
#include "dwt_stream.hpp"
#include "mq_kernels.hpp"

using namespace htj2k;

/* ------------------------------------------------------------------ k_gather
 * Puts the byte pool together on the device from the uploaded packets: the host parser only emits the gather table
 * (J2kSeg, j2k_plan.h) and never touches a code-block byte.  One wavefront per code-block walks the block's pieces in
 * order; a piece goes from an arbitrary byte offset of the packet to its place in the 16-byte aligned region of the
 * block -- single bytes up to the first aligned destination dword, then a dword per lane from two aligned source
 * dwords (v_alignbyte), single bytes at the end -- and a terminated Part-1 segment gets its 0xFF 0xFF.  The pool was
 * cleared before, so the pads stay zero.  (jpeg2000dec.c:1508-1516 is the host loop this replaces in the reference:
 * bytestream2_get_bufferu into cblk->data, packet by packet.) */
__global__ void __launch_bounds__(256) k_gather(const J2kSeg *__restrict__ segs, const uint32_t *__restrict__ first_seg, int ngroups,
                                                 const uint8_t *__restrict__ pkt, const uint8_t *__restrict__ lit,
                                                 uint8_t *__restrict__ pool)
{
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= ngroups) return;
    const uint32_t s1 = first_seg[g + 1];
    for (uint32_t s = first_seg[g]; s < s1; s++) {
        const J2kSeg sg = segs[s];
        const uint8_t *src = ((sg.flags & J2K_SEG_LIT) ? lit : pkt) + sg.src;
        uint8_t *dst = pool + sg.dst;
        const uint32_t head = min(sg.len, (uint32_t)(-(intptr_t)dst & 3));
        if ((uint32_t)lane < head) dst[lane] = src[lane];
        const uint32_t body = (sg.len - head) >> 2, tail0 = head + 4 * body;
        const uint8_t *bs = src + head;
        const uint32_t sh = (uint32_t)((uintptr_t)bs & 3);
        const uint32_t *al = (const uint32_t *)(bs - sh);
        uint32_t *out = (uint32_t *)(dst + head);
        for (uint32_t k = lane; k < body; k += 64) {
            const uint32_t lo = al[k], hi = sh ? al[k + 1] : 0u;
            out[k] = __builtin_amdgcn_alignbyte(hi, lo, sh);
        }
        if ((uint32_t)lane < sg.len - tail0) dst[tail0 + lane] = src[tail0 + lane];
        if ((sg.flags & J2K_SEG_TERM) && lane < 2) dst[sg.len + lane] = 0xFF;
    }
}

#define LOG_ERROR 16
#define LOG_WARNING 24
#define LOG_INFO 32

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t n)
    {
        if (n <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = n + n / 8 + 256;
        if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return HTJ2K_ERR_ENOMEM; }
        cap = want;
        /* test aid: fresh device memory is not zero in a long-lived process; HTJ2K_POISON=1 makes every new buffer
         * start as 0xA5 bytes so that a read of memory no kernel wrote shows up at once (tools/gpu_random_configs.py) */
        static const bool poison = getenv("HTJ2K_POISON") && atoi(getenv("HTJ2K_POISON"));
        if (poison) {                                     /* (the fill runs on the null stream and need not have happened when
                                                            * hipMemset returns; the jobs' streams do not wait for that stream) */
            (void)hipMemset(p, 0xA5, want);
            (void)hipDeviceSynchronize();
        }
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct HostBuf {                       /* pinned host memory (async H2D at PCIe rate), grow-only */
    void *p = nullptr;
    size_t cap = 0;
    int device = 0;
    void *ensure(size_t n)
    {
        if (n <= cap) return p;
        /* called from the parse threads: bind the thread to the device only when something has to be
         * allocated (the first HIP call of a fresh thread costs milliseconds) */
        if (hipSetDevice(device) != hipSuccess) return nullptr;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        const size_t want = n + n / 4 + 4096;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; return nullptr; }
        cap = want;
        return p;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

struct htj2k_ctx {
    htj2k_opts opts;
    int device = 0;
    char devname[256];
    htj2k_log_fn log = nullptr;
    void *log_opaque = nullptr;
    uint16_t *d_tables = nullptr;      /* 2 x 1024 CxtVLC decode entries */
    J2kParser *probe_parser = nullptr;
    htj2k_job *own_job = nullptr;      /* used by htj2k_decode */
    int idwt_mode = 3;                 /* 0 = generic two-pass kernels, 1 = LDS tile kernel, 3 = register-streaming kernel (dwt_stream.hpp) */
    int fuse_pack = 1;                 /* idwt_mode 3, IDWT and pack stages run in one call: the final level writes the frame */
    int idwt_x3 = 1;                   /* 1: jobs with 16-bit LL bands run the first three 5/3 levels as one launch (k_idwt_stream_ll16_x3) */
    int ht_pair = 1;                   /* 1: jobs with 16-bit sub-bands use k_ht_decode_pair (two blocks per wave, a lane per quad) */
    int ht_multi = 1;                  /* 1: jobs with 32-bit sub-bands whose HT blocks qualify use k_ht_decode_multi (2 or 4 blocks per wave) */
    int idwt_pk = 1;                   /* 1: fused final 5/3 levels of 8-bit pictures on pairs of 16-bit samples where the bounds allow (pk16_bounds) */
    int ll16_test_bits = 16;           /* tests: an LL sample "overflows" when it does not fit this many bits */
    int ll16 = 1;                      /* 1: such jobs also keep the LL bands between the IDWT levels as int16_t (overflow is detected
                                        * on the device and the transform run again with 32-bit LL bands, job_settle) */
    int coef16 = 1;                    /* 1: reversible jobs whose coefficients fit 16 bits keep them as int16_t between the
                                        * block decoder and the IDWT (see coef16_ok) */
    int ht_mode = 1;                   /* 0 = one kernel (serial stage on lane 0), 1 = k_ht_vlc (lane per block) + k_ht_decode<true> */
    int max_dyn_lds = 64 * 1024;
    int parse_threads = 0;             /* host threads that parse the frames of a batch; 0 = min(cores, 16) */
    int packet_threads = 1;            /* > 1: a frame parsed on its own (htj2k_decode, batches of one) has the packets of tiles with a
                                        * PLT list read by this many threads (j2k_parser_set_packet_threads); off by default:
                                        * at 0.4 ms per 4K frame waking the threads costs what they save */
    int device_gather = 1;             /* 1: the packets are uploaded as they are and k_gather puts the byte pool together on the
                                        * device (the parser does not touch code-block bytes); 0: the parser gathers on the host */
    std::mutex log_mutex;
    std::atomic<int> refs{1};          /* the caller's, + one per pipe that is still open or has device frames out (htj2k_pipe.cpp) */
};

struct LevelLaunch {                   /* one IDWT launch: all planes (of all frames) of one type that have this level */
    int type, level;
    int count;                         /* planes */
    size_t table_off;                  /* byte offset of its DwtLevel / DwtTileArgs table in d_desc */
    int max_lh, max_lv, min_l;         /* min_l: smallest line length of any plane (1-sample lines need k_idwt_tile) */
    double alg_bytes;                  /* sum over its planes of 2 * 4 * lh * lv */
    double hbm_bytes = 0;              /* least HBM traffic: alg_bytes, or for a fused final level 4 * lh * lv read +
                                        * the frame bytes written */
    int nc = 0;                        /* 0: table of DwtTileArgs; 1/3/4: table of DwtFusedArgs (fused final level) */
    int outk = -1;                     /* fused level: the fast store all its entries qualify for (stream_fast_outk), -1: general path */
    bool all_fast = false;             /* every entry qualifies for the streaming kernels' fast path */
    int pk_bits = 0;                   /* fused 5/3 level with an 8-bit fast store: the packed 16-bit kernel is exact when its LL input fits
                                        * this many bits (pk16_bounds); 0: not at all */
};

struct FrameSlot {                     /* one frame of a batch */
    J2kParser *parser = nullptr;
    const J2kPlan *plan = nullptr;
    HostBuf h_pkt;                     /* device gather: the packet staged in pinned memory (the caller's is only borrowed for the call) */
    const uint8_t *h2d_src = nullptr;  /* ... or the caller's packet itself when that is page-locked (htj2k_job_parse_batch_ex) */
    bool h2d_own = false;              /* h2d_src is h_pkt (64 bytes of zero padding behind the packet) */
    size_t pkt_base = 0;               /* ... and where it goes in d_pkt */
    HostBuf h_bytes;                   /* host gather: the parser gathers the codeblock bytes straight into pinned memory */
    float ms_stage = 0, ms_parse = 0;  /* host time of the last parse_batch: staging copy, parser */
    uint32_t block_base = 0, tc_base = 0, sample_base = 0;
    size_t bytes_base = 0;
    DevBuf d_out[4];
    OutPlanes out;
};

struct htj2k_job {
    std::vector<FrameSlot> frames;
    int nframes = 0;
    /* merged (re-based) tables of the whole batch */
    std::vector<J2kBlock> blocks;
    std::vector<J2kTileComp> tilecomps;
    std::vector<int> tc_frame;         /* frame index of each merged tile-component */
    size_t nbytes = 0, nsamples = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    std::vector<hipEvent_t> lev_ev;    /* per-IDWT-launch brackets (roofline measurement) */
    int lev_ev_used = 0;
    std::vector<double> lev_bytes, lev_hbm;   /* algorithmic / least-HBM bytes of each recorded launch */
    DevBuf d_bytes, d_blocks, d_status, d_coef, d_t0, d_t1, d_desc, d_qsym, d_qoff, d_vlcu, d_melu, d_reflist, d_roff, d_refbits;
    /* Part-1 (MQ-coded) blocks sit behind the HT blocks in `blocks`: [nht, blocks.size()) */
    /* every sample the block decoder writes fits int16_t and only the FASTONLY streaming IDWT kernels read them: all
     * planes coded, reversible 5/3 with at least one level, all blocks HT cleanup-only, <= 64 columns, M_b <= 15,
     * step size 1, no ROI shift, all levels of fast geometry.  The sub-bands then travel as 2 bytes per sample. */
    bool pair_ok = false;              /* ... and k_ht_decode_pair's dword stores are aligned: even widths, strides, offsets */
    int multi_nb = 0;                  /* k_ht_decode_multi: blocks per wave (2: blocks up to 64 columns, 4: up to 32), 0: not eligible */
    int multi_t = 0;                   /* ... and the transform all HT blocks of the job share */
    bool coef16_ok = false;
    bool coef_is16 = false;            /* what the last HT stage run actually wrote */
    int ht_bpw = 0;                    /* ... and with how many blocks per wave in the MagSgn kernel */
    bool ll16_run = false;             /* the last IDWT run wrote its LL bands as int16_t ... */
    bool ll16_checked = true;          /* ... and the overflow flag behind d_status has been looked at since */
    bool force_ll32 = false;           /* the re-run after an overflow */
    int ll16_fallbacks = 0;
    int nht = 0;
    std::vector<MqWave> mqwaves;       /* one per 64 Part-1 blocks */
    size_t mq_scratch_units = 0;       /* 512-byte row slots of k_mq_decode's scratch */
    uint32_t mq_planes = 1;            /* most bit-planes of a Part-1 block: sizes the LDS of k_mq_decode */
    DevBuf d_mqwaves, d_mqscratch;
    /* device gather: all packets of the batch, the parsers' literal bytes, the merged gather table, first piece of every block */
    bool dev_gather = false;
    DevBuf d_pkt, d_lit, d_gsegs, d_ggroups;
    std::vector<J2kSeg> gsegs;
    std::vector<uint32_t> ggroups;
    std::vector<uint8_t> lit;
    size_t pkt_bytes = 0;
    bool pkt_h2d_issued = false;       /* htj2k_job_parse_batch has sent the packets on their way already: the next upload does not */
    std::vector<uint32_t> qoff;        /* first quad of every (sorted) block in d_qsym */
    std::vector<uint32_t> reflist, roff;   /* blocks with refinement passes that k_ht_refine handles; first mask of each in d_refbits */
    size_t nrefmasks = 0;
    uint32_t ref_max_w = 0;            /* widest block of reflist: k_ht_refine keeps 32-bit row masks when it is at most 32 */
    uint32_t ref_max_h = 0;            /* tallest: k_ht_decode_multi stages the SigProp masks of its blocks in LDS (at most 64 rows) */
    uint32_t max_lref = 0;
    size_t nquads = 0;
    HtLds lds_ext;                     /* LDS layout of k_ht_decode<true> (no VLC windows, no tables) */
    uint32_t max_qw = 1;
    std::vector<uint8_t> h_desc;       /* host image of d_desc: level tables + pack tiles */
    std::vector<uint8_t> h_desc_dev;   /* ... and what d_desc holds (empty after a parse: the tables are rebuilt and d_desc may move) */
    std::vector<LevelLaunch> launches_generic, launches_tile;
    std::vector<LevelLaunch> launches_fused;   /* idwt_mode 3 with the pack stage fused into the final level */
    std::vector<uint8_t> tile_fusable;         /* per PackTile */
    std::vector<uint8_t> plane_fused;          /* per tilecomp: the last IDWT run never wrote its final plane */
    bool any_fusable = false;
    /* k_idwt_stream_ll16_x3: the first three 5/3 levels of every plane in one launch (jobs with 16-bit LL bands) */
    int pk_used = 0;                   /* the last run launched a packed final level: the LL bands were checked against this many bits */
    bool pk_run = false;               /* this run takes the packed kernels where a launch qualifies (the knob idwt_pk) */
    int pk_bits = 0;                   /* the LL bands of this job must fit this many bits for its packed 16-bit final levels (0: none) */
    bool x3_ok = false;
    size_t x3_tab[3] = { 0, 0, 0 };            /* the three levels' DwtTileArgs tables in d_desc (same planes, same order) */
    int x3_count = 0, x3_lh[3] = { 0, 0, 0 }, x3_lv2 = 0;
    double x3_alg = 0, x3_hbm = 0;
    bool fused_last = false;                   /* the last run used launches_fused */
    size_t pack_off = 0; int npack = 0; int pack_maxw = 0, pack_maxh = 0;
    HtLds lds;
    int uploaded = 0, ran = 0;
    std::vector<int> final_buf;        /* per tilecomp, tile mode: 0 = coef, 1 = t0, 2 = t1 */
    std::vector<int> final_eff;        /* where each plane actually is after the last IDWT run */
    bool tile_ok = true;               /* no empty levels: all planes of a launch ping-pong in step */
};

static void clog(htj2k_ctx *c, int level, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    if (!c || !c->log) return;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    c->log(c->log_opaque, level, buf);
}

static void parser_log_tramp(void *opaque, int level, const char *msg)
{
    htj2k_ctx *c = (htj2k_ctx *)opaque;
    if (c && c->log) {
        std::lock_guard<std::mutex> lk(c->log_mutex);      /* frames of a batch are parsed by several threads */
        c->log(c->log_opaque, level, msg);
    }
}

#define HIP_TRY(c, expr)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            clog(c, LOG_ERROR, "HIP error %s at %s:%d (%s)\n", hipGetErrorString(e_), __FILE__, __LINE__, #expr); \
            return e_ == hipErrorOutOfMemory ? HTJ2K_ERR_ENOMEM : HTJ2K_ERR_EXTERNAL;         \
        }                                                                                     \
    } while (0)

/* ------------------------------------------------------------------ CxtVLC decode tables
 * expanded on the host from the Annex C rows (the reference's dec_cxt_vlc_table0/1,
 * libavcodec/jpeg2000htdec.c:1342-1502, hold the same constants pre-expanded);
 * entry: bit0 u_off, bits1-3 len, 4-7 rho, 8-11 e_k, 12-15 e_1 (:320-327) */
static uint16_t h_tables[2][1024];
static void tbl_row(int t, int ctx, int rho, int uoff, int ek, int e1, int cwd, int len)
{
    uint16_t v = (uint16_t)(uoff | (len << 1) | (rho << 4) | (ek << 8) | (e1 << 12));
    for (int hi = 0; hi < (1 << (7 - len)); hi++)
        h_tables[t][(ctx << 7) | (hi << len) | cwd] = v;
}
#define DROW0(c, r, u, k, o, w, l) tbl_row(0, c, r, u, k, o, w, l);
#define DROW1(c, r, u, k, o, w, l) tbl_row(1, c, r, u, k, o, w, l);
static void build_tables()
{
    HT_CXTVLC_ROWS0(DROW0)
    HT_CXTVLC_ROWS1(DROW1)
}

/* ------------------------------------------------------------------ open / close */
extern "C" const char *htj2k_version(void) { return "htj2k-amd 0.1 (gfx950)"; }

extern "C" int htj2k_open(const htj2k_opts *opts, htj2k_ctx **out)
{
    int ndev = 0;
    if (!out) return HTJ2K_ERR_EINVAL;
    *out = nullptr;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return HTJ2K_ERR_ENOSYS;                 /* no GPU: fail loudly, there is no CPU path */
    htj2k_ctx *c = new (std::nothrow) htj2k_ctx();
    if (!c) return HTJ2K_ERR_ENOMEM;
    memset(&c->opts, 0, sizeof(c->opts));
    c->opts.req_pix_fmt = HTJ2K_PIX_NONE;
    if (opts) c->opts = *opts;
    c->device = c->opts.device_id;
    if (c->device < 0 || c->device >= ndev) { delete c; return HTJ2K_ERR_EINVAL; }
    if (hipSetDevice(c->device) != hipSuccess) { delete c; return HTJ2K_ERR_ENOSYS; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) { delete c; return HTJ2K_ERR_ENOSYS; }
    snprintf(c->devname, sizeof(c->devname), "%s", prop.gcnArchName);
    c->max_dyn_lds = (int)prop.sharedMemPerBlock;
    build_tables();
    if (hipMalloc((void **)&c->d_tables, sizeof(h_tables)) != hipSuccess ||
        hipMemcpy(c->d_tables, h_tables, sizeof(h_tables), hipMemcpyHostToDevice) != hipSuccess) {
        delete c;
        return HTJ2K_ERR_ENOSYS;
    }
    c->probe_parser = j2k_parser_new();
    if (!c->probe_parser) { (void)hipFree(c->d_tables); delete c; return HTJ2K_ERR_ENOMEM; }
    const char *hm = getenv("HTJ2K_HT");
    if (hm && !strcmp(hm, "fused")) c->ht_mode = 0;
    if (hm && !strcmp(hm, "split")) c->ht_mode = 1;
    const char *m = getenv("HTJ2K_IDWT");
    if (m && !strcmp(m, "generic")) c->idwt_mode = 0;
    if (m && !strcmp(m, "tile")) c->idwt_mode = 1;
    if (m && !strcmp(m, "stream")) c->idwt_mode = 3;
    const char *l16 = getenv("HTJ2K_LL16");
    if (l16) c->ll16 = atoi(l16) ? 1 : 0;
    const char *pk = getenv("HTJ2K_PK");
    if (pk) c->idwt_pk = atoi(pk) ? 1 : 0;
    const char *fz = getenv("HTJ2K_FUSE");
    if (fz) c->fuse_pack = atoi(fz) != 0;
    *out = c;
    return 0;
}

extern "C" void htj2k_set_log(htj2k_ctx *c, htj2k_log_fn fn, void *opaque)
{
    if (!c) return;
    c->log = fn;
    c->log_opaque = opaque;
    j2k_parser_set_log(c->probe_parser, fn ? parser_log_tramp : nullptr, c);
}

extern "C" int htj2k_set_int(htj2k_ctx *c, const char *name, int value)
{
    if (!c || !name) return HTJ2K_ERR_EINVAL;
    if (!strcmp(name, "idwt_mode")) { c->idwt_mode = value <= 0 ? 0 : (value >= 3 ? 3 : 1); return 0; }
    if (!strcmp(name, "fuse_pack")) { c->fuse_pack = value ? 1 : 0; return 0; }
    if (!strcmp(name, "parse_threads")) { c->parse_threads = value < 0 ? 0 : (value > 64 ? 64 : value); return 0; }
    if (!strcmp(name, "ht_mode")) { c->ht_mode = value ? 1 : 0; return 0; }
    if (!strcmp(name, "device_gather")) { c->device_gather = value ? 1 : 0; return 0; }
    if (!strcmp(name, "packet_threads")) { c->packet_threads = value < 1 ? 1 : (value > 16 ? 16 : value); return 0; }
    if (!strcmp(name, "coef16")) { c->coef16 = value ? 1 : 0; return 0; }
    if (!strcmp(name, "ht_pair")) { c->ht_pair = value ? 1 : 0; return 0; }
    if (!strcmp(name, "idwt_x3")) { c->idwt_x3 = value ? 1 : 0; return 0; }
    if (!strcmp(name, "idwt_pk")) { c->idwt_pk = value ? 1 : 0; return 0; }
    if (!strcmp(name, "ht_multi")) { c->ht_multi = value ? 1 : 0; return 0; }
    if (!strcmp(name, "ll16")) { c->ll16 = value ? 1 : 0; return 0; }
    if (!strcmp(name, "ll16_test_bits")) { if (value < 2 || value > 16) return HTJ2K_ERR_EINVAL; c->ll16_test_bits = value; return 0; }
    if (!strcmp(name, "bitexact")) { c->opts.bitexact = value; return 0; }
    if (!strcmp(name, "reduction_factor")) { c->opts.reduction_factor = value; return 0; }
    return HTJ2K_ERR_EINVAL;
}

extern "C" const char *htj2k_device_name(htj2k_ctx *c) { return c ? c->devname : ""; }

extern "C" void *htj2k_host_alloc(htj2k_ctx *c, size_t size)
{
    void *p = nullptr;
    if (!c || !size || hipSetDevice(c->device) != hipSuccess) return nullptr;
    if (hipHostMalloc(&p, size, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

extern "C" void htj2k_host_free(htj2k_ctx *c, void *ptr)
{
    (void)c;
    if (ptr) (void)hipHostFree(ptr);
}

extern "C" int htj2k_device_to_host(htj2k_ctx *c, void *dst, const void *src, size_t size)
{
    if (!c || !dst || !src) return HTJ2K_ERR_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpy(dst, src, size, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" void htj2k_job_free(htj2k_ctx *c, htj2k_job *j)
{
    (void)c;
    if (!j) return;
    if (j->stream) (void)hipStreamSynchronize(j->stream);
    j->d_bytes.release(); j->d_blocks.release(); j->d_status.release(); j->d_coef.release();
    j->d_t0.release(); j->d_t1.release(); j->d_desc.release(); j->d_qsym.release(); j->d_qoff.release();
    j->d_vlcu.release(); j->d_melu.release(); j->d_reflist.release(); j->d_roff.release(); j->d_refbits.release();
    j->d_mqwaves.release(); j->d_mqscratch.release();
    j->d_pkt.release(); j->d_lit.release(); j->d_gsegs.release(); j->d_ggroups.release();
    for (FrameSlot &f : j->frames) {
        for (int i = 0; i < 4; i++) f.d_out[i].release();
        j2k_parser_free(f.parser);
        f.h_bytes.release();
        f.h_pkt.release();
    }
    for (int i = 0; i < 6; i++) if (j->ev[i]) (void)hipEventDestroy(j->ev[i]);
    for (hipEvent_t e : j->lev_ev) if (e) (void)hipEventDestroy(e);
    if (j->stream) (void)hipStreamDestroy(j->stream);
    delete j;
}

/* a pipe keeps its context alive: htj2k_close by the caller while a closed pipe still has device frames out (reference-
 * counted frames may outlive the decoder) must not free what those frames' release path needs */
extern "C" void htj2k_ctx_ref_(htj2k_ctx *c) { if (c) c->refs.fetch_add(1); }
/* htj2k_opts.frames_in_flight: the pipeline depth of a pipe opened with depth 0 */
extern "C" int htj2k_ctx_default_depth_(const htj2k_ctx *c)
{
    const int d = c ? c->opts.frames_in_flight : 0;
    return d <= 0 ? 3 : (d > 16 ? 16 : d);
}

extern "C" void htj2k_close(htj2k_ctx *c)
{
    if (c && c->refs.fetch_sub(1) != 1) return;            /* the last reference frees */
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->own_job) htj2k_job_free(c, c->own_job);
    if (c->d_tables) (void)hipFree(c->d_tables);
    j2k_parser_free(c->probe_parser);
    delete c;
}

extern "C" int htj2k_probe(htj2k_ctx *c, const uint8_t *pkt, int size, htj2k_info *info)
{
    const J2kPlan *pl = nullptr;
    if (!c || !pkt || !info) return HTJ2K_ERR_EINVAL;
    int ret = j2k_parse(c->probe_parser, pkt, size, &c->opts, 1, &pl);
    if (ret < 0) return ret;
    *info = pl->info;
    return 0;
}

/* ------------------------------------------------------------------ job: parse */
static int job_new(htj2k_ctx *c, htj2k_job **out)
{
    htj2k_job *j = new (std::nothrow) htj2k_job();
    if (!j) return HTJ2K_ERR_ENOMEM;
    if (hipSetDevice(c->device) != hipSuccess || hipStreamCreateWithFlags(&j->stream, hipStreamNonBlocking) != hipSuccess) {
        delete j; return HTJ2K_ERR_EXTERNAL;
    }
    for (int i = 0; i < 6; i++)
        if (hipEventCreate(&j->ev[i]) != hipSuccess) { htj2k_job_free(c, j); return HTJ2K_ERR_EXTERNAL; }
    *out = j;
    return 0;
}

/* A job holds a batch of frames.  Every frame is parsed by its own parser; the descriptor
 * tables are then concatenated (offsets re-based into one device arena) so that each stage
 * of the whole batch is a single launch: frames are independent, so a batch is simply a
 * longer codeblock table, more planes per IDWT level, more tiles to pack. */
/* one packet -> its place in d_pkt, asynchronously on the job's stream.  A staging copy of ours carries its 64 bytes of zero
 * padding; a caller's page-locked packet is read up to its last byte only and the padding is set on the device (k_gather reads
 * whole dwords, up to 3 bytes past a segment) */
static hipError_t job_packet_h2d(htj2k_job *j, const FrameSlot &F, size_t size)
{
    uint8_t *dst = (uint8_t *)j->d_pkt.p + F.pkt_base;
    hipError_t e = hipMemcpyAsync(dst, F.h2d_src, size + (F.h2d_own ? 64 : 0), hipMemcpyHostToDevice, j->stream);
    if (e == hipSuccess && !F.h2d_own) e = hipMemsetAsync(dst + size, 0, 64, j->stream);
    return e;
}

extern "C" int htj2k_job_parse_batch(htj2k_ctx *c, const uint8_t *const *pkts, const int *sizes, int n, htj2k_job **job)
{
    return htj2k_job_parse_batch_ex(c, pkts, sizes, n, nullptr, job);
}

extern "C" int htj2k_job_parse_batch_ex(htj2k_ctx *c, const uint8_t *const *pkts, const int *sizes, int n, const uint8_t *pinned,
                                        htj2k_job **job)
{
    if (!c || !pkts || !sizes || n <= 0 || !job) return HTJ2K_ERR_EINVAL;
    if (!*job) {
        int r = job_new(c, job);
        if (r < 0) return r;
    }
    htj2k_job *j = *job;
    /* the previous batch of this job may still be in flight and reads the parsers' arenas */
    if (hipStreamSynchronize(j->stream) != hipSuccess) return HTJ2K_ERR_EXTERNAL;
    j->uploaded = j->ran = 0;
    j->nframes = 0;
    if ((int)j->frames.size() < n) j->frames.resize(n);
    j->blocks.clear(); j->tilecomps.clear(); j->tc_frame.clear();
    size_t nbytes = 0, nsamples = 0;
    j->dev_gather = c->device_gather != 0;
    for (int f = 0; f < n; f++) {
        FrameSlot &F = j->frames[f];
        if (!F.parser) {
            F.parser = j2k_parser_new();
            if (!F.parser) return HTJ2K_ERR_ENOMEM;
            j2k_parser_set_log(F.parser, parser_log_tramp, c);
        }
        /* host gather: the codeblock bytes are gathered straight into pinned memory; device gather: the packet is staged
         * in pinned memory as it is.  Either way the H2D copy of the upload then runs at PCIe rate and truly asynchronously
         * (the hook is re-registered on every call: the FrameSlot may have moved when the vector grew) */
        F.h_bytes.device = c->device;
        F.h_pkt.device = c->device;
        j2k_parser_set_gather(F.parser, !j->dev_gather);
        j2k_parser_set_packet_threads(F.parser, n == 1 ? c->packet_threads : 1);   /* batches: one frame per thread instead */
        j2k_parser_set_bytes_alloc(F.parser, [](void *opaque, size_t nb) -> void * { return ((HostBuf *)opaque)->ensure(nb); },
                                   &F.h_bytes);
        F.plan = nullptr;
    }
    /* frames are independent: stage and parse them on several host threads */
    {
        std::vector<int> rc(n, 0);
        int nthreads = c->parse_threads > 0 ? c->parse_threads : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        if (nthreads > n) nthreads = n;
        auto parallel_frames = [&](auto &&fn) {
            std::atomic<int> next(0);
            auto work = [&]() {
                for (;;) {
                    const int f = next.fetch_add(1);
                    if (f >= n) break;
                    fn(f);
                }
            };
            if (nthreads <= 1) {
                work();
            } else {
                std::vector<std::thread> pool;
                for (int t = 1; t < nthreads; t++) pool.emplace_back(work);
                work();
                for (std::thread &t : pool) t.join();
            }
        };
        const bool stage = j->dev_gather;
        /* 1. device gather: every packet into page-locked memory (unless the caller's already is) ... */
        std::vector<const uint8_t *> src(pkts, pkts + n);
        /* One frame on its own (htj2k_decode): the staging copy of a 4K packet (15.7 MB, 0.55 ms on one core) was the longest
         * item of the call.  Helper threads copy it in pieces, each piece goes to the device as soon as it is in page-locked
         * memory, and this thread meanwhile parses the caller's packet itself -- the parser reads headers only and, with the
         * device gathering, nothing looks at the packet again after the parse. */
        bool solo_done = false;
        if (stage && n == 1 && !(pinned && pinned[0]) && sizes[0] >= (4 << 20) && hipSetDevice(c->device) == hipSuccess) {
            FrameSlot &F = j->frames[0];
            const size_t sz = (size_t)sizes[0];
            const auto t0 = std::chrono::steady_clock::now();
            uint8_t *h = (uint8_t *)F.h_pkt.ensure(sz + 64);
            if (!h) return HTJ2K_ERR_ENOMEM;
            if (j->d_pkt.ensure(((sz + 64 + 15) & ~(size_t)15) + 256) == 0) {
                F.pkt_base = 0;
                F.h2d_src = h; F.h2d_own = true;
                const size_t piece = 2 << 20, npieces = (sz + piece - 1) / piece;
                const int nhelp = (int)std::min<size_t>(3, std::max(1u, std::thread::hardware_concurrency()) > 1 ? 3 : 1);
                std::atomic<size_t> nextp(0);
                std::atomic<int> bad(0);
                auto copy_pieces = [&]() {
                    if (hipSetDevice(c->device) != hipSuccess) { bad = 1; return; }
                    for (;;) {
                        const size_t k = nextp.fetch_add(1);
                        if (k >= npieces) break;
                        const size_t off = k * piece, len = std::min(piece, sz - off), pad = k + 1 == npieces ? 64 : 0;
                        memcpy(h + off, pkts[0] + off, len);
                        if (pad) memset(h + sz, 0, 64);
                        if (hipMemcpyAsync((uint8_t *)j->d_pkt.p + off, h + off, len + pad, hipMemcpyHostToDevice, j->stream) != hipSuccess) bad = 1;
                    }
                };
                std::vector<std::thread> helpers;
                for (int t = 0; t < nhelp; t++) helpers.emplace_back(copy_pieces);
                const auto t1 = std::chrono::steady_clock::now();
                rc[0] = j2k_parse(F.parser, pkts[0], sizes[0], &c->opts, 0, &F.plan);
                F.ms_parse = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t1).count();
                for (std::thread &t : helpers) t.join();
                F.ms_stage = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count() - F.ms_parse;
                if (bad) return HTJ2K_ERR_EXTERNAL;
                if (rc[0] < 0) return rc[0];
                j->pkt_h2d_issued = true;
                solo_done = true;
            }
        }
        if (solo_done) {
        } else {
        if (stage)
            parallel_frames([&](int f) {
                FrameSlot &F = j->frames[f];
                const auto t0 = std::chrono::steady_clock::now();
                F.h2d_src = nullptr;
                F.h2d_own = false;
                if (pinned && pinned[f]) {
                    F.h2d_src = src[f];                     /* page-locked and stable: the upload reads the packet itself */
                } else {
                    const size_t sz = sizes[f] > 0 ? (size_t)sizes[f] : 0;
                    uint8_t *h = (uint8_t *)F.h_pkt.ensure(sz + 64);
                    if (!h) { rc[f] = HTJ2K_ERR_ENOMEM; return; }
                    memcpy(h, src[f], sz);
                    memset(h + sz, 0, 64);
                    src[f] = h;
                    F.h2d_src = h;
                    F.h2d_own = true;
                }
                F.ms_stage = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
            });
        for (int f = 0; f < n; f++)
            if (rc[f] < 0) return rc[f];
        /* 2. ... and on its way to the device BEFORE it is parsed: where a packet goes in d_pkt follows from the sizes
         * alone, and the parser needs nothing from the device, so the transfer of a frame (0.35 ms for a 4K frame) runs
         * under its parse (0.36 ms) instead of behind it */
        j->pkt_h2d_issued = false;
        if (stage) {
            size_t total = 0;
            bool fits = true;
            for (int f = 0; f < n; f++) {
                j->frames[f].pkt_base = total;
                total += ((size_t)(sizes[f] > 0 ? sizes[f] : 0) + 64 + 15) & ~(size_t)15;
                if (total > 0xFFFFFF00ull) fits = false;
            }
            if (fits && hipSetDevice(c->device) == hipSuccess && j->d_pkt.ensure(total + 256) == 0) {
                bool ok = true;
                for (int f = 0; f < n && ok; f++) ok = job_packet_h2d(j, j->frames[f], sizes[f] > 0 ? (size_t)sizes[f] : 0) == hipSuccess;
                j->pkt_h2d_issued = ok;
            }
        }
        /* 3. parse */
        parallel_frames([&](int f) {
            FrameSlot &F = j->frames[f];
            const auto t1 = std::chrono::steady_clock::now();
            rc[f] = j2k_parse(F.parser, src[f], sizes[f], &c->opts, 0, &F.plan);
            F.ms_parse = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t1).count();
            if (!stage) F.ms_stage = 0;
        });
        for (int f = 0; f < n; f++)
            if (rc[f] < 0) return rc[f];                   /* the first failing frame in submission order */
        }
    }
    j->gsegs.clear(); j->ggroups.clear(); j->lit.clear();
    size_t pkt_bytes = 0;
    for (int f = 0; f < n; f++) {
        FrameSlot &F = j->frames[f];
        const J2kPlan *pl = F.plan;
        if (nsamples + pl->nsamples > 0xFFFFFF00ull || nbytes + pl->nbytes > 0xFFFFFF00ull)
            return HTJ2K_ERR_PATCHWELCOME;                   /* sample and byte offsets of a job are 32-bit */
        F.block_base = (uint32_t)j->blocks.size();
        F.tc_base = (uint32_t)j->tilecomps.size();
        F.sample_base = (uint32_t)nsamples;
        nbytes += 16;                                  /* k_ht_unstuff's backward dword loads may start up to 3 bytes
                                                        * in front of a block: never in front of the buffer */
        F.bytes_base = nbytes;
        for (int i = 0; i < pl->nblocks; i++) {
            J2kBlock b = pl->blocks[i];
            b.data_off += (uint32_t)nbytes;
            b.plane_off += (uint32_t)nsamples;
            b.tcomp = (uint8_t)(b.tcomp + F.tc_base);          /* informational only (wraps beyond 255 tile-components) */
            j->blocks.push_back(b);
        }
        for (int t = 0; t < pl->ntilecomps; t++) {
            J2kTileComp tc = pl->tilecomps[t];
            tc.plane_off += (uint32_t)nsamples;
            j->tilecomps.push_back(tc);
            j->tc_frame.push_back(f);
        }
        if (j->dev_gather) {
            /* the frame's gather table joins the job's: sources re-based to where the packet / the literal bytes will
             * sit on the device, destinations to the frame's part of the pool */
            if (pkt_bytes + (size_t)pl->pkt_size > 0xFFFFFF00ull || j->lit.size() + pl->nlit > 0xFFFFFF00ull ||
                j->gsegs.size() + pl->nsegs > 0xFFFFFF00ull)
                return HTJ2K_ERR_PATCHWELCOME;
            if (F.pkt_base != pkt_bytes) j->pkt_h2d_issued = false;   /* (cannot happen: same formula as above) */
            F.pkt_base = pkt_bytes;
            const uint32_t lit_base = (uint32_t)j->lit.size(), seg_base = (uint32_t)j->gsegs.size();
            for (uint32_t k = 0; k < pl->nsegs; k++) {
                J2kSeg g = pl->segs[k];
                g.src += (g.flags & J2K_SEG_LIT) ? lit_base : (uint32_t)pkt_bytes;
                g.dst += (uint32_t)nbytes;
                j->gsegs.push_back(g);
            }
            for (int i = 0; i < pl->nblocks; i++) j->ggroups.push_back(seg_base + pl->blk_seg0[i]);
            j->lit.insert(j->lit.end(), pl->lit, pl->lit + pl->nlit);
            pkt_bytes += ((size_t)pl->pkt_size + 64 + 15) & ~(size_t)15;
        }
        nbytes += (pl->nbytes + 63) & ~(size_t)63;
        nsamples += pl->nsamples;
    }
    if (j->dev_gather) j->ggroups.push_back((uint32_t)j->gsegs.size());
    j->pkt_bytes = pkt_bytes;
    j->nbytes = nbytes;
    j->nsamples = nsamples;
    j->nframes = n;
    return 0;
}

extern "C" int htj2k_job_parse(htj2k_ctx *c, const uint8_t *pkt, int size, htj2k_job **job)
{
    const uint8_t *pk[1] = { pkt };
    int sz[1] = { size };
    if (!pkt) return HTJ2K_ERR_EINVAL;
    return htj2k_job_parse_batch(c, pk, sz, 1, job);
}

extern "C" int htj2k_job_num_frames(const htj2k_job *j) { return j ? j->nframes : HTJ2K_ERR_EINVAL; }

extern "C" int htj2k_job_host_ms(const htj2k_job *j, float *ms_parse, float *ms_stage)
{
    if (!j || j->nframes <= 0) return HTJ2K_ERR_EINVAL;
    float p = 0, s = 0;
    for (int f = 0; f < j->nframes; f++) { p += j->frames[f].ms_parse; s += j->frames[f].ms_stage; }
    if (ms_parse) *ms_parse = p / j->nframes;
    if (ms_stage) *ms_stage = s / j->nframes;
    return 0;
}

extern "C" int htj2k_job_frame_info(const htj2k_job *j, int frame, htj2k_info *info)
{
    if (!j || frame < 0 || frame >= j->nframes || !j->frames[frame].plan || !info) return HTJ2K_ERR_EINVAL;
    *info = j->frames[frame].plan->info;
    return 0;
}
extern "C" int htj2k_job_info(const htj2k_job *j, htj2k_info *info) { return htj2k_job_frame_info(j, 0, info); }
extern "C" int htj2k_job_bytes_consumed(const htj2k_job *j)
{
    return j && j->nframes > 0 && j->frames[0].plan ? j->frames[0].plan->bytes_consumed : HTJ2K_ERR_EINVAL;
}
extern "C" int htj2k_job_num_tilecomps(const htj2k_job *j) { return j && j->nframes ? (int)j->tilecomps.size() : HTJ2K_ERR_EINVAL; }
extern "C" int htj2k_job_num_blocks(const htj2k_job *j) { return j && j->nframes ? (int)j->blocks.size() : HTJ2K_ERR_EINVAL; }
extern "C" int htj2k_job_tilecomp_dims(const htj2k_job *j, int tc, int *w, int *h, int *is_float)
{
    if (!j || !j->nframes || tc < 0 || tc >= (int)j->tilecomps.size()) return HTJ2K_ERR_EINVAL;
    if (w) *w = j->tilecomps[tc].w;
    if (h) *h = j->tilecomps[tc].h;
    if (is_float) *is_float = j->tilecomps[tc].transform == J2K_DWT97;
    return 0;
}

/* ------------------------------------------------------------------ job: upload */
static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

/* LDS windows of the HT kernel are sized from the largest cleanup prefix / suffix, quad row
 * and (for blocks with refinement passes) state bitmap in the table */
static int ht_lds_layout(htj2k_ctx *c, uint32_t max_p, uint32_t max_s, uint32_t max_qw, uint32_t bm_words, HtLds *out, HtLds *out_ext);

/* LDS sizing from a block table + byte pool (unit entry point; jobs take the figures from the plans) */
static int build_ht_lds(htj2k_ctx *c, const J2kBlock *blocks, int nblocks, const uint8_t *const *bases,
                        const uint32_t *base_of_block, HtLds *out, HtLds *out_ext = nullptr)
{
    uint32_t max_p = 0, max_s = 2, max_qw = 1, bm_words = 0;
    for (int i = 0; i < nblocks; i++) {
        const J2kBlock &b = blocks[i];
        if (!b.npasses) continue;
        uint32_t qw = (b.w + 1u) >> 1;
        if (qw > max_qw) max_qw = qw;
        if (b.lcup >= 2) {
            const uint8_t *D = bases[base_of_block ? base_of_block[i] : 0] + b.data_off;
            uint32_t scup = ((uint32_t)D[b.lcup - 1] << 4) + (D[b.lcup - 2] & 0x0F);
            if (scup >= 2 && scup <= b.lcup && scup <= 4079) {
                if (scup > max_s) max_s = scup;
                if (b.lcup - scup > max_p) max_p = b.lcup - scup;
            }
        }
        int rem = b.npasses % 3, plhd = rem ? b.npasses - rem : b.npasses - 3;
        if (b.npasses - plhd > 1) {
            uint32_t wds = ((uint32_t)(b.w + 2) * (b.h + 2) + 31) / 32 + 1;
            if (wds > bm_words) bm_words = wds;
        }
    }
    return ht_lds_layout(c, max_p, max_s, max_qw, bm_words, out, out_ext);
}

static int ht_lds_layout(htj2k_ctx *c, uint32_t max_p, uint32_t max_s, uint32_t max_qw, uint32_t bm_words, HtLds *out, HtLds *out_ext)
{
    HtLds &L = *out;
    size_t off = 4096;                                   /* the two CxtVLC tables */
    L.ms_words = ((max_p * 8 + 31) / 32 + 3 + 3) & ~3u;  /* a multiple of 4: the kernels clear the array 16 bytes per lane */
    L.off_ms = (uint32_t)off;  off += (size_t)L.ms_words * 4;
    L.vlc_words = (max_s * 8 + 31) / 32 + 2;
    L.off_vlc = (uint32_t)off; off += (size_t)L.vlc_words * 4;
    L.suf_bytes = (uint32_t)align_up(max_s, 4);
    L.off_suf = (uint32_t)off; off += L.suf_bytes;
    L.max_qw = max_qw;
    L.off_qinfo = (uint32_t)off; off += (size_t)2 * max_qw * 4;
    L.off_E = (uint32_t)off;   off += align_up((size_t)2 * (2 * max_qw + 8), 4);
    L.bm_words = bm_words;
    L.off_bm = (uint32_t)off;  off += (size_t)4 * bm_words * 4;
    L.total = (uint32_t)align_up(off, 16);
    if (out_ext) {                                       /* MagSgn-only kernel: bit array, exponents, bitmaps */
        HtLds &X = *out_ext;
        size_t o = 0;
        X = L;
        X.off_ms = 0; o += (size_t)X.ms_words * 4;
        X.off_vlc = (uint32_t)o; X.vlc_words = 0;
        X.off_suf = (uint32_t)o; X.suf_bytes = 0;
        X.off_qinfo = (uint32_t)o; o += (size_t)2 * max_qw * 4;
        X.off_E = (uint32_t)o; o += align_up((size_t)2 * (2 * max_qw + 8), 4);
        X.off_bm = (uint32_t)o; o += (size_t)4 * bm_words * 4;
        X.total = (uint32_t)align_up(o, 16);
    }
    if ((int)L.total > 160 * 1024) {
        clog(c, LOG_ERROR, "a codeblock needs %u bytes of LDS\n", L.total);
        return HTJ2K_ERR_PATCHWELCOME;
    }
    return 0;
}

/* Which blocks k_ht_refine handles (SigProp / MagRef passes, at most 64 columns, no ROI shift) and where their masks
 * go: the block on lane l of the kernel's wave g (reflist[64 g + l]) has mask k of row y at
 * roff[block] + (3 y + k) * HT_REF_STRIDE, roff[block] = base of wave g + l; a wave takes 64 * 3 * (its tallest block)
 * words.  Returns the total in 64-bit words; *max_w = the widest such block. */
static size_t ref_layout(const J2kBlock *blocks, size_t nblocks, std::vector<uint32_t> &reflist, std::vector<uint32_t> &roff, uint32_t *max_w,
                         uint32_t *max_h = nullptr)
{
    reflist.clear();
    roff.assign(nblocks + 1, 0);
    *max_w = 0;
    if (max_h) *max_h = 0;
    for (size_t i = 0; i < nblocks; i++) {
        const J2kBlock &b = blocks[i];
        const int rem = b.npasses % 3, plhd = rem ? b.npasses - rem : b.npasses - 3;
        if (b.npasses && !(b.flags & J2K_BLK_PART1) && b.npasses - plhd > 1 && b.w <= 64 && b.roi_shift == 0) {
            reflist.push_back((uint32_t)i);
            if (b.w > *max_w) *max_w = b.w;
            if (max_h && b.h > *max_h) *max_h = b.h;
        }
    }
    size_t nm = 0;
    for (size_t g = 0; g < reflist.size(); g += 64) {
        uint32_t hmax = 0;
        for (size_t l = g; l < std::min(g + 64, reflist.size()); l++) hmax = std::max<uint32_t>(hmax, blocks[reflist[l]].h);
        for (size_t l = g; l < std::min(g + 64, reflist.size()); l++) roff[reflist[l]] = (uint32_t)(nm + (l - g));
        nm += (size_t)HT_REF_STRIDE * 3 * hmax;
    }
    return nm;
}

static void push_bytes(std::vector<uint8_t> &v, const void *p, size_t n)
{
    const uint8_t *b = (const uint8_t *)p;
    v.insert(v.end(), b, b + n);
}

/* descriptor tables for the IDWT launches and the pack stage */
/* ---- packed 16-bit final level (dwt_stream.hpp, PK): when is it exact? ----
 * The kernel computes the horizontal and vertical 5/3 lifting of the final level, the inverse RCT and the clip on pairs of
 * 16-bit samples with wrapping sums.  It equals the 32-bit arithmetic of the reference exactly when no intermediate leaves
 * 16 bits.  What bounds the inputs: a reversible HT coefficient is a 31-bit magnitude shifted down by 31 - M_b
 * (jpeg2000dec.c:2135-2144), so |coefficient| <= 2^M_b - 1 whatever the block's bytes hold; and the LL band is the output
 * of the level below, which the kernels that write it check against `bits` bits (their `ovf` flag: a job that fails runs
 * again with int32 LL bands and the 32-bit kernels).  Interval arithmetic through the step -- every sum, every shifted sum --
 * says whether a given `bits` is enough.  bound() returns the largest magnitude of a component's output samples, or -1. */
static long pk16_lift_bound(long ll, long hl, long lh, long hh)
{
    const long LIM = 32767;
    bool ok = true;
    auto s1 = [&](long c0, long a) { if (2 * a + 2 > LIM) ok = false; return c0 + (2 * a + 2) / 4; };   /* c - ((a + b + 2) >> 2) */
    auto s2 = [&](long c0, long a) { if (2 * a > LIM) ok = false; return c0 + a; };                     /* c + ((a + b) >> 1) */
    const long e = s1(ll, hl), o = s2(hl, e);              /* horizontal, vertical-low rows */
    const long eh = s1(lh, hh), oh = s2(hh, eh);           /* horizontal, vertical-high rows */
    const long re0 = s1(e, eh), ro0 = s2(eh, re0);         /* vertical, even columns */
    const long re1 = s1(o, oh), ro1 = s2(oh, re1);         /* vertical, odd columns */
    const long m = std::max(std::max(re0, ro0), std::max(re1, ro1));
    return ok && m <= LIM && e <= LIM && o <= LIM && eh <= LIM && oh <= LIM ? m : -1;
}
static bool pk16_bounds(const long (*B)[4], int nc, bool rct)     /* B[c] = { LL, HL, LH, HH } magnitude bounds of component c */
{
    long bc[3] = { 0, 0, 0 };
    for (int c = 0; c < nc; c++)
        if ((bc[c] = pk16_lift_bound(B[c][0], B[c][1], B[c][2], B[c][3])) < 0) return false;
    if (rct) {                                              /* g = y - ((cr + cb) >> 2); g + cr and g + cb saturate */
        if (bc[1] + bc[2] > 32767) return false;
        if (bc[0] + (bc[1] + bc[2]) / 4 + 1 > 32767) return false;
    }
    return true;
}

extern "C" long htj2k_pk16_lift_bound(long ll, long hl, long lh, long hh) { return pk16_lift_bound(ll, hl, lh, hh); }
extern "C" int htj2k_pk16_bounds(const long (*b)[4], int nc, int rct) { return b && nc >= 1 && nc <= 3 && pk16_bounds(b, nc, rct != 0) ? 1 : 0; }

static void pk16_eligibility(htj2k_job *j)
{
    j->pk_bits = 0;
    for (LevelLaunch &L : j->launches_fused) L.pk_bits = 0;
    if (!j->coef16_ok) return;
    const int ntc = (int)j->tilecomps.size();
    /* the largest M_b among the blocks of every band: [plane][level][0 LL (level 0 only), 1 HL, 2 LH, 3 HH] */
    std::vector<std::pair<uint32_t, int>> start(ntc);
    for (int t = 0; t < ntc; t++) start[t] = { j->tilecomps[t].plane_off, t };
    std::sort(start.begin(), start.end());
    auto plane_of = [&](uint32_t off) {
        auto it = std::upper_bound(start.begin(), start.end(), std::make_pair(off, 1 << 30));
        return it == start.begin() ? -1 : (it - 1)->second;
    };
    std::vector<uint8_t> mb((size_t)ntc * J2K_MAX_DWTLEV * 4, 0);
    for (const J2kBlock &b : j->blocks) {
        const int t = plane_of(b.plane_off);
        if (t < 0) return;
        const J2kTileComp &tc = j->tilecomps[t];
        if (tc.ndeclevels < 1 || tc.ndeclevels > J2K_MAX_DWTLEV || tc.w <= 0) return;
        const uint32_t off = b.plane_off - tc.plane_off;
        const int x = (int)(off % (uint32_t)tc.w), y = (int)(off / (uint32_t)tc.w);
        int lev = tc.ndeclevels - 1, band = 0;
        for (; lev >= 0; lev--) {                           /* the outermost level whose low-pass quadrant does not hold the block */
            const int mh = tc.mod[lev][0], mv = tc.mod[lev][1];
            const int nlx = ((mh + tc.linelen[lev][0] + 1) >> 1) - ((mh + 1) >> 1), nly = ((mv + tc.linelen[lev][1] + 1) >> 1) - ((mv + 1) >> 1);
            band = (x >= nlx ? 1 : 0) + (y >= nly ? 2 : 0);
            if (band || lev == 0) break;
        }
        uint8_t &m = mb[((size_t)t * J2K_MAX_DWTLEV + lev) * 4 + band];
        if (b.M_b > m) m = b.M_b;
    }
    /* the most bits (16 .. 10 for 8-bit components) of LL input with which plane t's
     * level lev stays inside 16 bits: 0 = none; level 0 reads the block decoder's LL band, bounded by its own M_b and unchecked */
    auto bounds_of = [&](int t, int lev, int k, long (&B)[4]) {
        const uint8_t *m = &mb[((size_t)t * J2K_MAX_DWTLEV + lev) * 4];
        B[0] = lev == 0 ? (1L << m[0]) - 1 : 1L << (k - 1);
        for (int q = 1; q < 4; q++) B[q] = (1L << m[q]) - 1;
    };
    int job_bits = 16;
    for (LevelLaunch &L : j->launches_fused) {
        if (L.type != J2K_DWT53) continue;
        int bits = 16;
        if (L.nc == 0) {                                    /* plain level: its output is checked, its input has to be bounded */
            const DwtTileArgs *ta = (const DwtTileArgs *)(j->h_desc.data() + L.table_off);
            for (int i = 0; i < L.count && bits; i++) {
                const int t = plane_of(ta[i].g.plane_off);
                int k = t >= 0 && j->tilecomps[t].plane_off == ta[i].g.plane_off ? bits : 0;
                /* (a component of n bits has LL bands of n + 1 bits with the RCT, and the check must leave an ordinary picture
                 * alone: twice that range at least) */
                const int kmin = k ? j->tilecomps[t].cbps + 2 : 0;
                for (; k >= kmin && k; k--) {
                    long B[4];
                    bounds_of(t, L.level, k, B);
                    if (pk16_lift_bound(B[0], B[1], B[2], B[3]) >= 0) break;
                    if (L.level == 0) { k = 0; break; }
                }
                bits = k >= kmin ? k : 0;
            }
        } else {
            if (!(L.outk == 0 || L.outk == 2)) continue;
            const DwtFusedArgs *fa = (const DwtFusedArgs *)(j->h_desc.data() + L.table_off);
            for (int i = 0; i < L.count && bits; i++) {
                const PackTile *PT = (const PackTile *)(j->h_desc.data() + j->pack_off) + fa[i].pack_tile;
                const bool rct = L.outk == 0 && PT->mct != 0;
                int tcs[3];
                bool ok = PT->precision == 8;
                for (int cc = 0; cc < L.nc && ok; cc++) {
                    tcs[cc] = plane_of(fa[i].a[cc].g.plane_off);
                    ok = tcs[cc] >= 0 && j->tilecomps[tcs[cc]].plane_off == fa[i].a[cc].g.plane_off && PT->c[L.outk == 2 ? fa[i].comp0 : cc].cbps == 8;
                }
                int k = ok ? bits : 0;
                for (; k >= 10; k--) {
                    long B[3][4];
                    for (int cc = 0; cc < L.nc; cc++) bounds_of(tcs[cc], L.level, k, B[cc]);
                    if (pk16_bounds(B, L.nc, rct)) break;
                    if (L.level == 0) { k = 0; break; }
                }
                bits = k >= 10 ? k : 0;
            }
        }
        L.pk_bits = bits;
        if (bits && L.level > 0) job_bits = std::min(job_bits, bits);
    }
    j->pk_bits = job_bits;
}

static int build_descriptors(htj2k_ctx *c, htj2k_job *j)
{
    const int ntc = (int)j->tilecomps.size();
    j->h_desc.clear();
    j->h_desc_dev.clear();                              /* d_desc is written again by the next run */
    j->launches_generic.clear();
    j->launches_tile.clear();
    j->final_buf.assign(ntc, 0);
    int maxlev = 0;
    for (int t = 0; t < ntc; t++)
        if (j->tilecomps[t].coded && j->tilecomps[t].ndeclevels > maxlev) maxlev = j->tilecomps[t].ndeclevels;
    for (int lev = 0; lev < maxlev; lev++)
        for (int type = 0; type < 3; type++) {
            LevelLaunch g;
            g.type = type; g.level = lev; g.count = 0; g.max_lh = g.max_lv = 0; g.alg_bytes = 0; g.min_l = 1 << 30;
            g.all_fast = true;
            std::vector<DwtLevel> lv;
            std::vector<DwtTileArgs> ta;
            for (int t = 0; t < ntc; t++) {
                const J2kTileComp &tc = j->tilecomps[t];
                if (!tc.coded || tc.transform != type || lev >= tc.ndeclevels) continue;
                if (tc.linelen[lev][0] <= 0 || tc.linelen[lev][1] <= 0) continue;   /* empty level: no-op */
                DwtLevel d;
                d.plane_off = tc.plane_off; d.stride = tc.w;
                d.lh = tc.linelen[lev][0]; d.lv = tc.linelen[lev][1];
                d.mh = tc.mod[lev][0]; d.mv = tc.mod[lev][1];
                d.last = (type == J2K_DWT97_INT && lev == tc.ndeclevels - 1) ? 1 : 0;
                lv.push_back(d);
                DwtTileArgs a;
                a.g = d;
                a.ll_off = tc.plane_off; a.ll_stride = tc.w;
                a.out_off = tc.plane_off; a.out_stride = tc.w;
                ta.push_back(a);
                if (d.lh > g.max_lh) g.max_lh = d.lh;
                if (d.lv > g.max_lv) g.max_lv = d.lv;
                if (d.lh < g.min_l) g.min_l = d.lh;
                if (d.lv < g.min_l) g.min_l = d.lv;
                g.alg_bytes += 8.0 * d.lh * d.lv;
                g.all_fast = g.all_fast && stream_fast_geom(d);
            }
            if (lv.empty()) continue;
            g.count = (int)lv.size();
            LevelLaunch tl = g;
            g.table_off = j->h_desc.size();
            push_bytes(j->h_desc, lv.data(), lv.size() * sizeof(DwtLevel));
            while (j->h_desc.size() % 16) j->h_desc.push_back(0);
            tl.table_off = j->h_desc.size();
            push_bytes(j->h_desc, ta.data(), ta.size() * sizeof(DwtTileArgs));
            while (j->h_desc.size() % 16) j->h_desc.push_back(0);
            j->launches_generic.push_back(g);
            j->launches_tile.push_back(tl);
        }
    /* tile mode ping-pongs between the two scratch buffers: the level that runs k-th for a
     * plane reads LL from buffer (k-1) and writes buffer k (1 = t0, 2 = t1) */
    j->tile_ok = true;
    for (int t = 0; t < ntc; t++) {
        const J2kTileComp &tc = j->tilecomps[t];
        int k = 0;
        if (tc.coded)
            for (int lev = 0; lev < tc.ndeclevels; lev++) {
                if (tc.linelen[lev][0] > 0 && tc.linelen[lev][1] > 0) k++;
                else j->tile_ok = false;       /* a level deeper than the size allows: planes would fall out of step */
            }
        j->final_buf[t] = k == 0 ? 0 : 1 + ((k - 1) & 1);
    }
    j->final_eff.assign(ntc, 0);
    /* pack tiles; their source pointers are patched at launch time */
    while (j->h_desc.size() % 16) j->h_desc.push_back(0);
    j->pack_off = j->h_desc.size();
    j->npack = 0;
    j->pack_maxw = j->pack_maxh = 0;
    for (int f = 0; f < j->nframes; f++) {
        const FrameSlot &F = j->frames[f];
        const J2kPlan *pl = F.plan;
        for (int ti = 0; ti < pl->ntiles; ti++) {
            PackTile T;
            memset(&T, 0, sizeof(T));
            T.out = F.out;
            T.ncomp = pl->info.ncomponents;
            T.out_bytes = pl->out_bytes;
            T.precision = pl->out_shift_precision;
            for (int cc = 0; cc < T.ncomp; cc++) {
                const J2kTileComp &tc = j->tilecomps[F.tc_base + ti * T.ncomp + cc];
                PackComp &C = T.c[cc];
                C.src = nullptr;
                C.w = tc.w; C.h = tc.h; C.transform = tc.transform; C.cbps = tc.cbps;
                C.out_plane = tc.out_plane; C.out_x = tc.out_x; C.out_y = tc.out_y;
                C.pix_step = tc.pix_step; C.pix_off = tc.pix_off;
                if (tc.w > T.maxw) T.maxw = tc.w;
                if (tc.h > T.maxh) T.maxh = tc.h;
                if (cc == 0) T.mct = tc.mct;
            }
            if (T.maxw > j->pack_maxw) j->pack_maxw = T.maxw;
            if (T.maxh > j->pack_maxh) j->pack_maxh = T.maxh;
            push_bytes(j->h_desc, &T, sizeof(T));
            j->npack++;
        }
    }
    /* ---- fused plan (idwt_mode 3): the final level of every fusable tile goes through
     * k_idwt_stream_pack, which writes the frame; everything else as in launches_tile ---- */
    j->launches_fused.clear();
    j->tile_fusable.assign(j->npack, 0);
    j->plane_fused.assign(ntc, 0);
    j->any_fusable = false;
    struct Group { int tc0, nc, comp0, pack_tile; };
    std::vector<Group> groups;
    std::vector<uint8_t> in_group(ntc, 0);
    if (j->tile_ok) {
        int ti = 0;
        for (int f = 0; f < j->nframes; f++) {
            const FrameSlot &F = j->frames[f];
            const J2kPlan *pl = F.plan;
            const int nc = pl->info.ncomponents;
            for (int k = 0; k < pl->ntiles; k++, ti++) {
                const int base = F.tc_base + k * nc;
                auto final_ok = [&](const J2kTileComp &tc) {
                    if (!tc.coded || tc.ndeclevels < 1) return false;
                    const int lev = tc.ndeclevels - 1;
                    return tc.linelen[lev][0] >= 2 && tc.linelen[lev][1] >= 2 && tc.linelen[lev][0] == tc.w && tc.linelen[lev][1] == tc.h;
                };
                auto same = [&](const J2kTileComp &a, const J2kTileComp &b) {
                    const int lev = a.ndeclevels - 1;
                    return a.w == b.w && a.h == b.h && a.ndeclevels == b.ndeclevels && a.transform == b.transform &&
                           a.mod[lev][0] == b.mod[lev][0] && a.mod[lev][1] == b.mod[lev][1];
                };
                bool ok = nc >= 1 && nc <= 4;
                for (int cc = 0; cc < nc && ok; cc++) ok = final_ok(j->tilecomps[base + cc]);
                if (!ok) continue;
                const J2kTileComp &t0 = j->tilecomps[base];
                std::vector<Group> mine;
                if (t0.pix_step > 1) {
                    ok = nc == 3 || nc == 4;
                    for (int cc = 1; cc < nc && ok; cc++)
                        ok = same(t0, j->tilecomps[base + cc]) && j->tilecomps[base + cc].pix_step == t0.pix_step &&
                             j->tilecomps[base + cc].out_plane == t0.out_plane;
                    mine.push_back({ base, nc, 0, ti });
                } else {
                    int first = 0;
                    if (t0.mct) {
                        ok = nc >= 3 && same(t0, j->tilecomps[base + 1]) && same(t0, j->tilecomps[base + 2]);
                        mine.push_back({ base, 3, 0, ti });
                        first = 3;
                    }
                    for (int cc = first; cc < nc; cc++) mine.push_back({ base + cc, 1, cc, ti });
                }
                if (!ok) continue;
                j->tile_fusable[ti] = 1;
                j->any_fusable = true;
                for (const Group &g : mine) {
                    groups.push_back(g);
                    for (int cc = 0; cc < g.nc; cc++) in_group[g.tc0 + cc] = 1;
                }
            }
        }
    }
    if (j->any_fusable) {
        auto level_args = [&](const J2kTileComp &tc, int lev) {
            DwtTileArgs a;
            a.g.plane_off = tc.plane_off; a.g.stride = tc.w;
            a.g.lh = tc.linelen[lev][0]; a.g.lv = tc.linelen[lev][1];
            a.g.mh = tc.mod[lev][0]; a.g.mv = tc.mod[lev][1];
            a.g.last = (tc.transform == J2K_DWT97_INT && lev == tc.ndeclevels - 1) ? 1 : 0;
            a.ll_off = tc.plane_off; a.ll_stride = tc.w;
            a.out_off = tc.plane_off; a.out_stride = tc.w;
            return a;
        };
        for (int lev = 0; lev < maxlev; lev++)
            for (int type = 0; type < 3; type++) {
                LevelLaunch g;
                g.type = type; g.level = lev; g.count = 0; g.max_lh = g.max_lv = 0; g.alg_bytes = 0; g.min_l = 1 << 30; g.nc = 0;
                g.all_fast = true;
                std::vector<DwtTileArgs> ta;
                for (int t = 0; t < ntc; t++) {
                    const J2kTileComp &tc = j->tilecomps[t];
                    if (!tc.coded || tc.transform != type || lev >= tc.ndeclevels) continue;
                    if (in_group[t] && lev == tc.ndeclevels - 1) continue;
                    const DwtTileArgs a = level_args(tc, lev);
                    ta.push_back(a);
                    g.max_lh = std::max(g.max_lh, a.g.lh); g.max_lv = std::max(g.max_lv, a.g.lv);
                    g.min_l = std::min(g.min_l, std::min(a.g.lh, a.g.lv));
                    g.alg_bytes += 8.0 * a.g.lh * a.g.lv;
                    g.all_fast = g.all_fast && stream_fast_geom(a.g);
                }
                if (!ta.empty()) {
                    g.count = (int)ta.size();
                    while (j->h_desc.size() % 16) j->h_desc.push_back(0);
                    g.table_off = j->h_desc.size();
                    push_bytes(j->h_desc, ta.data(), ta.size() * sizeof(DwtTileArgs));
                    j->launches_fused.push_back(g);
                }
                /* one launch per (components in the group, fast store kind): groups whose geometry and frame qualify for
                 * a fast store go to the FASTONLY kernel of that kind, the rest to the general kernel (outk -1) */
                for (int nc = 1; nc <= 4; nc++)
                for (int outk = -1; outk <= 3; outk++) {
                    LevelLaunch fz = g;
                    fz.count = 0; fz.max_lh = fz.max_lv = 0; fz.alg_bytes = 0; fz.hbm_bytes = 0; fz.min_l = 1 << 30; fz.nc = nc;
                    fz.all_fast = outk >= 0;
                    fz.outk = outk;
                    std::vector<DwtFusedArgs> fa;
                    for (const Group &gr : groups) {
                        const J2kTileComp &tc = j->tilecomps[gr.tc0];
                        if (gr.nc != nc || tc.transform != type || tc.ndeclevels - 1 != lev) continue;
                        {
                            const PackTile *PT0 = (const PackTile *)(j->h_desc.data() + j->pack_off) + gr.pack_tile;
                            const DwtLevel g0 = level_args(tc, lev).g;
                            int kind = stream_fast_geom(g0) ? stream_fast_outk(*PT0, g0, nc, gr.comp0) : -1;
                            if (kind > 0 && type == J2K_DWT97_INT) kind = -1;      /* rgb48 / plane stores: 5/3 and 9/7 float only */
                            if (kind != outk) continue;
                        }
                        DwtFusedArgs A;
                        memset(&A, 0, sizeof(A));
                        for (int cc = 0; cc < nc; cc++) A.a[cc] = level_args(j->tilecomps[gr.tc0 + cc], lev);
                        A.ncomp = nc; A.pack_tile = gr.pack_tile; A.comp0 = gr.comp0;
                        fa.push_back(A);
                        fz.max_lh = std::max(fz.max_lh, A.a[0].g.lh); fz.max_lv = std::max(fz.max_lv, A.a[0].g.lv);
                        fz.min_l = std::min(fz.min_l, std::min(A.a[0].g.lh, A.a[0].g.lv));
                        const PackTile *PT = (const PackTile *)(j->h_desc.data() + j->pack_off) + gr.pack_tile;
                        fz.alg_bytes += (double)nc * A.a[0].g.lh * A.a[0].g.lv * 8.0;
                        fz.hbm_bytes += (double)nc * A.a[0].g.lh * A.a[0].g.lv * (4.0 + PT->out_bytes);
                    }
                    if (fa.empty()) continue;
                    fz.count = (int)fa.size();
                    while (j->h_desc.size() % 16) j->h_desc.push_back(0);
                    fz.table_off = j->h_desc.size();
                    push_bytes(j->h_desc, fa.data(), fa.size() * sizeof(DwtFusedArgs));
                    j->launches_fused.push_back(fz);
                }
            }
        for (int t = 0; t < ntc; t++) j->plane_fused[t] = in_group[t];
    }
    /* levels 0-2 of the reversible planes as one launch: the three plain launches must list the same planes (none of them has
     * its final level there), every level of fast geometry, every plane at the origin of its level */
    j->x3_ok = false;
    {
        const LevelLaunch *L3[3] = { nullptr, nullptr, nullptr };
        for (const LevelLaunch &L : j->launches_fused)
            if (L.type == J2K_DWT53 && L.nc == 0 && L.level < 3) L3[L.level] = &L;
        if (L3[0] && L3[1] && L3[2] && L3[0]->count == L3[1]->count && L3[1]->count == L3[2]->count) {
            bool ok = true;
            double alg = 0, hbm = 0;
            for (int k = 0; k < 3 && ok; k++) {
                const DwtTileArgs *A = (const DwtTileArgs *)(j->h_desc.data() + L3[k]->table_off);
                for (int i = 0; i < L3[k]->count && ok; i++) {
                    ok = stream_fast_geom(A[i].g) && A[i].g.mh == 0 && A[i].g.mv == 0;
                    const double n = (double)A[i].g.lh * A[i].g.lv;
                    alg += 8.0 * n;
                    hbm += 2.0 * n * (k == 0 ? 1.0 : 0.75) + (k == 2 ? 2.0 * n : 0.0);   /* 16-bit: the sub-bands in, the third level out */
                }
                j->x3_tab[k] = L3[k]->table_off;
                j->x3_lh[k] = L3[k]->max_lh;
            }
            /* (the planes of a launch are listed in tile-component order by every level: same index, same plane) */
            j->x3_ok = ok;
            j->x3_count = L3[0]->count;
            j->x3_lv2 = L3[2]->max_lv;
            j->x3_alg = alg; j->x3_hbm = hbm;
        }
    }
    (void)c;
    return 0;
}

extern "C" int htj2k_job_upload(htj2k_ctx *c, htj2k_job *j)
{
    if (!c || !j || j->nframes <= 0) return HTJ2K_ERR_EINVAL;
    int r;
    HIP_TRY(c, hipSetDevice(c->device));
    {
        uint32_t max_p = 0, max_s = 2, max_qw = 1, bm_words = 0;
        for (int f = 0; f < j->nframes; f++) {
            const J2kPlan *pl = j->frames[f].plan;
            max_p = std::max(max_p, pl->max_pcup); max_s = std::max(max_s, pl->max_scup);
            max_qw = std::max(max_qw, pl->max_qw); bm_words = std::max(bm_words, pl->max_bm_words);
        }
        if ((r = ht_lds_layout(c, max_p, max_s, max_qw, bm_words, &j->lds, &j->lds_ext)) < 0) return r;
        j->max_qw = j->lds.max_qw;
        j->max_lref = 0;
        for (int f = 0; f < j->nframes; f++) j->max_lref = std::max(j->max_lref, j->frames[f].plan->max_lref);
    }
    {
        /* blocks are independent: order the table by quad count (then width), largest first, so that
         * the 64 lanes of a k_ht_vlc wave (one lane per block) run similar trip counts; then lay the
         * quad-symbol arrays out.  Two stable counting passes (least significant key first). */
        auto quads = [](const J2kBlock &b) { return b.npasses ? (uint32_t)((b.w + 1) >> 1) * ((b.h + 1) >> 1) : 0u; };
        const size_t nb = j->blocks.size();
        std::vector<J2kBlock> tmp(nb);
        {
            std::vector<uint32_t> cnt(1026, 0);
            for (const J2kBlock &b : j->blocks) cnt[1024 - std::min<uint32_t>(b.w, 1024) + 1]++;
            for (size_t i = 1; i < cnt.size(); i++) cnt[i] += cnt[i - 1];
            for (const J2kBlock &b : j->blocks) tmp[cnt[1024 - std::min<uint32_t>(b.w, 1024)]++] = b;
        }
        {
            uint32_t maxq = 0;
            for (const J2kBlock &b : tmp) maxq = std::max(maxq, quads(b));
            std::vector<uint32_t> cnt((size_t)maxq + 2, 0);
            for (const J2kBlock &b : tmp) cnt[maxq - quads(b) + 1]++;
            for (size_t i = 1; i < cnt.size(); i++) cnt[i] += cnt[i - 1];
            for (const J2kBlock &b : tmp) j->blocks[cnt[maxq - quads(b)]++] = b;
        }
        /* Part-1 blocks behind the HT blocks, ordered so that the 64 lanes of a k_mq_decode wave walk blocks of
         * one size with similar pass counts (the sort above already grouped sizes) */
        {
            auto part1 = [](const J2kBlock &b) { return (b.flags & J2K_BLK_PART1) != 0; };
            auto mid = std::stable_partition(j->blocks.begin(), j->blocks.end(), [&](const J2kBlock &b) { return !part1(b); });
            j->nht = (int)(mid - j->blocks.begin());
            std::stable_sort(mid, j->blocks.end(), [&](const J2kBlock &a, const J2kBlock &b) {
                if ((a.w > 64) != (b.w > 64)) return a.w > 64;     /* the waves that need k_mq_decode<true> come first */
                if (a.h != b.h) return a.h > b.h;
                if (a.w != b.w) return a.w > b.w;
                return a.npasses > b.npasses;
            });
            j->mqwaves.clear();
            j->mq_planes = 1;
            size_t units = 0;
            for (size_t i = (size_t)j->nht; i < j->blocks.size(); i += 64) {
                MqWave W;
                int hmax = 0, wmax = 0, pmax = 0;
                for (size_t k = i; k < std::min(i + 64, j->blocks.size()); k++) {
                    const J2kBlock &b = j->blocks[k];
                    hmax = std::max<int>(hmax, b.h); wmax = std::max<int>(wmax, b.w); pmax = std::max<int>(pmax, b.npasses);
                }
                W.hmax = (uint16_t)hmax; W.wmax = (uint16_t)wmax; W.pmax = (uint16_t)pmax;
                W.rows = (uint16_t)(((hmax + 3) & ~3) + 2);
                W.chunks = (uint16_t)((wmax + 63) / 64);
                W.pad = 0;
                if (units > 0xFFFFFF00ull) return HTJ2K_ERR_PATCHWELCOME;
                W.soff = (uint32_t)units;
                units += (size_t)(4 + std::min((pmax + 1) / 3 + 1, 32)) * W.rows * W.chunks;
                j->mq_planes = std::max<uint32_t>(j->mq_planes, (uint32_t)std::min((pmax + 1) / 3 + 1, 32));
                j->mqwaves.push_back(W);
            }
            j->mq_scratch_units = units;
        }
        j->qoff.resize(j->blocks.size() + 1);
        size_t q = 0;
        for (size_t i = 0; i < j->blocks.size(); i++) {
            j->qoff[i] = (uint32_t)q;
            const J2kBlock &b = j->blocks[i];
            if (b.npasses && !(b.flags & J2K_BLK_PART1)) q += ht_qsym_words(b.w, b.h);
        }
        if (q > 0xFFFFFF00ull) return HTJ2K_ERR_PATCHWELCOME;
        j->nquads = q;
        /* blocks with SigProp / MagRef passes go through k_ht_refine (one lane per block) when
         * k_ht_decode's row-mask path can take them: up to 64 columns, no ROI shift */
        const size_t nm = ref_layout(j->blocks.data(), j->blocks.size(), j->reflist, j->roff, &j->ref_max_w, &j->ref_max_h);
        if (nm > 0xFFFFFF00ull) return HTJ2K_ERR_PATCHWELCOME;
        j->nrefmasks = nm;
    }
    const size_t coef_bytes = (j->nsamples + 64) * sizeof(uint32_t);
    const int nblocks = (int)j->blocks.size();
    if ((r = j->d_bytes.ensure(j->nbytes + 256)) < 0) return r;        /* k_mq_decode's 128-byte windows start inside a block */
    if (!j->mqwaves.empty() &&
        ((r = j->d_mqwaves.ensure(j->mqwaves.size() * sizeof(MqWave))) < 0 ||
         (r = j->d_mqscratch.ensure(j->mq_scratch_units * 512 + 512)) < 0)) return r;
    if ((r = j->d_blocks.ensure((size_t)(nblocks + 1) * sizeof(J2kBlock))) < 0) return r;
    if ((r = j->d_status.ensure((size_t)(nblocks + 1) * sizeof(int))) < 0) return r;
    if ((r = j->d_coef.ensure(coef_bytes)) < 0) return r;
    if ((r = j->d_t0.ensure(coef_bytes)) < 0) return r;
    if ((r = j->d_t1.ensure(coef_bytes)) < 0) return r;
    /* + the scratch line of k_ht_vlc's flush, + what k_ht_decode_pair's symbol preload reads past the last block (32 rows of 34) */
    if ((r = j->d_qsym.ensure(j->nquads * sizeof(ht_sym_t) + 256 + 8192)) < 0) return r;
    if ((r = j->d_qoff.ensure((j->qoff.size() + 1) * sizeof(uint32_t))) < 0) return r;
    /* a corrupt block can run k_ht_vlc's bit positions past its own arrays (38 VLC / 18 MEL bits per quad pair at
     * most, 4096 samples per block: < 3 KB): the last block of the pool must still read inside the allocation */
    if ((r = j->d_vlcu.ensure(j->nbytes + 16384)) < 0 || (r = j->d_melu.ensure(j->nbytes + 16384)) < 0) return r;
    if ((r = j->d_reflist.ensure((j->reflist.size() + 1) * sizeof(uint32_t))) < 0 ||
        (r = j->d_roff.ensure(j->roff.size() * sizeof(uint32_t))) < 0 ||
        (r = j->d_refbits.ensure((j->nrefmasks + 8) * sizeof(uint64_t))) < 0) return r;
    for (int f = 0; f < j->nframes; f++) {
        FrameSlot &F = j->frames[f];
        const J2kPlan *pl = F.plan;
        memset(&F.out, 0, sizeof(F.out));
        for (int p = 0; p < pl->info.nplanes; p++) {
            const int ls = pl->info.plane_width[p] * pl->info.plane_bytes_per_sample[p];
            if ((r = F.d_out[p].ensure((size_t)ls * pl->info.plane_height[p] + 64)) < 0) return r;
            F.out.ptr[p] = (uint8_t *)F.d_out[p].p;
            F.out.linesize[p] = ls;
            F.out.width[p] = pl->info.plane_width[p];
            F.out.height[p] = pl->info.plane_height[p];
        }
        /* Samples of the picture that no tile-component covers (image offsets with subsampling can leave a chroma row
         * or column out, write_frame_8/16 jpeg2000dec.c:2312-2358) are never written, in the reference either; clear
         * such planes so that what the caller gets there does not depend on what the buffer held before */
        for (int p = 0; p < pl->info.nplanes && p < 4; p++) {
            if (pl->info.has_palette && p == 1) continue;
            long long covered = 0;
            int first = -1;                                      /* packed formats: the components share the pixels */
            for (int t = 0; t < pl->ntilecomps; t++)
                if (pl->tilecomps[t].out_plane == p && (first < 0 || pl->tilecomps[t].comp < first)) first = pl->tilecomps[t].comp;
            for (int t = 0; t < pl->ntilecomps; t++) {
                const J2kTileComp &tc = pl->tilecomps[t];
                if (tc.out_plane == p && tc.comp == first && tc.out_w > 0 && tc.out_h > 0) covered += (long long)tc.out_w * tc.out_h;
            }
            const long long need = (long long)pl->info.plane_width[p] * pl->info.plane_height[p];
            if (covered < need)
                HIP_TRY(c, hipMemsetAsync(F.d_out[p].p, 0, (size_t)F.out.linesize[p] * F.out.height[p], j->stream));
        }
    }
    if ((r = build_descriptors(c, j)) < 0) return r;
    {
        bool ok = j->nht == (int)j->blocks.size() && j->reflist.empty() && j->tile_ok && j->any_fusable && !j->blocks.empty();
        for (size_t t = 0; ok && t < j->tilecomps.size(); t++) {
            const J2kTileComp &tc = j->tilecomps[t];
            ok = tc.coded && tc.transform == J2K_DWT53 && tc.ndeclevels >= 1;
        }
        for (size_t i = 0; ok && i < j->blocks.size(); i++) {
            const J2kBlock &b = j->blocks[i];
            ok = b.w <= 64 && b.M_b <= 15 && b.roi_shift == 0 && b.i_step == 32768 && (b.flags & 3) == J2K_DWT53;
            if (ok && b.npasses) { const int rem = b.npasses % 3; ok = b.npasses - (rem ? b.npasses - rem : b.npasses - 3) == 1; }
        }
        for (size_t i = 0; ok && i < j->launches_fused.size(); i++) {
            const LevelLaunch &L = j->launches_fused[i];
            ok = L.all_fast && L.type == J2K_DWT53 && L.min_l >= 2;
        }
        for (size_t i = 0; ok && i < j->tile_fusable.size(); i++) ok = j->tile_fusable[i] != 0;
        j->coef16_ok = ok;
        pk16_eligibility(j);
        for (size_t i = 0; ok && i < j->blocks.size(); i++) {
            const J2kBlock &b = j->blocks[i];
            ok = !(b.w & 1) && !(b.stride & 1) && !(b.plane_off & 1);
        }
        j->pair_ok = ok;
        /* k_ht_decode_multi: every block an HT block with the cleanup pass only, at most 64 columns, no ROI shift, one
         * transform for all of them; four blocks per wave when none is wider than 32 columns */
        /* (blocks with SigProp / MagRef passes of such jobs are all on k_ht_refine's list: same conditions) */
        bool mok = j->nht == (int)j->blocks.size() && !j->blocks.empty();
        int maxw = 0;
        const int t0 = mok ? (j->blocks[0].flags & 3) : 0;
        for (size_t i = 0; mok && i < j->blocks.size(); i++) {
            const J2kBlock &b = j->blocks[i];
            mok = b.w <= 64 && b.roi_shift == 0 && (b.flags & 3) == t0;
            if (b.w > maxw) maxw = b.w;
        }
        j->multi_nb = mok ? (maxw <= 32 ? 4 : 2) : 0;
        j->multi_t = t0;
        /* the kernel holds the un-stuffed MagSgn bits of all its blocks in LDS: with long segments (16-bit material: 10 KB
         * per 64 x 64 block) two blocks per wave leave 2 waves per SIMD and the column-per-lane kernel is quicker
         * (C4 gray16 1.50 -> 1.64 ms per 16 frames), with short ones it is not (int32 C2 2.93 -> 2.74, C3 2.15 -> 1.65) */
        const size_t multi_lds_max = getenv("HTJ2K_MULTI_LDS") ? (size_t)atoi(getenv("HTJ2K_MULTI_LDS")) : 12 * 1024;
        if (mok && (size_t)j->multi_nb * (j->lds_ext.ms_words + 4) * 4 > multi_lds_max) j->multi_nb = 0;
        /* blocks with refinement passes: their SigProp masks and MagRef bits are staged in LDS, up to 64 sample rows */
        if (j->multi_nb && !j->reflist.empty() &&
            (j->ref_max_h > 64 || ht_multi_refine_lds(j->multi_nb, ht_nsp(j->max_lref), j->ref_max_h) > 24 * 1024)) j->multi_nb = 0;
    }
    if ((r = j->d_desc.ensure(j->h_desc.size() + 64)) < 0) return r;
    HIP_TRY(c, hipEventRecord(j->ev[0], j->stream));
    if (j->dev_gather) {
        if ((r = j->d_pkt.ensure(j->pkt_bytes + 256)) < 0 || (r = j->d_lit.ensure(j->lit.size() + 64)) < 0 ||
            (r = j->d_gsegs.ensure((j->gsegs.size() + 1) * sizeof(J2kSeg))) < 0 ||
            (r = j->d_ggroups.ensure((j->ggroups.size() + 1) * sizeof(uint32_t))) < 0) return r;
        if (!j->pkt_h2d_issued)
            for (int f = 0; f < j->nframes; f++)
                HIP_TRY(c, job_packet_h2d(j, j->frames[f], (size_t)j->frames[f].plan->pkt_size));
        j->pkt_h2d_issued = false;
        if (!j->lit.empty())
            HIP_TRY(c, hipMemcpyAsync(j->d_lit.p, j->lit.data(), j->lit.size(), hipMemcpyHostToDevice, j->stream));
        if (!j->gsegs.empty())
            HIP_TRY(c, hipMemcpyAsync(j->d_gsegs.p, j->gsegs.data(), j->gsegs.size() * sizeof(J2kSeg), hipMemcpyHostToDevice, j->stream));
        HIP_TRY(c, hipMemcpyAsync(j->d_ggroups.p, j->ggroups.data(), j->ggroups.size() * sizeof(uint32_t), hipMemcpyHostToDevice, j->stream));
        HIP_TRY(c, hipMemsetAsync(j->d_bytes.p, 0, j->nbytes + 256, j->stream));
        const int ngroups = (int)j->ggroups.size() - 1;
        if (ngroups > 0 && !j->gsegs.empty()) {
            hipLaunchKernelGGL(k_gather, dim3((ngroups + 3) / 4), dim3(256), 0, j->stream, (const J2kSeg *)j->d_gsegs.p,
                               (const uint32_t *)j->d_ggroups.p, ngroups, (const uint8_t *)j->d_pkt.p, (const uint8_t *)j->d_lit.p,
                               (uint8_t *)j->d_bytes.p);
            HIP_TRY(c, hipGetLastError());
        }
    } else {
        for (int f = 0; f < j->nframes; f++) {
            const FrameSlot &F = j->frames[f];
            if (F.plan->nbytes)
                HIP_TRY(c, hipMemcpyAsync((uint8_t *)j->d_bytes.p + F.bytes_base, F.plan->bytes, F.plan->nbytes,
                                          hipMemcpyHostToDevice, j->stream));
        }
    }
    for (int f = 0; f < j->nframes; f++) {                 /* PAL8: the palette is the second plane (jpeg2000dec.c:2900-2901) */
        const FrameSlot &F = j->frames[f];
        if (F.plan->info.has_palette && F.plan->info.nplanes > 1)
            HIP_TRY(c, hipMemcpyAsync(F.out.ptr[1], F.plan->palette, 256 * sizeof(uint32_t), hipMemcpyHostToDevice, j->stream));
    }
    if (nblocks) {
        HIP_TRY(c, hipMemcpyAsync(j->d_blocks.p, j->blocks.data(), (size_t)nblocks * sizeof(J2kBlock), hipMemcpyHostToDevice, j->stream));
        HIP_TRY(c, hipMemcpyAsync(j->d_qoff.p, j->qoff.data(), j->qoff.size() * sizeof(uint32_t), hipMemcpyHostToDevice, j->stream));
        HIP_TRY(c, hipMemcpyAsync(j->d_roff.p, j->roff.data(), j->roff.size() * sizeof(uint32_t), hipMemcpyHostToDevice, j->stream));
        if (!j->reflist.empty())
            HIP_TRY(c, hipMemcpyAsync(j->d_reflist.p, j->reflist.data(), j->reflist.size() * sizeof(uint32_t), hipMemcpyHostToDevice, j->stream));
        if (!j->mqwaves.empty())
            HIP_TRY(c, hipMemcpyAsync(j->d_mqwaves.p, j->mqwaves.data(), j->mqwaves.size() * sizeof(MqWave), hipMemcpyHostToDevice, j->stream));
    }
    HIP_TRY(c, hipEventRecord(j->ev[1], j->stream));
    j->uploaded = 1;
    return 0;
}

/* ------------------------------------------------------------------ job: run */
static uint32_t *buf_ptr(htj2k_job *j, int which)
{
    return (uint32_t *)(which == 0 ? j->d_coef.p : which == 1 ? j->d_t0.p : j->d_t1.p);
}

template <int TYPE>
static void launch_generic_level(htj2k_job *j, const LevelLaunch &L)
{
    const DwtLevel *tab = (const DwtLevel *)((uint8_t *)j->d_desc.p + L.table_off);
    dim3 gh((L.max_lh + 255) / 256, L.max_lv, L.count), gv((L.max_lh + 255) / 256, L.max_lv, L.count);
    hipLaunchKernelGGL((k_idwt_h<TYPE>), gh, dim3(256), 0, j->stream, tab, (const uint32_t *)j->d_coef.p, (uint32_t *)j->d_t0.p);
    hipLaunchKernelGGL((k_idwt_v<TYPE>), gv, dim3(256), 0, j->stream, tab, (const uint32_t *)j->d_t0.p, (uint32_t *)j->d_coef.p);
}

#define TILE_W 64
#define TILE_H 32

/* rows per wave of the streaming kernels.  Measured on the 4K batch (tools/gpu_strip.py): the
 * launches are bound by the memory pipeline, not by the HALO rows a strip re-reads (those hit in
 * the XCD's L2, see stream_strip()), and 12..18 rows per strip is the flat optimum -- many short
 * waves keep more loads in flight and leave a shorter tail than few long ones. */
static int stream_strip_rows(int max_lh, int max_lv, int count)
{
    const char *e = getenv("HTJ2K_STRIP");
    if (e && atoi(e) >= 8) return atoi(e) & ~1;
    const long cols = (max_lh + STREAM_TW - 1) / STREAM_TW;
    return cols * ((max_lv + 15) / 16) * count >= 2048 ? 16 : 8;
}

/* Output columns per wave.  out_bytes = bytes one output column takes in the row a strip stores (2 / 4: a 16- / 32-bit
 * LL band; 3 / 6 / 1 / 2: rgb24 / rgb48 / 8- / 16-bit plane of the fused final level; 0: general path). */
static int stream_strip_cols(int max_lh, int out_bytes)
{
    const char *e = getenv(out_bytes == 2 ? "HTJ2K_TW16" : out_bytes == 4 ? "HTJ2K_TW32" : "HTJ2K_TWF");
    if (e && atoi(e) >= 64 && atoi(e) <= STREAM_TW) return atoi(e) & ~3;
    /* 224 columns are 448 / 896 / 1344 bytes of 16-bit / 32-bit / rgb48 output: whole 64-byte pieces, so that two
     * neighbouring strips never write into the same piece of a line (244 columns end mid-line, and half of a 16-bit
     * strip's lines are shared with a neighbour).  It idles 5 more lanes of 64 and adds a strip per 11, which only pays
     * once a row has several strips: measured per level on C2 / C3 / C4, 16-bit rows from 1920
     * columns (288 -> 265 us at 1920, 67 -> 73 at 960), 32-bit rows from 960 (237 -> 221 us at 1920, 76 -> 66 at 960).
     * rgb24 rows (732 bytes per strip, never a multiple of 64 below 256 columns) keep the full width. */
    if (out_bytes == 2) return max_lh >= 1920 ? 224 : STREAM_TW;
    if (out_bytes == 4 || out_bytes == 6) return max_lh >= 960 ? 224 : STREAM_TW;
    return STREAM_TW;
}

static StreamGrid stream_grid(int max_lh, int max_lv, int th, int count, int tw = STREAM_TW, int wpb = 1)
{
    StreamGrid G;
    G.tw = tw;
    G.gx = (max_lh + tw - 1) / tw;
    G.gy = (max_lv + th - 1) / th;
    G.nstrips = G.gx * G.gy * count;
    G.per_xcd = ((G.nstrips + 7) / 8 + wpb - 1) / wpb * wpb;      /* launches of 8 * per_xcd / wpb workgroups of wpb waves */
    return G;
}
/* Waves per workgroup of the streaming kernels that take more than one (the 16-bit FASTONLY ones).  Eight neighbouring
 * strips -- half a row of a 4K plane -- as one workgroup walk down the same rows of neighbouring column ranges together on one
 * CU, so what they ask of a DRAM page comes close together in time: final level of C2 740 -> 711 us with the 32-bit arithmetic
 * (4 waves per SIMD), 774 -> 660 with the packed one at two such workgroups per CU (below); 6 or 10 waves, which do not divide
 * the 16 strips of a row, 718 / 724; 16 waves 700-707; level 4 250 -> 238-243 us. */
static int stream_wpb(int max_lh, int tw)
{
    const int e = getenv("HTJ2K_WPB") ? atoi(getenv("HTJ2K_WPB")) : 0;      /* (per launch: tools/gpu_idwt_ab.py flips it between runs) */
    if (e >= 1 && e <= 8) return e;
    return std::max(1, std::min(8, (max_lh + tw - 1) / tw));
}
/* Dynamic LDS the packed final-level kernel is launched with -- it uses none: it is there to keep the CU at 16 waves.  The
 * kernel needs 61 VGPRs and would run 8 waves per SIMD, and with that many strips in flight the memory system does worse:
 * C2 final level, one wave per workgroup, 792 us at 32 waves per CU, 754 at 20, 732 at 16, 718 at 12, 757 at 8; workgroups
 * of eight waves 713 us at three per CU, 660-683 at two, 739 at one. */
/* experiments: dynamic LDS (bytes per wave) for the other multi-wave streaming launches, as an occupancy limit */
static int stream_occ_lds(int wpb)
{
    const int e = getenv("HTJ2K_OCC_LDS") ? atoi(getenv("HTJ2K_OCC_LDS")) : 0;
    return std::min(e * wpb, 48 * 1024);
}
static int stream_pk_lds(int wpb)
{
    const int e = getenv("HTJ2K_PK_LDS") ? atoi(getenv("HTJ2K_PK_LDS")) : -1;
    if (e >= 0) return e;
    const int per_cu = wpb >= 8 ? 2 : wpb >= 4 ? 3 : 12 / wpb;          /* workgroups */
    return (160 * 1024 / per_cu) & ~1023;
}

template <int TYPE>
static void launch_tile_generic(const void *tab, int max_lh, int max_lv, int min_l, int count, int mode, hipStream_t s,
                                const uint32_t *ll, const uint32_t *band, uint32_t *out, bool all_fast, int coef16 = 0)
{
    if (mode >= 3 && min_l >= 2) {
        const int th = stream_strip_rows(max_lh, max_lv, count);
        const int tw = stream_strip_cols(max_lh, all_fast ? 4 : 0);
        const int wpb = all_fast || (TYPE == J2K_DWT53 && coef16) ? stream_wpb(max_lh, tw) : 1;
        const StreamGrid G = stream_grid(max_lh, max_lv, th, count, tw, wpb);
        const dim3 g(8 * G.per_xcd / wpb), blk(64 * wpb);
        const int lds = stream_occ_lds(wpb);
        if (TYPE == J2K_DWT53 && coef16 == 2) hipLaunchKernelGGL((k_idwt_stream<J2K_DWT53, true, true, true>), g, blk, lds, s, (const DwtTileArgs *)tab, ll, band, out, th, G);
        else if (TYPE == J2K_DWT53 && coef16 == 1) hipLaunchKernelGGL((k_idwt_stream<J2K_DWT53, true, true, false>), g, blk, lds, s, (const DwtTileArgs *)tab, ll, band, out, th, G);
        else if (all_fast) hipLaunchKernelGGL((k_idwt_stream<TYPE, true>), g, blk, lds, s, (const DwtTileArgs *)tab, ll, band, out, th, G);
        else hipLaunchKernelGGL((k_idwt_stream<TYPE, false>), g, dim3(64), 0, s, (const DwtTileArgs *)tab, ll, band, out, th, G);
    } else {
        dim3 g((max_lh + TILE_W - 1) / TILE_W, (max_lv + TILE_H - 1) / TILE_H, count);
        hipLaunchKernelGGL((k_idwt_tile<TYPE, TILE_W, TILE_H>), g, dim3(256), 0, s, (const DwtTileArgs *)tab, ll, band, out);
    }
}

/* the un-stuffing kernel with `g` codeblocks per wavefront (4, 2 or 1; fewer if the LDS of g blocks would not fit) */
static hipError_t launch_unstuff(hipStream_t st, const J2kBlock *blocks, int nblocks, const uint8_t *bytes, uint32_t *vlcu, uint32_t *melu,
                                 uint32_t us_words, int g)
{
    const size_t us_lds = (size_t)us_words * 4;
    hipError_t e = hipSuccess;
    if (g >= 4 && 4 * us_lds <= 64 * 1024) {
        if (4 * us_lds > 48 * 1024) e = hipFuncSetAttribute((const void *)k_ht_unstuff_g<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * us_lds));
        if (e == hipSuccess) hipLaunchKernelGGL(k_ht_unstuff_g<16>, dim3((nblocks + 3) / 4), dim3(64), 4 * us_lds, st, blocks, nblocks, bytes, vlcu, melu, us_words);
    } else if (g >= 2 && 2 * us_lds <= 64 * 1024) {
        if (2 * us_lds > 48 * 1024) e = hipFuncSetAttribute((const void *)k_ht_unstuff_g<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * us_lds));
        if (e == hipSuccess) hipLaunchKernelGGL(k_ht_unstuff_g<32>, dim3((nblocks + 1) / 2), dim3(64), 2 * us_lds, st, blocks, nblocks, bytes, vlcu, melu, us_words);
    } else {
        if (us_lds > 48 * 1024) e = hipFuncSetAttribute((const void *)k_ht_unstuff, hipFuncAttributeMaxDynamicSharedMemorySize, (int)us_lds);
        if (e == hipSuccess) hipLaunchKernelGGL(k_ht_unstuff, dim3(nblocks), dim3(64), us_lds, st, blocks, nblocks, bytes, vlcu, melu, us_words);
    }
    return e;
}

/* how many bits the 16-bit LL bands of a job must fit: 16, fewer where a packed final level needs it (pk16_eligibility) or a test says so */
static int ll16_check_bits(const htj2k_ctx *c, const htj2k_job *j)
{
    return std::min(c->ll16_test_bits, c->idwt_pk && j->pk_bits ? j->pk_bits : 16);
}

template <int TYPE>
static void launch_tile_level(htj2k_ctx *c, htj2k_job *j, const LevelLaunch &L, const uint32_t *ll, uint32_t *out)
{
    if (TYPE == J2K_DWT53 && j->ll16_run && c->idwt_mode >= 3 && L.min_l >= 2) {    /* 16-bit sub-bands and LL bands, in and out */
        const int th = stream_strip_rows(L.max_lh, L.max_lv, L.count);
        const int tw = stream_strip_cols(L.max_lh, 2), wpb = stream_wpb(L.max_lh, tw);
        const StreamGrid G = stream_grid(L.max_lh, L.max_lv, th, L.count, tw, wpb);
        if (L.pk_bits && j->pk_run)
            hipLaunchKernelGGL(k_idwt_stream_ll16<true>, dim3(8 * G.per_xcd / wpb), dim3(64 * wpb), 0, j->stream,
                               (const DwtTileArgs *)((uint8_t *)j->d_desc.p + L.table_off), ll, (const uint32_t *)j->d_coef.p, out, th, G,
                               (int *)j->d_status.p + j->blocks.size(), ll16_check_bits(c, j));
        else
            hipLaunchKernelGGL(k_idwt_stream_ll16<false>, dim3(8 * G.per_xcd / wpb), dim3(64 * wpb), 0, j->stream,
                               (const DwtTileArgs *)((uint8_t *)j->d_desc.p + L.table_off), ll, (const uint32_t *)j->d_coef.p, out, th, G,
                               (int *)j->d_status.p + j->blocks.size(), ll16_check_bits(c, j));
        return;
    }
    launch_tile_generic<TYPE>((uint8_t *)j->d_desc.p + L.table_off, L.max_lh, L.max_lv, L.min_l, L.count, c->idwt_mode,
                              j->stream, ll, (const uint32_t *)j->d_coef.p, out, L.all_fast,
                              j->coef_is16 ? (L.level == 0 ? 2 : 1) : 0);   /* 16-bit sub-bands; at level 0 the LL band too */
}

template <int TYPE>
static void launch_fused_level(htj2k_job *j, const LevelLaunch &L, const uint32_t *ll)
{
    const int th = stream_strip_rows(L.max_lh, L.max_lv, L.count);
    const int tw = stream_strip_cols(L.max_lh, L.outk == 0 ? 3 : L.outk == 1 ? 6 : L.outk == 2 ? 1 : L.outk == 3 ? 2 : 0);
    const bool multi = (TYPE == J2K_DWT53 && j->coef_is16) || (TYPE != J2K_DWT97_INT && (L.outk >= 1 || (L.nc == 3 && L.all_fast)));   /* FASTONLY kernels */
    const int wpb = multi ? stream_wpb(L.max_lh, tw) : 1;
    const int occ_lds = stream_occ_lds(wpb);
    const StreamGrid G = stream_grid(L.max_lh, L.max_lv, th, L.count, tw, wpb);
    dim3 g(8 * G.per_xcd / wpb);
    const dim3 blk(64 * wpb);
    const DwtFusedArgs *tab = (const DwtFusedArgs *)((uint8_t *)j->d_desc.p + L.table_off);
    const PackTile *tiles = (const PackTile *)((uint8_t *)j->d_desc.p + j->pack_off);
    const uint32_t *band = (const uint32_t *)j->d_coef.p;
#define FUSED_FAST(NC_, C16_, LL16_, K_) hipLaunchKernelGGL((k_idwt_stream_pack<TYPE == J2K_DWT97_INT && (K_) ? J2K_DWT53 : TYPE, NC_, true, C16_, LL16_, K_>), g, blk, occ_lds, j->stream, tab, ll, band, tiles, th, G)
    if (TYPE == J2K_DWT53 && j->coef_is16) {             /* coef16_ok: every fused launch is a fast-store one */
        const bool l0 = L.level == 0 || j->ll16_run;      /* the LL band is 16-bit: the block decoder's, or the level below wrote it so */
        const bool pk = l0 && L.pk_bits && j->pk_run;   /* pairs of 16-bit samples all the way (pk16_eligibility) */
        const int pk_lds = pk ? stream_pk_lds(wpb) : 0;
        if (pk && (L.outk == 0 || L.outk == 2)) j->pk_used = L.level == 0 ? 16 : std::min(16, j->pk_bits);
        if (pk && L.outk == 0) {
            if (pk_lds > 48 * 1024)
                (void)hipFuncSetAttribute((const void *)k_idwt_stream_pack<J2K_DWT53, 3, true, true, true, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, pk_lds);
            hipLaunchKernelGGL((k_idwt_stream_pack<J2K_DWT53, 3, true, true, true, 0, true>), g, blk, pk_lds, j->stream, tab, ll, band, tiles, th, G);
        } else if (pk && L.outk == 2) {
            if (pk_lds > 48 * 1024)
                (void)hipFuncSetAttribute((const void *)k_idwt_stream_pack<J2K_DWT53, 1, true, true, true, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, pk_lds);
            hipLaunchKernelGGL((k_idwt_stream_pack<J2K_DWT53, 1, true, true, true, 2, true>), g, blk, pk_lds, j->stream, tab, ll, band, tiles, th, G);
        }
        else if (L.outk == 0) { if (l0) FUSED_FAST(3, true, true, 0); else FUSED_FAST(3, true, false, 0); }
        else if (L.outk == 1) { if (l0) FUSED_FAST(3, true, true, 1); else FUSED_FAST(3, true, false, 1); }
        else if (L.outk == 2) { if (l0) FUSED_FAST(1, true, true, 2); else FUSED_FAST(1, true, false, 2); }
        else { if (l0) FUSED_FAST(1, true, true, 3); else FUSED_FAST(1, true, false, 3); }
    } else if (L.outk == 1) FUSED_FAST(3, false, false, 1);
    else if (L.outk == 2) FUSED_FAST(1, false, false, 2);
    else if (L.outk == 3) FUSED_FAST(1, false, false, 3);
    else if (L.nc == 1) hipLaunchKernelGGL((k_idwt_stream_pack<TYPE, 1, false>), g, dim3(64), 0, j->stream, tab, ll, band, tiles, th, G);
    else if (L.nc == 3 && L.all_fast) hipLaunchKernelGGL((k_idwt_stream_pack<TYPE, 3, true>), g, blk, occ_lds, j->stream, tab, ll, band, tiles, th, G);
    else if (L.nc == 3) hipLaunchKernelGGL((k_idwt_stream_pack<TYPE, 3, false>), g, dim3(64), 0, j->stream, tab, ll, band, tiles, th, G);
    else hipLaunchKernelGGL((k_idwt_stream_pack<TYPE, 4, false>), g, dim3(64), 0, j->stream, tab, ll, band, tiles, th, G);
}

static hipEvent_t lev_event(htj2k_job *j)
{
    if (j->lev_ev_used >= (int)j->lev_ev.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        j->lev_ev.push_back(e);
    }
    return j->lev_ev[j->lev_ev_used++];
}

static int run_idwt(htj2k_ctx *c, htj2k_job *j, bool use_tile, bool fuse)
{
    j->lev_ev_used = 0;
    j->lev_bytes.clear();
    j->lev_hbm.clear();
    j->pk_run = c->idwt_pk != 0;
    j->pk_used = 0;
    const std::vector<LevelLaunch> &LL = fuse ? j->launches_fused : use_tile ? j->launches_tile : j->launches_generic;
    bool x3 = false;
    if (fuse && j->ll16_run && j->x3_ok && c->idwt_x3) {
        const int th = getenv("HTJ2K_X3_TH") ? std::max(8, atoi(getenv("HTJ2K_X3_TH")) & ~3) : 24;   /* C2: 91 us at 24 rows, 93 at 36, 97 at 16 or 48, 104 at 68 (three launches: 123) */
        const size_t lds = ((size_t)x3_win0_rows(th) * j->x3_lh[0] + (size_t)x3_win1_rows(th) * j->x3_lh[1]) * sizeof(uint16_t);
        if (lds <= 64 * 1024) {
            x3 = true;
            const int nbands = (j->x3_lv2 + th - 1) / th, total = nbands * j->x3_count;
            hipEvent_t e0 = lev_event(j);
            if (e0) (void)hipEventRecord(e0, j->stream);
            bool x3pk = j->pk_run;                            /* all three levels on pairs of 16-bit samples, or none */
            for (const LevelLaunch &L : LL)
                if (L.type == J2K_DWT53 && L.nc == 0 && L.level < 3 && !L.pk_bits) x3pk = false;
            auto kern = x3pk ? k_idwt_stream_ll16_x3<true> : k_idwt_stream_ll16_x3<false>;
            if (lds > 48 * 1024)
                (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(8 * ((total + 7) / 8)), dim3(256), lds, j->stream,
                               (const DwtTileArgs *)((uint8_t *)j->d_desc.p + j->x3_tab[0]), (const DwtTileArgs *)((uint8_t *)j->d_desc.p + j->x3_tab[1]),
                               (const DwtTileArgs *)((uint8_t *)j->d_desc.p + j->x3_tab[2]), (const uint32_t *)j->d_coef.p, buf_ptr(j, 1),
                               th, nbands, j->x3_count, j->x3_lh[0], j->x3_lh[1], (int *)j->d_status.p + j->blocks.size(), ll16_check_bits(c, j));
            hipEvent_t e1 = lev_event(j);
            if (e1) (void)hipEventRecord(e1, j->stream);
            j->lev_bytes.push_back(j->x3_alg);
            j->lev_hbm.push_back(j->x3_hbm);
        }
    }
    for (const LevelLaunch &L : LL) {
        if (x3 && L.type == J2K_DWT53 && L.nc == 0 && L.level < 3) continue;     /* done by k_idwt_stream_ll16_x3 */
        hipEvent_t e0 = lev_event(j);
        if (e0) (void)hipEventRecord(e0, j->stream);
        if (L.nc) {
            const uint32_t *ll = buf_ptr(j, L.level == 0 ? 0 : 1 + ((L.level - 1) & 1));
            if (L.type == J2K_DWT53) launch_fused_level<J2K_DWT53>(j, L, ll);
            else if (L.type == J2K_DWT97) launch_fused_level<J2K_DWT97>(j, L, ll);
            else launch_fused_level<J2K_DWT97_INT>(j, L, ll);
        } else if (!use_tile) {
            if (L.type == J2K_DWT53) launch_generic_level<J2K_DWT53>(j, L);
            else if (L.type == J2K_DWT97) launch_generic_level<J2K_DWT97>(j, L);
            else launch_generic_level<J2K_DWT97_INT>(j, L);
        } else {
            const uint32_t *ll = buf_ptr(j, L.level == 0 ? 0 : 1 + ((L.level - 1) & 1));
            uint32_t *out = buf_ptr(j, 1 + (L.level & 1));
            if (L.type == J2K_DWT53) launch_tile_level<J2K_DWT53>(c, j, L, ll, out);
            else if (L.type == J2K_DWT97) launch_tile_level<J2K_DWT97>(c, j, L, ll, out);
            else launch_tile_level<J2K_DWT97_INT>(c, j, L, ll, out);
        }
        hipEvent_t e1 = lev_event(j);
        if (e1) (void)hipEventRecord(e1, j->stream);
        j->lev_bytes.push_back(L.alg_bytes);
        /* bytes that have to move: a level reads its LL quarter and three sub-band quarters and writes every sample
         * (alg_bytes = 8 per sample); the fused level writes the packed pixels instead (hbm_bytes = 4 + out per
         * sample).  With 16-bit sub-bands three quarters of the reads are 2 bytes (all of them at level 0). */
        double hb = L.nc ? L.hbm_bytes : L.alg_bytes;
        if (j->coef_is16) hb -= L.alg_bytes / 8.0 * ((L.level == 0 || j->ll16_run) ? 2.0 : 1.5);
        if (j->ll16_run && !L.nc) hb -= L.alg_bytes / 8.0 * 2.0;            /* ... and writes 2 instead of 4 bytes per sample */
        j->lev_hbm.push_back(hb);
    }
    return 0;
}

extern "C" int htj2k_job_run_stages(htj2k_ctx *c, htj2k_job *j, int mask)
{
    if (!c || !j || j->nframes <= 0 || !j->uploaded) return HTJ2K_ERR_EINVAL;
    const int nall = (int)j->blocks.size(), ntc = (int)j->tilecomps.size();
    const int nblocks = j->nht;                        /* HT blocks; the Part-1 blocks follow */
    HIP_TRY(c, hipSetDevice(c->device));
    if (mask & 1) {
        HIP_TRY(c, hipEventRecord(j->ev[2], j->stream));
        if (nall) HIP_TRY(c, hipMemsetAsync(j->d_status.p, 0, ((size_t)nall + 1) * sizeof(int), j->stream));   /* + the LL overflow flag */
        if (nall > nblocks) {
            const uint32_t area = mq_lds_area(j->mq_planes);
            unsigned nwide = 0;                                      /* waves holding a block wider than 64 columns */
            while (nwide < j->mqwaves.size() && j->mqwaves[nwide].chunks > 1) nwide++;
            const unsigned nnarrow = (unsigned)j->mqwaves.size() - nwide;
            if (nwide)
                hipLaunchKernelGGL(k_mq_decode<true>, dim3(nwide), dim3(64), area + MQ_LDS_TABLES, j->stream,
                                   (const J2kBlock *)j->d_blocks.p + nblocks, nall - nblocks, (const uint8_t *)j->d_bytes.p,
                                   (uint32_t *)j->d_coef.p, (int *)j->d_status.p + nblocks, (const MqWave *)j->d_mqwaves.p,
                                   (uint64_t *)j->d_mqscratch.p, area);
            if (nnarrow)
                hipLaunchKernelGGL(k_mq_decode<false>, dim3(nnarrow), dim3(64), area + MQ_LDS_TABLES, j->stream,
                                   (const J2kBlock *)j->d_blocks.p + nblocks + 64 * (size_t)nwide, nall - nblocks - 64 * (int)nwide,
                                   (const uint8_t *)j->d_bytes.p, (uint32_t *)j->d_coef.p, (int *)j->d_status.p + nblocks + 64 * (size_t)nwide,
                                   (const MqWave *)j->d_mqwaves.p + nwide, (uint64_t *)j->d_mqscratch.p, area);
            HIP_TRY(c, hipGetLastError());
        }
        j->coef_is16 = false;
        if (nblocks) {
            const bool vlc_narrow = j->max_qw <= 32;               /* no block wider than 64 columns: k_ht_vlc2 */
            const size_t vlc_lds = vlc_narrow ? (size_t)HT_VLC2_LDS : ht_vlc_lds_bytes(j->max_qw);
            /* 16-bit sub-bands only when this very call also runs the (fused, streaming) IDWT that reads them */
            j->coef_is16 = c->coef16 && j->coef16_ok && c->ht_mode == 1 && vlc_lds <= 160 * 1024 && mask == 7 &&
                           c->idwt_mode == 3 && c->fuse_pack;
            if (c->ht_mode == 1 && vlc_lds <= 160 * 1024) {
                if (vlc_lds > 48 * 1024)
                    HIP_TRY(c, hipFuncSetAttribute((const void *)k_ht_vlc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)vlc_lds));
                if ((int)j->lds_ext.total > 48 * 1024)
                    HIP_TRY(c, hipFuncSetAttribute((const void *)k_ht_decode<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)j->lds_ext.total));
                const uint32_t us_words = 2 * std::max(j->lds.vlc_words, j->reflist.empty() ? 0u : ht_nsp(j->max_lref));
                /* blocks per wave of the un-stuffing kernel: four (16 lanes each) where no block is wider than 32 columns, else
                 * two -- per 128 frames of C2 (64 x 64 blocks) 404 us with one, 378 with two, 456 with four (more passes, each
                 * with its fixed cost); per 96 frames of C3 (32 x 32) 820 / 529 / 451 */
                const int us_g = getenv("HTJ2K_UNSTUFF_G") ? atoi(getenv("HTJ2K_UNSTUFF_G")) : (j->max_qw <= 16 ? 4 : 2);
                HIP_TRY(c, launch_unstuff(j->stream, (const J2kBlock *)j->d_blocks.p, nblocks, (const uint8_t *)j->d_bytes.p,
                                          (uint32_t *)j->d_vlcu.p, (uint32_t *)j->d_melu.p, us_words, us_g));
                const int vlc_wg = vlc_narrow ? 64 * HT_VLC_NARROW_WAVES : 64;
                if (vlc_narrow)
                    hipLaunchKernelGGL(k_ht_vlc2, dim3((nblocks + vlc_wg - 1) / vlc_wg), dim3(vlc_wg), HT_VLC2_LDS, j->stream,
                                       (const J2kBlock *)j->d_blocks.p, nblocks, (const uint8_t *)j->d_bytes.p,
                                       (const uint16_t *)c->d_tables, (ht_sym_t *)j->d_qsym.p, (const uint32_t *)j->d_qoff.p,
                                       (const uint32_t *)j->d_vlcu.p, (uint32_t *)j->d_qsym.p + j->nquads / 2 + 32);
                else
                hipLaunchKernelGGL(k_ht_vlc, dim3((nblocks + vlc_wg - 1) / vlc_wg), dim3(vlc_wg), vlc_lds, j->stream,
                                   (const J2kBlock *)j->d_blocks.p, nblocks, (const uint8_t *)j->d_bytes.p,
                                   (const uint16_t *)c->d_tables, (ht_sym_t *)j->d_qsym.p, (const uint32_t *)j->d_qoff.p, j->max_qw,
                                   (const uint32_t *)j->d_vlcu.p, (const uint32_t *)j->d_melu.p, (uint32_t *)j->d_qsym.p + j->nquads / 2 + 32);
                const bool multi_run = !(j->coef_is16 && j->pair_ok && c->ht_pair) && !j->coef_is16 && j->multi_nb && c->ht_multi;
                if (!j->reflist.empty()) {
                    /* k_ht_decode_multi deals out the MagRef bits itself; the column-per-lane kernel takes them from k_ht_refine */
                    auto kref = j->ref_max_w <= 32 ? (multi_run ? k_ht_refine<uint32_t, false> : k_ht_refine<uint32_t, true>)
                                                   : (multi_run ? k_ht_refine<uint64_t, false> : k_ht_refine<uint64_t, true>);
                    hipLaunchKernelGGL(kref, dim3(((unsigned)j->reflist.size() + 63) / 64), dim3(64), 0, j->stream,
                                       (const J2kBlock *)j->d_blocks.p, (const uint32_t *)j->d_reflist.p, (int)j->reflist.size(),
                                       (const uint8_t *)j->d_bytes.p, (const ht_sym_t *)j->d_qsym.p, (const uint32_t *)j->d_qoff.p,
                                       (const uint32_t *)j->d_vlcu.p, (const uint32_t *)j->d_melu.p,
                                       (uint64_t *)j->d_refbits.p, (const uint32_t *)j->d_roff.p);
                }
                j->ht_bpw = (j->coef_is16 && j->pair_ok && c->ht_pair) ? 2 : (!j->coef_is16 && j->multi_nb && c->ht_multi) ? j->multi_nb : 1;
                if (j->coef_is16 && j->pair_ok && c->ht_pair)
                    hipLaunchKernelGGL(k_ht_decode_pair, dim3((nblocks + 1) / 2), dim3(64), 2 * (j->lds_ext.ms_words + 4) * 4, j->stream,
                                       (const J2kBlock *)j->d_blocks.p, nblocks, (const uint8_t *)j->d_bytes.p,
                                       (uint32_t *)j->d_coef.p, (int *)j->d_status.p, j->lds_ext.ms_words,
                                       (const ht_sym_t *)j->d_qsym.p, (const uint32_t *)j->d_qoff.p,
                                       (uint32_t *)j->d_coef.p + j->nsamples + 32);
                else if (!j->coef_is16 && j->multi_nb && c->ht_multi) {
                    const int nb = j->multi_nb;
                    const uint32_t mr_words = j->reflist.empty() ? 0u : ht_nsp(j->max_lref), rg_rows = j->reflist.empty() ? 0u : j->ref_max_h;
                    const size_t lds = (size_t)nb * (j->lds_ext.ms_words + 4) * 4 + (j->reflist.empty() ? 0 : ht_multi_refine_lds(nb, mr_words, rg_rows));
#define HT_MULTI_R(NB_, T_, R_) do { \
                        if (lds > 48 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void *)k_ht_decode_multi<NB_, T_, R_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                        hipLaunchKernelGGL((k_ht_decode_multi<NB_, T_, R_>), dim3((nblocks + NB_ - 1) / NB_), dim3(64), lds, j->stream, \
                                           (const J2kBlock *)j->d_blocks.p, nblocks, (const uint8_t *)j->d_bytes.p, \
                                           (uint32_t *)j->d_coef.p, (int *)j->d_status.p, j->lds_ext.ms_words, \
                                           (const ht_sym_t *)j->d_qsym.p, (const uint32_t *)j->d_qoff.p, \
                                           (uint32_t *)j->d_coef.p + j->nsamples + 32, \
                                           (const uint64_t *)j->d_refbits.p, (const uint32_t *)j->d_roff.p, \
                                           (const uint32_t *)j->d_melu.p, mr_words, rg_rows); } while (0)
#define HT_MULTI(NB_, T_) do { if (j->reflist.empty()) HT_MULTI_R(NB_, T_, false); else HT_MULTI_R(NB_, T_, true); } while (0)
                    if (nb == 4) {
                        if (j->multi_t == J2K_DWT53) HT_MULTI(4, J2K_DWT53); else if (j->multi_t == J2K_DWT97) HT_MULTI(4, J2K_DWT97); else HT_MULTI(4, J2K_DWT97_INT);
                    } else {
                        if (j->multi_t == J2K_DWT53) HT_MULTI(2, J2K_DWT53); else if (j->multi_t == J2K_DWT97) HT_MULTI(2, J2K_DWT97); else HT_MULTI(2, J2K_DWT97_INT);
                    }
#undef HT_MULTI
#undef HT_MULTI_R
                } else
                hipLaunchKernelGGL(k_ht_decode<true>, dim3(nblocks), dim3(64), j->lds_ext.total, j->stream,
                                   (const J2kBlock *)j->d_blocks.p, nblocks, (const uint8_t *)j->d_bytes.p,
                                   (uint32_t *)j->d_coef.p, (const uint16_t *)c->d_tables, (int *)j->d_status.p, j->lds_ext,
                                   (const ht_sym_t *)j->d_qsym.p, (const uint32_t *)j->d_qoff.p,
                                   (uint32_t *)j->d_coef.p + j->nsamples + 32,
                                   (const uint64_t *)j->d_refbits.p, (const uint32_t *)j->d_roff.p, j->coef_is16 ? 1 : 0);
            } else {
                j->ht_bpw = 1;
                if ((int)j->lds.total > 48 * 1024)
                    HIP_TRY(c, hipFuncSetAttribute((const void *)k_ht_decode<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)j->lds.total));
                hipLaunchKernelGGL(k_ht_decode<false>, dim3(nblocks), dim3(64), j->lds.total, j->stream,
                                   (const J2kBlock *)j->d_blocks.p, nblocks, (const uint8_t *)j->d_bytes.p,
                                   (uint32_t *)j->d_coef.p, (const uint16_t *)c->d_tables, (int *)j->d_status.p, j->lds,
                                   (const ht_sym_t *)nullptr, (const uint32_t *)nullptr, (uint32_t *)j->d_coef.p + j->nsamples + 32,
                                   (const uint64_t *)nullptr, (const uint32_t *)nullptr);
            }
            HIP_TRY(c, hipGetLastError());
        }
        HIP_TRY(c, hipEventRecord(j->ev[3], j->stream));
    }
    if (mask & 6) {
        const bool use_tile = c->idwt_mode != 0 && j->tile_ok;
        const bool fuse = use_tile && c->idwt_mode == 3 && c->fuse_pack && (mask & 6) == 6 && j->any_fusable;
        if (mask & 2)
            for (int t = 0; t < ntc; t++) j->final_eff[t] = use_tile ? j->final_buf[t] : 0;
        j->fused_last = fuse;
        /* one upload of all descriptor tables, pack source pointers patched for where the
         * planes are (or will be) after the IDWT */
        PackTile *T = (PackTile *)(j->h_desc.data() + j->pack_off);
        int ti = 0;
        for (int f = 0; f < j->nframes; f++) {
            const FrameSlot &F = j->frames[f];
            for (int k = 0; k < F.plan->ntiles; k++, ti++) {
                T[ti].fused = fuse && j->tile_fusable[ti];
                for (int cc = 0; cc < T[ti].ncomp; cc++) {
                    const int t = F.tc_base + k * T[ti].ncomp + cc;
                    T[ti].c[cc].src = buf_ptr(j, j->final_eff[t]) + j->tilecomps[t].plane_off;
                }
            }
        }
        if (!(mask & 1)) HIP_TRY(c, hipEventRecord(j->ev[3], j->stream));
        /* the tables only change with the knobs (which buffer a plane ends in, fused or not): a run whose tables are what the
         * device already holds uploads nothing -- a pageable copy in the middle of the stream, once per step, otherwise */
        if (j->h_desc != j->h_desc_dev) {
            HIP_TRY(c, hipMemcpyAsync(j->d_desc.p, j->h_desc.data(), j->h_desc.size(), hipMemcpyHostToDevice, j->stream));
            j->h_desc_dev = j->h_desc;
        }
        if (mask & 2) {
            /* coef16_ok has checked that every level of every plane is a FASTONLY streaming launch and every plane ends fused */
            j->ll16_run = c->ll16 && j->coef_is16 && fuse && use_tile && !j->force_ll32;
            j->ll16_checked = !j->ll16_run;
            int r = run_idwt(c, j, use_tile, fuse);
            if (r < 0) return r;
            HIP_TRY(c, hipGetLastError());
        }
        HIP_TRY(c, hipEventRecord(j->ev[4], j->stream));
    }
    if (mask & 4) {
        bool all_fused = j->fused_last;
        for (size_t i = 0; all_fused && i < j->tile_fusable.size(); i++) all_fused = j->tile_fusable[i] != 0;
        if (j->npack && j->pack_maxw > 0 && j->pack_maxh > 0 && !all_fused) {
            dim3 g((j->pack_maxw + 1023) / 1024, j->pack_maxh, j->npack);
            hipLaunchKernelGGL(k_mct_pack, g, dim3(256), 0, j->stream,
                               (const PackTile *)((uint8_t *)j->d_desc.p + j->pack_off));
            HIP_TRY(c, hipGetLastError());
        }
        HIP_TRY(c, hipEventRecord(j->ev[5], j->stream));
    }
    j->ran |= mask;
    return 0;
}

extern "C" int htj2k_job_run(htj2k_ctx *c, htj2k_job *j) { return htj2k_job_run_stages(c, j, 7); }

/* Wait for the job's stream; if its last IDWT run kept the LL bands as 16-bit samples, look at the overflow flag once and,
 * if a sample did not fit, run the transform again with 32-bit LL bands (the sub-bands are still there) */
static int job_settle(htj2k_ctx *c, htj2k_job *j)
{
    HIP_TRY(c, hipStreamSynchronize(j->stream));
    if (j->ll16_checked) return 0;
    j->ll16_checked = true;
    int flag = 0;
    HIP_TRY(c, hipMemcpy(&flag, (const int *)j->d_status.p + j->blocks.size(), sizeof(int), hipMemcpyDeviceToHost));
    if (!flag) return 0;
    clog(c, LOG_INFO, "an LL band left the 16-bit range: inverse DWT run again with 32-bit LL bands\n");
    j->ll16_fallbacks++;
    j->force_ll32 = true;
    const int r = htj2k_job_run_stages(c, j, 6);
    j->force_ll32 = false;
    if (r < 0) return r;
    HIP_TRY(c, hipStreamSynchronize(j->stream));
    return 0;
}

extern "C" int htj2k_job_wait(htj2k_ctx *c, htj2k_job *j)
{
    if (!c || !j) return HTJ2K_ERR_EINVAL;
    return job_settle(c, j);
}

/* 0: the last run's LL bands were 32-bit, 1: 16-bit, 2: 16-bit, overflowed and run again with 32 (after htj2k_job_wait) */
extern "C" int htj2k_job_idwt_packed(const htj2k_job *j) { return j ? j->pk_used : HTJ2K_ERR_EINVAL; }
extern "C" int htj2k_job_ll16(const htj2k_job *j) { return j ? (j->ll16_run ? 1 : (j->ll16_fallbacks ? 2 : 0)) : HTJ2K_ERR_EINVAL; }

extern "C" int htj2k_job_stage_ms(htj2k_ctx *c, htj2k_job *j, float *ms_ht, float *ms_idwt, float *ms_pack)
{
    if (!c || !j) return HTJ2K_ERR_EINVAL;
    HIP_TRY(c, hipStreamSynchronize(j->stream));
    float a = 0, b = 0, d = 0;
    if (j->ran & 1) (void)hipEventElapsedTime(&a, j->ev[2], j->ev[3]);
    if (j->ran & 2) (void)hipEventElapsedTime(&b, j->ev[3], j->ev[4]);
    if (j->ran & 4) (void)hipEventElapsedTime(&d, j->ev[4], j->ev[5]);
    if (ms_ht) *ms_ht = a;
    if (ms_idwt) *ms_idwt = b;
    if (ms_pack) *ms_pack = d;
    return 0;
}

/* per-launch device time and algorithmic bytes (2 * 4 * lh * lv summed over the planes of
 * the launch) of the IDWT kernels of the job's last run: the roofline inputs of bench.py */
extern "C" int htj2k_job_idwt_launches(htj2k_ctx *c, htj2k_job *j, float *ms, double *bytes, int cap)
{
    if (!c || !j) return HTJ2K_ERR_EINVAL;
    HIP_TRY(c, hipStreamSynchronize(j->stream));
    const int n = (int)j->lev_bytes.size();
    for (int i = 0; i < n && i < cap; i++) {
        float t = 0;
        (void)hipEventElapsedTime(&t, j->lev_ev[2 * i], j->lev_ev[2 * i + 1]);
        if (ms) ms[i] = t;
        if (bytes) bytes[i] = j->lev_bytes[i];
    }
    return n;
}

/* 1 when the last htj2k_job_run kept the sub-bands as 16-bit samples between the block decoder and the IDWT */
extern "C" int htj2k_job_coef16(const htj2k_job *j) { return j ? (j->coef_is16 ? 1 : 0) : HTJ2K_ERR_EINVAL; }
extern "C" int htj2k_job_ht_blocks_per_wave(const htj2k_job *j) { return j ? j->ht_bpw : HTJ2K_ERR_EINVAL; }

extern "C" int htj2k_job_idwt_hbm_bytes(htj2k_ctx *c, htj2k_job *j, double *bytes, int cap)
{
    if (!c || !j) return HTJ2K_ERR_EINVAL;
    const int n = (int)j->lev_hbm.size();
    for (int i = 0; i < n && i < cap; i++)
        if (bytes) bytes[i] = j->lev_hbm[i];
    return n;
}

extern "C" int htj2k_job_block_errors(htj2k_ctx *c, htj2k_job *j)
{
    if (!c || !j || j->nframes <= 0) return HTJ2K_ERR_EINVAL;
    const int n = (int)j->blocks.size();
    if (!n) return 0;
    std::vector<int> st(n);
    HIP_TRY(c, hipStreamSynchronize(j->stream));
    HIP_TRY(c, hipMemcpy(st.data(), j->d_status.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    int e = 0;
    for (int i = 0; i < n; i++) e += st[i] != 0;
    return e;
}

extern "C" int htj2k_job_read_plane(htj2k_ctx *c, htj2k_job *j, int tc, void *dst, size_t dst_bytes)
{
    if (!c || !j || j->nframes <= 0 || tc < 0 || tc >= (int)j->tilecomps.size()) return HTJ2K_ERR_EINVAL;
    const J2kTileComp &t = j->tilecomps[tc];
    const size_t n = (size_t)t.w * t.h * 4;
    if (dst_bytes < n) return HTJ2K_ERR_EINVAL;
    const int fb = j->final_eff.empty() ? 0 : j->final_eff[tc];
    if (j->fused_last && !j->plane_fused.empty() && j->plane_fused[tc]) return HTJ2K_ERR_EINVAL;   /* never materialised */
    if (!j->ll16_checked) { const int r = job_settle(c, j); if (r < 0) return r; }
    HIP_TRY(c, hipStreamSynchronize(j->stream));
    HIP_TRY(c, hipMemcpy(dst, buf_ptr(j, fb) + t.plane_off, n, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int htj2k_job_download_frame(htj2k_ctx *c, htj2k_job *j, int f, htj2k_frame *frame)
{
    if (!c || !j || f < 0 || f >= j->nframes || !frame) return HTJ2K_ERR_EINVAL;
    const FrameSlot &F = j->frames[f];
    const J2kPlan *pl = F.plan;
    if (!j->ll16_checked) { const int r = job_settle(c, j); if (r < 0) return r; }
    for (int p = 0; p < pl->info.nplanes; p++) {
        const int rowbytes = pl->info.plane_width[p] * pl->info.plane_bytes_per_sample[p];
        if (!frame->data[p] || frame->linesize[p] < rowbytes) return HTJ2K_ERR_EINVAL;
        HIP_TRY(c, hipMemcpy2DAsync(frame->data[p], frame->linesize[p], F.d_out[p].p, F.out.linesize[p],
                                    rowbytes, pl->info.plane_height[p], hipMemcpyDeviceToHost, j->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(j->stream));
    frame->width = pl->info.width;
    frame->height = pl->info.height;
    frame->pix_fmt = pl->info.pix_fmt;
    return 0;
}

extern "C" int htj2k_job_download(htj2k_ctx *c, htj2k_job *j, htj2k_frame *frame)
{
    return htj2k_job_download_frame(c, j, 0, frame);
}

/* device addresses / pitches of the output planes of frame f (for callers that keep frames on the GPU); valid
 * until the job is parsed again */
extern "C" int htj2k_job_device_frame(htj2k_ctx *c, htj2k_job *j, int f, htj2k_frame *frame)
{
    if (!c || !j || f < 0 || f >= j->nframes || !frame) return HTJ2K_ERR_EINVAL;
    /* the planes are final only once the 16-bit LL check of the last run has been looked at (and the transform run
     * again if it tripped): that also waits for the job's stream */
    if (!j->ll16_checked) { const int r = job_settle(c, j); if (r < 0) return r; }
    const FrameSlot &F = j->frames[f];
    const J2kPlan *pl = F.plan;
    memset(frame, 0, sizeof(*frame));
    for (int p = 0; p < pl->info.nplanes; p++) {
        frame->data[p] = F.out.ptr[p];
        frame->linesize[p] = F.out.linesize[p];
    }
    frame->width = pl->info.width;
    frame->height = pl->info.height;
    frame->pix_fmt = pl->info.pix_fmt;
    return 0;
}

/* device address of an output plane of frame 0 (for callers that keep frames on the GPU); htj2k_job_wait must have
 * returned since the last run: only then are the planes final (this call has no context to wait with) */
extern "C" void *htj2k_job_device_plane(htj2k_job *j, int plane, int *linesize)
{
    if (!j || j->nframes <= 0 || plane < 0 || plane > 3) return nullptr;
    if (linesize) *linesize = j->frames[0].out.linesize[plane];
    return j->frames[0].out.ptr[plane];
}

/* ------------------------------------------------------------------ one-call decode */
extern "C" int htj2k_decode(htj2k_ctx *c, const uint8_t *pkt, int size, htj2k_frame *frame, htj2k_stats *stats)
{
    if (!c || !pkt || !frame) return HTJ2K_ERR_EINVAL;
    auto t0 = std::chrono::steady_clock::now();
    int r = htj2k_job_parse(c, pkt, size, &c->own_job);
    if (r < 0) return r;
    auto t1 = std::chrono::steady_clock::now();
    htj2k_job *j = c->own_job;
    if ((r = htj2k_job_upload(c, j)) < 0) return r;
    if ((r = htj2k_job_run(c, j)) < 0) return r;
    auto t2 = std::chrono::steady_clock::now();
    if ((r = htj2k_job_download(c, j, frame)) < 0) return r;
    auto t3 = std::chrono::steady_clock::now();
    int nerr = htj2k_job_block_errors(c, j);
    if (nerr > 0)
        clog(c, LOG_ERROR, "Bad HT cleanup segment in %d codeblock(s)\n", nerr);   /* jpeg2000htdec.c:1305 */
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->n_codeblocks = (int)j->blocks.size();
        stats->n_block_errors = nerr > 0 ? nerr : 0;
        stats->ms_parse = std::chrono::duration<float, std::milli>(t1 - t0).count();
        (void)hipEventElapsedTime(&stats->ms_h2d, j->ev[0], j->ev[1]);
        (void)hipEventElapsedTime(&stats->ms_kernels, j->ev[2], j->ev[5]);
        stats->ms_d2h = std::chrono::duration<float, std::milli>(t3 - t2).count();
        htj2k_job_stage_ms(c, j, &stats->ms_ht, &stats->ms_idwt, &stats->ms_pack);
    }
    return j->frames[0].plan->bytes_consumed;
}

/* ------------------------------------------------------------------ kernel-level entry points */
static void fill_levels(const int border[2][2], int levels, int32_t linelen[][2], uint8_t mod[][2])
{
    int b[2][2], lev = levels;
    for (int i = 0; i < 2; i++) for (int k = 0; k < 2; k++) b[i][k] = border[i][k];
    while (--lev >= 0)
        for (int i = 0; i < 2; i++) {
            linelen[lev][i] = b[i][1] - b[i][0];
            mod[lev][i] = b[i][0] & 1;
            for (int k = 0; k < 2; k++) b[i][k] = (b[i][k] + 1) >> 1;
        }
}

struct PlaneSet {                 /* nplanes identical-geometry planes laid out back to back */
    int w, h, levels, type, nplanes;
    size_t plane_samples;
    std::vector<std::vector<uint8_t>> tables_generic, tables_tile;   /* per level */
    std::vector<int> lh, lv;
    std::vector<uint8_t> fast;        /* per level: stream_fast_geom */
};

static void planeset_build(PlaneSet &P, const int border[2][2], int levels, int type, int nplanes)
{
    int32_t linelen[J2K_MAX_DWTLEV][2]; uint8_t mod[J2K_MAX_DWTLEV][2];
    P.w = border[0][1] - border[0][0]; P.h = border[1][1] - border[1][0];
    P.levels = levels; P.type = type; P.nplanes = nplanes;
    P.plane_samples = align_up((size_t)P.w * P.h, 64);
    fill_levels(border, levels, linelen, mod);
    for (int lev = 0; lev < levels; lev++) {
        std::vector<uint8_t> tg, tt;
        for (int p = 0; p < nplanes; p++) {
            DwtLevel d;
            d.plane_off = (uint32_t)(p * P.plane_samples); d.stride = P.w;
            d.lh = linelen[lev][0]; d.lv = linelen[lev][1]; d.mh = mod[lev][0]; d.mv = mod[lev][1];
            d.last = (type == J2K_DWT97_INT && lev == levels - 1);
            DwtTileArgs a; a.g = d; a.ll_off = d.plane_off; a.ll_stride = P.w; a.out_off = d.plane_off; a.out_stride = P.w;
            push_bytes(tg, &d, sizeof(d));
            push_bytes(tt, &a, sizeof(a));
        }
        P.tables_generic.push_back(tg); P.tables_tile.push_back(tt);
        P.lh.push_back(linelen[lev][0]); P.lv.push_back(linelen[lev][1]);
        {
            DwtLevel d;
            d.plane_off = 0; d.stride = P.w; d.lh = linelen[lev][0]; d.lv = linelen[lev][1]; d.mh = mod[lev][0]; d.mv = mod[lev][1]; d.last = 0;
            P.fast.push_back(stream_fast_geom(d));
        }
    }
}

template <int TYPE>
static void planeset_launch_level(const PlaneSet &P, int lev, int mode, int kth, const void *d_tab,
                                  uint32_t *coef, uint32_t *t0, uint32_t *t1, hipStream_t s)
{
    if (mode == 0) {
        dim3 g((P.lh[lev] + 255) / 256, P.lv[lev], P.nplanes);
        hipLaunchKernelGGL((k_idwt_h<TYPE>), g, dim3(256), 0, s, (const DwtLevel *)d_tab, (const uint32_t *)coef, t0);
        hipLaunchKernelGGL((k_idwt_v<TYPE>), g, dim3(256), 0, s, (const DwtLevel *)d_tab, (const uint32_t *)t0, coef);
    } else {
        const uint32_t *ll = kth == 0 ? coef : ((kth - 1) & 1) ? t1 : t0;
        uint32_t *out = (kth & 1) ? t1 : t0;
        launch_tile_generic<TYPE>(d_tab, P.lh[lev], P.lv[lev], P.lh[lev] < P.lv[lev] ? P.lh[lev] : P.lv[lev], P.nplanes, mode, s,
                                  ll, (const uint32_t *)coef, out, P.fast[lev] != 0);
    }
}

/* returns which buffer (0 coef, 1 t0, 2 t1) holds the result */
static int planeset_run(const PlaneSet &P, int mode, uint8_t *d_tabs, const std::vector<size_t> &tab_off,
                        uint32_t *coef, uint32_t *t0, uint32_t *t1, hipStream_t s)
{
    int kth = 0;
    for (int lev = 0; lev < P.levels; lev++) {
        if (P.lh[lev] <= 0 || P.lv[lev] <= 0) continue;
        const void *tab = d_tabs + tab_off[lev];
        if (P.type == J2K_DWT53) planeset_launch_level<J2K_DWT53>(P, lev, mode, kth, tab, coef, t0, t1, s);
        else if (P.type == J2K_DWT97) planeset_launch_level<J2K_DWT97>(P, lev, mode, kth, tab, coef, t0, t1, s);
        else planeset_launch_level<J2K_DWT97_INT>(P, lev, mode, kth, tab, coef, t0, t1, s);
        kth++;
    }
    if (mode == 0 || kth == 0) return 0;
    return 1 + ((kth - 1) & 1);
}

static int planeset_upload_tables(htj2k_ctx *c, const PlaneSet &P, int mode, DevBuf &d_tabs, std::vector<size_t> &off)
{
    std::vector<uint8_t> all;
    off.clear();
    for (int lev = 0; lev < P.levels; lev++) {
        while (all.size() % 16) all.push_back(0);
        off.push_back(all.size());
        const std::vector<uint8_t> &t = mode == 0 ? P.tables_generic[lev] : P.tables_tile[lev];
        all.insert(all.end(), t.begin(), t.end());
    }
    int r = d_tabs.ensure(all.size() + 64);
    if (r < 0) return r;
    if (!all.empty()) HIP_TRY(c, hipMemcpy(d_tabs.p, all.data(), all.size(), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int htj2k_idwt_plane(htj2k_ctx *c, void *plane, const int border[2][2], int levels, int type)
{
    if (!c || !plane || levels < 0 || levels > J2K_MAX_DWTLEV || type < 0 || type > 2) return HTJ2K_ERR_EINVAL;
    if (border[0][1] <= border[0][0] || border[1][1] <= border[1][0]) return HTJ2K_ERR_EINVAL;
    if (levels == 0) return 0;
    HIP_TRY(c, hipSetDevice(c->device));
    PlaneSet P;
    planeset_build(P, border, levels, type, 1);
    DevBuf coef, t0, t1, tabs;
    std::vector<size_t> off;
    const size_t bytes = P.plane_samples * 4 + 256;
    int r;
    if ((r = coef.ensure(bytes)) < 0 || (r = t0.ensure(bytes)) < 0 || (r = t1.ensure(bytes)) < 0 ||
        (r = planeset_upload_tables(c, P, c->idwt_mode, tabs, off)) < 0) {
        coef.release(); t0.release(); t1.release(); tabs.release();
        return r;
    }
    hipError_t e = hipMemcpy(coef.p, plane, (size_t)P.w * P.h * 4, hipMemcpyHostToDevice);
    int fb = 0;
    if (e == hipSuccess) {
        fb = planeset_run(P, c->idwt_mode, (uint8_t *)tabs.p, off, (uint32_t *)coef.p, (uint32_t *)t0.p, (uint32_t *)t1.p, 0);
        e = hipDeviceSynchronize();
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess)
        e = hipMemcpy(plane, fb == 0 ? coef.p : fb == 1 ? t0.p : t1.p, (size_t)P.w * P.h * 4, hipMemcpyDeviceToHost);
    coef.release(); t0.release(); t1.release(); tabs.release();
    if (e != hipSuccess) { clog(c, LOG_ERROR, "HIP error %s in htj2k_idwt_plane\n", hipGetErrorString(e)); return HTJ2K_ERR_EXTERNAL; }
    return 0;
}

__global__ void k_fill_pattern(uint32_t *p, size_t n, int is_float)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t h = (uint32_t)i * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    int v = (int)(h % 511u) - 255;
    p[i] = is_float ? __float_as_uint((float)v * 0.25f) : (uint32_t)v;
}

extern "C" int htj2k_idwt_bench(htj2k_ctx *c, int w, int h, int levels, int type, int nplanes, int iters, float *ms_per_iter)
{
    if (!c || w <= 0 || h <= 0 || levels <= 0 || levels > J2K_MAX_DWTLEV || type < 0 || type > 2 || nplanes <= 0 || iters <= 0 || !ms_per_iter)
        return HTJ2K_ERR_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    const int border[2][2] = { { 0, w }, { 0, h } };
    PlaneSet P;
    planeset_build(P, border, levels, type, nplanes);
    DevBuf coef, t0, t1, tabs;
    std::vector<size_t> off;
    const size_t n = P.plane_samples * nplanes, bytes = n * 4 + 256;
    int r;
    if ((r = coef.ensure(bytes)) < 0 || (r = t0.ensure(bytes)) < 0 || (r = t1.ensure(bytes)) < 0 ||
        (r = planeset_upload_tables(c, P, c->idwt_mode, tabs, off)) < 0) {
        coef.release(); t0.release(); t1.release(); tabs.release();
        return r;
    }
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_fill_pattern, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (uint32_t *)coef.p, n, type == J2K_DWT97);
        planeset_run(P, c->idwt_mode, (uint8_t *)tabs.p, off, (uint32_t *)coef.p, (uint32_t *)t0.p, (uint32_t *)t1.p, s);   /* warm-up */
        e = hipStreamSynchronize(s);
    }
    if (e == hipSuccess) {
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < iters; i++)
            planeset_run(P, c->idwt_mode, (uint8_t *)tabs.p, off, (uint32_t *)coef.p, (uint32_t *)t0.p, (uint32_t *)t1.p, s);
        (void)hipEventRecord(e1, s);
        e = hipStreamSynchronize(s);
    }
    float ms = 0;
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (s) (void)hipStreamDestroy(s);
    coef.release(); t0.release(); t1.release(); tabs.release();
    if (e != hipSuccess) { clog(c, LOG_ERROR, "HIP error %s in htj2k_idwt_bench\n", hipGetErrorString(e)); return HTJ2K_ERR_EXTERNAL; }
    *ms_per_iter = ms / iters;
    return 0;
}

/* Calibration for the roofline figures: what a kernel that does nothing but move bytes reaches on THIS device, so
 * that a box whose memory system copies at 5.1 TB/s is not read as a kernel at 0.55 of 8 TB/s.  A grid-stride copy of
 * 16-byte elements, `mbytes` MB read and as many written per launch; the launch shapes of tools/ubench/membw.hip,
 * the best of them is returned (GB/s, read + written). */
__global__ void __launch_bounds__(256) k_copy_linear(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

extern "C" int htj2k_copy_bench(htj2k_ctx *c, int mbytes, int iters, float *gbps)
{
    if (!c || mbytes <= 0 || mbytes > 8192 || iters <= 0 || !gbps) return HTJ2K_ERR_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = (size_t)mbytes * 1000000 / 16;
    DevBuf src, dst;
    int r;
    if ((r = src.ensure(n * 16)) < 0 || (r = dst.ensure(n * 16)) < 0) { src.release(); dst.release(); return r; }
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipMemsetAsync(src.p, 1, n * 16, s);
    float best = 0;
    const int grids[3] = { 8192, 65536, 262144 };
    for (int g = 0; g < 3 && e == hipSuccess; g++) {
        for (int i = 0; i < 2; i++)
            hipLaunchKernelGGL(k_copy_linear, dim3(grids[g]), dim3(256), 0, s, (const uint4 *)src.p, (uint4 *)dst.p, n);
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < iters; i++)
            hipLaunchKernelGGL(k_copy_linear, dim3(grids[g]), dim3(256), 0, s, (const uint4 *)src.p, (uint4 *)dst.p, n);
        (void)hipEventRecord(e1, s);
        e = hipStreamSynchronize(s);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess && ms > 0) best = std::max(best, (float)(2.0 * n * 16 * iters / (ms * 1e-3) / 1e9));
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (s) (void)hipStreamDestroy(s);
    src.release(); dst.release();
    if (e != hipSuccess) { clog(c, LOG_ERROR, "HIP error %s in htj2k_copy_bench\n", hipGetErrorString(e)); return HTJ2K_ERR_EXTERNAL; }
    *gbps = best;
    return 0;
}

extern "C" int htj2k_mct_planes(htj2k_ctx *c, void *p0, void *p1, void *p2, int csize, int type)
{
    if (!c || !p0 || !p1 || !p2 || csize <= 0 || type < 0 || type > 2) return HTJ2K_ERR_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf d[3];
    void *h[3] = { p0, p1, p2 };
    int r = 0;
    hipError_t e = hipSuccess;
    for (int i = 0; i < 3 && r >= 0; i++) r = d[i].ensure((size_t)csize * 4);
    for (int i = 0; i < 3 && r >= 0 && e == hipSuccess; i++) e = hipMemcpy(d[i].p, h[i], (size_t)csize * 4, hipMemcpyHostToDevice);
    if (r >= 0 && e == hipSuccess) {
        hipLaunchKernelGGL(k_mct_only, dim3((csize + 255) / 256), dim3(256), 0, 0, (uint32_t *)d[0].p, (uint32_t *)d[1].p, (uint32_t *)d[2].p, csize, type);
        e = hipDeviceSynchronize();
    }
    for (int i = 0; i < 3 && r >= 0 && e == hipSuccess; i++) e = hipMemcpy(h[i], d[i].p, (size_t)csize * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < 3; i++) d[i].release();
    if (r < 0) return r;
    return e == hipSuccess ? 0 : HTJ2K_ERR_EXTERNAL;
}

/* HT block decoder alone: decode `n` codeblocks given as a descriptor table + byte pool into
 * a sample buffer (unit parity against ff_jpeg2000_decode_htj2k + dequantisation) */
/* decode_cblk() + dequantisation on a caller-built table of Part-1 blocks: every descriptor carries J2K_BLK_PART1 and
 * its bytes are laid out as j2k_plan.c does it (j2k_plan.h: segments, 0xFF 0xFF terminators, J2kPart1Trailer) */
/* A caller-built descriptor (the unit entry points below) gets the guarantees the job path has from the host parser:
 * the block is at most 4096 samples and 1024 in either direction (jpeg2000htdec.c:1230-1232, what the kernels' LDS and
 * quad-symbol sizing assume), its window lies inside the coefficient buffer, the numbers the kernels shift by are in
 * range. */
static bool block_desc_ok(const J2kBlock &b, size_t nsamples)
{
    if (!b.w || !b.h || b.w > 1024 || b.h > 1024 || (uint32_t)b.w * b.h > 4096 || b.stride < b.w) return false;
    if ((size_t)b.plane_off + (size_t)(b.h - 1) * b.stride + b.w > nsamples) return false;
    if (b.roi_shift > 30 || b.npasses >= 100) return false;
    return true;
}

extern "C" int htj2k_mq_blocks(htj2k_ctx *c, const void *blocks_in, int nblocks, const uint8_t *bytes_in, size_t nbytes_in,
                               void *coef, size_t nsamples, int *status)
{
    if (!c || !blocks_in || nblocks <= 0 || !bytes_in || !coef) return HTJ2K_ERR_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    std::vector<J2kBlock> blk((const J2kBlock *)blocks_in, (const J2kBlock *)blocks_in + nblocks);
    std::vector<uint8_t> pool(16, 0);
    for (int i = 0; i < nblocks; i++) {
        J2kBlock &b = blk[i];
        if (!(b.flags & J2K_BLK_PART1) || !block_desc_ok(b, nsamples) || b.M_b > 37) return HTJ2K_ERR_EINVAL;
        const size_t len = J2K_P1_TRAILER_OFF(b.lcup) + 4 + 2 * (size_t)b.lref;
        if ((size_t)b.data_off + len > nbytes_in) return HTJ2K_ERR_EINVAL;
        {   /* the trailer is trusted by the kernel: segment count and starts must lie inside the block's bytes */
            const J2kPart1Trailer *tr = (const J2kPart1Trailer *)(bytes_in + b.data_off + J2K_P1_TRAILER_OFF(b.lcup));
            if (tr->nterm != b.lref || tr->bandpos > 3) return HTJ2K_ERR_EINVAL;
            for (uint32_t k = 0; k < tr->nterm; k++)
                if (tr->start[k] > b.lcup) return HTJ2K_ERR_EINVAL;
        }
        const size_t o = pool.size();
        pool.resize(o + J2K_P1_REGION(b.lcup, b.lref), 0);
        memcpy(pool.data() + o, bytes_in + b.data_off, len);
        b.data_off = (uint32_t)o;
    }
    pool.resize(pool.size() + 256, 0);
    std::vector<MqWave> waves;
    size_t units = 0;
    uint32_t planes = 1;
    bool wide = false;
    for (int i = 0; i < nblocks; i += 64) {
        MqWave W;
        int hmax = 0, wmax = 0, pmax = 0;
        for (int k = i; k < std::min(i + 64, nblocks); k++) {
            hmax = std::max<int>(hmax, blk[k].h); wmax = std::max<int>(wmax, blk[k].w); pmax = std::max<int>(pmax, blk[k].npasses);
        }
        const int np = std::min((pmax + 1) / 3 + 1, 32);
        W.hmax = (uint16_t)hmax; W.wmax = (uint16_t)wmax; W.pmax = (uint16_t)pmax;
        W.rows = (uint16_t)(((hmax + 3) & ~3) + 2);
        W.chunks = (uint16_t)((wmax + 63) / 64); W.pad = 0;
        W.soff = (uint32_t)units;
        units += (size_t)(4 + np) * W.rows * W.chunks;
        planes = std::max<uint32_t>(planes, (uint32_t)np);
        wide = wide || W.chunks > 1;
        waves.push_back(W);
    }
    DevBuf db, dby, dc, ds, dw, dsc;
    auto release_all = [&]() { db.release(); dby.release(); dc.release(); ds.release(); dw.release(); dsc.release(); };
    int r;
    if ((r = db.ensure((size_t)nblocks * sizeof(J2kBlock))) < 0 || (r = dby.ensure(pool.size())) < 0 ||
        (r = dc.ensure(nsamples * 4 + 64)) < 0 || (r = ds.ensure((size_t)nblocks * 4)) < 0 ||
        (r = dw.ensure(waves.size() * sizeof(MqWave))) < 0 || (r = dsc.ensure(units * 512 + 512)) < 0) {
        release_all();
        return r;
    }
    hipError_t e = hipMemcpy(db.p, blk.data(), (size_t)nblocks * sizeof(J2kBlock), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dby.p, pool.data(), pool.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dc.p, coef, nsamples * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(ds.p, 0, (size_t)nblocks * 4);
    if (e == hipSuccess) e = hipMemcpy(dw.p, waves.data(), waves.size() * sizeof(MqWave), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const uint32_t area = mq_lds_area(planes);
        auto kern = wide ? k_mq_decode<true> : k_mq_decode<false>;
        hipLaunchKernelGGL(kern, dim3((unsigned)waves.size()), dim3(64), area + MQ_LDS_TABLES, 0, (const J2kBlock *)db.p, nblocks,
                           (const uint8_t *)dby.p, (uint32_t *)dc.p, (int *)ds.p, (const MqWave *)dw.p, (uint64_t *)dsc.p, area);
        e = hipDeviceSynchronize();
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(coef, dc.p, nsamples * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && status) e = hipMemcpy(status, ds.p, (size_t)nblocks * 4, hipMemcpyDeviceToHost);
    release_all();
    if (e != hipSuccess) { clog(c, LOG_ERROR, "HIP error %s in htj2k_mq_blocks\n", hipGetErrorString(e)); return HTJ2K_ERR_EXTERNAL; }
    return 0;
}

extern "C" int htj2k_ht_blocks(htj2k_ctx *c, const void *blocks_in, int nblocks, const uint8_t *bytes_in, size_t nbytes_in,
                               void *coef, size_t nsamples, int *status)
{
    if (!c || !blocks_in || nblocks <= 0 || !bytes_in || !coef) return HTJ2K_ERR_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    /* the kernels want the layout j2k_plan.c produces: every block's bytes at a 16-byte aligned
     * offset, J2K_BLOCK_PAD bytes behind them and 16 in front of the first: re-pack the caller's pool */
    std::vector<J2kBlock> blk((const J2kBlock *)blocks_in, (const J2kBlock *)blocks_in + nblocks);
    std::vector<uint8_t> pool(16, 0);
    for (int i = 0; i < nblocks; i++) {
        J2kBlock &b = blk[i];
        const size_t len = (size_t)b.lcup + b.lref;
        if ((b.flags & J2K_BLK_PART1) || !block_desc_ok(b, nsamples) || b.M_b > 30) return HTJ2K_ERR_EINVAL;
        if ((size_t)b.data_off + len > nbytes_in) return HTJ2K_ERR_EINVAL;
        const size_t o = pool.size();
        pool.resize(o + J2K_BLOCK_REGION(len), 0);
        memcpy(pool.data() + o, bytes_in + b.data_off, len);
        b.data_off = (uint32_t)o;
    }
    const void *blocks = blk.data();
    const uint8_t *bytes = pool.data();
    const size_t nbytes = pool.size();
    struct { HtLds lds, ext; } tmp;
    const uint8_t *bases[1] = { bytes };
    int r = build_ht_lds(c, (const J2kBlock *)blocks, nblocks, bases, nullptr, &tmp.lds, &tmp.ext);
    if (r < 0) return r;
    std::vector<uint32_t> qoff(nblocks + 1);
    size_t nq = 0;
    for (int i = 0; i < nblocks; i++) {
        const J2kBlock &b = ((const J2kBlock *)blocks)[i];
        qoff[i] = (uint32_t)nq;
        if (b.npasses) nq += ht_qsym_words(b.w, b.h);
    }
    /* blocks with refinement passes that k_ht_refine handles (same rule as htj2k_job_upload) */
    std::vector<uint32_t> reflist, roff;
    uint32_t max_lref = 0, ref_max_w = 0;
    for (int i = 0; i < nblocks; i++) max_lref = std::max<uint32_t>(max_lref, ((const J2kBlock *)blocks)[i].lref);
    const size_t nmasks = ref_layout((const J2kBlock *)blocks, (size_t)nblocks, reflist, roff, &ref_max_w);
    DevBuf db, dby, dc, ds, dq, dqo, du[2], drl, dro, drb;
    auto release_all = [&]() {
        db.release(); dby.release(); dc.release(); ds.release(); dq.release(); dqo.release(); du[0].release(); du[1].release();
        drl.release(); dro.release(); drb.release();
    };
    if ((r = dq.ensure((nq + 64) * 4)) < 0 || (r = dqo.ensure((size_t)(nblocks + 1) * 4)) < 0 ||
        (r = du[0].ensure(nbytes + 16384)) < 0 || (r = du[1].ensure(nbytes + 16384)) < 0 ||
        (r = db.ensure((size_t)nblocks * sizeof(J2kBlock))) < 0 || (r = dby.ensure(nbytes + 64)) < 0 ||
        (r = dc.ensure(nsamples * 4 + 64)) < 0 || (r = ds.ensure((size_t)nblocks * 4)) < 0 ||
        (r = drl.ensure((reflist.size() + 1) * 4)) < 0 || (r = dro.ensure(roff.size() * 4)) < 0 ||
        (r = drb.ensure((nmasks + 8) * 8)) < 0) {
        release_all();
        return r;
    }
    hipError_t e = hipMemcpy(db.p, blocks, (size_t)nblocks * sizeof(J2kBlock), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dby.p, bytes, nbytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dc.p, coef, nsamples * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(ds.p, 0, (size_t)nblocks * 4);
    if (e == hipSuccess) e = hipMemcpy(dqo.p, qoff.data(), (size_t)nblocks * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dro.p, roff.data(), roff.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && !reflist.empty()) e = hipMemcpy(drl.p, reflist.data(), reflist.size() * 4, hipMemcpyHostToDevice);
    const size_t vlc_lds = ht_vlc_lds_bytes(tmp.lds.max_qw);
    if (e == hipSuccess && c->ht_mode == 1 && vlc_lds <= 160 * 1024) {
        if (vlc_lds > 48 * 1024)
            e = hipFuncSetAttribute((const void *)k_ht_vlc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)vlc_lds);
        if (e == hipSuccess && (int)tmp.ext.total > 48 * 1024)
            e = hipFuncSetAttribute((const void *)k_ht_decode<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tmp.ext.total);
        if (e == hipSuccess) {
            const uint32_t us_words = 2 * std::max(tmp.lds.vlc_words, reflist.empty() ? 0u : ht_nsp(max_lref));
            /* (one block per wave unless a test asks otherwise: the unit entry keeps the plain kernels exercised) */
            e = launch_unstuff(0, (const J2kBlock *)db.p, nblocks, (const uint8_t *)dby.p, (uint32_t *)du[0].p, (uint32_t *)du[1].p, us_words,
                               getenv("HTJ2K_UNSTUFF_G") ? atoi(getenv("HTJ2K_UNSTUFF_G")) : 1);
            hipLaunchKernelGGL(k_ht_vlc, dim3((nblocks + 63) / 64), dim3(64), vlc_lds, 0, (const J2kBlock *)db.p, nblocks,
                               (const uint8_t *)dby.p, (const uint16_t *)c->d_tables, (ht_sym_t *)dq.p, (const uint32_t *)dqo.p,
                               tmp.lds.max_qw, (const uint32_t *)du[0].p, (const uint32_t *)du[1].p, (uint32_t *)dq.p + nq / 2 + 32);
            if (!reflist.empty())
                hipLaunchKernelGGL((ref_max_w <= 32 ? k_ht_refine<uint32_t, true> : k_ht_refine<uint64_t, true>), dim3(((unsigned)reflist.size() + 63) / 64), dim3(64), 0, 0,
                                   (const J2kBlock *)db.p, (const uint32_t *)drl.p, (int)reflist.size(), (const uint8_t *)dby.p,
                                   (const ht_sym_t *)dq.p, (const uint32_t *)dqo.p, (const uint32_t *)du[0].p, (const uint32_t *)du[1].p,
                                   (uint64_t *)drb.p, (const uint32_t *)dro.p);
            hipLaunchKernelGGL(k_ht_decode<true>, dim3(nblocks), dim3(64), tmp.ext.total, 0, (const J2kBlock *)db.p, nblocks,
                               (const uint8_t *)dby.p, (uint32_t *)dc.p, (const uint16_t *)c->d_tables, (int *)ds.p, tmp.ext,
                               (const ht_sym_t *)dq.p, (const uint32_t *)dqo.p, (uint32_t *)dc.p + nsamples + 8,
                               (const uint64_t *)drb.p, (const uint32_t *)dro.p);
            e = hipDeviceSynchronize();
        }
    } else if (e == hipSuccess) {
        if ((int)tmp.lds.total > 48 * 1024)
            e = hipFuncSetAttribute((const void *)k_ht_decode<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tmp.lds.total);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_ht_decode<false>, dim3(nblocks), dim3(64), tmp.lds.total, 0, (const J2kBlock *)db.p, nblocks,
                               (const uint8_t *)dby.p, (uint32_t *)dc.p, (const uint16_t *)c->d_tables, (int *)ds.p, tmp.lds,
                               (const ht_sym_t *)nullptr, (const uint32_t *)nullptr, (uint32_t *)dc.p + nsamples + 8,
                               (const uint64_t *)nullptr, (const uint32_t *)nullptr);
            e = hipDeviceSynchronize();
        }
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(coef, dc.p, nsamples * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && status) e = hipMemcpy(status, ds.p, (size_t)nblocks * 4, hipMemcpyDeviceToHost);
    release_all();
    if (e != hipSuccess) { clog(c, LOG_ERROR, "HIP error %s in htj2k_ht_blocks\n", hipGetErrorString(e)); return HTJ2K_ERR_EXTERNAL; }
    return 0;
}

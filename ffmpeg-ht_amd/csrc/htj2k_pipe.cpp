/*
 * htj2k_pipe.cpp -- packets in, frames out: the asynchronous face of the decoder.
 *
 * The reference gets its throughput from FFmpeg's frame threads: N decoder contexts, one packet
 * each (libavcodec/pthread_frame.c:856-889).  On a GPU the equivalent is to keep `depth` device
 * jobs of `batch` frames each in flight: while job i runs its kernels, the host threads parse
 * job i+1 (frames are independent: htj2k_job_parse_batch parses them in parallel, straight into
 * pinned memory) and the caller copies the frames of job i-1 out.  Host parsing, PCIe and the
 * kernels then overlap instead of adding up.
 *
 * Only the public C ABI of include/htj2k_amd.h is used here.  Frames come back in the order
 * their packets went in.  A packet that does not parse fails the batch it is in; the frames of
 * such a batch are then decoded one by one, so that only the bad packet reports an error --
 * the same per-packet error behaviour as the synchronous htj2k_decode().
 */
#include <condition_variable>
#include <deque>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include <string.h>

#include "../../include/htj2k_amd.h"

extern "C" void htj2k_ctx_ref_(htj2k_ctx *ctx);     /* htj2k_device.hip: one more htj2k_close is needed to free the context */
extern "C" int htj2k_ctx_default_depth_(const htj2k_ctx *ctx);   /* htj2k_opts.frames_in_flight */

namespace {

constexpr int PIPE_PAD = 64;           /* AV_INPUT_BUFFER_PADDING_SIZE: the private packet copies carry it */

enum SlotState { SLOT_FREE, SLOT_FILLING, SLOT_RUNNING, SLOT_DONE,
                 SLOT_HELD };          /* all frames handed out as device pointers: the job may not be reused yet */

struct PinBuf { uint8_t *p = nullptr; size_t cap = 0; };   /* page-locked packet memory (htj2k_host_alloc), recycled */

struct Packet {                        /* either a private copy (in page-locked memory) or a reference the caller keeps alive */
    const uint8_t *data = nullptr;
    int size = 0;
    PinBuf own;
    void (*release)(void *) = nullptr;
    void *opaque = nullptr;
};

struct Slot {
    htj2k_job *job = nullptr;
    std::vector<Packet> pkts;
    SlotState state = SLOT_FREE;
    int rc = 0;                        /* result of the batch run */
    int next_out = 0;                  /* next frame of the batch to hand out */
    bool device_handout = false;       /* some frame of the batch left as device pointers */
    long release_at = 0;               /* SLOT_HELD: free again once this many batches have been handed out in full */
    int outstanding = 0;               /* device frames handed out by htj2k_pipe_receive_device_ref and not released yet */
    std::vector<uint8_t> ref_out;      /* ... which ones (by position in the batch): a token releases its own frame, once */
    uint32_t generation = 0;           /* bumped every time the slot is started: stale release tokens are ignored */
    std::vector<htj2k_job *> retry;    /* failed batch, device hand-out: one single-frame job per frame (the pointers stay valid
                                        * as long as the slot's) */
    std::thread worker;
};

}  // namespace

struct htj2k_pipe {
    htj2k_ctx *ctx = nullptr;
    int batch = 1, depth = 2;
    /* 2 depth - 1 slots: `depth` batches may be in flight (being filled, running or waiting to be received) while up to
     * depth - 1 more are HELD because their frames went out as device pointers.  (With `depth` slots in a strict ring a
     * consumer of device frames had ONE batch in flight: a slot came free only when the batches behind it had been
     * received, so the next batch could not even start before that -- 1 300 frames per second where PCIe carries 2 700.)
     * A new batch takes any FREE slot; frames still come out in the order their packets went in (`started`). */
    std::vector<Slot> slots;
    int cur = -1;                      /* slot being filled, -1: none */
    std::deque<int> started;           /* slots in the order their batches were started: the front one is handed out */
    std::mutex m;
    std::condition_variable cv;
    htj2k_job *single = nullptr;       /* per-frame retry of a failed batch */
    std::vector<PinBuf> spare;         /* packet buffers waiting for re-use */
    long batches_out = 0;              /* batches whose frames have all been handed out */
    bool closing = false;              /* htj2k_pipe_close was called while device frames were out: the last release frees */
};

/* caller holds the lock */
static void drop_packet(htj2k_pipe *p, Packet &pk)
{
    if (pk.release) pk.release(pk.opaque);
    if (pk.own.p) p->spare.push_back(pk.own);
    pk = Packet();
}

static void run_slot(htj2k_pipe *p, Slot *s)
{
    std::vector<const uint8_t *> ptr(s->pkts.size());
    std::vector<int> len(s->pkts.size());
    std::vector<uint8_t> pinned(s->pkts.size());
    for (size_t i = 0; i < s->pkts.size(); i++) { ptr[i] = s->pkts[i].data; len[i] = s->pkts[i].size; pinned[i] = s->pkts[i].own.p != nullptr; }
    /* the pipe's own packet copies sit in page-locked memory and outlive the upload: no second staging copy */
    int r = htj2k_job_parse_batch_ex(p->ctx, ptr.data(), len.data(), (int)ptr.size(), pinned.data(), &s->job);
    if (r >= 0) r = htj2k_job_upload(p->ctx, s->job);
    if (r >= 0) r = htj2k_job_run(p->ctx, s->job);
    if (r >= 0) r = htj2k_job_wait(p->ctx, s->job);
    {
        std::lock_guard<std::mutex> lk(p->m);
        s->rc = r;
        s->state = SLOT_DONE;
    }
    p->cv.notify_all();
}

/* caller holds the lock */
static void start_slot(htj2k_pipe *p, Slot &s)
{
    s.state = SLOT_RUNNING;
    s.rc = 0;
    s.next_out = 0;
    s.device_handout = false;
    s.outstanding = 0;
    s.ref_out.clear();
    s.generation++;
    if (s.worker.joinable()) s.worker.join();
    s.worker = std::thread(run_slot, p, &s);
}

extern "C" int htj2k_pipe_open(htj2k_ctx *ctx, int batch, int depth, htj2k_pipe **pipe)
{
    if (ctx && depth == 0) depth = htj2k_ctx_default_depth_(ctx);
    if (!ctx || !pipe || batch < 1 || batch > 256 || depth < 1 || depth > 16) return HTJ2K_ERR_EINVAL;
    htj2k_pipe *p = new (std::nothrow) htj2k_pipe();
    if (!p) return HTJ2K_ERR_ENOMEM;
    p->ctx = ctx;
    htj2k_ctx_ref_(ctx);               /* given back by the last act of htj2k_pipe_close (which may come after the caller's htj2k_close) */
    p->batch = batch;
    p->depth = depth;
    p->slots.resize((size_t)(2 * depth - 1));
    *pipe = p;
    return 0;
}

/* caller holds the lock.  The slot a packet goes into: the one being filled, or a FREE one if fewer than `depth` batches
 * are in flight; nullptr: receive first (HTJ2K_ERR_EAGAIN) */
static Slot *fill_slot(htj2k_pipe *p)
{
    if (p->cur >= 0) return &p->slots[(size_t)p->cur];
    if ((int)p->started.size() >= p->depth) return nullptr;
    for (size_t i = 0; i < p->slots.size(); i++)
        if (p->slots[i].state == SLOT_FREE) { p->cur = (int)i; return &p->slots[i]; }
    return nullptr;
}

/* caller holds the lock */
static void start_current(htj2k_pipe *p)
{
    start_slot(p, p->slots[(size_t)p->cur]);
    p->started.push_back(p->cur);
    p->cur = -1;
}

static int queue_packet(htj2k_pipe *p, Slot &s, const Packet &pk)
{
    s.pkts.push_back(pk);
    s.state = SLOT_FILLING;
    if ((int)s.pkts.size() >= p->batch) start_current(p);
    return 0;
}

extern "C" int htj2k_pipe_send(htj2k_pipe *p, const uint8_t *pkt, int size)
{
    if (!p || !pkt || size <= 0) return HTJ2K_ERR_EINVAL;
    std::unique_lock<std::mutex> lk(p->m);
    Slot *sp = fill_slot(p);
    if (!sp) return HTJ2K_ERR_EAGAIN;                      /* receive first */
    /* private copy with the input padding an AVPacket carries (AV_INPUT_BUFFER_PADDING_SIZE), in page-locked memory:
     * the H2D transfer starts from this very copy */
    Packet pk;
    const size_t need = (size_t)size + PIPE_PAD;
    for (size_t i = 0; i < p->spare.size(); i++)
        if (p->spare[i].cap >= need) { pk.own = p->spare[i]; p->spare.erase(p->spare.begin() + (long)i); break; }
    if (!pk.own.p) {
        if (p->spare.size() > 4 * (size_t)p->batch * (size_t)p->depth) {   /* too small for this stream: do not hoard them */
            htj2k_host_free(p->ctx, p->spare.back().p);
            p->spare.pop_back();
        }
        pk.own.cap = need + need / 4;
        pk.own.p = (uint8_t *)htj2k_host_alloc(p->ctx, pk.own.cap);
        if (!pk.own.p) return HTJ2K_ERR_ENOMEM;
    }
    memcpy(pk.own.p, pkt, (size_t)size);
    memset(pk.own.p + size, 0, PIPE_PAD);
    pk.data = pk.own.p;
    pk.size = size;
    return queue_packet(p, *sp, pk);
}

/* no copy: `pkt` stays valid until `release(opaque)` is called, which happens when the packet's
 * frame has been received or skipped (or the pipe is closed) -- what an AVPacket reference gives */
extern "C" int htj2k_pipe_send_ref(htj2k_pipe *p, const uint8_t *pkt, int size, void (*release)(void *), void *opaque)
{
    if (!p || !pkt || size <= 0) return HTJ2K_ERR_EINVAL;
    std::unique_lock<std::mutex> lk(p->m);
    Slot *sp = fill_slot(p);
    if (!sp) return HTJ2K_ERR_EAGAIN;
    Packet pk;
    pk.data = pkt;
    pk.size = size;
    pk.release = release;
    pk.opaque = opaque;
    return queue_packet(p, *sp, pk);
}

extern "C" int htj2k_pipe_flush(htj2k_pipe *p)
{
    if (!p) return HTJ2K_ERR_EINVAL;
    std::unique_lock<std::mutex> lk(p->m);
    if (p->cur >= 0 && p->slots[(size_t)p->cur].state == SLOT_FILLING) start_current(p);
    return 0;
}

/* the slot whose frame is due; waits for its job.  HTJ2K_ERR_EAGAIN: nothing is in flight (send or flush first) */
static int due_slot(htj2k_pipe *p, std::unique_lock<std::mutex> &lk, Slot **out)
{
    if (p->started.empty()) return HTJ2K_ERR_EAGAIN;
    Slot &s = p->slots[(size_t)p->started.front()];
    p->cv.wait(lk, [&] { return s.state == SLOT_DONE; });
    *out = &s;
    return 0;
}

/* caller holds the lock.  A batch whose frames went out as device pointers keeps its job untouched until the frames of
 * depth - 1 further batches have been handed out (what htj2k_pipe_receive_device promises): its slot is HELD, and
 * htj2k_pipe_send answers EAGAIN when it comes round to it. */
static void pop_frame(htj2k_pipe *p, Slot &s)
{
    drop_packet(p, s.pkts[s.next_out]);
    if (++s.next_out >= (int)s.pkts.size()) {
        s.pkts.clear();
        s.next_out = 0;
        p->batches_out++;
        if (s.device_handout && p->depth > 1) {
            s.state = SLOT_HELD;
            s.release_at = p->batches_out + p->depth - 1;
        } else {
            s.state = s.outstanding > 0 ? SLOT_HELD : SLOT_FREE;
            s.release_at = 0;
        }
        for (Slot &o : p->slots)
            if (o.state == SLOT_HELD && p->batches_out >= o.release_at && o.outstanding == 0) o.state = SLOT_FREE;
        p->started.pop_front();
    }
}

extern "C" int htj2k_pipe_info(htj2k_pipe *p, htj2k_info *info)
{
    if (!p || !info) return HTJ2K_ERR_EINVAL;
    std::unique_lock<std::mutex> lk(p->m);
    Slot *s = nullptr;
    int r = due_slot(p, lk, &s);
    if (r < 0) return r;
    if (s->rc >= 0) return htj2k_job_frame_info(s->job, s->next_out, info);
    /* failed batch: the per-frame retry in htj2k_pipe_receive decides; probe this packet alone */
    const Packet &pk = s->pkts[s->next_out];
    return htj2k_probe(p->ctx, pk.data, pk.size, info);
}

extern "C" int htj2k_pipe_receive(htj2k_pipe *p, htj2k_frame *frame)
{
    if (!p || !frame) return HTJ2K_ERR_EINVAL;
    std::unique_lock<std::mutex> lk(p->m);
    Slot *s = nullptr;
    int r = due_slot(p, lk, &s);
    if (r < 0) return r;
    /* a finished slot belongs to the (single) consumer: copy out without holding the lock, so
     * that the workers of the other slots and htj2k_pipe_send are not held up by the D2H */
    lk.unlock();
    if (s->rc >= 0) {
        r = htj2k_job_download_frame(p->ctx, s->job, s->next_out, frame);
    } else {
        const Packet &pk = s->pkts[s->next_out];
        r = htj2k_job_parse(p->ctx, pk.data, pk.size, &p->single);
        if (r >= 0) r = htj2k_job_upload(p->ctx, p->single);
        if (r >= 0) r = htj2k_job_run(p->ctx, p->single);
        if (r >= 0) r = htj2k_job_download(p->ctx, p->single, frame);
    }
    lk.lock();
    pop_frame(p, *s);
    return r;
}

extern "C" int htj2k_pipe_receive_device(htj2k_pipe *p, htj2k_frame *frame)
{
    if (!p || !frame) return HTJ2K_ERR_EINVAL;
    std::unique_lock<std::mutex> lk(p->m);
    Slot *s = nullptr;
    int r = due_slot(p, lk, &s);
    if (r < 0) return r;
    lk.unlock();
    if (s->rc >= 0) {
        r = htj2k_job_device_frame(p->ctx, s->job, s->next_out, frame);
    } else {                                               /* failed batch: this packet alone, in a job of its own that lives as long as the slot's */
        const Packet &pk = s->pkts[s->next_out];
        if ((int)s->retry.size() <= s->next_out) s->retry.resize((size_t)s->next_out + 1, nullptr);
        htj2k_job **one = &s->retry[(size_t)s->next_out];
        r = htj2k_job_parse(p->ctx, pk.data, pk.size, one);
        if (r >= 0) r = htj2k_job_upload(p->ctx, *one);
        if (r >= 0) r = htj2k_job_run(p->ctx, *one);
        if (r >= 0) r = htj2k_job_wait(p->ctx, *one);
        if (r >= 0) r = htj2k_job_device_frame(p->ctx, *one, 0, frame);
    }
    lk.lock();
    if (r >= 0) s->device_handout = true;
    pop_frame(p, *s);
    return r;
}

/* as htj2k_pipe_receive_device, for consumers that keep frames for as long as they like (a reference-counted
 * AVFrame): the batch's job is not reused before every frame handed out this way has been given back with
 * htj2k_pipe_release_device(token) -- from any thread */
extern "C" int htj2k_pipe_receive_device_ref(htj2k_pipe *p, htj2k_frame *frame, uint64_t *token)
{
    if (!p || !frame || !token) return HTJ2K_ERR_EINVAL;
    int slot_index = 0, frame_index = 0;
    uint32_t gen = 0;
    {
        std::unique_lock<std::mutex> lk(p->m);
        Slot *s = nullptr;
        int r = due_slot(p, lk, &s);
        if (r < 0) return r;
        slot_index = (int)(s - p->slots.data());
        frame_index = s->next_out;
        gen = s->generation;
        s->outstanding++;                                  /* before the hand-out: pop_frame must see it */
        if ((int)s->ref_out.size() <= frame_index) s->ref_out.resize((size_t)frame_index + 1, 0);
        s->ref_out[(size_t)frame_index] = 1;
    }
    const int r = htj2k_pipe_receive_device(p, frame);
    std::unique_lock<std::mutex> lk(p->m);
    Slot &s = p->slots[(size_t)slot_index];
    if (r < 0) {
        if (s.generation == gen && s.outstanding > 0) { s.outstanding--; s.ref_out[(size_t)frame_index] = 0; }
        if (s.state == SLOT_HELD && s.outstanding == 0 && p->batches_out >= s.release_at) s.state = SLOT_FREE;
        return r;
    }
    s.release_at = 0;                                      /* explicit release rules this slot, not the depth - 1 rule */
    if (s.state == SLOT_HELD && s.outstanding == 0) s.state = SLOT_FREE;
    *token = ((uint64_t)(uint32_t)slot_index << 48) | ((uint64_t)((uint32_t)frame_index & 0xFFFFu) << 32) | gen;
    return r;
}

/* caller holds no lock.  What htj2k_pipe_close leaves for later when device frames are still out */
static void destroy_pipe(htj2k_pipe *p)
{
    for (Slot &s : p->slots) {
        if (s.job) htj2k_job_free(p->ctx, s.job);
        for (htj2k_job *j : s.retry) if (j) htj2k_job_free(p->ctx, j);
    }
    if (p->single) htj2k_job_free(p->ctx, p->single);
    for (PinBuf &b : p->spare) htj2k_host_free(p->ctx, b.p);
    htj2k_ctx *ctx = p->ctx;
    delete p;
    htj2k_close(ctx);                  /* the pipe's reference */
}

extern "C" int htj2k_pipe_release_device(htj2k_pipe *p, uint64_t token)
{
    if (!p) return HTJ2K_ERR_EINVAL;
    bool last = false;
    {
        std::unique_lock<std::mutex> lk(p->m);
        const size_t idx = (size_t)(token >> 48), fi = (size_t)((token >> 32) & 0xFFFFu);
        if (idx >= p->slots.size()) return HTJ2K_ERR_EINVAL;
        Slot &s = p->slots[idx];
        if (s.generation != (uint32_t)token || s.outstanding <= 0 || fi >= s.ref_out.size() || !s.ref_out[fi]) return HTJ2K_ERR_EINVAL;
        s.ref_out[fi] = 0;
        if (--s.outstanding == 0 && s.state == SLOT_HELD && p->batches_out >= s.release_at) s.state = SLOT_FREE;
        if (p->closing) {
            last = true;
            for (const Slot &o : p->slots) if (o.outstanding > 0) last = false;
        }
    }
    if (last) destroy_pipe(p);         /* closed long ago: this was the last frame the consumer held */
    return 0;
}

/* drop the next frame without copying it out (e.g. after htj2k_pipe_info reported an error) */
extern "C" int htj2k_pipe_skip(htj2k_pipe *p)
{
    if (!p) return HTJ2K_ERR_EINVAL;
    std::unique_lock<std::mutex> lk(p->m);
    Slot *s = nullptr;
    int r = due_slot(p, lk, &s);
    if (r < 0) return r;
    pop_frame(p, *s);
    return 0;
}

/* Frames handed out by htj2k_pipe_receive_device_ref may outlive the pipe (and the context: the pipe holds a reference):
 * with such frames out, close stops the workers and drops what is queued, and the last htj2k_pipe_release_device frees the
 * jobs -- the planes the consumer still reads -- and the pipe itself.  The handle stays valid for that call only. */
extern "C" void htj2k_pipe_close(htj2k_pipe *p)
{
    if (!p) return;
    for (Slot &s : p->slots)
        if (s.worker.joinable()) s.worker.join();
    bool out = false;
    {
        std::lock_guard<std::mutex> lk(p->m);
        for (Slot &s : p->slots) {
            for (size_t i = (s.state == SLOT_FREE || s.state == SLOT_HELD ? s.pkts.size() : (size_t)s.next_out); i < s.pkts.size(); i++)
                drop_packet(p, s.pkts[i]);
            s.pkts.clear();
            if (s.outstanding > 0) { out = true; s.state = SLOT_HELD; s.release_at = 0; }
        }
        p->started.clear();
        p->cur = -1;
        p->closing = out;
    }
    if (!out) destroy_pipe(p);
}

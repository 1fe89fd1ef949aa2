// group.hip — several GPUs behind the C ABI: aukit_partition, aukit_group_* (include/aukit_hip.h).
//
// No function of the path reads another stream, so a batch shards by stream index and every device runs the single-GPU path
// on its shard; the only data movement is one scatter of input bytes and one gather of results.  A Lua host is ONE process (one Lua state)
// that owns every GPU of the node, so the group is a set of contexts in this process — one per device, each with its own stream and
// scratch — and the transport is device-to-device DMA between them:
//   * peer copies (default): hipMemcpyPeerAsync on the RECEIVING context's stream for a scatter / the SENDING context's stream for a
//     gather, one per peer, all in flight at once (each peer's share crosses its own xGMI link; nothing is serialised on the root's
//     stream), ordered against the kernels on both sides with events;
//   * RCCL (AUKIT_GROUP_TRANSPORT=rccl, distinct devices only): ncclCommInitAll once per group, then one ncclGroupStart / ncclGroupEnd
//     around the root's ncclSend to every peer and every peer's ncclRecv — what the multi-process path (aukit_amd/shard.py over
//     torch.distributed) does with one process per GPU.  librccl is dlopen'ed: the library has no link-time dependency on it.
// The root's own shard is a zero-copy view of its bytes.  Devices may repeat in a group (several contexts on one GPU): that is how the
// tests of a one-GPU box drive every line of the peer path.
#include <dlfcn.h>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include "common.h"

struct aukit_group {
    std::vector<aukit_ctx *> ctx;
    std::vector<int> dev;
    std::vector<hipEvent_t> ev;   // one per member
    bool rccl = false;
    void *lib = nullptr;
    std::vector<void *> comms;
    int (*p_group_start)() = nullptr;
    int (*p_group_end)() = nullptr;
    int (*p_send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*p_recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*p_destroy)(void *) = nullptr;
    // aukit_group_run: one worker thread per member, parked on `cv` between runs
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv, cv_done;
    uint64_t run_id = 0;                 // bumped by every aukit_group_run
    uint32_t pending = 0;                // members still working on the current run
    bool quit = false;
    const aukit_group_call *calls = nullptr;
    uint32_t n_per = 0;
    std::vector<int> status;
    std::vector<std::string> message;
    std::vector<double> t_start, t_end;  // ms since the run's epoch
    std::chrono::steady_clock::time_point epoch;
};

namespace aukit {

// one device-to-device message: `bytes` from src (member `from`) to dst (member `to`)
struct GroupMsg { const void *src; void *dst; size_t bytes; uint32_t from, to; };

// Runs the messages.  Peer copies go on the stream of `on_receiver ? to : from`; the other side's stream is ordered with events:
// before — the copy waits for what the other side has queued so far; after — `join` (a member) waits for every copy.
static int group_move(aukit_group *g, const std::vector<GroupMsg> &msgs, bool on_receiver, uint32_t join) {
    if (msgs.empty()) return AUKIT_OK;
    const uint32_t W = (uint32_t)g->ctx.size();
    std::vector<char> involved(W, 0);
    for (const GroupMsg &m : msgs) { involved[m.from] = 1; involved[m.to] = 1; }
    // every copy starts after what its two ends have queued so far
    for (uint32_t r = 0; r < W; r++) {
        if (!involved[r]) continue;
        AUKIT_HIP_CHECK(hipSetDevice(g->dev[r]));
        AUKIT_HIP_CHECK(hipEventRecord(g->ev[r], g->ctx[r]->stream));
    }
    if (g->rccl) {
        for (const GroupMsg &m : msgs) {   // (RCCL orders a message on the two streams it is given; the other members' work was recorded above)
            AUKIT_HIP_CHECK(hipSetDevice(g->dev[m.from]));
            AUKIT_HIP_CHECK(hipStreamWaitEvent(g->ctx[m.from]->stream, g->ev[m.to], 0));
            AUKIT_HIP_CHECK(hipSetDevice(g->dev[m.to]));
            AUKIT_HIP_CHECK(hipStreamWaitEvent(g->ctx[m.to]->stream, g->ev[m.from], 0));
        }
        if (g->p_group_start()) return fail(AUKIT_E_HIP, "ncclGroupStart failed");
        for (const GroupMsg &m : msgs) {
            if (!m.bytes) continue;
            if (g->p_send(m.src, m.bytes, 1 /* ncclUint8 */, (int)m.to, g->comms[m.from], g->ctx[m.from]->stream)) return fail(AUKIT_E_HIP, "ncclSend failed");
            if (g->p_recv(m.dst, m.bytes, 1, (int)m.from, g->comms[m.to], g->ctx[m.to]->stream)) return fail(AUKIT_E_HIP, "ncclRecv failed");
        }
        if (g->p_group_end()) return fail(AUKIT_E_HIP, "ncclGroupEnd failed");
    } else {
        for (const GroupMsg &m : msgs) {
            const uint32_t on = on_receiver ? m.to : m.from, other = on_receiver ? m.from : m.to;
            AUKIT_HIP_CHECK(hipSetDevice(g->dev[on]));
            if (other != on) AUKIT_HIP_CHECK(hipStreamWaitEvent(g->ctx[on]->stream, g->ev[other], 0));
            if (m.bytes) AUKIT_HIP_CHECK(hipMemcpyPeerAsync(m.dst, g->dev[m.to], m.src, g->dev[m.from], m.bytes, g->ctx[on]->stream));
        }
    }
    // the member that goes on with the data waits for every stream that carried a copy
    std::vector<char> carried(W, 0);
    for (const GroupMsg &m : msgs) { if (g->rccl) { carried[m.from] = carried[m.to] = 1; } else carried[on_receiver ? m.to : m.from] = 1; }
    for (uint32_t r = 0; r < W; r++) {
        if (!carried[r] || r == join) continue;
        AUKIT_HIP_CHECK(hipSetDevice(g->dev[r]));
        AUKIT_HIP_CHECK(hipEventRecord(g->ev[r], g->ctx[r]->stream));
        AUKIT_HIP_CHECK(hipSetDevice(g->dev[join]));
        AUKIT_HIP_CHECK(hipStreamWaitEvent(g->ctx[join]->stream, g->ev[r], 0));
    }
    return AUKIT_OK;
}

static_assert(sizeof(aukit_group_call) == 160, "aukit_group_call layout (aukit_amd/_native.py GroupCall, aukit_amd/lua/aukit.lua)");
// one call of a member's list: the single-GPU entry point it names, on the member's context
static int group_call(aukit_ctx *ctx, const aukit_group_call &c) {
    switch (c.op) {
    case AUKIT_GOP_NONE: return AUKIT_OK;
    case AUKIT_GOP_DECODE: return aukit_decode(ctx, c.batch, c.desc, c.dtype, c.out_audio);
    case AUKIT_GOP_DECODE_RESAMPLE: return aukit_decode_resample(ctx, c.batch, c.desc, c.new_rate, c.interp, c.dtype, c.out_audio);
    case AUKIT_GOP_STREAM_DECODE: return aukit_stream_decode(ctx, c.batch, c.desc, c.interp, c.mono, c.dtype, c.out_audio, c.out_chunks);
    case AUKIT_GOP_RESAMPLE: return aukit_resample(ctx, c.audio, c.new_rate, c.interp, c.out_audio);
    case AUKIT_GOP_MONO: return aukit_mono(ctx, c.audio, c.out_audio);
    case AUKIT_GOP_EFFECT: return aukit_effect(ctx, c.audio, c.effect_id, c.args, c.nargs);
    case AUKIT_GOP_DFPWM_ENCODE: return aukit_dfpwm_encode(ctx, c.audio, c.interleaved, c.out_batch);
    case AUKIT_GOP_DFPWM_TRANSCODE_MONO: return aukit_dfpwm_transcode_mono(ctx, c.batch, c.channels, c.out_batch);
    case AUKIT_GOP_ENCODE_PCM: return aukit_encode_pcm(ctx, c.audio, c.bit_depth, c.data_type, c.interleaved, c.out_audio);
    case AUKIT_GOP_SYNC: return aukit_ctx_sync(ctx);
    }
    return fail(AUKIT_E_ARG, "aukit_group_run: unknown op %d", c.op);
}

static void group_worker(aukit_group *g, uint32_t r) {
    (void)hipSetDevice(g->dev[r]);   // (the device is per-thread state of the HIP runtime)
    uint64_t seen = 0;
    for (;;) {
        const aukit_group_call *calls;
        uint32_t n_per;
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv.wait(lk, [&] { return g->quit || g->run_id != seen; });
            if (g->quit) return;
            seen = g->run_id;
            calls = g->calls; n_per = g->n_per;
        }
        const auto t0 = std::chrono::steady_clock::now();
        int rc = AUKIT_OK;
        std::string msg;
        try {   // (an exception must not leave this thread — std::terminate would take the host process with it: ADVICE r04)
            for (uint32_t k = 0; k < n_per; k++) {
                const int r1 = group_call(g->ctx[r], calls[(size_t)r * n_per + k]);
                if (r1 && !rc) { rc = r1; msg = aukit_last_error(); }   // (the message is this thread's: carried to the caller below)
                if (r1) break;   // a member's list is a pipeline: what follows a failed call would read its missing output
            }
        } catch (const std::bad_alloc &) {
            if (!rc) { rc = AUKIT_E_NOMEM; msg = "out of host memory in a group member's call list"; }
        } catch (const std::exception &e) {
            if (!rc) { rc = AUKIT_E_HIP; msg = std::string("exception in a group member's call list: ") + e.what(); }
        }
        if (hipStreamSynchronize(g->ctx[r]->stream) != hipSuccess && !rc) { rc = AUKIT_E_HIP; msg = "hipStreamSynchronize failed in a group member"; }
        const auto t1 = std::chrono::steady_clock::now();
        {
            std::lock_guard<std::mutex> lk(g->mu);
            g->status[r] = rc; g->message[r] = msg;
            g->t_start[r] = std::chrono::duration<double, std::milli>(t0 - g->epoch).count();
            g->t_end[r] = std::chrono::duration<double, std::milli>(t1 - g->epoch).count();
            if (--g->pending == 0) g->cv_done.notify_all();
        }
    }
}

}  // namespace aukit

using namespace aukit;

extern "C" {

int aukit_group_run(aukit_group *g, const aukit_group_call *calls, uint32_t n_per_member) {
    if (!g || (!calls && n_per_member)) return fail(AUKIT_E_ARG, "null argument");
    const uint32_t W = (uint32_t)g->ctx.size();
    if (g->workers.empty()) {   // started with the first run: a group that only scatters and gathers has no threads
        g->status.assign(W, 0); g->message.assign(W, ""); g->t_start.assign(W, 0); g->t_end.assign(W, 0);
        for (uint32_t r = 0; r < W; r++) g->workers.emplace_back(group_worker, g, r);
    }
    {
        std::unique_lock<std::mutex> lk(g->mu);
        g->calls = calls; g->n_per = n_per_member;
        g->pending = W;
        g->epoch = std::chrono::steady_clock::now();
        g->run_id++;
        g->cv.notify_all();
        g->cv_done.wait(lk, [&] { return g->pending == 0; });
    }
    for (uint32_t r = 0; r < W; r++)
        if (g->status[r]) return fail(g->status[r], "%s", g->message[r].c_str());
    return AUKIT_OK;
}

int aukit_group_last_run(const aukit_group *g, double *start_ms, double *end_ms) {
    if (!g || !start_ms || !end_ms) return fail(AUKIT_E_ARG, "null argument");
    if (g->t_start.empty()) return fail(AUKIT_E_ARG, "no aukit_group_run yet");
    const double t0 = *std::min_element(g->t_start.begin(), g->t_start.end());
    for (size_t r = 0; r < g->ctx.size(); r++) { start_ms[r] = g->t_start[r] - t0; end_ms[r] = g->t_end[r] - t0; }
    return AUKIT_OK;
}

// Contiguous stream ranges per rank, balanced by input bytes: rank g takes the streams whose cumulative byte midpoint falls in
// [g / world, (g + 1) / world) of the total — contiguous (outputs concatenate in rank order) and within one stream of the ideal split.
// cuts[g] .. cuts[g + 1] are rank g's streams.  (The same arithmetic as aukit_amd/shard.py, which calls this.)
int aukit_partition(const uint64_t *sizes, uint32_t n, uint32_t world, uint32_t *cuts) {
    if (!cuts || (n && !sizes)) return fail(AUKIT_E_ARG, "null argument");
    if (world < 1) return fail(AUKIT_E_ARG, "world must be >= 1");
    double total = 0;
    for (uint32_t i = 0; i < n; i++) total += (double)sizes[i];
    if (n == 0) { for (uint32_t g = 0; g <= world; g++) cuts[g] = 0; return AUKIT_OK; }
    if (!(total > 0)) { for (uint32_t g = 0; g <= world; g++) cuts[g] = (uint32_t)(((uint64_t)n * g) / world); return AUKIT_OK; }
    uint32_t g = 0;
    double cum = 0;
    cuts[0] = 0;
    for (uint32_t i = 0; i < n; i++) {
        cum += (double)sizes[i];
        const double mid = cum - (double)sizes[i] / 2;
        long long owner = (long long)(mid / total * (double)world);
        if (owner > (long long)world - 1) owner = (long long)world - 1;
        while ((long long)g < owner) cuts[++g] = i;   // the first stream owned by a rank beyond g starts that rank (and the empty ones between)
    }
    while (g < world) cuts[++g] = n;
    return AUKIT_OK;
}

int aukit_group_create(aukit_group **out, const int *devices, uint32_t n_devices) {
    if (!out || !devices || !n_devices) return fail(AUKIT_E_ARG, "null argument");
    aukit_group *g = new aukit_group();
    bool distinct = true;
    for (uint32_t r = 0; r < n_devices; r++) {
        for (uint32_t q = 0; q < r; q++) distinct = distinct && devices[q] != devices[r];
        aukit_ctx *c = nullptr;
        int rc = aukit_ctx_create(&c, devices[r]);
        if (rc) { for (aukit_ctx *x : g->ctx) aukit_ctx_destroy(x); delete g; return rc; }
        g->ctx.push_back(c);
        g->dev.push_back(devices[r]);
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { aukit_group_destroy(g); return fail(AUKIT_E_HIP, "hipEventCreate failed"); }
        g->ev.push_back(e);
    }
    // peer access between distinct devices (a no-op where it is already on; not fatal where the topology has none: the copies then stage)
    for (uint32_t r = 0; r < n_devices; r++)
        for (uint32_t q = 0; q < n_devices; q++) {
            if (devices[r] == devices[q]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[r], devices[q]) == hipSuccess && can) {
                (void)hipSetDevice(devices[r]);
                hipError_t e = hipDeviceEnablePeerAccess(devices[q], 0);
                if (e != hipSuccess) (void)hipGetLastError();   // hipErrorPeerAccessAlreadyEnabled
            }
        }
    const char *tr = getenv("AUKIT_GROUP_TRANSPORT");
    if (tr && !strcmp(tr, "rccl") && distinct) {
        g->lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!g->lib) g->lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        int (*init_all)(void **, int, const int *) = nullptr;
        if (g->lib) {
            init_all = reinterpret_cast<int (*)(void **, int, const int *)>(dlsym(g->lib, "ncclCommInitAll"));
            g->p_group_start = reinterpret_cast<int (*)()>(dlsym(g->lib, "ncclGroupStart"));
            g->p_group_end = reinterpret_cast<int (*)()>(dlsym(g->lib, "ncclGroupEnd"));
            g->p_send = reinterpret_cast<int (*)(const void *, size_t, int, int, void *, hipStream_t)>(dlsym(g->lib, "ncclSend"));
            g->p_recv = reinterpret_cast<int (*)(void *, size_t, int, int, void *, hipStream_t)>(dlsym(g->lib, "ncclRecv"));
            g->p_destroy = reinterpret_cast<int (*)(void *)>(dlsym(g->lib, "ncclCommDestroy"));
        }
        if (!init_all || !g->p_group_start || !g->p_group_end || !g->p_send || !g->p_recv || !g->p_destroy) { aukit_group_destroy(g); return fail(AUKIT_E_UNSUPPORTED, "AUKIT_GROUP_TRANSPORT=rccl: librccl could not be loaded"); }
        g->comms.assign(n_devices, nullptr);
        if (init_all(g->comms.data(), (int)n_devices, devices)) { g->comms.clear(); aukit_group_destroy(g); return fail(AUKIT_E_HIP, "ncclCommInitAll failed"); }
        g->rccl = true;
    }
    *out = g;
    return AUKIT_OK;
}

void aukit_group_destroy(aukit_group *g) {
    if (!g) return;
    if (!g->workers.empty()) {
        { std::lock_guard<std::mutex> lk(g->mu); g->quit = true; }
        g->cv.notify_all();
        for (std::thread &t : g->workers) t.join();
        g->workers.clear();
    }
    for (size_t r = 0; r < g->ctx.size(); r++) { (void)hipSetDevice(g->dev[r]); (void)hipStreamSynchronize(g->ctx[r]->stream); }
    if (g->p_destroy) for (void *c : g->comms) if (c) g->p_destroy(c);
    for (size_t r = 0; r < g->ev.size(); r++) { (void)hipSetDevice(g->dev[r]); if (g->ev[r]) (void)hipEventDestroy(g->ev[r]); }
    for (aukit_ctx *c : g->ctx) aukit_ctx_destroy(c);
    if (g->lib) dlclose(g->lib);
    delete g;
}

int aukit_group_info(const aukit_group *g, uint32_t *n_devices, int *transport) {
    if (!g) return fail(AUKIT_E_ARG, "null argument");
    if (n_devices) *n_devices = (uint32_t)g->ctx.size();
    if (transport) *transport = g->rccl ? 1 : 0;
    return AUKIT_OK;
}

aukit_ctx *aukit_group_ctx(aukit_group *g, uint32_t rank) { return g && rank < g->ctx.size() ? g->ctx[rank] : nullptr; }

int aukit_group_sync(aukit_group *g) {
    if (!g) return fail(AUKIT_E_ARG, "null argument");
    for (size_t r = 0; r < g->ctx.size(); r++) { int rc = aukit_ctx_sync(g->ctx[r]); if (rc) return rc; }
    return AUKIT_OK;
}

// `whole` lives on member `root`'s device.  shards[r] (r = 0 .. size-1) receives member r's streams (cuts[r] .. cuts[r + 1], byte-balanced:
// aukit_partition) as a batch of member r's context; the root's is a view of `whole` (keep `whole` alive while it is in use).
int aukit_group_scatter(aukit_group *g, uint32_t root, const aukit_batch *whole, aukit_batch **shards, uint32_t *cuts) {
    if (!g || !whole || !shards || !cuts) return fail(AUKIT_E_ARG, "null argument");
    const uint32_t W = (uint32_t)g->ctx.size();
    if (root >= W) return fail(AUKIT_E_ARG, "root %u out of range", root);
    std::vector<uint64_t> sizes(whole->n);
    for (uint32_t s = 0; s < whole->n; s++) sizes[s] = whole->off[s + 1] - whole->off[s];
    int rc = aukit_partition(sizes.data(), whole->n, W, cuts);
    if (rc) return rc;
    std::vector<GroupMsg> msgs;
    // a failure part-way must not leave the caller with a half-filled shards[]: what this call allocated is freed and nulled
    auto undo = [&](int code) { for (uint32_t q = 0; q < W; q++) if (shards[q]) { aukit_batch_free(shards[q]); shards[q] = nullptr; } return code; };
    for (uint32_t r = 0; r < W; r++) {
        const uint32_t lo = cuts[r], hi = cuts[r + 1];
        std::vector<uint64_t> off((size_t)(hi - lo) + 1);
        for (uint32_t s = lo; s <= hi; s++) off[s - lo] = whole->off[s] - whole->off[lo];
        const uint64_t bytes = off.back();
        if (shards[r]) { aukit_batch_free(shards[r]); shards[r] = nullptr; }
        if (r == root) {
            if ((rc = aukit_batch_wrap_device(g->ctx[r], &shards[r], whole->data() + whole->off[lo], off.data(), hi - lo))) return undo(rc);
            continue;
        }
        // an owned batch of member r's device with the same offsets
        if (hipSetDevice(g->dev[r]) != hipSuccess) return undo(fail(AUKIT_E_HIP, "hipSetDevice failed"));
        aukit_batch *b = new aukit_batch();
        b->n = hi - lo; b->off = off; b->front_pad = 64; b->cap = (size_t)bytes + 128; b->own = true;
        if (hipMalloc((void **)&b->base, b->cap) != hipSuccess) { delete b; return undo(fail(AUKIT_E_NOMEM, "hipMalloc(%zu) failed", b->cap)); }
        if (hipMalloc((void **)&b->d_off, (off.size()) * 8) != hipSuccess) { (void)hipFree(b->base); delete b; return undo(fail(AUKIT_E_NOMEM, "hipMalloc failed")); }
        if ((rc = h2d_table(g->ctx[r], b->d_off, off.data(), off.size() * 8))) { aukit_batch_free(b); return undo(rc); }
        b->version = 1;
        shards[r] = b;
        msgs.push_back(GroupMsg{whole->data() + whole->off[lo], b->data(), (size_t)bytes, root, r});
    }
    rc = group_move(g, msgs, true, root);
    return rc ? undo(rc) : AUKIT_OK;
}

// parts[r] is member r's result (same channel count, rate, dtype); *whole — on member `root` — gets every stream of every part in rank order.
int aukit_group_gather_audio(aukit_group *g, uint32_t root, aukit_audio *const *parts, aukit_audio **whole) {
    if (!g || !parts || !whole) return fail(AUKIT_E_ARG, "null argument");
    const uint32_t W = (uint32_t)g->ctx.size();
    if (root >= W) return fail(AUKIT_E_ARG, "root %u out of range", root);
    const aukit_audio *first = nullptr;
    std::vector<uint64_t> lens;
    for (uint32_t r = 0; r < W; r++) {
        const aukit_audio *p = parts[r];
        if (!p) return fail(AUKIT_E_ARG, "part %u is null", r);
        if (p->pend_norm || p->lazy_rs) { int frc = audio_flush(g->ctx[r], p); if (frc) return frc; }   // deferred work (a normalize, a FLAC resample) is done before the rows travel
        if (!first) first = p;
        else if (p->channels != first->channels || p->rate != first->rate || p->dtype != first->dtype) return fail(AUKIT_E_ARG, "parts differ in channels / rate / dtype");
        lens.insert(lens.end(), p->len.begin(), p->len.end());
    }
    aukit_audio *o = *whole;
    AUKIT_HIP_CHECK(hipSetDevice(g->dev[root]));
    int rc = audio_prepare(g->ctx[root], &o, (uint32_t)lens.size(), first->channels, first->rate, first->dtype, lens.data());
    if (rc) return rc;
    *whole = o;
    const size_t esz = dtype_size(first->dtype);
    std::vector<GroupMsg> msgs;
    uint32_t at = 0;
    for (uint32_t r = 0; r < W; r++) {
        const aukit_audio *p = parts[r];
        // a part's rows lie back to back in the order and with the strides audio_prepare gives the same lengths: one block per part
        if (p->n) {
            const uint64_t dst0 = o->row_off[at];
            for (uint32_t s = 0; s < p->n; s++)
                if (o->row_off[at + s] - dst0 != p->row_off[s] - p->row_off[0] || o->row_stride[at + s] != p->row_stride[s]) return fail(AUKIT_E_HIP, "audio layouts differ (internal)");
            const uint64_t elems = p->total - p->row_off[0];
            msgs.push_back(GroupMsg{reinterpret_cast<const char *>(p->dev) + p->row_off[0] * esz, reinterpret_cast<char *>(o->dev) + dst0 * esz, (size_t)(elems * esz), r, root});
        }
        at += p->n;
    }
    return group_move(g, msgs, false, root);
}

// the same for byte results (Audio:dfpwm, the DFPWM transcode, aukit.pack)
int aukit_group_gather_batch(aukit_group *g, uint32_t root, aukit_batch *const *parts, aukit_batch **whole) {
    if (!g || !parts || !whole) return fail(AUKIT_E_ARG, "null argument");
    const uint32_t W = (uint32_t)g->ctx.size();
    if (root >= W) return fail(AUKIT_E_ARG, "root %u out of range", root);
    std::vector<uint64_t> off(1, 0);
    for (uint32_t r = 0; r < W; r++) {
        if (!parts[r]) return fail(AUKIT_E_ARG, "part %u is null", r);
        for (uint32_t s = 0; s < parts[r]->n; s++) off.push_back(off.back() + (parts[r]->off[s + 1] - parts[r]->off[s]));
    }
    AUKIT_HIP_CHECK(hipSetDevice(g->dev[root]));
    if (*whole) { aukit_batch_free(*whole); *whole = nullptr; }
    aukit_batch *b = new aukit_batch();
    b->n = (uint32_t)off.size() - 1; b->off = off; b->front_pad = 64; b->cap = (size_t)off.back() + 128; b->own = true;
    if (hipMalloc((void **)&b->base, b->cap) != hipSuccess) { delete b; return fail(AUKIT_E_NOMEM, "hipMalloc(%zu) failed", b->cap); }
    if (hipMalloc((void **)&b->d_off, off.size() * 8) != hipSuccess) { (void)hipFree(b->base); delete b; return fail(AUKIT_E_NOMEM, "hipMalloc failed"); }
    int rc = h2d_table(g->ctx[root], b->d_off, off.data(), off.size() * 8);
    if (rc) { aukit_batch_free(b); return rc; }
    b->version = 1;
    *whole = b;
    std::vector<GroupMsg> msgs;
    uint64_t at = 0;
    for (uint32_t r = 0; r < W; r++) {
        const uint64_t bytes = parts[r]->total();
        msgs.push_back(GroupMsg{parts[r]->data(), b->data() + at, (size_t)bytes, r, root});
        at += bytes;
    }
    return group_move(g, msgs, false, root);
}

}  // extern "C"

// runtime.hip — contexts, batches, audio objects, scratch tables, timers (host side of the C ABI).
#include <algorithm>
#include <mutex>
#include <set>
#include "common.h"

namespace aukit {

static thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

int DevBuf::ensure(size_t bytes) {
    if (bytes <= cap && p) return AUKIT_OK;
    size_t want = std::max<size_t>(bytes + bytes / 4, 4096);
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { p = nullptr; return fail(AUKIT_E_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); }
    cap = want;
    return AUKIT_OK;
}
void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

// Host -> device copy of a descriptor table on the context's stream that never blocks the host on the stream's earlier work.
int h2d_table(aukit_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!bytes) return AUKIT_OK;
    static const bool no_ring = getenv("AUKIT_NO_TABLE_RING") != nullptr;
    const size_t HALF = (size_t)16 << 20;
    if (!no_ring && bytes <= HALF) {   // (a table that fills most of a half just turns the ring over sooner)
        if (!ctx->tab_ring) {
            if (hipHostMalloc(reinterpret_cast<void **>(&ctx->tab_ring), 2 * HALF, hipHostMallocDefault) != hipSuccess) { ctx->tab_ring = nullptr; (void)hipGetLastError(); }
            else if (hipEventCreateWithFlags(&ctx->tab_ev[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->tab_ev[1], hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError(); (void)hipHostFree(ctx->tab_ring); ctx->tab_ring = nullptr;
            } else { ctx->tab_half = HALF; ctx->tab_head = 0; ctx->tab_cur = 0; ctx->tab_used[0] = true; ctx->tab_used[1] = false; }
        }
        if (ctx->tab_ring) {
            if (ctx->tab_head + bytes > ctx->tab_half) {   // this half is full: everything copied out of it so far is ahead of this event
                AUKIT_HIP_CHECK(hipEventRecord(ctx->tab_ev[ctx->tab_cur], ctx->stream));
                ctx->tab_cur ^= 1;
                ctx->tab_head = 0;
                if (ctx->tab_used[ctx->tab_cur]) AUKIT_HIP_CHECK(hipEventSynchronize(ctx->tab_ev[ctx->tab_cur]));
                ctx->tab_used[ctx->tab_cur] = true;
            }
            char *st = ctx->tab_ring + (size_t)ctx->tab_cur * ctx->tab_half + ctx->tab_head;
            memcpy(st, src, bytes);
            ctx->tab_head += (bytes + 63) & ~(size_t)63;
            AUKIT_HIP_CHECK(hipMemcpyAsync(dst, st, bytes, hipMemcpyHostToDevice, ctx->stream));
            return AUKIT_OK;
        }
    }
    AUKIT_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return AUKIT_OK;
}

int ctx_side_fork(aukit_ctx *ctx, hipStream_t *side) {
    if (!ctx->side_stream) {
        AUKIT_HIP_CHECK(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) AUKIT_HIP_CHECK(hipEventCreateWithFlags(&ctx->side_ev[i], hipEventDisableTiming));
    }
    AUKIT_HIP_CHECK(hipEventRecord(ctx->side_ev[0], ctx->stream));
    AUKIT_HIP_CHECK(hipStreamWaitEvent(ctx->side_stream, ctx->side_ev[0], 0));
    *side = ctx->side_stream;
    return AUKIT_OK;
}
int ctx_pre_stream(aukit_ctx *ctx, hipStream_t *s) {
    if (!ctx->pre_stream) {
        AUKIT_HIP_CHECK(hipStreamCreateWithFlags(&ctx->pre_stream, hipStreamNonBlocking));   // (a higher queue priority changes nothing: profiles/r06_flac_lookahead.txt)
        AUKIT_HIP_CHECK(hipEventCreateWithFlags(&ctx->pre_ev, hipEventDisableTiming));
        AUKIT_HIP_CHECK(hipEventCreateWithFlags(&ctx->entry_ev[0], hipEventDisableTiming));
        AUKIT_HIP_CHECK(hipEventCreateWithFlags(&ctx->entry_ev[1], hipEventDisableTiming));
        AUKIT_HIP_CHECK(hipEventCreateWithFlags(&ctx->scratch_ev, hipEventDisableTiming));
    }
    *s = ctx->pre_stream;
    return AUKIT_OK;
}
int ctx_side_join(aukit_ctx *ctx) {
    AUKIT_HIP_CHECK(hipEventRecord(ctx->side_ev[1], ctx->side_stream));
    AUKIT_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->side_ev[1], 0));
    return AUKIT_OK;
}

int upload_table(aukit_ctx *ctx, DevBuf &buf, const void *src, size_t bytes) {
    if (&buf == &ctx->seg_buf || &buf == &ctx->tile_buf) ctx->plan_key.clear();  // (plan_tiles_sized sets it again after its own uploads)
    int rc = buf.ensure(std::max<size_t>(bytes, 16));
    if (rc) return rc;
    return h2d_table(ctx, buf.p, src, bytes);
}

bool exact_div_verified(aukit_ctx *ctx, double d, uint64_t count) {
    auto it = ctx->div_ok.find(d);
    if (it != ctx->div_ok.end()) {
        if (it->second == 0) return false;          // known counter-example
        if (it->second >= count) return true;
    }
    const double r = 1.0 / d;
    uint64_t from = (it != ctx->div_ok.end()) ? it->second : 0;
    for (uint64_t n = from; n < count; n++) {
        double a = (double)n;
        double q0 = a * r;
        double e = std::fma(-d, q0, a);
        double q = std::fma(e, r, q0);
        if (q != a / d) { ctx->div_ok[d] = 0; return false; }
    }
    ctx->div_ok[d] = count;
    return true;
}

int ctx_begin_kernel(aukit_ctx *ctx) {
    if (ctx->ktiming && !ctx->ktiming_nested) AUKIT_HIP_CHECK(hipEventRecord(ctx->kev0, ctx->stream));
    return AUKIT_OK;
}
int ctx_end_kernel(aukit_ctx *ctx, const char *name, uint64_t algorithmic_bytes) {
    if (ctx->ktiming_nested) return AUKIT_OK;   // (the outer call reports: its name, its bytes, its interval)
    ctx->last_kernel = name ? name : "";
    ctx->last_bytes = algorithmic_bytes;
    ctx->timer_launches++;
    ctx->timer_bytes += algorithmic_bytes;
    if (ctx->ktiming) {
        AUKIT_HIP_CHECK(hipEventRecord(ctx->kev1, ctx->stream));
        AUKIT_HIP_CHECK(hipEventSynchronize(ctx->kev1));
        AUKIT_HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->kev0, ctx->kev1));
    }
    return AUKIT_OK;
}

int audio_rowmax_ensure(aukit_audio *a) {
    const size_t need = std::max<size_t>((size_t)a->n * a->channels, 1) * sizeof(uint64_t);
    if (need <= a->rowmax_cap && a->d_rowmax) return AUKIT_OK;
    if (a->d_rowmax) (void)hipFree(a->d_rowmax);
    a->d_rowmax = nullptr; a->rowmax_cap = 0; a->rowmax_valid = false;
    hipError_t e = hipMalloc((void **)&a->d_rowmax, need);
    if (e != hipSuccess) return fail(AUKIT_E_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
    a->rowmax_cap = need;
    return AUKIT_OK;
}

int audio_prepare(aukit_ctx *ctx, aukit_audio **out, uint32_t n, int channels, double rate, int dtype, const uint64_t *lens) {
    if (!out) return fail(AUKIT_E_ARG, "out is null");
    if (channels < 1) return fail(AUKIT_E_ARG, "channels out of range");
    aukit_audio *a = *out;
    if (!a) a = new aukit_audio();
    if (a->lazy_rs || a->lazy_rows.p) lazy_drop(ctx, a);   // an owed resample of the old contents is moot; its rows' buffer goes back to the context
    a->n = n; a->channels = channels; a->rate = rate; a->dtype = dtype;
    a->rowmax_valid = false;  // new contents are on their way
    a->pend_norm = false;
    a->len.assign(lens, lens + n);
    a->row_off.resize(n);
    a->row_stride.resize(n);
    uint64_t tot = 0;
    for (uint32_t s = 0; s < n; s++) {
        a->row_stride[s] = round_up(std::max<uint64_t>(lens[s], 1), 16);
        a->row_off[s] = tot;
        tot += a->row_stride[s] * (uint64_t)channels;
    }
    a->total = tot;
    size_t need = (size_t)tot * dtype_size(dtype) + 64;
    if (need > a->cap_bytes || !a->dev) {
        if (a->dev) (void)hipFree(a->dev);
        a->dev = nullptr; a->cap_bytes = 0;
        hipError_t e = hipMalloc(&a->dev, need);
        if (e != hipSuccess) { if (!*out) delete a; return fail(AUKIT_E_NOMEM, "hipMalloc(%zu) failed: %s", need, hipGetErrorString(e)); }
        a->cap_bytes = need;
    }
    size_t mbytes = (size_t)n * 3 * sizeof(uint64_t);
    if (mbytes > a->meta_cap || !a->d_meta) {
        if (a->d_meta) (void)hipFree(a->d_meta);
        a->d_meta = nullptr; a->meta_cap = 0;
        hipError_t e = hipMalloc((void **)&a->d_meta, std::max<size_t>(mbytes, 64));
        if (e != hipSuccess) { if (!*out) delete a; return fail(AUKIT_E_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e)); }
        a->meta_cap = std::max<size_t>(mbytes, 64);
    }
    if (n) {
        std::vector<uint64_t> m(3 * (size_t)n);
        std::copy(a->len.begin(), a->len.end(), m.begin());
        std::copy(a->row_off.begin(), a->row_off.end(), m.begin() + n);
        std::copy(a->row_stride.begin(), a->row_stride.end(), m.begin() + 2 * (size_t)n);
        { int hrc = h2d_table(ctx, a->d_meta, m.data(), mbytes); if (hrc) return hrc; }
    }
    a->version++;
    *out = a;
    return AUKIT_OK;
}

}  // namespace aukit

using namespace aukit;

extern "C" {

int aukit_abi_version(void) { return AUKIT_ABI_VERSION; }
const char *aukit_last_error(void) { return g_err; }

}  // extern "C"
namespace aukit {
static std::mutex g_live_mu;
static std::set<const aukit_ctx *> g_live;
static uint64_t g_next_ctx_id = 1;
// (the id is read under the registry's lock, after the address was found in it: a destroyed context is never dereferenced)
bool ctx_is_live(const aukit_ctx *c, uint64_t id) { std::lock_guard<std::mutex> lk(g_live_mu); return c && g_live.count(c) != 0 && c->id == id; }
// `ctx` is about to finish work that `owner` queued the inputs of (a decoder's rows behind a deferred resample, a filter pass behind a deferred
// normalize): when they are different contexts, ctx's stream waits for owner's — if owner still exists; an audio outlives its context legally
int owner_ready(aukit_ctx *ctx, aukit_ctx *owner, uint64_t owner_id) {
    if (!owner || (owner == ctx && ctx->id == owner_id) || !ctx_is_live(owner, owner_id)) return AUKIT_OK;
    AUKIT_HIP_CHECK(hipSetDevice(owner->device));
    AUKIT_HIP_CHECK(hipStreamSynchronize(owner->stream));   // (rare: a host wait is the simple, device-agnostic order)
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    return AUKIT_OK;
}
}  // namespace aukit
extern "C" {

int aukit_ctx_create(aukit_ctx **out, int device) {
    if (!out) return fail(AUKIT_E_ARG, "out is null");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return fail(AUKIT_E_HIP, "no HIP device available (%s): libaukit_hip has no CPU fallback", e != hipSuccess ? hipGetErrorString(e) : "0 devices");
    if (device < 0 || device >= count) return fail(AUKIT_E_ARG, "device %d out of range (0..%d)", device, count - 1);
    AUKIT_HIP_CHECK(hipSetDevice(device));
    aukit_ctx *c = new aukit_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    AUKIT_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    AUKIT_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
    AUKIT_HIP_CHECK(hipEventCreate(&c->ev0));
    AUKIT_HIP_CHECK(hipEventCreate(&c->ev1));
    AUKIT_HIP_CHECK(hipEventCreate(&c->kev0));
    AUKIT_HIP_CHECK(hipEventCreate(&c->kev1));
    { std::lock_guard<std::mutex> lk(g_live_mu); c->id = g_next_ctx_id++; g_live.insert(c); }
    *out = c;
    return AUKIT_OK;
}

void aukit_ctx_destroy(aukit_ctx *c) {
    if (!c) return;
    { std::lock_guard<std::mutex> lk(g_live_mu); g_live.erase(c); }
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    c->seg_buf.release(); c->tile_buf.release(); c->misc_buf.release(); c->tmp_buf.release(); c->tmp_buf2.release(); c->tmp_buf3.release(); c->wt_buf.release(); c->enc_state_buf.release(); c->dfx_lut.release(); c->dfx_gather.release(); if (c->dfx_sub_out) { aukit_batch_free(c->dfx_sub_out); c->dfx_sub_out = nullptr; }
    if (c->aux_stream) {
        (void)hipStreamSynchronize(c->aux_stream);
        if (c->dec_stream) { (void)hipStreamSynchronize(c->dec_stream); (void)hipStreamDestroy(c->dec_stream); }
        for (int i = 0; i < 10; i++) if (c->aux_ev[i]) (void)hipEventDestroy(c->aux_ev[i]);
        (void)hipStreamDestroy(c->aux_stream);
    }
    if (c->stream_full) { aukit_audio_free(c->stream_full); c->stream_full = nullptr; }
    delete c->spcm_ck;
    if (c->host_stage) (void)hipHostFree(c->host_stage);
    if (c->pre_stream) { (void)hipStreamSynchronize(c->pre_stream); (void)hipStreamDestroy(c->pre_stream); (void)hipEventDestroy(c->pre_ev); (void)hipEventDestroy(c->entry_ev[0]); (void)hipEventDestroy(c->entry_ev[1]); (void)hipEventDestroy(c->scratch_ev); c->pre_stream = nullptr; }
    c->flac_set[0].release(); c->flac_set[1].release(); c->scan_buf.release(); for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++) c->qoa_set[i][j].release();
    if (c->side_stream) { (void)hipStreamSynchronize(c->side_stream); (void)hipStreamDestroy(c->side_stream); (void)hipEventDestroy(c->side_ev[0]); (void)hipEventDestroy(c->side_ev[1]); }
    if (c->tab_ring) { (void)hipHostFree(c->tab_ring); (void)hipEventDestroy(c->tab_ev[0]); (void)hipEventDestroy(c->tab_ev[1]); }
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->kev0) (void)hipEventDestroy(c->kev0);
    if (c->kev1) (void)hipEventDestroy(c->kev1);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int aukit_ctx_set_stream(aukit_ctx *c, void *s) {
    if (!c) return fail(AUKIT_E_ARG, "ctx is null");
    AUKIT_HIP_CHECK(hipSetDevice(c->device));
    AUKIT_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)s;
    c->own_stream = false;
    return AUKIT_OK;
}
void *aukit_ctx_get_stream(aukit_ctx *c) { return c ? (void *)c->stream : nullptr; }
int aukit_ctx_sync(aukit_ctx *c) {
    if (!c) return fail(AUKIT_E_ARG, "ctx is null");
    AUKIT_HIP_CHECK(hipStreamSynchronize(c->stream));
    return AUKIT_OK;
}
int aukit_ctx_set_dtype(aukit_ctx *c, int dtype) {
    if (!c || (dtype != AUKIT_F64 && dtype != AUKIT_F32)) return fail(AUKIT_E_ARG, "dtype must be AUKIT_F64 or AUKIT_F32");
    c->dtype = dtype;
    return AUKIT_OK;
}
int aukit_ctx_set_sinc_window(aukit_ctx *c, int w) {
    if (!c || w < 1 || w > 30) return fail(AUKIT_E_ARG, "sinc window out of range");
    c->sinc_w = w;
    return AUKIT_OK;
}
int aukit_ctx_set_option(aukit_ctx *c, int option, int value) {
    if (!c) return fail(AUKIT_E_ARG, "ctx is null");
    if (option == AUKIT_OPT_EXACT_MATH) c->exact_math = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (option == AUKIT_OPT_STORE_X4) c->fast_store_x4 = value != 0;
    else if (option == AUKIT_OPT_COLLECT_STATS) c->collect_stats = value != 0;
    else if (option == AUKIT_OPT_DFPWM_SPECULATE) c->dfx_off = value == 0;
    else return fail(AUKIT_E_ARG, "unknown option %d", option);
    return AUKIT_OK;
}

int aukit_ctx_get_counter(aukit_ctx *c, int counter, uint64_t *value) {
    if (!c || !value) return fail(AUKIT_E_ARG, "null argument");
    if (counter < 0 || counter >= 8) return fail(AUKIT_E_ARG, "unknown counter %d", counter);
    *value = c->counters[counter];
    return AUKIT_OK;
}
int aukit_timer_begin(aukit_ctx *c) {
    if (!c) return fail(AUKIT_E_ARG, "ctx is null");
    AUKIT_HIP_CHECK(hipSetDevice(c->device));
    c->timer_launches = 0;
    c->timer_bytes = 0;
    AUKIT_HIP_CHECK(hipEventRecord(c->ev0, c->stream));
    return AUKIT_OK;
}
int aukit_timer_stats(aukit_ctx *c, uint64_t *launches, uint64_t *algorithmic_bytes) {
    if (!c) return fail(AUKIT_E_ARG, "ctx is null");
    if (launches) *launches = c->timer_launches;
    if (algorithmic_bytes) *algorithmic_bytes = c->timer_bytes;
    return AUKIT_OK;
}
int aukit_timer_end(aukit_ctx *c, float *ms) {
    if (!c || !ms) return fail(AUKIT_E_ARG, "null argument");
    AUKIT_HIP_CHECK(hipSetDevice(c->device));
    AUKIT_HIP_CHECK(hipEventRecord(c->ev1, c->stream));
    AUKIT_HIP_CHECK(hipEventSynchronize(c->ev1));
    AUKIT_HIP_CHECK(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return AUKIT_OK;
}
int aukit_ctx_set_kernel_timing(aukit_ctx *c, int enabled) {
    if (!c) return fail(AUKIT_E_ARG, "ctx is null");
    c->ktiming = enabled != 0;
    return AUKIT_OK;
}
int aukit_ctx_last_kernel(aukit_ctx *c, const char **name, float *ms, uint64_t *bytes) {
    if (!c) return fail(AUKIT_E_ARG, "ctx is null");
    if (name) *name = c->last_kernel.c_str();
    if (ms) *ms = c->last_ms;
    if (bytes) *bytes = c->last_bytes;
    return AUKIT_OK;
}

// ---------------------------------------------------------------- batches
static int batch_set_offsets(aukit_ctx *ctx, aukit_batch *b, const uint64_t *offsets, uint32_t n) {
    b->n = n;
    b->off.assign(offsets, offsets + n + 1);
    for (uint32_t i = 0; i < n; i++)
        if (b->off[i + 1] < b->off[i]) return fail(AUKIT_E_ARG, "offsets must be non-decreasing");
    if (b->d_off) (void)hipFree(b->d_off);
    b->d_off = nullptr;
    AUKIT_HIP_CHECK(hipMalloc((void **)&b->d_off, ((size_t)n + 1) * sizeof(uint64_t)));
    AUKIT_HIP_CHECK(hipMemcpyAsync(b->d_off, b->off.data(), ((size_t)n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    b->version++;
    return AUKIT_OK;
}

static void *host_stage(aukit_ctx *ctx, size_t bytes);

int aukit_batch_upload(aukit_ctx *ctx, aukit_batch **out, const uint8_t *bytes, const uint64_t *offsets, uint32_t n) {
    if (!ctx || !out || !offsets) return fail(AUKIT_E_ARG, "null argument");
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    aukit_batch *b = new aukit_batch();
    uint64_t total = offsets[n] - offsets[0];
    b->front_pad = 64;
    b->cap = (size_t)total + 128;
    b->own = true;
    hipError_t e = hipMalloc((void **)&b->base, b->cap);
    if (e != hipSuccess) { delete b; return fail(AUKIT_E_NOMEM, "hipMalloc(%zu) failed: %s", b->cap, hipGetErrorString(e)); }
    std::vector<uint64_t> rel(n + 1);
    for (uint32_t i = 0; i <= n; i++) rel[i] = offsets[i] - offsets[0];
    int rc = batch_set_offsets(ctx, b, rel.data(), n);
    if (rc) { aukit_batch_free(b); return rc; }
    if (total) {
        const void *srcp = bytes + offsets[0];
        if (void *st = total <= ((size_t)64 << 20) ? host_stage(ctx, (size_t)total) : nullptr) { memcpy(st, srcp, (size_t)total); srcp = st; }  // small uploads: through pinned memory
        hipError_t e2 = hipMemcpyAsync(b->data(), srcp, total, hipMemcpyHostToDevice, ctx->stream);
        if (e2 != hipSuccess) { aukit_batch_free(b); return fail(AUKIT_E_HIP, "hipMemcpyAsync failed: %s", hipGetErrorString(e2)); }
    }
    AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // the caller may free `bytes` right away
    *out = b;
    return AUKIT_OK;
}

int aukit_batch_wrap_device(aukit_ctx *ctx, aukit_batch **out, const void *dev_bytes, const uint64_t *offsets, uint32_t n) {
    if (!ctx || !out || !offsets) return fail(AUKIT_E_ARG, "null argument");
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));  // the offsets table below must land on the context's device
    aukit_batch *b = new aukit_batch();
    b->base = (uint8_t *)const_cast<void *>(dev_bytes);
    b->front_pad = 0;
    b->cap = (size_t)offsets[n];
    b->own = false;
    int rc = batch_set_offsets(ctx, b, offsets, n);
    if (rc) { aukit_batch_free(b); return rc; }
    // (what the caller queued on this stream before the wrap — the bytes' producer, the offsets' upload above — in front of readers on other streams)
    if (hipEventCreateWithFlags(&b->ready, hipEventDisableTiming) != hipSuccess || hipEventRecord(b->ready, ctx->stream) != hipSuccess) { aukit_batch_free(b); return fail(AUKIT_E_HIP, "hipEventRecord failed"); }
    *out = b;
    return AUKIT_OK;
}

int aukit_batch_info(const aukit_batch *b, uint32_t *n, uint64_t *total) {
    if (!b) return fail(AUKIT_E_ARG, "batch is null");
    if (n) *n = b->n;
    if (total) *total = b->total();
    return AUKIT_OK;
}
int aukit_batch_offsets(const aukit_batch *b, uint64_t *offsets) {
    if (!b || !offsets) return fail(AUKIT_E_ARG, "null argument");
    std::copy(b->off.begin(), b->off.end(), offsets);
    return AUKIT_OK;
}
const void *aukit_batch_device_ptr(const aukit_batch *b) { return b ? b->data() : nullptr; }
int aukit_batch_download(aukit_ctx *ctx, const aukit_batch *b, uint8_t *dst) {
    if (!ctx || !b || !dst) return fail(AUKIT_E_ARG, "null argument");
    void *st = b->total() && b->total() <= ((size_t)64 << 20) ? host_stage(ctx, (size_t)b->total()) : nullptr;
    if (b->total()) AUKIT_HIP_CHECK(hipMemcpyAsync(st ? st : dst, b->data(), b->total(), hipMemcpyDeviceToHost, ctx->stream));
    AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (st) memcpy(dst, st, (size_t)b->total());
    return AUKIT_OK;
}
void aukit_batch_free(aukit_batch *b) {
    if (!b) return;
    if (b->own && b->base) (void)hipFree(b->base);
    if (b->d_off) (void)hipFree(b->d_off);
    if (b->ready) (void)hipEventDestroy(b->ready);
    delete b;
}

// ---------------------------------------------------------------- audio objects
int aukit_audio_info(const aukit_audio *a, uint32_t *n, int *channels, double *rate, int *dtype, uint64_t *total) {
    if (!a) return fail(AUKIT_E_ARG, "audio is null");
    if (n) *n = a->n;
    if (channels) *channels = a->channels;
    if (rate) *rate = a->rate;
    if (dtype) *dtype = a->dtype;
    if (total) *total = a->total;
    return AUKIT_OK;
}
int aukit_audio_layout(const aukit_audio *a, uint64_t *lens, uint64_t *row_off, uint64_t *row_stride) {
    if (!a) return fail(AUKIT_E_ARG, "audio is null");
    if (lens) std::copy(a->len.begin(), a->len.end(), lens);
    if (row_off) std::copy(a->row_off.begin(), a->row_off.end(), row_off);
    if (row_stride) std::copy(a->row_stride.begin(), a->row_stride.end(), row_stride);
    return AUKIT_OK;
}
void *aukit_audio_device_ptr(const aukit_audio *a) {
    if (!a) return nullptr;
    if (a->pend_norm || a->lazy_rs) {   // the caller reads the samples: deferred work is done first, on the context that queued it — if that still exists
        aukit_ctx *owner = a->lazy_rs ? a->lazy_ctx : a->pend_ctx;
        if (!ctx_is_live(owner, a->lazy_rs ? a->lazy_ctx_id : a->pend_ctx_id)) { fail(AUKIT_E_ARG, "audio has deferred work and its context is gone: pass it through an entry point that takes a context first"); return nullptr; }
        if (audio_flush(owner, a)) return nullptr;
    }
    const_cast<aukit_audio *>(a)->rowmax_valid = false;              // ... and may write them
    return a->dev;
}

int aukit_audio_download_raw(aukit_ctx *ctx, const aukit_audio *a, void *dst) {
    if (!ctx || !a || !dst) return fail(AUKIT_E_ARG, "null argument");
    AUKIT_FLUSH(ctx, a);
    if (a->total) AUKIT_HIP_CHECK(hipMemcpyAsync(dst, a->dev, (size_t)a->total * dtype_size(a->dtype), hipMemcpyDeviceToHost, ctx->stream));
    AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return AUKIT_OK;
}

// pinned staging buffer of the context, grown on demand (up to 1 GiB; beyond that the caller's pageable memory is used directly)
static void *host_stage(aukit_ctx *ctx, size_t bytes) {
    if (bytes > ((size_t)1 << 30)) return nullptr;
    if (ctx->host_stage_cap < bytes) {
        if (ctx->host_stage) (void)hipHostFree(ctx->host_stage);
        ctx->host_stage = nullptr; ctx->host_stage_cap = 0;
        const size_t cap = std::max<size_t>(bytes + bytes / 4, (size_t)1 << 20);
        if (hipHostMalloc(&ctx->host_stage, cap, hipHostMallocDefault) != hipSuccess) { ctx->host_stage = nullptr; (void)hipGetLastError(); return nullptr; }
        ctx->host_stage_cap = cap;
    }
    return ctx->host_stage;
}

int aukit_audio_download(aukit_ctx *ctx, const aukit_audio *a, double *dst) {
    if (!ctx || !a || !dst) return fail(AUKIT_E_ARG, "null argument");
    const size_t bytes = (size_t)a->total * dtype_size(a->dtype);
    std::vector<unsigned char> pageable;
    const unsigned char *raw = static_cast<const unsigned char *>(host_stage(ctx, bytes + 8));
    if (!raw) { pageable.resize(bytes + 8); raw = pageable.data(); }
    int rc = aukit_audio_download_raw(ctx, a, const_cast<unsigned char *>(raw));
    if (rc) return rc;
    size_t w = 0;
    for (uint32_t s = 0; s < a->n; s++)
        for (int c = 0; c < a->channels; c++) {
            const size_t base = (size_t)a->row_off[s] + (size_t)c * a->row_stride[s];
            const uint64_t L = a->len[s];
            if (a->dtype == AUKIT_F64) memcpy(dst + w, reinterpret_cast<const double *>(raw) + base, (size_t)L * 8);
            else if (a->dtype == AUKIT_F32) { const float *p = reinterpret_cast<const float *>(raw) + base; for (uint64_t i = 0; i < L; i++) dst[w + i] = (double)p[i]; }
            else { const signed char *p = reinterpret_cast<const signed char *>(raw) + base; for (uint64_t i = 0; i < L; i++) dst[w + i] = (double)p[i]; }
            w += (size_t)L;
        }
    return AUKIT_OK;
}

int aukit_audio_upload(aukit_ctx *ctx, aukit_audio **out, const double *samples, const uint64_t *lens, uint32_t n, int channels,
                       double rate, int dtype) {
    if (!ctx || !out || !lens) return fail(AUKIT_E_ARG, "null argument");
    if (dtype != AUKIT_F64 && dtype != AUKIT_F32 && dtype != AUKIT_I8) return fail(AUKIT_E_ARG, "bad dtype");
    AUKIT_HIP_CHECK(hipSetDevice(ctx->device));
    aukit_audio *a = *out;
    int rc = audio_prepare(ctx, &a, n, channels, rate, dtype, lens);
    if (rc) return rc;
    std::vector<unsigned char> raw((size_t)a->total * dtype_size(dtype) + 8, 0);
    size_t r = 0;
    for (uint32_t s = 0; s < n; s++)
        for (int c = 0; c < channels; c++) {
            size_t base = (size_t)a->row_off[s] + (size_t)c * a->row_stride[s];
            for (uint64_t i = 0; i < lens[s]; i++) {
                double v = samples[r++];
                if (dtype == AUKIT_F64) reinterpret_cast<double *>(raw.data())[base + i] = v;
                else if (dtype == AUKIT_F32) reinterpret_cast<float *>(raw.data())[base + i] = (float)v;
                else reinterpret_cast<signed char *>(raw.data())[base + i] = (signed char)(int)v;
            }
        }
    if (a->total) AUKIT_HIP_CHECK(hipMemcpyAsync(a->dev, raw.data(), (size_t)a->total * dtype_size(dtype), hipMemcpyHostToDevice, ctx->stream));
    AUKIT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *out = a;
    return AUKIT_OK;
}

int aukit_audio_clone(aukit_ctx *ctx, const aukit_audio *a, aukit_audio **out) {
    if (!ctx || !a || !out) return fail(AUKIT_E_ARG, "null argument");
    AUKIT_FLUSH(ctx, a);
    aukit_audio *b = *out;
    int rc = audio_prepare(ctx, &b, a->n, a->channels, a->rate, a->dtype, a->len.data());
    if (rc) return rc;
    if (a->total) AUKIT_HIP_CHECK(hipMemcpyAsync(b->dev, a->dev, (size_t)a->total * dtype_size(a->dtype), hipMemcpyDeviceToDevice, ctx->stream));
    *out = b;
    return AUKIT_OK;
}

void aukit_audio_free(aukit_audio *a) {
    if (!a) return;
    if (a->dev) (void)hipFree(a->dev);
    if (a->d_meta) (void)hipFree(a->d_meta);
    if (a->d_rowmax) (void)hipFree(a->d_rowmax);
    a->lazy_rows.release();
    a->lazy_tab.release();
    delete a;
}

int aukit_chunks_info(const aukit_chunks *c, uint32_t *n, uint32_t *max_chunks) {
    if (!c) return fail(AUKIT_E_ARG, "chunks is null");
    if (n) *n = c->n;
    if (max_chunks) *max_chunks = c->max_chunks;
    return AUKIT_OK;
}
int aukit_chunks_get(const aukit_chunks *c, uint32_t *nchunks, uint32_t *lens, double *pos, int32_t *status, double *length_seconds) {
    if (!c) return fail(AUKIT_E_ARG, "chunks is null");
    if (nchunks) std::copy(c->nchunks.begin(), c->nchunks.end(), nchunks);
    if (lens) std::copy(c->lens.begin(), c->lens.end(), lens);
    if (pos) std::copy(c->pos.begin(), c->pos.end(), pos);
    if (status) std::copy(c->status.begin(), c->status.end(), status);
    if (length_seconds) std::copy(c->length_seconds.begin(), c->length_seconds.end(), length_seconds);
    return AUKIT_OK;
}
void aukit_chunks_free(aukit_chunks *c) { delete c; }

}  // extern "C"

namespace aukit { void *ctx_host_stage(aukit_ctx *ctx, size_t bytes) { return host_stage(ctx, bytes); } }

// common.h — host-side object model and device-side exact-arithmetic helpers of libaukit_hip.so.
// gfx950 (MI355X) only.  Compiled with -ffp-contract=off: every fused multiply-add in this library
// is an explicit __builtin_fma, because the reference's arithmetic (Lua doubles, aukit.lua) has none.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>
#include "../../include/aukit_hip.h"

namespace aukit {

// ---------------------------------------------------------------- errors
int fail(int code, const char *fmt, ...);
#define AUKIT_HIP_CHECK(expr)                                                                          \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) return ::aukit::fail(AUKIT_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// ---------------------------------------------------------------- device buffers
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes);  // grows (never shrinks); contents are NOT preserved
    void release();
};

}  // namespace aukit

// ---------------------------------------------------------------- opaque handle bodies
struct aukit_ctx {
    uint64_t id = 0;            // unique over the process's life (runtime.hip): what an audio remembers of the context it owes work to, beside its address
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int dtype = AUKIT_F64;
    int sinc_w = 10;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;    // user timer
    hipEvent_t kev0 = nullptr, kev1 = nullptr;  // per-kernel timing
    bool ktiming = false;
    int ktiming_nested = 0;     // > 0: an entry point running inside another one (dfpwm_spec.hip's hard streams): ctx_begin_kernel / ctx_end_kernel leave the outer call's events alone
    int exact_math = 0;         // AUKIT_OPT_EXACT_MATH: 1 = F32 storage is computed in fp64 (wave_f64.hip; reference-order kernels where that
                                //   one does not apply), 2 = always the reference-order kernels; 0 = f32 taps
    bool fast_store_x4 = true;  // AUKIT_OPT_STORE_X4: LDS-transposed 16-byte stores in the fast kernels
    std::string last_kernel;
    float last_ms = 0.f;
    uint64_t last_bytes = 0;
    uint64_t timer_launches = 0, timer_bytes = 0;  // since aukit_timer_begin: kernel launches and their algorithmic bytes
    int num_cus = 256;
    // scratch tables (segment/tile/stream descriptors); plan_key caches the last uploaded plan
    aukit::DevBuf seg_buf, tile_buf, misc_buf, tmp_buf, tmp_buf2, tmp_buf3;
    aukit::DevBuf enc_state_buf;  // DFPWM encoder states between the time slices of a transcode (dfpwm_par.hip)
    hipStream_t aux_stream = nullptr, dec_stream = nullptr;  // aux_stream: the sliced transcode's encoder runs here (dec_stream: unused, kept for the CU-mask experiment)
    hipEvent_t aux_ev[10] = {};
    hipStream_t side_stream = nullptr;   // independent kernels of one call run next to each other (FLAC's order classes): ctx_side_stream()
    // round 6: the FLAC loader's first stages (stream headers, the sync search) read nothing but the input batch: they run on a stream of their own, into one of
    // two alternating sets of tables, so that call k + 1's search overlaps what call k still has queued on ctx->stream (its filter and normalize passes);
    // the decoder waits for them by event.  AUKIT_FLAC_NO_LOOKAHEAD=1: everything on ctx->stream as before
    hipStream_t pre_stream = nullptr;
    hipEvent_t pre_ev = nullptr;
    aukit::DevBuf flac_set[2];
    int flac_par = 0;
    // round 6, late: the WHOLE decode of a FLAC call (search, decoder, chain) on the look-ahead stream — the call before's last passes (its normalize: no LDS,
    // HBM-bound) run beside the decoder (VALU-bound, the CU's whole LDS).  What orders the two streams: entry_ev[k & 1], recorded on ctx->stream when call k
    // begins (the look-ahead stream waits for the one of call k - 1 before it writes call k's table set: everything that read that set — call k - 2's — was
    // queued before it); scratch_ev, recorded on ctx->stream whenever a frame scratch goes back to tmp_buf3 (its last reader was queued just before); pre_ev
    // as before (ctx->stream behind the look-ahead stream)
    aukit::DevBuf qoa_set[2][3];   // stream.qoa's two walks on the look-ahead stream: their words ([0] walk records + fill table, [1] call records, [2] decode jobs), two sets alternating like flac_set
    int qoa_par = 0;
    aukit::DevBuf scan_buf;   // stream.adpcm's header scan (k_ima_scan_headers) on the look-ahead stream: its own words, not misc_buf (ctx->stream's kernels may still read that)
    hipEvent_t entry_ev[2] = {nullptr, nullptr};
    hipEvent_t scratch_ev = nullptr;
    bool scratch_ev_set = false;
    bool scratch_dirty = false;   // tmp_buf3 holds a call's frames and nothing tracks who reads them (the call's own gather / stream tail on ctx->stream): the next decoder waits for all of ctx->stream
    uint64_t flac_calls = 0;
    hipEvent_t side_ev[2] = {};
    int aux_enc_cus = -1;
    bool fused_attr_set = false;
    aukit::DevBuf dfx_lut;        // the 64 KiB stereo → mono mix table of the chunk-speculative transcoder (dfpwm_spec.hip), built once
    bool dfx_attr_set = false;
    bool dfx_off = false;         // AUKIT_OPT_DFPWM_SPECULATE = 0
    bool dfx_disable = false;     // set while that transcoder hands its hard streams to the older schedules (a nested aukit_dfpwm_transcode_mono)
    aukit::DevBuf dfx_gather;     // ... their bytes, gathered
    aukit_batch *dfx_sub_out = nullptr;  // ... and their result before it is scattered
    aukit_audio *stream_full = nullptr;  // stream.qoa with `mono`: the per-channel chunks before the mix, kept between calls (3.9 GB allocated and freed per call otherwise)  // k_df_fused's dynamic LDS size was announced on this context's device
    aukit::DevBuf fmt_flag;     // fast_fmt.hip: "a sample beyond ±1 was staged" (the launch condition of the reference-order redo)
    aukit::DevBuf wt_buf;       // phase-weight table of wave_f64.hip, cached per (b, interpolation)
    unsigned wt_b = 0, wt_doubles = 0;
    int wt_interp = -1;
    // pinned host staging for downloads / uploads of whole audios (a pageable copy runs at a fraction of the PCIe rate)
    void *host_stage = nullptr;
    size_t host_stage_cap = 0;
    // descriptor tables travel through a two-half pinned ring (upload_table): a pageable hipMemcpyAsync would block the host until the
    // previous kernel on the stream has finished, serialising the planner of call k + 1 with the kernel of call k
    char *tab_ring = nullptr;
    size_t tab_half = 0, tab_head = 0;
    int tab_cur = 0;
    bool tab_used[2] = {false, false};
    hipEvent_t tab_ev[2] = {};
    // a resumable stream handle (stream_handle.hip) that has dropped the bytes of delivered iterator calls decodes the REST of its stream: what it dropped
    // — bytes, 48 kHz outputs per channel — enters the chunk positions and the length of the stream factories that support it (stream.pcm / g711 /
    // adpcm / msadpcm), so that the rest's chunks carry the whole stream's numbers.  Zero everywhere else.
    int sb_dfpwm[6] = {0, 0, 0, 0, 0, 0};   // stream.dfpwm: the decoder's state (dfpwm_dev.h: n, strength, pb, lpf, pn) where the rest of the stream starts; sb_dfpwm_on: use it
    int sb_dfpwm2[6] = {0, 0, 0, 0, 0, 0};  // stream.mdfpwm: decoderR's (sb_dfpwm is decoderL's)
    int sb_dfpwm_n = 1;                      // states in use: 1 (stream.dfpwm), 2 (stream.mdfpwm: even pseudo-streams L, odd ones R)
    bool sb_dfpwm_on = false;
    double sb_pos = 0;   // stream.flac: the accumulated position (aukit.lua:3188) at the cut, exactly as the sums before it left it
    uint64_t sb_bytes = 0, sb_outputs = 0, sb_samples = 0;   // (sb_samples: decoded samples per channel in front — stream.qoa's file_pos)
    bool lazy_suppress = false;   // set while an owed resample is being materialised: audio_from_int_rows must not defer it again
    std::string plan_key;   // non-empty: seg_buf / tile_buf still hold the tables of plan_segs (any other upload into them clears it)
    std::vector<unsigned char> plan_segs;   // the segment descriptors of that plan, byte for byte
    int plan_tile_out = 0;
    unsigned plan_n_tiles = 0, plan_tiles_per_seg = 0;
    // verified range of the reciprocal-based exact division per ratio (see exact_div_verified)
    std::map<double, uint64_t> div_ok;
    // aukit.stream.pcm: the chunk plan of the last call (segments before their row offsets, per-stream lengths, the chunk table), reused when
    // the same batch comes back with the same descriptor — a 4096-stream plan is 1.5 ms of host work against a 2.1 ms kernel
    std::string spcm_key;
    std::vector<unsigned char> spcm_segs;
    std::vector<uint64_t> spcm_lens;
    uint64_t spcm_in_bytes = 0, spcm_out_elems = 0;
    struct aukit_chunks *spcm_ck = nullptr;
    // AUKIT_OPT_COLLECT_STATS: counters of the last call that has any (aukit_ctx_get_counter)
    bool collect_stats = false;
    uint64_t counters[8] = {};
};

struct aukit_batch {
    uint32_t n = 0;
    std::vector<uint64_t> off;  // n+1, relative to data()
    uint8_t *base = nullptr;    // allocation (owned) or wrapped pointer
    size_t front_pad = 0, cap = 0;
    bool own = true;
    uint64_t *d_off = nullptr;  // device copy of off (n+1)
    uint64_t version = 0;       // bumped whenever contents/layout change (plan cache key)
    hipEvent_t ready = nullptr; // wrapped batches: recorded on the wrapping context's stream when the batch was wrapped — what the caller queued there before (the bytes' producer, the offsets' upload) is ordered in front of a reader on another stream that waits for it (null: complete)
    uint8_t *data() const { return base + front_pad; }
    uint64_t total() const { return off.empty() ? 0 : off.back(); }
};

struct aukit_audio {
    uint32_t n = 0;
    int channels = 1;
    double rate = 48000;
    int dtype = AUKIT_F64;
    std::vector<uint64_t> len, row_off, row_stride;
    uint64_t total = 0;  // elements
    void *dev = nullptr;
    size_t cap_bytes = 0;
    uint64_t *d_meta = nullptr;  // device: len[n], row_off[n], row_stride[n]
    size_t meta_cap = 0;
    uint64_t version = 0;
    // max |x| of every row (bit pattern of a non-negative double), a by-product of the pass that last rewrote the rows (k_onepole):
    // effects.normalize then needs no pass of its own over the samples to find its peak (aukit.lua:3439-3444)
    uint64_t *d_rowmax = nullptr;
    size_t rowmax_cap = 0;
    bool rowmax_valid = false;
    // a deferred element-wise map: effects.normalize's `ch[i] = clamp(ch[i] * mult, -1, 1)` (:3455) has been asked for but not yet applied;
    // the next reader applies it on the fly (Audio:mono does: config 5's tail) or audio_flush() materialises it first
    bool pend_norm = false;
    double pend_peak = 1;
    int pend_independent = 0;
    aukit_ctx *pend_ctx = nullptr;
    uint64_t pend_ctx_id = 0;   // ... and that context's id: an address can come back with a later aukit_ctx_create
    // a deferred resample (flac_tail.hip): aukit_decode_resample on FLAC with F32 storage leaves the decoder's int32 rows here and the resample
    // owed; effects.highpass / lowpass pay it inside their own pass, anything else that reads the samples materialises it first (audio_flush)
    bool lazy_rs = false;
    // round 4, late: effects.highpass / lowpass on a STEREO audio whose resample is owed is owed too (1: low-pass, 2: high-pass, coefficient kept) —
    // if Audio:mono reads the audio next (BASELINE config 5: ... highpass -> normalize -> mono), resample + filter + channel mean run as one pass
    // that never writes the stereo rows (k_rs_onepole<..., 2 waves>); anything else resolves what is owed in order (lazy_resolve)
    int lazy_fx = 0;
    double lazy_fx_coef = 0;
    aukit::DevBuf lazy_rows;                             // taken out of the context's scratch; handed back when the resample is paid
    std::vector<uint64_t> lazy_row_off, lazy_row_len;    // per (stream, channel): element offset / samples in lazy_rows
    double lazy_rate = 0, lazy_full = 1;                 // lazy_full: int32 rows, `v / full`
    int lazy_src = 8;                                    // SrcKind of the rows: SRC_I32 (FLAC), SRC_I16 (IMA / MS-ADPCM / QOA), SRC_I8 (DFPWM) — round 4
    double lazy_norm_pos = 1, lazy_norm_neg = 1;         // `v / (v < 0 and norm_neg or norm_pos)` of the loader the rows came from
    int lazy_interp = 0;
    aukit_ctx *lazy_ctx = nullptr;
    uint64_t lazy_ctx_id = 0;
    // round 4: the rows may still lie frame by frame where the fused FLAC decoder left them (lazy_rows = its scratch) — lazy_tab then holds the
    // frame records in stream order, each stream's first record, each stream's block size and the (stream, channel) offsets of the contiguous
    // rows they would make; k_rs_onepole follows the records, every other consumer gathers the rows first (lazy_materialize)
    bool lazy_indirect = false;
    bool lazy_scratch16 = false;                         // ... as int16 (LazyFrames::scratch16)
    aukit::DevBuf lazy_tab;
    uint64_t lazy_nfr = 0, lazy_tot = 0;
    int lazy_min_bs = 0;                                 // the shortest full-size frame of any stream with more than one (k_rsp's window must fit two)
    size_t lazy_o_fbase = 0, lazy_o_bs0 = 0, lazy_o_rowoff = 0;
};

struct aukit_chunks {
    uint32_t n = 0, max_chunks = 0;
    std::vector<uint32_t> nchunks, lens;
    std::vector<double> pos, length_seconds;
    std::vector<int32_t> status;
    // stream.flac (fused decoder): the byte, relative to the stream's start, behind the last frame of every chunk, and where the first frame starts —
    // what a bounded reader-function handle cuts at (stream_handle.hip); empty where it is not known
    std::vector<uint64_t> in_end;
    std::vector<uint64_t> in_first;
};

namespace aukit {

static inline size_t dtype_size(int dt) { return dt == AUKIT_F64 ? 8 : (dt == AUKIT_F32 ? 4 : 1); }
static inline uint64_t round_up(uint64_t v, uint64_t m) { return (v + m - 1) / m * m; }

// (re)shape *out for n streams of the given lengths; reuses its buffers when they are large enough.
int audio_prepare(aukit_ctx *ctx, aukit_audio **out, uint32_t n, int channels, double rate, int dtype, const uint64_t *lens);
// time slices of the chunk-parallel DFPWM decoder (dfpwm_par.hip): after_slice(k, slices, fed_lo, fed_hi) is called on the host once the
// kernels of slice k are enqueued on ctx->stream — fed bytes [fed_lo, fed_hi) of every stream are then final in stream order
struct DfSliceHook {
    int slices = 1;
    std::function<int(unsigned, unsigned, unsigned long long, unsigned long long)> after_slice;
};
// contexts that exist (runtime.hip): an audio remembers the context its deferred work was queued on (pend_ctx, lazy_ctx) and must not touch it once
// it is gone; owner_ready() makes `ctx` (the context about to pay the work) wait for what `owner` has queued when the two differ
bool ctx_is_live(const aukit_ctx *c, uint64_t id);   // the context at this address is still the one with this id (ids are never reused)
int owner_ready(aukit_ctx *ctx, aukit_ctx *owner, uint64_t owner_id);
// applies a deferred map (aukit_audio::pend_norm) in place; every entry point that reads an audio's samples calls it first
int audio_flush(aukit_ctx *ctx, const aukit_audio *a);
#define AUKIT_FLUSH(ctx, a)                                                   \
    do {                                                                      \
        if ((a) && ((a)->pend_norm || (a)->lazy_rs)) { int _frc = ::aukit::audio_flush((ctx), (a)); if (_frc) return _frc; } \
    } while (0)
// the deferred resample of flac_tail.hip
void lazy_drop(aukit_ctx *ctx, aukit_audio *a);
int lazy_materialize(aukit_ctx *ctx, aukit_audio *a);
struct LazyFrames {   // the fused FLAC decoder's frames (flac_dev.h), for a deferred resample that reads them in place
    const void *d_frames; uint64_t nfr; const unsigned long long *d_fbase, *d_rowoff; const std::vector<int> *bs0; bool uniform; uint64_t tot_elems;
    const std::vector<unsigned> *nframes;
    bool scratch16 = false;   // the frames hold int16 finals (k_flac_decode<..., O16>): [channel 0][channel 1] from twice the record's offset
};
bool lazy_resample_try(aukit_ctx *ctx, const std::vector<uint64_t> &row_off, const std::vector<uint64_t> &row_len, uint32_t n, int C, double rate, double new_rate, int interp,
                       double full, aukit_audio **out, int *rc, const LazyFrames *frames = nullptr, int src_kind = 8 /* SRC_I32 */, double norm_pos = 0, double norm_neg = 0);
bool lazy_onepole_try(aukit_ctx *ctx, aukit_audio *a, double coef, bool highpass, int *rc, aukit_audio *mono_out = nullptr);
int lazy_resolve(aukit_ctx *ctx, aukit_audio *a);   // everything owed on the rows themselves (resample, then a deferred filter), effects.hip
// the frames one full iterator call of aukit.stream.pcm moves its table on by (K of SURVEY Q1: aukit.lua:2417-2419), api_resample.hip
long stream_pcm_call_frames(double sample_rate, int interp);
int mdfpwm_state_after(aukit_ctx *ctx, const unsigned char *dev_payload, uint64_t calls, const int *in12, bool in_on, int *out12);   // codecs2.hip
int dfpwm_state_after(aukit_ctx *ctx, const unsigned char *dev_bytes, uint64_t calls, uint64_t adv, const int *in6, bool in_on, int *out6);   // codecs2.hip
int audio_rowmax_ensure(aukit_audio *a);  // allocates a->d_rowmax for n × channels rows
// the context's pinned host staging buffer, grown to `bytes` (nullptr beyond 1 GiB or when pinning fails: use pageable memory then); one user at a time
void *ctx_host_stage(aukit_ctx *ctx, size_t bytes);
int ctx_begin_kernel(aukit_ctx *ctx);
int ctx_end_kernel(aukit_ctx *ctx, const char *name, uint64_t algorithmic_bytes);
// uploads a host table into a ctx scratch buffer on the ctx stream
int upload_table(aukit_ctx *ctx, DevBuf &buf, const void *src, size_t bytes);
int h2d_table(aukit_ctx *ctx, void *dst, const void *src, size_t bytes);
int ctx_side_fork(aukit_ctx *ctx, hipStream_t *side);  // side stream that starts after everything queued on ctx->stream so far (runtime.hip)
int ctx_side_join(aukit_ctx *ctx);
int ctx_pre_stream(aukit_ctx *ctx, hipStream_t *s);   // the look-ahead stream of the FLAC loader's first stages (created on first use)                     // ctx->stream continues after everything queued on the side stream  // pinned-ring H2D on ctx->stream (runtime.hip)
// q = RN(n / d) computed as fma(fma(-d, n*r, n), r, n*r) with r = RN(1/d) is exact for the integers
// n in [0, count): verified on the host once per (d, count) and cached in ctx->div_ok.
bool exact_div_verified(aukit_ctx *ctx, double d, uint64_t count);

// ---------------------------------------------------------------- device helpers
#define AUKIT_DEV __device__ __forceinline__

// Lua clamp(n, min, max)  aukit.lua:228-232 (NaN falls through unchanged, like the Lua)
AUKIT_DEV double lua_clamp(double n, double mn, double mx) { return n < mn ? mn : (n > mx ? mx : n); }

// correctly rounded a / b from r = RN(1/b): Markstein's correction step (two fmas)
AUKIT_DEV double div_rcp(double a, double b, double r) {
    double q0 = a * r;
    double e = __builtin_fma(-b, q0, a);
    return __builtin_fma(e, r, q0);
}

// correctly rounded fx^3 (what an exact pow(fx, 3) returns): double-double product
AUKIT_DEV double pow3_rn(double fx) {
    double hi = fx * fx;
    double lo = __builtin_fma(fx, fx, -hi);
    double p = hi * fx;
    double pl = __builtin_fma(hi, fx, -p);
    double t = __builtin_fma(lo, fx, pl);
    return p + t;
}

// interpolate.cubic  aukit.lua:261-266 — same operation order, no contraction
AUKIT_DEV double cubic_exact(double p0, double p1, double p2, double p3, double fx) {
    double f2 = fx * fx;       // fx^2: pow(fx, 2) is exactly RN(fx*fx)
    double f3 = pow3_rn(fx);   // fx^3
    double c3 = -0.5 * p0 + 1.5 * p1 - 1.5 * p2 + 0.5 * p3;
    double c2 = p0 - 2.5 * p1 + 2 * p2 - 0.5 * p3;
    double c1 = -0.5 * p0 + 0.5 * p2;
    return c3 * f3 + c2 * f2 + c1 * fx + p1;
}
// interpolate.linear  aukit.lua:257-260
AUKIT_DEV double linear_exact(double a, double b, double fx) { return a + (b - a) * fx; }

}  // namespace aukit

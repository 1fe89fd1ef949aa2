/* aukit_hip.h — C ABI of libaukit_hip.so, the MI355X (gfx950) implementation of AUKit's batched
 * decode → resample-to-48 kHz → effects/mixdown → DFPWM re-encode hot path.
 *
 * The reference (MCJack123/AUKit 1.10.0, /root/reference/aukit.lua) is a pure-Lua module with no
 * FFI/plugin interface: its boundary is the table returned by `require "aukit"` (aukit.lua:97-113,
 * :3620).  This header is the native face a Lua host binds instead (LuaJIT `ffi.cdef` of this file —
 * see INTEGRATION.md and aukit_amd/lua/aukit.lua); each entry point cites the reference function(s)
 * whose per-sample loops it replaces.  Plain C: opaque handles, pointers and sizes only.
 *
 * Conventions
 *   - every function returns 0 (AUKIT_OK) or a negative aukit_status; aukit_last_error() returns a
 *     thread-local message.  AUKIT_E_LUA means "the reference raises a Lua error here"; the message
 *     is the reference's own string where it has one, so a Lua shim can `error(msg, 2)`.
 *   - a *batch* is N independent byte strings (the `data` argument of the reference functions, N times);
 *     an *audio* is N independent aukit.Audio objects with a common channel count and sample rate.
 *     Streams never interact (no reference function reads another stream), so a batch shards across
 *     GPUs by stream index with no collective.
 *   - one aukit_ctx = one GPU + one HIP stream; use one ctx per thread (the reference is single-threaded).
 *     All work is enqueued on the ctx stream; download/sync calls wait for it.
 *   - sample storage (`aukit_dtype`): AUKIT_F64 mirrors the reference's Lua doubles exactly;
 *     AUKIT_F32 stores fp64 arithmetic results as float (1e-6 RMS tolerance);
 *     AUKIT_I8 is used for the integer-valued stream outputs of stream.{g711,adpcm,msadpcm,mdfpwm}.
 *   - layout of an audio on the device: row (stream s, channel c) starts at element
 *     row_off[s] + c * row_stride[s], rows are padded to a multiple of 16 elements.
 */
#ifndef AUKIT_HIP_H
#define AUKIT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AUKIT_ABI_VERSION 2   /* 2: aukit_stream_next takes the capacity of dst; aukit_codec_desc carries AUKIT_MAX_PLANAR_CHANNELS predictors / step indices */
#define AUKIT_MAX_CHANNELS 8          /* FLAC (its format's own limit), the WAV IMA splitter and MS-ADPCM take the reference's 1 or 2 */
#define AUKIT_MAX_PLANAR_CHANNELS 64  /* PCM, G.711, QOA, IMA ADPCM and DFPWM (loaders and streams), the Audio methods and effects: planar rows of any count up to this
                                         (the reference takes any channel count, aukit.lua:1049-1171, :2228; round 4) */

typedef struct aukit_ctx aukit_ctx;
typedef struct aukit_batch aukit_batch;
typedef struct aukit_audio aukit_audio;

typedef enum { AUKIT_OK = 0, AUKIT_E_ARG = -1, AUKIT_E_LUA = -2, AUKIT_E_NOMEM = -3, AUKIT_E_UNSUPPORTED = -4, AUKIT_E_HIP = -5 } aukit_status;
typedef enum { AUKIT_F64 = 0, AUKIT_F32 = 1, AUKIT_I8 = 2 } aukit_dtype;
/* aukit.defaultInterpolation / the `interpolation` argument (aukit.lua:96-99, :253-284) */
typedef enum { AUKIT_INTERP_NONE = 0, AUKIT_INTERP_LINEAR = 1, AUKIT_INTERP_CUBIC = 2, AUKIT_INTERP_SINC = 3 } aukit_interp;
typedef enum { AUKIT_SIGNED = 0, AUKIT_UNSIGNED = 1, AUKIT_FLOAT = 2 } aukit_pcm_type;

typedef enum {
    AUKIT_CODEC_PCM = 0,     /* aukit.pcm / stream.pcm        aukit.lua:1049, :2228 */
    AUKIT_CODEC_G711 = 1,    /* aukit.g711 / stream.g711      aukit.lua:1361, :2850 */
    AUKIT_CODEC_ADPCM = 2,   /* aukit.adpcm (raw nibbles)     aukit.lua:1183        */
    AUKIT_CODEC_ADPCM_WAV = 3, /* IMA blocks: aukit.wav splitter :1509-1548 / stream.adpcm :2753 */
    AUKIT_CODEC_MSADPCM = 4, /* aukit.msadpcm / stream.msadpcm aukit.lua:1283, :2588 */
    AUKIT_CODEC_DFPWM = 5,   /* aukit.dfpwm / stream.dfpwm    aukit.lua:1392, :2439 */
    AUKIT_CODEC_MDFPWM = 6,  /* aukit.mdfpwm / stream.mdfpwm  aukit.lua:1420, :2507 */
    AUKIT_CODEC_QOA = 7,     /* aukit.qoa / stream.qoa        aukit.lua:1706, :3202 */
    AUKIT_CODEC_FLAC = 8     /* aukit.flac / stream.flac      aukit.lua:311-619, :1657, :3124 */
} aukit_codec;

/* Arguments of the reference loader / stream factory for one codec (unused fields are ignored). */
typedef struct {
    int32_t codec;         /* aukit_codec */
    int32_t channels;      /* `channels` (PCM, G711, ADPCM*, MSADPCM, DFPWM)                */
    double sample_rate;    /* `sampleRate`                                                   */
    int32_t bit_depth;     /* PCM: 8/16/24/32                                                */
    int32_t data_type;     /* PCM: aukit_pcm_type                                            */
    int32_t big_endian;    /* PCM: `bigEndian`                                               */
    int32_t interleaved;   /* PCM/ADPCM: `interleaved` (default true in the reference)       */
    int32_t ulaw;          /* G711: `ulaw`                                                   */
    int32_t top_first;     /* ADPCM: `topFirst`                                              */
    int32_t block_align;   /* ADPCM_WAV / MSADPCM: `blockAlign`                              */
    int32_t ncoef;         /* MSADPCM: number of coefficient pairs (0 → the 7 defaults :1304) */
    int16_t coef1[32];     /* MSADPCM `coefficients[1]`                                      */
    int16_t coef2[32];     /* MSADPCM `coefficients[2]`                                      */
    int32_t predictor[AUKIT_MAX_PLANAR_CHANNELS];  /* ADPCM: initial predictor(s)            */
    int32_t step_index[AUKIT_MAX_PLANAR_CHANNELS]; /* ADPCM: initial step index(es)          */
} aukit_codec_desc;

/* ---- container front-ends (host-side byte parsing, no GPU): aukit.wav / aukit.aiff / aukit.au (aukit.lua:1456-1651) and the header
 * half of aukit.stream.wav / .aiff / .au (:2927-3113).  The walk reproduces what a caller of the Lua can observe: the `string.unpack`
 * error on a truncated chunk header, aukit.wav walking on behind a `data` chunk, WAVE_FORMAT_EXTENSIBLE GUIDs (:131-139), the AIFF
 * 80-bit rate (:1603-1605), `sowt` little-endian for aukit.aiff but big-endian for stream.aiff (:1613 / :3065), AU's 0-based offset used
 * as a 1-based index (:1643).  The result is the descriptor to hand to aukit_decode / aukit_stream_decode with bytes
 * [payload_off, payload_off + payload_len) of the file. */
typedef enum { AUKIT_CONTAINER_WAV = 0, AUKIT_CONTAINER_AIFF = 1, AUKIT_CONTAINER_AU = 2 } aukit_container_kind;
/* `dataType` of a WAV file (Audio.info.dataType, aukit.lua:1474-1504) */
typedef enum { AUKIT_WAVDT_SIGNED = 0, AUKIT_WAVDT_UNSIGNED = 1, AUKIT_WAVDT_FLOAT = 2, AUKIT_WAVDT_ALAW = 3, AUKIT_WAVDT_ULAW = 4, AUKIT_WAVDT_ADPCM = 5,
               AUKIT_WAVDT_MSADPCM = 6, AUKIT_WAVDT_DFPWM = 7 } aukit_wav_data_type;
typedef struct {
    aukit_codec_desc desc;   /* loader / stream factory arguments for the payload                                             */
    uint64_t payload_off;    /* first payload byte (0-based) and                                                              */
    uint64_t payload_len;    /* byte count: what the Lua's str_sub hands to the loader                                        */
    int32_t wav_data_type;   /* WAV: aukit_wav_data_type (Audio.info.dataType)                                                */
    int32_t bit_depth;       /* WAV: Audio.info.bitDepth; AIFF: COMM bit depth; AU: bits of the encoding                      */
    double length_seconds;   /* stream.*: the factory's second return value where it computes one itself (:2994-2996, :3064-3069,
                                :3107-3113), NaN where the codec's own stream factory supplies it                              */
} aukit_container;
/* stream = 0: aukit.wav / .aiff / .au; 1: the header walk of aukit.stream.wav / .aiff / .au on a string; 2: on the first piece a reader
 * function returned ("the first chunk MUST contain the ENTIRE header", :2918): the payload is what that piece holds of it */
int aukit_parse_container(const uint8_t *bytes, uint64_t n, int kind, int stream, aukit_container *out);

/* ids for aukit_effect(); args in the reference's argument order after `audio` (aukit.lua:3356-3618) */
typedef enum {
    AUKIT_FX_AMPLIFY = 0,   /* (multiplier)                                      :3356 */
    AUKIT_FX_SPEED = 1,     /* (multiplier, default_interp)                      :3376 */
    AUKIT_FX_FADE = 2,      /* (startTime, startAmplitude, endTime, endAmplitude) :3394 */
    AUKIT_FX_INVERT = 3,    /* ()                                                :3417 */
    AUKIT_FX_NORMALIZE = 4, /* (peakAmplitude=1, independent=0)                  :3431 */
    AUKIT_FX_CENTER = 5,    /* ()                                                :3464 */
    AUKIT_FX_TRIM = 6,      /* (threshold) — always AUKIT_E_LUA like the reference (:3495) */
    AUKIT_FX_DELAY = 7,     /* (delay, multiplier=0.5)                           :3505 */
    AUKIT_FX_ECHO = 8,      /* (delay=1, multiplier=0.5)                         :3524 */
    AUKIT_FX_REVERB = 9,    /* (delay=100, decay=0.3, wet=1, dry=0)              :3546 */
    AUKIT_FX_LOWPASS = 10,  /* (frequency)                                       :3586 */
    AUKIT_FX_HIGHPASS = 11  /* (frequency)                                       :3604 */
} aukit_effect_id;

/* ---- library / context ---- */
int aukit_abi_version(void);
const char *aukit_last_error(void);
int aukit_ctx_create(aukit_ctx **out, int device);
void aukit_ctx_destroy(aukit_ctx *ctx);
/* enqueue on an existing hipStream_t (e.g. torch's current stream) instead of the ctx's own */
int aukit_ctx_set_stream(aukit_ctx *ctx, void *hip_stream);
void *aukit_ctx_get_stream(aukit_ctx *ctx);
int aukit_ctx_sync(aukit_ctx *ctx);
/* default storage type of float results (AUKIT_F64 on creation) */
int aukit_ctx_set_dtype(aukit_ctx *ctx, int dtype);
/* sincWindowSize (aukit.lua:129): 10, or 30 to mirror LuaJIT hosts */
int aukit_ctx_set_sinc_window(aukit_ctx *ctx, int w);
/* tuning / fidelity switches */
typedef enum {
    AUKIT_OPT_EXACT_MATH = 0, /* arithmetic behind AUKIT_F32 storage.  0 (default): f32 FMA taps on exact rational positions, ≤ 1e-6 RMS
                                 from the reference.  1: fp64 arithmetic, rounded to f32 once when stored (aukit.lua computes in
                                 doubles, :261-266, :662-669) — fp64 values throughout, NOT the reference's operation order: the phase-weight
                                 kernel (16-bit mono PCM, stream.pcm on it), the fp64 coefficient kernel (G.711 mono), fp64 tables with a
                                 Horner form on the exact rational position (every other interleaved PCM format of one or two channels,
                                 G.711 stereo); anything else runs the reference-order kernels.  Results may differ from value 2 by one
                                 f32 ulp.  2: always the reference-order fp64 kernels (stream tails included). */
    AUKIT_OPT_STORE_X4 = 1,   /* 1 (default): fast kernels transpose results through LDS and store 16 B per lane */
    AUKIT_OPT_COLLECT_STATS = 2, /* 1: calls that have counters (aukit_ctx_get_counter) read them back — one more device→host sync per call.
                                   0 (default): they do not. */
    AUKIT_OPT_DFPWM_SPECULATE = 3 /* 1 (default): aukit_dfpwm_transcode_mono cuts every stream into time chunks that are decoded, mixed and ENCODED
                                   by a lane each from a guessed encoder state, verified afterwards (dfpwm_spec.hip) — several times faster
                                   on signal (silence a stream STARTS with included), same bytes always; a few short streams (up to 16 of
                                   up to 40 stream-seconds together) stay with the exact parallel encoder either way; up to ~1.6x slower than
                                   0 where the guesses fail and the probe does not notice (noise-like streams; silence INSIDE the streams of a
                                   batch too large for a second round is noticed and declined).  0: one encoder lane per stream behind the
                                   chunk-parallel decoder. */
} aukit_option;
int aukit_ctx_set_option(aukit_ctx *ctx, int option, int value);
/* counters of the most recent call that produced them (AUKIT_OPT_COLLECT_STATS = 1) */
typedef enum {
    AUKIT_COUNTER_DFPWM_CHUNKS = 0,        /* chunks the parallel DFPWM decoder cut the batch into (aukit.dfpwm, stream.dfpwm, the transcode) */
    AUKIT_COUNTER_DFPWM_CHUNKS_REDONE = 1, /* of those, the ones whose warmed-up start state differed from the true one and were decoded again */
    AUKIT_COUNTER_TIER1_ERR_NANO = 3,      /* the bit-exact floor()ed stream paths (stream.g711, stream.adpcm, stream.msadpcm) answer most outputs from an f32 evaluation
                                              that is only taken when it lies further from an integer than a guard (5e-4 / 6e-4) set at twice its derived error
                                              bound.  With COLLECT_STATS the call runs an audited instantiation: the largest |f32 value - fp64 value| it saw, in 1e-9 */
    AUKIT_COUNTER_TIER1_OUTPUTS = 4,       /* ... and how many outputs it compared */
    AUKIT_COUNTER_DFPWM_RESPECULATED = 5,  /* the chunk-speculative transcoder / encoder (dfpwm_spec.hip): how many times a stream's remaining chunks were speculated
                                              again because its true encoder had changed its class (a clamp of the strength on the way) */
    AUKIT_COUNTER_RECURRENCE_F32 = 7,      /* the most recent one-pole filter launch (effects.lowpass / highpass with an owed resample, the stream.qoa / stream.flac tails:
                                              k_rs_onepole) ran its recurrence and scan in f32 (1) or in fp64 (0) — set with or without AUKIT_OPT_COLLECT_STATS */
    AUKIT_COUNTER_DFPWM_HARD = 6,          /* ... and how many streams it gave up on and left to the schedule with one encoder lane per stream (noise-like input) */
    AUKIT_COUNTER_FLAC_FUSED = 2           /* 1: the most recent FLAC decode was served by the fused decoder (flac_fused.hip); 0: a frame it declines was on
                                              the chain (or the batch is deeper than 24 bits) and the two-kernel decoder ran.  Set without COLLECT_STATS. */
} aukit_counter;
int aukit_ctx_get_counter(aukit_ctx *ctx, int counter, uint64_t *value);
/* hipEvent pair on the ctx stream: begin(); ...launches...; end() → elapsed milliseconds */
int aukit_timer_begin(aukit_ctx *ctx);
int aukit_timer_end(aukit_ctx *ctx, float *ms);
/* kernel launches since aukit_timer_begin and the sum of their algorithmic bytes (input read once + output written once) */
int aukit_timer_stats(aukit_ctx *ctx, uint64_t *launches, uint64_t *algorithmic_bytes);
/* name and duration (ms, hipEvents around the launch) of the most recent kernel launched through
 * aukit_decode_resample / aukit_stream_decode when profiling is enabled */
int aukit_ctx_set_kernel_timing(aukit_ctx *ctx, int enabled);
int aukit_ctx_last_kernel(aukit_ctx *ctx, const char **name, float *ms, uint64_t *algorithmic_bytes);

/* ---- batches of byte strings ---- */
int aukit_batch_upload(aukit_ctx *ctx, aukit_batch **out, const uint8_t *bytes, const uint64_t *offsets /* n+1 */, uint32_t n);
/* zero-copy: bytes already on this device (e.g. received by RCCL); the caller keeps ownership.  Ordering: whatever was queued on the context's
 * stream BEFORE this call (the bytes' producer, if it ran there) is in front of every reader of the batch — also of the FLAC loader's decode, of stream.adpcm's header scan and of stream.qoa's frame walks,
 * which since round 6 read it on a stream of the library's own (an event recorded here orders them).  The bytes must be complete, or on their way on
 * that stream, when the batch is wrapped, and must not change while the batch is in use: wrap again after rewriting them. */
int aukit_batch_wrap_device(aukit_ctx *ctx, aukit_batch **out, const void *dev_bytes, const uint64_t *offsets /* host, n+1 */, uint32_t n);
int aukit_batch_info(const aukit_batch *b, uint32_t *n, uint64_t *total_bytes);
int aukit_batch_offsets(const aukit_batch *b, uint64_t *offsets /* n+1 */);
const void *aukit_batch_device_ptr(const aukit_batch *b);
int aukit_batch_download(aukit_ctx *ctx, const aukit_batch *b, uint8_t *dst);
void aukit_batch_free(aukit_batch *b);

/* ---- audio objects ---- */
int aukit_audio_upload(aukit_ctx *ctx, aukit_audio **out, const double *samples /* packed [s][c][len_s] */,
                       const uint64_t *lens /* n */, uint32_t n, int channels, double sample_rate, int dtype);
int aukit_audio_info(const aukit_audio *a, uint32_t *n, int *channels, double *sample_rate, int *dtype, uint64_t *total_elems);
int aukit_audio_layout(const aukit_audio *a, uint64_t *lens /* n */, uint64_t *row_off /* n */, uint64_t *row_stride /* n */);
void *aukit_audio_device_ptr(const aukit_audio *a);
/* packed [s][c][len_s] as doubles (exact for every dtype) */
int aukit_audio_download(aukit_ctx *ctx, const aukit_audio *a, double *dst);
/* raw padded device buffer in its own dtype (total_elems elements) */
int aukit_audio_download_raw(aukit_ctx *ctx, const aukit_audio *a, void *dst);
int aukit_audio_clone(aukit_ctx *ctx, const aukit_audio *a, aukit_audio **out);
void aukit_audio_free(aukit_audio *a);

/* ---- loaders: aukit.<codec>(data, ...) → Audio   (aukit.lua:1049-1777) ----
 * `*out` may point to an audio returned by an earlier identical call: its buffers are reused. */
int aukit_decode(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *desc, int dtype, aukit_audio **out);
/* fused aukit.<codec>(data, ...):resample(new_rate, interp): decoded samples never touch HBM for
 * PCM and G.711; block codecs decode to a compact integer intermediate first.  Same results as
 * aukit_decode followed by aukit_resample. */
int aukit_decode_resample(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *desc, double new_rate, int interp,
                          int dtype, aukit_audio **out);

/* ---- Audio methods ---- */
int aukit_resample(aukit_ctx *ctx, const aukit_audio *in, double new_rate, int interp, aukit_audio **out); /* Audio:resample :653 */
int aukit_mono(aukit_ctx *ctx, const aukit_audio *in, aukit_audio **out);                                  /* Audio:mono :677 */
/* Audio:mix(amplifier, ...) :804 — audios[0] is `self`; all at the same sample rate (resample first) */
int aukit_mix(aukit_ctx *ctx, const aukit_audio *const *audios, int count, double amplifier, aukit_audio **out);
/* aukit.effects.<name>(audio, ...) in place :3356-3618 */
int aukit_effect(aukit_ctx *ctx, aukit_audio *inout, int effect_id, const double *args, int nargs);
/* Audio:dfpwm(interleaved) :1005-1018 → one byte string per stream.  Every stream is cut into time chunks that a lane each encodes from a guessed encoder
 * state; a verify pass keeps only what follows from the true state and the rest is encoded again (dfpwm_spec.hip): the reference's bytes, whatever the
 * batch size.  Raises (AUKIT_E_LUA) for an amplitude outside -128..127 like the cc.audio.dfpwm encoder.  Synchronises with the host before it returns. */
int aukit_dfpwm_encode(aukit_ctx *ctx, const aukit_audio *in, int interleaved, aukit_batch **out);
/* fused pipeline  aukit.dfpwm(data, channels, sr):mono():dfpwm()  (:1392-1414, :677-689, :1005-1018; BASELINE config 4) — the decoded samples never
 * leave the lane that decodes them.  Same bytes as aukit_decode(DFPWM) → aukit_mono → aukit_dfpwm_encode with AUKIT_F64.  channels == 2: a lane per
 * (stream, time chunk) decodes, mixes and encodes (AUKIT_OPT_DFPWM_SPECULATE; one host synchronisation per call); other channel counts: a lane per stream. */
int aukit_dfpwm_transcode_mono(aukit_ctx *ctx, const aukit_batch *in, int channels, aukit_batch **out);
/* Audio:pcm(bitDepth, dataType, interleaved) :901 → unfloored numbers, packed per stream */
int aukit_encode_pcm(aukit_ctx *ctx, const aukit_audio *in, int bit_depth, int data_type, int interleaved, aukit_audio **out);

/* ---- either side of the hot path: structural Audio methods, generators, string packing ----
 * Group calls take `count` audios with the same stream count, dtype and sample rate (the reference resamples mismatched
 * rates first, :702 / :756 — do that with aukit_resample); stream i of the result is built from stream i of every input. */
int aukit_concat(aukit_ctx *ctx, const aukit_audio *const *audios, uint32_t count, aukit_audio **out);   /* Audio:concat :695 */
int aukit_sub(aukit_ctx *ctx, const aukit_audio *in, double start, double last, aukit_audio **out);      /* Audio:sub :725 (0 = default) */
int aukit_combine(aukit_ctx *ctx, const aukit_audio *const *audios, uint32_t count, aukit_audio **out);  /* Audio:combine :751 */
/* one result of Audio:split :781 — `channels` = 1-based channel numbers of the new object */
int aukit_split(aukit_ctx *ctx, const aukit_audio *in, const int32_t *channels, uint32_t count, aukit_audio **out);
int aukit_rep(aukit_ctx *ctx, const aukit_audio *in, double count, aukit_audio **out);                   /* Audio:rep :839 */
int aukit_reverse(aukit_ctx *ctx, const aukit_audio *in, aukit_audio **out);                             /* Audio:reverse :856 */
/* aukit.pcm(data, ...) with `data` a TABLE of numbers, aukit.lua:1077-1096 + :1161-1171 — `n` tables as one host array of doubles with
 * element offsets [n + 1]; d->codec = AUKIT_CODEC_PCM, bit_depth / data_type / channels / sample_rate / interleaved as for the string.
 * Values are taken as they are (no range check, fractions allowed: s / (s < 0 and 2^(b-1) or 2^(b-1)-1), :1082).  Storage = the context's dtype. */
int aukit_decode_table(aukit_ctx *ctx, const double *values, const uint64_t *offsets, uint32_t n, const aukit_codec_desc *d, aukit_audio **out);
/* aukit.adpcm(data, ...) with `data` a TABLE of nibbles, aukit.lua:1183-1184 + :1232-1238 (`read()` hands out data[pos]; `len = #data / channels`):
 * `n` tables as one host array of nibbles (one per byte, 0..15; anything else indexes ima_index_table with nil: AUKIT_E_LUA) with element
 * offsets [n + 1]; d->codec = AUKIT_CODEC_ADPCM, channels / sample_rate / interleaved / predictor / step_index as for the string (top_first
 * has no meaning for a table). */
int aukit_decode_nibbles(aukit_ctx *ctx, const uint8_t *nibbles, const uint64_t *offsets, uint32_t n, const aukit_codec_desc *d, int dtype, aukit_audio **out);
typedef enum { AUKIT_WAVE_NONE = 0 /* aukit.new: silence */, AUKIT_WAVE_SINE = 1, AUKIT_WAVE_TRIANGLE = 2, AUKIT_WAVE_SAWTOOTH = 3, AUKIT_WAVE_SQUARE = 4 } aukit_wave;
/* aukit.new(duration, channels, sampleRate) :1783 / aukit.tone(frequency, duration, amplitude, waveType, duty, channels, sampleRate)
 * :1808 — `n` identical audios (aukit.noise draws from the host VM's math.random and cannot be reproduced) */
int aukit_tone(aukit_ctx *ctx, uint32_t n, double frequency, double duration, double amplitude, int wave, double duty, int channels,
               double sample_rate, int dtype, aukit_audio **out);
/* aukit.noise(duration, amplitude, channels, sampleRate) :1840-1853 — `n` audios of white noise, (random() * 2 - 1) * amplitude per sample.  The
 * reference draws from its VM's math.random: NOT reproducible, by anyone.  Here: Philox4x32-10 on the device, keyed by `seed`, counter =
 * (sample index, channel, stream) — every sample of every channel of every stream its own draw, 53 random bits to a double in [0, 1), the
 * same audio for the same seed.  Same argument checks and error strings as the reference's. */
int aukit_noise(aukit_ctx *ctx, uint32_t n, double duration, double amplitude, int channels, double sample_rate, uint64_t seed, int dtype, aukit_audio **out);
/* What string.pack does with a sample that has no integer representation is the host VM's business, not aukit.lua's
 * (Audio:pcm hands it unfloored numbers, :875): truncate like a Java (long) cast (CC: Tweaked's VM), floor, or raise like PUC Lua 5.3. */
typedef enum { AUKIT_PACK_TRUNC = 0, AUKIT_PACK_FLOOR = 1, AUKIT_PACK_STRICT = 2,
               AUKIT_PACK_PREENCODED = 8 /* OR-ed in: the audio already holds Audio:pcm's numbers (aukit.pack on a number table, aukit.lua:1861): only packed */
} aukit_pack_mode;
/* aukit.pack(audio:pcm(bitDepth, dataType, interleaved), bitDepth, dataType, bigEndian) :901 + :1861 → one byte string per
 * stream; with (bitDepth, bitDepth == 8 ? unsigned : signed, little-endian, interleaved) these are the sample bytes of Audio:wav :966-971 */
int aukit_pack_pcm(aukit_ctx *ctx, const aukit_audio *in, int bit_depth, int data_type, int big_endian, int interleaved, int int_mode,
                   aukit_batch **out);

/* ---- aukit.stream.<codec>(data, ...) with string input, every call of the iterator at once ----
 * out audio: per stream the concatenation of all chunks (`channels` = number of chunk tables);
 * chunk metadata is read back with aukit_stream_chunks(). */
typedef struct aukit_chunks aukit_chunks;
int aukit_stream_decode(aukit_ctx *ctx, const aukit_batch *in, const aukit_codec_desc *desc, int interp, int mono,
                        int dtype, aukit_audio **out, aukit_chunks **chunks);
/* aukit.stream.pcm(data, ...) with `data` a TABLE of numbers, aukit.lua:2255-2290 (`read()` hands out `data[pos]` normalised like the string's
 * samples; `len = #data / channels`): every iterator call at once, as aukit_stream_decode.  Values are taken as they are; reference-order
 * fp64 arithmetic whatever the storage type (AUKIT_F64 / AUKIT_F32); sinc and rates above 48 kHz refused like the string's (Q3). */
int aukit_stream_decode_table(aukit_ctx *ctx, const double *values, const uint64_t *offsets /* n + 1, elements */, uint32_t n, const aukit_codec_desc *d, int interp,
                              int mono, int dtype, aukit_audio **out, aukit_chunks **chunks);
/* per stream s: nchunks[s]; chunk k of stream s: length and the iterator's second return value.
 * status[s]: 0 = iterator ended with nil, AUKIT_E_LUA = the reference iterator raises after the last chunk. */
int aukit_chunks_info(const aukit_chunks *c, uint32_t *n, uint32_t *max_chunks);
int aukit_chunks_get(const aukit_chunks *c, uint32_t *nchunks /* n */, uint32_t *lens /* n*max */, double *pos /* n*max */,
                     int32_t *status /* n */, double *length_seconds /* n */);
void aukit_chunks_free(aukit_chunks *c);

/* ---- several GPUs of one node.  No function of the path reads another stream, so a batch shards by stream index with no
 * data-path collective: aukit_partition cuts it into contiguous, byte-balanced ranges; a GROUP is one context per device in this process
 * (the Lua host is one process that owns the node's GPUs); aukit_group_scatter hands every member its range of a batch that lives on the
 * root's device — device to device over xGMI, every peer's share in flight at once (hipMemcpyPeerAsync per peer, or, with
 * AUKIT_GROUP_TRANSPORT=rccl, grouped ncclSend / ncclRecv on an ncclCommInitAll communicator) — each member then runs the ordinary calls
 * on aukit_group_ctx(g, r) with its shard, and aukit_group_gather_* concatenates the results on the root in rank order.  The reference has
 * no counterpart (a CC computer has one speaker bus): this is the boundary a host such as austream.lua's source loop (:85-92) would use to
 * feed a whole node.  One process per GPU instead: aukit_amd/shard.py over torch.distributed (RCCL), same partition. ---- */
typedef struct aukit_group aukit_group;
/* cuts[g] .. cuts[g + 1] (g = 0 .. world-1; cuts has world + 1 entries) are rank g's streams */
int aukit_partition(const uint64_t *sizes /* n: bytes per stream */, uint32_t n, uint32_t world, uint32_t *cuts);
int aukit_group_create(aukit_group **out, const int *devices, uint32_t n_devices);   /* a device may appear more than once (tests on one GPU) */
void aukit_group_destroy(aukit_group *g);
int aukit_group_info(const aukit_group *g, uint32_t *n_devices, int *transport /* 0 = peer copies, 1 = RCCL */);
aukit_ctx *aukit_group_ctx(aukit_group *g, uint32_t rank);                            /* owned by the group */
int aukit_group_sync(aukit_group *g);
/* shards: n_devices entries (NULL or batches to replace); the root's shard is a view of `whole` — keep `whole` alive while shards are used */
int aukit_group_scatter(aukit_group *g, uint32_t root, const aukit_batch *whole, aukit_batch **shards, uint32_t *cuts /* n_devices + 1 */);
int aukit_group_gather_audio(aukit_group *g, uint32_t root, aukit_audio *const *parts /* n_devices */, aukit_audio **whole);
int aukit_group_gather_batch(aukit_group *g, uint32_t root, aukit_batch *const *parts /* n_devices */, aukit_batch **whole);
/* The members' work, side by side (round 4).  Calling the single-GPU entry points on aukit_group_ctx(g, r) one member after another from the
 * host's one thread runs them one AFTER another wherever an entry point waits for its device (the block codecs read counts back: FLAC five
 * times per call, DFPWM, MS-ADPCM, QOA, IMA) — eight shards of BASELINE configs 4 and 5 in sequence.  aukit_group_run hands every member a
 * list of calls; the group's worker threads — one per member, each bound to its member's device and context — run the lists concurrently
 * and the call returns when all are through ("one scatter, N independent runs, one gather": SURVEY.md §8e).  A call is one entry point:
 * its `op` names it, the other fields are that entry point's arguments (unused ones ignored); a member's calls run in order, so a pipeline
 * (decode_resample -> effect -> effect -> mono) is one list.  `calls` holds n_per_member entries per member, member-major; an entry with op
 * AUKIT_GOP_NONE is skipped.  Returns the first failing member's status (its message in aukit_last_error()); the other members finish their
 * lists regardless.  No callbacks into the host: a LuaJIT host calls this from its one Lua state. */
typedef enum {
    AUKIT_GOP_NONE = 0, AUKIT_GOP_DECODE = 1, AUKIT_GOP_DECODE_RESAMPLE = 2, AUKIT_GOP_STREAM_DECODE = 3, AUKIT_GOP_RESAMPLE = 4, AUKIT_GOP_MONO = 5,
    AUKIT_GOP_EFFECT = 6, AUKIT_GOP_DFPWM_ENCODE = 7, AUKIT_GOP_DFPWM_TRANSCODE_MONO = 8, AUKIT_GOP_ENCODE_PCM = 9, AUKIT_GOP_SYNC = 10
} aukit_group_op;
typedef struct {
    int32_t op;                     /* aukit_group_op */
    int32_t dtype, interp, mono;    /* decode / decode_resample / stream_decode */
    const aukit_batch *batch;       /* decode, decode_resample, stream_decode, dfpwm_transcode_mono: the member's shard */
    const aukit_codec_desc *desc;
    aukit_audio *audio;             /* resample, mono, dfpwm_encode, encode_pcm: the input; effect: the audio changed in place */
    aukit_audio **out_audio;        /* decode*, stream_decode, resample, mono, encode_pcm */
    aukit_batch **out_batch;        /* dfpwm_encode, dfpwm_transcode_mono */
    aukit_chunks **out_chunks;      /* stream_decode (may be NULL) */
    double new_rate;                /* decode_resample, resample */
    int32_t effect_id, nargs;       /* effect */
    double args[8];
    int32_t channels, interleaved;  /* dfpwm_transcode_mono: channels; dfpwm_encode / encode_pcm: interleaved */
    int32_t bit_depth, data_type;   /* encode_pcm */
} aukit_group_call;
int aukit_group_run(aukit_group *g, const aukit_group_call *calls /* n_devices * n_per_member */, uint32_t n_per_member);
/* when the members of the last aukit_group_run started and ended, in milliseconds after the first of them started (host clock around each
 * member's list, the member's stream drained at the end): evidence of the overlap */
int aukit_group_last_run(const aukit_group *g, double *start_ms /* n_devices */, double *end_ms /* n_devices */);

/* ---- aukit.stream.<codec>(fn, ...): the reader-FUNCTION input (aukit.lua:2776-2786 and siblings; austream.lua:19-64), as a resumable handle.
 * Bytes are fed in any pieces; the chunks handed out are exactly those aukit.stream.<codec>(s, ...) hands out for the string s = every
 * byte fed (the reference's own function mode cuts chunks wherever the reader's buffers end, SURVEY Q6: nothing to reproduce there).
 * A chunk is delivered once it is decided — a later chunk exists, or aukit_stream_finish was called.  One stream per handle. */
typedef struct aukit_stream aukit_stream;
typedef enum { AUKIT_STREAM_CHUNK = 0, AUKIT_STREAM_NEED_INPUT = 1, AUKIT_STREAM_END = 2 } aukit_stream_state;
int aukit_stream_open(aukit_ctx *ctx, const aukit_codec_desc *desc, int interp, int mono, int dtype, aukit_stream **out);
int aukit_stream_feed(aukit_stream *s, const uint8_t *bytes, uint64_t n);   /* fn() returned a string */
int aukit_stream_finish(aukit_stream *s);                                    /* fn() returned nil */
/* the iterator call: state = CHUNK (`*len` samples per channel at dst + c * cap, `*pos` = its second return value), NEED_INPUT, or END (nil).
 * `dst_elems` = doubles available at dst: a chunk of `*channels` channels needs channels * cap of them — a call that offers fewer is refused
 * with AUKIT_E_ARG before anything is written (`*len`, `*channels` say what the chunk needs; it stays undelivered: offer more and call again).
 * Where the reference's iterator raises instead of ending, the call returns AUKIT_E_LUA.  A chunk never exceeds 48000 samples per channel
 * except stream.flac / stream.qoa (one coded block resampled: at most 65535 * 48000 / sampleRate). */
int aukit_stream_next(aukit_stream *s, double *dst, uint64_t dst_elems, uint32_t cap, uint32_t *len, int32_t *channels, double *pos, int32_t *state);
int aukit_stream_length(aukit_stream *s, double *seconds);                   /* the factory's second return value, for the bytes fed so far */
/* stream bytes resident on the device, bytes dropped in front of them (stream.pcm / g711 / adpcm / msadpcm drop what delivered calls consumed:
 * austream.lua:19-64 feeds live sources for hours), and the input bytes of every decode so far, summed */
int aukit_stream_resident(const aukit_stream *s, uint64_t *resident, uint64_t *dropped, uint64_t *decoded_total);
void aukit_stream_close(aukit_stream *s);

#ifdef __cplusplus
}
#endif
#endif

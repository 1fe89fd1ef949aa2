// resample.h — the segment-based fused "stage → interpolate → epilogue" engine.
//
// Every resampling loop of the reference has the same shape (aukit.lua:662-671, :2395-2405,
// :2818-2828, :2900-2910 ...): a Lua table `d` with valid indices w_lo..w_hi, and outputs
// i = 1..n_out at position x = (i-1)/ratio + 1, `if x % 1 == 0 then d[x] else interp(d, x)`.
// A *segment* is one such (table, output line) pair; the host expands an API call into segments
// (one per stream for the Audio path, one per stream × chunk for stream.pcm / stream.g711 ...),
// and one kernel walks tiles of segments: stage the needed window of the table into LDS as fp64
// (decoding PCM / G.711 bytes on the way, so decoded samples never touch HBM), interpolate with the
// reference's fp64 operation order, apply the epilogue, store coalesced.
#pragma once
#include "common.h"

namespace aukit {

struct Seg {
    long long src_base;   // source frame index (within the stream / row) of table index 0
    int w_lo, w_hi;       // valid table indices (anything else reads as nil)
    unsigned n_out;       // number of outputs
    unsigned stream;      // source stream (batch sources) or source row (audio sources)
    unsigned long long out_off;  // element offset of output channel 0, output index 0
    unsigned out_stride;  // element stride between output channels
    unsigned pad;
};
static_assert(sizeof(Seg) == 40, "Seg layout");

enum SrcKind { SRC_PCM_GENERIC = 0, SRC_PCM_S16LE_MONO = 1, SRC_G711 = 2, SRC_G711_MONO = 3, SRC_AUDIO_F64 = 4, SRC_AUDIO_F32 = 5, SRC_I16 = 6, SRC_I8 = 7, SRC_I32 = 8,
               SRC_PCM_S16LE_STEREO = 9 /* fast path only (fast_s16x2.hip): interleaved 16-bit stereo frames */,
               SRC_PCM8_MONO = 10 /* fast path only (fast_stream_u8.hip): 8-bit mono, signed or unsigned (P.data_type) */ };
enum EpiKind {
    EPI_AUDIO = 0,       // Audio:resample  :666-668  (integer x copies unclamped, else clamp ±1)
    EPI_STREAM_PCM = 1,  // stream.pcm      :2397-2403 (no clamp of interp, 2-tap FIR, ×127/128, clamp ±128/127)
    EPI_STREAM_FLOOR = 2,// stream.g711/adpcm/msadpcm :2900-2910 (optional mono mean, floor, clamp)
    EPI_STREAM_DFPWM = 3 // stream.dfpwm :2478-2489 (interp clamped to ±128/127, same sample to every channel)
};

struct ResampleParams {
    // tiling
    const Seg *segs;
    const unsigned *tile_seg;   // tile → segment (ragged batches), or null
    const unsigned *seg_tile0;  // segment → its first tile (ragged batches)
    unsigned tiles_per_seg;     // != 0: uniform batch, tile → (tile / tps, tile % tps)
    unsigned n_tiles;
    int tile_out;               // outputs per tile
    int cap;                    // LDS doubles per staged channel
    // position arithmetic: x = (i-1)/ratio + 1
    double ratio, rcp;
    int exact_rcp;              // 1: RN((i-1)/ratio) via rcp + two fmas is verified exact for this launch
    int halo_l, halo_r;         // taps below / above floor(x)
    int sinc_w;
    int sinc_hole;   // stream.msadpcm stereo (:2640-2643): table indices -N .. -1 are the block before (entry i of it at i - N - 1), index 0 is nil — sinc only reaches there
    // source
    const unsigned char *src;
    const unsigned long long *src_off;     // per stream: byte offset (batch) / element offset (audio rows)
    const unsigned long long *src_frames;  // per stream: frames per channel (planar PCM)
    const unsigned char *safe_lo, *safe_hi; // the allocation: a 16-byte vector load at p needs safe_lo <= p and p + 16 <= safe_hi
    int channels;      // channels in the source data
    int stage_channels;// channels staged per tile (1 when the source is pre-mixed or planar rows)
    int bit_depth, data_type, big_endian, planar, ulaw;
    int premix_mono;   // stream.pcm mono: mean of the channels at read time (:2368)
    double norm_pos, norm_neg;  // SRC_I16 / SRC_I8 / SRC_I32: v / (v < 0 ? norm_neg : norm_pos)
    double g711_scale;          // 1/0x2000 (aukit.g711) or 1/0x40 (stream.g711): exact power of two
    // epilogue
    void *out;
    double lp_alpha;
    int mix_mono;      // EPI_STREAM_FLOOR: 1 = mean over channels after interpolation (:2905-2908), 2 = l + r/2 (:2672)
    int pos_mul;       // x = ((i-1) * pos_mul) / ratio + 1 (stream.dfpwm steps i by `channels`); 0 means 1
    int out_channels;  // EPI_STREAM_DFPWM: rows written per output
    int nt_store;      // fast kernels: non-temporal output stores (tuning knob AUKIT_NT_STORE)
    int table;         // SRC_PCM_GENERIC: the "string" is a Lua TABLE of numbers (doubles, 8 bytes each: aukit.lua:2255-2290), read as they are
                       // (last members: the offsets of everything the wave kernels read stay what they were)
    const int *only_if;// k_resample: non-null → the launch does nothing unless *only_if != 0 (fast_fmt.hip's flag: a float string with samples beyond ±1)
};

// launches the right instantiation; `name` receives a static string naming the kernel
int launch_resample(aukit_ctx *ctx, int src_kind, int interp, int epi, int out_dtype, const ResampleParams &P, size_t lds_bytes,
                    uint64_t algorithmic_bytes, const char **name);

// host-side tiling helper: fills the tile tables for a list of segments, uploads segs/tiles to ctx scratch
// and completes P.{segs,tile_seg,seg_tile0,tiles_per_seg,n_tiles,tile_out,cap,halo_*,ratio,rcp,exact_rcp}.
int plan_tiles(aukit_ctx *ctx, const std::vector<Seg> &segs, double ratio, int interp, int stage_channels, ResampleParams &P, size_t *lds_bytes);

int plan_tiles_sized(aukit_ctx *ctx, const std::vector<Seg> &segs, int tile_out, ResampleParams &P);

// fast.hip / fast2.hip: f32 tolerance path (exact rational positions, f32 FMA taps)
struct FastParams {
    unsigned a, b;       // x - 1 = o * a / b
    unsigned magic;      // ceil(2^32 / b): q = mulhi(n, magic) exact for n * b < 2^32
    float inv_b;
    int tile_out;        // v1: outputs per workgroup tile (multiple of 1024)
    int cap;             // LDS floats per staged window (v1: per workgroup, v2: per wave)
    float scale_pos, scale_neg;  // s16: 1/32767, 1/32768
    unsigned wc, wd;     // v2: (1024 * a) = wc * b + wd  (one wave tile = 1024 outputs)
    int epi;             // wave kernel epilogue: 0 = Audio:resample, 1 = stream.pcm (fast_stream.hip); appended last so that the
    float alpha;         //   kernel-argument offsets the headline kernel reads stay what they were.  alpha: stream.pcm's low-pass weight
    unsigned dq64, dr64; // 64 * a = dq64 * b + dr64: one row of a wave tile further down, (q, rem) advance by (dq64, dr64) with one carry
};
int launch_fast_wave_stream(aukit_ctx *ctx, int interp, int nv, const ResampleParams &P, const FastParams &F, size_t lds, unsigned grid);
int launch_fast_wave_stream_u8(aukit_ctx *ctx, int interp, int nv, const ResampleParams &P, const FastParams &F, size_t lds, unsigned grid);   // the same on 8-bit mono strings
int launch_fast_wave_stream_f32(aukit_ctx *ctx, int interp, int nv, const ResampleParams &P, const FastParams &F, size_t lds, unsigned grid);  // the same on f32 rows
int launch_fast_wave_s16x2(aukit_ctx *ctx, int interp, int nv, const ResampleParams &P, const FastParams &F, unsigned grid);
bool exact_wave_try(aukit_ctx *ctx, int src_kind, int interp, double old_rate, double new_rate, const std::vector<Seg> &segs, ResampleParams &P, int dtype,
                    uint64_t algorithmic_bytes, int *rc, int epi = 0);  // exact_wave.hip: reference-order fp64 on the wave tile engine (epi 1: stream.pcm)
bool wave_f64_try(aukit_ctx *ctx, int src_kind, int interp, double old_rate, double new_rate, const std::vector<Seg> &segs, ResampleParams &P,
                  uint64_t algorithmic_bytes, int *rc, int epi = 0, double alpha = 0);  // wave_f64.hip: fp64 arithmetic, f32 store (AUKIT_OPT_EXACT_MATH = 1)
bool wave_coef_f64_try(aukit_ctx *ctx, int src_kind, int interp, double old_rate, double new_rate, const std::vector<Seg> &segs, ResampleParams &P,
                       uint64_t algorithmic_bytes, int *rc);  // wave_coef_f64.hip: G.711 mono, up-sampling by > 4.6, fp64 arithmetic, f32 store
int launch_fast_wave_stream_s16x2(aukit_ctx *ctx, int interp, int nv, const ResampleParams &P, const FastParams &F, unsigned grid);  // F.epi 1: stereo, 2: mono
int launch_fast_wave_coef(aukit_ctx *ctx, int src_kind, int interp, int nv, int win, const ResampleParams &P, const FastParams &F, unsigned grid);
int launch_fast_wave(aukit_ctx *ctx, int src_kind, int interp, const std::vector<Seg> &segs, ResampleParams &P, FastParams &F,
                     uint64_t algorithmic_bytes, bool *taken);
bool fast_fmt_try(aukit_ctx *ctx, const aukit_codec_desc *d, int interp, double new_rate, const std::vector<Seg> &segs, ResampleParams &P,
                  uint64_t algorithmic_bytes, int *rc, const int **only_if, int epi = 0, double alpha = 0);  // fast_fmt.hip: every other interleaved PCM format / G.711 stereo, one launch (epi 1 / 2: stream.pcm per channel / on the channels' mean)
bool fast_try(aukit_ctx *ctx, int src_kind, int interp, double old_rate, double new_rate, const std::vector<Seg> &segs, ResampleParams &P,
              uint64_t algorithmic_bytes, int *rc, int epi = 0, double alpha = 0);

// position of output o (0-based) exactly as the reference computes it on the host
static inline double host_pos(uint64_t o, double ratio) { return ((double)o) / ratio + 1; }

}  // namespace aukit

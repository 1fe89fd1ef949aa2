/* ork.h — CPU ORACLE for the aukit_amd hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a scalar fp64, op-for-op restatement in plain C of the per-sample
 * arithmetic of MCJack123/AUKit 1.10.0 (`aukit.lua`, cited as aukit.lua:LINE
 * on every function).  It exists so that the HIP path can be checked against
 * the reference's arithmetic; it is NOT part of the product:
 *   - only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 *     load it; aukit_amd/ never imports, links or calls anything in oracle/.
 *
 * PARITY PINNING STATUS:
 *   - The reference is pure Lua for the ComputerCraft sandbox, ships no tests,
 *     fixtures or golden vectors, and no Lua interpreter exists in the build
 *     image, so the reference cannot be executed to pin this restatement.
 *     What pins it: independent known-answer checks that do not depend on the
 *     restatement (ITU-T G.711 tables, FLAC losslessness, QOA int32 reference
 *     semantics, hand-computed micro-vectors for each reference quirk), and a
 *     second, structurally different reading of the same Lua: line-by-line
 *     Python transliterations of every stream iterator (pcm, adpcm, msadpcm,
 *     g711, flac, qoa, dfpwm), of the effects and of Audio:resample / mono /
 *     mix / pcm, run next to this restatement — see tests/test_oracle_*.py.
 *   - DFPWM arithmetic is NOT in the reference tree (`require "cc.audio.dfpwm"`,
 *     aukit.lua:85, a CC: Tweaked ROM module with no pinned version).  It is
 *     restated here from the published DFPWM1a algorithm: **parity unpinned**.
 *
 * Conventions: Lua tables are 1-based; a C array `a` of length n holds Lua
 * a[1..n] at a[0..n-1] unless a function says otherwise.  "nil" reads are
 * modelled explicitly.  Where the reference would raise a Lua runtime error the
 * function returns ORK_E_LUA and sets ork_last_error().
 */
#ifndef ORK_H
#define ORK_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORK_MAX_CH 64   /* (channels the checker carries: the product's AUKIT_MAX_PLANAR_CHANNELS) */

enum { ORK_OK = 0, ORK_E_ARG = -1, ORK_E_LUA = -2, ORK_E_NOMEM = -3, ORK_E_UNSUPPORTED = -4 };
enum { ORK_INTERP_NONE = 0, ORK_INTERP_LINEAR = 1, ORK_INTERP_CUBIC = 2, ORK_INTERP_SINC = 3 };
enum { ORK_SIGNED = 0, ORK_UNSIGNED = 1, ORK_FLOAT = 2 };

/* aukit.Audio (aukit.lua:116-123): planar per-channel arrays of doubles. */
typedef struct {
    int channels;
    double sample_rate;
    size_t len[ORK_MAX_CH];
    double *data[ORK_MAX_CH];
} ork_audio;

/* Everything a stream.* iterator returns until it returns nil, concatenated. */
typedef struct {
    int channels;            /* number of output channel tables per chunk            */
    int nchunks;
    size_t *chunk_len;       /* [nchunks*channels] length of chunk[c][ch]            */
    double *chunk_pos;       /* [nchunks] second return value of the iterator        */
    size_t len[ORK_MAX_CH];  /* total samples per channel                            */
    double *data[ORK_MAX_CH];
    double length_seconds;   /* second return value of the factory                   */
    int final_status;        /* ORK_OK: iterator returned nil; ORK_E_LUA: it raised  */
} ork_stream;

const char *ork_last_error(void);
void ork_audio_free(ork_audio *a);
void ork_stream_free(ork_stream *s);
void ork_free(void *p);
/* sincWindowSize (aukit.lua:129): 10, or 30 when running under LuaJIT. */
void ork_set_sinc_window(int w);

/* ---- helpers (aukit.lua:228-284) ---- */
double ork_clamp(double n, double mn, double mx);
/* interpolate[mode](data, x) on a plain table data[1..n]; returns ORK_E_LUA on nil arithmetic. */
int ork_interp(int mode, const double *data, size_t n, double x, double *out);

/* ---- Audio methods ---- */
int ork_resample(const ork_audio *in, double new_rate, int interp, ork_audio *out);      /* aukit.lua:653-673 */
int ork_mono(const ork_audio *in, ork_audio *out);                                      /* aukit.lua:677-689 */
int ork_mix(const ork_audio *const *audios, int n, double amplifier, ork_audio *out);   /* aukit.lua:804-835 */
/* Audio:pcm(bitDepth,dataType,interleaved) → number table (aukit.lua:868-910) */
int ork_encode_pcm(const ork_audio *in, int bit_depth, int data_type, int interleaved, double **out, size_t *n);
/* Audio:dfpwm(interleaved) (aukit.lua:1005-1018) → bytes */
int ork_audio_dfpwm(const ork_audio *in, int interleaved, uint8_t **out, size_t *n);

/* ---- loaders ---- */
int ork_pcm(const uint8_t *data, size_t nbytes, int bit_depth, int data_type, int channels, double sample_rate,
            int interleaved, int big_endian, ork_audio *out);                            /* aukit.lua:1049-1171 */
int ork_pcm_table(const double *values, size_t n, int bit_depth, int data_type, int channels, double sample_rate,
                  int interleaved, ork_audio *out);                                      /* aukit.lua:1077-1096 */
int ork_adpcm(const uint8_t *data, size_t nbytes, int channels, double sample_rate, int top_first, int interleaved,
              const int *predictor, const int *step_index, ork_audio *out);             /* aukit.lua:1183-1274 (string input) */
int ork_adpcm_nibbles(const uint8_t *nib, size_t n, int channels, double sample_rate, int interleaved,
                      const int *predictor, const int *step_index, ork_audio *out);     /* aukit.lua:1183-1274 (table input) */
int ork_wav_adpcm(const uint8_t *data, size_t nbytes, int block_align, int channels, double sample_rate, ork_audio *out); /* aukit.lua:1509-1548 */
int ork_msadpcm(const uint8_t *data, size_t nbytes, int block_align, int channels, double sample_rate,
                const int *coef1, const int *coef2, int ncoef, ork_audio *out);         /* aukit.lua:1283-1353 */
int ork_g711(const uint8_t *data, size_t nbytes, int ulaw, int channels, double sample_rate, ork_audio *out); /* aukit.lua:1361-1384 */
int ork_dfpwm(const uint8_t *data, size_t nbytes, int channels, double sample_rate, ork_audio *out);          /* aukit.lua:1392-1414 */
int ork_mdfpwm(const uint8_t *data, size_t nbytes, ork_audio *out);                                            /* aukit.lua:1420-1448 */
int ork_qoa(const uint8_t *data, size_t nbytes, ork_audio *out);                                               /* aukit.lua:1706-1777 */
int ork_flac(const uint8_t *data, size_t nbytes, ork_audio *out);                                              /* aukit.lua:311-619, 1657 */

/* ---- cc.audio.dfpwm (external; restated, parity unpinned) ---- */
typedef struct { int charge, strength, previous_bit; } ork_dfpwm_pred;
typedef struct { ork_dfpwm_pred p; int low_pass_charge, previous_charge, previous_bit; } ork_dfpwm_dec;
typedef struct { ork_dfpwm_pred p; int previous_charge; } ork_dfpwm_enc;
void ork_dfpwm_dec_init(ork_dfpwm_dec *d);
void ork_dfpwm_enc_init(ork_dfpwm_enc *e);
void ork_dfpwm_decode(ork_dfpwm_dec *d, const uint8_t *in, size_t nbytes, int8_t *out /* 8*nbytes */);
/* returns ORK_E_LUA if a floor(sample) is outside [-128,127]; out has ceil(n/8) bytes */
int ork_dfpwm_encode(ork_dfpwm_enc *e, const double *samples, size_t n, uint8_t *out);

/* ---- effects (aukit.lua:3356-3618), in place ---- */
int ork_fx_amplify(ork_audio *a, double multiplier);
int ork_fx_speed(ork_audio *a, double multiplier, int default_interp);
int ork_fx_fade(ork_audio *a, double start_time, double start_amp, double end_time, double end_amp);
int ork_fx_invert(ork_audio *a);
int ork_fx_normalize(ork_audio *a, double peak, int independent);
int ork_fx_center(ork_audio *a);
int ork_fx_trim(ork_audio *a, double threshold);
int ork_fx_delay(ork_audio *a, double delay, double multiplier);
int ork_fx_echo(ork_audio *a, double delay, double multiplier);
int ork_fx_reverb(ork_audio *a, double delay, double decay, double wet, double dry);
int ork_fx_lowpass(ork_audio *a, double frequency);
int ork_fx_highpass(ork_audio *a, double frequency);

/* ---- aukit.stream.* with string input; `interp` = aukit.defaultInterpolation ---- */
int ork_stream_pcm(const uint8_t *data, size_t nbytes, int bit_depth, int data_type, int channels, double sample_rate,
                   int big_endian, int mono, int interp, ork_stream *out);              /* aukit.lua:2228-2424 */
int ork_stream_dfpwm(const uint8_t *data, size_t nbytes, double sample_rate, int channels, int mono, int interp, ork_stream *out); /* :2439-2496 */
int ork_stream_mdfpwm(const uint8_t *data, size_t nbytes, int mono, ork_stream *out);   /* :2507-2572 */
int ork_stream_msadpcm(const uint8_t *data, size_t nbytes, int block_align, int channels, double sample_rate, int mono,
                       const int *coef1, const int *coef2, int ncoef, int interp, ork_stream *out); /* :2588-2736 */
int ork_stream_adpcm(const uint8_t *data, size_t nbytes, int block_align, int channels, double sample_rate, int mono,
                     int interp, ork_stream *out);                                       /* :2753-2835 */
/* stream.g711 with string input never returns nil (Q13); max_calls bounds the emulation. */
int ork_stream_g711(const uint8_t *data, size_t nbytes, int ulaw, int channels, double sample_rate, int mono,
                    int interp, int max_calls, ork_stream *out);                         /* :2850-2913 */
int ork_stream_flac(const uint8_t *data, size_t nbytes, int interp, ork_stream *out);   /* :3124-3191 */
int ork_stream_qoa(const uint8_t *data, size_t nbytes, int mono, int interp, ork_stream *out); /* :3202-3337 */

#ifdef __cplusplus
}
#endif
#endif

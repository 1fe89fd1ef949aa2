/* ork_internal.h — shared internals of the CPU ORACLE (test infrastructure only). */
#ifndef ORK_INTERNAL_H
#define ORK_INTERNAL_H
#define _GNU_SOURCE
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ork.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

int ork__fail(int code, const char *fmt, ...);
int ork__audio_alloc(ork_audio *a, int channels, size_t len, double rate);
extern int ork__sinc_window;

/* models `data[idx]`: 1 = value, 0 = nil, <0 = error raised by an __index metamethod */
typedef int (*ork_getter)(void *ctx, long idx, double *v);
int ork__interp_get(int mode, ork_getter get, void *ctx, double x, double *out);

/* plain Lua table with values at indices lo..hi */
typedef struct { const double *p; long lo, hi; } ork_plain;
int ork__plain_get(void *ctx, long idx, double *v);

/* growable double vector (Lua array part being appended to) */
typedef struct { double *p; size_t n, cap; } ork_vec;
static inline int ork__vec_push(ork_vec *v, double x) {
    if (v->n == v->cap) {
        size_t nc = v->cap ? v->cap * 2 : 1024;
        double *np = (double *)realloc(v->p, nc * sizeof(double));
        if (!np) return -1;
        v->p = np; v->cap = nc;
    }
    v->p[v->n++] = x;
    return 0;
}
static inline int ork__vec_set(ork_vec *v, size_t idx0, double x) { /* v[idx0] = x, growing; gaps are zero-filled */
    while (idx0 >= v->cap) {
        size_t nc = v->cap ? v->cap * 2 : 1024;
        double *np = (double *)realloc(v->p, nc * sizeof(double));
        if (!np) return -1;
        memset(np + v->cap, 0, (nc - v->cap) * sizeof(double));
        v->p = np; v->cap = nc;
    }
    v->p[idx0] = x;
    if (idx0 >= v->n) v->n = idx0 + 1;
    return 0;
}

/* stream result builder */
typedef struct {
    ork_vec data[ORK_MAX_CH];
    ork_vec chunk_len; /* stored as doubles */
    ork_vec chunk_pos;
    int channels, nchunks;
} ork_sbuild;
int ork__sb_chunk(ork_sbuild *b, double *const *chunk, const size_t *len, double pos);
int ork__sb_finish(ork_sbuild *b, ork_stream *out, double length_seconds, int final_status);
void ork__sb_abort(ork_sbuild *b);

/* PCM sample unpack shared by aukit.pcm and stream.pcm: returns the integer/float
 * value string.unpack would produce for format i<n>/I<n>/f with the given endianness. */
double ork__unpack_sample(const uint8_t *p, int byte_depth, int data_type, int big_endian);
/* normalisation of aukit.lua:1082/1088/1133/1152 (and 2265/2276/2336/2360) */
static inline double ork__pcm_norm(double s, int data_type, double maxValue) {
    if (data_type == ORK_SIGNED) return s / (s < 0 ? maxValue : maxValue - 1);
    if (data_type == ORK_UNSIGNED) return (s - 128) / (s < 128 ? maxValue : maxValue - 1); /* Q4 */
    return s;
}

/* IMA tables aukit.lua:156-171, MS-ADPCM table :173-176, QOA :1662-1679 */
extern const int ork__ima_index_table[16];
extern const int ork__ima_step_table[89];
int ork__msadpcm_adapt(int nib /* -8..7 */);
extern const int ork__qoa_dequant_tab[16][8];

/* G.711 byte → linear magnitude/sign per aukit.lua:1374-1379; returns m, *neg = divisor is negative */
int ork__g711_expand(int byte, int ulaw, int *neg);

/* FLAC frame-by-frame decode used by ork_flac and ork_stream_flac */
typedef struct ork_flac_dec ork_flac_dec;
int ork__flac_open(const uint8_t *data, size_t n, ork_flac_dec **d, double *sample_rate, int *channels, int *depth, double *num_samples);
/* returns 1 = frame decoded (block_size, out[ch][i] normalised doubles, caller frees out[ch]); 0 = end; <0 error */
int ork__flac_frame(ork_flac_dec *d, double **out, size_t *block_size);
void ork__flac_close(ork_flac_dec *d);

/* QOA helpers aukit.lua:1681-1701 */
typedef struct { double history[4], weights[4]; } ork_qoa_lms;
double ork__qoa_predict(const ork_qoa_lms *l);
void ork__qoa_update(ork_qoa_lms *l, double sample, double residual);

#endif

/* ork_stream.c — CPU ORACLE (test infrastructure only; see ork.h header).
 * aukit.stream.* iterator factories of aukit.lua (AUKit 1.10.0) with STRING input,
 * run to exhaustion; every quirk of SURVEY.md §8.0 is reproduced on purpose.
 */
#include "ork_internal.h"

/* ---------- result builder ---------- */
int ork__sb_chunk(ork_sbuild *b, double *const *chunk, const size_t *len, double pos) {
    for (int c = 0; c < b->channels; c++) {
        for (size_t i = 0; i < len[c]; i++)
            if (ork__vec_push(&b->data[c], chunk[c][i])) return ork__fail(ORK_E_NOMEM, "out of memory");
        if (ork__vec_push(&b->chunk_len, (double)len[c])) return ork__fail(ORK_E_NOMEM, "out of memory");
    }
    if (ork__vec_push(&b->chunk_pos, pos)) return ork__fail(ORK_E_NOMEM, "out of memory");
    b->nchunks++;
    return ORK_OK;
}
void ork__sb_abort(ork_sbuild *b) {
    for (int c = 0; c < ORK_MAX_CH; c++) free(b->data[c].p);
    free(b->chunk_len.p);
    free(b->chunk_pos.p);
    memset(b, 0, sizeof *b);
}
int ork__sb_finish(ork_sbuild *b, ork_stream *out, double length_seconds, int final_status) {
    memset(out, 0, sizeof *out);
    out->channels = b->channels;
    out->nchunks = b->nchunks;
    out->length_seconds = length_seconds;
    out->final_status = final_status;
    size_t ncl = (size_t)b->nchunks * b->channels;
    out->chunk_len = (size_t *)malloc((ncl ? ncl : 1) * sizeof(size_t));
    out->chunk_pos = (double *)malloc((b->nchunks ? b->nchunks : 1) * sizeof(double));
    if (!out->chunk_len || !out->chunk_pos) { ork__sb_abort(b); return ork__fail(ORK_E_NOMEM, "out of memory"); }
    for (size_t i = 0; i < ncl; i++) out->chunk_len[i] = (size_t)b->chunk_len.p[i];
    for (int i = 0; i < b->nchunks; i++) out->chunk_pos[i] = b->chunk_pos.p[i];
    for (int c = 0; c < b->channels; c++) {
        out->data[c] = b->data[c].p ? b->data[c].p : (double *)malloc(8);
        out->len[c] = b->data[c].n;
        b->data[c].p = NULL;
    }
    free(b->chunk_len.p);
    free(b->chunk_pos.p);
    memset(b, 0, sizeof *b);
    return ORK_OK;
}

/* ======================= aukit.stream.pcm  aukit.lua:2228-2424 ======================= */
typedef struct {
    const uint8_t *data;
    size_t total, pos; /* whole samples in the string; next sample */
    int bd, data_type, be;
    double maxValue;
} pcm_reader;

/* read()  aukit.lua:2290-2361: 1 = value, 0 = returned nil (float at end), -1 = raised (int formats at end) */
static int pcm_read(pcm_reader *r, double *v) {
    if (r->pos >= r->total) {
        if (r->data_type == ORK_FLOAT) return 0;
        ork__fail(ORK_E_LUA, "attempt to compare nil with number");
        return -1;
    }
    double s = ork__unpack_sample(r->data + r->pos * (size_t)r->bd, r->bd, r->data_type, r->be);
    r->pos++;
    *v = ork__pcm_norm(s, r->data_type, r->maxValue);
    return 1;
}

/* the lazy per-channel table d[j] with its __index metamethod  aukit.lua:2367-2371 */
typedef struct {
    double *val;
    unsigned char *has; /* 0 = absent, 1 = number, 2 = explicitly stored nil (float EOF: self[i] = nil is a no-op) */
    long lo, cap;
    pcm_reader *rd;
    int mono, channels;
} lazy_tab;

static int lazy_get(void *ctx, long idx, double *v) {
    lazy_tab *t = (lazy_tab *)ctx;
    long k = idx - t->lo;
    if (k < 0 || k >= t->cap) { ork__fail(ORK_E_UNSUPPORTED, "stream.pcm oracle: lazy table index %ld outside modelled window", idx); return -4; }
    if (t->has[k] == 1) { *v = t->val[k]; return 1; }
    /* __index(self, i) */
    if (t->mono) {
        double acc = 0; /* (rawget(self, i) or 0) */
        for (int c = 0; c < t->channels; c++) {
            double s;
            int r = pcm_read(t->rd, &s);
            if (r < 0) return r;
            if (r == 0) { ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value"); return -1; }
            acc = acc + s;
        }
        t->val[k] = acc / t->channels;
        t->has[k] = 1;
    } else {
        double s;
        int r = pcm_read(t->rd, &s);
        if (r < 0) return r;
        if (r == 0) return 0; /* self[i] = nil; rawget → nil */
        t->val[k] = s;
        t->has[k] = 1;
    }
    *v = t->val[k];
    return 1;
}
static int lazy_set(lazy_tab *t, long idx, double v) {
    long k = idx - t->lo;
    if (k < 0 || k >= t->cap) return ork__fail(ORK_E_UNSUPPORTED, "stream.pcm oracle: lazy table index outside modelled window");
    t->val[k] = v;
    t->has[k] = 1;
    return ORK_OK;
}
/* #t : highest n with t[1..n] all present (the table is filled in order; holes → unsupported) */
static long lazy_len(const lazy_tab *t) {
    long n = 0;
    while (1 - t->lo + n < t->cap && t->has[1 - t->lo + n] == 1) n++;
    return n;
}

static const int interpolation_start[4] = {1, 1, 0, 0}; /* aukit.lua:283 */
static const int interpolation_end[4] = {1, 2, 3, 0};   /* aukit.lua:284 */

int ork_stream_pcm(const uint8_t *data, size_t nbytes, int bit_depth, int data_type, int channels, double sample_rate,
                   int big_endian, int mono, int interp, ork_stream *out) {
    if (bit_depth != 8 && bit_depth != 16 && bit_depth != 24 && bit_depth != 32) return ork__fail(ORK_E_ARG, "bad argument #2 (invalid bit depth)");
    if (data_type < 0 || data_type > 2) return ork__fail(ORK_E_ARG, "bad argument #3 (invalid data type)");
    if (data_type == ORK_FLOAT && bit_depth != 32) return ork__fail(ORK_E_ARG, "bad argument #2 (float audio must have 32-bit depth)");
    if (channels < 1) return ork__fail(ORK_E_ARG, "bad argument #4 (number outside of range)");
    if (sample_rate < 1) return ork__fail(ORK_E_ARG, "bad argument #5 (number outside of range)");
    if (channels > ORK_MAX_CH) return ork__fail(ORK_E_UNSUPPORTED, "too many channels for the oracle");
    if (interp < 0 || interp > 3) return ork__fail(ORK_E_ARG, "invalid interpolation");
    if (channels == 1) mono = 0; /* :2243 */
    int bd = bit_depth / 8;
    if (nbytes % (size_t)bd != 0) return ork__fail(ORK_E_UNSUPPORTED, "trailing partial sample (string.rep with a fractional count)");
    pcm_reader rd = {data, nbytes / (size_t)bd, 0, bd, data_type, big_endian, ldexp(1.0, bit_depth - 1)};
    double len = ((double)nbytes / bd) / channels;                   /* :2245 */
    double ratio = 48000 / sample_rate;                              /* :2364 */
    double lp_alpha = 1 - exp(-(sample_rate / 96000) * 2 * M_PI);    /* :2365 */
    int nd = mono ? 1 : channels;
    long W = ork__sinc_window;
    long cap = (long)(48000.0 / ratio) + 64 + 2 * W;
    lazy_tab d[ORK_MAX_CH];
    memset(d, 0, sizeof d);
    ork_sbuild sb;
    memset(&sb, 0, sizeof sb);
    sb.channels = nd;
    double *chunk[ORK_MAX_CH] = {0};
    int rc = ORK_OK, final_status = ORK_OK;
    for (int j = 0; j < nd; j++) {
        d[j].val = (double *)calloc((size_t)cap, sizeof(double));
        d[j].has = (unsigned char *)calloc((size_t)cap, 1);
        chunk[j] = (double *)malloc(48000 * sizeof(double));
        if (!d[j].val || !d[j].has || !chunk[j]) { rc = ork__fail(ORK_E_NOMEM, "out of memory"); goto done; }
        d[j].lo = -W - 2; d[j].cap = cap; d[j].rd = &rd; d[j].mono = mono; d[j].channels = channels;
    }
    double n = 0;
    int ok = 1;
    for (;;) {
        /* ---- one iterator call  :2374-2423 ---- */
        if (!ok) break; /* `complete` is only ever set in function-input mode */
        int ended = 0;
        for (int i = (n == 0 ? interpolation_start[interp] : 1); i <= interpolation_end[interp] && !ended; i++) { /* :2376-2386 */
            if (mono) {
                double s = 0;
                for (int j = 0; j < channels; j++) {
                    double c;
                    int r = pcm_read(&rd, &c);
                    if (r < 0) { final_status = ORK_E_LUA; ended = 1; break; }
                    if (r == 0) { ended = 1; break; } /* if not c then return nil */
                    s = s + c;
                }
                if (!ended) lazy_set(&d[0], i, s / channels);
            } else {
                for (int j = 0; j < channels; j++) {
                    double c;
                    int r = pcm_read(&rd, &c);
                    if (r < 0) { final_status = ORK_E_LUA; ended = 1; break; }
                    if (r == 0) { ended = 1; break; }
                    lazy_set(&d[j], i, c);
                }
            }
        }
        if (ended) break;
        size_t clen[ORK_MAX_CH] = {0};
        double ls[ORK_MAX_CH];
        for (int y = 0; y < nd; y++) { double s = 0; ls[y] = s / (s < 0 ? 128 : 127); } /* chunk[y][0] or 0 → 0 */
        int raised = 0;
        for (int i = 1; i <= 48000 && !raised; i++) { /* pcall body :2389-2406 */
            for (int y = 0; y < nd; y++) {
                double x = (((double)i - 1) / ratio) + 1;
                double s;
                int r;
                if (x == floor(x)) {
                    r = lazy_get(&d[y], (long)x, &s);
                    if (r == 0) { raised = 1; break; } /* s = nil → arithmetic on nil */
                    if (r < 0) { if (r == -4) { rc = ORK_E_UNSUPPORTED; goto done; } raised = 1; break; }
                } else {
                    r = ork__interp_get(interp, lazy_get, &d[y], x, &s);
                    if (r == ORK_E_UNSUPPORTED || r == -4) { rc = ORK_E_UNSUPPORTED; goto done; }
                    if (r != ORK_OK) { raised = 1; break; }
                }
                double ns = ls[y] + lp_alpha * (s - ls[y]);
                chunk[y][i - 1] = ork_clamp(ns * (ns < 0 ? 128 : 127), -128, 127);
                clen[y] = (size_t)i;
                ls[y] = s; /* Q2: the RAW sample, not ns */
            }
        }
        ok = !raised;
        if (clen[0] == 0) break;                 /* :2407 */
        n = n + (double)clen[0];                 /* :2408 */
        for (int y = 0; y < nd; y++) {           /* window re-base :2409-2421 */
            long l = lazy_len(&d[y]);
            if (interp == ORK_INTERP_SINC) {
                double t[64];
                unsigned char th[64];
                for (long i = -W; i <= 0; i++) {
                    double v;
                    int r = lazy_get(&d[y], l + i, &v); /* may invoke __index exactly like the Lua */
                    if (r == -4) { rc = ORK_E_UNSUPPORTED; goto done; }
                    th[i + W] = (r == 1);
                    t[i + W] = r == 1 ? v : 0;
                    if (r < 0) { final_status = ORK_E_LUA; ok = 0; }
                }
                memset(d[y].has, 0, (size_t)cap);
                for (long i = -W; i <= 0; i++) if (th[i + W]) lazy_set(&d[y], i, t[i + W]);
            } else {
                double l2 = 0, l1 = 0;
                int h2 = ok ? lazy_get(&d[y], l - 1, &l2) : 0;
                int h1 = ok ? lazy_get(&d[y], l, &l1) : 0;
                if (h2 == -4 || h1 == -4) { rc = ORK_E_UNSUPPORTED; goto done; }
                memset(d[y].has, 0, (size_t)cap);
                if (h2 == 1) lazy_set(&d[y], -1, l2);
                if (h1 == 1) lazy_set(&d[y], 0, l1);
            }
        }
        if ((rc = ork__sb_chunk(&sb, chunk, clen, (n - (double)clen[0]) / 48000))) goto done; /* :2422 */
    }
done:
    for (int j = 0; j < nd; j++) { free(d[j].val); free(d[j].has); free(chunk[j]); }
    if (rc) { ork__sb_abort(&sb); return rc; }
    return ork__sb_finish(&sb, out, len / sample_rate, final_status);
}

/* ======================= aukit.stream.dfpwm  aukit.lua:2439-2496 ======================= */
int ork_stream_dfpwm(const uint8_t *data, size_t nbytes, double sample_rate, int channels, int mono, int interp, ork_stream *out) {
    if (sample_rate < 1) return ork__fail(ORK_E_ARG, "bad argument #2 (number outside of range)");
    if (channels < 1 || channels > ORK_MAX_CH) return ork__fail(ORK_E_ARG, "bad argument #3 (number outside of range)");
    if (channels == 1) mono = 0;
    ork_dfpwm_dec dec;
    ork_dfpwm_dec_init(&dec);
    size_t slice = 6000 * (size_t)channels + 1; /* str_sub(data, pos, pos + 6000 * channels) */
    int8_t *tmp = (int8_t *)malloc(slice * 8);
    double *audio = (double *)malloc((slice * 8 + 1) * sizeof(double)); /* audio[0..#audio] */
    ork_sbuild sb;
    memset(&sb, 0, sizeof sb);
    sb.channels = mono ? 1 : channels;
    double *lines[ORK_MAX_CH] = {0};
    double ratio = 48000 / sample_rate;
    size_t maxout = (size_t)((double)(slice * 8) * ratio) + 8;
    int rc = ORK_OK;
    for (int j = 0; j < sb.channels; j++) lines[j] = (double *)malloc(maxout * sizeof(double));
    if (!tmp || !audio) { rc = ork__fail(ORK_E_NOMEM, "out of memory"); goto done; }
    double last = 0;
    for (size_t pos = 0; pos < nbytes; pos += 6000 * (size_t)channels) {
        size_t cnt = nbytes - pos < slice ? nbytes - pos : slice;
        ork_dfpwm_decode(&dec, data + pos, cnt, tmp);
        size_t na = cnt * 8;
        if (na == 0) break;
        for (size_t i = 0; i < na; i++) audio[i + 1] = tmp[i];
        audio[0] = last;          /* audio[0], last = last, audio[#audio]  :2470 */
        last = audio[na];
        double newlen = (double)na * ratio;
        ork_plain tab = {audio, 0, (long)na};
        size_t clen[ORK_MAX_CH] = {0};
        for (double i = 1; i <= newlen; i += channels) { /* :2478 */
            double nacc = 0;
            size_t oi = (size_t)ceil(i / channels);
            for (int j = 0; j < channels; j++) {
                double x = (i - 1) / ratio + 1; /* Q11: does not depend on j */
                double s;
                if (x == floor(x)) {
                    if (!ork__plain_get(&tab, (long)x, &s)) { rc = ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value"); goto done; }
                } else {
                    if ((rc = ork__interp_get(interp, ork__plain_get, &tab, x, &s))) goto done;
                    s = ork_clamp(s, -128, 127);
                }
                if (mono) nacc = nacc + s;
                else { lines[j][oi - 1] = s; clen[j] = oi; }
            }
            if (mono) { lines[0][oi - 1] = nacc / channels; clen[0] = oi; }
        }
        double p = (double)(pos + 1);
        if ((rc = ork__sb_chunk(&sb, lines, clen, p * 8 / sample_rate / channels))) goto done;
    }
done:
    free(tmp); free(audio);
    for (int j = 0; j < ORK_MAX_CH; j++) free(lines[j]);
    if (rc) { ork__sb_abort(&sb); return rc; }
    return ork__sb_finish(&sb, out, (double)nbytes * 8 / sample_rate / channels, ORK_OK);
}

/* ======================= aukit.stream.mdfpwm  aukit.lua:2507-2572 ======================= */
int ork_stream_mdfpwm(const uint8_t *data, size_t nbytes, int mono, ork_stream *out) {
    if (nbytes < 7 || memcmp(data, "MDFPWM\3", 7) != 0) return ork__fail(ORK_E_ARG, "bad argument #1 (invalid MDFPWM data)");
    size_t hp = 7;
    if (hp + 4 > nbytes) return ork__fail(ORK_E_LUA, "data string too short");
    double length = (double)(data[hp] | data[hp + 1] << 8 | data[hp + 2] << 16 | (uint32_t)data[hp + 3] << 24);
    hp += 4;
    for (int k = 0; k < 3; k++) {
        if (hp + 1 > nbytes) return ork__fail(ORK_E_LUA, "data string too short");
        hp += 1 + (size_t)data[hp];
        if (hp > nbytes) return ork__fail(ORK_E_LUA, "data string too short");
    }
    double headerSize = (double)hp; /* pos - 1 */
    ork_dfpwm_dec dl, dr;
    ork_dfpwm_dec_init(&dl);
    ork_dfpwm_dec_init(&dr);
    int8_t *tl = (int8_t *)malloc(48000), *tr = (int8_t *)malloc(48000);
    double *lines[ORK_MAX_CH] = {0};
    lines[0] = (double *)malloc(48000 * sizeof(double));
    lines[1] = (double *)malloc(48000 * sizeof(double));
    ork_sbuild sb;
    memset(&sb, 0, sizeof sb);
    sb.channels = mono ? 1 : 2;
    int rc = ORK_OK, final_status = ORK_OK;
    if (!tl || !tr || !lines[0] || !lines[1]) { rc = ork__fail(ORK_E_NOMEM, "out of memory"); goto done; }
    for (size_t pos = hp; pos < nbytes;) { /* pos here is 0-based; Lua pos = pos + 1 */
        size_t nl = nbytes - pos < 6000 ? nbytes - pos : 6000;
        size_t nr = pos + 6000 < nbytes ? (nbytes - pos - 6000 < 6000 ? nbytes - pos - 6000 : 6000) : 0;
        ork_dfpwm_decode(&dl, data + pos, nl, tl);
        if (nl == 0) break;
        if (nr == 0) break; /* decoderR("") → #audioR == 0 → return nil (decoderL state already advanced) */
        ork_dfpwm_decode(&dr, data + pos + 6000, nr, tr);
        double lpos = (double)(pos + 1);
        int trimmed = (lpos - headerSize + 12000 > length); /* :2553 */
        size_t clen[ORK_MAX_CH] = {0};
        if (trimmed || nl < 6000 || nr < 6000) {
            /* Q12: samples (length/2)%6000+1 .. 6000 are set to nil; a short final block leaves nils below 48000 */
            if (mono) { final_status = ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value (field '?')"); break; }
            ork__fail(ORK_E_UNSUPPORTED, "stream.mdfpwm final chunk: channel tables with holes (length of the table is ill-defined)");
            final_status = ORK_E_UNSUPPORTED;
            break;
        }
        if (mono) {
            for (int i = 0; i < 48000; i++) lines[0][i] = ork_clamp(floor((double)tl[i] + (double)tr[i] / 2), -128, 127); /* :2563 */
            clen[0] = 48000;
        } else {
            for (int i = 0; i < 48000; i++) { lines[0][i] = tl[i]; lines[1][i] = tr[i]; }
            clen[0] = clen[1] = 48000;
        }
        double p = lpos - headerSize;
        if ((rc = ork__sb_chunk(&sb, lines, clen, p / 12000))) goto done;
        pos += nl + nr;
    }
done:
    free(tl); free(tr); free(lines[0]); free(lines[1]);
    if (rc) { ork__sb_abort(&sb); return rc; }
    return ork__sb_finish(&sb, out, length / 12000, final_status);
}

/* ======================= aukit.stream.msadpcm  aukit.lua:2588-2736 ======================= */
static const int ms_coef1_default[7] = {256, 512, 0, 192, 240, 460, 392};
static const int ms_coef2_default[7] = {0, -256, 0, 64, 0, -208, -232};
static inline int rd_i16(const uint8_t *p) { return (int16_t)(p[0] | p[1] << 8); }
static inline double ms_step(double *sample1, double *sample2, double *delta, double c1, double c2, int nib) {
    double predictor = ork_clamp(floor((*sample1 * c1 + *sample2 * c2) / 256) + nib * *delta, -32768, 32767);
    *sample2 = *sample1;
    *sample1 = predictor;
    double nd = floor(ork__msadpcm_adapt(nib) * *delta / 256);
    *delta = nd < 16 ? 16 : nd;
    return predictor;
}
/* table with the current block at 1..n, nil at 0, and the previous block at -m..-1 (aukit.lua:2642-2645) */
typedef struct { const double *cur; long n; const double *prev; long m; } hist_tab;
static int hist_get(void *ctx, long idx, double *v) {
    const hist_tab *t = (const hist_tab *)ctx;
    if (idx >= 1 && idx <= t->n) { *v = t->cur[idx - 1]; return 1; }
    if (idx <= -1 && idx >= -t->m && t->prev) { *v = t->prev[idx + t->m]; return 1; } /* left[i-#last-1] = last[i] */
    return 0;
}

int ork_stream_msadpcm(const uint8_t *data, size_t nbytes, int block_align, int channels, double sample_rate, int mono,
                       const int *coef1, const int *coef2, int ncoef, int interp, ork_stream *out) {
    if (sample_rate < 1) return ork__fail(ORK_E_ARG, "bad argument #4 (number outside of range)");
    if (!coef1 || !coef2) { coef1 = ms_coef1_default; coef2 = ms_coef2_default; ncoef = 7; }
    if (channels != 1 && channels != 2) return ork__fail(ORK_E_LUA, "Unsupported number of channels: %d", channels);
    if (block_align < (channels == 2 ? 15 : 8)) return ork__fail(ORK_E_ARG, "blockAlign too small");
    double ratio = 48000 / sample_rate;
    double samplesPerBlock = channels == 2 ? block_align - 14 : (block_align - 7) * 2; /* :2617 / :2682 */
    double iterPerSecond = ceil(sample_rate / samplesPerBlock);
    double bytesPerSecond = block_align * iterPerSecond;
    long newlen = (long)floor(samplesPerBlock * ratio);
    long blk_samples = channels == 2 ? (block_align - 14) + 2 : (block_align - 7) * 2 + 2;
    ork_sbuild sb;
    memset(&sb, 0, sizeof sb);
    sb.channels = (channels == 2 && !mono) ? 2 : 1;
    double *left = (double *)malloc((size_t)blk_samples * sizeof(double)), *right = (double *)malloc((size_t)blk_samples * sizeof(double));
    double *lastL = (double *)malloc((size_t)blk_samples * sizeof(double)), *lastR = (double *)malloc((size_t)blk_samples * sizeof(double));
    long nlast = 0;
    int have_last = 0;
    size_t outcap = (size_t)((iterPerSecond + 1) * (double)(newlen > 0 ? newlen : 0)) + 16;
    double *ret[ORK_MAX_CH] = {0};
    ret[0] = (double *)malloc(outcap * sizeof(double));
    ret[1] = (double *)malloc(outcap * sizeof(double));
    int rc = ORK_OK, final_status = ORK_OK;
    if (!left || !right || !lastL || !lastR || !ret[0] || !ret[1]) { rc = ork__fail(ORK_E_NOMEM, "out of memory"); goto done; }
    double n = 1; /* 1-based like the Lua */
    for (;;) {
        double target = n + bytesPerSecond;
        size_t rp = 0;
        int raised = 0;
        while (n < target) {
            if (n > (double)nbytes) break;
            size_t b0 = (size_t)n - 1;
            long cnt = 0;
            if (channels == 2) {
                if (b0 + 14 > nbytes) { raised = 1; ork__fail(ORK_E_LUA, "data string too short"); break; }
                int piL = data[b0], piR = data[b0 + 1];
                double deltaL = rd_i16(data + b0 + 2), deltaR = rd_i16(data + b0 + 4);
                double s1L = rd_i16(data + b0 + 6), s1R = rd_i16(data + b0 + 8);
                double s2L = rd_i16(data + b0 + 10), s2R = rd_i16(data + b0 + 12);
                if (piL >= ncoef || piR >= ncoef) { raised = 1; ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value (local 'c1L')"); break; }
                double c1L = coef1[piL], c2L = coef2[piL], c1R = coef1[piR], c2R = coef2[piR];
                left[0] = floor(s2L / (s2L < 0 ? 128 : 127));  /* :2648-2651 (floored in the stereo path only, Q9) */
                left[1] = floor(s1L / (s1L < 0 ? 128 : 127));
                right[0] = floor(s2R / (s2R < 0 ? 128 : 127));
                right[1] = floor(s1R / (s1R < 0 ? 128 : 127));
                cnt = 2;
                for (int i = 14; i <= block_align - 1; i++) {
                    if (b0 + (size_t)i >= nbytes) { raised = 1; ork__fail(ORK_E_LUA, "bad argument #1 to 'rshift' (number expected, got nil)"); break; }
                    int b = data[b0 + i], hi = b >> 4, lo = b & 0x0F;
                    if (hi >= 8) hi -= 16;
                    if (lo >= 8) lo -= 16;
                    double p = ms_step(&s1L, &s2L, &deltaL, c1L, c2L, hi);
                    left[cnt] = floor(p / (p < 0 ? 128 : 127));
                    p = ms_step(&s1R, &s2R, &deltaR, c1R, c2R, lo);
                    right[cnt] = floor(p / (p < 0 ? 128 : 127));
                    cnt++;
                }
                if (raised) break;
                hist_tab tl = {left, cnt, have_last ? lastL : NULL, nlast}, tr2 = {right, cnt, have_last ? lastR : NULL, nlast};
                for (long i = 1; i <= newlen; i++) { /* :2667-2674 */
                    double x = ((double)i - 1) / ratio + 1;
                    double l, r;
                    if (x == floor(x)) {
                        if (!hist_get(&tl, (long)x, &l) || !hist_get(&tr2, (long)x, &r)) { raised = 1; ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value"); break; }
                    } else {
                        if (ork__interp_get(interp, hist_get, &tl, x, &l) || ork__interp_get(interp, hist_get, &tr2, x, &r)) { raised = 1; break; }
                    }
                    if (mono) ret[0][rp + (size_t)i - 1] = ork_clamp(floor(l + r / 2), -128, 127); /* l + r/2 (precedence, Q9) */
                    else { ret[0][rp + (size_t)i - 1] = ork_clamp(floor(l), -128, 127); ret[1][rp + (size_t)i - 1] = ork_clamp(floor(r), -128, 127); }
                }
                if (raised) break;
                memcpy(lastL, left, (size_t)cnt * sizeof(double));
                memcpy(lastR, right, (size_t)cnt * sizeof(double));
                nlast = cnt;
                have_last = 1;
            } else {
                if (nbytes < 7) { raised = 1; ork__fail(ORK_E_LUA, "data string too short"); break; }
                int pi = data[0]; /* header always read at offset 1 (Q9) :2706 */
                double delta = rd_i16(data + 1), s1 = rd_i16(data + 3), s2 = rd_i16(data + 5);
                if (pi >= ncoef) { raised = 1; ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value (local 'c1')"); break; }
                double c1 = coef1[pi], c2 = coef2[pi];
                left[0] = s2 / (s2 < 0 ? 128 : 127); /* not floored in the mono path */
                left[1] = s1 / (s1 < 0 ? 128 : 127);
                cnt = 2;
                for (int i = 7; i <= block_align - 1; i++) {
                    if (b0 + (size_t)i >= nbytes) { raised = 1; ork__fail(ORK_E_LUA, "bad argument #1 to 'rshift' (number expected, got nil)"); break; }
                    int b = data[b0 + i], hi = b >> 4, lo = b & 0x0F;
                    if (hi >= 8) hi -= 16;
                    if (lo >= 8) lo -= 16;
                    double p = ms_step(&s1, &s2, &delta, c1, c2, hi);
                    left[cnt++] = p / (p < 0 ? 128 : 127);
                    p = ms_step(&s1, &s2, &delta, c1, c2, lo);
                    left[cnt++] = p / (p < 0 ? 128 : 127);
                }
                if (raised) break;
                ork_plain tl = {left, 1, cnt};
                for (long i = 1; i <= newlen; i++) { /* :2724-2728 */
                    double x = ((double)i - 1) / ratio + 1;
                    double v;
                    if (x == floor(x)) {
                        if (!ork__plain_get(&tl, (long)x, &v)) { raised = 1; ork__fail(ORK_E_LUA, "bad argument #1 to 'floor' (number expected, got nil)"); break; }
                    } else if (ork__interp_get(interp, ork__plain_get, &tl, x, &v)) { raised = 1; break; }
                    ret[0][rp + (size_t)i - 1] = ork_clamp(floor(v), -128, 127);
                }
                if (raised) break;
            }
            rp += (size_t)(newlen > 0 ? newlen : 0);
            n = n + block_align;
        }
        if (raised) { final_status = ORK_E_LUA; break; }
        if (rp == 0) break; /* #retval[1] == 0 → nil */
        size_t clen[ORK_MAX_CH] = {rp, rp};
        if ((rc = ork__sb_chunk(&sb, ret, clen, (n + 0) / bytesPerSecond))) goto done;
    }
done:
    free(left); free(right); free(lastL); free(lastR); free(ret[0]); free(ret[1]);
    if (rc) { ork__sb_abort(&sb); return rc; }
    return ork__sb_finish(&sb, out, (double)nbytes / block_align * samplesPerBlock / sample_rate, final_status);
}

/* ======================= aukit.stream.adpcm  aukit.lua:2753-2835 ======================= */
int ork_stream_adpcm(const uint8_t *data, size_t nbytes, int block_align, int channels, double sample_rate, int mono,
                     int interp, ork_stream *out) {
    if (sample_rate < 1) return ork__fail(ORK_E_ARG, "bad argument #4 (number outside of range)");
    if (channels < 1 || channels > ORK_MAX_CH) return ork__fail(ORK_E_ARG, "channels out of range");
    if (block_align <= 4 * channels) return ork__fail(ORK_E_ARG, "blockAlign too small");
    double ratio = 48000 / sample_rate;
    double samplesPerBlock = (double)(block_align - 4 * channels) * 2 / channels; /* :2765 */
    double iterPerSecond = ceil(sample_rate / samplesPerBlock);
    double bytesPerSecond = block_align * iterPerSecond;
    long newlen = (long)floor(samplesPerBlock * ratio);
    long dcap = (long)((double)block_align * 2 / channels) + 32;
    ork_sbuild sb;
    memset(&sb, 0, sizeof sb);
    sb.channels = mono ? 1 : channels;
    double *d[ORK_MAX_CH] = {0}, *ret[ORK_MAX_CH] = {0};
    size_t outcap = (size_t)((iterPerSecond + 1) * (double)(newlen > 0 ? newlen : 0)) + 16;
    int rc = ORK_OK, final_status = ORK_OK;
    for (int j = 0; j < channels; j++) {
        d[j] = (double *)malloc((size_t)dcap * sizeof(double));
        ret[j] = (double *)malloc(outcap * sizeof(double));
        if (!d[j] || !ret[j]) { rc = ork__fail(ORK_E_NOMEM, "out of memory"); goto done; }
    }
    double n = 1;
    for (;;) {
        double target = n + bytesPerSecond;
        size_t rp = 0;
        int raised = 0;
        while (n < target) {
            if (n + channels * 4 > (double)nbytes) break; /* :2794 */
            size_t b0 = (size_t)n - 1;
            int predictor[ORK_MAX_CH], step_index[ORK_MAX_CH];
            long dl[ORK_MAX_CH] = {0};
            for (int i = 0; i < channels; i++) { /* "<hB" :2799 — step index used unmasked (Q6) */
                predictor[i] = rd_i16(data + b0 + (size_t)i * 4);
                step_index[i] = data[b0 + (size_t)i * 4 + 2];
            }
            for (int i = channels * 4; i <= block_align; i += channels * 4) { /* inclusive upper bound: junk word (Q6) :2800 */
                long p = (long)((double)(i - channels * 4) / channels * 2 + 1);
                if ((double)nbytes < n + i + channels * 4) break; /* :2802 */
                for (int j = 0; j < channels; j++) {
                    const uint8_t *w = data + b0 + (size_t)i + (size_t)j * 4;
                    uint32_t num = (uint32_t)w[0] | (uint32_t)w[1] << 8 | (uint32_t)w[2] << 16 | (uint32_t)w[3] << 24;
                    for (int k = 0; k <= 7; k++) {
                        int nibble = (num >> (k * 4)) & 15;
                        if (step_index[j] > 88) { raised = 1; ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value (field '?')"); goto block_done; }
                        int step = ork__ima_step_table[step_index[j]];
                        step_index[j] = (int)ork_clamp(step_index[j] + ork__ima_index_table[nibble], 0, 88);
                        int diff = (((nibble % 8) * step) >> 2) + (step >> 3);
                        if (nibble >= 8) predictor[j] = (int)ork_clamp(predictor[j] - diff, -32768, 32767);
                        else predictor[j] = (int)ork_clamp(predictor[j] + diff, -32768, 32767);
                        d[j][p + k - 1] = (double)predictor[j] / (predictor[j] < 0 ? 128 : 127); /* :2812 */
                        dl[j] = p + k;
                    }
                }
            }
            if ((double)dl[0] < samplesPerBlock) newlen = (long)floor((double)dl[0] * ratio); /* :2817, sticky */
            for (long i = 1; i <= newlen; i++) { /* :2818-2828 */
                double x = ((double)i - 1) / ratio + 1;
                double c[ORK_MAX_CH];
                for (int j = 0; j < channels; j++) {
                    ork_plain t = {d[j], 1, dl[j]};
                    if (x == floor(x)) {
                        if (!ork__plain_get(&t, (long)x, &c[j])) { raised = 1; ork__fail(ORK_E_LUA, "bad argument #1 to 'floor' (number expected, got nil)"); goto block_done; }
                    } else if (ork__interp_get(interp, ork__plain_get, &t, x, &c[j])) { raised = 1; goto block_done; }
                }
                if (mono) {
                    double acc = 0;
                    for (int j = 0; j < channels; j++) acc = acc + c[j];
                    ret[0][rp + (size_t)i - 1] = ork_clamp(floor(acc / channels), -128, 127);
                } else
                    for (int j = 0; j < channels; j++) ret[j][rp + (size_t)i - 1] = ork_clamp(floor(c[j]), -128, 127);
            }
            rp += (size_t)(newlen > 0 ? newlen : 0);
            n = n + block_align;
        }
    block_done:
        if (raised) { final_status = ORK_E_LUA; break; }
        if (rp == 0) break; /* :2832 */
        size_t clen[ORK_MAX_CH];
        for (int j = 0; j < ORK_MAX_CH; j++) clen[j] = rp;
        if ((rc = ork__sb_chunk(&sb, ret, clen, n / bytesPerSecond))) goto done; /* (n + pos) / bytesPerSecond, pos = 0 */
    }
done:
    for (int j = 0; j < ORK_MAX_CH; j++) { free(d[j]); free(ret[j]); }
    if (rc) { ork__sb_abort(&sb); return rc; }
    return ork__sb_finish(&sb, out, (double)nbytes / block_align * samplesPerBlock / sample_rate, final_status);
}

/* ======================= aukit.stream.g711  aukit.lua:2850-2913 ======================= */
int ork_stream_g711(const uint8_t *data, size_t nbytes, int ulaw, int channels, double sample_rate, int mono,
                    int interp, int max_calls, ork_stream *out) {
    if (channels < 1 || channels > ORK_MAX_CH) return ork__fail(ORK_E_ARG, "channels out of range");
    if (sample_rate != floor(sample_rate) || sample_rate < 1) return ork__fail(ORK_E_UNSUPPORTED, "non-integer sample rate");
    double ratio = 48000 / sample_rate;
    size_t per_call = (size_t)sample_rate * (size_t)channels;
    ork_sbuild sb;
    memset(&sb, 0, sizeof sb);
    sb.channels = mono ? 1 : channels;
    double *retval[ORK_MAX_CH] = {0}, *resamp[ORK_MAX_CH] = {0};
    size_t outcap = (size_t)(sample_rate * ratio) + 16;
    int rc = ORK_OK, final_status = ORK_OK;
    for (int j = 0; j < channels; j++) {
        retval[j] = (double *)malloc(((size_t)sample_rate + 1) * sizeof(double));
        resamp[j] = (double *)malloc(outcap * sizeof(double));
        if (!retval[j] || !resamp[j]) { rc = ork__fail(ORK_E_NOMEM, "out of memory"); goto done; }
    }
    size_t pos = 0; /* 0-based; Lua pos = pos + 1 */
    for (int call = 0; call < max_calls; call++) {
        double lp = (double)(pos + 1);
        size_t cnt = pos < nbytes ? (nbytes - pos < per_call ? nbytes - pos : per_call) : 0;
        size_t rl[ORK_MAX_CH] = {0};
        for (size_t q = 0; q < cnt; q++) { /* :2883-2892 */
            int neg, m = ork__g711_expand(data[pos + q], ulaw, &neg);
            retval[q % channels][q / channels] = (double)m / (neg ? -0x40 : 0x40);
            rl[q % channels] = q / channels + 1;
        }
        pos += per_call;
        long newlen = (long)floor((double)rl[0] * ratio); /* :2897 */
        int raised = 0;
        for (long i = 1; i <= newlen && !raised; i++) {
            double x = ((double)i - 1) / ratio + 1;
            double c[ORK_MAX_CH];
            for (int j = 0; j < channels; j++) {
                ork_plain t = {retval[j], 1, (long)rl[j]};
                if (x == floor(x)) {
                    if (!ork__plain_get(&t, (long)x, &c[j])) { raised = 1; ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value"); break; }
                } else if (ork__interp_get(interp, ork__plain_get, &t, x, &c[j])) { raised = 1; break; }
            }
            if (raised) break;
            if (mono) {
                double acc = 0;
                for (int j = 0; j < channels; j++) acc = acc + c[j];
                resamp[0][i - 1] = ork_clamp(floor(acc / channels), -128, 127);
            } else
                for (int j = 0; j < channels; j++) resamp[j][i - 1] = ork_clamp(floor(c[j]), -128, 127);
        }
        if (raised) { final_status = ORK_E_LUA; break; }
        size_t clen[ORK_MAX_CH];
        for (int j = 0; j < ORK_MAX_CH; j++) clen[j] = (size_t)(newlen > 0 ? newlen : 0);
        if ((rc = ork__sb_chunk(&sb, resamp, clen, (lp - 1) / sample_rate / channels))) goto done;
    }
done:
    for (int j = 0; j < ORK_MAX_CH; j++) { free(retval[j]); free(resamp[j]); }
    if (rc) { ork__sb_abort(&sb); return rc; }
    return ork__sb_finish(&sb, out, (double)nbytes / sample_rate / channels, final_status);
}

/* ======================= aukit.stream.flac  aukit.lua:3124-3191 ======================= */
typedef struct { const double *src; long n; double m1, z0; } flac_src;
static int flac_src_get(void *ctx, long idx, double *v) {
    const flac_src *t = (const flac_src *)ctx;
    if (idx >= 1 && idx <= t->n) { *v = t->src[idx - 1]; return 1; }
    if (idx == 0) { *v = t->z0; return 1; }
    if (idx == -1) { *v = t->m1; return 1; }
    return 0;
}

int ork_stream_flac(const uint8_t *data, size_t nbytes, int interp, ork_stream *out) {
    ork_flac_dec *dec;
    double sampleRate, len;
    int nch, depth;
    int rc = ork__flac_open(data, nbytes, &dec, &sampleRate, &nch, &depth, &len);
    if (rc) return rc; /* error(sampleRate, 2) :3152 */
    double ratio = 48000 / sampleRate;
    double lp_alpha = 1 - exp(-(sampleRate / 96000) * 2 * M_PI);
    double last[2] = {0, 0};
    double pos = 0;
    ork_sbuild sb;
    memset(&sb, 0, sizeof sb);
    sb.channels = nch;
    ork_vec chunk[ORK_MAX_CH];
    memset(chunk, 0, sizeof chunk);
    int dead = 0;
    while (!dead) { /* one iterator call per pass */
        for (int c = 0; c < nch; c++) chunk[c].n = 0;
        while ((double)chunk[0].n < sampleRate) {
            double *res[ORK_MAX_CH];
            size_t bs;
            int r = ork__flac_frame(dec, res, &bs);
            if (r <= 0) { dead = 1; break; } /* not ok (error swallowed) or res == nil: coroutine is dead afterwards */
            for (int c = 0; c < nch; c++) {
                flac_src src = {res[c], (long)bs, last[0], last[1]}; /* src[0] = last[2]; src[-1] = last[1] */
                double ls = last[1] / (last[1] < 0 ? 128 : 127);
                long cnt = (long)floor((double)bs * ratio);
                int bad = 0;
                for (long i = 1; i <= cnt; i++) {
                    double x = (((double)i - 1) / ratio) + 1;
                    double s;
                    if (x == floor(x)) { if (!flac_src_get(&src, (long)x, &s)) { bad = 1; break; } }
                    else if (ork__interp_get(interp, flac_src_get, &src, x, &s)) { bad = 1; break; }
                    s = ls + lp_alpha * (s - ls);
                    ls = s; /* recursive (Q14) */
                    if (ork__vec_push(&chunk[c], ork_clamp(s * (s < 0 ? 128 : 127), -128, 127))) { rc = ork__fail(ORK_E_NOMEM, "out of memory"); }
                }
                if (!bad) {
                    double a = 0, b = 0;
                    flac_src_get(&src, (long)bs - 1, &a);
                    flac_src_get(&src, (long)bs, &b);
                    last[0] = a; last[1] = b; /* shared across channels (Q14) */
                }
            }
            for (int c = 0; c < nch; c++) free(res[c]);
            if (rc) goto done;
        }
        pos = pos + (double)chunk[0].n / 48000; /* :3188 */
        double *cp[ORK_MAX_CH];
        size_t clen[ORK_MAX_CH] = {0};
        for (int c = 0; c < nch; c++) { cp[c] = chunk[c].p; clen[c] = chunk[c].n; }
        if ((rc = ork__sb_chunk(&sb, cp, clen, pos))) goto done;
    }
done:
    for (int c = 0; c < ORK_MAX_CH; c++) free(chunk[c].p);
    ork__flac_close(dec);
    if (rc) { ork__sb_abort(&sb); return rc; }
    return ork__sb_finish(&sb, out, len / sampleRate, ORK_OK);
}

/* ======================= aukit.stream.qoa  aukit.lua:3202-3337 ======================= */
typedef struct { const double *p; long n; double m1, z0; } qoa_tab; /* chunk[i] = {[-1]=..., [0]=..., 1..n} */
static int qoa_tab_get(void *ctx, long idx, double *v) {
    const qoa_tab *t = (const qoa_tab *)ctx;
    if (idx >= 1 && idx <= t->n) { *v = t->p[idx - 1]; return 1; }
    if (idx == 0) { *v = t->z0; return 1; }
    if (idx == -1) { *v = t->m1; return 1; }
    return 0;
}
static inline uint32_t rd_be32(const uint8_t *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
static inline int rd_be16s(const uint8_t *p) { return (int16_t)(p[0] << 8 | p[1]); }

int ork_stream_qoa(const uint8_t *data, size_t nbytes, int mono, int interp, ork_stream *out) {
    if (nbytes < 8) return ork__fail(ORK_E_LUA, "Not a QOA file");
    if (memcmp(data, "qoaf", 4) != 0) return ork__fail(ORK_E_ARG, "Not a QOA file");
    double file_samples = rd_be32(data + 4);
    size_t pos = 8;
    if (pos >= nbytes) return ork__fail(ORK_E_LUA, "Not a QOA file"); /* assert(peek(4)) */
    if (pos + 4 > nbytes) return ork__fail(ORK_E_LUA, "data string too short");
    int file_channels = data[pos];
    double file_rate = (double)((uint32_t)data[pos + 1] << 16 | (uint32_t)data[pos + 2] << 8 | data[pos + 3]);
    if (file_channels < 1 || file_channels > ORK_MAX_CH) return ork__fail(ORK_E_UNSUPPORTED, "QOA channel count");
    ork_qoa_lms lms[ORK_MAX_CH];
    memset(lms, 0, sizeof lms);
    double last[ORK_MAX_CH][2];
    memset(last, 0, sizeof last);
    double file_pos = 0;
    double ratio = 48000 / file_rate;
    double lp_alpha = 1 - exp(-(file_rate / 96000) * 2 * M_PI);
    ork_sbuild sb;
    memset(&sb, 0, sizeof sb);
    sb.channels = mono ? 1 : file_channels;
    ork_vec chunk[ORK_MAX_CH], lines[ORK_MAX_CH];
    memset(chunk, 0, sizeof chunk);
    memset(lines, 0, sizeof lines);
    int rc = ORK_OK, final_status = ORK_OK;
    for (;;) { /* one iterator call per pass */
        for (int c = 0; c < file_channels; c++) chunk[c].n = 0;
        double sample_pos = 0;
        int raised = 0;
        while (sample_pos < file_rate) {
            if (pos >= nbytes) break; /* read(8) → nil */
            if (pos + 8 > nbytes) { raised = 1; ork__fail(ORK_E_LUA, "data string too short"); break; }
            int channels = data[pos];
            double samplerate = (double)((uint32_t)data[pos + 1] << 16 | (uint32_t)data[pos + 2] << 8 | data[pos + 3]);
            int samples = data[pos + 4] << 8 | data[pos + 5];
            int frame_size = data[pos + 6] << 8 | data[pos + 7];
            pos += 8;
            int data_size = frame_size - 8 - 4 * 4 * channels;
            int num_slices = (int)floor((double)data_size / 8);
            int max_total_samples = num_slices * 20;
            if (channels != file_channels || samplerate != file_rate || samples * channels > max_total_samples) break; /* :3270-3277 */
            for (int c = 0; c < channels && !raised; c++) {
                if (pos >= nbytes) { raised = 1; ork__fail(ORK_E_LUA, "Invalid QOA data"); break; }
                if (pos + 8 > nbytes) { raised = 1; ork__fail(ORK_E_LUA, "data string too short"); break; }
                for (int k = 0; k < 4; k++) lms[c].history[k] = rd_be16s(data + pos + 2 * k);
                pos += 8;
                if (pos >= nbytes) { raised = 1; ork__fail(ORK_E_LUA, "Invalid QOA data"); break; }
                if (pos + 8 > nbytes) { raised = 1; ork__fail(ORK_E_LUA, "data string too short"); break; }
                for (int k = 0; k < 4; k++) lms[c].weights[k] = rd_be16s(data + pos + 2 * k);
                pos += 8;
            }
            for (int sample_index = 1; sample_index <= samples && !raised; sample_index += 20) {
                for (int c = 0; c < channels; c++) {
                    if (pos >= nbytes) { raised = 1; ork__fail(ORK_E_LUA, "Invalid QOA data"); break; }
                    if (pos + 8 > nbytes) { raised = 1; ork__fail(ORK_E_LUA, "data string too short"); break; }
                    uint32_t sliceH = rd_be32(data + pos), sliceL = rd_be32(data + pos + 4);
                    pos += 8;
                    int scalefactor = (sliceH >> 28) & 15;
                    for (int si = sample_index; si <= sample_index + 19; si++) {
                        double predicted = ork__qoa_predict(&lms[c]);
                        int quantized = (sliceH >> 25) & 7;
                        double dequantized = ork__qoa_dequant_tab[scalefactor][quantized];
                        double reconstructed = fmin(fmax(predicted + dequantized, -32768), 32767);
                        if (ork__vec_set(&chunk[c], (size_t)(sample_pos + si) - 1, floor(reconstructed / 256))) { rc = ork__fail(ORK_E_NOMEM, "out of memory"); goto done; } /* :3299 */
                        sliceH = (sliceH << 3) + ((sliceL >> 29) & 7);
                        sliceL = sliceL << 3;
                        ork__qoa_update(&lms[c], reconstructed, dequantized);
                    }
                }
            }
            if (raised) break;
            sample_pos = sample_pos + samples;
        }
        if (raised) { final_status = ORK_E_LUA; break; }
        if (chunk[0].n == 0) break; /* :3310 */
        double newlen = (double)chunk[0].n * ratio;
        double ls[ORK_MAX_CH];
        for (int j = 0; j < file_channels; j++) { ls[j] = last[j][1]; lines[j].n = 0; }
        for (double i = 1; i <= newlen; i += 1) { /* :3317-3330 */
            double nacc = 0;
            for (int j = 0; j < file_channels; j++) {
                qoa_tab t = {chunk[j].p, (long)chunk[j].n, last[j][0], last[j][1]};
                double x = (i - 1) / ratio + 1;
                double s;
                if (x == floor(x)) {
                    if (!qoa_tab_get(&t, (long)x, &s)) { raised = 1; ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value"); break; }
                } else {
                    if (ork__interp_get(interp, qoa_tab_get, &t, x, &s)) { raised = 1; break; }
                    s = ork_clamp(s, -128, 127);
                }
                s = ls[j] + lp_alpha * (s - ls[j]);
                ls[j] = s;
                if (mono) nacc = nacc + s;
                else if (ork__vec_push(&lines[j], s)) { rc = ork__fail(ORK_E_NOMEM, "out of memory"); goto done; }
            }
            if (raised) break;
            if (mono && ork__vec_push(&lines[0], nacc / file_channels)) { rc = ork__fail(ORK_E_NOMEM, "out of memory"); goto done; }
        }
        if (raised) { final_status = ORK_E_LUA; break; }
        double p = file_pos / file_rate;
        file_pos = file_pos + sample_pos;
        for (int i = 0; i < file_channels; i++) {
            qoa_tab t = {chunk[i].p, (long)chunk[i].n, last[i][0], last[i][1]};
            double a = 0, b = 0;
            qoa_tab_get(&t, (long)chunk[i].n - 1, &a);
            qoa_tab_get(&t, (long)chunk[i].n, &b);
            last[i][0] = a; last[i][1] = b;
        }
        double *lp[ORK_MAX_CH];
        size_t clen[ORK_MAX_CH] = {0};
        for (int j = 0; j < sb.channels; j++) { lp[j] = lines[j].p; clen[j] = lines[j].n; }
        if ((rc = ork__sb_chunk(&sb, lp, clen, p))) goto done;
    }
done:
    for (int c = 0; c < ORK_MAX_CH; c++) { free(chunk[c].p); free(lines[c].p); }
    if (rc) { ork__sb_abort(&sb); return rc; }
    return ork__sb_finish(&sb, out, file_samples / file_rate, final_status);
}

/* ork_core.c — CPU ORACLE (test infrastructure only; see ork.h header).
 * Helpers, interpolation, Audio methods and aukit.effects.* restated from
 * aukit.lua (AUKit 1.10.0).  fp64, same operation order as the Lua.
 * Build with -ffp-contract=off: no fused multiply-adds may be introduced.
 */
#include "ork_internal.h"

static __thread char g_err[256];
int ork__sinc_window = 10; /* aukit.lua:129 */

const char *ork_last_error(void) { return g_err; }
int ork__fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
void ork_set_sinc_window(int w) { ork__sinc_window = w; }
void ork_free(void *p) { free(p); }

void ork_audio_free(ork_audio *a) {
    if (!a) return;
    for (int c = 0; c < ORK_MAX_CH; c++) { free(a->data[c]); a->data[c] = NULL; a->len[c] = 0; }
    a->channels = 0;
}
void ork_stream_free(ork_stream *s) {
    if (!s) return;
    for (int c = 0; c < ORK_MAX_CH; c++) { free(s->data[c]); s->data[c] = NULL; s->len[c] = 0; }
    free(s->chunk_len); s->chunk_len = NULL;
    free(s->chunk_pos); s->chunk_pos = NULL;
    s->nchunks = 0;
}

int ork__audio_alloc(ork_audio *a, int channels, size_t len, double rate) {
    memset(a, 0, sizeof *a);
    if (channels < 1 || channels > ORK_MAX_CH) return ork__fail(ORK_E_ARG, "channels out of range");
    a->channels = channels;
    a->sample_rate = rate;
    for (int c = 0; c < channels; c++) {
        a->len[c] = len;
        a->data[c] = (double *)malloc((len ? len : 1) * sizeof(double));
        if (!a->data[c]) return ork__fail(ORK_E_NOMEM, "out of memory");
    }
    return ORK_OK;
}

/* aukit.lua:228-232 */
double ork_clamp(double n, double mn, double mx) {
    if (n < mn) return mn;
    else if (n > mx) return mx;
    else return n;
}

/* ------------------------------------------------------------------------
 * interpolate.{none,linear,cubic,sinc}  aukit.lua:253-282
 * `get(ctx, idx, &v)` models `data[idx]`: 1 = value, 0 = nil, <0 = the index
 * metamethod raised.  Reads happen in the same order as in the Lua so that
 * the lazy tables of stream.pcm (aukit.lua:2367-2371) see the same sequence.
 * ---------------------------------------------------------------------- */
int ork__interp_get(int mode, ork_getter get, void *ctx, double x, double *out) {
    int r;
    switch (mode) {
    case ORK_INTERP_NONE: { /* :254-256 */
        double v;
        r = get(ctx, (long)floor(x), &v);
        if (r < 0) return r;
        if (r == 0) return ork__fail(ORK_E_LUA, "interpolate.none returned nil");
        *out = v;
        return ORK_OK;
    }
    case ORK_INTERP_LINEAR: { /* :257-260 */
        double ffx = floor(x);
        long k = (long)ffx;
        double a, b;
        r = get(ctx, k, &a);
        if (r < 0) return r;
        if (r == 0) return ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value (field '?')");
        r = get(ctx, k + 1, &b);
        if (r < 0) return r;
        if (r == 0) b = a; /* data[ffx+1] or data[ffx] */
        *out = a + (b - a) * (x - ffx);
        return ORK_OK;
    }
    case ORK_INTERP_CUBIC: { /* :261-266 */
        double ffx = floor(x);
        long k = (long)ffx;
        double p0, p1, p2, p3, fx;
        int h0, h1, h2, h3;
        h0 = get(ctx, k - 1, &p0); if (h0 < 0) return h0;
        h1 = get(ctx, k, &p1);     if (h1 < 0) return h1;
        h2 = get(ctx, k + 1, &p2); if (h2 < 0) return h2;
        h3 = get(ctx, k + 2, &p3); if (h3 < 0) return h3;
        fx = x - ffx;
        if (!h1) return ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value (local 'p1')");
        /* p0, p2, p3 = p0 or p1, p2 or p1, p3 or p2 or p1  (right side uses the OLD p2) */
        {
            double np0 = h0 ? p0 : p1;
            double np2 = h2 ? p2 : p1;
            double np3 = h3 ? p3 : (h2 ? p2 : p1);
            p0 = np0; p2 = np2; p3 = np3;
        }
        /* fx^3 and fx^2 are Lua `^`, i.e. C pow() */
        *out = (-0.5 * p0 + 1.5 * p1 - 1.5 * p2 + 0.5 * p3) * pow(fx, 3) + (p0 - 2.5 * p1 + 2 * p2 - 0.5 * p3) * pow(fx, 2) +
               (-0.5 * p0 + 0.5 * p2) * fx + p1;
        return ORK_OK;
    }
    case ORK_INTERP_SINC: { /* :267-281 */
        double ffx = floor(x);
        long k = (long)ffx;
        double fx = x - ffx;
        double sum = 0;
        for (int n = -ork__sinc_window; n <= ork__sinc_window; n++) {
            double d;
            r = get(ctx, k + n, &d);
            if (r < 0) return r;
            if (r) {
                double px = M_PI * (fx - n);
                if (px == 0) sum = sum + d;
                else sum = sum + d * sin(px) / px;
            }
        }
        *out = sum;
        return ORK_OK;
    }
    }
    return ork__fail(ORK_E_ARG, "bad argument #2 (invalid interpolation type)");
}

int ork__plain_get(void *ctx, long idx, double *v) {
    const ork_plain *t = (const ork_plain *)ctx;
    if (idx < t->lo || idx > t->hi) return 0;
    *v = t->p[idx - t->lo];
    return 1;
}

int ork_interp(int mode, const double *data, size_t n, double x, double *out) {
    ork_plain t = {data, 1, (long)n};
    return ork__interp_get(mode, ork__plain_get, &t, x, out);
}

/* Audio:resample  aukit.lua:653-673 */
int ork_resample(const ork_audio *in, double new_rate, int interp, ork_audio *out) {
    if (interp < 0 || interp > 3) return ork__fail(ORK_E_ARG, "bad argument #2 (invalid interpolation type)");
    double ratio = new_rate / in->sample_rate;         /* :658 */
    double newlen = (double)in->len[0] * ratio;        /* :659 */
    size_t cnt = newlen >= 1 ? (size_t)floor(newlen) : 0; /* for i = 1, newlen */
    int rc = ork__audio_alloc(out, in->channels, cnt, new_rate);
    if (rc) return rc;
    for (int y = 0; y < in->channels; y++) {
        ork_plain c = {in->data[y], 1, (long)in->len[y]};
        double *line = out->data[y];
        for (size_t i = 1; i <= cnt; i++) {
            double x = ((double)i - 1) / ratio + 1;    /* :666 */
            if (x == floor(x)) {                       /* x % 1 == 0 */
                double v;
                if (!ork__plain_get(&c, (long)x, &v)) return ork__fail(ORK_E_LUA, "resample: c[x] is nil at i=%zu", i);
                line[i - 1] = v;                       /* :667 copied unclamped */
            } else {
                double v;
                rc = ork__interp_get(interp, ork__plain_get, &c, x, &v);
                if (rc) return rc;
                line[i - 1] = ork_clamp(v, -1, 1);     /* :668 */
            }
        }
    }
    return ORK_OK;
}

/* Audio:mono  aukit.lua:677-689 */
int ork_mono(const ork_audio *in, ork_audio *out) {
    int cn = in->channels;
    size_t n = in->len[0];
    int rc = ork__audio_alloc(out, 1, n, in->sample_rate);
    if (rc) return rc;
    for (size_t i = 0; i < n; i++) {
        double s = 0;
        for (int c = 0; c < cn; c++) {
            if (i >= in->len[c]) return ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value");
            s = s + in->data[c][i];
        }
        out->data[0][i] = s / cn;
    }
    return ORK_OK;
}

/* Audio:mix  aukit.lua:804-835 (all audios already at the same rate) */
int ork_mix(const ork_audio *const *audios, int n, double amplifier, ork_audio *out) {
    size_t len = audios[0]->len[0];
    int cn = audios[0]->channels;
    for (int a = 1; a < n; a++) {
        if (audios[a]->sample_rate != audios[0]->sample_rate) return ork__fail(ORK_E_UNSUPPORTED, "mix: resample first");
        if (audios[a]->len[0] > len) len = audios[a]->len[0];
        if (audios[a]->channels > cn) cn = audios[a]->channels;
    }
    int rc = ork__audio_alloc(out, cn, len, audios[0]->sample_rate);
    if (rc) return rc;
    for (int c = 0; c < cn; c++) {
        for (size_t i = 0; i < len; i++) {
            double s = 0;
            for (int a = 0; a < n; a++)
                if (c < audios[a]->channels) s = s + (i < audios[a]->len[c] ? audios[a]->data[c][i] : 0);
            out->data[c][i] = ork_clamp(s * amplifier, -1, 1);
        }
    }
    return ORK_OK;
}

/* encodePCM as used by Audio:pcm  aukit.lua:868-910 (info.multiple = nil) */
int ork_encode_pcm(const ork_audio *in, int bit_depth, int data_type, int interleaved, double **out, size_t *n) {
    if (bit_depth != 8 && bit_depth != 16 && bit_depth != 24 && bit_depth != 32) return ork__fail(ORK_E_ARG, "bad argument #2 (invalid bit depth)");
    if (data_type < 0 || data_type > 2) return ork__fail(ORK_E_ARG, "bad argument #3 (invalid data type)");
    if (data_type == ORK_FLOAT && bit_depth != 32) return ork__fail(ORK_E_ARG, "bad argument #2 (float audio must have 32-bit depth)");
    double maxValue = ldexp(1.0, bit_depth - 1);
    double add = data_type == ORK_UNSIGNED ? maxValue : 0;
    int nc = in->channels;
    size_t len = in->len[0];
    size_t total = (size_t)nc * len;
    double *data = (double *)malloc((total ? total : 1) * sizeof(double));
    if (!data) return ork__fail(ORK_E_NOMEM, "out of memory");
    for (int c = 0; c < nc; c++) {
        if (in->len[c] < len) { free(data); return ork__fail(ORK_E_LUA, "attempt to compare nil with number"); }
        for (size_t k = 0; k < len; k++) {
            double d = in->data[c][k];
            double e = data_type == ORK_FLOAT ? d : d * (d < 0 ? maxValue : maxValue - 1) + add; /* :874 */
            if (interleaved) data[k * nc + c] = e; /* :880 */
            else data[(size_t)c * len + k] = e;     /* :892 */
        }
    }
    *out = data;
    *n = (size_t)nc * len;
    return ORK_OK;
}

/* Audio:dfpwm  aukit.lua:1005-1018.  Both branches feed ONE encoder with one table. */
int ork_audio_dfpwm(const ork_audio *in, int interleaved, uint8_t **out, size_t *n) {
    double *pcm;
    size_t np;
    int rc = ork_encode_pcm(in, 8, ORK_SIGNED, interleaved, &pcm, &np);
    if (rc) return rc;
    size_t nb = (np + 7) / 8;
    uint8_t *bytes = (uint8_t *)malloc(nb ? nb : 1);
    if (!bytes) { free(pcm); return ork__fail(ORK_E_NOMEM, "out of memory"); }
    ork_dfpwm_enc e;
    ork_dfpwm_enc_init(&e);
    rc = ork_dfpwm_encode(&e, pcm, np, bytes);
    free(pcm);
    if (rc) { free(bytes); return rc; }
    *out = bytes;
    *n = nb;
    return ORK_OK;
}

/* ========================= aukit.effects.* ========================= */

/* aukit.lua:3356-3369 */
int ork_fx_amplify(ork_audio *a, double multiplier) {
    if (multiplier == 1) return ORK_OK;
    for (int c = 0; c < a->channels; c++)
        for (size_t i = 0; i < a->len[c]; i++) a->data[c][i] = ork_clamp(a->data[c][i] * multiplier, -1, 1);
    return ORK_OK;
}

/* aukit.lua:3376-3385 */
int ork_fx_speed(ork_audio *a, double multiplier, int default_interp) {
    if (multiplier == 1) return ORK_OK;
    double rate = a->sample_rate;
    a->sample_rate = a->sample_rate * multiplier;
    ork_audio nw;
    int rc = ork_resample(a, rate, default_interp, &nw);
    a->sample_rate = rate;
    if (rc) return rc;
    for (int c = 0; c < a->channels; c++) { free(a->data[c]); a->data[c] = nw.data[c]; a->len[c] = nw.len[c]; }
    return ORK_OK;
}

/* aukit.lua:3394-3412 */
int ork_fx_fade(ork_audio *a, double startTime, double startAmp, double endTime, double endAmp) {
    if (startAmp == 1 && endAmp == 1) return ORK_OK;
    for (int c = 0; c < a->channels; c++) {
        double *ch = a->data[c];
        double start = startTime * a->sample_rate;
        double m = (endAmp - startAmp) / ((endTime - startTime) * a->sample_rate);
        double limit = endTime * a->sample_rate;
        for (double i = start; i <= limit; i = i + 1) {
            if (i != floor(i) || i < 1 || i > (double)a->len[c])
                return ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value (field '?')"); /* ch[i] is nil (Q17) */
            size_t k = (size_t)i - 1;
            ch[k] = ork_clamp(ch[k] * (m * (i - start) + startAmp), -1, 1);
        }
    }
    return ORK_OK;
}

/* aukit.lua:3417-3424 */
int ork_fx_invert(ork_audio *a) {
    for (int c = 0; c < a->channels; c++)
        for (size_t i = 0; i < a->len[c]; i++) a->data[c][i] = -a->data[c][i];
    return ORK_OK;
}

/* math.max(a, b) in Lua: (a < b) ? b : a ... PUC-Lua: if (lua_compare(L, imax, i, LUA_OPLT)) imax = i  */
static double lua_max(double mx, double v) { return mx < v ? v : mx; }

/* aukit.lua:3431-3459 */
int ork_fx_normalize(ork_audio *a, double peak, int independent) {
    double mult = 0;
    if (!independent) {
        double max = 0;
        for (int c = 0; c < a->channels; c++)
            for (size_t i = 0; i < a->len[c]; i++) max = lua_max(max, fabs(a->data[c][i]));
        mult = peak / max;
    }
    for (int c = 0; c < a->channels; c++) {
        double *ch = a->data[c];
        if (independent) {
            double max = 0;
            for (size_t i = 0; i < a->len[c]; i++) max = lua_max(max, fabs(ch[i]));
            mult = peak / max;
        }
        for (size_t i = 0; i < a->len[c]; i++) ch[i] = ork_clamp(ch[i] * mult, -1, 1);
    }
    return ORK_OK;
}

/* aukit.lua:3464-3477 */
int ork_fx_center(ork_audio *a) {
    double sr = a->sample_rate;
    if (sr != floor(sr) || sr < 1) return ork__fail(ORK_E_LUA, "center: non-integer sample rate indexes nil");
    size_t step = (size_t)sr;
    for (int c = 0; c < a->channels; c++) {
        double *ch = a->data[c];
        size_t n = a->len[c];
        for (size_t i = 0; i + 1 <= n; i += step) { /* for i = 0, #ch - 1, sampleRate */
            double avg = 0;
            size_t l = (n - i) < step ? (n - i) : step;
            for (size_t j = 1; j <= l; j++) avg = avg + ch[i + j - 1];
            avg = avg / (double)l;
            for (size_t j = 1; j <= l; j++) ch[i + j - 1] = ork_clamp(ch[i + j - 1] - avg, -1, 1);
        }
    }
    return ORK_OK;
}

/* aukit.lua:3483-3498: str_sub(audio, ...) on a table always raises (Q17). */
int ork_fx_trim(ork_audio *a, double threshold) {
    (void)a; (void)threshold;
    return ork__fail(ORK_E_LUA, "bad argument #1 to 'sub' (string expected, got table)");
}

/* aukit.lua:3505-3517 */
int ork_fx_delay(ork_audio *a, double delay, double multiplier) {
    double sd = floor(delay * a->sample_rate);
    if (sd < 0) return ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value (field '?')");
    for (int c = 0; c < a->channels; c++) {
        double *o = a->data[c];
        size_t n = a->len[c];
        if (sd >= (double)n) continue;
        size_t samples = (size_t)sd;
        double *original = (double *)malloc((n ? n : 1) * sizeof(double));
        if (!original) return ork__fail(ORK_E_NOMEM, "out of memory");
        memcpy(original, o, n * sizeof(double));
        for (size_t i = samples + 1; i <= n; i++) o[i - 1] = ork_clamp(o[i - 1] + original[i - samples - 1] * multiplier, -1, 1);
        free(original);
    }
    return ORK_OK;
}

/* aukit.lua:3524-3534 */
int ork_fx_echo(ork_audio *a, double delay, double multiplier) {
    double sd = floor(delay * a->sample_rate);
    if (sd < 0) return ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value (field '?')");
    for (int c = 0; c < a->channels; c++) {
        double *o = a->data[c];
        size_t n = a->len[c];
        if (sd >= (double)n) continue;
        size_t samples = (size_t)sd;
        for (size_t i = samples + 1; i <= n; i++) o[i - 1] = ork_clamp(o[i - 1] + o[i - samples - 1] * multiplier, -1, 1);
    }
    return ORK_OK;
}

static const double combDelayShift[4] = {0, -11.73, 19.31, -7.97}; /* aukit.lua:3536 */
static const double combDecayShift[4] = {0, 0.1313, 0.2743, 0.31}; /* aukit.lua:3537 */

/* aukit.lua:3546-3580 */
int ork_fx_reverb(ork_audio *a, double delay, double decay, double wet, double dry) {
    for (int c = 0; c < a->channels; c++) {
        double *o = a->data[c];
        long n = (long)a->len[c];
        double *sum = (double *)calloc((size_t)(n ? n : 1), sizeof(double));
        double *comb = (double *)malloc((size_t)(n ? n : 1) * sizeof(double));
        if (!sum || !comb) { free(sum); free(comb); return ork__fail(ORK_E_NOMEM, "out of memory"); }
        for (int k = 0; k < 4; k++) {
            double sf = floor((delay + combDelayShift[k]) / 1000 * a->sample_rate);
            double multiplier = decay - combDecayShift[k];
            if (sf < 1 && n > 0) { free(sum); free(comb); return ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value (field '?')"); }
            long samples = sf > (double)n ? n : (long)sf;
            for (long i = 1; i <= samples; i++) { comb[i - 1] = o[i - 1]; sum[i - 1] = sum[i - 1] + o[i - 1]; } /* (sum[i] or 0) + o[i] */
            if (sf < (double)n)
                for (long i = (long)sf + 1; i <= n; i++) {
                    double s = o[i - 1] + comb[i - (long)sf - 1] * multiplier;
                    comb[i - 1] = s;
                    sum[i - 1] = sum[i - 1] + s;
                }
        }
        free(comb);
        for (long i = 1; i <= n; i++) sum[i - 1] = sum[i - 1] * wet + o[i - 1] * dry; /* :3571 */
        long samples = (long)floor(0.08927 * a->sample_rate);                         /* :3573 */
        if (samples + 1 > n || samples < 20) { free(sum); return ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value (field '?')"); }
        sum[samples] = sum[samples] - 0.131 * sum[0];                                 /* :3574 */
        for (long i = samples + 2; i <= n; i++) sum[i - 1] = sum[i - 1] - 0.131 * sum[i - samples - 1] + 0.131 * sum[i + 20 - samples - 1];
        o[samples] = ork_clamp(sum[samples] - 0.131 * sum[0], -1, 1);                 /* :3576 */
        for (long i = samples + 2; i <= n; i++) o[i - 1] = ork_clamp(sum[i - 1] - 0.131 * sum[i - samples - 1] + 0.131 * sum[i + 20 - samples - 1], -1, 1);
        free(sum);
    }
    return ORK_OK;
}

/* aukit.lua:3586-3598 */
int ork_fx_lowpass(ork_audio *a, double frequency) {
    double al = 1 - exp(-(frequency / a->sample_rate) * 2 * M_PI);
    for (int c = 0; c < a->channels; c++) {
        double *d = a->data[c];
        for (size_t i = 1; i < a->len[c]; i++) {
            double l = d[i - 1];
            d[i] = l + al * (d[i] - l);
        }
    }
    return ORK_OK;
}

/* aukit.lua:3604-3618 */
int ork_fx_highpass(ork_audio *a, double frequency) {
    double al = 1 / (2 * M_PI * (frequency / a->sample_rate) + 1);
    for (int c = 0; c < a->channels; c++) {
        double *d = a->data[c];
        if (a->len[c] == 0) continue;
        double lx = d[0];
        for (size_t i = 1; i < a->len[c]; i++) {
            double llx = d[i];
            d[i] = al * (d[i - 1] + llx - lx);
            lx = llx;
        }
    }
    return ORK_OK;
}

/* ork_codecs.c — CPU ORACLE (test infrastructure only; see ork.h header).
 * Whole-buffer loaders of aukit.lua (AUKit 1.10.0): pcm, adpcm, msadpcm, g711,
 * dfpwm, mdfpwm, qoa, flac, plus the cc.audio.dfpwm predictor/encoder/decoder.
 */
#include "ork_internal.h"

/* aukit.lua:156-159 */
const int ork__ima_index_table[16] = {-1, -1, -1, -1, 2, 4, 6, 8, -1, -1, -1, -1, 2, 4, 6, 8};
/* aukit.lua:161-171 */
const int ork__ima_step_table[89] = {
    7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 19, 21, 23, 25, 28, 31, 34, 37, 41, 45,
    50, 55, 60, 66, 73, 80, 88, 97, 107, 118, 130, 143, 157, 173, 190, 209, 230, 253, 279, 307,
    337, 371, 408, 449, 494, 544, 598, 658, 724, 796, 876, 963, 1060, 1166, 1282, 1411, 1552, 1707, 1878, 2066,
    2272, 2499, 2749, 3024, 3327, 3660, 4026, 4428, 4871, 5358, 5894, 6484, 7132, 7845, 8630, 9493, 10442, 11487, 12635, 13899,
    15289, 16818, 18500, 20350, 22385, 24623, 27086, 29794, 32767};
/* aukit.lua:173-176 */
int ork__msadpcm_adapt(int nib) {
    static const int pos[8] = {230, 230, 230, 230, 307, 409, 512, 614};
    static const int neg[8] = {768, 614, 512, 409, 307, 230, 230, 230}; /* [-8..-1] */
    return nib >= 0 ? pos[nib] : neg[nib + 8];
}
/* aukit.lua:1662-1679 */
const int ork__qoa_dequant_tab[16][8] = {
    {1, -1, 3, -3, 5, -5, 7, -7},
    {5, -5, 18, -18, 32, -32, 49, -49},
    {16, -16, 53, -53, 95, -95, 147, -147},
    {34, -34, 113, -113, 203, -203, 315, -315},
    {63, -63, 210, -210, 378, -378, 588, -588},
    {104, -104, 345, -345, 621, -621, 966, -966},
    {158, -158, 528, -528, 950, -950, 1477, -1477},
    {228, -228, 760, -760, 1368, -1368, 2128, -2128},
    {316, -316, 1053, -1053, 1895, -1895, 2947, -2947},
    {422, -422, 1405, -1405, 2529, -2529, 3934, -3934},
    {548, -548, 1828, -1828, 3290, -3290, 5117, -5117},
    {696, -696, 2320, -2320, 4176, -4176, 6496, -6496},
    {868, -868, 2893, -2893, 5207, -5207, 8099, -8099},
    {1064, -1064, 3548, -3548, 6386, -6386, 9933, -9933},
    {1286, -1286, 4288, -4288, 7718, -7718, 12005, -12005},
    {1536, -1536, 5120, -5120, 9216, -9216, 14336, -14336},
};

/* string.unpack("<i2" etc.)  — value as a Lua number */
double ork__unpack_sample(const uint8_t *p, int bd, int data_type, int be) {
    if (data_type == ORK_FLOAT) {
        uint32_t u = be ? ((uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3])
                        : ((uint32_t)p[3] << 24 | (uint32_t)p[2] << 16 | (uint32_t)p[1] << 8 | p[0]);
        float f;
        memcpy(&f, &u, 4);
        return (double)f;
    }
    uint64_t u = 0;
    if (be) for (int i = 0; i < bd; i++) u = (u << 8) | p[i];
    else for (int i = bd - 1; i >= 0; i--) u = (u << 8) | p[i];
    if (data_type == ORK_SIGNED) {
        uint64_t sign = 1ull << (bd * 8 - 1);
        if (u & sign) return (double)((int64_t)u - (int64_t)(1ull << (bd * 8)));
    }
    return (double)u;
}

static int pcm_check(int bit_depth, int data_type, int channels, double sample_rate) {
    if (bit_depth != 8 && bit_depth != 16 && bit_depth != 24 && bit_depth != 32) return ork__fail(ORK_E_ARG, "bad argument #2 (invalid bit depth)");
    if (data_type < 0 || data_type > 2) return ork__fail(ORK_E_ARG, "bad argument #3 (invalid data type)");
    if (data_type == ORK_FLOAT && bit_depth != 32) return ork__fail(ORK_E_ARG, "bad argument #2 (float audio must have 32-bit depth)");
    if (channels < 1) return ork__fail(ORK_E_ARG, "bad argument #4 (number outside of range)");
    if (sample_rate < 1) return ork__fail(ORK_E_ARG, "bad argument #5 (number outside of range)");
    if (channels > ORK_MAX_CH) return ork__fail(ORK_E_UNSUPPORTED, "oracle supports at most %d channels", ORK_MAX_CH);
    return ORK_OK;
}

/* aukit.pcm with a string  aukit.lua:1049-1171 */
int ork_pcm(const uint8_t *data, size_t nbytes, int bit_depth, int data_type, int channels, double sample_rate,
            int interleaved, int big_endian, ork_audio *out) {
    int rc = pcm_check(bit_depth, data_type, channels, sample_rate);
    if (rc) return rc;
    int bd = bit_depth / 8;
    if (nbytes % ((size_t)bd * channels) != 0) return ork__fail(ORK_E_ARG, "bad argument #1 (uneven amount of data per channel)"); /* :1064 */
    size_t len = nbytes / bd / channels;
    double maxValue = ldexp(1.0, bit_depth - 1);
    rc = ork__audio_alloc(out, channels, len, sample_rate);
    if (rc) return rc;
    size_t pos = 0;
    if (interleaved && channels > 1) {
        for (size_t i = 0; i < len; i++)
            for (int j = 0; j < channels; j++, pos += bd) out->data[j][i] = ork__pcm_norm(ork__unpack_sample(data + pos, bd, data_type, big_endian), data_type, maxValue);
    } else {
        for (int j = 0; j < channels; j++)
            for (size_t i = 0; i < len; i++, pos += bd) out->data[j][i] = ork__pcm_norm(ork__unpack_sample(data + pos, bd, data_type, big_endian), data_type, maxValue);
    }
    return ORK_OK;
}

/* aukit.pcm with a table of numbers  aukit.lua:1077-1096 */
int ork_pcm_table(const double *values, size_t n, int bit_depth, int data_type, int channels, double sample_rate,
                  int interleaved, ork_audio *out) {
    int rc = pcm_check(bit_depth, data_type, channels, sample_rate);
    if (rc) return rc;
    if (n % (size_t)channels != 0) return ork__fail(ORK_E_ARG, "bad argument #1 (uneven amount of data per channel)");
    size_t len = n / channels;
    double maxValue = ldexp(1.0, bit_depth - 1);
    rc = ork__audio_alloc(out, channels, len, sample_rate);
    if (rc) return rc;
    size_t pos = 0;
    if (interleaved && channels > 1) {
        for (size_t i = 0; i < len; i++)
            for (int j = 0; j < channels; j++) out->data[j][i] = ork__pcm_norm(values[pos++], data_type, maxValue);
    } else {
        for (int j = 0; j < channels; j++)
            for (size_t i = 0; i < len; i++) out->data[j][i] = ork__pcm_norm(values[pos++], data_type, maxValue);
    }
    return ORK_OK;
}

/* core of aukit.adpcm  aukit.lua:1241-1273 on an explicit nibble sequence; len may be fractional in the Lua (floor) */
static int adpcm_core(const uint8_t *nib, size_t nnib, size_t len, int channels, double sample_rate, int interleaved,
                      const int *predictor_in, const int *step_index_in, ork_audio *out) {
    if (channels < 1 || channels > ORK_MAX_CH) return ork__fail(ORK_E_ARG, "bad argument #2 (number outside of range)");
    int predictor[ORK_MAX_CH], step_index[ORK_MAX_CH];
    for (int j = 0; j < channels; j++) {
        predictor[j] = predictor_in ? predictor_in[j] : 0;
        step_index[j] = step_index_in ? step_index_in[j] : 0;
        if (predictor[j] < -32768 || predictor[j] > 32767) return ork__fail(ORK_E_ARG, "bad argument #6 (number outside of range)");
        if (step_index[j] < 0 || step_index[j] > 88) return ork__fail(ORK_E_ARG, "bad argument #7 (number outside of range)");
    }
    if (len * (size_t)channels > nnib) return ork__fail(ORK_E_ARG, "adpcm_core: not enough nibbles");
    int rc = ork__audio_alloc(out, channels, len, sample_rate);
    if (rc) return rc;
    size_t pos = 0;
    if (interleaved) {
        for (size_t i = 0; i < len; i++)
            for (int j = 0; j < channels; j++) {
                int nibble = nib[pos++];
                int step = ork__ima_step_table[step_index[j]];                                   /* :1250 */
                step_index[j] = (int)ork_clamp(step_index[j] + ork__ima_index_table[nibble], 0, 88); /* :1251 */
                int diff = (((nibble % 8) * step) >> 2) + (step >> 3);                          /* :1252 (Q5) */
                if (nibble >= 8) predictor[j] = (int)ork_clamp(predictor[j] - diff, -32768, 32767);
                else predictor[j] = (int)ork_clamp(predictor[j] + diff, -32768, 32767);
                out->data[j][i] = (double)predictor[j] / (predictor[j] < 0 ? 32768 : 32767);    /* :1255 */
            }
    } else {
        for (int j = 0; j < channels; j++) {
            int p = predictor[j], si = step_index[j];
            for (size_t i = 0; i < len; i++) {
                int nibble = nib[pos++];
                int step = ork__ima_step_table[si];
                si = (int)ork_clamp(si + ork__ima_index_table[nibble], 0, 88);
                int diff = (((nibble % 8) * step) >> 2) + (step >> 3);
                if (nibble >= 8) p = (int)ork_clamp(p - diff, -32768, 32767);
                else p = (int)ork_clamp(p + diff, -32768, 32767);
                out->data[j][i] = (double)p / (p < 0 ? 32768 : 32767);                          /* :1269 */
            }
        }
    }
    return ORK_OK;
}

/* aukit.adpcm, string input  aukit.lua:1183-1274 */
int ork_adpcm(const uint8_t *data, size_t nbytes, int channels, double sample_rate, int top_first, int interleaved,
              const int *predictor, const int *step_index, ork_audio *out) {
    if (channels < 1) return ork__fail(ORK_E_ARG, "bad argument #2 (number outside of range)");
    uint8_t *nib = (uint8_t *)malloc(nbytes * 2 + 1);
    if (!nib) return ork__fail(ORK_E_NOMEM, "out of memory");
    for (size_t i = 0; i < nbytes; i++) { /* :1224-1228 */
        if (top_first) { nib[2 * i] = data[i] >> 4; nib[2 * i + 1] = data[i] & 0x0F; }
        else { nib[2 * i] = data[i] & 0x0F; nib[2 * i + 1] = data[i] >> 4; }
    }
    size_t len = nbytes * 2 / (size_t)channels; /* :1231 math.floor */
    int rc = adpcm_core(nib, nbytes * 2, len, channels, sample_rate, interleaved, predictor, step_index, out);
    free(nib);
    return rc;
}

/* aukit.adpcm, table-of-nibbles input  aukit.lua:1233-1238 */
int ork_adpcm_nibbles(const uint8_t *nib, size_t n, int channels, double sample_rate, int interleaved,
                      const int *predictor, const int *step_index, ork_audio *out) {
    if (channels < 1) return ork__fail(ORK_E_ARG, "bad argument #2 (number outside of range)");
    return adpcm_core(nib, n, n / (size_t)channels, channels, sample_rate, interleaved, predictor, step_index, out);
}

/* IMA-in-WAV block splitter of aukit.wav  aukit.lua:1509-1548 */
int ork_wav_adpcm(const uint8_t *data, size_t nbytes, int block_align, int channels, double sample_rate, ork_audio *out) {
    if (block_align <= 0) return ork__fail(ORK_E_ARG, "bad blockAlign");
    if (channels != 1 && channels != 2) return ork__fail(ORK_E_UNSUPPORTED, "IMA WAV splitter handles 1 or 2 channels");
    if (nbytes == 0) return ork__fail(ORK_E_LUA, "attempt to index a nil value (field '?')"); /* blocks[1]:concat */
    ork_vec acc[2] = {{0}, {0}};
    int rc = ORK_OK;
    for (size_t n = 0; n < nbytes && rc == ORK_OK; n += (size_t)block_align) {
        ork_audio blk;
        memset(&blk, 0, sizeof blk);
        if (channels == 2) {
            if (n + 8 > nbytes) { rc = ork__fail(ORK_E_LUA, "data string too short"); break; }
            int pred[2], idx[2];
            pred[0] = (int16_t)(data[n] | data[n + 1] << 8); idx[0] = data[n + 2];      /* "<hBxhB" :1513 */
            pred[1] = (int16_t)(data[n + 4] | data[n + 5] << 8); idx[1] = data[n + 6];
            size_t cap = (size_t)block_align * 2;
            uint8_t *nib = (uint8_t *)calloc(cap + 16, 1);
            if (!nib) { rc = ork__fail(ORK_E_NOMEM, "out of memory"); break; }
            size_t nn = 0;
            for (int i = 8; i <= block_align - 1; i += 8) { /* :1515 */
                if (n + (size_t)i + 8 > nbytes) { rc = ork__fail(ORK_E_LUA, "bad argument #1 to 'band' (number expected, got nil)"); break; }
                for (int k = 0; k < 4; k++) { /* left: bytes i..i+3, low nibble first, odd table slots */
                    uint8_t b = data[n + i + k];
                    nib[((size_t)(i - 7 + 2 * k) * 2 - 1) - 1] = b & 0x0F;
                    nib[((size_t)(i - 6 + 2 * k) * 2 - 1) - 1] = b >> 4;
                }
                for (int k = 0; k < 4; k++) { /* right: bytes i+4..i+7, even table slots */
                    uint8_t b = data[n + i + 4 + k];
                    nib[((size_t)(i - 7 + 2 * k) * 2) - 1] = b & 0x0F;
                    nib[((size_t)(i - 6 + 2 * k) * 2) - 1] = b >> 4;
                }
                nn = (size_t)i * 2;
            }
            if (rc == ORK_OK) rc = ork_adpcm_nibbles(nib, nn, 2, sample_rate, 1, pred, idx, &blk); /* :1541 */
            free(nib);
        } else {
            if (n + 3 > nbytes) { rc = ork__fail(ORK_E_LUA, "data string too short"); break; }
            int pred = (int16_t)(data[n] | data[n + 1] << 8);  /* "<hB" :1543 */
            int idx = data[n + 2] & 0x0F;                       /* :1544 (Q8) */
            size_t s = n + 4, e = n + (size_t)block_align;      /* str_sub(data, n+4, n+blockAlign-1) 1-based */
            if (e > nbytes) e = nbytes;
            size_t cnt = e > s ? e - s : 0;
            rc = ork_adpcm(data + (cnt ? s : 0), cnt, 1, sample_rate, 0, 0, &pred, &idx, &blk); /* :1545 */
        }
        if (rc == ORK_OK)
            for (int c = 0; c < channels; c++)
                for (size_t i = 0; i < blk.len[c]; i++)
                    if (ork__vec_push(&acc[c], blk.data[c][i])) rc = ork__fail(ORK_E_NOMEM, "out of memory");
        ork_audio_free(&blk);
    }
    if (rc) { free(acc[0].p); free(acc[1].p); return rc; }
    memset(out, 0, sizeof *out);
    out->channels = channels;
    out->sample_rate = sample_rate;
    for (int c = 0; c < channels; c++) {
        out->data[c] = acc[c].p ? acc[c].p : (double *)malloc(8);
        out->len[c] = acc[c].n;
    }
    return ORK_OK;
}

static const int ms_coef1_default[7] = {256, 512, 0, 192, 240, 460, 392};  /* aukit.lua:1304 */
static const int ms_coef2_default[7] = {0, -256, 0, 64, 0, -208, -232};

/* one MS-ADPCM nibble step  aukit.lua:1321-1324 */
static inline double ms_step(double *sample1, double *sample2, double *delta, double c1, double c2, int nib) {
    double predictor = ork_clamp(floor((*sample1 * c1 + *sample2 * c2) / 256) + nib * *delta, -32768, 32767);
    *sample2 = *sample1;
    *sample1 = predictor;
    double nd = floor(ork__msadpcm_adapt(nib) * *delta / 256);
    *delta = nd < 16 ? 16 : nd; /* math.max(nd, 16) */
    return predictor;
}
static inline int rd_i16(const uint8_t *p) { return (int16_t)(p[0] | p[1] << 8); }

/* aukit.msadpcm  aukit.lua:1283-1353 */
int ork_msadpcm(const uint8_t *data, size_t nbytes, int block_align, int channels, double sample_rate,
                const int *coef1, const int *coef2, int ncoef, ork_audio *out) {
    if (sample_rate < 1) return ork__fail(ORK_E_ARG, "bad argument #4 (number outside of range)");
    if (!coef1 || !coef2) { coef1 = ms_coef1_default; coef2 = ms_coef2_default; ncoef = 7; }
    if (channels != 1 && channels != 2) return ork__fail(ORK_E_LUA, "Unsupported number of channels: %d", channels);
    if (block_align <= 0) return ork__fail(ORK_E_ARG, "bad blockAlign");
    ork_vec left = {0}, right = {0};
    int rc = ORK_OK;
#define PUSH(v, x) do { if (ork__vec_push(&(v), (x))) { rc = ork__fail(ORK_E_NOMEM, "out of memory"); goto done; } } while (0)
    for (size_t n = 0; n < nbytes; n += (size_t)block_align) {
        if (channels == 2) {
            if (n + 14 > nbytes) { rc = ork__fail(ORK_E_LUA, "data string too short"); goto done; }
            int piL = data[n], piR = data[n + 1];                         /* "<BBhhhhhh" :1310 */
            double deltaL = rd_i16(data + n + 2), deltaR = rd_i16(data + n + 4);
            double s1L = rd_i16(data + n + 6), s1R = rd_i16(data + n + 8);
            double s2L = rd_i16(data + n + 10), s2R = rd_i16(data + n + 12);
            if (piL >= ncoef || piR >= ncoef) { rc = ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value (local 'c1L')"); goto done; }
            double c1L = coef1[piL], c2L = coef2[piL], c1R = coef1[piR], c2R = coef2[piR];
            PUSH(left, s2L / (s2L < 0 ? 32768 : 32767));
            PUSH(left, s1L / (s1L < 0 ? 32768 : 32767));
            PUSH(right, s2R / (s2R < 0 ? 32768 : 32767));
            PUSH(right, s1R / (s1R < 0 ? 32768 : 32767));
            for (int i = 14; i <= block_align - 1; i++) {
                if (n + (size_t)i >= nbytes) { rc = ork__fail(ORK_E_LUA, "bad argument #1 to 'rshift' (number expected, got nil)"); goto done; }
                int b = data[n + i], hi = b >> 4, lo = b & 0x0F;
                if (hi >= 8) hi -= 16;
                if (lo >= 8) lo -= 16;
                double p = ms_step(&s1L, &s2L, &deltaL, c1L, c2L, hi);
                PUSH(left, p / (p < 0 ? 32768 : 32767));
                p = ms_step(&s1R, &s2R, &deltaR, c1R, c2R, lo);
                PUSH(right, p / (p < 0 ? 32768 : 32767));
            }
        } else {
            if (nbytes < 7) { rc = ork__fail(ORK_E_LUA, "data string too short"); goto done; }
            /* str_unpack("<!1Bhhh", data) has NO position: every block re-reads the header at offset 1 (Q9) :1331 */
            int pi = data[0];
            double delta = rd_i16(data + 1), s1 = rd_i16(data + 3), s2 = rd_i16(data + 5);
            if (pi >= ncoef) { rc = ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value (local 'c1')"); goto done; }
            double c1 = coef1[pi], c2 = coef2[pi];
            PUSH(left, s2 / (s2 < 0 ? 32768 : 32767));
            PUSH(left, s1 / (s1 < 0 ? 32768 : 32767));
            for (int i = 7; i <= block_align - 1; i++) {
                if (n + (size_t)i >= nbytes) { rc = ork__fail(ORK_E_LUA, "bad argument #1 to 'rshift' (number expected, got nil)"); goto done; }
                int b = data[n + i], hi = b >> 4, lo = b & 0x0F;
                if (hi >= 8) hi -= 16;
                if (lo >= 8) lo -= 16;
                double p = ms_step(&s1, &s2, &delta, c1, c2, hi);
                PUSH(left, p / (p < 0 ? 32768 : 32767));
                p = ms_step(&s1, &s2, &delta, c1, c2, lo);
                PUSH(left, p / (p < 0 ? 32768 : 32767));
            }
        }
    }
done:
#undef PUSH
    if (rc) { free(left.p); free(right.p); return rc; }
    memset(out, 0, sizeof *out);
    out->channels = channels;
    out->sample_rate = sample_rate;
    out->data[0] = left.p ? left.p : (double *)malloc(8);
    out->len[0] = left.n;
    if (channels == 2) { out->data[1] = right.p ? right.p : (double *)malloc(8); out->len[1] = right.n; }
    else free(right.p);
    return ORK_OK;
}

/* aukit.lua:1374-1378 (same code at :2886-2890) */
int ork__g711_expand(int byte, int ulaw, int *neg) {
    int b = byte ^ (ulaw ? 0xFF : 0x55);
    int m = b & 0x0F, e = (b >> 4) & 7;
    if (!ulaw && e == 0) m = m * 4 + 2;
    else m = (m * 2 + 33) << e;
    if (ulaw) m = m - 33;
    *neg = (((b & 0x80) != 0) == (ulaw != 0));
    return m;
}

/* aukit.g711  aukit.lua:1361-1384 */
int ork_g711(const uint8_t *data, size_t nbytes, int ulaw, int channels, double sample_rate, ork_audio *out) {
    if (channels < 1 || channels > ORK_MAX_CH) return ork__fail(ORK_E_ARG, "channels out of range");
    memset(out, 0, sizeof *out);
    out->channels = channels;
    out->sample_rate = sample_rate;
    for (int c = 0; c < channels; c++) {
        size_t l = nbytes / channels + ((size_t)c < nbytes % channels ? 1 : 0);
        out->len[c] = l;
        out->data[c] = (double *)malloc((l ? l : 1) * sizeof(double));
        if (!out->data[c]) return ork__fail(ORK_E_NOMEM, "out of memory");
    }
    for (size_t p = 0; p < nbytes; p++) {
        int neg, m = ork__g711_expand(data[p], ulaw, &neg);
        out->data[p % channels][p / channels] = (double)m / (neg ? -0x2000 : 0x2000); /* :1379 */
    }
    return ORK_OK;
}

/* ---------------- cc.audio.dfpwm (CC: Tweaked ROM; restated, parity unpinned) ----------------
 * DFPWM1a: PREC = 10, strength floor 2^(PREC-8+1) = 8, anti-jerk + LPF(140/256) in the decoder. */
#define DF_PREC 10
#define DF_PREC_POW (1 << DF_PREC)
#define DF_PREC_POW_HALF (1 << (DF_PREC - 1))
#define DF_STRENGTH_MIN (1 << (DF_PREC - 8 + 1))

static inline int floordiv(int a, int b) { /* math.floor(a / b), b > 0 */
    int q = a / b, r = a % b;
    return (r != 0 && r < 0) ? q - 1 : q;
}

static inline int dfpwm_predict(ork_dfpwm_pred *p, int current_bit) {
    int target = current_bit ? 127 : -128;
    int next_charge = p->charge + floordiv(p->strength * (target - p->charge) + DF_PREC_POW_HALF, DF_PREC_POW);
    if (next_charge == p->charge && next_charge != target) next_charge += current_bit ? 1 : -1;
    int z = (current_bit == p->previous_bit) ? DF_PREC_POW - 1 : 0;
    int next_strength = p->strength;
    if (next_strength != z) next_strength += (current_bit == p->previous_bit) ? 1 : -1;
    if (next_strength < DF_STRENGTH_MIN) next_strength = DF_STRENGTH_MIN;
    p->charge = next_charge;
    p->strength = next_strength;
    p->previous_bit = current_bit;
    return next_charge;
}

void ork_dfpwm_dec_init(ork_dfpwm_dec *d) { memset(d, 0, sizeof *d); }
void ork_dfpwm_enc_init(ork_dfpwm_enc *e) { memset(e, 0, sizeof *e); }

void ork_dfpwm_decode(ork_dfpwm_dec *d, const uint8_t *in, size_t nbytes, int8_t *out) {
    for (size_t i = 0; i < nbytes; i++) {
        int input_byte = in[i];
        for (int k = 0; k < 8; k++) {
            int current_bit = input_byte & 1;
            int charge = dfpwm_predict(&d->p, current_bit);
            int antijerk = charge;
            if (current_bit != d->previous_bit) antijerk = floordiv(charge + d->previous_charge + 1, 2);
            d->previous_charge = charge;
            d->previous_bit = current_bit;
            d->low_pass_charge += floordiv((antijerk - d->low_pass_charge) * 140 + 0x80, 256);
            *out++ = (int8_t)d->low_pass_charge;
            input_byte >>= 1;
        }
    }
}

int ork_dfpwm_encode(ork_dfpwm_enc *e, const double *samples, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i += 8) {
        int this_byte = 0;
        for (int j = 0; j < 8; j++) {
            double v = (i + j < n) ? samples[i + j] : 0; /* input[i + j] or 0 */
            double fv = floor(v);
            if (fv > 127 || fv < -128 || fv != fv)
                return ork__fail(ORK_E_LUA, "Amplitude at position %zu was %g, but should be between -128 and 127", i + j + 1, fv);
            int inp_charge = (int)fv;
            int current_bit = inp_charge > e->previous_charge || (inp_charge == e->previous_charge && inp_charge == 127);
            this_byte = (this_byte >> 1) + (current_bit ? 128 : 0);
            e->previous_charge = dfpwm_predict(&e->p, current_bit);
        }
        *out++ = (uint8_t)this_byte;
    }
    return ORK_OK;
}

/* aukit.dfpwm  aukit.lua:1392-1414 — 6001-byte slices advanced by 6000 (Q10) */
int ork_dfpwm(const uint8_t *data, size_t nbytes, int channels, double sample_rate, ork_audio *out) {
    if (channels < 1) return ork__fail(ORK_E_ARG, "bad argument #2 (number outside of range)");
    if (sample_rate < 1) return ork__fail(ORK_E_ARG, "bad argument #3 (number outside of range)");
    size_t nslices = (nbytes + 5999) / 6000;
    double *audio = (double *)malloc(((nbytes + nslices) * 8 + 8) * sizeof(double));
    int8_t *tmp = (int8_t *)malloc(6001 * 8);
    if (!audio || !tmp) { free(audio); free(tmp); return ork__fail(ORK_E_NOMEM, "out of memory"); }
    ork_dfpwm_dec dec;
    ork_dfpwm_dec_init(&dec);
    size_t last = 0;
    for (size_t pos = 0; pos < nbytes; pos += 6000) {      /* while pos <= #data (1-based) */
        size_t cnt = nbytes - pos < 6001 ? nbytes - pos : 6001; /* str_sub(data, pos, pos + 6000) */
        ork_dfpwm_decode(&dec, data + pos, cnt, tmp);
        for (size_t i = 0; i < cnt * 8; i++) audio[last + i] = tmp[i];
        last += cnt * 8;
    }
    free(tmp);
    int rc = ork_pcm_table(audio, last, 8, ORK_SIGNED, channels, sample_rate, 1, out); /* :1413 */
    free(audio);
    return rc;
}

/* "<Is1s1s1" after the 7-byte magic; returns 0-based offset of the payload or <0 */
static long mdfpwm_header(const uint8_t *data, size_t nbytes, uint32_t *length) {
    if (nbytes < 7 || memcmp(data, "MDFPWM\3", 7) != 0) return -1;
    size_t pos = 7;
    if (pos + 4 > nbytes) return -2;
    *length = data[pos] | data[pos + 1] << 8 | data[pos + 2] << 16 | (uint32_t)data[pos + 3] << 24;
    pos += 4;
    for (int k = 0; k < 3; k++) {
        if (pos + 1 > nbytes) return -2;
        size_t l = data[pos];
        pos += 1 + l;
        if (pos > nbytes) return -2;
    }
    return (long)pos;
}

/* aukit.mdfpwm  aukit.lua:1420-1448 */
int ork_mdfpwm(const uint8_t *data, size_t nbytes, ork_audio *out) {
    uint32_t length;
    long hp = mdfpwm_header(data, nbytes, &length);
    if (hp == -1) return ork__fail(ORK_E_ARG, "bad argument #1 (not a MDFPWM file)");
    if (hp < 0) return ork__fail(ORK_E_LUA, "data string too short");
    size_t cap = (nbytes - (size_t)hp) * 8 + 16;
    double *audio = (double *)calloc(cap, sizeof(double));
    int8_t *tl = (int8_t *)malloc(6000 * 8), *tr = (int8_t *)malloc(6000 * 8);
    if (!audio || !tl || !tr) { free(audio); free(tl); free(tr); return ork__fail(ORK_E_NOMEM, "out of memory"); }
    ork_dfpwm_dec dl, dr;
    ork_dfpwm_dec_init(&dl);
    ork_dfpwm_dec_init(&dr);
    size_t last = 0;
    int rc = ORK_OK;
    for (size_t pos = (size_t)hp; pos < nbytes; pos += 12000) {
        size_t nl = nbytes - pos < 6000 ? nbytes - pos : 6000;
        ork_dfpwm_decode(&dl, data + pos, nl, tl);
        if (nl == 0) break;
        size_t nr = pos + 6000 < nbytes ? (nbytes - pos - 6000 < 6000 ? nbytes - pos - 6000 : 6000) : 0;
        if (nr != nl) { rc = ork__fail(ORK_E_UNSUPPORTED, "MDFPWM payload is not a whole number of L/R block pairs (table would have holes)"); break; }
        for (size_t i = 1; i <= nl * 8; i++) audio[last + i * 2 - 1 - 1] = tl[i - 1];
        ork_dfpwm_decode(&dr, data + pos + 6000, nr, tr);
        for (size_t i = 1; i <= nr * 8; i++) audio[last + i * 2 - 1] = tr[i - 1];
        last += nl * 8 + nr * 8;
    }
    free(tl); free(tr);
    if (rc) { free(audio); return rc; }
    size_t n = last;
    if ((double)length * 8 < (double)n) n = (size_t)length * 8; /* :1444 */
    rc = ork_pcm_table(audio, n, 8, ORK_SIGNED, 2, 48000, 1, out);
    free(audio);
    return rc;
}

/* ---------------- QOA  aukit.lua:1681-1701 ---------------- */
/* signed_rshift(a, b): bit32.arshift reduces a mod 2^32, sign-extends bit 31 */
static double signed_rshift(double a, int b) {
    double m = fmod(a, 4294967296.0);
    if (m < 0) m += 4294967296.0;
    uint32_t u = (uint32_t)m;
    int32_t s = (int32_t)u;
    return (double)(s >> b);
}
double ork__qoa_predict(const ork_qoa_lms *l) {
    return signed_rshift(l->weights[0] * l->history[0] + l->weights[1] * l->history[1] + l->weights[2] * l->history[2] + l->weights[3] * l->history[3], 13);
}
void ork__qoa_update(ork_qoa_lms *l, double sample, double residual) {
    double delta = signed_rshift(residual, 4);
    for (int i = 0; i < 4; i++) l->weights[i] = l->weights[i] + (l->history[i] < 0 ? -delta : delta);
    l->history[0] = l->history[1]; l->history[1] = l->history[2]; l->history[2] = l->history[3]; l->history[3] = sample;
}
static inline uint32_t rd_be32(const uint8_t *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
static inline int rd_be16s(const uint8_t *p) { return (int16_t)(p[0] << 8 | p[1]); }

/* aukit.qoa  aukit.lua:1706-1777 */
int ork_qoa(const uint8_t *data, size_t nbytes, ork_audio *out) {
    if (nbytes < 12) return ork__fail(ORK_E_LUA, "data string too short");
    if (memcmp(data, "qoaf", 4) != 0) return ork__fail(ORK_E_ARG, "Not a QOA file");
    double file_samples = rd_be32(data + 4);
    size_t pos = 8; /* 0-based; Lua pos = 9 */
    int file_channels = data[pos];
    double file_rate = (double)((uint32_t)data[pos + 1] << 16 | (uint32_t)data[pos + 2] << 8 | data[pos + 3]);
    if (file_channels < 1 || file_channels > ORK_MAX_CH) return ork__fail(ORK_E_UNSUPPORTED, "QOA channel count");
    ork_vec ch[ORK_MAX_CH];
    memset(ch, 0, sizeof ch);
    ork_qoa_lms lms[ORK_MAX_CH];
    memset(lms, 0, sizeof lms);
    double sample_pos = 0;
    int rc = ORK_OK;
    /* while pos + 16*ch + 8 <= #data (1-based pos) and sample_pos < file_samples */
    while ((pos + 1) + 16 * (size_t)file_channels + 8 <= nbytes && sample_pos < file_samples) {
        int channels = data[pos];
        double samplerate = (double)((uint32_t)data[pos + 1] << 16 | (uint32_t)data[pos + 2] << 8 | data[pos + 3]);
        int samples = data[pos + 4] << 8 | data[pos + 5];
        int frame_size = data[pos + 6] << 8 | data[pos + 7];
        pos += 8;
        int data_size = frame_size - 8 - 4 * 4 * channels;
        int num_slices = (int)floor((double)data_size / 8);
        int max_total_samples = num_slices * 20;
        if (channels != file_channels || samplerate != file_rate || (double)frame_size > (double)nbytes - (double)(pos + 1) + 1 ||
            samples * channels > max_total_samples)
            break; /* :1732-1740 */
        for (int c = 0; c < channels; c++) {
            if (pos + 16 > nbytes) { rc = ork__fail(ORK_E_LUA, "data string too short"); goto done; }
            for (int k = 0; k < 4; k++) lms[c].history[k] = rd_be16s(data + pos + 2 * k);
            pos += 8;
            for (int k = 0; k < 4; k++) lms[c].weights[k] = rd_be16s(data + pos + 2 * k);
            pos += 8;
        }
        for (int sample_index = 1; sample_index <= samples; sample_index += 20) {
            for (int c = 0; c < channels; c++) {
                if (pos + 8 > nbytes) { rc = ork__fail(ORK_E_LUA, "data string too short"); goto done; }
                uint32_t sliceH = rd_be32(data + pos), sliceL = rd_be32(data + pos + 4);
                pos += 8;
                int scalefactor = (sliceH >> 28) & 15;
                for (int si = sample_index; si <= sample_index + 19; si++) {
                    double predicted = ork__qoa_predict(&lms[c]);
                    int quantized = (sliceH >> 25) & 7;
                    double dequantized = ork__qoa_dequant_tab[scalefactor][quantized];
                    double reconstructed = fmin(fmax(predicted + dequantized, -32768), 32767);
                    if (ork__vec_set(&ch[c], (size_t)(sample_pos + si) - 1, reconstructed / (reconstructed < 0 ? 32768 : 32767))) { rc = ork__fail(ORK_E_NOMEM, "out of memory"); goto done; }
                    sliceH = (sliceH << 3) + ((sliceL >> 29) & 7);
                    sliceL = sliceL << 3;
                    ork__qoa_update(&lms[c], reconstructed, dequantized);
                }
            }
        }
        sample_pos = sample_pos + samples;
    }
done:
    if (rc) { for (int c = 0; c < ORK_MAX_CH; c++) free(ch[c].p); return rc; }
    memset(out, 0, sizeof *out);
    out->channels = file_channels;
    out->sample_rate = file_rate;
    for (int c = 0; c < file_channels; c++) { out->data[c] = ch[c].p ? ch[c].p : (double *)malloc(8); out->len[c] = ch[c].n; }
    return ORK_OK;
}

/* ---------------- FLAC  aukit.lua:311-619 ---------------- */
struct ork_flac_dec {
    const uint8_t *data;
    size_t n, pos;          /* pos: 0-based index of next byte */
    uint64_t bitBuffer;     /* kept mod 2^44 (aukit.lua:357) */
    int bitBufferLen;
    int eof;                /* a readUint returned nil */
    int numChannels, sampleDepth;
};

/* BitInputStream.readUint  aukit.lua:351-364.  On nil sets d->eof and returns 0. */
static double flac_readUint(ork_flac_dec *d, int n) {
    if (n == 0) return 0;
    while (d->bitBufferLen < n) {
        if (d->pos >= d->n) { d->eof = 1; return 0; }
        uint64_t temp = d->data[d->pos++];
        d->bitBuffer = (d->bitBuffer * 256 + temp) % 0x100000000000ull;
        d->bitBufferLen += 8;
    }
    d->bitBufferLen -= n;
    uint64_t result = d->bitBuffer >> d->bitBufferLen; /* math.floor(bitBuffer / 2^bitBufferLen) */
    if (n < 32) result = result % (1ull << n);
    return (double)result;
}
static double flac_readSignedInt(ork_flac_dec *d, int n) { /* :365-369 */
    double v = flac_readUint(d, n);
    if (v >= ldexp(1.0, n - 1)) v = v - ldexp(1.0, n);
    return v;
}
static double flac_readRice(ork_flac_dec *d, int param) { /* :370-376 */
    double val = 0;
    while (flac_readUint(d, 1) == 0) { if (d->eof) return 0; val = val + 1; }
    val = val * ldexp(1.0, param) + flac_readUint(d, param);
    double m = fmod(val, 4294967296.0);
    if (((uint32_t)m) & 1) return -floor(val / 2) - 1;
    else return floor(val / 2);
}

static int flac_residuals(ork_flac_dec *d, int warmup, int blockSize, double *result) { /* :380-409 */
    int method = (int)flac_readUint(d, 2);
    if (d->eof) return ork__fail(ORK_E_LUA, "attempt to compare nil with number");
    if (method >= 2) return ork__fail(ORK_E_LUA, "Reserved residual coding method %d", method);
    int paramBits = method == 0 ? 4 : 5;
    int escapeParam = method == 0 ? 0xF : 0x1F;
    int partitionOrder = (int)flac_readUint(d, 4);
    int numPartitions = 1 << partitionOrder;
    if (blockSize % numPartitions != 0) return ork__fail(ORK_E_LUA, "Block size not divisible by number of Rice partitions");
    int partitionSize = blockSize / numPartitions;
    for (int i = 0; i < numPartitions; i++) {
        int start = i * partitionSize + (i == 0 ? warmup : 0);
        int endd = (i + 1) * partitionSize;
        int param = (int)flac_readUint(d, paramBits);
        if (param < escapeParam) {
            for (int j = start; j < endd; j++) result[j] = flac_readRice(d, param);
        } else {
            int numBits = (int)flac_readUint(d, 5);
            for (int j = start; j < endd; j++) result[j] = flac_readSignedInt(d, numBits);
        }
        if (d->eof) return ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value");
    }
    return ORK_OK;
}

static void flac_restore_lpc(double *result, const double *coefs, int ncoefs, int shift, int blockSize) { /* :411-419 */
    double div = ldexp(1.0, shift); /* 2^shift, shift may be negative */
    for (int i = ncoefs; i <= blockSize - 1; i++) {
        double sum = 0;
        for (int j = 0; j <= ncoefs - 1; j++) sum = sum + result[i - j - 1] * coefs[j];
        result[i] = result[i] + floor(sum / div);
    }
}

static const double FIXED_COEFS[5][4] = {{0}, {1}, {2, -1}, {3, -3, 1}, {4, -6, 4, -1}}; /* :334-340 */

static int flac_subframe(ork_flac_dec *d, int sampleDepth, int blockSize, double *result) { /* :443-470 */
    flac_readUint(d, 1);
    int type = (int)flac_readUint(d, 6);
    int shift = (int)flac_readUint(d, 1);
    if (shift == 1)
        while (flac_readUint(d, 1) == 0) { if (d->eof) break; shift = shift + 1; }
    if (d->eof) return ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value");
    sampleDepth = sampleDepth - shift;
    int rc;
    if (type == 0) {
        double c = flac_readSignedInt(d, sampleDepth);
        for (int i = 0; i < blockSize; i++) result[i] = c;
    } else if (type == 1) {
        for (int i = 0; i < blockSize; i++) result[i] = flac_readSignedInt(d, sampleDepth);
    } else if (8 <= type && type <= 12) { /* :421-427 */
        int predOrder = type - 8;
        for (int i = 0; i < predOrder; i++) result[i] = flac_readSignedInt(d, sampleDepth);
        if ((rc = flac_residuals(d, predOrder, blockSize, result))) return rc;
        flac_restore_lpc(result, FIXED_COEFS[predOrder], predOrder, 0, blockSize);
    } else if (32 <= type && type <= 63) { /* :429-441 */
        int lpcOrder = type - 31;
        double coefs[32];
        for (int i = 0; i < lpcOrder; i++) result[i] = flac_readSignedInt(d, sampleDepth);
        int precision = (int)flac_readUint(d, 4) + 1;
        int lshift = (int)flac_readSignedInt(d, 5);
        for (int i = 0; i < lpcOrder; i++) coefs[i] = flac_readSignedInt(d, precision);
        if ((rc = flac_residuals(d, lpcOrder, blockSize, result))) return rc;
        flac_restore_lpc(result, coefs, lpcOrder, lshift, blockSize);
    } else {
        return ork__fail(ORK_E_LUA, "Reserved subframe type");
    }
    if (d->eof) return ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value");
    double mul = ldexp(1.0, shift);
    for (int i = 0; i < blockSize; i++) result[i] = result[i] * mul;
    return ORK_OK;
}

int ork__flac_open(const uint8_t *data, size_t n, ork_flac_dec **dd, double *sample_rate, int *channels, int *depth, double *num_samples) { /* :569-614 */
    if (n < 4) return ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value");
    if (rd_be32(data) != 0x664C6143) return ork__fail(ORK_E_LUA, "Invalid magic string");
    size_t pos = 4;
    int last = 0, have = 0;
    double sr = 0, ns = 0;
    int nc = 0, sd = 0;
    while (!last) {
        if (pos + 4 > n) return ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value");
        int temp = data[pos++];
        last = (temp & 0x80) != 0;
        int type = temp & 0x7F;
        size_t length = (size_t)data[pos] << 16 | (size_t)data[pos + 1] << 8 | data[pos + 2];
        pos += 3;
        if (type == 0) {
            if (pos + 34 > n) return ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value");
            const uint8_t *p = data + pos;
            sr = (double)(p[10] << 8 | p[11]) * 16 + (p[12] >> 4);
            nc = ((p[12] >> 1) & 7) + 1;
            sd = (p[12] & 1) * 16 + (p[13] >> 4) + 1;
            ns = (double)rd_be32(p + 14) + (double)(p[13] & 15) * 4294967296.0;
            pos += 34;
            have = 1;
        } else {
            pos += length; /* type 4 is parsed for metadata in the reference; it ends at the same offset */
        }
    }
    if (!have) return ork__fail(ORK_E_LUA, "Stream info metadata block absent");
    if (sd % 8 != 0) return ork__fail(ORK_E_LUA, "Sample depth not supported");
    ork_flac_dec *d = (ork_flac_dec *)calloc(1, sizeof *d);
    if (!d) return ork__fail(ORK_E_NOMEM, "out of memory");
    d->data = data; d->n = n; d->pos = pos > n ? n : pos;
    d->numChannels = nc; d->sampleDepth = sd;
    *dd = d; *sample_rate = sr; *channels = nc; *depth = sd; *num_samples = ns;
    return ORK_OK;
}
void ork__flac_close(ork_flac_dec *d) { free(d); }

/* decodeFrame + decodeSubframes  aukit.lua:472-567 */
int ork__flac_frame(ork_flac_dec *d, double **out, size_t *block_size) {
    /* temp = inp.readByte(); if temp == nil then return false */
    double temp = flac_readUint(d, 8);
    if (d->eof) return 0;
    double sync = temp * 64 + flac_readUint(d, 6);
    if (d->eof) return ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value");
    if (sync != 0x3FFE) return ork__fail(ORK_E_LUA, "Sync code expected");
    flac_readUint(d, 2);
    int blockSizeCode = (int)flac_readUint(d, 4);
    int sampleRateCode = (int)flac_readUint(d, 4);
    int chanAsgn = (int)flac_readUint(d, 4);
    flac_readUint(d, 4);
    int t = (int)flac_readUint(d, 8);
    if (d->eof) return ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value");
    int t2 = -1;
    for (int i = 7; i >= 0; i--) { if (!(t & (1 << i))) break; t2 = t2 + 1; }
    for (int i = 1; i <= t2; i++) flac_readUint(d, 8);
    int blockSize;
    if (blockSizeCode == 1) blockSize = 192;
    else if (2 <= blockSizeCode && blockSizeCode <= 5) blockSize = 576 << (blockSizeCode - 2);
    else if (blockSizeCode == 6) blockSize = (int)flac_readUint(d, 8) + 1;
    else if (blockSizeCode == 7) blockSize = (int)flac_readUint(d, 16) + 1;
    else if (8 <= blockSizeCode && blockSizeCode <= 15) blockSize = 256 << (blockSizeCode - 8);
    else return ork__fail(ORK_E_LUA, "Reserved block size");
    if (sampleRateCode == 12) flac_readUint(d, 8);
    else if (sampleRateCode == 13 || sampleRateCode == 14) flac_readUint(d, 16);
    flac_readUint(d, 8);
    if (d->eof) return ork__fail(ORK_E_LUA, "attempt to perform arithmetic on a nil value");

    int nch = d->numChannels, sampleDepth = d->sampleDepth, rc = ORK_OK;
    double *sub[ORK_MAX_CH] = {0};
    for (int c = 0; c < nch; c++) {
        sub[c] = (double *)calloc((size_t)blockSize + 32, sizeof(double)); /* +32: warm-up entries past blockSize (a Lua table just grows) */
        if (!sub[c]) { rc = ork__fail(ORK_E_NOMEM, "out of memory"); goto fail; }
    }
    if (0 <= chanAsgn && chanAsgn <= 7) { /* :475-478 */
        for (int ch = 0; ch < nch; ch++)
            if ((rc = flac_subframe(d, sampleDepth, blockSize, sub[ch]))) goto fail;
    } else if (8 <= chanAsgn && chanAsgn <= 10) { /* :479-497 */
        if (nch < 2) { rc = ork__fail(ORK_E_LUA, "attempt to index a nil value (field '?')"); goto fail; }
        if ((rc = flac_subframe(d, sampleDepth + (chanAsgn == 9 ? 1 : 0), blockSize, sub[0]))) goto fail;
        if ((rc = flac_subframe(d, sampleDepth + (chanAsgn == 9 ? 0 : 1), blockSize, sub[1]))) goto fail;
        if (chanAsgn == 8) for (int i = 0; i < blockSize; i++) sub[1][i] = sub[0][i] - sub[1][i];
        else if (chanAsgn == 9) for (int i = 0; i < blockSize; i++) sub[0][i] = sub[0][i] + sub[1][i];
        else for (int i = 0; i < blockSize; i++) {
            double side = sub[1][i];
            double right = sub[0][i] - floor(side / 2);
            sub[1][i] = right;
            sub[0][i] = right + side;
        }
    } else { rc = ork__fail(ORK_E_LUA, "Reserved channel assignment"); goto fail; }
    {
        double half = ldexp(1.0, sampleDepth - 1), full = ldexp(1.0, sampleDepth);
        for (int ch = 0; ch < nch; ch++)
            for (int i = 0; i < blockSize; i++) { /* :501-507 (Q14) */
                double s = sub[ch][i];
                if (s >= half) s = s - full;
                sub[ch][i] = s / full;
            }
    }
    d->bitBufferLen -= d->bitBufferLen % 8; /* alignToByte :345-347 */
    flac_readUint(d, 16);
    /* a nil here is discarded (value unused), so EOF inside the CRC is not an error; the NEXT readByte returns nil */
    for (int c = 0; c < nch; c++) out[c] = sub[c];
    *block_size = (size_t)blockSize;
    return 1;
fail:
    for (int c = 0; c < nch; c++) free(sub[c]);
    return rc;
}

/* aukit.flac  aukit.lua:1657-1660 → decodeFLAC(data, nil, head=nil) */
int ork_flac(const uint8_t *data, size_t nbytes, ork_audio *out) {
    ork_flac_dec *d;
    double sr, ns;
    int nch, depth;
    int rc = ork__flac_open(data, nbytes, &d, &sr, &nch, &depth, &ns);
    if (rc) return rc;
    ork_vec acc[ORK_MAX_CH];
    memset(acc, 0, sizeof acc);
    for (;;) {
        double *fr[ORK_MAX_CH];
        size_t bs;
        int r = ork__flac_frame(d, fr, &bs);
        if (r == 0) break;
        if (r < 0) { rc = r; break; }
        for (int c = 0; c < nch; c++) {
            for (size_t i = 0; i < bs; i++)
                if (ork__vec_push(&acc[c], fr[c][i])) rc = ork__fail(ORK_E_NOMEM, "out of memory");
            free(fr[c]);
        }
        if (rc) break;
    }
    ork__flac_close(d);
    if (rc) { for (int c = 0; c < ORK_MAX_CH; c++) free(acc[c].p); return rc; }
    memset(out, 0, sizeof *out);
    out->channels = nch;
    out->sample_rate = sr;
    for (int c = 0; c < nch; c++) { out->data[c] = acc[c].p ? acc[c].p : (double *)malloc(8); out->len[c] = acc[c].n; }
    return ORK_OK;
}

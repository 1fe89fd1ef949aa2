--- aukit.lua (aukit_amd) — LuaJIT-FFI shim: AUKit's Lua API surface for the decode → resample → effects hot path, backed
--- by libaukit_hip.so (include/aukit_hip.h).  Host code stays Lua: `local aukit = require "aukit"` keeps working, the
--- per-sample loops run on the MI355X.
---
--- NOT EXECUTED IN THIS REPOSITORY: the build image has no Lua / LuaJIT (DESIGN.md §1).  The Python mirror aukit_amd/aukit.py
--- implements the same mapping over the same C entry points and is what the GPU tests drive; this file is the binding a
--- maintainer would add next to the reference (INTEGRATION.md has the per-function support matrix).  Strings in, tables out —
--- the shapes of the reference: Audio.data[c][i] doubles, stream iterators returning `chunk, pos` with chunk[c][i] in [-128, 127].
---
--- Covered (reference line numbers in aukit.lua): loaders aukit.pcm :1049, .adpcm :1183, .msadpcm :1283, .g711 :1361, .dfpwm :1392,
--- .mdfpwm :1420, .wav :1456, .aiff :1580, .au :1639, .flac :1657, .qoa :1706; aukit.new :1783, .tone :1808, .pack :1861;
--- Audio:len / :channels / :resample :653 / :mono :677 / :concat :695 / :sub :725 / :combine :751 / :split :781 / :mix :804 /
--- :rep :839 / :reverse :856 / :pcm :901 / :stream :921 / :wav :954 / :dfpwm :1005; aukit.stream.* :2228-3337; aukit.effects.* :3356-3618.
--- aukit.noise :1840 (the device draws: Philox keyed by a seed — the reference's math.random stream is not reproducible), aukit.detect :2156 (host-side
--- bytes), Audio:wav's LIST / INFO chunk :946-956.  Left to the reference's own Lua (not on the path: SURVEY.md §8): aukit.play / Player, ID3 / LIST READING.
local ffi = require "ffi"

ffi.cdef [[
typedef struct aukit_ctx aukit_ctx; typedef struct aukit_batch aukit_batch; typedef struct aukit_audio aukit_audio; typedef struct aukit_chunks aukit_chunks;
typedef struct { int32_t codec, channels; double sample_rate; int32_t bit_depth, data_type, big_endian, interleaved, ulaw, top_first, block_align, ncoef;
                 int16_t coef1[32], coef2[32]; int32_t predictor[64], step_index[64]; } aukit_codec_desc;
typedef struct { aukit_codec_desc desc; uint64_t payload_off, payload_len; int32_t wav_data_type, bit_depth; double length_seconds; } aukit_container;
const char *aukit_last_error(void);
int aukit_ctx_create(aukit_ctx **out, int device); void aukit_ctx_destroy(aukit_ctx *ctx);
int aukit_parse_container(const uint8_t *bytes, uint64_t n, int kind, int stream, aukit_container *out);
int aukit_batch_upload(aukit_ctx *, aukit_batch **, const uint8_t *bytes, const uint64_t *offsets, uint32_t n);
int aukit_batch_info(const aukit_batch *, uint32_t *n, uint64_t *total); int aukit_batch_download(aukit_ctx *, const aukit_batch *, uint8_t *dst); void aukit_batch_free(aukit_batch *);
int aukit_audio_upload(aukit_ctx *, aukit_audio **, const double *samples, const uint64_t *lens, uint32_t n, int channels, double rate, int dtype);
int aukit_audio_info(const aukit_audio *, uint32_t *n, int *channels, double *rate, int *dtype, uint64_t *total);
int aukit_audio_layout(const aukit_audio *, uint64_t *lens, uint64_t *row_off, uint64_t *row_stride);
int aukit_audio_download(aukit_ctx *, const aukit_audio *, double *dst); void aukit_audio_free(aukit_audio *);
int aukit_decode(aukit_ctx *, const aukit_batch *, const aukit_codec_desc *, int dtype, aukit_audio **out);
int aukit_resample(aukit_ctx *, const aukit_audio *, double new_rate, int interp, aukit_audio **out);
int aukit_mono(aukit_ctx *, const aukit_audio *, aukit_audio **out);
int aukit_mix(aukit_ctx *, const aukit_audio *const *audios, int count, double amplifier, aukit_audio **out);
int aukit_effect(aukit_ctx *, aukit_audio *inout, int effect_id, const double *args, int nargs);
int aukit_dfpwm_encode(aukit_ctx *, const aukit_audio *, int interleaved, aukit_batch **out);
int aukit_encode_pcm(aukit_ctx *, const aukit_audio *, int bit_depth, int data_type, int interleaved, aukit_audio **out);
int aukit_concat(aukit_ctx *, const aukit_audio *const *audios, uint32_t count, aukit_audio **out);
int aukit_sub(aukit_ctx *, const aukit_audio *, double start, double last, aukit_audio **out);
int aukit_combine(aukit_ctx *, const aukit_audio *const *audios, uint32_t count, aukit_audio **out);
int aukit_split(aukit_ctx *, const aukit_audio *, const int32_t *channels, uint32_t count, aukit_audio **out);
int aukit_rep(aukit_ctx *, const aukit_audio *, double count, aukit_audio **out);
int aukit_reverse(aukit_ctx *, const aukit_audio *, aukit_audio **out);
int aukit_decode_table(aukit_ctx *, const double *values, const uint64_t *offsets, uint32_t n, const aukit_codec_desc *d, aukit_audio **out);
typedef struct aukit_group aukit_group;
int aukit_partition(const uint64_t *sizes, uint32_t n, uint32_t world, uint32_t *cuts);
int aukit_group_create(aukit_group **out, const int *devices, uint32_t n_devices); void aukit_group_destroy(aukit_group *g);
aukit_ctx *aukit_group_ctx(aukit_group *g, uint32_t rank); int aukit_group_sync(aukit_group *g);
int aukit_group_scatter(aukit_group *g, uint32_t root, const aukit_batch *whole, aukit_batch **shards, uint32_t *cuts);
int aukit_group_gather_audio(aukit_group *g, uint32_t root, aukit_audio *const *parts, aukit_audio **whole);
int aukit_group_gather_batch(aukit_group *g, uint32_t root, aukit_batch *const *parts, aukit_batch **whole);
typedef struct {
    int32_t op;
    int32_t dtype, interp, mono;
    const aukit_batch *batch;
    const aukit_codec_desc *desc;
    aukit_audio *audio;
    aukit_audio **out_audio;
    aukit_batch **out_batch;
    aukit_chunks **out_chunks;
    double new_rate;
    int32_t effect_id, nargs;
    double args[8];
    int32_t channels, interleaved;
    int32_t bit_depth, data_type;
} aukit_group_call;
int aukit_group_run(aukit_group *g, const aukit_group_call *calls, uint32_t n_per_member);
int aukit_decode_nibbles(aukit_ctx *, const uint8_t *nibbles, const uint64_t *offsets, uint32_t n, const aukit_codec_desc *d, int dtype, aukit_audio **out);
int aukit_stream_decode_table(aukit_ctx *, const double *values, const uint64_t *offsets, uint32_t n, const aukit_codec_desc *d, int interp, int mono, int dtype, aukit_audio **out, aukit_chunks **chunks);
int aukit_tone(aukit_ctx *, uint32_t n, double frequency, double duration, double amplitude, int wave, double duty, int channels, double sample_rate, int dtype, aukit_audio **out);
int aukit_noise(aukit_ctx *, uint32_t n, double duration, double amplitude, int channels, double sample_rate, uint64_t seed, int dtype, aukit_audio **out);
int aukit_pack_pcm(aukit_ctx *, const aukit_audio *, int bit_depth, int data_type, int big_endian, int interleaved, int int_mode, aukit_batch **out);
int aukit_stream_decode(aukit_ctx *, const aukit_batch *, const aukit_codec_desc *, int interp, int mono, int dtype, aukit_audio **out, aukit_chunks **chunks);
int aukit_chunks_info(const aukit_chunks *, uint32_t *n, uint32_t *max_chunks);
int aukit_chunks_get(const aukit_chunks *, uint32_t *nchunks, uint32_t *lens, double *pos, int32_t *status, double *length_seconds); void aukit_chunks_free(aukit_chunks *);
typedef struct aukit_stream aukit_stream;
int aukit_stream_open(aukit_ctx *, const aukit_codec_desc *, int interp, int mono, int dtype, aukit_stream **out);
int aukit_stream_feed(aukit_stream *, const uint8_t *bytes, uint64_t n); int aukit_stream_finish(aukit_stream *);
int aukit_stream_next(aukit_stream *, double *dst, uint64_t dst_elems, uint32_t cap, uint32_t *len, int32_t *channels, double *pos, int32_t *state);
int aukit_stream_length(aukit_stream *, double *seconds); void aukit_stream_close(aukit_stream *);
]]

local C = ffi.load(os.getenv("AUKIT_HIP_LIB") or "aukit_hip")
local F64, I8 = 0, 2
local INTERP = {none = 0, linear = 1, cubic = 2, sinc = 3}
local DTYPE = {signed = 0, unsigned = 1, float = 2}
local DTYPE_NAME = {[0] = "signed", "unsigned", "float"}
local CODEC = {pcm = 0, g711 = 1, adpcm = 2, adpcm_wav = 3, msadpcm = 4, dfpwm = 5, mdfpwm = 6, qoa = 7, flac = 8}
local WAVDT = {[0] = "signed", "unsigned", "float", "alaw", "ulaw", "adpcm", "msadpcm", "dfpwm"}
local FX = {amplify = 0, speed = 1, fade = 2, invert = 3, normalize = 4, center = 5, trim = 6, delay = 7, echo = 8, reverb = 9, lowpass = 10, highpass = 11}

local aukit = {_VERSION = "1.10.0", defaultInterpolation = "linear", effects = {}, stream = {}}

-- cc.expect (aukit.lua:84): "bad argument #n (expected T, got U)" / "(number outside of range)", raised at the caller's level
local function expect(n, v, ...)
    local t = type(v)
    for i = 1, select("#", ...) do if t == select(i, ...) then return v end end
    local names = {...}
    local want = #names < 3 and table.concat(names, " or ") or (table.concat(names, ", ", 1, #names - 1) .. ", or " .. names[#names])
    error(("bad argument #%d (expected %s, got %s)"):format(n, want, t), 3)
end
local function range(n, v, lo, hi)
    if v ~= v or v < (lo or -math.huge) or v > (hi or math.huge) then error(("bad argument #%d (number outside of range)"):format(n), 3) end
    return v
end

local unpack = table.unpack or unpack  -- LuaJIT is Lua 5.1: no table.unpack, no string.pack
-- little-endian integer fields of a RIFF header without string.pack
local function u16(v) v = math.floor(v) % 65536 return string.char(v % 256, math.floor(v / 256)) end
local function u32(v) v = math.floor(v) % 4294967296 return string.char(v % 256, math.floor(v / 256) % 256, math.floor(v / 65536) % 256, math.floor(v / 16777216)) end

local ctxp = ffi.new("aukit_ctx*[1]")
local function ctx()
    if ctxp[0] == nil and C.aukit_ctx_create(ctxp, 0) ~= 0 then error(ffi.string(C.aukit_last_error()), 3) end
    return ctxp[0]
end
-- every native status becomes the Lua error the reference raises (AUKIT_E_ARG / AUKIT_E_LUA carry the reference's own message)
local function check(rc) if rc ~= 0 then error(ffi.string(C.aukit_last_error()), 3) end end

local function upload(str, first, len)
    first = first or 0
    len = len or #str
    local off = ffi.new("uint64_t[2]", 0, len)
    local b = ffi.new("aukit_batch*[1]")
    check(C.aukit_batch_upload(ctx(), b, ffi.cast("const uint8_t*", str) + first, off, 1))
    return ffi.gc(b[0], C.aukit_batch_free)
end
local function batch_string(b)
    local n, tot = ffi.new("uint32_t[1]"), ffi.new("uint64_t[1]")
    check(C.aukit_batch_info(b, n, tot))
    local buf = ffi.new("uint8_t[?]", math.max(tonumber(tot[0]), 1))
    check(C.aukit_batch_download(ctx(), b, buf))
    return ffi.string(buf, tot[0])
end

local function desc(t)
    local d = ffi.new("aukit_codec_desc")
    d.codec = CODEC[t.codec]; d.channels = t.channels or 1; d.sample_rate = t.sampleRate or 48000
    d.bit_depth = t.bitDepth or 8; d.data_type = DTYPE[t.dataType or "signed"]; d.big_endian = t.bigEndian and 1 or 0
    d.interleaved = t.interleaved == false and 0 or 1; d.ulaw = t.ulaw and 1 or 0; d.top_first = t.topFirst == false and 0 or 1
    d.block_align = t.blockAlign or 0
    if t.coefficients then  -- aukit.msadpcm's `coefficients` = {{c1...}, {c2...}} (aukit.lua:1304)
        local n = #t.coefficients[1]
        if n > 32 then error("at most 32 MS-ADPCM coefficient pairs are supported", 3) end
        d.ncoef = n
        for i = 1, n do d.coef1[i - 1] = t.coefficients[1][i]; d.coef2[i - 1] = t.coefficients[2][i] end
    end
    for c = 1, 64 do  -- aukit.adpcm's predictor / step_index: a number or one per channel (:1190-1215)
        local p, s = t.predictor, t.step_index
        if type(p) == "table" then p = p[c] end
        if type(s) == "table" then s = s[c] end
        d.predictor[c - 1] = p or 0
        d.step_index[c - 1] = s or 0
    end
    return d
end

--- Audio objects wrap a device handle.  `data` is downloaded on first access and cached; the in-place effects drop the cache.
--- Lua code that edits audio.data by hand must call aukit.upload(audio) before handing the object back to this module.
local Audio = {}
local Audio_mt = {__name = "Audio"}
local function wrap(h, metadata, info)
    return setmetatable({_h = ffi.gc(h, C.aukit_audio_free), metadata = metadata or {}, info = info or {}}, Audio_mt)
end
local function geom(self)
    local n, ch, rate, dt, tot = ffi.new("uint32_t[1]"), ffi.new("int[1]"), ffi.new("double[1]"), ffi.new("int[1]"), ffi.new("uint64_t[1]")
    check(C.aukit_audio_info(self._h, n, ch, rate, dt, tot))
    local len = ffi.new("uint64_t[1]")
    check(C.aukit_audio_layout(self._h, len, nil, nil))
    return ch[0], rate[0], tonumber(len[0])
end
-- the samples of one audio as ONE FFI array [channel][i] (a memcpy from pinned staging, no Lua tables yet)
local function fetch(self)
    local ch, _, len = geom(self)
    local buf = ffi.new("double[?]", math.max(ch * len, 1))
    check(C.aukit_audio_download(ctx(), self._h, buf))
    return buf, ch, len
end
function Audio_mt.__index(self, k)
    if k == "sampleRate" then local _, r = geom(self) return r end
    if k == "data" then
        local buf, ch, len = fetch(self)
        local data = {}
        for c = 1, ch do local t = {} for i = 1, len do t[i] = buf[(c - 1) * len + i - 1] end data[c] = t end
        rawset(self, "data", data)
        return data
    end
    return Audio[k]
end
local function expectAudio(n, var)
    if type(var) == "table" and getmetatable(var) == Audio_mt then return var end
    expect(n, var, "Audio")  -- always fails (aukit.lua:234-237)
end
--- pushes a hand-edited audio.data back to the device (no reference counterpart: there the table IS the audio)
function aukit.upload(audio)
    expectAudio(1, audio)
    local data = rawget(audio, "data")
    if not data then return audio end
    local ch, len = #data, #data[1]
    local buf = ffi.new("double[?]", math.max(ch * len, 1))
    for c = 1, ch do for i = 1, len do buf[(c - 1) * len + i - 1] = data[c][i] end end
    local lens = ffi.new("uint64_t[1]", len)
    local o = ffi.new("aukit_audio*[1]")
    check(C.aukit_audio_upload(ctx(), o, buf, lens, 1, ch, audio.sampleRate, F64))
    audio._h = ffi.gc(o[0], C.aukit_audio_free)
    return audio
end

function Audio:len() local _, r, len = geom(self) return len / r end   -- :638
function Audio:channels() return (geom(self)) end                      -- :645
function Audio:resample(sampleRate, interpolation)                     -- :653
    expect(1, sampleRate, "number")
    interpolation = expect(2, interpolation, "string", "nil") or aukit.defaultInterpolation
    if not INTERP[interpolation] then error("bad argument #2 (invalid interpolation type)", 2) end
    local o = ffi.new("aukit_audio*[1]")
    check(C.aukit_resample(ctx(), self._h, sampleRate, INTERP[interpolation], o))
    return wrap(o[0], self.metadata, self.info)
end
function Audio:mono()                                                  -- :677
    local o = ffi.new("aukit_audio*[1]")
    check(C.aukit_mono(ctx(), self._h, o))
    return wrap(o[0], self.metadata, self.info)
end
-- group calls want equal sample rates: resample first, like :702 / :756 / :812; the returned list keeps the temporaries alive
local function group(self, first_argn, ...)
    local audios = {self, ...}
    local arr = ffi.new("const aukit_audio*[?]", #audios)
    for i, a in ipairs(audios) do
        if i > 1 then expectAudio(first_argn + i - 2, a) end
        if a.sampleRate ~= self.sampleRate then a = a:resample(self.sampleRate) audios[i] = a end
        arr[i - 1] = a._h
    end
    return arr, #audios, audios
end
function Audio:mix(amplifier, ...)                                     -- :804
    local arr, n, keep
    if type(amplifier) == "number" then arr, n, keep = group(self, 2, ...)
    else arr, n, keep = group(self, 1, amplifier, ...) amplifier = 1 end
    local o = ffi.new("aukit_audio*[1]")
    check(C.aukit_mix(ctx(), arr, n, amplifier, o))
    keep = nil
    return wrap(o[0], self.metadata, self.info)
end
function Audio:concat(...)                                             -- :695
    local arr, n, keep = group(self, 1, ...)
    local o = ffi.new("aukit_audio*[1]")
    check(C.aukit_concat(ctx(), arr, n, o))
    keep = nil
    return wrap(o[0], self.metadata, self.info)
end
function Audio:combine(...)                                            -- :751
    local arr, n, keep = group(self, 1, ...)
    local o = ffi.new("aukit_audio*[1]")
    check(C.aukit_combine(ctx(), arr, n, o))
    keep = nil
    return wrap(o[0], self.metadata, self.info)
end
function Audio:sub(start, last)                                        -- :725
    start = expect(1, start, "number", "nil") or 0
    last = expect(2, last, "number", "nil") or 0
    local o = ffi.new("aukit_audio*[1]")
    check(C.aukit_sub(ctx(), self._h, start, last, o))
    return wrap(o[0], self.metadata, self.info)
end
function Audio:split(...)                                              -- :781
    local res = {}
    for n, cl in ipairs {...} do
        expect(n, cl, "table")
        if #cl == 0 then error("bad argument #" .. n .. " (cannot use empty table)", 2) end
        local ch = ffi.new("int32_t[?]", #cl, cl)
        local o = ffi.new("aukit_audio*[1]")
        check(C.aukit_split(ctx(), self._h, ch, #cl, o))
        res[#res + 1] = wrap(o[0], self.metadata, self.info)
    end
    return unpack(res)
end
function Audio:rep(count)                                              -- :839
    expect(1, count, "number")
    local o = ffi.new("aukit_audio*[1]")
    check(C.aukit_rep(ctx(), self._h, count, o))
    return wrap(o[0], self.metadata, self.info)
end
function Audio:reverse()                                               -- :856
    local o = ffi.new("aukit_audio*[1]")
    check(C.aukit_reverse(ctx(), self._h, o))
    return wrap(o[0], self.metadata, self.info)
end

local function check_pcm_args(bitDepth, dataType)  -- :908-910, :925-927
    if bitDepth ~= 8 and bitDepth ~= 16 and bitDepth ~= 24 and bitDepth ~= 32 then error("bad argument #2 (invalid bit depth)", 3) end
    if dataType ~= "signed" and dataType ~= "unsigned" and dataType ~= "float" then error("bad argument #3 (invalid data type)", 3) end
    if dataType == "float" and bitDepth ~= 32 then error("bad argument #2 (float audio must have 32-bit depth)", 3) end
end
-- encodePCM (:868-894) on the device: d * (d < 0 and maxValue or maxValue - 1) + add, as doubles in the reference's order
local function encoded(self, bitDepth, dataType, interleaved)
    local o = ffi.new("aukit_audio*[1]")
    check(C.aukit_encode_pcm(ctx(), self._h, bitDepth, DTYPE[dataType], interleaved and 1 or 0, o))
    local tmp = wrap(o[0])
    local buf, _, len = fetch(tmp)  -- one row of channels × length values
    return buf, len
end
function Audio:pcm(bitDepth, dataType, interleaved)                    -- :901
    bitDepth = expect(1, bitDepth, "number", "nil") or 8
    dataType = expect(2, dataType, "string", "nil") or "signed"
    expect(3, interleaved, "boolean", "nil")
    if interleaved == nil then interleaved = true end
    check_pcm_args(bitDepth, dataType)
    local buf, len = encoded(self, bitDepth, dataType, interleaved)
    local t = {}
    for i = 1, len do t[i] = buf[i - 1] end
    return t
end
function Audio:stream(chunkSize, bitDepth, dataType)                   -- :921
    chunkSize = expect(1, chunkSize, "number", "nil") or 131072
    bitDepth = expect(2, bitDepth, "number", "nil") or 8
    dataType = expect(3, dataType, "string", "nil") or "signed"
    check_pcm_args(bitDepth, dataType)
    local ch, rate, len = geom(self)
    local buf = encoded(self, bitDepth, dataType, false)  -- channel after channel: buf[(c - 1) * len + n - 1]
    local pos = 1
    return function()
        if pos == nil or pos > len then pos = nil return nil end  -- :878
        local p = pos / rate
        local chunk = {}
        for c = 1, ch do
            local t = {}
            for n = pos, math.min(pos + chunkSize - 1, len) do t[n - pos + 1] = buf[(c - 1) * len + n - 1] end
            chunk[c] = t
        end
        pos = pos + chunkSize
        return chunk, p
    end, len / rate
end
function Audio:dfpwm(interleaved)                                      -- :1005
    expect(1, interleaved, "boolean", "nil")
    local b = ffi.new("aukit_batch*[1]")
    check(C.aukit_dfpwm_encode(ctx(), self._h, interleaved == false and 0 or 1, b))
    local s = batch_string(b[0])
    C.aukit_batch_free(b[0])
    return s
end
-- packed sample bytes (aukit.pack ∘ Audio:pcm).  int_mode 0 = truncate like the CC: Tweaked VM's string.pack (a Java long cast); 1 = floor; 2 = PUC
-- Lua 5.3's "number has no integer representation" error — what string.pack does with a non-integer is the VM's business (include/aukit_hip.h)
local function packed(self, bitDepth, dataType, bigEndian, interleaved)
    local b = ffi.new("aukit_batch*[1]")
    check(C.aukit_pack_pcm(ctx(), self._h, bitDepth, DTYPE[dataType], bigEndian and 1 or 0, interleaved and 1 or 0, aukit.packIntegerMode or 0, b))
    local s = batch_string(b[0])
    C.aukit_batch_free(b[0])
    return s
end
local wavExtensibleChannels = {0x04, 0x03, 0x07, 0x33, 0x37, 0x3F, 0x637, 0x63F, 0x50F7, 0x50FF, 0x56F7, 0x56FF}  -- :141-154 (by channel count)
local dfpwmGUID = "\x3a\xc1\xfa\x38\x81\x1d\x43\x61\xa4\x0d\xce\x53\xca\x60\x7c\xd1"                          -- :138
-- wavMetadata :198-220 inverted: the tag Audio:wav writes for a metadata key (the reference finds it by walking pairs(wavMetadata), whose order a
-- Lua VM does not define — trackNumber has two tags, IPRT and ITRK: the first of the table here)
local wavTag = {album = "IPRD", title = "INAM", artist = "IART", author = "IWRI", composer = "IMUS", producer = "IPRO", trackNumber = "IPRT",
    trackCount = "IFRM", partNumber = "PRT1", partCount = "PRT2", length = "TLEN", rating = "IRTD", date = "ICRD", encodedBy = "ITCH",
    encoder = "ISFT", media = "ISRF", genre = "IGNR", comment = "ICMT", copyright = "ICOP", language = "ILNG"}
--- "LIST" .. s4("INFO" .. (tag .. s4(tostring(value)) .. pad to even)...) for self.metadata (:946-956): str_pack("!2<c4" .. ("c4s4Xh"):rep(n), "INFO", ...)
local function wav_list(self)
    if not self.metadata or not next(self.metadata) then return "" end
    local parts = {"INFO"}
    local at = 4
    for k, v in pairs(self.metadata) do
        local tag = wavTag[k]
        if tag then
            local val = tostring(v)
            parts[#parts + 1] = tag .. u32(#val) .. val
            at = at + 8 + #val
            if at % 2 == 1 then parts[#parts + 1] = "\0" at = at + 1 end  -- Xh
        end
    end
    local list = table.concat(parts)
    return "LIST" .. u32(#list) .. list
end
function Audio:wav(bitDepth)                                           -- :940-997
    bitDepth = expect(1, bitDepth, "number", "nil") or 16
    local ch, rate, len = geom(self)
    local list = wav_list(self)  -- (the RIFF size field does not count it, with or without: `#str + 72` / `#str + 36`, as the reference writes it)
    if bitDepth == 1 then
        local str = self:dfpwm(true)
        -- "<c4Ic4c4IHHIIHHHHIc16c4II[c4s4]c4I" (:957-967, :969-975)
        return "RIFF" .. u32(#str + 72) .. "WAVE" .. "fmt " .. u32(40) .. u16(0xFFFE) .. u16(ch) .. u32(rate) .. u32(rate * ch / 8) .. u16(math.ceil(ch / 8)) .. u16(1) ..
            u16(22) .. u16(1) .. u32(wavExtensibleChannels[ch] or 0) .. dfpwmGUID .. "fact" .. u32(4) .. u32(len) .. list .. "data" .. u32(#str) .. str
    elseif bitDepth ~= 8 and bitDepth ~= 16 and bitDepth ~= 24 and bitDepth ~= 32 then error("bad argument #2 (invalid bit depth)", 2) end
    local str = packed(self, bitDepth, bitDepth == 8 and "unsigned" or "signed", false, true)
    -- "<c4Ic4c4IHHIIHH[c4s4]c4I" (:991-995)
    return "RIFF" .. u32(#str + 36) .. "WAVE" .. "fmt " .. u32(16) .. u16(1) .. u16(ch) .. u32(rate) .. u32(rate * ch * bitDepth / 8) .. u16(ch * bitDepth / 8) .. u16(bitDepth) ..
        list .. "data" .. u32(#str) .. str
end

-- ---------------------------------------------------------------- generators, packing
local WAVE = {sine = 1, triangle = 2, sawtooth = 3, square = 4}
function aukit.new(duration, channels, sampleRate)                     -- :1783
    expect(1, duration, "number")
    channels = expect(2, channels, "number", "nil") or 1
    sampleRate = expect(3, sampleRate, "number", "nil") or 48000
    range(2, channels, 1) range(3, sampleRate, 1)
    local o = ffi.new("aukit_audio*[1]")
    check(C.aukit_tone(ctx(), 1, 0, duration, 1, 0, 0.5, channels, sampleRate, F64, o))
    return wrap(o[0], {}, {})
end
function aukit.tone(frequency, duration, amplitude, waveType, duty, channels, sampleRate)  -- :1808
    expect(1, frequency, "number") expect(2, duration, "number")
    amplitude = expect(3, amplitude, "number", "nil") or 1
    waveType = expect(4, waveType, "string", "nil") or "sine"
    duty = expect(5, duty, "number", "nil") or 0.5
    channels = expect(6, channels, "number", "nil") or 1
    sampleRate = expect(7, sampleRate, "number", "nil") or 48000
    local w = WAVE[waveType]
    if not w then error("bad argument #4 (invalid wave type)", 2) end
    local o = ffi.new("aukit_audio*[1]")
    check(C.aukit_tone(ctx(), 1, frequency, duration, amplitude, w, duty, channels, sampleRate, F64, o))
    return wrap(o[0], {}, {})
end
--- aukit.noise(duration, amplitude, channels, sampleRate) :1840.  The reference draws from math.random; here the device draws (Philox4x32-10,
--- aukit_noise) from a seed: a fifth argument names it (the same seed, the same audio), otherwise every call draws a fresh one from math.random.
local noise_calls = 0
function aukit.noise(duration, amplitude, channels, sampleRate, seed)
    expect(1, duration, "number")
    amplitude = expect(2, amplitude, "number", "nil") or 1
    channels = expect(3, channels, "number", "nil") or 1
    sampleRate = expect(4, sampleRate, "number", "nil") or 48000
    range(2, amplitude, 0, 1) range(3, channels, 1) range(4, sampleRate, 1)
    if seed == nil then
        noise_calls = noise_calls + 1
        seed = math.floor(math.random() * 4294967296) * 4294967296 + math.floor(math.random() * 4294967296) + noise_calls
    end
    local o = ffi.new("aukit_audio*[1]")
    check(C.aukit_noise(ctx(), 1, duration, amplitude, channels, sampleRate, ffi.cast("uint64_t", seed), F64, o))
    return wrap(o[0], {}, {})
end
--- aukit.pack(data, bitDepth, dataType, bigEndian) :1861 takes the number table Audio:pcm returns; with an Audio as first argument the
--- round trip through Lua tables is skipped (the device packs its own encodePCM output)
function aukit.pack(data, bitDepth, dataType, bigEndian)
    bitDepth = expect(2, bitDepth, "number", "nil") or 8
    dataType = expect(3, dataType, "string", "nil") or "signed"
    check_pcm_args(bitDepth, dataType)
    if type(data) == "table" and getmetatable(data) == Audio_mt then return packed(data, bitDepth, dataType, bigEndian, true) end
    expect(1, data, "string", "table")
    if type(data) == "string" then return data end                    -- :1867
    -- a table of numbers (what Audio:pcm returned): upload it as one row and let the device pack it with the same rules
    local n = #data
    local buf = ffi.new("double[?]", math.max(n, 1))
    for i = 1, n do buf[i - 1] = data[i] end
    local lens, o = ffi.new("uint64_t[1]", n), ffi.new("aukit_audio*[1]")
    check(C.aukit_audio_upload(ctx(), o, buf, lens, 1, 1, 48000, F64))
    local row = wrap(o[0])
    -- the numbers are already encodePCM's output (:868-894): pack them as they are — a "float" pass-through of bitDepth-wide integers is what
    -- aukit_pack_pcm does when its input is flagged pre-encoded (int_mode + 8)
    local b = ffi.new("aukit_batch*[1]")
    check(C.aukit_pack_pcm(ctx(), row._h, bitDepth, DTYPE[dataType], bigEndian and 1 or 0, 1, (aukit.packIntegerMode or 0) + 8, b))
    local str = batch_string(b[0])
    C.aukit_batch_free(b[0])
    return str
end

-- ---------------------------------------------------------------- loaders
local function loader(d, data, info, first, len)
    local o = ffi.new("aukit_audio*[1]")
    check(C.aukit_decode(ctx(), upload(data, first, len), d, F64, o))
    return wrap(o[0], {}, info)
end
function aukit.pcm(data, bitDepth, dataType, channels, sampleRate, interleaved, bigEndian)  -- :1049
    expect(1, data, "string", "table")
    bitDepth = expect(2, bitDepth, "number", "nil") or 8
    dataType = expect(3, dataType, "string", "nil") or "signed"
    channels = expect(4, channels, "number", "nil") or 1
    sampleRate = expect(5, sampleRate, "number", "nil") or 48000
    expect(6, interleaved, "boolean", "nil") expect(7, bigEndian, "boolean", "nil")
    if type(data) == "table" then  -- :1077-1096: the numbers as they are, one double each
        local n = #data
        local vals = ffi.new("double[?]", math.max(n, 1))
        for i = 1, n do vals[i - 1] = data[i] end
        local offs = ffi.new("uint64_t[2]", {0, n})
        local o = ffi.new("aukit_audio*[1]")
        check(C.aukit_decode_table(ctx(), vals, offs, 1, desc {codec = "pcm", bitDepth = bitDepth, dataType = dataType, channels = channels, sampleRate = sampleRate,
            interleaved = interleaved, bigEndian = bigEndian}, o))
        return wrap(o[0], {}, {bitDepth = bitDepth, dataType = dataType})
    end
    return loader(desc {codec = "pcm", bitDepth = bitDepth, dataType = dataType, channels = channels, sampleRate = sampleRate, interleaved = interleaved, bigEndian = bigEndian},
        data, {bitDepth = bitDepth, dataType = dataType})
end
function aukit.adpcm(data, channels, sampleRate, topFirst, interleaved, predictor, step_index)  -- :1183 (string input)
    expect(1, data, "string", "table")
    channels = expect(2, channels, "number", "nil") or 1
    sampleRate = expect(3, sampleRate, "number", "nil") or 48000
    expect(4, topFirst, "boolean", "nil") expect(5, interleaved, "boolean", "nil")
    expect(6, predictor, "number", "table", "nil") expect(7, step_index, "number", "table", "nil")
    if type(data) == "table" then  -- :1232-1238: one nibble per entry, `len = #data / channels`
        local n = #data
        local nib = ffi.new("uint8_t[?]", math.max(n, 1))
        for i = 1, n do
            local v = data[i]
            if type(v) ~= "number" or v % 1 ~= 0 or v < 0 or v > 15 then error("attempt to perform arithmetic on a nil value (field '?')", 2) end  -- ima_index_table[nibble]
            nib[i - 1] = v
        end
        local offs = ffi.new("uint64_t[2]", {0, n})
        local o = ffi.new("aukit_audio*[1]")
        check(C.aukit_decode_nibbles(ctx(), nib, offs, 1, desc {codec = "adpcm", channels = channels, sampleRate = sampleRate, interleaved = interleaved,
            predictor = predictor, step_index = step_index}, F64, o))
        return wrap(o[0], {}, {bitDepth = 16, dataType = "signed"})
    end
    return loader(desc {codec = "adpcm", channels = channels, sampleRate = sampleRate, topFirst = topFirst, interleaved = interleaved, predictor = predictor, step_index = step_index},
        data, {bitDepth = 16, dataType = "signed"})
end
function aukit.msadpcm(data, blockAlign, channels, sampleRate, coefficients)  -- :1283
    expect(1, data, "string") expect(2, blockAlign, "number")
    channels = expect(3, channels, "number", "nil") or 1
    sampleRate = expect(4, sampleRate, "number", "nil") or 48000
    expect(5, coefficients, "table", "nil")
    return loader(desc {codec = "msadpcm", blockAlign = blockAlign, channels = channels, sampleRate = sampleRate, coefficients = coefficients}, data, {bitDepth = 16, dataType = "signed"})
end
function aukit.g711(data, ulaw, channels, sampleRate)                  -- :1361
    expect(1, data, "string") expect(2, ulaw, "boolean")
    channels = expect(3, channels, "number", "nil") or 1
    sampleRate = expect(4, sampleRate, "number", "nil") or 8000
    return loader(desc {codec = "g711", ulaw = ulaw, channels = channels, sampleRate = sampleRate}, data, {bitDepth = 8, dataType = "signed"})
end
function aukit.dfpwm(data, channels, sampleRate)                       -- :1392
    expect(1, data, "string")
    channels = expect(2, channels, "number", "nil") or 1
    sampleRate = expect(3, sampleRate, "number", "nil") or 48000
    return loader(desc {codec = "dfpwm", channels = channels, sampleRate = sampleRate}, data, {bitDepth = 8, dataType = "signed"})
end
function aukit.mdfpwm(data, head)                                      -- :1420 (artist / title / album strings: read them in Lua as :1423-1426 does)
    expect(1, data, "string")
    if data:sub(1, 7) ~= "MDFPWM\003" then error("bad argument #1 (not a MDFPWM file)", 2) end
    return loader(desc {codec = "mdfpwm"}, data, {bitDepth = 8, dataType = "signed"})
end
function aukit.qoa(data, head) expect(1, data, "string") return loader(desc {codec = "qoa"}, data, {bitDepth = 16, dataType = "signed"}) end   -- :1706
function aukit.flac(data, head) expect(1, data, "string") return loader(desc {codec = "flac"}, data, {dataType = "signed"}) end               -- :1657

-- container front-ends: the header walk is the library's (aukit_parse_container = aukit.lua:1456-1651 / :2927-3113, host-side C),
-- the payload range goes to the loader without a copy on the Lua side
local function container(data, kind, stream)
    local c = ffi.new("aukit_container")
    check(C.aukit_parse_container(ffi.cast("const uint8_t*", data), #data, kind, stream and 1 or 0, c))
    return c
end
local function pcm_info(d) return d.codec == 0 and {bitDepth = d.bit_depth, dataType = DTYPE_NAME[d.data_type]} or {bitDepth = 8, dataType = "signed"} end
function aukit.wav(data, head)                                         -- :1456
    expect(1, data, "string")
    local c = container(data, 0, false)
    if head then return aukit.new(0, c.desc.channels, c.desc.sample_rate) end
    return loader(c.desc, data, {dataType = WAVDT[c.wav_data_type], bitDepth = c.bit_depth}, tonumber(c.payload_off), tonumber(c.payload_len))
end
function aukit.aiff(data, head)                                        -- :1580
    expect(1, data, "string")
    local c = container(data, 1, false)
    if head then return aukit.new(0, c.desc.channels, c.desc.sample_rate) end
    return loader(c.desc, data, pcm_info(c.desc), tonumber(c.payload_off), tonumber(c.payload_len))
end
function aukit.au(data)                                                -- :1639
    expect(1, data, "string")
    local c = container(data, 2, false)
    return loader(c.desc, data, pcm_info(c.desc), tonumber(c.payload_off), tonumber(c.payload_len))
end

-- ---------------------------------------------------------------- effects (in place, return the same object: :3356-3618)
local FX_ARGS = {  -- argument checks of the reference, in its order: {type, optional}
    amplify = {{"number"}}, speed = {{"number"}}, fade = {{"number"}, {"number"}, {"number"}, {"number"}}, invert = {}, normalize = {{"number", true}, {"boolean", true}},
    center = {}, trim = {{"number", true}}, delay = {{"number"}, {"number", true}}, echo = {{"number", true}, {"number", true}},
    reverb = {{"number", true}, {"number", true}, {"number", true}, {"number", true}}, lowpass = {{"number"}}, highpass = {{"number"}},
}
for name, id in pairs(FX) do
    aukit.effects[name] = function(audio, ...)
        expectAudio(1, audio)
        local args = {...}
        for i, spec in ipairs(FX_ARGS[name]) do
            if spec[2] then expect(i + 1, args[i], spec[1], "nil") else expect(i + 1, args[i], spec[1]) end
        end
        if name == "speed" then args[2] = INTERP[aukit.defaultInterpolation] end
        if name == "normalize" then args[1] = args[1] or 1 args[2] = args[2] and 1 or 0 end
        local n = 0
        for i = 1, 4 do if args[i] ~= nil then n = i end end
        local a = ffi.new("double[?]", math.max(n, 1))
        for i = 1, n do a[i - 1] = args[i] or 0 end
        check(C.aukit_effect(ctx(), audio._h, id, a, n))
        rawset(audio, "data", nil)  -- drop the cached Lua copy: the device buffer changed in place
        return audio
    end
end

-- ---------------------------------------------------------------- stream factories (string input)
--- String input: every iterator call of the reference is computed in one launch; the samples come back as ONE FFI array and each call of the
--- returned closure builds only its own chunk's tables.  Reader-function input (austream's http / websocket / file sources): the library's
--- resumable handle — pieces are fed as the closure runs dry, and the chunks are those of the string version for the concatenated input.
local CHUNK_CAP = 1048576
local function streamer_fn(d, fn, first, mono, dtype)
    local hp = ffi.new("aukit_stream*[1]")
    check(C.aukit_stream_open(ctx(), d, INTERP[aukit.defaultInterpolation], mono and 1 or 0, dtype, hp))
    local h = ffi.gc(hp[0], C.aukit_stream_close)
    local function feed(s) check(C.aukit_stream_feed(h, ffi.cast("const uint8_t*", s), #s)) end
    feed(first)
    local buf_ch = 2
    local buf = ffi.new("double[?]", CHUNK_CAP * buf_ch)
    local len, ch, st, pos = ffi.new("uint32_t[1]"), ffi.new("int32_t[1]"), ffi.new("int32_t[1]"), ffi.new("double[1]")
    local done, ended = false, false
    local length = ffi.new("double[1]")
    check(C.aukit_stream_length(h, length))
    return function()
        while not ended do
            local rc = C.aukit_stream_next(h, buf, CHUNK_CAP * buf_ch, CHUNK_CAP, len, ch, pos, st)
            if rc == -1 and ch[0] > buf_ch and len[0] <= CHUNK_CAP then
                -- more channels than the buffer was sized for (the library refuses before it writes; the chunk is still there): grow, ask again
                buf_ch = ch[0]
                buf = ffi.new("double[?]", CHUNK_CAP * buf_ch)
                rc = C.aukit_stream_next(h, buf, CHUNK_CAP * buf_ch, CHUNK_CAP, len, ch, pos, st)
            end
            check(rc)
            if st[0] == 0 then      -- AUKIT_STREAM_CHUNK
                local chunk = {}
                for c = 1, ch[0] do
                    local t = {}
                    for i = 1, len[0] do t[i] = buf[(c - 1) * CHUNK_CAP + i - 1] end
                    chunk[c] = t
                end
                return chunk, pos[0]
            elseif st[0] == 2 then  -- AUKIT_STREAM_END
                ended = true
            else                    -- AUKIT_STREAM_NEED_INPUT
                local piece = (not done) and fn() or nil
                if piece == nil then done = true check(C.aukit_stream_finish(h)) else feed(piece) end
            end
        end
        return nil
    end, length[0]
end
local function streamer(d, data, mono, dtype, first, len)
    if type(data) == "function" then
        local piece = data()
        expect(1, piece, "string")
        return streamer_fn(d, data, piece, mono, dtype)
    end
    local o, ck = ffi.new("aukit_audio*[1]"), ffi.new("aukit_chunks*[1]")
    if type(data) == "table" then  -- aukit.stream.pcm on a table of numbers (:2255-2290)
        local n = #data
        local vals = ffi.new("double[?]", math.max(n, 1))
        for i = 1, n do vals[i - 1] = data[i] end
        check(C.aukit_stream_decode_table(ctx(), vals, ffi.new("uint64_t[2]", {0, n}), 1, d, INTERP[aukit.defaultInterpolation], mono and 1 or 0, dtype, o, ck))
    else
        check(C.aukit_stream_decode(ctx(), upload(data, first, len), d, INTERP[aukit.defaultInterpolation], mono and 1 or 0, dtype, o, ck))
    end
    local audio = wrap(o[0])
    local n, mx = ffi.new("uint32_t[1]"), ffi.new("uint32_t[1]")
    check(C.aukit_chunks_info(ck[0], n, mx))
    local m = math.max(mx[0], 1)
    local nch, lens, pos, status, length = ffi.new("uint32_t[1]"), ffi.new("uint32_t[?]", m), ffi.new("double[?]", m), ffi.new("int32_t[1]"), ffi.new("double[1]")
    check(C.aukit_chunks_get(ck[0], nch, lens, pos, status, length))
    C.aukit_chunks_free(ck[0])
    local buf, ch, total = fetch(audio)
    local k, off = 0, 0
    return function()
        if k >= nch[0] then
            -- the reference's iterator RAISES here (end of data inside a prefill / a malformed block): same message as the library's
            if status[0] == -2 then error("attempt to perform arithmetic on a nil value (field '?')", 2) end
            return nil
        end
        local chunk = {}
        for c = 1, ch do
            local t = {}
            for i = 1, lens[k] do t[i] = buf[(c - 1) * total + off + i - 1] end
            chunk[c] = t
        end
        off = off + lens[k]; k = k + 1
        return chunk, pos[k - 1]
    end, length[0]
end
function aukit.stream.pcm(data, bitDepth, dataType, channels, sampleRate, bigEndian, mono)  -- :2228
    expect(1, data, "string", "table", "function")
    bitDepth = expect(2, bitDepth, "number", "nil") or 8
    dataType = expect(3, dataType, "string", "nil") or "signed"
    channels = expect(4, channels, "number", "nil") or 1
    sampleRate = expect(5, sampleRate, "number", "nil") or 48000
    expect(6, bigEndian, "boolean", "nil") expect(7, mono, "boolean", "nil")
    return streamer(desc {codec = "pcm", bitDepth = bitDepth, dataType = dataType, channels = channels, sampleRate = sampleRate, bigEndian = bigEndian}, data, mono, F64)
end
function aukit.stream.g711(input, ulaw, channels, sampleRate, mono)    -- :2850
    expect(1, input, "string", "function") expect(2, ulaw, "boolean")
    return streamer(desc {codec = "g711", ulaw = ulaw, channels = channels, sampleRate = sampleRate or 8000}, input, mono, I8)
end
function aukit.stream.adpcm(input, blockAlign, channels, sampleRate, mono)    -- :2753
    expect(1, input, "string", "function") expect(2, blockAlign, "number")
    return streamer(desc {codec = "adpcm_wav", blockAlign = blockAlign, channels = channels, sampleRate = sampleRate}, input, mono, I8)
end
function aukit.stream.msadpcm(input, blockAlign, channels, sampleRate, mono, coefficients)  -- :2588
    expect(1, input, "string", "function") expect(2, blockAlign, "number") expect(6, coefficients, "table", "nil")
    return streamer(desc {codec = "msadpcm", blockAlign = blockAlign, channels = channels, sampleRate = sampleRate, coefficients = coefficients}, input, mono, I8)
end
function aukit.stream.dfpwm(data, sampleRate, channels, mono)          -- :2439
    expect(1, data, "string", "function")
    return streamer(desc {codec = "dfpwm", channels = channels, sampleRate = sampleRate}, data, mono, F64)
end
function aukit.stream.mdfpwm(data, mono) expect(1, data, "string", "function") return streamer(desc {codec = "mdfpwm"}, data, mono, I8) end  -- :2507
function aukit.stream.flac(data, mono) expect(1, data, "string", "function") return streamer(desc {codec = "flac"}, data, mono, F64) end     -- :3124
function aukit.stream.qoa(data, mono) expect(1, data, "string", "function") return streamer(desc {codec = "qoa"}, data, mono, F64) end       -- :3202
-- `ignoreHeader` in reader-function mode (:2983-2989, :3053-3060, :3097-3101): a LATER piece that starts with the container's magic loses its
-- header — cut with the reference's own patterns, so a header they do not fit fails the way it does there
local function strip_headers(fn, kind)
    return function()
        local d = fn()
        if not d then return nil end
        if kind == 0 then
            if d:match "^RIFF....WAVE" then return d:sub(d:match("^RIFF....WAVE.?data....()")) end
        elseif kind == 1 then
            if d:match "^FORM....AIF[FC]" then
                local n, p = d:match("^FORM....AIF[FC].-SSND(....)....()")
                return d:sub(p + ((">I"):unpack(n)))
            end
        elseif d:match "^.snd" then return d:sub(((">I"):unpack(d:sub(5, 8)))) end
        return d
    end
end
local function stream_container(data, kind, mono, ignoreHeader)
    expect(1, data, "string", "function")
    if type(data) == "function" then  -- "the first chunk MUST contain the ENTIRE header" (:2918): the walk runs on it, what follows the header is the first piece
        local fn, piece = data, data()
        expect(1, piece, "string")
        if ignoreHeader then fn = strip_headers(fn, kind) end
        local c = ffi.new("aukit_container")
        check(C.aukit_parse_container(ffi.cast("const uint8_t*", piece), #piece, kind, 2, c))
        local dtype = (c.desc.codec == 0 or c.desc.codec == 5) and F64 or I8
        local it, length = streamer_fn(c.desc, fn, piece:sub(tonumber(c.payload_off) + 1, tonumber(c.payload_off + c.payload_len)), mono, dtype)
        if c.length_seconds == c.length_seconds then length = c.length_seconds end
        return it, length
    end
    local c = container(data, kind, true)
    local dtype = (c.desc.codec == 0 or c.desc.codec == 5) and F64 or I8  -- what stream.pcm / .dfpwm vs .g711 / .adpcm / .msadpcm hand out
    local it, length = streamer(c.desc, data, mono, dtype, tonumber(c.payload_off), tonumber(c.payload_len))
    if c.length_seconds == c.length_seconds then length = c.length_seconds end  -- not NaN: the container factory computes its own (:2994-2996, :3064-3069, :3107-3113)
    return it, length
end
function aukit.stream.wav(data, mono, ignoreHeader) return stream_container(data, 0, mono, ignoreHeader) end    -- :2927
function aukit.stream.aiff(data, mono, ignoreHeader) expect(2, mono, "boolean", "nil") return stream_container(data, 1, mono, ignoreHeader) end  -- :3016
function aukit.stream.au(data, mono, ignoreHeader) expect(2, mono, "boolean", "nil") return stream_container(data, 2, mono, ignoreHeader) end    -- :3086

-- ---------------------------------------------------------------- several GPUs (no counterpart in the reference: include/aukit_hip.h, "several GPUs")
-- aukit.gpus{0, 1, ...} -> group.  group:map(strings, fn) runs fn(ctx_index, shard_batch, ctx) on every member's byte-balanced range of the
-- strings and returns the gathered result: fn returns an aukit_audio* (a loader / resample chain) or an aukit_batch* (Audio:dfpwm, the
-- transcode).  Streams are independent, so this is a scatter, N independent single-GPU runs, and a gather — nothing else crosses xGMI.
function aukit.gpus(devices)
    local n = #devices
    local devs = ffi.new("int[?]", n)
    for i = 1, n do devs[i - 1] = devices[i] end
    local g = ffi.new("aukit_group*[1]")
    check(C.aukit_group_create(g, devs, n))
    local group = {handle = ffi.gc(g[0], C.aukit_group_destroy), size = n}
    function group:ctx(rank) return C.aukit_group_ctx(self.handle, rank) end
    -- group:run(lists): lists[r + 1] = member r's calls, each a table {op = "decode_resample", batch = b, desc = d, new_rate = 48000, interp = "cubic",
    -- dtype = 1, out = ffi.new("aukit_audio*[1]")} / {op = "effect", audio = a, name = "highpass", args = {20}} / {op = "mono", audio = a, out = o} /
    -- {op = "dfpwm_transcode_mono", batch = b, channels = 2, out = ffi.new("aukit_batch*[1]")} ...  The members' lists run SIDE BY SIDE on the
    -- library's worker threads (aukit_group_run) — from this one Lua state, with no callback into it — where calling the entry points member
    -- after member from here would run the shards of a codec that reads counts back (FLAC, DFPWM, the ADPCMs, QOA) one after another.
    local GOP = {decode = 1, decode_resample = 2, stream_decode = 3, resample = 4, mono = 5, effect = 6, dfpwm_encode = 7, dfpwm_transcode_mono = 8, encode_pcm = 9, sync = 10}
    function group:run(lists)
        local per = 0
        for r = 1, n do per = math.max(per, #(lists[r] or {})) end
        local calls = ffi.new("aukit_group_call[?]", n * math.max(per, 1))
        for r = 1, n do
            for k, c in ipairs(lists[r] or {}) do
                local g = calls[(r - 1) * per + k - 1]
                g.op = GOP[c.op]
                g.dtype = c.dtype or F64; g.interp = INTERP[c.interp or "linear"]; g.mono = c.mono and 1 or 0
                g.batch = c.batch; g.desc = c.desc; g.audio = c.audio
                if c.op == "dfpwm_encode" or c.op == "dfpwm_transcode_mono" then g.out_batch = c.out else g.out_audio = c.out end
                g.out_chunks = c.chunks_out
                g.new_rate = c.new_rate or 0
                if c.op == "effect" then
                    g.effect_id = FX[c.name]; g.nargs = #(c.args or {})
                    for i, v in ipairs(c.args or {}) do g.args[i - 1] = v end
                end
                g.channels = c.channels or 1; g.interleaved = (c.interleaved == false) and 0 or 1
                g.bit_depth = c.bit_depth or 8; g.data_type = DTYPE[c.data_type or "signed"]
            end
        end
        check(C.aukit_group_run(self.handle, calls, per))
    end
    -- group:scatter(strings, root) -> shards (aukit_batch*[n], 0-based), cuts; keep the returned `whole` alive while the shards are in use
    function group:scatter(strings, root)
        root = root or 0
        local total, offs = 0, ffi.new("uint64_t[?]", #strings + 1)
        for i, s in ipairs(strings) do offs[i - 1] = total; total = total + #s end
        offs[#strings] = total
        local whole = ffi.new("aukit_batch*[1]")
        check(C.aukit_batch_upload(self:ctx(root), whole, ffi.cast("const uint8_t*", table.concat(strings)), offs, #strings))
        local shards, cuts = ffi.new("aukit_batch*[?]", n), ffi.new("uint32_t[?]", n + 1)
        check(C.aukit_group_scatter(self.handle, root, whole[0], shards, cuts))
        return shards, cuts, whole
    end
    function group:gather_batches(outs, root)   -- outs[r] = aukit_batch*[1] of member r - 1
        local parts, o = ffi.new("aukit_batch*[?]", n), ffi.new("aukit_batch*[1]")
        for r = 1, n do parts[r - 1] = outs[r][0] end
        check(C.aukit_group_gather_batch(self.handle, root or 0, parts, o))
        check(C.aukit_group_sync(self.handle))
        return o[0]
    end
    function group:gather_audios(outs, root)    -- outs[r] = aukit_audio*[1] of member r - 1
        local parts, o = ffi.new("aukit_audio*[?]", n), ffi.new("aukit_audio*[1]")
        for r = 1, n do parts[r - 1] = outs[r][0] end
        check(C.aukit_group_gather_audio(self.handle, root or 0, parts, o))
        check(C.aukit_group_sync(self.handle))
        return o[0]
    end
    function group:map(strings, fn, root)
        root = root or 0
        local total, offs = 0, ffi.new("uint64_t[?]", #strings + 1)
        for i, s in ipairs(strings) do offs[i - 1] = total; total = total + #s end
        offs[#strings] = total
        local whole = ffi.new("aukit_batch*[1]")
        check(C.aukit_batch_upload(self:ctx(root), whole, ffi.cast("const uint8_t*", table.concat(strings)), offs, #strings))
        local shards, cuts = ffi.new("aukit_batch*[?]", n), ffi.new("uint32_t[?]", n + 1)
        check(C.aukit_group_scatter(self.handle, root, whole[0], shards, cuts))
        local audios, batches, is_audio = ffi.new("aukit_audio*[?]", n), ffi.new("aukit_batch*[?]", n), nil
        for r = 0, n - 1 do
            local res, kind = fn(r, shards[r], self:ctx(r))   -- kind: "audio" | "batch"
            is_audio = kind ~= "batch"
            if is_audio then audios[r] = res else batches[r] = res end
        end
        local out
        if is_audio then
            local o = ffi.new("aukit_audio*[1]")
            check(C.aukit_group_gather_audio(self.handle, root, audios, o))
            out = o[0]
        else
            local o = ffi.new("aukit_batch*[1]")
            check(C.aukit_group_gather_batch(self.handle, root, batches, o))
            out = o[0]
        end
        check(C.aukit_group_sync(self.handle))
        for r = 0, n - 1 do
            C.aukit_batch_free(shards[r])
            if is_audio then C.aukit_audio_free(audios[r]) else C.aukit_batch_free(batches[r]) end
        end
        C.aukit_batch_free(whole[0])
        return out
    end
    return group
end

-- ---------------------------------------------------------------- aukit.detect  :2136-2195 (host-side bytes; LuaJIT has no string.unpack: eight numbers by hand)
local datafmts = {{1, 8, "signed"}, {1, 8, "unsigned"}, {2, 16, "signed"}, {4, 32, "signed"}, {4, 32, "float"}, {3, 24, "signed"}, {4, 32, "unsigned"},
    {3, 24, "unsigned"}, {2, 16, "unsigned"}}
--- pcall(str_unpack, fmt, data, init) for eight little-endian values of `size` bytes → the numbers, or nil where string.unpack raises
--- ("initial position out of string", "data string too short")
local function unpack8(data, size, kind, init)
    local n = #data
    if init < 0 then init = (-init > n) and 0 or n + init + 1 end
    if init < 1 or init - 1 > n then return nil end
    if init - 1 + 8 * size > n then return nil end
    local out = {}
    for k = 0, 7 do
        local p = init + k * size
        if kind == "float" then
            out[k + 1] = ffi.cast("const float*", ffi.cast("const uint8_t*", data) + (p - 1))[0]   -- (unaligned loads are fine on the hosts LuaJIT runs on)
        else
            local v = 0
            for b = size - 1, 0, -1 do v = v * 256 + data:byte(p + b) end
            if kind == "signed" and v >= 2 ^ (8 * size - 1) then v = v - 2 ^ (8 * size) end
            out[k + 1] = v
        end
    end
    return out
end
function aukit.detect(data)
    expect(1, data, "string")
    if data:match "^RIFF....WAVE" then return "wav"
    elseif data:match "^FORM....AIF[FC]" then return "aiff"
    elseif data:match "^%.snd" then return "au"
    elseif data:match "^fLaC" then return "flac"
    elseif data:match "^MDFPWM\3" then return "mdfpwm"
    elseif data:match "^qoaf" then return "qoa"
    end
    -- the start or the end of the audio is expected to be (near) silence
    for _, f in ipairs(datafmts) do
        local size, bits, kind = f[1], f[2], f[3]
        local mid = kind == "unsigned" and 2 ^ (bits - 1) or 0
        local gap = kind == "float" and 0.001 or 8 * 2 ^ (bits - 8)
        for _, init in ipairs {1, #data - bits} do   -- (:2183 subtracts the bit depth, not the byte count)
            local nums = unpack8(data, size, kind, init)
            if nums then
                local allzero, ok = true, true
                for _, v in ipairs(nums) do
                    if v ~= mid then allzero = false end
                    if v < mid - gap or v > mid + gap then ok = false break end
                end
                if ok and not allzero then return "pcm", bits, kind end
            end
        end
    end
    if data:find(("\x55"):rep(12), 1, true) or data:find(("\xAA"):rep(12), 1, true) then return "dfpwm" end
    return nil
end

return aukit

--- aukit.lua (aukit_amd) — LuaJIT-FFI shim: AUKit's Lua API surface for the decode → resample → effects hot path, backed
--- by libaukit_hip.so (include/aukit_hip.h).  Host code stays Lua: `local aukit = require "aukit"` keeps working, the
--- per-sample loops run on the MI355X.
---
--- NOT EXECUTED IN THIS REPOSITORY'S CI: the build image has no Lua / LuaJIT (see DESIGN.md §1); the Python mirror
--- aukit_amd/aukit.py implements the same mapping and is what the GPU tests drive.  This file is the binding a
--- maintainer would add next to the reference (INTEGRATION.md).  Strings in, tables out — same shapes as the reference:
---   Audio.data[c][i] doubles, stream iterators returning `chunk, pos` with chunk[c][i] in [-128, 127].
local ffi = require "ffi"

ffi.cdef [[
typedef struct aukit_ctx aukit_ctx; typedef struct aukit_batch aukit_batch; typedef struct aukit_audio aukit_audio; typedef struct aukit_chunks aukit_chunks;
typedef struct { int32_t codec, channels; double sample_rate; int32_t bit_depth, data_type, big_endian, interleaved, ulaw, top_first, block_align, ncoef;
                 int16_t coef1[32], coef2[32]; int32_t predictor[8], step_index[8]; } aukit_codec_desc;
const char *aukit_last_error(void);
int aukit_ctx_create(aukit_ctx **out, int device); void aukit_ctx_destroy(aukit_ctx *ctx);
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
int aukit_tone(aukit_ctx *, uint32_t n, double frequency, double duration, double amplitude, int wave, double duty, int channels, double sample_rate, int dtype, aukit_audio **out);
int aukit_pack_pcm(aukit_ctx *, const aukit_audio *, int bit_depth, int data_type, int big_endian, int interleaved, int int_mode, aukit_batch **out);
int aukit_stream_decode(aukit_ctx *, const aukit_batch *, const aukit_codec_desc *, int interp, int mono, int dtype, aukit_audio **out, aukit_chunks **chunks);
int aukit_chunks_info(const aukit_chunks *, uint32_t *n, uint32_t *max_chunks);
int aukit_chunks_get(const aukit_chunks *, uint32_t *nchunks, uint32_t *lens, double *pos, int32_t *status, double *length_seconds); void aukit_chunks_free(aukit_chunks *);
]]

local C = ffi.load(os.getenv("AUKIT_HIP_LIB") or "aukit_hip")
local F64, I8 = 0, 2
local INTERP = {none = 0, linear = 1, cubic = 2, sinc = 3}
local DTYPE = {signed = 0, unsigned = 1, float = 2}
local CODEC = {pcm = 0, g711 = 1, adpcm = 2, adpcm_wav = 3, msadpcm = 4, dfpwm = 5, mdfpwm = 6, qoa = 7, flac = 8}
local FX = {amplify = 0, speed = 1, fade = 2, invert = 3, normalize = 4, center = 5, trim = 6, delay = 7, echo = 8, reverb = 9, lowpass = 10, highpass = 11}

local aukit = {_VERSION = "1.10.0", defaultInterpolation = "linear", effects = {}, stream = {}}

local ctxp = ffi.new("aukit_ctx*[1]")
local function ctx()
    if ctxp[0] == nil and C.aukit_ctx_create(ctxp, 0) ~= 0 then error(ffi.string(C.aukit_last_error()), 3) end
    return ctxp[0]
end
-- every native status becomes the Lua error the reference raises (AUKIT_E_LUA carries the reference's own message)
local function check(rc) if rc ~= 0 then error(ffi.string(C.aukit_last_error()), 3) end end

local function upload(str)
    local off = ffi.new("uint64_t[2]", 0, #str)
    local b = ffi.new("aukit_batch*[1]")
    check(C.aukit_batch_upload(ctx(), b, ffi.cast("const uint8_t*", str), off, 1))
    return ffi.gc(b[0], C.aukit_batch_free)
end

local function desc(t)
    local d = ffi.new("aukit_codec_desc")
    d.codec = CODEC[t.codec]; d.channels = t.channels or 1; d.sample_rate = t.sampleRate or 48000
    d.bit_depth = t.bitDepth or 8; d.data_type = DTYPE[t.dataType or "signed"]; d.big_endian = t.bigEndian and 1 or 0
    d.interleaved = t.interleaved == false and 0 or 1; d.ulaw = t.ulaw and 1 or 0; d.top_first = t.topFirst == false and 0 or 1
    d.block_align = t.blockAlign or 0
    return d
end

--- Audio objects wrap a device handle; `data` is materialised on first access like any other Lua table.
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
function Audio_mt.__index(self, k)
    if k == "sampleRate" then local _, r = geom(self) return r end
    if k == "data" then
        local ch, _, len = geom(self)
        local buf = ffi.new("double[?]", math.max(ch * len, 1))
        check(C.aukit_audio_download(ctx(), self._h, buf))
        local data = {}
        for c = 1, ch do local t = {} for i = 1, len do t[i] = buf[(c - 1) * len + i - 1] end data[c] = t end
        rawset(self, "data", data)  -- NB: Lua code that edits audio.data by hand must call aukit.upload(audio) afterwards
        return data
    end
    return Audio[k]
end
function Audio:len() local _, r, len = geom(self) return len / r end
function Audio:channels() return (geom(self)) end
function Audio:resample(sampleRate, interpolation)
    interpolation = interpolation or aukit.defaultInterpolation
    if not INTERP[interpolation] then error("bad argument #2 (invalid interpolation type)", 2) end
    local o = ffi.new("aukit_audio*[1]")
    check(C.aukit_resample(ctx(), self._h, sampleRate, INTERP[interpolation], o))
    return wrap(o[0], self.metadata, self.info)
end
function Audio:mono()
    local o = ffi.new("aukit_audio*[1]")
    check(C.aukit_mono(ctx(), self._h, o))
    return wrap(o[0], self.metadata, self.info)
end
function Audio:mix(amplifier, ...)
    local audios = {self, ...}
    if type(amplifier) ~= "number" then table.insert(audios, 2, amplifier) amplifier = 1 end
    local arr = ffi.new("const aukit_audio*[?]", #audios)
    for i, a in ipairs(audios) do
        if a.sampleRate ~= self.sampleRate then a = a:resample(self.sampleRate) audios[i] = a end
        arr[i - 1] = a._h
    end
    local o = ffi.new("aukit_audio*[1]")
    check(C.aukit_mix(ctx(), arr, #audios, amplifier, o))
    return wrap(o[0], self.metadata, self.info)
end
function Audio:dfpwm(interleaved)
    local b = ffi.new("aukit_batch*[1]")
    check(C.aukit_dfpwm_encode(ctx(), self._h, interleaved == false and 0 or 1, b))
    local n, tot = ffi.new("uint32_t[1]"), ffi.new("uint64_t[1]")
    check(C.aukit_batch_info(b[0], n, tot))
    local buf = ffi.new("uint8_t[?]", math.max(tonumber(tot[0]), 1))
    check(C.aukit_batch_download(ctx(), b[0], buf))
    C.aukit_batch_free(b[0])
    return ffi.string(buf, tot[0])
end

-- structural methods (aukit.lua:690-866): device-side row copies, the source objects are never touched
local function group(self, ...)
    local audios = {self, ...}
    local arr = ffi.new("const aukit_audio*[?]", #audios)
    for i, a in ipairs(audios) do
        if a.sampleRate ~= self.sampleRate then a = a:resample(self.sampleRate) audios[i] = a end  -- :702, :756
        arr[i - 1] = a._h
    end
    return arr, #audios, audios  -- `audios` keeps the resampled temporaries alive across the call
end
function Audio:concat(...)
    local arr, n, keep = group(self, ...)
    local o = ffi.new("aukit_audio*[1]")
    check(C.aukit_concat(ctx(), arr, n, o))
    return wrap(o[0], self.metadata, self.info), keep and nil
end
function Audio:combine(...)
    local arr, n, keep = group(self, ...)
    local o = ffi.new("aukit_audio*[1]")
    check(C.aukit_combine(ctx(), arr, n, o))
    return wrap(o[0], self.metadata, self.info), keep and nil
end
function Audio:sub(start, last)
    local o = ffi.new("aukit_audio*[1]")
    check(C.aukit_sub(ctx(), self._h, start or 0, last or 0, o))
    return wrap(o[0], self.metadata, self.info)
end
function Audio:split(...)
    local res = {}
    for n, cl in ipairs {...} do
        if #cl == 0 then error("bad argument #" .. n .. " (cannot use empty table)") end
        local ch = ffi.new("int32_t[?]", #cl, cl)
        local o = ffi.new("aukit_audio*[1]")
        check(C.aukit_split(ctx(), self._h, ch, #cl, o))
        res[#res + 1] = wrap(o[0], self.metadata, self.info)
    end
    return table.unpack(res)
end
function Audio:rep(count)
    local o = ffi.new("aukit_audio*[1]")
    check(C.aukit_rep(ctx(), self._h, count, o))
    return wrap(o[0], self.metadata, self.info)
end
function Audio:reverse()
    local o = ffi.new("aukit_audio*[1]")
    check(C.aukit_reverse(ctx(), self._h, o))
    return wrap(o[0], self.metadata, self.info)
end
local WAVE = {sine = 1, triangle = 2, sawtooth = 3, square = 4}
function aukit.new(duration, channels, sampleRate)
    local o = ffi.new("aukit_audio*[1]")
    check(C.aukit_tone(ctx(), 1, 0, duration, 1, 0, 0.5, channels or 1, sampleRate or 48000, F64, o))
    return wrap(o[0], {}, {})
end
function aukit.tone(frequency, duration, amplitude, waveType, duty, channels, sampleRate)
    local w = WAVE[waveType or "sine"]
    if not w then error("bad argument #4 (invalid wave type)", 2) end
    local o = ffi.new("aukit_audio*[1]")
    check(C.aukit_tone(ctx(), 1, frequency, duration, amplitude or 1, w, duty or 0.5, channels or 1, sampleRate or 48000, F64, o))
    return wrap(o[0], {}, {})
end
-- the sample bytes of Audio:wav (aukit.lua:966-971); the RIFF header is string.pack'ed around them exactly as :993-996 does.
-- int_mode 0 = truncate like the CC: Tweaked VM's string.pack (a Java long cast); see include/aukit_hip.h
local function wav_body(self, bitDepth)
    local b = ffi.new("aukit_batch*[1]")
    check(C.aukit_pack_pcm(ctx(), self._h, bitDepth, bitDepth == 8 and 1 or 0, 0, 1, 0, b))
    local n, tot = ffi.new("uint32_t[1]"), ffi.new("uint64_t[1]")
    check(C.aukit_batch_info(b[0], n, tot))
    local buf = ffi.new("uint8_t[?]", math.max(tonumber(tot[0]), 1))
    check(C.aukit_batch_download(ctx(), b[0], buf))
    C.aukit_batch_free(b[0])
    return ffi.string(buf, tot[0])
end
aukit._wav_body = wav_body

local function loader(d, data, info)
    local o = ffi.new("aukit_audio*[1]")
    check(C.aukit_decode(ctx(), upload(data), d, F64, o))
    return wrap(o[0], {}, info)
end
function aukit.pcm(data, bitDepth, dataType, channels, sampleRate, interleaved, bigEndian)
    return loader(desc {codec = "pcm", bitDepth = bitDepth, dataType = dataType, channels = channels, sampleRate = sampleRate, interleaved = interleaved, bigEndian = bigEndian},
        data, {bitDepth = bitDepth or 8, dataType = dataType or "signed"})
end
function aukit.g711(data, ulaw, channels, sampleRate) return loader(desc {codec = "g711", ulaw = ulaw, channels = channels, sampleRate = sampleRate or 8000}, data) end
function aukit.dfpwm(data, channels, sampleRate) return loader(desc {codec = "dfpwm", channels = channels, sampleRate = sampleRate}, data) end
function aukit.msadpcm(data, blockAlign, channels, sampleRate) return loader(desc {codec = "msadpcm", blockAlign = blockAlign, channels = channels, sampleRate = sampleRate}, data) end
function aukit.mdfpwm(data) return loader(desc {codec = "mdfpwm"}, data) end
function aukit.qoa(data) return loader(desc {codec = "qoa"}, data) end
function aukit.flac(data) return loader(desc {codec = "flac"}, data) end
-- aukit.wav / aiff / au: parse the container header in Lua exactly as the reference does (aukit.lua:1456-1651),
-- then call the loader above on the payload (IMA blocks: codec = "adpcm_wav" with blockAlign).

for name, id in pairs(FX) do
    aukit.effects[name] = function(audio, ...)
        local args = {...}
        if name == "speed" then args[2] = INTERP[aukit.defaultInterpolation] end
        if name == "normalize" then args[1] = args[1] or 1 args[2] = args[2] and 1 or 0 end
        local a = ffi.new("double[?]", math.max(#args, 1))
        for i, v in ipairs(args) do a[i - 1] = v end
        check(C.aukit_effect(ctx(), audio._h, id, a, #args))
        rawset(audio, "data", nil)  -- drop the cached Lua copy: the device buffer changed in place
        return audio
    end
end

--- stream factories: every iterator call is computed up front in one launch; the iterator just hands out slices.
local function streamer(d, data, mono, dtype)
    local o, ck = ffi.new("aukit_audio*[1]"), ffi.new("aukit_chunks*[1]")
    check(C.aukit_stream_decode(ctx(), upload(data), d, INTERP[aukit.defaultInterpolation], mono and 1 or 0, dtype, o, ck))
    local audio = wrap(o[0])
    local n, mx = ffi.new("uint32_t[1]"), ffi.new("uint32_t[1]")
    check(C.aukit_chunks_info(ck[0], n, mx))
    local m = math.max(mx[0], 1)
    local nch, lens, pos, status, length = ffi.new("uint32_t[1]"), ffi.new("uint32_t[?]", m), ffi.new("double[?]", m), ffi.new("int32_t[1]"), ffi.new("double[1]")
    check(C.aukit_chunks_get(ck[0], nch, lens, pos, status, length))
    C.aukit_chunks_free(ck[0])
    local k, off, chans = 0, 0, audio.data
    return function()
        if k >= nch[0] then
            if status[0] == -2 then error("attempt to compare nil with number", 2) end  -- the reference raises here too
            return nil
        end
        local chunk = {}
        for c = 1, #chans do local t = {} for i = 1, lens[k] do t[i] = chans[c][off + i] end chunk[c] = t end
        off = off + lens[k]; k = k + 1
        return chunk, pos[k - 1]
    end, length[0]
end
function aukit.stream.pcm(data, bitDepth, dataType, channels, sampleRate, bigEndian, mono)
    return streamer(desc {codec = "pcm", bitDepth = bitDepth, dataType = dataType, channels = channels, sampleRate = sampleRate, bigEndian = bigEndian}, data, mono, F64)
end
function aukit.stream.g711(input, ulaw, channels, sampleRate, mono) return streamer(desc {codec = "g711", ulaw = ulaw, channels = channels, sampleRate = sampleRate or 8000}, input, mono, I8) end
function aukit.stream.adpcm(input, blockAlign, channels, sampleRate, mono) return streamer(desc {codec = "adpcm_wav", blockAlign = blockAlign, channels = channels, sampleRate = sampleRate}, input, mono, I8) end
function aukit.stream.msadpcm(input, blockAlign, channels, sampleRate, mono) return streamer(desc {codec = "msadpcm", blockAlign = blockAlign, channels = channels, sampleRate = sampleRate}, input, mono, I8) end
function aukit.stream.dfpwm(data, sampleRate, channels, mono) return streamer(desc {codec = "dfpwm", channels = channels, sampleRate = sampleRate}, data, mono, F64) end
function aukit.stream.mdfpwm(data, mono) return streamer(desc {codec = "mdfpwm"}, data, mono, I8) end
function aukit.stream.flac(data, mono) return streamer(desc {codec = "flac"}, data, mono, F64) end
function aukit.stream.qoa(data, mono) return streamer(desc {codec = "qoa"}, data, mono, F64) end

return aukit

"""ctypes loader for libaukit_hip.so (the C ABI of include/aukit_hip.h).

There is deliberately NO CPU fallback: if the HIP library is missing, or no MI355X is
visible when a context is created, this module raises.  The CPU oracle under oracle/ is
test infrastructure and is never imported from here.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_HERE, "libaukit_hip.so")
if os.environ.get("AUKIT_LIB"):  # A/B of library builds on one box (tools/build_variant.sh): never set in tests or by the driver
    LIB_PATH = os.path.abspath(os.environ["AUKIT_LIB"])
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["runtime.hip", "resample.hip", "fast.hip", "fast2.hip", "fast_stream.hip", "fast_stream_f32.hip", "fast_stream_u8.hip", "fast_stream_dfpwm.hip", "fast_stream_s16x2.hip", "fast_coef.hip", "fast_s16x2.hip", "fast_fmt.hip", "floor_wave.hip", "exact_wave.hip", "wave_f64.hip", "wave_coef_f64.hip", "container.hip", "stream_handle.hip", "api_resample.hip", "codecs.hip", "codecs2.hip", "msadpcm.hip", "qoa_stream.hip", "qoa.hip", "stream_tail.hip", "effects.hip", "flac.hip", "flac_fused.hip", "flac_stream.hip", "flac_pq.hip", "flac_tail.hip", "rs_periodic.hip", "ops.hip", "dfpwm_par.hip", "dfpwm_spec.hip", "group.hip"]
HEADERS = ["common.h", "resample.h", "fast_wave_dev.h", "fast_stream_body.h", "resample_dev.h", "dfpwm_dev.h", "dfpwm_par_dev.h", "stream_tail.h", "flac_dev.h", "flac_stream_dev.h", "rs_onepole_dev.h", os.path.join(_ROOT, "include", "aukit_hip.h")]

OK, E_ARG, E_LUA, E_NOMEM, E_UNSUPPORTED, E_HIP = 0, -1, -2, -3, -4, -5
F64, F32, I8 = 0, 1, 2
INTERP = {"none": 0, "linear": 1, "cubic": 2, "sinc": 3}
PCM_TYPE = {"signed": 0, "unsigned": 1, "float": 2}
CODEC_PCM, CODEC_G711, CODEC_ADPCM, CODEC_ADPCM_WAV, CODEC_MSADPCM, CODEC_DFPWM, CODEC_MDFPWM, CODEC_QOA, CODEC_FLAC = range(9)
FX = {"amplify": 0, "speed": 1, "fade": 2, "invert": 3, "normalize": 4, "center": 5, "trim": 6, "delay": 7, "echo": 8, "reverb": 9,
      "lowpass": 10, "highpass": 11}
MAX_CH = 8            # AUKIT_MAX_CHANNELS
MAX_PLANAR_CH = 64    # AUKIT_MAX_PLANAR_CHANNELS (the descriptor's predictor / step_index arrays)
OPT_EXACT_MATH, OPT_STORE_X4, OPT_COLLECT_STATS, OPT_DFPWM_SPECULATE = 0, 1, 2, 3
COUNTER_DFPWM_CHUNKS, COUNTER_DFPWM_CHUNKS_REDONE, COUNTER_FLAC_FUSED, COUNTER_TIER1_ERR_NANO, COUNTER_TIER1_OUTPUTS = 0, 1, 2, 3, 4
COUNTER_DFPWM_RESPECULATED, COUNTER_DFPWM_HARD, COUNTER_RECURRENCE_F32 = 5, 6, 7
WAVE_NONE, WAVE_SINE, WAVE_TRIANGLE, WAVE_SAWTOOTH, WAVE_SQUARE = 0, 1, 2, 3, 4
PACK_TRUNC, PACK_FLOOR, PACK_STRICT = 0, 1, 2
STREAM_CHUNK, STREAM_NEED_INPUT, STREAM_END = 0, 1, 2

# every symbol include/aukit_hip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "aukit_abi_version", "aukit_last_error", "aukit_ctx_create", "aukit_ctx_destroy", "aukit_ctx_set_stream", "aukit_ctx_get_stream",
    "aukit_ctx_sync", "aukit_ctx_set_dtype", "aukit_ctx_set_option", "aukit_ctx_get_counter", "aukit_ctx_set_sinc_window", "aukit_timer_begin", "aukit_timer_end", "aukit_timer_stats",
    "aukit_ctx_set_kernel_timing", "aukit_ctx_last_kernel",
    "aukit_batch_upload", "aukit_batch_wrap_device", "aukit_batch_info", "aukit_batch_offsets", "aukit_batch_device_ptr",
    "aukit_batch_download", "aukit_batch_free",
    "aukit_audio_upload", "aukit_audio_info", "aukit_audio_layout", "aukit_audio_device_ptr", "aukit_audio_download",
    "aukit_audio_download_raw", "aukit_audio_clone", "aukit_audio_free",
    "aukit_parse_container", "aukit_decode", "aukit_decode_table", "aukit_decode_nibbles", "aukit_stream_decode_table", "aukit_decode_resample", "aukit_resample", "aukit_mono", "aukit_mix", "aukit_effect", "aukit_dfpwm_encode", "aukit_dfpwm_transcode_mono",
    "aukit_encode_pcm", "aukit_stream_decode", "aukit_chunks_info", "aukit_chunks_get", "aukit_chunks_free",
    "aukit_stream_open", "aukit_stream_feed", "aukit_stream_finish", "aukit_stream_next", "aukit_stream_length", "aukit_stream_resident", "aukit_stream_close",
    "aukit_partition", "aukit_group_create", "aukit_group_destroy", "aukit_group_info", "aukit_group_ctx", "aukit_group_sync", "aukit_group_scatter",
    "aukit_group_gather_audio", "aukit_group_gather_batch", "aukit_group_run", "aukit_group_last_run",
    "aukit_concat", "aukit_sub", "aukit_combine", "aukit_split", "aukit_rep", "aukit_reverse", "aukit_tone", "aukit_noise", "aukit_pack_pcm",
]


class CodecDesc(C.Structure):
    _fields_ = [("codec", C.c_int32), ("channels", C.c_int32), ("sample_rate", C.c_double), ("bit_depth", C.c_int32),
                ("data_type", C.c_int32), ("big_endian", C.c_int32), ("interleaved", C.c_int32), ("ulaw", C.c_int32),
                ("top_first", C.c_int32), ("block_align", C.c_int32), ("ncoef", C.c_int32), ("coef1", C.c_int16 * 32),
                ("coef2", C.c_int16 * 32), ("predictor", C.c_int32 * MAX_PLANAR_CH), ("step_index", C.c_int32 * MAX_PLANAR_CH)]


GOP = {"none": 0, "decode": 1, "decode_resample": 2, "stream_decode": 3, "resample": 4, "mono": 5, "effect": 6, "dfpwm_encode": 7, "dfpwm_transcode_mono": 8,
       "encode_pcm": 9, "sync": 10}


class GroupCall(C.Structure):
    """aukit_group_call (include/aukit_hip.h): one entry-point call of a member's list in aukit_group_run"""
    _fields_ = [("op", C.c_int32), ("dtype", C.c_int32), ("interp", C.c_int32), ("mono", C.c_int32), ("batch", C.c_void_p), ("desc", C.c_void_p),
                ("audio", C.c_void_p), ("out_audio", C.c_void_p), ("out_batch", C.c_void_p), ("out_chunks", C.c_void_p), ("new_rate", C.c_double),
                ("effect_id", C.c_int32), ("nargs", C.c_int32), ("args", C.c_double * 8), ("channels", C.c_int32), ("interleaved", C.c_int32),
                ("bit_depth", C.c_int32), ("data_type", C.c_int32)]


class Container(C.Structure):
    _fields_ = [("desc", CodecDesc), ("payload_off", C.c_uint64), ("payload_len", C.c_uint64), ("wav_data_type", C.c_int32), ("bit_depth", C.c_int32),
                ("length_seconds", C.c_double)]


CONTAINER_WAV, CONTAINER_AIFF, CONTAINER_AU = 0, 1, 2
WAVDT = ("signed", "unsigned", "float", "alaw", "ulaw", "adpcm", "msadpcm", "dfpwm")


class AukitError(RuntimeError):
    """Raised for every non-zero aukit_status; `.code` is the status, `.msg` the reference's error text."""

    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code
        self.msg = msg


def build(force=False, verbose=False, jobs=None):
    """hipcc --offload-arch=gfx950 every file of csrc/ into an object under aukit_amd/build/ (only the ones whose source or any
    header changed; up to `jobs` compilers at once), then link aukit_amd/libaukit_hip.so (cross-compiles without a GPU)."""
    import concurrent.futures
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]
    bdir = os.path.join(_HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    hdr_m = max(os.path.getmtime(h) for h in hdrs)
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(d) <= os.path.getmtime(LIB_PATH) for d in srcs + hdrs):
        return LIB_PATH  # up to date (the objects under build/ do not travel to the GPU box; the library does)
    todo, objs = [], []
    for s in srcs:
        o = os.path.join(bdir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_m):
            todo.append((s, o))

    def cc(so):
        cmd = [hipcc] + flags + ["-c", so[0], "-o", so[1]]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if todo:
        with concurrent.futures.ThreadPoolExecutor(jobs or min(8, os.cpu_count() or 1)) as ex:
            list(ex.map(cc, todo))
    if todo or not os.path.exists(LIB_PATH) or any(os.path.getmtime(o) > os.path.getmtime(LIB_PATH) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(aukit_amd has no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    L.aukit_last_error.restype = C.c_char_p
    L.aukit_ctx_get_stream.restype = C.c_void_p
    L.aukit_batch_device_ptr.restype = C.c_void_p
    L.aukit_audio_device_ptr.restype = C.c_void_p
    L.aukit_group_ctx.restype = C.c_void_p
    L.aukit_group_ctx.argtypes = [C.c_void_p, C.c_uint32]
    L.aukit_ctx_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.aukit_batch_wrap_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    for name in ("aukit_ctx_destroy", "aukit_batch_free", "aukit_audio_free", "aukit_chunks_free", "aukit_stream_close", "aukit_group_destroy"):
        getattr(L, name).restype = None
        getattr(L, name).argtypes = [C.c_void_p]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise AukitError(rc, lib().aukit_last_error().decode(errors="replace"))

// stubs.hip — placeholders for entry points that are implemented in later files (removed as they land).
#include "common.h"
namespace aukit {
int decode_block_codec(aukit_ctx *, const aukit_batch *, const aukit_codec_desc *d, double, int, bool, int, aukit_audio **) {
    return fail(AUKIT_E_UNSUPPORTED, "codec %d is not implemented yet", d->codec);
}
int stream_block_codec(aukit_ctx *, const aukit_batch *, const aukit_codec_desc *d, int, int, int, aukit_audio **, aukit_chunks **) {
    return fail(AUKIT_E_UNSUPPORTED, "stream codec %d is not implemented yet", d->codec);
}
}
using namespace aukit;
extern "C" {
int aukit_mono(aukit_ctx *, const aukit_audio *, aukit_audio **) { return fail(AUKIT_E_UNSUPPORTED, "not implemented yet"); }
int aukit_mix(aukit_ctx *, const aukit_audio *const *, int, double, aukit_audio **) { return fail(AUKIT_E_UNSUPPORTED, "not implemented yet"); }
int aukit_effect(aukit_ctx *, aukit_audio *, int, const double *, int) { return fail(AUKIT_E_UNSUPPORTED, "not implemented yet"); }
int aukit_dfpwm_encode(aukit_ctx *, const aukit_audio *, int, aukit_batch **) { return fail(AUKIT_E_UNSUPPORTED, "not implemented yet"); }
int aukit_encode_pcm(aukit_ctx *, const aukit_audio *, int, int, int, aukit_audio **) { return fail(AUKIT_E_UNSUPPORTED, "not implemented yet"); }
}

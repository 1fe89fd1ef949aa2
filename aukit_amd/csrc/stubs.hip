// stubs.hip — placeholders for entry points that are implemented in later files (removed as they land).
#include "common.h"
namespace aukit {
int decode_flac_audio(aukit_ctx *, const aukit_batch *, const aukit_codec_desc *, double, int, bool, int, aukit_audio **) { return fail(AUKIT_E_UNSUPPORTED, "FLAC is not implemented yet"); }
int stream_flac(aukit_ctx *, const aukit_batch *, const aukit_codec_desc *, int, int, int, aukit_audio **, aukit_chunks **) { return fail(AUKIT_E_UNSUPPORTED, "stream.flac is not implemented yet"); }
int stream_qoa(aukit_ctx *, const aukit_batch *, const aukit_codec_desc *, int, int, int, aukit_audio **, aukit_chunks **) { return fail(AUKIT_E_UNSUPPORTED, "stream.qoa is not implemented yet"); }
}

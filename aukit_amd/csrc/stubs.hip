// stubs.hip — placeholders for entry points that are implemented in later files (removed as they land).
#include "common.h"
namespace aukit {
int stream_qoa(aukit_ctx *, const aukit_batch *, const aukit_codec_desc *, int, int, int, aukit_audio **, aukit_chunks **) { return fail(AUKIT_E_UNSUPPORTED, "stream.qoa is not implemented yet"); }
}

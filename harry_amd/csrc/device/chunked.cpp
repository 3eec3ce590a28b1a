// Chunked profile (.hry v0.2) and decode entry points -- placeholders until the chunked kernels land.
#include "context.hpp"

namespace hry {

void encode_chunked(Context &, Mesh &, int, std::vector<uint8_t> &) { throw Error(HRY_E_UNSUPPORTED, "chunked profile: not built yet"); }
Mesh *decode_any(Context &, const uint8_t *, size_t) { throw Error(HRY_E_UNSUPPORTED, "decode: not built yet"); }

}   // namespace hry

// bf16 MFMA grouped GEMMs for the expert FFN -- placeholder until the tiled kernels land.
#include "ffn_mfma.h"

namespace eve {

bool ffn_mfma_supported(int, int) { return false; }

int ffn_fwd_mfma(const void*, const void*, const void*, const void*, const void*, void*, void*, void*, const int32_t*,
                 int, int, int, int, int, int, hipStream_t) {
  set_error("ffn_fwd_mfma: not built");
  return EVE_ERR_UNSUPPORTED;
}

int ffn_bwd_mfma(const void*, const void*, const void*, const void*, const void*, const void*, void*, void*, void*,
                 void*, void*, void*, const int32_t*, int, int, int, int, int, int, hipStream_t) {
  set_error("ffn_bwd_mfma: not built");
  return EVE_ERR_UNSUPPORTED;
}

}  // namespace eve

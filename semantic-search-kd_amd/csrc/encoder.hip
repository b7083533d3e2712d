// e5-small-v2 shaped BERT encoder forward (bf16 MFMA) — C-ABI entry points.
// Kernels land here next; until then the entry points refuse loudly.
#include "common.h"

extern "C" {

size_t sskd_encoder_workspace_bytes(const sskd_encoder_config* cfg, int B, int S) {
  (void)cfg; (void)B; (void)S;
  return 0;
}

int sskd_encoder_forward(const sskd_encoder_config*, const sskd_encoder_weights*, const int32_t*,
                         const int32_t*, int, int, int, float*, void*, size_t, void*) {
  return sskd::fail(SSKD_ERR_UNSUPPORTED, "encoder_forward: kernels not built in this revision");
}

int sskd_encoder_hidden(const sskd_encoder_config*, const sskd_encoder_weights*, const int32_t*,
                        const int32_t*, int, int, void*, void*, size_t, void*) {
  return sskd::fail(SSKD_ERR_UNSUPPORTED, "encoder_hidden: kernels not built in this revision");
}

}  // extern "C"

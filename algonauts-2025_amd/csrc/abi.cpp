// Error plumbing and version of the C ABI (include/tribe_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/tribe_hip.h"

static thread_local char g_err[512] = "";

void tribe_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int tribe_version(void) { return TRIBE_ABI_VERSION; }
extern "C" int tribe_abi_struct_sizes(int64_t* sizes, int32_t n) {
  const int64_t all[] = {sizeof(tribe_gemm_desc), sizeof(tribe_attention_desc), sizeof(tribe_encoder_layer), sizeof(tribe_encoder_desc),
                         sizeof(tribe_vit_layer), sizeof(tribe_vit_fp8_layer), sizeof(tribe_vjepa2_desc), sizeof(tribe_conformer_layer),
                         sizeof(tribe_conformer_fp8_layer), sizeof(tribe_w2vbert_desc), sizeof(tribe_llama_layer), sizeof(tribe_llama_fp8_layer),
                         sizeof(tribe_llama_desc), sizeof(tribe_feature_piece), sizeof(tribe_adam_tensor)};
  const int total = (int)(sizeof(all) / sizeof(all[0]));
  for (int i = 0; sizes && i < n && i < total; ++i) sizes[i] = all[i];
  return total;
}
extern "C" const char* tribe_last_error(void) { return g_err; }

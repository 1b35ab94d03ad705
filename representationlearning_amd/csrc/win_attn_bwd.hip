// placeholder until the backward kernel lands (same translation unit name as the final kernel)
#include "win_attn.cuh"
extern "C" int rssf_winattn_bwd(const rssf_winattn_bwd_params* p, void* stream) {
  (void)p; (void)stream;
  rssf::set_error("winattn_bwd: not built yet");
  return RSSF_ERR_UNSUPPORTED;
}

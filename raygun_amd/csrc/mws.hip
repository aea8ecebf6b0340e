// Mutex watershed — placeholder until the union-find kernels land.
#include "common.h"
extern "C" int64_t rg_mws_workspace_bytes(const rg_mws_params* p) { (void)p; return 0; }
extern "C" int rg_mws_segment(const float* affs, const int32_t* offsets, const uint8_t* mask, uint64_t* labels,
                              const rg_mws_params* p, void* workspace, size_t ws_bytes, rg_stream s) {
  (void)affs; (void)offsets; (void)mask; (void)labels; (void)p; (void)workspace; (void)ws_bytes; (void)s;
  RG_FAIL(RG_ERR_UNSUPPORTED, "rg_mws_segment: not implemented yet");
}

/* oracle/ref_shim.h -- TEST INFRASTRUCTURE (oracle/build_ref_shim.sh).  Force-included in front of the reference's
 * translation units whose hot-path member bodies are replaced by libhmx calls: the context INTEGRATION.md section 2
 * creates next to the reference's globals, made on first use from those globals (the decoder sets them when it
 * activates the sequence parameter set, before the first block is reconstructed). */
#ifndef HMX_REF_SHIM_H
#define HMX_REF_SHIM_H
#include <cstdio>
#include <cstdlib>

#include "hmx.h"
typedef unsigned int UInt_shim;
extern UInt_shim g_uiBitDepth, g_uiBitIncrement, g_uiMaxCUWidth; /* TLibCommon/TComRom.h:90,169-170 */
static inline hmx_ctx *hmx_shim_ctx() {
  static hmx_ctx *ctx = nullptr;
  if (!ctx) {
    hmx_config cfg = {(int)(g_uiBitDepth + g_uiBitIncrement), 0, nullptr, (int)g_uiMaxCUWidth};
    if (hmx_create(&cfg, &ctx) != HMX_OK) {
      fprintf(stderr, "libhmx shim: hmx_create failed (no MI355X?)\n");
      exit(EXIT_FAILURE);
    }
    fprintf(stderr, "libhmx shim: context for %d-bit samples, CTU %d\n", cfg.bit_depth, cfg.ctu_size);
  }
  return ctx;
}
namespace { /* internal linkage: every translation unit has ITS OWN counter and ITS OWN report (an inline destructor with
               external linkage would be merged across the four units and print one unit's counter four times) */
unsigned long g_hmx_shim_calls = 0;
struct HmxShimReport {
  const char *unit;
  ~HmxShimReport() { fprintf(stderr, "libhmx shim: %lu calls from %s\n", g_hmx_shim_calls, unit); }
};
} // namespace
#define HMX_SHIM_CHECK(call)                                                                   \
  do {                                                                                         \
    ++g_hmx_shim_calls;                                                                        \
    if ((call) != HMX_OK) {                                                                    \
      fprintf(stderr, "libhmx shim: %s failed: %s\n", #call, hmx_last_error(hmx_shim_ctx()));  \
      exit(EXIT_FAILURE);                                                                      \
    }                                                                                          \
  } while (0)
#endif

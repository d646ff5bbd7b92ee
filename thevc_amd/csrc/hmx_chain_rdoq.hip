// hmx_chain_rdoq.hip -- part of libhmx (include/hmx.h), gfx950.  See hmx_host.h for how the library is cut into translation units.
// The packed schedule's persistent kernel with xRateDistOptQuant as the chain's quantiser (hmx_set_rdoq): the two
// instantiations (with and without the distortion output) compile as long as the rest of the library, so they are a
// translation unit of their own.
#include "hmx_chain_dev.h"

int packed_rdoq_max_blocks(int *nb) {
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(nb, k_intra_packed<true, 64, true, true>, 64, 0) == hipSuccess ? 0 : -1;
}
void launch_packed_rdoq(const PackArgs &A, bool sse, unsigned n_wg, hipStream_t st) {
  const dim3 grid(n_wg), blk(64);
  if (sse) hipLaunchKernelGGL((k_intra_packed<true, 64, true, true>), grid, blk, 0, st, A);
  else hipLaunchKernelGGL((k_intra_packed<true, 64, false, true>), grid, blk, 0, st, A);
}

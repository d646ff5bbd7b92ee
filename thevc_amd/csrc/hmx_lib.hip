// hmx_lib.hip -- libhmx as ONE translation unit: every part of the library, in the order of hmx_host.h's list.
// The default build (__graft_entry__.build()) compiles the parts separately and in parallel; this file is for builds
// that need all device symbols in one module (-DHMX_PACK_PROFILE: the phase profile of the packed schedule reads the
// RDOQ counters next to the schedule's own) and as a cross-check that the parts do not depend on their build order.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -DHMX_PACK_PROFILE -o thevc_amd/libhmx_prof.so thevc_amd/csrc/hmx_lib.hip
#define HMX_RDOQ_KERNELS 1
#include "hmx_core.hip"
#include "hmx_list.hip"
#include "hmx_scalar.hip"
#include "hmx_plan.hip"
#include "hmx_chain.hip"
#include "hmx_chain_rdoq.hip"
#include "hmx_inter.hip"
#include "hmx_loop.hip"

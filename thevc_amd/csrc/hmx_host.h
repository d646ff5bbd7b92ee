// hmx_host.h -- what the translation units of libhmx share: the structures that travel between host and device, the
// context, and the prototypes of the host helpers that cross files.  The library is cut into
//   hmx_core.hip        context, device-memory plumbing, quantiser parameters, argument arena
//   hmx_list.hip        block-list kernels (k_list, k_inter4, k_inter32) and the hmx_batch_* entry points over lists
//   hmx_scalar.hip      scalar drop-ins of TComTrQuant / TComPattern / TComPrediction / TComRdCost, RDOQ block list
//   hmx_plan.hip        intra plans: dependency analysis on the host and on the device
//   hmx_chain.hip       whole-picture all-intra chain: level / wave / packed schedules, resident pools
//   hmx_chain_rdoq.hip  the packed schedule's kernel with xRateDistOptQuant as its quantiser (a translation unit of its own:
//                       its instantiations take as long to compile as everything else together)
//   hmx_inter.hip       interpolation filters, motion compensation, sub-pel cost fan-out, border extension
//   hmx_loop.hip        deblocking, SAO, YUV file formats
// and hmx_lib.hip includes all of them for a single-translation-unit build (-DHMX_PACK_PROFILE builds read device
// symbols of several parts).  Build: __graft_entry__.build() compiles the parts in parallel and links libhmx.so.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "hmx_kernels.h"
#include "hmx_rdoq.h"

using namespace hmx;

struct DTu { // device descriptor of the list kernels: hmx_tu + index in the caller's order
  hmx_tu t;
  uint32_t idx;
};
__device__ __forceinline__ DTu load_dtu(const DTu *p) { // one 12-byte load instead of one per field read
  typedef __attribute__((address_space(1))) const int gint;
  int w[3];
  __builtin_memcpy(w, (gint *)p, 12);
  DTu d;
  __builtin_memcpy(&d, w, 12);
  return d;
}

// list kernels: 256-thread workgroups = four autonomous waves; blocks per workgroup
template <int N>
struct Slots {
  static constexpr int v = N == 64 ? 1 : N == 32 ? 4 : 256 / N; // 32x32 scratch (9.5 KB) is kept to four blocks; 64x64 (prediction only, 19.5 KB): one
};
#define HMX_SMEM_BYTES (16 * (int)sizeof(TuLds<16>)) /* largest of Slots<N> * sizeof(TuLds<N>) */
static_assert(64 * sizeof(TuLds<4>) <= HMX_SMEM_BYTES && 32 * sizeof(TuLds<8>) <= HMX_SMEM_BYTES &&
                  4 * sizeof(TuLds<32>) <= HMX_SMEM_BYTES && sizeof(TuLds<64>) <= HMX_SMEM_BYTES,
              "LDS scratch");

enum ListOp { OP_TRANSFORM_NXN, OP_INVTRANSFORM_NXN, OP_XT, OP_XIT, OP_XQUANT, OP_XDEQUANT, OP_PRED, OP_TRANSFORM_RECON };

struct ListPic { // planes of one picture of a multi-picture list call
  PlanesDev a, b;
  LevelsDev lev;
  PlanesDev rec; // OP_TRANSFORM_RECON: reconstruction out
};
struct ListArgs {
  const DTu *tus;
  int n;
  const ListPic *pics; // != NULL: grid.y pictures, planes from this table instead of a / b / lev
  int n_pics, abs_stride;
  PlanesDev a;   // residual in (transform) / prediction in (inverse with recon) / recon (pred)
  PlanesDev b;   // output planes
  LevelsDev lev; // levels / coefficients (Int)
  LevelsDev lev2;
  uint32_t *abs_sum;
  uint32_t *sse; // OP_TRANSFORM_RECON: xGetSSE(org, rec) per block, indexed like abs_sum (NULL = none)
  int have_pred;
  PlanesDev org;        // OP_PRED with cost: the original the predictions are costed against
  uint32_t *cost;       // OP_PRED: calcHAD of every (block, mode), [block idx][n_modes]; NULL = none
  const uint8_t *modes; // OP_PRED fan-out
  int n_modes;
  size_t mode_elems[3];
  PicDev P;
};

template <typename T>
__device__ __forceinline__ T pick3(const T (&a)[3], int i) { // a[i] without a run-time index (see k_list)
  T r = a[0];
  r = i == 1 ? a[1] : r;
  return i == 2 ? a[2] : r;
}
// The same for a table entry that is uniform over the wave: the three values are pinned as wave-uniform (scalar
// loads), or the compiler turns the select of loads back into one per-lane load from a selected address.
template <typename T>
__device__ __forceinline__ T uniform3v(T a0, T a1, T a2, int i) {
  const T v0 = wave_uniform(a0), v1 = wave_uniform(a1), v2 = wave_uniform(a2);
  T r = v0;
  r = i == 1 ? v1 : r;
  return i == 2 ? v2 : r;
}
#define uniform3(arr, i) uniform3v((arr)[0], (arr)[1], (arr)[2], (i))

// ---- whole-picture all-intra reconstruction: one launch per CTU diagonal ----
// Work item = (picture, CTU of the diagonal, plane), owned by ONE autonomous wave (64-thread
// workgroup): no workgroup barrier anywhere.  The host plan lists the item's blocks as segments of
// equal size and equal dependency level; a wave walks its segments, 64/N blocks at a time on the
// VALU path (N <= 16), one 32x32 block at a time on the matrix cores.
struct Seg { // a run of same-size blocks of one dependency level of one (CTU, plane)
  uint32_t start;
  uint16_t count;
  uint8_t log2n;
  uint8_t new_level; // 1: first segment of a dependency level (needs the previous level's recon)
};
struct FTu { // block descriptor of the frame path: geometry + precomputed neighbour availability
  hmx_tu t;
  uint32_t avail_lo, avail_hi;
};
struct LevelRow { // blocks of one picture-wide dependency level, bucketed by size (log2n - 2)
  uint32_t start[4];
  uint32_t count[4];
};
struct PicWork { // per picture: working planes + the plan it follows
  TiledPlane org[3], rec[3]; // tiled working copies (see TiledPlane)
  int *lev[3];
  int lev_stride[3];         // > 0: plane geometry; 0: the reference's Z-order coefficient layout
  const FTu *tus;
  const Seg *segs;
  const uint32_t *seg_range; // [(ctu*3+plane)*2 + {0,1}] -> begin,end in segs
  const FTu *ltus;           // level schedule: blocks sorted by (level, size)
  const LevelRow *ltab;      // [n_levels]
  int n_levels;
};
struct FrameArgs {
  const PicWork *pics;
  const uint32_t *wave_ctus; // CTU ids of this diagonal
  int n_wave_ctus;
  PicDev P;
};
// plane <-> tiled conversion: one thread per tile row (4 samples), a 256-thread workgroup = 64 tiles
// = a 32x32 region in Z-order: tiled side fully coalesced, plane side whole 64-byte sectors.
struct ConvJob { // one plane of one picture
  short *plane;
  int stride, w, h;
  TiledPlane T;
};
struct PackRow { // one (level, group)
  uint32_t wave_base;    // index of its first wave-item in the call's descriptor array
  uint32_t n_waves;
  uint32_t item_base[4]; // first entry of each size class in the call's item array
  uint32_t count[4];     // blocks per size class
  uint32_t pad[2];
};
struct PackDesc { // one wave-item
  uint32_t item_off;   // first item
  uint32_t n_s;        // items | size class << 28
  uint32_t row;
  uint32_t dep_target; // wave-items of the previous row of the group (0: nothing to wait for)
};
struct PackHdr {
  uint32_t shard_base[9]; // wave-items of shard s: [shard_base[s], shard_base[s+1])
  uint32_t total_items;
  uint32_t pad0[22];
  uint32_t abort;         // set by a wave whose wait timed out
  uint32_t pad1[31];
  uint32_t ticket[8][32]; // one 128-byte line per shard
  uint32_t owner[8][32];  // 0: unclaimed, x + 1: claimed by XCD x
  // -DHMX_PACK_PROFILE builds only: phases of a wave-item in ticks of the 100 MHz wall clock, summed over all wave-items
  // [0] draw a ticket [1] descriptor [2] item loads issued up to the wait [3] wait for the previous row [4] references +
  // arithmetic + stores issued [5] drain of the stores [6] count; [7] wave-items; [8] polls; [9] waves' lifetimes
  unsigned long long prof[16];
};
struct PackPic { // per picture: where its levels go, and the plan it follows
  int *lev[3];
  int lev_stride[3];
  int n_levels;
  const LevelRow *ltab;
  const FTu *ltus;
  uint32_t *sse[3]; // distortion output per plane (hmx_set_sse_output), NULL = none
};
struct PackGeom {
  int n_pics, I, n_groups, n_shards, max_levels, slots4, slots8;
};
__host__ __device__ __forceinline__ uint32_t pack_slots(int s, int slots4, int slots8) { return s == 0 ? (uint32_t)slots4 : s == 1 ? (uint32_t)slots8 : s == 2 ? 4u : 1u; }

struct hmx_ctx {
  hmx_config cfg;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  // scratch for the scalar drop-ins (one block): device staging
  char *d_scratch = nullptr;
  size_t scratch_bytes = 0;
  // working pictures of the whole-picture path in tiled layout (grow-only pool, one slot per picture)
  // ONE allocation per direction, picture i at element offset i * tiled_pic_elems, its planes at
  // tiled_plane_off[]: a wave that works across pictures reaches any picture with a multiply-add
  short *pool_org = nullptr, *pool_rec = nullptr; // pools of the call being issued: the context's own or the caller's (hmx_tpool)
  short *own_pool_org = nullptr, *own_pool_rec = nullptr; // allocations behind the plane-geometry entry points
  int pool_pics = 0;
  size_t tiled_pic_elems = 0;
  uint32_t tiled_plane_off[3] = {0, 0, 0};
  int tiled_cw = 0, tiled_ch = 0; // CTU grid of the call being issued
  int own_cw = 0, own_ch = 0;     // CTU grid the context's own pools were sized for
  bool resident_call = false;     // the call works on the caller's resident pools: no layout conversion
  ConvJob *d_jobs = nullptr;      // [2][n_pics*3]: to-tiled jobs, then from-tiled jobs
  int jobs_cap = 0;
  // whole-picture calls recorded as HIP graphs (see frame_intra)
  struct GraphEntry {
    uint64_t key;
    int n_pics;
    hipGraphExec_t exec;
    PicWork *d_work;
    uint64_t stamp;
  };
  std::vector<GraphEntry> graphs;
  uint64_t graph_clock = 0;
  int last_schedule = 0, last_groups = 1; // of the last whole-picture call (hmx_last_call_shape)
  bool onto_call = false;                 // this call reconstructs onto what the reconstruction planes already hold
  bool pipeline_conv = false;  // this call converts CTU row by CTU row, overlapped with the chain
  bool across_call = false;    // the call being issued uses the across-pictures schedule (interleaved pool)
  int level_mode_min_pics = 1; // measured: the level schedule is at least as fast as the wave schedule at every batch size
  // optional timing of the last whole-picture call: events around the layout conversions and the chain
  bool timing = false;
  hipEvent_t tev[4] = {};
  hipEvent_t tev_prep = nullptr; // packed schedule: behind the tables' prep kernels, when the timed call rebuilt them
  bool tev_prep_valid = false;
  bool tev_valid = false;
  // level schedule: picture groups run on side streams so that launches of different groups overlap
  static const int kMaxSide = 8;
  hipStream_t side[kMaxSide] = {};
  // layout conversions pipelined with the chain (across schedule): one stream for the conversions, one event per CTU row
  // and direction, one per (group, CTU row) for "this row is final"
  hipStream_t conv_stream = nullptr;
  std::vector<hipEvent_t> ev_rows; // [ch] converted in, [ch] converted out marker unused, then [groups][ch] row final
  hipEvent_t ev_conv_join = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join[kMaxSide] = {};
  int n_side = 0;
  // Argument arena: small per-call tables (picture planes, job lists) travel through a pinned host ring
  // and a device ring by asynchronous copies; the stream is synchronised only when the ring wraps.
  double *rdoq_wd = nullptr; // RDOQ per-lane records (hmx_rdoq.h), sized for rdoq_T lanes
  int *rdoq_wi = nullptr;
  RdoqBlock *rdoq_blocks = nullptr;
  EstBitsDev *rdoq_est = nullptr;
  int rdoq_T = 0, rdoq_est_cap = 0, rdoq_blocks_cap = 0;
  uint64_t rdoq_key = 0;      // of the block list and tables resident on the device
  bool rdoq_resident = false;
  size_t rdoq_class_n[4] = {0, 0, 0, 0}; // blocks of 32, 16, 8, 4 in the resident list
  uint64_t rdoq_in_key = 0;              // of the caller's arguments that produced the resident list (hmx_batch_xRateDistOptQuant)
  bool rdoq_join[2] = {false, false}; // side streams of the current RDOQ call still to be joined
  double rdoq_consts_h[4] = {};
  bool rdoq_consts_valid = false;
  double *rdoq_consts = nullptr; // k_rdoq_tiles: lambda [luma, chroma], then the Int64 factors of sign hiding
  int *d_mcmap = nullptr; // cell -> PU maps of the last motion-compensation call
  size_t mcmap_cap = 0;
  char *arena_h = nullptr, *arena_d = nullptr;
  size_t arena_cap = 0, arena_head = 0;
  // packed schedule (k_intra_packed): tables of the last call; rebuilt on the device when the pictures / plans change
  struct Packed {
    PackPic *d_pics = nullptr;
    PackRow *d_rows = nullptr;
    PackDesc *d_descs = nullptr;
    FTu *d_items = nullptr;
    uint32_t *d_done = nullptr;
    PackHdr *d_hdr = nullptr;
    size_t cap_pics = 0, cap_rows = 0, cap_descs = 0, cap_items = 0, cap_done = 0;
    uint64_t key = 0;
    bool valid = false;
    PackGeom G{};
    int n_wg = 0;
    uint64_t waves_bound = 0;
  } pk;
  int max_resident_waves = 0; // of k_intra_packed on this device
  int pack_I = 1;             // pictures per group (interleave domain of the pool) of the call being issued
  const hmx_levels *call_lev = nullptr; // the call's level planes (host array, valid while the call is issued)
  bool pk_pending = false;    // a packed launch was issued since the last check of its abort word
  // hmx_set_rdoq: xRateDistOptQuant as the quantiser of the next whole-picture encode calls
  struct ChainRdoq {
    int n = 0;                    // pictures described (1: one set for all pictures of a call); 0 = off
    EstBitsDev *d_est = nullptr;  // [n][2][4]
    size_t cap_est = 0;
    std::vector<double> lambda;   // [n][2]
    double *d_lambda = nullptr;   // [n][2], then the Int64 factors [n][2]
    size_t cap_lambda = 0;
    int max_waves = 0;            // resident waves of the RDOQ kernel variant
    uint64_t serial = 0;          // counts hmx_set_rdoq calls (part of the schedule key)
    uint64_t lambda_key = 0;      // of the multiplier table resident in d_lambda (serial + quantiser parameters)
    bool lambda_valid = false;
  } crq;
  // hmx_intra_plan_create_device: work buffers (grow-only) and a cache of freed slabs (a pipeline that rebuilds its plans every
  // batch gets the previous batch's memory back instead of a hipMalloc / hipFree pair per call)
  struct PlanDev {
    void *buf[12] = {};
    size_t cap[12] = {};
    unsigned long long *d_need = nullptr; // what a mode reads: [4 sizes][chroma, luma][35 modes] unit masks
    std::vector<std::pair<void *, size_t>> slabs;
  } pd;
  std::vector<hmx_sse> sse_out; // hmx_set_sse_output: per-picture distortion arrays of the next whole-picture encode calls
  uint64_t table_key = 0;     // of the picture table resident in d_jobs (whole-picture calls)
  bool table_valid = false;
  std::vector<unsigned char> table_bytes; // what table_key was hashed from (the table is re-used only when these bytes are equal)
  uint64_t table_salt = 0;
  // knobs, read once from the environment in hmx_create (A/B runs and the cross-checks of the tests)
  struct Knobs {
    int schedule = -1;     // HMX_INTRA_SCHEDULE: wave / level / packed (default: packed)
    int across = -1;       // HMX_INTRA_ACROSS: 0 keeps shared-plan batches of the level schedule per picture
    int streams = 0;       // HMX_INTRA_STREAMS: picture groups of the across schedule
    bool pipeline_conv = false, graph = false;
    int slots8 = 0;        // HMX_PACK_SLOTS8: 8 or 16 8x8 blocks per wave-item of the packed schedule: eight or four lanes per block (0: by batch size)
    int slots4 = 0;        // HMX_PACK_SLOTS4: 16 or 64 4x4 blocks per wave-item (0: by batch size)
    int pack_group = 0;    // HMX_PACK_GROUP: pictures per group, 1..64 (0: by batch size, see pack_group_size)
    int pack_waves = 0;    // HMX_PACK_WAVES: persistent waves (0: by batch size)
    int pack_sleep0 = -1, pack_sleep1 = -1; // HMX_PACK_SLEEP0 / 1: poll back-off, units of 64 clocks (-1: default)
    bool plan_one_stream = false; // HMX_PLAN_STREAMS=1: the luma and chroma level walks of the device plan builder one after the other (A/B)
    int plan_rows = 0;     // HMX_PLAN_ROWS: rows of the level table per picture the device plan builder starts with (0: from the picture size)
    bool rdoq_lane_only = false; // HMX_RDOQ_LANE: every block through the one-lane-per-block kernel (round 1's, A/B and cross-check)
  } knob;
};

// The tables of plans built on the device (hmx_intra_plan_create_device): ONE slab each for the sorted block lists and the
// level tables of all pictures of a call; the plans of the call point into them and share ownership.
struct PlanSet {
  FTu *d_ltus = nullptr;
  LevelRow *d_ltab = nullptr;
  size_t ltus_bytes = 0, ltab_bytes = 0;
  int refs = 0;
};
struct hmx_intra_plan {
  int n_levels = 0;            // picture-wide dependency levels (= level_chunks.size() for plans analysed on the host)
  PlanSet *set = nullptr;      // != NULL: built on the device; d_ltus / d_ltab point into the set's slabs, the other tables do not exist
  int n_diagonals = 0;
  FTu *d_tus = nullptr;
  Seg *d_segs = nullptr;
  uint32_t *d_seg_range = nullptr;
  uint32_t *d_wave_ctus = nullptr;
  std::vector<std::pair<uint32_t, uint32_t>> waves; // offset,count into d_wave_ctus
  FTu *d_ltus = nullptr;       // level schedule
  LevelRow *d_ltab = nullptr;
  std::vector<uint32_t> level_chunks; // waves needed per level
  std::vector<LevelRow> h_ltab;       // host copy of the level table
  uint64_t size_total[4] = {0, 0, 0, 0}; // blocks per transform size
  uint64_t serial = 0;                    // unique per plan: a freed plan's address may be handed out again
  // per CTU row: the first and the last dependency level that touches it (the layout conversions are pipelined by CTU
  // row: a row is converted in before its first level and out after its last one)
  std::vector<int> row_first_level, row_last_level;
  PicDev P;
  int n_tu = 0;
  int qp = 0, chroma_qp_offset = 0, slice_type = 0;
};

// 4x4 blocks per wave in the across-pictures level schedule: 16 = four lanes per block (one row each, through LDS like the 8x8 and
// 16x16 blocks), 64 = one lane per block (wave_chain_4_lane).  Measured at 1024 pictures of the 2160p mix: four lanes
// +3 % encoder direction, +16 % decoder direction (more, shorter waves); the one-lane form stays for A/B builds.
#ifndef HMX_SLOTS4
#define HMX_SLOTS4 16
#endif
constexpr int kSlots4 = HMX_SLOTS4; // across pictures
constexpr int kSlots4Own = 64;       // per-picture level kernel: one lane per block (four lanes: 51.5 vs 54.5 Gpx/s at 1024 pictures)
int fail(hmx_ctx *c, int code, const char *what, hipError_t e = hipSuccess);
static const int kQuantScales[6] = {26214, 23302, 20560, 18396, 16384, 14564}; // TComRom.cpp:293
static const int kInvQuantScales[6] = {40, 45, 51, 57, 64, 72};                // TComRom.cpp:298
#define HIPCHK(ctx, call)                                                   \
  do {                                                                      \
    hipError_t e_ = (call);                                                 \
    if (e_ != hipSuccess) return fail(ctx, HMX_ERR_DEVICE, #call, e_);      \
  } while (0)

static inline int ilog2i(int n) {
  int l = 0;
  while ((1 << l) < n) l++;
  return l;
}
// A block list resident on the device, bucketed by block size.
struct hmx_tu_list {
  DTu *d = nullptr;
  int off[5] = {0, 0, 0, 0, 0}, cnt[5] = {0, 0, 0, 0, 0}; // size classes 4 .. 64 (64: luma prediction units, hmx_batch_predIntra[_cost] only)
  int n = 0;
};
struct hmx_tpool {
  short *base = nullptr;
  int n_pics = 0, I = 1;   // groups of I pictures are interleaved quad by quad
  int cw = 0, ch = 0, ctu = 64, pic_w = 0, pic_h = 0;
  size_t pic_elems = 0;
  uint32_t plane_off[3] = {0, 0, 0};
};
inline TiledPlane tpool_plane(const hmx_tpool *t, int i, int p) {
  const int g0 = i / t->I * t->I, clog = ilog2i(t->ctu);
  const size_t base = (size_t)g0 * t->pic_elems + (size_t)t->plane_off[p] * t->I + (size_t)(i - g0) * 64;
  return TiledPlane{t->base + base, t->cw, p ? clog - 1 : clog, 64u * (unsigned)t->I};
}

// ---- host helpers that cross translation units ----
QuantDev make_qd(const hmx_qp &qp, int per_base, int slice_type);                 // hmx_core.hip
PicDev make_picdev(const hmx_ctx *c, const hmx_pic_param *pp);
PlanesDev to_dev(const hmx_pic *p);
LevelsDev to_dev(const hmx_levels *p);
void *arena_push(hmx_ctx *c, const void *src, size_t bytes);
int grow_dev(hmx_ctx *c, void **p, size_t *cap, size_t need);
int check_packed_abort(hmx_ctx *c);                                               // hmx_chain.hip
int pack_group_size(const hmx_ctx *c, int n_pics);
int launch_op(hmx_ctx *c, int op, int log2n, const ListArgs &A);                  // hmx_list.hip
unsigned long long intra_dependency_mask(int n_s, bool luma, int mode, unsigned long long avail); // hmx_plan.hip
int plan_host_tables(hmx_ctx *c, const hmx_intra_plan *pl); // device-built plan: fetch the level table for the level schedule / queries
// the packed schedule's kernel with RDOQ as the quantiser lives in hmx_chain_rdoq.hip
struct PackArgs;
int packed_rdoq_max_blocks(int *nb);
void launch_packed_rdoq(const PackArgs &A, bool sse, unsigned n_wg, hipStream_t st);

// ---- scalar drop-ins: one block through the batch kernels (host pointers in and out) ----
struct Scratch { // carve the context's device scratch
  hmx_ctx *c;
  size_t off = 0;
  template <typename T>
  T *take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T *p = reinterpret_cast<T *>(c->d_scratch + off);
    off += n * sizeof(T);
    return p;
  }
};

inline int up2d(hmx_ctx *c, void *dst, const void *src, size_t elem, int w, int h, size_t src_stride_elems) {
  HIPCHK(c, hipMemcpy2DAsync(dst, w * elem, src, src_stride_elems * elem, w * elem, h, hipMemcpyHostToDevice, c->stream));
  return HMX_OK;
}
inline int down2d(hmx_ctx *c, void *dst, size_t dst_stride_elems, const void *src, size_t elem, int w, int h) {
  HIPCHK(c, hipMemcpy2DAsync(dst, dst_stride_elems * elem, src, w * elem, w * elem, h, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HMX_OK;
}

struct One { // a one-block launch: dense N x N buffers at plane origin
  ListArgs A{};
  DTu *d_tu;
};
inline int one_block(hmx_ctx *c, Scratch &s, One &o, int n, int plane, unsigned mode, unsigned flags, const PicDev &P) {
  DTu h{};
  h.t.x = h.t.y = 0;
  h.t.log2n = (uint8_t)ilog2i(n);
  h.t.plane = (uint8_t)plane;
  h.t.mode = (uint8_t)(mode > 255 ? 255 : mode);
  h.t.flags = (uint8_t)flags;
  h.idx = 0;
  o.d_tu = s.take<DTu>(1);
  HIPCHK(c, hipMemcpyAsync(o.d_tu, &h, sizeof(h), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream)); // h is a stack object
  o.A.tus = o.d_tu;
  o.A.n = 1;
  o.A.P = P;
  return HMX_OK;
}
inline bool size_ok(int w, int h) { return w == h && (w == 4 || w == 8 || w == 16 || w == 32); }
// intra prediction also runs at 64 x 64, the prediction unit of a 64 x 64 coding unit (TComPrediction.cpp:343-345 asserts 4..128)
inline bool size_ok_intra(int w, int h) { return size_ok(w, h) || (w == 64 && h == 64); }

inline PicDev scalar_picdev(hmx_ctx *c, const hmx_qp *qp, int per_base, int slice_type, int sign_hide) {
  PicDev P{};
  P.pic_w = P.pic_h = 1 << 14;
  P.ctu = c->cfg.ctu_size;
  P.bit_depth = c->cfg.bit_depth;
  P.sign_hide = sign_hide;
  hmx_qp q = qp ? *qp : hmx_qp{0, 0, 0, 15};
  P.qd[0] = P.qd[1] = make_qd(q, per_base, slice_type);
  return P;
}

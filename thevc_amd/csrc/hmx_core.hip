// hmx_core.hip: context, device-memory plumbing, quantiser parameters, argument arena -- part of libhmx (include/hmx.h), gfx950.  See hmx_host.h for how the library is cut into translation units.
#include "hmx_host.h"

int fail(hmx_ctx *c, int code, const char *what, hipError_t e) {
  if (c) {
    c->err = what;
    if (e != hipSuccess) {
      c->err += ": ";
      c->err += hipGetErrorString(e);
    }
  }
  return code;
}
static int chroma_scale(int idx) { // g_aucChromaScale[58], TComRom.cpp:380
  static const unsigned char mid[13] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37};
  return idx < 30 ? idx : (idx >= 43 ? idx - 6 : mid[idx - 30]);
}

extern "C" hmx_qp hmx_setQPforQuant(int qpy, int text_type, int qp_bd_offset, int chroma_qp_offset) {
  int q;
  if (text_type == HMX_TEXT_LUMA)
    q = qpy + qp_bd_offset;
  else {
    q = std::min(57, std::max(-qp_bd_offset, qpy + chroma_qp_offset));
    q = q < 0 ? q + qp_bd_offset : chroma_scale(q) + qp_bd_offset;
  }
  hmx_qp r = {q, q / 6, q % 6, 15 + q / 6};
  return r;
}

QuantDev make_qd(const hmx_qp &qp, int per_base, int slice_type) {
  QuantDev d;
  d.q = kQuantScales[qp.rem];
  d.per_qbits = per_base >= 0 ? per_base : qp.per;
  d.iq_scale = kInvQuantScales[qp.rem] << qp.per;
  d.rnd_factor = slice_type == HMX_I_SLICE ? 171 : 85;
  return d;
}

PicDev make_picdev(const hmx_ctx *c, const hmx_pic_param *pp) {
  PicDev P;
  P.pic_w = pp->pic_w;
  P.pic_h = pp->pic_h;
  P.ctu = c->cfg.ctu_size;
  P.bit_depth = c->cfg.bit_depth;
  P.sign_hide = pp->sign_hide;
  const int bd = 6 * (c->cfg.bit_depth - 8);
  P.qd[0] = make_qd(hmx_setQPforQuant(pp->qp, HMX_TEXT_LUMA, bd, 0), -1, pp->slice_type);
  P.qd[1] = make_qd(hmx_setQPforQuant(pp->qp, HMX_TEXT_CHROMA, bd, pp->chroma_qp_offset), -1, pp->slice_type);
  return P;
}

// Tuning knobs: read from the environment ONCE, in hmx_create; hmx_set_option changes one afterwards (A/B runs, and the
// parity tests that hold the schedules against each other).  value == NULL restores the default.
static const char *const kKnobNames[] = {"HMX_INTRA_SCHEDULE", "HMX_INTRA_ACROSS", "HMX_INTRA_STREAMS", "HMX_PIPELINE_CONV", "HMX_GRAPH",
                                         "HMX_PACK_SLOTS4",    "HMX_PACK_SLOTS8",    "HMX_PACK_GROUP",      "HMX_PACK_WAVES",    "HMX_PACK_SLEEP0",   "HMX_PACK_SLEEP1",  "HMX_PLAN_ROWS",      "HMX_PLAN_STREAMS",
                                         "HMX_RDOQ_LANE"};
static bool apply_knob(hmx_ctx *c, const char *name, const char *v) {
  auto &k = c->knob;
  const std::string n(name);
  if (n == "HMX_INTRA_SCHEDULE") k.schedule = !v ? -1 : v[0] == 'w' ? 0 : v[0] == 'l' ? 1 : v[0] == 'p' ? 3 : -1;
  else if (n == "HMX_INTRA_ACROSS") k.across = !v ? -1 : v[0] != '0';
  else if (n == "HMX_INTRA_STREAMS") k.streams = v ? atoi(v) : 0;
  else if (n == "HMX_PIPELINE_CONV") k.pipeline_conv = v && v[0] != '0';
  else if (n == "HMX_GRAPH") k.graph = v != nullptr;
  else if (n == "HMX_PACK_SLOTS4") k.slots4 = !v ? 0 : atoi(v) == 16 ? 16 : 64;
  else if (n == "HMX_PACK_SLOTS8") k.slots8 = !v ? 0 : atoi(v) == 16 ? 16 : 8;
  else if (n == "HMX_PACK_GROUP") k.pack_group = v ? std::min(64, std::max(1, atoi(v))) : 0;
  else if (n == "HMX_PACK_WAVES") k.pack_waves = v ? std::max(1, atoi(v)) : 0;
  else if (n == "HMX_PACK_SLEEP0") k.pack_sleep0 = v ? std::max(0, atoi(v)) : -1;
  else if (n == "HMX_PACK_SLEEP1") k.pack_sleep1 = v ? std::max(0, atoi(v)) : -1;
  else if (n == "HMX_RDOQ_LANE") k.rdoq_lane_only = v && v[0] != '0';
  else if (n == "HMX_PLAN_ROWS") k.plan_rows = v ? std::max(1, atoi(v)) : 0;
  else if (n == "HMX_PLAN_STREAMS") k.plan_one_stream = v && atoi(v) == 1;
  else return false;
  return true;
}
extern "C" int hmx_set_option(hmx_ctx *c, const char *name, const char *value) {
  if (!c || !name) return HMX_ERR_ARG;
  return apply_knob(c, name, value) ? HMX_OK : fail(c, HMX_ERR_ARG, "hmx_set_option: unknown option");
}

extern "C" int hmx_create(const hmx_config *cfg, hmx_ctx **out) {
  if (!cfg || !out) return HMX_ERR_ARG;
  if (cfg->bit_depth < 8 || cfg->bit_depth > 12 || (cfg->ctu_size != 64 && cfg->ctu_size != 32 && cfg->ctu_size != 16))
    return HMX_ERR_ARG;
  hmx_ctx *c = new hmx_ctx;
  c->cfg = *cfg;
  hipError_t e = hipSetDevice(cfg->device);
  if (e != hipSuccess) {
    delete c;
    return HMX_ERR_DEVICE;
  }
  if (cfg->stream)
    c->stream = (hipStream_t)cfg->stream;
  else {
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete c;
      return HMX_ERR_DEVICE;
    }
    c->own_stream = true;
  }
  c->scratch_bytes = 1 << 20;
  e = hipMalloc((void **)&c->d_scratch, c->scratch_bytes);
  if (e != hipSuccess) {
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
    return HMX_ERR_NOMEM;
  }
  for (const char *name : kKnobNames)
    if (const char *v = getenv(name)) apply_knob(c, name, v);
  *out = c;
  return HMX_OK;
}

extern "C" void hmx_destroy(hmx_ctx *c) {
  if (!c) return;
  hipStreamSynchronize(c->stream);
  hipFree(c->d_scratch);
  for (auto &e : c->graphs) {
    hipGraphExecDestroy(e.exec);
    hipFree(e.d_work);
  }
  hipFree(c->own_pool_org);
  hipFree(c->own_pool_rec);
  hipFree(c->d_jobs);
  for (void *b : c->pd.buf) hipFree(b);
  hipFree(c->pd.d_need);
  for (auto &sl : c->pd.slabs) hipFree(sl.first);
  if (c->arena_h) hipHostFree(c->arena_h);
  hipFree(c->arena_d);
  hipFree(c->d_mcmap);
  hipFree(c->rdoq_wd);
  hipFree(c->rdoq_wi);
  hipFree(c->rdoq_blocks);
  hipFree(c->rdoq_est);
  hipFree(c->rdoq_consts);
  hipFree(c->crq.d_est);
  hipFree(c->crq.d_lambda);
  hipFree(c->pk.d_pics);
  hipFree(c->pk.d_rows);
  hipFree(c->pk.d_descs);
  hipFree(c->pk.d_items);
  hipFree(c->pk.d_done);
  hipFree(c->pk.d_hdr);
  for (int g = 0; g < c->n_side; g++) {
    hipStreamDestroy(c->side[g]);
    hipEventDestroy(c->ev_join[g]);
  }
  if (c->ev_fork) hipEventDestroy(c->ev_fork);
  if (c->conv_stream) hipStreamDestroy(c->conv_stream);
  for (auto e : c->ev_rows) hipEventDestroy(e);
  if (c->ev_conv_join) hipEventDestroy(c->ev_conv_join);
  for (int i = 0; i < 4; i++)
    if (c->tev[i]) hipEventDestroy(c->tev[i]);
  if (c->tev_prep) hipEventDestroy(c->tev_prep);
  if (c->own_stream) hipStreamDestroy(c->stream);
  delete c;
}
extern "C" const char *hmx_last_error(const hmx_ctx *c) { return c ? c->err.c_str() : "null context"; }
extern "C" int hmx_sync(hmx_ctx *c) {
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return check_packed_abort(c);
}
extern "C" int hmx_malloc(hmx_ctx *c, size_t bytes, void **dptr) {
  hipError_t e = hipMalloc(dptr, bytes);
  if (e != hipSuccess) return fail(c, HMX_ERR_NOMEM, "hipMalloc", e);
  return HMX_OK;
}
extern "C" int hmx_free(hmx_ctx *c, void *dptr) {
  HIPCHK(c, hipFree(dptr));
  return HMX_OK;
}
extern "C" int hmx_upload(hmx_ctx *c, void *dst, const void *src, size_t bytes) {
  HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HMX_OK;
}
extern "C" int hmx_download(hmx_ctx *c, void *dst, const void *src, size_t bytes) {
  HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return check_packed_abort(c);
}
extern "C" int hmx_memset(hmx_ctx *c, void *dst, int value, size_t bytes) {
  HIPCHK(c, hipMemsetAsync(dst, value, bytes, c->stream));
  return HMX_OK;
}
extern "C" int hmx_event_create(hmx_ctx *c, void **ev) {
  hipEvent_t e;
  HIPCHK(c, hipEventCreate(&e));
  *ev = (void *)e;
  return HMX_OK;
}
extern "C" int hmx_event_record(hmx_ctx *c, void *ev) {
  HIPCHK(c, hipEventRecord((hipEvent_t)ev, c->stream));
  return HMX_OK;
}
extern "C" int hmx_event_elapsed_ms(hmx_ctx *c, void *a, void *b, float *ms) {
  HIPCHK(c, hipEventSynchronize((hipEvent_t)b));
  HIPCHK(c, hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
  return HMX_OK;
}
extern "C" int hmx_event_destroy(hmx_ctx *c, void *ev) {
  HIPCHK(c, hipEventDestroy((hipEvent_t)ev));
  return HMX_OK;
}

// ---- list launches ----
PlanesDev to_dev(const hmx_pic *p) {
  PlanesDev d;
  for (int i = 0; i < 3; i++) {
    d.p[i] = p ? p->plane[i] : nullptr;
    d.s[i] = p ? p->stride[i] : 0;
  }
  return d;
}
LevelsDev to_dev(const hmx_levels *p) {
  LevelsDev d;
  for (int i = 0; i < 3; i++) {
    d.p[i] = p ? p->plane[i] : nullptr;
    d.s[i] = p ? p->stride[i] : 0;
  }
  return d;
}

// device copy of a small host table, valid for the kernels issued after it on the context's stream
void *arena_push(hmx_ctx *c, const void *src, size_t bytes) {
  const size_t kCap = 8u << 20;
  if (!c->arena_h) {
    if (hipHostMalloc((void **)&c->arena_h, kCap, hipHostMallocDefault) != hipSuccess) return nullptr;
    if (hipMalloc((void **)&c->arena_d, kCap) != hipSuccess) return nullptr;
    c->arena_cap = kCap;
  }
  const size_t raw = bytes;
  bytes = (bytes + 255) & ~(size_t)255;
  if (bytes > c->arena_cap) return nullptr;
  if (c->arena_head + bytes > c->arena_cap) { // wrap: everything issued so far has consumed its tables after this
    if (hipStreamSynchronize(c->stream) != hipSuccess) return nullptr;
    c->arena_head = 0;
  }
  char *h = c->arena_h + c->arena_head, *d = c->arena_d + c->arena_head;
  memcpy(h, src, raw);
  if (hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) return nullptr;
  c->arena_head += bytes;
  return d;
}

int grow_dev(hmx_ctx *c, void **p, size_t *cap, size_t need) {
  if (need <= *cap) return HMX_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  hipFree(*p);
  *p = nullptr;
  *cap = 0;
  const size_t want = need + need / 16 + 256;
  if (hipMalloc(p, want) != hipSuccess) return fail(c, HMX_ERR_NOMEM, "hipMalloc packed schedule tables");
  *cap = want;
  return HMX_OK;
}

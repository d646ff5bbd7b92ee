"""Soak of the packed schedule: random batch sizes, packing groups, persistent-wave counts, 4x4 wave shapes, with and without
RDOQ and the distortion output, every picture with its own plan -- each run held against the oracle.  python3 tools/soak.py [runs]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
from thevc_amd import capi, workload  # noqa: E402

w, h = 264, 200


def soak(runs, seed, verbose=True):
  L = capi.lib()
  rng = np.random.default_rng(seed)
  cache = {}
  for it in range(runs):
      B = int(rng.choice([8, 10]))
      n = int(rng.choice([1, 2, 3, 5, 9, 17, 33, 70]))
      group = int(rng.choice([1, 2, 3, 4, 7, 64]))
      waves = int(rng.choice([3, 17, 256, 4096]))
      slots = int(rng.choice([16, 64]))
      slots8 = int(rng.choice([8, 16]))          # 8x8 blocks on eight or on four lanes (flat quantiser, slots4 = 64)
      zlev = bool(rng.integers(0, 2))            # levels in the reference's coefficient layout (cooperative 4x4 stores) or in planes
      dev_plans = bool(rng.integers(0, 2))       # plans built on the device
      rdoq = bool(rng.integers(0, 2))
      if rdoq:
          group = min(group, 2)
      qp = int(rng.integers(22, 38))
      ctx = capi.Context(bit_depth=B)
      ctx.set_option("HMX_PACK_GROUP", group)
      ctx.set_option("HMX_PACK_WAVES", waves)
      ctx.set_option("HMX_PACK_SLOTS4", slots)
      ctx.set_option("HMX_PACK_SLOTS8", slots8)
      pp = capi.PicParam(w, h, qp, 0, capi.I_SLICE, 1)
      seeds = [int(rng.integers(0, 12)) for _ in range(n)]
      tus = [workload.with_cbf_ctx(workload.make_tus(8000 + sd, w, h, "mix")) for sd in seeds]
      if dev_plans:
          cat = np.ascontiguousarray(np.concatenate(tus), capi.TU_DTYPE)
          d_cat = ctx.to_device(cat)
          plans = ctx.intra_plans_device(d_cat.ptr, np.concatenate([[0], np.cumsum([len(t) for t in tus])]), pp)
          d_cat.free()
      else:
          plans = ctx.intra_plans(tus, pp)
      orgs = [workload.make_planes(8100 + sd, w, h, B, "texture") for sd in seeds]
      ests = [[workload.make_est_bits(8200 + 8 * sd + k) for k in range(8)] for sd in seeds]
      lams = workload.rdoq_lambdas(qp)
      if rdoq:
          ctx.set_rdoq([(ests[i], lams[0], lams[1]) for i in range(n)])
      d_org = [capi.DevPicture(ctx, w, h).upload(o) for o in orgs]
      d_rec = [capi.DevPicture(ctx, w, h).zero() for _ in range(n)]
      d_lev = [capi.DevLevelsZ(ctx, w, h) if zlev else capi.DevPicture(ctx, w, h, dtype=np.int32).zero() for _ in range(n)]
      A = lambda lst, T: (T * n)(*[x.as_pic() for x in lst])
      parr = (C.c_void_p * n)(*[p.value for p in plans])
      for rep in range(2):
          ctx._chk(L.hmx_frame_intra_encode_multi(ctx.h, parr, n, A(d_org, capi.Pic), A(d_rec, capi.Pic), A(d_lev, capi.Levels)))
      ctx.sync()
      d_dec = [capi.DevPicture(ctx, w, h).zero() for _ in range(n)]  # and back: the decoder direction on the same levels
      ctx._chk(L.hmx_frame_intra_decode_multi(ctx.h, parr, n, A(d_dec, capi.Pic), A(d_lev, capi.Levels)))
      ctx.sync()
      for i in range(n):
          key = (B, qp, seeds[i], rdoq)
          if key not in cache:
              cache[key] = (ol.o_intra_frame_encode_rdoq(tus[i], w, h, B, qp, orgs[i], ests[i], lams) if rdoq
                            else ol.o_intra_frame_encode(tus[i], w, h, B, qp, orgs[i]))
          rr, lr = cache[key]
          rec, dec = d_rec[i].download(), d_dec[i].download()
          lev = d_lev[i].to_planes(tus[i]) if zlev else d_lev[i].download()
          for p in range(3):
              assert np.array_equal(rec[p], rr[p]) and np.array_equal(lev[p], lr[p]) and np.array_equal(dec[p], rr[p]), \
                  (it, i, p, B, n, group, waves, slots, slots8, zlev, dev_plans, rdoq)
      if verbose:
        print(f"run {it}: B {B} pictures {n} group {group} waves {waves} slots4 {slots} slots8 {slots8} zlev {int(zlev)} device plans {int(dev_plans)} "
              f"rdoq {int(rdoq)} qp {qp}: ok", flush=True)
      ctx.close()
  if verbose:
    print("soak: all runs identical to the oracle")


if __name__ == "__main__":
    soak(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(time.time()) & 0xffff)

"""ctypes access to the CPU oracle (oracle/libhmx_oracle.so) and, when present, to the compiled
reference (oracle/_ref/libhmref.so).  TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg import this module; the product never does."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libhmx_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libhmref.so")

i16p = np.ctypeslib.ndpointer(np.int16, flags="C")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C")
ci = C.c_int
cu = C.c_uint


class QuantCfg(C.Structure):
    _fields_ = [("per", ci), ("rem", ci), ("per_qbits", ci), ("intra_slice", ci),
                ("sign_hide", ci), ("scan_idx", ci)]


class Qp(C.Structure):
    _fields_ = [("qp", ci), ("per", ci), ("rem", ci)]


class EstBits(C.Structure):  # hmo_est_bits == estBitsSbacStruct (COM/TComTrQuant.h:59-72), 1/32768 bit
    _fields_ = [("sig_cg", (C.c_int32 * 2) * 2), ("sig", (C.c_int32 * 2) * 42), ("last_x", C.c_int32 * 32),
                ("last_y", C.c_int32 * 32), ("greater1", (C.c_int32 * 2) * 24), ("greater2", (C.c_int32 * 2) * 6),
                ("cbf", (C.c_int32 * 2) * 15), ("root_cbf", (C.c_int32 * 2) * 4), ("scan_zigzag", C.c_int32 * 2),
                ("scan_nonzigzag", C.c_int32 * 2)]


class RdoqCfg(C.Structure):  # hmo_rdoq_cfg
    _fields_ = [("per", ci), ("rem", ci), ("is_luma", ci), ("is_intra", ci), ("scan_idx", ci), ("root_cbf", ci),
                ("cbf_ctx", ci), ("sign_hide", ci), ("lam", C.c_double)]


def make_est_bits(rng):
    """A plausible CABAC bit-estimate table: every context holds a probability p of the bin being 1;
    bits[0] = -log2(1-p), bits[1] = -log2(p) in 1/32768 bit (TEncBinCABAC's entropy-bits scale)."""
    e = EstBits()

    def pair(dst):
        p = float(rng.uniform(0.03, 0.97))
        dst[0] = int(round(-np.log2(1 - p) * 32768))
        dst[1] = int(round(-np.log2(p) * 32768))

    for name, n in (("sig_cg", 2), ("sig", 42), ("greater1", 24), ("greater2", 6), ("cbf", 15), ("root_cbf", 4)):
        arr = getattr(e, name)
        for i in range(n):
            pair(arr[i])
    for i in range(32):  # cumulative cost of the truncated-unary last-position prefix
        e.last_x[i] = int(rng.integers(8000, 60000) * (1 + i // 4))
        e.last_y[i] = int(rng.integers(8000, 60000) * (1 + i // 4))
    pair(e.scan_zigzag)
    pair(e.scan_nonzigzag)
    return e


class FrameCfg(C.Structure):
    _fields_ = [("pic_w", ci), ("pic_h", ci), ("ctu", ci), ("B", ci), ("qp", ci),
                ("chroma_qp_offset", ci), ("sign_hide", ci), ("inter_slice", ci)]


TU_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("log2n", "u1"), ("plane", "u1"),
                     ("mode", "u1"), ("flags", "u1")])
PU_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("w", "u1"), ("h", "u1"), ("ref0", "u1"),
                     ("ref1", "u1"), ("mv0x", "<i2"), ("mv0y", "<i2"), ("mv1x", "<i2"),
                     ("mv1y", "<i2")])

_oracle = None
_ref = None


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO) or (os.path.getmtime(ORACLE_SO) <
                                             os.path.getmtime(os.path.join(ROOT, "oracle", "hmx_oracle.c"))):
            build_oracle()
        L = C.CDLL(ORACLE_SO)
        L.hmo_dct_matrix.argtypes = [ci, i16p]
        L.hmo_dst_matrix.argtypes = [i16p]
        L.hmo_scan.argtypes = [ci, ci]
        L.hmo_scan.restype = C.POINTER(C.c_uint32)
        L.hmo_fwd_pass.argtypes = [i16p, i16p, ci, ci, ci, ci]
        L.hmo_inv_pass.argtypes = [i16p, i16p, ci, ci, ci, ci]
        L.hmo_xTrMxN.argtypes = [i16p, i16p, ci, cu, ci]
        L.hmo_xITrMxN.argtypes = [i16p, i16p, ci, cu, ci]
        L.hmo_xT.argtypes = [cu, i16p, ci, i32p, ci, ci]
        L.hmo_xIT.argtypes = [cu, i32p, i16p, ci, ci, ci]
        L.hmo_xTransformSkip.argtypes = [i16p, ci, i32p, ci, ci]
        L.hmo_xITransformSkip.argtypes = [i32p, i16p, ci, ci, ci]
        L.hmo_setQPforQuant.argtypes = [ci, ci, ci, ci]
        L.hmo_setQPforQuant.restype = Qp
        L.hmo_coef_scan_idx.argtypes = [ci, ci, ci, ci]
        L.hmo_xQuant.argtypes = [i32p, i32p, ci, ci, C.POINTER(QuantCfg), C.POINTER(C.c_uint32)]
        L.hmo_xRateDistOptQuant.argtypes = [i32p, i32p, ci, ci, C.POINTER(RdoqCfg), C.POINTER(EstBits), C.POINTER(C.c_uint32)]
        L.hmo_xRateDistOptQuant.restype = None
        L.hmo_xDeQuant.argtypes = [i32p, i32p, ci, ci, ci, ci]
        L.hmo_xDeQuant_scaled.argtypes = [i32p, i32p, ci, ci, ci, i32p]
        L.hmo_xDeQuant_scaled.restype = None
        L.hmo_arlCoeff.argtypes = [i32p, i32p, ci, ci, C.POINTER(QuantCfg), ci, C.c_void_p]
        L.hmo_xQuant_scaled.argtypes = [i32p, i32p, ci, ci, C.POINTER(QuantCfg), C.POINTER(C.c_uint32), C.c_void_p]
        L.hmo_xQuant_scaled.restype = None
        L.hmo_xRateDistOptQuant_scaled.argtypes = [i32p, i32p, ci, ci, C.POINTER(RdoqCfg), C.POINTER(EstBits), C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p]
        L.hmo_xRateDistOptQuant_scaled.restype = None
        L.hmo_arlCoeff.restype = None
        L.hmo_transformNxN.argtypes = [i16p, ci, i32p, ci, ci, cu, ci, ci, C.POINTER(QuantCfg),
                                       C.POINTER(C.c_uint32)]
        L.hmo_invtransformNxN.argtypes = [ci, cu, i16p, ci, i32p, ci, ci, ci, ci, ci]
        L.hmo_intra_avail.argtypes = [ci, ci, ci, ci, ci, ci, u8p]
        L.hmo_fillReferenceSamples.argtypes = [C.c_void_p, ci, u8p, ci, ci, ci, ci, i32p]
        L.hmo_filterAdi.argtypes = [i32p, ci]
        L.hmo_use_filtered_refs.argtypes = [ci, ci]
        L.hmo_predIntraLumaAng.argtypes = [i32p, ci, i16p, ci, ci, ci]
        L.hmo_predIntraChromaAng.argtypes = [i32p, ci, i16p, ci, ci, ci]
        for n in ("hmo_filterHorLuma", "hmo_filterHorChroma"):
            getattr(L, n).argtypes = [C.c_void_p, ci, C.c_void_p, ci, ci, ci, ci, ci, ci]
        for n in ("hmo_filterVerLuma", "hmo_filterVerChroma"):
            getattr(L, n).argtypes = [C.c_void_p, ci, C.c_void_p, ci, ci, ci, ci, ci, ci, ci]
        L.hmo_predInterLumaBlk.argtypes = [C.c_void_p, ci, ci, ci, ci, ci, i16p, ci, ci, ci]
        L.hmo_predInterChromaBlk.argtypes = [C.c_void_p, ci, ci, ci, ci, ci, i16p, ci, ci, ci]
        L.hmo_addAvg.argtypes = [i16p, ci, i16p, ci, i16p, ci, ci, ci, ci]
        L.hmo_addClip.argtypes = [i16p, ci, i16p, ci, i16p, ci, ci, ci, ci]
        L.hmo_subtract.argtypes = [i16p, ci, i16p, ci, i16p, ci, ci, ci]
        L.hmo_extendPicBorder.argtypes = [C.c_void_p, ci, ci, ci, ci, ci]
        L.hmo_clipMv.argtypes = [C.POINTER(ci), C.POINTER(ci), ci, ci, ci, ci, ci]
        P3 = C.c_void_p * 3
        I3 = ci * 3
        L.hmo_intra_frame_encode.argtypes = [C.POINTER(FrameCfg), C.c_void_p, ci, P3, I3, P3, I3, P3]
        L.hmo_intra_frame_encode_rdoq.argtypes = [C.POINTER(FrameCfg), C.c_void_p, ci, P3, I3, P3, I3, P3, C.POINTER(EstBits),
                                                  C.POINTER(C.c_double)]
        L.hmo_intra_frame_decode.argtypes = [C.POINTER(FrameCfg), C.c_void_p, ci, P3, I3, P3]
        L.hmo_mc_frame.argtypes = [C.c_void_p, ci, ci, C.c_void_p, I3, P3, I3]
        _oracle = L
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    """The compiled reference (only in the build container; absent on the GPU box)."""
    global _ref
    if _ref is None:
        L = C.CDLL(REF_SO)
        L.ref_init.argtypes = [ci, ci, ci, ci]
        L.ref_tables.argtypes = [i16p, i16p, i16p, i16p, i16p, i32p, i32p, u8p]
        L.ref_scan.argtypes = [ci, ci, u32p]
        L.ref_partialButterfly.argtypes = [ci, i16p, i16p, ci, ci]
        L.ref_partialButterflyInverse.argtypes = [ci, i16p, i16p, ci, ci]
        L.ref_fastForwardDst.argtypes = [i16p, i16p, ci]
        L.ref_fastInverseDst.argtypes = [i16p, i16p, ci]
        L.ref_xTrMxN.argtypes = [i16p, i16p, ci, cu]
        L.ref_xITrMxN.argtypes = [i16p, i16p, ci, cu]
        L.ref_xT.argtypes = [cu, i16p, cu, i32p, ci]
        L.ref_xIT.argtypes = [cu, i32p, i16p, cu, ci]
        L.ref_xTransformSkip.argtypes = [i16p, cu, i32p, ci]
        L.ref_xITransformSkip.argtypes = [i32p, i16p, cu, ci]
        L.ref_setQPforQuant.argtypes = [ci, ci, ci, ci, i32p]
        L.ref_xDeQuant.argtypes = [ci, ci, ci, ci, i32p, i32p, ci]
        L.ref_xDeQuant_scaled.argtypes = [ci, ci, ci, ci, i32p, i32p, i32p, ci]
        L.ref_transformNxN.argtypes = [ci, ci, ci, ci, ci, ci, ci, i16p, cu, i32p, ci,
                                       C.POINTER(C.c_uint32)]
        L.ref_invtransformNxN.argtypes = [ci, ci, ci, cu, i16p, cu, i32p, ci, ci]
        L.ref_fillReferenceSamples.argtypes = [C.c_void_p, ci, u8p, ci, ci, ci, i32p]
        L.ref_set_recon.argtypes = [i16p, i16p, i16p]
        L.ref_initAdiPattern.argtypes = [ci, ci, ci, ci, ci, i32p]
        L.ref_initAdiPatternChroma.argtypes = [ci, ci, ci, ci, ci, i32p]
        L.ref_predIntraLumaAng.argtypes = [i32p, ci, i16p, cu, ci]
        L.ref_predIntraChromaAng.argtypes = [i32p, ci, i16p, cu, ci]
        for n in ("ref_filterHorLuma", "ref_filterHorChroma"):
            getattr(L, n).argtypes = [C.c_void_p, ci, C.c_void_p, ci, ci, ci, ci, ci]
        for n in ("ref_filterVerLuma", "ref_filterVerChroma"):
            getattr(L, n).argtypes = [C.c_void_p, ci, C.c_void_p, ci, ci, ci, ci, ci, ci]
        L.ref_predInterBlk.argtypes = [ci, ci, ci, ci, ci, ci, ci, i16p, i16p, i16p, ci]
        L.ref_clipMv.argtypes = [ci, ci, C.POINTER(ci), C.POINTER(ci)]
        L.ref_addAvg.argtypes = [C.c_void_p * 3, C.c_void_p * 3, C.c_void_p * 3, ci, ci]
        L.ref_extended_luma.argtypes = [i16p]
        L.ref_intra_frame_encode.argtypes = [C.c_void_p, ci, ci, i16p, i16p, i16p, i16p, i16p, i16p, i32p, i32p, i32p]
        _ref = L
    return _ref


def ptr(a, elem_offset=0):
    """Raw pointer into a numpy array at an element offset (for origin-inside-plane arguments)."""
    return C.c_void_p(a.ctypes.data + elem_offset * a.itemsize)


# ---- convenience wrappers over the oracle used by the parity tests ----

def quant_cfg(per, rem, intra_slice=1, sign_hide=1, scan_idx=3, per_qbits=None):
    return QuantCfg(per, rem, per if per_qbits is None else per_qbits, intra_slice, sign_hide,
                    scan_idx)


def o_transformNxN(resi, N, B, mode, ts, cfg, bypass=0):
    resi = np.ascontiguousarray(resi, np.int16)
    lvl = np.zeros(N * N, np.int32)
    s = C.c_uint32(0)
    oracle().hmo_transformNxN(resi, N, lvl, N, B, mode, ts, bypass, C.byref(cfg), C.byref(s))
    return lvl.reshape(N, N), s.value


def o_invtransformNxN(lvl, N, B, mode, per, rem, ts, bypass=0):
    lvl = np.ascontiguousarray(lvl, np.int32).reshape(-1)
    resi = np.zeros(N * N, np.int16)
    oracle().hmo_invtransformNxN(bypass, mode, resi, N, lvl, N, B, per, rem, ts)
    return resi.reshape(N, N)


def o_intra_pred(rec_plane, stride, x, y, N, mode, B, pic_w, pic_h, chroma, ctu=64):
    """refs from a recon plane (flat int16 array, origin at element 0) -> N x N prediction."""
    L = oracle()
    flags = np.zeros(65, np.uint8)
    c = 1 if chroma else 0
    nav = L.hmo_intra_avail(x << c, y << c, N << c, pic_w, pic_h, ctu, flags)
    W = 2 * N + 1
    adi = np.zeros(2 * W * W, np.int32)
    L.hmo_fillReferenceSamples(ptr(rec_plane, y * stride + x), stride, flags, nav, 2 if chroma else 4,
                               N, B, adi)
    pred = np.zeros((N, N), np.int16)
    if chroma:
        L.hmo_predIntraChromaAng(adi, mode, pred.reshape(-1), N, N, B)
    else:
        L.hmo_filterAdi(adi, N)
        L.hmo_predIntraLumaAng(adi, mode, pred.reshape(-1), N, N, B)
    return pred


def frame_cfg(w, h, B, qp, sign_hide=1, chroma_qp_offset=0, ctu=64, inter_slice=0):
    return FrameCfg(w, h, ctu, B, qp, chroma_qp_offset, sign_hide, inter_slice)


def o_intra_frame_encode(tus, w, h, B, qp, org, sign_hide=1):
    """oracle: decisions + original planes -> (recon planes, level planes)"""
    cfg = frame_cfg(w, h, B, qp, sign_hide)
    rec = [np.zeros_like(p) for p in org]
    lev = [np.zeros(p.shape, np.int32) for p in org]
    P3, I3 = C.c_void_p * 3, C.c_int * 3
    st = I3(w, w // 2, w // 2)
    t = np.ascontiguousarray(tus, TU_DTYPE)
    oracle().hmo_intra_frame_encode(C.byref(cfg), t.ctypes.data, len(t), P3(*[p.ctypes.data for p in org]), st,
                                    P3(*[p.ctypes.data for p in rec]), st, P3(*[p.ctypes.data for p in lev]))
    return rec, lev


def o_intra_frame_encode_rdoq(tus, w, h, B, qp, org, ests, lambdas, sign_hide=1):
    """oracle: the same chain with xRateDistOptQuant as the quantiser; ests = 8 EstBits [luma, chroma][4 sizes], lambdas = (luma,
    chroma); the blocks' cbf contexts ride in bits 4..7 of tus['flags']"""
    cfg = frame_cfg(w, h, B, qp, sign_hide)
    rec = [np.zeros_like(p) for p in org]
    lev = [np.zeros(p.shape, np.int32) for p in org]
    P3, I3 = C.c_void_p * 3, C.c_int * 3
    st = I3(w, w // 2, w // 2)
    t = np.ascontiguousarray(tus, TU_DTYPE)
    est_arr = (EstBits * 8)()
    for k in range(8):  # any structure with estBitsSbacStruct's layout (this module's EstBits, capi.EstBits)
        assert C.sizeof(ests[k]) == C.sizeof(EstBits)
        C.memmove(C.byref(est_arr[k]), C.byref(ests[k]), C.sizeof(EstBits))
    lam = (C.c_double * 2)(*lambdas)
    oracle().hmo_intra_frame_encode_rdoq(C.byref(cfg), t.ctypes.data, len(t), P3(*[p.ctypes.data for p in org]), st,
                                         P3(*[p.ctypes.data for p in rec]), st, P3(*[p.ctypes.data for p in lev]), est_arr, lam)
    return rec, lev


def r_intra_frame_encode(tus, w, h, B, qp, org, sign_hide=1):
    """the compiled reference's own functions chained over the same decisions"""
    R = ref()
    R.ref_init(B, w, h, sign_hide)
    rec = [np.zeros(p.shape, np.int16) for p in org]
    lev = [np.zeros(p.shape, np.int32) for p in org]
    t = np.ascontiguousarray(tus, TU_DTYPE)
    o = [np.ascontiguousarray(p, np.int16).reshape(-1) for p in org]
    R.ref_intra_frame_encode(t.ctypes.data, len(t), qp, o[0], o[1], o[2], rec[0].reshape(-1), rec[1].reshape(-1),
                             rec[2].reshape(-1), lev[0].reshape(-1), lev[1].reshape(-1), lev[2].reshape(-1))
    return rec, lev


def o_rdoq(coef, N, B, cfg, est):
    coef = np.ascontiguousarray(coef, np.int32).reshape(-1)
    lvl = np.zeros(N * N, np.int32)
    s = C.c_uint32(0)
    oracle().hmo_xRateDistOptQuant(coef, lvl, N, B, C.byref(cfg), C.byref(est), C.byref(s))
    return lvl.reshape(N, N), s.value


def scaling_tables(rng, N, B, rem, flat=False):
    """Per-position tables as HM's setScalingList leaves them for one (list, remainder, size): quant coefficient = (quantScale << 4) /
    list entry (processScalingListEnc, TComTrQuant.cpp:2953-2977), error scale from it (setErrScaleCoeff :2794-2818), de-quantiser
    coefficient = invQuantScale * list entry (:2979-3003).  Entries 1..255 at random (or the flat 16)."""
    q6, iq6 = (26214, 23302, 20560, 18396, 16384, 14564), (40, 45, 51, 57, 64, 72)
    entry = np.full(N * N, 16, np.int64) if flat else rng.integers(1, 256, N * N).astype(np.int64)
    qtab = ((q6[rem] << 4) // entry).astype(np.int32)
    lg = int(np.log2(N))
    e = np.float64(1 << 15) * np.float64(2.0) ** np.float64(-2 * (15 - B - lg))
    estab = e / qtab.astype(np.float64) / qtab.astype(np.float64) / np.float64(1 << (2 * (B - 8)))
    return qtab, estab, (iq6[rem] * entry).astype(np.int32)


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def o_arl(coef, N, B, cfg, rdoq, qtab=None):
    coef = np.ascontiguousarray(coef, np.int32).reshape(-1)
    qtab = None if qtab is None else np.ascontiguousarray(qtab, np.int32)
    arl = np.zeros(N * N, np.int32)
    oracle().hmo_arlCoeff(coef, arl, N, B, C.byref(cfg), int(rdoq), _vp(qtab))
    return arl.reshape(N, N)


def o_quant_scaled(coef, N, B, cfg, qtab):
    coef, qtab = np.ascontiguousarray(coef, np.int32).reshape(-1), np.ascontiguousarray(qtab, np.int32)
    lvl, s = np.zeros(N * N, np.int32), C.c_uint32(0)
    oracle().hmo_xQuant_scaled(coef, lvl, N, B, C.byref(cfg), C.byref(s), _vp(qtab))
    return lvl.reshape(N, N), s.value


def o_rdoq_scaled(coef, N, B, cfg, est, qtab, estab):
    coef = np.ascontiguousarray(coef, np.int32).reshape(-1)
    qtab, estab = np.ascontiguousarray(qtab, np.int32), np.ascontiguousarray(estab, np.float64)
    lvl, s = np.zeros(N * N, np.int32), C.c_uint32(0)
    oracle().hmo_xRateDistOptQuant_scaled(coef, lvl, N, B, C.byref(cfg), C.byref(est), C.byref(s), _vp(qtab), _vp(estab))
    return lvl.reshape(N, N), s.value


def r_quant_arl(coef, N, qpy, qp_base, slice_type, ttype, is_intra, dir_mode, tr_idx, rdoq, lam, est, qtab=None, estab=None):
    """the compiled reference's xQuant with AdaptiveQpSelection on (flat branch or xRateDistOptQuant): levels, pArlDes, uiAcSum;
    qtab / estab: the per-position tables of a scaling list for the call"""
    R = ref()
    R.ref_xQuant_arl.argtypes = [ci, ci, ci, ci, ci, ci, ci, ci, C.c_double, C.POINTER(EstBits), i32p, i32p, i32p, ci, C.POINTER(C.c_uint32),
                                 C.c_void_p, C.c_void_p]
    R.ref_xQuant_arl.restype = None
    coef = np.ascontiguousarray(coef, np.int32).reshape(-1).copy()
    qtab = None if qtab is None else np.ascontiguousarray(qtab, np.int32)
    estab = None if estab is None else np.ascontiguousarray(estab, np.float64)
    lvl, arl = np.zeros(N * N, np.int32), np.zeros(N * N, np.int32)
    s = C.c_uint32(0)
    R.ref_xQuant_arl(qpy, qp_base, slice_type, ttype, is_intra, dir_mode, tr_idx, int(rdoq), lam, C.byref(est), coef, lvl, arl, N, C.byref(s),
                     _vp(qtab), _vp(estab))
    return lvl.reshape(N, N), arl.reshape(N, N), s.value


def r_rdoq(coef, N, qpy, slice_type, ttype, is_intra, dir_mode, tr_idx, lam, est):
    """the compiled reference's xRateDistOptQuant; ref().ref_init(B, ...) must have been called"""
    R = ref()
    R.ref_xRateDistOptQuant.argtypes = [ci, ci, ci, ci, ci, ci, C.c_double, C.POINTER(EstBits), i32p, i32p, ci,
                                        C.POINTER(C.c_uint32)]
    R.ref_xRateDistOptQuant.restype = None
    assert R.ref_sizeof_estbits() == C.sizeof(EstBits)
    coef = np.ascontiguousarray(coef, np.int32).reshape(-1).copy()
    lvl = np.zeros(N * N, np.int32)
    s = C.c_uint32(0)
    R.ref_xRateDistOptQuant(qpy, slice_type, ttype, is_intra, dir_mode, tr_idx, lam, C.byref(est), coef, lvl, N, C.byref(s))
    return lvl.reshape(N, N), s.value

"""ctypes binding of libhmx's C-ABI (include/hmx.h) for the Python test and bench harness.

This is plumbing, not product: the product is thevc_amd/libhmx.so (HIP, gfx950) behind include/hmx.h.
There is NO fallback: if the library is missing or a call fails, this module raises."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HMX_LIB_PATH", os.path.join(HERE, "libhmx.so"))  # the override is for A/B runs of a variant build

REG_DCT = 65535
TEXT_LUMA, TEXT_CHROMA, TEXT_CHROMA_U, TEXT_CHROMA_V = 0, 1, 2, 3
B_SLICE, P_SLICE, I_SLICE = 0, 1, 2
TU_TRANSFORM_SKIP, TU_INTER = 1, 2

TU_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("log2n", "u1"), ("plane", "u1"), ("mode", "u1"),
                     ("flags", "u1")])
PU_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("w", "u1"), ("h", "u1"), ("ref0", "u1"), ("ref1", "u1"),
                     ("mv0x", "<i2"), ("mv0y", "<i2"), ("mv1x", "<i2"), ("mv1y", "<i2")])


class HmxError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("bit_depth", C.c_int), ("device", C.c_int), ("stream", C.c_void_p), ("ctu_size", C.c_int)]


class Qp(C.Structure):
    _fields_ = [("qp", C.c_int), ("per", C.c_int), ("rem", C.c_int), ("bits", C.c_int)]


class QuantParam(C.Structure):
    _fields_ = [("qp", Qp), ("per_base", C.c_int), ("slice_type", C.c_int), ("sign_hide", C.c_int),
                ("is_intra", C.c_int), ("dir_mode", C.c_int)]


class RdoqParam(C.Structure):  # hmx_rdoq_param
    _fields_ = [("qp", Qp), ("sign_hide", C.c_int), ("is_intra", C.c_int), ("dir_mode", C.c_int), ("root_cbf", C.c_int),
                ("cbf_ctx", C.c_int), ("lam", C.c_double)]


class PicParam(C.Structure):
    _fields_ = [("pic_w", C.c_int), ("pic_h", C.c_int), ("qp", C.c_int), ("chroma_qp_offset", C.c_int),
                ("slice_type", C.c_int), ("sign_hide", C.c_int)]


class Pic(C.Structure):
    _fields_ = [("plane", C.c_void_p * 3), ("stride", C.c_int * 3)]


class Levels(C.Structure):
    _fields_ = [("plane", C.c_void_p * 3), ("stride", C.c_int * 3)]


class Sse(C.Structure):  # hmx_sse
    _fields_ = [("plane", C.c_void_p * 3)]


class EstBits(C.Structure):  # hmx_est_bits == estBitsSbacStruct (TComTrQuant.h:59-72)
    _fields_ = [("significantCoeffGroupBits", (C.c_int32 * 2) * 2), ("significantBits", (C.c_int32 * 2) * 42),
                ("lastXBits", C.c_int32 * 32), ("lastYBits", C.c_int32 * 32), ("greaterOneBits", (C.c_int32 * 2) * 24),
                ("levelAbsBits", (C.c_int32 * 2) * 6), ("blockCbpBits", (C.c_int32 * 2) * 15),
                ("blockRootCbpBits", (C.c_int32 * 2) * 4), ("scanZigzag", C.c_int32 * 2), ("scanNonZigzag", C.c_int32 * 2)]


class RdoqPic(C.Structure):  # hmx_rdoq_pic
    _fields_ = [("est", EstBits * 8), ("lambda_luma", C.c_double), ("lambda_chroma", C.c_double)]


class RdoqSide(C.Structure):  # hmx_rdoq_side
    _fields_ = [("est_idx", C.c_uint16), ("root_cbf", C.c_uint8), ("cbf_ctx", C.c_uint8)]


class McJob(C.Structure):  # hmx_mc_job
    _fields_ = [("d_pus", C.c_void_p), ("n_pus", C.c_int), ("refs", C.POINTER(Pic)), ("n_refs", C.c_int), ("dst", C.POINTER(Pic)),
                ("pic_w", C.c_int), ("pic_h", C.c_int)]


_lib = None


def lib():
    """Load libhmx.so; fails loudly when the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HmxError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        try:
            # PyTorch ships its own libamdhip64; if libhmx pulls in the system HIP runtime first, a later
            # torch.cuda initialisation in the same process finds no device.  Load torch's runtime first.
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, ci, cu = C.c_void_p, C.c_int, C.c_uint
        L.hmx_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
        L.hmx_destroy.argtypes = [vp]
        L.hmx_destroy.restype = None
        L.hmx_last_error.argtypes = [vp]
        L.hmx_last_error.restype = C.c_char_p
        L.hmx_sync.argtypes = [vp]
        L.hmx_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
        L.hmx_malloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
        L.hmx_free.argtypes = [vp, vp]
        L.hmx_upload.argtypes = [vp, vp, vp, C.c_size_t]
        L.hmx_download.argtypes = [vp, vp, vp, C.c_size_t]
        L.hmx_memset.argtypes = [vp, vp, ci, C.c_size_t]
        L.hmx_event_create.argtypes = [vp, C.POINTER(vp)]
        L.hmx_event_record.argtypes = [vp, vp]
        L.hmx_event_elapsed_ms.argtypes = [vp, vp, vp, C.POINTER(C.c_float)]
        L.hmx_event_destroy.argtypes = [vp, vp]
        L.hmx_setQPforQuant.argtypes = [ci, ci, ci, ci]
        L.hmx_setQPforQuant.restype = Qp
        L.hmx_xT.argtypes = [vp, cu, vp, cu, vp, ci, ci]
        L.hmx_xIT.argtypes = [vp, cu, vp, vp, cu, ci, ci]
        L.hmx_xTransformSkip.argtypes = [vp, vp, cu, vp, ci, ci]
        L.hmx_xITransformSkip.argtypes = [vp, vp, vp, cu, ci, ci]
        L.hmx_xQuant.argtypes = [vp, vp, vp, ci, ci, C.POINTER(C.c_uint32), ci, C.POINTER(QuantParam)]
        L.hmx_arlCoeff.argtypes = [vp, vp, vp, ci, ci, ci, C.POINTER(QuantParam), ci, vp]
        L.hmx_xQuant_scaled.argtypes = [vp, vp, vp, ci, ci, C.POINTER(C.c_uint32), ci, C.POINTER(QuantParam), vp]
        L.hmx_xRateDistOptQuant_scaled.argtypes = [vp, vp, vp, ci, ci, C.POINTER(C.c_uint32), ci, C.POINTER(RdoqParam), C.POINTER(EstBits), vp, vp]
        L.hmx_xDeQuant_scaled.argtypes = [vp, vp, vp, ci, ci, C.POINTER(Qp), vp]
        L.hmx_xDeQuant.argtypes = [vp, vp, vp, ci, ci, C.POINTER(Qp)]
        L.hmx_transformNxN.argtypes = [vp, vp, cu, vp, cu, cu, C.POINTER(C.c_uint32), ci, C.POINTER(QuantParam),
                                       ci, ci]
        L.hmx_invtransformNxN.argtypes = [vp, ci, ci, cu, vp, cu, vp, cu, cu, C.POINTER(Qp), ci]
        L.hmx_initAdiPattern.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, ci, vp]
        L.hmx_predIntraLumaAng.argtypes = [vp, vp, cu, vp, cu, ci, ci]
        L.hmx_predIntraChromaAng.argtypes = [vp, vp, cu, vp, cu, ci, ci]
        L.hmx_calcHAD.argtypes = [vp, vp, ci, vp, ci, ci, ci, C.POINTER(C.c_uint32)]
        L.hmx_getSSE.argtypes = [vp, vp, ci, vp, ci, ci, ci, C.POINTER(C.c_uint32)]
        L.hmx_batch_predIntra_cost.argtypes = [vp, vp, C.POINTER(Pic), C.POINTER(Pic), C.POINTER(PicParam), vp, ci, vp]
        L.hmx_predIntraGetPredValDC.argtypes = [vp, vp, ci, ci, ci, ci, C.POINTER(C.c_int16)]
        L.hmx_xPredIntraPlanar.argtypes = [vp, vp, vp, cu, ci, ci]
        L.hmx_xPredIntraAng.argtypes = [vp, vp, vp, cu, ci, ci, cu, ci, ci, ci]
        for n in ("hmx_filterHorLuma", "hmx_filterHorChroma"):
            getattr(L, n).argtypes = [vp, vp, ci, vp, ci, ci, ci, ci, ci]
        for n in ("hmx_filterVerLuma", "hmx_filterVerChroma"):
            getattr(L, n).argtypes = [vp, vp, ci, vp, ci, ci, ci, ci, ci, ci]
        L.hmx_addAvg.argtypes = [vp, vp, ci, vp, ci, vp, ci, ci, ci]
        L.hmx_xPredInterLumaBlk.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp, ci, ci]
        L.hmx_xPredInterChromaBlk.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp, ci, ci]
        L.hmx_motionCompensation.argtypes = [vp, C.POINTER(Pic), C.POINTER(ci), C.POINTER(Pic), C.POINTER(ci), ci, ci, ci, ci, C.POINTER(Pic)]
        L.hmx_tu_list_create.argtypes = [vp, vp, ci, C.POINTER(vp)]
        L.hmx_tu_list_destroy.argtypes = [vp, vp]
        L.hmx_tu_list_destroy.restype = None
        L.hmx_batch_transformNxN.argtypes = [vp, vp, C.POINTER(Pic), C.POINTER(Levels), vp, C.POINTER(PicParam)]
        L.hmx_batch_residual_transformNxN.argtypes = [vp, vp, C.POINTER(Pic), C.POINTER(Pic), C.POINTER(Levels), vp,
                                                      C.POINTER(PicParam)]
        L.hmx_batch_invtransformNxN.argtypes = [vp, vp, C.POINTER(Levels), C.POINTER(Pic), C.POINTER(Pic),
                                                C.POINTER(PicParam)]
        L.hmx_batch_predIntra.argtypes = [vp, vp, C.POINTER(Pic), C.POINTER(Pic), C.POINTER(PicParam), vp, ci,
                                          C.POINTER(C.c_size_t * 3)]
        L.hmx_intra_plan_create.argtypes = [vp, vp, ci, C.POINTER(PicParam), C.POINTER(vp)]
        L.hmx_intra_dependency_mask.argtypes = [ci, ci, ci, C.c_uint64]
        L.hmx_intra_dependency_mask.restype = C.c_uint64
        L.hmx_intra_avail_mask.argtypes = [ci, ci, ci, ci, ci, ci]
        L.hmx_intra_avail_mask.restype = C.c_uint64
        L.hmx_intra_plan_create_multi.argtypes = [vp, C.POINTER(vp), C.POINTER(ci), ci, C.POINTER(PicParam), C.POINTER(vp)]
        L.hmx_intra_plan_create_device.argtypes = [vp, vp, C.POINTER(C.c_uint32), ci, C.POINTER(PicParam), C.POINTER(vp)]
        L.hmx_intra_plan_download.argtypes = [vp, vp, vp, vp]
        L.hmx_intra_plan_destroy_many.argtypes = [vp, C.POINTER(vp), ci]
        L.hmx_intra_plan_destroy_many.restype = None
        L.hmx_last_call_tables_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.hmx_intra_plan_destroy.argtypes = [vp, vp]
        L.hmx_intra_plan_destroy.restype = None
        L.hmx_intra_plan_info.argtypes = [vp, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)]
        L.hmx_intra_plan_level.argtypes = [vp, ci, C.POINTER(C.c_uint32 * 4), C.POINTER(C.c_uint32)]
        L.hmx_intra_schedule_for.argtypes = [vp, ci]
        L.hmx_last_call_shape.argtypes = [vp, C.POINTER(ci), C.POINTER(ci)]
        L.hmx_set_timing.argtypes = [vp, ci]
        L.hmx_last_call_timing.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.hmx_frame_intra_encode.argtypes = [vp, vp, ci, C.POINTER(Pic), C.POINTER(Pic), C.POINTER(Levels)]
        L.hmx_frame_intra_decode.argtypes = [vp, vp, ci, C.POINTER(Pic), C.POINTER(Levels)]
        L.hmx_frame_intra_encode_multi.argtypes = [vp, vp, ci, C.POINTER(Pic), C.POINTER(Pic), C.POINTER(Levels)]
        L.hmx_frame_intra_decode_multi.argtypes = [vp, vp, ci, C.POINTER(Pic), C.POINTER(Levels)]
        L.hmx_frame_intra_decode_onto.argtypes = [vp, vp, ci, C.POINTER(Pic), C.POINTER(Levels)]
        L.hmx_frame_intra_encode_onto.argtypes = [vp, vp, ci, C.POINTER(Pic), C.POINTER(Pic), C.POINTER(Levels)]
        L.hmx_batch_motionCompensation.argtypes = [vp, vp, ci, C.POINTER(Pic), ci, C.POINTER(Pic)]
        L.hmx_batch_subpel_cost.argtypes = [vp, vp, ci, C.POINTER(Pic), ci, C.POINTER(Pic), vp, ci, ci, vp]
        L.hmx_pic_extend_border.argtypes = [vp, C.POINTER(Pic), ci, ci, ci, ci]
        L.hmx_batch_motionCompensation_multi.argtypes = [vp, ci, C.POINTER(McJob)]
        L.hmx_xRateDistOptQuant.argtypes = [vp, vp, vp, ci, ci, C.POINTER(C.c_uint32), ci, C.POINTER(RdoqParam), C.POINTER(EstBits)]
        L.hmx_batch_xRateDistOptQuant.argtypes = [vp, vp, C.POINTER(RdoqSide), ci, C.POINTER(Levels), C.POINTER(Levels), vp,
                                                  C.POINTER(PicParam), C.POINTER(EstBits), ci, C.c_double, C.c_double]
        L.hmx_batch_residual_transformNxN_multi.argtypes = [vp, vp, ci, C.POINTER(Pic), C.POINTER(Pic), C.POINTER(Levels), vp,
                                                            C.POINTER(PicParam)]
        L.hmx_batch_residual_transform_recon_multi.argtypes = [vp, vp, ci, C.POINTER(Pic), C.POINTER(Pic), C.POINTER(Levels),
                                                               C.POINTER(Pic), vp, C.POINTER(PicParam)]
        L.hmx_batch_residual_transform_recon_sse_multi.argtypes = [vp, vp, ci, C.POINTER(Pic), C.POINTER(Pic), C.POINTER(Levels),
                                                                   C.POINTER(Pic), vp, vp, C.POINTER(PicParam)]
        L.hmx_batch_invtransformNxN_multi.argtypes = [vp, vp, ci, C.POINTER(Levels), C.POINTER(Pic), C.POINTER(Pic),
                                                      C.POINTER(PicParam)]
        L.hmx_pic_extend_border_multi.argtypes = [vp, ci, C.POINTER(Pic), ci, ci, ci, ci]
        L.hmx_sao_picture.argtypes = [vp, C.POINTER(Pic), C.POINTER(Pic), ci, ci, vp, ci]
        L.hmx_deblock_strengths.argtypes = [vp, vp, vp, vp, ci, ci, ci, vp, vp]
        L.hmx_deblock_picture.argtypes = [vp, C.POINTER(Pic), ci, ci, vp, vp, vp, vp, ci, ci]
        L.hmx_yuv_frame_bytes.argtypes = [ci, ci, ci]
        L.hmx_yuv_frame_bytes.restype = C.c_size_t
        L.hmx_yuv_unpack.argtypes = [vp, vp, ci, C.POINTER(Pic), ci, ci, ci, ci]
        L.hmx_yuv_pack.argtypes = [vp, C.POINTER(Pic), ci, ci, ci, ci, ci, vp]
        L.hmx_set_sse_output.argtypes = [vp, vp, ci]
        L.hmx_set_rdoq.argtypes = [vp, C.POINTER(RdoqPic), ci]
        L.hmx_tpool_create.argtypes = [vp, ci, ci, ci, C.POINTER(vp)]
        L.hmx_tpool_destroy.argtypes = [vp, vp]
        L.hmx_tpool_destroy.restype = None
        L.hmx_tpool_import.argtypes = [vp, vp, ci, ci, C.POINTER(Pic)]
        L.hmx_tpool_export.argtypes = [vp, vp, ci, ci, C.POINTER(Pic)]
        L.hmx_yuv_unpack_resident.argtypes = [vp, vp, ci, vp, ci, ci, ci]
        L.hmx_yuv_pack_resident.argtypes = [vp, vp, ci, ci, ci, ci, vp]
        L.hmx_frame_intra_encode_resident.argtypes = [vp, vp, ci, ci, vp, vp, C.POINTER(Levels)]
        L.hmx_frame_intra_decode_resident.argtypes = [vp, vp, ci, ci, vp, C.POINTER(Levels)]
        L.hmx_clipMv.argtypes = [C.POINTER(ci), C.POINTER(ci), ci, ci, ci, ci, ci]
        L.hmx_clipMv.restype = None
        _lib = L
    return _lib


def _hp(a):
    return C.c_void_p(a.ctypes.data)


class DevBuf:
    """A device allocation owned by a Context."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, nbytes
        p = C.c_void_p()
        ctx._chk(lib().hmx_malloc(ctx.h, max(nbytes, 4), C.byref(p)))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.ctx._chk(lib().hmx_upload(self.ctx.h, self.ptr, _hp(arr), arr.nbytes))
        return self

    def download(self, dtype, count=None):
        dtype = np.dtype(dtype)
        count = self.nbytes // dtype.itemsize if count is None else count
        out = np.empty(count, dtype)
        self.ctx._chk(lib().hmx_download(self.ctx.h, _hp(out), self.ptr, out.nbytes))
        return out

    def zero(self):
        self.ctx._chk(lib().hmx_memset(self.ctx.h, self.ptr, 0, self.nbytes))
        return self

    def free(self):
        if self.ptr:
            lib().hmx_free(self.ctx.h, self.ptr)
            self.ptr = None


class Context:
    """hmx_ctx wrapper.  One per GPU/process."""

    def __init__(self, bit_depth=8, device=0, stream=None, ctu_size=64):
        cfg = Config(bit_depth, device, stream, ctu_size)
        h = C.c_void_p()
        rc = lib().hmx_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise HmxError(f"hmx_create failed ({rc}): no usable HIP device?")
        self.h = h
        self.bit_depth = bit_depth

    def _chk(self, rc):
        if rc != 0:
            raise HmxError(f"libhmx error {rc}: {lib().hmx_last_error(self.h).decode()}")

    def close(self):
        if self.h:
            lib().hmx_destroy(self.h)
            self.h = None

    def sync(self):
        self._chk(lib().hmx_sync(self.h))

    def set_option(self, name, value):
        """A tuning knob of include/hmx.h (hmx_set_option); value None restores the default."""
        self._chk(lib().hmx_set_option(self.h, name.encode(), None if value is None else str(value).encode()))

    def set_rdoq(self, pics):
        """hmx_set_rdoq: pics = [(eight EstBits [luma, chroma][4 sizes], lambda_luma, lambda_chroma), ...] for the pictures
        of the next whole-picture encode calls (one entry = the same for all); None turns RDOQ off."""
        if not pics:
            self._chk(lib().hmx_set_rdoq(self.h, None, 0))
            return
        arr = (RdoqPic * len(pics))()
        for i, (ests, ll, lc) in enumerate(pics):
            for k in range(8):
                C.memmove(C.byref(arr[i].est[k]), C.byref(ests[k]), C.sizeof(EstBits))
            arr[i].lambda_luma, arr[i].lambda_chroma = ll, lc
        self._chk(lib().hmx_set_rdoq(self.h, arr, len(pics)))

    def alloc(self, nbytes):
        return DevBuf(self, nbytes)

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        return DevBuf(self, arr.nbytes).upload(arr)

    # --- timing on the context's stream ---
    def event(self):
        e = C.c_void_p()
        self._chk(lib().hmx_event_create(self.h, C.byref(e)))
        return e

    def record(self, e):
        self._chk(lib().hmx_event_record(self.h, e))

    def elapsed_ms(self, a, b):
        ms = C.c_float()
        self._chk(lib().hmx_event_elapsed_ms(self.h, a, b, C.byref(ms)))
        return ms.value

    # --- scalar drop-ins (host numpy arrays) ---
    def xT(self, mode, resi, stride, n):
        resi = np.ascontiguousarray(resi, np.int16)
        coef = np.zeros(n * n, np.int32)
        self._chk(lib().hmx_xT(self.h, mode, _hp(resi), stride, _hp(coef), n, n))
        return coef

    def xIT(self, mode, coef, stride, n):
        coef = np.ascontiguousarray(coef, np.int32)
        resi = np.zeros(n * stride, np.int16)
        self._chk(lib().hmx_xIT(self.h, mode, _hp(coef), _hp(resi), stride, n, n))
        return resi

    def xTransformSkip(self, resi, stride, n):
        resi = np.ascontiguousarray(resi, np.int16)
        coef = np.zeros(n * n, np.int32)
        self._chk(lib().hmx_xTransformSkip(self.h, _hp(resi), stride, _hp(coef), n, n))
        return coef

    def xITransformSkip(self, coef, stride, n):
        coef = np.ascontiguousarray(coef, np.int32)
        resi = np.zeros(n * stride, np.int16)
        self._chk(lib().hmx_xITransformSkip(self.h, _hp(coef), _hp(resi), stride, n, n))
        return resi

    def xQuant(self, src, n, text_type, qparam, ac_sum=0):
        src = np.ascontiguousarray(src, np.int32)
        dst = np.zeros(n * n, np.int32)
        s = C.c_uint32(ac_sum)
        self._chk(lib().hmx_xQuant(self.h, _hp(src), _hp(dst), n, n, C.byref(s), text_type, C.byref(qparam)))
        return dst, s.value

    def xDeQuant_scaled(self, src, n, qp, table):
        src, table = np.ascontiguousarray(src, np.int32), np.ascontiguousarray(table, np.int32)
        dst = np.zeros(n * n, np.int32)
        self._chk(lib().hmx_xDeQuant_scaled(self.h, _hp(src), _hp(dst), n, n, C.byref(qp), _hp(table)))
        return dst

    def arlCoeff(self, src, n, text_type, qparam, rdoq_form, qtab=None):
        src = np.ascontiguousarray(src, np.int32)
        qtab = None if qtab is None else np.ascontiguousarray(qtab, np.int32)
        arl = np.zeros(n * n, np.int32)
        self._chk(lib().hmx_arlCoeff(self.h, _hp(src), _hp(arl), n, n, text_type, C.byref(qparam), int(rdoq_form), None if qtab is None else _hp(qtab)))
        return arl

    def xQuant_scaled(self, src, n, text_type, qparam, qtab, ac_sum=0):
        src, qtab = np.ascontiguousarray(src, np.int32), np.ascontiguousarray(qtab, np.int32)
        dst = np.zeros(n * n, np.int32)
        s = C.c_uint32(ac_sum)
        self._chk(lib().hmx_xQuant_scaled(self.h, _hp(src), _hp(dst), n, n, C.byref(s), text_type, C.byref(qparam), _hp(qtab)))
        return dst, s.value

    def xRateDistOptQuant_scaled(self, src, n, text_type, rparam, est, qtab, estab, abs_sum=0):
        src, qtab, estab = np.ascontiguousarray(src, np.int32), np.ascontiguousarray(qtab, np.int32), np.ascontiguousarray(estab, np.float64)
        dst = np.zeros(n * n, np.int32)
        s = C.c_uint32(abs_sum)
        self._chk(lib().hmx_xRateDistOptQuant_scaled(self.h, _hp(src), _hp(dst), n, n, C.byref(s), text_type, C.byref(rparam), C.byref(est), _hp(qtab), _hp(estab)))
        return dst, s.value

    def xRateDistOptQuant(self, src, n, text_type, rparam, est, abs_sum=0):
        src = np.ascontiguousarray(src, np.int32)
        dst = np.zeros(n * n, np.int32)
        s = C.c_uint32(abs_sum)
        self._chk(lib().hmx_xRateDistOptQuant(self.h, _hp(src), _hp(dst), n, n, C.byref(s), text_type, C.byref(rparam), C.byref(est)))
        return dst, s.value

    def xDeQuant(self, src, n, qp):
        src = np.ascontiguousarray(src, np.int32)
        dst = np.zeros(n * n, np.int32)
        self._chk(lib().hmx_xDeQuant(self.h, _hp(src), _hp(dst), n, n, C.byref(qp)))
        return dst

    def transformNxN(self, resi, stride, n, text_type, qparam, ts=0, bypass=0):
        resi = np.ascontiguousarray(resi, np.int16)
        lvl = np.zeros(n * n, np.int32)
        s = C.c_uint32(0)
        self._chk(lib().hmx_transformNxN(self.h, _hp(resi), stride, _hp(lvl), n, n, C.byref(s), text_type,
                                         C.byref(qparam), ts, bypass))
        return lvl, s.value

    def invtransformNxN(self, lvl, stride, n, text_type, mode, qp, ts=0, bypass=0):
        lvl = np.ascontiguousarray(lvl, np.int32)
        resi = np.zeros(n * stride, np.int16)
        self._chk(lib().hmx_invtransformNxN(self.h, bypass, text_type, mode, _hp(resi), stride, _hp(lvl), n, n,
                                            C.byref(qp), ts))
        return resi

    def initAdiPattern(self, rec, stride, x, y, n, is_chroma, pic_w, pic_h):
        rec = np.ascontiguousarray(rec, np.int16)
        W = 2 * n + 1
        adi = np.zeros(2 * W * W, np.int32)
        self._chk(lib().hmx_initAdiPattern(self.h, _hp(rec), stride, x, y, n, is_chroma, pic_w, pic_h, _hp(adi)))
        return adi

    def predIntraLumaAng(self, adi, mode, stride, n):
        adi = np.ascontiguousarray(adi, np.int32)
        pred = np.zeros(n * stride, np.int16)
        self._chk(lib().hmx_predIntraLumaAng(self.h, _hp(adi), mode, _hp(pred), stride, n, n))
        return pred

    def predIntraChromaAng(self, adi, mode, stride, n):
        adi = np.ascontiguousarray(adi, np.int32)
        pred = np.zeros(n * stride, np.int16)
        self._chk(lib().hmx_predIntraChromaAng(self.h, _hp(adi), mode, _hp(pred), stride, n, n))
        return pred

    def predIntraGetPredValDC(self, adi, n, above, left):
        adi = np.ascontiguousarray(adi, np.int32)
        dc = C.c_int16(0)
        self._chk(lib().hmx_predIntraGetPredValDC(self.h, _hp(adi), n, n, above, left, C.byref(dc)))
        return dc.value

    def xPredIntraPlanar(self, adi, n):
        adi = np.ascontiguousarray(adi, np.int32)
        pred = np.zeros(n * n, np.int16)
        self._chk(lib().hmx_xPredIntraPlanar(self.h, _hp(adi), _hp(pred), n, n, n))
        return pred

    def xPredIntraAng(self, adi, n, mode, above=1, left=1, filt=0):
        adi = np.ascontiguousarray(adi, np.int32)
        pred = np.zeros(n * n, np.int16)
        self._chk(lib().hmx_xPredIntraAng(self.h, _hp(adi), _hp(pred), n, n, n, mode, above, left, filt))
        return pred

    def filter(self, name, src, src_off, ss, ds, w, h, frac, is_first=None, is_last=1):
        """name in filterHorLuma/filterVerLuma/filterHorChroma/filterVerChroma; src is a padded host plane."""
        src = np.ascontiguousarray(src, np.int16)
        dst = np.zeros(h * ds, np.int16)
        sp = C.c_void_p(src.ctypes.data + 2 * src_off)
        f = getattr(lib(), "hmx_" + name)
        if "Hor" in name:
            self._chk(f(self.h, sp, ss, _hp(dst), ds, w, h, frac, is_last))
        else:
            self._chk(f(self.h, sp, ss, _hp(dst), ds, w, h, frac, is_first, is_last))
        return dst

    def addAvg(self, a, b, w, h):
        a = np.ascontiguousarray(a, np.int16)
        b = np.ascontiguousarray(b, np.int16)
        d = np.zeros(w * h, np.int16)
        self._chk(lib().hmx_addAvg(self.h, _hp(a), w, _hp(b), w, _hp(d), w, w, h))
        return d

    # --- batched device path ---
    def tu_list(self, tus):
        tus = np.ascontiguousarray(tus, TU_DTYPE)
        h = C.c_void_p()
        self._chk(lib().hmx_tu_list_create(self.h, _hp(tus), len(tus), C.byref(h)))
        return h

    def intra_plan(self, tus, pp):
        tus = np.ascontiguousarray(tus, TU_DTYPE)
        h = C.c_void_p()
        self._chk(lib().hmx_intra_plan_create(self.h, _hp(tus), len(tus), C.byref(pp), C.byref(h)))
        return h

    def intra_plans(self, tus_list, pp):
        """hmx_intra_plan_create_multi: the plans of several pictures, their host-side analysis on all host threads."""
        arrs = [np.ascontiguousarray(t, TU_DTYPE) for t in tus_list]
        n = len(arrs)
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        cnts = (C.c_int * n)(*[len(a) for a in arrs])
        out = (C.c_void_p * n)()
        self._chk(lib().hmx_intra_plan_create_multi(self.h, ptrs, cnts, n, C.byref(pp), out))
        return [C.c_void_p(out[i]) for i in range(n)]


    def intra_plans_device(self, d_tus, offsets, pp):
        """hmx_intra_plan_create_device: d_tus = device address of the pictures' decision lists back to back (hmx_tu, coding
        order), offsets = n_pics + 1 block offsets.  Returns the plan handles."""
        n = len(offsets) - 1
        off = (C.c_uint32 * (n + 1))(*[int(o) for o in offsets])
        out = (C.c_void_p * n)()
        self._chk(lib().hmx_intra_plan_create_device(self.h, d_tus, off, n, C.byref(pp), out))
        return [C.c_void_p(out[i]) for i in range(n)]


    def plan_tables(self, plan):
        """hmx_intra_plan_download: (blocks, levels) of a plan as numpy arrays -- blocks: the sorted block list (hmx_tu fields +
        `avail`), levels: per dependency level start[4], count[4]."""
        nb, nl, nd = C.c_int(), C.c_int(), C.c_int()
        lib().hmx_intra_plan_info(plan, C.byref(nb), C.byref(nl), C.byref(nd))
        blocks = np.zeros(nb.value, np.dtype(TU_DTYPE.descr + [("avail", "<u8")]))
        levels = np.zeros((nl.value, 8), np.uint32)
        self._chk(lib().hmx_intra_plan_download(self.h, plan, _hp(blocks), _hp(levels)))
        return blocks, levels


def qp_for(qpy, text_type, bit_depth, chroma_qp_offset=0):
    return lib().hmx_setQPforQuant(qpy, text_type, 6 * (bit_depth - 8), chroma_qp_offset)


class DevPicture:
    """Three Pel planes in HBM with the reference's margin layout (TComPicYuv.cpp:82-94):
    luma margin mx,my; chroma half of it.  Also usable without margins (mx = my = 0)."""

    def __init__(self, ctx, w, h, mx=0, my=0, dtype=np.int16, pad=0, skew=0):
        """pad: extra samples per row (an odd stride with pad=1); skew: samples the planes start after their
        allocation (skew=1: plane addresses that are not dword-aligned) -- for tests of the alignment paths."""
        self.ctx, self.w, self.h, self.mx, self.my = ctx, w, h, mx, my
        self.pad, self.skew = pad, skew
        self.dtype = np.dtype(dtype)
        self.dims = [(w, h, mx, my), (w // 2, h // 2, mx // 2, my // 2), (w // 2, h // 2, mx // 2, my // 2)]
        self.strides = [pw + 2 * pmx + pad for (pw, ph, pmx, pmy) in self.dims]
        self.elems = [(pw + 2 * pmx + pad) * (ph + 2 * pmy) + skew for (pw, ph, pmx, pmy) in self.dims]
        self.bufs = [ctx.alloc(e * self.dtype.itemsize) for e in self.elems]

    def origin_ptr(self, p):
        pw, ph, pmx, pmy = self.dims[p]
        return self.bufs[p].ptr + (self.skew + pmy * self.strides[p] + pmx) * self.dtype.itemsize

    def as_pic(self):
        s = Pic() if self.dtype == np.int16 else Levels()
        for p in range(3):
            s.plane[p] = self.origin_ptr(p)
            s.stride[p] = self.strides[p]
        return s

    def upload(self, planes):
        """planes: three 2-D arrays (h x w) without margins"""
        for p in range(3):
            pw, ph, pmx, pmy = self.dims[p]
            flat = np.zeros(self.elems[p], self.dtype)
            full = flat[self.skew:].reshape(ph + 2 * pmy, self.strides[p])
            full[pmy:pmy + ph, pmx:pmx + pw] = np.asarray(planes[p]).reshape(ph, pw)
            self.bufs[p].upload(flat)
        return self

    def download(self, with_margins=False):
        out = []
        for p in range(3):
            pw, ph, pmx, pmy = self.dims[p]
            flat = self.bufs[p].download(self.dtype)
            full = flat[self.skew:].reshape(ph + 2 * pmy, self.strides[p])[:, :pw + 2 * pmx]
            out.append(full.copy() if with_margins else full[pmy:pmy + ph, pmx:pmx + pw].copy())
        return out

    def zero(self):
        for b in self.bufs:
            b.zero()
        return self

    def free(self):
        for b in self.bufs:
            b.free()


def _spread4(t):
    return (t & 1) | ((t & 2) << 1) | ((t & 4) << 2) | ((t & 8) << 3)


class DevLevelsZ:
    """Quantised levels in the reference's own coefficient layout (stride 0 in hmx_levels): per plane,
    CTU blocks in raster order (C*C ints, C = 64 luma / 32 chroma); inside a CTU the N x N block of a
    transform block whose origin is 4x4 unit (ux, uy) sits at 16 * Zorder(ux, uy), row-major
    (TComDataCU::m_pcTrCoeffY + 16 * partition index)."""

    def __init__(self, ctx, w, h, ctu=64):
        self.ctx, self.w, self.h, self.ctu = ctx, w, h, ctu
        self.cw, self.ch = -(-w // ctu), -(-h // ctu)
        self.elems = [self.cw * self.ch * ctu * ctu, self.cw * self.ch * ctu * ctu // 4, self.cw * self.ch * ctu * ctu // 4]
        self.bufs = [ctx.alloc(4 * e) for e in self.elems]

    def as_pic(self):
        s = Levels()
        for p in range(3):
            s.plane[p] = self.bufs[p].ptr
            s.stride[p] = 0
        return s

    def zero(self):
        for b in self.bufs:
            b.zero()
        return self

    def free(self):
        for b in self.bufs:
            b.free()

    def block_offset(self, plane, x, y):
        c = self.ctu >> (1 if plane else 0)
        m = c - 1
        z = _spread4((x & m) >> 2) | (_spread4((y & m) >> 2) << 1)
        return ((y // c) * self.cw + (x // c)) * c * c + z * 16

    def to_planes(self, tus):
        """Gather into plane geometry (numpy) following the block list."""
        raw = [b.download(np.int32) for b in self.bufs]
        out = [np.zeros((self.h, self.w), np.int32), np.zeros((self.h // 2, self.w // 2), np.int32),
               np.zeros((self.h // 2, self.w // 2), np.int32)]
        for t in tus:
            n, p, x, y = 1 << int(t["log2n"]), int(t["plane"]), int(t["x"]), int(t["y"])
            o = self.block_offset(p, x, y)
            out[p][y:y + n, x:x + n] = raw[p][o:o + n * n].reshape(n, n)
        return out

    def from_planes(self, planes, tus):
        raw = [np.zeros(e, np.int32) for e in self.elems]
        for t in tus:
            n, p, x, y = 1 << int(t["log2n"]), int(t["plane"]), int(t["x"]), int(t["y"])
            o = self.block_offset(p, x, y)
            raw[p][o:o + n * n] = np.asarray(planes[p])[y:y + n, x:x + n].reshape(-1)
        for p in range(3):
            self.bufs[p].upload(raw[p])
        return self


class ResidentPool:
    """hmx_tpool: n pictures of one size resident in the library's working layout (include/hmx.h)."""

    def __init__(self, ctx, w, h, n):
        self.ctx, self.w, self.h, self.n = ctx, w, h, n
        h_ = C.c_void_p()
        ctx._chk(lib().hmx_tpool_create(ctx.h, w, h, n, C.byref(h_)))
        self.h_ = h_

    def import_planes(self, first, dev_pictures):
        n = len(dev_pictures)
        arr = (Pic * n)(*[d.as_pic() for d in dev_pictures])
        self.ctx._chk(lib().hmx_tpool_import(self.ctx.h, self.h_, first, n, arr))
        return self

    def export_planes(self, first, dev_pictures):
        n = len(dev_pictures)
        arr = (Pic * n)(*[d.as_pic() for d in dev_pictures])
        self.ctx._chk(lib().hmx_tpool_export(self.ctx.h, self.h_, first, n, arr))
        return self

    def free(self):
        if self.h_:
            lib().hmx_tpool_destroy(self.ctx.h, self.h_)
            self.h_ = None


class DevLevelsZSlab:
    """The levels of n pictures in the reference's coefficient layout (DevLevelsZ) as ONE allocation per plane, picture i
    at offset i * elems: what a pipeline that owns its level buffers would allocate (and the whole-picture calls then
    address by arithmetic instead of through the picture table)."""

    def __init__(self, ctx, w, h, n, ctu=64):
        self.ctx, self.w, self.h, self.n, self.ctu = ctx, w, h, n, ctu
        self.cw, self.ch = -(-w // ctu), -(-h // ctu)
        self.elems = [self.cw * self.ch * ctu * ctu, self.cw * self.ch * ctu * ctu // 4, self.cw * self.ch * ctu * ctu // 4]
        self.bufs = [ctx.alloc(4 * e * n) for e in self.elems]

    def zero(self):
        for b in self.bufs:
            b.zero()
        return self

    def as_pic(self, i):
        s = Levels()
        for p in range(3):
            s.plane[p] = self.bufs[p].ptr + 4 * self.elems[p] * i
            s.stride[p] = 0
        return s

    def picture(self, i):
        """A DevLevelsZ-like view of picture i (to_planes)."""
        v = DevLevelsZ.__new__(DevLevelsZ)
        v.ctx, v.w, v.h, v.ctu, v.cw, v.ch, v.elems = self.ctx, self.w, self.h, self.ctu, self.cw, self.ch, self.elems
        parent = self

        class _View:
            def __init__(self, p):
                self.p = p

            def download(self, dtype, count=None):
                out = np.empty(parent.elems[self.p], np.int32)
                parent.ctx._chk(lib().hmx_download(parent.ctx.h, _hp(out), parent.bufs[self.p].ptr + 4 * parent.elems[self.p] * i, out.nbytes))
                return out

        v.bufs = [_View(p) for p in range(3)]
        return v

    def free(self):
        for b in self.bufs:
            b.free()

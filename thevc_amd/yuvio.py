"""Planar 4:2:0 YUV files <-> pictures in HBM (the reference's TVideoIOYuv, TLibVideoIO/TVideoIOYuv.cpp).
A frame is read from the file as bytes, uploaded as bytes, and widened / scaled to the internal bit depth / padded
on the device (hmx_yuv_unpack); writing goes the other way (hmx_yuv_pack)."""
import ctypes as C

import numpy as np

from . import capi


class YuvReader:
    """TVideoIOYuv::open(read) + read: frames of `w` x `h` samples at `file_bits` per sample."""

    def __init__(self, ctx, path, w, h, file_bits=8):
        self.ctx, self.w, self.h, self.file_bits = ctx, w, h, file_bits
        self.f = open(path, "rb")
        self.frame_bytes = capi.lib().hmx_yuv_frame_bytes(w, h, file_bits)
        self.staging = ctx.alloc(self.frame_bytes)

    def skip_frames(self, n):  # TVideoIOYuv::skipFrames
        self.f.seek(n * self.frame_bytes, 1)

    def read(self, pic, pad_x=0, pad_y=0):
        """Next frame into DevPicture `pic` of (w + pad_x) x (h + pad_y); False at end of file."""
        raw = self.f.read(self.frame_bytes)
        if len(raw) < self.frame_bytes:
            return False
        self.staging.upload(np.frombuffer(raw, np.uint8))
        p = pic.as_pic()
        self.ctx._chk(capi.lib().hmx_yuv_unpack(self.ctx.h, self.staging.ptr, self.file_bits, C.byref(p), self.w + pad_x,
                                                self.h + pad_y, pad_x, pad_y))
        return True

    def close(self):
        self.f.close()
        self.staging.free()


class YuvWriter:
    """TVideoIOYuv::open(write) + write."""

    def __init__(self, ctx, path, file_bits=8):
        self.ctx, self.file_bits = ctx, file_bits
        self.f = open(path, "wb")
        self.staging = None

    def write(self, pic, w, h, crop_right=0, crop_bottom=0):
        n = capi.lib().hmx_yuv_frame_bytes(w - crop_right, h - crop_bottom, self.file_bits)
        if self.staging is None or self.staging.nbytes < n:
            if self.staging is not None:
                self.staging.free()
            self.staging = self.ctx.alloc(n)
        p = pic.as_pic()
        self.ctx._chk(capi.lib().hmx_yuv_pack(self.ctx.h, C.byref(p), w, h, crop_right, crop_bottom, self.file_bits, self.staging.ptr))
        self.f.write(self.staging.download(np.uint8, n).tobytes())

    def close(self):
        self.f.close()
        if self.staging is not None:
            self.staging.free()

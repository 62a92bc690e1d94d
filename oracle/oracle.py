"""ctypes wrapper around oracle/liboracle.so (the plain-C restatement) and oracle/_ref/jpeg
(the real reference binary, when built).  TEST INFRASTRUCTURE ONLY: importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from libjpeg_amd/."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
REF_BIN = os.path.join(HERE, "_ref", "jpeg")
REF_SRC = os.environ.get("LIBJPEG_REFERENCE", "/root/reference")


class OjInfo(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int), ("precision", C.c_int), ("ncomp", C.c_int),
        ("comp_id", C.c_int * 4), ("hs", C.c_int * 4), ("vs", C.c_int * 4), ("tq", C.c_int * 4),
        ("subx", C.c_int * 4), ("suby", C.c_int * 4), ("hmax", C.c_int), ("vmax", C.c_int),
        ("mcus_x", C.c_int), ("mcus_y", C.c_int), ("bw", C.c_int * 4), ("bh", C.c_int * 4),
        ("cw", C.c_int * 4), ("ch", C.c_int * 4), ("restart_interval", C.c_int),
        ("adobe_transform", C.c_int), ("ycbcr", C.c_int), ("quant", (C.c_uint16 * 64) * 4),
        ("quant_defined", C.c_int * 4),
        ("scan_state_valid", C.c_int), ("cquant", (C.c_uint16 * 64) * 4), ("comp_seen", C.c_int * 4),
        ("ref_error", C.c_int), ("warnings", C.c_int),
        ("dnl", C.c_int), ("rows", C.c_int * 4), ("residual_type", C.c_int), ("transformer_refused", C.c_int),
    ]


def build(ref: bool = True) -> None:
    """Compile liboracle.so and, if the reference sources are present, oracle/_ref/jpeg."""
    subprocess.run(["make", "-s", "-C", HERE], check=True)
    if ref and os.path.isdir(REF_SRC) and not os.path.exists(REF_BIN):
        subprocess.run(["make", "-s", "-j8", "-C", HERE, "ref", f"REF={REF_SRC}"], check=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build(ref=False)
        L = C.CDLL(LIB)
        L.oj_read_info.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(OjInfo)]
        L.oj_decode_coefficients.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(OjInfo), C.POINTER(C.c_void_p)]
        L.oj_decode_coefficients_residual.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(OjInfo), C.POINTER(C.c_void_p)]
        L.oj_idct_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.oj_idct_block.restype = None
        L.oj_idct_plane.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.oj_idct_plane.restype = None
        L.oj_upsample_block.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 7
        L.oj_upsample_block.restype = None
        L.oj_reconstruct.argtypes = [C.POINTER(OjInfo), C.POINTER(C.c_void_p), C.c_void_p, C.c_int]
        L.oj_reconstruct16.argtypes = [C.POINTER(OjInfo), C.POINTER(C.c_void_p), C.c_void_p, C.c_int]
        L.oj_decode_xt.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(OjInfo), C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        L.oj_decode_xt_ex.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(OjInfo), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int]
        L.oj_decode_xt_planes.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(OjInfo), C.POINTER(C.c_void_p)]
        L.oj_decode_xt_planes2.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(OjInfo), C.POINTER(C.c_void_p), C.POINTER(OjInfo), C.POINTER(C.c_void_p)]
        L.oj_decode_alpha.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(OjInfo), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.POINTER(C.c_int), C.POINTER(C.c_uint32)]
        L.oj_free.argtypes = [C.c_void_p]
        L.oj_forward.argtypes = [C.POINTER(OjInfo), C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.oj_fdct_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.oj_free.restype = None
        L.oj_requester_new.restype = C.c_void_p
        L.oj_requester_new.argtypes = [C.POINTER(OjInfo), C.POINTER(C.c_void_p)]
        L.oj_xt_requester_new.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(OjInfo), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.oj_requester_free.argtypes = [C.c_void_p]
        L.oj_requester_free.restype = None
        L.oj_requester_cursor.argtypes = [C.c_void_p, C.c_int]
        L.oj_requester_display.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                                                         C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
        _lib = L
    return _lib


def read_info(data: bytes) -> OjInfo:
    info = OjInfo()
    rc = lib().oj_read_info(data, len(data), C.byref(info))
    if rc:
        raise ValueError(f"oracle: oj_read_info failed rc={rc}")
    return info


def forward(info: OjInfo, pixels: np.ndarray, use_ycbcr: int = 1):
    """Encoder direction: interleaved 8-bit pixels (H, W, ncomp) -> [int32 ndarray (bh, bw, 64)] quantised coefficients for
    the geometry, sampling factors and quantiser tables of `info` (e.g. read_info() of a file the reference wrote)."""
    px = np.ascontiguousarray(pixels, np.uint8)
    planes = [np.zeros((info.bh[c], info.bw[c], 64), np.int32) for c in range(info.ncomp)]
    ptrs = (C.c_void_p * 4)(*[p.ctypes.data for p in planes] + [None] * (4 - info.ncomp))
    rc = lib().oj_forward(C.byref(info), px.ctypes.data, use_ycbcr, ptrs)
    if rc:
        raise ValueError(f"oracle: oj_forward failed rc={rc}")
    return planes


def decode_coefficients(data: bytes, info: OjInfo | None = None):
    """-> (info, [int32 ndarray (bh, bw, 64) per component]) quantised, natural order."""
    info = info or read_info(data)
    planes = [np.zeros((info.bh[c], info.bw[c], 64), np.int32) for c in range(info.ncomp)]
    ptrs = (C.c_void_p * 4)(*[p.ctypes.data for p in planes] + [None] * (4 - info.ncomp))
    rc = lib().oj_decode_coefficients(data, len(data), C.byref(info), ptrs)
    if rc:
        raise ValueError(f"oracle: oj_decode_coefficients failed rc={rc}")
    return info, planes


def xt_box_payload(data: bytes, box_type: bytes) -> bytes | None:
    """The payload of the first JPEG XT box of this type in front of the first scan, its APP11 segments put together (boxes/box.cpp:93-200;
    well-formed framing assumed: for tests that damage what is INSIDE a box)."""
    out, i, found = b"", 2, False
    while i + 4 <= len(data) and data[i] == 0xFF and data[i + 1] != 0xDA:
        ln = (data[i + 2] << 8) | data[i + 3]
        if data[i + 1] == 0xEB and data[i + 4:i + 6] == b"JP" and data[i + 16:i + 20] == box_type:
            out += data[i + 20:i + 2 + ln]
            found = True
        i += 2 + ln
    return out if found else None


def decode_residual_coefficients(data: bytes):
    """JPEG XT: -> (info, planes) of the residual codestream in the file's RESI box, walked as the reference walks it."""
    resi = xt_box_payload(data, b"RESI")
    info = OjInfo()
    L = lib()
    L.oj_read_info_residual.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(OjInfo)]
    rc = L.oj_read_info_residual(resi, len(resi), C.byref(info))
    if rc:
        raise ValueError(f"oracle: oj_read_info_residual failed rc={rc}")
    planes = [np.zeros((info.bh[c], info.bw[c], 64), np.int32) for c in range(info.ncomp)]
    ptrs = (C.c_void_p * 4)(*[p.ctypes.data for p in planes] + [None] * (4 - info.ncomp))
    rc = lib().oj_decode_coefficients_residual(resi, len(resi), C.byref(info), ptrs)
    if rc:
        raise ValueError(f"oracle: oj_decode_coefficients_residual failed rc={rc}")
    return info, planes


def reconstruct(info: OjInfo, planes, use_ycbcr: int = -1) -> np.ndarray:
    """Coefficient planes -> (H, W, ncomp) uint8."""
    planes = [np.ascontiguousarray(p, np.int32) for p in planes]
    ptrs = (C.c_void_p * 4)(*[p.ctypes.data for p in planes] + [None] * (4 - info.ncomp))
    out = np.zeros((info.height, info.width, info.ncomp), np.uint8)
    rc = lib().oj_reconstruct(C.byref(info), ptrs, out.ctypes.data, use_ycbcr)
    if rc:
        raise ValueError(f"oracle: oj_reconstruct failed rc={rc}")
    return out


def reconstruct16(info: OjInfo, planes, use_ycbcr: int = -1) -> np.ndarray:
    """Coefficient planes of a precision 8 or 12 frame -> (H, W, ncomp) uint16."""
    planes = [np.ascontiguousarray(p, np.int32) for p in planes]
    ptrs = (C.c_void_p * 4)(*[p.ctypes.data for p in planes] + [None] * (4 - info.ncomp))
    out = np.zeros((info.height, info.width, info.ncomp), np.uint16)
    rc = lib().oj_reconstruct16(C.byref(info), ptrs, out.ctypes.data, use_ycbcr)
    if rc:
        raise ValueError(f"oracle: oj_reconstruct16 failed rc={rc}")
    return out


def decode16(data: bytes, use_ycbcr: int = -1) -> np.ndarray:
    info, planes = decode_coefficients(data)
    return reconstruct16(info, planes, use_ycbcr)


def decode_xt(data: bytes):
    """JPEG XT profile C -> ((H, W, 3) uint16 codes, is_float).  With is_float the codes are half-float bit patterns."""
    info = OjInfo()
    px = C.c_void_p()
    isf = C.c_int(0)
    rc = lib().oj_decode_xt(data, len(data), C.byref(info), C.byref(px), C.byref(isf))
    if rc:
        raise ValueError(f"oracle: oj_decode_xt failed rc={rc}")
    n = info.width * info.height * info.ncomp
    out = np.ctypeslib.as_array((C.c_uint16 * n).from_address(px.value)).reshape(info.height, info.width, info.ncomp).copy()
    lib().oj_free(px)
    return out, bool(isf.value)


def decode_xt_planes(data: bytes):
    """JPEG XT: -> (info, planes) of the LEGACY frame as the merge sees it: visible scans moved up by the hidden bits, the hidden
    refinement scans of the FINE boxes applied (info.precision includes them)."""
    info = OjInfo()
    ptrs = (C.c_void_p * 4)()
    rc = lib().oj_decode_xt_planes(data, len(data), C.byref(info), ptrs)
    if rc:
        raise ValueError(f"oracle: oj_decode_xt_planes failed rc={rc}")
    planes = []
    for c in range(info.ncomp):
        n = info.bw[c] * info.bh[c] * 64
        planes.append(np.ctypeslib.as_array((C.c_int32 * n).from_address(ptrs[c])).reshape(info.bh[c], info.bw[c], 64).copy())
        lib().oj_free(ptrs[c])
    return info, planes


def decode_xt_residual_planes(data: bytes):
    """JPEG XT: -> (rinfo, planes) of the RESIDUAL frame as the merge sees it (hidden bits of the RFIN boxes included)."""
    info, rinfo = OjInfo(), OjInfo()
    ptrs, rptrs = (C.c_void_p * 4)(), (C.c_void_p * 4)()
    rc = lib().oj_decode_xt_planes2(data, len(data), C.byref(info), ptrs, C.byref(rinfo), rptrs)
    if rc:
        raise ValueError(f"oracle: oj_decode_xt_planes2 failed rc={rc}")
    planes = []
    for c in range(info.ncomp):
        lib().oj_free(ptrs[c])
        n = rinfo.bw[c] * rinfo.bh[c] * 64
        planes.append(np.ctypeslib.as_array((C.c_int32 * n).from_address(rptrs[c])).reshape(rinfo.bh[c], rinfo.bw[c], 64).copy())
        lib().oj_free(rptrs[c])
    return rinfo, planes


def alpha_read_error(data: bytes):
    """What the alpha channel does to JPEG::Read: the reference's code where the alpha image's codestreams (or the form of its boxes)
    stop the read, 0 where they do not -- a refusal of the alpha image's colour transformer is NOT a read error: it arrives with the
    first request for alpha pixels, and a client that never asks (the command line without -al, or any client of a file whose alpha
    merging specification names no compositing method) never sees it.  None: no alpha channel / outside the restatement."""
    info = OjInfo()
    px = C.c_void_p()
    isf, omax, mode = C.c_int(0), C.c_int(0), C.c_int(-1)
    matte = (C.c_uint32 * 3)()
    rc = lib().oj_decode_alpha(data, len(data), C.byref(info), C.byref(px), C.byref(isf), C.byref(omax), C.byref(mode), matte)
    if rc == 0:
        lib().oj_free(px)
        return 0
    if rc == -2 or not info.ref_error:
        return None
    return 0 if info.transformer_refused else info.ref_error


def decode_alpha(data: bytes):
    """The alpha channel of a JPEG XT file -> (codes (H, W) uint16 or None, is_float, out_max, mode, matte (r, g, b), ref_error):
    ref_error 0 with a plane, None where the file has no alpha channel or the restatement does not follow, else the reference's code
    (of the read, or of the first request for alpha pixels: alpha_read_error tells them apart)."""
    info = OjInfo()
    px = C.c_void_p()
    isf, omax, mode = C.c_int(0), C.c_int(0), C.c_int(-1)
    matte = (C.c_uint32 * 3)()
    rc = lib().oj_decode_alpha(data, len(data), C.byref(info), C.byref(px), C.byref(isf), C.byref(omax), C.byref(mode), matte)
    if rc:
        return None, False, 0, mode.value, tuple(matte), (info.ref_error if (rc != -2 and info.ref_error) else None)
    n = info.width * info.height
    out = np.ctypeslib.as_array((C.c_uint16 * n).from_address(px.value)).reshape(info.height, info.width).copy()
    lib().oj_free(px)
    return out, bool(isf.value), omax.value, mode.value, tuple(matte), 0


def decode_xt_status(data: bytes, no_color_transform: bool = False):
    """-> (codes or None, is_float, ref_error): ref_error = 0, the JPGERR_* code the reference fails with where the oracle knows
    it (merging specification errors), or None (outside the oracle's subset / other failure).
    no_color_transform: as `jpeg -c in.jpg out` decodes it (the standard YCbCr L transformation replaced by the identity)."""
    info = OjInfo()
    px = C.c_void_p()
    isf = C.c_int(0)
    rc = lib().oj_decode_xt_ex(data, len(data), C.byref(info), C.byref(px), C.byref(isf), 1 if no_color_transform else 0)
    if rc:
        return None, False, (info.ref_error if (rc != -2 and info.ref_error) else None)
    n = info.width * info.height * info.ncomp
    out = np.ctypeslib.as_array((C.c_uint16 * n).from_address(px.value)).reshape(info.height, info.width, info.ncomp).copy()
    lib().oj_free(px)
    return out, bool(isf.value), 0


def half_codes_to_float(codes: np.ndarray) -> np.ndarray:
    """cmd/iohelpers.hpp:60-77 (HalfToDouble): exact for finite codes, so numpy's float16 view does the same; every code with
    exponent 31 -- the NaN patterns an unclamped merge of a damaged stream can produce included -- is HUGE_VAL with the code's sign."""
    codes = np.ascontiguousarray(codes, np.uint16)
    f = codes.view(np.float16).astype(np.float32)
    top = (codes & 0x7C00) == 0x7C00
    f[top] = np.where(codes[top] & 0x8000, -np.inf, np.inf).astype(np.float32)
    return f


def decode(data: bytes, use_ycbcr: int = -1) -> np.ndarray:
    info, planes = decode_coefficients(data)
    return reconstruct(info, planes, use_ycbcr)


def idct_block(coef: np.ndarray, quant: np.ndarray, precision: int = 8) -> np.ndarray:
    coef = np.ascontiguousarray(coef, np.int32).reshape(64)
    quant = np.ascontiguousarray(quant, np.uint16).reshape(64)
    out = np.zeros(64, np.int32)
    lib().oj_idct_block(out.ctypes.data, coef.ctypes.data, quant.ctypes.data, precision)
    return out.reshape(8, 8)


def idct_plane(coef: np.ndarray, quant: np.ndarray, precision: int = 8) -> np.ndarray:
    bh, bw, _ = coef.shape
    coef = np.ascontiguousarray(coef, np.int32)
    quant = np.ascontiguousarray(quant, np.uint16).reshape(64)
    out = np.zeros((bh * 8, bw * 8), np.int32)
    lib().oj_idct_plane(out.ctypes.data, coef.ctypes.data, bw, bh, quant.ctypes.data, precision)
    return out


def upsample_block(plane: np.ndarray, cw: int, ch: int, sx: int, sy: int, X0: int, Y0: int) -> np.ndarray:
    plane = np.ascontiguousarray(plane, np.int32)
    out = np.zeros(64, np.int32)
    lib().oj_upsample_block(out.ctypes.data, plane.ctypes.data, plane.shape[1], cw, ch, sx, sy, X0, Y0)
    return out.reshape(8, 8)


# ---------------------------------------------------------------------------- real reference
def decode_status(data: bytes, max_bytes: int = 1 << 28):
    """Whole decode the way the reference CLI does it, damaged streams included:
    -> (pixels or None, ref_error, warnings).  ref_error is the JPGERR_* code the reference fails with (0: decoded);
    pixels are (H, W, ncomp) uint8, or uint16 for 12-bit frames.  rc -2 (a coding process outside the path) and frames
    beyond max_bytes of coefficients -> ref_error None."""
    info = OjInfo()
    rc = lib().oj_read_info(data, len(data), C.byref(info))
    if rc == -2:
        return None, None, info.warnings
    if rc:
        return None, info.ref_error, info.warnings
    if sum(info.bh[c] * info.bw[c] for c in range(info.ncomp)) * 256 > max_bytes:
        return None, None, info.warnings  # a damaged frame header announces a giant picture: not worth the memory
    planes = [np.zeros((info.bh[c], info.bw[c], 64), np.int32) for c in range(info.ncomp)]
    ptrs = (C.c_void_p * 4)(*[p.ctypes.data for p in planes] + [None] * (4 - info.ncomp))
    rc = lib().oj_decode_coefficients(data, len(data), C.byref(info), ptrs)
    if rc == -2:
        return None, None, info.warnings
    if rc:
        return None, info.ref_error, info.warnings
    if info.ncomp in (2, 4):
        # the CLI writes such frames component by component (PGX, cmd/reconstruct.cpp:272-303), and the library's state
        # between those calls shows: see oj_requester in jpeg_oracle.c
        _, rcs, canvas = run_requests(data, cli_requests(info.width, info.height, info.ncomp), decoded=(info, planes))
        if any(rcs):
            raise ValueError(f"oracle: request sequence failed {rcs}")
        return np.ascontiguousarray(np.moveaxis(canvas, 0, -1)), 0, info.warnings
    if info.precision == 8:
        out = np.zeros((info.height, info.width, info.ncomp), np.uint8)
        rc = lib().oj_reconstruct(C.byref(info), ptrs, out.ctypes.data, -1)
    else:
        out = np.zeros((info.height, info.width, info.ncomp), np.uint16)
        rc = lib().oj_reconstruct16(C.byref(info), ptrs, out.ctypes.data, -1)
    if rc:
        raise ValueError(f"oracle: reconstruction failed rc={rc}")
    return out, 0, info.warnings


REF_RECT_CALLS = os.path.join(HERE, "_ref", "rect_calls_ref")


def cli_requests(width: int, height: int, ncomp: int, ctrafo: int = 1):
    """The DisplayRectangle calls of the reference's command line (cmd/reconstruct.cpp:272-342, cmd/bitmaphook.cpp:122) with
    upsampling on: one or three components -> stripes of eight lines over all components; otherwise (PGX) the same stripes
    component by component.  Tuples as tests/cxx/rect_calls.cpp reads them."""
    stripes = [(y, min(y + 7, height - 1)) for y in range(0, height, 8)]
    if ncomp in (1, 3):
        return [(0, y0, -1, y1, 0, ncomp - 1, 1, ctrafo, 8) for y0, y1 in stripes]
    return [(0, y0, -1, y1, c, c, 1, ctrafo, 8) for c in range(ncomp) for y0, y1 in stripes]


def run_requests(data: bytes, requests, cursors=None, decoded=None):
    """A sequence of JPEG::DisplayRectangle calls through the oracle's stateful restatement (oj_requester): requests =
    [(minx, miny, maxx, maxy, c0, c1, upsample, ctrafo, hmode)] as tests/cxx/rect_calls.cpp reads them ->
    (info, [return code per call], canvas planes (ncomp, H, W) initialised to 0xAA)."""
    info, planes = decoded if decoded is not None else decode_coefficients(data)
    planes = [np.ascontiguousarray(p, np.int32) for p in planes]
    sb = 2 if info.precision > 8 else 1
    W, H, nc = info.width, info.height, info.ncomp
    canvas = np.full((nc, H, W), 0xAAAA if sb == 2 else 0xAA, np.uint16 if sb == 2 else np.uint8)
    ptrs = (C.c_void_p * 4)(*[p.ctypes.data for p in planes] + [None] * (4 - nc))
    rq = lib().oj_requester_new(C.byref(info), ptrs)
    rcs = []
    try:
        for minx, miny, maxx, maxy, c0, c1, ups, ctrafo, hmode in requests:
            maxx = W - 1 if maxx < 0 else maxx
            maxy = H - 1 if maxy < 0 else maxy
            dst = (C.c_void_p * 4)(*[canvas[c].ctypes.data for c in range(nc)] + [None] * (4 - nc))
            bpp = (C.c_int * 4)(sb, sb, sb, sb)
            bpr = (C.c_int * 4)(W * sb, W * sb, W * sb, W * sb)
            hh = miny + hmode if hmode else H
            bmh = (C.c_int * 4)(hh, hh, hh, hh)
            bmw = (C.c_int * 4)(W, W, W, W)
            rcs.append(lib().oj_requester_display(rq, minx, miny, maxx, maxy, c0, c1, ups, ctrafo, dst, bpp, bpr, bmw, bmh, sb))
            if cursors is not None:  # the row each component's cursor stands at after the call
                cursors.append([lib().oj_requester_cursor(rq, c) for c in range(nc)])
    finally:
        lib().oj_requester_free(rq)
    return info, rcs, canvas


def run_requests_xt(data: bytes, requests, cursors=None):
    """run_requests on a JPEG XT stream (profile C): the residual image's cursors and upsamplers beside the legacy image's
    (oj_xt_requester_new) -> (info, [return code per call], canvas planes (3, H, W) of uint16 codes -- or uint8 samples when the
    output conversion has no extra range bits -- initialised to 0xAA.., is_float).  cursors: per call the six cursor rows (legacy, residual)."""
    info, rq, is_float, out_max = OjInfo(), C.c_void_p(), C.c_int(0), C.c_int(0)
    rc = lib().oj_xt_requester_new(data, len(data), C.byref(info), C.byref(rq), C.byref(is_float), C.byref(out_max))
    if rc:
        raise OracleError(rc, "oj_xt_requester_new")
    sb = 2 if out_max.value > 255 else 1
    W, H, nc = info.width, info.height, 3
    canvas = np.full((nc, H, W), 0xAAAA if sb == 2 else 0xAA, np.uint16 if sb == 2 else np.uint8)
    rcs = []
    try:
        for minx, miny, maxx, maxy, c0, c1, ups, ctrafo, hmode in requests:
            maxx = W - 1 if maxx < 0 else maxx
            maxy = H - 1 if maxy < 0 else maxy
            dst = (C.c_void_p * 4)(*[canvas[c].ctypes.data for c in range(nc)] + [None])
            bpp = (C.c_int * 4)(sb, sb, sb, sb)
            bpr = (C.c_int * 4)(W * sb, W * sb, W * sb, W * sb)
            hh = miny + hmode if hmode else H
            bmh = (C.c_int * 4)(hh, hh, hh, hh)
            bmw = (C.c_int * 4)(W, W, W, W)
            rcs.append(lib().oj_requester_display(rq, minx, miny, maxx, maxy, c0, c1, ups, ctrafo, dst, bpp, bpr, bmw, bmh, sb))
            if cursors is not None:
                cursors.append([lib().oj_requester_cursor(rq, c) for c in range(3)] + [lib().oj_requester_cursor(rq, 4 + c) for c in range(3)])
    finally:
        lib().oj_requester_free(rq)
    return info, rcs, canvas, bool(is_float.value)


def run_requests_client(exe: str, data: bytes, requests, env=None, timeout=60):
    """The same sequence through tests/cxx/rect_calls.cpp linked against the real reference (REF_RECT_CALLS) or against
    libmijpeg.so -> (stdout lines, canvas planes or None)."""
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=tmpdir) as d:
        src, scr, dst = os.path.join(d, "in.jpg"), os.path.join(d, "script"), os.path.join(d, "out.bin")
        with open(src, "wb") as f:
            f.write(data)
        with open(scr, "w") as f:
            for r in requests:
                f.write(" ".join(str(int(v)) for v in r) + "\n")
        r = subprocess.run([exe, src, scr, dst], capture_output=True, text=True, timeout=timeout, env=env)
        lines = r.stdout.strip().splitlines()
        if not os.path.exists(dst) or not lines or not lines[0].startswith("info"):
            return lines, None
        W, H, nc, prec = (int(v) for v in lines[0].split()[1:5])
        raw = np.fromfile(dst, np.uint16 if prec > 8 else np.uint8)
        return lines, raw.reshape(nc, H, W)


def have_reference() -> bool:
    return os.path.exists(REF_BIN) and os.access(REF_BIN, os.X_OK)


def read_pnm(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        d = f.read()
    magic, rest = d.split(b"\n", 1)
    dims, rest = rest.split(b"\n", 1)
    maxv, rest = rest.split(b"\n", 1)
    w, h = map(int, dims.split())
    ch = 3 if magic == b"P6" else 1
    return np.frombuffer(rest, np.uint8, w * h * ch).reshape(h, w, ch)


def read_pnm_any(path: str) -> np.ndarray:
    """P5 / P6 with 8 or 16 bit samples (big endian, as cmd/bitmaphook.cpp writes them)."""
    with open(path, "rb") as f:
        d = f.read()
    magic, rest = d.split(b"\n", 1)
    dims, rest = rest.split(b"\n", 1)
    maxv, rest = rest.split(b"\n", 1)
    w, h = map(int, dims.split())
    sb = 2 if int(maxv) > 255 else 1
    ch = 1 if magic == b"P5" else max(3, len(rest) // (w * h * sb))  # four-component frames go out as "P6" with 4 samples per pixel
    if sb == 2:
        return np.frombuffer(rest, ">u2", w * h * ch).reshape(h, w, ch).astype(np.uint16)
    return np.frombuffer(rest, np.uint8, w * h * ch).reshape(h, w, ch).copy()


def reference_decode_status(data: bytes, extra_args=(), timeout=10):
    """Run the real reference CLI on a possibly damaged stream: -> (pixels or None, error).  error is 0 when a picture
    was written, the JPGERR_* code the CLI prints otherwise (cmd/reconstruct.cpp:360-364), or "crash:<signal>"."""
    import re
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=tmpdir) as d:
        src, dst = os.path.join(d, "in.jpg"), os.path.join(d, "out.ppm")
        with open(src, "wb") as f:
            f.write(data)
        try:
            r = subprocess.run([REF_BIN, *extra_args, src, dst], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout)
        except subprocess.TimeoutExpired:
            return None, "timeout"
        if r.returncode < 0:
            return None, f"crash:{-r.returncode}"
        m = re.search(rb"failed - error (-?\d+)", r.stderr)
        if m:
            return None, int(m.group(1))
        if not os.path.exists(dst):
            return None, "no output"
        try:
            with open(dst, "rb") as f:
                head = f.read(2)
            if head in (b"P5", b"P6"):
                return read_pnm_any(dst), 0
            # frames of two or four components go out as PGX: a list of raw planes with a header file each
            planes = []
            for line in open(dst).read().split():
                m = re.match(rb"PG ML \+(\d+) (\d+) (\d+)", open(line[:-4] + ".h", "rb").read())
                bits, w, h = (int(x) for x in m.groups())
                planes.append(np.fromfile(line, np.uint8 if bits <= 8 else ">u2").reshape(h, w))
            return np.stack(planes, axis=-1).astype(np.uint8 if planes[0].dtype == np.uint8 else np.uint16), 0
        except Exception:
            return None, "short output"


def write_ppm(path: str, img: np.ndarray) -> None:
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    with open(path, "wb") as f:
        f.write(b"P%d\n%d %d\n255\n" % (6 if ch == 3 else 5, w, h))
        f.write(np.ascontiguousarray(img, np.uint8).tobytes())


def read_pfm_reference(path: str) -> np.ndarray:
    """PFM as the reference CLI writes it (cmd/reconstruct.cpp:321-323, cmd/bitmaphook.cpp:282-305):
    'PF' / 'Pf' header, scale line, then big-endian float32 samples, top line first."""
    with open(path, "rb") as f:
        d = f.read()
    magic, rest = d.split(b"\n", 1)
    dims, rest = rest.split(b"\n", 1)
    scale, rest = rest.split(b"\n", 1)
    w, h = map(int, dims.split())
    ch = 3 if magic == b"PF" else 1
    return np.frombuffer(rest, ">f4", w * h * ch).reshape(h, w, ch).astype(np.float32)


def write_pfm(path: str, img: np.ndarray) -> None:
    """Standard little-endian PFM (bottom line first) as the reference encoder reads it."""
    h, w = img.shape[:2]
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (w, h))
        f.write(np.ascontiguousarray(img[::-1], "<f4").tobytes())


def reference_decode_hdr(data: bytes, extra_args=()) -> np.ndarray:
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=tmpdir) as d:
        src, dst = os.path.join(d, "in.jpg"), os.path.join(d, "out.pfm")
        with open(src, "wb") as f:
            f.write(data)
        subprocess.run([REF_BIN, *extra_args, src, dst], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return read_pfm_reference(dst)


def reference_encode_hdr(img: np.ndarray, args) -> bytes:
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=tmpdir) as d:
        src, dst = os.path.join(d, "in.pfm"), os.path.join(d, "out.jpg")
        write_pfm(src, img)
        subprocess.run([REF_BIN, *args, src, dst], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        with open(dst, "rb") as f:
            return f.read()


def reference_decode(data: bytes, extra_args=()) -> np.ndarray:
    """Decode with the real reference CLI (cmd/main.cpp:746-747 -> cmd/reconstruct.cpp:68)."""
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=tmpdir) as d:
        src, dst = os.path.join(d, "in.jpg"), os.path.join(d, "out.ppm")
        with open(src, "wb") as f:
            f.write(data)
        subprocess.run([REF_BIN, *extra_args, src, dst], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        return read_pnm(dst).copy()


def reference_encode(img: np.ndarray, args) -> bytes:
    """Encode with the real reference CLI, e.g. args=['-bl','-q','85','-s','1x1,2x2,2x2','-z','8']."""
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=tmpdir) as d:
        src, dst = os.path.join(d, "in.ppm"), os.path.join(d, "out.jpg")
        write_ppm(src, img)
        subprocess.run([REF_BIN, *args, src, dst], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        with open(dst, "rb") as f:
            return f.read()

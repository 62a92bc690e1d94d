"""ctypes binding of libjpeg_amd/libmijpeg.so (include/mijpeg.h) for tests and bench.py.

This is plumbing above the C ABI: it mirrors the decode half of the reference's `class JPEG`
(interface/jpeg.hpp:185-252): Read -> `Decoder.read`, GetInformation -> `Decoder.info`,
DisplayRectangle -> `Decoder.reconstruct` / `reconstruct_rect`, LastError -> `MijpegError`.
The product path has no CPU fallback: if the shared library (HIP kernels inside) is missing, `lib()`
fails loudly; if no GPU is present, every reconstruct call raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys
import weakref

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIJPEG_LIBRARY") or os.path.join(HERE, "libmijpeg.so")  # MIJPEG_LIBRARY: another build of the same library (kernel experiments)

FLAG_NO_COLOR_TRANSFORM = 1
FLAG_FORCE_GENERIC = 2
FLAG_FORCE_SAFE = 4
FLAG_DEVICE_OUTPUT = 8
FLAG_NO_UPSAMPLING = 16
FLAG_SPECULATIVE = 32
FLAG_FORCE_DOT2 = 64  # (testing) the packed 4:2:0 kernel's 16-bit second pass whatever the range check says

ERR_DEVICE = -8191
ERR_NOT_AVAILABLE = -1029


class MijpegInfo(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("components", C.c_int32), ("precision", C.c_int32),
        ("hsamp", C.c_int32 * 4), ("vsamp", C.c_int32 * 4), ("subx", C.c_int32 * 4), ("suby", C.c_int32 * 4),
        ("quant_index", C.c_int32 * 4), ("mcus_x", C.c_int32), ("mcus_y", C.c_int32),
        ("blocks_w", C.c_int32 * 4), ("blocks_h", C.c_int32 * 4), ("restart_interval", C.c_int32),
        ("ycbcr", C.c_int32), ("fast_arith", C.c_int32), ("coef_offset", C.c_int64 * 4),
        ("coef_count", C.c_int64), ("quant", (C.c_uint16 * 64) * 4), ("range_max", C.c_int32 * 4),
        ("sample_bytes", C.c_int32), ("xt", C.c_int32), ("is_float", C.c_int32), ("progressive", C.c_int32),
        ("coef_wide", C.c_int32), ("dnl", C.c_int32), ("rows", C.c_int32 * 4),
    ]


class MijpegXtParams(C.Structure):
    _fields_ = [
        ("residual", MijpegInfo), ("ltable", (C.c_int32 * 4096) * 3), ("ltable_entries", C.c_int32), ("hidden_bits", C.c_int32),
        ("residual_hidden_bits", C.c_int32), ("residual_wide", C.c_int32), ("ltrafo_ycbcr", C.c_int32), ("rtrafo_ycbcr", C.c_int32),
        ("out_max", C.c_int32), ("out_shift", C.c_int32), ("is_float", C.c_int32), ("clamp", C.c_int32),
        ("general", C.c_int32), ("lmat", C.c_int32 * 9), ("rmat", C.c_int32 * 9), ("cmat", C.c_int32 * 9),
        ("rdct_bypass", C.c_int32), ("noise_shaping", C.c_int32), ("qtable_entries", C.c_int32),
        ("qtable", C.c_void_p * 3), ("r2table", C.c_void_p * 3), ("no_residual", C.c_int32), ("ltrafo_standard", C.c_int32), ("rct", C.c_int32), ("rbits", C.c_int32),
    ]


class MijpegBatch(C.Structure):
    _fields_ = [
        ("info", MijpegInfo), ("coef_dev", C.c_void_p), ("coef_frame_stride", C.c_int64),
        ("quant_dev", C.c_void_p), ("out_dev", C.c_void_p), ("out_frame_stride", C.c_int64),
        ("out_row_stride", C.c_int64), ("frames", C.c_int32), ("flags", C.c_uint32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("xt", C.POINTER(MijpegXtParams)),
    ]


class MijpegForwardBatch(C.Structure):
    _fields_ = [
        ("info", MijpegInfo), ("pixels_dev", C.c_void_p), ("pixel_frame_stride", C.c_int64), ("pixel_row_stride", C.c_int64),
        ("coef_dev", C.c_void_p), ("coef_frame_stride", C.c_int64), ("frames", C.c_int32), ("flags", C.c_uint32),
    ]


class MijpegBitmap(C.Structure):
    _fields_ = [("data", C.c_void_p), ("bytes_per_pixel", C.c_int32), ("bytes_per_row", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32)]


class MijpegError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"mijpeg error {code}: {message}")
        self.code = code
        self.message = message


def build(force: bool = False) -> str:
    """Compile libmijpeg.so in-tree (hipcc --offload-arch=gfx950); cross-compiles without a GPU."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.run(["make", "-s", "-j4", "-C", os.path.join(HERE, "csrc")], check=True)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the reconstruction path)")
        # When PyTorch is used in the same process (tests and bench.py use it for device buffers, streams and
        # torch.distributed) its bundled HIP runtime must be the one this library binds to: two HIP runtimes in
        # one process do not see each other's device state.  Importing torch first makes the dynamic linker
        # resolve libamdhip64 to the copy that is already loaded.  Without torch the system ROCm runtime is used.
        try:
            if not os.environ.get("MIJPEG_NO_TORCH"):  # (host-only helper processes have no use for it)
                import torch  # noqa: F401
        except Exception:  # torch is plumbing, not a dependency of the C ABI
            pass
        L = C.CDLL(LIB_PATH)
        P = C.POINTER
        L.mijpeg_version.restype = C.c_char_p
        L.mijpeg_create.argtypes = [P(C.c_void_p), C.c_int]
        L.mijpeg_destroy.argtypes = [C.c_void_p]
        L.mijpeg_destroy.restype = None
        L.mijpeg_set_input.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.mijpeg_read_header.argtypes = [C.c_void_p, P(MijpegInfo)]
        L.mijpeg_get_info.argtypes = [C.c_void_p, P(MijpegInfo)]
        L.mijpeg_get_xt_params.argtypes = [C.c_void_p, P(MijpegXtParams)]
        L.mijpeg_decode_coefficients.argtypes = [C.c_void_p, C.c_int]
        L.mijpeg_decode_coefficients_device.argtypes = [C.c_void_p, C.c_int]
        L.mijpeg_decode_batch_device.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_int]
        L.mijpeg_submit_batch_device.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_int]
        L.mijpeg_finish_batch_device.argtypes = [C.c_void_p]
        L.mijpeg_batch_speculation.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.mijpeg_reconstruct_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_uint32, C.c_int]
        L.mijpeg_device_walk_rounds.argtypes = [C.c_void_p]
        L.mijpeg_device_walk_rounds.restype = C.c_int
        L.mijpeg_speculative_scans.argtypes = [C.POINTER(C.c_int64)]
        L.mijpeg_speculative_scans.restype = C.c_int64
        L.mijpeg_coefficients.argtypes = [C.c_void_p, C.c_int]
        L.mijpeg_coefficients.restype = C.c_void_p
        L.mijpeg_coefficients32.argtypes = [C.c_void_p, C.c_int]
        L.mijpeg_coefficients32.restype = C.c_void_p
        L.mijpeg_device_coefficients.argtypes = [C.c_void_p]
        L.mijpeg_device_coefficients.restype = C.c_void_p
        L.mijpeg_reconstruct_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_uint32, C.c_int]
        L.mijpeg_reconstruct_rect.argtypes = [C.c_void_p] + [C.c_int32] * 6 + [C.c_uint32, P(C.c_void_p), P(C.c_int32), P(C.c_int32)]
        L.mijpeg_reconstruct_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_uint32]
        L.mijpeg_display_rect.argtypes = [C.c_void_p] + [C.c_int32] * 6 + [C.c_uint32, P(MijpegBitmap)]
        L.mijpeg_display_plan.argtypes = [C.c_void_p] + [C.c_int32] * 6 + [C.c_uint32, P(C.c_uint32), P(C.c_int32)]
        L.mijpeg_display_cursor.argtypes = [C.c_void_p, C.c_int]
        L.mijpeg_host_alloc.argtypes = [C.c_size_t]
        L.mijpeg_host_alloc.restype = C.c_void_p
        L.mijpeg_host_free.argtypes = [C.c_void_p]
        L.mijpeg_host_free.restype = None
        L.mijpeg_last_error.argtypes = [C.c_void_p, P(C.c_char_p)]
        L.mijpeg_alpha_channel.argtypes = [C.c_void_p]
        L.mijpeg_alpha_channel.restype = C.c_void_p
        L.mijpeg_has_alpha.argtypes = [C.c_void_p]
        L.mijpeg_batch_pipeline_create.argtypes = [P(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int]
        L.mijpeg_batch_pipeline_destroy.argtypes = [C.c_void_p]
        L.mijpeg_batch_pipeline_destroy.restype = None
        L.mijpeg_batch_pipeline_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.mijpeg_batch_pipeline_schedule.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.mijpeg_batch_pipeline_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.mijpeg_batch_pipeline_last_error.argtypes = [C.c_void_p, P(C.c_char_p)]
        L.mijpeg_batch_pipeline_speculation.argtypes = [C.c_void_p, C.c_int]
        L.mijpeg_batch_pipeline_decoder.argtypes = [C.c_void_p, C.c_int]
        L.mijpeg_batch_pipeline_decoder.restype = C.c_void_p
        L.mijpeg_alpha_info.argtypes = [C.c_void_p, P(C.c_int32), P(C.c_int32)]
        L.mijpeg_last_timing.argtypes = [C.c_void_p, P(C.c_double)]
        L.mijpeg_launch_reconstruct.argtypes = [P(MijpegBatch), C.c_void_p]
        L.mijpeg_kernel_name.argtypes = [P(MijpegBatch)]
        L.mijpeg_kernel_name.restype = C.c_char_p
        L.mijpeg_workspace_bytes.argtypes = [P(MijpegBatch)]
        L.mijpeg_workspace_bytes.restype = C.c_size_t
        _lib = L
    return _lib


def _foreign_work_done() -> None:
    """The decoder object works on a stream of its own (non-blocking): device buffers handed to it must not have work
    pending on other streams -- a torch fill queued a moment ago, say (include/mijpeg.h states the requirement).  These
    bindings are the tests' and bench.py's: they simply wait for torch's streams."""
    t = sys.modules.get("torch")
    if t is not None and t.cuda.is_available():
        t.cuda.synchronize()


class Decoder:
    """One image at a time.  device=None -> host-only object (parsing + Huffman decoding)."""

    def __init__(self, device: int | None = 0):
        self._h = C.c_void_p()
        rc = lib().mijpeg_create(C.byref(self._h), -1 if device is None else int(device))
        if rc:
            raise MijpegError(rc, "mijpeg_create failed (no usable HIP device?)")
        self._data = None
        self.info: MijpegInfo | None = None

    def close(self):
        # the alpha decoders handed out are owned by this object's handle: they die with it
        for ref in getattr(self, "_children", ()):
            child = ref()
            if child is not None:
                child._h = C.c_void_p()
        self._children = []
        if self._h:
            if not getattr(self, "_borrowed", False):
                lib().mijpeg_destroy(self._h)
            self._h = C.c_void_p()

    def alpha_channel(self) -> "Decoder | None":
        """The decoder object of the file's alpha channel (JPEG XT ALFA box), owned by this one and valid until it reads again; None
        when the decoded file has none.  Its `info` is the alpha image's (one component)."""
        h = lib().mijpeg_alpha_channel(self._h)
        if not h:
            return None
        a = Decoder.__new__(Decoder)
        a._h = C.c_void_p(h)
        a._borrowed = True
        a._data = None
        a._owner = self  # keep the owning object alive
        if not hasattr(self, "_children"):
            self._children = []
        self._children.append(weakref.ref(a))
        a.info = MijpegInfo()
        a._check(lib().mijpeg_get_info(a._h, C.byref(a.info)))
        return a

    def alpha_info(self):
        """-> (mode, (r, g, b)): compositing method and matte colour of the alpha merging specification's AMUL box (mode -1: none)."""
        mode = C.c_int32(-1)
        matte = (C.c_int32 * 3)()
        self._check(lib().mijpeg_alpha_info(self._h, C.byref(mode), matte))
        return mode.value, tuple(matte)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc:
            if not self._h:
                raise MijpegError(rc, "the decoder object is closed")
            msg = C.c_char_p()
            lib().mijpeg_last_error(self._h, C.byref(msg))
            raise MijpegError(rc, (msg.value or b"").decode())

    def read_header(self, data: bytes) -> MijpegInfo:
        """Tables + frame header (JPEG::Read up to the first scan, then JPEG::GetInformation)."""
        self._data = data  # keep alive: the library borrows the bytes
        self._check(lib().mijpeg_set_input(self._h, data, len(data)))
        info = MijpegInfo()
        self._check(lib().mijpeg_read_header(self._h, C.byref(info)))
        self.info = info
        return info

    def read(self, data: bytes, threads: int = 0, entropy: str = "host") -> MijpegInfo:
        """JPEG::Read: parse everything and entropy-decode all scans (uploads when a device is attached).
        entropy = "host" (restart-interval parallel on the CPU), "gpu" (on the device, error if the stream does not
        qualify), "auto" (device when it qualifies and has enough restart intervals to occupy it) or "prefer-gpu" (device
        whenever it qualifies, however small; host otherwise -- damaged streams always end up there)."""
        self._data = data
        self._check(lib().mijpeg_set_input(self._h, data, len(data)))
        self.entropy_used = "host"
        if entropy in ("gpu", "auto", "prefer-gpu"):
            rc = lib().mijpeg_decode_coefficients_device(self._h, 0 if entropy == "auto" else 1)
            if rc == 0:
                self.entropy_used = "gpu"
            elif rc != ERR_NOT_AVAILABLE or entropy == "gpu":
                self._check(rc)
                raise MijpegError(rc, "stream does not qualify for on-device entropy decoding")
        if self.entropy_used == "host":
            self._check(lib().mijpeg_decode_coefficients(self._h, threads))
        info = MijpegInfo()
        self._check(lib().mijpeg_get_info(self._h, C.byref(info)))
        self.info = info
        return info

    def device_walk_rounds(self) -> int:
        """Rounds the on-device self-synchronising walk took in the last device entropy decode (0 = none needed)."""
        return int(lib().mijpeg_device_walk_rounds(self._h))

    def encode(self, img: np.ndarray, quality: int = 85, subsampling: str = "444", restart_mcus: int = 0, optimize: bool = False,
               coder: str = "gpu") -> bytes:
        """mijpeg_encode_image_ex: (H, W, 3) RGB or (H, W) grey uint8 picture -> baseline JPEG; forward transform on the device,
        entropy coder on the device (coder="gpu") or on the host cores (coder="host")."""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape[:2]
        nc = 1 if img.ndim == 2 else img.shape[2]
        hs, vs = {"444": ((1, 1, 1), (1, 1, 1)), "420": ((2, 1, 1), (2, 1, 1)), "422": ((2, 1, 1), (1, 1, 1)), "440": ((1, 1, 1), (2, 1, 1)),
                  "411": ((4, 1, 1), (1, 1, 1))}[subsampling]
        L = lib()
        L.mijpeg_encode_image_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int, C.POINTER(C.c_int32),
                                             C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.mijpeg_free.argtypes = [C.c_void_p]
        p, n = C.c_void_p(), C.c_size_t()
        self._check(L.mijpeg_encode_image_ex(self._h, img.ctypes.data, w, h, nc, w * nc, quality, (C.c_int32 * 4)(*hs, 1), (C.c_int32 * 4)(*vs, 1),
                                             restart_mcus, 1 if optimize else 0, 1 if coder == "host" else 0, C.byref(p), C.byref(n)))
        try:
            return C.string_at(p, n.value)
        finally:
            L.mijpeg_free(p)

    def encode_batch_device(self, info: MijpegInfo, pixels_dev: int, coef_dev: int, frames: int, pixel_row_stride: int, pixel_frame_stride: int,
                            restart_mcus: int = 0, optimize: bool = False):
        """mijpeg_encode_batch_device: frames resident in HBM -> list of baseline JPEG streams (forward kernels + device entropy coder)."""
        _foreign_work_done()
        b = MijpegForwardBatch()
        C.memmove(C.byref(b.info), C.byref(info), C.sizeof(MijpegInfo))
        b.pixels_dev, b.pixel_frame_stride, b.pixel_row_stride = pixels_dev, pixel_frame_stride, pixel_row_stride
        b.coef_dev, b.coef_frame_stride, b.frames = coef_dev, info.coef_count, frames
        L = lib()
        L.mijpeg_encode_batch_device.argtypes = [C.c_void_p, C.POINTER(MijpegForwardBatch), C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.mijpeg_free.argtypes = [C.c_void_p]
        ptrs, sizes = (C.c_void_p * frames)(), (C.c_size_t * frames)()
        self._check(L.mijpeg_encode_batch_device(self._h, C.byref(b), restart_mcus, 1 if optimize else 0, ptrs, sizes))
        out = []
        for f in range(frames):
            out.append(C.string_at(ptrs[f], sizes[f]))
            L.mijpeg_free(ptrs[f])
        return out

    def xt_params(self) -> MijpegXtParams:
        xt = MijpegXtParams()
        self._check(lib().mijpeg_get_xt_params(self._h, C.byref(xt)))
        return xt

    def coefficients(self, comp: int) -> np.ndarray:
        f = self.info
        wide = bool(f.coef_wide)  # a damaged stream whose coefficients left the 16-bit range: int32 planes
        p = (lib().mijpeg_coefficients32 if wide else lib().mijpeg_coefficients)(self._h, comp)
        if not p:
            raise MijpegError(-1031, "no decoded coefficients")
        n = f.blocks_w[comp] * f.blocks_h[comp] * 64
        arr = np.ctypeslib.as_array(((C.c_int32 if wide else C.c_int16) * n).from_address(p))
        return arr.reshape(f.blocks_h[comp], f.blocks_w[comp], 64).copy()

    def residual_coefficients(self, comp: int) -> np.ndarray:
        """JPEG XT: the residual codestream's planes, which follow the legacy ones in the same coefficient buffer
        (mijpeg_xt_params.residual.coef_offset, in int16 units): int32 when the residual frame has hidden bits."""
        x = self.xt_params()
        base = lib().mijpeg_coefficients(self._h, 0)
        if not base or not self.info.xt:
            raise MijpegError(-1031, "no decoded residual coefficients")
        r = x.residual
        wide = bool(x.residual_wide)
        n = r.blocks_w[comp] * r.blocks_h[comp] * 64
        addr = base + 2 * int(r.coef_offset[comp])
        arr = np.ctypeslib.as_array(((C.c_int32 if wide else C.c_int16) * n).from_address(addr))
        return arr.reshape(r.blocks_h[comp], r.blocks_w[comp], 64).copy()

    def device_coefficients(self) -> int:
        return lib().mijpeg_device_coefficients(self._h) or 0

    def reconstruct(self, flags: int = 0, out: np.ndarray | None = None) -> np.ndarray:
        """JPEG::DisplayRectangle over the whole canvas -> (H, W, C) uint8 in host memory."""
        f = self.info
        return self.reconstruct_rect(0, 0, f.width - 1, f.height - 1, flags=flags, out=out)

    def reconstruct_rect(self, x0, y0, x1, y1, comp0=0, comp1=None, flags: int = 0,
                         out: np.ndarray | None = None) -> np.ndarray:
        f = self.info
        nc = f.components
        sb = max(1, f.sample_bytes)  # 2: precision 12 or JPEG XT (16-bit codes)
        comp1 = nc - 1 if comp1 is None else comp1
        if out is None:
            out = np.zeros((f.height, f.width, nc), np.uint8 if sb == 1 else np.uint16)
        base = out.ctypes.data
        dst = (C.c_void_p * 4)(*[base + c * sb if c < nc else None for c in range(4)])
        bpp = (C.c_int32 * 4)(*([nc * sb] * 4))
        bpr = (C.c_int32 * 4)(*([out.strides[0]] * 4))
        self._check(lib().mijpeg_reconstruct_rect(self._h, x0, y0, x1, y1, comp0, comp1, flags, dst, bpp, bpr))
        return out

    def display_rect(self, canvas: np.ndarray, x0, y0, x1, y1, comp0=0, comp1=None, flags: int = 0, bm_height: int | None = None) -> None:
        """mijpeg_display_rect: one JPEG::DisplayRectangle call of a sequence (the object keeps the reference's state between
        calls).  canvas: (ncomp, H, W) planar uint8 / uint16, written in place; bm_height: BIO_HEIGHT the hook would report."""
        f = self.info
        nc = f.components
        comp1 = nc - 1 if comp1 is None else comp1
        sb = canvas.dtype.itemsize
        H, W = canvas.shape[1:]
        maps = (MijpegBitmap * 4)()
        for c in range(nc):
            maps[c].data = canvas[c].ctypes.data
            maps[c].bytes_per_pixel = sb
            maps[c].bytes_per_row = canvas.strides[1]
            maps[c].width = W
            maps[c].height = H if bm_height is None else bm_height
        self._check(lib().mijpeg_display_rect(self._h, x0, y0, x1, y1, comp0, comp1, flags, maps))

    def reconstruct_cli(self, flags: int = 0) -> np.ndarray:
        """What the reference's command line shows of the image (cmd/reconstruct.cpp:272-342 with upsampling): frames of one
        or three components are requested in stripes over all components -- the plain picture --, frames of two or four
        component by component (PGX), where the state JPEG::DisplayRectangle keeps between calls shows (mijpeg_display_rect).
        -> (H, W, C).  Like the reference's loop it can run once per decoded image."""
        f = self.info
        if f.components in (1, 3):
            return self.reconstruct(flags)
        canvas = np.zeros((f.components, f.height, f.width), np.uint8 if max(1, f.sample_bytes) == 1 else np.uint16)
        for c in range(f.components):
            for y in range(0, f.height, 8):
                self.display_rect(canvas, 0, y, f.width - 1, min(y + 7, f.height - 1), c, c, flags, bm_height=y + 8)
        return np.ascontiguousarray(np.moveaxis(canvas, 0, -1))

    def display_plan(self, x0, y0, x1, y1, comp0, comp1, flags: int, bm_height: int):
        """mijpeg_display_plan (no device needed): advance the request state, return the plan as a dict."""
        out = (C.c_int32 * 32)()
        hh = (C.c_uint32 * 4)(bm_height, bm_height, bm_height, bm_height)
        self._check(lib().mijpeg_display_plan(self._h, x0, y0, x1, y1, comp0, comp1, flags, hh, out))
        o = list(out)
        return dict(nothing=o[0], plain=o[1], ycc=o[2], view=o[3], region=tuple(o[4:8]),
                    comps=[dict(cursor=o[8 + 6 * c], g0=o[9 + 6 * c], g1=o[10 + 6 * c], wstart=o[11 + 6 * c], wlimit=o[12 + 6 * c],
                                zeros=o[13 + 6 * c]) for c in range(4)])

    def display_cursor(self, component: int) -> int:
        """mijpeg_display_cursor: the row the cursor of `component` stands at (JPEG XT: 4 + c = the residual image's)."""
        return int(lib().mijpeg_display_cursor(self._h, component))

    def decode_batch_device(self, streams, min_intervals: int = 0) -> MijpegInfo:
        """mijpeg_decode_batch_device: n streams of one shape -> n coefficient stores in HBM with one Huffman kernel launch."""
        n = len(streams)
        self._batch = list(streams)  # keep the bytes alive during the call
        arr = (C.c_char_p * n)(*self._batch)
        sizes = (C.c_size_t * n)(*[len(s) for s in self._batch])
        self._check(lib().mijpeg_decode_batch_device(self._h, arr, sizes, n, min_intervals))
        info = MijpegInfo()
        self._check(lib().mijpeg_get_info(self._h, C.byref(info)))
        self.info = info
        self.batch_frames = n
        return info

    def prepare_batch_host(self, streams) -> None:
        """mijpeg_prepare_batch_host: the host half of a batch submit alone (no device needed)."""
        n = len(streams)
        arr = (C.c_char_p * n)(*streams)
        sizes = (C.c_size_t * n)(*[len(s) for s in streams])
        L = lib()
        L.mijpeg_prepare_batch_host.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int]
        self._check(L.mijpeg_prepare_batch_host(self._h, arr, sizes, n))

    @staticmethod
    def stream_arrays(streams):
        """(pointer array, size array, list that keeps the bytes alive) for the batch calls: a pipeline builds them once per chunk
        instead of once per call (a hundred microseconds of ctypes per 24 streams, on the thread that feeds the link)."""
        keep = list(streams)
        n = len(keep)
        return (C.c_char_p * n)(*keep), (C.c_size_t * n)(*[len(s) for s in keep]), keep

    def submit_batch_device(self, streams, min_intervals: int = 0, arrays=None) -> None:
        """mijpeg_submit_batch_device: parse, gather, enqueue upload + Huffman kernel; returns without waiting for the device.
        arrays: what stream_arrays(streams) returned (optional)."""
        arr, sizes, self._batch = arrays if arrays is not None else self.stream_arrays(streams)
        n = len(self._batch)
        self._check(lib().mijpeg_submit_batch_device(self._h, arr, sizes, n, min_intervals))
        self.batch_frames = n

    def finish_batch_device(self) -> MijpegInfo:
        """mijpeg_finish_batch_device: wait for a submitted batch, raise what its decode reported."""
        self._check(lib().mijpeg_finish_batch_device(self._h))
        info = MijpegInfo()
        self._check(lib().mijpeg_get_info(self._h, C.byref(info)))
        self.info = info
        return info

    def batch_speculation(self):
        """-> (last validation had to reconstruct again, speculative launches of this object, of them redone): MIJPEG_FLAG_SPECULATIVE"""
        launched, redone = C.c_int64(0), C.c_int64(0)
        again = lib().mijpeg_batch_speculation(self._h, C.byref(launched), C.byref(redone))
        return bool(again == 1), launched.value, redone.value

    def reconstruct_batch_device(self, dst_ptr: int, frame_stride: int, row_stride: int, flags: int = 0, sync: bool = True,
                                 wait_foreign: bool = True):
        """wait_foreign=False: the caller vouches that nothing is pending on other streams for the destination (pipelines that
        must not stall the device between their stages)."""
        if wait_foreign:
            _foreign_work_done()
        self._check(lib().mijpeg_reconstruct_batch_device(self._h, dst_ptr, frame_stride, row_stride, flags, 1 if sync else 0))

    def reconstruct_unsampled(self, comp: int, flags: int = 0) -> np.ndarray:
        """JPGTAG_DECODER_UPSAMPLE = false: component `comp` on its own sample grid, no colour transformation
        (what the reference CLI's -U writes into out_<comp>.raw)."""
        f = self.info
        sb = max(1, f.sample_bytes)
        w, h = -(-f.width // f.subx[comp]), -(-f.height // f.suby[comp])
        out = np.zeros((h, w), np.uint8 if sb == 1 else np.uint16)
        dst = (C.c_void_p * 4)(*[out.ctypes.data if c == comp else None for c in range(4)])
        bpp = (C.c_int32 * 4)(*([sb] * 4))
        bpr = (C.c_int32 * 4)(*([out.strides[0]] * 4))
        self._check(lib().mijpeg_reconstruct_rect(self._h, 0, 0, f.width - 1, f.height - 1, comp, comp,
                                                  flags | FLAG_NO_UPSAMPLING, dst, bpp, bpr))
        return out

    def reconstruct_rect_device(self, x0, y0, x1, y1, ptrs, bytes_per_pixel, bytes_per_row, comp0=0, comp1=None,
                                flags: int = 0):
        """mijpeg_reconstruct_rect with MIJPEG_FLAG_DEVICE_OUTPUT: `ptrs[c]` is the device address of canvas pixel
        (0,0) of component c (0/None = component not wanted), strides in bytes as in the reference's ImageBitMap."""
        _foreign_work_done()
        nc = self.info.components
        comp1 = nc - 1 if comp1 is None else comp1
        ptrs = list(ptrs) + [None] * (4 - len(ptrs))
        dst = (C.c_void_p * 4)(*[p or None for p in ptrs])
        bpp = (C.c_int32 * 4)(*(list(bytes_per_pixel) + [0] * (4 - len(bytes_per_pixel))))
        bpr = (C.c_int32 * 4)(*(list(bytes_per_row) + [0] * (4 - len(bytes_per_row))))
        self._check(lib().mijpeg_reconstruct_rect(self._h, x0, y0, x1, y1, comp0, comp1, flags | FLAG_DEVICE_OUTPUT,
                                                  dst, bpp, bpr))

    def reconstruct_into(self, out: np.ndarray, flags: int = 0) -> np.ndarray:
        """Whole frame with the device-to-host copy landing directly in `out` (fast when `out` is pinned, see
        pinned_frame()); `out` is (H, W, C) with contiguous pixels, any row stride."""
        self._check(lib().mijpeg_reconstruct_host(self._h, out.ctypes.data, out.strides[0], flags))
        return out

    def reconstruct_device(self, dst_ptr: int, row_stride: int, flags: int = 0, sync: bool = True):
        _foreign_work_done()
        self._check(lib().mijpeg_reconstruct_device(self._h, dst_ptr, row_stride, flags, 1 if sync else 0))

    def last_warning(self):
        """JPEG::LastWarning: (code, message) -- (0, None) when the reference would not warn."""
        L = lib()
        L.mijpeg_last_warning.argtypes = [C.c_void_p, C.POINTER(C.c_char_p)]
        msg = C.c_char_p()
        code = L.mijpeg_last_warning(self._h, C.byref(msg))
        return code, (msg.value.decode() if msg.value else None)

    def synchronize(self):
        """Wait for everything this object has enqueued on its stream (mijpeg_finish_batch_device is the batch flavour;
        a reconstruction launched with sync=False is waited for by the next synchronous call on the object: this one)."""
        L = lib()
        L.mijpeg_synchronize.argtypes = [C.c_void_p]
        self._check(L.mijpeg_synchronize(self._h))

    def stream_wait(self, client_stream: int = 0):
        """Everything this object has enqueued so far happens before what the caller enqueues on `client_stream` (a hipStream_t as
        an integer, e.g. torch.cuda.Stream().cuda_stream) from now on; the host does not block (mijpeg_stream_wait)."""
        L = lib()
        L.mijpeg_stream_wait.argtypes = [C.c_void_p, C.c_void_p]
        self._check(L.mijpeg_stream_wait(self._h, C.c_void_p(client_stream)))

    def timing(self):
        t = (C.c_double * 4)()
        lib().mijpeg_last_timing(self._h, t)
        return dict(huffman=t[0], h2d_wait=t[1], kernel=t[2], d2h=t[3])


class PinnedFrame:
    """(H, W, C) frame buffer in pinned host memory (mijpeg_host_alloc); .array is the numpy view."""

    def __init__(self, height: int, width: int, channels: int, dtype=np.uint8):
        line = (width * channels * np.dtype(dtype).itemsize + 7) & ~7
        self._bytes = line * height
        self._p = lib().mijpeg_host_alloc(self._bytes)
        if not self._p:
            raise MemoryError("mijpeg_host_alloc failed")
        raw = np.ctypeslib.as_array((C.c_uint8 * self._bytes).from_address(self._p)).reshape(height, line)
        self.array = raw[:, :width * channels * np.dtype(dtype).itemsize].view(dtype).reshape(height, width, channels)

    def close(self):
        if self._p:
            self.array = None
            lib().mijpeg_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def speculative_scans():
    """(scans, pieces) decoded by the host's self-synchronising parallel path since the library was loaded."""
    p = C.c_int64(0)
    n = lib().mijpeg_speculative_scans(C.byref(p))
    return int(n), int(p.value)


def trim_cache() -> int:
    """Free the buffers destroyed decoder objects left in the process-wide cache; returns the bytes freed."""
    L = lib()
    L.mijpeg_trim_cache.restype = C.c_size_t
    return int(L.mijpeg_trim_cache())


def default_threads() -> int:
    L = lib()
    L.mijpeg_default_threads.restype = C.c_int
    return int(L.mijpeg_default_threads())


def decode(data: bytes, device: int = 0, threads: int = 0, flags: int = 0) -> np.ndarray:
    """`jpeg in.jpg out.ppm` in one call: bytes -> (H, W, C) uint8."""
    d = Decoder(device)
    try:
        d.read(data, threads)
        return d.reconstruct(flags)
    finally:
        d.close()


def launch_reconstruct(info: MijpegInfo, coef_dev: int, out_dev: int, frames: int, out_row_stride: int,
                       out_frame_stride: int, coef_frame_stride: int | None = None, flags: int = 0,
                       workspace: int = 0, workspace_bytes: int = 0, stream: int = 0, xt: MijpegXtParams | None = None,
                       quant_dev: int = 0) -> None:
    """Stateless batch launch (device-resident coefficients -> device pixels), asynchronous.  quant_dev: device address of
    per-frame tables, u16 [frames][4][64] (component, natural order); info.quant then holds the batch-wide maxima."""
    b = MijpegBatch()
    C.memmove(C.byref(b.info), C.byref(info), C.sizeof(MijpegInfo))
    b.coef_dev = coef_dev
    b.coef_frame_stride = info.coef_count if coef_frame_stride is None else coef_frame_stride
    b.out_dev = out_dev
    b.out_frame_stride = out_frame_stride
    b.out_row_stride = out_row_stride
    b.frames = frames
    b.flags = flags
    b.workspace = workspace
    b.workspace_bytes = workspace_bytes
    b.quant_dev = quant_dev or None
    if xt is not None:
        b.xt = C.pointer(xt)
    rc = lib().mijpeg_launch_reconstruct(C.byref(b), stream)
    if rc:
        raise MijpegError(rc, "mijpeg_launch_reconstruct failed")


def frame_layout(width: int, height: int, components: int, hsamp, vsamp, quant, quant_index=None, ycbcr: int = 1) -> MijpegInfo:
    """mijpeg_frame_layout: geometry of a frame to be coded (encoder direction).  quant: up to four tables of 64 deltas,
    natural order; quant_index[c]: table of component c (default 0 for the first, 1 for the others)."""
    f = MijpegInfo()
    f.width, f.height, f.components, f.precision, f.ycbcr = width, height, components, 8, ycbcr
    for c in range(components):
        f.hsamp[c], f.vsamp[c] = hsamp[c], vsamp[c]
        f.quant_index[c] = (0 if c == 0 else min(1, len(quant) - 1)) if quant_index is None else quant_index[c]
    for t, tab in enumerate(quant):
        for i in range(64):
            f.quant[t][i] = int(tab[i])
    L = lib()
    L.mijpeg_frame_layout.argtypes = [C.POINTER(MijpegInfo)]
    rc = L.mijpeg_frame_layout(C.byref(f))
    if rc:
        raise MijpegError(rc, "mijpeg_frame_layout failed")
    return f


def launch_forward(info: MijpegInfo, pixels_dev: int, coef_dev: int, frames: int, pixel_row_stride: int, pixel_frame_stride: int,
                   coef_frame_stride: int | None = None, stream: int = 0) -> None:
    """Encoder direction, stateless: device-resident interleaved pixels -> quantised coefficient planes in HBM (asynchronous)."""
    b = MijpegForwardBatch()
    C.memmove(C.byref(b.info), C.byref(info), C.sizeof(MijpegInfo))
    b.pixels_dev = pixels_dev
    b.pixel_frame_stride = pixel_frame_stride
    b.pixel_row_stride = pixel_row_stride
    b.coef_dev = coef_dev
    b.coef_frame_stride = info.coef_count if coef_frame_stride is None else coef_frame_stride
    b.frames = frames
    L = lib()
    L.mijpeg_launch_forward.argtypes = [C.POINTER(MijpegForwardBatch), C.c_void_p]
    rc = L.mijpeg_launch_forward(C.byref(b), stream)
    if rc:
        raise MijpegError(rc, "mijpeg_launch_forward failed")


def encode_coefficients(info: MijpegInfo, coef: np.ndarray, restart_interval: int = 0, optimize: bool = False, threads: int = 0) -> bytes:
    """mijpeg_encode_coefficients: quantised coefficient planes (host int16, info.coef_count of them) -> baseline JPEG stream."""
    coef = np.ascontiguousarray(coef, np.int16).reshape(-1)
    assert coef.size == info.coef_count
    L = lib()
    L.mijpeg_encode_coefficients.argtypes = [C.POINTER(MijpegInfo), C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.mijpeg_free.argtypes = [C.c_void_p]
    p, n = C.c_void_p(), C.c_size_t()
    rc = L.mijpeg_encode_coefficients(C.byref(info), coef.ctypes.data, restart_interval, 1 if optimize else 0, threads, C.byref(p), C.byref(n))
    if rc:
        raise MijpegError(rc, "mijpeg_encode_coefficients failed")
    try:
        return C.string_at(p, n.value)
    finally:
        L.mijpeg_free(p)


def workspace_bytes(info: MijpegInfo, frames: int, flags: int = 0, own_tables: bool = False, xt: "MijpegXtParams | None" = None) -> int:
    b = MijpegBatch()
    C.memmove(C.byref(b.info), C.byref(info), C.sizeof(MijpegInfo))
    b.frames = frames
    b.flags = flags
    if xt is not None:
        b.xt = C.pointer(xt)
    if own_tables:  # only asked whether it is set
        b.quant_dev = 16
    return int(lib().mijpeg_workspace_bytes(C.byref(b)))


def kernel_name(info: MijpegInfo, flags: int = 0, xt: "MijpegXtParams | None" = None) -> str:
    b = MijpegBatch()
    C.memmove(C.byref(b.info), C.byref(info), C.sizeof(MijpegInfo))
    b.flags = flags
    if xt is not None:
        b.xt = C.pointer(xt)
    return lib().mijpeg_kernel_name(C.byref(b)).decode()

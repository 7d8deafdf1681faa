"""ctypes binding of libgsr.so (include/gsr.h). No CPU fallback: if the HIP library is
missing this module raises, loudly, at first use."""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch  # imported first so that libgsr.so binds to the HIP runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
# GSR_LIB=<path> loads another build of the same ABI (A/B measurements of two source revisions)
LIB_PATH = os.environ.get("GSR_LIB") or os.path.join(_HERE, "libgsr.so")

GSR_MAX_VIEWS = 16      # include/gsr.h
GSR_ABI_VERSION = 6     # include/gsr.h
GSR_VIEW_VIEWMATRIX_T, GSR_VIEW_PROJMATRIX_T, GSR_VIEW_NO_BACKWARD, GSR_VIEW_ASYNC_STATS, GSR_VIEW_DETERMINISTIC = 1, 2, 4, 8, 16   # GsrView.flags

EXPORTS = ("gsr_forward", "gsr_forward_complete", "gsr_backward", "gsr_forward_views", "gsr_backward_views", "gsr_mark_visible", "gsr_dist2", "gsr_extract_fields", "gsr_densify_stats",
           "gsr_adam_step", "gsr_mask_compact", "gsr_gather_rows", "gsr_concat_rows",
           "gsr_profile_enable", "gsr_profile_read", "gsr_profile_reset",
           "gsr_geom_bytes", "gsr_img_bytes", "gsr_last_error", "gsr_version", "gsr_abi_version", "gsr_testing_override")


class GsrView(C.Structure):
    _fields_ = [("image_height", C.c_int32), ("image_width", C.c_int32),
                ("tanfovx", C.c_float), ("tanfovy", C.c_float),
                ("scale_modifier", C.c_float), ("sh_degree", C.c_int32),
                ("prefiltered", C.c_int32), ("debug", C.c_int32),
                ("bg", C.c_void_p), ("viewmatrix", C.c_void_p),
                ("projmatrix", C.c_void_p), ("campos", C.c_void_p),
                ("raw_activations", C.c_int32), ("flags", C.c_int32),
                ("shs_rest", C.c_void_p), ("dL_dshs_rest", C.c_void_p),
                ("grad_clear", C.c_void_p), ("grad_clear_floats", C.c_int64)]


RESIZE_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)


class GsrAlloc(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("resize", RESIZE_FN)]


class GsrAdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_int64), ("lr", C.c_double)]


class GsrGatherTensor(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("width", C.c_int32), ("reserved", C.c_int32)]


class GsrConcatTensor(C.Structure):
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("dst", C.c_void_p), ("width", C.c_int32), ("reserved", C.c_int32)]


class GsrStats(C.Structure):
    _fields_ = [("num_instances", C.c_int64), ("num_instances_ref", C.c_int64),
                ("num_visible", C.c_int64), ("max_tile_count", C.c_int64), ("bin_capacity", C.c_int64),
                ("seg_shift", C.c_int64), ("bwd_prepared", C.c_int64), ("speculated", C.c_int64), ("pending", C.c_int64)]


_lock = threading.Lock()
_lib = None


def load() -> C.CDLL:
    """Load (once) and type the library. Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
                "`python -m dreamgaussian_amd.build` (or __graft_entry__.build()); there is no "
                "CPU fallback for the rasterizer.")
        lib = C.CDLL(LIB_PATH)
        try:                                              # (a library older than the version check does not export the symbol)
            lib.gsr_abi_version.restype = C.c_int
            found = lib.gsr_abi_version()
        except AttributeError:
            found = None
        if found != GSR_ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} reports ABI version {found}, this binding is written for "
                               f"{GSR_ABI_VERSION} (include/gsr.h): rebuild with `python -m dreamgaussian_amd.build`")
        p, i32, vp = C.c_void_p, C.c_int32, C.c_void_p
        lib.gsr_forward.restype = C.c_int
        lib.gsr_forward.argtypes = [C.POINTER(GsrView), i32, i32] + [p] * 7 + [p] * 4 + \
            [GsrAlloc, GsrAlloc, GsrAlloc, C.POINTER(GsrStats), vp]
        lib.gsr_forward_complete.restype = C.c_int
        lib.gsr_forward_complete.argtypes = [C.POINTER(GsrStats)]
        lib.gsr_backward.restype = C.c_int
        lib.gsr_backward.argtypes = [C.POINTER(GsrView), i32, i32] + [p] * 7 + [p] + [p] * 3 + \
            [p] * 3 + [C.POINTER(GsrStats)] + [p] * 8 + [GsrAlloc, vp]
        lib.gsr_forward_views.restype = C.c_int
        lib.gsr_forward_views.argtypes = [C.POINTER(GsrView), i32, i32, i32] + [p] * 7 + [p] * 4 + \
            [GsrAlloc, GsrAlloc, GsrAlloc, C.POINTER(GsrStats), vp]
        lib.gsr_backward_views.restype = C.c_int
        lib.gsr_backward_views.argtypes = [C.POINTER(GsrView), i32, i32, i32] + [p] * 7 + [p] + [p] * 3 + \
            [p] * 3 + [C.POINTER(GsrStats)] + [p] * 8 + [GsrAlloc, vp]
        lib.gsr_mark_visible.restype = C.c_int
        lib.gsr_mark_visible.argtypes = [C.POINTER(GsrView), i32, p, p, vp]
        lib.gsr_dist2.restype = C.c_int
        lib.gsr_dist2.argtypes = [i32, p, p, GsrAlloc, vp]
        lib.gsr_densify_stats.restype = C.c_int
        lib.gsr_densify_stats.argtypes = [i32, p, p, p, p, p, vp]
        lib.gsr_extract_fields.restype = C.c_int
        lib.gsr_extract_fields.argtypes = [i32, p, p, p, p, i32, i32, i32, p, p, p, p, p, GsrAlloc, vp]
        lib.gsr_adam_step.restype = C.c_int
        lib.gsr_adam_step.argtypes = [i32, C.POINTER(GsrAdamTensor), i32, C.c_double, C.c_double, C.c_double, vp]
        lib.gsr_mask_compact.restype = C.c_int
        lib.gsr_mask_compact.argtypes = [i32, p, p, p, GsrAlloc, vp]
        lib.gsr_gather_rows.restype = C.c_int
        lib.gsr_gather_rows.argtypes = [i32, C.POINTER(GsrGatherTensor), i32, p, vp]
        lib.gsr_concat_rows.restype = C.c_int
        lib.gsr_concat_rows.argtypes = [i32, C.POINTER(GsrConcatTensor), i32, i32, vp]
        lib.gsr_geom_bytes.restype = C.c_size_t
        lib.gsr_geom_bytes.argtypes = [i32, i32, i32]
        lib.gsr_img_bytes.restype = C.c_size_t
        lib.gsr_img_bytes.argtypes = [i32, i32]
        lib.gsr_profile_enable.restype = C.c_int
        lib.gsr_profile_enable.argtypes = [C.c_int]
        lib.gsr_profile_reset.restype = C.c_int
        lib.gsr_profile_reset.argtypes = []
        lib.gsr_profile_read.restype = C.c_int
        lib.gsr_profile_read.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_float),
                                         C.POINTER(C.c_int)]
        lib.gsr_last_error.restype = C.c_char_p
        lib.gsr_testing_override.restype = C.c_int
        lib.gsr_testing_override.argtypes = [C.c_char_p, i32]
        lib.gsr_version.restype = C.c_char_p
        _lib = lib
        return lib


_tls = threading.local()


class Scratch:
    """GsrAlloc backed by the torch caching allocator; keeps the tensor it handed out.

    The ctypes callback object is the expensive part (a few microseconds to create, and it closes a reference cycle through
    the bound method): instances are pooled per thread -- `Scratch(dev)` takes one, `release()` hands back the tensor and
    returns the instance to the pool."""

    __slots__ = ("device", "tensor", "_cb", "alloc")

    def __new__(cls, device):
        pool = getattr(_tls, "pool", None)
        if pool:
            self = pool.pop()
        else:
            self = object.__new__(cls)
            self._cb = RESIZE_FN(self._resize)
            self.alloc = GsrAlloc(None, self._cb)
        self.device = device
        self.tensor = None
        return self

    def __init__(self, device):
        pass

    def release(self):
        """The tensor the library asked for (or None); the instance goes back to the pool and must not be used again."""
        t, self.tensor = self.tensor, None
        pool = getattr(_tls, "pool", None)
        if pool is None:
            pool = _tls.pool = []
        if len(pool) < 16:
            pool.append(self)
        return t

    def _resize(self, _ctx, nbytes):
        try:
            self.tensor = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=self.device)
            return self.tensor.data_ptr()
        except Exception:  # surfaces as "scratch allocation failed" on the C side
            self.tensor = None
            return 0


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def check(rc: int, what: str):
    if rc != 0:
        msg = load().gsr_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def profile_enable(on: bool):
    """Bracket every kernel of the calling thread's gsr_* calls with hipEvents (bench.py)."""
    load().gsr_profile_enable(1 if on else 0)


def profile_reset():
    load().gsr_profile_reset()


def profile_read(cap: int = 64) -> dict:
    """{kernel name: (total ms, launches)} since the last reset; waits for the events."""
    names = (C.c_char_p * cap)()
    ms = (C.c_float * cap)()
    cnt = (C.c_int * cap)()
    n = load().gsr_profile_read(cap, names, ms, cnt)
    return {names[i].decode(): (float(ms[i]), int(cnt[i])) for i in range(n)}

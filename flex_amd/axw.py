"""ctypes binding of include/flex_axw.h (libflex_axw.so): the GCN layer product Out = A * X * W around
the engine's SpMM (≙ run1 / run2 of the reference's cusp.cu).  No fallback: raises if the library is missing."""
from __future__ import annotations

import ctypes as C
import os

from . import binding

FLEX_AXW_AUTO, FLEX_AXW_A_XW, FLEX_AXW_AX_W = 0, 1, 2
_lib = None


def lib():
    global _lib
    if _lib is None:
        binding.lib()  # torch's HIP runtime first, then the engine, then this library on top of it
        path = os.path.join(os.path.dirname(binding.lib_path()), "libflex_axw.so")
        if not os.path.exists(path):
            raise binding.FlexError(f"{path} is missing: build it with `make -C flex_amd/csrc all`")
        L = C.CDLL(path)
        vp = C.c_void_p
        L.flex_axw_create.argtypes = [C.POINTER(vp), C.POINTER(binding._Csr), C.c_int, C.c_int, C.c_int, C.c_uint]
        L.flex_axw_run.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.flex_axw_destroy.argtypes = [vp]
        L.flex_axw_ld.argtypes = [C.c_int]
        _lib = L
    return _lib


class Axw:
    """flex_axw handle: plans A for both SpMM widths once; run() computes A @ X @ W in either order."""

    def __init__(self, a: binding.HostCsr, dim: int, c: int, device: int = 0, order: int = binding.FLEX_ORDER_CLUSTER):
        self._h = C.c_void_p()
        v = a.view()
        binding._check(lib().flex_axw_create(C.byref(self._h), C.byref(v), dim, c, device, order), "flex_axw_create")
        self.n, self.dim, self.c, self.ld = a.n, dim, c, lib().flex_axw_ld(c)

    def run(self, X, W, order: int = FLEX_AXW_AUTO, timed: bool = False):
        """X [n, dim], W [dim, c]: contiguous float32 cuda tensors.  Returns Out [n, ld] (columns >= c are zero)
        and, if timed, (gemm_ms, spmm_ms)."""
        import torch
        assert X.is_cuda and W.is_cuda and X.dtype == W.dtype == torch.float32 and X.is_contiguous() and W.is_contiguous()
        assert tuple(X.shape) == (self.n, self.dim) and tuple(W.shape) == (self.dim, self.c)
        out = torch.empty((self.n, self.ld), dtype=torch.float32, device=X.device)
        g, s = C.c_float(), C.c_float()
        binding._check(lib().flex_axw_run(self._h, order, X.data_ptr(), W.data_ptr(), out.data_ptr(),
                                          torch.cuda.current_stream(X.device).cuda_stream,
                                          C.byref(g) if timed else None, C.byref(s) if timed else None), "flex_axw_run")
        return (out, (g.value, s.value)) if timed else out

    def destroy(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().flex_axw_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

"""ctypes wrapper of the TEST-ONLY host plan interpreter (tests/interp.cpp -> tests/libsbmae_planref.so).

Used by the CPU suite to check the C++ plan builder / weight packer (shared with the product
library) against the oracle and the golden fixtures without a GPU.  Never imported by the product.
"""
import ctypes
import os
import subprocess

import numpy as np

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                    "score_based_multimodal_autoencoder_amd", "csrc")
HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libsbmae_planref.so")


def build():
    srcs = [os.path.join(CSRC, "plan.cpp"), os.path.join(HERE, "interp.cpp")]
    deps = srcs + [os.path.join(CSRC, "plan.h")]
    if os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return LIB
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I", CSRC, "-o", LIB] + srcs)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        L.sbmref_error.restype = ctypes.c_char_p
        L.sbmref_build.restype = ctypes.c_void_p
        L.sbmref_build.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int,
                                   ctypes.POINTER(ctypes.c_void_p)]
        L.sbmref_num_specs.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.sbmref_spec.restype = ctypes.c_long
        L.sbmref_spec.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
        L.sbmref_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.sbmref_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
        L.sbmref_free.argtypes = [ctypes.c_void_p]
        L.sbmref_run_aux.argtypes = [ctypes.c_void_p] * 5
        L.sbmref_run_range.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.sbmref_op_name.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
        L.sbmref_workspace.restype = ctypes.POINTER(ctypes.c_float)
        L.sbmref_workspace.argtypes = [ctypes.c_void_p]
        _lib = L
    return _lib


def _ints(v):
    return (ctypes.c_int * len(v))(*[int(x) for x in v])


def unet_cfg_ints(dim, channels, dim_mults, H, W, init_dim=0, out_dim=0, with_time_emb=1, convnext_mult=2):
    return [dim, channels, init_dim or 0, out_dim or 0, int(with_time_emb), convnext_mult, H, W, len(dim_mults)] + list(dim_mults)


def resdec_cfg_ints(variant, enc, dec, size_in, size_z, img_ch):
    v = [variant, size_in, size_z, img_ch, len(enc), len(dec)]
    for e in enc:
        v += list(e)
    for d in dec:
        v += list(d)
    return v


def unet_oa_cfg_ints(in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, channel_mult, H, W,
                     num_heads=1, z_dim=0, use_z=False):
    return [in_channels, model_channels, out_channels, num_res_blocks, num_heads, z_dim or 0, int(bool(use_z)), H, W,
            len(channel_mult)] + list(channel_mult) + [len(attention_resolutions)] + list(attention_resolutions)


def specs(kind, cfg):
    L = lib()
    c = _ints(cfg)
    out = []
    buf = ctypes.create_string_buffer(256)
    for i in range(L.sbmref_num_specs(kind, c)):
        n = L.sbmref_spec(kind, c, i, buf, 256)
        out.append((buf.value.decode(), int(n)))
    return out


class RefPlan:
    def __init__(self, kind, cfg, B, sd, z_ld=0):
        L = lib()
        self.kind, self.B = kind, B
        sp = specs(kind, cfg)
        self._keep = [np.ascontiguousarray(sd[name], dtype=np.float32) for name, _ in sp]
        for (name, n), a in zip(sp, self._keep):
            assert a.size == n, (name, a.size, n)
        ptrs = (ctypes.c_void_p * len(sp))(*[a.ctypes.data for a in self._keep])
        self.h = L.sbmref_build(kind, _ints(cfg), B, z_ld, ptrs)
        if not self.h:
            raise RuntimeError(L.sbmref_error().decode())
        info = (ctypes.c_double * 6)()
        L.sbmref_info(self.h, info)
        self.info = dict(n_ops=int(info[0]), ws_floats=int(info[1]), params_floats=int(info[2]),
                         flops_dense=info[3], flops_exec=info[4], n_gemm=int(info[5]))

    def run(self, x, t, out_shape):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.full(out_shape, np.nan, dtype=np.float32)
        tp = None
        if t is not None:
            t = np.ascontiguousarray(t, dtype=np.float32)
            tp = t.ctypes.data
        lib().sbmref_run(self.h, x.ctypes.data, tp, out.ctypes.data)
        return out

    def run_aux(self, x, t, aux, out_shape):
        x = np.ascontiguousarray(x, dtype=np.float32); t = np.ascontiguousarray(t, dtype=np.float32)
        out = np.full(out_shape, np.nan, dtype=np.float32)
        ap = None
        if aux is not None:
            aux = np.ascontiguousarray(aux, dtype=np.float32)
            ap = aux.ctypes.data
        lib().sbmref_run_aux(self.h, x.ctypes.data, t.ctypes.data, ap, out.ctypes.data)
        return out

    def run_range(self, x, t, out, first, last):
        tp = t.ctypes.data if t is not None else None
        lib().sbmref_run_range(self.h, x.ctypes.data, tp, out.ctypes.data, first, last)

    def op_name(self, i):
        buf = ctypes.create_string_buffer(256)
        lib().sbmref_op_name(self.h, i, buf, 256)
        return buf.value.decode()

    def workspace(self):
        """numpy view (no copy) of the interpreter's workspace."""
        return np.ctypeslib.as_array(lib().sbmref_workspace(self.h), shape=(self.info["ws_floats"],))

    def __del__(self):
        if getattr(self, "h", None):
            lib().sbmref_free(self.h)
            self.h = None

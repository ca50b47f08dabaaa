"""ctypes binding of the C ABI in include/sbmae.h (library csrc/libsbmae_hip.so).

There is deliberately no fallback: if the library is missing or no GPU is visible, every compute
entry point raises.  `build()` compiles the library in-tree with hipcc for gfx950.
"""
import ctypes
import os
import subprocess
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libsbmae_hip.so")
if os.environ.get("SBM_LIB"):          # A/B measurements of two builds inside ONE gpurun call (boxes differ by a few percent)
    LIB_PATH = os.path.abspath(os.environ["SBM_LIB"])
SOURCES = ["plan.cpp", "kernels.hip", "chain.hip", "rowrun.hip", "pixrun.hip", "sampler.hip", "guidance.hip", "fid.hip", "api.hip"]
HEADERS = ["plan.h", "device.h", os.path.join("..", "..", "include", "sbmae.h")]

SBM_UNET, SBM_RESDEC, SBM_ATTRDEC, SBM_UNET_OA, SBM_MLP, SBM_RESENC, SBM_ATTRENC, SBM_CLG = 0, 1, 2, 3, 4, 5, 6, 7
OP_KIND_NAMES = ["seg_gemm", "dwconv7", "linattn", "attn", "gn_res", "sinus", "nchw_to_pbc", "bilinear", "gn_final", "gn32",
                 "attn_qkv", "chain_kernel", "avgpool", "rowrun_kernel", "pixrun_kernel"]

EXPORTS = ["sbm_last_error", "sbm_abi_version", "sbm_spec_count", "sbm_spec", "sbm_model_create", "sbm_model_destroy", "sbm_model_check",
           "sbm_model_info", "sbm_model_note", "sbm_model_forward", "sbm_model_forward_aux", "sbm_model_set_aux", "sbm_model_profile", "sbm_model_profile_mix", "sbm_model_gemm_flops", "sbm_em_step", "sbm_langevin_step", "sbm_langevin_norms", "sbm_langevin_update", "sbm_clg_forward", "sbm_clg_guide",
           "sbm_fill_normal", "sbm_sampler_run", "sbm_cond_sampler_run", "sbm_guided_cond_sampler_run", "sbm_langevin_fixed_step", "sbm_langevin_run", "sbm_renoise_observed", "sbm_debug_forward_partial", "sbm_debug_forward_range", "sbm_debug_workspace_floats",
           "sbm_debug_read_workspace", "sbm_debug_op_name", "sbm_debug_profile_ops", "sbm_debug_gemm_stamps", "sbm_debug_chain_sync", "sbm_debug_chain_ctl", "sbm_debug_chain_stamps", "sbm_debug_row_stamps", "sbm_debug_pix_stamps",
           "sbm_ddpm_step", "sbm_fid_accumulate", "sbm_fid_finalize_stats", "sbm_fid_work_doubles", "sbm_fid_frechet", "sbm_debug_dgemm_tn"]


class SdeCfg(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("b0", ctypes.c_double), ("b1", ctypes.c_double), ("N", ctypes.c_int),
                ("target_snr", ctypes.c_double), ("probability_flow", ctypes.c_int)]


class CondCfg(ctypes.Structure):
    _fields_ = [("obs_mask", ctypes.c_uint), ("noise_obs", ctypes.c_int), ("n_mod", ctypes.c_int)]


class GuideJob(ctypes.Structure):
    _fields_ = [("clg", ctypes.c_void_p), ("ch1", ctypes.c_int), ("ch2", ctypes.c_int), ("id1", ctypes.c_int), ("id2", ctypes.c_int),
                ("upd_mask", ctypes.c_uint)]


class SamplerCfg(ctypes.Structure):
    _fields_ = [("pc", ctypes.c_int), ("n_steps", ctypes.c_int), ("order", ctypes.c_int), ("use_graph", ctypes.c_int),
                ("seed", ctypes.c_uint64), ("sample_offset", ctypes.c_uint64)]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def source_hash():
    """Identity of the kernels a measurement was taken on: sha256 over the library's sources and headers (the binary's own bytes are
    not reproducible across rebuilds, the sources are).  Stamped into the committed profile summaries and checked by bench.py."""
    import hashlib
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> csrc/libsbmae_hip.so (cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIB_PATH
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", LIB_PATH] + SOURCES
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB_PATH


_lib = None
_lock = threading.Lock()


def lib():
    """Load the HIP library (raises if it has not been built -- there is no CPU path)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc, gfx950). This package has no CPU or eager-PyTorch fallback.")
        L = ctypes.CDLL(LIB_PATH)
        vp, ip, i, dbl = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_double
        u64 = ctypes.c_uint64
        L.sbm_last_error.restype = ctypes.c_char_p
        L.sbm_spec_count.argtypes = [i, ip]
        L.sbm_spec.restype = ctypes.c_long
        L.sbm_spec.argtypes = [i, ip, i, ctypes.c_char_p, i]
        L.sbm_model_create.argtypes = [i, ip, i, i, ctypes.POINTER(vp), i, ctypes.POINTER(vp)]
        L.sbm_model_destroy.argtypes = [vp]
        L.sbm_model_check.argtypes = [vp, i, vp]
        L.sbm_model_info.argtypes = [vp, ctypes.POINTER(dbl)]
        L.sbm_model_note.argtypes = [vp, ctypes.c_char_p, i]
        L.sbm_model_forward.argtypes = [vp, vp, vp, vp, vp]
        L.sbm_model_forward_aux.argtypes = [vp, vp, vp, vp, vp, vp]
        L.sbm_model_set_aux.argtypes = [vp, vp]
        L.sbm_model_profile.argtypes = [vp, vp, vp, vp, i, ctypes.POINTER(dbl), ip, i]
        L.sbm_model_profile_mix.argtypes = [vp, vp, vp, vp, i, i, ctypes.POINTER(dbl), ip, i]
        L.sbm_model_gemm_flops.argtypes = [vp, ctypes.POINTER(dbl)]
        L.sbm_em_step.argtypes = [ctypes.POINTER(SdeCfg), vp, vp, vp, vp, vp, vp, i, i, u64, u64, i, i, vp]
        L.sbm_langevin_step.argtypes = [ctypes.POINTER(SdeCfg), vp, vp, vp, vp, vp, vp, vp, vp, i, i, u64, u64, i, i, vp]
        L.sbm_langevin_run.argtypes = [vp, vp, vp, ctypes.c_uint, i, i, vp, vp, u64, u64, i, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), vp]
        L.sbm_langevin_norms.argtypes = [vp, vp, vp, vp, i, i, u64, u64, i, i, vp]
        L.sbm_langevin_update.argtypes = [ctypes.POINTER(SdeCfg), vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_long, i, i, u64, u64, i, i, vp]
        L.sbm_clg_forward.argtypes = [vp, vp, vp, i, i, vp, i, vp]
        L.sbm_clg_guide.argtypes = [vp, vp, vp, i, i, i, i, ctypes.c_float, ctypes.c_uint, vp, vp, i, i, i, vp]
        L.sbm_langevin_fixed_step.argtypes = [vp, vp, vp, vp, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_uint, i, i, i, u64, u64, i, i,
                                              vp]
        L.sbm_renoise_observed.argtypes = [ctypes.POINTER(SdeCfg), vp, vp, vp, vp, ctypes.c_uint, i, i, i, vp]
        L.sbm_fill_normal.argtypes = [vp, ctypes.c_long, i, u64, u64, i, i, ctypes.c_float, vp]
        L.sbm_sampler_run.argtypes = [vp, ctypes.POINTER(SdeCfg), ctypes.POINTER(SamplerCfg), vp, vp, vp, vp, vp, i, i,
                                      ctypes.POINTER(ctypes.c_float), vp]
        L.sbm_cond_sampler_run.argtypes = [vp, ctypes.POINTER(SdeCfg), ctypes.POINTER(SamplerCfg), ctypes.POINTER(CondCfg), vp, vp, vp, vp, vp, vp, i, i,
                                           ctypes.POINTER(ctypes.c_float), vp]
        L.sbm_guided_cond_sampler_run.argtypes = [vp, ctypes.POINTER(SdeCfg), ctypes.POINTER(SamplerCfg), ctypes.POINTER(CondCfg), vp,
                                                  ctypes.POINTER(GuideJob), i, ctypes.c_float, vp, vp, vp, vp, vp, i, i, ctypes.POINTER(ctypes.c_float), vp]
        L.sbm_debug_forward_partial.argtypes = [vp, vp, vp, vp, i, vp]
        L.sbm_debug_forward_range.argtypes = [vp, vp, vp, vp, i, i, vp]
        L.sbm_debug_workspace_floats.restype = ctypes.c_long
        L.sbm_debug_workspace_floats.argtypes = [vp]
        L.sbm_debug_read_workspace.argtypes = [vp, ctypes.c_long, ctypes.c_long, vp]
        L.sbm_debug_op_name.argtypes = [vp, i, ctypes.c_char_p, i]
        L.sbm_debug_gemm_stamps.argtypes = [vp, vp, vp, i, ctypes.POINTER(ctypes.c_uint64)]
        L.sbm_debug_chain_sync.argtypes = [vp, ctypes.POINTER(ctypes.c_uint), i]
        L.sbm_debug_chain_ctl.argtypes = [vp, ctypes.c_uint, i]
        L.sbm_debug_row_stamps.argtypes = [vp, i, i, ctypes.POINTER(ctypes.c_uint64), ip, i, ip]
        L.sbm_debug_pix_stamps.argtypes = [vp, i, i, ctypes.POINTER(ctypes.c_uint64), ip, i, ip]
        L.sbm_debug_chain_stamps.argtypes = [vp, i, i, ctypes.POINTER(ctypes.c_uint64)]
        L.sbm_debug_profile_ops.argtypes = [vp, vp, vp, vp, i, ctypes.POINTER(dbl), ctypes.POINTER(dbl), i]
        L.sbm_ddpm_step.argtypes = [vp, vp, vp, vp, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, i, ctypes.c_long, u64, u64, i, vp]
        L.sbm_fid_accumulate.argtypes = [vp, ctypes.c_long, i, ctypes.c_long, vp, vp, vp]
        L.sbm_fid_finalize_stats.argtypes = [vp, vp, ctypes.c_long, i, vp, vp, vp]
        L.sbm_fid_work_doubles.restype = ctypes.c_long
        L.sbm_fid_work_doubles.argtypes = [i]
        L.sbm_fid_frechet.argtypes = [vp, vp, vp, vp, i, dbl, i, vp, ctypes.POINTER(dbl), vp]
        L.sbm_debug_dgemm_tn.argtypes = [vp, vp, vp, i, i, i, dbl, dbl, dbl, vp]
        _lib = L
        return L


def check(rc):
    if rc != 0:
        raise RuntimeError("sbmae: " + lib().sbm_last_error().decode(errors="replace"))


def ints(v):
    return (ctypes.c_int * len(v))(*[int(x) for x in v])


def specs(kind, cfg):
    """[(state_dict key, numel)] in the order sbm_model_create expects."""
    L = lib()
    c = ints(cfg)
    buf = ctypes.create_string_buffer(256)
    out = []
    for k in range(L.sbm_spec_count(kind, c)):
        n = L.sbm_spec(kind, c, k, buf, 256)
        out.append((buf.value.decode(), int(n)))
    return out


def unet_cfg_ints(dim, channels, dim_mults, H, W, init_dim=None, out_dim=None, with_time_emb=True, convnext_mult=2):
    return [dim, channels, init_dim or 0, out_dim or 0, int(bool(with_time_emb)), convnext_mult, H, W, len(dim_mults)] + \
        [int(m) for m in dim_mults]


def unet_oa_cfg_ints(in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, channel_mult, H, W,
                     num_heads=1, z_dim=0, use_z=False):
    return [in_channels, model_channels, out_channels, num_res_blocks, num_heads, z_dim or 0, int(bool(use_z)), H, W,
            len(channel_mult)] + [int(m) for m in channel_mult] + [len(attention_resolutions)] + [int(a) for a in attention_resolutions]


def resdec_cfg_ints(variant, enc, dec, size_in, size_z, img_ch):
    v = [variant, size_in, size_z, img_ch, len(enc), len(dec)]
    for e in enc:
        v += [int(a) for a in e]
    for d in dec:
        v += [int(a) for a in d]
    return v


def _require_cuda(t, what):
    import torch
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{what} must be a tensor on the GPU: this package runs only on HIP kernels (no CPU path)")


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class Engine:
    """A device-resident model for one batch size: packed weights + fused kernel plan."""

    def __init__(self, kind, cfg, batch, state_dict, in_pitch=0):
        import numpy as np
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: this package has no CPU path")
        L = lib()
        self.kind, self.cfg, self.batch = kind, list(cfg), batch
        sp = specs(kind, cfg)
        keep = []
        for name, n in sp:
            if name not in state_dict:
                raise KeyError(f"state_dict is missing '{name}'")
            a = state_dict[name]
            if isinstance(a, torch.Tensor):
                a = a.detach().to("cpu", torch.float32).contiguous().numpy()
            a = np.ascontiguousarray(a, dtype=np.float32)
            if a.size != n:
                raise ValueError(f"'{name}' has {a.size} elements, expected {n}")
            keep.append(a)
        ptrs = (ctypes.c_void_p * len(sp))(*[a.ctypes.data for a in keep])
        h = ctypes.c_void_p()
        check(L.sbm_model_create(kind, ints(cfg), batch, in_pitch, ptrs, len(sp), ctypes.byref(h)))
        self.h = h
        info = (ctypes.c_double * 10)()
        check(L.sbm_model_info(self.h, info))
        self.info = dict(n_ops=int(info[0]), workspace_bytes=int(info[1]), flops_dense=info[3], flops_exec=info[4], flops_dense_gemm=info[7],
                         n_gemm=int(info[5]), uses_pix=bool(info[8]), pix_fragment_mb=info[9])
        note = ctypes.create_string_buffer(512)
        check(L.sbm_model_note(self.h, note, 512))
        self.info["pix_note"] = note.value.decode(errors="replace")

    def forward(self, x, t, out):
        check(lib().sbm_model_forward(self.h, x.data_ptr(), t.data_ptr() if t is not None else None, out.data_ptr(),
                                      stream_ptr()))
        return out

    def forward_aux(self, x, t, aux, out):
        check(lib().sbm_model_forward_aux(self.h, x.data_ptr(), t.data_ptr(), aux.data_ptr() if aux is not None else None,
                                          out.data_ptr(), stream_ptr()))
        return out

    def profile(self, x, t, out, reps=5, sampler_mix=False):
        """Per-kernel time of one forward; sampler_mix: the launch mix of one step of the sampling loop."""
        n = len(OP_KIND_NAMES)
        ms = (ctypes.c_double * n)()
        cnt = (ctypes.c_int * n)()
        check(lib().sbm_model_profile_mix(self.h, x.data_ptr(), t.data_ptr() if t is not None else None, out.data_ptr(), reps,
                                          int(bool(sampler_mix)), ms, cnt, n))
        return {OP_KIND_NAMES[k]: dict(ms=ms[k], launches=cnt[k]) for k in range(n)}

    def gemm_flops(self, sampler_mix=False):
        """(dense GEMM FLOP per forward run by seg_gemm launches, by chain_kernel launches, by rowrun_kernel launches); sampler_mix:
        without the time path"""
        v = (ctypes.c_double * 6)()
        check(lib().sbm_model_gemm_flops(self.h, v))
        return (v[0] - v[2], v[1] - v[3], v[4]) if sampler_mix else (v[0], v[1], v[4])

    def check(self, sync=True):
        """Raise if a chain-kernel barrier timed out on this handle since the last check (then ALL results produced on it since are
        invalid).  sync=True first waits for the current stream and the model's own, so the answer covers everything enqueued."""
        check(lib().sbm_model_check(self.h, int(bool(sync)), stream_ptr()))

    def close(self):
        """Destroy the handle (waits for the device).  Raises if the tear-down failed or an unreported chain time-out was pending."""
        if getattr(self, "h", None):
            h, self.h = self.h, None
            check(lib().sbm_model_destroy(h))

    def __del__(self):
        try:
            self.close()
        except Exception as e:                      # an exception cannot leave a finaliser: say it where it can be seen
            try:
                import sys
                print(f"sbmae: error while releasing a model handle: {e}", file=sys.stderr)
            except Exception:                       # interpreter shutdown: nothing left to print with
                pass

"""Shared host-side plumbing of the drop-in nn.Modules.

Each module owns parameters under exactly the reference's `state_dict` key names (so reference
checkpoints load with `load_state_dict`), and a cache of device engines (one per batch size) that is
rebuilt whenever a parameter changes.  The forward never touches the parameters with torch ops: the
numerics run in the HIP library.
"""
import math

import torch
from torch import nn

from . import _lib

_BUFFER_LEAVES = ("running_mean", "running_var", "num_batches_tracked")


_REG_EPOCH = [0]


def _bump_registration_epoch(*_args):
    _REG_EPOCH[0] += 1


torch.nn.modules.module.register_module_parameter_registration_hook(_bump_registration_epoch)
torch.nn.modules.module.register_module_buffer_registration_hook(_bump_registration_epoch)
torch.nn.modules.module.register_module_module_registration_hook(_bump_registration_epoch)      # model.block = other_block


class ParamTreeModule(nn.Module):
    """nn.Module whose sub-module / parameter tree is generated from (dotted name, shape) entries."""

    def _build_tree(self, entries):
        entries = list(entries)
        shapes = dict(entries)
        for name, shape in entries:
            parts = name.split(".")
            mod = self
            for p in parts[:-1]:
                if p not in mod._modules:
                    mod.add_module(p, nn.Module())
                mod = mod._modules[p]
            leaf = parts[-1]
            if leaf in _BUFFER_LEAVES:
                if leaf == "num_batches_tracked":
                    mod.register_buffer(leaf, torch.tensor(0, dtype=torch.long))
                else:
                    mod.register_buffer(leaf, torch.zeros(shape) if leaf == "running_mean" else torch.ones(shape))
                continue
            t = torch.empty(shape)
            if len(shape) >= 2:                                  # conv / linear weight: PyTorch's default init
                fan_in = int(math.prod(shape[1:]))
                bound = 1.0 / math.sqrt(fan_in)
                nn.init.uniform_(t, -bound, bound)
            elif leaf == "bias" and (name[:-4] + "weight") in shapes and len(shapes[name[:-4] + "weight"]) >= 2:
                fan_in = int(math.prod(shapes[name[:-4] + "weight"][1:]))
                bound = 1.0 / math.sqrt(fan_in)
                nn.init.uniform_(t, -bound, bound)
            elif leaf == "weight":                                # norm scale
                nn.init.ones_(t)
            else:                                                 # norm shift
                nn.init.zeros_(t)
            mod.register_parameter(leaf, nn.Parameter(t))
        self._sbm_engines = {}
        self._sbm_fingerprint = None

    # ---- engine cache ------------------------------------------------------------------------
    def _fingerprint(self):
        """(storage, version) of every parameter and buffer: an engine holds a packed COPY of the weights, so any change must rebuild it.
        Walking the module tree costs ~0.7 ms for the 286 tensors of `Unet(64)` -- as much as the forward itself -- so the tensor LIST
        is cached and re-collected only when some module of the process (re)registered a parameter or buffer since (a global
        registration hook bumps `_REG_EPOCH`: assigning a new `nn.Parameter`, `register_buffer`, `.to()` / `.half()` replacing
        tensors); in-place updates (`copy_`, optimizer steps, `load_state_dict`) and `.data =` swaps show up in the cheap per-tensor
        check that is still made on every call (~60 us)."""
        d = self.__dict__
        if d.get("_sbm_tensors_epoch") == _REG_EPOCH[0]:
            # writes that bypass registration (`mod._parameters[name] = t`: parametrize / functional_call-style swaps) bump no epoch:
            # the cached list is checked against the dicts it was read from (identity of every entry, ~20 us for Unet(64))
            for dct, vals in d["_sbm_dicts"]:
                if len(dct) != len(vals) or any(a is not b for a, b in zip(dct.values(), vals)):
                    d.pop("_sbm_tensors_epoch")
                    break
        if d.get("_sbm_tensors_epoch") != _REG_EPOCH[0]:
            d["_sbm_tensors"] = list(self.parameters()) + list(self.buffers())
            d["_sbm_dicts"] = [(dct, tuple(dct.values())) for m in self.modules() for dct in (m._parameters, m._buffers, m._modules) if dct]
            d["_sbm_tensors_epoch"] = _REG_EPOCH[0]
        return tuple((p.data_ptr(), p._version) for p in d["_sbm_tensors"])

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)          # .to() / .cuda() / .float(): tensors may have been replaced without a registration
        self.__dict__.pop("_sbm_tensors_epoch", None)
        return out

    def _engine(self, key, factory):
        fp = self._fingerprint()
        if fp != self._sbm_fingerprint:
            self.release_engines()
            self._sbm_fingerprint = fp
        eng = self._sbm_engines.get(key)
        if eng is None:
            eng = factory(self.state_dict())
            self._sbm_engines[key] = eng
        return eng

    def release_engines(self):
        """Close every cached engine.  `Engine.close()` raises on a tear-down error or an unreported kernel failure: every engine is
        still closed and the cache is empty afterwards in any case (no dead handle can be fetched later); the first error is re-raised."""
        engines, self._sbm_engines, self._sbm_fingerprint = list(self._sbm_engines.values()), {}, None
        first = None
        for e in engines:
            try:
                e.close()
            except Exception as exc:           # noqa: BLE001 -- collected, re-raised below
                first = first or exc
        if first is not None:
            raise first


def as_f32_cuda(t, what):
    _lib._require_cuda(t, what)
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    return t.contiguous()


def fresh_seed(seed=None):
    """Philox key of one sampler call.  `None` draws a fresh 64-bit key from torch's default generator, so successive calls get
    independent in-loop noise (as the reference's per-call `torch.randn_like`, sde_helper2.py:47,96), `torch.manual_seed`
    reproduces a run, and the generator state advances.  Always LOCAL: a sampling helper never issues a collective (a rank-conditional
    `if rank == 0: sample(...)` inside a distributed job must not hang).  Ranks that shard ONE global batch need one common key -- the
    noise is a single stream indexed by the global sample number: `distributed.shared_seed()` agrees on it explicitly
    (`distributed.sample_sharded` is already a collective and does so) and the key is passed down as `seed=`."""
    if seed is not None:
        return int(seed) & 0xFFFFFFFFFFFFFFFF
    return int(torch.empty((), dtype=torch.int64).random_().item()) & 0xFFFFFFFFFFFFFFFF

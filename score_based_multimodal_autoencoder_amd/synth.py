"""Closed-form synthetic weights.

No trained checkpoints ship with the reference (its ``models/`` directory is empty), so tests,
``bench.py`` and ``smoke()`` fill every ``state_dict`` entry from a deterministic function of the
entry's *name and shape only*.  The GPU box can therefore regenerate the exact weights the golden
fixtures were produced with, without the reference being present.

The filler re-randomises parameters whose default init would make parity tests vacuous
(zero-initialised convs of ``unet_openai.zero_module``, fresh BatchNorm running statistics).
"""
import zlib
import numpy as np


def _rng(name, salt=0):
    return np.random.Generator(np.random.Philox(key=[zlib.crc32(name.encode()), 0x5BAE ^ salt]))


def fill(name, shape):
    """Value for state_dict entry `name` of `shape` (float32; int64 for num_batches_tracked)."""
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.asarray(100, dtype=np.int64)
    r = _rng(name)
    if leaf == "running_var":
        return r.uniform(0.5, 1.5, shape).astype(np.float32)
    if leaf == "running_mean":
        return r.uniform(-0.2, 0.2, shape).astype(np.float32)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        if name.endswith("ds_conv.weight"):
            fan_in = int(np.prod(shape[2:]))
        return (r.standard_normal(shape) * (1.0 / np.sqrt(fan_in))).astype(np.float32)
    if leaf == "weight":                 # norm scales
        return r.uniform(0.8, 1.2, shape).astype(np.float32)
    return r.uniform(-0.1, 0.1, shape).astype(np.float32)          # biases / norm shifts


def fill_state_dict(shapes):
    """shapes: mapping name -> shape (e.g. {k: v.shape for k, v in module.state_dict().items()})."""
    return {k: fill(k, s) for k, s in shapes.items()}


def fill_module_(module):
    """In-place: load the closed-form weights into a torch module (ours or the reference's)."""
    import torch
    sd = module.state_dict()
    new = {k: torch.from_numpy(np.asarray(fill(k, v.shape))).to(v.dtype).reshape(v.shape) for k, v in sd.items()}
    module.load_state_dict(new)
    return module


def noise(seed, index, shape):
    """Injected-noise tensor #index of a parity run (numpy Philox; identical here and on the GPU box)."""
    g = np.random.Generator(np.random.Philox(key=[int(seed) & 0xFFFFFFFF, int(index)]))
    return g.standard_normal(shape).astype(np.float32)

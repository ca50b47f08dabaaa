import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from score_based_multimodal_autoencoder_amd import _lib, synth
from score_based_multimodal_autoencoder_amd.unet_model import Unet
m = synth.fill_module_(Unet(dim=64, channels=5, dim_mults=(1, 2, 2, 2)).eval()).cuda()
B = 4
x = torch.from_numpy(synth.noise(9, 0, (B, 5, 8, 8))).cuda(); t = torch.full((B,), 0.5, device="cuda"); o = torch.empty_like(x)
eng = m._engine_for(B, 8, 8)
L = _lib.lib(); n = eng.info["n_ops"]; buf = ctypes.create_string_buffer(256)
nws = L.sbm_debug_workspace_floats(eng.h); host = np.empty(nws, np.float32)
prev_bad = 0
for k in range(n):
    _lib.check(L.sbm_debug_forward_range(eng.h, x.data_ptr(), t.data_ptr(), o.data_ptr(), 0, k + 1, None))
    _lib.check(L.sbm_debug_read_workspace(eng.h, 0, nws, host.ctypes.data))
    bad = int((~np.isfinite(host)).sum())
    L.sbm_debug_op_name(eng.h, k, buf, 256)
    if bad != prev_bad or k > n - 3:
        print(k, buf.value.decode()[:50], "non-finite workspace floats:", bad, flush=True)
    prev_bad = bad
    if bad and k > 45: break

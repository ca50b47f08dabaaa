#!/usr/bin/env python3
"""bench.py -- samples/sec of the latent score-model sampling path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config {2,4,5}]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Started WITHOUT a launcher (`WORLD_SIZE` unset) and with `--gpus N > 1`, the script launches itself: the parent, before it makes
any GPU call, starts N fresh child processes (one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on
127.0.0.1), relays rank 0's JSON line and exits with the worst child status.  `--dry` runs the same sharding / gather / timing
plumbing on the CPU with `gloo` and a toy sample function (no GPU, no library): the launcher's own test.

Workloads (BASELINE.json `configs`).  Default = config 2 (configs[1], and configs[2] when N > 1):
  2  PolyMNIST 5-modality unconditional sampling, 1000-step reverse SDE (VPSDE beta=(0.1,5), Euler-Maruyama predictor, pc=False),
     Unet(64, ch=5, (1,2,2,2)) over the (B,5,8,8) joint latent, then the five ResVAE decoders -> (B,3,32,32); batch 256 per GPU
  4  CelebAMaskHQ 3-modality: Unet(256, ch=3, (1,2,2,2,2)) over (B,3,16,16) latents, VPSDE beta=(0.1,20), 1000 steps, then the
     ResVAEN image / mask decoders (128x128) and the CelebAAttrNewBN attribute decoder; batch 64 per GPU
  5  CelebAMaskHQ conditional: UNetModel(3,128,3,2,(),0.1,(1,2,4,8),use_z,z_dim=512) with z_cond ~ N(0,I), otherwise as 4
each sharded over N GPUs by batch (weak scaling) through distributed.sample_sharded: ONE all-gather of the decoded outputs.

A "step" is one full pass of that path over one batch (prior -> 1000 SDE steps -> decoders -> gather).  Inputs (weights, prior,
z_cond) are resident in HBM before the timed region; the in-loop noise is drawn in-kernel (Philox).  Rank 0 prints ONE JSON line.
"""
import argparse
import hashlib
import json
import math
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FP32_MATRIX_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
N_SDE = 1000
# per-sample dense-convention FLOP: score eval, decoders
WORK = {
    2: dict(name="PolyMNIST 5-modality, 1000-step reverse SDE (VPSDE EM predictor, pc=False), batch 256 per GPU, Unet(64,ch=5,(1,2,2,2)) + 5 ResVAE decoders -> 32x32 images",
            B=256, n_mod=5, hw=8, beta=(0.1, 5.0), score_flop=158_055_936, dec_flop=5 * 243_662_848),
    4: dict(name="CelebAMaskHQ 3-modality, 1000-step reverse SDE (VPSDE EM predictor, pc=False), batch 64 per GPU, Unet(256,ch=3,(1,2,2,2,2)) on 16x16 latents + ResVAEN image/mask + CelebAAttrNewBN decoders",
            B=64, n_mod=3, hw=16, beta=(0.1, 20.0), score_flop=9_349_614_848, dec_flop=8_277_000_000 + 3_064_000_000 + 1_600_000),
    5: dict(name="CelebAMaskHQ conditional, 1000-step reverse SDE (VPSDE EM predictor, pc=False), batch 64 per GPU, UNetModel(3,128,3,2,(),0.1,(1,2,4,8),use_z,z_dim=512)+z_cond on 16x16 latents + the CelebA decoders",
            B=64, n_mod=3, hw=16, beta=(0.1, 20.0), score_flop=5_467_734_016, dec_flop=8_277_000_000 + 3_064_000_000 + 1_600_000),
    # the PolyMNIST recipe as shipped (train_poly.sh:17: 10 modalities -> Unet(128, ch=10), beta = (1, 5), --pc=1): not a BASELINE config, an
    # extra workload; `--pc` turns its corrector on in the timed pass (order predictor -> corrector, train_poly_unet_cont.py:180-186)
    10: dict(name="PolyMNIST 10-modality recipe (train_poly.sh:17), 1000-step reverse SDE (VPSDE beta=(1,5)), batch 256 per GPU, Unet(128,ch=10,(1,2,2,2)) + 10 ResVAE decoders -> 32x32 images",
             B=256, n_mod=10, hw=8, beta=(1.0, 5.0), score_flop=605_100_000, dec_flop=10 * 243_662_848),
}


def log(msg):
    print(f"[bench +{time.perf_counter() - T_START:6.1f}s] {msg}", file=sys.stderr, flush=True)


T_START = time.perf_counter()


def host_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU box gives
    one GPU's share of a big host; running one thread per HOST core would oversubscribe it badly)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = int(q) / int(p)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    if quota:
        n = min(n, max(1, int(quota)))
    elif n > 16:
        n = 16                                   # documented share of a 1-GPU box
    return int(os.environ.get("SBM_CPU_THREADS", n))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def lib_hash():
    """Identity of the kernels: hash of the library's SOURCES (reproducible across rebuilds; see _lib.source_hash)."""
    from score_based_multimodal_autoencoder_amd import _lib
    return _lib.source_hash()


def build_models(cfg, device):
    """Score net + decoders of the workload with the closed-form filler weights (no checkpoints exist)."""
    from score_based_multimodal_autoencoder_amd import arch, synth
    from score_based_multimodal_autoencoder_amd.h_vae_model import CelebAAttrNewBN
    from score_based_multimodal_autoencoder_amd.h_vae_model_copy import ResVAE, ResVAEN
    from score_based_multimodal_autoencoder_amd.unet_model import Unet
    from score_based_multimodal_autoencoder_amd.unet_openai import UNetModel
    if cfg == 2:
        score = Unet(dim=64, channels=5, dim_mults=(1, 2, 2, 2))
        decs = [ResVAE(arch.POLY_ENC, arch.POLY_DEC, 32, 64, 3) for _ in range(5)]
    elif cfg == 10:
        score = Unet(dim=128, channels=10, dim_mults=(1, 2, 2, 2))
        decs = [ResVAE(arch.POLY_ENC, arch.POLY_DEC, 32, 64, 3) for _ in range(10)]
    else:
        score = Unet(dim=256, channels=3, dim_mults=(1, 2, 2, 2, 2)) if cfg == 4 else \
            UNetModel(3, 128, 3, 2, (), 0.1, (1, 2, 4, 8), num_heads=1, use_z=True, z_dim=512)
        decs = [ResVAEN(arch.CEL_IMG_ENC, arch.CEL_IMG_DEC, 128, 256, 3), ResVAEN(arch.CEL_MASK_ENC, arch.CEL_MASK_DEC, 128, 256, 1), CelebAAttrNewBN(256)]
    score = synth.fill_module_(score.eval()).to(device)
    decs = [synth.fill_module_(d.eval()).to(device) for d in decs]
    return score, decs


def one_pass(cfg, score, decs, sde, x0, zc, seed, start, device, timing=False, pc=False):
    """prior -> N-step sampler -> decoders on this rank's rows [start, start + B)."""
    from score_based_multimodal_autoencoder_amd.sde_helper2 import uncond_sampler
    W = WORK[cfg]
    B = x0.shape[0]
    kw = {} if zc is None else {"z_cond": zc}
    if pc:
        kw.update(pc=True, order="pc" if cfg in (2, 10) else "cp")          # PolyMNIST: predictor -> corrector; CelebA evaluate(): corrector -> predictor
    r = uncond_sampler((B, W["n_mod"], W["hw"], W["hw"]), score, device, sde, x0=x0, seed=seed, sample_offset=start, return_timing=timing, **kw)
    z, ms = (r if timing else (r, None))
    sz = W["hw"] * W["hw"]
    outs = [decs[i].decoder(z[:, i, :, :].view(B, sz)) for i in range(W["n_mod"])]       # train_poly_unet_cont.py:302-303 / train_lat_celebhq_unet_cont2.py:318-319
    return z, outs, ms


def cpu_baseline(cfg, reps=3, n_steps=20):
    """The reference's CPU path on the host cores: oracle/torch_port.py (the ATen operator sequence the reference executes on a CPU),
    `reps` repetitions of `n_steps` SDE steps at the workload's batch (median), + the decoders once, extrapolated linearly to
    N = 1000 (steps are identical work).  Config 2 additionally times BASELINE config 1 (2-mod, B=16, N=100) in full."""
    from oracle import torch_port as TP
    from score_based_multimodal_autoencoder_amd import arch, synth
    cores = host_cores()
    torch.set_num_threads(cores)
    W = WORK[cfg]

    def sd_of(shapes):
        return {k: torch.from_numpy(np.asarray(v)) for k, v in synth.fill_state_dict(shapes).items()}

    out = {"unit": "samples/s", "cores": cores, "kind": "port", "cpu_model": cpu_model()}
    with torch.no_grad():
        if cfg == 2:
            B = W["B"]
            sd = sd_of(arch.unet_state_shapes(dim=64, channels=5, dim_mults=(1, 2, 2, 2)))
            dsd = sd_of(arch.resvae_state_shapes("ResVAE", arch.POLY_ENC, arch.POLY_DEC, 32, 64, 3))
            fwd = lambda x, vt: TP.unet_forward(sd, x, vt, 64, 4)
            dec = lambda xm: [TP.poly_decoder(dsd, xm[:, i].reshape(B, 64), arch.POLY_DEC) for i in range(5)]
        elif cfg == 10:
            B = 64                                                   # bounded sample: 1/4 of the batch, scaled linearly
            sd = sd_of(arch.unet_state_shapes(dim=128, channels=10, dim_mults=(1, 2, 2, 2)))
            fwd = lambda x, vt: TP.unet_forward(sd, x, vt, 128, 4)
            dec = None
        elif cfg == 4:
            B = 8                                                    # bounded sample: 1/8 of the batch, scaled linearly
            sd = sd_of(arch.unet_state_shapes(dim=256, channels=3, dim_mults=(1, 2, 2, 2, 2)))
            fwd = lambda x, vt: TP.unet_forward(sd, x, vt, 256, 5)
            dec = None
        else:
            B = 8
            kw = dict(in_channels=3, model_channels=128, out_channels=3, num_res_blocks=2, attention_resolutions=(), channel_mult=(1, 2, 4, 8), z_dim=512, use_z=True)
            sd = sd_of(dict(arch.unet_openai_entries(**kw)))
            zc = torch.from_numpy(synth.noise(3, 0, (B, 512)))
            fwd = lambda x, vt: TP.unet_openai_forward(sd, x, vt, zc)
            dec = None
        x = torch.from_numpy(synth.noise(0, 0, (B, W["n_mod"], W["hw"], W["hw"])))
        ts = torch.linspace(1.0, 1e-3, N_SDE)
        if cfg == 10:
            n_steps, reps = 5, 3
        elif cfg != 2:
            n_steps, reps = 3, 3
        log(f"cpu_baseline: {cores} threads ({out['cpu_model']}), {reps} x {n_steps} SDE steps at B={B}")
        fwd(x, torch.ones(B) * ts[0])                                                  # warm-up
        per_step = []
        for r in range(reps):
            t0 = time.perf_counter()
            for i in range(n_steps):
                vt = torch.ones(B) * ts[i]
                z = torch.randn_like(x)
                s = fwd(x, vt)
                x, xm = TP.vp_em_step(x, vt, s, z, W["beta"][0], W["beta"][1], N_SDE)
            per_step.append((time.perf_counter() - t0) / n_steps)
        t_step = statistics.median(per_step)
        t_dec = 0.0
        if dec is not None:
            t0 = time.perf_counter()
            dec(xm)
            t_dec = time.perf_counter() - t0
        total = N_SDE * t_step + t_dec
        out["value"] = B / total
        out["sample"] = (f"oracle/torch_port.py (ATen CPU ops = the reference's CPU path), B={B}" + (f" of {W['B']} (per-sample cost scaled linearly)" if B != W["B"] else "") +
                         f", median of {reps} x {n_steps} of {N_SDE} SDE steps ({t_step * 1e3:.1f} ms/step; all: {[round(p * 1e3, 1) for p in per_step]})" +
                         (f" + the decoders ({t_dec:.2f} s)" if dec is not None else " (decoders not timed: < 1 % of the pass)") + ", extrapolated linearly to 1000 steps")
        if cfg == 2:
            # BASELINE configs[0], in full: 2-mod, 32x32 images, 100-step EM, batch 16 (the reference's own CPU-runnable case)
            sd1 = sd_of(arch.unet_state_shapes(dim=64, channels=2, dim_mults=(1, 2, 2, 2)))
            x1 = torch.from_numpy(synth.noise(1, 0, (16, 2, 8, 8)))
            ts1 = torch.linspace(1.0, 1e-3, 100)
            t0 = time.perf_counter()
            for i in range(100):
                vt = torch.ones(16) * ts1[i]
                s = TP.unet_forward(sd1, x1, vt, 64, 4)
                x1, xm1 = TP.vp_em_step(x1, vt, s, torch.randn_like(x1), 0.1, 5.0, 100)
            for i in range(2):
                TP.poly_decoder(dsd, xm1[:, i].reshape(16, 64), arch.POLY_DEC)
            t1 = time.perf_counter() - t0
            out["config1_full"] = {"value": 16 / t1, "unit": "samples/s", "seconds": t1,
                                   "sample": "BASELINE configs[0] in full: 2-modality, 100-step EM, batch 16, + 2 decoders"}
    return out


def extra_measurements(cfg, score, decs, sde, x0, zc, device):
    """The rest of the path's surface, timed with the sampler's own event timing or wall clock around a synchronised region (VERDICT r2
    item 6; SURVEY 8(d): "with and without the corrector").  Config 2 only carries the PolyMNIST-specific entries."""
    from score_based_multimodal_autoencoder_amd import langevin as LG
    from score_based_multimodal_autoencoder_amd import synth
    from score_based_multimodal_autoencoder_amd.sde_helper2 import cond_sampler, uncond_sampler
    W = WORK[cfg]
    B = x0.shape[0]
    shape = (B, W["n_mod"], W["hw"], W["hw"])
    kw = {} if zc is None else {"z_cond": zc}
    out = {}

    def wall(fn, reps=1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps, r

    # predictor-corrector sampling (sde_helper2.py:124-126): two score evaluations per step
    _, ms = uncond_sampler(shape, score, device, sde, x0=x0, seed=11, pc=True, return_timing=True, **kw)
    out["pc_true"] = {"ms_per_sde_step": ms / N_SDE, "samples_per_s_loop_only": B / (ms * 1e-3), "note": "corrector (n_steps=1, snr=0.16) + predictor, fused device loop"}
    # conditional loop (train_poly_unet_cont.py:340-381 / train_lat_celebhq_unet_cont2.py:286-316): modality 0 observed, re-noised every step
    z_obs = x0.clone()
    try:
        t, _ = wall(lambda: cond_sampler(z_obs, [0], score, sde, noise_obs=True, seed=12, **kw))
        out["cond_sampler_fused"] = {"ms_per_sde_step": 1e3 * t / N_SDE, "note": "observed modality re-noised in front of every step, one captured graph"}
    except Exception as exc:                      # a signature this build does not offer must not cost the headline line
        out["cond_sampler_fused"] = {"error": str(exc)[:200]}
    if cfg == 2:
        try:
            # fid_upd10.gen_ae shape (:84-99): sigma ladder x inner iterations, the Unet called with the level index as its time input
            sig = torch.exp(torch.linspace(math.log(1.0), math.log(0.01), 50))
            n_calls = 50 * 4
            z0 = x0.reshape(B, W["n_mod"], -1)
            LG.annealed_langevin(score, z0, sig, iter_num=4, seed=13, hw=(W["hw"], W["hw"]))             # engines / graph warm
            t_dev, _ = wall(lambda: LG.annealed_langevin(score, z0, sig, iter_num=4, seed=13, hw=(W["hw"], W["hw"])))
            t_py, _ = wall(lambda: LG.annealed_langevin(score, z0, sig, iter_num=4, seed=13, hw=(W["hw"], W["hw"]), fused=False))
            out["annealed_langevin"] = {"score_calls_per_s": n_calls / t_dev, "ms_per_score_call": 1e3 * t_dev / n_calls,
                                        "python_loop_ms_per_score_call": 1e3 * t_py / n_calls,
                                        "note": "50 sigma levels x 4 iterations; device-resident loop (sbm_langevin_run) vs the Python loop over the same kernels"}
        except Exception as exc:
            out["annealed_langevin"] = {"error": str(exc)[:200]}
        try:
            # MLP-score Langevin chain (fid_upd10.calc_poly_uncond :311-339): LSMPoly64_sm over the joint latent, 200 iterations
            from score_based_multimodal_autoencoder_amd.lat_sm2_model import LSMPoly64_sm
            mlp = synth.fill_module_(LSMPoly64_sm(n_mod=W["n_mod"], size_z=W["hw"] * W["hw"]).eval()).to(device)
            zm = x0.reshape(B, W["n_mod"], -1)
            LG.mlp_langevin(mlp, zm, n_comp=200, seed=15)
            t_dev, _ = wall(lambda: LG.mlp_langevin(mlp, zm, n_comp=200, seed=15))
            t_py, _ = wall(lambda: LG.mlp_langevin(mlp, zm, n_comp=200, seed=15, fused=False))
            out["mlp_langevin"] = {"score_calls_per_s": 200 / t_dev, "ms_per_score_call": 1e3 * t_dev / 200, "python_loop_ms_per_score_call": 1e3 * t_py / 200,
                                   "note": "LSMPoly64_sm + fixed-step update, 200 iterations; device-resident loop (sbm_langevin_run) vs the Python loop over the same kernels"}
            mlp.release_engines()
        except Exception as exc:
            out["mlp_langevin"] = {"error": str(exc)[:200]}
        try:
            # classifier-guided conditional loop (sde_helper2.py:283-312, 65-94): modality 0 observed, one ClwithTime2 per (given, predicted) pair
            from score_based_multimodal_autoencoder_amd.lat_sm2_model import ClwithTime2
            mods = "".join(str(i) for i in range(W["n_mod"]))
            cl = {f"0{j}": synth.fill_module_(ClwithTime2(n_mod=2, size_z=W["hw"] * W["hw"], n_class=1).eval()).to(device) for j in range(1, W["n_mod"])}
            gkw = dict(noise_obs=True, seed=12, cl_g=cl, cl_s=1.0, all_mods=mods)
            cond_sampler(z_obs, "0", score, sde, **gkw)
            t, _ = wall(lambda: cond_sampler(z_obs, "0", score, sde, **gkw))
            t_py, _ = wall(lambda: cond_sampler(z_obs, "0", score, sde, fused_guidance=False, **gkw))
            out["guided_cond_sampler"] = {"ms_per_sde_step": 1e3 * t / N_SDE, "python_loop_ms_per_sde_step": 1e3 * t_py / N_SDE,
                                          "note": "guidance of the pairs '01' and '02' after every score evaluation; device-resident loop (sbm_guided_cond_sampler_run) vs the loop driven from Python over the same kernels"}
        except Exception as exc:
            out["guided_cond_sampler"] = {"error": str(exc)[:200]}
        try:
            # DiffuseVAE refiner (eval_lat_celeba_hq_all.py:62-122, 448): the real UNetModel(6 -> 3) at 128x128, B = 64
            from score_based_multimodal_autoencoder_amd.diffuse_vae import evaluate_diff_vae
            from score_based_multimodal_autoencoder_amd.unet_openai import UNetModel
            ref = synth.fill_module_(UNetModel(6, 128, 3, 2, (16,), 0.1, (1, 2, 2, 3, 4), num_heads=8).eval()).to(device)
            vo = torch.rand(64, 3, 128, 128, device=device)
            evaluate_diff_vae(vo, ref, device, timesteps=1, seed=14)
            t, _ = wall(lambda: evaluate_diff_vae(vo, ref, device, timesteps=4, seed=14))
            out["refiner"] = {"ms_per_ddpm_step": 1e3 * t / 4, "batch": 64, "note": "UNetModel(6,128,3,2,(16,),0.1,(1,2,2,3,4),num_heads=8) at 128x128 + fused ancestral update"}
            ref.release_engines()
        except Exception as exc:
            out["refiner"] = {"error": str(exc)[:200]}
    return out


def committed_profile(name):
    """A committed rocprofv3 summary of THIS build (profiles/<name>), or None when its recorded library hash is not the hash of the
    library this run loaded (the kernels changed since it was taken: the number would be stale)."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fh:
            j = json.load(fh)
    except Exception as exc:
        log(f"no {name}: {exc}")
        return None
    if j.get("lib_hash") != lib_hash():
        log(f"{name} was taken with library {j.get('lib_hash')}, this run loads {lib_hash()}: not used")
        return None
    return j


def self_launch(n, argv):
    """`bench.py --gpus N` without a launcher: N child processes, one rank each.  The parent never touches the GPU (importing torch
    does not initialise it), so no process that has initialised a GPU is ever replaced or forked."""
    import socket
    import subprocess
    import tempfile
    import threading
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        logs.append(None if r == 0 else tempfile.TemporaryFile())          # the other ranks' stdout + stderr are kept and shown if anything fails
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE if r == 0 else logs[r],
                                      stderr=None if r == 0 else subprocess.STDOUT))
    # rank 0's pipe is drained by a thread while ALL children are polled: the first rank that fails (or the overall time limit) ends
    # its siblings at once, so nobody sits in a rendezvous holding a GPU until the collective time-out
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + float(os.environ.get("SBM_BENCH_TIMEOUT_S", "3000"))
    failed = None
    while True:
        rcs = [p.poll() for p in procs]                     # every child, every round (no short-circuit)
        if all(rc is not None for rc in rcs):
            break
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad or time.monotonic() > deadline:
            failed = f"rank {bad[0][0]} exited with {bad[0][1]}" if bad else "overall time limit reached"
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_kill = time.monotonic() + 10
            while any(p.poll() is None for p in procs) and time.monotonic() < t_kill:
                time.sleep(0.1)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10)
    sys.stdout.write(b"".join(c for c in chunks if c).decode())
    sys.stdout.flush()
    if failed or any(rcs):
        print(f"bench.py: {failed or 'a rank failed'}; rank exit codes {rcs} (negative = killed by that signal)", file=sys.stderr)
        for r in range(1, n):
            logs[r].seek(0)
            tail = logs[r].read().decode(errors="replace")[-2000:]
            if tail.strip():
                print(f"---- rank {r} output (tail) ----\n{tail}", file=sys.stderr)
        return 1
    return 0


def dry_run(args):
    """The launcher / sharding / gather / max-over-ranks timing path with no GPU: gloo on the CPU, a toy sample function of the
    GLOBAL row indices.  Rank 0 prints one JSON line whose n_gpus is the world size the process group actually has."""
    from score_based_multimodal_autoencoder_amd import distributed as D
    if os.environ.get("SBM_BENCH_DRY_FAIL_RANK") == os.environ.get("RANK", "0"):
        sys.exit(3)                               # the launcher's test: a rank that dies at start-up must take its siblings down
    rank, world = D.init_from_env("gloo") if int(os.environ.get("WORLD_SIZE", "1")) > 1 else (0, 1)
    B = 8

    def sample_fn(first, count):
        rows = torch.arange(first, first + count, dtype=torch.float32)
        return rows[:, None] * torch.ones(1, 3), rows[:, None, None] * torch.ones(1, 2, 2) + 0.5

    def barrier():
        if world > 1:
            torch.distributed.barrier()
    for _ in range(args.warmup):
        D.sample_sharded(sample_fn, world * B)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        a, b = D.sample_sharded(sample_fn, world * B)
    barrier()
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    ok = bool(torch.equal(a[:, 0], torch.arange(world * B, dtype=torch.float32)) and torch.equal(b[:, 1, 1], torch.arange(world * B, dtype=torch.float32) + 0.5))
    if rank == 0:
        print(json.dumps({"dry": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "gathered_rows": int(a.shape[0]), "rows_in_global_order": ok,
                          "ms_per_step": 1e3 * float(tmax.item()) / max(args.steps, 1), "scaling": "weak"}), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=(2, 4, 5, 10), help="BASELINE config 2 / 4 / 5; 10 = the PolyMNIST 10-modality recipe of train_poly.sh (extra workload)")
    ap.add_argument("--pc", action="store_true", help="predictor-corrector sampling in the timed pass (what the recipes run: train_poly.sh:17, train_cel.sh:8): two score evaluations per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=20)
    ap.add_argument("--cpu-reps", type=int, default=3)
    ap.add_argument("--dry", action="store_true", help="CPU / gloo rehearsal of the launcher and the sharded path (no GPU)")
    ap.add_argument("--extra", action="store_true", help="also time the rest of the surface (corrector on, conditional loops, annealed Langevin): an `extra` block")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))       # BEFORE any GPU call of this process
    if args.dry:
        sys.exit(dry_run(args))
    cfg = args.config
    W = WORK[cfg]
    B = W["B"]

    from score_based_multimodal_autoencoder_amd import distributed as D
    from score_based_multimodal_autoencoder_amd.sde_helper2 import VPSDE
    rank, world = D.init_from_env("nccl") if int(os.environ.get("WORLD_SIZE", "1")) > 1 else (0, 1)
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}; the line reports n_gpus = {world}", file=sys.stderr)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    from score_based_multimodal_autoencoder_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):          # the in-tree library normally travels with the tree; compile it (hipcc, gfx950) if it did not
        if local == 0:
            _lib.build(verbose=True)
        if world > 1:
            torch.distributed.barrier()
    log(f"rank {rank}/{world}: building models for config {cfg}")
    score, decs = build_models(cfg, device)
    sde = VPSDE(W["beta"][0], W["beta"][1], N_SDE)
    start = rank * B
    chw = W["n_mod"] * W["hw"] * W["hw"]
    # prior (and z_cond) resident in HBM before the timed region (Philox, keyed by the global sample index)
    x0 = torch.empty((B, W["n_mod"], W["hw"], W["hw"]), device=device)
    _lib.check(_lib.lib().sbm_fill_normal(x0.data_ptr(), chw, B, 1234, start, 0, 255, 1.0, None))
    zc = None
    if cfg == 5:
        zc = torch.empty((B, 512), device=device)
        _lib.check(_lib.lib().sbm_fill_normal(zc.data_ptr(), 512, B, 1235, start, 0, 254, 1.0, None))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def job(seed):
        # the gloo-tested helper IS the benched path: this rank's rows, then ONE all-gather (RCCL over xGMI) of the packed outputs
        def sample_fn(first, count):
            assert first == start and count == B
            _, outs, _ = one_pass(cfg, score, decs, sde, x0, zc, seed, first, device, pc=args.pc)
            return tuple(outs)
        return D.sample_sharded(sample_fn, world * B)

    for w in range(args.warmup):
        job(100 + w)
        log(f"warm-up pass {w} done")
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        out = job(200 + k)
    barrier()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(tmax.item())
    assert all(torch.isfinite(o).all() for o in out) and out[0].shape[0] == world * B
    log(f"timed {args.steps} passes in {elapsed:.3f} s")

    if rank == 0:
        value = world * B * args.steps / elapsed
        flop_per_sample = W["score_flop"] * N_SDE * (2 if args.pc else 1) + W["dec_flop"]
        line = {
            "metric": "samples/sec (1000-step reverse SDE, PolyMNIST 5-mod)" if cfg == 2 else
                      ("samples/sec (1000-step reverse SDE, PolyMNIST 10-mod recipe)" if cfg == 10 else f"samples/sec (1000-step reverse SDE, BASELINE config {cfg})"),
            "value": value, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": W["name"] + (" -- predictor-corrector (pc=True, n_steps=1, snr=0.16): two score evaluations per step" if args.pc else ""),
                       "baseline_config": cfg if cfg != 10 else None, "global_batch": world * B, "sde_steps": N_SDE, "pc": bool(args.pc),
                       "parallelism": f"batch-shard x{world}, 1 all-gather"},
        }
        # ---- roofline of the dominant GEMM kernel, measured live with HIP events riding on each kernel's own dispatch ----
        eng = score._engine_for(B, W["hw"], W["hw"]) if cfg != 5 else score._engine_for(B, W["hw"], W["hw"], True)
        xs = x0.clone()
        tt = torch.ones(B, device=device) * 0.5
        so = torch.empty_like(xs)
        if cfg == 5:
            _lib.check(_lib.lib().sbm_model_set_aux(eng.h, zc.data_ptr()))
            eng.forward_aux(xs, tt, zc, so)
        prof = eng.profile(xs, tt, so, reps=20 if cfg == 2 else 5, sampler_mix=True)      # the launches one SDE step of the timed loop makes
        log("per-kernel profile done")
        f_seg, f_chain, f_row = eng.gemm_flops(sampler_mix=True)
        uses_pix = bool(eng.info.get("uses_pix"))
        kern = {}
        for name, label, fl in (("seg_gemm", "seg_gemm<BM,BN> (fp32 MFMA implicit-GEMM conv, one launch per layer)", f_seg),
                                ("pixrun_kernel", "pixrun_kernel (every level from 8x8 down as runs of sample-owned workgroups, one workgroup per sample: activations in LDS, weights streamed "
                                                  "from L2 in MFMA fragment order through an AGPR ring; fp32 MFMA 16x16x4 / 4x4x1 + dwconv/attention/GroupNorm glue)", f_chain if uses_pix else 0.0),
                                ("chain_kernel", "chain_kernel (legacy: XCD-resident run of the 4x4 and 2x2 layers)", 0.0 if uses_pix else f_chain),
                                ("rowrun_kernel", "rowrun_kernel (legacy: the 1x1 level, four samples per workgroup)", f_row)):
            k = prof.get(name, dict(ms=0.0, launches=0))
            if k["launches"] and fl > 0:
                kern[name] = dict(kernel=label, launches_per_forward=k["launches"], avg_launch_us=1e3 * k["ms"] / k["launches"], ms_per_forward=k["ms"],
                                  dense_gemm_flop_per_forward=fl, achieved=fl / (k["ms"] * 1e-3) / 1e12)
        dom = max(kern, key=lambda n: kern[n]["ms_per_forward"])
        # HBM-side traffic per launch and the matrix pipe's duty cycle from the committed rocprofv3 --pmc passes of THIS library build
        # (separate FETCH_SIZE / WRITE_SIZE / SQ passes over eager forwards, tools/traffic_probe.py; FETCH_SIZE doubled as
        # MI355X_MICROARCH.md prescribes for gfx950; unit KB).  Null when the profile was taken with a different build.
        traffic = mfma_busy = None
        tag = f"r04_c{cfg}"
        tj = committed_profile(f"{tag}_traffic.json")
        if tj:
            rows = [v for k, v in tj["kernels"].items() if dom in k]
            if rows:
                fetch = sum(r["FETCH_SIZE"]["avg_per_launch_raw"] * r["FETCH_SIZE"]["launches"] for r in rows)
                write = sum(r["WRITE_SIZE"]["avg_per_launch_raw"] * r["WRITE_SIZE"]["launches"] for r in rows)
                n = sum(r["FETCH_SIZE"]["launches"] for r in rows)
                traffic = (2.0 * fetch + write) * 1024.0 / n
        sj = committed_profile(f"{tag}_sq_counters.json")
        if sj:
            rows = [v for k, v in sj["kernels"].items() if dom in k and "mfma_busy_frac" in v]
            if rows:
                mfma_busy = sum(r["mfma_busy_frac"] * r["GRBM_GUI_ACTIVE"] * r["launches"] for r in rows) / sum(r["GRBM_GUI_ACTIVE"] * r["launches"] for r in rows)
        _, _, loop_ms = one_pass(cfg, score, decs, sde, x0, zc, 7, start, device, timing=True, pc=args.pc)
        fwd_ms = sum(v["ms"] for v in prof.values())
        line["roofline"] = {
            "bound": "mfma", "achieved": kern[dom]["achieved"], "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": kern[dom]["achieved"] / FP32_MATRIX_PEAK_TFLOPS, "traffic": traffic,
            "mfma_busy_frac": mfma_busy,
            "frac_note": "frac = dense-convention GEMM FLOP (zero-padding taps counted, SURVEY 8(d)) / launch time / fp32-MFMA peak; mfma_busy_frac = "
                         "SQ_VALU_MFMA_BUSY_CYCLES / (launch cycles x 1024 SIMDs): the EXECUTED share (padding taps skipped)",
            "traffic_note": f"bytes per launch at the L2's memory side (Infinity-Cache hits included), rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, profiles/{tag}_traffic.json",
            "kernel": kern[dom]["kernel"], "launches_per_forward": kern[dom]["launches_per_forward"],
            "avg_launch_us": kern[dom]["avg_launch_us"], "dense_flop_per_launch": kern[dom]["dense_gemm_flop_per_forward"] / kern[dom]["launches_per_forward"],
            "gemm_kernels": {n: {k: (round(v, 4) if isinstance(v, float) and k != "dense_gemm_flop_per_forward" else v) for k, v in d.items() if k != "kernel"} for n, d in kern.items()},
            "all_kernels_ms_per_forward": fwd_ms, "loop_ms_per_sde_step": loop_ms / N_SDE,
            "whole_job_frac_of_fp32_mfma_peak": value / world * flop_per_sample / 1e12 / FP32_MATRIX_PEAK_TFLOPS,
            "per_kernel_ms": {k: round(v["ms"], 4) for k, v in prof.items() if v["launches"]},
            "lib_hash": lib_hash(),
        }
        if uses_pix and "pixrun_kernel" in kern:
            # the sample-owned kernel is bounded by two things at once: the matrix pipe on the 8x8 / 4x4 levels and the L2 -> CU weight stream on
            # the 2x2 / 1x1 levels (every workgroup reads every weight byte once per launch)
            mb = eng.info["pix_fragment_mb"]
            line["roofline"]["pixrun_weight_stream"] = {
                "fragment_mb_per_workgroup": mb, "workgroups": B,
                "l2_to_cu_GBps_per_workgroup": mb * 1e6 / (kern["pixrun_kernel"]["ms_per_forward"] * 1e-3) / 1e9,
                "l2_aggregate_TBps": B * mb * 1e6 / (kern["pixrun_kernel"]["ms_per_forward"] * 1e-3) / 1e12,
                "note": "measured ceiling of this access pattern (tools/pixgemm_probe.hip, memory-only): 118 GB/s per workgroup, 30 TB/s aggregate"}
        if world == 1 and args.extra:
            line["extra"] = extra_measurements(cfg, score, decs, sde, x0, zc, device)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, args.cpu_reps, args.cpu_steps)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()                  # rank 0 profiles for a few seconds: nobody tears the group down under it
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

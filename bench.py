#!/usr/bin/env python3
"""bench.py -- samples/sec of the latent score-model sampling path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]/[2]): PolyMNIST 5-modality unconditional sampling, 1000-step reverse
SDE (VPSDE beta=(0.1,5), Euler-Maruyama predictor, pc=False), Unet(64, ch=5, (1,2,2,2)) score net over the
(B,5,8,8) joint latent, then the five ResVAE decoders -> (B,3,32,32) images; batch 256 per GPU,
sharded over N GPUs (weak scaling) with ONE all-gather of the decoded images at the end.

A "step" is one full pass of that path over one batch (prior -> 1000 SDE steps -> 5 decoders ->
gather).  Inputs (weights, prior) are resident in HBM before the timed region; the in-loop noise is
drawn in-kernel (Philox).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FP32_MATRIX_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
N_MOD, N_SDE, B_PER_GPU = 5, 1000, 256
BETA = (0.1, 5.0)


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


def build_models(device):
    from score_based_multimodal_autoencoder_amd import arch, synth
    from score_based_multimodal_autoencoder_amd.h_vae_model_copy import ResVAE
    from score_based_multimodal_autoencoder_amd.unet_model import Unet
    score = synth.fill_module_(Unet(dim=64, channels=N_MOD, dim_mults=(1, 2, 2, 2)).eval()).to(device)
    # one decoder per modality; synthetic weights (same closed-form filler for each: no checkpoints exist)
    decs = [synth.fill_module_(ResVAE(arch.POLY_ENC, arch.POLY_DEC, 32, 64, 3).eval()).to(device) for _ in range(N_MOD)]
    return score, decs


def one_pass(score, decs, sde, x0, seed, start, device, timing=False):
    from score_based_multimodal_autoencoder_amd.sde_helper2 import uncond_sampler
    B = x0.shape[0]
    r = uncond_sampler((B, N_MOD, 8, 8), score, device, sde, x0=x0, seed=seed, sample_offset=start, return_timing=timing)
    z, ms = (r if timing else (r, None))
    imgs = [decs[i].decoder(z[:, i, :, :].view(B, 64)) for i in range(N_MOD)]        # train_poly_unet_cont.py:302-303
    return z, imgs, ms


def cpu_baseline(n_steps=10):
    """The reference's CPU path on the host cores: oracle/torch_port.py (same ATen operator sequence as the
    reference on CPU), B=256, n_steps SDE steps + the 5 decoders, extrapolated linearly to 1000 steps."""
    from oracle import torch_port as TP
    from score_based_multimodal_autoencoder_amd import arch, synth
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: {cores} threads, {n_steps} SDE steps at B={B_PER_GPU}")
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.fill_state_dict(
        arch.unet_state_shapes(dim=64, channels=N_MOD, dim_mults=(1, 2, 2, 2))).items()}
    dsd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.fill_state_dict(
        arch.resvae_state_shapes("ResVAE", arch.POLY_ENC, arch.POLY_DEC, 32, 64, 3)).items()}
    x = torch.from_numpy(synth.noise(0, 0, (B_PER_GPU, N_MOD, 8, 8)))
    ts = torch.linspace(1.0, 1e-3, N_SDE)
    with torch.no_grad():
        TP.unet_forward(sd, x, torch.ones(B_PER_GPU) * ts[0], 64, 4)                  # warm-up
        t0 = time.perf_counter()
        for i in range(n_steps):
            vt = torch.ones(B_PER_GPU) * ts[i]
            z = torch.randn_like(x)
            s = TP.unet_forward(sd, x, vt, 64, 4)
            x, xm = TP.vp_em_step(x, vt, s, z, BETA[0], BETA[1], N_SDE)
        t_step = (time.perf_counter() - t0) / n_steps
        t0 = time.perf_counter()
        for i in range(N_MOD):
            TP.poly_decoder(dsd, xm[:, i].reshape(B_PER_GPU, 64), arch.POLY_DEC)
        t_dec = time.perf_counter() - t0
    total = N_SDE * t_step + t_dec
    return {"value": B_PER_GPU / total, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"oracle/torch_port.py (ATen CPU ops = the reference's CPU path), B={B_PER_GPU}, {n_steps} of {N_SDE} "
                      f"SDE steps ({t_step * 1e3:.1f} ms/step) + 5 decoders ({t_dec:.2f} s), extrapolated linearly"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=10)
    args = ap.parse_args()

    from score_based_multimodal_autoencoder_amd import distributed as D
    from score_based_multimodal_autoencoder_amd.sde_helper2 import VPSDE
    rank, world = D.init_from_env("nccl") if int(os.environ.get("WORLD_SIZE", "1")) > 1 else (0, 1)
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    from score_based_multimodal_autoencoder_amd import _lib as _L
    if not os.path.exists(_L.LIB_PATH):          # the in-tree library normally travels with the tree; compile it (hipcc, gfx950) if it did not
        if local == 0:
            _L.build(verbose=True)
        if world > 1:
            torch.distributed.barrier()
    log(f"rank {rank}/{world}: building models")
    score, decs = build_models(device)
    sde = VPSDE(BETA[0], BETA[1], N_SDE)
    start = rank * B_PER_GPU
    from score_based_multimodal_autoencoder_amd import _lib
    # prior resident in HBM before the timed region (Philox, keyed by the global sample index)
    x0 = torch.empty((B_PER_GPU, N_MOD, 8, 8), device=device)
    _lib.check(_lib.lib().sbm_fill_normal(x0.data_ptr(), N_MOD * 64, B_PER_GPU, 1234, start, 0, 255, 1.0, None))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def job(seed):
        z, imgs, _ = one_pass(score, decs, sde, x0, seed, start, device)
        flat = torch.cat([i.reshape(B_PER_GPU, -1) for i in imgs], dim=1)
        return D.all_gather_cat(flat)                 # the single collective (RCCL all-gather over xGMI)

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
    assert torch.isfinite(out).all()
    log(f"timed {args.steps} passes in {elapsed:.3f} s")

    if rank == 0:
        value = world * B_PER_GPU * args.steps / elapsed
        line = {
            "metric": "samples/sec (1000-step reverse SDE, PolyMNIST 5-mod)", "value": value, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "PolyMNIST 5-modality, 1000-step reverse SDE (VPSDE EM predictor, pc=False), "
                                   "batch 256 per GPU, Unet(64,ch=5,(1,2,2,2)) + 5 ResVAE decoders -> 32x32 images",
                       "global_batch": world * B_PER_GPU, "sde_steps": N_SDE, "parallelism": f"batch-shard x{world}, 1 all-gather"},
        }
        # ---- roofline of the dominant kernel (seg_gemm), measured live with HIP events on its stream ----
        eng = score._engine_for(B_PER_GPU, 8, 8)
        xs = x0.clone()
        tt = torch.ones(B_PER_GPU, device=device) * 0.5
        so = torch.empty_like(xs)
        prof = eng.profile(xs, tt, so, reps=20, sampler_mix=True)      # the launches one SDE step of the timed loop makes
        log("per-kernel profile done")
        f_seg, f_chain = eng.gemm_flops(sampler_mix=True)
        kern = {}
        for name, label, fl in (("seg_gemm", "seg_gemm<BM,BN> (fp32 MFMA implicit-GEMM conv, one launch per 8x8 layer)", f_seg),
                                ("chain_kernel", "chain_kernel (XCD-resident run of the 4x4/2x2/1x1 layers: fp32 MFMA GEMMs + dwconv/attention/GroupNorm glue)", f_chain)):
            k = prof.get(name, dict(ms=0.0, launches=0))
            if k["launches"]:
                kern[name] = dict(kernel=label, launches_per_forward=k["launches"], avg_launch_us=1e3 * k["ms"] / k["launches"], ms_per_forward=k["ms"],
                                  dense_gemm_flop_per_forward=fl, achieved=fl / (k["ms"] * 1e-3) / 1e12)
        dom = max(kern, key=lambda n: kern[n]["ms_per_forward"])
        # HBM-side traffic per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate runs of
        # tools/traffic_probe.py; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; unit KB)
        traffic = None
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_traffic.json")) as fh:
                tj = json.load(fh)
            rows = [v for k, v in tj.items() if dom in k]
            fetch = sum(r["FETCH_SIZE"]["avg_per_launch_raw"] * r["FETCH_SIZE"]["launches"] for r in rows)
            write = sum(r["WRITE_SIZE"]["avg_per_launch_raw"] * r["WRITE_SIZE"]["launches"] for r in rows)
            n = sum(r["FETCH_SIZE"]["launches"] for r in rows)
            traffic = (2.0 * fetch + write) * 1024.0 / n
        except Exception as exc:                       # the profile file is optional
            log(f"no PMC traffic summary: {exc}")
        _, _, loop_ms = one_pass(score, decs, sde, x0, 7, start, device, timing=True)
        fwd_ms = sum(v["ms"] for v in prof.values())
        line["roofline"] = {
            "bound": "mfma", "achieved": kern[dom]["achieved"], "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": kern[dom]["achieved"] / FP32_MATRIX_PEAK_TFLOPS, "traffic": traffic,
            "traffic_note": "bytes per launch at the L2's memory side (Infinity-Cache hits included), rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_traffic.json",
            "kernel": kern[dom]["kernel"], "launches_per_forward": kern[dom]["launches_per_forward"],
            "avg_launch_us": kern[dom]["avg_launch_us"], "dense_flop_per_launch": kern[dom]["dense_gemm_flop_per_forward"] / kern[dom]["launches_per_forward"],
            "gemm_kernels": {n: {k: (round(v, 4) if isinstance(v, float) and k != "dense_gemm_flop_per_forward" else v) for k, v in d.items() if k != "kernel"} for n, d in kern.items()},
            "all_kernels_ms_per_forward": fwd_ms, "loop_ms_per_sde_step": loop_ms / N_SDE,
            "whole_job_frac_of_fp32_mfma_peak": value / world * (158_055_936 * N_SDE + 5 * 243_662_848) / 1e12 / FP32_MATRIX_PEAK_TFLOPS,
            "per_kernel_ms": {k: round(v["ms"], 4) for k, v in prof.items() if v["launches"]},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_steps)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

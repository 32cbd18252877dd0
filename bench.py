#!/usr/bin/env python3
"""bench.py -- images/sec of DiffusionPipeline.sample() on MI355X (BASELINE.json metric).

One "step" = one full `sample()` of the workload: cfg2 = 16 images per GPU, latent (8,32,32) -> 256x256,
150 DDIM iterations (eta=1), unconditional, published architecture (UNet 194 M params + VAE), fp32, synthetic
seeded weights, device Philox noise, VAE decode and the image all-gather INCLUDED; weights resident in HBM.
N>1: one process per GPU (torch.distributed.run), batch rows sharded (weak scaling: 16 images per GPU),
no collective in the loop, one RCCL all-gather of the images per step.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch

PEAK_FP32_TFLOPS = 157.3          # MI355X fp32 matrix/vector peak (MI355X_MICROARCH.md)
PEAK_BF16_TFLOPS = 16 * 157.3      # dense bf16 MFMA peak (= 2516.8; "~2.5 PF dense", same guide): the split-mode conv kernel runs on it
GFLOP_PER_IMAGE_CFG2 = 7743.2     # SURVEY §8d / BASELINE.md: 150 x 51.202 (UNet) + 62.923 (VAE decode)
WORKLOADS = {
    # name: (per-GPU batch, latent, ddim steps, use_ddim, num_classes, guidance)
    "cfg2": dict(batch=16, latent=(8, 32, 32), steps=150, use_ddim=True, classes=None, guidance=1.0),
    "cfg3_g1": dict(batch=16, latent=(8, 32, 32), steps=150, use_ddim=True, classes=3, guidance=1.0),
    "cfg3_g8": dict(batch=16, latent=(8, 32, 32), steps=150, use_ddim=True, classes=3, guidance=8.0),
    "cfg4": dict(batch=8, latent=(8, 32, 32), steps=1000, use_ddim=False, classes=None, guidance=1.0),
    "cfg5": dict(batch=8, latent=(8, 64, 64), steps=150, use_ddim=True, classes=None, guidance=1.0),
}


def build_pipeline(dev, classes):
    import medfusion_amd as M
    from oracle import restate as R   # configs only (kwargs dicts) ...
    from oracle import synth as S     # ... and the deterministic synthetic weight fill (inputs, not compute)
    from tests.util import to_product_kwargs

    from medfusion_amd.utils import no_init

    with no_init():  # every tensor is overwritten by the synthetic fill below
        pipe = M.DiffusionPipeline(M.GaussianNoiseScheduler, M.UNet, None, R.published_scheduler_kwargs(),
                                   to_product_kwargs(R.published_unet_kwargs(classes)), estimator_objective="x_T", clip_x0=False)
        pipe.latent_embedder = M.VAE(**R.published_vae_kwargs(8))
    S.synth_state_dict(pipe.noise_estimator, "published.unet.")
    S.synth_state_dict(pipe.latent_embedder, "published.vae.")
    return pipe.to(dev).eval()


def cpu_baseline(classes):
    """The oracle (CPU restatement of the reference, kind='port') on this box's host cores, bounded sample:
    2 timed UNet forwards at B=4 and 1 VAE decode at B=1 -> extrapolated cfg2 images/s."""
    from oracle import restate as R
    from oracle import synth as S

    cores = min(32, os.cpu_count() or 1)  # ATen's CPU convs scale poorly past ~32 threads (256 threads: 50x slower, measured)
    torch.set_num_threads(cores)
    unet = R.UNet(**R.published_unet_kwargs(classes)).eval()
    vae = R.VAE(**R.published_vae_kwargs(8)).eval()
    S.synth_state_dict(unet, "published.unet.")
    S.synth_state_dict(vae, "published.vae.")
    x = S.synth_input("cpu_x", (4, 8, 32, 32))
    t = torch.full((4,), 500)
    with torch.no_grad():
        unet(x, t)  # warm-up
        t0 = time.perf_counter()
        for _ in range(2):
            unet(x, t)
        t_unet = (time.perf_counter() - t0) / 2
        z = x[:1]
        vae.decode(z)
        t0 = time.perf_counter()
        vae.decode(z)
        t_dec = time.perf_counter() - t0
    ips = 4.0 / (150 * t_unet + 4 * t_dec)
    return {"value": round(ips, 5), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle (CPU restatement, torch fp32, {cores} threads): 2 UNet forwards at B=4 ({t_unet:.3f} s each) + 1 VAE decode at B=1 "
                      f"({t_dec:.3f} s), extrapolated to 150 steps"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg2", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch override")
    ap.add_argument("--ddim-steps", type=int, default=None, help="override the number of denoise iterations (non-headline)")
    ap.add_argument("--graph", action="store_true", help="replay the denoise iteration as a captured hipGraph (default for cfg4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--conv-precision", type=int, default=None, choices=[0, 1, 2, 4],
                    help="arithmetic of the conv kernel (MF_CONV_*): 1 = fp32 via exact 3 x bf16 split (default), 0 = fp32 MFMA, 2 = split + chunk sums, "
                         "4 = opt-in REDUCED precision (bf16 operands): not the headline metric")
    ap.add_argument("--no-alt-path", action="store_true", help="skip the extra timed step on the other conv arithmetic")
    args = ap.parse_args()

    import medfusion_amd as M
    from medfusion_amd import dist as D
    from medfusion_amd import kernels as K
    import torch.distributed as dist

    from medfusion_amd import blocks as BLK
    if args.conv_precision is not None:
        BLK.CONV_PRECISION = args.conv_precision
    prec = BLK.CONV_PRECISION
    rank, local, world = D.init_from_env()
    assert world == args.gpus or world == 1, f"launched with WORLD_SIZE={world} but --gpus {args.gpus}"
    dev = torch.device("cuda", local if world > 1 else 0)
    torch.cuda.set_device(dev)
    wl = dict(WORKLOADS[args.workload])
    if args.batch:
        wl["batch"] = args.batch
    if args.ddim_steps:
        wl["steps"] = args.ddim_steps
    B, n_global = wl["batch"], wl["batch"] * world
    pipe = build_pipeline(dev, wl["classes"])
    cond = (torch.arange(n_global, device=dev) % wl["classes"]) if wl["classes"] else None
    kw = dict(steps=wl["steps"], use_ddim=wl["use_ddim"])
    if args.graph or args.workload == "cfg4":
        kw["use_graph"] = True
    if cond is not None:
        kw.update(guidance_scale=wl["guidance"], un_cond=None)

    def one_step(seed):
        return D.sample_sharded(pipe, n_global, wl["latent"], condition=cond, noise=M.PhiloxDeviceNoise(seed), **kw)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        img = one_step(1000 + w)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        img = one_step(k)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert img.shape[0] == n_global and bool(torch.isfinite(img).all())
    ips = n_global * args.steps / dt

    roof = None
    if not args.no_roofline and rank == 0:
        # live launch timing of the dominant kernel (conv_igemm): one more full step of the SAME workload with every launch
        # bracketed by hipEvents on its stream (mf_prof_*); rocprofv3 --kernel-trace of this command sees the same mix
        with K.prof() as p:
            pipe.sample(B, wl["latent"], condition=None if cond is None else cond[:B], noise=M.PhiloxDeviceNoise(7), steps=wl["steps"],
                        use_ddim=wl["use_ddim"], **({} if cond is None else dict(guidance_scale=wl["guidance"], un_cond=None)))
        tab = p.table()
        ms, n, fl, _ = tab["conv_igemm"]
        total_ms = sum(v[0] for v in tab.values())
        alg = fl / (ms * 1e-3) / 1e12          # algorithmic FLOPs of the reference convolutions / launch time
        if prec == 0:
            mode = 0
            name, peak, executed = "conv_igemm_kernel<MODE 0> (v_mfma_f32_32x32x2_f32 implicit-GEMM conv)", PEAK_FP32_TFLOPS, 1
        elif prec == 4:
            mode = 5
            name, peak, executed = "conv_igemm_kernel<..., MODE 5> (REDUCED precision: operands rounded to bf16, one MFMA term, fp32 accumulate)", PEAK_BF16_TFLOPS, 1
        else:   # six bf16 MFMA terms per fp32 product: the matrix pipe executes 6x the algorithmic FLOPs
            mode = 3 if (prec == 1 and BLK.PRESPLIT_WEIGHTS) else prec
            name, peak, executed = "conv_igemm_kernel<..., MODE %d> (fp32 via exact 3 x bf16 split, v_mfma_f32_32x32x16_bf16, fp32 accumulate%s)" % (
                mode, "; weights pre-split at load" if mode == 3 else ""), PEAK_BF16_TFLOPS, 6
        ach = alg * executed
        traffic, traffic_src = None, None
        try:  # HBM bytes per launch of the dominant kernel: PMC counters cannot be read live, so they come from the committed
            # rocprofv3 passes over this same command (scripts/pmc_bench_traffic.sh -> profiles/pmc_bench_traffic.json)
            pj = json.load(open(Path(__file__).resolve().parent / "profiles" / "pmc_bench_traffic.json"))
            cand = [e for e in pj["kernels"] if "conv_igemm_kernel" in e["kernel"] and (", %d>" % mode if prec else ", 0>") in e["kernel"]]
            if cand:
                traffic = max(cand, key=lambda e: e["total_fetch_KiB_raw"])["hbm_bytes_per_launch"]
                traffic_src = "profiles/pmc_bench_traffic.json: " + pj["method"]
        except (OSError, KeyError, ValueError):
            pass
        roof = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": round(peak, 1),
                "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic, "traffic_unit": "HBM bytes per launch of the most frequent conv tile",
                "traffic_source": traffic_src, "launches": int(n), "avg_launch_ms": round(ms / n, 5),
                "algorithmic_tflops": round(alg, 2), "executed_over_algorithmic": executed,
                "share_of_gpu_time": round(ms / total_ms, 4),
                "families_ms": {k: round(v[0], 3) for k, v in sorted(tab.items(), key=lambda kv: -kv[1][0])}}
    alt = None
    if not args.no_alt_path and rank == 0 and world == 1:
        # the same step on the other conv arithmetic, timed the same way (1 warm-up + max(1, steps) runs), for comparison
        BLK.CONV_PRECISION = 0 if prec in (1, 2) else 1
        one_step(2000)
        fence()
        t1 = time.perf_counter()
        for k in range(max(1, args.steps)):
            one_step(k)
        fence()
        dta = (time.perf_counter() - t1) / max(1, args.steps)
        alt = {"conv_precision": BLK.CONV_PRECISION, "value": round(n_global / dta, 4), "unit": "images/s", "ms_per_step": round(dta * 1e3, 2)}
        BLK.CONV_PRECISION = prec
    cpu = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        cpu = cpu_baseline(wl["classes"])

    if rank == 0:
        gflop_img = GFLOP_PER_IMAGE_CFG2 if args.workload == "cfg2" and not args.ddim_steps else None
        out = {
            "metric": "images/sec at 256x256, 150 DDIM steps (DiffusionPipeline.sample incl. VAE decode)",
            "value": round(ips, 4), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 operands / f32 accumulate (opt-in, NOT the headline configuration)" if prec == 4 else "f32", "conv_arithmetic": {4: "REDUCED precision: conv operands rounded to bf16 (round to nearest even), one MFMA term, fp32 accumulate; everything else fp32", 0: "fp32 MFMA (v_mfma_f32_32x32x2_f32)", 1: "fp32 operands split exactly into 3 bf16 terms, 6 product terms on the bf16 MFMA, fp32 accumulate (error vs fp64 <= the fp32-MFMA kernel's: tests/test_kernels_gpu.py)", 2: "as 1, per-chunk sums added by the VALU"}[prec],
            "data": "synthetic (seeded weights of the published architecture, device Philox noise)",
            "config": {"workload": f"{args.workload}: {B} images/GPU, latent {wl['latent']}, {wl['steps']} {'DDIM' if wl['use_ddim'] else 'DDPM'} iterations, "
                                   f"{'uncond' if cond is None else 'cond %d-class g=%s' % (wl['classes'], wl['guidance'])}, decode to {8 * wl['latent'][1]}x{8 * wl['latent'][2]}",
                       "global_batch": n_global, "parallelism": f"dp{world} (batch rows sharded, 1 all-gather of images)"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if alt:
            out["other_conv_arithmetic"] = alt
        if gflop_img:
            out["whole_path_algorithmic_tflops_per_gpu"] = round(ips / world * gflop_img / 1e3, 2)
            if prec == 0:
                out["whole_path_frac_of_fp32_peak"] = round(ips / world * gflop_img / 1e3 / PEAK_FP32_TFLOPS, 4)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

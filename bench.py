#!/usr/bin/env python3
"""bench.py -- images/sec of DiffusionPipeline.sample() on MI355X (BASELINE.json metric).

One "step" = one full `sample()` of the workload: cfg2 = 16 images per GPU, latent (8,32,32) -> 256x256,
150 DDIM iterations (eta=1), unconditional, published architecture (UNet 194 M params + VAE), fp32, seeded
weights, device Philox noise generated inside the timed region, VAE decode and the image all-gather INCLUDED;
weights resident in HBM.
N>1: one process per GPU, batch rows sharded (weak scaling: 16 images per GPU), no collective in the loop, one
RCCL all-gather of the images per step.  `python bench.py --gpus N` launches its own N ranks
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`) when it is not already
running under a launcher, and fails loudly when fewer than N devices are visible.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch

PEAK_FP32_TFLOPS = 157.3           # MI355X fp32 matrix/vector peak (MI355X_MICROARCH.md): SURVEY §8(d)'s denominator for the ALGORITHMIC flops
PEAK_MFMA16_TFLOPS = 16 * 157.3    # dense bf16 / fp16 MFMA peak (= 2516.8, "~2.5 PF dense", same guide): what the split-arithmetic kernels EXECUTE on
GFLOP_PER_IMAGE_CFG2 = 7743.2      # SURVEY §8d / BASELINE.md: 150 x 51.202 (UNet) + 62.923 (VAE decode)
WORKLOADS = {
    # name: (per-GPU batch, latent, ddim steps, use_ddim, num_classes, guidance)
    "cfg2": dict(batch=16, latent=(8, 32, 32), steps=150, use_ddim=True, classes=None, guidance=1.0),
    "cfg3_g1": dict(batch=16, latent=(8, 32, 32), steps=150, use_ddim=True, classes=3, guidance=1.0),
    "cfg3_g8": dict(batch=16, latent=(8, 32, 32), steps=150, use_ddim=True, classes=3, guidance=8.0),
    "cfg4": dict(batch=8, latent=(8, 32, 32), steps=1000, use_ddim=False, classes=None, guidance=1.0),
    "cfg5": dict(batch=8, latent=(8, 64, 64), steps=150, use_ddim=True, classes=None, guidance=1.0),
    # not a BASELINE config: the chunk the reference's own bulk generator samples at (scripts/helpers/sample_dataset.py:26-27,38: 200 images per call, 2-class
    # condition, guidance 1) -- no plan table holds this batch, the planner's cost model and the Winograd rule decide (round 6)
    "bulk200": dict(batch=200, latent=(8, 32, 32), steps=150, use_ddim=True, classes=2, guidance=1.0),
}
ARITH = {
    0: dict(kernel="conv_igemm_kernel<..., MODE 0>", pmc_match=("conv_igemm_kernel<", ", 0, "), terms=1, peak=PEAK_FP32_TFLOPS, dtype="f32",
            text="fp32 MFMA (v_mfma_f32_32x32x2_f32): bit-for-bit an fp32 fma chain"),
    1: dict(kernel="conv_igemm_kernel<..., MODE 3>", pmc_match=("conv_igemm_kernel<", ", 3, "), terms=6, peak=PEAK_MFMA16_TFLOPS, dtype="f32",
            text="fp32 operands split EXACTLY into 3 bf16 terms (24 bits), 6 product terms on v_mfma_f32_32x32x16_bf16, fp32 accumulate; weights "
                 "pre-split at load.  Round 6: the 3x3 stride-1 convolutions of the 8 x 8 and 16 x 16 levels in their Winograd F(2x2,3x3) form on THIS arithmetic "
                 "(fp32 transforms, component GEMMs on this kernel with exactly split operands; `direct_form_everywhere` = the configuration of rounds 1-5)"),
    4: dict(kernel="conv_igemm_kernel<..., MODE 5>", pmc_match=("conv_igemm_kernel<", ", 5, "), terms=1, peak=PEAK_MFMA16_TFLOPS,
            dtype="bf16 operands / f32 accumulate (opt-in, NOT the headline configuration)",
            text="REDUCED precision: conv operands rounded to bf16 (round to nearest even), one MFMA term, fp32 accumulate; everything else fp32"),
    5: dict(kernel="conv_f16x2_kernel<BM, BN, WM, WN> (+ its halo form conv_halo_kernel and conv_group_kernel: conv_res in the grid of its ResBlock's 3x3; the same kernel runs the component GEMMs of the Winograd form)", pmc_match=("conv_f16x2_kernel<",), terms=3, peak=PEAK_MFMA16_TFLOPS,
            dtype="f32 (emulated: fp16 pairs -- 23-bit operands, and 23-bit STORAGE of the tensors between the UNet's conv blocks: they exist as fp16 pairs only)",
            text="fp32 EMULATED through PAIRS of fp16: every operand stored as hi + lo/2048 (23 of 24 significand bits, error <= one fp32 ulp, zero for 3 values "
                 "of 4), 3 product terms on v_mfma_f32_32x32x16_f16 (the lo*lo term is dropped), fp32 accumulate; both operands moved HBM->LDS by "
                 "LDS-DMA.  Error vs an fp64 convolution: < 3x the fp32-MFMA kernel's + 1e-6 (asserted on every test shape, "
                 "tests/test_kernels_gpu.py::test_conv_f16x2), 0.4-2.1x measured (profiles/r02_split_accuracy.txt, profiles/r04_parity_measured.txt); not "
                 "bit-width-equal to fp32 -- the exact-operand arithmetics are timed in the same run (other_conv_arithmetic).  Round 5: the 3x3 stride-1 "
                 "convolutions of the 8 x 8 and 16 x 16 levels run in their Winograd F(2x2,3x3) form on this arithmetic (16 component GEMMs on the same "
                 "kernel, 2.25x fewer matrix instructions; which shapes: mf_wino_preferred, a rule since round 6; error vs fp64 at or below the direct form's, profiles/r05_winograd_ab.txt)"),
    6: dict(kernel="conv_f16x2_kernel<BM, BN, WM, WN, NST, 1>", pmc_match=("conv_f16x2_kernel<", ", 1>"), terms=1, peak=PEAK_MFMA16_TFLOPS,
            dtype="f16 operands / f32 accumulate (opt-in, NOT the headline configuration)",
            text="REDUCED precision on the LDS-DMA kernel: conv operands rounded to fp16 (11 significant bits, per-sample power-of-two scales), one MFMA "
                 "term, fp32 accumulate; everything else fp32"),
}


def self_launch(args) -> int:
    """`python bench.py --gpus N` outside a launcher: become N ranks under torch.distributed.run on this node."""
    n_vis = torch.cuda.device_count()
    if n_vis < args.gpus and os.environ.get("MEDFUSION_BENCH_SHARE_GPU") != "1":
        print(f"bench.py: --gpus {args.gpus} requested but only {n_vis} ROCm device(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def pin_rank(dev, local: int, world: int):
    """Give this rank its own CPUs (a rank needs ~0.55 core of launch work per step; eight unpinned ranks migrate across sockets and share
    cores with each other's Python threads): the CPUs of the NUMA node its GPU hangs on (sysfs, by PCI address), divided among the ranks
    that share that node by local rank; an even split of the allowed CPUs when sysfs does not tell.  Returns what it did (for the JSON line)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    if world <= 1 or len(allowed) < 2 * world:
        return {"cpus": len(allowed), "pinned": False}
    node_cpus, node = None, None
    try:
        pr = torch.cuda.get_device_properties(dev)
        addr = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{addr}/numa_node").read())
        if node >= 0:
            cpus = set()
            for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
            node_cpus = sorted(cpus & set(allowed))
    except (OSError, ValueError, AttributeError):
        node_cpus = None
    if node_cpus and len(node_cpus) >= 2:
        # ranks are spread evenly over the nodes on the usual 8-GPU boards: take the slice of this node by the rank's position among `world`
        per = max(2, len(node_cpus) * max(1, len(allowed) // len(node_cpus)) // world)
        k = (local * per) % max(1, len(node_cpus) - per + 1) if per < len(node_cpus) else 0
        mine = node_cpus[k:k + per]
    else:
        per = len(allowed) // world
        mine = allowed[local * per:(local + 1) * per]
    try:
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(4, len(mine))))
    except OSError:
        return {"cpus": len(allowed), "pinned": False}
    return {"cpus": len(mine), "first": mine[0], "last": mine[-1], "numa_node": node, "pinned": True}


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


MEASURE_ALL_THREADS = False
CHECK = {}


def cpu_baseline(classes):
    """The oracle (CPU restatement of the reference, kind='port') on this box's host cores, bounded sample:
    3 timed UNet forwards at B=4 and 1 VAE decode at B=1 -> extrapolated cfg2 images/s."""
    from oracle import restate as R   # the ONLY place bench.py touches oracle/: the CPU leg
    from oracle import synth as S

    ncpu = os.cpu_count() or 1
    threads = min(32, ncpu)  # ATen's CPU convolutions collapse past ~32 threads on this host class (256 threads: 50x slower, measured in round 1)
    torch.set_num_threads(threads)
    unet = R.UNet(**R.published_unet_kwargs(classes)).eval()
    vae = R.VAE(**R.published_vae_kwargs(8)).eval()
    S.synth_state_dict(unet, "published.unet.")
    S.synth_state_dict(vae, "published.vae.")
    x = S.synth_input("cpu_x", (4, 8, 32, 32))
    t = torch.full((4,), 500)
    with torch.no_grad():
        unet(x, t)  # warm-up
        t0 = time.perf_counter()
        for _ in range(3):
            y_unet = unet(x, t)[0]
        t_unet = (time.perf_counter() - t0) / 3
        z = x[:1]
        vae.decode(z)
        t0 = time.perf_counter()
        y_dec = vae.decode(z)
        t_dec = time.perf_counter() - t0
        CHECK.update(x=x, t=t, y_unet=y_unet, z=z, y_dec=y_dec)   # (the timed evaluations double as the checker of the GPU path: parity_check)
        # SURVEY 8(d) says os.cpu_count() threads: measured here on ONE UNet forward at B=1, next to the same forward on `threads` (bounded: the
        # all-threads run was 50x slower in round 1 -- ATen's CPU convolutions oversubscribe -- which is why the baseline above uses `threads`)
        t1 = torch.full((1,), 500)
        unet(z, t1)
        t0 = time.perf_counter()
        unet(z, t1)
        t_b1 = time.perf_counter() - t0
        all_threads = None   # (measured only with --cpu-all-threads, ~100 s: round 4 on this host class gave 51.0 s at 256 threads vs 0.198 s at 32 -- profiles/r04_baseline_bench_cfg2.json)
        if ncpu > threads and MEASURE_ALL_THREADS:
            torch.set_num_threads(ncpu)
            unet(z, t1)
            t0 = time.perf_counter()
            unet(z, t1)
            t_b1_all = time.perf_counter() - t0
            torch.set_num_threads(threads)
            all_threads = {"threads": ncpu, "unet_b1_seconds": round(t_b1_all, 4), f"unet_b1_seconds_at_{threads}_threads": round(t_b1, 4),
                           "slowdown_vs_baseline_threads": round(t_b1_all / t_b1, 2),
                           "what": f"one published-UNet forward at B=1 on all {ncpu} logical CPUs vs on {threads}: the reason cpu_baseline.cores is {threads}, not os.cpu_count()"}
        # BASELINE.json configs[0] timed IN FULL: 64x64 images = latent (8, 8, 8), B=2, unconditional, 50 DDIM iterations + decode
        pipe = R.DiffusionPipeline(noise_scheduler=R.GaussianNoiseScheduler(**R.published_scheduler_kwargs()), noise_estimator=unet, latent_embedder=vae,
                                   estimator_objective="x_T", clip_x0=False)
        pipe.set_noise_fn(S.PhiloxNoise(3))
        t0 = time.perf_counter()
        img = pipe.sample(2, (8, 8, 8), steps=50, use_ddim=True)
        t_cfg1 = time.perf_counter() - t0
        assert img.shape == (2, 3, 64, 64) and bool(torch.isfinite(img).all())
    ips = 4.0 / (150 * t_unet + 4 * t_dec)
    return {"value": round(ips, 5), "unit": "images/s", "cores": threads, "host_logical_cpus": ncpu, "cpu_model": cpu_model(), "kind": "port",
            "sample": f"oracle (CPU restatement of the reference, torch fp32, {threads} threads of {ncpu} logical CPUs): 3 UNet forwards at B=4 "
                      f"({t_unet:.3f} s each) + 1 VAE decode at B=1 ({t_dec:.3f} s), extrapolated to 150 iterations x 4 images",
            "all_threads": all_threads, "unet_b1_seconds": round(t_b1, 4),
            "cfg1_full": {"value": round(2.0 / t_cfg1, 4), "unit": "images/s", "seconds": round(t_cfg1, 2),
                          "what": "BASELINE configs[0] run in full on the same threads: 2 unconditional 64x64 images (latent 8x8x8), 50 DDIM iterations + VAE "
                                  "decode, published architecture"}}


def pmc_traffic(match):
    """HBM bytes per conv launch (launch-weighted average over the kernel's tile instantiations) and of its most frequent tile: PMC counters
    cannot be read live, so they come from the committed rocprofv3 passes over this same command (scripts/pmc_bench_traffic.sh ->
    profiles/pmc_bench_traffic.json; FETCH_SIZE doubled for gfx950).  Third value: {kernel name as rocprofv3 prints it: HBM bytes per launch} of
    EVERY kernel of the profile (roofline.instantiations[] joins on it)."""
    try:
        pj = json.load(open(ROOT / "profiles" / "pmc_bench_traffic.json"))
        # every instantiation of the arithmetic's matrix kernels: the 9-copy kernel, the halo kernel and the grouped launches (two convolutions
        # in one grid); match[1:] narrows to the single-term instantiations where the arithmetic has its own
        names = ("conv_f16x2_kernel<", "conv_halo_kernel<", "conv_group_kernel<")
        cand = [e for e in pj["kernels"] if any(nm in e["kernel"] for nm in names) and all(m in e["kernel"] for m in match[1:])]
        by_name = {e["kernel"]: e for e in pj["kernels"]}
        if cand:
            top = max(cand, key=lambda e: e.get("launches", 0))
            n = sum(e["launches"] for e in cand)
            avg = sum(e["launches"] * e["hbm_bytes_per_launch"] for e in cand) / n
            from medfusion_amd.build import conv_source_stamp
            stamp = pj.get("conv_source_stamp")
            return int(avg), {"file": "profiles/pmc_bench_traffic.json", "command": pj.get("command"), "method": pj["method"], "launches_profiled": n,
                              "conv_source_stamp_of_the_profile": stamp, "conv_source_stamp_now": conv_source_stamp(),
                              "stale": stamp != conv_source_stamp(),
                              "most_frequent_tile": {"kernel": top["kernel"][:80], "launches": top["launches"],
                                                     "hbm_bytes_per_launch": top["hbm_bytes_per_launch"]}}, by_name
    except (OSError, KeyError, ValueError):
        pass
    return None, None, {}


def pmc_match(by_name, kernel):
    """the PMC entry of `kernel` (mf_prof_tag_name's spelling vs rocprofv3's: same text up to a leading `void ` and the parameter list)"""
    norm = lambda k: k.replace("void ", "").split("(")[0].replace(" ", "")
    for k, e in by_name.items():
        if norm(k) == norm(kernel):
            return e
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg2", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch override")
    ap.add_argument("--ddim-steps", type=int, default=None, help="override the number of denoise iterations (non-headline)")
    ap.add_argument("--graph", action="store_true", help="force replaying the denoise iteration as a captured hipGraph (default: decided by denoise() from the problem size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--conv-precision", type=int, default=None, choices=sorted(ARITH),
                    help="arithmetic of the conv kernel (MF_CONV_*): see ARITH in this file; 4 is the opt-in REDUCED precision mode, never the headline")
    ap.add_argument("--alt-precision", type=int, default=None, choices=sorted(ARITH), help="also time the same step on this arithmetic (default: the other fp32-class ones)")
    ap.add_argument("--no-alt-path", action="store_true", help="skip the extra timed steps on the other conv arithmetics")
    ap.add_argument("--cpu-all-threads", action="store_true", help="re-measure the oracle's UNet forward on ALL logical CPUs next to the 32-thread baseline (~100 s)")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the other BASELINE.json configs (cfg3 g=1 / g=8, cfg4, cfg5) that a default cfg2 run also times")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    global MEASURE_ALL_THREADS
    MEASURE_ALL_THREADS = args.cpu_all_threads

    import medfusion_amd as M
    from medfusion_amd import blocks as BLK
    from medfusion_amd import dist as D
    from medfusion_amd import kernels as K
    from medfusion_amd import published as P
    import torch.distributed as dist

    if args.conv_precision is not None:
        BLK.CONV_PRECISION = args.conv_precision
    prec = BLK.CONV_PRECISION
    # MEDFUSION_BENCH_SHARE_GPU=1 (tests on a 1-GPU box only): every rank uses device 0 and the group runs over gloo -- exercises the whole
    # N > 1 code path (sharding, fences, MAX over ranks, gather, teardown); the line it prints says so in `parallelism`
    share = os.environ.get("MEDFUSION_BENCH_SHARE_GPU") == "1"
    rank, local, world = D.init_from_env("gloo" if share else None)
    if world != args.gpus:
        raise SystemExit(f"bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}")
    if share:
        local = 0
    if torch.cuda.device_count() < (local + 1 if world > 1 else 1):
        raise SystemExit(f"bench.py: rank {rank} needs device {local}, {torch.cuda.device_count()} visible")
    dev = torch.device("cuda", local if world > 1 else 0)   # device = LOCAL_RANK of the launcher: no HIP_VISIBLE_DEVICES games
    torch.cuda.set_device(dev)
    pinned = pin_rank(dev, local, world)
    wl = dict(WORKLOADS[args.workload])
    if args.batch:
        wl["batch"] = args.batch
    if args.ddim_steps:
        wl["steps"] = args.ddim_steps
    B, n_global = wl["batch"], wl["batch"] * world
    pipe = P.build_published_pipeline(dev, wl["classes"])
    cond = (torch.arange(n_global, device=dev) % wl["classes"]) if wl["classes"] else None
    kw = dict(steps=wl["steps"], use_ddim=wl["use_ddim"])
    if args.graph:
        kw["use_graph"] = True    # (default: denoise() decides -- the native command list wherever the iteration is replayable)
    if cond is not None:
        kw.update(guidance_scale=wl["guidance"], un_cond=None)

    def one_step(seed):
        return D.sample_sharded(pipe, n_global, wl["latent"], condition=cond, noise=M.PhiloxDeviceNoise(seed), **kw)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(nsteps, seed0=0):
        fence()
        t0 = time.perf_counter()
        for k in range(nsteps):
            img = one_step(seed0 + k)
        fence()
        dt = time.perf_counter() - t0
        per_rank[:] = [dt]
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
            every = [torch.empty_like(tt) for _ in range(world)]
            dist.all_gather(every, tt)
            per_rank[:] = [float(v.item()) for v in every]
            dt = max(per_rank)                                   # MAX over ranks
        assert img.shape[0] == n_global and bool(torch.isfinite(img).all())
        return dt

    per_rank = []
    for w in range(args.warmup):
        one_step(1000 + w)
    dt = timed(args.steps)
    rank_ms = [round(v / args.steps * 1e3, 2) for v in per_rank]
    ips = n_global * args.steps / dt

    roof = None
    ar = ARITH[prec]
    if not args.no_roofline and rank == 0:
        # live launch timing: one more full step of the SAME workload with every launch bracketed by hipEvents on its stream (mf_prof_*);
        # rocprofv3 --kernel-trace of this command sees the same mix (profiles/)
        with K.prof() as p:
            pipe.sample(B, wl["latent"], condition=None if cond is None else cond[:B], noise=M.PhiloxDeviceNoise(7), steps=wl["steps"],
                        use_ddim=wl["use_ddim"], **({} if cond is None else dict(guidance_scale=wl["guidance"], un_cond=None)))
        tab = p.table()
        ms, n, fl, alg_bytes, ex = tab["conv_igemm"]
        if "conv_gn_fused" in tab:   # (opt-in MEDFUSION_FUSED_APPLY=1: those launches also hold the GroupNorm-apply work -- counted whole)
            ms, n, fl, alg_bytes, ex = (a + b for a, b in zip(tab["conv_igemm"], tab["conv_gn_fused"]))
        total_ms = sum(v[0] for v in tab.values())
        alg = fl / (ms * 1e-3) / 1e12     # algorithmic FLOPs of the reference convolutions / their launch time
        exe = ex / (ms * 1e-3) / 1e12     # FLOPs the matrix pipe executes: terms per product x the MACs actually done (sub-pixel up-convs: 4/9)
        traffic, traffic_src, pmc_by_name = pmc_traffic(ar["pmc_match"])
        sustained = None
        if ar["peak"] > 1000:   # the 16-bit matrix pipe: what it sustains on THIS device with nothing but MFMAs in flight, by operand data
            rnd, zer = K.mfma_sustained_tflops(dev, "random"), K.mfma_sustained_tflops(dev, "zeros")
            sustained = {"random_operands_tflops": round(rnd, 1), "zero_operands_tflops": round(zer, 1),
                         "how": "mf_mfma_rate_probe_f16 live on this device: v_mfma_f32_32x32x16_f16 from registers only, 4 chains per wave, 2 waves "
                                "per SIMD, one workgroup per CU, 0.6 ms bursts, best of 3 (scripts/mfma_power_probe.hip, "
                                "profiles/r02_mfma_power_probe.txt): the pipe is power-limited by the data it multiplies"}
        # per kernel instantiation (name as rocprofv3 --kernel-trace prints it), so that every row can be recomputed from the committed
        # profiles/*_kernel_stats.csv: launches, average duration, algorithmic and executed GFLOP per launch, executed fraction of the peak
        inst = {}
        for r in K.prof_rows("conv_igemm") + (K.prof_rows("conv_gn_fused") if "conv_gn_fused" in tab else []):
            e = inst.setdefault(r["kernel"], dict(kernel=r["kernel"], launches=0, ms=0.0, flops=0.0, exec_flops=0.0, bytes=0.0, winograd_launches=0, winograd_ms=0.0))
            e["launches"] += r["launches"]; e["ms"] += r["ms"]; e["flops"] += r["flops"]; e["exec_flops"] += r["exec_flops"]; e["bytes"] += r["bytes"]
            if r["variant"] == 1:
                e["winograd_launches"] += r["launches"]; e["winograd_ms"] += r["ms"]
        instantiations = [dict(kernel=e["kernel"], launches=int(e["launches"]), avg_us=round(e["ms"] / e["launches"] * 1e3, 2), total_ms=round(e["ms"], 3),
                               algorithmic_gflop_per_launch=round(e["flops"] / e["launches"] / 1e9, 3), executed_gflop_per_launch=round(e["exec_flops"] / e["launches"] / 1e9, 3),
                               algorithmic_tflops=round(e["flops"] / (e["ms"] * 1e-3) / 1e12, 1), executed_frac_of_peak=round(e["exec_flops"] / (e["ms"] * 1e-3) / 1e12 / ar["peak"], 4),
                               winograd_component_gemm_launches=int(e["winograd_launches"]), winograd_ms=round(e["winograd_ms"], 3),
                               # algorithmic bytes PER FORM (direct: x + w + y; Winograd component GEMM: V + U + M in the transform domain; grouped
                               # launches: both members), the PMC bytes of the same instantiation from the committed passes, and their ratio
                               algorithmic_bytes_per_launch=int(e["bytes"] / e["launches"]),
                               hbm_bytes_per_launch_pmc=(pmc_match(pmc_by_name, e["kernel"]) or {}).get("hbm_bytes_per_launch"),
                               traffic_ratio=(round(pmc_match(pmc_by_name, e["kernel"])["hbm_bytes_per_launch"] / (e["bytes"] / e["launches"]), 3)
                                              if pmc_match(pmc_by_name, e["kernel"]) else None))
                          for e in sorted(inst.values(), key=lambda e: -e["ms"])]
        wino_ms = sum(e["winograd_ms"] for e in inst.values())
        wino_n = sum(e["winograd_launches"] for e in inst.values())
        hbm = {}
        for fam in ("gn_apply", "wino_xform", "splitk_reduce", "gn_stats", "sched", "noise"):
            if fam in tab and tab[fam][0] > 0:
                hbm[fam] = {"ms": round(tab[fam][0], 3), "launches": int(tab[fam][1]), "algorithmic_GBps": round(tab[fam][3] / (tab[fam][0] * 1e-3) / 1e9, 1)}
        roof = {"bound": "mfma", "kernel": f"{ar['kernel']} -- {ar['text']}",
                "achieved": round(exe, 2), "peak": round(ar["peak"], 1), "unit": "TFLOP/s", "frac": round(exe / ar["peak"], 4),
                "mfma_sustained": sustained,
                "frac_of_sustained_random_operands": None if sustained is None else round(exe / sustained["random_operands_tflops"], 4),
                "frac_is": "EXECUTED matrix flops of the implicit-GEMM conv kernel / the dense MFMA peak of the pipe it runs on "
                           f"({ar['terms']} matrix term(s) per product; a convolution on its Winograd form executes 4/9 of its multiplications -- the "
                           "form REMOVES work, so this fraction falls while algorithmic_tflops and images/s rise: frac_arithmetic_ceiling is the "
                           "round-over-round figure)",
                "winograd": {"component_gemm_launches": int(wino_n), "ms": round(wino_ms, 3), "share_of_conv_time": round(wino_ms / ms, 4),
                             "tail_and_transform_ms": round(tab["wino_xform"][0], 3) if "wino_xform" in tab else 0.0,
                             "what": "3x3 stride-1 convolutions mf_wino_preferred admits (a rule in Cin, Cout, H W, N since ABI 240; it reproduces csrc/wino_plan_table.inc) as 16 component GEMMs on the same kernel (algorithmic flops: the "
                                     "convolution's own 2 M Cout 9 Cin; executed: 3 terms x 2 (4 M / 4) Cout Cin); their output transform + GroupNorm + Swish + "
                                     "residual + next input transform is ONE tail launch each (family wino_xform, an HBM-bound pass: hbm_bound_passes)"},
                "instantiations": instantiations,
                "algorithmic_tflops": round(alg, 2),
                "x_over_fp32_peak_algorithmic": round(alg / PEAK_FP32_TFLOPS, 4),   # (a ratio, not a roofline fraction: the work runs on the fp16 pipe)
                "frac_arithmetic_ceiling": round(alg / (ar["peak"] / ar["terms"]), 4),
                "frac_note": "frac_arithmetic_ceiling credits a Winograd convolution with the flops of the 3x3 it stands for (2 M Cout 9 Cin) against the ceiling "
                             "of the DIRECT three-term form (peak / 3): a round-over-round figure of merit, NOT a hardware fraction (it can exceed what the direct form "
                             "could ever reach); the hardware fraction is `frac` = executed matrix flops / nominal dense fp16 peak, per instantiation in instantiations[]",
                "arithmetic_ceiling_tflops": round(ar["peak"] / ar["terms"], 1),
                "executed_over_algorithmic": round(ex / fl, 4),
                "traffic": traffic, "traffic_unit": "HBM bytes per conv launch, launch-weighted average over the tile instantiations (PMC)",
                "traffic_from_committed_profile": True,   # PMC counters cannot be read inside this process: NOT measured in this run
                "traffic_stale": None if traffic_src is None else traffic_src["stale"],   # the conv sources / tile table changed since the PMC passes
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": int(alg_bytes / n),   # per form (see instantiations[]), launch-weighted average over the conv family
                "traffic_ratio": None if not traffic else round(traffic / (alg_bytes / n), 3),
                "launches": int(n), "avg_launch_ms": round(ms / n, 5), "share_of_gpu_time": round(ms / total_ms, 4),
                "launch_timing": "every launch's own start/stop HIP events (hipExtLaunchKernel: the dispatch's execution interval, what rocprofv3 --kernel-trace "
                                 "reports; no event packets between dependent kernels -- until round 3 two hipEventRecord per launch inflated the conv family "
                                 "by ~4 % against the profiler); kernel boundaries are therefore NOT in families_ms",
                "families_ms": {k: round(v[0], 3) for k, v in sorted(tab.items(), key=lambda kv: -kv[1][0])},
                "hbm_bound_passes": hbm}
    alts = []
    if not args.no_alt_path and rank == 0 and world == 1:
        # the same step on the other conv arithmetics, timed the same way (1 warm-up + max(1, steps) runs), for comparison
        others = [args.alt_precision] if args.alt_precision is not None else [a for a in (5, 1, 0) if a != prec][:2]
        for a in others:
            BLK.CONV_PRECISION = a
            one_step(2000)
            dta = timed(max(1, args.steps)) / max(1, args.steps)
            ent = {"conv_precision": a, "arithmetic": ARITH[a]["text"], "value": round(n_global / dta, 4), "unit": "images/s", "ms_per_step": round(dta * 1e3, 2)}
            if a == 1:
                # round 6: the exact bf16-triplet arithmetic takes the Winograd F(2x2,3x3) form on the shapes mf_wino_preferred admits (fp32 transforms, the component
                # GEMMs on its own kernel, the same tail; blocks.WINOGRAD_F32).  The direct form everywhere -- what this entry measured in rounds 1-5 -- next to it.
                ent["winograd_form"] = {0: "none: direct form everywhere", 1: "on the shapes mf_wino_preferred admits (8 x 8 and 16 x 16 levels)", 2: "wherever the library can"}[BLK.WINOGRAD_F32]
                if BLK.WINOGRAD_F32:
                    wf = BLK.WINOGRAD_F32
                    BLK.WINOGRAD_F32 = 0
                    one_step(2100)
                    k = max(1, min(3, args.steps))
                    dtd = timed(k) / k
                    BLK.WINOGRAD_F32 = wf
                    ent["direct_form_everywhere"] = {"value": round(n_global / dtd, 4), "unit": "images/s", "ms_per_step": round(dtd * 1e3, 2), "steps": k}
            if a == 0:
                ent["winograd_form"] = "none: this entry is the bit-for-bit fp32 fma chain of the direct convolution, on purpose"
                # ... and, next to it, the same kernel running the Winograd component GEMMs (opt-in blocks.WINOGRAD_F32_MFMA: fp32 transforms, v_mfma_f32_32x32x2_f32
                # on unsplit fp32 operands, fp32 tail -- no operand splitting anywhere; another summation than the direct chain)
                BLK.WINOGRAD_F32_MFMA = 1
                one_step(2200)
                k = max(1, min(3, args.steps))
                dtw = timed(k) / k
                BLK.WINOGRAD_F32_MFMA = 0
                ent["with_the_winograd_form_opt_in"] = {"value": round(n_global / dtw, 4), "unit": "images/s", "ms_per_step": round(dtw * 1e3, 2), "steps": k,
                                                        "what": "MEDFUSION_WINOGRAD_F32_MFMA=1: 3x3 stride-1 convolutions of the 8 x 8 / 16 x 16 levels as fp32 Winograd F(2x2,3x3)"}
            alts.append(ent)
        BLK.CONV_PRECISION = prec
    reduced = []
    if not args.no_alt_path and rank == 0 and world == 1 and args.alt_precision is None and args.conv_precision is None:
        # the two opt-in REDUCED-precision modes on the same step (never the headline).  The single-term fp16 mode is also the
        # gauge VERDICT r03 asked for: one matrix term instead of three on the same kernel -- what it gains is what is NOT fixed cost.
        for a in (6, 4):
            BLK.CONV_PRECISION = a
            one_step(2500)
            dta = timed(2) / 2
            reduced.append({"conv_precision": a, "arithmetic": ARITH[a]["text"], "value": round(n_global / dta, 4), "unit": "images/s", "ms_per_step": round(dta * 1e3, 2),
                            "headline": False})
        BLK.CONV_PRECISION = prec
    others_wl = []
    if (not args.no_other_workloads and rank == 0 and world == 1 and args.workload == "cfg2" and not args.ddim_steps and not args.batch
            and args.conv_precision is None):
        # every other BASELINE.json config at its per-GPU size, in THIS process and on the default arithmetic: a warm-up
        # (short for cfg4: its loop is the 1000-iteration one), then the timed steps, fenced like the headline.  ~15 s in all.
        pipes = {None: pipe}
        for name, nsteps in (("cfg3_g1", 2), ("cfg3_g8", 2), ("cfg5", 2), ("cfg4", 1), ("bulk200", 1)):
            w2 = WORKLOADS[name]
            if w2["classes"] not in pipes:
                pipes[w2["classes"]] = P.build_published_pipeline(dev, w2["classes"])
            p2 = pipes[w2["classes"]]
            c2 = (torch.arange(w2["batch"], device=dev) % w2["classes"]) if w2["classes"] else None
            k2 = dict(use_ddim=w2["use_ddim"])
            if c2 is not None:
                k2.update(guidance_scale=w2["guidance"], un_cond=None)
            p2.sample(w2["batch"], w2["latent"], condition=c2, noise=M.PhiloxDeviceNoise(3000), steps=min(w2["steps"], 20), **k2)   # warm-up
            if name not in ("cfg4", "bulk200"):
                p2.sample(w2["batch"], w2["latent"], condition=c2, noise=M.PhiloxDeviceNoise(3001), steps=w2["steps"], **k2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(nsteps):
                img2 = p2.sample(w2["batch"], w2["latent"], condition=c2, noise=M.PhiloxDeviceNoise(3100 + k), steps=w2["steps"], **k2)
            torch.cuda.synchronize()
            d2 = (time.perf_counter() - t0) / nsteps
            assert img2.shape[0] == w2["batch"] and bool(torch.isfinite(img2).all())
            others_wl.append({"workload": name, "value": round(w2["batch"] / d2, 4), "unit": "images/s", "steps": nsteps, "ms_per_step": round(d2 * 1e3, 2),
                              "config": {"workload": f"{name}: {w2['batch']} images/GPU, latent {w2['latent']}, {w2['steps']} {'DDIM' if w2['use_ddim'] else 'DDPM'} "
                                                     f"iterations, {'uncond' if c2 is None else 'cond %d-class g=%s' % (w2['classes'], w2['guidance'])}, decode to "
                                                     f"{8 * w2['latent'][1]}x{8 * w2['latent'][2]}", "n_gpus": 1,
                                         "baseline_config": {"cfg3_g1": "configs[2] per-GPU share (128 / 8), guidance 1", "cfg3_g8": "configs[2] per-GPU share, guidance 8 (2B-row UNet calls)",
                                                             "cfg4": "configs[3]", "cfg5": "configs[4] per-GPU share (32 / 4)",
                                                             "bulk200": "none -- the reference's bulk harness chunk (scripts/helpers/sample_dataset.py:26-27), a batch outside every plan table"}[name]}})
        del pipes
    cpu = parity = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        cpu = cpu_baseline(wl["classes"])
        if CHECK and wl["classes"] is None:
            # the oracle evaluations that were just timed, as the CHECKER of what was benchmarked (same seeded weights): one UNet forward at
            # B = 4 and one VAE decode through the product path on this device.  A number from a path that computes something else is no number
            # (round 4: a bound-propagation bug produced finite, constant images at full speed) -- bench.py fails loudly instead of printing it.
            import importlib.util
            spec = importlib.util.spec_from_file_location("_medfusion_tests_util", ROOT / "tests" / "util.py")   # (by path: an installed top-level `tests` package must not shadow it)
            tu = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(tu)
            relerr, relerr_rows = tu.relerr, tu.relerr_rows
            with torch.no_grad():
                g_unet = pipe.noise_estimator(CHECK["x"].to(dev), CHECK["t"].to(dev), None)[0]
                g_dec = pipe.latent_embedder.decode(CHECK["z"].to(dev))
            e1, e2 = relerr_rows(g_unet, CHECK["y_unet"]), relerr(g_dec, CHECK["y_dec"])
            parity = {"unet_forward_B4_per_sample_relerr": float(f"{e1:.3e}"), "vae_decode_relerr": float(f"{e2:.3e}"), "tolerance": 1e-4,
                      "against": "the oracle (CPU restatement of the reference) on the same seeded weights and inputs", "ok": bool(e1 < 1e-4 and e2 < 1e-4)}
            if not parity["ok"]:
                raise SystemExit(f"bench.py: the benchmarked path FAILS its parity check against the oracle: {parity}")

    if rank == 0:
        gflop_img = GFLOP_PER_IMAGE_CFG2 if args.workload == "cfg2" and not args.ddim_steps else None
        out = {
            "metric": "images/sec at 256x256, 150 DDIM steps (DiffusionPipeline.sample incl. VAE decode)",
            "value": round(ips, 4), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ar["dtype"], "conv_precision": prec, "conv_arithmetic": ar["text"],
            "data": "synthetic (seeded weights of the published architecture, device Philox noise)",
            "config": {"workload": f"{args.workload}: {B} images/GPU, latent {wl['latent']}, {wl['steps']} {'DDIM' if wl['use_ddim'] else 'DDPM'} iterations, "
                                   f"{'uncond' if cond is None else 'cond %d-class g=%s' % (wl['classes'], wl['guidance'])}, decode to {8 * wl['latent'][1]}x{8 * wl['latent'][2]}",
                       "global_batch": n_global, "parallelism": f"dp{world} (batch rows sharded, 1 all-gather of images)" + (" -- TEST MODE: ranks share one GPU, gloo" if share else ""),
                       "world": world, "backend": (dist.get_backend() if world > 1 else None), "ms_per_step_by_rank": rank_ms, "cpu_pinning_rank0": pinned},
            "roofline": roof, "cpu_baseline": cpu, "parity_check": parity,
        }
        if alts:
            out["other_conv_arithmetic"] = alts
        if reduced:
            out["opt_in_reduced_precision"] = reduced
        if others_wl:
            out["other_workloads"] = others_wl
        if gflop_img:
            out["whole_path_algorithmic_tflops_per_gpu"] = round(ips / world * gflop_img / 1e3, 2)
            out["whole_path_x_over_fp32_peak"] = round(ips / world * gflop_img / 1e3 / PEAK_FP32_TFLOPS, 4)   # (a ratio, see roofline.x_over_...)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()   # rank 0 has been profiling on its own: every rank leaves the group together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

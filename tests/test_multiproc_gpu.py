"""The N > 1 path on real devices: one process per rank, REAL `DiffusionPipeline.sample(shard=...)` on the GPU, the
gather through torch.distributed.

* two ranks that SHARE the one GPU of a single-GPU box, gather over gloo (host tensors): everything of the multi-GPU path except the
  RCCL transport -- runs everywhere;
* two ranks on two GPUs over nccl (= RCCL on ROCm) when the box has them;
* `bench.py --gpus N` launches its own ranks, and refuses loudly when fewer than N devices are visible.
"""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parents[1]

WORKER = r"""
import os, sys, torch
sys.path.insert(0, {root!r})
import torch.distributed as dist
import medfusion_amd as M
from medfusion_amd import dist as D
from medfusion_amd import published as P
from medfusion_amd.unet import UNet, TimeEmbbeding, LabelEmbedder
from medfusion_amd.vae import VAE
backend, ngpu = sys.argv[1], int(sys.argv[2])
rank, local, world = D.init_from_env(backend)
dev = torch.device("cuda", local % ngpu)
torch.cuda.set_device(dev)
ukw = dict(in_ch=8, out_ch=8, spatial_dims=2, hid_chs=[32, 32, 64, 64], kernel_sizes=[3] * 4, strides=[1, 2, 2, 2], time_embedder=TimeEmbbeding,
           time_embedder_kwargs={{"emb_dim": 64}}, cond_embedder=LabelEmbedder, cond_embedder_kwargs={{"emb_dim": 64, "num_classes": 3}},
           deep_supervision=False, use_res_block=True, use_attention="none")
pipe = M.DiffusionPipeline(M.GaussianNoiseScheduler, UNet, None, P.published_scheduler_kwargs(), ukw, estimator_objective="x_T", clip_x0=False)
pipe.latent_embedder = VAE(in_channels=3, out_channels=3, emb_channels=8, spatial_dims=2, hid_chs=[32, 32, 64, 64], kernel_sizes=[3] * 4, strides=[1, 2, 2, 2], deep_supervision=1)
P.seeded_fill(pipe.noise_estimator, "mp.unet.")
P.seeded_fill(pipe.latent_embedder, "mp.vae.")
pipe.to(dev).eval()
n = 5                                            # odd: the shards differ by one row; with 8 ranks three shards are EMPTY
cond = (torch.arange(n, device=dev) % 3)
full = D.sample_sharded(pipe, n, (8, 8, 8), condition=cond, noise=M.PhiloxDeviceNoise(31), guidance_scale=4.0, un_cond=None, steps=3, use_ddim=True)
assert full.shape == (n, 3, 64, 64) and full.device == dev
if rank == 0:
    alone = pipe.sample(n, (8, 8, 8), condition=cond, noise=M.PhiloxDeviceNoise(31), guidance_scale=4.0, un_cond=None, steps=3, use_ddim=True)
    err = float((full - alone).abs().max() / alone.abs().max())
    print("MULTIPROC_OK", world, backend, err, flush=True)
    assert err < 1e-5, err                       # (another batch size => another conv tiling / split-K order: not bit-equal)
dist.barrier()
dist.destroy_process_group()
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(backend, world, ngpu, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=str(ROOT)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), str(script), backend, str(ngpu)]
    r = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "MULTIPROC_OK" in r.stdout, r.stdout[-2000:]
    return r.stdout


def test_two_ranks_sharing_one_gpu_gather_over_gloo(tmp_path):
    assert torch.cuda.is_available()
    _launch("gloo", 2, 1, tmp_path)


def test_eight_ranks_sharing_one_gpu_with_empty_shards(tmp_path):
    """the world size of the metric's node: 5 samples over 8 ranks -- ranks 5..7 own no row, launch nothing, and still take part in the
    gather (pad / trim of dist.gather_images); every rank ends with the 5 images of the single-process run"""
    assert torch.cuda.is_available()
    out = _launch("gloo", 8, 1, tmp_path)
    assert "MULTIPROC_OK 8 gloo" in out


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs for the RCCL transport")
def test_two_ranks_two_gpus_rccl(tmp_path):
    _launch("nccl", 2, 2, tmp_path)


RCCL1 = r"""
import os, sys, torch
sys.path.insert(0, {root!r})
os.environ["MEDFUSION_FORCE_COLLECTIVE"] = "1"
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1")
import torch.distributed as dist
import medfusion_amd as M
from medfusion_amd import dist as D
rank, local, world = D.init_from_env("nccl")           # a ONE-rank RCCL communicator on device 0
assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
dev = torch.device("cuda", 0)
x = torch.arange(5 * 3 * 8 * 8, dtype=torch.float32, device=dev).reshape(5, 3, 8, 8) * 0.5 - 7.0
out = D.gather_images(x, 5, 0, 1)                       # all_gather on device memory through RCCL (no early return)
torch.cuda.synchronize()
assert out.is_cuda and out.data_ptr() != x.data_ptr() and torch.equal(out, x)
# the library's own output through the same path: a tiny sample() gathered by the 1-rank group equals the un-gathered one
from medfusion_amd import published as P
from medfusion_amd.unet import UNet, TimeEmbbeding
ukw = dict(in_ch=8, out_ch=8, spatial_dims=2, hid_chs=[32, 32, 64, 64], kernel_sizes=[3] * 4, strides=[1, 2, 2, 2], time_embedder=TimeEmbbeding,
           time_embedder_kwargs={{"emb_dim": 64}}, cond_embedder=None, deep_supervision=False, use_res_block=True, use_attention="none")
pipe = M.DiffusionPipeline(M.GaussianNoiseScheduler, UNet, None, P.published_scheduler_kwargs(), ukw, estimator_objective="x_T", clip_x0=False)
P.seeded_fill(pipe.noise_estimator, "mp1.unet.")
pipe.to(dev).eval()
a = D.sample_sharded(pipe, 3, (8, 8, 8), noise=M.PhiloxDeviceNoise(5), steps=2, use_ddim=True)
b = pipe.sample(3, (8, 8, 8), noise=M.PhiloxDeviceNoise(5), steps=2, use_ddim=True)
assert torch.equal(a, b)
t = torch.ones(1 << 20, device=dev)
dist.all_reduce(t)                                      # (one more collective of the same communicator, 4 MB)
torch.cuda.synchronize()
assert float(t.sum()) == float(1 << 20)
print("RCCL1_OK", "rccl", ".".join(str(v) for v in torch.cuda.nccl.version()), flush=True)
dist.destroy_process_group()
"""


def test_rccl_world1_device_allgather(tmp_path):
    """RCCL executed on the hardware at hand: a ONE-rank nccl (= RCCL) process group on device 0, the gather of
    dist.gather_images on DEVICE tensors through it (MEDFUSION_FORCE_COLLECTIVE=1 removes the world == 1 early return), a sharded
    sample() through the same call, and an all-reduce on the same communicator.  No scaling number -- the transport, the communicator
    set-up and the device-memory path of the gather run for real."""
    script = tmp_path / "rccl1.py"
    script.write_text(RCCL1.format(root=str(ROOT)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, str(script)], cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "RCCL1_OK" in r.stdout, r.stdout[-2000:]
    print("[measured]", " | ".join(ln for ln in r.stdout.splitlines() if "RCCL1_OK" in ln or "rccl" in ln.lower()))


def test_bench_launches_its_own_ranks_or_refuses_loudly(tmp_path):
    """`python bench.py --gpus N` (no launcher around it): N ranks under torch.distributed.run when N devices exist, else exit code 2
    with a message -- never a silent 1-GPU run labelled otherwise."""
    n_vis = torch.cuda.device_count()
    want = n_vis + 1 if n_vis < 2 else 2
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(want), "--steps", "1", "--warmup", "0", "--ddim-steps", "2", "--no-cpu-baseline",
                        "--no-alt-path", "--no-roofline"], cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900)
    if n_vis >= 2:
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 32
    else:
        assert r.returncode == 2 and "visible" in r.stderr, (r.returncode, r.stderr[-500:])


def test_bench_multi_rank_path_on_one_gpu(tmp_path):
    """the whole N > 1 path of bench.py (self-launch under torch.distributed.run, sharded sample, fences, MAX over ranks, gather, teardown) with
    two ranks sharing device 0 over gloo (MEDFUSION_BENCH_SHARE_GPU=1: a test switch, the printed line says so)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MEDFUSION_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1", "--ddim-steps", "2", "--batch", "2", "--no-cpu-baseline",
                        "--no-alt-path"], cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 4 and line["value"] > 0 and "TEST MODE" in line["config"]["parallelism"]
    assert line["roofline"] is not None and line["roofline"]["launches"] > 0

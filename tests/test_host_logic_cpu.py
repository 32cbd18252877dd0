"""Host-side logic of the product that needs no GPU: loop timesteps (Q4/Q5), the per-step scalar records against the
oracle's own scalar algebra, sharding, the gloo world_size-2 path, and the Lightning-free checkpoint reader."""
import os
import pickle
import sys
import types
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import medfusion_amd as M
from medfusion_amd import dist as D
from oracle import restate as R
from oracle import synth as S

ROOT = Path(__file__).resolve().parents[1]


def test_loop_timesteps_quirks():
    sch = M.GaussianNoiseScheduler(**R.published_scheduler_kwargs())
    ts, n = sch.loop_timesteps(150, True)
    assert n == 150 and ts[:4] == [0, 6, 13, 20] and ts[-3:] == [985, 992, 999]       # truncated linspace
    ts, _ = sch.loop_timesteps(50, True)
    assert ts[:6] == [0, 20, 40, 61, 81, 101] and ts[-3:] == [958, 978, 999]
    ts, n = sch.loop_timesteps(7, False)
    assert ts == list(range(7)) and n == 7                                             # Q5: the FIRST `steps` entries
    ts, n = sch.loop_timesteps(None, False)
    assert ts == list(range(1000))
    ts, n = sch.loop_timesteps(None, True)
    assert n == 1000 and ts == list(range(1000))
    for steps in (1, 2, 3, 50, 150, 999):
        want = [int(v) for v in torch.linspace(0, 999, steps, dtype=torch.long)]
        assert sch.loop_timesteps(steps, True)[0] == want


@pytest.mark.parametrize("use_ddim,steps", [(True, 150), (True, 5), (False, 9), (True, 1)])
def test_step_records_equal_the_oracle_scalars(use_ddim, steps):
    """Every per-iteration scalar must be the exact fp32 value the reference arithmetic produces."""
    psch = M.GaussianNoiseScheduler(**R.published_scheduler_kwargs())
    osch = R.GaussianNoiseScheduler(**R.published_scheduler_kwargs())
    ts, n = psch.loop_timesteps(steps, use_ddim)
    recs = psch.step_records(ts, use_ddim)
    arr = torch.tensor(ts)
    for i, t in enumerate(reversed(arr)):
        r = recs[i]
        tt = t.expand(1)
        assert r.t == int(t)
        assert r.sqrt_recip_ac == float(osch.sqrt_recip_alphas_cumprod[t]) and r.sqrt_recipm1_ac == float(osch.sqrt_recipm1_alphas_cumprod[t])
        assert r.coef1 == float(osch.posterior_mean_coef1[t]) and r.coef2 == float(osch.posterior_mean_coef2[t])
        var = osch.estimate_variance_t(tt, 1, True, 0)
        std = torch.exp(0.5 * var)
        std[tt == 0] = 0.0
        assert r.std_fixed == float(std[0])
        if use_ddim and n - i - 1 > 0:  # diffusion_pipeline.py:297-302 verbatim arithmetic
            t_next = arr[n - i - 2]
            alpha, alpha_next = osch.alphas_cumprod[t], osch.alphas_cumprod[t_next]
            sigma = 1 * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
            c = (1 - alpha_next - sigma ** 2).sqrt()
            assert r.mode == 1 and r.ddim_sigma == float(sigma) and r.ddim_c == float(c) and r.ddim_sqrt_an == float(alpha_next.sqrt())
        else:
            assert r.mode == 0
    assert recs[-1].std_fixed == 0.0 if ts[0] == 0 else True  # last iteration is t == 0 -> std forced to 0 (Q9)
    raw = psch.upload_records(recs, "cpu")
    assert raw.numel() == 48 * len(recs)
    back = np.frombuffer(raw.numpy().tobytes(), dtype=np.float32).reshape(len(recs), 12)
    assert back[0, 0] == np.float32(recs[0].sqrt_recip_ac)


def test_shard_rows_partition():
    for n in (1, 7, 16, 128, 129):
        for w in (1, 2, 3, 8):
            parts = [D.shard_rows(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        D.shard_rows(4, 4, 4)


def test_host_noise_is_shard_invariant():
    src = lambda: M.HostNoise(lambda shape: S.PhiloxNoise(3)(torch.empty(tuple(shape))))
    full = src()
    full.begin(6, torch.device("cpu"))
    a = full.draw((6, 8, 4, 4))
    part = src()
    part.begin(2, torch.device("cpu"), sample_offset=4, global_batch=6)
    b = part.draw((2, 8, 4, 4))
    assert torch.equal(a[4:6], b)
    t0 = M.torch_cpu_noise(0)
    t0.begin(3, torch.device("cpu"))
    g = torch.Generator().manual_seed(0)
    assert torch.equal(t0.draw((3, 8, 2, 2)), torch.randn((3, 8, 2, 2), generator=g))


class _FakePipe:
    """Stands in for the GPU pipeline in the CPU multi-process test: 'images' are a deterministic function of the
    rank's noise rows and condition rows, so sharding + gather can be checked without a device."""

    def sample(self, num_samples, img_size, condition=None, noise=None, shard=None, **kw):
        lo, hi = D.shard_rows(num_samples, *shard)
        noise.begin(hi - lo, torch.device("cpu"), sample_offset=lo, global_batch=num_samples)
        x = noise.draw((hi - lo, *img_size))
        x = x + noise.draw((hi - lo, *img_size)) * 0.5
        if condition is not None:
            x = x + condition[lo:hi].reshape(-1, 1, 1, 1).float()
        return x


def _worker(rank, world, port, n, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, _, w = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    cond = torch.arange(n) % 3
    noise = M.HostNoise(lambda shape: S.PhiloxNoise(11)(torch.empty(tuple(shape))))
    full = D.sample_sharded(_FakePipe(), n, (2, 4, 4), condition=cond, noise=noise)
    torch.save(full, Path(out_dir) / f"r{rank}.pt")
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 6), (2, 5), (8, 5), (8, 16)])
def test_multi_process_gloo_shard_and_gather(tmp_path, world, n):
    """world 2 (even / uneven shards) and world 8 -- the node size of the metric -- incl. n = 5 over 8 ranks: three ranks own NO row, their
    empty shard is padded for the collective and trimmed after it (dist.gather_images)"""
    port = 29600 + os.getpid() % 300 + n + 16 * world
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    a, b = outs[0], outs[-1]
    assert all(torch.equal(a, o) for o in outs) and a.shape == (n, 2, 4, 4)
    assert [D.shard_rows(5, r, 8) for r in range(8)] == [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 5), (5, 5), (5, 5)]
    # equals the single-process run of the global batch (shard invariance)
    noise = M.HostNoise(lambda shape: S.PhiloxNoise(11)(torch.empty(tuple(shape))))
    want = _FakePipe().sample(n, (2, 4, 4), condition=torch.arange(n) % 3, noise=noise, shard=(0, 1))
    assert torch.equal(a, want)


def test_checkpoint_reader_without_lightning(tmp_path):
    """A Lightning-style .ckpt whose hyper-parameters reference `medical_diffusion.*` classes loads into the product
    classes with no reference / Lightning import."""
    from tests.util import to_product_kwargs
    # fabricate the reference's module tree just long enough to pickle class references into the file
    names = ["medical_diffusion", "medical_diffusion.models", "medical_diffusion.models.estimators", "medical_diffusion.models.estimators.unet2",
             "medical_diffusion.models.noise_schedulers", "medical_diffusion.models.noise_schedulers.gaussian_scheduler",
             "medical_diffusion.models.embedders", "medical_diffusion.models.embedders.time_embedder", "medical_diffusion.models.embedders.cond_embedders"]
    mods = {n: types.ModuleType(n) for n in names}

    def fake(modname, clsname):
        c = type(clsname, (), {})
        c.__module__, c.__qualname__ = modname, clsname
        setattr(mods[modname], clsname, c)
        return c

    RefUNet = fake("medical_diffusion.models.estimators.unet2", "UNet")
    RefSch = fake("medical_diffusion.models.noise_schedulers.gaussian_scheduler", "GaussianNoiseScheduler")
    RefTime = fake("medical_diffusion.models.embedders.time_embedder", "TimeEmbbeding")
    RefLabel = fake("medical_diffusion.models.embedders.cond_embedders", "LabelEmbedder")
    ukw = R.tiny_unet_kwargs(2, "none")
    src = M.DiffusionPipeline(M.GaussianNoiseScheduler, M.UNet, None, R.published_scheduler_kwargs(), to_product_kwargs(ukw), clip_x0=False)
    S.synth_state_dict(src.noise_estimator, "ckpt.unet.")
    hp = dict(noise_scheduler=RefSch, noise_estimator=RefUNet, latent_embedder=None, noise_scheduler_kwargs=R.published_scheduler_kwargs(),
              noise_estimator_kwargs=dict(ukw, time_embedder=RefTime, cond_embedder=RefLabel), estimator_objective="x_T", clip_x0=False,
              optimizer=torch.optim.AdamW, loss=torch.nn.L1Loss)
    sys.modules.update(mods)
    try:
        torch.save({"state_dict": src.state_dict(), "hyper_parameters": hp, "pytorch-lightning_version": "1.8.6"}, tmp_path / "last.ckpt")
    finally:
        for n in names:
            sys.modules.pop(n, None)
    pipe = M.DiffusionPipeline.load_from_checkpoint(tmp_path / "last.ckpt")
    assert isinstance(pipe.noise_estimator, M.UNet) and isinstance(pipe.noise_scheduler, M.GaussianNoiseScheduler)
    assert pipe.clip_x0 is False and pipe.estimator_objective == "x_T"
    for (k, a), (_, b) in zip(pipe.state_dict().items(), src.state_dict().items()):
        assert torch.equal(a, b), k
    assert "medical_diffusion" not in sys.modules


# ----------------------------------------------------------------------------- the real checkpoint layout
CKPT = Path(__file__).resolve().parent / "golden" / "ckpt" / "runs"


def _ref_tensors(path):
    """state_dict of a fixture checkpoint, read with the product's Lightning-free reader"""
    from medfusion_amd.checkpoint import read_checkpoint
    return read_checkpoint(path)


def test_checkpoint_written_by_the_reference_classes_loads(tmp_path):
    """tests/golden/ckpt/ was written by the REFERENCE's DiffusionPipeline / VAE through oracle/gen_ckpt_fixture.py: hyper-parameters
    captured by save_hyperparameters() hold reference class objects, torch.optim.AdamW, torch.nn.L1Loss; the VAE path baked into them is
    relative to the training cwd (scripts/train_diffusion.py:114); use_ema=True.  Three ways to the nested VAE, one result."""
    import shutil
    ck = _ref_tensors(CKPT / "tiny_diffusion" / "last.ckpt")
    hp, sd = ck["hyper_parameters"], ck["state_dict"]
    assert isinstance(hp, dict) and hp["noise_estimator"] is M.UNet and hp["latent_embedder"] is M.VAE and hp["noise_scheduler"] is M.GaussianNoiseScheduler
    assert hp["noise_estimator_kwargs"]["time_embedder"] is M.TimeEmbbeding and hp["noise_estimator_kwargs"]["cond_embedder"] is M.LabelEmbedder
    assert hp["latent_embedder_checkpoint"] == "runs/tiny_vae/last_vae.ckpt" and hp["use_ema"] is True
    assert any(k.startswith("ema_model.averaged_model.") for k in sd) and any(k.startswith("latent_embedder.") for k in sd)

    def check(pipe):
        assert pipe.use_ema and isinstance(pipe.latent_embedder, M.VAE)
        got = pipe.state_dict()
        for k, v in sd.items():
            if k.startswith(("loss_fct", "latent_embedder.perceiver", "latent_embedder.loss")):
                continue
            assert torch.equal(got[k], v), k
        assert not torch.equal(got["noise_estimator.outc.conv.conv.weight"], got["ema_model.averaged_model.outc.conv.conv.weight"])

    # 1. the baked relative path resolved against the checkpoint's own `runs/` tree
    check(M.DiffusionPipeline.load_from_checkpoint(CKPT / "tiny_diffusion" / "last.ckpt"))
    # 2. the pipeline checkpoint alone (what usually reaches a sampling box): VAE hyper-parameters inferred from the tensor shapes
    lone = tmp_path / "last.ckpt"
    shutil.copy(CKPT / "tiny_diffusion" / "last.ckpt", lone)
    pipe = M.DiffusionPipeline.load_from_checkpoint(lone)
    check(pipe)
    from medfusion_amd.checkpoint import infer_vae_kwargs
    kw = infer_vae_kwargs({k[len("latent_embedder."):]: v for k, v in sd.items() if k.startswith("latent_embedder.")})
    assert kw["hid_chs"] == [32, 32, 32, 32] and kw["emb_channels"] == 8 and kw["strides"] == [1, 2, 2, 2] and kw["deep_supervision"] == 1
    # 3. an explicit path
    check(M.DiffusionPipeline.load_from_checkpoint(lone, latent_embedder_checkpoint=str(CKPT / "tiny_vae" / "last_vae.ckpt")))
    # the VAE checkpoint on its own (VAE.load_from_checkpoint of scripts/helpers/sample_latent_embedder.py:50)
    from medfusion_amd.checkpoint import load_module_from_checkpoint
    vae = load_module_from_checkpoint(M.VAE, CKPT / "tiny_vae" / "last_vae.ckpt")
    assert torch.equal(vae.state_dict()["outc.conv.weight"], sd["latent_embedder.outc.conv.weight"])


def test_checkpoint_reader_refuses_foreign_globals_and_wrong_architectures(tmp_path):
    """the unpickler resolves nothing outside its allow-list (a checkpoint is data), and a state dict that does not fit raises"""
    import pickle
    from medfusion_amd.checkpoint import load_module_from_checkpoint, read_checkpoint
    evil = tmp_path / "evil.ckpt"
    torch.save({"state_dict": {}, "hyper_parameters": {"f": os.system}}, evil)
    with pytest.raises(pickle.UnpicklingError):
        read_checkpoint(evil)
    # the escape hatches of an allow-list -- a nested unrestricted unpickle (torch.storage._load_from_bytes is
    # torch.load(..., weights_only=False)), and getattr / object (getattr(object, "__subclasses__")() reaches every class)
    import io
    from medfusion_amd.checkpoint import _RemapUnpickler

    def refuses(module, name):
        blob = pickle.PROTO + bytes([2]) + pickle.GLOBAL + f"{module}\n{name}\n".encode() + pickle.STOP
        with pytest.raises(pickle.UnpicklingError, match="allow-list"):
            _RemapUnpickler(io.BytesIO(blob)).load()

    for module, name in (("torch.storage", "_load_from_bytes"), ("builtins", "getattr"), ("builtins", "object"), ("builtins", "eval"),
                         ("os", "system"), ("posix", "system"), ("subprocess", "Popen"), ("torch", "load"), ("builtins", "__import__")):
        refuses(module, name)
    inner = io.BytesIO()
    torch.save({"f": os.getcwd}, inner)                      # what the PoC smuggled through _load_from_bytes
    nested = tmp_path / "nested.ckpt"
    torch.save({"state_dict": {}, "hyper_parameters": {"payload": inner.getvalue()}}, nested, pickle_protocol=4)
    assert isinstance(read_checkpoint(nested)["hyper_parameters"]["payload"], bytes)   # bytes stay bytes: nothing unpickles them
    with pytest.raises(RuntimeError, match="missing"):
        load_module_from_checkpoint(M.VAE, CKPT / "tiny_vae" / "last_vae.ckpt", deep_supervision=2)   # a head the checkpoint has no weights for


def test_seeded_fill_matches_the_test_fill():
    """medfusion_amd.published.seeded_fill (bench / harness weights) == oracle.synth.synth_state_dict (what the parity tests give the oracle)"""
    import torch.nn as nn
    from medfusion_amd import published as P
    a = nn.Sequential(nn.Conv2d(8, 16, 3), nn.GroupNorm(4, 16), nn.Linear(16, 4), nn.Embedding(3, 4))
    b = nn.Sequential(nn.Conv2d(8, 16, 3), nn.GroupNorm(4, 16), nn.Linear(16, 4), nn.Embedding(3, 4))
    S.synth_state_dict(a, "x.embedding.")
    P.seeded_fill(b, "x.embedding.")
    for (k, u), (_, v) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(u, v), k
    assert P.published_unet_kwargs(3)["cond_embedder_kwargs"]["num_classes"] == 3
    ok, pk = R.published_unet_kwargs(2), P.published_unet_kwargs(2)
    assert {k: v for k, v in ok.items() if "embedder" not in k} == {k: v for k, v in pk.items() if "embedder" not in k}
    assert R.published_vae_kwargs(8) == P.published_vae_kwargs(8) and R.published_scheduler_kwargs() == P.published_scheduler_kwargs()


def test_default_noise_is_stateful_and_reseedable():
    """the default Philox source takes a fresh key from torch's CPU generator at every begin(): successive calls differ,
    torch.manual_seed reproduces them; an explicit seed is a fixed key"""
    dev = torch.device("cpu")
    torch.manual_seed(0)
    a, b = M.PhiloxDeviceNoise(), M.PhiloxDeviceNoise()
    a.begin(2, dev); b.begin(2, dev)
    k1, k2 = a._seed, b._seed
    a.begin(2, dev)
    assert len({k1, k2, a._seed}) == 3
    torch.manual_seed(0)
    c = M.PhiloxDeviceNoise()
    c.begin(2, dev)
    assert c._seed == k1
    e = M.PhiloxDeviceNoise(5)
    e.begin(2, dev)
    assert e._seed == 5


def _forced_worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MEDFUSION_FORCE_COLLECTIVE="1")
    r, _, w = D.init_from_env(backend="gloo")
    assert (r, w) == (0, 1) and dist.is_initialized()       # a process group although the world is 1
    calls = []
    real = dist.all_gather
    dist.all_gather = lambda bufs, src, *a, **k: (calls.append(len(bufs)), real(bufs, src, *a, **k))[1]
    noise = M.HostNoise(lambda shape: S.PhiloxNoise(11)(torch.empty(tuple(shape))))
    full = D.sample_sharded(_FakePipe(), 5, (2, 4, 4), condition=torch.arange(5) % 3, noise=noise)
    assert calls == [1]                                       # the gather went through the collective, not the world-1 shortcut
    torch.save(full, Path(out_dir) / "forced.pt")
    dist.destroy_process_group()


def test_forced_collective_on_a_one_rank_group(tmp_path):
    """MEDFUSION_FORCE_COLLECTIVE=1 (dist.init_from_env / gather_images): the 1-rank group runs the same all-gather the N-rank job runs --
    the switch tests/test_multiproc_gpu.py::test_rccl_world1_device_allgather uses on the single-GPU box, here over gloo"""
    mp.spawn(_forced_worker, args=(29950 + os.getpid() % 40, str(tmp_path)), nprocs=1, join=True)
    noise = M.HostNoise(lambda shape: S.PhiloxNoise(11)(torch.empty(tuple(shape))))
    want = _FakePipe().sample(5, (2, 4, 4), condition=torch.arange(5) % 3, noise=noise, shard=(0, 1))
    assert torch.equal(torch.load(tmp_path / "forced.pt"), want)
    assert os.environ.get("MEDFUSION_FORCE_COLLECTIVE") is None   # the child's environment did not leak

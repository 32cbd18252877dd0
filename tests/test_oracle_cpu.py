"""The oracle against the golden vectors produced by the REAL reference (oracle/gen_golden.py) + known answers."""
import numpy as np
import pytest
import torch

from oracle import restate as R
from oracle import synth as S
from tests.util import T, gold, relerr

TOL = 2e-6  # same ATen ops as the reference; only the thread count (summation order) may differ from the fixture run


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    kat = [((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
           ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
           ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1))]
    for ctr, key, want in kat:
        got = S.philox4x32_10(*ctr, *key)
        assert tuple(int(v) for v in got) == want


def test_philox_normal_moments_and_shard_invariance():
    full = S.philox_normal(5, 3, np.arange(8), 4096)
    assert abs(full.mean()) < 0.02 and abs(full.std() - 1) < 0.02
    part = S.philox_normal(5, 3, np.arange(4, 8), 4096)
    assert np.array_equal(full[4:], part)
    assert not np.array_equal(S.philox_normal(5, 4, np.arange(8), 4096), full)


def test_scheduler_tables_and_kats():
    g = gold("sched_published")
    sch = R.GaussianNoiseScheduler(**R.published_scheduler_kwargs())
    for k, v in sch.state_dict().items():
        assert torch.equal(v, T(g[f"tab_{k}"])), k
    # SURVEY §8a S1 known answers
    assert float(sch.betas[0]) == 0.0020000000949949026
    assert float(sch.alphas_cumprod[-1]) == 7.470230775652453e-05
    assert float(sch.sqrt_recip_alphas_cumprod[-1]) == 115.69990539550781
    assert float(sch.posterior_mean_coef1[0]) == 1.0 and float(sch.posterior_mean_coef2[0]) == 0.0 and float(sch.posterior_variance[0]) == 0.0
    for tag, kw in (("cosine", dict(timesteps=1000)), ("linear", dict(timesteps=200, schedule_strategy="linear"))):
        g = gold(f"sched_{tag}")
        s2 = R.GaussianNoiseScheduler(**kw)
        for k, v in s2.state_dict().items():
            assert torch.equal(v, T(g[f"tab_{k}"])), (tag, k)
        xt = s2.estimate_x_t(T(g["x_0"]), T(g["t"]), T(g["x_T"]))
        assert torch.equal(xt, T(g["x_t"]))
        s2.noise_fn = S.PhiloxNoise(7)
        pr, x0 = s2.estimate_x_t_prior_from_x_T(xt, T(g["t"]), T(g["x_T"]), clip_x0=False)
        assert torch.equal(pr, T(g["x_t_prior"])) and torch.equal(x0, T(g["x_0_est"]))


def test_embedders():
    g = gold("embedders")
    assert relerr(R.SinusoidalPosEmb(20, max_period=10)(T(g["t_sin"])), T(g["sin20"])) <= TOL
    te = R.TimeEmbbeding(64)
    S.synth_state_dict(te, "time64.")
    assert relerr(te(T(g["t_long"])), T(g["time64"])) <= TOL
    assert relerr(te(T(g["t_float"])), T(g["time64_float"])) <= TOL
    le = R.LabelEmbedder(64, 3)
    S.synth_state_dict(le, "label64.")
    assert torch.equal(le(T(g["cond"])), T(g["label64"]))


@torch.no_grad()
def test_vae_forward():
    """VAE.forward (latent_embedders.py:771-790): reconstruction, deep-supervision outputs, KL term against the reference's own output"""
    g = gold("vae_forward")
    m = R.VAE(**dict(R.tiny_vae_kwargs(), deep_supervision=2)).eval()
    S.synth_state_dict(m, "vae_fwd.")
    nz = S.PhiloxNoise(int(g["seed"]))
    m.quantizer.noise_fn = lambda shape, device: nz(torch.empty(shape))
    out, hor, kl = m(T(g["img"]))
    assert relerr(out, T(g["out"])) <= TOL and len(hor) == 2
    assert relerr(hor[0], T(g["hor0"])) <= TOL and relerr(hor[1], T(g["hor1"])) <= TOL
    assert abs(float(kl) - float(g["kl"][0])) <= 1e-5 * abs(float(g["kl"][0]))


@torch.no_grad()
def test_learned_sinusoidal_posemb():
    """time_embedder.py:31-49 against the reference's own output (even and odd emb_dim): the first column is t itself, bit for bit"""
    g = gold("learned_posemb")
    for e in (16, 33):
        m = R.LearnedSinusoidalPosEmb(e)
        S.synth_state_dict(m, f"learned{e}.")
        assert torch.equal(m.weights.detach(), T(g[f"w{e}"]))
        y = m(T(g["t"]))
        assert y.shape == (5, e + 1) and torch.equal(y[:, 0], T(g["t"])) and relerr(y, T(g[f"y{e}"])) <= TOL


GN32 = ("GROUP", {"num_groups": 32, "affine": True})
GN8 = ("GROUP", {"num_groups": 8, "affine": True})


@torch.no_grad()
def test_blocks():
    g = gold("blocks")
    blk = R.UnetResBlock(32, 64, 3, 1, GN32, True, 48).eval()
    S.synth_state_dict(blk, "resblk.")
    assert relerr(blk(T(g["res_x"]).clone(), T(g["res_emb"])), T(g["res_y"])) <= TOL
    bb = R.UnetBasicBlock(32, 32, 3, 1, GN8, True, 48).eval()
    S.synth_state_dict(bb, "basicblk.")
    assert relerr(bb(T(g["res_x"]).clone(), T(g["res_emb"])), T(g["basic_y"])) <= TOL
    d = R.BasicDown(32, 32, 3, 2)
    S.synth_state_dict(d, "down.")
    assert relerr(d(T(g["down_x"])), T(g["down_y"])) <= TOL
    u = R.BasicUp(32, 32, 2, 2)
    S.synth_state_dict(u, "up.")
    assert relerr(u(T(g["up_x"])), T(g["up_y"])) <= TOL


@torch.no_grad()
def test_attention():
    g = gold("attention")
    x, e = T(g["x"]), T(g["emb"])
    for tag, emb_dim, emb in (("self", None, None), ("cross", 48, e)):
        m = R.LinearTransformer(32, 32, 4, 8, GN8, emb_dim)
        S.synth_state_dict(m, f"lt_{tag}.")
        assert relerr(m(x, emb), T(g[f"lt_{tag}_y"])) <= TOL
    m = R.SpatialTransformer(32, 32, 3, 32, GN8, None, 1)
    S.synth_state_dict(m, "st_self.")
    assert relerr(m(x), T(g["st_self_y"])) <= TOL
    m = R.SpatialTransformer(32, 32, 4, 8, GN8, 48, 1)
    S.synth_state_dict(m, "st_emb.")
    assert relerr(m(x, e), T(g["st_emb_y"])) <= TOL


UNET_CASES = {
    "none": lambda: R.tiny_unet_kwargs(2, "none"),
    "linear": lambda: R.tiny_unet_kwargs(2, "linear"),
    "spatial": lambda: R.tiny_unet_kwargs(3, "spatial"),
    "mixed": lambda: R.tiny_unet_kwargs(2, ["none", "none", "linear", "spatial"]),
    "basicblk_var_selfcond": lambda: R.tiny_unet_kwargs(2, "none", use_res_block=False, estimate_variance=True, use_self_conditioning=True,
                                                        deep_supervision=True),
}


@pytest.mark.parametrize("tag", list(UNET_CASES))
@torch.no_grad()
def test_unet_tiny(tag):
    g = gold(f"unet_tiny_{tag}")
    m = R.UNet(**UNET_CASES[tag]()).eval()
    S.synth_state_dict(m, f"unet_{tag}.")
    y, ver = m(T(g["x"]), T(g["t"]), T(g["cond"]))
    assert relerr(y, T(g["y"])) <= TOL
    for i, v in enumerate(ver):
        assert relerr(v, T(g[f"y_ver{i}"])) <= TOL
    yu, _ = m(T(g["x"]), T(g["t"]), None)
    assert relerr(yu, T(g["y_uncond"])) <= TOL


@torch.no_grad()
def test_nonlearnable_down_up_and_unet():
    """learnable_interpolation=False (conv_blocks.py:57-63 AvgPool, :128-130 nearest-exact) against what the REFERENCE computed
    (oracle/gen_golden.py::case_nonlearnable)"""
    g = gold("blocks_nonlearnable")
    d, u = R.BasicDown(32, 32, 3, 2, learnable_interpolation=False), R.BasicUp(32, 32, 2, 2, learnable_interpolation=False)
    for tag in ("even", "odd"):
        assert torch.equal(d(T(g[f"down_{tag}_x"])), T(g[f"down_{tag}_y"]))
    assert torch.equal(u(T(g["up_x"])), T(g["up_y"]))
    g = gold("unet_tiny_nonlearnable")
    m = R.UNet(**R.tiny_unet_kwargs(2, "none", learnable_interpolation=False)).eval()
    S.synth_state_dict(m, "unet_nonlearnable.")
    y, _ = m(T(g["x"]), T(g["t"]), T(g["cond"]))
    assert relerr(y, T(g["y"])) <= TOL


@torch.no_grad()
def test_use_res_down_up():
    """BasicDown / BasicUp with use_res=True (PixelUnshuffle / PixelShuffle skips, conv_blocks.py:54-55,68-69,114-115,125-126) against what the
    REFERENCE computed (oracle/gen_golden.py::case_use_res)"""
    g = gold("blocks_use_res")
    d, u = R.BasicDown(32, 128, 3, 2, use_res=True), R.BasicUp(128, 32, 2, 2, use_res=True)
    S.synth_state_dict(d, "ur_down.")
    S.synth_state_dict(u, "ur_up.")
    assert torch.equal(d(T(g["down_x"])), T(g["down_y"]))
    assert torch.equal(u(T(g["up_x"])), T(g["up_y"]))


REFTEST_KW = dict(in_ch=3, out_ch=3, spatial_dims=2, hid_chs=[32, 64, 128, 256], kernel_sizes=[1, 3, 3, 3], strides=[1, 2, 2, 2],
                  time_embedder=R.TimeEmbbeding, time_embedder_kwargs={"emb_dim": 64}, cond_embedder=R.LabelEmbedder,
                  cond_embedder_kwargs={"emb_dim": 64, "num_classes": 2}, deep_supervision=True, use_res_block=True, use_attention="linear")


@torch.no_grad()
def test_unet_reference_test_config():
    g = gold("unet_reftest_cfg")
    m = R.UNet(**REFTEST_KW).eval()
    S.synth_state_dict(m, "unet_reftest.")
    y, ver = m(T(g["x"]), T(g["t"]), T(g["cond"]))
    assert relerr(y, T(g["y"])) <= TOL
    for i, v in enumerate(ver):
        assert relerr(v, T(g[f"y_ver{i}"])) <= TOL


@torch.no_grad()
def test_vae_tiny():
    g = gold("vae_tiny")
    m = R.VAE(**R.tiny_vae_kwargs()).eval()
    S.synth_state_dict(m, "vae_tiny.")
    assert relerr(m.decode(T(g["z"])), T(g["x_dec"])) <= TOL
    nz = S.PhiloxNoise(int(g["enc_seed"]))
    m.quantizer.noise_fn = lambda shape, device: nz(torch.empty(shape))
    assert relerr(m.encode(T(g["img"])), T(g["z_enc"])) <= TOL


def build_oracle_pipe(unet_kw, vae_kw, tag, clip_x0=False, objective="x_T", estimate_variance=False, self_cond=False, sched_kw=None):
    ok = dict(unet_kw, estimate_variance=estimate_variance, use_self_conditioning=self_cond)
    pipe = R.DiffusionPipeline(R.GaussianNoiseScheduler(**(sched_kw or R.published_scheduler_kwargs())), R.UNet(**ok),
                               R.VAE(**vae_kw) if vae_kw else None, estimator_objective=objective, estimate_variance=estimate_variance,
                               use_self_conditioning=self_cond, clip_x0=clip_x0).eval()
    S.synth_state_dict(pipe.noise_estimator, f"{tag}.unet.")
    if vae_kw:
        S.synth_state_dict(pipe.latent_embedder, f"{tag}.vae.")
    return pipe


SAMPLE_CASES = {
    # name: (pipe builder args, sample kwargs)
    "sample_tiny_ddim5_uncond": (dict(unet=(3, "none"), tag="pipe_tiny"), dict(steps=5, use_ddim=True)),
    "sample_tiny_ddim6_cfg8": (dict(unet=(3, "none"), tag="pipe_tiny"), dict(steps=6, use_ddim=True, guidance_scale=8, un_cond=None)),
    "sample_tiny_ddim4_g1": (dict(unet=(3, "none"), tag="pipe_tiny"), dict(steps=4, use_ddim=True, guidance_scale=1.0, un_cond=None)),
    "sample_tiny_ddpm7": (dict(unet=(3, "none"), tag="pipe_tiny"), dict(steps=7, use_ddim=False)),
    "sample_tiny_x0obj_clip": (dict(unet=(None, "none"), tag="pipe_tiny_x0", clip_x0=True, objective="x_0"), dict(steps=5, use_ddim=True)),
    "sample_tiny_attn": (dict(unet=(2, ["none", "none", "linear", "spatial"]), tag="pipe_tiny_attn"), dict(steps=3, use_ddim=True, guidance_scale=2.0)),
    "sample_tiny_var_selfcond": (dict(unet=(2, "none"), tag="pipe_tiny_var", estimate_variance=True, self_cond=True),
                                 dict(steps=4, use_ddim=False, guidance_scale=1.0)),
}


def sample_case_pipe(name):
    spec, kw = SAMPLE_CASES[name]
    spec = dict(spec)
    ncls, att = spec.pop("unet")
    tag = spec.pop("tag")
    return build_oracle_pipe(R.tiny_unet_kwargs(ncls, att), R.tiny_vae_kwargs(), tag, **spec), dict(kw)


@pytest.mark.parametrize("name", list(SAMPLE_CASES))
@torch.no_grad()
def test_sample_tiny(name):
    g = gold(name)
    pipe, kw = sample_case_pipe(name)
    if "condition" in g:
        kw["condition"] = T(g["condition"])
    pipe.set_noise_fn(S.PhiloxNoise(int(g["seed"])))
    trace = []
    img = pipe.sample(int(g["n"]), tuple(int(v) for v in g["size"]), trace=trace, **kw)
    assert pipe.noise_fn.draw == int(g["draws"])  # Q3 draw count
    assert relerr(img, T(g["image"])) <= 5e-5  # thread-count (summation-order) drift through the recurrent loop
    assert relerr(trace[0][0], T(g["x0_step0"])) <= 5e-5
    # torch's own generator (reference harness recipe torch.manual_seed(0))
    pipe.set_noise_fn(None)
    torch.manual_seed(0)
    img2 = pipe.sample(int(g["n"]), tuple(int(v) for v in g["size"]), **kw)
    assert relerr(img2, T(g["image_torchseed0"])) <= 5e-5


def test_eta_raises_like_reference():
    pipe, kw = sample_case_pipe("sample_tiny_ddim5_uncond")
    with pytest.raises(TypeError):
        pipe.sample(1, (8, 8, 8), eta=0.0, **kw)


@torch.no_grad()
def test_cfg1_published_architecture_64px():
    """BASELINE.json configs[0]: 64x64, 50 DDIM steps, published architecture, CPU path."""
    g = gold("cfg1_published_64px")
    pipe = build_oracle_pipe(R.published_unet_kwargs(2), R.published_vae_kwargs(8), "published")
    y, _ = pipe.noise_estimator(T(g["unet_x"]), T(g["unet_t"]), T(g["unet_c"]))
    assert relerr(y, T(g["unet_y"])) <= 1e-5
    pipe.set_noise_fn(S.PhiloxNoise(int(g["seed"])))
    img = pipe.sample(2, (8, 8, 8), steps=int(g["steps"]), use_ddim=True)
    assert relerr(img, T(g["image"])) <= 1e-4

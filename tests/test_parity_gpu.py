"""Model- and pipeline-level parity on a real MI355X, through the product API (which calls the C-ABI):
 * against the committed golden vectors (outputs of the REAL reference, tests/golden/),
 * against the oracle on the same seeded inputs at the published sizes,
 * and through size-independent properties at BASELINE.json's full sizes (batch-row independence,
   shard invariance, determinism).

Tolerance (fp32 path, stated per SURVEY §8d): max-norm relative error <= 1e-4 everywhere -- single network
evaluations, short and full-length trajectories, with and without classifier-free guidance (measured: 1e-6 .. 1.3e-5,
profiles/r03_guided_trajectory.txt); the 150-iteration guided trajectory of the published model is held to 5e-5.
fp32-MFMA accumulates in a different order than ATen's CPU kernels; nothing on these tests is reduced-precision
(the two opt-in reduced modes have their own tests and tolerances).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import medfusion_amd as M
from medfusion_amd import kernels as K
from oracle import restate as R
from oracle import synth as S
from tests.test_oracle_cpu import REFTEST_KW, SAMPLE_CASES, UNET_CASES, build_oracle_pipe
from tests.util import T, gold, oracle_fp64_drift, oracle_noise, relerr, relerr_rms, relerr_rows, to_product_kwargs

TOL = 1e-4
# guided cases (classifier-free guidance 2 .. 8) on the tiny synthetic models and short trajectories of the published widths: measured on
# MI355X 2e-6 .. 1.3e-5 on all three arithmetics (profiles/r03_guided_trajectory.txt) -- the path's own 1e-4 holds, the blanket 1e-3 of
# rounds 1-2 is gone
GUIDED_TINY_TOL = TOL
# cold diffusion re-derives x_T from the x_0 estimate at every iteration (a division by sqrt(1 - alpha_bar), small at the first timesteps of
# the tiny model's 4-iteration loop): the case is ILL-CONDITIONED -- the fp32 oracle lands 1.7e-4 from its own fp64 evaluation on the DDIM
# variant (8e-7 on the DDPM one; profiles/r06_cold_diffusion_conditioning.txt).  The test measures that drift itself (tests/util.py::
# oracle_fp64_drift) and holds the product to max(TOL, COLD_DRIFT_FACTOR x drift): no hand-widened constant.
COLD_DRIFT_FACTOR = 2.0
_ORACLE_CACHE = {}
GN32 = ("GROUP", {"num_groups": 32, "affine": True})
GN8 = ("GROUP", {"num_groups": 8, "affine": True})
ACT = ("SWISH", {})


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(params=[(5, 1, 1), (5, 2, 1), (1, 1, 0), (1, 1, 2), (0, 1, 1)], ids=["f16x2", "f16x2_winograd_everywhere", "split3", "split3_winograd_everywhere", "fp32mfma"],
                autouse=True)
def conv_precision(request):
    """every parity test runs on all three fp32-class conv arithmetics (MF_CONV_FP32_F16X2, MF_CONV_FP32_SPLIT3_W3, MF_CONV_FP32), same tolerances;
    the default arithmetic twice: as shipped (the Winograd form where mf_wino_preferred says so) and with that form wherever the library can run it
    (blocks.WINOGRAD = 2); the exact bf16-triplet arithmetic twice as well: on the direct form (WINOGRAD_F32 = 0) and with the round-6 Winograd form of
    the exact arithmetics wherever the library can run it (WINOGRAD_F32 = 2); the fp32 MFMA chain always runs direct"""
    from medfusion_amd import blocks as BLK
    old = BLK.CONV_PRECISION, BLK.WINOGRAD, BLK.WINOGRAD_F32
    BLK.CONV_PRECISION, BLK.WINOGRAD, BLK.WINOGRAD_F32 = request.param
    yield request.param[0]
    BLK.CONV_PRECISION, BLK.WINOGRAD, BLK.WINOGRAD_F32 = old


def nhwc(x, dev):
    return K.nchw_to_nhwc(x.to(dev))


def nchw(y):
    return K.nhwc_to_nchw(y).cpu()


def test_blocks_golden(dev):
    from medfusion_amd import blocks as B
    g = gold("blocks")
    blk = B.UnetResBlock(2, 32, 64, 3, 1, GN32, ACT, 0.0, 48)
    S.synth_state_dict(blk, "resblk.")
    blk.to(dev)
    emb = blk.local_embed(T(g["res_emb"]).to(dev))
    assert relerr(nchw(blk(nhwc(T(g["res_x"]), dev), emb)), T(g["res_y"])) < TOL
    bb = B.UnetBasicBlock(2, 32, 32, 3, 1, GN8, ACT, None, 48)
    S.synth_state_dict(bb, "basicblk.")
    bb.to(dev)
    assert relerr(nchw(bb(nhwc(T(g["res_x"]), dev), bb.local_embed(T(g["res_emb"]).to(dev)))), T(g["basic_y"])) < TOL
    d = B.BasicDown(2, 32, 32, 3, 2)
    S.synth_state_dict(d, "down.")
    assert relerr(nchw(d.to(dev)(nhwc(T(g["down_x"]), dev))), T(g["down_y"])) < TOL
    u = B.BasicUp(2, 32, 32, 2, 2)
    S.synth_state_dict(u, "up.")
    assert relerr(nchw(u.to(dev)(nhwc(T(g["up_x"]), dev))), T(g["up_y"])) < TOL


def test_nonlearnable_down_up_golden(dev):
    """learnable_interpolation=False: AvgPool (bit-exact: same window order, one division) and nearest-exact x2 (a copy) against the
    reference's outputs; a tiny UNet built that way against the reference's"""
    from medfusion_amd import blocks as B
    g = gold("blocks_nonlearnable")
    d, u = B.BasicDown(2, 32, 32, 3, 2, learnable_interpolation=False), B.BasicUp(2, 32, 32, 2, 2, learnable_interpolation=False)
    assert len(list(d.parameters())) == 0 and len(list(u.parameters())) == 0
    for tag in ("even", "odd"):
        assert torch.equal(nchw(d(nhwc(T(g[f"down_{tag}_x"]), dev))), T(g[f"down_{tag}_y"])), tag
    assert torch.equal(nchw(u(nhwc(T(g["up_x"]), dev))), T(g["up_y"]))
    g = gold("unet_tiny_nonlearnable")
    m = M.UNet(**to_product_kwargs(R.tiny_unet_kwargs(2, "none", learnable_interpolation=False)))
    S.synth_state_dict(m, "unet_nonlearnable.")
    m.to(dev)
    y, _ = m(T(g["x"]).to(dev), T(g["t"]).to(dev), T(g["cond"]).to(dev))
    assert relerr(y, T(g["y"])) < TOL


def test_use_res_down_up_golden(dev):
    """BasicDown / BasicUp with use_res=True: the stride-2 / up convolution plus the PixelUnshuffle / PixelShuffle skip of the input
    (mf_pixel_unshuffle2_add_nhwc_f32, mf_pixel_shuffle2_add_nhwc_f32) against the reference's outputs; the sum carries no stale operand mirror"""
    from medfusion_amd import blocks as B
    from medfusion_amd import kernels as K
    g = gold("blocks_use_res")
    d, u = B.BasicDown(2, 32, 128, 3, 2, use_res=True), B.BasicUp(2, 128, 32, 2, 2, use_res=True)
    S.synth_state_dict(d, "ur_down.")
    S.synth_state_dict(u, "ur_up.")
    yd = d.to(dev)(nhwc(T(g["down_x"]), dev))
    assert relerr(nchw(yd), T(g["down_y"])) < TOL
    yu = u.to(dev)(nhwc(T(g["up_x"]), dev))
    assert relerr(nchw(yu), T(g["up_y"])) < TOL
    for y in (yd, yu):     # whoever reads the sum as a convolution operand measures IT, not the convolution's output before the add
        assert torch.equal(K.bound_of(y), y.abs().amax(dim=(1, 2, 3)))


def test_attention_golden(dev):
    from medfusion_amd import blocks as B
    g = gold("attention")
    x, e = nhwc(T(g["x"]), dev), T(g["emb"]).to(dev)
    for tag, emb_dim, emb in (("self", None, None), ("cross", 48, e)):
        m = B.LinearTransformer(2, 32, 32, 4, 8, GN8, None, emb_dim)
        S.synth_state_dict(m, f"lt_{tag}.")
        assert relerr(nchw(m.to(dev)(x, emb)), T(g[f"lt_{tag}_y"])) < TOL, tag
    m = B.SpatialTransformer(2, 32, 32, 3, 32, GN8, None, None, 1)
    S.synth_state_dict(m, "st_self.")
    assert relerr(nchw(m.to(dev)(x)), T(g["st_self_y"])) < TOL
    m = B.SpatialTransformer(2, 32, 32, 4, 8, GN8, None, 48, 1)
    S.synth_state_dict(m, "st_emb.")
    assert relerr(nchw(m.to(dev)(x, e)), T(g["st_emb_y"])) < TOL


def test_embedders_golden(dev):
    g = gold("embedders")
    te = M.TimeEmbbeding(64)
    S.synth_state_dict(te, "time64.")
    te.to(dev)
    assert relerr(te(T(g["t_long"]).to(dev)), T(g["time64"])) < TOL
    assert relerr(te(T(g["t_float"]).to(dev)), T(g["time64_float"])) < TOL
    le = M.LabelEmbedder(64, 3)
    S.synth_state_dict(le, "label64.")
    assert torch.equal(le.to(dev)(T(g["cond"]).to(dev)).cpu(), T(g["label64"]))


def test_vae_forward_golden(dev):
    """VAE.forward -- the evaluation-time reconstruction pass (latent_embedders.py:771-790): output, both deep-supervision outputs and the KL term
    of the quantizer against what the REFERENCE returns for the same weights, image and injected noise"""
    g = gold("vae_forward")
    m = M.VAE(**dict(R.tiny_vae_kwargs(), deep_supervision=2))
    S.synth_state_dict(m, "vae_fwd.")
    m.to(dev).eval()
    out, hor, kl = m(T(g["img"]).to(dev), noise=oracle_noise(int(g["seed"])))
    assert out.shape == (2, 3, 32, 32) and len(hor) == 2
    assert relerr(out, T(g["out"])) < TOL
    assert relerr(hor[0], T(g["hor0"])) < TOL and relerr(hor[1], T(g["hor1"])) < TOL
    assert abs(float(kl) - float(g["kl"][0])) < 1e-5 * abs(float(g["kl"][0]))
    # decode(encode(x)) is the same reconstruction
    z = m.encode(T(g["img"]).to(dev), noise=oracle_noise(int(g["seed"])))
    assert relerr(m.decode(z), T(g["out"])) < TOL


def test_learned_sinusoidal_posemb_golden(dev, conv_precision):
    """LearnedSinusoidalPosEmb (time_embedder.py:31-49) against the reference's output; and TimeEmbbeding refuses it the way the reference's
    first nn.Linear does (emb_dim + 1 features into Linear(emb_dim, ...))"""
    if conv_precision != 5:
        pytest.skip("runs once")
    g = gold("learned_posemb")
    for e in (16, 33):
        m = M.LearnedSinusoidalPosEmb(e)
        S.synth_state_dict(m, f"learned{e}.")
        y = m.to(dev)(T(g["t"]).to(dev))
        assert y.shape == (5, e + 1) and torch.equal(y[:, 0].cpu(), T(g["t"]))
        # the angle 2 pi t w reaches ~6e3 rad: one fp32 ulp of it is 5e-4, so sin / cos agree with ATen's to that absolute level, not to 1e-6
        assert float((y.cpu() - T(g[f"y{e}"])).abs().max()) < 2e-3
        small = T(g["t"])[:3]                                  # |angle| < ~10: sinf / cosf agree to a few ulp
        ys = m(small.to(dev)).cpu()
        assert float((ys - T(g[f"y{e}"])[:3]).abs().max()) < 1e-5
    te = M.TimeEmbbeding(64, pos_embedder=M.LearnedSinusoidalPosEmb, pos_embedder_kwargs={"emb_dim": 16}).to(dev)
    with pytest.raises(RuntimeError, match="shapes cannot be multiplied"):
        te(torch.tensor([1.0, 2.0], device=dev))


@pytest.mark.parametrize("tag", list(UNET_CASES))
def test_unet_tiny_golden(dev, tag):
    g = gold(f"unet_tiny_{tag}")
    m = M.UNet(**to_product_kwargs(UNET_CASES[tag]()))
    S.synth_state_dict(m, f"unet_{tag}.")
    m.to(dev)
    x, t, c = T(g["x"]).to(dev), T(g["t"]).to(dev), T(g["cond"]).to(dev)
    y, ver = m(x, t, c)
    assert relerr(y, T(g["y"])) < TOL
    for i, v in enumerate(ver):
        assert relerr(v, T(g[f"y_ver{i}"])) < TOL
    yu, _ = m(x, t, None)
    assert relerr(yu, T(g["y_uncond"])) < TOL


def test_unet_reference_test_config_golden(dev):
    g = gold("unet_reftest_cfg")
    m = M.UNet(**to_product_kwargs(REFTEST_KW))
    S.synth_state_dict(m, "unet_reftest.")
    m.to(dev)
    y, ver = m(T(g["x"]).to(dev), T(g["t"]).to(dev), T(g["cond"]).to(dev))
    assert relerr(y, T(g["y"])) < TOL
    for i, v in enumerate(ver):
        assert relerr(v, T(g[f"y_ver{i}"])) < TOL


def test_vae_tiny_golden(dev):
    g = gold("vae_tiny")
    m = M.VAE(**R.tiny_vae_kwargs())
    S.synth_state_dict(m, "vae_tiny.")
    m.to(dev)
    assert relerr(m.decode(T(g["z"]).to(dev)), T(g["x_dec"])) < TOL
    z = m.encode(T(g["img"]).to(dev), noise=oracle_noise(int(g["enc_seed"])))
    assert relerr(z, T(g["z_enc"])) < TOL


def build_product_pipe(unet_kw, vae_kw, tag, dev, clip_x0=False, objective="x_T", estimate_variance=False, self_cond=False):
    pipe = M.DiffusionPipeline(noise_scheduler=M.GaussianNoiseScheduler, noise_estimator=M.UNet, latent_embedder=None,
                               noise_scheduler_kwargs=R.published_scheduler_kwargs(), noise_estimator_kwargs=to_product_kwargs(unet_kw),
                               estimator_objective=objective, estimate_variance=estimate_variance, use_self_conditioning=self_cond, clip_x0=clip_x0)
    S.synth_state_dict(pipe.noise_estimator, f"{tag}.unet.")
    if vae_kw:
        pipe.latent_embedder = M.VAE(**vae_kw)
        S.synth_state_dict(pipe.latent_embedder, f"{tag}.vae.")
    return pipe.to(dev).eval()


@pytest.mark.parametrize("name", list(SAMPLE_CASES))
def test_sample_tiny_golden(dev, name):
    g = gold(name)
    spec, kw = SAMPLE_CASES[name]
    spec, kw = dict(spec), dict(kw)
    ncls, att = spec.pop("unet")
    tag = spec.pop("tag")
    pipe = build_product_pipe(R.tiny_unet_kwargs(ncls, att), R.tiny_vae_kwargs(), tag, dev, **spec)
    if "condition" in g:
        kw["condition"] = T(g["condition"]).to(dev)
    noise = oracle_noise(int(g["seed"]))
    trace = []
    img = pipe.sample(int(g["n"]), tuple(int(v) for v in g["size"]), noise=noise, trace=trace, **kw)
    assert noise.draw_index == int(g["draws"])  # Q3: same number of draws as the reference
    tol = GUIDED_TINY_TOL if kw.get("guidance_scale", 1.0) not in (1.0,) else TOL
    assert relerr(trace[0][0], T(g["x0_step0"])) < TOL
    e_x0, e_img = relerr(trace[-1][0], T(g["x0_final"])), relerr(img, T(g["image"]))
    print(f"[measured] sample_tiny_golden {name}: x0_final {e_x0:.1e} image {e_img:.1e} (tolerance {tol:.0e})")
    assert e_x0 < tol
    assert e_img < tol


def test_eta_raises_like_reference(dev):
    pipe = build_product_pipe(R.tiny_unet_kwargs(3, "none"), None, "pipe_tiny", dev)
    with pytest.raises(TypeError):
        pipe.sample(1, (8, 8, 8), steps=2, eta=0.0)


def _trained_like(model, tag):
    """seeded weights, then every GroupNorm scale log-uniform in [0.01, 30] and every shift uniform in [-5, 5] -- the ranges a trained checkpoint can
    hold and the synthetic init (gamma ~ 1, beta ~ 0) never visits; zero-init second convolutions are already non-zero under synth_state_dict"""
    S.synth_state_dict(model, tag)
    with torch.no_grad():
        for key, p in model.named_parameters():
            if key.endswith("norm.weight") or key.endswith("norm_x.weight"):
                u = S.synth_input("tl_gamma:" + key, tuple(p.shape)).clamp(-1.7320508, 1.7320508) / 1.7320508     # uniform in [-1, 1]
                p.copy_((torch.exp((u + 1) * 0.5 * np.log(30 / 0.01)) * 0.01).to(p.device))
            elif key.endswith("norm.bias") or key.endswith("norm_x.bias"):
                u = S.synth_input("tl_beta:" + key, tuple(p.shape)).clamp(-1.7320508, 1.7320508) / 1.7320508
                p.copy_((5.0 * u).to(p.device))
    return model


def test_trained_like_weights_and_bound_slack(dev, conv_precision):
    """GroupNorm scales up to 30 and shifts up to +-5 (what a trained checkpoint may hold) -- one evaluation of the published
    UNet (B = 4: the 8 x 8 and 16 x 16 levels take the Winograd form where the library can) and one VAE decode against the oracle, on every
    arithmetic; on the fp16-pair arithmetic every derived operand bound of the two passes is audited against the data it scales
    (kernels.AUDIT): bound / true max < 2^20 at every site (the pair format keeps 23 bits down to 2^-28 of the bound)."""
    from medfusion_amd import blocks as BLK
    ora_u = _trained_like(R.UNet(**R.published_unet_kwargs(2)).eval(), "trained_like.unet.")
    ora_v = _trained_like(R.VAE(**R.published_vae_kwargs(8)).eval(), "trained_like.vae.")
    unet = _trained_like(M.UNet(**to_product_kwargs(R.published_unet_kwargs(2))), "trained_like.unet.").to(dev).eval()
    vae = _trained_like(M.VAE(**R.published_vae_kwargs(8)), "trained_like.vae.").to(dev).eval()
    x = S.synth_input("tl_x", (4, 8, 32, 32))
    t = torch.tensor([999, 500, 17, 0])
    c = torch.tensor([0, 1, 1, 0])
    z = S.synth_input("tl_z", (1, 8, 32, 32))
    with torch.no_grad():
        want_u = ora_u(x, t, c)[0]
        want_v = ora_v.decode(z)
    K.AUDIT_LOG.clear()
    K.AUDIT = conv_precision == 5
    try:
        got_u = unet(x.to(dev), t.to(dev), c.to(dev))[0]
        got_v = vae.decode(z.to(dev))
    finally:
        K.AUDIT = False
    e_u, e_v = relerr_rows(got_u, want_u), relerr(got_v, want_v)
    print(f"[measured] trained-like weights (gamma in [0.01, 30], beta in [-5, 5]), arithmetic {conv_precision}, WINOGRAD {BLK.WINOGRAD}: "
          f"UNet per-sample relerr {e_u:.2e}, VAE decode {e_v:.2e}")
    assert e_u < TOL and e_v < TOL, (e_u, e_v)
    if conv_precision == 5:
        assert K.AUDIT_LOG, "the audit saw no fp16-pair producer"
        worst = {}
        for site, shape, smax, smin, finite in K.AUDIT_LOG:
            assert finite, (site, shape)
            assert smin >= 1.0 - 1e-6, (site, shape, smin)          # a bound below the data would saturate the pairs
            w = worst.get(site)
            if w is None or smax > w[0]:
                worst[site] = (smax, shape)
        for site, (smax, shape) in sorted(worst.items(), key=lambda kv: -kv[1][0]):
            print(f"[measured] bound slack, worst over {sum(1 for e in K.AUDIT_LOG if e[0] == site)} tensors: 2^{np.log2(smax):.1f} at {shape} -- {site}")
        assert max(v[0] for v in worst.values()) < 2.0 ** 20, worst


def test_block_outputs_as_pairs_only(dev, conv_precision):
    """round 5: where every reader of a conv block's output takes fp16 pairs (UNet._pairs_only_outputs: the published architecture on the
    default arithmetic), the blocks do not write the fp32 form.  The convolutions read the same operands either way; a residual add that found
    the fp32 form used its 24th significand bit, which the pair form does not carry (split_f16.h: at most the last bit cleared) -- so the
    network's output moves by rounding noise, an order of magnitude below its distance to the oracle; on the other arithmetics and on models
    that do not qualify the switch is inert (same bits)"""
    from medfusion_amd import unet as UN
    net = M.UNet(**to_product_kwargs(R.published_unet_kwargs(2)))
    S.synth_state_dict(net, "pairs_only_outputs.unet.")
    net.to(dev).eval()
    x = S.synth_input("po_x", (4, 8, 32, 32)).to(dev)
    t, c = torch.tensor([999, 500, 17, 0], device=dev), torch.tensor([0, 1, 1, 0], device=dev)
    old = UN.PAIRS_ONLY_BLOCK_OUTPUTS
    try:
        UN.PAIRS_ONLY_BLOCK_OUTPUTS = False
        assert not net._pairs_only_outputs((4, 32, 32, 256))
        want = net(x, t, c)[0].clone()
        UN.PAIRS_ONLY_BLOCK_OUTPUTS = True
        assert net._pairs_only_outputs((4, 32, 32, 256)) == (conv_precision == 5)
        got = net(x, t, c)[0]
    finally:
        UN.PAIRS_ONLY_BLOCK_OUTPUTS = old
    e = relerr_rows(got, want)
    print(f"[measured] pairs-only block outputs vs fp32 + pairs, arithmetic {conv_precision}: per-sample relerr {e:.2e}")
    if conv_precision == 5:
        assert e < 5e-7, e
    else:
        assert torch.equal(got, want)


def test_progress_callback_in_every_loop_form(dev):
    """denoise(progress_cb=...): the hook where the reference drives st.progress / tqdm (diffusion_pipeline.py:289-291) -- monotone, ends at
    (total, total), same images with and without it, in the Python loop, the command-list replay and the graph replay"""
    pipe = build_product_pipe(R.tiny_unet_kwargs(3, "none"), R.tiny_vae_kwargs(), "pipe_tiny", dev)
    for loop in ("eager", "cmdlist", "graph"):
        seen = []
        a = pipe.sample(2, (8, 8, 8), steps=25, noise=M.PhiloxDeviceNoise(5), loop=loop, progress_cb=lambda done, total: seen.append((done, total)))
        b = pipe.sample(2, (8, 8, 8), steps=25, noise=M.PhiloxDeviceNoise(5), loop=loop)
        assert torch.equal(a, b), loop
        assert seen and seen[-1] == (25, 25) and all(t == 25 for _, t in seen), (loop, seen)
        assert [d for d, _ in seen] == sorted(set(d for d, _ in seen)), (loop, seen)
        assert len(seen) == 25 if loop == "eager" else len(seen) >= 20, (loop, len(seen))
        assert seen[0] == (1, 25), (loop, seen[:3])      # the eager first iteration of the replayed loops is reported too
    # a loop that is re-run after a fused rendezvous fault (K.with_fused_fallback: rewind + second pass) never reports a count twice
    from medfusion_amd import kernels as K
    orig = K.with_fused_fallback

    def twice(device, fn, rewind=None):
        fn()
        if rewind is not None:
            rewind()
        return fn()
    K.with_fused_fallback = twice
    try:
        seen = []
        c = pipe.sample(2, (8, 8, 8), steps=25, noise=M.PhiloxDeviceNoise(5), loop="eager", progress_cb=lambda done, total: seen.append(done))
    finally:
        K.with_fused_fallback = orig
    assert seen == list(range(1, 26)), seen
    assert torch.equal(c, b)


@pytest.fixture(scope="module")
def published(dev):
    """Published architecture (194 M-parameter UNet + VAE), synthetic weights; oracle on CPU, product on GPU."""
    torch.set_num_threads(max(1, torch.get_num_threads()))
    ora = build_oracle_pipe(R.published_unet_kwargs(2), R.published_vae_kwargs(8), "published")
    pipe = build_product_pipe(R.published_unet_kwargs(2), R.published_vae_kwargs(8), "published", dev)
    return ora, pipe


def test_cfg1_published_golden(dev, published):
    """BASELINE.json configs[0] against the reference's own output: 64x64, 50 DDIM steps."""
    _, pipe = published
    g = gold("cfg1_published_64px")
    y, _ = pipe.noise_estimator(T(g["unet_x"]).to(dev), T(g["unet_t"]).to(dev), T(g["unet_c"]).to(dev))
    assert relerr(y, T(g["unet_y"])) < TOL
    trace = []
    img = pipe.sample(2, (8, 8, 8), steps=int(g["steps"]), use_ddim=True, noise=oracle_noise(int(g["seed"])), trace=trace)
    assert relerr(trace[0][0], T(g["x0_step0"])) < TOL
    assert relerr(trace[-1][0], T(g["x0_final"])) < TOL
    assert relerr(img, T(g["image"])) < TOL


@torch.no_grad()
def test_fp32_mfma_with_the_winograd_form_opt_in(dev, published, conv_precision):
    """blocks.WINOGRAD_F32_MFMA = 1 (opt-in; bench.py times it next to the bit-for-bit direct chain): the plain fp32 MFMA kernel runs the component GEMMs of the
    8 x 8 / 16 x 16 levels -- no operand splitting anywhere.  Published UNet, B = 4 (n T = 64 tile rows per component at the 8 x 8 level), against the oracle and
    against the direct chain of the same arithmetic."""
    if conv_precision != 0:
        pytest.skip("runs once")
    from medfusion_amd import blocks as BLK
    ora, pipe = published
    x = S.synth_input("pub256_x4", (4, 8, 32, 32))
    t = torch.tensor([731, 12, 400, 999])
    c = torch.tensor([1, 0, 1, 0])
    want, _ = ora.noise_estimator(x, t, c)
    direct, _ = pipe.noise_estimator(x.to(dev), t.to(dev), c.to(dev))
    BLK.WINOGRAD_F32_MFMA = 1
    try:
        got, _ = pipe.noise_estimator(x.to(dev), t.to(dev), c.to(dev))
        got2, _ = pipe.noise_estimator(x.to(dev), t.to(dev), c.to(dev))      # second evaluation: the producers write V themselves now
    finally:
        BLK.WINOGRAD_F32_MFMA = 0
    e, ed = relerr_rows(got, want), relerr_rows(direct, want)
    print(f"[measured] fp32 MFMA with the Winograd form (opt-in): UNet per-sample relerr vs the oracle {e:.2e} (direct chain: {ed:.2e}); vs the direct chain {relerr_rows(got, direct):.2e}")
    assert e < TOL and not torch.equal(got, direct)          # (it really took the other form)
    assert torch.equal(got, got2)


@torch.no_grad()
def test_published_unet_and_decode_vs_oracle_256px(dev, published):
    """Full published shapes: latent (8,32,32) -> 256x256 image.  Oracle evaluated on CPU on the same inputs."""
    ora, pipe = published
    x = S.synth_input("pub256_x", (2, 8, 32, 32))
    t = torch.tensor([731, 731])
    c = torch.tensor([1, 0])
    want, _ = ora.noise_estimator(x, t, c)
    got, _ = pipe.noise_estimator(x.to(dev), t.to(dev), c.to(dev))
    assert relerr(got, want) < TOL
    wantu, _ = ora.noise_estimator(x, t, None)
    gotu, _ = pipe.noise_estimator(x.to(dev), t.to(dev), None)
    assert relerr(gotu, wantu) < TOL
    z = S.synth_input("pub256_z", (1, 8, 32, 32))
    assert relerr(pipe.latent_embedder.decode(z.to(dev)), ora.latent_embedder.decode(z)) < TOL
    img = S.synth_input("pub256_img", (1, 3, 256, 256), 0.5)
    nz = S.PhiloxNoise(21)
    ora.latent_embedder.quantizer.noise_fn = lambda shape, device: nz(torch.empty(shape))
    assert relerr(pipe.latent_embedder.encode(img.to(dev), noise=oracle_noise(21)), ora.latent_embedder.encode(img)) < TOL


@torch.no_grad()
def test_published_short_trajectory_vs_oracle_256px(dev, published):
    """3 DDIM iterations of the real loop at (8,32,32), B=2, CFG on, decoded to 256x256 -- oracle on CPU."""
    ora, pipe = published
    cond = torch.tensor([1, 0])
    ora.set_noise_fn(S.PhiloxNoise(33))
    tr_o = []
    want = ora.sample(2, (8, 32, 32), condition=cond, guidance_scale=2.0, steps=3, use_ddim=True, trace=tr_o)
    tr_p = []
    got = pipe.sample(2, (8, 32, 32), condition=cond.to(dev), guidance_scale=2.0, steps=3, use_ddim=True, noise=oracle_noise(33), trace=tr_p)
    for (a0, a1), (b0, b1) in zip(tr_p, tr_o):
        assert relerr(a0, b0) < TOL and relerr(a1, b1) < TOL
    assert relerr(got, want) < TOL


@torch.no_grad()
def test_full_size_properties_cfg2(dev, published):
    """BASELINE.json configs[1] shape (B=16, latent 8x32x32) via size-independent properties, device Philox noise:
    rows are independent (row i of the B=16 run == the same sample generated alone or in another shard), and the
    run is deterministic.  Few steps keep the test short; the per-step arithmetic is what the 150-step run repeats."""
    _, pipe = published
    kw = dict(steps=4, use_ddim=True)
    full = pipe.sample(16, (8, 32, 32), noise=M.PhiloxDeviceNoise(1234), decode=False, **kw)
    again = pipe.sample(16, (8, 32, 32), noise=M.PhiloxDeviceNoise(1234), decode=False, **kw)
    assert torch.equal(full, again)
    shard = pipe.sample(16, (8, 32, 32), noise=M.PhiloxDeviceNoise(1234), shard=(3, 4), decode=False, **kw)  # rows 12..15
    assert relerr(shard, full[12:16]) < 1e-5   # different batch => different conv tiling/split-K order, not bit-equal
    one = pipe.sample(16, (8, 32, 32), noise=M.PhiloxDeviceNoise(1234), shard=(5, 16), decode=False, **kw)   # row 5 alone
    assert relerr(one, full[5:6]) < 1e-5
    assert full.isfinite().all()


@torch.no_grad()
def test_conditional_3class_cfg3_shape(dev):
    """configs[2]: LabelEmbedder with 3 classes, guidance 8 vs guidance 1 code paths on the published widths (B=3)."""
    pipe = build_product_pipe(R.published_unet_kwargs(3), None, "published3", dev)
    ora = build_oracle_pipe(R.published_unet_kwargs(3), None, "published3")
    cond = torch.arange(3) % 3
    for g in (1.0, 8.0):
        ora.set_noise_fn(S.PhiloxNoise(44))
        want = ora.sample(3, (8, 16, 16), condition=cond, guidance_scale=g, steps=2, use_ddim=True)
        got = pipe.sample(3, (8, 16, 16), condition=cond.to(dev), guidance_scale=g, steps=2, use_ddim=True, noise=oracle_noise(44))
        e = relerr(got, want)
        print(f"[measured] cfg3 shape, guidance {g}: {e:.1e}")
        assert e < (GUIDED_TINY_TOL if g != 1.0 else TOL)


@torch.no_grad()
def test_batched_cfg_pair_equals_two_passes(dev):
    """Classifier-free guidance as ONE 2B-row UNet call (default) vs the reference's two sequential calls (un-guided first)."""
    pipe = build_product_pipe(R.tiny_unet_kwargs(3, ["none", "none", "linear", "spatial"]), R.tiny_vae_kwargs(), "pipe_cfgpair", dev)
    cond = torch.tensor([2, 0, 1], device=dev)
    for un_cond in (None, torch.tensor([0, 1, 2], device=dev)):
        outs = []
        for flag in (True, False):
            pipe.batch_cfg = flag
            outs.append(pipe.sample(3, (8, 8, 8), condition=cond, un_cond=un_cond, guidance_scale=3.0, steps=3, use_ddim=True, noise=M.PhiloxDeviceNoise(5)))
        assert relerr(outs[0], outs[1]) < 1e-5


@pytest.mark.parametrize("use_ddim,cond", [(True, False), (False, False), (True, True)])
@torch.no_grad()
def test_hipgraph_captured_step_equals_eager(dev, use_ddim, cond):
    """configs[3]: the denoise iteration captured once as a hipGraph and replayed (device step counter drives t, the scheduler
    table and the Philox draw index) must give bit-identical latents to the eager loop."""
    pipe = build_product_pipe(R.tiny_unet_kwargs(3, "none"), R.tiny_vae_kwargs(), "pipe_graph", dev)
    kw = dict(steps=9, use_ddim=use_ddim)
    if cond:
        kw.update(condition=torch.tensor([2, 0, 1, 1], device=dev), guidance_scale=4.0)
    eager = pipe.sample(4, (8, 8, 8), noise=M.PhiloxDeviceNoise(77), loop="eager", **kw)
    graph = pipe.sample(4, (8, 8, 8), noise=M.PhiloxDeviceNoise(77), use_graph=True, **kw)
    assert torch.equal(eager, graph)
    pipe.hoist_embeddings = False     # the embedding path evaluated INSIDE the loop (no table, no per-iteration gather): the same bits
    try:
        assert torch.equal(eager, pipe.sample(4, (8, 8, 8), noise=M.PhiloxDeviceNoise(77), loop="eager", **kw))
    finally:
        pipe.hoist_embeddings = True
    # the native command list of the loop body (iteration 1 recorded by the library, the rest re-issued from C): the same bits, and it is
    # what sample() does by default
    pipe.last_cmdlist_launches = 0
    listed = pipe.sample(4, (8, 8, 8), noise=M.PhiloxDeviceNoise(77), loop="cmdlist", **kw)
    assert torch.equal(eager, listed) and pipe.last_cmdlist_launches > 20
    pipe.last_cmdlist_launches = 0
    assert torch.equal(eager, pipe.sample(4, (8, 8, 8), noise=M.PhiloxDeviceNoise(77), **kw)) and pipe.last_cmdlist_launches > 20
    with pytest.raises(ValueError, match="cmdlist"):
        pipe.sample(2, (8, 8, 8), noise=oracle_noise(1), loop="cmdlist", steps=5)
    again = pipe.sample(4, (8, 8, 8), noise=M.PhiloxDeviceNoise(77), use_graph=True, **kw)
    assert torch.equal(graph, again)
    with pytest.raises(RuntimeError, match="Philox"):
        pipe.sample(2, (8, 8, 8), noise=oracle_noise(1), use_graph=True, steps=3)


@torch.no_grad()
def test_ddpm_1000_schedule_cfg4_prefix(dev, published):
    """configs[3]: non-DDIM posterior sampling; `steps` < T takes the FIRST timesteps (Q5).  6 iterations vs oracle."""
    ora, pipe = published
    ora.set_noise_fn(S.PhiloxNoise(55))
    want = ora.sample(1, (8, 16, 16), steps=6, use_ddim=False)
    got = pipe.sample(1, (8, 16, 16), steps=6, use_ddim=False, noise=oracle_noise(55))
    assert relerr(got, want) < TOL


@torch.no_grad()
def test_interpolate_vs_oracle(dev):
    """DiffusionPipeline.interpolate (the reference method itself raises: it forwards clip_x0= to
    estimate_x_t; the oracle restates its evident intent)."""
    ora = build_oracle_pipe(R.tiny_unet_kwargs(2, "none"), R.tiny_vae_kwargs(), "pipe_interp")
    pipe = build_product_pipe(R.tiny_unet_kwargs(2, "none"), R.tiny_vae_kwargs(), "pipe_interp", dev)
    a, b = S.synth_input("interp_a", (2, 8, 8, 8)), S.synth_input("interp_b", (2, 8, 8, 8))
    cond = torch.tensor([1, 0])
    ora.set_noise_fn(S.PhiloxNoise(61))
    want = ora.interpolate(a, b, i=6, condition=cond, lam=0.3, guidance_scale=2.0)
    got = pipe.interpolate(a.to(dev), b.to(dev), i=6, condition=cond.to(dev), lam=0.3, guidance_scale=2.0, noise=oracle_noise(61))
    assert relerr(got, want) < TOL
    with pytest.raises(TypeError):
        pipe.interpolate(a.to(dev), b.to(dev))


@torch.no_grad()
def test_forward_single_step_api_incl_cold_diffusion(dev):
    """DiffusionPipeline.forward (diffusion_pipeline.py:232-275) as a stand-alone call, regular and cold-diffusion branches."""
    ora = build_oracle_pipe(R.tiny_unet_kwargs(3, "none"), None, "pipe_fwd", clip_x0=True)
    pipe = build_product_pipe(R.tiny_unet_kwargs(3, "none"), None, "pipe_fwd", dev, clip_x0=True)
    x, nz = S.synth_input("fwd_x", (3, 8, 8, 8)), S.synth_input("fwd_n", (3, 8, 8, 8))
    t, cond = torch.full((3,), 400), torch.tensor([0, 2, 1])
    for cold in (False, True):
        ora.set_noise_fn(lambda like: nz)
        want = ora(x, t, cond, guidance_scale=2.5, cold_diffusion=cold)
        got = pipe(x.to(dev), t.to(dev), cond.to(dev), guidance_scale=2.5, cold_diffusion=cold, noise=nz.to(dev))
        for w, g in zip(want[:3], got[:3]):
            assert relerr(g, w) < TOL, cold


@pytest.mark.parametrize("shape", [(1, 8, 8, 8), (5, 8, 12, 20), (33, 8, 8, 8), (2, 8, 4, 4)])
@torch.no_grad()
def test_ragged_and_nonsquare_shapes_vs_oracle(dev, shape):
    """Edge shapes: batch 1, a batch that leaves ragged GEMM tiles (33), non-square latents, the smallest legal latent (4x4)."""
    ora = build_oracle_pipe(R.tiny_unet_kwargs(2, "none"), R.tiny_vae_kwargs(), "pipe_ragged")
    pipe = build_product_pipe(R.tiny_unet_kwargs(2, "none"), R.tiny_vae_kwargs(), "pipe_ragged", dev)
    b = shape[0]
    x = S.synth_input(f"rag{shape}", shape)
    t = torch.full((b,), 123)
    c = torch.arange(b) % 2
    want, _ = ora.noise_estimator(x, t, c)
    got, _ = pipe.noise_estimator(x.to(dev), t.to(dev), c.to(dev))
    assert relerr(got, want) < TOL
    assert relerr(pipe.latent_embedder.decode(x.to(dev)), ora.latent_embedder.decode(x)) < TOL
    ora.set_noise_fn(S.PhiloxNoise(71))
    want = ora.sample(b, shape[1:], steps=2, use_ddim=True)
    got = pipe.sample(b, shape[1:], steps=2, use_ddim=True, noise=oracle_noise(71))
    assert relerr(got, want) < TOL


@torch.no_grad()
def test_four_channel_latent_model_family(dev):
    """The eye / colon Medfusion models use in_ch = out_ch = emb_channels = 4 (streamlit/pages/eye.py:34, colon.py:36)."""
    ukw = R.tiny_unet_kwargs(2, "none")
    ukw.update(in_ch=4, out_ch=4)
    vkw = R.tiny_vae_kwargs(emb_channels=4)
    ora = build_oracle_pipe(ukw, vkw, "pipe_4ch")
    pipe = build_product_pipe(ukw, vkw, "pipe_4ch", dev)
    ora.set_noise_fn(S.PhiloxNoise(81))
    cond = torch.tensor([1, 0])
    want = ora.sample(2, (4, 16, 16), condition=cond, guidance_scale=1.0, steps=3, use_ddim=True)
    got = pipe.sample(2, (4, 16, 16), condition=cond.to(dev), guidance_scale=1.0, steps=3, use_ddim=True, noise=oracle_noise(81))
    assert relerr(got, want) < TOL
    img = S.synth_input("img4", (2, 3, 32, 32), 0.5)
    nz = S.PhiloxNoise(82)
    ora.latent_embedder.quantizer.noise_fn = lambda shape, device: nz(torch.empty(shape))
    assert relerr(pipe.latent_embedder.encode(img.to(dev), noise=oracle_noise(82)), ora.latent_embedder.encode(img)) < TOL


@torch.no_grad()
def test_cfg5_512px_shape_properties(dev, published):
    """configs[4]: latent (8,64,64) -> 512x512 on the published widths.  One UNet evaluation + decode vs the oracle at B=1
    (the oracle needs ~10 s on CPU), then determinism, row independence and a decode at the per-GPU workload (32 images over 4 GPUs = 8 rows)."""
    ora, pipe = published
    x = S.synth_input("cfg5_x", (1, 8, 64, 64))
    t = torch.tensor([250])
    want, _ = ora.noise_estimator(x, t, None)
    got, _ = pipe.noise_estimator(x.to(dev), t.to(dev), None)
    assert relerr(got, want) < TOL
    img = pipe.latent_embedder.decode(x.to(dev))
    assert img.shape == (1, 3, 512, 512)
    assert relerr(img, ora.latent_embedder.decode(x)) < TOL
    full = pipe.sample(8, (8, 64, 64), steps=2, use_ddim=True, noise=M.PhiloxDeviceNoise(9), decode=False)
    assert torch.equal(full, pipe.sample(8, (8, 64, 64), steps=2, use_ddim=True, noise=M.PhiloxDeviceNoise(9), decode=False))
    one = pipe.sample(8, (8, 64, 64), steps=2, use_ddim=True, noise=M.PhiloxDeviceNoise(9), decode=False, shard=(2, 8))
    assert relerr(one, full[2:3]) < 1e-5
    imgs = pipe.latent_embedder.decode(full)
    assert imgs.shape == (8, 3, 512, 512) and bool(imgs.isfinite().all())
    assert relerr(imgs[5:6], pipe.latent_embedder.decode(full[5:6])) < 1e-5


@torch.no_grad()
def test_full_length_cfg2_trajectory_vs_oracle(dev, published, conv_precision):
    """The whole of BASELINE.json configs[1] for ONE sample: 150 DDIM iterations (eta = 1, 300 noise draws) at latent (8,32,32) on
    the published architecture, decoded to 256x256 -- HIP path vs the oracle on the GPU box's CPU, identical injected noise.
    Checks the x_0 estimate every 10 iterations (error growth through the recurrence) and the final image."""
    ora, pipe = published
    if "full" not in _ORACLE_CACHE:  # 150 UNet evaluations on the CPU: once for both conv arithmetics
        torch.set_num_threads(min(32, torch.get_num_threads()))
        ora.set_noise_fn(S.PhiloxNoise(2024))
        tr = []
        _ORACLE_CACHE["full"] = (ora.sample(1, (8, 32, 32), steps=150, use_ddim=True, trace=tr), tr)
    want, tr_o = _ORACLE_CACHE["full"]
    tr_p = []
    got = pipe.sample(1, (8, 32, 32), steps=150, use_ddim=True, noise=oracle_noise(2024), trace=tr_p)
    errs = [relerr(tr_p[i][0], tr_o[i][0]) for i in range(0, 150, 10)] + [relerr(tr_p[-1][0], tr_o[-1][0])]
    rms = [relerr_rms(tr_p[i][0], tr_o[i][0]) for i in range(0, 150, 10)] + [relerr_rms(tr_p[-1][0], tr_o[-1][0])]
    print(f"conv precision {conv_precision}: x0 rel-err every 10 iterations:", " ".join(f"{e:.1e}" for e in errs), "| image:", f"{relerr(got, want):.1e}",
          "| RMS-relative:", " ".join(f"{e:.1e}" for e in rms), "| image:", f"{relerr_rms(got, want):.1e}")
    assert max(errs) < TOL and max(rms) < TOL
    assert relerr(got, want) < TOL and relerr_rms(got, want) < TOL


@torch.no_grad()
def test_full_length_trajectory_in_the_unit_range_vs_oracle(dev, conv_precision):
    """150 DDIM iterations in the regime real checkpoints live in, |x_t| = O(1): with the synthetic weights the
    x_T-objective trajectories above reach |x| ~ 1e4 .. 1e7 (the network is no denoiser, so x_0 = (x_t - sqrt(1-abar) eps) / sqrt(abar) explodes
    at the first timesteps), and the per-sample power-of-two scales of the fp16-pair operands carry them.  Here the SAME published
    architecture runs with estimator_objective = 'x_0' (the network output IS the x_0 estimate, O(1) by the fan-in scaling of its last
    convolution; diffusion_pipeline.py:263-266): every tensor of the loop stays within ~10, the scales sit near 2^-11, and the
    comparison is against the oracle with identical injected noise.  Asserts the max norm, the per-sample max norm and the RMS-relative
    error at the path's tolerance, and that the trajectory really stayed in the unit range."""
    pipe = build_product_pipe(R.published_unet_kwargs(None), R.published_vae_kwargs(8), "published", dev, objective="x_0")
    if "unit" not in _ORACLE_CACHE:
        ora = build_oracle_pipe(R.published_unet_kwargs(None), R.published_vae_kwargs(8), "published", objective="x_0")
        torch.set_num_threads(min(32, torch.get_num_threads()))
        ora.set_noise_fn(S.PhiloxNoise(909))
        tr = []
        _ORACLE_CACHE["unit"] = (ora.sample(1, (8, 32, 32), steps=150, use_ddim=True, trace=tr), tr)
    want, tr_o = _ORACLE_CACHE["unit"]
    tr_p = []
    got = pipe.sample(1, (8, 32, 32), steps=150, use_ddim=True, noise=oracle_noise(909), trace=tr_p)
    idx = list(range(0, 150, 10)) + [149]
    peak = max(float(t.abs().max()) for e in tr_o for t in e if torch.is_tensor(t))
    errs = [max(relerr(tr_p[i][k], tr_o[i][k]) for k in (0, 1)) for i in idx]
    rms = [max(relerr_rms(tr_p[i][k], tr_o[i][k]) for k in (0, 1)) for i in idx]
    print(f"unit-range trajectory (objective x_0), conv precision {conv_precision}: max |x| over the loop {peak:.3g}; x0 / x_t rel-err at iterations "
          f"{idx[0]}..{idx[-1]}:", " ".join(f"{e:.1e}" for e in errs), "| RMS-relative:", " ".join(f"{e:.1e}" for e in rms),
          f"| image: {relerr(got, want):.1e} (RMS {relerr_rms(got, want):.1e})")
    assert peak < 100.0, "the trajectory left the unit range: this test no longer covers the O(1) regime"
    assert max(errs) < TOL and max(rms) < TOL
    assert relerr(got, want) < TOL and relerr_rms(got, want) < TOL


BF16_TOL = 5e-2


@torch.no_grad()
def test_bf16_opt_in_mode_has_its_own_tolerance(dev, published, conv_precision):
    """MF_CONV_BF16 (opt-in): one UNet evaluation and a 20-iteration trajectory at the published size against the
    fp32 oracle, with the mode's own tolerance (operands carry 8 significant bits: ~3e-3 per convolution)."""
    if conv_precision != 1:
        pytest.skip("runs once")
    from medfusion_amd import blocks as BLK
    ora, pipe = published
    BLK.CONV_PRECISION = 4
    try:
        x = S.synth_input("pub256_x", (2, 8, 32, 32))
        t = torch.tensor([731, 731])
        want, _ = ora.noise_estimator(x, t, None)
        got, _ = pipe.noise_estimator(x.to(dev), t.to(dev), None)
        e_unet = relerr(got, want)
        ora.set_noise_fn(S.PhiloxNoise(77))
        tr_o, tr_p = [], []
        want_img = ora.sample(1, (8, 32, 32), steps=20, use_ddim=True, trace=tr_o)
        got_img = pipe.sample(1, (8, 32, 32), steps=20, use_ddim=True, noise=oracle_noise(77), trace=tr_p)
        errs = [relerr(a[0], b[0]) for a, b in zip(tr_p, tr_o)]
        print(f"bf16 mode: UNet {e_unet:.1e} | x0 along 20 iterations: " + " ".join(f"{e:.1e}" for e in errs[::3]) + f" | image {relerr(got_img, want_img):.1e}")
        assert e_unet < BF16_TOL and max(errs) < BF16_TOL and relerr(got_img, want_img) < BF16_TOL
    finally:
        BLK.CONV_PRECISION = conv_precision


@torch.no_grad()
def test_full_length_guided_trajectory_vs_oracle(dev, published, conv_precision):
    """What scripts/sample.py really runs (/root/reference/scripts/sample.py:45): classifier-free guidance 8, un_cond=None, 150 DDIM
    iterations -- ONE sample of the 2-class published architecture at latent (8,32,32), decoded to 256x256, against the oracle on the GPU
    box's CPU (300 UNet evaluations there, once for the three arithmetics), identical injected noise.  x_0 is compared every 10 iterations.
    Tolerance: stated from what was measured on MI355X (profiles/r03_guided_trajectory.txt), not a blanket 1e-3: on the published
    architecture the x_0 error does NOT grow along the guided trajectory -- 3.5e-6 at iteration 0, 5e-6 .. 8e-6 at every tenth iteration
    up to 149, image 5.6e-6 on the default arithmetic (7.2e-6 bf16 triplets, 9.6e-6 fp32 MFMA; the oracle's two evaluations per
    iteration differ from ours in summation order only) -- so the bound is 5e-5 throughout, ~5x the largest measured value and half the
    path's stated 1e-4."""
    ora, pipe = published
    cond = torch.tensor([1])
    if "guided" not in _ORACLE_CACHE:
        torch.set_num_threads(min(32, torch.get_num_threads()))
        ora.set_noise_fn(S.PhiloxNoise(4242))
        tr = []
        _ORACLE_CACHE["guided"] = (ora.sample(1, (8, 32, 32), condition=cond, guidance_scale=8.0, un_cond=None, steps=150, use_ddim=True, trace=tr), tr)
    want, tr_o = _ORACLE_CACHE["guided"]
    tr_p = []
    src = oracle_noise(4242)
    got = pipe.sample(1, (8, 32, 32), condition=cond.to(dev), guidance_scale=8.0, un_cond=None, steps=150, use_ddim=True, noise=src, trace=tr_p)
    assert src.draw_index == 300
    idx = list(range(0, 150, 10)) + [149]
    errs = [relerr(tr_p[i][0], tr_o[i][0]) for i in idx]
    rms = [relerr_rms(tr_p[i][0], tr_o[i][0]) for i in idx]
    e_img, r_img = relerr(got, want), relerr_rms(got, want)
    print(f"guided (g=8) trajectory, conv precision {conv_precision}: x0 rel-err at iterations {idx[0]}..{idx[-1]}:", " ".join(f"{e:.1e}" for e in errs), "| image:", f"{e_img:.1e}",
          "| RMS-relative:", " ".join(f"{e:.1e}" for e in rms), "| image:", f"{r_img:.1e}")
    for i, e, r in zip(idx, errs, rms):
        assert e < GUIDED_TOL(i) and r < GUIDED_TOL(i), (i, e, r)
    assert e_img < GUIDED_TOL(149) and r_img < GUIDED_TOL(149)


def GUIDED_TOL(iteration: int) -> float:
    """bound of the x_0 error of the guided trajectory at `iteration` (see the test above and profiles/r03_guided_trajectory.txt)"""
    return GUIDED_TOL_START + (GUIDED_TOL_END - GUIDED_TOL_START) * iteration / 149.0


GUIDED_TOL_START, GUIDED_TOL_END = 5e-5, 5e-5     # flat: no growth was measured (see the docstring)


@torch.no_grad()
def test_cfg2_rows_of_the_benchmarked_batch_vs_oracle(dev, conv_precision):
    """BASELINE configs[1] at the batch that is benchmarked: B = 16 unconditional samples at latent (8,32,32), device Philox noise, 24 DDIM
    iterations (the per-iteration arithmetic is what the 150-iteration run repeats, the tiles and split-K factors are those of B = 16, not
    of the B <= 4 the other oracle comparisons use).  Rows 0, 7 and 15 against the oracle fed the SAME global Philox rows -- x_0 along the
    trajectory, the final latents and the decoded images."""
    from medfusion_amd import published as P
    pipe = P.build_published_pipeline(dev, num_classes=None)
    ora = build_oracle_pipe(R.published_unet_kwargs(None), R.published_vae_kwargs(8), "published")
    rows = [0, 7, 15]
    key = "cfg2rows"
    if key not in _ORACLE_CACHE:
        torch.set_num_threads(min(32, torch.get_num_threads()))
        nz = S.PhiloxNoise(515)
        ora.set_noise_fn(lambda like: nz(torch.empty((16, *like.shape[1:])))[rows])     # rows 0, 7, 15 of the 16-row draws
        tr = []
        _ORACLE_CACHE[key] = (ora.sample(3, (8, 32, 32), steps=24, use_ddim=True, trace=tr), tr)
    want, tr_o = _ORACLE_CACHE[key]
    tr_p = []
    got = pipe.sample(16, (8, 32, 32), steps=24, use_ddim=True, noise=M.PhiloxDeviceNoise(515), trace=tr_p)
    errs = [relerr(tr_p[i][0][rows], tr_o[i][0]) for i in range(0, 24, 4)] + [relerr(tr_p[-1][1][rows], tr_o[-1][1])]
    # per SAMPLE (each row against its own max / its own RMS: the rows differ in scale, a joint max norm would hide the small one)
    per_row = [relerr_rows(tr_p[i][0][rows], tr_o[i][0]) for i in range(0, 24, 4)] + [relerr_rows(tr_p[-1][1][rows], tr_o[-1][1])]
    rms = [relerr_rms(tr_p[i][0][rows], tr_o[i][0]) for i in range(0, 24, 4)] + [relerr_rms(tr_p[-1][1][rows], tr_o[-1][1])]
    scales = [float(tr_o[-1][1][k].abs().max()) for k in range(len(rows))]
    print(f"cfg2 batch rows {rows}, conv precision {conv_precision}: x0 rel-err every 4 iterations + final latents:", " ".join(f"{e:.1e}" for e in errs),
          "| images:", f"{relerr(got[rows], want):.1e}", "| per-sample max-norm:", " ".join(f"{e:.1e}" for e in per_row), f"images {relerr_rows(got[rows], want):.1e}",
          "| per-sample RMS-relative:", " ".join(f"{e:.1e}" for e in rms), f"images {relerr_rms(got[rows], want):.1e}", "| max|x| of the three rows:", " ".join(f"{v:.2g}" for v in scales))
    assert max(errs) < TOL and relerr(got[rows], want) < TOL
    assert max(per_row) < TOL and relerr_rows(got[rows], want) < TOL
    assert max(rms) < TOL and relerr_rms(got[rows], want) < TOL


@torch.no_grad()
def test_bulk_tail_chunk_of_69_rows_vs_oracle(dev, conv_precision):
    """The reference's bulk generator samples 7869 images per class in chunks of 200: its last chunk has 69 rows (scripts/helpers/sample_dataset.py:
    26-27,38).  No plan table holds B = 69 or B = 200: the planner's cost model and the Winograd rule decide (round 6).  69 unconditional samples at
    latent (8,32,32), device Philox noise, 12 DDIM iterations + decode; rows 0, 33 and 68 against the oracle fed the SAME global Philox rows --
    the odd batch also exercises the partial last tiles of the direct form (69 x 64 = 4416 = 34.5 x 128 rows at the 8 x 8 level, where the component
    GEMMs' tiles do not divide a component and the convolutions stay direct) next to the Winograd form at the 16 x 16 level."""
    if conv_precision != 5:
        pytest.skip("the default arithmetic (the batch-general plan rule is its planner's)")
    from medfusion_amd import published as P
    from medfusion_amd import blocks as BLK
    pipe = P.build_published_pipeline(dev, num_classes=None)
    ora = build_oracle_pipe(R.published_unet_kwargs(None), R.published_vae_kwargs(8), "published")
    rows = [0, 33, 68]
    torch.set_num_threads(min(32, torch.get_num_threads()))
    nz = S.PhiloxNoise(6900)
    ora.set_noise_fn(lambda like: nz(torch.empty((69, *like.shape[1:])))[rows])
    tr_o, tr_p = [], []
    want = ora.sample(3, (8, 32, 32), steps=12, use_ddim=True, trace=tr_o)
    got = pipe.sample(69, (8, 32, 32), steps=12, use_ddim=True, noise=M.PhiloxDeviceNoise(6900), trace=tr_p)
    assert got.shape == (69, 3, 256, 256)
    errs = [relerr_rows(tr_p[i][0][rows], tr_o[i][0]) for i in range(0, 12, 3)] + [relerr_rows(tr_p[-1][1][rows], tr_o[-1][1])]
    e_img, e_rms = relerr_rows(got[rows], want), relerr_rms(got[rows], want)
    d16, d8 = K.make_conv_desc(69, 16, 16, 512, 0, 512, 3, 1, 1, 0, precision=5), K.make_conv_desc(69, 8, 8, 1024, 0, 1024, 3, 1, 1, 0, precision=5)
    print(f"[measured] B = 69 (bulk tail chunk), rows {rows}: x0 per-sample rel-err every 3 iterations + final latents: " + " ".join(f"{e:.1e}" for e in errs) +
          f" | images {e_img:.1e} (RMS-relative {e_rms:.1e}) | Winograd form at 16 x 16: {K.wino_preferred(d16)}, at 8 x 8: {K.wino_preferred(d8)} "
          f"| direct plan of 512 -> 512 @16^2: {K.conv_plan(d16)}")
    import os
    if os.environ.get("MF_WINO_RULE", "1") != "0":      # (the A/B switch restores the round-5 exact-shape table, which has no B = 69 row)
        assert K.wino_preferred(d16) and not K.wino_ok(d8)
    assert max(errs) < TOL and e_img < TOL and e_rms < TOL


F16_TOL = 1e-2


@torch.no_grad()
def test_f16_single_term_opt_in_mode_has_its_own_tolerance(dev, published, conv_precision):
    """MF_CONV_F16 (opt-in, on the LDS-DMA kernel of the default arithmetic): one UNet evaluation and a 20-iteration
    trajectory at the published size against the fp32 oracle, with the mode's own tolerance (operands carry 11 significant bits)."""
    if conv_precision != 1:
        pytest.skip("runs once")
    from medfusion_amd import blocks as BLK
    ora, pipe = published
    BLK.CONV_PRECISION = 6
    try:
        x = S.synth_input("pub256_x", (2, 8, 32, 32))
        t = torch.tensor([731, 731])
        want, _ = ora.noise_estimator(x, t, None)
        got, _ = pipe.noise_estimator(x.to(dev), t.to(dev), None)
        e_unet = relerr(got, want)
        ora.set_noise_fn(S.PhiloxNoise(77))
        tr_o, tr_p = [], []
        want_img = ora.sample(1, (8, 32, 32), steps=20, use_ddim=True, trace=tr_o)
        got_img = pipe.sample(1, (8, 32, 32), steps=20, use_ddim=True, noise=oracle_noise(77), trace=tr_p)
        errs = [relerr(a[0], b[0]) for a, b in zip(tr_p, tr_o)]
        print(f"fp16 single-term mode: UNet {e_unet:.1e} | x0 along 20 iterations: " + " ".join(f"{e:.1e}" for e in errs[::3]) + f" | image {relerr(got_img, want_img):.1e}")
        assert 1e-5 < e_unet < F16_TOL and max(errs) < F16_TOL and relerr(got_img, want_img) < F16_TOL
    finally:
        BLK.CONV_PRECISION = conv_precision


@torch.no_grad()
def test_cold_diffusion_through_the_loop(dev):
    """`denoise(..., cold_diffusion=True)` (the reference forwards the flag to forward() on every iteration, diffusion_pipeline.py:294):
    no posterior draw, the DDIM update unchanged -- against the oracle on the tiny pipeline."""
    pipe = build_product_pipe(R.tiny_unet_kwargs(3, "none"), R.tiny_vae_kwargs(), "pipe_tiny", dev)
    ora = build_oracle_pipe(R.tiny_unet_kwargs(3, "none"), R.tiny_vae_kwargs(), "pipe_tiny")
    cond = torch.tensor([2, 0])
    for use_ddim in (True, False):
        drift = oracle_fp64_drift(ora, 61, 2, (8, 8, 8), condition=cond, guidance_scale=2.0, steps=4, use_ddim=use_ddim, cold_diffusion=True)
        want = ora.sample(2, (8, 8, 8), condition=cond, guidance_scale=2.0, steps=4, use_ddim=use_ddim, cold_diffusion=True)
        src = oracle_noise(61)
        got = pipe.sample(2, (8, 8, 8), condition=cond.to(dev), guidance_scale=2.0, steps=4, use_ddim=use_ddim, cold_diffusion=True, noise=src)
        assert src.draw_index == ora.noise_fn.draw
        e = relerr(got, want)
        bound = max(TOL, COLD_DRIFT_FACTOR * drift)
        print(f"[measured] cold diffusion through the loop, ddim={use_ddim}: {e:.1e} (the fp32 oracle vs its fp64 self on this case: {drift:.1e}; bound {bound:.1e})")
        assert e < bound

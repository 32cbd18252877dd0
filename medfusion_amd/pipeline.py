"""DiffusionPipeline (sampling half) on the HIP kernels -- mirror of
medical_diffusion/models/pipelines/diffusion_pipeline.py: ctor :20-74, forward :232-275, denoise :278-310,
sample :312-317, interpolate :320-332.  Same signatures, keyword names, quirks (Q1-Q17) and state-dict
prefixes (`noise_estimator.`, `noise_scheduler.`, `latent_embedder.`, `ema_model.averaged_model.`).

Differences, all outside the arithmetic: no streamlit/tqdm (Q16); noise comes from a NoiseSource
(`noise=` keyword; default = device Philox keyed from torch's default CPU generator, which it advances: consecutive
`sample()` calls differ and `torch.manual_seed(0)` before `sample()` is reproducible, like the reference harness); `shard=(rank, world)` keyword partitions the batch
rows across GPUs with shard-invariant noise.  Training (`_step`) is out of scope.
"""
from __future__ import annotations

import copy
import ctypes
import os
from typing import Optional

import torch
import torch.nn as nn
import torch.utils._python_dispatch

from . import kernels as K
from . import lib as L
from .noise import NoiseSource, default_noise
from .scheduler import GaussianNoiseScheduler


_CMD_POOLS = {}       # (device index, stream) -> torch.cuda.MemPool the recorded iteration of the command-list loop allocates from (process-wide)
_GRAPH_STREAMS = {}   # device index -> the one side stream graph captures run on


class _PureLaunchGuard(torch.utils._python_dispatch.TorchDispatchMode):
    """Active while the library records one loop iteration: notes every ATen operator that touches device memory other than allocation and
    metadata -- such work is not a launch of the library, so a replay of the recorded list would miss it."""
    ALLOWED = {"aten::empty", "aten::empty_like", "aten::empty_strided", "aten::new_empty", "aten::new_empty_strided", "aten::view", "aten::_unsafe_view",
               "aten::reshape", "aten::_reshape_alias", "aten::slice", "aten::select", "aten::as_strided", "aten::expand", "aten::unsqueeze", "aten::squeeze",
               "aten::t", "aten::transpose", "aten::permute", "aten::detach", "aten::alias", "aten::unbind", "aten::split", "aten::narrow",
               "aten::lift_fresh", "aten::is_pinned", "aten::stride", "aten::size", "aten::sym_size", "aten::sym_stride", "aten::sym_numel",
               "aten::sym_storage_offset", "aten::is_contiguous", "aten::contiguous", "aten::_to_copy", "aten::resize_"}

    def __init__(self):
        super().__init__()
        self.foreign = []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func._schema.name
        if name not in self.ALLOWED or name in ("aten::contiguous", "aten::_to_copy"):
            def on_device(v):
                if isinstance(v, torch.Tensor):
                    return v.is_cuda
                if isinstance(v, (list, tuple)):
                    return any(on_device(u) for u in v)
                return False
            touches = on_device(out) or on_device(args) or on_device(list((kwargs or {}).values()))
            # contiguous / _to_copy of a device tensor launch a copy kernel only when they really copy
            if touches and not (name == "aten::contiguous" and out is (args[0] if args else None)):
                self.foreign.append(name)
        return out


class EMAModel(nn.Module):
    """Inference view of utils/train_utils.py:5-88: only `averaged_model` is read (diffusion_pipeline.py:234-235)."""

    def __init__(self, model, **_kw):
        super().__init__()
        self.averaged_model = copy.deepcopy(model).eval()
        self.averaged_model.requires_grad_(False)


class DiffusionPipeline(nn.Module):
    def __init__(self, noise_scheduler, noise_estimator, latent_embedder=None, noise_scheduler_kwargs={}, noise_estimator_kwargs={},
                 latent_embedder_checkpoint="", estimator_objective="x_T", estimate_variance=False, use_self_conditioning=False,
                 classifier_free_guidance_dropout=0.5, num_samples=4, do_input_centering=True, clip_x0=True, use_ema=False, ema_kwargs={},
                 **_training_only):
        super().__init__()
        ne_kwargs = dict(noise_estimator_kwargs)  # the reference mutates the caller's dict (:50-51); we do not
        ne_kwargs["estimate_variance"] = estimate_variance
        ne_kwargs["use_self_conditioning"] = use_self_conditioning
        # classes + kwargs like the reference, or ready-made instances
        self.noise_scheduler = noise_scheduler(**dict(noise_scheduler_kwargs)) if isinstance(noise_scheduler, type) else noise_scheduler
        self.noise_estimator = noise_estimator(**ne_kwargs) if isinstance(noise_estimator, type) else noise_estimator
        if latent_embedder is None:
            self.latent_embedder = None
        elif isinstance(latent_embedder, type):
            from .checkpoint import load_module_from_checkpoint

            self.latent_embedder = load_module_from_checkpoint(latent_embedder, latent_embedder_checkpoint)
        else:
            self.latent_embedder = latent_embedder
        if self.latent_embedder is not None:
            for p in self.latent_embedder.parameters():
                p.requires_grad = False
        self.estimator_objective = estimator_objective
        self.use_self_conditioning = use_self_conditioning
        self.num_samples = num_samples
        self.classifier_free_guidance_dropout = classifier_free_guidance_dropout
        self.do_input_centering = do_input_centering
        self.estimate_variance = estimate_variance
        self.clip_x0 = clip_x0
        self.batch_cfg = True  # classifier-free guidance as one 2B-row UNet call (same arithmetic per row)
        self.hoist_embeddings = True  # denoise(): time/label/local embeddings of all iterations evaluated once, before the loop
        # (the memory pools of the command-list loop and the capture streams live in module-level registries keyed by device, _CMD_POOLS /
        # _GRAPH_STREAMS below: a torch.cuda.MemPool or Stream held by the nn.Module would break copy.deepcopy / pickling of the pipeline)
        self.last_cmdlist_launches = 0          # launches in the list the last command-list loop replayed (0: it ran eagerly)
        self.last_cmdlist_foreign_ops = []      # ATen operators that put device work into the recorded iteration (replay refused)
        self.time_cmdlist = False               # measurement aid: time the host side of one replayed iteration (costs a device sync)
        self.last_cmdlist_host_ms = None
        self.use_ema = use_ema
        if use_ema:
            self.ema_model = EMAModel(self.noise_estimator, **ema_kwargs)

    @property
    def device(self):
        return next(self.parameters()).device

    @classmethod
    def load_from_checkpoint(cls, path, map_location=None, **overrides):
        from .checkpoint import load_pipeline_from_checkpoint

        return load_pipeline_from_checkpoint(cls, path, map_location=map_location, **overrides)

    # ------------------------------------------------------------------ one denoise iteration
    def _estimator(self):
        return self.ema_model.averaged_model if self.use_ema else self.noise_estimator

    def _predict(self, x_t, t, condition, self_cond, guidance_scale, un_cond, emb=None):
        """UNet call(s) of forward :240-253.  Returns (pred, pred_uncond|None, pred_var|None).
        emb = (table, i, cols_cond, cols_uncond, cols_uncond ++ cols_cond): the loop's precomputed embeddings
        (UNet.precompute_embeddings), or None."""
        est = self._estimator()
        cfg = (condition is not None) and (guidance_scale != 1.0)
        ec = (lambda cols: None) if emb is None else (lambda cols: est.step_embeddings(emb[0], emb[1], cols))
        if cfg:
            if self.estimate_variance:
                raise RuntimeError("estimate_variance with guidance_scale != 1 raises in the reference too "
                                   "(diffusion_pipeline.py:243-249 never chunks `pred`); use guidance_scale=1")
            if self.batch_cfg and not self.use_self_conditioning:
                # both passes of diffusion_pipeline.py:242-243 as ONE UNet call over 2B rows (rows are independent):
                # rows [0,B) = un-guided (condition = un_cond, possibly None), rows [B,2B) = guided.
                pred2 = est.forward_cfg_pair(x_t, t, condition, un_cond, emb_cache=None if emb is None else ec(emb[4]))
                B = x_t.shape[0]
                return pred2[B:], pred2[:B], None
            pred_uncond, _ = est(x_t, t, condition=un_cond, self_cond=self_cond, emb_cache=None if emb is None else ec(emb[3]))  # un-guided pass FIRST (Q6)
            pred_cond, _ = est(x_t, t, condition=condition, self_cond=self_cond, emb_cache=None if emb is None else ec(emb[2]))
            return pred_cond, pred_uncond, None
        if self.estimate_variance:
            pred, pred_var = est.forward_split(x_t, t, condition=condition, self_cond=self_cond, emb_cache=None if emb is None else ec(emb[2]))
            return pred, None, pred_var
        pred, _ = est(x_t, t, condition=condition, self_cond=self_cond, emb_cache=None if emb is None else ec(emb[2]))
        return pred, None, None

    @torch.no_grad()
    def forward(self, x_t, t, condition=None, self_cond=None, guidance_scale=1.0, cold_diffusion=False, un_cond=None, noise=None):
        """One reverse step like diffusion_pipeline.py:232-275 -> (x_t_prior, x_0, x_T, self_cond).
        `noise`: the posterior draw of gaussian_scheduler.py:99 (tensor); N(0,1) from the default source if None."""
        if self.estimator_objective not in ("x_T", "x_0"):
            raise ValueError("Unknown Objective")
        sch = self.noise_scheduler
        pred, pred_uncond, pred_var = self._predict(x_t, t, condition, self_cond, guidance_scale, un_cond)
        if cold_diffusion:  # off the hot loop: composed from the per-row scheduler API (gaussian_scheduler.py:88-93)
            if pred_var is not None:
                raise NotImplementedError("cold_diffusion with a learned variance head")
            if pred_uncond is not None:
                g = torch.full((x_t.shape[0],), float(guidance_scale), device=x_t.device)
                m1 = torch.full((x_t.shape[0],), -1.0, device=x_t.device)
                pred = K.rows_axpby(pred_uncond, None, K.rows_axpby(pred, None, pred_uncond, m1), g)   # pu + g*(pc - pu)
            if self.estimator_objective == "x_0":
                prior, x0 = sch.estimate_x_t_prior_from_x_0(x_t, t, pred, clip_x0=self.clip_x0, cold_diffusion=True)
                xT = sch.estimate_x_T(x_t, x_0=pred, t=t, clip_x0=self.clip_x0)
                return prior, x0, xT, xT
            prior, x0 = sch.estimate_x_t_prior_from_x_T(x_t, t, pred, clip_x0=self.clip_x0, cold_diffusion=True)
            return prior, x0, pred, x0
        rec = sch.step_records([sch._uniform_t(t)], use_ddim=False)[0]
        table = sch.upload_records([rec], x_t.device)
        if noise is None:
            src = default_noise()
            src.begin(x_t.shape[0], x_t.device)
            noise = src.draw(tuple(x_t.shape))
        x_t = x_t.contiguous()
        prior, x0, xT = torch.empty_like(x_t), torch.empty_like(x_t), torch.empty_like(x_t)
        a = L.MfSchedArgs(x_t.data_ptr(), pred.data_ptr(), None if pred_uncond is None else pred_uncond.data_ptr(),
                          None if pred_var is None else pred_var.data_ptr(), noise.data_ptr(), None, 0, prior.data_ptr(), x0.data_ptr(),
                          xT.data_ptr(), table.data_ptr(), None, 0, 0 if self.estimator_objective == "x_T" else 1, int(bool(self.clip_x0)),
                          float(guidance_scale), x_t.numel())
        K.sched_step(a, outputs=(prior, x0, xT))
        self_cond = x0 if self.estimator_objective == "x_T" else xT
        return prior, x0, xT, self_cond

    # ------------------------------------------------------------------ the loop
    @torch.no_grad()
    def denoise(self, x_t, steps=None, condition=None, use_ddim=True, noise: Optional[NoiseSource] = None, trace=None, decode=True, use_graph=None,
                loop: Optional[str] = None, progress_cb=None, **kwargs):
        """diffusion_pipeline.py:278-310.  kwargs: guidance_scale, un_cond, cold_diffusion (forwarded to forward()
        by the reference); `eta` raises like the reference's forward() would (Q2).
        progress_cb(done, total): the hook that stands where the reference drives `st.progress` and `tqdm` (diffusion_pipeline.py:289-291; SURVEY Q16:
        no streamlit import here).  Called on the host after the iterations up to `done` have been ENQUEUED -- the loop never waits for the device;
        a callback that wants finished iterations synchronises the stream itself.  The replayed loops (command list, graph) call it after the eager
        first iteration, then every max(1, total // 20) iterations; the Python loop every iteration.  The counts a caller sees are STRICTLY
        INCREASING: when the loop has to be re-run (a fused rendezvous timed out, K.with_fused_fallback) the counts already reported are not
        reported again."""
        if not x_t.is_cuda:
            raise RuntimeError("medfusion_amd.DiffusionPipeline runs on a ROCm device only (no CPU fallback)")
        with torch.cuda.device(x_t.device):   # (see sample())
            from .noise import PhiloxDeviceNoise
            draws0 = None if noise is None else noise.draw_index
            trace0 = None if trace is None else len(trace)

            def rewind():   # (a rendezvous of a fused conv + GroupNorm launch timed out: K.with_fused_fallback re-runs the loop un-fused)
                if trace is not None:
                    del trace[trace0:]
                if noise is None:
                    return
                if not isinstance(noise, PhiloxDeviceNoise):
                    raise RuntimeError("medfusion_amd: the sampling loop must be re-run on the two-launch GroupNorm form, but its host noise source "
                                       "cannot be rewound: set MEDFUSION_FUSED_APPLY=0 when several processes share one GPU")
                noise.draw_index = draws0

            if progress_cb is not None:
                user_cb, reported = progress_cb, [0]

                def progress_cb(done, total):   # monotone across a re-run of the loop
                    if done > reported[0]:
                        reported[0] = done
                        user_cb(done, total)

            return K.with_fused_fallback(x_t.device, lambda: self._denoise(x_t, steps, condition, use_ddim, noise, trace, decode, use_graph, loop, progress_cb,
                                                                           **dict(kwargs)), rewind)

    def _denoise(self, x_t, steps, condition, use_ddim, noise, trace, decode, use_graph, loop, progress_cb=None, **kwargs):
        K.SyncWords.reset(x_t.device)   # (the split-K counters of the convolutions: zero by invariant, re-zeroed once per loop for robustness)
        if "eta" in kwargs:
            raise TypeError("forward() got an unexpected keyword argument 'eta'")
        guidance_scale = kwargs.pop("guidance_scale", 1.0)
        un_cond = kwargs.pop("un_cond", None)
        cold_diffusion = bool(kwargs.pop("cold_diffusion", False))
        if kwargs:
            raise TypeError(f"forward() got an unexpected keyword argument '{next(iter(kwargs))}'")
        if self.estimator_objective not in ("x_T", "x_0"):
            raise ValueError("Unknown Objective")
        if not x_t.is_cuda:
            raise RuntimeError("medfusion_amd.DiffusionPipeline runs on a ROCm device only (no CPU fallback)")
        if use_graph and trace is not None:
            raise ValueError("use_graph=True cannot record a trace (the captured step is replayed, nothing returns to the host)")
        # How the loop body reaches the device (all three give the same bits: test_hipgraph_captured_step_equals_eager,
        # test_cmdlist_replay_equals_eager):
        #   "cmdlist" (default where possible): iteration 0 eager, iteration 1 recorded by the library while it runs (mf_cmdlist_begin/end),
        #             every further iteration re-issued from C (mf_cmdlist_replay): stream-ordered launches at ~1 us of host time each
        #             instead of 2.0 ms of Python -> ctypes work per iteration (profiles/r03_host_enqueue_time.txt);
        #   "graph":  the iteration as ONE captured hipGraph (use_graph=True; BASELINE configs[3] names it): no host work either, but every
        #             kernel node dispatches ~1 us later than a stream-ordered launch -- slower than the eager loop from B ~ 6 on;
        #   "eager":  the Python loop (needed for a trace, a host noise source, cold diffusion).
        from .noise import PhiloxDeviceNoise
        replayable = (trace is None and not cold_diffusion and (noise is None or isinstance(noise, PhiloxDeviceNoise)) and (steps is None or steps >= 4))
        if loop is not None and loop not in ("cmdlist", "graph", "eager"):
            raise ValueError(f"loop={loop!r}: 'cmdlist', 'graph' or 'eager'")
        # precedence: loop= argument, use_graph= (True: graph, False: eager), MEDFUSION_LOOP in the environment, then the default
        mode = loop or ("graph" if use_graph else "eager" if use_graph is False else os.environ.get("MEDFUSION_LOOP", ""))
        explicit = loop is not None or use_graph is not None
        if mode not in ("cmdlist", "graph", "eager"):
            mode = "cmdlist"
        if mode == "cmdlist" and (not replayable or K.prof_active()):
            if loop == "cmdlist" and not replayable:
                raise ValueError("loop='cmdlist' replays recorded launches: no trace, no host noise source, no cold diffusion, >= 4 iterations")
            mode = "eager"
        if mode == "graph" and not replayable and not explicit:
            mode = "eager"
        use_graph = mode == "graph"
        sch = self.noise_scheduler
        dev = x_t.device
        B = x_t.shape[0]
        timesteps, steps = sch.loop_timesteps(steps, use_ddim)
        recs = sch.step_records(timesteps, use_ddim)
        table = sch.upload_records(recs, dev)
        if noise is None:  # x_t supplied by the caller (interpolate): fresh source, draws start at 0
            noise = default_noise()
            noise.begin(B, dev)
        rev = list(reversed(timesteps))
        objective = 0 if self.estimator_objective == "x_T" else 1
        x_t = x_t.contiguous().clone()
        if cold_diffusion:
            # diffusion_pipeline.py:294 forwards the flag to forward() on every iteration: off the hot loop, so composed from the
            # single-step API (forward() takes no posterior draw in this mode; the DDIM update and its draw are those of :297-304)
            if use_graph or trace is not None:
                raise ValueError("cold_diffusion runs through the single-step API: no graph capture, no trace")
            for i, t in enumerate(rev):
                tt = torch.full((B,), int(t), dtype=torch.long, device=dev)
                x_prior, x_0, x_T, _ = self.forward(x_t, tt, condition, None, guidance_scale=guidance_scale, cold_diffusion=True, un_cond=un_cond)
                if use_ddim and i < len(rev) - 1:
                    r = recs[i]
                    n_ddim = noise.draw(tuple(x_t.shape))
                    a = torch.full((B,), r.ddim_sqrt_an, dtype=torch.float32, device=dev)
                    c = torch.full((B,), r.ddim_c, dtype=torch.float32, device=dev)
                    sg = torch.full((B,), r.ddim_sigma, dtype=torch.float32, device=dev)
                    x_t = K.rows_axpby(K.rows_axpby(x_0, a, x_T, c), None, n_ddim, sg)   # x0 sqrt(a') + c x_T + sigma noise
                else:
                    x_t = x_prior
                if progress_cb is not None:
                    progress_cb(i + 1, len(rev))
            if decode and self.latent_embedder is not None:
                x_t = self.latent_embedder.decode(x_t)
            return x_t
        if mode in ("graph", "cmdlist"):
            self._denoise_graph(x_t, rev, recs, table, condition, guidance_scale, un_cond, use_ddim, noise, objective, cmdlist=(mode == "cmdlist"),
                                progress_cb=progress_cb)
        else:
            t_all = torch.tensor(rev, dtype=torch.float32, device=dev).reshape(-1, 1).expand(-1, B).contiguous()  # t.expand(B) per iteration (Q7)
            n_post = torch.empty_like(x_t)
            n_ddim = torch.empty_like(x_t)
            x0 = torch.empty_like(x_t)
            self_cond = None
            est = self._estimator()
            emb_tab = self._hoisted_embeddings(est, t_all[:, 0].contiguous(), condition, un_cond, B, dev)
            for i in range(len(rev)):
                pred, pred_uncond, pred_var = self._predict(x_t, t_all[i], condition, self_cond, guidance_scale, un_cond,
                                                            emb=None if emb_tab is None else (emb_tab[0], i, emb_tab[1], emb_tab[2], emb_tab[3]))
                noise.draw(tuple(x_t.shape), out=n_post)          # gaussian_scheduler.py:99 -- drawn on every iteration (Q3)
                ddim = recs[i].mode == 1
                if ddim:
                    noise.draw(tuple(x_t.shape), out=n_ddim)      # diffusion_pipeline.py:303
                a = L.MfSchedArgs(x_t.data_ptr(), pred.data_ptr(), None if pred_uncond is None else pred_uncond.data_ptr(),
                                  None if pred_var is None else pred_var.data_ptr(), n_post.data_ptr(), n_ddim.data_ptr() if ddim else None, 0,
                                  x_t.data_ptr(), x0.data_ptr(), None, table.data_ptr(), None, i, objective, int(bool(self.clip_x0)),
                                  float(guidance_scale), x_t.numel())
                K.sched_step(a, outputs=(x_t, x0))
                self_cond = x0 if self.use_self_conditioning else None  # only None-ness matters downstream (Q11)
                if trace is not None:
                    trace.append((x0.clone(), x_t.clone()))
                if progress_cb is not None:
                    progress_cb(i + 1, len(rev))
        if decode and self.latent_embedder is not None:
            x_t = self.latent_embedder.decode(x_t)
        return x_t

    def _hoisted_embeddings(self, est, t_steps, condition, un_cond, B, dev):
        """All timesteps of a loop are known before it starts: the embedding path leaves the loop (UNet.precompute_embeddings; bit-identical
        to evaluating it per iteration).  -> (table, columns of `condition`, columns of `un_cond`) or None when the estimator cannot."""
        if not (self.hoist_embeddings and hasattr(est, "can_precompute_embeddings") and est.can_precompute_embeddings()):
            return None
        has_c = est.cond_embedder is not None
        used = set()   # labels that occur in this loop (one host read before the loop starts)
        for lab in (condition, un_cond):
            if has_c and lab is not None:
                used.update(int(v) for v in lab.reshape(-1).tolist())
        tab = est.precompute_embeddings(t_steps, classes=used if has_c else None)
        cc, cu = est.embedding_columns(condition if has_c else None, B, dev, tab), est.embedding_columns(un_cond if has_c else None, B, dev, tab)
        return (tab, cc, cu, torch.cat([cu, cc]))   # (the last: columns of the 2B-row classifier-free-guidance pair, un-guided rows first)

    def _denoise_graph(self, x_t, rev, recs, table, condition, guidance_scale, un_cond, use_ddim, noise, objective, cmdlist=False, progress_cb=None):
        """The loop body as ONE captured hipGraph replayed `steps` times (BASELINE.json configs[3]) -- or, cmdlist=True, recorded by the
        library while it runs once and re-issued from C (mf_cmdlist_*: the native command list of the loop body).  Everything the
        reference reads on the host each iteration (t, alphas_cumprod[t], the t==0 test, the RNG state) is indexed by a
        DEVICE step counter: `t` is broadcast from a device table, the scheduler scalars come from the MfSchedStep table,
        the Philox draw index is draw_base + stride*step, and the graph advances the counter itself.  The last DDIM
        iteration still fills the (unused) DDIM noise buffer: counter-based draws do not shift any other draw."""
        from .noise import PhiloxDeviceNoise

        if not isinstance(noise, PhiloxDeviceNoise):
            raise RuntimeError("use_graph=True needs the device Philox noise source (a host generator cannot be captured)")
        dev, B = x_t.device, x_t.shape[0]
        t_table = torch.tensor(rev, dtype=torch.float32, device=dev)
        counter = torch.zeros(2, dtype=torch.int32, device=dev)    # (step, ticket word of the fused tail launch)
        step_dev = counter[:1]
        t_cur = torch.zeros(B, dtype=torch.float32, device=dev)   # (filled per iteration only when the estimator reads t: not with hoisted embeddings)
        n_post, n_ddim, x0 = torch.empty_like(x_t), torch.empty_like(x_t), torch.empty_like(x_t)
        stride = 2 if use_ddim else 1
        base = noise.draw_index  # draws consumed so far (x_T)
        clip, g = int(bool(self.clip_x0)), float(guidance_scale)

        emb_tab = self._hoisted_embeddings(self._estimator(), t_table, condition, un_cond, B, dev)
        emb = None if emb_tab is None else (emb_tab[0], step_dev, emb_tab[1], emb_tab[2], emb_tab[3])   # rows of iteration *step_dev, gathered on the device

        # the tail of an iteration -- posterior draw, DDIM draw, scheduler step, counter += 1 -- as ONE launch (mf_sched_step_philox_f32: the same
        # Philox quads and the same scheduler arithmetic, bit for bit), unless MEDFUSION_LOOP_TAIL=0 (A/B) or the latent is not a whole
        # number of quads per sample
        fused_tail = os.environ.get("MEDFUSION_LOOP_TAIL", "1") != "0" and (x_t.numel() // B) % 4 == 0

        def body():
            if emb is None:   # (with the embedding rows hoisted out of the loop the estimator never reads t)
                K.broadcast_from_table(t_table, step_dev, t_cur)
            pred, pred_uncond, pred_var = self._predict(x_t, t_cur, condition, None if not self.use_self_conditioning else x0, g, un_cond, emb=emb)
            if fused_tail:
                a = L.MfSchedArgs(x_t.data_ptr(), pred.data_ptr(), None if pred_uncond is None else pred_uncond.data_ptr(),
                                  None if pred_var is None else pred_var.data_ptr(), None, None, 0, x_t.data_ptr(), x0.data_ptr(), None, table.data_ptr(),
                                  step_dev.data_ptr(), 0, objective, clip, g, x_t.numel())
                K.sched_step_philox(a, noise._seed, base, stride, noise.sample_offset, B, counter, outputs=(x_t, x0))
                return pred
            noise.draw_indexed(n_post, base, stride, step_dev)
            if use_ddim:
                noise.draw_indexed(n_ddim, base + 1, stride, step_dev)
            a = L.MfSchedArgs(x_t.data_ptr(), pred.data_ptr(), None if pred_uncond is None else pred_uncond.data_ptr(),
                              None if pred_var is None else pred_var.data_ptr(), n_post.data_ptr(), n_ddim.data_ptr() if use_ddim else None, 0,
                              x_t.data_ptr(), x0.data_ptr(), None, table.data_ptr(), step_dev.data_ptr(), 0, objective, clip, g, x_t.numel())
            K.sched_step(a, outputs=(x_t, x0))
            K.counter_add(step_dev, 1)
            return pred  # keep alive until the end of capture

        def first_iteration():
            # Q11: with self-conditioning the first call sees self_cond=None -> run it eagerly in that form
            if self.use_self_conditioning:
                if emb is None:
                    K.broadcast_from_table(t_table, step_dev, t_cur)
                pred, pu, pv = self._predict(x_t, t_cur, condition, None, g, un_cond, emb=emb)
                noise.draw_indexed(n_post, base, stride, step_dev)
                if use_ddim:
                    noise.draw_indexed(n_ddim, base + 1, stride, step_dev)
                a = L.MfSchedArgs(x_t.data_ptr(), pred.data_ptr(), None if pu is None else pu.data_ptr(), None if pv is None else pv.data_ptr(),
                                  n_post.data_ptr(), n_ddim.data_ptr() if use_ddim else None, 0, x_t.data_ptr(), x0.data_ptr(), None, table.data_ptr(),
                                  step_dev.data_ptr(), 0, objective, clip, g, x_t.numel())
                K.sched_step(a, outputs=(x_t, x0))
                K.counter_add(step_dev, 1)
            else:
                body()  # eager warm-up iteration 0: sizes every workspace / packs weights on THIS stream

        if cmdlist:
            # the native command list: iteration 0 eager (weights packed, workspaces sized, kernel attributes set), iteration 1 recorded
            # while it runs -- inside a private memory pool, so that every buffer the recorded launches point at stays reserved, what a
            # graph capture's pool does -- and iterations 2 .. re-issued from C on the same stream
            lib = L.load()
            cur = K.stream(dev.index)
            first_iteration()
            if progress_cb is not None:
                progress_cb(1, len(rev))
            if len(rev) > 1:
                pool = _CMD_POOLS.get((dev.index, cur))    # one pool per (device, stream): two threads driving two streams record independently
                if pool is None:
                    pool = _CMD_POOLS[(dev.index, cur)] = torch.cuda.MemPool()
                handle = ctypes.c_void_p()
                guard = _PureLaunchGuard()
                grew = K.scratch_growth_count()
                try:
                    with torch.cuda.use_mem_pool(pool, device=dev), guard:
                        L.check(lib.mf_cmdlist_begin(), "mf_cmdlist_begin")
                        try:
                            keep = body()
                        finally:
                            L.check(lib.mf_cmdlist_end(ctypes.byref(handle)), "mf_cmdlist_end")
                except BaseException:
                    if handle:
                        lib.mf_cmdlist_free(handle)   # (body() raised while recording: the list is dropped with it)
                    raise
                try:
                    self.last_cmdlist_launches = lib.mf_cmdlist_count(handle)
                    self.last_cmdlist_foreign_ops = sorted(set(guard.foreign))
                    if K.scratch_growth_count() != grew:
                        # the workspace or the split-K counters GREW during the recorded iteration (it needed more scratch than iteration 0):
                        # launches recorded before the growth point at the released buffer -- the list is not replayable
                        guard.foreign.append("medfusion_amd: scratch growth inside the recorded iteration")
                        self.last_cmdlist_foreign_ops = sorted(set(guard.foreign))
                    if guard.foreign:
                        # torch itself put device work into the iteration (self-conditioning, an attention variant, ...): the recorded list
                        # is not the whole iteration -- drop it and run the remaining iterations through Python (same bits, host-bound)
                        self.last_cmdlist_launches = 0
                        for k in range(2, len(rev)):
                            body()
                            if progress_cb is not None:
                                progress_cb(k + 1, len(rev))
                    elif len(rev) > 2:
                        left = len(rev) - 2
                        if self.time_cmdlist and left > 1:   # measurement aid (scripts/enqueue_time.py): the host cost of ONE replayed
                            import time                       # iteration with an empty queue in front of it
                            torch.cuda.synchronize(dev)
                            t0 = time.perf_counter()
                            L.check(lib.mf_cmdlist_replay(handle, 1, cur), "mf_cmdlist_replay")
                            self.last_cmdlist_host_ms = (time.perf_counter() - t0) * 1e3
                            left -= 1
                        if progress_cb is None:
                            L.check(lib.mf_cmdlist_replay(handle, left, cur), "mf_cmdlist_replay")
                        else:   # the same launches in slices of ~5 % of the loop, the callback between them
                            chunk, done = max(1, len(rev) // 20), len(rev) - left
                            progress_cb(done, len(rev))
                            while left > 0:
                                k = min(chunk, left)
                                L.check(lib.mf_cmdlist_replay(handle, k, cur), "mf_cmdlist_replay")
                                left, done = left - k, done + k
                                progress_cb(done, len(rev))
                finally:
                    lib.mf_cmdlist_free(handle)    # (the kernarg bytes were copied at every launch)
                del keep
            noise.draw_index = base + stride * len(rev)
            return
        side = _GRAPH_STREAMS.get(dev.index)      # ONE capture stream per device (per-stream workspaces / split-K counters stay bounded)
        if side is None:
            side = _GRAPH_STREAMS[dev.index] = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            K.SyncWords.reset(dev)    # (the counters of THIS stream: _denoise reset those of the caller's stream)
            first_iteration()
            if progress_cb is not None:
                progress_cb(1, len(rev))
            done = 1
            if len(rev) > done:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    keep = body()
                chunk = max(1, len(rev) // 20)
                for k in range(done, len(rev)):
                    graph.replay()
                    if progress_cb is not None and ((k + 1) % chunk == 0 or k + 1 == len(rev)):
                        progress_cb(k + 1, len(rev))
                del keep
        torch.cuda.current_stream(dev).wait_stream(side)
        noise.draw_index = base + stride * len(rev)

    @torch.no_grad()
    def sample(self, num_samples, img_size, condition=None, noise: Optional[NoiseSource] = None, shard=None, **kwargs):
        """diffusion_pipeline.py:312-317.  `shard=(rank, world)`: this process generates rows
        [rank*num_samples/world, (rank+1)*num_samples/world) of the global batch (condition/un_cond are global)."""
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("medfusion_amd.DiffusionPipeline.sample: move the pipeline to a ROCm device first (no CPU fallback)")
        lo, hi = 0, num_samples
        if shard is not None:
            from .dist import shard_rows

            lo, hi = shard_rows(num_samples, shard[0], shard[1])
            if condition is not None:
                condition = condition[lo:hi]
            if kwargs.get("un_cond") is not None:
                kwargs["un_cond"] = kwargs["un_cond"][lo:hi]
        if noise is None:
            noise = default_noise()
        if hi == lo:   # more ranks than samples: this rank's shard is empty -- no launch, the gather pads it (dist.gather_images)
            empty = torch.empty((0, *img_size), dtype=torch.float32, device=dev)
            if kwargs.get("decode", True) and self.latent_embedder is not None:
                return self.latent_embedder.decode(empty)
            return empty
        with torch.cuda.device(dev):   # launches go to the CURRENT device's stream: make the pipeline's device current for the call
            noise.begin(hi - lo, dev, sample_offset=lo, global_batch=num_samples)
            x_T = noise.draw((hi - lo, *img_size))  # noise_scheduler.x_final(template): draw #0 (Q3)
            return self.denoise(x_T, condition=condition, noise=noise, **kwargs)

    @torch.no_grad()
    def interpolate(self, img1, img2, i=None, condition=None, lam=0.5, noise: Optional[NoiseSource] = None, **kwargs):
        """diffusion_pipeline.py:320-332: diffuse both inputs to timestep i, lerp, denoise with `i` passed as `steps`.
        NB the reference method ALWAYS raises (it forwards `clip_x0=` to estimate_x_t, which has no such parameter); this
        implements its evident intent (what oracle/restate.py restates).  `i=None` raises TypeError like the reference's
        `torch.full(shape, None)`.  Noise draws: #0 and #1 = the two x_T of estimate_x_t, then the denoise loop's."""
        assert img1.shape == img2.shape, "Image 1 and 2 must have equal shape"
        if i is None:
            raise TypeError("full() received an invalid combination of arguments - got (torch.Size, NoneType, device=torch.device)")
        if noise is None:
            noise = default_noise()
        B = img1.shape[0]
        noise.begin(B, img1.device)
        t = torch.full(img1.shape[:1], i, device=img1.device)
        sch = self.noise_scheduler
        img1_t = sch.estimate_x_t(img1, t=t, noise=noise.draw(tuple(img1.shape)))
        img2_t = sch.estimate_x_t(img2, t=t, noise=noise.draw(tuple(img2.shape)))
        a = torch.full((B,), 1 - lam, dtype=torch.float32, device=img1.device)
        c = torch.full((B,), lam, dtype=torch.float32, device=img1.device)
        img = K.rows_axpby(img1_t, a, img2_t, c)  # (1 - lam) * img1_t + lam * img2_t
        return self.denoise(img, i, condition, noise=noise, **kwargs)

"""Noise scheduler -- mirror of medical_diffusion/models/noise_schedulers/{scheduler_base,gaussian_scheduler}.py.

Tables are built exactly like the reference (fp64 -> fp32 buffers, same buffer names, gaussian_scheduler.py:9-58).
The per-iteration scalar algebra of the denoise loop (diffusion_pipeline.py:285-304 and
gaussian_scheduler.py:95-124) is evaluated ON THE HOST with the same fp32 torch scalar ops the reference
uses, once per `denoise` call, into a device table of `MfSchedStep` records: the reference's per-step host
syncs (`alphas_cumprod[t]`, `std[t==0]=0`) disappear and the fused step kernel becomes graph-capturable.
Tensor-level methods (estimate_x_0, ...) run on the GPU through the same fused kernel.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import kernels as K
from . import lib as L


class BasicNoiseScheduler(nn.Module):
    """scheduler_base.py:7-46"""

    def __init__(self, timesteps=1000, T=None):
        super().__init__()
        self.timesteps = timesteps
        self.T = timesteps if T is None else T
        self.register_buffer("timesteps_array", torch.linspace(0, self.T - 1, self.timesteps, dtype=torch.long))

    def __len__(self):
        return self.timesteps

    @staticmethod
    def extract(x, t, ndim):
        return x.gather(0, t).reshape(-1, *((1,) * (ndim - 1)))


class GaussianNoiseScheduler(BasicNoiseScheduler):
    TABLES = ("betas", "alphas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
              "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1", "posterior_mean_coef2", "posterior_variance")

    def __init__(self, timesteps=1000, T=None, schedule_strategy="cosine", beta_start=0.0001, beta_end=0.02, betas=None):
        super().__init__(timesteps, T)
        self.schedule_strategy = schedule_strategy
        if betas is not None:
            betas = torch.as_tensor(betas, dtype=torch.float64)
        elif schedule_strategy == "linear":
            betas = torch.linspace(beta_start, beta_end, timesteps, dtype=torch.float64)
        elif schedule_strategy == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, timesteps, dtype=torch.float64) ** 2
        elif schedule_strategy == "cosine":
            s = 0.008
            x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64)
            ac = torch.cos(((x / timesteps) + s) / (1 + s) * torch.pi * 0.5) ** 2
            ac = ac / ac[0]
            betas = torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)
        else:
            raise NotImplementedError(f"{schedule_strategy} does is not implemented for {self.__class__}")
        alphas = 1 - betas
        ac = torch.cumprod(alphas, dim=0)
        ac_prev = F.pad(ac[:-1], (1, 0), value=1.0)
        reg = lambda name, val: self.register_buffer(name, val.to(torch.float32))
        reg("betas", betas)
        reg("alphas", alphas)
        reg("alphas_cumprod", ac)
        reg("alphas_cumprod_prev", ac_prev)
        reg("sqrt_alphas_cumprod", torch.sqrt(ac))
        reg("sqrt_one_minus_alphas_cumprod", torch.sqrt(1.0 - ac))
        reg("sqrt_recip_alphas_cumprod", torch.sqrt(1.0 / ac))
        reg("sqrt_recipm1_alphas_cumprod", torch.sqrt(1.0 / ac - 1))
        reg("posterior_mean_coef1", betas * torch.sqrt(ac_prev) / (1.0 - ac))
        reg("posterior_mean_coef2", (1.0 - ac_prev) * torch.sqrt(alphas) / (1.0 - ac))
        reg("posterior_variance", betas * (1.0 - ac_prev) / (1.0 - ac))
        self._host = None

    # ------------------------------------------------------------------ host-side scalar algebra
    def host_tables(self) -> dict:
        """CPU fp32 copies of the buffers (they may live on the GPU after .to(device))."""
        key = tuple(getattr(self, n)._version for n in self.TABLES)
        if self._host is None or self._host[0] != key:
            self._host = (key, {n: getattr(self, n).detach().to("cpu", torch.float32).clone() for n in self.TABLES})
        return self._host[1]

    def loop_timesteps(self, steps: Optional[int], use_ddim: bool) -> Tuple[List[int], int]:
        """diffusion_pipeline.py:283-287: DDIM -> truncated linspace (Q4); else the FIRST `steps` entries (Q5)."""
        if use_ddim:
            steps = self.timesteps if steps is None else steps
            arr = torch.linspace(0, self.T - 1, steps, dtype=torch.long)
        else:
            arr = self.timesteps_array.detach().cpu()[slice(0, steps)]
        return [int(v) for v in arr], (steps if steps is not None else len(arr))

    def step_records(self, timesteps: List[int], use_ddim: bool, eta=1) -> List[L.MfSchedStep]:
        """One record per loop iteration i (t = reversed(timesteps)[i]), every scalar computed with the same
        fp32 torch ops as the reference: gaussian_scheduler.py:95-98,110-116 and diffusion_pipeline.py:297-302."""
        tb = self.host_tables()
        steps = len(timesteps)
        recs = []
        rev = list(reversed(timesteps))
        for i, t in enumerate(rev):
            tt = torch.tensor([t])
            var_min = torch.log(tb["posterior_variance"].gather(0, tt).clamp(min=1e-20))
            var_max = torch.log(tb["betas"].gather(0, tt).clamp(min=1e-20))
            variance = 0 * var_max + (1 - 0) * var_min  # var_scale == 0 (python int) when estimate_variance is off
            std = torch.exp(0.5 * variance)
            std[tt == 0] = 0.0
            r = L.MfSchedStep()
            r.sqrt_recip_ac = float(tb["sqrt_recip_alphas_cumprod"][t])
            r.sqrt_recipm1_ac = float(tb["sqrt_recipm1_alphas_cumprod"][t])
            r.coef1 = float(tb["posterior_mean_coef1"][t])
            r.coef2 = float(tb["posterior_mean_coef2"][t])
            r.std_fixed = float(std[0])
            r.log_var_min = float(var_min[0])
            r.log_var_max = float(var_max[0])
            r.t = int(t)
            if use_ddim and (steps - i - 1 > 0):
                t_next = timesteps[steps - i - 2]
                alpha = tb["alphas_cumprod"][t]
                alpha_next = tb["alphas_cumprod"][t_next]
                sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
                c = (1 - alpha_next - sigma ** 2).sqrt()
                r.ddim_sqrt_an, r.ddim_c, r.ddim_sigma, r.mode = float(alpha_next.sqrt()), float(c), float(sigma), 1
            else:
                r.ddim_sqrt_an, r.ddim_c, r.ddim_sigma, r.mode = 0.0, 0.0, 0.0, 0
            recs.append(r)
        return recs

    @staticmethod
    def upload_records(recs: List[L.MfSchedStep], device) -> torch.Tensor:
        arr = (L.MfSchedStep * len(recs))(*recs)
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
        return raw.to(device)

    # ------------------------------------------------------------------ tensor-level API (row S2), shared t per call
    def _uniform_t(self, t) -> int:
        tv = t.detach().cpu().reshape(-1)
        if not bool((tv == tv[0]).all()):
            raise NotImplementedError("HIP scheduler methods take one timestep for the whole batch (the sampling loop's t.expand(B), Q7)")
        return int(tv[0])

    def _one_step(self, x_t, pred, t, objective, clip_x0, noise_post=None):
        rec = self.step_records([self._uniform_t(t)], use_ddim=False)[0]
        table = self.upload_records([rec], x_t.device)
        x_t = x_t.contiguous()
        pred = pred.contiguous()
        out, x0, xT = torch.empty_like(x_t), torch.empty_like(x_t), torch.empty_like(x_t)
        a = L.MfSchedArgs(x_t.data_ptr(), pred.data_ptr(), None, None, None if noise_post is None else noise_post.data_ptr(), None, 0,
                          out.data_ptr(), x0.data_ptr(), xT.data_ptr(), table.data_ptr(), None, 0, objective, int(bool(clip_x0)), 1.0, x_t.numel())
        K.sched_step(a)
        return out, x0, xT

    def estimate_x_0(self, x_t, x_T, t, clip_x0=True):
        """gaussian_scheduler.py:119-124"""
        return self._one_step(x_t, x_T, t, 0, clip_x0)[1]

    def estimate_x_T(self, x_t, x_0, t, clip_x0=True):
        """gaussian_scheduler.py:127-131"""
        return self._one_step(x_t, x_0, t, 1, clip_x0)[2]

    def estimate_x_t_prior_from_x_T(self, x_t, t, x_T, use_log=True, clip_x0=True, var_scale=0, cold_diffusion=False, noise=None):
        """gaussian_scheduler.py:80-82 (+ :85-101).  `noise`: the posterior draw (tensor); N(0,1) Philox if None."""
        return self._prior(x_t, t, x_T, 0, use_log, clip_x0, var_scale, cold_diffusion, noise)

    def estimate_x_t_prior_from_x_0(self, x_t, t, x_0, use_log=True, clip_x0=True, var_scale=0, cold_diffusion=False, noise=None):
        return self._prior(x_t, t, x_0, 1, use_log, clip_x0, var_scale, cold_diffusion, noise)

    def _prior(self, x_t, t, pred, objective, use_log, clip_x0, var_scale, cold_diffusion, noise):
        if cold_diffusion or not use_log or not (isinstance(var_scale, (int, float)) and var_scale == 0):
            raise NotImplementedError("HIP scheduler: cold_diffusion / use_log=False / tensor var_scale are off the sampling path")
        if noise is None:
            from .noise import default_noise
            src = default_noise()
            src.begin(x_t.shape[0], x_t.device)
            noise = src.draw(tuple(x_t.shape))
        out, x0, _ = self._one_step(x_t, pred, t, objective, clip_x0, noise_post=noise.contiguous())
        return out, x0

    @classmethod
    def x_final(cls, x):
        """gaussian_scheduler.py:134-136 -- N(0,1) of x's shape from the default device noise source."""
        from .noise import default_noise
        src = default_noise()
        src.begin(x.shape[0], x.device)
        return src.draw(tuple(x.shape))

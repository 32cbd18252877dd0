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

    # ------------------------------------------------------------------ tensor-level API, t PER ROW
    # Coefficient rows are gathered on the host like `extract()` (scheduler_base.py:43-46) and applied by
    # mf_rows_axpby_f32; every method is bit-identical to the reference's elementwise chain on the same inputs.
    def _rows(self, name: str, t: torch.Tensor, device, negate: bool = False) -> torch.Tensor:
        v = self.host_tables()[name].gather(0, t.detach().to("cpu", torch.long).reshape(-1))
        return (-v if negate else v).to(device)

    def _uniform_t(self, t) -> int:
        tv = t.detach().cpu().reshape(-1)
        if not bool((tv == tv[0]).all()):
            raise NotImplementedError("this fused path takes one timestep for the whole batch (the sampling loop's t.expand(B), Q7)")
        return int(tv[0])

    @classmethod
    def _clip_x_0(cls, x_0):
        """gaussian_scheduler.py:138-151: static thresholding to [-1, 1]"""
        return K.rows_axpby(x_0, clamp=(-1.0, 1.0))

    def estimate_x_t(self, x_0, t, x_T=None, noise: Optional[torch.Tensor] = None):
        """gaussian_scheduler.py:61-77: rows with t<0 return x_0, t>=T return x_T, else sqrt(ac)*x_0 + sqrt(1-ac)*x_T."""
        if x_T is None:
            x_T = noise if noise is not None else self.x_final(x_0)
        tc = t.detach().to("cpu", torch.long).reshape(-1)
        tb = self.host_tables()
        idx = tc.clamp(0, self.T - 1)
        a = torch.where(tc < 0, torch.ones(()), torch.where(tc >= self.T, torch.zeros(()), tb["sqrt_alphas_cumprod"].gather(0, idx)))
        c = torch.where(tc < 0, torch.zeros(()), torch.where(tc >= self.T, torch.ones(()), tb["sqrt_one_minus_alphas_cumprod"].gather(0, idx)))
        return K.rows_axpby(x_0, a.to(x_0.device), x_T, c.to(x_0.device))

    def estimate_x_0(self, x_t, x_T, t, clip_x0=True):
        """gaussian_scheduler.py:119-124"""
        dev = x_t.device
        return K.rows_axpby(x_t, self._rows("sqrt_recip_alphas_cumprod", t, dev), x_T, self._rows("sqrt_recipm1_alphas_cumprod", t, dev, negate=True),
                            clamp=(-1.0, 1.0) if clip_x0 else None)

    def estimate_x_T(self, x_t, x_0, t, clip_x0=True):
        """gaussian_scheduler.py:127-131"""
        dev = x_t.device
        x_0 = self._clip_x_0(x_0) if clip_x0 else x_0
        minus_one = torch.full((x_t.shape[0],), -1.0, device=dev)
        return K.rows_axpby(x_t, self._rows("sqrt_recip_alphas_cumprod", t, dev), x_0, minus_one, self._rows("sqrt_recipm1_alphas_cumprod", t, dev))

    def estimate_mean_t(self, x_t, x_0, t):
        """gaussian_scheduler.py:104-107"""
        dev = x_t.device
        return K.rows_axpby(x_0, self._rows("posterior_mean_coef1", t, dev), x_t, self._rows("posterior_mean_coef2", t, dev))

    def estimate_variance_t(self, t, ndim, log=True, var_scale=0, eps=1e-20):
        """gaussian_scheduler.py:110-116 (scalar var_scale): [B,1,...] tensor on t's device, evaluated with the reference's ops."""
        tb = self.host_tables()
        tc = t.detach().to("cpu", torch.long).reshape(-1)
        mn, mx = self.extract(tb["posterior_variance"], tc, ndim), self.extract(tb["betas"], tc, ndim)
        if log:
            mn, mx = torch.log(mn.clamp(min=eps)), torch.log(mx.clamp(min=eps))
        return (var_scale * mx + (1 - var_scale) * mn).to(t.device)

    def estimate_x_t_prior_from_x_T(self, x_t, t, x_T, use_log=True, clip_x0=True, var_scale=0, cold_diffusion=False, noise=None):
        """gaussian_scheduler.py:80-82"""
        x_0 = self.estimate_x_0(x_t, x_T, t, clip_x0)
        return self.estimate_x_t_prior_from_x_0(x_t, t, x_0, use_log, clip_x0, var_scale, cold_diffusion, noise)

    def estimate_x_t_prior_from_x_0(self, x_t, t, x_0, use_log=True, clip_x0=True, var_scale=0, cold_diffusion=False, noise=None):
        """gaussian_scheduler.py:85-101.  `noise`: the posterior draw (tensor); N(0,1) from the default device source if None."""
        x_0 = self._clip_x_0(x_0) if clip_x0 else x_0
        if cold_diffusion:  # https://arxiv.org/abs/2208.09392, :88-93
            x_T_est = self.estimate_x_T(x_t, x_0, t)
            x_t_est = self.estimate_x_t(x_0, t, x_T=x_T_est)
            x_t_prior = self.estimate_x_t(x_0, t - 1, x_T=x_T_est)
            m1 = torch.full((x_t.shape[0],), -1.0, device=x_t.device)
            noise_t = K.rows_axpby(x_t_est, None, x_t_prior, m1)        # x_t_est - x_t_prior
            return K.rows_axpby(x_t, None, noise_t, m1), x_0             # x_t - noise_t
        if torch.is_tensor(var_scale):
            raise NotImplementedError("tensor var_scale (learned variance) runs through the fused step kernel of DiffusionPipeline")
        mean = self.estimate_mean_t(x_t, x_0, t)
        tc = t.detach().to("cpu", torch.long).reshape(-1)
        variance = self.estimate_variance_t(tc, 1, use_log, var_scale)
        std = torch.exp(0.5 * variance) if use_log else torch.sqrt(variance)
        std[tc == 0] = 0.0
        if noise is None:
            noise = self.x_final(x_t)
        return K.rows_axpby(mean, None, noise, std.to(x_t.device)), x_0

    def sample(self, x_0):
        """scheduler_base.py:20-24 (the training-side entry of the forward process; kept for API completeness): one random t in [0, T) per
        row (torch's generator of x_0's device, like the reference), x_T = x_final(x_0), returns (x_t, x_T, t)."""
        t = torch.randint(0, self.T, (x_0.shape[0],), dtype=torch.long, device=x_0.device)
        x_T = self.x_final(x_0)
        return self.estimate_x_t(x_0, t, x_T), x_T, t

    @classmethod
    def x_final(cls, x):
        """gaussian_scheduler.py:134-136 -- N(0,1) of x's shape from the default device noise source."""
        from .noise import default_noise
        src = default_noise()
        src.begin(x.shape[0], x.device)
        return src.draw(tuple(x.shape))

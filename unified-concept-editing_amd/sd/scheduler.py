"""PNDM scheduler as SD-1.4 ships it (skip_prk_steps=True, steps_offset=1, scaled_linear betas
0.00085 -> 0.012 over 1000 steps, epsilon prediction, set_alpha_to_one=False): the scheduler
`pipe(...)` uses inside evalscripts/generate-images-sd.py:37-42.  Restated from the published
diffusers==0.33.0 schedulers/scheduling_pndm.py (not vendored in the reference, not installed
here: numerical parity with diffusers is unpinned).  50 inference steps = 51 U-Net calls.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch


class PNDMScheduler:
    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 steps_offset: int = 1):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]          # set_alpha_to_one=False
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        self.init_noise_sigma = 1.0
        self.timesteps: Optional[torch.Tensor] = None
        self.ets: List[torch.Tensor] = []
        self.counter = 0
        self.cur_sample = None
        self.step_ratio = 1

    def set_timesteps(self, num_inference_steps: int, device=None) -> None:
        self.step_ratio = self.num_train_timesteps // num_inference_steps
        base = (np.arange(0, num_inference_steps) * self.step_ratio).round() + self.steps_offset
        plms = np.concatenate([base[:-1], base[-2:-1], base[-1:]])[::-1].copy()   # second step repeated
        self.timesteps = torch.from_numpy(plms.astype(np.int64)).to(device)
        self.ets = []
        self.counter = 0
        self.cur_sample = None

    def _prev_sample(self, sample, t: int, t_prev: int, eps):
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[t_prev] if t_prev >= 0 else self.final_alpha_cumprod
        b_t, b_prev = 1 - a_t, 1 - a_prev
        sample_coeff = (a_prev / a_t) ** 0.5
        denom = a_t * b_prev ** 0.5 + (a_t * b_t * a_prev) ** 0.5
        return float(sample_coeff) * sample - float((a_prev - a_t) / denom) * eps

    def step(self, model_output: torch.Tensor, timestep: int, sample: torch.Tensor) -> torch.Tensor:
        """PLMS branch (skip_prk_steps=True)."""
        t = int(timestep)
        t_prev = t - self.step_ratio
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(model_output)
        else:
            t_prev = t
            t = t + self.step_ratio
        n = len(self.ets)
        if n == 1 and self.counter == 0:
            self.cur_sample = sample
        elif n == 1 and self.counter == 1:
            model_output = (model_output + self.ets[-1]) / 2
            sample = self.cur_sample
            self.cur_sample = None
        elif n == 2:
            model_output = (3 * self.ets[-1] - self.ets[-2]) / 2
        elif n == 3:
            model_output = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
        else:
            model_output = (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4]) / 24
        prev = self._prev_sample(sample, t, t_prev, model_output)
        self.counter += 1
        return prev

    def step_fused(self, eps_raw: torch.Tensor, cfg: bool, guidance_scale: float, timestep: int, sample: torch.Tensor,
                   handle) -> torch.Tensor:
        """The same PLMS step with the guidance combine in front of it as ONE HIP launch (uce_cfg_pndm_step): `eps_raw`
        is the U-Net output for [uncond; cond] (cfg) or the plain output.  Branch selection as in `step`; the linear
        combination of the stored outputs and the two coefficients of `_prev_sample` go to the kernel as scalars."""
        t = int(timestep)
        t_prev = t - self.step_ratio
        second = self.counter == 1                       # the repeated timestep: uses the first output, appends nothing
        if second:
            t_prev, t = t, t + self.step_ratio
        hist = list(reversed(self.ets[-3:])) if not second else [self.ets[-1]]
        n_after = len(self.ets[-3:]) + 1 if not second else 1
        if second:
            w, src = (0.5, 0.5, 0.0, 0.0), self.cur_sample
        elif n_after == 1:
            w, src = (1.0, 0.0, 0.0, 0.0), sample
        elif n_after == 2:
            w, src = (1.5, -0.5, 0.0, 0.0), sample
        elif n_after == 3:
            w, src = (23 / 12, -16 / 12, 5 / 12, 0.0), sample
        else:
            w, src = (55 / 24, -59 / 24, 37 / 24, -9 / 24), sample
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[t_prev] if t_prev >= 0 else self.final_alpha_cumprod
        b_t, b_prev = 1 - a_t, 1 - a_prev
        cs = float((a_prev / a_t) ** 0.5)
        ce = float((a_prev - a_t) / (a_t * b_prev ** 0.5 + (a_t * b_t * a_prev) ** 0.5))
        eps, prev = handle.cfg_pndm_step(eps_raw.contiguous(), cfg, guidance_scale, hist, w, src.contiguous(), cs, ce)
        if not second:
            self.ets = self.ets[-3:]
            self.ets.append(eps)
            if n_after == 1 and self.counter == 0:
                self.cur_sample = sample
        else:
            self.cur_sample = None
        self.counter += 1
        return prev


class EulerDiscreteScheduler:
    """Euler (ancestral-free) discrete scheduler as SDXL-base ships it (scheduler/scheduler_config.json: scaled_linear betas
    0.00085 -> 0.012 over 1000 steps, epsilon prediction, timestep_spacing "leading", steps_offset 1, linear sigma
    interpolation, no Karras sigmas, s_churn 0): what `pipe(...)` runs inside uce_sd_debias.py:22-26 when the model is
    SDXL.  Restated from the published diffusers==0.33.0 schedulers/scheduling_euler_discrete.py (not installed here)."""

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 steps_offset: int = 1):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        ac = torch.cumprod(1.0 - betas, dim=0).numpy().astype(np.float64)
        self.train_sigmas = ((1 - ac) / ac) ** 0.5
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        self.timesteps: Optional[torch.Tensor] = None
        self.sigmas: Optional[np.ndarray] = None
        self.init_noise_sigma = float((self.train_sigmas.max() ** 2 + 1) ** 0.5)
        self._index = 0

    def set_timesteps(self, num_inference_steps: int, device=None) -> None:
        step_ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.float32) + self.steps_offset
        sig = np.interp(ts, np.arange(0, len(self.train_sigmas)), self.train_sigmas)
        self.sigmas = np.concatenate([sig, [0.0]]).astype(np.float32)
        self.timesteps = torch.from_numpy(ts).to(device)
        self.init_noise_sigma = float((self.sigmas.max() ** 2 + 1) ** 0.5)        # "leading" spacing
        self._index = 0

    def scale_model_input(self, sample: torch.Tensor, timestep=None) -> torch.Tensor:
        sigma = float(self.sigmas[self._index])
        return sample / ((sigma ** 2 + 1) ** 0.5)

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor) -> torch.Tensor:
        """x_{i+1} = x_i + eps * (sigma_{i+1} - sigma_i)  (epsilon prediction: the derivative IS the model output);
        computed in fp32 and cast back, as diffusers does."""
        sigma, sigma_next = float(self.sigmas[self._index]), float(self.sigmas[self._index + 1])
        prev = sample.float() + model_output.float() * (sigma_next - sigma)
        self._index += 1
        return prev.to(model_output.dtype)

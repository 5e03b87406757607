"""Host-side diffusion pieces of the OFT diffusion head (tiny tensors: a chunk of actions is 8x7 / 25x14 numbers).

  DDIMScheduler                  diffusers' DDIMScheduler(num_train_timesteps=T, beta_schedule="squaredcos_cap_v2") as the
                                 reference constructs it (prismatic/models/action_heads.py:163) with the library defaults
                                 (clip_sample=True, prediction_type="epsilon", eta=0, timestep_spacing="leading",
                                 set_alpha_to_one=True).  diffusers is not installed here: restated from its published
                                 algorithm, PARITY UNPINNED against the library.
  SinusoidalPositionalEncoding   prismatic/models/action_heads.py:12-35
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
import torch


class DDIMScheduler:
    def __init__(self, num_train_timesteps: int = 100, beta_schedule: str = "squaredcos_cap_v2"):
        if beta_schedule != "squaredcos_cap_v2":
            raise NotImplementedError(beta_schedule)
        T = num_train_timesteps

        def alpha_bar(t):
            return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2

        betas = [min(1 - alpha_bar((i + 1) / T) / alpha_bar(i / T), 0.999) for i in range(T)]
        self.betas = torch.tensor(betas, dtype=torch.float32)
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0)
        self.config = SimpleNamespace(num_train_timesteps=T)
        self.num_inference_steps = None
        self.timesteps = torch.arange(T - 1, -1, -1)

    def set_timesteps(self, num_inference_steps: int):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        self.timesteps = torch.from_numpy((np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64))

    def add_noise(self, original_samples, noise, timesteps):
        ac = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        a = ac[timesteps] ** 0.5
        s = (1 - ac[timesteps]) ** 0.5
        while a.dim() < original_samples.dim():
            a, s = a.unsqueeze(-1), s.unsqueeze(-1)
        return a * original_samples + s * noise

    def step(self, model_output, timestep: int, sample):
        t = int(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        x0 = ((sample - (1 - a_t) ** 0.5 * model_output) / a_t ** 0.5).clamp(-1.0, 1.0)
        eps = (sample - a_t ** 0.5 * x0) / (1 - a_t) ** 0.5
        return SimpleNamespace(prev_sample=a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps)


class SinusoidalPositionalEncoding:
    def __init__(self, dim: int):
        self.dim = dim

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        assert self.dim % 2 == 0, f"# dimensions must be even but got {self.dim}"
        half = self.dim // 2
        exponent = torch.arange(half, device=x.device) * -math.log(10000) / (half - 1)
        emb = x[:, None] * torch.exp(exponent)[None, :]
        return torch.cat((emb.sin(), emb.cos()), dim=-1)

    forward = __call__

"""ORACLE (test infrastructure) -- DPM-Solver++(2M) sampler restatement.

Follows, for the one configuration the reference ships
(cosine betas, v_prediction, dpmsolver++, order 2, midpoint, linspace
timesteps, final sigma = 0; modeling_vibevoice.py:138-142 and
configs/qwen2.5_1.5b_64k.json:66-77):

  betas_for_alpha_bar            vibevoice/schedule/dpm_solver.py:28-83
  __init__ (alphas_cumprod)      dpm_solver.py:203-295
  set_timesteps                  dpm_solver.py:321-423
  _sigma_to_alpha_sigma_t        dpm_solver.py:483-487
  convert_model_output (v-pred)  dpm_solver.py:581-584
  first-order update             dpm_solver.py:669-677
  second-order midpoint update   dpm_solver.py:738-764
  step() order selection         dpm_solver.py:974-1006
  sample_speech_tokens (CFG)     modeling_vibevoice_inference.py:697-710

All scalar coefficient math is done with 0-dim fp32 torch tensors exactly
as the reference does, so the results agree to the last bit on CPU.
"""
import math

import numpy as np
import torch


def cosine_alphas_cumprod(num_train_timesteps=1000, max_beta=0.999):
    """dpm_solver.py:52-56,79-83,250-251."""
    def alpha_bar_fn(t):
        return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    betas = []
    for i in range(num_train_timesteps):
        t1 = i / num_train_timesteps
        t2 = (i + 1) / num_train_timesteps
        betas.append(min(1 - alpha_bar_fn(t2) / alpha_bar_fn(t1), max_beta))
    betas = torch.tensor(betas, dtype=torch.float32)
    alphas = 1.0 - betas
    return torch.cumprod(alphas, dim=0)


class Schedule:
    """Timesteps + sigmas for N inference steps (dpm_solver.py:321-423)."""

    def __init__(self, num_inference_steps, num_train_timesteps=1000):
        ac = cosine_alphas_cumprod(num_train_timesteps)
        self.alphas_cumprod = ac
        alpha_t = torch.sqrt(ac)
        sigma_t = torch.sqrt(1 - ac)
        lambda_t = torch.log(alpha_t) - torch.log(sigma_t)
        # lambda_min_clipped = -inf  ->  clipped_idx = 0 (dpm_solver.py:353-354)
        clipped_idx = torch.searchsorted(torch.flip(lambda_t, [0]), -float("inf"))
        last_timestep = int((num_train_timesteps - clipped_idx).item())
        timesteps = (
            np.linspace(0, last_timestep - 1, num_inference_steps + 1)
            .round()[::-1][:-1].copy().astype(np.int64))
        sigmas = (((1 - ac) / ac) ** 0.5).cpu().numpy()          # .cpu(): a no-op here; bench.py's eager-GPU leg runs under a cuda default device
        sigmas = np.interp(timesteps, np.arange(0, len(sigmas)), sigmas)
        sigmas = np.concatenate([sigmas, [0]]).astype(np.float32)
        self.timesteps = torch.from_numpy(timesteps)
        self.sigmas = torch.from_numpy(sigmas)
        self.n = len(timesteps)


def _alpha_sigma(sigma):
    alpha_t = 1 / ((sigma ** 2 + 1) ** 0.5)
    sigma_t = sigma * alpha_t
    return alpha_t, sigma_t


class DPMState:
    """The mutable part of the scheduler (model_outputs history, counters)."""

    def __init__(self, sched: Schedule, algorithm_type: str = "dpmsolver++"):
        # "dpmsolver++" (the configuration the model classes build, modeling_vibevoice.py:138-142) or "sde-dpmsolver++" (what
        # demo/gradio_demo.py:142-146 swaps in through noise_scheduler.from_config)
        assert algorithm_type in ("dpmsolver++", "sde-dpmsolver++"), algorithm_type
        self.s = sched
        self.sde = algorithm_type == "sde-dpmsolver++"
        self.step_index = 0
        self.lower_order_nums = 0
        self.model_outputs = [None, None]

    def step(self, model_output, sample, noise=None):
        """One scheduler.step() (dpm_solver.py:935-1022) for solver_order=2.  `noise`: the fp32 variance noise of the stochastic
        solver (drawn by the reference inside step(), :994-997, shape of model_output)."""
        s = self.s
        i = self.step_index
        lower_order_final = (i == s.n - 1)          # final_sigmas_type == "zero"
        # convert_model_output, v_prediction (dpm_solver.py:581-584); done in
        # the dtype of `sample`/`model_output` BEFORE the fp32 upcast (:987 vs :993)
        sigma = s.sigmas[i]
        a, sg = _alpha_sigma(sigma)
        x0 = a * sample - sg * model_output
        self.model_outputs[0] = self.model_outputs[1]
        self.model_outputs[1] = x0
        out_dtype = x0.dtype
        sample = sample.to(torch.float32)
        if self.lower_order_nums < 1 or lower_order_final:
            sigma_t, sigma_s = s.sigmas[i + 1], s.sigmas[i]
            alpha_t, sigma_t = _alpha_sigma(sigma_t)
            alpha_s, sigma_s = _alpha_sigma(sigma_s)
            lambda_t = torch.log(alpha_t) - torch.log(sigma_t)
            lambda_s = torch.log(alpha_s) - torch.log(sigma_s)
            h = lambda_t - lambda_s
            if self.sde:        # dpm_solver.py:680-686
                prev = ((sigma_t / sigma_s * torch.exp(-h)) * sample + (alpha_t * (1 - torch.exp(-2.0 * h))) * x0
                        + sigma_t * torch.sqrt(1.0 - torch.exp(-2 * h)) * noise.to(torch.float32))
            else:
                prev = (sigma_t / sigma_s) * sample - (alpha_t * (torch.exp(-h) - 1.0)) * x0
        else:
            sigma_t, sigma_s0, sigma_s1 = s.sigmas[i + 1], s.sigmas[i], s.sigmas[i - 1]
            alpha_t, sigma_t = _alpha_sigma(sigma_t)
            alpha_s0, sigma_s0 = _alpha_sigma(sigma_s0)
            alpha_s1, sigma_s1 = _alpha_sigma(sigma_s1)
            lambda_t = torch.log(alpha_t) - torch.log(sigma_t)
            lambda_s0 = torch.log(alpha_s0) - torch.log(sigma_s0)
            lambda_s1 = torch.log(alpha_s1) - torch.log(sigma_s1)
            m0, m1 = self.model_outputs[-1], self.model_outputs[-2]
            h, h_0 = lambda_t - lambda_s0, lambda_s0 - lambda_s1
            r0 = h_0 / h
            D0, D1 = m0, (1.0 / r0) * (m0 - m1)
            if self.sde:        # dpm_solver.py:785-793 (midpoint)
                prev = ((sigma_t / sigma_s0 * torch.exp(-h)) * sample
                        + (alpha_t * (1 - torch.exp(-2.0 * h))) * D0
                        + 0.5 * (alpha_t * (1 - torch.exp(-2.0 * h))) * D1
                        + sigma_t * torch.sqrt(1.0 - torch.exp(-2 * h)) * noise.to(torch.float32))
            else:
                prev = ((sigma_t / sigma_s0) * sample
                        - (alpha_t * (torch.exp(-h) - 1.0)) * D0
                        - 0.5 * (alpha_t * (torch.exp(-h) - 1.0)) * D1)
        if self.lower_order_nums < 2:
            self.lower_order_nums += 1
        self.step_index += 1
        return prev.to(out_dtype)


def sample_speech_tokens(head_fn, condition, neg_condition, cfg_scale, num_steps,
                         noise, t_cast_dtype=None, algorithm_type="dpmsolver++", step_noise=None):
    """modeling_vibevoice_inference.py:697-710.

    head_fn(noisy[2n,64], t[2n], cond[2n,H]) -> [2n,64]
    noise: the pre-drawn [2n, 64] tensor the reference gets from
           torch.randn(...) at :701 (injected so both sides share the RNG).
    t_cast_dtype: the reference casts the integer timestep to the model dtype
           (`t.repeat(..).to(combined)`, :705); pass torch.bfloat16 to
           reproduce the bf16 GPU path's 999->1000 rounding, None for fp32.
    """
    # step_noise (stochastic solver only): [num_steps, 2n, 64] -- the draws scheduler.step() makes, one per solver step, with
    # the shape of the model output (both CFG halves; only the first half survives the next step's `speech[:n]`)
    sched = Schedule(num_steps)
    st = DPMState(sched, algorithm_type)
    condition = torch.cat([condition, neg_condition], dim=0)
    speech = noise.to(condition)
    n2 = speech.shape[0]
    for i, t in enumerate(sched.timesteps):
        half = speech[: n2 // 2]
        combined = torch.cat([half, half], dim=0)
        tt = t.repeat(n2).to(combined)
        if t_cast_dtype is not None:
            tt = tt.to(t_cast_dtype).to(combined.dtype)
        eps = head_fn(combined, tt, condition)
        cond_eps, uncond_eps = torch.split(eps, n2 // 2, dim=0)
        half_eps = uncond_eps + cfg_scale * (cond_eps - uncond_eps)
        eps = torch.cat([half_eps, half_eps], dim=0)
        speech = st.step(eps, speech, None if step_noise is None else step_noise[i])
    return speech[: n2 // 2]

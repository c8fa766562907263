"""Host-side DPM-Solver++(2M) coefficient table for the HIP sampler.

The reference recomputes these scalars on the CPU for every generated frame
(`noise_scheduler.set_timesteps(N)` + ~30 0-dim tensor ops per solver step,
vibevoice/schedule/dpm_solver.py:321-423, 669-677, 738-764).  They depend only
on N, so they are computed once here -- in fp32 with the same formulas and
operation order -- and handed to `vv_set_schedule`; the device kernel
(vv_cfg_dpm_kernel) then applies

    x0  = a_i * x - s_i * v                       (v-prediction, :581-584)
    x'  = cs_i * x + c0_i * x0 + c1_i * (x0 - x0_prev)

Configuration: the one the reference ships (cosine betas, v_prediction,
dpmsolver++, solver_order 2, midpoint, linspace spacing, final sigma 0;
modeling_vibevoice.py:138-142, configs/*.json:66-77).
"""
import math

import numpy as np
import torch


def _alphas_cumprod(num_train=1000, max_beta=0.999):
    # betas_for_alpha_bar(cosine)  (dpm_solver.py:52-56, 79-83)
    f = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    betas = [min(1 - f((i + 1) / num_train) / f(i / num_train), max_beta) for i in range(num_train)]
    return torch.cumprod(1.0 - torch.tensor(betas, dtype=torch.float32), dim=0)


def timesteps_and_sigmas(n_steps, num_train=1000):
    ac = _alphas_cumprod(num_train)
    ts = np.linspace(0, num_train - 1, n_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
    sig = (((1 - ac) / ac) ** 0.5).numpy()
    sig = np.interp(ts, np.arange(0, len(sig)), sig)
    sig = np.concatenate([sig, [0]]).astype(np.float32)
    return ts, torch.from_numpy(sig)


def _as(sigma):
    a = 1 / ((sigma ** 2 + 1) ** 0.5)
    return a, sigma * a


ALGORITHMS = ("dpmsolver++", "sde-dpmsolver++")


def make_table(n_steps, t_cast_bf16=False, algorithm_type="dpmsolver++"):
    """-> (t_values float32[n], coef float32[n,5]) for the deterministic solver the model classes build, or coef float32[n,6]
    = {a, s, cs, c0, c1, cn} for "sde-dpmsolver++" (what demo/gradio_demo.py:142-146 swaps in):
        x' = cs x + c0 x0 + c1 (x0 - x0_prev) + cn eps_i,  eps_i ~ N(0, 1) drawn once per solver step
    with cs = (sigma_t / sigma_s) e^{-h}, c0 = alpha_t (1 - e^{-2h}), c1 = c0 / (2 r0), cn = sigma_t sqrt(1 - e^{-2h})
    (dpm_solver.py:680-686, 785-793)."""
    if algorithm_type not in ALGORITHMS:
        raise NotImplementedError(f"noise scheduler algorithm_type={algorithm_type!r}: the HIP sampler implements {ALGORITHMS}")
    sde = algorithm_type == "sde-dpmsolver++"
    ts, sig = timesteps_and_sigmas(n_steps)
    coef = np.zeros((n_steps, 6 if sde else 5), dtype=np.float32)
    for i in range(n_steps):
        a_i, s_i = _as(sig[i])
        a_t, s_t = _as(sig[i + 1])
        lam_t = torch.log(a_t) - torch.log(s_t)
        lam_s = torch.log(a_i) - torch.log(s_i)
        h = lam_t - lam_s
        if sde:
            c0 = a_t * (1 - torch.exp(-2.0 * h))
            cs = s_t / s_i * torch.exp(-h)
            cn = s_t * torch.sqrt(1.0 - torch.exp(-2 * h))
        else:
            c0 = -(a_t * (torch.exp(-h) - 1.0))
            cs = s_t / s_i
        c1 = torch.zeros(())
        first_order = (i == 0) or (i == n_steps - 1)      # lower_order_nums<1 / final sigma == 0
        if not first_order:
            a_p, s_p = _as(sig[i - 1])
            lam_p = torch.log(a_p) - torch.log(s_p)
            r0 = (lam_s - lam_p) / h
            c1 = 0.5 * c0 * (1.0 / r0)
        coef[i, :5] = [float(a_i), float(s_i), float(cs), float(c0), float(c1)]
        if sde:
            coef[i, 5] = float(cn)
    tv = torch.from_numpy(ts).to(torch.float32)
    if t_cast_bf16:
        # the reference feeds `t.repeat(..).to(combined)` to the head
        # (modeling_vibevoice_inference.py:705): under bf16 weights the integer
        # timestep is rounded to bf16 (999 -> 1000, 949 -> 948, ...)
        tv = tv.to(torch.bfloat16).to(torch.float32)
    return tv.numpy().astype(np.float32), coef

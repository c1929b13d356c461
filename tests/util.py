"""Shared helpers for the parity tests (oracle = checker, parrot_amd = product)."""
import numpy as np
import torch


def rel_err(a, b):
    """max |a-b| / max |b| (norm-wise relative error, b = oracle)."""
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    denom = float(b.abs().max())
    if denom == 0.0:
        return float((a - b).abs().max())
    return float((a - b).abs().max()) / denom


def rel_err_elem(a, b, floor_frac=1e-3):
    """Element-wise relative error max_i |a_i - b_i| / max(|b_i|, floor_frac * max|b|): every element is held to its
    own magnitude, down to a floor of floor_frac of the largest one (below that an fp32 result has no relative
    accuracy left to speak of).  Stricter than rel_err by up to 1 / floor_frac."""
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    top = float(b.abs().max())
    if top == 0.0:
        return float((a - b).abs().max())
    return float(((a - b).abs() / b.abs().clamp_min(floor_frac * top)).max())


def assert_close(a, b, tol, what=""):
    e = rel_err(a, b)
    assert e <= tol, f"{what}: relative error {e:.3e} (element-wise {rel_err_elem(a, b):.3e}) > {tol:.1e}"
    return e


def make_batch(cfg, T, B, U, seed=0, ragged=False, dtype=torch.float64, speaker=False):
    g = torch.Generator().manual_seed(seed)
    O = cfg['output_dim']
    feat = torch.randn(T + 1, B, O, generator=g, dtype=torch.float64).to(dtype)
    fmask = torch.ones(T + 1, B, dtype=dtype)
    labels = torch.randint(0, cfg['num_characters'], (B, U), generator=g)
    lmask = torch.ones(B, U, dtype=dtype)
    if ragged:
        for b in range(B):
            tl = int(torch.randint(max(2, T // 2), T + 2, (1,), generator=g))
            ul = int(torch.randint(max(2, U // 2), U + 1, (1,), generator=g))
            fmask[tl:, b] = 0
            lmask[b, ul:] = 0
    spk = torch.randint(0, cfg['num_speakers'], (B, 1), generator=g) if speaker else None
    return feat, fmask, labels, lmask, spk

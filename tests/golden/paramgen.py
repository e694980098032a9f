"""Deterministic, reference-free parameter and input generator for fixtures.

Golden outputs are produced by loading THESE tensors into the reference model
(make_golden.py, run once in the authoring container) and are re-created at
test time from the same (key, seed) pairs, so the 11 MB state dict never has
to be committed.  numpy's PCG64 stream is stable across platforms/versions.

Norm/PReLU/bias values are pushed away from their defaults (1 / 0 / 0.25): the
defaults hide channel-indexing bugs.
"""
from __future__ import annotations

import zlib
from typing import Dict, Mapping, Tuple

import numpy as np


def _rng(seed: int, key: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(key.encode())])


def make_param(key: str, shape: Tuple[int, ...], kind: str, fan_in: int, seed: int, gain: float = 1.7) -> np.ndarray:
    g = _rng(seed, key)
    if kind in ("conv_w", "convT_w", "lin_w", "lstm"):
        a = gain / np.sqrt(max(fan_in, 1))       # 1.7: a little hotter than PyTorch's default (gain 1)
        return g.uniform(-a, a, size=shape).astype(np.float32)
    if kind == "bias":
        return g.uniform(-0.2, 0.2, size=shape).astype(np.float32)
    if kind in ("norm_w", "ln_w"):
        return g.uniform(0.5, 1.5, size=shape).astype(np.float32)
    if kind in ("norm_b", "ln_b"):
        return g.uniform(-0.3, 0.3, size=shape).astype(np.float32)
    if kind == "prelu":
        return g.uniform(0.05, 0.45, size=shape).astype(np.float32)
    if kind == "bn_mean":
        return g.uniform(-0.2, 0.2, size=shape).astype(np.float32)
    if kind == "bn_var":
        return g.uniform(0.3, 1.2, size=shape).astype(np.float32)
    if kind == "bn_count":
        return np.asarray(7, dtype=np.int64)
    raise ValueError(kind)


def make_params(specs: Mapping[str, object], seed: int, gain: float = 1.7) -> Dict[str, np.ndarray]:
    """specs: key -> object with .shape/.kind/.fan_in (eabnet_amd.spec.ParamSpec)."""
    return {k: make_param(k, tuple(s.shape), s.kind, s.fan_in, seed, gain) for k, s in specs.items()}


def make_spec_input(B: int, T: int, F: int, M: int, seed: int, scale: float = 0.3) -> np.ndarray:
    """A (B,T,F,M,2) 'compressed spectrogram' input."""
    g = _rng(seed, f"spec_input/{B}/{T}/{F}/{M}")
    return (scale * g.standard_normal((B, T, F, M, 2))).astype(np.float32)


def make_wave(B: int, M: int, L: int, seed: int, scale: float = 0.05) -> np.ndarray:
    """(B,M,L) noisy waves: a shared source delayed per mic plus white noise
    (SURVEY §8d synthetic input)."""
    g = _rng(seed, f"wave/{B}/{M}/{L}")
    src = 0.1 * g.standard_normal((B, 1, L + 16))
    # crude low-pass so that the spectrum is not flat
    src = (src + np.roll(src, 1, -1) + np.roll(src, 2, -1)) / 3.0
    x = scale * g.standard_normal((B, M, L))
    for m in range(M):
        d = m % 9
        x[:, m, :] += src[:, 0, d:d + L]
    return x.astype(np.float32)

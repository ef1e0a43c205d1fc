"""The fast GELU of the bf16 GEGLU epilogues (`fmc_gelu_fast`, synfmc_amd/csrc/common.h) against the exact erf form the reference computes
(diffusers GEGLU.gelu -> torch.nn.functional.gelu, attention.py of diffusers 0.24): the constants are read out of the header, the formula is evaluated
in float32 exactly as the kernel does (fminf, two fmaf, exp2, reciprocal), and the absolute error bound quoted in the header is asserted."""
import os
import re

import numpy as np
import torch

HDR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "synfmc_amd", "csrc", "common.h")


def _constants():
    src = open(HDR).read()
    body = src[src.index("float fmc_gelu_fast(float g)"):]
    body = body[:body.index("}")]
    clamp = float(re.search(r"fminf\(g \* g, ([0-9.eE+-]+)f\)", body).group(1))
    c2, c1, c0 = (float(x) for x in re.search(r"fmaf\(fmaf\(([0-9.eE+-]+)f, g2, ([0-9.eE+-]+)f\), g2, ([0-9.eE+-]+)f\)", body).groups())
    return clamp, c0, c1, c2


def _fast(g: np.ndarray) -> np.ndarray:
    clamp, c0, c1, c2 = (np.float32(v) for v in _constants())
    g = g.astype(np.float32)
    with np.errstate(over="ignore"):
        g2 = np.minimum(g * g, clamp)
    s = (c2 * g2 + c1) * g2 + c0
    with np.errstate(over="ignore", invalid="ignore"):
        return g * (np.float32(1) / (np.float32(1) + np.exp2(g * s)))


def test_fast_gelu_error_bound():
    g = np.linspace(-12.0, 12.0, 480001)
    ref = torch.nn.functional.gelu(torch.from_numpy(g)).numpy()           # float64 erf form
    err = np.abs(_fast(g).astype(np.float64) - ref)
    assert err.max() <= 2.6e-5, err.max()
    # relative to a bf16 result: below half an ulp (2^-9) wherever |gelu| >= 0.014; the tails are exact
    big = np.abs(ref) >= 0.014
    assert (err[big] / np.abs(ref[big])).max() < 2.0 ** -9


def test_fast_gelu_tails_and_specials():
    g = np.array([-1e30, -1e4, -40.0, -9.5, 0.0, 9.5, 40.0, 1e4, 1e30, np.inf, -np.inf], dtype=np.float32)
    out = _fast(g)
    assert np.all(np.abs(out[:4]) < 1e-10) and out[4] == 0          # true value at -9.5: -1e-20
    assert np.array_equal(out[5:10], g[5:10])
    assert out[10] == 0 or np.isnan(out[10])          # -inf * 0: the reference gives -0; no finite activation reaches it
    assert np.isnan(_fast(np.array([np.nan]))[0])

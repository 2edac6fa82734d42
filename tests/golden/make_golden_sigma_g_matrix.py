"""Generates tests/golden/sigma_g_matrix.npz in THIS container BY RUNNING THE REFERENCE'S OWN FUNCTION:
`SigmaGClipping.compute_clipped_sigma_g_matrix` of /root/reference/src/kbmod/filters/sigma_g_filter.py, loaded from where it
lies (round 5).  The module's one import that is not available here, `from kbmod.search import DebugTimer` (the compiled
extension; the name is used by another method only), is satisfied by an empty placeholder module for the duration of the load --
no line of the function under test is replaced.  `valid_*` in the fixture are that function's outputs.

Next to them the script keeps its own spelling of the same call sequence on torch (sigma_g_filter.py:132-165:
torch.tensor(float32) -> optional torch.where(lh > 0, lh, nan) -> torch.nanquantile(q, dim=1) ->
delta floor 1e-5 -> bounds -> isfinite & < & >) for seeded inputs, together with the bounds so that
the consumers can exclude points within two ulps of a bound (torch's lerp kernels are fused
differently on different devices).
"""

import os

import numpy as np
import torch
from scipy.special import erfinv


def coeff(lo, hi):
    def inv(z):
        s = -1 if z < 0.5 else 1
        return float(s * np.sqrt(2) * erfinv(s * (2 * z - 1)))

    return 1 / (inv(hi / 100.0) - inv(lo / 100.0))


def reference_sequence(lh, lo, hi, n_sigma, clip_negative):
    torch_lh = torch.tensor(lh, device="cpu", dtype=torch.float32)
    masked = torch.where(torch_lh > 0.0, torch_lh, np.nan) if clip_negative else torch_lh
    q = torch.tensor([lo / 100.0, 0.5, hi / 100.0], dtype=torch.float32)
    lower_per, median, upper_per = torch.nanquantile(masked, q, dim=1)
    delta = upper_per - lower_per
    delta[delta < 1e-5] = 1e-5
    n_sigma_g = n_sigma * coeff(lo, hi) * delta
    lower = (median - n_sigma_g).reshape(-1, 1)
    upper = (median + n_sigma_g).reshape(-1, 1)
    valid = torch.isfinite(torch_lh) & (torch_lh < upper) & (torch_lh > lower)
    return valid.numpy().astype(bool), lower.numpy().ravel(), upper.numpy().ravel()


def load_reference_class():
    """SigmaGClipping of the reference, loaded from its source file."""
    import importlib.util
    import sys
    import types

    path = "/root/reference/src/kbmod/filters/sigma_g_filter.py"
    placeholder = types.ModuleType("kbmod.search")
    placeholder.DebugTimer = object          # (imported by name at module level, used by apply_clipped_sigma_g only)
    saved = {k: sys.modules.get(k) for k in ("kbmod", "kbmod.search")}
    sys.modules.setdefault("kbmod", types.ModuleType("kbmod"))
    sys.modules["kbmod.search"] = placeholder
    try:
        spec = importlib.util.spec_from_file_location("ref_sigma_g_filter", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod.SigmaGClipping


def main():
    SigmaGClipping = load_reference_class()
    rng = np.random.default_rng(20260929)
    out = {}
    cases = [(40, 20, 25, 75, 2.0, False), (40, 20, 25, 75, 2.0, True), (30, 57, 10, 90, 3.0, False),
             (25, 64, 25, 75, 2.0, True), (16, 7, 25, 75, 1.5, False), (12, 1, 25, 75, 2.0, False)]
    for i, (n, t, lo, hi, ns, clip) in enumerate(cases):
        lh = (10.0 * rng.random((n, t)) - 1.5).astype(np.float64)
        for r in range(n):  # outliers, masked points, an all-NaN row, an all-negative row
            for _ in range(r % 4):
                lh[r, int(t * rng.random())] = 100.0 * rng.random() - 50.0
            if r % 5 == 0 and t > 3:
                lh[r, int(t * rng.random())] = np.nan
        if n > 8:
            lh[7, :] = np.nan
            lh[8, :] = -np.abs(lh[8, :]) - 0.1
        valid, lower, upper = reference_sequence(lh, lo, hi, ns, clip)
        # the reference's function itself; the bounds (which it does not return) come from the sequence above, whose mask must
        # be the same
        valid_ref = SigmaGClipping(lo, hi, ns, clip).compute_clipped_sigma_g_matrix(lh)
        assert valid_ref.shape == valid.shape and np.array_equal(valid_ref, valid), i
        valid = valid_ref
        out[f"lh_{i}"] = lh
        out[f"cfg_{i}"] = np.array([lo, hi, ns, float(clip)])
        out[f"valid_{i}"] = valid
        out[f"lower_{i}"] = lower
        out[f"upper_{i}"] = upper
    out["torch_version"] = np.array(torch.__version__)
    out["source"] = np.array("kbmod.filters.sigma_g_filter.SigmaGClipping.compute_clipped_sigma_g_matrix (reference source, imported)")
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sigma_g_matrix.npz"), **out)
    print("wrote", len(cases), "cases from the reference's own function, torch", torch.__version__)


if __name__ == "__main__":
    main()

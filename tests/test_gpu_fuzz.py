"""Randomised configurations: the LDS-staged kernel, the direct kernel and the oracle must agree bit
for bit whatever the image size, search bounds, K, epoch spacing, velocity ranges (negative,
off-image, boundary-exact), masks, encodings and thresholds."""

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

DIRECT, LDS, ENCODED_STAGING, TALL_TILES, WIDE_TILES = 2, 4, 16, 64, 128


def _config(seed):
    rng = np.random.default_rng(seed)
    T = int(rng.integers(2, 40))
    H, W = int(rng.integers(5, 90)), int(rng.integers(5, 150))
    if rng.random() < 0.3:
        times = np.arange(T) * float(rng.choice([0.25, 0.5, 1.0]))  # dyadic: rounding-boundary shifts
    else:
        times = np.sort(rng.random(T) * float(rng.uniform(0.5, 6.0)))
        times[0] = 0.0
    n_c = int(rng.choice([8, 24, 40, 64, 100]))
    vmax = float(rng.choice([2.0, 8.0, 25.0]))
    base_v = rng.uniform(-vmax, vmax, 2)
    vx = (base_v[0] + np.cumsum(rng.uniform(-0.4, 0.6, n_c))).astype(np.float32)  # neighbouring candidates stay close
    vy = (base_v[1] + np.cumsum(rng.uniform(-0.5, 0.4, n_c))).astype(np.float32)
    if rng.random() < 0.4:
        vx[rng.integers(0, n_c)] = np.float32(rng.choice([1.0, -3.0, 2.0, 0.0]))
    cfg = {"K": int(rng.choice([1, 3, 8, 12, 20])), "min_lh": float(rng.choice([-1e30, 0.0, 1.5])),
           "min_obs": int(rng.integers(0, T))}
    if rng.random() < 0.5:
        x0, y0 = int(rng.integers(-20, 10)), int(rng.integers(-20, 5))
        cfg["xb"] = (x0, x0 + int(rng.integers(1, W + 30)))
        cfg["yb"] = (y0, y0 + int(rng.integers(1, H + 30)))
    if rng.random() < 0.3:
        cfg["sigmag"] = (0.25, 0.75, 0.7413, float(rng.choice([-5.0, 2.0])))
    num_bytes = int(rng.choice([-1, -1, 1, 2]))
    mask = float(rng.choice([0.0, 0.02, 0.2]))
    objects = [(int(rng.integers(0, W)), int(rng.integers(0, H)), float(vx[0]), float(vy[0]), 200.0)]
    stack = util.make_stack(T, H, W, seed=seed, noise=float(rng.uniform(0.5, 4.0)), psf=float(rng.choice([0.5, 1.0])),
                            objects=objects, mask_fraction=mask, times=times)
    return stack, vx, vy, cfg, num_bytes


@pytest.mark.parametrize("seed", range(1000, 1030))
def test_random_configuration(kb, orc, seed):
    stack, vx, vy, cfg, num_bytes = _config(seed)
    a, exp, s1 = util.run_both(kb, orc, stack, vx, vy, cfg, num_bytes=num_bytes, flags=DIRECT)
    b, _, s2 = util.run_both(kb, orc, stack, vx, vy, cfg, num_bytes=num_bytes, flags=LDS | WIDE_TILES)
    assert a.shape == exp.shape and np.array_equal(a, exp), f"direct kernel differs from the oracle (seed {seed})"
    assert np.array_equal(b, exp), f"kernel variant {s2.last_search_stats()['kernel_variant']} differs (seed {seed})"
    t, _, _ = util.run_both(kb, orc, stack, vx, vy, cfg, num_bytes=num_bytes, flags=LDS | TALL_TILES)
    assert np.array_equal(t, exp), f"64 x 16 tiles differ (seed {seed})"
    if num_bytes != -1:
        c, _, _ = util.run_both(kb, orc, stack, vx, vy, cfg, num_bytes=num_bytes, flags=LDS | ENCODED_STAGING)
        assert np.array_equal(c, exp)


# The code-path switches of the search entry points (include/kbmod_hip.h `flags`) and the environment switches the library reads
# per search (KBMOD_LIST_MODE, KBMOD_CHUNK, KBMOD_EDGE_COUNTS): none of them may change a bit of the result.
FLAG_BITS = [8,      # double-precision decode of encoded samples
             32,     # never the count-free specialisation
             1024,   # the lists' floor (search_all applies the post-filter the flag relies on)
             2048]   # re-make the padded copy
SWITCHES = {"KBMOD_LIST_MODE": ["0", "1", "2", "3", "4"], "KBMOD_CHUNK": ["8", "16", "32"], "KBMOD_EDGE_COUNTS": ["0", "1"]}


@pytest.mark.parametrize("seed", range(2000, 2040))
def test_random_flags_and_switches(kb, orc, seed, monkeypatch):
    """Every seed draws its own subset of the flag bits and of the environment switches on top of a random configuration; the
    staged kernel (either tile height, drawn too) must still equal the oracle bit for bit, and must say which switches were set."""
    stack, vx, vy, cfg, num_bytes = _config(seed)
    rng = np.random.default_rng(seed + 77)
    flags = LDS | int(rng.choice([TALL_TILES, WIDE_TILES, 0]))
    for bit in FLAG_BITS:
        if rng.random() < 0.35:
            flags |= bit
    if num_bytes != -1 and rng.random() < 0.3:
        flags |= ENCODED_STAGING
    set_names = []
    for name, values in SWITCHES.items():
        if rng.random() < 0.5:
            monkeypatch.setenv(name, str(rng.choice(values)))
            set_names.append(name)
    got, exp, s = util.run_both(kb, orc, stack, vx, vy, cfg, num_bytes=num_bytes, flags=flags)
    stats = s.last_search_stats()
    assert got.shape == exp.shape and np.array_equal(got, exp), \
        f"flags {flags} switches {set_names} kernel variant {stats['kernel_variant']} differ from the oracle (seed {seed})"
    order = ["KBMOD_CHUNK", "KBMOD_LIST_MODE", "KBMOD_EDGE_COUNTS"]   # bit i of env_overrides (search_kernels.hip: kSwitches)
    for name in set_names:
        assert stats["env_overrides"] & (1 << order.index(name)), (name, stats["env_overrides"])

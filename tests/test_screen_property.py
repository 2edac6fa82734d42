"""The screen of the per-chunk finish (kbmod_amd/csrc/search_device.h: screen_floor / screen_key / screened_out) restated in
numpy float32, operation for operation, and held to the property the search kernels rely on: a candidate the screen rejects
ALSO fails the exact test `lh > threshold` with the correctly rounded likelihood psi / sqrt(phi) (kernels.cu:186-190, 318-330).
Whatever the screen lets through is decided exactly by the kernels; only a false rejection could change a result."""

import numpy as np

F32 = np.float32
FLT_MIN = F32(1.17549435e-38)
SURE_SEEN = [0]
FTZ = [False]   # emulate a multiplier that flushes denormal results to zero (either behaviour of the device must be safe)


def _mul(a, b):
    with np.errstate(all="ignore"):
        r = (a * b).astype(F32)
        if FTZ[0]:
            r = np.where(np.abs(r) < FLT_MIN, np.copysign(F32(0.0), r), r).astype(F32)
    return r


def screen_floor(thr):
    with np.errstate(all="ignore"):
        return (thr - np.abs(thr) * F32(3.814697265625e-06) - FLT_MIN).astype(F32)


def screen_key(thr):
    with np.errstate(all="ignore"):
        f = screen_floor(thr)
        f = np.where(np.abs(f) < F32(2.0 ** -40), F32(-(2.0 ** -40)), f).astype(F32)   # (NaN compares false: stays NaN)
        k = _mul(f, np.abs(f))
        normal = np.isfinite(k) & (np.abs(k) >= FLT_MIN)
    return np.where(normal, k, F32(np.nan)).astype(F32)


def screened_out(psi, phi, key):
    with np.errstate(all="ignore"):
        s = _mul(psi, np.abs(psi))
        t = _mul(key, phi)
        normal = np.isfinite(t) & (np.abs(t) >= FLT_MIN)
        return (phi > 0) & normal & (s <= t)


def sure_key(thr):
    with np.errstate(all="ignore"):
        f = (thr + np.abs(thr) * F32(3.814697265625e-06) + FLT_MIN).astype(F32)
        f = np.where(np.abs(f) < F32(2.0 ** -40), F32(2.0 ** -40), f).astype(F32)   # (NaN compares false: stays NaN)
        k = _mul(f, np.abs(f))
        normal = np.isfinite(k) & (np.abs(k) >= FLT_MIN)
    return np.where(normal, k, F32(np.nan)).astype(F32)


def surely_in(psi, phi, key_hi):
    with np.errstate(all="ignore"):
        s = _mul(psi, np.abs(psi))
        t = _mul(key_hi, phi)
        normal = np.isfinite(t) & (np.abs(t) >= FLT_MIN)
        return (phi > 0) & normal & (s > t)


def exact_lh(psi, phi):
    with np.errstate(all="ignore"):
        lh = (psi / np.sqrt(phi, dtype=F32)).astype(F32)   # IEEE sqrt and divide: correctly rounded, like the kernels'
    return np.where(phi > 0, lh, F32(-1.0)).astype(F32)


def _check(psi, phi, thr):
    psi, phi, thr = (np.asarray(v, dtype=F32) for v in (psi, phi, thr))
    out = screened_out(psi, phi, screen_key(thr))
    lh = exact_lh(psi, phi)
    with np.errstate(invalid="ignore"):
        enters = lh > thr
    bad = out & enters
    assert not bad.any(), (psi[bad][:5], phi[bad][:5], thr[bad][:5], lh[bad][:5])
    # the other side (the emitting instances, search_device.h: sure_key / surely_in): what the products put above the raised
    # threshold passes the exact test `!(lh < threshold)` of kernels.cu:201-203
    sure = surely_in(psi, phi, sure_key(thr))
    with np.errstate(invalid="ignore"):
        fails = lh < thr
    bad = sure & fails
    assert not bad.any(), (psi[bad][:5], phi[bad][:5], thr[bad][:5], lh[bad][:5])
    assert not (sure & out).any()
    SURE_SEEN[0] += int(sure.sum())
    return out


import pytest  # noqa: E402


@pytest.fixture(params=[False, True], ids=["denormals", "flush_to_zero"], autouse=True)
def _multiplier(request):
    FTZ[0] = request.param
    yield
    FTZ[0] = False


def test_rejected_candidates_fail_the_exact_test_random():
    rng = np.random.default_rng(5)
    n = 2_000_000
    # sums of the magnitude a search sees, thresholds around the likelihoods they give
    psi = (rng.standard_normal(n) * 30).astype(F32)
    phi = (rng.random(n) * 60 + 0.01).astype(F32)
    lh = exact_lh(psi, phi)
    thr = (lh * (1 + rng.standard_normal(n).astype(F32) * F32(1e-3))).astype(F32)
    out = _check(psi, phi, thr)
    assert 0.3 < out.mean() < 0.7          # and the screen does reject: about half of these lie below their threshold
    sure = surely_in(psi, phi, sure_key(thr))
    assert 0.3 < sure.mean() < 0.7 and (out | sure).mean() > 0.98   # ... decides the other half, and leaves a thin band
    # thresholds within a few ulps of the likelihood, both sides, both signs
    for k in (-4, -2, -1, 0, 1, 2, 4):
        t = lh.copy()
        for _ in range(abs(k)):
            t = np.nextafter(t, F32(np.inf if k > 0 else -np.inf))
        _check(psi, phi, t)


def test_rejected_candidates_fail_the_exact_test_at_the_edges():
    rng = np.random.default_rng(6)
    specials = np.array([0.0, -0.0, 1e-45, -1e-45, 1.17549435e-38, -1.17549435e-38, 1e-30, -1e-30, 1e-19, -1e-19, 1e-10, 1.0, -1.0,
                         3.0, 1e10, -1e10, 1.8e19, -1.8e19, 3e38, -3e38, np.inf, -np.inf, np.nan, -3.4028234663852886e38], dtype=F32)
    grid = np.array(np.meshgrid(specials, specials, specials)).reshape(3, -1)
    _check(grid[0], grid[1], grid[2])
    # random exponents over the whole float range
    n = 1_000_000
    mant = lambda: (rng.random(n) + 1).astype(F32)  # noqa: E731
    psi = (mant() * np.exp2(rng.integers(-140, 127, n)).astype(F32) * rng.choice([-1, 1], n)).astype(F32)
    phi = (mant() * np.exp2(rng.integers(-140, 127, n)).astype(F32)).astype(F32)
    thr = (mant() * np.exp2(rng.integers(-140, 127, n)).astype(F32) * rng.choice([-1, 1], n)).astype(F32)
    _check(psi, phi, thr)
    # thresholds at and around zero (min_lh = 0 under a list that is not full): negative likelihoods ARE rejected
    zero = np.zeros(n, dtype=F32)
    out0 = _check(psi, phi, zero)
    _check(psi, phi, np.full(n, -1.4e-45, dtype=F32))
    assert out0[(psi < -1e-3) & (phi > 1e-3) & (phi < 1e3)].all()
    # an empty slot's threshold: nothing may be rejected
    assert not screened_out(psi, phi, screen_key(np.full(n, -3.4028234663852886e38, dtype=F32))).any()

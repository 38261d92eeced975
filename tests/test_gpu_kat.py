"""GPU: the closed-form known answers of tests/test_oracle_kat.py asked of the HIP kernels directly (through the
drop-in API, not through the oracle): on-axis isotropic splats have one-line formulas for every output."""
import math

import numpy as np
import pytest

import hip_runner
from test_oracle_kat import CX, CY, axis_kwargs

pytestmark = pytest.mark.gpu


def _hip(z, s, op, col, bg=(0.0, 0.0, 0.0)):
    kw, cam = axis_kwargs(z, s, op, col, bg)
    color, radii, depth, alpha = hip_runner.run_hip(kw)["fwd"]
    return color, radii, depth, alpha, cam


def test_single_splat_closed_form():
    z, s, op, col, bg = 4.0, 0.25, 0.6, (0.9, 0.5, 0.1), (0.2, 0.3, 0.4)
    color, radii, depth, alpha, cam = _hip([z], [s], [op], [col], bg)
    s2 = (cam["fx"] * s / z) ** 2 + 0.3
    for d in (0, 1, 2, 3):
        a = op * math.exp(-0.5 * d * d / s2)
        a = 0.0 if a < 1.0 / 255.0 else a
        for ch in range(3):
            assert abs(color[ch, CY, CX + d] - (col[ch] * a + bg[ch] * (1 - a))) < 1e-5
        assert abs(depth[0, CY, CX + d] - z * a) < 1e-5 and abs(alpha[0, CY, CX + d] - a) < 1e-5
    assert radii[0] == math.ceil(3.0 * math.sqrt(s2 + math.sqrt(0.1)))


def test_clamp_thresholds_cull_and_stop():
    assert abs(_hip([3.0], [0.3], [1.0], [(1.0, 1.0, 1.0)])[3][0, CY, CX] - 0.99) < 1e-6
    c, r, d, a, _ = _hip([3.0], [0.3], [1.0 / 255.0 - 1e-5], [(1.0, 1.0, 1.0)], bg=(0.5, 0.5, 0.5))
    assert np.all(c == 0.5) and np.all(a == 0.0)
    c, r, d, a, _ = _hip([0.2], [0.02], [0.9], [(1.0, 0.0, 0.0)], bg=(0.1, 0.1, 0.1))
    assert np.all(a == 0.0) and r[0] == 0
    # transmittance stop: alpha 0.95 each -> T = 5e-2, 2.5e-3, 1.25e-4, then 6.25e-6 < 1e-4: splats 4 and 5 are not blended
    cols = [(1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (1.0, 1.0, 0.0), (0.0, 0.0, 1.0), (0.0, 0.0, 1.0)]
    c, r, d, a, _ = _hip([1.0, 2.0, 3.0, 4.0, 5.0], [0.3] * 5, [0.95] * 5, cols)
    w = [0.95, 0.95 * 0.05, 0.95 * 0.0025]
    assert abs(a[0, CY, CX] - sum(w)) < 1e-6 and c[2, CY, CX] == 0.0 and abs(c[0, CY, CX] - (w[0] + w[2])) < 1e-6


def test_compositing_order():
    z, op = [5.0, 2.0, 3.5], [0.5, 0.4, 0.7]
    cols = [(1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)]
    c, r, d, a, _ = _hip(z, [0.4] * 3, op, cols, bg=(1.0, 1.0, 1.0))
    T, C, D, A = 1.0, np.zeros(3), 0.0, 0.0
    for i in np.argsort(z):
        C += np.array(cols[i]) * op[i] * T; D += z[i] * op[i] * T; A += op[i] * T
        T *= 1.0 - op[i]
    assert np.allclose(c[:, CY, CX], C + T, atol=1e-5) and abs(d[0, CY, CX] - D) < 1e-5 and abs(a[0, CY, CX] - A) < 1e-5

"""CPU: known-answer tests of the oracle against the PUBLISHED algorithm, worked out by hand.

The rasterizer source is not in /root/reference (un-vendored submodule; parity unpinned, see oracle/gsr_oracle.c), so
the oracle cannot be checked against reference output.  What can be checked independently of our own restatement are
closed-form consequences of the published method (Kerbl et al. 2023, sections 4-6, with the constants and the depth /
alpha outputs of the fork listed in SURVEY.md Appendix A): for a single isotropic Gaussian on the optical axis of an
identity-pose camera everything -- projected covariance, low-pass, alpha, clamp, thresholds, compositing, radius,
pixel-coordinate convention -- has a one-line formula.  These tests hold the oracle to those formulas; the HIP kernels
are then held to the oracle by the parity tests."""
import math

import numpy as np
import pytest
import torch

import parity
from oracle import binding

W = H = 17          # odd: the principal point falls exactly on the centre of pixel (8, 8): ((0 + 1) 17 - 1) / 2 = 8
CX = CY = 8


def axis_kwargs(z, scale, opacity, color, bg=(0.0, 0.0, 0.0)):
    """kwargs (+ camera) of isotropic Gaussians on the optical axis of the identity-pose camera of synthetic.make_camera."""
    cam = parity.syn.make_camera(W, H)
    n = len(z)
    means = torch.tensor([[0.0, 0.0, zi] for zi in z], dtype=torch.float32)
    kw = dict(means3D=means, opacities=torch.tensor(opacity, dtype=torch.float32).view(n, 1), viewmatrix=cam["viewmatrix"],
              projmatrix=cam["projmatrix"], campos=cam["campos"], bg=torch.tensor(bg, dtype=torch.float32), image_height=H,
              image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], sh_degree=0,
              colors_precomp=torch.tensor(color, dtype=torch.float32).view(n, 3),
              scales=torch.tensor([[s, s, s] for s in scale], dtype=torch.float32),
              rotations=torch.tensor([[1.0, 0.0, 0.0, 0.0]] * n, dtype=torch.float32))
    return kw, cam


def _render(z, scale, opacity, color, bg=(0.0, 0.0, 0.0)):
    kw, cam = axis_kwargs(z, scale, opacity, color, bg)
    o = binding.OracleRender(**kw)
    o.forward()
    return o, cam


def _sigma2(cam, s, z):
    return (cam["fx"] * s / z) ** 2 + 0.3          # EWA projection of an isotropic splat on the axis + the 0.3 low-pass


def test_single_gaussian_centre_pixel_and_falloff():
    z, s, op, col, bg = 4.0, 0.25, 0.6, (0.9, 0.5, 0.1), (0.2, 0.3, 0.4)
    o, cam = _render([z], [s], [op], [col], bg)
    s2 = _sigma2(cam, s, z)
    for d in (0, 1, 2, 3):                           # pixels along the row through the centre
        a = op * math.exp(-0.5 * d * d / s2)
        if a < 1.0 / 255.0:
            a = 0.0
        for ch in range(3):
            assert abs(o.color[ch, CY, CX + d] - (col[ch] * a + bg[ch] * (1 - a))) < 2e-6
        assert abs(o.depth[0, CY, CX + d] - z * a) < 1e-5      # fork: depth = sum z alpha T
        assert abs(o.alpha[0, CY, CX + d] - a) < 2e-6          # fork: alpha = sum alpha T
    # ceil(3 sqrt(lambda_max)), lambda = mid + sqrt(max(0.1, mid^2 - det)): for an isotropic splat mid^2 = det, so the
    # published floor of 0.1 under the root adds sqrt(0.1) to the eigenvalue
    assert o.radii[0] == math.ceil(3.0 * math.sqrt(s2 + math.sqrt(0.1)))
    # isotropy: same value at equal distance along x, y and the diagonal-free symmetry
    assert abs(o.color[0, CY, CX + 2] - o.color[0, CY + 2, CX]) < 1e-7 and abs(o.color[0, CY, CX + 2] - o.color[0, CY, CX - 2]) < 1e-7


def test_alpha_clamp_and_skip_threshold():
    o, _ = _render([3.0], [0.3], [1.0], [(1.0, 1.0, 1.0)])
    assert abs(o.alpha[0, CY, CX] - 0.99) < 1e-7               # alpha = min(0.99, o G)
    o, _ = _render([3.0], [0.3], [1.0 / 255.0 - 1e-5], [(1.0, 1.0, 1.0)], bg=(0.5, 0.5, 0.5))
    assert np.all(o.color == 0.5) and np.all(o.alpha == 0.0)   # alpha < 1/255 contributes nowhere
    o, _ = _render([3.0], [0.3], [1.0 / 255.0 + 1e-4], [(1.0, 1.0, 1.0)])
    assert o.alpha[0, CY, CX] > 0.0                            # ... and just above the threshold it does (at the centre only)
    assert o.alpha[0, CY, CX + 3] == 0.0


def test_near_plane_cull():
    o, _ = _render([0.2], [0.02], [0.9], [(1.0, 0.0, 0.0)], bg=(0.1, 0.1, 0.1))     # z_view <= 0.2 is culled
    assert np.all(o.alpha == 0.0) and o.radii[0] == 0 and np.allclose(o.color, 0.1)
    o, _ = _render([0.21], [0.02], [0.9], [(1.0, 0.0, 0.0)])
    assert o.alpha[0, CY, CX] > 0.5 and o.radii[0] > 0


def test_front_to_back_compositing_order():
    z, op = [5.0, 2.0, 3.5], [0.5, 0.4, 0.7]                       # given out of order: sorted by depth inside
    cols = [(1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)]
    o, _ = _render(z, [0.4, 0.4, 0.4], op, cols, bg=(1.0, 1.0, 1.0))
    order = np.argsort(z)
    T, C, D, A = 1.0, np.zeros(3), 0.0, 0.0
    for i in order:
        a = op[i]                                                   # centre pixel: G = 1
        C += np.array(cols[i]) * a * T; D += z[i] * a * T; A += a * T
        T *= 1.0 - a
    assert np.allclose(o.color[:, CY, CX], C + T * 1.0, atol=2e-6)
    assert abs(o.depth[0, CY, CX] - D) < 1e-5 and abs(o.alpha[0, CY, CX] - A) < 2e-6


def test_transmittance_stop():
    """alpha = 0.95 each: T = 1 -> 5e-2 -> 2.5e-3 -> 1.25e-4; the fourth splat would leave T (1 - a) = 6.25e-6 < 1e-4, so it
    (and everything behind it) is not blended: the pixel keeps T = 1.25e-4 and shows nothing of splats 4 and 5.
    (0.95 on purpose: with the clamp value 0.99 the second product is 1e-4 in exact arithmetic and 9.99998e-5 in
    binary32 -- a knife edge on which a float64 and a binary32 implementation legitimately disagree.)"""
    z = [1.0, 2.0, 3.0, 4.0, 5.0]
    cols = [(1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (1.0, 1.0, 0.0), (0.0, 0.0, 1.0), (0.0, 0.0, 1.0)]
    o, _ = _render(z, [0.3] * 5, [0.95] * 5, cols)
    w = [0.95, 0.95 * 0.05, 0.95 * 0.0025]
    assert abs(o.alpha[0, CY, CX] - sum(w)) < 1e-6
    assert o.color[2, CY, CX] == 0.0
    assert abs(o.color[0, CY, CX] - (w[0] + w[2])) < 1e-6 and abs(o.color[1, CY, CX] - (w[1] + w[2])) < 1e-6


def test_pixel_coordinate_convention():
    """ndc2Pix(v, S) = ((v + 1) S - 1) / 2: a point at x_ndc lands at pixel ((x_ndc + 1) W - 1) / 2; shifting the
    splat by exactly one pixel on the image plane moves the peak by one pixel."""
    cam = parity.syn.make_camera(W, H)
    z = 4.0
    dx = z / cam["fx"]                                              # one pixel on the image plane at depth z
    kw = dict(means3D=torch.tensor([[3 * dx, -2 * dx, z]]), opacities=torch.tensor([[0.8]]), viewmatrix=cam["viewmatrix"],
              projmatrix=cam["projmatrix"], campos=cam["campos"], bg=torch.zeros(3), image_height=H, image_width=W,
              tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], sh_degree=0, colors_precomp=torch.ones(1, 3),
              scales=torch.full((1, 3), 0.2), rotations=torch.tensor([[1.0, 0.0, 0.0, 0.0]]))
    o = binding.OracleRender(**kw)
    o.forward()
    iy, ix = np.unravel_index(np.argmax(o.alpha[0]), o.alpha[0].shape)
    assert (iy, ix) == (CY - 2, CX + 3) and abs(o.alpha[0, iy, ix] - 0.8) < 1e-5

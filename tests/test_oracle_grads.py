"""Pins the oracle's hand-derived VJPs (K2, K9, K12, K13, warp) with fp64 central finite differences.

The reference holds no tests for this path, so the backward formulas are validated against the
oracle's own forward.  CPU only.
"""
import math

import numpy as np
import pytest

W, H = 64, 48
K = np.array([[60.0, 0, 31.5], [0, 60.0, 23.5], [0, 0, 1]])


def _scene(rng, n, c):
    means = np.stack([rng.uniform(-1, 1, n), rng.uniform(-0.8, 0.8, n), rng.uniform(1.5, 4.0, n)], -1)
    quats = rng.normal(size=(n, 4))
    scales = np.exp(rng.uniform(math.log(0.05), math.log(0.3), (n, 3)))
    views = []
    for i in range(c):
        a = 0.05 * i
        V = np.eye(4)
        V[:3, :3] = np.array([[math.cos(a), 0, -math.sin(a)], [0, 1, 0], [math.sin(a), 0, math.cos(a)]])
        V[:3, 3] = [0.02 * i, -0.01 * i, 0.03 * i]
        views.append(V)
    return means, quats, scales, np.stack(views), np.stack([K] * c)


def _fd(f, x, idxs, h=1e-6):
    out = []
    for i in idxs:
        xp, xm = x.copy(), x.copy()
        xp.flat[i] += h
        xm.flat[i] -= h
        out.append((f(xp) - f(xm)) / (2 * h))
    return np.array(out)


def test_project_bwd_fd(oracle64):
    o = oracle64
    rng = np.random.default_rng(0)
    n, c = 8, 2
    means, quats, scales, views, Ks = _scene(rng, n, c)
    radii, m2d, dep, con, comp = o.project_fwd(means, quats, scales, views, Ks, W, H, calc_compensations=True)
    assert (radii > 0).all()
    w_m, w_d, w_c, w_k = (rng.normal(size=m2d.shape), rng.normal(size=dep.shape), rng.normal(size=con.shape),
                          rng.normal(size=comp.shape))

    def loss(means_, quats_, scales_, views_):
        r, a, b, cc, k = o.project_fwd(means_, quats_, scales_, views_, Ks, W, H, calc_compensations=True)
        return (a * w_m).sum() + (b * w_d).sum() + (cc * w_c).sum() + (k * w_k).sum()

    vm, vq, vs, vv = o.project_bwd(means, quats, scales, views, Ks, W, H, radii, w_m, w_d, w_c, w_k)
    for name, x, g, f in [
        ("means", means, vm, lambda x: loss(x, quats, scales, views)),
        ("quats", quats, vq, lambda x: loss(means, x, scales, views)),
        ("scales", scales, vs, lambda x: loss(means, quats, x, views)),
    ]:
        idxs = rng.choice(x.size, size=min(12, x.size), replace=False)
        num = _fd(f, x, idxs)
        np.testing.assert_allclose(g.flat[idxs], num, rtol=2e-5, atol=1e-7, err_msg=name)
    # viewmats: only the top 3 rows are inputs of the projection
    idxs = [r * 4 + cc_ + 16 * ci for ci in range(c) for r in range(3) for cc_ in range(4)]
    num = _fd(lambda x: loss(means, quats, scales, x), views, idxs)
    np.testing.assert_allclose(vv.flat[idxs], num, rtol=2e-5, atol=1e-6, err_msg="viewmats")
    assert np.all(vv[:, 3, :] == 0)


def test_project_bwd_fd_frustum_clamp(oracle64):
    """Gaussians outside the 1.3x frustum exercise the clamp branch of the J VJP (SURVEY §9.1 VJP note)."""
    o = oracle64
    rng = np.random.default_rng(1)
    means = np.array([[2.6, 0.1, 2.0], [-2.7, 0.2, 2.1], [0.1, 2.1, 2.0], [0.2, -2.2, 2.2]])
    quats = rng.normal(size=(4, 4))
    scales = np.full((4, 3), 0.8)
    views, Ks = np.eye(4)[None], K[None]
    radii, m2d, dep, con, _ = o.project_fwd(means, quats, scales, views, Ks, W, H)
    assert (radii > 0).all(), radii
    w_c = rng.normal(size=con.shape)
    w_m = rng.normal(size=m2d.shape)
    loss = lambda x: ((o.project_fwd(x, quats, scales, views, Ks, W, H)[3] * w_c).sum()
                      + (o.project_fwd(x, quats, scales, views, Ks, W, H)[1] * w_m).sum())
    vm, _, _, _ = o.project_bwd(means, quats, scales, views, Ks, W, H, radii, w_m, None, w_c)
    num = _fd(loss, means, range(means.size))
    np.testing.assert_allclose(vm.ravel(), num, rtol=2e-5, atol=1e-7)


def _raster_inputs(o, rng, n, c, ch):
    means, quats, scales, views, Ks = _scene(rng, n, c)
    radii, m2d, dep, con, _ = o.project_fwd(means, quats, scales, views, Ks, W, H)
    colors = rng.uniform(0.1, 0.9, (c, n, ch))
    opac = rng.uniform(0.2, 0.8, (c, n))
    bg = rng.uniform(0, 1, (c, ch))
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    tpg, ids, flat = o.isect_tiles(m2d, radii, dep, 16, tw, th)
    off = o.isect_offset_encode(ids, c, tw, th)
    return m2d, con, colors, opac, bg, off, flat


@pytest.mark.parametrize("ch", [3, 5])
def test_raster_bwd_fd(oracle64, ch):
    o = oracle64
    rng = np.random.default_rng(2)
    n, c = 10, 2
    m2d, con, colors, opac, bg, off, flat = _raster_inputs(o, rng, n, c, ch)
    render, alphas, last_ids, nt = o.raster_fwd(m2d, con, colors, opac, bg, W, H, 16, off, flat)
    assert (last_ids >= 0).any()
    w_r, w_a = rng.normal(size=render.shape), rng.normal(size=alphas.shape)

    def loss(m2d_, con_, col_, op_):
        r, a, _, _ = o.raster_fwd(m2d_, con_, col_, op_, bg, W, H, 16, off, flat)
        return (r * w_r).sum() + (a * w_a).sum()

    vm, vc, vcol, vop, _ = o.raster_bwd(m2d, con, colors, opac, bg, W, H, 16, off, flat, alphas, last_ids, w_r, w_a)
    for name, x, g, f in [
        ("means2d", m2d, vm, lambda x: loss(x, con, colors, opac)),
        ("conics", con, vc, lambda x: loss(m2d, x, colors, opac)),
        ("colors", colors, vcol, lambda x: loss(m2d, con, x, opac)),
        ("opacities", opac, vop, lambda x: loss(m2d, con, colors, x)),
    ]:
        idxs = rng.choice(x.size, size=min(10, x.size), replace=False)
        num = _fd(f, x, idxs, h=1e-6)
        # the compositing is piecewise smooth (alpha<1/255 / T<=1e-4 cuts); allow a little slack
        np.testing.assert_allclose(g.flat[idxs], num, rtol=5e-4, atol=5e-5, err_msg=name)


def test_raster_absgrad_and_visibility(oracle64):
    o = oracle64
    rng = np.random.default_rng(3)
    m2d, con, colors, opac, bg, off, flat = _raster_inputs(o, rng, 10, 1, 3)
    render, alphas, last_ids, nt = o.raster_fwd(m2d, con, colors, opac, bg, W, H, 16, off, flat, vis_min_T=0.5)
    nt0 = o.raster_fwd(m2d, con, colors, opac, bg, W, H, 16, off, flat, vis_min_T=0.0)[3]
    assert (nt0 >= nt).all() and nt0.sum() > nt.sum() > 0
    vr = rng.normal(size=render.shape)
    vm, _, _, _, vabs = o.raster_bwd(m2d, con, colors, opac, bg, W, H, 16, off, flat, alphas, last_ids, vr,
                                     np.zeros_like(alphas), absgrad=True)
    assert (vabs >= np.abs(vm) - 1e-12).all()


def test_sh_bwd_fd(oracle64):
    o = oracle64
    rng = np.random.default_rng(4)
    for deg in (0, 1, 2, 3):
        c, n, kc = 2, 5, 16
        dirs = rng.normal(size=(c, n, 3))
        coeffs = rng.normal(size=(n, kc, 3)) * 0.3
        w = rng.normal(size=(c, n, 3))
        vco, vd = o.sh_bwd(deg, dirs, coeffs, w)
        f_c = lambda x: (o.sh_fwd(deg, dirs, x) * w).sum()
        f_d = lambda x: (o.sh_fwd(deg, x, coeffs) * w).sum()
        idx = rng.choice(coeffs.size, 20, replace=False)
        np.testing.assert_allclose(vco.flat[idx], _fd(f_c, coeffs, idx), rtol=1e-5, atol=1e-8)
        idx = range(dirs.size)
        np.testing.assert_allclose(vd.ravel(), _fd(f_d, dirs, idx), rtol=1e-5, atol=1e-7)
        nb = (deg + 1) ** 2
        assert np.all(vco[:, nb:, :] == 0)


@pytest.mark.parametrize("padding", ["same", "valid"])
def test_ssim_bwd_fd(oracle64, padding):
    o = oracle64
    rng = np.random.default_rng(5)
    a = rng.uniform(0, 1, (1, 2, 20, 24))
    b = np.clip(a + rng.normal(scale=0.1, size=a.shape), 0, 1)
    val, g = o.fused_ssim(a, b, padding)
    f = lambda x: float(o.fused_ssim(x, b, padding)[0])
    idx = rng.choice(a.size, 15, replace=False)
    np.testing.assert_allclose(g.flat[idx], _fd(f, a, idx, h=1e-6), rtol=1e-5, atol=1e-9)
    assert 0.0 < val < 1.0
    assert abs(float(o.fused_ssim(a, a, padding)[0]) - 1.0) < 1e-12


def test_warp_bwd_fd(oracle64):
    o = oracle64
    rng = np.random.default_rng(6)
    Hh, Ww = 24, 32
    Kc = np.array([[30.0, 0, 15.5], [0, 30.0, 11.5], [0, 0, 1]])
    Kinv = np.linalg.inv(Kc)
    c1 = rng.uniform(0, 1, (Hh, Ww, 3))
    d1 = rng.uniform(1, 2, (Hh, Ww))
    T = np.eye(4)
    T[:3, :3] += rng.normal(scale=0.01, size=(3, 3))
    T[:3, 3] = rng.normal(scale=0.02, size=3)
    w_r, w_n = rng.normal(size=(Hh, Ww, 3)), rng.normal(size=(1, Hh, Ww, 2)) * 0.1

    def loss(Tm):
        r, nw, _ = o.warp_fwd(Tm, Kc, Kinv, c1, d1)
        return (r * w_r).sum() + (nw * w_n).sum()

    vT = o.warp_bwd(T, Kc, Kinv, c1, d1, w_r, w_n)
    idx = [r * 4 + c for r in range(3) for c in range(4)]
    # bilinear sampling is piecewise linear in the warp: tiny step keeps every pixel inside its cell
    np.testing.assert_allclose(vT.flat[idx], _fd(loss, T, idx, h=1e-7), rtol=2e-4, atol=1e-4)
    assert np.all(vT[3] == 0)

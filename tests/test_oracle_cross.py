"""the C oracle (oracle/gsx_oracle.c) against a SECOND restatement of the same published
algorithms written independently in vectorised torch-CPU ops (oracle/torch_oracle.py), whose backward is torch autograd.
CPU only; nothing here touches the GPU or /root/reference.

What this pins: that the two restatements of SURVEY §9 agree (forward values, cull decisions, tile rectangles, list order,
last contributing entry) and that the oracle's hand-derived VJPs (K2, K9, K12) equal the automatic derivative of an
independently written forward, in float64 to ~1e-10.  What it cannot pin: the INFERRED constants themselves (see the module
docstring of oracle/torch_oracle.py and DESIGN.md §2)."""
import math

import numpy as np
import pytest
import torch

from oracle import torch_oracle as to


def _scene(n, seed, scale_shift=0.0):
    from gslam_amd.synthetic import make_scene
    sc = make_scene(n, seed)
    return (sc["means"].double(), sc["quats"].double(), torch.exp(sc["scales"].double() + scale_shift),
            torch.sigmoid(sc["opacities"].double()), torch.sigmoid(sc["colors"].double()))


def _cams(c, W, H):
    from gslam_amd.synthetic import make_cameras
    v, k = make_cameras(c, W, H)
    return v.double(), k.double()


EDGE_MEANS = [[0, 0, -1.0], [0, 0, 0.005], [0, 0, 2.0], [50.0, 0, 2.0], [0.0, 40.0, 2.0], [1.21, 0.9, 2.0], [0, 0, 0.02],
              [0.3, 0.2, 5.0], [-0.6, -0.45, 1.0], [0.0, 0.0, 1e11]]


@pytest.mark.parametrize("n,c,W,H", [(1, 1, 640, 480), (7, 2, 640, 480), (1000, 2, 640, 480), (1500, 8, 100, 70)])
def test_projection_second_restatement(oracle64, n, c, W, H):
    means, quats, scales, _, _ = _scene(n, 3, 0.5 if W < 200 else 0.0)
    if n >= 1000:      # behind the camera, nearer than near, beyond far, far off-image, huge splat, on the image border
        k = len(EDGE_MEANS)
        means[:k] = torch.tensor(EDGE_MEANS, dtype=torch.float64)
        quats[:k] = torch.tensor([1.0, 0, 0, 0], dtype=torch.float64)
        scales[:k] = 0.05
        scales[6] = 3.0
    viewmats, Ks = _cams(c, W, H)
    if W < 200:
        Ks = Ks.clone()
        Ks[:, :2] *= 0.6          # wider field of view on the small image: more splats on screen
        Ks[:, 0, 2], Ks[:, 1, 2] = W / 2 - 0.5, H / 2 - 0.5
    gin = [t.clone().requires_grad_(True) for t in (means, quats, scales, viewmats)]
    radii, m2d, dep, con, _ = to.project(gin[0], gin[1], gin[2], gin[3], Ks, W, H)
    o = oracle64.project_fwd(means.numpy(), quats.numpy(), scales.numpy(), viewmats.numpy(), Ks.numpy(), W, H)
    assert np.array_equal(radii.numpy(), o[0])                                   # same cull decisions, same radii
    if n >= 1000:
        assert (o[0] > 0).sum() > 50 and (o[0] == 0).sum() > 50
    np.testing.assert_allclose(m2d.detach().numpy(), o[1], rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(dep.detach().numpy(), o[2], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(con.detach().numpy(), o[3], rtol=1e-9, atol=1e-12)
    # tile rectangles: the torch restatement of §9.2 against the C count kernel
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    x0, y0, x1, y1 = to.tile_rects(m2d, radii, 16, tw, th)
    tpg, ids, flat = oracle64.isect_tiles(o[1], o[0], o[2], 16, tw, th)
    assert np.array_equal(((x1 - x0) * (y1 - y0)).numpy(), tpg)
    # K2: the oracle's hand-derived VJP against autograd of the independent forward
    g = torch.Generator().manual_seed(11)
    w = [torch.randn(t.shape, generator=g, dtype=torch.float64) for t in (m2d, dep, con)]
    (m2d * w[0]).sum().add((dep * w[1]).sum()).add((con * w[2]).sum()).backward()
    ov = oracle64.project_bwd(means.numpy(), quats.numpy(), scales.numpy(), viewmats.numpy(), Ks.numpy(), W, H, o[0],
                              w[0].numpy(), w[1].numpy(), w[2].numpy())
    for name, t, ref in zip(("means", "quats", "scales", "viewmats"), gin, ov):
        got = t.grad.numpy()
        if name == "viewmats":
            got = got.copy()
            got[:, 3, :] = 0.0                     # the bottom row [0 0 0 1] is not an input of the kernels
            ref = ref.copy()
            ref[:, 3, :] = 0.0
        scale = np.abs(ref).max() + 1e-30
        assert np.abs(got - ref).max() / scale < 1e-9, (name, np.abs(got - ref).max() / scale)


@pytest.mark.parametrize("n,c,W,H,ch,with_bg", [(1200, 1, 96, 64, 5, True), (800, 2, 100, 70, 3, False),
                                                 (400, 1, 40, 40, 1, True)])
def test_rasteriser_second_restatement(oracle64, n, c, W, H, ch, with_bg):
    """K8 forward and K9 backward: dense per-tile compositing in torch (autograd) against the C oracle, float64; deep lists
    (fat splats) so that the alpha >= 1/255, alpha <= 0.999 and T <= 1e-4 branches are all taken; ragged tiles (100x70)"""
    means, quats, scales, opac, _ = _scene(n, 5, 1.2)
    viewmats, Ks = _cams(c, W, H)
    Ks = Ks.clone()
    Ks[:, :2] *= 0.25
    Ks[:, 0, 2], Ks[:, 1, 2] = W / 2 - 0.5, H / 2 - 0.5
    o = oracle64.project_fwd(means.numpy(), quats.numpy(), scales.numpy(), viewmats.numpy(), Ks.numpy(), W, H)
    radii, m2d, dep, con = (torch.from_numpy(a) for a in o[:4])
    g = torch.Generator().manual_seed(2)
    colors = torch.rand(c, n, ch, generator=g, dtype=torch.float64)
    opacs = torch.rand(c, n, generator=g, dtype=torch.float64) * 1.2               # some above the 0.999 clamp
    opacs = opacs.clamp(0.02, 1.0)
    bg = torch.rand(c, ch, generator=g, dtype=torch.float64) if with_bg else None
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    tpg, ids, flat = oracle64.isect_tiles(o[1], o[0], o[2], 16, tw, th)
    off = oracle64.isect_offset_encode(ids, c, tw, th)
    o_render, o_alpha, o_last, _ = oracle64.raster_fwd(o[1], o[3], colors.numpy(), opacs.numpy(),
                                                       None if bg is None else bg.numpy(), W, H, 16, off, flat)
    gin = [t.clone().requires_grad_(True) for t in (m2d, con, colors, opacs)]
    tbg = None if bg is None else bg.clone().requires_grad_(True)
    render, alphas, last, counts = to.rasterize(gin[0], gin[1], gin[2], gin[3], radii, dep, W, H, backgrounds=tbg)
    # list lengths and order: per-tile counts, and the last contributing entry as a rank inside its tile's list
    offs = np.concatenate([off.reshape(-1), [len(flat)]])
    assert np.array_equal(counts.numpy().reshape(-1), np.diff(offs))
    tile_of = (np.arange(c)[:, None, None] * th + (np.arange(H) // 16)[None, :, None]) * tw + (np.arange(W) // 16)[None, None, :]
    rank = np.where(o_last >= 0, o_last - offs[tile_of], -1)
    assert np.array_equal(rank, last.numpy())
    assert (o_last >= 0).mean() > 0.5
    np.testing.assert_allclose(render.detach().numpy(), o_render, rtol=0, atol=1e-12)
    np.testing.assert_allclose(alphas.detach().numpy(), o_alpha, rtol=0, atol=1e-12)
    assert (1.0 - o_alpha).min() < 1e-3 < (1.0 - o_alpha).max()                      # saturated and open pixels
    # K9 against autograd
    w_r = torch.randn(render.shape, generator=g, dtype=torch.float64)
    w_a = torch.randn(alphas.shape, generator=g, dtype=torch.float64)
    ((render * w_r).sum() + (alphas * w_a).sum()).backward()
    ov = oracle64.raster_bwd(o[1], o[3], colors.numpy(), opacs.numpy(), None if bg is None else bg.numpy(), W, H, 16, off,
                             flat, o_alpha, o_last, w_r.numpy(), w_a.numpy())
    for name, t, ref in zip(("means2d", "conics", "colors", "opacities"), gin, ov[:4]):
        scale = np.abs(ref).max() + 1e-30
        err = np.abs(t.grad.numpy() - ref).max() / scale
        assert err < 1e-9, (name, err)
    if tbg is not None:
        ref_bg = (w_r.numpy() * (1.0 - o_alpha)).sum(axis=(1, 2))
        np.testing.assert_allclose(tbg.grad.numpy(), ref_bg, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("padding", ["same", "valid"])
@pytest.mark.parametrize("shape", [(2, 3, 37, 53), (1, 3, 11, 12), (1, 1, 64, 48)])
def test_ssim_second_formulation(oracle32, oracle64, padding, shape):
    """fused-ssim as five zero-padded F.conv2d + crop 5 (autograd gradient) against gsxo_ssim_fwd / gsxo_ssim_bwd"""
    g = torch.Generator().manual_seed(4)
    a = torch.rand(shape, generator=g, dtype=torch.float64)
    b = (a + 0.15 * torch.randn(shape, generator=g, dtype=torch.float64)).clamp(0, 1)
    ta = a.clone().requires_grad_(True)
    val = to.fused_ssim(ta, b, padding)
    val.backward()
    m64, _, _, _ = oracle64.ssim_maps(a.numpy(), b.numpy())
    np.testing.assert_allclose(to.ssim_map(a, b).numpy(), m64, rtol=0, atol=1e-12)
    oval, og = oracle64.fused_ssim(a.numpy(), b.numpy(), padding)
    assert abs(float(val) - float(oval)) < 1e-13
    assert np.abs(ta.grad.numpy() - og).max() < 1e-11 * max(1.0, np.abs(og).max()) + 1e-15
    # the float32 oracle (the one the HIP kernels are compared with) stays within float32 round-off of it
    oval32, og32 = oracle32.fused_ssim(a.numpy(), b.numpy(), padding)
    assert abs(float(oval32) - float(val)) < 2e-6
    assert np.abs(og32 - ta.grad.numpy()).max() < 2e-4 * np.abs(og).max() + 1e-9
    # window: 11 taps, sigma 1.5, normalised (the six leading digits SURVEY §9.5 quotes)
    w = to.gaussian_window()
    np.testing.assert_allclose(w[:6].numpy(), [0.0010284, 0.0075988, 0.0360008, 0.1093607, 0.2130055, 0.2660117], atol=5e-8)

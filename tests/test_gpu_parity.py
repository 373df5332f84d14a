"""HIP path vs CPU oracle on identical seeded inputs (the parity tests proper).  Run on the MI355X box:
    python -m pytest tests -m gpu -x -q
Bars (BASELINE.json north_star): bit-exact integer outputs (radii, tile counts, isect ids, flatten ids, offsets);
render within 1e-4 L1 per pixel; gradients within fp32 round-off of the oracle's.  Nothing here reads
/root/reference."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _scene(n, seed, c, W=640, H=480, dev=None):
    from gslam_amd.synthetic import make_cameras, make_scene
    sc = make_scene(n, seed)
    viewmats, Ks = make_cameras(c, W, H)
    if dev is not None:
        sc = {k: v.to(dev) for k, v in sc.items()}
        viewmats, Ks = viewmats.to(dev), Ks.to(dev)
    return sc, viewmats, Ks


def _np(t):
    return t.detach().cpu().numpy()


def test_selftest_wave_primitives(dev):
    from gslam_amd._lib import check, lib
    scratch = torch.empty(1 << 16, dtype=torch.uint8, device=dev)
    check(lib.gsx_selftest(scratch.data_ptr(), scratch.numel(), torch.cuda.current_stream().cuda_stream), "selftest")


@pytest.mark.parametrize("n,c", [(1, 1), (7, 2), (10000, 2), (50000, 8)])
def test_projection_forward_bit_exact(dev, oracle32, n, c):
    from gslam_amd import ops
    sc, viewmats, Ks = _scene(n, 0, c)
    scales = torch.exp(sc["scales"])
    radii, m2d, dep, con, comp = ops.fully_fused_projection(
        sc["means"].to(dev), None, sc["quats"].to(dev), scales.to(dev), viewmats.to(dev), Ks.to(dev), 640, 480,
        calc_compensations=True)
    o = oracle32.project_fwd(_np(sc["means"]), _np(sc["quats"]), _np(scales), _np(viewmats), _np(Ks), 640, 480,
                             calc_compensations=True)
    assert np.array_equal(_np(radii), o[0])
    assert np.array_equal(_np(m2d), o[1])
    assert np.array_equal(_np(dep), o[2])
    assert np.array_equal(_np(con), o[3])
    assert np.array_equal(_np(comp), o[4])
    if n >= 10000:
        assert (o[0] > 0).sum() > 100 and (o[0] == 0).sum() > 100  # both culled and visible rows are exercised


def test_projection_culling_edge_cases(dev, oracle32):
    """behind camera, beyond far, off-image bbox, huge radius, radius_clip."""
    from gslam_amd import ops
    means = torch.tensor([[0, 0, -1.0], [0, 0, 0.005], [0, 0, 2.0], [50.0, 0, 2.0], [0.0, 40.0, 2.0],
                          [1.21, 0.9, 2.0], [0, 0, 0.02], [0.3, 0.2, 5.0]])
    quats = torch.tensor([[1.0, 0, 0, 0]] * 8)
    scales = torch.tensor([[0.05, 0.05, 0.05]] * 8)
    scales[6] = 3.0
    viewmats, Ks = torch.eye(4)[None], torch.tensor([[[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1]]])
    for clip in (0.0, 40.0):
        r = ops.fully_fused_projection(means.to(dev), None, quats.to(dev), scales.to(dev), viewmats.to(dev),
                                       Ks.to(dev), 640, 480, radius_clip=clip)
        o = oracle32.project_fwd(_np(means), _np(quats), _np(scales), _np(viewmats), _np(Ks), 640, 480,
                                 radius_clip=clip)
        assert np.array_equal(_np(r[0]), o[0]) and np.array_equal(_np(r[1]), o[1])
    assert list(o[0][0][:2]) == [0, 0] and o[0][0][2] == 0  # clip=40 removes the small central splat too
    # packed variant (rasterization.py:390-419): compacted in flatten-id order
    cam, gid, pr, pm, pd, pc, _ = ops.fully_fused_projection(means.to(dev), None, quats.to(dev), scales.to(dev),
                                                             viewmats.to(dev), Ks.to(dev), 640, 480, packed=True)
    o = oracle32.project_fwd(_np(means), _np(quats), _np(scales), _np(viewmats), _np(Ks), 640, 480)
    sel = np.nonzero(o[0] > 0)
    assert np.array_equal(_np(cam), sel[0]) and np.array_equal(_np(gid), sel[1])
    assert np.array_equal(_np(pr), o[0][sel]) and np.array_equal(_np(pm), o[1][sel])


def test_projection_backward_matches_oracle(dev, oracle32, oracle64):
    from gslam_amd import ops
    n, c = 3000, 3
    sc, viewmats, Ks = _scene(n, 1, c)
    scales = torch.exp(sc["scales"])
    ins = [sc["means"], sc["quats"], scales, viewmats]
    gin = [t.clone().to(dev).requires_grad_(True) for t in ins]
    radii, m2d, dep, con, comp = ops.fully_fused_projection(gin[0], None, gin[1], gin[2], gin[3], Ks.to(dev), 640,
                                                            480, calc_compensations=True)
    g = torch.Generator().manual_seed(5)
    w = [torch.randn(t.shape, generator=g) for t in (m2d, dep, con, comp)]
    loss = sum((a * b.to(dev)).sum() for a, b in zip((m2d, dep, con, comp), w))
    loss.backward()
    # fp64 oracle is the yardstick; the fp32 oracle shows what fp32 round-off looks like
    o64 = oracle64.project_bwd(_np(ins[0]), _np(ins[1]), _np(ins[2]), _np(ins[3]), _np(Ks), 640, 480, _np(radii),
                               _np(w[0]), _np(w[1]), _np(w[2]), _np(w[3]))
    o32 = oracle32.project_bwd(_np(ins[0]), _np(ins[1]), _np(ins[2]), _np(ins[3]), _np(Ks), 640, 480, _np(radii),
                               _np(w[0]), _np(w[1]), _np(w[2]), _np(w[3]))
    for name, got, ref64, ref32 in zip(("means", "quats", "scales", "viewmats"), gin, o64, o32):
        got = _np(got.grad).astype(np.float64)
        scale = np.abs(ref64).max() + 1e-12
        err = np.abs(got - ref64).max() / scale
        err32 = np.abs(ref32 - ref64).max() / scale
        assert err < max(5e-5, 20 * err32), (name, err, err32)


@pytest.mark.parametrize("n,c,W,H", [(5000, 1, 640, 480), (20000, 2, 640, 480), (3000, 8, 640, 480),
                                      (4000, 1, 1920, 1080), (500, 1, 70, 50)])
def test_isect_bit_exact(dev, oracle32, n, c, W, H):
    from gslam_amd import ops
    from gslam_amd.synthetic import make_cameras, make_scene
    sc = make_scene(n, 2)
    viewmats, Ks = make_cameras(c, W, H)
    scales = torch.exp(sc["scales"]) * (3.0 if W < 100 else 1.0)
    o = oracle32.project_fwd(_np(sc["means"]), _np(sc["quats"]), _np(scales), _np(viewmats), _np(Ks), W, H)
    radii, m2d, dep = o[0], o[1], o[2]
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    tpg, ids, flat = ops.isect_tiles(torch.from_numpy(m2d).to(dev), torch.from_numpy(radii).to(dev),
                                     torch.from_numpy(dep).to(dev), 16, tw, th, n_cameras=c)
    off = ops.isect_offset_encode(ids, c, tw, th)
    otpg, oids, oflat = oracle32.isect_tiles(m2d, radii, dep, 16, tw, th)
    ooff = oracle32.isect_offset_encode(oids, c, tw, th)
    assert np.array_equal(_np(tpg), otpg)
    assert np.array_equal(_np(ids), oids)
    assert np.array_equal(_np(flat), oflat)
    assert np.array_equal(_np(off), ooff)
    # `off` came from the binning (attached to ids by isect_tiles); the standalone K7 kernel on a bare copy of the keys
    assert np.array_equal(_np(ops.isect_offset_encode(ids.clone(), c, tw, th)), ooff)
    assert len(oids) > 0
    # unsorted emission order is part of the contract too (sort=False)
    _, ids_u, flat_u = ops.isect_tiles(torch.from_numpy(m2d).to(dev), torch.from_numpy(radii).to(dev),
                                       torch.from_numpy(dep).to(dev), 16, tw, th, sort=False)
    _, oids_u, oflat_u = oracle32.isect_tiles(m2d, radii, dep, 16, tw, th, sort=False)
    assert np.array_equal(_np(ids_u), oids_u) and np.array_equal(_np(flat_u), oflat_u)


def test_isect_ties_and_empty(dev, oracle32):
    """equal depths must keep ascending flatten-id order; zero intersections give all-zero offsets."""
    from gslam_amd import ops
    m2d = np.array([[[100.0, 100.0], [100.0, 100.0], [104.0, 98.0], [100.0, 100.0]]], np.float32)
    radii = np.array([[20, 20, 20, 20]], np.int32)
    dep = np.array([[2.0, 1.0, 2.0, 2.0]], np.float32)
    tpg, ids, flat = ops.isect_tiles(torch.from_numpy(m2d).to(dev), torch.from_numpy(radii).to(dev),
                                     torch.from_numpy(dep).to(dev), 16, 40, 30)
    otpg, oids, oflat = oracle32.isect_tiles(m2d, radii, dep, 16, 40, 30)
    assert np.array_equal(_np(ids), oids) and np.array_equal(_np(flat), oflat)
    radii0 = np.zeros((1, 4), np.int32)
    tpg, ids, flat = ops.isect_tiles(torch.from_numpy(m2d).to(dev), torch.from_numpy(radii0).to(dev),
                                     torch.from_numpy(dep).to(dev), 16, 40, 30)
    assert ids.numel() == 0 and flat.numel() == 0 and int(tpg.sum()) == 0
    off = ops.isect_offset_encode(ids, 1, 40, 30)
    assert int(off.abs().sum()) == 0


def _oracle_pipeline(o, sc, viewmats, Ks, W, H, ch, seed=3):
    scales = torch.exp(sc["scales"])
    radii, m2d, dep, con, _ = o.project_fwd(_np(sc["means"]), _np(sc["quats"]), _np(scales), _np(viewmats), _np(Ks),
                                            W, H)
    c, n = radii.shape
    rng = np.random.default_rng(seed)
    colors = rng.uniform(0, 1, (c, n, ch)).astype(np.float32)
    opac = rng.uniform(0.05, 0.95, (c, n)).astype(np.float32)
    bg = rng.uniform(0, 1, (c, ch)).astype(np.float32)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    tpg, ids, flat = o.isect_tiles(m2d, radii, dep, 16, tw, th)
    off = o.isect_offset_encode(ids, c, tw, th)
    return m2d, con, colors, opac, bg, off, flat


@pytest.mark.parametrize("n,c,W,H,ch", [(3000, 1, 640, 480, 5), (3000, 2, 320, 240, 3), (2000, 1, 200, 120, 1),
                                         (2000, 1, 330, 250, 4), (2000, 1, 330, 250, 2), (2500, 4, 640, 480, 5),
                                         (2500, 4, 640, 480, 3), (14000, 4, 640, 480, 4)])
def test_raster_forward_backward_vs_oracle(dev, oracle32, n, c, W, H, ch):
    """the raster kernels the launch selects for each shape - one camera: quadrant kernels; (.., 4, 640x480): 4800 tiles, the
    full-chip backward (round 5: raster_bwd_full_kernel, the transposed contraction with opacity and colour columns; the 14 000
    Gaussian case has tile lists deeper than its 320 accumulator rows: the straight-to-memory path) - against the CPU oracle"""
    from gslam_amd import ops
    from gslam_amd.synthetic import make_cameras, make_scene
    sc = make_scene(n, 4)
    sc["scales"] = sc["scales"] + 0.7   # fatter splats: deeper per-pixel lists, exercises early termination
    viewmats, Ks = make_cameras(c, W, H)
    m2d, con, colors, opac, bg, off, flat = _oracle_pipeline(oracle32, sc, viewmats, Ks, W, H, ch)
    o_render, o_alpha, o_last, o_nt = oracle32.raster_fwd(m2d, con, colors, opac, bg, W, H, 16, off, flat)
    t = lambda a: torch.from_numpy(a).to(dev)
    gin = [t(m2d).requires_grad_(True), t(con).requires_grad_(True), t(colors).requires_grad_(True),
           t(opac).requires_grad_(True)]
    tbg = t(bg).requires_grad_(True)
    render, alphas, n_touched = ops.rasterize_to_pixels(gin[0], gin[1], gin[2], gin[3], W, H, 16, t(off), t(flat),
                                                        backgrounds=tbg)
    l1 = np.abs(_np(render) - o_render).mean()
    assert l1 < 1e-5, l1
    assert np.abs(_np(render) - o_render).max() < 2e-3
    # a pixel whose alpha sits on the 1/255 cut may flip between CPU expf and the GPU's exp2 (<= 1/255 of T each)
    assert np.abs(_np(alphas) - o_alpha).max() < 5e-3 and np.abs(_np(alphas) - o_alpha).mean() < 1e-6
    nt_mismatch = (_np(n_touched) != o_nt).mean()
    assert nt_mismatch < 2e-3, nt_mismatch
    assert o_nt.sum() > 0 and (o_last >= 0).mean() > 0.3
    # backward
    g = torch.Generator().manual_seed(7)
    w_r, w_a = torch.randn(render.shape, generator=g), torch.randn(alphas.shape, generator=g)
    (render * w_r.to(dev)).sum().add((alphas * w_a.to(dev)).sum()).backward()
    o_v = oracle32.raster_bwd(m2d, con, colors, opac, bg, W, H, 16, off, flat, o_alpha, o_last, _np(w_r), _np(w_a))
    for name, got, ref in zip(("means2d", "conics", "colors", "opacities"), gin, o_v[:4]):
        got = _np(got.grad)
        scale = np.abs(ref).max() + 1e-12
        # a pixel whose alpha sits on the 1/255 or T<=1e-4 cut may flip between CPU expf and GPU __expf:
        # bound the worst case loosely and the bulk tightly
        assert np.abs(got - ref).max() / scale < 5e-3, (name, np.abs(got - ref).max() / scale)
        assert np.abs(got - ref).mean() / (np.abs(ref).mean() + 1e-12) < 2e-4, name
    ref_bg = (_np(w_r) * (1.0 - o_alpha)).sum(axis=(1, 2))
    np.testing.assert_allclose(_np(tbg.grad), ref_bg, rtol=1e-3, atol=1e-3)


def test_raster_many_channels_and_absgrad(dev, oracle32):
    """CH > 5 goes through channel chunks; absgrad side channel as in gsplat."""
    from gslam_amd import ops
    from gslam_amd.synthetic import make_cameras, make_scene
    n, c, W, H, ch = 1500, 1, 200, 150, 7
    sc = make_scene(n, 9)
    sc["scales"] = sc["scales"] + 0.7
    viewmats, Ks = make_cameras(c, W, H)
    m2d, con, colors, opac, bg, off, flat = _oracle_pipeline(oracle32, sc, viewmats, Ks, W, H, ch)
    o_render, o_alpha, o_last, _ = oracle32.raster_fwd(m2d, con, colors, opac, bg, W, H, 16, off, flat)
    t = lambda a: torch.from_numpy(a).to(dev)
    tm, tc, tcol, top = (t(a).requires_grad_(True) for a in (m2d, con, colors, opac))
    render, alphas, _ = ops.rasterize_to_pixels(tm, tc, tcol, top, W, H, 16, t(off), t(flat),
                                                backgrounds=t(bg), absgrad=True)
    assert render.shape[-1] == 7 and np.abs(_np(render) - o_render).mean() < 1e-5
    render[..., :5].sum().backward()
    vr = np.zeros_like(o_render)
    vr[..., :5] = 1.0
    ov = oracle32.raster_bwd(m2d, con, colors, opac, bg, W, H, 16, off, flat, o_alpha, o_last, vr,
                             np.zeros_like(o_alpha), absgrad=True)
    # CH = 7 backward (two channel chunks) and the absgrad side channel (sum over pixels of |v_mean2d|, gsplat's
    # densification statistic) against the oracle's values
    for name, got, ref in zip(("means2d", "conics", "colors", "opacities"), (tm, tc, tcol, top), ov[:4]):
        got = _np(got.grad)
        scale = np.abs(ref).max() + 1e-12
        assert np.abs(got - ref).max() / scale < 5e-3, (name, np.abs(got - ref).max() / scale)
        assert np.abs(got - ref).mean() / (np.abs(ref).mean() + 1e-12) < 2e-4, name
    assert hasattr(tm, "absgrad") and tuple(tm.absgrad.shape) == (c, n, 2)
    got_abs, ref_abs = _np(tm.absgrad), ov[4]
    assert (ref_abs >= 0).all() and ref_abs.max() > 0 and (got_abs >= 0).all()
    assert np.abs(got_abs - ref_abs).max() / ref_abs.max() < 5e-3
    assert np.abs(got_abs - ref_abs).mean() / ref_abs.mean() < 2e-4
    assert (got_abs + 1e-6 * ref_abs.max() >= np.abs(_np(tm.grad))).all()          # |sum| <= sum of |.|


@pytest.mark.parametrize("mode,with_unc", [("RGB", True), ("RGB+D", True), ("RGB+D", False), ("RGB", False)])
def test_gslam_rasterization_fused_vs_oracle(dev, oracle32, mode, with_unc):
    from gslam_amd.rasterization import rasterization
    n, c, W, H = 4000, 2, 640, 480
    sc, viewmats, Ks = _scene(n, 11, c)
    sc["log_uncertainties"] = torch.linspace(-6.0, 1.0, n)
    d = {k: v.to(dev) for k, v in sc.items()}
    out = rasterization(d["means"], d["quats"], d["scales"], d["opacities"], d["colors"], viewmats.to(dev),
                        Ks.to(dev), W, H, packed=False, render_mode=mode,
                        log_uncertainties=d["log_uncertainties"] if with_unc else None,
                        backgrounds=torch.zeros(c, 3, device=dev))
    scales_gpu = torch.exp(d["scales"]).cpu().numpy()     # same exp as the device (bit-exact index contract)
    o = oracle32.gslam_rasterization(_np(sc["means"]), _np(sc["quats"]), _np(sc["scales"]), _np(sc["opacities"]),
                                     _np(sc["colors"]), _np(viewmats), _np(Ks), W, H, render_mode=mode,
                                     log_uncertainties=_np(sc["log_uncertainties"]) if with_unc else None,
                                     backgrounds=np.zeros((c, 3), np.float32), scales_override=scales_gpu)
    for f in ("radii", "tiles_per_gauss", "isect_ids", "flatten_ids", "isect_offsets"):
        assert np.array_equal(_np(getattr(out, f)), o[f]), f
    for f in ("means2d", "depths", "conics"):
        assert np.array_equal(_np(getattr(out, f)), o[f]), f
    assert np.abs(_np(out.rgbs) - o["rgbs"]).mean() < 1e-5
    assert np.abs(_np(out.alphas) - o["alphas"]).max() < 1e-4
    np.testing.assert_allclose(_np(out.opacities), o["opacities"], atol=1e-6)
    if mode == "RGB+D":
        assert np.abs(_np(out.depthmaps) - o["depthmaps"]).mean() < 1e-5
    else:
        assert out.depthmaps is None
    if with_unc:
        assert np.abs(_np(out.betas) - o["betas"]).mean() < 1e-5
        empty = o["alphas"][..., 0] == 0
        np.testing.assert_allclose(_np(out.betas)[empty], np.e, rtol=1e-6)
    else:
        assert out.betas is None
    assert out.n_touched.dtype == torch.int64 and (_np(out.n_touched) != o["n_touched"]).mean() < 2e-3
    assert (out.tile_width, out.tile_height, out.width, out.height, out.tile_size, out.n_cameras) == (40, 30, W, H, 16, c)


@pytest.mark.parametrize("mode", ["RGB", "RGB+D"])
def test_gslam_rasterization_vs_reference_generated_fixture(dev, mode):
    """the HIP rasterization() held DIRECTLY to tests/golden/rasterization_host_logic.npz: every RasterizationOutput field the
    reference's own gslam/rasterization.py:121-360 produced when oracle/gen_golden.py ran it (activations, channel order,
    e^1 beta background, output split), no oracle in between"""
    from gslam_amd.rasterization import rasterization
    g = dict(np.load(os.path.join(GOLD, "rasterization_host_logic.npz")))
    tag = mode.replace("+", "p")
    W, H = int(g["width"]), int(g["height"])
    C = g["viewmats"].shape[0]
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    out = rasterization(t("means"), t("quats"), t("scales"), t("opacities"), t("colors"), t("viewmats"), t("Ks"), W, H,
                        packed=False, render_mode=mode, log_uncertainties=t("log_uncertainties"),
                        backgrounds=torch.zeros(C, 3, device=dev))
    for f in ("radii", "tiles_per_gauss", "isect_ids", "flatten_ids", "isect_offsets"):
        assert np.array_equal(_np(getattr(out, f)), g[f"{tag}__{f}"]), f
    for f in ("means2d", "depths"):
        assert np.array_equal(_np(getattr(out, f)), g[f"{tag}__{f}"]), f
    # the fixture's scales went through the HOST's exp (the reference ran on the CPU): one ulp of a scale moves a conic by
    # an ulp or two - the integers above are what has to be (and is) identical
    np.testing.assert_allclose(_np(out.conics), g[f"{tag}__conics"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(_np(out.opacities), g[f"{tag}__opacities"], atol=1e-6)
    # bar: 1e-4 L1 per pixel (north star); a pixel on the alpha >= 1/255 cut may flip between CPU expf and the GPU's exp2
    # (one flipped contribution moves a channel by alpha * value <= value / 255: colours and alpha <= 1, beta <= e)
    for f, a, vmax in (("rgbs", out.rgbs, 1.0), ("alphas", out.alphas, 1.0), ("betas", out.betas, math.e)):
        err = np.abs(_np(a) - g[f"{tag}__{f}"])
        assert err.mean() < 1e-5 and err.max() < 1.2 * vmax / 255.0, (f, err.mean(), err.max())
    if mode == "RGB+D":
        err = np.abs(_np(out.depthmaps) - g[f"{tag}__depthmaps"])
        assert err.mean() < 1e-5 and err.max() < 2e-2, (err.mean(), err.max())
    else:
        assert out.depthmaps is None
    nt, ntg = _np(out.n_touched), g[f"{tag}__n_touched"]
    # touched-pixel counts: a pixel on the T > 0.5 / alpha >= 1/255 cut may flip (600 (camera, Gaussian) pairs here)
    assert nt.dtype == np.int64 and int((nt != ntg).sum()) <= 3 and int(np.abs(nt - ntg).max()) <= 2
    assert [out.tile_width, out.tile_height, out.width, out.height, out.tile_size, out.n_cameras] == list(g[f"{tag}__meta"])


def test_integer_outputs_under_host_expf(dev, oracle32):
    """The bit-exact integer tests feed the oracle the DEVICE's exp(log_scales) (scales_override): integers are exact given the
    activation.  This test says what the activation itself changes: with the host's expf (glibc) in the oracle instead, how
    many scales differ in the last bit, and how many radii / tile rectangles / tile-list entries move with them - printed,
    and bounded."""
    from gslam_amd.rasterization import rasterization
    n, c, W, H = 100_000, 1, 640, 480
    sc, viewmats, Ks = _scene(n, 0, c)
    d = {k: v.to(dev) for k, v in sc.items()}
    out = rasterization(d["means"], d["quats"], d["scales"], d["opacities"], d["colors"], viewmats.to(dev), Ks.to(dev), W, H,
                        packed=False, render_mode="RGB+D", log_uncertainties=d["log_uncertainties"],
                        backgrounds=torch.zeros(c, 3, device=dev))
    scales_gpu = torch.exp(d["scales"]).cpu().numpy()
    scales_host = np.exp(_np(sc["scales"]).astype(np.float32))          # glibc expf: what scales_override=None gives
    ulp_diff = int((scales_gpu.view(np.int32) != scales_host.view(np.int32)).sum())
    o = oracle32.project_fwd(_np(sc["means"]), _np(sc["quats"]), scales_host, _np(viewmats), _np(Ks), W, H)
    radii, tpg = _np(out.radii), _np(out.tiles_per_gauss)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    otpg, oids, oflat = oracle32.isect_tiles(o[1], o[0], o[2], 16, tw, th)
    d_radii = int((radii != o[0]).sum())
    d_rect = int((tpg != otpg).sum())
    d_M = abs(int(tpg.sum()) - int(otpg.sum()))
    print(f"host expf vs device exp: {ulp_diff} of {scales_gpu.size} scales differ (last bit), {d_radii} of {radii.size} radii, "
          f"{d_rect} tile rectangles, |dM| = {d_M} of {int(tpg.sum())} intersections")
    assert np.abs(scales_gpu.view(np.int32) - scales_host.view(np.int32)).max() <= 2       # ocml expf vs glibc expf: <= 2 ulp
    assert d_radii <= 1e-4 * radii.size and d_rect <= 1e-4 * radii.size and d_M <= 1e-4 * int(tpg.sum())
    vis = (radii > 0) & (o[0] > 0)
    assert np.abs(_np(out.means2d)[vis] - o[1][vis]).max() == 0.0      # means2d does not depend on the scales


def test_gslam_rasterization_fused_grads_match_unfused(dev):
    """The fused path (activations+packing inside K1/K2) must give the same gradients as the reference's op
    sequence (torch activations -> projection -> cat -> rasterize) built from the low-level ops; and
    means2d.retain_grad() must work (gslam/backend.py:326)."""
    from gslam_amd import ops
    from gslam_amd.rasterization import rasterization
    n, c, W, H = 3000, 2, 320, 240
    sc, viewmats, Ks = _scene(n, 12, c, W, H)
    sc["scales"] = sc["scales"] + 0.5
    names = ("means", "quats", "scales", "opacities", "colors", "log_uncertainties")
    g = torch.Generator().manual_seed(3)
    gt = torch.rand(c, H, W, 3, generator=g).to(dev)

    def loss_of(out_rgbs, out_depth, out_betas, alphas):
        return ((out_rgbs - gt).square().sum(-1) / (2 * out_betas.square())).mean() + 0.01 * out_depth.mean() \
            + 0.1 * alphas.mean()

    # fused
    pa = {k: sc[k].clone().to(dev).requires_grad_(True) for k in names}
    va = viewmats.clone().to(dev).requires_grad_(True)
    out = rasterization(pa["means"], pa["quats"], pa["scales"], pa["opacities"], pa["colors"], va, Ks.to(dev), W, H,
                        packed=False, render_mode="RGB+D", log_uncertainties=pa["log_uncertainties"],
                        backgrounds=torch.zeros(c, 3, device=dev))
    out.means2d.retain_grad()
    loss_of(out.rgbs, out.depthmaps, out.betas, out.alphas).backward()
    assert out.means2d.grad is not None and out.means2d.grad.shape == (c, n, 2)
    assert out.means2d.grad.abs().sum() > 0

    # unfused: the reference's own op sequence (rasterization.py:145-348) on the low-level ops
    pb = {k: sc[k].clone().to(dev).requires_grad_(True) for k in names}
    vb = viewmats.clone().to(dev).requires_grad_(True)
    opac = torch.sigmoid(pb["opacities"])
    cols = torch.sigmoid(pb["colors"])
    scales = torch.exp(pb["scales"])
    betas = torch.exp(pb["log_uncertainties"]).clamp(min=0.01)
    radii, m2d, dep, con, _ = ops.fully_fused_projection(pb["means"], None, pb["quats"], scales, vb, Ks.to(dev), W, H)
    m2d.retain_grad()
    colors = torch.cat((cols.expand(c, -1, -1), dep[..., None], betas.expand(c, -1)[..., None]), dim=-1)
    bg = torch.cat([torch.zeros(c, 4, device=dev), torch.full((c, 1), math.e, device=dev)], -1)
    tpg, ids, flat = ops.isect_tiles(m2d, radii, dep, 16, math.ceil(W / 16), math.ceil(H / 16))
    off = ops.isect_offset_encode(ids, c, math.ceil(W / 16), math.ceil(H / 16))
    render, alphas, nt = ops.rasterize_to_pixels(m2d, con, colors, opac.repeat(c, 1), W, H, 16, off, flat,
                                                 backgrounds=bg)
    assert torch.equal(radii, out.radii) and torch.equal(flat, out.flatten_ids)
    assert (render[..., :3] - out.rgbs).abs().max() < 1e-5
    loss_of(render[..., :3], render[..., 3], render[..., 4], alphas).backward()
    for k in names:
        a, b = pa[k].grad, pb[k].grad
        assert (a - b).abs().max() <= 1e-4 * b.abs().max() + 1e-9, (k, (a - b).abs().max(), b.abs().max())
    assert (va.grad - vb.grad).abs().max() <= 1e-3 * vb.grad.abs().max() + 1e-9
    assert (out.means2d.grad - m2d.grad).abs().max() <= 1e-4 * m2d.grad.abs().max() + 1e-9


@pytest.mark.parametrize("padding", ["same", "valid"])
def test_fused_ssim_vs_oracle(dev, oracle32, padding):
    from gslam_amd.ssim import fused_ssim
    g = torch.Generator().manual_seed(0)
    B, H, W = 2, 75, 131
    a = torch.rand(B, H, W, 3, generator=g)
    b = (a + 0.1 * torch.randn(B, H, W, 3, generator=g)).clamp(0, 1)
    ta = a.to(dev).requires_grad_(True)
    val = fused_ssim(ta.permute(0, 3, 1, 2), b.to(dev).permute(0, 3, 1, 2), padding=padding)   # backend.py:303-307
    val.backward()
    oval, og = oracle32.fused_ssim(_np(a.permute(0, 3, 1, 2)), _np(b.permute(0, 3, 1, 2)), padding)
    assert abs(float(val) - float(oval)) < 2e-6
    got = _np(ta.grad.permute(0, 3, 1, 2))
    assert np.abs(got - og).max() < 1e-3 * np.abs(og).max() + 1e-9
    with torch.no_grad():
        same = fused_ssim(ta.permute(0, 3, 1, 2), ta.permute(0, 3, 1, 2), padding=padding, train=False)
    assert abs(float(same) - 1.0) < 1e-6


@pytest.mark.parametrize("name", ["warp_48x64.npz", "warp_120x160.npz"])
def test_warp_vs_reference_golden(dev, name):
    from gslam_amd.warp import Warp
    g = dict(np.load(os.path.join(GOLD, name)))
    H, W = g["d1"].shape
    warp = Warp(torch.from_numpy(g["K"]).to(dev), H, W)
    f1 = torch.from_numpy(g["f1_pose"]).to(dev).requires_grad_(True)
    f2 = torch.from_numpy(g["f2_pose"]).to(dev).requires_grad_(True)
    res, nw, keep = warp(f1, f2, torch.from_numpy(g["c1"]).to(dev), torch.from_numpy(g["d1"]).to(dev))
    assert res.shape == (H, W, 3) and nw.shape == (1, H, W, 2) and keep.dtype == torch.bool
    np.testing.assert_allclose(_np(nw), g["normalized_warps"], atol=3e-6)
    border = (np.abs(np.abs(g["normalized_warps"][0]) - 1.0) < 1e-5).any(-1)
    assert (_np(keep) == g["keep_mask"])[~border].all()
    assert np.abs(_np(res) - g["result"]).max() < 2e-4 and np.abs(_np(res) - g["result"]).mean() < 5e-6
    (res[keep].sum() + 0.1 * nw.square().sum()).backward()             # same loss as oracle/gen_golden.py
    scale = max(np.abs(g["grad_f1"]).max(), 1.0)
    # all four rows: the reference's T = f1 @ inv(f2) hands a gradient to the bottom row of f2 (and zeros to that of f1)
    np.testing.assert_allclose(_np(f1.grad), g["grad_f1"], atol=5e-4 * scale)
    np.testing.assert_allclose(_np(f2.grad), g["grad_f2"], atol=5e-4 * scale)
    assert np.abs(g["grad_f2"][3]).max() > 1.0 and np.abs(g["grad_f1"][3]).max() == 0.0
    eye = torch.eye(4, device=dev)
    res_id, _, _ = warp(eye, eye, torch.from_numpy(g["c1"]).to(dev), torch.from_numpy(g["d1"]).to(dev))
    if "result_identity" in g and name == "warp_48x64.npz":
        assert np.abs(_np(res_id) - g["result_identity"]).max() < 1e-4


def test_sh_vs_oracle(dev, oracle32):
    from gslam_amd import ops
    g = torch.Generator().manual_seed(1)
    c, n = 2, 5000
    dirs = torch.randn(c, n, 3, generator=g)
    coeffs = torch.randn(n, 16, 3, generator=g) * 0.3
    radii = (torch.rand(c, n, generator=g) > 0.2).to(torch.int32)
    w = torch.randn(c, n, 3, generator=g)
    for deg in (0, 1, 2, 3):
        td, tc = dirs.to(dev).requires_grad_(True), coeffs.to(dev).requires_grad_(True)
        col = ops.spherical_harmonics(deg, td, tc, masks=radii.to(dev))
        ocol = oracle32.sh_fwd(deg, _np(dirs), _np(coeffs), _np(radii))
        assert np.abs(_np(col) - ocol).max() < 2e-6
        (col * w.to(dev)).sum().backward()
        ovc, ovd = oracle32.sh_bwd(deg, _np(dirs), _np(coeffs), _np(w), _np(radii))
        assert np.abs(_np(tc.grad) - ovc).max() < 1e-5 * max(1.0, np.abs(ovc).max())
        assert np.abs(_np(td.grad) - ovd).max() < 1e-4 * max(1.0, np.abs(ovd).max())


def test_quat_scale_to_covar_preci(dev, oracle32):
    from gslam_amd import ops
    g = torch.Generator().manual_seed(2)
    q, s = torch.randn(1000, 4, generator=g), torch.rand(1000, 3, generator=g) + 0.1
    cov, pre = ops.quat_scale_to_covar_preci(q.to(dev), s.to(dev))
    oc, op = oracle32.quat_scale_to_covar_preci(_np(q), _np(s))
    assert np.array_equal(_np(cov), oc) and np.array_equal(_np(pre), op)
    cov6, pre6 = ops.quat_scale_to_covar_preci(q.to(dev), s.to(dev), triu=True)      # (xx, xy, xz, yy, yz, zz)
    iu = ([0, 0, 0, 1, 1, 2], [0, 1, 2, 1, 2, 2])
    assert tuple(cov6.shape) == (1000, 6) and np.array_equal(_np(cov6), oc[:, iu[0], iu[1]])
    assert np.array_equal(_np(pre6), op[:, iu[0], iu[1]])
    only_p = ops.quat_scale_to_covar_preci(q.to(dev), s.to(dev), compute_covar=False)
    assert only_p[0] is None and np.array_equal(_np(only_p[1]), op)


def test_adam_matches_torch_and_oracle(dev, oracle32):
    from gslam_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(3)
    shapes = [(1000, 3), (1000, 4), (1000, 3), (1000,), (1000, 3), (1000,)]
    lrs = [1.6e-3, 5e-3, 5e-3, 2.5e-2, 1e-2, 2.5e-3]                # gslam/backend.py:53-58
    p0 = [torch.randn(s, generator=g) for s in shapes]
    ours = [p.clone().to(dev).requires_grad_(True) for p in p0]
    ref = [p.clone().to(dev).requires_grad_(True) for p in p0]
    opt = FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(ours, lrs)])
    ropts = [torch.optim.Adam([p], lr=lr) for p, lr in zip(ref, lrs)]
    cpu = [(_np(p), np.zeros(p.shape, np.float32), np.zeros(p.shape, np.float32)) for p in p0]
    for step in range(1, 6):
        grads = [torch.randn(s, generator=g) for s in shapes]
        for p, q, gr in zip(ours, ref, grads):
            p.grad = gr.clone().to(dev)
            q.grad = gr.clone().to(dev)
        opt.step()
        for ro in ropts:
            ro.step()
        cpu = [oracle32.adam(p, _np(gr), m, v, lr, step) for (p, m, v), gr, lr in zip(cpu, grads, lrs)]
    for p, q, (cp, _, _) in zip(ours, ref, cpu):
        assert (p - q).abs().max() < 1e-6
        assert np.abs(_np(p) - cp).max() < 1e-6


def test_adam_untouched_rows_stay_bit_identical(dev):
    """Gaussians no camera has seen yet have zero gradient and zero moments: the kernel skips their stores, torch.optim.Adam
    rewrites them with the same values - parameters and moments agree bit for bit, before and after such rows wake up"""
    from gslam_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(5)
    shapes = [(3000, 3), (3000, 4), (3000,)]
    p0 = [torch.randn(s, generator=g) for s in shapes]
    ours = [p.clone().to(dev).requires_grad_(True) for p in p0]
    ref = [p.clone().to(dev).requires_grad_(True) for p in p0]
    opt = FusedAdam([{"params": [p], "lr": 1e-2} for p in ours])
    ropts = [torch.optim.Adam([p], lr=1e-2) for p in ref]
    seen = torch.zeros(3000, dtype=torch.bool)
    for step in range(1, 7):
        seen |= torch.rand(3000, generator=g) < 0.25            # a quarter more of the rows gets a gradient each step
        for p, q, shp in zip(ours, ref, shapes):
            gr = torch.randn(shp, generator=g)
            gr[~seen] = 0.0
            p.grad = gr.clone().to(dev)
            q.grad = gr.clone().to(dev)
        opt.step()
        for ro in ropts:
            ro.step()
        for p, q, p_init in zip(ours, ref, p0):
            assert torch.equal(p.detach()[~seen.to(dev)], p_init.to(dev)[~seen.to(dev)])       # untouched rows: untouched
            assert (p - q).abs().max() < 1e-6
    assert int((~seen).sum()) > 100


@pytest.mark.parametrize("regularize,c", [(True, 2), (False, 2), (True, 17)])
def test_fused_mapping_loss_matches_reference_formulation(dev, regularize, c):
    """csrc/loss.hip (one pass, analytic gradient) vs the reference's own torch formulation of
    gslam/backend.py:273-318 (gslam_amd.mapping.mapping_loss) - values and every gradient."""
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.mapping import MapConfig, mapping_loss
    from gslam_amd.losses import fused_mapping_loss
    from gslam_amd.primitives import Camera, PoseZhou
    from gslam_amd.synthetic import make_cameras, make_scene
    n, W, H = 3000, 200, 152          # c=17: more cameras than gsx_loss_finish's split path handles in one row
    sc = make_scene(n, 21)
    sc["scales"] = sc["scales"] + 0.6
    viewmats, Ks = make_cameras(c, W, H)
    g = torch.Generator().manual_seed(2)
    gt = torch.rand(c, H, W, 3, generator=g).to(dev)
    conf = MapConfig()
    res = []
    for fused in (False, True):
        splats = GaussianSplattingData.from_dict(sc, dev)
        cams = [Camera(Ks[i].to(dev), H, W) for i in range(c)]
        poses = [PoseZhou(viewmats[i].to(dev)).to(dev) for i in range(c)]
        exposure = (torch.tensor([[0.1, -0.05], [-0.2, 0.03]]).repeat((c + 1) // 2, 1)[:c]
                    * torch.linspace(1.0, 0.5, c)[:, None]).to(dev).requires_grad_(True)
        out = splats(cams, poses, render_depth=True)
        if fused:
            total, pm = fused_mapping_loss(out, gt, exposure, splats.scales, ssim_weight=conf.ssim_weight,
                                           iso_weight=conf.isotropic_regularization_weight * 50,
                                           tv_weight=(conf.depth_regularization_weight * 1e3) if regularize else 0.0)
        else:
            conf2 = MapConfig(isotropic_regularization_weight=conf.isotropic_regularization_weight * 50,
                              depth_regularization_weight=conf.depth_regularization_weight * 1e3)
            total, pm = mapping_loss(splats, out, gt, exposure, conf2, regularize)
        total.backward()
        grads = {k: getattr(splats, k).grad.clone() for k in ("means", "quats", "scales", "opacities", "colors",
                                                              "log_uncertainties")}
        grads["exposure"] = exposure.grad.clone()
        grads["dR"] = torch.stack([p.dR.grad for p in poses])
        grads["dt"] = torch.stack([p.dt.grad for p in poses])
        res.append((float(total), float(pm), grads))
    (t0, p0, g0), (t1, p1, g1) = res
    assert abs(t0 - t1) < 1e-5 * max(1.0, abs(t0)) and abs(p0 - p1) < 1e-5 * max(1.0, abs(p0)), (t0, t1, p0, p1)
    for k in g0:
        scale = g0[k].abs().max().item() + 1e-12
        assert (g0[k] - g1[k]).abs().max().item() < 2e-3 * scale, (k, (g0[k] - g1[k]).abs().max().item(), scale)
        assert (g0[k] - g1[k]).abs().mean().item() < 1e-4 * scale, k


def test_fused_tracking_loss_matches_reference_formulation(dev):
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.losses import fused_tracking_loss
    from gslam_amd.primitives import Camera, PoseZhou
    from gslam_amd.synthetic import make_cameras, make_scene
    from gslam_amd.tracking import TrackingConfig, tracking_loss
    n, W, H = 3000, 200, 152
    sc = make_scene(n, 22)
    sc["scales"] = sc["scales"] + 0.6
    viewmats, Ks = make_cameras(1, W, H)
    gt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(4)).to(dev)
    splats = GaussianSplattingData.from_dict(sc, dev)
    res = []
    for fused in (False, True):
        pose = PoseZhou(viewmats[0].to(dev)).to(dev)
        exposure = torch.tensor([0.1, -0.05], device=dev).requires_grad_(True)
        out = splats([Camera(Ks[0].to(dev), H, W)], [pose], render_depth=True)
        if fused:
            loss = fused_tracking_loss(out, gt, exposure)
        else:
            rgb = out.rgbs[0] * exposure[0].exp() + exposure[1]
            loss = tracking_loss(TrackingConfig(), rgb, gt, out.betas[0], out.depthmaps[0], None)
        loss.backward()
        res.append((float(loss), pose.dR.grad.clone(), pose.dt.grad.clone(), exposure.grad.clone()))
    assert abs(res[0][0] - res[1][0]) < 1e-5 * abs(res[0][0])
    for a, b in zip(res[0][1:], res[1][1:]):
        assert (a - b).abs().max() < 2e-3 * a.abs().max() + 1e-9


def test_graphed_ba_step_equals_eager(dev):
    """HIP-graph replay of a whole BA iteration (capturable Adam, sync-free render) == eager iterations."""
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.mapping import BundleAdjuster, GraphedBundleAdjuster
    from gslam_amd.primitives import Camera, Frame, PoseZhou
    from gslam_amd.synthetic import make_cameras, make_scene
    n, W, H = 5000, 320, 240
    sc = make_scene(n, 31)
    sc["scales"] = sc["scales"] + 0.5
    viewmats, Ks = make_cameras(2, W, H)
    gt = torch.rand(2, H, W, 3, generator=torch.Generator().manual_seed(9)).to(dev)

    def build(capturable):
        splats = GaussianSplattingData.from_dict(sc, dev)
        window = [Frame(img=gt[i], timestamp=0.0, camera=Camera(Ks[i].to(dev), H, W),
                        pose=PoseZhou(viewmats[i].to(dev)).to(dev), gt_pose=viewmats[i].to(dev), index=i,
                        exposure_params=torch.zeros(2, device=dev)) for i in range(2)]
        return splats, window, BundleAdjuster(splats, capturable=capturable)

    sa, wa, ba_a = build(False)
    for _ in range(4):
        ta, _ = ba_a.step(wa)
    sb, wb, ba_b = build(True)
    gba = GraphedBundleAdjuster(ba_b, wb, warmup=2)      # 2 eager plan iterations; the capture itself does not execute
    for _ in range(2):
        tb, _ = gba.step()
    assert gba.capacity_ok() and gba.plan.graph.captured
    assert abs(float(ta) - float(tb)) < 1e-3 * abs(float(ta)) + 1e-6
    # float atomics in K9 make gradients differ in the last bits run to run, and Adam's m/sqrt(v) turns last-bit noise on
    # near-zero gradients into +-lr steps: compare in the mean (tight) and in the max (a few lr)
    for k in ("means", "quats", "scales", "opacities", "colors", "log_uncertainties"):
        a, b = getattr(sa, k), getattr(sb, k)
        assert (a - b).abs().mean() < 2e-5, (k, (a - b).abs().mean())
        assert (a - b).abs().max() < 0.2, (k, (a - b).abs().max())
    assert (wa[1].pose.dR - wb[1].pose.dR).abs().max() < 2e-3


def test_pose_batch_matches_reference_and_torch(dev):
    from gslam_amd.primitives import PoseZhou, pose_batch
    g = dict(np.load(os.path.join(GOLD, "pose_zhou.npz")))
    gen = torch.Generator().manual_seed(5)
    poses = []
    for i in range(4):
        Rt = torch.from_numpy(g["Rt"]).clone()
        Rt[:3, 3] += 0.1 * i
        p = PoseZhou(Rt.to(dev), is_learnable=(i != 2)).to(dev)
        with torch.no_grad():
            p.dR.copy_((torch.from_numpy(g["dR"]) + 0.01 * torch.randn(6, generator=gen)).to(dev))
            p.dt.copy_((torch.from_numpy(g["dt"]) + 0.01 * torch.randn(3, generator=gen)).to(dev))
        poses.append(p)
    w = torch.randn(4, 4, 4, generator=gen).to(dev)
    V = pose_batch(poses)
    (V * w).sum().backward()
    got = [(p.dR.grad.clone() if p.is_learnable else None, p.dt.grad.clone() if p.is_learnable else None) for p in poses]
    for p in poses:
        p.dR.grad = None
        p.dt.grad = None
    Vt = torch.stack([p() for p in poses])
    (Vt * w).sum().backward()
    assert (V - Vt).abs().max() < 1e-6
    for p, (gr, gt_) in zip(poses, got):
        if p.is_learnable:
            assert (p.dR.grad - gr).abs().max() < 1e-5 and (p.dt.grad - gt_).abs().max() < 1e-5
    # reference golden (first pose has the golden dR/dt only up to the added noise -> rebuild exactly)
    p0 = PoseZhou(torch.from_numpy(g["Rt"]).to(dev)).to(dev)
    with torch.no_grad():
        p0.dR.copy_(torch.from_numpy(g["dR"]).to(dev))
        p0.dt.copy_(torch.from_numpy(g["dt"]).to(dev))
    V0 = pose_batch([p0])
    np.testing.assert_allclose(_np(V0[0]), g["viewmat"], atol=1e-6)
    (V0[0] * torch.from_numpy(g["w"]).to(dev)).sum().backward()
    np.testing.assert_allclose(_np(p0.dR.grad), g["grad_dR"], atol=1e-5)
    np.testing.assert_allclose(_np(p0.dt.grad), g["grad_dt"], atol=1e-5)


def test_gslam_rasterization_packed_default(dev, oracle32):
    """rasterization(packed=True) - the signature's DEFAULT (gslam/rasterization.py:58) - returns the same images and the
    per-pair arrays packed over the visible pairs in flatten-id order, as gsplat's packed projection lays them out
    (rasterization.py:174-182); gradients through the packed means2d equal those of the dense call"""
    from gslam_amd.rasterization import rasterization
    n, c, W, H = 3000, 2, 320, 240
    sc, viewmats, Ks = _scene(n, 14, c, W, H)
    names = ("means", "quats", "scales", "opacities", "colors", "log_uncertainties")
    res = []
    for packed in (False, True):
        pa = {k: sc[k].clone().to(dev).requires_grad_(True) for k in names}
        kw = {} if packed else {"packed": False}                  # packed=True via the default
        out = rasterization(pa["means"], pa["quats"], pa["scales"], pa["opacities"], pa["colors"], viewmats.to(dev),
                            Ks.to(dev), W, H, render_mode="RGB+D", log_uncertainties=pa["log_uncertainties"],
                            backgrounds=torch.zeros(c, 3, device=dev), **kw)
        out.means2d.retain_grad()
        (out.rgbs.square().sum() + out.depthmaps.sum() + 0.1 * out.means2d.sum()).backward()
        res.append((out, pa))
    (d, pd), (p, pp) = res
    assert torch.equal(d.rgbs, p.rgbs) and torch.equal(d.alphas, p.alphas) and torch.equal(d.depthmaps, p.depthmaps)
    vis = d.radii > 0
    sel = torch.nonzero(vis.reshape(-1)).squeeze(1)
    nnz = int(sel.shape[0])
    assert 0 < nnz < c * n and p.camera_ids.shape == p.gaussian_ids.shape == (nnz,)
    assert torch.equal(p.camera_ids * n + p.gaussian_ids, sel)
    assert torch.equal(p.radii, d.radii.reshape(-1)[sel]) and torch.equal(p.means2d, d.means2d.reshape(-1, 2)[sel])
    assert torch.equal(p.depths, d.depths.reshape(-1)[sel]) and torch.equal(p.conics, d.conics.reshape(-1, 3)[sel])
    assert torch.equal(p.opacities, d.opacities.reshape(-1)[sel]) and torch.equal(p.n_touched, d.n_touched.reshape(-1)[sel])
    assert torch.equal(sel[p.flatten_ids.long()], d.flatten_ids.long())        # packed rows <-> flatten ids, same lists
    assert torch.equal(p.isect_ids, d.isect_ids) and torch.equal(p.isect_offsets, d.isect_offsets)
    assert p.means2d.grad.shape == (nnz, 2)
    assert float((p.means2d.grad - d.means2d.grad.reshape(-1, 2)[sel]).abs().max()) <= 1e-4 * float(d.means2d.grad.abs().max())
    for k in names:
        a, b = pp[k].grad, pd[k].grad
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()) + 1e-9, k


def test_sh_from_means_equals_direction_form(dev, oracle32):
    """the SH kernels that form the view directions in registers (means - camera centre) against the direction-array form
    and the oracle: colours bit-exact, gradients of coefficients / means / camera centres equal"""
    from gslam_amd import ops
    g = torch.Generator().manual_seed(3)
    c, n = 3, 4000
    means = torch.randn(n, 3, generator=g) * 2.0
    campos = torch.randn(c, 3, generator=g)
    coeffs = torch.randn(n, 16, 3, generator=g) * 0.3
    radii = (torch.rand(c, n, generator=g) > 0.25).to(torch.int32)
    w = torch.randn(c, n, 3, generator=g)
    for deg in (0, 2, 3):
        ma, ca, coa = (t.clone().to(dev).requires_grad_(True) for t in (means, campos, coeffs))
        col_a = ops.spherical_harmonics_from_means(deg, ma, ca, coa, masks=radii.to(dev))
        (col_a * w.to(dev)).sum().backward()
        mb, cb, cob = (t.clone().to(dev).requires_grad_(True) for t in (means, campos, coeffs))
        col_b = ops.spherical_harmonics(deg, mb[None] - cb[:, None], cob, masks=radii.to(dev))
        (col_b * w.to(dev)).sum().backward()
        assert torch.equal(col_a, col_b)
        ocol = oracle32.sh_fwd(deg, _np(means[None] - campos[:, None]), _np(coeffs), _np(radii))
        assert np.abs(_np(col_a) - ocol).max() < 2e-6
        assert float((coa.grad - cob.grad).abs().max()) <= 1e-6 * float(cob.grad.abs().max()) + 1e-9
        assert float((ma.grad - mb.grad).abs().max()) <= 1e-5 * float(mb.grad.abs().max()) + 1e-9
        if deg > 0:
            assert float(cb.grad.abs().max()) > 0
        assert float((ca.grad - cb.grad).abs().max()) <= 2e-4 * float(cb.grad.abs().max()) + 1e-7


def test_packed_operator_pipeline_equals_dense(dev):
    """SURVEY 8b surface (i) with packed=True end to end (gslam/rasterization.py:174-182, :261-272, :325-339): packed projection ->
    isect_tiles(packed, camera_ids, gaussian_ids) -> isect_offset_encode -> rasterize_to_pixels(packed) against the dense
    pipeline on the same scene: the same keys, the same lists (flatten ids name packed rows), bit-equal images and per-row
    touched counts, and the same gradients on the packed arrays' rows"""
    from gslam_amd import ops
    n, c, W, H = 3000, 3, 320, 240
    sc, viewmats, Ks = _scene(n, 21, c, W, H, dev)
    scales = torch.exp(sc["scales"] + 0.5)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    g = torch.Generator().manual_seed(3)
    colors = torch.rand(c, n, 4, generator=g).to(dev)
    opac = torch.rand(c, n, generator=g).to(dev) * 0.9 + 0.05
    bg = torch.rand(c, 4, generator=g).to(dev)
    # dense
    radii, m2d, dep, con, _ = ops.fully_fused_projection(sc["means"], None, sc["quats"], scales, viewmats, Ks, W, H)
    tpg, ids, flat = ops.isect_tiles(m2d, radii, dep, 16, tw, th, n_cameras=c)
    off = ops.isect_offset_encode(ids, c, tw, th)
    dm, dc, dcol, dop = (t.clone().requires_grad_(True) for t in (m2d, con, colors, opac))
    r_d, a_d, nt_d = ops.rasterize_to_pixels(dm, dc, dcol, dop, W, H, 16, off, flat, backgrounds=bg)
    # packed
    cam, gid, pr, pm, pd, pc, _ = ops.fully_fused_projection(sc["means"], None, sc["quats"], scales, viewmats, Ks, W, H,
                                                             packed=True)
    nnz = cam.numel()
    assert 0 < nnz < c * n
    ptpg, pids, pflat = ops.isect_tiles(pm, pr, pd, 16, tw, th, packed=True, n_cameras=c, camera_ids=cam, gaussian_ids=gid)
    poff = ops.isect_offset_encode(pids, c, tw, th)
    assert torch.equal(pids, ids) and torch.equal(poff, off)
    assert torch.equal(ptpg, tpg[cam, gid])
    assert torch.equal((cam * n + gid)[pflat.long()], flat.long())       # the packed rows the lists name are the same pairs
    pmr, pcr = pm.clone().requires_grad_(True), pc.clone().requires_grad_(True)
    pcol, pop = colors[cam, gid].clone().requires_grad_(True), opac[cam, gid].clone().requires_grad_(True)
    r_p, a_p, nt_p = ops.rasterize_to_pixels(pmr, pcr, pcol, pop, W, H, 16, poff, pflat, backgrounds=bg, packed=True)
    assert torch.equal(r_p, r_d) and torch.equal(a_p, a_d)
    assert torch.equal(nt_p, nt_d[cam, gid]) and tuple(nt_p.shape) == (nnz,)
    w = torch.randn(r_d.shape, generator=g).to(dev)
    (r_d * w).sum().backward()
    (r_p * w).sum().backward()
    for a, b in ((pmr.grad, dm.grad[cam, gid]), (pcr.grad, dc.grad[cam, gid]), (pcol.grad, dcol.grad[cam, gid]),
                 (pop.grad, dop.grad[cam, gid])):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7 * float(b.abs().max()))
    # unsorted camera ids (rows of two cameras interleaved) keep each camera's row order: same keys, lists name the same pairs
    rows = torch.argsort((torch.arange(nnz, device=dev) % 7) * nnz + torch.arange(nnz, device=dev), stable=True)
    _, sids, sflat = ops.isect_tiles(pm[rows], pr[rows], pd[rows], 16, tw, th, packed=True, n_cameras=c,
                                     camera_ids=cam[rows], gaussian_ids=gid[rows])
    assert torch.equal(sids, ids)
    same_depth_tie_free = (cam[rows] * n + gid[rows])[sflat.long()]
    assert torch.equal(torch.sort(same_depth_tie_free)[0], torch.sort(flat.long())[0])

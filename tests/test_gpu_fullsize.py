"""Parity at BASELINE.json's FULL sizes through size-independent properties (the CPU oracle would take minutes to
hours there): sortedness and counts of the tile lists, agreement between independent implementations (rocPRIM global
sort vs tile-binned sort; three generations of raster kernels), linearity of the compositing in the colours, bounds,
determinism.  Run with -m gpu."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _render_inputs(n, c, W, H, dev, seed=0, scale_shift=0.0):
    from gslam_amd import ops
    from gslam_amd.synthetic import make_cameras, make_scene
    sc = {k: v.to(dev) for k, v in make_scene(n, seed).items()}
    viewmats, Ks = make_cameras(c, W, H)
    viewmats, Ks = viewmats.to(dev), Ks.to(dev)
    scales = torch.exp(sc["scales"] + scale_shift)
    radii, m2d, dep, con, _ = ops.fully_fused_projection(sc["means"], None, sc["quats"], scales, viewmats, Ks, W, H)
    return sc, radii, m2d, dep, con


def _check_tile_lists(radii, m2d, dep, ids, flat, off, c, tw, th):
    """properties that pin the integer contract without an oracle"""
    M = flat.shape[0]
    n_tiles = tw * th
    tnb = int(n_tiles).bit_length()
    # 1. keys are non-decreasing, and strictly increasing in (key, flatten id): the stable order
    assert bool((ids[1:] >= ids[:-1]).all())
    same = ids[1:] == ids[:-1]
    assert bool((flat[1:][same] > flat[:-1][same]).all())
    # 2. every key is consistent with its payload: depth bits and camera of the Gaussian it names
    dbits = dep.reshape(-1).view(torch.int32)[flat.long()].long() & 0xFFFFFFFF
    assert bool(((ids & 0xFFFFFFFF) == dbits).all())
    N = radii.shape[1]
    cam = flat.long() // N
    assert bool(((ids >> (32 + tnb)) == cam).all())
    # 3. per-tile counts == rectangle coverage recomputed independently (2-D difference grid in torch, fp32 like K3)
    tile_of = ((ids >> 32) & ((1 << tnb) - 1)) + cam * n_tiles
    counts = torch.bincount(tile_of, minlength=c * n_tiles)
    vis = radii.reshape(-1) > 0
    mx, my = m2d.reshape(-1, 2)[vis, 0], m2d.reshape(-1, 2)[vis, 1]
    r = radii.reshape(-1)[vis].float()
    camv = (torch.nonzero(vis).squeeze(1) // N)
    x0 = torch.floor(mx / 16.0 - r / 16.0).clamp(0, tw).long()
    x1 = torch.ceil(mx / 16.0 + r / 16.0).clamp(0, tw).long()
    y0 = torch.floor(my / 16.0 - r / 16.0).clamp(0, th).long()
    y1 = torch.ceil(my / 16.0 + r / 16.0).clamp(0, th).long()
    grid = torch.zeros(c, th + 1, tw + 1, dtype=torch.int64, device=radii.device)
    for yy, xx, sgn in ((y0, x0, 1), (y0, x1, -1), (y1, x0, -1), (y1, x1, 1)):
        grid.index_put_((camv, yy, xx), torch.full_like(yy, sgn), accumulate=True)
    cover = grid.cumsum(1).cumsum(2)[:, :th, :tw].reshape(-1)
    assert torch.equal(cover, counts)
    assert int(cover.sum()) == M
    # 4. offsets are the exclusive scan of the counts
    assert torch.equal(off.reshape(-1).long(), torch.cumsum(counts, 0) - counts)


@pytest.mark.parametrize("n,c,W,H", [(500_000, 1, 640, 480), (2_000_000, 1, 640, 480), (1_000_000, 8, 640, 480),
                                      (5_000_000, 1, 1920, 1080)])
def test_tile_lists_full_size(dev, n, c, W, H):
    """configs 3 / 4 (per-GPU shard and a full 8-camera window) / 5 of BASELINE.json"""
    from gslam_amd import ops
    sc, radii, m2d, dep, con = _render_inputs(n, c, W, H, dev)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    tpg, ids, flat = ops.isect_tiles(m2d, radii, dep, 16, tw, th, n_cameras=c)          # tile-binned sort
    off = ops.isect_offset_encode(ids, c, tw, th)
    assert int(tpg.sum()) == ids.shape[0] > 0
    _check_tile_lists(radii, m2d, dep, ids, flat, off, c, tw, th)
    if n <= 1_000_000:     # second implementation: unsorted emission + a device-wide stable sort of the 64-bit keys
        _, ids_u, flat_u = ops.isect_tiles(m2d, radii, dep, 16, tw, th, sort=False, n_cameras=c)
        assert ids_u.shape[0] == ids.shape[0]
        ids1, order = torch.sort(ids_u, stable=True)
        assert torch.equal(ids1, ids) and torch.equal(flat_u[order], flat)
        del ids_u, flat_u, ids1, order
    # sync-free path (capacity buffers, offsets with T+1 entries) gives the same lists
    cap = int(ids.shape[0] * 1.5) + 4096
    flat_buf = torch.empty(cap, dtype=torch.int32, device=dev)
    off1, M_dev, status = ops.isect_bin_sort(m2d, radii, dep, tw, th, cap, None, flat_buf)
    assert int(M_dev) == ids.shape[0] and int(status) == 0
    assert torch.equal(flat_buf[:ids.shape[0]], flat) and torch.equal(off1[:-1], off.reshape(-1))
    # overflow is reported, never written past the capacity
    small = ids.shape[0] // 2
    flat_small = torch.full((small + 64,), -7, dtype=torch.int32, device=dev)
    _, M2, st2 = ops.isect_bin_sort(m2d, radii, dep, tw, th, small, None, flat_small)
    assert int(st2) & 1 and int(M2) == ids.shape[0] and bool((flat_small[small:] == -7).all())


def test_hot_tile_exceeds_lds_window(dev, oracle32):
    """one tile with ~30x the average load: chunks sorted in LDS + merge levels in global memory; vs the oracle"""
    from gslam_amd import ops
    g = torch.Generator().manual_seed(5)
    n = 30000
    m2d = torch.rand(1, n, 2, generator=g) * torch.tensor([640.0, 480.0])
    m2d[0, :20000] = torch.tensor([200.0, 200.0]) + torch.rand(20000, 2, generator=g) * 4.0
    radii = torch.full((1, n), 3, dtype=torch.int32)
    dep = torch.rand(1, n, generator=g) * 5 + 0.5
    dep[0, 100:200] = dep[0, 100]                                    # ties inside the hot tile
    tpg, ids, flat = ops.isect_tiles(m2d.to(dev), radii.to(dev), dep.to(dev), 16, 40, 30)
    otpg, oids, oflat = oracle32.isect_tiles(m2d.numpy(), radii.numpy(), dep.numpy(), 16, 40, 30)
    assert np.array_equal(ids.cpu().numpy(), oids) and np.array_equal(flat.cpu().numpy(), oflat)
    assert np.bincount((oids >> 32) & 2047).max() > 8192            # really beyond the largest LDS window


def test_hot_tile_degenerate_depths(dev, oracle32):
    """large tiles whose depths defeat the counting sort (thousands of equal depths, or all but one key in one bucket):
    flagged on the device and sorted by the merge launch; mixed with ordinary large tiles in the same call"""
    from gslam_amd import ops
    g = torch.Generator().manual_seed(6)
    n = 24000
    m2d = torch.rand(1, n, 2, generator=g) * torch.tensor([640.0, 480.0])
    m2d[0, :6000] = torch.tensor([200.0, 200.0]) + torch.rand(6000, 2, generator=g) * 4.0     # equal depths
    m2d[0, 6000:12000] = torch.tensor([400.0, 100.0]) + torch.rand(6000, 2, generator=g) * 4.0  # one far outlier
    m2d[0, 12000:18000] = torch.tensor([100.0, 400.0]) + torch.rand(6000, 2, generator=g) * 4.0  # well spread
    radii = torch.full((1, n), 3, dtype=torch.int32)
    dep = torch.rand(1, n, generator=g) * 5 + 0.5
    dep[0, :6000] = 2.5
    dep[0, 6000:12000] = 1.0 + torch.rand(6000, generator=g) * 1e-3
    dep[0, 6000] = 5000.0
    tpg, ids, flat = ops.isect_tiles(m2d.to(dev), radii.to(dev), dep.to(dev), 16, 40, 30)
    otpg, oids, oflat = oracle32.isect_tiles(m2d.numpy(), radii.numpy(), dep.numpy(), 16, 40, 30)
    assert np.array_equal(ids.cpu().numpy(), oids) and np.array_equal(flat.cpu().numpy(), oflat)
    assert np.sort(np.bincount((oids >> 32) & 2047))[-3] > 2048      # three tiles beyond the LDS counting sort


@pytest.mark.parametrize("n,c,W,H,ch", [(500_000, 1, 640, 480, 5), (200_000, 8, 640, 480, 5),
                                         (2_000_000, 1, 640, 480, 5), (5_000_000, 1, 1920, 1080, 3)])
def test_raster_properties_full_size(dev, n, c, W, H, ch):
    """configs 3 / 4 (one camera of the 2 M window) / 5: forward properties AND the backward cross-check at every size"""
    from gslam_amd import ops
    sc, radii, m2d, dep, con = _render_inputs(n, c, W, H, dev)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    _, ids, flat = ops.isect_tiles(m2d, radii, dep, 16, tw, th, n_cameras=c)
    off = ops.isect_offset_encode(ids, c, tw, th)
    g = torch.Generator().manual_seed(1)
    c1 = torch.rand(c, n, ch, generator=g).to(dev)
    c2 = torch.rand(c, n, ch, generator=g).to(dev)
    opac = (torch.rand(c, n, generator=g) * 0.8 + 0.1).to(dev)

    def render(cols, grad=False, absgrad=False):
        ins = [m2d.clone().requires_grad_(grad), con.clone().requires_grad_(grad), cols.clone().requires_grad_(grad),
               opac.clone().requires_grad_(grad)]
        r, a, nt = ops.rasterize_to_pixels(ins[0], ins[1], ins[2], ins[3], W, H, 16, off, flat, absgrad=absgrad)
        grads = None
        if grad:
            wts = torch.linspace(0.5, 1.5, ch, device=dev)
            ((r * wts).sum() + 0.3 * a.sum()).backward()
            grads = [t.grad for t in ins]
        return r, a, nt, grads

    r1, a1, nt1, _ = render(c1)
    r1b, a1b, nt1b, _ = render(c1)
    assert torch.equal(r1, r1b) and torch.equal(a1, a1b) and torch.equal(nt1, nt1b)      # forward is deterministic
    assert bool(torch.isfinite(r1).all()) and float(a1.min()) >= 0.0 and float(a1.max()) <= 1.0
    assert float(r1.min()) >= 0.0 and float(r1.max()) <= 1.0 + 1e-5                        # convex combination of colours
    # linearity in the colours at fixed geometry (zero background): R(2 c1 + 3 c2) = 2 R(c1) + 3 R(c2)
    r2, _, _, _ = render(c2)
    r12, a12, _, _ = render(2.0 * c1 + 3.0 * c2)
    assert torch.equal(a12, a1)
    assert float((r12 - (2.0 * r1 + 3.0 * r2)).abs().max()) < 5e-5
    # two backward kernels with different work decompositions agree at full size (2 M and 5 M / 1080p included): the
    # launch-selected one (quadrant wavefronts or the full-chip two-pixel kernel, register reduce-scatter, plain stores) and
    # the absgrad kernel (two pixels per lane, LDS row sums)
    del r2, r12, a12
    ref = render(c1, grad=True)
    got = render(c1, grad=True, absgrad=True)
    assert torch.equal(got[0], ref[0]) and torch.equal(got[2], ref[2])                    # same forward kernel
    for gg, gr in zip(got[3], ref[3]):
        assert bool(torch.isfinite(gg).all())
        scale = float(gr.abs().max()) + 1e-12
        assert float((gg - gr).abs().max()) / scale < 5e-3
        assert float((gg - gr).abs().mean()) / (float(gr.abs().mean()) + 1e-12) < 1e-4
    # linearity of the backward in the upstream gradient: d/d colours under weights w equals alpha T per pixel, independent of
    # the colours themselves - the colour gradient of R(c1) and of R(c2) under the same weights are the same array
    ref2 = render(c2, grad=True)
    assert float((ref2[3][2] - ref[3][2]).abs().max()) <= 1e-4 * float(ref[3][2].abs().max()) + 1e-9


def test_full_pipeline_500k_sync_free_equals_reference_shaped(dev):
    """config 3: fused gslam rasterization at 500 k; sync-free path == reference-shaped path (M read back)"""
    from gslam_amd.rasterization import IsectCapacity, rasterization
    from gslam_amd.synthetic import make_cameras, make_scene
    n, W, H = 500_000, 640, 480
    sc = {k: v.to(dev) for k, v in make_scene(n, 0).items()}
    viewmats, Ks = make_cameras(1, W, H)
    viewmats, Ks = viewmats.to(dev), Ks.to(dev)

    def run(capacity):
        return rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["colors"], viewmats, Ks, W, H,
                             packed=False, render_mode="RGB+D", log_uncertainties=sc["log_uncertainties"],
                             backgrounds=torch.zeros(1, 3, device=dev), capacity=capacity)
    cap = IsectCapacity(dev)
    a = run(cap)
    a2 = run(cap)                                   # second render of the shape: no probe, no read-back
    assert cap.validate() and cap.last_M == int(a.tiles_per_gauss.sum())
    assert torch.equal(a.rgbs, a2.rgbs)
    b = run(None)                                   # default: exact sizes from one read-back of M, like the reference
    assert torch.equal(a.flatten_ids, b.flatten_ids) and torch.equal(a.isect_ids, b.isect_ids)
    assert torch.equal(a.isect_offsets, b.isect_offsets) and torch.equal(a.radii, b.radii)
    assert torch.equal(a.rgbs, b.rgbs) and torch.equal(a.n_touched, b.n_touched)
    assert a.n_touched.dtype == torch.int64 and int(a.flatten_ids.shape[0]) == int(a.tiles_per_gauss.sum())


def test_sh3_1080p_5m(dev):
    """config 5: 5 M Gaussians, SH degree 3, 1920x1080 through the gsplat-shaped entry point"""
    from gslam_amd.rendering import rasterization
    from gslam_amd.synthetic import make_cameras, make_scene
    n, W, H = 5_000_000, 1920, 1080
    sc = {k: v.to(dev) for k, v in make_scene(n, 0, sh_degree=3).items()}
    viewmats, Ks = make_cameras(1, W, H)
    colors, alphas, meta = rasterization(sc["means"], sc["quats"], torch.exp(sc["scales"]),
                                         torch.sigmoid(sc["opacities"]), sc["sh_coeffs"], viewmats.to(dev), Ks.to(dev),
                                         W, H, sh_degree=3, packed=False)
    assert colors.shape == (1, H, W, 3) and alphas.shape == (1, H, W, 1)
    assert bool(torch.isfinite(colors).all()) and float(colors.min()) >= 0.0
    assert float(alphas.max()) <= 1.0 and float(alphas.mean()) > 0.5
    assert meta["tile_width"] == 120 and meta["tile_height"] == 68          # ragged last tile row (1080 / 16 = 67.5)
    assert int(meta["tiles_per_gauss"].sum()) == meta["flatten_ids"].shape[0]
    # degree-0-only evaluation equals 0.5 + C0 * dc where the splat is visible
    dc_only = torch.zeros_like(sc["sh_coeffs"])
    dc_only[:, 0] = sc["sh_coeffs"][:, 0]
    c0, _, _ = rasterization(sc["means"], sc["quats"], torch.exp(sc["scales"]), torch.sigmoid(sc["opacities"]), dc_only,
                             viewmats.to(dev), Ks.to(dev), W, H, sh_degree=3, packed=False)
    flat_col = (0.5 + 0.2820947917738781 * sc["sh_coeffs"][:, 0]).clamp(min=0)
    c0b, _, _ = rasterization(sc["means"], sc["quats"], torch.exp(sc["scales"]), torch.sigmoid(sc["opacities"]), flat_col,
                              viewmats.to(dev), Ks.to(dev), W, H, packed=False)
    assert float((c0 - c0b).abs().max()) < 1e-5


def test_gsplat_entry_point_sync_free_equals_default(dev):
    """gsplat.rendering.rasterization-shaped entry point (pipeline.py:106-116) with a caller-owned IsectCapacity: no read-back
    of the intersection count, same render and gradients as the default (exact-size) path, SH degree 3"""
    from gslam_amd.rasterization import IsectCapacity
    from gslam_amd.rendering import rasterization
    from gslam_amd.synthetic import make_cameras, make_scene
    n, W, H = 200_000, 640, 480
    sc = {k: v.to(dev) for k, v in make_scene(n, 2, sh_degree=3).items()}
    viewmats, Ks = make_cameras(2, W, H)
    viewmats, Ks = viewmats.to(dev), Ks.to(dev)
    cap = IsectCapacity(dev)
    res = []
    for capacity in (None, cap, cap):
        coeffs = sc["sh_coeffs"].clone().requires_grad_(True)
        means = sc["means"].clone().requires_grad_(True)
        colors, alphas, meta = rasterization(means, sc["quats"], torch.exp(sc["scales"]), torch.sigmoid(sc["opacities"]),
                                             coeffs, viewmats, Ks, W, H, sh_degree=3, packed=False, capacity=capacity)
        (colors.sum() + alphas.sum()).backward()
        res.append((colors.detach(), alphas.detach(), coeffs.grad, means.grad, meta))
    assert cap.validate() and cap.last_M == int(res[0][4]["flatten_ids"].shape[0])
    assert res[1][4]["isect_ids"] is None and int(res[1][4]["n_isects"]) == cap.last_M
    for a, b in zip(res[0][:2], res[1][:2]):
        assert torch.equal(a, b)
    for k in (2, 3):
        scale = float(res[0][k].abs().max())
        assert float((res[0][k] - res[1][k]).abs().max()) < 2e-3 * scale           # float atomics in the backward
    assert torch.equal(res[1][0], res[2][0])


def test_ba_step_2m_window8_graph_equals_eager(dev):
    """configs[3] at FULL size on one GPU: one bundle-adjustment iteration over 2 M Gaussians and an 8-keyframe 640x480
    window (gslam/backend.py:260-359) replayed from the HIP graph of gslam_amd.plan.MappingStep, against the eager
    autograd-shaped step (gslam_amd.mapping.BundleAdjuster.render_backward / update) on identical inputs: loss values,
    every gradient of the step bucket (mean and max), the visible-camera counts exactly, the tile lists exactly, and the
    parameters after the update."""
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.mapping import BundleAdjuster, MapConfig
    from gslam_amd.primitives import Camera, Frame, PoseZhou
    from gslam_amd.synthetic import make_cameras, make_scene
    n, C, W, H = 2_000_000, 8, 640, 480
    sc = make_scene(n, 0)
    viewmats, Ks = make_cameras(C, W, H)
    gt = torch.rand(C, H, W, 3, generator=torch.Generator().manual_seed(77)).to(dev)

    def build(capturable):
        splats = GaussianSplattingData.from_dict({k: v.clone() for k, v in sc.items()}, dev)
        window = [Frame(img=gt[i], timestamp=0.0, camera=Camera(Ks[i].to(dev), H, W),
                        pose=PoseZhou(viewmats[i].to(dev)).to(dev), gt_pose=viewmats[i].to(dev), index=i,
                        exposure_params=torch.zeros(2, device=dev)) for i in range(C)]
        return splats, window, BundleAdjuster(splats, MapConfig(), capturable=capturable)

    names = ("means", "quats", "scales", "opacities", "colors", "log_uncertainties")
    sa, wa, ba_a = build(False)
    ta, pa = ba_a.render_backward(wa)
    grads_a = {k: getattr(sa, k).grad.detach().clone() for k in names}
    vis_a = ba_a._vis_count.clone()
    flat_a = ba_a.last_outputs.flatten_ids.clone()
    off_a = ba_a.last_outputs.isect_offsets.clone()
    pose_g_a = [(f.pose.dR.grad.clone(), f.pose.dt.grad.clone()) for f in wa]
    ba_a.update()
    ba_a.last_outputs = None
    torch.cuda.synchronize()

    sb, wb, ba_b = build(True)
    # (the tile lists are compared entry for entry below: the plan lists the reference's 3-sigma squares here - through the packed
    # rectangles of the projection, RECT_LISTS - not the tight ones it uses by default, which drop what no pixel can see)
    from gslam_amd.plan import RenderPlan
    tight_default = RenderPlan.TIGHT_LISTS
    RenderPlan.TIGHT_LISTS = False
    try:
        plan = ba_b.plan(wb)
    finally:
        RenderPlan.TIGHT_LISTS = tight_default
    assert plan.r.rect_lists and not plan.r.tight_lists
    plan.prepare()                       # capacity probe + one eager render / loss / backward WITHOUT update + capture
    assert plan.graph.captured
    tb, pb = plan.step()                 # ONE graph launch: the whole iteration
    torch.cuda.synchronize()
    assert plan.capacity_ok()
    M = int(plan.r.M_dev.item())
    assert M == flat_a.shape[0] > 30_000_000, M                                    # ~48 M intersections
    assert torch.equal(plan.r.flat[:M], flat_a)                                    # bit-exact tile lists at 2 M x 8
    assert torch.equal(plan.r.offsets[:-1].view(off_a.shape), off_a)
    assert torch.equal(plan.r.vis_count, vis_a.to(plan.r.vis_count.dtype))         # visible-camera counts: exact
    assert abs(float(tb) - float(ta)) < 2e-5 * abs(float(ta)), (float(tb), float(ta))
    assert abs(float(pb) - float(pa)) < 2e-5 * abs(float(pa)), (float(pb), float(pa))
    for k in names:
        ga, gb = grads_a[k], plan.grad_views[k]
        scale = float(ga.abs().max()) + 1e-20
        assert float((ga - gb).abs().max()) / scale < 5e-3, (k, float((ga - gb).abs().max()) / scale)
        assert float((ga - gb).abs().mean()) / (float(ga.abs().mean()) + 1e-20) < 2e-4, k
        assert float(ga.abs().mean()) > 0.0
    for i in range(C):
        for a, b in zip(pose_g_a[i], (plan.g_dR[i], plan.g_dt[i])):
            assert float((a - b).abs().max()) <= 2e-3 * float(a.abs().max()) + 1e-9, i
    # the update (six splat Adams + pose Adam + opacity decay in one launch, device step counters) against the eager one:
    # Adam's m / sqrt(v) turns last-bit atomic noise on near-zero gradients into +-lr steps: tight in the mean, a few lr in the max
    for k in names:
        a, b = getattr(sa, k).detach(), getattr(sb, k).detach()
        assert float((a - b).abs().mean()) < 2e-5, (k, float((a - b).abs().mean()))
        assert float((a - b).abs().max()) < 0.06, (k, float((a - b).abs().max()))
    assert float((wa[3].pose.dR - wb[3].pose.dR).abs().max()) < 2e-3


def _crop_vs_oracle(out, sc, cam, tx0, ty0, oracle32, nt=4):
    """A (16 nt)^2-pixel crop of camera ``cam`` of a gslam rasterization() output against the CPU oracle's rasteriser run over
    the crop's nt x nt tiles only: the crop is posed as an image of its own (means2d shifted to the crop's origin, the tiles'
    segments of the GPU's sorted list as its tile lists) - per-Gaussian inputs of the oracle are the GPU's projection outputs
    (those are held to the oracle bit for bit at the baseline sizes, tests/test_gpu_baseline_sizes.py).  -> L1 per pixel."""
    N = sc["means"].shape[0]
    off = out.isect_offsets[cam].cpu().numpy()                     # [tile_h, tile_w]
    flat = out.flatten_ids
    M = int(flat.shape[0])
    offs_flat = np.concatenate([out.isect_offsets.reshape(-1).cpu().numpy(), [M]])
    th, tw = off.shape
    segs, sub_off, run = [], np.zeros((1, nt, nt), np.int32), 0
    for j in range(nt):
        for i in range(nt):
            t = (cam * th + ty0 + j) * tw + tx0 + i
            lo, hi = int(offs_flat[t]), int(offs_flat[t + 1])
            sub_off[0, j, i] = run
            segs.append(flat[lo:hi].cpu().numpy().astype(np.int64) - cam * N)
            run += hi - lo
    ids = np.concatenate(segs).astype(np.int32)
    assert ids.size > 0 and ids.min() >= 0 and ids.max() < N
    used = np.unique(ids)                                           # compact the per-Gaussian inputs to the listed ones
    remap = np.zeros(N, np.int32)
    remap[used] = np.arange(used.size, dtype=np.int32)
    ut = torch.from_numpy(used).to(out.means2d.device)
    m2 = out.means2d[cam][ut].detach().cpu().numpy() - np.array([tx0 * 16.0, ty0 * 16.0], np.float32)
    con = out.conics[cam][ut].detach().cpu().numpy()
    op = out.opacities[cam][ut].detach().cpu().numpy() if out.opacities.dim() == 2 else torch.sigmoid(sc["opacities"][ut]).cpu().numpy()
    cols = torch.sigmoid(sc["colors"][ut]).cpu().numpy()
    depth = out.depths[cam][ut].detach().cpu().numpy()
    betas = np.maximum(np.exp(sc["log_uncertainties"][ut].cpu().numpy()), np.float32(0.01))
    packed = np.concatenate([cols, depth[:, None], betas[:, None]], -1)[None].astype(np.float32)
    bg = np.array([[0.0, 0.0, 0.0, 0.0, np.e]], np.float32)
    S = 16 * nt
    ren, alp, _, _ = oracle32.raster_fwd(m2[None], con[None], packed, op[None], bg, S, S, 16, sub_off, remap[ids])
    x0, y0 = tx0 * 16, ty0 * 16
    got = out.rgbs[cam, y0:y0 + S, x0:x0 + S].detach().cpu().numpy()
    l1 = float(np.abs(got - ren[0, ..., :3]).mean())
    l1d = float(np.abs(out.depthmaps[cam, y0:y0 + S, x0:x0 + S].detach().cpu().numpy() - ren[0, ..., 3]).mean())
    la = float(np.abs(out.alphas[cam, y0:y0 + S, x0:x0 + S, 0].detach().cpu().numpy() - alp[0, ..., 0]).mean())
    return l1, l1d, la, ids.size


def test_crops_of_the_two_largest_configs_against_the_live_oracle(dev, oracle32):
    """the two largest BASELINE configs were only self-compared.  Here a 64x64-pixel crop of each - 5 M
    Gaussians at 1920x1080 (configs[4]'s size through gslam's rasterization()), and one keyframe of 2 M x 8 at 640x480
    (configs[3]) - is rendered again by the CPU oracle from the tiles' own lists: 1e-4 L1 per pixel, the bar of the path."""
    from gslam_amd.rasterization import rasterization
    from gslam_amd.synthetic import make_cameras, make_scene
    for n, C, W, H, cam, tx0, ty0 in ((5_000_000, 1, 1920, 1080, 0, 58, 32), (2_000_000, 8, 640, 480, 5, 18, 13)):
        sc = {k: v.to(dev) for k, v in make_scene(n, 0).items()}
        viewmats, Ks = make_cameras(C, W, H)
        out = rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["colors"], viewmats.to(dev),
                            Ks.to(dev), W, H, packed=False, render_mode="RGB+D", log_uncertainties=sc["log_uncertainties"],
                            backgrounds=torch.zeros(C, 3, device=dev))
        l1, l1d, la, n_ent = _crop_vs_oracle(out, sc, cam, tx0, ty0, oracle32)
        print(f"crop of {n} x {C} @ {W}x{H}: {n_ent} list entries in 16 tiles, L1 rgb {l1:.2e} depth {l1d:.2e} alpha {la:.2e}")
        assert n_ent > 16 * 50
        assert l1 < 1e-4 and la < 1e-4 and l1d < 1e-3, (n, l1, l1d, la)
        del out, sc
        torch.cuda.empty_cache()


@pytest.mark.parametrize("n,left_share,depth_mode", [(220_000, 0.5, "spread"), (70_000, 0.5, "ties"), (35_000, 0.75, "spread"),
                                                     (70_000, 0.5, "piles"), (36_000, 0.75, "piles")])
def test_deep_tile_lists_one_pass_sort(dev, oracle32, n, left_share, depth_mode):
    """tile_sort_deep_kernel (lists of more than 2048 keys, chosen by the list capacity per tile): a 64x48 image is 12 tiles, so
    a few ten thousand small Gaussians give lists of 3000-35000 keys - the 12160-key kernel with one window (70 k), with
    three depth windows that re-read the segment (220 k), the 5376-key kernel with lists on both sides of its window (35 k,
    three quarters of them in the left half of the image), runs of equal depths inside a bucket ("ties") and piles of equal
    depths that send a list to the merge sort in half the window ("piles") - against the oracle, bit for bit"""
    from gslam_amd import ops
    g = torch.Generator().manual_seed(11)
    W, H, tw, th = 64, 48, 4, 3
    m2d = torch.rand(1, n, 2, generator=g) * torch.tensor([float(W), float(H)])
    n_left = int(n * left_share)
    m2d[0, :n_left, 0] *= 0.5                                        # that share of the Gaussians in the left half
    m2d[0, n_left:, 0] = 0.5 * W + 0.5 * m2d[0, n_left:, 0]
    radii = torch.full((1, n), 3, dtype=torch.int32)
    dep = torch.rand(1, n, generator=g) * 5 + 0.5
    if depth_mode == "ties":
        dep[0, 1000:1400] = dep[0, 1000]                             # 400 equal depths: one bucket, ranked by id
        dep[0, 5000:5040] = dep[0, 5000]
    elif depth_mode == "piles":
        dep[0, : n // 2] = 2.5                                       # half of every list at one depth: no depth window can cut it
    tpg, ids, flat = ops.isect_tiles(m2d.to(dev), radii.to(dev), dep.to(dev), 16, tw, th)
    otpg, oids, oflat = oracle32.isect_tiles(m2d.numpy(), radii.numpy(), dep.numpy(), 16, tw, th)
    per_tile = np.bincount(((oids >> 32) & 15).astype(np.int64), minlength=12)
    cap_per_tile = len(oids) // 12
    if n >= 70_000:
        assert cap_per_tile > 5500                                   # the 12160-key kernel
        assert (per_tile.max() > 12160) == (n >= 220_000)
    else:
        assert 3500 < cap_per_tile <= 5500                           # the 5376-key kernel
        assert per_tile.max() > 5376 > per_tile.min()
    assert np.array_equal(tpg.cpu().numpy(), otpg)
    assert np.array_equal(ids.cpu().numpy(), oids) and np.array_equal(flat.cpu().numpy(), oflat)

"""One bundle-adjustment iteration end to end: the fused step (one loss-finish launch, isotropic term added into
scales.grad in place, means2d.grad as a view of the gradient records, splat + pose Adam and the opacity decay in one
launch, gradient records cleared by the projection kernel) against the reference's own formulation written in torch ops
(mapping.mapping_loss: gslam/backend.py:273-318) and torch.optim.Adam.  Run with -m gpu."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _window(dev, n=4000, W=320, H=240, cams=2, seed=33):
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.primitives import Camera, Frame, PoseZhou
    from gslam_amd.synthetic import make_cameras, make_scene
    sc = make_scene(n, seed)
    sc["scales"] = sc["scales"] + 0.5
    viewmats, Ks = make_cameras(cams, W, H)
    gt = torch.rand(cams, H, W, 3, generator=torch.Generator().manual_seed(seed + 1)).to(dev)
    splats = GaussianSplattingData.from_dict({k: v.clone() for k, v in sc.items()}, dev)
    window = [Frame(img=gt[i], timestamp=0.0, camera=Camera(Ks[i].to(dev), H, W),
                    pose=PoseZhou(viewmats[i].to(dev)).to(dev), gt_pose=viewmats[i].to(dev), index=i,
                    exposure_params=torch.zeros(2, device=dev, requires_grad=True)) for i in range(cams)]
    return splats, window


PARAMS = ("means", "quats", "scales", "opacities", "colors", "log_uncertainties")


def _close(a, b, rel, name):
    scale = float(b.abs().max()) + 1e-12
    assert float((a - b).abs().max()) / scale < rel, (name, float((a - b).abs().max()) / scale)


def test_fused_ba_iteration_equals_torch_formulation(dev):
    from gslam_amd.mapping import BundleAdjuster
    sa, wa = _window(dev)
    sb, wb = _window(dev)
    ba_f = BundleAdjuster(sa, fused_loss=True)
    ba_t = BundleAdjuster(sb, fused_loss=False)
    tf, pf = ba_f.render_backward(wa)
    tt, pt = ba_t.render_backward(wb)
    assert abs(float(tf) - float(tt)) < 2e-5 * abs(float(tt)) and abs(float(pf) - float(pt)) < 2e-5 * abs(float(pt))
    for k in PARAMS:
        _close(getattr(sa, k).grad, getattr(sb, k).grad, 2e-3, k)
    for fa, fb in zip(wa, wb):
        _close(fa.pose.dR.grad, fb.pose.dR.grad, 2e-3, "dR")
        _close(fa.pose.dt.grad, fb.pose.dt.grad, 2e-3, "dt")
        _close(fa.exposure_params.grad, fb.exposure_params.grad, 2e-3, "exposure")
    # backend.py:326: the image-plane gradient of every (camera, Gaussian), here a view of the gradient records
    ga, gb = ba_f.last_outputs.means2d.grad, ba_t.last_outputs.means2d.grad
    assert ga.shape == gb.shape == (2, sa.means.shape[0], 2)
    _close(ga, gb, 2e-3, "means2d.grad")

    # the update: fused multi-tensor Adam with the opacity decay riding in it == torch.optim.Adam per parameter
    # (backend.py:565-602 learning rates) followed by the masked multiply of backend.py:356-359
    conf = ba_f.conf
    from gslam_amd.mapping import SPLAT_LRS
    ref = {k: getattr(sa, k).detach().clone().requires_grad_(True) for k in PARAMS}
    grads = {k: getattr(sa, k).grad.detach().clone() for k in PARAMS}
    opts = [torch.optim.Adam([ref[name]], lr=getattr(conf, lr)) for name, lr in SPLAT_LRS]
    for k in PARAMS:
        ref[k].grad = grads[k]
    for o in opts:
        o.step()
    vis = (ba_f.last_outputs.radii > 0).sum(dim=0)
    with torch.no_grad():
        ref["opacities"][vis > 1] *= conf.opacity_decay
    assert int((vis > 1).sum()) > 100 and int((vis <= 1).sum()) > 100          # both branches of the mask are exercised
    pose_before = wa[1].pose.dR.detach().clone()
    ba_f.update()
    for k in PARAMS:
        assert float((getattr(sa, k).detach() - ref[k].detach()).abs().max()) < 2e-6, k
    assert float((wa[1].pose.dR.detach() - pose_before).abs().max()) > 0       # the pose Adam ran in the same launch


def test_pose_only_projection_backward_equals_full(dev):
    """tracking (frozen map) takes the pose-only projection backward: same pose gradient as with the map learnable"""
    sa, wa = _window(dev, cams=1)
    sb, wb = _window(dev, cams=1)
    for k in PARAMS:
        getattr(sb, k).requires_grad_(False)
    outs = []
    for s, w in ((sa, wa), (sb, wb)):
        out = s([w[0].camera], [w[0].pose], render_depth=True)
        loss = (out.rgbs - w[0].img[None]).square().mean() + 0.1 * out.depthmaps.mean() + 0.05 * out.betas.mean()
        loss.backward()
        outs.append((w[0].pose.dR.grad.clone(), w[0].pose.dt.grad.clone()))
    assert sb.means.grad is None and sa.means.grad is not None
    _close(outs[1][0], outs[0][0], 1e-4, "dR")
    _close(outs[1][1], outs[0][1], 1e-4, "dt")


def test_tile_launch_order_is_a_permutation_and_does_not_change_results(dev):
    from gslam_amd import ops
    from gslam_amd.synthetic import make_cameras, make_scene
    n, C, W, H = 20000, 2, 320, 240
    sc = {k: v.to(dev) for k, v in make_scene(n, 5).items()}
    viewmats, Ks = make_cameras(C, W, H)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    radii, m2d, dep, con, _ = ops.fully_fused_projection(sc["means"], None, sc["quats"], torch.exp(sc["scales"]) * 2.0,
                                                         viewmats.to(dev), Ks.to(dev), W, H)
    tpg, ids, flat = ops.isect_tiles(m2d, radii, dep, 16, tw, th, n_cameras=C)
    M = ids.shape[0]
    order = torch.empty(C * tw * th, dtype=torch.int32, device=dev)
    flat2 = torch.empty(M, dtype=torch.int32, device=dev)
    off, M_dev, st = ops.isect_bin_sort(m2d, radii, dep, tw, th, M, None, flat2, tile_order=order)
    assert int(M_dev) == M and int(st) == 0 and torch.equal(flat2, flat)
    assert torch.equal(torch.sort(order.long())[0], torch.arange(C * tw * th, device=dev))
    counts = (off[1:] - off[:-1]).long()
    lead = counts[order.long()[:16]].float().mean()
    assert float(lead) >= float(counts.float().mean())              # the launch starts with the long lists
    g = torch.Generator().manual_seed(2)
    cols = torch.rand(C, n, 3, generator=g).to(dev)
    opac = (torch.rand(C, n, generator=g) * 0.8 + 0.1).to(dev)
    rec = ops._PackRecords.apply(m2d, con, cols, opac)
    res = []
    for o in (None, order):
        r, a, nt, last = ops._RasterizeRecords.apply(rec, m2d, con, None, off, flat, 3, W, H, 0.5, False, True, True,
                                                     None, o)
        res.append((r, a, nt, last))
    for x, y in zip(res[0], res[1]):
        assert torch.equal(x, y)


@pytest.mark.parametrize("n", [60000, 140000])
def test_binning_variants_give_identical_tile_lists(dev, n):
    """direct placement (60 k x 8 = 480 k instances) and the spatial pre-sort of the instances (140 k x 8 > 2^20,
    csrc/isect_bin.hip 3c) against a device-wide stable sort of the unsorted emission; 8 cameras, ragged last tile row"""
    from gslam_amd import ops
    from gslam_amd.synthetic import make_cameras, make_scene
    C, W, H = 8, 330, 250
    sc = {k: v.to(dev) for k, v in make_scene(n, 6).items()}
    viewmats, Ks = make_cameras(C, W, H)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    radii, m2d, dep, _, _ = ops.fully_fused_projection(sc["means"], None, sc["quats"], torch.exp(sc["scales"]) * 2.0,
                                                       viewmats.to(dev), Ks.to(dev), W, H)
    # yardstick: gsplat's unsorted emission (own scan + emit) put in order by a device-wide stable sort of the 64-bit keys
    _, ids_u, flat_u = ops.isect_tiles(m2d, radii, dep, 16, tw, th, sort=False, n_cameras=C)
    ids1, order = torch.sort(ids_u, stable=True)
    flat1 = flat_u[order]
    _, ids2, flat2 = ops.isect_tiles(m2d, radii, dep, 16, tw, th, n_cameras=C)
    assert ids1.shape[0] > 100000 and torch.equal(ids1, ids2) and torch.equal(flat1, flat2)


def test_geometry_only_backward_gives_the_same_pose_gradient(dev):
    """frozen map, no depth channel: the rasteriser backward reduces only the xy / conic columns (GEOM kernels); the
    pose gradient equals the one of the full backward (map learnable, same render)"""
    sa, wa = _window(dev, cams=1)
    sb, wb = _window(dev, cams=1)
    for k in PARAMS:
        getattr(sb, k).requires_grad_(False)
    outs = []
    for s, w in ((sa, wa), (sb, wb)):
        out = s([w[0].camera], [w[0].pose], render_depth=False)           # RGB + beta: four channels, no depth
        loss = ((out.rgbs - w[0].img[None]).square().sum(-1) / out.betas.square()).mean() + 0.01 * out.alphas.mean()
        loss.backward()
        outs.append((w[0].pose.dR.grad.clone(), w[0].pose.dt.grad.clone()))
    _close(outs[1][0], outs[0][0], 1e-4, "dR")
    _close(outs[1][1], outs[0][1], 1e-4, "dt")


@pytest.mark.parametrize("second_use", [False, True])
def test_pose_gradient_through_the_projection_partials_equals_the_two_launch_path(dev, second_use):
    """primitives.pose_batch links its view matrices to the projection backward (the pose partials go straight into
    gsx_pose_zhou_bwd_partials); torch.stack of the module forwards (primitives.PoseZhou.forward, torch ops) takes the
    finishing pass + torch autograd instead.  With ``second_use`` the view matrices also feed the loss directly, so
    that a second gradient reaches them beside the projection's."""
    from gslam_amd.primitives import pose_batch
    splats, window = _window(dev, n=3000, W=160, H=128, cams=3, seed=57)
    cams = [f.camera for f in window]
    wmat = torch.randn(len(window), 4, 4, generator=torch.Generator().manual_seed(3)).to(dev)
    gen = torch.Generator().manual_seed(9)
    for f in window:                                    # away from the identity: the pose Jacobian is not trivial
        with torch.no_grad():
            f.pose.dR.add_(0.05 * torch.randn(6, generator=gen).to(dev))
            f.pose.dt.add_(0.05 * torch.randn(3, generator=gen).to(dev))
    grads = []
    for linked in (True, False):
        for f in window:
            f.pose.dR.grad = f.pose.dt.grad = None
        viewmats = pose_batch([f.pose for f in window]) if linked else torch.stack([f.pose() for f in window])
        assert (getattr(viewmats, "_gsx_pose_link", None) is not None) == linked
        out = splats._render(cams, viewmats, "RGB+D", 0.5)
        loss = (out.rgbs * out.rgbs).sum() + out.depthmaps.sum()
        if second_use:
            loss = loss + (viewmats * wmat).sum()
        loss.backward()
        grads.append([(f.pose.dR.grad.clone(), f.pose.dt.grad.clone()) for f in window])
    for (a_r, a_t), (b_r, b_t) in zip(*grads):
        _close(a_r, b_r, 2e-4, "dR")
        _close(a_t, b_t, 2e-4, "dt")

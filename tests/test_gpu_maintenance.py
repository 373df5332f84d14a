"""Map maintenance: gslam/pruning.py and gslam/insertion.py counterparts on the one-launch
row re-packing kernels.  Run with -m gpu."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _map(dev, n=4000, seed=0):
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.synthetic import make_scene
    sc = make_scene(n, seed)
    sc["scales"] = sc["scales"] + 0.5
    sc["ages"] = torch.arange(n) % 7
    return GaussianSplattingData.from_dict(sc, dev)


def _frames(dev, m, W=320, H=240, n=3):
    from gslam_amd.primitives import Camera, Frame, PoseZhou
    from gslam_amd.synthetic import make_intrinsics, make_viewmat
    K = make_intrinsics(W, H).to(dev)
    cam = Camera(K, H, W)
    out = []
    for i in range(n):
        V = make_viewmat(i).to(dev)
        with torch.no_grad():
            o = m([cam], [PoseZhou(V, is_learnable=False).to(dev)], render_depth=True)
        f = Frame(img=(o.rgbs[0].clamp(0.02, 0.98) * 0.9).contiguous(), timestamp=0.0, camera=cam, pose=PoseZhou(V).to(dev),
                  gt_pose=V, index=i, exposure_params=torch.zeros(2, device=dev))
        f.est_depths = o.depthmaps[0].detach().clone()
        out.append(f)
    return cam, out


def test_gather_and_concat_rows_match_torch(dev):
    from gslam_amd.pruning import concat_rows, gather_rows
    g = torch.Generator().manual_seed(0)
    n = 1237
    ts = [torch.randn(n, 3, generator=g), torch.randn(n, 4, generator=g), torch.randn(n, generator=g),
          torch.randint(0, 1 << 40, (n,), generator=g), torch.randn(n, 16, 3, generator=g),
          torch.randint(0, 100, (n,), generator=g).int()]
    ts = [t.to(dev) for t in ts]
    keep = (torch.rand(n, generator=g) > 0.4).to(dev)
    idx = torch.nonzero(keep).reshape(-1)
    for got, t in zip(gather_rows(ts, idx), ts):
        assert got.dtype == t.dtype and torch.equal(got, t[keep])
    perm = torch.randperm(n, generator=g)[:200].to(dev)          # arbitrary (repeating, unordered) indices as well
    for got, t in zip(gather_rows(ts, perm), ts):
        assert torch.equal(got, t[perm])
    nb = 77
    bs = [torch.randn(nb, 3, generator=g).to(dev), None, torch.randn(nb, generator=g).to(dev),
          torch.randint(0, 1 << 40, (nb,), generator=g).to(dev), None, None]
    for got, a, b in zip(concat_rows(ts, bs, nb), ts, bs):
        ref = torch.cat([a, torch.zeros((nb,) + tuple(a.shape[1:]), dtype=a.dtype, device=dev) if b is None else b])
        assert torch.equal(got, ref)
    assert gather_rows(ts, idx[:0])[0].shape == (0, 3)


def test_prune_using_mask_and_strategies(dev):
    from gslam_amd.mapping import BundleAdjuster, MapConfig
    from gslam_amd.pruning import (PruneByVisibility, PruneIllConditionedGaussians, PruneLargeGaussians,
                                   PruneLowOpacity, prune_using_mask)
    m = _map(dev)
    cam, frames = _frames(dev, m)
    ba = BundleAdjuster(m, MapConfig(), need_n_touched=True)
    for _ in range(2):
        ba.step(frames[:2])
    torch.cuda.synchronize()
    out = ba.last_outputs
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    st_before = {n: {k: v.clone() for k, v in ba.optimizers.splat_opt.state[p].items() if torch.is_tensor(v)}
                 for n, p in m.named_parameters() if p in ba.optimizers.splat_opt.state}
    assert len(st_before) == 6
    # strategies = the reference's mask formulas
    low = PruneLowOpacity(0.3).step(m, None)
    assert torch.equal(low, torch.sigmoid(m.opacities) < 0.3)
    big = PruneLargeGaussians(40.0).step(m, None, out.radii.max(dim=0).values)
    assert torch.equal(big, out.radii.max(dim=0).values > 40.0)
    ill = PruneIllConditionedGaussians(0).step(m, None, out.radii, out.n_touched)
    assert torch.equal(ill, ((out.radii > 0) & (out.n_touched == 0)).sum(0) > 0)
    vis = PruneByVisibility(8, 2).step(m, None, (out.radii > 0).sum(0), latest_kf_age=6)
    assert vis.dtype == torch.bool and vis.shape == low.shape
    keep = ~(low | big)
    extra = [out.radii.max(dim=0).values.clone()]
    n_pruned = prune_using_mask(m, ba.optimizers, keep, extra)
    assert int(n_pruned) == int((~keep).sum())
    for n, p in m.named_parameters():
        assert torch.equal(p.detach(), before[n][keep]), n
        assert p.requires_grad == (n != "ages")
        if n in st_before:
            st = ba.optimizers.splat_opt.state[p]
            for k, v in st_before[n].items():
                assert torch.equal(st[k], v[keep] if v.dim() >= 1 and v.shape[0] == keep.shape[0] else v), (n, k)
    assert extra[0].shape[0] == int(keep.sum())
    # an all-false mask is refused like the reference (pruning.py:16-17)
    assert prune_using_mask(m, ba.optimizers, torch.zeros_like(keep[keep])) == 0
    # the optimiser keeps working on the re-packed map
    ba.map_changed()
    total, _ = ba.step(frames[:2])
    torch.cuda.synchronize()
    assert torch.isfinite(total).item() and m.means.shape[0] == int(keep.sum())


def test_insertion_strategies(dev):
    from gslam_amd.insertion import InsertFromDepthMap, InsertUsingImagePlaneGradients, SequentialInsertion
    from gslam_amd.mapping import BundleAdjuster, MapConfig
    torch.manual_seed(0)
    m = _map(dev, n=3000)
    cam, frames = _frames(dev, m)
    ba = BundleAdjuster(m, MapConfig())
    ba.step(frames[:2])
    torch.cuda.synchronize()
    out = ba.last_outputs
    n0 = m.means.shape[0]
    opt = ba.optimizers.splat_opt
    m_before = m.means.detach().clone()
    ea_before = opt.state[m.means]["exp_avg"].clone()
    # densification by image-plane gradients (insertion.py:287-347)
    g = out.means2d.grad.clone()
    g[..., 0] *= out.width / 2.0 * out.n_cameras
    g[..., 1] *= out.height / 2.0 * out.n_cameras
    thr = float(g.norm(dim=-1).mean(dim=0).quantile(0.9))
    nd, ns = InsertUsingImagePlaneGradients(thr, 0.05).step(m, ba.optimizers, out, frames[0], 0)
    n1 = m.means.shape[0]
    assert nd + ns > 0 and n1 == n0 + nd + ns and out.radii.shape[1] == n1
    assert torch.equal(m.means.detach()[:n0], m_before)
    st = opt.state[m.means]
    assert torch.equal(st["exp_avg"][:n0], ea_before) and float(st["exp_avg"][n0:].abs().max()) == 0.0
    assert float((m.log_uncertainties.detach()[n0:] - 1.0).abs().max()) == 0.0
    if ns > 0:   # split children: scales shrunk by 1.6 (insertion.py:95-96)
        assert float(m.scales.detach()[n0 + nd:].max()) <= float(m.scales.detach()[:n0].max()) - math.log(1.6) + 1e-5
    # insertion from the depth map of a new frame (insertion.py:100-284)
    with torch.no_grad():
        o2 = m([cam], [frames[2].pose], render_depth=True)
    ins = InsertFromDepthMap(0.05, 0.2, 0.6, 0.5)
    added = ins.step(m, ba.optimizers, o2, frames[2], 500, frames)
    n2 = m.means.shape[0]
    assert 0 < added <= 500 and n2 == n1 + added
    assert bool((m.ages.detach()[n1:] == frames[2].index).all()) and m.ages.dtype == torch.int64
    assert abs(float(torch.sigmoid(m.opacities.detach()[n1:]).mean()) - 0.5) < 1e-5
    # the new means lie in front of the inserting camera
    V = frames[2].pose().detach()
    pc = m.means.detach()[n1:] @ V[:3, :3].t() + V[:3, 3]
    assert float(pc[:, 2].min()) >= 0.1 - 1e-4
    assert opt.state[m.means]["exp_avg"].shape[0] == n2 and opt.param_groups[0]["params"][0] is m.means
    # and BA continues on the grown map
    SequentialInsertion([]).step(m, ba.optimizers, o2, frames[2], 0)
    ba.map_changed()
    total, _ = ba.step(frames[:3])
    torch.cuda.synchronize()
    assert torch.isfinite(total).item()

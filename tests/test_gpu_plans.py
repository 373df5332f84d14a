"""Launch plans (gslam_amd/plan.py) against the autograd-shaped operators they replace in the optimisation loops: same
C-ABI launches, so the numbers must agree to float noise (atomics in the rasteriser backward).  Also the regression test
for the round-1 crash: the window pose refiner run in the context the SLAM loop runs it in (eager BA on the default
stream first, then the refiner, then a prune that re-packs the map, then the refiner again).  Run with -m gpu."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _world(dev, n=20000, W=320, H=240, seed=0):
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.primitives import Camera, Frame, PoseZhou
    from gslam_amd.synthetic import make_intrinsics, make_scene, make_viewmat
    K = make_intrinsics(W, H).to(dev)
    cam = Camera(K, H, W)
    sc = make_scene(n, seed)
    sc["scales"] = sc["scales"] + 0.4
    m = GaussianSplattingData.from_dict(sc, dev)

    def frame(i, start, index=None):
        V = make_viewmat(i).to(dev)
        with torch.no_grad():
            img = m([cam], [PoseZhou(V, is_learnable=False).to(dev)], render_depth=False).rgbs[0].clamp(0, 1)
        V0 = make_viewmat(start).to(dev)
        return Frame(img=img.contiguous(), timestamp=0.0, camera=cam, pose=PoseZhou(V0).to(dev), gt_pose=V,
                     index=i if index is None else index, exposure_params=torch.zeros(2, device=dev))
    return m, cam, frame


def test_hip_graph_runtime_roundtrip(dev):
    """csrc/runtime.hip: capture a chain of libgsx launches on an own stream, instantiate, replay on another stream"""
    import ctypes as C
    from gslam_amd._lib import check, lib
    from gslam_amd.plan import HipGraph
    buf = torch.ones(1000, dtype=torch.int32, device=dev)
    ctr = torch.zeros(1, dtype=torch.int64, device=dev)
    s = torch.cuda.Stream()

    def chain(st):
        check(lib.gsx_zero_words(buf.data_ptr(), 500, st), "zero")
        check(lib.gsx_counters_add(1, (C.c_void_p * 1)(ctr.data_ptr()), 3, st), "add")
    g = HipGraph()
    torch.cuda.synchronize()
    g.capture(s, chain)
    torch.cuda.synchronize()
    assert g.captured and g.nodes == 2
    assert int(buf.sum()) == 1000 and int(ctr) == 0                       # capturing did not execute anything
    g.launch(count=4)
    torch.cuda.synchronize()
    assert int(buf[:500].sum()) == 0 and int(buf[500:].sum()) == 500 and int(ctr) == 12
    g.destroy()
    assert not g.captured
    # pinned, device-mapped host words
    h, d = C.c_void_p(), C.c_void_p()
    check(lib.gsx_host_alloc(C.byref(h), C.byref(d), 64), "host_alloc")
    assert h.value and d.value
    check(lib.gsx_host_free(h), "host_free")


def test_track_closure_matches_autograd_operators(dev):
    """TrackClosure (host tail) == splats(...) + tracking loss + backward through the autograd operators"""
    from gslam_amd.losses import fused_tracking_loss
    from gslam_amd.plan import TrackClosure, current_stream_ptr
    from gslam_amd.primitives import PoseZhou
    m, cam, frame = _world(dev)
    mf = m.no_grad_clone()
    f = frame(2, 1)
    exposure = torch.tensor([0.05, -0.02], device=dev)
    # autograd-shaped
    pose = PoseZhou(f.pose.Rt.clone()).to(dev)
    with torch.no_grad():
        pose.dR.copy_(torch.tensor([0.01, -0.02, 0.005, 0.0, 0.01, -0.01], device=dev))
        pose.dt.copy_(torch.tensor([0.02, -0.01, 0.015], device=dev))
    ex = exposure.clone().requires_grad_(True)
    out = mf([cam], [pose], render_depth=False, need_n_touched=False)
    loss = fused_tracking_loss(out, f.img, ex)
    loss.backward()
    # plan
    c = TrackClosure(mf, cam, tail='host')
    c.load(f.pose.Rt, f.img, exposure)
    with torch.no_grad():
        c.dR.copy_(pose.dR)
        c.dt.copy_(pose.dt)
    c.r.probe()
    c.enqueue(current_stream_ptr(dev))
    torch.cuda.synchronize()
    assert c.r.check_capacity()
    assert c.r.compact and torch.equal(c.r.render, out._render)       # same render, bit for bit (records per instance)
    assert abs(float(c.loss) - float(loss)) < 1e-6 * abs(float(loss))
    for a, b in ((c.g_dR, pose.dR.grad), (c.g_dt, pose.dt.grad), (c.g_exposure, ex.grad)):
        assert (a - b).abs().max() < 2e-4 * b.abs().max() + 1e-9, (a, b)
    # the fused variant evaluates the same loss in the forward rasteriser's epilogue (gsx_raster_fwd_track_loss): same
    # gradient of the render, same loss and exposure gradient from its per-tile rows
    c2 = TrackClosure(mf, cam, tail='fused')
    c2.load(f.pose.Rt, f.img, exposure)
    c2.r.viewmats.copy_(c.r.viewmats)
    c2.r.probe()
    denom = c2.r.H * c2.r.W
    c2.r.forward(current_stream_ptr(dev), track_loss=(c2.img, c2.exposure, 1.0 / denom, c2.loss_rows))
    torch.cuda.synchronize()
    assert torch.equal(c2.r.alphas, c.r.alphas) and torch.equal(c2.r.last_ids, c.r.last_ids)
    assert (c2.r.v_render - c.r.v_render).abs().max() <= 1e-6 * c.r.v_render.abs().max()
    rows = c2.loss_rows.double().sum(0)
    assert abs(float(rows[0]) / denom - float(loss)) < 1e-5 * abs(float(loss))
    assert (rows[3:5].float() - c.g_exposure).abs().max() < 1e-4 * c.g_exposure.abs().max() + 1e-9
    # captured and replayed: same numbers, idempotent
    c.prepare()
    vals = []
    for _ in range(3):
        c.graph.replay()
        torch.cuda.synchronize()
        vals.append((float(c.loss), c.g_dR.clone(), c.g_dt.clone()))
    assert c.r.check_capacity()
    for v, gr, gt in vals:
        assert abs(v - float(loss)) < 1e-6 * abs(float(loss))
        assert (gr - pose.dR.grad).abs().max() < 2e-4 * pose.dR.grad.abs().max() + 1e-9


def test_mapping_step_plan_matches_autograd_step(dev):
    """MappingStep (eager launches and graph replay) == BundleAdjuster.step on the autograd operators: loss values, every
    map gradient, the pose gradients, and the parameters after the update"""
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.mapping import BundleAdjuster
    from gslam_amd.primitives import Camera, Frame, PoseZhou
    from gslam_amd.synthetic import make_cameras, make_scene
    n, W, H = 6000, 320, 240
    sc = make_scene(n, 11)
    sc["scales"] = sc["scales"] + 0.5
    viewmats, Ks = make_cameras(3, W, H)
    gt = torch.rand(3, H, W, 3, generator=torch.Generator().manual_seed(4)).to(dev)

    def build(capturable):
        splats = GaussianSplattingData.from_dict(sc, dev)
        window = [Frame(img=gt[i].contiguous(), timestamp=0.0, camera=Camera(Ks[i].to(dev), H, W),
                        pose=PoseZhou(viewmats[i].to(dev), is_learnable=(i != 0)).to(dev), gt_pose=viewmats[i].to(dev),
                        index=i, exposure_params=torch.tensor([0.03 * i, -0.01 * i], device=dev)) for i in range(3)]
        return splats, window, BundleAdjuster(splats, capturable=capturable)

    sa, wa, ba = build(False)
    ta, pa = ba.render_backward(wa)
    grads_a = {k: getattr(sa, k).grad.clone() for k in ("means", "quats", "scales", "opacities", "colors",
                                                         "log_uncertainties")}
    pose_a = [(f.pose.dR.grad.clone(), f.pose.dt.grad.clone()) for f in wa[1:]]
    ba.update()
    sb, wb, bb = build(True)
    plan = bb.plan(wb)
    assert plan.learnable == [False, True, True] and plan.mine == [0, 1, 2]
    tb, pb = plan.render_backward()
    torch.cuda.synchronize()
    assert plan.capacity_ok()
    assert abs(float(ta) - float(tb)) < 1e-5 * abs(float(ta)) and abs(float(pa) - float(pb)) < 1e-5 * abs(float(pa))
    for k, ga in grads_a.items():
        gb = plan.grad_views[k]
        assert (ga - gb).abs().max() < 2e-4 * ga.abs().max() + 1e-10, k
    for (gr, gtv), i in zip(pose_a, (1, 2)):
        assert (plan.g_dR[i] - gr).abs().max() < 1e-3 * gr.abs().max() + 1e-9
        assert (plan.g_dt[i] - gtv).abs().max() < 1e-3 * gtv.abs().max() + 1e-9
    out = plan.as_output()
    assert out.means2d.grad.shape == (3, n, 2) and out.depthmaps.shape == (3, H, W) and out.radii.shape == (3, n)
    assert torch.equal(out.radii, ba.last_outputs.radii)
    # render_backward() keeps the gradient records for the densification to read ...
    ref_g = ba.last_outputs.means2d.grad
    assert float((out.means2d.grad - ref_g).abs().max()) < 2e-3 * float(ref_g.abs().max())
    # the update: one graph replay from the same start (the eager render_backward above did not touch the parameters)
    plan.step()
    torch.cuda.synchronize()
    # ... a step() consumes them: every row the projection backward read is zero again (no forward clears them)
    assert float(plan.r.v_rec.abs().max()) == 0.0
    assert plan.graph.captured and plan.capacity_ok()
    for k in grads_a:
        a, b = getattr(sa, k), getattr(sb, k)
        assert (a - b).abs().mean() < 2e-5, (k, float((a - b).abs().mean()))
    assert (wa[1].pose.dR - wb[1].pose.dR).abs().max() < 1e-3
    assert torch.equal(wb[0].pose.dR, torch.zeros_like(wb[0].pose.dR))            # the fixed pose did not move


def test_refiner_after_eager_ba_and_prune(dev):
    """Regression for the round-1 SIGSEGV (GraphedPoseRefiner.capture inside Backend.idle_step): eager autograd BA over the
    keyframes' own poses on the default stream, then the refiner on the same keyframes, then a prune that re-packs the
    map (new tensors, new N), then the refiner again on a slid window.  The refiner owns its pose slots and its stream
    and records plain launches, so neither the autograd history of the poses nor the re-pack can reach its capture."""
    from gslam_amd.mapping import BundleAdjuster, GraphedPoseRefiner
    from gslam_amd.pruning import prune_using_mask
    m, cam, frame = _world(dev, n=15000)
    window = [frame(0, 0, index=0), frame(2, 1), frame(4, 3), frame(6, 5)]
    for p in window[0].pose.parameters():
        p.requires_grad_(False)
    ba = BundleAdjuster(m, capturable=True)
    for _ in range(3):                                     # leaves AccumulateGrad history on the window's pose leaves
        ba.step(window)
    assert ba.last_outputs is not None                      # the autograd graph of the last step is still referenced
    ref = GraphedPoseRefiner(m, window)
    before = [float((f.pose()[:3, 3] - f.gt_pose[:3, 3]).norm()) for f in window]
    loss1, n1 = ref.run(window)
    torch.cuda.synchronize()
    assert 2 <= n1 <= 26 and loss1 == loss1
    g1 = ref.graph
    loss1b, _ = ref.run(window)                             # same window again: same graph
    assert ref.graph is g1
    # prune a third of the map: new parameter tensors -> the old refiner no longer matches, a new one is built
    keep = torch.rand(m.means.shape[0], device=dev) > 0.33
    assert prune_using_mask(m, ba.optimizers, keep) > 0
    ba.map_changed()
    assert not ref.matches(m, window)
    window2 = window[1:] + [frame(8, 7)]                    # slid window: no fixed frame any more
    ref2 = GraphedPoseRefiner(m, window2)
    loss2, n2 = ref2.run(window2)
    torch.cuda.synchronize()
    assert 2 <= n2 <= 26 and loss2 == loss2
    # the same refiner serves another window of the same shape without re-capturing
    window3 = window2[1:] + [frame(10, 9)]
    g2 = ref2.graph
    assert ref2.matches(m, window3)
    ref2.run(window3)
    assert ref2.graph is g2
    ba.step(window2)                                         # and eager BA still works afterwards
    torch.cuda.synchronize()
    after = [float((f.pose()[:3, 3] - f.gt_pose[:3, 3]).norm()) for f in window]
    assert after[0] == before[0]                             # frame 0 fixed (backend.py:459-462)


@pytest.mark.parametrize("n,n_cams,grads", [(20000, 1, 'full'), (20000, 1, 'pose'), (9000, 3, 'full'), (300000, 1, 'pose'),
                                            (37, 2, 'full'), (20001, 2, 'full'), (20001, 1, 'pose')])
def test_fused_front_equals_projection_plus_bin_sort(dev, n, n_cams, grads):
    """gsx_front_fwd (projection + count rows + instance records, in-kernel offset scan, striped placement) against
    gsx_project_fwd + gsx_isect_bin_sort: every integer output bit for bit, and the render that follows"""
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.plan import RenderPlan, current_stream_ptr
    from gslam_amd.synthetic import make_cameras, make_scene
    W, H = 640, 480
    sc = make_scene(n, 5)
    # (20001: Gaussians e^1.5 times larger and strongly anisotropic - many of them straddle the frustum's border, where the
    # front's conservative cull (radius bound from the largest scale and |R|_2) must not drop what the exact projection keeps)
    sc["scales"] = sc["scales"] + (0.3 if n <= 20000 else 1.5 if n == 20001 else 0.0)
    if n == 20001:
        sc["scales"][:, 0] += 1.0
    m = GaussianSplattingData.from_dict(sc, dev).no_grad_clone()
    viewmats, Ks = make_cameras(n_cams, W, H)
    plans = []
    for front in (False, True):
        r = RenderPlan(m, n_cams, W, H, render_depth=(grads == 'full'), grads=grads, front=front)
        assert r.front == front
        r.Ks.copy_(Ks.to(dev))
        r.viewmats.copy_(viewmats.to(dev))
        r.probe()
        if r.v_rec is not None:
            r.v_rec.fill_(7.0)                              # left behind by a backward(keep=True): the forward must clear
            r._v_rec_dirty = True                           # the rows it will accumulate into
        r.forward(current_stream_ptr(dev))
        torch.cuda.synchronize()
        assert r.check_capacity()
        plans.append(r)
    a, b = plans
    M = a.last_M
    assert M == b.last_M and M == int(a.tiles.sum())
    assert torch.equal(a.offsets, b.offsets)
    if not b.compact:
        assert torch.equal(a.radii, b.radii) and torch.equal(a.tiles, b.tiles) and torch.equal(a.vis_count, b.vis_count)
        assert torch.equal(a.flat[:M], b.flat[:M])          # the sorted (tile, depth, id) order: bit-exact assignment
        vis = a.radii > 0
        assert torch.equal(a.rec[vis], b.rec[vis])
        assert torch.equal(a.rec, b.rec) and torch.equal(a.means2d, b.means2d)
        assert torch.equal(a.as_output().conics, b.as_output().conics)
        assert torch.equal(a.depths, b.depths) and float(b.v_rec.abs().max()) == 0.0
    else:
        # pose-only plans keep one record per visible instance; the tile lists carry slots: mapped back to flatten ids they
        # are the same lists bit for bit, and every listed slot holds the record of its Gaussian with a cleared gradient row
        ids = b.slot_flatten_ids()
        slots = b.flat[:M].long()
        assert torch.equal(ids[slots], a.flat[:M].long())
        assert torch.equal(b.rec[slots], a.rec.view(-1, 12)[a.flat[:M].long()])
        assert float(b.v_rec[slots].abs().max()) == 0.0
    assert torch.equal(a.render, b.render) and torch.equal(a.alphas, b.alphas) and torch.equal(a.last_ids, b.last_ids)


@pytest.mark.parametrize("n,n_cams", [(30000, 1), (12000, 3)])
def test_front_pose_backward_equals_projection_backward(dev, n, n_cams):
    """gsx_front_pose_bwd (one thread per visible instance record) against gsx_project_bwd's pose-only pass over all
    Gaussians: the summed pose partials d loss / d [R | t] of every camera"""
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.plan import RenderPlan, current_stream_ptr
    from gslam_amd.synthetic import make_cameras, make_scene
    W, H = 320, 240
    sc = make_scene(n, 8)
    sc["scales"] = sc["scales"] + 0.4
    m = GaussianSplattingData.from_dict(sc, dev).no_grad_clone()
    viewmats, Ks = make_cameras(n_cams, W, H)
    sums = []
    for front in (False, True):
        r = RenderPlan(m, n_cams, W, H, render_depth=False, grads='pose', front=front)
        r.Ks.copy_(Ks.to(dev))
        r.viewmats.copy_(viewmats.to(dev))
        r.probe()
        st = current_stream_ptr(dev)
        r.forward(st)
        r.v_render.copy_(torch.randn(r.v_render.shape, generator=torch.Generator().manual_seed(3)).to(dev) * 1e-3)
        r.backward(st)
        torch.cuda.synchronize()
        assert r.check_capacity()
        part = r.pose_ws[:r.pose_blocks * n_cams * 48].view(torch.float32).view(r.pose_blocks, n_cams, 12)
        sums.append(part.double().sum(dim=0))
    a, b = sums
    assert float(a.abs().max()) > 0
    assert float((a - b).abs().max()) < 2e-4 * float(a.abs().max()), (a, b)


@pytest.mark.parametrize("T,G", [(1200, 256), (300, 256), (2048, 256), (1024, 256), (200, 256), (1201, 304)])
def test_tile_balance_order(dev, T, G):
    """gsx_tile_balance: always a permutation; the groups {i, i + G, i + 2G, ...} of the launch order carry near-equal weight
    (light_rate = 1), or the groups without a tile in the last round carry light_rate x the others' weight"""
    import numpy as np
    from gslam_amd._lib import check, lib, ptr
    from gslam_amd.plan import current_stream_ptr
    g = torch.Generator().manual_seed(T)
    chunks = torch.randint(4, 20, (T,), generator=g)
    trips = (torch.randn(T, generator=g).mul(0.35).exp() * 450).to(torch.int64)
    work = torch.stack([chunks, trips], 1).to(torch.int32).to(dev).contiguous()
    w = (trips.double() + 3.0 * chunks.double() + 1.0).numpy()
    st = current_stream_ptr(dev)

    def run(work_t, light_rate):
        order = torch.full((T,), -1, dtype=torch.int32, device=dev)
        check(lib.gsx_tile_balance(ptr(work_t), T, 3.0, light_rate, G, ptr(order), st), "gsx_tile_balance")
        torch.cuda.synchronize()
        o = order.cpu().numpy()
        assert sorted(o.tolist()) == list(range(T))
        return o

    def sums(o):
        s = np.zeros(G)
        np.add.at(s, np.arange(T) % G, w[o])
        return s

    def emulate(light):
        """the algorithm of csrc/tile_balance.h with exact sorts"""
        srt = np.argsort(-w, kind="stable")
        rounds, last = -(-T // G), T - (-(-T // G) - 1) * G
        v, S = w[srt[T - last:]].mean(), w.sum() / (last + (G - last) * light)
        s = np.where(np.arange(G) < last, max(0.0, v - (1 - light) * S) if last < G else 0.0, 0.0)
        real = np.zeros(G)
        for r in range(rounds):
            nb = last if r == rounds - 1 else G
            rank = np.argsort(np.argsort(s[:nb], kind="stable"), kind="stable")
            for b in range(nb):
                t = srt[r * G + rank[b]]
                s[b] += w[t]
                real[b] += w[t]
        return real

    o = run(work, 1.0)
    run(torch.zeros_like(work), 0.92)                         # no measurements yet: still a permutation
    if T <= G:
        return
    last = T - (-(-T // G) - 1) * G
    for light in (1.0, 0.9):
        s, e = sums(run(work, light)), emulate(light)
        # the kernel sorts the tiles on 1024 weight levels: the group sums follow the exact algorithm closely, not exactly
        assert abs(s.max() - e.max()) < 0.02 * e.mean() and abs(s.min() - e.min()) < 0.02 * e.mean(), (light, s.max(), e.max())
        if last < G:
            assert abs(s[last:].mean() / s[:last].mean() - e[last:].mean() / e[:last].mean()) < 0.01
    s, ident = sums(o), sums(np.arange(T))
    if T >= 4 * G:
        assert s.std() < 0.5 * ident.std()                    # and that is a balance: spread of the group sums halved at least
    else:
        assert s.max() <= ident.max()


def test_balanced_track_closure_equals_identity_order(dev):
    """the CU-balanced launch order changes which workgroup renders which tile, nothing else: loss and pose gradient of a
    tracking closure agree with the identity order to float-atomic reordering"""
    from gslam_amd.plan import TrackClosure, current_stream_ptr
    from gslam_amd.primitives import Camera
    from gslam_amd.synthetic import make_intrinsics, make_scene, make_viewmat
    from gslam_amd.map import GaussianSplattingData
    W, H = 640, 480
    sc = make_scene(60000, 5)
    sc["scales"] = sc["scales"] + 0.4
    splats = GaussianSplattingData.from_dict(sc, dev)
    cam = Camera(make_intrinsics(W, H).to(dev), H, W)
    c = TrackClosure(splats, cam)
    assert c.r.balanced_order is not None and c.r.T == 1200
    img = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(3)).to(dev)
    c.load(make_viewmat(1.0).to(dev), img, torch.tensor([0.01, -0.02], device=dev))
    c.r.probe()
    st = current_stream_ptr(dev)

    denom = H * W

    def once():
        c.r.forward(st, track_loss=(c.img, c.exposure, 1.0 / denom, c.loss_rows))
        c.r.backward(st)
        torch.cuda.synchronize()
        return (c.loss_rows.sum(0).clone(), c.r.v_render.clone(), None,
                c.r.pose_ws.view(torch.float32)[:c.r.pose_blocks * 12].clone())

    a = once()
    assert c.r.check_capacity() and int(c.r.tile_work[:, 0].sum()) > 0 and int(c.r.tile_work[:, 1].sum()) > 0
    c.rebalance()
    torch.cuda.synchronize()
    o = c.r.balanced_order.cpu()
    assert sorted(o.tolist()) == list(range(1200)) and not torch.equal(o, torch.arange(1200, dtype=torch.int32))
    b = once()
    assert torch.equal(a[1], b[1])                                        # per-pixel results do not depend on the order
    torch.testing.assert_close(a[0], b[0], rtol=1e-5, atol=1e-6)
    assert float((a[3] - b[3]).abs().max()) < 1e-4 * float(a[3].abs().max()) + 1e-12


@pytest.mark.parametrize("front", [True, False])
def test_fused_raster_launch_equals_forward_then_backward(dev, front):
    """gsx_raster_track_fused (forward rasteriser + tracking loss + geometry-only backward of a tile in one workgroup) against
    gsx_raster_fwd_track_loss followed by gsx_raster_bwd: same statements in the same order - tile work bit for bit, loss rows
    to the last bits, gradient records and pose partials to the order of the float atomics"""
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.plan import TrackClosure, current_stream_ptr
    from gslam_amd.primitives import Camera
    from gslam_amd.synthetic import make_intrinsics, make_scene, make_viewmat
    W, H = 640, 480
    sc = make_scene(80000, 6)
    sc["scales"] = sc["scales"] + 0.4
    splats = GaussianSplattingData.from_dict(sc, dev)
    cam = Camera(make_intrinsics(W, H).to(dev), H, W)
    img = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(8)).to(dev)
    st = current_stream_ptr(dev)
    res = []
    for fuse in (False, True):
        c = TrackClosure(splats, cam, fuse_raster=fuse, front=front, row_keys=False)   # (tile-major segments compared)
        assert c.fuse_raster == fuse and c.r.front == front
        c.load(make_viewmat(2.0).to(dev), img, torch.tensor([0.02, -0.01], device=dev))
        c.r.probe()
        denom = H * W
        tl = (c.img, c.exposure, 1.0 / denom, c.loss_rows)
        if fuse:
            c.r.forward_track_fused(st, tl)
            torch.cuda.synchronize()
            v_rec = c.r.v_rec.clone()
            c.r.backward(st, rasterised=True)
        else:
            c.r.forward(st, track_loss=tl)
            check_v = None
            from gslam_amd._lib import check, lib, ptr
            r = c.r
            check(lib.gsx_raster_bwd(ptr(r.rec), r.CH, ptr(r.backgrounds), ptr(r.offsets), ptr(r.flat), r.capacity, 1, r.C, r.W,
                                     r.H, r.tile_w, r.tile_h, ptr(r.alphas), ptr(r.last_ids), ptr(r.v_render), None,
                                     ptr(r.v_rec), None, ptr(r.launch_order), 1, st), "gsx_raster_bwd")
            torch.cuda.synchronize()
            v_rec = c.r.v_rec.clone()
            c.r.backward(st, rasterised=True)
        torch.cuda.synchronize()
        assert c.r.check_capacity()
        work = c.r.tile_work.clone() if c.r.tile_work is not None else torch.zeros(1)
        res.append((c.loss_rows.clone(), work, v_rec,
                    c.r.pose_ws.view(torch.float32)[:c.r.pose_blocks * 12].clone(), c.r.flat[:c.r.last_M].clone()))
    a, b = res
    assert torch.equal(a[4], b[4])                                        # same tile lists (same front)
    # the same statements, but compiled inside another kernel (fused multiply-adds are contracted by context): last bits
    assert float((a[0] - b[0]).abs().max()) < 2e-6 * float(a[0].abs().max())
    assert torch.equal(a[1], b[1])                                        # work counters
    assert float(a[2].abs().max()) > 0
    assert float((a[2] - b[2]).abs().max()) < 1e-4 * float(a[2].abs().max())
    assert float((a[3] - b[3]).abs().max()) < 1e-4 * float(a[3].abs().max()) + 1e-12


def test_overflowed_ba_iteration_applies_no_update_and_bit2_raises(dev):
    """single rank: a BA iteration whose tile lists overflow inside the replayed graph leaves map, poses, Adam moments and step
    counters bit-identical (the update launches are gated on the overflow flag on the device), finish_step() reports it, the
    redo with the grown lists equals a run that never overflowed; and the hardening flag of the binning kernels (status bit
    2: a clamped, corrupt tile count) is an exception, never a silent ok"""
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.mapping import BundleAdjuster
    from gslam_amd.primitives import Camera, Frame, PoseZhou
    from gslam_amd.synthetic import make_cameras, make_scene
    n, W, H, C = 6000, 320, 240, 3
    sc = make_scene(n, 41)
    sc["scales"] = sc["scales"] + 0.5
    viewmats, Ks = make_cameras(C, W, H)
    gt = torch.rand(C, H, W, 3, generator=torch.Generator().manual_seed(3)).to(dev)

    def build(grow):
        splats = GaussianSplattingData.from_dict({k: v.clone() for k, v in sc.items()}, dev)
        window = [Frame(img=gt[i], timestamp=0.0, camera=Camera(Ks[i].to(dev), H, W),
                        pose=PoseZhou(viewmats[i].to(dev)).to(dev), gt_pose=viewmats[i].to(dev), index=i,
                        exposure_params=torch.zeros(2, device=dev)) for i in range(C)]
        ba = BundleAdjuster(splats, capturable=True)
        plan = ba.plan(window)
        plan.r.GROW = grow
        return splats, window, ba, plan

    names = ("means", "quats", "scales", "opacities", "colors", "log_uncertainties")
    res = []
    for grow in (1.0, 4.0):                      # no head-room (overflows after the map fattens) / plenty
        splats, window, ba, plan = build(grow)
        plan.step()
        assert plan.finish_step()[2]
        with torch.no_grad():
            splats.scales.add_(0.25)
        before = {k: getattr(splats, k).detach().clone() for k in names}
        ctr = int(ba.optimizers.splat_opt._shared_step.item())
        plan.step()
        total, pm, ok = plan.finish_step()
        if grow == 1.0:
            assert not ok
            assert all(torch.equal(before[k], getattr(splats, k).detach()) for k in names)
            assert int(ba.optimizers.splat_opt._shared_step.item()) == ctr
            assert float(window[1].pose.dR.abs().max()) > 0        # (moved by the first, clean iteration only)
            dR = window[1].pose.dR.detach().clone()
            plan.step()                                            # re-captured with the grown lists
            total, pm, ok = plan.finish_step()
            assert ok and not torch.equal(dR, window[1].pose.dR.detach())
        assert ok and int(ba.optimizers.splat_opt._shared_step.item()) == ctr + 1
        res.append(({k: getattr(splats, k).detach().clone() for k in names}, pm))
    assert abs(res[0][1] - res[1][1]) < 1e-5 * abs(res[1][1])
    for k in names:
        assert float((res[0][0][k] - res[1][0][k]).abs().mean()) < 2e-5, k
    # status bit 2: poison the sticky status word the way the binning kernels do when they clamp a corrupt count
    splats, window, ba, plan = build(2.0)
    plan.step()
    torch.cuda.synchronize()
    plan.r.status.fill_(2)
    with pytest.raises(RuntimeError, match="corrupt tile counts"):
        plan.capacity_ok()
    assert int(plan.r.status.item()) == 0 and plan.capacity_ok()


def _track_closure(dev, n, seed, candidates, W=640, H=480, scale_shift=0.0):
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.plan import TrackClosure
    from gslam_amd.primitives import Camera
    from gslam_amd.synthetic import make_intrinsics, make_scene
    sc = make_scene(n, seed)
    sc["scales"] = sc["scales"] + scale_shift
    splats = GaussianSplattingData.from_dict(sc, dev)
    cam = Camera(make_intrinsics(W, H).to(dev), H, W)
    # (full sorted lists are compared below: the stand-alone tile sort, not the one inside the rasteriser that sorts what it
    # composites - tested on its own in test_tile_sort_inside_the_rasteriser_equals_the_sort_launch)
    # candidates: False = the plain path (culls the map arrays, rebuilds covariances), True = per-frame candidate set with
    # margins, 'records' = records of the whole map + per-closure cull (the default of TrackClosure)
    return splats, TrackClosure(splats, cam, tail='fused', candidates=candidates is True, defer_sort=False,
                                map_records=candidates == 'records')


@pytest.mark.parametrize("n,W,H,shift", [(60000, 640, 480, 0.3), (500000, 640, 480, 0.0), (9000, 320, 240, 0.8)])
def test_candidate_set_closures_equal_full_path(dev, n, W, H, shift):
    """per-frame candidate set (gsx_front_candidates): closures that project the candidates' pose-independent records give the
    SAME tile lists, records and pose gradient, bit for bit, as closures that cull the whole map - at the frame's first pose,
    at poses inside the margins, and (by falling back on their own) at a pose outside them; and so do closures over the
    records of the whole map (GSX_PROJ_MAP_RECORDS: their own cull from the packed cull rows, one record per survivor), which
    are valid at every pose"""
    from gslam_amd.synthetic import make_viewmat
    res = {}
    g = torch.Generator().manual_seed(3)
    img = torch.rand(H, W, 3, generator=g).to(dev)
    deltas = [(0.0, 0.0), (0.004, 0.006), (0.012, 0.012), (0.2, 0.3)]        # the last one leaves the 0.02 / 0.02 margins
    for cand in (False, True, 'records'):
        splats, c = _track_closure(dev, n, 5, cand, W, H, shift)
        assert c.r.map_records == (cand == 'records') and c.r.candidates == (cand is not False)
        V0 = make_viewmat(2.5).to(dev)
        c.load(V0, img, torch.zeros(2, device=dev))
        c.prepare()
        c.load(V0, img, torch.zeros(2, device=dev))
        if cand:
            n_cand, _, _ = c.r.candidate_stats()
            assert (0 < n_cand < n) if cand is True else n_cand == n
        out = []
        for k, (dr, dtr) in enumerate(deltas):
            # move the slot's pose (dR / dt of PoseZhou) and re-evaluate through the captured closure; the optimiser's step is
            # disabled (lr 0) so that the evaluation point is the one set here
            with torch.no_grad():
                c.slots.dR.zero_(); c.slots.dt.zero_()
                c.slots.dR[0, 1] = dr
                c.slots.dt[0, 0] = dtr
            c.slots.forward(c.r.viewmats, torch.cuda.current_stream().cuda_stream)
            c.init_optimizer(10, 0.0, 5, 25)
            c.launch(1)
            torch.cuda.synchronize()
            assert c.r.check_capacity()
            M = int(c.r.M_dev.item())
            flat_ids = c.r.slot_flatten_ids()[c.r.flat[:M].long()]
            out.append((M, c.r.flat[:M].clone(), flat_ids.clone(), c.r.offsets.clone(), c.r.rec.clone(),
                        c.r.pose_ws.clone()[:c.r.pose_blocks * 12 * 4], c.loss_rows.clone()))
            if cand:
                _, mode, fell_back = c.r.candidate_stats()
                assert mode == (1 if k < 3 or cand == 'records' else 0), (k, mode)
                assert fell_back == (0 if k < 3 or cand == 'records' else 1)
        res[cand] = out
    for other, k in [(o, k) for o in (True, 'records') for k in range(len(deltas))]:
        a, b = res[False][k], res[other][k]
        assert a[0] == b[0] > 0
        assert torch.equal(a[1], b[1])                       # the same slots in the same order: tile lists bit-exact
        assert torch.equal(a[2], b[2])                       # ... naming the same (camera, Gaussian) pairs
        assert torch.equal(a[3], b[3])
        live = torch.zeros(a[4].shape[0], dtype=torch.bool, device=dev)
        live[a[1].long()] = True
        assert torch.equal(a[4][live], b[4][live])           # records of every listed instance: same bits
        assert torch.equal(a[6], b[6])                       # loss partials of the fused rasteriser: same bits
        pa, pb = a[5].view(torch.float32), b[5].view(torch.float32)
        assert float((pa - pb).abs().max()) <= 2e-4 * float(pa.abs().max()) + 1e-12      # (float atomics in the backward)


def test_balanced_order_placement_probe(dev):
    """the start-up self-check behind the CU-balanced launch order: on a drained MI355X workgroups i and i + G share a compute
    unit (G = 256); a shape for which that cannot hold (more distinct units asked for than exist) reports failure with a
    reason instead of a silent order"""
    import warnings
    from gslam_amd.plan import placement_ok
    n_cus = int(torch.cuda.get_device_properties(dev).multi_processor_count)
    ok, note = placement_ok(dev, 1200, n_cus)
    assert ok, note
    assert "probe ok" in note
    from gslam_amd.plan import PLACEMENT_RETRIES, _PLACEMENT
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        # a wrong G: the pattern cannot hold.  A failed probe is not believed at once (another stream may have been busy): it
        # is repeated by the next PLACEMENT_RETRIES callers, and only then does the verdict stick - with ONE warning
        for attempt in range(PLACEMENT_RETRIES + 1):
            bad, why = placement_ok(dev, 1200, n_cus + 37)
            assert not bad and "FAILED" in why
            assert len(w) == (1 if attempt == PLACEMENT_RETRIES else 0)
        assert any(k[1:3] == (1200, n_cus + 37) for k in _PLACEMENT)          # now cached
        placement_ok(dev, 1200, n_cus + 37)
        assert len(w) == 1


def _fused_closure_outputs(c, st, H, W):
    tl = (c.img, c.exposure, 1.0 / (H * W), c.loss_rows)
    c.r.v_rec.zero_()
    c.r.forward_track_fused(st, tl)
    torch.cuda.synchronize()
    v_rec = c.r.v_rec.clone()
    c.r.backward(st, rasterised=True)
    torch.cuda.synchronize()
    assert c.r.check_capacity()
    return (c.loss_rows.clone(), c.r.tile_work.clone(), v_rec, c.r.flat[:c.r.last_M].clone(),
            c.r.offsets[:c.r.T + 1].clone())


@pytest.mark.parametrize("near_place", [False, True])
@pytest.mark.parametrize("n_gauss,scale_up", [(80000, 0.4), (300000, 0.9)])
def test_tile_sort_inside_the_rasteriser_equals_the_sort_launch(dev, n_gauss, scale_up, near_place):
    """gsx_raster_track_fused_sorting (the front stops after the placement; every tile's workgroup sorts its keys slab by slab
    of depth in LDS - first up to the tile's cut-off of the previous closure - and composites each slab before the next is
    sorted) against the stand-alone tile sort + gsx_raster_track_fused: whatever part of a tile's list ended up sorted equals
    the full list's prefix entry for entry, and loss rows / gradient records are those of the full sort - in the first closure
    (no cut-off yet: the nearest keys that fit LDS, then more while pixels live), in the second (cut-offs of the first: one
    slab almost everywhere, most keys never sorted) and with cut-offs forced far too tight (an empty first slab everywhere).
    The second scene has tiles of several thousand keys: windows that have to be cut down to what the LDS sort takes.
    near_place (round 5, gsx_front_fwd_near + gsx_raster_track_fused_near): the front does not even WRITE the keys behind a tile's
    cut-off (offsets still describe the full lists); a tile whose pixels outlive the placed keys appends the rest itself from the
    front's instance records.  Same comparisons, plus: the placed counts shrink with the cut-offs, every tile completes itself when
    the cut-offs are far too tight, a pose jump between two closures (cut-offs of the OLD pose, lists of the new one)."""
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.plan import TrackClosure, current_stream_ptr
    from gslam_amd.primitives import Camera
    from gslam_amd.synthetic import make_intrinsics, make_scene, make_viewmat
    W, H = 640, 480
    sc = make_scene(n_gauss, 6)
    sc["scales"] = sc["scales"] + scale_up
    splats = GaussianSplattingData.from_dict(sc, dev)
    cam = Camera(make_intrinsics(W, H).to(dev), H, W)
    img = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(8)).to(dev)
    st = current_stream_ptr(dev)
    ref = TrackClosure(splats, cam, defer_sort=False)
    new = TrackClosure(splats, cam, defer_sort=True, near_place=near_place, row_keys=False)   # (the front builds the tile segments)
    assert new.r.defer_sort and not ref.r.defer_sort and new.r.near_place == near_place
    for c in (ref, new):
        c.load(make_viewmat(2.0).to(dev), img, torch.tensor([0.02, -0.01], device=dev))
        c.r.probe()
    a = _fused_closure_outputs(ref, st, H, W)
    off = a[4].cpu().numpy()
    sizes = off[1:] - off[:-1]
    if n_gauss >= 300000:
        assert sizes.max() > 1152, "the second scene must have segments that do not fit the LDS sort"

    def placed():
        return new.r.tile_placed.cpu().numpy() if near_place else sizes

    def compare(b, what):
        assert torch.equal(a[4], b[4]), what                                          # same offsets
        near = new.r.tile_near.cpu().numpy()
        fa, fb = a[3].cpu().numpy(), b[3].cpu().numpy()
        for t in range(new.r.T):
            lo, n = int(off[t]), int(near[t])
            assert 0 <= n <= sizes[t], (what, t, n, sizes[t])
            assert (fa[lo:lo + n] == fb[lo:lo + n]).all(), f"{what}: tile {t}: sorted near part differs from the full list"
        if what == "first closure":
            assert torch.equal(a[1], b[1]), what                                      # same chunks / trips per tile
        # (later closures: the work counters - weights of the launch order, not results - count the survivors of whole 64-entry
        # chunks; a near list that ends inside a chunk, or a fall-back that starts a chunk of its own, shifts them)
        assert float((a[0] - b[0]).abs().max()) <= 2e-6 * float(a[0].abs().max()), what
        assert float(a[2].abs().max()) > 0
        assert float((a[2] - b[2]).abs().max()) < 1e-4 * float(a[2].abs().max()), what
        return near

    near1 = compare(_fused_closure_outputs(new, st, H, W), "first closure")
    assert (near1 <= sizes).all() and near1.max() > 0
    assert (placed() == sizes).all()                                                  # no cut-off yet: every key placed
    cuts = new.r.tile_cut.clone()
    assert int((cuts != 0x7f800000).sum()) > new.r.T // 2                             # cut-offs left for the next closure
    new.r.sort_stats.zero_()
    near2 = compare(_fused_closure_outputs(new, st, H, W), "second closure")
    stats = new.r.sort_stats.cpu().tolist()
    # most keys stay unsorted (the first scene is thin - few pixels saturate early - and the near placement's margin is 0.5)
    frac = 0.8 if (n_gauss >= 300000 or not near_place) else 0.97
    assert near2.sum() < frac * sizes.sum(), (near2.sum(), sizes.sum())
    assert stats[0] <= new.r.T // 20, stats                                           # same pose: the cut-offs hold
    assert stats[3] == 0, stats
    if near_place:
        p2 = placed()
        assert (p2 <= sizes).all() and p2.sum() < frac * sizes.sum(), (p2.sum(), sizes.sum())  # ... and are not even written
        assert stats[2] <= new.r.T // 20, stats                                       # hardly any tile had to append its far keys
        assert (near2 >= np.minimum(p2, near2)).all()
    # a pose jump: the cut-offs are those of the old pose, the lists those of the new one (tiles whose surfaces moved away
    # sort further slabs / append their far keys); the reference closure moves with it
    cuts2 = new.r.tile_cut.clone()
    V2 = make_viewmat(2.0)
    cth, sth = float(np.cos(0.06)), float(np.sin(0.06))
    Rj = torch.tensor([[cth, 0.0, sth, 0.0], [0.0, 1.0, 0.0, 0.0], [-sth, 0.0, cth, 0.0], [0.0, 0.0, 0.0, 1.0]])
    V2 = Rj @ V2
    V2[0, 3] += 0.05
    for c in (ref, new):
        c.r.viewmats[0].copy_(V2.to(dev))
    a = _fused_closure_outputs(ref, st, H, W)
    off = a[4].cpu().numpy()
    sizes = off[1:] - off[:-1]
    new.r.sort_stats.zero_()
    nearj = compare(_fused_closure_outputs(new, st, H, W), "pose jump")
    stats = new.r.sort_stats.cpu().tolist()
    assert stats[3] == 0, stats
    if n_gauss < 300000:
        assert stats[0] + stats[2] > 0, stats                                         # some cut-offs failed, all were repaired
    assert nearj.max() > 0
    # cut-offs far too tight: the first slab of every tile is empty, the tiles whose pixels composite anything take a second
    # (near placement: NOTHING is placed, every tile with keys appends all of them itself)
    new.r.tile_cut.copy_(torch.full_like(cuts, 0x3a83126f))                           # depth 0.001
    new.r.sort_stats.zero_()
    near3 = compare(_fused_closure_outputs(new, st, H, W), "cut-offs too tight")
    stats = new.r.sort_stats.cpu().tolist()
    assert stats[0] >= int((nearj > 0).sum()) * 0.9, stats
    assert (near3 > 0).sum() >= (nearj > 0).sum() * 0.9
    assert stats[1] == 0 and stats[3] == 0, stats                                     # nothing went through memory
    if near_place:
        assert int(placed().sum()) == 0 and stats[2] == int((sizes > 0).sum()), (stats, int((sizes > 0).sum()))
        # the segments now hold every key of their tiles: as multisets they are the reference's lists
        lay = (C.c_int64 * 3)()
        from gslam_amd._lib import lib as _l
        _l.gsx_front_keys(new.r.N, new.r.C, new.r.tile_w, new.r.tile_h, new.r.capacity, 32, lay)
        M = int(off[-1])
        keys = new.r.isect_ws[int(lay[0]):int(lay[0]) + 8 * M].view(torch.int64).cpu().numpy()
        fa = a[3].cpu().numpy()
        for t in range(0, new.r.T, 7):
            lo, hi = int(off[t]), int(off[t + 1])
            assert sorted((keys[lo:hi] & 0xFFFFFFFF).tolist()) == sorted(fa[lo:hi].tolist()), f"tile {t}: key multiset differs"


def test_tile_sort_inside_the_rasteriser_with_piles_of_equal_depths(dev):
    """the slow path of the sort inside the rasteriser: thousands of Gaussians on ONE plane perpendicular to the optical axis have
    the same depth bits, no depth window can cut such a pile down to what the LDS sort takes, and the tile's workgroup falls back
    to the merge sort through memory (sort_stats[1] counts those tiles).  Ties are broken by id in both implementations: the
    consumed prefix is still the stand-alone sort's list entry for entry, and so are loss rows and gradient records.  A second
    closure (cut-offs of the first = the one depth there is) and an almost transparent variant (pixels never saturate: the whole
    pile is composited) take the same path."""
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.plan import TrackClosure, current_stream_ptr
    from gslam_amd.primitives import Camera
    from gslam_amd.synthetic import make_intrinsics, make_scene
    W, H = 640, 480
    for opac_shift in (0.0, -3.0):
        sc = make_scene(150000, 9)
        g = torch.Generator().manual_seed(4)
        sc["means"][:, 0] = (torch.rand(150000, generator=g) - 0.5) * 4.0
        sc["means"][:, 1] = (torch.rand(150000, generator=g) - 0.5) * 3.0
        sc["means"][:, 2] = 0.0                                     # one plane ...
        sc["scales"] = sc["scales"] + 0.9
        sc["opacities"] = sc["opacities"] + opac_shift
        splats = GaussianSplattingData.from_dict(sc, dev)
        cam = Camera(make_intrinsics(W, H).to(dev), H, W)
        V = torch.eye(4)
        V[2, 3] = 3.0                                               # ... seen head-on from 3 m: every depth is exactly 3.0
        img = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(8)).to(dev)
        st = current_stream_ptr(dev)
        ref = TrackClosure(splats, cam, defer_sort=False)
        new = TrackClosure(splats, cam, defer_sort=True, row_keys=False)
        for c in (ref, new):
            c.load(V.to(dev), img, torch.zeros(2, device=dev))
            c.r.probe()
        a = _fused_closure_outputs(ref, st, H, W)
        off = a[4].cpu().numpy()
        sizes = off[1:] - off[:-1]
        assert sizes.max() > 1152, sizes.max()
        for closure in range(2):
            new.r.sort_stats.zero_()
            b = _fused_closure_outputs(new, st, H, W)
            stats = new.r.sort_stats.cpu().tolist()
            assert stats[1] > 0, stats                              # piles went through the memory merge sort
            assert torch.equal(a[4], b[4])
            near = new.r.tile_near.cpu().numpy()
            fa, fb = a[3].cpu().numpy(), b[3].cpu().numpy()
            for t in range(new.r.T):
                lo, n = int(off[t]), int(near[t])
                assert 0 <= n <= sizes[t]
                assert (fa[lo:lo + n] == fb[lo:lo + n]).all(), f"closure {closure}: tile {t}: sorted part differs from the full list"
            assert float((a[0] - b[0]).abs().max()) <= 2e-6 * float(a[0].abs().max()) + 1e-12
            assert float((a[2] - b[2]).abs().max()) <= 1e-4 * float(a[2].abs().max()) + 1e-12


def _row_keys_outputs(c, st, H, W):
    tl = (c.img, c.exposure, 1.0 / (H * W), c.loss_rows)
    c.r.v_rec.zero_()
    c.r.forward_track_fused(st, tl)
    torch.cuda.synchronize()
    v_rec = c.r.v_rec.clone()
    c.r.backward(st, rasterised=True)
    torch.cuda.synchronize()
    assert c.r.check_capacity()
    return (c.loss_rows.clone(), c.r.tile_work.clone(), v_rec, c.r.flat.clone(), c.r.tile_near.clone(),
            c.r.pose_ws.view(torch.float32)[:c.r.pose_blocks * 12].clone())


@pytest.mark.parametrize("n_gauss,scale_up,planes", [(80000, 0.4, False), (300000, 0.9, False), (150000, 0.9, True)])
def test_tiles_collect_their_keys_from_the_projection_rows(dev, n_gauss, scale_up, planes):
    """GSX_PROJ_ROW_KEYS + gsx_raster_track_fused_rows (round 5): the front ends with its projection launch - every projection
    workgroup leaves its row's keys grouped by tile in the row's own segment - and the rasteriser's tile workgroups collect their
    keys themselves (one atomic on the render's key counter hands out the segments).  Against the front that builds the tile
    segments (count matrix -> column scan -> placement): the same number of keys in every tile, the same keys (as multisets: the
    segments are unsorted), the consumed prefix of every tile's sorted list entry for entry, loss rows bit for bit, gradient
    records and pose partials to float-atomic noise, M equal - in a first closure (no cut-offs), a second one, after a pose jump
    (cut-offs of the old pose: further slabs are read from the tile's own copy of its keys) and with cut-offs far too tight; on a
    scene whose tiles outgrow the LDS sort; on piles of equal depths (the through-memory sort over the collected segment); and
    through the captured closure: same end point of ten Adam steps."""
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.plan import TrackClosure, current_stream_ptr
    from gslam_amd.primitives import Camera
    from gslam_amd.synthetic import make_intrinsics, make_scene, make_viewmat
    from gslam_amd._lib import lib as _l
    W, H = 640, 480
    sc = make_scene(n_gauss, 6)
    sc["scales"] = sc["scales"] + scale_up
    V0 = make_viewmat(2.0)
    if planes:
        g = torch.Generator().manual_seed(4)
        sc["means"][:, 0] = (torch.rand(n_gauss, generator=g) - 0.5) * 4.0
        sc["means"][:, 1] = (torch.rand(n_gauss, generator=g) - 0.5) * 3.0
        sc["means"][:, 2] = 0.0
        V0 = torch.eye(4)
        V0[2, 3] = 3.0
    splats = GaussianSplattingData.from_dict(sc, dev)
    cam = Camera(make_intrinsics(W, H).to(dev), H, W)
    img = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(8)).to(dev)
    st = current_stream_ptr(dev)
    ref = TrackClosure(splats, cam, defer_sort=True, row_keys=False)
    new = TrackClosure(splats, cam, defer_sort=True)
    assert new.r.row_keys and not ref.r.row_keys and new.r.tile_exact and ref.r.tile_exact
    for c in (ref, new):
        c.load(V0.to(dev), img, torch.tensor([0.02, -0.01], device=dev))
        c.r.probe()
    lay = (C.c_int64 * 3)()
    _l.gsx_front_keys(new.r.N, new.r.C, new.r.tile_w, new.r.tile_h, new.r.capacity, 32, lay)

    def compare(what, check_keys=False):
        a, b = _row_keys_outputs(ref, st, H, W), _row_keys_outputs(new, st, H, W)
        off = ref.r.offsets[:ref.r.T + 1].cpu().numpy()
        sizes = off[1:] - off[:-1]
        span = new.r.tile_span.cpu().numpy()
        assert (span[:, 1] == sizes).all(), what                                   # every tile collected what the front would place
        assert new.r.keys_total() == ref.r.keys_total() == int(sizes.sum()), what
        assert torch.equal(a[4], b[4]), what                                       # same consumed prefix lengths
        near = b[4].cpu().numpy()
        fa, fb = a[3].cpu().numpy(), b[3].cpu().numpy()
        for t in range(new.r.T):
            n = int(near[t])
            assert (fa[off[t]:off[t] + n] == fb[span[t, 0]:span[t, 0] + n]).all(), f"{what}: tile {t}: consumed prefix differs"
        assert torch.equal(a[0], b[0]), what                                       # loss rows: same lists, same arithmetic
        assert torch.equal(a[1], b[1]), what                                       # work counters
        assert float(a[2].abs().max()) > 0
        assert float((a[2] - b[2]).abs().max()) < 1e-4 * float(a[2].abs().max()), what
        assert float((a[5] - b[5]).abs().max()) < 2e-4 * float(a[5].abs().max()) + 1e-12, what
        if check_keys:
            # the tiles' own copies of their keys, as multisets (written only by tiles that scanned their keys a second time - here:
            # every tile that composited anything, its first window was empty)
            ka = ref.r.isect_ws[int(lay[0]):int(lay[0]) + 8 * int(off[-1])].view(torch.int64).cpu().numpy()
            kb = new.r.isect_ws[int(lay[0]):int(lay[0]) + 8 * new.r.capacity].view(torch.int64).cpu().numpy()   # (segments anywhere)
            checked = 0
            for t in range(0, new.r.T, 5):
                if near[t] > 0:
                    assert sorted(ka[off[t]:off[t + 1]].tolist()) == sorted(kb[span[t, 0]:span[t, 0] + span[t, 1]].tolist()), (what, t)
                    checked += 1
            assert checked > 20, checked
        return sizes, near

    sizes, near1 = compare("first closure")
    if n_gauss >= 300000:
        assert sizes.max() > 1152
    new.r.sort_stats.zero_()
    sizes, near2 = compare("second closure")
    assert planes or near2.sum() < 0.97 * sizes.sum()
    if not planes:
        V2 = V0.clone()
        cth, sth = float(np.cos(0.06)), float(np.sin(0.06))
        V2 = torch.tensor([[cth, 0.0, sth, 0.0], [0.0, 1.0, 0.0, 0.0], [-sth, 0.0, cth, 0.0], [0.0, 0.0, 0.0, 1.0]]) @ V2
        V2[0, 3] += 0.05
        for c in (ref, new):
            c.r.viewmats[0].copy_(V2.to(dev))
        compare("pose jump")
        assert int(new.r.sort_stats[0].item()) > 0 or n_gauss >= 300000            # cut-offs failed, later slabs were sorted
    for c in (ref, new):
        c.r.tile_cut.fill_(0x3a83126f)                                             # depth 0.001: an empty first slab everywhere
    compare("cut-offs too tight", check_keys=True)
    if planes:
        assert int(new.r.sort_stats[1].item()) > 0                                 # piles went through the memory merge sort
    # the whole closure, captured
    for c in (ref, new):
        c.load(V0.to(dev), img, torch.tensor([0.02, -0.01], device=dev))
        c.prepare()
        c.load(V0.to(dev), img, torch.tensor([0.02, -0.01], device=dev))
        c.init_optimizer(12, 1e-3, 5, 8)
        c.launch(10)
    torch.cuda.synchronize()
    assert ref.r.check_capacity() and new.r.check_capacity()
    ra, rb = ref.read_report().cpu(), new.read_report().cpu()
    # (same lists, same per-pixel arithmetic; the gradient records are summed by float atomics in another order.  Ten closures inside
    # the Adam warm-up: smooth steps, no discrete line-search decision that the last bits of a sum could flip)
    assert torch.equal(ra[:4], rb[:4]) and float(ra[7]) == float(rb[7]) == 10.0, (ra, rb)
    assert abs(float(ra[4]) - float(rb[4])) <= 1e-3 * abs(float(ra[4])) + 1e-12, (ra, rb)
    assert float((ref.r.viewmats - new.r.viewmats).abs().max()) < 2e-3


@pytest.mark.parametrize("n_gauss,scale_up,low_opacity", [(120000, 0.5, False), (300000, 0.9, False), (100000, 1.2, True)])
def test_exact_tile_test_drops_only_pairs_no_pixel_can_see(dev, n_gauss, scale_up, low_opacity):
    """GSX_PROJ_TILE_EXACT (round 5): the fused front of a pose-only closure lists an instance only in the tiles of its 3-sigma square
    that hold a pixel centre inside the bounding box of its alpha >= 1/255 ellipse.  Against the same closure with every tile of the
    square listed: the loss rows bit for bit (the forward composites the same entries in the same order), pose partials to
    float-atomic noise, fewer keys; every tile's keys a sub-multiset of the full list's; and every pair that was dropped checked pixel
    by pixel in float64 from its record - no pixel centre of the tile reaches 1/255 (with a scene of low opacities, where whole
    instances vanish, too)."""
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.plan import TrackClosure, current_stream_ptr
    from gslam_amd.primitives import Camera
    from gslam_amd.synthetic import make_intrinsics, make_scene, make_viewmat
    from gslam_amd._lib import lib as _l
    W, H = 640, 480
    sc = make_scene(n_gauss, 9)
    sc["scales"] = sc["scales"] + scale_up
    if low_opacity:
        sc["opacities"] = sc["opacities"] - 4.0                # logits: most opacities between 0.002 and 0.1
    splats = GaussianSplattingData.from_dict(sc, dev)
    cam = Camera(make_intrinsics(W, H).to(dev), H, W)
    img = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(8)).to(dev)
    st = current_stream_ptr(dev)
    full = TrackClosure(splats, cam, defer_sort=True, tile_exact=False)
    new = TrackClosure(splats, cam, defer_sort=True)
    assert new.r.tile_exact and not full.r.tile_exact and full.r.row_keys
    V0 = make_viewmat(2.0)
    for c in (full, new):
        c.load(V0.to(dev), img, torch.tensor([0.02, -0.01], device=dev))
        c.r.probe()
    lay = (C.c_int64 * 3)()
    _l.gsx_front_keys(new.r.N, new.r.C, new.r.tile_w, new.r.tile_h, new.r.capacity, 32, lay)
    for what in ("first closure", "second closure", "cut-offs too tight"):
        if what == "cut-offs too tight":
            for c in (full, new):
                c.r.tile_cut.fill_(0x3a83126f)                 # depth 0.001: every tile scans its keys twice and keeps its own copy
        a, b = _row_keys_outputs(full, st, H, W), _row_keys_outputs(new, st, H, W)
        assert torch.equal(a[0], b[0]), what                   # loss rows
        # (instances no tile lists any more have no slot: the workgroups of the pose backward cut the rest differently - the sums agree)
        pa, pb = a[5].view(-1, 12).double().sum(0), b[5].view(-1, 12).double().sum(0)
        assert float((pa - pb).abs().max()) < 2e-4 * float(pa.abs().max()) + 1e-12, (what, pa, pb)
        sa, sb = full.r.tile_span.cpu().numpy(), new.r.tile_span.cpu().numpy()
        assert (sb[:, 1] <= sa[:, 1]).all() and new.r.keys_total() == int(sb[:, 1].sum()), what
        assert new.r.keys_total() < 0.9 * full.r.keys_total(), (what, new.r.keys_total(), full.r.keys_total())
    near = b[4].cpu().numpy()
    ka = full.r.isect_ws[int(lay[0]):int(lay[0]) + 8 * full.r.capacity].view(torch.int64).cpu().numpy()
    kb = new.r.isect_ws[int(lay[0]):int(lay[0]) + 8 * new.r.capacity].view(torch.int64).cpu().numpy()
    rec = full.r.rec.view(-1, 12).cpu().numpy().astype(np.float64)
    px = np.arange(16) + 0.5
    checked = dropped = 0
    worst = 0.0
    for t in range(0, new.r.T, 7):
        if near[t] == 0:
            continue
        la = ka[sa[t, 0]:sa[t, 0] + sa[t, 1]]
        lb = kb[sb[t, 0]:sb[t, 0] + sb[t, 1]]
        da, db = np.sort(la >> 32), np.sort(lb >> 32)
        # sub-multiset by depth bits
        rest = list(da)
        import collections
        cnt_a, cnt_b = collections.Counter(da.tolist()), collections.Counter(db.tolist())
        assert all(cnt_a[k] >= v for k, v in cnt_b.items()), t
        tx, ty = t % new.r.tile_w, t // new.r.tile_w
        gone = cnt_a - cnt_b
        for key in la.tolist():
            d = key >> 32
            if gone.get(d, 0) > 0 and cnt_a[d] == gone[d]:      # (an unambiguous depth: this very key was dropped)
                slot = key & 0xffffffff
                mx, my, c0, c1, c2, op = rec[slot, :6]
                dx, dy = tx * 16 + px - mx, ty * 16 + px - my
                sig = 0.5 * (c0 * dx[None, :] ** 2 + c2 * dy[:, None] ** 2) + c1 * dx[None, :] * dy[:, None]
                alpha = op * np.exp(-sig)
                worst = max(worst, float(alpha.max()))
                assert float(alpha.max()) < 1.0 / 255.0, (t, slot, float(alpha.max()))
                dropped += 1
        checked += 1
    assert checked > 20 and dropped > 200, (checked, dropped)
    # the whole closure, captured: where the optimiser ends up
    for c in (full, new):
        c.load(V0.to(dev), img, torch.tensor([0.02, -0.01], device=dev))
        c.prepare()
        c.load(V0.to(dev), img, torch.tensor([0.02, -0.01], device=dev))
        c.init_optimizer(12, 1e-3, 5, 8)                       # (ten closures inside the Adam warm-up: no line-search decision to flip)
        c.launch(10)
    torch.cuda.synchronize()
    assert full.r.check_capacity() and new.r.check_capacity()
    ra, rb = full.read_report().cpu(), new.read_report().cpu()
    assert torch.equal(ra[:4], rb[:4]) and float(ra[7]) == float(rb[7]) == 10.0, (ra, rb)
    assert abs(float(ra[4]) - float(rb[4])) <= 1e-3 * abs(float(ra[4])) + 1e-12, (ra, rb)
    assert float((full.r.viewmats - new.r.viewmats).abs().max()) < 2e-3


@pytest.mark.parametrize("n_gauss,n_cams,W,H", [(60000, 3, 640, 480), (200000, 8, 640, 480), (30000, 2, 325, 245)])
def test_window_closure_on_the_tracking_machinery_equals_the_generic_path(dev, n_gauss, n_cams, W, H):
    """WindowClosure(fused=True) (round 5): the closure of Backend.optimize_poses_lbfgs (gslam/backend.py:465-504) on the fused front
    with row keys, the tile sort inside the rasteriser and forward + the refiner's loss (err^2 / (2 beta^2) + log(beta)^2 / 2) +
    geometry-only backward in one launch - against the generic chain (projection, binning, sort, rasteriser forward, gsx_map_loss,
    rasteriser backward, projection backward): loss and the gradient of every pose parameter of every camera, at the window's
    poses, at perturbed ones (cut-offs of the previous evaluation), with and without the map's records, on an image whose size
    is no multiple of the tile; then the captured closures with the device L-BFGS: same place after six evaluations."""
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.plan import WindowClosure, current_stream_ptr
    from gslam_amd.primitives import Camera, Frame, PoseZhou
    from gslam_amd.synthetic import make_intrinsics, make_scene, make_viewmat
    sc = make_scene(n_gauss, 3)
    sc["scales"] = sc["scales"] + 0.5
    splats = GaussianSplattingData.from_dict(sc, dev)
    cam = Camera(make_intrinsics(W, H).to(dev), H, W)
    g = torch.Generator().manual_seed(5)
    window = []
    for i in range(n_cams):
        pose = PoseZhou(make_viewmat(0.6 * i).to(dev)).to(dev)
        window.append(Frame(img=torch.rand(H, W, 3, generator=g).to(dev), timestamp=float(i), camera=cam, pose=pose,
                            gt_pose=make_viewmat(0.6 * i).to(dev), index=i,
                            exposure_params=torch.tensor([0.03 * i, -0.01 * i], device=dev)))
    learnable = [i > 0 for i in range(n_cams)]
    ref = WindowClosure(splats, [cam] * n_cams, learnable, fused=False)
    new = WindowClosure(splats, [cam] * n_cams, learnable, fused=True)
    assert new.fused and not ref.fused and new.r.row_keys and new.r.map_records
    st = current_stream_ptr(dev)
    for c in (ref, new):
        c.load(window)
        c.slots.forward(c.r.viewmats, st)
        c.r.probe()
    for k in range(4):
        if k == 2:
            new.r.build_candidates()                     # (evaluations 0 and 1 ran without records of the map: full projection path)
        if k:
            with torch.no_grad():
                for c in (ref, new):
                    c.slots.dt[1:, 0] += 0.004 * k
                    c.slots.dR[1:, 1] += 0.003
        for c in (ref, new):
            c.enqueue(st, advance=False)
        torch.cuda.synchronize()
        assert ref.r.check_capacity() and new.r.check_capacity()
        la, lb = float(ref.out2[0]), float(new.out2[0])
        assert abs(la - lb) <= 2e-6 * abs(la), (k, la, lb)
        for name in ("v_dt", "v_dR"):
            a, b = getattr(ref.slots, name), getattr(new.slots, name)
            assert float(a.abs().max()) > 0
            assert float((a - b).abs().max()) <= 3e-4 * float(a.abs().max()), (k, name, a, b)
        assert new.r.candidate_stats()[1] == (1 if k >= 2 else 0)
    # captured, with the device optimiser
    for c in (ref, new):
        c.load(window)
        c.prepare()
        c.load(window)
        c.init_optimizer(8)
        c.launch(6)
    torch.cuda.synchronize()
    assert ref.r.check_capacity() and new.r.check_capacity()
    ra, rb = ref.read_report().cpu(), new.read_report().cpu()
    # (six closures of a line search: its discrete decisions may flip on the last bits of a float-atomic sum - compared is where the
    # two ended up, loosely; the closures themselves are compared above to 2e-6 / 3e-4)
    assert abs(float(ra[5]) - float(rb[5])) <= 2e-2 * abs(float(ra[5])) + 1e-12, (ra, rb)
    assert float((ref.r.viewmats - new.r.viewmats).abs().max()) < 1e-2


def test_projection_backward_range_by_range_equals_one_launch(dev):
    """gsx_project_bwd_range over ranges that tile the map (what the ranged multi-GPU exchange issues) leaves the
    six map gradients and the pose partials exactly as ONE gsx_project_bwd over the same gradient records - bit for bit (one
    thread per Gaussian, no atomics); a range that is not made of whole workgroups is refused"""
    from gslam_amd._lib import GsxError
    from gslam_amd.mapping import BundleAdjuster
    from gslam_amd.plan import current_stream_ptr
    m, cam, frame = _world(dev, n=7000)
    window = [frame(i, i) for i in range(3)]
    plan = BundleAdjuster(m, capturable=True).plan(window)
    plan.render_backward()                                    # keep = True: the gradient records stay as accumulated
    torch.cuda.synchronize()
    assert plan.capacity_ok()
    r, st = plan.r, current_stream_ptr(dev)
    r.backward_project(st, keep=True)                         # ONE launch over the records (no isotropic term on top)
    torch.cuda.synchronize()
    want = {k: v.clone() for k, v in plan.grad_views.items()}
    want_pose = r.pose_ws.clone()
    assert all(float(v.abs().max()) > 0 for v in want.values())
    for rows_list in ([(0, 2816), (2816, 5632), (5632, 7000)], [(0, 256), (256, 6912), (6912, 7000)], [(0, 7000)]):
        for v in plan.grad_views.values():
            v.fill_(7.0)
        r.pose_ws.fill_(3)
        for rows in rows_list:
            r.backward_project(st, keep=True, rows=rows)
        torch.cuda.synchronize()
        for k, v in plan.grad_views.items():
            assert torch.equal(v, want[k]), (k, rows_list)
        nb = r.pose_blocks * r.C * 12 * 4
        assert torch.equal(r.pose_ws[:nb], want_pose[:nb]), rows_list
    with pytest.raises(GsxError):
        r.backward_project(st, keep=True, rows=(100, 7000))
    with pytest.raises(GsxError):
        r.backward_project(st, keep=True, rows=(0, 1000))


def test_range_copy_gathers_and_scatters_the_parts_of_a_range(dev):
    """gsx_range_copy (staging copies of the ranged exchange) against the same copy written with views"""
    from gslam_amd import dist as gdist
    n, world, K = 5000, 4, 3
    shapes = [(n, 3), (n, 4), (n, 3), (n,), (n, 3), (n,)]
    for rank in (0, 3):
        b = gdist.StepBucket(shapes, 2, dev, world=world, rank=rank, ranges=K)
        h = gdist.StepBucket(shapes, 2, "cpu", world=world, rank=rank, ranges=K)
        whole = torch.randn(b.S_pad, generator=torch.Generator().manual_seed(rank))
        for k in range(b.ranges):
            for own in (False, True):
                sd = torch.full(((1 if own else world) * b.Lk,), -1.0, device=dev)
                sh = torch.full(((1 if own else world) * b.Lk,), -1.0)
                b._range_copy(whole.to(dev), k, sd, own_only=own, to_flat=False)
                h._range_copy(whole, k, sh, own_only=own, to_flat=False)
                assert torch.equal(sd.cpu(), sh) and bool((sh != -1.0).all())
                back_d, back_h = torch.zeros(b.S_pad, device=dev), torch.zeros(b.S_pad)
                b._range_copy(back_d, k, sd, own_only=own, to_flat=True)
                h._range_copy(back_h, k, sh, own_only=own, to_flat=True)
                assert torch.equal(back_d.cpu(), back_h)
                # what came back is the range's rows (all parts, or this rank's) and nothing else
                mask = back_h != 0
                assert torch.equal(back_h[mask], whole[mask]) and int(mask.sum()) >= (1 if own else world) * b.Lk - 8


@pytest.mark.parametrize("n,n_cams", [(9000, 3), (150000, 8)])
def test_tight_tile_lists_of_the_generic_chain_change_no_output(dev, n, n_cams):
    """gsx_project_fwd_rects | GSX_PROJ_TILE_EXACT -> gsx_isect_bin_sort_rects (round 5) in the BA plan (direct binning at 9 k x 3, the spatial pre-sort at 150 k x 8): against the
    plan with the reference's full 3-sigma squares - render, alphas and loss bit for bit (the same entries are composited in the
    same order), all six map gradients and the pose gradients to float-atomic noise, fewer keys."""
    from gslam_amd.mapping import BundleAdjuster
    from gslam_amd.plan import RenderPlan
    res = []
    for tight in (False, True):
        m, cam, frame = _world(dev, n=n)
        window = [frame(i, i) for i in range(n_cams)]
        old = RenderPlan.TIGHT_LISTS
        RenderPlan.TIGHT_LISTS = tight
        try:
            plan = BundleAdjuster(m, capturable=True).plan(window)
        finally:
            RenderPlan.TIGHT_LISTS = old
        assert plan.r.tight_lists == (tight and not plan.r.front) and plan.r.tile_exact == (tight and plan.r.front)
        total, pm = plan.render_backward()
        torch.cuda.synchronize()
        assert plan.capacity_ok()
        res.append((plan.r.render.clone(), plan.r.alphas.clone(), float(total), float(pm),
                    {k: v.clone() for k, v in plan.grad_views.items()}, plan.g_dR.clone(), plan.g_dt.clone(),
                    plan.r.keys_total(), plan.r.radii.clone()))
    a, b = res
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3]
    assert torch.equal(a[8], b[8])
    for k in a[4]:
        assert float((a[4][k] - b[4][k]).abs().max()) < 2e-4 * float(a[4][k].abs().max()) + 1e-10, k
    assert float((a[5] - b[5]).abs().max()) < 1e-3 * float(a[5].abs().max()) + 1e-9
    assert float((a[6] - b[6]).abs().max()) < 1e-3 * float(a[6].abs().max()) + 1e-9
    assert b[7] < 0.9 * a[7], (a[7], b[7])


@pytest.mark.parametrize("W,H,n_cams", [(320, 240, 3), (325, 245, 2), (640, 480, 1)])
def test_ssim_backward_and_loss_block_in_one_launch_equal_the_two_launches(dev, W, H, n_cams):
    """gsx_ssim_bwd_map_loss (round 5) against gsx_ssim_bwd + gsx_map_loss inside the BA plan: d loss / d render bit for bit (the
    pixel's expressions are shared, csrc/loss_pixel.h), loss values and exposure gradients to the rounding of a different grouping
    of the partial sums (one row per 32 x 16 tile instead of one per 256 pixels), map gradients to float-atomic noise - also on an
    image that is not made of whole tiles"""
    from gslam_amd.mapping import BundleAdjuster
    from gslam_amd.plan import MappingStep
    res = []
    for fused in (False, True):
        m, cam, frame = _world(dev, n=8000, W=W, H=H)
        window = [frame(i, i) for i in range(n_cams)]
        for i, f in enumerate(window):
            f.exposure_params = torch.tensor([0.03 * i, -0.02 * i], device=dev)
        old = MappingStep.FUSE_SSIM_LOSS
        MappingStep.FUSE_SSIM_LOSS = fused
        try:
            plan = BundleAdjuster(m, capturable=True).plan(window)
            total, pm = plan.render_backward()
            torch.cuda.synchronize()
        finally:
            MappingStep.FUSE_SSIM_LOSS = old
        assert plan.capacity_ok() and plan.conf.ssim_weight > 0
        res.append((plan.r.v_render.clone(), float(total), float(pm), plan.g_exposure.clone(),
                    {k: v.clone() for k, v in plan.grad_views.items()}))
    a, b = res
    assert float(a[0].abs().max()) > 0 and torch.equal(a[0], b[0])
    assert abs(a[1] - b[1]) <= 2e-6 * abs(a[1]) and abs(a[2] - b[2]) <= 2e-6 * abs(a[2]), (a[1], b[1], a[2], b[2])
    assert float((a[3] - b[3]).abs().max()) <= 1e-5 * float(a[3].abs().max()) + 1e-12
    for k in a[4]:
        assert float((a[4][k] - b[4][k]).abs().max()) < 2e-4 * float(a[4][k].abs().max()) + 1e-10, k


def test_closure_tail_inside_the_pose_backward_launch_equals_the_tail_launch(dev):
    """gsx_front_pose_bwd_tail (round 5): the closure's tail run by the last workgroup of the pose backward launch, against
    gsx_front_pose_bwd + gsx_track_opt_tail.  One closure from the same state: loss, pose gradient (the optimiser's g), the new
    parameters and the next view matrix agree to the rounding of a different summation tree over the partial rows; the ticket
    counters are back at zero; ten captured closures of the Adam warm-up: same end point, the state machine's counters equal."""
    from gslam_amd.plan import TrackClosure, current_stream_ptr
    splats, _ = _track_closure(dev, 150000, 3, candidates=False)
    from gslam_amd.primitives import Camera
    from gslam_amd.synthetic import make_intrinsics, make_viewmat
    W, H = 640, 480
    cam = Camera(make_intrinsics(W, H).to(dev), H, W)
    img = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(8)).to(dev)
    V0 = make_viewmat(2.0).to(dev)
    a = TrackClosure(splats, cam, merge_tail=False)
    b = TrackClosure(splats, cam, merge_tail=True)
    assert b.merge_tail and not a.merge_tail and b.r.row_keys
    st = current_stream_ptr(dev)
    out = []
    for c in (a, b):
        c.load(V0, img, torch.tensor([0.02, -0.01], device=dev))
        c.prepare()
        c.load(V0, img, torch.tensor([0.02, -0.01], device=dev))
        c.init_optimizer(4, 1e-3, 5, 25)
        c.enqueue(st)
        torch.cuda.synchronize()
        out.append((c.read_report().cpu().clone(), c.r.viewmats.clone(), c.slots.dt.clone(), c.slots.dR.clone(), c.exposure.clone()))
    ra, rb = out[0][0], out[1][0]
    assert abs(float(ra[4]) - float(rb[4])) <= 1e-5 * abs(float(ra[4])), (ra, rb)          # the closure's loss
    assert torch.equal(ra[:4], rb[:4]) and float(ra[7]) == float(rb[7])                        # phase, evaluations, iterations, Adam step
    for k in (1, 2, 3, 4):
        d = float((out[0][k] - out[1][k]).abs().max())
        assert d <= 1e-5 * float(out[0][k].abs().max()) + 1e-7, (k, d)
    assert int(b.tail_tickets.abs().sum()) == 0
    # (ten captured closures inside the Adam warm-up: smooth steps, no discrete line-search decision that the last bits of a sum
    # could flip - the end points agree closely; the state machine's counters agree exactly)
    for c in (a, b):
        c.load(V0, img, torch.tensor([0.02, -0.01], device=dev))
        c.init_optimizer(12, 1e-3, 5, 25)
        c.launch(10)
    torch.cuda.synchronize()
    assert a.r.check_capacity() and b.r.check_capacity() and int(b.tail_tickets.abs().sum()) == 0
    ra, rb = a.read_report().cpu(), b.read_report().cpu()
    assert torch.equal(ra[:4], rb[:4]) and float(ra[7]) == float(rb[7]) == 10.0, (ra, rb)
    assert abs(float(ra[4]) - float(rb[4])) <= 1e-3 * abs(float(ra[4])) + 1e-12, (ra, rb)
    assert float((a.r.viewmats - b.r.viewmats).abs().max()) < 2e-3


def test_lean_rows_of_the_ba_projection_change_no_result_and_as_output_restores_the_zeros(dev):
    """RenderPlan.lean_rows (round 5): with packed rectangles the projection of a BA plan writes radii = 0 / rects = 0 for a culled
    (camera, Gaussian) and nothing else.  Against the plan that writes every row: render, loss and v_render bit for bit, gradients
    to float-atomic noise - over iterations in which Gaussians LEAVE the frustum (stale rows behind them) - and as_output() shows
    the reference's zeros in every culled row of means2d / depths / conics while the gradient records of the densification
    iteration stay as accumulated."""
    from gslam_amd.mapping import BundleAdjuster
    from gslam_amd.plan import RenderPlan
    res = []
    for lean in (False, True):
        m, cam, frame = _world(dev, n=150000)
        window = [frame(i, i) for i in range(8)]
        old = RenderPlan.LEAN_ROWS
        RenderPlan.LEAN_ROWS = lean
        try:
            plan = BundleAdjuster(m, capturable=True).plan(window)
        finally:
            RenderPlan.LEAN_ROWS = old
        assert plan.r.lean_rows == lean and not plan.r.front and plan.r.tight_lists
        total, pm = plan.render_backward()                    # from identical maps: compared below
        torch.cuda.synchronize()
        first = (plan.r.render.clone(), plan.r.v_render.clone(), float(total), float(pm),
                 {k: v.clone() for k, v in plan.grad_views.items()}, plan.r.radii.clone())
        plan.step()
        plan.step()
        with torch.no_grad():                                 # the cameras turn: part of the map leaves every frustum
            for f in window:
                f.pose.dR.data[0] += 0.12
        plan.render_backward()
        torch.cuda.synchronize()
        assert plan.capacity_ok()
        vrec_before = plan.r.v_rec.clone()
        out = plan.as_output()
        torch.cuda.synchronize()
        assert torch.equal(plan.r.v_rec, vrec_before)         # the densification's records survive the view
        culled = out.radii == 0
        assert int(culled.sum()) > 0 and int((first[5] > 0)[culled].sum()) > 0      # rows that WERE visible are among them
        for t in (out.means2d, out.depths, out.conics):
            assert float(t.detach()[culled].abs().max()) == 0.0
        res.append(first)
    a, b = res
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3] and torch.equal(a[5], b[5])
    for k in a[4]:
        assert float((a[4][k] - b[4][k]).abs().max()) < 2e-4 * float(a[4][k].abs().max()) + 1e-10, k

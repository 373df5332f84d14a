"""Tracking on the GPU: the graphed closure must be replayable (regression test for the memset-node corruption), and
the device-resident optimiser (csrc/track_opt.h through the C ABI) must do what torch.optim.Adam + torch.optim.LBFGS
do on the host (gslam/frontend.py:604-662).  Run with -m gpu."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _setup(dev, n=20000, W=320, H=240):
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.primitives import Camera, Frame, PoseZhou
    from gslam_amd.synthetic import make_intrinsics, make_scene, make_viewmat
    K = make_intrinsics(W, H).to(dev)
    cam = Camera(K, H, W)
    sc = make_scene(n, 0)
    sc["scales"] = sc["scales"] + 0.4
    m = GaussianSplattingData.from_dict(sc, dev).no_grad_clone()

    def frame(i, start):
        V = make_viewmat(i).to(dev)
        with torch.no_grad():
            img = m([cam], [PoseZhou(V, is_learnable=False).to(dev)], render_depth=False).rgbs[0].clamp(0, 1)
        V0 = make_viewmat(start).to(dev)
        return Frame(img=img.contiguous(), timestamp=0.0, camera=cam, pose=PoseZhou(V0).to(dev), gt_pose=V, index=i,
                     exposure_params=torch.zeros(2, device=dev))
    return m, cam, frame


def _pose_err(frame):
    with torch.no_grad():
        return float((frame.pose()[:3, 3] - frame.gt_pose[:3, 3]).norm())


def test_graph_replays_are_idempotent(dev):
    """Second and later replays of the captured closure give the first replay's numbers (they did not while the
    difference grid was cleared by a hipMemsetAsync graph node)."""
    from gslam_amd.tracking import GraphedTracker
    m, cam, frame = _setup(dev)
    tr = GraphedTracker(m, cam, device_optimizer=False)
    f = frame(1, 0)
    tr.load(f)
    tr.capture()
    tr.load(f)
    vals = []
    for _ in range(4):
        l = tr.closure()
        torch.cuda.synchronize()
        vals.append((float(l), [g.clone() for g in (p.grad for p in tr.params)]))
    assert tr.plan.r.check_capacity() and int(tr.plan.r.status.item()) == 0
    assert tr.graph is not None and tr.graph.nodes >= 8          # the closure is one graph of kernel nodes
    for v, gs in vals[1:]:
        assert abs(v - vals[0][0]) <= 1e-6 * max(1.0, abs(vals[0][0]))
        for a, b in zip(gs, vals[0][1]):
            assert torch.allclose(a, b, rtol=1e-3, atol=1e-7)


def test_device_optimizer_matches_host_state_machine(dev):
    """gsx_track_opt_* (device) against the same C code compiled for the host, on an analytic objective."""
    from gslam_amd._lib import check, lib
    so = os.path.join(HERE, "_build", "libtrackopt_host.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-ffp-contract=off", "-o", so,
                           os.path.join(HERE, "trackopt_host.c"), "-lm"])
    host = C.CDLL(so)
    host.trackopt_state_bytes.restype = C.c_long
    host.trackopt_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_double, C.c_int, C.c_int, C.c_int,
                                   C.c_double, C.c_double]
    host.trackopt_advance.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]

    def fn(x):
        return (100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1.0 - x[:-1]) ** 2).sum()

    n, n_adam, lr = 11, 3, 0.5
    x0 = (torch.arange(n, dtype=torch.float32) / n - 0.5)
    st_h = C.create_string_buffer(host.trackopt_state_bytes())
    host.trackopt_init(st_h, n, n_adam, lr, lr, 5, 20, 25, 1e-7, 1e-9)
    st_d = torch.zeros(int(lib.gsx_track_opt_state_bytes()), dtype=torch.uint8, device=dev)
    check(lib.gsx_track_opt_init(st_d.data_ptr(), n, n_adam, lr, lr, 5, 20, 25, 1e-7, 1e-9, None), "init")
    ph = x0.clone().numpy()
    pd = [x0[0:3].clone().to(dev), x0[3:9].clone().to(dev), x0[9:11].clone().to(dev)]   # three tensors like the tracker
    rep = torch.zeros(8, device=dev)
    for k in range(40):
        xh = torch.from_numpy(ph.copy()).requires_grad_(True)
        lh = fn(xh)
        lh.backward()
        xd = torch.cat(pd).detach().cpu()
        assert torch.allclose(xd, torch.from_numpy(ph), rtol=1e-5, atol=1e-6), (k, xd, ph)
        gh = xh.grad.numpy().astype(np.float32)
        host.trackopt_advance(st_h, ph.ctypes.data, gh.ctypes.data, float(lh))
        gd = torch.from_numpy(gh).to(dev)
        gds = [gd[0:3].contiguous(), gd[3:9].contiguous(), gd[9:11].contiguous()]
        loss_d = torch.tensor([float(lh)], device=dev)
        check(lib.gsx_track_opt_advance(st_d.data_ptr(), 3, (C.c_void_p * 3)(*[t.data_ptr() for t in pd]),
                                        (C.c_void_p * 3)(*[t.data_ptr() for t in gds]), (C.c_int * 3)(3, 6, 2),
                                        loss_d.data_ptr(), None), "advance")
        torch.cuda.synchronize()
    check(lib.gsx_track_opt_report(st_d.data_ptr(), rep.data_ptr(), None), "report")
    r = rep.cpu()
    assert int(r[0]) == 4 and int(r[1]) == host.trackopt_evals(st_h) and int(r[2]) == host.trackopt_iters(st_h)
    assert int(r[3]) == host.trackopt_stop_reason(st_h)


def test_device_optimizer_tracks_like_host_optimizer(dev):
    from gslam_amd.tracking import GraphedTracker
    m, cam, frame = _setup(dev)
    res = {}
    for mode in (False, True):
        tr = GraphedTracker(m, cam, device_optimizer=mode)
        errs, losses, evals = [], [], []
        for i in (1, 2, 3):
            f = frame(i, i - 1)                       # start from the previous frame's pose, as the frontend does
            e0 = _pose_err(f)
            loss, n = tr.track(f)
            torch.cuda.synchronize()
            errs.append((e0, _pose_err(f)))
            losses.append(loss)
            evals.append(n)
        res[mode] = (errs, losses, evals)
    for mode in (False, True):
        errs, losses, evals = res[mode]
        for (e0, e1), n in zip(errs, evals):
            assert e1 < 0.5 * e0, (mode, e0, e1)      # both optimisers pull the pose towards the ground truth
            assert 11 <= n <= 37
    for lh, ld in zip(res[False][1], res[True][1]):
        # near the optimum the line-search branches hang on float32 noise (tests/test_track_opt_cpu.py): same order only
        assert max(lh, ld) <= 10.0 * min(lh, ld) + 1e-4, (lh, ld)


def test_optimize_poses_lbfgs(dev):
    """gslam/backend.py:447-506: window pose refinement; frame 0 is not touched, the others move towards the truth."""
    from gslam_amd.mapping import optimize_poses_lbfgs
    m, cam, frame = _setup(dev)
    window = [frame(0, 0), frame(2, 1), frame(4, 3)]          # (true pose index, starting pose index)
    window[0].index = 0
    before = [_pose_err(f) for f in window]
    p0 = window[0].pose().detach().clone()
    last = optimize_poses_lbfgs(m, window, max_eval=20)
    torch.cuda.synchronize()
    after = [_pose_err(f) for f in window]
    assert last is not None and np.isfinite(last)
    assert torch.equal(window[0].pose().detach(), p0)
    assert after[1] < 0.6 * before[1] and after[2] < 0.6 * before[2], (before, after)
    assert all(not p.requires_grad for p in m.parameters())   # frozen-map state restored as it was


def test_device_pose_refiner_matches_host_lbfgs(dev):
    """SURVEY 8f rank 2, backend side: the window pose L-BFGS of gslam/backend.py:447-506 as a device state machine
    (mapping.GraphedPoseRefiner, csrc/window_opt.hip) against torch.optim.LBFGS on the host over the same closure."""
    from gslam_amd.mapping import GraphedPoseRefiner, optimize_poses_lbfgs
    m, cam, frame = _setup(dev)

    def make():
        w = [frame(0, 0), frame(2, 1), frame(4, 3)]
        w[0].index = 0
        return w
    wh, wd = make(), make()
    before = [_pose_err(f) for f in wd]
    last_h = optimize_poses_lbfgs(m, wh, max_eval=25)
    p0 = wd[0].pose().detach().clone()
    ref = GraphedPoseRefiner(m, wd, max_eval=25)
    assert ref.n == 18 and len(ref.params) == 4
    last_d, n_evals = ref.run()
    torch.cuda.synchronize()
    assert torch.equal(wd[0].pose().detach(), p0)                       # frame 0 stays fixed (backend.py:459-462)
    assert 2 <= n_evals <= 26 and np.isfinite(last_d)
    after_d, after_h = [_pose_err(f) for f in wd], [_pose_err(f) for f in wh]
    assert after_d[1] < 0.6 * before[1] and after_d[2] < 0.6 * before[2], (before, after_d)
    # same optimiser, same closure: the two runs end at the same poses up to float noise in the line search
    # (float atomics in the rasteriser backward + a strong-Wolfe line search: compared with slack)
    assert abs(last_d - last_h) < 5e-2 * abs(last_h), (last_d, last_h)
    for fd, fh in zip(wd[1:], wh[1:]):
        assert (fd.pose().detach() - fh.pose().detach()).abs().max() < 2e-2
    # a second refinement of the same window replays the same graph
    g = ref.graph
    last2, _ = ref.run()
    assert ref.graph is g and last2 <= last_d * 1.001
    assert all(not p.requires_grad for p in m.parameters())


def test_fused_closure_tail_equals_split_launches(dev):
    """the tracker's one-launch closure tail (pose partials -> PoseZhou backward -> optimiser step -> PoseZhou forward,
    tracking loss finished inside) against the separate finish / pose_bwd / advance / pose_fwd launches.  The float
    atomics of the rasteriser backward make two runs of ONE variant differ in the last bits, and the strong-Wolfe line
    search amplifies that near the optimum, so each variant is held to the ground truth and the two to each other
    only loosely; the first closure of a frame (before any optimiser decision) is compared tightly."""
    from gslam_amd.tracking import GraphedTracker
    m, cam, frame = _setup(dev)
    res = {}
    for mode in ("split", "fused"):
        tr = GraphedTracker(m, cam, device_optimizer=True, tail=mode)
        assert tr.fused_tail == (mode == "fused")
        out = []
        for i in (1, 2):
            f = frame(i, i - 1)
            e0 = _pose_err(f)
            loss, n = tr.track(f)
            torch.cuda.synchronize()
            out.append((f.pose().detach().clone(), e0, _pose_err(f), loss, n))
        # one closure from a fixed start: identical inputs to the tail, so the first Adam step must agree closely
        f = frame(3, 2)
        tr.load(f)
        from gslam_amd._lib import check, lib
        check(lib.gsx_track_opt_init(tr._state.data_ptr(), 11, tr.conf.n_adam_warmup, tr.conf.pose_optim_lr,
                                     tr.conf.pose_optim_lr, tr.conf.lbfgs_history, 20, 25, 1e-7, 1e-9, None), "init")
        tr.graph.replay()
        torch.cuda.synchronize()
        res[mode] = (out, torch.cat([tr.pose.dt.detach(), tr.pose.dR.detach(), tr.exposure.detach()]).clone())
    for (pa, e0a, e1a, la, na), (pb, e0b, e1b, lb, nb) in zip(res["split"][0], res["fused"][0]):
        assert e1a < 0.5 * e0a and e1b < 0.5 * e0b, (e0a, e1a, e0b, e1b)
        assert 11 <= na <= 37 and 11 <= nb <= 37
        assert (pa - pb).abs().max() < 5e-2
    step_a, step_b = res["split"][1], res["fused"][1]
    assert step_a.abs().max() > 0 and (step_a - step_b).abs().max() < 1e-5 * max(1.0, float(step_a.abs().max())) + 2e-6


def _torch_warp(K, f1, f2, c1, d1):
    """depth-based inverse warp of gslam/warp.py:35-82 written with plain torch ops in [H, W] layout (pinned against the
    reference-generated fixture below before it is used as the yardstick of warp_track)"""
    import torch.nn.functional as F
    H, W = d1.shape
    T = f1 @ torch.linalg.inv(f2)
    v, u = torch.meshgrid(torch.arange(H, device=d1.device), torch.arange(W, device=d1.device), indexing="ij")
    pix = torch.stack([u, v, torch.ones_like(u)], dim=-1).to(d1.dtype)
    pts = d1[..., None] * (pix @ torch.linalg.inv(K).t()) + 1e-10
    uvw = (pts @ T[:3, :3].t() + T[:3, 3]) @ K.t()
    uv = uvw[..., :2] / uvw[..., 2:3]
    nw = torch.stack([uv[..., 0] * (2.0 / W) - 1.0, uv[..., 1] * (2.0 / H) - 1.0], dim=-1)[None]
    res = F.grid_sample(c1.permute(2, 0, 1)[None], nw, padding_mode="zeros", align_corners=False)[0].permute(1, 2, 0)
    keep = (nw[0, ..., 0] < 1.0) & (nw[0, ..., 1] < 1.0) & (nw[0, ..., 0] > -1.0) & (nw[0, ..., 1] > -1.0)
    return res, nw, keep


def test_warp_track_matches_torch_formulated_loop(dev):
    """gslam/frontend.py:521-569 (`method='warp'`): Nesterov SGD on the pose through the depth-based warp of the reference
    keyframe, masked L1 with the exposure affine.  tracking.warp_track (HIP Warp forward / backward) against the same loop
    over the torch formulation of the warp, on the reference-generated warp fixture."""
    from gslam_amd.primitives import Camera, Frame, PoseZhou
    from gslam_amd.tracking import TrackingConfig, warp_track
    g = dict(np.load(os.path.join(HERE, "golden", "warp_120x160.npz")))
    H, W = g["d1"].shape
    K = torch.from_numpy(g["K"]).to(dev)
    c1, d1 = torch.from_numpy(g["c1"]).to(dev), torch.from_numpy(g["d1"]).to(dev)
    f1, f2 = torch.from_numpy(g["f1_pose"]).to(dev), torch.from_numpy(g["f2_pose"]).to(dev)
    # the yardstick itself against what the imported reference produced (oracle/gen_golden.py)
    res, nw, keep = _torch_warp(K, f1, f2, c1, d1)
    np.testing.assert_allclose(nw.cpu().numpy(), g["normalized_warps"], atol=3e-6)
    assert np.abs(res.cpu().numpy() - g["result"]).max() < 2e-4
    # tracking needs image structure (the fixture's image is white noise): a smooth pattern as the reference keyframe; the
    # new frame = that image seen from f2, brightened - the tracker has to find f2 and the exposure
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H, device=dev), torch.linspace(0, 1, W, device=dev), indexing="ij")
    c1 = torch.stack([0.5 + 0.4 * torch.sin(7 * xx + 3 * yy), 0.5 + 0.4 * torch.cos(5 * yy - 2 * xx),
                      0.5 + 0.4 * torch.sin(4 * xx * yy + 1.0)], dim=-1).contiguous()
    with torch.no_grad():
        res = _torch_warp(K, f1, f2, c1, d1)[0]
        target = (res * 1.05 + 0.02).contiguous()
    cam = Camera(K, H, W)
    ref = Frame(img=c1, timestamp=0.0, camera=cam, pose=PoseZhou(f1, is_learnable=False).to(dev), gt_pose=f1, index=0,
                exposure_params=torch.zeros(2, device=dev))
    start = f2.clone()
    start[:3, 3] += torch.tensor([0.01, -0.008, 0.005], device=dev)
    conf = TrackingConfig(num_tracking_iters=40)

    def new_frame():
        return Frame(img=target, timestamp=0.1, camera=cam, pose=PoseZhou(start.clone()).to(dev), gt_pose=f2, index=1,
                     exposure_params=torch.zeros(2, device=dev, requires_grad=True))
    fa = new_frame()
    loss_a = warp_track(fa, ref, c1, d1, conf)
    # the same loop, torch-formulated warp
    fb = new_frame()
    opt = torch.optim.SGD(list(fb.pose.parameters()), lr=conf.pose_optim_lr, momentum=0.8, nesterov=True)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=conf.pose_optim_lr_decay)
    opt.add_param_group({"params": fb.exposure_params, "lr": 0.01})
    first = None
    for _ in range(conf.num_tracking_iters):
        opt.zero_grad()
        r, _, k = _torch_warp(K, f1, fb.pose(), c1, d1)
        r = r * fb.exposure_params[0].exp() + fb.exposure_params[1]
        loss_b = torch.nn.functional.l1_loss(r[k], target[k])
        first = float(loss_b) if first is None else first
        loss_b.backward()
        opt.step()
        sched.step()
    torch.cuda.synchronize()
    assert float(loss_b) < 0.8 * first                                   # the loop does track
    # an L1 loss has sign() gradients: residuals near zero flip between the two warps' last bits and 40 momentum steps
    # amplify that, so the trajectories are compared loosely and ONE iteration tightly (below)
    assert abs(float(loss_a) - float(loss_b)) < 0.1 * float(loss_b) + 1e-6, (float(loss_a), float(loss_b))
    assert (fa.pose().detach() - fb.pose().detach()).abs().max() < 2e-3
    assert (fa.exposure_params - fb.exposure_params).abs().max() < 5e-3
    # one iteration from the same start: loss value and the gradients that reach the pose delta and the exposure pair
    from gslam_amd.warp import Warp
    grads = []
    for use_hip in (True, False):
        f = new_frame()
        if use_hip:
            r, _, k = Warp(K, H, W).to(dev)(f1, f.pose(), c1, d1)
        else:
            r, _, k = _torch_warp(K, f1, f.pose(), c1, d1)
        r = r * f.exposure_params[0].exp() + f.exposure_params[1]
        loss = ((r - target).abs() * k[..., None]).sum() / k.sum().clamp(min=1) / 3.0
        loss.backward()
        grads.append((float(loss), f.pose.dR.grad.clone(), f.pose.dt.grad.clone(), f.exposure_params.grad.clone()))
    assert abs(grads[0][0] - grads[1][0]) < 1e-5 * grads[1][0]
    for ga, gb in zip(grads[0][1:], grads[1][1:]):
        assert (ga - gb).abs().max() < 2e-3 * gb.abs().max() + 1e-8, (ga, gb)


def test_host_igs_track_lbfgs_against_device_state_machine(dev):
    """tracking.igs_track_lbfgs (torch.optim.Adam + torch.optim.LBFGS on the host over the autograd operators, exactly the
    reference's frontend.py:604-662) and the device state machine on the launch plan, same frames: both pull the pose to
    the truth, with the same number of closures up to the line search's float noise"""
    from gslam_amd.tracking import GraphedTracker, igs_track_lbfgs
    m, cam, frame = _setup(dev)
    tr = GraphedTracker(m, cam, device_optimizer=True)
    for i in (1, 2):
        fh, fd = frame(i, i - 1), frame(i, i - 1)
        e0 = _pose_err(fh)
        loss_h, n_h = igs_track_lbfgs(m, fh)
        loss_d, n_d = tr.track(fd)
        torch.cuda.synchronize()
        eh, ed = _pose_err(fh), _pose_err(fd)
        assert eh < 0.5 * e0 and ed < 0.5 * e0, (e0, eh, ed)
        assert 11 <= n_h <= 37 and 11 <= n_d <= 37
        assert max(loss_h, loss_d) <= 10.0 * min(loss_h, loss_d) + 1e-4, (loss_h, loss_d)
        assert (fh.pose().detach() - fd.pose().detach()).abs().max() < 5e-2


def test_warp_640x480_matches_the_fixture_pinned_torch_formulation(dev):
    """SURVEY 8(c) G1 asked for Warp at 480x640; the reference-generated fixtures are 48x64 and 120x160.  Here the HIP Warp
    (forward, keep mask, both pose gradients incl. the bottom row of f2) runs at the headline resolution against the torch
    formulation of gslam/warp.py:35-82 - which is first held to the reference-generated fixture itself."""
    from gslam_amd.warp import Warp
    g = dict(np.load(os.path.join(HERE, "golden", "warp_120x160.npz")))
    Ks, cs, ds = torch.from_numpy(g["K"]).to(dev), torch.from_numpy(g["c1"]).to(dev), torch.from_numpy(g["d1"]).to(dev)
    f1s, f2s = torch.from_numpy(g["f1_pose"]).to(dev), torch.from_numpy(g["f2_pose"]).to(dev)
    res, nw, keep = _torch_warp(Ks, f1s, f2s, cs, ds)
    np.testing.assert_allclose(nw.cpu().numpy(), g["normalized_warps"], atol=3e-6)       # the yardstick is the reference's
    assert np.abs(res.cpu().numpy() - g["result"]).max() < 2e-4
    H, W = 480, 640
    gen = torch.Generator().manual_seed(5)
    K = torch.tensor([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1]], device=dev)
    c1 = torch.rand(H, W, 3, generator=gen).to(dev)
    d1 = (1.0 + torch.rand(H, W, generator=gen)).to(dev)
    f1, f2 = f1s.clone().requires_grad_(True), f2s.clone().requires_grad_(True)
    warp = Warp(K, H, W)
    r_h, nw_h, keep_h = warp(f1, f2, c1, d1)
    (r_h[keep_h].sum() + 0.1 * nw_h.square().sum()).backward()
    # the yardstick at this size in FLOAT64 (300 k pixels summed into 12 numbers: the float32 autograd of the same formulation
    # is itself ~0.3 % off in the largest entries - printed below), result and mask in float32 as the kernel computes them
    t1, t2 = f1s.double().requires_grad_(True), f2s.double().requires_grad_(True)
    r_t, nw_t, keep_t = _torch_warp(K.double(), t1, t2, c1.double(), d1.double())
    (r_t[keep_t].sum() + 0.1 * nw_t.square().sum()).backward()
    u1, u2 = f1s.clone().requires_grad_(True), f2s.clone().requires_grad_(True)
    r_u, nw_u, keep_u = _torch_warp(K, u1, u2, c1, d1)
    (r_u[keep_u].sum() + 0.1 * nw_u.square().sum()).backward()
    np.testing.assert_allclose(nw_h.detach().cpu().numpy(), nw_t.detach().float().cpu().numpy(), atol=3e-6)
    border = ((nw_t[0].detach().abs() - 1.0).abs() < 1e-5).any(-1)
    assert bool((keep_h == keep_t)[~border].all())
    d = (r_h.double() - r_t).detach().abs()
    # (float32 pixel coordinates up to 640 carry an ulp of 6e-5 px into the bilinear weights: 7e-6 mean against float64)
    assert float(d.max()) < 4e-4 and float(d.mean()) < 2e-5
    scale = max(float(t1.grad.abs().max()), 1.0)
    e_hip = max(float((f1.grad.double() - t1.grad).abs().max()), float((f2.grad.double() - t2.grad).abs().max())) / scale
    e_f32 = max(float((u1.grad.double() - t1.grad).abs().max()), float((u2.grad.double() - t2.grad).abs().max())) / scale
    print(f"warp 640x480 pose gradients against float64: HIP {e_hip:.2e}, torch float32 autograd {e_f32:.2e} (of the largest entry)")
    # measured: HIP 6.8e-3, torch float32 autograd 1.5e-2 (pixels whose keep bit differs between float32 and float64 enter or
    # leave the sum): the bar is the float32 autograd of the same formulation, not float64 round-off
    assert e_hip < 2e-2 and e_hip <= 1.5 * e_f32 + 1e-3, (e_hip, e_f32)
    assert float(t2.grad[3].abs().max()) > 1.0

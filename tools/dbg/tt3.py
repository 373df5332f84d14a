"""One variant per process: which eager action between two replays of the tracking graph corrupts the next replay?"""
import os, sys, subprocess
VARIANTS = ["A_replay_only", "C_adam"]
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.primitives import Camera, Frame, PoseZhou
    from gslam_amd import rasterization as R
    from gslam_amd.synthetic import make_intrinsics, make_scene, make_viewmat
    from gslam_amd.tracking import GraphedTracker, TrackingConfig
    v = sys.argv[1]
    dev = torch.device("cuda:0")
    W, H, N = 640, 480, 20000
    K = make_intrinsics(W, H).to(dev)
    cam = Camera(K, H, W)
    m = GaussianSplattingData.from_dict(make_scene(N, 0), dev).no_grad_clone()
    V = make_viewmat(0).to(dev)
    f = Frame(img=torch.rand(H, W, 3, device=dev), timestamp=0.0, camera=cam, pose=PoseZhou(V).to(dev), gt_pose=V,
              index=0, exposure_params=torch.zeros(2, device=dev))
    tr = GraphedTracker(m, cam, TrackingConfig())
    tr.load(f)
    if v != "F_adam_nograph":
        tr.capture()
        tr.load(f)
    p = R._pool(dev)

    def run(tag):
        l = tr.closure()
        torch.cuda.synchronize()
        st = int(p.status.item())
        M = int(p._graph_M.item()) if p._graph_M is not None else p.last_M
        g = [float(x.grad.abs().sum()) if x.grad is not None else None for x in tr.params]
        print(f"{v} {tag}: loss {float(l):.5f} M {M} status {st} grads {g}", flush=True)
        p.status.zero_()
    run("r1")
    run("r2")
    if v == "B_inplace_param":
        with torch.no_grad():
            tr.pose.dt.add_(0.002)
            tr.pose.dR.add_(0.002)
    elif v in ("C_adam", "F_adam_nograph"):
        opt = torch.optim.Adam(tr.params, 0.002)
        opt.step()
    elif v == "D_allocs":
        keep = [torch.zeros(11, device=dev) for _ in range(20)] + [torch.zeros(1 << 20, device=dev) for _ in range(4)]
    elif v == "E_sgd_manual":
        with torch.no_grad():
            for q in tr.params:
                q.add_(q.grad, alpha=-1e-4)
    torch.cuda.synchronize()
    run("r3")
    run("r4")
    print("VARIANT_DONE", v)
else:
    for v in VARIANTS:
        r = subprocess.run([sys.executable, __file__, v], capture_output=True, text=True, timeout=150)
        out = [l for l in (r.stdout + r.stderr).splitlines() if l.startswith(v) or "fault" in l.lower() or "Error" in l]
        print("\n".join(out[-8:]), flush=True)

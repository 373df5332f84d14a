import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gslam_amd.map import GaussianSplattingData
from gslam_amd.mapping import GraphedPoseRefiner, optimize_poses_lbfgs
from gslam_amd.primitives import Camera, Frame, PoseZhou
from gslam_amd.synthetic import make_intrinsics, make_scene, make_viewmat
dev = torch.device("cuda:0")
for N in (5000, 100000, 500000):
    W, H = 640, 480
    K = make_intrinsics(W, H).to(dev); cam = Camera(K, H, W)
    m = GaussianSplattingData.from_dict(make_scene(N, 0), dev)
    gt = GaussianSplattingData.from_dict(make_scene(N, 1), dev)
    def frames():
        out = []
        for i in range(8):
            V = make_viewmat(i).to(dev)
            with torch.no_grad():
                img = gt([cam], [PoseZhou(V, is_learnable=False).to(dev)]).rgbs[0].clamp(0, 1).contiguous()
            out.append(Frame(img=img, timestamp=0.0, camera=cam, pose=PoseZhou(make_viewmat(i + 1).to(dev)).to(dev), gt_pose=V, index=i,
                             exposure_params=torch.zeros(2, device=dev)))
        return out
    def t(fn):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3, r
    w = frames(); optimize_poses_lbfgs(m, w)      # warm
    w = frames(); th, _ = t(lambda: optimize_poses_lbfgs(m, w))
    w = frames(); ref = None
    def first():
        global ref
        ref = GraphedPoseRefiner(m, w); return ref.run()
    t1, r1 = t(first)
    t2, r2 = t(lambda: ref.run())
    print(f"N={N}: host L-BFGS {th:.1f} ms; device first call (capture + run) {t1:.1f} ms {r1}; second call {t2:.1f} ms {r2}")

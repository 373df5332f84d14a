"""Per-tile list statistics of the synthetic scenes (how deep the compositing goes, load balance)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gslam_amd.map import GaussianSplattingData
from gslam_amd.primitives import Camera, PoseZhou
from gslam_amd.synthetic import make_intrinsics, make_scene, make_viewmat
dev = torch.device("cuda:0")
W, H = 640, 480
K = make_intrinsics(W, H).to(dev)
cam = Camera(K, H, W)
for N in (100_000, 500_000):
    m = GaussianSplattingData.from_dict(make_scene(N, 0), dev)
    with torch.no_grad():
        out = m([cam], [PoseZhou(make_viewmat(0).to(dev), is_learnable=False).to(dev)], render_depth=True)
    off = out.isect_offsets.flatten().long()
    M = out.flatten_ids.shape[0]
    cnt = torch.diff(torch.cat([off, torch.tensor([M], device=dev)])).float()
    vis = int((out.radii > 0).sum())
    # depth reached per tile: max last index - start
    li = None
    q = torch.tensor([0.5, 0.9, 0.99], device=dev)
    print(f"N={N} visible={vis} M={M} tiles={cnt.numel()} mean={cnt.mean():.0f} p50/p90/p99={torch.quantile(cnt, q).tolist()} max={cnt.max():.0f}")
    a = out.alphas[0, ..., 0]
    print(f"   alpha mean {a.mean():.3f}  frac saturated(>0.999) {(a > 0.999).float().mean():.3f}")
    r = out.radii[out.radii > 0].float()
    print(f"   radii mean {r.mean():.1f} p50/p90/p99 {torch.quantile(r, q).tolist()} max {r.max():.0f}")

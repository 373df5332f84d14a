import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from gslam_amd.map import GaussianSplattingData
from gslam_amd.synthetic import make_scene, sequence_param
from gslam_amd.tracking import GraphedTracker, TrackingConfig
import gslam_amd.plan as P
dev = torch.device('cuda:0')
N, W, H = 500000, 640, 480
gt = GaussianSplattingData.from_dict(make_scene(N, 1), dev)
m = GaussianSplattingData.from_dict(make_scene(N, 0), dev)
frames, cam = bench.make_frames([sequence_param(i) for i in range(4)], W, H, dev, gt)
fm = m.no_grad_clone()
tr = GraphedTracker(fm, cam, TrackingConfig())
tr.track(frames[0]); torch.cuda.synchronize()
# time the pieces of a rebuild
def T(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print(f"{label:40s} {(time.perf_counter() - t0) * 1e3:8.2f} ms"); return r
fm2 = GaussianSplattingData.from_dict({k: v[:N - 1000].clone() for k, v in make_scene(N, 0).items()}, dev).no_grad_clone()
c = T("TrackClosure()", lambda: P.TrackClosure(fm2, cam))
T("load", lambda: c.load(frames[1].pose().detach(), frames[1].img, frames[1].exposure_params))
T("prepare (probe + capture)", lambda: c.prepare())
T("probe alone", lambda: c.r.probe())
T("build_candidates", lambda: c.r.build_candidates(P.current_stream_ptr(dev)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
c2 = P.TrackClosure(fm2, cam); c2.load(frames[1].pose().detach(), frames[1].img, frames[1].exposure_params); c2.prepare(); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)

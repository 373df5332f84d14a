import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gslam_amd.map import GaussianSplattingData
from gslam_amd.primitives import Camera, PoseZhou
from gslam_amd.synthetic import make_intrinsics, make_scene, make_viewmat
dev = torch.device("cuda:0")
W, H = 640, 480
K = make_intrinsics(W, H).to(dev)
cam = Camera(K, H, W)
for N in (100_000,):
    m = GaussianSplattingData.from_dict(make_scene(N, 0), dev)
    for it in range(3):
        with torch.no_grad():
            out = m([cam], [PoseZhou(make_viewmat(0).to(dev), is_learnable=False).to(dev)], render_depth=True)
        torch.cuda.synchronize()
    a = out.alphas[0, :, :, 0]
    # wave 0 of each tile wrote at (ty*16, tx*16..+2); wave 1 at (ty*16+8, ...)
    rows = []
    for wv in (0, 1):
        sub = a[wv * 8::16, :]
        tb, te, n = sub[:, 0::16], sub[:, 1::16], sub[:, 2::16]
        rows.append((tb, te, n))
    tb = torch.stack([rows[0][0], rows[1][0]]).flatten().double()
    te = torch.stack([rows[0][1], rows[1][1]]).flatten().double()
    n = torch.stack([rows[0][2], rows[1][2]]).flatten()
    t0 = tb.min()
    dur = (te - tb) * 0.01  # us (100 MHz)
    print(f"N={N}: waves {tb.numel()} start spread {float((tb.max()-t0)*0.01):.1f}us  end max {float((te.max()-t0)*0.01):.1f}us")
    q = torch.tensor([0.1, 0.5, 0.9, 0.99, 1.0], dtype=torch.float64, device=dev)
    print("  wave duration us quantiles 10/50/90/99/100:", [round(x, 1) for x in torch.quantile(dur, q).tolist()])
    print("  start time us quantiles:", [round(x, 1) for x in torch.quantile((tb - t0) * 0.01, q).tolist()])
    print("  entries/tile mean", float(n.mean()), "corr(dur, n)", float(torch.corrcoef(torch.stack([dur.float(), n.float()]))[0, 1]))

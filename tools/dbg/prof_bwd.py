import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gslam_amd import ops
from gslam_amd._lib import check, lib, ptr, stream_ptr
from gslam_amd.rasterization import rasterization, validate
from gslam_amd.synthetic import make_cameras, make_scene
dev = torch.device("cuda:0")
N, C, W, H = 100000, 1, 640, 480
tw, th = 40, 30
sc = {k: v.to(dev) for k, v in make_scene(N, 0).items()}
viewmats, Ks = make_cameras(C, W, H); viewmats, Ks = viewmats.to(dev), Ks.to(dev)
with torch.no_grad():
    out = rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["colors"], viewmats, Ks, W, H, packed=False, render_mode="RGB+D", log_uncertainties=sc["log_uncertainties"], backgrounds=torch.zeros(C, 3, device=dev))
    validate(dev)
M = out.flatten_ids.shape[0]
radii, m2d, dep, con, _, rec, tiles, _ = ops._Projection.apply(sc["means"], sc["quats"], sc["scales"], viewmats, Ks, sc["opacities"], sc["colors"], sc["log_uncertainties"], W, H, 0.3, 0.01, 1e10, 0.0, False, 7, True, True)
off, flat = out.isect_offsets.contiguous(), out.flatten_ids.contiguous()
bg = torch.zeros(C, 5, device=dev); bg[:, 4] = 2.718281828
render_t = torch.empty(C, H, W, 5, device=dev); alphas = torch.empty(C, H, W, 1, device=dev)
last = torch.empty(C, H, W, dtype=torch.int32, device=dev); nt = torch.zeros(C, N, dtype=torch.int32, device=dev)
v_render = torch.randn(C, H, W, 5, device=dev); v_alpha = torch.randn(C, H, W, 1, device=dev)
v_rec = torch.zeros(C, N, 12, device=dev)
st = stream_ptr(dev)
check(lib.gsx_raster_fwd(ptr(rec), 5, ptr(bg), ptr(off), ptr(flat), M, 0, C, W, H, tw, th, 0.5, ptr(render_t), ptr(alphas), ptr(last), ptr(nt), None, st), "fwd")
for _ in range(3):
    v_rec.zero_()
    check(lib.gsx_raster_bwd(ptr(rec), 5, ptr(bg), ptr(off), ptr(flat), M, 0, C, W, H, tw, th, ptr(alphas), ptr(last), ptr(v_render), ptr(v_alpha), ptr(v_rec), None, None, 0, st), "bwd")
    torch.cuda.synchronize()
d = v_rec[0, :6 * 2400, 11].reshape(2400, 6).double()
names = ["alpha part (all survivors)", "gradient math", "DPP reduce", "LDS add", "n contributing", "cull+stage per chunk total"]
for i, nm in enumerate(names):
    print(f"{nm:32s} mean per wave: {float(d[:, i].mean()):10.0f}")
nc = d[:, 4].mean()
print("per contributing survivor cycles: alpha", float(d[:,0].sum()/d[:,4].sum()), "math", float(d[:,1].sum()/d[:,4].sum()), "dpp", float(d[:,2].sum()/d[:,4].sum()), "lds", float(d[:,3].sum()/d[:,4].sum()))

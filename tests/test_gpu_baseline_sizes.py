"""Parity against the CPU oracle AT the BASELINE.json sizes, not only through size-independent properties:
configs[1] (100 k Gaussians, 640x480, CH = 5: forward, tile lists, backward, fused SSIM) with the oracle run live
(about 2 s on one core), and the forward + tile lists of configs[2]'s map size (500 k, C = 1; marked slow).
Printed with -s: L1, max-abs and the three proxies for flipped cull decisions that SURVEY.md 9.3 asks for - pixels whose
last contributing entry differs (a flipped alpha < 1/255 or T' <= 1e-4 decision moves it), (camera, Gaussian) pairs whose
touched-pixel count differs, pixels off by more than 1e-5.  Run with -m gpu."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _np(t):
    return t.detach().cpu().numpy()


def _run_pair(dev, oracle32, n, W=640, H=480, with_backward=True):
    from gslam_amd.rasterization import rasterization
    from gslam_amd.synthetic import make_cameras, make_scene
    sc = make_scene(n, 0)
    viewmats, Ks = make_cameras(1, W, H)
    d = {k: v.to(dev) for k, v in sc.items()}
    leaves = {}
    if with_backward:
        for k in ("means", "quats", "scales", "opacities", "colors", "log_uncertainties"):
            leaves[k] = d[k].clone().requires_grad_(True)
        d.update(leaves)
    vm = viewmats.to(dev).requires_grad_(with_backward)
    out = rasterization(d["means"], d["quats"], d["scales"], d["opacities"], d["colors"], vm, Ks.to(dev), W, H,
                        packed=False, render_mode="RGB+D", log_uncertainties=d["log_uncertainties"],
                        backgrounds=torch.zeros(1, 3, device=dev))
    scales_gpu = torch.exp(d["scales"].detach()).cpu().numpy()       # the device's exp: integer outputs compare bit for bit
    o = oracle32.gslam_rasterization(_np(sc["means"]), _np(sc["quats"]), _np(sc["scales"]), _np(sc["opacities"]),
                                     _np(sc["colors"]), _np(viewmats), _np(Ks), W, H, render_mode="RGB+D",
                                     log_uncertainties=_np(sc["log_uncertainties"]),
                                     backgrounds=np.zeros((1, 3), np.float32), scales_override=scales_gpu)
    return sc, viewmats, Ks, out, o, leaves, vm


def _report(tag, out, o):
    render, o_render = _np(out._render), o["render"]
    diff = np.abs(render - o_render)
    # one flipped alpha >= 1/255 decision moves a pixel by at most (1/255) * T * |channel value|: colours <= 1, beta <= e,
    # the depth channel up to the scene depth (6 m here) - the max-abs is bounded per channel class below
    l1, mx = float(diff.mean()), float(diff[..., :3].max())
    mx_depth = float(diff[..., 3].max())
    print(f"[{tag}] max-abs rgb {mx:.3e}, depth channel {mx_depth:.3e} (bounds 1/255 and depth_max/255)")
    assert mx_depth < 6.5 / 255.0 and float(diff[..., 4].max()) < 2.72 / 255.0
    px_off = int((diff.max(axis=-1) > 1e-5).sum())
    nt = _np(out.n_touched)
    nt_diff = int((nt != o["n_touched"]).sum())
    print(f"[{tag}] M={o['flatten_ids'].shape[0]} render L1={l1:.3e} max-abs={mx:.3e} pixels off by >1e-5: {px_off} "
          f"of {diff.shape[1] * diff.shape[2]}; n_touched differs on {nt_diff} of {nt.size} (camera, Gaussian) pairs")
    return l1, mx, px_off, nt_diff


def test_config1_100k_forward_lists_backward_vs_live_oracle(dev, oracle32):
    """BASELINE.json configs[1]: 100 k Gaussians, 640x480, RGB + depth + beta"""
    n, W, H = 100_000, 640, 480
    sc, viewmats, Ks, out, o, leaves, vm = _run_pair(dev, oracle32, n)
    # integer outputs: bit-exact (radii, tiles per Gaussian, sorted keys, flatten ids, offsets) and the float rows K1 writes
    assert np.array_equal(_np(out.radii), o["radii"])
    assert np.array_equal(_np(out.tiles_per_gauss), o["tiles_per_gauss"])
    assert np.array_equal(_np(out.isect_ids), o["isect_ids"]) and np.array_equal(_np(out.flatten_ids), o["flatten_ids"])
    assert np.array_equal(_np(out.isect_offsets), o["isect_offsets"])
    assert np.array_equal(_np(out.means2d), o["means2d"]) and np.array_equal(_np(out.depths), o["depths"])
    assert np.array_equal(_np(out.conics), o["conics"])
    l1, mx, px_off, nt_diff = _report("configs[1] 100k", out, o)
    assert l1 < 1e-5 and mx < 1.0 / 255.0                   # bar: 1e-4 L1 per pixel; a flipped cut moves a pixel by < 1/255
    assert px_off < 0.002 * W * H                           # flipped cull decisions: a handful of pixels
    assert nt_diff < 0.002 * n
    assert float(np.abs(_np(out.alphas) - o["alphas"]).mean()) < 1e-6
    # last contributing entry per pixel: the same for (almost) every pixel - it moves only with a flipped decision
    # (the eager operator does not return last_ids; the launch plan's forward does - same kernels)
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.plan import RenderPlan, current_stream_ptr
    m = GaussianSplattingData.from_dict(sc, dev).no_grad_clone()
    r = RenderPlan(m, 1, W, H, render_depth=True, grads='none')
    r.Ks.copy_(Ks.to(dev)); r.viewmats.copy_(viewmats.to(dev))
    r.probe()
    r.forward(current_stream_ptr(dev))
    torch.cuda.synchronize()
    assert r.check_capacity() and r.front
    assert torch.equal(r.render, out._render.detach())       # plan (fused front) == eager operators, bit for bit
    last_diff = int((_np(r.last_ids) != o["last_ids"]).sum())
    print(f"[configs[1] 100k] pixels whose last contributing entry differs: {last_diff} of {W * H}")
    assert last_diff < 0.002 * W * H
    # backward of a fixed linear functional of the render: every map gradient and the view-matrix gradient
    g = torch.Generator().manual_seed(3)
    w_r = torch.randn(out._render.shape, generator=g) * 1e-3
    w_a = torch.randn(out.alphas.shape, generator=g) * 1e-3
    ((out._render * w_r.to(dev)).sum() + (out.alphas * w_a.to(dev)).sum()).backward()
    vm2d, vcon, vcol, vop, _ = oracle32.raster_bwd(o["means2d"], o["conics"], o["colors_packed"], o["opacities"],
                                                   o["backgrounds_packed"], W, H, 16, o["isect_offsets"], o["flatten_ids"],
                                                   o["alphas"], o["last_ids"], _np(w_r), _np(w_a))
    scales = np.exp(_np(sc["scales"]))
    ref = oracle32.project_bwd(_np(sc["means"]), _np(sc["quats"]), scales, _np(viewmats), _np(Ks), W, H, o["radii"], vm2d,
                               vcol[..., 3], vcon)
    checks = (("means", leaves["means"].grad, ref[0]), ("quats", leaves["quats"].grad, ref[1]),
              ("log_scales", leaves["scales"].grad, ref[2] * scales), ("viewmats", vm.grad, ref[3]))
    for name, got, want in checks:
        got = _np(got)
        scale = np.abs(want).max() + 1e-12
        rel_max = float(np.abs(got - want).max() / scale)
        rel_mean = float(np.abs(got - want).mean() / (np.abs(want).mean() + 1e-12))
        print(f"[configs[1] 100k] grad {name}: max rel {rel_max:.2e} mean rel {rel_mean:.2e}")
        assert rel_max < 5e-3 and rel_mean < 5e-4, (name, rel_max, rel_mean)
    # colour / opacity / uncertainty gradients through the activations (sigmoid, exp with the 0.01 clamp)
    sig = lambda x: 1.0 / (1.0 + np.exp(-x.astype(np.float64)))
    so, scol = sig(_np(sc["opacities"])), sig(_np(sc["colors"]))
    want_op = (vop.sum(0) * so * (1 - so)).astype(np.float32)
    want_col = (vcol[..., :3].sum(0) * scol * (1 - scol)).astype(np.float32)
    for name, got, want in (("logit_opacities", leaves["opacities"].grad, want_op),
                            ("logit_colors", leaves["colors"].grad, want_col)):
        got = _np(got)
        rel_max = float(np.abs(got - want).max() / (np.abs(want).max() + 1e-12))
        print(f"[configs[1] 100k] grad {name}: max rel {rel_max:.2e}")
        assert rel_max < 5e-3, (name, rel_max)


def test_config1_fused_ssim_640x480_vs_live_oracle(dev, oracle32):
    """fused_ssim('valid') forward value and gradient at the BASELINE image size"""
    from gslam_amd.ssim import fused_ssim
    g = torch.Generator().manual_seed(5)
    a = torch.rand(1, 3, 480, 640, generator=g)
    b = (a + 0.1 * torch.randn(1, 3, 480, 640, generator=g)).clamp(0, 1)
    x = a.to(dev).requires_grad_(True)
    val = fused_ssim(x, b.to(dev), padding="valid")
    val.backward()
    oval, ograd = oracle32.fused_ssim(a.numpy(), b.numpy(), "valid")
    assert abs(float(val) - float(oval)) < 2e-6
    got = _np(x.grad)
    rel = float(np.abs(got - ograd).max() / (np.abs(ograd).max() + 1e-12))
    print(f"[configs[1] ssim 640x480] value diff {abs(float(val) - float(oval)):.2e} grad max rel {rel:.2e}")
    assert rel < 1e-4


@pytest.mark.slow
def test_config2_500k_forward_and_lists_vs_live_oracle(dev, oracle32):
    """the map size of BASELINE.json configs[2] (500 k Gaussians, C = 1): forward render and tile lists, oracle live"""
    n, W, H = 500_000, 640, 480
    sc, viewmats, Ks, out, o, _, _ = _run_pair(dev, oracle32, n, with_backward=False)
    assert np.array_equal(_np(out.radii), o["radii"]) and np.array_equal(_np(out.tiles_per_gauss), o["tiles_per_gauss"])
    assert np.array_equal(_np(out.flatten_ids), o["flatten_ids"]) and np.array_equal(_np(out.isect_offsets), o["isect_offsets"])
    assert np.array_equal(_np(out.isect_ids), o["isect_ids"])
    l1, mx, px_off, nt_diff = _report("configs[2] 500k", out, o)
    assert l1 < 1e-5 and mx < 1.0 / 255.0 and px_off < 0.004 * W * H and nt_diff < 0.004 * n

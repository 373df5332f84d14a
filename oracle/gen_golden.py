"""Generates tests/golden/*.npz from the importable parts of the reference.  TEST INFRASTRUCTURE ONLY.

Runs ONLY in the build container (needs /root/reference).  The Python reference never leaves this container;
only the input/output vectors written here (data, not source) are committed.  Re-run:  python oracle/gen_golden.py

What is importable here: gslam.warp, gslam.utils, gslam.primitives (with ``pypose`` stubbed - it
is used only by unused alternates), and gslam.rasterization once ``gsplat.cuda._wrapper`` resolves.  The four
gsplat CUDA ops do not exist here, so G3 injects the build's CPU oracle ops under that module name: G3 pins the
reference's HOST LOGIC (activations, channel packing, beta background e^1, output split), not the kernels.
"""
import math
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from oracle.oracle import Oracle  # noqa: E402

O32 = Oracle(np.float32)


def _np(t):
    return t.detach().cpu().numpy()


def small_pose(rng, rot=0.02, trans=0.05):
    w = rng.normal(scale=rot, size=3)
    th = np.linalg.norm(w)
    Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = np.eye(3) + math.sin(th) / th * Kx + (1 - math.cos(th)) / th ** 2 * (Kx @ Kx)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = rng.normal(scale=trans, size=3)
    return T.astype(np.float32)


def gen_warp():
    sys.path.insert(0, REF)
    from gslam.warp import Warp
    for name, (H, W) in {"warp_48x64": (48, 64), "warp_120x160": (120, 160)}.items():
        rng = np.random.default_rng(10 + H)
        s = W / 640.0
        K = np.array([[525.0 * s, 0, 319.5 * s], [0, 525.0 * s, 239.5 * s], [0, 0, 1]], np.float32)
        c1 = rng.uniform(0, 1, (H, W, 3)).astype(np.float32)
        d1 = rng.uniform(1, 2, (H, W)).astype(np.float32)
        p1, p2 = small_pose(rng), small_pose(rng)
        warp = Warp(torch.from_numpy(K), H, W)
        t1 = torch.from_numpy(p1).requires_grad_(True)
        t2 = torch.from_numpy(p2).requires_grad_(True)
        result, nwarps, keep = warp(t1, t2, torch.from_numpy(c1), torch.from_numpy(d1))
        loss = result[keep].sum() + 0.1 * nwarps.square().sum()
        loss.backward()
        # identity-pose quirk (half-pixel shift)
        eye = torch.eye(4)
        res_id, _, _ = warp(eye, eye, torch.from_numpy(c1), torch.from_numpy(d1))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), K=K, c1=c1, d1=d1, f1_pose=p1, f2_pose=p2,
                            result=_np(result), normalized_warps=_np(nwarps), keep_mask=_np(keep),
                            grad_f1=_np(t1.grad), grad_f2=_np(t2.grad), loss=_np(loss), result_identity=_np(res_id))


def gen_utils():
    sys.path.insert(0, REF)
    from gslam.utils import StopOnPlateau, create_batch, edge_aware_tv
    rng = np.random.default_rng(20)
    depth = rng.uniform(0.5, 3, (2, 24, 32)).astype(np.float32)
    rgb = rng.uniform(0, 1, (2, 24, 32, 3)).astype(np.float32)
    alphas = rng.uniform(0, 1, (2, 24, 32, 1)).astype(np.float32)
    d = torch.from_numpy(depth).requires_grad_(True)
    r = torch.from_numpy(rgb).requires_grad_(True)
    mask = torch.from_numpy(alphas)[..., 0] > 0.4
    tv = edge_aware_tv(d, r, mask)
    tv.backward()
    losses = np.array([0.5, 0.4, 0.02, 0.011, 0.0105, 0.0104, 0.0108, 0.0103, 0.0102, 0.0101, 0.01, 0.0099],
                      np.float64)
    sp = StopOnPlateau(3, 0.012)
    stops = np.array([sp.stop(float(x)) for x in losses])
    batch = create_batch([torch.arange(3.0), torch.arange(3.0) + 1])
    np.savez_compressed(os.path.join(OUT, "utils.npz"), depth=depth, rgb=rgb, alphas=alphas, tv=_np(tv),
                        grad_depth=_np(d.grad), grad_rgb=_np(r.grad), plateau_losses=losses, plateau_stops=stops,
                        batch=_np(batch))


def gen_pose():
    sys.path.insert(0, REF)
    sys.modules.setdefault("pypose", mock.MagicMock())  # only the unused alternates touch it
    from gslam.primitives import PoseZhou, rotation_6d_to_matrix
    rng = np.random.default_rng(30)
    Rt = small_pose(rng, rot=0.3, trans=0.5)
    pose = PoseZhou(torch.from_numpy(Rt))
    dR = rng.normal(scale=0.05, size=6).astype(np.float32)
    dt = rng.normal(scale=0.05, size=3).astype(np.float32)
    with torch.no_grad():
        pose.dR.copy_(torch.from_numpy(dR))
        pose.dt.copy_(torch.from_numpy(dt))
    V = pose()
    w = rng.normal(size=(4, 4)).astype(np.float32)
    (V * torch.from_numpy(w)).sum().backward()
    d6 = rng.normal(size=(5, 6)).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "pose_zhou.npz"), Rt=Rt, dR=dR, dt=dt, viewmat=_np(V), w=w,
                        grad_dR=_np(pose.dR.grad), grad_dt=_np(pose.dt.grad), d6=d6,
                        rot6d=_np(rotation_6d_to_matrix(torch.from_numpy(d6))),
                        viewmat_fixed=_np(PoseZhou(torch.from_numpy(Rt), is_learnable=False)()))


def _install_oracle_as_gsplat():
    """gslam/rasterization.py:9-14 imports four CUDA ops; resolve them with the CPU oracle (host-logic check)."""
    def fully_fused_projection(means, covars, quats, scales, viewmats, Ks, width, height, eps2d=0.3, packed=False,
                               near_plane=0.01, far_plane=1e10, radius_clip=0.0, sparse_grad=False,
                               calc_compensations=False, camera_model="pinhole"):
        assert covars is None and not packed and camera_model == "pinhole"
        r, m, d, c, k = O32.project_fwd(_np(means), _np(quats), _np(scales), _np(viewmats), _np(Ks), width, height,
                                        eps2d, near_plane, far_plane, radius_clip, calc_compensations)
        f = torch.from_numpy
        return f(r), f(m), f(d), f(c), (None if k is None else f(k))

    def isect_tiles(means2d, radii, depths, tile_size, tile_width, tile_height, sort=True, packed=False,
                    n_cameras=None, camera_ids=None, gaussian_ids=None):
        t, i, fl = O32.isect_tiles(_np(means2d), _np(radii), _np(depths), tile_size, tile_width, tile_height, sort)
        return torch.from_numpy(t), torch.from_numpy(i), torch.from_numpy(fl)

    def isect_offset_encode(isect_ids, n_cameras, tile_width, tile_height):
        return torch.from_numpy(O32.isect_offset_encode(_np(isect_ids), n_cameras, tile_width, tile_height))

    def rasterize_to_pixels(means2d, conics, colors, opacities, image_width, image_height, tile_size, isect_offsets,
                            flatten_ids, backgrounds=None, masks=None, packed=False, absgrad=False,
                            visibility_min_T=0.5):
        bg = None if backgrounds is None else _np(backgrounds)
        r, a, _, nt = O32.raster_fwd(_np(means2d), _np(conics), _np(colors), _np(opacities), bg, image_width,
                                     image_height, tile_size, _np(isect_offsets), _np(flatten_ids), visibility_min_T)
        return torch.from_numpy(r), torch.from_numpy(a), torch.from_numpy(nt)

    pkg = types.ModuleType("gsplat")
    cuda = types.ModuleType("gsplat.cuda")
    wrap = types.ModuleType("gsplat.cuda._wrapper")
    wrap.fully_fused_projection = fully_fused_projection
    wrap.isect_tiles = isect_tiles
    wrap.isect_offset_encode = isect_offset_encode
    wrap.rasterize_to_pixels = rasterize_to_pixels
    pkg.cuda, cuda._wrapper = cuda, wrap
    sys.modules.update({"gsplat": pkg, "gsplat.cuda": cuda, "gsplat.cuda._wrapper": wrap})


def gen_rasterization_host_logic():
    sys.path.insert(0, REF)
    _install_oracle_as_gsplat()
    from gslam.rasterization import rasterization
    sys.path.insert(0, ROOT)
    from gslam_amd.synthetic import make_scene
    W, H, N, C = 64, 48, 300, 2
    sc = make_scene(N, seed=7)
    sc["means"][:, :2] *= 0.5
    K = torch.tensor([[52.5, 0, 31.5], [0, 52.5, 23.5], [0, 0, 1]])
    Ks = K[None].repeat(C, 1, 1)
    viewmats = torch.stack([torch.eye(4), torch.eye(4)])
    viewmats[1, 0, 3] = -0.1
    sc["scales"] = sc["scales"] + 1.0  # bigger splats for the tiny image
    sc["log_uncertainties"] = torch.linspace(-6.0, 1.0, N)  # exercises the 0.01 clamp (rasterization.py:149)
    save = {k: _np(v) for k, v in sc.items()}
    save.update(Ks=_np(Ks), viewmats=_np(viewmats), width=W, height=H)
    for mode in ("RGB", "RGB+D"):
        out = rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["colors"], viewmats, Ks, W, H,
                            packed=False, render_mode=mode, log_uncertainties=sc["log_uncertainties"],
                            backgrounds=torch.zeros(C, 3), visibility_min_T=0.5)
        tag = mode.replace("+", "p")
        for f in ("rgbs", "alphas", "depthmaps", "betas", "radii", "means2d", "depths", "conics", "opacities",
                  "n_touched", "tiles_per_gauss", "isect_ids", "flatten_ids", "isect_offsets"):
            v = getattr(out, f)
            if v is not None:
                save[f"{tag}__{f}"] = _np(v)
        save[f"{tag}__meta"] = np.array([out.tile_width, out.tile_height, out.width, out.height, out.tile_size,
                                         out.n_cameras])
    np.savez_compressed(os.path.join(OUT, "rasterization_host_logic.npz"), **save)


def gen_trajectory():
    """gslam/trajectory.py:14-97 - kabsch_umeyama, average_translation_error, evaluate_trajectories (the figure is dropped) -
    on seeded trajectories: a similarity-transformed noisy copy, a mirrored one (the determinant correction), a short one."""
    sys.path.insert(0, REF)
    sys.modules.setdefault("pypose", mock.MagicMock())
    from gslam.trajectory import average_translation_error, evaluate_trajectories, kabsch_umeyama
    rng = np.random.default_rng(40)
    save = {}
    cases = {}
    n = 40
    t_ = np.linspace(0, 1, n)
    A = np.stack([np.sin(2 * t_), 0.3 * t_, np.cos(3 * t_) * 0.5], 1) + rng.normal(scale=0.002, size=(n, 3))
    Rg = small_pose(rng, rot=0.8, trans=0.0)[:3, :3].astype(np.float64)
    B = (A - 0.2) @ Rg.T * 1.7 + np.array([0.5, -0.1, 0.3]) + rng.normal(scale=0.004, size=(n, 3))
    cases["similar"] = (A, B)
    cases["mirrored"] = (A, B * np.array([1.0, 1.0, -1.0]))
    cases["short"] = (A[:3], B[:3])
    for name, (a, b) in cases.items():
        R, c, t = kabsch_umeyama(a, b)
        save[f"{name}__A"], save[f"{name}__B"] = a, b
        save[f"{name}__R"], save[f"{name}__c"], save[f"{name}__t"] = R, np.float64(c), t
        save[f"{name}__ate"] = np.float64(average_translation_error(a, b))

    class F:                                            # what evaluate_trajectories reads of a Frame
        def __init__(self, gt, est):
            self.gt_pose, self._est = torch.from_numpy(gt), torch.from_numpy(est)

        def pose(self):
            return self._est

    def frames(a, b):
        out = []
        for i in range(len(a)):
            g, e = np.eye(4, dtype=np.float32), np.eye(4, dtype=np.float32)
            g[:3, 3], e[:3, 3] = a[i], b[i]
            out.append(F(g, e))
        return out
    trajs = {"tracking": frames(*cases["similar"]), "keyframes": frames(cases["similar"][0][::5], cases["similar"][1][::5])}
    fig, ates = evaluate_trajectories(trajs, keyframe_indices=[0, 5, 10])
    for k, v in ates.items():
        save[f"eval__{k}"] = np.float64(v)
    np.savez_compressed(os.path.join(OUT, "trajectory.npz"), **save)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    if "--only-trajectory" in sys.argv:
        gen_trajectory()
        sys.exit(0)
    gen_warp()
    gen_utils()
    gen_pose()
    gen_rasterization_host_logic()
    gen_trajectory()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))

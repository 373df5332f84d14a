"""Pins the CPU oracle against golden vectors produced by the importable parts of the reference
(oracle/gen_golden.py -> tests/golden/*.npz).  CPU only; never reads /root/reference."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return dict(np.load(os.path.join(GOLD, name)))


@pytest.mark.parametrize("name", ["warp_48x64.npz", "warp_120x160.npz"])
def test_warp_matches_reference(oracle32, name):
    g = _load(name)
    f1 = torch.from_numpy(g["f1_pose"]).requires_grad_(True)
    f2 = torch.from_numpy(g["f2_pose"]).requires_grad_(True)
    T = f1 @ torch.linalg.inv(f2)                                     # gslam/warp.py:44
    K = g["K"]
    Kinv = torch.linalg.inv(torch.from_numpy(K)).numpy()             # gslam/warp.py:13
    res, nw, mask = oracle32.warp_fwd(T.detach().numpy(), K, Kinv, g["c1"], g["d1"])
    np.testing.assert_allclose(nw, g["normalized_warps"], atol=2e-6)
    # pixels whose warp falls within 1e-5 of the +-1 border may flip the strict inequality
    border = (np.abs(np.abs(g["normalized_warps"][0]) - 1.0) < 1e-5).any(-1)
    assert (mask == g["keep_mask"])[~border].all()
    # fp32 round-off of the sample position scales with the image width; bound max and mean error
    np.testing.assert_allclose(res, g["result"], atol=1e-4)
    assert np.abs(res - g["result"]).mean() < 5e-6
    # backward: loss = result[keep].sum() + 0.1*nwarps^2.sum()
    v_res = np.repeat(g["keep_mask"][..., None], 3, -1).astype(np.float32)
    v_nw = 0.2 * g["normalized_warps"]
    vT = oracle32.warp_bwd(T.detach().numpy(), K, Kinv, g["c1"], g["d1"], v_res, v_nw)
    T.backward(torch.from_numpy(vT))
    scale = max(np.abs(g["grad_f1"]).max(), 1.0)
    np.testing.assert_allclose(f1.grad.numpy()[:3], g["grad_f1"][:3], atol=2e-4 * scale)
    np.testing.assert_allclose(f2.grad.numpy()[:3], g["grad_f2"][:3], atol=2e-4 * scale)


def test_warp_identity_half_pixel_quirk(oracle32):
    """align_corners=False with u*2/W-1 normalisation: identity poses do NOT reproduce c1."""
    g = _load("warp_48x64.npz")
    K = g["K"]
    Kinv = torch.linalg.inv(torch.from_numpy(K)).numpy()
    res, _, _ = oracle32.warp_fwd(np.eye(4, dtype=np.float32), K, Kinv, g["c1"], g["d1"])
    np.testing.assert_allclose(res, g["result_identity"], atol=2e-5)
    assert np.abs(res - g["c1"]).max() > 0.1


@pytest.mark.parametrize("mode", ["RGB", "RGB+D"])
def test_rasterization_host_logic_matches_reference(oracle32, mode):
    g = _load("rasterization_host_logic.npz")
    tag = mode.replace("+", "p")
    W, H = int(g["width"]), int(g["height"])
    C = g["viewmats"].shape[0]
    scales = torch.exp(torch.from_numpy(g["scales"])).numpy()       # same exp as the reference run
    out = oracle32.gslam_rasterization(g["means"], g["quats"], g["scales"], g["opacities"], g["colors"],
                                       g["viewmats"], g["Ks"], W, H, render_mode=mode,
                                       log_uncertainties=g["log_uncertainties"],
                                       backgrounds=np.zeros((C, 3), np.float32), scales_override=scales)
    for f in ("radii", "tiles_per_gauss", "isect_ids", "flatten_ids", "isect_offsets"):
        assert np.array_equal(out[f], g[f"{tag}__{f}"]), f
    for f in ("means2d", "depths", "conics"):
        assert np.array_equal(out[f], g[f"{tag}__{f}"]), f
    np.testing.assert_allclose(out["opacities"], g[f"{tag}__opacities"], atol=1e-6)
    np.testing.assert_allclose(out["rgbs"], g[f"{tag}__rgbs"], atol=2e-6)
    np.testing.assert_allclose(out["alphas"], g[f"{tag}__alphas"], atol=2e-6)
    np.testing.assert_allclose(out["betas"], g[f"{tag}__betas"], atol=5e-6)
    if mode == "RGB+D":
        np.testing.assert_allclose(out["depthmaps"], g[f"{tag}__depthmaps"], atol=1e-5)
    else:
        assert "depthmaps" not in out
    nt, ntg = out["n_touched"], g[f"{tag}__n_touched"]
    assert nt.dtype == np.int64 and (nt != ntg).mean() < 1e-3
    assert list(g[f"{tag}__meta"]) == [out["tile_width"], out["tile_height"], W, H, 16, C]
    # beta background is e^1 where nothing was rendered (rasterization.py:253)
    empty = out["alphas"][..., 0] == 0
    if empty.any():
        np.testing.assert_allclose(out["betas"][empty], np.e, rtol=1e-6)

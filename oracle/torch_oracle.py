"""A SECOND, independent restatement of the external kernels in vectorised torch-CPU ops.  TEST INFRASTRUCTURE ONLY.

SURVEY.md §8(c) asks for two cross-checks of the C oracle (oracle/gsx_oracle.c) beyond finite differences:

* G4 - the projection (K1), the tile rectangles (K3) and the rasteriser (K8) written a second time, from the spec in
  SURVEY.md §9.1-9.4 and not from the C file, as dense tensor algebra whose BACKWARD is torch autograd: the C oracle's
  hand-derived VJPs (K2, K9) are then held to an automatic derivative of an independent forward, in float64;
* G6 - fused-ssim (K11/K12) as five zero-padded ``F.conv2d`` with the 11-tap sigma = 1.5 window, cropped by 5 for
  ``padding='valid'``, gradient by autograd (reference call site: gslam/backend.py:303-307).

Nothing here is on the product path; only tests/ import it.  The constants are the INFERRED ones of SURVEY.md §9
(fov slack 0.3, radius floor 0.01, 3 sigma, alpha in [1/255, 0.999], T_min 1e-4, C1 = 1e-4, C2 = 9e-4): a wrong reading of
the published kernels would be wrong in both restatements - what this module removes is the risk of a slip in ONE of them
(an index, a sign, a transposed Jacobian, a mis-derived VJP).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

ALPHA_MAX, ALPHA_MIN, T_MIN = 0.999, 1.0 / 255.0, 1e-4


# ---- K1: 3-D covariance -> 2-D projection -----------------------------------------------------------
def quat_to_rotmat(q: torch.Tensor) -> torch.Tensor:
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(q.shape[:-1] + (3, 3))


def project(means, quats, scales, viewmats, Ks, W, H, eps2d=0.3, near=0.01, far=1e10, radius_clip=0.0):
    """-> radii [C,N] int32, means2d [C,N,2], depths [C,N], conics [C,N,3]; culled rows are zeros.  Differentiable in means,
    quats, scales, viewmats (the cull decisions are constants of the graph, as in the kernels)."""
    dt = means.dtype
    R, t = viewmats[:, :3, :3], viewmats[:, :3, 3]                         # [C,3,3], [C,3]
    p = torch.einsum('cij,nj->cni', R, means) + t[:, None, :]              # [C,N,3]
    Rq = quat_to_rotmat(quats)
    Mm = Rq * scales[:, None, :]                                           # Rq diag(s)
    Sig = Mm @ Mm.transpose(-1, -2)                                        # [N,3,3]
    Sc = torch.einsum('cij,njk,clk->cnil', R, Sig, R)                      # R Sigma R^T  [C,N,3,3]
    fx, fy, cx, cy = Ks[:, 0, 0], Ks[:, 1, 1], Ks[:, 0, 2], Ks[:, 1, 2]
    tanx, tany = 0.5 * W / fx, 0.5 * H / fy
    lim_xp, lim_xn = (W - cx) / fx + 0.3 * tanx, cx / fx + 0.3 * tanx
    lim_yp, lim_yn = (H - cy) / fy + 0.3 * tany, cy / fy + 0.3 * tany
    z = p[..., 2]
    rz = 1.0 / z
    tx = z * torch.minimum(lim_xp[:, None], torch.maximum(-lim_xn[:, None], p[..., 0] * rz))
    ty = z * torch.minimum(lim_yp[:, None], torch.maximum(-lim_yn[:, None], p[..., 1] * rz))
    zero = torch.zeros_like(z)
    J = torch.stack([torch.stack([fx[:, None] * rz, zero, -fx[:, None] * tx * rz * rz], -1),
                     torch.stack([zero, fy[:, None] * rz, -fy[:, None] * ty * rz * rz], -1)], -2)   # [C,N,2,3]
    S2 = J @ Sc @ J.transpose(-1, -2)                                      # [C,N,2,2]
    a = S2[..., 0, 0] + eps2d
    b = S2[..., 0, 1]
    c = S2[..., 1, 1] + eps2d
    det = a * c - b * b
    m2d = torch.stack([fx[:, None] * p[..., 0] * rz + cx[:, None], fy[:, None] * p[..., 1] * rz + cy[:, None]], -1)
    conic = torch.stack([c / det, -b / det, a / det], -1)
    mid = 0.5 * (a + c)
    radius = torch.ceil(3.0 * torch.sqrt(mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.01))))
    ok = (z >= near) & (z <= far) & (det > 0) & (radius > radius_clip)
    ok = ok & (m2d[..., 0] + radius > 0) & (m2d[..., 0] - radius < W) & (m2d[..., 1] + radius > 0) & (m2d[..., 1] - radius < H)
    radii = torch.where(ok, radius, torch.zeros_like(radius)).to(torch.int32)
    okf = ok.to(dt)
    # culled rows: zeros (and no NaN leaks into the graph from a division by a non-positive determinant)
    safe = lambda v, m: torch.where(m, v, torch.zeros_like(v))
    return radii, safe(m2d, ok[..., None]), safe(z, ok), safe(conic, ok[..., None]), okf


# ---- K3: tile rectangles -----------------------------------------------------------------------------
def tile_rects(means2d, radii, tile_size, tile_w, tile_h):
    """float32 arithmetic as the kernels: -> (x0, y0, x1, y1) int64 [C,N], half-open tile ranges (empty where radius <= 0)"""
    m = means2d.detach().to(torch.float32)
    r = radii.to(torch.float32) / float(tile_size)
    tx, ty = m[..., 0] / float(tile_size), m[..., 1] / float(tile_size)
    x0 = torch.floor(tx - r).clamp(0, tile_w).long()
    x1 = torch.ceil(tx + r).clamp(0, tile_w).long()
    y0 = torch.floor(ty - r).clamp(0, tile_h).long()
    y1 = torch.ceil(ty + r).clamp(0, tile_h).long()
    vis = radii > 0
    z = torch.zeros_like(x0)
    return torch.where(vis, x0, z), torch.where(vis, y0, z), torch.where(vis, x1, z), torch.where(vis, y1, z)


# ---- K8 (+ K9 by autograd): tiled alpha compositing ---------------------------------------------------
def rasterize(means2d, conics, colors, opacities, radii, depths, W, H, backgrounds=None, tile_size=16):
    """Dense per-tile evaluation: for every 16x16 tile the Gaussians whose tile rectangle covers it, ordered by
    (float32 depth bits, flatten id), composited front to back for its 256 pixels at once.  -> render [C,H,W,CH],
    alphas [C,H,W,1], last [C,H,W] (rank of the last contributing entry inside its tile's list, -1 if none),
    counts [C,tile_h,tile_w].  Differentiable in means2d, conics, colors, opacities, backgrounds."""
    Cn, N, CH = colors.shape
    dt = colors.dtype
    tile_w, tile_h = math.ceil(W / tile_size), math.ceil(H / tile_size)
    x0, y0, x1, y1 = tile_rects(means2d, radii, tile_size, tile_w, tile_h)
    dbits = depths.detach().to(torch.float32).contiguous().view(torch.int32).long()
    render = torch.zeros(Cn, H, W, CH, dtype=dt)
    alphas = torch.zeros(Cn, H, W, 1, dtype=dt)
    last = torch.full((Cn, H, W), -1, dtype=torch.int64)
    counts = torch.zeros(Cn, tile_h, tile_w, dtype=torch.int64)
    rows, rows_a = [], []
    for c in range(Cn):
        for ti in range(tile_h):
            for tj in range(tile_w):
                sel = torch.nonzero((x0[c] <= tj) & (tj < x1[c]) & (y0[c] <= ti) & (ti < y1[c])).squeeze(1)
                py, px = torch.meshgrid(torch.arange(ti * tile_size, min((ti + 1) * tile_size, H)),
                                        torch.arange(tj * tile_size, min((tj + 1) * tile_size, W)), indexing='ij')
                P = py.numel()
                bg = backgrounds[c] if backgrounds is not None else torch.zeros(CH, dtype=dt)
                counts[c, ti, tj] = sel.numel()
                if sel.numel() == 0:
                    rows.append((c, py, px, bg.expand(P, CH), torch.zeros(P, dtype=dt), torch.full((P,), -1)))
                    continue
                key = dbits[c, sel] * (Cn * N) + (c * N + sel)                # (depth bits, flatten id): unique, stable order
                sel = sel[torch.argsort(key)]
                fxp = px.reshape(-1).to(dt) + 0.5
                fyp = py.reshape(-1).to(dt) + 0.5
                dx = means2d[c, sel, 0][None, :] - fxp[:, None]               # [P,n]
                dy = means2d[c, sel, 1][None, :] - fyp[:, None]
                ca, cb, cc = conics[c, sel, 0][None], conics[c, sel, 1][None], conics[c, sel, 2][None]
                sig = 0.5 * (ca * dx * dx + cc * dy * dy) + cb * dx * dy
                al = torch.clamp(opacities[c, sel][None] * torch.exp(-sig), max=ALPHA_MAX)
                valid = (sig.detach() >= 0) & (al.detach() >= ALPHA_MIN)
                ae = al * valid.to(dt)
                T_after = torch.cumprod(1.0 - ae, dim=1)
                stopped = T_after.detach() <= T_MIN                            # monotone: once stopped, always stopped
                use = valid & ~stopped
                au = al * use.to(dt)
                T_after_u = torch.cumprod(1.0 - au, dim=1)
                T_before = torch.cat([torch.ones(P, 1, dtype=dt), T_after_u[:, :-1]], dim=1)
                wgt = au * T_before                                            # [P,n]
                out = wgt @ colors[c, sel]                                     # [P,CH]
                T_fin = T_after_u[:, -1]
                idx = torch.arange(sel.numel())[None, :].expand(P, -1)
                lst = torch.where(use, idx, torch.full_like(idx, -1)).max(dim=1).values
                rows.append((c, py, px, out + T_fin[:, None] * bg[None, :], 1.0 - T_fin, lst))
    # assemble without in-place writes on graph tensors
    render_parts = torch.zeros(Cn, H, W, CH, dtype=dt)
    idx_c, idx_y, idx_x, vals, avals, lvals = [], [], [], [], [], []
    for c, py, px, v, a, l in rows:
        idx_c.append(torch.full((py.numel(),), c, dtype=torch.long))
        idx_y.append(py.reshape(-1)); idx_x.append(px.reshape(-1)); vals.append(v); avals.append(a); lvals.append(l)
    ic, iy, ix = torch.cat(idx_c), torch.cat(idx_y), torch.cat(idx_x)
    render = render_parts.index_put((ic, iy, ix), torch.cat(vals))
    alphas = torch.zeros(Cn, H, W, dtype=dt).index_put((ic, iy, ix), torch.cat(avals))[..., None]
    last = last.index_put((ic, iy, ix), torch.cat(lvals))
    return render, alphas, last, counts


# ---- K11 / K12: fused-ssim ----------------------------------------------------------------------------
def gaussian_window(dtype=torch.float64) -> torch.Tensor:
    x = torch.arange(11, dtype=torch.float64) - 5.0
    g = torch.exp(-(x * x) / (2.0 * 1.5 * 1.5))
    return (g / g.sum()).to(dtype)


def ssim_map(img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:
    """[B,CH,H,W] x 2 -> SSIM map [B,CH,H,W]; every moment is the same zero-padded 'same' convolution with the separable
    11-tap window (weights are not renormalised at the border)"""
    B, CH, H, W = img1.shape
    g = gaussian_window(img1.dtype)
    k2 = (g[:, None] * g[None, :])[None, None].expand(CH, 1, 11, 11).contiguous()
    conv = lambda x: F.conv2d(x, k2, padding=5, groups=CH)
    mu1, mu2 = conv(img1), conv(img2)
    s11 = conv(img1 * img1) - mu1 * mu1
    s22 = conv(img2 * img2) - mu2 * mu2
    s12 = conv(img1 * img2) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2))


def fused_ssim(img1: torch.Tensor, img2: torch.Tensor, padding: str = 'same') -> torch.Tensor:
    m = ssim_map(img1, img2)
    if padding == 'valid':
        m = m[:, :, 5:-5, 5:-5]
    return m.mean()

"""fused SSIM forward + backward at the BA shape (8 x 3 x 480 x 640) for two input layouts: planar contiguous [B,3,H,W] and the
rasteriser's channel-interleaved render ([B,H,W,5] viewed as [B,3,H,W] with strides) - is the layout the limiter?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gslam_amd.ssim import fused_ssim  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B, H, W = 8, 480, 640
    g = torch.Generator().manual_seed(0)
    render = torch.rand(B, H, W, 5, generator=g).to(dev)
    gt_nhwc = torch.rand(B, H, W, 3, generator=g).to(dev)
    cases = {
        "interleaved render [B,H,W,5] + NHWC gt": (render[..., :3].permute(0, 3, 1, 2), gt_nhwc.permute(0, 3, 1, 2)),
        "planar contiguous both": (render[..., :3].permute(0, 3, 1, 2).contiguous(), gt_nhwc.permute(0, 3, 1, 2).contiguous()),
    }
    for name, (a, b) in cases.items():
        a = a.detach().requires_grad_(True)
        for phase in ("fwd", "fwd+bwd"):
            def run():
                v = fused_ssim(a, b, padding="valid")
                if phase == "fwd+bwd":
                    a.grad = None
                    v.backward()
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                run()
            torch.cuda.synchronize()
            print(f"{name:45s} {phase:8s} {(time.perf_counter() - t0) / 50 * 1e6:8.1f} us")


main()

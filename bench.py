#!/usr/bin/env python3
"""bench.py — headline benchmark of the gslam hot path on MI355X (contract: see the task prompt / DESIGN.md §Measurement).

A "step" is one keyframe bundle-adjustment iteration of the reference's mapping loop (gslam/backend.py:260-359) on
synthetic TUM-shape input: differentiable render forward (RGB + depth + beta, CH = 5) of the window's keyframes ->
loss (active-NeRF photometric + 0.2*(1-fused_ssim 'valid') + isotropic + edge-aware depth TV) -> backward through
K9/K2 -> [all-reduce of the [N,15] gradient bucket when N_gpus > 1] -> fused Adam on the six splat tensors + poses
-> opacity decay.  Nothing is skipped inside the timed region.

Workload (BASELINE.json configs[1]): 100 k Gaussians, 640x480, one keyframe per GPU (weak scaling: the window has
n_gpus keyframes, every rank renders one against its replica of the map).  value = keyframe renders fwd+bwd per second
over the whole job.  Inputs are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--gaussians 100000] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)   # the first ~50 steps after an idle GPU run ~6 % slow (clock ramp)
    ap.add_argument("--gaussians", type=int, default=100_000)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-timing", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a HIP graph")
    return ap.parse_args()


def make_window(n_frames, W, H, dev, gt_scene, own):
    """Synthetic keyframes: pose c = 0.05*c m along x + 1 deg*c yaw; gt = render of scene(seed+1)."""
    from gslam_amd.primitives import Camera, Frame, PoseZhou
    from gslam_amd.synthetic import make_intrinsics, make_viewmat
    K = make_intrinsics(W, H).to(dev)
    frames = []
    for c in range(n_frames):
        V = make_viewmat(c).to(dev)
        pose = PoseZhou(V, is_learnable=True).to(dev)
        img = None
        if c in own:
            with torch.no_grad():
                out = gt_scene([Camera(K, H, W)], [pose], render_depth=False)
                img = out.rgbs[0].clamp(0, 1).contiguous()
        frames.append(Frame(img=img, timestamp=c / 30.0, camera=Camera(K, H, W), pose=pose, gt_pose=V, index=c,
                            exposure_params=torch.zeros(2, device=dev)))
    return frames


class StageTimer:
    """Times every C-ABI launch with HIP events on the stream the kernels are launched on (torch's current stream)."""
    STAGES = ("gsx_pose_zhou_fwd", "gsx_project_fwd", "gsx_isect_bin_sort", "gsx_isect_scan", "gsx_isect_emit_sort",
              "gsx_isect_offset_encode", "gsx_raster_fwd", "gsx_ssim_fwd", "gsx_ssim_bwd", "gsx_map_loss",
              "gsx_raster_bwd", "gsx_project_bwd", "gsx_pose_zhou_bwd", "gsx_isotropic_loss", "gsx_isotropic_loss_acc",
              "gsx_loss_finish", "gsx_counters_add", "gsx_adam_multi", "gsx_adam_multi_steps",
              "gsx_adam_multi_steps_decay", "gsx_opacity_decay")

    def __init__(self):
        self.events = {s: [] for s in self.STAGES}

    def __enter__(self):
        from gslam_amd import _lib
        self._lib = _lib.lib
        self._orig = {}
        for s in self.STAGES:
            orig = getattr(self._lib, s)
            self._orig[s] = orig

            def wrapped(*a, _o=orig, _s=s):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = _o(*a)
                e1.record()
                self.events[_s].append((e0, e1))
                return rc

            setattr(self._lib, s, wrapped)
        return self

    def __exit__(self, *exc):
        for s, o in self._orig.items():
            setattr(self._lib, s, o)
        torch.cuda.synchronize()

    def mean_us(self, skip=0):
        out = {}
        for s, ev in self.events.items():
            ts = [a.elapsed_time(b) * 1e3 for a, b in ev]
            per_step = ts[skip:]
            if per_step:
                out[s] = sum(per_step) / len(per_step)
        return out


def algorithmic_bytes(N, C, M, P, CH):
    """SURVEY.md §8(d) per-launch algorithmic bytes (every array touched once)."""
    return {
        "gsx_project_fwd": C * N * (40 + 28),
        "gsx_isect_emit_sort": M * 12 + M * 24,
        "gsx_isect_bin_sort": C * N * 16 * 2 + M * 8 + M * 8 + M * 4,
        "gsx_map_loss": P * (20 + 4 + 12 + 12 + 20),
        "gsx_isect_offset_encode": M * 8,
        "gsx_raster_fwd": M * (28 + 4 * CH) + P * (4 * CH + 8) + C * N * 4,
        "gsx_raster_bwd": P * (4 * CH + 12) + M * (28 + 4 * CH) + C * N * (24 + 4 * CH),
        "gsx_project_bwd": C * N * (40 + 28 + 24) + N * 40 + C * 64,
        "gsx_ssim_fwd": 72 * P,
        "gsx_ssim_bwd": 72 * P,
        "gsx_adam_multi": 420 * N,
        "gsx_adam_multi_steps": 420 * N,
        "gsx_adam_multi_steps_decay": 420 * N + 4 * N,
    }


def cpu_baseline(N, W, H, budget_s=25.0):
    """CPU port (the oracle, scalar C, 1 core) of the same step on the same synthetic inputs; bounded sample."""
    import numpy as np
    from oracle.oracle import Oracle
    from gslam_amd.synthetic import make_cameras, make_scene
    o = Oracle(np.float32)
    sc = {k: v.numpy() for k, v in make_scene(N, 0).items()}
    viewmats, Ks = make_cameras(1, W, H)
    viewmats, Ks = viewmats.numpy(), Ks.numpy()
    gt = np.random.default_rng(1).uniform(0, 1, (1, H, W, 3)).astype(np.float32)
    times = []
    t_start = time.perf_counter()
    while True:
        t0 = time.perf_counter()
        out = o.gslam_rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["colors"], viewmats, Ks,
                                    W, H, render_mode="RGB+D", log_uncertainties=sc["log_uncertainties"],
                                    backgrounds=np.zeros((1, 3), np.float32))
        rgb = np.ascontiguousarray(out["rgbs"].transpose(0, 3, 1, 2))
        _, g_ssim = o.fused_ssim(rgb, np.ascontiguousarray(gt.transpose(0, 3, 1, 2)), "valid")
        v_render = np.zeros_like(out["render"])
        v_render[..., :3] = (out["rgbs"] - gt) * 1e-3 - 0.2 * g_ssim.transpose(0, 2, 3, 1)
        v_render[..., 3:] = 1e-6
        vm, vc, vcol, vop, _ = o.raster_bwd(out["means2d"], out["conics"], out["colors_packed"], out["opacities"],
                                            out["backgrounds_packed"], W, H, 16, out["isect_offsets"],
                                            out["flatten_ids"], out["alphas"], out["last_ids"], v_render,
                                            np.zeros_like(out["alphas"]))
        scales = np.exp(sc["scales"])
        g = o.project_bwd(sc["means"], sc["quats"], scales, viewmats, Ks, W, H, out["radii"], vm, vcol[..., 3], vc)
        for p, gr in ((sc["means"], g[0]), (sc["quats"], g[1]), (sc["scales"], g[2] * scales)):
            o.adam(p, gr, np.zeros_like(p), np.zeros_like(p), 1e-3, 1)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start + times[-1] > budget_s or len(times) >= 5:
            break
    times.sort()
    med = times[len(times) // 2]
    return {"value": 1.0 / med, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{len(times)} full steps (fwd+bwd+SSIM+Adam) of the same {N}-Gaussian {W}x{H} C=1 workload, "
                      f"median {med:.2f} s/step, oracle/gsx_oracle.c scalar C on 1 host core"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
            sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback for the product path)"
    # rehearsal knobs (1-GPU box): GSX_FORCE_DEVICE=0 puts every rank on cuda:0, GSX_DIST_BACKEND=gloo avoids RCCL's
    # duplicate-GPU check.  The driver's multi-GPU runs use neither: one rank per GPU over RCCL/xGMI.
    dev_index = int(os.environ.get("GSX_FORCE_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    from gslam_amd import dist as gdist
    import torch.distributed as td
    gdist.init_from_env(backend=os.environ.get("GSX_DIST_BACKEND"), device=dev)

    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.mapping import BundleAdjuster, MapConfig
    from gslam_amd.synthetic import make_scene

    N, W, H = args.gaussians, args.width, args.height
    splats = GaussianSplattingData.from_dict(make_scene(N, 0), dev)
    gt_scene = GaussianSplattingData.from_dict(make_scene(N, 1), dev)
    own = {c for c in range(world) if c % world == rank}
    window = make_window(world, W, H, dev, gt_scene, own)
    del gt_scene
    use_graph = not args.no_graph
    ba = BundleAdjuster(splats, MapConfig(), capturable=use_graph)
    step_fn = lambda: ba.step(window)
    if use_graph:
        from gslam_amd.mapping import GraphedBundleAdjuster
        gba, ok = None, 1
        try:
            gba = GraphedBundleAdjuster(ba, window)
        except Exception as e:  # e.g. a runtime that cannot capture next to a live RCCL communicator
            ok = 0
            print(f"bench.py[rank {rank}]: HIP-graph capture failed ({e!r}); running eagerly", file=sys.stderr)
        if world > 1:           # all ranks must take the same path
            flag = torch.tensor([ok], device=dev, dtype=torch.int32)
            td.all_reduce(flag, op=td.ReduceOp.MIN)
            ok = int(flag.item())
        if ok:
            step_fn = gba.step
        else:
            use_graph = False

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()

    from gslam_amd.rasterization import validate
    for attempt in range(3):
        for _ in range(args.warmup):
            step_fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_fn()
        barrier()
        elapsed = time.perf_counter() - t0
        # the render is sync-free (no M read-back inside a step); check afterwards that no step overflowed its
        # intersection buffers - a truncated step would be skipped work, so such a measurement is discarded
        if validate(dev):
            break
        print(f"bench.py: intersection capacity overflow on attempt {attempt}, re-running", file=sys.stderr)
        if use_graph:
            gba = GraphedBundleAdjuster(ba, window)
            step_fn = gba.step
        else:
            step_fn = lambda: ba.step(window)
    else:
        raise RuntimeError("intersection buffers kept overflowing")
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    value = world * args.steps / elapsed

    roofline, stages = None, None
    if not args.no_stage_timing:
        # every rank runs the instrumented steps (they contain the collectives); only rank 0 keeps the timers
        if rank == 0:
            with StageTimer() as st:
                for _ in range(10):
                    ba.step(window)
            stages = st.mean_us(skip=0)
        else:
            for _ in range(10):
                ba.step(window)
    if rank == 0 and stages is not None:
        out = ba.last_outputs
        M = int(out.flatten_ids.shape[0])
        P = H * W
        algo = algorithmic_bytes(N, 1, M, P, 5)
        dom = max((s for s in stages if s in algo), key=lambda s: stages[s])
        achieved = algo[dom] / (stages[dom] * 1e-6) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_r01.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("workload_gaussians") == N and tj.get("kernel") == dom:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "algorithmic_bytes": int(algo[dom]), "avg_launch_us": round(stages[dom], 2), "n_isects": M}

    if rank == 0:
        line = {
            "metric": "keyframe BA render fwd+bwd fps @640x480 / N Gaussians (tracking+mapping hot path)",
            "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: 100k Gaussians, 640x480, render fwd+bwd (RGB+D+beta, CH=5) "
                                   "+ fused-SSIM + full mapping loss + fused Adam; 1 keyframe per GPU",
                       "gaussians": N, "width": W, "height": H, "keyframes_per_gpu": 1, "window": world,
                       "parallelism": f"keyframe-sharded BA x{world}, 1 all-reduce of the [N,15] grad bucket",
                       "launch": ("eager" if not use_graph else "hip-graph replay of the whole step" if world == 1 else
                                  "hip-graph replay (render+loss+backward | Adam) around one eager all-reduce")},
        }
        if roofline is not None:
            line["roofline"] = roofline
            line["stage_us"] = {k: round(v, 2) for k, v in stages.items()}
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(N, W, H)
            except Exception as e:  # the baseline is informational; never lose the GPU number over it
                line["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 1, "kind": "port",
                                        "sample": f"failed: {e!r}"}
        print(json.dumps(line), flush=True)
    if world > 1:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()

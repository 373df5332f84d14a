"""CPU-only tests (-m "not gpu"): the C-ABI library loads and exports every symbol include/gsx.h declares, the host
mirror of the reference interface matches the reference's golden vectors, the product refuses to run on the CPU, and
the multi-GPU bundle-adjustment plumbing is exercised with world_size-2 gloo."""
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_library_is_rebuilt_from_current_sources():
    """a stale libgsx.so (sources edited, library not rebuilt) must never be what the tests or the GPU box run"""
    from gslam_amd.csrc.build import OUT, SOURCES, HERE, build
    build()                                                    # no-op when fresh, recompiles what changed
    so = os.path.getmtime(OUT)
    for src in list(SOURCES) + ["gsx_common.h", "raster_v4.inc", "project_core.h", "track_opt.h"]:
        assert os.path.getmtime(os.path.join(HERE, src)) <= so, f"{src} is newer than libgsx.so"


def test_library_exports_every_declared_symbol():
    import ctypes as C
    from gslam_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "gsx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gsx_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    raw = C.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        getattr(raw, name)                                    # dlsym; AttributeError if missing
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert _lib.lib.gsx_version() >= 100
    assert _lib.lib.gsx_record_stride(5) == 12 and _lib.lib.gsx_record_stride(2) == 8
    assert _lib.lib.gsx_record_stride(6) < 0


def test_argument_validation_without_gpu():
    """error plumbing: bad arguments are rejected before anything touches a device"""
    from gslam_amd._lib import lib
    rc = lib.gsx_isect_count(None, None, 10, 40, 30, None, None)
    assert rc == -1 and b"invalid argument" in lib.gsx_last_error()
    rc = lib.gsx_raster_fwd(None, 9, None, None, None, 0, 0, 1, 640, 480, 40, 30, 0.5, None, None, None, None, None,
                            None)
    assert rc < 0


def test_product_fails_loudly_on_cpu_tensors():
    from gslam_amd import ops
    from gslam_amd._lib import GsxError
    from gslam_amd.ssim import fused_ssim
    from gslam_amd.warp import Warp
    m = torch.zeros(4, 3)
    with pytest.raises(GsxError):
        ops.fully_fused_projection(m, None, torch.ones(4, 4), torch.ones(4, 3), torch.eye(4)[None], torch.eye(3)[None],
                                   64, 48)
    with pytest.raises(GsxError):
        fused_ssim(torch.zeros(1, 3, 32, 32), torch.zeros(1, 3, 32, 32))
    with pytest.raises(GsxError):
        Warp(torch.eye(3), 8, 8)(torch.eye(4), torch.eye(4), torch.zeros(8, 8, 3), torch.ones(8, 8))


def test_no_product_module_imports_the_oracle():
    """the oracle is test infrastructure: nothing under gslam_amd/ may import, link or call it"""
    pkg = os.path.join(ROOT, "gslam_amd")
    bad = re.compile(r"(^|\n)\s*(import\s+oracle|from\s+oracle)|gsxo_|libgsx_oracle|oracle/|oracle\.oracle|#include\s+\".*oracle")
    n = 0
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc")):
                n += 1
                src = open(os.path.join(dirpath, f)).read()
                assert not bad.search(src), f
    assert n > 20


def test_pose_zhou_matches_reference():
    from gslam_amd.primitives import PoseZhou, rotation_6d_to_matrix
    g = dict(np.load(os.path.join(GOLD, "pose_zhou.npz")))
    pose = PoseZhou(torch.from_numpy(g["Rt"]))
    with torch.no_grad():
        pose.dR.copy_(torch.from_numpy(g["dR"]))
        pose.dt.copy_(torch.from_numpy(g["dt"]))
    V = pose()
    np.testing.assert_allclose(V.detach().numpy(), g["viewmat"], atol=1e-6)
    (V * torch.from_numpy(g["w"])).sum().backward()
    np.testing.assert_allclose(pose.dR.grad.numpy(), g["grad_dR"], atol=1e-5)
    np.testing.assert_allclose(pose.dt.grad.numpy(), g["grad_dt"], atol=1e-5)
    np.testing.assert_allclose(rotation_6d_to_matrix(torch.from_numpy(g["d6"])).numpy(), g["rot6d"], atol=1e-6)
    fixed = PoseZhou(torch.from_numpy(g["Rt"]), is_learnable=False)
    assert np.array_equal(fixed().numpy(), g["viewmat_fixed"]) and not fixed.dR.requires_grad


def test_utils_match_reference():
    from gslam_amd.utils import StopOnPlateau, create_batch, edge_aware_tv
    g = dict(np.load(os.path.join(GOLD, "utils.npz")))
    d = torch.from_numpy(g["depth"]).requires_grad_(True)
    r = torch.from_numpy(g["rgb"]).requires_grad_(True)
    tv = edge_aware_tv(d, r, torch.from_numpy(g["alphas"])[..., 0] > 0.4)
    tv.backward()
    np.testing.assert_allclose(tv.item(), g["tv"], rtol=1e-5)
    np.testing.assert_allclose(d.grad.numpy(), g["grad_depth"], atol=1e-5)
    np.testing.assert_allclose(r.grad.numpy(), g["grad_rgb"], atol=1e-5)
    sp = StopOnPlateau(3, 0.012)
    assert [sp.stop(float(x)) for x in g["plateau_losses"]] == list(g["plateau_stops"])
    assert np.array_equal(create_batch([torch.arange(3.0), torch.arange(3.0) + 1]).numpy(), g["batch"])


def test_messages_and_output_shape_of_the_api():
    from gslam_amd.messages import BackendMessage, FrontendMessage
    from gslam_amd.rasterization import RasterizationOutput
    assert FrontendMessage.ADD_FRAME == "add_frame" and str(FrontendMessage.REQUEST_INIT) == "request_init"
    assert BackendMessage.SYNC.value == "sync" and BackendMessage.END_SYNC == "end_sync"
    a, d = torch.zeros(1, 4, 4, 1), torch.ones(1, 4, 4)
    out = RasterizationOutput(None, a, d)                               # positional use at gslam/backend.py:619
    assert out.rgbs is None and out.alphas is a and out.depthmaps is d and out.flatten_ids is None
    assert RasterizationOutput._FIELDS[:4] == ('rgbs', 'alphas', 'depthmaps', 'betas')
    assert RasterizationOutput.per_gaussian_params == ('radii', 'means2d')


def test_synthetic_scene_is_deterministic():
    from gslam_amd.synthetic import make_cameras, make_scene
    a, b = make_scene(1000, 3), make_scene(1000, 3)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert not torch.equal(a["means"], make_scene(1000, 4)["means"])
    V, K = make_cameras(3)
    assert V.shape == (3, 4, 4) and K.shape == (3, 3, 3) and float(K[0, 0, 0]) == 525.0
    R = V[2, :3, :3]
    assert torch.allclose(R @ R.T, torch.eye(3), atol=1e-6)


# ---------------------------------------------------------------------------------------------------------------------
# world_size-2 gloo: the data-path collective of keyframe-sharded BA
# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as td
    from gslam_amd import dist as gdist
    r, w = gdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)

    n, cw = 50, 5                                   # 5 keyframes over 2 ranks -> 3 + 2
    shapes = [(n, 3), (n, 4), (n, 3), (n,), (n, 3), (n,)]
    shard = gdist.KeyframeShard()
    window = list(range(cw))
    mine = shard.select(window)
    assert mine == [i for i in window if i % world == rank]
    bucket = gdist.StepBucket(shapes, cw, "cpu")
    offs, numels, s_pad, L = gdist.bucket_layout(shapes, world)
    assert bucket.flat.numel() == s_pad == world * L and L % 4 == 0 and s_pad >= n * 15
    assert bucket.head.numel() == n + cw * 9 + 4 and bucket.world_size == world and bucket.gchunk.numel() == L
    assert [tuple(v.shape) for v in bucket.views] == shapes and all(o % 4 == 0 for o in offs)
    # the chunks partition the six arrays: every element of every array belongs to exactly one rank's pieces
    cover = [torch.zeros(k, dtype=torch.int32) for k in numels]
    for r_ in range(world):
        for k, a, ln, c_off in bucket.pieces(r_):
            cover[k][a:a + ln] += 1
            assert a % 4 == 0 and c_off % 4 == 0 and 0 <= c_off and c_off + ln <= L
    assert all(bool((c == 1).all()) for c in cover)
    # what a rank's launch plan writes: map gradients of its cameras (per-camera mean rule C_local / C of SURVEY 8e folded
    # into the weights), its cameras' pose rows, its share of the loss; rows of other ranks' cameras stay zero
    for k, v in enumerate(bucket.views):
        v.fill_((k + 1) * sum(c + 1 for c in mine) / cw)
    for c in mine:
        bucket.g_dt[c] = float(c + 1)
        bucket.g_dR[c] = float(10 * (c + 1))
    bucket.out2[0] = float(len(mine))
    bucket.out2[1] = 0.5 * len(mine)
    bucket.overflow[0] = 1.0 if rank == 1 else 0.0             # rank 1's tile lists overflowed: both ranks must see it
    local_vis = torch.full((n,), len(mine), dtype=torch.int32)
    seen = []
    bucket.reduce(local_vis, between=lambda: seen.append(int(bucket.vis_i32[0])))
    assert seen == [cw]                                         # the hook runs with the window-wide visibility known
    # reduce-scatter: this rank holds the window-wide sums of ITS chunk
    ok = True
    for k, a, ln, c_off in bucket.pieces():
        expect = (k + 1) * sum(c + 1 for c in window) / cw
        ok = ok and torch.allclose(bucket.gchunk[c_off:c_off + ln], torch.full((ln,), expect))
    ok = ok and torch.equal(bucket.vis_i32, torch.full((n,), cw, dtype=torch.int32))
    ok = ok and torch.equal(bucket.g_dt[:, 0], torch.arange(1, cw + 1, dtype=torch.float32))
    ok = ok and torch.equal(bucket.g_dR[:, 5], 10.0 * torch.arange(1, cw + 1, dtype=torch.float32))
    ok = ok and float(bucket.out2[0]) == cw and float(bucket.out2[1]) == 0.5 * cw and float(bucket.overflow[0]) == 1.0
    # all-gather of the chunks each rank "updated" (its own slice of a flat buffer): everybody ends with all of it
    whole = torch.full((s_pad,), -1.0)
    lo, hi = bucket.chunk_range()
    whole[lo:hi] = torch.arange(lo, hi, dtype=torch.float32)
    bucket.gather(whole, torch.zeros(L))
    ok = ok and torch.equal(whole, torch.arange(s_pad, dtype=torch.float32))
    # a window shorter than the world: the rank without cameras contributes zeros and still joins the collectives
    b2 = gdist.StepBucket(shapes, 1, "cpu")
    if rank == 0:
        for v in b2.views:
            v.fill_(2.0)
        b2.g_dt[0] = 3.0
        b2.reduce(torch.ones(n, dtype=torch.int32))
    else:
        for v in b2.views:
            v.fill_(123.0)                           # stale values of an earlier window must not leak into the sum
        b2.tail.zero_()                              # (the plan's first launch of an iteration clears the tail)
        b2.reduce(None)
    for k, a, ln, c_off in b2.pieces():
        ok = ok and torch.equal(b2.gchunk[c_off:c_off + ln], torch.full((ln,), 2.0))
    ok = ok and torch.equal(b2.vis_i32, torch.ones(n, dtype=torch.int32)) and float(b2.g_dt[0, 0]) == 3.0
    vis = shard.all_reduce_sum(torch.tensor([len(mine)], dtype=torch.int32))
    mx = shard.all_reduce_max(torch.tensor([rank]))
    t = torch.tensor([float(rank)])
    shard.broadcast_([t], src=1)
    q.put((rank, ok, int(vis), int(mx), float(t), bucket.flat.numel()))
    td.barrier()
    td.destroy_process_group()


def test_keyframe_sharded_ba_collectives_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, vis, mx, t, numel in res:
        assert ok and vis == 5 and mx == 1 and t == 1.0 and numel == 760          # six arrays of 50 rows, each padded to 16 bytes, in 2 chunks of 380 floats


def _worker_ranged(rank, world, port, q):
    """the RANGED exchange (StepBucket(ranges = K)) piece by piece, as MappingStep drives it: counts, ranges in order, tail;
    then the all-gather range by range; then the same bucket through the one-shot API (reduce / gather)"""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as td
    from gslam_amd import dist as gdist
    gdist.init_from_env(backend="gloo")
    n, cw, K = 1300, 3, 3
    shapes = [(n, 3), (n, 4), (n, 3), (n,), (n, 3), (n,)]
    b = gdist.StepBucket(shapes, cw, "cpu", ranges=K)
    offs, numels, s_pad, L, nr, k_eff, dims = gdist.ranged_layout(shapes, world, K)
    ok = (b.ranges, b.Nr, b.S_pad, b.L) == (k_eff, nr, s_pad, L) == (3, 512, 1536 * 15, 1536 * 15 // 2)
    ok = ok and b.gchunk.numel() == L and b.vis_all.numel() == 1536 and b.vis_i32.numel() == n
    g = torch.Generator().manual_seed(7)
    local = [torch.randn(s, generator=g) for s in shapes]       # both ranks draw the same numbers ...
    for v, t in zip(b.views, local):
        v.copy_(t * (rank + 1))                                  # ... and scale them by their rank: the sum is 3 x
    b.g_dt[rank] = float(rank + 1)
    b.out2[0] = 1.0
    b.overflow[0] = float(rank)
    b.reduce_counts(torch.full((n,), rank + 1, dtype=torch.int32))
    ok = ok and torch.equal(b.vis_i32, torch.full((n,), 3, dtype=torch.int32)) and int(b.vis_all[n:].abs().sum()) == 0
    ok = ok and float(b.g_dt[1 - rank, 0]) == 0.0               # the counts' all-reduce leaves the tail alone
    for k in range(b.ranges):
        b.reduce_range(k)
    b.reduce_tail()
    ok = ok and float(b.g_dt[0, 0]) == 1.0 and float(b.g_dt[1, 0]) == 2.0 and float(b.out2[0]) == 2.0
    ok = ok and float(b.overflow[0]) == 1.0
    padded = []
    for t, d in zip(local, dims):
        full = torch.zeros(1536 * d)
        full[:n * d] = t.reshape(-1)
        padded.append(full)
    for k_t, a, ln, c_off in b.pieces():
        ok = ok and torch.equal(b.gchunk[c_off:c_off + ln], padded[k_t][a:a + ln] * 1.0 + padded[k_t][a:a + ln] * 2.0)
    # all-gather range by range: every rank "updates" its pieces of a flat buffer
    whole = torch.full((s_pad,), -1.0)
    for k_t, a, ln, c_off in b.pieces():
        o = b.offsets[k_t] + a
        whole[o:o + ln] = torch.arange(o, o + ln, dtype=torch.float32)
    for k in range(b.ranges):
        b.gather_range(whole, k)
    ok = ok and torch.equal(whole, torch.arange(s_pad, dtype=torch.float32))
    # the one-shot API on the ranged layout: same sums
    first = b.gchunk.clone()
    b.gchunk.zero_()
    b.tail.zero_()
    b.g_dt[rank] = float(rank + 1)
    seen = []
    b.reduce(torch.full((n,), rank + 1, dtype=torch.int32), between=lambda: seen.append(int(b.vis_i32[5])))
    ok = ok and seen == [3] and torch.equal(b.gchunk, first)
    whole2 = torch.full((s_pad,), -1.0)
    for k_t, a, ln, c_off in b.pieces():
        o = b.offsets[k_t] + a
        whole2[o:o + ln] = torch.arange(o, o + ln, dtype=torch.float32)
    b.gather(whole2, torch.zeros(1))
    ok = ok and torch.equal(whole2, whole)
    try:
        b.chunk_range()
        ok = False
    except RuntimeError:
        pass
    q.put((rank, ok))
    td.barrier()
    td.destroy_process_group()


def test_ranged_exchange_collectives_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ranged, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_ranged_layout_partitions_the_padded_map_for_every_world_size():
    """gslam_amd.dist.ranged_layout / StepBucket(ranges = K).pieces: ranges of whole projection workgroups, every rank's part of
    every array of every range 16-byte aligned, every element of the PADDED arrays in exactly one rank's pieces, the pieces of a
    range contiguous in the rank's gradient chunk - no process group needed"""
    import torch
    from gslam_amd import dist as gdist
    for n in (1, 300, 1300, 5000, 100001):
        shapes = [(n, 3), (n, 4), (n, 3), (n,), (n, 3), (n,)]
        for world in (2, 3, 4, 6, 8):
            for K in (1, 2, 4, 7):
                offs, numels, s_pad, L, nr, k_eff, dims = gdist.ranged_layout(shapes, world, K)
                assert dims == [3, 4, 3, 1, 3, 1] and nr % 256 == 0 and nr % (4 * world) == 0
                assert 1 <= k_eff <= K and (k_eff - 1) * nr < n <= k_eff * nr and s_pad == 15 * k_eff * nr == world * L
                cover = [torch.zeros(k_eff * nr * d, dtype=torch.int32) for d in dims]
                for r in range(world):
                    b = gdist.StepBucket(shapes, 8, "cpu", world=world, rank=r, ranges=K)
                    assert b.ranges == k_eff and b.flat.numel() == s_pad and b.gchunk.numel() == L
                    assert [tuple(v.shape) for v in b.views] == shapes
                    last_end = 0
                    for t, a, ln, c_off in b.pieces():
                        cover[t][a:a + ln] += 1
                        assert ln > 0 and a % 4 == 0 and c_off == last_end and c_off + ln <= L
                        last_end = c_off + ln
                    assert last_end == L
                    for k in range(k_eff):
                        assert b.pieces_of_range(k) == b.pieces()[6 * k:6 * k + 6]
                assert all(bool((c == 1).all()) for c in cover), (n, world, K)


def test_step_bucket_layout_partitions_the_map_for_every_world_size():
    """the sharded update's layout (gslam_amd.dist.bucket_layout / StepBucket.pieces) for the rank counts the scaling runs use
    and a few awkward ones: equal 16-byte-aligned chunks, every element of every per-Gaussian array in exactly one rank's
    pieces, pieces inside their chunk - no process group needed (the collectives themselves: the gloo test below)"""
    import torch
    from gslam_amd import dist as gdist
    for n in (1, 7, 50, 1001, 4096):
        shapes = [(n, 3), (n, 4), (n, 3), (n,), (n, 3), (n,)]
        for world in (1, 2, 3, 4, 5, 8):
            offs, numels, s_pad, L = gdist.bucket_layout(shapes, world)
            assert numels == [3 * n, 4 * n, 3 * n, n, 3 * n, n]
            assert s_pad == world * L and L % 4 == 0 and all(o % 4 == 0 for o in offs)
            assert s_pad >= offs[-1] + numels[-1] and s_pad - (offs[-1] + (numels[-1] + 3) // 4 * 4) < 4 * world
            cover = [torch.zeros(k, dtype=torch.int32) for k in numels]
            for r in range(world):
                b = gdist.StepBucket(shapes, 8, "cpu", world=world, rank=r)
                assert b.flat.numel() == s_pad and b.chunk_range() == (r * L, (r + 1) * L)
                assert (b.gchunk is None) if world == 1 else (b.gchunk.numel() == L)
                assert b.head.numel() == n + 8 * 9 + 4
                last_end = 0
                for k, a, ln, c_off in b.pieces():
                    cover[k][a:a + ln] += 1
                    assert ln > 0 and a % 4 == 0 and c_off % 4 == 0 and c_off >= last_end and c_off + ln <= L
                    last_end = c_off + ln
                    # the slice of the flat buffer IS the slice of the tensor's view
                    assert b.views[k].reshape(-1)[a:a + ln].data_ptr() == b.flat[r * L + c_off:].data_ptr()
            assert all(bool((c == 1).all()) for c in cover), (n, world)


def test_bench_gpus_n_starts_its_own_ranks_as_a_child(monkeypatch):
    """`python bench.py --gpus N` run plainly (no launcher) must start N ranks under torch.distributed.run itself - as a child
    process, before anything touches the GPU - and hand the child's exit code on."""
    import importlib.util
    import subprocess
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.setattr(torch.cuda, "is_available", lambda: (_ for _ in ()).throw(AssertionError("GPU touched before the launch")))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"

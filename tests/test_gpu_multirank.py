"""Keyframe-sharded bundle adjustment on two ranks: a real BA iteration - render, loss, backward,
all-reduce of the step bucket, Adam - run by two processes that share the one GPU of the test box (gloo between them; the
driver's multi-GPU runs use one rank per GPU over RCCL), against the same iteration on one rank.  Run with -m gpu."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(dev, n_kf, n=8000, W=320, H=240):
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.primitives import Camera, Frame, PoseZhou
    from gslam_amd.synthetic import make_cameras, make_scene
    sc = make_scene(n, 21)
    sc["scales"] = sc["scales"] + 0.5
    splats = GaussianSplattingData.from_dict(sc, dev)
    viewmats, Ks = make_cameras(n_kf, W, H)
    gt = torch.rand(n_kf, H, W, 3, generator=torch.Generator().manual_seed(17)).to(dev)
    window = [Frame(img=gt[i].contiguous(), timestamp=0.0, camera=Camera(Ks[i].to(dev), H, W),
                    pose=PoseZhou(viewmats[i].to(dev), is_learnable=(i != 0)).to(dev), gt_pose=viewmats[i].to(dev), index=i,
                    exposure_params=torch.tensor([0.02 * i, -0.01 * i], device=dev)) for i in range(n_kf)]
    return splats, window


def _worker(rank, world, port, n_kf, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as td
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    from gslam_amd import dist as gdist
    from gslam_amd.mapping import BundleAdjuster
    from gslam_amd.plan import MappingStep
    assert gdist.init_from_env(backend="gloo") == (rank, world)
    # ---- one rank (no shard): the reference of this very process ---------------------------------------------------------
    s1, w1 = _build(dev, n_kf)
    ba1 = BundleAdjuster(s1, capturable=True)
    ref = MappingStep(s1, ba1.optimizers, w1, ba1.conf, shard=None)
    assert ref.world == 1
    ref.render_backward()
    torch.cuda.synchronize()
    ref_flat = ref.flat.clone()
    ref_head = ref.bucket.head.clone()
    ref_vis = ref.r.vis_count.clone()
    ref.step(graphed=False)                                  # same start: render_backward moved nothing
    ref.step(graphed=False)
    torch.cuda.synchronize()
    # ---- two ranks ---------------------------------------------------------------------------------------------------------
    s2, w2 = _build(dev, n_kf)
    ba2 = BundleAdjuster(s2, capturable=True)
    assert ba2.shard.world_size == world
    plan = ba2.plan(w2)
    assert plan.world == world and plan.mine == [i for i in range(n_kf) if i % world == rank]
    b = plan.bucket
    assert b.S_pad == ref.bucket.S_pad and b.offsets == ref.bucket.offsets and b.L * world == b.S_pad
    # the map and both moments live in flat buffers of the bucket's layout now (views keep the tensors' shapes)
    fs = plan.flat_state
    assert s2.means.data_ptr() == fs["pflat"].data_ptr() and s2.quats.data_ptr() == fs["pflat"].data_ptr() + 4 * b.offsets[1]
    plan.render_backward()                                   # includes the collectives: head all-reduce + reduce-scatter
    torch.cuda.synchronize()
    ok = plan.capacity_ok()
    lo, hi = b.chunk_range()
    # this rank's chunk of the window-wide gradient sums against the same slice of the one-rank bucket (the isotropic term
    # is in both: inside the loss launch on one rank, added by rank 0 between the collectives on two)
    scale = float(ref_flat.abs().max())
    d_map = float((b.gchunk - ref_flat[lo:hi]).abs().max()) / scale
    d_cnt = int((plan.vis_i32 - ref_vis).abs().max())
    N_ = s2.means.shape[0]
    head, rhead = b.head.clone(), ref_head
    # tail of the head: pose rows | total, photometric | overflow flag, spare
    d_pose = float((head[N_:-4] - rhead[N_:-4]).abs().max()) / (float(rhead[N_:-4].abs().max()) + 1e-9)
    d_photo = abs(float(head[-3]) - float(rhead[-3])) / abs(float(rhead[-3]))
    ok = ok and float(head[-2]) == 0.0
    # two full iterations (graph | head all-reduce, reduce-scatter | graph: Adam on this rank's chunk | all-gather), then the
    # parameters against the one-rank run
    plan.step()
    plan.step()
    torch.cuda.synchronize()
    ok = ok and plan.capacity_ok() and plan.graph.captured and plan.graph2.captured and fs["sharded"]
    d_par = {k: float((getattr(s1, k) - getattr(s2, k)).abs().mean()) for k in
             ("means", "quats", "scales", "opacities", "colors", "log_uncertainties")}
    d_posepar = max(float((a.pose.dR - b_.pose.dR).abs().max()) for a, b_ in zip(w1, w2))
    n_learn = sum(plan.learnable)
    moved = max([float(f.pose.dR.abs().max()) for f in w2[1:]], default=1.0)
    # every replica holds the same poses: compare rank 0's and rank 1's after the steps
    mine = torch.cat([torch.cat([f.pose.dR, f.pose.dt]) for f in w2]).detach().cpu()
    both = [torch.zeros_like(mine) for _ in range(world)]
    td.all_gather(both, mine)
    same_poses = bool(torch.equal(both[0], both[1]))
    means = fs["pflat"].detach().cpu()                       # the WHOLE flat parameter buffer: all six arrays
    mboth = [torch.zeros_like(means) for _ in range(world)]
    td.all_gather(mboth, means)
    same_map = bool(torch.equal(mboth[0], mboth[1]))
    # the moments are sharded: valid on their owner only, until gather_moments() makes them whole (before a re-pack);
    # then they equal the one-rank run's moments
    m1 = ba1.optimizers.splat_opt.state[s1.means]["exp_avg"]
    own_before = float((fs["mflat"][lo:hi] - torch.cat([ba1.optimizers.splat_opt.state[getattr(s1, k)]["exp_avg"].reshape(-1)
                                                        for k in ("means", "quats", "scales", "opacities", "colors",
                                                                  "log_uncertainties")])[lo:hi]).abs().max())
    ba2.sync_moments()
    m2 = ba2.optimizers.splat_opt.state[s2.means]["exp_avg"]
    d_mom = float((m1 - m2).abs().max()) / (float(m1.abs().max()) + 1e-12)
    why = dict(capacity=ok, sharded_cleared=not fs["sharded"], own_before=own_before, m1max=float(m1.abs().max()), d_mom=d_mom)
    # (bounds: float-atomic noise of the backward, amplified by Adam's normalised first step; a stale or missing moment
    # would be off by its whole magnitude)
    ok = ok and not fs["sharded"] and own_before < 2e-2 * float(m1.abs().max()) + 1e-9 and d_mom < 2e-2
    if not ok:
        print("rank", rank, "not ok:", why, flush=True)
    q.put((rank, ok, d_map, d_cnt, d_pose, d_photo, d_par, d_posepar, moved, same_poses, same_map))
    td.barrier()
    td.destroy_process_group()


@pytest.mark.parametrize("n_kf", [2, 5, 1])
def test_two_rank_ba_step_equals_one_rank(n_kf):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_kf, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, ok, d_map, d_cnt, d_pose, d_photo, d_par, d_posepar, moved, same_poses, same_map in res:
        assert ok, rank
        # the scale columns differ by the isotropic term (added after the reduction when sharded): bounded by its weight
        assert d_map < 5e-3, (rank, d_map)
        assert d_cnt == 0 and d_pose < 2e-3 and d_photo < 1e-5, (rank, d_cnt, d_pose, d_photo)
        for k, v in d_par.items():
            assert v < 2e-5, (rank, k, v)
        assert d_posepar < 2e-3 and moved > 0, (rank, d_posepar, moved)
        assert same_poses and same_map                        # replicas stay bit-identical: no pose broadcast needed


def _ranged_worker(rank, world, port, n_kf, overlap, q):
    """the RANGED, overlapped exchange (MappingStep(exchange_ranges = K)) against the one-shot exchange, both on two
    ranks that render different cameras (so their visible sets differ): (1) the exchange machinery alone on identical local
    gradients - window-wide sums equal BIT FOR BIT; (2) whole iterations - parameters, poses, moments within the float-atomic
    noise of two runs of the same backward, replicas bit-identical"""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as td
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    from gslam_amd import dist as gdist
    from gslam_amd.mapping import BundleAdjuster
    assert gdist.init_from_env(backend="gloo") == (rank, world)
    names = ("means", "quats", "scales", "opacities", "colors", "log_uncertainties")
    sA, wA = _build(dev, n_kf)
    baA = BundleAdjuster(sA, capturable=True)
    A = baA.plan(wA)
    sB, wB = _build(dev, n_kf)
    baB = BundleAdjuster(sB, capturable=True, exchange_ranges=3, exchange_overlap=overlap)
    B = baB.plan(wB)
    bA, bB = A.bucket, B.bucket
    N = sA.means.shape[0]
    ok = B.ranged and B.overlap == overlap and bB.ranges == 3 and bB.Nr == 2816 and not A.ranged
    ok = ok and (B.comm is not None) == overlap and len(B.adam_rest) == 2
    ok = ok and sB.means.data_ptr() == B.flat_state["pflat"].data_ptr()
    ok = ok and sB.quats.data_ptr() == B.flat_state["pflat"].data_ptr() + 4 * bB.offsets[1] and bB.offsets[1] == 3 * 3 * 2816

    def whole_sum(plan):
        """the window-wide gradient sums, every rank's pieces put together: six [N * d] arrays"""
        b = plan.bucket
        w = torch.zeros(b.S_pad, device=dev)
        for t, a, ln, c_off in b.pieces():
            o = b.offsets[t] + a
            w[o:o + ln] = b.gchunk[c_off:c_off + ln]
        b.gather(w, torch.zeros(b.L, device=dev))
        torch.cuda.synchronize()
        return [w[o:o + k].clone() for o, k in zip(b.offsets, b.numels)], w

    # ---- (1) the exchange alone, on the SAME local gradients ---------------------------------------------------------------
    A.render_backward()
    torch.cuda.synchronize()
    sumA, _ = whole_sum(A)
    vis_A = A.vis_i32.clone()
    for vB, vA in zip(bB.views, bA.views):                  # A's local gradients (rank 0's already carry the isotropic term)
        vB.copy_(vA)
    rB, wiso = B.r, B.conf.isotropic_regularization_weight
    B.r, B.conf.isotropic_regularization_weight = None, 0.0   # no projection launches, no second isotropic term
    B._exchange_ranged(torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    B.r, B.conf.isotropic_regularization_weight = rB, wiso
    sumB, wB_flat = whole_sum(B)
    exact = all(torch.equal(x, y) for x, y in zip(sumA, sumB))
    pad_zero = all(float(wB_flat[o + k:o + bB.ranges * bB.Nr * d].abs().sum()) == 0.0
                   for o, k, d in zip(bB.offsets, bB.numels, bB.dims))
    # ---- (2) whole iterations ----------------------------------------------------------------------------------------------
    B.render_backward()
    torch.cuda.synchronize()
    ok = ok and B.capacity_ok() and A.capacity_ok()
    sumB2, _ = whole_sum(B)
    scale = max(float(x.abs().max()) for x in sumA)
    d_grad = max(float((x - y).abs().max()) for x, y in zip(sumA, sumB2)) / scale
    d_cnt = int((B.vis_i32 - vis_A).abs().max())
    differ = int((A.r.vis_count != vis_A).sum()) if A.r is not None else -1      # this rank sees less than the window does
    hA, hB = bA.head.clone(), bB.head.clone()
    d_tail = float((hA[N:] - hB[N:]).abs().max()) / (float(hA[N:].abs().max()) + 1e-9)
    for _ in range(3):
        A.step()
        B.step()
    torch.cuda.synchronize()
    ok = ok and A.capacity_ok() and B.capacity_ok() and B.graph.captured and not B.graph2.captured and B.flat_state["sharded"]
    d_par = {k: float((getattr(sA, k) - getattr(sB, k)).abs().mean()) for k in names}
    d_posepar = max(float((a.pose.dR - b_.pose.dR).abs().max()) for a, b_ in zip(wA, wB))
    steps = (int(baA.optimizers.splat_opt._shared_step.item()), int(baB.optimizers.splat_opt._shared_step.item()))

    def same(t):
        t = t.detach().contiguous().cpu()
        both = [torch.zeros_like(t) for _ in range(world)]
        td.all_gather(both, t)
        return all(torch.equal(both[0], b) for b in both[1:])

    replicas = same(B.flat_state["pflat"]) and same(torch.cat([torch.cat([f.pose.dR, f.pose.dt]) for f in wB]))
    baA.sync_moments()
    baB.sync_moments()
    torch.cuda.synchronize()
    d_mom = 0.0
    for k in names:
        for mom in ("exp_avg", "exp_avg_sq"):
            m1 = baA.optimizers.splat_opt.state[getattr(sA, k)][mom]
            m2 = baB.optimizers.splat_opt.state[getattr(sB, k)][mom]
            d_mom = max(d_mom, float((m1 - m2).abs().max()) / (float(m1.abs().max()) + 1e-12))
    ok = ok and not B.flat_state["sharded"] and same(B.flat_state["mflat"]) and same(B.flat_state["vflat"])
    q.put((rank, ok, exact, pad_zero, d_grad, d_cnt, differ, d_tail, d_par, d_posepar, steps, replicas, d_mom))
    td.barrier()
    td.destroy_process_group()


@pytest.mark.parametrize("n_kf,overlap", [(3, True), (3, False), (1, True)])
def test_ranged_overlapped_exchange_equals_the_one_shot_exchange(n_kf, overlap):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ranged_worker, args=(r, 2, port, n_kf, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, ok, exact, pad_zero, d_grad, d_cnt, differ, d_tail, d_par, d_posepar, steps, replicas, d_mom in res:
        assert ok, rank
        assert exact and pad_zero, (rank, exact, pad_zero)      # same local gradients in -> the same sums out, bit for bit
        assert d_grad < 2e-4 and d_cnt == 0 and d_tail < 2e-3, (rank, d_grad, d_cnt, d_tail)
        if n_kf == 3:
            assert differ > 0, rank                              # the rank's own visible set is NOT the window's
        for k, v in d_par.items():
            assert v < 2e-5, (rank, k, v)
        assert d_posepar < 2e-3 and steps == (3, 3) and replicas and d_mom < 2e-2, (rank, d_posepar, steps, replicas, d_mom)


def _overflow_worker(rank, world, port, q):
    """rank 1's tile lists are sized with no head-room; the map then grows fatter on both replicas: rank 1's render overflows
    inside the replayed graph, rank 0's does not.  The flag travels in the iteration's all-reduce, the update launches are
    gated on it on the device: NOTHING moves on either rank, both read ok = False, rank 1 re-captures, both redo."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as td
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    from gslam_amd import dist as gdist
    from gslam_amd.mapping import BundleAdjuster
    assert gdist.init_from_env(backend="gloo") == (rank, world)
    s2, w2 = _build(dev, 4)
    ba2 = BundleAdjuster(s2, capturable=True)
    plan = ba2.plan(w2)
    plan.r.GROW = 1.0 if rank == 1 else 3.0                   # instance attribute: head-room of the capacity probe
    plan.step()
    total0, pm0, ok0 = plan.finish_step()
    names = ("means", "quats", "scales", "opacities", "colors", "log_uncertainties")
    with torch.no_grad():
        s2.scales.add_(0.25)                                  # every splat 28 % wider: M grows by more than rank 1's slack
    torch.cuda.synchronize()
    before = {k: getattr(s2, k).detach().clone() for k in names}
    poses_before = torch.cat([torch.cat([f.pose.dR, f.pose.dt]) for f in w2]).detach().clone()
    m_before = ba2.optimizers.splat_opt.state[s2.means]["exp_avg"].clone()
    ctr_before = int(ba2.optimizers.splat_opt._shared_step.item())
    cap_before = plan.r.capacity
    plan.step()
    total1, pm1, ok1 = plan.finish_step()                     # overflowed on rank 1 -> False on BOTH ranks
    untouched = all(torch.equal(before[k], getattr(s2, k).detach()) for k in names)
    untouched = untouched and torch.equal(poses_before, torch.cat([torch.cat([f.pose.dR, f.pose.dt]) for f in w2]).detach())
    untouched = untouched and torch.equal(m_before, ba2.optimizers.splat_opt.state[s2.means]["exp_avg"])
    untouched = untouched and int(ba2.optimizers.splat_opt._shared_step.item()) == ctr_before
    grew = plan.r.capacity > cap_before
    for _ in range(3):
        plan.step()                                           # rank 1 re-captures with the grown lists; both redo
        total2, pm2, ok2 = plan.finish_step()
        if ok2:
            break
    moved = not torch.equal(before["means"], s2.means.detach())
    stepped = int(ba2.optimizers.splat_opt._shared_step.item()) == ctr_before + 1

    def same(t):
        t = t.detach().contiguous().cpu()
        both = [torch.zeros_like(t) for _ in range(world)]
        td.all_gather(both, t)
        return all(torch.equal(both[0], b) for b in both[1:])

    replicas = all(same(getattr(s2, k)) for k in names) and same(torch.stack([f.pose().detach() for f in w2]))
    q.put((rank, ok0, ok1, untouched, grew, ok2, moved, stepped, replicas, pm1, pm2))
    td.barrier()
    td.destroy_process_group()


def test_two_rank_overflow_is_collective_and_applies_no_update():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overflow_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, ok0, ok1, untouched, grew, ok2, moved, stepped, replicas, pm1, pm2 in res:
        assert ok0, rank                                      # the first iteration fitted on both ranks
        assert not ok1, rank                                  # the overflow of rank 1 is seen by BOTH ranks
        assert untouched, rank                                # ... and nothing moved on either: map, poses, moments, counters
        assert grew == (rank == 1), (rank, grew)              # only the rank that overflowed grows its lists
        assert ok2 and moved and stepped and replicas, (rank, ok2, moved, stepped, replicas)
    assert res[0][-1] == res[1][-1]                           # same loss value read on both ranks


def _backend_worker(rank, world, port, q):
    """one Backend replica per rank through the reference's message loop (REQUEST_INIT, ADD_FRAME -> keyframes, BA with a
    densification every 7th iteration, pruning, window pose refinement): everything that must stay identical between the
    replicas is gathered at the end"""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import queue

    import torch.distributed as td
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    from gslam_amd import dist as gdist
    from gslam_amd.backend import Backend, MapConfig
    from gslam_amd.map import GaussianSplattingData
    from gslam_amd.messages import FrontendMessage
    from gslam_amd.primitives import Camera, Frame, PoseZhou
    from gslam_amd.synthetic import make_intrinsics, make_scene, make_viewmat
    assert gdist.init_from_env(backend="gloo") == (rank, world)
    W, H = 320, 240
    cam = Camera(make_intrinsics(W, H).to(dev), H, W)
    sc = make_scene(30000, 3)
    sc["scales"] = sc["scales"] + 0.6
    scene = GaussianSplattingData.from_dict(sc, dev)

    def sensor_frame(i):
        V = make_viewmat(2.0 * i).to(dev)
        with torch.no_grad():
            img = scene([cam], [PoseZhou(V, is_learnable=False).to(dev)], render_depth=False).rgbs[0].clamp(0, 1)
        # the tracked pose the frontend would ship: the true one, off by a millimetre
        Vt = V.clone()
        Vt[:3, 3] += 1e-3
        return Frame(img=img.contiguous(), timestamp=i / 30.0, camera=cam, pose=PoseZhou(Vt).to(dev), gt_pose=V, index=i,
                     exposure_params=torch.zeros(2, device=dev))

    conf = MapConfig(num_iters_initialization=20, num_iters_mapping=5, kf_m=0.02, densify_every=7, seed=5)
    be = Backend(conf, queue.Queue(), queue.Queue())
    assert be.handle((FrontendMessage.REQUEST_INIT, sensor_frame(0)))
    assert be.ba.shard.world_size == world
    n_hist = [be.splats.means.shape[0]]
    for i in range(1, 6):
        assert be.handle((FrontendMessage.ADD_FRAME, sensor_frame(i)))
        be.idle_step()
        n_hist.append(be.splats.means.shape[0])
    torch.cuda.synchronize()

    def same(t):
        t = t.detach().contiguous().cpu()
        both = [torch.zeros_like(t) for _ in range(world)]
        td.all_gather(both, t)
        return all(torch.equal(both[0], b) for b in both[1:])

    sizes = torch.tensor(n_hist + [len(be.keyframes), be.total_step])
    ok_sizes = same(sizes)                                    # before any tensor gather: shapes must agree first
    res = dict(sizes=ok_sizes, n=n_hist, n_kf=len(be.keyframes), steps=be.total_step)
    if ok_sizes:
        for k in ("means", "quats", "scales", "opacities", "colors", "log_uncertainties", "ages"):
            res[k] = same(getattr(be.splats, k))
        res["adam"] = same(be.ba.optimizers.splat_opt.state[be.splats.means]["exp_avg"])
        res["poses"] = same(torch.stack([f.pose().detach() for f in be.keyframes.values()]))
        res["depths"] = same(torch.stack([f.est_depths for f in be.keyframes.values() if f.est_depths is not None
                                          and f.est_depths.dim() == 2][:2]))
        res["finite"] = bool(torch.isfinite(be.splats.means).all())
    q.put((rank, res))
    td.barrier()
    td.destroy_process_group()


def test_two_backend_replicas_keep_identical_maps():
    """keyframe-sharded mapping inside the whole backend loop - depth-map insertion and densification draw the
    same random numbers, the pruning statistics are reduced over ranks, the refined window poses are handed back from rank
    0 - so two replicas fed the same frames hold bit-identical maps, Adam state and keyframe poses afterwards."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_backend_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, r in res:
        assert r["sizes"], (rank, r)
        assert r["n_kf"] >= 3 and r["steps"] >= 40, (rank, r)
        assert len(set(r["n"])) > 2, (rank, r)                # the map grew / shrank along the way
        for k, v in r.items():
            if isinstance(v, bool):
                assert v, (rank, k, r)

"""world_size-2 gloo test of the landmark-sharded exchange (CPU only).

The product's shard rule (cugo_shard_range, host logic of libcugo_hip.so) splits the landmarks;
each rank builds the partial Schur system of ITS landmarks with the CPU oracle, the ranks
all-reduce [Hsc | bsc | chi2] over gloo exactly as the GPU path all-reduces its device buffer,
and the sum must equal the unsharded system (lambda enters once, after the reduction).  The
product's own sharded host path runs too (plan-only graphs: flattening of the shard, global Hsc
pattern, per-shard product lists), its counts all-reduced over gloo."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    for p in (HERE, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import oracle
    cugo = importlib.import_module("cuda-bundle-adjustment_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = cugo.synth(40, 500, 2100, seed=4)
    full = oracle.Problem(d["pose"], d["pose_fixed"], d["lm"], d["lm_fixed"], d["e_pose"], d["e_lm"],
                          d["e_stereo"], d["e_meas"], d["e_omega"], d["e_cam"])
    # landmark indices == ids here (no fixed landmarks): edges per landmark in index order
    per_lm = np.bincount(d["e_lm"], minlength=len(d["lm"])).astype(np.int32)
    l0, l1 = cugo.shard_range(per_lm, rank, world)
    sel = (d["e_lm"] >= l0) & (d["e_lm"] < l1)
    part = oracle.Problem(d["pose"], d["pose_fixed"], d["lm"], d["lm_fixed"], d["e_pose"][sel], d["e_lm"][sel],
                          d["e_stereo"][sel], d["e_meas"][sel], d["e_omega"][sel], d["e_cam"][sel])
    lam = 2.5
    H, b = part.schur_dense(lam)
    if rank > 0:
        H = H - lam * np.eye(H.shape[0])  # lambda is added once, after the reduction
    chi = part.compute_errors()
    payload = torch.from_numpy(np.concatenate([H.ravel(), b, [chi, float(sel.sum())]]))
    dist.all_reduce(payload, op=dist.ReduceOp.SUM)
    got = payload.numpy()
    n = H.shape[0]
    Href, bref = full.schur_dense(lam)
    ok_H = np.allclose(got[:n * n].reshape(n, n), Href, rtol=0, atol=1e-11 * np.abs(Href).max())
    ok_b = np.allclose(got[n * n:n * n + n], bref, rtol=0, atol=1e-11 * np.abs(bref).max())
    ok_chi = abs(got[-2] - full.compute_errors()) <= 1e-12 * got[-2]
    ok_cover = int(round(got[-1])) == full.n_edges
    # max-reduction used for the first lambda
    mx = torch.tensor([float(np.abs(np.diag(part.schur_dense(0.0)[0])).max())], dtype=torch.float64)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    # the PRODUCT's own sharded host path (plan-only graph, no GPU): every rank flattens its shard and
    # builds the structure; the Hsc pattern must be the global one on every rank (same all-reduce
    # payload layout) and the ranks' block-product lists must partition the unsharded list
    g = cugo.graph_from_arrays(d, plan_only=True)
    g.set_shard(rank, world, lambda ptr, n, op: None)
    g.initialize()
    ss = g.structure_stats()
    g.close()
    gu = cugo.graph_from_arrays(d, plan_only=True)
    gu.initialize()
    su = gu.structure_stats()
    gu.close()
    cnt = torch.tensor([ss["offdiag_products"], ss["hsc_blocks"]], dtype=torch.float64)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    ok_plan = int(cnt[0].item()) == int(su["offdiag_products"]) and int(cnt[1].item()) == world * int(su["hsc_blocks"])
    with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as f:
        f.write("%d %d %d %d %d %d %d\n" % (ok_H, ok_b, ok_chi, ok_cover, l0, l1, ok_plan))
    dist.destroy_process_group()


def test_two_rank_gloo_exchange_matches_unsharded(tmp_path):
    import torch.multiprocessing as mp
    importlib.import_module("cuda-bundle-adjustment_amd").build()
    import oracle
    oracle.build()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    ranges = []
    for r in range(2):
        vals = [int(v) for v in open(tmp_path / ("rank%d.txt" % r)).read().split()]
        assert vals[:4] == [1, 1, 1, 1] and vals[6] == 1, vals
        ranges.append(vals[4:6])
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == 500


def _agree_worker(rank, world, port, out_dir):
    for p in (HERE, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import threading
    import time
    import torch.distributed as dist
    cugo = importlib.import_module("cuda-bundle-adjustment_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []
    log = []

    def ok_comm():
        calls.append("ok")
        return "comm-of-rank-%d" % rank

    def failing():
        calls.append("fail")
        if rank == 1:
            raise RuntimeError("induced")
        return "comm-of-rank-%d" % rank

    def hanging():
        calls.append("hang")
        if rank == 1:
            threading.Event().wait()   # never comes back (a daemon thread: the process can still end)
        return "comm-of-rank-%d" % rank
    res = []
    res.append(cugo.create_comm_agreed(dist, rank, world, ok_comm, can_try=True, deadline_s=20, log=log.append))
    n_before = len(calls)
    res.append(cugo.create_comm_agreed(dist, rank, world, ok_comm, can_try=(rank == 0), deadline_s=20, log=log.append))
    not_entered = len(calls) == n_before          # nobody may enter the collective when a rank cannot try
    res.append(cugo.create_comm_agreed(dist, rank, world, failing, can_try=True, deadline_s=20, log=log.append))
    t0 = time.perf_counter()
    res.append(cugo.create_comm_agreed(dist, rank, world, hanging, can_try=True, deadline_s=2.0, log=log.append))
    waited = time.perf_counter() - t0
    with open(os.path.join(out_dir, "agree%d.txt" % rank), "w") as f:
        f.write("%s|%s|%s|%s|%d|%.2f|%d\n" % (res[0][0], res[1][0], res[2][0], res[3][0], not_entered, waited,
                                              sum("rank 1" in m for m in log)))
        f.write(repr(res[0][1]) + "\n")
    dist.destroy_process_group()


def test_communicator_creation_is_agreed_on_and_never_hangs(tmp_path):
    """cugo.create_comm_agreed (what bench.py --gpus N puts round cugo_comm_create), world size 2 over gloo on the CPU
    with stand-ins for the RCCL call: every rank gets a communicator -> "native" on both; one rank cannot even try ->
    "fallback" on both and NOBODY enters the collective; one rank's init raises -> "failed" on both; one rank's init
    never returns -> "failed" on both once the deadline has passed (the other rank is not left waiting)."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_agree_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        lines = open(tmp_path / ("agree%d.txt" % r)).read().splitlines()
        a, b, c, d, not_entered, waited, rank1_msgs = lines[0].split("|")
        assert (a, b, c, d) == ("native", "fallback", "failed", "failed"), lines
        assert int(not_entered) == 1
        assert 1.5 <= float(waited) < 15.0, waited
        assert lines[1] == repr("comm-of-rank-%d" % r)
        assert int(rank1_msgs) == (2 if r == 1 else 0)   # the failing rank says why, the other one has nothing to say

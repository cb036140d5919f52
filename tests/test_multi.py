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

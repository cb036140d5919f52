"""Test helpers: device arrays through the C ABI and a numpy flattening of a flat-array
problem into the cugo_edges layout documented in include/cugo_hip.h."""
import ctypes as C
import importlib

import numpy as np

cugo = importlib.import_module("cuda-bundle-adjustment_amd")


class Ctx:
    def __init__(self):
        self.h = C.c_void_p()
        cugo.check(cugo.lib().cugo_ctx_create(-1, C.byref(self.h)))
        self.allocs = []

    def to_dev(self, a):
        a = np.ascontiguousarray(a)
        p = C.c_void_p()
        cugo.check(cugo.lib().cugo_malloc(C.byref(p), max(a.nbytes, 16)))
        self.allocs.append(p)
        if a.nbytes:
            cugo.check(cugo.lib().cugo_memcpy_h2d(self.h, p, a.ctypes.data_as(C.c_void_p), a.nbytes))
        return p

    def empty(self, n, dtype=np.float64):
        return self.to_dev(np.zeros(max(int(n), 1), dtype))

    def to_host(self, p, shape, dtype=np.float64):
        out = np.zeros(shape, dtype)
        if out.nbytes:
            cugo.check(cugo.lib().cugo_memcpy_d2h(self.h, out.ctypes.data_as(C.c_void_p), p, out.nbytes))
        return out

    def sync(self):
        cugo.check(cugo.lib().cugo_ctx_sync(self.h))

    def close(self):
        self.sync()
        for p in self.allocs:
            cugo.lib().cugo_free(p)
        self.allocs = []
        cugo.lib().cugo_ctx_destroy(self.h)


def flatten(prob):
    """numpy restatement of the documented cugo_edges layout (landmark-major, free-first
    indices). prob: tests/oracle.Problem. Returns host arrays + index maps."""
    pidx, lidx, P, L = prob.indices()
    active = ~((prob.pose_fixed[prob.e_pose] != 0) & (prob.lm_fixed[prob.e_lm] != 0))
    src = np.flatnonzero(active)
    ip, il = pidx[prob.e_pose[src]], lidx[prob.e_lm[src]]
    order = np.lexsort((ip, il))
    src, ip, il = src[order], ip[order].astype(np.int32), il[order].astype(np.int32)
    E = len(src)
    flags = ((prob.lm_fixed[prob.e_lm[src]] != 0) * 1 + (prob.pose_fixed[prob.e_pose[src]] != 0) * 2 +
             (prob.e_stereo[src] != 0) * 4).astype(np.uint8)
    meas = np.ascontiguousarray(prob.e_meas[src].T)  # planar [3][E]
    cams, cam_id = np.unique(prob.e_cam[src], axis=0, return_inverse=True)
    Pall, Lall = prob.n_poses, prob.n_landmarks
    lm_ptr = np.zeros(Lall + 1, np.int32)
    np.add.at(lm_ptr, il + 1, 1)
    lm_ptr = np.cumsum(lm_ptr).astype(np.int32)
    pose_edge = np.lexsort((il, ip)).astype(np.int32)
    pose_ptr = np.zeros(Pall + 1, np.int32)
    np.add.at(pose_ptr, ip + 1, 1)
    pose_ptr = np.cumsum(pose_ptr).astype(np.int32)
    poses = np.zeros((Pall, 7)); poses[pidx] = prob.pose
    lms = np.zeros((Lall, 3)); lms[lidx] = prob.lm
    return dict(E=E, P=P, L=L, Pall=Pall, Lall=Lall, src=src, pose=ip, lm=il, flags=flags, meas=meas,
                omega=np.ascontiguousarray(prob.e_omega[src]), cams=np.ascontiguousarray(cams),
                cam_id=cam_id.astype(np.uint16).reshape(-1), lm_ptr=lm_ptr, pose_ptr=pose_ptr,
                pose_edge=pose_edge, poses=poses, lms=lms, pidx=pidx, lidx=lidx)


def upload_edges(ctx, f):
    ev = cugo.Edges()
    ev.n_edges, ev.n_poses_total, ev.n_landmarks_total = f["E"], f["Pall"], f["Lall"]
    ev.n_poses_free, ev.n_landmarks_free = f["P"], f["L"]
    ev.d_pose, ev.d_lm = ctx.to_dev(f["pose"]), ctx.to_dev(f["lm"])
    ev.d_meas, ev.d_omega, ev.n_omega = ctx.to_dev(f["meas"]), ctx.to_dev(f["omega"]), f["E"]
    ev.d_flags = ctx.to_dev(f["flags"])
    ev.n_cams = len(f["cams"])
    ev.d_cams = ctx.to_dev(f["cams"])
    ev.d_cam = ctx.to_dev(f["cam_id"]) if ev.n_cams > 1 else None
    ev.d_lm_ptr, ev.d_pose_ptr = ctx.to_dev(f["lm_ptr"]), ctx.to_dev(f["pose_ptr"])
    ev.d_pose_edge = ctx.to_dev(f["pose_edge"])
    return ev


def hsc_structure(f):
    """upper block CSR pattern + off-diagonal contribution lists (numpy restatement of
    ref src/sparse_block_matrix.cpp:63-156 and findHschureMulBlockIndices)."""
    P = f["P"]
    ff = (f["flags"] & 11) == 0  # free-free and active
    rows = [set([p]) for p in range(P)]
    per_lm = []
    for l in range(f["L"]):
        es = [e for e in range(f["lm_ptr"][l], f["lm_ptr"][l + 1]) if ff[e]]
        per_lm.append(es)
        ps = [int(f["pose"][e]) for e in es]
        for i, p in enumerate(ps):
            rows[p].update(ps[i:])
    rowptr, colind = [0], []
    for p in range(P):
        cols = sorted(rows[p])
        assert cols[0] == p
        colind += cols
        rowptr.append(len(colind))
    rowptr, colind = np.array(rowptr, np.int32), np.array(colind, np.int32)
    B = len(colind)
    lists = [[] for _ in range(B)]
    for es in per_lm:
        for a in range(len(es)):
            pa = int(f["pose"][es[a]])
            for b in range(a + 1, len(es)):
                pb = int(f["pose"][es[b]])
                k = rowptr[pa] + int(np.searchsorted(colind[rowptr[pa]:rowptr[pa + 1]], pb))
                lists[k].append((es[a], es[b]))
    off_ptr = np.zeros(B + 1, np.int32)
    off_ptr[1:] = np.cumsum([len(x) for x in lists])
    ei = np.array([a for x in lists for a, _ in x], np.int32)
    ej = np.array([b for x in lists for _, b in x], np.int32)
    return rowptr, colind, off_ptr, ei, ej


def pad_to_groups(f, group=256):
    """the engine's slot layout (csrc/host/engine.cpp): INACTIVE padding slots are inserted so that
    no landmark with <= `group` edges straddles a `group`-slot boundary; a padding slot belongs to
    the landmark before it.  Returns a new flattened dict (same keys as flatten())."""
    E0 = f["E"]
    src_of = []  # per new slot: old slot or -1
    lm_of = []
    pos = 0
    last_lm = -1
    for l in range(f["Lall"]):
        a, b = int(f["lm_ptr"][l]), int(f["lm_ptr"][l + 1])
        k = b - a
        if k > 0 and k <= group and pos % group + k > group and last_lm >= 0:
            npad = group - pos % group
            src_of += [-1] * npad
            lm_of += [last_lm] * npad
            pos += npad
        src_of += list(range(a, b))
        lm_of += [l] * k
        pos += k
        if k > 0:
            last_lm = l
    src_of = np.array(src_of, np.int64)
    real = src_of >= 0
    E = len(src_of)
    g = dict(f)
    g["E"] = E
    take = np.where(real, src_of, 0)
    g["pose"] = np.where(real, f["pose"][take], 0).astype(np.int32)
    g["lm"] = np.array(lm_of, np.int32)
    g["flags"] = np.where(real, f["flags"][take], 8).astype(np.uint8)
    g["meas"] = np.ascontiguousarray(np.where(real[None, :], f["meas"][:, take], 0.0))
    g["omega"] = np.ascontiguousarray(np.where(real, f["omega"][take], 0.0))
    g["cam_id"] = np.where(real, f["cam_id"][take], 0).astype(np.uint16)
    g["src"] = np.where(real, f["src"][take], -1)
    lm_ptr = np.zeros(f["Lall"] + 1, np.int32)
    np.add.at(lm_ptr, g["lm"] + 1, 1)
    g["lm_ptr"] = np.cumsum(lm_ptr).astype(np.int32)
    slots = np.flatnonzero(real)
    order = np.lexsort((g["lm"][slots], g["pose"][slots]))
    g["pose_edge"] = np.concatenate([slots[order], np.zeros(E - len(slots), np.int64)]).astype(np.int32)
    pose_ptr = np.zeros(f["Pall"] + 1, np.int32)
    np.add.at(pose_ptr, g["pose"][slots] + 1, 1)
    g["pose_ptr"] = np.cumsum(pose_ptr).astype(np.int32)
    assert E >= E0 and int(real.sum()) == E0
    return g

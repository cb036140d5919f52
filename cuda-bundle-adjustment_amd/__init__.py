"""ctypes binding of libcugo_hip.so — the MI355X bundle-adjustment hot path.

The directory name contains a hyphen, so import it with
    importlib.import_module("cuda-bundle-adjustment_amd")
(tests/conftest.py and bench.py do).  This module is plumbing only: all compute lives in the
HIP library; if the library is missing every entry point raises (no CPU fallback).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CUGO_LIB selects another build of the same library (tools/run_san.sh: the sanitizer build; the diagnosis tools:
# the hooks build)
LIB_PATH = os.environ.get("CUGO_LIB") or os.path.join(_HERE, "libcugo_hip.so")
HOOKS_LIB_PATH = os.path.join(_HERE, "libcugo_hip_hooks.so")  # make HOOKS=1: delay patterns, checksums, fault injection
_lib = None

OK = 0
RK_NONE, RK_CAUCHY, RK_TUKEY = 0, 1, 2
EDGE_FIXED_L, EDGE_FIXED_P, EDGE_STEREO, EDGE_INACTIVE = 1, 2, 4, 8

_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_u16p = C.POINTER(C.c_uint16)

EXCHANGE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)


class CugoError(RuntimeError):
    pass


class Edges(C.Structure):
    _fields_ = [("n_edges", C.c_int), ("n_poses_total", C.c_int), ("n_landmarks_total", C.c_int),
                ("n_poses_free", C.c_int), ("n_landmarks_free", C.c_int),
                ("d_pose", C.c_void_p), ("d_lm", C.c_void_p), ("d_meas", C.c_void_p),
                ("d_omega", C.c_void_p), ("n_omega", C.c_int), ("d_flags", C.c_void_p),
                ("d_cam", C.c_void_p), ("d_cams", C.c_void_p), ("n_cams", C.c_int),
                ("d_lm_ptr", C.c_void_p), ("d_pose_ptr", C.c_void_p), ("d_pose_edge", C.c_void_p),
                ("block_f32", C.c_int)]


class Robust(C.Structure):
    _fields_ = [("type", C.c_int), ("delta", C.c_double), ("type_stereo", C.c_int),
                ("delta_stereo", C.c_double)]


class HscStruct(C.Structure):
    _fields_ = [("n_blocks", C.c_int), ("d_rowptr", C.c_void_p), ("d_colind", C.c_void_p),
                ("d_off_ptr", C.c_void_p), ("d_off_ei", C.c_void_p), ("d_off_ej", C.c_void_p),
                # landmark-major product plan (cugo_hsc_plan_create fills these; NULL = gather kernels)
                ("n_groups", C.c_int), ("n_slots", C.c_int), ("n_rhs", C.c_int),
                ("d_grp_ptr", C.c_void_p), ("d_grp_nwave", C.c_void_p), ("d_slot_rhs", C.c_void_p),
                ("d_slot_ptr", C.c_void_p),
                ("d_prod", C.c_void_p), ("d_red_ptr", C.c_void_p), ("d_red_slot", C.c_void_p),
                ("d_blk_pose", C.c_void_p), ("d_part_H", C.c_void_p), ("d_part_b", C.c_void_p)]


class SynthParams(C.Structure):
    _fields_ = [("n_poses", C.c_int), ("n_landmarks", C.c_int), ("n_edges", C.c_int),
                ("stereo_fraction", C.c_double), ("pixel_noise", C.c_double),
                ("pose_rot_noise", C.c_double), ("pose_trans_noise", C.c_double),
                ("landmark_noise_rel", C.c_double), ("n_loop_closures", C.c_int),
                ("seed", C.c_uint64)]


def build(force=False):
    """compile libcugo_hip.so in-tree with hipcc for gfx950 (no GPU needed to build)"""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-j8", "-s"])
    else:
        subprocess.check_call(["make", "-C", _HERE, "-j8", "-s"])  # make decides what is stale
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CugoError("libcugo_hip.so is not built (run __graft_entry__.build()); "
                            "there is no CPU fallback")
        _lib = C.CDLL(LIB_PATH)
        _lib.cugo_last_error.restype = C.c_char_p
        _lib.cugo_ctx_stream.restype = C.c_void_p
    return _lib


def check(rc):
    if rc != OK:
        raise CugoError("cugo error %d: %s" % (rc, lib().cugo_last_error().decode()))


def device_count():
    return lib().cugo_device_count()


UNIQUE_ID_BYTES = 128


def set_device(device):
    check(lib().cugo_set_device(int(device)))


class Comm:
    """cugo_comm_*: RCCL communicator over the ranks of a multi-GPU job (collective constructor)"""

    def __init__(self, unique_id, rank, world):
        self._c = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), UNIQUE_ID_BYTES)
        check(lib().cugo_comm_create(buf, int(rank), int(world), C.byref(self._c)))

    def close(self):
        if self._c:
            lib().cugo_comm_destroy(self._c)
            self._c = C.c_void_p()


def create_comm_agreed(dist, rank, world, make_comm, can_try=True, deadline_s=180.0, log=None):
    """Collective creation of the RCCL communicator of a multi-process job WITHOUT the two ways such a step hangs a
    job: a rank that cannot even try (no librccl, no device of its own) while the others block inside
    ncclCommInitRank waiting for it, and a rank whose init never returns.  `dist`: an initialised torch.distributed
    group used as the rendezvous (gloo); `make_comm()`: creates this rank's communicator (cugo.Comm(...)), called in
    a helper thread.  Every rank first reports — without entering a collective — whether it can try (`can_try`); only
    if ALL can, they create their communicators, each with a deadline, and then agree on the outcome.

    Returns ("native", comm) when every rank has a communicator, ("fallback", None) when some rank could not try
    (nobody entered the collective: the job may go on with the host-staged exchange), ("failed", None) when some
    rank's init raised or missed the deadline — the caller must then END THE PROCESS on every rank (os._exit: a
    thread stuck inside RCCL cannot be joined, and the ranks that did get a communicator must not wait for the
    others in the next collective)."""
    import threading
    import torch
    pre = torch.tensor([1.0 if can_try else 0.0])
    dist.all_reduce(pre, op=dist.ReduceOp.MIN)
    if float(pre.item()) < 0.5:
        return "fallback", None
    result = {}

    def run():
        try:
            result["comm"] = make_comm()
        except BaseException as e:  # noqa: BLE001 (reported below; the agreement decides what happens)
            result["error"] = e
    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(deadline_s)
    mine_ok = "comm" in result and not th.is_alive()
    if not mine_ok and log is not None:
        log("rank %d: communicator %s" % (rank, "failed: %s" % result["error"] if "error" in result
                                          else "did not come back within %.0f s" % deadline_s))
    ok = torch.tensor([1.0 if mine_ok else 0.0])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if float(ok.item()) < 0.5:
        return "failed", None
    return "native", result["comm"]


def comm_unique_id():
    """rank 0 of a multi-GPU job: the RCCL unique id to hand to every rank (bytes)"""
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    check(lib().cugo_comm_unique_id(buf))
    return buf.raw


def _p(a, t):
    return a.ctypes.data_as(t)


def synth(n_poses, n_landmarks, n_edges, seed, stereo_fraction=0.7, pixel_noise=1.0,
          pose_rot_noise=0.005, pose_trans_noise=0.05, landmark_noise_rel=0.01, n_loop_closures=0):
    """seeded synthetic graph (host only). Returns dict of numpy arrays (ids == positions)."""
    prm = SynthParams(n_poses, n_landmarks, n_edges, stereo_fraction, pixel_noise, pose_rot_noise,
                      pose_trans_noise, landmark_noise_rel, n_loop_closures, seed)
    d = dict(pose=np.zeros((n_poses, 7)), lm=np.zeros((n_landmarks, 3)),
             e_pose=np.zeros(n_edges, np.int32), e_lm=np.zeros(n_edges, np.int32),
             e_stereo=np.zeros(n_edges, np.uint8), e_meas=np.zeros((n_edges, 3)),
             e_omega=np.zeros(n_edges), cam=np.zeros(5))
    check(lib().cugo_synth_generate(C.byref(prm), _p(d["pose"], _f64p), _p(d["lm"], _f64p),
                                    _p(d["e_pose"], _i32p), _p(d["e_lm"], _i32p),
                                    _p(d["e_stereo"], _u8p), _p(d["e_meas"], _f64p),
                                    _p(d["e_omega"], _f64p), _p(d["cam"], _f64p)))
    d["pose_fixed"] = np.zeros(n_poses, np.uint8)
    d["pose_fixed"][0] = 1
    d["lm_fixed"] = np.zeros(n_landmarks, np.uint8)
    d["e_cam"] = np.tile(d["cam"], (n_edges, 1))
    return d


def shard_range(edges_per_landmark, rank, world):
    e = np.ascontiguousarray(edges_per_landmark, np.int32)
    a, b = C.c_int(), C.c_int()
    check(lib().cugo_shard_range(len(e), _p(e, _i32p), rank, world, C.byref(a), C.byref(b)))
    return a.value, b.value


class Graph:
    """cugo_graph_*: the reference's public optimiser class behind a flat-array interface
    (ref: include/cuda_graph_optimisation.h:132-252)."""

    def __init__(self, per_edge_information=True, per_edge_camera=True, plan_only=False):
        self._g = C.c_void_p()
        create = lib().cugo_graph_create_plan_only if plan_only else lib().cugo_graph_create
        check(create(int(per_edge_information), int(per_edge_camera), C.byref(self._g)))
        self._cb = None
        self.pose_ids = np.zeros(0, np.int32)
        self.lm_ids = np.zeros(0, np.int32)

    def close(self):
        if self._g:
            lib().cugo_graph_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_poses(self, ids, q_t7, fixed):
        ids = np.ascontiguousarray(ids, np.int32); q = np.ascontiguousarray(q_t7, np.float64)
        f = np.ascontiguousarray(fixed, np.uint8)
        check(lib().cugo_graph_add_poses(self._g, len(ids), _p(ids, _i32p), _p(q, _f64p), _p(f, _u8p)))
        self.pose_ids = np.concatenate([self.pose_ids, ids])

    def add_landmarks(self, ids, xyz, fixed):
        ids = np.ascontiguousarray(ids, np.int32); x = np.ascontiguousarray(xyz, np.float64)
        f = np.ascontiguousarray(fixed, np.uint8)
        check(lib().cugo_graph_add_landmarks(self._g, len(ids), _p(ids, _i32p), _p(x, _f64p), _p(f, _u8p)))
        self.lm_ids = np.concatenate([self.lm_ids, ids])

    def add_edges(self, dim, pose_ids, lm_ids, meas, info, cam5=None):
        n = len(pose_ids)
        if n == 0:
            return
        pi = np.ascontiguousarray(pose_ids, np.int32); li = np.ascontiguousarray(lm_ids, np.int32)
        m = np.ascontiguousarray(np.asarray(meas, np.float64)[:, :dim])
        w = np.ascontiguousarray(info, np.float64)
        cam = None if cam5 is None else np.ascontiguousarray(cam5, np.float64)
        check(lib().cugo_graph_add_edges(self._g, dim, n, _p(pi, _i32p), _p(li, _i32p), _p(m, _f64p),
                                         _p(w, _f64p), None if cam is None else _p(cam, _f64p)))

    def set_camera(self, dim, cam5):
        c = np.ascontiguousarray(cam5, np.float64)
        check(lib().cugo_graph_set_camera(self._g, dim, _p(c, _f64p)))

    def set_information(self, dim, info):
        check(lib().cugo_graph_set_information(self._g, dim, C.c_double(info)))

    def set_robust_kernel(self, dim, rk_type, delta):
        check(lib().cugo_graph_set_robust_kernel(self._g, dim, int(rk_type), C.c_double(delta)))

    def set_outlier_threshold(self, dim, threshold):
        """edges of the set (dim 2 mono / 3 stereo) with chi2 > threshold are inactivated at the
        end of optimize(); 0 disables (ref: EdgeSet::setOutlierThreshold / updateEdges)"""
        check(lib().cugo_graph_set_outlier_threshold(self._g, dim, C.c_double(threshold)))

    def n_outliers(self, dim):
        return lib().cugo_graph_n_outliers(self._g, dim)

    def edge_active(self, dim, n):
        out = np.zeros(n, np.uint8)
        check(lib().cugo_graph_get_edge_active(self._g, dim, n, _p(out, _u8p)))
        return out.astype(bool)

    def set_shard(self, rank, world, fn):
        """fn(device_ptr:int, n_doubles:int, op:int) must all-reduce in place"""
        def _cb(ptr, n, op, user):
            fn(ptr, n, op)
        self._cb = EXCHANGE_FN(_cb)
        check(lib().cugo_graph_set_shard(self._g, rank, world, self._cb, None))

    def set_comm(self, comm):
        """native RCCL exchange on the solver's stream (comm: a Comm object)"""
        check(lib().cugo_graph_set_comm(self._g, comm._c))
        self._comm = comm  # keep it alive

    def exchange_stats(self):
        b, c = C.c_double(), C.c_int32()
        check(lib().cugo_graph_exchange_stats(self._g, C.byref(b), C.byref(c)))
        return dict(bytes=b.value, calls=c.value)

    def set_verbose(self, v):
        lib().cugo_graph_set_verbose(self._g, int(v))

    def initialize(self):
        check(lib().cugo_graph_initialize(self._g))

    def set_option(self, name, value):
        """one run-time switch of this optimiser: "flatten_reuse", "structure_reuse", "init_timing" (the CUGO_*
        environment variables are read once, when the optimiser is created)"""
        check(lib().cugo_graph_set_option(self._g, name.encode(), int(value)))

    def flatten_reuses(self):
        """initialize() calls that found the graph unchanged and only refreshed the estimates"""
        return lib().cugo_graph_flatten_reuses(self._g)

    def optimize(self, n):
        check(lib().cugo_graph_optimize(self._g, int(n)))

    def stats(self):
        n = lib().cugo_graph_n_stats(self._g)
        it = np.zeros(max(n, 1), np.int32); chi = np.zeros(max(n, 1))
        lam = np.zeros(max(n, 1)); rho = np.zeros(max(n, 1)); tr = np.zeros(max(n, 1), np.int32)
        n1 = lib().cugo_graph_get_stats(self._g, _p(it, _i32p), _p(chi, _f64p), n)
        lib().cugo_graph_get_trace(self._g, _p(lam, _f64p), _p(rho, _f64p), _p(tr, _i32p), n)
        return [dict(iteration=int(it[i]), chi2=float(chi[i]), lam=float(lam[i]), rho=float(rho[i]),
                     trials=int(tr[i])) for i in range(n1)]

    def poses(self, ids=None):
        ids = self.pose_ids if ids is None else np.ascontiguousarray(ids, np.int32)
        out = np.zeros((len(ids), 7))
        check(lib().cugo_graph_get_poses(self._g, len(ids), _p(ids, _i32p), _p(out, _f64p)))
        return out

    def landmarks(self, ids=None):
        ids = self.lm_ids if ids is None else np.ascontiguousarray(ids, np.int32)
        out = np.zeros((len(ids), 3))
        check(lib().cugo_graph_get_landmarks(self._g, len(ids), _p(ids, _i32p), _p(out, _f64p)))
        return out

    def set_poses(self, ids, q_t7):
        ids = np.ascontiguousarray(ids, np.int32); q = np.ascontiguousarray(q_t7, np.float64)
        check(lib().cugo_graph_set_poses(self._g, len(ids), _p(ids, _i32p), _p(q, _f64p)))

    def set_landmarks(self, ids, xyz):
        ids = np.ascontiguousarray(ids, np.int32); x = np.ascontiguousarray(xyz, np.float64)
        check(lib().cugo_graph_set_landmarks(self._g, len(ids), _p(ids, _i32p), _p(x, _f64p)))

    def n_active_edges(self):
        return lib().cugo_graph_n_active_edges(self._g)

    def time_profile(self):
        buf = C.create_string_buffer(2048); ms = np.zeros(16)
        n = lib().cugo_graph_time_profile(self._g, buf, 2048, _p(ms, _f64p), 16)
        names = buf.value.decode().split("\n")[:n]
        return dict(zip(names, ms[:n].tolist()))

    def set_float32(self, on):
        """fp32-internal mode: float storage of the Hpl / T block streams (next initialize())"""
        check(lib().cugo_graph_set_float32(self._g, int(on)))

    def set_kernel_timing(self, on):
        """1: event pairs per kernel group and per kernel; 2: one event per group boundary (sums exactly); 0: off"""
        lib().cugo_graph_set_kernel_timing(self._g, int(on))

    def kernel_times(self):
        buf = C.create_string_buffer(8192); ms = np.zeros(96); cnt = np.zeros(96, np.int32)
        n = lib().cugo_graph_kernel_times(self._g, buf, 8192, _p(ms, _f64p), _p(cnt, _i32p), 96)
        names = buf.value.decode().split("\n")[:n]
        return {names[i]: dict(ms=float(ms[i]), launches=int(cnt[i])) for i in range(n)}

    def structure_stats(self):
        o = np.zeros(24)
        n = lib().cugo_graph_structure_stats(self._g, _p(o, _f64p), 24)
        keys = ["hsc_blocks", "products", "nnzL", "chol_flops", "supernodes", "stages", "front_bytes",
                "offdiag_products", "up_potrf_flops", "up_trsm_flops", "up_syrk_flops", "up_ea_bytes",
                "backward_bytes", "schur_slots", "chol_rank_flops", "chol_top_flops", "chol_bcast_bytes",
                "chol_bcasts", "trial_sync_retries", "xchg_sys_bytes", "xchg_sys_full_bytes"]
        return dict(zip(keys[:n], o[:n].tolist()))


def graph_from_arrays(d, per_edge_information=True, per_edge_camera=True, rk=(RK_NONE, 1.0),
                      pose_ids=None, lm_ids=None, plan_only=False):
    """Build a Graph from the flat-array problem dict used by tests/oracle.Problem
    (positions are used as ids unless ids are given)."""
    g = Graph(per_edge_information, per_edge_camera, plan_only)
    P, L = len(d["pose"]), len(d["lm"])
    pid = np.arange(P, dtype=np.int32) if pose_ids is None else np.asarray(pose_ids, np.int32)
    lid = np.arange(L, dtype=np.int32) if lm_ids is None else np.asarray(lm_ids, np.int32)
    g.add_poses(pid, d["pose"], d["pose_fixed"])
    g.add_landmarks(lid, d["lm"], d["lm_fixed"])
    st = np.asarray(d["e_stereo"]).astype(bool)
    ep, el = np.asarray(d["e_pose"]), np.asarray(d["e_lm"])
    meas, om = np.asarray(d["e_meas"], np.float64), np.asarray(d["e_omega"], np.float64)
    cam = np.asarray(d["e_cam"], np.float64).reshape(-1, 5)
    if len(cam) == 1:
        cam = np.tile(cam, (len(ep), 1))
    for dim, sel in ((2, ~st), (3, st)):
        if not per_edge_camera and sel.any():
            g.set_camera(dim, cam[sel][0])
        if not per_edge_information and sel.any():
            g.set_information(dim, float(om[sel][0]))
        g.add_edges(dim, pid[ep[sel]], lid[el[sel]], meas[sel], om[sel],
                    cam[sel] if per_edge_camera else None)
        g.set_robust_kernel(dim, rk[0], rk[1])
    return g


# ---- reference graph-file format (ref: samples/sample_ba_from_file/main.cpp:89-160) ---------
def save_ba_json(path, d, pose_ids=None, lm_ids=None):
    """write a flat-array problem in the JSON layout of the reference's ba_kitti_*.json"""
    import json
    P, L = len(d["pose"]), len(d["lm"])
    pid = np.arange(P) if pose_ids is None else np.asarray(pose_ids)
    lid = np.arange(L) if lm_ids is None else np.asarray(lm_ids)
    cam = np.asarray(d["e_cam"], np.float64).reshape(-1, 5)[0]
    out = {"fx": cam[0], "fy": cam[1], "cx": cam[2], "cy": cam[3], "bf": cam[4],
           "pose_vertices": [{"id": int(pid[i]), "fixed": int(d["pose_fixed"][i]),
                              "q": [float(x) for x in d["pose"][i][:4]],
                              "t": [float(x) for x in d["pose"][i][4:]]} for i in range(P)],
           "landmark_vertices": [{"id": int(lid[i]), "fixed": int(d["lm_fixed"][i]),
                                  "Xw": [float(x) for x in d["lm"][i]]} for i in range(L)],
           "monocular_edges": [], "stereo_edges": []}
    for e in range(len(d["e_pose"])):
        st = bool(d["e_stereo"][e])
        out["stereo_edges" if st else "monocular_edges"].append(
            {"vertexP": int(pid[d["e_pose"][e]]), "vertexL": int(lid[d["e_lm"][e]]),
             "measurement": [float(x) for x in d["e_meas"][e][:3 if st else 2]],
             "information": float(d["e_omega"][e])})
    with open(path, "w") as f:
        json.dump(out, f)


def load_ba_json(path):
    """read ba_kitti_*.json (reference sample format) into the flat-array dict + id arrays"""
    import json
    j = json.load(open(path))
    pv, lv = j["pose_vertices"], j["landmark_vertices"]
    pose_ids = np.array([v["id"] for v in pv], np.int32)
    lm_ids = np.array([v["id"] for v in lv], np.int32)
    ppos = {int(i): k for k, i in enumerate(pose_ids)}
    lpos = {int(i): k for k, i in enumerate(lm_ids)}
    cam = np.array([j["fx"], j["fy"], j["cx"], j["cy"], j["bf"]], np.float64)
    ep, el, st, meas, om = [], [], [], [], []
    for key, stereo in (("monocular_edges", 0), ("stereo_edges", 1)):
        for e in j.get(key, []):
            ep.append(ppos[e["vertexP"]]); el.append(lpos[e["vertexL"]]); st.append(stereo)
            m = list(e["measurement"]) + [0.0] * (3 - len(e["measurement"]))
            meas.append(m); om.append(e["information"])
    d = dict(pose=np.array([v["q"] + v["t"] for v in pv], np.float64).reshape(-1, 7),
             pose_fixed=np.array([v["fixed"] for v in pv], np.uint8),
             lm=np.array([v["Xw"] for v in lv], np.float64).reshape(-1, 3),
             lm_fixed=np.array([v["fixed"] for v in lv], np.uint8),
             e_pose=np.array(ep, np.int32), e_lm=np.array(el, np.int32), e_stereo=np.array(st, np.uint8),
             e_meas=np.array(meas, np.float64).reshape(-1, 3), e_omega=np.array(om, np.float64),
             e_cam=np.tile(cam, (len(ep), 1)))
    return d, pose_ids, lm_ids

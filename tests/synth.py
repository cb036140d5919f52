"""Small seeded ORB-SLAM-style BA graphs for tests (numpy only, no product code).

Poses are world->camera (Xc = R Xw + t), quaternion (x,y,z,w), matching
/root/reference/src/cuda/cuda_block_solver.cu:379-402.
"""
import numpy as np

KITTI_CAM = np.array([718.856, 718.856, 607.1928, 185.2157, 386.1448])


def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz])


def quat_from_rotvec(r):
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.array([0.5 * r[0], 0.5 * r[1], 0.5 * r[2], 1.0])
    s = np.sin(th / 2) / th
    return np.array([r[0] * s, r[1] * s, r[2] * s, np.cos(th / 2)])


def quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def make_problem(n_poses=6, n_landmarks=40, mean_obs=3.5, stereo_frac=0.6, pix_noise=1.0,
                 pose_noise=(0.01, 0.05), lm_noise=0.02, seed=0, fixed_poses=(0,),
                 fixed_landmarks=(), window=None, loop_closure=False, per_edge_cam=False):
    """Returns dict of arrays with the field names of tests/oracle.Problem."""
    rng = np.random.default_rng(seed)
    cam = KITTI_CAM.copy()
    # trajectory: camera moves along +z (optical axis) with gentle yaw
    Rwc, twc = [], []
    yaw = 0.0
    pos = np.zeros(3)
    for i in range(n_poses):
        yaw += rng.normal(0, 0.01)
        q = quat_from_rotvec(np.array([rng.normal(0, 0.002), yaw, rng.normal(0, 0.002)]))
        R = quat_to_R(q)            # camera->world rotation
        Rwc.append(R)
        twc.append(pos.copy())
        pos = pos + R @ np.array([0, 0, 1.0 + rng.normal(0, 0.05)])
    pose_gt = np.zeros((n_poses, 7))
    for i in range(n_poses):
        Rcw = Rwc[i].T
        tcw = -Rcw @ twc[i]
        # quaternion of Rcw: conjugate of the cam->world quaternion
        # recover from matrix robustly
        q = _R_to_quat(Rcw)
        pose_gt[i, :4] = q
        pose_gt[i, 4:] = tcw

    window = window or max(2, int(round(2 * mean_obs)))
    lm_gt = np.zeros((n_landmarks, 3))
    e_pose, e_lm, e_st, e_meas, e_om = [], [], [], [], []
    for l in range(n_landmarks):
        first = int(rng.integers(0, max(1, n_poses - 1)))
        k = int(np.clip(rng.geometric(1.0 / max(mean_obs - 1.0, 1.0)) + 1, 2, window))
        obs = list(range(first, min(n_poses, first + k)))
        if len(obs) < 2:
            obs = [n_poses - 2, n_poses - 1]
        if loop_closure and rng.random() < 0.15:
            far = int(rng.integers(0, n_poses))
            if far not in obs:
                obs.append(far)
        # place the point in front of the middle observing camera
        mid = obs[len(obs) // 2]
        depth = rng.uniform(6, 40)
        u = rng.uniform(100, 1100)
        v = rng.uniform(30, 340)
        Xc = np.array([(u - cam[2]) / cam[0] * depth, (v - cam[3]) / cam[1] * depth, depth])
        lm_gt[l] = Rwc[mid] @ Xc + twc[mid]
        stereo = rng.random() < stereo_frac
        for p in sorted(obs):
            R = quat_to_R(pose_gt[p, :4])
            xc = R @ lm_gt[l] + pose_gt[p, 4:]
            if xc[2] < 0.5:
                continue
            uu = cam[0] * xc[0] / xc[2] + cam[2] + rng.normal(0, pix_noise)
            vv = cam[1] * xc[1] / xc[2] + cam[3] + rng.normal(0, pix_noise)
            ur = uu - cam[4] / xc[2] + rng.normal(0, pix_noise)
            lvl = int(rng.integers(0, 8))
            e_pose.append(p); e_lm.append(l); e_st.append(1 if stereo else 0)
            e_meas.append([uu, vv, ur if stereo else 0.0])
            e_om.append(1.0 / (1.2 ** lvl) ** 2)
    pose = pose_gt.copy()
    for i in range(n_poses):
        if i in fixed_poses:
            continue
        dq = quat_from_rotvec(rng.normal(0, pose_noise[0], 3))
        q = quat_mul(dq, pose[i, :4])
        pose[i, :4] = q / np.linalg.norm(q) * (1 if q[3] >= 0 else -1)
        pose[i, 4:] += rng.normal(0, pose_noise[1], 3)
    lm = lm_gt + rng.normal(0, 1, lm_gt.shape) * lm_noise * np.linalg.norm(lm_gt - np.array(twc).mean(0), axis=1, keepdims=True) * 0.1
    for l in fixed_landmarks:
        lm[l] = lm_gt[l]
    pf = np.zeros(n_poses, np.uint8); pf[list(fixed_poses)] = 1
    lf = np.zeros(n_landmarks, np.uint8)
    if len(fixed_landmarks):
        lf[list(fixed_landmarks)] = 1
    E = len(e_pose)
    cams = np.tile(cam, (E, 1))
    if per_edge_cam:
        cams = cams * (1 + rng.normal(0, 1e-3, (E, 1)))
    return dict(pose=pose, pose_fixed=pf, lm=lm, lm_fixed=lf,
                e_pose=np.array(e_pose, np.int32), e_lm=np.array(e_lm, np.int32),
                e_stereo=np.array(e_st, np.uint8), e_meas=np.array(e_meas),
                e_omega=np.array(e_om), e_cam=cams, pose_gt=pose_gt, lm_gt=lm_gt)


def _R_to_quat(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s,
                      0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[3] = (R[k, j] - R[j, k]) / s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def problem_fields(d):
    keys = ["pose", "pose_fixed", "lm", "lm_fixed", "e_pose", "e_lm", "e_stereo", "e_meas",
            "e_omega", "e_cam"]
    return [d[k] for k in keys]

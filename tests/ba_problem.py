"""Deterministic synthetic BA problems (SURVEY §8d config 4 shape), numpy only."""
import numpy as np

F, CX, CY = 718.856, 607.1928, 185.2157
W, H = 1241, 376


def quat_from_yaw(a):
    return np.array([np.cos(a / 2), 0.0, np.sin(a / 2), 0.0])


def quat_to_R(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def make_problem(seed, K, N, max_len=None, noise=0.5, pt_sigma=0.10, pose_sigma=(0.05, 0.008), dense=False):
    """poses are world-wrt-camera [qw qx qy qz tx ty tz] (src/bundle_adjuster.hpp:16-21,50)."""
    rng = np.random.default_rng(seed)
    max_len = max_len or K
    poses = []
    for k in range(K):
        yaw = 0.01 * k
        Rcw = quat_to_R(quat_from_yaw(yaw)).T          # camera-in-world rotation is yaw; world-wrt-camera is its transpose
        C = np.array([0.05 * k * k * 0.1, 0.0, 1.0 * k])
        q = quat_from_yaw(-yaw)
        t = -quat_to_R(q) @ C
        poses.append(np.concatenate([q, t]))
    poses = np.array(poses)
    pts = np.stack([rng.uniform(-12, 12, N), rng.uniform(-2.5, 2.0, N), rng.uniform(4, 60 + K, N)], 1)
    op, oj, uv = [], [], []
    for j in range(N):
        Lj = K if dense else int(rng.integers(2, max_len + 1))
        s = 0 if dense else int(rng.integers(0, K - Lj + 1))
        for k in range(s, s + Lj):
            Xc = quat_to_R(poses[k, :4]) @ pts[j] + poses[k, 4:]
            if Xc[2] < 1.0:
                continue
            u = F * Xc[0] / Xc[2] + CX
            v = F * Xc[1] / Xc[2] + CY
            if not (0 <= u < W and 0 <= v < H):
                continue
            op.append(k); oj.append(j); uv.append([u + rng.normal(0, noise), v + rng.normal(0, noise)])
    op, oj, uv = np.array(op, np.int32), np.array(oj, np.int32), np.array(uv)
    # drop landmarks with < 2 observations, re-index compactly (keeps landmark-major order)
    cnt = np.bincount(oj, minlength=N)
    keep = cnt[oj] >= 2
    op, oj, uv = op[keep], oj[keep], uv[keep]
    used = np.unique(oj)
    remap = -np.ones(N, np.int64); remap[used] = np.arange(len(used))
    oj = remap[oj].astype(np.int32)
    pts = pts[used]
    pts0 = pts + rng.normal(0, pt_sigma, pts.shape)
    poses0 = poses.copy()
    for k in range(1, K):
        dq = np.concatenate([[1.0], rng.normal(0, pose_sigma[1] / 2, 3)])
        q = poses0[k, :4]
        w1, x1, y1, z1 = dq; w2, x2, y2, z2 = q
        poses0[k, :4] = [w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                         w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2]
        poses0[k, 4:] += rng.normal(0, pose_sigma[0], 3)
    return dict(poses_gt=poses, points_gt=pts, poses0=poses0, points0=pts0, op=op, oj=oj, uv=uv)


def pose_error(a, b):
    """(max translation diff, max rotation angle diff) between two K x 7 pose sets."""
    dt = np.abs(a[:, 4:] - b[:, 4:]).max()
    ang = 0.0
    for qa, qb in zip(a[:, :4], b[:, :4]):
        Ra, Rb = quat_to_R(qa), quat_to_R(qb)
        c = (np.trace(Ra.T @ Rb) - 1) / 2
        ang = max(ang, float(np.arccos(np.clip(c, -1, 1))))
    return dt, ang

"""Seeded synthetic inputs for the hot path: procedural dual-camera images,
random descriptor sets with CSR vocabulary buckets, and the C4 local-BA problem.

numpy only; used by tests/, bench.py and __graft_entry__.smoke(). The real sequence
(indoor_lab_loop.avi) and vocabulary (ORBvoc) of the reference are external downloads and are
not available, so every input is generated here and `data` is reported as "synthetic".
"""
import numpy as np

# shipped rig calibration (reference Dual-LenaCV.yaml:12-47), parsed to float like Tracking.cc:109-160
RIG = {
    "cam0": dict(fx=558.4684, fy=560.0944, cx=326.7993, cy=262.9017),
    "cam1": dict(fx=546.597961663159, fy=546.254254416417, cx=332.758939924785, cy=247.385425357685,
                 q=(0.82351, -0.00262741, 0.567257, 0.00665084),   # qw, qx, qy, qz
                 t=(0.069481, -0.000909887, -0.0713882)),
}


def _draw_scene(rng, width, height, n_rect, n_disc):
    img = np.full((height, width), 128.0, np.float32)
    for _ in range(n_rect):
        w, h = rng.integers(8, 81, 2)
        x, y = rng.integers(-20, width), rng.integers(-20, height)
        img[max(y, 0):max(y + h, 0), max(x, 0):max(x + w, 0)] = float(rng.integers(0, 256))
    for _ in range(n_disc):
        r = int(rng.integers(4, 41))
        cx, cy = int(rng.integers(0, width)), int(rng.integers(0, height))
        x0, x1, y0, y1 = max(cx - r, 0), min(cx + r + 1, width), max(cy - r, 0), min(cy + r + 1, height)
        yy, xx = np.mgrid[y0:y1, x0:x1]
        mask = (xx - cx) ** 2 + (yy - cy) ** 2 <= r * r
        img[y0:y1, x0:x1][mask] = float(rng.integers(0, 256))
    gy, gx = np.mgrid[0:height, 0:width]
    img += 20.0 * (gx / width - 0.5) + 12.0 * (gy / height - 0.5)
    return img


def _warp_h(img, H):
    """bilinear warp: out(x,y) = img(H^-1 (x,y))."""
    h, w = img.shape
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    Hi = np.linalg.inv(H)
    d = Hi[2, 0] * xs + Hi[2, 1] * ys + Hi[2, 2]
    sx = (Hi[0, 0] * xs + Hi[0, 1] * ys + Hi[0, 2]) / d
    sy = (Hi[1, 0] * xs + Hi[1, 1] * ys + Hi[1, 2]) / d
    sx = np.clip(sx, 0, w - 1.001)
    sy = np.clip(sy, 0, h - 1.001)
    x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
    fx, fy = sx - x0, sy - y0
    a = img[y0, x0] * (1 - fx) + img[y0, x0 + 1] * fx
    b = img[y0 + 1, x0] * (1 - fx) + img[y0 + 1, x0 + 1] * fx
    return (a * (1 - fy) + b * fy).astype(np.float32)


_SCENES = {}


def frame_pair(width=640, height=480, stream=0, frame=0):
    """Two u8 images (cam0, cam1) of stream `stream` at time `frame`.

    Scene: mid-grey canvas + random filled rectangles / discs (sizes 8-80 px) + smooth gradient;
    frame t = scene translated by (3t, t) px; cam1 = cam0's view warped by a fixed homography;
    i.i.d. Gaussian noise sigma=3 with seed 1234 + stream*100 + cam*10 + frame.
    """
    big = max(width, height) >= 1000
    key = (width, height, stream)
    if key not in _SCENES:
        rng = np.random.default_rng(1234 + stream * 100)
        pad = 256
        _SCENES.clear()
        _SCENES[key] = _draw_scene(rng, width + pad, height + pad, 1200 if big else 400, 600 if big else 200)
    scene = _SCENES[key]
    ox, oy = (3 * frame) % 250, frame % 250
    view = scene[oy:oy + height, ox:ox + width]
    H = np.array([[0.96, 0.03, 14.0], [-0.025, 0.98, 9.0], [2.0e-5, -1.0e-5, 1.0]])
    out = []
    for cam in (0, 1):
        base = view if cam == 0 else _warp_h(view, H)
        rng = np.random.default_rng(1234 + stream * 100 + cam * 10 + frame)
        noisy = base + rng.normal(0.0, 3.0, base.shape).astype(np.float32)
        out.append(np.clip(np.rint(noisy), 0, 255).astype(np.uint8))
    return out[0], out[1]


def random_descriptors(n, seed=7):
    return np.random.default_rng(seed).integers(0, 256, (n, 32), dtype=np.uint8)


def noisy_copy(desc, flip_bits=20, seed=11):
    """descriptors with `flip_bits` random bits flipped per row (so knn2 finds real matches)."""
    rng = np.random.default_rng(seed)
    out = desc.copy()
    for i in range(len(out)):
        bits = rng.choice(256, flip_bits, replace=False)
        for b in bits:
            out[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return out


def csr_buckets(n_items, n_buckets=100, seed=3):
    """Random partition emulating a DBoW2 FeatureVector (Frame.cc:400-402): every item in exactly
    one node; returns (node_ids ascending, off[n_nodes+1], idx) with idx ascending inside a node."""
    rng = np.random.default_rng(seed)
    node = rng.integers(0, n_buckets, n_items)
    ids = np.unique(node)
    off = [0]
    idx = []
    for k in ids:
        members = np.nonzero(node == k)[0]
        idx.extend(members.tolist())
        off.append(len(idx))
    return ids.astype(np.int32), np.asarray(off, np.int32), np.asarray(idx, np.int32)


# ----------------------------------------------------------------------------- BA (C4)
def _quat_from_R(R):
    """Shoemake / Eigen quaternion from rotation matrix, returns (x,y,z,w) normalised, w >= 0."""
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        s = np.sqrt(t + 1.0)
        w = 0.5 * s
        s = 0.5 / s
        x, y, z = (R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        v = np.zeros(3)
        v[i] = 0.5 * s
        s = 0.5 / s
        w = (R[k, j] - R[j, k]) * s
        v[j] = (R[j, i] + R[i, j]) * s
        v[k] = (R[k, i] + R[i, k]) * s
        x, y, z = v
    q = np.array([x, y, z, w])
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def _rodrigues(rv):
    th = np.linalg.norm(rv)
    if th < 1e-12:
        return np.eye(3)
    k = rv / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def rig_extrinsics_f32():
    """4x4 float32 T_c (rig -> camera c) for the shipped rig, built like Tracking.cc:146-171."""
    T0 = np.eye(4, dtype=np.float32)
    qw, qx, qy, qz = (float(np.float32(v)) for v in RIG["cam1"]["q"])
    t = [float(np.float32(v)) for v in RIG["cam1"]["t"]]
    R = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                  [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                  [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
    T1 = np.eye(4)
    T1[:3, :3] = R
    T1[:3, 3] = t
    return T0, T1.astype(np.float32)


def rig_adjoint_f32(T, exact=False):
    """Cameras::setExtrinsics adjoint (Cameras.cc:27-37): [[R, R t^],[0, R]] in float32 (Q1: the
    reference leaves the lower-left block uninitialised, 0 here); exact -> g2o adj() [[R,0],[t^R,R]].
    Returns (adj 6x6 float64, ext7 = tx,ty,tz,qx,qy,qz,qw float64)."""
    T = np.asarray(T, np.float32)
    R32, t32 = T[:3, :3], T[:3, 3]
    q = _quat_from_R(R32.astype(np.float64))
    ext7 = np.concatenate([t32.astype(np.float64), q])
    adj = np.zeros((6, 6))
    if not exact:
        th = np.array([[0, -t32[2], t32[1]], [t32[2], 0, -t32[0]], [-t32[1], t32[0], 0]], np.float32)
        Rt = np.zeros((3, 3), np.float32)
        for i in range(3):
            for j in range(3):
                acc = np.float32(0)
                for k in range(3):
                    acc = np.float32(acc + np.float32(R32[i, k] * th[k, j]))
                Rt[i, j] = acc
        adj[:3, :3] = R32
        adj[3:, 3:] = R32
        adj[:3, 3:] = Rt
    else:
        x, y, z, w = q
        Rn = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                       [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                       [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        t = t32.astype(np.float64)
        th = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        adj[:3, :3] = Rn
        adj[3:, 3:] = Rn
        adj[3:, :3] = th @ Rn
    return adj, ext7


def ba_problem(n_poses=50, n_fixed=10, n_points=2000, obs_per_point=10, seed=42, outlier_frac=0.05,
               exact_adjoint=False, noise=True):
    """poses on a loop, points in a 12x4x12 m box, dual-camera observations with
    the shipped rig, octave-dependent pixel noise, 5 % gross outliers, perturbed initial state.
    All state passes through float32 like the reference's cv::Mat CV_32F (Converter.cc:58-68).

    Returns dict of numpy arrays (poses[P,7] = tx,ty,tz,qx,qy,qz,qw world->rig; cams = list of
    (fx,fy,cx,cy,ext7,adj6x6)), plus ground truth under 'gt_*'.
    """
    rng = np.random.default_rng(seed)
    P, L = n_poses, n_points
    if obs_per_point > P:
        raise ValueError("obs_per_point %d > n_poses %d" % (obs_per_point, P))
    T0, T1 = rig_extrinsics_f32()
    cams = []
    for c, T in enumerate((T0, T1)):
        k = RIG["cam%d" % c]
        adj, ext7 = rig_adjoint_f32(T, exact_adjoint)
        cams.append(dict(fx=float(np.float32(k["fx"])), fy=float(np.float32(k["fy"])),
                         cx=float(np.float32(k["cx"])), cy=float(np.float32(k["cy"])),
                         ext7=ext7, adj=adj, T=T.astype(np.float64)))
    # ground-truth poses: centres on a circle of circumference ~10 m, small rotations (<= 10 deg)
    R_gt, t_gt = [], []
    radius = 10.0 / (2 * np.pi)
    for i in range(P):
        a = 2 * np.pi * i / P
        centre = np.array([radius * np.cos(a), 0.1 * np.sin(3 * a), radius * np.sin(a)])
        rv = rng.uniform(-1, 1, 3)
        rv = rv / np.linalg.norm(rv) * np.deg2rad(rng.uniform(0, 10))
        R = _rodrigues(rv)
        R_gt.append(R)
        t_gt.append(-R @ centre)
    R_gt, t_gt = np.array(R_gt), np.array(t_gt)

    def project_all(X):
        """X[n,3] -> uv[n,P,2,2], z[n,P,2]"""
        pr = np.einsum("pij,nj->npi", R_gt, X) + t_gt[None]
        uv = np.zeros((len(X), P, 2, 2))
        z = np.zeros((len(X), P, 2))
        for c, cam in enumerate(cams):
            pc = pr @ cam["T"][:3, :3].T + cam["T"][:3, 3]
            z[:, :, c] = pc[:, :, 2]
            with np.errstate(divide="ignore", invalid="ignore"):
                uv[:, :, c, 0] = cam["fx"] * pc[:, :, 0] / pc[:, :, 2] + cam["cx"]
                uv[:, :, c, 1] = cam["fy"] * pc[:, :, 1] / pc[:, :, 2] + cam["cy"]
        return uv, z

    pts, e_pose, e_point, e_cam, e_uv = [], [], [], [], []
    while len(pts) < L:
        X = np.column_stack([rng.uniform(-6, 6, 4 * L), rng.uniform(-2, 2, 4 * L), rng.uniform(-6, 6, 4 * L)])
        uv, z = project_all(X)
        vis = (z > 0.1) & (uv[..., 0] >= 0) & (uv[..., 0] < 640) & (uv[..., 1] >= 0) & (uv[..., 1] < 480)
        for n in range(len(X)):
            if len(pts) >= L:
                break
            poses_ok = np.nonzero(vis[n].any(axis=1))[0]
            if len(poses_ok) < obs_per_point:
                continue
            chosen = np.sort(rng.choice(poses_ok, obs_per_point, replace=False))
            l = len(pts)
            pts.append(X[n])
            for p in chosen:
                both = vis[n, p]
                c = int(rng.integers(0, 2)) if both.all() else int(np.argmax(both))
                e_pose.append(p); e_point.append(l); e_cam.append(c); e_uv.append(uv[n, p, c])
    pts = np.array(pts)
    e_pose, e_point, e_cam = (np.asarray(a, np.int32) for a in (e_pose, e_point, e_cam))
    e_uv = np.array(e_uv)
    E = len(e_pose)
    octave = rng.integers(0, 8, E)
    scale = np.ones(8, np.float32)
    for i in range(1, 8):
        scale[i] = np.float32(np.float64(scale[i - 1]) * np.float64(np.float32(1.2)))
    inv_sigma2 = (np.float32(1.0) / (scale * scale)).astype(np.float32)
    obs = e_uv.copy()
    if noise:
        obs += rng.normal(0, 1, (E, 2)) * scale[octave][:, None]
        n_out = int(round(outlier_frac * E))
        bad = rng.choice(E, n_out, replace=False)
        obs[bad] += rng.choice([-50.0, 50.0], (n_out, 2))
    obs = obs.astype(np.float32).astype(np.float64)
    # fixed: pose 0 (fixId, Optimizer.cc:483) and the last n_fixed poses (fixed-camera set :489-500)
    fixed = np.zeros(P, np.uint8)
    fixed[0] = 1
    fixed[P - n_fixed:] = 1
    poses = np.zeros((P, 7))
    for i in range(P):
        R, t = R_gt[i], t_gt[i]
        if noise and not fixed[i]:
            R = _rodrigues(rng.normal(0, 0.02, 3)) @ R
            t = t + rng.normal(0, 0.05, 3)
        R32, t32 = R.astype(np.float32).astype(np.float64), t.astype(np.float32).astype(np.float64)
        poses[i, :3] = t32
        poses[i, 3:] = _quat_from_R(R32)
    gt_poses = np.zeros((P, 7))
    for i in range(P):
        gt_poses[i, :3] = t_gt[i]
        gt_poses[i, 3:] = _quat_from_R(R_gt[i])
    points = pts + (rng.normal(0, 0.05, pts.shape) if noise else 0.0)
    points = points.astype(np.float32).astype(np.float64)
    return dict(poses=poses, pose_fixed=fixed, points=points, edge_pose=e_pose, edge_point=e_point,
                edge_cam=e_cam, obs=obs, inv_sigma2=inv_sigma2[octave].astype(np.float64), cams=cams,
                huber_delta=float(np.float32(np.sqrt(5.991))), chi2_th=5.991, iters1=5, iters2=10,
                gt_poses=gt_poses, gt_points=pts)


def pose_problem(n_frames=16, obs_per_frame=400, seed=5, outlier_frac=0.1, point_noise=0.01):
    """Synthetic input of Optimizer::PoseOptimization (Optimizer.cc:250-405) for a batch of frames: frame f = ground-truth
    pose f of a ba_problem scene, observing `obs_per_frame` map points with the dual-camera rig; map points carry a
    small error (they are float32 state of the map), observations octave-dependent noise + gross outliers, the initial
    pose is the motion-model guess (perturbed ground truth). Frames 0/1 are degenerate on purpose (2 and 9 edges)."""
    per_point = min(6, n_frames)
    n_points = max(60, n_frames * obs_per_frame // per_point + 50)
    ba = ba_problem(n_poses=n_frames, n_fixed=0, n_points=n_points, obs_per_point=per_point, seed=seed, outlier_frac=outlier_frac)
    rng = np.random.default_rng(seed + 1000)
    xw_all = (ba["gt_points"] + rng.normal(0, point_noise, ba["gt_points"].shape)).astype(np.float32).astype(np.float64)
    off, xw, obs, w, cam = [0], [], [], [], []
    for f in range(n_frames):
        e = np.nonzero(ba["edge_pose"] == f)[0]
        keep = obs_per_frame if f > 1 else (2 if f == 0 else 9)
        e = e[:keep]
        xw.append(xw_all[ba["edge_point"][e]]); obs.append(ba["obs"][e]); w.append(ba["inv_sigma2"][e]); cam.append(ba["edge_cam"][e])
        off.append(off[-1] + len(e))
    return dict(poses=ba["poses"].copy(), edge_off=np.asarray(off, np.int32), xw=np.concatenate(xw), obs=np.concatenate(obs),
                inv_sigma2=np.concatenate(w), edge_cam=np.concatenate(cam).astype(np.int32), cams=ba["cams"],
                huber_delta=float(np.float32(np.sqrt(5.991))), chi2_th=[float(np.float32(5.991))] * 4, its=[10, 10, 10, 10],
                gt_poses=ba["gt_poses"])


def projection_problem(n_per_cam=900, n_queries=700, seed=13, th=1.0, big_windows=0):
    """Synthetic input of ORBmatcher::SearchByProjection / SearchByProjectionOnCam: a dual-camera frame (keypoints with
    octaves / angles / descriptors, some features already holding a map point) and ordered queries (map points or the
    last frame's features) that project near features of the frame: most carry a noisy copy of their feature's
    descriptor, several target the SAME feature (exercises the "already matched" rule), some are invalid, some are
    distractors. `big_windows` queries get a window covering most of the image (candidate lists beyond any fixed cap).
    The grid (CSR) is NOT included: build it with dcs_frame_grid / oracle.frame_grid."""
    rng = np.random.default_rng(seed)
    n_cams = 2
    cam_off = np.array([0, n_per_cam, 2 * n_per_cam], np.int32)
    N = int(cam_off[-1])
    scale = np.ones(8, np.float32)
    for i in range(1, 8):
        scale[i] = np.float32(np.float64(scale[i - 1]) * np.float64(np.float32(1.2)))
    kp_x = rng.uniform(-3, 643, N).astype(np.float32)            # undistortion can push keypoints slightly outside
    kp_y = rng.uniform(-3, 483, N).astype(np.float32)
    kp_x[rng.choice(N, 40, replace=False)] = np.float32(320.5)   # crowded columns: several features per cell, exact ties in x
    octave = np.minimum(rng.geometric(0.35, N) - 1, 7).astype(np.int32)
    angle = rng.uniform(0, 360, N).astype(np.float32)
    desc = random_descriptors(N, seed=seed + 1)
    taken = (rng.random(N) < 0.08).astype(np.uint8)
    min_x, max_x, min_y, max_y = np.float32(-2.5), np.float32(642.0), np.float32(-1.5), np.float32(481.0)
    frame = dict(cam_off=cam_off, kp_x=kp_x, kp_y=kp_y, kp_octave=octave, kp_angle=angle, desc=desc, taken=taken,
                 min_x=np.full(n_cams, min_x, np.float32), min_y=np.full(n_cams, min_y, np.float32),
                 grid_w_inv=np.full(n_cams, np.float32(64) / np.float32(max_x - min_x), np.float32),
                 grid_h_inv=np.full(n_cams, np.float32(48) / np.float32(max_y - min_y), np.float32))
    target = rng.integers(0, N, n_queries)
    target[1::9] = target[0::9][:len(target[1::9])]              # duplicates right after the original
    cam = np.searchsorted(cam_off, target, side="right").astype(np.int32) - 1
    level = np.clip(octave[target] + rng.integers(-1, 2, n_queries), 0, 7).astype(np.int32)
    view_cos = rng.uniform(0.99, 1.0, n_queries)
    r = np.where(view_cos > 0.998, np.float32(2.5), np.float32(4.0)).astype(np.float32)
    if th != 1.0:
        r = (r * np.float32(th)).astype(np.float32)
    radius = (r * scale[level]).astype(np.float32)
    u = (kp_x[target] + rng.normal(0, 1.5, n_queries)).astype(np.float32)
    v = (kp_y[target] + rng.normal(0, 1.5, n_queries)).astype(np.float32)
    qdesc = noisy_copy(desc[target], flip_bits=18, seed=seed + 2)
    distract = rng.random(n_queries) < 0.15
    qdesc[distract] = random_descriptors(int(distract.sum()), seed=seed + 3)
    valid = (rng.random(n_queries) > 0.05).astype(np.uint8)
    qangle = ((angle[target] + rng.normal(0, 4, n_queries)) % 360).astype(np.float32)
    flip = rng.random(n_queries) < 0.2                          # rotation-inconsistent matches for the histogram test
    qangle[flip] = rng.uniform(0, 360, int(flip.sum())).astype(np.float32)
    if big_windows:
        big = rng.choice(n_queries, big_windows, replace=False)
        radius[big] = np.float32(400.0)
    queries = dict(valid=valid, cam=cam, u=u, v=v, radius=radius, min_level=(level - 1).astype(np.int32),
                   max_level=(level + 1).astype(np.int32), desc=qdesc, angle=qangle)
    return frame, queries


def initialization_problem(n_per_cam=500, seed=21, window=100.0, crowd=0.3):
    """Synthetic input of ORBmatcher::SearchForInitialization (ORBmatcher.cc:1117-1251): frame F2 (dual camera, grid NOT included)
    and F1's key points as ordered queries. Most level-0 key points of F1's camera 0 have a moved, noisy copy in F2; a fraction
    `crowd` of them are rivals of an earlier key point (same target, a few pixels away, another noise level), so that later
    queries find features held by earlier ones with a smaller / equal / larger distance (the vMatchedDistance gate and the
    stealing rule both fire). Angles are consistent up to noise except for a random 20 % (rotation histogram)."""
    rng = np.random.default_rng(seed)
    n_cams = 2
    cam_off = np.array([0, n_per_cam, 2 * n_per_cam], np.int32)
    N = int(cam_off[-1])
    kp_x2 = rng.uniform(5, 635, N).astype(np.float32)
    kp_y2 = rng.uniform(5, 475, N).astype(np.float32)
    oct2 = np.where(rng.random(N) < 0.6, 0, np.minimum(rng.geometric(0.4, N), 7)).astype(np.int32)
    ang2 = rng.uniform(0, 360, N).astype(np.float32)
    desc2 = random_descriptors(N, seed=seed + 1)
    min_x, max_x, min_y, max_y = np.float32(0.0), np.float32(640.0), np.float32(0.0), np.float32(480.0)
    frame2 = dict(cam_off=cam_off, kp_x=kp_x2, kp_y=kp_y2, kp_octave=oct2, kp_angle=ang2, desc=desc2, taken=np.zeros(N, np.uint8),
                  min_x=np.full(n_cams, min_x, np.float32), min_y=np.full(n_cams, min_y, np.float32),
                  grid_w_inv=np.full(n_cams, np.float32(64) / np.float32(max_x - min_x), np.float32),
                  grid_h_inv=np.full(n_cams, np.float32(48) / np.float32(max_y - min_y), np.float32))
    # F1: global order = camera 0 then camera 1
    n1 = N
    cam1 = np.repeat(np.arange(n_cams, dtype=np.int32), n_per_cam)
    lvl0_cam0 = np.nonzero((oct2[:n_per_cam] == 0))[0]
    target = rng.choice(lvl0_cam0, n1)                                   # F2 feature each F1 key point derives from
    rival = rng.random(n1) < crowd
    prev = np.maximum(np.arange(n1) - rng.integers(1, 6, n1), 0)
    target[rival] = target[prev[rival]]                                   # same target as a key point a few places earlier
    kp_x1 = (kp_x2[target] + rng.uniform(-40, 40, n1)).astype(np.float32)
    kp_y1 = (kp_y2[target] + rng.uniform(-40, 40, n1)).astype(np.float32)
    oct1 = np.where(rng.random(n1) < 0.7, 0, rng.integers(1, 8, n1)).astype(np.int32)
    desc1 = desc2[target].copy()
    flips = rng.choice([4, 8, 8, 12, 16, 24, 40, 70], n1)                 # equal noise levels on purpose: distance ties
    for i in range(n1):
        bits = rng.choice(256, int(flips[i]), replace=False)
        for b in bits:
            desc1[i, b >> 3] ^= np.uint8(1 << (b & 7))
    ang1 = ((ang2[target] + rng.normal(0, 5, n1)) % 360).astype(np.float32)
    odd = rng.random(n1) < 0.2
    ang1[odd] = rng.uniform(0, 360, int(odd.sum())).astype(np.float32)
    valid = ((cam1 == 0) & (oct1 == 0)).astype(np.uint8)
    queries = dict(valid=valid, cam=cam1, u=kp_x1, v=kp_y1, radius=np.full(n1, np.float32(window), np.float32),
                   min_level=oct1.copy(), max_level=oct1.copy(), desc=desc1, angle=ang1)
    return frame2, queries


# ----------------------------------------------------------------------------- BoW vocabulary
def vocabulary(k=10, L=6, seed=1, flip=24, ragged=0.0, early_leaf=0.0, stop_frac=0.0, dup_frac=0.0):
    """Synthetic DBoW2 vocabulary in the column form of the reference's text file (TemplatedVocabulary.h:1362-1446; the real
    ORBvoc.txt is an external download, Vocabulary/download_link.txt): row i = node i + 1 -> (parent, is_leaf, desc[32], weight).
    A k-ary tree of depth L built breadth-first (children of a node are consecutive rows, as DBoW2's k-means training numbers
    them); a child's descriptor = the parent's with ~`flip`/depth random bits flipped, so the greedy descent is meaningful.
    ragged: probability that an inner node has fewer than k children; early_leaf: probability that a node above depth L is a
    leaf; stop_frac: leaves with weight 0 (stopped words); dup_frac: children that copy a sibling's descriptor (ties -> the
    first child wins). Returns dict(k, L, parent, is_leaf, desc, weight, depth)."""
    rng = np.random.default_rng(seed)
    parent, is_leaf, desc, depth = [], [], [], []
    frontier = [(0, np.zeros(32, np.uint8), 0)]                   # (node id, descriptor, depth); root descriptor unused
    n_nodes = 1
    while frontier:
        nxt = []
        for (nid, d, dep) in frontier:
            n_kids = k if (dep == 0 or rng.random() >= ragged) else int(rng.integers(1, k + 1))
            kids = []
            for c in range(n_kids):
                if dep == 0:
                    cd = rng.integers(0, 256, 32, dtype=np.uint8)
                elif kids and rng.random() < dup_frac:
                    cd = kids[int(rng.integers(0, len(kids)))].copy()
                else:
                    bits = np.unpackbits(d)
                    nflip = max(1, int(flip // (dep + 1)) + int(rng.integers(0, 4)))
                    bits[rng.choice(256, nflip, replace=False)] ^= 1
                    cd = np.packbits(bits)
                kids.append(cd)
                leaf = (dep + 1 == L) or (dep + 1 >= 2 and rng.random() < early_leaf)
                parent.append(nid); is_leaf.append(1 if leaf else 0); desc.append(cd); depth.append(dep + 1)
                if not leaf:
                    nxt.append((n_nodes, cd, dep + 1))
                n_nodes += 1
        frontier = nxt
    is_leaf = np.asarray(is_leaf, np.uint8)
    weight = np.zeros(len(parent))
    nl = int(is_leaf.sum())
    w = rng.uniform(0.3, 9.0, nl)                                  # idf-like: -log(Ni/N)
    w[rng.random(nl) < stop_frac] = 0.0
    weight[is_leaf != 0] = w
    return dict(k=k, L=L, parent=np.asarray(parent, np.int32), is_leaf=is_leaf, desc=np.asarray(desc, np.uint8).reshape(-1, 32),
                weight=weight, depth=np.asarray(depth, np.int32))


def vocabulary_fast(k=10, L=6, seed=1, flip=24):
    """Full k-ary tree like vocabulary() but vectorised level by level (for the k = 10, L = 6 size: 1.1 M nodes)."""
    rng = np.random.default_rng(seed)
    parents, descs, leafs = [], [], []
    prev_ids = np.zeros(1, np.int64); prev_desc = np.zeros((1, 32), np.uint8)
    next_id = 1
    for dep in range(L):
        n_par = len(prev_ids)
        par = np.repeat(prev_ids, k)
        if dep == 0:
            d = rng.integers(0, 256, (n_par * k, 32), dtype=np.uint8)
        else:
            bits = np.unpackbits(np.repeat(prev_desc, k, axis=0), axis=1)
            nflip = max(1, flip // (dep + 1))
            cols = rng.integers(0, 256, (n_par * k, nflip))
            np.put_along_axis(bits, cols, 1 - np.take_along_axis(bits, cols, 1), 1)
            d = np.packbits(bits, axis=1)
        ids = np.arange(next_id, next_id + n_par * k)
        next_id += n_par * k
        parents.append(par); descs.append(d); leafs.append(np.full(n_par * k, 1 if dep + 1 == L else 0, np.uint8))
        prev_ids, prev_desc = ids, d
    # rows must be in node-id order: breadth-first numbering already is
    parent = np.concatenate(parents).astype(np.int32); desc = np.concatenate(descs); is_leaf = np.concatenate(leafs)
    weight = np.zeros(len(parent)); nl = int(is_leaf.sum()); weight[is_leaf != 0] = rng.uniform(0.3, 9.0, nl)
    return dict(k=k, L=L, parent=parent, is_leaf=is_leaf, desc=desc, weight=weight)


def vocabulary_to_text(voc, path, scoring=0, weighting=0):
    """saveToTextFile (TemplatedVocabulary.h:1451-1480): 'k L  scoring weighting' then 'parent leaf b0 .. b31 weight' per node."""
    with open(path, "w") as f:
        f.write("%d %d  %d %d\n" % (voc["k"], voc["L"], scoring, weighting))
        for i in range(len(voc["parent"])):
            f.write("%d %d %s %r\n" % (voc["parent"][i], voc["is_leaf"][i], " ".join(str(int(b)) for b in voc["desc"][i]), float(voc["weight"][i])))


def vocabulary_from_text(path):
    """loadFromTextFile (TemplatedVocabulary.h:1362-1446) -> the column form."""
    with open(path) as f:
        k, L, n1, n2 = (int(x) for x in f.readline().split())
        rows = [ln.split() for ln in f if ln.strip()]
    parent = np.array([int(r[0]) for r in rows], np.int32)
    is_leaf = np.array([1 if int(r[1]) > 0 else 0 for r in rows], np.uint8)
    desc = np.array([[int(x) for x in r[2:34]] for r in rows], np.uint8).reshape(-1, 32)
    weight = np.array([float(r[34]) for r in rows])
    return dict(k=k, L=L, parent=parent, is_leaf=is_leaf, desc=desc, weight=weight, scoring=n1, weighting=n2)


def descriptors_near_words(voc, n, seed=3, flip=10):
    """n descriptors: noisy copies of random leaf descriptors (so the descent lands in populated parts of the tree)."""
    rng = np.random.default_rng(seed)
    leaves = np.nonzero(voc["is_leaf"])[0]
    d = voc["desc"][leaves[rng.integers(0, len(leaves), n)]].copy()
    return noisy_copy(d, flip_bits=flip, seed=seed + 1)


# ----------------------------------------------------------------------------- isInFrustum
def keyframe_database(n_db=400, n_words=4000, words_per_kf=300, n_places=25, seed=9):
    """Synthetic KeyFrameDatabase content (src/KeyFrameDatabase.cc): n_db key frames visiting n_places places in a loop; the BowVector of
    a key frame = most of its place's words + some of its own (L1-normalised tf-idf-like values, ascending word ids), so that
    revisits share many words. Returns dict(db=[(word, val)], place, covis=[neighbour entry ids, best-10 style], queries=[(word, val,
    place)])."""
    rng = np.random.default_rng(seed)
    place_words = [np.sort(rng.choice(n_words, words_per_kf, replace=False)) for _ in range(n_places)]

    def bow_of(place):
        keep = place_words[place][rng.random(words_per_kf) < 0.7]
        own = rng.choice(n_words, words_per_kf // 4, replace=False)
        w = np.unique(np.concatenate([keep, own])).astype(np.int32)
        v = rng.uniform(0.2, 3.0, len(w))
        return w, (v / v.sum()).astype(np.float64)
    place = (np.arange(n_db) * n_places * 2 // n_db) % n_places            # two laps
    db = [bow_of(int(pl)) for pl in place]
    covis = []
    for k in range(n_db):
        near = [j for j in range(max(0, k - 7), min(n_db, k + 8)) if j != k]
        covis.append([int(j) for j in rng.permutation(near)[:10]])
    queries = [bow_of(int(pl)) + (int(pl),) for pl in rng.integers(0, n_places, 12)]
    return dict(db=db, place=place, covis=covis, queries=queries)


def frustum_problem(n_points=3000, seed=4, n_cams=2):
    """Synthetic input of Frame::isInFrustum (Frame.cc:244-312) as called by Tracking::SearchLocalPoints (Tracking.cc:1617-1680):
    a frame pose, the rig's cameras (Tsw = Tsc * Tcw and the camera centres, formed in float32 like the caller's cv::Mat code),
    and local map points all around the rig -- in front of cam0, only visible to cam1 (it looks ~69 deg to the side), behind
    both, outside the image bounds, outside the scale-invariance distances, seen from a bad angle, and looked at head-on
    (viewCos > 0.998 -> the narrow window). Returns (frame dict, points dict)."""
    rng = np.random.default_rng(seed)
    T0, T1 = rig_extrinsics_f32()
    ext = [T0, T1][:n_cams]
    rv = rng.normal(0, 0.2, 3)
    Tcw = np.eye(4, dtype=np.float32)
    Tcw[:3, :3] = _rodrigues(rv).astype(np.float32)
    Tcw[:3, 3] = rng.normal(0, 0.5, 3).astype(np.float32)
    Rsw, tsw, Ow = [], [], []
    for T in ext:
        Tsw = (T.astype(np.float32) @ Tcw).astype(np.float32)
        R, t = Tsw[:3, :3], Tsw[:3, 3]
        Rsw.append(R.reshape(9)); tsw.append(t); Ow.append((-(R.T @ t)).astype(np.float32))
    scale = np.ones(8, np.float32)
    for i in range(1, 8):
        scale[i] = np.float32(np.float64(scale[i - 1]) * np.float64(np.float32(1.2)))
    k = [RIG["cam0"], RIG["cam1"]][:n_cams]
    frame = dict(Rsw=np.array(Rsw, np.float32), tsw=np.array(tsw, np.float32), Ow=np.array(Ow, np.float32),
                 fx=np.array([c["fx"] for c in k], np.float32), fy=np.array([c["fy"] for c in k], np.float32),
                 cx=np.array([c["cx"] for c in k], np.float32), cy=np.array([c["cy"] for c in k], np.float32),
                 min_x=np.full(n_cams, -2.5, np.float32), max_x=np.full(n_cams, 642.0, np.float32),
                 min_y=np.full(n_cams, -1.5, np.float32), max_y=np.full(n_cams, 481.0, np.float32),
                 log_scale_factor=np.float32(np.log(np.float32(1.2))), scale_factors=scale)
    # points on a shell around the rig centre (all directions), 1..12 m
    d = rng.normal(0, 1, (n_points, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    dist = rng.uniform(1.0, 12.0, n_points)
    centre = Ow[0].astype(np.float64)
    pos = (centre + d * dist[:, None]).astype(np.float32)
    # mean viewing direction: from the observing keyframes towards the point; mostly aligned with the current ray, some oblique,
    # some exactly head-on
    ray = pos.astype(np.float64) - centre; ray /= np.linalg.norm(ray, axis=1, keepdims=True)
    jitter = rng.normal(0, 1, (n_points, 3)) * rng.choice([0.0, 0.05, 0.4, 1.5], n_points, p=[0.15, 0.45, 0.3, 0.1])[:, None]
    nrm = ray + jitter; nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    # scale-invariance range: the point was created at distance d0 on level l0: max = d0 * s[l0], min = max / s[7]
    d0 = dist * rng.uniform(0.4, 2.5, n_points)
    l0 = rng.integers(0, 8, n_points)
    max_dist = (d0 * scale[l0]).astype(np.float32)
    min_dist = (max_dist / scale[7]).astype(np.float32)
    cand = (rng.random(n_points) < 0.9).astype(np.uint8)
    return frame, dict(pos=pos, normal=nrm.astype(np.float32), min_dist=min_dist, max_dist=max_dist, candidate=cand)


def tracking_problem(n_frames=6, n_points=1500, n_features=1000, seed=17, th=1.0, pre_matched=0.25):
    """Synthetic input of the Tracking thread's steady-state chain -- SearchLocalPoints (Tracking.cc:1617-1680) + PoseOptimization
    (:1321) -- for a batch of frames, laid out like dcs_track_frame: per frame a rig pose guess (motion model: perturbed truth) with the
    view matrices the reference forms in float (Tsw = Tsc * Tcw, camera centres), a local map (points all around the rig: visible to
    cam0, only to cam1, behind, out of range, oblique), and frame features = noisy projections of a subset of the visible map points
    (+ distractor features); `pre_matched` of the features that stem from a map point already hold it (from TrackWithMotionModel:
    has_point / taken, point_xw), and their map points are no candidates. The grid is NOT included (dcs_frame_grid / oracle.frame_grid)."""
    rng = np.random.default_rng(seed)
    T0, T1 = rig_extrinsics_f32()
    ext = [T0, T1]
    kk = [RIG["cam0"], RIG["cam1"]]
    cams = []
    for c, T in enumerate(ext):
        adj, ext7 = rig_adjoint_f32(T, False)
        cams.append(dict(fx=float(np.float32(kk[c]["fx"])), fy=float(np.float32(kk[c]["fy"])), cx=float(np.float32(kk[c]["cx"])),
                         cy=float(np.float32(kk[c]["cy"])), ext7=ext7, adj=adj))
    scale = np.ones(8, np.float32)
    for i in range(1, 8):
        scale[i] = np.float32(np.float64(scale[i - 1]) * np.float64(np.float32(1.2)))
    inv_sigma2 = (np.float32(1.0) / (scale * scale)).astype(np.float32)
    min_x, max_x, min_y, max_y = np.float32(-2.5), np.float32(642.0), np.float32(-1.5), np.float32(481.0)
    frames = []
    for f in range(n_frames):
        # ground truth and guess (float 4 x 4 like mTcw)
        rv = rng.normal(0, 0.2, 3)
        Tgt = np.eye(4, dtype=np.float32)
        Tgt[:3, :3] = _rodrigues(rv).astype(np.float32); Tgt[:3, 3] = rng.normal(0, 0.5, 3).astype(np.float32)
        Tg = np.eye(4, dtype=np.float32)
        Tg[:3, :3] = (_rodrigues(rng.normal(0, 0.004, 3)) @ Tgt[:3, :3].astype(np.float64)).astype(np.float32)
        Tg[:3, 3] = (Tgt[:3, 3] + rng.normal(0, 0.01, 3)).astype(np.float32)
        Rsw, tsw, Ow, Tsw_gt = [], [], [], []
        for T in ext:
            Tsw = (T.astype(np.float32) @ Tg).astype(np.float32)
            R, t = Tsw[:3, :3], Tsw[:3, 3]
            Rsw.append(R.reshape(9)); tsw.append(t); Ow.append((-(R.T @ t)).astype(np.float32))
            Tsw_gt.append(T.astype(np.float64) @ Tgt.astype(np.float64))
        view = dict(Rsw=np.array(Rsw, np.float32), tsw=np.array(tsw, np.float32), Ow=np.array(Ow, np.float32),
                    fx=np.array([c["fx"] for c in cams], np.float32), fy=np.array([c["fy"] for c in cams], np.float32),
                    cx=np.array([c["cx"] for c in cams], np.float32), cy=np.array([c["cy"] for c in cams], np.float32),
                    min_x=np.full(2, min_x, np.float32), max_x=np.full(2, max_x, np.float32), min_y=np.full(2, min_y, np.float32),
                    max_y=np.full(2, max_y, np.float32), log_scale_factor=np.float32(np.log(np.float32(1.2))), scale_factors=scale)
        # local map: a shell around the rig centre, biased towards the two viewing directions
        centre = (-(Tgt[:3, :3].astype(np.float64).T @ Tgt[:3, 3].astype(np.float64)))
        d = rng.normal(0, 1, (n_points, 3))
        for c in (0, 1):                                           # half of the points in front of each camera
            sel = slice(c * (n_points // 3), (c + 1) * (n_points // 3))
            zc = Tsw_gt[c][2, :3]
            d[sel] = zc + rng.normal(0, 0.35, (n_points // 3, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        dist = rng.uniform(1.5, 10.0, n_points)
        pos = (centre + d * dist[:, None]).astype(np.float32)
        ray = pos.astype(np.float64) - centre; ray /= np.linalg.norm(ray, axis=1, keepdims=True)
        nrm = ray + rng.normal(0, 1, (n_points, 3)) * rng.choice([0.0, 0.05, 0.4, 1.5], n_points, p=[0.1, 0.6, 0.25, 0.05])[:, None]
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        l0 = rng.integers(0, 6, n_points)
        max_dist = (dist * rng.uniform(0.9, 1.6, n_points) * scale[l0]).astype(np.float32)
        min_dist = (max_dist / scale[7]).astype(np.float32)
        pdesc = random_descriptors(n_points, seed=seed * 100 + f)
        # features: project with the TRUE pose, keep what lands in an image, noisy pixel + octave
        feats = []                                                  # (cam, u, v, octave, point)
        for p in range(n_points):
            for c in (0, 1):
                Pc = Tsw_gt[c][:3, :3] @ pos[p].astype(np.float64) + Tsw_gt[c][:3, 3]
                if Pc[2] <= 0.2: continue
                u = cams[c]["fx"] * Pc[0] / Pc[2] + cams[c]["cx"]; v = cams[c]["fy"] * Pc[1] / Pc[2] + cams[c]["cy"]
                if 5 < u < 635 and 5 < v < 475:
                    lvl = int(np.clip(np.ceil(np.log(max_dist[p] / np.linalg.norm(pos[p].astype(np.float64) - Ow[c].astype(np.float64))) / np.log(1.2)), 0, 7))
                    feats.append((c, u, v, lvl, p)); break
        order = rng.permutation(len(feats))[:int(n_features * 0.8)]
        feats = [feats[i] for i in order]
        n_dis = n_features - len(feats)
        per_cam = [[ft for ft in feats if ft[0] == c] for c in (0, 1)]
        kp_x, kp_y, octv, fdesc, src, cam_off = [], [], [], [], [], [0]
        for c in (0, 1):
            k_dis = n_dis // 2 if c == 0 else n_dis - n_dis // 2
            ent = [(u + rng.normal(0, 0.8 * float(scale[l])), v + rng.normal(0, 0.8 * float(scale[l])), int(np.clip(l + rng.integers(-1, 2), 0, 7)), p) for (_, u, v, l, p) in per_cam[c]]
            ent += [(rng.uniform(0, 640), rng.uniform(0, 480), int(min(rng.geometric(0.35) - 1, 7)), -1) for _ in range(k_dis)]
            for i in rng.permutation(len(ent)):
                u, v, l, p = ent[i]
                kp_x.append(u); kp_y.append(v); octv.append(l); src.append(p)
            cam_off.append(len(kp_x))
        N = len(kp_x)
        src = np.asarray(src, np.int64)
        fdesc = random_descriptors(N, seed=seed * 100 + 50 + f)
        from_map = src >= 0
        fdesc[from_map] = noisy_copy(pdesc[src[from_map]], flip_bits=16, seed=seed * 100 + 70 + f)
        has_point = (from_map & (rng.random(N) < pre_matched)).astype(np.uint8)
        taken = (has_point.astype(bool) & (rng.random(N) < 0.9)).astype(np.uint8)          # a few hold a point without observations: searchable, and the search may replace it
        point_xw = np.zeros((N, 3), np.float32)
        point_xw[has_point != 0] = (pos[src[has_point != 0]].astype(np.float64) + rng.normal(0, 0.01, (int(has_point.sum()), 3))).astype(np.float32)
        candidate = np.ones(n_points, np.uint8)
        candidate[src[taken != 0]] = 0                                                     # already matched in this frame (Tracking.cc:1625-1640)
        candidate[rng.random(n_points) < 0.03] = 0                                         # bad points
        features = dict(cam_off=np.asarray(cam_off, np.int32), kp_x=np.asarray(kp_x, np.float32), kp_y=np.asarray(kp_y, np.float32),
                        kp_octave=np.asarray(octv, np.int32), kp_angle=np.zeros(N, np.float32), desc=fdesc, taken=taken,
                        min_x=np.full(2, min_x, np.float32), min_y=np.full(2, min_y, np.float32),
                        grid_w_inv=np.full(2, np.float32(64) / np.float32(max_x - min_x), np.float32),
                        grid_h_inv=np.full(2, np.float32(48) / np.float32(max_y - min_y), np.float32))
        q = _quat_from_R(Tg[:3, :3].astype(np.float64))
        pose = np.concatenate([Tg[:3, 3].astype(np.float64), q])
        frames.append(dict(features=features, has_point=has_point, point_xw=point_xw, view=view, pose=pose, Tcw_guess=Tg, Tcw_gt=Tgt,
                           points=dict(pos=pos, normal=nrm.astype(np.float32), min_dist=min_dist, max_dist=max_dist, candidate=candidate), desc=pdesc,
                           feature_source=src))
    params = dict(viewing_cos_limit=0.5, th=float(th), th_high=100, nn_ratio=0.8, inv_level_sigma2=inv_sigma2, cams=cams,
                  huber_delta=float(np.float32(np.sqrt(5.991))), chi2_th=[float(np.float32(5.991))] * 4, its=[10, 10, 10, 10])
    return frames, params


def motion_model_problem(n_frames=3, n_points=1200, n_features=900, seed=31, th=7.0, seen=0.85, lost=0.1):
    """Input of Tracking::TrackWithMotionModel's search (src/Tracking.cc:1384-1427): tracking_problem's frames with nothing held yet
    (mvpMapPoints is cleared, :1396), key-point angles, and per frame the LAST frame's features that hold a good map point as queries
    (fr["mm"]: pos = GetWorldPos, desc = the map point's descriptor, q_cam = keypointToCam, q_octave / q_angle = the last key point), in the
    last frame's feature order (camera-major). `seen` of this frame's map-born features were also seen by the last frame (octave -+ 1,
    angle = this frame's + a common rotation + noise, a tenth of them arbitrary); `lost` adds map points this frame does not see."""
    frames, prm = tracking_problem(n_frames=n_frames, n_points=n_points, n_features=n_features, seed=seed, pre_matched=0.0)
    rng = np.random.default_rng(seed + 4099)
    for fr in frames:
        ft, src = fr["features"], fr["feature_source"]
        N = len(src)
        ft["kp_angle"] = rng.uniform(0, 360, N).astype(np.float32)
        ft["taken"] = np.zeros(N, np.uint8); fr["has_point"] = np.zeros(N, np.uint8); fr["point_xw"] = np.zeros((N, 3), np.float32)
        cam_of = (np.searchsorted(ft["cam_off"], np.arange(N), side="right") - 1).astype(np.int32)
        born = np.nonzero(src >= 0)[0]
        keep = born[rng.random(len(born)) < seen]
        rot = float(rng.uniform(0, 30))
        ang = (ft["kp_angle"][keep].astype(np.float64) + rot + rng.normal(0, 2.0, len(keep))) % 360.0
        wild = rng.random(len(keep)) < 0.1
        ang[wild] = rng.uniform(0, 360, int(wild.sum()))
        pts = src[keep]
        qc, qo = cam_of[keep], np.clip(ft["kp_octave"][keep] + rng.integers(-1, 2, len(keep)), 0, 7).astype(np.int32)
        n_lost = int(lost * len(keep))
        others = rng.choice(len(fr["points"]["pos"]), n_lost, replace=False) if n_lost else np.zeros(0, np.int64)
        pts = np.concatenate([pts, others]); qc = np.concatenate([qc, rng.integers(0, 2, n_lost).astype(np.int32)])
        qo = np.concatenate([qo, rng.integers(0, 8, n_lost).astype(np.int32)]); ang = np.concatenate([ang, rng.uniform(0, 360, n_lost)])
        truth = np.concatenate([keep, np.full(n_lost, -1)])
        order = np.lexsort((rng.random(len(pts)), qc))                  # the last frame's features: camera-major, arbitrary inside a camera
        fr["mm"] = dict(pos=fr["points"]["pos"][pts][order].astype(np.float32), desc=fr["desc"][pts][order], q_cam=qc[order].astype(np.int32),
                        q_octave=qo[order].astype(np.int32), q_angle=ang[order].astype(np.float32), truth_feature=truth[order])
    prm = dict(prm)
    prm["th"] = float(th); prm["nn_ratio"] = 0.0
    return frames, prm


def scene_from_features(kp_per_cam, desc_per_cam, seed=91, seen=0.8, bounds=None):
    """A map that explains a dual frame's REAL extracted features (bench.py's per_frame_total leg: the chain is fed from the extractor's slots, so
    the map has to fit what the extractor found): every key point of camera c is back-projected to a random depth with a true rig pose ->
    a map point (position + noise, normal towards the camera, scale-invariance distances from the key point's octave, descriptor = the
    feature's with a few flipped bits). Returns (frame pieces, params): view / pose of a perturbed pose guess, points (the local map of
    TrackLocalMap), desc, mm (the last frame's share of them for TrackWithMotionModel: camera, octave, angle of the key point), cams."""
    rng = np.random.default_rng(seed)
    _, prm = tracking_problem(n_frames=1, n_points=30, n_features=20, seed=1)
    cams, scale = prm["cams"], None
    scale = np.ones(8, np.float32)
    for i in range(1, 8):
        scale[i] = np.float32(np.float64(scale[i - 1]) * np.float64(np.float32(1.2)))
    T0, T1 = rig_extrinsics_f32()
    ext = [T0, T1]
    Tgt = np.eye(4, dtype=np.float32)
    Tgt[:3, :3] = _rodrigues(rng.normal(0, 0.1, 3)).astype(np.float32); Tgt[:3, 3] = rng.normal(0, 0.3, 3).astype(np.float32)
    Tg = np.eye(4, dtype=np.float32)
    Tg[:3, :3] = (_rodrigues(rng.normal(0, 0.003, 3)) @ Tgt[:3, :3].astype(np.float64)).astype(np.float32)
    Tg[:3, 3] = (Tgt[:3, 3] + rng.normal(0, 0.008, 3)).astype(np.float32)
    # image bounds per camera (Frame::ComputeImageBounds): the image itself, or what the caller derived from the undistorted corners
    if bounds is None:
        bounds = (np.zeros(2, np.float32), np.full(2, 640.0, np.float32), np.zeros(2, np.float32), np.full(2, 480.0, np.float32))
    bmin_x, bmax_x, bmin_y, bmax_y = (np.asarray(v, np.float32) for v in bounds)
    Rsw, tsw, Ow = [], [], []
    for T in ext:
        Tsw = (T.astype(np.float32) @ Tg).astype(np.float32)
        R, t = Tsw[:3, :3], Tsw[:3, 3]
        Rsw.append(R.reshape(9)); tsw.append(t); Ow.append((-(R.T @ t)).astype(np.float32))
    view = dict(Rsw=np.array(Rsw, np.float32), tsw=np.array(tsw, np.float32), Ow=np.array(Ow, np.float32),
                fx=np.array([c["fx"] for c in cams], np.float32), fy=np.array([c["fy"] for c in cams], np.float32),
                cx=np.array([c["cx"] for c in cams], np.float32), cy=np.array([c["cy"] for c in cams], np.float32),
                min_x=bmin_x, max_x=bmax_x, min_y=bmin_y, max_y=bmax_y, log_scale_factor=np.float32(np.log(np.float32(1.2))), scale_factors=scale)
    pos, nrm, mind, maxd, pdesc, qc, qo, qa = [], [], [], [], [], [], [], []
    for c in (0, 1):
        kp, de = kp_per_cam[c], desc_per_cam[c]
        Tsw_gt = ext[c].astype(np.float64) @ Tgt.astype(np.float64)
        Rg, tg = Tsw_gt[:3, :3], Tsw_gt[:3, 3]
        centre = -(Rg.T @ tg)
        d = rng.uniform(2.0, 8.0, len(kp))
        lvl = kp["octave"].astype(np.int64)
        pc = np.stack([(kp["x"].astype(np.float64) - cams[c]["cx"]) / cams[c]["fx"] * d, (kp["y"].astype(np.float64) - cams[c]["cy"]) / cams[c]["fy"] * d, d], 1)
        pw = (pc - tg) @ Rg
        pos.append((pw + rng.normal(0, 0.004, pw.shape)).astype(np.float32))
        ray = pw - centre; ray /= np.linalg.norm(ray, axis=1, keepdims=True)
        nrm.append(ray.astype(np.float32))
        md = (d * scale[np.clip(lvl, 0, 7)] * 1.05).astype(np.float32)
        maxd.append(md); mind.append((md / scale[7]).astype(np.float32))
        pdesc.append(noisy_copy(de, flip_bits=8, seed=seed * 10 + c))
        qc.append(np.full(len(kp), c, np.int32)); qo.append(kp["octave"].astype(np.int32)); qa.append(kp["angle"].astype(np.float32))
    pos, nrm, mind, maxd, pdesc = np.concatenate(pos), np.concatenate(nrm), np.concatenate(mind), np.concatenate(maxd), np.concatenate(pdesc)
    qc, qo, qa = np.concatenate(qc), np.concatenate(qo), np.concatenate(qa)
    sel = np.nonzero(rng.random(len(pos)) < seen)[0]                   # ascending = the last frame's feature order (camera-major)
    q = _quat_from_R(Tg[:3, :3].astype(np.float64))
    frame = dict(view=view, pose=np.concatenate([Tg[:3, 3].astype(np.float64), q]), points=dict(pos=pos, normal=nrm, min_dist=mind, max_dist=maxd), desc=pdesc,
                 mm=dict(pos=pos[sel], desc=pdesc[sel], q_cam=qc[sel], q_octave=qo[sel], q_angle=qa[sel], point=sel),
                 grid=dict(min_x=bmin_x, min_y=bmin_y, grid_w_inv=(np.float32(64) / (bmax_x - bmin_x)).astype(np.float32),
                           grid_h_inv=(np.float32(48) / (bmax_y - bmin_y)).astype(np.float32)))
    return frame, prm

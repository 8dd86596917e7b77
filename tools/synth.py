"""Deterministic synthetic inputs for the hot path.

There is no KITTI data and no network here or on the GPU box, so every test and bench input is
generated: "KITTI-00-shaped" means the intrinsics, image size and stereo baseline of the reference's
config/kitti_00.yaml:3-26 (fx=fy=718.856, cx=607.1928, cy=185.2157, 1241x376, bf=386.1448).

Nothing in this module touches the GPU, the oracle or the reference.
"""
from __future__ import annotations

import numpy as np

KITTI_K = (718.856, 718.856, 607.1928, 185.2157)   # fx fy cx cy  (kitti_00.yaml:3-6)
KITTI_BF = 386.1448                                  # kitti_00.yaml:26
KITTI_W, KITTI_H = 1241, 376                         # kitti_00.yaml:23-24
KITTI_BASELINE = KITTI_BF / KITTI_K[0]               # system.cpp:69-70

IDENT_POSE = np.array([0, 0, 0, 1, 0, 0, 0], dtype=np.float64)  # qx qy qz qw tx ty tz


def quat_mul(a, b):
    """Hamilton product, (x,y,z,w) order."""
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by + ay * bw + az * bx - ax * bz,
        aw * bz + az * bw + ax * by - ay * bx,
        aw * bw - ax * bx - ay * by - az * bz])


def quat_rot(q, p):
    x, y, z, w = q
    R = np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return R @ p


def small_rot_quat(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.array([0.5 * w[0], 0.5 * w[1], 0.5 * w[2], 1.0])
    s = np.sin(th / 2) / th
    return np.array([s * w[0], s * w[1], s * w[2], np.cos(th / 2)])


def stereo_cam_ext(baseline=KITTI_BASELINE):
    """cam_ext[2x7]: left = identity, right = (I, (-baseline,0,0))  (system.cpp:63,71)."""
    ext = np.zeros((2, 7))
    ext[:, 3] = 1.0
    ext[1, 4] = -baseline
    return ext


def make_ba_problem(P=10, L=4000, obs_per_lm=5, seed=1, frac_fixed=0.15, frac_gross=0.03,
                    pix_sigma=0.5, gross_sigma=30.0, pose_t_noise=0.02, pose_r_noise=0.002,
                    point_noise=0.0, loop=False, fix_first_pose=False, K=KITTI_K, wrap=False, shuffle_poses=False, uv_f32=False):
    """Synthetic BA graph of SURVEY.md section 8-D.

    C3 (local BA): P=10, L=4000, obs_per_lm=5 -> E=20000, 15 % fixed landmarks, no pose fixed
    (as Backend::OptimizeActiveMap, backend.cpp:93-103).  C4 (global BA): P=500, L=80000,
    obs_per_lm=6, loop=True, fix_first_pose=True.

    wrap=True (with loop=True): landmarks near the end of the loop are also seen by the first keyframes -- a closed
    loop, the co-visibility band wraps around.  shuffle_poses=True renumbers the keyframes at random (same graph, no
    band structure left in the pose order).

    uv_f32=True: the measurements are float values widened to double, as the reference's are (cv::KeyPoint::pt is a
    Point2f; backend.cpp:126-160 hands them to g2o as Vector2d) -- the library then sends them across PCIe as floats.

    Returns a dict of flat arrays in the layout of ssx_ba_problem (include/ssx.h).
    """
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = K
    # ground-truth poses T_cw (world -> camera)
    gt = np.zeros((P, 7))
    centers = np.zeros((P, 3))
    yaw = np.zeros(P)
    if loop:
        R = 400.0 / (2 * np.pi)
        for i in range(P):
            a = 2 * np.pi * i / P
            centers[i] = (R * np.sin(a), 0.0, R * (1 - np.cos(a)))
            yaw[i] = a   # heading rotates about y
    else:
        for i in range(P):
            centers[i] = (0.0, 0.0, 0.8 * i)
    for i in range(P):
        # R_wc = rot_y(yaw): camera z axis points along heading
        q_wc = np.array([0.0, np.sin(yaw[i] / 2), 0.0, np.cos(yaw[i] / 2)])
        q_cw = q_wc * np.array([-1, -1, -1, 1])
        gt[i, :4] = q_cw
        gt[i, 4:] = -quat_rot(q_cw, centers[i])

    first = rng.integers(0, P if wrap else max(P - obs_per_lm + 1, 1), size=L)
    local = np.stack([rng.uniform(-15, 15, L), rng.uniform(-3, 3, L), rng.uniform(6, 46, L)], 1)
    pts = np.zeros((L, 3))
    for j in range(L):
        i0 = first[j]
        q_cw = gt[i0, :4]
        q_wc = q_cw * np.array([-1, -1, -1, 1])
        pts[j] = quat_rot(q_wc, local[j]) + centers[i0]

    k = min(obs_per_lm, P)
    E = L * k
    edge_pose = np.zeros(E, dtype=np.int32)
    edge_point = np.zeros(E, dtype=np.int32)
    edge_uv = np.zeros((E, 2))
    e = 0
    for j in range(L):
        for d in range(k):
            i = (first[j] + d) % P
            pc = quat_rot(gt[i, :4], pts[j]) + gt[i, 4:]
            edge_pose[e] = i
            edge_point[e] = j
            edge_uv[e] = (fx * pc[0] / pc[2] + cx, fy * pc[1] / pc[2] + cy)
            e += 1
    edge_uv += rng.normal(0, pix_sigma, edge_uv.shape)
    gross = rng.random(E) < frac_gross
    edge_uv[gross] += rng.normal(0, gross_sigma, (int(gross.sum()), 2))
    if uv_f32:
        edge_uv = edge_uv.astype(np.float32).astype(np.float64)

    poses = gt.copy()
    for i in range(P):
        if fix_first_pose and i == 0:
            continue
        dq = small_rot_quat(rng.uniform(-pose_r_noise, pose_r_noise, 3))
        q = quat_mul(dq, poses[i, :4])
        poses[i, :4] = q / np.linalg.norm(q)
        poses[i, 4:] = quat_rot(dq, poses[i, 4:]) + rng.uniform(-pose_t_noise, pose_t_noise, 3)
    points = pts + (rng.normal(0, point_noise, pts.shape) if point_noise > 0 else 0.0)
    point_fixed = (rng.random(L) < frac_fixed).astype(np.uint8)
    pose_fixed = np.zeros(P, dtype=np.uint8)
    if fix_first_pose:
        pose_fixed[0] = 1
    if shuffle_poses:
        perm = np.random.default_rng(seed + 7919).permutation(P)       # new index of old keyframe i
        inv = np.argsort(perm)
        poses, gt, pose_fixed = poses[inv], gt[inv], pose_fixed[inv]
        edge_pose = perm[edge_pose].astype(np.int32)
    return dict(P=P, L=L, E=E, poses=np.ascontiguousarray(poses), pose_fixed=pose_fixed,
                points=np.ascontiguousarray(points), point_fixed=point_fixed,
                edge_pose=edge_pose, edge_point=edge_point, edge_uv=np.ascontiguousarray(edge_uv),
                edge_cam=np.zeros(E, dtype=np.uint8), K=np.array(K, dtype=np.float64),
                cam_ext=stereo_cam_ext(), gt_poses=gt, gt_points=pts)


def make_pose_only_problem(M=200, seed=3, frac_gross=0.1, K=KITTI_K):
    """One frame of FrontEnd::EstimateCurrentPose (frontend.cpp:184-300): M map points, noisy pose."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = K
    dq = small_rot_quat(rng.uniform(-0.02, 0.02, 3))
    gt = np.concatenate([dq / np.linalg.norm(dq), rng.uniform(-0.3, 0.3, 3)])
    xyz = np.stack([rng.uniform(-12, 12, M), rng.uniform(-3, 3, M), rng.uniform(5, 45, M)], 1)
    uv = np.zeros((M, 2))
    for i in range(M):
        pc = quat_rot(gt[:4], xyz[i]) + gt[4:]
        uv[i] = (fx * pc[0] / pc[2] + cx, fy * pc[1] / pc[2] + cy)
    uv += rng.normal(0, 0.5, uv.shape)
    gross = rng.random(M) < frac_gross
    uv[gross] += rng.normal(0, 25.0, (int(gross.sum()), 2))
    # KeyPoint coordinates are float32 in the reference (cv::Point2f -> double, frontend.cpp:214,222)
    uv = uv.astype(np.float32).astype(np.float64)
    init = IDENT_POSE.copy()
    return dict(M=M, pose=init, gt_pose=gt, K=np.array(K), xyz=np.ascontiguousarray(xyz),
                uv=np.ascontiguousarray(uv))


def pose_mul(a, b):
    """SE3 product of two (qx qy qz qw tx ty tz) poses."""
    q = quat_mul(a[:4], b[:4])
    return np.concatenate([q / np.linalg.norm(q), a[4:] + quat_rot(a[:4], b[4:])])


def pose_inv(a):
    qi = np.array([-a[0], -a[1], -a[2], a[3]])
    return np.concatenate([qi, -quat_rot(qi, a[4:])])


def make_pose_graph_problem(P=60, n_loops=2, seed=11, n_active=7, drift=0.02, meas_noise=0.002):
    """Pose graph of LoopClosing::PoseGraphOptimization (loopclosing.cpp:458-539): P keyframes along a closed
    circuit (T_cw poses), one temporal edge per keyframe (to its predecessor) + n_loops loop edges from the newest
    keyframes to old ones; measurement = T_i * T_j^-1 with a little noise; the initial estimate accumulates drift.
    Fixed as in the reference: keyframe 0, the last n_active ("active") keyframes and the loop keyframes."""
    rng = np.random.default_rng(seed)
    gt = []
    for i in range(P):
        ang = 2 * np.pi * i / P
        # camera moving on a circle of radius 40 m, looking along the tangent: T_wc then inverted to T_cw
        q_wc = small_rot_quat(np.array([0.0, ang, 0.0]))
        t_wc = np.array([40 * np.sin(ang), 0.1 * np.sin(3 * ang), 40 * (1 - np.cos(ang))])
        gt.append(pose_inv(np.concatenate([q_wc / np.linalg.norm(q_wc), t_wc])))
    gt = np.array(gt)
    ei, ej, meas = [], [], []

    def add_edge(i, j):
        m = pose_mul(gt[i], pose_inv(gt[j]))
        n = np.concatenate([small_rot_quat(rng.normal(0, meas_noise, 3)), rng.normal(0, 5 * meas_noise, 3)])
        n[:4] /= np.linalg.norm(n[:4])
        ei.append(i); ej.append(j); meas.append(pose_mul(n, m))
    for i in range(1, P):
        add_edge(i, i - 1)
    loops = []
    for k in range(n_loops):
        i, j = P - 1 - k, (k * 3) % max(P // 4, 1)
        add_edge(i, j)
        loops.append(j)
    # initial estimate: integrate the measurements with drift (the loop does not close)
    init = [gt[0].copy()]
    for i in range(1, P):
        d = np.concatenate([small_rot_quat(rng.normal(0, drift * 0.1, 3)), rng.normal(0, drift, 3)])
        d[:4] /= np.linalg.norm(d[:4])
        init.append(pose_mul(pose_mul(d, meas[i - 1]), init[i - 1]))
    init = np.array(init)
    fixed = np.zeros(P, np.uint8)
    fixed[0] = 1
    fixed[P - n_active:] = 1
    for j in loops:
        fixed[j] = 1
    # the fixed keyframes sit at their true poses (the reference corrects the active window before this step)
    init[fixed > 0] = gt[fixed > 0]
    return dict(P=P, E=len(ei), poses=np.ascontiguousarray(init), fixed=fixed, ei=np.array(ei, np.int32),
                ej=np.array(ej, np.int32), meas=np.ascontiguousarray(np.array(meas)), gt_poses=gt)


# ----------------------------------------------------------------------------------------------
# synthetic stereo images: textured left image + piecewise-planar disparity
# ----------------------------------------------------------------------------------------------
def _smooth_noise(rng, h, w, cell):
    gh, gw = h // cell + 2, w // cell + 2
    g = rng.random((gh, gw)).astype(np.float32)
    ys = np.arange(h, dtype=np.float32) / cell
    xs = np.arange(w, dtype=np.float32) / cell
    y0 = ys.astype(np.int32); x0 = xs.astype(np.int32)
    fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def make_stereo_pair(seed=0, h=KITTI_H, w=KITTI_W, n_blobs=4000, bf=KITTI_BF):
    """Left image = band-limited noise octaves + random bright/dark blobs and L-corners;
    right image = left warped by a piecewise-planar disparity map d = bf/z, z in [4,80] m
    (d in [4.8,96.5] px), zero vertical disparity, + N(0,2) noise.  Returns (left,right,disp)."""
    rng = np.random.default_rng(1000 + seed)
    img = np.zeros((h, w), dtype=np.float32)
    for cell, amp in ((64, 60.0), (16, 40.0), (4, 24.0)):
        img += amp * (_smooth_noise(rng, h, w, cell) - 0.5)
    img += 128.0
    bx = rng.integers(4, w - 12, n_blobs); by = rng.integers(4, h - 12, n_blobs)
    bs = rng.integers(3, 9, n_blobs); bv = rng.choice([-70.0, 70.0, -110.0, 110.0], n_blobs)
    kind = rng.integers(0, 2, n_blobs)
    for x, y, s, v, k in zip(bx, by, bs, bv, kind):
        if k == 0:
            img[y:y + s, x:x + s] += v
        else:   # L-corner
            img[y:y + s, x:x + 2] += v
            img[y + s - 2:y + s, x:x + s] += v
    left = np.clip(img, 0, 255).astype(np.uint8)
    # piecewise-planar depth: vertical strips with linear depth ramps
    n_strip = 6
    edges = np.linspace(0, w, n_strip + 1).astype(int)
    z = np.zeros((h, w), dtype=np.float32)
    for s in range(n_strip):
        z0, z1 = rng.uniform(4, 80, 2)
        ramp = np.linspace(z0, z1, h, dtype=np.float32)[:, None]
        z[:, edges[s]:edges[s + 1]] = ramp
    disp = (bf / z).astype(np.float32)
    xs = np.arange(w, dtype=np.float32)[None, :] + disp      # right(u) = left(u + d)
    x0 = np.clip(np.floor(xs).astype(np.int32), 0, w - 1)
    x1 = np.clip(x0 + 1, 0, w - 1)
    fx = np.clip(xs - np.floor(xs), 0, 1)
    rows = np.arange(h)[:, None]
    lf = left.astype(np.float32)
    right = lf[rows, x0] * (1 - fx) + lf[rows, x1] * fx
    right += rng.normal(0, 2.0, right.shape).astype(np.float32)
    right = np.clip(np.rint(right), 0, 255).astype(np.uint8)
    return left, right, disp


def fast9_density(img, t=20):
    """Fraction of the interior pixels of `img` that are FAST-9 corners at threshold t (numpy, ~1 s per KITTI-sized image):
    what the bench quotes beside the front-end figures (make_stereo_pair: 12.8 % at the default 4000 blobs, 1.5 % at 400 --
    photographs of streets: 1-3 %)."""
    I = img.astype(np.int16)
    h, w = I.shape
    dx = [0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1]
    dy = [3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3]
    c = I[3:h - 3, 3:w - 3]
    ring = np.stack([I[3 + dy[k]:h - 3 + dy[k], 3 + dx[k]:w - 3 + dx[k]] for k in range(16)])

    def arc(m):
        m2 = np.concatenate([m, m[:8]])
        ok = np.zeros(m.shape[1:], bool)
        for s0 in range(16):
            ok |= m2[s0:s0 + 9].all(axis=0)
        return ok
    return float((arc(ring > c + t) | arc(ring < c - t)).mean())


def make_lateral_sequence(n_frames=12, step=0.12, seed=0, h=KITTI_H, w=KITTI_W, n_blobs=4000, bf=KITTI_BF, baseline=KITTI_BASELINE):
    """A stereo SEQUENCE of the make_stereo_pair scene seen from a rig that moves sideways (along +x, the direction of
    the right camera) by `step` baselines per frame: a camera at lateral offset a * baseline sees the texture shifted
    by a * disparity, so frame k is left = base(u + a_k d), right = base(u + (a_k + 1) d).  Returns
    (frames [(left, right)], T_cw [n,7] ground-truth poses with world = camera 0, disp)."""
    base, _, disp = make_stereo_pair(seed=seed, h=h, w=w, n_blobs=n_blobs, bf=bf)
    rng = np.random.default_rng(5000 + seed)
    lf = base.astype(np.float32)
    rows = np.arange(h)[:, None]

    def render(alpha):
        xs = np.arange(w, dtype=np.float32)[None, :] + np.float32(alpha) * disp
        x0 = np.clip(np.floor(xs).astype(np.int32), 0, w - 1)
        x1 = np.clip(x0 + 1, 0, w - 1)
        fx = np.clip(xs - np.floor(xs), 0, 1)
        img = lf[rows, x0] * (1 - fx) + lf[rows, x1] * fx
        img += rng.normal(0, 1.5, img.shape).astype(np.float32)
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)
    frames, poses = [], []
    for k in range(n_frames):
        a = step * k
        frames.append((render(a), render(a + 1.0)))
        poses.append(np.array([0, 0, 0, 1, -a * baseline, 0, 0], dtype=np.float64))       # T_cw: camera centre at (+a b, 0, 0)
    return frames, np.array(poses), disp


def _cell_hash(a, b, salt):
    """deterministic pseudo-random value in [0, 1) per integer cell (a, b)"""
    a = a.astype(np.int64) + 100003; b = b.astype(np.int64) + 100019          # keep the cell indices positive
    h = (a * 73856093) ^ (b * 19349663) ^ (np.asarray(salt, dtype=np.int64) * 83492791)
    h = (h ^ (h >> 13)) * 1274126177
    h = h ^ (h >> 16)
    return (h & 0xFFFF).astype(np.float32) / 65536.0


def _block_texture(a, b, salt):
    """piecewise-constant random blocks at three scales (0.2 m, 0.8 m, 3.2 m): block junctions are FAST corners at any
    viewing distance between a few and a few tens of metres"""
    t = np.zeros(a.shape, np.float32)
    for k, (size, wgt) in enumerate(((0.2, 0.45), (0.8, 0.35), (3.2, 0.2))):
        t += wgt * _cell_hash(np.floor(a / size), np.floor(b / size), salt * 7 + k)
    return t


# the keys System / FrontEnd / Backend read, with the values of the reference's config/kitti_00.yaml
KITTI00_SETTINGS = {
    "Camera1.fx": 718.856, "Camera1.fy": 718.856, "Camera1.cx": 607.1928, "Camera1.cy": 185.2157,
    "Camera2.fx": 718.856, "Camera2.fy": 718.856, "Camera2.cx": 607.1928, "Camera2.cy": 185.2157,
    "Camera.width": 1241, "Camera.height": 376, "Camera.Base.Line": 386.1448, "Camera.NeedUndistortion": 0, "Camera.fps": 10,
    "Map.ActiveMap.Size": 12,
    "numFeatures.initGood": 100, "numFeatures.trackingGood": 50, "numFeatures.trackingBad": 10,
    "ORBextractor.nInitFeatures": 300, "ORBextractor.nNewFeatures": 100, "ORBextractor.scaleFactor": 1.2, "ORBextractor.nLevels": 8,
    "ORBextractor.iniThFAST": 20, "ORBextractor.minThFAST": 7,
    "Min.Init.Landmark.Num": 200,
    "Viewer.ViewpointY": "1000 # a trailing comment",
    "Backend.Open": 1,
    "Trajectory.Save.Path": '"trajectory.txt"',
}


def write_settings(path, overrides=None):
    """a settings file in the reference's format (flat %YAML:1.0 map) with kitti_00.yaml's values + overrides"""
    cfg = dict(KITTI00_SETTINGS)
    cfg.update(overrides or {})
    with open(path, "w") as f:
        f.write("%YAML:1.0\n# settings of the headless runner\n")
        for k, v in cfg.items():
            f.write(f"{k}: {v}\n")
    return path


def _corridor_frame(args):
    """one stereo pair of make_corridor_sequence (module level: rendered by a process pool when asked for)"""
    k, n_frames, step, seed, h, w, K, baseline, lateral_amp, half_width, cam_height, wall_height, backdrop, ss = args
    rng = np.random.default_rng([9000 + seed, k])                       # per frame: the frames can be rendered in any order
    fx, fy, cx, cy = K
    us = (np.arange(w * ss, dtype=np.float32) + 0.5) / ss - 0.5
    vs = (np.arange(h * ss, dtype=np.float32) + 0.5) / ss - 0.5
    dx = ((us - cx) / fx)[None, :].repeat(h * ss, 0)                   # ray direction (dx, dy, 1)
    dy = ((vs - cy) / fy)[:, None].repeat(w * ss, 1)
    z_end = step * n_frames + backdrop

    def render(c):
        big = np.float32(1e9)
        # positive ray parameters of the four surfaces (inf where the ray does not hit the half-space in front)
        with np.errstate(divide="ignore", invalid="ignore"):
            t_ground = np.where(dy > 1e-6, (cam_height - c[1]) / dy, big)
            t_left = np.where(dx < -1e-6, (-half_width - c[0]) / dx, big)
            t_right = np.where(dx > 1e-6, (half_width - c[0]) / dx, big)
        t_back = np.full_like(dx, np.float32(z_end - c[2]))
        t = np.minimum(np.minimum(t_ground, t_back), np.minimum(t_left, t_right)).astype(np.float32)
        X = c[0] + t * dx; Y = c[1] + t * dy; Z = c[2] + t
        which = np.select([t == t_ground, t == t_left, t == t_right], [0, 1, 2], 3)
        # walls end at wall_height above the ground: above that, the backdrop
        above = ((which == 1) | (which == 2)) & (Y < cam_height - wall_height)
        t2 = np.where(above, t_back, t)
        X = np.where(above, c[0] + t2 * dx, X); Y = np.where(above, c[1] + t2 * dy, Y); Z = np.where(above, c[2] + t2, Z)
        which = np.where(above, 3, which)
        # texture coordinates of the surface hit: ground (X, Z), walls (Y, Z), far plane (X, Y) / 10 (2 m ... 32 m blocks)
        ta = np.select([which == 0, which == 3], [X, X * 0.1], Y)
        tb = np.where(which == 3, Y * 0.1, Z)
        tex = _block_texture(ta, tb, seed * 4 + which)
        img = 30.0 + 200.0 * tex
        img = img.reshape(h, ss, w, ss).mean(axis=(1, 3))
        img += rng.normal(0, 1.0, img.shape)
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)

    c = np.array([lateral_amp * np.sin(0.15 * k), 0.0, step * k], dtype=np.float32)
    left = render(c)
    right = render(c + np.array([baseline, 0, 0], dtype=np.float32))
    return left, right, c.astype(np.float64)


def make_corridor_sequence(n_frames=40, step=0.8, seed=0, h=KITTI_H, w=KITTI_W, K=KITTI_K, baseline=KITTI_BASELINE, lateral_amp=0.3,
                           half_width=6.0, cam_height=1.65, wall_height=6.0, backdrop=120.0, supersample=2, workers=1):
    """A KITTI-00-shaped stereo sequence: the rig drives FORWARD (+z) by `step` metres per frame (0.8 m ~ KITTI at 10 Hz)
    with a slow lateral sway, through a textured corridor (ground plane, two walls, a far backdrop), rendered by ray
    casting each pixel against the planes -- exact perspective, exact stereo geometry, ground-truth poses.  Rotations
    are identity.  The backdrop is a world plane `backdrop` metres beyond the end of the drive.  workers > 1 renders the
    frames in that many processes (0.65 s per pair on one core; same images).
    Returns (frames [(left, right)], T_cw [n,7], centres [n,3])."""
    jobs = [(k, n_frames, step, seed, h, w, tuple(K), baseline, lateral_amp, half_width, cam_height, wall_height, backdrop, supersample)
            for k in range(n_frames)]
    if workers > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(min(workers, n_frames)) as pool:
            out = pool.map(_corridor_frame, jobs, chunksize=1)
    else:
        out = [_corridor_frame(j) for j in jobs]
    frames = [(l, r) for l, r, _ in out]
    centres = [c for _, _, c in out]
    poses = [np.array([0, 0, 0, 1, -c[0], -c[1], -c[2]], dtype=np.float64) for c in centres]     # T_cw, identity rotation
    return frames, np.array(poses), np.array(centres)


def write_kitti_sequence(root, frames, dt=0.1):
    """<root>/{times.txt, image_0/%06d.png, image_1/%06d.png}: the KITTI odometry layout the reference's loader reads
    (include/common/read_kitii_dataset.hpp:16-60)"""
    import os

    from PIL import Image
    for sub in ("image_0", "image_1"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    for i, (L, R) in enumerate(frames):
        Image.fromarray(L).save(os.path.join(root, "image_0", f"{i:06d}.png"))
        Image.fromarray(R).save(os.path.join(root, "image_1", f"{i:06d}.png"))
    with open(os.path.join(root, "times.txt"), "w") as f:               # written last: its presence marks a complete sequence
        for i in range(len(frames)):
            f.write(f"{i * dt:e}\n")
    return root


def make_vocabulary(k=10, L=3, seed=0, stop_fraction=0.02):
    """A synthetic DBoW2-style vocabulary: a full k-ary tree of depth L with random 256-bit node descriptors (children
    are noisy copies of their parent, so descents are meaningful), idf-like leaf weights, a few stopped words (weight 0).
    Nodes in breadth-first order like DBoW2 writes them.  Returns dict(k, L, parent, is_leaf, desc [n,32], weight)."""
    rng = np.random.default_rng(4000 + seed)
    parent, is_leaf, desc, weight = [-1], [0], [np.zeros(32, np.uint8)], [0.0]
    level = [0]
    root_desc = rng.integers(0, 256, 32, dtype=np.uint8)
    node_desc = {0: root_desc}
    for depth in range(1, L + 1):
        nxt = []
        for p in level:
            for _ in range(k):
                flips = rng.random(256) < (0.25 if depth == 1 else 0.12)
                d = np.unpackbits(node_desc[p]) ^ flips.astype(np.uint8)
                d = np.packbits(d)
                nid = len(parent)
                parent.append(p); is_leaf.append(1 if depth == L else 0); desc.append(d); node_desc[nid] = d
                weight.append(0.0 if (depth == L and rng.random() < stop_fraction) else (float(rng.uniform(0.5, 9.0)) if depth == L else 0.0))
                nxt.append(nid)
        level = nxt
    return dict(k=k, L=L, parent=np.array(parent, np.int32), is_leaf=np.array(is_leaf, np.uint8), desc=np.stack(desc).astype(np.uint8),
                weight=np.array(weight, np.float64))


def write_vocabulary_text(path, voc, scoring=0, weighting=0):
    """TemplatedVocabulary::saveToTextFile layout (the ORBvoc.txt format): 'k L scoring weighting', one node per line"""
    with open(path, "w") as f:
        f.write(f"{voc['k']} {voc['L']} {scoring} {weighting}\n")
        for i in range(1, len(voc["parent"])):
            f.write(f"{voc['parent'][i]} {int(voc['is_leaf'][i])} " + " ".join(str(int(b)) for b in voc["desc"][i]) + f" {float(voc['weight'][i])!r}\n")


if __name__ == "__main__":
    # python -m tools.synth corridor <dir> <n_frames> [workers]: render + write a KITTI-layout corridor drive (bench.py's C1 leg
    # starts this in a process of its own, before it touches the GPU, and collects it later)
    import os
    import sys
    if len(sys.argv) >= 4 and sys.argv[1] == "corridor":
        n = int(sys.argv[3])
        nw = int(sys.argv[4]) if len(sys.argv) > 4 else max(1, min(os.cpu_count() or 1, 32))
        step = float(sys.argv[5]) if len(sys.argv) > 5 else 0.8
        amp = float(sys.argv[6]) if len(sys.argv) > 6 else 0.3
        fr, _, cen = make_corridor_sequence(n_frames=n, workers=nw, step=step, lateral_amp=amp)
        os.makedirs(sys.argv[2], exist_ok=True)
        np.save(os.path.join(sys.argv[2], "centres.npy"), cen)
        write_kitti_sequence(sys.argv[2], fr)
        print(f"wrote {n} stereo pairs to {sys.argv[2]}")
    else:
        print("usage: python -m tools.synth corridor <dir> <n_frames> [workers [step_m [lateral_amp_m]]]", file=sys.stderr)
        sys.exit(2)

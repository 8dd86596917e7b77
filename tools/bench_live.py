"""bench.py's headline leg: B resident sliding local-BA windows driven the way a live backend drives them.

What Backend::InsertKeyFrame -> Map::InsertKeyFrame -> RemoveOldActiveKeyframe / RemoveOldActiveMapPoints ->
Backend::OptimizeActiveMap do per keyframe (/root/reference/src/ssvio/backend.cpp:57-76, 88-169, map.cpp:27-56, 89-160):
the window the backend optimised last time loses its oldest keyframe, gains the new one (pose, the landmarks it
introduces, its observations), is optimised where it lies, and poses + landmarks go back to the map.  Here: B window
objects (ssx_ba_window) in G groups, one host thread + library context per group; per step and group ONE
ssx_ba_window_update_batch call (pop + push for each of its windows) and ONE ssx_ba_window_solve_batch call (poses and
landmarks of every window downloaded).  Nothing but the new keyframe's ~75 KB crosses PCIe on the way in.

The groups are NOT kept in lock step: the caller releases a step for all groups (release), each group works through the
released steps at its own pace, and the caller waits for a step to be finished one step late (wait_done) -- so that one
group's host phase (edits, counting tables) runs beside the other groups' kernels.

The drive is synthetic (make_drive): a straight forward drive, 0.8 m per keyframe, every keyframe introduces `lm_per_kf`
landmarks that are observed by `obs_per_lm` consecutive keyframes -- 10 keyframes x 2000 observations = BASELINE
configs[2]'s 20 000 edges on a window that moves (the landmark count of a moving window is higher than C3's 4000: ~5200,
the partially observed ones at both ends included).  Test / bench infrastructure, not part of the product."""
from __future__ import annotations

import ctypes as C
import os
import threading
import time

import numpy as np

KITTI_K = (718.856, 718.856, 607.1928, 185.2157)


def _quat_rot(q, p):
    """rotate points p [n,3] by unit quaternions q [n,4] (x y z w)"""
    u = np.cross(q[:, :3], p)
    u += u
    return p + q[:, 3:4] * u + np.cross(q[:, :3], u)


def make_drive(n_kf, lm_per_kf=400, obs_per_lm=5, seed=0, K=KITTI_K, frac_fixed=0.15, frac_gross=0.03, pix_sigma=0.5, gross_sigma=30.0,
               pose_t_noise=0.02, pose_r_noise=0.002):
    """-> list of n_kf keyframe feeds: dict(pose [7], new_ids int64, new_xyz [n,3], new_fixed u8, obs_lm int64 ids, obs_uv [m,2]);
    numpy-vectorised (the graph generator of the tests, synth.make_ba_problem, is a Python loop over observations); same noise
    model: 0.5 px measurement noise, 3 % gross outliers of 30 px, measurements that are float values (cv::KeyPoint::pt)."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = K
    k = obs_per_lm
    n_lm = lm_per_kf * (n_kf + k - 1)
    first = np.arange(n_lm) // lm_per_kf - (k - 1)                       # may be negative: partially observed at the start
    seen0 = np.maximum(first, 0)
    centers = np.zeros((n_kf + k, 3)); centers[:, 2] = 0.8 * np.arange(n_kf + k)
    # a landmark lies inside the 1241 x 376 image of the LAST keyframe that observes it (the closest view), hence of all of them
    z = rng.uniform(4, 40, n_lm)
    local = np.stack([z * rng.uniform(-0.8, 0.8, n_lm), z * rng.uniform(-0.24, 0.24, n_lm), z], 1)
    pts = local + centers[first + k - 1]                                  # camera axes = world axes (straight drive)
    fixed = (rng.random(n_lm) < frac_fixed).astype(np.uint8)
    # noisy initial poses T_cw = (dq, rot(dq, -c) + noise)
    r = rng.uniform(-pose_r_noise, pose_r_noise, (n_kf, 3))
    dq = np.concatenate([0.5 * r, np.ones((n_kf, 1))], 1)
    dq /= np.linalg.norm(dq, axis=1, keepdims=True)
    t = _quat_rot(dq, -centers[:n_kf]) + rng.uniform(-pose_t_noise, pose_t_noise, (n_kf, 3))
    poses = np.ascontiguousarray(np.concatenate([dq, t], 1))
    feeds = []
    lm_all = np.arange(n_lm)
    for i in range(n_kf):
        vis = lm_all[(first <= i) & (i < first + k)]                     # ascending ids
        pc = pts[vis] - centers[i]
        uv = np.stack([fx * pc[:, 0] / pc[:, 2] + cx, fy * pc[:, 1] / pc[:, 2] + cy], 1)
        uv += rng.normal(0, pix_sigma, uv.shape)
        gross = rng.random(len(vis)) < frac_gross
        uv[gross] += rng.normal(0, gross_sigma, (int(gross.sum()), 2))
        uv = uv.astype(np.float32).astype(np.float64)
        new = vis[seen0[vis] == i]
        feeds.append(dict(pose=np.ascontiguousarray(poses[i]), new_ids=np.ascontiguousarray(new.astype(np.int64)),
                          new_xyz=np.ascontiguousarray(pts[new]), new_fixed=np.ascontiguousarray(fixed[new]),
                          obs_lm=np.ascontiguousarray(vis.astype(np.int64)), obs_uv=np.ascontiguousarray(uv)))
    return feeds


class LiveBackend:
    """B resident sliding windows of `win_kf` keyframes in `threads` groups; see the module docstring."""

    def __init__(self, ssvio_amd, dev_index, B, n_steps, threads=2, win_kf=10, n_traj=4, seed=900, cam_ext=None, preroll=12):
        from ssvio_amd import ba
        from ssvio_amd._lib import BaResult, BaWindowUpdate, dbl_p, ptr, u8_p
        self.B, self.win_kf, self.n_traj = B, win_kf, n_traj
        i64_p, i32_p = C.POINTER(C.c_int64), C.POINTER(C.c_int32)
        self.G = G = max(1, min(int(threads), B))
        n_kf = win_kf + preroll + n_steps + 1
        self.feeds = [make_drive(n_kf, seed=seed + q) for q in range(n_traj)]
        K = np.array(KITTI_K, dtype=np.float64)
        from tools.synth import stereo_cam_ext
        ext = stereo_cam_ext() if cam_ext is None else np.asarray(cam_ext, dtype=np.float64)
        # SSX_BENCH_GROUP_PRIO="-1,0,1": the groups' streams at different HIP priorities (-1 high .. 1 low; group g takes entry g mod len).
        # Equal priorities share the chip evenly, so the groups finish -- and enter their host phases -- together (a convoy: the 9 %
        # of idle span in profiles/r06/live_idle_gaps.txt); unequal ones finish one after the other.
        prio = [int(x) for x in os.environ.get("SSX_BENCH_GROUP_PRIO", "").split(",") if x.strip()]
        self._prio_streams = []
        # SSX_BENCH_PAIR_STREAMS=k: k consecutive groups share ONE stream (their contexts are created on it): a group's kernels then queue
        # behind its siblings' instead of beside them, and its host phases lie beside the siblings' kernels -- a pipeline of half-groups
        # without more streams than today (profiles/r06/live_pair_streams.txt)
        pair = int(os.environ.get("SSX_BENCH_PAIR_STREAMS", "0") or 0)
        if pair > 1 and not prio:
            hip_path = next((ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln), "libamdhip64.so")
            hip = C.CDLL(hip_path)
            hip.hipSetDevice(C.c_int(dev_index))
            for q in range((G + pair - 1) // pair):
                st = C.c_void_p()
                rc = hip.hipStreamCreateWithFlags(C.byref(st), C.c_uint(1))                                   # hipStreamNonBlocking
                if rc != 0 or not st.value:
                    raise RuntimeError(f"hipStreamCreateWithFlags -> {rc}")
                self._prio_streams.append(st.value)
            self.ctx = [ssvio_amd.Context(dev_index, stream=self._prio_streams[g // pair]) for g in range(G)]
        elif prio:
            # the HIP runtime this process already holds (torch's bundled copy): a second copy would not know the library's device
            hip_path = next((ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln), "libamdhip64.so")
            hip = C.CDLL(hip_path)
            hip.hipSetDevice(C.c_int(dev_index))
            for g in range(G):
                st = C.c_void_p()
                rc = hip.hipStreamCreateWithPriority(C.byref(st), C.c_uint(1), C.c_int(prio[g % len(prio)]))    # 1 = hipStreamNonBlocking
                if rc != 0 or not st.value:
                    raise RuntimeError(f"hipStreamCreateWithPriority -> {rc}")
                self._prio_streams.append(st.value)
            self.ctx = [ssvio_amd.Context(dev_index, stream=self._prio_streams[g]) for g in range(G)]
        else:
            self.ctx = [ssvio_amd.Context(dev_index) for _ in range(G)]
        lib = self.lib = self.ctx[0].lib
        lib.ssx_ba_device_turns.restype = None
        lib.ssx_ba_device_turns.argtypes = [C.c_int32]
        self.turns = G > 1 and os.environ.get("SSX_BENCH_TURNS") is not None   # (measured: no gain, profiles/r05/live_backend_orchestration.md)
        lib.ssx_ba_device_turns(1 if self.turns else 0)
        lib.ssx_ba_set_batch_groups.restype = C.c_int32
        lib.ssx_ba_set_batch_groups.argtypes = [C.c_void_p, C.c_int32]
        self.batch_groups = int(os.environ.get("SSX_BENCH_BATCH_GROUPS", "1" if G > 1 else "0"))
        for c in self.ctx:
            c.check(lib.ssx_ba_set_batch_groups(c.handle, self.batch_groups))
        lib.ssx_ba_window_update_batch.restype = C.c_int32
        lib.ssx_ba_window_update_batch.argtypes = [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(BaWindowUpdate), C.POINTER(C.c_int32)]
        # A backend keeps, with every map point, the slot the window gave it (ssx_ba_window_push_keyframe_slots).  The slots of a
        # push / pop sequence are deterministic: one untimed pass over a scratch window per trajectory records them, and the steps
        # pass slot arrays instead of ids (what a C++ caller reads out of its MapPoint objects).
        for q, fd in enumerate(self.feeds):
            scratch = ba.BaWindow(self.ctx[0], K, ext)
            n_lm = int(max(a["obs_lm"].max() for a in fd)) + 1
            slot_of = np.full(n_lm, -10 ** 9, dtype=np.int64)
            is_new = np.zeros(n_lm, dtype=bool); rank_new = np.zeros(n_lm, dtype=np.int64)
            for k, a in enumerate(fd):
                if k >= win_kf:
                    scratch.pop(k - win_kf)
                is_new[:] = False; is_new[a["new_ids"]] = True
                rank_new[a["new_ids"]] = np.arange(len(a["new_ids"]))
                lm = a["obs_lm"]
                a["obs_slot"] = np.ascontiguousarray(np.where(is_new[lm], -1 - rank_new[lm], slot_of[lm]).astype(np.int32))
                slot_of[a["new_ids"]] = scratch.push_slots(k, a["pose"], new_ids=a["new_ids"], new_xyz=a["new_xyz"], new_fixed=a["new_fixed"],
                                                           obs_slot=a["obs_slot"], obs_uv=a["obs_uv"])
            scratch.close()
        self.grp_of = [i * G // B for i in range(B)]
        self.wins = [ba.BaWindow(self.ctx[self.grp_of[i]], K, ext) for i in range(B)]
        for i, w in enumerate(self.wins):
            for k in range(win_kf):
                a = self.feeds[i % n_traj][k]
                w.push_slots(k, a["pose"], new_ids=a["new_ids"], new_xyz=a["new_xyz"], new_fixed=a["new_fixed"], obs_slot=a["obs_slot"], obs_uv=a["obs_uv"])
        self._keep = []
        self.groups = []
        max_lm = 0
        for fd in self.feeds:
            for k in range(len(fd) - win_kf):
                max_lm = max(max_lm, len(np.unique(np.concatenate([a["obs_lm"] for a in fd[k:k + win_kf]]))))
        for g in range(G):
            idx = [i for i in range(B) if self.grp_of[i] == g]
            hs = (C.c_void_p * len(idx))(*[self.wins[i].handle for i in idx])
            res = (BaResult * len(idx))()
            for j in range(len(idx)):                                   # result buffers: poses + landmarks of the window, every step
                po = np.zeros((win_kf + 2, 7)); pt = np.zeros((max_lm + 64, 3))
                self._keep.append((po, pt))
                res[j].poses_out = ptr(po, dbl_p); res[j].points_out = ptr(pt, dbl_p)
            upd = {}
            for k in range(win_kf, n_kf):                               # the edits of every step, built once (a C caller fills them from its map)
                arr = (BaWindowUpdate * len(idx))()
                for j, i in enumerate(idx):
                    a = self.feeds[i % n_traj][k]
                    so = np.zeros(len(a["new_ids"]), dtype=np.int32)
                    self._keep.append(so)
                    u = arr[j]
                    u.pop = 1; u.pop_kf_id = k - win_kf; u.push = 1; u.kf_id = k
                    u.pose7 = ptr(a["pose"], dbl_p); u.pose_fixed = 0; u.n_new = len(a["new_ids"]); u.new_ids = ptr(a["new_ids"], i64_p)
                    u.new_xyz = ptr(a["new_xyz"], dbl_p); u.new_fixed = ptr(a["new_fixed"], u8_p); u.new_slots_out = ptr(so, i32_p)
                    u.n_obs = len(a["obs_slot"]); u.obs_slot = ptr(a["obs_slot"], i32_p); u.obs_uv = ptr(a["obs_uv"], dbl_p)
                upd[k] = arr
            self.groups.append((idx, hs, res, upd))
        self.next_kf = win_kf                                           # the keyframe the next step pushes
        self.last_kf = n_kf
        self.go = [threading.Semaphore(0) for _ in range(G)]
        self.done = [threading.Semaphore(0) for _ in range(G)]
        self.todo = [0] * G
        self.quit = False
        self.err = []
        self.t_solve = [0.0] * G; self.t_edit = [0.0] * G; self.iters = [0] * G; self.steps_done = [0] * G
        # (the group threads live as long as the object: a new thread's first HIP call costs ~10 ms of runtime set-up)
        self.threads = [threading.Thread(target=self._loop, args=(g,), daemon=True) for g in range(G)]
        for t in self.threads:
            t.start()
        for g in range(G):                                              # first optimisation of the initial windows
            self.ctx[g].check(lib.ssx_ba_window_solve_batch(len(self.groups[g][0]), self.groups[g][1], self.groups[g][2]))
        # one full turnover of every window: their storage has been rewritten once, every buffer has its final size (a grown arena
        # is a hipFree, i.e. a device-wide synchronisation: set-up, like the marshalling of a resident batch)
        self.run(preroll)

    # ---- group thread: works through the released steps --------------------------------------------------------------------------
    def _loop(self, g):
        idx, hs, res, upd = self.groups[g]
        k = self.win_kf
        while True:
            self.go[g].acquire()
            if self.quit:
                return
            try:
                if not self.err:
                    t0 = time.perf_counter()
                    self.ctx[g].check(self.lib.ssx_ba_window_update_batch(len(idx), hs, upd[k], None))
                    t1 = time.perf_counter()
                    self.ctx[g].check(self.lib.ssx_ba_window_solve_batch(len(idx), hs, res))
                    t2 = time.perf_counter()
                    self.t_edit[g] += t1 - t0; self.t_solve[g] += t2 - t1
                    self.iters[g] += sum(res[j].n_iters for j in range(len(idx)))
                    self.steps_done[g] += 1
                    k += 1
            except Exception as exc:                                    # noqa: BLE001 -- reported by the caller's thread
                self.err.append(exc)
            self.done[g].release()

    def release(self):
        """one more step (pop + push + optimise for every window) may be worked on by every group"""
        if self.next_kf >= self.last_kf:
            raise RuntimeError("LiveBackend: the drive is used up")
        self.next_kf += 1
        for s in self.go:
            s.release()

    def wait_done(self):
        """wait until every group has finished one more of the released steps"""
        for s in self.done:
            s.acquire()
        if self.err:
            raise self.err[0]

    def run(self, n, lag=1, each_step=None):
        """n steps; each_step() (the front-end's part of the step) runs on the caller's thread right before a step is released"""
        for i in range(n):
            if each_step:
                each_step()
            self.release()
            if i >= lag:
                self.wait_done()
        for _ in range(min(lag, n)):
            self.wait_done()

    def reset_counters(self):
        self.t_solve = [0.0] * self.G; self.t_edit = [0.0] * self.G; self.iters = [0] * self.G; self.steps_done = [0] * self.G

    def synchronize(self):
        for c in self.ctx:
            c.synchronize()

    def window_size(self):
        return self.wins[0].size()

    def close(self):
        self.lib.ssx_ba_device_turns(0)
        self.quit = True
        for s in self.go:
            s.release()
        for t in self.threads:
            t.join(timeout=5)
        for w in self.wins:
            w.close()
        for c in self.ctx:
            c.close()

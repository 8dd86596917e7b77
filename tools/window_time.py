"""Resident sliding windows (ssx_ba_window): where a step's time goes.  python tools/window_time.py [windows] [steps]"""
import os, sys, time, ctypes as C, concurrent.futures as cf
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ssvio_amd
from ssvio_amd import ba
from ssvio_amd._lib import BaResult, dbl_p, u8_p, ptr
from tools.synth import make_ba_problem
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ctx = ssvio_amd.Context(0)
n_kf_total = 10 + STEPS + 3
traj = [make_ba_problem(P=n_kf_total, L=400 * (n_kf_total - 4), seed=900 + k) for k in range(4)]
i64_p = C.POINTER(C.c_int64)
def feed_of(pr):
    first = np.full(pr["L"], 10 ** 9, dtype=np.int64); np.minimum.at(first, pr["edge_point"], pr["edge_pose"]); out = []
    for k in range(pr["P"]):
        new = np.nonzero(first == k)[0]; e = np.nonzero(pr["edge_pose"] == k)[0]
        a = dict(pose=np.ascontiguousarray(pr["poses"][k]), new_ids=np.ascontiguousarray(new.astype(np.int64)), new_xyz=np.ascontiguousarray(pr["points"][new]),
                 new_fixed=np.ascontiguousarray(pr["point_fixed"][new]), obs_lm=np.ascontiguousarray(pr["edge_point"][e].astype(np.int64)), obs_uv=np.ascontiguousarray(pr["edge_uv"][e]))
        a["args"] = (ptr(a["pose"], dbl_p), 0, len(new), ptr(a["new_ids"], i64_p), ptr(a["new_xyz"], dbl_p), ptr(a["new_fixed"], u8_p), len(e), ptr(a["obs_lm"], i64_p), ptr(a["obs_uv"], dbl_p), None)
        out.append(a)
    return out
feeds = [feed_of(t) for t in traj]
i32p = C.POINTER(C.c_int32)
for q, fd in enumerate(feeds):                      # record the landmark slots of the push / pop sequence (see bench.py)
    scratch = ba.BaWindow(ctx, traj[q]["K"], traj[q]["cam_ext"])
    slot_of = np.full(traj[q]["L"], -10 ** 9, dtype=np.int64)
    for k, a in enumerate(fd):
        if k >= 10: scratch.pop(k - 10)
        is_new = np.zeros(traj[q]["L"], dtype=bool); is_new[a["new_ids"]] = True
        rank_new = np.zeros(traj[q]["L"], dtype=np.int64); rank_new[a["new_ids"]] = np.arange(len(a["new_ids"]))
        lm = a["obs_lm"]
        a["obs_slot"] = np.ascontiguousarray(np.where(is_new[lm], -1 - rank_new[lm], slot_of[lm]).astype(np.int32))
        a["slots_out"] = np.zeros(len(a["new_ids"]), dtype=np.int32)
        a["args_slots"] = (ptr(a["pose"], dbl_p), 0, len(a["new_ids"]), ptr(a["new_ids"], i64_p), ptr(a["new_xyz"], dbl_p), ptr(a["new_fixed"], u8_p),
                           ptr(a["slots_out"], i32p), len(lm), ptr(a["obs_slot"], i32p), ptr(a["obs_uv"], dbl_p), None)
        ctx.check(lib_ := ctx.lib.ssx_ba_window_push_keyframe_slots(scratch.handle, k, *a["args_slots"]))
        slot_of[a["new_ids"]] = a["slots_out"]
    scratch.close()
USE_SLOTS = not os.environ.get("IDS")
wins = [ba.BaWindow(ctx, traj[i % 4]["K"], traj[i % 4]["cam_ext"]) for i in range(B)]
lib = ctx.lib
for i, w in enumerate(wins):
    for k in range(10): ctx.check(lib.ssx_ba_window_push_keyframe(w.handle, k, *feeds[i % 4][k]["args"]))
hs = (C.c_void_p * B)(*[w.handle for w in wins]); res = (BaResult * B)(); keep = []
for i in range(B):
    po = np.zeros((16, 7)); pt = np.zeros((6400, 3)); keep.append((po, pt)); res[i].poses_out = ptr(po, dbl_p); res[i].points_out = ptr(pt, dbl_p)
pool = cf.ThreadPoolExecutor(max_workers=16)
tpop = tpush = 0.0
def churn_one(i, k):
    global tpop, tpush
    t0 = time.perf_counter()
    lib.ssx_ba_window_pop_keyframe(wins[i].handle, k - 10)
    t1 = time.perf_counter()
    if USE_SLOTS: r = lib.ssx_ba_window_push_keyframe_slots(wins[i].handle, k, *feeds[i % 4][k]["args_slots"])
    else: r = lib.ssx_ba_window_push_keyframe(wins[i].handle, k, *feeds[i % 4][k]["args"])
    t2 = time.perf_counter()
    tpop += t1 - t0; tpush += t2 - t1
    return r
ctx.check(lib.ssx_ba_window_solve_batch(B, hs, res))
tc = ts = 0.0
for k in range(10, 10 + STEPS):
    t0 = time.perf_counter()
    if os.environ.get("SERIAL"):
        for i in range(B): churn_one(i, k)
    else:
        list(pool.map(lambda i: churn_one(i, k), range(B)))
    t1 = time.perf_counter()
    ctx.check(lib.ssx_ba_window_solve_batch(B, hs, res))
    t2 = time.perf_counter()
    if k >= 12: tc += t1 - t0; ts += t2 - t1
n = STEPS - 2
print(f"pop {tpop / STEPS / B * 1e6:.1f} us, push {tpush / STEPS / B * 1e6:.1f} us per window ({'slots' if USE_SLOTS else 'ids'})")
print(f"GPU clock of the last solve_batch (kernels + download, after the uploads): {res[0].ms_total:.3f} ms")
print(f"B={B}: churn (pop + push, {'serial' if os.environ.get('SERIAL') else '16 threads'}) {tc / n * 1e3:.3f} ms, solve_batch {ts / n * 1e3:.3f} ms per step; window {wins[0].size()}")
os.environ["SSX_WIN_TIMING"] = "1"

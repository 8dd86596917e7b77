"""GPU parity of the pyramidal LK tracker (N1) against the CPU oracle: BIT-EXACT pyramids, Scharr images, tracked
positions (f32 bit patterns), status flags and errors -- the normal matrix / mismatch sums are exact integers on
both sides, every float operation is the same IEEE operation in the same order.  (The oracle itself is "parity
unpinned vs OpenCV": tests/test_oracle_lk.py.)"""
import numpy as np
import pytest

from ssvio_amd import lk
from tools.synth import make_stereo_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kitti():
    return make_stereo_pair(seed=0)


def _points(po, img, n=400):
    k, _ = po.orb_extract(img, prm=po.orb_params(nfeatures=n, nlevels=1))
    return np.stack([k["x"], k["y"]], 1).astype(np.float32)


def _same(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()


def test_pyramid_and_scharr_bit_exact(ctx, po, kitti):
    L, R, _ = kitti
    pts = _points(po, L, 50)
    lk.calcOpticalFlowPyrLK(ctx, L, R, pts, pts)
    prev, nxt = L, R
    for level in range(4):
        if level > 0:
            prev, nxt = po.lk_pyr_down(prev), po.lk_pyr_down(nxt)
        assert np.array_equal(lk.stage_level(ctx, 0, level), prev), f"prev pyramid level {level}"
        assert np.array_equal(lk.stage_level(ctx, 1, level), nxt), f"next pyramid level {level}"
        assert np.array_equal(lk.stage_deriv(ctx, level), po.lk_scharr(prev)), f"Scharr level {level}"


def test_stereo_tracking_with_initial_flow_bit_exact(ctx, po, kitti):
    """FindFeaturesInRight (frontend.cpp:374-384): left -> right, initial guess = the left position"""
    L, R, disp = kitti
    pts = _points(po, L, 1500)
    g = lk.calcOpticalFlowPyrLK(ctx, L, R, pts, pts)
    o = po.lk_track(L, R, pts, pts)
    assert g[3] == o[3] == 3
    assert _same(g[1], o[1]) and _same(g[0], o[0]) and _same(g[2], o[2])
    ok = g[1] > 0
    d = (pts[:, 0] - g[0][:, 0])[ok]
    true = disp[np.clip(np.round(pts[ok, 1]).astype(int), 0, 375), np.clip(np.round(pts[ok, 0]).astype(int), 0, 1240)]
    assert ok.mean() > 0.9 and np.median(np.abs(d - true)) < 0.1          # sanity against the synthetic ground truth


def test_temporal_tracking_bit_exact(ctx, po, kitti):
    """TrackLastFrame (frontend.cpp:156-166): last left -> current left; here a shifted + noisy copy, with and
    without initial guesses, other windows and iteration limits"""
    L = kitti[0]
    rng = np.random.default_rng(1)
    cur = np.roll(np.roll(L, 2, axis=0), -6, axis=1)
    cur = np.clip(cur.astype(np.int16) + rng.integers(-3, 4, cur.shape), 0, 255).astype(np.uint8)
    pts = _points(po, L, 800)
    guess = pts + np.array([-5.0, 1.5], np.float32)
    for kw, init in ((dict(), guess), (dict(), None), (dict(winSize=7, maxLevel=2), guess), (dict(winSize=15, maxLevel=1, maxCount=5), None),
                     (dict(maxLevel=0, epsilon=0.1), guess)):
        g = lk.calcOpticalFlowPyrLK(ctx, L, cur, pts, init, **kw)
        prm = po.lk_params(win=kw.get("winSize", 11), max_level=kw.get("maxLevel", 3), max_iters=kw.get("maxCount", 30),
                           eps=kw.get("epsilon", 0.01), use_initial_flow=int(init is not None))
        o = po.lk_track(L, cur, pts, init, prm=prm)
        assert g[3] == o[3]
        assert _same(g[1], o[1]) and _same(g[0], o[0]) and _same(g[2], o[2]), kw
    g = lk.calcOpticalFlowPyrLK(ctx, L, cur, pts, guess)
    ok = g[1] > 0
    assert ok.mean() > 0.9 and np.abs(np.median((g[0] - pts)[ok], 0) - np.array([-6.0, 2.0])).max() < 0.05


def test_edge_cases_bit_exact(ctx, po):
    """points outside / at the border / on flat areas, a pyramid cut short, no points, bad arguments"""
    L, R, _ = make_stereo_pair(seed=5, h=120, w=160, n_blobs=120)
    pts = np.array([[-40.0, 30.0], [30.0, 400.0], [500.0, 30.0], [0.0, 0.0], [159.0, 119.0], [5.2, 3.7], [155.5, 60.0],
                    [80.0, 60.0], [-5.5, -5.5], [164.9, 124.9]], np.float32)
    for a, b in ((L, R), (L, L), (np.full_like(L, 90), np.full_like(L, 90))):
        g = lk.calcOpticalFlowPyrLK(ctx, a, b, pts, pts)
        o = po.lk_track(a, b, pts, pts)
        assert g[3] == o[3] and _same(g[1], o[1]) and _same(g[0], o[0]) and _same(g[2], o[2])
    tiny = make_stereo_pair(seed=6, h=40, w=60, n_blobs=30)[0]
    g = lk.calcOpticalFlowPyrLK(ctx, tiny, tiny, np.array([[30.0, 20.0]], np.float32))
    o = po.lk_track(tiny, tiny, np.array([[30.0, 20.0]], np.float32), prm=po.lk_params(use_initial_flow=0))
    assert g[3] == o[3] == 1 and _same(g[0], o[0]) and _same(g[1], o[1])
    g = lk.calcOpticalFlowPyrLK(ctx, L, R, np.zeros((0, 2), np.float32))
    assert g[0].shape == (0, 2) and g[1].shape == (0,)
    with pytest.raises(Exception):
        lk.calcOpticalFlowPyrLK(ctx, L, R, pts, pts, winSize=10)          # even window
    with pytest.raises(ValueError):
        lk.calcOpticalFlowPyrLK(ctx, L, R[:50], pts)


def test_chained_tracking_reuses_the_previous_pyramid(ctx, po):
    """ssx_lk_track_next: frame t -> t+1 with the pyramid of frame t kept from the previous call; same bits as the
    two-image call and as the oracle, over a 4-frame sequence; refused without a matching previous call"""
    from ssvio_amd import Context
    from ssvio_amd._lib import SsxError
    from tools.synth import make_lateral_sequence
    frames = [f[0] for f in make_lateral_sequence(n_frames=4, seed=2)[0]]
    pts = _points(po, frames[0], 600)
    fresh = Context(0)
    with pytest.raises(SsxError):
        lk.calcOpticalFlowPyrLK(fresh, None, frames[1], pts)              # nothing to chain to
    g = lk.calcOpticalFlowPyrLK(fresh, frames[0], frames[1], pts)
    o = po.lk_track(frames[0], frames[1], pts, prm=po.lk_params(use_initial_flow=0))
    assert _same(g[0], o[0]) and _same(g[1], o[1])
    for t in (1, 2):
        keep = g[1] > 0
        pts = g[0][keep]
        g = lk.calcOpticalFlowPyrLK(fresh, None, frames[t + 1], pts)
        o = po.lk_track(frames[t], frames[t + 1], pts, prm=po.lk_params(use_initial_flow=0))
        assert _same(g[0], o[0]) and _same(g[1], o[1]) and _same(g[2], o[2]), t
        assert np.array_equal(lk.stage_level(fresh, 0, 2), po.lk_pyr_down(po.lk_pyr_down(frames[t])))
        assert (g[1] > 0).mean() > 0.8
    with pytest.raises(SsxError):
        lk.calcOpticalFlowPyrLK(fresh, None, frames[3], pts, winSize=7)   # other window: the kept pyramid does not fit
    with pytest.raises(SsxError):
        lk.calcOpticalFlowPyrLK(fresh, None, frames[3][:100], pts)        # other size
    fresh.close()


def test_track_batch_equals_single_calls(ctx, po):
    """ssx_lk_track_batch -- one frame of each of n streams in one call, every stream on its own slot of the context -- returns,
    per job, the bits of ssx_lk_track / ssx_lk_track_next on a context of the job's own: fresh and chained jobs mixed in one
    call, different point counts (one job without points), three chained steps, the jobs listed in another order than their slots."""
    import ssvio_amd
    S, steps = 6, 3
    rng = np.random.default_rng(7)
    seqs = []
    for s_ in range(S):
        base = make_stereo_pair(seed=20 + s_, h=240, w=400, n_blobs=500)[0]
        seqs.append([np.ascontiguousarray(np.roll(np.roll(base, k * (1 + s_ % 3), axis=1), k * (s_ % 2), axis=0)) for k in range(steps + 1)])
    pts = [_points(po, seqs[s_][0], 40 + 37 * s_) if s_ != 4 else np.zeros((0, 2), np.float32) for s_ in range(S)]
    # reference: every stream alone on a context of its own
    ref = []
    for s_ in range(S):
        c = ssvio_amd.Context(0)
        p = pts[s_]
        out = []
        for k in range(steps):
            prev = seqs[s_][k] if (k == 0 or (s_ == 2 and k == 2)) else None          # stream 2 restarts its chain at step 2
            g = lk.calcOpticalFlowPyrLK(c, prev, seqs[s_][k + 1], p, p)
            out.append(g)
            p = g[0]
        ref.append(out)
        c.close()
    p_cur = [p.copy() for p in pts]
    order = [3, 0, 5, 1, 4, 2]
    for k in range(steps):
        jobs = [dict(slot=s_, prev=(seqs[s_][k] if (k == 0 or (s_ == 2 and k == 2)) else None), next=seqs[s_][k + 1], prev_pts=p_cur[s_], next_pts=p_cur[s_])
                for s_ in order]
        got = lk.track_batch(ctx, jobs)
        for s_, g in zip(order, got):
            r = ref[s_][k]
            assert _same(g[0], r[0]) and _same(g[1], r[1]) and _same(g[2], r[2]), (k, s_)
            p_cur[s_] = g[0]
    import pytest as _pt
    with _pt.raises(ssvio_amd.SsxError):                     # a slot twice in one call
        lk.track_batch(ctx, [dict(slot=1, prev=seqs[0][0], next=seqs[0][1], prev_pts=pts[0]), dict(slot=1, prev=seqs[1][0], next=seqs[1][1], prev_pts=pts[1])])
    with _pt.raises(ssvio_amd.SsxError):                     # a chained job on a slot that holds nothing
        lk.track_batch(ctx, [dict(slot=77, prev=None, next=seqs[0][1], prev_pts=pts[0])])


@pytest.mark.parametrize("h,w,win,max_level", [(120, 160, 11, 3), (121, 163, 11, 3), (97, 250, 7, 3), (113, 113, 15, 2), (200, 333, 5, 3), (64, 90, 11, 3),
                                               (111, 112, 11, 1), (130, 97, 13, 3), (376, 1241, 11, 2)])
def test_pyramids_borders_and_chains_over_sizes_and_windows(ctx, po, h, w, win, max_level):
    """Odd and even sizes at every level, windows 5 .. 15 (border 6 .. 16), 1 - 3 pyramid levels above the image, small top levels: the
    pyramid / derivative images of both frames level by level, points on a grid that reaches into the image corners (their windows
    read the reflected border of every level), and a chained third frame (the derivative images of a chained job's previous image were
    written one call earlier) -- all bit for bit the oracle's.  Covers k_lk_pyramid (one launch per image) and, where a level is too
    small for it, the per-level kernels."""
    from ssvio_amd import Context
    rng = np.random.default_rng(h * 1000 + w)
    L, R, _ = make_stereo_pair(seed=40 + h % 7, h=h, w=w, n_blobs=max(60, h * w // 150))
    T = np.clip(np.roll(L, (1, 2), (0, 1)).astype(np.int16) + rng.integers(-2, 3, L.shape), 0, 255).astype(np.uint8)
    gx, gy = np.meshgrid(np.linspace(0.5, w - 1.5, 12), np.linspace(0.5, h - 1.5, 9))
    pts = np.stack([gx.ravel(), gy.ravel()], 1).astype(np.float32)
    pts = np.concatenate([pts, np.array([[0, 0], [w - 1, h - 1], [w - 1, 0], [0, h - 1], [-3.5, 4.0], [w + 2.0, h / 2]], np.float32)])
    c = Context(0)
    prm = po.lk_params(win=win, max_level=max_level)
    g = lk.calcOpticalFlowPyrLK(c, L, R, pts, pts, winSize=win, maxLevel=max_level)
    o = po.lk_track(L, R, pts, pts, prm=prm)
    assert g[3] == o[3]
    assert _same(g[1], o[1]) and _same(g[0], o[0]) and _same(g[2], o[2])
    prev, nxt = L, R
    for level in range(g[3] + 1):
        if level > 0:
            prev, nxt = po.lk_pyr_down(prev), po.lk_pyr_down(nxt)
        assert np.array_equal(lk.stage_level(c, 0, level), prev) and np.array_equal(lk.stage_level(c, 1, level), nxt), level
        assert np.array_equal(lk.stage_deriv(c, level), po.lk_scharr(prev)), level
    # chained: R is now the previous image
    g2 = lk.calcOpticalFlowPyrLK(c, None, T, pts, pts, winSize=win, maxLevel=max_level)
    o2 = po.lk_track(R, T, pts, pts, prm=prm)
    assert _same(g2[1], o2[1]) and _same(g2[0], o2[0]) and _same(g2[2], o2[2])
    prev = R
    for level in range(g2[3] + 1):
        if level > 0:
            prev = po.lk_pyr_down(prev)
        assert np.array_equal(lk.stage_deriv(c, level), po.lk_scharr(prev)), ("chained", level)
    c.close()


def test_wide_and_narrow_calls_interleave(ctx, po):
    """Calls of up to 16 jobs build their pyramids and derivative images with k_lk_pyramid (one launch per image, the derivative
    images of the new frame kept for the chained job that follows); wider calls use the per-level kernels and keep none.  A slot that
    goes wide -> narrow -> narrow -> wide gets the oracle's bits at every step (the narrow call after a wide one computes the missing
    derivative images itself), and so do the slots that only see the wide calls."""
    from ssvio_amd import Context
    c = Context(0)
    S, few = 20, 4
    base = [make_stereo_pair(seed=60 + s_, h=120, w=168, n_blobs=200)[0] for s_ in range(S)]
    frames = [[np.ascontiguousarray(np.roll(base[s_], (k * (s_ % 2), k * (1 + s_ % 3)), (0, 1))) for k in range(5)] for s_ in range(S)]
    gx, gy = np.meshgrid(np.linspace(6, 160, 8), np.linspace(6, 112, 6))
    pts = np.stack([gx.ravel(), gy.ravel()], 1).astype(np.float32)
    cur = [0] * S                                                       # the frame each slot's kept pyramid belongs to

    def call(slots, fresh):
        jobs = [dict(slot=s_, prev=frames[s_][cur[s_]] if fresh else None, next=frames[s_][cur[s_] + 1], prev_pts=pts, next_pts=pts) for s_ in slots]
        got = lk.track_batch(c, jobs)
        for s_, g in zip(slots, got):
            o = po.lk_track(frames[s_][cur[s_]], frames[s_][cur[s_] + 1], pts, pts)
            assert _same(g[0], o[0]) and _same(g[1], o[1]) and _same(g[2], o[2]), (s_, cur[s_], fresh, len(slots))
            cur[s_] += 1

    call(range(S), True)                   # wide, fresh
    call(range(few), False)                # narrow, chained to a wide call
    call(range(few), False)                # narrow, chained to a narrow call
    call(range(S), False)                  # wide, chained: slots 0 .. 3 come from a narrow call, the others from the first wide one
    call(range(few, 2 * few), False)       # narrow after wide for other slots
    c.close()

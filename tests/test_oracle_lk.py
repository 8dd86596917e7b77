"""CPU oracle of the pyramidal LK tracker (N1: frontend.cpp:156-166, 374-384 call cv::calcOpticalFlowPyrLK).
OpenCV is not available anywhere, so these are known-answer and property tests of the restatement
(oracle/src/lk_oracle.cpp): PARITY UNPINNED vs OpenCV, like the other OpenCV-resident arithmetic."""
import numpy as np

from tools.synth import make_stereo_pair


def test_pyr_down_known_answers(po):
    c = np.full((37, 50), 93, np.uint8)
    d = po.lk_pyr_down(c)
    assert d.shape == (19, 25) and (d == 93).all()                      # ((rows+1)/2, (cols+1)/2), weights sum to 256
    ramp = np.tile((np.arange(64) * 2).astype(np.uint8), (16, 1))
    d = po.lk_pyr_down(ramp)
    # interior: the binomial kernel preserves a linear ramp sampled at even columns
    assert np.array_equal(d[:, 2:-2], np.tile((np.arange(32) * 4).astype(np.uint8), (8, 1))[:, 2:-2])
    # reflect-101 at the left edge: taps (-2,-1,0,1,2) -> columns (2,1,0,1,2) = values (4,2,0,2,4):
    # horizontal 4 + 8 + 0 + 8 + 4 = 24, vertical x16 = 384, (384 + 128) >> 8 = 2
    assert d[0, 0] == 2


def test_scharr_known_answers(po):
    xr = np.tile(np.arange(40, dtype=np.uint8) * 3, (12, 1))
    g = po.lk_scharr(xr)
    assert (g[:, 1:-1, 0] == 3 * 2 * 16).all() and (g[..., 1] == 0).all()   # d/dx = slope * 2 * (3+10+3)
    assert (g[:, 0, 0] == 0).all() and (g[:, -1, 0] == 0).all()             # reflect-101: x-1 == x+1 at the edge
    yr = xr.T.copy()
    g = po.lk_scharr(yr)
    assert (g[1:-1, :, 1] == 96).all() and (g[..., 0] == 0).all()


def test_translation_is_recovered(po):
    L = make_stereo_pair(seed=3)[0]
    k, _ = po.orb_extract(L, prm=po.orb_params(nfeatures=200, nlevels=1))
    pts = np.stack([k["x"], k["y"]], 1).astype(np.float32)
    nxt = np.roll(np.roll(L, 4, axis=0), -7, axis=1)                     # flow = (-7, +4)
    out, st, err, top = po.lk_track(L, nxt, pts, prm=po.lk_params(use_initial_flow=0))
    assert top == 3 and st.mean() > 0.97
    d = (out - pts)[st > 0]
    assert np.abs(d - np.array([-7.0, 4.0])).max() < 0.05 and np.median(err[st > 0]) < 0.05


def test_status_rules(po):
    flat = np.full((120, 160), 77, np.uint8)
    pts = np.array([[50.0, 40.0], [80.5, 60.25]], np.float32)
    out, st, err, top = po.lk_track(flat, flat, pts)
    assert (st == 0).all()                                               # min eigenvalue below 1e-4 everywhere
    L = make_stereo_pair(seed=5, h=120, w=160, n_blobs=120)[0]
    far = np.array([[-40.0, 30.0], [30.0, 400.0], [500.0, 30.0]], np.float32)
    out, st, err, top = po.lk_track(L, L, far)
    assert (st == 0).all() and (err == 0).all()                          # window outside the image at level 0
    # a pyramid level not larger than the window ends the pyramid (buildOpticalFlowPyramid)
    tiny = make_stereo_pair(seed=6, h=40, w=60, n_blobs=30)[0]
    out, st, err, top = po.lk_track(tiny, tiny, np.array([[30.0, 20.0]], np.float32))
    assert top == 1                                                      # 40x60 -> 20x30 -> (10x15: 10 <= 11)
    # identical images: the first iteration already meets the epsilon test, the point does not move
    pts = np.array([[70.0, 50.0]], np.float32)
    out, st, err, top = po.lk_track(L, L, pts)
    assert st[0] == 1 and np.abs(out - pts).max() < 1e-3 and err[0] == 0.0

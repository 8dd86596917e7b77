"""GPU parity of the ORB front-end (pyramid, grid FAST, octree, orientation, blur, BRIEF), the row-band matcher
and the triangulation against the CPU oracle: BIT-EXACT for every integer / byte / index / f32 output, 1e-9
relative for the f64 triangulated points.  (The oracle itself is "parity unpinned vs OpenCV", see
tests/test_oracle_orb.py; what is proven here is HIP == CPU restatement.)"""
import numpy as np
import pytest

from ssvio_amd import orb as sorb
from tools.synth import KITTI_BASELINE, KITTI_K, make_stereo_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pair_small():
    return make_stereo_pair(seed=7, h=160, w=260, n_blobs=260)


@pytest.fixture(scope="module")
def pair_kitti():
    return make_stereo_pair(seed=0)


def test_pyramid_and_blur_bit_exact(ctx, po, pair_kitti):
    L = pair_kitti[0]
    ex = sorb.ORBextractor(ctx)
    ex.DetectAndCompute(L)
    rows, cols = po.level_sizes(L.shape[0], L.shape[1])
    prev = L
    for l in range(8):
        lvl = ex.stage_level(l)
        assert lvl.shape == (rows[l], cols[l])
        ref = L if l == 0 else po.resize_linear(prev, rows[l], cols[l])
        assert np.array_equal(lvl, ref), f"pyramid level {l}"
        assert np.array_equal(ex.stage_level(l, blurred=True), po.gauss7(ref)), f"blur level {l}"
        prev = ref


def test_grid_fast_candidates_bit_exact(ctx, po, pair_kitti, pair_small):
    for img in (pair_kitti[0], pair_kitti[1], pair_small[0]):
        ex = sorb.ORBextractor(ctx, nfeatures=2000)
        ex.Detect(img)
        g = ex.stage_candidates(0)
        o = po.orb_grid_fast(img)
        assert len(g) == len(o) and g.tobytes() == o.tobytes()


@pytest.mark.parametrize("nfeat", [100, 300, 2000, 3600])
def test_detect_bit_exact(ctx, po, pair_kitti, nfeat):
    """ORBextractor::Detect with the reference's three budgets (kitti_00.yaml:37-39 and the C2 setting); 3600 is
    past the LDS node-table budget of the octree kernel and takes its global-scratch variant."""
    L = pair_kitti[0]
    g = sorb.ORBextractor(ctx, nfeatures=nfeat).Detect(L)
    o = po.orb_detect(L, prm=po.orb_params(nfeatures=nfeat))
    assert len(g) == len(o) and g.tobytes() == o.tobytes()


def test_detect_with_mask_bit_exact(ctx, po, pair_kitti):
    """FrontEnd::DetectFeatures masks 21x21 boxes around existing features (frontend.cpp:304-312)."""
    L = pair_kitti[0]
    first = sorb.ORBextractor(ctx, nfeatures=300).Detect(L)
    mask = np.full(L.shape, 255, np.uint8)
    for k in first:
        x, y = int(round(float(k["x"]))), int(round(float(k["y"])))
        mask[max(y - 10, 0):y + 11, max(x - 10, 0):x + 11] = 0
    g = sorb.ORBextractor(ctx, nfeatures=100).Detect(L, mask)
    o = po.orb_detect(L, mask=mask, prm=po.orb_params(nfeatures=100))
    assert len(g) == len(o) > 0 and g.tobytes() == o.tobytes()


def test_detect_with_boxes_equals_the_rasterised_mask(ctx, po, pair_kitti):
    """ssx_orb_detect_boxes (A4: the mask of FrontEnd::DetectFeatures rasterised on the device from its rectangles, 16 bytes per
    tracked feature over PCIe instead of the 466 KB mask): the keypoints of ssx_orb_detect on the host-rasterised mask and of the
    oracle, bit for bit -- boxes clipped at all four image borders, overlapping boxes, an empty list (= no mask at all)."""
    L = pair_kitti[0]
    h, w = L.shape
    first = sorb.ORBextractor(ctx, nfeatures=300).Detect(L)
    boxes = []
    for k in first:
        x, y = float(k["x"]), float(k["y"])
        boxes.append((max(0, int(np.rint(x - 10))), max(0, int(np.rint(y - 10))), min(w - 1, int(np.rint(x + 10))), min(h - 1, int(np.rint(y + 10)))))
    boxes += [(-30, -5, 12, 9), (w - 7, h - 9, w + 40, h + 3), (100, -20, 140, 4), (-3, h - 4, 25, h + 9), (50, 60, 49, 80)]   # past the borders; an empty one
    mask = np.full(L.shape, 255, np.uint8)
    for x0, y0, x1, y1 in boxes:
        if x1 >= x0 and y1 >= y0:
            mask[max(y0, 0):min(y1, h - 1) + 1, max(x0, 0):min(x1, w - 1) + 1] = 0
    ex = sorb.ORBextractor(ctx, nfeatures=100)
    g = ex.DetectBoxes(L, np.array(boxes, np.int32))
    m = ex.Detect(L, mask)
    o = po.orb_detect(L, mask=mask, prm=po.orb_params(nfeatures=100))
    assert len(g) == len(o) > 0 and g.tobytes() == m.tobytes() == o.tobytes()
    assert ex.DetectBoxes(L, np.zeros((0, 4), np.int32)).tobytes() == ex.Detect(L).tobytes()


def test_detect_boxes_batch_equals_single_calls(ctx, pair_kitti):
    """ssx_orb_detect_boxes_batch -- one keyframe of each of n streams in one call -- returns, per image, the bits of
    ssx_orb_detect_boxes: different images, different numbers of rectangles (one image without any), batches of 5, 2 and 7 on the
    same context (the plan made for the largest batch serves the smaller ones), then a single call again."""
    from tools.synth import make_stereo_pair
    imgs = [pair_kitti[0], pair_kitti[1]] + [make_stereo_pair(seed=30 + k)[0] for k in range(5)]
    ex = sorb.ORBextractor(ctx, nfeatures=100)
    rng = np.random.default_rng(3)
    boxes = []
    for k, im in enumerate(imgs):
        nb = 0 if k == 2 else 40 + 30 * k
        x = rng.integers(0, im.shape[1] - 1, nb); y = rng.integers(0, im.shape[0] - 1, nb)
        boxes.append(np.stack([np.maximum(x - 10, 0), np.maximum(y - 10, 0), np.minimum(x + 10, im.shape[1] - 1), np.minimum(y + 10, im.shape[0] - 1)], 1).astype(np.int32))
    ones = [ex.DetectBoxes(im, bx) for im, bx in zip(imgs, boxes)]
    for sel in ([0, 1, 2, 3, 4], [5, 2], [6, 5, 4, 3, 2, 1, 0]):
        got = ex.DetectBoxesBatch([imgs[i] for i in sel], [boxes[i] for i in sel])
        for i, g in zip(sel, got):
            assert len(g) == len(ones[i]) > 0 and g.tobytes() == ones[i].tobytes(), (sel, i)
    assert ex.DetectBoxes(imgs[3], boxes[3]).tobytes() == ones[3].tobytes()


def test_triangulate_batch_equals_single_calls(ctx):
    """ssx_triangulate_batch returns per job the bits of ssx_triangulate (with and without T_wc, an empty job, 9 jobs)"""
    rng = np.random.default_rng(5)
    jobs = []
    for k in range(9):
        n = 0 if k == 4 else 50 + 77 * k
        uvL = np.stack([rng.uniform(50, 1200, n), rng.uniform(20, 350, n)], 1)
        uvR = uvL - np.stack([rng.uniform(-2, 90, n), np.zeros(n)], 1)
        T = None if k % 2 else np.array([0.01 * k, -0.02, 0.005, 1.0, 0.3 * k, -0.1, 2.0 + k])
        if T is not None:
            T[:4] /= np.linalg.norm(T[:4])
        jobs.append(dict(uvL=uvL, uvR=uvR, T_wc=T))
    ones = [sorb.triangulate(ctx, j["uvL"], j["uvR"], T_wc=j["T_wc"]) for j in jobs]
    got = sorb.triangulate_batch(ctx, jobs)
    for k, (a, b) in enumerate(zip(got, ones)):
        assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes(), k


@pytest.mark.parametrize("which", ["kitti_left", "kitti_right", "small"])
def test_extract_bit_exact(ctx, po, pair_kitti, pair_small, which):
    img = {"kitti_left": pair_kitti[0], "kitti_right": pair_kitti[1], "small": pair_small[0]}[which]
    prm = dict(nfeatures=2000, nlevels=8) if which != "small" else dict(nfeatures=300, nlevels=4)
    gk, gd = sorb.ORBextractor(ctx, nfeatures=prm["nfeatures"], nlevels=prm["nlevels"]).DetectAndCompute(img)
    ok, od = po.orb_extract(img, prm=po.orb_params(**prm))
    assert len(gk) == len(ok)
    assert gk.tobytes() == ok.tobytes()            # positions, size, angle (f32), response, octave: bit-exact
    assert np.array_equal(gd, od)                  # 256-bit descriptors


def test_extract_with_mask_and_empty(ctx, po, pair_small):
    L = pair_small[0]
    mask = np.full(L.shape, 255, np.uint8); mask[:, :120] = 0
    gk, gd = sorb.ORBextractor(ctx, nfeatures=300, nlevels=4).DetectAndCompute(L, mask)
    ok, od = po.orb_extract(L, mask=mask, prm=po.orb_params(nfeatures=300, nlevels=4))
    assert gk.tobytes() == ok.tobytes() and np.array_equal(gd, od)
    k, d = sorb.ORBextractor(ctx).DetectAndCompute(np.zeros((0, 0), np.uint8))
    assert len(k) == 0 and d.shape == (0, 32)
    flat = np.full((200, 300), 90, np.uint8)        # no texture: no corners at either threshold
    k, d = sorb.ORBextractor(ctx, nfeatures=300, nlevels=4).DetectAndCompute(flat)
    assert len(k) == 0


def test_stereo_match_bit_exact(ctx, po, pair_kitti):
    L, R, _ = pair_kitti
    kL, dL = po.orb_extract(L); kR, dR = po.orb_extract(R)
    for mp in (dict(), dict(band_px=1.0, max_dist=50), dict(max_octave_diff=0, max_disp=60.0)):
        gi, gd = sorb.stereo_match(ctx, kL, dL, kR, dR, sorb.match_params(**mp))
        oi, od = po.stereo_match(kL, dL, kR, dR, po.match_params(**mp))
        assert np.array_equal(gi, oi) and np.array_equal(gd, od)
    assert (gi >= 0).sum() > 100
    # edge cases: empty right / empty left
    gi, gd = sorb.stereo_match(ctx, kL, dL, kR[:0], dR[:0])
    assert (gi == -1).all() and (gd == 257).all()
    assert len(sorb.stereo_match(ctx, kL[:0], dL[:0], kR, dR)[0]) == 0
    bi, bd = sorb.bf_match(ctx, dL[:500], dR)
    oi, od = po.bf_match(dL[:500], dR)
    assert np.array_equal(bi, oi) and np.array_equal(bd, od)


def test_triangulate_matches_oracle_and_reference_golden(ctx, po):
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_golden.npz"))
    xyz, ok = sorb.triangulate(ctx, G["tri_uvL"], G["tri_uvR"])
    np.testing.assert_array_equal(ok, G["tri_ok"])                      # vs the REAL reference function
    m = G["tri_ok"].astype(bool)
    np.testing.assert_allclose(xyz[m], G["tri_xyz"][m], rtol=1e-9)
    # The device routine is a latency-oriented Jacobi (round-robin pair order, fma, early exit) and the oracle the
    # plain cyclic one: same singular vector, different rounding.  Tolerance 1e-9 relative / 1e-9 m absolute, five
    # orders tighter than the 1e-4 residual bound of the north star.
    o = po.triangulate(G["tri_uvL"], G["tri_uvR"], KITTI_K, KITTI_BASELINE)
    np.testing.assert_array_equal(ok, o["ok"])
    print("triangulate max |dev - oracle| =", np.abs(xyz - o["xyz"]).max())
    np.testing.assert_allclose(xyz, o["xyz"], rtol=1e-9, atol=1e-9)
    T = np.array([0.01, -0.02, 0.03, 0.9993, 1.0, -2.0, 0.5]); T[:4] /= np.linalg.norm(T[:4])
    xyz2, _ = sorb.triangulate(ctx, G["tri_uvL"], G["tri_uvR"], T_wc=T)
    o2 = po.triangulate(G["tri_uvL"], G["tri_uvR"], KITTI_K, KITTI_BASELINE, T_wc=T)
    np.testing.assert_allclose(xyz2, o2["xyz"], rtol=1e-9, atol=1e-9)


def test_stereo_frame_end_to_end(ctx, po, pair_kitti):
    """extract + match + triangulate in one device-resident call == the oracle pipeline, bit for bit"""
    L, R, disp = pair_kitti
    r = sorb.stereo_frame(ctx, L, R)
    kL, dL = po.orb_extract(L); kR, dR = po.orb_extract(R)
    assert r["kL"].tobytes() == kL.tobytes() and r["kR"].tobytes() == kR.tobytes()
    assert np.array_equal(r["dL"], dL) and np.array_equal(r["dR"], dR)
    oi, od = po.stereo_match(kL, dL, kR, dR)
    assert np.array_equal(r["match_idx"], oi) and np.array_equal(r["match_dist"], od)
    m = oi >= 0
    uvL = np.stack([kL["x"][m], kL["y"][m]], 1).astype(np.float64)
    uvR = np.stack([kR["x"][oi[m]], kR["y"][oi[m]]], 1).astype(np.float64)
    t = po.triangulate(uvL, uvR, KITTI_K, KITTI_BASELINE)
    np.testing.assert_array_equal(r["ok"][m], t["ok"])
    np.testing.assert_allclose(r["xyz"][m], t["xyz"], rtol=1e-9, atol=1e-9)
    assert r["n_matched"] == int(m.sum()) and r["n_triangulated"] == int(t["ok"].sum())
    # sanity against the synthetic ground truth: depth = bf / disparity
    good = m & (r["ok"] == 1)
    z = r["xyz"][good, 2]
    d_true = disp[np.clip(np.round(kL["y"][good]).astype(int), 0, 375), np.clip(np.round(kL["x"][good]).astype(int), 0, 1240)]
    rel = np.abs(z - 386.1448 / d_true) / (386.1448 / d_true)
    assert np.median(rel) < 0.15      # sanity only (scaled octaves quantise the disparity); parity is asserted above


def test_stereo_batch_matches_single_frames(ctx, po):
    """the batched device-resident path (bench.py's path) gives, per pair, exactly the single-frame results"""
    import torch
    pairs = 3
    imgs = np.stack([np.stack(make_stereo_pair(seed=s)[:2]) for s in range(pairs)])     # [pairs][2][H][W]
    dev = torch.from_numpy(imgs).to("cuda:0")
    torch.cuda.synchronize()
    counts = sorb.stereo_batch_dev(ctx, dev.data_ptr(), pairs, imgs.shape[3], imgs.shape[2], imgs.shape[3])
    for p in range(pairs):
        single = sorb.stereo_frame(ctx, imgs[p, 0], imgs[p, 1])
        # re-run the batch (the single-frame call re-planned the workspace) and fetch pair p
        counts = sorb.stereo_batch_dev(ctx, dev.data_ptr(), pairs, imgs.shape[3], imgs.shape[2], imgs.shape[3])
        b = sorb.stereo_batch_fetch(ctx, p, cap=4096)
        assert counts[p].tolist() == [len(single["kL"]), len(single["kR"]), single["n_matched"], single["n_triangulated"]]
        for key in ("kL", "kR"):
            assert b[key].tobytes() == single[key].tobytes()
        for key in ("dL", "dR", "match_idx", "match_dist", "ok"):
            assert np.array_equal(b[key], single[key])
        np.testing.assert_array_equal(b["xyz"], single["xyz"])


def test_describe_at_bit_exact(ctx, po, pair_kitti):
    """loop-closing descriptor path: every tracked feature is replicated over octaves 0..7 at the
    same image location (loopclosing.cpp:607-619), screened and described (orbextractor.cpp:844-991)."""
    L = pair_kitti[0]
    base = po.orb_detect(L, prm=po.orb_params(nfeatures=300))
    rep = np.zeros(len(base) * 8, dtype=po.KP_DTYPE)
    for o in range(8):
        rep[o::8] = base
        rep["octave"][o::8] = o
    rep = np.concatenate([rep, rep[:3]])
    rep["octave"][-1] = 9                       # out-of-range octave: dropped
    rep["x"][-2] = 5.0                          # too close to the border: dropped
    ex = sorb.ORBextractor(ctx, nfeatures=2000)
    gk, gd = ex.ScreenAndComputeKPsParams_CalcDescriptors(L, rep)
    ok, od = po.orb_describe_at(L, rep, prm=po.orb_params(nfeatures=2000))
    assert 0 < len(ok) < len(rep)
    assert len(gk) == len(ok) and gk.tobytes() == ok.tobytes() and np.array_equal(gd, od)
    e_k, e_d = ex.ScreenAndComputeKPsParams_CalcDescriptors(L, rep[:0])
    assert len(e_k) == 0 and e_d.shape == (0, 32)


@pytest.mark.parametrize("scale,nlevels,nfeat", [(1.5, 4, 400), (2.0, 3, 300), (2.3, 3, 300), (1.1, 6, 500)])
def test_extract_other_pyramids_bit_exact(ctx, po, pair_small, pair_kitti, scale, nlevels, nfeat):
    """other scale factors / level counts than the yaml defaults: the resize tables (incl. the byte-load fallback of
    k_resize for quads spanning more than 8 source bytes, scale > 2), the per-level budgets and the cell grids"""
    for img in (pair_small[0], pair_kitti[1][40:300, 100:900]):
        ex = sorb.ORBextractor(ctx, nfeatures=nfeat, scaleFactor=scale, nlevels=nlevels)
        gk, gd = ex.DetectAndCompute(np.ascontiguousarray(img))
        ok, od = po.orb_extract(np.ascontiguousarray(img), prm=po.orb_params(nfeatures=nfeat, scale_factor=scale, nlevels=nlevels))
        assert len(gk) == len(ok) and gk.tobytes() == ok.tobytes() and np.array_equal(gd, od)


def test_tiny_budgets_return_more_than_the_budget(ctx, po):
    """a wide image with a handful of features per level: the first quadtree subdivision already makes up to
    4 * nIni nodes, so a level returns more keypoints than its budget + 3 (found by tools/fuzz_parity.py: the output
    arrays used to be sized for budget + 4 per level)"""
    from tools.synth import make_stereo_pair
    L = make_stereo_pair(seed=1008, h=368, w=1341, n_blobs=4100)[0]
    for nfeat, nlev, sf in ((50, 8, 1.1), (8, 8, 1.2), (20, 3, 1.5)):
        ex = sorb.ORBextractor(ctx, nfeatures=nfeat, scaleFactor=sf, nlevels=nlev, iniThFAST=35, minThFAST=3)
        gk, gd = ex.DetectAndCompute(L)
        ok, od = po.orb_extract(L, prm=po.orb_params(nfeatures=nfeat, scale_factor=sf, nlevels=nlev, ini_th=35, min_th=3))
        assert len(gk) == len(ok) and gk.tobytes() == ok.tobytes() and np.array_equal(gd, od)
        assert len(gk) > nfeat + 3 * nlev
        dk = ex.Detect(L); do = po.orb_detect(L, prm=po.orb_params(nfeatures=nfeat, scale_factor=sf, nlevels=nlev, ini_th=35, min_th=3))
        assert dk.tobytes() == do.tobytes()


def test_triangulation_without_positive_disparity(ctx, po):
    """uL == uR is a point at infinity: the homogeneous w is rounding noise, so z > 0 is decided on the disparity --
    flag 0 and a zeroed point, the same in the kernel and in the oracle (found by tools/fuzz_parity2.py); with T_wc the
    zeroed camera point maps to the camera centre"""
    uvL = np.array([[100.0, 50.0], [640.25, 180.5], [300.0, 200.0], [900.0, 100.0], [20.0, 300.0]])
    uvR = uvL - np.array([[0.0, 0.0], [0.0, 0.3], [-3.0, 0.0], [1e-9, 0.0], [25.0, 0.1]])
    for T in (None, np.array([0, 0, np.sin(0.1), np.cos(0.1), 1.0, -2.0, 0.5])):
        xyz, ok = sorb.triangulate(ctx, uvL, uvR, T_wc=T)
        t = po.triangulate(uvL, uvR, KITTI_K, KITTI_BASELINE, T_wc=T)
        assert np.array_equal(np.asarray(ok).astype(bool), t["ok"].astype(bool))
        assert np.asarray(ok).tolist() == [0, 0, 0, 1, 1]
        z = np.zeros(3) if T is None else T[4:]
        assert np.array_equal(np.asarray(xyz)[:3], np.tile(z, (3, 1))) and np.array_equal(t["xyz"][:3], np.tile(z, (3, 1)))
        assert np.abs(np.asarray(xyz)[4] - t["xyz"][4]).max() < 1e-9 * np.abs(t["xyz"][4]).max()


def test_more_levels_than_the_image_has_pixels_for(ctx, po):
    """48x79 at scale 2.0 has no pixels left from level 6 on (OpenCV's resize would throw in the reference): the empty
    levels contribute nothing, the others are as always (found by tools/fuzz_parity.py)"""
    from tools.synth import make_stereo_pair
    for (h, w, nfeat, nlev, sf) in ((48, 79, 300, 8, 2.0), (63, 69, 2000, 8, 2.0), (40, 40, 100, 8, 1.5)):
        L = make_stereo_pair(seed=h + w, h=h, w=w, n_blobs=max(h * w // 100, 8))[0]
        ex = sorb.ORBextractor(ctx, nfeatures=nfeat, scaleFactor=sf, nlevels=nlev, iniThFAST=10, minThFAST=3)
        gk, gd = ex.DetectAndCompute(L)
        ok, od = po.orb_extract(L, prm=po.orb_params(nfeatures=nfeat, scale_factor=sf, nlevels=nlev, ini_th=10, min_th=3))
        assert len(gk) == len(ok) and gk.tobytes() == ok.tobytes() and np.array_equal(gd, od)


def test_low_contrast_cells_fall_back_to_min_threshold(ctx, po, pair_small):
    """cells without a corner at iniThFAST are re-run at minThFAST (orbextractor.cpp:598-606): a low-contrast
    image makes most cells take the second pass, a high iniThFAST all of them"""
    L = pair_small[0]
    soft = (L.astype(np.float32) * 0.22 + 90).astype(np.uint8)
    for img, ini, mn in ((soft, 20, 7), (L, 120, 7), (L, 20, 20), (soft, 7, 3)):
        ex = sorb.ORBextractor(ctx, nfeatures=300, nlevels=4, iniThFAST=ini, minThFAST=mn)
        gk, gd = ex.DetectAndCompute(img)
        ok, od = po.orb_extract(img, prm=po.orb_params(nfeatures=300, nlevels=4, ini_th=ini, min_th=mn))
        assert len(gk) == len(ok) and gk.tobytes() == ok.tobytes() and np.array_equal(gd, od)
        gdet = ex.Detect(img)
        odet = po.orb_detect(img, prm=po.orb_params(nfeatures=300, nlevels=4, ini_th=ini, min_th=mn))
        assert gdet.tobytes() == odet.tobytes()


def test_matcher_and_triangulation_edge_cases(ctx, po, pair_small):
    """no keypoints on one side, a single keypoint, everything out of the row band"""
    L, R = pair_small[0], pair_small[1]
    kL, dL = po.orb_extract(L, prm=po.orb_params(nfeatures=300, nlevels=4))
    kR, dR = po.orb_extract(R, prm=po.orb_params(nfeatures=300, nlevels=4))
    empty_k, empty_d = kR[:0], dR[:0]
    for a, b, c, d in ((kL, dL, empty_k, empty_d), (empty_k, empty_d, kR, dR), (kL[:1], dL[:1], kR, dR), (kL, dL, kR[:1], dR[:1])):
        gi, gd = sorb.stereo_match(ctx, a, b, c, d)
        oi, od = po.stereo_match(a, b, c, d)
        assert np.array_equal(gi, oi) and np.array_equal(gd, od)
    far = kR.copy(); far["y"] += 500.0                      # no right keypoint inside any left keypoint's band
    gi, gd = sorb.stereo_match(ctx, kL, dL, far, dR)
    oi, od = po.stereo_match(kL, dL, far, dR)
    assert np.array_equal(gi, oi) and (gi < 0).all()
    xyz, ok = sorb.triangulate(ctx, np.zeros((0, 2)), np.zeros((0, 2)))
    assert xyz.shape == (0, 3) and ok.shape == (0,)
    # zero and negative disparity: behind / at infinity -> rejected like the oracle
    uvL = np.array([[300.0, 100.0], [300.0, 100.0], [10.0, 20.0]]); uvR = np.array([[300.0, 100.0], [310.0, 100.0], [9.0, 21.5]])
    xyz, ok = sorb.triangulate(ctx, uvL, uvR)
    o = po.triangulate(uvL, uvR, KITTI_K, KITTI_BASELINE)
    assert np.array_equal(ok, o["ok"])
    m = o["ok"].astype(bool)
    np.testing.assert_allclose(xyz[m], o["xyz"][m], rtol=1e-9, atol=1e-9)


def test_other_image_sizes(ctx, po):
    """641x479 (octree keys in LDS, <= 16384 candidates per level) and 1280x720 / 1920x1080 (keys in the global scratch
    block, <= 65536 candidates per level) are all bit-exact"""
    for h, w, nb in ((479, 641, 1500), (720, 1280, 6000), (1080, 1920, 12000)):
        L = make_stereo_pair(seed=2, h=h, w=w, n_blobs=nb)[0]
        gk, gd = sorb.ORBextractor(ctx, nfeatures=2000).DetectAndCompute(L)
        ok, od = po.orb_extract(L, prm=po.orb_params(nfeatures=2000))
        assert len(gk) == len(ok) and gk.tobytes() == ok.tobytes() and np.array_equal(gd, od)
        gdet = sorb.ORBextractor(ctx, nfeatures=300).Detect(L)
        odet = po.orb_detect(L, prm=po.orb_params(nfeatures=300))
        assert gdet.tobytes() == odet.tobytes()


def test_degenerate_small_images(ctx, po):
    """pyramid levels that shrink below the border (no cells), images barely larger than the border: same (possibly
    empty) result as the oracle, no error"""
    for h, w, nl, nf in ((100, 130, 8, 500), (60, 90, 8, 200), (41, 41, 8, 50), (45, 300, 6, 300), (300, 45, 6, 300)):
        L = make_stereo_pair(seed=3, h=h, w=w, n_blobs=max(h * w // 120, 10))[0]
        gk, gd = sorb.ORBextractor(ctx, nfeatures=nf, nlevels=nl).DetectAndCompute(L)
        ok, od = po.orb_extract(L, prm=po.orb_params(nfeatures=nf, nlevels=nl))
        assert len(gk) == len(ok) and gk.tobytes() == ok.tobytes() and np.array_equal(gd, od)


def test_streamed_batches_equal_resident_batches(ctx):
    """ssx_stereo_batch_upload / _run / _counts (batches that arrive from the host, one upload kept ahead on the library's copy
    stream, two device buffers): every batch's counts and per-pair results are those of ssx_stereo_batch_dev on the same images"""
    import torch
    from tools.synth import make_stereo_pair
    B, H, W = 4, 200, 320
    prm = sorb.OrbParams(300, 1.2, 4, 20, 7)
    batches = [np.stack([np.stack(make_stereo_pair(seed=50 + 10 * k + i, h=H, w=W, n_blobs=400)[:2]) for i in range(B)]) for k in range(3)]
    want = []
    for hb in batches:
        dev = torch.from_numpy(hb).cuda()
        c = sorb.stereo_batch_dev(ctx, dev.data_ptr(), B, W, H, W, orb=prm).copy()
        want.append((c, [sorb.stereo_batch_fetch(ctx, p, 2048) for p in range(B)]))
    pinned = [torch.from_numpy(hb).pin_memory() for hb in batches]
    st = sorb.StereoStream(ctx, B, H, W, orb=prm)
    st.upload(pinned[0].data_ptr())
    for k in range(3):
        if k + 1 < 3:
            st.upload(pinned[k + 1].data_ptr())                       # one ahead
        st.run()
        c = st.wait_counts()
        assert np.array_equal(c, want[k][0]), k
        for p in range(B):
            got, ref = sorb.stereo_batch_fetch(ctx, p, 2048), want[k][1][p]
            for key in ("kL", "kR"):
                assert got[key].tobytes() == ref[key].tobytes()
            for key in ("dL", "dR", "match_idx", "match_dist", "xyz", "ok"):
                assert np.array_equal(got[key], ref[key]), (k, p, key)
    # collecting one batch behind: run(k + 1) before counts(k) -- counts() returns the OLDEST uncollected batch
    st.upload(pinned[0].data_ptr())
    st.run()
    for k in range(1, 3):
        st.upload(pinned[k].data_ptr())
        st.run()                                                      # batch k runs ...
        assert np.array_equal(st.wait_counts(), want[k - 1][0]), k    # ... while the counts of batch k - 1 are collected
    assert np.array_equal(st.wait_counts(), want[2][0])
    from ssvio_amd._lib import SsxError
    with pytest.raises(SsxError):
        st.wait_counts()                                              # nothing left to collect
    # the one-call form, and misuse: a third upload while two are waiting, a run without an upload
    st.enqueue(pinned[1].data_ptr())
    assert np.array_equal(st.wait_counts(), want[1][0])
    from ssvio_amd._lib import SsxError
    with pytest.raises(SsxError):
        st.run()
    st.upload(pinned[0].data_ptr()); st.upload(pinned[1].data_ptr())
    with pytest.raises(SsxError):
        st.upload(pinned[2].data_ptr())
    st.run(); st.run()
    with pytest.raises(SsxError):
        st.upload(pinned[2].data_ptr()); st.run()                     # a third run while the counts of two batches are waiting
    assert np.array_equal(st.wait_counts(), want[0][0]) and np.array_equal(st.wait_counts(), want[1][0])
    st.run()
    assert np.array_equal(st.wait_counts(), want[2][0])

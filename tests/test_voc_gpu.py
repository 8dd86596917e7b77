"""GPU parity of the bag-of-words path (N2: ORBVocabulary::transform / score, loopclosing.cpp:84,633) against the CPU
oracle: per-feature word ids and weights and the BowVector's ids are BIT-EXACT, its values are the same doubles (same
accumulation and normalisation order), the score is the same double."""
import numpy as np
import pytest

from ssvio_amd import voc as svoc
from ssvio_amd._lib import SsxError
from tools.synth import make_stereo_pair, make_vocabulary, write_vocabulary_text

pytestmark = pytest.mark.gpu


def _features(po, seed, n=1500):
    img = make_stereo_pair(seed=seed)[0]
    _, d = po.orb_extract(img, prm=po.orb_params(nfeatures=n))
    return d


@pytest.mark.parametrize("k,L,weighting", [(10, 3, 0), (10, 4, 0), (7, 2, 1), (3, 5, 2), (20, 2, 3), (2, 8, 0)])
def test_transform_and_score_match_the_oracle(ctx, po, k, L, weighting):
    voc = make_vocabulary(k=k, L=L, seed=k + L)
    V = svoc.Vocabulary.from_arrays(ctx, k, L, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"], weighting=weighting)
    assert (V.k, V.L, V.n_nodes, V.n_words, V.weighting) == (k, L, len(voc["parent"]), int(voc["is_leaf"].sum()), weighting)
    bows = []
    for seed in (1, 2):
        feats = _features(po, seed)
        ids, vals, words, weights = V.transform(feats, with_features=True)
        ow, owt = po.voc_transform_features(voc, feats)
        assert np.array_equal(words, ow) and weights.tobytes() == owt.tobytes()
        oi, ov = po.bow_vector(ow, owt, weighting=weighting)
        assert np.array_equal(ids, oi) and vals.tobytes() == ov.tobytes()
        assert abs(vals.sum() - 1.0) < 1e-12
        bows.append(((ids, vals), (oi, ov)))
    s_gpu = V.score(bows[0][0], bows[1][0]); s_cpu = po.bow_score_l1(bows[0][1], bows[1][1])
    assert s_gpu == s_cpu and 0.0 <= s_gpu <= 1.0
    self_gpu = V.score(bows[0][0], bows[0][0])
    assert self_gpu == po.bow_score_l1(bows[0][1], bows[0][1]) and abs(self_gpu - 1.0) < 1e-12
    V.close()


def test_text_vocabulary_and_edge_cases(ctx, po, tmp_path):
    voc = make_vocabulary(k=10, L=3, seed=5)
    path = tmp_path / "voc.txt"
    write_vocabulary_text(path, voc)
    V = svoc.Vocabulary.loadFromTextFile(ctx, path)                     # the ORBvoc.txt format
    assert (V.k, V.L, V.n_nodes, V.n_words) == (10, 3, 1111, 1000)
    feats = _features(po, 3, 700)
    ids, vals = V.transform(feats)
    oi, ov = po.voc_transform(voc, feats)
    assert np.array_equal(ids, oi) and vals.tobytes() == ov.tobytes()
    e_ids, e_vals = V.transform(np.zeros((0, 32), np.uint8))
    assert len(e_ids) == 0 and len(e_vals) == 0
    one = V.transform(feats[:1])
    assert len(one[0]) <= 1
    V.close()
    empty = svoc.Vocabulary.from_arrays(ctx, 10, 3, [-1], [0], np.zeros((1, 32), np.uint8), [0.0])
    assert empty.transform(feats)[0].size == 0                          # no vocabulary: empty BowVector
    _, _, w_e, wt_e = empty.transform(feats, with_features=True)        # per-feature outputs as ssx.h documents: -1 / 0
    assert (w_e == -1).all() and (wt_e == 0.0).all()
    empty.close()
    with pytest.raises(SsxError):
        svoc.Vocabulary.loadFromTextFile(ctx, tmp_path / "missing.txt")
    (tmp_path / "bad.txt").write_text("this is not a vocabulary\n1 2 3\n")
    with pytest.raises(SsxError):
        svoc.Vocabulary.loadFromTextFile(ctx, tmp_path / "bad.txt")
    with pytest.raises(SsxError):
        svoc.Vocabulary.from_arrays(ctx, 10, 3, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"], scoring=1)     # L2: unsupported
    bad_parent = voc["parent"].copy(); bad_parent[5] = 900
    with pytest.raises(SsxError):
        svoc.Vocabulary.from_arrays(ctx, 10, 3, bad_parent, voc["is_leaf"], voc["desc"], voc["weight"])


def test_loop_candidate_ranking(ctx, po):
    """DetectLoop (loopclosing.cpp:72-105) in miniature: the database keyframe showing the same place scores highest"""
    voc = make_vocabulary(k=10, L=4, seed=9)
    V = svoc.Vocabulary.from_arrays(ctx, 10, 4, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
    places = [make_stereo_pair(seed=s) for s in (11, 12, 13, 14)]
    db = [V.transform(po.orb_extract(p[0], prm=po.orb_params(nfeatures=1000))[1]) for p in places]
    rng = np.random.default_rng(0)
    again = np.clip(places[2][0].astype(np.int16) + rng.integers(-2, 3, places[2][0].shape), 0, 255).astype(np.uint8)   # place 2 revisited
    query = V.transform(po.orb_extract(again, prm=po.orb_params(nfeatures=1000))[1])
    scores = [V.score(query, b) for b in db]
    assert int(np.argmax(scores)) == 2 and scores[2] > 1.5 * max(scores[:2] + scores[3:]), scores
    V.close()

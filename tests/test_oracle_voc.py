"""Known answers for the bag-of-words restatement (oracle/src/voc_oracle.cpp: DBoW2 tree descent, BowVector,
L1 score).  The real DBoW2 cannot be built here (its FORB class is written on cv::Mat), the reference has no test for
it: these hand-checkable cases ARE the pin ("parity unpinned vs DBoW2")."""
import numpy as np

from oracle import pyoracle as po


def _desc(bits_set):
    d = np.zeros(256, np.uint8); d[list(bits_set)] = 1
    return np.packbits(d)


def _tiny_voc():
    """root -> A (bits 0..7 set), B (no bits); A -> A1 (bits 0..7 + 100..103), A2 (bits 0..3); B -> B1 (bit 200), B2 (bits 200..215)"""
    nodes = [(-1, 0, set(), 0.0),
             (0, 0, set(range(8)), 0.0), (0, 0, set(), 0.0),
             (1, 1, set(range(8)) | set(range(100, 104)), 2.0), (1, 1, set(range(4)), 3.0),
             (2, 1, {200}, 5.0), (2, 1, set(range(200, 216)), 0.0)]         # B2 is a stopped word
    return dict(k=2, L=2, parent=np.array([n[0] for n in nodes], np.int32), is_leaf=np.array([n[1] for n in nodes], np.uint8),
                desc=np.stack([_desc(n[2]) for n in nodes]), weight=np.array([n[3] for n in nodes]))


def test_tree_descent_known_answers():
    po.build()
    voc = _tiny_voc()
    feats = np.stack([
        _desc(set(range(8)) | {100, 101}),     # -> A (0 vs 10), then A1 (2) vs A2 (6)                   word 0 weight 2
        _desc(set(range(4))),                  # level 1: A 4, B 4: tie -> the FIRST child, A; then A2     word 1 weight 3
        _desc({200, 201}),                     # -> B (2 vs 10), B1 (1) vs B2 (14)                        word 2 weight 5
        _desc(set(range(200, 216))),           # -> B, B2 (0)                                            word 3 weight 0 (stopped)
        _desc(set(range(8)) | set(range(100, 104)) | set(range(4, 8))),  # A; A1 (0) vs A2 (8)            word 0
    ])
    word, w = po.voc_transform_features(voc, feats)
    assert word.tolist() == [0, 1, 2, 3, 0] and w.tolist() == [2.0, 3.0, 5.0, 0.0, 2.0]
    # TF_IDF: word 0 twice -> 4, word 1 -> 3, word 2 -> 5, stopped word dropped; L1 norm 12
    ids, vals = po.bow_vector(word, w, weighting=0)
    assert ids.tolist() == [0, 1, 2] and np.allclose(vals, [4 / 12, 3 / 12, 5 / 12], rtol=0, atol=1e-16)
    # IDF: first occurrence only -> 2, 3, 5 / 10
    ids, vals = po.bow_vector(word, w, weighting=2)
    assert ids.tolist() == [0, 1, 2] and np.allclose(vals, [0.2, 0.3, 0.5], rtol=0, atol=1e-16)
    e_ids, e_vals = po.bow_vector(word[:0], w[:0])
    assert len(e_ids) == 0 and len(e_vals) == 0
    empty = dict(k=2, L=1, parent=np.array([-1], np.int32), is_leaf=np.array([0], np.uint8), desc=np.zeros((1, 32), np.uint8), weight=np.zeros(1))
    word, w = po.voc_transform_features(empty, feats)
    assert (word == -1).all() and (w == 0).all()


def test_l1_score_known_answers():
    po.build()
    a = (np.array([0, 1, 2], np.int32), np.array([4 / 12, 3 / 12, 5 / 12]))
    assert po.bow_score_l1(a, a) == 1.0                                   # identical normalised vectors
    b = (np.array([5, 9], np.int32), np.array([0.5, 0.5]))
    assert po.bow_score_l1(a, b) == 0.0                                   # no common word
    c = (np.array([1, 2, 7], np.int32), np.array([0.25, 0.25, 0.5]))
    # common words 1 and 2: (|.25-.25| - .25 - .25) + (|5/12 - .25| - 5/12 - .25) = -0.5 + (1/6 - 2/3) = -1.0 -> 0.5
    assert abs(po.bow_score_l1(a, c) - 0.5) < 1e-15 and abs(po.bow_score_l1(c, a) - 0.5) < 1e-15
    assert po.bow_score_l1(a, (np.zeros(0, np.int32), np.zeros(0))) == 0.0


def test_random_vocabulary_properties():
    """a 10^3 vocabulary: every descent ends in a leaf, the leaf is the arg-min path, values are L1-normalised, score in [0, 1]"""
    from tools.synth import make_vocabulary
    po.build()
    voc = make_vocabulary(k=10, L=3, seed=1)
    rng = np.random.default_rng(0)
    n_nodes = len(voc["parent"])
    assert n_nodes == 1 + 10 + 100 + 1000 and int(voc["is_leaf"].sum()) == 1000
    leaves = np.nonzero(voc["is_leaf"])[0]
    feats = voc["desc"][rng.choice(leaves, 300)].copy()
    flip = rng.random((300, 256)) < 0.03
    feats = np.packbits(np.unpackbits(feats, axis=1) ^ flip.astype(np.uint8), axis=1)
    word, w = po.voc_transform_features(voc, feats)
    assert ((word >= 0) & (word < 1000)).all()
    # brute-force the greedy descent in numpy for a few features
    pop = np.array([bin(i).count("1") for i in range(256)])
    kids = {p: np.nonzero(voc["parent"] == p)[0] for p in range(n_nodes)}
    word_of = -np.ones(n_nodes, int); word_of[leaves] = np.arange(1000)
    for f in range(40):
        node = 0
        while len(kids[node]):
            d = pop[voc["desc"][kids[node]] ^ feats[f]].sum(1)
            node = kids[node][np.argmin(d)]                                 # argmin = first minimum
        assert word_of[node] == word[f] and voc["weight"][node] == w[f]
    ids, vals = po.bow_vector(word, w)
    assert (np.diff(ids) > 0).all() and abs(vals.sum() - 1.0) < 1e-12 and (vals > 0).all()
    other = po.voc_transform(voc, feats[::-1][:150])
    s = po.bow_score_l1((ids, vals), other)
    assert 0.0 < s < 1.0 and abs(s - po.bow_score_l1(other, (ids, vals))) < 1e-15


def test_bow_vector_matches_the_reference_bowvector(po):
    """the BowVector half of transform() against vectors produced by the reference's OWN DBoW2::BowVector (BowVector.cpp
    compiled into oracle/_ref; tests/golden/ref_bow.npz by make_golden.py): same ids, bit-identical values for TF_IDF / TF
    (addWeight, input order) and IDF / BINARY (addIfNotExist), L1 normalisation.  The GPU path's BowVector is compared
    bit-for-bit with this oracle function in tests/test_voc_gpu.py."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import bow_inputs
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_bow.npz"))
    for case in range(4):
        word, weight = bow_inputs(case)
        for weighting in range(4):
            ids, vals = po.bow_vector(word, weight, weighting)
            assert np.array_equal(ids, G[f"bow{case}_w{weighting}_ids"]), (case, weighting)
            assert vals.tobytes() == G[f"bow{case}_w{weighting}_vals"].tobytes(), (case, weighting)
            if po.have_ref() and os.path.exists(os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle", "_ref", "libssvio_ref.so")):
                li, lv = po.bow_vector(word, weight, weighting, which="ref")
                assert np.array_equal(li, ids) and lv.tobytes() == vals.tobytes()

"""The constant tables of the ORB extractor, taken from the reference's own source and checked against the product's and the
oracle's copies.

Run in the BUILD container (where /root/reference is mounted):

    python tests/golden/check_tables.py            # compares, and (re)writes tests/golden/ref_tables.npz

  * the rBRIEF sampling pattern: the 256 x 4 integers of /root/reference/src/ssvio/orbpattern.cpp:9-266, PARSED from the file
    (numbers only -- data, not code), against ssvio_amd/csrc/brief_pattern.inc (product) and oracle/src/brief_pattern.inc;
  * the tables the ORBextractor constructor computes (/root/reference/src/ssvio/orbextractor.cpp:127-192): mvScaleFactor,
    mvInvScaleFactor, mnFeaturesPerLevel, umax -- recomputed here in numpy float32 / float64 arithmetic statement by statement
    (cvRound = round half to even, cvFloor, cvCeil), against the oracle's tables (orc_features_per_level, orc_umax,
    orc_level_sizes).  The product's device tables are pinned through the oracle by the GPU tests (byte-equal keypoints).

ref_tables.npz travels to the GPU box (no /root/reference there); tests/test_oracle_orb.py::test_tables_* compare the oracle
and the product's pattern file with it everywhere, and re-derive it from the reference where the reference is present.
"""
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
NPZ = os.path.join(HERE, "ref_tables.npz")
CASES = [(2000, 1.2, 8), (1000, 1.2, 8), (500, 1.2, 8), (300, 1.2, 8), (100, 1.2, 8), (300, 1.2, 4), (1500, 1.5, 5), (800, 2.0, 3)]


def parse_ints(path, first_line=None, last_line=None):
    """every integer literal of a C initialiser list, comments stripped"""
    text = open(path).read().splitlines()
    if first_line is not None:
        text = text[first_line - 1:last_line]
    body = re.sub(r"/\*.*?\*/", " ", "\n".join(text), flags=re.S)
    body = re.sub(r"//[^\n]*", " ", body)
    if "{" in body:
        body = body[body.index("{") + 1:]
    if "}" in body:
        body = body[:body.index("}")]
    return np.array([int(t) for t in re.findall(r"-?\d+", body)], dtype=np.int64)


def reference_pattern():
    v = parse_ints(os.path.join(REF, "src/ssvio/orbpattern.cpp"), 9, 266)
    assert v.size == 1024, v.size
    return v.reshape(256, 4).astype(np.int8)


def cv_round(x):
    return int(np.rint(x))                                              # cvRound: lrint, round half to even


def ctor_tables(nfeatures, scale_factor, nlevels):
    """orbextractor.cpp:133-192, statement by statement, in the arithmetic types the reference uses"""
    f32 = np.float32
    scale = np.zeros(nlevels, f32); scale[0] = f32(1.0)
    for i in range(1, nlevels):
        scale[i] = f32(scale[i - 1] * f32(scale_factor))               # :140
    inv = np.array([f32(1.0) / s for s in scale], dtype=f32)            # :148
    factor = f32(1.0) / f32(scale_factor)                               # :156
    n_desired = f32(f32(nfeatures) * (f32(1) - factor) / (f32(1) - f32(np.float64(factor) ** np.float64(nlevels))))   # :157-158
    feats, total = [], 0
    for _ in range(nlevels - 1):                                        # :161-166
        feats.append(cv_round(n_desired)); total += feats[-1]
        n_desired = f32(n_desired * factor)
    feats.append(max(nfeatures - total, 0))                             # :167
    hp = 15
    umax = [0] * (hp + 1)
    vmax = int(np.floor(f32(hp) * np.sqrt(f32(2.0)) / f32(2) + f32(1)))      # :177 cvFloor(HALF_PATCH_SIZE * sqrt(2.f) / 2 + 1)
    vmin = int(np.ceil(f32(hp) * np.sqrt(f32(2.0)) / f32(2)))               # :178
    hp2 = float(hp * hp)
    for v in range(vmax + 1):
        umax[v] = cv_round(np.sqrt(hp2 - v * v))                        # :181
    v0 = 0
    for v in range(hp, vmin - 1, -1):                                   # :184-190
        while umax[v0] == umax[v0 + 1]:
            v0 += 1
        umax[v] = v0
        v0 += 1
    return scale, inv, np.array(feats, dtype=np.int32), np.array(umax, dtype=np.int32)


def level_sizes(rows, cols, inv):
    return (np.array([cv_round(np.float32(rows) * s) for s in inv], dtype=np.int32),     # orbextractor.cpp:998-999
            np.array([cv_round(np.float32(cols) * s) for s in inv], dtype=np.int32))


def build():
    g = {"pattern": reference_pattern(), "cases": np.array(CASES, dtype=np.float64)}
    for k, (n, sf, nl) in enumerate(CASES):
        scale, inv, feats, umax = ctor_tables(n, sf, nl)
        g[f"c{k}_scale"] = scale; g[f"c{k}_inv"] = inv; g[f"c{k}_feats"] = feats
        g["umax"] = umax
        r, c = level_sizes(376, 1241, inv)
        g[f"c{k}_rows"] = r; g[f"c{k}_cols"] = c
    return g


def compare(g, oracle=True):
    """the product's and the oracle's tables against g; returns a list of mismatches (empty = all equal)"""
    bad = []
    for name in ("ssvio_amd/csrc/brief_pattern.inc", "oracle/src/brief_pattern.inc"):
        v = parse_ints(os.path.join(ROOT, name))
        if v.size != 1024 or not np.array_equal(v.reshape(256, 4), g["pattern"].astype(np.int64)):
            bad.append(name)
    if oracle:
        sys.path.insert(0, ROOT)
        from oracle import pyoracle as po
        if not np.array_equal(po.brief_pattern(), g["pattern"]):
            bad.append("oracle pattern (compiled)")
        if po.umax().tolist() != g["umax"].tolist():
            bad.append("oracle umax")
        for k, (n, sf, nl) in enumerate(g["cases"]):
            if po.features_per_level(int(n), float(sf), int(nl)).tolist() != g[f"c{k}_feats"].tolist():
                bad.append(f"oracle features_per_level{(int(n), sf, int(nl))}")
            r, c = po.level_sizes(376, 1241, float(sf), int(nl))
            if r.tolist() != g[f"c{k}_rows"].tolist() or c.tolist() != g[f"c{k}_cols"].tolist():
                bad.append(f"oracle level_sizes{(sf, int(nl))}")
    return bad


if __name__ == "__main__":
    assert os.path.isdir(REF), "needs /root/reference"
    g = build()
    bad = compare(g)
    assert not bad, bad
    np.savez_compressed(NPZ, **g)
    print("ref_tables.npz:", os.path.getsize(NPZ), "bytes; pattern, umax and", len(CASES), "constructor cases equal the oracle's and the product's tables")

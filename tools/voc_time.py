"""ssx_voc_transform on a vocabulary of ORBvoc's shape (k = 10; L = 5 here: 111 111 nodes -- ORBvoc.txt has L = 6, 1 082 073
nodes) for 2000 descriptors; run under rocprofv3 for the kernel time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ssvio_amd
from ssvio_amd import voc as svoc
ctx = ssvio_amd.Context(0)
rng = np.random.default_rng(0)
k, L = 10, int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = sum(k ** d for d in range(L + 1))
parent = np.zeros(n, np.int32); parent[0] = -1
first = np.cumsum([0] + [k ** d for d in range(L + 1)])
for d in range(1, L + 1):
    ids = np.arange(first[d], first[d + 1]); parent[ids] = first[d - 1] + (ids - first[d]) // k
leaf = np.zeros(n, np.uint8); leaf[first[L]:] = 1
desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
weight = np.where(leaf, rng.uniform(0.5, 9, n), 0.0)
t = time.perf_counter(); V = svoc.Vocabulary.from_arrays(ctx, k, L, parent, leaf, desc, weight); t_up = time.perf_counter() - t
feats = rng.integers(0, 256, (2000, 32), dtype=np.uint8)
for _ in range(3): V.transform(feats)
t = time.perf_counter(); N = 30
for _ in range(N): ids, vals = V.transform(feats)
dt = (time.perf_counter() - t) / N
print(f"k {k} L {L}: {n} nodes ({n * 32 / 1e6:.1f} MB of descriptors) uploaded in {t_up * 1e3:.1f} ms; transform of 2000 descriptors {dt * 1e3:.3f} ms/call, {len(ids)} words")

"""ORBVocabulary (DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>) as LoopClosing uses it -- ssx_voc_* of include/ssx.h:
transform(descriptors) -> BowVector, score(a, b) (reference: src/ssvio/loopclosing.cpp:84, 633)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import Context, dbl_p, i32_p, u8_p


class Vocabulary:
    def __init__(self, ctx: Context, handle):
        self.ctx, self.handle = ctx, handle
        k, L, nn, nw, wt = (C.c_int32() for _ in range(5))
        ctx.lib.ssx_voc_info(handle, C.byref(k), C.byref(L), C.byref(nn), C.byref(nw), C.byref(wt))
        self.k, self.L, self.n_nodes, self.n_words, self.weighting = k.value, L.value, nn.value, nw.value, wt.value

    @classmethod
    def from_arrays(cls, ctx: Context, k, L, parent, is_leaf, desc, weight, scoring=0, weighting=0):
        parent = np.ascontiguousarray(parent, dtype=np.int32); is_leaf = np.ascontiguousarray(is_leaf, dtype=np.uint8)
        desc = np.ascontiguousarray(desc, dtype=np.uint8).reshape(-1, 32); weight = np.ascontiguousarray(weight, dtype=np.float64)
        h = C.c_void_p()
        ctx.check(ctx.lib.ssx_voc_create(ctx.handle, int(k), int(L), int(scoring), int(weighting), len(parent), parent.ctypes.data_as(i32_p),
                                         is_leaf.ctypes.data_as(u8_p), desc.ctypes.data_as(u8_p), weight.ctypes.data_as(dbl_p), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def loadFromTextFile(cls, ctx: Context, path):
        h = C.c_void_p()
        ctx.check(ctx.lib.ssx_voc_load_text(ctx.handle, str(path).encode(), C.byref(h)))
        return cls(ctx, h)

    def close(self):
        if self.handle:
            self.ctx.lib.ssx_voc_destroy(self.handle)
            self.handle = None

    def transform(self, desc, with_features=False):
        """-> (ids [m] int32 ascending, values [m] float64, L1-normalised); with_features also the per-feature (word, weight)"""
        desc = np.ascontiguousarray(desc, dtype=np.uint8).reshape(-1, 32)
        n = len(desc)
        words = np.zeros(n, np.int32); weights = np.zeros(n, np.float64)
        cap = max(n, 1)
        ids = np.zeros(cap, np.int32); vals = np.zeros(cap, np.float64); m = C.c_int32(0)
        self.ctx.check(self.ctx.lib.ssx_voc_transform(self.handle, desc.ctypes.data_as(u8_p), n, words.ctypes.data_as(i32_p), weights.ctypes.data_as(dbl_p),
                                                      cap, ids.ctypes.data_as(i32_p), vals.ctypes.data_as(dbl_p), C.byref(m)))
        out = (ids[:m.value].copy(), vals[:m.value].copy())
        return out + (words, weights) if with_features else out

    def score(self, a, b):
        lib = self.ctx.lib
        lib.ssx_bow_score_l1.restype = C.c_double
        ia, va = np.ascontiguousarray(a[0], dtype=np.int32), np.ascontiguousarray(a[1], dtype=np.float64)
        ib, vb = np.ascontiguousarray(b[0], dtype=np.int32), np.ascontiguousarray(b[1], dtype=np.float64)
        return float(lib.ssx_bow_score_l1(len(ia), ia.ctypes.data_as(i32_p), va.ctypes.data_as(dbl_p), len(ib), ib.ctypes.data_as(i32_p), vb.ctypes.data_as(dbl_p)))

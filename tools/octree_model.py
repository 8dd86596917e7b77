"""Executable model of the DATA-PARALLEL formulation of DistributeOctTree used by the HIP kernel
(ssvio_amd/csrc/orb.hip: k_octree).  Development aid: it states, with flat arrays and prefix sums only, the
same selection the sequential std::list algorithm of the reference produces
(/root/reference/src/ssvio/orbextractor.cpp:340-568) with the deterministic tie-break (creation order).
tests/test_octree_model.py checks it against the CPU oracle; the kernel follows this file step by step.

State: keys[pos] (candidate ids, every node owns a contiguous range, insertion order preserved),
node table (begin,end,ulx,uly,brx,bry,no_more), order[] = node ids front-to-back of the std::list.
One ROUND divides a list `proc` of nodes (in processing order):
  phase 1 : proc = every node of the list that is not no_more, in list order
  phase 2 : proc = the expandable nodes of the previous round sorted by (size, creation id) DESCENDING,
            truncated at the first prefix that brings the node count to >= N
after which
  new order      = [children of proc[m-1] (q=3..0), ..., children of proc[0] (q=3..0)] + old order minus proc
  new expandable = children with > 1 key, in creation order (proc index ascending, q ascending)
"""
import math

import numpy as np


def _f32(x):
    return np.float32(x)


def octree_parallel(cand_x, cand_y, cand_resp, minX, maxX, minY, maxY, N):
    M = len(cand_x)
    if M == 0:
        return []
    nIni = int(math.floor(float(_f32(maxX - minX) / _f32(maxY - minY)) + 0.5))
    if nIni < 1:
        return []
    hX = _f32(maxX - minX) / _f32(nIni)
    # node table
    nb, ne, ulx, uly, brx, bry, nomore = [], [], [], [], [], [], []
    # initial assignment: stable counting sort of the candidates by root node
    root = np.minimum((cand_x.astype(np.float32) / hX).astype(np.int32), nIni - 1)
    keys = np.argsort(root, kind="stable").astype(np.int32)
    counts = np.bincount(root, minlength=nIni)
    start = 0
    order = []
    for i in range(nIni):
        c = int(counts[i])
        if c > 0:
            nb.append(start); ne.append(start + c)
            ulx.append(int(hX * _f32(i))); uly.append(0); brx.append(int(hX * _f32(i + 1))); bry.append(maxY - minY)
            nomore.append(c == 1)
            order.append(len(nb) - 1)
        start += c
    expandable = []

    def divide_round(proc):
        """divide the nodes proc[0..m) ; returns the new expandable list"""
        nonlocal keys, order
        m = len(proc)
        newkeys = keys.copy()
        child_ids = []          # per proc index: list of (q, node id) of non-empty children
        new_exp = []
        for i, n in enumerate(proc):
            b, e = nb[n], ne[n]
            halfX = int(math.ceil(float(_f32(brx[n] - ulx[n]) / _f32(2))))
            halfY = int(math.ceil(float(_f32(bry[n] - uly[n]) / _f32(2))))
            mx, my = ulx[n] + halfX, uly[n] + halfY
            ks = keys[b:e]
            q = np.where(cand_x[ks] < mx, np.where(cand_y[ks] < my, 0, 2), np.where(cand_y[ks] < my, 1, 3))
            cnt = np.bincount(q, minlength=4)
            # stable 4-way partition (rank inside a quadrant = exclusive prefix count)
            off = b
            kids = []
            bounds = [(ulx[n], uly[n], mx, my), (mx, uly[n], brx[n], my), (ulx[n], my, mx, bry[n]), (mx, my, brx[n], bry[n])]
            for qq in range(4):
                sel = ks[q == qq]
                newkeys[off:off + len(sel)] = sel
                if len(sel) > 0:
                    nb.append(off); ne.append(off + len(sel))
                    ulx.append(bounds[qq][0]); uly.append(bounds[qq][1]); brx.append(bounds[qq][2]); bry.append(bounds[qq][3])
                    nomore.append(len(sel) == 1)
                    nid = len(nb) - 1
                    kids.append((qq, nid))
                    if len(sel) > 1:
                        new_exp.append(nid)
                off += len(sel)
            child_ids.append(kids)
        keys = newkeys
        divided = set(proc)
        front = []
        for i in range(m - 1, -1, -1):
            for qq, nid in reversed(child_ids[i]):
                front.append(nid)
        order = front + [n for n in order if n not in divided]
        return new_exp

    finish = False
    while not finish:
        prev = len(order)
        proc = [n for n in order if not nomore[n]]
        expandable = divide_round(proc)
        nToExpand = len(expandable)
        if len(order) >= N or len(order) == prev:
            finish = True
        elif len(order) + 3 * nToExpand > N:
            while not finish:
                prev = len(order)
                # descending (size, creation id): largest first, newest first among equals
                cand = sorted(expandable, key=lambda n: (ne[n] - nb[n], n), reverse=True)
                # how many can be divided before the node count reaches N: needs each node's non-empty child count
                size = len(order)
                t = 0
                for n in cand:
                    b, e = nb[n], ne[n]
                    halfX = int(math.ceil(float(_f32(brx[n] - ulx[n]) / _f32(2))))
                    halfY = int(math.ceil(float(_f32(bry[n] - uly[n]) / _f32(2))))
                    mx, my = ulx[n] + halfX, uly[n] + halfY
                    ks = keys[b:e]
                    q = np.where(cand_x[ks] < mx, np.where(cand_y[ks] < my, 0, 2), np.where(cand_y[ks] < my, 1, 3))
                    cc = len(np.unique(q))
                    size += cc - 1
                    t += 1
                    if size >= N:
                        break
                expandable = divide_round(cand[:t])
                if len(order) >= N or len(order) == prev:
                    finish = True
    out = []
    for n in order:
        ks = keys[nb[n]:ne[n]]
        best = ks[0]
        for k in ks[1:]:
            if cand_resp[k] > cand_resp[best]:
                best = k
        out.append(int(best))
    return out

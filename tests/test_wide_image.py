"""The two-level ("wide") traversal image of the batched kernel (round 4; rt-octree_amd/csrc/rto_abi.cpp build_wide_image):
derived data -- for every point the leaf it answers with (level, slot, sigma) must be the one the plain root walk over
child[] finds (n3tree_query.hpp:22-47).  Host-only probe of the C ABI: runs without a GPU."""
import ctypes as C

import numpy as np
import pytest

from rt_octree_amd import synth
from rt_octree_amd._lib import check, lib


def plain_walk(child, pts, depth_bits=24):
    """(level, slot) of the leaf holding each 24-bit fixed-point point: query_single_from_root in integers"""
    flat = child.reshape(-1)
    lv = np.zeros(len(pts), np.int32)
    sl = np.zeros(len(pts), np.int64)
    for i, (x, y, z) in enumerate(pts):
        node, lvl = 0, 0
        while True:
            sh = depth_bits - 1 - lvl
            ci = ((int(x) >> sh) & 1) << 2 | ((int(y) >> sh) & 1) << 1 | ((int(z) >> sh) & 1)
            c = int(flat[node * 8 + ci])
            if c == 0:
                lv[i], sl[i] = lvl, node * 8 + ci
                break
            node += c
            lvl += 1
    return lv, sl


def probe(tree, pts, G):
    child = np.ascontiguousarray(tree.child.reshape(-1), np.int32)
    sigma = np.ascontiguousarray(tree.data.reshape(-1, tree.data.shape[-1])[:, -1]).view(np.uint16)
    pts = np.ascontiguousarray(pts, np.uint32)
    n = len(pts)
    lv, sl, sg = np.zeros(n, np.int32), np.zeros(n, np.int64), np.zeros(n, np.uint16)
    wn = C.c_int64(0)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    check(lib().rto_wide_image_probe(P(child), P(sigma), child.size // 8, int(tree.depth_limit), G, P(pts), n, P(lv), P(sl), P(sg),
                                     C.byref(wn)))
    return lv, sl, sg, wn.value, sigma


@pytest.mark.parametrize("depth,G", [(1, 0), (2, 0), (3, 2), (4, 3), (5, 4), (6, 5), (7, 6), (8, 6), (9, 6)])
def test_wide_image_answers_like_the_root_walk(depth, G):
    """every depth parity around the grid levels (G = the levels the top grid covers: min(6, depth - 1), 0 below depth 3),
    leaves at both levels of a pair, above the grid, at the deepest level; random points + cell corners"""
    tree = synth.make_tree(depth_limit=depth, basis_dim=4, seed=100 + depth, shell=1.5)
    rs = np.random.RandomState(depth)
    pts = rs.randint(0, 1 << 24, (4000, 3)).astype(np.uint32)
    edge = (rs.randint(0, 1 << min(depth + 1, 10), (1000, 3)).astype(np.uint64) << (24 - min(depth + 1, 10))).astype(np.uint32)
    pts = np.concatenate([pts, edge, np.maximum(edge, 1) - 1, [[0, 0, 0], [(1 << 24) - 1] * 3]]).astype(np.uint32)
    lv, sl, sg, n_wide, sigma = probe(tree, pts, G)
    lv0, sl0 = plain_walk(tree.child, pts)
    assert np.array_equal(lv, lv0), "leaf levels differ"
    assert np.array_equal(sl, sl0), "leaf slots differ"
    assert np.array_equal(sg, sigma[sl0]), "sigma differs"
    levels = np.bincount(lv0, minlength=depth + 1)
    assert n_wide >= 1 and levels.max() > 0


def test_wide_image_counts_the_nodes_of_every_second_level():
    tree = synth.make_tree(depth_limit=8, basis_dim=4, seed=5, shell=1.5)
    _, _, _, n_wide, _ = probe(tree, np.zeros((1, 3), np.uint32), 6)
    per_level = tree.stats["levels"]
    assert n_wide == per_level[6]  # levels 6 and 7 form the only pair: one wide node per level-6 node
    _, _, _, n_wide0, _ = probe(tree, np.zeros((1, 3), np.uint32), 0)
    assert n_wide0 == per_level[0] + per_level[2] + per_level[4] + per_level[6]

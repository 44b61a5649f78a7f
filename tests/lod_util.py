"""Synthetic level-of-detail trees in the buffer layout TensorTree keeps (LoG/model/tensor_tree.py:57-90): built
level by level the way `split` appends nodes and children, plus holes (tree entries set to -1) the way `remove`
leaves them.  Used where the golden trees (grown by the reference's own class) are too small."""
import math

import numpy as np


def synth_tree(n_roots, n_levels, max_child, split_prob=0.75, hole_prob=0.03, seed=0, extent=1.0, root_scale=0.05):
    rng = np.random.default_rng(seed)
    xyz = ((rng.random((n_roots, 3)) - 0.5) * extent).astype(np.float32)
    scaling = np.log(rng.random((n_roots, 3)) * root_scale + 0.2 * root_scale).astype(np.float32)
    rotation = rng.standard_normal((n_roots, 4)).astype(np.float32)
    node_index = np.full(n_roots, -1, np.int32)
    depth = np.zeros(n_roots, np.int8)
    rows = []
    num_nodes = 0
    for level in range(n_levels):
        cand = np.nonzero((node_index == -1) & (depth == level))[0]
        parent = cand[rng.random(cand.shape[0]) < split_prob]
        ns = parent.shape[0]
        if ns == 0:
            break
        P = node_index.shape[0]
        node_index[parent] = num_nodes + np.arange(ns, dtype=np.int32)
        child = (P + np.arange(ns * max_child, dtype=np.int32)).reshape(ns, max_child)
        rows.append(child)
        num_nodes += ns
        rep = np.repeat(parent, max_child)
        sig = np.exp(scaling[rep]).max(axis=1, keepdims=True)
        xyz = np.concatenate([xyz, xyz[rep] + (rng.standard_normal((rep.shape[0], 3)) * sig).astype(np.float32)])
        scaling = np.concatenate([scaling, (scaling[rep] - math.log(1.6)
                                            + 0.2 * rng.standard_normal((rep.shape[0], 3))).astype(np.float32)])
        rotation = np.concatenate([rotation, rotation[rep] + 0.3 * rng.standard_normal((rep.shape[0], 4)).astype(np.float32)])
        node_index = np.concatenate([node_index, np.full(rep.shape[0], -1, np.int32)])
        depth = np.concatenate([depth, np.full(rep.shape[0], level + 1, np.int8)])
    tree = np.concatenate(rows) if rows else np.zeros((0, max_child), np.int32)
    if hole_prob > 0 and tree.size:
        holes = (rng.random(tree.shape) < hole_prob) & (node_index[tree] == -1)     # only leaves are ever removed
        tree = np.where(holes, -1, tree).astype(np.int32)
    return dict(node_index=node_index, tree=np.ascontiguousarray(tree), depth=depth,
                xyz=np.ascontiguousarray(xyz, np.float32), scaling=np.ascontiguousarray(scaling, np.float32),
                rotation=np.ascontiguousarray(rotation, np.float32), root_index=np.arange(n_roots, dtype=np.int64))

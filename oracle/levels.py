"""Definitional topological-level scheduler (numpy, integer, bit-exact target).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference has no level scheduler: it walks nodes in index order and relies
on the adjacency being strictly lower-triangular
(/root/reference/dvae/models/dvae_model.py:63,93,196).  The
level of a node is therefore pinned by definition:

    level(i) = 0                          if pred(i) is empty
             = 1 + max_{u in pred(i)} level(u)   otherwise

with pred(i) = {u : g[i,u] != 0} for the forward sweep and
pred_rev(i) = {u : g[u,i] != 0} (transposed graph, dvae_model.py:90) for the
reverse sweep.  The decoder's final states follow the same recurrence (node 0
is merely seeded with h0), so they use the forward levels.
"""
import numpy as np


def forward_levels(g):
    """g [B,N,N] (row = node, col = predecessor), strictly lower-triangular.
    Returns int32 [B,N]."""
    g = np.asarray(g) != 0
    B, N, _ = g.shape
    lev = np.zeros((B, N), dtype=np.int32)
    for i in range(N):
        for u in range(i):
            has = g[:, i, u]
            lev[:, i] = np.where(has, np.maximum(lev[:, i], lev[:, u] + 1), lev[:, i])
    return lev


def reverse_levels(g, sizes=None):
    """Levels on the transposed graph (successors act as predecessors)."""
    g = np.asarray(g) != 0
    B, N, _ = g.shape
    lev = np.zeros((B, N), dtype=np.int32)
    for i in range(N - 1, -1, -1):
        for u in range(N - 1, i, -1):
            has = g[:, u, i]
            lev[:, i] = np.where(has, np.maximum(lev[:, i], lev[:, u] + 1), lev[:, i])
    return lev


def decoder_levels(g):
    """Decoder final states obey the encoder recurrence except that node 0 is
    seeded with h0 (dvae_model.py:203-205); levels are the forward levels."""
    return forward_levels(g)


def pred_lists_desc(g):
    """Per graph, per node: predecessor indices in DESCENDING order (the order
    in which the teacher-forced decoder's inner loop meets them,
    dvae_model.py:210,223-227)."""
    g = np.asarray(g) != 0
    B, N, _ = g.shape
    return [[[u for u in range(i - 1, -1, -1) if g[b, i, u]] for i in range(N)]
            for b in range(B)]

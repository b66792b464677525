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


def shared_encoder_rows(g, types, sizes=None):
    """Definitional value numbering of the encoder's forward sweep (the library's "shared encoder rows").

    The state of node i (dvae_model.py:63-87) is computed from its type (:83, the GRU input), its index
    (:71, the position one-hot of the gate/map input) and the states of its predecessors (:77-80) -- so two
    nodes with equal (index, type, predecessor values in descending index order) hold equal states.  The
    last node of a graph is kept apart (its row takes that graph's gradient).

    g [B,N,N], types [B,N] ints, sizes [B] or None.  Returns
      rep      [B*N] node id (b*N+i) of the lowest node id with the same value (-1 for absent nodes)
      node2row [B*N] value row of every node; rows ordered by (level, graph, index) of the representative
      csr      list over rows of predecessor rows (descending node index)
      csc      list over rows of successor rows, ascending
    """
    g = np.asarray(g) != 0
    types = np.asarray(types)
    B, N, _ = g.shape
    lev = forward_levels(g)
    sizes = np.full(B, N, dtype=np.int64) if sizes is None else np.asarray(sizes)
    value_of = {}
    val = -np.ones((B, N), dtype=np.int64)
    rep = -np.ones(B * N, dtype=np.int64)
    first = {}
    for b in range(B):
        for i in range(int(sizes[b])):
            key = (i, int(types[b, i]), tuple(int(val[b, u]) for u in range(i - 1, -1, -1) if g[b, i, u]))
            if i == int(sizes[b]) - 1:
                key = key + (('last', b),)
            v = value_of.setdefault(key, len(value_of))
            val[b, i] = v
            first.setdefault(v, b * N + i)
            rep[b * N + i] = first[v]
    reps = sorted((int(lev[r // N, r % N]), r // N, r % N) for r in set(first.values()))
    row_of_rep = {b * N + i: k for k, (_, b, i) in enumerate(reps)}
    node2row = np.array([row_of_rep[r] if r >= 0 else -1 for r in rep], dtype=np.int64)
    csr = [[int(node2row[b * N + u]) for u in range(i - 1, -1, -1) if g[b, i, u]] for (_, b, i) in reps]
    csc = [[] for _ in reps]
    for w, preds in enumerate(csr):
        for u in preds:
            csc[u].append(w)
    return rep, node2row, csr, [sorted(c) for c in csc]

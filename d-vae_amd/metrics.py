"""Exact-match reconstruction accuracy (reference: dvae/utils/metrics.py:4-28) and,
as labelled additions, validity / uniqueness / novelty of decoded DAGs."""
import torch


def reconstruction_accuracy(true_encodings, true_deps, pred_encodings, pred_deps):
    """true_encodings [B,N,C], true_deps [B,N,N]; pred_* carry an extra sample
    axis [B,S,N,.].  A decoded graph counts when every node type and every
    adjacency entry equals the input's.  Returns a python float."""
    batch_size, sample_num = pred_encodings.shape[0], pred_encodings.shape[1]
    nodes_ok = (pred_encodings == true_encodings.unsqueeze(1)).flatten(2).all(dim=2)
    edges_ok = (pred_deps == true_deps.unsqueeze(1)).flatten(2).all(dim=2)
    return (nodes_ok & edges_ok).sum().item() / (batch_size * sample_num)


# ---------------------------------------------------------------------------
# Additions (NOT in the reference, which only has the exact-match accuracy above):
# validity / uniqueness / novelty of decoded DAGs, BASELINE config 5's "validity/novelty
# eval path".  Definitions follow the D-VAE paper's evaluation (Zhang et al., NeurIPS 2019,
# sec. 4.1 and its is_valid_ENAS / is_valid_BN checks), restated on this repo's encoding
# (dataloader.py:57-88: node 0 = start token of type C-2, node N-1 = end token of type
# C-1, adjacency [i,j] = 1 <=> edge j->i, strictly lower-triangular).  Pure tensor ops:
# they run on whatever device the decoded tensors live on.
# ---------------------------------------------------------------------------

def dag_validity(edges, onehots, task_type='enas'):
    """edges [M,N,N] 0/1, onehots [M,N,C] -> bool [M].  A decoded graph is valid when
    * node 0 is the only start-type node and node N-1 the only end-type node, every node has
      exactly one type;
    * the adjacency is 0/1 and strictly lower-triangular (a DAG in index order);
    * every node but the start has a predecessor, every node but the end a successor;
    * 'enas': every node i in 1..N-2 is fed by node i-1 (the loader always sets that edge,
      dataloader.py:82-84); 'bn': the N-2 variable nodes carry N-2 distinct types."""
    if task_type not in ('enas', 'bn'):
        raise NotImplementedError
    M, N, C = onehots.shape
    assert edges.shape == (M, N, N)
    start, end = C - 2, C - 1
    one_type = ((onehots == 0) | (onehots == 1)).flatten(1).all(1) & (onehots.sum(2) == 1).all(1)
    types = onehots.argmax(2)
    idx = torch.arange(N, device=edges.device)
    tok_ok = (types[:, 0] == start) & (types[:, N - 1] == end)
    tok_ok &= ((types == start).sum(1) == 1) & ((types == end).sum(1) == 1)
    lower = idx.view(N, 1) > idx.view(1, N)
    binary = ((edges == 0) | (edges == 1)).flatten(1).all(1) & (edges * (~lower)).flatten(1).eq(0).all(1)
    has_pred = (edges.sum(2) > 0)[:, 1:].all(1)
    has_succ = (edges.sum(1) > 0)[:, :N - 1].all(1)
    ok = one_type & tok_ok & binary & has_pred & has_succ
    if task_type == 'enas':
        if N > 2:
            ok &= (edges[:, idx[1:N - 1], idx[0:N - 2]] == 1).all(1)
    else:
        inner = torch.nn.functional.one_hot(types[:, 1:N - 1], C).sum(1)       # [M,C] type counts
        ok &= (inner <= 1).all(1)
    return ok


def graph_keys(edges, onehots):
    """Two independent 64-bit keys per graph ([M,2] int64) over its lower-triangular adjacency
    bits and node types: equal graphs get equal keys, different graphs collide with probability
    ~2^-126 per pair (wrap-around integer arithmetic, identical on CPU and GPU)."""
    M, N, _ = onehots.shape
    dev = edges.device
    idx = torch.arange(N, device=dev)
    lower = (idx.view(N, 1) > idx.view(1, N)).flatten()
    feats = torch.cat([(edges.flatten(1)[:, lower] != 0).long(), onehots.argmax(2).long() + 2], dim=1)
    gen = torch.Generator().manual_seed(0x5eed)
    w = torch.randint(-(1 << 62), 1 << 62, (2, feats.shape[1]), generator=gen, dtype=torch.int64) | 1
    return (feats.unsqueeze(1) * w.to(dev).unsqueeze(0)).sum(2)


def _select(keys, mask):
    return keys if mask is None else keys[mask]


def uniqueness(edges, onehots, mask=None):
    """Fraction of the (masked, e.g. valid) decoded graphs that are distinct from each other."""
    keys = _select(graph_keys(edges, onehots), mask)
    if keys.shape[0] == 0:
        return 0.0
    return torch.unique(keys, dim=0).shape[0] / keys.shape[0]


def novelty(edges, onehots, train_edges, train_onehots, mask=None):
    """Fraction of the (masked, e.g. valid) decoded graphs that do not occur in the training set."""
    keys = _select(graph_keys(edges, onehots), mask)
    if keys.shape[0] == 0:
        return 0.0
    tk = graph_keys(train_edges.to(edges.device), train_onehots.to(edges.device))
    seen = torch.isin(keys[:, 0], tk[:, 0]) & torch.isin(keys[:, 1], tk[:, 1])
    return 1.0 - seen.float().mean().item()


def generation_report(edges, onehots, train_edges=None, train_onehots=None, task_type='enas'):
    """{'validity', 'uniqueness', 'novelty'} of a decoded set (uniqueness / novelty among the valid)."""
    ok = dag_validity(edges, onehots, task_type)
    out = {'validity': ok.float().mean().item() if ok.numel() else 0.0,
           'uniqueness': uniqueness(edges, onehots, ok)}
    if train_edges is not None:
        out['novelty'] = novelty(edges, onehots, train_edges, train_onehots, ok)
    return out

"""Exact-match reconstruction accuracy (reference: dvae/utils/metrics.py:4-28)."""


def reconstruction_accuracy(true_encodings, true_deps, pred_encodings, pred_deps):
    """true_encodings [B,N,C], true_deps [B,N,N]; pred_* carry an extra sample
    axis [B,S,N,.].  A decoded graph counts when every node type and every
    adjacency entry equals the input's.  Returns a python float."""
    batch_size, sample_num = pred_encodings.shape[0], pred_encodings.shape[1]
    nodes_ok = (pred_encodings == true_encodings.unsqueeze(1)).flatten(2).all(dim=2)
    edges_ok = (pred_deps == true_deps.unsqueeze(1)).flatten(2).all(dim=2)
    return (nodes_ok & edges_ok).sum().item() / (batch_size * sample_num)

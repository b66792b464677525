"""Dense-loop CPU restatement of the reference D-VAE hot path (torch, fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function works on a
plain ``dict`` of weights keyed by the reference's ``state_dict()`` names
 and keeps the reference's op order and O(N^3) work, so that
timing it is a fair stand-in for the reference's CPU path
(BASELINE.md section 3).  It is written functionally (no in-place row writes) so
that torch autograd gives the gradients the parity tests compare against.

Reference being restated: /root/reference/dvae/models/dvae_model.py
  Encoder.forward       :53-123
  Decoder.forward       :178-261
  Decoder.predict       :263-362
  Dvae.reparamterize    :410-413
  Dvae.training_step    :421-440
  Dvae.validation_step  :442-459
"""
import math

import torch

LN2 = math.log(2.0)


# --------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------
def _lin(sd, name, v):
    return v @ sd[name + ".weight"].t() + sd[name + ".bias"]


def gru_cell(sd, prefix, x, h):
    """torch.nn.GRUCell arithmetic (gate order r, z, n), used at
    dvae_model.py:83,112,205,207,244,296,298,339."""
    gi = x @ sd[prefix + ".weight_ih"].t() + sd[prefix + ".bias_ih"]
    gh = h @ sd[prefix + ".weight_hh"].t() + sd[prefix + ".bias_hh"]
    H = h.shape[-1]
    r = torch.sigmoid(gi[..., :H] + gh[..., :H])
    u = torch.sigmoid(gi[..., H:2 * H] + gh[..., H:2 * H])
    n = torch.tanh(gi[..., 2 * H:] + r * gh[..., 2 * H:])
    return (1.0 - u) * n + u * h


def _gate_map(sd, gate, mapper, feat):
    """gating_network(feat) * mapping_network(feat)  (dvae_model.py:77-78)."""
    return torch.sigmoid(_lin(sd, gate + ".0", feat)) * _lin(sd, mapper, feat)


def _states_tensor(states, B, N, H, like):
    """[B,N,H] table with zeros for rows not yet produced
    (node_hidden_state of dvae_model.py:57-58,186-187)."""
    cols = []
    for i in range(N):
        s = states[i]
        cols.append(s if s is not None else like.new_zeros(B, H))
    return torch.stack(cols, dim=1)


def _dense_h_in(sd, gate, mapper, mod, table, x, mask, n_pos):
    """Masked dense aggregation over ALL N rows (dvae_model.py:67-80, 223-241).

    table [B,N,H] hidden-state table, x [B,N,C], mask [B,N] (1 = predecessor).
    n_pos = number of position columns the gate/map Linear has after the H
    hidden columns (= max_seq of the module that owns the weights).
    """
    B, N, _ = table.shape
    m = mask.unsqueeze(-1)
    if mod is None:
        eye = torch.eye(n_pos, dtype=table.dtype)[:N].unsqueeze(0).expand(B, N, n_pos)
        feat = torch.cat([table, eye], dim=-1) * m
    elif mod == "bn":
        feat = x * m
    else:
        raise NotImplementedError(mod)
    return (_gate_map(sd, gate, mapper, feat) * m).sum(dim=1)


def _n_pos(sd, gate_key, H, mod):
    w = sd[gate_key + ".0.weight"]
    return (w.shape[1] - H) if mod is None else 0


# --------------------------------------------------------------------------
# encoder (dvae_model.py:53-123)
# --------------------------------------------------------------------------
def encoder_forward(sd, g, x, mod=None, bidirectional=False, return_states=False):
    B, N, _ = g.shape
    H = sd["encoder.gru.weight_hh"].shape[1]
    n_pos = _n_pos(sd, "encoder.gating_network", H, mod)
    states = [None] * N
    hv = None
    for i in range(N):                                   # :63
        table = _states_tensor(states, B, N, H, g)
        h_in = _dense_h_in(sd, "encoder.gating_network", "encoder.mapping_network",
                           mod, table, x, g[:, i], n_pos)
        hv = gru_cell(sd, "encoder.gru", x[:, i], h_in)  # :83
        states[i] = hv
    back = [None] * N
    if bidirectional:
        gt = g.transpose(1, 2)                           # :90
        for i in range(N - 1, -1, -1):                   # :93
            table = _states_tensor(back, B, N, H, g)
            h_in = _dense_h_in(sd, "encoder.back_gating_network",
                               "encoder.back_mapping_network", mod, table, x, gt[:, i], n_pos)
            hb = gru_cell(sd, "encoder.back_gru", x[:, i], h_in)
            back[i] = hb
            hv = _lin(sd, "encoder.lin01", torch.cat([hv, hb], dim=-1))   # :119 (inside loop)
    mu = _lin(sd, "encoder.lin11", hv)                   # :121
    logvar = _lin(sd, "encoder.lin12", hv)               # :122
    if return_states:
        fw = torch.stack(states, dim=1)
        bw = torch.stack(back, dim=1) if bidirectional else None
        return hv, mu, logvar, fw, bw
    return hv, mu, logvar


# --------------------------------------------------------------------------
# reparameterise (dvae_model.py:410-413), eps injected
# --------------------------------------------------------------------------
def reparameterize(mu, logvar, eps):
    return eps * torch.exp(0.5 * logvar) + mu


# --------------------------------------------------------------------------
# teacher-forced decoder (dvae_model.py:178-261)
# --------------------------------------------------------------------------
def _mlp2(sd, name, v):
    return _lin(sd, name + ".2", torch.relu(_lin(sd, name + ".0", v)))


def decoder_forward(sd, z, g, x, mod=None, return_states=False):
    B, N, _ = g.shape
    C = x.shape[-1]
    H = sd["decoder.gru.weight_hh"].shape[1]
    n_pos = _n_pos(sd, "decoder.gating_network", H, mod)
    h0 = _lin(sd, "decoder.lin1", z)                     # :194
    graph_state = h0
    states = [None] * N
    type_rows = []
    edge_rows = []
    for i in range(N):                                   # :196
        type_rows.append(_mlp2(sd, "decoder.add_vertex", graph_state))   # :199
        if i == 0:
            hv = gru_cell(sd, "decoder.gru", x[:, 0], h0)                 # :205
        else:
            hv = gru_cell(sd, "decoder.gru", x[:, i], torch.zeros_like(h0))  # :207
        row = [None] * N
        for j in range(i - 1, -1, -1):                   # :210
            hj = states[j]
            if mod == "bn":
                inp = torch.cat([hv, hj, h0], dim=1)     # :213
            else:
                inp = torch.cat([hv, hj], dim=1)         # :215
            row[j] = _mlp2(sd, "decoder.add_edge", inp).view(-1)          # :218
            mask = g[:, i].clone()
            mask[:, :j] = 0                              # :226
            # table holds finished rows < i, and row i's running state (:248)
            cur = list(states)
            cur[i] = hv if j < i - 1 else None
            table = _states_tensor(cur, B, N, H, g)
            h_in = _dense_h_in(sd, "decoder.gating_network", "decoder.mapping_network",
                               mod, table, x, mask, n_pos)
            hv = gru_cell(sd, "decoder.gru", x[:, i], h_in)               # :244
        states[i] = hv                                   # :249
        edge_rows.append(torch.stack(
            [r if r is not None else g.new_zeros(B) for r in row], dim=1))
        if mod == "bn":
            graph_state = hv if i == 0 else graph_state + hv              # :251-255
        else:
            graph_state = hv                             # :257
    edge_logits = torch.stack(edge_rows, dim=1)          # [B,N,N]
    type_logits = torch.stack(type_rows, dim=1)          # [B,N,C]
    assert type_logits.shape == (B, N, C)
    if return_states:
        return edge_logits, type_logits, torch.stack(states, dim=1), h0
    return edge_logits, type_logits


# --------------------------------------------------------------------------
# ELBO (dvae_model.py:421-459)
# --------------------------------------------------------------------------
def elbo(edge_logits, type_logits, g, x, mu, logvar, beta=0.005):
    """Returns (loss, edge_loss, vertex_loss, kl_loss).

    training_step hard-codes beta=0.005 (:432); validation_step uses self.beta
    (:454).  BCE is summed over the FULL NxN matrix (:425), so the never-written
    entries (logit 0, target 0) contribute ln2 each.
    """
    edge_loss = torch.nn.functional.binary_cross_entropy_with_logits(
        edge_logits, g, reduction="sum")
    vertex_loss = (-torch.log_softmax(type_logits, dim=2) * x).sum()
    kl = -0.5 * torch.sum(1 + logvar - mu ** 2 - logvar.exp())
    return edge_loss + vertex_loss + beta * kl, edge_loss, vertex_loss, kl


def train_loss(sd, g, x, eps, mod=None, bidirectional=False, beta=0.005):
    """Dvae.forward + training_step loss with eps injected (:415-432)."""
    _, mu, logvar = encoder_forward(sd, g, x, mod, bidirectional)
    z = reparameterize(mu, logvar, eps)
    e, t = decoder_forward(sd, z, g, x, mod)
    return elbo(e, t, g, x, mu, logvar, beta) + (mu, logvar, e, t)


# --------------------------------------------------------------------------
# autoregressive decode (dvae_model.py:263-362)
# --------------------------------------------------------------------------
def decoder_predict(sd, z, N, C, mod=None, uniforms=None, return_margin=False):
    """Greedy (uniforms=None) or stochastic decode.

    Stochastic mode is driven by INJECTED uniforms (device generators cannot be
    matched): uniforms = {'type': [M,N], 'edge': [M,N,N]} in [0,1).
      type_i  = first class c with cumsum(softmax)[c] > u   (stand-in for
                torch.multinomial at :285, same distribution)
      edge_ij = 1 if u < sigmoid(logit)                      (torch.bernoulli, :314)
    Greedy: argmax (:287) and sigmoid >= 0.5 (:316-317).
    Returns (edges [M,N,N], onehot types [M,N,C]) (+ min decision margin).
    """
    M = z.shape[0]
    H = sd["decoder.gru.weight_hh"].shape[1]
    n_pos = _n_pos(sd, "decoder.gating_network", H, mod)
    h0 = _lin(sd, "decoder.lin1", z)
    graph_state = h0
    states = []
    onehots = []
    edges = z.new_zeros(M, N, N)
    margin = float("inf")
    for i in range(N):
        logits = _mlp2(sd, "decoder.add_vertex", graph_state)
        p = torch.softmax(logits, dim=1)                  # :282
        if uniforms is None:
            t = p.argmax(dim=1)                           # :287
            top2 = logits.topk(2, dim=1).values
            margin = min(margin, float((top2[:, 0] - top2[:, 1]).min()))
        else:
            cdf = torch.cumsum(p, dim=1)
            t = (cdf <= uniforms["type"][:, i:i + 1]).sum(dim=1).clamp(max=C - 1)
        oh = torch.nn.functional.one_hot(t, C).to(z.dtype)
        onehots.append(oh)
        if i == 0:
            hv = gru_cell(sd, "decoder.gru", oh, h0)      # :296
        else:
            hv = gru_cell(sd, "decoder.gru", oh, torch.zeros_like(h0))    # :298
        possible = z.new_zeros(M, i)                      # :300
        for j in range(i - 1, -1, -1):
            hj = states[j]
            inp = torch.cat([hv, hj, h0], 1) if mod == "bn" else torch.cat([hv, hj], 1)
            logit = _mlp2(sd, "decoder.add_edge", inp).view(-1)
            s = torch.sigmoid(logit)                      # :312
            if uniforms is None:
                e = (s >= 0.5).to(z.dtype)                # :316-317
                margin = min(margin, float(logit.abs().min()))
            else:
                e = (uniforms["edge"][:, i, j] < s).to(z.dtype)
            possible = possible.clone()
            possible[:, j] = e                            # :318
            if mod is None:
                table = torch.stack(states, dim=1)        # [M,i,H]  :324-327
                eye = torch.eye(n_pos, dtype=z.dtype)[:i].unsqueeze(0).expand(M, i, n_pos)
                feat = torch.cat([table, eye], dim=-1)
            else:
                feat = torch.stack(onehots[:-1], dim=1)   # :329-330
            gm = _gate_map(sd, "decoder.gating_network", "decoder.mapping_network", feat)
            h_in = (gm * possible.unsqueeze(-1)).sum(dim=1)               # :332-336
            hv = gru_cell(sd, "decoder.gru", oh, h_in)    # :339
        if i > 0:
            edges[:, i, :i] = possible                    # :359-360
        states.append(hv)
        if mod == "bn":
            graph_state = hv if i == 0 else graph_state + hv
        else:
            graph_state = hv
    out = (edges, torch.stack(onehots, dim=1))
    return out + (margin,) if return_margin else out


# --------------------------------------------------------------------------
# model-level decode paths (dvae_model.py:492-524) and the metric (dvae/utils/metrics.py:4-28)
# --------------------------------------------------------------------------
def reconstruction_accuracy(true_enc, true_deps, pred_enc, pred_deps):
    """metrics.py:4-28: fraction of decoded graphs whose node types (:18-21) AND adjacency (:24-25)
    equal the input's; pred_* carry a sample axis [B,S,N,.]."""
    B, S = pred_enc.shape[0], pred_enc.shape[1]
    node_eq = (true_enc.unsqueeze(1) == pred_enc).all(dim=-1).all(dim=-1)
    edge_eq = (true_deps.unsqueeze(1) == pred_deps).all(dim=-1).all(dim=-1) & node_eq
    return int(edge_eq.sum()) / (B * S)


def model_predict(sd, g, x, eps, mod=None, bidirectional=False):
    """Dvae.predict (:492-496) with eps injected: encode -> reparameterise -> greedy decode."""
    _, mu, logvar = encoder_forward(sd, g, x, mod, bidirectional)
    z = reparameterize(mu, logvar, eps)
    return decoder_predict(sd, z, g.shape[1], x.shape[2], mod)


def model_eval(sd, g, x, eps, sample_num, decode_num, mod=None, bidirectional=False, uniforms=None):
    """Dvae.model_eval (:498-524): eps [B,sample_num,Z]; z rows are laid out as
    stack_z.repeat(1, decode_num, 1).view(B*S*D, Z) (:506-508), i.e. row = (b, d, s).
    Greedy when uniforms is None, else stochastic with injected uniforms (decoder_predict).
    Returns (recon_acc, gen_edges [B,D*S,N,N], gen_nodes [B,D*S,N,C], z rows)."""
    B, N, C = g.shape[0], g.shape[1], x.shape[2]
    _, mu, logvar = encoder_forward(sd, g, x, mod, bidirectional)
    stack_mu = torch.stack([mu] * sample_num, dim=1)                  # :501
    stack_logvar = torch.stack([logvar] * sample_num, dim=1)          # :502
    stack_z = reparameterize(stack_mu, stack_logvar, eps)             # :503
    z = stack_z.repeat(1, decode_num, 1).reshape(B * sample_num * decode_num, -1)   # :506-508
    e, t = decoder_predict(sd, z, N, C, mod, uniforms=uniforms)
    e = e.view(B, decode_num * sample_num, N, N)                       # :512-515
    t = t.view(B, decode_num * sample_num, N, C)
    return reconstruction_accuracy(x, g, t, e), e, t, z


# --------------------------------------------------------------------------
# ragged batches: outside the reference (it forces N == max_seq, :60-62).
# Oracle per SURVEY.md section 5: group graphs by size n and run the dense path on
# [n,n] blocks; position one-hots of nodes < n only touch the first n position
# columns of the max_seq-wide gate/map weights, which `_dense_h_in` honours
# through n_pos.  The graph embedding is the state of node n-1.
# --------------------------------------------------------------------------
def ragged_groups(sizes):
    groups = {}
    for b, n in enumerate(sizes.tolist()):
        groups.setdefault(int(n), []).append(b)
    return groups


def train_loss_ragged(sd, g, x, sizes, eps, mod=None, bidirectional=False, beta=0.005):
    """Sum of per-size-group losses; g [B,Nmax,Nmax], x [B,Nmax,C] zero-padded."""
    total = None
    parts = [0.0, 0.0, 0.0]
    for n, idx in sorted(ragged_groups(sizes).items()):
        idx = torch.tensor(idx)
        out = train_loss(sd, g[idx][:, :n, :n], x[idx][:, :n], eps[idx], mod, bidirectional, beta)
        total = out[0] if total is None else total + out[0]
        for k in range(3):
            parts[k] = parts[k] + out[1 + k]
    return (total,) + tuple(parts)

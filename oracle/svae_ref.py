"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch fp32/fp64, explicit loops) of the reference's
S-VAE baseline, dvae/models/svae_model.py.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this; the product (d-vae_amd/svae.py) never does.

Parity is PINNED: tests/golden/make_svae_golden.py runs the real reference (torch.nn.GRU and all) in
the build container, asserts that this file reproduces its outputs, losses and gradients, and commits
the vectors as tests/golden/svae_*.npz.

Functions take a state dict with the reference's key names.
"""
import torch
import torch.nn.functional as F


def gru_seq(x_steps, w_ih, w_hh, b_ih, b_hh, reverse=False):
    """torch.nn.GRU, one layer, h0 = 0, over a list of [B, I] step inputs (svae_model.py:24-26,36,58,76):
    r = s(W_ir x + b_ir + W_hr h + b_hr), z likewise, n = tanh(W_in x + b_in + r * (W_hn h + b_hn)),
    h' = (1 - z) * n + z * h.  Returns the list of states in time order and the final state."""
    B = x_steps[0].shape[0]
    H = w_hh.shape[1]
    h = x_steps[0].new_zeros(B, H)
    out = [None] * len(x_steps)
    order = range(len(x_steps) - 1, -1, -1) if reverse else range(len(x_steps))
    for t in order:
        gi = x_steps[t] @ w_ih.t() + b_ih
        gh = h @ w_hh.t() + b_hh
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
        out[t] = h
    return out, h


def encoder(sd, x, bidir):
    """Encoder.forward (svae_model.py:35-44): x [B, T, I] -> (mu, logvar)."""
    p = 'encoder.gru_layer.'
    steps = [x[:, t] for t in range(x.shape[1])]
    _, h = gru_seq(steps, sd[p + 'weight_ih_l0'], sd[p + 'weight_hh_l0'], sd[p + 'bias_ih_l0'], sd[p + 'bias_hh_l0'])
    if bidir:
        _, hb = gru_seq(steps, sd[p + 'weight_ih_l0_reverse'], sd[p + 'weight_hh_l0_reverse'],
                        sd[p + 'bias_ih_l0_reverse'], sd[p + 'bias_hh_l0_reverse'], reverse=True)
        # h_n [2, B, H] -> transpose -> view(B, 2H) = [h_fwd | h_bwd] (:38-41)
        h = torch.cat((h, hb), dim=1) @ sd['encoder.hg_unify.weight'].t() + sd['encoder.hg_unify.bias']
    mu = h @ sd['encoder.fc_mu.weight'].t() + sd['encoder.fc_mu.bias']
    logvar = h @ sd['encoder.fc_var.weight'].t() + sd['encoder.fc_var.bias']
    return mu, logvar


def decoder(sd, z, T):
    """Decoder.forward (svae_model.py:73-79): z [B, Z] -> (type_scores [B,T,C], edge_scores [B,T,N])."""
    p = 'decoder.gru_layer.'
    h0 = torch.relu(z @ sd['decoder.fc_h0.weight'].t() + sd['decoder.fc_h0.bias'])
    hs, _ = gru_seq([h0] * T, sd[p + 'weight_ih_l0'], sd[p + 'weight_hh_l0'], sd[p + 'bias_ih_l0'], sd[p + 'bias_hh_l0'])
    h_out = torch.stack(hs, dim=1)

    def mlp(name):
        hid = torch.relu(h_out @ sd[name + '.0.weight'].t() + sd[name + '.0.bias'])
        return hid @ sd[name + '.2.weight'].t() + sd[name + '.2.bias']
    return mlp('decoder.add_vertex'), mlp('decoder.add_edges')


def relu_margin(sd, z, T):
    """Smallest |pre-activation| over the decoder's three ReLUs (:74,61-70).  An fp32 implementation may
    land on the other side of a ReLU whose input is within rounding noise of 0, which changes that unit's
    gradient by a whole term; tests pick inputs whose margin is above that noise."""
    p = 'decoder.gru_layer.'
    pre0 = z @ sd['decoder.fc_h0.weight'].t() + sd['decoder.fc_h0.bias']
    hs, _ = gru_seq([torch.relu(pre0)] * T, sd[p + 'weight_ih_l0'], sd[p + 'weight_hh_l0'], sd[p + 'bias_ih_l0'],
                    sd[p + 'bias_hh_l0'])
    h_out = torch.stack(hs, dim=1)
    pres = [pre0] + [h_out @ sd[n + '.0.weight'].t() + sd[n + '.0.bias'] for n in ('decoder.add_vertex', 'decoder.add_edges')]
    return min(float(v.abs().min()) for v in pres)


def train_loss(sd, x, eps, bidir, node_type, beta=0.005):
    """Svae.training_step (svae_model.py:163-182) with eps injected into reparamterize (:107-110).
    -> (loss, edge_loss, vertex_loss, kl, mu, logvar, type_scores, edge_scores)"""
    mu, logvar = encoder(sd, x, bidir)
    z = eps * torch.exp(0.5 * logvar) + mu
    ts, es = decoder(sd, z, x.shape[1])
    C = node_type
    edge = F.binary_cross_entropy_with_logits(es, x[:, :, C:], reduction='sum')
    target = torch.argmax(x[:, :, :C].reshape(-1, C), dim=1)
    vertex = F.cross_entropy(ts.reshape(-1, C), target, reduction='sum')
    kl = -0.5 * torch.sum(1 + logvar - mu ** 2 - logvar.exp())
    return edge + vertex + beta * kl, edge, vertex, kl, mu, logvar, ts, es


def predict_from_logits(ts, es, u_type=None, u_edge=None):
    """Svae._predict_from_logits (svae_model.py:112-131).  Greedy: argmax / sigmoid > 0.5.  With
    uniforms: inverse-CDF draws (the build's injectable stand-in for torch.multinomial / bernoulli)."""
    C = ts.shape[-1]
    if u_type is None:
        pick = torch.argmax(ts, dim=-1)
        flags = (torch.sigmoid(es) > 0.5).float()
    else:
        m = ts.max(dim=-1, keepdim=True).values
        e = torch.exp(ts - m)
        cdf = torch.cumsum(e, dim=-1)
        u = (u_type * e.sum(dim=-1)).unsqueeze(-1)
        pick = torch.clamp((cdf <= u).sum(dim=-1), max=C - 1)
        flags = (u_edge < torch.sigmoid(es)).float()
    return F.one_hot(pick, num_classes=C).float(), flags


def min_margin(ts, es):
    """Distance of the greedy decisions from their thresholds (tests skip comparing closer calls)."""
    top = torch.topk(ts, 2, dim=-1).values
    return min(float((top[..., 0] - top[..., 1]).min()), float(es.abs().min()))

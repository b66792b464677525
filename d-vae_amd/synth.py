"""Synthetic DAG batches.  Generated on the CPU from a
seeded torch.Generator so that every rank / every box sees the same graphs."""
import torch


def random_dags(B, N, C, p, seed, chain=False, fix_orphans=True):
    """g [B,N,N] fp32 strictly lower-triangular (g[b,i,j]=1 <=> edge j->i),
    x [B,N,C] one-hot fp32.

    chain=True adds the (i-1)->i edge the ENAS loader always sets
    (dataloader.py:82-84); fix_orphans gives every node i>0 without a
    predecessor the edge 0->i (the SOS rule, dataloader.py:77-78)."""
    gen = torch.Generator().manual_seed(int(seed))
    g = torch.tril((torch.rand(B, N, N, generator=gen) < p).float(), diagonal=-1)
    if chain:
        idx = torch.arange(1, N)
        g[:, idx, idx - 1] = 1.0
    if fix_orphans:
        orphan = g.sum(dim=2) == 0
        orphan[:, 0] = False
        g[:, :, 0] = torch.where(orphan, torch.ones_like(g[:, :, 0]), g[:, :, 0])
    types = torch.randint(0, C, (B, N), generator=gen)
    x = torch.nn.functional.one_hot(types, C).float()
    return g, x


def enas_like(B, seed, N=8, C=8):
    """Headline shape: 8-node ENAS-like DAGs.  The chain edge (i-1)->i is always
    present and every other lower-triangular entry is drawn with p = 1/3, which
    gives E ~ 14 (density 0.5 of the 28 entries) and depth 7 -- the statistics of
    final_structures6.txt (mean E 13.9, density 0.497, depth 7)."""
    return random_dags(B, N, C, 1.0 / 3.0, seed, chain=True)


def ragged_dags(B, C, seed, nmin=8, nmax=128):
    """Mixed-size batch (config 4): sizes ~ U{nmin..nmax}, p = min(0.5, 4/n);
    returns zero-padded g [B,nmax,nmax], x [B,nmax,C] and int32 sizes [B]."""
    gen = torch.Generator().manual_seed(int(seed))
    sizes = torch.randint(nmin, nmax + 1, (B,), generator=gen).to(torch.int32)
    g = torch.zeros(B, nmax, nmax)
    x = torch.zeros(B, nmax, C)
    for b in range(B):
        n = int(sizes[b])
        gb, xb = random_dags(1, n, C, min(0.5, 4.0 / n), int(seed) * 100003 + b)
        g[b, :n, :n] = gb[0]
        x[b, :n] = xb[0]
    return g, x, sizes


def graph_stats(g, sizes=None):
    """(mean edges per graph, mean depth) -- reported next to every result."""
    B, N, _ = g.shape
    adj = g != 0
    lev = torch.zeros(B, N, dtype=torch.long)
    for i in range(N):
        for u in range(i):
            lev[:, i] = torch.where(adj[:, i, u], torch.maximum(lev[:, i], lev[:, u] + 1), lev[:, i])
    return float(g.sum() / B), float(lev.max(dim=1).values.float().mean())

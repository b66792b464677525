"""ENAS / Bayesian-network structure files -> batches for the D-VAE path.

Mirror of the reference's data layer (dvae/utils/dataloader.py:10-127): same text
format, same SOS/EOS/chain rules, same 80/10/10 split (pandas ``sample`` with
``random_state=42``), same batch dict ``{'graph','node_encoding','acc'}``, same
``get_dataloaders`` return value -- including the 1-tuple ``val_loader`` that
``Dvae.val_dataloader`` indexes with ``[0]`` (dataloader.py:121-123,
dvae_model.py:485).  Host-side only; nothing here is on the timed path.
"""
import json

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

_SHAPES = {'enas': (6, 6), 'bn': (8, 8)}      # task -> (node types, nodes) before SOS/EOS


def parse_line(line):
    """``[[t, p1, p2..], ...], acc`` -> (rows, acc)."""
    body, acc = line.rsplit(']],', 1)
    return json.loads(body + ']]'), float(acc)


def encode_structure(rows, task_type='enas'):
    """One structure -> (adjacency [n+2,n+2], one-hot [n+2,types+2]) with the
    start node at 0 and the end node last (dataloader.py:57-88):
    node k+1's parent flags fill columns 1.., a node without parents hangs off the
    start node, ENAS additionally always links the previous node, and every node
    nobody consumes feeds the end node."""
    types, n = _SHAPES[task_type]
    assert len(rows) == n, 'expected %d nodes, got %d' % (n, len(rows))
    N, C = n + 2, types + 2
    x = np.zeros((N, C), dtype=np.float32)
    g = np.zeros((N, N), dtype=np.float32)
    x[0, C - 2] = 1.0
    x[N - 1, C - 1] = 1.0
    childless = set(range(n))
    for k, info in enumerate(rows):
        i = k + 1
        x[i, info[0]] = 1.0
        parents = np.asarray(info[1:], dtype=np.float32)
        g[i, 1:1 + len(parents)] = parents
        if parents.sum() == 0:
            g[i, 0] = 1.0
        else:
            childless -= set(np.nonzero(parents)[0].tolist())
        if task_type == 'enas':
            g[i, k] = 1.0
    for k in childless:
        g[N - 1, k + 1] = 1.0
    return torch.from_numpy(g), torch.from_numpy(x)


def split_indices(n):
    """Row indices of the reference's train/val/test split (dataloader.py:35-38)."""
    import pandas as pd
    df = pd.DataFrame(index=range(n))
    train = df.sample(frac=0.8, random_state=42)
    rest = df.drop(train.index)
    val = rest.sample(frac=0.5, random_state=42)
    test = rest.drop(val.index)
    return {'train': train.index.to_numpy(), 'val': val.index.to_numpy(), 'test': test.index.to_numpy()}


class NASDataset(Dataset):
    def __init__(self, file_path, split='train', fmt=None, task_type='enas'):
        super(NASDataset, self).__init__()
        if task_type not in _SHAPES or split not in ('train', 'val', 'test'):
            raise NotImplementedError
        self.file_path, self.fmt, self.task_type = file_path, fmt, task_type
        self.node_types, self.seq_len = _SHAPES[task_type]
        with open(file_path) as f:
            self.lines = [ln for ln in (l.strip() for l in f) if ln]
        self.index = split_indices(len(self.lines))[split]

    def __len__(self):
        return len(self.index)

    def __getitem__(self, i):
        rows, acc = parse_line(self.lines[int(self.index[i])])
        g, x = encode_structure(rows, self.task_type)
        if self.fmt == 'str':       # S-VAE string form (dataloader.py:96-103)
            return {'graph': torch.cat((x[1:], g[1:]), -1), 'acc': acc}
        return {'graph': g, 'node_encoding': x, 'acc': acc}


def get_dataloaders(batch_size, file_path, fmt=None, task_type='enas'):
    mk = lambda split, shuffle: DataLoader(NASDataset(file_path, split, fmt, task_type), batch_size=batch_size,
                                           pin_memory=True, shuffle=shuffle)
    return mk('train', True), (mk('val', False),), mk('test', False)


class ScheduledLoader:
    """Loader wrapper for the fused train step ("emit CSR + levels directly"): walks
    `loader` ONE BATCH AHEAD, copies the next batch to the model's device and queues its level schedule --
    levels, CSR / CSC lists, prefix rows, built on the device from the dense adjacency by
    `Dvae.prepare` -- on a stream of its own, so that the schedule's host round trip is over by the
    time the step needs it.  Yields `(batch_on_device, schedule)`; pass both to
    `model.loss_and_grad(batch, sched=schedule)`.  The dense `[B,N,N]` batch format of the reference
    (dataloader.py:104-108) is kept: at 256 bytes per 8-node graph the device-side build (~30 us per
    batch) is cheaper than any host-side CSR."""

    def __init__(self, loader, model, shard=None, on_side_stream=False):
        # on_side_stream: build the schedules on the library's side stream instead of a third stream of this
        # process (Dvae.prepare) -- for processes that also hold RCCL's streams (4 hardware queues per process)
        self.loader, self.model, self.shard, self.on_side_stream = loader, model, shard, on_side_stream

    def __len__(self):
        if not hasattr(self.loader, '__len__'):
            raise TypeError("the wrapped loader (%s) has no length" % type(self.loader).__name__)
        return len(self.loader)

    def _stage(self, batch):
        dev = next(self.model.parameters()).device
        if self.shard is not None:
            batch = self.shard(batch)
        batch = {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
        return batch, self.model.prepare(batch, on_side_stream=self.on_side_stream)   # prepare() orders its stream after the copies

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        for raw in it:
            cur, nxt = nxt, self._stage(raw)
            yield cur
        yield nxt

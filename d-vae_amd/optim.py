"""The optimizer of the reference's train step -- `torch.optim.Adam(self.parameters(), lr=1e-4)`
(dvae_model.py:471-475) -- over flat buffers: one HIP launch per module instead of torch's multi-tensor
launch over ~40 parameter tensors (110 us -> 35 us at hs=501; the step has nothing to overlap it with).

`FlatAdam` IS a `torch.optim.Optimizer`: `param_groups[0]['lr']` drives it (ReduceLROnPlateau works
unchanged) and its `state_dict()` has torch Adam's layout (`step`, `exp_avg`, `exp_avg_sq` per parameter), so
checkpoints move between the two.  The model's parameters become views of one flat buffer per module (same
`nn.Parameter` objects, same values), laid out like the flat gradient buffers `Dvae.loss_and_grad` fills.
"""
import ctypes

import torch

from . import engine as E


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        enc_p, dec_p = model.encoder.hip_params(), model.decoder.hip_params()
        every = enc_p + dec_p
        if len({id(p) for p in every}) != len(every):
            raise ValueError("FlatAdam: modules share parameters (same_mapper_gater); use torch.optim.Adam")
        if {id(p) for p in every} != {id(p) for p in model.parameters()}:
            raise ValueError("FlatAdam: the model has parameters outside encoder / decoder")
        for p in every:
            if not p.is_cuda or p.dtype != torch.float32:
                raise ValueError("FlatAdam: fp32 parameters on a GPU")
        super().__init__(every, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._model = model
        self._t = 0
        self._mods = []
        with torch.no_grad():
            for plist in (enc_p, dec_p):
                flat = torch.cat([p.detach().reshape(-1) for p in plist])
                m, v = torch.zeros_like(flat), torch.zeros_like(flat)
                o = 0
                for p in plist:
                    n = p.numel()
                    p.data = flat[o:o + n].view_as(p)           # same Parameter object, storage inside `flat`
                    self.state[p] = {'step': torch.tensor(0.0), 'exp_avg': m[o:o + n].view_as(p),
                                     'exp_avg_sq': v[o:o + n].view_as(p)}
                    o += n
                self._mods.append((plist, flat, m, v))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        fe, fd, ve, vd = self._model._flat_grads()
        self._t += 1
        lib = E.load()
        stream = E._stream()
        for (plist, flat, m, v), fg, views in zip(self._mods, (fe, fd), (ve, vd)):
            # loss_and_grad leaves p.grad as views of the flat gradient buffer; gradients from autograd are copied in
            for p, gv in zip(plist, views):
                if p.grad is None:
                    gv.zero_()
                elif p.grad.data_ptr() != gv.data_ptr():
                    gv.copy_(p.grad)
            E.check(lib.dvae_adam_step(E._ptr(flat), E._ptr(fg), E._ptr(m), E._ptr(v), flat.numel(), float(g['lr']),
                                       float(g['betas'][0]), float(g['betas'][1]), float(g['eps']),
                                       float(g['weight_decay']), self._t, stream))
        return loss

    def state_dict(self):
        for st in self.state.values():
            st['step'] = torch.tensor(float(self._t))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        views = {id(p): (st['exp_avg'], st['exp_avg_sq']) for p, st in self.state.items()}
        super().load_state_dict(state_dict)
        steps = set()
        with torch.no_grad():
            for p, st in self.state.items():
                m, v = views[id(p)]
                m.copy_(st['exp_avg'])
                v.copy_(st['exp_avg_sq'])
                steps.add(int(float(st['step'])))
                st['exp_avg'], st['exp_avg_sq'] = m, v
        if len(steps) > 1:
            raise ValueError("FlatAdam: parameters with different step counts")
        self._t = steps.pop() if steps else 0

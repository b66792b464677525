"""Training shell around the D-VAE path: what the reference gets from pytorch_lightning
(train.py:39-47 `Trainer(max_nb_epochs=..., default_save_path=...).fit(model)`), reduced to the
hooks `Dvae` defines -- `configure_optimizers` (Adam 1e-4 + ReduceLROnPlateau,
dvae_model.py:471-475), `train_dataloader` / `val_dataloader`, `training_step`,
`validation_step` / `validation_end` -- plus checkpoint files in Lightning's
`{'state_dict': ...}` layout that `prediction.py:14-22` reads back by key name.
Host-side only; nothing here is on the timed path.
"""
import os

import torch


def save_checkpoint(path, model, optimizer=None, scheduler=None, epoch=0, extra=None):
    """`{'epoch', 'state_dict', 'optimizer_states', 'lr_schedulers'}` -- the keys a Lightning
    checkpoint of the reference carries; tensors are moved to the CPU."""
    ckpt = {'epoch': int(epoch),
            'state_dict': {k: v.detach().cpu() for k, v in model.state_dict().items()}}
    if optimizer is not None:
        ckpt['optimizer_states'] = [optimizer.state_dict()]
    if scheduler is not None:
        ckpt['lr_schedulers'] = [scheduler.state_dict()]
    if extra:
        ckpt.update(extra)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = path + '.tmp'
    torch.save(ckpt, tmp)
    os.replace(tmp, path)          # a crash never leaves a truncated checkpoint behind
    return path


def load_checkpoint(path, model=None, optimizer=None, scheduler=None, strict=True):
    """Read a checkpoint written by `save_checkpoint` or by the reference's Lightning run
    (`{'state_dict': ...}`), or a bare state_dict.  Loaded with `weights_only=True`: nothing in
    the file is executed.  Parameters are matched by NAME, so a reference
    checkpoint loads into `Dvae` unchanged.  Returns the checkpoint dict."""
    ckpt = torch.load(path, map_location='cpu', weights_only=True)
    if not isinstance(ckpt, dict):
        raise ValueError('%s does not hold a checkpoint dict' % path)
    sd = ckpt['state_dict'] if 'state_dict' in ckpt else ckpt
    if model is not None:
        model.load_state_dict(sd, strict=strict)
    if optimizer is not None and ckpt.get('optimizer_states'):
        optimizer.load_state_dict(ckpt['optimizer_states'][0])
    if scheduler is not None and ckpt.get('lr_schedulers'):
        scheduler.load_state_dict(ckpt['lr_schedulers'][0])
    return ckpt


def _to_device(batch, device):
    return {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}


class Trainer:
    """Epoch loop over the model's own hooks.

    max_epochs        train.py:44 `max_nb_epochs`
    save_dir          train.py:41 `default_save_path`; one `_ckpt_epoch_<n>.ckpt` per epoch
                      (the file name prediction.py:14 loads), older ones removed unless keep_all
    fused             use `model.loss_and_grad` (forward + hand-written backward, gradients written
                      straight into flat buffers) when the model has it; else
                      `training_step(...)['loss'].backward()` through autograd
    flat_adam         with `fused`: the model's Adam as one launch per module over flat buffers (optim.FlatAdam; same
                      update, same checkpoint layout as torch.optim.Adam) when `configure_optimizers` offers it
    dist              torch.distributed module (initialised) for data-parallel runs: every rank
                      trains on its shard of each batch, gradients are SUM all-reduced (dp.py)
    A batch dict may carry 'eps' [B,Z]: the reparameterisation draw per GLOBAL graph index (sharded with the
    batch), which makes a data-parallel run reproduce the single-device one; without it every rank draws its own.
    """

    def __init__(self, max_epochs=300, save_dir=None, fused=True, dist=None, keep_all=False, log=print,
                 max_steps_per_epoch=None, flat_adam=True):
        self.max_epochs, self.save_dir, self.fused, self.dist = int(max_epochs), save_dir, fused, dist
        self.flat_adam = flat_adam
        self.keep_all, self.log, self.max_steps = keep_all, log, max_steps_per_epoch
        self.history = []
        self.epoch = 0

    def _device(self, model):
        return next(model.parameters()).device

    def _world(self):
        d = self.dist
        return (d.get_rank(), d.get_world_size()) if d is not None and d.is_initialized() else (0, 1)

    def fit(self, model, resume_from=None):
        from . import dp
        import inspect
        flat = self.flat_adam and self.fused and 'flat' in inspect.signature(model.configure_optimizers).parameters
        (opt,), (sched,) = model.configure_optimizers(flat=True) if flat else model.configure_optimizers()
        if resume_from:
            self.epoch = load_checkpoint(resume_from, model, opt, sched).get('epoch', -1) + 1
        dev = self._device(model)
        rank, world = self._world()
        sync = dp.make_grad_sync(self.dist)
        fused = self.fused and hasattr(model, 'loss_and_grad')
        if world > 1:
            # data parallel = the single-device big-batch step (dp.py) only if every rank starts from the
            # same weights and walks the same shuffled batches: rank 0's parameters and a per-epoch
            # shuffle seed are broadcast
            with torch.no_grad():
                for t in list(model.parameters()) + list(model.buffers()):
                    self.dist.broadcast(t.data, 0)
        while self.epoch < self.max_epochs:
            model.train()
            sums, steps = torch.zeros(4, dtype=torch.float64), 0
            if world > 1:
                seed = torch.randint(0, 2 ** 31 - 1, (1,), dtype=torch.int64).to(dev)
                self.dist.broadcast(seed, 0)
                torch.manual_seed(int(seed.item()))     # the loader's RandomSampler seeds itself from this generator
            loader = model.train_dataloader()
            prepared = fused and hasattr(model, 'prepare')      # Dvae: level schedules built ahead of the step
            if prepared:
                # D-VAE fused path: batches staged and their level schedules built one batch ahead
                from .data import ScheduledLoader

                def stage(b):
                    if world > 1 and next(v for v in b.values() if torch.is_tensor(v)).shape[0] < world:
                        return None
                    return dp.shard_batch(b, rank, world) if world > 1 else b
                # with a collective library in the process the schedules are built on the library's side stream, not on
                # a third stream (4 hardware queues per process; bench.py does the same)
                loader = ScheduledLoader((b for b in (stage(raw) for raw in loader) if b is not None), model,
                                         on_side_stream=world > 1)
            for i, item in enumerate(loader):
                if self.max_steps is not None and i >= self.max_steps:
                    break
                if prepared:
                    batch, level_sched = item
                    out4 = model.loss_and_grad(batch, eps=batch.get('eps'), grad_sync=sync, sched=level_sched)
                    opt.step()
                    sums += dp.allreduce_losses(self.dist, out4.detach().clone()).double().cpu()
                    steps += 1
                    continue
                batch = item
                if world > 1 and next(v for v in batch.values() if torch.is_tensor(v)).shape[0] < world:
                    continue    # a tail batch with fewer graphs than ranks would leave a rank without work (the
                                # reference's loader has no drop_last, dataloader.py:118-126): all ranks drop it
                batch = _to_device(dp.shard_batch(batch, rank, world) if world > 1 else batch, dev)
                if fused:
                    out4 = model.loss_and_grad(batch, eps=batch.get('eps'), grad_sync=sync)
                else:
                    opt.zero_grad(set_to_none=True)
                    out = model.training_step(batch, i)
                    out['loss'].backward()
                    if sync is not None:
                        for p in model.parameters():
                            if p.grad is not None:
                                sync(p.grad).wait()
                    out4 = torch.tensor([out['log'][k] for k in ('train/edge_loss', 'train/vertex_loss',
                                                                 'train/kl_loss', 'train/loss')])
                opt.step()
                sums += dp.allreduce_losses(self.dist, out4.detach().clone()).double().cpu()
                steps += 1
            rec = {'epoch': self.epoch, 'steps': steps}
            if steps:
                for k, v in zip(('train/edge_loss', 'train/vertex_loss', 'train/kl_loss', 'train/loss'), sums / steps):
                    rec[k] = float(v)
            val = self.validate(model)
            if val is not None:
                rec.update({k: float(v) for k, v in val['log'].items()})
                sched.step(float(val['avg_val_loss']))       # ReduceLROnPlateau monitors the validation loss
            rec['lr'] = opt.param_groups[0]['lr']
            self.history.append(rec)
            self.log(rec)
            if self.save_dir and rank == 0:
                path = os.path.join(self.save_dir, '_ckpt_epoch_%d.ckpt' % self.epoch)
                save_checkpoint(path, model, opt, sched, self.epoch)
                old = os.path.join(self.save_dir, '_ckpt_epoch_%d.ckpt' % (self.epoch - 1))
                if not self.keep_all and os.path.exists(old):
                    os.remove(old)
            self.epoch += 1
        return self.history

    def validate(self, model):
        loader = model.val_dataloader()
        if loader is None:
            return None
        dev = self._device(model)
        model.eval()
        outputs = []
        with torch.no_grad():
            for i, batch in enumerate(loader):
                if self.max_steps is not None and i >= self.max_steps:
                    break
                outputs.append(model.validation_step(_to_device(batch, dev), i))
        return model.validation_end(outputs) if outputs else None

"""One optimisation step of the reference's training loop (scripts/train_drug3d.py:88-109, scripts/train_bond.py) on the
MI355X path: ``get_loss`` forward + backward on the HIP layer operators, gradient all-reduce across data-parallel
ranks (one process per GPU, RCCL), global-norm clipping and AdamW as single kernels over flat buffers.

Layout: all trainable parameters are re-pointed into ONE contiguous fp32 buffer (``FlatParams``), their ``.grad`` into
a second one, so that
  * autograd accumulates every layer's weight gradient straight into the flat gradient buffer,
  * data parallelism is one large all-reduce per step over that buffer (xGMI rings are per-link bound: few large
    collectives beat many small ones; the buffer is ~10 M floats),
  * clip_grad_norm_ + AdamW (utils/train.py:64-70: lr 1e-4, betas (0.99, 0.999), weight_decay 1e-8; max_grad_norm 50)
    are two kernel launches (``mdx_op_sumsq``, ``mdx_op_adamw``) regardless of the number of parameter tensors.
The reference trains under fp16 autocast with a GradScaler (use_amp: True, scripts/train_drug3d.py:86-109).  `precision='fp16'`
is that arithmetic (train_ops.precision) with the scaler's dynamic loss scale kept ON THE DEVICE: the loss is multiplied by a device
scalar, one library call unscales, measures the norm, clips, steps or skips and grows / backs off the scale -- no host round trip
(torch's GradScaler synchronises on found_inf every step).  `precision='f32'` runs the same call with the scale pinned to 1.
"""
import math

import torch

from . import _lib


class FlatParams:
    """Views of a module's trainable parameters (and their gradients) inside two flat buffers.  Pure memory layout:
    works on any device (the CPU tests exercise the all-reduce bookkeeping with it)."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        dev = self.params[0].device
        self.numel = sum(p.numel() for p in self.params)
        self.data = torch.empty(self.numel, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            self.data[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.data[off:off + n].view(p.shape)
            p.grad = self.grad[off:off + n].view(p.shape)
            off += n

    def zero_grad(self):
        self.grad.zero_()
        off = 0
        for p in self.params:      # re-attach in case someone set .grad = None
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + n].view(p.shape)
            off += n


def allreduce_mean_(flat_grad, group=None):
    """Average the flat gradient over the data-parallel ranks (no-op for a single process)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    world = dist.get_world_size(group)
    if world > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
        flat_grad.mul_(1.0 / world)
    return world


def broadcast_replicas_(buffers, src=0, group=None):
    """Make every data-parallel rank hold rank `src`'s copy of the given flat buffers (no-op for a single process)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) <= 1:
        return 1
    for buf in buffers:
        dist.broadcast(buf, src, group=group)
    return dist.get_world_size(group)


class PlateauScheduler:
    """torch.optim.lr_scheduler.ReduceLROnPlateau (mode='min', rel. threshold 1e-4) as host logic on a Trainer's lr
    (utils/train.py get_scheduler 'plateau': factor, patience, min_lr)."""

    def __init__(self, trainer, factor=0.8, patience=1000, min_lr=1e-5, threshold=1e-4):
        self.trainer, self.factor, self.patience, self.min_lr, self.threshold = trainer, factor, patience, min_lr, threshold
        self.best, self.bad = math.inf, 0

    def step(self, metric):
        metric = float(metric)
        if metric < self.best * (1.0 - self.threshold):
            self.best, self.bad = metric, 0
        else:
            self.bad += 1
        if self.bad > self.patience:
            self.trainer.lr = max(self.trainer.lr * self.factor, self.min_lr)
            self.bad = 0

    def state_dict(self):
        return {'best': self.best, 'bad': self.bad}


class Trainer:
    def __init__(self, model, lr=1e-4, betas=(0.99, 0.999), eps=1e-8, weight_decay=1e-8, max_grad_norm=50.0, group=None,
                 precision='f32', init_scale=None, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        p0 = next(model.parameters())
        _lib._need_gpu(p0)
        from . import train_ops
        if precision not in train_ops.KINDS:
            raise ValueError(precision)
        self.model, self.group = model, group
        self._mods = None
        self._deferred = None
        # 'f32' | 'bf16' (GEMM operands only) | 'fp16' (the reference's autocast + GradScaler semantics) | 'bf16_autocast'
        self.precision = precision
        self.flat = FlatParams(model)
        self.lr, self.betas, self.eps, self.weight_decay, self.max_grad_norm = lr, betas, eps, weight_decay, max_grad_norm
        dev = p0.device
        self.m = torch.zeros_like(self.flat.data)
        self.v = torch.zeros_like(self.flat.data)
        self.ws = torch.empty(1024, dtype=torch.float32, device=dev)
        # torch.cuda.amp.GradScaler's defaults (init 2**16, x2 after 2000 clean steps, x0.5 on overflow) for float16; every other
        # mode needs no scaling: scale pinned to 1, same code path
        # (derived from the arithmetic, not the mode's name: every float16-autocast mode -- 'fp16' and 'fp16_f32store' -- underflows
        # without a loss scale)
        kind = train_ops.KINDS[precision]
        scaled = kind is not None and kind[0] == 2 and kind[1]
        self._scaled = scaled
        self.growth = (float(growth_factor), float(backoff_factor), int(growth_interval)) if scaled else (1.0, 1.0, 0)
        self.state = torch.zeros(16, dtype=torch.float32, device=dev)     # mdx_op_amp_adamw: [scale, tracker, steps, skipped, norm2, ...]
        self.state[0] = float(init_scale if init_scale is not None else (65536.0 if scaled else 1.0))
        # W^T of every weight for the grad_input GEMMs: one launch per step instead of one per layer (train_ops.TransposedParams)
        self.wt = train_ops.TransposedParams(self.flat)
        self.sync_replicas()

    @property
    def steps(self):
        """optimizer steps actually taken (skipped steps do not count, like torch's GradScaler + optimizer); reads the device state"""
        return int(self.state[2].item())

    @property
    def skipped(self):
        return int(self.state[3].item())

    @property
    def loss_scale(self):
        return float(self.state[0].item())

    def sync_replicas(self, src=0):
        """Data-parallel replicas must start from ONE set of weights: broadcast rank `src`'s flat parameter buffer (nn.Linear /
        LayerNorm default initialisation draws from each process's own torch RNG stream) and the optimizer moments."""
        broadcast_replicas_((self.flat.data, self.m, self.v), src, self.group)
        self._stale()

    def _stale(self):
        # the fused sampling/validation engine caches a packed copy of the weights keyed on tensor versions; the optimizer
        # kernel writes through raw pointers, so drop that cache explicitly
        if self._mods is None:
            self._mods = list(self.model.modules())     # (walking the module tree costs ~0.7 ms per step)
        for mod in self._mods:
            if hasattr(mod, '_eng_sig'):
                mod._eng_sig = None

    def zero_grad(self):
        self.flat.zero_grad()

    def backward_and_step(self, loss):
        """(scale x loss).backward() -> gradient averaging over ranks -> unscale, clip, AdamW or skip, scale update (one library
        call, state on the device).  Returns the pre-clip global gradient norm of the UNSCALED gradient (0-d device tensor, like
        clip_grad_norm_ after scaler.unscale_; inf / nan on a skipped step)."""
        L = _lib.lib()
        from . import train_ops
        (loss * self.state[0]).backward()
        train_ops.flush_grad_sink()      # weight / LayerNorm gradients recorded during backward -> flat gradient buffer, one launch
        allreduce_mean_(self.flat.grad, self.group)
        f = self.flat
        max_norm = float(self.max_grad_norm) if self.max_grad_norm is not None else 0.0
        _lib.check(L.mdx_op_amp_adamw(_lib.ptr(f.data), _lib.ptr(f.grad), _lib.ptr(self.m), _lib.ptr(self.v), f.numel, self.lr,
                                      self.betas[0], self.betas[1], self.eps, self.weight_decay, max_norm, _lib.ptr(self.state),
                                      self.growth[0], self.growth[1], self.growth[2], _lib.ptr(self.ws), _lib.stream()))
        self._stale()
        return self.state[4].sqrt()

    def step(self, *batch, **kw):
        """zero_grad -> model.get_loss(*batch) -> backward_and_step.  Returns the loss dict plus 'grad_norm'."""
        from . import train_ops
        self.zero_grad()
        # grad_sink: parameter gradients of the HIP layer operators bypass autograd's accumulation and are reduced straight into the
        # flat gradient buffer by one launch (train_ops.flush_grad_sink)
        from .diffusion import deferred_class_checks
        self.check_deferred()        # the previous step's class-range assert (its values reached the host long ago)
        self.wt.refresh()
        with train_ops.grad_sink(self.flat), train_ops.transposed_params(self.wt):
            with train_ops.precision(self.precision), deferred_class_checks() as chk:
                out = self.model.get_loss(*batch, **kw)
            self._deferred = chk.finish(group=self.group)
            gn = self.backward_and_step(out['loss'])
        res = {k: v.detach() for k, v in out.items()}
        res['grad_norm'] = gn
        return res

    def check_deferred(self):
        """Raise the class-range AssertionError (models/diffusion.py:54) of the last step's batch, if any: Trainer.step postpones that
        check by one step instead of waiting for the GPU at the top of every step (diffusion.deferred_class_checks)."""
        verify, self._deferred = self._deferred, None
        if verify is not None:
            verify()

    def state_dict(self):
        return {'m': self.m, 'v': self.v, 'steps': self.steps, 'lr': self.lr, 'amp_state': self.state.clone(),
                'amp_scaled': bool(self._scaled), 'precision': self.precision}

    def load_state_dict(self, sd):
        self.m.copy_(sd['m']); self.v.copy_(sd['v'])
        self.lr = float(sd['lr'])
        if 'amp_state' in sd:
            self.state.copy_(sd['amp_state'])
            # a checkpoint written in another precision mode carries that mode's loss scale: an unscaled mode must run at scale 1
            # (its growth is disabled), a scaled mode must not start from the 1.0 an unscaled run stored
            if not self._scaled:
                self.state[0], self.state[1] = 1.0, 0.0
            elif not sd.get('amp_scaled', float(sd['amp_state'][0]) != 1.0):
                # written by an unscaled mode (checkpoints older than the 'amp_scaled' flag: inferred from a scale of exactly 1.0; a
                # float16 run that legitimately backed off to 2^0 is told apart by the flag)
                self.state[0], self.state[1] = 65536.0, 0.0
        else:
            self.state[2] = float(sd['steps'])

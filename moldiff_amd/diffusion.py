"""Noise schedules and log-space helpers of the denoising path (host side, float64 numpy).

Mirrors the names the reference exposes in ``models/diffusion.py`` so call sites read the same:
``get_beta_schedule`` (:153-192), ``advance_schedule`` (:110-131), ``segment_schedule`` (:133-148),
``to_torch_const`` (:41-44), ``extract`` (:60-72), ``index_to_log_onehot`` (:53-57),
``log_sample_categorical`` (:79-85).  The schedules run once at model construction; nothing here is
on the per-step device path (the per-step math lives in ``csrc/transition.hip``).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def advance_schedule(timesteps, scale_start, scale_end, width, return_alphas_bar=False):
    """alphas_bar follows a rescaled sigmoid from `scale_start` down to `scale_end`."""
    amp = (scale_end - scale_start) / (sigmoid(-width) - sigmoid(width))
    shift = 0.5 * (scale_end + scale_start - amp)
    alphas_bar = amp * sigmoid(-width * np.linspace(-1, 1, timesteps)) + shift
    betas = _betas_of(alphas_bar)
    return (betas, alphas_bar) if return_alphas_bar else betas


def _betas_of(alphas_bar):
    ratio = np.concatenate([alphas_bar[:1], alphas_bar[1:] / alphas_bar[:-1]])
    return np.clip(1.0 - ratio, 0, 1)


def segment_schedule(timesteps, time_segment, segment_diff):
    """Piecewise 'advance' curve; each piece is drawn on n+1 points and its first point dropped."""
    assert np.sum(time_segment) == timesteps
    chunks = [advance_schedule(int(n) + 1, return_alphas_bar=True, **dict(p))[1][1:]
              for n, p in zip(time_segment, segment_diff)]
    return _betas_of(np.concatenate(chunks))


def get_beta_schedule(beta_schedule, num_timesteps, **kwargs):
    """Same selector strings and keyword names as the reference."""
    T = num_timesteps
    if beta_schedule == 'quad':
        betas = np.linspace(kwargs['beta_start'] ** 0.5, kwargs['beta_end'] ** 0.5, T, dtype=np.float64) ** 2
    elif beta_schedule == 'linear':
        betas = np.linspace(kwargs['beta_start'], kwargs['beta_end'], T, dtype=np.float64)
    elif beta_schedule == 'const':
        betas = kwargs['beta_end'] * np.ones(T, dtype=np.float64)
    elif beta_schedule == 'jsd':
        betas = 1.0 / np.linspace(T, 1, T, dtype=np.float64)
    elif beta_schedule == 'sigmoid':
        s = kwargs.get('s', 6)
        betas = sigmoid(np.linspace(-s, s, T)) * (kwargs['beta_end'] - kwargs['beta_start']) + kwargs['beta_start']
    elif beta_schedule == 'cosine':
        s = kwargs.get('s', 0.008)
        x = np.linspace(0, T + 1, T + 1)
        ab = np.cos(((x / (T + 1)) + s) / (1 + s) * np.pi * 0.5) ** 2
        ab = ab / ab[0]
        betas = np.clip(1 - ab[1:] / ab[:-1], 0, 0.999)
    elif beta_schedule == 'advance':
        betas = advance_schedule(T, kwargs.get('scale_start', 0.999), kwargs.get('scale_end', 0.001),
                                 kwargs.get('width', 2))
    elif beta_schedule == 'segment':
        betas = segment_schedule(T, kwargs['time_segment'], kwargs['segment_diff'])
    else:
        raise NotImplementedError(beta_schedule)
    assert betas.shape == (T,)
    return betas


def to_torch_const(x):
    """Frozen float32 parameter: ends up in state_dict (checkpoint contract) but never trains."""
    return nn.Parameter(torch.from_numpy(np.asarray(x)).float(), requires_grad=False)


def extract(coef, t, batch, ndim=2):
    out = coef[t][batch]
    if ndim == 1:
        return out
    if ndim == 2:
        return out.unsqueeze(-1)
    if ndim == 3:
        return out.unsqueeze(-1).unsqueeze(-1)
    raise NotImplementedError('ndim > 3')


class deferred_class_checks:
    """Context: the class-range assert of `index_to_log_onehot` (models/diffusion.py:54) reads a device value, i.e. makes the host
    wait for the GPU -- twice at the top of every training step, where it keeps the host from issuing the step while the previous
    one still runs.  Inside this context the maxima are kept on the device; `finish()` starts ONE asynchronous copy of them to pinned
    memory and returns a callable that raises the same AssertionError when called later (Trainer.step: at the start of the next
    step, when the copy has long completed)."""
    _active = None

    def __enter__(self):
        self.prev, deferred_class_checks._active = deferred_class_checks._active, self
        self.items = []
        return self

    def __exit__(self, *a):
        deferred_class_checks._active = self.prev

    _limit_cache = {}

    def finish(self, group=None):
        """group: the process group the caller's gradient collectives run on (Trainer's `group`; None = WORLD).  The flag all-reduce
        must pair with THAT group's ranks -- on a sub-group, ranks outside it never reach this point."""
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        if not self.items and not multi:
            return None
        limits = [k for _, k in self.items]
        if self.items:
            dev = torch.stack([m for m, _ in self.items])
        else:   # nothing to check on this rank (empty batch): it still takes part in the collective below
            dev = torch.zeros(0, dtype=torch.int64, device='cuda' if torch.cuda.is_available() else 'cpu')
        # Multi-rank: every rank must fail in the SAME step, or the healthy ranks sit in the next gradient all-reduce until the
        # collective's watchdog ends them.  One MAX all-reduce of a flag (a device op on the step's stream, no host round trip; EVERY
        # rank issues it once per step at the same point, whatever its batch holds) tells everybody that somebody's batch is out of
        # range.
        peers = None
        if multi:
            if self.items:
                key = (tuple(limits), dev.dtype, dev.device)     # the limits tensor is built once per (class counts, device): no H2D copy per step
                lim = deferred_class_checks._limit_cache.get(key)
                if lim is None:
                    lim = deferred_class_checks._limit_cache[key] = torch.tensor(limits, dtype=dev.dtype, device=dev.device)
                bad = (dev >= lim).any()
            else:
                bad = torch.zeros((), dtype=torch.bool, device=dev.device)
            peers = bad.to(torch.int32).reshape(1)
            dist.all_reduce(peers, op=dist.ReduceOp.MAX, group=group)
            dev = torch.cat([dev, peers.to(dev.dtype)])
        host = torch.empty(dev.shape, dtype=dev.dtype, pin_memory=dev.is_cuda)
        host.copy_(dev, non_blocking=True)
        ev = None
        if dev.is_cuda:
            ev = torch.cuda.Event()
            ev.record()

        def verify():
            if ev is not None:
                ev.synchronize()
            vals = host.tolist()
            for m, k in zip(vals, limits):
                assert m < k, f'Error: {m} >= {k}'
            if peers is not None:
                assert vals[-1] == 0, 'Error: class id out of range on another rank (models/diffusion.py:54 raised there)'
        return verify


def check_class_range(x, num_classes):
    """the reference's class-range assert (models/diffusion.py:54) on its own: immediately, or recorded inside `deferred_class_checks`"""
    if x.numel():
        if deferred_class_checks._active is not None and x.is_cuda:
            deferred_class_checks._active.items.append((x.max(), num_classes))
        else:
            assert x.max().item() < num_classes, f'Error: {x.max().item()} >= {num_classes}'


def index_to_log_onehot(x, num_classes, checked=True):
    """log of the one-hot encoding, clamped at log(1e-30).  checked: the reference's assert on the class range (models/diffusion.py:54),
    a host-device synchronisation; callers whose ids come out of an argmax over `num_classes` logits pass False (in range by
    construction), and inside `deferred_class_checks` the comparison is postponed instead of skipped."""
    if checked and x.numel():
        if deferred_class_checks._active is not None and x.is_cuda:
            deferred_class_checks._active.items.append((x.max(), num_classes))
            # one_hot on an out-of-range id trips a device-side assert long before the postponed check would report it: encode
            # the clamped ids, so that the reference's AssertionError (one step later) is what the caller sees.  In a multi-rank run
            # finish() all-reduces the flag, so every rank raises in the same step.
            x = x.clamp(max=num_classes - 1)
        else:
            assert x.max().item() < num_classes, f'Error: {x.max().item()} >= {num_classes}'
    return torch.log(F.one_hot(x, num_classes).float().clamp(min=1e-30))


def log_sample_categorical(logits):
    """Gumbel-max draw using torch's generator (host/compat helper; the device path draws Philox noise)."""
    u = torch.rand_like(logits)
    return (logits - torch.log(-torch.log(u + 1e-30) + 1e-30)).argmax(dim=-1)


def categorical_kl(log_prob1, log_prob2):
    """KL(p1 || p2) over the last axis, both given as log-probabilities."""
    return (log_prob1.exp() * (log_prob1 - log_prob2)).sum(dim=-1)


def log_categorical(log_x_start, log_prob):
    """log-likelihood of the (log one-hot) `log_x_start` under `log_prob`."""
    return (log_x_start.exp() * log_prob).sum(dim=-1)

"""Forward/posterior transition kernels of the diffusion (host tables + device entry points).

Same class names, constructor arguments, frozen-parameter names (``state_dict`` contract, incl. the
reference's ``transpopse_q_onestep_mats`` spelling) and method signatures as the reference's
``models/transition.py`` (ContigousTransition :9-69, GeneralCategoricalTransition :178-339).  The tables
are built once on the host in float64; every per-row method that runs during sampling dispatches to
the HIP kernels in ``csrc/mdx_transition.hip`` through the C-ABI -- there is no CPU fallback.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .diffusion import to_torch_const, index_to_log_onehot, categorical_kl, log_categorical, check_class_range
from . import _lib

_FUSED_NOISE = __import__('os').environ.get('MDX_FUSED_ADD_NOISE', '1') != '0'


class ContigousTransition(nn.Module):
    """Gaussian diffusion over R^d (used for atom positions)."""

    def __init__(self, betas, num_classes=None, scaling=1.):
        super().__init__()
        self.num_classes = num_classes
        self.scaling = scaling
        alphas = 1. - betas
        abar = np.cumprod(alphas, axis=0)
        abar_prev = np.concatenate([[1.], abar[:-1]])
        self.betas = to_torch_const(betas)
        self.alphas = to_torch_const(alphas)
        self.alphas_bar = to_torch_const(abar)
        self.alphas_bar_prev = to_torch_const(abar_prev)
        # coefficients of q(x_{t-1} | x_0, x_t)
        self.coef_x0 = to_torch_const(np.sqrt(abar_prev) * betas / (1 - abar))
        self.coef_xt = to_torch_const(np.sqrt(alphas) * (1 - abar_prev) / (1 - abar))
        self.std = to_torch_const(np.sqrt((1 - abar_prev) * betas / (1 - abar)))

    def add_noise(self, x, time_step, batch, eps=None):
        """q(x_t | x_0); loss-side helper (not on the sampling path), plain torch ops on `x`'s device.
        `eps` may be injected (parity tests), otherwise drawn from torch's generator like the reference."""
        if self.num_classes is not None:
            x = F.one_hot(x, self.num_classes).float()
        x = x / self.scaling
        a_bar = self.alphas_bar.index_select(0, time_step).index_select(0, batch).unsqueeze(-1)
        pert = a_bar.sqrt() * x + (1 - a_bar).sqrt() * (torch.randn_like(x) if eps is None else eps)
        return pert if self.num_classes is None else (pert, x)

    def get_prev_from_recon(self, x_t, x_recon, t, batch, eps=None):
        """mu = coef_x0[t] x0_hat + coef_xt[t] x_t ; x_{t-1} = mu + std[t] eps, and exactly mu where t == 0.
        `eps` may be injected (parity tests); otherwise it is drawn with torch's generator like the reference."""
        if eps is None:
            eps = torch.randn_like(x_t)
        return _lib.pos_posterior(self.coef_x0, self.coef_xt, self.std, x_t, x_recon, eps, t, batch)

    def sample_init(self, shape):
        # the reference draws the prior on the host generator and moves it to the device
        if self.num_classes is None:
            return torch.randn(shape).to(self.betas.device)
        return torch.randn([shape, self.num_classes]).to(self.betas.device)


class GeneralCategoricalTransition(nn.Module):
    """Categorical diffusion with an absorbing/marginal prior: Q_t = beta_t 1 p0^T + (1 - beta_t) I."""

    def __init__(self, betas, num_classes, init_prob=None):
        super().__init__()
        self.eps = 1e-30
        self.num_classes = K = num_classes
        if init_prob is None or (isinstance(init_prob, str) and init_prob == 'uniform'):
            p0 = np.ones(K)
        elif isinstance(init_prob, str) and init_prob == 'absorb':
            p0 = np.full(K, 0.01); p0[0] = 1.
        elif isinstance(init_prob, str) and init_prob == 'tomask':
            p0 = np.full(K, 0.001); p0[-1] = 1.
        else:
            p0 = np.asarray(init_prob, dtype=np.float64)
        self.init_prob = p0 / p0.sum()
        self.betas = betas
        self.num_timesteps = T = len(betas)
        one_step = np.stack([self._get_transition_mat(t) for t in range(T)], axis=0)
        cumulative = np.empty_like(one_step)
        cumulative[0] = one_step[0]
        for t in range(1, T):
            cumulative[t] = cumulative[t - 1] @ one_step[t]
        self.q_mats = to_torch_const(cumulative)
        self.transpopse_q_onestep_mats = to_torch_const(np.ascontiguousarray(one_step.transpose(0, 2, 1)))

    def _get_transition_mat(self, t):
        b = self.betas[t]
        return b * np.tile(self.init_prob[None, :], (self.num_classes, 1)) + (1. - b) * np.eye(self.num_classes)

    # --- device entry points -------------------------------------------------------------
    def onehot_encode(self, v):
        return F.one_hot(v, self.num_classes).float()

    def q_v_posterior(self, log_v0, log_vt, t, batch, v0_prob):
        """log q(v_{t-1} | v_t, v_0); `log_v0` holds log-probabilities when `v0_prob` else is arg-maxed."""
        if not v0_prob:
            log_v0 = index_to_log_onehot(log_v0.argmax(dim=-1), self.num_classes, checked=False)
        if log_v0.ndim != 2:
            raise NotImplementedError('ndim not supported')
        return _lib.cat_posterior(self.q_mats, self.transpopse_q_onestep_mats, log_v0, log_vt, t, batch)

    def q_v_posterior_autograd(self, log_v0, log_vt, t, batch):
        """Same quantity as a chain of differentiable torch tensor ops on (rows, K <= 8) tensors: the loss tail of the
        training path (the gradient wrt `log_v0` = log-softmax of the predicted logits flows through here)."""
        tb = t[batch]
        tm1 = torch.clamp(t - 1, min=0)[batch]
        # (rows,K) x (rows,K,K) contractions written as broadcast-multiply-sum: a batched GEMM with K <= 8 per row is
        # three orders of magnitude off any library kernel's sweet spot
        f1 = (log_vt.exp().unsqueeze(-1) * self.transpopse_q_onestep_mats[tb]).sum(dim=1)
        f2 = (log_v0.exp().unsqueeze(-1) * self.q_mats[tm1]).sum(dim=1)
        out = torch.log(f1 + self.eps).clamp_min(-32.) + torch.log(f2 + self.eps).clamp_min(-32.)
        out = out - torch.logsumexp(out, dim=-1, keepdim=True)
        return torch.where((tb == 0).unsqueeze(-1), log_v0, out)

    def q_vt_pred(self, log_v0, t, batch):
        q = self.q_mats[t][batch]
        return torch.log((log_v0.exp().unsqueeze(-1) * q).sum(dim=-2) + self.eps).clamp_min(-32.)

    def q_vt_sample(self, log_v0, t, batch, u=None):
        logits = self.q_vt_pred(log_v0, t, batch)
        cls = _lib.gumbel_argmax(logits, torch.rand_like(logits) if u is None else u)
        return cls, index_to_log_onehot(cls, self.num_classes, checked=False)   # ids from an argmax over num_classes logits

    def add_noise(self, v, time_step, batch, u=None):
        if _FUSED_NOISE and v.is_cuda and v.dim() == 1 and v.numel() and 2 <= self.num_classes <= 8:
            # round 6: index_to_log_onehot, q_vt_pred, the Gumbel-max draw and the two one-hot encodings in one launch (csrc
            # cat_add_noise_kernel) instead of ~28 (rows x K) torch launches; the class-range assert keeps its (deferred) form
            check_class_range(v, self.num_classes)
            if u is None:
                u = torch.rand(v.shape[0], self.num_classes, dtype=torch.float32, device=v.device)
            return _lib.cat_add_noise(self.q_mats, v, time_step, batch, u, self.num_classes)
        log_v0 = index_to_log_onehot(v, self.num_classes)
        cls, log_vt = self.q_vt_sample(log_v0, time_step, batch, u)
        return F.one_hot(cls, self.num_classes).float(), log_vt, log_v0

    def compute_v_Lt(self, log_v_post_true, log_v_post_pred, log_v0, t, batch):
        """Per-row variational term: KL(q(v_{t-1}|v_t,v_0) || p) for t > 0, decoder NLL at t == 0."""
        if log_v_post_true.ndim != 2:
            raise NotImplementedError('ndim not supported')
        kl_v = categorical_kl(log_v_post_true, log_v_post_pred)
        nll_v = -log_categorical(log_v0, log_v_post_pred)
        mask = (t == 0).float()[batch]
        return mask * nll_v + (1 - mask) * kl_v

    def sample_init(self, n, u=None):
        """Draw from the prior by Gumbel-max on float64 logits (the reference's dtype at this point, models/transition.py:331-339):
        one library call (``mdx_prior_draw``).  u: optional (n,K) uniforms, float64 or float32."""
        dev = self.q_mats.device
        K = self.num_classes
        if u is None:
            u = torch.rand(n, K, dtype=torch.float64, device=dev)
        cls = torch.empty(n, dtype=torch.int64, device=dev)
        oh = torch.empty(n, K, dtype=torch.float32, device=dev)
        lg = torch.empty(n, K, dtype=torch.float32, device=dev)
        if n > 0:
            _lib.prior_draw(self.init_prob, u.to(dev), n, cls=cls, onehot=oh, log_onehot=lg)
        return cls, oh, lg

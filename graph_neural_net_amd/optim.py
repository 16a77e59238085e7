"""Optimizer / scheduler next to the hot path: Adam over the flat
parameter buffer as ONE fused HIP kernel right after the gradient all-reduce, and the reference's
ReduceLROnPlateau policy (models/trainers.py:92-104: Adam lr 1e-3, factor 0.5, patience 3, min_lr 1e-5)."""
import torch

from . import _lib


class FlatAdam:
    """torch.optim.Adam(amsgrad=False, weight_decay=0) semantics on one flat fp32 device buffer."""

    def __init__(self, params_flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if not params_flat.is_cuda:
            raise RuntimeError('FlatAdam: parameters must live on the GPU (no CPU path)')
        self.params = params_flat
        self.lr, self.betas, self.eps = lr, betas, eps
        self.exp_avg = torch.zeros_like(params_flat)
        self.exp_avg_sq = torch.zeros_like(params_flat)
        self.t = 0

    def step(self, grads_flat, grad_scale=1.0):
        self.t += 1
        _lib.call('fgnn_adam_step', _lib.ptr(self.params), _lib.ptr(grads_flat), _lib.ptr(self.exp_avg),
                  _lib.ptr(self.exp_avg_sq), self.params.numel(), float(self.lr), float(self.betas[0]),
                  float(self.betas[1]), float(self.eps), self.t, float(grad_scale), _lib.stream_ptr())
        # the replayable form keeps its own step count in device memory: keep the two counters equal, so that
        # eager and captured / device-side steps can be mixed freely
        if getattr(self, '_state', None) is not None:
            self._state[0:1].fill_(self.t)

    # -- graph-replayable form: step count and hyper-parameters live in device memory ---------------------
    def _dev_state(self):
        if getattr(self, '_hp', None) is None:
            self._hp = torch.zeros(5, dtype=torch.float64, device=self.params.device)
            self._state = torch.tensor([self.t, 0], dtype=torch.int32, device=self.params.device)
            self._hp_host = None
        return self._hp, self._state

    def sync_hyper_parameters(self, grad_scale=1.0):
        """Push lr / betas / eps (and, unless it is None, grad_scale) to the device copy if they changed (outside any
        graph capture).  grad_scale=None leaves hp[4] alone: the caller writes it on the device
        (set_grad_scale_reciprocal)."""
        hp, _ = self._dev_state()
        cur = (float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
               None if grad_scale is None else float(grad_scale))
        if cur != self._hp_host:
            n = 4 if grad_scale is None else 5
            hp[:n].copy_(torch.tensor(cur[:n], dtype=torch.float64))
            self._hp_host = cur

    def set_grad_scale_reciprocal(self, denom):
        """grad_scale <- 1 / denom for the device-side step, `denom` a 1-element device tensor (no host sync):
        the global loss normaliser that arrives with the all-reduced gradient buffer."""
        hp, _ = self._dev_state()
        torch.reciprocal(denom.to(torch.float64), out=hp[4:5])
        if self._hp_host is not None:
            self._hp_host = self._hp_host[:4] + (None,)

    def step_dev(self, grads_flat):
        """One update whose launch can be captured in a HIP graph and replayed (call sync_hyper_parameters first)."""
        hp, state = self._dev_state()
        self.t += 1
        _lib.call('fgnn_adam_step_dev', _lib.ptr(self.params), _lib.ptr(grads_flat), _lib.ptr(self.exp_avg),
                  _lib.ptr(self.exp_avg_sq), self.params.numel(), _lib.ptr(hp), _lib.ptr(state), _lib.stream_ptr())


class ReduceLROnPlateau:
    """mode='min', relative threshold 1e-4 -- the torch defaults the reference relies on."""

    def __init__(self, optimizer, factor=0.5, patience=3, min_lr=1e-5, threshold=1e-4):
        self.opt, self.factor, self.patience, self.min_lr, self.threshold = optimizer, factor, patience, min_lr, threshold
        self.best = float('inf')
        self.bad = 0

    def step(self, metric):
        if metric < self.best * (1.0 - self.threshold):
            self.best = metric
            self.bad = 0
        else:
            self.bad += 1
        if self.bad > self.patience:
            self.opt.lr = max(self.opt.lr * self.factor, self.min_lr)
            self.bad = 0
        return self.opt.lr

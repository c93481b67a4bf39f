"""Estimators with the sklearn protocol the reference uses for ``[trials, channels, time]`` arrays
(``clf.fit(X, y)`` / ``clf.predict(X)``, notebooks/svm_baseline.ipynb:307,316), trained the way the
reference trains FAST (src/fast/train/trainer.py): CrossEntropyLoss, AdamW(lr 5e-4), per-step cosine
multiplier with 10 warm-up epochs and the ``[global_step - 1]`` indexing quirk (trainer.py:15-27,38,48-54).

The step itself bypasses autograd: feature kernels -> conv stack -> FC head -> softmax-CE -> backward
kernels write straight into ONE flat gradient block, which data-parallel training all-reduces once
(RCCL over xGMI through ``torch.distributed``), followed by one AdamW step on the flat parameter.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .constants import BANDS_9, BANDS_40, CLASSES, ELECTRODES, ZONES
from .features import FeatureExtractor
from .nn import (FAST, EEGNet_Encoder, FeatureCNN, _FlatParamMixin, _dropout_seed, _stream, eegnet_backward,
                 eegnet_forward, fast_config, token_mean_predict)


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0):
    """Per-step multiplier table (src/fast/train/trainer.py:15-27): linear warm-up then half cosine."""
    total = epochs * niter_per_ep
    warm_iters = warmup_epochs * niter_per_ep
    warm = np.linspace(start_warmup_value, base_value, warm_iters) if warmup_epochs > 0 else np.array([])
    n = total - warm_iters
    sched = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * np.arange(n) / n))
    sched = np.concatenate((warm, sched))
    assert len(sched) == total
    return sched


def lr_multiplier(table, global_step):
    """LambdaLR of trainer.py:52 evaluates ``table[global_step - 1]``: step 0 reads the LAST entry."""
    return float(table[(global_step - 1) % len(table)])


# ----------------------------------------------------------------------------- flat-parameter models
class _FlatModel(nn.Module, _FlatParamMixin):
    """A model whose parameters are packed in one block: [conv stack | dense layers]."""

    def _dense_layers(self):
        raise NotImplementedError

    def _conv(self):
        raise NotImplementedError

    def _ordered_params(self):
        ps = list(self._conv()._ordered_params())
        for lin in self._dense_layers():
            ps += [lin.weight, lin.bias]
        return ps


class _FeatureModel(_FlatModel):
    def __init__(self, in_channels, feature_dim, n_classes, n_layers, act_dtype="f32"):
        super().__init__()
        self.net = FeatureCNN(in_channels, feature_dim, n_classes, n_layers, act_dtype)

    def _conv(self):
        return self.net.cnn

    def _dense_layers(self):
        return [self.net.fc]

    def conv_plan(self, x):
        return self.net.cnn._plan(x.shape[-1])

    def _conv_channels(self):
        return self.net.cnn.channels

    hidden_act = ()


class _FastModel(_FlatModel):
    def __init__(self, config):
        super().__init__()
        self.net = FAST(config)

    def _conv(self):
        return self.net.head

    def _dense_layers(self):
        return [self.net.input_layer[0], self.net.last_layer]

    def conv_plan(self, x):
        c = self.net.config
        return self.net.head._plan(c.window_len, c.slide_step)

    def _conv_channels(self):
        return len(self.net.config.electrodes)

    hidden_act = (True,)


class HotPath:
    """Autograd-free forward/backward of  conv stack -> [Linear+GELU] -> Linear -> token-mean CE.

    Gradients land in ``model.flat_grads()`` in the order of ``model.flat_params()``.
    """

    def __init__(self, model):
        self.model = model
        self._ws = {}

    def _buf(self, key, numel, device):
        t = self._ws.get(key)
        if t is None or t.numel() < numel or t.device != device:
            t = torch.empty(max(int(numel), 1), dtype=torch.float32, device=device)
            self._ws[key] = t
        return t

    def _layout(self):
        m = self.model
        flat = m.flat_params()
        n_conv = sum(p.numel() for p in m._conv()._ordered_params())
        offs, o = [], n_conv
        for lin in m._dense_layers():
            offs.append((o, o + lin.weight.numel(), lin.out_features, lin.in_features))
            o += lin.weight.numel() + lin.bias.numel()
        return flat, n_conv, offs

    def forward(self, x, labels=None, global_batch=None, want_grad=False):
        """x f32 CUDA [B, C, T] -> dict(loss, logits, pred); with labels and want_grad also fills the gradients."""
        self._check_inputs(x, labels)
        with torch.cuda.device(x.device):
            return self._forward(x, labels, global_batch, want_grad)

    def _check_inputs(self, x, labels):
        """The kernels take raw pointers: everything they assume about the operands is checked here."""
        if not isinstance(x, torch.Tensor):
            raise TypeError("x must be a torch tensor")
        if x.dtype == torch.bfloat16:
            if getattr(self.model._conv(), "act_dtype", "f32") != "bf16":
                raise TypeError("a bfloat16 feature map needs a model with act_dtype='bf16' (BASELINE config 3)")
        elif x.dtype != torch.float32:
            raise TypeError(f"x must be float32 (or bfloat16 for a bf16 model), got {x.dtype}")
        if x.dim() != 3:
            raise ValueError("x must be [batch, channels, time]")
        if not x.is_contiguous():
            raise ValueError("x must be contiguous (pass x.contiguous())")
        c_total = self.model._conv_channels()
        if x.shape[1] != c_total:
            raise ValueError(f"expected {c_total} channels, got {x.shape[1]}")
        if labels is not None:
            if not isinstance(labels, torch.Tensor) or labels.device != x.device:
                raise TypeError("labels must be a tensor on the device of x")
            if labels.dtype not in (torch.uint8, torch.int64):
                raise TypeError(f"labels must be uint8 or int64, got {labels.dtype}")
            if labels.dim() != 1 or labels.shape[0] != x.shape[0] or not labels.is_contiguous():
                raise ValueError("labels must be a contiguous [batch] vector")
        if not x.is_cuda:
            raise TypeError("x must be a CUDA tensor (the product has no CPU path)")

    def _forward(self, x, labels, global_batch, want_grad):
        m, L = self.model, _lib.lib()
        flat, n_conv, offs = self._layout()
        gflat = m.flat_grads() if want_grad else None
        plan = m.conv_plan(x)
        B, _, T = x.shape
        N = plan.windows(T)
        dev, st = x.device, _stream()
        fp, f4 = flat.data_ptr(), 4
        ws_bytes = int(L.isd_conv4_workspace_bytes(plan._h, B, T))
        ws = self._buf("conv", ws_bytes // 4, dev)
        if len(offs) == 1 and not m.hidden_act and L.isd_featcnn_supported(plan._h, B, T, offs[0][2]):
            # classifier on spec-S features: the whole step is one library call (three kernels + reductions)
            wo, bo, n_cls, _ = offs[0]
            logits = torch.empty((B, n_cls), dtype=torch.float32, device=dev)
            pred = torch.empty((B,), dtype=torch.int64, device=dev)
            loss = torch.empty((), dtype=torch.float32, device=dev) if labels is not None else None
            gp = gflat.data_ptr() if (want_grad and labels is not None) else 0
            step = L.isd_featcnn_step_bf16 if x.dtype == torch.bfloat16 else L.isd_featcnn_step
            _lib.check(step(plan._h, x.data_ptr(), fp, fp + wo * f4, fp + bo * f4,
                                          0 if labels is None else labels.data_ptr(),
                                          0 if labels is None else labels.element_size(), gp, gp + wo * f4 if gp else 0,
                                          logits.data_ptr(), pred.data_ptr(), 0 if loss is None else loss.data_ptr(),
                                          ws.data_ptr(), B, T, n_cls, 1.0 / float(global_batch or B), st))
            out = {"logits": logits, "pred": pred}
            if loss is not None:
                out["loss"] = loss
            return out
        if x.dtype != torch.float32:
            raise TypeError("a bfloat16 input is taken by the one-call classifier step only (isd_featcnn_step_bf16)")
        feat = self._buf("feat", B * N * plan.n_zones * plan.F, dev)
        _lib.check(L.isd_conv4_forward(plan._h, x.data_ptr(), fp, feat.data_ptr(), ws.data_ptr(), B, T, st))
        # dense layers
        acts, pres = [feat], []
        M, K = B * N, plan.n_zones * plan.F
        n_lin = len(offs)
        for i, (wo, bo, nout, nin) in enumerate(offs):
            assert nin == K
            act = 1 if i < n_lin - 1 else 0
            y = self._buf(f"y{i}", M * nout, dev)
            pre = self._buf(f"pre{i}", M * nout, dev) if act else None
            _lib.check(L.isd_linear_forward(acts[-1].data_ptr(), fp + wo * f4, fp + bo * f4, y.data_ptr(),
                                            0 if pre is None else pre.data_ptr(), M, K, nout, act, st))
            acts.append(y)
            pres.append(pre)
            K = nout
        n_cls = K
        logits = torch.empty((B, n_cls), dtype=torch.float32, device=dev)
        pred = torch.empty((B,), dtype=torch.int64, device=dev)
        out = {"logits": logits, "pred": pred}
        if labels is None:
            _lib.check(L.isd_softmax_ce(acts[-1].data_ptr(), 0, 0, logits.data_ptr(), 0, 0, pred.data_ptr(), B, N,
                                        n_cls, 1.0, 0, st))
            return out
        loss = torch.empty((), dtype=torch.float32, device=dev)
        dlt = self._buf("dlt", M * n_cls, dev)
        scale = 1.0 / float(global_batch or B)
        cws = self._buf("cews", int(L.isd_softmax_ce_workspace_bytes(B)) // 4 + 1, dev)
        _lib.check(L.isd_softmax_ce(acts[-1].data_ptr(), labels.data_ptr(), labels.element_size(), logits.data_ptr(),
                                    loss.data_ptr(), dlt.data_ptr() if want_grad else 0, pred.data_ptr(), B, N, n_cls,
                                    scale, cws.data_ptr(), st))
        out["loss"] = loss
        if not want_grad:
            return out
        gp = gflat.data_ptr()
        dy = dlt
        for i in range(n_lin - 1, -1, -1):
            wo, bo, nout, nin = offs[i]
            act = 1 if i < n_lin - 1 else 0
            dx = self._buf(f"dx{i}", M * nin, dev)
            lws = self._buf(f"lws{i}", int(L.isd_linear_workspace_bytes(M, nin, nout)) // 4 + 64, dev)
            _lib.check(L.isd_linear_backward(acts[i].data_ptr(), fp + wo * f4, dy.data_ptr(),
                                             0 if pres[i] is None else pres[i].data_ptr(), dx.data_ptr(),
                                             gp + wo * f4, gp + bo * f4, lws.data_ptr(), M, nin, nout, act, st))
            dy = dx
        _lib.check(L.isd_conv4_backward(plan._h, x.data_ptr(), fp, dy.data_ptr(), gp, ws.data_ptr(), B, T, st))
        return out


class _EEGNetFeatureNet(nn.Module):
    """``EEGNet_Encoder(in_channels, F)`` through the unmodified head contract, then ``Linear(F, n_classes)``:
    the "EEGNet-style depthwise CNN" classifier of BASELINE config 5."""

    def __init__(self, in_channels, feature_dim, n_classes, kernel_length, dropout):
        super().__init__()
        self.enc = EEGNet_Encoder(in_channels, feature_dim, kernel_length, dropout)
        self.fc = nn.Linear(feature_dim, n_classes)

    def token_logits(self, feats):
        from .nn import linear
        B = feats.shape[0]
        h = self.enc(feats.reshape(B, -1, feats.shape[-1]))
        return linear(h, self.fc.weight, self.fc.bias).unsqueeze(1)      # [B, 1, n_classes]

    def forward(self, feats):
        return self.token_logits(feats).squeeze(1)


class _EEGNetFeatureModel(_FlatModel):
    """Flat block = [EEGNet_Encoder parameters in state_dict order | fc.weight | fc.bias]."""

    def __init__(self, in_channels, feature_dim, n_classes, kernel_length=64, dropout=0.25):
        super().__init__()
        self.net = _EEGNetFeatureNet(in_channels, feature_dim, n_classes, kernel_length, dropout)

    def _conv(self):
        return self.net.enc

    def _dense_layers(self):
        return [self.net.fc]

    def _conv_channels(self):
        return self.net.enc.in_channels

    def make_path(self):
        return EEGNetPath(self)


class EEGNetPath(HotPath):
    """Autograd-free step of  EEGNet_Encoder -> Linear -> softmax-CE  (same contract as ``HotPath``).

    With ``want_grad`` the encoder runs in training mode (batch statistics, running buffers and
    ``num_batches_tracked`` updated, dropout with the module's own counter-based stream); otherwise in eval mode."""

    def _forward(self, x, labels, global_batch, want_grad):
        m, L = self.model, _lib.lib()
        enc, fc = m.net.enc, m.net.fc
        flat = m.flat_params()
        gflat = m.flat_grads() if want_grad else None
        bufs = enc.flat_buffers()
        B, _, T = x.shape
        plan = enc._plan_for(T)
        dev, st = x.device, _stream()
        n_enc = plan.n_params
        F_, n_cls = fc.in_features, fc.out_features
        fp, f4 = flat.data_ptr(), 4
        wo, bo = n_enc, n_enc + fc.weight.numel()
        training = bool(want_grad and labels is not None)
        bn = enc._bns()[0]
        p_drop = enc.p if training else 0.0
        enc._calls += 1
        seed = _dropout_seed(enc._stream_id, enc._calls)
        if training:
            for b in enc._bns():
                b.num_batches_tracked += 1
        ws = self._buf("eeg", int(L.isd_eegnet_workspace_bytes(plan._h, B)) // 4, dev)
        h = self._buf("h", B * F_, dev)
        # data parallel: BatchNorm over the global batch (the sum blocks are all-reduced between the stages)
        world = eegnet_forward(plan, x, flat, bufs, h, ws, training, 0.1 if bn.momentum is None else float(bn.momentum),
                               float(bn.eps), float(p_drop), seed, getattr(enc, "sync_bn", True))
        ytok = self._buf("ytok", B * n_cls, dev)
        _lib.check(L.isd_linear_forward(h.data_ptr(), fp + wo * f4, fp + bo * f4, ytok.data_ptr(), 0, B, F_, n_cls, 0, st))
        logits = torch.empty((B, n_cls), dtype=torch.float32, device=dev)
        pred = torch.empty((B,), dtype=torch.int64, device=dev)
        out = {"logits": logits, "pred": pred}
        if labels is None:
            _lib.check(L.isd_softmax_ce(ytok.data_ptr(), 0, 0, logits.data_ptr(), 0, 0, pred.data_ptr(), B, 1, n_cls,
                                        1.0, 0, st))
            return out
        loss = torch.empty((), dtype=torch.float32, device=dev)
        dlt = self._buf("dlt", B * n_cls, dev)
        cws = self._buf("cews", int(L.isd_softmax_ce_workspace_bytes(B)) // 4 + 1, dev)
        _lib.check(L.isd_softmax_ce(ytok.data_ptr(), labels.data_ptr(), labels.element_size(), logits.data_ptr(),
                                    loss.data_ptr(), dlt.data_ptr() if training else 0, pred.data_ptr(), B, 1, n_cls,
                                    1.0 / float(global_batch or B), cws.data_ptr(), st))
        out["loss"] = loss
        if not training:
            return out
        gp = gflat.data_ptr()
        dh = self._buf("dh", B * F_, dev)
        lws = self._buf("lws", int(L.isd_linear_workspace_bytes(B, F_, n_cls)) // 4 + 64, dev)
        _lib.check(L.isd_linear_backward(h.data_ptr(), fp + wo * f4, dlt.data_ptr(), 0, dh.data_ptr(), gp + wo * f4,
                                         gp + bo * f4, lws.data_ptr(), B, F_, n_cls, 0, st))
        eegnet_backward(plan, x, flat, dh, gflat, ws, float(p_drop), seed, world)
        return out


# ----------------------------------------------------------------------------- data parallel helper
class GradientBucket:
    """One flat gradient all-reduce per step.  With ``torch.distributed`` initialised
    (backend 'nccl' == RCCL on ROCm, 'gloo' on CPU) the bucket is summed over ranks; the loss kernel
    already divided by the GLOBAL batch, so the sum is the global-mean gradient."""

    def __init__(self, process_group=None, always_collective=False):
        import torch.distributed as dist
        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.group = process_group
        # issue the collectives even in a group of one rank (the RCCL start / stream-wait path on a single GPU)
        self.always_collective = bool(always_collective)

    @property
    def world_size(self):
        return self.dist.get_world_size(self.group) if self.dist else 1

    @property
    def rank(self):
        return self.dist.get_rank(self.group) if self.dist else 0

    def _sum(self, t):
        if t.is_cuda and self.dist.get_backend(self.group) == "gloo":
            # gloo (CPU rehearsals, single-GPU multi-process tests): stage through host memory
            h = t.detach().cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def _skip(self):
        return self.dist is None or (self.world_size == 1 and not self.always_collective)

    def all_reduce_(self, flat_grad, extra=None):
        if self._skip():
            return
        self._sum(flat_grad)
        if extra is not None:
            self._sum(extra)

    def all_reduce_start(self, flat_grad):
        """Start the bucket's all-reduce and return a handle for ``all_reduce_wait`` (None: nothing pending).
        RCCL runs it on its own stream behind the kernels already queued on the current one, so work launched
        between start and wait that does not touch the bucket (the next batch's feature extraction) overlaps it."""
        if self._skip():
            return None
        if flat_grad.is_cuda and self.dist.get_backend(self.group) == "gloo":
            self._sum(flat_grad)
            return None
        return self.dist.all_reduce(flat_grad, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)

    @staticmethod
    def all_reduce_wait(work):
        if work is not None:
            work.wait()                                   # the current stream waits for the collective; the host does not

    def broadcast_(self, flat_params, src=0):
        if not self._skip():
            if flat_params.is_cuda and self.dist.get_backend(self.group) == "gloo":
                h = flat_params.detach().cpu()
                self.dist.broadcast(h, src=src, group=self.group)
                flat_params.copy_(h)
            else:
                self.dist.broadcast(flat_params, src=src, group=self.group)

    def shard(self, n):
        """Contiguous shard [lo, hi) of n items owned by this rank (equal shards, remainder to the first ranks)."""
        w, r = self.world_size, self.rank
        base, rem = divmod(n, w)
        lo = r * base + min(r, rem)
        return lo, lo + base + (1 if r < rem else 0)


class Trainer:
    """AdamW(lr 5e-4, wd 1e-2, torch defaults) on the flat parameter + the reference's per-step schedule."""

    def __init__(self, model, lr=5e-4, weight_decay=1e-2, schedule=None, bucket=None):
        self.model = model
        self.path = model.make_path() if hasattr(model, "make_path") else HotPath(model)
        self.bucket = bucket or GradientBucket()
        flat = model.flat_params()
        gflat = model.flat_grads()
        self.flat, self.flat_grad = flat, gflat
        # AdamW runs on the flat block as ONE elementwise HIP kernel (csrc/adamw.hip, isd_adamw_step; torch's
        # operation order and defaults).  Until round 3 the block went to torch.optim.AdamW(fused=True) as 36 equal
        # views: its multi-tensor kernel gives every tensor its own few workgroups and took 29 us per step for 0.6 M
        # parameters (tools/adamw_probe.py); the flat kernel spreads them over the chip.
        if not flat.is_cuda:
            raise RuntimeError("Trainer needs the parameters on a HIP device (the AdamW step is a HIP kernel)")
        self.betas, self.eps, self.weight_decay = (0.9, 0.999), 1e-8, float(weight_decay)
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.base_lr = lr
        self.lr = float(lr)
        self.schedule = schedule
        self.global_step = 0
        self.bucket.broadcast_(flat)

    def step(self, x, labels, global_batch=None):
        """One optimisation step on a device-resident batch.  Returns the (local share of the) loss tensor."""
        out = self.step_begin(x, labels, global_batch)
        self.step_finish()
        return out

    def step_begin(self, x, labels, global_batch=None):
        """Forward + backward, then start the gradient all-reduce.  Work queued before ``step_finish`` that does
        not read the parameters or the gradient bucket runs under the collective."""
        if getattr(self, "_pending", None) is not None:
            raise RuntimeError("step_begin called twice without step_finish")
        out = self.path.forward(x, labels, global_batch=global_batch, want_grad=True)
        self._pending = (self.bucket.all_reduce_start(self.flat_grad),)
        return out

    def step_finish(self, after_wait=None):
        """Wait for the all-reduce started by ``step_begin`` and apply AdamW.  ``after_wait``: a ``torch.cuda.Event``
        recorded on the current stream between the wait and AdamW (bench.py: how long the stream stood still for the
        collective)."""
        if getattr(self, "_pending", None) is None:
            raise RuntimeError("step_finish without step_begin")
        self.bucket.all_reduce_wait(self._pending[0])
        self._pending = None
        if after_wait is not None:
            after_wait.record(torch.cuda.current_stream())
        if self.schedule is not None:
            self.lr = self.base_lr * lr_multiplier(self.schedule, self.global_step)
        self.global_step += 1
        _lib.check(_lib.lib().isd_adamw_step(
            self.flat.data_ptr(), self.flat_grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
            self.flat.numel(), self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.global_step,
            None, None, _stream()))

    def optimizer_state(self):
        """The moment estimates and the step count (``load_optimizer_state`` puts them back)."""
        return {"exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(), "step": self.global_step}

    def load_optimizer_state(self, state):
        self.exp_avg.copy_(state["exp_avg"])
        self.exp_avg_sq.copy_(state["exp_avg_sq"])
        self.global_step = int(state["step"])


# ----------------------------------------------------------------------------- estimators
def _to_device(X, device):
    if isinstance(X, torch.Tensor):
        return X.to(device=device, dtype=torch.float32).contiguous()
    return torch.as_tensor(np.ascontiguousarray(X, dtype=np.float32)).to(device)


class NotFittedError(ValueError, AttributeError):
    """Same base classes as ``sklearn.exceptions.NotFittedError`` (raised by predict before fit)."""


class _Estimator:
    """sklearn estimator protocol: constructor arguments are stored verbatim under their own names
    (``get_params`` / ``set_params`` / ``sklearn.base.clone`` work), ``fit`` starts from scratch unless
    ``warm_start=True``, fitted state lives in trailing-underscore attributes."""
    classes_ = np.arange(len(CLASSES))
    class_names_ = list(CLASSES)
    _param_names = ("max_epochs", "batch_size", "lr", "weight_decay", "warmup_epochs", "seed", "device", "shuffle",
                    "verbose", "warm_start")

    def __init__(self, max_epochs=200, batch_size=64, lr=5e-4, weight_decay=1e-2, warmup_epochs=10, seed=42,
                 device=None, shuffle=True, verbose=False, warm_start=False):
        self.max_epochs, self.batch_size, self.lr, self.weight_decay = max_epochs, batch_size, lr, weight_decay
        self.warmup_epochs, self.seed, self.shuffle, self.verbose = warmup_epochs, seed, shuffle, verbose
        self.device = device
        self.warm_start = warm_start
        self._reset()

    def _reset(self):
        self.model_ = None
        self.trainer_ = None
        self.history_ = []

    def get_params(self, deep=True):
        return {k: getattr(self, k) for k in self._param_names}

    def set_params(self, **params):
        for k, v in params.items():
            if k not in self._param_names:
                raise ValueError(f"invalid parameter {k!r} for {type(self).__name__}")
            setattr(self, k, v)
        return self

    def _fitted_model(self):
        if self.model_ is None:
            raise NotFittedError(f"this {type(self).__name__} instance is not fitted yet: call fit(X, y) (or load a "
                                 "state dict) before predict / decision_function")
        return self.model_

    def _device(self):
        if self.device is not None:
            return torch.device(self.device)
        if not torch.cuda.is_available():
            raise RuntimeError("isd_amd classifiers need an MI355X GPU: there is no CPU fallback")
        return torch.device("cuda", torch.cuda.current_device())

    def _build(self, X):
        raise NotImplementedError

    def _n_classes(self):
        raise NotImplementedError

    def _inputs(self, xb):
        return xb

    def _prepare_fit(self, X):
        """Hook: a per-fit transformation of the whole training set (identity here).  Returns (data, per-batch fn)."""
        return X, self._inputs

    def _ensure_model(self, X):
        if self.model_ is None:
            torch.manual_seed(self.seed)
            self.model_ = self._build(X).to(self._device())
        return self.model_

    def fit(self, X, y):
        """X [n, C, T] (ndarray or CUDA tensor), y int [n] in label order CLASSES.  Returns self."""
        yh = np.asarray(y.cpu() if isinstance(y, torch.Tensor) else y)
        n = len(X)
        if yh.ndim != 1 or n != yh.shape[0]:
            raise ValueError("X and y disagree on the number of trials")
        if yh.dtype.kind not in "iu":
            raise TypeError(f"y must hold integer class indices, got dtype {yh.dtype}")
        n_cls = self._n_classes()
        if n and (int(yh.min()) < 0 or int(yh.max()) >= n_cls):
            # torch's CE asserts here; the loss kernel indexes a register array with the label
            raise ValueError(f"labels must lie in [0, {n_cls}) (order = CLASSES); got "
                             f"[{int(yh.min())}, {int(yh.max())}]")
        dev = self._device()
        X = _to_device(X, dev)
        y = torch.as_tensor(yh).to(dev)
        if y.dtype not in (torch.uint8, torch.int64):
            y = y.long()
        if not self.warm_start:
            self._reset()                     # sklearn semantics: a second fit() does not continue the first
        model = self._ensure_model(X)
        X, batch_inputs = self._prepare_fit(X)
        bs = min(self.batch_size, n)
        iters = (n + bs - 1) // bs
        warm = min(self.warmup_epochs, max(self.max_epochs - 1, 0))
        table = cosine_scheduler(1, 0.1, self.max_epochs, iters, warmup_epochs=warm)
        self.trainer_ = Trainer(model, self.lr, self.weight_decay, schedule=table)
        gen = torch.Generator(device="cpu").manual_seed(self.seed)
        for ep in range(self.max_epochs):
            order = torch.randperm(n, generator=gen).to(dev) if self.shuffle else torch.arange(n, device=dev)
            tot, cnt = 0.0, 0
            for i in range(iters):
                idx = order[i * bs:(i + 1) * bs]
                xb, yb = batch_inputs(X[idx].contiguous()), y[idx].contiguous()
                out = self.trainer_.step(xb, yb)
                if self.verbose or ep == self.max_epochs - 1:
                    tot += float(out["loss"]) * len(idx)
                    cnt += len(idx)
            if cnt:
                self.history_.append(tot / cnt)
                if self.verbose:
                    print(f"epoch {ep + 1}/{self.max_epochs} loss {tot / cnt:.4f}")
        return self

    def decision_function(self, X, batch_size=4096):
        """Logits [n, n_classes] as a NumPy array."""
        model = self._fitted_model()
        dev = self._device()
        path = self.trainer_.path if self.trainer_ is not None else (
            model.make_path() if hasattr(model, "make_path") else HotPath(model))
        outs = []
        for i in range(0, len(X), batch_size):
            xb = self._inputs(_to_device(X[i:i + batch_size], dev))
            outs.append(path.forward(xb)["logits"].clone())
        return torch.cat(outs).cpu().numpy() if outs else np.zeros((0, len(CLASSES)), np.float32)

    def predict(self, X, batch_size=4096):
        """Class indices int64 [n] (argmax, ties -> lowest index; order = CLASSES)."""
        model = self._fitted_model()
        dev = self._device()
        path = self.trainer_.path if self.trainer_ is not None else (
            model.make_path() if hasattr(model, "make_path") else HotPath(model))
        outs = []
        for i in range(0, len(X), batch_size):
            xb = self._inputs(_to_device(X[i:i + batch_size], dev))
            outs.append(path.forward(xb)["pred"].clone())
        return torch.cat(outs).cpu().numpy() if outs else np.zeros((0,), np.int64)

    def score(self, X, y):
        return float((self.predict(X) == np.asarray(y)).mean())


class FilterbankCNNClassifier(_Estimator):
    """extract_features (Butterworth filterbank -> STFT -> log band power) -> 4-layer CNN -> Linear.

    BASELINE config 2: 9 bands, nperseg 64 / noverlap 32, Conv4Layers(nb*C, 32) + Linear(32, 5).
    ``n_layers=2`` with the 5-band set is config 1.
    """

    _param_names = _Estimator._param_names + ("fs", "bands", "order", "nperseg", "noverlap", "eps", "feature_dim",
                                              "n_classes", "n_layers", "fused", "precision", "cache_features")

    def __init__(self, fs=256.0, bands=BANDS_9, order=4, nperseg=64, noverlap=None, eps=1e-10, feature_dim=32,
                 n_classes=5, n_layers=4, fused=None, precision="fp32", cache_features=True, **kw):
        super().__init__(**kw)
        self.cache_features = cache_features   # fit(): extract the features of the training set once, not every epoch
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        self.precision = precision          # 'bf16' = BASELINE config 3 (bf16 activations/grads, fp32 accumulate)
        self.fs, self.bands, self.order = fs, bands, order
        self.nperseg, self.noverlap, self.eps = nperseg, noverlap, eps
        self.feature_dim, self.n_classes, self.n_layers, self.fused = feature_dim, n_classes, n_layers, fused

    def _reset(self):
        super()._reset()
        self.extractor_ = None

    def _extractor(self, T):
        if self.extractor_ is None or self.extractor_.stft.T != T:
            self.extractor_ = FeatureExtractor(T, self.fs, self.bands, self.order, self.nperseg, self.noverlap,
                                               self.eps)
        return self.extractor_

    def extract_features(self, trials):
        """trials f32 CUDA [B, C, T] -> [B, nb, C, J] (same as the module-level ``extract_features``).  With
        ``precision='bf16'`` the short-row fused extractor writes the map as bf16 -- the rounding the bf16 first layer
        applies to an fp32 map anyway (bit-identical results), half the bytes for the classifier to read."""
        fx = self._extractor(trials.shape[-1])
        st = fx.stft
        if (self.precision == "bf16" and self.fused is not False and fx.can_fuse and st.nperseg == 64 and st.noverlap == 32
                and trials.shape[-1] <= 1024 and (fx.n_bands * trials.shape[1]) % 8 == 0 and fx.n_frames <= 17):
            return fx(trials, fused=True, out_dtype=torch.bfloat16)
        return fx(trials, fused=self.fused)

    def _n_classes(self):
        return int(self.n_classes)

    def _build(self, X):
        fx = self._extractor(X.shape[-1])
        return _FeatureModel(fx.n_bands * X.shape[1], self.feature_dim, self.n_classes, self.n_layers,
                             "bf16" if self.precision == "bf16" else "f32")

    def _inputs(self, xb):
        f = self.extract_features(xb)
        return f.view(f.shape[0], -1, f.shape[-1])

    def _prepare_fit(self, X):
        """The features do not depend on the parameters: with more than one epoch they are extracted once (in
        batches, resident in HBM: 39 KB per trial at the default shape) and the epochs run on the cached tensor."""
        fx = self._extractor(X.shape[-1])
        n = X.shape[0]
        need = n * fx.n_bands * X.shape[1] * fx.n_frames * 4
        free = torch.cuda.mem_get_info(X.device)[0]
        if not self.cache_features or self.max_epochs < 2 or need > free // 2:
            return X, self._inputs
        feats = torch.empty((n, fx.n_bands * X.shape[1], fx.n_frames), dtype=torch.float32, device=X.device)
        for i in range(0, n, 4096):
            feats[i:i + 4096] = self._inputs(X[i:i + 4096].contiguous())
        return feats, (lambda fb: fb)


class FilterbankEEGNetClassifier(FilterbankCNNClassifier):
    """extract_features -> ``EEGNet_Encoder(nb*C, feature_dim)`` (fast.py:122-167, through the head contract
    fast.py:203-210) -> ``Linear(feature_dim, n_classes)``: BASELINE config 5, the high-resolution stress
    configuration (128 ch, 4 s @ 1024 Hz, 40 two-hertz bands, 1024-point STFT with hop 64 -> 65 frames)."""

    _param_names = FilterbankCNNClassifier._param_names + ("kernel_length", "dropout")

    def __init__(self, fs=1024.0, bands=BANDS_40, nperseg=1024, noverlap=960, kernel_length=64, dropout=0.25, **kw):
        super().__init__(fs=fs, bands=bands, nperseg=nperseg, noverlap=noverlap, **kw)
        if self.precision != "fp32":
            raise ValueError("the EEGNet head computes in fp32")
        self.kernel_length, self.dropout = kernel_length, dropout

    def _build(self, X):
        fx = self._extractor(X.shape[-1])
        return _EEGNetFeatureModel(fx.n_bands * X.shape[1], self.feature_dim, self.n_classes, self.kernel_length,
                                   self.dropout)


class FASTHeadClassifier(_Estimator):
    """The reference's FAST in ``forward_mode='train_head'`` on raw EEG: zone-wise Conv4Layers over sliding
    windows -> Linear(256, 32) + GELU -> Linear(32, 5) -> mean over windows (fast.py:273-278)."""

    _param_names = _Estimator._param_names + ("config",)

    def __init__(self, config=None, **kw):
        super().__init__(**kw)
        self.config = config

    def _n_classes(self):
        return int(self.config.n_classes) if self.config is not None else len(CLASSES)

    def _build(self, X):
        cfg = self.config or fast_config(ELECTRODES, ZONES, seq_len=int(X.shape[-1]))
        return _FastModel(cfg)

    def load_reference_state_dict(self, sd, X_like=None):
        """Load a reference FAST state_dict (keys ``head.encoders...``, ``input_layer.0...``, ``last_layer...``).
        The estimator then counts as fitted; ``fit`` continues from these weights only with ``warm_start=True``."""
        if self.model_ is None:
            self.model_ = _FastModel(self.config or fast_config()).to(self._device())
        missing, unexpected = self.model_.net.load_state_dict(sd, strict=False)
        if missing:
            raise KeyError(f"state_dict lacks hot-path parameters: {missing}")
        return unexpected


def smoke_classifier():
    """Tiny end-to-end step on cuda:0 (used by __graft_entry__.smoke)."""
    torch.manual_seed(0)
    x = torch.randn(8, 64, 512, device="cuda")
    y = torch.randint(0, 5, (8,), device="cuda")
    clf = FilterbankCNNClassifier(max_epochs=2, batch_size=8)
    clf.fit(x, y)
    p = clf.predict(x)
    assert p.shape == (8,) and np.isfinite(clf.history_[-1])

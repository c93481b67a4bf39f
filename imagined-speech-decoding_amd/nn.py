"""torch.nn-compatible front end of the HIP CNN / FC-head / loss kernels.

Mirrors the reference's module surface for the hot path
(src/fast/models/fast.py): ``Conv4Layers(channels, dim)`` with the head
contract ``Head_cls(n_zone_channels, feature_dim)`` -> ``encoder(x[B',Cz,T]) ->
[B', feature_dim]`` (fast.py:203-210), ``Head(head, electrodes, zone_dict,
feature_dim)``, and ``FAST(config)`` in ``forward_mode='train_head'``
(fast.py:273-278).  Parameter names and shapes equal the reference's
state_dict, so its checkpoints load.  All arithmetic runs in libisd_hip.so; the
autograd bridge passes ``tensor.data_ptr()`` through ctypes.  PyTorch holds the
parameters (one flat block, aliased by the named nn.Parameters) and runs the
optimizer.
"""
import ctypes as C
import math
import os

import torch
import torch.nn as nn

from . import _lib
from .constants import ELECTRODES, ZONES


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32c(t, name):
    if not (t.is_cuda and t.dtype == torch.float32):
        raise TypeError(f"{name} must be a float32 CUDA tensor (the product has no CPU path)")
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------- plans
class ConvStackPlan:
    """isd_conv4_plan wrapper: static geometry of a zone-wise Conv4Layers stack."""

    def __init__(self, c_total, zone_idx, feature_dim, n_layers, window_len, slide_step, act_dtype="f32"):
        self.c_total, self.zone_idx = int(c_total), [list(map(int, z)) for z in zone_idx]
        self.F, self.n_layers = int(feature_dim), int(n_layers)
        self.window_len, self.slide_step = int(window_len), int(slide_step)
        self._h = C.c_void_p()
        sizes = [len(z) for z in self.zone_idx]
        flat = [c for z in self.zone_idx for c in z]
        _lib.check(_lib.lib().isd_conv4_plan_create(C.byref(self._h), self.c_total, len(sizes), _lib.int_array(sizes),
                                                    _lib.int_array(flat), self.F, self.n_layers, self.window_len,
                                                    self.slide_step))
        self.n_params = int(_lib.lib().isd_conv4_param_count(self._h))
        if act_dtype not in ("f32", "bf16"):
            raise ValueError("act_dtype must be 'f32' or 'bf16'")
        self.act_dtype = act_dtype
        _lib.check(_lib.lib().isd_conv4_plan_set_activation_dtype(self._h, 1 if act_dtype == "bf16" else 0))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().isd_conv4_plan_destroy(h)
            except Exception:
                pass

    @property
    def n_zones(self):
        return len(self.zone_idx)

    def offset(self, zone, which):
        return int(_lib.lib().isd_conv4_param_offset(self._h, zone, which))

    def windows(self, T):
        n = _lib.lib().isd_conv4_windows(self._h, int(T))
        if n < 1:
            raise ValueError(f"T={T} is shorter than window_len={self.window_len}")
        return n

    def workspace(self, B, T, device):
        nbytes = int(_lib.lib().isd_conv4_workspace_bytes(self._h, int(B), int(T)))
        if nbytes < 0:
            raise ValueError("bad conv stack geometry")
        return torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=device)


# ----------------------------------------------------------------------------- autograd bridge
class _ConvStackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, flat, plan):
        x, flat = _f32c(x, "x"), _f32c(flat, "params")
        B, Ct, T = x.shape
        if Ct != plan.c_total:
            raise ValueError(f"expected {plan.c_total} channels, got {Ct}")
        N = plan.windows(T)
        feat = torch.empty((B * N, plan.n_zones, plan.F), dtype=torch.float32, device=x.device)
        ws = plan.workspace(B, T, x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().isd_conv4_forward(plan._h, x.data_ptr(), flat.data_ptr(), feat.data_ptr(),
                                                    ws.data_ptr(), B, T, _stream()))
        ctx.plan, ctx.ws = plan, ws
        ctx.save_for_backward(x, flat)
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        x, flat = ctx.saved_tensors
        B, _, T = x.shape
        dflat = torch.empty_like(flat)
        dfeat = _f32c(dfeat, "dfeat")
        dx = None
        with torch.cuda.device(x.device):
            if ctx.needs_input_grad[0]:             # attributions: layer-wise backward + the input-gradient kernel
                dx = torch.empty_like(x)
                _lib.check(_lib.lib().isd_conv4_backward_x(ctx.plan._h, x.data_ptr(), flat.data_ptr(),
                                                           dfeat.data_ptr(), dflat.data_ptr(), dx.data_ptr(),
                                                           ctx.ws.data_ptr(), B, T, _stream()))
            else:
                _lib.check(_lib.lib().isd_conv4_backward(ctx.plan._h, x.data_ptr(), flat.data_ptr(), dfeat.data_ptr(),
                                                         dflat.data_ptr(), ctx.ws.data_ptr(), B, T, _stream()))
        ctx.ws = None
        return dx, dflat, None


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, act):
        x, w = _f32c(x, "x"), _f32c(w, "weight")
        b = None if b is None else _f32c(b, "bias")
        K = x.shape[-1]
        N = w.shape[0]
        M = x.numel() // K
        y = torch.empty(tuple(x.shape[:-1]) + (N,), dtype=torch.float32, device=x.device)
        pre = torch.empty_like(y) if act else None
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().isd_linear_forward(x.data_ptr(), w.data_ptr(), 0 if b is None else b.data_ptr(),
                                                     y.data_ptr(), 0 if pre is None else pre.data_ptr(), M, K, N,
                                                     int(act), _stream()))
        ctx.act, ctx.has_bias = int(act), b is not None
        ctx.save_for_backward(x, w, pre)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, pre = ctx.saved_tensors
        dy = _f32c(dy, "dy")
        K, N = x.shape[-1], w.shape[0]
        M = x.numel() // K
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        if M == 0:                                   # no rows: nothing reaches the weights
            return dx, torch.zeros_like(w), torch.zeros(N, device=x.device) if ctx.has_bias else None, None
        dw = torch.empty_like(w)
        db = torch.empty(N, dtype=torch.float32, device=x.device) if ctx.has_bias else None
        ws = torch.empty(max(int(_lib.lib().isd_linear_workspace_bytes(M, K, N)) // 4, 1), dtype=torch.float32,
                         device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().isd_linear_backward(x.data_ptr(), w.data_ptr(), dy.data_ptr(),
                                                      0 if pre is None else pre.data_ptr(),
                                                      0 if dx is None else dx.data_ptr(), dw.data_ptr(),
                                                      0 if db is None else db.data_ptr(), ws.data_ptr(), M, K, N,
                                                      ctx.act, _stream()))
        return dx, dw, db, None


_unit_grads = {}


def unit_grad(device):
    """A cached scalar 1.0 on ``device`` to start backward passes from: ``loss.backward(unit_grad(loss.device))``.
    ``loss.backward()`` fills a fresh ones tensor every time and ``_SoftmaxCEFn.backward`` multiplied its stored
    gradient by it -- two 5-us kernels per step to multiply by one; started from THIS tensor (recognised by its
    address) the backward pass does neither."""
    key = str(torch.device(device))
    t = _unit_grads.get(key)
    if t is None:
        t = _unit_grads[key] = torch.ones((), dtype=torch.float32, device=device)
    return t


class _SoftmaxCEFn(torch.autograd.Function):
    """loss = CrossEntropyLoss()(logits_tok.mean(1), y) * (B / global_batch)."""

    @staticmethod
    def forward(ctx, logits_tok, labels, grad_scale):
        lt = _f32c(logits_tok, "logits")
        B, n_tok, n_cls = lt.shape
        labels = labels.contiguous()
        if labels.dtype not in (torch.uint8, torch.int64):
            labels = labels.long()
        loss = torch.empty((), dtype=torch.float32, device=lt.device)
        dlt = torch.empty_like(lt)
        ws = torch.empty(int(_lib.lib().isd_softmax_ce_workspace_bytes(B)) // 4 + 1, dtype=torch.float32,
                         device=lt.device)
        with torch.cuda.device(lt.device):
            _lib.check(_lib.lib().isd_softmax_ce(lt.data_ptr(), labels.data_ptr(), labels.element_size(), 0,
                                                 loss.data_ptr(), dlt.data_ptr(), 0, B, n_tok, n_cls,
                                                 float(grad_scale), ws.data_ptr(), _stream()))
        ctx.save_for_backward(dlt)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dlt,) = ctx.saved_tensors
        u = _unit_grads.get(str(g.device))
        if u is not None and g.data_ptr() == u.data_ptr():       # the cached unit gradient: nothing to scale
            return dlt, None, None
        return dlt * g, None, None


class EEGNetPlan:
    def __init__(self, in_channels, feature_dim, kernel_length, T, cvblock=False):
        self._h = C.c_void_p()
        if cvblock:
            _lib.check(_lib.lib().isd_cvblock_plan_create(C.byref(self._h), int(in_channels), int(feature_dim), int(T)))
            self.flat_dim = int(_lib.lib().isd_cvblock_flat_dim(self._h))
        else:
            _lib.check(_lib.lib().isd_eegnet_plan_create(C.byref(self._h), int(in_channels), int(feature_dim),
                                                         int(kernel_length), int(T)))
        self.n_params = int(_lib.lib().isd_eegnet_param_count(self._h))
        self.F = int(feature_dim)
        self._seed_dev = None

    def set_seed_counter(self, counter):
        """``counter``: int64 device tensor (or None) mixed into every pass's dropout seed (graph replay)."""
        self._seed_dev = counter                          # keeps the tensor alive
        _lib.check(_lib.lib().isd_eegnet_plan_set_seed_counter(self._h, 0 if counter is None else counter.data_ptr()))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().isd_eegnet_plan_destroy(h)
            except Exception:
                pass


def _bn_sync_world(sync, training):
    """(torch.distributed, world size) when BatchNorm statistics are to be synchronised, else (None, 1)."""
    import torch.distributed as dist
    if sync and training and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist, dist.get_world_size()
    return None, 1


def _all_reduce_block(dist, ws, byte_off, n_words, integer=False):
    """SUM over ranks of a block of batch sums inside the workspace -- fp64 values, or (``integer``) the int64 words of
    exact accumulators, which add exactly (RCCL on the current stream; gloo, used by rehearsals with ranks sharing a
    card, stages through the host)."""
    blk = ws.view(torch.uint8)[byte_off:byte_off + 8 * n_words].view(torch.int64 if integer else torch.float64)
    if blk.is_cuda and dist.get_backend() == "gloo":
        h = blk.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        blk.copy_(h)
    else:
        dist.all_reduce(blk, op=dist.ReduceOp.SUM)


def eegnet_forward(plan, x, flat, bufs, out, ws, training, momentum, eps, dropout_p, seed, sync=True):
    """EEGNet_Encoder / CVBlock forward through the C ABI.  With torch.distributed initialised (world > 1) and
    ``sync`` the three BatchNorm layers use the statistics of the GLOBAL batch: the pass runs in four
    stages and the fp64 sum block of each stage is all-reduced in between.  Returns the world size that was used
    (pass it to ``eegnet_backward``)."""
    L, B, st = _lib.lib(), x.shape[0], _stream()
    dist, world = _bn_sync_world(sync, int(training) == 1)      # training: 0 eval, 1 batch statistics, 2 eval + keep
    args = (x.data_ptr(), flat.data_ptr(), bufs.data_ptr(), out.data_ptr(), ws.data_ptr(), B, int(training),
            float(momentum), float(eps), float(dropout_p), int(seed))
    if world == 1:
        _lib.check(L.isd_eegnet_forward(plan._h, *args, st))
        return 1
    off, n = C.c_int64(), C.c_int64()
    for stage in range(4):
        _lib.check(L.isd_eegnet_forward_stage(plan._h, stage, *args, world, st))
        if stage < 3:
            _lib.check(L.isd_eegnet_sync_block(plan._h, B, 0, stage, C.byref(off), C.byref(n)))
            _all_reduce_block(dist, ws, off.value, n.value, bool(L.isd_eegnet_sync_block_kind(0, stage)))
    return world


def eegnet_backward(plan, x, flat, dout, dflat, ws, dropout_p, seed, world=1):
    """Parameter gradients of the forward above; ``world`` > 1: the BatchNorm backward sums are all-reduced between the
    stages and the gradients assembled from global sums arrive pre-divided, so that the gradient all-reduce (SUM) of
    data-parallel training yields the single-device gradient."""
    L, B, st = _lib.lib(), x.shape[0], _stream()
    args = (x.data_ptr(), flat.data_ptr(), dout.data_ptr(), dflat.data_ptr(), ws.data_ptr(), B, float(dropout_p),
            int(seed))
    if world == 1:
        _lib.check(L.isd_eegnet_backward(plan._h, *args, st))
        return
    import torch.distributed as dist
    off, n = C.c_int64(), C.c_int64()
    for stage in range(4):
        _lib.check(L.isd_eegnet_backward_stage(plan._h, stage, *args, world, st))
        if stage < 3:
            _lib.check(L.isd_eegnet_sync_block(plan._h, B, 1, stage, C.byref(off), C.byref(n)))
            _all_reduce_block(dist, ws, off.value, n.value, bool(L.isd_eegnet_sync_block_kind(1, stage)))


class _EEGNetFn(torch.autograd.Function):
    """EEGNet_Encoder / CVBlock.  Differentiable w.r.t. the parameters and -- single device -- the input trials, in
    train mode (batch statistics: BatchNorm's mean / variance paths are part of the input gradient) and in eval mode
    (running statistics: what attribution methods differentiate; the forward then keeps its activations)."""

    @staticmethod
    def forward(ctx, x, flat, bufs, plan, training, momentum, eps, dropout_p, seed, sync):
        x, flat = _f32c(x, "x"), _f32c(flat, "params")
        B = x.shape[0]
        out = torch.empty((B, plan.F), dtype=torch.float32, device=x.device)
        ws = torch.empty(max(int(_lib.lib().isd_eegnet_workspace_bytes(plan._h, B)) // 4, 1), dtype=torch.float32,
                         device=x.device)
        want_bwd = any(ctx.needs_input_grad[:2])
        mode = 1 if training else (2 if want_bwd else 0)     # 2: eval-mode statistics, activations kept for backward
        with torch.cuda.device(x.device):
            world = eegnet_forward(plan, x, flat, bufs, out, ws, mode, momentum, eps, dropout_p, seed, sync)
        ctx.plan, ctx.ws, ctx.dp, ctx.seed, ctx.mode, ctx.world = plan, ws, float(dropout_p), int(seed), mode, world
        ctx.save_for_backward(x, flat)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, flat = ctx.saved_tensors
        dflat = torch.empty_like(flat)
        dout = _f32c(dout, "dout")
        dx = None
        with torch.cuda.device(x.device):
            if ctx.needs_input_grad[0] or ctx.mode == 2:
                if ctx.world != 1:
                    raise NotImplementedError("the input gradient of the BatchNorm heads is single-device")
                dx = torch.empty_like(x)
                _lib.check(_lib.lib().isd_eegnet_backward_x(ctx.plan._h, x.data_ptr(), flat.data_ptr(), dout.data_ptr(),
                                                            dflat.data_ptr(), dx.data_ptr(), ctx.ws.data_ptr(), x.shape[0],
                                                            ctx.mode, ctx.dp, ctx.seed, _stream()))
                if not ctx.needs_input_grad[0]:
                    dx = None
            else:
                eegnet_backward(ctx.plan, x, flat, dout, dflat, ctx.ws, ctx.dp, ctx.seed, ctx.world)
        ctx.ws = None
        return dx, dflat, None, None, None, None, None, None, None, None


class PaperHeadPlan:
    def __init__(self, in_channels, feature_dim, T):
        self._h = C.c_void_p()
        _lib.check(_lib.lib().isd_paperhead_plan_create(C.byref(self._h), int(in_channels), int(feature_dim), int(T)))
        self.n_params = int(_lib.lib().isd_paperhead_param_count(self._h))
        self.F = int(feature_dim)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().isd_paperhead_plan_destroy(h)
            except Exception:
                pass


class _PaperHeadFn(torch.autograd.Function):
    """HeadConv_Paper_Version: parameter gradients and the input gradient, after a train-mode (batch statistics) or an
    eval-mode (running statistics) forward.  With torch.distributed initialised (world > 1) the four BatchNorm layers
    use the statistics of the GLOBAL batch: the pass runs in stages and each layer's fp64 sum block is all-reduced in
    between, as for EEGNet_Encoder / CVBlock."""

    @staticmethod
    def forward(ctx, x, flat, bufs, plan, training, momentum, eps, sync):
        x, flat = _f32c(x, "x"), _f32c(flat, "params")
        B = x.shape[0]
        L = _lib.lib()
        out = torch.empty((B, plan.F), dtype=torch.float32, device=x.device)
        ws = torch.empty(max(int(L.isd_paperhead_workspace_bytes(plan._h, B)) // 4, 1), dtype=torch.float32,
                         device=x.device)
        dist, world = _bn_sync_world(sync, training)
        args = (x.data_ptr(), flat.data_ptr(), bufs.data_ptr(), out.data_ptr(), ws.data_ptr(), B, int(training),
                float(momentum), float(eps))
        with torch.cuda.device(x.device):
            if world == 1:
                _lib.check(L.isd_paperhead_forward(plan._h, *args, _stream()))
            else:
                off, n = C.c_int64(), C.c_int64()
                for stage in range(5):
                    _lib.check(L.isd_paperhead_forward_stage(plan._h, stage, *args, world, _stream()))
                    if stage < 4:
                        _lib.check(L.isd_paperhead_sync_block(plan._h, B, 0, stage, C.byref(off), C.byref(n)))
                        _all_reduce_block(dist, ws, off.value, n.value, bool(L.isd_paperhead_sync_block_kind(0, stage)))
        ctx.plan, ctx.ws, ctx.training, ctx.world = plan, ws, training, world
        ctx.save_for_backward(x, flat)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, flat = ctx.saved_tensors
        dflat = torch.empty_like(flat)
        dout = _f32c(dout, "dout")
        dx = None
        L, B = _lib.lib(), x.shape[0]
        with torch.cuda.device(x.device):
            if ctx.needs_input_grad[0] or not ctx.training:
                if ctx.world != 1:
                    raise NotImplementedError("the input gradient of the BatchNorm heads is single-device")
                dx = torch.empty_like(x)
                _lib.check(L.isd_paperhead_backward_x(ctx.plan._h, x.data_ptr(), flat.data_ptr(), dout.data_ptr(),
                                                      dflat.data_ptr(), dx.data_ptr(), ctx.ws.data_ptr(), B,
                                                      int(ctx.training), _stream()))
                if not ctx.needs_input_grad[0]:
                    dx = None
            elif ctx.world == 1:
                _lib.check(L.isd_paperhead_backward(ctx.plan._h, x.data_ptr(), flat.data_ptr(), dout.data_ptr(),
                                                    dflat.data_ptr(), ctx.ws.data_ptr(), B, _stream()))
            else:
                import torch.distributed as dist
                off, n = C.c_int64(), C.c_int64()
                for stage in range(5):
                    _lib.check(L.isd_paperhead_backward_stage(ctx.plan._h, stage, x.data_ptr(), flat.data_ptr(),
                                                              dout.data_ptr(), dflat.data_ptr(), ctx.ws.data_ptr(), B,
                                                              ctx.world, _stream()))
                    if stage < 4:
                        _lib.check(L.isd_paperhead_sync_block(ctx.plan._h, B, 1, stage, C.byref(off), C.byref(n)))
                        _all_reduce_block(dist, ctx.ws, off.value, n.value, bool(L.isd_paperhead_sync_block_kind(1, stage)))
        ctx.ws = None
        return dx, dflat, None, None, None, None, None, None


class _BNZonesFn(torch.autograd.Function):
    """All zones of a BatchNorm head (EEGNet_Encoder / CVBlock / HeadConv_Paper_Version instances, fast.py:203-210) in
    zone-batched launches: the per-zone C-ABI calls are recorded and launch i of every zone goes out as ONE kernel
    (include/isd_hip.h, ``isd_zone_batch_*``).  Eight chains of ~20 short kernels per direction are launch-bound
    however they are queued.  Parameter gradients only, single device; ``Head`` keeps the per-zone path for the rest
    (input gradients, eval-mode gradients, synchronised BatchNorm across ranks)."""

    @staticmethod
    def forward(ctx, xw, idxs, calls, *thetas):
        L, B, dev = _lib.lib(), xw.shape[0], xw.device
        xs = [_f32c(xw.index_select(1, idx), "x") for idx in idxs]
        thetas = [_f32c(t, "params") for t in thetas]
        kind = calls[0][0]
        ws_bytes = L.isd_eegnet_workspace_bytes if kind == "eeg" else L.isd_paperhead_workspace_bytes
        outs = [torch.empty((B, c[2].F), dtype=torch.float32, device=dev) for c in calls]
        wss = [torch.empty(max(int(ws_bytes(c[2]._h, B)) // 4, 1), dtype=torch.float32, device=dev) for c in calls]
        with torch.cuda.device(dev):
            st = _stream()
            _lib.check(L.isd_zone_batch_begin())
            try:
                for z, (c, x, th, out, ws) in enumerate(zip(calls, xs, thetas, outs, wss)):
                    if z:
                        _lib.check(L.isd_zone_batch_next())
                    if kind == "eeg":
                        _, bufs, plan, training, momentum, eps, p, seed = c
                        _lib.check(L.isd_eegnet_forward(plan._h, x.data_ptr(), th.data_ptr(), bufs.data_ptr(),
                                                        out.data_ptr(), ws.data_ptr(), B, int(bool(training)),
                                                        float(momentum), float(eps), float(p), int(seed), st))
                    else:
                        _, bufs, plan, training, momentum, eps = c
                        _lib.check(L.isd_paperhead_forward(plan._h, x.data_ptr(), th.data_ptr(), bufs.data_ptr(),
                                                           out.data_ptr(), ws.data_ptr(), B, int(bool(training)),
                                                           float(momentum), float(eps), st))
                _lib.check(L.isd_zone_batch_launch(st))
            except Exception:
                L.isd_zone_batch_abort()
                raise
        ctx.calls, ctx.wss, ctx.n = calls, wss, len(calls)
        ctx.save_for_backward(*xs, *thetas)
        return torch.stack(outs, dim=1)

    @staticmethod
    def backward(ctx, dout):
        L, n = _lib.lib(), ctx.n
        xs, thetas = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        B = xs[0].shape[0]
        dz = _f32c(dout.permute(1, 0, 2), "dout")                  # [Z][B'][F]: one contiguous block per zone
        dflats = [torch.empty_like(t) for t in thetas]
        kind = ctx.calls[0][0]
        with torch.cuda.device(dz.device):
            st = _stream()
            _lib.check(L.isd_zone_batch_begin())
            try:
                for z, (c, x, th, dfl, ws) in enumerate(zip(ctx.calls, xs, thetas, dflats, ctx.wss)):
                    if z:
                        _lib.check(L.isd_zone_batch_next())
                    if kind == "eeg":
                        _lib.check(L.isd_eegnet_backward(c[2]._h, x.data_ptr(), th.data_ptr(), dz[z].data_ptr(),
                                                         dfl.data_ptr(), ws.data_ptr(), B, float(c[6]), int(c[7]), st))
                    else:
                        _lib.check(L.isd_paperhead_backward(c[2]._h, x.data_ptr(), th.data_ptr(), dz[z].data_ptr(),
                                                            dfl.data_ptr(), ws.data_ptr(), B, st))
                _lib.check(L.isd_zone_batch_launch(st))
            except Exception:
                L.isd_zone_batch_abort()
                raise
        ctx.wss = None
        return (None, None, None) + tuple(dflats)


class _LinearResFn(torch.autograd.Function):
    """y = x @ w.T + b + res (residual fused in the epilogue)."""

    @staticmethod
    def forward(ctx, x, w, b, res):
        x, w, b, res = _f32c(x, "x"), _f32c(w, "weight"), _f32c(b, "bias"), _f32c(res, "res")
        K, N = x.shape[-1], w.shape[0]
        M = x.numel() // K
        y = torch.empty(tuple(x.shape[:-1]) + (N,), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().isd_linear_residual_forward(x.data_ptr(), w.data_ptr(), b.data_ptr(), res.data_ptr(),
                                                              y.data_ptr(), M, K, N, _stream()))
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _f32c(dy, "dy")
        K, N = x.shape[-1], w.shape[0]
        M = x.numel() // K
        dx, dw = torch.empty_like(x), torch.empty_like(w)
        db = torch.empty(N, dtype=torch.float32, device=x.device)
        ws = torch.empty(max(int(_lib.lib().isd_linear_workspace_bytes(M, K, N)) // 4, 1), dtype=torch.float32,
                         device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().isd_linear_backward(x.data_ptr(), w.data_ptr(), dy.data_ptr(), 0, dx.data_ptr(),
                                                      dw.data_ptr(), db.data_ptr(), ws.data_ptr(), M, K, N, 0,
                                                      _stream()))
        return dx, dw, db, dy


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        x, w, b = _f32c(x, "x"), _f32c(w, "weight"), _f32c(b, "bias")
        D = x.shape[-1]
        M = x.numel() // D
        y = torch.empty_like(x)
        stats = torch.empty((M, 2), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().isd_layernorm_forward(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(),
                                                        stats.data_ptr(), M, D, float(eps), _stream()))
        ctx.save_for_backward(x, w, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, stats = ctx.saved_tensors
        dy = _f32c(dy, "dy")
        D = x.shape[-1]
        M = x.numel() // D
        dx, dw, db = torch.empty_like(x), torch.empty_like(w), torch.empty_like(w)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().isd_layernorm_backward(x.data_ptr(), w.data_ptr(), dy.data_ptr(), stats.data_ptr(),
                                                         dx.data_ptr(), dw.data_ptr(), db.data_ptr(), M, D, _stream()))
        return dx, dw, db, None


class _AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, num_heads, dropout_p, seed):
        qkv = _f32c(qkv, "qkv")
        B, S, D3 = qkv.shape
        D = D3 // 3
        out = torch.empty((B, S, D), dtype=torch.float32, device=qkv.device)
        probs = torch.empty((B, num_heads, S, S), dtype=torch.float32, device=qkv.device)
        with torch.cuda.device(qkv.device):
            _lib.check(_lib.lib().isd_attention_forward(qkv.data_ptr(), out.data_ptr(), probs.data_ptr(), B, S,
                                                        num_heads, D // num_heads, float(dropout_p), int(seed),
                                                        _stream()))
        ctx.cfg = (num_heads, float(dropout_p), int(seed))
        ctx.save_for_backward(qkv, probs)
        return out

    @staticmethod
    def backward(ctx, dctx):
        qkv, probs = ctx.saved_tensors
        H, p, seed = ctx.cfg
        dctx = _f32c(dctx, "dctx")
        B, S, D3 = qkv.shape
        dqkv = torch.empty_like(qkv)
        with torch.cuda.device(qkv.device):
            _lib.check(_lib.lib().isd_attention_backward(qkv.data_ptr(), probs.data_ptr(), dctx.data_ptr(),
                                                         dqkv.data_ptr(), B, S, H, (D3 // 3) // H, p, seed, _stream()))
        return dqkv, None, None, None


class _EmbedFn(torch.autograd.Function):
    """tokens = cat(cls, x) + pos  (fast.py:263-265)."""

    @staticmethod
    def forward(ctx, x, cls, pos):
        x, cls, pos = _f32c(x, "x"), _f32c(cls, "cls_token"), _f32c(pos, "pos_embedding")
        B, N, D = x.shape
        tok = torch.empty((B, N + 1, D), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().isd_embed_forward(x.data_ptr(), cls.data_ptr(), pos.data_ptr(), tok.data_ptr(), B, N, D,
                                                    _stream()))
        ctx.shape = (B, N, D, tuple(cls.shape), tuple(pos.shape))
        return tok

    @staticmethod
    def backward(ctx, dtok):
        B, N, D, cshape, pshape = ctx.shape
        dtok = _f32c(dtok, "dtok")
        dx = torch.empty((B, N, D), dtype=torch.float32, device=dtok.device)
        dcls = torch.empty(D, dtype=torch.float32, device=dtok.device)
        dpos = torch.zeros(pshape, dtype=torch.float32, device=dtok.device)        # rows past N+1 get no gradient
        with torch.cuda.device(dtok.device):
            _lib.check(_lib.lib().isd_embed_backward(dtok.data_ptr(), dx.data_ptr(), dcls.data_ptr(), dpos.data_ptr(), B,
                                                     N, D, _stream()))
        return dx, dcls.view(cshape), dpos


class _TailFusedFn(torch.autograd.Function):
    """cls/pos embedding -> every AttentionBlock -> cls dropout -> last_layer in one launch per direction
    (csrc/tailfused.hip).  The parameters are read from ``flat`` (the module's packed tail block); ``params`` are the
    same tensors as autograd inputs, so that their gradients -- views of one flat gradient block -- reach them."""

    @staticmethod
    def forward(ctx, tokin, flat, cfg, *params):
        tokin = _f32c(tokin, "tokens")
        B, N, D = tokin.shape
        n_pos, H, L, hidden, n_cls, p_attn, p_mlp, p_cls, seed, train, seed_dev = cfg
        L_ = _lib.lib()
        logits = torch.empty((B, n_cls), dtype=torch.float32, device=tokin.device)
        save = xfinal = None
        if train:
            save = torch.empty(int(L_.isd_tail_fused_save_floats(B, N + 1, D, L)), dtype=torch.float32,
                               device=tokin.device)
            xfinal = torch.empty((B, D), dtype=torch.float32, device=tokin.device)
        with torch.cuda.device(tokin.device):
            _lib.check(L_.isd_tail_fused_forward(flat.data_ptr(), tokin.data_ptr(), logits.data_ptr(),
                                                 save.data_ptr() if train else 0, xfinal.data_ptr() if train else 0,
                                                 B, N, n_pos, D, H, L, hidden, n_cls, p_attn, p_mlp, p_cls, seed,
                                                 0 if seed_dev is None else seed_dev.data_ptr(), _stream()))
        ctx.cfg, ctx.dims = cfg, (B, N, D)
        ctx.shapes = [tuple(p.shape) for p in params]
        if train:
            ctx.save_for_backward(flat, save, xfinal)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        flat, save, xfinal = ctx.saved_tensors
        n_pos, H, L, hidden, n_cls, p_attn, p_mlp, p_cls, seed, _, seed_dev = ctx.cfg
        B, N, D = ctx.dims
        dlogits = _f32c(dlogits, "dlogits")
        L_ = _lib.lib()
        dtok = torch.empty((B, N, D), dtype=torch.float32, device=flat.device)
        dflat = torch.empty_like(flat)
        ws = torch.empty(int(L_.isd_tail_fused_workspace_floats(B, N, n_pos, D, L, n_cls)), dtype=torch.float32,
                         device=flat.device)
        with torch.cuda.device(flat.device):
            _lib.check(L_.isd_tail_fused_backward(flat.data_ptr(), save.data_ptr(), xfinal.data_ptr(),
                                                  dlogits.data_ptr(), dtok.data_ptr(), dflat.data_ptr(), ws.data_ptr(),
                                                  B, N, n_pos, D, H, L, hidden, n_cls, p_attn, p_mlp, p_cls, seed,
                                                  0 if seed_dev is None else seed_dev.data_ptr(), _stream()))
        grads, off = [], 0
        for shp in ctx.shapes:
            n = math.prod(shp)
            grads.append(dflat[off:off + n].view(shp))
            off += n
        return (dtok, None, None, *grads)


def linear(x, weight, bias=None, act=False):
    return _LinearFn.apply(x, weight, bias, act)


def linear_residual(x, weight, bias, res):
    return _LinearResFn.apply(x, weight, bias, res)


def layer_norm(x, weight, bias, eps=1e-5):
    return _LayerNormFn.apply(x, weight, bias, eps)


def token_mean_cross_entropy(logits_tok, labels, global_batch=None):
    """CE of the token-mean logits (fast.py:277 + trainer.py:59).  logits_tok [B, n_tok, n_cls]."""
    if logits_tok.dim() == 2:
        logits_tok = logits_tok.unsqueeze(1)
    B = logits_tok.shape[0]
    return _SoftmaxCEFn.apply(logits_tok, labels, 1.0 / float(global_batch or B))


def token_mean_predict(logits_tok):
    """-> (mean logits [B, n_cls], argmax int64 [B]); ties -> lowest index (trainer.py:89)."""
    if logits_tok.dim() == 2:
        logits_tok = logits_tok.unsqueeze(1)
    lt = _f32c(logits_tok.detach(), "logits")
    B, n_tok, n_cls = lt.shape
    lm = torch.empty((B, n_cls), dtype=torch.float32, device=lt.device)
    pred = torch.empty((B,), dtype=torch.int64, device=lt.device)
    with torch.cuda.device(lt.device):
        _lib.check(_lib.lib().isd_softmax_ce(lt.data_ptr(), 0, 0, lm.data_ptr(), 0, 0, pred.data_ptr(), B, n_tok,
                                             n_cls, 1.0, 0, _stream()))
    return lm, pred


# ----------------------------------------------------------------------------- modules
def _conv_init_(t, fan_in, gen=None):
    """nn.Conv2d / nn.Linear default: kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))."""
    bound = 1.0 / math.sqrt(fan_in)
    with torch.no_grad():
        t.uniform_(-bound, bound, generator=gen)


class _Conv4Params(nn.Module):
    """Parameter holder with the reference's Conv4Layers names/shapes (fast.py:106-109)."""

    def __init__(self, channels, dim, n_layers=4):
        super().__init__()
        self.channels, self.dim, self.n_layers = channels, dim, n_layers
        self.cnn1 = nn.Conv2d(1, dim, (1, 5), bias=True)
        self.cnn2 = nn.Conv2d(dim, dim, (channels, 1), padding=0, bias=False)
        if n_layers == 4:
            self.cnn3 = nn.Conv2d(dim, dim, (1, 5), padding=(0, 2), bias=False)
            self.cnn4 = nn.Conv2d(dim, dim, (1, 5), padding=(0, 2), bias=False)

    def ordered(self):
        ps = [self.cnn1.weight, self.cnn1.bias, self.cnn2.weight]
        if self.n_layers == 4:
            ps += [self.cnn3.weight, self.cnn4.weight]
        return ps


class _PackedTheta(torch.autograd.Function):
    """The packed parameter block as ONE differentiable tensor without copying it: forward hands out the flat block the
    parameters already alias (``flat_params``), backward hands every parameter its slice of the flat gradient as a
    view.  (``torch.cat`` of the parameters did the same with a copy kernel per forward pass -- 5 us of a 0.8 ms step
    at the reference's batch -- and split the gradient the same way.)"""

    @staticmethod
    def forward(ctx, flat, *params):
        ctx.shapes = [tuple(p.shape) for p in params]
        return flat.view_as(flat)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        out, off = [], 0
        for shp in ctx.shapes:
            n = 1
            for d in shp:
                n *= d
            out.append(g[off:off + n].view(shp))
            off += n
        return (None, *out)


class _FlatParamMixin:
    """Keeps a list of nn.Parameters packed, in order, in one contiguous flat buffer.

    The C ABI takes one parameter block; data-parallel training all-reduces one
    gradient block.  Parameters stay ordinary named leaves (state_dict works);
    their storage is re-packed lazily if someone moved them (``.cuda()``, ``load_state_dict``
    keeps aliasing because it copies in place).
    """

    def _ordered_params(self):
        raise NotImplementedError

    @staticmethod
    def _as_flat(ps):
        """A flat tensor over ``ps`` if they already sit back to back in one storage, else None."""
        base = ps[0]
        st = base.untyped_storage()
        off = base.storage_offset()
        pos = off
        for p in ps:
            if (p.dtype != torch.float32 or not p.is_contiguous() or p.storage_offset() != pos
                    or p.untyped_storage().data_ptr() != st.data_ptr()):
                return None
            pos += p.numel()
        return torch.empty(0, dtype=torch.float32, device=base.device).set_(st, off, (pos - off,))

    def flat_params(self):
        ps = self._ordered_params()
        flat = self._as_flat([p.data for p in ps])
        if flat is None:
            flat = torch.cat([p.detach().reshape(-1).float() for p in ps]).contiguous()
            off = 0
            for p in ps:
                p.data = flat[off:off + p.numel()].view(p.shape)
                off += p.numel()
        return flat

    def packed_theta(self):
        """The flat block as the differentiable operand of the module's autograd function (see ``_PackedTheta``)."""
        flat = self.flat_params()
        if not torch.is_grad_enabled():
            return flat
        return _PackedTheta.apply(flat.detach(), *self._ordered_params())

    def flat_grads(self):
        """One flat gradient buffer aliased by every ``p.grad`` (allocated when the grads are not packed yet)."""
        flat = self.flat_params()
        ps = self._ordered_params()
        g = self._as_flat([p.grad for p in ps]) if all(p.grad is not None for p in ps) else None
        if g is None or g.device != flat.device:
            g = torch.zeros_like(flat)
            off = 0
            for p in ps:
                p.grad = g[off:off + p.numel()].view(p.shape)
                off += p.numel()
        return g


class Conv4Layers(_Conv4Params, _FlatParamMixin):
    """Drop-in for the reference's ``Conv4Layers(channels, dim=32)`` (fast.py:103-119).

    ``forward(x[B', channels, T]) -> [B', dim]``; honours the head contract
    ``Head_cls(n_zone_channels, feature_dim)`` of fast.py:203-210.
    """

    def __init__(self, channels, dim=32, n_layers=4, act_dtype="f32"):
        super().__init__(channels, dim, n_layers)
        self.act_dtype = act_dtype          # 'bf16': bf16 activations / gradients, fp32 accumulate (config 3)
        self._plans = {}

    def _ordered_params(self):
        return self.ordered()

    def _plan(self, T):
        pl = self._plans.get(T)
        if pl is None:
            pl = ConvStackPlan(self.channels, [list(range(self.channels))], self.dim, self.n_layers, T, 1,
                               self.act_dtype)
            self._plans[T] = pl
        return pl

    def forward(self, x):
        if x.dim() != 3:
            raise ValueError("expected [batch, channels, time]")
        return _ConvStackFn.apply(x, self.packed_theta(), self._plan(x.shape[-1])).squeeze(1)


_dropout_streams = [0]


def _new_dropout_stream():
    """Every module instance that draws dropout masks owns a stream id: the zone encoders of one Head are called in
    lockstep on identically shaped tensors, so (global seed + call index) alone would hand them identical masks."""
    _dropout_streams[0] += 1
    return _dropout_streams[0]


def reset_dropout_streams():
    """Restart the per-module dropout stream numbering (and the attention call counter): a training run that builds its
    model after this call draws the same masks wherever and in whatever order it is scheduled (the experiment driver
    calls it per fold, so packing folds over worker processes does not change any result)."""
    _dropout_streams[0] = 0
    AttentionBlock._calls = 0


def _dropout_seed(stream_id, call):
    """Counter-based seed = f(torch seed, data-parallel rank, module instance, call index): splitmix64-style mixing of
    the four words so neighbouring streams / ranks / calls are unrelated."""
    rank = 0
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        rank = torch.distributed.get_rank()
    z = (torch.initial_seed() + 0x9E3779B97F4A7C15 * (stream_id + 1) + 0xBF58476D1CE4E5B9 * (rank + 1)
         + 0x94D049BB133111EB * call) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return (z ^ (z >> 31)) & 0x7FFFFFFFFFFFFFFF


class _BNStackMixin(_FlatParamMixin):
    """Shared plumbing of the BatchNorm heads: packed running buffers, per-length plans, the autograd call."""

    def flat_buffers(self):
        bufs = [t for bn in self._bns() for t in (bn.running_mean, bn.running_var)]
        flat = self._as_flat(bufs)
        if flat is None:
            flat = torch.cat([t.detach().reshape(-1).float() for t in bufs]).contiguous()
            off = 0
            for bn in self._bns():
                for name in ("running_mean", "running_var"):
                    n = bn._buffers[name].numel()
                    bn._buffers[name] = flat[off:off + n]
                    off += n
        return flat

    def _plan_for(self, T):
        plan = self._plans.get(T)
        if plan is None:
            plan = self._plans[T] = self._make_plan(T)
            if getattr(self, "_seed_dev", None) is not None:
                plan.set_seed_counter(self._seed_dev)
        return plan

    def set_seed_counter(self, counter):
        """Device-resident dropout step counter (int64 tensor, or None): see ``isd_amd.graph``."""
        self._seed_dev = counter
        for plan in self._plans.values():
            plan.set_seed_counter(counter)

    def _zone_call(self, T):
        """One forward's bookkeeping (call counter, ``num_batches_tracked``) and the arguments of ``_EEGNetFn`` after
        ``x`` -- also what ``Head`` hands to the zone-batched ``_BNZonesFn``."""
        plan = self._plan_for(T)
        theta = self.packed_theta()
        bn = self._bns()[0]
        self._calls += 1
        if self.training:
            for b in self._bns():
                b.num_batches_tracked += 1
        return ("eeg", theta, self.flat_buffers(), plan, self.training, 0.1 if bn.momentum is None else bn.momentum,
                bn.eps, self.p if self.training else 0.0, _dropout_seed(self._stream_id, self._calls),
                getattr(self, "sync_bn", True))

    def _run(self, x):
        return _EEGNetFn.apply(x, *self._zone_call(x.shape[-1])[1:])


class EEGNet_Encoder(nn.Module, _BNStackMixin):
    """Drop-in for the reference's ``EEGNet_Encoder(in_channels, feature_dim, kernel_length=64, dropout=0.25)``
    (fast.py:122-167); same sub-module / parameter / buffer names, ``forward(x[B', C, T]) -> [B', feature_dim]``.
    Train-mode dropout uses the library's own counter-based stream (statistically nn.Dropout)."""

    def __init__(self, in_channels, feature_dim, kernel_length=64, dropout=0.25):
        super().__init__()
        self.in_channels, self.feature_dim, self.kernel_length, self.p = in_channels, feature_dim, kernel_length, dropout
        F1, F2 = 8, 16
        self.temporal_conv = nn.Sequential(
            nn.Conv2d(1, F1, (1, kernel_length), padding=(0, kernel_length // 2), bias=False), nn.BatchNorm2d(F1))
        self.spatial_conv = nn.Sequential(
            nn.Conv2d(F1, F2, (in_channels, 1), groups=F1, bias=False), nn.BatchNorm2d(F2), nn.ELU(),
            nn.AvgPool2d((1, 4)), nn.Dropout(dropout))
        self.separable_conv = nn.Sequential(
            nn.Conv2d(F2, F2, (1, 16), padding=(0, 8), groups=F2, bias=False), nn.Conv2d(F2, F2, (1, 1), bias=False),
            nn.BatchNorm2d(F2), nn.ELU(), nn.AvgPool2d((1, 8)), nn.Dropout(dropout))
        self.projector = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), nn.Flatten(), nn.Linear(F2, feature_dim))
        self._plans = {}
        self._calls = 0
        self._stream_id = _new_dropout_stream()

    def _bns(self):
        return [self.temporal_conv[1], self.spatial_conv[1], self.separable_conv[2]]

    def _ordered_params(self):
        b1, b2, b3 = self._bns()
        return [self.temporal_conv[0].weight, b1.weight, b1.bias, self.spatial_conv[0].weight, b2.weight, b2.bias,
                self.separable_conv[0].weight, self.separable_conv[1].weight, b3.weight, b3.bias,
                self.projector[2].weight, self.projector[2].bias]

    def _make_plan(self, T):
        return EEGNetPlan(self.in_channels, self.feature_dim, self.kernel_length, T)

    def forward(self, x):
        if x.dim() != 3:
            raise ValueError("expected [batch, channels, time]")
        return self._run(x)


class CVBlock(nn.Module, _BNStackMixin):
    """Drop-in for the reference's ``CVBlock(n_channels, dim_token, dropout=0.5)`` (fast.py:32-100); same parameter /
    buffer names.  The projector width is fixed by a 250-sample window exactly as in the reference (fast.py:66-74),
    so other window lengths raise like the reference's shape mismatch does.  3-D or 4-D ([B', 1, C, T]) input."""

    def __init__(self, n_channels, dim_token, dropout=0.5):
        super().__init__()
        self.C, self.F1, self.D, self.F2, self.Kc, self.Kc2 = n_channels, 8, 2, 16, 64, 16
        self.in_channels, self.feature_dim, self.p = n_channels, dim_token, dropout
        self.conv1 = nn.Conv2d(1, 8, (1, 64), padding=(0, 32), bias=False)
        self.bn1 = nn.BatchNorm2d(8)
        self.conv2 = nn.Conv2d(8, 16, (n_channels, 1), groups=8, bias=False)
        self.bn2 = nn.BatchNorm2d(16)
        self.conv3 = nn.Conv2d(16, 16, (1, 16), padding=(0, 8), bias=False)
        self.bn3 = nn.BatchNorm2d(16)
        self.flat_dim = 16 * ((((250 + 1) // 8) + 1) // 2)             # the reference's 250-sample dummy pass
        self.projector = nn.Linear(self.flat_dim, dim_token)
        self._plans = {}
        self._calls = 0
        self._stream_id = _new_dropout_stream()

    def _bns(self):
        return [self.bn1, self.bn2, self.bn3]

    def _ordered_params(self):
        return [self.conv1.weight, self.bn1.weight, self.bn1.bias, self.conv2.weight, self.bn2.weight, self.bn2.bias,
                self.conv3.weight, self.bn3.weight, self.bn3.bias, self.projector.weight, self.projector.bias]

    def _make_plan(self, T):
        plan = EEGNetPlan(self.in_channels, self.feature_dim, 64, T, cvblock=True)
        if plan.flat_dim != self.flat_dim:
            raise RuntimeError(f"CVBlock: a {T}-sample window flattens to {plan.flat_dim} features but the projector "
                               f"expects {self.flat_dim} (fixed by the 250-sample window, fast.py:66-74)")
        return plan

    def forward(self, x):
        if x.dim() == 4 and x.shape[1] == 1:
            x = x[:, 0]
        if x.dim() != 3:
            raise ValueError("expected [batch, channels, time] or [batch, 1, channels, time]")
        return self._run(x)


class HeadConv_Paper_Version(nn.Module, _BNStackMixin):
    """Drop-in for the reference's ``HeadConv_Paper_Version(in_channels, feature_dim=32)`` (fast.py:170-196); same
    parameter / buffer names, ``forward(x[B', C, T]) -> [B', feature_dim]``."""

    def __init__(self, in_channels, feature_dim=32):
        super().__init__()
        F1, F2, F3, F4 = feature_dim // 2, feature_dim // 3, feature_dim // 3, feature_dim
        self.in_channels, self.feature_dim = in_channels, feature_dim
        self.cnn1_t = nn.Conv2d(1, F1, (1, 3), bias=True)
        self.cnn1_s = nn.Conv2d(F1, F1, (in_channels, 1), padding=0, bias=False)
        self.norm1 = nn.BatchNorm2d(F1)
        self.cnn2 = nn.Conv2d(F1, F2, (1, 3), bias=False)
        self.norm2 = nn.BatchNorm2d(F2)
        self.cnn3 = nn.Conv2d(F2, F3, (1, 3), bias=False)
        self.norm3 = nn.BatchNorm2d(F3)
        self.cnn4 = nn.Conv2d(F3, F4, (1, 3), bias=False)
        self.norm4 = nn.BatchNorm2d(F4)
        self._plans = {}

    def _bns(self):
        return [self.norm1, self.norm2, self.norm3, self.norm4]

    def _ordered_params(self):
        return [self.cnn1_t.weight, self.cnn1_t.bias, self.cnn1_s.weight, self.norm1.weight, self.norm1.bias,
                self.cnn2.weight, self.norm2.weight, self.norm2.bias, self.cnn3.weight, self.norm3.weight,
                self.norm3.bias, self.cnn4.weight, self.norm4.weight, self.norm4.bias]

    def _zone_call(self, T):
        plan = self._plans.get(T)
        if plan is None:
            plan = self._plans[T] = PaperHeadPlan(self.in_channels, self.feature_dim, T)
        theta = self.packed_theta()
        if self.training:
            for b in self._bns():
                b.num_batches_tracked += 1
        bn = self.norm1
        return ("paper", theta, self.flat_buffers(), plan, self.training, 0.1 if bn.momentum is None else bn.momentum,
                bn.eps, getattr(self, "sync_bn", True))

    def forward(self, x):
        if x.dim() != 3:
            raise ValueError("expected [batch, channels, time]")
        return _PaperHeadFn.apply(x, *self._zone_call(x.shape[-1])[1:])


class Head(nn.Module, _FlatParamMixin):
    """Drop-in for the reference's ``Head(head, electrodes, zone_dict, feature_dim)`` (fast.py:199-210).

    All zones run in one launch per layer; ``forward(x[B', C, T]) -> [B', Z, F]``.
    ``forward_windows(x[B, C, T], window_len, slide_step)`` additionally folds the
    sliding windows of ``FAST.forward_head`` into the kernels' index arithmetic.
    """

    def __init__(self, head, electrodes, zone_dict, feature_dim, act_dtype="f32"):
        super().__init__()
        self.act_dtype = act_dtype
        if head not in HEAD_REGISTRY:
            raise KeyError(f"head '{head}' is not in the head registry {sorted(HEAD_REGISTRY)}")    # globals()[head]
        self.head_name = head
        self.fused = head == "Conv4Layers"
        self.electrodes = list(electrodes)
        self.index_dict = {}
        self.encoders = nn.ModuleDict()
        for area, ch_names in zone_dict.items():
            self.index_dict[area] = torch.tensor([self.electrodes.index(ch) for ch in ch_names])
            self.encoders[area] = (_Conv4Params(len(ch_names), feature_dim) if self.fused
                                   else HEAD_REGISTRY[head](len(ch_names), feature_dim))
        self.feature_dim = feature_dim
        self._plans = {}
        self._zone_streams = {}

    def _ordered_params(self):
        if not self.fused:
            return [p for enc in self.encoders.values() for p in enc._ordered_params()]
        return [p for enc in self.encoders.values() for p in enc.ordered()]

    def _plan(self, window_len, slide_step):
        key = (window_len, slide_step)
        pl = self._plans.get(key)
        if pl is None:
            pl = ConvStackPlan(len(self.electrodes), [v.tolist() for v in self.index_dict.values()], self.feature_dim,
                               4, window_len, slide_step, self.act_dtype)
            self._plans[key] = pl
        return pl

    def _theta(self):
        return self.packed_theta()

    @staticmethod
    def _zone_batchable(encs, xw):
        """The BatchNorm heads' zones go out in zone-batched launches (``_BNZonesFn``) when only parameter gradients
        of a train-mode pass, or no gradients, are asked for on a single device."""
        if os.environ.get("ISD_ZONE_BATCH_OFF") or not 1 <= len(encs) <= 8 or xw.dim() != 3 or xw.requires_grad:
            return False
        kind = type(encs[0])
        if kind not in (EEGNet_Encoder, CVBlock, HeadConv_Paper_Version) or any(type(e) is not kind for e in encs):
            return False
        if any(e.training != encs[0].training for e in encs) or encs[0].feature_dim > 64:
            return False                                           # (wider projectors spread their weight gradient over grid.z)
        if kind is not HeadConv_Paper_Version and len({e.in_channels >= 128 for e in encs}) > 1:
            return False                # launch i must be the same kernel in every zone: the spatial projection takes
                                        # whole rows per workgroup from 128 channels on (csrc/eegnet.hip, stage 1)
        if not encs[0].training and torch.is_grad_enabled() and any(p.requires_grad for e in encs for p in e.parameters()):
            return False                                           # eval-mode gradients: the per-zone backward_x path
        return _bn_sync_world(getattr(encs[0], "sync_bn", True), encs[0].training)[1] == 1

    def _per_zone(self, xw):
        """Registry heads other than Conv4Layers: one encoder call per zone on its gathered channels (fast.py:210)."""
        encs = list(self.encoders.values())
        for area in self.encoders:
            if self.index_dict[area].device != xw.device:
                self.index_dict[area] = self.index_dict[area].to(xw.device)
        if self._zone_batchable(encs, xw):
            calls = [enc._zone_call(xw.shape[-1]) for enc in encs]
            return _BNZonesFn.apply(xw, [self.index_dict[a] for a in self.encoders],
                                    [(c[0],) + tuple(c[2:-1]) for c in calls], *[c[1] for c in calls])
        # The zones are independent until the stack: each runs on its own HIP stream (forked from / joined to the
        # caller's), so their short kernels overlap on the GPU and, in a captured graph, form parallel branches.
        main = torch.cuda.current_stream(xw.device)
        streams = self._zone_streams.get(xw.device)
        if streams is None:
            streams = self._zone_streams[xw.device] = [torch.cuda.Stream(xw.device) for _ in self.encoders]
        outs = []
        for (area, enc), st in zip(self.encoders.items(), streams):
            idx = self.index_dict[area]
            if idx.device != xw.device:
                idx = self.index_dict[area] = idx.to(xw.device)
            st.wait_stream(main)
            with torch.cuda.stream(st):
                outs.append(enc(xw.index_select(1, idx)))
            xw.record_stream(st)
        for st, o in zip(streams, outs):
            main.wait_stream(st)
            o.record_stream(main)
        return torch.stack(outs, dim=1)

    def forward_windows(self, x, window_len, slide_step):
        """x [B, C, T] -> [B*N, Z, F] with N sliding windows per trial (fast.py:247-251)."""
        if not self.fused:
            xw = x.unfold(-1, int(window_len), int(slide_step))                       # B C N T (view)
            B, Cc, N, T = xw.shape
            return self._per_zone(xw.permute(0, 2, 1, 3).reshape(B * N, Cc, T))
        return _ConvStackFn.apply(x, self._theta(), self._plan(int(window_len), int(slide_step)))

    def forward(self, x):
        if not self.fused:
            return self._per_zone(x)
        return _ConvStackFn.apply(x, self._theta(), self._plan(int(x.shape[-1]), 1))


# the reference resolves ``globals()[config.head]`` in fast.models.fast (fast.py:203); this is that namespace
HEAD_REGISTRY = {"Conv4Layers": Conv4Layers, "EEGNet_Encoder": EEGNet_Encoder, "CVBlock": CVBlock,
                 "HeadConv_Paper_Version": HeadConv_Paper_Version}


def register_head(name, cls):
    """Add a head class honouring ``cls(n_zone_channels, feature_dim)`` / ``forward(x[B', Cz, T]) -> [B', F]``."""
    HEAD_REGISTRY[name] = cls


class AttentionBlock(nn.Module):
    """Drop-in for the reference's ``AttentionBlock(embed_dim, hidden_dim, num_heads, dropout)`` (fast.py:10-29):
    pre-LN multi-head self-attention + pre-LN MLP, both with residuals.  Same sub-module / parameter names."""
    _calls = 0

    def __init__(self, embed_dim, hidden_dim, num_heads, dropout=0.0):
        super().__init__()
        self.layer_norm_1 = nn.LayerNorm(embed_dim)
        self.attn = nn.MultiheadAttention(embed_dim, num_heads, dropout=dropout, batch_first=True)
        self.layer_norm_2 = nn.LayerNorm(embed_dim)
        self.linear = nn.Sequential(nn.Linear(embed_dim, hidden_dim), nn.GELU(), nn.Dropout(dropout),
                                    nn.Linear(hidden_dim, embed_dim), nn.Dropout(dropout))
        self.num_heads, self.p = num_heads, dropout
        self._stream_id = _new_dropout_stream()

    def forward(self, x):
        p = self.p if self.training else 0.0
        AttentionBlock._calls += 1
        seed = _dropout_seed(self._stream_id, AttentionBlock._calls)
        h = layer_norm(x, self.layer_norm_1.weight, self.layer_norm_1.bias, self.layer_norm_1.eps)
        qkv = linear(h, self.attn.in_proj_weight, self.attn.in_proj_bias)
        ctx = _AttentionFn.apply(qkv, self.num_heads, p, seed)
        x = linear_residual(ctx, self.attn.out_proj.weight, self.attn.out_proj.bias, x)
        h = layer_norm(x, self.layer_norm_2.weight, self.layer_norm_2.bias, self.layer_norm_2.eps)
        h = linear(h, self.linear[0].weight, self.linear[0].bias, act=True)
        if p > 0.0:
            h = torch.nn.functional.dropout(h, p, True)
            return x + torch.nn.functional.dropout(linear(h, self.linear[3].weight, self.linear[3].bias), p, True)
        return linear_residual(h, self.linear[3].weight, self.linear[3].bias, x)


class FAST(nn.Module):
    """The reference's ``FAST(config)`` (fast.py:213-284) on the HIP kernels: ``forward_head``,
    ``batched_forward_head``, ``forward_transformer`` and the three forward modes.

    Parameter names equal the reference's (``head.encoders.<Zone>.cnn1.weight``, ``input_layer.0.weight``,
    ``transformer.0.attn.in_proj_weight``, ``pos_embedding``, ``cls_token``, ``last_layer.weight`` ...), so
    ``load_state_dict(reference_state_dict)`` works unchanged.
    """
    name = "FAST"

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.n_tokens = (config.seq_len - config.window_len) // config.slide_step + 1
        self.head = Head(config.head, config.electrodes, config.zone_dict, config.dim_cnn,
                         getattr(config, "act_dtype", "f32"))
        self.input_layer = nn.Sequential(nn.Linear(config.dim_cnn * len(config.zone_dict), config.dim_token), nn.GELU())
        self.transformer = nn.Sequential(*[AttentionBlock(config.dim_token, config.dim_token * 2, config.num_heads,
                                                          dropout=config.dropout) for _ in range(config.num_layers)])
        self.pos_embedding = nn.Parameter(torch.randn(1, self.n_tokens + 1, config.dim_token))
        self.cls_token = nn.Parameter(torch.randn(1, 1, config.dim_token))
        self.last_layer = nn.Linear(config.dim_token, config.n_classes)
        self.dropout = nn.Dropout(config.dropout)
        self.fuse_tail = True               # one launch per direction for the transformer tail where it applies
        self._tail_stream, self._tail_calls = _new_dropout_stream(), 0
        self.seed_dev = None                # int64 device counter mixed into the tail's dropout seed (graph replay)

    def _tail_params(self):
        """The tail's parameters in the order of the fused kernels' flat block (include/isd_hip.h)."""
        ps = [self.pos_embedding, self.cls_token]
        for blk in self.transformer:
            ps += [blk.layer_norm_1.weight, blk.layer_norm_1.bias, blk.attn.in_proj_weight, blk.attn.in_proj_bias,
                   blk.attn.out_proj.weight, blk.attn.out_proj.bias, blk.layer_norm_2.weight, blk.layer_norm_2.bias,
                   blk.linear[0].weight, blk.linear[0].bias, blk.linear[3].weight, blk.linear[3].bias]
        return ps + [self.last_layer.weight, self.last_layer.bias]

    def _tail_flat(self, ps):
        """The tail's parameters as one contiguous block (re-packed in place if something moved them)."""
        flat = _FlatParamMixin._as_flat([p.data for p in ps])
        if flat is None:
            flat = torch.cat([p.detach().reshape(-1).float() for p in ps]).contiguous()
            off = 0
            for p in ps:
                p.data = flat[off:off + p.numel()].view(p.shape)
                off += p.numel()
        return flat

    def _tail_fusable(self, tok):
        c = self.config
        blk = self.transformer[0] if len(self.transformer) else None
        return (self.fuse_tail and blk is not None and tok.is_cuda and tok.dtype == torch.float32
                and blk.layer_norm_1.eps == 1e-5 and blk.layer_norm_2.eps == 1e-5
                and bool(_lib.lib().isd_tail_fused_supported(tok.shape[1], c.dim_token, c.num_heads, len(self.transformer),
                                                             blk.linear[0].out_features, c.n_classes)))

    def forward_head(self, x, step_override=None):
        step = self.config.slide_step if step_override is None else step_override
        B = x.shape[0]
        feat = self.head.forward_windows(x, self.config.window_len, step)
        return feat.view(B, -1, feat.shape[1], feat.shape[2])          # [B, N, Z, F]

    def batched_forward_head(self, x, step, batch_size):
        return torch.cat([self.forward_head(mb, step) for mb in torch.split(x, batch_size, dim=0)], dim=0)

    def token_logits(self, x):
        feat = self.forward_head(x)
        B, N, Z, Fd = feat.shape
        tok = linear(feat.reshape(B, N, Z * Fd), self.input_layer[0].weight, self.input_layer[0].bias, act=True)
        return linear(tok, self.last_layer.weight, self.last_layer.bias)   # [B, N, n_classes]

    def forward_transformer(self, feature):
        """feature [B, N, Z, F] -> logits [B, n_classes]  (fast.py:260-268)."""
        B, N, Z, Fd = feature.shape
        tok = linear(feature.reshape(B, N, Z * Fd), self.input_layer[0].weight, self.input_layer[0].bias, act=True)
        if self._tail_fusable(tok):
            c = self.config
            ps = self._tail_params()
            self._tail_calls += 1
            p_blk = float(self.transformer[0].p) if self.training else 0.0
            p_cls = float(self.dropout.p) if self.training else 0.0
            # the training kernel (dropout + the backward's record) whenever a gradient may be asked for or masks are drawn
            train = (torch.is_grad_enabled() and (tok.requires_grad or any(p.requires_grad for p in ps))) \
                or p_blk > 0.0 or p_cls > 0.0
            cfg = (self.pos_embedding.shape[1], c.num_heads, len(self.transformer), self.transformer[0].linear[0].out_features,
                   c.n_classes, p_blk, p_blk, p_cls, _dropout_seed(self._tail_stream, self._tail_calls), train,
                   self.seed_dev)
            return _TailFusedFn.apply(tok, self._tail_flat(ps), cfg, *ps)
        tok = _EmbedFn.apply(tok, self.cls_token, self.pos_embedding[:, :N + 1].contiguous())
        tok = self.transformer(tok)
        cls = self.dropout(tok[:, 0].contiguous())
        return linear(cls, self.last_layer.weight, self.last_layer.bias)

    def forward(self, x, forward_mode="default"):
        if forward_mode == "default":
            return self.forward_transformer(self.forward_head(x))
        if forward_mode == "train_head":
            lt = self.token_logits(x)
            if torch.is_grad_enabled():
                return lt.mean(dim=1)
            return token_mean_predict(lt)[0]
        if forward_mode == "train_transformer":
            with torch.no_grad():
                feat = self.forward_head(x)
            return self.forward_transformer(feat)
        raise NotImplementedError


class FeatureCNN(nn.Module):
    """Build-defined classifier over spec-S features: ``Conv4Layers(nb*C, F)`` through the
    unmodified head contract, then ``Linear(F, n_classes)``.  ``n_layers=2`` is BASELINE config 1's 2-layer CNN."""

    def __init__(self, in_channels, feature_dim=32, n_classes=5, n_layers=4, act_dtype="f32"):
        super().__init__()
        self.cnn = Conv4Layers(in_channels, feature_dim, n_layers, act_dtype)
        self.fc = nn.Linear(feature_dim, n_classes)

    def token_logits(self, feats):
        B = feats.shape[0]
        h = self.cnn(feats.reshape(B, -1, feats.shape[-1]))
        return linear(h, self.fc.weight, self.fc.bias).unsqueeze(1)      # [B, 1, n_classes]

    def forward(self, feats):
        return self.token_logits(feats).squeeze(1)


def fast_config(electrodes=None, zone_dict=None, **kw):
    """Attribute bag with the reference's production values (scripts/train_fast.py:293-307)."""
    import types
    d = dict(electrodes=ELECTRODES if electrodes is None else electrodes, zone_dict=ZONES if zone_dict is None else zone_dict,
             dim_cnn=32, dim_token=32, seq_len=800, window_len=250, slide_step=125, head="Conv4Layers", n_classes=5,
             num_layers=4, num_heads=8, dropout=0.1)
    d.update(kw)
    return types.SimpleNamespace(**d)

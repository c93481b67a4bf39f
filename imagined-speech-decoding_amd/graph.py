"""Whole optimisation steps as HIP graphs.

The reference fine-tunes at batch 64 (scripts/train_fast.py:274): a step is ~40 kernels of a few microseconds each
and its wall time is the host's launch path, not the GPU.  ``GraphedTrainStep`` captures one step -- batch gather,
forward, loss, backward, AdamW -- once and replays it: one ``hipGraphLaunch`` per step.  What changes between
replays lives in device memory the graph reads: the batch's row indices, the learning rate (a tensor:
``AdamW(capturable=True)``) and the dropout step counter (``FAST.seed_dev``, mixed into the counter-based masks by
the kernels, include/isd_hip.h ``seed_dev``).
"""
import torch

from .nn import token_mean_cross_entropy, unit_grad


def graph_safe(model):
    """True when every dropout mask of ``model`` advances through device memory: the fused transformer tail
    (``seed_dev``), ``EEGNet_Encoder`` / ``CVBlock`` zones (``set_seed_counter``); ``Conv4Layers`` and
    ``HeadConv_Paper_Version`` draw none.  The per-operator transformer blocks take their seeds by value and would
    replay one mask for ever."""
    from . import nn as inn
    from . import _lib
    if not isinstance(model, inn.FAST) or not model.fuse_tail:
        return False
    if not model.head.fused and not all(isinstance(e, (inn.EEGNet_Encoder, inn.CVBlock, inn.HeadConv_Paper_Version))
                                        for e in model.head.encoders.values()):
        return False                                           # an unknown registered head: cannot vouch for its masks
    import torch.distributed as dist
    if not model.head.fused and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return False                                           # synchronised BatchNorm exchanges sums between stages
    c = model.config
    return bool(_lib.lib().isd_tail_fused_supported(model.n_tokens, c.dim_token, c.num_heads, len(model.transformer),
                                                    2 * c.dim_token, c.n_classes))


class GraphedTrainStep:
    """``step(idx, lr)``: one AdamW step on rows ``idx`` of the device-resident ``X`` / ``y``.  Full batches replay the
    captured graph; a ragged last batch runs the same code eagerly.  ``loss_sum`` accumulates ``loss * len(idx)`` on
    the device (read it once per epoch: no per-step host synchronisation)."""

    def __init__(self, model, opt, X, y, batch_size, forward_mode="default", warmup=3):
        if not (X.is_cuda and y.is_cuda):
            raise ValueError("GraphedTrainStep needs device-resident trials and labels")
        g0 = opt.param_groups[0]
        if not (isinstance(g0["lr"], torch.Tensor) and g0["lr"].is_cuda and g0.get("capturable", False)):
            raise ValueError("GraphedTrainStep needs AdamW(lr=<device tensor>, capturable=True)")
        self.model, self.opt, self.X, self.y, self.mode = model, opt, X, y, forward_mode
        self.bs = int(batch_size)
        dev = X.device
        self.idx = torch.zeros(self.bs, dtype=torch.int64, device=dev)
        self.loss_sum = torch.zeros((), dtype=torch.float32, device=dev)
        self.lr = g0["lr"]
        # the dropout step counter = the number of optimisation steps COMPLETED so far (0 while the first step runs): the
        # optimizer's own device-side step count when it keeps one (FusedAdamW advances it inside its kernel, at the end
        # of the step: no launch of ours), else a counter this class advances with a one-element add behind the step.
        # Both counters therefore give a step the same masks.  Sharing has consequences that are wanted: restoring the
        # optimizer (``opt.load_state_dict``) puts the mask stream back where that optimizer state was taken -- a resumed
        # run draws the masks the uninterrupted one would have drawn; a step without any gradient still advances it
        # (optim.FusedAdamW.step).
        from .optim import FusedAdamW
        self._own_counter = not (isinstance(opt, FusedAdamW) and "step" in opt.state["flat"])
        model.seed_dev = (torch.zeros(1, dtype=torch.int64, device=dev) if self._own_counter
                          else opt.state["flat"]["step"][:1])
        for enc in getattr(model.head, "encoders", {}).values():
            if hasattr(enc, "set_seed_counter"):
                enc.set_seed_counter(model.seed_dev)
        # warm-up steps build plans, workspaces and optimizer state; the training itself must not see them:
        # parameters, optimizer state AND module buffers (BatchNorm running statistics / num_batches_tracked of the
        # EEGNet_Encoder / CVBlock / HeadConv_Paper_Version heads) are put back after the capture
        params = [p for g in opt.param_groups for p in g["params"]]
        keep = [p.detach().clone() for p in params]
        keep_buf = {n: b.detach().clone() for n, b in model.named_buffers()}     # by NAME: the first forward of a
        # BatchNorm head re-registers its running statistics as views of one packed block (nn._BNStackMixin)
        self.lr.zero_()                                        # lr 0 + restored parameters: the warm-up is a no-op
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._body(self.idx)
        cur.wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._body(self.idx)
        with torch.no_grad():
            for p, k in zip(params, keep):
                p.copy_(k)
            for n, b in model.named_buffers():
                b.copy_(keep_buf[n])
            for st in opt.state.values():                      # fresh optimizer state, as before the warm-up
                for v in st.values():
                    if isinstance(v, torch.Tensor):
                        v.zero_()
            self.loss_sum.zero_()
            if self._own_counter:
                model.seed_dev.zero_()

    def _body(self, idx):
        self.opt.zero_grad(set_to_none=True)
        xb, yb = self.X.index_select(0, idx), self.y.index_select(0, idx)
        loss = token_mean_cross_entropy(self.model(xb, forward_mode=self.mode), yb)
        loss.backward(unit_grad(loss.device))             # (no ones fill, no multiply by it: nn.unit_grad)
        self.opt.step()
        if self._own_counter:
            self.model.seed_dev.add_(1)
        self.loss_sum.add_(loss.detach(), alpha=float(idx.numel()))      # one kernel (a product first would be two)

    def step(self, idx, lr):
        self.lr.fill_(float(lr))
        if idx.numel() == self.bs:
            self.idx.copy_(idx, non_blocking=True)
            self.graph.replay()
        else:
            self._body(idx)

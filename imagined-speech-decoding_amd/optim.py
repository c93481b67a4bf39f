"""AdamW for the autograd modules (isd_amd.nn) as one HIP launch.

The reference's training loop is ``optim.AdamW(self.parameters(), lr=0.0005)`` (src/fast/train/trainer.py:49).  FAST
keeps one ``nn.Parameter`` per reference tensor (~60 of them) and autograd gives every gradient its own buffer;
torch's multi-tensor AdamW takes three to four launches of 5-10 us each for them -- 33 us of the 0.87 ms the whole step
takes at the reference's batch of 64.  ``FusedAdamW`` hands the pointers to ``isd_adamw_multi_step`` (csrc/adamw.hip):
one launch, torch's operation order, the learning rate and the step count optionally in device memory so that a
captured HIP graph (isd_amd.graph) replays the update with nothing to advance on the host.

``FusedAdamW`` is a ``torch.optim.Optimizer`` (one parameter group): ``param_groups`` / ``zero_grad`` / LR schedulers
(``LambdaLR`` writes ``param_groups[0]["lr"]``, float or one-element device tensor) work as with torch's class; the
moment estimates live in two flat blocks, exposed as ``state["flat"]``, and ``state_dict()`` / ``load_state_dict()``
carry those blocks and the step count.  One step count serves all tensors (torch counts per tensor: the same thing
whenever a tensor either always or never receives a gradient, as in the reference's modes).
"""
import ctypes as C
import struct

import torch

from . import _lib


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, capturable=False):
        if isinstance(lr, torch.Tensor) and not (lr.is_cuda and lr.dtype == torch.float32 and lr.numel() == 1):
            raise TypeError("FusedAdamW: a tensor lr must be one float32 element on the device")
        if capturable and not isinstance(lr, torch.Tensor):
            raise ValueError("FusedAdamW(capturable=True) needs lr as a device tensor (a replayed graph reads it)")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=float(eps), weight_decay=float(weight_decay),
                                      capturable=bool(capturable)))
        if len(self.param_groups) != 1:
            raise ValueError("FusedAdamW takes one parameter group (uniform hyper-parameters: one launch)")
        g = self.param_groups[0]
        g["params"] = [p for p in g["params"] if p.requires_grad]
        params = g["params"]
        if not params:
            raise ValueError("FusedAdamW: no trainable parameters")
        for p in params:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise TypeError("FusedAdamW needs contiguous float32 parameters on a HIP device (the step is a HIP kernel)")
        dev = params[0].device
        # both moments of every tensor in one block each; a tensor starts on a 16-byte boundary
        offs, tot = [], 0
        for p in params:
            offs.append(tot)
            tot += (p.numel() + 3) & ~3
        self._exp_avg = torch.zeros(tot, dtype=torch.float32, device=dev)
        self._exp_avg_sq = torch.zeros(tot, dtype=torch.float32, device=dev)
        self._step_dev = torch.zeros(4, dtype=torch.int64, device=dev) if capturable else None
        self._step = 0
        n = len(params)
        self._numel = (C.c_int64 * n)(*[p.numel() for p in params])
        self._m = (C.c_void_p * n)(*[self._exp_avg.data_ptr() + 4 * o for o in offs])
        self._v = (C.c_void_p * n)(*[self._exp_avg_sq.data_ptr() + 4 * o for o in offs])
        self._p = (C.c_void_p * n)()
        self._g = (C.c_void_p * n)()
        self._n_active = (C.c_int64 * n)()
        self.state["flat"] = {"exp_avg": self._exp_avg, "exp_avg_sq": self._exp_avg_sq}
        if capturable:
            self.state["flat"]["step"] = self._step_dev           # [steps taken, kernel scratch, b1^t, b2^t (double bits)]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g = self.param_groups[0]
        for i, p in enumerate(g["params"]):
            gr = p.grad
            if gr is None:                                         # as torch: a parameter without a gradient is skipped
                self._n_active[i] = 0
                continue
            if not (gr.is_cuda and gr.dtype == torch.float32 and gr.is_contiguous()):
                raise TypeError("FusedAdamW: gradients must be contiguous float32 device tensors")
            self._p[i], self._g[i], self._n_active[i] = p.data_ptr(), gr.data_ptr(), self._numel[i]
        lr = g["lr"]
        lr_dev = lr.data_ptr() if isinstance(lr, torch.Tensor) else None
        self._step += 1
        if self._step_dev is not None and not any(self._n_active[i] for i in range(len(g["params"]))):
            # no gradient at all: the kernel has nothing to launch, but the device-side step count (which a captured
            # training step also reads as its dropout counter, graph.GraphedTrainStep) still counts the step
            self._step_dev[0:1].add_(1)
            return loss
        _lib.check(_lib.lib().isd_adamw_multi_step(
            len(g["params"]), self._p, self._g, self._m, self._v, self._n_active, 0.0 if lr_dev else float(lr),
            g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self._step, lr_dev,
            self._step_dev.data_ptr() if self._step_dev is not None else None,
            C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return loss

    def state_dict(self):
        g = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        if isinstance(g["lr"], torch.Tensor):
            g["lr"] = float(g["lr"])
        step = int(self._step_dev[0]) if self._step_dev is not None else self._step
        return {"exp_avg": self._exp_avg.clone(), "exp_avg_sq": self._exp_avg_sq.clone(), "step": step, "group": g}

    def load_state_dict(self, sd):
        """Restores what ``state_dict`` returned: both moment blocks, the step count and the group's hyper-parameters.
        The layout is this class's own (two flat blocks in the order of the parameters, ONE step count for all tensors --
        torch keeps one per tensor, so a parameter that only starts to receive gradients later is bias-corrected as if it
        had been stepped from the start); a ``torch.optim.AdamW`` state dict is refused, not misread."""
        if not isinstance(sd, dict) or set(sd) != {"exp_avg", "exp_avg_sq", "step", "group"}:
            raise ValueError("FusedAdamW.load_state_dict: not a FusedAdamW state dict (keys exp_avg, exp_avg_sq, step, "
                             f"group; got {sorted(sd) if isinstance(sd, dict) else type(sd).__name__})")
        for k in ("exp_avg", "exp_avg_sq"):
            if tuple(sd[k].shape) != tuple(self._exp_avg.shape):
                raise ValueError(f"FusedAdamW.load_state_dict: {k} has {tuple(sd[k].shape)} elements, this optimizer's "
                                 f"parameters need {tuple(self._exp_avg.shape)} (another model or parameter order)")
        g = self.param_groups[0]
        for k, v in sd["group"].items():
            if k == "lr" and isinstance(g["lr"], torch.Tensor):
                g["lr"].fill_(float(v))                            # the device tensor a captured graph reads stays in place
            elif k in ("betas", "eps", "weight_decay", "lr"):
                g[k] = tuple(v) if k == "betas" else float(v)
        self._exp_avg.copy_(sd["exp_avg"])
        self._exp_avg_sq.copy_(sd["exp_avg_sq"])
        self._step = int(sd["step"])
        if self._step_dev is not None:
            b1, b2 = self.param_groups[0]["betas"]
            bits = [struct.unpack("q", struct.pack("d", b ** self._step))[0] for b in (b1, b2)]
            self._step_dev.copy_(torch.tensor([self._step, 0] + bits, dtype=torch.int64))

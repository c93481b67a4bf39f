"""AdamW for the autograd modules (isd_amd.nn) as one HIP launch.

The reference's training loop is ``optim.AdamW(self.parameters(), lr=0.0005)`` (src/fast/train/trainer.py:49).  FAST
keeps one ``nn.Parameter`` per reference tensor (~60 of them) and autograd gives every gradient its own buffer;
torch's multi-tensor AdamW takes three to four launches of 5-10 us each for them -- 33 us of the 0.87 ms the whole step
takes at the reference's batch of 64.  ``FusedAdamW`` hands the pointers to ``isd_adamw_multi_step`` (csrc/adamw.hip):
one launch, torch's operation order, the learning rate and the step count optionally in device memory so that a
captured HIP graph (isd_amd.graph) replays the update with nothing to advance on the host.

Only the part of the ``torch.optim.Optimizer`` surface that this package's training loops use is provided:
``param_groups`` (one group; ``lr`` may be a float or a one-element device tensor), ``step()``, ``zero_grad()``,
``state`` (the moment blocks and the device step counter, as tensors) and ``state_dict()`` / ``load_state_dict()``.
"""
import ctypes as C

import torch

from . import _lib


class FusedAdamW:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, capturable=False):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("FusedAdamW: no trainable parameters")
        for p in params:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise TypeError("FusedAdamW needs contiguous float32 parameters on a HIP device (the step is a HIP kernel)")
        dev = params[0].device
        if isinstance(lr, torch.Tensor) and not (lr.is_cuda and lr.dtype == torch.float32 and lr.numel() == 1):
            raise TypeError("FusedAdamW: a tensor lr must be one float32 element on the device")
        if capturable and not isinstance(lr, torch.Tensor):
            raise ValueError("FusedAdamW(capturable=True) needs lr as a device tensor (a replayed graph reads it)")
        self.param_groups = [dict(params=params, lr=lr, betas=tuple(betas), eps=float(eps),
                                  weight_decay=float(weight_decay), capturable=bool(capturable))]
        # both moments of every tensor in one block each; a tensor starts on a 16-byte boundary
        offs, tot = [], 0
        for p in params:
            offs.append(tot)
            tot += (p.numel() + 3) & ~3
        self._exp_avg = torch.zeros(tot, dtype=torch.float32, device=dev)
        self._exp_avg_sq = torch.zeros(tot, dtype=torch.float32, device=dev)
        self._step_dev = torch.zeros(2, dtype=torch.int64, device=dev) if capturable else None
        self._step = 0
        n = len(params)
        self._numel = (C.c_int64 * n)(*[p.numel() for p in params])
        self._m = (C.c_void_p * n)(*[self._exp_avg.data_ptr() + 4 * o for o in offs])
        self._v = (C.c_void_p * n)(*[self._exp_avg_sq.data_ptr() + 4 * o for o in offs])
        self._p = (C.c_void_p * n)()
        self._g = (C.c_void_p * n)()
        self._n_active = (C.c_int64 * n)()
        self.state = {"flat": {"exp_avg": self._exp_avg, "exp_avg_sq": self._exp_avg_sq}}
        if capturable:
            self.state["flat"]["step"] = self._step_dev

    def zero_grad(self, set_to_none=True):
        for p in self.param_groups[0]["params"]:
            if p.grad is None:
                continue
            if set_to_none:
                p.grad = None
            else:
                p.grad.detach_()
                p.grad.zero_()

    @torch.no_grad()
    def step(self):
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
        _lib.check(_lib.lib().isd_adamw_multi_step(
            len(g["params"]), self._p, self._g, self._m, self._v, self._n_active, 0.0 if lr_dev else float(lr),
            g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self._step, lr_dev,
            self._step_dev.data_ptr() if self._step_dev is not None else None,
            C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def state_dict(self):
        g = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        if isinstance(g["lr"], torch.Tensor):
            g["lr"] = float(g["lr"])
        step = int(self._step_dev[0]) if self._step_dev is not None else self._step
        return {"exp_avg": self._exp_avg.clone(), "exp_avg_sq": self._exp_avg_sq.clone(), "step": step, "group": g}

    def load_state_dict(self, sd):
        self._exp_avg.copy_(sd["exp_avg"])
        self._exp_avg_sq.copy_(sd["exp_avg_sq"])
        self._step = int(sd["step"])
        if self._step_dev is not None:
            self._step_dev[0] = self._step
            self._step_dev[1] = 0

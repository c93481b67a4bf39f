"""Measured deviation of the bf16 (BASELINE config 3) paths: the feature classifier step against G15 (the reference's
Conv4Layers(576, 32) + Linear under torch.autocast(bfloat16) and in fp32), and the raw-EEG fused pair against the fp32
kernels.  Prints max |a - b| / max |b| per tensor -- the numbers the tolerances in tests/test_cnn_gpu.py are set from."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import load_golden, rel_err
from test_oracle import g15_params
import isd_amd
import isd_amd.nn as inn
from isd_amd.classifier import _FeatureModel
from oracle import cnn as ocnn

g = load_golden("g15_bf16_autocast.npz")
p, x, y = g15_params()
print("reference autocast vs reference fp32: logits %.2e" % rel_err(g["bf16.logits"], g["fp32.logits"]))
for k in [k for k in g.files if k.startswith("bf16.") and "grad" in k]:
    print("   ", k[5:], "%.2e" % rel_err(g[k], g["fp32." + k[5:]]))
m = _FeatureModel(576, 32, 5, 4, "bf16").cuda()
m.net.cnn.load_state_dict({k: v for k, v in p.items() if k.startswith("cnn")})
m.net.fc.load_state_dict({k[3:]: v for k, v in p.items() if k.startswith("fc.")})
out = isd_amd.HotPath(m).forward(x.cuda().contiguous(), y.cuda(), want_grad=True)
for ref in ("bf16", "fp32"):
    print(f"HIP bf16 vs reference {ref}: logits %.2e loss %.2e" % (rel_err(out["logits"].cpu().numpy(), g[f"{ref}.logits"]),
                                                                   abs(float(out["loss"]) - float(g[f"{ref}.loss"]))))
    for k, q in m.net.named_parameters():
        want = g[f"{ref}." + ("fc.grad." + k[3:] if k.startswith("fc.") else "cnn.grad." + k[4:])]
        got = q.grad.detach().cpu().numpy()
        got = got[:, :, ::9] if k == "cnn.cnn2.weight" else got
        print("   ", k, "%.2e" % rel_err(got, want))
torch.manual_seed(3)
h32 = inn.Head("Conv4Layers", ocnn.ELECTRODES, ocnn.ZONES, 32).cuda()
h16 = inn.Head("Conv4Layers", ocnn.ELECTRODES, ocnn.ZONES, 32, act_dtype="bf16").cuda()
h16.load_state_dict(h32.state_dict())
for B, T in ((37, 512), (64, 800)):
    xx = torch.randn(B, 64, T, device="cuda")
    n_win = (T - 250) // 125 + 1
    w = torch.randn(B * n_win, 8, 32, device="cuda")
    outs = []
    for h in (h32, h16):
        h.zero_grad(set_to_none=True)
        f = h.forward_windows(xx, 250, 125)
        (f * w).sum().backward()
        outs.append((f.detach().cpu(), {k: q.grad.detach().cpu().clone() for k, q in h.named_parameters()}))
    (f32, g32), (f16, g16) = outs
    worst = max((rel_err(g16[k], g32[k]), k) for k in g32)
    print(f"raw-EEG fused pair B={B} T={T}: features %.2e, worst gradient %.2e ({worst[1]})" % (rel_err(f16, f32), worst[0]))

# FAST(small_config) with bf16 zone-CNN activations vs G16 (reference under autocast) and G5 (reference fp32)
import test_cnn_gpu as tc
g5, g16 = load_golden("g5_fast_small.npz"), load_golden("g16_fast_small_autocast.npz")
got = tc._bf16_small_run(inn, g5)
for name, ref in (("autocast", g16), ("fp32", g5)):
    print(f"FAST small bf16 vs reference {name}: features %.2e" % rel_err(got["features"], ref["features"]))
    for mode in ("train_head", "default"):
        worst = max((rel_err(got[k], ref[k]), k) for k in ref.files if k.startswith(f"{mode}.grad."))
        print(f"    {mode}: logits %.2e loss %.2e worst gradient %.2e ({worst[1]})" % (
            rel_err(got[f"{mode}.logits"], ref[f"{mode}.logits"]), abs(got[f"{mode}.loss"] - float(ref[f"{mode}.loss"])), worst[0]))

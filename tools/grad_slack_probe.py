"""Where the gradient tolerances above 1e-4 come from: the HIP gradients of G5 (FAST small,
'train_head' and 'default') and G14 (cfg5 composed) against the goldens, per tensor, worst first.  The goldens' own fp32
error against an fp64 run of the reference is 5e-7 (measured in the build container), so what is listed is HIP's."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden, rel_err
import isd_amd, isd_amd.nn as inn
import test_cnn_gpu as tc

g = load_golden("g5_fast_small.npz")
m = inn.FAST(tc._small_cfg(inn)).cuda()
m.load_state_dict(tc._sd(g, "sd."))
x = torch.from_numpy(g["x"]).cuda()
lt = m.token_logits(x)
inn.token_mean_cross_entropy(lt, torch.from_numpy(g["labels"]).cuda()).backward()
w = sorted(((rel_err(p.grad.cpu(), g[f"train_head.grad.{k}"]), k) for k, p in m.named_parameters() if f"train_head.grad.{k}" in g.files), reverse=True)
print("G5 train_head worst:", [(f"{e:.2e}", k) for e, k in w[:6]])
m.zero_grad(set_to_none=True)
torch.nn.functional.cross_entropy(m(x), torch.from_numpy(g["labels"]).long().cuda()).backward()
w = sorted(((rel_err(p.grad.cpu(), g[f"default.grad.{k}"]), k) for k, p in m.named_parameters()), reverse=True)
print("G5 default worst:", [(f"{e:.2e}", k) for e, k in w[:8]])
# absolute scale of the worst tensors
for e, k in w[:4]:
    gg = g[f"default.grad.{k}"]
    print(f"   {k}: |grad|max {np.abs(gg).max():.3e}, all-grad max {max(np.abs(g[q]).max() for q in g.files if q.startswith('default.grad.')):.3e}")

# G14 (cfg5 composed): the golden's gradients come from scipy's features; how much of the 1e-3 is the features' 1e-4?
import test_cfg5_gpu as t5
from isd_amd.classifier import _EEGNetFeatureModel
from oracle import dsp as odsp
g = load_golden("g14_cfg5_composed.npz")
B = int(g["cfg"][0])
X = np.random.default_rng(14).standard_normal((B, t5.C, t5.T)).astype(np.float32)
fx = isd_amd.FeatureExtractor(t5.T, t5.FS, isd_amd.BANDS_40, nperseg=t5.NPERSEG, noverlap=t5.NOVERLAP)
feat_hip = fx(torch.from_numpy(X).cuda())
feat_ref = torch.from_numpy(odsp.extract_features_scipy(X, fs=t5.FS, bands=odsp.BANDS_40, nperseg=t5.NPERSEG,
                                                        noverlap=t5.NOVERLAP)).cuda()
print("features hip vs scipy: max |d| %.2e" % float((feat_hip - feat_ref).abs().max()))
scale = max(float(np.abs(g[k]).max()) for k in g.files if k.startswith("enc.grad."))
for name, feat in (("HIP features", feat_hip), ("scipy features", feat_ref)):
    m = _EEGNetFeatureModel(40 * t5.C, 32, 5, dropout=0.0).cuda()
    m.net.enc.load_state_dict(t5._sd(g, "enc.sd."))
    m.net.fc.load_state_dict(t5._sd(g, "fc.sd."))
    out = m.make_path().forward(feat.contiguous().view(B, 40 * t5.C, 65), torch.from_numpy(g["labels"]).cuda(), want_grad=True)
    errs = []
    for k, p in m.net.enc.named_parameters():
        want = g[f"enc.grad.{k}"]
        floor = 5e-2 if k.startswith("temporal_conv.1.") else 1e-3
        errs.append((float(np.abs(p.grad.cpu().numpy() - want).max() / max(float(np.abs(want).max()), floor * scale)), k))
    errs.sort(reverse=True)
    print(f"G14 with {name}: logits {rel_err(out['logits'].cpu(), g['logits']):.2e}; worst grads", [(f"{e:.2e}", k) for e, k in errs[:4]],
          "fc.weight %.2e fc.bias %.2e" % (rel_err(m.net.fc.weight.grad.cpu(), g["fc.grad.weight"]), rel_err(m.net.fc.bias.grad.cpu(), g["fc.grad.bias"])))

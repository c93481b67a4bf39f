"""Oracle (test infrastructure): the reference's CNN / FC-head / loss path on CPU.

A from-scratch restatement in functional PyTorch (CPU, fp32 or fp64) of

* ``Conv4Layers``      src/fast/models/fast.py:103-119
* ``EEGNet_Encoder``   src/fast/models/fast.py:122-167
* ``CVBlock``          src/fast/models/fast.py:32-100
* ``HeadConv_Paper_Version``  src/fast/models/fast.py:170-196
* ``Head``             src/fast/models/fast.py:199-210   (zone gather + stack)
* ``forward_head``     src/fast/models/fast.py:242-252   (unfold windows)
* ``train_head`` mode  src/fast/models/fast.py:273-278   (FC 256->32 GELU, 32->5, mean)
* ``AttentionBlock`` / ``forward_transformer`` / 'default' mode   src/fast/models/fast.py:10-29,260-272
* CE loss / argmax     src/fast/train/trainer.py:37,59,89
* ``cosine_scheduler`` src/fast/train/trainer.py:15-27 and the LambdaLR quirk :52
* dataset constants    src/fast/data/preprocess.py:20-42 (label / zone order)

Parameters are passed as plain dicts keyed with the reference's state_dict
names (``head.encoders.<Zone>.cnn1.weight`` ...), so golden state dicts
captured from the real reference load directly.  Gradients come from autograd
over these functional ops.  Pinned by tests/golden/g4..g11 (captured from the
importable reference by tests/golden/make_golden.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# src/fast/data/preprocess.py:20
CLASSES = ["hello", "help-me", "stop", "thank-you", "yes"]
# src/fast/data/preprocess.py:24-30
ELECTRODES = [
    "Fp1", "Fp2", "F7", "F3", "Fz", "F4", "F8", "FC5", "FC1", "FC2", "FC6", "T7", "C3", "Cz", "C4",
    "T8", "TP9", "CP5", "CP1", "CP2", "CP6", "TP10", "P7", "P3", "Pz", "P4", "P8", "PO9", "O1", "Oz",
    "O2", "PO10", "AF7", "AF3", "AF4", "AF8", "F5", "F1", "F2", "F6", "FT9", "FT7", "FC3", "FC4", "FT8",
    "FT10", "C5", "C1", "C2", "C6", "TP7", "CP3", "CPz", "CP4", "TP8", "P5", "P1", "P2", "P6", "PO7",
    "PO3", "POz", "PO4", "PO8",
]
# src/fast/data/preprocess.py:33-42 (dict order = zone order)
ZONES = {
    "Pre-frontal": ["AF7", "Fp1", "Fp2", "AF8", "AF3", "AF4"],
    "Frontal": ["F7", "F5", "F3", "F1", "Fz", "F2", "F4", "F6", "F8"],
    "Pre-central": ["FC1", "FC2", "FC3", "FC4", "FC5", "FC6"],
    "Central": ["C1", "C2", "C3", "Cz", "C4", "C5", "C6"],
    "Post-central": ["CP1", "CP2", "CP3", "CPz", "CP4", "CP5", "CP6"],
    "Temporal": ["T7", "T8", "FT7", "FT8", "TP7", "TP8", "TP9", "TP10", "FT9", "FT10"],
    "Parietal": ["P1", "P2", "P3", "P4", "Pz", "P5", "P6", "P7", "P8", "PO3", "PO4", "PO7", "PO8",
                 "PO9", "PO10"],
    "Occipital": ["O1", "O2", "Oz", "POz"],
}


def zone_index_lists(electrodes=None, zones=None):
    """fast.py:206: ``[electrodes.index(ch) for ch in ch_names]`` per zone."""
    electrodes = ELECTRODES if electrodes is None else electrodes
    zones = ZONES if zones is None else zones
    return [[electrodes.index(ch) for ch in names] for names in zones.values()]


def conv4layers(x, p, prefix="", n_layers=4):
    """fast.py:111-119.  x [B', Cz, T] -> [B', F].

    ``n_layers=2`` is the build-defined "2-layer CNN" of BASELINE config 1
    (cnn1+cnn2 then GELU+mean).
    """
    h = x.unsqueeze(1)                                        # B 1 C T
    h = F.conv2d(h, p[prefix + "cnn1.weight"], p[prefix + "cnn1.bias"])
    h = F.conv2d(h, p[prefix + "cnn2.weight"])
    if n_layers == 4:
        h = F.conv2d(h, p[prefix + "cnn3.weight"], padding=(0, 2))
        h = F.conv2d(h, p[prefix + "cnn4.weight"], padding=(0, 2))
    h = F.gelu(h)                                             # exact erf GELU
    return h.mean(dim=-1).squeeze(-1)                         # 'B F 1 T -> B F'


def eegnet_encoder(x, p, prefix="", training=False, eps=1e-5, momentum=0.1,
                   kernel_length=64):
    """fast.py:161-167 with dropout disabled.  x [B', C, T] -> [B', feature_dim].

    In training mode batch statistics are used and the running buffers in
    ``p`` are updated in place like nn.BatchNorm2d does.
    """
    def bn(h, name):
        return F.batch_norm(h, p[name + ".running_mean"], p[name + ".running_var"],
                            p[name + ".weight"], p[name + ".bias"], training, momentum, eps)

    h = x.unsqueeze(1)
    h = F.conv2d(h, p[prefix + "temporal_conv.0.weight"], padding=(0, kernel_length // 2))
    h = bn(h, prefix + "temporal_conv.1")
    w = p[prefix + "spatial_conv.0.weight"]
    h = F.conv2d(h, w, groups=p[prefix + "temporal_conv.0.weight"].shape[0])
    h = F.elu(bn(h, prefix + "spatial_conv.1"))
    h = F.avg_pool2d(h, (1, 4))
    w = p[prefix + "separable_conv.0.weight"]
    h = F.conv2d(h, w, padding=(0, 8), groups=w.shape[0])
    h = F.conv2d(h, p[prefix + "separable_conv.1.weight"])
    h = F.elu(bn(h, prefix + "separable_conv.2"))
    h = F.avg_pool2d(h, (1, 8))
    h = h.mean(dim=(-2, -1))                                  # AdaptiveAvgPool(1,1)+Flatten
    return F.linear(h, p[prefix + "projector.2.weight"], p[prefix + "projector.2.bias"])


def _bn(h, p, name, training, momentum, eps):
    return F.batch_norm(h, p[name + ".running_mean"], p[name + ".running_var"], p[name + ".weight"],
                        p[name + ".bias"], training, momentum, eps)


def cvblock(x, p, prefix="", training=False, eps=1e-5, momentum=0.1):
    """fast.py:78-100 with dropout disabled.  x [B', C, T] (or [B', 1, C, T]) -> [B', dim_token].
    The projector's input width is fixed by the 250-sample dummy of fast.py:66-74, so T must pool to that width."""
    h = x.unsqueeze(1) if x.dim() == 3 else x
    w1 = p[prefix + "conv1.weight"]
    h = _bn(F.conv2d(h, w1, padding=(0, w1.shape[-1] // 2)), p, prefix + "bn1", training, momentum, eps)
    h = F.conv2d(h, p[prefix + "conv2.weight"], groups=w1.shape[0])
    h = F.avg_pool2d(F.elu(_bn(h, p, prefix + "bn2", training, momentum, eps)), (1, 8))
    w3 = p[prefix + "conv3.weight"]
    h = F.conv2d(h, w3, padding=(0, w3.shape[-1] // 2))
    h = F.avg_pool2d(F.elu(_bn(h, p, prefix + "bn3", training, momentum, eps)), (1, 2))
    return F.linear(h.flatten(start_dim=1), p[prefix + "projector.weight"], p[prefix + "projector.bias"])


def headconv_paper(x, p, prefix="", training=False, eps=1e-5, momentum=0.1):
    """fast.py:185-196.  x [B', C, T] -> [B', feature_dim]: (1,3) conv + spatial conv, then three (1,3) convs, each
    followed by BatchNorm -> exact GELU -> MaxPool(1,2); mean over the remaining time steps."""
    h = x.unsqueeze(1)
    h = F.conv2d(h, p[prefix + "cnn1_t.weight"], p[prefix + "cnn1_t.bias"])
    h = F.conv2d(h, p[prefix + "cnn1_s.weight"])
    h = F.max_pool2d(F.gelu(_bn(h, p, prefix + "norm1", training, momentum, eps)), (1, 2), stride=(1, 2))
    for i in (2, 3, 4):
        h = F.conv2d(h, p[prefix + f"cnn{i}.weight"])
        h = F.max_pool2d(F.gelu(_bn(h, p, prefix + f"norm{i}", training, momentum, eps)), (1, 2), stride=(1, 2))
    return h.mean(dim=-1).squeeze(-1)


def head_forward(xw, p, zone_names, zone_idx, encoder=conv4layers, **kw):
    """fast.py:209-210: stack over zones of encoder(x[:, idx]) -> [B', Z, F]."""
    outs = []
    for name, idx in zip(zone_names, zone_idx):
        idx_t = torch.as_tensor(idx, dtype=torch.long)
        outs.append(encoder(xw[:, idx_t], p, prefix=f"head.encoders.{name}.", **kw))
    return torch.stack(outs, dim=1)


def forward_head(x, p, zone_names, zone_idx, window_len=250, slide_step=125,
                 encoder=conv4layers, **kw):
    """fast.py:242-252.  x [B, C, T] -> features [B, N, Z, F]."""
    xw = x.unfold(-1, window_len, slide_step)                 # B C N T
    B, C, N, T = xw.shape
    xw = xw.permute(0, 2, 1, 3).reshape(B * N, C, T)          # '(B N) C T'
    feat = head_forward(xw, p, zone_names, zone_idx, encoder, **kw)
    return feat.reshape(B, N, feat.shape[1], feat.shape[2])


def train_head_logits(x, p, zone_names, zone_idx, window_len=250, slide_step=125, **kw):
    """fast.py:273-278 (``forward_mode='train_head'``).  -> logits [B, n_classes]."""
    feat = forward_head(x, p, zone_names, zone_idx, window_len, slide_step, **kw)
    B, N, Z, Fd = feat.shape
    tok = F.gelu(F.linear(feat.reshape(B, N, Z * Fd), p["input_layer.0.weight"],
                          p["input_layer.0.bias"]))
    return F.linear(tok, p["last_layer.weight"], p["last_layer.bias"]).mean(dim=1)


def attention_block(x, p, prefix, num_heads):
    """fast.py:10-29 (AttentionBlock, dropout disabled): pre-LN MHA + pre-LN MLP with residuals.  x [B, S, D]."""
    D = x.shape[-1]
    h = F.layer_norm(x, (D,), p[prefix + "layer_norm_1.weight"], p[prefix + "layer_norm_1.bias"], 1e-5)
    qkv = F.linear(h, p[prefix + "attn.in_proj_weight"], p[prefix + "attn.in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    B, S, _ = x.shape
    dh = D // num_heads

    def heads(t):
        return t.reshape(B, S, num_heads, dh).transpose(1, 2)                # B H S dh
    att = torch.softmax(heads(q) @ heads(k).transpose(-1, -2) / math.sqrt(dh), dim=-1)
    ctx = (att @ heads(v)).transpose(1, 2).reshape(B, S, D)
    x = x + F.linear(ctx, p[prefix + "attn.out_proj.weight"], p[prefix + "attn.out_proj.bias"])
    h = F.layer_norm(x, (D,), p[prefix + "layer_norm_2.weight"], p[prefix + "layer_norm_2.bias"], 1e-5)
    h = F.gelu(F.linear(h, p[prefix + "linear.0.weight"], p[prefix + "linear.0.bias"]))
    return x + F.linear(h, p[prefix + "linear.3.weight"], p[prefix + "linear.3.bias"])


def default_logits(x, p, zone_names, zone_idx, num_heads, num_layers, window_len=250, slide_step=125, **kw):
    """fast.py:260-272 (``forward_mode='default'``, dropout disabled): CNN head -> input_layer -> cls token +
    positional embedding -> ``num_layers`` AttentionBlocks -> last_layer on the cls token."""
    feat = forward_head(x, p, zone_names, zone_idx, window_len, slide_step, **kw)
    B, N, Z, Fd = feat.shape
    tok = F.gelu(F.linear(feat.reshape(B, N, Z * Fd), p["input_layer.0.weight"], p["input_layer.0.bias"]))
    tok = torch.cat((p["cls_token"].expand(B, -1, -1), tok), dim=1)
    tok = tok + p["pos_embedding"][:, :N + 1]
    for i in range(num_layers):
        tok = attention_block(tok, p, f"transformer.{i}.", num_heads)
    return F.linear(tok[:, 0], p["last_layer.weight"], p["last_layer.bias"])


def feature_cnn_logits(feats, p, n_layers=4):
    """Build-defined classifier over spec-S features:
    ``Conv4Layers(nb*C, F)`` on [B, nb*C, J] through the unmodified head
    contract (fast.py:207,210), then ``Linear(F, n_classes)``.
    """
    B = feats.shape[0]
    h = conv4layers(feats.reshape(B, -1, feats.shape[-1]), p, prefix="cnn.", n_layers=n_layers)
    return F.linear(h, p["fc.weight"], p["fc.bias"])


def cross_entropy(logits, y):
    """trainer.py:37,59: nn.CrossEntropyLoss() (mean); uint8 labels accepted."""
    return F.cross_entropy(logits, torch.as_tensor(y).long())


def predict(logits):
    """trainer.py:89: argmax over classes (ties -> lowest index)."""
    return torch.argmax(logits, dim=1)


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0,
                     start_warmup_value=0):
    """trainer.py:15-27 restated."""
    total = epochs * niter_per_ep
    wi = warmup_epochs * niter_per_ep
    warm = np.linspace(start_warmup_value, base_value, wi) if warmup_epochs > 0 else np.array([])
    n = total - wi
    i = np.arange(n)
    cos = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * i / n))
    out = np.concatenate((warm, cos))
    assert len(out) == total
    return out


def lr_multiplier(table, global_step):
    """trainer.py:52: LambdaLR indexes ``table[global_step - 1]`` (step 0 -> last)."""
    return float(table[global_step - 1])


def init_conv4_params(channels, dim=32, prefix="", seed=0, n_layers=4):
    """Random parameters with nn.Conv2d-like scale (for tests without goldens)."""
    g = torch.Generator().manual_seed(seed)

    def u(shape, fan_in):
        b = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * b
    p = {prefix + "cnn1.weight": u((dim, 1, 1, 5), 5), prefix + "cnn1.bias": u((dim,), 5),
         prefix + "cnn2.weight": u((dim, dim, channels, 1), dim * channels)}
    if n_layers == 4:
        p[prefix + "cnn3.weight"] = u((dim, dim, 1, 5), dim * 5)
        p[prefix + "cnn4.weight"] = u((dim, dim, 1, 5), dim * 5)
    return p

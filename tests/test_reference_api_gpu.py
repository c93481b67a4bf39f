"""The reference's own model tests (tests/test_model.py: shapes, dtypes, no NaN, self-consistency, gradient
presence) re-stated against the HIP modules.  Line numbers cite the reference test that makes the same assertion."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def inn():
    import isd_amd.nn as m
    assert torch.cuda.is_available()
    return m


def small_config(inn, **kw):
    # tests/conftest.py:32-54
    electrodes = ["Fp1", "Fp2", "F3", "F4", "C3", "C4", "O1", "O2"]
    zones = {"Frontal": ["Fp1", "Fp2", "F3", "F4"], "Central": ["C3", "C4"], "Occipital": ["O1", "O2"]}
    d = dict(dim_cnn=16, dim_token=16, seq_len=500, n_classes=3, num_layers=1, num_heads=4, dropout=0.0)
    d.update(kw)
    return inn.fast_config(electrodes, zones, **d)


@pytest.mark.parametrize("mode", ["default", "train_head", "train_transformer"])
def test_forward_modes_shape_dtype_finite(inn, mode):
    # test_model.py:39-61 (production config, dummy_eeg_batch = randn(4, 64, 800))
    torch.manual_seed(0)
    m = inn.FAST(inn.fast_config()).cuda().eval()
    with torch.no_grad():
        out = m(torch.randn(4, 64, 800, device="cuda"), forward_mode=mode)
    assert out.shape == (4, 5) and out.dtype == torch.float32 and torch.isfinite(out).all()
    assert m.name == "FAST"


def test_unknown_mode_and_token_counts(inn):
    # test_model.py:63-67, :119-127
    m = inn.FAST(small_config(inn)).cuda()
    with pytest.raises(NotImplementedError):
        m(torch.randn(2, 8, 500, device="cuda"), forward_mode="bogus")
    assert inn.FAST(inn.fast_config()).n_tokens == 5 and m.n_tokens == 3
    assert m.pos_embedding.shape == (1, 4, 16) and m.cls_token.shape == (1, 1, 16)


def test_forward_head_and_batched_forward_head(inn):
    # test_model.py:180-190, :202-211 (the reference's only numeric equality)
    torch.manual_seed(1)
    m = inn.FAST(small_config(inn)).cuda().eval()
    x = torch.randn(5, 8, 500, device="cuda")
    with torch.no_grad():
        f = m.forward_head(x)
        fb = m.batched_forward_head(x, m.config.slide_step, 2)
    assert f.shape == (5, 3, 3, 16) and torch.equal(f, fb)


def test_every_parameter_gets_a_gradient_and_head_is_frozen_in_train_transformer(inn):
    # test_model.py:144-153, :155-164
    torch.manual_seed(2)
    m = inn.FAST(small_config(inn)).cuda().train()
    x = torch.randn(2, 8, 500, device="cuda")
    m(x).sum().backward()
    missing = [k for k, p in m.named_parameters() if p.grad is None]
    assert not missing, missing
    m.zero_grad(set_to_none=True)
    m(x, forward_mode="train_transformer").sum().backward()
    assert all(p.grad is None for k, p in m.named_parameters() if k.startswith("head."))
    assert m.last_layer.weight.grad is not None


@pytest.mark.parametrize("T", [250, 500, 125])
@pytest.mark.parametrize("dim", [16, 32])
def test_conv4layers_output_shape(inn, T, dim):
    # test_model.py:253-275; long windows take the one-item-per-workgroup tiles, forward and backward
    m = inn.Conv4Layers(6, dim).cuda()
    y = m(torch.randn(3, 6, T, device="cuda"))
    assert y.shape == (3, dim) and torch.isfinite(y).all()
    y.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def test_conv4layers_long_window_vs_oracle(inn):
    from oracle import cnn as ocnn
    from conftest import rel_err
    torch.manual_seed(4)
    m = inn.Conv4Layers(7, 32).cuda()
    x = torch.randn(2, 7, 560)
    y = m(x.cuda())
    y.square().sum().backward()
    p = {k: v.detach().cpu().double().requires_grad_() for k, v in m.state_dict().items()}
    ref = ocnn.conv4layers(x.double(), p)
    ref.square().sum().backward()
    assert rel_err(y.detach().cpu(), ref.detach()) < 1e-4
    for k, q in m.named_parameters():
        assert rel_err(q.grad.cpu(), p[k].grad) < 1e-4, k


@pytest.mark.parametrize("C,T,dim,n_layers", [(7, 1500, 32, 4), (64, 800, 32, 4), (3, 4096, 16, 2), (40, 2100, 32, 4)])
def test_conv4layers_very_long_windows_vs_oracle(inn, C, T, dim, n_layers):
    """Rows beyond the LDS tile: fewer channels per forward chunk, time segments in the weight gradient."""
    from oracle import cnn as ocnn
    from conftest import rel_err
    torch.manual_seed(C + T)
    m = inn.Conv4Layers(C, dim, n_layers).cuda()
    x = torch.randn(2, C, T)
    w = torch.randn(2, dim)
    y = m(x.cuda())
    (y * w.cuda()).sum().backward()
    p = {k: v.detach().cpu().double().requires_grad_() for k, v in m.state_dict().items()}
    ref = ocnn.conv4layers(x.double(), p, n_layers=n_layers)
    (ref * w.double()).sum().backward()
    assert rel_err(y.detach().cpu(), ref.detach()) < 1e-4
    for k, q in m.named_parameters():
        assert rel_err(q.grad.cpu(), p[k].grad) < 1e-4, k


@pytest.mark.parametrize("K", [64, 32])
@pytest.mark.parametrize("T", [250, 128])
def test_eegnet_encoder_output_shape(inn, K, T):
    # test_model.py:277-321
    m = inn.EEGNet_Encoder(6, 32, kernel_length=K).cuda().eval()
    with torch.no_grad():
        y = m(torch.randn(3, 6, T, device="cuda"))
    assert y.shape == (3, 32) and torch.isfinite(y).all()


def test_cvblock_and_paper_head_output_shapes(inn):
    # test_model.py:323-359
    x = torch.randn(4, 9, 250, device="cuda")
    cv = inn.CVBlock(9, 24).cuda().eval()
    ph = inn.HeadConv_Paper_Version(9, 32).cuda().eval()
    with torch.no_grad():
        a, b, c = cv(x), cv(x.unsqueeze(1)), ph(x)
    assert a.shape == (4, 24) and torch.equal(a, b) and c.shape == (4, 32)
    assert torch.isfinite(a).all() and torch.isfinite(c).all()


def test_head_zone_tables(inn):
    # test_model.py:374-404: index_dict values and the per-zone spatial kernel height
    import isd_amd
    h = inn.Head("Conv4Layers", isd_amd.ELECTRODES, isd_amd.ZONES, 32)
    for zone, names in isd_amd.ZONES.items():
        assert h.index_dict[zone].tolist() == [isd_amd.ELECTRODES.index(n) for n in names]
        assert h.encoders[zone].cnn2.weight.shape == (32, 32, len(names), 1)
    sizes = [len(v) for v in isd_amd.ZONES.values()]
    assert sizes == [6, 9, 6, 7, 7, 10, 15, 4] and sum(sizes) == 64 and len(isd_amd.CLASSES) == 5

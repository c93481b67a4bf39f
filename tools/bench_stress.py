"""Stress configuration 5 (128 ch, 4 s @ 1024 Hz, 40 bands, 1024-pt STFT, EEGNet head): per-stage times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isd_amd
import isd_amd.nn as inn


def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    C, T, fs = 128, 4096, 1024.0
    x = torch.randn(B, C, T, device="cuda")
    fx = isd_amd.FeatureExtractor(T, fs, isd_amd.BANDS_40, nperseg=1024, noverlap=960)
    nb = fx.n_bands
    y = torch.empty(B, nb, C, T, device="cuda")
    feat = torch.empty(B, nb, C, fx.n_frames, device="cuda")
    t_fb = t(lambda: fx.fb.forward(x, out=y))
    by = (1 + nb) * C * T * 4 * B
    print(f"B={B}: filterbank[{fx.fb.precision}] {t_fb:.2f} ms = {by / t_fb / 1e6:.0f} GB/s ({by / B / 1e6:.1f} MB/trial)")
    t_bp = t(lambda: fx.stft.bandpower(y, fx.bins, out=feat))
    rd = B * nb * C * T * 4
    print(f"       STFT 1024/960 band log-power (block-sum kernel, J={fx.n_frames}) {t_bp:.2f} ms = {rd / t_bp / 1e6:.0f} GB/s read")
    t_fu = t(lambda: fx(x, fused=True, out=feat))
    print(f"       fused filterbank -> block sums -> band log-power (one kernel per precision set) {t_fu:.2f} ms "
          f"(materialising pair: {t_fb + t_bp:.2f} ms)")
    m = inn.EEGNet_Encoder(nb * C, 32, dropout=0.25).cuda().train()
    f2 = feat.view(B, nb * C, fx.n_frames)
    def step():
        m.zero_grad(set_to_none=True)
        m(f2).square().mean().backward()
    t_e = t(step)
    print(f"       EEGNet_Encoder({nb * C}, 32) fwd+bwd on features [{B},{nb * C},{fx.n_frames}] {t_e:.2f} ms")
    m2 = inn.EEGNet_Encoder(C, 32, dropout=0.25).cuda().train()
    def step2():
        m2.zero_grad(set_to_none=True)
        m2(x).square().mean().backward()
    t_r = t(step2)
    print(f"       EEGNet_Encoder({C}, 32) fwd+bwd on raw EEG [{B},{C},{T}] {t_r:.2f} ms = {B / t_r * 1e3:.0f} trials/s "
          f"(x read = {x.numel() * 4 / 1e9:.2f} GB)")
    print(f"       end-to-end (features path) ~ {B / (t_fu + t_e) * 1e3:.0f} trials/s fused, "
          f"{B / (t_fb + t_bp + t_e) * 1e3:.0f} materialising")

if __name__ == "__main__":
    main()

"""Micro-benchmark of the feature-extraction kernels (device-resident input, torch events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isd_amd


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    B, C = 4096, 64
    for name, bands, prec, T, fs in (("9band/f32 T=512", isd_amd.BANDS_9, "f32", 512, 256.0),
                                     ("9band/f64 T=512", isd_amd.BANDS_9, "f64", 512, 256.0),
                                     ("5band/auto T=512", isd_amd.BANDS_5, "auto", 512, 256.0),
                                     # the reference-native trial (src/fast/data/preprocess.py:62)
                                     ("9band/f32 T=800 @250 Hz", isd_amd.BANDS_9, "f32", 800, 250.0)):
        x = torch.randn(B, C, T, device="cuda")
        fx = isd_amd.FeatureExtractor(T, fs, bands, precision=prec)
        nb = fx.n_bands
        y = torch.empty(B, nb, C, T, device="cuda")
        out = torch.empty(B, nb, C, fx.n_frames, device="cuda")
        t_fb = timeit(lambda: fx.fb.forward(x, out=y))
        t_bp = timeit(lambda: fx.stft.bandpower(y, fx.bins, out=out))
        t_fu = timeit(lambda: fx(x, fused=True, out=out))
        by = (1 + nb) * C * T * 4 * B
        print(f"{name}: fb {t_fb:.3f} ms ({by / t_fb / 1e6:.0f} GB/s, {B / t_fb * 1e3:.0f} trials/s) | "
              f"bandpower {t_bp:.3f} ms | fused {t_fu:.3f} ms ({B / t_fu * 1e3:.0f} trials/s)")


if __name__ == "__main__":
    main()

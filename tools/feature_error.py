"""Max log-domain error of the HIP feature extractor vs the fp64 oracle, per band set and precision (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import isd_amd
from oracle import dsp

def main():
    for name, bands, fs, T in (("9-band", isd_amd.BANDS_9, 256.0, 512), ("5-band", isd_amd.BANDS_5, 250.0, 500),
                               ("9-band T=250", isd_amd.BANDS_9, 250.0, 250)):
        X, _ = dsp.synth_trials(6, 64, T, fs, seed=3)
        ref = dsp.extract_features_scipy(X, fs=fs, bands=bands)
        for prec in ("f32", "f64", "auto"):
            fx = isd_amd.FeatureExtractor(T, fs, bands, precision=prec)
            x = torch.from_numpy(X).cuda()
            for fused in (True, False):
                got = fx(x, fused=fused).cpu().numpy()
                err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
                print(f"{name:14s} precision={prec:4s} fused={fused!s:5s} max|dlog|={err.max():.3e} mean={err.mean():.3e}")

if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def worst(T=250, fs=250.0, bands=isd_amd.BANDS_9, prec="f32"):
    """Where the largest log-domain differences sit (band, frame, reference value)."""
    X, _ = dsp.synth_trials(6, 64, T, fs, seed=3)
    ref = dsp.extract_features_scipy(X, fs=fs, bands=bands).astype(np.float64)
    got = isd_amd.FeatureExtractor(T, fs, bands, precision=prec)(torch.from_numpy(X).cuda(), fused=True).cpu().numpy()
    err = np.abs(got - ref)
    for flat in np.argsort(err.ravel())[::-1][:8]:
        b, band, ch, j = np.unravel_index(flat, err.shape)
        print(f"trial {b} band {band} ch {ch} frame {j}/{err.shape[-1]}: ref log P = {ref[b, band, ch, j]:.4f} "
              f"(typical {np.median(ref[b, band, ch]):.2f}), |dlog| = {err[b, band, ch, j]:.2e}")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "worst":
    worst()
    worst(512, 256.0)

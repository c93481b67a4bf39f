"""Which bands of the 40-band stress set need the fp64 in-chunk recursion?  For several thresholds on the plan's
round-off amplification estimate (ISD_FB_AUTO_LIMIT): how many bands run in fp64, the worst feature error against the
scipy fp64 oracle in the test's metric (|got - ref| / max(1, |ref|), gate 1e-4) per band, and the extraction time."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch, time
    import isd_amd
    from oracle import dsp as odsp
    fs, T, C = 1024.0, 4096, 128
    X, _ = odsp.synth_trials(2, 16, T, fs, seed=4)
    ref = np.load(sys.argv[2])["ref"] if os.path.exists(sys.argv[2]) else None
    if ref is None:
        ref = odsp.extract_features_scipy(X, fs=fs, bands=odsp.BANDS_40, nperseg=1024, noverlap=960).astype(np.float64)
        np.savez(sys.argv[2], ref=ref)
    fx = isd_amd.FeatureExtractor(T, fs, isd_amd.BANDS_40, nperseg=1024, noverlap=960)
    got = fx(torch.from_numpy(X).cuda()).cpu().numpy().astype(np.float64)
    err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    per_band = err.max(axis=(0, 2, 3))
    xb = torch.randn(128, C, T, device="cuda")
    out = torch.empty(128, 40, C, fx.n_frames, device="cuda")
    for _ in range(2):
        fx(xb, fused=True, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fx(xb, fused=True, out=out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(json.dumps({"precision": fx.fb.precision, "max_err": float(err.max()), "ms_per_128": round(ms, 3),
                      "bands_over_5e-5": [int(b) for b in np.where(per_band > 5e-5)[0]],
                      "per_band_first20": [float(f"{v:.2e}") for v in per_band[:20]]}))
    sys.exit(0)
ref_path = "/tmp/auto_limit_ref.npz"
for lim in ("2000", "4000", "8000", "16000", "1e9"):
    env = dict(os.environ, ISD_FB_AUTO_LIMIT=lim)
    r = subprocess.run([sys.executable, __file__, "child", ref_path], env=env, capture_output=True, text=True)
    print("limit", lim, r.stdout.strip() or r.stderr[-400:])

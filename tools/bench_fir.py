"""Zero-phase FIR band-pass (csrc/fir.hip): time per launch, FMA rate and HBM rate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isd_amd


def run(name, flt, x, n=20):
    y = torch.empty_like(x)
    for _ in range(3):
        flt(x, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        flt(x, out=y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    taps = len(flt.taps)
    flop = 2.0 * taps * x.numel()
    by = 2.0 * x.numel() * x.element_size()
    print(f"{name}: {ms:.3f} ms  {flop / ms / 1e9:.1f} TFLOP/s ({taps} taps)  {by / ms / 1e6:.0f} GB/s  "
          f"{x.shape[0] / ms * 1e3:.0f} trials/s")


def main():
    torch.manual_seed(0)
    run("fp32 [4096,64,512] 256 Hz 4-40", isd_amd.FirFilter(256, 4, 40), torch.randn(4096, 64, 512, device="cuda"))
    run("fp32 [4096,64,800] 250 Hz 4-40", isd_amd.FirFilter(250, 4, 40), torch.randn(4096, 64, 800, device="cuda"))
    run("fp64 [350,64,795] 250 Hz 4-40 (notebook)", isd_amd.FirFilter(250, 4, 40),
        torch.randn(350, 64, 795, device="cuda", dtype=torch.float64))
    run("fp64 [4096,64,512] 256 Hz 4-40", isd_amd.FirFilter(256, 4, 40),
        torch.randn(4096, 64, 512, device="cuda", dtype=torch.float64))


if __name__ == "__main__":
    main()

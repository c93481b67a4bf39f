#!/usr/bin/env python3
"""Compile one csrc/*.hip for gfx950 with -Rpass-analysis=kernel-resource-usage and print one line per kernel
(VGPRs, waves per SIMD, SGPR / VGPR spills, LDS).  `python tools/kernel_resources.py fb` (no GPU needed)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "fb"
    src = os.path.join(ROOT, "imagined-speech-decoding_amd", "csrc", name + ".hip")
    keep = sys.argv[2] if len(sys.argv) > 2 else None                 # directory that receives the .s (optional)
    with tempfile.TemporaryDirectory() as tmp:
        out = keep or tmp
        os.makedirs(out, exist_ok=True)
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o",
                            os.path.join(out, name + ".o"), "-save-temps", "-Rpass-analysis=kernel-resource-usage"],
                           cwd=out, stderr=subprocess.PIPE, text=True)
        if r.returncode:
            sys.exit(r.stderr)
        for b in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
            kern = subprocess.run(["c++filt", b.split(" ")[0]], stdout=subprocess.PIPE,
                                  text=True).stdout.strip().split("(")[0]
            g = lambda k: re.search(k + r": (\d+)", b).group(1)      # noqa: E731
            print("%-58s VGPR %4s  waves/SIMD %s  SGPR spill %4s  VGPR spill %3s  LDS %6s" % (
                kern[-58:], g("VGPRs"), g(r"Occupancy \[waves/SIMD\]"), g("SGPRs Spill"), g("VGPRs Spill"),
                g(r"LDS Size \[bytes/block\]")))


if __name__ == "__main__":
    main()

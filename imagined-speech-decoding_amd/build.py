"""Build libisd_hip.so (hipcc, gfx950 only) in-tree next to this file."""
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libisd_hip.so")
ARCH = "gfx950"


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _deps():
    inc = os.path.join(os.path.dirname(PKG_DIR), "include", "isd_hip.h")
    return sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [inc]


def is_fresh():
    if not os.path.exists(LIB_PATH):
        return False
    t = os.path.getmtime(LIB_PATH)
    return all(os.path.getmtime(p) <= t for p in _deps())


def hipcc_path():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into one shared library.  Returns its path."""
    if not force and is_fresh():
        return LIB_PATH
    objs = []
    jobs = []
    os.makedirs(os.path.join(PKG_DIR, "build"), exist_ok=True)
    for src in sources():
        obj = os.path.join(PKG_DIR, "build", os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [hipcc_path(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj]
        cmd += os.environ.get("ISD_HIPCC_FLAGS", "").split()          # e.g. -DISD_CF_TIMING (tools/conv_phases.py)
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        jobs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in jobs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
    cmd = [hipcc_path(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH + ".tmp"] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc link failed:\n{r.stdout}")
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

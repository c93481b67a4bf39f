"""Phase timing of the bf16 fused zone-CNN forward (conv4_fused_fwd_bf16_kernel): shader-clock stamps of wave 0 of
one workgroup on its 5th item.  Needs a library built with the stamps compiled in:

    ISD_HIPCC_FLAGS=-DISD_CF_TIMING python -m isd_amd.build --force && python tools/conv_phases.py
    python -m isd_amd.build --force        # back to the product build afterwards
"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isd_amd
from isd_amd import _lib
from isd_amd.classifier import _FastModel
from isd_amd.nn import fast_config

torch.manual_seed(0)
B = int(os.environ.get("ISD_PROF_B", "4096"))
m = _FastModel(fast_config(seq_len=512, act_dtype="bf16")).cuda()
tr = isd_amd.Trainer(m)
x = torch.randn(B, 64, 512, device="cuda")
y = torch.randint(0, 5, (B,), device="cuda")
for _ in range(3):
    tr.step(x, y)
torch.cuda.synchronize()
h = C.CDLL(_lib.LIB_PATH)
t = (C.c_longlong * 32)()
assert h.isd_debug_conv_marks(t) == 0
names = ["x registers -> bf16 xt tile", "barrier", "fetch next item's x", "cnn1.cnn2 mfma", "store t2 (+bias)",
         "barrier", "guards + a2 copy-out issue", "cnn3 mfma", "store t3", "barrier", "a3 copy-out issue", "cnn4 mfma",
         "gelu / gelu' + store", "row sums", "barrier", "feat + a4 copy-out issue"]
for k, n in enumerate(names):
    print(f"{n:36s} {t[k + 1] - t[k]:8d} clk")
print(f"{'item total':36s} {t[16] - t[0]:8d} clk (shader clock cycles)")
bnames = ["wait for the fetched tiles + barrier", "x registers -> bf16 xt tile", "G4 = dfeat/T1 * GELU'(A4)",
          "fetch next item (DMA + registers)", "barrier", "dW4 (G4 x A3)", "G3 = W4^T * G4, store", "barrier",
          "G2 = W3^T * G3, store", "barrier", "dW3 (G3 x A2)", "dWeff, dbeff (G2 x x)"]
print("backward (conv4_fused_bwd_bf16_kernel):")
for k, n in enumerate(bnames):
    print(f"{n:36s} {t[18 + k] - t[17 + k]:8d} clk")
print(f"{'item total (without the loop edge)':36s} {t[29] - t[17]:8d} clk")

"""Driver for a counter pass over the self-attention forms: 4 packed launches each of the by-shape default and of k_sattn_h."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uce_amd import edit as E
B, L, Hh, dh = int(os.environ.get("SB", 32)), int(os.environ.get("SL", 4096)), 8, 40
qkv = torch.randn(B, L, 3 * Hh * dh, device="cuda").bfloat16()
for qt in os.environ.get("QTS", "0,4").split(","):
    os.environ["UCE_SATTN_QT"] = qt
    h = E.UceHandle("cuda:0")
    for _ in range(4):
        h.sattn_packed(qkv, Hh)
    torch.cuda.synchronize()

"""Workload for the PMC passes: the five conv weight-gradient launches of one training step (B=64, 128x512), through the
C ABI entry lxo_conv3x3_wgrad on random bf16 operands."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latex_ocr_amd import _abi
lib = _abi.load()
B = 64
c = lambda n: -(-n // 2)
H1, W1 = 64, 256; H2, W2 = 32, 128; H4 = 16; W5 = 64
layers = [(H1, W1, 64, 128, 1), (H2, W2, 128, 256, 1), (H2, W2, 256, 256, 1), (H4, W2, 256, 512, 1), (H4, W5, 512, 512, 0)]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())
for h, w, ci, co, same in layers:
    ho, wo = (h, w) if same else (h - 2, w - 2)
    x = torch.randn(B, h, w, ci, dtype=torch.bfloat16, device="cuda")
    dy = torch.randn(B, ho, wo, co, dtype=torch.bfloat16, device="cuda")
    dw = torch.zeros(9 * ci, co, dtype=torch.float32, device="cuda")
    for _ in range(3):
        rc = lib.lxo_conv3x3_wgrad(_abi.LXO_BF16, p(x), p(dy), p(dw), B, h, w, ci, ho, wo, co, 1 if same else 0, st)
        assert rc == 0
torch.cuda.synchronize()
print("ok")

#!/usr/bin/env python
"""Measurement aid: the forward attention launch pair (attn_fwd part + combine) of the benchmark shape
(64 samples x 868 regions, E=256, C=512, bf16) in isolation, N back-to-back launches between two events.
The env switches of csrc/decoder_kernels.hip (LXO_ATT_PIPE, LXO_ATT_U) select the variant."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from latex_ocr_amd import _abi
L = _abi.load()
nv, R, E, C = 64, 868, 256, 512
g = torch.Generator().manual_seed(0)
att_img = (torch.randn(nv, R, E, generator=g)).to(torch.bfloat16).cuda()
img = (torch.randn(nv, R, C, generator=g)).to(torch.bfloat16).cuda()
att_h = torch.randn(nv, E, generator=g).cuda(); beta = (torch.randn(E, generator=g) * 0.3).cuda()
Rp = (R + 7) // 8 * 8
alpha = torch.zeros(nv, Rp, device="cuda"); part = torch.zeros(nv * 32 * (C + 2), device="cuda"); ctx = torch.zeros(nv, C, device="cuda")
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(n):
    for _ in range(n):
        rc = L.lxo_attention_fwd(1, p(att_img), p(img), p(att_h), p(beta), p(alpha), p(part), p(ctx), C, nv, R, E, C, 1, st)
        assert rc == 0
run(20); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(5):
    e0.record(); run(200); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
print("variant PIPE=%s ABL=%s U=%s: %.2f us per part+combine pair (back to back)" % (os.environ.get("LXO_ATT_PIPE", "1"), os.environ.get("LXO_ATT_ABL", "0"), os.environ.get("LXO_ATT_U", "8"), best))

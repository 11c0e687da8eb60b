import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from latex_ocr_amd import _abi
L = ctypes.CDLL(_abi.LIB_PATH)
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M = 64
for name, N, K in [("K1", 2048, 1024), ("K2", 256, 512), ("K4", 512, 1024), ("B1", 1024, 512), ("B3", 512, 256), ("B4", 1024, 2048)]:
    A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    KS = K // 128
    slab = torch.empty(KS, M, N, device="cuda")
    C = torch.empty(M, N, device="cuda")
    def run_slab():
        return L.lxo_gemm_slab(1, p(A), p(B), p(slab), M, N, K, K, K, N, ctypes.c_longlong(M * N), st)
    def run_skinny():
        return L.lxo_gemm_nt(1, 1, 1, 1, p(A), p(B), p(C), M, N, K, K, K, N, None, 0, ctypes.c_float(1.0), 0, st)
    for fn, tag in ((run_slab, "slab"), (run_skinny, "skinny")):
        for _ in range(10): assert fn() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): fn()
        e1.record(); e1.synchronize()
        print("%s N=%d K=%d %s: %.2f us/launch" % (name, N, K, tag, e0.elapsed_time(e1) * 1000 / 200))
    ref = A.to(torch.bfloat16).float() @ B.float().t()
    print("   err slab %.2e skinny %.2e" % ((slab.sum(0) - ref).abs().max().item() / ref.abs().max().item(), (C - ref).abs().max().item() / ref.abs().max().item()))

"""Workload for the PMC passes: the ten implicit-GEMM conv launches of one training step
(B=64, 128x512), via tools/conv_gemm_timing.py (each launch repeated 2 + reps times)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from latex_ocr_amd.engine import Engine
from conv_gemm_timing import time_conv_gemms
eng = Engine(500, dtype="bf16")
flops, secs, per = time_conv_gemms(eng, 64, 128, 512, reps=1)
print(flops, secs)

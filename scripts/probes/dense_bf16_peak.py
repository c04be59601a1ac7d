#!/usr/bin/env python3
"""What a plain dense bf16 GEMM (torch.matmul -> hipBLASLt) reaches on this box, and at which shader clock -- the practical
matrix-core ceiling under the chip's power limit, next to the 2500 TFLOP/s spec peak (2.4 GHz) that roofline.frac is priced against.
  python scripts/probes/dense_bf16_peak.py [M N K] [--reps R]      (events; under rocprofv3 --pmc GRBM_GUI_ACTIVE the summary
  of scripts/probes/dense_bf16_peak.sh gives the clock: cycles / duration)"""
import sys
import torch

args = [a for a in sys.argv[1:] if not a.startswith("--")]
M, N, K = (int(a) for a in args[:3]) if len(args) >= 3 else (8192, 8192, 8192)
reps = 20
dev = torch.device("cuda", 0)
a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
for _ in range(5):
    c = a @ b
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    c = a @ b
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"dense bf16 {M}x{N}x{K}: {ms * 1e3:.1f} us per GEMM, {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s ({2.0 * M * N * K / ms / 1e9 / 2500:.3f} of the 2500 TF spec peak)")

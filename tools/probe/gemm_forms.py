"""The three operand forms of the 256x256 GEMM on one ViT-sized problem each, a few launches (to be run under
rocprofv3 --pmc ...: LDS bank conflicts / LDS-array cycles per form).  Usage: python tools/probe/gemm_forms.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import youku_mplug_amd
from youku_mplug_amd import ops

dev = torch.device("cuda:0")
r = lambda *s: (torch.rand(*s, device=dev) * 2 - 1).bfloat16()
M, N, K = 50432, 2304, 768
a, w = r(M, K), r(N, K)
for _ in range(3):
    ops.gemm(a, w, M, N, K, tile_hint=256)                                  # <0,0>
dy, w2 = r(M, N), r(N, K)
for _ in range(3):
    ops.gemm(dy, w2, M, K, N, trans_b=True, tile_hint=256)                  # <0,1>: dX[M,K] = dY[M,N] W[N,K]
for _ in range(3):
    ops.gemm(dy, a, N, K, M, trans_a=True, trans_b=True, tile_hint=256)     # <1,1>: dW[N,K] = dY^T X
torch.cuda.synchronize()
print("ok")

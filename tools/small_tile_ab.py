"""Few-tile products of the step (the top decoder layer on the loss window, the abstractor): the 256x256 kernel (library's choice) against
the 128x128 kernel (tile_hint=128: 2 workgroups per CU, tail split along K), isolated loops.  MEASUREMENT TOOL.
    python tools/small_tile_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import youku_mplug_amd  # noqa: E402,F401
from youku_mplug_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
SHAPES = [  # (M, N, K, trans_a, trans_b, bias)
    (1024, 2048, 8192, False, False, True), (1024, 8192, 2048, False, False, True), (1024, 2048, 2048, False, False, True),
    (1024, 6144, 2048, False, False, True), (4096, 768, 3072, False, False, True), (4096, 3072, 768, False, False, True),
    (4096, 768, 768, False, False, True), (4096, 2048, 768, False, False, True), (4096, 768, 2048, False, True, False),
    (4096, 768, 3072, False, True, False), (3072, 768, 4096, True, True, False), (768, 3072, 4096, True, True, False),
    (2048, 768, 4096, True, True, False), (768, 768, 4096, True, True, False), (512, 2560, 10240, False, False, True),
    (2560, 2560, 2560, False, False, True),
]


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


print("| M | N | K | form | library's choice us | tile_hint=128 us | TF/s | TF/s |")
for M, N, K, ta, tb, bias in SHAPES:
    a = torch.randn((K, M) if ta else (M, K), device=dev).bfloat16()
    b = torch.randn((K, N) if tb else (N, K), device=dev).bfloat16()
    bi = torch.randn(N, device=dev).bfloat16() if bias else None
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    r = []
    for hint in (0, 128):
        r.append(timeit(lambda: ops.gemm(a, b, M, N, K, trans_a=ta, trans_b=tb, bias=bi, out=out, tile_hint=hint)))
    fl = 2.0 * M * N * K
    print(f"| {M} | {N} | {K} | <{int(ta)},{int(tb)}> | {r[0]:.1f} | {r[1]:.1f} | {fl / r[0] / 1e6:.0f} | {fl / r[1] / 1e6:.0f} |")

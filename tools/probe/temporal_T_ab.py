"""temporal attention at the three frame counts of the recipes (4: shipped pre-train YAML, 8: benchmark configuration, 16: retrieval), isolated.
Usage (GPU box): python tools/probe/temporal_T_ab.py   (MPV_LIB_PATH=<other build> for the other arm)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import youku_mplug_amd  # noqa: F401
from youku_mplug_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
def rnd(*s): return (torch.rand(*s, device=dev) * 2 - 1).bfloat16()
def timeit(fn, iters=10):
    fn(); fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
for B, T in ((48, 4), (32, 8), (96, 16)):
    N, heads, hd = 196, 8, 96
    N1, D = N + 1, heads * hd
    qkv = rnd(B * T * N1, 3 * D)
    out, dout, dqkv = torch.zeros(B * T * N1, D, dtype=torch.bfloat16, device=dev), rnd(B * T * N1, D), torch.zeros_like(qkv)
    tf = min(timeit(lambda: ops.temporal_attn_fwd(qkv, out, B, T * N1, N, 1, N1, T, heads, hd, hd ** -0.5)) for _ in range(2))
    tb = min(timeit(lambda: ops.temporal_attn_bwd(qkv, dout, dqkv, B, T * N1, N, 1, N1, T, heads, hd, hd ** -0.5)) for _ in range(2))
    by = B * T * N * D * 2
    print(f"{os.path.basename(os.environ.get('MPV_LIB_PATH', 'this build')):20s} B={B:3d} T={T:2d}: fwd {tf*1e6:7.1f} us ({4*by/tf/1e9:5.0f} GB/s)  bwd {tb*1e6:7.1f} us ({8*by/tb/1e9:5.0f} GB/s)  "
          f"checksums {out.float().abs().sum().item():.6e} {dqkv.float().abs().sum().item():.6e}", flush=True)

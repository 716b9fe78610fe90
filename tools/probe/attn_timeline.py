"""Phase timeline of attn_bwd_dkv_res_kernel from the instrumented build:
hipcc -DMPV_ATTN_TIMING ... ; MPV_LIB_PATH=.../libmpv_hip_timing.so python tools/probe/attn_timeline.py"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import youku_mplug_amd
from youku_mplug_amd import ops, _lib
from tools.bench_kernels import rnd, dev
B, H, Sq, Sk, hd = 256, 8, 197, 197, 96
q, k, v = rnd(B, Sq, H, hd), rnd(B, Sk, H, hd), rnd(B, Sk, H, hd)
o, do = torch.empty_like(q), rnd(B, Sq, H, hd)
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
lay = ops.AttnLayout((Sq * H * hd, hd, H * hd), (Sk * H * hd, hd, H * hd), (Sk * H * hd, hd, H * hd), (Sq * H * hd, hd, H * hd))
for _ in range(3):
    lse = ops.attn_fwd(q, k, v, o, lay, B, H, Sq, Sk, hd, scale=hd ** -0.5, scale_q_bf16=True)
    ops.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, lay, B, H, Sq, Sk, hd, scale=hd ** -0.5, scale_q_bf16=True)
torch.cuda.synchronize()
buf = (C.c_longlong * (4096 * 8))()
fn = _lib.lib().mpv_attn_read_timeline
fn.restype, fn.argtypes = C.c_int, [C.c_void_p]
assert fn(buf) == 0
t = np.frombuffer(buf, dtype=np.int64).reshape(4096, 8)[:2048].astype(np.float64)
names = ["issue DMA + K/V row loads", "wait DMA + barrier", "scale-q pass", "main loop", "stores"]
for i, n in enumerate(names):
    d = t[:, i + 1] - t[:, i]
    print(f"{n:28s} median {np.median(d):9.0f} clk   p90 {np.percentile(d, 90):9.0f}")
tot = t[:, 5] - t[:, 0]
print(f"workgroup lifetime median {np.median(tot):.0f} clk; first start -> last end {t[:,5].max() - t[:,0].min():.0f} clk")

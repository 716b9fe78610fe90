import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, youku_mplug_amd
from youku_mplug_amd import ops
from tools.bench_kernels import rnd, dev
B, H, Sq, Sk, hd = 256, 8, 197, 197, 96
q, k, v = rnd(B, Sq, H, hd), rnd(B, Sk, H, hd), rnd(B, Sk, H, hd)
o = torch.empty_like(q)
lay = ops.AttnLayout((Sq * H * hd, hd, H * hd), (Sk * H * hd, hd, H * hd), (Sk * H * hd, hd, H * hd), (Sq * H * hd, hd, H * hd))
for _ in range(3):
    ops.attn_fwd(q, k, v, o, lay, B, H, Sq, Sk, hd, scale=hd ** -0.5, scale_q_bf16=True)
torch.cuda.synchronize()

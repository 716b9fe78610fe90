import sys; sys.path.insert(0,'/root/repo')
import torch, youku_mplug_amd
from youku_mplug_amd import ops
dev=torch.device('cuda:0')
v=torch.randn(32,3,8,224,224,device=dev).bfloat16()
for _ in range(3): ops.im2col_patches(v,32,3,8,224,224,16,768)
torch.cuda.synchronize()
s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): ops.im2col_patches(v,32,3,8,224,224,16,768)
e.record(); torch.cuda.synchronize()
print("im2col us", s.elapsed_time(e)/20*1e3)

"""Which part of the step survives HIP-graph capture?  python tools/graph_debug.py <stage>  (run each stage in its own process)"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import youku_mplug_amd
from youku_mplug_amd import engine as eng, ops
from youku_mplug_amd.pretrain import synthetic_model
from oracle.weights import CONFIG_TINY, make_inputs

stage = sys.argv[1]
dev = torch.device("cuda:0")
model = synthetic_model(CONFIG_TINY, device=dev)
model.train()
groups = eng.get_parameter_groups(model, 0.05, model.no_weight_decay(), visual_backbone_scale=True)
e, opt, _, _ = eng.initialize(model=model, model_parameters=groups, config=dict(lr=2e-3, clip_grad=3.0))
video, ids, mask = make_inputs(CONFIG_TINY, 4, 12, seed=70, ragged=False)
v = video.to(dev).to(torch.bfloat16)
text = types.SimpleNamespace(input_ids=ids.to(dev), attention_mask=mask.to(dev))
e.enable_device_step_state()
for _ in range(2):
    loss, _ = e(v, text); e.backward(loss); e.step()
torch.cuda.synchronize()
print("eager ok", loss.item(), flush=True)
g = torch.cuda.CUDAGraph()
if stage == "gemm":
    a = torch.randn(256, 256, device=dev).to(torch.bfloat16); b = torch.randn(256, 256, device=dev).to(torch.bfloat16)
    ops.gemm(a, b, 256, 256, 256)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        c = ops.gemm(a, b, 256, 256, 256)
elif stage == "fwd":
    with torch.cuda.graph(g):
        with torch.no_grad():
            loss, _ = e(v, text)
elif stage == "fwd_grad":
    with torch.cuda.graph(g):
        loss, _ = e(v, text)
elif stage == "fwd_bwd":
    with torch.cuda.graph(g):
        loss, _ = e(v, text)
        loss.backward()
elif stage == "opt":
    with torch.cuda.graph(g):
        e.optimizer.step(grad_scale=1.0, upload=False)
elif stage == "all":
    with torch.cuda.graph(g):
        loss = model.forward_backward(v, text)
        e.optimizer.step(grad_scale=1.0, upload=False)
print("capture ok", stage, flush=True)
g.replay()
torch.cuda.synchronize()
print("replay ok", stage, flush=True)

"""Per-step losses and parameter digests of the config-B step in three launch modes (by-value eager / device-scalar eager / graph replay), same
weights, same inputs: where does a graph replay first differ from the eager step?   python tools/graph_vs_eager_configB.py [steps]"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import youku_mplug_amd  # noqa: F401
from youku_mplug_amd import engine as eng
from youku_mplug_amd.pretrain import synthetic_model
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
B, T, L = 32, 8, 32
S = bench.Shapes
torch.manual_seed(7)
video = torch.randn(B, 3, T, S.img_size, S.img_size, device=dev).to(torch.bfloat16)
ids = torch.randint(0, S.vocab, (B, L), device=dev)
text = types.SimpleNamespace(input_ids=ids, attention_mask=torch.ones(B, L, dtype=torch.long, device=dev))


def run(mode):
    torch.manual_seed(1234)
    model = synthetic_model(S, device=dev, num_frames=T)
    with torch.no_grad():
        for blk in model.visual_encoder.blocks:
            blk.temporal_fc.weight.normal_(0, 0.015)
        model.visual_encoder.temporal_embed.normal_(0, 0.015)
    model.train()
    groups = eng.get_parameter_groups(model, 0.05, model.no_weight_decay(), visual_backbone_scale=True)
    e, opt, _, _ = eng.initialize(model=model, model_parameters=groups, config=dict(lr=1e-4, opt_betas=(0.9, 0.999), opt_eps=1e-6, clip_grad=3.0))
    if mode == "device":
        e.enable_device_step_state()
    out = []
    for i in range(steps):
        for g in opt.param_groups:
            g["lr"] = 1e-4 * (i + 1) / 2000.0 * g["lr_scale"]
        if mode == "graph":
            loss = e.graph_step(video, text)
        else:
            loss, _ = e(video, text)
            e.backward(loss)
            e.step()
        torch.cuda.synchronize()
        out.append((loss.item(), e.flat.params.float().sum().item(), e.flat.grads.float().abs().sum().item(), model.text_decoder.step_seed))
    del e, opt, model
    torch.cuda.empty_cache()
    return out


res = {m: run(m) for m in ("value", "device", "graph")}
for i in range(steps):
    print(f"step {i}: " + " | ".join(f"{m}: loss {res[m][i][0]:.6f} sum(p) {res[m][i][1]:.4f} sum|g| {res[m][i][2]:.4f} next-seed {res[m][i][3] & 0xffff:04x}" for m in res))
print("value == device:", res["value"] == res["device"], " value == graph:", res["value"] == res["graph"])

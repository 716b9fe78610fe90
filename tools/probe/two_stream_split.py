"""Feasibility probe: the step as two half-batches on two HIP streams (forward + backward of each half enqueued
back-to-back; gradients race, only the timing is meaningful) against one full batch on one stream."""
import sys, os, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import youku_mplug_amd
from youku_mplug_amd import engine as eng
from youku_mplug_amd.pretrain import synthetic_model
import bench

dev = torch.device("cuda:0")
Shapes = bench.Shapes
Shapes.num_frames = 8
torch.manual_seed(1234)
model = synthetic_model(Shapes, device=dev, num_frames=8)
with torch.no_grad():
    for blk in model.visual_encoder.blocks:
        blk.temporal_fc.weight.normal_(0, 0.015)
    model.visual_encoder.temporal_embed.normal_(0, 0.015)
model.train()
groups = eng.get_parameter_groups(model, 0.05, model.no_weight_decay(), visual_backbone_scale=True)
engine, opt, _, _ = eng.initialize(model=model, model_parameters=groups, config=dict(lr=1e-4, opt_betas=(0.9, 0.999), opt_eps=1e-6, clip_grad=3.0))
B, T, L = 32, 8, 32
video = torch.randn(B, 3, T, 224, 224, device=dev).to(torch.bfloat16)
ids = torch.randint(0, Shapes.vocab, (B, L), device=dev)
mk = lambda a, b: types.SimpleNamespace(input_ids=ids[a:b].contiguous(), attention_mask=torch.ones(b - a, L, dtype=torch.long, device=dev))
full = (video, mk(0, B))
halves = [(video[:16].contiguous(), mk(0, 16)), (video[16:].contiguous(), mk(16, 32))]
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def step_full():
    loss, _ = engine(*full)
    engine.backward(loss)
    engine.step()


def step_split():
    main = torch.cuda.current_stream()
    losses = []
    for s, (v, t) in zip((sa, sb), halves):
        s.wait_stream(main)
        with torch.cuda.stream(s):
            losses.append(model(v, t)[0])
    for s, l in zip((sa, sb), losses):
        with torch.cuda.stream(s):
            l.backward()
    main.wait_stream(sa)
    main.wait_stream(sb)
    engine.micro_steps += 1
    engine.step()


for name, fn in (("full batch, one stream", step_full), ("two halves, two streams", step_split), ("full batch, one stream", step_full),
                 ("two halves, two streams", step_split)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(15):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 15 * 1e3:.2f} ms/step", flush=True)

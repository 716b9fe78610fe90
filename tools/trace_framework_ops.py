"""Which framework (aten) kernels would a steady-state training step launch from the HOST pipelines, and from which line?

DESIGN section 2 promises "no framework tensor ops on the hot path"; the r03 kernel trace still showed at::native fills / copies /
copyBuffer in every step.  A GPU trace names the kernels, not the call sites.  This tool runs the real
pipelines (vision.py / gpt3.py / pretrain.py / engine.py) on CPU over the torch stand-ins of tests/standin_ops.py (test
infrastructure: the stand-ins replace the C entry points, so everything THEY do is excluded) under a TorchDispatchMode and prints
every non-view aten op of the third step with the product line that issued it.  No GPU needed:
    python tools/trace_framework_ops.py            # tiny config, eager engine step (CPU, stand-ins)
    python tools/trace_framework_ops.py --gpu      # the benchmark configuration (config B) on the real entry points: every aten op a
                                                   # steady-state step dispatches, by call site (GPU box)
"""
import collections
import os
import sys
import traceback
import types

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

VIEW_OPS = {"view", "reshape", "as_strided", "slice", "select", "expand", "permute", "transpose", "t", "unsqueeze", "squeeze", "detach",
            "alias", "_unsafe_view", "unbind", "split", "split_with_sizes", "chunk", "narrow", "empty", "empty_like", "empty_strided",
            "new_empty", "_local_scalar_dense", "item", "lift_fresh", "sym_size", "sym_stride", "sym_numel", "set_", "resize_", "unfold",
            "diagonal", "_reshape_alias", "view_as", "flatten", "unflatten", "is_same_size", "is_nonzero", "_has_compatible_shallow_copy_type"}


class Tracer(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.on = False
        self.hits = collections.Counter()

    def __torch_dispatch__(self, func, types_, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        if self.on:
            name = func.__name__ if hasattr(func, "__name__") else str(func)
            full = str(func)
            if full.split("aten.")[-1].split(".")[0] not in VIEW_OPS:
                st = traceback.extract_stack()
                if not any(f.filename.endswith(("standin_ops.py", "test_engine_cpu.py")) for f in st):
                    site = next((f for f in reversed(st) if "youku-mplug_amd" in f.filename or f.filename.endswith(("bench.py",))), None)
                    where = f"{os.path.relpath(site.filename, ROOT)}:{site.lineno} {site.line}" if site else "<outside the product>"
                    numel = next((a.numel() for a in args if torch.is_tensor(a)), 0)
                    self.hits[(full, where, numel)] += 1
        return out


class _MP:
    def setattr(self, obj, name, val):
        setattr(obj, name, val)


def main_gpu():
    """Config B through the real library: anything listed here is a framework kernel (or copy engine call) in the timed step."""
    import bench
    from youku_mplug_amd import engine as eng
    from youku_mplug_amd.pretrain import synthetic_model
    dev = torch.device("cuda", 0)
    S = bench.Shapes
    model = synthetic_model(S, device=dev, num_frames=8)
    model.train()
    groups = eng.get_parameter_groups(model, 0.05, model.no_weight_decay(), visual_backbone_scale=True)
    e, opt, _, _ = eng.initialize(model=model, model_parameters=groups, config=dict(lr=1e-4, clip_grad=3.0))
    B, L = 32, 32
    video = torch.randn(B, 3, 8, 224, 224, device=dev).to(torch.bfloat16)
    text = types.SimpleNamespace(input_ids=torch.randint(0, S.vocab, (B, L), device=dev), attention_mask=torch.ones(B, L, dtype=torch.long, device=dev))
    tr = Tracer()
    with tr:
        for step in range(4):
            tr.on = step == 3
            for g in opt.param_groups:
                g["lr"] = 1e-4 * g["lr_scale"]
            loss, _ = e(video, text)
            e.backward(loss)
            e.step()
    torch.cuda.synchronize()
    report(tr, "config B (12 ViT blocks, 24 decoder layers), real entry points")


def report(tr, what):
    print(f"{'count':>5}  {'numel':>9}  op  <-  call site")
    for (op, where, numel), n in sorted(tr.hits.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        print(f"{n:5d}  {numel:9d}  {op}  <-  {where}")
    print(f"total: {sum(tr.hits.values())} framework ops in one steady-state step ({what})")


def main():
    if "--gpu" in sys.argv:
        return main_gpu()
    import standin_ops
    from oracle.weights import CONFIG_TINY, make_inputs, make_state_dict
    from youku_mplug_amd import engine as eng
    from youku_mplug_amd.pretrain import synthetic_model
    standin_ops.install(_MP())
    import test_engine_cpu
    test_engine_cpu._stub_optimizer_kernels(_MP())
    cfg = CONFIG_TINY
    model = synthetic_model(cfg, device="cpu")
    model.load_state_dict({k: v.to(torch.bfloat16) for k, v in make_state_dict(cfg, 0).items()})
    model.train()
    for m in model.modules():          # the stand-ins do not model dropout
        for a in ("hidden_dropout", "attention_dropout"):
            if hasattr(m, a):
                setattr(m, a, 0.0)
    if hasattr(model.text_decoder, "config"):
        for a in ("hidden_dropout", "attention_dropout"):
            if hasattr(model.text_decoder.config, a):
                setattr(model.text_decoder.config, a, 0.0)
    groups = eng.get_parameter_groups(model, 0.05, model.no_weight_decay(), visual_backbone_scale=True)
    e, _, _, _ = eng.initialize(model=model, model_parameters=groups, config=dict(lr=1e-4, clip_grad=3.0))
    video, ids, mask = make_inputs(cfg, 2, 8, seed=1, ragged=False)
    text = types.SimpleNamespace(input_ids=ids, attention_mask=mask)
    tr = Tracer()
    with tr:
        for step in range(3):
            tr.on = step == 2
            loss, _ = e(video.to(torch.bfloat16), text)
            e.backward(loss)
            e.step()
            e.zero_grad()
    report(tr, f"tiny config: {cfg.vit_depth} ViT blocks, {cfg.layers} decoder layers; CPU stand-ins")


if __name__ == "__main__":
    main()

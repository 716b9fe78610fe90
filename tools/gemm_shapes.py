"""Per-shape breakdown of the GEMM launches of one config-B training step (HIP events around every launch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def main():
    import types
    from bench import Shapes
    from youku_mplug_amd import engine as eng
    from youku_mplug_amd.pretrain import synthetic_model
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    torch.manual_seed(1234)
    B, T, L = 32, 8, 32
    Shapes.num_frames = T
    model = synthetic_model(Shapes, device=dev, num_frames=T)
    with torch.no_grad():
        for blk in model.visual_encoder.blocks:
            blk.temporal_fc.weight.normal_(0, 0.015)
    model.train()
    groups = eng.get_parameter_groups(model, 0.05, model.no_weight_decay(), visual_backbone_scale=True)
    engine, opt, _, _ = eng.initialize(model=model, model_parameters=groups,
                                       config=dict(lr=1e-4, opt_betas=(0.9, 0.999), opt_eps=1e-6, clip_grad=3.0))
    video = torch.randn(B, 3, T, 224, 224, device=dev).to(torch.bfloat16)
    ids = torch.randint(0, Shapes.vocab, (B, L), device=dev)
    text = types.SimpleNamespace(input_ids=ids, attention_mask=torch.ones(B, L, dtype=torch.long, device=dev))

    def one():
        loss, _ = engine(video, text)
        engine.backward(loss)
        engine.step()
    for _ in range(2):
        one()
    from youku_mplug_amd import ops
    orig = ops.gemm
    recs = []

    def timed(a, b, M, N, K, **kw):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = orig(a, b, M, N, K, **kw)
        e.record()
        kind = "wgrad" if kw.get("trans_a") else ("dgrad" if kw.get("trans_b") else "fwd")
        recs.append(((kind, M, N, K), s, e))
        return r
    ops.gemm = timed
    steps = 3
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    ops.gemm = orig
    agg = {}
    for key, s, e in recs:
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += s.elapsed_time(e)
    tot = sum(a[1] for a in agg.values()) / steps
    print(f"total GEMM {tot:.2f} ms/step")
    print(f"{'kind':6s} {'M':>6s} {'N':>6s} {'K':>6s} {'calls':>5s} {'ms/step':>8s} {'us/call':>8s} {'TF/s':>7s} {'%':>5s}")
    for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        kind, M, N, K = key
        ms = a[1] / steps
        us = a[1] / a[0] * 1e3
        print(f"{kind:6s} {M:6d} {N:6d} {K:6d} {a[0]//steps:5d} {ms:8.2f} {us:8.1f} {2.0*M*N*K/us/1e6:7.0f} {100*ms/tot:5.1f}")


if __name__ == "__main__":
    main()

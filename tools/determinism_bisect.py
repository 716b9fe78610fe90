"""Run-to-run determinism of a training step, bisected to the first entry point whose output bits differ.

    python tools/determinism_bisect.py [--config D] [--batch 16] [--eval]

Two steps from ONE state (engine.snapshot_state / restore_state) with every `ops.*` call wrapped: after each call the raw bits of every
tensor it was handed or returned are summed (int64) and logged in call order; the two logs are compared and the first call whose
checksum differs is printed (name, index in the step, which tensor).  A step whose loss differs from run to run on the same inputs and
seeds has a race or an uninitialised read somewhere; this finds where.  MEASUREMENT / DEBUG TOOL: not part of the product."""
import argparse
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def checksum(t):
    if not torch.is_tensor(t) or t.device.type != "cuda" or t.numel() == 0:
        return None
    try:
        c = t.contiguous() if not t.is_contiguous() else t
        raw = c.view(-1).view(torch.uint8) if c.dtype != torch.bool else c.view(-1).to(torch.uint8)
        n = raw.numel() // 8 * 8
        s = int(raw[:n].view(torch.int64).sum().item()) if n else 0
        return (s + int(raw[n:].to(torch.int64).sum().item())) & 0xFFFFFFFFFFFFFFFF
    except Exception:
        return None


def flatten(x, out, tag):
    if torch.is_tensor(x):
        out.append((tag, x))
    elif isinstance(x, (list, tuple)):
        for i, y in enumerate(x):
            flatten(y, out, f"{tag}[{i}]")
    elif isinstance(x, dict):
        for k, y in x.items():
            flatten(y, out, f"{tag}.{k}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="D")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--eval", action="store_true")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=2, help="steps before the logged ones (0: the logged step starts from the seeded initial weights -- "
                    "what two PROCESSES must agree on from the first call)")
    ap.add_argument("--poison", type=float, default=0.0,
                    help="GiB of device memory to fill with 0xFF bytes (NaN in bf16 and fp32) and hand back to the caching allocator BEFORE anything "
                         "is built: every later allocation then starts as NaN, and an uninitialised read that reaches the loss shows up as a NaN "
                         "loss; the tensors that hold NaNs after each call are listed in call order")
    ap.add_argument("--zero-empty", action="store_true", help="torch.empty / empty_like return zero-filled tensors in this process: unwritten "
                    "rows of row-mapped outputs are then the same bits in every process, and a result that depends on uninitialised memory stops varying")
    ap.add_argument("--dump", default=None, help="write run 0's call log (name, tensor tags, shapes, checksums) as JSON: compare two PROCESSES with --compare")
    ap.add_argument("--compare", nargs=2, default=None, help="two --dump files: print the first calls whose checksums differ")
    args = ap.parse_args()
    if args.compare:
        import json
        a, b = (json.load(open(f)) for f in args.compare)
        print(f"{len(a)} / {len(b)} calls")
        shown = 0
        for i, ((na, ta), (nb, tb)) in enumerate(zip(a, b)):
            if na == "workspace":
                continue
            bad = [(x[0], x[1]) for x, y in zip(ta, tb) if x[2] != y[2]]
            if na != nb or bad:
                print(f"call {i}: {na} differs in {bad}")
                shown += 1
                if shown >= 12:
                    break
        if not shown:
            print("the two processes agree on every call")
        return
    if args.zero_empty:
        _e, _el = torch.empty, torch.empty_like
        torch.empty = lambda *a, **k: _e(*a, **k).zero_()
        torch.empty_like = lambda *a, **k: _el(*a, **k).zero_()
    import bench
    import youku_mplug_amd  # noqa: F401
    from youku_mplug_amd import engine as eng, ops
    from youku_mplug_amd.pretrain import synthetic_model
    S = bench.ShapesD if args.config == "D" else bench.Shapes
    geo = bench.GEOMETRY[args.config]
    B, T, L = args.batch or geo["batch"], geo["frames"], geo["text_len"]
    S.num_frames = T
    dev = torch.device("cuda", 0)
    def poison():
        torch.cuda.synchronize()
        torch.cuda.empty_cache()        # nothing cached but what is poisoned next
        chunks = [torch.empty(int(2 ** 30), dtype=torch.uint8, device=dev) for _ in range(int(args.poison))]
        for c in chunks:
            c.fill_(0xFF)
        torch.cuda.synchronize()
        del chunks                      # (stays reserved: the blocks are what later allocations are carved from)
    torch.manual_seed(1234)
    model = synthetic_model(S, device=dev, num_frames=T)
    with torch.no_grad():
        for blk in model.visual_encoder.blocks:
            blk.temporal_fc.weight.normal_(0, 0.015)
    model.train()
    if args.eval:
        model.eval()
    groups = eng.get_parameter_groups(model, 0.05, model.no_weight_decay(), visual_backbone_scale=True)
    e, opt, _, _ = eng.initialize(model=model, model_parameters=groups, config=dict(lr=1e-4, clip_grad=3.0))
    video = torch.randn(B, 3, T, S.img_size, S.img_size, device=dev).to(torch.bfloat16)
    ids = torch.randint(0, S.vocab, (B, L), device=dev)
    text = types.SimpleNamespace(input_ids=ids, attention_mask=torch.ones(B, L, dtype=torch.long, device=dev))
    for _ in range(args.warmup):            # lazily created buffers exist, the allocator is warm
        loss, _ = e(video, text)
        e.backward(loss)
        e.step()
    torch.cuda.synchronize()
    snap = e.snapshot_state()
    names = [n for n in dir(ops) if callable(getattr(ops, n)) and not n.startswith("_") and getattr(getattr(ops, n), "__module__", "") == ops.__name__
             and isinstance(getattr(ops, n), types.FunctionType)]
    logs = []
    nan_log = []
    for run in range(args.steps):
        e.restore_state(snap)
        if args.poison > 0:
            poison()                    # every activation of this step is carved out of NaN-filled memory
        log = []
        orig = {n: getattr(ops, n) for n in names}

        def wrap(n, f):
            def g(*a, **k):
                r = f(*a, **k)
                ts = []
                flatten(a, ts, "arg")
                flatten(k, ts, "kw")
                flatten(r, ts, "ret")
                log.append((n, [(tag, tuple(t.shape), checksum(t)) for tag, t in ts]))
                if args.poison > 0 and run == 0:
                    for tag, t in ts:
                        if t.is_floating_point() and t.device.type == "cuda" and t.numel():
                            nn_ = int(torch.isnan(t).sum().item())
                            if nn_:
                                nan_log.append((len(log) - 1, n, tag, tuple(t.shape), nn_, t.numel()))
                return r
            return g
        for n, f in orig.items():
            setattr(ops, n, wrap(n, f))
        try:
            loss, _ = e(video, text)
            e.backward(loss)
            e.step()
            torch.cuda.synchronize()
        finally:
            for n, f in orig.items():
                setattr(ops, n, f)
        log.append(("FINAL", [("loss", (), checksum(loss.detach().float().view(1))), ("params", (), checksum(e.flat.params)), ("grads", (), checksum(e.flat.grads))]))
        logs.append(log)
        print(f"run {run}: {len(log)} calls, loss {loss.item():.6f}")
        if args.dump and run == 0:
            import json
            json.dump(log, open(args.dump, "w"))
    if args.poison > 0:
        seen = set()
        print(f"tensors holding NaNs after a call (first occurrence per (entry point, tensor, shape)); {len(nan_log)} in all:")
        for i, n, tag, shape, cnt, tot in nan_log:
            key = (n, tag, shape)
            if key in seen:
                continue
            seen.add(key)
            print(f"    call {i:5d} {n:28s} {tag:14s} {str(shape):24s} {cnt} of {tot} NaN")
        return
    a = logs[0]
    for r, b in enumerate(logs[1:], 1):
        if len(a) != len(b):
            print(f"run {r}: {len(b)} calls against {len(a)}")
        first = None
        ndiff = 0
        for i, ((na, ta), (nb, tb)) in enumerate(zip(a, b)):
            bad = [(x[0], x[1]) for x, y in zip(ta, tb) if x[2] != y[2]]
            if na == "workspace":           # the shared scratch buffer: its contents are whatever the last user left
                continue
            if na != nb or bad:
                ndiff += 1
                if first is None:
                    first = (i, na, nb, bad)
        if first is None:
            print(f"run {r}: every one of {len(a)} calls bit-identical to run 0 (loss, parameters and gradients included)")
        else:
            i, na, nb, bad = first
            print(f"run {r}: {ndiff} calls differ; FIRST at call {i}: {na} -- tensors {bad}")
            for j in range(max(0, i - 3), min(len(a), i + 2)):
                print("    ", j, a[j][0], [(t[0], t[1]) for t in a[j][1]][:6])


if __name__ == "__main__":
    main()

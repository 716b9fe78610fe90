"""CPU tests of the host-side engine logic (no GPU): the real MplugEngine at world size 2 over gloo (parameter broadcast at
initialize(), stage order / bucket slices / hooks, 1/world folded into the optimizer, clip on the reduced gradient),
gradient accumulation, DeepSpeed-layout checkpoint round trip, `pretrained_ckpt` initialisation and the pos / temporal
embedding resize.  The two optimizer kernels exist only as HIP: these tests replace exactly those two kernel calls
(ops.grad_sumsq / ops.adamw_step_grouped / ops.add) with their arithmetic in torch -- test infrastructure, never the product."""
import math
import os
import sys
import tempfile

import pytest
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import youku_mplug_amd  # noqa: E402,F401


# ------------------------------------------------------------------------------ kernel-call stubs (arithmetic of optim/adamw.py:66-115)
def _stub_optimizer_kernels(monkeypatch_target):
    from youku_mplug_amd import engine as eng

    def grad_sumsq(g, out):
        out.add_(g.float().pow(2).sum())

    def adamw_step_grouped(p16, master, m, v, g16, tile_group, lrs, wds, beta1, beta2, eps, step, grad_scale=1.0, sumsq=None,
                           max_norm=0.0):
        g = g16.float() * grad_scale
        if max_norm > 0:
            norm = math.sqrt(float(sumsq)) * grad_scale
            g = g * min(1.0, max_norm / (norm + 1e-6))
        tg = tile_group.long().repeat_interleave(eng.TILE)
        for gi, (lr, wd) in enumerate(zip(lrs, wds)):
            sel = tg == gi
            if not sel.any():
                continue
            master[sel] *= 1.0 - lr * wd
            m[sel] = beta1 * m[sel] + (1 - beta1) * g[sel]
            v[sel] = beta2 * v[sel] + (1 - beta2) * g[sel] ** 2
            denom = (v[sel].sqrt() / math.sqrt(1 - beta2 ** step)) + eps
            master[sel] -= (lr / (1 - beta1 ** step)) * m[sel] / denom
        p16.copy_(master.to(p16.dtype))

    def add(a, b, out=None):
        out = out if out is not None else torch.empty_like(a)
        torch.add(a, b, out=out)
        return out

    def accum_f32(acc, g, first):
        if first:
            acc.copy_(g.float())
        else:
            acc.add_(g.float())
        return acc

    def f32_to_bf16(src, dst):
        dst.copy_(src.to(dst.dtype))
        return dst

    from youku_mplug_amd import ops
    monkeypatch_target.setattr(ops, "accum_f32", accum_f32)
    monkeypatch_target.setattr(ops, "f32_to_bf16", f32_to_bf16)
    monkeypatch_target.setattr(ops, "grad_sumsq", grad_sumsq)
    monkeypatch_target.setattr(ops, "adamw_step_grouped", adamw_step_grouped)
    monkeypatch_target.setattr(ops, "add", add)


class _MP:      # minimal monkeypatch for spawned workers (pytest's fixture does not cross processes)
    def setattr(self, obj, name, val):
        setattr(obj, name, val)


# ------------------------------------------------------------------------------ a stub model with the real model's staging surface
class _StubVit(nn.Module):
    def __init__(self, dim, depth):
        super().__init__()
        self.stem = nn.Linear(dim, dim)
        self.blocks = nn.ModuleList([nn.Linear(dim, dim) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim)
        self.on_block_grads_ready = None


class _StubFn(torch.autograd.Function):
    """One autograd node whose backward replays an explicit, stage-ordered backward (head -> block n-1 .. 0 -> stem) and
    fires the engine's hooks exactly as vision.TimeSformer.backward_features / pretrain._backward_pipeline do."""

    @staticmethod
    def forward(ctx, anchor, model, x, y):
        with torch.enable_grad():
            h = model.visual_encoder.stem(x)
            for blk in model.visual_encoder.blocks:
                h = h + torch.tanh(blk(h))
            loss = ((model.head(model.visual_encoder.norm(h)) - y) ** 2).mean()
        ctx.model, ctx.loss = model, loss
        return loss.detach()

    @staticmethod
    def backward(ctx, g):
        m = ctx.model
        ve = m.visual_encoder
        params = [p for p in m.parameters() if p.requires_grad]
        grads = dict(zip([id(p) for p in params], torch.autograd.grad(ctx.loss, params)))

        def put(ps):
            for p in ps:
                p.grad.copy_(grads[id(p)] * g)
        put(list(m.head.parameters()))
        m.on_stage_grads_ready("head")
        put(list(ve.norm.parameters()))
        for bi in range(len(ve.blocks) - 1, -1, -1):
            put(list(ve.blocks[bi].parameters()))
            ve.on_block_grads_ready(bi)
        put(list(ve.stem.parameters()))
        ve.on_block_grads_ready(-1)
        return torch.zeros(1), None, None, None


class _StubModel(nn.Module):
    def __init__(self, dim=24, depth=3, seed=0):
        super().__init__()
        torch.manual_seed(seed)
        self.visual_encoder = _StubVit(dim, depth)
        self.head = nn.Linear(dim, 5)
        self.on_stage_grads_ready = None
        self._anchor = torch.zeros(1, requires_grad=True)

    def forward(self, x, y):
        return _StubFn.apply(self._anchor, self, x, y), None


def _data(n=8, dim=24):
    g = torch.Generator().manual_seed(77)
    return torch.randn(n, dim, generator=g), torch.randn(n, 5, generator=g)


def _engine_worker(rank, world, port, q, zero=0, ckpt_dir=None):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _stub_optimizer_kernels(_MP())
    from youku_mplug_amd import engine as eng
    model = _StubModel(seed=1234 + rank)                  # the reference seeds every rank differently (run_pretrain...py:210)
    launched = []
    groups = eng.get_parameter_groups(model, 0.05)
    cfg = dict(lr=1e-2, clip_grad=0.5, opt_eps=1e-6)
    if zero:
        cfg["zero_optimization"] = {"stage": zero}        # as utils.py:528-529 writes it into the DeepSpeed config
    gathers = [0]
    orig_gather = dist.all_gather_into_tensor

    def counted_gather(*a, **k):
        gathers[0] += 1
        return orig_gather(*a, **k)
    dist.all_gather_into_tensor = counted_gather
    engine, opt, _, _ = eng.initialize(model=model, model_parameters=groups, config=cfg)
    orig = engine.reducer.stage_ready
    engine.reducer.stage_ready = lambda name: (launched.append(name), orig(name))[1]
    model.visual_encoder.on_block_grads_ready = lambda bi: engine.reducer.stage_ready("stem" if bi < 0 else f"block{bi}")
    model.on_stage_grads_ready = engine.reducer.stage_ready
    p0 = engine.flat.params.clone()
    x, y = _data()
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    for _ in range(2):
        loss, _ = engine(xs, ys)
        engine.backward(loss)
        engine.step()
    extra = None
    if zero:
        assert engine.zero_shards is not None and opt.master.numel() == opt.hi - opt.lo < engine.flat.numel
        engine.save_checkpoint(ckpt_dir, tag="z")
        files = sorted(os.listdir(os.path.join(ckpt_dir, "z")))
        before = (engine.flat.params.clone(), opt.exp_avg.clone(), opt.step_count)
        engine.flat.params.zero_()
        opt.exp_avg.zero_()
        engine.load_checkpoint(ckpt_dir, tag="z")
        extra = (files, torch.equal(before[0], engine.flat.params) and torch.equal(before[1], opt.exp_avg) and opt.step_count == before[2],
                 (opt.lo, opt.hi), gathers[0], engine.flat.numel)
    q.put((rank, p0, engine.flat.params.clone(), list(engine.flat.stage_slices.items()), launched[:6], opt._global_grad_norm, extra))
    dist.barrier()
    dist.destroy_process_group()


def test_real_engine_world2_gloo(monkeypatch):
    """MplugEngine at N=2: replicas are identical after initialize() although built from different seeds; the buckets go
    out in backward-completion order over the expected flat slices; two steps match a single process that sees the whole
    batch (mean gradient, clip on the reduced gradient, 1/world folded into AdamW)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 23000 + os.getpid() % 3000
    procs = [ctx.Process(target=_engine_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r[0]: r[1:] for r in (q.get(timeout=180) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.equal(res[0][0], res[1][0]), "parameters must be broadcast from rank 0 at initialize()"
    assert torch.equal(res[0][1], res[1][1]), "replicas diverged"
    names = [n for n, _ in res[0][2]]
    assert names == ["head", "block2", "block1", "block0", "stem"]
    assert res[0][3] == ["head", "block2", "block1", "block0", "stem", "head"], "hooks fire in backward-completion order, every step"
    slices = dict(res[0][2])
    assert slices["head"][0] == 0 and all(a % 256 == 0 and b % 256 == 0 for a, b in slices.values())
    # single-process reference on the whole batch, same kernels-as-arithmetic
    _stub_optimizer_kernels(monkeypatch)
    from youku_mplug_amd import engine as eng
    model = _StubModel(seed=1234)                 # rank 0's initialisation
    groups = eng.get_parameter_groups(model, 0.05)
    engine, opt, _, _ = eng.initialize(model=model, model_parameters=groups, config=dict(lr=1e-2, clip_grad=0.5, opt_eps=1e-6))
    assert torch.equal(engine.flat.params, res[0][0])
    x, y = _data()
    for _ in range(2):
        loss, _ = engine(x, y)
        engine.backward(loss)
        engine.step()
    assert torch.allclose(engine.flat.params, res[0][1], atol=2e-6), (engine.flat.params - res[0][1]).abs().max()
    assert abs(opt._global_grad_norm - res[0][4]) < 1e-5


def test_zero_stage1_world2_gloo_matches_replicated_optimizer(tmp_path):
    """ZeRO stage 1 (utils.py:528-529): optimizer states partitioned over the ranks, every rank updates its own run of the flat buffer and
    the runs are handed round -- the parameters after two steps are BIT-identical to the replicated optimizer's, each rank holds only
    its shard of master / exp_avg / exp_avg_sq, and the checkpoint is DeepSpeed's ZeRO layout (one optimizer-state file per rank)
    that loads back."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    results = {}
    for zero in (0, 1):
        q = ctx.Queue()
        port = 26000 + (os.getpid() + 17 * zero) % 3000
        d = str(tmp_path / f"ck{zero}")
        os.makedirs(d, exist_ok=True)
        procs = [ctx.Process(target=_engine_worker, args=(r, 2, port, q, zero, d)) for r in range(2)]
        for p in procs:
            p.start()
        res = {r[0]: r[1:] for r in (q.get(timeout=180) for _ in range(2))}
        for p in procs:
            p.join(timeout=60)
        results[zero] = res
    for r in range(2):
        assert torch.equal(results[0][r][1], results[1][r][1]), f"rank {r}: ZeRO-1 parameters differ from the replicated optimizer's"
    assert torch.equal(results[1][0][1], results[1][1][1])
    files, reloaded, shard0, gathers, numel = results[1][0][5]
    _, _, shard1, _, _ = results[1][1][5]
    assert reloaded and results[1][1][5][1]
    # equal runs of whole tiles over the (padded) flat buffer, shared by ONE all-gather per optimizer step
    assert gathers == 2 and shard1[1] == numel and shard0[1] - shard0[0] == shard1[1] - shard1[0] and numel % 512 == 0
    assert "mp_rank_00_model_states.pt" in files and "zero_pp_rank_0_mp_rank_00_optim_states.pt" in files and \
        "zero_pp_rank_1_mp_rank_00_optim_states.pt" in files and "mp_rank_00_optim_states.pt" not in files
    assert shard0[0] == 0 and shard0[1] == shard1[0] and shard1[1] >= shard1[0]


def test_flat_params_padding_for_equal_zero_runs():
    """FlatParams(pad_tiles_to=w): the tile count becomes a multiple of w; the padding tiles carry group 255 (the optimizer kernel
    skips groups >= 8) and no parameter lives in them."""
    from youku_mplug_amd.engine import TILE, FlatParams
    ps = [nn.Parameter(torch.randn(300).to(torch.bfloat16)), nn.Parameter(torch.randn(5, 7).to(torch.bfloat16)), nn.Parameter(torch.randn(700).to(torch.bfloat16))]
    group_of = {id(p): i % 2 for i, p in enumerate(ps)}
    plain = FlatParams([("head", ps[:2]), ("stem", ps[2:])], group_of)
    assert plain.numel == (2 + 1 + 3) * TILE
    ps2 = [nn.Parameter(p.detach().clone()) for p in ps]
    padded = FlatParams([("head", ps2[:2]), ("stem", ps2[2:])], {id(p): i % 2 for i, p in enumerate(ps2)}, pad_tiles_to=4)
    assert padded.numel == 8 * TILE and padded.stage_slices == plain.stage_slices
    assert padded.tile_group.tolist() == plain.tile_group.tolist() + [255, 255]
    assert torch.equal(padded.params[:plain.numel], plain.params) and padded.params[plain.numel:].abs().sum().item() == 0


def test_gradient_accumulation_matches_big_batch(monkeypatch):
    """`--update_freq 2` (DeepSpeed gradient_accumulation_steps): two micro-batches, one optimizer step on their mean
    gradient; step() in the middle of the window applies nothing (run_pretrain_distributed_gpt3.py:46-53, 88-96)."""
    _stub_optimizer_kernels(monkeypatch)
    from youku_mplug_amd import engine as eng
    x, y = _data()
    ma, mb = _StubModel(seed=5), _StubModel(seed=5)
    ea, _, _, _ = eng.initialize(model=ma, model_parameters=eng.get_parameter_groups(ma, 0.05), config=dict(lr=1e-2, update_freq=2))
    eb, _, _, _ = eng.initialize(model=mb, model_parameters=eng.get_parameter_groups(mb, 0.05), config=dict(lr=1e-2))
    assert ea.gas == 2 and eb.gas == 1
    start = ea.flat.params.clone()
    loss, _ = ea(x[:4], y[:4])
    ea.backward(loss)
    ea.step()
    assert torch.equal(ea.flat.params, start) and ea.global_steps == 0 and not ea.is_gradient_accumulation_boundary()
    loss, _ = ea(x[4:], y[4:])
    ea.backward(loss)
    ea.step()
    assert ea.global_steps == 1
    loss, _ = eb(x, y)
    eb.backward(loss)
    eb.step()
    assert torch.allclose(ea.flat.params, eb.flat.params, atol=2e-6)


def test_epoch_reset_of_micro_steps_drops_a_partial_accumulation_window(monkeypatch):
    """the training loops write `model.micro_steps = 0` at every epoch start (run_pretrain_distributed_gpt3.py:72-73).
    With len(loader) % update_freq != 0 the last window of an epoch is partial; DeepSpeed's boundary is micro_steps % gas, so
    the reset realigns it and the partial sum is discarded.  Here: one stray micro-batch, epoch reset, then a full window must
    equal the same full window on a fresh engine."""
    _stub_optimizer_kernels(monkeypatch)
    from youku_mplug_amd import engine as eng
    x, y = _data()
    ma, mb = _StubModel(seed=5), _StubModel(seed=5)
    ea, _, _, _ = eng.initialize(model=ma, model_parameters=eng.get_parameter_groups(ma, 0.05), config=dict(lr=1e-2, update_freq=2))
    eb, _, _, _ = eng.initialize(model=mb, model_parameters=eng.get_parameter_groups(mb, 0.05), config=dict(lr=1e-2, update_freq=2))
    loss, _ = ea(3.0 * x[:4], y[:4])          # the ragged tail of "epoch 0": half a window
    ea.backward(loss)
    ea.step()
    assert not ea.is_gradient_accumulation_boundary() and ea.global_steps == 0
    ea.micro_steps = 0                         # epoch 1 starts
    ea.zero_grad()
    assert ea.is_gradient_accumulation_boundary() and ea.micro_steps == 0
    for e in (ea, eb):
        for sl in (slice(0, 4), slice(4, 8)):
            loss, _ = e(x[sl], y[sl])
            e.backward(loss)
            e.step()
    assert ea.global_steps == 1 and eb.global_steps == 1
    assert torch.equal(ea.flat.params, eb.flat.params)


def test_checkpoint_round_trip_deepspeed_layout(monkeypatch):
    """save_checkpoint -> load_checkpoint into a fresh engine: module weights, fp32 master / moments and the step counter
    survive; files follow utils.py:440-480 (<dir>/<tag>/mp_rank_00_model_states.pt with key 'module', <dir>/latest)."""
    _stub_optimizer_kernels(monkeypatch)
    from youku_mplug_amd import engine as eng
    x, y = _data()
    m1 = _StubModel(seed=9)
    e1, _, _, _ = eng.initialize(model=m1, model_parameters=eng.get_parameter_groups(m1, 0.05), config=dict(lr=1e-2))
    for _ in range(3):
        loss, _ = e1(x, y)
        e1.backward(loss)
        e1.step()
    with tempfile.TemporaryDirectory() as d:
        e1.save_checkpoint(d, tag="checkpoint-3", client_state={"epoch": 3})
        assert open(os.path.join(d, "latest")).read().strip() == "checkpoint-3"
        raw = torch.load(os.path.join(d, "checkpoint-3", "mp_rank_00_model_states.pt"))
        # 'micro_batches_seen': the dropout stream's position travels with the model file (every rank of a ZeRO resume reads it there)
        assert set(raw) == {"module", "epoch", "micro_batches_seen"} and set(raw["module"]) == set(m1.state_dict())
        assert raw["micro_batches_seen"] == 3
        m2 = _StubModel(seed=10)
        e2, _, _, _ = eng.initialize(model=m2, model_parameters=eng.get_parameter_groups(m2, 0.05), config=dict(lr=1e-2))
        path, client = e2.load_checkpoint(d)
        assert client == {"epoch": 3} and path.endswith("checkpoint-3")
        assert e2.micro_batches_seen == 3
        # a rank WITHOUT an optimizer-state file of its own (a ZeRO-1 resume at a larger world size: ADVICE r05) still continues the dropout
        # stream where the checkpoint left it: the position is read from the model-states file, which every rank loads
        os.remove(os.path.join(d, "checkpoint-3", "mp_rank_00_optim_states.pt"))
        m3 = _StubModel(seed=11)
        e3, _, _, _ = eng.initialize(model=m3, model_parameters=eng.get_parameter_groups(m3, 0.05), config=dict(lr=1e-2))
        e3.load_checkpoint(d)
        assert e3.micro_batches_seen == 3 and e3.optimizer.step_count == 0 and torch.equal(e3.flat.params, e1.flat.params)
    assert torch.equal(e1.flat.params, e2.flat.params)
    for a, b in ((e1.optimizer.master, e2.optimizer.master), (e1.optimizer.exp_avg, e2.optimizer.exp_avg),
                 (e1.optimizer.exp_avg_sq, e2.optimizer.exp_avg_sq)):
        assert torch.equal(a, b)
    assert e2.optimizer.step_count == 3
    for e in (e1, e2):          # and the two continue identically
        loss, _ = e(x, y)
        e.backward(loss)
        e.step()
    assert torch.equal(e1.flat.params, e2.flat.params)


# ------------------------------------------------------------------------------ pretrained_ckpt / resize
def test_pretrained_ckpt_initialises_vision_tower(tmp_path, monkeypatch):
    """visual config `pretrained_ckpt: clip/<file>` (configs/models/clip-b16.json:2; models/distributed_gpt3.py:56-72):
    a CLIP-layout checkpoint (fused qkv.bias, a head) lands in q_bias / v_bias, strict=False."""
    from oracle.weights import CONFIG_TINY
    from youku_mplug_amd.gpt3 import GPT3Config
    from youku_mplug_amd.pretrain import DistributedGPT3_Pretrain, synthetic_model
    donor = synthetic_model(CONFIG_TINY, device="cpu").visual_encoder
    with torch.no_grad():
        for p in donor.parameters():
            p.copy_(torch.randn_like(p.float()).to(p.dtype))
    clip = {}
    for k, v in donor.state_dict().items():
        if k.endswith("q_bias"):
            vb = donor.state_dict()[k.replace("q_bias", "v_bias")]
            clip[k.replace("q_bias", "qkv.bias")] = torch.cat([v, torch.zeros_like(v), vb]).float()
        elif k.endswith("v_bias") or "temporal" in k:
            continue                                        # an image CLIP ViT has no temporal branch
        else:
            clip[k] = v.float()
    clip["head.weight"] = torch.zeros(3, 3)
    monkeypatch.chdir(tmp_path)
    torch.save(clip, "clip_vit_tiny.pth")
    s = CONFIG_TINY
    vis = dict(img_size=s.img_size, patch_size=s.patch_size, depth=s.vit_depth, num_frames=s.num_frames, embed_dim=s.vit_dim,
               num_heads=s.vit_heads, mlp_ratio=s.vit_mlp_ratio, clip_model=True, pretrained_ckpt="clip/clip_vit_tiny.pth")
    txt = GPT3Config(vocab_size=s.vocab, hidden_size=s.hidden, ffn_hidden_size=s.ffn, num_hidden_layers=s.layers,
                     num_attention_heads=s.heads, max_position_embeddings=s.max_pos, layernorm_epsilon=s.gpt_ln_eps)
    model = DistributedGPT3_Pretrain({"num_learnable_token": s.num_queries, "_synthetic": True}, visual_cfg=vis, text_cfg=txt, device="cpu")
    got = model.visual_encoder.state_dict()
    for k, v in donor.state_dict().items():
        if "temporal" in k:
            continue
        assert torch.equal(got[k], v), k
    assert any("temporal" in k for k in got)
    vis["pretrained_ckpt"] = "clip/missing.pth"
    with pytest.raises(FileNotFoundError):
        DistributedGPT3_Pretrain({"num_learnable_token": s.num_queries, "_synthetic": True}, visual_cfg=vis, text_cfg=txt, device="cpu")


def test_resize_pos_and_temporal_embed():
    from youku_mplug_amd.vision import resize_pos_embed, resize_temporal_embed
    g = torch.Generator().manual_seed(3)
    pos = torch.randn(1, 1 + 14 * 14, 32, generator=g)
    new = resize_pos_embed(pos, torch.zeros(1, 1 + 16 * 16, 32))
    assert new.shape == (1, 257, 32) and torch.equal(new[:, 0], pos[:, 0])
    assert torch.equal(resize_pos_embed(pos, torch.zeros(1, 197, 32)), pos)         # same grid: identity
    tmp = torch.randn(1, 4, 32, generator=g)
    t8 = resize_temporal_embed(tmp, torch.zeros(1, 8, 32))
    assert t8.shape == (1, 8, 32) and torch.allclose(t8[:, 0], tmp[:, 0]) and torch.allclose(t8[:, -1], tmp[:, -1])
    pad = resize_temporal_embed(tmp, torch.zeros(1, 6, 32), mode="padding")
    assert torch.equal(pad[:, :4], tmp) and pad[:, 4:].abs().sum() == 0
    from oracle import ref_loader
    if ref_loader.reference_available():      # the reference's own functions (build container only), same tensors
        vt, _, _ = ref_loader.import_reference()
        assert torch.equal(vt.resize_pos_embed(pos, torch.zeros(1, 257, 32)), new)
        assert torch.equal(vt.resize_temporal_embed(tmp, torch.zeros(1, 8, 32)), t8)
        w = {"blocks.0.attn.qkv.bias": torch.arange(9.0), "head.weight": torch.zeros(2), "blocks.0.attn.qkv.weight": torch.ones(3)}
        from youku_mplug_amd.vision import convert_pretrained_vit
        a, b = convert_pretrained_vit(dict(w)), vt._convert_pretrained_vit(dict(w))
        assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)


def test_load_checkpoint_resizes_visual_embeds(monkeypatch, tmp_path):
    """downstream `--resume` path (downstream/run_retrieval_distributed_gpt3.py:402-420): a checkpoint trained at 4 frames /
    another grid loads into an 8-frame model through resize_pos_embed / resize_temporal_embed."""
    _stub_optimizer_kernels(monkeypatch)
    from oracle.weights import CONFIG_TINY
    from youku_mplug_amd import engine as eng
    from youku_mplug_amd.pretrain import synthetic_model
    small = synthetic_model(CONFIG_TINY, device="cpu", num_frames=2)
    big = synthetic_model(CONFIG_TINY, device="cpu", num_frames=4)
    with torch.no_grad():
        small.visual_encoder.temporal_embed.copy_(torch.randn(1, 2, CONFIG_TINY.vit_dim))
    e_small, _, _, _ = eng.initialize(model=small, model_parameters=eng.get_parameter_groups(small, 0.05), config=dict(lr=1e-3))
    e_small.save_checkpoint(str(tmp_path), tag="t")
    e_big, _, _, _ = eng.initialize(model=big, model_parameters=eng.get_parameter_groups(big, 0.05), config=dict(lr=1e-3))
    e_big.load_checkpoint(str(tmp_path), tag="t")
    te = big.visual_encoder.temporal_embed
    assert te.shape[1] == 4 and torch.allclose(te[:, 0].float(), small.visual_encoder.temporal_embed[:, 0].float(), atol=1e-2)
    assert torch.equal(big.visual_encoder.blocks[0].attn.qkv.weight, small.visual_encoder.blocks[0].attn.qkv.weight)


# ------------------------------------------------------------------------------ wire dtype of the gradient sum, world 8
def _sum_worker(rank, world, port, q, comm):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from youku_mplug_amd import engine as eng
    _stub_optimizer_kernels(_MP())
    n = 4096 * 3
    p = nn.Parameter(torch.zeros(n, dtype=torch.bfloat16))
    flat = eng.FlatParams([("a", [p])])
    g = torch.Generator().manual_seed(100 + rank)
    # gradient-like values: a wide dynamic range and rank-to-rank cancellation, the hard case for a low-precision running sum
    flat.grads.copy_((torch.randn(n, generator=g) * torch.logspace(-3, 0, n)).to(torch.bfloat16))
    red = eng.DPReducer(flat, comm_dtype=comm)
    red.stage_ready("a")
    red.finish()
    if rank == 0:
        q.put(flat.grads.float().clone())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("comm", ["bf16", "fp32"])
def test_bf16_bucket_sum_error_world8_gloo(comm):
    """The DP gradient sum over 8 ranks through the real DPReducer on gloo: the bf16 wire format (DeepSpeed's default: model dtype)
    against the exact fp32 sum of the same per-rank bf16 gradients stays two orders of magnitude inside the gradient gates of
    the parity tests (2-5e-2 of the norm); MPV_DP_COMM_DTYPE=fp32 rounds once."""
    import torch.multiprocessing as mp
    world, n = 8, 4096 * 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 28000 + os.getpid() % 1500 + (7 if comm == "fp32" else 0)
    procs = [ctx.Process(target=_sum_worker, args=(r, world, port, q, comm)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exact = torch.zeros(n)
    for r in range(world):
        g = torch.Generator().manual_seed(100 + r)
        exact += (torch.randn(n, generator=g) * torch.logspace(-3, 0, n)).to(torch.bfloat16).float()
    err = ((got - exact).norm() / exact.norm()).item()
    worst = ((got - exact).abs().max() / exact.abs().max()).item()
    print(f"world-8 gradient sum, wire {comm}: norm-relative error {err:.2e}, max-abs / max-abs {worst:.2e}")
    assert err < (6e-3 if comm == "bf16" else 2.5e-3) and worst < 1e-2


# ------------------------------------------------------------------ graph segments: the reducer's cut / issue / drain protocol, world 2
def _cut_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from youku_mplug_amd import engine as eng
    _stub_optimizer_kernels(_MP())
    n = 1024
    ps = [nn.Parameter(torch.zeros(n, dtype=torch.bfloat16)) for _ in range(3)]
    flat = eng.FlatParams([("head", [ps[0]]), ("block0", [ps[1]]), ("stem", [ps[2]])])
    red = eng.DPReducer(flat)

    def fill(step):
        g = torch.Generator().manual_seed(1000 * step + rank)
        flat.grads.copy_(torch.randn(flat.grads.numel(), generator=g).to(torch.bfloat16))

    # "capture": what engine.graph_step does while it records a data-parallel step -- every ready bucket and the final wait become
    # actions of the schedule; nothing touches the communicator
    actions = []
    red.capture_cut = actions.append
    red.stage_ready("head")
    red.stage_ready("head")            # announced twice: one action
    red.stage_ready("block0")
    red.finish()                       # announces what is left ("stem"), then the wait
    red.capture_cut = None
    assert actions == [("bucket", "head"), ("bucket", "block0"), ("bucket", "stem"), ("finish",)], actions
    assert not red.pending and not red.launched
    # "replays": the recorded schedule issued eagerly, twice with different gradients; against the eager protocol on the same data
    out = []
    for step in (1, 2):
        fill(step)
        for a in actions:
            if a[0] == "bucket":
                red.issue(a[1])
            else:
                red.drain()
        assert not red.pending
        replayed = flat.grads.float().clone()
        fill(step)
        for name in ("head", "block0"):
            red.stage_ready(name)
        red.finish()
        assert not red.pending and not red.launched
        out.append((replayed, flat.grads.float().clone()))
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_reducer_cut_issue_drain_protocol_world2_gloo():
    """engine.graph_step of a data-parallel step replays a chain of graph segments and issues the collectives between them from a schedule
    the reducer recorded during capture (DPReducer.capture_cut -> issue / drain).  Two gloo ranks: the recorded schedule holds every
    bucket once, in announcement order, then the wait; replaying it sums the ranks' gradients exactly as the eager protocol does."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 26500 + os.getpid() % 1500
    procs = [ctx.Process(target=_cut_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for step, (replayed, eager) in zip((1, 2), out):
        assert torch.equal(replayed, eager)
        exact = sum(torch.randn(3 * 1024, generator=torch.Generator().manual_seed(1000 * step + r)).to(torch.bfloat16).float() for r in range(2))
        assert torch.allclose(replayed[:exact.numel()], exact, atol=2e-2, rtol=2e-2)

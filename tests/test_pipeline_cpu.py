"""The real host pipelines (vision.py / gpt3.py / pretrain.py) on CPU, with the device entry points replaced by the torch
stand-ins of tests/standin_ops.py (each restating its C contract of include/mpv.h): row maps, strides, tapes, the loss
window of the decoder and the composed temporal-projection backward are host logic and are checked here against the
reference goldens (tests/golden/tiny.pt, produced by the reference's own modules) without a GPU.  The kernels themselves
are checked on the GPU (`-m gpu`) against the same goldens."""
import math
import os
import types

import pytest
import torch

import standin_ops

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def _build(monkeypatch, wseed):
    from oracle.weights import CONFIG_TINY, make_state_dict
    from youku_mplug_amd.pretrain import synthetic_model
    standin_ops.install(monkeypatch)
    model = synthetic_model(CONFIG_TINY, device="cpu")
    sd = make_state_dict(CONFIG_TINY, wseed)
    model.load_state_dict({k: v.to(torch.bfloat16) for k, v in sd.items()}, strict=True)
    return model, CONFIG_TINY


def _inputs(cfg, meta):
    from oracle.weights import make_inputs
    video, ids, mask = make_inputs(cfg, meta["batch"], meta["text_len"], seed=meta["input_seed"], ragged=meta["ragged"])
    return video.to(torch.bfloat16), types.SimpleNamespace(input_ids=ids, attention_mask=mask)


def _grads(model):
    return {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.requires_grad and p.grad is not None}


def test_pipelines_on_standins_vs_reference_golden(monkeypatch):
    g = torch.load(os.path.join(GOLD, "tiny.pt"))
    meta, f32, b16 = g["meta"], g["fp32"], g["bf16"]
    model, cfg = _build(monkeypatch, meta["weight_seed"])
    video, text = _inputs(cfg, meta)
    model.eval()
    out = model.forward_outputs(video, text)                      # full evaluation: logits / hidden / per-token losses
    sfrom, vstep = meta["logits_seq_from"], meta["logits_vocab_step"]
    hs = max(1, cfg.hidden // 256)
    for what, mine, key, floor in (("logits", out.logits[:, sfrom:, ::vstep], "logits", 1e-2),
                                   ("hidden", out.last_hidden_state[:, :, ::hs], "last_hidden_state", 2e-2),
                                   ("losses", out.losses, "losses", 1e-2)):
        e, e_ref = rel(mine, f32[key]), rel(b16[key], f32[key])
        assert math.isfinite(e) and e <= max(floor, 1.5 * e_ref), f"{what}: {e:.3e} (reference bf16 deviation {e_ref:.3e})"
    loss, _ = model(video, text)                                  # training entry: loss window + explicit backward pipelines
    assert abs(loss.item() - f32["loss"].item()) <= max(2e-3 * abs(f32["loss"].item()), 2 * abs(b16["loss"].item() - f32["loss"].item()))
    loss.backward()
    bad = []
    for n, gr in _grads(model).items():
        assert n in f32["grad_norm"], n
        e = abs(gr.norm().item() - f32["grad_norm"][n]) / (f32["grad_norm"][n] + 1e-12)
        e_ref = abs(b16["grad_norm"][n] - f32["grad_norm"][n]) / (f32["grad_norm"][n] + 1e-12)
        f = gr.reshape(-1)
        smp = f[::max(1, f.numel() // 64)][:64]
        es, es_ref = rel(smp, f32["grad_sample"][n]), rel(b16["grad_sample"][n], f32["grad_sample"][n])
        if e > max(2e-2, 3 * e_ref) or es > max(8e-2, 3 * es_ref):
            bad.append((n, e, e_ref, es, es_ref))
    assert not bad, bad[:8]


def test_trainable_decoder_gradients_on_standins(monkeypatch):
    """freeze_text_decoder: false (models/distributed_gpt3.py:91-93): the decoder's weight, bias, LayerNorm and (tied) embedding
    gradients of the explicit backward against autograd through the fp32 restatement, on the stand-ins (host logic: which saved
    activation meets which gradient, the LM-head + lookup halves of the word-embedding gradient, the position-embedding sum)."""
    from oracle import restate
    from oracle.weights import CONFIG_TINY, make_inputs, make_state_dict
    from youku_mplug_amd.pretrain import DistributedGPT3_Pretrain
    from youku_mplug_amd.gpt3 import GPT3Config
    standin_ops.install(monkeypatch)
    cfg = CONFIG_TINY
    vis = dict(img_size=cfg.img_size, patch_size=cfg.patch_size, depth=cfg.vit_depth, num_frames=cfg.num_frames, embed_dim=cfg.vit_dim,
               num_heads=cfg.vit_heads, mlp_ratio=cfg.vit_mlp_ratio, clip_model=True)
    txt = GPT3Config(vocab_size=cfg.vocab, hidden_size=cfg.hidden, ffn_hidden_size=cfg.ffn, num_hidden_layers=cfg.layers,
                     num_attention_heads=cfg.heads, max_position_embeddings=cfg.max_pos, layernorm_epsilon=cfg.gpt_ln_eps)
    model = DistributedGPT3_Pretrain({"num_learnable_token": cfg.num_queries, "_synthetic": True, "freeze_text_decoder": False},
                                     visual_cfg=vis, text_cfg=txt, device="cpu")
    sd = make_state_dict(cfg, 17)
    model.load_state_dict({k: v.to(torch.bfloat16) for k, v in sd.items()}, strict=True)
    model.eval()                                                     # dropout off: the masks are not reproducible across implementations
    assert model.text_decoder.trainable
    video, ids, mask = make_inputs(cfg, 3, 10, seed=23, ragged=True)
    text = types.SimpleNamespace(input_ids=ids, attention_mask=mask)
    loss, _ = model(video.to(torch.bfloat16), text)
    loss.backward()
    sdr = {k: v.bfloat16().float().requires_grad_(True) for k, v in sd.items()}
    ref = restate.pretrain_forward(video.bfloat16().float(), ids, mask, sdr, cfg)
    ref["loss"].backward()
    assert abs(loss.item() - ref["loss"].item()) <= 5e-3 * abs(ref["loss"].item())
    lm = "text_decoder.dist_model.language_model."
    keys = [lm + "embedding.word_embeddings.weight", lm + "embedding.position_embeddings.weight", lm + "encoder.final_layernorm.weight",
            lm + "encoder.final_layernorm.bias"]
    for i in range(cfg.layers):
        b = f"{lm}encoder.layers.{i}."
        keys += [b + n for n in ("input_layernorm.weight", "input_layernorm.bias", "post_attention_layernorm.weight", "self_attention.query_key_value.weight",
                                 "self_attention.query_key_value.bias", "self_attention.dense.weight", "self_attention.dense.bias", "mlp.dense_h_to_4h.weight",
                                 "mlp.dense_h_to_4h.bias", "mlp.dense_4h_to_h.weight", "mlp.dense_4h_to_h.bias")]
    keys += ["visual_fc.weight", "visual_encoder.blocks.0.attn.qkv.weight"]      # and the frozen-decoder gradients are still right
    params = dict(model.named_parameters())
    bad = []
    for k in keys:
        g, r = params[k].grad, sdr[k].grad
        assert g is not None and r is not None, k
        e = rel(g, r)
        if not (math.isfinite(e) and e <= 4e-2):
            bad.append((k, e))
    assert not bad, bad


def test_composed_temporal_backward_matches_two_launch_backward(monkeypatch):
    """TimeSformer.forward_features / backward_features: proj + temporal_fc as one composed projection vs the reference's two
    products, two dgrads and two wgrads -- the same loss and gradients up to bf16 rounding of the intermediate products, for
    every parameter."""
    from youku_mplug_amd import vision
    g = torch.load(os.path.join(GOLD, "tiny.pt"))
    meta = g["meta"]
    res = {}
    for compose in (True, False):
        monkeypatch.setattr(vision, "COMPOSE_TEMPORAL_OUT", compose)
        model, cfg = _build(monkeypatch, meta["weight_seed"])
        with torch.no_grad():      # the golden weights leave temporal_fc at its zero init: make the composition non-trivial
            gen = torch.Generator().manual_seed(3)
            for blk in model.visual_encoder.blocks:
                blk.temporal_fc.weight.copy_((torch.randn(blk.temporal_fc.weight.shape, generator=gen) * 0.05).to(torch.bfloat16))
                blk.temporal_fc.bias.copy_((torch.randn(blk.temporal_fc.bias.shape, generator=gen) * 0.05).to(torch.bfloat16))
                blk.temporal_attn.proj.bias.copy_((torch.randn(blk.temporal_attn.proj.bias.shape, generator=gen) * 0.05).to(torch.bfloat16))
        video, text = _inputs(cfg, meta)
        model.eval()
        loss, _ = model(video, text)
        loss.backward()
        res[compose] = (loss.item(), _grads(model))
    assert abs(res[True][0] - res[False][0]) <= 2e-3 * abs(res[False][0]), "composed forward (Wc = Wf Wp rounded) vs two products (proj(a) rounded)"
    for n, ga in res[True][1].items():
        gb = res[False][1][n]
        e = (ga - gb).abs().max().item() / (gb.abs().max().item() + 1e-12)
        assert e <= 3e-2, f"{n}: composed vs two-launch backward differ by {e:.3e}"
        en = abs(ga.norm().item() - gb.norm().item()) / (gb.norm().item() + 1e-12)
        assert en <= 5e-3, f"{n}: gradient norms differ by {en:.3e}"


def test_loss_window_is_exact_on_standins(monkeypatch):
    """gpt3.DistributedGPT3.forward_lm / backward_lm: the loss-window form (LM head, CE, top-layer projection / LN2 / MLP /
    final LayerNorm on the window rows) against the full evaluation: same loss, same per-token losses inside the window, same
    gradient of the query features."""
    model, cfg = _build(monkeypatch, 0)
    model.eval()
    dec = model.text_decoder
    from oracle.weights import make_inputs
    B, L = 3, 12
    _, ids, mask = make_inputs(cfg, B, L, seed=77, ragged=True)
    Q, H = model.num_learnable_token, model.text_width
    gen = torch.Generator().manual_seed(5)
    qf = (torch.randn(B * Q, H, generator=gen) * 0.5).to(torch.bfloat16)
    targets = torch.cat([torch.full((B, Q), 100, dtype=torch.long), ids[:, 1:], ids[:, 1:2]], dim=1)
    loss_mask = torch.cat([torch.zeros((B, Q), dtype=torch.long), mask[:, 1:]], dim=1)
    one = torch.ones((), dtype=torch.float32)
    res = {}
    for name, win in (("full", None), ("window", (Q, L))):
        tape = {}
        out = dec.forward_lm(qf, ids, targets, loss_mask, tape, loss_window=win)
        res[name] = (out["loss"].float().item(), out["losses"].float(), dec.backward_lm(tape, one).float(), out["last_hidden_state"])
    assert res["full"][3] is not None and res["window"][3] is None
    assert abs(res["full"][0] - res["window"][0]) <= 1e-6 * abs(res["full"][0]) + 1e-7
    assert torch.equal(res["full"][1][:, Q:], res["window"][1][:, Q:])
    assert res["window"][1][:, :Q].abs().max().item() == 0.0
    assert rel(res["window"][2], res["full"][2]) <= 1e-2


def _on_cpu(monkeypatch):
    standin_ops.install(monkeypatch)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    return torch.device("cpu")


def test_retrieval_itc_pipeline_on_standins(monkeypatch):
    """retrieval.DistributedGPT3_Retrieval (ITC; models/distributed_gpt3.py:817-985) through the same golden check as the GPU
    test, host pipelines on the stand-ins."""
    import test_model_gpu as t
    t.test_retrieval_itc_vs_reference_golden(_on_cpu(monkeypatch))


def test_eva_image_pipeline_on_standins(monkeypatch):
    """eva_vit.EvaVisionTransformer inside DistributedGPT3_Pretrain_Image (models/eva_vit.py, distributed_gpt3.py:229-427)."""
    import test_model_gpu as t
    t.test_eva_image_model_vs_reference_golden(_on_cpu(monkeypatch))


@pytest.mark.parametrize("kind", ["itm", "cls"])
def test_generation_cls_heads_pipeline_on_standins(monkeypatch, kind):
    """downstream.DistributedGPT3_Retrieval_Cls / DistributedGPT3_Cls (models/distributed_gpt3.py:988-1218, 431-657): both
    losses, every gradient and the train=False scores, host pipelines on the stand-ins."""
    import test_model_gpu as t
    t.test_generation_cls_heads_vs_reference_golden(_on_cpu(monkeypatch), kind)


# ------------------------------------------------------------------------------------------------ data parallel, real model
class _MP:      # minimal monkeypatch for spawned workers
    def setattr(self, obj, name, val):
        setattr(obj, name, val)


def _dp_setup(seed, mp=None):
    """mp: the patch target -- pytest's monkeypatch in the parent process (undone at test end, so later tests see the real
    device wrappers again), the undo-less _MP only inside spawned workers."""
    from oracle.weights import CONFIG_TINY, make_inputs
    from test_engine_cpu import _stub_optimizer_kernels
    from youku_mplug_amd import engine as eng
    from youku_mplug_amd.pretrain import synthetic_model
    mp = mp if mp is not None else _MP()
    standin_ops.install(mp)
    _stub_optimizer_kernels(mp)
    torch.manual_seed(seed)
    model = synthetic_model(CONFIG_TINY, device="cpu")
    with torch.no_grad():                            # the module default zeroes temporal_fc and the biases: make the composed projection's
        for blk in model.visual_encoder.blocks:      # chain rule (dWf = dWc Wp^T + d(bc) bp^T, d(bp) = Wf^T d(bc)) non-trivial
            blk.temporal_fc.weight.normal_(0, 0.05)
            blk.temporal_fc.bias.normal_(0, 0.05)
            blk.temporal_attn.proj.bias.normal_(0, 0.05)
    model.eval()                                     # the stand-ins do not model the hash dropout
    groups = eng.get_parameter_groups(model, 0.05, model.no_weight_decay(), visual_backbone_scale=True)
    engine, opt, _, _ = eng.initialize(model=model, model_parameters=groups, config=dict(lr=1e-4, clip_grad=3.0))
    video, ids, mask = make_inputs(CONFIG_TINY, 4, 8, seed=9, ragged=False)     # equal token counts: mean of rank means = batch mean
    return engine, video.to(torch.bfloat16), ids, mask


def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    engine, video, ids, mask = _dp_setup(1234 + rank)            # the reference seeds every rank differently
    launched = []
    orig = engine.reducer.stage_ready
    engine.reducer.stage_ready = lambda name: (launched.append(name), orig(name))[1]
    engine.module.visual_encoder.on_block_grads_ready = lambda bi: engine.reducer.stage_ready("stem" if bi < 0 else f"block{bi}")
    engine.module.on_stage_grads_ready = engine.reducer.stage_ready
    p0 = engine.flat.params.clone()
    sl = slice(rank * 2, rank * 2 + 2)
    text = types.SimpleNamespace(input_ids=ids[sl], attention_mask=mask[sl])
    loss, _ = engine(video[sl], text)
    engine.backward(loss)
    first = list(launched)
    engine.reducer.finish()
    q.put((rank, p0, engine.flat.grads.clone(), first, loss.item()))
    dist.barrier()
    dist.destroy_process_group()


def test_real_model_data_parallel_world2_gloo(monkeypatch):
    """The REAL tiny DistributedGPT3_Pretrain under the real MplugEngine / DPReducer on two gloo ranks (device entry points on
    the stand-ins): replicas identical after initialize() although seeded differently, buckets announced by the real backward
    pipelines in completion order (head, ViT blocks from the last to the first, stem), and the reduced gradient / world equal
    to the gradient of the whole batch in one process (run_pretrain_distributed_gpt3.py:263-274, DeepSpeed's averaged all-reduce)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 26000 + os.getpid() % 3000
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r[0]: r[1:] for r in (q.get(timeout=300) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.equal(res[0][0], res[1][0]), "parameters must be broadcast from rank 0 at initialize()"
    assert torch.equal(res[0][1], res[1][1]), "every rank holds the same reduced gradient"
    assert res[0][2] == ["head", "block1", "block0", "stem"], res[0][2]
    engine, video, ids, mask = _dp_setup(1234, monkeypatch)        # one process, whole batch, rank 0's initialisation
    assert torch.equal(engine.flat.params, res[0][0])
    loss, _ = engine(video, types.SimpleNamespace(input_ids=ids, attention_mask=mask))
    engine.backward(loss)
    assert abs(loss.item() - 0.5 * (res[0][3] + res[1][3])) <= 2e-3 * abs(loss.item())
    full, summed = engine.flat.grads.float(), res[0][1].float()
    for name, (a, b) in engine.flat.stage_slices.items():
        ga, gb = 0.5 * summed[a:b], full[a:b]
        cos = torch.dot(ga, gb) / (ga.norm() * gb.norm() + 1e-30)
        assert cos > 0.999 and abs(ga.norm() - gb.norm()) <= 2e-2 * gb.norm(), (name, cos.item(), ga.norm().item(), gb.norm().item())
    # ... and PER TENSOR (round 6: a stage-level cosine hid a gradient that was finished from an already-reduced operand -- the composed
    # projection's batched chain rule read d(bc) after its bucket had gone out: one rank-1 term counted `world` times in two small tensors)
    names = {id(p): n for n, p in engine.module.named_parameters()}
    for p_, o, n in engine.flat.slots:
        ga, gb = 0.5 * summed[o:o + n], full[o:o + n]
        if gb.norm() < 1e-6:
            assert ga.norm() < 1e-4, names[id(p_)]
            continue
        cos = torch.dot(ga, gb) / (ga.norm() * gb.norm() + 1e-30)
        assert cos > 0.995 and abs(ga.norm() - gb.norm()) <= 4e-2 * gb.norm(), (names[id(p_)], cos.item(), ga.norm().item(), gb.norm().item())


def test_entrypoint_loop_on_standins(tmp_path, monkeypatch):
    """run_pretrain_distributed_gpt3.py (the reference's entry point restated on this engine: CLI, YAML + JSON configs, epoch
    loop, per-step schedule mutation, loss all-gather / NaN guard, checkpoint + `latest` + log cadence, auto-resume) driven on
    CPU / gloo for 2 epochs x 3 steps with the device entry points on the stand-ins -- the same assertions as the GPU test
    (tests/test_entrypoint_gpu.py), dropout set to 0 in the decoder config because the stand-ins do not model it."""
    import json
    import test_entrypoint_gpu as t
    from test_engine_cpu import _stub_optimizer_kernels
    orig = t._write_configs

    def write(d, update_freq=1):
        path = orig(d, update_freq)
        cfg = json.load(open(os.path.join(d, "txt.json")))
        cfg.update(hidden_dropout=0.0, attention_dropout=0.0)
        json.dump(cfg, open(os.path.join(d, "txt.json"), "w"))
        return path
    monkeypatch.setattr(t, "_write_configs", write)
    _stub_optimizer_kernels(monkeypatch)
    monkeypatch.setenv("MASTER_PORT", str(27000 + os.getpid() % 2000))
    t.test_entrypoint_three_steps_from_yaml(tmp_path, _on_cpu(monkeypatch))
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


def test_retrieval_entrypoint_on_standins(tmp_path, monkeypatch):
    """downstream/run_retrieval_distributed_gpt3.py (train loop with padding='longest' titles, evaluation, itm_eval, checkpoint, log
    line, --evaluate_only --resume with re-fitted temporal embeddings) on CPU / gloo through the stand-ins -- the assertions of the
    GPU test (tests/test_entrypoint_gpu.py)."""
    import json
    import test_entrypoint_gpu as t
    from test_engine_cpu import _stub_optimizer_kernels
    orig = t._write_configs

    def write(d, update_freq=1):
        path = orig(d, update_freq)
        cfg = json.load(open(os.path.join(d, "txt.json")))
        cfg.update(hidden_dropout=0.0, attention_dropout=0.0)
        json.dump(cfg, open(os.path.join(d, "txt.json"), "w"))
        return path
    monkeypatch.setattr(t, "_write_configs", write)
    _stub_optimizer_kernels(monkeypatch)
    monkeypatch.setenv("MASTER_PORT", str(27500 + os.getpid() % 2000))
    t.test_retrieval_entrypoint_train_eval_resume(tmp_path, _on_cpu(monkeypatch))
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


def _finetune_entry_on_standins(tmp_path, monkeypatch, test_name, port):
    import json
    import test_entrypoint_gpu as t
    from test_engine_cpu import _stub_optimizer_kernels
    orig = t._write_configs

    def write(d, update_freq=1):
        path = orig(d, update_freq)
        cfg = json.load(open(os.path.join(d, "txt.json")))
        cfg.update(hidden_dropout=0.0, attention_dropout=0.0)          # the stand-ins do not model the hash dropout
        json.dump(cfg, open(os.path.join(d, "txt.json"), "w"))
        return path
    monkeypatch.setattr(t, "_write_configs", write)
    _stub_optimizer_kernels(monkeypatch)
    monkeypatch.setenv("MASTER_PORT", str(port + os.getpid() % 2000))
    getattr(t, test_name)(tmp_path, _on_cpu(monkeypatch))
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


def test_itm_entrypoint_on_standins(tmp_path, monkeypatch):
    """downstream/run_retrieval_distributed_gpt3_itm.py (derangement negatives, generation + matching losses, re-ranking evaluation,
    checkpoint, --evaluate_only --resume) on CPU / gloo through the stand-ins -- the assertions of the GPU test."""
    _finetune_entry_on_standins(tmp_path, monkeypatch, "test_itm_entrypoint_train_eval", 30000)


def test_cls_entrypoint_on_standins(tmp_path, monkeypatch):
    """downstream/run_cls_distributed_gpt3.py (class-name generation + cls_head losses, top-k accuracies) on the stand-ins."""
    _finetune_entry_on_standins(tmp_path, monkeypatch, "test_cls_entrypoint_train_eval", 32000)


def test_caption_entrypoint_on_standins(tmp_path, monkeypatch):
    """downstream/run_caption_distributed_gpt3.py (caption loss; beam-search evaluation, result files, metric line) on the stand-ins."""
    _finetune_entry_on_standins(tmp_path, monkeypatch, "test_caption_entrypoint_train_generate", 34000)


def test_finetune_host_pieces_on_cpu():
    import test_entrypoint_gpu as t
    t.test_itm_random_derangement_and_labels()
    t.test_synthetic_tokenizer_pair_contract()
    t.test_caption_metrics_known_answers()


def _entry_world2_worker(rank, world, port, d, which, q):
    """one rank of a two-process run of an entry point on the stand-ins (gloo); reports the final flat parameters and statistics"""
    import json
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)        # two ranks on a shared 8-core host: neither may oversubscribe it (a loaded host once ran this test into its timeout)
    from test_engine_cpu import _stub_optimizer_kernels
    import test_entrypoint_gpu as t
    mp_ = _MP()
    standin_ops.install(mp_)
    _stub_optimizer_kernels(mp_)
    sys.path.insert(0, t.ROOT)
    sys.path.insert(0, os.path.join(t.ROOT, "downstream"))
    from youku_mplug_amd import engine as eng
    seen = {}
    orig_init = eng.initialize

    def spy(**kw):
        r = orig_init(**kw)
        seen["engine"] = r[0]
        return r
    eng.initialize = spy
    out = os.path.join(d, "out")
    if which == "pretrain":
        import run_pretrain_distributed_gpt3 as entry
        entry.mpv_engine.initialize = spy
        cfg = os.path.join(d, "pretrain.yaml")
        args, config = entry.get_args(["--config", cfg, "--output_dir", out, "--bf16", "--enable_deepspeed", "--synthetic_steps", "2", "--seed", "7"])
    else:
        import run_retrieval_distributed_gpt3_itm as entry
        import finetune_common as ft
        ft.mpv_engine.initialize = spy
        cfg = os.path.join(d, "itm.yaml")
        args, config = entry.get_args(["--config", cfg, "--output_dir", out, "--bf16", "--enable_deepspeed", "--synthetic_steps", "2", "--seed", "5",
                                       "--eval_freq", "1"])
    stats = entry.main(args, config)
    e = seen["engine"]
    q.put((rank, e.flat.params.clone(), e.zero_shards, {k: v for k, v in stats.items() if isinstance(v, (int, float))}))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("which", ["pretrain", "itm"])
def test_entrypoints_world2_gloo_zero1(tmp_path, which):
    """run_pretrain_distributed_gpt3.py and downstream/run_retrieval_distributed_gpt3_itm.py as TWO gloo ranks on the stand-ins, with
    the command line's default `--zero_stage 1` (the reference's default launch, utils.py:528-529): both ranks finish, end with
    identical parameters (every rank updated only its optimizer-state partition and received the rest), the checkpoint holds one
    zero_pp_rank_<r> optimizer file per rank next to the model states, and the ITM evaluation (clips split over the ranks, score
    matrices summed) reports the same metrics on both ranks."""
    import json
    import torch.multiprocessing as mp
    import test_entrypoint_gpu as t
    d = str(tmp_path)
    t._write_configs(d)
    if which == "itm":
        t._write_finetune_config(d, "itm", "use_cls: true")
    cfg = json.load(open(os.path.join(d, "txt.json")))
    cfg.update(hidden_dropout=0.0, attention_dropout=0.0)              # the stand-ins do not model the hash dropout
    json.dump(cfg, open(os.path.join(d, "txt.json"), "w"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 36000 + os.getpid() % 3000 + (7 if which == "itm" else 0)
    procs = [ctx.Process(target=_entry_world2_worker, args=(r, 2, port, d, which, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2 * 150):                                            # (a worker that died early must not cost the whole timeout)
        try:
            r = q.get(timeout=2)
            res[r[0]] = r[1:]
        except Exception:
            if any(p.exitcode not in (None, 0) for p in procs):
                break
        if len(res) == 2:
            break
    for p in procs:
        p.join(timeout=120 if len(res) == 2 else 5)
        if p.is_alive():                                                # its peer died: it would wait in a collective for ever
            p.kill()
    assert [p.exitcode for p in procs] == [0, 0]
    assert torch.equal(res[0][0], res[1][0]), "ranks must end with identical parameters"
    shards = res[0][1]
    assert shards is not None and len(shards) == 2 and shards[0][1] == shards[1][0] and shards[1][1] == res[0][0].numel()
    last = "checkpoint-1" if which == "pretrain" else "checkpoint-0"
    files = sorted(os.listdir(os.path.join(d, "out", last)))
    assert files == ["mp_rank_00_model_states.pt", "zero_pp_rank_0_mp_rank_00_optim_states.pt", "zero_pp_rank_1_mp_rank_00_optim_states.pt"], files
    if which == "itm":
        for k in ("val_gen_r_mean", "val_cls_r_mean", "test_gen_r_mean"):
            assert res[0][2][k] == res[1][2][k], k


def test_itm_eval_on_cpu():
    import test_entrypoint_gpu as t
    t.test_itm_eval_recall_metrics()


def test_kv_cache_decode_on_standins(monkeypatch):
    """generation.DecodeState (prefill + single-token steps + beam re-order over the KV caches) against one full causal forward,
    host logic on the stand-ins (the C decode step restated in tests/standin_ops.decode_step)."""
    import test_model_gpu as t
    t.test_kv_cache_decode_matches_full_forward(_on_cpu(monkeypatch))


def test_caption_generate_on_standins(monkeypatch):
    """DistributedGPT3_Caption.generate: beam search over the KV-cache path, token-exact against the reference module's golden."""
    import test_model_gpu as t
    from youku_mplug_amd import downstream
    orig = downstream.synthetic_gencls_model

    def build(*a, **k):      # the test ends with a train-mode caption loss: the stand-ins do not model the decoder's dropout
        m = orig(*a, **k)
        m.text_decoder.config.hidden_dropout = m.text_decoder.config.attention_dropout = 0.0
        return m
    monkeypatch.setattr(downstream, "synthetic_gencls_model", build)
    t.test_caption_generate_vs_reference_golden(_on_cpu(monkeypatch))


@pytest.mark.parametrize("kind", ["itm", "cls", "caption"])
def test_gencls_models_refuse_options_only_the_pretrain_class_implements(monkeypatch, kind):
    """`connect_ln` and `freeze_text_decoder: false` are implemented in DistributedGPT3_Pretrain; the
    generation / classification pipelines of downstream.py inherit its constructor but apply neither (no visual_norm stage in
    _query_features, no decoder weight-gradient inputs in the hidden-only prompt pass): they must refuse, not train silently wrong."""
    from oracle.weights import CONFIG_TINY as c
    from youku_mplug_amd.downstream import DistributedGPT3_Caption, DistributedGPT3_Cls, DistributedGPT3_Retrieval_Cls
    from youku_mplug_amd.gpt3 import GPT3Config
    standin_ops.install(monkeypatch)
    klass = {"itm": DistributedGPT3_Retrieval_Cls, "cls": DistributedGPT3_Cls, "caption": DistributedGPT3_Caption}[kind]
    vis = dict(img_size=c.img_size, patch_size=c.patch_size, depth=c.vit_depth, num_frames=c.num_frames, embed_dim=c.vit_dim,
               num_heads=c.vit_heads, mlp_ratio=c.vit_mlp_ratio, clip_model=True)
    txt = GPT3Config(vocab_size=c.vocab, hidden_size=c.hidden, ffn_hidden_size=c.ffn, num_hidden_layers=c.layers,
                     num_attention_heads=c.heads, max_position_embeddings=c.max_pos, layernorm_epsilon=c.gpt_ln_eps)
    base = {"num_learnable_token": c.num_queries, "_synthetic": True, "use_cls": kind != "caption", "num_classes": 2}
    klass(base, visual_cfg=vis, text_cfg=txt, device="cpu")                                   # the shipped form builds
    with pytest.raises(NotImplementedError, match="connect_ln"):
        klass(base, visual_cfg=dict(vis, connect_ln=True), text_cfg=txt, device="cpu")
    with pytest.raises(NotImplementedError, match="freeze_text_decoder"):
        klass(dict(base, freeze_text_decoder=False), visual_cfg=vis, text_cfg=txt, device="cpu")


_BENCH_WORKER = r'''
import os, sys
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.set_num_threads(1)
import standin_ops, test_engine_cpu
class MP:
    def setattr(self, o, n, v): setattr(o, n, v)
standin_ops.install(MP()); test_engine_cpu._stub_optimizer_kernels(MP())
import bench
from youku_mplug_amd import gpt3
S = bench.Shapes                                    # the control flow is what is under test: tiny dims
S.img_size, S.patch_size, S.vit_dim, S.vit_depth, S.vit_heads, S.num_queries = 64, 16, 192, 2, 2, 32
S.hidden, S.layers, S.heads, S.ffn, S.vocab, S.max_pos = 256, 2, 4, 1024, 1024, 256
_init = gpt3.GPT3Config.__init__
def _no_dropout(self, *a, **k):                     # the stand-ins do not model the hash dropout
    _init(self, *a, **k); self.hidden_dropout = self.attention_dropout = 0.0
gpt3.GPT3Config.__init__ = _no_dropout
sys.argv = ["bench.py", "--gpus", os.environ["WORLD_SIZE"], "--steps", "2", "--warmup", "1", "--batch", "2", "--frames", "4", "--text-len", "8", "--_test-cpu"]
bench.main()
'''


def test_bench_control_flow_world8_gloo(tmp_path):
    """bench.py's N > 1 branch has never run with N > 1 (gpurun boxes have one GPU).  Its control flow --
    process-group bring-up, parameter broadcast, the bucketed all-reduce in every backward, barrier + sync fences on both sides of
    the timed region, MAX over ranks of the region's time, the post-run roofline steps on EVERY rank (each holds collectives), ONE
    JSON line from rank 0 and nothing on any other rank's stdout -- runs here as EIGHT gloo ranks launched the way the driver
    launches them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), on the stand-ins (`--_test-cpu`: a test hook that
    stamps the line as a TEST, tiny dims).  value must be global batch x steps / the slowest rank's time."""
    import json
    import subprocess
    import sys
    world = 8
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(_BENCH_WORKER)
    port = 26000 + os.getpid() % 3000
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        env.pop("MPV_BENCH_FORCE_DIST", None)
        env.pop("MPV_BENCH_DEVICE", None)
        procs.append(subprocess.Popen([sys.executable, str(script), root], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                      stdin=subprocess.DEVNULL))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0, (r, e[-1500:])
    for r in range(1, world):
        assert outs[r][1].strip() == "", (r, outs[r][1][:300])            # only rank 0 speaks on stdout
    lines = [ln for ln in outs[0][1].splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == world and rec["steps"] == 2 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["config"]["global_batch"] == world * 2 and rec["config"]["parallelism"] == "dp8"
    assert rec["value"] == pytest.approx(world * 2 * 2 / (rec["ms_per_step"] * 2e-3), rel=1e-3)
    assert rec["cpu_baseline"] is None                                      # rank 0 at N = 1 only
    assert rec["data"].startswith("TEST"), rec["data"]                      # a stand-in run can never pass for a measurement
    assert rec["step_mode"] == "eager" and rec["host"]["cpu_ms_per_step"] > 0
    assert rec["roofline"] is not None and rec["roofline"]["launches_per_step"] > 0

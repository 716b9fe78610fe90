"""Debug: two ranks on one GPU (gloo on device tensors): where do the eager step and graph_step's first (eager) call differ at world 2?
    python tools/probe/two_rank_debug.py"""
import os
import sys
import types

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    import youku_mplug_amd  # noqa: F401
    import test_model_gpu as t
    from oracle.weights import CONFIG_TINY, make_inputs
    from youku_mplug_amd import engine as eng
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    video, ids, mask = make_inputs(CONFIG_TINY, 4, 12, seed=300 + 10 * rank, ragged=False)
    v, text = video.to(dev).to(torch.bfloat16), types.SimpleNamespace(input_ids=ids.to(dev), attention_mask=mask.to(dev))

    def make_engine():
        model, _ = t.build(CONFIG_TINY, dev, 3)
        model.train()
        groups = eng.get_parameter_groups(model, 0.05, model.no_weight_decay(), visual_backbone_scale=True)
        e, opt, _, _ = eng.initialize(model=model, model_parameters=groups, config=dict(lr=2e-3, clip_grad=3.0))
        return e

    out = {}
    for arm in ("eager_autograd", "eager_fb", "graph1"):
        e = make_engine()
        e.enable_device_step_state()
        if arm == "eager_autograd":
            loss, _ = e(v, text)
            e.backward(loss)
            torch.cuda.synchronize()
            e.reducer.finish()
            torch.cuda.synchronize()
            g = e.flat.grads.clone()
            e.optimizer.step(grad_scale=0.5)
        elif arm == "eager_fb":
            e.reducer.hold = False
            loss = e.module.forward_backward(v, text)
            e.reducer.finish()
            torch.cuda.synchronize()
            g = e.flat.grads.clone()
            e.optimizer.step(grad_scale=0.5)
        else:
            loss = e.graph_step(v, text)
            torch.cuda.synchronize()
            g = e.flat.grads.clone()
        torch.cuda.synchronize()
        peers = [torch.zeros_like(g) for _ in range(world)]
        dist.all_gather(peers, g)
        out[arm] = (loss.item(), g, e.flat.params.clone(), torch.equal(peers[0], peers[1]), e.optimizer.sumsq.item(), dict(e.flat.stage_slices))
    rep = []
    base = out["eager_autograd"]
    for arm, (l, g, p, same, ss, sl) in out.items():
        bad = [n for n, (a, b) in sl.items() if not torch.equal(g[a:b], base[1][a:b])]
        rep.append(f"rank {rank} {arm:15s} loss {l:.6f} sumsq {ss:.6e} reduced grads identical across ranks: {same}; stages whose reduced gradient differs from eager_autograd's: {bad}; params equal: {torch.equal(p, base[2])}")
    q.put(rep)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, 2, 38123, q)) for r in range(2)]
    for p in ps:
        p.start()
    for _ in range(2):
        for ln in q.get(timeout=600):
            print(ln)
    for p in ps:
        p.join()

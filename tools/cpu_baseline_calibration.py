"""Ties bench.py's `cpu_baseline.kind: "port"` (oracle/restate.py) to the reference's OWN modules: the same full training step
(forward + backward + AdamW over the trainable parameters, bf16, config A = BASELINE.json configs[0]: 2 x (4 frames of 224^2 + 16
tokens), 1.3B dims) timed for (a) the reference modules imported unmodified from /root/reference through oracle/ref_loader.py
(+ shims for megatron_util etc.) and (b) the restatement, on the same host cores.  Build-container only (the reference tree does
not exist on the GPU box).  Usage: python tools/cpu_baseline_calibration.py [steps] > profiles/r03_cpu_baseline_calibration.txt"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_loader, restate  # noqa: E402
from oracle.weights import CONFIG_A, make_inputs, make_state_dict  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    threads = os.cpu_count() or 8
    torch.set_num_threads(threads)
    cfg = CONFIG_A
    video, ids, mask = make_inputs(cfg, 2, 16, seed=1234)
    video = video.bfloat16()

    # (a) the reference's own modules, bf16, train mode (dropout live), AdamW of optim/adamw.py's math via torch.optim.AdamW
    model, sd = ref_loader.build_reference_model(cfg, seed=0, dtype=torch.bfloat16)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-4, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.05)

    def ref_step():
        opt.zero_grad(set_to_none=True)
        loss, _, _ = ref_loader.reference_forward(model, video, ids, mask, train=True)
        loss.backward()
        opt.step()
        return loss.item()
    ref_step()
    t0 = time.time()
    for _ in range(steps):
        lr = ref_step()
    t_ref = (time.time() - t0) / steps

    # (b) the restatement (what bench.py's cpu_baseline times), same shapes / dtype / optimizer math
    sdb = {k: v.bfloat16() for k, v in make_state_dict(cfg, 0).items()}
    trainable = [k for k in sdb if not k.startswith("text_decoder.")]
    for k in trainable:
        sdb[k].requires_grad_(True)
    state = {k: (sdb[k].detach().float(), torch.zeros_like(sdb[k], dtype=torch.float32), torch.zeros_like(sdb[k], dtype=torch.float32)) for k in trainable}

    def port_step(step):
        for k in trainable:
            sdb[k].grad = None
        out = restate.pretrain_forward(video, ids, mask, sdb, cfg)
        out["loss"].backward()
        with torch.no_grad():
            for k in trainable:
                p, m, v = state[k]
                restate.adamw_step(p, sdb[k].grad.float(), m, v, step, 1e-4, 0.9, 0.999, 1e-6, 0.05)
                sdb[k].copy_(p)
        return out["loss"].item()
    port_step(1)
    t0 = time.time()
    for i in range(steps):
        lp = port_step(2 + i)
    t_port = (time.time() - t0) / steps
    print(f"config A (B=2, T=4, L=16, 1.3B dims, bf16, fwd + bwd + AdamW) on {threads} threads, {steps} timed steps each:")
    print(f"  reference modules (oracle/ref_loader.py, /root/reference unmodified + shims): {t_ref:.2f} s/step = {2 / t_ref:.3f} samples/s (loss {lr:.4f}, dropout live)")
    print(f"  restatement (oracle/restate.py, bench.py's cpu_baseline 'port', eval graph): {t_port:.2f} s/step = {2 / t_port:.3f} samples/s (loss {lp:.4f})")
    print(f"  ratio port / reference step time: {t_port / t_ref:.2f}")


if __name__ == "__main__":
    main()

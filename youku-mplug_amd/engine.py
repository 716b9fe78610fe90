"""Native data-parallel engine with the DeepSpeed-shaped surface the reference loop uses
(run_pretrain_distributed_gpt3.py:46-53,72-73,88-96,109,136-137,263-267; utils.py:379-480):

    engine, optimizer, _, _ = initialize(args=args, model=model, model_parameters=param_groups)
    loss, _ = engine(video, text); engine.backward(loss); engine.step()

MI355X-first design (not a DeepSpeed translation):
  * all trainable parameters / gradients live in ONE flat bf16 buffer each, laid out in
    backward-completion order (head -> ViT block 11 ... 0 -> stem); fp32 master/m/v are flat too.
    Parameters and their .grad are views, so the backward kernels write gradients in place.
  * a DP bucket is a contiguous slice of the flat gradient buffer: as soon as a stage's backward
    has been launched the engine issues an RCCL all-reduce of that slice (torch.distributed
    'nccl' == RCCL; ProcessGroupNCCL runs it on its own HIP stream, ordered after the compute
    stream's work so far), so every bucket overlaps the remaining backward.  The frozen
    GPT's dgrad runs first and produces no gradients, so all traffic hides under the ViT
    backward.  No ZeRO sharding: 130 M trainable params -> 1.6 GB of state.
  * step(): one sum-of-squares kernel + ONE grouped AdamW launch over the flat buffer
    (per-256-element-tile group ids select lr*lr_scale / weight decay), global-norm clip
    folded in, bf16 parameter write-back fused.
The bucketing / averaging logic is device-agnostic (CPU + gloo in tests); the optimizer
kernels exist only as HIP (no CPU fallback).
"""
from __future__ import annotations

import collections
import math
import os
import time
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import nn

TILE = 256


def get_parameter_groups(model: nn.Module, weight_decay=0.05, skip_list=(), visual_backbone_scale=False):
    """optim/optim_factory.py:219-265 restated: decay / no_decay (1-D, *.bias, skip list, names
    containing 'bias' or 'LayerNorm.weight') x optional 'visual_encoder_' lr_scale 0.1 groups."""
    groups: Dict[str, dict] = {}
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        nd = p.dim() == 1 or name.endswith(".bias") or name in skip_list or "bias" in name or "LayerNorm.weight" in name
        g = "no_decay" if nd else "decay"
        vis = visual_backbone_scale and "visual_encoder." in name and "temporal" not in name
        if vis:
            g = "visual_encoder_" + g
        if g not in groups:
            groups[g] = {"weight_decay": 0.0 if nd else weight_decay, "params": [], "lr_scale": 0.1 if vis else 1.0, "name": g}
        groups[g]["params"].append(p)
    return list(groups.values())


def default_stages(model: nn.Module) -> List[Tuple[str, List[nn.Parameter]]]:
    """Backward-completion order of a DistributedGPT3_Pretrain-shaped model; any other module is one stage."""
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    ve = getattr(model, "visual_encoder", None)
    if ve is None or not hasattr(ve, "blocks"):
        return [("all", [p for _, p in named])]
    depth = len(ve.blocks)
    stages: List[Tuple[str, List[nn.Parameter]]] = [("head", [])] + [(f"block{i}", []) for i in range(depth - 1, -1, -1)] + [("stem", [])]
    index = {name: i for i, (name, _) in enumerate(stages)}
    late = {id(p) for p in ve.late_grad_params()} if hasattr(ve, "late_grad_params") else set()
    for n, p in named:
        if id(p) in late:
            stages[index["stem"]][1].append(p)      # complete only at the end of the tower's backward (vision.TimeSformer.late_grad_params)
        elif n.startswith("visual_encoder.blocks."):
            bi = int(n.split(".")[2])
            stages[index[f"block{bi}"]][1].append(p)
        elif n.startswith("visual_encoder.norm."):
            stages[index[f"block{depth - 1}"]][1].append(p)      # final LN grads are complete before block depth-1's
        elif n.startswith("visual_encoder."):
            stages[index["stem"]][1].append(p)
        else:
            stages[index["head"]][1].append(p)
    return [s for s in stages if s[1]]


class FlatParams:
    """Flat bf16 parameter/gradient buffers (+ stage slices) with the parameters re-pointed at views."""

    def __init__(self, stages: Sequence[Tuple[str, Sequence[nn.Parameter]]], group_of: Optional[Dict[int, int]] = None,
                 dtype=None, pad_tiles_to: int = 1):
        """pad_tiles_to: the buffers end with as many padding tiles (group 255: skipped by the optimizer, gradients stay zero) as it
        takes to make the tile count a multiple of it -- ZeRO-1 cuts the buffer into `world` EQUAL runs, one all-gather shares them."""
        params = [p for _, ps in stages for p in ps]
        assert params, "no trainable parameters"
        self.device = params[0].device
        self.dtype = dtype or params[0].dtype
        off = 0
        self.slots: List[Tuple[nn.Parameter, int, int]] = []
        self.stage_slices: Dict[str, Tuple[int, int]] = {}
        for name, ps in stages:
            start = off
            for p in ps:
                n = p.numel()
                self.slots.append((p, off, n))
                off += (n + TILE - 1) // TILE * TILE
            self.stage_slices[name] = (start, off)
        quantum = TILE * max(1, int(pad_tiles_to))
        off = (off + quantum - 1) // quantum * quantum
        self.numel = off
        self.params = torch.zeros(off, dtype=self.dtype, device=self.device)
        self.grads = torch.zeros(off, dtype=self.dtype, device=self.device)
        tiles = torch.full((off // TILE,), 255, dtype=torch.uint8)
        for p, o, n in self.slots:
            self.params[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.params[o:o + n].view(p.shape)
            p.grad = self.grads[o:o + n].view(p.shape)
            if group_of is not None:
                tiles[o // TILE:(o + n + TILE - 1) // TILE] = group_of[id(p)]
        self.tile_group = tiles.to(self.device)
        # Fingerprint of the flat LAYOUT (which tensor sits where): the optimizer's fp32 state is one flat buffer in this order, so a
        # checkpoint's moments only mean anything under the layout they were saved with.  (Round 6 moved three tensors per ViT block
        # into the stem's stage -- vision.TimeSformer.late_grad_params --: same element count, other order.)
        import hashlib
        self.layout = hashlib.sha1(";".join(f"{o}:{n}:{tuple(p.shape)}" for p, o, n in self.slots).encode() + f"|{off}".encode()).hexdigest()[:16]


class DPReducer:
    """Bucketed gradient all-reduce over contiguous slices of FlatParams.grads (sum; the 1/world
    average is folded into the optimizer's grad_scale).  Device-agnostic: RCCL on GPU, gloo on CPU."""

    def __init__(self, flat: FlatParams, process_group=None, comm_dtype: Optional[str] = None):
        self.flat = flat
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.pending = []
        self.launched = set()
        self.always = False      # run the collectives even on a 1-rank group (single-GPU validation of the exchange path)
        # Wire dtype of the gradient sum.  "bf16" (default): the flat bf16 slices are reduced in place, as DeepSpeed's bf16 engine
        # communicates in the model dtype -- a ring all-reduce over 8 ranks then rounds every element up to 7 times (measured on
        # CPU, world 8: 3e-3 of the gradient norm, tests/test_engine_cpu.py::test_bf16_bucket_sum_error_world8_gloo).  "fp32"
        # (MPV_DP_COMM_DTYPE=fp32): every bucket is widened into an fp32 staging slice, summed in fp32 and rounded to bf16 once
        # (twice the wire bytes: 520 MB per step at config B, still far below the backward it hides under).
        self.comm_dtype = comm_dtype or os.environ.get("MPV_DP_COMM_DTYPE", "bf16")
        assert self.comm_dtype in ("bf16", "fp32"), self.comm_dtype
        self._stage32: Dict[str, torch.Tensor] = {}
        # MplugEngine.graph_step, while it CAPTURES a data-parallel step: a callable(action).  A bucket that becomes ready (and the
        # final wait) then does not touch the communicator: it ends the graph segment being captured, and the replay loop issues
        # the collective eagerly between two segment launches -- RCCL never runs inside a stream capture (round 5).
        self.capture_cut = None

    @property
    def active(self):
        return self.world > 1 or self.always

    def stage_ready(self, name: str):
        if getattr(self, "hold", False):
            return      # gradient accumulation: buckets go out from step(), once the window's sum is in place
        if not self.active or name in self.launched or name not in self.flat.stage_slices:
            return
        self.launched.add(name)
        if self.capture_cut is not None:
            self.capture_cut(("bucket", name))
            return
        self.issue(name)

    def issue(self, name: str):
        """the all-reduce of one stage's slice of the flat gradient buffer, asynchronous, behind the current stream's work"""
        a, b = self.flat.stage_slices[name]
        if self.comm_dtype == "fp32":
            from . import ops
            st = self._stage32.get(name)
            if st is None or st.numel() != b - a:
                st = self._stage32[name] = torch.empty(b - a, dtype=torch.float32, device=self.flat.device)
            ops.accum_f32(st, self.flat.grads[a:b], first=True)          # widen (slices start on 256-element tiles: aligned)
            self.pending.append((dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), name))
        else:
            self.pending.append((dist.all_reduce(self.flat.grads[a:b], op=dist.ReduceOp.SUM, group=self.pg, async_op=True), name))

    def finish(self):
        """Launch whatever was not announced, then make the current stream wait for every bucket."""
        if self.active:
            for name in self.flat.stage_slices:
                self.stage_ready(name)
            if self.capture_cut is not None:
                self.capture_cut(("finish",))
                self.launched.clear()
                return
            self.drain()
        self.pending.clear()
        self.launched.clear()

    def drain(self):
        """the current stream waits for every bucket in flight (fp32 wire: the sums are rounded into the bf16 gradients once)"""
        for w, name in self.pending:
            w.wait()
            if self.comm_dtype == "fp32":
                from . import ops
                a, b = self.flat.stage_slices[name]
                ops.f32_to_bf16(self._stage32[name], self.flat.grads[a:b])     # one rounding of the fp32 sum
        self.pending.clear()


class FlatAdamW:
    """torch.optim-shaped facade (param_groups with lr / lr_scale / weight_decay / betas mutated by the
    training loop every step) over the grouped HIP AdamW kernel."""

    def __init__(self, flat: FlatParams, param_groups: List[dict], lr=1e-4, betas=(0.9, 0.999), eps=1e-6, clip_grad=0.0,
                 shard: Optional[Tuple[int, int]] = None):
        """shard = (lo, hi), multiples of the 256-element tile: ZeRO stage 1 -- this rank keeps fp32 master / exp_avg / exp_avg_sq of
        flat elements [lo, hi) only and updates only those (the engine then shares the updated bf16 slices between the ranks)."""
        assert len(param_groups) <= 8
        self.flat = flat
        self.lo, self.hi = shard if shard is not None else (0, flat.numel)
        assert self.lo % TILE == 0 and (self.hi % TILE == 0 or self.hi == flat.numel) and 0 <= self.lo <= self.hi <= flat.numel
        self.param_groups = param_groups
        for g in param_groups:
            g.setdefault("lr", lr * g.get("lr_scale", 1.0))
            g.setdefault("betas", list(betas))
            g.setdefault("eps", eps)
        self.clip_grad = clip_grad
        self.master = flat.params[self.lo:self.hi].float()
        self.exp_avg = torch.zeros_like(self.master)
        self.exp_avg_sq = torch.zeros_like(self.master)
        self.sumsq = torch.zeros((), dtype=torch.float32, device=flat.device)
        self.step_count = 0
        self.cur_scale = self.loss_scale = 1.0
        self._grad_scale = 1.0
        self.hyper_dev = None      # MplugEngine.enable_device_step_state(): float32[18] device tensor

    @property
    def _global_grad_norm(self):
        return math.sqrt(max(float(self.sumsq.item()), 0.0)) * self._grad_scale

    def step(self, grad_scale: float = 1.0, upload: bool = True):
        from . import ops
        self.step_count += 1
        self._grad_scale = grad_scale
        self.sumsq.zero_()
        ops.grad_sumsq(self.flat.grads, self.sumsq)          # the norm of the WHOLE (reduced) gradient, also under ZeRO-1
        g0 = self.param_groups[0]
        if self.hi == self.lo:
            return
        p16, g16 = self.flat.params[self.lo:self.hi], self.flat.grads[self.lo:self.hi]
        tg = self.flat.tile_group[self.lo // TILE:(self.hi + TILE - 1) // TILE]
        if self.hyper_dev is not None:
            if upload:                                     # (not while a graph is being captured: the values would be frozen with it)
                self.upload_hyper()
            ops.adamw_step_grouped_dev(p16, self.master, self.exp_avg, self.exp_avg_sq, g16, tg,
                                       self.hyper_dev, float(g0["betas"][0]), float(g0["betas"][1]), float(g0["eps"]), grad_scale,
                                       self.sumsq, float(self.clip_grad or 0.0))
            return
        ops.adamw_step_grouped(p16, self.master, self.exp_avg, self.exp_avg_sq, g16, tg,
                               [float(g["lr"]) for g in self.param_groups], [float(g["weight_decay"]) for g in self.param_groups],
                               float(g0["betas"][0]), float(g0["betas"][1]), float(g0["eps"]), self.step_count, grad_scale,
                               self.sumsq, float(self.clip_grad or 0.0))

    def upload_hyper(self):
        """this step's lr / weight decay (param_groups, mutated by the training loop) and bias corrections (step_count) -> device"""
        from . import ops
        g0 = self.param_groups[0]
        ops.adamw_hyper_upload([float(g["lr"]) for g in self.param_groups], [float(g["weight_decay"]) for g in self.param_groups],
                               float(g0["betas"][0]), float(g0["betas"][1]), self.step_count, self.hyper_dev)

    def state_dict(self):
        return {"master": self.master, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "step": self.step_count,
                "shard": (self.lo, self.hi), "layout": self.flat.layout,
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        assert tuple(sd.get("shard", (0, self.flat.numel))) == (self.lo, self.hi), "optimizer state of another ZeRO partition"
        self.master.copy_(sd["master"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count = sd["step"]
        self.flat.params[self.lo:self.hi].copy_(self.master)


def _graph_queue_depth() -> int:
    return int(os.environ.get("MPV_GRAPH_QUEUE_DEPTH", "2"))


def init_process_group_for_dp(backend: Optional[str] = None, **kw):
    """torch.distributed.init_process_group with the one choice that matters for overlapping the gradient all-reduce with this
    backward: RCCL's kernels go on a HIGH-PRIORITY HIP stream.  The 256x256 GEMM workgroup owns a whole CU (160 KiB of LDS and all
    512 registers of every SIMD), so a communication workgroup can never be co-resident with one -- it gets a CU only when a tile
    retires, and with equal priority the next GEMM tile of the same launch is just as likely to take it.  At high priority the
    (few, long-lived) RCCL workgroups win that arbitration once and keep their CUs for the bucket's duration, while the GEMM grid
    flows around them.  (A static CU reservation for communication was considered and rejected: it costs its share of the chip
    for the whole step, not for the ~1-3 ms a step's buckets are on the wire.)  Falls back to default options where the backend
    has no such knob (gloo)."""
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        try:
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            return dist.init_process_group(backend, pg_options=opts, **kw)
        except (AttributeError, TypeError):
            # this torch build has no such option / keyword: nothing has been initialised yet, fall through.  A RuntimeError
            # (store, rendezvous, communicator) is a real failure of the first attempt and must surface as itself, not as the
            # "initialised twice" / "address in use" a blind retry would raise on top of it.
            pass
    return dist.init_process_group(backend, **kw)


def broadcast_module_state(model: nn.Module, flat: Optional["FlatParams"], process_group=None, src: int = 0):
    """What deepspeed.initialize does for the reference (run_pretrain_distributed_gpt3.py:210 seeds every rank
    differently, :263-267 hands the model to DeepSpeed, whose engine broadcasts it from rank 0): after this call every
    rank holds rank `src`'s parameters and buffers.  Trainable parameters travel as the one flat buffer."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return
    g_src = dist.get_global_rank(process_group, src) if process_group is not None else src
    in_flat = set()
    if flat is not None:
        dist.broadcast(flat.params, src=g_src, group=process_group)
        in_flat = {id(p) for p, _, _ in flat.slots}
    for t in list(model.parameters()) + list(model.buffers()):
        if id(t) not in in_flat:
            dist.broadcast(t.data, src=g_src, group=process_group)


class MplugEngine(nn.Module):
    def __init__(self, model: nn.Module, param_groups: List[dict], lr=1e-4, betas=(0.9, 0.999), eps=1e-6, clip_grad=0.0,
                 process_group=None, gradient_accumulation_steps: int = 1, zero_stage: int = 0):
        super().__init__()
        self.module = model
        self.gas = max(1, int(gradient_accumulation_steps))
        group_of = {id(p): gi for gi, g in enumerate(param_groups) for p in g["params"]}
        if hasattr(model, "unused_parameters"):          # never receive gradients: skip them (tile group 255), as an
            for p in model.unused_parameters():          # optimizer over .grad=None parameters would
                group_of[id(p)] = 255
        stages = default_stages(model)
        stages = [(n, [p for p in ps if id(p) in group_of]) for n, ps in stages]
        stages = [s for s in stages if s[1]]
        world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.flat = FlatParams(stages, group_of, pad_tiles_to=world if zero_stage == 1 else 1)
        broadcast_module_state(model, self.flat, process_group)     # replicas start identical (before the fp32 master copy)
        self.reducer = DPReducer(self.flat, process_group)
        # the window sum of micro-batch gradients is kept in fp32 (DeepSpeed's bf16 optimizer does the same): in bf16 every add
        # rounds to 8 mantissa bits and small contributions vanish against a large running sum
        self.grad_acc = torch.zeros(self.flat.numel, dtype=torch.float32, device=self.flat.device) if self.gas > 1 else None
        # ZeRO stage 1 (utils.py:528-529, `--zero_stage 1`): optimizer states partitioned over the data-parallel ranks.  The flat
        # buffer (padded to a multiple of `world` tiles) is cut into `world` equal runs of whole 256-element tiles; every rank still
        # receives the whole reduced gradient (the bucketed all-reduce is unchanged: the clip needs its global norm), updates its own
        # run, and ONE in-place all-gather hands the updated bf16 runs round.  (Stages 2 / 3 -- gradient / parameter partitioning --
        # are not built.)
        assert zero_stage in (0, 1), "ZeRO stages 2 and 3 are not built"
        self.zero_shards = None
        shard = None
        if zero_stage == 1 and self.reducer.world > 1:
            ntiles = (self.flat.numel + TILE - 1) // TILE
            per = (ntiles + self.reducer.world - 1) // self.reducer.world
            self.zero_shards = [(min(r * per * TILE, self.flat.numel), min((r + 1) * per * TILE, self.flat.numel)) for r in range(self.reducer.world)]
            shard = self.zero_shards[dist.get_rank(process_group)]
        self.process_group = process_group
        self.optimizer = FlatAdamW(self.flat, param_groups, lr=lr, betas=betas, eps=eps, clip_grad=clip_grad, shard=shard)
        self._micro_steps = 0                 # `micro_steps`: the training loop resets it every epoch (run_pretrain_distributed_gpt3.py:72-73)
        self._window_fill = 0                 # micro-batches summed into the current accumulation window
        self.micro_batches_seen = 0           # never reset, saved with the optimizer state: drives the dropout seed
        self.global_steps = 0
        self._upload_seeds = True
        self._graph = None                    # graph_step(): (torch.cuda.CUDAGraph, static inputs, static loss)
        self._graph_inflight = collections.deque()   # blocking-sync events of the replays still queued / running
        self._graph_calls = 0
        self._set_dropout_seed()              # rank-distinct masks from the very first micro-batch on
        ve = getattr(model, "visual_encoder", None)
        if ve is not None and hasattr(ve, "on_block_grads_ready"):
            depth = len(ve.blocks)
            ve.on_block_grads_ready = lambda bi: self.reducer.stage_ready("stem" if bi < 0 else f"block{bi}")
            assert depth > 0
        if hasattr(model, "on_stage_grads_ready"):
            model.on_stage_grads_ready = self.reducer.stage_ready

    def forward(self, *a, **k):
        return self.module(*a, **k)

    @property
    def micro_steps(self) -> int:
        return self._micro_steps

    @micro_steps.setter
    def micro_steps(self, value: int):
        """The reference's loops write `model.micro_steps = 0` at the top of every epoch (run_pretrain_distributed_gpt3.py:72-73):
        DeepSpeed's accumulation boundary is `micro_steps % gas`, so that write realigns the window and the partial window a
        ragged epoch left behind (len(loader) % update_freq != 0) is discarded with the following zero_grad().  Same here: a
        write that lands on a window boundary drops the partial fp32 sum (the next accumulate starts with first=True)."""
        self._micro_steps = int(value)
        if self._micro_steps % max(1, self.gas) == 0:
            self._window_fill = 0

    def is_gradient_accumulation_boundary(self) -> bool:
        return self._window_fill == 0

    def backward(self, loss):
        """DeepSpeed semantics (`--update_freq`, run_pretrain_distributed_gpt3.py:46-53,88-96): the gradient of a window of
        `gradient_accumulation_steps` micro-batches is their mean.  The backward kernels overwrite .grad, so micro-batch
        gradients are summed into a second flat buffer and the 1/steps factor rides in the optimizer's grad_scale; the
        overlapped bucket all-reduce only exists without accumulation (with it, the window's sum is reduced in step())."""
        self.reducer.hold = self.gas > 1
        loss.backward()
        self._micro_steps += 1
        self.micro_batches_seen += 1
        if self.gas > 1:
            from . import ops
            ops.accum_f32(self.grad_acc, self.flat.grads, first=self._window_fill == 0)
            self._window_fill = (self._window_fill + 1) % self.gas
        self._set_dropout_seed()

    def _set_dropout_seed(self):
        """Fresh, rank-distinct dropout streams for the next micro-batch: a 64-bit mix of (micro-batches seen so far, rank).
        The counter is monotonic over the whole run (`micro_steps` is reset by the loop every epoch, which would replay the same
        mask sequence each epoch and after every resume) and travels in the optimizer-state file of a checkpoint."""
        td = getattr(self.module, "text_decoder", None)
        if td is not None and hasattr(td, "step_seed"):
            rank = dist.get_rank() if dist.is_initialized() else 0
            x = (self.micro_batches_seen * 0x9E3779B97F4A7C15 + (rank + 1) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
            x ^= x >> 30
            x = (x * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
            x ^= x >> 27
            x = (x * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
            x ^= x >> 31
            td.step_seed = x & 0x3FFFFFFFFFFFFFFF      # bit 63 clear also after the per-pass stride is added (bit 63 set = indirect seed)
            if getattr(td, "seed_dev", None) is not None and self._upload_seeds:
                from . import ops
                from .gpt3 import SEED_PASS_STRIDE
                ops.store_u64(td.seed_dev, [td.step_seed + k * SEED_PASS_STRIDE for k in range(td.seed_dev.numel())])

    def step(self):
        if not self.is_gradient_accumulation_boundary():
            return                                        # mid-window micro step: nothing to apply yet
        if self.gas > 1:
            from . import ops
            ops.f32_to_bf16(self.grad_acc, self.flat.grads)     # one rounding of the fp32 window sum, then the bucketed reduce
        self.reducer.hold = False
        self.reducer.finish()
        self.optimizer.step(grad_scale=1.0 / (self.reducer.world * self.gas))
        self._share_zero_shards()
        self.global_steps += 1

    def _share_zero_shards(self):
        """ZeRO-1: every rank has updated its own run of the flat bf16 parameters; hand the runs round"""
        if self.zero_shards is None:
            return
        lo, hi = self.zero_shards[dist.get_rank(self.process_group)]
        if all(b - a == hi - lo for a, b in self.zero_shards) and self.zero_shards[-1][1] == self.flat.numel:
            # equal runs (FlatParams(pad_tiles_to=world)): in place, rank r's input IS slice r of the output
            dist.all_gather_into_tensor(self.flat.params, self.flat.params[lo:hi], group=self.process_group)
            return
        for r, (a, b) in enumerate(self.zero_shards):       # (a flat buffer built without the padding)
            if b > a:
                src = dist.get_global_rank(self.process_group, r) if self.process_group is not None else r
                dist.broadcast(self.flat.params[a:b], src=src, group=self.process_group)

    def zero_grad(self):
        pass      # every gradient is overwritten (never accumulated) by the next backward

    # ---- the step as ONE replayed HIP graph (MPV_GRAPH=1 in bench.py / the entrypoints) -------------------------------------
    def enable_device_step_state(self):
        """From here on the scalars that change from step to step are read by the kernels from DEVICE memory instead of travelling
        in the launches: the dropout seeds (include/mpv.h: MPV_SEED_FROM_DEVICE) and AdamW's learning rates / weight decays / bias
        corrections (mpv_adamw_step_grouped_dev).  The host writes them before each step with mpv_store_words.  Same values, same
        arithmetic: an eager step in this mode is bit-identical to one without it -- and it is the precondition for capturing the
        step into a graph, whose kernel arguments are frozen."""
        td = getattr(self.module, "text_decoder", None)
        dev = self.flat.device
        if td is not None and hasattr(td, "step_seed") and getattr(td, "seed_dev", None) is None:
            td.seed_dev = torch.zeros(4, dtype=torch.int64, device=dev)
        if self.optimizer.hyper_dev is None:
            self.optimizer.hyper_dev = torch.zeros(18, dtype=torch.float32, device=dev)
        self._set_dropout_seed()

    def graph_step(self, *inputs):
        """forward + backward + optimizer step of one micro-batch; from the third call on a replay of a captured HIP graph.
        Call 1 runs eagerly (every lazily created buffer and kernel attribute exists afterwards), call 2 captures the step and
        replays it, later calls copy the inputs into the captured buffers, write the step's scalars and replay.  Shapes must
        not change between calls (pre-training: they do not).  Returns the loss tensor of the step (a static buffer from call
        2 on).  No gradient accumulation.

        Data parallel (world > 1, or the reducer's `always` switch): the step is captured as a CHAIN of graph segments cut where a
        gradient bucket becomes ready and where the step waits for the buckets; a replay launches the segments in order and issues the
        collectives EAGERLY between them (same order, same streams and events as the eager step: ProcessGroupNCCL orders each
        all-reduce behind the segment just launched, `wait()` orders the next segment behind the reduction).  RCCL inside a stream
        capture (round 4's MPV_GRAPH_DP=1) took the process down in hipStreamEndCapture once in nine suite runs; it is gone.  Host
        cost of a replay: ~16 graph launches + 14 all-reduce calls instead of one launch."""
        assert self.gas == 1, "graph_step: no gradient accumulation"
        assert self.zero_shards is None or not self.reducer.active, "graph_step: ZeRO-1 shares its shards with a collective after the optimizer step; use the eager step"
        self.enable_device_step_state()
        self._graph_calls += 1

        assert hasattr(self.module, "forward_backward"), "graph_step needs the model's autograd-free forward_backward()"

        def run(args):
            self.reducer.hold = False
            loss = self.module.forward_backward(*args)       # (not module(...) + loss.backward(): see forward_backward)
            self.reducer.finish()
            self.optimizer.step(grad_scale=1.0 / self.reducer.world, upload=False)
            self._share_zero_shards()
            return loss

        def after():
            self._micro_steps += 1
            self.micro_batches_seen += 1
            self.global_steps += 1
            self._set_dropout_seed()           # next step's seeds -> device (stream-ordered behind this step)

        def clone_in(x):
            if torch.is_tensor(x):
                return x.clone()
            if hasattr(x, "__dict__"):
                return type(x)(**{k: clone_in(v) for k, v in vars(x).items()})
            return x

        def copy_in(dst, src):
            if torch.is_tensor(dst):
                if dst.shape != src.shape or dst.dtype != src.dtype:
                    raise ValueError(f"graph_step: input {tuple(src.shape)} {src.dtype} differs from the captured {tuple(dst.shape)} {dst.dtype}; "
                                     "a captured step replays fixed shapes (pad the batch, or use the eager step for this one)")
                dst.copy_(src, non_blocking=True)
            elif hasattr(dst, "__dict__"):
                for k, v in vars(dst).items():
                    copy_in(v, getattr(src, k))

        if self._graph_calls == 1:
            self.optimizer.step_count += 1
            self.optimizer.upload_hyper()
            self.optimizer.step_count -= 1
            loss = run(inputs)
            after()
            return loss
        if self._graph is None:
            assert self.module.training, "graph_step captures the training step (model.train())"
            static_in = tuple(clone_in(x) for x in inputs)
            counters = (self.optimizer.step_count,)
            torch.cuda.synchronize()
            self._upload_seeds = False
            segs = []                                        # [(graph, actions after it)]: ("bucket", stage) | ("finish",)
            try:
                if not self.reducer.active:
                    g = torch.cuda.CUDAGraph()
                    # thread_local: a DataLoader's pin-memory thread (hipHostMalloc / event calls) must not invalidate the capture
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        static_loss = run(static_in)
                    segs.append((g, []))
                else:
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    # every segment allocates from ONE private pool (the segments are replayed in capture order, never concurrently)
                    from . import _lib
                    state = {"g": None, "open": False, "pool": torch.cuda.graph_pool_handle(), "calls": 0}

                    def begin():
                        state["g"] = torch.cuda.CUDAGraph()
                        state["g"].capture_begin(pool=state["pool"], capture_error_mode="thread_local")
                        state["open"], state["calls"] = True, _lib.n_calls

                    def cut(action):
                        # Only the final wait may ride on the previous segment (it follows the last bucket with nothing in between).  A
                        # BUCKET always ends the segment being captured: "nothing launched since the last cut" can only be told from
                        # the C ABI's call counter, which does not see a torch-native kernel a stage may have captured (a .copy_ into a
                        # gradient view) -- merging such a bucket would issue its all-reduce before those writes at replay.
                        if segs and action[0] == "finish" and _lib.n_calls == state["calls"]:
                            segs[-1][1].append(action)
                            return
                        state["g"].capture_end()
                        state["open"] = False
                        segs.append((state["g"], [action]))
                        begin()
                    with torch.cuda.stream(side):
                        self.reducer.capture_cut = cut
                        try:
                            begin()
                            static_loss = run(static_in)
                            state["g"].capture_end()
                            state["open"] = False
                            segs.append((state["g"], []))
                        finally:
                            self.reducer.capture_cut = None
                            if state["open"]:            # an exception inside the step: close the capture before the graph object dies
                                try:                     # (destroying a graph whose stream still captures terminates the process)
                                    state["g"].capture_end()
                                except Exception:
                                    pass
                    torch.cuda.current_stream().wait_stream(side)
            finally:
                self._upload_seeds = True
            (self.optimizer.step_count,) = counters          # the capture executed nothing
            self._graph = (segs, static_in, static_loss)
        segs, static_in, static_loss = self._graph
        # Bounded host run-ahead.  hipGraphLaunch on a FULL launch queue busy-waits (measured r03: 72 ms of CPU per 77 ms step,
        # where the eager path sleeps in the driver) -- on a host shared by eight ranks that spin is what starves the others.  So
        # the host launches step k + 1 only once step k + 1 - depth has FINISHED, and waits for that ASLEEP: a query / sleep poll
        # (measured r04 call 1: hipEventSynchronize on a blocking-sync event still spins on this stack -- 72.4 ms of CPU per
        # 75.3 ms step -- so the wait is time.sleep between hipEventQuery calls; 0.2 ms of granularity against a 75 ms step, with a
        # whole step still queued behind the running one).  depth 2 is all the device needs (a replay is one launch);
        # MPV_GRAPH_QUEUE_DEPTH=0 restores the unbounded queue.
        depth = _graph_queue_depth()
        if depth > 0:
            while len(self._graph_inflight) >= depth:
                ev = self._graph_inflight.popleft()
                while not ev.query():
                    time.sleep(2e-4)
        for d, s_ in zip(static_in, inputs):
            copy_in(d, s_)
        self.optimizer.step_count += 1
        self.optimizer.upload_hyper()                        # this step's lr / bias corrections -> device
        for g, actions in segs:
            g.replay()
            for action in actions:
                if action[0] == "bucket":
                    self.reducer.issue(action[1])
                else:
                    self.reducer.drain()
        if depth > 0:
            ev = torch.cuda.Event()
            ev.record()
            self._graph_inflight.append(ev)
        after()
        return static_loss

    # ---- start-up self-check of graph replay against the eager step (bench.py, the entry points) ------------------------------
    def snapshot_state(self):
        """Everything a step changes: the flat bf16 parameters, the optimizer's fp32 state and the counters (the dropout stream's
        position among them).  Gradients and activations are overwritten by the next step and are not part of it."""
        o = self.optimizer
        return {"params": self.flat.params.clone(), "master": o.master.clone(), "exp_avg": o.exp_avg.clone(),
                "exp_avg_sq": o.exp_avg_sq.clone(), "step_count": o.step_count,
                "counters": (self._micro_steps, self._window_fill, self.micro_batches_seen, self.global_steps)}

    def restore_state(self, snap):
        o = self.optimizer
        self.flat.params.copy_(snap["params"])
        o.master.copy_(snap["master"])
        o.exp_avg.copy_(snap["exp_avg"])
        o.exp_avg_sq.copy_(snap["exp_avg_sq"])
        o.step_count = snap["step_count"]
        self._micro_steps, self._window_fill, self.micro_batches_seen, self.global_steps = snap["counters"]
        self._set_dropout_seed()

    def graph_self_check(self, *inputs, steps: int = 3, before_step=None):
        """Is a replayed step the eager step?  From ONE state: `steps` eager steps (device-resident step scalars, as the replay reads
        them), rewind, the same steps through graph_step (call 1 eager, call 2 capture + replay, later calls replay), rewind again.
        Returns (ok, report): ok only if every loss and the final flat parameters are BIT-identical on every rank (the verdict is
        the AND over the process group -- all ranks must take the same mode or none may).  On a data-parallel group this is the
        chain-of-segments replay with the bucket all-reduces between the segments, run on the real communicator: the check a
        2+ rank box never got in the test suite runs here, at start-up, on the ranks that are about to train.  An exception while
        capturing (a stack that cannot capture) counts as a failed check; the eager step keeps working after it.
        `before_step(i)`: the loop's per-step host work (learning-rate mutation), applied identically in both arms."""
        assert steps >= 3, "the third graph_step call is the first pure replay"
        snap = self.snapshot_state()
        self.enable_device_step_state()

        def fingerprint(losses):
            return [float(l) for l in losses], self.flat.params.clone()

        losses = []
        for i in range(steps):
            if before_step is not None:
                before_step(i)
            out = self.module(*inputs)
            loss = out[0] if isinstance(out, (tuple, list)) else out
            self.backward(loss)
            self.step()
            losses.append(loss.detach().float().item())
        ref_losses, ref_params = fingerprint(losses)
        self.restore_state(snap)
        ok, why = True, "bit-identical"
        try:
            losses = []
            for i in range(steps):
                if before_step is not None:
                    before_step(i)
                losses.append(self.graph_step(*inputs).detach().float().item())
            got_losses, got_params = fingerprint(losses)
            if got_losses != ref_losses:
                ok, why = False, f"losses differ: eager {ref_losses} vs replay {got_losses}"
            elif not torch.equal(got_params, ref_params):
                n = int((got_params != ref_params).sum().item())
                ok, why = False, f"{n} of {ref_params.numel()} parameters differ after {steps} steps"
        except Exception as e:                                  # capture refused by the stack: eager mode is what is left
            ok, why = False, f"graph capture failed: {type(e).__name__}: {e}"
            self._graph = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.process_group) > 1:
            flag = torch.tensor([1 if ok else 0], device=self.flat.device, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.process_group)
            if ok and flag.item() == 0:
                ok, why = False, "another rank's self-check failed"
        if torch.device(self.flat.device).type == "cuda":
            torch.cuda.synchronize()
        self._graph_inflight.clear()
        self.restore_state(snap)
        return ok, why

    # ---- DeepSpeed-layout checkpoints: <dir>/<tag>/mp_rank_00_model_states.pt with key 'module' (utils.py:476-480)
    def save_checkpoint(self, save_dir, tag=None, client_state=None):
        tag = tag or f"global_step{self.global_steps}"
        d = os.path.join(save_dir, str(tag))
        rank0 = not dist.is_initialized() or dist.get_rank() == 0
        if rank0:
            os.makedirs(d, exist_ok=True)
            state = {"module": {k: v.detach().cpu() for k, v in self.module.state_dict().items()},
                     # the dropout stream's position travels with the MODEL file: on a ZeRO-1 resume at a larger world size the new
                     # ranks have no optimizer shard to read it from, and every rank must continue the same mask sequence
                     "micro_batches_seen": self.micro_batches_seen}
            state.update(client_state or {})
            torch.save(state, os.path.join(d, "mp_rank_00_model_states.pt"))
            if self.zero_shards is None:
                osd = dict(self.optimizer.state_dict(), micro_batches_seen=self.micro_batches_seen)
                torch.save({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in osd.items()}, os.path.join(d, "mp_rank_00_optim_states.pt"))
        if dist.is_initialized():
            dist.barrier()                     # the directory exists for every rank
        if self.zero_shards is not None:       # DeepSpeed's ZeRO layout: one optimizer-state file per data-parallel rank
            osd = dict(self.optimizer.state_dict(), micro_batches_seen=self.micro_batches_seen)
            torch.save({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in osd.items()}, os.path.join(d, self._zero_file()))
            dist.barrier()                     # every rank's shard is on disk ...
        if rank0:                              # ... before `latest` names the checkpoint: a crash in between leaves the previous one current
            with open(os.path.join(save_dir, "latest"), "w") as f:
                f.write(str(tag))
        if dist.is_initialized():
            dist.barrier()
        return True

    def _zero_file(self):
        return f"zero_pp_rank_{dist.get_rank(self.process_group)}_mp_rank_00_optim_states.pt"

    def load_checkpoint(self, load_dir, tag=None):
        if tag is None:
            with open(os.path.join(load_dir, "latest")) as f:
                tag = f.read().strip()
        d = os.path.join(load_dir, str(tag))
        state = torch.load(os.path.join(d, "mp_rank_00_model_states.pt"), map_location="cpu")
        from .vision import resize_visual_embeds_in_state_dict
        sd = resize_visual_embeds_in_state_dict(state.pop("module"), self.module)     # other resolution / frame count (downstream --resume)
        missing, unexpected = self.module.load_state_dict(sd, strict=False)
        if unexpected:
            raise KeyError(f"checkpoint {d} holds keys this model does not have: {sorted(unexpected)[:8]}")
        self.last_load_missing_keys = list(missing)
        op = os.path.join(d, self._zero_file() if self.zero_shards is not None else "mp_rank_00_optim_states.pt")
        osd = torch.load(op, map_location=self.flat.device) if os.path.isfile(op) else None
        # the dropout stream's position is read BEFORE an unusable shard is discarded (a resume at another world size restarts the
        # moments, not the mask sequence)
        seen_osd = state.pop("micro_batches_seen", None)          # (the model file: the same for every rank)
        if seen_osd is None and osd is not None:                  # checkpoints written before round 6 kept it per optimizer file
            seen_osd = osd.get("micro_batches_seen")
        # (a state saved under another flat layout -- or before layouts were recorded -- has the right SIZE and the wrong order: unusable)
        usable = osd is not None and osd.get("layout") == self.flat.layout and osd["master"].numel() == self.optimizer.master.numel() and \
            tuple(osd.get("shard", (0, self.flat.numel))) == (self.optimizer.lo, self.optimizer.hi)
        if self.zero_shards is not None:
            # every rank must take the SAME branch (one rank resuming its Adam moments while another restarts from zero would run
            # different bias corrections on shards of one model): the decision is the AND over ranks, and it is said out loud
            flags = [None] * dist.get_world_size(self.process_group)
            dist.all_gather_object(flags, bool(usable), group=self.process_group)
            if not all(flags) and any(flags) and dist.get_rank(self.process_group) == 0:
                print(f"load_checkpoint: optimizer shards of {d} are missing or of another partition on ranks "
                      f"{[r for r, f in enumerate(flags) if not f]} (saved at another world size?): ALL ranks restart the optimizer state")
            usable = all(flags)
        elif osd is not None and not usable:
            print(f"load_checkpoint: optimizer state of {d} does not fit this model's flat buffer (resized embeddings, or saved under another flat layout / before layouts were recorded): fresh optimizer state")
        if usable:
            self.optimizer.load_state_dict(osd)
        else:       # weights only, or a checkpoint of another shape (resized embeddings): fresh optimizer state, as the
            self.optimizer.master.copy_(self.flat.params[self.optimizer.lo:self.optimizer.hi].float())     # downstream scripts build a new optimizer after --resume
            self.optimizer.exp_avg.zero_()
            self.optimizer.exp_avg_sq.zero_()
            self.optimizer.step_count = 0
            osd = None if self.zero_shards is not None else osd
        # a resume starts a fresh accumulation window (a NaN auto-resume may arrive mid-window with a stale partial sum) and
        # continues the dropout stream where the checkpoint left it
        self._window_fill = 0
        self.micro_steps = 0
        self.reducer.pending.clear()
        self.reducer.launched.clear()
        if seen_osd is not None:
            self.micro_batches_seen = int(seen_osd)
        self._set_dropout_seed()
        return d, state


def initialize(args=None, model=None, model_parameters=None, dist_init_required=None, mpu=None, config=None, **kw):
    """deepspeed.initialize-shaped entry (run_pretrain_distributed_gpt3.py:263-267)."""
    cfg = dict(config or {})

    def pick(name, default):
        if name in cfg:
            return cfg[name]
        return getattr(args, name, default) if args is not None else default

    groups = list(model_parameters) if model_parameters is not None else get_parameter_groups(model, pick("weight_decay", 0.05))
    groups = [g if isinstance(g, dict) else {"params": [g], "weight_decay": 0.0, "lr_scale": 1.0} for g in groups]
    gas = pick("gradient_accumulation_steps", None) or pick("update_freq", 1) or 1
    zero = pick("zero_stage", 0) or 0                                                     # utils.py:528-529 (`--zero_stage`)
    if isinstance(cfg.get("zero_optimization"), dict):                                    # ... or the DeepSpeed config it is copied into
        zero = cfg["zero_optimization"].get("stage", zero)
    engine = MplugEngine(model, groups, lr=pick("lr", 1e-4), betas=tuple(pick("opt_betas", (0.9, 0.999))), eps=pick("opt_eps", 1e-6),
                         clip_grad=pick("clip_grad", 0.0) or 0.0, process_group=kw.get("process_group"),
                         gradient_accumulation_steps=gas, zero_stage=int(zero))
    return engine, engine.optimizer, None, None


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0.0, warmup_steps=-1,
                     sched_type="cos"):
    """Per-iteration schedule table the training loop indexes every step (utils.py:350-372): linear warm-up from
    `start_warmup_value` to `base_value`, then half-cosine (or linear) decay to `final_value`."""
    total = int(epochs * niter_per_ep)
    warm = int(warmup_steps if warmup_steps > 0 else warmup_epochs * niter_per_ep)
    out = []
    for i in range(warm):
        out.append(start_warmup_value + (base_value - start_warmup_value) * (i / (warm - 1) if warm > 1 else 1.0))
    n = total - warm
    for i in range(n):
        if sched_type in ("cos", "cosine"):
            out.append(final_value + 0.5 * (base_value - final_value) * (1.0 + math.cos(math.pi * i / n)))
        elif sched_type == "linear":
            out.append(base_value + (final_value - base_value) * (i / (n - 1) if n > 1 else 0.0))
        else:
            raise NotImplementedError(sched_type)
    assert len(out) == total
    return out

"""What the reference's four fine-tuning entry points (downstream/run_{retrieval,retrieval_itm,cls,caption}_distributed_gpt3.py)
repeat verbatim, stated once on this engine: the per-step protocol of train_one_epoch (schedule tables written into
optimizer.param_groups, the cross-rank loss all-gather with the NaN / Inf guard, `loss /= update_freq`, engine.backward /
engine.step, the statistics a log line carries), the command line, the way a YAML + its visual JSON become (args, config), the
`--resume` of a pre-training checkpoint, and the epoch loop's checkpoint / log cadence.  An entry point supplies what differs:
how a batch becomes model inputs and named losses, and its evaluation.

Synthetic runs (`--synthetic_steps N`; no dataset, tokenizer model or opencv on the box) use SyntheticTextTokenizer -- one token
per character, the same call contract as DistributedGPT3Tokenizer incl. [prompt, text] pairs with `prompt_lengths`
(models/modeling_distributed_gpt3.py:180-319) -- so the loops below run the reference's own string templates unchanged."""
import argparse
import datetime
import json
import os
import random
import sys
import time
import types
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import youku_mplug_amd  # noqa: E402,F401
from run_pretrain_distributed_gpt3 import _ConfigLoader, init_distributed  # noqa: E402
from youku_mplug_amd import engine as mpv_engine  # noqa: E402
from youku_mplug_amd.vision import resize_visual_embeds_in_state_dict  # noqa: E402


class _Encoding(dict):
    """BatchEncoding's two habits the loops rely on: attribute access and `.to(device)`."""

    __getattr__ = dict.__getitem__

    def to(self, device):
        return _Encoding({k: (v.to(device) if torch.is_tensor(v) else v) for k, v in self.items()})


class SyntheticTextTokenizer:
    """One token per character (CJK block -> 5 + code point offset, anything else folded into the same range), <sep> = 1 opens a
    text and <|endoftext|> = 0 closes and pads it, as JiebaBPETokenizer's bos / eos / pad do (:61-65).  Strings: bos + tokens + eos,
    padded to the longest of the batch (capped by max_length) or to max_length (:236-274).  [prompt, text] pairs: bos + prompt +
    text + eos padded to max_length in both modes, `prompt_lengths` = tokens of the prompt, over-long pairs cut the prompt first
    and the target only when the prompt cannot give enough (:209-233, 276-317)."""
    BASE = 5

    def __init__(self, vocab_size=51200):
        self.vocab = int(vocab_size)
        self.tokenizer = types.SimpleNamespace(bos=1, eos=0, pad=0)

    def _ids(self, s):
        span = self.vocab - self.BASE
        return [self.BASE + ((ord(c) - 0x4E00) % span) for c in s]

    def decode(self, tokens, **_):
        if torch.is_tensor(tokens):
            tokens = tokens.detach().cpu().tolist()
        return "".join(chr(0x4E00 + t - self.BASE) for t in tokens if t >= self.BASE)

    def __call__(self, data, padding="longest", truncation=True, max_length=None, return_tensors="pt", add_special_tokens=True, **_):
        tk = self.tokenizer
        rows, masks, plens = [], [], []
        if isinstance(data[0], str):
            toks = [([tk.bos] if add_special_tokens else []) + self._ids(s) + ([tk.eos] if add_special_tokens else []) for s in data]
            longest = max(len(t) for t in toks)
            if padding == "max_length":
                L = max_length
            else:
                L = min(longest, max_length) if truncation and max_length is not None else longest
            for t in toks:
                t = t[:L]
                rows.append(t + [tk.pad] * (L - len(t)))
                masks.append([1] * len(t) + [0] * (L - len(t)))
        else:
            pairs = [(self._ids(p), self._ids(t)) for p, t in data]
            longest = max(len(p) + len(t) + 2 for p, t in pairs)
            L = max_length if (truncation or padding == "max_length") else longest
            for p, t in pairs:
                if len(p) + len(t) + 2 >= L:
                    room = L - len(t) - 2
                    if len(p) >= room >= 0:
                        p = p[:room]
                    else:
                        t = t[:L - 2 - len(p)]
                seq = [tk.bos] + p + t + [tk.eos]
                n = min(len(seq), L)
                rows.append(seq[:L] + [tk.pad] * (L - n))
                masks.append([1] * n + [0] * (L - n))
                plens.append(len(p))
        out = _Encoding(input_ids=torch.tensor(rows, dtype=torch.long), attention_mask=torch.tensor(masks, dtype=torch.long))
        if plens:
            out["prompt_lengths"] = torch.tensor(plens, dtype=torch.long)
        return out


def synthetic_titles(n, max_chars, generator, alphabet=400):
    """n random titles of 3..max_chars characters from the first `alphabet` code points of the CJK block."""
    lens = torch.randint(3, max_chars + 1, (n,), generator=generator)
    return ["".join(chr(0x4E00 + int(c)) for c in torch.randint(0, alphabet, (int(l),), generator=generator)) for l in lens]


class SyntheticSplit:
    """Batches of synthetic clips with whatever per-sample fields an entry point's datasets carry.  `fields(i)` -> tuple of the
    sample's non-video fields; `collate(columns)` turns the per-field lists of a batch into what the reference's collate yields."""

    def __init__(self, n, batch_size, frames, res, seed, fields, collate=None, drop_last=False):
        self.n, self.bs, self.shape, self.seed = n, max(1, batch_size), (3, frames, res, res), seed
        self.fields, self.collate, self.drop_last = fields, collate, drop_last
        self.dataset = self
        self.sampler = types.SimpleNamespace(set_epoch=lambda e: None)

    def __len__(self):
        return self.n // self.bs if self.drop_last else (self.n + self.bs - 1) // self.bs

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed + 1)
        for b in range(len(self)):
            ids = list(range(b * self.bs, min(self.n, (b + 1) * self.bs)))
            video = torch.randn((len(ids),) + self.shape, generator=g)
            cols = list(zip(*[self.fields(i) for i in ids]))
            yield (video,) + tuple(self.collate(c) if self.collate else c for c in cols)


def write_schedules(optimizer, it, lr_schedule_values, wd_schedule_values):
    """The step-level lr / weight-decay tables -> optimizer.param_groups (every entry point's train_one_epoch, e.g.
    run_retrieval_distributed_gpt3_itm.py:96-105)."""
    for group in optimizer.param_groups:
        if lr_schedule_values is not None:
            group["lr"] = lr_schedule_values[it] * group["lr_scale"]
        if wd_schedule_values is not None and group["weight_decay"] > 0:
            group["weight_decay"] = wd_schedule_values[it]


def train_one_epoch(model, data_loader, optimizer, device, epoch, num_training_steps_per_epoch, step_fn, update_freq=1, start_steps=0,
                    lr_schedule_values=None, wd_schedule_values=None, args=None, log=print):
    """step_fn(batch) -> {name: loss tensor}; the step's loss is their sum (`loss_generation + loss_cls`, :142).  Everything else
    is the protocol the four loops share (:72-224): zero_grad + micro_steps reset, schedule write, loss all-gather + NaN / Inf guard
    (auto-resume from the last checkpoint when there is one, else exit), `loss /= update_freq`, backward, step, statistics."""
    model.train()
    model.zero_grad()
    model.micro_steps = 0
    sums, count = {}, 0
    world = dist.get_world_size()
    for data_iter_step, batch in enumerate(data_loader):
        t0 = time.time()
        step = data_iter_step // update_freq
        if step >= num_training_steps_per_epoch:
            continue
        it = start_steps + step
        write_schedules(optimizer, it, lr_schedule_values, wd_schedule_values)
        losses = step_fn(batch)
        loss = sum(losses.values())
        loss_value = loss.item()
        gathered = [torch.zeros_like(loss) for _ in range(world)]
        dist.all_gather(gathered, loss.detach())
        all_losses = torch.stack([g.float() for g in gathered])
        if torch.isnan(all_losses).any() or torch.isinf(all_losses).any():
            log(f" ========== non-finite loss on some rank at iteration {it}: {all_losses.tolist()} ========== ")
            if args is not None and args.output_dir and getattr(args, "auto_resume_iter", False) and os.path.isfile(os.path.join(args.output_dir, "latest")):
                model.load_checkpoint(args.output_dir)
                continue
            raise SystemExit(1)
        model.backward(loss / update_freq)
        model.step()
        grad_norm = optimizer._global_grad_norm
        if device.type == "cuda":
            torch.cuda.synchronize()
        lrs = [g["lr"] for g in optimizer.param_groups]
        stats = dict(loss=loss_value, **{k: v.item() for k, v in losses.items()}, all_loss_mean=all_losses.mean().item(), lr=max(lrs),
                     min_lr=min(lrs), grad_norm=grad_norm, loss_scale=optimizer.cur_scale, time=time.time() - t0,
                     weight_decay=max([g["weight_decay"] for g in optimizer.param_groups] + [0.0]))
        for k, v in stats.items():
            sums[k] = sums.get(k, 0.0) + float(v)
        count += 1
        if data_iter_step % 10 == 0:
            log(f"Epoch: [{epoch}] [{data_iter_step}/{len(data_loader)}] " + "  ".join(f"{k}: {v:.6g}" for k, v in stats.items()))
    return {k: v / max(count, 1) for k, v in sums.items()}


def load_resume_state(model, path):
    """`--resume` (e.g. run_retrieval_distributed_gpt3_itm.py:402-420): a pre-training checkpoint under key 'model' (.pth) or
    'module' (DeepSpeed layout); position / temporal embeddings refitted to this model's patch grid and frame count; strict=False
    (the fine-tuning heads are new)."""
    checkpoint = torch.load(path, map_location="cpu")
    state_dict = checkpoint["model"] if "model" in checkpoint else checkpoint["module"]
    state_dict = resize_visual_embeds_in_state_dict(dict(state_dict), model)
    own = model.state_dict()
    state_dict = {k: (v.to(own[k].dtype) if k in own and torch.is_tensor(v) else v) for k, v in state_dict.items()}
    msg = model.load_state_dict(state_dict, strict=False)
    print("load checkpoint from %s" % path)
    print(msg)
    return msg


def get_args(argv=None, extra=None, config_defaults=None):
    """The command line the four entry points share (:527-560) + `--synthetic_steps`; YAML `optimizer` / `schedular` sections fill
    the arguments the command line left unset (:572-581); the visual JSON's img_size / num_frames / clip_model go into the config
    (:586-590).  `extra(parser)` adds an entry point's own flags, `config_defaults` its forced config entries (:591)."""
    p = argparse.ArgumentParser()
    p.add_argument("--config", default="./configs/Pretrain.yaml")
    p.add_argument("--output_dir", default="Pretrain/")
    p.add_argument("--log_dir", default=None)
    p.add_argument("--device", default="cuda")
    p.add_argument("--seed", default=42, type=int)
    p.add_argument("--world_size", default=1, type=int)
    p.add_argument("--local_rank", default=-1, type=int)
    p.add_argument("--dist_url", default="env://")
    p.add_argument("--distributed", default=True, type=bool)
    p.add_argument("--resume", default=None)
    p.add_argument("--auto_resume", action="store_true")
    p.add_argument("--auto_resume_iter", action="store_true")
    p.add_argument("--no_auto_resume", action="store_false", dest="auto_resume")
    p.set_defaults(auto_resume=True, auto_resume_iter=True)
    p.add_argument("--update_freq", default=1, type=int)
    p.add_argument("--bf16", action="store_true")
    p.add_argument("--save_ckpt_freq", default=1, type=int)
    p.add_argument("--enable_deepspeed", action="store_true", default=False)
    p.add_argument("--zero_stage", default=1, type=int)
    p.add_argument("--evaluate_only", action="store_true", default=False)
    p.add_argument("--synthetic_steps", default=0, type=int, help="run on N synthetic batches per epoch instead of the dataset")
    if extra is not None:
        extra(p)
    args, _ = p.parse_known_args(argv)
    config = yaml.load(open(args.config, "r"), Loader=_ConfigLoader)
    Path(args.output_dir).mkdir(parents=True, exist_ok=True)
    for section in ("optimizer", "schedular"):
        for name, val in (config.get(section) or {}).items():
            if getattr(args, name, None) is None:
                setattr(args, name, val)
    args.max_length, args.batch_size, args.num_workers = config["max_length"], config["batch_size"], config.get("num_workers", 0)
    vis = json.load(open(config["visual_cfg"], "r"))
    config["image_res"] = vis["img_size"]
    config["num_frames"] = config.get("num_frames", vis["num_frames"])
    config["clip_model"] = vis.get("clip_model", False)
    config["visual_config"] = vis
    config.update(config_defaults or {})
    if args.log_dir is None:
        args.log_dir = os.path.join(args.output_dir, "tensorboard_logs")
    yaml.safe_dump(config, open(os.path.join(args.output_dir, "config.yaml"), "w"))
    return args, config


def setup(args, config, real_data):
    """Process group, seeds (`args.seed + rank`, :354-358) and the tokenizer; on real-data runs also the loaders (`real_data()` ->
    (loaders, tokenizer) from the reference's own `dataset` package), on synthetic runs the entry point fills s.loaders itself."""
    init_distributed(args)
    device = torch.device(args.device if torch.cuda.is_available() else "cpu")
    seed = args.seed + dist.get_rank()
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    s = types.SimpleNamespace(device=device, seed=seed)
    if args.synthetic_steps > 0:
        text_cfg = json.load(open(config["text_cfg"], "r")) if os.path.isfile(str(config.get("text_cfg", ""))) else {}
        s.vocab = int(text_cfg.get("vocab_size", 51200))
        s.tokenizer = SyntheticTextTokenizer(s.vocab)
        s.loaders = None
    else:
        try:
            s.loaders, s.tokenizer = real_data()
        except ImportError as e:
            raise SystemExit(f"real-data runs need the reference's `dataset` package and tokenizer on PYTHONPATH ({e}); "
                             "use --synthetic_steps N for synthetic clips")
    return s


def build_engine(args, config, model_cls, s):
    """model, `--resume`, parameter groups, engine and schedule tables for the loaders in s.loaders (:386-455)."""
    s.steps_per_epoch = len(s.loaders[0])   # as the reference (:383): the tables cover len(data_loader) steps per epoch whatever --update_freq is
    model = model_cls(config=config, tokenizer=s.tokenizer, device=s.device)
    s.n_parameters = sum(p.numel() for p in model.parameters() if p.requires_grad)
    print("number of params (B):", s.n_parameters / 1e9)
    if args.resume:
        load_resume_state(model, args.resume)
    groups = mpv_engine.get_parameter_groups(model, config["optimizer"]["weight_decay"], model.no_weight_decay(),
                                             visual_backbone_scale=config.get("clip_model", False))
    s.model, s.optimizer, _, _ = mpv_engine.initialize(args=args, model=model, model_parameters=groups)
    s.lr_values = mpv_engine.cosine_scheduler(args.lr, args.min_lr, args.epochs, s.steps_per_epoch, warmup_epochs=getattr(args, "warmup_epochs", 0),
                                              warmup_steps=getattr(args, "warmup_steps", -1), sched_type=getattr(args, "lr_sched_type", "cos"))
    s.wd_values = mpv_engine.cosine_scheduler(args.weight_decay, args.weight_decay, args.epochs, s.steps_per_epoch)
    return s


def epoch_loop(args, s, step_fn, after_epoch):
    """The epoch loop of main() (:472-541): set_epoch, train_one_epoch, checkpoint every save_ckpt_freq epochs and after the last,
    `after_epoch(epoch) -> dict` of evaluation statistics (may be empty), one log.txt line on rank 0."""
    t_start = time.time()
    log_stats = {}
    for epoch in range(0, args.epochs):
        s.loaders[0].sampler.set_epoch(epoch)
        train_stats = train_one_epoch(s.model, s.loaders[0], s.optimizer, s.device, epoch, s.steps_per_epoch, step_fn, update_freq=args.update_freq,
                                      start_steps=epoch * s.steps_per_epoch, lr_schedule_values=s.lr_values, wd_schedule_values=s.wd_values, args=args)
        if args.output_dir and ((epoch + 1) % args.save_ckpt_freq == 0 or epoch + 1 == args.epochs):
            s.model.save_checkpoint(save_dir=args.output_dir, tag=f"checkpoint-{epoch}", client_state={"epoch": epoch})
        log_stats = {**{f"train_{k}": v for k, v in train_stats.items()}, **after_epoch(epoch), "epoch": epoch, "n_parameters": s.n_parameters}
        write_log(args, log_stats)
    print("Training time {}".format(str(datetime.timedelta(seconds=int(time.time() - t_start)))))
    return log_stats


def write_log(args, log_stats):
    if args.output_dir and dist.get_rank() == 0:
        with open(os.path.join(args.output_dir, "log.txt"), "a", encoding="utf-8") as f:
            f.write(json.dumps(log_stats) + "\n")

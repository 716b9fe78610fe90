"""Video-text retrieval fine-tuning / evaluation on the MI355X-native path -- drop-in for the reference's
downstream/run_retrieval_distributed_gpt3.py (same command line, same YAML / JSON configs, same per-step protocol and the
same evaluation numbers), with DistributedGPT3_Retrieval and the engine coming from youku_mplug_amd (BASELINE.json configs[4]).

Kept from the reference because downstream tooling depends on it:
  * train_one_epoch (:107-240): step-level lr / weight-decay tables written into optimizer.param_groups, titles tokenised with
    padding='longest' (the text length changes from batch to batch), bf16 video cast, `model(video, text, idx)`, the cross-rank
    loss all-gather with the NaN / Inf guard, `loss /= update_freq`, engine.backward / engine.step, grad-norm read-out;
  * evaluation (:245-293): text features of the whole split in chunks of 32 titles padded to max(64, max_length), video features
    batch by batch, the two similarity matrices as numpy arrays;
  * itm_eval (:296-339): recall@1/5/10 both ways from the similarity matrices and the dataset's txt2vid / vid2txt tables;
  * main (:342-520): `--resume` of a pre-training checkpoint ('model' or 'module' key) with the position / temporal embeddings
    refitted to this run's resolution and frame count (:402-420), `--evaluate_only`, one DeepSpeed-layout checkpoint, one
    evaluation pass over val and test and one log.txt line per epoch.
Data: `--synthetic_steps N` runs on synthetic clips / titles (no datasets on the box); otherwise the reference's own `dataset`
package and tokenizer are imported from PYTHONPATH, untouched."""
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
from youku_mplug_amd.retrieval import DistributedGPT3_Retrieval  # noqa: E402
from youku_mplug_amd.vision import resize_visual_embeds_in_state_dict  # noqa: E402


class SyntheticTokenizer:
    """Stand-in for DistributedGPT3Tokenizer on synthetic runs: a title is a tuple of token ids; padding='longest' pads to the
    longest title of the batch, padding='max_length' to max_length (the two modes the reference loop / evaluation use)."""

    def __call__(self, texts, padding="longest", truncation=True, max_length=80, return_tensors="pt", **_):
        rows = [list(t)[:max_length] for t in texts]
        L = max_length if padding == "max_length" else max(len(r) for r in rows)
        ids = torch.zeros((len(rows), L), dtype=torch.long)
        mask = torch.zeros((len(rows), L), dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = torch.tensor(r, dtype=torch.long)
            mask[i, :len(r)] = 1
        out = types.SimpleNamespace(input_ids=ids, attention_mask=mask)
        out.to = lambda device: types.SimpleNamespace(input_ids=ids.to(device), attention_mask=mask.to(device))
        return out


class SyntheticRetrievalSet:
    """`n` (clip, title) pairs with the attributes the reference's retrieval datasets expose: iteration yields training batches
    (video, titles, idx) or evaluation batches (video, ids); `.text`, `.txt2vid`, `.vid2txt` serve evaluation / itm_eval."""

    def __init__(self, n, batch_size, frames, res, max_length, vocab, seed, train):
        g = torch.Generator().manual_seed(seed)
        self.n, self.bs, self.shape, self.train, self.seed = n, batch_size, (3, frames, res, res), train, seed
        lens = torch.randint(3, max_length + 1, (n,), generator=g)
        self.text = [tuple(torch.randint(5, vocab, (int(l),), generator=g).tolist()) for l in lens]
        self.txt2vid = {i: i for i in range(n)}
        self.vid2txt = {i: [i] for i in range(n)}
        self.dataset = self
        self.sampler = types.SimpleNamespace(set_epoch=lambda e: None)

    def __len__(self):
        return (self.n + self.bs - 1) // self.bs if not self.train else self.n // self.bs

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed + 1)
        for b in range(len(self)):
            ids = list(range(b * self.bs, min(self.n, (b + 1) * self.bs)))
            video = torch.randn((len(ids),) + self.shape, generator=g)
            if self.train:
                yield video, [self.text[i] for i in ids], torch.tensor(ids)
            else:
                yield video, torch.tensor(ids)


def train_one_epoch(model, tokenizer, data_loader, optimizer, device, epoch, num_training_steps_per_epoch, update_freq=1,
                    start_steps=0, lr_schedule_values=None, wd_schedule_values=None, args=None, log=print):
    model.train()
    model.zero_grad()
    model.micro_steps = 0
    sums, count = {}, 0
    world = dist.get_world_size()
    for data_iter_step, (video, text, idx) in enumerate(data_loader):
        t0 = time.time()
        step = data_iter_step // update_freq
        if step >= num_training_steps_per_epoch:
            continue
        it = start_steps + step
        if lr_schedule_values is not None or wd_schedule_values is not None:
            for group in optimizer.param_groups:
                if lr_schedule_values is not None:
                    group["lr"] = lr_schedule_values[it] * group["lr_scale"]
                if wd_schedule_values is not None and group["weight_decay"] > 0:
                    group["weight_decay"] = wd_schedule_values[it]
        video = video.to(device, non_blocking=True).bfloat16()
        text_input = tokenizer(text, padding="longest", truncation=True, max_length=args.max_length, return_tensors="pt").to(device)
        idx = idx.to(device, non_blocking=True)
        loss_ita = model(video, text_input, idx)
        loss = loss_ita
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
        loss = loss / update_freq
        model.backward(loss)
        model.step()
        grad_norm = optimizer._global_grad_norm
        if device.type == "cuda":
            torch.cuda.synchronize()
        lrs = [g["lr"] for g in optimizer.param_groups]
        stats = dict(loss=loss_value, loss_ita=loss_value, all_loss_mean=all_losses.mean().item(), lr=max(lrs), min_lr=min(lrs),
                     grad_norm=grad_norm, loss_scale=optimizer.cur_scale, time=time.time() - t0, text_len=text_input.input_ids.shape[1],
                     weight_decay=max([g["weight_decay"] for g in optimizer.param_groups] + [0.0]))
        for k, v in stats.items():
            sums[k] = sums.get(k, 0.0) + float(v)
        count += 1
        if data_iter_step % 10 == 0:
            log(f"Epoch: [{epoch}] [{data_iter_step}/{len(data_loader)}] " + "  ".join(f"{k}: {v:.6g}" for k, v in stats.items()))
    return {k: v / max(count, 1) for k, v in sums.items()}


@torch.no_grad()
def evaluation(model, data_loader, tokenizer, device, config):
    """-> (sims_v2t [videos, texts], sims_t2v [texts, videos]) as numpy arrays (downstream/run_retrieval_distributed_gpt3.py:245-293)."""
    module = getattr(model, "module", model)
    module.eval()
    t0 = time.time()
    texts = data_loader.dataset.text
    text_embeds = []
    for i in range(0, len(texts), 32):
        text_input = tokenizer(texts[i:i + 32], padding="max_length", truncation=True, max_length=max(64, config["max_length"]),
                               return_tensors="pt").to(device)
        text_embeds.append(module.extract_text_feature(text_input).float())
    text_embeds = torch.cat(text_embeds, dim=0)
    v2t, t2v = [], []
    for video, _vid in data_loader:
        video_embed = module.extract_vision_feature(video.to(device).bfloat16()).float()
        v2t.append(video_embed @ text_embeds.t())
        t2v.append(text_embeds @ video_embed.t())
    print("Evaluation time {}".format(str(datetime.timedelta(seconds=int(time.time() - t0)))))
    return torch.cat(v2t, dim=0).cpu().numpy(), torch.cat(t2v, dim=1).cpu().numpy()


def itm_eval(scores_i2t, scores_t2i, txt2img, img2txt):
    """Recall@1/5/10 of video->text and text->video retrieval (downstream/run_retrieval_distributed_gpt3.py:296-339)."""
    ranks = np.zeros(scores_i2t.shape[0])
    for index, score in enumerate(scores_i2t):
        inds = np.argsort(score)[::-1]
        ranks[index] = min(np.where(inds == i)[0][0] for i in img2txt[index])
    tr1, tr5, tr10 = (100.0 * len(np.where(ranks < k)[0]) / len(ranks) for k in (1, 5, 10))
    ranks = np.zeros(scores_t2i.shape[0])
    for index, score in enumerate(scores_t2i):
        inds = np.argsort(score)[::-1]
        ranks[index] = np.where(inds == txt2img[index])[0][0]
    ir1, ir5, ir10 = (100.0 * len(np.where(ranks < k)[0]) / len(ranks) for k in (1, 5, 10))
    tr_mean, ir_mean = (tr1 + tr5 + tr10) / 3, (ir1 + ir5 + ir10) / 3
    return {"txt_r1": tr1, "txt_r5": tr5, "txt_r10": tr10, "txt_r_mean": tr_mean, "vid_r1": ir1, "vid_r5": ir5, "vid_r10": ir10,
            "vid_r_mean": ir_mean, "r_mean": (tr_mean + ir_mean) / 2}


def load_resume_state(model, path):
    """`--resume` (:402-420): a pre-training checkpoint under key 'model' (.pth) or 'module' (DeepSpeed layout); position and
    temporal embeddings are refitted to this model's patch grid / frame count; strict=False (the ITC heads are new)."""
    checkpoint = torch.load(path, map_location="cpu")
    state_dict = checkpoint["model"] if "model" in checkpoint else checkpoint["module"]
    state_dict = resize_visual_embeds_in_state_dict(dict(state_dict), model)
    own = model.state_dict()
    state_dict = {k: (v.to(own[k].dtype) if k in own and torch.is_tensor(v) else v) for k, v in state_dict.items()}
    msg = model.load_state_dict(state_dict, strict=False)
    print("load checkpoint from %s" % path)
    print(msg)
    return msg


def main(args, config):
    init_distributed(args)
    device = torch.device(args.device if torch.cuda.is_available() else "cpu")
    seed = args.seed + dist.get_rank()
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    visual_cfg = json.load(open(config["visual_cfg"], "r"))
    if args.synthetic_steps > 0:
        text_cfg = json.load(open(config["text_cfg"], "r")) if os.path.isfile(str(config.get("text_cfg", ""))) else {}
        vocab = int(text_cfg.get("vocab_size", 51200))
        mk = lambda n, s, train: SyntheticRetrievalSet(n, config["batch_size"], config["num_frames"], visual_cfg["img_size"], config["max_length"], vocab, s, train)
        data_loader = mk(args.synthetic_steps * args.update_freq * config["batch_size"], seed, True)
        val_loader, test_loader = mk(2 * config["batch_size"] + 1, 1001, False), mk(config["batch_size"] + 3, 1002, False)
        tokenizer = SyntheticTokenizer()
    else:
        try:
            from dataset import create_dataset, create_loader, create_sampler
            from models.modeling_distributed_gpt3 import DistributedGPT3Tokenizer
        except ImportError as e:
            raise SystemExit(f"real-data runs need the reference's `dataset` package and tokenizer on PYTHONPATH ({e}); "
                             "use --synthetic_steps N for synthetic clips")
        datasets = create_dataset("video_retrieval", config)
        samplers = create_sampler(datasets, [True], dist.get_world_size(), dist.get_rank()) + [None, None]
        data_loader, val_loader, test_loader = create_loader(datasets, samplers, batch_size=[args.batch_size] * 3, num_workers=[args.num_workers] * 3,
                                                             is_trains=[True, False, False], collate_fns=[None, None, None])
        tokenizer = DistributedGPT3Tokenizer(config["text_decoder"])
    steps_per_epoch = len(data_loader)      # as the reference (:383): the tables cover len(data_loader) steps per epoch whatever --update_freq is
    model = DistributedGPT3_Retrieval(config=config, tokenizer=tokenizer, device=device)
    n_parameters = sum(p.numel() for p in model.parameters() if p.requires_grad)
    print("number of params (B):", n_parameters / 1e9)
    if args.resume:
        load_resume_state(model, args.resume)
    groups = mpv_engine.get_parameter_groups(model, config["optimizer"]["weight_decay"], model.no_weight_decay(),
                                             visual_backbone_scale=config.get("clip_model", False))
    model, optimizer, _, _ = mpv_engine.initialize(args=args, model=model, model_parameters=groups)
    model.module.process_group = None          # ITC features are all-gathered over the default (data-parallel) group
    lr_values = mpv_engine.cosine_scheduler(args.lr, args.min_lr, args.epochs, steps_per_epoch, warmup_epochs=getattr(args, "warmup_epochs", 0),
                                            warmup_steps=getattr(args, "warmup_steps", -1), sched_type=getattr(args, "lr_sched_type", "cos"))
    wd_values = mpv_engine.cosine_scheduler(args.weight_decay, args.weight_decay, args.epochs, steps_per_epoch)

    def evaluate(loader, name):
        v2t, t2v = evaluation(model, loader, tokenizer, device, config)
        stats = {"sim_{}".format(k): v for k, v in itm_eval(v2t, t2v, loader.dataset.txt2vid, loader.dataset.vid2txt).items()}
        print(f"{name} Performance:", stats)
        return stats

    if args.evaluate_only:
        return {"val": evaluate(val_loader, "Validation"), "test": evaluate(test_loader, "Test")}
    t_start = time.time()
    log_stats = {}
    for epoch in range(0, args.epochs):
        data_loader.sampler.set_epoch(epoch)
        train_stats = train_one_epoch(model, tokenizer, data_loader, optimizer, device, epoch, steps_per_epoch, update_freq=args.update_freq,
                                      start_steps=epoch * steps_per_epoch, lr_schedule_values=lr_values, wd_schedule_values=wd_values, args=args)
        if args.output_dir and ((epoch + 1) % args.save_ckpt_freq == 0 or epoch + 1 == args.epochs):
            model.save_checkpoint(save_dir=args.output_dir, tag=f"checkpoint-{epoch}", client_state={"epoch": epoch})
        val_stats, test_stats = evaluate(val_loader, "Validation"), evaluate(test_loader, "Test")
        log_stats = {**{f"train_{k}": v for k, v in train_stats.items()}, **{f"val_{k}": v for k, v in val_stats.items()},
                     **{f"test_{k}": v for k, v in test_stats.items()}, "epoch": epoch, "n_parameters": n_parameters}
        if args.output_dir and dist.get_rank() == 0:
            with open(os.path.join(args.output_dir, "log.txt"), "a", encoding="utf-8") as f:
                f.write(json.dumps(log_stats) + "\n")
    print("Training time {}".format(str(datetime.timedelta(seconds=int(time.time() - t_start)))))
    return log_stats


def get_args(argv=None):
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
    config["num_frames"] = config.get("num_frames", vis["num_frames"])                  # :596: the YAML may override the visual config
    config["clip_model"] = vis.get("clip_model", False)
    config["visual_config"] = vis
    if args.log_dir is None:
        args.log_dir = os.path.join(args.output_dir, "tensorboard_logs")
    yaml.safe_dump(config, open(os.path.join(args.output_dir, "config.yaml"), "w"))
    return args, config


if __name__ == "__main__":
    main(*get_args())

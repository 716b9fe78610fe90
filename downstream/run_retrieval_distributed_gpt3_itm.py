"""Video-text matching (ITM) fine-tuning / re-ranking evaluation on the MI355X-native path -- drop-in for the reference's
downstream/run_retrieval_distributed_gpt3_itm.py (same command line, YAML / JSON configs, per-step protocol), with
DistributedGPT3_Retrieval_Cls and the engine coming from youku_mplug_amd.

What this entry point adds to downstream/finetune_common.py:
  * the training batch (:109-131): two random derangements of the batch give 2 x B negative titles per step, a negative whose
    video id equals the anchor's is labelled a match, every (video, title) pair becomes the generation sample
    ["标题：<title> 这个视频与标题匹配吗？", "是" | "否"] (prompt length masked out of the caption loss) and the bare title feeds the
    classification pass; `model(video, text, prompt_text, negative_indices, labels)` -> (loss_generation, loss_cls);
  * evaluation (:228-286): every video against every title of the split in chunks of 8 titles with the answer fixed to "是" --
    the generation score (minus the summed caption loss) and the matching probability of cls_head fill two [videos, texts] score
    matrices, each ranked both ways by itm_eval (recall@1/5/10).
Two deliberate differences, both only visible where the reference misbehaves: the evaluation statistics of the classification
scores are logged under `cls_*` (the reference overwrites the `gen_*` entries with them, :459-462), and with more than one rank
the evaluation loaders are left unsharded and their batches are split over the ranks and summed (the reference places a rank's
rows at rank * (videos // world + 1) although its DistributedSampler hands every rank a STRIDED shard, :246-250, :372; its own
launch script says to evaluate on one process)."""
import os
import random
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import finetune_common as ft  # noqa: E402
from run_retrieval_distributed_gpt3 import itm_eval  # noqa: E402
from youku_mplug_amd.downstream import DistributedGPT3_Retrieval_Cls  # noqa: E402

QUESTION = "标题：{} 这个视频与标题匹配吗？"
ANSWER = {0: "否", 1: "是"}
TEXT_BS = 8


def random_derangement(n):
    """A uniformly random permutation without fixed points, by early refusal (:42-53): build a Fisher-Yates shuffle from the top,
    start over as soon as an element would stay in place.  Draws from `random` in the reference's order, so a seeded run picks the
    same negatives.  (n = 1 has no derangement: the reference loops for ever, this raises.)"""
    if n < 2:
        raise ValueError("a batch of one title has no negative to pair it with")
    while True:
        v = list(range(n))
        ok = True
        for j in range(n - 1, -1, -1):
            p = random.randint(0, j)
            if v[p] == j:
                ok = False
                break
            v[j], v[p] = v[p], v[j]
        if ok and v[0] != 0:
            return v


def make_training_batch(video, text, idx, tokenizer, device, max_length):
    """(:109-131) -> (video, text_input, prompt_text_input, negative_indices, labels)"""
    bz = len(text)
    negative_indices = random_derangement(bz) + random_derangement(bz)
    ids = [int(i) for i in idx]
    negative_labels = [1 if ids[i % bz] == ids[n] else 0 for i, n in enumerate(negative_indices)]
    labels = torch.tensor([1] * bz + negative_labels, dtype=torch.long)
    text_all = list(text) + [text[n] for n in negative_indices]
    input_text = [[QUESTION.format(x[:max_length - 20]), ANSWER[la]] for x, la in zip(text_all, labels.tolist())]
    text_input = tokenizer(input_text, padding="max_length", truncation=True, max_length=max_length, return_tensors="pt").to(device)
    prompt_text_input = tokenizer(text_all, padding="max_length", truncation=True, max_length=max_length, return_tensors="pt").to(device)
    return video.to(device, non_blocking=True).bfloat16(), text_input, prompt_text_input, negative_indices, labels.to(device)


@torch.no_grad()
def evaluation(model, data_loader, tokenizer, device, config, max_length):
    """-> (gen_v2t, gen_t2v, cls_v2t, cls_t2v) numpy score matrices (:228-286)."""
    module = getattr(model, "module", model)
    module.eval()
    texts = data_loader.dataset.text
    num_text, num_video = len(texts), len(data_loader.dataset.video)
    world, rank = dist.get_world_size(), dist.get_rank()
    gen = torch.zeros((num_video, num_text), device=device)
    cls = torch.zeros((num_video, num_text), device=device)
    start = 0
    for b, (video, _vid) in enumerate(data_loader):
        if b % world == rank:
            video = video.to(device).bfloat16()
            gens, clss = [], []
            for i in range(0, num_text, TEXT_BS):
                chunk = texts[i:i + TEXT_BS]
                input_text = [[QUESTION.format(t[:max_length - 20]), ANSWER[1]] for _ in range(len(video)) for t in chunk]
                prompt_text = list(chunk) * len(video)
                text_input = tokenizer(input_text, padding="max_length", truncation=True, max_length=max_length, return_tensors="pt").to(device)
                prompt_text_input = tokenizer(prompt_text, padding="max_length", truncation=True, max_length=max_length, return_tensors="pt").to(device)
                g, c = module(video, text_input, prompt_text_input, train=False)
                gens.append(g.float())
                clss.append(c.float())
            gen[start:start + len(video)] = torch.cat(gens, dim=1)
            cls[start:start + len(video)] = torch.cat(clss, dim=1)
        start += len(video)
    if world > 1:
        dist.all_reduce(gen)
        dist.all_reduce(cls)
    return gen.cpu().numpy(), gen.t().cpu().numpy(), cls.cpu().numpy(), cls.t().cpu().numpy()


def synthetic_loaders(args, config, seed):
    """(clip, title, idx) training batches; (clip, ids) evaluation batches with .text / .video / .txt2vid / .vid2txt"""
    bs, frames, res = config["batch_size"], config["num_frames"], config["image_res"]

    def split(n, s, train):
        titles = ft.synthetic_titles(n, max(4, config["max_length"] // 4), torch.Generator().manual_seed(s))
        if train:
            sp = ft.SyntheticSplit(n, bs, frames, res, s, lambda i: (titles[i], i), drop_last=True)
        else:
            sp = ft.SyntheticSplit(n, bs, frames, res, s, lambda i: (i,))
        sp.text, sp.video = titles, list(range(n))
        sp.txt2vid, sp.vid2txt = {i: i for i in range(n)}, {i: [i] for i in range(n)}
        return sp
    return split(args.synthetic_steps * args.update_freq * bs, seed, True), split(bs + 1, 1001, False), split(bs // 2 + 2, 1002, False)


def real_loaders(args, config):
    from dataset import create_dataset, create_loader, create_sampler
    from models.modeling_distributed_gpt3 import DistributedGPT3Tokenizer
    datasets = create_dataset("video_retrieval", config)
    # evaluation loaders stay unsharded (every rank walks the whole split; `evaluation` scores the batches b % world == rank)
    samplers = create_sampler(datasets[:1], [True], dist.get_world_size(), dist.get_rank()) + [None, None]
    loaders = create_loader(datasets, samplers, batch_size=[args.batch_size] * 3, num_workers=[args.num_workers] * 3,
                            is_trains=[True, False, False], collate_fns=[None, None, None])
    return loaders, DistributedGPT3Tokenizer(config["text_decoder"])


def main(args, config):
    s = ft.setup(args, config, lambda: real_loaders(args, config))
    if s.loaders is None:
        s.loaders = synthetic_loaders(args, config, s.seed)
    ft.build_engine(args, config, DistributedGPT3_Retrieval_Cls, s)
    data_loader, val_loader, test_loader = s.loaders

    def step_fn(batch):
        loss_generation, loss_cls = s.model(*make_training_batch(*batch, s.tokenizer, s.device, args.max_length))
        return {"loss_generation": loss_generation, "loss_cls": loss_cls}

    def evaluate(loader, name):
        gen_v2t, gen_t2v, cls_v2t, cls_t2v = evaluation(s.model, loader, s.tokenizer, s.device, config, args.max_length)
        ds = loader.dataset
        stats = {f"gen_{k}": v for k, v in itm_eval(gen_v2t, gen_t2v, ds.txt2vid, ds.vid2txt).items()}
        stats.update({f"cls_{k}": v for k, v in itm_eval(cls_v2t, cls_t2v, ds.txt2vid, ds.vid2txt).items()})
        print(f"{name} Performance:", stats)
        return stats

    if args.evaluate_only:
        return {"val": evaluate(val_loader, "Validation")}

    def after_epoch(epoch):                                     # :510-527: val + test every `eval_freq` epochs (11 in the reference)
        if (epoch + 1) % args.eval_freq != 0:
            return {}
        val, test = evaluate(val_loader, "Validation"), evaluate(test_loader, "Test")
        return {**{f"val_{k}": v for k, v in val.items()}, **{f"test_{k}": v for k, v in test.items()}}
    return ft.epoch_loop(args, s, step_fn, after_epoch)


def get_args(argv=None):
    return ft.get_args(argv, extra=lambda p: p.add_argument("--eval_freq", default=11, type=int, help="evaluate every N epochs (:510)"),
                       config_defaults={"num_classes": 2})


if __name__ == "__main__":
    main(*get_args())

"""Video classification fine-tuning / evaluation on the MI355X-native path -- drop-in for the reference's
downstream/run_cls_distributed_gpt3.py (same command line, YAML / JSON configs, per-step protocol, accuracy numbers), with
DistributedGPT3_Cls and the engine coming from youku_mplug_amd.

What this entry point adds to downstream/finetune_common.py:
  * the training batch (:94-99): the generation sample of a clip is ["视频标题：<title> 视频类目：", <name of its class>] (prompt
    length masked out of the caption loss), the bare title feeds the classification pass;
    `model(video, text, prompt_text, labels)` -> (loss_generation, loss_cls);
  * evaluation (:205-247): every clip against EVERY class name through the generation pass (softmax over the classes of minus
    the summed caption loss) and once through cls_head; top-1 / top-5 accuracy of both, averaged over the split with batch-size
    weights and over the ranks;
  * main (:293-297, 428-466): evaluation batches of a tenth of the training batch size, val + test after every epoch."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import finetune_common as ft  # noqa: E402
from youku_mplug_amd.downstream import DistributedGPT3_Cls  # noqa: E402

PROMPT = "视频标题：{} 视频类目："


def make_training_batch(video, text, labels, idx2label, tokenizer, device, max_length):
    """(:94-99) -> (video, text_input, prompt_text_input, labels)"""
    labels = torch.as_tensor(labels, dtype=torch.long)
    input_text = [[PROMPT.format(x[:max_length - 15]), idx2label[la]] for x, la in zip(text, labels.tolist())]
    text_input = tokenizer(input_text, padding="max_length", truncation=True, max_length=max_length, return_tensors="pt").to(device)
    prompt_text_input = tokenizer(list(text), padding="max_length", truncation=True, max_length=max_length, return_tensors="pt").to(device)
    return video.to(device, non_blocking=True).bfloat16(), text_input, prompt_text_input, labels.to(device)


def topk_accuracy(output, target, topk=(1,)):
    """precision@k in percent (:250-263)"""
    pred = output.topk(max(topk), dim=1, largest=True, sorted=True).indices
    hit = pred.eq(target.view(-1, 1))
    return [hit[:, :k].any(dim=1).float().sum().item() * 100.0 / target.shape[0] for k in topk]


@torch.no_grad()
def evaluation(model, data_loader, tokenizer, device, config, max_length):
    """-> {gen_top1_accuracy, gen_top5_accuracy[, cls_top1_accuracy, cls_top5_accuracy]} (:205-247)"""
    module = getattr(model, "module", model)
    module.eval()
    class_names = [data_loader.dataset.idx2label[i] for i in range(config["num_classes"])]
    sums, weight = {}, 0.0
    for video, text, labels in data_loader:
        video = video.to(device, non_blocking=True).bfloat16()
        labels = torch.as_tensor(labels, dtype=torch.long)
        input_text = [[PROMPT.format(t[:max_length - 15]), c] for t in text for c in class_names]
        text_input = tokenizer(input_text, padding="max_length", truncation=True, max_length=max_length, return_tensors="pt").to(device)
        prompt_text_input = tokenizer(list(text), padding="max_length", truncation=True, max_length=max_length, return_tensors="pt").to(device)
        generation_logits, cls_logits = module(video, text_input, prompt_text_input, train=False)
        n = video.shape[0]
        acc = dict(zip(("gen_top1_accuracy", "gen_top5_accuracy"), topk_accuracy(generation_logits.float().cpu(), labels, (1, 5))))
        if cls_logits is not None:
            acc.update(zip(("cls_top1_accuracy", "cls_top5_accuracy"), topk_accuracy(cls_logits.float().cpu(), labels, (1, 5))))
        for k, v in acc.items():
            sums[k] = sums.get(k, 0.0) + v * n
        weight += n
    keys = sorted(sums)
    t = torch.tensor([sums[k] for k in keys] + [weight], dtype=torch.float64, device=device if device.type == "cuda" else "cpu")
    if dist.get_world_size() > 1:
        dist.all_reduce(t)
    stats = {k: (t[i] / t[-1]).item() for i, k in enumerate(keys)}
    print("* Generation Top-1 Accuracy {:.3f}  Top-5 Accuracy {:.3f}".format(stats["gen_top1_accuracy"], stats["gen_top5_accuracy"]))
    return stats


def synthetic_loaders(args, config, seed):
    """(clip, title, label) batches; `.idx2label` names the classes (:94, :210)"""
    bs, frames, res, C = config["batch_size"], config["num_frames"], config["image_res"], config["num_classes"]
    names = ft.synthetic_titles(C, 4, torch.Generator().manual_seed(7))

    def split(n, s, batch, train):
        g = torch.Generator().manual_seed(s)
        titles = ft.synthetic_titles(n, max(4, config["max_length"] // 4), g)
        labels = torch.randint(0, C, (n,), generator=g).tolist()
        sp = ft.SyntheticSplit(n, batch, frames, res, s, lambda i: (titles[i], labels[i]), drop_last=train)
        sp.idx2label = dict(enumerate(names))
        return sp
    eval_bs = max(1, int(bs * 0.1))
    return split(args.synthetic_steps * args.update_freq * bs, seed, bs, True), split(2 * eval_bs + 1, 1001, eval_bs, False), split(eval_bs + 1, 1002, eval_bs, False)


def real_loaders(args, config):
    from dataset import create_dataset, create_loader, create_sampler
    from models.modeling_distributed_gpt3 import DistributedGPT3Tokenizer
    datasets = create_dataset("video_cls", config)
    samplers = create_sampler(datasets, [True, False, False], dist.get_world_size(), dist.get_rank())
    loaders = create_loader(datasets, samplers, batch_size=[args.batch_size] + [int(args.batch_size * 0.1)] * 2, num_workers=[args.num_workers] * 3,
                            is_trains=[True, False, False], collate_fns=[None, None, None])
    return loaders, DistributedGPT3Tokenizer(config["text_decoder"])


def main(args, config):
    s = ft.setup(args, config, lambda: real_loaders(args, config))
    if s.loaders is None:
        s.loaders = synthetic_loaders(args, config, s.seed)
    ft.build_engine(args, config, DistributedGPT3_Cls, s)
    data_loader, val_loader, test_loader = s.loaders

    def step_fn(batch):
        loss_generation, loss_cls = s.model(*make_training_batch(*batch, data_loader.dataset.idx2label, s.tokenizer, s.device, args.max_length))
        return {"loss_generation": loss_generation, "loss_cls": loss_cls}

    def evaluate_both():
        val = evaluation(s.model, val_loader, s.tokenizer, s.device, config, args.max_length)
        print("Validation Performance:", val)
        test = evaluation(s.model, test_loader, s.tokenizer, s.device, config, args.max_length)
        print("Test Performance:", test)
        return {**{f"val_{k}": v for k, v in val.items()}, **{f"test_{k}": v for k, v in test.items()}}

    if args.evaluate_only:                                           # :387-404: one log line with epoch -1
        log_stats = {**evaluate_both(), "epoch": -1, "n_parameters": s.n_parameters}
        ft.write_log(args, log_stats)
        return log_stats
    return ft.epoch_loop(args, s, step_fn, lambda epoch: evaluate_both())


def get_args(argv=None):
    return ft.get_args(argv)


if __name__ == "__main__":
    main(*get_args())

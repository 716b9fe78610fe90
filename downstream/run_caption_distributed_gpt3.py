"""Video captioning fine-tuning / evaluation on the MI355X-native path -- drop-in for the reference's
downstream/run_caption_distributed_gpt3.py (same command line, YAML / JSON configs, per-step protocol, result files), with
DistributedGPT3_Caption (caption loss; beam search over the KV-cache decode path) and the engine coming from youku_mplug_amd
(SURVEY.md section 8(f) ranks 1 and 3).

What this entry point adds to downstream/finetune_common.py:
  * the training batch (:113-115): [config["prompt"], caption] pairs padded to max_length, prompt masked out of the loss;
    `model(video, text)` -> loss_generation;
  * evaluation (:208-238): the prompt alone (padded to 20 tokens) + the clip's query features -> `model.generate` (one beam search
    per clip), decoded, blanks removed, the prompt cut off -> [{video_id, pred_caption, gold_caption}];
  * save_result (dataset/utils.py:114-160): one JSON per rank, merged by rank 0; cal_metric (:240-298): predictions and references
    reduced to their CJK characters, one token per character, scored by pycocoevalcap's COCOEvalCap when that package is importable.
    It is not on this image: then BLEU-1..4 (corpus level, closest-reference brevity penalty) and ROUGE-L (beta 1.2, best reference)
    are computed here as that package defines them, and METEOR / CIDEr are left out -- said in the log line, not guessed;
  * main (:418-440, 446-470): `--evaluate_only` scores the validation split and writes one log line; training epochs do not
    evaluate (the reference has that block commented out)."""
import json
import math
import os
import re
import sys
from collections import Counter

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import finetune_common as ft  # noqa: E402
from youku_mplug_amd.downstream import DistributedGPT3_Caption  # noqa: E402


def make_training_batch(video, text, prompt, tokenizer, device, max_length):
    input_text = [[prompt, x] for x in text]
    text_input = tokenizer(input_text, padding="max_length", truncation=True, max_length=max_length, return_tensors="pt").to(device)
    return video.to(device, non_blocking=True).bfloat16(), text_input


@torch.no_grad()
def evaluation(model, data_loader, tokenizer, device, config):
    """-> [{video_id, pred_caption, gold_caption}] (:208-238)"""
    module = getattr(model, "module", model)
    module.eval()
    result = []
    for n, (video, video_ids, gold_caption) in enumerate(data_loader):
        video = video.to(device, non_blocking=True).bfloat16()
        text = [config["prompt"] for _ in range(video.shape[0])]
        text_input = tokenizer(text, padding="max_length", truncation=True, max_length=20, return_tensors="pt").to(device)
        res = module.generate(video, text_input)
        for video_id, r, gold in zip(video_ids, res, gold_caption):
            ans = tokenizer.decode(r.tolist()[0]).replace(" ", "").strip()
            if config["prompt"] != "":
                ans = ans.split(config["prompt"])[-1].strip()
            result.append({"video_id": video_id, "pred_caption": ans, "gold_caption": list(gold)})
        if n == 0:
            print(result)
    return result


def save_result(result, result_dir, filename):
    """per-rank JSON, then rank 0 concatenates them in rank order (dataset/utils.py:114-160) -> path of the merged file"""
    rank, world = dist.get_rank(), dist.get_world_size()
    json.dump(result, open(os.path.join(result_dir, "%s_rank%d.json" % (filename, rank)), "w"))
    final = os.path.join(result_dir, "%s.json" % filename)
    dist.barrier()
    if rank == 0:
        merged = []
        for r in range(world):
            merged += json.load(open(os.path.join(result_dir, "%s_rank%d.json" % (filename, r)), "r"))
        json.dump(merged, open(final, "w"))
        print("result file saved to %s" % final)
    dist.barrier()
    return final


def normalize(text):
    """keep the CJK characters, one token each (:240-244)"""
    return " ".join(re.sub("[^\u4e00-\u9fa5]+", "", text))


def _ngrams(tokens, n):
    return Counter(tuple(tokens[i:i + n]) for i in range(len(tokens) - n + 1))


def bleu_scores(pairs, max_n=4):
    """Corpus BLEU-1..max_n as pycocoevalcap's Bleu scorer computes it (option 'closest'): clipped n-gram counts against the
    references' maximum counts summed over the corpus, brevity penalty from the summed closest reference lengths; the 1e-9 / 1e-15
    guards keep an empty numerator from producing log(0), as there."""
    tiny, small = 1e-15, 1e-9
    guess, correct = [0] * max_n, [0] * max_n
    test_len = ref_len = 0
    for hyp, refs in pairs:
        test_len += len(hyp)
        ref_len += min((abs(len(r) - len(hyp)), len(r)) for r in refs)[1]
        for n in range(1, max_n + 1):
            h = _ngrams(hyp, n)
            best = Counter()
            for r in refs:
                for g, c in _ngrams(r, n).items():
                    best[g] = max(best[g], c)
            guess[n - 1] += max(0, len(hyp) - n + 1)
            correct[n - 1] += sum(min(c, best[g]) for g, c in h.items())
    bp = 1.0 if test_len >= ref_len else math.exp(1 - ref_len / (test_len + small))
    out, logsum = [], 0.0
    for n in range(max_n):
        logsum += math.log((correct[n] + tiny) / (guess[n] + small))
        out.append(bp * math.exp(logsum / (n + 1)))
    return out


def rouge_l(pairs, beta=1.2):
    """mean over clips of the F-measure of the longest common subsequence, precision and recall each maximised over the references
    (pycocoevalcap's Rouge)"""
    def lcs(a, b):
        row = [0] * (len(b) + 1)
        for x in a:
            prev = 0
            for j, y in enumerate(b, 1):
                prev, row[j] = row[j], (prev + 1 if x == y else max(row[j], row[j - 1]))
        return row[-1]
    scores = []
    for hyp, refs in pairs:
        p = max((lcs(hyp, r) / len(hyp) if hyp else 0.0) for r in refs)
        rc = max((lcs(hyp, r) / len(r) if r else 0.0) for r in refs)
        scores.append((1 + beta ** 2) * p * rc / (rc + beta ** 2 * p) if p > 0 and rc > 0 else 0.0)
    return sum(scores) / max(len(scores), 1)


def cal_metric(result_file):
    """(:246-298) first prediction per video id; predictions and references normalised to CJK characters"""
    result_list = json.load(open(result_file, "r"))
    seen, preds, golds = set(), {}, {}
    for each in result_list:
        if each["video_id"] in seen:
            continue
        seen.add(each["video_id"])
        preds[each["video_id"]] = normalize(each["pred_caption"])
        golds[each["video_id"]] = [normalize(c) for c in each["gold_caption"]]
    try:
        from pycocoevalcap.eval import COCOEvalCap
        from pycocotools.coco import COCO
    except ImportError:
        pairs = [(preds[k].split(), [g.split() for g in golds[k]]) for k in preds]
        b = bleu_scores(pairs)
        return {"Bleu_1": b[0], "Bleu_2": b[1], "Bleu_3": b[2], "Bleu_4": b[3], "ROUGE_L": rouge_l(pairs),
                "scorer": "built-in BLEU / ROUGE-L (pycocoevalcap not installed: no METEOR / CIDEr)"}
    stem = os.path.basename(result_file).replace(".json", "")
    pred_file, gt_file = "/tmp/%s_coco_format.json" % stem, "/tmp/%s_gt_file.json" % stem
    json.dump([{"image_id": k, "caption": v} for k, v in preds.items()], open(pred_file, "w"), ensure_ascii=False)
    gt = {"annotations": [], "images": [{"id": k, "file_name": k} for k in golds], "type": None, "info": None, "licenses": None}
    for k, caps in golds.items():
        for cap in caps:
            gt["annotations"].append({"image_id": k, "caption": cap, "id": len(gt["annotations"])})
    json.dump(gt, open(gt_file, "w"), ensure_ascii=False)
    coco = COCO(gt_file)
    coco_res = coco.loadRes(pred_file)
    ev = COCOEvalCap(coco, coco_res)
    ev.params["image_id"] = coco_res.getImgIds()
    ev.evaluate()
    return ev.eval


def synthetic_loaders(args, config, seed):
    """(clip, caption) training batches; (clip, video ids, [gold captions]) evaluation batches (test_collect_fn, :46-52)"""
    bs, frames, res = config["batch_size"], config["num_frames"], config["image_res"]

    def split(n, s, train):
        caps = ft.synthetic_titles(2 * n, max(4, config["max_length"] // 4), torch.Generator().manual_seed(s))
        if train:
            return ft.SyntheticSplit(n, bs, frames, res, s, lambda i: (caps[i],), drop_last=True)
        return ft.SyntheticSplit(n, bs, frames, res, s, lambda i: (f"video{s}_{i}", [caps[i], caps[n + i]]), collate=list)
    return split(args.synthetic_steps * args.update_freq * bs, seed, True), split(bs + 1, 1001, False), split(bs // 2 + 1, 1002, False)


def real_loaders(args, config):
    from dataset import create_dataset, create_loader, create_sampler
    from models.modeling_distributed_gpt3 import DistributedGPT3Tokenizer

    def test_collect_fn(batch):
        videos, ids, golds = zip(*batch)
        return torch.stack(videos, dim=0), list(ids), list(golds)
    datasets = create_dataset("video_caption", config)
    samplers = create_sampler(datasets, [True, False, False], dist.get_world_size(), dist.get_rank())
    loaders = create_loader(datasets, samplers, batch_size=[args.batch_size] * 3, num_workers=[args.num_workers] * 3,
                            is_trains=[True, False, False], collate_fns=[None, test_collect_fn, test_collect_fn])
    return loaders, DistributedGPT3Tokenizer(config["text_decoder"])


def main(args, config):
    s = ft.setup(args, config, lambda: real_loaders(args, config))
    if s.loaders is None:
        s.loaders = synthetic_loaders(args, config, s.seed)
    ft.build_engine(args, config, DistributedGPT3_Caption, s)
    data_loader, val_loader, _test_loader = s.loaders

    def step_fn(batch):
        video, text = batch
        return {"loss_generation": s.model(*make_training_batch(video, text, config["prompt"], s.tokenizer, s.device, args.max_length))}

    if args.evaluate_only:
        result = evaluation(s.model, val_loader, s.tokenizer, s.device, config)
        val_stats = cal_metric(save_result(result, args.result_dir, "val_caption_result"))
        print("* Validation Stats:", val_stats)
        log_stats = {**{f"val_{k}": v for k, v in val_stats.items()}, "n_parameters": s.n_parameters}
        ft.write_log(args, log_stats)
        return log_stats
    return ft.epoch_loop(args, s, step_fn, lambda epoch: {})


def get_args(argv=None):
    args, config = ft.get_args(argv, extra=lambda p: p.add_argument("--no_zero_shot", action="store_true"))
    config.setdefault("prompt", "")
    args.result_dir = os.path.join(args.output_dir, "result")
    os.makedirs(args.result_dir, exist_ok=True)
    return args, config


if __name__ == "__main__":
    main(*get_args())

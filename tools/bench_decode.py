"""Decode-regime measurement: beam search (beam 5) over the KV-cache path at GPT-3 1.3B dims, visual prefix of 256
query tokens + an 8-token prompt, 48 generated tokens.  HBM-bound: every step streams the decoder weights once
(2.42 GB of layer weights + 0.21 GB tied LM head in bf16)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import youku_mplug_amd
from youku_mplug_amd.gpt3 import DistributedGPT3, GPT3Config
from youku_mplug_amd import generation


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cfg = GPT3Config(vocab_size=51200, hidden_size=2048, ffn_hidden_size=8192, num_hidden_layers=24, num_attention_heads=32,
                     max_position_embeddings=2048, layernorm_epsilon=1e-5, tokens_to_generate=48, eod_id=7)
    gpt = DistributedGPT3(config=cfg, device=dev).eval()
    Q, P, beam = 256, 8, 5
    qe = (torch.randn(1, Q, 2048, device=dev) * 0.5).to(torch.bfloat16)
    tokens = torch.randint(8, 51200, (1, P), device=dev)
    steps = []
    orig = generation.DecodeState.step

    def timed(self, *a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = orig(self, *a, **k)
        torch.cuda.synchronize()
        steps.append(time.perf_counter() - t0)
        return r
    generation.DecodeState.step = timed
    for rep in range(2):
        steps.clear()
        t0 = time.perf_counter()
        out = gpt.generate(tokens, do_sample=False, query_embeds=qe, beam_size=beam, termination_id=-1, prompt_length=P)
        torch.cuda.synchronize()
        tot = time.perf_counter() - t0
    nparam = sum(p.numel() for p in gpt.parameters())
    wbytes = 2.0 * (nparam - cfg.max_position_embeddings * cfg.hidden_size)
    dec = steps[1:]
    print(f"prefill ({Q}+{P} positions x {beam} beams): {steps[0]*1e3:.2f} ms")
    print(f"decode: {len(dec)} steps, {sum(dec)/len(dec)*1e3:.3f} ms/step (kernel path only), end-to-end {tot/len(steps)*1e3:.3f} ms/step incl. host search")
    print(f"weights streamed per step {wbytes/1e9:.2f} GB -> {wbytes/(sum(dec)/len(dec))/1e12:.2f} TB/s of 8 TB/s HBM peak; {beam/(tot/len(steps)):.0f} beam-tokens/s")
    import json
    print(json.dumps({"metric": "decode step (kernel path), GPT3-1.3B, beam 5, 256-query prefix", "ms_per_step": round(sum(dec) / len(dec) * 1e3, 3),
                      "end_to_end_ms_per_step": round(tot / len(steps) * 1e3, 3),
                      "roofline": {"bound": "hbm", "achieved": round(wbytes / (sum(dec) / len(dec)) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                   "frac": round(wbytes / (sum(dec) / len(dec)) / 8e12, 4), "algorithmic_bytes_per_step": int(wbytes)}}))


if __name__ == "__main__":
    main()

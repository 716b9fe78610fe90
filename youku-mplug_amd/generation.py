"""KV-cache generation on the gfx950 kernels -- the inference half of models/modeling_distributed_gpt3.py
(InferenceParams :1444-1473, sample :1398-1441 / :1620-1735, beam_search :1737-1873, BeamHypotheses :1908-1960)
and DistributedGPT3_Caption.generate (models/distributed_gpt3.py:790-809).

Layout: one cache per layer, cache[l] = [beams, max_len, 3H] bf16 in the decoder's own head-interleaved q|k|v row
format -- the qkv GEMM of every new position writes its row straight into the cache through the GEMM's C row map, and
the fused attention reads K/V from it with (batch, head, row) strides; nothing is transposed or copied (the reference
keeps [max_s, b, np, hn] K and V tensors and copies into them, :905-915).  A step processes the n new positions of
every sequence (prefill: the visual prefix + prompt; afterwards one token), attends causally over the cached rows and
runs the LM head on the last position only.  Beam re-ordering (swap_key_value_dict) is one strided row gather per
layer into the twin buffer.  The search logic itself stays on the host, as in the reference.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Callable, List, Optional

import torch

from . import ops


class BeamHypotheses:
    """models/modeling_distributed_gpt3.py:1908-1960 (length_penalty 1.0, early_stopping False as constructed at :1763)."""

    def __init__(self, num_beams: int, length_penalty: float = 1.0, early_stopping: bool = False):
        self.length_penalty, self.early_stopping, self.num_beams = length_penalty, early_stopping, num_beams
        self.beams: List[tuple] = []
        self.worst_score = 1e9

    def __len__(self):
        return len(self.beams)

    def add(self, hyp: torch.Tensor, sum_logprobs: float, beam_indices=None):
        score = sum_logprobs / (hyp.shape[-1] ** self.length_penalty)                      # :1936
        if len(self) < self.num_beams or score > self.worst_score:
            self.beams.append((score, hyp, beam_indices))
            if len(self) > self.num_beams:
                order = sorted([(s, i) for i, (s, _, _) in enumerate(self.beams)])
                del self.beams[order[0][1]]
                self.worst_score = order[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs: float, cur_len: int) -> bool:
        if len(self) < self.num_beams:
            return False
        if self.early_stopping:
            return True
        return self.worst_score >= best_sum_logprobs / cur_len ** self.length_penalty        # :1958-1960


class DecodeState:
    """Per-generation KV caches of a DistributedGPT3 (the InferenceParams of :1444-1473).  A step is ONE call into the
    C ABI (mpv_gpt_decode_step issues every launch of the incremental forward; the step is launch-bound otherwise)."""

    def __init__(self, gpt, batch: int, max_len: int):
        import ctypes as C
        from . import _lib
        cfg = gpt.config
        self.gpt, self.batch, self.max_len = gpt, batch, max_len
        self.H, self.np_, self.hn, self.V = cfg.hidden_size, cfg.num_attention_heads, cfg.kv_channels, cfg.vocab_size
        assert self.V % 8 == 0, "vocab_size must be a multiple of 8"
        lm = gpt.dist_model.language_model
        dev = lm.embedding.word_embeddings.weight.device
        self.nl = nl = len(lm.encoder.layers)
        self.cache = torch.zeros((nl, batch * max_len, 3 * self.H), dtype=torch.bfloat16, device=dev)
        self.twin = None                      # allocated on the first beam re-order
        self.pos = 0                          # rows cached so far (per sequence)
        # weight table (device pointers; the parameters outlive the state)
        self._layers = (_lib.GptLayerWeights * nl)()
        for i, layer in enumerate(lm.encoder.layers):
            att, mlp, w = layer.self_attention, layer.mlp, self._layers[i]
            w.ln1_w, w.ln1_b = layer.input_layernorm.weight.data_ptr(), layer.input_layernorm.bias.data_ptr()
            w.qkv_w, w.qkv_b = att.query_key_value.weight.data_ptr(), att.query_key_value.bias.data_ptr()
            w.dense_w, w.dense_b = att.dense.weight.data_ptr(), att.dense.bias.data_ptr()
            w.ln2_w, w.ln2_b = layer.post_attention_layernorm.weight.data_ptr(), layer.post_attention_layernorm.bias.data_ptr()
            w.fc1_w, w.fc1_b = mlp.dense_h_to_4h.weight.data_ptr(), mlp.dense_h_to_4h.bias.data_ptr()
            w.fc2_w, w.fc2_b = mlp.dense_4h_to_h.weight.data_ptr(), mlp.dense_4h_to_h.bias.data_ptr()
        fl = lm.encoder.final_layernorm
        self._w = _lib.GptWeights(nl, self.H, self.np_, lm.encoder.layers[0].mlp.dense_h_to_4h.out_features, self.V,
                                  float(fl.eps), self._layers, lm.embedding.word_embeddings.weight.data_ptr(),
                                  lm.embedding.position_embeddings.weight.data_ptr(), fl.weight.data_ptr(), fl.bias.data_ptr(),
                                  int(lm.embedding.position_embeddings.weight.shape[0]))
        self._ptrs = lambda t: (C.c_void_p * nl)(*[t[i].data_ptr() for i in range(nl)])
        self._cache_ptrs = self._ptrs(self.cache)
        self._twin_ptrs = None

    # ------------------------------------------------------------------ one incremental forward
    def step(self, tokens: Optional[torch.Tensor], query_embeds: Optional[torch.Tensor] = None) -> torch.Tensor:
        """tokens [B, n_tok] int64, query_embeds [B, Q, H] for the first call.  Appends Q + n_tok positions per
        sequence and returns the logits of the last one: [B, V] bf16."""
        import ctypes as C
        from . import _lib
        B, H = self.batch, self.H
        qf = None if query_embeds is None else query_embeds.reshape(-1, H).to(torch.bfloat16).contiguous()
        if query_embeds is not None and self.twin is not None:
            self._twin_shared = 0                    # a new visual prefix: the twin's copy of the shared positions is stale (reorder re-gathers from 0)
        Q = 0 if qf is None else qf.shape[0] // B
        L = 0 if tokens is None else tokens.shape[1]
        ids = tokens.contiguous() if L else None
        wsn = _lib.lib().mpv_gpt_decode_workspace_size(C.byref(self._w), B, Q + L)
        ws = ops.workspace(wsn, self.cache.device)
        logits = torch.empty((B, self.V), dtype=torch.bfloat16, device=self.cache.device)
        ops.check(_lib.lib().mpv_gpt_decode_step(C.byref(self._w), self._cache_ptrs, B, self.max_len, self.pos, ops._p(qf), Q, ops._p(ids), L,
                                                 ws.data_ptr(), ws.numel(), logits.data_ptr(), ops._stream()), "mpv_gpt_decode_step")
        self.pos += Q + L
        return logits

    def reorder(self, batch_idx: torch.Tensor, shared_prefix: int = 0):
        """swap_key_value_dict(:1459-1473): sequence j continues from the cache of sequence batch_idx[j].  The layers
        are one allocation, so the whole re-order is a single strided row gather over (layer, sequence) rows.
        `shared_prefix`: the caller's promise that the first `shared_prefix` cached positions are the same in every
        sequence (beam search: the visual prefix and the prompt, 264 of ~290 positions at the caption shapes).  Once both
        copies of the cache hold them they are not moved again: the gather starts at that position (round 4: 122 -> 9 us
        per step)."""
        first = self.twin is None
        if first:
            self.twin = torch.empty_like(self.cache)
            self._twin_ptrs = self._ptrs(self.twin)
            self._layer_base = torch.arange(self.nl, device=self.cache.device, dtype=torch.int64)[:, None] * self.batch
            self._twin_shared = 0                    # leading positions known to be identical in cache AND twin
        B, nl = self.batch, self.nl
        shared_prefix = min(int(shared_prefix), self.pos)
        lo = shared_prefix if shared_prefix <= self._twin_shared else 0
        idx = (self._layer_base + batch_idx.to(torch.int64)[None]).reshape(-1)
        pitch = self.max_len * 3 * self.H
        if self.pos > lo:
            off = lo * 3 * self.H
            ops.gather_rows_ld(self.cache.view(-1)[off:], idx, self.twin.view(-1)[off:], nl * B, (self.pos - lo) * 3 * self.H, pitch, pitch)
        self._twin_shared = shared_prefix            # after a gather from 0 the twin holds them too; the source did already
        self.cache, self.twin = self.twin, self.cache
        self._cache_ptrs, self._twin_ptrs = self._twin_ptrs, self._cache_ptrs


# ---------------------------------------------------------------------------------------------- sampling helpers
def _filter_top_k(logits, top_k):                                                            # :1369-1373
    kth = torch.topk(logits, top_k)[0][..., -1, None]
    logits.masked_fill_(logits < kth, float("-inf"))


def _filter_top_p(logits, top_p):                                                            # :1376-1395
    sorted_logits, sorted_indices = torch.sort(logits, descending=True)
    cumulative = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
    filt = cumulative > top_p
    filt[:, 1:] = filt[:, :-1].clone()
    filt[..., 0] = 0
    logits.masked_fill_(filt.scatter(1, sorted_indices, filt), float("-inf"))


def sample_token(logits: torch.Tensor, top_k=0, top_p=0.0, temperature=1.0, vocab_size=None, generator=None) -> torch.Tensor:
    """models/modeling_distributed_gpt3.py:1398-1441.  Greedy (top_k == 1) runs on mpv_logprob_topk; the stochastic
    filters are a handful of torch calls on the [B, V] logits, as in the reference."""
    assert logits.ndim == 2
    if top_k == 1:
        assert top_p == 0.0, "cannot set both greedy and top-p samplings."
        samples = ops.logprob_topk(logits, 1)[1].view(-1)
    else:
        lg = logits.float().clone()
        if temperature != 1.0:
            lg.div_(temperature)
        if top_k > 1:
            assert top_p == 0.0, "cannot set both top-k and top-p samplings."
            _filter_top_k(lg, top_k)
        elif top_p > 0.0:
            assert top_p <= 1.0
            _filter_top_p(lg, top_p)
        samples = torch.multinomial(lg.softmax(dim=-1), num_samples=1, generator=generator).view(-1)
    if vocab_size:
        samples = torch.clamp(samples, min=0, max=vocab_size - 1)
    return samples


# ---------------------------------------------------------------------------------------------- search loops
def _gen_config(gpt):
    ex = gpt.config.extra
    return SimpleNamespace(tokens_to_generate=ex.get("tokens_to_generate", 100), eod_id=ex.get("eod_id", 7), top_k=ex.get("top_k", 0),
                           top_p=ex.get("top_p", 0.9), max_position_embeddings=gpt.config.max_position_embeddings,
                           vocab_size=gpt.config.vocab_size)


StepFn = Callable[[Optional[torch.Tensor], Optional[torch.Tensor]], torch.Tensor]


def beam_search_loop(cfg, make_state: Callable[[int, int], object], tokens: torch.Tensor, query_embeds=None, beam_size=5,
                     num_return_gen=1, stop_token=None, prompt_length=None, topk_fn=None):
    """models/modeling_distributed_gpt3.py:1737-1873.  `make_state(batch, max_len)` returns an object with
    .step(tokens, query_embeds) -> logits [beams, V] and .reorder(batch_idx, shared_prefix=n); `topk_fn(logits, k, add)` ->
    (values, indices) of log_softmax + add.  Both are injectable so the host logic can be driven from any logits source."""
    assert tokens.size(0) == 1
    dev = tokens.device
    prompt_length = int(prompt_length) if prompt_length is not None else tokens.size(1)
    stop_token = cfg.eod_id if stop_token is None else int(stop_token)
    tokens = torch.cat((tokens, torch.full((1, cfg.tokens_to_generate), stop_token, dtype=torch.long, device=dev)), dim=-1)
    final_len = min(tokens.size(1), cfg.max_position_embeddings)
    if prompt_length >= final_len:
        raise ValueError("context length + tokens_to_generate too large")
    Q = 0 if query_embeds is None else query_embeds.size(1)
    state = make_state(beam_size, final_len + Q)
    hyp = BeamHypotheses(beam_size)
    done = False
    # The search state lives on the host (round 4): the token table and the beam scores are python-side bookkeeping in the
    # reference too (:1799-1840 walks the candidates one .item() at a time); per step the device sees ONE [2, beams] int64
    # upload (source beam, new token), the scores it adds to the log-probabilities, and two small downloads of the candidates.
    scores_h = [0.0] * beam_size
    scores = torch.zeros(beam_size, dtype=torch.float32, device=dev)
    tokens_h = tokens.cpu().repeat(beam_size, 1)
    if query_embeds is not None:
        query_embeds = query_embeds.repeat(beam_size, 1, 1)
    prev, total_prompt, total_final = 0, prompt_length + Q, final_len + Q
    context_length = total_prompt
    t2u = tokens_h[:, :total_prompt - Q].to(dev)
    for context_length in range(total_prompt, total_final):
        logits = state.step(t2u, query_embeds if context_length == total_prompt else None)
        V = logits.size(-1)
        k2 = 2 * beam_size
        vals, idxs = topk_fn(logits, k2, scores)                                             # log_softmax + scores (:1790-1791)
        vals, idxs = vals.tolist(), idxs.tolist()
        if context_length == total_prompt:                                                   # all beams identical: row 0 only (:1793-1795)
            cand = [(vals[0][j], 0, idxs[0][j]) for j in range(k2)]
        else:
            cand = [(vals[b][j], b, idxs[b][j]) for b in range(beam_size) for j in range(k2)]
            cand.sort(key=lambda c: (-c[0], c[1] * V + c[2]))
            cand = cand[:k2]
        next_beams = []
        for rank, (beam_score, beam_id, token_id) in enumerate(cand):
            if token_id == stop_token:
                if rank >= beam_size:                                                        # :1809-1812
                    continue
                hyp.add(tokens_h[beam_id].clone(), beam_score, context_length + 1 - total_prompt)
            else:
                next_beams.append((token_id, beam_score, beam_id))
            if len(next_beams) == beam_size:
                break
        if hyp.is_done(max(c[0] for c in cand), context_length + 1 - total_prompt):
            done = True
            break
        best_h = [b[2] for b in next_beams]
        tokens_h = tokens_h[best_h, :]
        tokens_h[:, context_length - Q] = torch.tensor([b[0] for b in next_beams], dtype=torch.long)
        scores_h = [b[1] for b in next_beams]
        moved = torch.tensor([best_h, [b[0] for b in next_beams]], dtype=torch.long).to(dev)
        scores = torch.tensor(scores_h, dtype=torch.float32).to(dev)
        state.reorder(moved[0], shared_prefix=total_prompt)
        t2u = moved[1].view(-1, 1)
        prev = context_length
    if not done:
        for b in range(beam_size):
            hyp.add(tokens_h[b].clone(), scores_h[b], context_length + 1 - total_prompt)
    ranked = sorted(hyp.beams, key=lambda x: x[0], reverse=True)
    num_return_gen = min(num_return_gen, len(ranked))
    return SimpleNamespace(sequences=torch.stack([ranked[i][1] for i in range(num_return_gen)], dim=0).to(dev),
                           scores=torch.tensor([ranked[i][0] for i in range(num_return_gen)], dtype=torch.float32))


def sample_loop(cfg, make_state, tokens: torch.Tensor, query_embeds=None, temperature=1.0, use_eod_token_for_early_termination=True,
                stop_on_double_eol=False, stop_on_eol=False, termination_id=None, prompt_length=None, generator=None):
    """models/modeling_distributed_gpt3.py:1620-1735."""
    B, dev = tokens.size(0), tokens.device
    lengths = prompt_length if prompt_length is not None else torch.tensor([tokens.size(1)], device=dev)
    lengths = torch.as_tensor(lengths, device=dev).view(-1)
    tokens = torch.cat((tokens, torch.full((B, cfg.tokens_to_generate), cfg.eod_id, dtype=torch.long, device=dev)), dim=-1)
    min_prompt = int(lengths.min().item())
    max_len = min(tokens.size(1), cfg.max_position_embeddings)
    if min_prompt >= max_len:
        raise ValueError("context length + tokens_to_generate too large")
    Q = 0 if query_embeds is None else query_embeds.size(1)
    state = make_state(B, max_len + Q)
    termination_id = cfg.eod_id if termination_id is None else int(termination_id)
    is_done = torch.zeros(B, dtype=torch.bool, device=dev)
    prev, total_min, total_max = 0, min_prompt + Q, max_len + Q
    context_length = total_min
    for context_length in range(total_min, total_max):
        t2u = tokens[:, max(prev - Q, 0):context_length - Q]
        logits = state.step(t2u, query_embeds if context_length == total_min else None)
        new = sample_token(logits, top_k=cfg.top_k, top_p=cfg.top_p, temperature=temperature, vocab_size=cfg.vocab_size,
                           generator=generator)
        started = lengths <= context_length - Q
        tokens[started, context_length - Q] = new[started]
        prev = context_length
        if stop_on_double_eol:
            done_token = ((new == 628) | ((new == 198) & (tokens[:, context_length - Q - 1] == 198))) & started
        elif stop_on_eol:
            done_token = ((new == 628) | (new == 198)) & started
        else:
            done_token = (new == termination_id) & started
        is_done |= done_token
        if use_eod_token_for_early_termination and bool(is_done.all()):
            break
    return tokens[:, :context_length + 1]      # (sic) the reference slices with the prefix-inclusive length (:1733)


def install(gpt_cls):
    """Adds sample / beam_search / generate to DistributedGPT3 (same signatures as the reference methods)."""

    def _make_state(self):
        return lambda batch, max_len: DecodeState(self, batch, max_len)

    def sample(self, tokens, query_embeds=None, temperature=1.0, **kw):
        return sample_loop(_gen_config(self), _make_state(self), tokens, query_embeds=query_embeds, temperature=temperature, **kw)

    def beam_search(self, tokens, query_embeds=None, beam_size=5, num_return_gen=1, stop_token=None, **kw):
        return beam_search_loop(_gen_config(self), _make_state(self), tokens, query_embeds=query_embeds, beam_size=beam_size,
                                num_return_gen=num_return_gen, stop_token=stop_token, prompt_length=kw.pop("prompt_length", None),
                                topk_fn=lambda lg, k, add: ops.logprob_topk(lg, k, add=add))

    @torch.no_grad()
    def generate(self, tokens, do_sample=True, termination_id=None, *args, **kw):                # :1875-1880
        was = self.training
        self.eval()
        try:
            if do_sample:
                return self.sample(tokens, termination_id=termination_id, *args, **kw)
            return self.beam_search(tokens, stop_token=termination_id, *args, **kw)
        finally:
            self.train(was)

    gpt_cls.sample, gpt_cls.beam_search, gpt_cls.generate = sample, beam_search, generate

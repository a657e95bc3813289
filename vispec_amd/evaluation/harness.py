"""Spec / AR answer generation on synthetic (image, prompt) requests, emitting the reference's JSONL record
(evaluation/gen_spec_answer_coco_caption.py:262-285: question_id, model_id, choices[{index, turns, idxs, new_tokens, wall_time,
acceptance_length}], tstamp) and the speed-up of speed.py:56-97 (mean tokens/s spec ÷ mean tokens/s AR, mean accept length)."""
from __future__ import annotations

import json
import time
from typing import Iterable, List

import torch


def get_model_answers(model, requests: Iterable, answer_file: str, model_id: str = "vispec-hip", temperature: float = 0.0,
                      max_new_tokens: int = 512, warmup: int = 3, baseline: bool = False):
    """requests: iterable of (question_id, input_ids [1,L], specgenerate kwargs).  3 warm-ups on the first request with
    manual_seed(0), then manual_seed(i) per choice — gen_spec_answer_coco_caption.py:160-232."""
    requests = list(requests)
    run = (lambda ids, kw: model.baseline_generate(ids, max_new_tokens=max_new_tokens, **kw)) if baseline else None
    for _ in range(warmup):
        torch.manual_seed(0)
        qid, ids, kw = requests[0]
        if baseline:
            run(ids, kw)
        else:
            model.specgenerate(ids, temperature=temperature, log=True, max_new_tokens=max_new_tokens, seed=0, **kw)
    with open(answer_file, "a") as fout:
        for i, (qid, ids, kw) in enumerate(requests):
            torch.manual_seed(i)  # :212 — the device-side sampler takes the same per-sample seed
            torch.cuda.synchronize()
            t0 = time.time()
            if baseline:
                out = run(ids, kw)
                new_token, idx, acc = out.shape[1] - ids.shape[1], out.shape[1] - ids.shape[1], []
            else:
                out, new_token, idx, acc = model.specgenerate(ids, temperature=temperature, log=True, return_acceptance_len=True,
                                                              max_new_tokens=max_new_tokens, seed=i, **kw)
            torch.cuda.synchronize()
            wall = time.time() - t0
            rec = {"question_id": qid, "model_id": model_id, "tstamp": time.time(),
                   "choices": [{"index": 0, "turns": [out[0, ids.shape[1]:].tolist()], "idxs": [int(idx)], "new_tokens": [int(new_token)],
                                "wall_time": [wall], "acceptance_length": [int(a) for a in acc]}]}
            fout.write(json.dumps(rec) + "\n")


def speed(spec_file: str, baseline_file: str):
    """speed.py:56-97: ratio of mean per-sample tokens/s, and the mean accept length."""
    def rates(path):
        r, acc = [], []
        for line in open(path):
            c = json.loads(line)["choices"][0]
            r.append(sum(c["new_tokens"]) / sum(c["wall_time"]))
            acc += c["acceptance_length"]
        return r, acc
    rs, acc = rates(spec_file)
    rb, _ = rates(baseline_file)
    return {"speedup": (sum(rs) / len(rs)) / (sum(rb) / len(rb)), "tau": sum(acc) / max(1, len(acc)),
            "spec_tokens_per_s": sum(rs) / len(rs), "ar_tokens_per_s": sum(rb) / len(rb)}


def requests_from_samples(task: str, samples, processor=None, model=None, device="cuda:0", id_key="question_id"):
    """Benchmark samples -> the (question_id, input_ids, specgenerate kwargs) triples get_model_answers takes, through the reference's
    prompt front-end for `task` (evaluation/prompts.py: the conversation, processor arguments and call of vispec/evaluation/<task>_prompt.py;
    gen_spec_answer_<task>.py:160-232 feed `model_inputs` to specgenerate the same way)."""
    from .prompts import build_prompt, make_processor
    if processor is None:
        processor = make_processor(model, task)
    for i, d in enumerate(samples):
        inputs = dict(build_prompt(task, d, processor=processor, device=device))
        ids = inputs.pop("input_ids")
        inputs.pop("attention_mask", None)  # (batch 1, no padding: the reference passes it on and its forward ignores it)
        yield d.get(id_key, i), ids, inputs


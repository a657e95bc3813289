"""The prompt front-ends (vispec_amd/evaluation/prompts.py) against fixture G18 — what the REFERENCE's own `build_prompt` functions
(vispec/evaluation/*_prompt.py) built when run with a recording processor (tests/golden/gen_golden.py g18): the chat-template conversation,
the arguments the processor was constructed with (Qwen pixel bounds or not), the arguments of the processor call, and the device the batch
is moved to — for all 13 single-turn benchmarks x {LLaVA, Qwen2.5-VL} and 10 ScienceQA few-shot formats x {with, without captions}."""
import json
import os
import sys
import types
from types import SimpleNamespace

import pytest

from helpers import ROOT  # noqa: F401
from vispec_amd.evaluation import prompts as P

G18 = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "g18_prompts.json")))
DATA = dict(image="<IMAGE>", text="What is on the table?", question="What colour is the car?", video_name="<VIDEO>", video="<VIDEO>")


class Rec:
    log = []

    def __init__(self, ctor):
        self.ctor, self.calls = ctor, []

    @classmethod
    def from_pretrained(cls, *a, **kw):
        r = cls(dict(args=list(a), kwargs=kw))
        cls.log.append(r)
        return r

    def apply_chat_template(self, conv, **kw):
        self.conv, self.tmpl_kw = conv, kw
        return "<PROMPT>"

    def __call__(self, **kw):
        self.calls.append(kw)
        return SimpleNamespace(to=lambda dev: dict(device=dev))


@pytest.fixture()
def recording(monkeypatch):
    import transformers
    real = transformers.AutoProcessor  # (resolves the lazy attribute — which REPLACES sys.modules["transformers"]: patch the live object)
    transformers = sys.modules["transformers"]
    transformers.AutoProcessor = Rec
    qv = types.ModuleType("qwen_vl_utils")
    qv.process_vision_info = lambda conv, return_video_kwargs=False: (None, ["<FRAMES>"], {"fps": [2.0]})
    monkeypatch.setitem(sys.modules, "qwen_vl_utils", qv)
    Rec.log.clear()
    try:
        yield Rec
    finally:
        transformers.AutoProcessor = real


def norm(x):
    return json.loads(json.dumps(x, sort_keys=True))


@pytest.mark.parametrize("key", [k for k in sorted(G18) if "|" in k and not k.startswith("scienceqa")])
def test_single_turn_benchmarks_build_the_references_prompt(recording, key):
    task, model = key.split("|")
    ret = P.build_prompt(task, dict(DATA), model=model)
    r, want = recording.log[-1], G18[key]
    assert norm(r.conv) == want["conversation"]
    assert norm(r.ctor) == want["ctor"], "processor construction (Qwen pixel bounds) differs from the reference's file for this benchmark"
    assert norm(r.tmpl_kw) == want["template_kwargs"] and norm(r.calls[-1]) == want["call"]
    assert norm(ret) == want["returned"]


@pytest.mark.parametrize("key", [k for k in sorted(G18) if k.startswith("scienceqa|")])
def test_scienceqa_few_shot_prompt(recording, key):
    _, fmt, cap = key.split("|")
    ret = P.build_prompt_scienceqa(G18["scienceqa_problems"], ["0", "3", "1"], "4", model="Qwen/Qwen2.5-VL-7B-Instruct", prompt_format=fmt,
                                   use_caption=bool(int(cap)), options=["A", "B", "C", "D", "E"])
    r, want = recording.log[-1], G18[key]
    assert norm(r.conv) == want["conversation"]
    assert norm(r.ctor) == want["ctor"] and norm(r.calls[-1]) == want["call"] and norm(ret) == want["returned"]


def test_unknown_benchmark_and_missing_processor_are_refused():
    with pytest.raises(KeyError, match="unknown benchmark"):
        P.build_prompt("imagenet", {}, processor=object())
    with pytest.raises(ValueError, match="processor or the checkpoint"):
        P.build_prompt("gqa", dict(DATA))
    with pytest.raises(ValueError, match="few-shot"):
        P.build_prompt("scienceqa", {}, processor=object())

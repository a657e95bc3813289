"""Prompt front-ends of the reference's evaluation harness (evaluation/*_prompt.py: one `build_prompt` per benchmark, each a chat-template
conversation handed to the HF processor) as ONE table-driven function.  Host-side only: it produces what `SpecModel.specgenerate(**inputs)`
takes (`input_ids`, `pixel_values`, `image_sizes` / `image_grid_thw` / `pixel_values_videos` ...), nothing here touches the hot path.

    inputs = build_prompt("coco_caption", {"image": pil_image}, model="llava-hf/llava-v1.6-vicuna-7b-hf")
    out = sm.specgenerate(**inputs, max_new_tokens=512)

What the reference does per benchmark (the conversations are pinned by fixture G18, captured from the reference's own functions):
  * system turn: the Vicuna system sentence (every file);
  * one user turn whose content list is, in order, the task's text parts and the image placeholder (video benchmarks: the video entry FIRST, with
    `max_pixels = 360 * 420`, `max_frames = 8`, and `qwen_vl_utils.process_vision_info` extracting the frames);
  * `processor.apply_chat_template(conversation, add_generation_prompt=True)` -> `processor(images=..., text=..., return_tensors="pt").to(device)`;
  * Qwen2.5-VL processors are built with `use_fast=True, min_pixels = 256 * 28 * 28, max_pixels = 1280 * 28 * 28` by the image benchmarks that
    look at the model name (gqa, mmbench, mme, mmvet, seed_bench, textvqa, vizwiz, vqav2, scienceqa — NOT coco_caption, synthdog, hr_bench, whose
    files build the default processor: coco_caption_prompt.py:5, synthdog_prompt.py:4, hr_bench_prompt.py:4-5).
ScienceQA (scienceqa_prompt.py) is few-shot: `shot_qids` solved examples as user / assistant turns, then the test question with the reference's
instruction sentence spliced in before "Answer:"."""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence

SYSTEM = ("A chat between a curious human and an artificial intelligence assistant. The assistant gives helpful, detailed, and polite answers to "
          "the human's questions.")
EXPLAIN = "Please answer with an explanation."
OCR_EXPLAIN = ("Perform an OCR task on the provided image. Please extract the text accurately and provide a detailed explanation of the process. "
               "Ensure the response is comprehensive and well-structured.")

# task -> (text parts of the user turn: "@key" = data[key], anything else literal; visual kind; data key of the visual; Qwen pixel bounds?)
TASKS: Dict[str, tuple] = {
    "coco_caption": (["Please provide a detailed description of the given image."], "image", "image", False),  # coco_caption_prompt.py:20-31
    "synthdog": (["Perform an OCR task on the provided image. Please extract the text accurately and ensure the response is comprehensive and "
                  "well-structured."], "image", "image", False),                                               # synthdog_prompt.py:19-30
    "gqa": (["@text", EXPLAIN], "image", "image", True),                                                        # gqa_prompt.py:26-41
    "mmbench": (["@text", EXPLAIN], "image", "image", True),
    "mme": (["@text", EXPLAIN], "image", "image", True),
    "seed_bench": (["@text", EXPLAIN], "image", "image", True),
    "vqav2": (["@text", EXPLAIN], "image", "image", True),
    "mmvet": (["@question", EXPLAIN], "image", "image", True),                                                  # mmvet_prompt.py:32
    "vizwiz": (["@question", EXPLAIN], "image", "image", True),
    "hr_bench": (["@question", EXPLAIN], "image", "image", False),                                              # hr_bench_prompt.py:4-5: `if False`
    "textvqa": (["@question", OCR_EXPLAIN], "image", "image", True),                                            # textvqa_prompt.py:32-37
    "msvd_qa": (["@question", EXPLAIN], "video", "video_name", False),                                          # msvd_qa_prompt.py:21-38
    "mvbench": (["@question", EXPLAIN], "video", "video", False),                                               # mvbench_prompt.py:21-38
}
QWEN_PIXELS = dict(use_fast=True, min_pixels=256 * 28 * 28, max_pixels=1280 * 28 * 28)


def make_processor(model: str, task: str):
    """The processor the reference's file for `task` builds from the checkpoint name (local directories work the same)."""
    from transformers import AutoProcessor
    bounds = TASKS[task][3] if task in TASKS else True  # (scienceqa: bounds)
    if bounds and "Qwen2.5-VL" in model:
        return AutoProcessor.from_pretrained(model, **QWEN_PIXELS)
    return AutoProcessor.from_pretrained(model)


def conversation(task: str, data: Dict[str, Any]) -> List[dict]:
    """The chat-template conversation of one benchmark sample."""
    parts, kind, vkey, _ = TASKS[task]
    content: List[dict] = []
    if kind == "video":
        content.append({"type": "video", "video": data[vkey], "max_pixels": 360 * 420, "max_frames": 8})
    content += [{"type": "text", "text": data[p[1:]] if p.startswith("@") else p} for p in parts]
    if kind == "image":
        content.append({"type": "image"})
    return [{"role": "system", "content": [{"type": "text", "text": SYSTEM}]}, {"role": "user", "content": content}]


def build_prompt(task: str, data: Dict[str, Any], processor=None, model: Optional[str] = None, device="cuda:0"):
    """-> the processor's batch on `device` (what the reference's `build_prompt(data, args)` returns for that benchmark)."""
    if task == "scienceqa":
        raise ValueError("scienceqa is few-shot: use build_prompt_scienceqa(problems, shot_qids, test_qid, ...)")
    if task not in TASKS:
        raise KeyError(f"unknown benchmark {task!r}: one of {sorted(TASKS) + ['scienceqa']}")
    if processor is None:
        if model is None:
            raise ValueError("build_prompt needs a processor or the checkpoint name / directory to build one from")
        processor = make_processor(model, task)
    conv = conversation(task, data)
    text = processor.apply_chat_template(conv, add_generation_prompt=True)
    if TASKS[task][1] == "video":
        from qwen_vl_utils import process_vision_info  # (what the reference imports: msvd_qa_prompt.py:2)
        image_inputs, video_inputs, video_kwargs = process_vision_info(conv, return_video_kwargs=True)
        return processor(text=text, images=image_inputs, videos=video_inputs, return_tensors="pt", **video_kwargs).to(device)
    return processor(images=[data[TASKS[task][2]]], text=text, return_tensors="pt").to(device)


# ------------------------------------------------------------------------------------------------ ScienceQA (scienceqa_prompt.py)
_IN_FIELDS = {"Q": "Question: {question}\n", "C": "Context: {context}\n", "M": "Options: {choice}\n"}


def sqa_example(fmt: str, problem: Dict[str, Any], options: Sequence[str], use_caption: bool = False, test_example: bool = True) -> str:
    """One formatted ScienceQA example (create_one_example, scienceqa_prompt.py:38-92).  fmt = "<input>-<output>", input a permutation of
    Q(uestion) C(ontext) M(options) with an optional L(ecture) / E(xplanation) group (one "BECAUSE:" line), output one of A AL AE ALE AEL LA EA
    LEA ELA.  The reference's AL uses the SOLUTION and AE the LECTURE (scienceqa_prompt.py:69-72): kept."""
    inp, outp = fmt.split("-")
    ctx = " ".join([problem["hint"], problem["caption"] if use_caption else ""]).strip() or "N/A"
    choice = " ".join("({}) {}".format(options[i], c) for i, c in enumerate(problem["choices"]))
    answer, lecture, solution = options[problem["answer"]], problem["lecture"], problem["solution"]
    text, i = "", 0
    while i < len(inp):
        ch = inp[i]
        if ch in _IN_FIELDS:
            text += _IN_FIELDS[ch].format(question=problem["question"], context=ctx, choice=choice)
            i += 1
        else:  # the L / E group
            j = i
            while j < len(inp) and inp[j] in "LE":
                j += 1
            text += "BECAUSE: " + " ".join(lecture if c == "L" else solution for c in inp[i:j]) + "\n"
            i = j
    if test_example:
        out = "Answer:"
    else:
        the = f"The answer is {answer}."
        out = {"A": f"Answer: {the}", "AL": f"Answer: {the} BECAUSE: {solution}", "AE": f"Answer: {the} BECAUSE: {lecture}",
               "ALE": f"Answer: {the} BECAUSE: {lecture} {solution}", "AEL": f"Answer: {the} BECAUSE: {solution} {lecture}",
               "LA": f"Answer: {lecture} {the}", "EA": f"Answer: {solution} {the}", "LEA": f"Answer: {lecture} {solution} {the}",
               "ELA": f"Answer: {solution} {lecture} {the}"}[outp]
    text = (text + out).replace("  ", " ").strip()
    if text.endswith("BECAUSE:"):
        text = text.replace("BECAUSE:", "").strip()
    return text


def conversation_scienceqa(problems: Dict[str, dict], shot_qids: Sequence[str], test_qid: str, prompt_format: str = "CQM-A",
                           options: Sequence[str] = ("A", "B", "C", "D", "E"), use_caption: bool = False):
    """-> (conversation, images): the solved shots as user / assistant turns, then the test question (scienceqa_prompt.py:116-199)."""
    conv = [{"role": "system", "content": [{"type": "text", "text": SYSTEM}]}]
    images = []
    for qid in shot_qids:
        q, a = sqa_example(prompt_format, problems[qid], options, use_caption, test_example=False).split("Answer:")[:2]
        conv.append({"role": "user", "content": [{"type": "text", "text": f"{q}Answer:"}, {"type": "image"}]})
        conv.append({"role": "assistant", "content": [{"type": "text", "text": a.strip()}]})
        images.append(problems[qid]["image"])
    test = sqa_example(prompt_format, problems[test_qid], options, use_caption, test_example=True)
    test = test.replace("Answer:", 'Your answer should begin with "The answer is". Please answer with an explanation. Answer:')
    conv.append({"role": "user", "content": [{"type": "text", "text": test}, {"type": "image"}]})
    images.append(problems[test_qid]["image"])
    return conv, images


def build_prompt_scienceqa(problems, shot_qids, test_qid, processor=None, model: Optional[str] = None, device="cuda:0", **fmt):
    if processor is None:
        if model is None:
            raise ValueError("build_prompt_scienceqa needs a processor or the checkpoint name / directory to build one from")
        processor = make_processor(model, "scienceqa")
    conv, images = conversation_scienceqa(problems, shot_qids, test_qid, **fmt)
    text = processor.apply_chat_template(conv, add_generation_prompt=True)
    return processor(images=images, text=text, return_tensors="pt").to(device)

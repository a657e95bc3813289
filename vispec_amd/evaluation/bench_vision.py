"""The vision front-end of a bench request inside the timed region (round 5: split out of the bench script).  The reference's wall clock
brackets the whole specgenerate call, vision tower included (evaluation/gen_spec_answer_coco_caption.py:221-232; spec_model_ours.py:339-356, 391-396);
bench.py installs an InLoopFrontEnd as `base_model.vision` of its models so that every timed request runs HF's own modules on its pixels."""
from types import SimpleNamespace

import torch


def build_front_end(model, tcfg, device, n_img):
    """The vision front-end a request of this model goes through in the reference (spec_model_ours.py:339-356, 391-396) — HF's own modules at the
    published architecture (LLaVA-1.6: CLIP ViT-L/14-336, 5 anyres tiles of a 640x427 image -> 2144 tokens (672x672 -> 2928), 2-layer projector,
    unpad + image_newline packing; Qwen2.5-VL: its 32-layer window-attention tower + patch merger), random-initialised in bf16 (no vision checkpoint
    exists on the box), on PyTorch-ROCm as the north star prescribes.  `model` = bench.py's --model name.
    -> (HFVisionFrontEnd, description, pixels(req_id) -> (pixel tensor, image_sizes))"""
    from vispec_amd.model.vision import HFVisionFrontEnd
    dt = torch.bfloat16
    if model.startswith("qwen"):
        from transformers import Qwen2_5_VLConfig
        from transformers.models.qwen2_5_vl.modeling_qwen2_5_vl import Qwen2_5_VisionTransformerPretrainedModel as Visual
        vc = Qwen2_5_VLConfig().vision_config
        # the published 7B checkpoint's vision_config (HF's class defaults are not it): 32 blocks of width 1280 / MLP 3420, merger to 3584
        vc.hidden_size, vc.intermediate_size, vc.num_heads, vc.depth, vc.out_hidden_size = 1280, 3420, 16, 32, tcfg.hidden_size
        grids = [(1, 32, 32)] * 4 if model == "qwen7b" else [(1, 68, 92)]
        fe = HFVisionFrontEnd("Qwen2_5_VLForConditionalGeneration", SimpleNamespace(vision_config=vc), Visual._from_config(vc).to(device, dt).eval(), None, None)
        n_patch = sum(t * h * w for t, h, w in grids)
        width = vc.in_channels * vc.temporal_patch_size * vc.patch_size ** 2

        def pixels(req_id):
            g = torch.Generator(device="cpu").manual_seed(5000 + int(req_id))
            return torch.randn(n_patch, width, generator=g).to(device, dt), None
        what = f"Qwen2.5-VL vision tower ({vc.depth} layers, hidden {vc.hidden_size}), grids {grids}"
    else:
        from transformers import AutoModel, LlavaNextConfig
        from transformers.models.llava_next.modeling_llava_next import LlavaNextMultiModalProjector
        c = LlavaNextConfig()
        c.text_config.hidden_size = tcfg.hidden_size
        size = {2144: (427, 640), 2928: (672, 672), 2340: (480, 640)}.get(n_img)
        if size is None:
            raise ValueError(f"no anyres image size known for {n_img} image tokens")
        fe = HFVisionFrontEnd("LlavaNextForConditionalGeneration", c, AutoModel.from_config(c.vision_config).to(device, dt).eval(),
                              LlavaNextMultiModalProjector(c).to(device, dt).eval(), torch.zeros(tcfg.hidden_size, device=device, dtype=dt))
        sizes = torch.tensor([list(size)])
        vc = c.vision_config

        def pixels(req_id):
            g = torch.Generator(device="cpu").manual_seed(5000 + int(req_id))
            return torch.randn(1, 5, 3, vc.image_size, vc.image_size, generator=g).to(device, dt), sizes
        what = f"CLIP ViT-L/{vc.patch_size}-{vc.image_size} ({vc.num_hidden_layers} layers) on 5 anyres tiles of a {size[1]}x{size[0]} image + projector + unpad/newline packing"
    return fe, what + " (random-initialised HF modules, bf16, PyTorch-ROCm)", pixels


class VisionInput:
    """What a bench request carries as `pixel_values` when the front end runs inside the timed region: the image's pixels (input of the tower)
    and the request's SURVEY §8(d) synthetic features (what the target and the draft see, whatever tower weights are on the box)."""
    __slots__ = ("pixels", "features", "image_sizes")

    def __init__(self, pixels, features, image_sizes):
        self.pixels, self.features, self.image_sizes = pixels, features, image_sizes


class InLoopFrontEnd:
    """`base_model.vision` of the bench's models (TargetLM.get_image_features routes every request through `.features`): the HF front-end's
    whole arithmetic runs on the request's pixels INSIDE specgenerate — the reference's wall clock brackets it, gen_spec_answer_coco_caption.py:
    221-232 — and the embeddings handed on are the request's synthetic features + 0 x the tower's output (a random-initialised tower's features
    would not be the workload SURVEY §8(d) defines; the dependency keeps its launches on the request's critical path)."""
    tower = True

    def __init__(self, fe, what, pixels):
        self.fe, self.what, self.pixels = fe, what, pixels

    @torch.no_grad()
    def _run(self, pv, image_grid_thw):
        out = self.fe.features(pv.pixels, image_sizes=pv.image_sizes, image_grid_thw=image_grid_thw)
        if tuple(out.shape) != tuple(pv.features.shape):
            raise ValueError(f"vision front-end produced {tuple(out.shape)}, the request's features are {tuple(pv.features.shape)}")
        return pv.features + out.mul(0).nan_to_num()

    def features(self, pv, image_sizes=None, image_grid_thw=None, **kw):
        return self._run(pv, image_grid_thw)

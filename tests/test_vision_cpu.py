"""Vision front-end for real checkpoints — CPU.  Tiny random HF checkpoints are written to a temp dir; the
front-end built from the files alone (vision tower + projector + anyres packing, no language model) must reproduce HF's own
`get_image_features` bit for bit, and `load_target_dir` must recover the language-model weights and config next to them."""
import os
import sys

import pytest

torch = pytest.importorskip("torch")
transformers = pytest.importorskip("transformers")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from vispec_amd.model.vision import HFVisionFrontEnd  # noqa: E402
from vispec_amd.weights_io import load_target_dir  # noqa: E402


def _pooled(out):
    out = getattr(out, "pooler_output", out)
    return out if torch.is_tensor(out) else torch.cat(list(out), 0)


def _clip_llama():
    from transformers import CLIPVisionConfig, LlamaConfig
    vc = CLIPVisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=3, num_attention_heads=2, image_size=28, patch_size=14)
    tc = LlamaConfig(vocab_size=128, hidden_size=48, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2)
    return vc, tc


def test_llava_next_front_end_matches_hf(tmp_path):
    from transformers import LlavaNextConfig, LlavaNextForConditionalGeneration
    vc, tc = _clip_llama()
    cfg = LlavaNextConfig(vision_config=vc, text_config=tc, image_grid_pinpoints=[[28, 56], [56, 28], [56, 56]], image_token_index=127)
    torch.manual_seed(0)
    m = LlavaNextForConditionalGeneration(cfg).eval()
    m.save_pretrained(tmp_path)
    fe = HFVisionFrontEnd.from_dir(str(tmp_path), "cpu", torch.float32)
    pv = torch.randn(2, 5, 3, 28, 28)
    sizes = torch.tensor([[50, 30], [28, 56]])
    with torch.no_grad():
        ref = _pooled(m.model.get_image_features(pv, sizes, vision_feature_layer=cfg.vision_feature_layer,
                                                 vision_feature_select_strategy=cfg.vision_feature_select_strategy))
    mine = fe.features(pv, image_sizes=sizes)
    assert mine.shape == ref.shape and mine.shape[1] == 48 and torch.equal(mine, ref)
    # the language model next to it: config + tensors under the names the engine expects
    tcfg, sd, _ = load_target_dir(str(tmp_path))
    assert (tcfg.hidden_size, tcfg.num_layers, tcfg.vocab_size, tcfg.image_token_index) == (48, 2, 128, 127)
    assert tcfg.architectures[0] == "LlavaNextForConditionalGeneration" and not tcfg.qkv_bias
    ref_sd = m.model.language_model.state_dict()
    assert torch.equal(sd["model.layers.1.mlp.down_proj.weight"], ref_sd["layers.1.mlp.down_proj.weight"])
    assert torch.equal(sd["lm_head.weight"], m.lm_head.weight) and not any("vision" in k or "newline" in k for k in sd)


def test_llava15_front_end_matches_hf(tmp_path):
    from transformers import LlavaConfig, LlavaForConditionalGeneration
    vc, tc = _clip_llama()
    cfg = LlavaConfig(vision_config=vc, text_config=tc, image_token_index=127)
    torch.manual_seed(1)
    m = LlavaForConditionalGeneration(cfg).eval()
    m.save_pretrained(tmp_path)
    fe = HFVisionFrontEnd.from_dir(str(tmp_path), "cpu", torch.float32)
    pv = torch.randn(2, 3, 28, 28)
    with torch.no_grad():
        ref = _pooled(m.model.get_image_features(pv, vision_feature_layer=cfg.vision_feature_layer,
                                                 vision_feature_select_strategy=cfg.vision_feature_select_strategy))
    mine = fe.features(pv)
    assert torch.equal(mine, ref.reshape(mine.shape)) and mine.shape == (8, 48)


def test_qwen25vl_front_end_and_config(tmp_path):
    from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration
    cfg = Qwen2_5_VLConfig(
        text_config=dict(vocab_size=160, hidden_size=64, intermediate_size=96, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=1,
                         rope_scaling={"type": "mrope", "mrope_section": [4, 6, 6]}, bos_token_id=1, eos_token_id=2),
        vision_config=dict(depth=2, hidden_size=32, intermediate_size=64, num_heads=2, out_hidden_size=64, patch_size=14, spatial_merge_size=2,
                           temporal_patch_size=2, window_size=56, fullatt_block_indexes=[1]),
        image_token_id=150, video_token_id=151)
    torch.manual_seed(2)
    m = Qwen2_5_VLForConditionalGeneration(cfg).eval()
    m.save_pretrained(tmp_path)
    fe = HFVisionFrontEnd.from_dir(str(tmp_path), "cpu", torch.float32, batched_windows=False)  # HF's own forward, call for call
    grid = torch.tensor([[1, 4, 6], [1, 2, 2]])
    pv = torch.randn(int(grid.prod(1).sum()), 3 * 2 * 14 * 14)
    with torch.no_grad():
        ref = _pooled(m.model.get_image_features(pv, grid))
    mine = fe.features(pv, image_grid_thw=grid)
    assert torch.equal(mine, ref) and mine.shape == (7, 64)  # (4*6 + 2*2) / merge 4
    # the default: equal-length windows attended in one batched call (vision.py: _qwen_vision_attention_batched) — the same numbers; a grid with
    # border windows of several lengths and two images
    fe_b = HFVisionFrontEnd.from_dir(str(tmp_path), "cpu", torch.float32)
    for g in (grid, torch.tensor([[1, 10, 14], [1, 6, 4]])):
        pv = torch.randn(int(g.prod(1).sum()), 3 * 2 * 14 * 14)
        with torch.no_grad():
            ref = _pooled(m.model.get_image_features(pv, g))
        got = fe_b.features(pv, image_grid_thw=g)
        assert got.shape == ref.shape
        torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-6)
        # a second image set on the same grid: the grid's index tables and chunk tables come from the front-end's cache
        pv2 = torch.randn_like(pv)
        with torch.no_grad():
            ref2 = _pooled(m.model.get_image_features(pv2, g))
        torch.testing.assert_close(fe_b.features(pv2, image_grid_thw=g), ref2, rtol=1e-5, atol=1e-6)
    assert len(fe_b._grid_tables) == 2 and all(v for v in fe_b._grid_tables.values())
    tcfg, sd, _ = load_target_dir(str(tmp_path))
    assert tcfg.qkv_bias and tcfg.attn_impl == "sdpa" and tuple(tcfg.mrope_section) == (4, 6, 6)
    assert (tcfg.image_token_index, tcfg.video_token_id, tcfg.num_kv_heads) == (150, 151, 1)
    assert "model.layers.0.self_attn.q_proj.bias" in sd and not any("visual" in k for k in sd)


def test_missing_vision_weights_are_reported(tmp_path):
    from transformers import LlamaConfig, LlamaForCausalLM
    LlamaForCausalLM(LlamaConfig(vocab_size=64, hidden_size=32, intermediate_size=48, num_hidden_layers=1, num_attention_heads=2)).save_pretrained(tmp_path)
    with pytest.raises(NotImplementedError):
        HFVisionFrontEnd.from_dir(str(tmp_path), "cpu", torch.float32)

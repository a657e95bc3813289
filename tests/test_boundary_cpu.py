"""CPU (no GPU): the C-ABI library builds/loads and exports every symbol include/vispec_hip.h declares; the product
path refuses to run without the HIP extension / a GPU (no silent fallback); host-side plumbing."""
import os
import re

import numpy as np
import pytest

from helpers import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "vispec_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vispec_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    from vispec_amd import lib as L
    assert declared_symbols() == sorted(L.SIGNATURES)


def test_library_loads_and_exports_every_symbol():
    from vispec_amd import lib as L
    lib = L.load(build_if_missing=True)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.vispec_version() >= 1


def test_ctypes_structs_match_header_layout():
    import ctypes as C
    from vispec_amd import lib as L
    assert C.sizeof(L.VispecConfig) == 23 * 4
    assert C.sizeof(L.LayerWeights) == 11 * 8 and C.sizeof(L.TargetMisc) == 6 * 8 and C.sizeof(L.DraftWeights) == 17 * 8


def test_no_silent_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vispec_amd import lib as L, synth
    from vispec_amd.engine import DraftConfig, TargetConfig
    from vispec_amd.model import SpecModel
    T = synth.TINY
    tcfg = TargetConfig(T["D"], T["H"], T["H"], T["I"], T["V"], T["NL"], T["max_pos"])
    dcfg = DraftConfig(T["D"], T["H"], T["I"], T["V"], T["max_pos"])
    tw = synth.make_target_weights(T["D"], T["H"], T["I"], T["V"], T["NL"])
    dw = synth.make_draft_weights(T["D"], T["H"], T["I"], T["V"])
    with pytest.raises((L.VispecError, RuntimeError, AssertionError)):
        SpecModel.from_weights(tcfg, dcfg, tw, dw, device="cuda:0")


def test_product_code_never_imports_the_oracle():
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "vispec_amd")):
        for f in files:
            if f.endswith(".py") and re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(d, f)).read(), flags=re.M):
                bad.append(f)
    assert not bad, bad


def test_synth_structured_pair_is_a_successor_model():
    from vispec_amd import synth
    V, D = 1008, 256
    w = synth.make_target_weights(D, 2, 704, V, 1, seed=3, structured=True)
    E, Hd = w["model.embed_tokens.weight"], w["lm_head.weight"]
    s = synth.succ_table(V)
    t = np.arange(3, V)
    assert (np.argmax(E[t] @ Hd.T, axis=1) == s[t]).mean() > 0.999
    assert s.min() >= 3


def test_weights_io_reads_hf_and_vispec_checkpoint_dirs(tmp_path):
    """On-disk formats: sharded HF safetensors of a LLaVA-NeXT target (4.x key layout) + ViSpec draft dir."""
    import json
    import torch
    from safetensors.torch import save_file
    from vispec_amd import synth, weights_io
    T = synth.TINY
    tw = synth.make_target_weights(T["D"], T["H"], T["I"], T["V"], T["NL"], seed=3)
    dw = synth.make_draft_weights(T["D"], T["H"], T["I"], T["V"], seed=4)
    tdir, ddir = tmp_path / "target", tmp_path / "draft"
    tdir.mkdir(); ddir.mkdir()
    hf = {("language_model." + k if k.startswith("model.") else "language_model." + k): torch.from_numpy(v).to(torch.bfloat16) for k, v in tw.items()}
    hf["vision_tower.dummy"] = torch.zeros(4)
    keys = sorted(hf)
    shards = {"model-00001-of-00002.safetensors": keys[: len(keys) // 2], "model-00002-of-00002.safetensors": keys[len(keys) // 2:]}
    wm = {}
    for fn, ks in shards.items():
        save_file({k: hf[k].contiguous() for k in ks}, str(tdir / fn))
        wm.update({k: fn for k in ks})
    json.dump({"weight_map": wm}, open(tdir / "model.safetensors.index.json", "w"))
    json.dump({"architectures": ["LlavaNextForConditionalGeneration"], "image_token_index": T["V"] - 1,
               "text_config": {"hidden_size": T["D"], "num_attention_heads": T["H"], "num_key_value_heads": T["H"], "intermediate_size": T["I"],
                               "vocab_size": T["V"], "num_hidden_layers": T["NL"], "rms_norm_eps": 1e-5}}, open(tdir / "config.json", "w"))
    save_file({k: torch.from_numpy(v).to(torch.bfloat16).contiguous() for k, v in dw.items()}, str(ddir / "model.safetensors"))
    json.dump({"hidden_size": T["D"], "num_attention_heads": T["H"], "intermediate_size": T["I"], "vocab_size": T["V"],
               "max_position_embeddings": T["max_pos"]}, open(ddir / "config.json", "w"))
    tcfg, sd, tok = weights_io.load_target_dir(str(tdir))
    assert (tcfg.hidden_size, tcfg.num_layers, tcfg.architectures[0], tcfg.image_token_index) == (T["D"], T["NL"], "LlavaNextForConditionalGeneration", T["V"] - 1)
    assert set(sd) == set(tw)
    for k in tw:
        np.testing.assert_array_equal(sd[k].float().numpy(), tw[k])
    dcfg, dsd = weights_io.load_draft_dir(str(ddir), tcfg)
    assert set(dsd) == set(dw) and dcfg.hidden_size == T["D"]


def test_bench_byte_model_reproduces_the_surveys_figures():
    """bench.py prices a round with SURVEY.md §8(d)'s formula; the figures the survey prints for the BASELINE models are reproduced
    (B_target 13.215 GB, draft layer 0.539 GB, lm_head 0.263 GB, KV 524 288 B per context row, 18.03 GB per round at n = 2960 / n_c = 820
    for LLaVA-7B; 25.7 GB target for 13B; 14.14 GB / 7.07 GB target and a 1.09 GB lm_head for Qwen2.5-VL-7B bf16 / fp8)."""
    import bench
    from vispec_amd.engine import LLAVA_16_7B, LLAVA_16_13B, QWEN25_VL_7B, TargetConfig
    t7, t13, tq = TargetConfig(**LLAVA_16_7B), TargetConfig(**LLAVA_16_13B), TargetConfig(**QWEN25_VL_7B)
    ar0 = bench.algorithmic_bytes_per_ar_step(t7, 0)
    assert ar0 == 2 * 6_607_339_520                                                   # B_target, exact
    assert bench.algorithmic_bytes_per_ar_step(t7, 1) - ar0 == 524_288                # KV_t per context row
    r0 = bench.algorithmic_bytes_per_round(t7, 0, 0)
    per_pass = (r0 - ar0) / 4                                                         # (1 + d) = 4 draft passes: layer + lm_head
    assert abs(per_pass - (0.539e9 + 0.263e9)) < 2e6
    assert bench.algorithmic_bytes_per_round(t7, 0, 1) - r0 == 4 * 16_384             # KV_d per compressed row, 4 passes
    assert abs(bench.algorithmic_bytes_per_round(t7, 2960, 820) / 1e9 - 18.03) < 0.01
    assert abs(bench.algorithmic_bytes_per_ar_step(t13, 0) / 1e9 - 25.7) < 0.05
    assert abs(bench.algorithmic_bytes_per_ar_step(tq, 0) / 1e9 - 14.14) < 0.01
    assert abs(bench.algorithmic_bytes_per_ar_step(tq, 0, fp8=True) / 1e9 - 7.07) < 0.01
    assert bench.algorithmic_bytes_per_ar_step(tq, 1) - bench.algorithmic_bytes_per_ar_step(tq, 0) == 57_344
    lm = 2 * tq.vocab_size * tq.hidden_size
    assert abs(lm / 1e9 - 1.09) < 0.005

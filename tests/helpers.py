"""Shared builders for tests: tiny seeded models for the oracle (numpy) — same seeds as tests/golden/gen_golden.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import vispec_oracle as vo  # noqa: E402  (tests are allowed to import the oracle)
from vispec_amd import synth  # noqa: E402

T = synth.TINY


def draft_cfg(num_q=2, **kw):
    return vo.DraftConfig(hidden_size=T["D"], num_heads=T["H"], intermediate_size=T["I"], vocab_size=T["V"],
                          max_position_embeddings=T["max_pos"], num_q=num_q, **kw)


def target_cfg(**kw):
    return vo.TargetConfig(hidden_size=T["D"], num_heads=T["H"], num_kv_heads=T["H"], intermediate_size=T["I"],
                           vocab_size=T["V"], num_layers=T["NL"], max_position_embeddings=T["max_pos"], **kw)


def oracle_draft(num_q=2, seed=1, structured=False, target_embed=None, rho=0.115, bf16=False):
    w = synth.make_draft_weights(T["D"], T["H"], T["I"], T["V"], num_q=num_q, seed=seed, structured=structured,
                                 target_embed=target_embed, rho=rho)
    return vo.DraftModel(draft_cfg(num_q), w, bf16=bf16), w


def oracle_target(seed=0, structured=False, bf16=False):
    w = synth.make_target_weights(T["D"], T["H"], T["I"], T["V"], T["NL"], seed=seed, structured=structured)
    return vo.TargetLlama(target_cfg(), w, bf16=bf16), w

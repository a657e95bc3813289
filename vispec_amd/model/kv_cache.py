"""Pre-allocated KV cache, API of reference vispec/model/kv_cache.py (KVCache :4-66, initialize_past_key_values :69-166).

Layout is the reference's: one tensor [2*layers, 1, H_kv, max_pos, head_dim] (K slab then V slab per layer) on the
device — this is the buffer the HIP kernels append to, attend over and compact in place.  The reference keeps the
per-slab lengths in a CPU int64 vector it touches ~10x per round; here the authoritative length is the device-side
`DevState.n_ctx` (csrc/kernels.h) and `current_length_data` is a host mirror refreshed at the loop's one sync per round.
"""
from __future__ import annotations

import torch


class KVCache:
    """View of one K or V slab + its length scalar (kv_cache.py:4-66)."""

    def __init__(self, data: torch.Tensor, current_length: torch.Tensor):
        self.data = data  # [1, H_kv, max_pos, hd]
        self.current_length = current_length  # 0-d view into current_length_data

    @property
    def shape(self):
        return (self.data.shape[0], self.data.shape[1], int(self.current_length.item()), self.data.shape[3])

    def copy(self, indices: torch.Tensor, prev_length: int, dim: int = 2):
        tgt = self.data.index_select(dim, indices)
        dst = self.data.narrow(dim, prev_length, tgt.shape[dim])
        dst.copy_(tgt, non_blocking=True)
        self.current_length.fill_(prev_length + tgt.shape[dim])

    def cat(self, tensor: torch.Tensor, dim: int = 2):
        n = int(self.current_length.item())
        dst = self.data.narrow(dim, n, tensor.shape[dim])
        dst.copy_(tensor)
        self.current_length.add_(tensor.shape[dim])
        return torch.narrow(self.data, 2, 0, n + tensor.shape[dim])


def initialize_past_key_values(model):
    """(past_key_values, [past_key_values_data], current_length_data) as kv_cache.py:69-166.
    `model` is the target wrapper (vispec_amd.model.target.TargetLM): the buffer is the engine's own KV tensor,
    so what this returns IS what the kernels use (no second allocation)."""
    eng = model.engine
    data = eng.target_kv
    nl = model.config.num_hidden_layers
    current_length_data = torch.zeros(nl * 2, dtype=torch.long, device="cpu")
    pkv = [[KVCache(data[2 * i + j], current_length_data[2 * i + j]) for j in range(2)] for i in range(nl)]
    return pkv, [data], current_length_data

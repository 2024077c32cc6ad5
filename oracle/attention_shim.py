"""ORACLE — test infrastructure. Minimal stand-in for `diffusers.models.attention_processor.Attention`
(0.19.3, NOT vendored in the reference; restated from its published behaviour):
the object the reference's processors receive as `attn`. Independent of the product package so that the
golden vectors (tests/golden/make_golden.py: REAL reference processors + this shim) do not depend on it.
"""
import torch
import torch.nn as nn


class Attention(nn.Module):  # class name matters: the reference matches `__class__.__name__ == 'Attention'`

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=40, bias=False, out_bias=True,
                 upcast_attention=False, upcast_softmax=False, dropout=0.0):
        super().__init__()
        inner = heads * dim_head
        cross = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.heads, self.scale = heads, dim_head**-0.5
        self.upcast_attention, self.upcast_softmax = upcast_attention, upcast_softmax
        self.spatial_norm = self.group_norm = self.norm_cross = None
        self.residual_connection, self.rescale_output_factor = False, 1.0
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(cross, inner, bias=bias)
        self.to_v = nn.Linear(cross, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=out_bias), nn.Dropout(dropout)])
        self.processor = None

    def set_processor(self, processor):
        self.processor = processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **cross_attention_kwargs):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **cross_attention_kwargs)

    def head_to_batch_dim(self, t):
        b, s, c = t.shape
        return t.reshape(b, s, self.heads, c // self.heads).permute(0, 2, 1, 3).reshape(b * self.heads, s, c // self.heads)

    def batch_to_head_dim(self, t):
        bh, s, d = t.shape
        return t.reshape(bh // self.heads, self.heads, s, d).permute(0, 2, 1, 3).reshape(bh // self.heads, s, d * self.heads)

    def get_attention_scores(self, query, key, attention_mask=None):
        dtype = query.dtype
        if self.upcast_attention:
            query, key = query.float(), key.float()
        empty = torch.empty(query.shape[0], query.shape[1], key.shape[1], dtype=query.dtype, device=query.device)
        scores = torch.baddbmm(empty, query, key.transpose(-1, -2), beta=0, alpha=self.scale)
        if self.upcast_softmax:
            scores = scores.float()
        return scores.softmax(dim=-1).to(dtype)

    def prepare_attention_mask(self, attention_mask, target_length, batch_size=None):
        assert attention_mask is None
        return None

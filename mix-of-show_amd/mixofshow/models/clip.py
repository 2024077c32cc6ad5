"""CLIP ViT-L/14 text encoder (the SD-1.5 text tower) with the module tree and state-dict keys of the
transformers version the reference was written against (`text_model.encoder.layers.N.self_attn.q_proj`,
...): ED-LoRA checkpoints name their text-encoder LoRA weights by `named_modules()` path
(reference trainer_edlora.py:106-112), and the `where:` option matches the class names
`CLIPAttention` / `CLIPEncoderLayer`. transformers 5.x renamed those paths, so the tower is restated here.
The 77-token causal self-attention (head dim 64) runs on the library's attention kernels on the device (torch SDPA on
the CPU / for fp32 inference); the q/k/v/out_proj Linear layers are LoRA sites and run through the HIP LoRA-linear
kernel once wrapped by LoRALinearLayer.
"""
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from mixofshow.hip.functional import add_layer_norm, layer_norm, quick_gelu


class CLIPTextEmbeddings(nn.Module):

    def __init__(self, vocab_size, hidden, max_pos):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab_size, hidden)
        self.position_embedding = nn.Embedding(max_pos, hidden)
        self.register_buffer('position_ids', torch.arange(max_pos).unsqueeze(0), persistent=False)

    # Trainable concept rows (EDLoRATrainer): ids >= concept_base read from the small `concept_rows` parameter
    # instead of the frozen table, so only 16k x 768 values carry gradients / optimizer state / all-reduce.
    concept_rows = None
    concept_base = None

    def forward(self, input_ids):
        tok = self.token_embedding(input_ids)
        if self.concept_rows is not None:
            is_new = input_ids >= self.concept_base
            new = F.embedding((input_ids - self.concept_base).clamp(min=0, max=self.concept_rows.shape[0] - 1),
                              self.concept_rows)
            tok = torch.where(is_new[..., None], new.to(tok.dtype), tok)
        return tok + self.position_embedding(self.position_ids[:, :input_ids.shape[1]])


class CLIPAttention(nn.Module):

    def __init__(self, hidden, heads):
        super().__init__()
        self.num_heads, self.head_dim = heads, hidden // heads
        self.k_proj = nn.Linear(hidden, hidden)
        self.v_proj = nn.Linear(hidden, hidden)
        self.q_proj = nn.Linear(hidden, hidden)
        self.out_proj = nn.Linear(hidden, hidden)

    def forward(self, x):
        b, s, c = x.shape
        from mixofshow.models.attention import _sites
        if all(getattr(m, '_mos_lora', None) is not None for m in (self.q_proj, self.k_proj, self.v_proj)) and \
                _sites(self.q_proj, self.k_proj, self.v_proj) is not None:
            # all three projections carry ED-LoRA branches (where: CLIPAttention): one fused [Wq;Wk;Wv] GEMM with
            # the three rank-r updates appended to the contraction instead of 3 x (GEMM + down + up) launches. (Ranks
            # whose sum exceeds the packed rank-16 operand, e.g. 3 x 8: one GEMM per projection below.)
            from mixofshow.hip import functional as F_hip
            from mixofshow.models.attention import project
            if getattr(self, '_mos_cache', None) is None:
                object.__setattr__(self, '_mos_cache', F_hip.WeightCache())
            cd = F_hip.compute_dtype_for(x)
            qkv = project(self, 'qkv', [self.q_proj, self.k_proj, self.v_proj], x if x.dtype == cd else x.to(cd), cd)
            if not (torch.is_autocast_enabled('cuda') or x.dtype == cd):
                qkv = qkv.to(x.dtype)
            if x.is_cuda and self.head_dim in (40, 64, 80, 160) and qkv.dtype in (torch.float16, torch.bfloat16):
                # the kernels read q/k/v as column blocks of the fused projection and write ONE (B, 77, 3C) gradient:
                # slicing here would cost autograd three zero-filled buffers, three scatters and two adds per layer
                o = F_hip.attention_qkv(qkv, self.num_heads, self.head_dim**-0.5, causal=True)
                return self.out_proj(o)
            q, k, v = qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:]
        else:
            q, k, v = self.q_proj(x), self.k_proj(x), self.v_proj(x)
        if x.is_cuda and self.head_dim in (40, 64, 80, 160) and (
                q.dtype in (torch.float16, torch.bfloat16) or torch.is_autocast_enabled('cuda')):
            # causal 77-token attention on the library kernels (token-major q/k/v, no head permutes, no aotriton)
            from mixofshow.hip import functional as F_hip
            cd = F_hip.compute_dtype_for(q)
            q, k, v = (t if t.dtype == cd else t.to(cd) for t in (q, k, v))
            o, _ = F_hip.attention(q, k, v, self.num_heads, self.head_dim**-0.5, causal=True)
            return self.out_proj(o)
        shape = (b, s, self.num_heads, self.head_dim)
        q, k, v = (t.view(shape).transpose(1, 2) for t in (q, k, v))
        o = F.scaled_dot_product_attention(q, k, v, is_causal=True)
        return self.out_proj(o.transpose(1, 2).reshape(b, s, c))


class CLIPMLP(nn.Module):

    def __init__(self, hidden, inter):
        super().__init__()
        self.fc1 = nn.Linear(hidden, inter)
        self.fc2 = nn.Linear(inter, hidden)

    def forward(self, x):
        return self.fc2(quick_gelu(self.fc1(x)))       # x * sigmoid(1.702 x): one fused kernel each way on the HIP device


class CLIPEncoderLayer(nn.Module):

    def __init__(self, hidden, heads, inter):
        super().__init__()
        self.self_attn = CLIPAttention(hidden, heads)
        self.layer_norm1 = nn.LayerNorm(hidden)
        self.mlp = CLIPMLP(hidden, inter)
        self.layer_norm2 = nn.LayerNorm(hidden)

    def forward(self, x):
        x = x + self.self_attn(layer_norm(self.layer_norm1, x))
        return x + self.mlp(layer_norm(self.layer_norm2, x))

    def forward_deferred(self, x, pending=None):
        """The same layer with its residual sums formed inside the LayerNorm kernels that consume them: `pending` is the
        previous layer's MLP output, not yet added to the stream x; returns (stream, this layer's pending MLP output)."""
        x, n = add_layer_norm(self.layer_norm1, x, pending)
        x, n = add_layer_norm(self.layer_norm2, x, self.self_attn(n))
        return x, self.mlp(n)


class CLIPEncoder(nn.Module):

    def __init__(self, hidden, heads, inter, layers):
        super().__init__()
        self.layers = nn.ModuleList([CLIPEncoderLayer(hidden, heads, inter) for _ in range(layers)])

    def forward(self, x):
        for l in self.layers:
            x = l(x)
        return x

    def forward_deferred(self, x):
        pending = None
        for l in self.layers:
            if l._forward_hooks or l._forward_pre_hooks:          # someone observes layer outputs: materialise them
                x = l(x if pending is None else x + pending)
                pending = None
            else:
                x, pending = l.forward_deferred(x, pending)
        return x, pending


class CLIPTextTransformer(nn.Module):

    def __init__(self, vocab_size, hidden, heads, inter, layers, max_pos):
        super().__init__()
        self.embeddings = CLIPTextEmbeddings(vocab_size, hidden, max_pos)
        self.encoder = CLIPEncoder(hidden, heads, inter, layers)
        self.final_layer_norm = nn.LayerNorm(hidden)

    def forward(self, input_ids):
        enc = self.encoder
        if enc._forward_hooks or enc._forward_pre_hooks:
            return layer_norm(self.final_layer_norm, enc(self.embeddings(input_ids)))
        x, pending = enc.forward_deferred(self.embeddings(input_ids))
        return add_layer_norm(self.final_layer_norm, x, pending)[1]


SD15_CLIP_CONFIG = dict(vocab_size=49408, hidden_size=768, num_attention_heads=12, intermediate_size=3072,
                        num_hidden_layers=12, max_position_embeddings=77)


class CLIPTextModel(nn.Module):

    def __init__(self, vocab_size=49408, hidden_size=768, num_attention_heads=12, intermediate_size=3072,
                 num_hidden_layers=12, max_position_embeddings=77):
        super().__init__()
        self.config = SimpleNamespace(vocab_size=vocab_size, hidden_size=hidden_size,
                                      num_attention_heads=num_attention_heads, intermediate_size=intermediate_size,
                                      num_hidden_layers=num_hidden_layers,
                                      max_position_embeddings=max_position_embeddings)
        self.text_model = CLIPTextTransformer(vocab_size, hidden_size, num_attention_heads, intermediate_size,
                                              num_hidden_layers, max_position_embeddings)

    @property
    def dtype(self):
        return self.text_model.final_layer_norm.weight.dtype

    @property
    def device(self):
        return self.text_model.final_layer_norm.weight.device

    def get_input_embeddings(self):
        return self.text_model.embeddings.token_embedding

    def resize_token_embeddings(self, new_num_tokens):
        old = self.text_model.embeddings.token_embedding
        if new_num_tokens == old.num_embeddings:
            return old
        new = nn.Embedding(new_num_tokens, old.embedding_dim, device=old.weight.device, dtype=old.weight.dtype)
        new.weight.requires_grad_(old.weight.requires_grad)
        n = min(old.num_embeddings, new_num_tokens)
        with torch.no_grad():
            new.weight[:n] = old.weight[:n]
        self.text_model.embeddings.token_embedding = new
        self.config.vocab_size = new_num_tokens
        return new

    def forward(self, input_ids, attention_mask=None):
        assert attention_mask is None
        return (self.text_model(input_ids), )

"""Inference-time conversion of an ED-LoRA checkpoint (reference mixofshow/utils/convert_edlora_to_diffusers.py:4-99):
add the concept tokens + embeddings to a pipeline and MERGE every LoRA into its base weight,
W' = W + alpha * up @ down — so sampling runs plain fused GEMMs with no LoRA branch."""
import torch

_TE_SITES = ('q_proj', 'k_proj', 'v_proj', 'out_proj', 'fc1', 'fc2')
_UNET_SITES = ('to_q', 'to_k', 'to_v', 'to_out.0', 'ff.net.0.proj', 'ff.net.2', 'proj_out', 'proj_in')


def load_new_concept(pipe, new_concept_embedding, enable_edlora=True):
    new_concept_cfg = {}
    n_per = 16 if enable_edlora else 1
    for idx, (concept_name, concept_embedding) in enumerate(new_concept_embedding.items()):
        names = [f'<new{idx * n_per + layer}>' for layer in range(n_per)]
        added = pipe.tokenizer.add_tokens(names)
        assert added == len(names), 'some token is already in tokenizer'
        ids = [pipe.tokenizer.convert_tokens_to_ids(n) for n in names]
        pipe.text_encoder.resize_token_embeddings(len(pipe.tokenizer))
        table = pipe.text_encoder.get_input_embeddings().weight.data
        table[ids] = concept_embedding.clone().to(table.device, dtype=table.dtype)
        new_concept_cfg[concept_name] = {'concept_token_ids': ids, 'concept_token_names': names}
    return pipe, new_concept_cfg


def lora_down_name(weight_name, model_type):
    for site in (_TE_SITES if model_type == 'text_encoder' else _UNET_SITES):
        weight_name = weight_name.replace(f'{site}.weight', f'{site}.lora_down.weight')
    return weight_name


def merge_lora_into_weight(original_state_dict, lora_state_dict, model_type, alpha):
    assert model_type in ['unet', 'text_encoder']
    merged = {k: v for k, v in original_state_dict.items()}
    count = 0
    for k in list(merged.keys()):
        down = lora_down_name(k, model_type)
        up = down.replace('lora_down', 'lora_up')
        if up in lora_state_dict:
            count += 1
            W = merged[k]
            d = lora_state_dict[down].to(W.device, torch.float32)
            u = lora_state_dict[up].to(W.device, torch.float32)
            delta = (u.squeeze() @ d.squeeze())[..., None, None] if W.dim() == 4 else u @ d
            merged[k] = (W.float() + alpha * delta).to(W.dtype)
    return merged, count


def convert_edlora(pipe, state_dict, enable_edlora, alpha=0.6):
    state_dict = state_dict['params'] if 'params' in state_dict.keys() else state_dict
    new_concept_cfg = {}
    if state_dict.get('new_concept_embedding'):
        pipe, new_concept_cfg = load_new_concept(pipe, state_dict['new_concept_embedding'], enable_edlora)
    merged, _ = merge_lora_into_weight(pipe.unet.state_dict(), state_dict['unet'], 'unet', alpha)
    pipe.unet.load_state_dict(merged)
    merged, _ = merge_lora_into_weight(pipe.text_encoder.state_dict(), state_dict['text_encoder'], 'text_encoder', alpha)
    pipe.text_encoder.load_state_dict(merged)
    return pipe, new_concept_cfg

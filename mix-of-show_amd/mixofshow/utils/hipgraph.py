"""hipGraph replay of a fixed-shape device computation (the UNet call of a sampling loop).

A 50-step sample issues the same ~1.6 k kernel launches 50 times with only the latents and the timestep changing.
`GraphedCall` captures `fn(*tensors)` once (inputs copied into static buffers) and replays it; the host then spends
one launch per denoising step instead of ~1.6 k, and the GPU never waits for the Python dispatcher.
The capture needs node-by-node graph launch on ROCm 7.2 (see mos_path.py).
"""
import os
import warnings

import torch


def sampling_default():
    """Default of the sampling pipelines' `hipgraph=None`: MOS_SAMPLING_HIPGRAPH (1 = replay the UNet call from a
    hipGraph from step 1 on, 0 = eager). On since the kernels outran the Python dispatcher: the 3-region 512x768 sample
    takes 544 ms replayed vs 774 ms eager on the same MI355X (GPU-busy time of the eager loop: ~620 ms)."""
    return os.environ.get('MOS_SAMPLING_HIPGRAPH', SAMPLING_HIPGRAPH_DEFAULT) != '0'


SAMPLING_HIPGRAPH_DEFAULT = '1'


def has_forward_hooks(module):
    """True if any sub-module carries a forward (pre-)hook: Python hooks run at capture time only, so a hooked model
    must be called eagerly."""
    return any(m._forward_hooks or m._forward_pre_hooks for m in module.modules())


def has_python_controllers(module):
    """True if any attention layer reports to a controller that keeps Python-side state (stores or edits attention maps,
    counts layers and steps: `revise_edlora_unet_attention_controller_forward(unet, controller)` with anything but a
    pass-through). Such a controller runs at capture time only -- a replayed UNet call would leave its stored maps, its
    `cur_step` / `between_steps` bookkeeping and its edits silently stale -- so the model must be called eagerly, however the
    controller got there (pipe.set_controller or the public revise_* function)."""
    for m in module.modules():
        ctrl = getattr(getattr(m, 'processor', None), 'controller', None)
        if ctrl is not None and not getattr(ctrl, 'is_passthrough', False):
            return True
    return False


def model_epoch(module):
    """A value that changes whenever a captured graph of `module` may have gone stale: parameter / buffer storage (address,
    dtype) or content (tensor version counters: load_state_dict, in-place merges, optimiser steps). The graph bakes in not
    only the parameters' addresses but those of derived copies (fused / cast weights, fp32 affine parameters) that the
    product re-allocates when a version changes."""
    h = 0
    for t in list(module.parameters()) + list(module.buffers()):
        h = (h * 1000003 + hash((t.data_ptr(), t._version, t.dtype))) & 0xFFFFFFFFFFFFFFFF
    # the derived copies are also keyed by COMPUTE dtype (functional.WeightCache / _ConvWeights): the same weights sampled under
    # another autocast dtype re-allocate them, and a graph captured before would replay against freed addresses
    ac = (torch.is_autocast_enabled('cuda'), torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled('cuda') else None)
    return (h * 1000003 + hash(ac)) & 0xFFFFFFFFFFFFFFFF


def clear_sampling_graphs(pipe):
    """Drop the captured UNet graphs a sampling pipeline keeps across calls (and with them their private memory pools): call
    when validation ends and training resumes, or before changing weights out of band."""
    graphs = pipe.__dict__.get('_sampling_graphs')
    n = len(graphs) if graphs else 0
    if graphs:
        graphs.clear()
    return n


def graphs_usable(device):
    return (torch.device(device).type == 'cuda' and torch.cuda.is_available()
            and os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE') == '0')


class GraphedCall:
    """`fn` must be free of host<->device synchronisation and allocate only through torch. Everything `fn` reads
    besides `example_inputs` (weights, prompt embeddings, cached K/V, adapter features) is baked in by address and must
    stay alive and unchanged in place for the lifetime of this object."""

    def __init__(self, fn, *example_inputs):
        self.static_in = [x.clone() for x in example_inputs]
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = fn(*self.static_in)

    def __call__(self, *inputs):
        for s, x in zip(self.static_in, inputs):
            s.copy_(x)
        self.graph.replay()
        return self.static_out


def try_capture(fn, *example_inputs):
    """GraphedCall or None (with a warning): capture is an optimisation of launch overhead, the eager path computes
    the same kernels."""
    prev = torch.cuda.current_stream()
    try:
        return GraphedCall(fn, *example_inputs)
    except Exception:  # noqa: BLE001
        import traceback
        # torch.cuda.graph.__exit__ does not restore the stream when ending an invalidated capture raises: put the
        # caller's stream back and retire the poisoned capture stream before going on eagerly
        torch.cuda.set_stream(prev)
        torch.cuda.graph.default_capture_stream = None
        warnings.warn('hipGraph capture failed; continuing eagerly\n' + traceback.format_exc())
        return None

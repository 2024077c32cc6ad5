"""The ED-LoRA optimisation step (reference train_edlora.py:105-162) as a reusable engine: used by the
`train_edlora.py` entry point, by bench.py and by the multi-rank tests.

Per step (one rank = one GPU): forward+backward of EDLoRATrainer under autocast, ONE all-reduce of the flat
LoRA+concept-row gradient bucket (mixofshow.parallel.dp), AdamW with the reference's three parameter groups and
linear LR decay, then the embedding-norm rule (:138-143): once the mean norm of the concept rows reaches
`emb_norm_threshold` the rows are frozen. The reference implements the freeze by restoring rows from a snapshot
after every step and reads the norm on the host each step; here snapshot/restore is a device-side `where` on a
device flag, so a training step contains no device->host synchronisation.
"""
import torch

from mixofshow.parallel import dp


class TrainEngine:

    def __init__(self, trainer, train_opt, total_iter, mixed_precision='fp16', grad_accum=1, frozen_weights_half=True,
                 channels_last=True):
        self.trainer = trainer
        self.total_iter = total_iter
        self.grad_accum = grad_accum
        optim_cfg = dict(train_opt['optim_g'])
        optim_type = optim_cfg.pop('type')
        assert optim_type == 'AdamW', 'only support AdamW now'
        groups = trainer.get_params_to_optimize()
        dev = trainer.concept_embedding.device
        # fused multi-tensor AdamW on the device (one kernel per group instead of ~10 foreach kernels over 105 small
        # tensors); it also takes GradScaler's grad_scale / found_inf on the device, so the fp16 step has no host sync
        if dev.type == 'cuda':
            self.optimizer = torch.optim.AdamW(groups, **optim_cfg, fused=True)
        else:
            self.optimizer = torch.optim.AdamW(groups, **optim_cfg, foreach=True)
        self.base_lrs = [g['lr'] for g in self.optimizer.param_groups]
        self.bucket = dp.FlatGradBucket(trainer.trainable_parameters())
        # the LoRA backward kernel accumulates straight into the bucket's `.grad` views (no per-parameter glue kernels):
        # switched on only around this engine's own forward/backward (F_hip.direct_grad_accumulation)
        self.mixed_precision = mixed_precision
        self.amp_dtype = {'fp16': torch.float16, 'bf16': torch.bfloat16}.get(mixed_precision)
        self.scaler = torch.amp.GradScaler('cuda', enabled=(mixed_precision == 'fp16' and dev.type == 'cuda'))
        if self.amp_dtype is not None and frozen_weights_half and dev.type == 'cuda':
            self._store_frozen_weights_in_half(trainer, self.amp_dtype)
        self.channels_last = bool(channels_last) and dev.type == 'cuda'
        if self.channels_last:
            # NHWC is the native layout of the token-major attention path and of MIOpen's fp16 implicit-GEMM convs: no
            # NCHW<->NHWC transposes around convolutions, no permute copies around the transformer blocks (measured
            # 59.3 -> 55.7 ms/step on configs[1]); GroupNorm runs on the channels-last kernels of mos_norm.hip
            trainer.unet.to(memory_format=torch.channels_last)
            trainer.vae.to(memory_format=torch.channels_last)
        self.threshold = float(train_opt.get('emb_norm_threshold', 5.5e-1))
        self.stop_flag = torch.zeros((), dtype=torch.bool, device=dev)          # stop_emb_update, on device
        self.frozen_rows = trainer.concept_embedding.detach().clone()
        self.global_step = 0
        self._micro = 0

    @staticmethod
    @torch.no_grad()
    def _store_frozen_weights_in_half(trainer, dtype):
        """Frozen Conv/Linear weights are only ever consumed through autocast, which rounds the fp32 tensor to
        half on EVERY use (~700 cast kernels and ~5 GB of traffic per step for SD-1.5). Rounding once and keeping
        the half tensor gives bit-identical operands. Trainable tensors (LoRA factors, concept rows) and everything
        autocast runs in fp32 (norm affine parameters, embeddings) keep their fp32 masters."""
        import torch.nn as nn
        for root in (trainer.vae, trainer.text_encoder, trainer.unet):
            for m in root.modules():
                if isinstance(m, (nn.Linear, nn.Conv2d)):
                    for p in (m.weight, m.bias):
                        if p is not None and not p.requires_grad and p.dtype == torch.float32:
                            p.data = p.data.to(dtype)

    # ---- hipGraph mode ------------------------------------------------------------------------------------
    def enable_graph(self, example_batch, warmup=2):
        """Capture forward + backward of one micro-batch (about 4.5 k kernel launches for SD-1.5) in a hipGraph and
        replay it every step. Inputs live in static device buffers; the host only tokenises, copies and replays.
        The all-reduce, the optimiser update, GradScaler bookkeeping and the embedding-norm rule stay eager (a
        handful of launches). Shapes (batch, image size, tokens per prompt) are fixed by `example_batch`."""
        assert self.grad_accum == 1, 'graph mode captures exactly one micro-batch per optimiser step'
        import os
        if os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE') != '0':
            raise RuntimeError('TrainEngine.enable_graph needs DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment '
                               'before the HIP runtime starts (mos_path.py sets it): the ROCm 7.2 packet-capture '
                               'graph path faults when eager launches interleave with replays of this graph')
        tr = self.trainer
        dev = tr.concept_embedding.device
        st = {}
        for k in ('images', 'masks', 'img_masks', 'noise', 'timesteps', 'latents', 'latent_noise'):
            v = example_batch.get(k)
            if torch.is_tensor(v):
                v = v.to(dev)
                if k == 'images' and self.channels_last:
                    v = v.contiguous(memory_format=torch.channels_last)
                st[k] = v.clone()
        B = (st.get('images', st.get('latents'))).shape[0]
        ids, pos = tr.tokenize(example_batch['prompts'], B)
        st['ids'] = ids.to(dev)
        st['pos'] = pos.to(dev) if pos is not None else None
        self._static = st
        self._token_cache = {}

        from mixofshow.hip import functional as F_hip

        # the captured launch of the deferred LoRA gradient sums bakes in ITS record table and workspaces: a store owned by
        # this graph alone (warm-up fills it, capture replays it, then it is frozen); eager steps use another one
        self._finals_graph = F_hip.new_deferred_finals()

        def fwd_bwd():
            self.bucket.zero()
            with F_hip.direct_grad_accumulation(defer_finals=dev.type == 'cuda', store=self._finals_graph):
                with torch.autocast(dev.type, dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
                    loss = tr(st.get('images'), None, st.get('masks', st['img_masks']), st['img_masks'],
                              noise=st.get('noise'), timesteps=st.get('timesteps'), latents=st.get('latents'),
                              latent_noise=st.get('latent_noise'),
                              text_input_ids=st['ids'], token_positions=st['pos'])
                self.scaler.scale(loss).backward()
            return loss.detach()

        self._graph, self._static_loss = self._capture(fwd_bwd, warmup)
        self._finals_graph.freeze()
        F_hip.freeze_lora_packs(True)          # the graph holds the descriptor table's address and group count
        return self

    @staticmethod
    def _capture(fwd_bwd, warmup):
        """Warm up on a side stream, then capture `fwd_bwd()` into a hipGraph. Returns (graph with .replay(), the static
        tensor the captured call returned). Kept apart from enable_graph so that the host logic around a replay (static input
        copies, eager all-reduce, GradScaler bookkeeping across ranks) can be exercised on the CPU with a stand-in."""
        from mixofshow.hip import functional as F_hip
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fwd_bwd()
        torch.cuda.current_stream().wait_stream(side)
        F_hip.invalidate_lora_packs()          # the one-launch repack of all LoRA operands must be part of the graph
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_loss = fwd_bwd()
        return graph, static_loss

    def disable_graph(self):
        """Back to the eager step (also releases the LoRA descriptor table)."""
        if getattr(self, '_graph', None) is not None:
            from mixofshow.hip import functional as F_hip
            self._graph = None
            self._finals_graph = None
            F_hip.freeze_lora_packs(False)

    def __del__(self):
        try:
            self.disable_graph()
        except Exception:
            pass

    def _graph_step(self, batch):
        tr, st = self.trainer, self._static
        # token ids of a caption set are a pure function of the strings: keep them on the device (a concept's
        # training set is a handful of captions), so a steady-state step has no host->device traffic at all
        key = tuple(batch['prompts'])
        ent = self._token_cache.get(key)
        if ent is None:
            ids, pos = tr.tokenize(batch['prompts'], st['ids'].shape[0] // (16 if tr.enable_edlora else 1))
            dev = st['ids'].device
            ent = (ids.to(dev), pos.to(dev) if pos is not None else None)     # blocking uploads
            if len(self._token_cache) < 4096:
                self._token_cache[key] = ent
        # the graph is valid for the captured shapes only (a caption with a different number of concept tokens, a ragged
        # last batch): such a step runs eagerly instead of silently broadcasting into the static buffers
        same = ent[0].shape == st['ids'].shape and ((ent[1] is None) == (st['pos'] is None)) and (
            ent[1] is None or ent[1].shape == st['pos'].shape)
        same = same and all(batch[k].shape == st[k].shape for k in st if k in batch and torch.is_tensor(batch.get(k)))
        if not same:
            return self._eager_step(batch)
        for k in ('images', 'masks', 'img_masks', 'noise', 'timesteps', 'latents', 'latent_noise'):
            if k in st and torch.is_tensor(batch.get(k)):
                v = batch[k]
                # an async copy out of pageable host memory may run after the host tensor is gone: only device or
                # pinned sources are copied without blocking
                st[k].copy_(v, non_blocking=(v.is_cuda or v.is_pinned()))
        st['ids'].copy_(ent[0])
        if ent[1] is not None:
            st['pos'].copy_(ent[1])
        self._graph.replay()
        return self._finish_step(self._static_loss)

    def lr_factor(self, step):
        # diffusers get_scheduler('linear', warmup 0): lr * max(0, (T - step) / T)  (train_edlora.py:85-90)
        return max(0.0, float(self.total_iter - step) / float(max(1.0, self.total_iter)))

    def step(self, batch):
        """One micro-batch; performs the optimiser update every `grad_accum` calls. Returns a dict of device scalars."""
        if getattr(self, '_graph', None) is not None:
            return self._graph_step(batch)
        return self._eager_step(batch)

    def _eager_step(self, batch):
        from mixofshow.hip import functional as F_hip
        tr = self.trainer
        if self._micro == 0:
            self.bucket.zero()
        masks = batch['masks'] if 'masks' in batch else batch['img_masks']
        extra = {k: batch[k] for k in ('noise', 'timesteps', 'latents', 'latent_noise') if k in batch}
        dev_type = tr.concept_embedding.device.type
        images = batch['images']
        if self.channels_last and images is not None:
            images = images.contiguous(memory_format=torch.channels_last)
        if getattr(self, '_finals_eager', None) is None:
            self._finals_eager = F_hip.new_deferred_finals()     # never the table a captured graph replays
        with F_hip.direct_grad_accumulation(defer_finals=dev_type == 'cuda', store=self._finals_eager):
            with torch.autocast(dev_type, dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
                loss = tr(images, batch['prompts'], masks, batch['img_masks'], **extra)
            self.scaler.scale(loss / self.grad_accum).backward()
        self._micro += 1
        if self._micro < self.grad_accum:
            return {'loss': loss.detach()}
        self._micro = 0
        return self._finish_step(loss.detach())

    def _finish_step(self, loss):
        tr = self.trainer
        out = {'loss': loss}
        self.bucket.allreduce_mean()                      # RCCL all-reduce of LoRA + concept-row grads only
        for g, base in zip(self.optimizer.param_groups, self.base_lrs):
            g['lr'] = base * self.lr_factor(self.global_step)
        self.scaler.step(self.optimizer)
        self.scaler.update()
        if tr.concept_embedding.is_cuda:
            # torch's fused AdamW writes the fp32 masters without bumping their version counters: tell the operand cache
            # (an eager step would otherwise run on the factors of the step before; replays repack inside the graph)
            from mixofshow.hip import functional as F_hip
            F_hip.invalidate_lora_packs()
        with torch.no_grad():
            rows = tr.concept_embedding
            # freeze rule: rows stay at their snapshot once stop_flag is set (reference :123-126,135-136)
            rows.copy_(torch.where(self.stop_flag, self.frozen_rows, rows))
            norm_mean = rows.norm(dim=-1).mean()
            newly = (~self.stop_flag) & (norm_mean >= self.threshold)
            self.frozen_rows = torch.where(newly, rows, self.frozen_rows)
            self.stop_flag = self.stop_flag | newly
        self.global_step += 1
        out['Norm_mean'] = norm_mean
        return out

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
                 channels_last=False):
        self.trainer = trainer
        self.total_iter = total_iter
        self.grad_accum = grad_accum
        optim_cfg = dict(train_opt['optim_g'])
        optim_type = optim_cfg.pop('type')
        assert optim_type == 'AdamW', 'only support AdamW now'
        groups = trainer.get_params_to_optimize()
        dev = trainer.concept_embedding.device
        self.optimizer = torch.optim.AdamW(groups, **optim_cfg, foreach=True)
        self.base_lrs = [g['lr'] for g in self.optimizer.param_groups]
        self.bucket = dp.FlatGradBucket(trainer.trainable_parameters())
        self.mixed_precision = mixed_precision
        self.amp_dtype = {'fp16': torch.float16, 'bf16': torch.bfloat16}.get(mixed_precision)
        self.scaler = torch.amp.GradScaler('cuda', enabled=(mixed_precision == 'fp16' and dev.type == 'cuda'))
        if self.amp_dtype is not None and frozen_weights_half and dev.type == 'cuda':
            self._store_frozen_weights_in_half(trainer, self.amp_dtype)
        self.channels_last = bool(channels_last) and dev.type == 'cuda'
        if self.channels_last:
            # NHWC is the native layout of the token-major attention path and of MIOpen's fp16 implicit-GEMM convs
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

    def lr_factor(self, step):
        # diffusers get_scheduler('linear', warmup 0): lr * max(0, (T - step) / T)  (train_edlora.py:85-90)
        return max(0.0, float(self.total_iter - step) / float(max(1.0, self.total_iter)))

    def step(self, batch):
        """One micro-batch; performs the optimiser update every `grad_accum` calls. Returns a dict of device scalars."""
        tr = self.trainer
        if self._micro == 0:
            self.bucket.zero()
        masks = batch['masks'] if 'masks' in batch else batch['img_masks']
        extra = {k: batch[k] for k in ('noise', 'timesteps', 'latents') if k in batch}
        dev_type = tr.concept_embedding.device.type
        images = batch['images']
        if self.channels_last and images is not None:
            images = images.contiguous(memory_format=torch.channels_last)
        with torch.autocast(dev_type, dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
            loss = tr(images, batch['prompts'], masks, batch['img_masks'], **extra)
        self.scaler.scale(loss / self.grad_accum).backward()
        self._micro += 1
        out = {'loss': loss.detach()}
        if self._micro < self.grad_accum:
            return out
        self._micro = 0
        self.bucket.allreduce_mean()                      # RCCL all-reduce of LoRA + concept-row grads only
        for g, base in zip(self.optimizer.param_groups, self.base_lrs):
            g['lr'] = base * self.lr_factor(self.global_step)
        self.scaler.step(self.optimizer)
        self.scaler.update()
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

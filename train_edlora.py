"""ED-LoRA training entry point — `python train_edlora.py -opt options/train/...yml`, or one process per GPU with
`python -m torch.distributed.run --nproc-per-node N train_edlora.py -opt ...` (the reference uses `accelerate launch`,
train_edlora.py:28-198). Same YAML option surface; the optimisation step lives in mixofshow.pipelines.train_loop."""
import argparse
import logging
import os
import os.path as osp
import time

import mos_path  # noqa: F401
import torch

from mixofshow.data.lora_dataset import build_train_dataset
from mixofshow.data.prompt_dataset import PromptDataset
from mixofshow.parallel import dp
from mixofshow.pipelines.pipeline_edlora import EDLoRAPipeline, StableDiffusionPipeline  # noqa: F401
from mixofshow.pipelines.train_loop import TrainEngine
from mixofshow.pipelines.trainer_edlora import EDLoRATrainer
from mixofshow.utils.convert_edlora_to_diffusers import convert_edlora
from mixofshow.utils.options import dict2str, load_options


def _to_device(batch, device):
    return {k: (v.to(device, non_blocking=v.is_pinned()) if torch.is_tensor(v) else v) for k, v in batch.items()}


def train(root_path, args):
    opt = load_options(args.opt)
    rank, world, local = dp.init_distributed()
    device = torch.device('cuda', local) if torch.cuda.is_available() else torch.device('cpu')
    logging.basicConfig(level=logging.INFO if rank == 0 else logging.WARNING,
                        format='%(asctime)s %(levelname)s: %(message)s')
    logger = logging.getLogger('mixofshow')
    exp_dir = osp.join(root_path, 'experiments', opt['name'])
    opt.setdefault('path', {})
    opt['path']['models'] = osp.join(exp_dir, 'models')
    opt['path']['visualization'] = osp.join(exp_dir, 'visualization')
    if rank == 0:
        os.makedirs(opt['path']['models'], exist_ok=True)
        logger.info(dict2str(opt))
    if opt.get('manual_seed') is not None:
        torch.manual_seed(opt['manual_seed'])      # same stream on every rank, like accelerate.set_seed (:47-48)

    trainer = EDLoRATrainer(**opt['models']).to(device)
    trainset_cfg = opt['datasets']['train']
    train_dataset = build_train_dataset(trainset_cfg)
    sampler = torch.utils.data.distributed.DistributedSampler(train_dataset, world, rank, shuffle=True) if world > 1 else None
    loader = torch.utils.data.DataLoader(train_dataset, batch_size=trainset_cfg['batch_size_per_gpu'],
                                         shuffle=sampler is None, sampler=sampler, drop_last=True)
    accum = opt.get('gradient_accumulation_steps', 1)
    total_batch = trainset_cfg['batch_size_per_gpu'] * world * accum
    total_iter = len(train_dataset) / total_batch       # a float, like the reference (:74)
    opt['train']['total_iter'] = total_iter
    engine = TrainEngine(trainer, opt['train'], total_iter, opt.get('mixed_precision', 'fp16'), accum,
                         channels_last=bool(opt['train'].get('channels_last', True)))
    logger.info(f'***** Running training *****  examples={len(train_dataset)} per-device batch='
                f"{trainset_cfg['batch_size_per_gpu']} total batch={total_batch} steps={total_iter} "
                f'grad bucket={engine.bucket.nbytes / 1e6:.2f} MB world={world}')

    def batches():
        epoch = 0
        while True:
            if sampler is not None:
                sampler.set_epoch(epoch)
            for b in loader:
                yield b
            epoch += 1
            dp.barrier()

    it = batches()
    trainer.unet.train()
    trainer.text_encoder.train()
    t0 = time.time()
    # default ON (bench.py measures this mode): forward+backward replayed from a hipGraph; `train.hipgraph: false` opts out
    use_graph = bool(opt['train'].get('hipgraph', True)) and device.type == 'cuda' and accum == 1
    while engine.global_step < total_iter:
        batch = _to_device(next(it), device)
        if use_graph and getattr(engine, '_graph', None) is None:
            try:
                engine.enable_graph(batch)   # steps whose shapes differ from this batch run eagerly (TrainEngine._graph_step)
                if rank == 0:
                    logger.info('forward+backward captured in a hipGraph')
            except Exception as e:           # e.g. DEBUG_CLR_GRAPH_PACKET_CAPTURE preset in the environment: train eagerly
                use_graph = False
                logger.warning(f'hipGraph capture unavailable ({type(e).__name__}: {e}); training eagerly')
        out = engine.step(batch)
        if 'Norm_mean' not in out:
            continue
        step = engine.global_step
        if step % opt['logger']['print_freq'] == 0:
            log = dp.reduce_loss_dict(out)
            lrs = [g['lr'] for g in engine.optimizer.param_groups]
            eta = (time.time() - t0) / step * (total_iter - step)
            logger.info(f'[iter {step}/{int(total_iter)}] lrs={lrs} eta={eta:.0f}s ' +
                        ' '.join(f'{k}: {float(v):.4e}' for k, v in log.items()))
        if step % int(opt['logger']['save_checkpoint_freq']) == 0:
            save_and_validation(opt, trainer, step, logger, rank)
    dp.barrier()
    save_and_validation(opt, trainer, 'latest', logger, rank)


def save_and_validation(opt, trainer, global_step, logger, rank):
    enable_edlora = opt['models']['enable_edlora']
    save_path = os.path.join(opt['path']['models'], f"{'edlora' if enable_edlora else 'lora'}_model-{global_step}.pth")
    if rank == 0:
        torch.save({'params': trainer.delta_state_dict()}, save_path)
        logger.info(f'Save state to {save_path}')
    dp.barrier()
    if opt.get('val', {}).get('val_during_save'):
        from test_edlora import visual_validation
        valset_cfg = opt['datasets']['val_vis']
        # the reference shards the validation loader over the ranks (`accelerator.prepare(val_dataloader)`, train_edlora.py:70):
        # rank r samples prompts r, r + world, ... -- every prompt once, no two ranks writing the same PNG
        val_set = PromptDataset(valset_cfg)
        if dp.world_size() > 1:
            val_set = torch.utils.data.Subset(val_set, range(rank, len(val_set), dp.world_size()))
        loader = torch.utils.data.DataLoader(val_set, batch_size=valset_cfg['batch_size_per_gpu'])
        for lora_alpha in opt['val']['alpha_list']:
            pipeclass = EDLoRAPipeline if enable_edlora else StableDiffusionPipeline           # reference :179
            pipe = pipeclass.from_pretrained(opt['models']['pretrained_path'], torch_dtype=torch.float16)
            pipe.to(trainer.concept_embedding.device)
            pipe, cfg = convert_edlora(pipe, torch.load(save_path, weights_only=False), enable_edlora=enable_edlora,
                                       alpha=lora_alpha)
            pipe.set_new_concept_cfg(cfg)
            visual_validation(None, pipe, loader, f'Iters-{global_step}_Alpha-{lora_alpha}', opt)     # reference :187
            del pipe
        dp.barrier()


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('-opt', type=str, default='options/train/EDLoRA/synthetic/8101_EDLoRA_potter_synthetic_B4.yml')
    args = parser.parse_args()
    train(osp.abspath(osp.join(__file__, osp.pardir)), args)

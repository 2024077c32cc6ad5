"""Data-parallel path on CPU: 2 processes, gloo backend, HIP primitives emulated by the oracle (tests only).
Checks that one all-reduce of the flat LoRA + concept-row bucket reproduces the single-process average of the
per-rank gradients (what the reference's DDP computes) and that the ranks stay in lock-step after AdamW."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup_emulation():
    sys.path.insert(0, ROOT)
    import mos_path  # noqa: F401
    import mixofshow.hip.ops as ops
    from oracle import emu_ops
    for name in emu_ops.EMULATED:
        setattr(ops, name, getattr(emu_ops, name))


def _make_engine(attn_reg=True):
    from tests.test_host_cpu import _trainer
    from mixofshow.pipelines.train_loop import TrainEngine
    tr = _trainer() if attn_reg else _trainer(attn_reg_weight=None)
    opt = dict(optim_g=dict(type='AdamW', lr=0.0, weight_decay=0.01, betas=[0.9, 0.999]), emb_norm_threshold=0.55)
    return tr, TrainEngine(tr, opt, total_iter=10, mixed_precision='no')


def _rank_batch(rank):
    g = torch.Generator().manual_seed(100 + rank)
    masks = torch.zeros(1, 1, 16, 16)
    masks[:, :, 4:12, 3:11] = 1
    return dict(images=None, prompts=['a <potter1> <potter2> in the park'], masks=masks,
                img_masks=torch.ones_like(masks), latents=torch.randn(1, 4, 16, 16, generator=g),
                noise=torch.randn(1, 4, 16, 16, generator=g), timesteps=torch.randint(0, 1000, (1, ), generator=g))


def _worker(rank, world, port, out_dir, attn_reg=True):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2 if world <= 2 else 1)
    _setup_emulation()
    from mixofshow.parallel import dp
    dp.init_distributed(backend='gloo')
    tr, engine = _make_engine(attn_reg)
    # gradient of this rank's batch, reduced
    engine.bucket.zero()
    loss = tr(**_rank_batch(rank))
    loss.backward()
    local = engine.bucket.flat.clone()
    engine.bucket.allreduce_mean()
    reduced = engine.bucket.flat.clone()
    # then one full engine step (forward, backward, all-reduce, AdamW, norm rule)
    out = engine.step(_rank_batch(rank))
    logged = dp.reduce_loss_dict(out)
    params = torch.cat([p.detach().reshape(-1) for p in tr.trainable_parameters()])
    torch.save(dict(local=local, reduced=reduced, params=params, loss=float(logged['loss']),
                    nbytes=engine.bucket.nbytes), os.path.join(out_dir, f'rank{rank}.pt'))
    dp.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_allreduce_matches_single_process_average(tmp_path, emulated_hip):
    world, port = 2, 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / 'rank0.pt')
    r1 = torch.load(tmp_path / 'rank1.pt')
    # bucket holds exactly concept rows + LoRA factors (tiny preset: 32x64 + 2*36 LoRA tensors), nothing else
    assert r0['nbytes'] == r1['nbytes'] and r0['local'].numel() == r0['nbytes'] // 4
    assert not torch.equal(r0['local'], r1['local'])                       # ranks saw different data
    torch.testing.assert_close(r0['reduced'], r1['reduced'], rtol=0, atol=0)  # identical after the collective
    torch.testing.assert_close(r0['reduced'], (r0['local'] + r1['local']) / 2, rtol=0, atol=1e-9)
    torch.testing.assert_close(r0['params'], r1['params'], rtol=0, atol=0)    # lock-step after AdamW
    assert abs(r0['loss'] - r1['loss']) < 1e-12                               # reduce_loss_dict averaged
    # single-process reference: same two batches, gradients averaged by hand (emulation through the fixture, so
    # the patch does not leak into later tests of this process)
    tr, engine = _make_engine()
    grads = []
    for rank in range(world):
        engine.bucket.zero()
        tr(**_rank_batch(rank)).backward()
        grads.append(engine.bucket.flat.clone())
    torch.testing.assert_close(r0['reduced'], (grads[0] + grads[1]) / 2, rtol=1e-5, atol=1e-8)


@pytest.mark.timeout(900)
@pytest.mark.parametrize('world', [4, 8])
def test_many_rank_allreduce_equals_global_batch_gradient(tmp_path, emulated_hip, world):
    """SURVEY 8(d) cfg #3 on CPU ranks: per-rank batches with rank-r seeds, ONE all-reduce(mean) of the flat bucket ==
    the gradient of a single process that sees the GLOBAL batch (world samples at once). The attention regulariser is
    off here: it normalises maps by maxima over the batch a process holds, so it is per-rank by construction (in the
    reference's DDP too); the 2-rank test above covers it against the hand-averaged per-rank gradients."""
    port = 33500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, str(tmp_path), False), nprocs=world, join=True)
    rs = [torch.load(tmp_path / f'rank{r}.pt') for r in range(world)]
    for r in rs[1:]:
        torch.testing.assert_close(r['reduced'], rs[0]['reduced'], rtol=0, atol=0)     # identical on every rank
        torch.testing.assert_close(r['params'], rs[0]['params'], rtol=0, atol=0)       # lock-step after AdamW
    mean_local = torch.stack([r['local'] for r in rs]).mean(0)
    torch.testing.assert_close(rs[0]['reduced'], mean_local, rtol=1e-6, atol=1e-9)
    # single process, global batch of `world` samples (same per-sample tensors, concatenated)
    tr, engine = _make_engine(attn_reg=False)
    bs = [_rank_batch(r) for r in range(world)]
    big = dict(images=None, prompts=[b['prompts'][0] for b in bs], masks=torch.cat([b['masks'] for b in bs]),
               img_masks=torch.cat([b['img_masks'] for b in bs]), latents=torch.cat([b['latents'] for b in bs]),
               noise=torch.cat([b['noise'] for b in bs]), timesteps=torch.cat([b['timesteps'] for b in bs]))
    engine.bucket.zero()
    tr(**big).backward()
    g = engine.bucket.flat.clone()
    err = ((rs[0]['reduced'] - g).norm() / g.norm()).item()
    print(f'[parity] dp{world} (gloo): all-reduced bucket vs single-process global-batch-{world} gradient: rel L2 {err:.2e}')
    # (the emulated kernels round activations to half like the HIP ones: a batch of `world` changes the fp32 summation order
    # inside the GEMMs, which flips half roundings -> ~1e-3 rel L2 of rounding noise, no systematic term)
    assert err <= 3e-3


def _cli_worker(rank, world, port, root, recipe):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    _setup_emulation()
    import argparse
    import train_edlora
    from mixofshow.pipelines.train_loop import TrainEngine
    seen = {}
    real_step = TrainEngine.step

    def spy(self, batch):                      # remember the last engine to read the final parameters
        seen['engine'] = self
        seen.setdefault('prompts', []).append(tuple(batch['prompts']))
        return real_step(self, batch)

    TrainEngine.step = spy
    train_edlora.train(root, argparse.Namespace(opt=recipe))
    eng = seen['engine']
    params = torch.cat([p.detach().reshape(-1) for p in eng.trainer.trainable_parameters()])
    torch.save(dict(params=params, steps=eng.global_step, n_batches=len(seen['prompts'])),
               os.path.join(root, f'cli_rank{rank}.pt'))


@pytest.mark.timeout(900)
def test_two_rank_train_cli(tmp_path):
    """train_edlora.train under a 2-rank gloo group: DistributedSampler shards the data, every step all-reduces the
    bucket, both ranks finish with identical parameters, the step count is len(dataset) / (batch * world) and only
    rank 0 writes the checkpoint."""
    from tests.test_train_cli_cpu import _recipe
    recipe = _recipe(tmp_path, total_images=4, val=False)      # 4 images x enlarge 2 = 8 samples
    world, port = 2, 31500 + (os.getpid() % 2000)
    mp.spawn(_cli_worker, args=(world, port, str(tmp_path), recipe), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / 'cli_rank0.pt'), torch.load(tmp_path / 'cli_rank1.pt')
    assert r0['steps'] == r1['steps'] == 2                     # 8 samples / (batch 2 x world 2)
    assert r0['n_batches'] == r1['n_batches'] == 2
    torch.testing.assert_close(r0['params'], r1['params'], rtol=0, atol=0)
    models = tmp_path / 'experiments' / 'cli_cpu' / 'models'
    assert sorted(os.listdir(models)) == ['edlora_model-latest.pth']


# ---- graph replay + eager all-reduce + GradScaler found_inf agreement across ranks ----------------------
class _FakeGraph:
    """Stand-in for torch.cuda.CUDAGraph on the CPU: replay() re-runs the captured closure and refreshes the static output
    in place, which is what a replay does to the captured kernels' output buffers."""

    def __init__(self, fn, out):
        self.fn, self.out, self.replays = fn, out, 0

    def replay(self):
        self.out.copy_(self.fn())
        self.replays += 1


def _graph_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), DEBUG_CLR_GRAPH_PACKET_CAPTURE='0')
    torch.set_num_threads(2)
    _setup_emulation()
    from mixofshow.parallel import dp
    from mixofshow.pipelines.train_loop import TrainEngine
    dp.init_distributed(backend='gloo')
    from tests.test_host_cpu import _trainer
    tr = _trainer()
    opt = dict(optim_g=dict(type='AdamW', lr=1e-3, weight_decay=0.0, betas=[0.9, 0.999]), emb_norm_threshold=1e9)
    engine = TrainEngine(tr, opt, total_iter=100, mixed_precision='no')
    engine.base_lrs = [1e-3 for _ in engine.base_lrs]
    engine.scaler = torch.amp.GradScaler('cpu', init_scale=1024.0, enabled=True)     # fp16 training's loss scaling, on the CPU

    def fake_capture(fwd_bwd, warmup):
        for _ in range(warmup):
            fwd_bwd()
        out = fwd_bwd().clone()
        return _FakeGraph(fwd_bwd, out), out

    engine._capture = fake_capture
    engine.enable_graph(_rank_batch(rank), warmup=1)
    assert isinstance(engine._graph, _FakeGraph) and engine._finals_graph is not None
    p0 = torch.cat([p.detach().reshape(-1) for p in tr.trainable_parameters()]).clone()
    rec = dict(scales=[], params=[], losses=[], found=[])
    for step in range(4):
        b = _rank_batch(rank)
        if step == 1 and rank == 1:
            b['latents'] = b['latents'].clone()
            b['latents'][0, 0, 0, 0] = float('inf')          # ONE rank overflows in this step
        out = engine.step(b)
        rec['losses'].append(float(out['loss']))
        rec['scales'].append(float(engine.scaler.get_scale()))
        rec['params'].append(torch.cat([p.detach().reshape(-1) for p in tr.trainable_parameters()]).clone())
    torch.save(dict(p0=p0, replays=engine._graph.replays, steps=engine.global_step, **rec), os.path.join(out_dir, f'g_rank{rank}.pt'))
    dp.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_graph_replay_allreduce_and_found_inf_agree_across_ranks(tmp_path, emulated_hip):
    """TrainEngine._graph_step on two gloo ranks (the replay itself emulated: the captured closure re-run): static-input copies,
    replay, EAGER all-reduce of the bucket, GradScaler unscale / found_inf / update, AdamW. An overflow on ONE rank must make
    BOTH ranks skip that update (the bucket is reduced before the scaler looks at it), halve BOTH loss scales, and leave the
    ranks in lock-step afterwards (reference: accelerate's GradScaler under DDP, train_edlora.py:34,70,128)."""
    world, port = 2, 35500 + (os.getpid() % 2000)
    mp.spawn(_graph_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / 'g_rank0.pt'), torch.load(tmp_path / 'g_rank1.pt')
    assert r0['replays'] == r1['replays'] == 4 and r0['steps'] == r1['steps'] == 4
    assert r0['scales'] == r1['scales'] == [1024.0, 512.0, 512.0, 512.0]      # the overflow of rank 1 halves BOTH scales
    for k in range(4):
        torch.testing.assert_close(r0['params'][k], r1['params'][k], rtol=0, atol=0)     # lock-step, every step
    assert not torch.equal(r0['params'][0], r0['p0'])                          # step 0 updated
    assert torch.equal(r0['params'][1], r0['params'][0])                       # step 1 skipped on both ranks
    assert not torch.equal(r0['params'][2], r0['params'][1])                   # and training goes on
    assert all(torch.isfinite(p).all() for p in r0['params'])
    assert r1['losses'][1] != r1['losses'][1] or abs(r1['losses'][1]) == float('inf')   # rank 1 saw the overflow
    assert abs(r0['losses'][1]) < float('inf')

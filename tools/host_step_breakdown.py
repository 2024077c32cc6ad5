"""What the HOST spends per replayed training step. Inside the pipelined loop the host's time per step (30 ms) only mirrors the
device's (33 ms): the launch queue is finite and a host that runs ahead blocks inside hipGraphLaunch / the optimiser's launches.
With the device idle at every step start the host NEEDS 5.1 ms: graph.replay 3.9 + optimiser / GradScaler / freeze rule 1.1 + input
copies 0.1 -- a 6x margin before a slow host could pace the step (round 6, box of call r06c12).

    python tools/host_step_breakdown.py [steps]            (DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 to try the runtime's fast graph launch)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mos_path  # noqa: E402,F401
import torch  # noqa: E402

from bench import TRAIN_OPT, build_trainer, synthetic_batch  # noqa: E402
from mixofshow.pipelines.train_loop import TrainEngine  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device('cuda:0')
tr = build_trainer('sd15', dev)
tr.unet.train(); tr.text_encoder.train()
eng = TrainEngine(tr, dict(TRAIN_OPT, optim_g=dict(TRAIN_OPT['optim_g'])), total_iter=1e9, mixed_precision='fp16', channels_last=True)
batches = [synthetic_batch(4, 512, dev, i) for i in range(2)]
eng.enable_graph(batches[0])
for i in range(5):
    eng.step(batches[i % 2])
torch.cuda.synchronize()
acc = dict(replay=0.0, finish=0.0, total=0.0)
real_replay, real_finish = eng._graph.replay, eng._finish_step


def replay():
    t = time.perf_counter()
    real_replay()
    acc['replay'] += time.perf_counter() - t


def finish(loss):
    t = time.perf_counter()
    out = real_finish(loss)
    acc['finish'] += time.perf_counter() - t
    return out


class _G:                                  # the captured graph object, with a timed replay
    def __getattr__(self, k):
        return getattr(eng.__dict__['_graph_real'], k)

    def replay(self):
        return replay()


eng.__dict__['_graph_real'] = eng._graph
eng._graph = _G()
eng._finish_step = finish
t0 = time.perf_counter()
for i in range(steps):
    eng.step(batches[i % 2])
host = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
# the same with the device idle at the start of every step: with an empty queue nothing the host does can block on the device, so
# this is what the host NEEDS per step (the pipelined figure above includes back-pressure: the launch queue is finite, a host that
# runs ahead of the device waits inside hipGraphLaunch / the optimiser's launches)
pipelined = dict(acc)
acc.update(replay=0.0, finish=0.0)
need = 0.0
for i in range(10):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    eng.step(batches[i % 2])
    need += time.perf_counter() - t1
torch.cuda.synchronize()
print(f'device idle at every step start: host needs {need / 10 * 1e3:.2f} ms/step = graph.replay {acc["replay"] / 10 * 1e3:.2f} + optimiser / scaler / '
      f'freeze rule {acc["finish"] / 10 * 1e3:.2f} + rest {(need - acc["replay"] - acc["finish"]) / 10 * 1e3:.2f}')
acc = pipelined
print(f'pipelined loop, {steps} steps: wall {wall / steps * 1e3:.2f} ms/step, host loop time {host / steps * 1e3:.2f} ms/step (incl. waiting on the '
      f'full launch queue) = graph.replay {acc["replay"] / steps * 1e3:.2f} + optimiser / scaler / freeze rule '
      f'{acc["finish"] / steps * 1e3:.2f} + input copies and the rest {(host - acc["replay"] - acc["finish"]) / steps * 1e3:.2f}')

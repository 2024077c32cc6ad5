"""Autograd wrappers around the HIP primitives (mixofshow.hip.ops).

`lora_linear`   — y = x W^T + sum_g alpha_g (x A_g^T) B_g^T + b for 1..4 LoRA sites sharing x
                  (reference LoRALinearLayer.forward, mixofshow/models/edlora.py:244-246; W frozen).
`attention`     — fused softmax(scale q k^T) v, optionally exporting the probabilities of a few
                  key positions (what cal_attn_reg needs, trainer_edlora.py:289-298).
Both save only what the backward kernels need (no (B*H, N, 77) probability tensors).
"""
import os as _os

import torch

from . import ops

_default_compute_dtype = torch.float16
MAX_PACKED_RANK = ops.MOS_LORA_PAD      # LoRA ranks of the sites fused into one GEMM share a 16-wide MFMA operand


def set_default_compute_dtype(dtype):
    global _default_compute_dtype
    assert dtype in (torch.float16, torch.bfloat16)
    _default_compute_dtype = dtype


def compute_dtype_for(x):
    """Half dtype the kernels run in: the autocast dtype if autocast is on, else x's half dtype, else default."""
    if torch.is_autocast_enabled('cuda'):
        dt = torch.get_autocast_dtype('cuda')
        if dt in (torch.float16, torch.bfloat16):
            return dt
    if x.dtype in (torch.float16, torch.bfloat16):
        return x.dtype
    return _default_compute_dtype


class WeightCache:
    """Half-precision (and transposed) copies of frozen base weights, rebuilt when a weight changes."""

    def __init__(self):
        self._store = {}

    @staticmethod
    def _key(tensors):
        return tuple((t.data_ptr(), t._version, t.dtype, t.device) for t in tensors)

    def weight(self, name, weights, dtype, transposed=False):
        key = (name, dtype)
        ver = self._key(weights)
        ent = self._store.get(key)
        if ent is None or ent['ver'] != ver:
            w = [x.detach().reshape(x.shape[0], -1) for x in weights]  # 1x1 conv (N,K,1,1) -> (N,K)
            W = (w[0] if len(w) == 1 else torch.cat(w, 0)).to(dtype).contiguous()
            ent = {'ver': ver, 'W': W, 'Wt': None}
            self._store[key] = ent
        if transposed:
            if ent['Wt'] is None:
                ent['Wt'] = ent['W'].t().contiguous()
            return ent['W'], ent['Wt']
        return ent['W'], None

    def bias(self, name, biases):
        if all(b is None for b in biases):
            return None
        key = (name, 'bias')
        present = [b for b in biases if b is not None]
        ver = self._key(present)
        ent = self._store.get(key)
        if ent is None or ent['ver'] != ver:
            assert len(present) == len(biases), 'fused sites must all have a bias or none'
            b = torch.cat([x.detach().float() for x in biases], 0).contiguous()
            ent = {'ver': ver, 'b': b}
            self._store[key] = ent
        return ent['b']


# ---- LoRA operand packing: every group of the process in ONE launch per optimiser update -------------------------------
_direct_grad = False


def set_direct_grad_accumulation(flag):
    """True: the backward kernel accumulates the LoRA factor gradients straight into existing fp32 `.grad` tensors and
    autograd receives None for those inputs — no per-parameter slice / transpose / scale / add kernels. False
    (default): gradients are returned to autograd. TrainEngine (which owns the flat gradient bucket) switches it on only
    around its own forward/backward (`direct_grad_accumulation()` below), so that other users of autograd in the
    process (torch.autograd.grad, hooks, a second trainer without a bucket) keep the ordinary semantics."""
    global _direct_grad
    _direct_grad = bool(flag)


# MOS_LORA_GRAD_BATCH (default 1; host-side switch for same-box A/B runs): the token reductions of the LoRA factor gradients of a
# backward pass as one launch per padded-rank class (mos_lora_grad_all) instead of one per projection group.
import os as _os_lgb
_lora_grad_batch = _os_lgb.environ.get('MOS_LORA_GRAD_BATCH', '1') != '0'


class _DeferredFinals:
    """The ordered final sums of the LoRA factor gradients of one backward pass, batched into ONE launch
    (mos_lora_grad_final_all) instead of one per projection group (~90 per SD-1.5 training step). Workspaces are persistent
    per group, so the record table is identical from step to step: it is uploaded once (in place, fixed-capacity buffer) and
    a captured hipGraph replays the single launch with the table it was captured with."""

    MAX_WORKSPACES = 512     # an unfrozen store forgets its workspaces beyond this (new shapes / new models keep adding keys)

    def __init__(self):
        self.ws = {}
        self.pending = []
        self.used = set()
        self.table = None
        self.table_bytes = None
        self.capacity = 512
        self.frozen = False      # set once a captured hipGraph holds this store's table address and workspaces
        # round 5: the token reductions too (mos_lora_grad_all, one launch per padded-rank class instead of one per group)
        self.defer_reduction = _lora_grad_batch
        self.jobs = []           # (LoraGradJob, dtype, keep-alive tensors)
        self.job_table = None    # uint8 device table
        self.job_host = None     # pinned host image (the H2D copy of a captured scope is a graph node reading it at every replay)
        self.job_bytes = None    # image the device table holds (eager passes: uploaded only when it changes)
        self.job_stage = None    # eager passes: two pinned staging buffers, each with the event of the copy that last read it
        self.job_slot = 0

    def freeze(self):
        """A captured graph replays ONE mos_lora_grad_final_all launch with this table's address and the workspace pointers
        inside it: from now on any change of the table (another batch shape, another model stepping through this store)
        raises instead of silently redirecting the replays to stale partial sums."""
        self.frozen = True

    def begin_scope(self):
        self.pending = []
        self.jobs = []
        self.used = set()
        if not self.frozen and len(self.ws) > self.MAX_WORKSPACES:
            self.ws.clear()          # (keys hold raw addresses the allocator may recycle: bounded, never while frozen)

    def workspace(self, key, n_floats, device):
        t = self.ws.get(key)
        if t is None or t.numel() < n_floats or t.device != device:
            if self.frozen:
                raise RuntimeError('LoRA gradient workspaces are held by a captured hipGraph; a step of another shape must '
                                   'run through its own store (TrainEngine keeps one per graph and one for eager steps)')
            t = torch.empty(n_floats, dtype=torch.float32, device=device)
            self.ws[key] = t
        return t

    def _flush_jobs(self):
        """The deferred token reductions: jobs grouped by (dtype, padded rank), ONE table, one launch per class. The table holds
        the addresses of this pass's activations, so it is rewritten every eager pass; inside a hipGraph capture the upload is a
        copy node from the pinned host image (the addresses of the capture are those of every replay)."""
        import ctypes
        from . import lib as _lib
        jobs, self.jobs = self.jobs, []
        if not jobs:
            return
        jobs.sort(key=lambda e: (e[1] != torch.float16, e[0].nj))            # stable: program order inside a class
        size = ctypes.sizeof(_lib.LoraGradJob)
        assert len(jobs) <= self.capacity, 'too many deferred LoRA gradient groups'
        classes, i = [], 0
        while i < len(jobs):
            k = i
            begin, fl, by = 0, 0.0, 0.0
            while k < len(jobs) and jobs[k][1] == jobs[i][1] and jobs[k][0].nj == jobs[i][0].nj:
                jobs[k][0].block_begin = begin
                begin += jobs[k][0].n_blocks
                fl += jobs[k][0].flops
                by += jobs[k][0].bytes
                k += 1
            classes.append((i, k - i, begin, jobs[i][0].nj, jobs[i][1], fl, by))
            i = k
        raw = bytes((_lib.LoraGradJob * len(jobs))(*[e[0] for e in jobs]))
        dev = torch.device('cuda', torch.cuda.current_device())
        if self.job_table is None or self.job_table.device != dev:
            self.job_table = torch.zeros(self.capacity * size, dtype=torch.uint8, device=dev)
            self.job_host = torch.zeros(self.capacity * size, dtype=torch.uint8).pin_memory()
            self.job_bytes = None
        if torch.cuda.is_current_stream_capturing():
            self.job_host[:len(raw)].copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
            self.job_table[:len(raw)].copy_(self.job_host[:len(raw)], non_blocking=True)      # a memcpy node of the graph
            self.job_bytes = None
        elif raw != self.job_bytes:
            # Eager pass: no blocking copy out of pageable memory. Under the caching allocator the activations
            # of a step usually sit at the addresses of the step before, so the image is mostly unchanged and nothing is
            # uploaded at all; when it did change it goes through one of two pinned staging buffers (the other one may still
            # be feeding the previous pass's copy) with a non-blocking H2D on the launch stream.
            if self.job_stage is None:
                self.job_stage = [[torch.zeros(self.capacity * size, dtype=torch.uint8).pin_memory(), None] for _ in range(2)]
            self.job_slot ^= 1
            host, ev = self.job_stage[self.job_slot]
            if ev is not None:
                ev.synchronize()         # the copy that read this buffer two uploads ago: long finished, never a real wait
            host[:len(raw)].copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
            self.job_table[:len(raw)].copy_(host[:len(raw)], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self.job_stage[self.job_slot][1] = ev
            self.job_bytes = raw
        for first, n, blocks, nj, dtype, fl, by in classes:
            ops.lora_grad_all(self.job_table, first * size, n, blocks, nj, dtype, fl, by)
        del jobs                     # (dt / x / t / dy of every group were held until the launches above were enqueued)

    def flush(self):
        import ctypes
        from . import lib as _lib
        self._flush_jobs()
        recs, self.pending = self.pending, []
        if not recs:
            return
        assert len(recs) <= self.capacity, 'too many deferred LoRA gradient groups'
        begin = 0
        for r in recs:
            r.block_begin = begin
            begin += r.n_blocks
        raw = bytes((_lib.LoraFinalRec * len(recs))(*recs))
        dev = torch.device('cuda', torch.cuda.current_device())
        if self.table is None or self.table.device != dev:
            self.table = torch.zeros(self.capacity * ctypes.sizeof(_lib.LoraFinalRec), dtype=torch.uint8, device=dev)
            self.table_bytes = None
        if raw != self.table_bytes:
            if self.frozen:
                raise RuntimeError('LoRA gradient record table changed while a captured hipGraph replays it')
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('LoRA gradient record table changed during hipGraph capture (warm-up steps must precede it)')
            self.table[:len(raw)].copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
            self.table_bytes = raw
        ops.lora_grad_final_all(self.table, len(recs), begin)


_deferred_finals = None      # a _DeferredFinals while a direct_grad_accumulation(defer_finals=True) scope is open


def new_deferred_finals():
    """A private record table + workspace set for the deferred LoRA gradient sums (one per captured graph, one per engine
    for its eager steps)."""
    return _DeferredFinals()


class direct_grad_accumulation:
    """with direct_grad_accumulation(): ... — scoped form of set_direct_grad_accumulation(True). `defer_finals` (HIP device
    only): the per-group final sums of the LoRA factor gradients are collected during the backward pass and issued as one
    launch when the scope closes (the caller runs forward AND backward inside the scope)."""
    _store = None        # default store of scopes that bring none (never one a graph was captured with)

    def __init__(self, defer_finals=False, store=None):
        """`store`: the _DeferredFinals to collect into. Whoever captures a scope into a hipGraph must own the store of that
        scope (new_deferred_finals()), freeze it after capture and run every OTHER scope (eager steps of another batch shape,
        a second engine) through a different one: the graph bakes the store's table address and workspace pointers in."""
        self.defer = bool(defer_finals)
        self.store = store

    def __enter__(self):
        global _direct_grad, _deferred_finals
        self.prev, _direct_grad = _direct_grad, True
        self.prev_def = _deferred_finals
        if self.defer:
            st = self.store
            if st is None:
                if direct_grad_accumulation._store is None:
                    direct_grad_accumulation._store = _DeferredFinals()
                st = direct_grad_accumulation._store
            if st.frozen and not torch.cuda.is_current_stream_capturing() and self.store is None:
                raise RuntimeError('the default LoRA gradient store is frozen')      # (cannot happen: only owned stores freeze)
            _deferred_finals = st
            st.begin_scope()
        return self

    def __exit__(self, exc_type, *exc):
        global _direct_grad, _deferred_finals
        store = _deferred_finals if self.defer else None
        _direct_grad, _deferred_finals = self.prev, self.prev_def
        if store is not None:
            if exc_type is None:
                store.flush()
            else:
                store.pending = []
                store.jobs = []
        return False


class _PackGroup:
    __slots__ = ('downs', 'ups', 'alphas', 'K', 'bufs', 'vers', 'epoch', 'desc', 'elems')

    def params(self):
        """The fp32 master tensors (weak references: the registry must not keep a discarded model alive)."""
        ps = [r() for r in (*self.downs, *self.ups)]
        return None if any(p is None for p in ps) else (ps[:len(self.downs)], ps[len(self.downs):])


class LoraPackRegistry:
    """fp32 master LoRA factors -> packed half MFMA operands (A16, A16T, Bp16, BpT; include/mos_hip.h). Groups register
    on first use; whenever any group's parameters changed since the last pack (tensor version counters: load_state_dict,
    manual edits, foreach optimisers) or a backward pass formed gradients of LoRA factors (an optimiser step follows, and
    torch's FUSED AdamW updates parameters without touching their version counters: round 6) ALL groups are repacked by one
    `mos_lora_pack_all` launch — 1 launch per training step instead of one per projection call. Inside a captured hipGraph the launch sits where the first projection asked
    for its operands (`invalidate()` before capture guarantees it is there).

    The descriptor table lives in ONE device buffer of fixed capacity that is only ever updated IN PLACE (a captured graph
    bakes its address and the group count into the pack launch): groups registered later are appended behind the ones
    the graph knows, pointer / alpha changes rewrite entries where they are. Outgrowing the buffer while a graph holds
    its address (`freeze()`, set by TrainEngine.enable_graph) raises instead of reallocating. Parameters are held
    weakly; groups whose model is gone are dropped at the next pack (never while frozen: the captured group count
    must stay valid — a dead group's entry then only makes the kernel rewrite operand buffers nobody reads)."""

    def __init__(self, device, dtype, capacity=512):
        self.device, self.dtype = device, dtype
        self.groups = {}
        self.epoch = 0
        self.capacity = capacity
        self.desc_dev = None
        self.table_dirty = True
        self.max_elems = 1
        self.frozen = 0

    def freeze(self):
        self.frozen += 1

    def unfreeze(self):
        self.frozen = max(0, self.frozen - 1)

    @staticmethod
    def _vers(g, ps):
        downs, ups = ps
        return tuple(p._version for p in downs) + tuple(p._version for p in ups) + tuple(
            p.data_ptr() for p in downs) + tuple(p.data_ptr() for p in ups) + tuple(g.alphas)

    def invalidate(self):
        self.epoch += 1

    def clear(self):
        """Forget every group (tests / explicit model teardown). Not while a captured graph uses the table."""
        if self.frozen:
            raise RuntimeError('LoraPackRegistry.clear(): a captured hipGraph still packs through this registry')
        self.groups.clear()
        self.table_dirty = True

    def get(self, downs, ups, alphas, K):
        import weakref
        key = tuple(id(p) for p in downs) + tuple(id(p) for p in ups)
        g = self.groups.get(key)
        if g is not None and g.params() is None:      # id() of a dead tensor was recycled by a new one
            g = None
        if g is None:
            g = _PackGroup()
            g.downs, g.ups = tuple(weakref.ref(p) for p in downs), tuple(weakref.ref(p) for p in ups)
            g.alphas, g.K = tuple(alphas), K
            N = sum(u.shape[0] for u in ups)
            dev, dt = self.device, self.dtype
            pad = ops.MOS_LORA_PAD
            g.bufs = (torch.empty((pad, K), dtype=dt, device=dev), torch.empty((K, pad), dtype=dt, device=dev),
                      torch.empty((N, pad), dtype=dt, device=dev), torch.empty((pad, N), dtype=dt, device=dev))
            g.elems = pad * max(K, N)
            g.vers, g.epoch = None, -1
            g.desc = None
            self.groups.pop(key, None)               # (a recycled id: the stale entry goes, the new one is appended)
            self.groups[key] = g
            self.table_dirty = True
        else:
            g.alphas = tuple(alphas)
        if g.vers != self._vers(g, (downs, ups)):
            self.epoch += 1                      # something changed since the last pack: everything is repacked once
        if g.epoch != self.epoch:
            self._pack_all()
        return g.bufs

    def _pack_all(self):
        import ctypes
        from . import lib as _lib
        if not self.frozen:                      # drop the groups of models that no longer exist
            dead = [k for k, g in self.groups.items() if g.params() is None]
            for k in dead:
                del self.groups[k]
            self.table_dirty = self.table_dirty or bool(dead)
        gs = list(self.groups.values())
        if len(gs) > self.capacity:
            if self.frozen:
                raise RuntimeError(f'LoraPackRegistry: {len(gs)} LoRA groups exceed the descriptor table ({self.capacity}) '
                                   'that a captured hipGraph holds; build every model before TrainEngine.enable_graph')
            self.capacity, self.desc_dev = max(2 * self.capacity, len(gs)), None
        stale = self.table_dirty or self.desc_dev is None
        for g in gs:
            ps = g.params()
            if ps is None:                       # frozen registry, model gone: keep the last descriptor
                g.epoch = self.epoch
                continue
            v = self._vers(g, ps)
            nv = len(ps[0]) + len(ps[1])
            if g.desc is None or g.vers is None or v[nv:] != g.vers[nv:]:
                g.desc = ops.lora_group_desc(ps[0], ps[1], g.alphas, g.K, *g.bufs)   # pointers / alphas changed
                stale = True
            g.vers = v
            g.epoch = self.epoch
        if stale:
            arr = (_lib.LoraGroup * len(gs))(*[g.desc for g in gs])
            raw = bytes(arr)
            if self.desc_dev is None:
                self.desc_dev = torch.zeros(self.capacity * ctypes.sizeof(_lib.LoraGroup), dtype=torch.uint8, device=self.device)
            host = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
            self.desc_dev[:len(raw)].copy_(host)                 # in place: the buffer's address never changes
            self.max_elems = max([self.max_elems] + [g.elems for g in gs])
            self.table_dirty = False
        ops.lora_pack_all(self.desc_dev, len(gs), self.max_elems, self.dtype)


_registries = {}


def lora_registry(device, dtype):
    key = (device.type, device.index, dtype)
    r = _registries.get(key)
    if r is None:
        r = LoraPackRegistry(device, dtype)
        _registries[key] = r
    return r


def invalidate_lora_packs():
    """Force a repack at the next use (call before capturing a step into a hipGraph)."""
    for r in _registries.values():
        r.invalidate()


def freeze_lora_packs(flag=True):
    """A captured graph holds the descriptor table's address and group count: pin both (see LoraPackRegistry)."""
    for r in _registries.values():
        r.freeze() if flag else r.unfreeze()


def _lora_operands(downs, ups, alphas, K, dtype, device):
    ok = device.type == 'cuda' and all(p.dtype == torch.float32 and p.is_contiguous() for p in (*downs, *ups))
    if ok:
        return lora_registry(device, dtype).get(downs, ups, alphas, K)
    return ops.lora_pack(downs, ups, alphas, K, dtype, device)     # per-call packing (host tests / odd layouts)


class _LoRALinear(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, W16, Wt16, bias32, alphas, residual, *params):
        cd = W16.dtype
        K = W16.shape[1]
        N = W16.shape[0]
        x2 = x.reshape(-1, K)
        if x2.dtype != cd:
            x2 = x2.to(cd)
        if x2.stride(1) != 1 or x2.stride(0) % 8 != 0:
            x2 = x2.contiguous()
        r2 = None
        if residual is not None:         # y = round(GEMM) + residual in the GEMM's epilogue (== the GEMM followed by a half add)
            r2 = residual.reshape(-1, N)
            if r2.dtype != cd:
                r2 = r2.to(cd)
            if r2.stride(1) != 1 or r2.stride(0) % 8 != 0:
                r2 = r2.contiguous()
        n_sites = len(params) // 2
        need_grad = any(ctx.needs_input_grad)
        if n_sites:
            downs, ups = params[0::2], params[1::2]
            A16, A16T, Bp16, BpT = _lora_operands(downs, ups, alphas, K, cd, x2.device)
            if r2 is None:
                y, t = ops.linear_fused_fwd(x2, W16, A16, Bp16, bias32, need_t=need_grad)
            else:
                y, t = ops.linear_fwd_ex(x2, W16, A16, Bp16, bias32, residual=r2, need_t=need_grad)
            ctx.save_for_backward(x2, t, A16T, BpT, Wt16)
            ctx.params = params if need_grad else None
        else:
            if r2 is None:
                y = ops.linear_fwd(x2, W16, None, None, bias32)
            else:
                y = ops.linear_fwd_ex(x2, W16, None, None, bias32, residual=r2)[0]
            ctx.save_for_backward(x2, None, None, None, Wt16)
            ctx.params = None
        ctx.alphas = alphas
        ctx.n_sites = n_sites
        ctx.rank = params[0].shape[0] if n_sites else 0
        ctx.x_shape, ctx.x_dtype = x.shape, x.dtype
        ctx.r_shape, ctx.r_dtype = (None, None) if residual is None else (residual.shape, residual.dtype)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, t, A16T, BpT, Wt16 = ctx.saved_tensors
        N = dy.shape[-1]
        dy2 = dy.reshape(-1, N)
        if dy2.dtype != x2.dtype:
            dy2 = dy2.to(x2.dtype)
        if dy2.stride(1) != 1 or dy2.stride(0) % 8 != 0:
            dy2 = dy2.contiguous()
        need_dx = ctx.needs_input_grad[0]
        if need_dx and Wt16 is None:
            raise RuntimeError('mixofshow.hip: backward to the input needs the transposed weight (Wt16)')
        dres = None
        if ctx.r_shape is not None and ctx.needs_input_grad[5]:      # the residual's gradient IS dy (no kernel)
            dres = dy.reshape(ctx.r_shape)
            if dres.dtype != ctx.r_dtype:
                dres = dres.to(ctx.r_dtype)
        grads = [None] * (2 * ctx.n_sites)
        if ctx.n_sites == 0:
            dx = ops.linear_bwd(dy2, x2, Wt16, None, None, None, need_dx=need_dx, need_lora=False)[0]
        else:
            targets = None
            if any(ctx.needs_input_grad[6:]):
                # gradients of the LoRA factors are being formed: an optimiser is about to change the masters, and not every
                # optimiser tells (torch.optim.AdamW(fused=True) leaves `_version` alone) -- the packed operands are repacked at
                # their next use (one launch for all groups). Python-side counter only: nothing is launched here.
                reg = _registries.get((dy2.device.type, dy2.device.index, x2.dtype))
                if reg is not None:
                    reg.invalidate()
                targets = []
                for g in range(ctx.n_sites):
                    pair, accs = [], []
                    for q in (2 * g, 2 * g + 1):
                        p = ctx.params[q]
                        if not ctx.needs_input_grad[6 + q]:
                            pair.append(None)
                            accs.append(False)
                        elif (_direct_grad and p.grad is not None and p.grad.dtype == torch.float32
                              and p.grad.is_contiguous() and p.grad.shape == p.shape):
                            pair.append(p.grad)          # accumulated in place by the kernel; autograd gets None
                            accs.append(True)
                        else:
                            gt = torch.empty(p.shape, dtype=torch.float32, device=dy2.device)
                            grads[q] = gt
                            pair.append(gt)
                            accs.append(False)
                    targets.append((pair[0], pair[1], ctx.alphas[g], ctx.params[2 * g + 1].shape[0], accs[0], accs[1]))
            st = _deferred_finals
            wkey = (BpT.data_ptr(), dy2.shape[0], dy2.shape[1], x2.shape[1])
            if st is not None and targets is not None and all(a and b for (_, _, _, _, a, b) in targets) \
                    and all(d is not None and u is not None for (d, u, *_r) in targets) and wkey not in st.used:
                st.used.add(wkey)        # (a layer called twice inside one scope: its second call sums immediately)
                # every gradient of this group is accumulated in place into an existing `.grad`: its final sum can wait
                dx, rec = ops.linear_fused_bwd(dy2, x2, Wt16, t, A16T, BpT, targets, ctx.rank, need_dx=need_dx,
                                               defer=lambda key, n: st.workspace(key, n, dy2.device),
                                               defer_reduction=st.defer_reduction)
                if isinstance(rec, tuple):           # (final-sum record, token-reduction job, tensors the job points to)
                    st.pending.append(rec[0])
                    st.jobs.append((rec[1], dy2.dtype, rec[2]))
                elif rec is not None:
                    st.pending.append(rec)
            else:
                dx = ops.linear_fused_bwd(dy2, x2, Wt16, t, A16T, BpT, targets, ctx.rank, need_dx=need_dx)
        if dx is not None:
            dx = dx.view(ctx.x_shape)
            if dx.dtype != ctx.x_dtype:
                dx = dx.to(ctx.x_dtype)
        return (dx, None, None, None, None, dres, *grads)


def lora_linear(x, W16, Wt16, bias32, sites, residual=None):
    """sites: list of (lora_down.weight (r,K[,1,1]), lora_up.weight (n,r[,1,1]), alpha float).
    residual (shape of the result): added in the GEMM's epilogue (same rounding points as GEMM + add kernel)."""
    params, alphas = [], []
    for down, up, alpha in sites:
        params += [down, up]
        alphas.append(float(alpha))
    ranks = sum(down.shape[0] for down, _, _ in sites)
    if ranks > MAX_PACKED_RANK:
        raise ValueError(f'LoRA rank {ranks} (summed over {len(sites)} fused site(s)) is not supported by the fused HIP path: the '
                         f'rank dimension is one {MAX_PACKED_RANK}-wide MFMA operand (rank <= {MAX_PACKED_RANK}); see INTEGRATION.md')
    return _LoRALinear.apply(x, W16, Wt16, bias32, tuple(alphas), residual, *params)


# ---- feed-forward of the transformer block (diffusers FeedForward: GEGLU projection, Linear) on the library's GEMM ---------
# Switches, defaults from the same-box measurements of round 4 (profiles/r04_kernel_bench_ff_gn_conv.txt,
# profiles/r04_ab_same_box_{train,regional}_switches.txt):
#   (FF1 with a GEGLU epilogue in the library's GEMM was an option of round 4: equal to hipBLASLt + the geglu kernel at level 0,
#    slower at the wide levels; removed in round 5.)
#   MOS_FF2_OWN (default 1): FF2 as the library's GEMM with the residual add in its epilogue where it wins: K = 4C <= 1280
#     (level 0: 19.4 vs 24.4 us sampling, 24.8 vs 25.7 training) and K <= 2560 up to 3072 rows (22.9 vs 25.7 us); the wide
#     levels stay on hipBLASLt + add (33.5 vs 27.3 us at 768 x 5120 -> 1280).
#   MOS_GEMM_RESIDUAL (default 1): the 1x1 proj_out's residual in the GEMM epilogue (bit-identical to GEMM + add).
_ff2_own = _os.environ.get('MOS_FF2_OWN', '1') != '0'
_gemm_residual = _os.environ.get('MOS_GEMM_RESIDUAL', '1') != '0'


def _plain_linear(linear):
    """An untouched, frozen nn.Linear: no LoRA branch (its forward is replaced), no hooks (gradient fusion records through
    nn.Module.__call__), nothing trainable."""
    return (type(linear) is torch.nn.Linear and getattr(linear, '_mos_lora', None) is None
            and linear.forward.__func__ is torch.nn.Linear.forward and not linear._forward_hooks
            and not linear._forward_pre_hooks and _frozen(linear.weight, linear.bias))


def _half_path(x):
    half = x.dtype in (torch.float16, torch.bfloat16)
    ac = torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') in (torch.float16, torch.bfloat16)
    if not (x.is_cuda and (half or ac)):
        return None
    return x.dtype if half else torch.get_autocast_dtype('cuda')


def linear_residual(linear, x, residual):
    """`linear(x) + residual` (transformer block: ff.net[2](h) + hidden_states) as ONE GEMM with the add in its epilogue, on the
    HIP path (frozen plain Linear, half activations, K and N multiples of 8); anything else: the two plain ops."""
    dt = _half_path(x) if (_ff2_own and _plain_linear(linear)) else None
    K, rows = linear.in_features, x.numel() // max(1, x.shape[-1])
    wins = K <= 1280 or (K <= 2560 and rows <= 3072)         # measured crossover against hipBLASLt + add, see above
    if (dt is None or not wins or K % 8 or linear.out_features % 8 or residual.shape[:-1] != x.shape[:-1]
            or residual.dtype != dt):        # (a residual stream of another dtype keeps `linear(x) + residual`'s promotion)
        return linear(x) + residual
    cache = linear.__dict__.get('_mos_cache')
    if cache is None:
        cache = WeightCache()
        object.__setattr__(linear, '_mos_cache', cache)
    need_bwd = torch.is_grad_enabled() and x.requires_grad
    W16, Wt16 = cache.weight('w', [linear.weight], dt, transposed=need_bwd)
    b32 = cache.bias('w', [linear.bias])
    return lora_linear(x if x.dtype == dt else x.to(dt), W16, Wt16, b32, [], residual=residual)


def linear_geglu(proj, x):
    """`geglu(proj(x))` (diffusers GEGLU): the projection GEMM (hipBLASLt through torch: measured faster than the library's GEMM
    at these widths) + ONE fused kernel each way for value * gelu(gate) (the backward needs the pre-activation)."""
    return geglu(proj(x))


class _Attention(torch.autograd.Function):
    """mode 'qkv': a = (B,N,3C) fused projection; mode 'q_kv': a = q (B,N,C), b = (B,M,2C); mode 'sep': a,b,c."""

    @staticmethod
    def forward(ctx, mode, heads, scale, tok_idx, a, b, c, causal=False):
        if mode == 'qkv':
            C = a.shape[-1] // 3
            q, k, v = a[..., :C], a[..., C:2 * C], a[..., 2 * C:]
        elif mode == 'q_kv':
            C = a.shape[-1]
            q, k, v = a, b[..., :C], b[..., C:]
        else:
            q, k, v = a, b, c
        need_grad = any(ctx.needs_input_grad[4:7])
        o, lse, pcols = ops.attn_fwd(q, k, v, heads, scale, tok_idx=tok_idx, need_lse=need_grad, causal=causal)
        ctx.mode, ctx.heads, ctx.scale, ctx.causal = mode, heads, scale, causal
        ctx.has_pcols = pcols is not None
        # (no zero-filled stand-ins: without tok_idx the second output is None, and a probability-column output nobody
        # consumed arrives in backward as None instead of a materialised zero tensor -- 2 fills per layer and step)
        ctx.set_materialize_grads(False)
        if need_grad:
            ctx.save_for_backward(a, b, c, o, lse, tok_idx, pcols)
        return o, pcols

    @staticmethod
    def backward(ctx, dO, dpcols=None):
        a, b, c, o, lse, tok_idx, pcols = ctx.saved_tensors
        mode = ctx.mode
        if dO is None:
            dO = torch.zeros_like(o)
        if dO.stride(-1) != 1 or dO.stride(1) % 8 != 0 or dO.dtype != o.dtype:
            dO = dO.to(o.dtype).contiguous()
        da = torch.empty_like(a)
        db = torch.empty_like(b) if b is not None else None
        dc = torch.empty_like(c) if c is not None else None
        if mode == 'qkv':
            C = a.shape[-1] // 3
            q, k, v = a[..., :C], a[..., C:2 * C], a[..., 2 * C:]
            dq, dk, dv = da[..., :C], da[..., C:2 * C], da[..., 2 * C:]
        elif mode == 'q_kv':
            C = a.shape[-1]
            q, k, v = a, b[..., :C], b[..., C:]
            dq, dk, dv = da, db[..., :C], db[..., C:]
        else:
            q, k, v = a, b, c
            dq, dk, dv = da, db, dc
        if ctx.has_pcols and dpcols is not None:
            dpc = dpcols.float().contiguous()
        else:
            dpc = None
        ops.attn_bwd(q, k, v, o, lse, dO, ctx.heads, ctx.scale, dq, dk, dv, tok_idx=tok_idx if dpc is not None else None,
                     pcols=pcols if dpc is not None else None, dpcols=dpc, causal=ctx.causal)
        return None, None, None, None, da, db, dc, None


def _check_half(*ts):
    for t in ts:
        if t is not None and t.dtype not in (torch.float16, torch.bfloat16):
            raise TypeError(f'mixofshow.hip.attention needs float16/bfloat16 activations, got {t.dtype}')


def attention_qkv(qkv, heads, scale, causal=False):
    _check_half(qkv)
    return _Attention.apply('qkv', heads, scale, None, qkv, None, None, causal)[0]


def attention_q_kv(q, kv, heads, scale, tok_idx=None):
    """Returns (o, pcols or None)."""
    _check_half(q, kv)
    o, pcols = _Attention.apply('q_kv', heads, scale, tok_idx, q, kv, None)
    return o, (pcols if tok_idx is not None else None)


def attention(q, k, v, heads, scale, tok_idx=None, causal=False):
    _check_half(q, k, v)
    o, pcols = _Attention.apply('sep', heads, scale, tok_idx, q, k, v, causal)
    return o, (pcols if tok_idx is not None else None)


class _AttnProbs(torch.autograd.Function):
    """softmax(scale q k^T) as a dense (B*H, Nq, Nkv) tensor WITH autograd (mos_attn_probs / mos_attn_probs_bwd): the
    reference's `attn.get_attention_scores(query, key)` (edlora.py:81) for controllers that take the full map in training."""

    @staticmethod
    def forward(ctx, q, k, heads, scale):
        probs = ops.attn_probs(q, k, heads, scale)
        ctx.save_for_backward(q, k, probs)
        ctx.heads, ctx.scale = heads, scale
        return probs

    @staticmethod
    def backward(ctx, dprobs):
        q, k, probs = ctx.saved_tensors           # (an in-place edit of the map by the controller trips autograd's version check)
        dq, dk = ops.attn_probs_bwd(q, k, probs, dprobs, ctx.heads, ctx.scale)
        return dq, dk, None, None


class _AttnPV(torch.autograd.Function):
    """torch.bmm(attention_probs, value) + batch_to_head_dim (edlora.py:83-85) with autograd (mos_attn_pv / mos_attn_pv_bwd)."""

    @staticmethod
    def forward(ctx, probs, v, heads):
        o = ops.attn_pv(probs, v, heads)
        ctx.save_for_backward(probs, v)
        ctx.heads = heads
        return o

    @staticmethod
    def backward(ctx, dO):
        probs, v = ctx.saved_tensors
        if dO.dtype != v.dtype:
            dO = dO.to(v.dtype)
        dprobs, dv = ops.attn_pv_bwd(probs, v, dO, ctx.heads)
        return dprobs, dv, None


def attn_probs(q, k, heads, scale):
    if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad):
        return _AttnProbs.apply(q, k, heads, scale)
    return ops.attn_probs(q, k, heads, scale)


def attn_pv(probs, v, heads):
    if torch.is_grad_enabled() and (probs.requires_grad or v.requires_grad):
        return _AttnPV.apply(probs, v, heads)
    return ops.attn_pv(probs, v, heads)


def _dense_format(x):
    """x as is if it is NCHW-contiguous or channels_last-contiguous (the kernels take both), else an NCHW copy."""
    if x.is_contiguous() or (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)):
        return x
    return x.contiguous()


class _GroupNormSiLU(torch.autograd.Function):
    """Fused GroupNorm (+ SiLU) on half NCHW or channels_last activations; affine parameters are frozen constants."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, silu, chan_part=None):
        xc = _dense_format(x)
        y, stats = ops.groupnorm_silu_fwd(xc, gamma, beta, groups, eps, silu, chan_part=chan_part if xc is x else None)
        ctx.save_for_backward(xc, gamma, beta, stats)
        ctx.groups, ctx.silu = groups, silu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, stats = ctx.saved_tensors
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        if not ops._same_layout(dy, x):
            dy = dy.contiguous(memory_format=torch.channels_last) if ops._is_nhwc(x) else dy.contiguous()
        return ops.groupnorm_silu_bwd(dy, x, gamma, beta, stats, ctx.groups, ctx.silu), None, None, None, None, None, None


class _GroupNormSiLUTap(torch.autograd.Function):
    """(x, y) = (x passed through, GroupNorm(+SiLU)(x)): callers route the skip path that bypasses the norm through the first
    output, so its gradient meets the norm's input gradient inside the backward kernel (mos_groupnorm_silu_bwd_nhwc_res)
    instead of in a separate autograd accumulation launch."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, silu, chan_part=None):
        xc = _dense_format(x)
        y, stats = ops.groupnorm_silu_fwd(xc, gamma, beta, groups, eps, silu, chan_part=chan_part if xc is x else None)
        ctx.save_for_backward(xc, gamma, beta, stats)
        ctx.groups, ctx.silu = groups, silu
        ctx.set_materialize_grads(False)
        return x, y

    @staticmethod
    def backward(ctx, ds, dy):
        x, gamma, beta, stats = ctx.saved_tensors
        if dy is None:
            return ds, None, None, None, None, None, None
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        if not ops._same_layout(dy, x):
            dy = dy.contiguous(memory_format=torch.channels_last) if ops._is_nhwc(x) else dy.contiguous()
        if ds is None:
            return ops.groupnorm_silu_bwd(dy, x, gamma, beta, stats, ctx.groups, ctx.silu), None, None, None, None, None, None
        if ops._is_nhwc(x) and ds.dtype == x.dtype and ds.shape == x.shape:
            if not _grads_in_place or ops.nhwc_pixel_stride(ds) is None:   # a channel slice of a concatenation's gradient is
                ds = ds.contiguous(memory_format=torch.channels_last)      # read in place (its pixel stride goes to the kernel)
            return (ops.groupnorm_silu_bwd(dy, x, gamma, beta, stats, ctx.groups, ctx.silu, ds=ds), None, None, None, None,
                    None, None)
        return ops.groupnorm_silu_bwd(dy, x, gamma, beta, stats, ctx.groups, ctx.silu) + ds, None, None, None, None, None, None


def _frozen(*params):
    """True when no gradient will be asked for these parameters: they are frozen, or autograd is off (sampling pipelines run
    under no_grad with ordinary requires_grad=True modules)."""
    return not (torch.is_grad_enabled() and any(p is not None and p.requires_grad for p in params))


def _affine32(norm):
    """fp32 copies of a norm's affine parameters (the kernels take fp32 gamma / beta; fp16 pipelines store them in half)."""
    w, b = norm.weight, norm.bias
    if w.dtype == torch.float32 and b.dtype == torch.float32:
        return w, b
    key = (w.data_ptr(), w._version, b.data_ptr(), b._version, w.dtype)
    ent = norm.__dict__.get('_mos_affine32')
    if ent is None or ent[0] != key:
        ent = (key, w.detach().float().contiguous(), b.detach().float().contiguous())
        object.__setattr__(norm, '_mos_affine32', ent)
    return ent[1], ent[2]


# Round 6: GroupNorm statistics from the producing convolution's epilogue. conv3x3(..., gn_groups=G) leaves the per-(tile,
# channel) sums of its output ON the output tensor (a Python attribute, with the tensor's version counter at that moment);
# group_norm_act picks them up if the tensor is still that object and has not been written to since -- anything else (a new
# tensor from an add or a concat, an in-place update) simply finds no statistics and takes the normal path.
# MOS_GN_FROM_CONV (host-side A/B switch, read once): '0' never; '1' only where the norm would read the map twice (large maps: the
# three-launch slice form); default '2': on every map whose convolution can leave the statistics -- with them the norm is ONE
# full-width launch (gn_pre_apply_kernel) instead of the 32-64-workgroup column kernel of the middle levels. Same box,
# profiles/r06c10_*: GroupNorm kernels 58.9 -> 52.4 ('1') -> 48.6 ms per regional sample, training step 33.63 -> 33.46 -> 33.38 ms.
_gn_from_conv = _os.environ.get('MOS_GN_FROM_CONV', '2') != '0'
_gn_from_conv_always = _os.environ.get('MOS_GN_FROM_CONV', '2') == '2'


def _attach_gn_stats(y, part):
    if part is not None:
        y._mos_gn_part = (part, y._version, y.data_ptr())


def _producer_gn_stats(x):
    ent = getattr(x, '_mos_gn_part', None)
    if ent is None or not _gn_from_conv:
        return None
    part, version, ptr = ent
    if version != x._version or ptr != x.data_ptr() or part.shape[0] != x.shape[0] or part.shape[2] != x.shape[1]:
        return None
    return part


def group_norm_act(norm, x, silu, tap=False):
    """`silu(norm(x))` (or `norm(x)`) for an nn.GroupNorm `norm`. tap=True returns (x, y): use the returned x for the skip
    path around the norm (residual / shortcut) -- on the HIP path that routes the skip's gradient into the norm's backward
    kernel (_GroupNormSiLUTap); elsewhere it is x itself.

    HIP path: half-precision device tensors with frozen affine parameters and HW % 8 == 0 — one fused kernel pair
    instead of autocast's cast / fp32 group_norm / fp32 silu / cast chain. Anything else (CPU oracle runs, fp32
    inference, trainable norms) takes the plain torch ops: this is plumbing around the hot path, not part of it."""
    hw = x.numel() // max(1, x.shape[0] * x.shape[1])
    nhwc = x.dim() == 4 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)
    use_hip = (x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and (hw % 8 == 0 or (nhwc and x.shape[1] % 8 == 0))
               and norm.weight is not None and norm.bias is not None and _frozen(norm.weight, norm.bias))
    if use_hip:
        gamma, beta = _affine32(norm)
        part = _producer_gn_stats(x) if nhwc else None
        if tap and _fuse_gn_res and torch.is_grad_enabled() and x.requires_grad:
            return _GroupNormSiLUTap.apply(x, gamma, beta, norm.num_groups, norm.eps, bool(silu), part)
        y = _GroupNormSiLU.apply(x, gamma, beta, norm.num_groups, norm.eps, bool(silu), part)
        return (x, y) if tap else y
    y = norm(x)
    y = torch.nn.functional.silu(y) if silu else y
    return (x, y) if tap else y


class _LayerNorm(torch.autograd.Function):
    """LayerNorm over the last dim, half in / half out, fp32 statistics; affine parameters are frozen constants."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        need = ctx.needs_input_grad[0]
        y, stats = ops.layernorm_fwd(x2, gamma, beta, eps, need_stats=need)
        if need:
            ctx.save_for_backward(x2, gamma, stats)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, gamma, stats = ctx.saved_tensors
        d2 = dy.reshape(x2.shape)
        if d2.dtype != x2.dtype or not d2.is_contiguous():
            d2 = d2.to(x2.dtype).contiguous()
        return ops.layernorm_bwd(d2, x2, gamma, stats).view(dy.shape), None, None, None


def layer_norm(norm, x):
    """`norm(x)` for an nn.LayerNorm over the last dim. HIP path: device tensors, half activations (or an fp32 stream under
    half autocast), frozen fp32 affine parameters — one kernel instead of autocast's cast / fp32 layer_norm / cast chain.
    Anything else takes the plain torch op (CPU oracle runs, fp32 inference, trainable norms): plumbing, not hot path."""
    C = x.shape[-1]
    half = x.dtype in (torch.float16, torch.bfloat16)
    ac = torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') in (torch.float16, torch.bfloat16)
    use_hip = (x.is_cuda and (half or ac) and len(norm.normalized_shape) == 1 and norm.weight is not None
               and norm.bias is not None and _frozen(norm.weight, norm.bias) and C % 8 == 0 and C <= 2048)
    if not use_hip:
        return norm(x)
    if not half:
        x = x.to(torch.get_autocast_dtype('cuda'))
    gamma, beta = _affine32(norm)
    return _LayerNorm.apply(x, gamma, beta, norm.eps)


class _AddLayerNorm(torch.autograd.Function):
    """(s, y) = (x + r, LN(x + r)) in one kernel; r None: s is x itself, passed through so that the gradient which bypasses
    the norm (the residual connection) meets the norm's input gradient inside the backward kernel instead of in a
    separate autograd accumulation launch."""

    @staticmethod
    def forward(ctx, x, r, gamma, beta, eps, half_dtype):
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        r2 = None
        if r is not None:
            r2 = r.reshape(-1, C)
            if not r2.is_contiguous():
                r2 = r2.contiguous()
        need = ctx.needs_input_grad[0] or (r is not None and ctx.needs_input_grad[1])
        s2, y, stats = ops.add_layernorm_fwd(x2, r2, gamma, beta, eps, need_stats=need, half_dtype=half_dtype)
        if need:
            ctx.save_for_backward(s2, gamma, stats)
        ctx.has_r, ctx.half_dtype = r is not None, y.dtype
        ctx.set_materialize_grads(False)
        return (s2.view(x.shape) if r is not None else x), y.view(x.shape)

    @staticmethod
    def backward(ctx, ds, dy):
        s2, gamma, stats = ctx.saved_tensors
        stream32 = s2.dtype == torch.float32
        want_r = ctx.has_r and ctx.needs_input_grad[1]
        if dy is None:                                   # the normalised output was not used: only the bypass gradient
            dr = None
            if ds is not None and want_r:
                dr = ds.to(ctx.half_dtype) if stream32 else ds
            return ds, dr, None, None, None, None
        d2 = dy.reshape(s2.shape)
        if d2.dtype != ctx.half_dtype or not d2.is_contiguous():
            d2 = d2.to(ctx.half_dtype).contiguous()
        ds2 = None
        if ds is not None:
            ds2 = ds.reshape(s2.shape)
            if ds2.dtype != s2.dtype or not ds2.is_contiguous():
                ds2 = ds2.to(s2.dtype).contiguous()
        dx, dxh = ops.add_layernorm_bwd(d2, ds2, s2, gamma, stats, half_copy=want_r and stream32)
        dx = dx.view(ds.shape if ds is not None else dy.shape)
        dr = None
        if want_r:
            dr = dxh.view(dx.shape) if stream32 else dx
        return dx, dr, None, None, None, None


# A/B switches (tests, bench): '0' routes through the separate add + norm kernels of round 2
# Round 6: a layer whose output went into a torch.cat along the channels (the UNet's skip concatenations) receives a CHANNEL SLICE of
# the concatenation's gradient. The kernels that consume it -- the tapped GroupNorm's bypass gradient, the dX 3x3 convolution -- take
# the slice's pixel stride and read it where it is (`ops.nhwc_pixel_stride`); MOS_GRADS_IN_PLACE=0 restores the contiguous copies
# (14 strided clones per SD-1.5 training step) for a same-box A/B. Read once.
_grads_in_place = _os.environ.get('MOS_GRADS_IN_PLACE', '1') != '0'
_fuse_add_ln = _os.environ.get('MOS_FUSE_ADD_LN', '1') != '0'
_fuse_gn_res = _os.environ.get('MOS_FUSE_GN_RES', '1') != '0'


def add_layer_norm(norm, x, r=None):
    """(s, y) with s = x + r (s is x when r is None) and y = norm(s): the residual add of a transformer block fused into the
    LayerNorm that consumes the sum, forward and backward (ops.add_layernorm_fwd / _bwd). Always use the returned `s` as the
    residual stream downstream -- that is what routes its bypass gradient into the norm's backward kernel.

    HIP path: like layer_norm, plus r in the half dtype and x either the same half dtype (UNet blocks) or fp32 under half
    autocast (the CLIP tower's residual stream). Anything else: the two plain ops."""
    C = x.shape[-1]
    half = x.dtype in (torch.float16, torch.bfloat16)
    ac = torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') in (torch.float16, torch.bfloat16)
    hd = x.dtype if half else (torch.get_autocast_dtype('cuda') if ac else None)
    use_hip = (_fuse_add_ln and x.is_cuda and hd is not None and (half or x.dtype == torch.float32)
               and (r is None or (r.dtype == hd and r.shape == x.shape))
               and len(norm.normalized_shape) == 1 and norm.weight is not None and norm.bias is not None
               and _frozen(norm.weight, norm.bias) and C % 8 == 0 and C <= 2048)
    if not use_hip:
        s = x if r is None else x + r
        return s, layer_norm(norm, s)
    gamma, beta = _affine32(norm)
    return _AddLayerNorm.apply(x, r, gamma, beta, norm.eps, hd)


class _GEGLU(torch.autograd.Function):

    @staticmethod
    def forward(ctx, h):
        h2 = h.reshape(-1, h.shape[-1])
        if not h2.is_contiguous():
            h2 = h2.contiguous()
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(h2)
        return ops.geglu_fwd(h2).view(*h.shape[:-1], h.shape[-1] // 2)

    @staticmethod
    def backward(ctx, dy):
        (h2, ) = ctx.saved_tensors
        d2 = dy.reshape(h2.shape[0], -1)
        if d2.dtype != h2.dtype or not d2.is_contiguous():
            d2 = d2.to(h2.dtype).contiguous()
        return ops.geglu_bwd(d2, h2).view(*dy.shape[:-1], h2.shape[-1])


class _QuickGELU(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x):
        xc = x if x.is_contiguous() else x.contiguous()
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(xc)
        return ops.quick_gelu_fwd(xc)

    @staticmethod
    def backward(ctx, dy):
        (xc, ) = ctx.saved_tensors
        d = dy if (dy.dtype == xc.dtype and dy.is_contiguous()) else dy.to(xc.dtype).contiguous()
        return ops.quick_gelu_bwd(d, xc)


def quick_gelu(x):
    """x * sigmoid(1.702 x) (CLIP text tower MLP). HIP path for half device tensors: one kernel each way instead of three
    elementwise launches forward and five backward."""
    if x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and x.numel() % 8 == 0:
        return _QuickGELU.apply(x)
    return x * torch.sigmoid(1.702 * x)


def geglu(h):
    """value * gelu(gate) for h = [value | gate] along the last dim (diffusers GEGLU). HIP path for half device tensors."""
    if h.is_cuda and h.dtype in (torch.float16, torch.bfloat16) and h.shape[-1] % 16 == 0:
        return _GEGLU.apply(h)
    a, g = h.chunk(2, dim=-1)
    return a * torch.nn.functional.gelu(g)


# ---- 3x3 convolution (frozen weights) on the implicit-GEMM kernel ---------------------------------------------------
class _ConvWeights:
    """(Cout,3,3,Cin) forward operand and (Cin,3,3,Cout) flipped/transposed backward-data operand of one frozen conv."""

    def __init__(self):
        self.key, self.fwd, self.bwd, self.bias = None, None, None, None

    def get(self, conv, dtype, need_bwd):
        w, b = conv.weight, conv.bias
        key = (w.data_ptr(), w._version, w.dtype, dtype, None if b is None else (b.data_ptr(), b._version))
        if key != self.key:
            self.key, self.bwd = key, None
            self.fwd = w.detach().permute(0, 2, 3, 1).to(dtype).contiguous()
            self.bias = None if b is None else b.detach().float().contiguous()
        if need_bwd and self.bwd is None:
            self.bwd = w.detach().flip(2, 3).permute(1, 2, 3, 0).to(dtype).contiguous()
        return self.fwd, self.bwd, self.bias


def _as_nhwc(t, dtype):
    if t.dtype != dtype:
        t = t.to(dtype)
    return t if ops._is_nhwc(t) else t.contiguous(memory_format=torch.channels_last)


class _Conv3x3(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, w_fwd, w_bwd, bias32, tbias, residual, upsample, gn_stats=False):
        dt = w_fwd.dtype
        xc = _as_nhwc(x, dt)
        tb = None if tbias is None else tbias.to(dt).contiguous()
        rs = None if residual is None else _as_nhwc(residual, dt)
        if gn_stats:
            y, part = ops.conv3x3_nhwc(xc, w_fwd, bias32, tb, rs, upsample, gn_stats=True)
            _attach_gn_stats(y, part)       # (the object returned here IS the one .apply hands to the caller)
        else:
            y = ops.conv3x3_nhwc(xc, w_fwd, bias32, tb, rs, upsample)
        ctx.save_for_backward(w_bwd)
        ctx.upsample, ctx.t_dtype, ctx.r_dtype, ctx.x_dtype = upsample, (None if tbias is None else tbias.dtype), (
            None if residual is None else residual.dtype), x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        (w_bwd, ) = ctx.saved_tensors
        cd = w_bwd.dtype if w_bwd is not None else dy.dtype
        if _grads_in_place and dy.dtype == cd and ops.nhwc_pixel_stride(dy) is not None:
            dyc = dy              # dense channels_last, or a channel slice of a concatenation's gradient: read in place
        else:
            dyc = _as_nhwc(dy, cd)
        dx = dt = dr = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv3x3_nhwc(dyc, w_bwd)
            if ctx.upsample:                   # adjoint of the nearest 2x upsample: sum of each 2x2 block
                dx = torch.nn.functional.avg_pool2d(dx, 2) * 4
            if dx.dtype != ctx.x_dtype:
                dx = dx.to(ctx.x_dtype)
        if ctx.t_dtype is not None and ctx.needs_input_grad[4]:
            dt = dyc.float().sum((2, 3)).to(ctx.t_dtype)
        if ctx.r_dtype is not None and ctx.needs_input_grad[5]:
            dr = dyc if dyc.dtype == ctx.r_dtype else dyc.to(ctx.r_dtype)
        return dx, None, None, None, dt, dr, None, None


def conv3x3(conv, x, tbias=None, residual=None, upsample=False, gn_groups=None):
    """`conv(x)` (+ tbias[:, :, None, None]) (+ residual) for an nn.Conv2d with a 3x3 / stride 1 / pad 1 kernel, optionally
    reading x through a nearest 2x upsample. HIP path (implicit-GEMM kernel, channels_last, epilogue-fused adds): device
    tensors, half activations (or half autocast), FROZEN weights, Cin % 64 == 0, Cout % 8 == 0. Anything else runs the
    plain torch ops (CPU oracle runs, fp32 inference, trainable convolutions, conv_in / conv_out): plumbing.
    gn_groups: the output (probably) feeds a GroupNorm of that many groups -- where that norm would read the map twice
    (large maps: level 0, the VAE) the kernel's epilogue leaves the norm's statistics with the output (see _attach_gn_stats)."""
    half = x.dtype in (torch.float16, torch.bfloat16)
    ac = torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') in (torch.float16, torch.bfloat16)
    ok = (x.is_cuda and (half or ac) and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)
          and conv.dilation == (1, 1) and conv.groups == 1 and conv.in_channels % 64 == 0 and conv.out_channels % 8 == 0
          and _frozen(conv.weight, conv.bias) and _conv_enabled
          and not conv._forward_hooks and not conv._forward_pre_hooks)
    need_bwd = torch.is_grad_enabled() and (x.requires_grad or (tbias is not None and tbias.requires_grad)
                                            or (residual is not None and residual.requires_grad))
    if ok:
        # measured against MIOpen on MI355X (profiles/r04_kernel_bench_ff_gn_conv.txt): with the split-K form of the
        # low-resolution levels the implicit-GEMM kernel wins at every UNet level (batch 4: 64x64 48 vs 112 us, 32x32 62 vs 77,
        # 16x16 68 vs 77, 8x8 26 vs 38; 512x768 sample: 32x48 50 vs 60, 16x24 50 vs 58, 8x12 23 vs 34 us) and takes the
        # time-embedding / residual adds with it; whole-step A/B of the dispatch threshold (MOS_CONV3X3_MIN_PIXELS, pixels below
        # which MIOpen is used): train 4096 -> 38.69, 0 -> 38.13 ms; regional sample 4096 -> 508.8, 3072 -> 500.2, 0 -> 475.0 ms
        pixels = x.shape[0] * x.shape[2] * x.shape[3] * (4 if upsample else 1)
        ok = pixels >= _conv_min_pixels and (not need_bwd or conv.out_channels % 64 == 0)
    if not ok:
        if upsample:
            x = torch.nn.functional.interpolate(x, scale_factor=2.0, mode='nearest')
        y = conv(x)
        if tbias is not None:
            y = y + tbias[:, :, None, None]
        if residual is not None:
            y = residual + y
        return y
    dt = x.dtype if half else torch.get_autocast_dtype('cuda')
    cache = conv.__dict__.get('_mos_conv_cache')
    if cache is None:
        cache = _ConvWeights()
        object.__setattr__(conv, '_mos_conv_cache', cache)
    w_fwd, w_bwd, bias32 = cache.get(conv, dt, need_bwd and x.requires_grad)
    want = False
    if gn_groups and _gn_from_conv and conv.out_channels % gn_groups == 0:
        up = 2 if upsample else 1
        want = _gn_from_conv_always or ops.groupnorm_reads_twice(x.shape[0], conv.out_channels, x.shape[2] * up * x.shape[3] * up,
                                                                 gn_groups)
    return _Conv3x3.apply(x, w_fwd, w_bwd, bias32, tbias, residual, bool(upsample), want)


_conv_s2_enabled = _os.environ.get('MOS_CONV3X3_S2', '1') != '0'      # host-side A/B switch (read once)


def conv3x3_stride2(conv, x, pad_bottom_right=False):
    """The down-samplers' 3x3 / stride-2 convolution (diffusers Downsample2D): `conv(x)` for padding 1 (UNet), or
    `conv(F.pad(x, (0, 1, 0, 1)))` for padding 0 (VAE encoder) WITHOUT building the padded copy. HIP path (round 6,
    mos_conv3x3_s2_nhwc: the raster implicit-GEMM kernel with stride-2 addressing): device tensors, half activations or half
    autocast, frozen weights, no gradient needed for x (forward only: the frozen VAE encoder, sampling); anything else -- the
    UNet down-samplers inside a training step need d/dx -- runs the torch ops."""
    half = x.dtype in (torch.float16, torch.bfloat16)
    ac = torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') in (torch.float16, torch.bfloat16)
    want_pad = (0, 0) if pad_bottom_right else (1, 1)
    ok = (_conv_s2_enabled and _conv_enabled and x.is_cuda and (half or ac) and conv.kernel_size == (3, 3) and conv.stride == (2, 2)
          and conv.padding == want_pad and conv.dilation == (1, 1) and conv.groups == 1 and conv.in_channels % 64 == 0
          and conv.out_channels % 8 == 0 and _frozen(conv.weight, conv.bias) and x.dim() == 4 and x.shape[2] > 1 and x.shape[3] > 1
          and not conv._forward_hooks and not conv._forward_pre_hooks
          and not (torch.is_grad_enabled() and x.requires_grad))
    if not ok:
        if pad_bottom_right:
            x = torch.nn.functional.pad(x, (0, 1, 0, 1))
        return conv(x)
    dt = x.dtype if half else torch.get_autocast_dtype('cuda')
    cache = conv.__dict__.get('_mos_conv_cache')
    if cache is None:
        cache = _ConvWeights()
        object.__setattr__(conv, '_mos_conv_cache', cache)
    w_fwd, _, bias32 = cache.get(conv, dt, False)
    return ops.conv3x3_s2_nhwc(_as_nhwc(x, dt), w_fwd, bias32, pad_mode=2 if pad_bottom_right else 1)


_conv_enabled = _os.environ.get('MOS_CONV3X3', '1') != '0'
_conv1x1_enabled = _os.environ.get('MOS_CONV1X1', '1') != '0'
_conv_min_pixels = int(_os.environ.get('MOS_CONV3X3_MIN_PIXELS', 0))


def set_conv3x3_enabled(flag):
    """A/B switch (bench / tests): False routes every 3x3 convolution through torch (MIOpen)."""
    global _conv_enabled
    _conv_enabled = bool(flag)


def conv1x1(conv, x, residual=None):
    """`conv(x)` (+ residual, in the GEMM's epilogue) for a frozen 1x1 nn.Conv2d (Transformer2DModel.proj_in / proj_out, ResnetBlock2D.conv_shortcut) as a plain
    GEMM of the library on the token-major view of a channels_last tensor (free view in, free view out). A LoRA-wrapped
    conv never gets here (its forward is LoRALinearLayer.forward); CPU / fp32 / trainable / NCHW inputs take torch."""
    half = x.dtype in (torch.float16, torch.bfloat16)
    ac = torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') in (torch.float16, torch.bfloat16)
    ok = (x.is_cuda and (half or ac) and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0)
          and conv.groups == 1 and conv.in_channels % 8 == 0 and conv.out_channels % 8 == 0
          and _frozen(conv.weight, conv.bias)
          and getattr(conv, '_mos_lora', None) is None and conv.forward.__func__ is torch.nn.Conv2d.forward
          and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and _conv_enabled and _conv1x1_enabled
          and not conv._forward_hooks and not conv._forward_pre_hooks)      # hooks (gradient fusion) need conv.__call__
    if not ok:
        y = conv(x)
        return y if residual is None else y + residual
    dt = x.dtype if half else torch.get_autocast_dtype('cuda')
    cache = conv.__dict__.get('_mos_cache')
    if cache is None:
        cache = WeightCache()
        object.__setattr__(conv, '_mos_cache', cache)
    b, c, h, w = x.shape
    W16, Wt16 = cache.weight('w', [conv.weight], dt, transposed=torch.is_grad_enabled() and x.requires_grad)
    b32 = cache.bias('w', [conv.bias])
    tokens = x.permute(0, 2, 3, 1).reshape(b * h * w, c)
    res2 = None
    if residual is not None:
        fuse = (_gemm_residual and residual.dim() == 4 and residual.shape == (b, conv.out_channels, h, w)
                and residual.dtype == dt and residual.is_contiguous(memory_format=torch.channels_last))
        if fuse:
            res2 = residual.permute(0, 2, 3, 1).reshape(b * h * w, conv.out_channels)     # free view of a channels_last tensor
    y = lora_linear(tokens if tokens.dtype == dt else tokens.to(dt), W16, Wt16, b32, [], residual=res2)
    y = y.view(b, h, w, -1).permute(0, 3, 1, 2)
    return y if (residual is None or res2 is not None) else y + residual

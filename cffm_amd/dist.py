"""Multi-GPU training step: batch-sharded data parallel, one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference is single-device only (no distributed code anywhere in /root/reference), so this is new
design.  Examples are independent in the forward pass; the only couplings are the loss normaliser (a mean
over the GLOBAL batch, CFFM.py:493) and the parameter gradients:

  1. local forward, per-example loss terms                      (no communication)
  2. local backward WITHOUT the 1/L normaliser                  (no communication; gradients are linear in dL/dout)
  3. ONE all-reduce (sum) of the flat dense gradient with the local loss-term sum in a spare slot
     (41 K floats at frappe, 5 M at F=32)
  4. 1/L = rsqrt(sum/Bg + 1e-10) is applied to the summed gradients inside the update kernels
  5. sparse tables, three modes:
       dense image (small vocabularies, e.g. frappe from 3 ranks up): every rank scatters its duplicates-summed row gradients
           into a zeroed dense [M][K|D|1] image that rides in the SAME all-reduce as the dense gradients; one collective
           per step and a plain sweep of the tables afterwards (rows nobody looked up carry an exact 0 and stay put);
       replicated (default while the tables fit one GPU):  ONE all-gather of packed (id, row gradients); every rank then
           runs the same sorted segment-sum + Adagrad over the Bg*F rows, so the replicas stay bit-identical;
       row-sharded (``ShardedTables``, vocabulary beyond one GPU's HBM): ids all-to-all to the owner
           (row r lives on rank r % G), owners gather and send rows back, row gradients return by all-to-all and
           the owner applies the duplicates-summed-first update locally.

``compute`` is any object with the HipEngine step-half interface (forward / backward / apply_dense /
apply_sparse + the tensors named below); the CPU tests plug the oracle in there, the product uses HipEngine.
"""
import numpy as np
import torch
import torch.distributed as dist


def sync_replicas(compute, group=None, tables=True, src=0):
    """Make what must be identical on every rank identical: broadcast rank ``src``'s replicated state (dense parameters
    and their optimizer slots; with ``tables`` also the three tables and their slots, which a data-parallel job
    replicates).  ``compute.replicated_state(tables)`` yields the tensors; a compute without it is left alone (the
    caller vouches for its replicas).  Gradients are only ever all-reduced, so replicas that start different stay
    different for ever: both step classes call this at construction."""
    state = getattr(compute, 'replicated_state', None)
    if state is None:
        return 0
    n = 0
    for t in state(tables):
        dist.broadcast(t, src=src, group=group)
        n += 1
    return n


def replicas_agree(compute, group=None, tables=False):
    """True when a checksum of the replicated state (sum and sum of squares in float64) is bit-identical on every
    rank: two all-reduces (MIN, MAX) of a few doubles."""
    state = getattr(compute, 'replicated_state', None)
    if state is None:
        return True
    ts = list(state(tables))
    if not ts:
        return True
    chk = torch.stack([v for t in ts for v in (t.double().sum(), (t.double() * t.double()).sum())])
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return bool(torch.equal(lo, hi))


class DataParallelStep(object):
    """Two collectives per step.  The loss normaliser 1/L (CFFM.py:493) needs the loss-term sum over the GLOBAL
    batch; instead of a separate scalar all-reduce between forward and backward, the backward pass runs with
    dL/dout = (out - y) / Bg (every gradient is linear in dL/dout), the local loss-term sum rides in a spare slot of
    the flat gradient buffer, and 1/L is applied to the summed gradients in the update:

        forward (local) -> backward_unscaled (local) -> all-reduce [grad | loss sum] -> all-gather packed rows
        (id | dEi | dEo | dfb) -> dp_apply: 1/L, dense Adagrad, sorted duplicates-first sparse Adagrad

    Every rank applies the same update to its replica, so the replicas stay bit-identical."""

    def __init__(self, compute, group=None, use_graph=False, mode='auto', sync=True):
        self.c = compute
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if sync and self.world > 1:
            # every variable is replicated in this mode: rank 0's parameters, tables and optimizer slots become everyone's
            sync_replicas(compute, group, tables=True)
        self._gathered = {}
        # use_graph: the whole step (local kernels + the two RCCL collectives + the update) is captured once per batch
        # shape into a HIP graph and replayed, with the batch copied into fixed input buffers first.  Eager, one step
        # costs ~200 us of host time (two torch.distributed calls), more than its ~150 us of GPU work at frappe.
        self.use_graph = use_graph
        self._graphs = {}
        self.mode = mode              # 'auto' | 'dense' (table gradients as a dense image, one all-reduce) | 'gather'

    def _eager(self, ids, y):
        c = self.c
        B = ids.shape[0]
        Bg = B * self.world
        dense = self.mode != 'gather' and hasattr(c, 'dp_dense_ok') and \
            c.dp_dense_ok(B, self.world if self.mode == 'auto' else 1 << 20)
        if dense:
            # small vocabulary: dense gradients and the dense image of the table gradients in ONE all-reduce
            flat = c.dp_local_dense(ids, y, B, Bg)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            return c.dp_apply_dense(flat, Bg)
        if hasattr(c, 'dp_local'):
            grad, rows = c.dp_local(ids, y, B, Bg)
        else:
            c.forward(ids, y)
            grad, rows = c.backward_unscaled(ids, y, B, Bg)
        out = self._gathered.get(rows.shape)
        if out is None:
            out = torch.empty((rows.shape[0] * self.world,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
            self._gathered[rows.shape] = out
        dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=self.group)
        dist.all_gather_into_tensor(out, rows, group=self.group)
        return c.dp_apply(grad, out, Bg, self.world) if hasattr(c, 'dp_local') else c.dp_apply(grad, out, Bg)

    def train_step(self, ids, y):
        """ids int32 [B,F], y fp32 [B]: this rank's shard of the global batch (same B on every rank).
        Returns the global loss as a device scalar tensor."""
        if not self.use_graph or not ids.is_cuda:
            return self._eager(ids, y)
        key = (tuple(ids.shape), ids.dtype, y.dtype)
        st = self._graphs.get(key)
        if st is None:
            st = {'ids': ids.clone(), 'y': y.clone(), 'calls': 0, 'graph': None, 'loss': None, 'ws_gen': -1}
            self._graphs[key] = st
        st['calls'] += 1
        self._drop_stale_graphs()
        if st['graph'] is None:
            if st['calls'] <= 2:                 # warm-up: workspaces, communicators and kernel attributes get created eagerly
                return self._eager(ids, y)
            st['ids'].copy_(ids)
            st['y'].copy_(y)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                st['loss'] = self._eager(st['ids'], st['y'])
            st['graph'] = g
            st['ws_gen'] = getattr(self.c, 'ws_generation', 0)
            if hasattr(self.c, 'pin_workspace'):
                self.c.pin_workspace()           # from now on an outgrown workspace is retired, not freed
                st['pinned'] = True
            g.replay()                           # capturing only records: run this batch now
            return st['loss']
        st['ids'].copy_(ids)
        st['y'].copy_(y)
        st['graph'].replay()
        return st['loss']

    def _drop_stale_graphs(self):
        """The engine re-allocated its workspace since some capture (a larger batch, evaluate()'s 8192-row blocks): those
        captured kernels still point at the old buffer, which the engine keeps alive while its generation is pinned.  EVERY
        graph of an older generation is dropped here - also the ones of batch shapes that may never be called again - so
        that the outgrown buffer (~50 GB at F32 D64 B8192) is freed now; a dropped graph is captured again, against the
        current buffer, at the next call of its shape."""
        gen = getattr(self.c, 'ws_generation', 0)
        stale = [st for st in self._graphs.values() if st['graph'] is not None and st['ws_gen'] != gen]
        if not stale:
            return 0
        if torch.cuda.is_available():
            torch.cuda.synchronize()             # no replay of a dropped graph is still running on the old buffer
        for st in stale:
            st['graph'] = None
            if st.pop('pinned', False):
                self.c.unpin_workspace(st['ws_gen'])
        return len(stale)


def shard_of(ids, world):
    """Owner rank and local row of every id under the r -> (r % G, r // G) row sharding."""
    return ids % world, ids // world


def route_ids(ids_flat, world):
    """Sort the lookups of one rank by owner: returns (perm, counts) with ids_flat[perm] grouped by owner
    rank in rank order and counts[g] lookups going to rank g.  Pure index arithmetic (no collective)."""
    owner = ids_flat % world
    perm = torch.argsort(owner, stable=True)
    counts = torch.bincount(owner, minlength=world)
    return perm, counts


class ShardedTables(object):
    """Row-sharded embedding tables: row r of every table lives on rank r % G at local index r // G.
    ``lookup`` and ``push_grads`` are the two exchange steps of a training step; both are all-to-all over
    the lookups of the batch, so their volume scales with B*F, not with the vocabulary."""

    def __init__(self, local_tables, group=None):
        # local_tables: dict name -> tensor [M_local, dim] holding rows rank, rank+G, rank+2G, ...
        self.t = local_tables
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._route = None

    def _exchange_counts(self, counts):
        recv = torch.empty_like(counts)
        dist.all_to_all_single(recv, counts, group=self.group)
        return recv

    def lookup(self, ids_flat):
        """ids_flat int64/int32 [n] global ids -> dict name -> [n, dim] rows, in the caller's order."""
        world = self.world
        perm, send_counts = route_ids(ids_flat.long(), world)
        recv_counts = self._exchange_counts(send_counts)
        sc, rc = send_counts.tolist(), recv_counts.tolist()
        req = torch.empty(sum(rc), dtype=torch.long, device=ids_flat.device)
        dist.all_to_all_single(req, ids_flat.long()[perm].contiguous(), rc, sc, group=self.group)
        local_rows = req // world
        out = {}
        for name, tab in self.t.items():
            rows = tab[local_rows].contiguous()
            back = torch.empty((ids_flat.numel(),) + tuple(tab.shape[1:]), dtype=tab.dtype, device=tab.device)
            dist.all_to_all_single(back, rows, sc, rc, group=self.group)
            res = torch.empty_like(back)
            res[perm] = back
            out[name] = res
        self._route = (perm, sc, rc, local_rows)
        return out

    def push_grads(self, row_grads):
        """row_grads: dict name -> [n, dim] gradients in the order of the last lookup.  Returns
        (local_rows [m], dict name -> [m, dim]) : what this rank's owner-side sparse update consumes."""
        perm, sc, rc, local_rows = self._route
        out = {}
        for name, g in row_grads.items():
            send = g[perm].contiguous()
            recv = torch.empty((sum(rc),) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
            dist.all_to_all_single(recv, send, rc, sc, group=self.group)
            out[name] = recv
        return local_rows, out


def local_rows_count(M, rank, world):
    """Rows of an M-row table owned by ``rank`` under r -> (r % G, r // G)."""
    return (M - rank + world - 1) // world


def shard_params(params, rank, world):
    """Global parameter dict -> this rank's view: the three tables keep rows rank, rank+G, ...; the rest is replicated."""
    out = dict(params)
    for k in ('inner_embeddings', 'outer_embeddings', 'feature_bias'):
        out[k] = params[k][rank::world].copy()
    return out


class _Plan(object):
    """Routing of one batch through the row-sharded tables: everything the exchange needs that depends only on the ids.
    Built on the device without a host sync; the per-destination counts reach the host through an asynchronous copy into
    pinned memory and are only waited for when the step that uses them starts (one step later when prefetched)."""
    __slots__ = ('ids', 'token', 'B', 'F', 'local_ids', 'order', 'uniq', 'pos', 'send_rows', 'counts_host', 'event', '_sc', '_rc')

    def counts(self):
        if self._sc is None:
            if self.event is not None:
                self.event.synchronize()
            c = self.counts_host.tolist()
            self._sc, self._rc = [int(v) for v in c[0]], [int(v) for v in c[1]]
        return self._sc, self._rc


def batch_token(ids):
    """What identifies the CONTENT of an id batch between the call that prefetched its plan and the call that uses it:
    storage address, geometry and torch's version counter.  Indexing a tensor (``X[i]``) returns a fresh Python object
    every time, so object identity would throw every prefetched plan away; an in-place write bumps ``_version``."""
    return (ids.data_ptr(), tuple(ids.shape), tuple(ids.stride()), ids.dtype, ids._version)


class ShardedStep(object):
    """Training step with ROW-SHARDED tables (vocabulary beyond one GPU's HBM; cfg5 of BASELINE.json): dense parameters
    replicated, row r of the three tables and of their Adagrad accumulators on rank r % G at local row r // G.

        plan (ids only, device-side, may run one step AHEAD): distinct (owner, local row) pairs of the batch in owner
            order, slot -> distinct-index map, per-owner counts + their all-to-all
        all-to-all (DISTINCT local rows)  -> owner-side cffm_gather_packed -> all-to-all (packed rows back)
        -> cffm_stage_packed (duplicates re-expanded into ws.Ei/Eo/fb) -> forward / backward_unscaled (local)
        -> all-reduce [dense grad | loss sum]
        -> cffm_pack_rows_dedup (duplicates of an id summed in slot order) -> all-to-all (row gradients, keyed by the
           owner's local row) -> dp_apply on the owner: 1/L, dense Adagrad (identical on every rank), duplicates across
           ranks summed first, then one sparse Adagrad update per row it owns

    Exchange volume scales with the DISTINCT lookups of the batch, never with the vocabulary.  The only host
    synchronisation is the read of the per-owner counts (RCCL's all-to-all takes its split sizes from the host); with
    ``train_step(ids, y, next_ids=...)`` the plan of the next batch is issued before this step's kernels, so that read
    never waits.  ``compute`` owns a LOCAL engine (cfg.M = local_rows_count): HipEngine in the product, the oracle in the
    CPU tests."""

    def __init__(self, compute, group=None, dedup=True, sync=True, M_global=None):
        self.c = compute
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.dedup = dedup
        # rows of the GLOBAL tables (the compute engine only knows its own share): bounds the local-row bits of the device-side
        # plan.  Default: the largest vocabulary whose shares are what the ranks own (local_rows_count is within 1 across ranks).
        if M_global is None and hasattr(compute, 'cfg'):
            on_host = dist.get_backend(group) == 'gloo'
            m_loc = torch.tensor([int(compute.cfg.M)], dtype=torch.int64, device='cpu' if on_host else getattr(compute, 'device', 'cpu'))
            if self.world > 1:
                dist.all_reduce(m_loc, op=dist.ReduceOp.MAX, group=group)
            M_global = int(m_loc.item()) * self.world
        self.M_global = M_global
        self._ahead = None
        self.plans_built = self.plans_reused = 0      # how often a prefetched plan was consumed (tests, bench)
        if sync and self.world > 1:
            # only the dense parameters are replicated here (every rank owns its own rows of the tables)
            sync_replicas(compute, group, tables=False)

    def _a2a(self, send, send_counts, recv_counts):
        recv = torch.empty((sum(recv_counts),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        dist.all_to_all_single(recv, send, recv_counts, send_counts, group=self.group)
        return recv

    def plan_torch(self, ids):
        """The plan as torch operations on any device: the specification of cffm_shard_plan (tests compare the two), and what the
        CPU tests (gloo, the oracle as compute) run."""
        G = self.world
        B, F = ids.shape
        n = B * F
        dev = ids.device
        flat = ids.reshape(-1).to(torch.int64)
        owner, local = flat % G, flat // G
        comp = owner * (1 << 32) + local                              # owner-major, then the owner's local row
        scomp, order = torch.sort(comp, stable=True)                  # slots ascend inside a run of equal keys
        if self.dedup:
            head = torch.ones(n, dtype=torch.bool, device=dev)
            head[1:] = scomp[1:] != scomp[:-1]
            uniq = torch.cumsum(head, 0) - 1                          # distinct-id index of every sorted position
            send_rows = torch.zeros(n, dtype=torch.int64, device=dev)
            send_rows[uniq] = scomp & 0xffffffff                      # capacity n, the first #distinct entries are used
            counts = torch.zeros(G, dtype=torch.int64, device=dev).index_add_(0, scomp >> 32, head.to(torch.int64))
        else:
            uniq = torch.arange(n, dtype=torch.int64, device=dev)
            send_rows = scomp & 0xffffffff
            counts = torch.bincount(owner, minlength=G)
        pos = torch.empty(n, dtype=torch.int64, device=dev)
        pos[order] = uniq                                             # slot -> record of the answer
        return (local.to(torch.int32).reshape(B, F).contiguous(), order.to(torch.int32), uniq.to(torch.int32), pos.to(torch.int32),
                send_rows.to(torch.int32), counts)

    def plan(self, ids):
        G = self.world
        B, F = ids.shape
        dev = ids.device
        if self.dedup and hasattr(self.c, 'shard_plan') and self.M_global is not None:
            # the product path: five small launches in the library instead of ~15 torch operations with int64 temporaries
            local_ids, order, uniq, pos, send_rows, counts = self.c.shard_plan(ids, G, self.M_global)
        else:
            local_ids, order, uniq, pos, send_rows, counts = self.plan_torch(ids)
        recv_counts = torch.empty_like(counts)
        dist.all_to_all_single(recv_counts, counts, group=self.group)
        both = torch.stack([counts, recv_counts])
        p = _Plan()
        self.plans_built += 1
        p.ids, p.token, p.B, p.F = ids, batch_token(ids), B, F
        p.local_ids, p.order, p.uniq, p.pos, p.send_rows = local_ids, order, uniq, pos, send_rows
        p._sc = p._rc = None
        if dev.type == 'cuda':
            p.counts_host = torch.empty((2, G), dtype=torch.int64, pin_memory=True)
            p.counts_host.copy_(both, non_blocking=True)
            p.event = torch.cuda.Event()
            p.event.record()
        else:
            p.counts_host, p.event = both, None
        return p

    def train_step(self, ids, y, next_ids=None):
        c = self.c
        plan, self._ahead = self._ahead, None
        if plan is None or plan.token != batch_token(ids):
            plan = self.plan(ids)
        else:
            self.plans_reused += 1
        if next_ids is not None:
            self._ahead = self.plan(next_ids)         # before this step's kernels: its counts are on the host long before needed
        B, Bg = plan.B, plan.B * self.world
        sc, rc = plan.counts()
        u = sum(sc)
        # 1) ask the owners for the DISTINCT rows, by their local row index; the answer is one packed record per row
        asked = self._a2a(plan.send_rows[:u], sc, rc)
        got = self._a2a(c.gather_packed(asked), rc, sc)                  # [u, K+D+4] in distinct-id order
        if u > 0 and hasattr(c, 'packed_ok') and c.packed_ok():
            # wide shapes: the records are consumed where they lie (slot i reads record pos[i] in the kernel that fetches it, the
            # product-path gather of the replicated tables with a record stride): nothing is staged into ws.Ei / ws.Eo
            c.forward_packed(got, plan.pos, y, B)
            grad = c.backward_unscaled_packed(got, plan.pos, y, B, Bg)
        else:
            c.stage_packed(got, plan.pos, B)                             # duplicates re-expanded: ws.Ei / ws.Eo / ws.fb
            c.forward_staged(y, B)
            # 2) local backward without 1/L, gradients keyed by the owner's local row, duplicates summed before they travel
            grad = c.backward_unscaled(plan.local_ids, y, B, Bg, pack=False)[0]
        dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=self.group)
        rows = c.pack_rows_dedup(plan.local_ids, plan.order, plan.uniq, B)
        recv = self._a2a(rows[:u], sc, rc)
        return c.dp_apply(grad, recv, Bg)


# ---- checkpoint of a row-sharded model ("sharded-table aware") ----------------------------------------------
# Every rank writes its own shard of the three tables and of their optimizer slots (local rows rank, rank + G, ...); the
# replicated dense parameters (identical on every rank) are written by rank 0 only.  Plain tensors: weights_only loads.
def save_sharded(engine, path, rank, world, opt_step=0):
    t = lambda d, keys: {k: torch.from_numpy(d[k].copy()) for k in keys}
    tables = ('inner_embeddings', 'outer_embeddings', 'feature_bias')
    params, accs = engine.export_params(), engine.export_accumulators()
    second = engine.export_second_moments() if hasattr(engine, 'export_second_moments') else None
    blob = {'format': 2, 'rank': int(rank), 'world': int(world), 'local_rows': int(params['inner_embeddings'].shape[0]),
            'tables': t(params, tables), 'table_slots': t(accs, tables),
            'table_slots2': t(second, tables) if second is not None else None, 'opt_step': int(opt_step)}
    if rank == 0:
        dense = [k for k in params if k not in tables]
        blob['dense'] = t(params, dense)
        blob['dense_slots'] = t(accs, [k for k in dense if k in accs])
        blob['dense_slots2'] = t(second, [k for k in dense if k in second]) if second is not None else None
    torch.save(blob, '%s.shard%d-of-%d.pt' % (path, rank, world))


def load_sharded(engine, path, rank, world):
    """Restores what save_sharded wrote at the SAME world size (a shard holds rows r with r % world == rank)."""
    own = torch.load('%s.shard%d-of-%d.pt' % (path, rank, world), weights_only=True)
    if own['world'] != world or own['rank'] != rank:
        raise ValueError('shard file written for rank %d of %d' % (own['rank'], own['world']))
    root = own if rank == 0 else torch.load('%s.shard0-of-%d.pt' % (path, world), weights_only=True)
    n = lambda d: {k: v.numpy() for k, v in d.items()}
    params, accs = n(root['dense']), n(root['dense_slots'])
    params.update(n(own['tables']))
    accs.update(n(own['table_slots']))
    second = None
    if own.get('table_slots2') is not None:
        second = n(root['dense_slots2'])
        second.update(n(own['table_slots2']))
    if params['inner_embeddings'].shape[0] != engine.cfg.M:
        raise ValueError('shard has %d local rows, this engine owns %d' % (params['inner_embeddings'].shape[0], engine.cfg.M))
    for k, v in engine.export_params().items():          # untrained reference variables kept on the host only
        params.setdefault(k, v)
        accs.setdefault(k, np.zeros_like(v))
    engine.load_params(params, accs, second)
    return int(own.get('opt_step', 0))

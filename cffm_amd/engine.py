"""Device-side state and step functions of the CFFM hot path: the replacement for the TF session.

``HipEngine`` owns what ``tf.Session`` owned in the reference (CFFM.py:160): the variables and their
Adagrad accumulators, resident in HBM for the life of the object as torch tensors, plus a per-batch-size
workspace.  ``predict`` and ``train_step`` are the two ``sess.run`` call shapes (CFFM.py:596, :200); both
are single calls into libcffm_hip.so, asynchronous on torch's current stream.
"""
import ctypes as C

import numpy as np
import torch

from . import hip
from .spec import ADAGRAD_INIT_ACC, CFFMConfig, init_params, param_shapes

# theta member -> reference variable name (reshaped flat, row-major)
_THETA_MEMBERS = [
    ('att_W', 'bias_W'), ('att_b', 'bias_b'), ('bias', 'bias'),
    ('inner_cw', 'inner_layer_conv_weight_0'), ('inner_cb', 'inner_layer_conv_bias_0'),
    ('inner_dw', 'dense_kernel'), ('inner_db', 'dense_bias'),
    ('d1_w', 'dense_1_kernel'), ('d1_b', 'dense_1_bias'), ('d2_w', 'dense_2_kernel'), ('d2_b', 'dense_2_bias'),
    ('lin_w', 'dense_3_kernel'), ('lin_b', 'dense_3_bias'),
]


def _ptr(t):
    """Borrowed device pointer of a torch tensor as a plain integer (0 for None): what both bindings take."""
    return t.data_ptr() if t is not None else 0


class WorkspacePool(object):
    """ONE grow-only byte buffer plus the bookkeeping that captured HIP graphs need.

    A captured graph has the buffer's address baked into its kernel arguments.  Every re-allocation starts a new
    GENERATION; ``pin()`` says "a graph now references the current generation" and returns that generation, ``unpin(gen)``
    says that graph was dropped or re-captured.  A buffer that is outgrown while its generation is pinned is RETIRED (kept
    alive) and freed the moment the last pin of ITS generation goes - not when the pin count of all generations reaches 0:
    with two captured batch shapes one regrowth used to leave a ~50 GB buffer (F32 D64 B8192) alive for as long as any
    graph existed.  ``alloc(nbytes)`` is the allocator (torch.empty on the device; tests pass a host one)."""

    def __init__(self, alloc):
        self._alloc = alloc
        self.buf = None
        self.generation = 0
        self.pins = {}          # generation -> captured graphs that reference that generation's buffer
        self.retired = {}       # generation -> outgrown buffer those graphs may still replay against

    def get(self, nbytes):
        nbytes = int(nbytes)
        if self.buf is None or self.buf.numel() < nbytes:
            if self.buf is not None:
                if self.pins.get(self.generation, 0) > 0:
                    self.retired[self.generation] = self.buf
                self.generation += 1
            self.buf = None                               # release (unless retired) before allocating the larger one
            self.buf = self._alloc(nbytes)
        return self.buf

    def pin(self):
        self.pins[self.generation] = self.pins.get(self.generation, 0) + 1
        return self.generation

    def unpin(self, generation):
        left = self.pins.get(generation, 0) - 1
        if left > 0:
            self.pins[generation] = left
        else:
            self.pins.pop(generation, None)
            self.retired.pop(generation, None)            # the last graph of that generation is gone: free its buffer

    def retired_bytes(self):
        return sum(int(b.numel()) for b in self.retired.values())


class HipEngine(object):
    def __init__(self, cfg, params=None, seed=2021, device='cuda:0', table_seed=None):
        if not torch.cuda.is_available():
            raise RuntimeError('cffm_amd.HipEngine needs an MI355X (torch.cuda.is_available() is False); '
                               'there is no CPU fallback')
        hip.load()
        self.lib = hip.fast()                         # pybind11 layer (ctypes when it is not built): integer pointers
        self.cfg = cfg
        self.device = torch.device(device)
        self.shape = hip.make_shape(cfg)
        self.tl = hip.theta_layout(self.shape)
        self._ws = {}
        self._pool = WorkspacePool(lambda nbytes: torch.empty(nbytes, dtype=torch.uint8, device=self.device))
        self._eval_scratch = None
        self._flat_dirty = False                      # dense-image route: the image is zero on entry, zero on exit
        self._host_only = {}
        # params == 'device': the three tables are drawn ON the GPU with the reference's distributions (N(0, 0.1),
        # N(0, 0.01), exact zeros - CFFM.py:257-277); for vocabularies where a host-side draw + copy of M*(K+D) floats
        # would dominate start-up (10 M features: 5 GB).  Dense parameters still come from init_params(seed); the device
        # draw of the tables uses table_seed (default: seed).  Row-sharded ranks pass ONE seed for the replicated dense
        # parameters and table_seed = seed + rank for their own shard of the tables.
        device_tables = isinstance(params, str) and params == 'device'
        if params is None or device_tables:
            params = init_params(cfg, seed=seed, tables=not device_tables)
        n = int(self.tl.n)
        dev = self.device
        self.theta = torch.zeros(n, dtype=torch.float32, device=dev)
        # optimizer slots (CFFM.py:517-529): Adagrad accumulators start at 1e-8, Momentum accumulators / Adam m, v at 0
        acc0 = ADAGRAD_INIT_ACC if cfg.optimizer == 'AdagradOptimizer' else 0.0
        self.acc0 = acc0
        self.theta_acc = torch.full((n,), acc0, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n + 4, dtype=torch.float32, device=dev)[:n]      # +4: loss-sum slot of the DP step
        self._grad_full = self.grad._base if self.grad._base is not None else self.grad
        M = cfg.M
        self.inner = torch.zeros((M, cfg.K), dtype=torch.float32, device=dev)
        self.outer = torch.zeros((M, cfg.D), dtype=torch.float32, device=dev)
        self.fbias = torch.zeros((M,), dtype=torch.float32, device=dev)
        self.inner_acc = torch.full_like(self.inner, acc0)
        self.outer_acc = torch.full_like(self.outer, acc0)
        self.fbias_acc = torch.full_like(self.fbias, acc0)
        self.opt_step = 0
        self.theta_acc2 = self.tables_acc2 = None
        if cfg.optimizer == 'AdamOptimizer':                         # second moment
            self.theta_acc2 = torch.zeros_like(self.theta_acc)
            self.inner_acc2, self.outer_acc2 = torch.zeros_like(self.inner), torch.zeros_like(self.outer)
            self.fbias_acc2 = torch.zeros_like(self.fbias)
            self.tables_acc2 = hip.Tables(self.inner_acc2.data_ptr(), self.outer_acc2.data_ptr(), self.fbias_acc2.data_ptr())
        self.tables = hip.Tables(self.inner.data_ptr(), self.outer.data_ptr(), self.fbias.data_ptr())
        self.tables_acc = hip.Tables(self.inner_acc.data_ptr(), self.outer_acc.data_ptr(), self.fbias_acc.data_ptr())
        self.loss_buf = torch.zeros(1, dtype=torch.float32, device=dev)
        # addresses of the small host structs (kept alive by self)
        self._s = C.addressof(self.shape)
        self._t = C.addressof(self.tables)
        self._ta = C.addressof(self.tables_acc)
        self._ta2 = C.addressof(self.tables_acc2) if self.tables_acc2 is not None else 0
        self.load_params(params)
        if device_tables:
            gen = torch.Generator(device=dev).manual_seed(int(seed if table_seed is None else table_seed))
            self.inner.normal_(0.0, 0.1, generator=gen)
            self.outer.normal_(0.0, 0.01, generator=gen)

    def replicated_state(self, tables=True):
        """Tensors that must be bit-identical on every rank of a multi-GPU job (cffm_amd.dist.sync_replicas): the dense
        parameters and their optimizer slots; with ``tables`` (data-parallel mode) also the three tables and their slots.
        In row-sharded mode the tables are per-rank shards and stay out."""
        out = [self.theta, self.theta_acc]
        if self.theta_acc2 is not None:
            out.append(self.theta_acc2)
        if tables:
            out += [self.inner, self.outer, self.fbias, self.inner_acc, self.outer_acc, self.fbias_acc]
            if self.theta_acc2 is not None:
                out += [self.inner_acc2, self.outer_acc2, self.fbias_acc2]
        return out

    # ---- named parameters <-> device buffers -------------------------------------------------------
    def _members(self):
        out = [(getattr(self.tl, m), name) for m, name in _THETA_MEMBERS]
        for l in range(self.tl.live):
            out.append((self.tl.conv_w[l], 'outer_layer_conv_weight_%d' % l))
            out.append((self.tl.conv_b[l], 'outer_layer_conv_bias_%d' % l))
        return out

    def _pack(self, name, v):
        """Reference variable -> its flat image inside theta (conv weights/biases are channel-padded to Pp)."""
        v = np.asarray(v, dtype=np.float32)
        P, Pp = self.tl.P, self.tl.Pp
        if name.startswith('outer_layer_conv_weight_'):
            out = np.zeros((4, Pp, Pp), dtype=np.float32)
            out[:, :P, :P] = v.reshape(4, P, P)
            return out.reshape(-1)
        if name.startswith('outer_layer_conv_bias_'):
            out = np.zeros(Pp, dtype=np.float32)
            out[:P] = v.reshape(-1)
            return out
        return v.reshape(-1)

    def _unpack(self, name, flat, off, shp):
        P, Pp = self.tl.P, self.tl.Pp
        if name.startswith('outer_layer_conv_weight_'):
            return flat[off:off + 4 * Pp * Pp].reshape(4, Pp, Pp)[:, :P, :P].reshape(shp).copy()
        if name.startswith('outer_layer_conv_bias_'):
            return flat[off:off + P].copy()
        n = int(np.prod(shp)) if shp else 1
        return flat[off:off + n].reshape(shp).copy()

    def load_params(self, params, accs=None, accs2=None):
        shapes = param_shapes(self.cfg)
        host = np.zeros(int(self.tl.n), dtype=np.float32)
        for off, name in self._members():
            v = self._pack(name, params[name])
            host[off:off + v.size] = v
        self.theta.copy_(torch.from_numpy(host))
        if 'inner_embeddings' in params:                  # absent: tables initialised on the device (params='device')
            self.inner.copy_(torch.from_numpy(np.asarray(params['inner_embeddings'], dtype=np.float32)))
            self.outer.copy_(torch.from_numpy(np.asarray(params['outer_embeddings'], dtype=np.float32)))
            self.fbias.copy_(torch.from_numpy(np.asarray(params['feature_bias'], dtype=np.float32).reshape(-1)))
        trained = set(n for _, n in self._members()) | {'inner_embeddings', 'outer_embeddings', 'feature_bias'}
        self._host_only = {k: np.array(params[k], dtype=np.float32) for k in shapes if k not in trained and k in params}
        if accs is not None:
            ha = np.full(int(self.tl.n), self.acc0, dtype=np.float32)
            for off, name in self._members():
                v = self._pack(name, accs[name])
                if name.startswith('outer_layer_conv_'):
                    v = np.where(v == 0, np.float32(self.acc0), v)     # pads keep the initial accumulator
                ha[off:off + v.size] = v
            self.theta_acc.copy_(torch.from_numpy(ha))
            self.inner_acc.copy_(torch.from_numpy(np.asarray(accs['inner_embeddings'], dtype=np.float32)))
            self.outer_acc.copy_(torch.from_numpy(np.asarray(accs['outer_embeddings'], dtype=np.float32)))
            self.fbias_acc.copy_(torch.from_numpy(np.asarray(accs['feature_bias'], dtype=np.float32).reshape(-1)))
        if accs2 is not None and self.theta_acc2 is not None:
            h2 = np.zeros(int(self.tl.n), dtype=np.float32)
            for off, name in self._members():
                v = self._pack(name, accs2[name])
                h2[off:off + v.size] = v
            self.theta_acc2.copy_(torch.from_numpy(h2))
            self.inner_acc2.copy_(torch.from_numpy(np.asarray(accs2['inner_embeddings'], dtype=np.float32)))
            self.outer_acc2.copy_(torch.from_numpy(np.asarray(accs2['outer_embeddings'], dtype=np.float32)))
            self.fbias_acc2.copy_(torch.from_numpy(np.asarray(accs2['feature_bias'], dtype=np.float32).reshape(-1)))

    def _export(self, flat, inner, outer, fbias):
        shapes = param_shapes(self.cfg)
        host = flat.detach().cpu().numpy()
        out = {}
        for off, name in self._members():
            out[name] = self._unpack(name, host, off, shapes[name])
        out['inner_embeddings'] = inner.detach().cpu().numpy().copy()
        out['outer_embeddings'] = outer.detach().cpu().numpy().copy()
        out['feature_bias'] = fbias.detach().cpu().numpy().reshape(-1, 1).copy()
        return out

    def export_params_dense(self):
        """Every variable except the three tables (for vocabularies where a host copy of the tables is not wanted)."""
        z = torch.zeros(1, device=self.device)
        out = self._export(self.theta, z, z, z)
        for k in ('inner_embeddings', 'outer_embeddings', 'feature_bias'):
            out.pop(k)
        out.update({k: v.copy() for k, v in self._host_only.items()})
        return out

    def export_params(self):
        out = self._export(self.theta, self.inner, self.outer, self.fbias)
        out.update({k: v.copy() for k, v in self._host_only.items()})
        return out

    def export_accumulators(self):
        return self._export(self.theta_acc, self.inner_acc, self.outer_acc, self.fbias_acc)

    def export_second_moments(self):
        """Adam's v slots (None for the other optimizers)."""
        if self.theta_acc2 is None:
            return None
        return self._export(self.theta_acc2, self.inner_acc2, self.outer_acc2, self.fbias_acc2)

    def export_grad(self):
        """Dense-parameter gradients of the last backward, by reference variable name."""
        z = torch.zeros(1, device=self.device)
        g = self._export(self.grad, z, z, z)
        for k in ('inner_embeddings', 'outer_embeddings', 'feature_bias'):
            g.pop(k)
        return g

    # ---- workspace ---------------------------------------------------------------------------------
    def workspace(self, B):
        """(buffer, layout) for a batch of B rows.  ONE buffer serves every batch size: the library recomputes the
        layout from B on every call, and a call only needs ``layout(B).bytes`` bytes from the start of the buffer, so
        the buffer is re-allocated exactly when a layout asks for more bytes than it has (the byte count is NOT monotonic
        in B - the slab plan of the backward changes with B - hence the check on bytes, never on B).  It only ever grows;
        at F32 D64 B8192 it is ~50 GB, so one copy per distinct B would exhaust HBM within a few steps.

        A captured HIP graph (DataParallelStep(use_graph=True)) has the buffer's address baked into its kernel arguments:
        see WorkspacePool - an outgrown buffer is kept alive exactly as long as a graph captured against it is pinned, and
        ``ws_generation`` changes so that the owner of the graph re-captures against the new buffer."""
        B = int(B)
        wl = self._ws.get(B)
        if wl is None:
            if len(self._ws) > 4096:                     # layouts are tiny structs, but do not hoard them forever
                self._ws = {k: v for k, v in self._ws.items() if not isinstance(k, int)}
            wl = hip.ws_layout(self.shape, B)
            self._ws[B] = wl
        return self._pool.get(wl.bytes), wl

    @property
    def ws_generation(self):
        return self._pool.generation

    def reserve_workspace(self, B):
        """Grow the workspace to what a batch of B rows needs (e.g. evaluate()'s 8192-row blocks) BEFORE a graph is
        captured, so that no later call outgrows the captured buffer."""
        return self.workspace(B)[0]

    def pin_workspace(self):
        """A graph that references the current workspace buffer exists from now on; returns the generation to unpin."""
        return self._pool.pin()

    def unpin_workspace(self, generation):
        """The graph pinned at `generation` was dropped or re-captured."""
        self._pool.unpin(generation)

    def ws_tensor(self, B, member, shape, dtype=torch.float32, index=None):
        """View of one workspace intermediate (for the parity tests)."""
        buf, wl = self.workspace(B)
        off = getattr(wl, member)
        if index is not None:
            off = off[index]
        n = int(np.prod(shape))
        itemsize = torch.empty(0, dtype=dtype).element_size()
        return buf[int(off):int(off) + n * itemsize].view(dtype).reshape(shape)

    def _stream(self):
        return int(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _ids(ids):
        if ids.dtype != torch.int32 or not ids.is_contiguous():
            ids = ids.to(torch.int32).contiguous()
        return ids

    # ---- the two sess.run call shapes ---------------------------------------------------------------
    def predict(self, ids):
        """sess.run(self.out) (CFFM.py:596): int32 ids [B,F] on device -> fp32 [B] on device."""
        ids = self._ids(ids)
        B = ids.shape[0]
        if B == 0:
            return torch.empty(0, dtype=torch.float32, device=self.device)
        buf, _ = self.workspace(B)
        out = torch.empty(B, dtype=torch.float32, device=self.device)
        hip.check(self.lib.cffm_predict(self._s, self._t, _ptr(self.theta), _ptr(ids),
                                        B, _ptr(buf), _ptr(out), self._stream()))
        return out

    def eval_sums(self, ids, y, lo, hi, block=8192):
        """evaluate()'s sweep (CFFM.py:590-614) entirely on the device: ordered blocks of rows through cffm_predict, the
        clip to [lo, hi] and the float64 sums the two metrics need (cffm_eval_sums).  Returns a float64 device tensor
        [sum (y - p)^2, sum y, sum y^2]; nothing is copied to the host and nothing synchronises here."""
        ids = self._ids(ids)
        y = y.reshape(-1)
        n = int(ids.shape[0])
        if self._eval_scratch is None:
            self._eval_scratch = torch.empty(int(self.lib.cffm_eval_scratch_bytes()), dtype=torch.uint8, device=self.device)
        sums = torch.zeros(3, dtype=torch.float64, device=self.device)
        for s0 in range(0, n, block):
            m = min(block, n - s0)
            out = self.predict(ids[s0:s0 + m])
            hip.check(self.lib.cffm_eval_sums(_ptr(out), _ptr(y[s0:s0 + m]), m, float(lo), float(hi),
                                              _ptr(self._eval_scratch), _ptr(sums), self._stream()))
        return sums

    def train_step(self, ids, y):
        """sess.run((self.loss, self.optimizer)) (CFFM.py:200).  Returns the loss as a device scalar
        (no host sync)."""
        ids = self._ids(ids)
        y = y.reshape(-1)
        if y.dtype != torch.float32 or not y.is_contiguous():
            y = y.to(torch.float32).contiguous()
        B = ids.shape[0]
        if B == 0:                                   # nothing to learn from: parameters and slots stay as they are
            return self.loss_buf
        buf, _ = self.workspace(B)
        if self.cfg.optimizer != 'AdagradOptimizer':
            self.opt_step += 1
            hip.check(self.lib.cffm_train_step_opt(
                self._s, self._t, self._ta,
                self._ta2, _ptr(self.theta),
                _ptr(self.theta_acc), _ptr(self.theta_acc2), _ptr(self.grad), _ptr(ids), _ptr(y), B, _ptr(buf),
                _ptr(self.loss_buf), self.opt_step, self._stream()))
            return self.loss_buf
        hip.check(self.lib.cffm_train_step(self._s, self._t, self._ta,
                                           _ptr(self.theta), _ptr(self.theta_acc), _ptr(self.grad), _ptr(ids),
                                           _ptr(y), B, _ptr(buf), _ptr(self.loss_buf), self._stream()))
        return self.loss_buf

    # ---- halves of the step (multi-GPU path and tests) ---------------------------------------------
    def forward(self, ids, y=None):
        ids = self._ids(ids)
        B = ids.shape[0]
        buf, _ = self.workspace(B)
        hip.check(self.lib.cffm_forward(self._s, self._t, _ptr(self.theta), _ptr(ids),
                                        _ptr(y), B, _ptr(buf), self._stream()))

    def gather_inner_fwd_ok(self):
        """Wide shapes (F*(F-1)/2 > 64, K == D in {32, 64}, F <= 32): predict / train_step never materialise the rows."""
        return bool(self.lib.cffm_gather_inner_fwd_ok(self._s))

    def gather_inner_fwd(self, ids):
        """cffm_gather_inner_fwd alone: the three lookups fused with the inner branch, the s0 pool and the first-order
        inputs (ws.inner_out, ws.t1[:, :D], ws.fb, ws.sort_keys); the kernel the gather roofline is quoted on."""
        ids = self._ids(ids)
        B = ids.shape[0]
        buf, _ = self.workspace(B)
        hip.check(self.lib.cffm_gather_inner_fwd(self._s, self._t, _ptr(self.theta), _ptr(ids), B, _ptr(buf), self._stream()))

    def shard_plan(self, ids, world, M_global):
        """Routing plan of a [B,F] id batch through tables of M_global rows sharded r -> (r % world, r // world), entirely on the
        device (cffm_shard_plan: pack, one radix sort, head flags, scan, scatter - no int64 temporaries, no torch.sort).
        Returns int32 tensors (local_ids [B,F], order, uniq, pos, send_rows [B*F]) and counts int64 [world]."""
        ids = self._ids(ids)
        n = int(ids.numel())
        key = ('plan', n)
        scratch = self._ws.get(key)
        if scratch is None:
            nbytes = int(self.lib.cffm_shard_plan_scratch_bytes(n))
            if nbytes < 0:
                raise RuntimeError('cffm_shard_plan_scratch_bytes failed')
            scratch = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws[key] = scratch
        dev = self.device
        local_ids = torch.empty(ids.shape, dtype=torch.int32, device=dev)
        order, uniq, pos, send_rows = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(4))
        counts = torch.empty(int(world), dtype=torch.int64, device=dev)
        hip.check(self.lib.cffm_shard_plan(_ptr(ids), n, int(world), int(M_global), _ptr(scratch), _ptr(local_ids), _ptr(order),
                                           _ptr(uniq), _ptr(pos), _ptr(send_rows), _ptr(counts), self._stream()))
        return local_ids, order, uniq, pos, send_rows, counts

    def gather_packed(self, local_rows):
        """Owner side of a row-sharded lookup: int32 [m] local rows -> [m, K+D+4] packed records
        (inner | outer | bias, 0, 0, 0), ONE kernel (cffm_gather_packed)."""
        m = int(local_rows.numel())
        Wp = self.cfg.K + self.cfg.D + 4
        out = torch.empty((m, Wp), dtype=torch.float32, device=self.device)
        if m:
            rows = self._ids(local_rows.reshape(-1))
            hip.check(self.lib.cffm_gather_packed(self._s, self._t, _ptr(rows), m, _ptr(out), self._stream()))
        return out

    def stage_packed(self, packed, pos, B):
        """Requester side: slot i of the batch takes record pos[i] of packed [n_records, K+D+4] -> ws.Ei / ws.Eo / ws.fb."""
        buf, _ = self.workspace(B)
        pos = self._ids(pos.reshape(-1)) if pos is not None else None
        hip.check(self.lib.cffm_stage_packed(self._s, _ptr(packed), _ptr(pos), int(packed.shape[0]), int(B), _ptr(buf),
                                             self._stream()))

    def packed_ok(self):
        """True when the row-sharded step can consume the packed records where they lie (the wide shapes)."""
        return bool(self.lib.cffm_gather_inner_fwd_ok(self._s))

    def forward_packed(self, packed, pos, y, B):
        """Forward half of a row-sharded step straight from the received records [n_records, K+D+4]: slot i reads record pos[i]
        in the kernel that fetches it; ws.Ei / ws.Eo are never written (cffm_forward_packed)."""
        buf, _ = self.workspace(B)
        pos = self._ids(pos.reshape(-1))
        hip.check(self.lib.cffm_forward_packed(self._s, _ptr(self.theta), _ptr(packed), _ptr(pos), int(packed.shape[0]), _ptr(y),
                                               int(B), _ptr(buf), self._stream()))

    def backward_unscaled_packed(self, packed, pos, y, B, B_global):
        """Backward half over the same records (dL/dout = (out - y) / B_global); the row gradients stay in the workspace for
        pack_rows_dedup.  Returns grad_full [n+4] with this rank's loss-term sum at index n."""
        buf, _ = self.workspace(B)
        pos = self._ids(pos.reshape(-1))
        hip.check(self.lib.cffm_backward_unscaled_packed(self._s, _ptr(self.theta), _ptr(packed), _ptr(pos), int(packed.shape[0]),
                                                         _ptr(y), int(B), int(B_global), _ptr(buf), _ptr(self._grad_full),
                                                         self._stream()))
        return self._grad_full

    def forward_staged(self, y, B):
        """cffm_forward over rows that are already staged in the workspace (tab = NULL)."""
        buf, _ = self.workspace(B)
        hip.check(self.lib.cffm_forward(self._s, 0, _ptr(self.theta), 0, _ptr(y), int(B), _ptr(buf), self._stream()))

    def forward_rows(self, Ei, Eo, fb, y, B=None):
        """Forward from rows that are already looked up, given as three tensors Ei [B,F,K], Eo [B,F,D], fb [B,F]."""
        F = self.cfg.F
        B = Ei.numel() // (F * self.cfg.K) if B is None else B
        self.workspace(B)
        self.ws_tensor(B, 'Ei', (B, F, self.cfg.K)).copy_(Ei.reshape(B, F, self.cfg.K))
        self.ws_tensor(B, 'Eo', (B, F, self.cfg.D)).copy_(Eo.reshape(B, F, self.cfg.D))
        self.ws_tensor(B, 'fb', (B, F)).copy_(fb.reshape(B, F))
        self.forward_staged(y, B)

    def pack_rows_dedup(self, local_ids, order, uniq, B):
        """Row-gradient message with the duplicates of an id summed first (cffm_pack_rows_dedup): returns [B*F, 1+K+D+1]
        of which the first #distinct records are valid."""
        buf, _ = self.workspace(B)
        W = 1 + self.cfg.K + self.cfg.D + 1
        key = ('dedup', B)
        out = self._ws.get(key)
        if out is None:
            out = torch.empty((B * self.cfg.F, W), dtype=torch.float32, device=self.device)
            self._ws[key] = out
        hip.check(self.lib.cffm_pack_rows_dedup(self._s, _ptr(self._ids(local_ids.reshape(-1))), _ptr(order), _ptr(uniq), int(B),
                                                _ptr(buf), _ptr(out), self._stream()))
        return out

    def backward(self, y, B, B_global=None):
        buf, _ = self.workspace(B)
        hip.check(self.lib.cffm_backward(self._s, _ptr(self.theta), _ptr(y), int(B),
                                         int(B if B_global is None else B_global), _ptr(buf), _ptr(self.grad),
                                         self._stream()))

    # ---- data-parallel halves with late loss normalisation (cffm_amd/dist.py) ----------------------------------
    def backward_unscaled(self, ids, y, B, B_global, pack=True):
        """Backward with dL/dout = (out - y) / B_global.  Returns (grad_full [n+4] with this rank's loss-term sum at
        index n, rows [B*F, 1+K+D+1] = (id bits | dEi | dEo | dfb)) - the operands of ONE all-reduce and ONE
        all-gather.  pack=False leaves the row gradients in the workspace (rows is None): the row-sharded step packs
        them itself with the duplicates summed (pack_rows_dedup)."""
        ids = self._ids(ids)
        buf, _ = self.workspace(B)
        W = 1 + self.cfg.K + self.cfg.D + 1
        key = ('rows', B)
        rows = self._ws.get(key) if pack else None
        if rows is None and pack:
            rows = torch.empty((B * self.cfg.F, W), dtype=torch.float32, device=self.device)
            self._ws[key] = rows
        hip.check(self.lib.cffm_backward_unscaled(self._s, _ptr(self.theta), _ptr(ids), _ptr(y), int(B),
                                                  int(B_global), _ptr(buf), _ptr(self._grad_full), _ptr(rows),
                                                  self._stream()))
        return self._grad_full, rows

    def dp_local(self, ids, y, B, B_global):
        """forward + backward_unscaled in one library call (the local half of a data-parallel step).  Returns
        (grad_full, block): block is ONE flat buffer [B*F*(1+K+D+1) packed rows | B*F sorted 64-bit keys], the unit
        DataParallelStep all-gathers; cffm_dp_apply merges the per-rank sorted runs instead of sorting again."""
        ids = self._ids(ids)
        buf, _ = self.workspace(B)
        W = 1 + self.cfg.K + self.cfg.D + 1
        key = ('block', B)
        block = self._ws.get(key)
        if block is None:
            block = torch.empty(B * self.cfg.F * (W + 2), dtype=torch.float32, device=self.device)
            self._ws[key] = block
        hip.check(self.lib.cffm_dp_local(self._s, self._t, _ptr(self.theta), _ptr(ids), _ptr(y), int(B),
                                         int(B_global), _ptr(buf), _ptr(self._grad_full), _ptr(block), self._stream()))
        return self._grad_full, block

    def dp_apply(self, grad_full, rows_all, B_global, n_runs=0):
        """n_runs = 0: rows_all [n_rows, 1+K+D+1] in any order (cffm_backward_unscaled rows).  n_runs > 0: the flat
        concatenation of n_runs dp_local blocks."""
        W = 1 + self.cfg.K + self.cfg.D + 1
        # sorted runs: only where the single-launch forward left them, and while all ids fit the merge kernel's LDS
        runs_ok = n_runs > 0 and rows_all.dim() == 1 and (rows_all.numel() // (W + 2)) * 4 <= 150 * 1024 and \
            bool(self.lib.cffm_dp_runs_ok(self._s, int(rows_all.numel() // (W + 2) // n_runs // self.cfg.F)))
        if n_runs > 0 and not runs_ok:
            # the blocks carry no sorted runs (shape outside the single-launch forward): strip the key areas
            m = rows_all.numel() // (W + 2) // n_runs
            rows_all = rows_all.reshape(n_runs, m * (W + 2))[:, :m * W].reshape(n_runs * m, W).contiguous()
            n_runs = 0
        n_rows = rows_all.numel() // (W + 2) if n_runs > 0 else rows_all.shape[0]
        B_ws = max(1, -(-n_rows // self.cfg.F))       # an owner that received no rows still applies the dense update
        buf, _ = self.workspace(B_ws)
        hip.check(self.lib.cffm_dp_apply(self._s, self._t, self._ta,
                                         _ptr(self.theta), _ptr(self.theta_acc), _ptr(grad_full), int(B_global),
                                         _ptr(rows_all), int(n_rows), _ptr(buf), int(B_ws), _ptr(self.loss_buf),
                                         int(n_runs), self._stream()))
        return self.loss_buf

    # ---- dense-table exchange (small vocabularies): ONE all-reduce per step ----------------------------------------
    def dp_dense_ok(self, B, world):
        """The dense image of the tables is at most twice what the ranks' row gradients add up to (it also saves the second
        collective and the merge + segment walk of the gathered rows: measured 133 vs 152 us per step at world size 1,
        frappe), and the single-launch forward (which leaves the sorted keys the scatter needs) covers this shape."""
        W = self.cfg.K + self.cfg.D + 1
        return bool(self.lib.cffm_dp_runs_ok(self._s, int(B))) and \
            self.cfg.M * W <= 2 * world * B * self.cfg.F * (W + 3) and self.cfg.optimizer == 'AdagradOptimizer'

    def dp_local_dense(self, ids, y, B, B_global):
        ids = self._ids(ids)
        buf, _ = self.workspace(B)
        flat = self._ws.get('flat')
        if flat is None:
            flat = torch.zeros(int(self.lib.cffm_dp_dense_floats(self._s)), dtype=torch.float32, device=self.device)
            self._ws['flat'] = flat
        elif self._flat_dirty:
            flat.zero_()                     # a step was abandoned between the two halves: the image contract is zero on entry
        self._flat_dirty = True
        hip.check(self.lib.cffm_dp_local_dense(self._s, self._t, _ptr(self.theta), _ptr(ids), _ptr(y),
                                               int(B), int(B_global), _ptr(buf), _ptr(flat), self._stream()))
        return flat

    def dp_apply_dense(self, flat_sum, B_global):
        hip.check(self.lib.cffm_dp_apply_dense(self._s, self._t, self._ta,
                                               _ptr(self.theta), _ptr(self.theta_acc), _ptr(flat_sum), int(B_global),
                                               _ptr(self.loss_buf), self._stream()))
        own = self._ws.get('flat')
        if own is not None and flat_sum.data_ptr() == own.data_ptr():
            self._flat_dirty = False         # cffm_dp_apply_dense cleared the image it consumed (zero on exit)
        return self.loss_buf

    def apply_dense(self):
        hip.check(self.lib.cffm_dense_adagrad(_ptr(self.theta), _ptr(self.theta_acc), _ptr(self.grad),
                                              int(self.tl.n), float(self.cfg.lr), self._stream()))

    def apply_sparse(self, ids, dEi, dEo, dfb, B_ws):
        ids = self._ids(ids.reshape(-1))
        buf, _ = self.workspace(B_ws)
        hip.check(self.lib.cffm_sparse_adagrad(self._s, self._t, self._ta,
                                               _ptr(ids), int(ids.numel()), _ptr(dEi), _ptr(dEo), _ptr(dfb),
                                               _ptr(buf), int(B_ws), self._stream()))

    # ---- views the data-parallel step needs (cffm_amd/dist.py) ----------------------------------------
    def loss_sum_local(self, B):
        return self.ws_tensor(B, 'scalars', (16,))[0:1]

    def set_loss_sum_global(self, B, s):
        self.ws_tensor(B, 'scalars', (16,))[3:4].copy_(s)

    def loss_value(self, B):
        return self.ws_tensor(B, 'scalars', (16,))[1:2]

    def row_grads(self, B):
        F = self.cfg.F
        dEi = self.ws_tensor(B, 'dEi', (B, F, self.cfg.K)) if self.cfg.inner_conv else None
        dEo = self.ws_tensor(B, 'dEo', (B, F, self.cfg.D)) if self.cfg.outer_conv else None
        return dEi, dEo, self.ws_tensor(B, 'dfb', (B, F))

    def gather(self, ids, want_inner=True, want_outer=True, want_bias=True):
        ids = self._ids(ids)
        B, F = ids.shape
        dev = self.device
        Ei = torch.empty((B, F, self.cfg.K), dtype=torch.float32, device=dev) if want_inner else None
        Eo = torch.empty((B, F, self.cfg.D), dtype=torch.float32, device=dev) if want_outer else None
        fb = torch.empty((B, F), dtype=torch.float32, device=dev) if want_bias else None
        hip.check(self.lib.cffm_gather(self._s, self._t, _ptr(ids), B, _ptr(Ei), _ptr(Eo),
                                       _ptr(fb), self._stream()))
        return Ei, Eo, fb

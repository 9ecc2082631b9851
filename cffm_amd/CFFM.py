"""Drop-in for the reference's CFFM.py surface on top of the HIP engine.

Same command line (24 flags, CFFM.py:24-78), same constructor signature (CFFM.py:98-101), same
``train(data)`` / ``evaluate(data) -> (RMSE, R2)`` methods, same public metric lists and helper methods,
same log line formats (CFFM.py:174-179, :218-221, :553, :658-664, :684-695).  What is different underneath:

* the TensorFlow session is replaced by ``cffm_amd.engine.HipEngine`` (hand-written gfx950 kernels behind
  the C ABI of include/cffm_hip.h); there is no CPU fallback;
* the libfm splits are packed once into int32/fp32 tensors resident in HBM; the reference's per-sample
  Python batchers (CFFM.py:560-581, :617-629) reduce to slicing those tensors.  The batch COMPOSITION rule
  is the reference's: a contiguous block from ``np.random.randint(0, N - batch_size)`` after a per-epoch
  ``sklearn.utils.shuffle(..., random_state=2021)`` (quirk Q9), ordered blocks with a ragged last one in
  ``evaluate``;
* quirks that crash the reference are not reproduced: ``--tensorboard 1`` (Q6) is accepted and ignored with
  a warning, ``--pretrain 1`` (Q7) restores THIS model's tensors, no CUDA_VISIBLE_DEVICES pin (Q8).
"""
import argparse
import ast
import logging
import math
import os
from time import time

import numpy as np
from sklearn.utils import shuffle

from . import LoadData as DATA
from .spec import CFFMConfig, logged_param_count


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Run CFFM.")
    add = parser.add_argument
    add('--path', nargs='?', default='data/', help='Input data path.')
    add('--dataset', nargs='?', default='frappe', help='Choose a dataset.')
    add('--epoch', type=int, default=50, help='Number of epochs.')
    add('--pretrain', type=int, default=0,
        help='flag for pretrain. 1: initialize from pretrain; 0: randomly initialize; -1: save the model to pretrain file')
    add('--batch_size', type=int, default=1024, help='Batch size.')
    add('--inner_dims', type=int, default=32, help='Number of inner dimensions.')
    add('--outer_dims', type=int, default=32, help='Number of outer dimensions.')
    add('--lamda', type=float, default=0, help='Regularizer for bilinear part.')
    add('--keep', nargs='?', default='[1.0,1.0]', help='Keep probility (1-dropout) of each layer (parsed, unused).')
    add('--lr', type=float, default=0.05, help='Learning rate.')
    add('--loss_type', nargs='?', default='square_loss',
        help='Specify a loss type (square_loss or log_loss or mse or mae).')
    add('--optimizer', nargs='?', default='AdagradOptimizer', help='Specify an optimizer type (AdagradOptimizer).')
    add('--verbose', type=int, default=1, help='Show the results per X epochs (0, 1 ... any positive integer)')
    add('--batch_norm', type=int, default=0, help='Parsed, unused (as in the reference graph).')
    add('--tensorboard', type=int, default=0, help='Accepted and ignored (the reference crashes with 1).')
    add('--num_field', type=int, default=3,
        help='Valid dimension of the dataset. (e.g. frappe=10, ml-tag=3, book-crossing=6)')
    add('--linear_att', type=int, default=1, help='Linear attention part (0 disable or 1 enable)')
    add('--att_dim', type=int, default=0, help='Dimension of linear attention (0 is the same as num_field)')
    add('--lamda_att', type=float, default=1.0, help='Softmax temperature of the linear attention part')
    add('--inner_conv', type=int, default=1, help='Inner convolution part (0 disable or 1 enable)')
    add('--gamma_inner', type=int, default=1.0, help='Parsed, unused (as in the reference graph).')
    add('--outer_conv', type=int, default=1, help='Outer convolution part (0 disable or 1 enable)')
    add('--beta_outer', type=int, default=1.0, help='Weight of the outer convolution component')
    add('--activation', nargs='?', default='relu', help='Activation function (relu, prelu, elu, selu, gelu)')
    return parser.parse_args(argv)


def configure_logging(logFilename):
    logging.basicConfig(level=logging.DEBUG, format='%(asctime)s %(filename)s:%(message)s',
                        datefmt='%Y-%m-%d %A %H:%M:%S', filename=logFilename, filemode='a')
    console = logging.StreamHandler()
    console.setLevel(logging.INFO)
    console.setFormatter(logging.Formatter('%(asctime)s %(filename)s:%(message)s'))
    logging.getLogger().addHandler(console)


class CFFM(object):
    def __init__(self, features_M, pretrain_flag, save_file, inner_dims, outer_dims, loss_type, epoch, batch_size,
                 learning_rate, lamda_bilinear, keep, optimizer_type, batch_norm, verbose, tensorboard, num_field,
                 linear_att, att_dim, lamda_att, inner_conv, gamma_inner, outer_conv, beta_outer,
                 activation_function, random_seed=2021, batch_rng=None):
        self.batch_size = batch_size
        self.learning_rate = learning_rate
        self.inner_dims = inner_dims
        self.outer_dims = outer_dims
        self.pretrain_flag = pretrain_flag
        self.save_file = save_file
        self.loss_type = loss_type
        self.features_M = features_M
        self.lamda_bilinear = lamda_bilinear
        self.keep = keep
        self.epoch = epoch
        self.random_seed = random_seed
        self.optimizer_type = optimizer_type
        self.batch_norm = batch_norm
        self.verbose = verbose
        self.tensorboard = tensorboard
        self.num_field = num_field
        self.linear_att = linear_att
        self.att_dim = num_field if att_dim == 0 else att_dim
        if self.linear_att == 1 and self.att_dim != num_field:
            # the reference's matmul [B,F] x [att_dim,att_dim] (CFFM.py:432) only type-checks for att_dim == F
            raise ValueError('att_dim must equal num_field (or be 0)')
        self.lamda_att = lamda_att
        self.inner_conv = inner_conv
        self.gamma_inner = gamma_inner
        self.outer_conv = outer_conv
        self.beta_outer = beta_outer
        self.num_interactions = int(self.num_field * (self.num_field - 1) / 2)
        self.activation_function = activation_function
        if optimizer_type not in ('AdagradOptimizer', 'AdamOptimizer', 'GradientDescentOptimizer', 'MomentumOptimizer'):
            # the reference leaves self.optimizer unset for any other string and dies at the first sess.run (CFFM.py:517-529)
            raise ValueError('unknown optimizer %r' % (optimizer_type,))
        if loss_type == 'square_loss' and lamda_bilinear > 0:
            # create_loss regularises self.weights['inner_embeddings'] and ['outer_embeddings'] (CFFM.py:489-491), which
            # only exist for an enabled branch (CFFM.py:255, :262): the reference dies with this KeyError at graph build
            for flag, name in ((inner_conv, 'inner_embeddings'), (outer_conv, 'outer_embeddings')):
                if flag != 1:
                    raise KeyError(name)
        if tensorboard > 0:
            logging.warning('--tensorboard is accepted and ignored (it crashes the reference, CFFM.py:194-196)')
        self.config = CFFMConfig(M=features_M, F=num_field, K=inner_dims, D=outer_dims, activation=activation_function,
                                 lamda_att=lamda_att, beta_outer=beta_outer, linear_att=linear_att,
                                 inner_conv=inner_conv, outer_conv=outer_conv, loss_type=loss_type,
                                 lamda_bilinear=lamda_bilinear, lr=learning_rate, optimizer=optimizer_type)
        self.create_save_folder(save_file)
        self.train_rmse, self.valid_rmse, self.test_rmse = [], [], []
        self.train_r2, self.valid_r2, self.test_r2 = [], [], []
        self.engine = None
        self._packed = {}
        self.examples_per_sec = []
        # Source of the random block starts (CFFM.py:561 draws them from the unseeded process-global np.random; SURVEY A.6
        # Q9: "behind an injectable RNG").  Anything with numpy's randint(low, high, size=None) works; the default IS the
        # global np.random, so an unpinned run behaves like the reference.  batch_starts keeps the draws, one array per epoch.
        self.batch_rng = np.random if batch_rng is None else batch_rng
        self.batch_starts = []
        # multi-GPU (set by build_graph under torch.distributed): one process per GPU, data parallel over the batch
        self.world, self.rank, self._dp = 1, 0, None

    # ---- engine / data residency -----------------------------------------------------------------------
    def build_graph(self):
        """Creates the device state (the reference builds the TF graph here, CFFM.py:531-541)."""
        import torch
        # one process per GPU: the device is the launcher's LOCAL_RANK (torch.distributed.run) or the process's current
        # device; it is made current so that the library's launches and torch's stream agree on it
        dev = int(os.environ.get('LOCAL_RANK', torch.cuda.current_device() if torch.cuda.is_available() else 0))
        if torch.cuda.is_available():
            torch.cuda.set_device(dev)
        self.engine = self._make_engine(dev)
        if self.pretrain_flag > 0:
            self.load(self.save_file)
        self._setup_dist()
        return self.engine

    engine_factory = None          # tests plug a CPU stand-in (the oracle) in here; the product path is HipEngine

    def _make_engine(self, dev):
        if self.engine_factory is not None:
            return type(self).engine_factory(self.config, self.random_seed)
        from .engine import HipEngine
        return HipEngine(self.config, seed=self.random_seed, device='cuda:%d' % dev)

    def _setup_dist(self):
        """One process per GPU under torch.distributed.run (RANK / WORLD_SIZE / LOCAL_RANK in the environment), or a process
        group the caller initialised: the train step becomes cffm_amd.dist.DataParallelStep - every rank its slice of the
        SAME global batch, two collectives per step, replicas bit-identical (rank 0's parameters are broadcast at
        construction) - and evaluate() splits the rows over the ranks and all-reduces the three metric sums.  The reference
        is single-device (CFFM.py:19 pins one GPU), so this replaces nothing in it; at world size 1 nothing changes."""
        import torch
        import torch.distributed as dist
        if not dist.is_available():
            return
        if not dist.is_initialized():
            if int(os.environ.get('WORLD_SIZE', '1')) <= 1:
                return
            on_gpu = torch.cuda.is_available()
            dist.init_process_group('nccl' if on_gpu else 'gloo',
                                    device_id=self.engine.device if on_gpu else None)
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        if self.world == 1:
            return
        if os.environ.get('CFFM_TABLES', 'replicated') != 'replicated':
            raise NotImplementedError('CFFM.train() runs replicated tables (DataParallelStep); row-sharded tables are driven '
                                      'through cffm_amd.dist.ShardedStep (bench.py --tables sharded)')
        if self.batch_size % self.world:
            raise ValueError('--batch_size %d is not a multiple of the %d ranks' % (self.batch_size, self.world))
        from .dist import DataParallelStep
        self._dp = DataParallelStep(self.engine)

    def _info(self, msg):
        if self.rank == 0:
            logging.info(msg)

    def _all_reduce_sum(self, t):
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

    def _device_split(self, data):
        """{'X': lists, 'Y': list} -> (ids int32 [N,F], y fp32 [N], (min label, max label)) in HBM, packed once per split
        object (re-packed when the caller swaps the lists)."""
        import torch
        key = id(data)
        hit = self._packed.get(key)
        if hit is not None and hit[2] is self._token(data):
            return hit[0], hit[1], hit[3]
        X, Y = DATA.LoadData.packed(data)
        if X.shape[1] != self.num_field:
            raise ValueError('rows have %d ids, --num_field is %d' % (X.shape[1], self.num_field))
        dev = self.engine.device
        ids, y = torch.from_numpy(X).to(dev), torch.from_numpy(Y).to(dev)
        span = (float(Y.min()), float(Y.max())) if Y.size else (0.0, 0.0)      # clip range of evaluate(), CFFM.py:607-609
        self._packed[key] = (ids, y, self._token(data), span)
        return ids, y, span

    @staticmethod
    def _token(data):
        """What identifies the current content of a split: the loader's packed array while its lists have not been built
        or re-bound (cffm_amd.LoadData._Split), else the 'X' list object itself."""
        arrays = getattr(data, '_arrays', None)
        return arrays[0] if arrays is not None else data['X']

    # ---- training loop (CFFM.py:157-228) -------------------------------------------------------------
    def train(self, data):
        import torch
        if self.engine is None:
            self.build_graph()
        eng = self.engine
        self.calculate_parameters()
        if self.verbose > 0:
            t2 = time()
            init_train_rmse, init_train_r2 = self.evaluate(data.Train_data)
            init_valid_rmse, init_validation_r2 = self.evaluate(data.Validation_data)
            init_test_rmse, init_test_r2 = self.evaluate(data.Test_data)
            self._info(("Init_RMSE: train=%.4f,validation=%.4f,test=%.4f | Init_R2: train=%.4f,validation=%.4f,"
                        "test=%.4f [%.1f s] " % (init_train_rmse, init_valid_rmse, init_test_rmse, init_train_r2,
                                                init_validation_r2, init_test_r2, time() - t2)))
        ids, y, span = self._device_split(data.Train_data)
        n = ids.shape[0]
        # shuffle_in_unison_scary (CFFM.py:183, :556-558): sklearn's shuffle with the SAME random_state every epoch, i.e. one
        # fixed permutation of n positions applied to the CURRENT order each time.  It is built once and kept on the
        # device; `order` tracks the composition so that the caller's lists can be left as the reference leaves them.
        perm = shuffle(np.arange(n), random_state=self.random_seed)
        pt = torch.from_numpy(perm).to(ids.device)
        order = np.arange(n)
        try:
            for epoch in range(self.epoch):
                t1 = time()
                ids, y = ids[pt].contiguous(), y[pt].contiguous()
                order = order[perm]
                total_batch = int(n / self.batch_size)
                # CFFM.py:561: one np.random.randint per step, from self.batch_rng (the unseeded global np.random unless the
                # caller injected one).  Drawn for the whole epoch at once (the same stream as one call per step); under
                # torch.distributed rank 0's draws are everyone's, so that the ranks cut their slices out of the SAME
                # global batch.
                starts = self.batch_rng.randint(0, n - self.batch_size, size=total_batch)
                if self.world > 1:
                    import torch.distributed as dist
                    st = torch.from_numpy(starts.astype(np.int64)).to(ids.device)
                    dist.broadcast(st, src=0)
                    starts = st.cpu().numpy()
                self.batch_starts.append(np.asarray(starts, dtype=np.int64))
                per = self.batch_size // self.world
                for start in starts:
                    start = int(start)
                    if self._dp is not None:
                        lo = start + self.rank * per
                        self._dp.train_step(ids[lo:lo + per], y[lo:lo + per])
                    else:
                        eng.train_step(ids[start:start + self.batch_size], y[start:start + self.batch_size])
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                t2 = time()
                self.examples_per_sec.append(total_batch * self.batch_size / max(t2 - t1, 1e-9))
                self._packed[id(data.Train_data)] = (ids, y, self._token(data.Train_data), span)   # evaluate() sees the shuffled order
                train_rmse, train_r2 = self.evaluate(data.Train_data)
                valid_rmse, valid_r2 = self.evaluate(data.Validation_data)
                test_rmse, test_r2 = self.evaluate(data.Test_data)
                self.train_rmse.append(train_rmse)
                self.valid_rmse.append(valid_rmse)
                self.test_rmse.append(test_rmse)
                self.train_r2.append(train_r2)
                self.valid_r2.append(valid_r2)
                self.test_r2.append(test_r2)
                if self.verbose > 0 and epoch % self.verbose == 0:
                    self._info(("Epoch %d [%.1f s] RMSE: train=%.4f,validation=%.4f,Test=%.4f | R2: train=%.4f,"
                                "validation=%.4f,Test=%.4f [%.1f s]" % (epoch + 1, t2 - t1, train_rmse, valid_rmse,
                                                                        test_rmse, train_r2, valid_r2, test_r2,
                                                                        time() - t2)))
                    self._info("Epoch %d throughput: %.0f training examples/s%s" % (
                        epoch + 1, self.examples_per_sec[-1], ' (%d ranks)' % self.world if self.world > 1 else ''))
                if self.eva_termination(self.valid_rmse):
                    break
                if self.pretrain_flag < 0 and self.rank == 0:        # the replicas are identical: rank 0 writes
                    logging.info("Save model to file as pretrain.")
                    self.save(self.save_file)
        finally:
            # the reference re-binds data.Train_data['X'] / ['Y'] to the shuffled lists every epoch (CFFM.py:183); the device
            # copy is what the loop trains on, so the caller's lists are brought to the same (composed) order once, here
            if not np.array_equal(order, np.arange(n)):
                X0, Y0 = data.Train_data['X'], data.Train_data['Y']
                data.Train_data['X'] = [X0[i] for i in order]
                data.Train_data['Y'] = [Y0[i] for i in order]
                self._packed[id(data.Train_data)] = (ids, y, self._token(data.Train_data), span)

    # ---- evaluation (CFFM.py:583-615) ------------------------------------------------------------------
    def evaluate(self, data):
        """RMSE and R2 of the clipped predictions (CFFM.py:583-615).  The whole sweep stays on the device: ordered blocks
        with a ragged last one (CFFM.py:590-596; the forward is per example, so the block size does not change a
        prediction - blocks of >= 8192 rows run the conv kernels at 12 M examples/s against 3.7 M at 256), clip to the
        split's label range, and the three float64 sums of the metrics; three doubles come back to the host."""
        if self.engine is None:
            self.build_graph()
        ids, y, (lo, hi) = self._device_split(data)
        num_example = int(ids.shape[0])
        if num_example == 0:
            raise ValueError('evaluate() needs at least one example')
        if self.world > 1:
            # the forward is per example: every rank sweeps its contiguous share of the rows, the three sums are added up
            share = -(-num_example // self.world)
            r0, r1 = min(self.rank * share, num_example), min((self.rank + 1) * share, num_example)
            sums = self._all_reduce_sum(self.engine.eval_sums(ids[r0:r1], y[r0:r1], lo, hi, block=max(int(self.batch_size), 8192)))
        else:
            sums = self.engine.eval_sums(ids, y, lo, hi, block=max(int(self.batch_size), 8192))
        ss_res, sy, syy = (float(v) for v in sums.cpu().numpy())
        if not math.isfinite(ss_res):
            # np.maximum/np.minimum propagate NaN and sklearn's mean_squared_error raises on it (CFFM.py:607-612): a
            # diverged model must not come back with a finite metric
            raise ValueError('evaluate(): predictions contain NaN or infinity')
        RMSE = math.sqrt(ss_res / num_example)                  # sqrt(mean_squared_error), CFFM.py:610-612
        ss_tot = syy - sy * sy / num_example                    # sum (y - mean(y))^2
        R2 = 1.0 - ss_res / ss_tot if ss_tot > 0 else (1.0 if ss_res == 0 else 0.0)    # sklearn r2_score, CFFM.py:614
        return RMSE, R2

    def predict_split(self, data):
        """Raw (unclipped) predictions of a split as a host float64 array, in the split's current order."""
        import torch
        ids, _, _ = self._device_split(data)
        block = max(int(self.batch_size), 8192)
        outs = [self.engine.predict(ids[s:s + block]) for s in range(0, ids.shape[0], block)]
        return torch.cat(outs).cpu().numpy().astype(np.float64) if outs else np.zeros((0,))

    # ---- host-side helpers with the reference's list semantics (CFFM.py:556-635) -------------------------
    def shuffle_in_unison_scary(self, x, y):
        x_, y_ = shuffle(x, y, random_state=self.random_seed)
        return x_, y_

    def get_random_block_from_data(self, data, batch_size):
        """A block from a random start, filled forward over rows as long as the start row, then BACKWARD from the
        same start (which re-adds the start row when the forward fill stopped short) - CFFM.py:560-581."""
        start_index = self.batch_rng.randint(0, len(data['Y']) - batch_size)
        want = len(data['X'][start_index])
        X, Y = [], []
        for step in (1, -1):
            i = start_index
            while len(X) < batch_size and 0 <= i < len(data['X']) and len(data['X'][i]) == want:
                Y.append([data['Y'][i]])
                X.append(data['X'][i])
                i += step
        return {'X': X, 'Y': Y}

    def get_ordered_block_from_data(self, data, batch_size, index):
        start_index = index * batch_size
        X, Y = [], []
        i = start_index
        while len(X) < batch_size and i < len(data['X']) and len(data['X'][i]) == len(data['X'][start_index]):
            Y.append(data['Y'][i])
            X.append(data['X'][i])
            i += 1
        return {'X': X, 'Y': Y}

    def eva_termination(self, valid):
        if len(valid) > 5:
            if valid[-1] > valid[-2] > valid[-3] > valid[-4] > valid[-5]:
                return True
        return False

    def calculate_parameters(self):
        total_parameters = logged_param_count(self.config)
        if self.verbose > 0:
            self._info("#params: %d" % total_parameters)
        return total_parameters

    def create_save_folder(self, save_file):
        if not os.path.exists(save_file):
            os.makedirs(save_file)

    # ---- checkpoint: this model's tensors AND the optimizer slots (the reference restore is broken, Q7) ---------------
    # A plain dict of tensors and scalars: loads with torch.load(weights_only=True), no pickled code.
    def save(self, save_file):
        import torch
        t = lambda d: None if d is None else {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}
        cfg = {k: (v if isinstance(v, (int, float, str)) else float(v)) for k, v in self.config.__dict__.items()}
        torch.save({'format': 2, 'config': cfg, 'params': t(self.engine.export_params()),
                    'accumulators': t(self.engine.export_accumulators()),
                    'second_moments': t(self.engine.export_second_moments()), 'opt_step': int(self.engine.opt_step)},
                   save_file + '.pt')

    def load(self, save_file):
        import torch
        blob = torch.load(save_file + '.pt', weights_only=True)
        n = lambda d: None if d is None else {k: (v.numpy() if hasattr(v, 'numpy') else np.asarray(v)) for k, v in d.items()}
        saved = blob.get('config', {})
        for k in ('M', 'F', 'K', 'D'):
            if k in saved and int(saved[k]) != int(getattr(self.config, k)):
                raise ValueError('checkpoint %s.pt was written for %s=%s, this model has %s' % (save_file, k, saved[k], getattr(self.config, k)))
        self.engine.load_params(n(blob['params']), n(blob['accumulators']), n(blob.get('second_moments')))
        self.engine.opt_step = int(blob.get('opt_step', 0))


def main(argv=None):
    args = parse_args(argv)
    configure_logging('logging.log')
    if args.verbose > 0:
        logging.info(
            "CFFM: dataset=%s, factors=%d, loss_type=%s, #epoch=%d, batch=%d, lr=%.4f, lambda=%.1e, keep=%s, optimizer=%s"
            ", batch_norm=%d, num_field=%d, linear_att=%d, att_dim=%d,lamda_att=%.2f,inner_conv=%d,gamma_inner=%.1f,"
            "outer_conv=%d,beta_outer=%.1f, activation=%s"
            % (args.dataset, args.inner_dims, args.loss_type, args.epoch, args.batch_size, args.lr, args.lamda,
               ast.literal_eval(args.keep), args.optimizer, args.batch_norm, args.num_field, args.linear_att, args.att_dim,
               args.lamda_att, args.inner_conv, args.gamma_inner, args.outer_conv, args.beta_outer, args.activation))
    data = DATA.LoadData(args.path, args.dataset, args.loss_type)
    save_file = 'pretrain/CFFM/%s_%d/%s_%d' % (args.dataset, args.inner_dims, args.dataset, args.inner_dims)
    t1 = time()
    cf_fm = CFFM(data.features_M, args.pretrain, save_file, args.inner_dims, args.outer_dims, args.loss_type,
                 args.epoch, args.batch_size, args.lr, args.lamda, ast.literal_eval(args.keep), args.optimizer, args.batch_norm,
                 args.verbose, args.tensorboard, args.num_field, args.linear_att, args.att_dim, args.lamda_att,
                 args.inner_conv, args.gamma_inner, args.outer_conv, args.beta_outer, args.activation)
    cf_fm.train(data)
    best_valid_score = min(cf_fm.valid_rmse)
    best_epoch = cf_fm.valid_rmse.index(best_valid_score)
    logging.info("Best Iter of RMSE (validation)= %d train = %.4f, valid = %.4f, test = %.4f [%.1f s]"
                 % (best_epoch + 1, cf_fm.train_rmse[best_epoch], cf_fm.valid_rmse[best_epoch],
                    cf_fm.test_rmse[best_epoch], time() - t1))
    best_r2 = cf_fm.valid_r2.index(max(cf_fm.valid_r2))
    logging.info("Best Iter of R2 (validation)= %d train = %.4f, valid = %.4f, test = %.4f [%.1f s]"
                 % (best_epoch + 1, cf_fm.train_r2[best_r2], cf_fm.valid_r2[best_r2], cf_fm.test_r2[best_r2],
                    time() - t1))       # prints best_epoch + 1 from the RMSE search, as the reference does (Q15)
    return cf_fm


if __name__ == '__main__':
    main()

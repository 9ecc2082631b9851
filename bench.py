#!/usr/bin/env python3
"""bench.py - training examples/sec of the CFFM hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one fused forward + backward + Adagrad update (the reference's ``sess.run((loss, optimizer))``,
CFFM.py:200) over one batch of synthetic libfm rows of the frappe shape (10 fields, dim 32, batch 256 per
GPU, selu - README.md:28).  Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON
line; besides the contract keys it carries

  roofline      the embedding-gather kernel (the kernel BASELINE.json's metric names) on the 1M-feature
                stress shape (32 fields, dim 64, batch 8192, uniform ids: tables 516 MB > Infinity Cache),
                algorithmic bytes F*(K+D+1)*4 + F*4 per example / HIP-event time per launch, vs 8 TB/s
  stage_us      HIP-event time of every stage of the timed workload (where the step time goes)
  cpu_baseline  the numpy oracle (CPU restatement of the TF1 graph, kind "port") timed on the host cores
                on a bounded sample of the same workload
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from cffm_amd import hip, synth  # noqa: E402
from cffm_amd.engine import HipEngine  # noqa: E402
from cffm_amd.spec import CFFMConfig, init_params  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak
N_POOL = 64                    # distinct synthetic batches cycled through

# BASELINE.json configs (README.md:20-28 commands; configs[3] is the synthetic stress shape).  The default, and the
# only one the contract line is quoted on, is frappe; the others are there for profiling.
WORKLOADS = {
    'frappe': dict(M=5382, F=10, K=32, D=32, act='selu', B=256,
                   text='frappe shape: synthetic libfm, 10 fields, 5382 features, inner/outer dim 32, batch 256 per '
                        'GPU, selu, Adagrad lr 0.05, square_loss (README.md:28)'),
    'mltag': dict(M=90445, F=3, K=32, D=32, act='elu', B=1024,
                  text='ml-tag shape: 3 fields, 90445 features, dim 32, batch 1024 per GPU, elu (README.md:24)'),
    'bookx': dict(M=226336, F=6, K=32, D=32, act='relu', B=512,
                  text='book-crossing shape: 6 fields, 226336 features, dim 32, batch 512 per GPU, relu (README.md:20)'),
    'syn1m': dict(M=1000000, F=32, K=64, D=64, act='relu', B=8192,
                  text='synthetic stress shape: 32 fields, 1M features, dim 64, batch 8192 per GPU, relu'),
    'syn10m': dict(M=10000000, F=32, K=64, D=64, act='relu', B=8192,
                   text='BASELINE.json configs[4] per-GPU share: 32 fields, 10M features, dim 64, batch 8192 per GPU, relu '
                        '(run with --tables sharded: embedding rows sharded r -> rank r % G)'),
}
BIG = ('syn1m', 'syn10m')


def workload_cfg(name):
    w = WORKLOADS[name]
    return CFFMConfig(M=w['M'], F=w['F'], K=w['K'], D=w['D'], activation=w['act'], lr=0.05, lamda_att=1.0), w['B']


def frappe_cfg():
    return workload_cfg('frappe')[0]


def step_flops_per_example(cfg):
    """Reference-algorithm FLOPs of one train step per example, live layers only: 3 x forward."""
    P, D, K, Lc = cfg.P, cfg.D, cfg.K, cfg.Lc
    fwd = P * D * D + sum(2 * (D >> (l + 1)) ** 2 * 4 * P * P for l in range(Lc - 1)) + P * D * D + 7 * P * K \
        + 2 * (2 * D - 2) * 32
    return 3 * fwd


def measured_peaks(device):
    """What THIS box delivers on the two rooflines: a float4 streaming copy of 1 GiB (read + write counted) and a
    dependency-free fp32 MFMA loop on every CU (cffm_probe_copy / cffm_probe_mfma)."""
    lib = hip.load()
    n = 1 << 30
    src = torch.empty(n, dtype=torch.uint8, device=device).fill_(1)
    dst = torch.empty_like(src)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ms = event_time_ms(lambda: hip.check(lib.cffm_probe_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), n, st)), 20)
    copy_gbs = 2 * n / (ms * 1e-3) / 1e9
    sink = torch.zeros(4096, device=device)
    ms = event_time_ms(lambda: hip.check(lib.cffm_probe_read(C.c_void_p(src.data_ptr()), C.c_void_p(sink.data_ptr()), n, st)), 20)
    read_gbs = n / (ms * 1e-3) / 1e9
    del src, dst
    torch.cuda.empty_cache()
    out = torch.zeros(4, device=device)
    flops = C.c_int64(0)
    ms = event_time_ms(lambda: hip.check(lib.cffm_probe_mfma(C.c_void_p(out.data_ptr()), 20000, C.byref(flops), st)), 5)
    f32_tf = flops.value / (ms * 1e-3) / 1e12
    ms = event_time_ms(lambda: hip.check(lib.cffm_probe_mfma_bf16(C.c_void_p(out.data_ptr()), 20000, C.byref(flops), st)), 5)
    return {'hbm_copy_GBs': round(copy_gbs, 1), 'hbm_read_GBs': round(read_gbs, 1), 'mfma_f32_TFLOPs': round(f32_tf, 1),
            'mfma_bf16_TFLOPs': round(flops.value / (ms * 1e-3) / 1e12, 1)}


def event_time_ms(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def gather_roofline(device, with_stress=False):
    """The embedding gather on the stress shape of BASELINE.json configs[3] (32 fields, dim 64, 1 M features, 8192 rows,
    uniform ids: 516 MB of tables, beyond the Infinity Cache), timed with HIP events on the stream the kernels are launched
    on (torch's current stream).  achieved = ALGORITHMIC bytes (F*(K+D+1)*4 + F*4 per example) / time.

    gather_inner_fwd_wide_kernel   cffm_gather_inner_fwd: the PRODUCT-PATH gather of the wide shapes - the three lookups fused
                                   with the inner branch, the s0 pool and the first-order inputs; rows go HBM -> LDS ->
                                   registers and are never written back (what cffm_train_step / cffm_predict run at this
                                   shape).  This is the roofline kernel.
    gather_packed_kernel           cffm_gather_packed: owner side of a row-sharded lookup (materialises one packed record per id)
    gather_rows_kernel             cffm_gather: three separate outputs (the stage API)"""
    cfg, B = workload_cfg('syn1m')
    M, F, K, D = cfg.M, cfg.F, cfg.K, cfg.D
    lib = hip.load()
    eng = HipEngine(cfg, params='device', seed=2021, device=str(device))
    eng.fbias.normal_(0.0, 0.3)                       # exactly 0 at init (CFFM.py:276): give the copy check something to see
    ids = torch.from_numpy(synth.sample_ids(np.random.default_rng(2021), M, F, B * 8)).to(device).reshape(8, B, F)
    state = {'i': 0}

    def run_fused():
        i = state['i'] = (state['i'] + 1) % 8
        eng.gather_inner_fwd(ids[i])
    ms_fused = event_time_ms(run_fused, 40)
    fb = eng.ws_tensor(B, 'fb', (B, F))
    assert torch.equal(fb, eng.fbias[ids[state['i']].long()])
    io = eng.ws_tensor(B, 'inner_out', (B,))
    assert bool(torch.isfinite(io).all()) and float(io.abs().max()) > 0

    shape, tabs = eng.shape, eng.tables
    Ei = torch.empty((B, F, K), device=device)
    Eo = torch.empty((B, F, D), device=device)
    fbo = torch.empty((B, F), device=device)
    packed = torch.empty((B * F, K + D + 4), device=device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run_rows():
        i = state['i'] = (state['i'] + 1) % 8
        hip.check(lib.cffm_gather(C.byref(shape), C.byref(tabs), C.c_void_p(ids[i].data_ptr()), B,
                                  C.c_void_p(Ei.data_ptr()), C.c_void_p(Eo.data_ptr()), C.c_void_p(fbo.data_ptr()), st))

    def run_packed():
        i = state['i'] = (state['i'] + 1) % 8
        hip.check(lib.cffm_gather_packed(C.byref(shape), C.byref(tabs), C.c_void_p(ids[i].data_ptr()), B * F,
                                         C.c_void_p(packed.data_ptr()), st))
    ms_rows = event_time_ms(run_rows, 40)
    assert torch.equal(Eo[5], eng.outer[ids[state['i']][5].long()])
    ms_packed = event_time_ms(run_packed, 40)
    idl = ids[state['i']].reshape(-1).long()
    assert torch.equal(packed[:, K:K + D], eng.outer[idl]) and torch.equal(packed[:, K + D], eng.fbias[idl])
    bytes_per_launch = B * (F * (K + D + 1) * 4 + F * 4)
    achieved = bytes_per_launch / (ms_fused * 1e-3) / 1e9
    traffic, source = None, None      # HBM bytes per launch: rocprofv3 PMC passes of this command, committed under profiles/
    try:
        source = 'profiles/r04_gather_pmc.json'
        with open(os.path.join(ROOT, source)) as fh:
            traffic = int(json.load(fh)['gather_inner_fwd_wide_kernel']['hbm_bytes_per_launch'])
    except Exception:
        traffic, source = None, None
    del Ei, Eo, fbo, packed
    stress = stress_step(eng, cfg, ids[0], B) if with_stress else None
    del eng
    torch.cuda.empty_cache()
    rate = lambda ms: round(bytes_per_launch / (ms * 1e-3) / 1e9, 1)
    return stress, {'bound': 'hbm', 'kernel': 'gather_inner_fwd_wide_kernel', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS,
            'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic, 'traffic_source': source,
            'hbm_busy': None if not traffic else round(traffic / (ms_fused * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            'bytes_per_launch': bytes_per_launch, 'us_per_launch': round(ms_fused * 1e3, 2),
            'note': 'product-path kernel (cffm_train_step / cffm_predict at this shape): lookups fused with the inner branch, '
                    'nothing materialised; VALU-bound, see DESIGN.md; hbm_busy = traffic (PMC bytes) / time / peak',
            'gather_packed_kernel': {'us_per_launch': round(ms_packed * 1e3, 2), 'achieved': rate(ms_packed)},
            'gather_rows_kernel': {'us_per_launch': round(ms_rows * 1e3, 2), 'achieved': rate(ms_rows)},
            'workload': 'synthetic libfm 32 fields dim 64 1M features batch 8192 uniform ids (tables 516 MB)'}


def executed_flops_per_example(cfg):
    """FLOPs of one train step per example AS THE KERNELS RUN IT: conv layer 0 in its factorised form (rank-1 input channels:
    T[dh][i][x][q] = sum_{dw, j>i} E_j[2x+dw] W[dh,dw,(i,j),q] costs 8 S0 P^2, C = sum_{dh,i} E_i[2y+dh] T costs 4 F S0^2 P),
    layers >= 1 as direct contractions, 3 contractions per layer (forward, input gradient, weight gradient); unpadded."""
    P, D, F, live = cfg.P, cfg.D, cfg.F, cfg.Lc - 1
    S0 = D >> 1
    l0 = 8 * S0 * P * P + 4 * F * S0 * S0 * P
    rest = sum(2 * (D >> (l + 1)) ** 2 * 4 * P * P for l in range(1, live))
    return 3 * (l0 + rest)


def stress_step(eng, cfg, ids, B):
    """BASELINE.json configs[3] driver-timed: 1 warm-up + 3 timed cffm_train_step calls (forward + backward + Adagrad) at F32 D64
    M = 1M B = 8192 on the engine the roofline leg built, HIP events on the launch stream."""
    y = torch.from_numpy(synth.sample_labels(np.random.default_rng(7), B)).to(ids.device)
    ms = event_time_ms(lambda: eng.train_step(ids, y), 3, warm=1)
    loss = float(eng.loss_buf[0].item())
    if not np.isfinite(loss):
        raise SystemExit('bench.py: stress-shape loss is not finite')
    ex_tf = executed_flops_per_example(cfg) * B / (ms * 1e-3) / 1e12
    ref_tf = step_flops_per_example(cfg) * B / (ms * 1e-3) / 1e12
    return {'workload': WORKLOADS['syn1m']['text'], 'steps': 3, 'warmup': 1, 'ms_per_step': round(ms, 2),
            'examples_per_s': round(B / (ms * 1e-3), 1), 'executed_TFLOPs': round(ex_tf, 1),
            'executed_frac_of_mfma_f32_peak': round(ex_tf / MFMA_F32_PEAK_TFLOPS, 4),
            'contraction': 'fp32 MFMA loops (CFFM_CONV_FP32=1)' if os.environ.get('CFFM_CONV_FP32') else
                           'direct conv layers on the bf16 MFMA pipe at fp32 accuracy (error-free 3-way bf16 split of both operands, six '
                           'cross terms, fp32 accumulate; DESIGN.md 3.4): executed_TFLOPs is the fp32-EQUIVALENT rate; layer 0 (rank-1 '
                           'factorised) on fp32 MFMAs',
            'reference_algorithm_TFLOPs': round(ref_tf, 1), 'loss': round(loss, 6),
            'note': 'executed = conv layer 0 factorised (rank-1 input channels), layers >= 1 direct, 3 contractions per layer, '
                    'unpadded; peak = 157.3 TFLOP/s dense fp32 MFMA'}


def stage_times(eng, ids, y):
    """HIP-event time of every stage of one step at the bench workload (rank 0, diagnostic)."""
    lib, s, B = hip.load(), eng.shape, ids.shape[0]
    buf, wl = eng.workspace(B)
    P = lambda t: C.c_void_p(t.data_ptr())
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    sref, tabs, acc = C.byref(s), C.byref(eng.tables), C.byref(eng.tables_acc)
    w = buf.data_ptr()
    live = eng.tl.live
    stages = [
        ('gather', lambda: lib.cffm_gather(sref, tabs, P(ids), B, C.c_void_p(w + wl.Ei), C.c_void_p(w + wl.Eo),
                                           C.c_void_p(w + wl.fb), st())),
        ('inner_fwd', lambda: lib.cffm_inner_fwd(sref, P(eng.theta), P(buf), B, st())),
        ('conv0_fwd', lambda: lib.cffm_outer_conv0_fwd(sref, P(eng.theta), P(buf), B, st())),
    ]
    for l in range(1, live):
        stages.append(('conv%d_fwd' % l, lambda l=l: lib.cffm_conv_fwd(sref, P(eng.theta), P(buf), B, l, st())))
    stages.append(('head_fwd', lambda: lib.cffm_head_fwd(sref, P(eng.theta), P(buf), P(y), B, st())))
    stages.append(('head_bwd', lambda: lib.cffm_head_bwd(sref, P(eng.theta), P(buf), P(y), B, B, st())))
    for l in range(live - 1, 0, -1):
        stages.append(('conv%d_bwd' % l, lambda l=l: lib.cffm_conv_bwd(sref, P(eng.theta), P(buf), B, l, st())))
    stages.append(('conv0_bwd', lambda: lib.cffm_outer_conv0_bwd(sref, P(eng.theta), P(buf), B, st())))
    stages.append(('inner_bwd', lambda: lib.cffm_inner_bwd(sref, P(eng.theta), P(buf), B, st())))
    stages.append(('reduce_slabs', lambda: lib.cffm_reduce_slabs(sref, P(buf), B, P(eng.grad), st())))
    out = {}
    for name, fn in stages:              # in step order, so every stage sees valid inputs
        out[name] = round(event_time_ms(lambda: hip.check(fn()), 20) * 1e3, 2)
    return out


def cpu_baseline(cfg, X, y, budget_s=9.0):
    """Two CPU restatements of the TF1 graph (the reference itself needs TensorFlow 1.14 and cannot run here) on the
    same batches, fp32, on this box's host cores: the numpy oracle and the torch-CPU autograd twin (conv2d /
    max_pool2d / matmul with torch's intra-op thread pool, the closest stand-in for TF's Eigen pool).  The faster one
    is the reported value."""
    from oracle import cffm_oracle as orc
    from oracle import cffm_twin_torch as twin
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    p = init_params(cfg, seed=2021, dtype=np.float32)
    acc = orc.init_accumulators(p)
    orc.train_step(p, acc, X[0], y[0], cfg)          # warm-up (page-in, BLAS threads)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s and n < 200:
        orc.train_step(p, acc, X[(n + 1) % X.shape[0]], y[(n + 1) % X.shape[0]], cfg)
        n += 1
    dt = time.perf_counter() - t0
    numpy_rate = n * X.shape[1] / dt

    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    tthreads = max(1, min(ncpu, 64))
    torch.set_num_threads(tthreads)
    tp = {k: torch.tensor(np.asarray(v, dtype=np.float32), requires_grad=True)
          for k, v in init_params(cfg, seed=2021, dtype=np.float32).items()}
    tacc = {k: torch.full_like(v, 1e-8) for k, v in tp.items()}
    Xt, yt = torch.from_numpy(X).long(), torch.from_numpy(y)
    twin.train_step(tp, tacc, Xt[0], yt[0], cfg)
    m, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s and m < 400:
        twin.train_step(tp, tacc, Xt[(m + 1) % X.shape[0]], yt[(m + 1) % X.shape[0]], cfg)
        m += 1
    dt2 = time.perf_counter() - t0
    torch_rate = m * X.shape[1] / dt2
    cpu_model = ''
    try:
        with open('/proc/cpuinfo') as fh:
            for line in fh:
                if line.startswith('model name'):
                    cpu_model = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    best_torch = torch_rate >= numpy_rate
    return {'value': round(max(torch_rate, numpy_rate), 1), 'unit': 'examples/s',
            'cores': int(tthreads if best_torch else threads), 'kind': 'port',
            'sample': 'fp32 train steps on the same batches: torch-CPU twin %d steps in %.1f s = %.0f ex/s (%d threads); '
                      'numpy oracle %d steps in %.1f s = %.0f ex/s (%d BLAS threads); host has %d cores; value = the '
                      'faster; CPU: %s' % (m, dt2, torch_rate, tthreads, n, dt, numpy_rate, threads, os.cpu_count() or 0, cpu_model),
            'torch_cpu': round(torch_rate, 1), 'numpy_oracle': round(numpy_rate, 1)}


def self_launch(n):
    """Run this same command line under torch.distributed.run with n ranks on this node (127.0.0.1 rendezvous: the
    container hostname may not resolve) and return its exit status.  Nothing here initialises the GPU."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: RCCL needs it on this driver
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dist', default='uniform', choices=['uniform', 'zipf'])
    ap.add_argument('--workload', default='frappe', choices=sorted(WORKLOADS),
                    help='frappe is the contract workload; the others are for profiling')
    ap.add_argument('--tables', default='replicated', choices=['replicated', 'sharded'],
                    help='sharded: row-sharded tables (cffm_amd.dist.ShardedStep), the mode for vocabularies beyond one GPU')
    ap.add_argument('--force-dp', action='store_true',
                    help='run the data-parallel step (2 collectives) even at world size 1: what one rank of an N-GPU job executes')
    ap.add_argument('--graph', action='store_true',
                    help='data-parallel step as a HIP-graph replay (RCCL collectives captured); eager by default')
    ap.add_argument('--dp-mode', default='auto', choices=['auto', 'dense', 'gather'],
                    help='data-parallel exchange: dense image of the table gradients in one all-reduce (small vocabularies) or '
                         'all-gather of the row gradients; auto picks by size')
    ap.add_argument('--blocks', type=int, default=0,
                    help='timed blocks of --steps steps; the median block is reported (default 25, 1 for the big shapes)')
    ap.add_argument('--quick', action='store_true', help='skip stage times, roofline, peaks and the CPU baseline')
    args = ap.parse_args()

    if (args.gpus > 1 or os.environ.get('CFFM_BENCH_SELF_LAUNCH')) and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, as children (one process per GPU over RCCL), BEFORE
        # this process touches the GPU - a process that has initialised HIP must never be replaced or forked into ranks.
        # Rank 0 prints the JSON line on the inherited stdout; the exit status is the launcher's.  (CFFM_BENCH_SELF_LAUNCH=1
        # takes the same route at N = 1: how the route is checked on a one-GPU box.)
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # CFFM_BENCH_REHEARSAL=1 (debug): every rank on cuda:0 over gloo - how the N > 1 flow of this file (sharding of the batches, the
    # barriers, MAX over the ranks, the replica check, the JSON line) is rehearsed on a ONE-GPU box, where RCCL cannot be given
    # two devices.  Its numbers mean nothing (N processes share one GPU, the collectives go through the host).
    rehearsal = bool(os.environ.get('CFFM_BENCH_REHEARSAL'))
    if rehearsal:
        local_rank = 0
    elif torch.cuda.device_count() < max(world, 1):
        # fewer visible GPUs than ranks: every rank sees the same count (device_count() does not initialise the GPU on this
        # image; the parent of a self-launch never asks), so all of them leave with status 3 before any collective could hang,
        # rank 0 says why in one line, and the launcher hands the non-zero status back
        if rank == 0:
            sys.stderr.write('bench.py: --gpus %d needs %d visible GPUs, this node shows %d\n' % (args.gpus, world, torch.cuda.device_count()))
        raise SystemExit(3)
    use_pg = world > 1 or args.tables == 'sharded' or args.force_dp
    if use_pg:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29533')
        if rehearsal:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
    if args.gpus != world:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 through torch.distributed.run)'
                         % (args.gpus, world))
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)

    cfg, B = workload_cfg(args.workload)                # B per GPU: weak scaling, global batch = B * N
    big = args.workload in BIG
    n_pool = N_POOL if not big else 4
    Xh, yh = synth.batches(cfg.M, cfg.F, B, n_pool * world, seed=2021, dist=args.dist)
    Xh, yh = Xh[rank::world], yh[rank::world]            # every rank its own shard of every global batch
    X = torch.from_numpy(Xh).to(device)
    y = torch.from_numpy(yh).to(device)
    # vocabularies of >= 5 M rows: the tables are drawn on the device (a host draw + copy of 5 GB would dominate start-up)
    on_device = cfg.M >= 5000000

    if args.tables == 'sharded':
        import copy
        from cffm_amd.dist import ShardedStep, local_rows_count, shard_params
        lcfg = copy.copy(cfg)
        lcfg.M = local_rows_count(cfg.M, rank, world)
        if on_device:
            # ONE seed for the replicated dense parameters, a per-rank seed for this rank's shard of the tables
            eng = HipEngine(lcfg, params='device', seed=2021, table_seed=2021 + rank, device=str(device))
        else:
            eng = HipEngine(lcfg, params=shard_params(init_params(cfg, seed=2021), rank, world), device=str(device))
        sh = ShardedStep(eng, M_global=cfg.M)
        Xb, yb = [X[i] for i in range(n_pool)], [y[i] for i in range(n_pool)]
        # the routing plan of the next batch is issued one step ahead (its per-owner counts are on the host before needed)
        step = lambda i: sh.train_step(Xb[i % n_pool], yb[i % n_pool], next_ids=Xb[(i + 1) % n_pool])
        barrier = lambda: dist.barrier()
    elif world > 1 or args.force_dp:
        from cffm_amd.dist import DataParallelStep
        eng = HipEngine(cfg, params='device' if on_device else None, seed=2021, device=str(device))
        dp = DataParallelStep(eng, use_graph=args.graph, mode=args.dp_mode)
        step = lambda i: dp.train_step(X[i % n_pool], y[i % n_pool])
        barrier = lambda: dist.barrier()
    else:
        eng = HipEngine(cfg, params='device' if on_device else None, seed=2021, device=str(device))
        step = lambda i: eng.train_step(X[i % n_pool], y[i % n_pool])
        barrier = lambda: None

    # W untimed warm-up steps, then blocks of EXACTLY K timed steps, each bracketed by barrier + synchronize on both sides
    # and reduced with MAX over the ranks.  A block of 20 frappe steps is 2.4 ms, so one block decides nothing: the
    # reported figure is the MEDIAN block (all block times are in ms_per_step_blocks); the big shapes run one block.
    n_blocks = args.blocks if args.blocks > 0 else (1 if big else 25)
    for i in range(args.warmup):
        step(i)
    block_dt = []
    it = args.warmup
    for _ in range(n_blocks):
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(it + i)
        barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        it += args.steps
        if use_pg:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        block_dt.append(dt)
    dt = float(np.median(block_dt))
    loss = float(eng.loss_buf[0].item())
    if not np.isfinite(loss):
        raise SystemExit('bench.py: loss is not finite')
    agree = None
    if use_pg:                                           # every rank takes part: checksums of the replicated state, MIN == MAX
        from cffm_amd.dist import replicas_agree
        agree = replicas_agree(eng, tables=(args.tables != 'sharded'))
        if not agree:
            raise SystemExit('bench.py: the replicas diverged (rank %d)' % rank)
    if big:
        torch.cuda.empty_cache()

    if rank == 0:
        res = {
            'metric': 'training examples/sec, frappe 10-field dim32; embedding-gather %HBM roofline',
            'value': round(B * world * args.steps / dt, 1), 'unit': 'examples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
            'blocks': n_blocks, 'ms_per_step_blocks': [round(v / args.steps * 1e3, 4) for v in block_dt],
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': WORKLOADS[args.workload]['text'],
                       'global_batch': B * world, 'id_distribution': args.dist, 'tables': args.tables,
                       'parallelism': 'dp%d' % world if world > 1 else 'single'},
            'final_loss': round(loss, 6),
        }
        tf = step_flops_per_example(cfg) * B * world * args.steps / dt / 1e12
        res['step_flops'] = {'reference_algorithm_TFLOPs': round(tf, 2), 'mfma_f32_peak_TFLOPs': MFMA_F32_PEAK_TFLOPS * world,
                             'frac': round(tf / (MFMA_F32_PEAK_TFLOPS * world), 4),
                             'note': 'conv0 runs factorised (rank-1 input channels), so executed FLOPs are lower'}
        res['binding'] = hip.binding_name()
        if use_pg:
            res['dist'] = {'world_size': dist.get_world_size(), 'backend': dist.get_backend(),
                           'rccl_version': '.'.join(str(v) for v in torch.cuda.nccl.version()),
                           'replicas_bit_identical_after_run': agree}
            if args.tables == 'sharded':
                res['dist']['plans_built'], res['dist']['plans_reused'] = sh.plans_built, sh.plans_reused
        if world == 1 and not args.quick and args.tables == 'replicated' and not args.force_dp:
            # the same loop with the host-side batcher of CFFM.train in it (random start on the host, slice of the
            # device-resident split): SURVEY 8d's "second figure including host batching"
            Xall, yall = X.reshape(-1, cfg.F), y.reshape(-1)
            nrow = Xall.shape[0]
            rs = np.random.RandomState(2021)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nb = max(args.steps, 200) if not big else args.steps
            for i in range(nb):
                s0 = rs.randint(0, nrow - B)
                eng.train_step(Xall[s0:s0 + B], yall[s0:s0 + B])
            torch.cuda.synchronize()
            res['value_with_host_batching'] = round(B * nb / (time.perf_counter() - t0), 1)
            res['stage_us'] = stage_times(eng, X[0], y[0])
            del X, y
            res['stress'], res['roofline'] = gather_roofline(device, with_stress=True)
            res['roofline']['measured_peaks'] = measured_peaks(device)
            if not args.no_cpu_baseline and not big:
                res['cpu_baseline'] = cpu_baseline(cfg, Xh, yh)
        elif world > 1 and not args.quick and args.tables == 'replicated':
            # N > 1: the roofline object of the dominant gather kernel again, measured on rank 0's GPU once the timed blocks are
            # over (the other ranks wait at the final barrier; cpu_baseline is an N = 1 figure by contract)
            del X, y
            res['roofline'] = gather_roofline(device)[1]
        print(json.dumps(res), flush=True)
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

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
N_POOL = 64                    # distinct synthetic batches cycled through


def frappe_cfg():
    return CFFMConfig(M=5382, F=10, K=32, D=32, activation='selu', lr=0.05, lamda_att=1.0)


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


def gather_roofline(device):
    """cffm_gather on the stress shape of BASELINE.json configs[3]; timed with HIP events on the stream the
    kernel is launched on (torch's current stream)."""
    M, F, K, D, B = 1000000, 32, 64, 64, 8192
    lib = hip.load()
    shape = hip.Shape(M=M, F=F, K=K, D=D, act=0, linear_att=1, inner_conv=1, outer_conv=1, loss=0,
                      lamda_att=1.0, beta_outer=1.0, lr=0.05)
    g = torch.Generator(device=device).manual_seed(2021)
    inner = torch.randn((M, K), device=device, generator=g)
    outer = torch.randn((M, D), device=device, generator=g)
    fbias = torch.randn((M,), device=device, generator=g)
    tabs = hip.Tables(inner.data_ptr(), outer.data_ptr(), fbias.data_ptr())
    ids = torch.from_numpy(synth.sample_ids(np.random.default_rng(2021), M, F, B * 8)).to(device).reshape(8, B, F)
    Ei = torch.empty((B, F, K), device=device)
    Eo = torch.empty((B, F, D), device=device)
    fb = torch.empty((B, F), device=device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    state = {'i': 0}

    def run():
        i = state['i'] = (state['i'] + 1) % 8
        hip.check(lib.cffm_gather(C.byref(shape), C.byref(tabs), C.c_void_p(ids[i].data_ptr()), B,
                                  C.c_void_p(Ei.data_ptr()), C.c_void_p(Eo.data_ptr()), C.c_void_p(fb.data_ptr()), st))
    ms = event_time_ms(run, 40)
    assert torch.equal(Eo[5], outer[ids[state['i']][5].long()])
    bytes_per_launch = B * (F * (K + D + 1) * 4 + F * 4)
    achieved = bytes_per_launch / (ms * 1e-3) / 1e9
    traffic = None                     # HBM bytes per launch from rocprofv3 PMC passes of this same command (profiles/)
    try:
        with open(os.path.join(ROOT, 'profiles', 'r01_gather_pmc.json')) as fh:
            traffic = int(json.load(fh)['hbm_bytes_per_launch'])
    except Exception:
        pass
    del inner, outer, fbias, Ei, Eo, fb
    torch.cuda.empty_cache()
    return {'bound': 'hbm', 'kernel': 'gather_rows_kernel', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS,
            'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
            'bytes_per_launch': bytes_per_launch, 'us_per_launch': round(ms * 1e3, 2),
            'workload': 'synthetic libfm 32 fields dim 64 1M features batch 8192 uniform ids (tables 516 MB)'}


def stage_times(eng, ids, y):
    """HIP-event time of every stage of one step at the bench workload (rank 0, diagnostic)."""
    lib, s, B = eng.lib, eng.shape, ids.shape[0]
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


def cpu_baseline(cfg, X, y, budget_s=12.0):
    from oracle import cffm_oracle as orc
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
    return {'value': round(n * X.shape[1] / dt, 1), 'unit': 'examples/s', 'cores': int(threads), 'kind': 'port',
            'sample': '%d fp32 train steps of the numpy oracle (op-by-op restatement of the TF1 graph, materialises '
                      'the outer map) on the same batches, %.1f s; host has %d cores' % (n, dt, os.cpu_count() or 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dist', default='uniform', choices=['uniform', 'zipf'])
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    if args.gpus != world:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 through torch.distributed.run)'
                         % (args.gpus, world))
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)

    cfg = frappe_cfg()
    B = 256                                             # per GPU: weak scaling, global batch = 256 * N
    eng = HipEngine(cfg, seed=2021, device=str(device))
    Xh, yh = synth.batches(cfg.M, cfg.F, B, N_POOL * world, seed=2021, dist=args.dist)
    Xh, yh = Xh[rank::world], yh[rank::world]            # every rank its own shard of every global batch
    X = torch.from_numpy(Xh).to(device)
    y = torch.from_numpy(yh).to(device)

    if world > 1:
        import torch.distributed as dist
        from cffm_amd.dist import DataParallelStep
        dp = DataParallelStep(eng)
        step = lambda i: dp.train_step(X[i % N_POOL], y[i % N_POOL])
        barrier = lambda: dist.barrier()
    else:
        step = lambda i: eng.train_step(X[i % N_POOL], y[i % N_POOL])
        barrier = lambda: None

    for i in range(args.warmup):
        step(i)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(eng.loss_buf[0].item())
    if not np.isfinite(loss):
        raise SystemExit('bench.py: loss is not finite')

    if rank == 0:
        res = {
            'metric': 'training examples/sec, frappe 10-field dim32; embedding-gather %HBM roofline',
            'value': round(B * world * args.steps / dt, 1), 'unit': 'examples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'frappe shape: synthetic libfm, 10 fields, 5382 features, inner/outer dim 32, '
                                   'batch 256 per GPU, selu, Adagrad lr 0.05, square_loss (README.md:28)',
                       'global_batch': B * world, 'id_distribution': args.dist,
                       'parallelism': 'dp%d' % world if world > 1 else 'single'},
            'final_loss': round(loss, 6),
        }
        if world == 1:
            res['stage_us'] = stage_times(eng, X[0], y[0])
            res['roofline'] = gather_roofline(device)
            if not args.no_cpu_baseline:
                res['cpu_baseline'] = cpu_baseline(cfg, Xh, yh)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

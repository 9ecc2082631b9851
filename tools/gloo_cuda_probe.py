"""Does this torch build's gloo backend take device tensors?  Two processes on cuda:0 (debug probe for a world-2 rehearsal of the
data-parallel step on a one-GPU box)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def run(rank, world):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = '29577'
    dist.init_process_group('gloo', rank=rank, world_size=world)
    x = torch.full((1000,), float(rank + 1), device='cuda:0')
    for name, fn in (('all_reduce', lambda: dist.all_reduce(x)),
                     ('broadcast', lambda: dist.broadcast(x, 0)),
                     ('all_gather', lambda: dist.all_gather_into_tensor(torch.empty(2000, device='cuda:0'), x)),
                     ('all_to_all', lambda: dist.all_to_all_single(torch.empty(1000, device='cuda:0'), x))):
        try:
            fn()
            torch.cuda.synchronize()
            if rank == 0:
                print(name, 'ok', float(x[0]), flush=True)
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                print(name, 'FAILED', type(e).__name__, str(e)[:120], flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    mp.spawn(run, args=(2,), nprocs=2)

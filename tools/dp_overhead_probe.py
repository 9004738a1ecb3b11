"""Where does the data-parallel bookkeeping of the training step go?  One-rank RCCL group on a 1-GPU box: the whmr_train step with the
reducer active (always_bucket) vs inactive, and the step's DP pieces timed one by one (synchronised sections of a diagnostic step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29531')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1'); os.environ.setdefault('LOCAL_RANK', '0')
import torch
import torch.distributed as dist
import bench
from whmr_amd.parallel import grad_reducer as GR

dev = torch.device('cuda:0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=dev)


def timed(step, n=8, w=3):
    for _ in range(w):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


res = {}
for ab in (False, True):
    args = bench.parse(['--workload', 'whmr_train', '--no-cpu'] + (['--always-bucket'] if ab else []))
    with torch.enable_grad():
        step = bench.build_workload(args, dev)[0]
        res[ab] = timed(step)
    if ab:
        # pieces: wrap finish / broadcast_buffers / pack with synchronised timers
        acc = {}
        def wrap(obj, name, key):
            orig = getattr(obj, name)
            def f(*a, **k):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                r = orig(*a, **k)
                torch.cuda.synchronize(); acc[key] = acc.get(key, 0.0) + (time.perf_counter() - t0) * 1e3
                return r
            setattr(obj, name, f)
        wrap(GR.GradReducer, 'finish', 'finish'); wrap(GR.GradReducer, '_pack_and_launch', 'pack+launch')
        import whmr_amd.parallel as P
        wrap(P, 'broadcast_buffers', 'broadcast_buffers'); wrap(bench, 'broadcast_buffers', 'bb') if hasattr(bench, 'broadcast_buffers') else None
        with torch.enable_grad():
            step(); acc.clear(); step()
        print('synchronised pieces of one step (ms):', {k: round(v, 3) for k, v in acc.items()})
print('whmr_train step: reducer inactive %.2f ms, active on a one-rank RCCL group %.2f ms' % (res[False], res[True]))
dist.destroy_process_group()

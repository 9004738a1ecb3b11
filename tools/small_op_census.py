"""Who issues the small torch ops (fills, copies, casts) inside one W-HMR training step?  Patches the Python entry points (torch.zeros / zeros_like /
cat / stack, Tensor.contiguous / clone / float / to / copy_ / zero_ / fill_ / __setitem__ ...) and counts calls per calling line inside the repo;
`contiguous` / `float` / `to` only count when they really copy.  Runs on CPU-less boxes only with a GPU (it drives bench.py's whmr_train step)."""
import os
import sys
import traceback
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
counts, active = Counter(), [False]


def caller():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if ROOT in fr.filename and 'tools/' not in fr.filename:
            return '%s:%d' % (os.path.relpath(fr.filename, ROOT), fr.lineno)
    return '?'


def wrap(owner, name, copies=None):
    orig = getattr(owner, name)

    def f(*a, **k):
        out = orig(*a, **k)
        if active[0] and (copies is None or copies(a, out)):
            counts[(name, caller())] += 1
        return out
    setattr(owner, name, f)


is_copy = lambda a, out: isinstance(out, torch.Tensor) and isinstance(a[0], torch.Tensor) and (out.data_ptr() != a[0].data_ptr() or out.dtype != a[0].dtype)
for n in ('zeros', 'zeros_like', 'ones', 'ones_like', 'cat', 'stack', 'full', 'arange', 'tensor'):
    wrap(torch, n)
for n in ('contiguous', 'float', 'to', 'bfloat16'):
    wrap(torch.Tensor, n, is_copy)
for n in ('clone', 'copy_', 'zero_', 'fill_', '__setitem__', 'repeat_interleave', 'expand_as', 'flip'):
    wrap(torch.Tensor, n)

args = bench.parse(['--workload', 'whmr_train', '--no-cpu'])
step = bench.build_workload(args, torch.device('cuda:0'))[0]
for _ in range(3):
    step()
torch.cuda.synchronize()
active[0] = True
step()
active[0] = False
print('%d patched-op calls in one step' % sum(counts.values()))
for (name, where), n in counts.most_common(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    print('%4d  %-16s %s' % (n, name, where))

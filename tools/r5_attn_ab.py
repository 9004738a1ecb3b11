"""Round 5 A/B on ONE box, interleaved: the persistent 16-row-tile attention kernels (bf16 and split-bf16) against the round-2 / round-3 kernels they
replaced, inside the headline workload (ViT-B/16 224^2, batch 64) and at 256x192.   python tools/r5_attn_ab.py"""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
from whmr_amd import _lib as L
dev = torch.device('cuda:0')
for workload, numerics in (('vit224', 'bf16'), ('vit256x192', 'bf16'), ('vit224', 'bf16x3'), ('vit256x192', 'bf16x3')):
    args = bench.parse(['--workload', workload, '--numerics', numerics, '--no-cpu', '--no-secondary'])
    with torch.no_grad():
        step, _, _, _ = bench.build_workload(args, dev)
        res = {'new': [], 'old': []}
        for rnd in range(3):
            for name in ('old', 'new'):
                L.attention_set_variant(1 | (16 if name == 'old' else 0))
                L.attention_x3_set_variant(1 if name == 'old' else 0)
                res[name].append(bench.time_steps(step, 30, 5))
        L.attention_set_variant(1)
        L.attention_x3_set_variant(0)
    print('%s %s: round-2/3 attention kernels %s ms, persistent 16-row-tile kernels %s ms' % (
        workload, numerics, ' '.join('%.3f' % v for v in res['old']), ' '.join('%.3f' % v for v in res['new'])))

"""Does replaying the headline ViT forward from ONE HIP graph shorten the 3.7-4.8 us gaps between its dependent kernels?  Interleaved eager / graph timing
on one box (bench.py's own workload and timer).   python tools/r5_vit_graph_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
dev = torch.device('cuda:0')
for workload, numerics in (('vit224', 'bf16'), ('vit224', 'bf16x3')):
    args = bench.parse(['--workload', workload, '--numerics', numerics, '--no-cpu', '--no-secondary'])
    with torch.no_grad():
        step, _, _, _ = bench.build_workload(args, dev)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        res = {'eager': [], 'graph': []}
        for rnd in range(3):
            res['eager'].append(bench.time_steps(step, 30, 5))
            res['graph'].append(bench.time_steps(g.replay, 30, 5))
    print('%s %s: eager %s ms, one HIP graph %s ms' % (workload, numerics, ' '.join('%.3f' % v for v in res['eager']), ' '.join('%.3f' % v for v in res['graph'])))

import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from oracle import synth
from oracle.vit import vit_forward
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
sd = synth.make_vit_state(1, (224, 224))
x = synth.make_inputs(16, 7, (224, 224))['x']
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    with torch.no_grad():
        vit_forward(sd, x[:2])
        t0 = time.perf_counter(); vit_forward(sd, x); dt = time.perf_counter() - t0
    print('threads %d: B=16 %.2f s  %.2f img/s' % (th, dt, 16 / dt))

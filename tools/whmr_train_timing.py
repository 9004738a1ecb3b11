"""Full W-HMR training step on one GPU (BASELINE configs[3] per-GPU share: batch 64, 256x192 crops, bf16 numerics, TRAIN.STAGE 2):
WHMR.forward(is_train=True) -> synthetic loss on the outputs core/trainer.py:500-600 supervises -> backward through the HIP autograd
nodes -> GradReducer.finish() (no optimizer).  `python tools/whmr_train_timing.py [B] [steps] [numerics]`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from whmr_amd.models import whmr_net
from whmr_amd.parallel import GradReducer
from whmr_amd.utils import synth

LOSS_KEYS = ('rotmat', 'pred_shape', 'pred_cam', 'kp_2d', 'kp_2d_w', 'kp_3d', 'verts', 'sub_verts', 'temp_verts')


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    numerics = sys.argv[3] if (len(sys.argv) > 3 and not sys.argv[3].startswith('--')) else 'bf16'
    dev = torch.device('cuda:0')
    assets = synth.make_assets(0)
    sd = synth.make_state_dict(0, assets, with_cam_model=False)
    m = whmr_net(None, assets=assets, numerics=numerics)
    m.load_state_dict(sd, strict=False)
    m = m.to(dev).train()
    for name, p in m.named_parameters():           # frozen / not in the loss: cam_model (detached in the reference), dp_head, global_orient
        if name.startswith(('cam_model', 'global_orient')):
            p.requires_grad_(False)
    params = [p for p in m.parameters() if p.requires_grad]
    use_graph = '--graph' in sys.argv
    red = None if use_graph else GradReducer(params)   # (its hooks keep the AccumulateGrad nodes of the first eager step alive)
    opt = torch.optim.Adam(params, lr=5e-5, fused=True) if '--adam' in sys.argv else None     # core/trainer.py:110-114
    inp = synth.make_inputs(B, 0)
    d = {k: inp[k].to(dev) for k in ('x', 'center', 'scale', 'bbox_height', 'orig_shape', 'bbox_info')}

    def step():
        for p in params:
            p.grad = None
        out, _ = m(d['x'], None, d['center'], d['scale'], d['bbox_height'], d['orig_shape'], d['bbox_info'], is_train=True)
        loss = 0.0
        for l in range(1, 4):
            for k in LOSS_KEYS:
                loss = loss + out['smpl_out'][l][k].float().pow(2).mean()
        for v in (out['dp_out'][0].values() if out['dp_out'] else ()):     # IUV head (AUX_SUPV_ON, the yaml default): core/trainer.py:466-480
            loss = loss + v.pow(2).mean()
        loss.backward()
        if red is not None:
            red.finish()
        if opt is not None and not use_graph:
            opt.step()
        return loss

    if not use_graph:
        for _ in range(2):
            loss = step()
        torch.cuda.synchronize()
    if use_graph:      # whole step captured once and replayed (whmr_amd.train.capture_train_step)
        from whmr_amd.train import capture_train_step
        replay, static_loss = capture_train_step(m, step)
        step = (lambda: (replay(), opt.step(), static_loss)[2]) if opt is not None else (lambda: (replay(), static_loss)[1])
        for _ in range(2):
            step()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    flops = 3 * (34.20e9 + 9.26e9 + 1.98e9 + 5.10e9) * B          # ViT + deconvs + Tz conv + IUV head (2*12288*2304*90)
    fmt = ('W-HMR train step B=%d %s' + (' [HIP graph replay]' if use_graph else '') + ' (forward + backward + gradient buckets' + (' + fused Adam' if opt is not None else ', no optimizer') + '): '
           '%.2f ms  %.0f img/s  %.0f TFLOP/s (3 x forward FLOPs of ViT + deconvs + Tz conv)  loss %.4f  peak memory %.1f GB')
    print(fmt % (B, numerics, dt * 1e3, B / dt, flops / dt / 1e12, float(loss.detach()), torch.cuda.max_memory_allocated() / 2 ** 30))
    if use_graph:
        return
    # forward / backward split
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for p in params:
        p.grad = None
    e[0].record()
    out, _ = m(d['x'], None, d['center'], d['scale'], d['bbox_height'], d['orig_shape'], d['bbox_info'], is_train=True)
    loss = sum(out['smpl_out'][l][k].float().pow(2).mean() for l in range(1, 4) for k in LOSS_KEYS)
    loss = loss + sum(v.pow(2).mean() for v in (out['dp_out'][0].values() if out['dp_out'] else ()))
    e[1].record()
    loss.backward()
    e[2].record()
    torch.cuda.synchronize()
    print('  forward %.2f ms, backward %.2f ms' % (e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])))


if __name__ == '__main__':
    main()

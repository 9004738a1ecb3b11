"""CPU suite: C-ABI library exports, host-side module logic, config object, loud failure without a GPU, gloo sharding."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

from conftest import ROOT


def test_library_builds_and_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from whmr_amd import _lib as L
    lib = ctypes.CDLL(L.LIB_PATH)
    hdr = open(os.path.join(ROOT, 'include', 'whmr_hip.h')).read()
    declared = set(re.findall(r'^\s*int\s+(whmr_\w+)\s*\(', hdr, flags=re.M))
    assert declared and declared == set(L.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_no_kernel_holds_the_packed_fp32_form_that_went_wrong():
    """Round 6 (profiles/r06_coresidency_probe.txt): ``v_pk_{fma,mul,add}_f32 ... op_sel:[...]`` (a lane reading the HIGH register of a source pair) returned
    wrong low lanes while one MFMA kernel ran on another stream.  The files whose kernels held that form are built without packed-fp32 instructions
    (w-hmr_amd/build.py); this disassembles every built code object and finds the form nowhere but in the canary that demonstrates it (ceilings.hip)."""
    import glob
    import subprocess
    import __graft_entry__ as ge
    ge.build()
    objdump = '/opt/rocm/lib/llvm/bin/llvm-objdump'
    if not os.path.exists(objdump):
        pytest.skip('no llvm-objdump in this image')
    objs = sorted(glob.glob(os.path.join(ROOT, 'w-hmr_amd', 'build', '*.o')))
    assert len(objs) >= 20
    found = {}
    for o in objs:
        blob = open(o, 'rb').read()
        starts = [m.start() for m in re.finditer(b'\x7fELF', blob)]
        assert len(starts) >= 2, o                                              # host object + the embedded gfx950 code object
        co = os.path.join(ROOT, 'w-hmr_amd', 'build', '_scan.co')
        with open(co, 'wb') as f:
            f.write(blob[starts[1]:])
        text = subprocess.run([objdump, '-d', '--mcpu=gfx950', co], capture_output=True, text=True, check=True).stdout
        os.remove(co)
        n = len(re.findall(r'v_pk_(?:fma|mul|add)_f32[^\n]*op_sel:\[', text))
        if n:
            found[os.path.basename(o)] = n
    assert found == {'ceilings.o': 1}, found


def test_product_path_never_imports_the_oracle():
    for dp, _, files in os.walk(os.path.join(ROOT, 'w-hmr_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), os.path.join(dp, f)


def test_no_cpu_fallback(assets, state_dict):
    from whmr_amd.models import whmr_net
    from whmr_amd.models.pose_vit import ViT
    from whmr_amd.utils import geometry as G
    with pytest.raises(RuntimeError):
        ViT(img_size=(256, 192), depth=1, qkv_bias=True)(torch.zeros(1, 3, 256, 192))
    with pytest.raises(RuntimeError):
        G.batch_rodrigues(torch.zeros(2, 3))
    m = whmr_net(None, assets=assets)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 256, 192))
    with pytest.raises(RuntimeError):                  # the training graph is HIP-only as well
        m(torch.zeros(1, 3, 256, 192), is_train=True)
    from whmr_amd.train import DeconvBNReLUFn, LinearFn, MAFSampleFn, SMPLFn
    with pytest.raises(RuntimeError):
        LinearFn.apply(torch.zeros(2, 4), torch.zeros(3, 4), None)
    with pytest.raises(RuntimeError):
        DeconvBNReLUFn.apply(torch.zeros(1, 2, 2, 64), torch.zeros(64, 64, 4, 4), torch.ones(64), torch.zeros(64), torch.nn.BatchNorm2d(64),
                             torch.float32)


def test_ddp_buffer_ignore_list_keeps_the_trained_batchnorm_statistics(assets):
    """the module tells torch's DistributedDataParallel (core/trainer.py:84-86) which buffers NOT to re-broadcast before every forward: the constant
    tables (SMPL arrays, dense down-sampling matrices, mean parameters, point grid) and the frozen camera network's statistics; the statistics of the
    four BatchNorm layers that train stay in the broadcast.  The list survives torch's SyncBatchNorm conversion (the other reference line, :83)."""
    from whmr_amd.models import whmr_net
    m = whmr_net(None, assets=assets)
    ignore = set(m._ddp_params_and_buffers_to_ignore)
    names = {n for n, _ in m.named_buffers()}
    assert ignore <= names and not any(n in ignore for n, _ in m.named_parameters())
    kept = sorted(names - ignore)
    assert kept and all(('running_' in n or 'num_batches_tracked' in n) and not n.startswith('cam_model.') for n in kept), kept
    assert {n.split('.running_')[0] for n in kept if '.running_mean' in n} == {'deconv_layers.1', 'deconv_layers.4', 'deconv_layers.7', 'est_Tz.2'}
    for n in ('points_grid', 'regressor.0.Dmap0', 'regressor.2.init_pose', 'global_orient.init_pose'):
        assert n in ignore, n
    big = sum(b.numel() * b.element_size() for n, b in m.named_buffers() if n in ignore)
    assert big > 50e6, big                                           # what a DDP forward would otherwise copy twice per step
    m2 = torch.nn.SyncBatchNorm.convert_sync_batchnorm(m)
    assert m2 is m and set(m2._ddp_params_and_buffers_to_ignore) == ignore
    assert {n for n, _ in m2.named_buffers()} == names               # the conversion keeps the buffer names


def test_state_dict_contract(assets, state_dict):
    """every own-code key of the reference state_dict exists with the right shape, strict on our side"""
    from whmr_amd.models import whmr_net
    m = whmr_net(None, assets=assets)
    own = m.state_dict()
    for k, v in state_dict.items():
        assert k in own and tuple(own[k].shape) == tuple(v.shape), k
    extra = [k for k in own if k not in state_dict]
    assert all('.smpl.' in k for k in extra), extra                  # third-party (smplx) buffer names only
    res = m.load_state_dict(state_dict, strict=False)
    assert not res.unexpected_keys
    assert torch.equal(m.points_grid, state_dict['points_grid'])
    assert torch.allclose(m.regressor[1].init_pose, state_dict['regressor.1.init_pose'], atol=1e-7)
    assert tuple(m.feature_extractor.backbone.pos_embed.shape) == (1, 193, 768)


def test_reference_checkpoint_adapter(assets, state_dict):
    from whmr_amd.models import whmr_net
    from whmr_amd.models.whmr import load_reference_state_dict
    m = whmr_net(None, assets=assets)
    ckpt = dict(state_dict)
    ckpt['regressor.0.vertex_joint_selector.extra_joints_idxs'] = torch.zeros(21, dtype=torch.long)   # smplx-internal key
    ckpt['regressor.1.smpl.betas'] = torch.zeros(1, 10)
    missing, unexpected, skipped = load_reference_state_dict(m, ckpt, verbose=False)
    assert not unexpected and len(skipped) == 2 and all('.smpl.' in k for k in missing)
    bad = dict(state_dict)
    bad['regressor.0.fc1.weight'] = torch.zeros(3, 3)
    with pytest.raises(KeyError):
        load_reference_state_dict(m, bad, verbose=False)


def test_strict_load_of_a_reference_shaped_checkpoint(assets, state_dict):
    """demo/tester.py:64-65: ``model.load_state_dict(ckpt['model'], strict=True)`` unchanged, on a checkpoint that also carries the
    smplx / pare internals of the reference's Regressor (regressor.N.smpl.*, regressor.N.vertex_joint_selector.*)"""
    from whmr_amd.models import whmr_net
    m = whmr_net(None, assets=assets)
    ckpt = {k: v.clone() for k, v in state_dict.items()}
    for n in range(3):
        ckpt['regressor.%d.smpl.betas' % n] = torch.zeros(1, 10)
        ckpt['regressor.%d.smpl.global_orient' % n] = torch.zeros(1, 3)
        ckpt['regressor.%d.smpl.faces_tensor' % n] = torch.zeros(13776, 3, dtype=torch.long)
        ckpt['regressor.%d.smpl.vertex_joint_selector.extra_joints_idxs' % n] = torch.zeros(21, dtype=torch.long)
        ckpt['regressor.%d.vertex_joint_selector.extra_joints_idxs' % n] = torch.zeros(21, dtype=torch.long)
    ckpt['regressor.1.fc1.weight'] = ckpt['regressor.1.fc1.weight'] + 1.0
    res = m.load_state_dict(ckpt, strict=True)                      # must not raise
    assert len(m.ignored_checkpoint_keys) == 15 and not res.unexpected_keys
    assert torch.equal(m.regressor[1].fc1.weight, ckpt['regressor.1.fc1.weight'])
    # still strict about own-code keys
    bad = dict(ckpt)
    del bad['regressor.0.fc2.bias']
    with pytest.raises(RuntimeError, match='missing own-code keys'):
        m.load_state_dict(bad, strict=True)
    bad = dict(ckpt)
    bad['regressor.0.not_a_real_key'] = torch.zeros(1)
    with pytest.raises(RuntimeError, match='unexpected keys'):
        m.load_state_dict(bad, strict=True)
    bad = dict(ckpt)
    bad['deconv_layers.0.weight'] = torch.zeros(3, 3)
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad, strict=True)


def test_cfg_object():
    from whmr_amd.core.cfgs import CfgNode, cfg
    assert cfg.MODEL.PyMAF.MLP_DIM == [256, 128, 64, 32] and cfg.IMG_RES.WIDTH == 256 and cfg.TRAIN.STAGE == 2
    c = cfg.clone()
    c.merge_from_list(['TRAIN.BATCH_SIZE', '8', 'MODEL.PyMAF.AUX_SUPV_ON', 'False'])
    assert c.TRAIN.BATCH_SIZE == 8 and c.MODEL.PyMAF.AUX_SUPV_ON is False and cfg.TRAIN.BATCH_SIZE == 64
    assert isinstance(CfgNode({'a': {'b': 1}}).a, CfgNode)
    assert 'PyMAF' in c.dump()


def test_gemm_wrapper_rejects_host_tensors():
    from whmr_amd import _lib as L
    with pytest.raises(RuntimeError):
        L.gemm(torch.zeros(4, 64), torch.zeros(128, 64), torch.zeros(4, 128))


def test_bench_sharding_two_ranks_gloo(tmp_path):
    """N>1 path of bench.py (rank/world bookkeeping, barrier, max-over-ranks, whole-job aggregate) on CPU/gloo, world 2"""
    code = r'''
import os, sys, json, torch, torch.distributed as dist
sys.path.insert(0, %r)
import bench
dist.init_process_group('gloo')
t = bench.reduce_max_time(0.5 + dist.get_rank(), dist, torch.device('cpu'))
v = bench.aggregate_value(dist.get_world_size(), 64, 10, t)
if dist.get_rank() == 0:
    print(json.dumps({'t': t, 'v': v, 'shard': bench.shard_batch(256, dist.get_world_size(), 1)}))
dist.destroy_process_group()
''' % ROOT
    script = tmp_path / 'w.py'
    script.write_text(code)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                        '127.0.0.1', '--master-port', '29533', str(script)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert out['t'] == 1.5 and abs(out['v'] - 2 * 64 * 10 / 1.5) < 1e-9 and out['shard'] == [128, 256]


def test_grad_reducer_two_ranks_gloo(tmp_path):
    """SURVEY 8e training row: bucketed gradient all-reduce(mean), world size 2 on CPU/gloo, against the single-process gradient."""
    import json
    script = tmp_path / 'ddp.py'
    script.write_text('''
import json, os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from whmr_amd.parallel import GradReducer, shard_batch, broadcast_buffers
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
torch.manual_seed(0)
model = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.GELU(), torch.nn.Linear(32, 8), torch.nn.LayerNorm(8))
x, y = torch.randn(12, 16), torch.randn(12, 8)
# single-process reference on the whole batch (mean loss == mean of the per-shard mean losses for equal shards)
ref = [torch.autograd.grad(((model(x) - y) ** 2).mean(), model.parameters())]
red = GradReducer(model.parameters(), bucket_bytes=1200)          # several small buckets: exercises the bucket bookkeeping
lo, hi = shard_batch(12, world, rank)
for step in range(2):                                             # twice: buckets are re-armed after finish()
    model.zero_grad(set_to_none=True)
    ((model(x[lo:hi]) - y[lo:hi]) ** 2).mean().backward()
    red.finish()
err = max((p.grad - g).abs().max().item() for p, g in zip(model.parameters(), ref[0]))
# running statistics: every rank saw a different shard; after broadcast_buffers all ranks hold rank 0's copy
bn = torch.nn.BatchNorm1d(16)
bn.train()
bn(x[lo:hi])
n_b = broadcast_buffers(bn)
gathered = [torch.zeros(16) for _ in range(world)]
dist.all_gather(gathered, bn.running_mean)
same = all(torch.equal(g, gathered[0]) for g in gathered)
if rank == 0:
    print(json.dumps({'err': err, 'buckets': len(red.buckets), 'shard': [lo, hi], 'n_buffers': n_b, 'buffers_equal': same}))
dist.destroy_process_group()
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                          '--master-port', '29533', str(script)], capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert res['err'] < 1e-6 and res['buckets'] >= 2 and res['shard'] == [0, 6]
    assert res['n_buffers'] == 2 and res['buffers_equal']


def test_grad_reducer_early_publish_two_ranks_gloo(tmp_path):
    """GradReducer.publish: a hand-driven backward node (like ViTFn) delivers its parameters' gradients itself, block by block, and returns
    None for them; the buckets fill and exchange as with hooks.  World size 2 on CPU/gloo against the single-process gradient."""
    import json
    script = tmp_path / 'pub.py'
    script.write_text('''
import json, os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from whmr_amd.parallel import GradReducer, shard_batch
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
torch.manual_seed(0)

class Body(torch.nn.Module):                                 # two "blocks" whose backward is ONE hand-driven node
    def __init__(self):
        super().__init__()
        self.w1 = torch.nn.Parameter(torch.randn(16, 16) * 0.3)
        self.w2 = torch.nn.Parameter(torch.randn(16, 16) * 0.3)
        self.grad_sink = None
    def forward(self, x):
        return BodyFn.apply(x, self, self.w1, self.w2)

class BodyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, w1, w2):
        h = torch.tanh(x @ w1.t())
        ctx.save_for_backward(x, h, w1, w2)
        ctx.mod = mod
        return h @ w2.t()
    @staticmethod
    def backward(ctx, dy):
        x, h, w1, w2 = ctx.saved_tensors
        sink = ctx.mod.grad_sink
        g2 = dy.t() @ h                                      # last block first
        pub2 = sink is not None and sink(ctx.mod.w2, g2)
        dh = (dy @ w2) * (1 - h * h)
        g1 = dh.t() @ x
        pub1 = sink is not None and sink(ctx.mod.w1, g1)
        return dh @ w1, None, (None if pub1 else g1), (None if pub2 else g2)

body, head = Body(), torch.nn.Linear(16, 4)
params = list(body.parameters()) + list(head.parameters())
x, y = torch.randn(12, 16), torch.randn(12, 4)
ref = torch.autograd.grad(((head(body(x)) - y) ** 2).mean(), params)
red = GradReducer(params, bucket_bytes=600, groups=[0, 0, 1, 1]).attach(body)
lo, hi = shard_batch(12, world, rank)
for step in range(2):
    for p in params:
        p.grad = None
    ((head(body(x[lo:hi])) - y[lo:hi]) ** 2).mean().backward()
    red.finish()
err = max((p.grad - g).abs().max().item() for p, g in zip(params, ref))
if rank == 0:
    print(json.dumps({'err': err, 'buckets': len(red.buckets)}))
dist.destroy_process_group()
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                          '--master-port', '29541', str(script)], capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert res['err'] < 1e-6 and res['buckets'] >= 2


def _run_bench(*extra, timeout=300):
    import json
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--dryrun-cpu', '--batch', '16'] + list(extra),
                       capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_entry_launches_its_own_ranks_gloo():
    """`python bench.py --gpus 2` (no torchrun around it) must start 2 ranks itself as a child process, and `n_gpus` is the
    communicator's own count (reference: train.py:26-33).  CPU/gloo stand-in step, real entry point."""
    out = _run_bench('--gpus', '2', '--steps', '3', '--warmup', '1', '--ref-1gpu', '1000')
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 32 and out['data'] == 'dryrun'
    assert out['scaling'] == 'weak' and 'replicas x2' in out['config']['parallelism']
    assert abs(out['efficiency_vs_1gpu'] - out['value'] / 2000.0) < 1e-12
    assert abs(out['value'] - 2 * 16 * 3 / (out['ms_per_step'] * 3e-3)) < 1e-6 * out['value']
    one = _run_bench('--gpus', '1', '--steps', '2', '--warmup', '0')
    assert one['n_gpus'] == 1 and one['config']['global_batch'] == 16


def test_bench_train_entry_two_ranks_gloo():
    """`bench.py --workload whmr_train --gpus 2` drives GradReducer (with a never-used parameter, as global_orient.* in W-HMR: the reference
    needs DDP's find_unused_parameters, core/trainer.py:84-91) and broadcast_buffers through the real entry point; the averaged gradient
    equals the mean of the two ranks' single-process gradients."""
    out = _run_bench('--gpus', '2', '--steps', '1', '--warmup', '0', '--workload', 'whmr_train', '--batchnorm', 'local')
    assert out['n_gpus'] == 2 and 'dp2' in out['config']['parallelism'] and 'local batch statistics' in out['config']['parallelism']
    d = out['dry']
    assert d['unused_grad_is_none'] and d['skipped_params'] == 2 and d['buckets'] >= 2
    grads, means = [], []
    for rank in range(2):
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.BatchNorm1d(64), torch.nn.GELU(), torch.nn.Linear(64, 8))
        x = torch.randn(16, 32, generator=torch.Generator().manual_seed(7 + rank))
        net(x).pow(2).mean().backward()
        grads.append([p.grad for p in net.parameters()])
        means.append(net[1].running_mean.sum().item())
    avg = [(a + b) / 2 for a, b in zip(*grads)]
    ref = float(sum(g.pow(2).sum() for g in avg).sqrt())
    assert abs(d['grad_norm'] - ref) < 1e-5 * ref
    assert abs(d['running_mean_sum'] - means[0]) < 1e-6              # every rank holds rank 0's running statistics


def test_bench_train_entry_two_ranks_sync_batchnorm_gloo():
    """the DEFAULT of `bench.py --workload whmr_train --gpus 2`: SyncBatchNorm as in the reference (core/trainer.py:83) -- the line says so, and the
    reduced gradient / the running statistics equal those of ONE process on the 32 samples of both ranks (global batch statistics)."""
    out = _run_bench('--gpus', '2', '--steps', '1', '--warmup', '0', '--workload', 'whmr_train')
    assert out['n_gpus'] == 2 and 'SyncBatchNorm as in the reference' in out['config']['parallelism']
    per_rank = out['multi_gpu']['gradient_exchange']['per_rank']
    assert len(per_rank) == 2 and all(r['sync_bn_collectives_total'] == 2 for r in per_rank)          # one forward + one backward exchange per step
    d = out['dry']
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.BatchNorm1d(64), torch.nn.GELU(), torch.nn.Linear(64, 8))
    x = torch.cat([torch.randn(16, 32, generator=torch.Generator().manual_seed(7 + rank)) for rank in range(2)])
    net(x).pow(2).mean().backward()
    ref = float(sum(p.grad.pow(2).sum() for p in net.parameters()).sqrt())
    assert abs(d['grad_norm'] - ref) < 1e-5 * ref
    assert abs(d['running_mean_sum'] - net[1].running_mean.sum().item()) < 1e-6


def test_bench_train_entry_two_ranks_torch_ddp_wrap_gloo():
    """`bench.py --workload whmr_train --gpus 2 --wrap ddp`: torch's OWN `DistributedDataParallel(find_unused_parameters=True)` (core/trainer.py:84-86)
    around the stand-in, with the cross-rank BatchNorm statistics exchanged by this package on the SAME communicator as DDP's buckets (torch refuses
    nn.SyncBatchNorm inside a CPU module, so the stand-in carries this package's mark; the torch class itself runs in the GPU twin,
    tests/test_train_gpu.py::test_reference_syncbn_ddp_wrap_world1_rccl): two gloo ranks, gradient / running statistics of ONE process on the 32
    samples of both ranks; then four steps (DDP rebuilds its buckets after the first)."""
    out = _run_bench('--gpus', '2', '--steps', '1', '--warmup', '0', '--workload', 'whmr_train', '--wrap', 'ddp')
    assert out['n_gpus'] == 2 and 'torch DistributedDataParallel' in out['config']['parallelism']
    d = out['dry']
    assert d['wrap'] == 'ddp' and d['unused_grad_is_none']
    # the timed step, the instrumented step is absent in a dry run, the clock-probe loop is GPU-only: one forward + one backward exchange per executed step
    per_rank = out['multi_gpu']['sync_batchnorm']['per_rank']
    assert len(per_rank) == 2 and per_rank[0] == per_rank[1] and per_rank[0]['sync_bn_collectives_total'] == 2
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.BatchNorm1d(64), torch.nn.GELU(), torch.nn.Linear(64, 8))
    x = torch.cat([torch.randn(16, 32, generator=torch.Generator().manual_seed(7 + rank)) for rank in range(2)])
    net(x).pow(2).mean().backward()
    ref = float(sum(p.grad.pow(2).sum() for p in net.parameters()).sqrt())
    assert abs(d['grad_norm'] - ref) < 1e-5 * ref
    assert abs(d['running_mean_sum'] - net[1].running_mean.sum().item()) < 1e-6
    out3 = _run_bench('--gpus', '2', '--steps', '3', '--warmup', '1', '--workload', 'whmr_train', '--wrap', 'ddp')
    assert out3['multi_gpu']['sync_batchnorm']['per_rank'][0]['sync_bn_collectives_total'] == 8


def test_grad_reducer_unused_and_misuse():
    """single process, always_bucket: unused parameters are skipped (grad None), a second backward before finish() raises, and a
    skipped parameter that later gets a gradient raises"""
    from whmr_amd.parallel import GradReducer
    torch.manual_seed(0)
    a, b, c = torch.nn.Linear(4, 4), torch.nn.Linear(4, 4), torch.nn.Linear(4, 2)
    params = list(a.parameters()) + list(b.parameters()) + list(c.parameters())
    red = GradReducer(params, bucket_bytes=64, always_bucket=True, groups=[0, 0, 1, 1, 2, 2])
    x = torch.randn(3, 4)
    ref = torch.autograd.grad(c(a(x)).sum(), list(a.parameters()) + list(c.parameters()))
    for step in range(2):
        for p in params:
            p.grad = None
        c(a(x)).sum().backward()
        red.finish()
        assert all(p.grad is None for p in b.parameters()) and len(red.skipped) == 2
        for p, g in zip(list(a.parameters()) + list(c.parameters()), ref):
            assert torch.allclose(p.grad, g)
    assert all(id(p) not in red._slot for p in b.parameters())
    c(a(x)).sum().backward()
    with pytest.raises(RuntimeError, match='second backward'):
        c(a(x)).sum().backward()
    red._fired.clear()
    for bk in red.buckets:
        bk['pending'], bk['flat'] = len(bk['params']), None
    with pytest.raises(RuntimeError, match='classified as unused'):
        c(b(a(x))).sum().backward()
    red.remove()


def test_grad_reducer_deferred_launch_keeps_bucket_order():
    """bucket k + 1 completes (and is packed) BEFORE bucket k -- its collective is held back by _launch_ready and started later, from
    another delivery; collectives still start in bucket order and each one sees a fully packed buffer."""
    from whmr_amd.parallel import GradReducer
    torch.manual_seed(0)
    a, b = torch.nn.Linear(4, 4), torch.nn.Linear(4, 4)
    params = list(a.parameters()) + list(b.parameters())
    red = GradReducer(params, bucket_bytes=1 << 20, always_bucket=True, groups=[0, 0, 1, 1])
    assert len(red.buckets) == 2 and red.buckets[0]['params'][0] is b.bias      # bucket 0 = the LAST parameters (backward order)
    launched = []
    real = red._launch
    red._launch = lambda bk: (launched.append(([i for i, q in enumerate(red.buckets) if q is bk][0], bk['flat'].clone())), real(bk))[1]
    grads = [torch.randn_like(p) for p in params]
    # deliver bucket 1 (a.*) first: it packs, but may not launch before bucket 0
    for p, g in zip(params[:2], grads[:2]):
        assert red.publish(p, g)
    assert red.buckets[1]['flat'] is not None and launched == []
    for p, g in zip(params[2:], grads[2:]):
        assert red.publish(p, g)
    assert [i for i, _ in launched] == [0, 1]
    want1 = torch.cat([grads[1].reshape(-1), grads[0].reshape(-1)])                # bucket 1 in backward order: a.bias, a.weight
    assert torch.equal(launched[1][1], want1)
    red.finish()
    for p, g in zip(params, grads):
        assert torch.equal(p.grad, g)
    red.remove()


def test_sync_batchnorm_two_ranks_reproduce_one_process_gloo(tmp_path):
    """VERDICT r4 item 2 (core/trainer.py:83): two ranks of batch B under convert_sync_batchnorm reproduce ONE process of batch 2B -- outputs, running
    statistics, data gradients, and parameter gradients (the ranks' local sums add up to the full ones, what the gradient reducer then averages).
    Both protocols run for real over gloo: the channels-last BatchNorm2d + ReLU of the deconv stages (whmr.py:497; the four HIP entry points are
    replaced by CPU restatements of their C contract, include/whmr_hip.h, written here -- the exchange, what is summed, what stays local and what
    is saved between forward and backward is the product code of whmr_amd/parallel/sync_bn.py) and the BatchNorm1d(1) of the Tz head
    (whmr.py:428; plain tensor arithmetic, the product function as it is)."""
    import json
    script = tmp_path / 'syncbn.py'
    script.write_text('''
import json, sys, torch, torch.nn.functional as F, torch.distributed as dist
sys.path.insert(0, %r)
import whmr_amd
from whmr_amd import _lib as L
from whmr_amd.parallel import sync_bn, convert_sync_batchnorm

# ---- CPU restatements of the C contract of the split BatchNorm entries (include/whmr_hip.h) -- test doubles, never product code
def bn_sums(z):
    zd = z.double()
    return torch.cat([zd.sum(0), (zd * zd).sum(0), zd.new_full((1,), float(z.shape[0]))])
def bn_stats_from_sums(sums, gamma, beta, eps, momentum=0.0, running_mean=None, running_var=None):
    C = (sums.numel() - 1) // 2
    n = sums[-1]
    mean = sums[:C] / n
    var = (sums[C:2 * C] / n - mean * mean).clamp_min(0)
    invstd = (1.0 / torch.sqrt(var + eps)).float()
    a = gamma * invstd
    stats = torch.stack([mean.float(), invstd, a, beta - mean.float() * a])
    if running_mean is not None:
        running_mean.mul_(1 - momentum).add_(momentum * mean.float())
        running_var.mul_(1 - momentum).add_(momentum * (var * n / (n - 1)).float())
    return stats
def bn_apply_relu(z, stats, y):
    y.copy_(torch.relu(z * stats[2] + stats[3]))
def bn_bwd_sums(z, dy, stats, dgamma, dbeta, accumulate=False):
    g = (dy * ((z * stats[2] + stats[3]) > 0)).double()
    xhat = ((z - stats[0]) * stats[1]).double()
    s = torch.cat([g.sum(0), (g * xhat).sum(0)])
    C = z.shape[1]
    dbeta.copy_(s[:C].float()); dgamma.copy_(s[C:].float())
    return s
def bn_bwd_apply(z, dy, stats, sums, count, dz):
    C = z.shape[1]
    g = dy * ((z * stats[2] + stats[3]) > 0)
    xhat = (z - stats[0]) * stats[1]
    dz.copy_(stats[2] * (g - (sums[:C] / count).float() - xhat * (sums[C:] / count).float()))
for f in (bn_sums, bn_stats_from_sums, bn_apply_relu, bn_bwd_sums, bn_bwd_apply):
    setattr(L, f.__name__, f)

dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
g = torch.Generator().manual_seed(3)
B, C, HW = 6, 16, 10                         # per-rank batch 3
z_full = torch.randn(B * HW, C, generator=g) * 2 + torch.linspace(-3, 3, C)          # channels-last rows, b-major
dy_full = torch.randn(B * HW, C, generator=g)
gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
# single process, batch 2B: torch's own BatchNorm (training) + ReLU and its autograd
zr = z_full.clone().requires_grad_(True)
gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
rm, rv = torch.zeros(C), torch.ones(C)
y_ref = torch.relu(F.batch_norm(zr, rm, rv, gr, br, True, 0.1, 1e-5))
y_ref.backward(dy_full)
# two ranks of batch B
bn = torch.nn.BatchNorm2d(C)
with torch.no_grad():
    bn.weight.copy_(gamma); bn.bias.copy_(beta)
convert_sync_batchnorm(bn)
sg = sync_bn.sync_of(bn)
assert sg is not None and sg.world() == 2
rows = slice(rank * (B // 2) * HW, (rank + 1) * (B // 2) * HW)
z, dy = z_full[rows].contiguous(), dy_full[rows].contiguous()
y, stats, count = sync_bn.bn_relu_forward(z, bn.weight.detach(), bn.bias.detach(), bn, True, sg)
dz, dg, db = torch.empty_like(z), torch.empty(C), torch.empty(C)
sync_bn.bn_relu_backward(z, dy, stats, count, dz, dg, db, sg)
err = lambda a, b: ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()
res = {'y': err(y, y_ref.detach()[rows]), 'dz': err(dz, zr.grad[rows]), 'rm': err(bn.running_mean, rm), 'rv': err(bn.running_var, rv), 'count': float(count)}
both = torch.stack([dg, db]); dist.all_reduce(both)                                   # the ranks' LOCAL parameter gradients add up to the full ones
res['dgamma'], res['dbeta'] = err(both[0], gr.grad), err(both[1], br.grad)
res['collectives'] = sg.collectives
# ---- BatchNorm1d(1) of the Tz head (whmr.py:428), the product function itself
x_full = (torch.randn(B, 1, generator=g) * 3 + 1.5)
c_full = torch.randn(B, 1, generator=g)
xr = x_full.clone().requires_grad_(True)
ref1 = torch.nn.BatchNorm1d(1)
bn1 = torch.nn.BatchNorm1d(1)
with torch.no_grad():
    for m in (ref1, bn1):
        m.weight.fill_(1.3); m.bias.fill_(-0.2)
(ref1(xr) * c_full).sum().backward()
convert_sync_batchnorm(bn1)
r1 = slice(rank * (B // 2), (rank + 1) * (B // 2))
xl = x_full[r1].clone().requires_grad_(True)
yl = sync_bn.batch_norm_1d(xl, bn1)
(yl * c_full[r1]).sum().backward()
y1_ref = (x_full - x_full.mean(0)) / torch.sqrt(x_full.var(0, unbiased=False) + 1e-5) * 1.3 - 0.2
res['y1'], res['dx1'] = err(yl.detach(), y1_ref[r1]), err(xl.grad, xr.grad[r1])
res['rm1'], res['rv1'] = err(bn1.running_mean, ref1.running_mean), err(bn1.running_var, ref1.running_var)
w1 = torch.stack([bn1.weight.grad, bn1.bias.grad]); dist.all_reduce(w1)
res['dw1'], res['db1'] = err(w1[0], ref1.weight.grad), err(w1[1], ref1.bias.grad)
# eval mode: the marked layer is the plain module (running statistics, no exchange)
bn1.eval()
n0 = sync_bn.sync_of(bn1).collectives
bn1(x_full)
sync_bn.batch_norm_1d(x_full, bn1)
res['eval_collectives'] = sync_bn.sync_of(bn1).collectives - n0
allres = [None] * world
dist.all_gather_object(allres, res)
if rank == 0:
    print(json.dumps(allres))
dist.destroy_process_group()
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                          '--master-port', '29547', str(script)], capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    allres = json.loads([l for l in out.stdout.splitlines() if l.startswith('[')][-1])
    assert len(allres) == 2
    for res in allres:
        assert res['count'] == 60.0 and res['collectives'] == 2 and res['eval_collectives'] == 0, res
        for k in ('y', 'dz', 'rm', 'rv', 'dgamma', 'dbeta', 'y1', 'dx1', 'rm1', 'rv1', 'dw1', 'db1'):
            assert res[k] < 5e-6, (k, res)                      # fp32 rounding (measured 0 .. 1.5e-6)
    print('sync BN, 2 ranks vs 1 process of 2B:', {k: '%.1e' % v for k, v in allres[0].items() if isinstance(v, float) and k != 'count'})


def test_integration_doc_lists_every_entry_point():
    """INTEGRATION.md's entry-point table names every symbol include/whmr_hip.h declares (and nothing the header lacks)"""
    import re
    header = open(os.path.join(ROOT, 'include', 'whmr_hip.h')).read()
    declared = set(re.findall(r'\b(whmr_[a-z0-9_]+)\s*\(', header))
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    missing = sorted(s for s in declared if s not in doc)
    assert not missing, 'INTEGRATION.md does not mention: %s' % missing
    table = doc[doc.index('| C entry point |'):]
    ghosts = sorted(s for s in set(re.findall(r'`(whmr_[a-z0-9_]+)`', table)) if s not in declared and not s.startswith('whmr_amd'))
    assert not ghosts, 'INTEGRATION.md names entry points the header does not declare: %s' % ghosts


def test_bench_launcher_kills_hung_ranks_and_exits_nonzero():
    """a rank that never finishes must end as a non-zero exit of the launcher, not as a hang of the driver -- the children
    run in their own process group, which a watchdog kills after --rank-timeout (here: shorter than the ranks' import time)"""
    import subprocess
    import sys
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dryrun-cpu', '--rank-timeout', '0.3'],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 124, (r.returncode, r.stderr[-400:])
    assert 'were killed' in r.stderr and '"metric"' not in r.stdout
    assert time.time() - t0 < 60


def test_bench_launcher_ends_ranks_that_really_hang():
    """torch elastic starts every worker in its own session, so killing the agent's process group leaves the ranks alive as
    orphans that hold the GPUs and the stdout pipe.  Here the two ranks are UP (rendezvous done, step running) and never finish: the launcher
    must return 124 within timeout + grace, and no process of the tree may survive it."""
    import subprocess
    import sys
    import time
    import psutil
    marker = 'whmr-hang-%d' % os.getpid()
    env = dict(os.environ, WHMR_BENCH_DRYRUN_HANG=marker)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dryrun-cpu', '--rank-timeout', '25', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=180, env=env)
    took = time.time() - t0
    assert r.returncode == 124, (r.returncode, r.stdout[-400:], r.stderr[-800:])
    assert 'were killed' in r.stderr and '"metric"' not in r.stdout
    assert 24 < took < 60, took                  # the ranks were running when the watchdog fired, and the launcher did not wait for them
    time.sleep(0.5)
    left = []
    for q in psutil.process_iter(['pid', 'environ', 'cmdline', 'status']):
        try:
            if (q.info['environ'] or {}).get('WHMR_BENCH_DRYRUN_HANG') == marker and q.info['status'] != psutil.STATUS_ZOMBIE:
                left.append((q.info['pid'], q.info['cmdline']))
        except (psutil.NoSuchProcess, psutil.AccessDenied):
            pass
    assert not left, left


def test_bench_multi_gpu_block_explains_the_run():
    """at world > 1 the line carries what explains a scaling curve -- every rank's own step time, the rank -> device map (unique
    devices), the backend, and for the training step the gradient exchange (bytes, collectives, buckets, exposed wait of finish())."""
    out = _run_bench('--gpus', '4', '--steps', '2', '--warmup', '1', '--workload', 'whmr_train')
    assert out['n_gpus'] == 4 and out['config']['global_batch'] == 64
    mg = out['multi_gpu']
    ms = mg['per_rank_ms_per_step']
    assert len(ms['all']) == 4 and ms['min'] <= ms['max'] <= out['ms_per_step'] * 1.0001 + 1e-9
    assert sorted(r['rank'] for r in mg['ranks']) == [0, 1, 2, 3] and len({r['device_id'] for r in mg['ranks']}) == 4
    assert len({r['pid'] for r in mg['ranks']}) == 4
    ge = mg['gradient_exchange']['per_rank']
    assert len(ge) == 4
    n_used = (32 * 64 + 64) + 2 * 64 + (64 * 8 + 8)                 # Linear, BatchNorm affine, Linear of the stand-in net; the unused Linear(8, 8) left the buckets
    for r in ge:
        assert r['unused_parameters'] == 2 and r['buckets'] >= 2
        assert sum(r['bucket_bytes']) == 4 * n_used
        # step 1 of the timed region still carries the two unused parameters' zeros (the census runs at the first finish() = the warm-up step)
        assert abs(r['bytes_exchanged_per_step'] - 4 * n_used) < 1e-6
        assert r['collectives_per_step'] == r['buckets']
        assert r['exposed_exchange_wait_ms_max'] >= r['exposed_exchange_wait_ms_mean'] >= 0.0
    # forward workloads: no exchange block, still the rank map; ViT-L at 32 crops per rank = BASELINE configs[4]'s per-GPU share (bookkeeping only here)
    fw = _run_bench('--gpus', '2', '--steps', '2', '--warmup', '0', '--workload', 'vitl256x192', '--batch', '32')
    assert fw['n_gpus'] == 2 and fw['config']['global_batch'] == 64 and 'gradient_exchange' not in fw['multi_gpu']
    assert abs(fw['value'] - 2 * 32 * 2 / (fw['ms_per_step'] * 2e-3)) < 1e-6 * fw['value']


def test_bench_refuses_two_ranks_on_one_device():
    """two ranks that report the same device id (a wrong LOCAL_RANK -> device map) end the run before anything is timed"""
    import subprocess
    import sys
    env = dict(os.environ, WHMR_BENCH_DRYRUN_SAME_DEVICE='1')
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dryrun-cpu', '--steps', '1', '--warmup', '0', '--rank-timeout', '120'],
                       capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode != 0 and '"metric"' not in r.stdout
    assert 'two ranks share a device' in r.stderr


def test_grad_reducer_four_ranks_unequal_parameter_use(tmp_path):
    """world 4 (gloo): the ranks use DIFFERENT parameters in some steps (a branch only even ranks take, as a loss term that is absent from part of
    a batch).  The static-graph contract: the first step's census must agree -- so step 0 uses everything everywhere -- and later a locally missing
    gradient travels as zeros; the result equals the mean over the four ranks of the per-rank gradients (zeros where unused)."""
    import json
    import subprocess
    import sys
    script = tmp_path / 'w4.py'
    script.write_text("""
import json, os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from whmr_amd.parallel import GradReducer
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
torch.manual_seed(0)
trunk, head_a, head_b = torch.nn.Linear(6, 6), torch.nn.Linear(6, 3), torch.nn.Linear(6, 2)
params = list(trunk.parameters()) + list(head_a.parameters()) + list(head_b.parameters())
red = GradReducer(params, bucket_bytes=96, groups=[0, 0, 1, 1, 2, 2])
worst = 0.0
for step in range(3):
    for p in params:
        p.grad = None
    xs = [torch.randn(5, 6, generator=torch.Generator().manual_seed(100 * step + r)) for r in range(world)]
    use_b = lambda r: step == 0 or r %% 2 == 0          # head_b: everywhere in step 0, on the even ranks afterwards
    def loss_of(r):
        h = trunk(xs[r])
        l = head_a(h).pow(2).mean()
        return l + head_b(h).pow(2).mean() if use_b(r) else l
    # reference: mean over ranks of the single-process gradients (a parameter a rank did not use contributes zeros)
    ref = [torch.zeros_like(p) for p in params]
    for r in range(world):
        gs = torch.autograd.grad(loss_of(r), params, allow_unused=True)
        for acc, g in zip(ref, gs):
            if g is not None:
                acc += g / world
    loss_of(rank).backward()
    red.finish()
    for p, g in zip(params, ref):
        worst = max(worst, float((p.grad - g).abs().max()))
if rank == 0:
    print(json.dumps({'err': worst, 'buckets': len(red.buckets), 'collectives': red.stats['collectives'], 'skipped': len(red.skipped)}))
dist.destroy_process_group()
""" % ROOT)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=4', '--master-addr', '127.0.0.1',
                          '--master-port', '29547', str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2500:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert res['err'] < 1e-6 and res['buckets'] == 4 and res['skipped'] == 0 and res['collectives'] == 12     # (96-byte buckets: the trunk's weight and bias travel apart)


def test_split_bf16_operand_pairs_and_elementwise_metric():
    """host side of the bf16x3 numerics: a hi / lo pair carries >= 16 significand bits, the K-concatenated weight layout pairs up with the
    activation layout ([x_hi | x_lo | x_hi] . [W_hi | W_hi | W_lo] = x_hi W_hi + x_lo W_hi + x_hi W_lo), and the element-wise parity metric
    holds small components to the tensor's own scale"""
    from conftest import ew_err
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(0)
    x = torch.randn(64, 96, generator=g) * 3
    w = torch.randn(32, 96, generator=g)
    xh, xl = L.split_bf16(x)
    assert xh.dtype == xl.dtype == torch.bfloat16
    assert ((xh.double() + xl.double() - x.double()).abs() / x.double().abs()).max() < 2.0 ** -16
    w3 = L.split3_weight(w)
    assert w3.shape == (32, 288) and w3.dtype == torch.bfloat16
    x3 = torch.cat([xh, xl, xh], 1)
    wh, wl = L.split_bf16(w)
    three = xh.double() @ wh.double().t() + xl.double() @ wh.double().t() + xh.double() @ wl.double().t()
    assert torch.allclose(x3.double() @ w3.double().t(), three, rtol=0, atol=1e-9)
    exact = x.double() @ w.double().t()
    assert ((three - exact).abs().max() / exact.abs().max()) < 2e-5              # the dropped lo.lo term + the pairs' rounding
    assert ((xh.double() @ wh.double().t() - exact).abs().max() / exact.abs().max()) > 1e-3     # plain bf16 for scale
    # ew_err: a 1e-3 error on a component 1000x below the tensor's largest entry passes max-rel at 1e-6 but not the element-wise gate
    b = torch.tensor([1000.0, 1.0, 1.0, 1.0])
    a = b.clone()
    a[1] += 1e-3
    assert ((a - b).abs().max() / b.abs().max()) < 1e-5 and ew_err(a, b) > 1e-6 and ew_err(b, b) == 0.0


def test_ctypes_structures_match_the_c_header_layout(tmp_path):
    """the drop-in boundary is a C ABI: every ctypes mirror in whmr_amd/_lib.py has the size and the field offsets gcc gives the struct of
    include/whmr_hip.h (a field added on one side only, or a wrong width, would shift every pointer behind it)"""
    import ctypes
    import subprocess
    from whmr_amd import _lib as L
    pairs = {'whmr_gemm': L.WhmrGemm, 'whmr_gemm_blk_desc': L.WhmrGemmBlk, 'whmr_tn_item': L.WhmrTnItem, 'whmr_smpl_model': L.WhmrSmplModel,
             'whmr_maf_weights': L.WhmrMafWeights, 'whmr_stage_tail': L.WhmrStageTail}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "whmr_hip.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append('  printf("%s %%zu\\n", sizeof(struct %s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('  printf("%s.%s %%zu\\n", offsetof(struct %s, %s));' % (cname, fname, cname, fname))
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), (cname, got[cname], ctypes.sizeof(cls))
        for fname, _ in cls._fields_:
            assert int(got['%s.%s' % (cname, fname)]) == getattr(cls, fname).offset, (cname, fname)


def test_composed_tz_weights_equal_the_two_convolutions():
    """whmr.py:418-421: conv1(conv0(x)) == Conv2d(256 -> 5, k25, s6) with the composed weights, in the space-to-depth GEMM + 25-term fold form the
    HIP path evaluates (models/whmr.py::compose_tz_weights / _tz_tokens_composed; csrc/tz_head.hip::tz_fold_kernel) -- host algebra, torch fp64."""
    import torch.nn.functional as F
    from whmr_amd.models.whmr import compose_tz_weights
    g = torch.Generator().manual_seed(0)
    B, H, W, C = 1, 128, 96, 16
    x = torch.randn(B, H, W, C, generator=g, dtype=torch.float64)
    w0 = torch.randn(64, C, 7, 7, generator=g, dtype=torch.float64) * 0.05
    w1 = torch.randn(5, 64, 7, 7, generator=g, dtype=torch.float64) * 0.05
    ref = F.conv2d(F.conv2d(x.permute(0, 3, 1, 2), w0, stride=3), w1, stride=2)
    G = compose_tz_weights(w0, w1)
    assert G.shape == (128, 36 * C) and G.dtype == torch.float32 and not G[125:].any()
    xp = torch.cat([x, x.new_zeros(B, 22 * 6 - H, W, C)], 1).view(B, 22, 6, 16, 6 * C).permute(0, 1, 3, 2, 4).reshape(B * 22 * 16, 36 * C)
    P = (xp @ G.double().t()).view(B, 22, 16, 128)
    tok = torch.zeros_like(ref)
    for jA in range(5):
        for jB in range(5):
            for o in range(5):
                tok[:, o] += P[:, jA:jA + 18, jB:jB + 12, (jA * 5 + jB) * 5 + o]
    assert ((tok - ref).abs().max() / ref.abs().max()).item() < 1e-6

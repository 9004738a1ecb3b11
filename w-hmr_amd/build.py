"""Build libwhmr_hip.so for gfx950 with hipcc (cross-compiles without a GPU).  `python -m whmr_amd.build`."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'libwhmr_hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result']


def _stale(target, deps):
    return not os.path.exists(target) or any(os.path.getmtime(d) > os.path.getmtime(target) for d in deps)


# Round 6 (profiles/r06_coresidency_probe.txt): `v_pk_fma_f32 ... op_sel:[0,1,0]` (both lanes times the HIGH half of a register pair -- what hipcc makes of "a
# pair times one scalar of a pair") returned wrong LOW lanes while one particular MFMA kernel of this library ran on another stream (a canary of nothing but
# that instruction: 457 552 wrong lanes beside it, none beside anything else, none for the plain form).  That kernel is gone; the files whose kernels
# contain the form are nevertheless built without packed-fp32 instructions -- they are memory- or latency-bound VALU kernels that run beside the GEMM
# streams (the host pass prints "not a recognized feature" for the flag and ignores it).
_NO_PK = ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']
NO_PACKED_FP32 = {f: _NO_PK for f in ('smpl_lbs.hip', 'smpl_fused.hip', 'geometry.hip', 'iuv_loss.hip', 'maf_sampler.hip', 'rasterize.hip', 'vit_ops.hip')}
if os.environ.get('WHMR_BUILD_PACKED_FP32', '0') == '1':                 # A/B
    NO_PACKED_FP32 = {}


def build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    hdrs = glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(CSRC, '*.inc'))
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + '.o')
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [hipcc] + FLAGS + NO_PACKED_FP32.get(os.path.basename(s), []) + ['-c', s, '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        noise = "'-packed-fp32-ops' is not a recognized feature for this target (ignoring feature)"      # the host pass of the files built without packed fp32
        err = '\n'.join(l for l in r.stderr.splitlines() if l.strip() != noise)
        if err:
            print(err, file=sys.stderr, flush=True)
        if r.returncode:
            raise subprocess.CalledProcessError(r.returncode, cmd)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, os.path.basename(s)[:-4] + '.o') for s in srcs]
    if force or jobs or _stale(OUT, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)

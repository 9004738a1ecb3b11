"""Round 6 probe: does CU MASKING take the camera-calibration branch out of the backbone's way?  The camera chain (74 small dependent launches, 0.79 ms alone)
costs 0.33 ms of the full forward because its workgroups land on CUs the one-workgroup-per-CU ViT / deconv GEMMs need.  Here the camera stream is created with
hipExtStreamCreateWithCUMask on a few CUs per XCD and the main + heavy-chain streams on the complement (eager launches: a HIP-graph replay does not keep a
capture stream's CU mask).   python tools/r6_cu_mask_probe.py"""
import ctypes
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd.models import whmr as W
from whmr_amd.models import whmr_net
from whmr_amd.utils import synth

dev = torch.device('cuda:0')
hip = ctypes.CDLL('libamdhip64.so')


def masked_stream(bits):
    """bits: iterable of CU mask bit indices (ROCr interleaves the bits over the XCCs: bit i -> XCC i % 8, CU i // 8)"""
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, 'hipExtStreamCreateWithCUMask failed: %d' % rc
    return torch.cuda.ExternalStream(s.value, device=dev)


assets = synth.make_assets(0)
sd = synth.make_state_dict(0, assets)
m = whmr_net(None, assets=assets, numerics=sys.argv[1] if len(sys.argv) > 1 else 'bf16')
m.load_state_dict(sd, strict=True)
m = m.to(dev).eval()
B = 64
inp = {k: v.to(dev) for k, v in synth.make_inputs(B, 7).items()}
a = (inp['x'], None, inp['center'], inp['scale'], inp['bbox_height'], inp['orig_shape'], inp['bbox_info'])
full = torch.randn(1, 3, 600, 800, generator=torch.Generator().manual_seed(11)).to(dev)


def timeit(fn, n=30, w=10):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    ref = {k: v.clone() for k, v in m(*a, full_x=full).items()}
    base = timeit(lambda: m(*a, full_x=full))
    nocam = timeit(lambda: m(*a))
    print('eager, no masks: %.3f ms with the camera branch, %.3f ms without any camera work' % (base, nocam), flush=True)
    for ncam in (8, 16, 24, 32):
        cam_bits, main_bits = range(ncam), range(ncam, 256)
        saved = dict(W._CAM_STREAMS)
        try:
            W._CAM_STREAMS[(dev, None)] = masked_stream(cam_bits)
            W._CAM_STREAMS[(dev, 'tz')] = masked_stream(main_bits)
            main = masked_stream(main_bits)

            def run():
                cur = torch.cuda.current_stream()
                main.wait_stream(cur)
                with torch.cuda.stream(main):
                    out = m(*a, full_x=full)
                cur.wait_stream(main)
                return out
            out = run()
            torch.cuda.synchronize()
            same = all(torch.equal(out[k], ref[k]) for k in ref)
            t = timeit(run)
            t_nocam = timeit(lambda: (main.wait_stream(torch.cuda.current_stream()), torch.cuda.stream(main).__enter__(), m(*a), torch.cuda.current_stream().__class__, None)[-1]) if False else float('nan')
            print('camera branch on %2d CUs (%d per XCD), backbone + heavy chain + loop on %3d: %.3f ms  (same bits: %s)' % (ncam, ncam // 8, 256 - ncam, t, same), flush=True)
        finally:
            W._CAM_STREAMS.clear()
            W._CAM_STREAMS.update(saved)
    # the same masks for the backbone alone: what do 240 / 248 CUs cost the main path when there is no camera work at all?
    for ncam in (8, 16):
        main = masked_stream(range(ncam, 256))
        W._CAM_STREAMS[(dev, 'tz')] = masked_stream(range(ncam, 256))

        def run2():
            cur = torch.cuda.current_stream()
            main.wait_stream(cur)
            with torch.cuda.stream(main):
                out = m(*a)
            cur.wait_stream(main)
            return out
        print('no camera work, everything on %3d CUs: %.3f ms' % (256 - ncam, timeit(run2)), flush=True)
        W._CAM_STREAMS.pop((dev, 'tz'), None)

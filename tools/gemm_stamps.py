"""Stamped time budget of the blocked bf16 GEMM launches INSIDE the ViT-B/16 224^2 batch-64 forward ("a stamped budget that
adds up to the measured launch").  Needs the lab build of the library with WHMR_BLK_STAMPS (tools/lab/libwhmr_hip_stamps.so; see tools/r5_gemm.sh):
every tile records s_memrealtime (100 MHz, chip-wide) at kernel entry / first half tile landed / main loop done / epilogue stores issued / stores
drained, the CU it ran on and its shader-clock count.  Per GEMM shape this prints, averaged over the 12 layers:

  launch span (first entry -> last drain), gap to the previous launch's last drain,
  per ROUND (first / second tile a CU ran): prologue, main loop (and per 32-deep half tile), epilogue issue, store drain, turnover gap on the CU,
  the critical CU's sum, and the effective shader clock.

    python tools/gemm_stamps.py [--lib tools/lab/libwhmr_hip_stamps.so] [--numerics bf16]
"""
import argparse
import collections
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402,F401  (puts the package alias in place)
import torch  # noqa: E402
from whmr_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lib', default=os.path.join(ROOT, 'tools', 'lab', 'libwhmr_hip_stamps.so'))
    ap.add_argument('--batch', type=int, default=64)
    args = ap.parse_args()
    L.LIB_PATH = os.path.abspath(args.lib)
    lib = L.lib()
    lib.whmr_debug_blk_stamps.argtypes = [ctypes.c_void_p, ctypes.c_long]
    lib.whmr_debug_blk_stamps_used.restype = ctypes.c_long
    from whmr_amd.models.pose_vit import ViT
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    m = ViT(img_size=224, qkv_bias=True, numerics='bf16').to(dev).eval()
    for p in m.parameters():
        if p.dim() > 1:
            torch.nn.init.normal_(p, std=0.02)
    x = torch.randn(args.batch, 3, 224, 224, device=dev)
    with torch.no_grad():
        for _ in range(3):
            m(x)
    torch.cuda.synchronize()
    buf = torch.zeros(2_000_000, dtype=torch.int64, device=dev)
    launches = []
    real = L.gemm_blk

    def wrapped(a, w, c, M, **kw):
        before = lib.whmr_debug_blk_stamps_used()
        r = real(a, w, c, M, **kw)
        after = lib.whmr_debug_blk_stamps_used()
        N = c.shape[1] * c.shape[3]
        K = a.shape[1] * a.shape[3]
        launches.append((before, after, M, N, K, kw.get('epi', 0)))
        return r
    L.gemm_blk = wrapped
    import whmr_amd.models.pose_vit as PV
    PV.L.gemm_blk = wrapped
    lib.whmr_debug_blk_stamps(buf.data_ptr(), buf.numel())
    with torch.no_grad():
        m(x)
    torch.cuda.synchronize()
    lib.whmr_debug_blk_stamps(None, 0)
    data = buf.cpu().numpy().astype('int64')
    TICK = 0.01                                       # us per s_memrealtime tick (100 MHz)
    names = {(2304, 768): 'qkv', (768, 768): 'proj', (3072, 768): 'fc1 + GELU', (768, 3072): 'fc2'}
    agg = collections.defaultdict(list)
    prev_end = None
    for before, after, M, N, K, epi in launches:
        if after == before:
            continue
        tiles = (after - before) // 16
        rec = data[before:after].reshape(tiles, 2, 8)
        ok = rec[:, 0, 0] > 0
        rec = rec[ok]
        t0 = rec[:, :, 0].min()
        tend = rec[:, :, 4].max()
        H = K // 32
        # per CU: tiles in entry order (wave group 0's stamps)
        cu = collections.defaultdict(list)
        for t in range(rec.shape[0]):
            hw = int(rec[t, 0, 5])
            key = (hw >> 32 & 0xf, hw >> 13 & 7, hw >> 12 & 1, hw >> 8 & 0xf)          # XCC, SE, SH, CU
            cu[key].append(rec[t])
        rounds = collections.defaultdict(lambda: collections.defaultdict(list))
        crit = 0.0
        for key, lst in cu.items():
            lst.sort(key=lambda r: r[0, 0])
            last_end = None
            for i, r in enumerate(lst):
                g0, g1 = r[0], r[1]
                rd = rounds[min(i, 2)]
                rd['start_after_launch'].append((g0[0] - t0) * TICK)
                rd['prologue'].append((g0[1] - g0[0]) * TICK)
                rd['loop_g0'].append((g0[2] - g0[1]) * TICK)
                rd['loop_g1'].append((g1[2] - g1[1]) * TICK)
                rd['epi_issue_g0'].append((g0[3] - g0[2]) * TICK)
                rd['epi_issue_g1'].append((g1[3] - g1[2]) * TICK)
                rd['drain_g1'].append((g1[4] - g1[3]) * TICK)
                rd['tile_total'].append((max(g0[4], g1[4]) - g0[0]) * TICK)
                clk = (g0[7] - g0[6]) / max(1e-9, (g0[4] - g0[0]) * TICK)          # shader cycles per us = MHz
                rd['clock_mhz'].append(clk)
                if last_end is not None:
                    rd['turnover'].append((g0[0] - last_end) * TICK)
                last_end = max(g0[4], g1[4])
            crit = max(crit, (last_end - t0) * TICK)
        name = names.get((N, K), '%dx%d' % (N, K))
        agg[name].append(dict(span=(tend - t0) * TICK, gap=None if prev_end is None else (t0 - prev_end) * TICK, tiles=int(rec.shape[0]), cus=len(cu), H=H,
                              rounds={k: {kk: sum(v) / len(v) for kk, v in d.items()} | {'n': len(d['prologue'])} for k, d in rounds.items()}))
        prev_end = tend
    print('# tools/gemm_stamps.py: ViT-B/16 224^2 batch %d bf16, one instrumented forward; s_memrealtime stamps (10 ns), us; mean over the launches of a shape' % args.batch)
    print('# (the stamps build drains the stores before the last stamp: "drain" is visible here, the product kernel ends without waiting)')
    for name, lst in agg.items():
        n = len(lst)
        span = sum(d['span'] for d in lst) / n
        gaps = [d['gap'] for d in lst if d['gap'] is not None]
        print('%-11s launches %2d tiles %3d on %3d CUs, H = %2d half tiles: span %6.2f us (min %6.2f max %6.2f), gap behind the previous GEMM launch %5.2f us (other kernels in between count)'
              % (name, n, lst[0]['tiles'], lst[0]['cus'], lst[0]['H'], span, min(d['span'] for d in lst), max(d['span'] for d in lst), sum(gaps) / max(1, len(gaps))))
        for rd in sorted(lst[0]['rounds']):
            keys = ['start_after_launch', 'prologue', 'loop_g0', 'loop_g1', 'epi_issue_g0', 'epi_issue_g1', 'drain_g1', 'tile_total', 'turnover', 'clock_mhz']
            vals = {}
            cnt = sum(d['rounds'].get(rd, {}).get('n', 0) for d in lst) / n
            for k in keys:
                v = [d['rounds'][rd][k] for d in lst if rd in d['rounds'] and k in d['rounds'][rd]]
                if v:
                    vals[k] = sum(v) / len(v)
            H = lst[0]['H']
            print('    round %d (%5.1f tiles): ' % (rd, cnt) + '  '.join('%s %.2f' % (k, v) for k, v in vals.items())
                  + '  | loop per half tile %.3f' % (vals.get('loop_g1', 0.0) / H))
    return 0


if __name__ == '__main__':
    raise SystemExit(main())

"""Phase stamps of the persistent blocked attention kernel (lab build of csrc/attention_blk16.hip with -DATT16_STAMPS, tools/r5_attn_stamps.sh):
s_memtime of every wave of workgroup 100 at: barrier passed / next item's DMA + Q loads issued / S^T done / softmax done / P V done / stores
issued / next item's operands landed (vmcnt).  Shader cycles relative to the first stamp of the workgroup.   python tools/attn_stamps.py [N]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: F401
import torch
from whmr_amd import _lib as L
L.LIB_PATH = os.path.join(ROOT, 'tools', 'lab', 'libwhmr_hip_att16_stamps.so')
lib = L.lib()
lib.whmr_debug_att16_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 196
B = 64
qkv = torch.randn(B, N, 2304, device=dev).bfloat16()
qb = L.to_blocked(qkv.view(B * N, 2304))
ob = torch.empty((B * N + 31) // 32, 96, 32, 8, device=dev, dtype=torch.bfloat16)
for _ in range(5):
    L.attention_blk(qb, ob, B, N, 12, 0.125)
torch.cuda.synchronize()
buf = torch.zeros(4 * 16 * 8, dtype=torch.int64, device=dev)
for abl, name in ((0, 'full'), (2, 'no operand traffic after the first item'), (4, 'no arithmetic after the first item')):
    buf.zero_()
    lib.whmr_debug_att16_stamps(buf.data_ptr())
    L.attention_set_variant(1 | abl)
    L.attention_blk(qb, ob, B, N, 12, 0.125)
    torch.cuda.synchronize()
    lib.whmr_debug_att16_stamps(None)
    L.attention_set_variant(1)
    d = buf.cpu().view(4, 16, 8)
    t0 = d[0, :, 0][d[0, :, 0] > 0].min().item()
    print('# %s; N = %d: wave x [barrier passed | DMA issued | S^T done | softmax done | PV done | stores issued | next operands landed], shader cycles since the workgroup passed its first barrier' % (name, N))
    for it in range(4):
        if d[it, 0, 0] == 0:
            continue
        for w in range(16):
            if d[it, w, 0] == 0:
                continue
            print('item %d wave %2d: ' % (it, w) + ' '.join('%7d' % (int(v) - t0 if v else -1) for v in d[it, w, :7]))

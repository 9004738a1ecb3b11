"""ViT-B backbone training step (forward with saved activations + hand-driven backward) on one GPU: ms / step and img/s."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd.models.pose_vit import ViT
from whmr_amd.parallel import GradReducer
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
m = ViT(img_size=224, qkv_bias=True, numerics='bf16').to(dev).train()
red = GradReducer(m.parameters())
x = torch.randn(B, 3, 224, 224, device=dev)
G = torch.randn(B, 768, 14, 14, device=dev)
def step():
    for p in m.parameters():
        p.grad = None
    out = m(x)
    (out * G).sum().backward()
    red.finish()
for _ in range(2):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
fl = 3 * 34.94e9 * B
print('ViT-B 224 B=%d bf16 train step (fwd + bwd, no optimizer): %.2f ms  %.0f img/s  %.0f TFLOP/s (3x forward FLOPs)' % (B, dt * 1e3, B / dt, fl / dt / 1e12))
m.eval()
with torch.no_grad():
    for _ in range(3): m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): m(x)
    torch.cuda.synchronize()
print('inference forward for scale: %.2f ms' % ((time.perf_counter() - t0) / steps * 1e3))
print('peak memory: %.1f GB' % (torch.cuda.max_memory_allocated() / 2**30))
# A/B: weight-gradient branch on a side stream
if len(sys.argv) > 3 and sys.argv[3] == 'ab':
    from whmr_amd.train import vit_autograd as VA
    m.train()
    for rnd in range(3):
        for ov in (False, True):
            VA.OVERLAP_DW = ov
            for _ in range(2): step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(steps): step()
            torch.cuda.synchronize()
            print('round %d: dW on a side stream = %-5s  %.2f ms / step' % (rnd, ov, (time.perf_counter() - t0) / steps * 1e3), flush=True)
if len(sys.argv) > 3 and sys.argv[3] == 'tn':
    from whmr_amd.train import vit_autograd as VA
    m.train()
    for rnd in range(3):
        for tn in (False, True):
            VA.USE_TN = tn
            for _ in range(2): step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(steps): step()
            torch.cuda.synchronize()
            print('round %d: TN weight-gradient GEMM = %-5s  %.2f ms / step' % (rnd, tn, (time.perf_counter() - t0) / steps * 1e3), flush=True)

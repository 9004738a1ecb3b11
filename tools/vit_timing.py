"""Quick on-box timing of the ViT-B forward and its GEMM shapes (development aid, not the bench contract)."""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd import _lib as L
from whmr_amd.models.pose_vit import ViT

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64


def timeit(fn, n=20, w=3):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


M = B * 196
for (N, K) in [(768, 768), (2304, 768), (3072, 768), (768, 3072)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for glds in (True, False):
        ms = timeit(lambda: L.gemm(a, w, out, glds=glds))
        print('gemm M=%d N=%d K=%d glds=%d: %.3f ms  %.1f TF' % (M, N, K, glds, ms, 2.0 * M * N * K / ms / 1e9))
    ms = timeit(lambda: torch.matmul(a, w.t()))
    print('   torch(hipBLASLt) same shape: %.3f ms  %.1f TF' % (ms, 2.0 * M * N * K / ms / 1e9))

qkv = torch.randn(B, 196, 2304, device=dev).bfloat16()
att = torch.empty(B, 196, 768, device=dev, dtype=torch.bfloat16)
for var in (0, 1, 0, 1):
    L.attention_set_variant(var)
    ms = timeit(lambda: L.attention(qkv, att, B, 196, 12, 64, 0.125))
    print('attention bf16 (chunked=%d): %.3f ms  %.1f TF' % (var, ms, 4.0 * B * 12 * 196 * 196 * 64 / ms / 1e9))
t = torch.randn(M, 768, device=dev)
h = torch.empty(M, 768, device=dev, dtype=torch.bfloat16)
g = torch.ones(768, device=dev)
ms = timeit(lambda: L.layernorm(t, g, g, h, 1e-6))
print('layernorm: %.3f ms  %.1f GB/s' % (ms, M * 768 * 6 / ms / 1e6))

m = ViT(img_size=224, qkv_bias=True, numerics='bf16').to(dev).eval()
x = torch.randn(B, 3, 224, 224, device=dev)
def fwd_ms():
    return min(timeit(lambda: m(x), n=10) for _ in range(3))
for rnd in range(3):
    for var in (0, 1):
        L.attention_set_variant(var)
        print('  round %d attention chunked=%d: %.3f ms per forward' % (rnd, var, fwd_ms()), flush=True)
# in-model A/B (same process, same box): per-shape tile overrides, interleaved with the default
for slot, name, tiles in ((0, 'qkv', (257, 259, 256, 320, 192)), (1, 'proj', (192, 257, 259, 128, 64)), (2, 'fc1', (320, 257, 259, 192)),
                          (3, 'fc2', (192, 257, 259, 320, 128))):
    res = []
    for t in tiles:
        L.set_option(100 + slot, 0); base = fwd_ms()
        L.set_option(100 + slot, t); res.append('%d: %+.0f us' % (t, (fwd_ms() - base) * 1e3))
    L.set_option(100 + slot, 0)
    print('  %-4s tile override vs chooser (per forward): %s' % (name, '  '.join(res)), flush=True)
ms = min(timeit(lambda: m(x), n=10) for _ in range(5))
print('ViT-B 224 B=%d bf16 forward: %.3f ms  %.0f img/s  %.1f TF' % (B, ms, B / ms * 1e3, 34.94e9 * B / ms / 1e9))

"""Run the four ViT-B GEMM shapes through torch.matmul (hipBLASLt) so that a rocprofv3 kernel trace shows which Tensile kernels it picks."""
import torch
dev = torch.device('cuda:0')
M = 12544
for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    a = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16()
    for _ in range(12): c = torch.nn.functional.linear(a, w)
    torch.cuda.synchronize()

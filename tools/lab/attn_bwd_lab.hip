// Where does a workgroup of the bf16 attention backward spend its time?  attention.hip compiled in with -DATT_LAB=1 (s_memtime stamps of workgroup 300),
// batch 64 x 12 heads, N = 192:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DATT_LAB=1 tools/lab/attn_bwd_lab.hip -o tools/lab/attn_bwd_lab
#include "../../w-hmr_amd/csrc/attention.hip"
#include <stdio.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main() {
    const int B = 64, N = 192, H = 12, C = H * 64;
    const size_t nq = (size_t)B * N * 3 * C, no = (size_t)B * N * C;
    std::vector<uint16_t> hq(nq), ho(no);
    std::vector<float> hd(no), hl((size_t)B * H * N);
    srand(7);
    auto bf = [](float x) { uint32_t u; memcpy(&u, &x, 4); return (uint16_t)(u >> 16); };
    for (auto& v : hq) v = bf((rand() & 0xffff) / 65536.f - 0.5f);
    for (auto& v : ho) v = bf((rand() & 0xffff) / 65536.f - 0.5f);
    for (auto& v : hd) v = (rand() & 0xffff) / 65536.f - 0.5f;
    for (auto& v : hl) v = 3.0f + (rand() & 0xff) / 256.f;
    void *q, *o, *dq; float *d, *l;
    CK(hipMalloc(&q, nq * 2)); CK(hipMalloc(&o, no * 2)); CK(hipMalloc(&dq, nq * 2)); CK(hipMalloc(&d, no * 4)); CK(hipMalloc(&l, hl.size() * 4));
    CK(hipMemcpy(q, hq.data(), nq * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(o, ho.data(), no * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d, hd.data(), no * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(l, hl.data(), hl.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) whmr_attention_bwd(q, o, d, l, dq, B, N, H, 64, 0.125f, 0);
    CK(hipEventRecord(e0));
    for (int w = 0; w < 10; ++w) whmr_attention_bwd(q, o, d, l, dq, B, N, H, 64, 0.125f, 0);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("attention backward, B 64 x H 12, N 192: %.1f us per launch\n", ms * 100.f);
#if ATT_LAB
    unsigned long long st[8 * 16];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(att_lab_stamps), sizeof(st)));
    printf("cycles from kernel entry (workgroup 300): [1 staging + K^T frags done] step 2: [2 start] [3 S^T / dP^T + exp] [4 dQ MFMAs] [5 dQ accumulated] [6 S / dP + exp] [7 dV / dK MFMAs] [8 barrier]  [9 loop end] [10 kernel end]\n");
    for (int w = 0; w < 6; ++w) {
        printf("  wave %d:", w);
        for (int i = 1; i <= 10; ++i) printf(" %7lld", (long long)(st[w * 16 + i] - st[w * 16]));
        printf("\n");
    }
#endif
    return 0;
}

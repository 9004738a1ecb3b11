// Where does a step of the TN weight-gradient main loop go?  Times the grouped launch of one ViT-B layer's four products (gemm_tn.hip compiled in,
// -DTN_LAB=n ablations: 1 no MFMAs, 2 no fragment reads, 4 no LDS-DMA -- wrong results, timing only) at K and 2 K: the difference is the main loop.
//   for n in 0 1 2 4 3 6; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTN_LAB=$n tools/lab/tn_lab.hip -o tools/lab/tn_lab_$n; done
#include "../../w-hmr_amd/csrc/gemm_tn.hip"
#include <stdio.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int Kmax = 24576, D = 768;
    const int shapes[4][2] = {{768, 3072}, {3072, 768}, {768, 768}, {2304, 768}};
    std::vector<uint16_t> h((size_t)Kmax * 3072);
    srand(5);
    const bool zeros = argc > 1 && argv[1][0] == 'z';          // zero operands: what the power limit costs
    if (!zeros) for (auto& v : h) { float x = ((rand() & 0xffff) / 65536.f - 0.5f); uint32_t u; memcpy(&u, &x, 4); v = (uint16_t)(u >> 16); }
    whmr_tn_item it[4];
    for (int i = 0; i < 4; ++i) {
        void *a, *b; float *c, *db;
        CK(hipMalloc(&a, (size_t)Kmax * shapes[i][0] * 2)); CK(hipMalloc(&b, (size_t)Kmax * shapes[i][1] * 2));
        CK(hipMalloc(&c, (size_t)shapes[i][0] * shapes[i][1] * 4)); CK(hipMalloc(&db, shapes[i][0] * 4));
        CK(hipMemcpy(a, h.data(), (size_t)Kmax * shapes[i][0] * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(b, h.data() + (shapes[i][1] < 3072 ? 12344 : 0), (size_t)Kmax * shapes[i][1] * 2, hipMemcpyHostToDevice));
        it[i] = whmr_tn_item{a, shapes[i][0], b, shapes[i][1], c, shapes[i][1], db, shapes[i][0], shapes[i][1]};
    }
    void* ws; const long wsb = 128l << 20;
    CK(hipMalloc(&ws, wsb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float t[2];
    for (int rep = 0; rep < 2; ++rep)
        for (int kk = 0; kk < 2; ++kk) {
            const int K = kk ? Kmax : Kmax / 2;
            for (int w = 0; w < 3; ++w) whmr_gemm_tn_bf16_group(it, 4, K, ws, wsb, 0);
            CK(hipEventRecord(e0));
            for (int w = 0; w < 10; ++w) whmr_gemm_tn_bf16_group(it, 4, K, ws, wsb, 0);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&t[kk], e0, e1));
            t[kk] *= 100.f;                                       // us per call (kernel + reduce)
        }
    const int nsp = getenv("TN_LAB_SPLITS") ? atoi(getenv("TN_LAB_SPLITS")) : 2;
    const double steps = (Kmax / 2) / 32 / nsp;                  // extra steps per block between the two runs
    printf("TN_LAB=%d%s  K=12288: %.1f us  K=24576: %.1f us  main loop %.3f us/step (%.0f TF/s over %d blocks)\n", TN_LAB, zeros ? " zeros" : "", t[0], t[1], (t[1] - t[0]) / steps,
           108 * nsp * 2.0 * 256 * 256 * 32 / ((t[1] - t[0]) / steps * 1e-6) / 1e12, 108 * nsp);
    if (TN_LAB & 8) {                                            // stamps of one block (logical block of physical block 8): see gemm_tn.hip
        std::vector<unsigned long long> st(64);
        CK(hipMemcpy(st.data(), (char*)ws + (96l << 20), 512, hipMemcpyDeviceToHost));
        {
            const size_t o = 0;
            printf("stamps (clocks after the step's barrier): wave: [1 MFMA done (group 0)] [2 reads issued] [3 DMA issued] [4 vmcnt] [5 lgkmcnt] [6 MFMA done (group 1)] [7 next barrier]\n");
            for (int w = 0; w < 8; ++w) {
                printf("  wave %d:", w);
                for (int i = 1; i < 8; ++i) printf(" %6lld", st[o + w * 8 + i] ? (long long)(st[o + w * 8 + i] - st[o + w * 8]) : -1ll);
                printf("\n");
            }
        }
    }
    return 0;
}

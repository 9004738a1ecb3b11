// Split-bf16 ("bf16x3") instantiations of the blocked-layout GEMM (gemm_blk_impl.h, X3 = true): the parity-grade numerics of the ViT
// inference path (vit.py:61-140,157 are fp32 in the reference; BASELINE north star: vertices within 1e-4 at MFMA rates).  Its own
// translation unit so that the 32 extra kernels compile beside the bf16 ones.
#include "gemm_blk_impl.h"

int blk_x3_launch_tile(const whmr_gemm_blk_desc* pp, int tile, void* stream, int sched) {
    return blk_launch_tile<true>(*pp, tile, (hipStream_t)stream, sched == 2 ? 2 : 1);
}

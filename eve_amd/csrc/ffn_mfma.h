// bf16 MFMA grouped-GEMM path of the expert FFN (implemented in ffn_mfma.hip).
#pragma once
#include "common.h"

namespace eve {

bool ffn_mfma_supported(int M, int I);

int ffn_fwd_mfma(const void* x, const void* w1, const void* b1, const void* w2, const void* b2, void* z, void* a,
                 void* y, const int32_t* rows, int G, int E_local, int R, int M, int I, int act, bool shared, hipStream_t st);

int ffn_bwd_mfma(const void* dy, const void* x, const void* z, const void* a, const void* w1, const void* w2, void* dz,
                 void* dx, void* dw1, void* db1, void* dw2, void* db2, const int32_t* rows, int G, int E_local, int R,
                 int M, int I, int act, bool shared, void* workspace, size_t workspace_bytes, hipStream_t st);

// bytes of scratch ffn_bwd_mfma needs for these shapes: fp32 partials of the bias gradient db1, plus the split-K
// partials of the weight gradients when few experts are local
size_t ffn_bwd_mfma_workspace_bytes(int G, int E_local, int R, int M, int I);

}  // namespace eve

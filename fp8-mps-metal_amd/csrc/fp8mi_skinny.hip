// 2 <= M <= 16 weight-streaming kernel (placeholder until the MFMA skinny
// kernel lands): reports "unsupported" so the dispatcher uses the tile GEMM.
#include "fp8mi_common.h"

bool fp8mi_skinny_supported(const MMParams &) { return false; }
int fp8mi_launch_skinny(const MMParams &, hipStream_t) { return FP8MI_E_UNSUPPORTED; }

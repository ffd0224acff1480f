// Fused non-causal Performer attention for the inference path (see performer_attn.hip).
#pragma once
#include "common.h"

// ctx[(b*8+h)][j][e] (266 x 64, dense) and ks[(b*8+h)][j] (row stride 268) from k, v (B*Fr, 512) and P (266, 64)
void performer_kv(hipStream_t st, const float* k, const float* v, const float* P, int B, int Fr, float* ctx, float* ks);
// attn (B*Fr, 512) from q (B*Fr, 512), P, ctx, ks
void performer_q(hipStream_t st, const float* q, const float* P, const float* ctx, const float* ks, int B, int Fr,
                 float* attn);

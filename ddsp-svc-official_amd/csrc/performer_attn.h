// Fused non-causal Performer attention for the inference path (see performer_attn.hip).
#pragma once
#include "common.h"

// Feature rows of the fused kernels are padded from 266 to 272 (17 tiles of 16).
constexpr int PERFORMER_LDJ = 272;
// Floats per (utterance, head) of the `ks` buffer: ks[272], then per feature tile the column sums of its rows of ctx
// (17 x 64) and the sum of its ks (17), padded.
constexpr int PERFORMER_KS_STRIDE = 272 + 17 * 64 + 32;
// ctxT[(b*8+h)][e][j] (64 x 272: the context matrix TRANSPOSED, pad features zero) and ks[(b*8+h)] (PERFORMER_KS_STRIDE floats each) from k, v (B*Fr, 512) and P (266, 64)
void performer_kv(hipStream_t st, const float* k, const float* v, const float* P, int B, int Fr, float* ctxT, float* ks);
// attn (B*Fr, 512) from q (B*Fr, 512), P, ctxT, ks
void performer_q(hipStream_t st, const float* q, const float* P, const float* ctxT, const float* ks, int B, int Fr,
                 float* attn);

// Fused non-causal Performer attention for the inference path (see performer_attn.hip).
#pragma once
#include "common.h"

// Feature rows of the fused kernels are padded from 266 to 272 (17 tiles of 16).
constexpr int PERFORMER_LDJ = 272;
// Floats per (utterance, head) of the `ks` buffer: ks[272], then per feature tile the column sums of its rows of ctx
// (17 x 64) and the sum of its ks (17), padded.
constexpr int PERFORMER_KS_STRIDE = 272 + 17 * 64 + 32;
// ---- split-bf16 variants (performer_attn_bf16.hip): 32-wide tiles, features padded to 288 = 9 tiles ------------------------
constexpr int PERFORMER_NJT32 = 9;
constexpr int PERFORMER_OFF_CPART32 = 288;                 // ks[288] | column-sum parts [9][64] | ks-sum parts [9]
constexpr int PERFORMER_OFF_KPART32 = 288 + 9 * 64;
static_assert(PERFORMER_OFF_KPART32 + 9 <= PERFORMER_KS_STRIDE, "ks record too small for the 32-wide layout");
constexpr int PERFORMER_CTXS_FLOATS = 9 * 512 * 4;        // per (utterance, head): ctx as bf16 hi/lo pieces in operand order
constexpr int PERFORMER_P3_BYTES = 9 * 768 * 16;          // per layer: the scaled projection matrix as three bf16 pieces
// p3 <- pieces of dn*log2(e)*P (266, 64), once per forward and layer
void performer_p3(hipStream_t st, const float* P0, const float* P1, const float* P2, void* p3);   // P1, P2 may be null
void performer_kv_bf16(hipStream_t st, const float* k, const float* v, const void* p3, int B, int Fr, float* ctxS, float* ks,
                       int ablate = 0);
void performer_q_bf16(hipStream_t st, const float* q, const void* p3, const float* ctxS, const float* ks, int B, int Fr,
                      float* attn, int ablate = 0, int out_split = 0);   // out_split: attn as bf16 hi/lo groups (gemm A_split)
// both sides in one kernel (round 3): ctx and ks stay in the LDS of the (utterance, head)'s workgroup
bool performer_fused_enabled();
hipError_t performer_fused_bf16(hipStream_t st, const float* q, const float* k, const float* v, const void* p3, int B, int Fr,
                                float* attn, int out_split = 0);
// ctxT[(b*8+h)][e][j] (64 x 272: the context matrix TRANSPOSED, pad features zero) and ks[(b*8+h)] (PERFORMER_KS_STRIDE floats each) from k, v (B*Fr, 512) and P (266, 64)
void performer_kv(hipStream_t st, const float* k, const float* v, const float* P, int B, int Fr, float* ctxT, float* ks);
// attn (B*Fr, 512) from q (B*Fr, 512), P, ctxT, ks
void performer_q(hipStream_t st, const float* q, const float* P, const float* ctxT, const float* ks, int B, int Fr,
                 float* attn);
// causal linear attention in chunks of 16 frames (`c: true`, pcmer.py:170-188), inference: attn (B*Fr, 512) from q, k, v and P
void performer_causal(hipStream_t st, const float* q, const float* k, const float* v, const float* P, int B, int Fr, float* attn);

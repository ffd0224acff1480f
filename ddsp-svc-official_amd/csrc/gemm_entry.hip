// Plain GEMM entry point of the fp32-MFMA kernel: unit tests of the building block and tile/schedule A/B timing.
#include "gemm_ws.h"
#include "gemm_ln.h"

extern "C" int ddsp_gemm_f32(ddsp_ctx* ctx, void* stream, const float* A, int64_t lda, int a_k_contig, const float* B,
                             int64_t ldb, int b_k_contig, const float* bias, float* C, int64_t ldc, int M, int N, int K,
                             int tile, int variant) {
    DDSP_REQUIRE(ctx, ctx && A && B && C, "ddsp_gemm_f32: null argument");
    DDSP_REQUIRE(ctx, M >= 1 && N >= 1 && K >= 1 && lda >= 1 && ldb >= 1 && ldc >= N, "ddsp_gemm_f32: bad shape");
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    gemm::Args g = gemm::make(A, lda, B, ldb, M, N, K);
    gemm::EpiStore e{C, ldc, bias, 1, 0, 0};
    ddsp_prof_begin(ctx, st, PF_OTHER);
#define GO(BM, BN, AK, BK_, NW) gemm::launch_tile<BM, BN, AK, BK_, gemm::A_PLAIN, gemm::EpiStore, NW>(st, g, 1, e)
#define TILE(AK, BK_)                                                        \
    if (tile == 0) gemm::launch<AK, BK_, gemm::A_PLAIN>(st, g, 1, e);        \
    else if (tile == 1) GO(64, 64, AK, BK_, 4);                              \
    else if (tile == 2) GO(64, 128, AK, BK_, 4);                             \
    else if (tile == 3) GO(128, 128, AK, BK_, 4);                            \
    else if (tile == 4) GO(128, 128, AK, BK_, 8);                            \
    else if (tile == 5) GO(128, 64, AK, BK_, 8);                             \
    else GO(256, 128, AK, BK_, 8);
    if (tile >= 70) {
        // wave-specialised kernel (gemm_ws.h).  variant: 0 fp32 products, 3 split-bf16 (split in the loop), 8 TIMING ONLY (the
        // operands are read as if already split: numbers meaningless); + 16: residual epilogue (C += ...), + 32: gated pair
        // epilogue (C gets N / 2 columns); ablations (timing only): + 256 no MFMA, + 512 no DMA, + 1024 no stores
        DDSP_REQUIRE(ctx, a_k_contig && b_k_contig && gemm::ws_ok(g, 5, (variant & 48) == 16), "ddsp_gemm_f32: ws tiles need row-major A, [N][K] B, K % 32 == 0, K >= 256, aligned rows");
        const int math = variant & 15, kind = variant & 48, abl = variant >> 8;   // (abl up to 511)
        DDSP_REQUIRE(ctx, (math == 0 || math == 3 || math == 8) && ((uintptr_t)C % 16) == 0 && ldc % 4 == 0 && N % 64 == 0, "ddsp_gemm_f32: ws variant");
        hipError_t he = hipSuccess;
#define WS(BM, BN, NS, MATH)                                                                                                   \
        do {                                                                                                                   \
            if (kind == 16) { DDSP_REQUIRE(ctx, false, "ws residual: tile 75"); }                                              \
            else if (kind == 32) { gemm::WsGlu e2{C, ldc, bias}; DDSP_REQUIRE(ctx, bias && N % BN == 0 && BN == 128, "ws gated pair needs a bias and whole 128-column tiles"); \
                if constexpr (BN == 128) he = gemm::launch_ws_one<BM, BN, gemm::WsGlu, NS, MATH, true>(st, g, e2); }           \
            else { gemm::WsStore e2{C, ldc, bias};                                                                             \
                he = (N % BN == 0) ? gemm::launch_ws_one<BM, BN, gemm::WsStore, NS, MATH, true>(st, g, e2)                     \
                                   : gemm::launch_ws_one<BM, BN, gemm::WsStore, NS, MATH, false>(st, g, e2); }                 \
        } while (0)
#define WSM(BM, BN, NS)                                                     \
        do {                                                                \
            if (math == 0) WS(BM, BN, NS, 0);                               \
            else if (math == 3) WS(BM, BN, NS, 3);                          \
            else WS(BM, BN, NS, 8);                                         \
        } while (0)
        g.xcd = 1;
        gemm::WsStore e0{C, ldc, bias};
        if (abl) {
            DDSP_REQUIRE(ctx, tile == 70 && kind == 0 && N % 128 == 0, "ws ablations: tile 70, plain store");
            if (abl == 2) he = gemm::launch_ws_one<128, 128, gemm::WsStore, 4, 8, true, 2>(st, g, e0);
            else if (abl == 4) he = gemm::launch_ws_one<128, 128, gemm::WsStore, 4, 8, true, 4>(st, g, e0);
            else if (abl == 8) he = gemm::launch_ws_one<128, 128, gemm::WsStore, 4, 8, true, 8>(st, g, e0);
            else if (abl == 16) he = gemm::launch_ws_one<128, 128, gemm::WsStore, 4, 8, true, 16>(st, g, e0);
            else if (abl == 18) he = gemm::launch_ws_one<128, 128, gemm::WsStore, 4, 8, true, 18>(st, g, e0);
            else if (abl == 32) he = gemm::launch_ws_one<128, 128, gemm::WsStore, 4, 8, true, 32>(st, g, e0);
            else if (abl == 34) he = gemm::launch_ws_one<128, 128, gemm::WsStore, 4, 8, true, 34>(st, g, e0);
            else if (abl == 50) he = gemm::launch_ws_one<128, 128, gemm::WsStore, 4, 8, true, 50>(st, g, e0);
            else if (abl == 64) he = gemm::launch_ws_one<128, 128, gemm::WsStore, 4, 8, true, 64>(st, g, e0);
            else if (abl == 178) he = gemm::launch_ws_one<128, 128, gemm::WsStore, 4, 8, true, 178>(st, g, e0);   // no DMA, lgkm, epilogue, LDS reads
            else if (abl == 306) he = gemm::launch_ws_one<128, 128, gemm::WsStore, 4, 8, true, 306>(st, g, e0);   // no DMA, lgkm, epilogue, barriers
            else if (abl == 434) he = gemm::launch_ws_one<128, 128, gemm::WsStore, 4, 8, true, 434>(st, g, e0);   // MFMAs only
            else return ddsp_fail(ctx, DDSP_ERR_ARG, "ddsp_gemm_f32", "unknown ws ablation");
        }
        else if (tile == 70) WSM(128, 128, 4);
        else if (tile == 72) { DDSP_REQUIRE(ctx, math == 8 && kind == 0 && N % 128 == 0, "tile 72: timing only"); he = gemm::launch_ws_one<128, 128, gemm::WsStore, 3, 8, true>(st, g, e0); }
        else if (tile == 75) { DDSP_REQUIRE(ctx, math == 8 && kind != 32 && N % 64 == 0, "tile 75: timing only");
            if (kind == 16) { gemm::WsResidual e2{C, C, ldc, bias}; he = gemm::launch_ws_one<128, 64, gemm::WsResidual, 5, 8, true>(st, g, e2); }
            else he = gemm::launch_ws_one<128, 64, gemm::WsStore, 5, 8, true>(st, g, e0); }
        else return ddsp_fail(ctx, DDSP_ERR_ARG, "ddsp_gemm_f32", "unknown ws tile");
#undef WSM
#undef WS
        DDSP_HIP(ctx, he);
    } else if (tile >= 10) {
        DDSP_REQUIRE(ctx, a_k_contig && b_k_contig && gemm::dma_ok(g), "ddsp_gemm_f32: DMA tiles need row-major A, [N][K] B, K % 32 == 0, aligned rows");
        if (tile == 10) gemm::launch_dma<128, 64>(st, g, 1, e);
        else if (tile == 11) gemm::launch_dma<128, 128>(st, g, 1, e);
        else if (tile == 12) gemm::launch_dma<256, 128>(st, g, 1, e);
        else if (tile == 13) gemm::launch_dma<128, 128, gemm::EpiStore, 2>(st, g, 1, e);
        else if (tile == 14) gemm::launch_dma<128, 64, gemm::EpiStore, 2>(st, g, 1, e);
        else if (tile == 15) gemm::launch_dma<64, 64, gemm::EpiStore, 3, 0, 4>(st, g, 1, e);   // 4 waves, 64x64
        else if (tile == 30) gemm::launch_dma<128, 128, gemm::EpiStore, 2, 0, 8, gemm::A_PLAIN, 3>(st, g, 1, e);  // split-bf16 x3 (experiment)
        else if (tile == 31) gemm::launch_dma<128, 128, gemm::EpiStore, 2, 0, 8, gemm::A_PLAIN, 6>(st, g, 1, e);  // split-bf16 x6 (experiment)
        else if (tile == 32) gemm::launch_dma<64, 64, gemm::EpiStore, 3, 0, 4, gemm::A_PLAIN, 3>(st, g, 1, e);    // split-bf16 x3, 64x64 / 4 waves
        else if (tile == 16) gemm::launch_dma<64, 128, gemm::EpiStore, 3, 0, 4>(st, g, 1, e);  // 4 waves, 64x128
        // 50..53: TIMING ONLY - the operands are read as if already split (mode 8), the numbers are meaningless
        else if (tile == 50) gemm::launch_dma<128, 128, gemm::EpiStore, 2, 0, 8, gemm::A_PLAIN, 8>(st, g, 1, e);
        else if (tile == 51) gemm::launch_dma<128, 128, gemm::EpiStore, 2, 0, 4, gemm::A_PLAIN, 8>(st, g, 1, e);
        else if (tile == 52) gemm::launch_dma<128, 256, gemm::EpiStore, 2, 0, 8, gemm::A_PLAIN, 8>(st, g, 1, e);
        else if (tile == 53) gemm::launch_dma<64, 64, gemm::EpiStore, 3, 0, 4, gemm::A_PLAIN, 8>(st, g, 1, e);
        else if (tile == 54) gemm::launch_dma<128, 64, gemm::EpiStore, 3, 0, 8, gemm::A_PLAIN, 8>(st, g, 1, e);
        else if (tile == 55) gemm::launch_dma<128, 128, gemm::EpiStore, 3, 0, 8, gemm::A_PLAIN, 8>(st, g, 1, e);
        else if (tile == 58) gemm::launch_dma<128, 128, gemm::EpiStore, 2, 4, 8, gemm::A_PLAIN, 8>(st, g, 1, e);   // 50 without epilogue stores
        else if (tile == 59) gemm::launch_dma<128, 128, gemm::EpiStore, 2, 1, 8, gemm::A_PLAIN, 8>(st, g, 1, e);   // 50 without MFMAs
        else if (tile == 60) gemm::launch_dma<128, 128, gemm::EpiStore, 2, 2, 8, gemm::A_PLAIN, 8>(st, g, 1, e);   // 50 without DMA
        else if (tile == 56) gemm::launch_dma<64, 128, gemm::EpiStore, 3, 0, 4, gemm::A_PLAIN, 8>(st, g, 1, e);
        else if (tile == 57) gemm::launch_dma<64, 128, gemm::EpiStore, 2, 0, 4, gemm::A_PLAIN, 8>(st, g, 1, e);
        else if (tile == 40) gemm::launch_dma<128, 128, gemm::EpiStore, 2, 0, 4, gemm::A_PLAIN, 3>(st, g, 1, e);  // split-bf16 x3, 128x128 on 4 waves (64x64 per wave)
        else if (tile == 41) gemm::launch_dma<128, 256, gemm::EpiStore, 2, 0, 8, gemm::A_PLAIN, 3>(st, g, 1, e);  // split-bf16 x3, 128x256 on 8 waves (64x64 per wave)
        else if (tile == 33) gemm::launch_dma<128, 64, gemm::EpiStore, 3, 0, 8, gemm::A_PLAIN, 3>(st, g, 1, e);   // split-bf16 x3, 128x64
        else if (tile == 34) gemm::launch_dma<64, 128, gemm::EpiStore, 3, 0, 4, gemm::A_PLAIN, 3>(st, g, 1, e);   // split-bf16 x3, 64x128 / 4 waves
        else if (tile == 35) gemm::launch_dma<128, 128, gemm::EpiStore, 2, 1, 8, gemm::A_PLAIN, 3>(st, g, 1, e);  // 30 without split + MFMA (timing only)
        else if (tile == 36) gemm::launch_dma<128, 128, gemm::EpiStore, 2, 2, 8, gemm::A_PLAIN, 3>(st, g, 1, e);  // 30 without DMA
        else if (tile == 37) gemm::launch_dma<128, 128, gemm::EpiStore, 2, 10, 8, gemm::A_PLAIN, 3>(st, g, 1, e); // 30 without DMA and barrier
        else if (tile == 38) gemm::launch_dma<128, 128, gemm::EpiStore, 2, 14, 8, gemm::A_PLAIN, 3>(st, g, 1, e); // 30: split + MFMA + LDS reads only
        else if (tile == 20) gemm::launch_dma<128, 64, gemm::EpiStore, 3, 1>(st, g, 1, e);   // no MFMA (timing only)
        else if (tile == 21) gemm::launch_dma<128, 64, gemm::EpiStore, 3, 2>(st, g, 1, e);   // no DMA (timing only)
        else if (tile == 22) gemm::launch_dma<128, 64, gemm::EpiStore, 3, 6>(st, g, 1, e);   // no DMA, no stores
        else if (tile == 23) gemm::launch_dma<128, 64, gemm::EpiStore, 3, 10>(st, g, 1, e);  // no DMA, no barrier
        else gemm::launch_dma<128, 64, gemm::EpiStore, 3, 14>(st, g, 1, e);                   // MFMA + LDS reads only
    } else if (a_k_contig && b_k_contig) {
        TILE(true, true)
    } else if (a_k_contig && !b_k_contig) {
        TILE(true, false)
    } else if (!a_k_contig && !b_k_contig) {
        TILE(false, false)
    } else {
        TILE(false, true)
    }
    ddsp_prof_end(ctx, st, 2.0 * M * N * (double)K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}

// x = res + A W^T + bias (M x 256) and y = LayerNorm(x) * gamma + beta in one launch (gemm_ln.h); A (M x K) and W (256 x K) in
// the pre-split operand layout.  The building block ddsp_unit2ctrl_fwd runs for its out-projection / pw2 layers at large batches.
extern "C" int ddsp_gemm_res_ln(ddsp_ctx* ctx, void* stream, const float* A_split, const float* W_split, const float* bias,
                                const float* res, const float* gamma, const float* beta, int M, int K, float* X, float* Y,
                                int y_split) {
    DDSP_REQUIRE(ctx, ctx && A_split && W_split && bias && res && gamma && beta && X && Y, "ddsp_gemm_res_ln: null argument");
    gemm::LnArgs a{A_split, W_split, K, K, M, K, bias, res, X, gamma, beta, Y, (y_split & 1) ? 1 : 0};   // y_split bit 1 (value 2): A is fp32 rows
    DDSP_REQUIRE(ctx, M >= 1 && K >= 64 && gemm::res_ln_ok(a), "ddsp_gemm_res_ln: K % 32 == 0, K >= 64, 16-byte aligned operands");
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    ddsp_prof_begin(ctx, st, PF_OTHER);
    DDSP_HIP(ctx, gemm::launch_res_ln(st, a, (y_split & 2) == 0));
    ddsp_prof_end(ctx, st, 2.0 * M * 256.0 * K, 4.0 * M * (K + 4.0 * 256));
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}

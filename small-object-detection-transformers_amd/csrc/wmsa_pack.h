// Byte offsets inside the bf16 parameter pack of a fused W-MSA block (sodt_wmsa_pack; layout defined by WL<bf16> in
// wmsa_common.h, which static_asserts these values).  Plain constants so that kernels outside the fused-forward translation
// units (the attention backward that recomputes q / k / v, attention.hip) can address the pack.
#pragma once
namespace wmsa_pack_bf16 {
constexpr int STAGE = 24576;        // bytes per head stage: Wq | Wk | Wv fragments, bias table, q/k/v bias
constexpr int WFRAG = 6144;         // one of Wq_h / Wk_h / Wv_h in MFMA A-fragment order: [k-step 0..5][lane 0..63][8 bf16]:
                                    //   lane (g = lane >> 4, t = lane & 15) holds row 16 h + t, columns 32 kk + 8 g .. + 7
constexpr int BQKV_OFF = 20352;     // f32 bq[16] | bk[16] | bv[16] | (bq x hd^-1/2 x log2 e)[16] of the head
constexpr int HGW_OFF = 372480;     // the contiguous stream of wmsa_hg.hip: [head][q|k|v][WFRAG], Wq there x hd^-1/2 x log2 e
}

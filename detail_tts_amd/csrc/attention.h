// Attention launchers (see attention.hip).
#pragma once
#include "common.h"

namespace dtts {

struct AttnParams {
    const float* qkv = nullptr;   // [B, rows, T]
    long long bs = 0;             // batch stride (floats)
    int cs = 0;                   // row (channel) stride == allocated T
    int q_off = 0, k_off = 0, v_off = 0, head_stride = 0;   // row of (head h, dim c) = off + h*head_stride + c
    float* out = nullptr;         // [B, H*D, T]
    long long o_bs = 0;
    int o_cs = 0;
    const int* lens = nullptr;    // [B] valid length (null -> T)
    int T = 0, B = 0, H = 0, D = 0;
    float scale = 1.f;            // multiplies q.k
    const float* bias_tab = nullptr;   // [H][129]: additive bias by clamp(s-t,-64,64)+64 (already scaled), or null
    int causal = 0;
    // VITS windowed relative-key logits (attentions.py:218-222): band[b,h,t,r], r = (s-t)+W, |s-t|<=W
    const float* band = nullptr;
    int band_w = 0;
    float* ml_out = nullptr;      // optional [B,H,T,2] (running max, denominator) for the rel-value fix-up
    int x3 = 0;                   // split-precision (3 x bf16) kernel (attention_x3.hip): head dim 48 + T5 bias only
    // x3 only: write the output as split-precision planes [B][H*D/8][3][x3_tp][8 bf16] (conv_x3.h layout, column t + 1) instead
    // of fp32 `out`; only columns t < len are written (halo / tail columns keep what the previous writer left: zeros)
    void* out_x3 = nullptr;
    int x3_tp = 0;
    // x3 only: q / k / v come as AttnPlanes images (qkv is ignored) written by the qkv conv
    const void* planes = nullptr;
    // attention_x3b only (set by its launcher): the keys of a (sample, head, query block) split over `ksplit` workgroups, whose waves
    // leave their unnormalised (O, l, m) in kpart; the last wave to arrive (kcount) merges them in split order
    int ksplit = 1;
    float* kpart = nullptr;
    int* kcount = nullptr;
};

// Operand images of the split-precision attention (head dim 48), written by the qkv conv's epilogue (conv_x3.hip) and consumed by
// LDS-DMA / direct fragment loads - no staging arithmetic in the attention kernel.  16-byte chunks of 8 fp16; two planes (split3.h);
// Q carries scale * log2(e) * 16, K and V carry 16.  Per (sample, head):
//   Q  [plane 2][c8 6][Tq]            chunk = 8 channels of one query
//   KV [tile][ K: plane 2 x (c8 6 x key 64) | V: plane 2 x (u 2 x j 2 x hh 2 x channel 48) ]   = the kernel's 24 KiB LDS stage image;
//      V chunk (u, j, hh, c) = keys 32 u + 16 j + 4 hh + {0..3, 8..11} of channel c - the order in which a lane of the 32 x 32 score
//      tile (rows (r & 3) + 8 (r >> 2) + 4 hh) holds its keys, so P feeds the PV product from registers; keys >= len hold zeros
struct AttnPlanes {
    static constexpr int D = 48, KT = 64, TILE_BYTES = 2 * (6 * 64 + 8 * 48) * 16;      // 24576
    __host__ __device__ static inline int tq(int T) { return (T + KT - 1) / KT * KT; }
    __host__ __device__ static inline int nt64(int T) { return (T + KT - 1) / KT; }
    __host__ __device__ static inline size_t q_bytes(int T) { return (size_t)2 * 6 * tq(T) * 16; }
    __host__ __device__ static inline size_t head_bytes(int T) { return q_bytes(T) + (size_t)nt64(T) * TILE_BYTES; }
    __host__ __device__ static inline size_t bytes(int B, int H, int T) { return (size_t)B * H * head_bytes(T); }
};

void launch_flash_attention(const AttnParams& p, hipStream_t stream);
void launch_flash_attention_x3(const AttnParams& p, hipStream_t stream);   // called by launch_flash_attention when p.x3 (fp32 q, k, v)
void launch_flash_attention_x3w(const AttnParams& p, hipStream_t stream);  // ... when the operands are AttnPlanes images (round 2-4 kernel: DTTS_ATTN_KERNEL=w)
void launch_flash_attention_x3b(const AttnParams& p, hipStream_t stream);  // ... the block-skewed kernel (default)
void set_attn_ksplit(int n);      // key ranges per (sample, head, query block) of attention_x3b launches of <= 2 samples that do not fill the CUs: 1 = off, 2 .. 4 (default: up to 4); process-wide
void set_attn_ksplit_cus(int n);  // ... split only while workgroups x S <= n (default 256)
int attn_ksplit();

// VITS relative-position helpers (vqvae/modules/attentions.py:198-239), W = window (4)
//   relk[b,h,t,r] = scale * sum_c q[c,t] * Ek[r][c]
void launch_vits_rel_key(const float* qkv, long long bs, int cs, int q_off, int head_stride, const float* Ek,
                         float* relk, const int* lens, int B, int H, int D, int T, int W, float scale, hipStream_t s);
//   out[b,h*D+c,t] += sum_r softmax(t, t+r-W) * Ev[r][c]    (uses ml from the flash kernel)
void launch_vits_rel_value(const float* qkv, long long bs, int cs, int q_off, int k_off, int head_stride,
                           const float* relk, const float* ml, const float* Ev, float* out, long long o_bs, int o_cs,
                           const int* lens, int B, int H, int D, int T, int W, float scale, hipStream_t s);

}  // namespace dtts

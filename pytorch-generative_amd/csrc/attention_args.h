// attention_args.h — kernel-argument block shared by attention.hip (VALU row-owner kernels, any
// head dims <= 64) and attention_mfma.hip (matrix-core kernels for d_k = d_v = 4).
#pragma once
#include "common.h"

struct PgAttnArgs {
  const float* q; const float* k; const float* v; const float* o; const float* d_o;
  const float* lse2_in;
  float* o_out; float* lse2_out; float* delta; float* dq; float* dk; float* dv;
  int N, heads, L, dk_dim, dv_dim, strict;
  long q_bs, k_bs, v_bs, o_bs, do_bs, dq_bs, dk_bs, dv_bs;
  float scale, scale2;  // 1/sqrt(dk), log2(e)/sqrt(dk)
  int blocks_per_wg;    // 64-row blocks per workgroup (<= 16); waves = ceil(blocks_per_wg / 2)
  int kt, kt2;          // VALU kernels: rows per LDS tile (fwd/dQ resp. dK/dV), multiples of 64,
                        // sized so that a whole (n, head) fits when LDS allows: barriers between
                        // tiles would re-serialise the balanced pairing
  int lp, vec;          // MFMA kernels: LDS plane stride (== 16 mod 64); float4 staging allowed
  // MFMA kernels: the 64-row blocks each wave of the workgroup processes, heaviest first (LPT
  // assignment made on the host so that every wave — and with a wave count that is a multiple of 4,
  // every SIMD — gets the same share of the causal triangle)
  unsigned char bcount[8];
  unsigned char blist[8][16];
};

enum { PG_ATTN_FWD = 0, PG_ATTN_DQ = 1, PG_ATTN_DKV = 2, PG_ATTN_BWD = 3 /* fused dQ + dK + dV (d_k = d_v = 4) */ };

// attention_mfma.hip. Returns 1 if the matrix-core path took the launch, 0 if the shape is not
// covered (caller uses the VALU kernels); launch errors are left for PG_LAUNCH_CHECK.
int pg_attn_mfma_launch(int which, const PgAttnArgs& a, hipStream_t st);

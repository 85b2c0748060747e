// conv_wgrad_b3.hip — convolution weight/bias gradient with its fp32 products evaluated on the bf16
// matrix pipe ("bf16x3", see conv_b3.hip): dy = dh + dm + dl, x = xh + xm + xl (exact 3-way bf16
// splits) and   dy.x = dh.xh + dh.xm + dm.xh + dh.xl + dl.xh + dm.xm   (6 x v_mfma_f32_16x16x32_bf16),
// fp32 accumulation. Same reference call sites as conv_wgrad.hip (the weight/bias outputs of
// aten::convolution_backward behind loss.backward(), trainer.py:180; unmasked taps,
// nn/convolution.py:42). conv_wgrad.hip keeps every shape this file does not take.
//
//   dw[co][ci][t] = sum_{n,r,c} dy[n,co,r,c] * act(x[n,ci,r+dr_t,c+dc_t])      db[co] = sum dy
//
// GEMM view: M = co, N = ci, K = output pixels. One K step = 32 pixels = four "pixel blocks" of 8
// consecutive pixels of one image row (W % 8 == 0, or W % 8 == 4 with a half-full last block): lane (i|j = lane & 15, kg = lane >> 4) holds the
// 8 pixels of block 4 ks + kg for dy channel i (A) / x channel j (B).
// The tap shift lives on the K axis here, and a packed-bf16 fragment cannot be read at an odd
// element offset, so x is staged once per DISTINCT COLUMN SHIFT dc in {-1, 0, +1} ("copies":
// copy_dc[row][c] = x[row][c + dc], zero outside the image — the column halo disappears); the row
// shift dr is a whole number of pixel blocks. LDS entries are 16 bytes (8 bf16):
//   dy: [piece][co tile][pixel block][16 channels]      x: [copy][piece][ci tile][pixel block][16 channels]
// so a fragment read is `scalar base + lane` (ds_read_b128, 256 B per 16 lanes: conflict free) and a
// staging thread (= 8 pixels of one channel, channel fastest across lanes) writes whole entries.
// A workgroup owns 64 co x 32 ci x all taps, walks pixel tiles of TR rows with the next tile's
// loads in flight under the MFMA loop, and writes ONE row of partial sums; conv_wgrad.hip's
// deterministic reduction adds the rows into dw / db.
// Measured (MI355X, 2x2 64 -> 64, N = 512, ELU prologue): 172 us = 100 TF/s algorithmic against 294 us for
// the fp32-MFMA kernel; phase ablation with PG_WB_DBG and the variants that were tried and dropped
// (producer / consumer waves, two half-workgroups in opposite phases): profiles/README.md, round 2,
// items 11-13, sources under tools/exp/rejected/.
#include "common.h"
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int WB_THREADS = 256;
constexpr int WB_CO = 64;   // dy channels per workgroup (4 MFMA row tiles: 2 per wave)
constexpr int WB_CI = 32;   // x channels per workgroup (2 MFMA column tiles: 1 per wave)
constexpr int WB_DS = 3;    // staging slots (8 pixels each) per thread per tile: dy
constexpr int WB_XS = 3;    // ... and x
constexpr int WB_MAXT = 9;
constexpr int WB_LDS_BUDGET = 76 * 1024;  // two workgroups per CU

struct WbArgs {
  const float* x; const float* dy; float* part;
  long part_stride;
  int N, Cin, Cout, H, W, T;
  int TR, xh, tiles_per_img, total_tiles, min_dr;
  int PBR, dpb, xpb, ksteps;   // pixel blocks per row / per dy tile / per x tile; K steps per tile
  int dslots, xslots;
  int ndc, dcs[3];             // distinct column shifts (copies)
  int x_off16;                 // first 16-byte entry of the x area
  int in_act, has_bias;
  int dbg;                     // ablation switches (PG_WB_DBG: 1 no loads, 2 no commit, 4 no MFMA), 0 in production
  int tap_base[WB_MAXT];       // entry offset of tap t inside the x area: copy plane + row-shift blocks
};

// 8 fp32 -> three bf16x8 (h, m, l pieces), element 2i in the low half of dword i.
// The pieces are TRUNCATIONS (h = top 16 bits of x, m = top 16 bits of x - h, l = x - h - m, whose
// low 16 bits are zero because 24 = 8 + 8 + 8 significand bits): h + m + l == x exactly, and a piece
// costs and + sub instead of the convert / unpack / sub of a round-to-nearest split (the staging
// arithmetic is what bounds this kernel). Against round-to-nearest pieces the dropped products
// m.l + l.m + l.l are up to 2^-23 |dy.x| instead of 2^-24 and share x.dy's sign (a relative bias of
// ~3e-8): still below one fp32 rounding of the product.
__device__ __forceinline__ void split8(const float (&x)[8], u32x4& h, u32x4& m, u32x4& l) {
  unsigned int xb[8], r1b[8], r2b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    xb[i] = __builtin_bit_cast(unsigned int, x[i]);
    const float r1 = x[i] - __builtin_bit_cast(float, xb[i] & 0xffff0000u);
    r1b[i] = __builtin_bit_cast(unsigned int, r1);
    const float r2 = r1 - __builtin_bit_cast(float, r1b[i] & 0xffff0000u);
    r2b[i] = __builtin_bit_cast(unsigned int, r2);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // v_perm_b32: {hi16(odd element), hi16(even element)}
    h[i] = __builtin_amdgcn_perm(xb[2 * i + 1], xb[2 * i], 0x07060302u);
    m[i] = __builtin_amdgcn_perm(r1b[2 * i + 1], r1b[2 * i], 0x07060302u);
    l[i] = __builtin_amdgcn_perm(r2b[2 * i + 1], r2b[2 * i], 0x07060302u);
  }
}

__device__ __forceinline__ void split1(float x, unsigned int& h, unsigned int& m, unsigned int& l) {
  const unsigned int xb = __builtin_bit_cast(unsigned int, x);
  const float r1 = x - __builtin_bit_cast(float, xb & 0xffff0000u);
  const unsigned int r1b = __builtin_bit_cast(unsigned int, r1);
  const float r2 = r1 - __builtin_bit_cast(float, r1b & 0xffff0000u);
  h = xb >> 16;
  m = r1b >> 16;
  l = __builtin_bit_cast(unsigned int, r2) >> 16;
}

// entry of the copy shifted by one pixel: prev = the pixel left of the entry (dc = -1) ...
__device__ __forceinline__ u32x4 shift_right1(const u32x4 d, unsigned int prev) {
  u32x4 r;
  r[0] = prev | (d[0] << 16);
  r[1] = __builtin_amdgcn_alignbit(d[1], d[0], 16);
  r[2] = __builtin_amdgcn_alignbit(d[2], d[1], 16);
  r[3] = __builtin_amdgcn_alignbit(d[3], d[2], 16);
  return r;
}
// ... next = the pixel right of the entry (dc = +1)
__device__ __forceinline__ u32x4 shift_left1(const u32x4 d, unsigned int next) {
  u32x4 r;
  r[0] = __builtin_amdgcn_alignbit(d[1], d[0], 16);
  r[1] = __builtin_amdgcn_alignbit(d[2], d[1], 16);
  r[2] = __builtin_amdgcn_alignbit(d[3], d[2], 16);
  r[3] = (d[3] >> 16) | (next << 16);
  return r;
}

#define MFMA16B(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16((A), (B), (C), 0, 0, 0)

// MR = MFMA row tiles (16 dy channels) per wave: 2 -> a workgroup owns 64 co (the shape the kernel was built for),
// 1 -> 32 co (round 3: the 32-channel 3x3 convolutions of PixelCNN / VD-VAE / beta-VAE, which otherwise stay on
// the fp32 kernel; half the MFMA work per staged x tile)
// WV = waves per workgroup (round 4). 4: a wave owns MR row tiles x one ci tile, two waves per SIMD. 8 (64 dy channels
// only): a wave owns ONE row tile x one ci tile — half the accumulators, two staging slots per thread instead of three,
// <= 128 registers: FOUR waves per SIMD (the measurement behind it: conv_b3p_kernel, profiles/README.md round 4 items 5-6 —
// a wave is serial, so two waves per SIMD leave the matrix pipe idle whenever both stage, wait or store).
// CIT = x channel tiles (16 channels) per workgroup (round 5). 2: the 32 x channels the kernel was built for. 4 ("big tiles",
// 8 waves, T <= 2): a wave owns MR / 2 row tiles x TWO ci tiles — 64 x 64 (MR = 2) or 128 x 64 (MR = 4) channels per
// workgroup. Why: the counters of the 1x1 128 -> 256 launch (profiles/r05_wgrad_pmc.json) show 11.4 VALU instructions per
// MFMA and the matrix pipe 21 % busy — with one tap every staged value (split into three pieces: ~6 VALU instructions and
// 3/8 of a 13-cycle ds_write_b128 each) feeds 6 MFMAs per OPPOSITE channel tile only, so a 64 x 32 tile stages
// 3072 values per 48 MFMAs; 128 x 64 stages 6144 per 192 (half the staging per MFMA, and dy / x are read 2 / 2 instead of
// 4 / 4 times for 128 -> 256). With several taps the x tile is reused per tap and the small tile is not staging bound.
template <int T, int MR = 2, int WV = 4, int CIT = 2>
__global__ void __launch_bounds__(64 * WV, WV / 2) conv_wgrad_b3_kernel(const WbArgs a) {
  static_assert(WV == 4 || (WV == 8 && (MR == 2 || MR == 4)), "8 waves: 64 or 128 dy channels per workgroup");
  static_assert(CIT == 2 || (CIT == 4 && WV == 8 && T <= 2), "big tiles: 8 waves, at most 2 taps");
  static_assert(MR != 4 || CIT == 4, "128 dy channels only with 64 x channels");
  constexpr int THREADS = 64 * WV;
  constexpr int NDS = WV == 4 ? WB_DS : 2, NXS = WV == 4 ? WB_XS : (MR == 4 ? 1 : 2);  // staging slots per thread (128 x 64: the
                                                                                          // host keeps the x tile within 512 slots)
  constexpr int MRW = WV == 4 ? MR : MR / 2;                           // row tiles per wave
  constexpr int NCIW = CIT / 2;                                        // ci tiles per wave
  constexpr int COT = 2 * MR;  // dy channel tiles per workgroup
  extern __shared__ __attribute__((aligned(16))) float lds[];
  u32x4* lds16 = reinterpret_cast<u32x4*>(lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = WV == 4 ? (wave & 1) : (wave & 3), wi = WV == 4 ? (wave >> 1) : (wave >> 2);  // row-tile group, ci tile of this wave
  const int co0 = blockIdx.y * (16 * COT), ci0 = blockIdx.z * (16 * CIT);
  const int dplane = COT * a.dpb * 16;       // entries per dy piece plane
  const int xplane = CIT * a.xpb * 16;       // entries per x (copy, piece) plane

  // ---- staging slots: the same (channel, tile row, column block) for every tile
  int d_goff[NDS], d_meta[NDS], x_goff[NXS], x_meta[NXS];  // meta: LDS entry | tile row << 20
  int x_edge = 0;  // bit k: slot k is the first column block of its row; bit 8 + k: the last
  int d_half = 0, x_half = 0;  // bit k: slot k is a half pixel block (W % 8 == 4: the last block of a row)
  const bool wpart = (a.W & 7) != 0;
#pragma unroll
  for (int k = 0; k < NDS; ++k) {
    int e = tid + k * THREADS;
    const bool in = e < a.dslots;
    e = in ? e : 0;
    const int i = e & 15; e >>= 4;
    const int cb = e % a.PBR; e /= a.PBR;
    const int tr = e % a.TR;
    const int cot = e / a.TR;
    d_goff[k] = in ? ((cot * 16 + i) * a.H + tr) * a.W + 8 * cb : -1;
    d_meta[k] = ((cot * a.dpb + tr * a.PBR + cb) * 16 + i) | (tr << 20);
    if (wpart && cb == a.PBR - 1) d_half |= 1 << k;
  }
#pragma unroll
  for (int k = 0; k < NXS; ++k) {
    int e = tid + k * THREADS;
    const bool in = e < a.xslots;
    e = in ? e : 0;
    const int i = e & 15; e >>= 4;
    const int cb = e % a.PBR; e /= a.PBR;
    const int tr = e % a.xh;
    const int cit = e / a.xh;
    x_goff[k] = in ? ((cit * 16 + i) * a.H + tr) * a.W + 8 * cb : -1;
    x_meta[k] = (a.x_off16 + (cit * a.xpb + tr * a.PBR + cb) * 16 + i) | (tr << 20);
    if (cb == 0) x_edge |= 1 << k;
    if (cb == a.PBR - 1) x_edge |= 256 << k;
    if (wpart && cb == a.PBR - 1) x_half |= 1 << k;
  }
  bool want_m1 = false, want_p1 = false;  // wave-uniform
  for (int v = 0; v < a.ndc; ++v) {
    want_m1 |= a.dcs[v] == -1;
    want_p1 |= a.dcs[v] == 1;
  }

  float4 dv[NDS][2], xv[NXS][2];
  float xe[NXS][2];  // the pixel left / right of the slot's 8
#pragma unroll
  for (int k = 0; k < NDS; ++k) dv[k][0] = dv[k][1] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < NXS; ++k) {
    xv[k][0] = xv[k][1] = make_float4(0.f, 0.f, 0.f, 0.f);
    xe[k][0] = xe[k][1] = 0.f;
  }
  int dok = 0, xok = 0;

#define PG_WB_ISSUE(TILE)                                                                          \
  {                                                                                                \
    const int n_ = (TILE) / a.tiles_per_img;                                                       \
    const int row0_ = ((TILE) - n_ * a.tiles_per_img) * a.TR;                                      \
    const float* dyb_ = a.dy + (((long)n_ * a.Cout + co0) * a.H + row0_) * (long)a.W;              \
    const float* xb_ = a.x + (((long)n_ * a.Cin + ci0) * a.H + (row0_ + a.min_dr)) * (long)a.W;    \
    dok = 0; xok = 0;                                                                              \
    _Pragma("unroll") for (int k = 0; k < NDS; ++k) {                                            \
      if (d_goff[k] >= 0 && row0_ + (d_meta[k] >> 20) < a.H) {                                     \
        int go_ = d_goff[k];  /* opaque: the 64-bit lane address is formed here, not kept across the tile loop */ \
        asm volatile("" : "+v"(go_));                                                              \
        const float4* p_ = reinterpret_cast<const float4*>(dyb_ + go_);                            \
        dv[k][0] = p_[0]; dv[k][1] = p_[((d_half >> k) & 1) ? 0 : 1];  /* a half block stays inside its row */ \
        dok |= 1 << k;                                                                             \
      }                                                                                            \
    }                                                                                              \
    _Pragma("unroll") for (int k = 0; k < NXS; ++k) {                                            \
      const int ir_ = row0_ + a.min_dr + (x_meta[k] >> 20);                                        \
      if (x_goff[k] >= 0 && ir_ >= 0 && ir_ < a.H) {                                               \
        int go_ = x_goff[k];                                                                       \
        asm volatile("" : "+v"(go_));                                                              \
        const float* q_ = xb_ + go_;                                                               \
        const float4* p_ = reinterpret_cast<const float4*>(q_);                                    \
        xv[k][0] = p_[0]; xv[k][1] = p_[((x_half >> k) & 1) ? 0 : 1];                              \
        /* edge slots load an in-image neighbour instead (zeroed at commit): no select on a   */   \
        /* loaded value here, it would put an s_waitcnt vmcnt into the issue phase             */   \
        int eg_ = x_edge;  /* opaque (as go_): the neighbour offsets are formed here */            \
        asm volatile("" : "+v"(eg_));                                                              \
        if (want_m1) xe[k][0] = q_[((eg_ >> k) & 1) ? 0 : -1];                                     \
        if (want_p1) xe[k][1] = q_[((eg_ >> (8 + k)) & 1) ? 3 : 8];                                \
        xok |= 1 << k;                                                                             \
      }                                                                                            \
    }                                                                                              \
  }

#define PG_WB_COMMIT_X(ACT)                                                                        \
  _Pragma("unroll") for (int k = 0; k < NXS; ++k) {                                              \
    if (x_goff[k] >= 0) {                                                                          \
      const bool ld_ = (xok >> k) & 1;                                                             \
      const float r_[8] = {xv[k][0].x, xv[k][0].y, xv[k][0].z, xv[k][0].w,                         \
                           xv[k][1].x, xv[k][1].y, xv[k][1].z, xv[k][1].w};                        \
      float e_[8];                                                                                 \
      const bool hb_ = (x_half >> k) & 1;                                                          \
      _Pragma("unroll") for (int c = 0; c < 8; ++c)                                                \
        e_[c] = (ld_ && !(hb_ && c >= 4)) ? pg_apply_act(r_[c], ACT) : 0.f;                        \
      u32x4 p_[3];                                                                                 \
      split8(e_, p_[0], p_[1], p_[2]);                                                             \
      const int ent_ = x_meta[k] & 0xfffff;                                                        \
      _Pragma("unroll") for (int v = 0; v < 3; ++v) {                                              \
        if (v < a.ndc) {                                                                           \
          const int dc_ = a.dcs[v];                                                                \
          u32x4* dst_ = lds16 + ent_ + v * 3 * xplane;                                             \
          if (dc_ == 0) {                                                                          \
            _Pragma("unroll") for (int q = 0; q < 3; ++q) dst_[q * xplane] = p_[q];                \
          } else {                                                                                 \
            unsigned int s_[3];                                                                    \
            const bool edge_ = (x_edge >> (dc_ < 0 ? k : 8 + k)) & 1;                              \
            split1(ld_ && !edge_ ? pg_apply_act(dc_ < 0 ? xe[k][0] : xe[k][1], ACT) : 0.f, s_[0], s_[1], s_[2]);     \
            _Pragma("unroll") for (int q = 0; q < 3; ++q)                                          \
              dst_[q * xplane] = dc_ < 0 ? shift_right1(p_[q], s_[q]) : shift_left1(p_[q], s_[q]); \
          }                                                                                        \
        }                                                                                          \
      }                                                                                            \
    }                                                                                              \
  }

  f32x4 acc[MRW][NCIW][T];
  f32x4 accb[MRW];
#pragma unroll
  for (int m = 0; m < MRW; ++m) {
    accb[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NCIW; ++j)
#pragma unroll
      for (int t = 0; t < T; ++t) acc[m][j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool bias_wave = a.has_bias && wi == 0 && blockIdx.z == 0;  // wave-uniform
  bf16x8 ones;
#pragma unroll
  for (int c = 0; c < 8; ++c) ones[c] = (__bf16)1.0f;

  const bf16x8* L = reinterpret_cast<const bf16x8*>(lds16) + lane;
  const int a_base = MRW * wc * a.dpb * 16;
  int b_base[T];
#pragma unroll
  for (int t = 0; t < T; ++t) b_base[t] = a.x_off16 + a.tap_base[t] + wi * NCIW * a.xpb * 16;

  int tile = blockIdx.x;
  if (tile < a.total_tiles && !PG_DBG_BIT(a.dbg, 1)) PG_WB_ISSUE(tile)
  for (; tile < a.total_tiles; tile += gridDim.x) {
    __syncthreads();  // the previous tile's fragment reads are done
    // every prefetch register is "used" here on every path: ONE s_waitcnt vmcnt(0) lands at this
    // point. Without it the slots whose commit is skipped leave their loads pending in the
    // compiler's model and it puts a vmcnt(0) in front of every slot of the next issue phase
    // (measured: the load phase did not overlap the MFMA loop at all).
#pragma unroll
    for (int k = 0; k < NDS; ++k)
      asm volatile("" :: "v"(dv[k][0].x), "v"(dv[k][0].y), "v"(dv[k][0].z), "v"(dv[k][0].w),
                         "v"(dv[k][1].x), "v"(dv[k][1].y), "v"(dv[k][1].z), "v"(dv[k][1].w));
#pragma unroll
    for (int k = 0; k < NXS; ++k)
      asm volatile("" :: "v"(xv[k][0].x), "v"(xv[k][0].y), "v"(xv[k][0].z), "v"(xv[k][0].w),
                         "v"(xv[k][1].x), "v"(xv[k][1].y), "v"(xv[k][1].z), "v"(xv[k][1].w),
                         "v"(xe[k][0]), "v"(xe[k][1]));
    if (!PG_DBG_BIT(a.dbg, 2)) {
#pragma unroll
    for (int k = 0; k < NDS; ++k) {
      if (d_goff[k] >= 0) {
        const bool ld = (dok >> k) & 1;
        const float r[8] = {dv[k][0].x, dv[k][0].y, dv[k][0].z, dv[k][0].w,
                            dv[k][1].x, dv[k][1].y, dv[k][1].z, dv[k][1].w};
        float e[8];
        const bool hb = (d_half >> k) & 1;
#pragma unroll
        for (int c = 0; c < 8; ++c) e[c] = (ld && !(hb && c >= 4)) ? r[c] : 0.f;
        u32x4 h, m, l;
        split8(e, h, m, l);
        u32x4* dst = lds16 + (d_meta[k] & 0xfffff);
        dst[0] = h; dst[dplane] = m; dst[2 * dplane] = l;
      }
    }
    switch (a.in_act) {  // wave-uniform
      case PG_ACT_RELU: PG_WB_COMMIT_X(PG_ACT_RELU) break;
      case PG_ACT_ELU:  PG_WB_COMMIT_X(PG_ACT_ELU) break;
      case PG_ACT_GELU: PG_WB_COMMIT_X(PG_ACT_GELU) break;
      default:          PG_WB_COMMIT_X(PG_ACT_NONE) break;
    }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < a.total_tiles && !PG_DBG_BIT(a.dbg, 1)) PG_WB_ISSUE(tile + (int)gridDim.x)
    // ---- MFMA over the tile's K steps
    if (!PG_DBG_BIT(a.dbg, 4))
    for (int ks = 0; ks < a.ksteps; ++ks) {
      const bf16x8* Lk = L + ks * 64;
      bf16x8 af[MRW][3];
#pragma unroll
      for (int m = 0; m < MRW; ++m)
#pragma unroll
        for (int p = 0; p < 3; ++p) af[m][p] = Lk[a_base + (p * COT + m) * a.dpb * 16];
      if (bias_wave) {
#pragma unroll
        for (int m = 0; m < MRW; ++m)
#pragma unroll
          for (int p = 0; p < 3; ++p) accb[m] = MFMA16B(af[m][p], ones, accb[m]);
      }
#pragma unroll
      for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int j = 0; j < NCIW; ++j) {
          bf16x8 bf[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) bf[p] = Lk[b_base[t] + j * a.xpb * 16 + p * xplane];
#pragma unroll
          for (int m = 0; m < MRW; ++m) {
            f32x4 c = acc[m][j][t];
            c = MFMA16B(af[m][2], bf[0], c);  // l.h
            c = MFMA16B(af[m][0], bf[2], c);  // h.l
            c = MFMA16B(af[m][1], bf[1], c);  // m.m
            c = MFMA16B(af[m][1], bf[0], c);  // m.h
            c = MFMA16B(af[m][0], bf[1], c);  // h.m
            c = MFMA16B(af[m][0], bf[0], c);  // h.h
            acc[m][j][t] = c;
          }
        }
      }
    }
  }
#undef PG_WB_ISSUE
#undef PG_WB_COMMIT_X

  // ---- this workgroup's row of partial sums: D[row = (lane >> 4) * 4 + r][col = lane & 15]
  float* prow = a.part + (size_t)blockIdx.x * a.part_stride;
#pragma unroll
  for (int m = 0; m < MRW; ++m) {
    const int co_b = co0 + (MRW * wc + m) * 16 + (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < NCIW; ++j) {
      const int ci = ci0 + (wi * NCIW + j) * 16 + (lane & 15);
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          prow[((size_t)(co_b + r) * a.Cin + ci) * T + t] = acc[m][j][t][r];
    }
    if (bias_wave && (lane & 15) == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) prow[(size_t)a.Cout * a.Cin * T + co_b + r] = accb[m][r];
    }
  }
}

// ---- "row ring" variant (round 6): 64 co x 64 ci per workgroup, one image row per tile, x rows kept in an LDS ring ----------
// The kernel above re-stages what it already had: with a row halo (2x2: TR + 1 x rows per TR output rows) and 32 x channels
// per workgroup it reads AND SPLITS dy once per 32-channel ci chunk and every halo row twice — 1.70x the algorithmic bytes
// on PixelSNAIL's 2x2 64 -> 64 (profiles/r06_wgrad_pmc.json) and, what matters more for a kernel bound by its staging
// arithmetic (5.5 VALU instructions per MFMA), 224 channel-rows of three-way splits per output row and 64 x channels where
// 128 suffice. Here a workgroup walks CONSECUTIVE rows of an image segment; a tile is ONE row (W == 32: exactly one K step of
// 4 pixel blocks), its dy row (64 channels) is staged once and meets all 64 x channels, and of x only the NEW row is staged —
// the rows above it are still in the ring (hr + 1 rows; a segment starts with hr x-only steps). 512 threads: waves 0-3 stage
// the dy row, waves 4-7 the x row (one 8-pixel slot per thread); every wave owns one 16-channel dy tile x two ci tiles x T taps.
// LDS: dy 12 KB + x copies * (hr + 1) * 12 KB (2x2: 60 KB, two workgroups per CU).
struct WrArgs {
  const float* x; const float* dy; float* part;
  long part_stride;
  int N, Cin, Cout, H, W, T;
  int nseg, seg_rows, units;   // row segments per image, rows per segment, N * nseg work units (strided over blockIdx.x)
  int P, RB;                   // x-only steps in front of a segment (= hr), ring rows (= hr + 1)
  int max_dr;
  int ndc, dcs[3];             // distinct column shifts (copies)
  int x_off16;                 // first 16-byte entry of the x area
  int in_act, has_bias;
  int dbg;                     // ablation switches (PG_WB_DBG: 1 no loads, 2 no commit, 4 no MFMA, 8 no fragment reads either), 0 in production
  int copy_of[6], q_of[6];     // per tap: its x copy; rows back from the newest ring row (max_dr - dr)
};

// staging slot kind of slot index k (slots e = tid + 512 k; the dy slots come first): 0 dy, 1 x, 2 none for EVERY thread, or -1 mixed
// (decided per wave at run time) — compile-time kinds keep the unrolled commit code to the branches that can occur
constexpr int wr_kind(int k, int dslots, int xslots) {
  return (k + 1) * 512 <= dslots ? 0 : (k * 512 >= dslots + xslots ? 2 : ((k * 512 >= dslots && (k + 1) * 512 <= dslots + xslots) ? 1 : -1));
}

// WM x WN = 16-channel dy tiles x ci tiles per WAVE; the 8 waves form a 4 (dy) x 2 (x) grid, so a workgroup owns 64 WM dy channels
// x 32 WN x channels: (1, 2) = 64 x 64 (two workgroups per CU: the k x k shapes with a row halo), (2, 2) = 128 x 64,
// (4, 4) = 256 x 128 (one tap: every dy / x row is staged ONCE for GatedPixelCNN's 1x1 128 -> 256; one workgroup per CU).
// D = prefetch distance in steps (register sets in flight). A step is short (one row: ~1.5 us of MFMA for the CU's two
// workgroups), the loaded-HBM latency is not.
// (A first measurement of the two-workgroup 128 x 64 two-tap tile said "no faster" and was WRONG: the weight-gradient workspace rule capped the
// grid at one workgroup per CU either way. With the rows it needs: 1x2 128->256 0.657 -> 0.552 ms, 2x1 128->256 0.620 -> 0.536.)
// two register sets and <= 128 registers (= two workgroups per CU) where that fits: the 128 x 128 one-tap tile (124) and the 128 x 64 tile with
// one / two taps (123)
template <int T, int WM = 1, int WN = 2, int D = ((WM == 2 && WN == 4) || (WM == 2 && WN == 2 && T <= 2) ? 2 : 3)>
__global__ void __launch_bounds__(512, (WM == 1 || (WM == 2 && WN == 4) || (WM == 2 && WN == 2 && T <= 2)) ? 4 : 2) conv_wgrad_b3r_kernel(const WrArgs a) {
  constexpr int COT = 4 * WM, CIT = 2 * WN;            // channel tiles per workgroup
  constexpr int DSLOTS = COT * 64, XSLOTS = CIT * 64;  // 8-pixel staging slots per row: [channel tile][4 column blocks][16 channels]
  constexpr int NS = (DSLOTS + XSLOTS + 511) / 512;    // slots per thread
  constexpr int DPLANE = COT * 64;                     // entries per dy piece plane
  constexpr int X_OFF = 3 * DPLANE;                    // first entry of the x area
  extern __shared__ __attribute__((aligned(16))) float lds[];
  u32x4* lds16 = reinterpret_cast<u32x4*>(lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave & 3, wi = wave >> 2;   // group of WM dy tiles; group of WN ci tiles
  const int co0 = blockIdx.y * (16 * COT), ci0 = blockIdx.z * (16 * CIT);
  const int xplane = CIT * a.RB * 64;        // entries per x (copy, piece) plane: [ci tile][ring row][pixel block][16 channels]
  // this thread's staging slots: 8 pixels (column block cb) of channel 16 ct + i of the row
  int kind[NS], goff[NS], ent[NS], s_cb[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int e = tid + k * 512;
    constexpr int dsl = DSLOTS, xsl = XSLOTS;
    const int kc = wr_kind(k, dsl, xsl);
    kind[k] = kc >= 0 ? kc : __builtin_amdgcn_readfirstlane(e < DSLOTS ? 0 : (e < DSLOTS + XSLOTS ? 1 : 2));  // wave-uniform (multiples of 64)
    const int el = kind[k] == 1 ? e - DSLOTS : (kind[k] == 0 ? e : 0);
    const int i = el & 15, cb = (el >> 4) & 3, ct = el >> 6;
    s_cb[k] = cb;
    goff[k] = (ct * 16 + i) * a.H * a.W + 8 * cb;
    ent[k] = kind[k] == 1 ? X_OFF + (ct * a.RB * 4 + cb) * 16 + i : (ct * 4 + cb) * 16 + i;   // x: + ring row * 64
  }
  float4 v0[D][NS], v1[D][NS];
  bool ok[D][NS];
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      v0[d][k] = v1[d][k] = make_float4(0.f, 0.f, 0.f, 0.f);
      ok[d][k] = false;
    }

  // step (unit, s): s in [-P, seg_rows): output row seg * seg_rows + s; the x row staged with it is that row + max_dr
#define PG_WR_ISSUE(K, UNIT, S)                                                                    \
  {                                                                                                \
    /* UNCONDITIONAL loads straight into set K (a load under `if (ok)` becomes load-to-temporary + select, i.e. a      */ \
    /* s_waitcnt vmcnt right behind the issue: measured, the loads then never overlap the MFMA phase); rows outside the */ \
    /* image / segment, steps past the workgroup's last and unused slots are clamped to a valid address and dropped at  */ \
    /* commit. Every wave issues the SAME 2 NS loads per set: the compiler's wait in front of a commit is then          */ \
    /* vmcnt(2 NS (D - 1)) for every wave (with different counts per path it merges them to the smaller one)            */ \
    const int u_ = PG_DBG_BIT(a.dbg, 16) ? 0 : ((UNIT) < a.units ? (UNIT) : a.units - 1);   /* 16: cache-resident loads */ \
    const int n_ = u_ / a.nseg;                                                                    \
    const int row_ = (u_ - n_ * a.nseg) * a.seg_rows + (S);                                        \
    _Pragma("unroll") for (int sk_ = 0; sk_ < NS; ++sk_) {  /* (not `k`: the caller passes its own k as K) */                                               \
      int go_ = goff[sk_];                                                                           \
      asm volatile("" : "+v"(go_));                                                                \
      const bool isx_ = kind[sk_] == 1;                                                              \
      const int ir_ = isx_ ? row_ + a.max_dr : row_;                                               \
      ok[K][sk_] = kind[sk_] == 2 ? false : (isx_ ? (ir_ >= 0 && ir_ < a.H) : (S) >= 0);               \
      const int rc_ = ir_ < 0 ? 0 : (ir_ >= a.H ? a.H - 1 : ir_);                                  \
      const float* q_ = (isx_ ? a.x + ((long)n_ * a.Cin + ci0) * a.H * (long)a.W                   \
                              : a.dy + ((long)n_ * a.Cout + co0) * a.H * (long)a.W) + (long)rc_ * a.W + go_; \
      const float4* p_ = reinterpret_cast<const float4*>(q_);                                      \
      v0[K][sk_] = p_[0]; v1[K][sk_] = p_[1];                                                          \
    }                                                                                              \
  }

#define PG_WR_COMMIT_X(K, KS, ACT, ROW)                                                            \
  {                                                                                                \
    const float r_[8] = {v0[K][KS].x, v0[K][KS].y, v0[K][KS].z, v0[K][KS].w, v1[K][KS].x, v1[K][KS].y, v1[K][KS].z, v1[K][KS].w}; \
    /* the pixel left / right of the slot's 8 = the last / first pixel of the neighbouring column block, which lane   */ \
    /* -+ 16 of this wave holds (ds_bpermute: no memory access; two more vector loads per thread and step measured     */ \
    /* slower — the launch carried ~0.9 us per step of load ISSUE cost even with cache-resident rows)                  */ \
    const float e0_ = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane - 16) & 63) << 2, __builtin_bit_cast(int, r_[7]))); \
    const float e1_ = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane + 16) & 63) << 2, __builtin_bit_cast(int, r_[0]))); \
    float x_[8];                                                                                   \
    _Pragma("unroll") for (int c = 0; c < 8; ++c) x_[c] = ok[K][KS] ? pg_apply_act(r_[c], ACT) : 0.f; \
    u32x4 p_[3];                                                                                   \
    split8(x_, p_[0], p_[1], p_[2]);                                                               \
    const int ent_ = ent[KS] + (ROW) * 64;                                                         \
    _Pragma("unroll") for (int v = 0; v < 3; ++v) {                                                \
      if (v < a.ndc) {                                                                             \
        const int dc_ = a.dcs[v];                                                                  \
        u32x4* dst_ = lds16 + ent_ + v * 3 * xplane;                                               \
        if (dc_ == 0) {                                                                            \
          _Pragma("unroll") for (int q = 0; q < 3; ++q) dst_[q * xplane] = p_[q];                  \
        } else {                                                                                   \
          unsigned int s_[3];                                                                      \
          const bool edge_ = dc_ < 0 ? s_cb[KS] == 0 : s_cb[KS] == 3;                              \
          split1(ok[K][KS] && !edge_ ? pg_apply_act(dc_ < 0 ? e0_ : e1_, ACT) : 0.f, s_[0], s_[1], s_[2]); \
          _Pragma("unroll") for (int q = 0; q < 3; ++q)                                            \
            dst_[q * xplane] = dc_ < 0 ? shift_right1(p_[q], s_[q]) : shift_left1(p_[q], s_[q]);   \
        }                                                                                          \
      }                                                                                            \
    }                                                                                              \
  }
#define PG_WR_ADVANCE(U, S) { (S) += 1; if ((S) == a.seg_rows) { (U) += (int)gridDim.x; (S) = -a.P; } }

  f32x4 acc[WM][WN][T];
  f32x4 accb[WM];
#pragma unroll
  for (int m = 0; m < WM; ++m) {
    accb[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int t = 0; t < T; ++t) acc[m][j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool bias_wave = a.has_bias && wi == 0 && blockIdx.z == 0;  // wave-uniform
  bf16x8 ones;
#pragma unroll
  for (int c = 0; c < 8; ++c) ones[c] = (__bf16)1.0f;
  const bf16x8* L = reinterpret_cast<const bf16x8*>(lds16) + lane;
  const int a_base = wc * WM * 64;
  int b_copy[T];
#pragma unroll
  for (int t = 0; t < T; ++t) b_copy[t] = X_OFF + a.copy_of[t] * 3 * xplane + wi * WN * a.RB * 64;

  int unit = blockIdx.x, s = -a.P;   // the step being committed / multiplied
  int unit_i = unit, s_i2 = s;       // the step whose loads are issued next (D steps ahead)
  int rb = 0;  // ring row the current step writes: (s + P) mod RB
#pragma unroll
  for (int k = 0; k < D; ++k) {
    if (!PG_DBG_BIT(a.dbg, 1)) PG_WR_ISSUE(k, unit_i, s_i2)
    PG_WR_ADVANCE(unit_i, s_i2)
  }
  while (unit < a.units) {
#pragma unroll
    for (int k = 0; k < D; ++k) {
      if (unit >= a.units) break;
      __syncthreads();  // the previous step's fragment reads are done
      // every register of THIS set is "used" here: one counted s_waitcnt vmcnt at this point (the younger sets stay in flight)
#pragma unroll
      for (int ks = 0; ks < NS; ++ks)
        asm volatile("" :: "v"(v0[k][ks].x), "v"(v0[k][ks].y), "v"(v0[k][ks].z), "v"(v0[k][ks].w), "v"(v1[k][ks].x), "v"(v1[k][ks].y),
                           "v"(v1[k][ks].z), "v"(v1[k][ks].w));
      if (!PG_DBG_BIT(a.dbg, 2)) {
#pragma unroll
        for (int ks = 0; ks < NS; ++ks) {
          if (kind[ks] == 0) {
            if (s >= 0) {
              const float r[8] = {v0[k][ks].x, v0[k][ks].y, v0[k][ks].z, v0[k][ks].w, v1[k][ks].x, v1[k][ks].y, v1[k][ks].z, v1[k][ks].w};
              u32x4 h, m, l;
              split8(r, h, m, l);
              u32x4* dst = lds16 + ent[ks];
              dst[0] = h; dst[DPLANE] = m; dst[2 * DPLANE] = l;
            }
          } else if (kind[ks] == 1) {
            switch (a.in_act) {  // wave-uniform
              case PG_ACT_RELU: PG_WR_COMMIT_X(k, ks, PG_ACT_RELU, rb) break;
              case PG_ACT_ELU:  PG_WR_COMMIT_X(k, ks, PG_ACT_ELU, rb) break;
              case PG_ACT_GELU: PG_WR_COMMIT_X(k, ks, PG_ACT_GELU, rb) break;
              default:          PG_WR_COMMIT_X(k, ks, PG_ACT_NONE, rb) break;
            }
          }
        }
      }
      __syncthreads();
      if (!PG_DBG_BIT(a.dbg, 1)) PG_WR_ISSUE(k, unit_i, s_i2)
      PG_WR_ADVANCE(unit_i, s_i2)
      if (s >= 0 && !PG_DBG_BIT(a.dbg, 4)) {
        bf16x8 af[WM][3];
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
          for (int p = 0; p < 3; ++p) af[m][p] = L[a_base + m * 64 + p * DPLANE];
        if (bias_wave) {
#pragma unroll
          for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int p = 0; p < 3; ++p) accb[m] = MFMA16B(af[m][p], ones, accb[m]);
        }
        // B fragments of group g = (tap, ci tile) are requested one group ahead of their MFMAs
        const bf16x8* Lb[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
          int rr = rb - a.q_of[t];
          rr = rr < 0 ? rr + a.RB : rr;
          Lb[t] = L + b_copy[t] + rr * 64;
        }
        bf16x8 bf[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) bf[0][p] = Lb[0][p * xplane];
#pragma unroll
        for (int g = 0; g < WN * T; ++g) {
          const int t = g / WN, j = g % WN;
          if (g + 1 < WN * T) {
#pragma unroll
            for (int p = 0; p < 3; ++p) bf[(g + 1) & 1][p] = Lb[(g + 1) / WN][((g + 1) % WN) * a.RB * 64 + p * xplane];
          }
#pragma unroll
          for (int m = 0; m < WM; ++m) {
            f32x4 c = acc[m][j][t];
            c = MFMA16B(af[m][2], bf[g & 1][0], c);  // l.h
            c = MFMA16B(af[m][0], bf[g & 1][2], c);  // h.l
            c = MFMA16B(af[m][1], bf[g & 1][1], c);  // m.m
            c = MFMA16B(af[m][1], bf[g & 1][0], c);  // m.h
            c = MFMA16B(af[m][0], bf[g & 1][1], c);  // h.m
            c = MFMA16B(af[m][0], bf[g & 1][0], c);  // h.h
            acc[m][j][t] = c;
          }
        }
      }
      PG_WR_ADVANCE(unit, s)
      rb = (s == -a.P) ? 0 : (rb + 1 == a.RB ? 0 : rb + 1);
    }
  }
#undef PG_WR_ISSUE
#undef PG_WR_COMMIT_X
#undef PG_WR_ADVANCE

  // ---- this workgroup's row of partial sums: D[row = (lane >> 4) * 4 + r][col = lane & 15]
  float* prow = a.part + (size_t)blockIdx.x * a.part_stride;
#pragma unroll
  for (int m = 0; m < WM; ++m) {
    const int co_b = co0 + (wc * WM + m) * 16 + (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int ci = ci0 + (wi * WN + j) * 16 + (lane & 15);
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) prow[((size_t)(co_b + r) * a.Cin + ci) * T + t] = acc[m][j][t][r];
    }
    if (bias_wave && (lane & 15) == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) prow[(size_t)a.Cout * a.Cin * T + co_b + r] = accb[m][r];
    }
  }
}

// ---- "shifted dy" variant: 32 co x 32 ci per workgroup, full NR x NC tap grids ---------------------------
// Same GEMM, the column shift moved to the OTHER operand:
//   dw[co][ci][dr, dc] = sum_{r, c'} dy[r][c' - dc] * act(x)[r + dr][c']
// so x is staged ONCE (a row shift is a whole number of pixel blocks, as before) and dy once per distinct
// column shift (copy_dc[c'] = dy[c' - dc], zero outside the row). With 32 dy channels per workgroup the dy tile
// (TR rows) is smaller than the x tile (TR + NR - 1 rows), so this halves the shift arithmetic of a 3x3 and
// cuts LDS from 3 x copies to 3 (smaller) dy copies: 64-pixel-wide images fit (the x-copy layout needs 110 KB
// there: those launches ran on the fp32 kernel), and 32-wide images get two output rows per tile.
// Rows whose width is a multiple of 4 but not of 8 (28 x 28) end in a half pixel block: its upper four K slots
// are zero in both operands.
// Per K step a wave holds the A fragments of all NC copies (NC x 3 pieces) and walks the NR row shifts with one
// set of B fragments each: NC * 3 + NR * 3 LDS reads for NR * NC * 6 MFMAs.
struct WsArgs {
  const float* x; const float* dy; float* part;
  long part_stride;
  int N, Cin, Cout, H, W, T;
  int TR, xh, tiles_per_img, total_tiles, min_dr, min_dc;
  int PBR, dpb, xpb, ksteps;
  int dslots, xslots;
  int x_off16;                 // first 16-byte entry of the x area (behind NC dy copies x 3 pieces)
  int v0;                      // dy copy with dc == 0 (bias sums)
  int wpart;                   // W % 8 != 0: the last pixel block of a row holds 4 pixels
  int in_act, has_bias;
  int tap_of[9];               // caller's tap index of grid position (ri, ci): ri * NC + ci
};

template <int NR, int NC>
__global__ void __launch_bounds__(WB_THREADS, 2) conv_wgrad_b3s_kernel(const WsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  u32x4* lds16 = reinterpret_cast<u32x4*>(lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave & 1, wi = wave >> 1;   // co tile, ci tile of this wave
  const int co0 = blockIdx.y * 32, ci0 = blockIdx.z * WB_CI;
  const int dplane = 2 * a.dpb * 16;         // entries per dy (copy, piece) plane
  const int xplane = 2 * a.xpb * 16;         // entries per x piece plane

  int d_goff[WB_DS], d_meta[WB_DS], x_goff[WB_XS], x_meta[WB_XS];  // meta: LDS entry | tile row << 20
  int d_edge = 0;  // bit k: slot k is the first column block of its row; bit 8 + k: the last
  int d_half = 0, x_half = 0;  // bit k: slot k is a half block (4 pixels)
#pragma unroll
  for (int k = 0; k < WB_DS; ++k) {
    int e = tid + k * WB_THREADS;
    const bool in = e < a.dslots;
    e = in ? e : 0;
    const int i = e & 15; e >>= 4;
    const int cb = e % a.PBR; e /= a.PBR;
    const int tr = e % a.TR;
    const int cot = e / a.TR;
    d_goff[k] = in ? ((cot * 16 + i) * a.H + tr) * a.W + 8 * cb : -1;
    d_meta[k] = ((cot * a.dpb + tr * a.PBR + cb) * 16 + i) | (tr << 20);
    if (cb == 0) d_edge |= 1 << k;
    if (cb == a.PBR - 1) {
      d_edge |= 256 << k;
      if (a.wpart) d_half |= 1 << k;
    }
  }
#pragma unroll
  for (int k = 0; k < WB_XS; ++k) {
    int e = tid + k * WB_THREADS;
    const bool in = e < a.xslots;
    e = in ? e : 0;
    const int i = e & 15; e >>= 4;
    const int cb = e % a.PBR; e /= a.PBR;
    const int tr = e % a.xh;
    const int cit = e / a.xh;
    x_goff[k] = in ? ((cit * 16 + i) * a.H + tr) * a.W + 8 * cb : -1;
    x_meta[k] = (a.x_off16 + (cit * a.xpb + tr * a.PBR + cb) * 16 + i) | (tr << 20);
    if (cb == a.PBR - 1 && a.wpart) x_half |= 1 << k;
  }
  // copy v holds dc = min_dc + v: dc = +1 needs the pixel LEFT of the slot's 8, dc = -1 the pixel right of them
  const bool want_prev = a.min_dc + NC - 1 >= 1, want_next = a.min_dc <= -1;  // wave-uniform

  float4 dv[WB_DS][2], xv[WB_XS][2];
  float de[WB_DS][2];
#pragma unroll
  for (int k = 0; k < WB_DS; ++k) {
    dv[k][0] = dv[k][1] = make_float4(0.f, 0.f, 0.f, 0.f);
    de[k][0] = de[k][1] = 0.f;
  }
#pragma unroll
  for (int k = 0; k < WB_XS; ++k) xv[k][0] = xv[k][1] = make_float4(0.f, 0.f, 0.f, 0.f);
  int dok = 0, xok = 0;

#define PG_WS_ISSUE(TILE)                                                                          \
  {                                                                                                \
    const int n_ = (TILE) / a.tiles_per_img;                                                       \
    const int row0_ = ((TILE) - n_ * a.tiles_per_img) * a.TR;                                      \
    const float* dyb_ = a.dy + (((long)n_ * a.Cout + co0) * a.H + row0_) * (long)a.W;              \
    const float* xb_ = a.x + (((long)n_ * a.Cin + ci0) * a.H + (row0_ + a.min_dr)) * (long)a.W;    \
    dok = 0; xok = 0;                                                                              \
    _Pragma("unroll") for (int k = 0; k < WB_DS; ++k) {                                            \
      if (d_goff[k] >= 0 && row0_ + (d_meta[k] >> 20) < a.H) {                                     \
        const float* q_ = dyb_ + d_goff[k];                                                        \
        const float4* p_ = reinterpret_cast<const float4*>(q_);                                    \
        const int hb_ = (d_half >> k) & 1;                                                         \
        dv[k][0] = p_[0]; dv[k][1] = p_[hb_ ? 0 : 1];  /* a half block never reads past its row */ \
        /* edge slots load an in-row pixel instead (zeroed at commit): no select on a loaded value */ \
        if (want_prev) de[k][0] = q_[((d_edge >> k) & 1) ? 0 : -1];                                \
        if (want_next) de[k][1] = q_[((d_edge >> (8 + k)) & 1) ? 3 : 8];                           \
        dok |= 1 << k;                                                                             \
      }                                                                                            \
    }                                                                                              \
    _Pragma("unroll") for (int k = 0; k < WB_XS; ++k) {                                            \
      const int ir_ = row0_ + a.min_dr + (x_meta[k] >> 20);                                        \
      if (x_goff[k] >= 0 && ir_ >= 0 && ir_ < a.H) {                                               \
        const float4* p_ = reinterpret_cast<const float4*>(xb_ + x_goff[k]);                       \
        xv[k][0] = p_[0]; xv[k][1] = p_[((x_half >> k) & 1) ? 0 : 1];                              \
        xok |= 1 << k;                                                                             \
      }                                                                                            \
    }                                                                                              \
  }

#define PG_WS_COMMIT_X(ACT)                                                                        \
  _Pragma("unroll") for (int k = 0; k < WB_XS; ++k) {                                              \
    if (x_goff[k] >= 0) {                                                                          \
      const bool ld_ = (xok >> k) & 1;                                                             \
      const bool hb_ = (x_half >> k) & 1;                                                          \
      const float r_[8] = {xv[k][0].x, xv[k][0].y, xv[k][0].z, xv[k][0].w,                         \
                           xv[k][1].x, xv[k][1].y, xv[k][1].z, xv[k][1].w};                        \
      float e_[8];                                                                                 \
      _Pragma("unroll") for (int c = 0; c < 8; ++c)                                                \
        e_[c] = (ld_ && !(hb_ && c >= 4)) ? pg_apply_act(r_[c], ACT) : 0.f;                        \
      u32x4 p_[3];                                                                                 \
      split8(e_, p_[0], p_[1], p_[2]);                                                             \
      u32x4* dst_ = lds16 + (x_meta[k] & 0xfffff);                                                 \
      _Pragma("unroll") for (int q = 0; q < 3; ++q) dst_[q * xplane] = p_[q];                      \
    }                                                                                              \
  }

  f32x4 acc[NR][NC];
  f32x4 accb = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool bias_wave = a.has_bias && wi == 0 && blockIdx.z == 0;  // wave-uniform
  bf16x8 ones;
#pragma unroll
  for (int c = 0; c < 8; ++c) ones[c] = (__bf16)1.0f;

  const bf16x8* L = reinterpret_cast<const bf16x8*>(lds16) + lane;
  const int a_base = wc * a.dpb * 16;
  const int b_base = a.x_off16 + wi * a.xpb * 16;

  int tile = blockIdx.x;
  if (tile < a.total_tiles) PG_WS_ISSUE(tile)
  for (; tile < a.total_tiles; tile += gridDim.x) {
    __syncthreads();  // the previous tile's fragment reads are done
    // every prefetch register is "used" here on every path: ONE s_waitcnt vmcnt(0) lands at this point
    // (see conv_wgrad_b3_kernel)
#pragma unroll
    for (int k = 0; k < WB_DS; ++k)
      asm volatile("" :: "v"(dv[k][0].x), "v"(dv[k][0].y), "v"(dv[k][0].z), "v"(dv[k][0].w),
                         "v"(dv[k][1].x), "v"(dv[k][1].y), "v"(dv[k][1].z), "v"(dv[k][1].w),
                         "v"(de[k][0]), "v"(de[k][1]));
#pragma unroll
    for (int k = 0; k < WB_XS; ++k)
      asm volatile("" :: "v"(xv[k][0].x), "v"(xv[k][0].y), "v"(xv[k][0].z), "v"(xv[k][0].w),
                         "v"(xv[k][1].x), "v"(xv[k][1].y), "v"(xv[k][1].z), "v"(xv[k][1].w));
#pragma unroll
    for (int k = 0; k < WB_DS; ++k) {
      if (d_goff[k] >= 0) {
        const bool ld = (dok >> k) & 1;
        const bool hb = (d_half >> k) & 1;
        const float r[8] = {dv[k][0].x, dv[k][0].y, dv[k][0].z, dv[k][0].w,
                            dv[k][1].x, dv[k][1].y, dv[k][1].z, dv[k][1].w};
        float e[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) e[c] = (ld && !(hb && c >= 4)) ? r[c] : 0.f;
        u32x4 p[3];
        split8(e, p[0], p[1], p[2]);
        u32x4* dst0 = lds16 + (d_meta[k] & 0xfffff);
#pragma unroll
        for (int v = 0; v < NC; ++v) {
          const int dc = a.min_dc + v;  // wave-uniform
          u32x4* dst = dst0 + v * 3 * dplane;
          if (dc == 0) {
#pragma unroll
            for (int q = 0; q < 3; ++q) dst[q * dplane] = p[q];
          } else {
            // copy[c'] = dy[c' - dc]: dc = +1 shifts towards higher columns (the pixel left of the block
            // enters), dc = -1 towards lower ones (the pixel right of it enters); zero at the row ends
            unsigned int s[3];
            const bool edge = (d_edge >> (dc > 0 ? k : 8 + k)) & 1;
            split1(ld && !edge ? (dc > 0 ? de[k][0] : de[k][1]) : 0.f, s[0], s[1], s[2]);
#pragma unroll
            for (int q = 0; q < 3; ++q)
              dst[q * dplane] = dc > 0 ? shift_right1(p[q], s[q]) : shift_left1(p[q], s[q]);
          }
        }
      }
    }
    switch (a.in_act) {  // wave-uniform
      case PG_ACT_RELU: PG_WS_COMMIT_X(PG_ACT_RELU) break;
      case PG_ACT_ELU:  PG_WS_COMMIT_X(PG_ACT_ELU) break;
      case PG_ACT_GELU: PG_WS_COMMIT_X(PG_ACT_GELU) break;
      default:          PG_WS_COMMIT_X(PG_ACT_NONE) break;
    }
    __syncthreads();
    if (tile + (int)gridDim.x < a.total_tiles) PG_WS_ISSUE(tile + (int)gridDim.x)
    for (int ks = 0; ks < a.ksteps; ++ks) {
      const bf16x8* Lk = L + ks * 64;
      bf16x8 af[NC][3];
#pragma unroll
      for (int v = 0; v < NC; ++v)
#pragma unroll
        for (int p = 0; p < 3; ++p) af[v][p] = Lk[a_base + (v * 3 + p) * dplane];
      if (bias_wave) {
#pragma unroll
        for (int p = 0; p < 3; ++p) accb = MFMA16B(Lk[a_base + (a.v0 * 3 + p) * dplane], ones, accb);
      }
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        bf16x8 bf[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) bf[p] = Lk[b_base + r * a.PBR * 16 + p * xplane];
#pragma unroll
        for (int v = 0; v < NC; ++v) {
          f32x4 c = acc[r][v];
          c = MFMA16B(af[v][2], bf[0], c);  // l.h
          c = MFMA16B(af[v][0], bf[2], c);  // h.l
          c = MFMA16B(af[v][1], bf[1], c);  // m.m
          c = MFMA16B(af[v][1], bf[0], c);  // m.h
          c = MFMA16B(af[v][0], bf[1], c);  // h.m
          c = MFMA16B(af[v][0], bf[0], c);  // h.h
          acc[r][v] = c;
        }
      }
    }
  }
#undef PG_WS_ISSUE
#undef PG_WS_COMMIT_X

  // ---- this workgroup's row of partial sums: D[row = (lane >> 4) * 4 + r][col = lane & 15]
  float* prow = a.part + (size_t)blockIdx.x * a.part_stride;
  const int ci = ci0 + wi * 16 + (lane & 15);
  const int co_b = co0 + wc * 16 + (lane >> 4) * 4;
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int v = 0; v < NC; ++v) {
      const int t = a.tap_of[r * NC + v];
#pragma unroll
      for (int q = 0; q < 4; ++q) prow[((size_t)(co_b + q) * a.Cin + ci) * a.T + t] = acc[r][v][q];
    }
  if (bias_wave && (lane & 15) == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) prow[(size_t)a.Cout * a.Cin * a.T + co_b + q] = accb[q];
  }
}

}  // namespace

// Launches the bf16x3 kernel when it takes the problem; returns the number of partial rows written
// (the caller runs the reduction over them), 0 when the shape stays on the fp32 kernels, < 0 on a
// launch error. Called by pg_conv2d_wgrad (conv_wgrad.hip) only.
int pg_wgrad_b3_launch(const float* x, const float* dy, float* part, long part_stride, long max_rows,
                       int has_bias, int N, int Cin, int IH, int IW, int Cout, int OH, int OW, int T,
                       const int* tap_dr, const int* tap_dc, int in_act, hipStream_t st) {
  static const bool on = []() { const char* e = PG_AB_ENV("PG_WGRAD_B3"); return !(e && e[0] == '0'); }();
  if (!on) return 0;
  if (IH != OH || IW != OW || OW % 4 != 0 || Cout % 32 != 0 || Cin % WB_CI != 0) return 0;
  int MR = Cout % WB_CO == 0 ? 2 : 1;  // 64 or 32 dy channels per workgroup
  if (MR == 1 && T == 1) return 0;  // 1x1 with 32 output channels: too little MFMA work per staged tile (measured on
                                    // PixelCNN's 64 -> 32 layers: 70.9 vs 70.6 k img/s against the fp32 direct-fragment kernel)
  if (!(T == 1 || T == 2 || T == 3 || T == 4 || T == 6 || T == 9)) return 0;
  if ((((uintptr_t)x | (uintptr_t)dy) & 15) != 0) return 0;
  WbArgs a;
  int min_dr = tap_dr[0], max_dr = tap_dr[0];
  a.ndc = 0;
  int copy_of[WB_MAXT];
  for (int t = 0; t < T; ++t) {
    if (tap_dc[t] < -1 || tap_dc[t] > 1) return 0;
    min_dr = tap_dr[t] < min_dr ? tap_dr[t] : min_dr;
    max_dr = tap_dr[t] > max_dr ? tap_dr[t] : max_dr;
    int v = 0;
    while (v < a.ndc && a.dcs[v] != tap_dc[t]) ++v;
    if (v == a.ndc) a.dcs[a.ndc++] = tap_dc[t];
    copy_of[t] = v;
  }
  for (int v = a.ndc; v < 3; ++v) a.dcs[v] = 0;
  const int hr = max_dr - min_dr;
  const int PBR = (OW + 7) / 8;  // W % 8 == 4: the last pixel block of a row is half full
  // row-ring kernel (round 6; PG_WGRAD_B3_RING=0 for A/B): 32-pixel rows; wave tiles (WM x WN 16-channel tiles) by shape:
  //   (1, 2) = 64 x 64 per workgroup, two workgroups per CU: the shapes with a row halo or three taps;
  //   (2, 2) = 128 x 64 when Cout % 128 == 0: dy AND x of a 64 -> 128 layer staged once;
  //   (4, 4) = 256 x 128 for one tap (GatedPixelCNN's 1x1 128 -> 256 / 256 -> 256), one workgroup per CU.
  // PG_WGRAD_B3_RING_CFG=<WM><WN> forces a tile in the ab library.
  static const bool ring_on = []() { const char* e = PG_AB_ENV("PG_WGRAD_B3_RING"); return !(e && e[0] == '0'); }();
  static const int ring_cfg = []() { const char* e = PG_AB_ENV("PG_WGRAD_B3_RING_CFG"); return e ? atoi(e) : 0; }();
  const bool ring_one_tap = T == 1 && Cout % 128 == 0 && Cin % 128 == 0;  // 128 x 128 tiles, two workgroups per CU: 0.357 / 0.672 ms against the
                                                                          // big-tile kernel's 0.366 / 0.702 on 1x1 128->256 / 256->256 at batch 512
  if (ring_on && OW == 32 && Cout % 64 == 0 && Cin % 64 == 0 && (T >= 2 || ring_one_tap || ring_cfg) && (T <= 4 || T == 6) && hr <= 2) {
    // measured (tools/exp/r06_ring_cfg_sweep.sh, profiles/r06_wgrad_ring_cfg_sweep.txt; ms, tiles 64x64 / 128x64 / 256x64 / old kernel):
    //   2x2 64->128 b1024 0.537 / 0.496 / - / 0.641;  2x1 256->256 b512 1.226 / 1.140 / 0.953 / 1.247;  1x3 128->256 0.896 / 0.840 / - / 1.010;
    //   1x2 128->256 0.719 / 0.652 / 0.840 / 0.684;  2x1 128->256 0.664 / 0.608 / 0.791 / 0.683.
    // One tap: 1x1 128->256 on the old 128 x 64 big-tile kernel (two workgroups per CU) 0.355 against 0.540 / 0.491 / 0.372 / 0.446 (256 x 128)
    // here — the larger tiles run ONE workgroup per CU (148-238 registers) and lose its phase overlap; only the 128 x 128 tile at two register
    // sets (124 registers, two workgroups per CU) is level with it (0.357) and reads less, so it takes the shapes it divides.
    int WM = 1, WN = 2;
    if (ring_one_tap) { WM = 2; WN = 4; }
    else if (T == 2 && Cout % 256 == 0 && Cin % 256 == 0) { WM = 4; WN = 2; }
    else if (Cout % 128 == 0 && T <= 4) { WM = 2; WN = 2; }
    if (ring_cfg) { WM = ring_cfg / 10; WN = ring_cfg % 10; }
    const bool cfg_ok = (WM == 1 && WN == 2) || (WM == 2 && WN == 2 && T <= 4) || (WM == 4 && WN == 4 && T == 1) || (WM == 4 && WN == 2 && T <= 2) ||
                        (WM == 2 && WN == 4 && T == 1);
    const bool shape_ok = cfg_ok && Cout % (64 * WM) == 0 && Cin % (32 * WN) == 0;
    WrArgs r;
    r.ndc = a.ndc;
    for (int v = 0; v < 3; ++v) r.dcs[v] = a.dcs[v];
    r.RB = hr + 1; r.P = hr; r.max_dr = max_dr;
    const int dplane = 4 * WM * 64, xplane = 2 * WN * r.RB * 64;
    const size_t shmem = ((size_t)3 * dplane + (size_t)r.ndc * 3 * xplane) * 16;
    const bool one_wg = (WM > 1 && !(WM == 2 && WN == 4) && !(WM == 2 && WN == 2 && T <= 2)) || shmem > (size_t)WB_LDS_BUDGET;   // the kernel's launch bounds / an LDS footprint above half a CU: two waves per SIMD (one 8-wave workgroup per CU)
    if (shape_ok && shmem <= (one_wg ? (size_t)150 * 1024 : (size_t)WB_LDS_BUDGET)) {
      r.x = x; r.dy = dy; r.part = part; r.part_stride = part_stride;
      r.N = N; r.Cin = Cin; r.Cout = Cout; r.H = OH; r.W = OW; r.T = T;
      r.x_off16 = 3 * dplane;
      r.in_act = in_act; r.has_bias = has_bias;
#ifdef PG_ABLATE
      { const char* e = getenv("PG_WB_DBG"); r.dbg = e ? atoi(e) : 0; }
#else
      r.dbg = 0;
#endif
      for (int t = 0; t < 6; ++t) {
        r.copy_of[t] = t < T ? copy_of[t] : 0;
        r.q_of[t] = t < T ? max_dr - tap_dr[t] : 0;
      }
      const int co_chunks = Cout / (64 * WM), ci_chunks = Cin / (32 * WN);
      long G = (one_wg ? 256 : 512) / ((long)co_chunks * ci_chunks);
      if (G < 16) G = 16;
      // work units = (image, segment of rows): whole images when there are enough of them, otherwise segments of >= 8 rows
      // (every segment starts with hr x-only steps)
      int nseg = 1;
      while ((long)N * nseg < G && OH % (2 * nseg) == 0 && OH / (2 * nseg) >= 8) nseg *= 2;
      r.nseg = nseg; r.seg_rows = OH / nseg; r.units = N * nseg;
      if (G > r.units) G = r.units;
      if (G > max_rows) G = max_rows;
      dim3 grid((unsigned)G, (unsigned)co_chunks, (unsigned)ci_chunks);
#define PG_WR_LAUNCH(TT, M_, N_)                                                                                  \
  {                                                                                                               \
    static const hipError_t attr_ = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_b3r_kernel<TT, M_, N_>), \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);  \
    if (attr_ != hipSuccess) return -1;                                                                           \
    hipLaunchKernelGGL((conv_wgrad_b3r_kernel<TT, M_, N_>), grid, dim3(512), shmem, st, r);                       \
  }
#define PG_WR_BY_T(M_, N_)                                         \
  switch (T) {                                                     \
    case 1: PG_WR_LAUNCH(1, M_, N_) break;                         \
    case 2: PG_WR_LAUNCH(2, M_, N_) break;                         \
    case 3: PG_WR_LAUNCH(3, M_, N_) break;                         \
    default: PG_WR_LAUNCH(4, M_, N_) break;                        \
  }
      if (WM == 4 && WN == 4) PG_WR_LAUNCH(1, 4, 4)
      else if (WM == 2 && WN == 4) PG_WR_LAUNCH(1, 2, 4)
      else if (WM == 4) { if (T == 1) PG_WR_LAUNCH(1, 4, 2) else PG_WR_LAUNCH(2, 4, 2) }
      else if (WM == 2) PG_WR_BY_T(2, 2)
      else if (T == 6) PG_WR_LAUNCH(6, 1, 2)
      else PG_WR_BY_T(1, 2)
#undef PG_WR_BY_T
#undef PG_WR_LAUNCH
      if (hipGetLastError() != hipSuccess) return -1;
      return (int)G;
    }
  }
  // 64 dy channels per workgroup: 8 waves (four per SIMD, 2 staging slots per thread); PG_WGRAD_B3_WAVES=4 for A/B
  static const int env_waves = []() { const char* e = PG_AB_ENV("PG_WGRAD_B3_WAVES"); return (e && atoi(e) == 4) ? 4 : 8; }();
  // tile rows: K steps of 4 pixel blocks, staging slots within the per-thread caps, LDS within budget
  auto pick_rows = [&](int co_, int ci_, long slot_cap_, long xslot_cap_ = 0) {
    if (xslot_cap_ == 0) xslot_cap_ = slot_cap_;
    int best = 0;
    for (int tr = 1; tr <= OH + 3; ++tr) {
      if ((tr * PBR) % 4 != 0) continue;
      const long dslots = (long)co_ * tr * PBR, xslots = (long)ci_ * (tr + hr) * PBR;
      const long bytes = (3 * dslots + 3L * a.ndc * xslots) * 16;
      if (dslots > slot_cap_ || xslots > xslot_cap_ || bytes > WB_LDS_BUDGET) break;
      best = tr;
      if (tr >= OH) break;
    }
    return best;
  };
  // big tiles (round 5; at most 2 taps, 8 waves): 64 x channels per workgroup, and 128 dy channels for one tap — the
  // 1x1 / 2x1 / 1x2 weight gradients are bound by staging (split + LDS writes per MFMA: profiles/r05_wgrad_pmc.json),
  // not by the matrix pipe. PG_WGRAD_B3_BIG=0 for A/B. A shape whose big tile does not fit LDS keeps the 64 x 32 tile.
  static const bool big_on = []() { const char* e = PG_AB_ENV("PG_WGRAD_B3_BIG"); return !(e && e[0] == '0'); }();
  bool big = big_on && MR == 2 && T <= 2 && Cin % 64 == 0;
  int TR = 0;
  if (big && T == 1 && Cout % 128 == 0 && (TR = pick_rows(128, 64, 2L * 512, 512)) > 0) MR = 4;
  if (big && TR == 0 && (TR = pick_rows(64, 64, 2L * 512)) == 0) big = false;
  const int CIT = big ? 4 : 2, wb_ci = 16 * CIT;
  const int wb_co = 32 * MR;
  const int waves = big ? 8 : (MR == 2 && T <= 4) ? env_waves : 4;  // (6 and 9 taps: the accumulators no longer fit 128 registers)
  if (!big) TR = pick_rows(wb_co, wb_ci, waves == 8 ? 2L * 512 : (long)WB_DS * WB_THREADS);
  if (TR == 0) return 0;
  a.x = x; a.dy = dy; a.part = part; a.part_stride = part_stride;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = OH; a.W = OW; a.T = T;
  a.TR = TR; a.xh = TR + hr; a.min_dr = min_dr;
  a.tiles_per_img = (OH + TR - 1) / TR;
  a.total_tiles = N * a.tiles_per_img;
  a.PBR = PBR; a.dpb = TR * PBR; a.xpb = a.xh * PBR; a.ksteps = a.dpb / 4;
  a.dslots = wb_co * a.dpb; a.xslots = wb_ci * a.xpb;
  a.x_off16 = 3 * (2 * MR) * a.dpb * 16;
  a.in_act = in_act; a.has_bias = has_bias;
#ifdef PG_ABLATE
  static const int dbg = []() { const char* e = getenv("PG_WB_DBG"); return e ? atoi(e) : 0; }();
  a.dbg = dbg;
#else
  a.dbg = 0;
#endif
  const int xplane = CIT * a.xpb * 16;
  for (int t = 0; t < T; ++t) a.tap_base[t] = copy_of[t] * 3 * xplane + (tap_dr[t] - min_dr) * PBR * 16;
  for (int t = T; t < WB_MAXT; ++t) a.tap_base[t] = 0;
  const size_t shmem = ((size_t)a.x_off16 + (size_t)a.ndc * 3 * xplane) * 16;
  const int co_chunks = Cout / wb_co, ci_chunks = Cin / wb_ci;
  long G = 512 / ((long)co_chunks * ci_chunks);
  if (G < 16) G = 16;
  if (G > a.total_tiles) G = a.total_tiles;
  if (G > max_rows) G = max_rows;
  dim3 grid((unsigned)G, (unsigned)co_chunks, (unsigned)ci_chunks);
  if (big) {
    if (T == 1 && MR == 4) hipLaunchKernelGGL((conv_wgrad_b3_kernel<1, 4, 8, 4>), grid, dim3(512), shmem, st, a);
    else if (T == 1) hipLaunchKernelGGL((conv_wgrad_b3_kernel<1, 2, 8, 4>), grid, dim3(512), shmem, st, a);
    else hipLaunchKernelGGL((conv_wgrad_b3_kernel<2, 2, 8, 4>), grid, dim3(512), shmem, st, a);
    if (hipGetLastError() != hipSuccess) return -1;
    return (int)G;
  }
#define PG_WB(TT)                                                                                        \
  {                                                                                                      \
    if (MR == 2) hipLaunchKernelGGL((conv_wgrad_b3_kernel<TT, 2>), grid, dim3(WB_THREADS), shmem, st, a); \
    else hipLaunchKernelGGL((conv_wgrad_b3_kernel<TT, 1>), grid, dim3(WB_THREADS), shmem, st, a);         \
  }
#define PG_WB8(TT)                                                                                       \
  {                                                                                                      \
    if (waves == 8) hipLaunchKernelGGL((conv_wgrad_b3_kernel<TT, 2, 8>), grid, dim3(512), shmem, st, a);  \
    else PG_WB(TT)                                                                                       \
  }
  switch (T) {
    case 1: PG_WB8(1); break;
    case 2: PG_WB8(2); break;
    case 3: PG_WB8(3); break;
    case 4: PG_WB8(4); break;
    case 6: PG_WB(6); break;
    default: PG_WB(9); break;
  }
#undef PG_WB8
#undef PG_WB
  if (hipGetLastError() != hipSuccess) return -1;
  return (int)G;
}

// The "shifted dy" kernel (32 co x 32 ci per workgroup, full tap grids): same return convention.
int pg_wgrad_b3s_launch(const float* x, const float* dy, float* part, long part_stride, long max_rows,
                        int has_bias, int N, int Cin, int IH, int IW, int Cout, int OH, int OW, int T,
                        const int* tap_dr, const int* tap_dc, int in_act, hipStream_t st) {
  static const bool on = []() {
    const char* e = PG_AB_ENV("PG_WGRAD_B3");
    const char* s = PG_AB_ENV("PG_WGRAD_B3S");
    return !(e && e[0] == '0') && !(s && s[0] == '0');
  }();
  if (!on) return 0;
  if (IH != OH || IW != OW || OW % 4 != 0 || Cout % 32 != 0 || Cin % WB_CI != 0 || T < 2 || T > 9) return 0;
  if ((((uintptr_t)x | (uintptr_t)dy) & 15) != 0) return 0;
  int min_dr = tap_dr[0], max_dr = tap_dr[0], min_dc = tap_dc[0], max_dc = tap_dc[0];
  for (int t = 1; t < T; ++t) {
    min_dr = tap_dr[t] < min_dr ? tap_dr[t] : min_dr;
    max_dr = tap_dr[t] > max_dr ? tap_dr[t] : max_dr;
    min_dc = tap_dc[t] < min_dc ? tap_dc[t] : min_dc;
    max_dc = tap_dc[t] > max_dc ? tap_dc[t] : max_dc;
  }
  const int NR = max_dr - min_dr + 1, NC = max_dc - min_dc + 1;
  if (min_dc < -1 || max_dc > 1 || NR > 3 || NR * NC != T) return 0;
  if (has_bias && (min_dc > 0 || max_dc < 0)) return 0;  // the bias sums read the unshifted copy
  WsArgs a;
  for (int i = 0; i < 9; ++i) a.tap_of[i] = -1;
  for (int t = 0; t < T; ++t) {
    const int slot = (tap_dr[t] - min_dr) * NC + (tap_dc[t] - min_dc);
    if (a.tap_of[slot] >= 0) return 0;  // a repeated tap: not a full grid
    a.tap_of[slot] = t;
  }
  const int hr = NR - 1;
  const int PBR = (OW + 7) / 8;
  int TR = 0;
  for (int tr = 1; tr <= OH + 3; ++tr) {
    if ((tr * PBR) % 4 != 0) continue;
    const long dslots = 32L * tr * PBR, xslots = (long)WB_CI * (tr + hr) * PBR;
    const long bytes = (3L * NC * dslots + 3 * xslots) * 16;
    if (dslots > WB_DS * WB_THREADS || xslots > WB_XS * WB_THREADS || bytes > WB_LDS_BUDGET) break;
    TR = tr;
    if (tr >= OH) break;
  }
  if (TR == 0) return 0;
  a.x = x; a.dy = dy; a.part = part; a.part_stride = part_stride;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = OH; a.W = OW; a.T = T;
  a.TR = TR; a.xh = TR + hr; a.min_dr = min_dr; a.min_dc = min_dc;
  a.tiles_per_img = (OH + TR - 1) / TR;
  a.total_tiles = N * a.tiles_per_img;
  a.PBR = PBR; a.dpb = TR * PBR; a.xpb = a.xh * PBR; a.ksteps = a.dpb / 4;
  a.dslots = 32 * a.dpb; a.xslots = WB_CI * a.xpb;
  a.x_off16 = NC * 3 * (2 * a.dpb * 16);
  a.v0 = -min_dc;
  a.wpart = (OW % 8) != 0;
  a.in_act = in_act; a.has_bias = has_bias;
  const size_t shmem = ((size_t)a.x_off16 + (size_t)3 * (2 * a.xpb * 16)) * 16;
  const int co_chunks = Cout / 32, ci_chunks = Cin / WB_CI;
  long G = 512 / ((long)co_chunks * ci_chunks);
  if (G < 16) G = 16;
  if (G > a.total_tiles) G = a.total_tiles;
  if (G > max_rows) G = max_rows;
  dim3 grid((unsigned)G, (unsigned)co_chunks, (unsigned)ci_chunks);
#define PG_WS(R, C) hipLaunchKernelGGL((conv_wgrad_b3s_kernel<R, C>), grid, dim3(WB_THREADS), shmem, st, a)
  if (NR == 3 && NC == 3) PG_WS(3, 3);
  else if (NR == 2 && NC == 2) PG_WS(2, 2);
  else if (NR == 1 && NC == 3) PG_WS(1, 3);
  else if (NR == 2 && NC == 1) PG_WS(2, 1);
  else if (NR == 2 && NC == 3) PG_WS(2, 3);
  else return 0;
#undef PG_WS
  if (hipGetLastError() != hipSuccess) return -1;
  return (int)G;
}

// conv_wgrad.hip — convolution weight/bias gradient on the fp32 matrix cores (gfx950).
//
// Replaces the weight/bias outputs of aten::convolution_backward that autograd reaches from
// loss.backward() (reference trainer.py:180) for every conv on the path. NOTE: the reference
// masks `weight.data` outside autograd (nn/convolution.py:42), so its weight.grad is the FULL
// unmasked correlation — callers pass all KH*KW taps for CausalConv2d to keep grad-norm /
// Adam-moment parity.
//
//   dw[co][ci][u_t][v_t] += sum_{n,r,c} dy[n,co,r,c] * act(x[n,ci,r+dr_t,c+dc_t])
//   db[co]               += sum_{n,r,c} dy[n,co,r,c]
//
// This is a GEMM whose contraction axis is the pixel axis (N*OH*OW ~ 1e5..1e6) and whose
// output is tiny (Cout x Cin x taps) — the one place on this path where MFMA is the natural
// fit: v_mfma_f32_16x16x4_f32 is exact fp32 (bitwise an fmaf chain) at the full fp32 rate.
//   A[i=lane&15][k=lane>>4] = dy[co_tile*16+i][pos+k]
//   B[k=lane>>4][j=lane&15] = x [ci_tile*16+j][pos+k+tapoff]
//   D[(lane>>4)*4+r][lane&15] accumulates dw[co][ci] for one tap.
// dy / x tiles (x with halo, zero filled) are staged in LDS with channel strides == 2 (mod 32)
// so both fragment reads are bank-conflict free. Each workgroup walks many pixel tiles and
// flushes once with fp32 atomics (caller zeroes dw/db once per step).
#include "common.h"
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WG_THREADS = 256;
constexpr int CO_CHUNK = 64;  // dy channels staged per block
constexpr int CI_CHUNK = 32;  // x channels staged per block (2 MFMA column tiles)
constexpr int DS = 4;    // float4 staging slots per thread per tile: dy
constexpr int XSW = 6;   // ... and x
constexpr int LDS_BUDGET_FLOATS = 18000;  // ~72 KB -> two workgroups per CU: one stages while the other runs MFMAs (measured best of 36k/18k/9.5k)

struct WgArgs {
  const float* x; const float* dy; float* dw; float* db;
  float* part;        // [gridDim.x][part_stride] per-block partial sums (no atomics: see below)
  long part_stride;   // = Cout*Cin*T + Cout
  int N, Cin, IH, IW, Cout, OH, OW, KH, KW, T;
  int TR, tiles_per_img, total_tiles, SW, xh, min_dr, min_dc;
  int S_dy, S_x, npos, in_act;
  int swp_shift, rpi;       // lanes per staged row = 1<<swp_shift, rows per wave iteration
  int dump;                 // LDS float index of the dump word (past every tile / scratch area)
  int vec_dy, vec_x, qshift_dy, qshift_x;  // float4 staging when rows are 16-byte aligned
  float inv_TR, inv_xh;
  int prefetch;             // float4 slot staging with register prefetch (both tensors 16-byte aligned rows)
  int Qd, Qx, dslots, xslots;  // quads per row and slots per tile (dy / x)
  int tapoff[PG_MAX_TAPS];
  int tap_u[PG_MAX_TAPS];
  int tap_v[PG_MAX_TAPS];
};

template <int NT, int NCIT, bool PF>
__global__ void __launch_bounds__(WG_THREADS, NT <= 4 ? 2 : 1) conv_wgrad_kernel(const WgArgs a) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int co0 = blockIdx.y * CO_CHUNK;
  const int ci0 = blockIdx.z * CI_CHUNK;
  const int nco = min(CO_CHUNK, a.Cout - co0);
  const int nci = min(CI_CHUNK, a.Cin - ci0);
  const int ncot_real = (nco + 15) >> 4;
  const int ncot = ncot_real >= 3 ? 4 : ncot_real;  // 1, 2 or 4 co tiles across the 4 waves
  const int ks = 4 / ncot;                          // K-split factor
  const int cot = wave % ncot;
  const int kpart = wave / ncot;
  const int ncit = (nci + 15) >> 4;

  float* dyl = lds;                                // [nco16][S_dy]
  float* xl = lds + (size_t)ncot * 16 * a.S_dy;    // [ncit*16][S_x]
  const int lds_floats = ncot * 16 * a.S_dy + ncit * 16 * a.S_x;
  for (int i = tid; i < lds_floats; i += WG_THREADS) lds[i] = 0.f;

  const int a_base = (cot * 16 + (lane & 15)) * a.S_dy + (lane >> 4);
  const int b_base = (lane & 15) * a.S_x + (lane >> 4);
  const bool do_bias = (a.db != nullptr) && (blockIdx.z == 0);

  // NT taps are accumulated per pass over the tiles; a last partial pass re-uses tap 0's offset
  // for its surplus slots (their results are simply not flushed).
  for (int t0 = 0; t0 < a.T; t0 += NT) {
    const int tcount = min(NT, a.T - t0);
    f32x4 acc[NCIT][NT];
    f32x4 accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCIT; ++c)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    if constexpr (PF) {
      // ---- slot staging: this thread's float4 share of a (dy, x) tile pair is the same set of
      // (channel, tile row, quad) elements for every tile; the loads of tile i+1 are issued before the
      // MFMA loop of tile i and committed to LDS after it (global latency under the matrix pipe)
      int d_goff[DS], d_meta[DS], x_goff[XSW], x_meta[XSW];  // meta: LDS offset | mask << 16 | row << 20
#pragma unroll
      for (int k = 0; k < DS; ++k) {
        int e = tid + k * WG_THREADS;
        const bool in = e < a.dslots;
        e = in ? e : 0;
        const int q = e % a.Qd; e /= a.Qd;
        const int tr = e % a.TR;
        const int ch = e / a.TR;
        d_goff[k] = in && ch < nco ? (ch * a.OH + tr) * a.OW + 4 * q : -1;
        d_meta[k] = (ch * a.S_dy + tr * a.SW + 4 * q) | (0xf << 16) | (tr << 20);
      }
#pragma unroll
      for (int k = 0; k < XSW; ++k) {
        int e = tid + k * WG_THREADS;
        const bool in = e < a.xslots;
        e = in ? e : 0;
        const int q = e % a.Qx; e /= a.Qx;
        const int tr = e % a.xh;
        const int ch = e / a.xh;
        const int tcol = 4 * q - a.min_dc;
        int mask = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (tcol + i >= 0 && tcol + i < a.SW) mask |= 1 << i;
        x_goff[k] = in && ch < nci ? (ch * a.IH + tr) * a.IW + 4 * q : -1;
        x_meta[k] = (ch * a.S_x + tr * a.SW + tcol + 4) | (mask << 16) | (tr << 20);
      }
      float4 dv[DS], xv[XSW];
#pragma unroll
      for (int k = 0; k < DS; ++k) dv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < XSW; ++k) xv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      int dok = 0, xok = 0;
#define PG_WG_ISSUE(TILE)                                                                     \
  {                                                                                           \
    const int n_ = (TILE) / a.tiles_per_img;                                                  \
    const int row0_ = ((TILE) - n_ * a.tiles_per_img) * a.TR;                                 \
    const float* dyb_ = a.dy + (((size_t)n_ * a.Cout + co0) * a.OH + row0_) * a.OW;           \
    const float* xb_ = a.x + (((size_t)n_ * a.Cin + ci0) * a.IH + (row0_ + a.min_dr)) * a.IW; \
    dok = 0; xok = 0;                                                                         \
    _Pragma("unroll") for (int k = 0; k < DS; ++k) {                                          \
      const int tr = d_meta[k] >> 20;                                                         \
      if (d_goff[k] >= 0 && row0_ + tr < a.OH) {                                              \
        dv[k] = *reinterpret_cast<const float4*>(dyb_ + d_goff[k]);                           \
        dok |= 1 << k;                                                                        \
      }                                                                                       \
    }                                                                                         \
    _Pragma("unroll") for (int k = 0; k < XSW; ++k) {                                         \
      const int ir = row0_ + a.min_dr + (x_meta[k] >> 20);                                    \
      if (x_goff[k] >= 0 && ir >= 0 && ir < a.IH) {                                           \
        xv[k] = *reinterpret_cast<const float4*>(xb_ + x_goff[k]);                            \
        xok |= 1 << k;                                                                        \
      }                                                                                       \
    }                                                                                         \
  }
#define PG_WG_COMMIT_X(ACT)                                                      \
  _Pragma("unroll") for (int k = 0; k < XSW; ++k) {                              \
    const bool ld = (xok >> k) & 1;                                              \
    const int loff = (x_meta[k] & 0xffff) - 4;                                   \
    const float e[4] = {xv[k].x, xv[k].y, xv[k].z, xv[k].w};                     \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                              \
      const bool wr = x_goff[k] >= 0 && ((x_meta[k] >> (16 + i)) & 1);           \
      xl[wr ? loff + i : xdump] = ld ? pg_apply_act(e[i], ACT) : 0.f;            \
    }                                                                            \
  }
      const int xdump = a.dump - (int)(xl - lds);
      int tile = blockIdx.x;
      if (tile < a.total_tiles) PG_WG_ISSUE(tile)
      for (; tile < a.total_tiles; tile += gridDim.x) {
        __syncthreads();  // the previous tile's fragment reads are done
#pragma unroll
        for (int k = 0; k < DS; ++k) {  // rows past the image (last tile) are written as zeros
          if (d_goff[k] >= 0) {  // (channel strides are == 2 mod 32: dword stores)
            const bool ld = (dok >> k) & 1;
            float* d = dyl + (d_meta[k] & 0xffff);
            d[0] = ld ? dv[k].x : 0.f; d[1] = ld ? dv[k].y : 0.f;
            d[2] = ld ? dv[k].z : 0.f; d[3] = ld ? dv[k].w : 0.f;
          }
        }
        switch (a.in_act) {  // wave-uniform
          case PG_ACT_RELU: PG_WG_COMMIT_X(PG_ACT_RELU) break;
          case PG_ACT_ELU:  PG_WG_COMMIT_X(PG_ACT_ELU) break;
          case PG_ACT_GELU: PG_WG_COMMIT_X(PG_ACT_GELU) break;
          default:          PG_WG_COMMIT_X(PG_ACT_NONE) break;
        }
        __syncthreads();
        if (tile + (int)gridDim.x < a.total_tiles) PG_WG_ISSUE(tile + (int)gridDim.x)
      // ---- MFMA over the tile's positions
        if (cot < ncot_real) {
          // tap offsets hoisted out of the K loop (a kernarg s_load per tap per step shares
          // lgkmcnt with the ds_reads), and the step body is branch-free: 1 + NT*NCIT ds_reads are
          // issued back to back, then the MFMAs.
          int toff[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) toff[t] = a.tapoff[t0 + (t < tcount ? t : 0)];
          const bool bias_pass = do_bias && t0 == 0;
#pragma unroll 2
          for (int p0 = kpart * 4; p0 < a.npos; p0 += 4 * ks) {
            const float av = dyl[a_base + p0];
            float bv[NCIT][NT];
#pragma unroll
            for (int c = 0; c < NCIT; ++c)
#pragma unroll
              for (int t = 0; t < NT; ++t) bv[c][t] = xl[b_base + c * 16 * a.S_x + toff[t] + p0];
            // keep the 1 + NT*NCIT ds_reads together in front of the MFMAs: left alone, the scheduler
            // interleaves them one by one (ds_read, s_waitcnt lgkmcnt(0), v_mfma, ...) to save registers
            // and every MFMA eats a full LDS latency
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < NCIT; ++c)
#pragma unroll
              for (int t = 0; t < NT; ++t)
                acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[c][t], acc[c][t], 0, 0, 0);
            if (bias_pass) accb = __builtin_amdgcn_mfma_f32_16x16x4f32(av, 1.0f, accb, 0, 0, 0);
          }
        }
      }
#undef PG_WG_ISSUE
#undef PG_WG_COMMIT_X
    } else
    for (int tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
      const int n = tile / a.tiles_per_img;
      const int row0 = (tile - n * a.tiles_per_img) * a.TR;
      __syncthreads();
      // ---- stage the dy rows and the x rows (+halo, zero filled, prologue activation applied once)
      if (a.vec_dy)
        pg_stage_rows_vec4<PG_ACT_NONE>(dyl, a.dump - (int)(dyl - lds), a.S_dy, a.TR, a.SW, a.inv_TR,
                                        a.dy + ((size_t)n * a.Cout + co0) * a.OH * a.OW,
                                        (size_t)a.OH * a.OW, a.OH, a.OW, nco, row0, 0, a.qshift_dy,
                                        wave, 4, lane);
      else
        pg_stage_rows<PG_ACT_NONE>(dyl, a.dump - (int)(dyl - lds), a.S_dy, a.TR, a.SW, a.inv_TR,
                                   a.dy + ((size_t)n * a.Cout + co0) * a.OH * a.OW,
                                   (size_t)a.OH * a.OW, a.OH, a.OW, nco, row0, 0, a.swp_shift, a.rpi,
                                   wave, 4, lane);
#define PG_STAGE_X(ACT)                                                                            \
  if (a.vec_x)                                                                                      \
    pg_stage_rows_vec4<ACT>(xl, a.dump - (int)(xl - lds), a.S_x, a.xh, a.SW, a.inv_xh,              \
                            a.x + ((size_t)n * a.Cin + ci0) * a.IH * a.IW, (size_t)a.IH * a.IW,     \
                            a.IH, a.IW, nci, row0 + a.min_dr, a.min_dc, a.qshift_x, wave, 4, lane);  \
  else                                                                                              \
    pg_stage_rows<ACT>(xl, a.dump - (int)(xl - lds), a.S_x, a.xh, a.SW, a.inv_xh,                   \
                       a.x + ((size_t)n * a.Cin + ci0) * a.IH * a.IW, (size_t)a.IH * a.IW, a.IH,    \
                       a.IW, nci, row0 + a.min_dr, a.min_dc, a.swp_shift, a.rpi, wave, 4, lane)
      switch (a.in_act) {  // wave-uniform
        case PG_ACT_RELU: PG_STAGE_X(PG_ACT_RELU); break;
        case PG_ACT_ELU:  PG_STAGE_X(PG_ACT_ELU); break;
        case PG_ACT_GELU: PG_STAGE_X(PG_ACT_GELU); break;
        default:          PG_STAGE_X(PG_ACT_NONE); break;
      }
#undef PG_STAGE_X
      __syncthreads();
      // ---- MFMA over the tile's positions
      if (cot < ncot_real) {
        // tap offsets hoisted out of the K loop (a kernarg s_load per tap per step shares
        // lgkmcnt with the ds_reads), and the step body is branch-free: 1 + NT*NCIT ds_reads are
        // issued back to back, then the MFMAs.
        int toff[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) toff[t] = a.tapoff[t0 + (t < tcount ? t : 0)];
        const bool bias_pass = do_bias && t0 == 0;
#pragma unroll 2
        for (int p0 = kpart * 4; p0 < a.npos; p0 += 4 * ks) {
          const float av = dyl[a_base + p0];
          float bv[NCIT][NT];
#pragma unroll
          for (int c = 0; c < NCIT; ++c)
#pragma unroll
            for (int t = 0; t < NT; ++t) bv[c][t] = xl[b_base + c * 16 * a.S_x + toff[t] + p0];
          // keep the 1 + NT*NCIT ds_reads together in front of the MFMAs: left alone, the scheduler
          // interleaves them one by one (ds_read, s_waitcnt lgkmcnt(0), v_mfma, ...) to save registers
          // and every MFMA eats a full LDS latency
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int c = 0; c < NCIT; ++c)
#pragma unroll
            for (int t = 0; t < NT; ++t)
              acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[c][t], acc[c][t], 0, 0, 0);
          if (bias_pass) accb = __builtin_amdgcn_mfma_f32_16x16x4f32(av, 1.0f, accb, 0, 0, 0);
        }
      }
    }
    // ---- flush this tap chunk: K-split partners reduce through LDS, one set of atomics per
    //      (co tile) instead of one per wave
    __syncthreads();
    if (ks > 1 && kpart > 0 && cot < ncot_real) {
      float* dst = lds + ((size_t)((kpart - 1) * ncot + cot) * (NCIT * NT + 1)) * 256 + lane * 4;
#pragma unroll
      for (int c = 0; c < NCIT; ++c)
#pragma unroll
        for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4*>(dst + (c * NT + t) * 256) = acc[c][t];
      *reinterpret_cast<f32x4*>(dst + NCIT * NT * 256) = accb;
    }
    __syncthreads();
    if (kpart == 0 && cot < ncot_real) {
      for (int kp = 1; kp < ks; ++kp) {
        const float* src = lds + ((size_t)((kp - 1) * ncot + cot) * (NCIT * NT + 1)) * 256 + lane * 4;
#pragma unroll
        for (int c = 0; c < NCIT; ++c)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[c][t] += *reinterpret_cast<const f32x4*>(src + (c * NT + t) * 256);
        accb += *reinterpret_cast<const f32x4*>(src + NCIT * NT * 256);
      }
      const int co_b = co0 + cot * 16 + (lane >> 4) * 4;
      float* prow = a.part + (size_t)blockIdx.x * a.part_stride;
#pragma unroll
      for (int c = 0; c < NCIT; ++c) {
        if (c < ncit) {
          const int ci = ci0 + c * 16 + (lane & 15);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            if (t < tcount && ci < a.Cin) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int co = co_b + r;
                if (co < a.Cout)
                  prow[((size_t)co * a.Cin + ci) * a.T + (t0 + t)] = acc[c][t][r];
              }
            }
          }
        }
      }
      if (do_bias && t0 == 0 && (lane & 15) == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co_b + r;
          if (co < a.Cout) prow[(size_t)a.Cout * a.Cin * a.T + co] = accb[r];
        }
      }
    }
    __syncthreads();
    // the reduction scratch overlapped the staging tiles: restore the never-rewritten zero margins
    if (t0 + NT < a.T)
      for (int i = tid; i < lds_floats; i += WG_THREADS) lds[i] = 0.f;
  }
}

// ---- 1x1 fast path: fragments straight from global memory, no LDS --------------------------
// dw[co][ci] += sum_p dy[co][p] * act(x[ci][p]). Each lane loads ONE float4 (4 consecutive pixels
// of "its" channel lane&15, pixel quad lane>>4) per 16-channel tile; MFMA #j of a 16-pixel group
// contracts element j of every lane's float4, i.e. pixels {4k + j : k = lane>>4} — A and B use
// the same pixel<->(lane, j) bijection, so no cross-lane movement is needed.
struct PwWgArgs {
  const float* x; const float* dy; float* dw; float* db;
  float* part;        // [gridDim.x][Cout*Cin + Cout]
  long part_stride;
  int N, Cin, Cout, L, G16, in_act;
  long total_groups;
};

template <int NCOT, int NCIT, int ACT>
__global__ void __launch_bounds__(WG_THREADS) conv_wgrad_pw_kernel(const PwWgArgs a) {
  // U pixel groups are loaded per iteration before any MFMA, so >= 2*U*(tiles) 16-byte loads are
  // in flight per lane (the loop is HBM-latency bound otherwise).
  constexpr int U = (NCOT + NCIT <= 3) ? 4 : 2;
  constexpr int NT = NCOT * NCIT + NCOT;  // accumulator tiles per wave (dw tiles + bias tiles)
  extern __shared__ float red[];           // [3 waves][NT][64 lanes][4]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;
  const int ch = lane & 15, quad = lane >> 4;
  const bool do_bias = a.db != nullptr && blockIdx.z == 0;

  f32x4 acc[NCOT][NCIT];
  f32x4 accb[NCOT];
#pragma unroll
  for (int i = 0; i < NCOT; ++i) {
    accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NCIT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const long wstride = (long)gridDim.x * (WG_THREADS / 64);
  // each wave takes U CONSECUTIVE 16-pixel groups per iteration: its U loads per channel cover one
  // contiguous U*64 B span (whole cache lines), issued back to back
  for (long g0 = ((long)blockIdx.x * (WG_THREADS / 64) + wave) * U; g0 < a.total_groups; g0 += U * wstride) {
    float4 av[U][NCOT], bv[U][NCIT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long grp = g0 + u;
      const bool gok = grp < a.total_groups;
      const long gc = gok ? grp : 0;
      const int n = (int)(gc / a.G16);
      const int p0 = (int)(gc - (long)n * a.G16) * 16 + quad * 4;
      const bool pok = gok && p0 < a.L;  // L % 4 == 0: the float4 is entirely in or out
#pragma unroll
      for (int i = 0; i < NCOT; ++i) {
        const int co = co0 + i * 16 + ch;
        av[u][i] = (pok && co < a.Cout)
                       ? *reinterpret_cast<const float4*>(a.dy + ((size_t)n * a.Cout + co) * a.L + p0)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int j = 0; j < NCIT; ++j) {
        const int ci = ci0 + j * 16 + ch;
        bv[u][j] = (pok && ci < a.Cin)
                       ? *reinterpret_cast<const float4*>(a.x + ((size_t)n * a.Cin + ci) * a.L + p0)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float be[NCIT][4];
#pragma unroll
      for (int j = 0; j < NCIT; ++j) {
        be[j][0] = bv[u][j].x; be[j][1] = bv[u][j].y; be[j][2] = bv[u][j].z; be[j][3] = bv[u][j].w;
        if (ACT != PG_ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) be[j][e] = pg_apply_act(be[j][e], ACT);  // act(0) == 0
        }
      }
#pragma unroll
      for (int i = 0; i < NCOT; ++i) {
        const float ae[4] = {av[u][i].x, av[u][i].y, av[u][i].z, av[u][i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int j = 0; j < NCIT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[e], be[j][e], acc[i][j], 0, 0, 0);
          if (do_bias) accb[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[e], 1.0f, accb[i], 0, 0, 0);
        }
      }
    }
  }
  // cross-wave reduction in LDS, then ONE set of atomics per workgroup
  if (wave > 0) {
    float* dst = red + ((size_t)(wave - 1) * NT) * 256 + lane * 4;
#pragma unroll
    for (int i = 0; i < NCOT; ++i) {
#pragma unroll
      for (int j = 0; j < NCIT; ++j)
        *reinterpret_cast<f32x4*>(dst + (i * NCIT + j) * 256) = acc[i][j];
      *reinterpret_cast<f32x4*>(dst + (NCOT * NCIT + i) * 256) = accb[i];
    }
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < WG_THREADS / 64 - 1; ++w) {
    const float* src = red + ((size_t)w * NT) * 256 + lane * 4;
#pragma unroll
    for (int i = 0; i < NCOT; ++i) {
#pragma unroll
      for (int j = 0; j < NCIT; ++j) acc[i][j] += *reinterpret_cast<const f32x4*>(src + (i * NCIT + j) * 256);
      accb[i] += *reinterpret_cast<const f32x4*>(src + (NCOT * NCIT + i) * 256);
    }
  }
  // D[row = quad*4 + r][col = ch] -> this block's row of the partial buffer
  float* prow = a.part + (size_t)blockIdx.x * a.part_stride;
#pragma unroll
  for (int i = 0; i < NCOT; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + i * 16 + quad * 4 + r;
      if (co < a.Cout) {
#pragma unroll
        for (int j = 0; j < NCIT; ++j) {
          const int ci = ci0 + j * 16 + ch;
          if (ci < a.Cin) prow[(size_t)co * a.Cin + ci] = acc[i][j][r];
        }
        if (do_bias && ch == 0) prow[(size_t)a.Cout * a.Cin + co] = accb[i][r];
      }
    }
  }
}

template <int NCOT, int NCIT>
void launch_pw_wgrad_act(const PwWgArgs& a, dim3 grid, hipStream_t st) {
  const size_t shm = (size_t)3 * (NCOT * NCIT + NCOT) * 256 * sizeof(float);
  switch (a.in_act) {
    case PG_ACT_RELU: hipLaunchKernelGGL((conv_wgrad_pw_kernel<NCOT, NCIT, PG_ACT_RELU>), grid, dim3(WG_THREADS), shm, st, a); break;
    case PG_ACT_ELU:  hipLaunchKernelGGL((conv_wgrad_pw_kernel<NCOT, NCIT, PG_ACT_ELU>),  grid, dim3(WG_THREADS), shm, st, a); break;
    case PG_ACT_GELU: hipLaunchKernelGGL((conv_wgrad_pw_kernel<NCOT, NCIT, PG_ACT_GELU>), grid, dim3(WG_THREADS), shm, st, a); break;
    default:          hipLaunchKernelGGL((conv_wgrad_pw_kernel<NCOT, NCIT, PG_ACT_NONE>), grid, dim3(WG_THREADS), shm, st, a); break;
  }
}

template <int NCOT>
void launch_pw_wgrad(const PwWgArgs& a, int ncit, dim3 grid, hipStream_t st) {
  if (ncit <= 1) launch_pw_wgrad_act<NCOT, 1>(a, grid, st);
  else if (ncit == 2) launch_pw_wgrad_act<NCOT, 2>(a, grid, st);
  else launch_pw_wgrad_act<NCOT, 4>(a, grid, st);
}

// ---- second stage: deterministic reduction of the per-block partials --------------------------
// (A chain of G same-address fp32 atomics costs ~65-100 ns per link on MI355X — measured: the
// atomic flush of 512-1024 workgroups dominated these kernels — so each workgroup stores its
// partial tile instead and this kernel sums the G rows: coalesced, atomic-free, run-to-run
// deterministic.)  slot s < Cout*Cin*T -> dw[co][ci][u_t][v_t]; then Cout bias slots.
struct RedArgs {
  const float* part; float* dw; float* db;
  long part_stride; int G, Cout, Cin, KH, KW, T;
  int tap_u[PG_MAX_TAPS];
  int tap_v[PG_MAX_TAPS];
};

// block = 64 consecutive slots x 4 row groups: every load instruction covers 256 contiguous bytes of
// one partial row (the first version read 32-byte pieces of 32 rows: 8.3 us per call on PixelSNAIL's
// 16 K-slot gradients); each thread sums G/4 rows with 8 loads in flight, the groups meet in LDS.
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const RedArgs a) {
  __shared__ float red[4][64];
  const int sl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const long s = (long)blockIdx.x * 64 + sl;
  const long nw = (long)a.Cout * a.Cin * a.T;
  const long total = nw + (a.db ? a.Cout : 0);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (s < total) {
    const float* p = a.part + s;
    int g = rg;
    for (; g + 28 < a.G; g += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += p[(size_t)(g + 4 * u) * a.part_stride];
    }
    for (; g < a.G; g += 4) acc[0] += p[(size_t)g * a.part_stride];
  }
  red[rg][sl] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (rg != 0 || s >= total) return;
  const float sum = (red[0][sl] + red[1][sl]) + (red[2][sl] + red[3][sl]);
  if (s < nw) {
    const int t = (int)(s % a.T);
    const long cc = s / a.T;  // co*Cin + ci
    a.dw[(cc * a.KH + a.tap_u[t]) * a.KW + a.tap_v[t]] += sum;
  } else {
    a.db[s - nw] += sum;
  }
}

int launch_reduce(const float* part, long stride, int G, float* dw, float* db, int Cout, int Cin,
                  int KH, int KW, int T, const int* tap_u, const int* tap_v, hipStream_t st) {
  RedArgs r;
  r.part = part; r.dw = dw; r.db = db; r.part_stride = stride; r.G = G;
  r.Cout = Cout; r.Cin = Cin; r.KH = KH; r.KW = KW; r.T = T;
  for (int t = 0; t < T; ++t) { r.tap_u[t] = tap_u[t]; r.tap_v[t] = tap_v[t]; }
  const long total = (long)Cout * Cin * T + (db ? Cout : 0);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, r);
  return 0;
}

inline int pad_stride(int s) {  // smallest s' >= s with s' % 32 == 2
  int r = s % 32;
  return r <= 2 ? s + (2 - r) : s + (34 - r);
}

}  // namespace

// conv_wgrad_small.hip: input layers (<= 4 input channels, many taps); same return convention
int pg_wgrad_small_launch(const float* x, const float* dy, float* part, long part_stride, long max_rows,
                          int has_bias, int N, int Cin, int IH, int IW, int Cout, int OH, int OW, int T,
                          const int* tap_dr, const int* tap_dc, int in_act, hipStream_t st);

// conv_wgrad_b3.hip: the bf16x3 kernel for the shapes it takes (returns the partial rows it wrote)
int pg_wgrad_b3_launch(const float* x, const float* dy, float* part, long part_stride, long max_rows,
                       int has_bias, int N, int Cin, int IH, int IW, int Cout, int OH, int OW, int T,
                       const int* tap_dr, const int* tap_dc, int in_act, hipStream_t st);

// ... and its "shifted dy" variant (32 output channels per workgroup, full tap grids, W % 4 == 0)
int pg_wgrad_b3s_launch(const float* x, const float* dy, float* part, long part_stride, long max_rows,
                        int has_bias, int N, int Cin, int IH, int IW, int Cout, int OH, int OW, int T,
                        const int* tap_dr, const int* tap_dc, int in_act, hipStream_t st);

static long wgrad_max_rows(int Cout, int Cin, int T) {
  const long chunks = (long)((Cout + 63) / 64) * ((Cin + 63) / 64);
  // one tap with Cout % 128 == 0 and Cin % 64 == 0: conv_wgrad_b3's 128 x 64 tiles — half as many (co, ci) chunks, so
  // twice the partial rows for the same number of workgroups
  // (round 6) one tap with Cout % 128 == 0 and Cin % 128 == 0: the ring kernel's 128 x 128 tiles — a quarter of the chunks, two
  // workgroups per CU want 512 / (chunks / 4) rows
  const long g = ((T == 1 && Cout % 128 == 0 && Cin % 128 == 0) ? 2048 : (Cout % 128 == 0 && Cin % 64 == 0) ? 1024 : 512) / chunks;  // (any tap count: the ring's 128 x 64 tiles)
  return g > 64 ? g : 64;
}

PG_EXPORT size_t pg_conv2d_wgrad_workspace_floats(int Cout, int Cin, int T) {
  return (size_t)wgrad_max_rows(Cout, Cin, T) * ((size_t)Cout * Cin * T + Cout);
}

PG_EXPORT int pg_conv2d_wgrad(const float* x, const float* dy, float* dw, float* db, int N,
                              int Cin, int IH, int IW, int Cout, int OH, int OW, int KH, int KW,
                              int T, const int* tap_dr, const int* tap_dc, const int* tap_u,
                              const int* tap_v, int in_act, float* workspace,
                              size_t workspace_floats, void* stream) {
  PG_REQUIRE(x && dy && dw && tap_dr && tap_dc && tap_u && tap_v && workspace, PG_EINVAL,
             "pg_conv2d_wgrad: null pointer");
  PG_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0, PG_EINVAL,
             "pg_conv2d_wgrad: non-positive dimension");
  PG_REQUIRE(T >= 1 && T <= PG_MAX_TAPS, PG_ESHAPE, "pg_conv2d_wgrad: T=%d not in [1,%d]", T,
             PG_MAX_TAPS);
  PG_REQUIRE(in_act >= PG_ACT_NONE && in_act <= PG_ACT_GELU, PG_EINVAL, "pg_conv2d_wgrad: bad in_act");
  PG_REQUIRE(workspace_floats >= pg_conv2d_wgrad_workspace_floats(Cout, Cin, T), PG_EINVAL,
             "pg_conv2d_wgrad: workspace of %zu floats < required %zu", workspace_floats,
             pg_conv2d_wgrad_workspace_floats(Cout, Cin, T));
  for (int t = 0; t < T; ++t)
    PG_REQUIRE(tap_u[t] >= 0 && tap_u[t] < KH && tap_v[t] >= 0 && tap_v[t] < KW, PG_EINVAL,
               "pg_conv2d_wgrad: tap %d outside the %dx%d filter", t, KH, KW);
  hipStream_t st = (hipStream_t)stream;
  const long stride = (long)Cout * Cin * T + Cout;
  const long max_rows = wgrad_max_rows(Cout, Cin, T);
  // 1x1 convolutions whose shape the bf16x3 kernel takes (Cout % 64 == 0, Cin % 32 == 0, W % 8 == 0) go there
  // first (PG_WGRAD_B3_PW=0: the fp32 direct-fragment kernel below, for A/B)
  static const bool b3_pw = []() { const char* e = PG_AB_ENV("PG_WGRAD_B3_PW"); return !(e && e[0] == '0'); }();
  if (b3_pw && T == 1 && KH == 1 && KW == 1 && tap_dr[0] == 0 && tap_dc[0] == 0 && IH == OH && IW == OW) {
    const int g = pg_wgrad_b3_launch(x, dy, workspace, stride, max_rows, db != nullptr, N, Cin, IH, IW,
                                     Cout, OH, OW, T, tap_dr, tap_dc, in_act, st);
    PG_REQUIRE(g >= 0, PG_EINVAL, "pg_conv2d_wgrad(bf16x3, 1x1): launch failed");
    if (g > 0) {
      launch_reduce(workspace, stride, g, dw, db, Cout, Cin, KH, KW, T, tap_u, tap_v, st);
      PG_LAUNCH_CHECK("pg_conv2d_wgrad(reduce)");
      return 0;
    }
  }
  if (T == 1 && KH == 1 && KW == 1 && tap_dr[0] == 0 && tap_dc[0] == 0 && IH == OH && IW == OW &&
      ((OH * OW) % 4 == 0) && (((uintptr_t)x | (uintptr_t)dy) & 15) == 0) {
    PwWgArgs p;
    p.x = x; p.dy = dy; p.dw = dw; p.db = db; p.part = workspace; p.part_stride = stride;
    p.N = N; p.Cin = Cin; p.Cout = Cout; p.L = OH * OW; p.in_act = in_act;
    p.G16 = (p.L + 15) / 16;
    p.total_groups = (long)N * p.G16;
    const int co_chunks = (Cout + 63) / 64, ci_chunks = (Cin + 63) / 64;
    const int ncot = ((Cout < 64 ? Cout : 64) + 15) / 16, ncit = ((Cin < 64 ? Cin : 64) + 15) / 16;
    long gx = (p.total_groups + 15) / 16;
    if (gx > max_rows) gx = max_rows;
    dim3 grid((unsigned)gx, (unsigned)co_chunks, (unsigned)ci_chunks);
    if (ncot <= 1) launch_pw_wgrad<1>(p, ncit, grid, st);
    else if (ncot == 2) launch_pw_wgrad<2>(p, ncit, grid, st);
    else if (ncot == 3) launch_pw_wgrad<3>(p, ncit, grid, st);  // 48 = the merged q|k|v projection
    else launch_pw_wgrad<4>(p, ncit, grid, st);
    PG_LAUNCH_CHECK("pg_conv2d_wgrad(1x1)");
    launch_reduce(workspace, stride, (int)gx, dw, db, Cout, Cin, KH, KW, T, tap_u, tap_v, st);
    PG_LAUNCH_CHECK("pg_conv2d_wgrad(reduce)");
    return 0;
  }
  {
    const int g = pg_wgrad_small_launch(x, dy, workspace, stride, max_rows, db != nullptr, N, Cin, IH, IW,
                                        Cout, OH, OW, T, tap_dr, tap_dc, in_act, st);
    PG_REQUIRE(g >= 0, PG_EINVAL, "pg_conv2d_wgrad(input layer): launch failed");
    if (g > 0) {
      launch_reduce(workspace, stride, g, dw, db, Cout, Cin, KH, KW, T, tap_u, tap_v, st);
      PG_LAUNCH_CHECK("pg_conv2d_wgrad(reduce)");
      return 0;
    }
  }
  {
    // 64 output channels per workgroup where the shape divides that way (the kernel PixelSNAIL / GatedPixelCNN were
    // tuned on), else — or when its x copies do not fit LDS (3x3 on 64-wide rows) — the shifted-dy variant
    // (measured, PG_WGRAD_B3S=2 prefers the shifted-dy kernel everywhere: PixelSNAIL 12.43 -> 12.31 k img/s,
    // GatedPixelCNN 5.26 -> 5.06 k, beta-VAE 38.1 -> 38.5 k: only the 3x3 grids gain, so those go there first)
    static const bool s_all = []() { const char* e = PG_AB_ENV("PG_WGRAD_B3S"); return e && e[0] == '2'; }();
    const bool s_first = s_all || T == 9;
    int g = (Cout % 64 == 0 && !s_first) ? pg_wgrad_b3_launch(x, dy, workspace, stride, max_rows, db != nullptr, N, Cin, IH, IW,
                                                Cout, OH, OW, T, tap_dr, tap_dc, in_act, st)
                           : 0;
    if (g == 0)
      g = pg_wgrad_b3s_launch(x, dy, workspace, stride, max_rows, db != nullptr, N, Cin, IH, IW, Cout, OH, OW, T,
                              tap_dr, tap_dc, in_act, st);
    if (g == 0 && (Cout % 64 != 0 || s_first))
      g = pg_wgrad_b3_launch(x, dy, workspace, stride, max_rows, db != nullptr, N, Cin, IH, IW, Cout, OH, OW, T,
                             tap_dr, tap_dc, in_act, st);
    PG_REQUIRE(g >= 0, PG_EINVAL, "pg_conv2d_wgrad(bf16x3): launch failed");
    if (g > 0) {
      launch_reduce(workspace, stride, g, dw, db, Cout, Cin, KH, KW, T, tap_u, tap_v, st);
      PG_LAUNCH_CHECK("pg_conv2d_wgrad(reduce)");
      return 0;
    }
  }
  WgArgs a;
  a.x = x; a.dy = dy; a.dw = dw; a.db = db;
  a.N = N; a.Cin = Cin; a.IH = IH; a.IW = IW; a.Cout = Cout; a.OH = OH; a.OW = OW;
  a.KH = KH; a.KW = KW; a.T = T; a.in_act = in_act;
  int min_dr = tap_dr[0], max_dr = tap_dr[0], min_dc = tap_dc[0], max_dc = tap_dc[0];
  a.part = workspace; a.part_stride = stride;
  for (int t = 0; t < T; ++t) {
    min_dr = tap_dr[t] < min_dr ? tap_dr[t] : min_dr;
    max_dr = tap_dr[t] > max_dr ? tap_dr[t] : max_dr;
    min_dc = tap_dc[t] < min_dc ? tap_dc[t] : min_dc;
    max_dc = tap_dc[t] > max_dc ? tap_dc[t] : max_dc;
  }
  const int hr = max_dr - min_dr, hc = max_dc - min_dc;
  a.min_dr = min_dr; a.min_dc = min_dc;
  a.SW = OW + hc;
  const int nco = Cout < CO_CHUNK ? ((Cout + 15) / 16) * 16 : CO_CHUNK;
  const int nco_alloc = nco == 48 ? 64 : nco;
  const int nci = Cin < CI_CHUNK ? ((Cin + 15) / 16) * 16 : CI_CHUNK;
  // per-row cost in floats: (nco+nci)*SW ; fixed: nci*(hr*SW + hc + 48) + nco*48
  const int fixed = nci * (hr * a.SW + hc + 48) + nco_alloc * 48;
  int TR = (LDS_BUDGET_FLOATS - fixed) / ((nco_alloc + nci) * a.SW);
  PG_REQUIRE(TR >= 1, PG_ESHAPE, "pg_conv2d_wgrad: row of %d (+%d halo) too wide for LDS", OW, hc);
  if (TR > OH) TR = OH;
  // register-prefetch staging: rows of both tensors 16-byte aligned in global memory
  static const bool pf_on = []() { const char* e = PG_AB_ENV("PG_WGRAD_PREFETCH"); return !(e && e[0] == '0'); }();
  a.prefetch = pf_on && (OW % 4 == 0) && (IW % 4 == 0) && ((((uintptr_t)x | (uintptr_t)dy) & 15) == 0);
  a.Qd = OW / 4;
  a.Qx = IW / 4;
  if (a.prefetch) {
    while (TR > 1 && ((long)nco * TR * a.Qd > (long)DS * WG_THREADS ||
                      (long)nci * (TR + hr) * a.Qx > (long)XSW * WG_THREADS))
      --TR;
    if ((long)nco * TR * a.Qd > (long)DS * WG_THREADS ||
        (long)nci * (TR + hr) * a.Qx > (long)XSW * WG_THREADS || TR + hr >= 2048)
      a.prefetch = 0;
  }
  a.TR = TR;
  a.xh = TR + hr;
  a.tiles_per_img = (OH + TR - 1) / TR;
  a.total_tiles = N * a.tiles_per_img;
  const int kq = 16;  // npos multiple of 4*ks for every ks in {1,2,4}
  a.npos = ((TR * a.SW + kq - 1) / kq) * kq;
  a.S_dy = pad_stride(a.npos);
  a.dslots = nco * a.TR * a.Qd;
  a.xslots = nci * a.xh * a.Qx;
  a.S_x = pad_stride(a.npos + hr * a.SW + hc + 4);
  for (int t = 0; t < T; ++t) {
    a.tapoff[t] = (tap_dr[t] - min_dr) * a.SW + (tap_dc[t] - min_dc);
    a.tap_u[t] = tap_u[t];
    a.tap_v[t] = tap_v[t];
  }
  int shift = 0;
  while ((1 << shift) < a.SW && shift < 6) ++shift;
  a.swp_shift = shift;
  a.rpi = 64 >> shift;
  a.inv_TR = 1.0f / (float)a.TR;
  a.inv_xh = 1.0f / (float)a.xh;
  a.vec_dy = (OW % 4 == 0) && (((uintptr_t)dy & 15) == 0) && (OW / 4 <= 64);
  a.vec_x = (IW % 4 == 0) && (((uintptr_t)x & 15) == 0) && (IW / 4 <= 64);
  a.qshift_dy = a.qshift_x = 0;
  while ((1 << a.qshift_dy) < OW / 4 && a.qshift_dy < 6) ++a.qshift_dy;
  while ((1 << a.qshift_x) < IW / 4 && a.qshift_x < 6) ++a.qshift_x;
  const int co_chunks = (Cout + CO_CHUNK - 1) / CO_CHUNK;
  const int ci_chunks = (Cin + CI_CHUNK - 1) / CI_CHUNK;
  int G = 512 / (co_chunks * ci_chunks);
  if (G < 64) G = 64;
  if (G > a.total_tiles) G = a.total_tiles;
  if (G > max_rows) G = (int)max_rows;
  size_t shmem = ((size_t)nco_alloc * a.S_dy + (size_t)nci * a.S_x) * sizeof(float);
  const int ncit_h = (nci + 15) / 16;
  const int NTsel = T >= 9 ? 9 : (T >= 4 ? 4 : T);  // taps per pass: 1, 2, 3, 4 or 9
  const size_t red_bytes = (size_t)3 * (ncit_h * NTsel + 1) * 256 * sizeof(float);  // cross-wave reduction scratch
  if (shmem < red_bytes) shmem = red_bytes;
  a.dump = (int)(shmem / sizeof(float));
  shmem += 16;
  PG_REQUIRE(shmem <= 160 * 1024, PG_ESHAPE, "pg_conv2d_wgrad: LDS %zu B over budget", shmem);
  dim3 grid((unsigned)G, (unsigned)co_chunks, (unsigned)ci_chunks);
#define PG_WG(NT, NC)                                                                              \
  {                                                                                               \
    if (a.prefetch) hipLaunchKernelGGL((conv_wgrad_kernel<NT, NC, true>), grid, dim3(WG_THREADS), shmem, st, a);  \
    else hipLaunchKernelGGL((conv_wgrad_kernel<NT, NC, false>), grid, dim3(WG_THREADS), shmem, st, a);            \
  }
  if (ncit_h <= 1) {
    switch (NTsel) { case 1: PG_WG(1, 1); break; case 2: PG_WG(2, 1); break; case 3: PG_WG(3, 1); break;
                     case 4: PG_WG(4, 1); break; default: PG_WG(9, 1); break; }
  } else {
    switch (NTsel) { case 1: PG_WG(1, 2); break; case 2: PG_WG(2, 2); break; case 3: PG_WG(3, 2); break;
                     case 4: PG_WG(4, 2); break; default: PG_WG(9, 2); break; }
  }
#undef PG_WG
  PG_LAUNCH_CHECK("pg_conv2d_wgrad");
  launch_reduce(workspace, stride, G, dw, db, Cout, Cin, KH, KW, T, tap_u, tap_v, st);
  PG_LAUNCH_CHECK("pg_conv2d_wgrad(reduce)");
  return 0;
}

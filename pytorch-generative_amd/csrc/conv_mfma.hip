// conv_mfma.hip — stride-1 convolution over an explicit tap list as an implicit GEMM on the fp32
// matrix cores (v_mfma_f32_16x16x4_f32), gfx950.
//
// Same contract as conv_direct.hip (pg_conv2d_taps): replaces torch.nn.Conv2d.forward as reached
// from CausalConv2d.forward (reference nn/convolution.py:41-43), the pad+crop convolutions of
// gated_pixel_cnn.py:63-96 / pixel_snail.py:41-55, every 1x1 convolution — and, with the
// transposed weight pack and negated taps, aten::convolution_backward's data gradient.
//
//   out[n,co,r,c] = epi( bias[co] + sum_{ci,t} W[co][ci][t] * act(in[n,ci,r+dr_t,c+dc_t]) )
//
// GEMM view: M = output channels, N = output pixels, K = (channel group of 4, tap): K step
// s = g*T + t contracts input channels 4g..4g+3 at tap t.
//   A[i = lane&15][k = lane>>4] = W[co0 + 16m + i][4g + k][t]            (pre-packed fragments)
//   B[k = lane>>4][j = lane&15] = x_tile[4g + k][pixel j + tap offset t]  (LDS, fp32)
//   D[(lane>>4)*4 + r][lane&15]  -> out[co0 + 16m + 4(lane>>4) + r][pixel j]
// CDNA4 mapping
//  * workgroup = 4 waves = up to 256 output pixels (whole rows of one image, or several whole
//    small images) x up to 64 output channels; wave w owns NT consecutive 16-pixel groups and ALL
//    channel tiles: MT x NT accumulator tiles (64 VGPRs at 4 x 4), 1 + MT + NT ds_read_b32 per
//    MT*NT MFMAs (33 cycles each) — the matrix pipe, not LDS, is the bound.
//  * x tile [ci][rows + halo][cols + halo] fp32; every thread owns up to 6 float4 'slots' of a channel
//    chunk, loads them for chunk c+1 before the MFMA loop of chunk c and commits them (prologue
//    activation applied once) after it, so global latency hides under the matrix pipe; channel stride == 16 (mod 32) so the two channels a
//    32-lane half of a B fragment read touches land on disjoint bank halves.
//  * weights arrive as ready-made A fragments (pg_pack_conv_weight_frag) and are copied to LDS with
//    float4 loads; every lane reads its own dword (conflict free).
#include "common.h"
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MF_THREADS = 256;
constexpr int MF_CO_CHUNK = 64;

struct MfArgs {
  const float* in;
  const float* wfrag;
  const float* bias;
  const float* res;
  const float* dact_src;
  float* out;
  int N, Cin, IH, IW, Cout, OH, OW, T;
  int NI, TR, tiles_per_grp, tile_h, tile_w, min_dr, min_dc, img_stride, ch_stride, CIB;
  int KQ;  // K steps of the whole problem: ceil(Cin / 4) * T
  int in_act, dact, out_act;
  int w_off, b_off, dump, buf_stride;  // LDS float offsets (w_off / dump inside a buffer)
  int Q, xslots;    // float4 staging: quads per input row, NI * CIB * tile_h * Q slots per chunk
  int tapoff[PG_MAX_TAPS];
};

constexpr int XS = 5;  // float4 staging slots per thread per channel chunk (x tile)
constexpr int WS = 4;  // float4 slots per thread per channel chunk (weight fragments)

template <int MT, int NT>
__global__ void __launch_bounds__(MF_THREADS, 2) conv_mfma_kernel(const MfArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // PERSISTENT workgroups: gridDim.x is a multiple of the row tiles per image group, so a workgroup
  // keeps ONE row-tile index and walks the image groups g = g0, g0 + gstep, ...: the staging-slot
  // geometry is computed once, and the (tile, channel chunk) steps form one software pipeline — the
  // loads of step s+1 (possibly the next tile's first chunk) fly under the MFMA loop of step s, the
  // epilogue's stores under the next tile's work. (One tile per workgroup measured 47 % MFMA busy:
  // every tile paid dispatch + set-up + a cold first load + the store tail, ~20 us per generation.)
  const int rt = blockIdx.x % a.tiles_per_grp;
  const int g0 = blockIdx.x / a.tiles_per_grp, gstep = gridDim.x / a.tiles_per_grp;
  const int row0 = rt * a.TR;
  const int rows = min(a.TR, a.OH - row0);  // NI > 1 implies TR == OH
  const int co0 = blockIdx.y * MF_CO_CHUNK;
  const int L = a.OH * a.OW;
  const int plane = a.IH * a.IW;
  const int ngroups = (a.N + a.NI - 1) / a.NI;
  const int nchunk = (a.Cin + a.CIB - 1) / a.CIB;
  const int ntiles = g0 < ngroups ? (ngroups - g0 + gstep - 1) / gstep : 0;
  const int nsteps = ntiles * nchunk;
  if (nsteps == 0) return;

  // lane's pixel of each of its 16-pixel groups (MFMA phase) and of the store phase
  int pixoff[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int p = (wave * NT + n) * 16 + (lane & 15);
    const int pc = p < a.NI * rows * a.OW ? p : 0;
    const int q = pc / a.OW;
    const int c = pc - q * a.OW;
    const int img = q / rows;
    const int r = q - img * rows;
    pixoff[n] = img * a.img_stride + r * a.tile_w + c;
  }
  const int pw = wave * (NT * 16) + lane;  // this lane's pixel in the store phase
  size_t so_rel;
  {
    const int pc = (lane < NT * 16 && pw < a.NI * rows * a.OW) ? pw : 0;
    const int q = pc / a.OW;
    const int c = pc - q * a.OW;
    const int img = q / rows;
    const int r = q - img * rows;
    so_rel = ((size_t)img * a.Cout + co0) * L + (size_t)((row0 + r) * a.OW + c);
  }

  // ---- staging slots: this thread's float4 share of a chunk's x tile is the same set of (image,
  // channel-in-chunk, tile row, quad) elements for every step
  int s_goff[XS], s_meta[XS];  // meta: (LDS offset + 4) | element mask << 16 | channel << 20 | image << 28
  int s_base = 0;              // bit k: slot k exists (in-range row)
#pragma unroll
  for (int k = 0; k < XS; ++k) {
    int e = tid + k * MF_THREADS;
    const bool in = e < a.xslots;
    e = in ? e : 0;
    const int q = e % a.Q;
    e /= a.Q;
    const int tr = e % a.tile_h;
    e /= a.tile_h;
    const int ch = e % a.CIB;
    const int img = e / a.CIB;
    const int ir = row0 + a.min_dr + tr;
    const bool ok = in && ir >= 0 && ir < a.IH;
    const int irc = ir < 0 ? 0 : (ir >= a.IH ? a.IH - 1 : ir);
    s_goff[k] = (img * a.Cin + ch) * plane + irc * a.IW + 4 * q;
    const int tcol = 4 * q - a.min_dc;
    const int loff = img * a.img_stride + ch * a.ch_stride + tr * a.tile_w + tcol;
    int mask = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (tcol + i >= 0 && tcol + i < a.tile_w) mask |= 1 << i;
    s_meta[k] = ((loff + 4) & 0xffff) | (mask << 16) | (ch << 20) | (img << 28);  // tcol >= -3: loff + 4 > 0
    if (ok) s_base |= 1 << k;
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // zero both tiles once: the float4 staging writes in-range elements only (the halo stays zero —
  // the in-range set is the same for every step of this workgroup), and the channels that pad the
  // last group of 4 must hold finite values (their weights are zero)
  for (int i = tid; i < a.CIB * a.ch_stride; i += MF_THREADS) {
    lds[i] = 0.f;
    lds[a.buf_stride + i] = 0.f;
  }
  if (tid < MF_CO_CHUNK) {
    const int co = co0 + tid;
    lds[a.b_off + tid] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
  }

  const float4* wsrc_b = reinterpret_cast<const float4*>(a.wfrag + (size_t)blockIdx.y * a.KQ * (MT * 64));
  float4 xv[XS], wv[WS];
#pragma unroll
  for (int k = 0; k < XS; ++k) xv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < WS; ++k) wv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  int xok = 0;  // slots loaded for the step in flight

  // step -> (tile, chunk): loads of the step's x slots and weight fragments into registers
#define PG_MF_ISSUE(STEP)                                                                 \
  {                                                                                       \
    const int tl_ = (STEP) / nchunk;                                                      \
    const int ci0_ = ((STEP) - tl_ * nchunk) * a.CIB;                                     \
    const int n0_ = (g0 + tl_ * gstep) * a.NI;                                            \
    const int ni_ = min(a.NI, a.N - n0_);                                                 \
    const int cib_ = min(a.CIB, a.Cin - ci0_);                                            \
    const int nw4_ = ((cib_ + 3) >> 2) * a.T * MT * 16;                                   \
    const float4* ws_ = wsrc_b + (size_t)(ci0_ >> 2) * a.T * (MT * 16);                   \
    _Pragma("unroll") for (int k = 0; k < WS; ++k) {                                      \
      const int i = tid + k * MF_THREADS;                                                 \
      if (i < nw4_) wv[k] = ws_[i];                                                       \
    }                                                                                     \
    xok = 0;                                                                              \
    const float* src_ = a.in + ((size_t)n0_ * a.Cin + ci0_) * plane;                      \
    _Pragma("unroll") for (int k = 0; k < XS; ++k) {                                      \
      const bool ok = ((s_base >> k) & 1) && ((s_meta[k] >> 20) & 0xff) < cib_ &&         \
                      ((s_meta[k] >> 28) & 0xf) < ni_;                                    \
      if (ok) {                                                                           \
        xv[k] = *reinterpret_cast<const float4*>(src_ + s_goff[k]);                       \
        xok |= 1 << k;                                                                    \
      }                                                                                   \
    }                                                                                     \
  }
  // (the empty asm makes the slot word opaque per use: otherwise the compiler hoists the 2 x 24 LDS
  //  addresses and their predicates out of the step loop and the kernel spills)
#define PG_MF_COMMIT(ACT, XL)                                               \
  _Pragma("unroll") for (int k = 0; k < XS; ++k) {                          \
    const bool ok = (xok >> k) & 1;                                         \
    int meta_ = s_meta[k];                                                  \
    asm volatile("" : "+v"(meta_));                                         \
    const int loff = (meta_ & 0xffff) - 4;                                  \
    const float e[4] = {xv[k].x, xv[k].y, xv[k].z, xv[k].w};                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                         \
      const bool oki = ok && ((meta_ >> (16 + i)) & 1);                     \
      (XL)[oki ? loff + i : a.dump] = pg_apply_act(e[i], ACT);              \
    }                                                                       \
  }
  // writes the step in flight (x slots with the prologue activation, weight fragments) into buffer BUF
#define PG_MF_COMMIT_ALL(BUF, STEP)                                                      \
  {                                                                                      \
    float* xl_ = lds + (BUF) * a.buf_stride;                                             \
    switch (a.in_act) { /* wave-uniform */                                               \
      case PG_ACT_RELU: PG_MF_COMMIT(PG_ACT_RELU, xl_) break;                            \
      case PG_ACT_ELU:  PG_MF_COMMIT(PG_ACT_ELU, xl_) break;                             \
      case PG_ACT_GELU: PG_MF_COMMIT(PG_ACT_GELU, xl_) break;                            \
      default:          PG_MF_COMMIT(PG_ACT_NONE, xl_) break;                            \
    }                                                                                    \
    const int ci0_ = ((STEP) % nchunk) * a.CIB;                                          \
    const int nw4_ = ((min(a.CIB, a.Cin - ci0_) + 3) >> 2) * a.T * MT * 16;              \
    float4* wdst_ = reinterpret_cast<float4*>(xl_ + a.w_off);                            \
    _Pragma("unroll") for (int k = 0; k < WS; ++k) {                                     \
      const int i = tid + k * MF_THREADS;                                                \
      if (i < nw4_) wdst_[i] = wv[k];                                                    \
    }                                                                                    \
  }

  // Two LDS buffers: while the waves run the MFMA loop of step s out of one, each wave that finishes
  // commits step s+1 (its loads were issued before the loop) into the other — one barrier per step.
  PG_MF_ISSUE(0)
  // The zero fill above is ordered before the first commit: other threads own the same entries there. (Round 5: without
  // this barrier a wave that started late zeroed entries another wave had already committed. Never seen with the GPU to
  // itself - the fill comes hundreds of cycles before the first loads land - but two PROCESSES sharing the GPU skew the
  // waves of a workgroup enough: PixelSNAIL's input and q/k/v convolutions came out wrong in 2-4 of 15 forwards,
  // tools/exp/conc_forward_selfcheck.py, profiles/README.md round 5 item 16. The loads of step 0 fly under the barrier.)
  __syncthreads();
  PG_MF_COMMIT_ALL(0, 0)
  __syncthreads();
  const int kb = lane >> 4;
  const int gstride = 4 * a.ch_stride;
  constexpr int EPS = 68;
  const float* bl = lds + a.b_off;  // this chunk's bias values
  int cur = 0;
  for (int step = 0; step < nsteps; ++step) {
    const int tl = step / nchunk;
    const int ci0 = (step - tl * nchunk) * a.CIB;
    const int cib = min(a.CIB, a.Cin - ci0);
    const int ng = (cib + 3) >> 2;  // channel groups of this chunk
    const bool more = step + 1 < nsteps;
    if (more) PG_MF_ISSUE(step + 1)  // prefetch: lands under the MFMA loop
    const float* xl = lds + cur * a.buf_stride;
    const float* wl = xl + a.w_off;
    for (int t = 0; t < a.T; ++t) {
      const float* xb = xl + a.tapoff[t] + kb * a.ch_stride;
      const float* wb = wl + t * (MT * 64) + lane;
#pragma unroll 2
      for (int g = 0; g < ng; ++g) {
        float av[MT], bv[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) av[m] = wb[(g * a.T * MT + m) * 64];
#pragma unroll
        for (int n = 0; n < NT; ++n) bv[n] = xb[g * gstride + pixoff[n]];
        // keep the reads together in front of the MFMAs (left alone the scheduler interleaves them
        // one by one and every MFMA waits a full LDS latency)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[n], acc[m][n], 0, 0, 0);
      }
    }
    if (more) PG_MF_COMMIT_ALL(cur ^ 1, step + 1)
    const bool last_chunk = ci0 + a.CIB >= a.Cin;
    if (more || last_chunk) __syncthreads();
    if (last_chunk) {
      // ---- epilogue of this tile: + bias, out_act, + res, * act'(dact_src). Buffer `cur` is free now
      // (every wave passed the barrier) and serves as transposition scratch: the accumulator layout
      // (lane = 4 channels x 1 pixel) would store 64-byte segments; each wave transposes one
      // 16-channel tile at a time ([16 co][64 px], row stride 68: conflict-free both ways) so that
      // a store instruction covers 64 consecutive pixels (256 B) of one channel plane.
      const int n0 = (g0 + tl * gstep) * a.NI;
      const int npx = min(a.NI, a.N - n0) * rows * a.OW;
      const bool sok = (lane < NT * 16) && pw < npx;
      // lanes that store nothing still LOAD the residual / act' operands below: in a partial last
      // image group their pixel lies in an image past N, so they are pointed at the group's first
      // pixel instead (an out-of-allocation read faulted at batch 128 / 512 with 15-image tiles)
      // Lv / cvalid_p: copies the optimiser cannot see through — with visible values it hoists the per-channel byte
      // offsets (cc * L * 4) and store predicates (cc < cvalid) of all 64 channels out of the step loop into scalar
      // registers, which then live in spilled lanes (v_readlane in front of every store)
      int Lv = L;
      asm volatile("" : "+s"(Lv));
      const size_t so = (sok ? so_rel : (size_t)co0 * L) + (size_t)n0 * a.Cout * L;
      // (scratch = the buffer's WEIGHT area, which every commit rewrites completely; the x tile's
      //  zero halo must survive)
      float* ep = lds + cur * a.buf_stride + a.w_off + wave * (16 * EPS);
      // Loads and stores share vmcnt and retire in order, so a load that follows a store drains it
      // (measured: a conditional residual load between the tiles cost a store round trip per tile).
      // Hence: without residual / act' operands the epilogue issues NO loads; with exactly one
      // (a forward residual, or the act' source of a data gradient — the combinations the model
      // code produces) ALL of its values are requested up front.
      const int cvalid = a.Cout - co0;  // channels of this chunk that exist (>= 1)
      const bool fullc = cvalid >= MT * 16;
      int cvalid_p = cvalid;
      asm volatile("" : "+s"(cvalid_p));
      float* outp = a.out + so;
      const bool has_res = a.res != nullptr, has_ds = a.dact_src != nullptr;
#define PG_MF_TILE_BODY(M)                                                                       \
  _Pragma("unroll") for (int n = 0; n < NT; ++n)                                                 \
  _Pragma("unroll") for (int r = 0; r < 4; ++r) ep[(kb * 4 + r) * EPS + n * 16 + (lane & 15)] = acc[M][n][r]; \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
  float v[16];                                                                                   \
  _Pragma("unroll") for (int c = 0; c < 16; ++c) v[c] = ep[c * EPS + lane] + bl[(M) * 16 + c];   \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
  switch (a.out_act) { /* wave-uniform */                                                        \
    case PG_ACT_RELU: _Pragma("unroll") for (int c = 0; c < 16; ++c) v[c] = pg_apply_act(v[c], PG_ACT_RELU); break; \
    case PG_ACT_ELU:  _Pragma("unroll") for (int c = 0; c < 16; ++c) v[c] = pg_apply_act(v[c], PG_ACT_ELU); break;  \
    case PG_ACT_GELU: _Pragma("unroll") for (int c = 0; c < 16; ++c) v[c] = pg_apply_act(v[c], PG_ACT_GELU); break; \
    default: break;                                                                              \
  }
#define PG_MF_TILE_STORE(M)                                                   \
  if (sok) {                                                                  \
    if (fullc) {                                                              \
      _Pragma("unroll") for (int c = 0; c < 16; ++c) outp[(size_t)((M) * 16 + c) * Lv] = v[c]; \
    } else {                                                                  \
      _Pragma("unroll") for (int c = 0; c < 16; ++c) {                        \
        const int cc = (M) * 16 + c;                                          \
        if (cc < cvalid_p) outp[(size_t)cc * Lv] = v[c];                      \
      }                                                                       \
    }                                                                         \
  }
      if (!has_res && !has_ds) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          PG_MF_TILE_BODY(m)
          PG_MF_TILE_STORE(m)
        }
      } else {
        // ov: every value of the (first) extra operand, requested before any store of this tile
        const float* op1 = (has_res ? a.res : a.dact_src) + so;
        const float* op2 = (has_res && has_ds) ? a.dact_src + so : nullptr;  // both: second one per tile
        // (two tiles' worth at a time: 32 registers; one load batch follows stores per pair)
        constexpr int MH = (MT == 4 && NT >= 3) ? 1 : (MT > 2 ? 2 : MT);  // 64 x 192+ px tiles: one tile of operands in flight (registers: no spills)
        float ov[MH][16];
        const int dsel = has_ds ? a.dact : PG_ACT_NONE;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          if (m % MH == 0) {
#pragma unroll
            for (int mm = 0; mm < MH; ++mm)
#pragma unroll
              for (int c = 0; c < 16; ++c) {
                const int cc = (m + mm) * 16 + c;
                ov[mm][c] = op1[(size_t)((fullc || cc < cvalid_p) ? cc : 0) * Lv];
              }
          }
          PG_MF_TILE_BODY(m)
          float sv[16];
          if (has_res) {
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] += ov[m % MH][c];
            if (op2) {
#pragma unroll
              for (int c = 0; c < 16; ++c) {
                const int cc = m * 16 + c;
                sv[c] = op2[(size_t)((fullc || cc < cvalid_p) ? cc : 0) * Lv];
              }
            }
          } else {
#pragma unroll
            for (int c = 0; c < 16; ++c) sv[c] = ov[m % MH][c];
          }
          switch (dsel) {
            case PG_ACT_RELU:
#pragma unroll
              for (int c = 0; c < 16; ++c) v[c] *= pg_act_grad(sv[c], PG_ACT_RELU);
              break;
            case PG_ACT_ELU:
#pragma unroll
              for (int c = 0; c < 16; ++c) v[c] *= pg_act_grad(sv[c], PG_ACT_ELU);
              break;
            case PG_ACT_GELU:
#pragma unroll
              for (int c = 0; c < 16; ++c) v[c] *= pg_act_grad(sv[c], PG_ACT_GELU);
              break;
            case PG_ACT_ELU_OUT:
#pragma unroll
              for (int c = 0; c < 16; ++c) v[c] *= pg_act_grad(sv[c], PG_ACT_ELU_OUT);
              break;
            default: break;
          }
          PG_MF_TILE_STORE(m)
        }
      }
#undef PG_MF_TILE_BODY
#undef PG_MF_TILE_STORE
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      // the scratch buffer receives the commit of step s+2: nobody may still be transposing in it
      if (more) __syncthreads();
    }
    cur ^= 1;
  }
#undef PG_MF_ISSUE
#undef PG_MF_COMMIT
#undef PG_MF_COMMIT_ALL
}

// ---- A-fragment weight pack ----------------------------------------------------------------
// wfrag[chunk][s = g*T + t][m][lane] = Wsel[chunk*64 + 16m + (lane&15)][c = 4g + (lane>>4)][t]
//   transpose==0: Wsel[o][c][t] = w[o][c][u_t][v_t]   (forward:       M = Cout, K channels = Cin)
//   transpose==1: Wsel[o][c][t] = w[c][o][u_t][v_t]   (data gradient: M = Cin,  K channels = Cout)
// zero outside (o >= M, c >= K channels).
struct FragPackJob {
  float* wfrag;
  int transpose, M, Kc, KQ, MT, chunks;
  long total;
};
struct FragPackArgs {
  const float* w;
  int Cout, Cin, KH, KW, T, njobs;
  FragPackJob job[2];  // forward and / or data-gradient fragments of the same weight in one launch
  int tap_u[PG_MAX_TAPS];
  int tap_v[PG_MAX_TAPS];
};

__global__ void pack_frag_kernel(const FragPackArgs p) {
  const long total = p.job[0].total + (p.njobs > 1 ? p.job[1].total : 0);
  for (long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x; i0 < total;
       i0 += (long)gridDim.x * blockDim.x) {
    const bool second = i0 >= p.job[0].total;
    const FragPackJob& jb = p.job[second ? 1 : 0];
    const long i = second ? i0 - p.job[0].total : i0;
    const int lane = (int)(i & 63);
    long rest = i >> 6;
    const int m = (int)(rest % jb.MT);
    rest /= jb.MT;
    const int kq = (int)(rest % jb.KQ);
    const int chunk = (int)(rest / jb.KQ);
    const int o = chunk * MF_CO_CHUNK + m * 16 + (lane & 15);
    const int g = kq / p.T;
    const int t = kq - g * p.T;
    const int c = g * 4 + (lane >> 4);
    float v = 0.f;
    if (o < jb.M && c < jb.Kc) {
      const int co = jb.transpose ? c : o;
      const int ci = jb.transpose ? o : c;
      v = p.w[(((size_t)co * p.Cin + ci) * p.KH + p.tap_u[t]) * p.KW + p.tap_v[t]];
    }
    jb.wfrag[i] = v;
  }
}

inline int mf_mt(int M) { return M >= MF_CO_CHUNK ? 4 : (M + 15) / 16; }
inline int mf_chunks(int M) { return (M + MF_CO_CHUNK - 1) / MF_CO_CHUNK; }

template <int MT, int NT>
void mf_launch_one(const MfArgs& a, dim3 grid, size_t shmem, hipStream_t st) {
  // function-local static initialiser: thread-safe one-time LDS opt-in (main + autograd thread)
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma_kernel<MT, NT>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  (void)attr;
  hipLaunchKernelGGL((conv_mfma_kernel<MT, NT>), grid, dim3(MF_THREADS), shmem, st, a);
}

template <int MT>
int mf_launch(const MfArgs& a, int nt, dim3 grid, size_t shmem, hipStream_t st) {
  switch (nt) {
    case 1: mf_launch_one<MT, 1>(a, grid, shmem, st); break;
    case 2: mf_launch_one<MT, 2>(a, grid, shmem, st); break;
    case 3: mf_launch_one<MT, 3>(a, grid, shmem, st); break;
    default: mf_launch_one<MT, 4>(a, grid, shmem, st); break;
  }
  return 0;
}

static void mf_pack_job(FragPackJob& j, float* wfrag, int Cout, int Cin, int T, int transpose) {
  j.wfrag = wfrag;
  j.transpose = transpose;
  j.M = transpose ? Cin : Cout;
  j.Kc = transpose ? Cout : Cin;
  j.KQ = ((j.Kc + 3) / 4) * T;
  j.MT = mf_mt(j.M);
  j.chunks = mf_chunks(j.M);
  j.total = (long)j.chunks * j.KQ * j.MT * 64;
}

}  // namespace

// conv_b3.hip: the same convolution with its fp32 products on the bf16 matrix pipe (format 2)
int pg_b3_applicable(int Kc, int M, int T, int OH, int OW, int hr, int hc);
size_t pg_b3_frag_floats(int Kc, int M, int T);
int pg_b3_pack2(const float* w, float* wfrag_fwd, float* wfrag_dgrad, int Cout, int Cin, int KH, int KW,
                int T, const int* tap_u, const int* tap_v, hipStream_t st, int gate_order = 0);
int pg_b3_conv(const float* in, const float* wfrag, const float* bias, const float* res, float* out,
               int N, int Cin, int IH, int IW, int Cout, int OH, int OW, int T, const int* tap_dr,
               const int* tap_dc, int in_act, const float* dact_src, int dact, int out_act,
               const float* res2, long res_bs, long res2_bs, hipStream_t st, int gate = 0, const float* gate_res = nullptr,
               float* gate_out = nullptr, float* dual_out2 = nullptr);
int pg_b3_dual_ok(int Cin, int Cout, int OH, int OW);
int pg_b3_gate_fusable(int Cin, int Cout, int T, int OH, int OW, int hr, int hc);

// Geometry of one fp32-MFMA launch: pixel tile, channel chunk, LDS layout. Needs a.N / Cin / IW / Cout / OH / OW / T; hr / hc = row /
// column extent of the tap list. Shared by the launch (pg_conv2d_mfma_ex) and by the routing query (pg_conv_mfma_supported), so that
// "supported" can never promise a shape the launch then refuses (round 5: 5x5 / 7x7 kernels with 13+ active taps were routed here
// and rejected with "exceeds the staging slots" — found by the CPU sweep tests/test_host_cpu.py::test_routing_never_promises_...).
enum { MF_FIT_OK = 0, MF_FIT_SLOTS = 1, MF_FIT_LDS = 2 };
static int mf_geometry(MfArgs& a, int hr, int hc, size_t* shmem_out) {
  const int N = a.N, Cin = a.Cin, IW = a.IW, Cout = a.Cout, OH = a.OH, OW = a.OW, T = a.T;
  // pixel tile: whole rows of one image (TR rows), or NI whole images when an image is <= 128 px
  const int L = OH * OW;
  static const int px_cap = []() { const char* e = PG_AB_ENV("PG_MF_PX"); const int v = e ? atoi(e) : 256; return (v == 64 || v == 128 || v == 192) ? v : 256; }();
  if (L <= 128) {
    a.NI = 256 / L;
    if (a.NI > 15) a.NI = 15;  // the image index of a staging slot is packed into 4 bits
    if (a.NI > N) a.NI = N;
    a.TR = OH;
  } else {
    a.NI = 1;
    a.TR = px_cap / OW;
    if (a.TR < 1) a.TR = 1;
    if (a.TR > OH) a.TR = OH;
    // even out the row tiles (e.g. 28 rows: 4 tiles of 7 instead of 9,9,9,1)
    const int nt_rows = (OH + a.TR - 1) / a.TR;
    a.TR = (OH + nt_rows - 1) / nt_rows;
  }
  a.tiles_per_grp = (OH + a.TR - 1) / a.TR;
  a.tile_h = a.TR + hr;
  a.tile_w = OW + hc;
  a.img_stride = a.tile_h * a.tile_w;
  {
    int cs = a.NI * a.img_stride;
    const int r = cs % 32;
    cs += (r <= 16) ? (16 - r) : (48 - r);  // channel stride == 16 (mod 32): conflict-free B reads
    a.ch_stride = cs;
  }
  const int MT = mf_mt(Cout);
  a.KQ = ((Cin + 3) / 4) * T;
  // channel chunk: x tile + weight fragments of ONE buffer within ~36 KB (two buffers per workgroup,
  // two workgroups per CU)
  const long budget = 36 * 1024 / 4;
  const long per_ci = a.ch_stride + (long)T * MT * 16;
  long CIB = (budget - 16) / per_ci;
  CIB = (CIB / 4) * 4;
  if (CIB < 4) CIB = 4;
  if (CIB > Cin) CIB = ((Cin + 3) / 4) * 4;
  a.Q = IW / 4;
  // x slots: NI * CIB * tile_h * Q float4 per chunk over XS * 256 thread slots
  while (CIB > 4 && (long)a.NI * CIB * a.tile_h * a.Q > (long)XS * MF_THREADS) CIB -= 4;
  while (CIB > 4 && (CIB / 4) * (long)T * MT * 16 > (long)WS * MF_THREADS) CIB -= 4;  // weight slots
  if (!((CIB / 4) * (long)T * MT * 16 <= (long)WS * MF_THREADS && (long)a.NI * CIB * a.tile_h * a.Q <= (long)XS * MF_THREADS))
    return MF_FIT_SLOTS;
  size_t shmem = 0;
  for (;;) {  // the scratch floor of the weight area can push a buffer over its budget: shrink the chunk
    a.CIB = (int)CIB;
    a.xslots = a.NI * a.CIB * a.tile_h * a.Q;
    const long x_floats = CIB * a.ch_stride;
    a.dump = (int)x_floats;
    a.w_off = (int)(((x_floats + 4 + 3) / 4) * 4);
    size_t w_area = (size_t)(CIB / 4) * T * MT * 64;
    if (w_area < (size_t)4 * 16 * 68) w_area = (size_t)4 * 16 * 68;  // doubles as the epilogue's transposition scratch
    a.buf_stride = (int)(((size_t)a.w_off + w_area + 3) / 4 * 4);
    shmem = (size_t)2 * a.buf_stride * sizeof(float);
    a.b_off = (int)(shmem / sizeof(float));
    shmem += MF_CO_CHUNK * sizeof(float);
    if (shmem <= 80 * 1024 || CIB <= 4) break;
    CIB -= 4;
  }
  *shmem_out = shmem;
  return shmem <= 80 * 1024 ? MF_FIT_OK : MF_FIT_LDS;
}

// The matrix-core path pays off once both channel extents fill MFMA tiles; tiny contractions
// (the 1- / 3-channel image convolutions, 4-channel query projections) stay on conv_direct.hip.
// Returns the weight-fragment FORMAT the matrix-core path uses for this problem: 0 = none (use
// pg_conv2d_taps), PG_CONV_FMT_F32 = fp32 MFMA fragments, PG_CONV_FMT_B3 = bf16x3 fragments.
// hr / hc: row / column extent of the tap list (max - min offset).
PG_EXPORT int pg_conv_mfma_supported(int Cin, int Cout, int T, int OH, int OW, int IW, int hr, int hc) {
  if (OW > 256 || T < 1 || T > PG_MAX_TAPS) return 0;
  if (pg_b3_applicable(Cin, Cout, T, OH, OW, hr, hc)) return PG_CONV_FMT_B3;
  if ((IW % 4) != 0) return 0;  // float4 staging slots
  if (OH * OW < 16) return 0;  // tiny images (VD-VAE's 2x2 / 1x1 levels): launch bound either way
  // Round 5: the 3- / 4-channel image convolutions with >= 32 output channels run on the fp32-MFMA kernel too (K padded to 4
  // per tap: the arithmetic is nothing, the kernel's transposed epilogue is what counts — the VALU tap kernel wrote their
  // 64-channel outputs at ~1.1 TB/s): beta-VAE 45.7 -> 47.9 k img/s, PixelSNAIL / GatedPixelCNN +0.7 % (same box;
  // PG_CONV_MFMA_MIN_CIN=8 restores the old routing for A/B. The ONE-channel input layers stay on the tap kernel: ImageGPT's
  // 3x3 1 -> 16 measured no different, PixelCNN's 24-tap 7x7 does not fit this kernel's tap table)
  static const int min_cin = []() { const char* e = PG_AB_ENV("PG_CONV_MFMA_MIN_CIN"); const int v = e ? atoi(e) : 3; return v >= 3 ? v : 3; }();
  if (!((Cin < 8 && Cin >= min_cin && Cout >= 32) || (Cin >= 8 && Cout >= 8))) return 0;
  // ... and only if the launch geometry fits (the batch is not known here: the largest image group, i.e. the most slots and LDS)
  MfArgs g;
  g.N = 1 << 20; g.Cin = Cin; g.IW = IW; g.Cout = Cout; g.OH = OH; g.OW = OW; g.T = T;
  size_t shmem = 0;
  return mf_geometry(g, hr, hc, &shmem) == MF_FIT_OK ? PG_CONV_FMT_F32 : 0;
}

PG_EXPORT size_t pg_conv_frag_floats(int K_channels, int M_channels, int T, int fmt) {
  if (fmt == PG_CONV_FMT_B3 || fmt == PG_CONV_FMT_B3_GATE) return pg_b3_frag_floats(K_channels, M_channels, T);
  const size_t KQ = (size_t)((K_channels + 3) / 4) * T;
  return (size_t)mf_chunks(M_channels) * KQ * mf_mt(M_channels) * 64;
}

PG_EXPORT int pg_pack_conv_weight_frag2(const float* w, float* wfrag_fwd, float* wfrag_dgrad,
                                        int Cout, int Cin, int KH, int KW, int T, const int* tap_u,
                                        const int* tap_v, int fmt_fwd, int fmt_dgrad, void* stream) {
  PG_REQUIRE(w && (wfrag_fwd || wfrag_dgrad) && tap_u && tap_v, PG_EINVAL,
             "pg_pack_conv_weight_frag: null pointer");
  PG_REQUIRE(T >= 1 && T <= PG_MAX_TAPS, PG_ESHAPE, "pg_pack_conv_weight_frag: T=%d not in [1,%d]",
             T, PG_MAX_TAPS);
  for (int t = 0; t < T; ++t)
    PG_REQUIRE(tap_u[t] >= 0 && tap_u[t] < KH && tap_v[t] >= 0 && tap_v[t] < KW, PG_EINVAL,
               "pg_pack_conv_weight_frag: tap %d (%d,%d) outside %dx%d", t, tap_u[t], tap_v[t], KH, KW);
  hipStream_t st = (hipStream_t)stream;
  {
    const bool gate_order = wfrag_fwd && fmt_fwd == PG_CONV_FMT_B3_GATE;  // pg_conv2d_mfma_gate's forward fragments
    float* b3_fwd = (wfrag_fwd && (fmt_fwd == PG_CONV_FMT_B3 || gate_order)) ? wfrag_fwd : nullptr;
    float* b3_dgrad = (wfrag_dgrad && fmt_dgrad == PG_CONV_FMT_B3) ? wfrag_dgrad : nullptr;
    if (b3_fwd || b3_dgrad) {  // both orientations in one launch
      const int rc = pg_b3_pack2(w, b3_fwd, b3_dgrad, Cout, Cin, KH, KW, T, tap_u, tap_v, st, gate_order);
      if (rc) return rc;
      if (b3_fwd) wfrag_fwd = nullptr;
      if (b3_dgrad) wfrag_dgrad = nullptr;
    }
  }
  if (!wfrag_fwd && !wfrag_dgrad) return 0;
  PG_REQUIRE((!wfrag_fwd || fmt_fwd == PG_CONV_FMT_F32) && (!wfrag_dgrad || fmt_dgrad == PG_CONV_FMT_F32),
             PG_EINVAL, "pg_pack_conv_weight_frag: unknown fragment format");
  FragPackArgs p;
  p.w = w; p.Cout = Cout; p.Cin = Cin; p.KH = KH; p.KW = KW; p.T = T;
  p.njobs = 0;
  if (wfrag_fwd) mf_pack_job(p.job[p.njobs++], wfrag_fwd, Cout, Cin, T, 0);
  if (wfrag_dgrad) mf_pack_job(p.job[p.njobs++], wfrag_dgrad, Cout, Cin, T, 1);
  if (p.njobs == 1) p.job[1] = p.job[0];
  for (int t = 0; t < T; ++t) {
    p.tap_u[t] = tap_u[t];
    p.tap_v[t] = tap_v[t];
  }
  const long total = p.job[0].total + (p.njobs > 1 ? p.job[1].total : 0);
  const int blocks = (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_frag_kernel, dim3(blocks), dim3(256), 0, st, p);
  PG_LAUNCH_CHECK("pg_pack_conv_weight_frag");
  return 0;
}

PG_EXPORT int pg_pack_conv_weight_frag(const float* w, float* wfrag, int Cout, int Cin, int KH,
                                       int KW, int T, const int* tap_u, const int* tap_v,
                                       int transpose, int fmt, void* stream) {
  return pg_pack_conv_weight_frag2(w, transpose ? nullptr : wfrag, transpose ? wfrag : nullptr, Cout,
                                   Cin, KH, KW, T, tap_u, tap_v, fmt, fmt, stream);
}

// Round 6: convolution + GatedActivation (+ the block's residual) in one launch — nn/convolution.py:62-66 behind a 2C-channel
// convolution (pixel_snail.py:41-56; gated_pixel_cnn.py:63-96, where the convolution carries a residual of its own: the link /
// vertical-stack sums). out (N, 2C, OH, OW) = the convolution + res (kept: the gate's backward reads it), gate_out
// (N, C, OH, OW) = gate_res + act(out[:, :C]) * sigmoid(out[:, C:]) (res, gate_res may be NULL); 2C a multiple of 128.
PG_EXPORT int pg_conv_gate_fusable(int Cin, int Cout, int OH, int OW, int T, const int* tap_dr, const int* tap_dc) {
  if (!tap_dr || !tap_dc || T < 1 || T > PG_MAX_TAPS) return 0;
  int a0 = tap_dr[0], a1 = tap_dr[0], b0 = tap_dc[0], b1 = tap_dc[0];
  for (int t = 1; t < T; ++t) {
    a0 = tap_dr[t] < a0 ? tap_dr[t] : a0; a1 = tap_dr[t] > a1 ? tap_dr[t] : a1;
    b0 = tap_dc[t] < b0 ? tap_dc[t] : b0; b1 = tap_dc[t] > b1 ? tap_dc[t] : b1;
  }
  return pg_b3_gate_fusable(Cin, Cout, T, OH, OW, a1 - a0, b1 - b0);
}

PG_EXPORT int pg_conv2d_mfma_gate(const float* in, const float* wfrag, const float* bias, const float* res, float* out, int N, int Cin,
                                  int IH, int IW, int Cout, int OH, int OW, int T, const int* tap_dr, const int* tap_dc, int in_act,
                                  int gate, const float* gate_res, float* gate_out, void* stream) {
  PG_REQUIRE(in && wfrag && out && gate_out && tap_dr && tap_dc, PG_EINVAL, "pg_conv2d_mfma_gate: null pointer");
  PG_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0, PG_EINVAL,
             "pg_conv2d_mfma_gate: non-positive dimension");
  PG_REQUIRE(T >= 1 && T <= PG_MAX_TAPS, PG_ESHAPE, "pg_conv2d_mfma_gate: T=%d not in [1,%d]", T, PG_MAX_TAPS);
  PG_REQUIRE(in_act >= PG_ACT_NONE && in_act < PG_ACT_GELU, PG_EINVAL, "pg_conv2d_mfma_gate: bad input activation (none / relu / elu)");
  PG_REQUIRE(gate == PG_GATE_TANH || gate == PG_GATE_IDENTITY, PG_EINVAL, "pg_conv2d_mfma_gate: bad gate id");
  PG_REQUIRE(pg_conv_gate_fusable(Cin, Cout, OH, OW, T, tap_dr, tap_dc), PG_ESHAPE,
             "pg_conv2d_mfma_gate: shape not covered (pg_conv_gate_fusable)");
  return pg_b3_conv(in, wfrag, bias, res, out, N, Cin, IH, IW, Cout, OH, OW, T, tap_dr, tap_dc, in_act, nullptr, PG_ACT_NONE,
                    PG_ACT_NONE, nullptr, 0, 0, (hipStream_t)stream, 1 + gate, gate_res, gate_out);
}

// Round 6: the "dual" 1x1 data gradient of PixelSNAIL's block tail (pixel_snail.py:109-119: both = elu(conv_a(..)) + r_out,
// out = elu(conv_o(elu(both))) + x): conv_o's data gradient owes both producers their ELU derivatives. With
// d = conv1x1(in) * act'(dact_src) (dact_src = both):  out = d * elu'(a) where elu(a) = both - r,  out2 = d * elu'(r) — the
// derivatives from the stored ELU outputs (y > 0 ? 1 : y + 1). Replaces two pg_act_bwd_from_out launches (7 streams) by one more
// read and one more write in this launch's epilogue.
PG_EXPORT int pg_conv_dual_ok(int Cin, int Cout, int OH, int OW) {
  static const bool on = []() { const char* e = PG_AB_ENV("PG_CONV_DUAL"); return !(e && e[0] == '0'); }();
  return on && pg_b3_dual_ok(Cin, Cout, OH, OW);
}

PG_EXPORT int pg_conv2d_mfma_dual(const float* in, const float* wfrag, float* out, float* out2, int N, int Cin, int OH, int OW,
                                  int Cout, const float* dact_src, int dact, const float* r, void* stream) {
  PG_REQUIRE(in && wfrag && out && out2 && dact_src && r, PG_EINVAL, "pg_conv2d_mfma_dual: null pointer");
  PG_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && OH > 0 && OW > 0, PG_EINVAL, "pg_conv2d_mfma_dual: non-positive dimension");
  PG_REQUIRE(dact >= PG_ACT_NONE && dact <= PG_ACT_ELU_OUT && dact != PG_ACT_GELU, PG_EINVAL, "pg_conv2d_mfma_dual: bad derivative id");
  PG_REQUIRE(pg_b3_dual_ok(Cin, Cout, OH, OW), PG_ESHAPE, "pg_conv2d_mfma_dual: shape not covered (pg_conv_dual_ok)");
  const int zero = 0;
  return pg_b3_conv(in, wfrag, nullptr, r, out, N, Cin, OH, OW, Cout, OH, OW, 1, &zero, &zero, PG_ACT_NONE, dact_src, dact,
                    PG_ACT_NONE, nullptr, 0, 0, (hipStream_t)stream, 0, nullptr, nullptr, out2);
}

PG_EXPORT int pg_conv2d_mfma(const float* in, const float* wfrag, const float* bias,
                             const float* res, float* out, int N, int Cin, int IH, int IW,
                             int Cout, int OH, int OW, int T, const int* tap_dr,
                             const int* tap_dc, int in_act, const float* dact_src, int dact,
                             int out_act, int fmt, void* stream) {
  return pg_conv2d_mfma_ex(in, wfrag, bias, res, out, N, Cin, IH, IW, Cout, OH, OW, T, tap_dr, tap_dc, in_act,
                           dact_src, dact, out_act, fmt, nullptr, 0, 0, stream);
}

PG_EXPORT int pg_conv2d_mfma_ex(const float* in, const float* wfrag, const float* bias,
                                const float* res, float* out, int N, int Cin, int IH, int IW,
                                int Cout, int OH, int OW, int T, const int* tap_dr,
                                const int* tap_dc, int in_act, const float* dact_src, int dact,
                                int out_act, int fmt, const float* res2, long res_bs, long res2_bs,
                                void* stream) {
  PG_REQUIRE(in && wfrag && out && tap_dr && tap_dc, PG_EINVAL, "pg_conv2d_mfma: null pointer");
  PG_REQUIRE(fmt == PG_CONV_FMT_F32 || fmt == PG_CONV_FMT_B3, PG_EINVAL, "pg_conv2d_mfma: bad format %d", fmt);
  PG_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0, PG_EINVAL,
             "pg_conv2d_mfma: non-positive dimension");
  PG_REQUIRE(T >= 1 && T <= PG_MAX_TAPS, PG_ESHAPE, "pg_conv2d_mfma: T=%d not in [1,%d]", T,
             PG_MAX_TAPS);
  PG_REQUIRE(OW <= 256, PG_ESHAPE, "pg_conv2d_mfma: OW=%d wider than a 256-pixel tile", OW);
  PG_REQUIRE(in_act >= PG_ACT_NONE && in_act <= PG_ACT_GELU && out_act >= PG_ACT_NONE &&
                 out_act <= PG_ACT_GELU, PG_EINVAL, "pg_conv2d_mfma: bad activation id");
  PG_REQUIRE(dact >= PG_ACT_NONE && dact <= PG_ACT_ELU_OUT && ((dact == PG_ACT_NONE) == (dact_src == nullptr)),
             PG_EINVAL, "pg_conv2d_mfma: dact_src / dact mismatch");
  hipStream_t st = (hipStream_t)stream;
  {  // PG_CONV_LOG=1: one line per launch on stderr (which shapes a model sends to which kernel family)
    static const bool log_on = []() { const char* e = getenv("PG_CONV_LOG"); return e && e[0] == '1'; }();
    if (log_on)
      fprintf(stderr, "pg_conv2d_mfma %s N=%d Cin=%d %dx%d Cout=%d %dx%d T=%d in_act=%d out_act=%d res=%d res2=%d dact=%d\n",
              fmt == PG_CONV_FMT_B3 ? "b3" : "f32", N, Cin, IH, IW, Cout, OH, OW, T, in_act, out_act, res != nullptr,
              res2 != nullptr, dact);
  }
  if (fmt == PG_CONV_FMT_B3)
    return pg_b3_conv(in, wfrag, bias, res, out, N, Cin, IH, IW, Cout, OH, OW, T, tap_dr, tap_dc, in_act,
                      dact_src, dact, out_act, res2, res_bs, res2_bs, st);
  PG_REQUIRE(res2 == nullptr && (res_bs == 0 || res_bs == (long)Cout * OH * OW), PG_ESHAPE,
             "pg_conv2d_mfma_ex: a second / strided residual needs the bf16x3 format");
  PG_REQUIRE(!(res && dact_src), PG_ESHAPE,
             "pg_conv2d_mfma_ex: residual + activation derivative together need the bf16x3 format");
  MfArgs a;
  a.in = in; a.wfrag = wfrag; a.bias = bias; a.res = res; a.dact_src = dact_src; a.out = out;
  a.N = N; a.Cin = Cin; a.IH = IH; a.IW = IW; a.Cout = Cout; a.OH = OH; a.OW = OW; a.T = T;
  a.in_act = in_act; a.dact = dact; a.out_act = out_act;
  int min_dr = tap_dr[0], max_dr = tap_dr[0], min_dc = tap_dc[0], max_dc = tap_dc[0];
  for (int t = 1; t < T; ++t) {
    min_dr = tap_dr[t] < min_dr ? tap_dr[t] : min_dr;
    max_dr = tap_dr[t] > max_dr ? tap_dr[t] : max_dr;
    min_dc = tap_dc[t] < min_dc ? tap_dc[t] : min_dc;
    max_dc = tap_dc[t] > max_dc ? tap_dc[t] : max_dc;
  }
  a.min_dr = min_dr; a.min_dc = min_dc;
  PG_REQUIRE((IW % 4) == 0 && (((uintptr_t)in & 15) == 0), PG_ESHAPE,
             "pg_conv2d_mfma: input rows must be 16-byte aligned (IW %% 4 == 0)");
  size_t shmem = 0;
  const int fit = mf_geometry(a, max_dr - min_dr, max_dc - min_dc, &shmem);
  PG_REQUIRE(fit != MF_FIT_SLOTS, PG_ESHAPE, "pg_conv2d_mfma: %d taps x %d-row tile exceeds the staging slots", T, a.tile_h);
  PG_REQUIRE(fit != MF_FIT_LDS, PG_ESHAPE, "pg_conv2d_mfma: tile %dx%d x %d taps needs %zu B of LDS (> 80 KB)", a.tile_h,
             a.tile_w, T, shmem);
  const int MT = mf_mt(Cout);
  const int groups = (N + a.NI - 1) / a.NI;
  for (int t = 0; t < T; ++t) a.tapoff[t] = (tap_dr[t] - min_dr) * a.tile_w + (tap_dc[t] - min_dc);
  const int npx_max = a.NI * a.TR * OW;
  const int nt = (npx_max + 63) / 64;  // 16-pixel groups per wave
  // persistent workgroups: ~2 per CU in total, a multiple of the row tiles per image group
  const int chunks_y = mf_chunks(Cout);
  long want = 512 / chunks_y;
  if (want < a.tiles_per_grp) want = a.tiles_per_grp;
  long gx = (want / a.tiles_per_grp) * a.tiles_per_grp;
  if (gx > (long)groups * a.tiles_per_grp) gx = (long)groups * a.tiles_per_grp;
  dim3 grid((unsigned)gx, (unsigned)chunks_y);
  switch (MT) {
    case 1: mf_launch<1>(a, nt, grid, shmem, st); break;
    case 2: mf_launch<2>(a, nt, grid, shmem, st); break;
    case 3: mf_launch<3>(a, nt, grid, shmem, st); break;
    default: mf_launch<4>(a, nt, grid, shmem, st); break;
  }
  PG_LAUNCH_CHECK("pg_conv2d_mfma");
  return 0;
}

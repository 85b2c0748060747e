// conv_b3_kernels.h — the bf16x3 convolution kernels (conv_b3_kernel, conv_b3_pw_kernel) and their launchers,
// included by TWO translation units: conv_b3.hip instantiates them without the GELU paths (GL = false: what
// PixelCNN / GatedPixelCNN / PixelSNAIL / the ReLU VAEs launch), conv_b3_gelu.hip with them (GL = true: VD-VAE,
// ImageGPT's unfused MLP). Why two sets: the kernels sit at 256 registers; with the branch-free GELU compiled into
// every instantiation the register allocator spilled 48-60 registers in ALL paths (PixelSNAIL 12.3 -> 11.4 k img/s,
// although it never takes a GELU branch) — and two files compile in parallel.
#pragma once
#include "common.h"
#include <stdint.h>
#include <stdlib.h>

constexpr int B3_MAXG = 20;    // groups per chunk (<= 5 K steps)

struct B3Args {
  const float* in;
  const float* wfrag;
  const float* bias;
  const float* res;
  const float* res2;      // second residual operand (data gradients with two pass-through gradients)
  const float* dact_src;
  float* out;
  long res_bs, res2_bs;   // batch strides of res / res2 in floats (a channel slice of a wider tensor)
  float res_scale;        // conv_b3_pw_kernel: res is added res_scale times (2 = both residual operands are ONE tensor)
  int N, Cin, IH, IW, Cout, OH, OW, T;
  int TR, tiles_per_img, tile_h, tile_w, min_dr, min_dc;
  int CIB, cgs, groups, ksteps;   // channels per chunk, CIB / 8, cgs * T, ceil(groups / 4)
  int plane16;                    // 16-byte entries per (channel group, piece) plane (multiple of 16)
  int w_off16, b_off, ep_off, dump16;  // LDS offsets: weights (16-byte units), bias / epilogue scratch (floats), dump entry
  int xslots, wslab4;             // staging slots per step; float4 per step's weight slab
  int in_act, dact, out_act;
  // fused GatedActivation (round 6, conv_b3_kernel<.., GT = true>: Cout == 128, the two 64-channel chunks of a workgroup are the
  // gate's two halves): gate_out (N, 64, L) = gate_res + act(a) * sigmoid(b) with [a | b] = this convolution's output, which is
  // still written to `out` (the gate's backward reads it). gate = 0: none, 1 + PG_GATE_*.
  const float* gate_res;
  float* gate_out;
  int gate;
  // "dual" data gradient (round 6, conv_b3_pw_kernel<.., MS = true>; pixel_snail.py:109-119): the convolution's input was
  // x = elu(a) + r with both producers' ELU derivatives owed by THIS launch (out_pre_scaled protocol). With d = conv * act'(dact_src),
  // dact_src = x and res = r:   out = d * elu'(a) with elu(a) = x - r (the gradient of a's pre-activation),   out2 = d * elu'(r's
  // pre-activation) — the derivatives from the stored ELU outputs (y > 0 ? 1 : y + 1). res is NOT added in this mode.
  float* out2;
  int dbg;                        // ablation switches (PG_B3_DBG; -DPG_ABLATE builds only), 0 otherwise
  int g_tapoff[B3_MAXG];          // per group: tap offset in tile pixels
  int g_cg[B3_MAXG];              // per group: channel group of the chunk
#ifdef PG_ABLATE
  long long* prof;                // per wave: cycles in {MFMA loop, barrier 1, commit, issue, epilogue, barrier 2, total, steps}
#endif
};

// phase clocks of conv_b3_kernel (ablation builds only: tools/exp/b3_phase_prof.py)
#ifdef PG_ABLATE
#define PG_PROF_DECL long long pt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long pc_ = clock64(); const long long pstart_ = pc_;
#define PG_PROF_MARK(I) { const long long n_ = clock64(); pt_[I] += n_ - pc_; pc_ = n_; }
#define PG_PROF_DUMP(WAVES, WAVE, STEPS)                                                              \
  if (a.prof && (threadIdx.x & 63) == 0) {                                                            \
    pt_[6] = clock64() - pstart_; pt_[7] = (STEPS);                                                   \
    long long* d_ = a.prof + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (WAVES) + (WAVE)) * 8;  \
    for (int i_ = 0; i_ < 8; ++i_) d_[i_] = pt_[i_];                                                  \
  }
#else
#define PG_PROF_DECL
#define PG_PROF_MARK(I)
#define PG_PROF_DUMP(WAVES, WAVE, STEPS)
#endif

// Which instantiation takes a launch: decided by the host code of conv_b3.hip, executed in the translation unit
// that holds the instantiations for GL (conv_b3.hip: GL = false; conv_b3_gelu.hip: GL = true).
struct B3Launch {
  int pw;            // 3: conv_b3q_kernel (overlapped, 16 waves), 2: conv_b3p_kernel (pipelined, 4 taps), 1: conv_b3_pw_kernel, 0: conv_b3_kernel
  int MT, nt, CG, ms, w9;
  int gate;          // the wide kernel's gate-fusing instantiation
  dim3 grid;
  size_t shmem;
};


namespace {


typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int B3_THREADS = 256;
constexpr int B3_CO_CHUNK = 64;
constexpr int B3_XS = 4;       // (pixel, channel-group) staging slots per thread per step: 8 loads each
constexpr int B3_WS = 6;       // float4 weight-fragment slots per thread per step of the FORMAT-level plan (conv_b3.hip; the kernel itself
                               // has moved its weight slabs by LDS-DMA since round 5 and holds no weight registers)



__device__ __forceinline__ unsigned int pack2(__bf16 a, __bf16 b) {
  return (unsigned int)__builtin_bit_cast(unsigned short, a) |
         ((unsigned int)__builtin_bit_cast(unsigned short, b) << 16);
}

// 8 fp32 -> three bf16x8 (h, m, l pieces)
__device__ __forceinline__ void split8(const float (&x)[8], u32x4& h, u32x4& m, u32x4& l) {
  __bf16 hh[8], mm[8], ll[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    hh[i] = (__bf16)x[i];
    const float r1 = x[i] - (float)hh[i];
    mm[i] = (__bf16)r1;
    ll[i] = (__bf16)(r1 - (float)mm[i]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = pack2(hh[2 * i], hh[2 * i + 1]);
    m[i] = pack2(mm[2 * i], mm[2 * i + 1]);
    l[i] = pack2(ll[2 * i], ll[2 * i + 1]);
  }
}

// the same split by TRUNCATION for the x tiles staged inside the kernel (h = top 16 bits of x,
// m = top 16 bits of x - h, l = x - h - m: exact, 24 = 8 + 8 + 8 significand bits): and + sub per
// piece instead of convert / unpack / sub — the staging arithmetic is paid per step by every
// workgroup (conv_wgrad_b3.hip measures the difference). Weights keep round-to-nearest pieces
// (packed once per step by b3_pack_kernel); the dropped products stay below one fp32 rounding.
__device__ __forceinline__ void split8t(const float (&x)[8], u32x4& h, u32x4& m, u32x4& l) {
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

#define MFMA16B(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16((A), (B), (C), 0, 0, 0)

// CG = output-channel chunks per workgroup. CG = 1: 4 waves, one 64-channel chunk (two workgroups per
// CU). CG = 2 ("wide", the default for Cout % 128 == 0; PG_CONV_B3_WIDE=0 for A/B): 8 waves = two
// 64-channel chunks x four pixel quarters sharing ONE staged x tile — with one chunk per workgroup every
// chunk re-stages the same tile (loads + activation + split are ~30 % of a launch). Measured on MI355X
// (round 3, tools/exp/conv_ab.py): forward 2x2 64 -> 128 75 -> 68 us, 2x1 256 -> 256 71 -> 64 us, 1x3
// 128 -> 256 59 -> 56 us; GatedPixelCNN 4.27 -> 4.54 k img/s at batch 512. Both halves of a gate input
// also meet in one workgroup. (A float4 "vector epilogue" variant was measured in the same call and
// dropped: 136.6 vs 133 us on the 2x2 64 -> 64, slower combined with CG = 2 — it spilled registers.)
// MS ("multi-stream epilogue"): v * act'(dact_src) + res + res2 with pipelined operand requests — its
// own instantiation, because the three operand buffers cost registers that the common kernels (at 256
// already) must not pay: with the code shared, the plain forward launch went from 136 to 185 us.
// W9 ("wide weights"): 9 weight slots and 2 x slots per thread instead of 6 and 4 — the plan of a 3x3 with 64 output
// channels (8-channel chunks: one channel group x 9 taps = 3 K steps, a 36 KB weight slab per step), which otherwise
// does not fit the staging slots and runs on the fp32-MFMA kernel.
// GT ("gate", round 6; CG = 2, Cout == 128): convolution + GatedActivation (+ residual) of PixelSNAIL's ResidualBlock
// (pixel_snail.py:41-56) in one launch. The fragments come in the gate-interleaved order (PG_CONV_FMT_B3_GATE): each chunk = the
// [a | b] halves of 32 gate channels, so a wave holds both operands of its gate channels in its own accumulators and writes
// z (natural order, for backward) and y = res + act(a) * sigmoid(b) from the same transposed tiles. (A first version kept the natural
// order and exchanged tiles between the two waves of a pixel quarter through their scratch, two barriers per tile: +0.7 % on
// PixelSNAIL instead of the gate kernel's 4.3 % — with one workgroup per CU every barrier behind a global load idles the CU.)
template <bool GL, int MT, int NT, int CG, bool MS = false, bool W9 = false, bool GT = false>
__global__ void __launch_bounds__(B3_THREADS * CG, CG == 1 ? 2 : 1) conv_b3_kernel(const B3Args a) {
  static_assert(!GT || (CG == 2 && MT == 4 && !MS && !W9), "gate fusion: the wide kernel's plain epilogue");
  constexpr int THREADS = B3_THREADS * CG;
  constexpr int XS = W9 ? 2 : (CG == 1 ? B3_XS : (B3_XS + 1) / 2);  // the tile's slots over twice the threads
  extern __shared__ __attribute__((aligned(16))) float lds[];
  u32x4* lds16 = reinterpret_cast<u32x4*>(lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cgp = CG == 1 ? 0 : wave_all >> 2;       // which of the workgroup's output chunks
  const int wave = CG == 1 ? wave_all : wave_all & 3;  // pixel quarter of the tile
  // persistent: one row-tile index per workgroup, images n = n_first, n_first + nstep, ...
  const int rt = blockIdx.x % a.tiles_per_img;
  const int n_first = blockIdx.x / a.tiles_per_img, nstep = gridDim.x / a.tiles_per_img;
  const int row0 = rt * a.TR;
  const int rows = min(a.TR, a.OH - row0);
  const int npx = rows * a.OW;
  const int co0 = (blockIdx.y * CG + cgp) * B3_CO_CHUNK;
  const int L = a.OH * a.OW;
  const int plane = a.IH * a.IW;
  const int nchunk = a.Cin / a.CIB;
  const int ntiles = n_first < a.N ? (a.N - n_first + nstep - 1) / nstep : 0;
  const int nsteps = ntiles * nchunk;
  if (nsteps == 0) return;
  const int kq = lane >> 4;

  int pixoff[NT];  // tile pixel (16-byte entry index) of this lane's pixel of each 16-pixel group
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int p = (wave * NT + n) * 16 + (lane & 15);
    const int pc = p < npx ? p : 0;
    const int r = pc / a.OW;
    pixoff[n] = r * a.tile_w + (pc - r * a.OW);
  }
  const int pw = wave * (NT * 16) + lane;  // store phase: lane = pixel
  const bool sok = (lane < NT * 16) && pw < npx;
  // addresses of the epilogue = uniform base (image, channel: scalar registers) + this 32-bit lane offset (the
  // lane's pixel inside the image); lanes that store nothing address pixel 0 of the tile's first row
  unsigned lane_px;
  {
    const int pc = sok ? pw : 0;
    const int r = pc / a.OW;
    lane_px = (unsigned)((row0 + r) * a.OW + (pc - r * a.OW));
  }
  // per group g = 4 ks + kq: where it lives — plane of its channel group + tap offset (LDS table)
  int* gtab = reinterpret_cast<int*>(lds + a.b_off + B3_CO_CHUNK * CG);
  if (tid < B3_MAXG) gtab[tid] = tid < a.groups ? a.g_cg[tid] * 3 * a.plane16 + a.g_tapoff[tid] : 0;

  // ---- staging slots: (channel group, tile row, tile column) -> 8 channel loads of one pixel
  int s_goff[XS], s_loff[XS];  // global offset of channel cg*8 (floats), LDS entry of piece 0; -1: no slot
#pragma unroll
  for (int k = 0; k < XS; ++k) {
    int e = tid + k * THREADS;
    const bool in = e < a.xslots;
    e = in ? e : 0;
    const int tc = e % a.tile_w;
    e /= a.tile_w;
    const int tr = e % a.tile_h;
    const int cg = e / a.tile_h;
    const int ir = row0 + a.min_dr + tr, ic = a.min_dc + tc;
    const bool ok = in && ir >= 0 && ir < a.IH && ic >= 0 && ic < a.IW;
    s_goff[k] = ok ? (cg * 8) * plane + ir * a.IW + ic : -1;
    s_loff[k] = cg * 3 * a.plane16 + tr * a.tile_w + tc;
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // zero the x planes once (halo / out-of-image entries are never written afterwards)
  for (int i = tid; i < a.cgs * 3 * a.plane16; i += THREADS) lds16[i] = u32x4{0u, 0u, 0u, 0u};
  if (tid < B3_CO_CHUNK * CG) {
    int co = CG == 1 ? co0 + tid : blockIdx.y * CG * B3_CO_CHUNK + tid;
    if constexpr (GT)  // the bias of the gate-interleaved channel this (chunk, tile, row) slot holds
      co = ((tid >> 5) & 1) * (a.Cout >> 1) + (blockIdx.y * 2 + (tid >> 6)) * 32 + (tid & 31);
    lds[a.b_off + tid] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
  }

  __syncthreads();  // the zero fill is ordered before the first commit (other threads own the same entries there)

  // weight slab of this thread's chunk: threads [256 c, 256 c + 256) load and commit chunk c's slab
  const float4* wsrc_b = reinterpret_cast<const float4*>(a.wfrag) +
                         (size_t)(CG == 1 ? blockIdx.y : blockIdx.y * CG + (tid >> 8)) * nchunk * a.wslab4;
  float xv[XS][8];
#pragma unroll
  for (int k = 0; k < XS; ++k)
#pragma unroll
    for (int c = 0; c < 8; ++c) xv[k][c] = 0.f;
  // Round 5: the weight slab of a step goes from global memory STRAIGHT INTO LDS (global_load_lds_dwordx4: 1 KB per
  // wave-instruction, destination = wave-uniform base + lane * 16 — exactly the ready-made fragment layout
  // [k step][co tile][piece][lane]) instead of through 6 float4 registers per thread + 6 ds_write_b128. Measured with the
  // kernel's clocks (profiles/r05_conv_wide_phase_clocks.txt): the commit phase of the wide kernel is bound by the
  // VGPR -> LDS store path (12 x 13-cycle ds_write_b128 per wave and step, half of them weights) and the issue phase spends
  // 1 245 cycles on 22 load instructions. 25-30 registers fewer per instantiation; same-box A/B of the two libraries:
  // GatedPixelCNN 5.88 -> 6.00 k img/s, PixelCNN++ 621 -> 629, every other workload +0.3 ... +0.6 % (profiles/README.md, round 5).
  // The DMA is issued by inline asm (the compiler never emits it, and its counter model must not see it: a visible DMA in
  // flight turns every later wait for an ordinary load into vmcnt(0)).
  // This wave's 1 KB pieces of its chunk's slab: piece = (wave of the chunk) + 4 j.
  const int wg_pieces = a.wslab4 >> 6;                       // 1 KB pieces per slab (a slab is k steps x MT x 3 pieces x 1 KB)
  const float4* wg_src = wsrc_b + (size_t)wave * 64 + lane;  // + chunk * wslab4 + 256 j
  const unsigned wg_dst = (unsigned)(size_t)((__attribute__((address_space(3))) char*)lds) +  // LDS byte address of this wave's piece 0
                          (unsigned)(a.w_off16 + (CG == 1 ? 0 : cgp * a.wslab4) + wave * 64) * 16u;
#define PG_B3_WGLDS(STEP)                                                                  \
  {                                                                                        \
    const int tlw_ = (STEP) / nchunk;                                                      \
    const int chw_ = (STEP) - tlw_ * nchunk;                                               \
    const float4* g_ = wg_src + (size_t)chw_ * a.wslab4;                                   \
    for (int j = wave; j < wg_pieces; j += 4) {                                            \
      unsigned keep_;                                                                      \
      const unsigned d_ = wg_dst + (unsigned)(j - wave) * 1024u;                           \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                   : "=&s"(keep_) : "v"(g_ + (size_t)(j - wave) * 64), "s"(d_) : "memory");  \
    }                                                                                      \
  }

#define PG_B3_ISSUE_X_LOADS()                                                              \
    const float* src_ = a.in + ((size_t)(n_first + tl_ * nstep) * a.Cin + ch_ * a.CIB) * plane; \
    if (!(PG_DBG_BIT(a.dbg, 1) && (STEP_) > 0))                                            \
    _Pragma("unroll") for (int k = 0; k < XS; ++k) {                                       \
      if (s_goff[k] >= 0) {                                                                \
        const float* p_ = src_ + s_goff[k];                                                \
        _Pragma("unroll") for (int c = 0; c < 8; ++c) xv[k][c] = p_[(size_t)c * plane];    \
      }                                                                                    \
    }
#define PG_B3_ISSUE_W()
#define PG_B3_COMMIT_W()
#define PG_B3_ISSUE(STEP)                                                                  \
  {                                                                                        \
    const int STEP_ = (STEP);                                                              \
    const int tl_ = (STEP) / nchunk;                                                       \
    const int ch_ = (STEP) - tl_ * nchunk;                                                 \
    PG_B3_ISSUE_W()                                                                        \
    PG_B3_ISSUE_X_LOADS()                                                                  \
  }
#define PG_B3_COMMIT_X(ACT)                                                                \
  _Pragma("unroll") for (int k = 0; k < XS; ++k) {                                         \
    int lo_ = s_loff[k];                                                                   \
    asm volatile("" : "+v"(lo_));                                                          \
    float e_[8];                                                                           \
    _Pragma("unroll") for (int c = 0; c < 8; ++c) e_[c] = pg_apply_act(xv[k][c], ACT);     \
    u32x4 h_, m_, l_;                                                                      \
    split8t(e_, h_, m_, l_);                                                               \
    const int dst_ = s_goff[k] >= 0 ? lo_ : a.dump16;                                      \
    lds16[dst_] = h_;                                                                      \
    lds16[dst_ + (s_goff[k] >= 0 ? a.plane16 : 0)] = m_;                                   \
    lds16[dst_ + (s_goff[k] >= 0 ? 2 * a.plane16 : 0)] = l_;                               \
  }
#define PG_B3_COMMIT_ALL()                                                                 \
  {                                                                                        \
    switch (a.in_act) { /* wave-uniform */                                                 \
      case PG_ACT_RELU: PG_B3_COMMIT_X(PG_ACT_RELU) break;                                 \
      case PG_ACT_ELU:  PG_B3_COMMIT_X(PG_ACT_ELU) break;                                  \
      case PG_ACT_GELU: if constexpr (GL) { PG_B3_COMMIT_X(PG_ACT_GELU) } break;                                 \
      default:          PG_B3_COMMIT_X(PG_ACT_NONE) break;                                 \
    }                                                                                      \
    PG_B3_COMMIT_W()                                                                       \
  }

  // Pipeline over the (tile, channel chunk) steps of this workgroup, ONE x / weight tile in LDS:
  //   MFMA(s) | barrier | weight slab(s+1) by LDS-DMA, commit x(s+1) (its loads were issued a whole step earlier) under it |
  //   issue x loads(s+2) | [epilogue of the tile that ended at s, in per-wave scratch] | barrier
  // so loads have a whole step (+ an epilogue) to land and the epilogue's stores fly under MFMA(s+1).
  constexpr int EPS = 68;
  const float* bl = lds + a.b_off + (CG == 1 ? 0 : cgp * B3_CO_CHUNK);
  const bf16x8* xl = reinterpret_cast<const bf16x8*>(lds16);
  const bf16x8* wl = reinterpret_cast<const bf16x8*>(lds16 + a.w_off16) + (CG == 1 ? 0 : cgp * a.wslab4) + lane;
  PG_B3_ISSUE(0)
  PG_B3_WGLDS(0)
  PG_B3_COMMIT_ALL()
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the slab has landed
  __syncthreads();
  if (nsteps > 1) PG_B3_ISSUE(1)
  PG_PROF_DECL
  for (int step = 0; step < nsteps; ++step) {
    const int tl = step / nchunk;
    const bool more = step + 1 < nsteps;
    PG_PROF_MARK(5)
    if (!PG_DBG_BIT(a.dbg, 4))
    for (int ks = 0; ks < a.ksteps; ++ks) {
      bf16x8 af[MT][3];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) af[m][pc] = wl[((ks * MT + m) * 3 + pc) * 64];
      const bf16x8* xb = xl + gtab[4 * ks + kq];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const bf16x8 bh = xb[pixoff[n]];
        const bf16x8 bm = xb[pixoff[n] + a.plane16];
        const bf16x8 bo = xb[pixoff[n] + 2 * a.plane16];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          f32x4 c = acc[m][n];
          c = MFMA16B(af[m][2], bh, c);  // small terms first
          c = MFMA16B(af[m][0], bo, c);
          c = MFMA16B(af[m][1], bm, c);
          c = MFMA16B(af[m][1], bh, c);
          c = MFMA16B(af[m][0], bm, c);
          c = MFMA16B(af[m][0], bh, c);
          acc[m][n] = c;
        }
      }
    }
    PG_PROF_MARK(0)
    // step + 1's x loads were issued a whole step ago: retire them HERE (a modelled s_waitcnt vmcnt(0)), so that the commit
    // below needs no wait of its own while the slab DMA — invisible to the compiler's counter model — is in flight
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();  // every wave is done with the tiles: the next commit may overwrite them
    PG_PROF_MARK(1)
    if (more) PG_B3_WGLDS(step + 1)
    if (more && !PG_DBG_BIT(a.dbg, 2)) PG_B3_COMMIT_ALL()
    // the slab has landed (nothing else is in flight). Waiting for it AFTER the issue of step + 2's x loads with a counted
    // vmcnt that leaves those loads in flight was measured and dropped: GatedPixelCNN 5.67 -> 5.61 k (round 5, item 6)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PG_PROF_MARK(2)
    if (step + 2 < nsteps) PG_B3_ISSUE(step + 2)
    PG_PROF_MARK(3)
    const bool last_chunk = (step + 1) % nchunk == 0;
    if (last_chunk && !PG_DBG_BIT(a.dbg, 8)) {
      // ---- epilogue: v = out_act(acc + bias) * act'(dact_src) + res + res2; per-wave transposition
      // scratch of its own (the tiles already hold the next step). The derivative comes BEFORE the
      // residuals: in a data gradient res / res2 are pass-through gradients of the same tensor (skip
      // connections), which the activation derivative of the convolution's own input does not touch.
      const int n_img = n_first + tl * nstep;
      // Lv: the plane size as a value the optimiser cannot see through — with a visible L it hoists the 64 per-channel
      // byte offsets (cc * L * 4) out of the step loop into 128 scalar registers, which then live in spilled lanes
      // (v_readlane in front of every store; the kernel carried ~400 scalar spills)
      int Lv = L;
      asm volatile("" : "+s"(Lv));
      const size_t so = ((size_t)n_img * a.Cout + co0) * Lv;   // uniform
      float* ep = lds + a.ep_off + wave_all * (16 * EPS);
      const int cvalid = a.Cout - co0;
      const bool fullc = cvalid >= MT * 16;   // every channel of the chunk exists: no per-channel store predicate
      int cvalid_p = cvalid;                  // the partial path's copy, opaque for the same reason as Lv
      asm volatile("" : "+s"(cvalid_p));
      float* outp = a.out + so;
      const bool has_res = a.res != nullptr, has_ds = a.dact_src != nullptr, has_res2 = a.res2 != nullptr;
      // epilogue accesses: channel CC of the chunk at this lane's pixel. PG_EP_LDC: a channel beyond the valid ones reads
      // channel 0 (value unused). (A raw-buffer-load form of these accesses and of the staging loads was measured in round 5 and
      // removed: profiles/README.md, round 5, item 2.)
#define PG_EP_RS(NAME, P)
#define PG_EP_LDC(RS, P, CC) ((P) + (size_t)((fullc || (CC) < cvalid_p) ? (CC) : 0) * Lv)[lane_px]
#define PG_EP_LDO(RS, P, OFF, CC) ((P) + (OFF))[lane_px]  /* OFF: the clamped channel offset, shared by the operands */
#define PG_EP_ST(RS, P, CC, V) ((P) + (size_t)(CC) * Lv)[lane_px] = (V)
      PG_EP_RS(rs_out, outp)
#define PG_B3_TILE_BODY(M)                                                                       \
  _Pragma("unroll") for (int n = 0; n < NT; ++n)                                                 \
  _Pragma("unroll") for (int r = 0; r < 4; ++r) ep[(kq * 4 + r) * EPS + n * 16 + (lane & 15)] = acc[M][n][r]; \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
  float v[16];                                                                                   \
  _Pragma("unroll") for (int c = 0; c < 16; ++c) v[c] = ep[c * EPS + lane] + bl[(M) * 16 + c];   \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
  switch (a.out_act) { /* wave-uniform */                                                        \
    case PG_ACT_RELU: _Pragma("unroll") for (int c = 0; c < 16; ++c) v[c] = pg_apply_act(v[c], PG_ACT_RELU); break; \
    case PG_ACT_ELU:  _Pragma("unroll") for (int c = 0; c < 16; ++c) v[c] = pg_apply_act(v[c], PG_ACT_ELU); break;  \
    case PG_ACT_GELU: if constexpr (GL) { _Pragma("unroll") for (int c = 0; c < 16; ++c) v[c] = pg_apply_act(v[c], PG_ACT_GELU); } break; \
    default: break;                                                                              \
  }
#define PG_B3_TILE_STORE(M)                                                   \
  if (sok) {                                                                  \
    if (fullc) {                                                              \
      _Pragma("unroll") for (int c = 0; c < 16; ++c)                          \
        PG_EP_ST(rs_out, outp, (M) * 16 + c, v[c]);                           \
    } else {                                                                  \
      _Pragma("unroll") for (int c = 0; c < 16; ++c) {                        \
        const int cc = (M) * 16 + c;                                          \
        if (cc < cvalid_p) PG_EP_ST(rs_out, outp, cc, v[c]);                  \
      }                                                                       \
    }                                                                         \
  }
      if constexpr (GT) {
        // Gate-interleaved fragments (PG_CONV_FMT_B3_GATE), G = Cout / 2 gate channels: chunk cg = 2 blockIdx.y + cgp holds gate channels
        // 32 cg .. 32 cg + 31; its tile m = z channels G (m >> 1) + 32 cg + 16 (m & 1) + 0..15 — tiles h and 2 + h are the [a | b] halves
        // of gate channels 32 cg + 16 h + 0..15, so the gate needs no exchange between waves. z (+ the convolution's own residual,
        // GatedPixelCNN's link / vertical-stack sums) goes out in its natural channel order (backward reads it), y beside it.
        const int G = a.Cout >> 1;
        const int cg32 = (blockIdx.y * 2 + cgp) * 32;
        float* zb = a.out + (size_t)n_img * a.Cout * Lv;
        const float* zres = has_res ? a.res + (size_t)n_img * a.res_bs : nullptr;
        const size_t go = (size_t)n_img * G * Lv;
#define PG_B3_TILE_T(M, V, CH0)                                                                  \
  _Pragma("unroll") for (int n = 0; n < NT; ++n)                                                 \
  _Pragma("unroll") for (int r = 0; r < 4; ++r) ep[(kq * 4 + r) * EPS + n * 16 + (lane & 15)] = acc[M][n][r]; \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
  _Pragma("unroll") for (int c = 0; c < 16; ++c) V[c] += ep[c * EPS + lane] + bl[(M) * 16 + c];  \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
  if (sok) { _Pragma("unroll") for (int c = 0; c < 16; ++c) (zb + (size_t)((CH0) + c) * Lv)[lane_px] = V[c]; }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int gc0 = cg32 + h * 16;  // uniform
          float r[16], va[16], vb[16];
          // the residuals' loads first: they are in flight over both transpositions
#pragma unroll
          for (int c = 0; c < 16; ++c) r[c] = (a.gate_res && sok) ? (a.gate_res + go + (size_t)(gc0 + c) * Lv)[lane_px] : 0.f;
#pragma unroll
          for (int c = 0; c < 16; ++c) {
            va[c] = (zres && sok) ? (zres + (size_t)(gc0 + c) * Lv)[lane_px] : 0.f;
            vb[c] = (zres && sok) ? (zres + (size_t)(G + gc0 + c) * Lv)[lane_px] : 0.f;
          }
          PG_B3_TILE_T(h, va, gc0)
          PG_B3_TILE_T(2 + h, vb, G + gc0)
#pragma unroll
          for (int c = 0; c < 16; ++c) {
            const float f = a.gate == 1 + PG_GATE_TANH ? tanhf(va[c]) : va[c];
            r[c] += f * (1.f / (1.f + expf(-vb[c])));
          }
          if (sok) {
#pragma unroll
            for (int c = 0; c < 16; ++c) (a.gate_out + go + (size_t)(gc0 + c) * Lv)[lane_px] = r[c];
          }
        }
#undef PG_B3_TILE_T
      } else if (!has_res && !has_ds) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          PG_B3_TILE_BODY(m)
          PG_B3_TILE_STORE(m)
        }
      } else {
        if constexpr (MS) {
        // Up to three operand streams (derivative source, res, res2). Loads and stores share vmcnt and
        // retire in order, so a load issued behind a tile's stores waits for them: the operands of tile
        // m + 1 are therefore requested BEFORE the stores of tile m (single buffer: right after tile
        // m's values have been used); the wait in front of their use then covers the loads only.
        const float* st0 = has_ds ? a.dact_src + so : nullptr;
        const float* st1 = has_res ? a.res + (size_t)co0 * Lv + (size_t)n_img * a.res_bs : nullptr;
        const float* st2 = has_res2 ? a.res2 + (size_t)co0 * Lv + (size_t)n_img * a.res2_bs : nullptr;
        const int dsel = has_ds ? a.dact : PG_ACT_NONE;
        // half tiles (8 channels) at a time: 24 operand registers instead of 48 (the kernel is at its
        // register limit; with whole-tile buffers this instantiation spilled 33 dwords)
        constexpr int HQ = 4, NH = 16 / HQ;
        float o0[HQ], o1[HQ], o2[HQ];
        PG_EP_RS(rs_s0, st0) PG_EP_RS(rs_s1, st1) PG_EP_RS(rs_s2, st2)
#define PG_B3_REQUEST(M, H)                                                                \
  _Pragma("unroll") for (int c = 0; c < HQ; ++c) {                                          \
    const int cc = (M) * 16 + (H) * HQ + c;                                                 \
    const size_t off_ = (size_t)((fullc || cc < cvalid_p) ? cc : 0) * Lv;                    \
    if (st0) o0[c] = PG_EP_LDO(rs_s0, st0, off_, cc);                                      \
    if (st1) o1[c] = PG_EP_LDO(rs_s1, st1, off_, cc);                                      \
    if (st2) o2[c] = PG_EP_LDO(rs_s2, st2, off_, cc);                                      \
  }
        PG_B3_REQUEST(0, 0)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          PG_B3_TILE_BODY(m)
#pragma unroll
          for (int hh = 0; hh < NH; ++hh) {
            switch (dsel) {
              case PG_ACT_RELU:
#pragma unroll
                for (int c = 0; c < HQ; ++c) v[hh * HQ + c] *= pg_act_grad(o0[c], PG_ACT_RELU);
                break;
              case PG_ACT_ELU:
#pragma unroll
                for (int c = 0; c < HQ; ++c) v[hh * HQ + c] *= pg_act_grad(o0[c], PG_ACT_ELU);
                break;
              case PG_ACT_GELU:
if constexpr (GL) {
#pragma unroll
                for (int c = 0; c < HQ; ++c) v[hh * HQ + c] *= pg_act_grad(o0[c], PG_ACT_GELU);
                }
                break;
              case PG_ACT_ELU_OUT:
#pragma unroll
                for (int c = 0; c < HQ; ++c) v[hh * HQ + c] *= pg_act_grad(o0[c], PG_ACT_ELU_OUT);
                break;
              default: break;
            }
            if (st1) {
#pragma unroll
              for (int c = 0; c < HQ; ++c) v[hh * HQ + c] += o1[c];
            }
            if (st2) {
#pragma unroll
              for (int c = 0; c < HQ; ++c) v[hh * HQ + c] += o2[c];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (hh + 1 < NH) { PG_B3_REQUEST(m, hh + 1) }
            else if (m + 1 < MT) { PG_B3_REQUEST(m + 1, 0) }
            __builtin_amdgcn_sched_barrier(0);
            if (sok) {
              if (fullc) {
#pragma unroll
                for (int c = 0; c < HQ; ++c) PG_EP_ST(rs_out, outp, m * 16 + hh * HQ + c, v[hh * HQ + c]);
              } else {
#pragma unroll
                for (int c = 0; c < HQ; ++c) {
                  const int cc = m * 16 + hh * HQ + c;
                  if (cc < cvalid_p) PG_EP_ST(rs_out, outp, cc, v[hh * HQ + c]);
                }
              }
            }
          }
        }
#undef PG_B3_REQUEST
        } else {
          // one or two operand streams as measured in round 2: the first one requested for two tiles
          // before any store, a derivative source behind a residual per tile (res + res -> * act')
        const float* op1 = (has_res ? a.res : a.dact_src) + so;
        const float* op2 = (has_res && has_ds) ? a.dact_src + so : nullptr;
        constexpr int MH = GL ? 1 : (MT > 2 ? 2 : MT);  // GELU instantiations: one tile of operands in flight (registers)
        PG_EP_RS(rs_o1, op1) PG_EP_RS(rs_o2, op2)
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
                ov[mm][c] = PG_EP_LDC(rs_o1, op1, cc);
              }
          }
          PG_B3_TILE_BODY(m)
          float sv[16];
          if (has_res) {
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] += ov[m % MH][c];
            if (op2) {
#pragma unroll
              for (int c = 0; c < 16; ++c) {
                const int cc = m * 16 + c;
                sv[c] = PG_EP_LDC(rs_o2, op2, cc);
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
if constexpr (GL) {
#pragma unroll
              for (int c = 0; c < 16; ++c) v[c] *= pg_act_grad(sv[c], PG_ACT_GELU);
              }
              break;
            case PG_ACT_ELU_OUT:
#pragma unroll
              for (int c = 0; c < 16; ++c) v[c] *= pg_act_grad(sv[c], PG_ACT_ELU_OUT);
              break;
            default: break;
          }
          PG_B3_TILE_STORE(m)
        }
        }
      }
#undef PG_B3_TILE_BODY
#undef PG_B3_TILE_STORE
#undef PG_EP_RS
#undef PG_EP_LDC
#undef PG_EP_LDO
#undef PG_EP_ST
    }
    if (last_chunk) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    PG_PROF_MARK(4)
    if (more) __syncthreads();  // the commit is visible before the next MFMA loop
  }
  PG_PROF_DUMP(THREADS / 64, wave_all, nsteps)
#undef PG_B3_ISSUE
#undef PG_B3_ISSUE_W
#undef PG_B3_COMMIT_W
#undef PG_B3_WGLDS
#undef PG_B3_ISSUE_X_LOADS
#undef PG_B3_COMMIT_X
#undef PG_B3_COMMIT_ALL
}

// ---- the pipelined kernel for 4-tap convolutions (round 4) -----------------------------------------------
// conv_b3_kernel runs its phases back to back: measured with its own clocks (tools/exp/b3_phase_prof.py, PixelSNAIL's
// 2x2 64 -> 64 at N = 512) a wave spends 37 % of its life in the MFMA loop, 26 % committing the next step (activation +
// split + LDS writes), 10 % issuing loads, 20 % in the epilogue and 7 % at its two barriers per step — and because every
// wave of a workgroup is in the same phase between two barriers, nothing overlaps. Here a step is ONE K step of 32 =
// 8 channels x 4 taps (CIB = 8), the x tile and the weight slab are double buffered in LDS (2 x 14.6 KB + 2 x 12 KB),
// and a step is a single stream
//     A fragments | B(n = 0) side(0) 24 MFMA | B(1) side(1) 24 MFMA | B(2) side(2) 24 MFMA | B(3) side(3) 24 MFMA | barrier
// whose "side" slices are the commit of step s + 1 into the OTHER buffers (slot 0, slot 1, weights) and the issue of
// step s + 2's loads: one barrier per step, and between barriers the waves drift apart by whole slices, so one wave's
// VALU / LDS / load slices run under the other waves' MFMAs (bf16 MFMAs co-execute with VALU work).
// Shapes: T = 4 taps (2 x 2 grids: PixelSNAIL's ResidualBlock, the four phase convolutions of the VAEs' 4 x 4 / stride 2
// and transposed convolutions), >= 64 output channels per chunk (MT = 4). Epilogue: v = out_act(acc + bias) *
// act'(dact_src) + res + res2, every operand optional, streamed in quarter tiles.
typedef float f32x2p __attribute__((ext_vector_type(2)));
constexpr int B3P_PX = 352;    // tile pixels with halo the plan allows (= B3_PX_CAP of conv_b3.hip)
constexpr int B3P_W4 = 768;    // 16-byte weight fragments per step: 4 co tiles x 3 pieces x 64 lanes

// WV = waves per workgroup. 4: a wave owns 64 channels x 64 pixels (NT <= 4 groups of 16), 256 registers, two waves per
// SIMD. 8 (round 4, second step): a wave owns 64 channels x 32 pixels (NT <= 2) — 32 accumulator registers instead of
// 64, one staging slot instead of two, 128 registers: FOUR waves per SIMD. Measured with the kernel's clocks, a wave of
// the 4-wave form spends 1536 cycles per step issuing its 96 MFMAs and ~3400 on everything else (commit VALU at ~6
// cycles per instruction under the other wave's MFMAs, LDS and load issue, the barrier): two waves keep the matrix
// pipe 61 % busy, and tools/exp/coexec_ubench.hip shows the pipe itself is not the limit (a second wave's plain VALU
// stream runs under a full-rate MFMA stream). More waves per SIMD is what fills it; the price is 12 + 6 instead of
// 12 + 12 fragment reads per 96 MFMAs of a wave pair (1.5x the LDS read traffic, still below the LDS pipe's rate).
template <bool GL, int NT, int WV>
__global__ void __launch_bounds__(64 * WV, WV / 2) conv_b3p_kernel(const B3Args a) {
  constexpr int MT = 4, THREADS = 64 * WV;
  constexpr int XS = (B3P_PX + THREADS - 1) / THREADS;      // staging slots per thread (2 / 1)
  constexpr int WSL = (B3P_W4 + THREADS - 1) / THREADS;     // weight-fragment slots per thread (3 / 2)
  static_assert(WV == 4 || (WV == 8 && NT <= 2), "8 waves: 32 pixels per wave");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  u32x4* lds16 = reinterpret_cast<u32x4*>(lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rt = blockIdx.x % a.tiles_per_img;
  const int n_first = blockIdx.x / a.tiles_per_img, nstep = gridDim.x / a.tiles_per_img;
  const int row0 = rt * a.TR;
  const int rows = min(a.TR, a.OH - row0);
  const int npx = rows * a.OW;
  const int co0 = blockIdx.y * B3_CO_CHUNK;
  const int L = a.OH * a.OW;
  const int plane = a.IH * a.IW;
  const int nchunk = a.Cin >> 3;
  const int ntiles = n_first < a.N ? (a.N - n_first + nstep - 1) / nstep : 0;
  const int nsteps = ntiles * nchunk;
  if (nsteps == 0) return;
  const int kq = lane >> 4;
  const int xbuf16 = 3 * a.plane16;   // 16-byte entries of one x buffer (one channel group, three pieces)

  int pixoff[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int p = (wave * NT + n) * 16 + (lane & 15);
    const int pc = p < npx ? p : 0;
    const int r = pc / a.OW;
    pixoff[n] = r * a.tile_w + (pc - r * a.OW);
  }
  // 8 waves: store phase of the epilogue, lane & 31 = pixel of the wave's (<= 32-pixel) tile
  unsigned opx32;
  bool sok32;
  {
    const int p = wave * (NT * 16) + (lane & 31);
    sok32 = (lane & 31) < NT * 16 && p < npx;
    const int pc = sok32 ? p : 0;
    const int r = pc / a.OW;
    opx32 = (unsigned)((row0 + r) * a.OW + (pc - r * a.OW));
  }
  const int pw = wave * (NT * 16) + lane;
  const bool sok = (lane < NT * 16) && pw < npx;
  unsigned lane_px;
  {
    const int pc = sok ? pw : 0;
    const int r = pc / a.OW;
    lane_px = (unsigned)((row0 + r) * a.OW + (pc - r * a.OW));
  }
  // this lane's tap (K group kq of the single K step): offset of its B fragments inside an x buffer
  int tapoff;
  {
    int* gtab = reinterpret_cast<int*>(lds + a.b_off + B3_CO_CHUNK);
    if (tid < 4) gtab[tid] = a.g_tapoff[tid];
    __syncthreads();
    tapoff = gtab[kq];
  }
  // staging slots: tile pixel -> 8 channel loads; out-of-image pixels load the tile's first valid word and commit
  // to the dump entry (the halo entries of both buffers are zeroed once and never written)
  int s_goff[XS], s_loff[XS];
  bool s_ok[XS];
#pragma unroll
  for (int k = 0; k < XS; ++k) {
    int e = tid + k * THREADS;
    const bool in = e < a.xslots;
    e = in ? e : 0;
    const int tc = e % a.tile_w;
    const int tr = e / a.tile_w;
    const int ir = row0 + a.min_dr + tr, ic = a.min_dc + tc;
    s_ok[k] = in && ir >= 0 && ir < a.IH && ic >= 0 && ic < a.IW;
    s_goff[k] = s_ok[k] ? ir * a.IW + ic : 0;
    s_loff[k] = s_ok[k] ? tr * a.tile_w + tc : -1;
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int i = tid; i < 2 * xbuf16; i += THREADS) lds16[i] = u32x4{0u, 0u, 0u, 0u};
  if (tid < B3_CO_CHUNK) lds[a.b_off + tid] = (a.bias && co0 + tid < a.Cout) ? a.bias[co0 + tid] : 0.f;
  __syncthreads();  // the zero fill is ordered before the first commit (other threads own the same entries there)

  const float4* wsrc_b = reinterpret_cast<const float4*>(a.wfrag) + (size_t)blockIdx.y * nchunk * B3P_W4;  // uniform
  float xv[XS][8];
  float4 wv0, wv1 = make_float4(0.f, 0.f, 0.f, 0.f), wv2 = wv1;  // (named, not an array: indexed inside the unrolled slice loop an array stays in scratch memory)

  // (tile, chunk) of the step whose loads are issued next
  int l_tl = 0, l_ch = 0;
#define PG_P_ISSUE_X()                                                                         \
  {                                                                                            \
    const float* src_ = a.in + ((size_t)(n_first + l_tl * nstep) * a.Cin + l_ch * 8) * plane;  \
    _Pragma("unroll") for (int k = 0; k < XS; ++k) {                                           \
      /* uniform channel base (scalar registers) + the slot's 32-bit offset */                 \
      _Pragma("unroll") for (int c = 0; c < 8; ++c) xv[k][c] = (src_ + (size_t)c * plane)[(unsigned)s_goff[k]]; \
    }                                                                                          \
  }
#define PG_P_ISSUE_W()                                                                         \
  {                                                                                            \
    const float4* ws_ = wsrc_b + (size_t)l_ch * B3P_W4;                                        \
    if constexpr (WV == 8) { /* 768 fragments over 512 threads: three 8-byte halves each, no predicate */ \
      const f32x2p* w2_ = reinterpret_cast<const f32x2p*>(ws_);                                  \
      const f32x2p h0_ = w2_[(unsigned)tid], h1_ = (w2_ + THREADS)[(unsigned)tid], h2_ = (w2_ + 2 * THREADS)[(unsigned)tid]; \
      wv0.x = h0_[0]; wv0.y = h0_[1]; wv0.z = h1_[0]; wv0.w = h1_[1]; wv1.x = h2_[0]; wv1.y = h2_[1]; \
    } else {                                                                                   \
      wv0 = ws_[(unsigned)tid];                                                                \
      wv1 = (ws_ + THREADS)[(unsigned)tid];                                                    \
      wv2 = (ws_ + 2 * THREADS)[(unsigned)tid];                                                \
    }                                                                                          \
    if (++l_ch == nchunk) { l_ch = 0; ++l_tl; }                                                \
  }
#define PG_P_COMMIT_SLOT(K, BUF, ACT)                                                          \
  {                                                                                            \
    float e_[8];                                                                               \
    _Pragma("unroll") for (int c = 0; c < 8; ++c) e_[c] = pg_apply_act(xv[K][c], ACT);         \
    u32x4 h_, m_, l_;                                                                          \
    split8t(e_, h_, m_, l_);                                                                   \
    const int dst_ = s_ok[K] ? (BUF) * xbuf16 + s_loff[K] : a.dump16;                          \
    const int pst_ = s_ok[K] ? a.plane16 : 0;                                                  \
    lds16[dst_] = h_;                                                                          \
    lds16[dst_ + pst_] = m_;                                                                   \
    lds16[dst_ + 2 * pst_] = l_;                                                               \
  }
#define PG_P_COMMIT_X(K, BUF)                                                                  \
  switch (a.in_act) { /* wave-uniform */                                                       \
    case PG_ACT_RELU: PG_P_COMMIT_SLOT(K, BUF, PG_ACT_RELU) break;                             \
    case PG_ACT_ELU:  PG_P_COMMIT_SLOT(K, BUF, PG_ACT_ELU) break;                              \
    case PG_ACT_GELU: if constexpr (GL) { PG_P_COMMIT_SLOT(K, BUF, PG_ACT_GELU) } break;       \
    default:          PG_P_COMMIT_SLOT(K, BUF, PG_ACT_NONE) break;                             \
  }
#define PG_P_COMMIT_W(BUF)                                                                     \
  {                                                                                            \
    float4* wd_ = reinterpret_cast<float4*>(lds16 + a.w_off16 + (BUF) * B3P_W4) + tid;         \
    if constexpr (WV == 8) {                                                                   \
      f32x2p* w2_ = reinterpret_cast<f32x2p*>(lds16 + a.w_off16 + (BUF) * B3P_W4) + tid;         \
      w2_[0] = f32x2p{wv0.x, wv0.y}; w2_[THREADS] = f32x2p{wv0.z, wv0.w}; w2_[2 * THREADS] = f32x2p{wv1.x, wv1.y}; \
    } else {                                                                                   \
      wd_[0] = wv0; wd_[THREADS] = wv1; wd_[2 * THREADS] = wv2;                                \
    }                                                                                          \
  }

  constexpr int EPS = NT * 16 + 4;  // floats per channel row of a wave's transposition scratch (== 4 mod 32: conflict free)
  const float* bl = lds + a.b_off;
  const bf16x8* xl = reinterpret_cast<const bf16x8*>(lds16) + tapoff;
  const bf16x8* wl = reinterpret_cast<const bf16x8*>(lds16 + a.w_off16) + lane;

  // prologue: step 0 committed, step 1 in the registers
  PG_P_ISSUE_X()
  PG_P_ISSUE_W()
  PG_P_COMMIT_X(0, 0)
  if constexpr (XS > 1) { PG_P_COMMIT_X(XS - 1, 0) }
  PG_P_COMMIT_W(0)
  if (nsteps > 1) {
    PG_P_ISSUE_X()
    PG_P_ISSUE_W()
  }
  __syncthreads();
  PG_PROF_DECL
  int chunk = 0, tl = 0;
  for (int step = 0; step < nsteps; ++step) {
    const int cur = step & 1, nxt = cur ^ 1;
    const bool more = step + 1 < nsteps, more2 = step + 2 < nsteps;
    PG_PROF_MARK(5)
    if constexpr (WV == 8) {
      // 128 registers: the whole side work runs BEFORE the step's fragments are loaded (its ~44 temporaries and the 48
      // A-fragment registers never live together); the four waves of a SIMD drift apart by themselves, one wave's
      // side block under the others' MFMA blocks (tools/exp/coexec_ubench.hip, mode 3)
      if (more) {
        PG_P_COMMIT_X(0, nxt)
        PG_P_COMMIT_W(nxt)
      }
      if (more2) {
        PG_P_ISSUE_X()
        PG_P_ISSUE_W()
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const bf16x8* xb = xl + cur * xbuf16;
    if constexpr (WV == 8) {
      // B fragments of the wave's (<= 2) pixel groups stay resident, the A fragments come in two halves of two output
      // tiles: 24 + 24 fragment registers instead of 48 + 12 at the same number of LDS reads
      bf16x8 bf[NT][3];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        bf[n][0] = xb[pixoff[n]];
        bf[n][1] = xb[pixoff[n] + a.plane16];
        bf[n][2] = xb[pixoff[n] + 2 * a.plane16];
      }
#pragma unroll
      for (int mh = 0; mh < 2; ++mh) {
        bf16x8 ah[2][3];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) ah[m][pc] = wl[cur * B3P_W4 + ((2 * mh + m) * 3 + pc) * 64];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            f32x4 c = acc[2 * mh + m][n];
            c = MFMA16B(ah[m][2], bf[n][0], c);  // small terms first
            c = MFMA16B(ah[m][0], bf[n][2], c);
            c = MFMA16B(ah[m][1], bf[n][1], c);
            c = MFMA16B(ah[m][1], bf[n][0], c);
            c = MFMA16B(ah[m][0], bf[n][1], c);
            c = MFMA16B(ah[m][0], bf[n][0], c);
            acc[2 * mh + m][n] = c;
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    bf16x8 af[MT][3];
    if constexpr (WV == 4) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) af[m][pc] = wl[cur * B3P_W4 + (m * 3 + pc) * 64];
    }
#pragma unroll
    for (int n = 0; n < (WV == 8 ? 0 : 4); ++n) {
      bf16x8 bh, bm, bo;
      if (n < NT) {
        bh = xb[pixoff[n < NT ? n : 0]];
        bm = xb[pixoff[n < NT ? n : 0] + a.plane16];
        bo = xb[pixoff[n < NT ? n : 0] + 2 * a.plane16];
      }
      __builtin_amdgcn_sched_barrier(0);
      // side slice n: commit of step + 1 into the other buffers, issue of step + 2
      if constexpr (WV == 4) {
        if (n == 0) { if (more) PG_P_COMMIT_X(0, nxt) }
        if (n == 1) { if constexpr (XS > 1) { if (more) PG_P_COMMIT_X(XS - 1, nxt) } }
        if (n == 2) { if (more) PG_P_COMMIT_W(nxt) if (more2) PG_P_ISSUE_X() }
        if (n == 3) { if (more2) PG_P_ISSUE_W() }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (n < NT) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          f32x4 c = acc[m][n < NT ? n : 0];
          c = MFMA16B(af[m][2], bh, c);  // small terms first
          c = MFMA16B(af[m][0], bo, c);
          c = MFMA16B(af[m][1], bm, c);
          c = MFMA16B(af[m][1], bh, c);
          c = MFMA16B(af[m][0], bm, c);
          c = MFMA16B(af[m][0], bh, c);
          acc[m][n < NT ? n : 0] = c;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    PG_PROF_MARK(0)
    if (++chunk == nchunk) {
      chunk = 0;
      // ---- epilogue of the tile: v = out_act(acc + bias) * act'(dact_src) + res + res2, transposed through the wave's
      // own scratch so that a store covers 256 contiguous bytes of one channel row
      const int n_img = n_first + tl * nstep;
      ++tl;
      int Lv = L;
      asm volatile("" : "+s"(Lv));
      const size_t so = ((size_t)n_img * a.Cout + co0) * Lv;
      float* ep = lds + a.ep_off + wave * (16 * EPS);
      const int cvalid = a.Cout - co0;
      const bool fullc = cvalid >= 64;
      int cvalid_p = cvalid;
      asm volatile("" : "+s"(cvalid_p));
      float* outp = a.out + so;
      const float* st0 = a.dact_src ? a.dact_src + so : nullptr;
      const float* st1 = a.res ? a.res + (size_t)co0 * Lv + (size_t)n_img * a.res_bs : nullptr;
      const float* st2 = a.res2 ? a.res2 + (size_t)co0 * Lv + (size_t)n_img * a.res2_bs : nullptr;
      const bool any_op = st0 || st1 || st2;
      const int dsel = st0 ? a.dact : PG_ACT_NONE;
      constexpr int HQ = 4, NH = 16 / HQ;
      float o0[HQ], o1[HQ], o2[HQ];
#define PG_P_REQUEST(M, H)                                                                 \
  _Pragma("unroll") for (int c = 0; c < HQ; ++c) {                                         \
    const int cc = (M) * 16 + (H) * HQ + c;                                                \
    const size_t off_ = (size_t)((fullc || cc < cvalid_p) ? cc : 0) * Lv;                  \
    if (st0) o0[c] = (st0 + off_)[lane_px];                                                \
    if (st1) o1[c] = (st1 + off_)[lane_px];                                                \
    if (st2) o2[c] = (st2 + off_)[lane_px];                                                \
  }
      if constexpr (WV == 8) {
        // 32-pixel wave tiles: transposed through the wave's scratch like the 1x1 kernel does it — lanes 0-31 take channels
        // 0-7 of a 16-channel tile, lanes 32-63 channels 8-15, lane & 31 = pixel: every lane busy, a store covers two runs
        // of 128 contiguous bytes. (Storing straight from the accumulator layout — four 64-byte row segments per
        // instruction, no LDS round trip — measured SLOWER: 1499 against 983 cycles per step, tools/exp/b3_phase_prof.py.)
        const int half = lane >> 5;
        unsigned ox = opx32;   // opaque copies: the lane's row addresses are formed here, not kept across the step loop
        int hrow = 8 * half;
        asm volatile("" : "+v"(ox), "+v"(hrow));
        float q0[4], q1[4], q2[4];
#define PG_P_REQ8(M, H)                                                                    \
  _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                          \
    const int cc = (M) * 16 + hrow + (H) * 4 + c;                                          \
    const size_t off_ = (size_t)((fullc || cc < cvalid_p) ? cc : 0) * Lv;                  \
    if (st0) q0[c] = (st0 + off_)[ox];                                                     \
    if (st1) q1[c] = (st1 + off_)[ox];                                                     \
    if (st2) q2[c] = (st2 + off_)[ox];                                                     \
  }
        if (any_op) { PG_P_REQ8(0, 0) }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
          for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) ep[(kq * 4 + r) * EPS + n * 16 + (lane & 15)] = acc[m][n][r];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          float v[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) v[c] = ep[(8 * half + c) * EPS + (lane & 31)] + bl[m * 16 + 8 * half + c];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          switch (a.out_act) { /* wave-uniform */
            case PG_ACT_RELU:
#pragma unroll
              for (int c = 0; c < 8; ++c) v[c] = pg_apply_act(v[c], PG_ACT_RELU);
              break;
            case PG_ACT_ELU:
#pragma unroll
              for (int c = 0; c < 8; ++c) v[c] = pg_apply_act(v[c], PG_ACT_ELU);
              break;
            case PG_ACT_GELU:
              if constexpr (GL) {
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = pg_apply_act(v[c], PG_ACT_GELU);
              }
              break;
            default: break;
          }
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            if (any_op) {
              switch (dsel) {
                case PG_ACT_RELU:
#pragma unroll
                  for (int c = 0; c < 4; ++c) v[hh * 4 + c] *= pg_act_grad(q0[c], PG_ACT_RELU);
                  break;
                case PG_ACT_ELU:
#pragma unroll
                  for (int c = 0; c < 4; ++c) v[hh * 4 + c] *= pg_act_grad(q0[c], PG_ACT_ELU);
                  break;
                case PG_ACT_GELU:
                  if constexpr (GL) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[hh * 4 + c] *= pg_act_grad(q0[c], PG_ACT_GELU);
                  }
                  break;
                case PG_ACT_ELU_OUT:
#pragma unroll
                  for (int c = 0; c < 4; ++c) v[hh * 4 + c] *= pg_act_grad(q0[c], PG_ACT_ELU_OUT);
                  break;
                default: break;
              }
              if (st1) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[hh * 4 + c] += q1[c];
              }
              if (st2) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[hh * 4 + c] += q2[c];
              }
              // the next operands are requested BEFORE these stores (loads and stores retire in order)
              __builtin_amdgcn_sched_barrier(0);
              if (hh == 0) { PG_P_REQ8(m, 1) }
              else if (m + 1 < MT) { PG_P_REQ8(m + 1, 0) }
              __builtin_amdgcn_sched_barrier(0);
            }
            if (sok32) {
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const int cc = m * 16 + hrow + hh * 4 + c;
                if (fullc || cc < cvalid_p) (outp + (size_t)cc * Lv)[ox] = v[hh * 4 + c];
              }
            }
          }
        }
#undef PG_P_REQ8
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
      if (any_op) { PG_P_REQUEST(0, 0) }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) ep[(kq * 4 + r) * EPS + n * 16 + (lane & 15)] = acc[m][n][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = ep[c * EPS + lane] + bl[m * 16 + c];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        switch (a.out_act) { /* wave-uniform */
          case PG_ACT_RELU:
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] = pg_apply_act(v[c], PG_ACT_RELU);
            break;
          case PG_ACT_ELU:
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] = pg_apply_act(v[c], PG_ACT_ELU);
            break;
          case PG_ACT_GELU:
            if constexpr (GL) {
#pragma unroll
              for (int c = 0; c < 16; ++c) v[c] = pg_apply_act(v[c], PG_ACT_GELU);
            }
            break;
          default: break;
        }
        if (!any_op) {
          if (sok) {
            if (fullc) {
#pragma unroll
              for (int c = 0; c < 16; ++c) (outp + (size_t)(m * 16 + c) * Lv)[lane_px] = v[c];
            } else {
#pragma unroll
              for (int c = 0; c < 16; ++c) {
                const int cc = m * 16 + c;
                if (cc < cvalid_p) (outp + (size_t)cc * Lv)[lane_px] = v[c];
              }
            }
          }
        } else {
#pragma unroll
          for (int hh = 0; hh < NH; ++hh) {
            switch (dsel) {
              case PG_ACT_RELU:
#pragma unroll
                for (int c = 0; c < HQ; ++c) v[hh * HQ + c] *= pg_act_grad(o0[c], PG_ACT_RELU);
                break;
              case PG_ACT_ELU:
#pragma unroll
                for (int c = 0; c < HQ; ++c) v[hh * HQ + c] *= pg_act_grad(o0[c], PG_ACT_ELU);
                break;
              case PG_ACT_GELU:
                if constexpr (GL) {
#pragma unroll
                  for (int c = 0; c < HQ; ++c) v[hh * HQ + c] *= pg_act_grad(o0[c], PG_ACT_GELU);
                }
                break;
              case PG_ACT_ELU_OUT:
#pragma unroll
                for (int c = 0; c < HQ; ++c) v[hh * HQ + c] *= pg_act_grad(o0[c], PG_ACT_ELU_OUT);
                break;
              default: break;
            }
            if (st1) {
#pragma unroll
              for (int c = 0; c < HQ; ++c) v[hh * HQ + c] += o1[c];
            }
            if (st2) {
#pragma unroll
              for (int c = 0; c < HQ; ++c) v[hh * HQ + c] += o2[c];
            }
            // the next quarter's operands are requested BEFORE this quarter's stores (loads and stores retire in order)
            __builtin_amdgcn_sched_barrier(0);
            if (hh + 1 < NH) { PG_P_REQUEST(m, hh + 1) }
            else if (m + 1 < MT) { PG_P_REQUEST(m + 1, 0) }
            __builtin_amdgcn_sched_barrier(0);
            if (sok) {
              if (fullc) {
#pragma unroll
                for (int c = 0; c < HQ; ++c) (outp + (size_t)(m * 16 + hh * HQ + c) * Lv)[lane_px] = v[hh * HQ + c];
              } else {
#pragma unroll
                for (int c = 0; c < HQ; ++c) {
                  const int cc = m * 16 + hh * HQ + c;
                  if (cc < cvalid_p) (outp + (size_t)cc * Lv)[lane_px] = v[hh * HQ + c];
                }
              }
            }
          }
        }
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#undef PG_P_REQUEST
    }
    PG_PROF_MARK(4)
    if (more) __syncthreads();  // the other buffers are committed; every wave is done with the current ones
  }
  PG_PROF_DUMP(WV, wave, nsteps)
#undef PG_P_ISSUE_X
#undef PG_P_ISSUE_W
#undef PG_P_COMMIT_SLOT
#undef PG_P_COMMIT_X
#undef PG_P_COMMIT_W
}

// ---- 1x1 convolutions: no x tile in LDS ---------------------------------------------------------------
// With one tap a wave's B fragments are read by that wave only, so staging x through LDS buys nothing and
// costs the workgroup barriers that keep the waves of conv_b3_kernel in lockstep (loads, split and MFMA
// phases back to back). Here every wave is on its own: the weights of the output chunk (all channel chunks,
// <= 24 KB) go to LDS once per workgroup, then a wave walks its 32-pixel tiles with no further barrier —
// float2 loads straight into the B-fragment layout, activation + split in registers, the next chunk's loads
// in flight under the MFMAs, and the other waves of the SIMD filling the matrix pipe meanwhile (a 64 x 32
// accumulator tile: two waves per SIMD at 64 output channels, three / four at 32 / 16).
// Column j of pixel group n is pixel 2 j + n of the tile: lane (j, kq) then needs 2 CONSECUTIVE pixels of
// its 8 channels = one float2 per channel (16 lanes x 8 B = 128 contiguous bytes per channel and load).
// Same packed weights and epilogue (bias, activation, derivative of the fused input activation, up to two
// residual streams) as conv_b3_kernel.
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int PW_EPS = 36;   // floats per channel row of a wave's 16 x 32 transposition scratch
constexpr int PW_WAVES = 4;

template <bool GL, int MT, bool MS>
__global__ void __launch_bounds__(64 * PW_WAVES, MT == 1 ? 4 : ((MT == 2 || (MT == 4 && !MS && !GL)) ? 3 : 2)) conv_b3_pw_kernel(const B3Args a) {
  constexpr int AH = (MT == 4 && !MS && !GL) ? 2 : MT;  // (the GELU instantiation spills 6 registers at 168: it keeps two waves)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, jc = lane & 15;
  const int L = a.OH * a.OW;
  const int nchunk = a.Cin / a.CIB;
  const int co0 = blockIdx.y * B3_CO_CHUNK;
  const int tpi = (L + 31) >> 5;  // 32-pixel tiles per image, the last one may be ragged (L % 2 == 0)
  const int nitems = a.N * tpi;

  {  // weights of this output chunk, every channel chunk: [j][co tile][piece][lane] 16-byte fragments
    const float4* wsrc = reinterpret_cast<const float4*>(a.wfrag) + (size_t)blockIdx.y * nchunk * a.wslab4;
    float4* wdst = reinterpret_cast<float4*>(lds);
    for (int i = tid; i < nchunk * a.wslab4; i += 64 * PW_WAVES) wdst[i] = wsrc[i];
    if (tid < B3_CO_CHUNK) lds[a.b_off + tid] = (a.bias && co0 + tid < a.Cout) ? a.bias[co0 + tid] : 0.f;
  }
  __syncthreads();  // the only barrier of the kernel

  const int gw = blockIdx.x * PW_WAVES + wave, GW = gridDim.x * PW_WAVES;
  if (gw >= nitems) return;
  float* ep = lds + a.ep_off + wave * (16 * PW_EPS);
  const float* bl = lds + a.b_off;
  const bf16x8* wl = reinterpret_cast<const bf16x8*>(lds) + lane;
  const bool kact = kq < a.cgs;  // K groups beyond the chunk's channels are zero (weights packed as zero too)
  const unsigned lane_in = (unsigned)((kact ? 8 * kq : 0) * L + 2 * jc);  // floats from the (image, chunk, tile) base
  const bool has_res = a.res != nullptr, has_ds = a.dact_src != nullptr, has_res2 = a.res2 != nullptr;
  const int half = lane >> 5, px = lane & 31;  // store phase: lane = (8-channel half, pixel)

  f32x2 raw[8], nxt[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) nxt[c] = f32x2{0.f, 0.f};
// addresses = uniform base (SGPR pair: image, channel chunk, tile) + 32-bit lane offset: one VGPR per stream
#define PG_PW_ISSUE(IT, J, DST)                                                                        \
  {                                                                                                    \
    const int ni_ = (IT) / tpi;                                                                        \
    const int t0_ = ((IT) - ni_ * tpi) * 32;                                                           \
    const bool ok_ = kact && t0_ + 2 * jc < L;                                                         \
    /* every lane loads from a valid address (its own when ok_, the tile's first otherwise): no branches */ \
    const unsigned lo_ = ok_ ? lane_in : 0u;                                                           \
    const float* sb_ = a.in + ((size_t)ni_ * a.Cin + (J) * a.CIB) * (size_t)L + t0_;                   \
    _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                                    \
      const f32x2 t_ = *reinterpret_cast<const f32x2*>(sb_ + c * (size_t)L + lo_);                     \
      DST[c] = f32x2{ok_ ? t_[0] : 0.f, ok_ ? t_[1] : 0.f};                                            \
    }                                                                                                  \
  }
// AH = output tiles whose A fragments are resident at a time: all MT (one read per chunk), or — MT = 4 without the multi-stream
// epilogue, round 6 — two, re-read per pixel group (24 LDS reads per chunk instead of 12, 24 fragment registers instead of 48):
// that instantiation then fits 168 registers = THREE waves per SIMD instead of two (it is HBM bound: 3.0 TB/s of algorithmic
// traffic on PixelCNN's 32 -> 64 at batch 1024 with two)
#define PG_PW_MFMA(ACT)                                                                   \
  _Pragma("unroll") for (int n = 0; n < 2; ++n) {                                         \
    float e_[8];                                                                          \
    _Pragma("unroll") for (int c = 0; c < 8; ++c) e_[c] = pg_apply_act(raw[c][n], ACT);   \
    u32x4 h_, m_, l_;                                                                     \
    split8t(e_, h_, m_, l_);                                                              \
    const bf16x8 bh = __builtin_bit_cast(bf16x8, h_);                                     \
    const bf16x8 bm = __builtin_bit_cast(bf16x8, m_);                                     \
    const bf16x8 bo = __builtin_bit_cast(bf16x8, l_);                                     \
    _Pragma("unroll") for (int mh = 0; mh < MT / AH; ++mh) {                              \
      bf16x8 af[AH][3];                                                                   \
      _Pragma("unroll") for (int m = 0; m < AH; ++m)                                      \
        _Pragma("unroll") for (int pc = 0; pc < 3; ++pc) af[m][pc] = wl[((j * MT + mh * AH + m) * 3 + pc) * 64]; \
      _Pragma("unroll") for (int m = 0; m < AH; ++m) {                                    \
        f32x4 c = acc[mh * AH + m][n];                                                    \
        c = MFMA16B(af[m][2], bh, c);                                                     \
        c = MFMA16B(af[m][0], bo, c);                                                     \
        c = MFMA16B(af[m][1], bm, c);                                                     \
        c = MFMA16B(af[m][1], bh, c);                                                     \
        c = MFMA16B(af[m][0], bm, c);                                                     \
        c = MFMA16B(af[m][0], bh, c);                                                     \
        acc[mh * AH + m][n] = c;                                                          \
      }                                                                                   \
    }                                                                                     \
  }

  PG_PW_ISSUE(gw, 0, raw)
  for (int it = gw; it < nitems; it += GW) {
    f32x4 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < nchunk; ++j) {
      const bool lastj = j + 1 == nchunk;
      const int it2 = lastj ? it + GW : it, j2 = lastj ? 0 : j + 1;
      if (it2 < nitems) PG_PW_ISSUE(it2, j2, nxt)
      switch (a.in_act) { /* wave-uniform */
        case PG_ACT_RELU: PG_PW_MFMA(PG_ACT_RELU) break;
        case PG_ACT_ELU:  PG_PW_MFMA(PG_ACT_ELU) break;
        case PG_ACT_GELU: if constexpr (GL) { PG_PW_MFMA(PG_ACT_GELU) } break;
        default:          PG_PW_MFMA(PG_ACT_NONE) break;
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) raw[c] = nxt[c];
    }
    // ---- epilogue: v = out_act(acc + bias) (+ res | * act'(dact_src)), transposed through the wave's own
    // scratch: lanes 0-31 take channels 0-7 of a 16-channel tile, lanes 32-63 channels 8-15, lane & 31 =
    // pixel, so a store covers two runs of 128 contiguous bytes
    const int n_img = it / tpi;
    const int t0 = (it - n_img * tpi) * 32;
    const bool sok = t0 + px < L;
    int Lq = L;  // opaque copy: keeps the per-channel offsets (cc * L) of the epilogue from being hoisted into scalar registers
    asm volatile("" : "+s"(Lq));
    const size_t cstride = (size_t)Lq;
    const size_t so = ((size_t)n_img * a.Cout + co0) * cstride + t0;   // uniform
    const int cvalid = a.Cout - co0 - 8 * half;
    // lane part; lanes whose 8-channel half lies entirely beyond Cout (Cout % 64 in 1..8) address the chunk's first channel
    const unsigned lo = (sok && cvalid > 0) ? (unsigned)(8 * half * L + px) : 0u;
    float* outp = a.out + so;
    // up to three operand streams: derivative source, residual, second residual (a residual may be a
    // batch-strided channel slice: res_bs); v = out_act(acc + bias) * act'(o0) + o1 + o2
    const size_t so_c = (size_t)co0 * cstride + t0;
    const float* st0 = has_ds ? a.dact_src + so : nullptr;
    const float* st1 = has_res ? a.res + (size_t)n_img * a.res_bs + so_c : nullptr;
    const float* st2 = has_res2 ? a.res2 + (size_t)n_img * a.res2_bs + so_c : nullptr;
    const int dsel = has_ds ? a.dact : PG_ACT_NONE;
    // MS = false: at most one of the streams (one buffer); MS = true: its own instantiation (MT = 4), because
    // two more operand buffers spill in the narrower kernels
    constexpr int NB = MS ? 8 : 1;
    float o0[8], o1[NB], o2[NB];
    const float* sts = st0 ? st0 : st1;  // the single stream of the MS = false kernels
#define PG_PW_REQUEST(M)                                                       \
  _Pragma("unroll") for (int c = 0; c < 8; ++c) {                              \
    const int cc = (M) * 16 + c;                                               \
    const size_t off_ = (size_t)(cc < cvalid ? cc : 0) * cstride;              \
    if constexpr (MS) {                                                        \
      if (st0) o0[c] = (st0 + off_)[lo];                                       \
      if (st1) o1[c] = (st1 + off_)[lo];                                       \
      if (st2) o2[c] = (st2 + off_)[lo];                                       \
    } else {                                                                   \
      if (sts) o0[c] = (sts + off_)[lo];                                       \
    }                                                                          \
  }
    PG_PW_REQUEST(0)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<f32x2*>(ep + (kq * 4 + r) * PW_EPS + 2 * jc) = f32x2{acc[m][0][r], acc[m][1][r]};
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      float v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = ep[(8 * half + c) * PW_EPS + px] + bl[m * 16 + 8 * half + c];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      switch (a.out_act) { /* wave-uniform */
        case PG_ACT_RELU:
#pragma unroll
          for (int c = 0; c < 8; ++c) v[c] = pg_apply_act(v[c], PG_ACT_RELU);
          break;
        case PG_ACT_ELU:
#pragma unroll
          for (int c = 0; c < 8; ++c) v[c] = pg_apply_act(v[c], PG_ACT_ELU);
          break;
        case PG_ACT_GELU:
if constexpr (GL) {
#pragma unroll
          for (int c = 0; c < 8; ++c) v[c] = pg_apply_act(v[c], PG_ACT_GELU);
          }
          break;
        default: break;
      }
      switch (dsel) {
        case PG_ACT_RELU:
#pragma unroll
          for (int c = 0; c < 8; ++c) v[c] *= pg_act_grad(o0[c], PG_ACT_RELU);
          break;
        case PG_ACT_ELU:
#pragma unroll
          for (int c = 0; c < 8; ++c) v[c] *= pg_act_grad(o0[c], PG_ACT_ELU);
          break;
        case PG_ACT_GELU:
if constexpr (GL) {
#pragma unroll
          for (int c = 0; c < 8; ++c) v[c] *= pg_act_grad(o0[c], PG_ACT_GELU);
          }
          break;
        case PG_ACT_ELU_OUT:
#pragma unroll
          for (int c = 0; c < 8; ++c) v[c] *= pg_act_grad(o0[c], PG_ACT_ELU_OUT);
          break;
        default: break;
      }
      float v2[MS ? 8 : 1];
      if constexpr (MS) {
        if (a.out2) {  // dual data gradient (B3Args::out2): two outputs, the residual stream is r, not an addend
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const float d = v[c];
            v2[c] = d * pg_act_grad(o1[c], PG_ACT_ELU_OUT);
            v[c] = d * pg_act_grad(o0[c] - o1[c], PG_ACT_ELU_OUT);
          }
        } else {
        if (st1) {
#pragma unroll
          for (int c = 0; c < 8; ++c) v[c] = fmaf(o1[c], a.res_scale, v[c]);
        }
        if (st2) {
#pragma unroll
          for (int c = 0; c < 8; ++c) v[c] += o2[c];
        }
        }
      } else {
        if (st1) {
#pragma unroll
          for (int c = 0; c < 8; ++c) v[c] = fmaf(o0[c], a.res_scale, v[c]);
        }
      }
      // the next tile's operand is requested BEFORE this tile's stores (loads and stores retire in order)
      __builtin_amdgcn_sched_barrier(0);
      if (m + 1 < MT) PG_PW_REQUEST(m + 1)
      __builtin_amdgcn_sched_barrier(0);
      if (sok) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int cc = m * 16 + c;
          if (cc < cvalid) (outp + (size_t)cc * cstride)[lo] = v[c];
        }
        if constexpr (MS) {
          if (a.out2) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              const int cc = m * 16 + c;
              if (cc < cvalid) (a.out2 + so + (size_t)cc * cstride)[lo] = v2[c];
            }
          }
        }
      }
    }
#undef PG_PW_REQUEST
  }
#undef PG_PW_ISSUE
#undef PG_PW_MFMA
}

#include "conv_b3q_kernel.h"

template <bool GL, int MT, int CG = 1, bool MS = false, bool W9 = false, bool GT = false>
void b3_launch(const B3Args& a, int nt, dim3 grid, size_t shmem, hipStream_t st) {
  // the LDS opt-in is set once per instantiation by a function-local static initialiser: thread-safe
  // (the library is entered from the main thread and from the autograd thread)
#define PG_B3_L(NTV)                                                                                  \
  {                                                                                                   \
    static const hipError_t attr_##NTV = hipFuncSetAttribute(                                         \
        reinterpret_cast<const void*>(conv_b3_kernel<GL, MT, NTV, CG, MS, W9, GT>),                       \
        hipFuncAttributeMaxDynamicSharedMemorySize, CG == 1 ? 80 * 1024 : 160 * 1024);                \
    (void)attr_##NTV;                                                                                 \
    hipLaunchKernelGGL((conv_b3_kernel<GL, MT, NTV, CG, MS, W9, GT>), grid, dim3(B3_THREADS * CG), shmem, st, a); \
  }
  switch (nt) {
    case 1: PG_B3_L(1) break;
    case 2: PG_B3_L(2) break;
    case 3: PG_B3_L(3) break;
    default: PG_B3_L(4) break;
  }
#undef PG_B3_L
}

template <bool GL>
void b3p_launch(const B3Args& a, int nt, int waves, dim3 grid, size_t shmem, hipStream_t st) {
#define PG_B3P_L(NTV, WVV)                                                                            \
  {                                                                                                   \
    static const hipError_t attr_##NTV##_##WVV = hipFuncSetAttribute(                                 \
        reinterpret_cast<const void*>(conv_b3p_kernel<GL, NTV, WVV>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); \
    (void)attr_##NTV##_##WVV;                                                                         \
    hipLaunchKernelGGL((conv_b3p_kernel<GL, NTV, WVV>), grid, dim3(64 * WVV), shmem, st, a);          \
  }
  if (waves == 8) {
    if (nt == 1) PG_B3P_L(1, 8) else PG_B3P_L(2, 8)
    return;
  }
  switch (nt) {
    case 1: PG_B3P_L(1, 4) break;
    case 2: PG_B3P_L(2, 4) break;
    case 3: PG_B3P_L(3, 4) break;
    default: PG_B3P_L(4, 4) break;
  }
#undef PG_B3P_L
}

template <bool GL, int MT, bool MS = false>
void b3_pw_launch(const B3Args& a, dim3 grid, size_t shmem, hipStream_t st) {
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_b3_pw_kernel<GL, MT, MS>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  (void)attr;
  hipLaunchKernelGGL((conv_b3_pw_kernel<GL, MT, MS>), grid, dim3(64 * PW_WAVES), shmem, st, a);
}


template <bool GL>
void b3_dispatch(const B3Args& a, const B3Launch& l, hipStream_t st) {
  if (l.pw == 3) {  // the overlapped 16-wave kernel (conv_b3q_kernel.h): nt = staging slots per thread, CG = 1 for two tiles per workgroup
    b3q_launch<GL>(a, l.nt, l.CG != 0, l.grid, l.shmem, st);
    return;
  }
  if (l.pw == 2) {  // the pipelined 4-tap kernel
    b3p_launch<GL>(a, l.nt, l.CG, l.grid, l.shmem, st);  // (CG carries the waves per workgroup: 4 / 8)
    return;
  }
  if (l.pw) {
    if (l.ms) b3_pw_launch<GL, 4, true>(a, l.grid, l.shmem, st);
    else switch (l.MT) {
      case 1: b3_pw_launch<GL, 1>(a, l.grid, l.shmem, st); break;
      case 2: b3_pw_launch<GL, 2>(a, l.grid, l.shmem, st); break;
      case 3: b3_pw_launch<GL, 3>(a, l.grid, l.shmem, st); break;
      default: b3_pw_launch<GL, 4>(a, l.grid, l.shmem, st); break;
    }
    return;
  }
  if (l.CG == 2) {
    if (l.ms) b3_launch<GL, 4, 2, true>(a, l.nt, l.grid, l.shmem, st);
    else if (l.gate) {
      if constexpr (!GL) b3_launch<GL, 4, 2, false, false, true>(a, l.nt, l.grid, l.shmem, st);  // (no GELU variant: the host refuses)
    } else b3_launch<GL, 4, 2>(a, l.nt, l.grid, l.shmem, st);
    return;
  }
  if (l.w9) {
    if (l.ms) b3_launch<GL, 4, 1, true, true>(a, l.nt, l.grid, l.shmem, st);
    else b3_launch<GL, 4, 1, false, true>(a, l.nt, l.grid, l.shmem, st);
    return;
  }
  if (l.ms) {
    b3_launch<GL, 4, 1, true>(a, l.nt, l.grid, l.shmem, st);
    return;
  }
  switch (l.MT) {
    case 1: b3_launch<GL, 1>(a, l.nt, l.grid, l.shmem, st); break;
    case 2: b3_launch<GL, 2>(a, l.nt, l.grid, l.shmem, st); break;
    case 3: b3_launch<GL, 3>(a, l.nt, l.grid, l.shmem, st); break;
    default: b3_launch<GL, 4>(a, l.nt, l.grid, l.shmem, st); break;
  }
}

}  // namespace

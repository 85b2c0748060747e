// conv_direct.hip — stride-1 direct convolution over an explicit tap list, fp32, gfx950.
//
// Replaces torch.nn.Conv2d.forward as reached from the reference's hot path
// (CausalConv2d.forward nn/convolution.py:41-43; the pad+crop convs of
// gated_pixel_cnn.py:63-96 and pixel_snail.py:41-55; every 1x1 conv). The causal mask,
// padding and cropping are all expressed as a list of (dr, dc) tap offsets, so masked
// taps are skipped instead of multiplied by zero, and the data gradient is the same kernel
// run with negated taps and a transposed weight pack.
//
// CDNA4 mapping
//  * lane  = 4 consecutive output pixels x 16 output channels (64 fp32 accumulators).
//  * weights are wave-uniform: they are read through the SCALAR unit (s_load_dwordx16 from
//    the packed [cin][tap][cout] layout) and enter v_pk_fma_f32 as SGPR operands — no LDS
//    or VGPR traffic for weights at all.
//  * the input tile (+halo, zero filled, prologue activation applied once) is staged in
//    LDS for the spatial kernels; the 1x1 kernel streams float4 straight from HBM/L2.
#include "common.h"

namespace {

constexpr int COB = 16;  // output channels per block (one s_load_dwordx16 per (cin, tap))
constexpr int PX = 4;    // pixels per lane

struct TapArgs {
  const float* in;
  const float* wpk;
  const float* bias;
  const float* res;
  const float* dact_src;  // optional: out *= act'(dact_src) (data-gradient of a fused input activation)
  float* out;
  int N, Cin, IH, IW, Cout, OH, OW, T;
  int b_pad;
  int TR, Wg, tiles_per_img;
  int tile_h, tile_w, min_dr, min_dc, ch_stride, CIB;
  int in_act, dact;
  int swp_shift, rpi;  // staging: lanes per tile row = 1 << swp_shift, rows per wave iteration
  int stage_vec, qshift;  // float4 staging (IW % 4 == 0, aligned): lanes per row = 1 << qshift
  int pix_threads, CT;    // threads per cout-tile group, cout tiles per block
  int w_off;              // float offset of the staged-weight area inside LDS
  float inv_tile_h;
  int vec;  // OW%4==0 and 16B-aligned out/res: float4 epilogue
  int tapoff[PG_MAX_TAPS];
};

template <int ACT>
__global__ void __launch_bounds__(512) conv_taps_kernel(const TapArgs a) {
  extern __shared__ float lds[];
  // blockDim = pix_threads * CT: thread group g = tid / pix_threads works on cout tile
  // blockIdx.y*CT + g of the SAME staged input tile (staging is amortised over CT cout tiles)
  const int tid_all = threadIdx.x;
  const int grp = tid_all / a.pix_threads;
  const int tid = tid_all - grp * a.pix_threads;
  const int n = blockIdx.x / a.tiles_per_img;
  const int tile = blockIdx.x - n * a.tiles_per_img;
  const int row0 = tile * a.TR;
  const int co0 = (blockIdx.y * a.CT + grp) * COB;
  const int co0c = co0 < a.b_pad ? co0 : a.b_pad - COB;  // odd tile count: idle group re-reads a valid tile

  const int lr = tid / a.Wg;
  const int cg = tid - lr * a.Wg;
  const bool active = (lr < a.TR) && (row0 + lr < a.OH);
  const int lr_c = lr < a.TR ? lr : a.TR - 1;  // keep LDS reads of idle lanes in range
  const int lane_base = lr_c * a.tile_w + cg * PX;

  float acc[PX][COB];
#pragma unroll
  for (int i = 0; i < PX; ++i)
#pragma unroll
    for (int j = 0; j < COB; ++j) acc[i][j] = 0.f;

  const int wave = tid_all >> 6, lane = tid_all & 63, nwaves = blockDim.x >> 6;
  const float* in_n = a.in + (size_t)n * a.Cin * a.IH * a.IW;
  if (a.stage_vec) {  // vec4 staging only writes in-range elements: zero the tile once
    for (int i = tid_all; i < a.CIB * a.ch_stride; i += blockDim.x) lds[i] = 0.f;
  }
  float* wl = lds + a.w_off;  // staged weights (16-byte aligned)

  for (int ci0 = 0; ci0 < a.Cin; ci0 += a.CIB) {
    const int cib = min(a.CIB, a.Cin - ci0);
    __syncthreads();
    if (a.stage_vec)
      pg_stage_rows_vec4<ACT>(lds, a.CIB * a.ch_stride, a.ch_stride, a.tile_h, a.tile_w, a.inv_tile_h,
                              in_n + (size_t)ci0 * a.IH * a.IW, (size_t)a.IH * a.IW, a.IH, a.IW, cib,
                              row0 + a.min_dr, a.min_dc, a.qshift, wave, nwaves, lane);
    else
      pg_stage_rows<ACT>(lds, a.CIB * a.ch_stride, a.ch_stride, a.tile_h, a.tile_w, a.inv_tile_h,
                         in_n + (size_t)ci0 * a.IH * a.IW, (size_t)a.IH * a.IW, a.IH, a.IW, cib,
                         row0 + a.min_dr, a.min_dc, a.swp_shift, a.rpi, wave, nwaves, lane);
    // this chunk's weights for the block's CT cout tiles: [ci][tap][CT][16] next to the x tile.
    // (They used to be read through the scalar cache inside the loop: with Cin*T*16 floats per
    // cout tile the working set thrashes the 16 KB scalar cache and every (ci, tap) step waited
    // ~0.5 us for L2 — measured 4x off the FMA bound. LDS broadcast reads have ~100-cycle latency.)
    {
      const int per_ci = a.T * a.CT * COB;
      const float* wsrc = a.wpk + (size_t)ci0 * a.T * a.b_pad;
      for (int i = tid_all; i < cib * per_ci; i += blockDim.x) {
        const int ci = i / per_ci;
        const int rem = i - ci * per_ci;
        const int t = rem / (a.CT * COB);
        const int gj = rem - t * (a.CT * COB);  // g*16 + j
        int co = blockIdx.y * a.CT * COB + gj;
        co = co < a.b_pad ? co : a.b_pad - 1;   // idle group of an odd tile count: any valid column
        wl[i] = wsrc[((size_t)ci * a.T + t) * a.b_pad + co];
      }
    }
    __syncthreads();
    for (int ci = 0; ci < cib; ++ci) {
      const float* xl = lds + ci * a.ch_stride + lane_base;
      const float4* wrow = reinterpret_cast<const float4*>(wl + (ci * a.T * a.CT + grp) * COB);
#pragma unroll 4
      for (int t = 0; t < a.T; ++t) {
        const float4* wp = wrow + t * a.CT * (COB / 4);  // group-uniform address: LDS broadcast
        const int off = a.tapoff[t];
        const float x0 = xl[off], x1 = xl[off + 1], x2 = xl[off + 2], x3 = xl[off + 3];
#pragma unroll
        for (int j4 = 0; j4 < COB / 4; ++j4) {
          const float4 w4 = wp[j4];
          const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int j = 4 * j4 + jj;
            acc[0][j] = fmaf(wv[jj], x0, acc[0][j]);
            acc[1][j] = fmaf(wv[jj], x1, acc[1][j]);
            acc[2][j] = fmaf(wv[jj], x2, acc[2][j]);
            acc[3][j] = fmaf(wv[jj], x3, acc[3][j]);
          }
        }
      }
    }
  }

  if (!active || co0 >= a.Cout) return;
  const int r = row0 + lr;
  const int c0 = cg * PX;
  const bool vec = a.vec != 0;
#pragma unroll
  for (int j = 0; j < COB; ++j) {
    const int co = co0 + j;
    if (co >= a.Cout) break;
    const float b = a.bias ? a.bias[co] : 0.f;
    const size_t o = (((size_t)n * a.Cout + co) * a.OH + r) * a.OW + c0;
    if (vec) {
      float4 v = make_float4(acc[0][j] + b, acc[1][j] + b, acc[2][j] + b, acc[3][j] + b);
      if (a.res) {
        const float4 rv = *reinterpret_cast<const float4*>(a.res + o);
        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
      }
      if (a.dact_src) {
        const float4 sv = *reinterpret_cast<const float4*>(a.dact_src + o);
        v.x *= pg_act_grad(sv.x, a.dact); v.y *= pg_act_grad(sv.y, a.dact);
        v.z *= pg_act_grad(sv.z, a.dact); v.w *= pg_act_grad(sv.w, a.dact);
      }
      *reinterpret_cast<float4*>(a.out + o) = v;
    } else {
#pragma unroll
      for (int i = 0; i < PX; ++i) {
        if (c0 + i < a.OW) {
          float v = acc[i][j] + b;
          if (a.res) v += a.res[o + i];
          if (a.dact_src) v *= pg_act_grad(a.dact_src[o + i], a.dact);
          a.out[o + i] = v;
        }
      }
    }
  }
}

// ---- 1x1 (pointwise) path: Y[Cout, N*L] = W[Cout,Cin] X[Cin, N*L] + b --------------------
struct PwArgs {
  const float* in;
  const float* wpk;
  const float* bias;
  const float* res;
  const float* dact_src;
  float* out;
  int N, Cin, Cout, L, G;  // G = ceil(L/4) lane groups per image
  int b_pad, in_act, dact;
};

template <bool VEC, int ACT, int DACT>
__global__ void __launch_bounds__(256) conv_pw_kernel(const PwArgs a) {
  const long g = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)a.N * a.G;
  const bool active = g < total;
  const long gc = active ? g : total - 1;
  const int n = (int)(gc / a.G);
  const int p0 = (int)(gc - (long)n * a.G) * PX;
  const int co0 = blockIdx.y * COB;

  float acc[PX][COB];
#pragma unroll
  for (int i = 0; i < PX; ++i)
#pragma unroll
    for (int j = 0; j < COB; ++j) acc[i][j] = 0.f;

  const float* xp = a.in + (size_t)n * a.Cin * a.L + p0;
  const float* wp = a.wpk + co0;
#pragma unroll 8
  for (int ci = 0; ci < a.Cin; ++ci) {
    float x0, x1 = 0.f, x2 = 0.f, x3 = 0.f;
    if (VEC) {
      const float4 v = *reinterpret_cast<const float4*>(xp);
      x0 = v.x; x1 = v.y; x2 = v.z; x3 = v.w;
    } else {
      x0 = xp[0];
      if (p0 + 1 < a.L) x1 = xp[1];
      if (p0 + 2 < a.L) x2 = xp[2];
      if (p0 + 3 < a.L) x3 = xp[3];
    }
    if (ACT != PG_ACT_NONE) {
      x0 = pg_apply_act(x0, ACT); x1 = pg_apply_act(x1, ACT);
      x2 = pg_apply_act(x2, ACT); x3 = pg_apply_act(x3, ACT);
    }
#pragma unroll
    for (int j = 0; j < COB; ++j) {
      const float w = wp[j];  // wave-uniform -> s_load_dwordx16
      acc[0][j] = fmaf(w, x0, acc[0][j]);
      acc[1][j] = fmaf(w, x1, acc[1][j]);
      acc[2][j] = fmaf(w, x2, acc[2][j]);
      acc[3][j] = fmaf(w, x3, acc[3][j]);
    }
    xp += a.L;
    wp += a.b_pad;
  }
  if (!active) return;
#pragma unroll
  for (int j = 0; j < COB; ++j) {
    const int co = co0 + j;
    if (co >= a.Cout) break;
    const float b = a.bias ? a.bias[co] : 0.f;
    const size_t o = ((size_t)n * a.Cout + co) * a.L + p0;
    if (VEC) {
      float4 v = make_float4(acc[0][j] + b, acc[1][j] + b, acc[2][j] + b, acc[3][j] + b);
      if (a.res) {
        const float4 rv = *reinterpret_cast<const float4*>(a.res + o);
        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
      }
      if (DACT != PG_ACT_NONE) {
        const float4 sv = *reinterpret_cast<const float4*>(a.dact_src + o);
        v.x *= pg_act_grad(sv.x, DACT); v.y *= pg_act_grad(sv.y, DACT);
        v.z *= pg_act_grad(sv.z, DACT); v.w *= pg_act_grad(sv.w, DACT);
      }
      *reinterpret_cast<float4*>(a.out + o) = v;
    } else {
#pragma unroll
      for (int i = 0; i < PX; ++i) {
        if (p0 + i < a.L) {
          float v = acc[i][j] + b;
          if (a.res) v += a.res[o + i];
          if (DACT != PG_ACT_NONE) v *= pg_act_grad(a.dact_src[o + i], DACT);
          a.out[o + i] = v;
        }
      }
    }
  }
}

template <bool VEC>
int launch_pw(const PwArgs& a, dim3 grid, hipStream_t st) {
#define PG_PW(ACT, DACT) hipLaunchKernelGGL((conv_pw_kernel<VEC, ACT, DACT>), grid, dim3(256), 0, st, a)
  if (a.dact == PG_ACT_NONE) {
    switch (a.in_act) {
      case PG_ACT_NONE: PG_PW(PG_ACT_NONE, PG_ACT_NONE); break;
      case PG_ACT_RELU: PG_PW(PG_ACT_RELU, PG_ACT_NONE); break;
      case PG_ACT_ELU:  PG_PW(PG_ACT_ELU, PG_ACT_NONE); break;
      case PG_ACT_GELU: PG_PW(PG_ACT_GELU, PG_ACT_NONE); break;
      default: return PG_EINVAL;
    }
  } else {
    if (a.in_act != PG_ACT_NONE) return PG_EINVAL;  // the fused act' epilogue is a dgrad feature
    switch (a.dact) {
      case PG_ACT_RELU: PG_PW(PG_ACT_NONE, PG_ACT_RELU); break;
      case PG_ACT_ELU:  PG_PW(PG_ACT_NONE, PG_ACT_ELU); break;
      case PG_ACT_GELU: PG_PW(PG_ACT_NONE, PG_ACT_GELU); break;
      default: return PG_EINVAL;
    }
  }
#undef PG_PW
  return 0;
}

// ---- weight packing ---------------------------------------------------------------------
struct PackArgs {
  const float* w;
  float* wpk;
  int Cout, Cin, KH, KW, T, transpose, A, B, b_pad;
  int tap_u[PG_MAX_TAPS];
  int tap_v[PG_MAX_TAPS];
};

__global__ void pack_weight_kernel(const PackArgs p) {
  const long total = (long)p.A * p.T * p.b_pad;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i % p.b_pad);
    const long at = i / p.b_pad;
    const int t = (int)(at % p.T);
    const int aa = (int)(at / p.T);
    float v = 0.f;
    if (b < p.B) {
      const int co = p.transpose ? aa : b;
      const int ci = p.transpose ? b : aa;
      v = p.w[(((size_t)co * p.Cin + ci) * p.KH + p.tap_u[t]) * p.KW + p.tap_v[t]];
    }
    p.wpk[i] = v;
  }
}

}  // namespace

PG_EXPORT int pg_conv_b_pad(int b) { return ((b + COB - 1) / COB) * COB; }

PG_EXPORT size_t pg_packed_weight_floats(int a, int T, int b) {
  return (size_t)a * T * pg_conv_b_pad(b);
}

PG_EXPORT int pg_pack_conv_weight(const float* w, float* wpk, int Cout, int Cin, int KH, int KW,
                                  int T, const int* tap_u, const int* tap_v, int transpose,
                                  int b_pad, void* stream) {
  PG_REQUIRE(w && wpk && tap_u && tap_v, PG_EINVAL, "pg_pack_conv_weight: null pointer");
  PG_REQUIRE(T >= 1 && T <= PG_MAX_TAPS, PG_ESHAPE, "pg_pack_conv_weight: T=%d not in [1,%d]", T,
             PG_MAX_TAPS);
  PackArgs p;
  p.w = w; p.wpk = wpk; p.Cout = Cout; p.Cin = Cin; p.KH = KH; p.KW = KW; p.T = T;
  p.transpose = transpose;
  p.A = transpose ? Cout : Cin;
  p.B = transpose ? Cin : Cout;
  PG_REQUIRE(b_pad == pg_conv_b_pad(p.B), PG_EINVAL, "pg_pack_conv_weight: b_pad=%d, expected %d",
             b_pad, pg_conv_b_pad(p.B));
  p.b_pad = b_pad;
  for (int t = 0; t < T; ++t) {
    PG_REQUIRE(tap_u[t] >= 0 && tap_u[t] < KH && tap_v[t] >= 0 && tap_v[t] < KW, PG_EINVAL,
               "pg_pack_conv_weight: tap %d (%d,%d) outside %dx%d", t, tap_u[t], tap_v[t], KH, KW);
    p.tap_u[t] = tap_u[t];
    p.tap_v[t] = tap_v[t];
  }
  const long total = (long)p.A * T * b_pad;
  const int blocks = (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  PG_LAUNCH_CHECK("pg_pack_conv_weight");
  return 0;
}

PG_EXPORT int pg_conv2d_taps(const float* in, const float* wpk, const float* bias,
                             const float* res, float* out, int N, int Cin, int IH, int IW,
                             int Cout, int OH, int OW, int T, const int* tap_dr,
                             const int* tap_dc, int in_act, const float* dact_src, int dact,
                             void* stream) {
  PG_REQUIRE(in && wpk && out && tap_dr && tap_dc, PG_EINVAL, "pg_conv2d_taps: null pointer");
  PG_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0, PG_EINVAL,
             "pg_conv2d_taps: non-positive dimension");
  PG_REQUIRE(T >= 1 && T <= PG_MAX_TAPS, PG_ESHAPE, "pg_conv2d_taps: T=%d not in [1,%d]", T,
             PG_MAX_TAPS);
  PG_REQUIRE(in_act >= PG_ACT_NONE && in_act <= PG_ACT_GELU, PG_EINVAL,
             "pg_conv2d_taps: bad in_act %d", in_act);
  PG_REQUIRE(dact >= PG_ACT_NONE && dact <= PG_ACT_GELU && ((dact == PG_ACT_NONE) == (dact_src == nullptr)),
             PG_EINVAL, "pg_conv2d_taps: dact_src / dact mismatch");
  hipStream_t st = (hipStream_t)stream;
  const int b_pad = pg_conv_b_pad(Cout);

  if (T == 1 && tap_dr[0] == 0 && tap_dc[0] == 0 && IH == OH && IW == OW) {
    PwArgs a;
    a.in = in; a.wpk = wpk; a.bias = bias; a.res = res; a.out = out; a.dact_src = dact_src;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.L = OH * OW; a.G = (a.L + PX - 1) / PX;
    a.b_pad = b_pad; a.in_act = in_act; a.dact = dact;
    const long groups = (long)N * a.G;
    dim3 grid((unsigned)((groups + 255) / 256), (unsigned)(b_pad / COB));
    const bool vec = (a.L % 4 == 0) && (((uintptr_t)in & 15) == 0) &&
                     (((uintptr_t)out & 15) == 0) && (!res || ((uintptr_t)res & 15) == 0) &&
                     (!dact_src || ((uintptr_t)dact_src & 15) == 0);
    int rc = vec ? launch_pw<true>(a, grid, st) : launch_pw<false>(a, grid, st);
    if (rc) { pg_set_error("pg_conv2d_taps: unsupported in_act/dact combination"); return rc; }
    PG_LAUNCH_CHECK("pg_conv2d_taps(1x1)");
    return 0;
  }

  TapArgs a;
  a.in = in; a.wpk = wpk; a.bias = bias; a.res = res; a.out = out; a.dact_src = dact_src;
  a.N = N; a.Cin = Cin; a.IH = IH; a.IW = IW; a.Cout = Cout; a.OH = OH; a.OW = OW; a.T = T;
  a.b_pad = b_pad; a.in_act = in_act; a.dact = dact;
  a.vec = ((OW % 4) == 0) && (((uintptr_t)out & 15) == 0) && (!res || ((uintptr_t)res & 15) == 0) &&
          (!dact_src || ((uintptr_t)dact_src & 15) == 0);
  int min_dr = tap_dr[0], max_dr = tap_dr[0], min_dc = tap_dc[0], max_dc = tap_dc[0];
  for (int t = 1; t < T; ++t) {
    min_dr = tap_dr[t] < min_dr ? tap_dr[t] : min_dr;
    max_dr = tap_dr[t] > max_dr ? tap_dr[t] : max_dr;
    min_dc = tap_dc[t] < min_dc ? tap_dc[t] : min_dc;
    max_dc = tap_dc[t] > max_dc ? tap_dc[t] : max_dc;
  }
  a.Wg = (OW + PX - 1) / PX;
  PG_REQUIRE(a.Wg <= 256, PG_ESHAPE, "pg_conv2d_taps: OW=%d too wide", OW);
  int TR = 256 / a.Wg;
  if (TR > OH) TR = OH;
  a.TR = TR;
  a.tiles_per_img = (OH + TR - 1) / TR;
  a.tile_h = TR + (max_dr - min_dr);
  a.tile_w = a.Wg * PX + (max_dc - min_dc);
  a.min_dr = min_dr; a.min_dc = min_dc;
  a.ch_stride = a.tile_h * a.tile_w;
  {
    int shift = 0;
    while ((1 << shift) < a.tile_w && shift < 6) ++shift;
    a.swp_shift = shift;
    a.rpi = 64 >> shift;
    a.inv_tile_h = 1.0f / (float)a.tile_h;
  }
  int CIB = (48 * 1024 / 4) / a.ch_stride;
  PG_REQUIRE(CIB >= 1, PG_ESHAPE, "pg_conv2d_taps: tile %dx%d does not fit LDS", a.tile_h,
             a.tile_w);
  if (CIB > Cin) CIB = Cin;
  a.CIB = CIB;
  for (int t = 0; t < T; ++t) a.tapoff[t] = (tap_dr[t] - min_dr) * a.tile_w + (tap_dc[t] - min_dc);
  int threads = ((TR * a.Wg + 63) / 64) * 64;
  a.pix_threads = threads;
  a.CT = (b_pad / COB >= 2 && threads <= 256) ? 2 : 1;  // two cout tiles share one staged tile
  a.stage_vec = ((IW % 4) == 0) && (((uintptr_t)in & 15) == 0) && (IW / 4 <= 64);
  {
    int qs = 0;
    while ((1 << qs) < IW / 4 && qs < 6) ++qs;
    a.qshift = qs;
  }
  const int ytiles = b_pad / COB;
  dim3 grid((unsigned)(N * a.tiles_per_img), (unsigned)((ytiles + a.CT - 1) / a.CT));
  threads *= a.CT;
  a.w_off = ((CIB * a.ch_stride + 4 + 3) / 4) * 4;  // x tile + dump word, rounded to 16 bytes
  // x tile + the chunk's weights must fit the 64 KB a kernel gets without opting in to more: shrink
  // the channel chunk (small images make CIB == Cin and the weight slab CIB*T*CT*16 floats unbounded)
  size_t shmem = ((size_t)a.w_off + (size_t)CIB * T * a.CT * COB) * sizeof(float);
  while (shmem > 64 * 1024 && CIB > 1) {
    CIB = (CIB + 1) / 2;
    a.CIB = CIB;
    a.w_off = ((CIB * a.ch_stride + 4 + 3) / 4) * 4;
    shmem = ((size_t)a.w_off + (size_t)CIB * T * a.CT * COB) * sizeof(float);
  }
  PG_REQUIRE(shmem <= 64 * 1024, PG_ESHAPE,
             "pg_conv2d_taps: a %dx%d tile with %d taps x %d output channels needs %zu B of LDS (> 64 KB)",
             a.tile_h, a.tile_w, T, a.CT * COB, shmem);
  switch (in_act) {
    case PG_ACT_RELU: hipLaunchKernelGGL(conv_taps_kernel<PG_ACT_RELU>, grid, dim3(threads), shmem, st, a); break;
    case PG_ACT_ELU:  hipLaunchKernelGGL(conv_taps_kernel<PG_ACT_ELU>, grid, dim3(threads), shmem, st, a); break;
    case PG_ACT_GELU: hipLaunchKernelGGL(conv_taps_kernel<PG_ACT_GELU>, grid, dim3(threads), shmem, st, a); break;
    default:          hipLaunchKernelGGL(conv_taps_kernel<PG_ACT_NONE>, grid, dim3(threads), shmem, st, a); break;
  }
  PG_LAUNCH_CHECK("pg_conv2d_taps");
  return 0;
}

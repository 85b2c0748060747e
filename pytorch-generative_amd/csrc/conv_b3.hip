// conv_b3.hip — the implicit-GEMM convolution of conv_mfma.hip with its fp32 products evaluated on the
// bf16 matrix pipe ("bf16x3"): x = xh + xm + xl, w = wh + wm + wl with bf16 pieces (8 + 8 + 8
// significand bits: the split is exact to 2^-24 |x|), and
//     w.x  =  wh.xh + wh.xm + wm.xh + wh.xl + wl.xh + wm.xm          (6 x v_mfma_f32_16x16x32_bf16)
// dropping wm.xl + wl.xm + wl.xl <= 3 * 2^-24 |w.x| — the size of one fp32 rounding; every
// bf16 x bf16 product is exact in the fp32 accumulator. Same reference call sites as conv_mfma.hip
// (nn/convolution.py:41-43, gated_pixel_cnn.py:63-96, pixel_snail.py:41-55, the 1x1 convolutions and
// aten::convolution_backward's data gradient); results agree with the fp32-MFMA kernel to ~1e-7.
//
// Why: measured on MI355X, v_mfma_f32_16x16x4_f32 costs 33 cycles per 1024 MACs and BLOCKS the VALU
// of its SIMD, v_mfma_f32_16x16x32_bf16 costs 17.7 cycles per 8192 MACs and co-executes with VALU
// work: six of them per 16x16x32 tile = 106 cycles against 264 for eight fp32 MFMAs, and the
// staging arithmetic (activation, splitting) hides under them.
//
// GEMM view: M = output channels, N = output pixels; a K step of 32 = four "groups" of 8 input
// channels at one tap. Lane (i|j = lane & 15, kq = lane >> 4) holds 8 consecutive K values
// (A: 8 channels of output channel i; B: 8 channels of pixel j) of group 4 ks + kq.
// LDS: x pieces as planes [channel group of 8][piece][tile pixel] of 16-byte entries, plane stride a
// multiple of 256 B — a B fragment read (ds_read_b128) of 16 consecutive pixels is conflict free for
// every tap offset, and a staging lane (= pixel) writes one entry per piece; weights as ready-made A
// fragments [k step][co tile][piece][lane] (pg_pack_conv_weight_frag*).
// Workgroups are persistent like conv_mfma's: (tile, channel chunk) steps with the next step's
// loads in flight under the MFMA loop.
#include "conv_b3_kernels.h"

// conv_b3_gelu.hip: the same launch on the instantiations that carry the GELU paths
void pg_b3_dispatch_gelu(const B3Args& a, const B3Launch& l, hipStream_t st);

namespace {

// ---- weight pack: A fragments of the three bf16 pieces ------------------------------------------
// wfrag (16-byte units) [co chunk][channel chunk][k step][co tile m][piece][lane]: the 8 bf16 of
//   Wsel[64 chunk + 16 m + (lane & 15)][channel = CIB * j + 8 cg(g) + 0..7][tap t(g)], g = 4 ks + (lane >> 4)
// with group g -> (t = g / cgs, cg = g % cgs); zero for g >= groups or an output channel >= M.
struct B3PackJob {
  unsigned int* wfrag;
  int transpose, M, Kc, MT, chunks, CIB, cgs, groups, ksteps, nchunk;
  int gate_order;  // PG_CONV_FMT_B3_GATE (M % 128 == 0): m-tile m of chunk c holds channels (M / 2) (m >> 1) + 32 c + 16 (m & 1) + ..,
                   // i.e. each chunk = 32 gate channels' [a | b] halves (conv_b3_kernel<.., GT = true> gates inside one wave)
  long total;  // lanes (pieces inside)
};

struct B3PackArgs {
  const float* w;
  int Cout, Cin, KH, KW, T, njobs;
  B3PackJob job[2];  // forward and data-gradient orientation in ONE launch
  int tap_u[PG_MAX_TAPS];
  int tap_v[PG_MAX_TAPS];
};

__global__ void b3_pack_kernel(const B3PackArgs a) {
  const long total = a.job[0].total + (a.njobs > 1 ? a.job[1].total : 0);
  for (long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x; i0 < total; i0 += (long)gridDim.x * blockDim.x) {
    const bool second = i0 >= a.job[0].total;
    const B3PackJob& p = a.job[second ? 1 : 0];
    const long i = second ? i0 - a.job[0].total : i0;
    const int lane = (int)(i & 63);
    long rest = i >> 6;
    const int m = (int)(rest % p.MT); rest /= p.MT;
    const int ks = (int)(rest % p.ksteps); rest /= p.ksteps;
    const int j = (int)(rest % p.nchunk);
    const int chunk = (int)(rest / p.nchunk);
    const int o = p.gate_order ? (m >> 1) * (p.M >> 1) + chunk * 32 + (m & 1) * 16 + (lane & 15)
                               : chunk * B3_CO_CHUNK + m * 16 + (lane & 15);
    const int g = 4 * ks + (lane >> 4);
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = 0.f;
    if (o < p.M && g < p.groups) {
      const int t = g / p.cgs, cg = g - t * p.cgs;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = p.CIB * j + 8 * cg + e;
        const int co = p.transpose ? c : o;
        const int ci = p.transpose ? o : c;
        x[e] = a.w[(((size_t)co * a.Cin + ci) * a.KH + a.tap_u[t]) * a.KW + a.tap_v[t]];
      }
    }
    u32x4 h, mm, l;
    split8(x, h, mm, l);
    // [chunk][j][ks][m][piece][lane] in 16-byte units
    const size_t base = ((((size_t)chunk * p.nchunk + j) * p.ksteps + ks) * p.MT + m) * 3 * 64 + lane;
    u32x4* dst = reinterpret_cast<u32x4*>(p.wfrag);
    dst[base] = h;
    dst[base + 64] = mm;
    dst[base + 128] = l;
  }
}

inline int b3_mt(int M) { return M >= B3_CO_CHUNK ? 4 : (M + 15) / 16; }
inline int b3_chunks(int M) { return (M + B3_CO_CHUNK - 1) / B3_CO_CHUNK; }

struct B3Plan {
  int ok, CIB, cgs, groups, ksteps, MT;
  size_t w_bytes;
  int w9;  // the plan needs the 9-weight-slot instantiation (conv_b3_kernel<.., W9 = true>)
  int px_cap;     // staged tile pixels (with halo) this plan's LDS budget assumes: 0 = the default for T (B3_PX_CAP), or the smaller
                  // tile (336) that lets a larger channel chunk fit (round 5)
  int pipelined;  // set ONLY by the PG_CONV_B3P branch of b3_plan: conv_b3p_kernel takes the launch (the generic planner can
                  // arrive at the same chunk shape — Cin % 16 != 0 with >= 128 output channels — and stays on conv_b3_kernel)
};

constexpr int B3_PX_CAP = 352;  // tile pixels (with halo) the LDS plan assumes for T > 1 (T == 1: 256)

// Format-level plan, a function of (K channels, M channels, taps) only, so that the weight pack does
// not depend on the image size: channel chunk CIB in {32, 16, 8}, groups of 8 channels x taps per
// chunk, K steps of 4 groups. ok = 0: the fp32-MFMA kernel takes the problem.
B3Plan b3_plan(int Kc, int M, int T) {
  B3Plan best = {};
  static const bool on = []() { const char* e = PG_AB_ENV("PG_CONV_B3"); return !(e && e[0] == '0'); }();
  if (!on || Kc % 8 != 0 || Kc < 16 || M < 16 || T < 1) return best;
  const int MT = b3_mt(M);
  const int px = T == 1 ? 256 : B3_PX_CAP;
  // 4 taps with >= 64 output channels: one 8-channel group x 4 taps = exactly one K step of 32 per chunk — the format
  // of the pipelined kernel (conv_b3p_kernel: double-buffered 14.6 KB x tile + 12 KB weight slab). PG_CONV_B3P=0 keeps
  // the 16-channel chunks of conv_b3_kernel (A/B).
  static const bool p_on = []() { const char* e = PG_AB_ENV("PG_CONV_B3P"); return !(e && e[0] == '0'); }();
  // (one output chunk only: with 128+ output channels the wide conv_b3_kernel, whose two chunks share ONE staged x tile,
  // measured faster — 247 against 282 us on the 2x2 64 -> 128 at N = 512 — while 64 -> 64 runs 119 -> 108 us here)
  // Round 5: also several output chunks when their number is odd-sized for the wide kernel (M % 128 != 0, e.g. PixelCNN++'s 160 / 320
  // filters): conv_b3_kernel then runs one chunk per workgroup as well, so both stage x once per chunk. PG_CONV_B3P_MULTI=0 for A/B.
  static const bool pm_on = []() { const char* e = PG_AB_ENV("PG_CONV_B3P_MULTI"); return !(e && e[0] == '0'); }();
  if (p_on && on && T == 4 && MT == 4 && (M <= B3_CO_CHUNK || (pm_on && M % (2 * B3_CO_CHUNK) != 0)) && Kc % 8 == 0 && Kc >= 16) {
    B3Plan pp = {1, 8, 1, 4, 1, 4, (size_t)MT * 3 * 1024, 0, 0, 1};
    return pp;
  }
  double best_cost = 1e30;
  // Round 5: further passes with smaller staged tiles (336, then 256 pixels instead of 352) — a plan is taken from them only when it
  // lets a LARGER channel chunk fit LDS (the cost below prefers fuller K steps, then fewer steps, then the larger tile):
  //   336: the 9-tap convolutions with 32 output channels (VD-VAE's / beta-VAE's 3x3 32 -> 32 and 64 -> 32) run 16-channel chunks (18
  //        of 20 K-step slots used, 2 chunk steps per 32 channels) instead of 8-channel ones (9 of 12, 4 steps): VD-VAE 4.77 -> 4.89 k;
  //   256: the 6-tap convolutions with >= 64 output channels (PixelCNN++'s 2x3, 31 % of its step) run 16-channel chunks (12 of 12
  //        slots) instead of 8-channel ones (6 of 8) on 192-pixel tiles: PixelCNN++ 669 -> 707 images/s. Nothing else changes plan
  //        (PG_CONV_B3_CAPS=352 in the ab library restores the round-4 plans). The wide kernel's real budget (x shared by two chunks in
  //        160 KB), which would give GatedPixelCNN's 1x3 32-channel chunks, measured +0.4 %: within noise, not taken.
  struct Caps { int n; int v[4]; };
  static const Caps caps = []() {
    Caps c = {3, {B3_PX_CAP, 336, 256, 0}};
    if (const char* e = PG_AB_ENV("PG_CONV_B3_CAPS")) {  // A/B: "352" = round-4 plans
      int a = 0, b = 0, d = 0, f = 0;
      const int n = sscanf(e, "%d,%d,%d,%d", &a, &b, &d, &f);
      if (n >= 1 && a == B3_PX_CAP) { c.n = n; c.v[0] = a; c.v[1] = b; c.v[2] = d; c.v[3] = f; }
    }
    return c;
  }();
  for (int pass = 0; pass < (T > 1 ? caps.n : 1); ++pass)
  for (int CIB = 32; CIB >= 8; CIB >>= 1) {
    if (Kc % CIB != 0) continue;
    const int pxp = pass == 0 ? px : caps.v[pass];
    if (pxp < 64) continue;
    const int cgs = CIB / 8, groups = cgs * T, ksteps = (groups + 3) / 4;
    if (ksteps > 5 || groups > B3_MAXG) continue;
    const size_t xb = (size_t)cgs * 3 * pxp * 16, wb = (size_t)ksteps * MT * 3 * 1024;
    const size_t wb_s = wb;
    const size_t scratch = (size_t)4 * 16 * 68 * 4;  // per-wave epilogue scratch of one chunk's four waves
    if (xb + wb_s + scratch + 1024 > 80 * 1024) continue;
    if ((long)cgs * pxp > (long)B3_XS * B3_THREADS) continue;
    // (round 4 and before: the slab went through 6 float4 registers per thread, so ksteps * MT * 192 <= 6 * 256 bounded the plan. The
    // slab is moved by LDS-DMA now — no such bound; PG_CONV_B3_WSLOTS=1 in the ab library restores it for A/B)
    static const bool wslots = []() { const char* e = PG_AB_ENV("PG_CONV_B3_WSLOTS"); return e && e[0] == '1'; }();
    if (wslots && (long)ksteps * MT * 192 > (long)B3_WS * B3_THREADS) continue;
    const double cost = (double)ksteps / CIB + 0.002 / CIB + (pass ? 1e-4 : 0.0);  // MFMA work per channel, then fewer steps, then the full tile
    if (cost < best_cost) {
      best_cost = cost;
      best = {1, CIB, cgs, groups, ksteps, MT, wb_s, 0, pass ? pxp : 0, 0};
    }
  }
  static const bool w9_on = []() { const char* e = PG_AB_ENV("PG_CONV_B3_W9"); return !(e && e[0] == '0'); }();
  if (!best.ok && w9_on && MT == 4 && Kc % 8 == 0) {
    // nothing fits 6 weight slots per thread: one 8-channel group per chunk with 9 slots (3x3, >= 64 output channels)
    const int cgs = 1, groups = T, ksteps = (groups + 3) / 4;
    const size_t xb = (size_t)cgs * 3 * px * 16, wb = (size_t)ksteps * MT * 3 * 1024;
    if (ksteps <= 5 && groups <= B3_MAXG && xb + wb + (size_t)4 * 16 * 68 * 4 + 1024 <= 80 * 1024 &&
        (long)cgs * px <= 2L * B3_THREADS && (long)ksteps * MT * 192 <= 9L * B3_THREADS)
      best = {1, 8, cgs, groups, ksteps, MT, wb, 1, 0, 0};
  }
  return best;
}

// rows per tile: as many as fit 256 output pixels and a staged tile (with halo) of at most `cap` pixels; cap = 0: the
// format-level assumption (B3_PX_CAP). pg_b3_conv passes the cap its launch's real LDS budget allows (round 4): the
// plan is made for 352 pixels whatever the image, but a plan with one channel group leaves room for more — on 64-wide
// rows a 3x3 then takes 4 output rows per tile (6 staged: 396 pixels) instead of 3 (5 staged), and the tile is a full
// 256 pixels instead of 192.
int b3_rows(int T, int OH, int OW, int hr, int hc, int cap = 0) {
  if (cap <= 0) cap = T == 1 ? 256 : B3_PX_CAP;
  int TR = 256 / OW;
  if (TR > OH) TR = OH;
  while (TR >= 1 && (TR + hr) * (OW + hc) > cap) --TR;
  if (TR < 1) return 0;
  const int nt_rows = (OH + TR - 1) / TR;
  return (OH + nt_rows - 1) / nt_rows;
}

void tap_extent(int T, const int* dr, const int* dc, int& min_dr, int& hr, int& min_dc, int& hc) {
  int a0 = dr[0], a1 = dr[0], b0 = dc[0], b1 = dc[0];
  for (int t = 1; t < T; ++t) {
    a0 = dr[t] < a0 ? dr[t] : a0; a1 = dr[t] > a1 ? dr[t] : a1;
    b0 = dc[t] < b0 ? dc[t] : b0; b1 = dc[t] > b1 ? dc[t] : b1;
  }
  min_dr = a0; hr = a1 - a0; min_dc = b0; hc = b1 - b0;
}

}  // namespace

#ifdef PG_ABLATE
// ablation builds only (not in include/pg_hip.h): destination of conv_b3_kernel's per-wave phase clocks
static long long* g_b3_prof = nullptr;
PG_EXPORT void pg_b3_set_prof(void* p) { g_b3_prof = (long long*)p; }
#endif

// ---- interface used by conv_mfma.hip's exported entry points --------------------------------------
int pg_b3_applicable(int Kc, int M, int T, int OH, int OW, int hr, int hc) {
  // Round 5: images down to 16 pixels (4 x 4) run here as well — a bf16x3 tile lives inside one image, so an 8 x 8 image fills a
  // quarter of the 256-pixel tile, and the six bf16 MFMAs per product still beat the fp32-MFMA kernel's multi-image tiles:
  // PixelCNN++ (its 8 x 8 level: 320-channel 2x3 / 2x2 convolutions) 606 -> 659 images/s, VD-VAE +2.7 %, beta-VAE +1.6 %
  // (same box, ab library; PG_CONV_B3_MIN_PX=256 there restores the round-4 routing)
  static const int min_px = []() { const char* e = PG_AB_ENV("PG_CONV_B3_MIN_PX"); const int v = e ? atoi(e) : 16; return v >= 16 ? v : 16; }();
  if (OW > 256 || OH * OW < min_px) return 0;
  const B3Plan pl = b3_plan(Kc, M, T);
  return pl.ok && b3_rows(T, OH, OW, hr, hc, pl.px_cap) >= 1;
}

size_t pg_b3_frag_floats(int Kc, int M, int T) {
  const B3Plan pl = b3_plan(Kc, M, T);
  if (!pl.ok) return 0;
  return (size_t)b3_chunks(M) * (Kc / pl.CIB) * pl.ksteps * pl.MT * 3 * 64 * 4;
}

// the 1x1 kernel's multi-stream instantiation takes the launch (the conditions of pg_b3_conv's routing, restated)
int pg_b3_dual_ok(int Cin, int Cout, int OH, int OW) {
  const B3Plan pl = b3_plan(Cin, Cout, 1);
  if (!pl.ok || pl.MT != 4 || pl.ksteps != 1 || (OH * OW) % 2 != 0 || OH * OW < 16 || OW > 256) return 0;
  return (size_t)(Cin / pl.CIB) * pl.MT * 3 * 1024 <= 24 * 1024;
}

int pg_b3_gate_fusable(int Cin, int Cout, int T, int OH, int OW, int hr, int hc) {
  if (Cout % (2 * B3_CO_CHUNK) != 0 || OW > 256 || OH * OW < 16) return 0;
  const B3Plan pl = b3_plan(Cin, Cout, T);
  // (one tap whose weights fit the 1x1 kernel runs without an x tile: pg_b3_conv's routing, restated)
  if (T == 1 && pl.ok && pl.ksteps == 1 && (size_t)(Cin / pl.CIB) * pl.MT * 3 * 1024 <= 24 * 1024) return 0;
  return pl.ok && !pl.pipelined && !pl.w9 && pl.MT == 4 && b3_rows(T, OH, OW, hr, hc, pl.px_cap) >= 1;
}

static int b3_pack_job(B3PackJob& p, float* wfrag, int Cout, int Cin, int T, int transpose, int gate_order = 0) {
  p.wfrag = reinterpret_cast<unsigned int*>(wfrag);
  p.transpose = transpose;
  p.gate_order = gate_order;
  PG_REQUIRE(!gate_order || (!transpose && Cout % (2 * B3_CO_CHUNK) == 0), PG_ESHAPE,
             "pg_pack_conv_weight_frag: the gate-interleaved order is for forward fragments with a multiple of 128 output channels");
  p.M = transpose ? Cin : Cout;
  p.Kc = transpose ? Cout : Cin;
  const B3Plan pl = b3_plan(p.Kc, p.M, T);
  PG_REQUIRE(pl.ok, PG_ESHAPE, "pg_pack_conv_weight_frag: shape not covered by the bf16x3 format");
  p.MT = pl.MT; p.chunks = b3_chunks(p.M); p.CIB = pl.CIB; p.cgs = pl.cgs; p.groups = pl.groups;
  p.ksteps = pl.ksteps; p.nchunk = p.Kc / pl.CIB;
  p.total = (long)p.chunks * p.nchunk * p.ksteps * p.MT * 64;
  return 0;
}

// either destination may be null (then only the other orientation is packed)
int pg_b3_pack2(const float* w, float* wfrag_fwd, float* wfrag_dgrad, int Cout, int Cin, int KH, int KW,
                int T, const int* tap_u, const int* tap_v, hipStream_t st, int gate_order) {
  B3PackArgs a;
  a.w = w; a.Cout = Cout; a.Cin = Cin; a.KH = KH; a.KW = KW; a.T = T; a.njobs = 0;
  if (wfrag_fwd) {
    const int rc = b3_pack_job(a.job[a.njobs++], wfrag_fwd, Cout, Cin, T, 0, gate_order);
    if (rc) return rc;
  }
  if (wfrag_dgrad) {
    const int rc = b3_pack_job(a.job[a.njobs++], wfrag_dgrad, Cout, Cin, T, 1);
    if (rc) return rc;
  }
  if (a.njobs == 0) return 0;
  if (a.njobs == 1) a.job[1] = a.job[0];
  for (int t = 0; t < T; ++t) { a.tap_u[t] = tap_u[t]; a.tap_v[t] = tap_v[t]; }
  const long total = a.job[0].total + (a.njobs > 1 ? a.job[1].total : 0);
  const int blocks = (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
  hipLaunchKernelGGL(b3_pack_kernel, dim3(blocks), dim3(256), 0, st, a);
  PG_LAUNCH_CHECK("pg_pack_conv_weight_frag(bf16x3)");
  return 0;
}

// gate != 0 (1 + PG_GATE_*; round 6): the launch must be the wide kernel (Cout % 128 == 0, at most one dense residual) — it then also writes
// gate_out = gate_res + act(a) * sigmoid(b) for [a | b] = out (+ res); pg_b3_gate_fusable() says beforehand whether a shape qualifies
int pg_b3_gate_fusable(int Cin, int Cout, int T, int OH, int OW, int hr, int hc);
int pg_b3_conv(const float* in, const float* wfrag, const float* bias, const float* res, float* out,
               int N, int Cin, int IH, int IW, int Cout, int OH, int OW, int T, const int* tap_dr,
               const int* tap_dc, int in_act, const float* dact_src, int dact, int out_act,
               const float* res2, long res_bs, long res2_bs, hipStream_t st, int gate, const float* gate_res,
               float* gate_out, float* dual_out2) {
  B3Args a;
  a.gate = gate; a.gate_res = gate_res; a.gate_out = gate_out;
  a.out2 = dual_out2;
  PG_REQUIRE(!dual_out2 || (res && dact_src && !res2 && !gate && out_act == PG_ACT_NONE && !bias && T == 1), PG_EINVAL,
             "pg_conv2d_mfma_dual: a 1x1 data gradient with a derivative source and r, nothing else");
  PG_REQUIRE(gate == 0 || (gate_out && !res2 && !dact_src && out_act == PG_ACT_NONE &&
                           (gate == 1 + PG_GATE_TANH || gate == 1 + PG_GATE_IDENTITY)), PG_EINVAL,
             "pg_conv2d_mfma_gate: the fused gate takes one residual of the convolution, no second one / derivative / output activation");
  a.in = in; a.wfrag = wfrag; a.bias = bias; a.res = res; a.dact_src = dact_src; a.out = out;
  PG_REQUIRE(res2 == nullptr || res != nullptr, PG_EINVAL, "pg_conv2d_mfma(bf16x3): res2 without res");
  // the weight slabs are moved by LDS-DMA in 16-byte units (conv_b3_kernel, round 5)
  PG_REQUIRE((((uintptr_t)wfrag) & 15) == 0, PG_EINVAL, "pg_conv2d_mfma(bf16x3): the fragment buffer must be 16-byte aligned");
  a.res2 = res2;
  a.res_scale = 1.f;
  a.res_bs = res_bs > 0 ? res_bs : (long)Cout * OH * OW;
  a.res2_bs = res2_bs > 0 ? res2_bs : (long)Cout * OH * OW;
  a.N = N; a.Cin = Cin; a.IH = IH; a.IW = IW; a.Cout = Cout; a.OH = OH; a.OW = OW; a.T = T;
  a.in_act = in_act; a.dact = dact; a.out_act = out_act;
  // the instantiations with the GELU paths live in conv_b3_gelu.hip (conv_b3_kernels.h says why)
  const bool gelu = in_act == PG_ACT_GELU || out_act == PG_ACT_GELU || dact == PG_ACT_GELU;
#ifdef PG_ABLATE
  { const char* e = getenv("PG_B3_DBG"); a.dbg = e ? atoi(e) : 0; }
  a.prof = g_b3_prof;
#else
  a.dbg = 0;
#endif
  int hr, hc;
  tap_extent(T, tap_dr, tap_dc, a.min_dr, hr, a.min_dc, hc);
  const B3Plan pl = b3_plan(Cin, Cout, T);
  int TR = b3_rows(T, OH, OW, hr, hc, pl.px_cap);
  PG_REQUIRE(pl.ok && TR >= 1 && OH * OW >= 16 && OW <= 256, PG_ESHAPE,
             "pg_conv2d_mfma(bf16x3): shape not covered");
  a.CIB = pl.CIB; a.cgs = pl.cgs; a.groups = pl.groups; a.ksteps = pl.ksteps;
  a.wslab4 = pl.ksteps * pl.MT * 192;
  // multi-stream epilogue (two residuals, residual + derivative, batch-strided residual): own instantiations
  const bool ms = res2 != nullptr || (res != nullptr && dact_src != nullptr) ||
                  (res != nullptr && a.res_bs != (long)Cout * OH * OW);
  {
    // 1x1 without an LDS x tile (conv_b3_pw_kernel) up to 64 input channels (weight slab of an output chunk <= 24 KB).
    // Measured forward, old -> new kernel (tools/exp/pw_ab.py): 64 -> 64 at N = 512, 32x32: 76.8 -> 65.4 us (4.1 TB/s of
    // algorithmic traffic), with ELU on both sides 83.0 -> 74.8; 64 -> 32 at N = 1024, 28x28: 84.1 -> 64.7; 32 -> 64:
    // 99.4 -> 88.7. 128 input channels are compute-heavy enough for the staged kernel: 182 -> 197 us, left there.
    static const bool pw_on = []() { const char* e = PG_AB_ENV("PG_CONV_B3_PW"); return !(e && e[0] == '0'); }();
    const int nchunk = Cin / pl.CIB;
    const size_t wbytes = (size_t)nchunk * pl.MT * 3 * 1024;
    // both residual operands the SAME tensor (PixelCNN's x + (x + net(x)) and the two equal skip gradients of its
    // backward): one stream, added twice
    const bool twice = res2 != nullptr && res2 == res && a.res2_bs == a.res_bs;
    const bool ms_pw = twice ? ((dact_src != nullptr) || a.res_bs != (long)Cout * OH * OW) : ms;  // (dual: res + dact_src => ms)
    if (pw_on && !gate && (!ms_pw || pl.MT == 4) && T == 1 && pl.ksteps == 1 && tap_dr[0] == 0 && tap_dc[0] == 0 && IH == OH && IW == OW &&
        (OH * OW) % 2 == 0 && (((uintptr_t)in) & 7) == 0 && wbytes <= 24 * 1024) {
      a.TR = 0; a.tile_h = a.tile_w = a.plane16 = a.tiles_per_img = 0;
      a.xslots = 0; a.dump16 = 0; a.w_off16 = 0;
      for (int g = 0; g < B3_MAXG; ++g) { a.g_tapoff[g] = 0; a.g_cg[g] = 0; }
      a.ep_off = (int)(wbytes / 4);
      a.b_off = a.ep_off + PW_WAVES * 16 * PW_EPS;
      const size_t shmem = ((size_t)a.b_off + B3_CO_CHUNK + 4) * sizeof(float);
      const int chunks_y = b3_chunks(Cout);
      const long items = (long)N * ((OH * OW + 31) / 32);
      // resident workgroups per CU: what the registers allow (waves per SIMD = 4 / 3 / 2 for 1 / 2 / 3-4 output
      // tiles of 16 channels), LDS permitting
      // (round 6: four output tiles WITHOUT the multi-stream epilogue hold their A fragments in halves: 167 registers, three too)
      const int by_regs = pl.MT == 1 ? 4 : ((pl.MT == 2 || (pl.MT == 4 && !ms_pw && !gelu)) ? 3 : 2);
      const int by_lds = (int)((160 * 1024) / shmem);
      const int per_cu = by_lds < by_regs ? by_lds : by_regs;
      long gx = (long)256 * per_cu / chunks_y;
      if (gx > (items + PW_WAVES - 1) / PW_WAVES) gx = (items + PW_WAVES - 1) / PW_WAVES;
      if (gx < 1) gx = 1;
      const dim3 grid((unsigned)gx, (unsigned)chunks_y);
      if (twice) { a.res2 = nullptr; a.res_scale = 2.f; }
      const B3Launch l = {1, pl.MT, 0, 1, ms_pw ? 1 : 0, 0, 0, grid, shmem};
      if (gelu) pg_b3_dispatch_gelu(a, l, st);
      else b3_dispatch<false>(a, l, st);
      PG_LAUNCH_CHECK("pg_conv2d_mfma(bf16x3, 1x1)");
      return 0;
    }
    PG_REQUIRE(!dual_out2, PG_ESHAPE, "pg_conv2d_mfma_dual: shape not on the 1x1 kernel (pg_conv_dual_ok)");
  }
  if (T > 1 && !pl.pipelined) {
    // conv_b3_kernel: the staged tile may be larger than the format-level 352 pixels if THIS launch's LDS has the room
    // (one 64-channel chunk per workgroup assumed here; the wide kernel is re-checked below and falls back)
    static const bool big_on = []() { const char* e = PG_AB_ENV("PG_CONV_B3_BIGTILE"); return !(e && e[0] == '0'); }();
    const long fixed = (long)pl.w_bytes + 4L * 16 * 68 * 4 + (B3_CO_CHUNK + B3_MAXG + 4) * 4 + 16 * 16 + 512;
    long cap = (80L * 1024 - fixed) / ((long)pl.cgs * 48);        // 16-byte entries per (group, piece) plane
    const long slot_cap = (pl.w9 ? 2L : (long)B3_XS) * B3_THREADS / pl.cgs;  // staging slots per thread
    if (cap > slot_cap) cap = slot_cap;
    cap = (cap / 16) * 16;
    const bool wide = !pl.w9 && pl.MT == 4 && Cout % (2 * B3_CO_CHUNK) == 0;  // (its LDS holds two weight slabs: keep the plan's tile)
    if (big_on && !wide && cap > (pl.px_cap ? pl.px_cap : B3_PX_CAP)) {
      const int TR2 = b3_rows(T, OH, OW, hr, hc, (int)cap);
      if (TR2 > TR) TR = TR2;
    }
  }
  a.TR = TR; a.tile_h = TR + hr; a.tile_w = OW + hc;
  a.plane16 = ((a.tile_h * a.tile_w + 15) / 16) * 16;
  a.tiles_per_img = (OH + TR - 1) / TR;
  PG_REQUIRE(!gate || (!pl.pipelined && !pl.w9 && pl.MT == 4 && Cout % (2 * B3_CO_CHUNK) == 0 && !gelu), PG_ESHAPE,
             "pg_conv2d_mfma_gate: the fused gate needs the wide bf16x3 kernel with a multiple of 128 output channels");
  // (Round 6, measured and NOT shipped: a row-ring forward kernel with register-resident weights for the 64-channel 4-tap layers —
  // tools/exp/rejected/conv_b3r_kernel.h (its README has the routing block that stood here) — parity green, 225 against 235 us on the
  // 2x2 64 -> 64 at batch 1024, PixelSNAIL +0.3 %: profiles/r06_conv_ring_forward_ab.txt.)
  if (pl.pipelined && a.tile_h * a.tile_w <= B3P_PX) {
    // ---- the pipelined kernel: LDS = x[2][3 pieces][plane16] | dump entry | w[2][768] | 4 x epilogue scratch | bias, taps
    for (int g = 0; g < B3_MAXG; ++g) { a.g_tapoff[g] = 0; a.g_cg[g] = 0; }
    for (int t = 0; t < 4; ++t) a.g_tapoff[t] = (tap_dr[t] - a.min_dr) * a.tile_w + (tap_dc[t] - a.min_dc);
    a.xslots = a.tile_h * a.tile_w;
    const size_t x16 = (size_t)2 * 3 * a.plane16;
    a.dump16 = (int)x16;
    a.w_off16 = (int)(((x16 + 1 + 15) / 16) * 16);
    // waves per workgroup: 8 (a wave owns 32 pixels, 128 registers, four waves per SIMD) unless PG_CONV_B3P_WAVES=4
    static const int env_waves = []() { const char* e = PG_AB_ENV("PG_CONV_B3P_WAVES"); return (e && atoi(e) == 4) ? 4 : 8; }();
    const int p_waves = env_waves;
    const int px = TR * OW;
    const int nt = p_waves == 8 ? (px + 127) / 128 : (px + 63) / 64;   // 16-pixel groups per wave
    size_t shmem = ((size_t)a.w_off16 + 2 * B3P_W4) * 16;
    a.ep_off = (int)(shmem / 4);
    shmem += (size_t)p_waves * 16 * (nt * 16 + 4) * 4;  // per-wave transposition scratch
    a.b_off = (int)(shmem / 4);
    shmem += (B3_CO_CHUNK + 8) * sizeof(float);
    PG_REQUIRE(shmem <= (size_t)80 * 1024, PG_ESHAPE, "pg_conv2d_mfma(bf16x3, pipelined): %zu B of LDS", shmem);
    const int chunks_y = b3_chunks(Cout);
    long want = 512 / chunks_y;  // resident workgroups: 2 per CU
    if (want < a.tiles_per_img) want = a.tiles_per_img;
    long gx = (want / a.tiles_per_img) * a.tiles_per_img;
    if (gx > (long)N * a.tiles_per_img) gx = (long)N * a.tiles_per_img;
    const dim3 grid((unsigned)gx, (unsigned)chunks_y);
    const B3Launch l = {2, 4, nt, p_waves, 0, 0, 0, grid, shmem};
    if (gelu) pg_b3_dispatch_gelu(a, l, st);
    else b3_dispatch<false>(a, l, st);
    PG_LAUNCH_CHECK("pg_conv2d_mfma(bf16x3, pipelined)");
    return 0;
  }
  {
    // ---- the overlapped 16-wave kernel (conv_b3q_kernel.h, round 6): >= 64 output channels, any tap count. Its LDS (one
    // workgroup per CU, up to 160 KB) = x tiles [NXT][2 buffers][cgs][3 pieces][plane16] | dump entry | slab ring [2][NCH][768] |
    // bias, K-group table; two output chunks share one x tile when Cout % 128 == 0, otherwise two tiles share one slab.
    static const bool q_on = []() { const char* e = PG_AB_ENV("PG_CONV_B3Q"); return !(e && e[0] == '0'); }();
    // Measured on MI355X (round 6, tools/exp/q_ab.py, same box, median of 5 x 20 launches, forward; profiles/r06_conv_q_ab.txt):
    //   two tiles per workgroup (ST): PixelCNN++'s 2x3 160 -> 320 265 -> 243 us, 320 -> 160 301 -> 283 us at batch 64 — taken;
    //     2x3 160 -> 160 152 -> 160 us and the 16 x 16 level (one 256-pixel tile per image: 96-160 workgroups for 256 CUs) 97 -> 147 us — not;
    //   two output chunks per workgroup (CG, GatedPixelCNN / PixelSNAIL 64 -> 128): 3-6 % SLOWER than the 8-wave wide kernel on every
    //     shape (1x1 256 -> 256 446 -> 463 us, 2x1 755 -> 792, 1x2 426 -> 439, 1x3 641 -> 675, 2x2 64 -> 128 392 -> 416): sixteen waves that
    //     meet at one barrier per K step run their phases in lockstep, and the kernel's ablation (profiles/r06_conv_q_ablation.txt) shows the
    //     phases ADD UP (MFMA 200 + loads / commit 130 + epilogue 95 + slab DMA 28 + empty K-step loop 100 of 564 us) instead of overlapping:
    //     the epilogue alone streams at 5.6 TB/s, i.e. at the HBM write rate, while no wave computes. Not taken (PG_CONV_B3Q_CG=1 in the ab
    //     library runs it for A/B).
    static const bool q_cg = []() { const char* e = PG_AB_ENV("PG_CONV_B3Q_CG"); return e && e[0] == '1'; }();
    const bool stm = Cout % (2 * B3_CO_CHUNK) != 0;
    const int NCH = stm ? 1 : 2, NXT = stm ? 2 : 1, GT = stm ? 512 : 1024;
    const long tail = (long)(NCH * B3_CO_CHUNK + B3_MAXG + 4) * 4 + 16 * 16 + 256;
    long cap = (160L * 1024 - 2L * NCH * B3Q_SLAB16 * 16 - tail) / ((long)NXT * 2 * pl.cgs * 48);
    cap = (cap / 16) * 16;
    const bool q_shape = !gate && (stm ? (T >= 6 && (Cin >= 256 || Cout >= 256)) : q_cg);
    const int TRq = (q_on && q_shape && pl.MT == 4 && !pl.pipelined && cap >= 64) ? b3_rows(T, OH, OW, hr, hc, (int)cap) : 0;
    if (TRq >= 1) {
      const int th = TRq + hr, tw = OW + hc;
      const int xs = (pl.cgs * th * tw + GT - 1) / GT;
      // at least 6 of the 8 pixel slices of 32 must be busy (small images stay on the 4-wave kernels), and in ST mode the
      // batch must provide pairs of images
      if (xs <= 2 && TRq * OW >= 192 && (!stm || N >= 2) && (OH + TRq - 1) / TRq >= 2) {
        a.TR = TRq; a.tile_h = th; a.tile_w = tw;
        a.plane16 = ((th * tw + 15) / 16) * 16;
        a.tiles_per_img = (OH + TRq - 1) / TRq;
        for (int g = 0; g < pl.groups; ++g) {
          const int t = g / pl.cgs, cg = g - t * pl.cgs;
          a.g_tapoff[g] = (tap_dr[t] - a.min_dr) * tw + (tap_dc[t] - a.min_dc);
          a.g_cg[g] = cg;
        }
        for (int g = pl.groups; g < B3_MAXG; ++g) { a.g_tapoff[g] = 0; a.g_cg[g] = 0; }
        a.xslots = pl.cgs * th * tw;
        const size_t x16 = (size_t)NXT * 2 * pl.cgs * 3 * a.plane16;
        a.dump16 = (int)x16;
        a.w_off16 = (int)(((x16 + 1 + 15) / 16) * 16);
        size_t shmem = ((size_t)a.w_off16 + 2 * NCH * B3Q_SLAB16) * 16;
        a.ep_off = 0;
        a.b_off = (int)(shmem / 4);
        shmem += (size_t)(NCH * B3_CO_CHUNK + B3_MAXG + 4) * sizeof(float);
        PG_REQUIRE(shmem <= (size_t)160 * 1024, PG_ESHAPE, "pg_conv2d_mfma(bf16x3, overlapped): %zu B of LDS", shmem);
        const int chunks_y = stm ? b3_chunks(Cout) : Cout / (2 * B3_CO_CHUNK);
        long want = 256 / chunks_y;  // resident workgroups: 1 per CU
        if (want < a.tiles_per_img) want = a.tiles_per_img;
        long gx = (want / a.tiles_per_img) * a.tiles_per_img;
        const long units = (stm ? (long)(N + 1) / 2 : (long)N) * a.tiles_per_img;  // ST: a workgroup takes images in pairs
        if (gx > units) gx = units;
        const dim3 grid((unsigned)gx, (unsigned)chunks_y);
        const B3Launch l = {3, 4, xs, stm ? 1 : 0, 0, 0, 0, grid, shmem};
        if (gelu) pg_b3_dispatch_gelu(a, l, st);
        else b3_dispatch<false>(a, l, st);
        PG_LAUNCH_CHECK("pg_conv2d_mfma(bf16x3, overlapped)");
        return 0;
      }
    }
  }
  for (int g = 0; g < pl.groups; ++g) {
    const int t = g / pl.cgs, cg = g - t * pl.cgs;
    a.g_tapoff[g] = (tap_dr[t] - a.min_dr) * a.tile_w + (tap_dc[t] - a.min_dc);
    a.g_cg[g] = cg;
  }
  for (int g = pl.groups; g < B3_MAXG; ++g) { a.g_tapoff[g] = 0; a.g_cg[g] = 0; }
  a.xslots = pl.cgs * a.tile_h * a.tile_w;
  const size_t x16 = (size_t)pl.cgs * 3 * a.plane16;
  a.dump16 = (int)x16;              // one spare 16-byte entry (+ padding to a 256-byte boundary)
  a.w_off16 = (int)(((x16 + 1 + 15) / 16) * 16);
  // wide workgroups (two output chunks share one staged x tile): default; PG_CONV_B3_WIDE=0 for A/B
  static const bool wide_on = []() { const char* e = PG_AB_ENV("PG_CONV_B3_WIDE"); return !(e && e[0] == '0'); }();
  const int CG = ((wide_on || gate) && !pl.w9 && pl.MT == 4 && Cout % (2 * B3_CO_CHUNK) == 0) ? 2 : 1;
  size_t shmem = (size_t)a.w_off16 * 16 + pl.w_bytes * CG;
  a.ep_off = (int)(shmem / 4);
  shmem += (size_t)4 * CG * 16 * 68 * 4;
  a.b_off = (int)(shmem / 4);
  shmem += (B3_CO_CHUNK * CG + B3_MAXG + 4) * sizeof(float);
  PG_REQUIRE(shmem <= (size_t)(CG == 1 ? 80 : 160) * 1024, PG_ESHAPE, "pg_conv2d_mfma(bf16x3): %zu B of LDS", shmem);
  const int nt = (TR * OW + 63) / 64;
  const int chunks_y = b3_chunks(Cout) / CG;
  long want = (CG == 1 ? 512 : 256) / chunks_y;  // resident workgroups: 2 per CU (4 waves) / 1 per CU (8 waves)
  if (want < a.tiles_per_img) want = a.tiles_per_img;
  long gx = (want / a.tiles_per_img) * a.tiles_per_img;
  if (gx > (long)N * a.tiles_per_img) gx = (long)N * a.tiles_per_img;
  dim3 grid((unsigned)gx, (unsigned)chunks_y);
  PG_REQUIRE(!ms || pl.MT == 4, PG_ESHAPE,
             "pg_conv2d_mfma_ex(bf16x3): the multi-stream epilogue is instantiated for >= 64 output channels");
  PG_REQUIRE(!gate || (CG == 2 && !ms), PG_ESHAPE, "pg_conv2d_mfma_gate: not the wide kernel's plain epilogue");
  const B3Launch l = {0, pl.MT, nt, CG, ms ? 1 : 0, pl.w9, gate ? 1 : 0, grid, shmem};
  if (gelu) pg_b3_dispatch_gelu(a, l, st);
  else b3_dispatch<false>(a, l, st);
  PG_LAUNCH_CHECK("pg_conv2d_mfma(bf16x3)");
  return 0;
}

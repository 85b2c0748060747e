// conv_b3r_kernel.h — row-ring convolution with REGISTER-RESIDENT weights (round 6): the 4-tap 64 -> 64 convolutions and data
// gradients of PixelSNAIL's ResidualBlock (pixel_snail.py:41-55) on 32-pixel rows. Included by conv_b3_kernels.h.
//
// conv_b3p_kernel stages an x tile with a row halo per (tile, 8-channel chunk) step and streams a weight slab per step; its K loop
// issues ~4 VALU instructions per MFMA (profiles/r06_snail_conv_pmc.json). With 64 input channels x 4 taps the WHOLE contraction is
// K = 256 = 8 K steps: the A fragments of one 16-channel output tile are 8 x 3 pieces x 4 registers = 96 registers — they fit a
// wave's register file for the entire launch. So: a workgroup = 4 waves = the four output tiles of a 64-channel chunk; it walks
// CONSECUTIVE rows of an image segment (the walk of conv_wgrad_b3r_kernel): per step ONE new input row is staged (64 channels x 32
// pixels, three-way split, into an LDS ring of 2 rows with a zero column on either side) and one output row is produced — 96 MFMAs per
// wave against one staging slot (8 values) and 8 output values per thread. No weight traffic after the prologue, no halo
// re-staging, one barrier pair per row. 3 workgroups per CU (<= 168 registers, 26 KB LDS each).
//
// Fragment format: the pipelined plan's (b3_plan: CIB = 8, one K step per 8-channel chunk j; lane group kg = tap kg).
// Epilogue: v = out_act(acc + bias) * act'(dact_src) + res + res2, straight from the accumulator layout (lane = 4 output
// channels x one pixel: 64-byte store segments), no transposition.

struct RfArgs {
  const float* in; const float* wfrag; const float* bias; float* out;
  const float* dact_src; const float* res; const float* res2;
  long res_bs, res2_bs;          // batch strides of the residuals (floats)
  int N, Cin, Cout, H;           // W == 32
  int nseg, seg_rows, units;     // row segments per image, rows per segment, N * nseg work units
  int P, max_dr;                 // x-only steps in front of a segment (= hr <= 1); largest tap row offset
  int in_act, out_act, dact;
  int tap_dr[4], tap_dc[4];
};

constexpr int RF_WP = 34;        // ring row: column -1 .. 32 (the two pad columns stay zero)
constexpr int RF_RB = 2;         // ring rows
constexpr int RF_D = 2;          // register sets of staging loads in flight

template <bool MSE>              // MSE: the epilogue with derivative source / residual streams (data gradients, skip sums)
__global__ void __launch_bounds__(256, MSE ? 2 : 3) conv_b3r_kernel(const RfArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  u32x4* lds16 = reinterpret_cast<u32x4*>(lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // output tile (16 channels) of the chunk
  const int kg = lane >> 4, jn = lane & 15;
  const int co_chunk = blockIdx.y * B3_CO_CHUNK;
  const int plane = a.H * 32;
  // ---- weights: this wave's A fragments of all 8 K steps, for the whole launch
  bf16x8 af[8][3];
  {
    const u32x4* wf = reinterpret_cast<const u32x4*>(a.wfrag) + (size_t)blockIdx.y * (8 * 4 * 3 * 64);
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) af[j][p] = __builtin_bit_cast(bf16x8, wf[((j * 4 + wave) * 3 + p) * 64 + lane]);
  }
  // ---- LDS ring [8 channel groups][3 pieces][2 rows][34 columns] of 16-byte entries, zero filled once (pad columns, rows above the image)
  for (int i = tid; i < 8 * 3 * RF_RB * RF_WP; i += 256) lds16[i] = u32x4{0u, 0u, 0u, 0u};
  // staging slot of this thread: channel group js (8 channels), pixel ps of the new row
  const int js = tid >> 5, ps = tid & 31;
  const int goff = (8 * js) * plane + ps;
  const int s_ent = (js * 3) * RF_RB * RF_WP + 1 + ps;
  // B-fragment addressing of this lane: tap kg -> rows back from the newest ring row, column shift
  const int q_lane = a.max_dr - a.tap_dr[kg];
  const int col_lane = 1 + jn + a.tap_dc[kg];
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.bias) bias4 = *reinterpret_cast<const float4*>(a.bias + co_chunk + 16 * wave + 4 * kg);
  const int co_lane = co_chunk + 16 * wave + 4 * kg;      // first of this lane's four output channels

  float xv[RF_D][8];
  bool ok[RF_D];
#pragma unroll
  for (int d = 0; d < RF_D; ++d) {
    ok[d] = false;
#pragma unroll
    for (int c = 0; c < 8; ++c) xv[d][c] = 0.f;
  }
#define PG_RF_ISSUE(K, UNIT, S)                                                                    \
  {                                                                                                \
    /* unconditional loads (clamped rows, dropped at commit): see conv_wgrad_b3r_kernel */         \
    const int u_ = (UNIT) < a.units ? (UNIT) : a.units - 1;                                        \
    const int n_ = u_ / a.nseg;                                                                    \
    const int ir_ = (u_ - n_ * a.nseg) * a.seg_rows + (S) + a.max_dr;                              \
    ok[K] = ir_ >= 0 && ir_ < a.H;                                                                 \
    const int rc_ = ir_ < 0 ? 0 : (ir_ >= a.H ? a.H - 1 : ir_);                                    \
    int go_ = goff;                                                                                \
    asm volatile("" : "+v"(go_));                                                                  \
    const float* q_ = a.in + (size_t)n_ * a.Cin * plane + rc_ * 32 + go_;                          \
    _Pragma("unroll") for (int c = 0; c < 8; ++c) xv[K][c] = q_[(size_t)c * plane];                \
  }
#define PG_RF_COMMIT(K, ACT, ROW)                                                                  \
  {                                                                                                \
    float e_[8];                                                                                   \
    _Pragma("unroll") for (int c = 0; c < 8; ++c) e_[c] = ok[K] ? pg_apply_act(xv[K][c], ACT) : 0.f; \
    u32x4 h_, m_, l_;                                                                              \
    split8t(e_, h_, m_, l_);                                                                       \
    u32x4* dst_ = lds16 + s_ent + (ROW) * RF_WP;                                                   \
    dst_[0] = h_; dst_[RF_RB * RF_WP] = m_; dst_[2 * RF_RB * RF_WP] = l_;                          \
  }
#define PG_RF_ADVANCE(U, S) { (S) += 1; if ((S) == a.seg_rows) { (U) += (int)gridDim.x; (S) = -a.P; } }

  const bf16x8* xl = reinterpret_cast<const bf16x8*>(lds16);
  int unit = blockIdx.x, s = -a.P;
  int unit_i = unit, s_i = s;
  int rb = 0;
#pragma unroll
  for (int k = 0; k < RF_D; ++k) {
    PG_RF_ISSUE(k, unit_i, s_i)
    PG_RF_ADVANCE(unit_i, s_i)
  }
  __syncthreads();  // the zero fill is ordered before the first commit
  while (unit < a.units) {
#pragma unroll
    for (int k = 0; k < RF_D; ++k) {
      if (unit >= a.units) break;
      __syncthreads();  // the previous step's fragment reads are done
      asm volatile("" :: "v"(xv[k][0]), "v"(xv[k][1]), "v"(xv[k][2]), "v"(xv[k][3]), "v"(xv[k][4]), "v"(xv[k][5]), "v"(xv[k][6]),
                         "v"(xv[k][7]));
      switch (a.in_act) {  // wave-uniform
        case PG_ACT_RELU: PG_RF_COMMIT(k, PG_ACT_RELU, rb) break;
        case PG_ACT_ELU:  PG_RF_COMMIT(k, PG_ACT_ELU, rb) break;
        default:          PG_RF_COMMIT(k, PG_ACT_NONE, rb) break;
      }
      __syncthreads();
      // this step's output row and its epilogue operands (requested BEFORE the staging loads of step + 2: the wait in front of the
      // epilogue then leaves exactly those 8 loads in flight)
      const int n_img = unit / a.nseg;
      const int orow = (unit - n_img * a.nseg) * a.seg_rows + s;
      const size_t obase = ((size_t)n_img * a.Cout + co_lane) * plane + (size_t)(orow < 0 ? 0 : orow) * 32 + jn;
      float o0[2][4], o1[2][4], o2[2][4];
      if constexpr (MSE) {
        if (s >= 0) {
#pragma unroll
          for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const size_t off = (size_t)r * plane + n * 16;
              o0[n][r] = a.dact_src ? a.dact_src[obase + off] : 0.f;
              o1[n][r] = a.res ? a.res[(size_t)n_img * a.res_bs + (size_t)co_lane * plane + (size_t)orow * 32 + jn + off] : 0.f;
              o2[n][r] = a.res2 ? a.res2[(size_t)n_img * a.res2_bs + (size_t)co_lane * plane + (size_t)orow * 32 + jn + off] : 0.f;
            }
        }
      }
      PG_RF_ISSUE(k, unit_i, s_i)
      PG_RF_ADVANCE(unit_i, s_i)
      if (s >= 0) {
        int rr = rb - q_lane;
        rr = rr < 0 ? rr + RF_RB : rr;
        const bf16x8* xb = xl + rr * RF_WP + col_lane;   // + ((j * 3 + p) * RF_RB) * RF_WP + 16 n
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        // B fragments one group ahead of their MFMAs (plain epilogue) / just in time (MSE: its operand registers take the room —
        // with both the kernel spilled 11 registers at the 168 of three waves per SIMD)
        constexpr int NB = MSE ? 1 : 2;
        bf16x8 bf[NB][3];
        if constexpr (!MSE) {
#pragma unroll
          for (int p = 0; p < 3; ++p) bf[0][p] = xb[(p * RF_RB) * RF_WP];
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {   // group g = (K step j = g >> 1, pixel half n = g & 1)
          const int j = g >> 1, n = g & 1;
          if constexpr (MSE) {
#pragma unroll
            for (int p = 0; p < 3; ++p) bf[0][p] = xb[((j * 3 + p) * RF_RB) * RF_WP + n * 16];
          } else if (g + 1 < 16) {
#pragma unroll
            for (int p = 0; p < 3; ++p) bf[(g + 1) & 1][p] = xb[((((g + 1) >> 1) * 3 + p) * RF_RB) * RF_WP + ((g + 1) & 1) * 16];
          }
          constexpr int NBm = NB - 1;
          f32x4 c = acc[n];
          c = MFMA16B(af[j][2], bf[g & NBm][0], c);  // small terms first
          c = MFMA16B(af[j][0], bf[g & NBm][2], c);
          c = MFMA16B(af[j][1], bf[g & NBm][1], c);
          c = MFMA16B(af[j][1], bf[g & NBm][0], c);
          c = MFMA16B(af[j][0], bf[g & NBm][1], c);
          c = MFMA16B(af[j][0], bf[g & NBm][0], c);
          acc[n] = c;
        }
        // ---- epilogue: lane = output channels co_lane + r, pixel 16 n + jn of the row
        const float bb[4] = {bias4.x, bias4.y, bias4.z, bias4.w};
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc[n][r] + bb[r];
            v = a.out_act == PG_ACT_ELU ? pg_apply_act(v, PG_ACT_ELU) : (a.out_act == PG_ACT_RELU ? pg_apply_act(v, PG_ACT_RELU) : v);
            if constexpr (MSE) {
              if (a.dact_src) {
                const float d = a.dact == PG_ACT_ELU ? pg_act_grad(o0[n][r], PG_ACT_ELU)
                              : a.dact == PG_ACT_ELU_OUT ? pg_act_grad(o0[n][r], PG_ACT_ELU_OUT)
                              : a.dact == PG_ACT_RELU ? pg_act_grad(o0[n][r], PG_ACT_RELU) : 1.f;
                v *= d;
              }
              v += o1[n][r] + o2[n][r];
            }
            a.out[obase + (size_t)r * plane + n * 16] = v;
          }
      }
      PG_RF_ADVANCE(unit, s)
      rb = (s == -a.P) ? 0 : (rb + 1 == RF_RB ? 0 : rb + 1);
    }
  }
#undef PG_RF_ISSUE
#undef PG_RF_COMMIT
#undef PG_RF_ADVANCE
}

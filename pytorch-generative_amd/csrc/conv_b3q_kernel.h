// conv_b3q_kernel.h — the overlapped bf16x3 convolution kernel for ANY tap count (round 6); included by conv_b3_kernels.h
// inside its anonymous namespace (it uses B3Args, split8t, MFMA16B and the fragment types defined there).
//
// Reference call sites: every CausalConv2d / nn.Conv2d forward and data gradient with >= 64 output channels that is not a
// small 1x1 (nn/convolution.py:41-43; gated_pixel_cnn.py:63-96: the 1x3 / 2x1 / 1x2 / 1x1 convolutions of a gated layer
// on 128 / 256 channels; pixel_snail.py:41-55: the 2x2 64 -> 128 convolution of ResidualBlock; PixelCNN++'s 2x3 / 2x2).
//
// Why: conv_b3_kernel's "wide" form (one 8-wave workgroup per CU, 208-223 registers = two waves per SIMD) runs its phases
// back to back — its own clocks (profiles/r05_conv_wide_phase_clocks.txt) show the matrix pipe idle 63 % of every (tile,
// chunk) step: x loads, activation + split + LDS commit, the wait for the weight slab and two barriers are all serial with
// the MFMA loop, and with two waves per SIMD there is nobody to fill the gaps. conv_b3p_kernel (round 4) showed what fills
// them — four waves per SIMD at <= 128 registers, 32-pixel wave tiles, double-buffered LDS, ONE barrier per K step — but only
// for 4 taps and one output chunk. This kernel is that structure for every shape:
//   * 16 waves = ONE workgroup per CU (1024 threads, <= 128 registers, up to 160 KB of LDS): 8 pixel slices of 32 output
//     pixels x 2 "halves". ST = false: the halves are two 64-channel OUTPUT CHUNKS sharing one staged x tile (Cout % 128 ==
//     0); ST = true: the halves are two TILES (two images) sharing one weight slab (any Cout >= 64: PixelCNN++'s 160 / 320).
//   * a chunk of CIB = 8 cgs input channels lives in a double-buffered x tile [channel group][piece][tile pixel] of 16-byte
//     entries (as conv_b3_kernel's) and is consumed by `ksteps` K steps of 32 = four (channel group, tap) pairs each;
//   * the weight slab of ONE K step (12 KB per output chunk: ready-made A fragments [co tile][piece][lane]) is streamed by
//     LDS-DMA (global_load_lds_dwordx4) into a two-stage ring, one K step ahead: it never passes through registers;
//   * per K step one barrier. The side work of a chunk (commit of the NEXT chunk's x: activation, truncation split, three
//     ds_write_b128 per slot; then the global loads of the chunk after it) runs in the chunk's first K step — BEFORE the MFMA
//     block in half of the waves of every SIMD and AFTER it in the other half, so that right after a barrier two waves per
//     SIMD feed the matrix pipe while the other two do VALU / LDS work (in conv_b3p_kernel all waves start a step with their
//     side work and the pipe waits for the first of them);
//   * the epilogue transposes the accumulators with ds_bpermute_b32 (register to register through the LDS crossbar: no
//     scratch memory — the LDS belongs to the tiles) into "lane = pixel" order and streams v = out_act(acc + bias) *
//     act'(dact_src) + res + res2 in quarter tiles like conv_b3p_kernel (operand requests ahead of the stores).
// Same packed weights ([co chunk][channel chunk][k step][co tile][piece][lane], conv_b3.hip) and the same B3Args as
// conv_b3_kernel: the host (pg_b3_conv) chooses per launch.
#pragma once

constexpr int B3Q_THREADS = 1024;
constexpr int B3Q_SLAB16 = 768;   // 16-byte fragments of one output chunk's K-step slab: 4 co tiles x 3 pieces x 64 lanes

template <bool GL, int XS, bool ST>
__global__ void __launch_bounds__(B3Q_THREADS, 4) conv_b3q_kernel(const B3Args a) {
  constexpr int MT = 4, NT = 2;
  // output tiles whose A fragments are resident at a time: two (24 registers; 6 LDS reads ahead of 24 MFMAs) with one staging slot
  // per thread, one with two slots (their 8 more load registers must fit the 128 of four waves per SIMD)
  constexpr int QMH = XS == 1 ? 2 : 1;
  constexpr int GT = ST ? 512 : 1024;   // threads that stage one x tile
  constexpr int NCH = ST ? 1 : 2;       // output chunks per workgroup
  constexpr int NXT = ST ? 2 : 1;       // x tiles per workgroup
  extern __shared__ __attribute__((aligned(16))) float lds[];
  u32x4* lds16 = reinterpret_cast<u32x4*>(lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 3, slice = wave & 7;
  const int rt = blockIdx.x % a.tiles_per_img;
  const int n_first = blockIdx.x / a.tiles_per_img, nstep = gridDim.x / a.tiles_per_img;
  const int row0 = rt * a.TR;
  const int rows = min(a.TR, a.OH - row0);
  const int npx = rows * a.OW;
  const int co0 = (ST ? (int)blockIdx.y : (int)blockIdx.y * 2 + half) * B3_CO_CHUNK;
  const int L = a.OH * a.OW;
  const int plane = a.IH * a.IW;
  const int nchunk = a.Cin / a.CIB;
  const int nimg = n_first < a.N ? (a.N - n_first + nstep - 1) / nstep : 0;  // images this workgroup walks
  const int ntiles = ST ? (nimg + 1) >> 1 : nimg;                            // tile rounds (ST: two images per round)
  const int nsteps = ntiles * nchunk;                                        // chunk steps
  if (nsteps == 0) return;
  const int nq = nsteps * a.ksteps;                                          // K steps
  const int kq = lane >> 4;
  const int xbuf16 = a.cgs * 3 * a.plane16;   // 16-byte entries of one x buffer
  const int xt = ST ? half : 0;               // the x tile this wave computes on / this thread stages
  // wave roles: a slice beyond the tile's pixels has nothing to compute; the order of side work and MFMA block alternates
  // between the two wave pairs of a SIMD (waves w, w + 4, w + 8, w + 12 share one)
  const bool has_px = slice * (NT * 16) < npx;
    const bool mfma_first = ((wave >> 2) & 1) != 0 && !PG_DBG_BIT(a.dbg, 32);

  int pixoff[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int p = (slice * NT + n) * 16 + (lane & 15);
    const int pc = p < npx ? p : 0;
    const int r = pc / a.OW;
    pixoff[n] = r * a.tile_w + (pc - r * a.OW);
  }
  // store phase of the epilogue: lane & 31 = pixel of the wave's 32-pixel slice
  unsigned opx32;
  bool sok32;
  {
    const int p = slice * (NT * 16) + (lane & 31);
    sok32 = p < npx;
    const int pc = sok32 ? p : 0;
    const int r = pc / a.OW;
    opx32 = (unsigned)((row0 + r) * a.OW + (pc - r * a.OW));
  }
  // K-group table: group g = 4 ks + kq lives at plane offset of its channel group + its tap's offset
  int* gtab = reinterpret_cast<int*>(lds + a.b_off + B3_CO_CHUNK * NCH);
  if (tid < B3_MAXG) gtab[tid] = tid < a.groups ? a.g_cg[tid] * 3 * a.plane16 + a.g_tapoff[tid] : 0;

  // staging slots: (channel group, tile row, tile column) -> 8 channel loads of one tile pixel; pixels outside the image
  // load the chunk's first word and commit to the dump entry (halo entries are zeroed once and never written)
  const int tg = ST ? (tid & 511) : tid;
  int s_goff[XS], s_loff[XS];
#pragma unroll
  for (int k = 0; k < XS; ++k) {
    int e = tg + k * GT;
    const bool in = e < a.xslots;
    e = in ? e : 0;
    const int tc = e % a.tile_w;
    e /= a.tile_w;
    const int tr = e % a.tile_h;
    const int cg = e / a.tile_h;
    const int ir = row0 + a.min_dr + tr, ic = a.min_dc + tc;
    const bool ok = in && ir >= 0 && ir < a.IH && ic >= 0 && ic < a.IW;
    s_goff[k] = ok ? (cg * 8) * plane + ir * a.IW + ic : 0;
    s_loff[k] = ok ? cg * 3 * a.plane16 + tr * a.tile_w + tc : -1;
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int i = tid; i < NXT * 2 * xbuf16; i += B3Q_THREADS) lds16[i] = u32x4{0u, 0u, 0u, 0u};
  if (tid < B3_CO_CHUNK * NCH) {
    const int co = (int)blockIdx.y * NCH * B3_CO_CHUNK + tid;
    lds[a.b_off + tid] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
  }
  __syncthreads();  // the zero fill is ordered before the first commit (other threads own the same entries there)

  float xv[XS][8];
  // the image of tile round `tl` for x tile `xt` (ST: image 2 tl + xt of this workgroup's list; clamped to a valid image when
  // the list has an odd length: the idle half stages a valid tile and computes nothing)
#define PG_Q_IMG(TL) (n_first + (ST ? min(2 * (TL) + xt, nimg - 1) : (TL)) * nstep)
  int l_tl = 0, l_ch = 0;   // (tile round, chunk) of the step whose loads are issued next
#define PG_Q_ISSUE_X()                                                                          \
  {                                                                                             \
    const float* src_ = a.in + ((size_t)PG_Q_IMG(l_tl) * a.Cin + l_ch * a.CIB) * plane;         \
    _Pragma("unroll") for (int k = 0; k < XS; ++k) {                                            \
      _Pragma("unroll") for (int c = 0; c < 8; ++c) xv[k][c] = (src_ + (size_t)c * plane)[(unsigned)s_goff[k]]; \
    }                                                                                           \
    if (++l_ch == nchunk) { l_ch = 0; ++l_tl; }                                                 \
  }
#define PG_Q_COMMIT_SLOT(K, BUF, ACT)                                                           \
  {                                                                                             \
    float e_[8];                                                                                \
    _Pragma("unroll") for (int c = 0; c < 8; ++c) e_[c] = pg_apply_act(xv[K][c], ACT);          \
    u32x4 h_, m_, l_;                                                                           \
    split8t(e_, h_, m_, l_);                                                                    \
    const bool ok_ = s_loff[K] >= 0;                                                            \
    const int dst_ = ok_ ? (xt * 2 + (BUF)) * xbuf16 + s_loff[K] : a.dump16;                    \
    const int pst_ = ok_ ? a.plane16 : 0;                                                       \
    lds16[dst_] = h_;                                                                           \
    lds16[dst_ + pst_] = m_;                                                                    \
    lds16[dst_ + 2 * pst_] = l_;                                                                \
  }
#define PG_Q_COMMIT_X(BUF)                                                                      \
  _Pragma("unroll") for (int k = 0; k < XS; ++k) {                                              \
    switch (a.in_act) { /* wave-uniform */                                                      \
      case PG_ACT_RELU: PG_Q_COMMIT_SLOT(k, BUF, PG_ACT_RELU) break;                            \
      case PG_ACT_ELU:  PG_Q_COMMIT_SLOT(k, BUF, PG_ACT_ELU) break;                             \
      case PG_ACT_GELU: if constexpr (GL) { PG_Q_COMMIT_SLOT(k, BUF, PG_ACT_GELU) } break;      \
      default:          PG_Q_COMMIT_SLOT(k, BUF, PG_ACT_NONE) break;                            \
    }                                                                                           \
  }
  // ---- weight slab of K step Q -> ring stage Q & 1, by LDS-DMA: NCH x 12 pieces of 1 KB, piece p by wave p (and p + 16).
  // The DMA is issued by inline asm (the compiler never emits it, and its counter model must not see it); every wave waits
  // for its own pieces with an explicit s_waitcnt before the barrier that publishes the stage.
  const float4* wsrc_b = reinterpret_cast<const float4*>(a.wfrag) + (size_t)blockIdx.y * NCH * nchunk * a.wslab4 + lane;
  const unsigned lds_b = (unsigned)(size_t)((__attribute__((address_space(3))) char*)lds);
#define PG_Q_DMA(J, KS, STAGE)  /* (channel chunk, K step) of the slab, ring stage */          \
  {                                                                                             \
    for (int p_ = wave; p_ < NCH * 12; p_ += 16) {                                              \
      const int c_ = p_ / 12, r_ = p_ - c_ * 12;                                                \
      const float4* g_ = wsrc_b + ((size_t)c_ * nchunk + (J)) * a.wslab4 + (KS) * B3Q_SLAB16 + r_ * 64; \
      const unsigned d_ = lds_b + (unsigned)(a.w_off16 + ((STAGE) * NCH + c_) * B3Q_SLAB16 + r_ * 64) * 16u; \
      unsigned keep_;                                                                           \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                   : "=&s"(keep_) : "v"(g_), "s"(d_) : "memory");                               \
    }                                                                                           \
  }
  const float* bl = lds + a.b_off + (ST ? 0 : half * B3_CO_CHUNK);
  const bf16x8* xl = reinterpret_cast<const bf16x8*>(lds16) + (size_t)xt * 2 * xbuf16;
  const bf16x8* wl = reinterpret_cast<const bf16x8*>(lds16 + a.w_off16) + (ST ? 0 : half * B3Q_SLAB16) + lane;

  // the MFMA block of one K step: B fragments of the wave's two pixel groups resident, A fragments in two halves of two
  // output tiles (24 + 24 fragment registers)
#define PG_Q_MFMA(KS, XBUF, STAGE)                                                              \
  if (has_px && tile_ok && !PG_DBG_BIT(a.dbg, 4)) {                                             \
    const bf16x8* xb_ = xl + (XBUF) * xbuf16 + gtab[4 * (KS) + kq];                             \
    const bf16x8* wb_ = wl + (STAGE) * (NCH * B3Q_SLAB16);                                      \
    bf16x8 bf_[NT][3];                                                                          \
    _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                            \
      bf_[n][0] = xb_[pixoff[n]];                                                               \
      bf_[n][1] = xb_[pixoff[n] + a.plane16];                                                   \
      bf_[n][2] = xb_[pixoff[n] + 2 * a.plane16];                                               \
    }                                                                                           \
    _Pragma("unroll") for (int mh = 0; mh < MT / QMH; ++mh) {                                   \
      bf16x8 ah_[QMH][3];                                                                       \
      _Pragma("unroll") for (int m = 0; m < QMH; ++m)                                           \
        _Pragma("unroll") for (int pc = 0; pc < 3; ++pc) ah_[m][pc] = wb_[((QMH * mh + m) * 3 + pc) * 64]; \
      /* the six piece products of one accumulator form a dependent chain ("small terms first"): the chains of the QMH x NT \
         tiles are interleaved term by term, so that consecutive MFMAs never wait for each other's result */ \
      _Pragma("unroll") for (int t = 0; t < 6; ++t) {                                           \
        constexpr int pa_[6] = {2, 0, 1, 1, 0, 0}, pb_[6] = {0, 2, 1, 0, 1, 0};                 \
        _Pragma("unroll") for (int n = 0; n < NT; ++n)                                          \
          _Pragma("unroll") for (int m = 0; m < QMH; ++m)                                       \
            acc[QMH * mh + m][n] = MFMA16B(ah_[m][pa_[t]], bf_[n][pb_[t]], acc[QMH * mh + m][n]); \
      }                                                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                        \
    }                                                                                           \
  }
  // side work of a chunk step (in its first K step): commit the next chunk's x (loaded one chunk step ago) into the other
  // buffer, then request the chunk after it
#define PG_Q_SIDE(XBUF)                                                                         \
  {                                                                                             \
    if (more && !PG_DBG_BIT(a.dbg, 2)) { PG_Q_COMMIT_X((XBUF) ^ 1) }                            \
    if (more2 && !PG_DBG_BIT(a.dbg, 1)) { PG_Q_ISSUE_X() }                                      \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  }

  // prologue: chunk step 0 committed, its first slab landed, chunk step 1 in the registers
  PG_Q_ISSUE_X()
  PG_Q_DMA(0, 0, 0)
  PG_Q_COMMIT_X(0)
  if (nsteps > 1) { PG_Q_ISSUE_X() }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // phase clocks (ablation builds only, tools/exp/b3_phase_prof.py): 0 MFMA block, 1 barrier, 2 side work, 3 load retire + DMA issue,
  // 4 epilogue, 5 wait for the slab
  PG_PROF_DECL
#ifdef PG_ABLATE
#define PG_Q_MARK(I) if (a.prof) PG_PROF_MARK(I)   /* no clock reads (s_memtime + lgkmcnt wait) in the plain ablation timings */
#else
#define PG_Q_MARK(I)
#endif
  int q = 0, chunk = 0, tl = 0;
  for (int step = 0; step < nsteps; ++step) {
    const int cur = step & 1;
    const bool more = step + 1 < nsteps, more2 = step + 2 < nsteps;
    const bool tile_ok = !ST || 2 * tl + xt < nimg;   // ST: the second half of the last round may have no image
    for (int ks = 0; ks < a.ksteps; ++ks, ++q) {
      const int stage = q & 1;
      const bool more_q = q + 1 < nq;
      // first K step of a chunk step: the side work (commit of the next chunk's x, loads of the one after) runs BEFORE the MFMA
      // block in the side-first waves and AFTER it in the others
      const bool side_before = ks == 0 && !mfma_first, side_after = ks == 0 && mfma_first;
      if (side_before) {
        // the x loads of the chunk to commit were issued a whole chunk step ago: retire them HERE (a modelled s_waitcnt
        // vmcnt(0)), so that the commit needs no wait of its own while the slab DMA — invisible to the compiler's counter
        // model — is in flight
        __builtin_amdgcn_s_waitcnt(0x0F70);
      }
      if (more_q && !PG_DBG_BIT(a.dbg, 16)) {
        // the next K step's slab: the same chunk's next K step, or the first K step of the next chunk (no divisions: a
        // wave-uniform integer division is ~25 VALU instructions, and the youngest waves of a SIMD get the leftover issue
        // slots — the first version spent 1 200-2 500 cycles per K step here, profiles/r06_conv_q_phase_clocks.txt)
        const bool same_ = ks + 1 < a.ksteps;
        const int nj_ = same_ ? chunk : (chunk + 1 == nchunk ? 0 : chunk + 1);
        PG_Q_DMA(nj_, same_ ? ks + 1 : 0, stage ^ 1)
      }
      PG_Q_MARK(3)
      if (side_before) PG_Q_SIDE(cur)
      PG_Q_MARK(2)
      PG_Q_MFMA(ks, cur, stage)
      PG_Q_MARK(0)
      // (side-after waves: the commit waits for this wave's x loads — issued one chunk step ago, at the end of that step — with
      // a vmcnt(0) of the compiler's; by then the slab DMA issued above has had the whole MFMA block to land)
      if (side_after) PG_Q_SIDE(cur)
      PG_Q_MARK(2)
      // this K step's DMA is older than the x loads a side block issued behind it (8 per slot): wait for everything but those
      if (ks == 0 && more2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 * XS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PG_Q_MARK(5)
      if (ks + 1 == a.ksteps && ++chunk == nchunk) {
        chunk = 0;
        // ---- epilogue of the tile: v = out_act(acc + bias) * act'(dact_src) + res + res2
        const int n_img = PG_Q_IMG(tl);
        ++tl;
        if (has_px && tile_ok && !PG_DBG_BIT(a.dbg, 8)) {
          int Lv = L;
          asm volatile("" : "+s"(Lv));
          const size_t so = ((size_t)n_img * a.Cout + co0) * Lv;
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
          // accumulator layout: lane (kq, j) holds channels 4 kq + r (r = register) of pixel 16 n + j; store layout: lane
          // (h8, n, j) = (lane >> 5, (lane >> 4) & 1, lane & 15) holds channels 8 h8 + c of pixel 16 n + j. Value c of the
          // store layout comes from lane (2 h8 + (c >> 2)) * 16 + j, register r = c & 3 of pixel group n: two ds_bpermute
          // (one per pixel group of the source) and a select on the destination's n.
          const int h8 = lane >> 5;
          const bool nsel = ((lane >> 4) & 1) != 0;
          int bp0 = ((2 * h8) * 16 + (lane & 15)) * 4;
          unsigned ox = opx32;   // opaque copies: formed here, not kept in registers across the step loop
          int hrow = 8 * h8;
          asm volatile("" : "+v"(ox), "+v"(hrow), "+v"(bp0));
          float q0[4], q1[4], q2[4];
#define PG_Q_REQ(M, H)                                                                     \
  _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                          \
    const int cc = (M) * 16 + hrow + (H) * 4 + c;                                          \
    const size_t off_ = (size_t)((fullc || cc < cvalid_p) ? cc : 0) * Lv;                  \
    if (st0) q0[c] = (st0 + off_)[ox];                                                     \
    if (st1) q1[c] = (st1 + off_)[ox];                                                     \
    if (st2) q2[c] = (st2 + off_)[ox];                                                     \
  }
          if (any_op) { PG_Q_REQ(0, 0) }
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              const int ad = bp0 + (c >> 2) * 64;
              // (element copies first: __builtin_bit_cast applied to a vector ELEMENT reads element 0 whatever the index — clang 22)
              const float a0 = acc[m][0][c & 3], a1 = acc[m][1][c & 3];
              const int t0 = __builtin_amdgcn_ds_bpermute(ad, __float_as_int(a0));
              const int t1 = __builtin_amdgcn_ds_bpermute(ad, __float_as_int(a1));
              v[c] = __int_as_float(nsel ? t1 : t0) + bl[m * 16 + 8 * h8 + c];
            }
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
                if (hh == 0) { PG_Q_REQ(m, 1) }
                else if (m + 1 < MT) { PG_Q_REQ(m + 1, 0) }
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
#undef PG_Q_REQ
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        // (the epilogue's loads / stores are younger than this K step's DMA, which the wait above has retired)
      }
      PG_Q_MARK(4)
      if (more_q) __syncthreads();  // the other x buffer / slab stage are committed; every wave is done with the current ones
      PG_Q_MARK(1)
    }
  }
  PG_PROF_DUMP(16, wave, nq)
#undef PG_Q_MARK
#undef PG_Q_IMG
#undef PG_Q_ISSUE_X
#undef PG_Q_COMMIT_SLOT
#undef PG_Q_COMMIT_X
#undef PG_Q_DMA
#undef PG_Q_MFMA
#undef PG_Q_SIDE
}

template <bool GL>
void b3q_launch(const B3Args& a, int xs, bool st_mode, dim3 grid, size_t shmem, hipStream_t st) {
#define PG_B3Q_L(XSV, STV)                                                                            \
  {                                                                                                   \
    static const hipError_t attr_##XSV##_##STV = hipFuncSetAttribute(                                 \
        reinterpret_cast<const void*>(conv_b3q_kernel<GL, XSV, STV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    (void)attr_##XSV##_##STV;                                                                         \
    hipLaunchKernelGGL((conv_b3q_kernel<GL, XSV, STV>), grid, dim3(B3Q_THREADS), shmem, st, a);       \
  }
  if (st_mode) {
    if (xs == 1) PG_B3Q_L(1, true) else PG_B3Q_L(2, true)
  } else {
    if (xs == 1) PG_B3Q_L(1, false) else PG_B3Q_L(2, false)
  }
#undef PG_B3Q_L
}

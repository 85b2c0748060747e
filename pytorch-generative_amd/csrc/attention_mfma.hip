// attention_mfma.hip — matrix-core variant of the causal attention kernels for d_k = 4.
//
// Both flagship configurations have d_k = 4 (ImageGPT 16 embed / 4 heads; PixelSNAIL
// attention_key_channels = 4), which is EXACTLY the contraction depth of
// v_mfma_f32_16x16x4_f32 (fp32 in / fp32 accumulate, bitwise an fmaf chain). One MFMA therefore
// produces a 16 keys x 16 queries tile of S^T = K Q^T with no padding, on the matrix pipe, leaving
// the VALU for the softmax and (for d_v = 4, where an MFMA tile would be 4x padding) the P.V update.
//
//   A[i = lane&15][k = lane>>4] = K[key0 + i][k]      (one ds_read_b32 from the K^T tile in LDS)
//   B[k = lane>>4][j = lane&15] = q[query0 + j][k]    (one VGPR per query tile, loaded once)
//   D[row = 4*(lane>>4) + r][col = lane&15] = S^T[key0 + row][query0 + col]
// so a lane holds, for query (lane&15) of every query tile, the scores of keys 4g..4g+3
// (g = lane>>4) of each 16-key tile: the online softmax runs lane-locally on that key subset
// (running max m, sum l, output o per query tile) with a LAZY rescale (only when some lane's new
// maximum exceeds the old one by more than 2^8 — wave-uniform branch), and the four key subsets
// are merged once at the end with two xor-shuffles.
// Work is handed out in balanced block pairs exactly as in attention.hip.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr float NEG_BIG = -1.0e30f;
constexpr float POS_BIG = 1.0e30f;
constexpr float RESCALE_TH = 8.0f;  // log2 units

struct MfmaArgs {
  const float* q; const float* k; const float* v;
  float* o; float* lse2;
  int N, heads, L, strict, blocks_per_wg, Lp;
  long q_bs, k_bs, v_bs, o_bs;
  float scale2;
};

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// ------------------------------------------------------------------------- forward, d_v = 4
__global__ void __launch_bounds__(512) attn_fwd_mfma44_kernel(const MfmaArgs a) {
  extern __shared__ float4 lds4[];
  float* kt = reinterpret_cast<float*>(lds4);                           // K^T [4][Lp]
  float4* vr = reinterpret_cast<float4*>(kt + 4 * (size_t)a.Lp);        // V rows [rows]
  const int h = blockIdx.y, n = blockIdx.z;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int qi = lane & 15, g = lane >> 4;
  const int L = a.L;
  const int NB = (L + 63) >> 6;
  const int wg = gridDim.x - 1 - blockIdx.x;
  const int first = wg * a.blocks_per_wg;
  const int nb = min(a.blocks_per_wg, NB - first);
  const int lo = first + wave, hi = first + nb - 1 - wave;
  const bool on = lo <= hi;
  const int bA = hi, bB = lo < hi ? lo : -1;

  const float* qp = a.q + (size_t)n * a.q_bs + (size_t)h * 4 * L;
  const float* kp = a.k + (size_t)n * a.k_bs + (size_t)h * 4 * L;
  const float* vp = a.v + (size_t)n * a.v_bs + (size_t)h * 4 * L;

  // stage every key/value row the workgroup needs (rows >= L zero filled)
  const int rows_needed = min(64 * (first + nb), ((L + 15) >> 4) << 4);
  for (int m = threadIdx.x; m < rows_needed; m += blockDim.x) {
    const bool ok = m < L;
    const int mc = ok ? m : L - 1;
    const float k0 = kp[mc], k1 = kp[(size_t)L + mc], k2 = kp[2 * (size_t)L + mc], k3 = kp[3 * (size_t)L + mc];
    const float v0 = vp[mc], v1 = vp[(size_t)L + mc], v2 = vp[2 * (size_t)L + mc], v3 = vp[3 * (size_t)L + mc];
    kt[m] = ok ? k0 : 0.f; kt[a.Lp + m] = ok ? k1 : 0.f; kt[2 * a.Lp + m] = ok ? k2 : 0.f; kt[3 * a.Lp + m] = ok ? k3 : 0.f;
    vr[m] = ok ? make_float4(v0, v1, v2, v3) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  if (!on) return;

  // query tiles: t in [0,4) -> block A, [4,8) -> block B
  float qf[8], mr[8], ls[8], acc[8][4];
  int myq[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int blk = t < 4 ? bA : bB;
    const int qidx = blk >= 0 ? 64 * blk + 16 * (t & 3) + qi : L;
    myq[t] = qidx;
    const int qc = min(qidx, L - 1);
    qf[t] = (qidx < L) ? qp[(size_t)g * L + qc] * a.scale2 : 0.f;
    mr[t] = NEG_BIG; ls[t] = 0.f;
    acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0.f;
  }
  const int ntile = bB >= 0 ? 8 : 4;
  const int endA = min(64 * bA + 63, L - 1) - a.strict + 1;  // keys [0, endA) needed by block A

  for (int k0 = 0; k0 < endA; k0 += 16) {
    const float kf = kt[g * a.Lp + k0 + qi];
    float4 vv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) vv[r] = vr[k0 + 4 * g + r];
    const int key0 = k0 + 4 * g;  // this lane's first key of the tile
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int blk = t < 4 ? bA : bB;
      const int qmin = 64 * blk + 16 * (t & 3);
      // wave-uniform skip: query tile not owned, or the whole key tile lies above its diagonal
      // (an `if` around the body, not break/continue: the t loop must unroll fully so that every
      // per-tile array stays in registers)
      if (t < ntile && k0 <= min(qmin + 15, L - 1) - a.strict) {
      f32x4 s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf, qf[t], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      const bool diag = (k0 + 15) > (qmin - a.strict);  // wave-uniform: needs per-lane predicates
      const int last = myq[t] - a.strict;               // last allowed key of this lane's query
      if (diag) {
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] = (key0 + r) <= last ? s[r] : NEG_BIG;
      }
      const float cmax = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
      if (__any(cmax > mr[t] + RESCALE_TH)) {
        const float mnew = fmaxf(mr[t], cmax);
        const float alpha = fast_exp2(mr[t] - mnew);
        ls[t] *= alpha;
        acc[t][0] *= alpha; acc[t][1] *= alpha; acc[t][2] *= alpha; acc[t][3] *= alpha;
        mr[t] = mnew;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = fast_exp2(s[r] - mr[t]);
        if (diag) p = (key0 + r) <= last ? p : 0.f;
        ls[t] += p;
        acc[t][0] = fmaf(p, vv[r].x, acc[t][0]);
        acc[t][1] = fmaf(p, vv[r].y, acc[t][1]);
        acc[t][2] = fmaf(p, vv[r].z, acc[t][2]);
        acc[t][3] = fmaf(p, vv[r].w, acc[t][3]);
      }
      }
    }
  }

  // merge the four key subsets (lane groups g) of every query, then group g writes channel g
  float* op = a.o + (size_t)n * a.o_bs + (size_t)h * 4 * L;
  float* lp = a.lse2 + ((size_t)n * a.heads + h) * L;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t < ntile) {
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
      const float mo = __shfl_xor(mr[t], off, 64), lo_ = __shfl_xor(ls[t], off, 64);
      const float mn = fmaxf(mr[t], mo);
      const float ca = fast_exp2(mr[t] - mn), cb = fast_exp2(mo - mn);
      ls[t] = ls[t] * ca + lo_ * cb;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float oo = __shfl_xor(acc[t][j], off, 64);
        acc[t][j] = acc[t][j] * ca + oo * cb;
      }
      mr[t] = mn;
    }
    if (myq[t] < L) {
      const float inv = ls[t] > 0.f ? 1.f / ls[t] : 0.f;
      const float val = g == 0 ? acc[t][0] : (g == 1 ? acc[t][1] : (g == 2 ? acc[t][2] : acc[t][3]));
      op[(size_t)g * L + myq[t]] = val * inv;
      if (g == 0) lp[myq[t]] = ls[t] > 0.f ? mr[t] + log2f(ls[t]) : POS_BIG;
    }
    }
  }
}

}  // namespace

// Returns 1 if the MFMA path handled the call, 0 if the shape is not covered (caller falls back to
// the VALU kernels of attention.hip), or -1000 - hipError_t if the launch failed.
int pg_attn_fwd_mfma_try(const float* q, const float* k, const float* v, float* o, float* lse2, int N,
                         int heads, int L, int dk, int dv, long q_bs, long k_bs, long v_bs, long o_bs,
                         int strict, hipStream_t st) {
  if (dk != 4 || dv != 4 || L > 4096) return 0;
  MfmaArgs a;
  a.q = q; a.k = k; a.v = v; a.o = o; a.lse2 = lse2;
  a.N = N; a.heads = heads; a.L = L; a.strict = strict;
  a.q_bs = q_bs; a.k_bs = k_bs; a.v_bs = v_bs; a.o_bs = o_bs;
  a.scale2 = 0.5f * 1.44269504088896340736f;  // log2(e) / sqrt(4)
  const int NB = (L + 63) / 64;
  const int bpw = NB < 16 ? NB : 16;
  a.blocks_per_wg = bpw;
  const int rows = ((L + 15) / 16) * 16;
  a.Lp = ((rows + 31) / 32) * 32 + 16;  // K^T row stride == 16 (mod 32): conflict-free fragment reads
  const size_t shmem = ((size_t)4 * a.Lp + (size_t)4 * rows) * sizeof(float);
  dim3 grid((unsigned)((NB + bpw - 1) / bpw), (unsigned)heads, (unsigned)N);
  dim3 block((unsigned)(64 * ((bpw + 1) / 2)));
  hipLaunchKernelGGL(attn_fwd_mfma44_kernel, grid, block, shmem, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    pg_set_error("pg_causal_attn_fwd(mfma): launch failed: %s", hipGetErrorString(e));
    return -1000 - (int)e;
  }
  return 1;
}

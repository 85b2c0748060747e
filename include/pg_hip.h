/*
 * pg_hip.h — C-ABI of libpg_hip.so: the MI355X (gfx950) kernels behind the
 * pytorch_generative.nn operator surface (masked conv + causal attention hot path).
 *
 * The reference (EugenHotaj/pytorch-generative) has no FFI: its arithmetic is torch
 * (SURVEY.md §8b). Each entry point below therefore cites the reference *call site*
 * whose torch op it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *  - plain pointers + sizes; all tensors fp32, dense NCHW ("L" = H*W contiguous).
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch's allocator);
 *    the library never allocates, frees, retains, or synchronises (hipGraph-capturable).
 *  - `stream` is a hipStream_t passed as void*.
 *  - return 0 on success; <0 = argument error (PG_E*); >0 = hipError_t of the launch.
 *    pg_last_error() returns a thread-local message for the last non-zero return.
 *  - functions are re-entrant (called from the autograd thread as well as the main thread).
 */
#ifndef PG_HIP_H_
#define PG_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PG_ABI_VERSION 1

#define PG_EINVAL (-1)  /* bad argument */
#define PG_ESHAPE (-2)  /* shape outside what the kernels support (no silent fallback) */

#define PG_MAX_TAPS 64

/* activation ids (pg_act_*, conv prologues) */
#define PG_ACT_NONE 0
#define PG_ACT_RELU 1 /* nn.ReLU: models/autoregressive/pixel_cnn.py:33-50 */
#define PG_ACT_ELU 2  /* F.elu alpha=1: models/autoregressive/pixel_snail.py:27-28 */
#define PG_ACT_GELU 3 /* nn.GELU() exact erf: models/autoregressive/image_gpt.py:44 */
/* only as the `dact` of pg_conv2d_mfma: dact_src holds ELU's OUTPUT y, derivative = y > 0 ? 1 : y + 1
 * (a producer convolution fused F.elu into its epilogue, pixel_snail.py:54) */
#define PG_ACT_ELU_OUT 4

/* gate kinds for pg_gated_* (nn/convolution.py:46-66) */
#define PG_GATE_TANH 0     /* tanh(a)*sigmoid(b): gated_pixel_cnn.py:57 */
#define PG_GATE_IDENTITY 1 /* a*sigmoid(b): pixel_snail.py:50 */

int pg_abi_version(void);
const char* pg_last_error(void);

/* ---------------------------------------------------------------------------------------
 * Direct 2-D convolution over an explicit tap list (stride 1).
 * Replaces torch.nn.Conv2d.forward as reached from
 *   CausalConv2d.forward           nn/convolution.py:41-43  (taps = the mask's non-zero taps)
 *   every nn.Conv2d on the path    gated_pixel_cnn.py:63-96, pixel_snail.py:41-55 (pad+crop
 *                                  become tap offsets + the out extent), attention.py:105-118,
 *                                  image_gpt.py:40-48 (1x1)
 * out[n,co,r,c] = bias[co] + sum_ci sum_t wpk[ci][t][co] * act(in[n,ci,r+dr[t],c+dc[t]])
 * (zero outside the input extent). `wpk` is the packed weight made by pg_pack_conv_weight.
 * The same kernel is the data-gradient: call it with the transposed pack and negated taps.
 * `res` (optional, may be NULL) is added to the result (fused residual).
 * `dact_src`/`dact` (optional): the result is multiplied by act'(dact_src) — the data gradient of
 * a convolution whose forward fused `in_act` (dact_src = that convolution's raw input).
 * ------------------------------------------------------------------------------------- */
int pg_conv2d_taps(const float* in, const float* wpk, const float* bias, const float* res,
                   float* out, int N, int Cin, int IH, int IW, int Cout, int OH, int OW,
                   int T, const int* tap_dr, const int* tap_dc, int in_act,
                   const float* dact_src, int dact, void* stream);

/* packed-weight helper: wpk[a][t][b] (b padded to b_pad, zero filled),
 *   transpose==0: = w[b][a][tap_u[t]][tap_v[t]]   (forward:   a=Cin,  b=Cout)
 *   transpose==1: = w[a][b][tap_u[t]][tap_v[t]]   (data grad: a=Cout, b=Cin)
 * w is the torch layout (Cout, Cin, KH, KW) (nn/convolution.py:35-36). */
int pg_pack_conv_weight(const float* w, float* wpk, int Cout, int Cin, int KH, int KW, int T,
                        const int* tap_u, const int* tap_v, int transpose, int b_pad,
                        void* stream);
/* number of floats pg_pack_conv_weight writes */
size_t pg_packed_weight_floats(int a, int T, int b);
/* padded extent of b the conv kernel expects */
int pg_conv_b_pad(int b);

/* ---------------------------------------------------------------------------------------
 * The same convolution as an implicit GEMM on the fp32 matrix cores (csrc/conv_mfma.hip):
 * M = output channels, N = output pixels, K = (input-channel group of 4, tap). Replaces the same
 * reference call sites as pg_conv2d_taps (nn/convolution.py:41-43, gated_pixel_cnn.py:63-96,
 * pixel_snail.py:41-55, every 1x1 convolution) once both channel counts fill MFMA tiles
 * (pg_conv_mfma_supported != 0; otherwise call pg_conv2d_taps). `wfrag` = A fragments made by
 * pg_pack_conv_weight_frag (transpose = 1 + negated taps: the data gradient).
 * Epilogue order: + bias, out_act (PG_ACT_*), + res, * dact'(dact_src).
 * ------------------------------------------------------------------------------------- */
int pg_conv2d_mfma(const float* in, const float* wfrag, const float* bias, const float* res,
                   float* out, int N, int Cin, int IH, int IW, int Cout, int OH, int OW, int T,
                   const int* tap_dr, const int* tap_dc, int in_act, const float* dact_src,
                   int dact, int out_act, int fmt, void* stream);
/* The same with the full epilogue of the bf16x3 format:
 *   out = out_act(conv + bias) * act'(dact_src) + res + res2
 * (the derivative BEFORE the residuals). In a data gradient res / res2 are the pass-through gradients
 * of skip connections on the convolution's input (autograd's gradient-sum `add` kernels, fused);
 * res_bs / res2_bs: their batch strides in floats (0 = dense Cout * OH * OW), so that a residual may
 * be a channel slice of a wider tensor (the slice of a concatenation's gradient). fmt = PG_CONV_FMT_F32
 * accepts only res2 = NULL, dense res and not (res and dact_src) together. */
int pg_conv2d_mfma_ex(const float* in, const float* wfrag, const float* bias, const float* res,
                      float* out, int N, int Cin, int IH, int IW, int Cout, int OH, int OW, int T,
                      const int* tap_dr, const int* tap_dc, int in_act, const float* dact_src,
                      int dact, int out_act, int fmt, const float* res2, long res_bs, long res2_bs,
                      void* stream);
/* Round 6: the convolution of a gated block together with its GatedActivation (+ residual) — nn/convolution.py:62-66 behind the
 * 2C-channel convolution of pixel_snail.py:41-56 and gated_pixel_cnn.py:63-96: out (N, 2C, OH, OW) = conv(in_act(in)) + bias + res is
 * written as usual (the gate's backward, pg_gated_bwd, reads it; res = the convolution's own residual — GatedPixelCNN's link and
 * vertical-stack sums — or NULL) and gate_out (N, C, OH, OW) = gate_res + act(out[:, :C]) * sigmoid(out[:, C:]); gate_res may be
 * NULL. 2C must be a multiple of 128 on the bf16x3 format's wide kernel; pg_conv_gate_fusable() returns 1 for the shapes it takes.
 * wfrag: pg_pack_conv_weight_frag(fmt = PG_CONV_FMT_B3_GATE). */
int pg_conv_gate_fusable(int Cin, int Cout, int OH, int OW, int T, const int* tap_dr, const int* tap_dc);
int pg_conv2d_mfma_gate(const float* in, const float* wfrag, const float* bias, const float* res, float* out, int N, int Cin,
                        int IH, int IW, int Cout, int OH, int OW, int T, const int* tap_dr, const int* tap_dc, int in_act, int gate,
                        const float* gate_res, float* gate_out, void* stream);
/* Round 6: the "dual" 1x1 data gradient of PixelSNAIL's block tail (pixel_snail.py:109-119: both = elu(conv_a(..)) + r,
 * out = elu(conv_o(elu(both))) + x). conv_o's data gradient owes BOTH producers of its input their ELU derivatives (they were
 * called with the out_pre_scaled protocol). in = dy of conv_o (N, Cin, OH, OW), wfrag = pg_pack_conv_weight_frag(transpose = 1,
 * PG_CONV_FMT_B3), dact_src = both, r (N, Cout, OH, OW):
 *   d    = conv1x1(in) * act'(dact_src)                      (dact: PG_ACT_ELU for conv_o's own input activation)
 *   out  = d * elu'(a)  with elu(a) = dact_src - r           (gradient of conv_a's pre-activation)
 *   out2 = d * elu'(r's pre-activation)                      (gradient of r's producer's pre-activation)
 * both derivatives from the stored ELU outputs (y > 0 ? 1 : y + 1). Replaces two pg_act_bwd_from_out launches.
 * pg_conv_dual_ok() returns 1 for the shapes the bf16x3 1x1 kernel takes (<= 64 input channels of the forward convolution). */
int pg_conv_dual_ok(int Cin, int Cout, int OH, int OW);
int pg_conv2d_mfma_dual(const float* in, const float* wfrag, float* out, float* out2, int N, int Cin, int OH, int OW, int Cout,
                        const float* dact_src, int dact, const float* r, void* stream);
/* Two arithmetic back ends share this entry point; they differ in the weight-fragment FORMAT:
 *   PG_CONV_FMT_F32: v_mfma_f32_16x16x4_f32 on fp32 fragments (csrc/conv_mfma.hip);
 *   PG_CONV_FMT_B3:  every fp32 product as six v_mfma_f32_16x16x32_bf16 on exact three-way bf16
 *                    splits of both operands (csrc/conv_b3.hip; fp32-level accuracy, ~2.5x the rate).
 * pg_conv_mfma_supported returns the format to use for a (Cin -> Cout, T taps, OH x OW outputs,
 * input rows of IW, tap-list extent hr x hc) problem, or 0 (call pg_conv2d_taps instead). */
#define PG_CONV_FMT_F32 1
#define PG_CONV_FMT_B3 2
/* pack-only (forward orientation, Cout % 128 == 0): PG_CONV_FMT_B3 fragments with the output channels ordered so that each 64-channel
 * chunk holds 32 gate channels' two halves — tile m of chunk c = channels (Cout / 2) (m >> 1) + 32 c + 16 (m & 1) + 0..15. The format
 * pg_conv2d_mfma_gate reads (one wave then owns both operands of its gate channels); `out` is still written in natural order. */
#define PG_CONV_FMT_B3_GATE 3
int pg_conv_mfma_supported(int Cin, int Cout, int T, int OH, int OW, int IW, int hr, int hc);
/* floats pg_pack_conv_weight_frag writes for K_channels contracted into M_channels over T taps */
size_t pg_conv_frag_floats(int K_channels, int M_channels, int T, int fmt);
/* wfrag[chunk][g*T + t][m][lane] = Wsel[64 chunk + 16 m + (lane & 15)][4 g + (lane >> 4)][t], with
 *   transpose==0: Wsel[o][c][t] = w[o][c][tap_u[t]][tap_v[t]]  (M = Cout, K channels = Cin)
 *   transpose==1: Wsel[o][c][t] = w[c][o][tap_u[t]][tap_v[t]]  (M = Cin,  K channels = Cout)
 * zero filled outside; w is the torch layout (Cout, Cin, KH, KW). */
int pg_pack_conv_weight_frag(const float* w, float* wfrag, int Cout, int Cin, int KH, int KW,
                             int T, const int* tap_u, const int* tap_v, int transpose, int fmt,
                             void* stream);
/* both orientations of one weight in ONE launch (either output may be NULL): the forward pass packs
 * the data-gradient fragments it will need in backward at the same time */
int pg_pack_conv_weight_frag2(const float* w, float* wfrag_fwd, float* wfrag_dgrad, int Cout,
                              int Cin, int KH, int KW, int T, const int* tap_u, const int* tap_v,
                              int fmt_fwd, int fmt_dgrad, void* stream);

/* Weight + bias gradient (MFMA f32 16x16x4). The result is ADDED to dw/db (the caller zeroes
 * them once per step); per-workgroup partial sums go through `workspace` and a second,
 * deterministic reduction kernel (no atomics). Replaces aten::convolution_backward's weight/bias
 * outputs.
 * dw[co][ci][tap_u[t]][tap_v[t]] += sum_{n,r,c} dy[n,co,r,c] * act(x[n,ci,r+dr[t],c+dc[t]])
 * db[co] += sum dy  (db may be NULL).
 * NOTE the reference's weight.grad is non-zero at masked taps (mask is applied to
 * weight.data outside autograd, nn/convolution.py:42) — pass all KH*KW taps for exact parity. */
int pg_conv2d_wgrad(const float* x, const float* dy, float* dw, float* db, int N, int Cin,
                    int IH, int IW, int Cout, int OH, int OW, int KH, int KW, int T,
                    const int* tap_dr, const int* tap_dc, const int* tap_u, const int* tap_v,
                    int in_act, float* workspace, size_t workspace_floats, void* stream);
/* floats of scratch pg_conv2d_wgrad needs for a (Cout, Cin, T) problem */
size_t pg_conv2d_wgrad_workspace_floats(int Cout, int Cin, int T);

/* w *= mask in place: nn/convolution.py:42 `self.weight.data *= self.mask`. */
int pg_mul_inplace(float* w, const float* mask, size_t n, void* stream);

/* ---------------------------------------------------------------------------------------
 * NCHW LayerNorm over C.  nn/convolution.py:69-75 (permute -> nn.LayerNorm(C) -> permute).
 * eps as given (1e-5), biased variance, affine. mean/rstd: (N*L) each, saved for backward.
 * ------------------------------------------------------------------------------------- */
int pg_nchw_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                          float* mean, float* rstd, int N, int C, int L, float eps,
                          void* stream);
/* dgamma/dbeta are ADDED to (per-block partial rows through `workspace`, then a deterministic
 * reduce kernel; no atomics). */
int pg_nchw_layernorm_bwd(const float* x, const float* gamma, const float* mean,
                          const float* rstd, const float* dy, float* dx, float* dgamma,
                          float* dbeta, int N, int C, int L, float* workspace,
                          size_t workspace_floats, void* stream);
size_t pg_nchw_layernorm_bwd_workspace_floats(int N, int C, int L);
/* Same, with dx = LN backward + dx_add: the gradient of the residual (skip) branch that leaves the
 * same x — `x + f(LN(x))` in every transformer / PixelSNAIL block (image_gpt.py:50-52) — is added in
 * the kernel's epilogue instead of by a separate autograd accumulation pass. dx_add must not alias dx. */
int pg_nchw_layernorm_bwd_res(const float* x, const float* gamma, const float* mean,
                              const float* rstd, const float* dy, const float* dx_add, float* dx,
                              float* dgamma, float* dbeta, int N, int C, int L, float* workspace,
                              size_t workspace_floats, void* stream);

/* ---------------------------------------------------------------------------------------
 * Fused causal attention core.  nn/attention.py:147-160:
 *   S = q k^T / sqrt(dk); masked_fill(mask==0,-inf); softmax; masked_fill(mask==0, 0); @ v
 * with mask = tril(ones(L,L), -strict) (nn/attention.py:60-63), never materialised.
 * q,k: (N, heads*dk, L) channel-major; v,o: (N, heads*dv, L); head h owns channels
 * [h*d,(h+1)*d) (attention.py:134). *_bs = batch stride in floats (k and v are views into
 * the _kv conv output, attention.py:144). lse2: (N, heads, L) log2-domain logsumexp saved for
 * backward. A row with no allowed key (l=0, strict) yields zeros (attention.py:153-157).
 * Instantiated head dims (PG_ESHAPE otherwise): dk = dv = 4 and every (dk <= 4, dv <= 32) / (dk <= 16, dv <= 16) at
 * any L; dk in {4, 16, 32, 64} x dv in {16, 32, 64} for L % 16 == 0 with 16-byte aligned planes. The reference's
 * CausalAttention accepts any embed / head split: the Python host (ops.causal_attention) zero-pads other head dims
 * and sequence lengths to the next instantiated size — exact, see its docstring.
 * pg_causal_attn_bwd: dk = 4 with dv in {4, 16, 32} runs ONE fused kernel whose dq is summed with fp32 atomics
 * (run-to-run last-bit differences); pg_attn_fused_bwd(0) selects the bit-reproducible two-kernel backward.
 * ------------------------------------------------------------------------------------- */
int pg_causal_attn_fwd(const float* q, const float* k, const float* v, float* o, float* lse2,
                       int N, int heads, int L, int dk, int dv, long q_bs, long k_bs, long v_bs,
                       long o_bs, int strict, void* stream);
/* dq/dk/dv written (not accumulated). delta: (N, heads, L) workspace. For d_k = d_v = 4 this is ONE
 * fused launch (attention_mfma.hip attn_bwd_m44_kernel: S, dP and exp2 evaluated once per pair; delta
 * is then not written); otherwise the two launches below. */
int pg_causal_attn_bwd(const float* q, const float* k, const float* v, const float* o,
                       const float* d_o, const float* lse2, float* delta, float* dq, float* dk,
                       float* dv, int N, int heads, int L, int dk_dim, int dv_dim, long q_bs,
                       long k_bs, long v_bs, long o_bs, long do_bs, long dq_bs, long dk_bs,
                       long dv_bs, int strict, void* stream);

/* The two launches of pg_causal_attn_bwd individually (same argument list): _dq writes dq and
 * delta, _dkv reads delta and writes dk, dv. Used by bench.py to time the dominant kernel. */
int pg_causal_attn_bwd_dq(const float* q, const float* k, const float* v, const float* o,
                          const float* d_o, const float* lse2, float* delta, float* dq, float* dk,
                          float* dv, int N, int heads, int L, int dk_dim, int dv_dim, long q_bs,
                          long k_bs, long v_bs, long o_bs, long do_bs, long dq_bs, long dk_bs,
                          long dv_bs, int strict, void* stream);
int pg_causal_attn_bwd_dkv(const float* q, const float* k, const float* v, const float* o,
                           const float* d_o, const float* lse2, float* delta, float* dq, float* dk,
                           float* dv, int N, int heads, int L, int dk_dim, int dv_dim, long q_bs,
                           long k_bs, long v_bs, long o_bs, long do_bs, long dq_bs, long dk_bs,
                           long dv_bs, int strict, void* stream);

/* Process-wide switch of the fused backward: enable = 1 / 0 sets it and returns the previous value,
 * enable < 0 only queries. The fused kernel sums dQ over key blocks in a timing-dependent order (last-bit
 * run-to-run differences in dQ); pg_attn_fused_bwd(0) selects the bit-reproducible two-kernel backward.
 * Initial value: 1, or the environment's PG_ATTN_FUSED_BWD at load time. */
int pg_attn_fused_bwd(int enable);

/* (N,2,H,W) pixel-coordinate encoding, nn/attention.py:37-57 (torch.arange(-.5,.5,1/h)). */
int pg_image_positional_encoding(float* out, int N, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------
 * Elementwise pieces of the path.
 * ------------------------------------------------------------------------------------- */
int pg_act_fwd(const float* x, float* y, size_t n, int act, void* stream);
/* dx = dy * act'(x) */
int pg_act_bwd(const float* x, const float* dy, float* dx, size_t n, int act, void* stream);
/* GatedActivation: x (N,2C,L) -> y (N,C,L).  nn/convolution.py:62-66 */
int pg_gated_fwd(const float* x, float* y, int N, int C, int L, int gate, void* stream);
int pg_gated_bwd(const float* x, const float* dy, float* dx, int N, int C, int L, int gate,
                 void* stream);
/* y = res + gate(x): GatedActivation followed by the block's residual add in one pass
 * (pixel_snail.py:55-56 `x + self._activation(out)`); L % 4 == 0, 16-byte aligned tensors. */
int pg_gated_fwd_res(const float* x, const float* res, float* y, int N, int C, int L, int gate,
                     void* stream);
/* dx = dy * act'(pre) with the derivative recovered from the activation's OUTPUT v = y - res
 * (res may be NULL): backward of the fused convolution epilogue y = act(conv + bias) + res
 * (pixel_snail.py:27-28,113-119). act = PG_ACT_ELU (v > 0 ? 1 : v + 1) or PG_ACT_RELU. */
int pg_act_bwd_from_out(const float* y, const float* res, const float* dy, float* dx, size_t n,
                        int act, void* stream);
/* out = a + b */
int pg_add(const float* a, const float* b, float* out, size_t n, void* stream);
/* out[i] = value: the zero fills of gradient sinks, loss scalars and the VD-VAE decoder's constant top input
 * (torch.zeros on the reference path, e.g. vd_vae.py:379 torch.zeros_like(...)); a kernel, not a memset node
 * (a hipMemset2DAsync node of a captured step went wrong from its second replay on: profiles/README.md round 4 item 12) */
int pg_fill(float* out, float value, size_t n, void* stream);
/* out[b * per + i] = sum_k rows[k][b * batch_strides[k] + i], b < n_batch, i < per, k < n_rows <= 32 (rows, batch_strides: HOST
 * arrays copied into the launch; batch_strides == NULL: every row dense, stride per): (a) the sum over the per-block KL terms of a
 * hierarchical VAE, vd_vae.py:400 `torch.stack(kl_divs).sum(dim=0)`, in one launch; (b) the gradient of a tensor with several
 * readers (autograd's chain of `add` kernels) in one launch, where a gradient may be a channel slice of a wider tensor (dense per
 * image, batch stride > per) */
int pg_sum_rows(const float* const* rows, const long* batch_strides, int n_rows, float* out, long n_batch, long per, void* stream);
/* y[n,i] = x[n,i] + p[i] (learned positional map, image_gpt.py:86,106); i < per */
int pg_add_bcast_fwd(const float* x, const float* p, float* y, int N, size_t per, void* stream);
/* dp[i] += sum_n dy[n,i] */
int pg_add_bcast_bwd(const float* dy, float* dp, int N, size_t per, void* stream);

/* BCE-with-logits summed over pixels, averaged over batch: image_gpt.py:158-162.
 * loss[0] += (1/N) * sum_{n,i} [max(z,0) - z*x + log1p(exp(-|z|))]  (loss zeroed by caller) */
int pg_bce_logits_fwd(const float* z, const float* x, float* loss, int N, size_t per,
                      void* stream);
/* dz = gscale[0] * (sigmoid(z) - x) / N */
int pg_bce_logits_bwd(const float* z, const float* x, const float* gscale, float* dz, int N,
                      size_t per, void* stream);

/* ---------------------------------------------------------------------------------------
 * VAE pieces (models/vae/vd_vae.py:224,256; models/vae/vaes.py:17-33; vae.py:91-93,149-159).
 * ------------------------------------------------------------------------------------- */
/* 2x2 average pool, stride 2: x (planes, 2*OH, 2*OW) -> y (planes, OH, OW); planes = N*C */
int pg_avgpool2_fwd(const float* x, float* y, int planes, int OH, int OW, void* stream);
int pg_avgpool2_bwd(const float* dy, float* dx, int planes, int OH, int OW, void* stream);
/* the same with a second gradient of the pooled tensor's INPUT added in (dx = 0.25 * expand(dy) + res): the input of an
 * encoder stack's AvgPool2d is also read by the decoder's top-down blocks (vd_vae.py:224, 141-189), whose gradients arrive
 * through pass-through aliases — no gradient-sum kernel of autograd */
int pg_avgpool2_bwd_res(const float* dy, const float* res, float* dx, int planes, int OH, int OW, void* stream);
/* nearest-neighbour x2 upsample: x (planes, IH, IW) -> y (planes, 2*IH, 2*IW) */
int pg_upsample2_fwd(const float* x, float* y, int planes, int IH, int IW, void* stream);
int pg_upsample2_bwd(const float* dy, float* dx, int planes, int IH, int IW, void* stream);
/* 2x2 phase split (space-to-depth) and its inverse: x (planes, 2H, 2W) <-> xs (4, planes, H, W),
 * xs[2*pr+pc][plane][r][c] = x[plane][2r+pr][2c+pc]. merge==0 writes xs, merge==1 writes x. */
int pg_phase_split2(float* x, float* xs, int planes, int H, int W, int merge, void* stream);
/* the same between x and FOUR separate phase tensors p[2*pr+pc] (planes, H, W) (HOST array of 4 device pointers): the four phase
 * convolutions of a 4x4 / stride-2 transposed convolution (vaes.py:228-235) write their own outputs, which are interleaved here
 * without a stacked copy; merge == 0 scatters x's gradient back into four tensors */
int pg_phase_merge4(float* x, float* const* p, int planes, int H, int W, int merge, void* stream);
/* The four 2x2 phase kernels of a 4x4 / stride-2 weight, out (4, Co, Ci, 2, 2) contiguous:
 *   transposed == 0 (Conv2d weight (Co, Ci, 4, 4), vaes.py:153-160):  out[2pr+pc][o][c][i][j] = w[o][c][2i+1-pr][2j+1-pc]
 *   transposed == 1 (ConvTranspose2d weight (Ci, Co, 4, 4), vaes.py:228-235): out[2pr+pc][o][c][i][j] = w[c][o][2(1-i)+1-pr][2(1-j)+1-pc]
 * A, B = the weight's first two dimensions. pg_phase_weights_bwd is the adjoint from four gradient tensors g[k] (Co, Ci, 2, 2)
 * (HOST array of 4 device pointers, a null entry = no gradient): dw = (accumulate ? dw : 0) + scatter(g). */
int pg_phase_weights(const float* w, float* out, int A, int B, int transposed, void* stream);
int pg_phase_weights_bwd(const float* const* g, float* dw, int A, int B, int transposed, int accumulate, void* stream);
/* Fused Gaussian head. q, p: (N, >=2C, L) conv outputs holding [mean | log_std] in channels
 * [0,C) / [C,2C), read in place through batch strides q_bs / p_bs (floats). eps, z: (N, C, L).
 *   mode 0: z = mu_q + exp(s_q) eps ; kl[n] += sum KL(q || N(0,1))       (vae.py:91-93)
 *   mode 1: z = mu_q + exp(s_q) eps ; kl[n] += sum KL(q || p)            (vd_vae.py:177-186)
 *   mode 2: z = mu_p + exp(s_p) eps (sampling from the prior, no KL)     (vd_vae.py:166-168)
 * kl (N floats) is accumulated: the caller zeroes it. */
int pg_gauss_head_fwd(const float* q, const float* p, const float* eps, float* z, float* kl, int N,
                      int C, int L, long q_bs, long p_bs, int mode, void* stream);
/* dq / dp receive the gradient of the [mean | log_std] channels (written, not accumulated);
 * dz: (N,C,L) or NULL, dkl: (N) or NULL. */
int pg_gauss_head_bwd(const float* q, const float* p, const float* eps, const float* dz,
                      const float* dkl, float* dq, float* dp, int N, int C, int L, long q_bs,
                      long p_bs, int mode, void* stream);
/* out[0] += mean(v[0..n)) ; out[i] = g[0]*scale (its backward) */
int pg_vec_mean_accum(const float* v, int n, float* out, void* stream);
int pg_fill_scaled(const float* g, float scale, float* out, int n, void* stream);

/* ---------------------------------------------------------------------------------------
 * The transformer block's position-wise MLP, fused.  models/autoregressive/image_gpt.py:43-52:
 *   y = res + W2 gelu(W1 x + b1) + b2     (Conv2d(C,Hd,1) -> GELU -> Conv2d(Hd,C,1), `x + self._out(..)`)
 * x, res, y, dy, dx: (N, C, L); w1: (Hd, C); w2: (C, Hd); exact (erf) GELU. The hidden tensor is
 * never written to memory; backward recomputes it. Instantiated for C = 16, Hd = 64, L % 16 == 0
 * (PG_ESHAPE otherwise: the caller runs the three unfused operators). res may be NULL.
 * Backward: dx is written; dw1/db1/dw2/db2 are ADDED to (per-workgroup partial rows in `workspace`,
 * then a deterministic reduce); d(res) = dy is the caller's. x and dy must be 16-byte aligned.
 * ------------------------------------------------------------------------------------- */
int pg_mlp_gelu_fwd(const float* x, const float* w1, const float* b1, const float* w2,
                    const float* b2, const float* res, float* y, int N, int C, int Hd, int L,
                    void* stream);
int pg_mlp_gelu_bwd(const float* x, const float* w1, const float* b1, const float* w2,
                    const float* dy, float* dx, float* dw1, float* db1, float* dw2, float* db2,
                    int N, int C, int Hd, int L, float* workspace, size_t workspace_floats,
                    void* stream);
size_t pg_mlp_gelu_bwd_workspace_floats(int N, int L);

/* ---------------------------------------------------------------------------------------
 * Everything of an ImageGPT transformer block except the attention core, fused (gpt_block.hip).
 * models/autoregressive/image_gpt.py:21-52 and the model loop :104-109, C = 16, hidden Hd = 64:
 *   head: qkv (N,48,L) = [W_q; W_kv] LN1(x) + [b_q; b_kv]       (_ln1, _attn._q, _attn._kv)
 *   tail: x_mid = x + W_p o + b_p ; x_new = x + x_mid + W_2 gelu(W_1 LN2(x_mid) + b_1) + b_2
 *         (_attn._proj + residual, _ln2, _out, residual, and the model loop's `x = x + block(x)`)
 * Backward recomputes x_mid, the LayerNorms and the hidden activations from (x, o).
 *   tail_bwd: dx_new -> d_o (gradient of the attention output) and gx (all of dx_new that reaches x
 *             except through LN1); head_bwd: dx = LN1'(W^T dqkv) + gx.
 * Parameter gradients are ADDED to (partial rows in `workspace` + deterministic reduce).
 * Instantiated for C = 16, Hd = 64, L % 16 == 0 (PG_ESHAPE otherwise: run the unfused operators).
 * dqkv, o and dx_new must be 16-byte aligned.
 * ------------------------------------------------------------------------------------- */
int pg_gpt_block_head_fwd(const float* x, const float* ln_w, const float* ln_b, const float* wq,
                          const float* bq, const float* wkv, const float* bkv, float* qkv, int N,
                          int C, int L, float eps, void* stream);
int pg_gpt_block_head_bwd(const float* x, const float* ln_w, const float* ln_b, const float* wq,
                          const float* wkv, const float* dqkv, const float* gx, float* dx,
                          float* dln_w, float* dln_b, float* dwq, float* dbq, float* dwkv,
                          float* dbkv, int N, int C, int L, float eps, float* workspace,
                          size_t workspace_floats, void* stream);
size_t pg_gpt_block_head_bwd_workspace_floats(int N, int L);
int pg_gpt_block_tail_fwd(const float* o, const float* x, const float* wp, const float* bp,
                          const float* ln_w, const float* ln_b, const float* w1, const float* b1,
                          const float* w2, const float* b2, float* x_new, int N, int C, int Hd,
                          int L, float eps, void* stream);
int pg_gpt_block_tail_bwd(const float* o, const float* x, const float* wp, const float* bp,
                          const float* ln_w, const float* ln_b, const float* w1, const float* b1,
                          const float* w2, const float* dx_new, float* d_o, float* gx, float* dwp,
                          float* dbp, float* dln_w, float* dln_b, float* dw1, float* db1,
                          float* dw2, float* db2, int N, int C, int Hd, int L, float eps,
                          float* workspace, size_t workspace_floats, void* stream);
size_t pg_gpt_block_tail_bwd_workspace_floats(int N, int L);
/* One reduce launch per block instead of two (at the reference's batch 64 the reductions were 6 % of the
 * step): _tail_bwd_partial runs the tail kernel and leaves its partial rows in `workspace` (keep it alive);
 * _head_bwd_with_tail of the SAME block then reduces both kernels' rows in one launch. Results identical to
 * pg_gpt_block_tail_bwd + pg_gpt_block_head_bwd (same summation order). */
int pg_gpt_block_tail_bwd_partial(const float* o, const float* x, const float* wp, const float* bp,
                                  const float* ln_w, const float* ln_b, const float* w1, const float* b1,
                                  const float* w2, const float* dx_new, float* d_o, float* gx, int N, int Cc,
                                  int Hd, int L, float eps, float* workspace, size_t workspace_floats,
                                  void* stream);
int pg_gpt_block_head_bwd_with_tail(const float* x, const float* ln_w, const float* ln_b, const float* wq,
                                    const float* wkv, const float* dqkv, const float* gx, float* dx,
                                    float* dln_w, float* dln_b, float* dwq, float* dbq, float* dwkv,
                                    float* dbkv, int N, int Cc, int L, float eps, float* workspace,
                                    size_t workspace_floats, const float* tail_workspace, float* t_dw1,
                                    float* t_db1, float* t_dw2, float* t_db2, float* t_dwp, float* t_dbp,
                                    float* t_dln_w, float* t_dln_b, void* stream);
/* Round 6: the same head backward WITHOUT its reduction (partial rows stay in `workspace`), and ONE launch that adds the partial
 * rows of up to 8 blocks' head and tail kernels into their gradients at the end of the backward pass (at the reference's batch 64
 * the eight per-block reduce launches were 3.9 % of a step's kernel time). grads: n_blocks x 14 device pointers (HOST array), per
 * block dln1_w, dln1_b, dwq, dbq, dwkv, dbkv, then the tail's dw1, db1, dw2, db2, dwp, dbp, dln2_w, dln2_b. */
int pg_gpt_block_head_bwd_partial(const float* x, const float* ln_w, const float* ln_b, const float* wq, const float* wkv,
                                  const float* dqkv, const float* gx, float* dx, int N, int C, int L, float eps,
                                  float* workspace, size_t workspace_floats, void* stream);
int pg_gpt_blocks_reduce(int n_blocks, const float* const* head_ws, const float* const* tail_ws, float* const* grads,
                         int N, int C, int L, void* stream);

/* ---------------------------------------------------------------------------------------
 * Incremental autoregressive sampling (models/base.py:97-120 runs H*W full forwards; the causal
 * models need only position p per step). Activations of one position live as (channels, ld) matrices,
 * column n = sample n, so the block kernels above apply with N = 1, L = ld.
 *   pg_sample_embed: out[co*ld + n] = pixel (r, c) of Conv2d(w (already masked), b, padding k/2)
 *                    applied to canvas (N,Cin,H,W) + pos (Cin,H,W or NULL)   (image_gpt.py:105)
 *   pg_attn_decode:  qkv = rows [q (heads*dk) | k (heads*dk) | v (heads*dv)] x ld; appends k, v to
 *                    the caches (N, heads*dk, L) / (N, heads*dv, L) at column p and writes
 *                    o (heads*dv rows x ld) = softmax over positions <= p - strict   (attention.py:147-160)
 * pos_dev (device int, or NULL): when given, the raster position p (and r = p / W, c = p % W) is read
 * from it instead of the host arguments, so that one captured hipGraph can be replayed per pixel.
 * ------------------------------------------------------------------------------------- */
int pg_sample_embed(const float* canvas, const float* pos, const float* w, const float* b, float* out,
                    int N, int Cin, int H, int W, int Cout, int KH, int KW, int r, int c, int ld,
                    const int* pos_dev, void* stream);
int pg_attn_decode(const float* qkv, float* k_cache, float* v_cache, float* o, int N, int heads,
                   int L, int p, int dk, int dv, int ld, int strict, const int* pos_dev,
                   void* stream);

/* ---------------------------------------------------------------------------------------
 * Optimiser step as timed by the reference (trainer.py:183-191): global grad L2 norm
 * (clip_grad_norm_) + torch.optim.Adam over ONE flat parameter/grad buffer.
 * state (device, 8 floats): [0]=step count, [1]=lr, [2]=sum of squares (scratch),
 *   [3]=grad norm (output), [4]=clip coefficient (output), [5]=lr multiplier per step,
 *   [6]=max_norm, [7]=grad pre-scale (1/world after all-reduce)
 * ------------------------------------------------------------------------------------- */
int pg_sumsq_accum(const float* g, size_t n, float* state, void* stream);
int pg_adam_prepare(float* state, void* stream);
int pg_adam_step(float* p, const float* g, float* m, float* v, size_t n, const float* state,
                 float beta1, float beta2, float eps, void* stream);

/* ---------------------------------------------------------------------------------------
 * SURVEY.md §8(f) rank 4 — vector quantisation (reference nn/utils.py:53-96) and the MSE loss of
 * the VQ-VAE recipes (models/vae/vq_vae.py:127-136). csrc/vq.hip; reached through
 * pytorch_generative_amd/nn/utils.py (VectorQuantizer, mse_loss). Validated on MI355X against outputs
 * of the reference (tests/test_gpu_f4.py).
 * ------------------------------------------------------------------------------------- */
/* x (N, D, L) NCHW planes, embedding (K, D), D <= 64. Per position p = n*L + l: idx[p] = first
 * argmin_k (|x|^2 + |e_k|^2) - 2 x.e_k (nn/utils.py:61-68); q = embedding[idx] in NCHW;
 * st = x + (q - x) (the straight-through VALUE, :95); loss[0] += mean((x - q)^2) (:79; zeroed by caller) */
int pg_vq_assign(const float* x, const float* embedding, int* idx, float* q, float* st, float* loss,
                 int N, int D, int L, int K, void* stream);
/* EMA codebook update in place (nn/utils.py:80-90): count / sum of x per code (workspaces count_ws
 * (K), sum_ws (K*D), zeroed inside), cluster_size = cluster_size*decay + count*(1-decay), the same
 * for embedding_avg, embedding = embedding_avg / (cluster_size + 1e-5) */
int pg_vq_ema_update(const float* x, const int* idx, float* cluster_size, float* embedding_avg,
                     float* embedding, float* count_ws, float* sum_ws, int N, int D, int L, int K,
                     float decay, void* stream);
/* dx = d_st + g_loss[0] * 2 (x - q) / n  (straight-through + commitment loss) */
int pg_vq_bwd(const float* x, const float* q, const float* d_st, const float* g_loss, float* dx,
              size_t n, void* stream);
/* loss[0] += mean((a - b)^2) (zeroed by caller); da = g_loss[0] * 2 (a - b) / n, db = -da
 * (either may be NULL) */
int pg_mse_fwd(const float* a, const float* b, float* loss, size_t n, void* stream);
int pg_mse_bwd(const float* a, const float* b, const float* g_loss, float* da, float* db, size_t n,
               void* stream);

/* PixelCNN++'s concatenated ELU (Salimans et al. 2017, section 2.3; not in the reference):
 * y (N, 2C, L) = [elu(x) | elu(-x)]; CL = C * L. */
int pg_concat_elu_fwd(const float* x, float* y, int N, long CL, void* stream);
int pg_concat_elu_bwd(const float* x, const float* dy, float* dx, int N, long CL, void* stream);

/* Discretized mixture-of-logistics negative log-likelihood (the PixelCNN++ loss of BASELINE.json
 * configs[2]; absent from the reference: Salimans et al., ICLR 2017, eq. (2)-(3), restated in
 * oracle/dmol.py). l (N, 10 K, L): K logits, then per sub-pixel c = 0..2: K means, K log-scales, K raw
 * coefficients; x (N, 3, L) in [-1, 1]; K <= 16.
 *   fwd: loss[0] += -(1 / N) sum log p(x) (nats; zeroed by the caller)
 *   bwd: dl = gscale[0] * d loss / d l */
int pg_dmol_fwd(const float* l, const float* x, float* loss, int N, int K, int L, void* stream);
int pg_dmol_bwd(const float* l, const float* x, const float* gscale, float* dl, int N, int K, int L,
                void* stream);

/* dst[r * dst_stride + i] (+)= src[r * src_stride + i] for r < rows, i < row_len (strides in floats).
 * The channel concatenation in front of a merged projection (nn/attention.py:139-143
 * torch.cat((x, extra_x), dim=1): a row = the channels of one image) and the assembly / gradient split
 * of that projection's merged weight. accumulate != 0 adds into dst. */
int pg_copy_rows(const float* src, float* dst, long rows, long row_len, long src_stride,
                 long dst_stride, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------
 * Data-parallel exchange step: replaces DistributedDataParallel's gradient
 * all-reduce (reference trainer.py:78-82; process group of train.py:27-37) by ONE RCCL all-reduce of
 * the flat gradient buffer over xGMI. csrc/comm.hip binds librccl at run time (the copy the process
 * already carries, if any). One communicator per process (= per GPU).
 *   rank 0: pg_comm_unique_id(id) -> ship the 128 bytes to every rank out of band (the host side uses
 *   the torch.distributed rendezvous for that and for nothing else) -> every rank: pg_comm_init.
 * pg_allreduce_sum / pg_broadcast are in place, asynchronous, enqueue only on `stream`, do not
 * allocate or synchronise: they may be captured inside the hipGraph of the training step.
 * Return values: 0, PG_E*, or 1000 + ncclResult_t.
 * ------------------------------------------------------------------------------------- */
#define PG_COMM_ID_BYTES 128
#define PG_DTYPE_F32 0
int pg_comm_unique_id(char id[PG_COMM_ID_BYTES]);
int pg_comm_init(int rank, int world, const char id[PG_COMM_ID_BYTES]); /* on the current device */
int pg_comm_world(void);        /* world size of the live communicator, 0 if none */
int pg_comm_rccl_version(void); /* ncclGetVersion of the bound library, 0 if it cannot be bound */
int pg_allreduce_sum(void* buf, size_t n, int dtype, void* stream);
int pg_broadcast(void* buf, size_t n, int dtype, int root, void* stream);
int pg_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* PG_HIP_H_ */

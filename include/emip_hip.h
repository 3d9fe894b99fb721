/* libemip_hip.so -- C ABI of the MI355X (gfx950) kernels behind the EMIP two-stream path.
 *
 * The reference (zhangxin06/EMIP) has no native code and no FFI of its own: its hot path is
 * PyTorch ops dispatched to cuDNN/cuBLAS.  Each entry point below
 * therefore names the reference call site(s) (paths under /root/reference) whose arithmetic it
 * replaces.  INTEGRATION.md shows the ctypes binding the host side uses.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller; nothing is allocated or freed here;
 *  - `stream` is a hipStream_t passed as void* (0 = default stream); calls are asynchronous,
 *    capture-safe (no allocation, no synchronisation) and re-entrant;
 *  - activations are channels-last: an image tensor is [B][H][W][C] with an explicit channel
 *    stride (`ld*`, elements) so that callers can read/write channel slices of wider buffers;
 *  - `dtype` selects the storage type of activations and packed weights: EMIP_F32 (parity
 *    mode, exact-f32 MFMA) or EMIP_BF16 (performance mode, f32 accumulation).  Biases,
 *    normalisation parameters and statistics are always f32;
 *  - return value: EMIP_OK, or a negative code.  Arguments are validated before any launch
 *    (a faulting kernel can reset the whole GPU host), no exception crosses the ABI.
 */
#ifndef EMIP_HIP_H
#define EMIP_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define EMIP_OK 0
#define EMIP_E_INVALID (-1) /* shape / alignment / enum check failed; nothing was launched */
#define EMIP_E_LAUNCH (-2)  /* the HIP runtime rejected the launch */

#define EMIP_F32 0
#define EMIP_BF16 1

#define EMIP_ACT_NONE 0
#define EMIP_ACT_RELU 1
#define EMIP_ACT_GELU 2 /* exact erf GELU (nn.GELU default) */

int emip_version(void);

/* ---- dense contractions ------------------------------------------------------------------- */

/* C[z][m][n] = act(sum_k A[z][m][k] * W[z][n][k] + bias[n]) + R[z][m][n]
 * nn.Linear / 1x1 conv with fused bias, GELU/ReLU and residual add:
 *   lib/pvt_v2.py:45-54 (fc1, fc2), :103 (q), :110 (kv), :126 (proj), :165-167 (residual adds);
 *   model/EMIP_short/motion/gmflow/transformer.py:167-196 (q/k/v/merge/mlp; A2 = the second half
 *   of torch.cat([source, message], -1), columns K1..K-1);
 *   model/EMIP_short/motion/PromptInteract.py:380-385,413-431 (1x1 convs of the MDTA block).
 * A2 may be NULL (then K1 is ignored).  bias, R may be NULL.  batch>=1 with element strides bs*. */
int emip_gemm(const void* A, const void* A2, const void* W, void* C, const float* bias, const void* R, int M, int N,
              int K, int K1, long lda, long lda2, long ldw, long ldc, long ldr, int act, int batch, long bsA, long bsW,
              long bsC, long bsR, int dtype, void* stream);

/* emip_gemm with the LayerNorm-elimination hooks (lib/pvt_v2.py:165-167: x + attn(norm1(x)), x + mlp(norm2(x)); :108-110:
 * kv(norm(sr(x)))).  The LayerNorm in front of a Linear / conv is folded away: gamma into the weights and beta into the
 * bias at pack time (y = LN(x) W^T + b = ((x - mean) rstd) (W * gamma)^T + (b + W beta)), the per-row mean / rstd from
 *   ln_stats  f32 [M][2] = (sum x, sum x^2) over the ln_C = K channels of every A row: the operand loader feeds
 *             (x - mean) * rstd (biased variance, eps = ln_eps) to the MFMA;
 *   out_stats f32 [M][2]: (sum, sum of squares) of the rows this launch stores, accumulated with atomics -- the ln_stats
 *             of the next consumer; it must be zero beforehand, e.g. through an earlier launch's
 *   zero_ptr / zero_bytes: scratch the first workgroup of this launch clears.
 * Every hook may be NULL; emip_gemm is this function with all of them NULL. */
int emip_gemm_ln(const void* A, const void* A2, const void* W, void* C, const float* bias, const void* R, int M, int N,
                 int K, int K1, long lda, long lda2, long ldw, long ldc, long ldr, int act, int batch, long bsA, long bsW,
                 long bsC, long bsR, const float* ln_stats, int ln_C, float ln_eps, float* out_stats, void* zero_ptr,
                 long zero_bytes, int dtype, void* stream);

/* emip_gemm_ln with a caller-owned workspace for reproducible row statistics without a second pass (round 4).  A bf16 launch
 * whose output rows span more than two column tiles cannot add the tiles' (sum, sum of squares) with atomics in a fixed order;
 * without a workspace the launch is followed by a statistics pass over its output (row_stats_kernel).  With one, every column
 * tile leaves its row partials there and draws a ticket, and the tile that draws the last ticket of its row tile adds the
 * partials in column order and stores out_stats -- one launch, same bits as the pass.  stats_ws: emip_gemm_stats_ws_bytes(M, N)
 * bytes, 64-byte aligned, the first 16 384 bytes (the ticket block, one size for every shape) ZERO before the first use (every launch leaves them
 * zero: launches of different shapes may use one workspace in turn); it must
 * not be shared by launches that may run at the same time.  NULL or too small: emip_gemm_ln's behaviour.
 * (lib/pvt_v2.py:126,165-169: proj / fc2 + the statistics of the next LayerNorm) */
long emip_gemm_stats_ws_bytes(int M, int N);
int emip_gemm_ln_ws(const void* A, const void* A2, const void* W, void* C, const float* bias, const void* R, int M, int N,
                    int K, int K1, long lda, long lda2, long ldw, long ldc, long ldr, int act, int batch, long bsA, long bsW,
                    long bsC, long bsR, const float* ln_stats, int ln_C, float ln_eps, float* out_stats, void* zero_ptr,
                    long zero_bytes, int dtype, void* stats_ws, long stats_ws_bytes, void* stream);

/* emip_gemm_ln with the LayerNorm applied on the OUTPUT side of the product (lib/pvt_v2.py:139-142 norm2 -> fc1,
 * :107-110 norm -> kv):  LN(x) W^T = rstd (x W^T) - rstd mean colsum(W) (+ the folded bias), so the main loop stages the raw
 * rows (LDS-DMA path) and the normalisation costs two FMAs per output element instead of one per staged operand element
 * and N tile.  ln_stats f32 [M][2] = (sum, sum of squares) over the K channels of every A row; colsum f32 [N] = sum_k of
 * the PACKED weights (the rounded values the MFMA multiplies). */
int emip_gemm_lne(const void* A, const void* W, void* C, const float* bias, const void* R, int M, int N, int K, long lda,
                  long ldw, long ldc, long ldr, int act, const float* ln_stats, const float* colsum, float ln_eps,
                  float* out_stats, void* zero_ptr, long zero_bytes, int dtype, void* stream);

/* Strided-batched dense GEMM on the 8-wave bf16 body: C[b] = act(A[b] W[b]^T + bias) for b < batch, element strides bsA / bsW /
 * bsC between batch entries (0 = the operand is shared), bias f32 [N] shared; K % 64 == 0, strides and leading dimensions
 * multiples of 8, every operand below 2 GB.  The two per-image GEMMs of conv_corr.0 through the correlation volume's rank-128
 * factors (model/EMIP_short/model.py:59,96 with gmflow/matching.py:16-20): G[b] = W' F1[b] and out[b] = patches(F0[b]) G[b]^T. */
int emip_gemm8_batched(const void* A, const void* W, void* C, const float* bias, int M, int N, int K, long lda, long ldw, long ldc,
                       int act, int batch, long bsA, long bsW, long bsC, int cfg, void* stream);

/* The large-launch bf16 body of emip_gemm / emip_gemm_ln(e) / emip_conv2d (gemm8.hip): 8-wave workgroups, 64..256-row
 * output tiles, operands by LDS-DMA through a 2-3 stage ring, epilogue stored 16 B per lane from registers.  Same
 * arithmetic and the same hooks as above (bias, output-side LayerNorm from ln_stats + colsum, GELU / ReLU, residual R,
 * out_stats of the stored rows, zero_ptr scratch), bf16 only, one problem per launch.  Replaces the same reference call
 * sites as emip_gemm (nn.Linear: lib/pvt_v2.py:45-54,101-129; gmflow/transformer.py:128-196) and emip_conv2d
 * (nn.Conv2d: model/EMIP_short/model.py:59-62 conv_corr; gmflow/backbone.py:44-47,83,97).
 * Requirements (else EMIP_E_INVALID): K % 64 == 0 (dense), Cin % 8 == 0 (conv), lda / ldw / ldx % 8 == 0, 16-byte aligned
 * bases, every operand extent < 2 GiB.  cfg: 0 = tile chosen by emip_gemm8_auto_cfg, 1.. = explicit (calibration). */
int emip_gemm8(const void* A, const void* A2, const void* W, void* C, const float* bias, const void* R, int M, int N,
               int K, int K1, long lda, long lda2, long ldw, long ldc, long ldr, int act, const float* ln_stats,
               const float* colsum, float ln_eps, float* out_stats, void* zero_ptr, long zero_bytes, int cfg,
               void* stream);
/* emip_gemm8 with a per-sample scale on the branch: C = R + rowscale[m / rs_rows] * act(A W^T + bias) -- stochastic depth
 * (lib/pvt_v2.py:167-169 DropPath in the training forward) in the epilogue of the proj / fc2 GEMM; rowscale f32 [ceil(M / rs_rows)] */
int emip_gemm8_rs(const void* A, const void* A2, const void* W, void* C, const float* bias, const void* R, int M, int N,
               int K, int K1, long lda, long lda2, long ldw, long ldc, long ldr, int act, const float* ln_stats,
               const float* colsum, float ln_eps, float* out_stats, void* zero_ptr, long zero_bytes, const float* rowscale,
                  int rs_rows, int cfg, void* stream);
/* emip_conv8 with ln_stats != NULL: a pad-0 patch conv behind a folded LayerNorm (the spatial-reduction conv of
 * lib/pvt_v2.py:106-108 reading norm1(x)) with the LayerNorm on the OUTPUT side, tap by tap:
 *   y = sum_tap rstd_tap (W_tap . x_tap - mean_tap tapsum_tap) + bias,
 * ln_stats f32 [B*H*Wd][2] = (sum, sum of squares) over the Cin channels of every input pixel, tapsum f32 [KH*KW][Cout] =
 * channel sums of the packed weights per tap.  The operand tiles stay raw (LDS-DMA), the statistics ride the same ring. */
int emip_conv8(const void* X, const void* W, void* Y, const float* bias, const void* R, int B, int H, int Wd, int Cin,
               long ldx, int Cout, int KH, int KW, int stride, int pad, long ldy, long ldr, int act,
               const float* ln_stats, const float* tapsum, float ln_eps, float* out_stats, void* zero_ptr,
               long zero_bytes, int cfg, void* stream);
/* emip_conv8 with the statistics workspace of emip_gemm_ln_ws (below): out_stats of rows that span more than two column tiles
 * are combined inside the launch instead of by a pass behind it (lib/pvt_v2.py:106-110: sr conv -> norm -> kv) */
int emip_conv8_ws(const void* X, const void* W, void* Y, const float* bias, const void* R, int B, int H, int Wd, int Cin,
                  long ldx, int Cout, int KH, int KW, int stride, int pad, long ldy, long ldr, int act,
                  const float* ln_stats, const float* tapsum, float ln_eps, float* out_stats, void* zero_ptr,
                  long zero_bytes, int cfg, void* stats_ws, long stats_ws_bytes, void* stream);
int emip_gemm8_auto_cfg(int M, int N, int K);
/* Introspection (bench.py names the kernel symbol a launch ran on): the configuration the dispatcher inside emip_gemm* /
 * emip_conv2d* hands a bf16 launch to (0 = the 4-wave body), and a configuration's tile BM*1000+BN / ring depth. */
int emip_gemm8_dispatch(int M, int N, int K, long lda, long ldw, int K1, int has_a2, long lda2);
int emip_conv8_dispatch(int M, int Cout, int Cin, int KH, int KW, long a_elems);
int emip_gemm8_cfg_tile(int cfg);
int emip_gemm8_cfg_stages(int cfg, int lnt);

/* PVTv2 spatial-reduction attention, bf16, head_dim 64, Lk <= 128 keys (lib/pvt_v2.py:113-125: attn = softmax(q k^T * scale),
 * x = attn v, all heads of all images in one launch).  Q [batch][Lq][C] with head hd at columns hd*64, KV [batch][Lk][2C]
 * as the kv Linear leaves it (k at hd*64, v at C + hd*64), O [batch][Lq][C]; C = heads*64.  The keys of an (image, head)
 * pair stay in registers and the values in LDS while a wave streams 32-query blocks (sra.hip). */
int emip_sra_attention(const void* Q, const void* KV, void* O, int batch, int heads, int Lq, int Lk, int C, float scale,
                       void* stream);
/* ... that also leaves L[b][head][q] = log2-sum-exp of the scaled scores (f32 [batch][heads][Lq]) for emip_sra_attention_bwd */
int emip_sra_attention_lse(const void* Q, const void* KV, void* O, float* L, int batch, int heads, int Lq, int Lk, int C,
                           float scale, void* stream);

/* Backward of the spatial-reduction attention (lib/pvt_v2.py:113-125 under loss.backward(), train.py:52-58), bf16, head_dim 64,
 * Lk <= 128, in ONE launch: recomputes P from Q, K and the saved L, and produces dQ [B, Lq, C] (bf16) and dKV f32
 * [B, Lk, 2C] (dK at columns 64 h, dV at C + 64 h), ADDED into a buffer the caller has cleared
 * (the query range of an (image, head) pair may be split over workgroups).  O = the forward output, dO its gradient.
 * Replaces the unfused chain of the first rounds (three batched GEMMs, two softmax passes, two transposed GEMMs, two copies
 * per block), which materialised the [B, heads, Lq, 128] score matrices four times. */
int emip_sra_attention_bwd(const void* Q, const void* KV, const void* O, const void* dO, const float* L, void* dQ, float* dKV,
                           int batch, int heads, int Lq, int Lk, int C, float scale, void* stream);
/* The same with dK | dV stored as bf16 [B][Lk][2C] (nothing pre-cleared, no conversion): one workgroup per (image, head). */
int emip_sra_attention_bwd_bf16(const void* Q, const void* KV, const void* O, const void* dO, const float* L, void* dQ, void* dKV,
                           int batch, int heads, int Lq, int Lk, int C, float scale, void* stream);

/* The attention half of a PVTv2 block in one launch (bf16 inference; lib/pvt_v2.py:95-127 Attention.forward with sr_ratio > 1
 * and the residual add of Block.forward, :165-168):  Out = X + proj(softmax((LN(X) Wq^T) k^T scale) v).
 * X / Out [B*N, ld] tokens (Out may alias X), stats [B*N, 2] = (sum, sum of squares) of the rows of X (norm1 is applied on
 * the output side of the q projection: Wq carries gamma, bq = bias + Wq beta, colsum_q = row sums of the packed Wq), KV
 * [B, Lk, 2C] as the kv Linear leaves it, Lk <= 128, C = 64 heads in {64, 128, 320}.  Wq has bits 2 and 3 of its ROW index
 * swapped inside every 16 rows, Wp of its row AND column index (emip_amd/lib/pvt_v2.py Block._folded builds both); the
 * vectors stay in channel order.  out_stats (may be null) receives the row sums / sums of squares of Out (stored).
 * Replaces three launches (q GEMM, emip_sra_attention, proj GEMM): Q, the scores and the attention output stay on the CU. */
int emip_sra_block_eligible(int C, int Lk);
int emip_sra_block(const void* X, long ldx, const float* stats, float eps, const void* Wq, const float* bq,
                   const float* colsum_q, const void* KV, const void* Wp, const float* bp, void* Out, long ldo,
                   float* out_stats, int B, int N, int Lk, int C, float scale, void* stream);

/* The q projection and the attention of emip_sra_block without the proj half, one HEAD per workgroup (C = 320, the 22 x 22
 * stage, where emip_sra_block has too few workgroups): O = softmax((LN(X) Wq^T) k^T scale) v, bf16 [B*N, ldo]; Q never reaches
 * memory.  Arguments as emip_sra_block; O must not alias X. */
int emip_sra_qattn(const void* X, long ldx, const float* stats, float eps, const void* Wq, const float* bq,
                   const float* colsum_q, const void* KV, void* O, long ldo, int B, int N, int Lk, int C, float scale,
                   void* stream);

/* Post-norm Linear in one launch (bf16): C = R + LayerNorm(A W^T + bias) * gamma + beta with the LayerNorm over the N <= 128
 * output columns evaluated in the GEMM epilogue (GMFlow transformer.py:87-113: message = norm1(merge(message)), message =
 * norm2(mlp(...)), return source + message: R = source, may alias C; R may be null).  K % 64 == 0. */
int emip_gemm8_lno(const void* A, const void* W, void* C, const float* bias, const void* R, const float* gamma,
                   const float* beta, float eps, int M, int N, int K, long lda, long ldw, long ldc, long ldr, void* stream);

/* Introspection: block tile (BM*1000+BN) emip_gemm / emip_conv2d dispatch for an (M, N, batch, K) problem. */
int emip_gemm_tile(long M, long N, long batch, long K);


/* NHWC convolution as implicit GEMM, weights packed [Cout][KH][KW][Cin], same epilogue.
 * nn.Conv2d call sites: lib/pvt_v2.py:187-188,208 (patch embed), :75,107 (SR conv);
 *   gmflow/backbone.py:44-47,83,97 (CNN encoder); gmflow/gmflow.py:47-49 (upsampler);
 *   model/EMIP_short/model.py:59-62 (conv_corr; X is the raw correlation [B][src][tgt], which IS the
 *   NHWC view of matching.py:18-20's permuted tensor); create_backbone.py:22-36 (ConvBR, BN folded
 *   into W/bias by the host in eval mode); model/EMIP_long/LTM.py:30-41,74-79.
 * zero_ptr (may be NULL): zero_bytes of scratch cleared by the kernel's first workgroup -- the statistics buffer of the
 * InstanceNorm / BatchNorm that follows, so that emip_chan_stats(prezeroed = 1) needs no separate zero-fill launch. */
int emip_conv2d(const void* X, const void* W, void* Y, const float* bias, const void* R, int B, int H, int Wd, int Cin,
                long ldx, int Cout, int KH, int KW, int stride, int pad, long ldy, long ldr, int act, void* zero_ptr,
                long zero_bytes, int dtype, void* stream);

/* emip_conv2d with the same hooks: ln_stats f32 [B*H*W][2] over the Cin channels of every INPUT pixel (zero padding stays
 * zero), out_stats f32 [B*Ho*Wo][2] of the stored output pixels. */
int emip_conv2d_ln(const void* X, const void* W, void* Y, const float* bias, const void* R, int B, int H, int Wd, int Cin,
                   long ldx, int Cout, int KH, int KW, int stride, int pad, long ldy, long ldr, int act, void* zero_ptr,
                   long zero_bytes, const float* ln_stats, float ln_eps, float* out_stats, int dtype, void* stream);

/* Split-K variant for small-M / long-K convs (the 121-token spatial-reduction convs of lib/pvt_v2.py:75,107: 16-80 output
 * tiles walking K = 1280-4096): ksplit workgroups share an output tile, each takes 1/ksplit of the K tiles and adds its
 * partial sums (bias with split 0) into acc_out f32 [B*Ho*Wo][ldacc] (zero beforehand) with atomics; Y / R / act /
 * out_stats are unused then.  emip_rows_finalize casts the complete rows to the storage type and takes their statistics. */
int emip_conv2d_splitk(const void* X, const void* W, void* Y, const float* bias, const void* R, int B, int H, int Wd,
                       int Cin, long ldx, int Cout, int KH, int KW, int stride, int pad, long ldy, long ldr, int act,
                       void* zero_ptr, long zero_bytes, const float* ln_stats, float ln_eps, float* out_stats,
                       float* acc_out, long ldacc, int ksplit, int dtype, void* stream);

/* emip_conv2d / emip_conv2d_ln with split-K reduced INSIDE the launch (no finalize launch): ksplit workgroups share every
 * 64 x 64 output tile, the last to arrive at the tile's ticket takes the sums back and runs the normal epilogue (bias, act,
 * storage type, row statistics).  acc: f32 [B*Ho*Wo][Cout], ticket: u32 [ceil(M/64) * ceil(Cout/64)], both ZERO on entry
 * and left zero.  ln_stats may be NULL (plain conv: the decoder-side reductions, create_backbone.py:199-208) or the row
 * statistics of the input pixels (the 8 x 8 / 4 x 4 spatial-reduction convs, lib/pvt_v2.py:106-108). */
int emip_conv2d_ksplit(const void* X, const void* W, void* Y, const float* bias, int B, int H, int Wd, int Cin, long ldx,
                       int Cout, int KH, int KW, int stride, int pad, long ldy, int act, const float* ln_stats, float ln_eps,
                       float* out_stats, float* acc, void* ticket, int ksplit, int dtype, void* stream);
int emip_rows_finalize(const float* A, long lda, void* Y, long ldy, float* out_stats, long M, int C, int dtype,
                       void* stream);

/* Two convs with the normalising loader in ONE launch (64x64 tiles; a 1x1 conv is a Linear over the tokens).  In the PVT
 * block the q projection and the spatial-reduction conv read the same normalised tokens and are independent
 * (lib/pvt_v2.py:103-110): the sr conv alone is 16-80 workgroups walking a long K, beside the q tiles it is hidden.
 * Neither problem may clear or produce what the other one consumes. */
typedef struct emip_conv_desc {
    const void* X; const void* W; void* Y; const float* bias; const void* R;
    int B, H, Wd, Cin; long ldx; int Cout, KH, KW, stride, pad; long ldy, ldr; int act;
    const float* ln_stats; float ln_eps; float* out_stats;
    /* fused split-K of the SECOND problem (ksplit > 1): acc f32 [B*Ho*Wo][Cout] and ticket u32 [output tiles of 64x64], both
     * zero before the first launch and left zero by every launch; the last split to arrive runs the epilogue */
    float* acc; unsigned* ticket; int ksplit;
    /* FIRST problem, 1x1 conv only: f32 [Cout] column sums of the packed weights -> it runs as a dense GEMM over the raw rows
     * with the LayerNorm on the output side (see emip_gemm_lne) */
    const float* colsum;
} emip_conv_desc;
int emip_conv2d_pair(const void* desc_a, const void* desc_b, int dtype, void* stream);   /* -> const emip_conv_desc* */

/* Fused attention  O = softmax(Q K^T * scale + mask) V  (online softmax, scores never stored unless S!=NULL).
 * Replaces: lib/pvt_v2.py:121-125 (SRA, D=DV=64, Lk=121); gmflow/transformer.py:46-105 (split-window
 *   attention; q_rows/k_rows [nwin][L] int32 tables fold torch.roll + split_feature + merge_splits into
 *   addressing, q_gid/k_gid [nwin][L] region ids reproduce generate_shift_window_attn_mask :19-43 as an
 *   additive -100); gmflow/matching.py:13-36 (S = raw correlation out, V = pixel grid -> correspondence);
 *   gmflow/transformer.py:526-531 (flow propagation); model/EMIP_long/LTM.py:57-65 (memory read).
 * (D,DV) in {(64,64),(128,128),(128,32)}.  Row r of batch b, head hd: X + b*x_bs + hd*x_hs + r*ldx.
 * The batch index of the launch is b*nwin + win.  O is T, or f32 when o_f32!=0.  S (optional, heads==1,
 * no q_rows): S[(b*nwin+win)*s_bs + q*lds + k] = scale * <Q_q, K_k> as T. */
int emip_attention(const void* Q, const void* K, const void* V, void* O, void* S, int batch, int heads, int nwin,
                   int Lq, int Lk, int D, int DV, long q_bs, long k_bs, long v_bs, long o_bs, long s_bs, long ldq,
                   long ldk, long ldv, long ldo, long lds, long q_hs, long k_hs, long v_hs, long o_hs,
                   const int* q_rows, const int* k_rows, const int* q_gid, const int* k_gid, float scale, int o_f32,
                   int dtype, void* stream);
/* emip_attention with the key range split over ksplit workgroups per query tile (flash-decoding style) and a small merge
 * launch: for long key sets on small grids (global matching / flow propagation, matching.py:8-41, transformer.py:503-533:
 * 1936 keys; the EMIP-long memory read, LTM.py:49-68: up to 9680) one workgroup otherwise walks every key tile serially.
 * ws: f32 [batch*nwin][heads][ksplit][Lq][DV + 2] scratch (16-byte aligned). */
int emip_attention_splitkv(const void* Q, const void* K, const void* V, void* O, void* S, int batch, int heads, int nwin,
                           int Lq, int Lk, int D, int DV, long q_bs, long k_bs, long v_bs, long o_bs, long s_bs, long ldq,
                           long ldk, long ldv, long ldo, long lds, long q_hs, long k_hs, long v_hs, long o_hs,
                           const int* q_rows, const int* k_rows, const int* q_gid, const int* k_gid, float scale,
                           int o_f32, int ksplit, float* ws, int dtype, void* stream);

/* emip_attention_splitkv with the keys / values of batch element b read from element (b + kv_batch_rot) mod batch: the cross
 * attention of gmflow/transformer.py:281-301 (source = the other frame of the pair) on a batch that holds both frames, without
 * copying or re-projecting the swapped halves. */
int emip_attention_rot(const void* Q, const void* K, const void* V, void* O, void* S, int batch, int heads, int nwin, int Lq,
                       int Lk, int D, int DV, long q_bs, long k_bs, long v_bs, long o_bs, long s_bs, long ldq, long ldk,
                       long ldv, long ldo, long lds, long q_hs, long k_hs, long v_hs, long o_hs, const int* q_rows,
                       const int* k_rows, const int* q_gid, const int* k_gid, float scale, int o_f32, int ksplit, float* ws,
                       int kv_batch_rot, int dtype, void* stream);

/* GMFlow global matching (gmflow/matching.py:8-41) and flow propagation (gmflow/transformer.py:503-533) on one kernel built
 * for the shape (bf16, D = 128, two value columns): Out[z][q] = sum_k softmax_k(scale <Q[z][q], K[zk][k]>) v[k] (- the query's
 * own pixel when sub_grid), zk = (z + kv_rot) mod Z.  Q, K: bf16 [Z][n][128] with row strides ldq / ldk and batch strides
 * q_bs / k_bs (elements); V: f32 [Z][n][2] indexed like the keys, or NULL = the pixel grid (x = k mod W, y = k / W:
 * geometry.py:5-21); S: bf16 [Zs][n][n] receives the raw correlation scale * q.k of the batches z < Zs, row q, column k -- the
 * volume matching.py:18-20 returns (permuted) and model.py:96 hands to conv_corr; Out: f32 [Z][n][2].  Both matching
 * directions are ONE launch (Z = 2B, kv_rot = B, Zs = B).  128 <= n <= 2048, n % 8 == 0. */
int emip_match(const void* Q, const void* K, const float* V, void* S, float* Out, int Z, int Zs, int n, int W, long ldq,
               long ldk, long q_bs, long k_bs, int kv_rot, float scale, int sub_grid, float* lse, void* stream);

/* Backward of emip_match (train.py:52-58 through matching.py:8-41 / transformer.py:485-533; no parameters: token gradients
 * only).  Out / dOut: f32 [Z][n][2] the forward's output and its gradient; lse: f32 [Z][n] the log2-sum-exp emip_match wrote;
 * dS: bf16 [Zs][n][n] gradient w.r.t. the scaled scores S of batches z < Zs (the correlation volume conv_corr consumed) or NULL;
 * stat: f32 [Z][n][4] workspace; dQ, dK: bf16 [Z][n][128]; the gradient of the keys batch z read is written to batch
 * (z + kv_rot) mod Z; accum_dk != 0 ADDS it to those rows (dK == dQ when Q == K: the token gradient of matching.py:13-14). */
int emip_match_bwd(const void* Q, const void* K, const float* V, const float* Out, const float* dOut, const float* lse,
                   const void* dS, float* stat, void* dQ, void* dK, int Z, int Zs, int n, int W, long ldq, long ldk, long q_bs,
                   long k_bs, int kv_rot, float scale, int sub_grid, int accum_dk, void* stream);

/* GMFlow split-window attention (gmflow/transformer.py:46-105, masks :19-43) on a kernel built for the shape (bf16, one head,
 * D = DV = 128, windows of L <= 512 tokens): O[rows[win][q]] = softmax_k(scale <Q[rows[win][q]], K[rows[win][k]]> - 100
 * [gid[win][q] != gid[win][k]]) V[rows[win][k]].  Q, K, V, O: token matrices of B frames of `tokens` tokens (row strides
 * ld*, batch strides *_bs, in elements); rows / gid: int [nwin][L] (gid NULL: no mask); keys / values of frame b come from
 * frame (b + kv_rot) mod B (cross attention between the two frames of a pair). */
int emip_window_attention(const void* Q, const void* K, const void* V, void* O, int B, int nwin, int L, long ldq, long ldk,
                          long ldv, long ldo, long q_bs, long k_bs, long v_bs, long o_bs, const int* rows, const int* gid,
                          int tokens, int kv_rot, float scale, float* lse, void* stream);

/* emip_window_attention followed IN THE LAUNCH by the layer's merge Linear, norm1 and the residual (gmflow/transformer.py:330-338):
 * O = Res + LayerNorm(attention Wm^T) * gamma + beta; the attention output never reaches memory.  Wm: merge.weight [128][128] in
 * MFMA-fragment order (emip_amd.ops.wattn_merge_pack); Res NULL or the residual tokens (may be O itself).  Wq: NULL, or q_proj.weight
 * in fragment order (emip_amd.ops.wattn_q_pack): Q then points at the token rows and q = tokens Wq^T is computed in the prologue. */
int emip_window_attention_merge(const void* Q, const void* K, const void* V, void* O, int B, int nwin, int L, long ldq, long ldk,
                                long ldv, long ldo, long q_bs, long k_bs, long v_bs, long o_bs, const int* rows, const int* gid,
                                int tokens, int kv_rot, float scale, const void* Wm, const float* gamma, const float* beta,
                                float eps, const void* Res, long ldr, long r_bs, const void* Wq, void* stream);

/* Backward of emip_window_attention in three launches (train.py:52-58 through transformer.py:46-105; GMFlow is frozen,
 * train.py:340-342, only the token gradients exist): delta = rowsum(dO o O); dQ with the queries stationary; dK, dV with the keys
 * stationary; P is recomputed from lse (the forward's log2-sum-exp, f32 [B][tokens]).  O, dO and the three gradients are
 * contiguous [B][tokens][128] bf16; Q, K, V may be column slices (ld*, *_bs in elements); delta: f32 [B][tokens] workspace.
 * Gradients of the keys / values of frame (b + kv_rot) mod B are written to that frame's rows.  nwin * L == tokens. */
int emip_window_attention_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse,
                              float* delta, void* dQ, void* dK, void* dV, int B, int nwin, int L, long ldq, long ldk, long ldv,
                              long q_bs, long k_bs, long v_bs, const int* rows, const int* gid, int tokens, int kv_rot,
                              float scale, void* stream);

/* GMFlow transformer FFN in one launch (gmflow/transformer.py:316-345: mlp = Linear(2C, 8C) -> GELU -> Linear(8C, C), both without
 * bias, norm2, `source + message`; C = 128): Out = Res + LayerNorm(GELU([X1 | X2] W0^T) W2^T) * gamma + beta, the [M][1024] hidden
 * tensor never leaves the CU.  W0p / W2p: mlp[0].weight [1024][256] and mlp[2].weight [128][1024] packed in MFMA-fragment order
 * (emip_amd.ops.ffn_block_packs).  Res may be NULL or alias Out; X1 may alias Out (a token's row is read and written by one lane). */
int emip_ffn_block(const void* X1, long ld1, const void* X2, long ld2, const void* W0p, const void* W2p, const float* gamma,
                   const float* beta, float eps, const void* Res, long ldr, void* Out, long ldo, long M, void* stream);

/* MDTA channel attention matrix: L2-normalise q,k over pixels, 64x64 Gram per head, * temperature, softmax.
 * PromptInteract.py:423-428.  ws: f32 scratch of emip_mdta_ws_floats(B, heads) floats -- the complete sums [G | nq | nk] per
 * (image, head) first (B*heads*(4096+128) floats, what the backward reads), then the per-workgroup partial sums the softmax
 * kernel adds in slot order (no atomics: reproducible bit for bit); attn out: T [B][heads][64][64]. */
int emip_mdta_ws_floats(int B, int heads);
int emip_mdta_attn(const void* Q, long ldq, long q_bs, const void* K, long ldk, long k_bs, const float* temperature,
                   float* ws, void* attn, int B, int heads, int P, int dtype, void* stream);

/* ---- normalisations ------------------------------------------------------------------------ */

/* Row LayerNorm over C channels (biased variance): Y = LN(X) [+ R].  nn.LayerNorm call sites lib/pvt_v2.py:137,144,189,
 * 78,302 (eps 1e-6 / 1e-5), gmflow/transformer.py:134,145 (R = the residual stream of `source + message`, :196),
 * PromptInteract.py:346-349 (WithBias_LayerNorm).  R may be NULL and may alias Y. */
int emip_layernorm(const void* X, long ldx, void* Y, long ldy, const float* gamma, const float* beta, const void* R,
                   long ldr, float* out_stats, long M, int C, float eps, int dtype, void* stream);
/* out_stats (may be NULL): f32 [M][2] = (sum, sum of squares) of every stored output row -- the input of a following
 * LayerNorm that is folded into its consumer (emip_gemm_ln / emip_conv2d_ln). */

/* Per-(group, channel) sum / sum-of-squares over `rows` rows -> sums f64 [groups][C][2] (zeroed inside unless
 * prezeroed != 0: the producing emip_conv2d cleared it through its zero_ptr).
 * groups = images for nn.InstanceNorm2d (gmflow/backbone.py:40,50-53), 1 for train-mode BatchNorm2d
 * (create_backbone.py:28, model.py:60). */
int emip_chan_stats(const void* X, long ldx, double* sums, long groups, long rows, int C, int prezeroed, int dtype,
                    void* stream);

/* y = [relu]( R + [relu]( (x-mean)*rstd [*gamma+beta] ) ) using emip_chan_stats output.
 * gmflow/backbone.py:61-69 (ResidualBlock tail) and train-mode ConvBR. */
int emip_chan_norm_apply(const void* X, long ldx, void* Y, long ldy, const void* R, long ldr, const double* sums,
                         const float* gamma, const float* beta, long groups, long rows, int C, float eps,
                         int relu_inner, int relu_outer, int dtype, void* stream);
/* emip_chan_norm_apply whose residual R is itself a RAW convolution output: with res_sums f64 [groups][C][2] the residual is
 * normalised on the fly from its own sums (+ ReLU with res_relu) and rounded to the storage type as the stored tensor would
 * have been -- the InstanceNorm pass that would have materialised it disappears (gmflow/backbone.py:39-69: the stem's norm1 +
 * relu feeding layer1's skip connection, the downsample branch's norm).  res_sums NULL: emip_chan_norm_apply. */
int emip_chan_norm_apply_res(const void* X, long ldx, void* Y, long ldy, const void* R, long ldr, const double* sums,
                             const float* gamma, const float* beta, long groups, long rows, int C, float eps, int relu_inner,
                             int relu_outer, const double* res_sums, int res_relu, int dtype, void* stream);

/* ---- depthwise / resampling / elementwise ---------------------------------------------------- */

/* Depthwise 3x3 (stride 1, pad 1), weights f32 [9][C].  lib/pvt_v2.py:316-327 (+GELU of :50 fused via act);
 * PromptInteract.py:402-404 (q_dwconv, kv_dwconv). */
int emip_dwconv3x3(const void* X, long ldx, void* Y, long ldy, const float* Wt, const float* bias, int B, int H,
                   int Wd, int C, int act, int dtype, void* stream);
/* same, also storing the pre-activation values Z (input of the GELU backward) from the same pass */
int emip_dwconv3x3_dual(const void* X, long ldx, void* Y, long ldy, void* Z, long ldz, const float* Wt, const float* bias,
                        int B, int H, int Wd, int C, int act, int dtype, void* stream);

/* Gated depthwise: X has C2 = 2*Ch channels, Y[c] = gelu(dw(X)[c]) * dw(X)[Ch+c], zero for Ch <= c < Cout_pad.
 * PromptInteract.py:380-384 (GDFN). */
int emip_dwconv3x3_gated(const void* X, long ldx, void* Y, long ldy, const float* Wt, const float* bias, int B, int H,
                         int Wd, int C2, int Cout_pad, int dtype, void* stream);

/* Bilinear resize, channels-last in/out, result scaled by mul.  nn.Upsample(x2, align_corners=True)
 * create_backbone.py:49. */
int emip_bilinear(const void* X, long ldx, void* Y, long ldy, int B, int H, int Wd, int C, int Ho, int Wo,
                  int align_corners, float mul, int dtype, void* stream);

/* Bilinear resize of channels xc..xc+C-1 of a channels-last tensor into a planar f32 [B][C][Ho][Wo]:
 * F.interpolate(pc, scale_factor=8) create_backbone.py:75 (align_corners=0); gmflow.py:58-60 (x8, align 1, mul 8). */
int emip_bilinear_planar(const void* X, long ldx, int xc, float* Y, int B, int H, int Wd, int C, int Ho, int Wo,
                         int align_corners, float mul, int dtype, void* stream);

/* Y = A*B (mode 0), A*B*C3 (1), A+B (2), A + B[row % period] (3) over [M][C].
 * create_backbone.py:63-64 (NCD products); gmflow/utils.py:74-75 (position add); LTM.py:39 (fea + corr). */
int emip_eltwise(const void* A, long lda, const void* Bp, long ldb, const void* C3, long ldc3, void* Y, long ldy,
                 long M, int C, int mode, long period, int dtype, void* stream);

/* planar f32 [B][C][P] -> channels-last T [B][P][ldy] (channels C..Cpad-1 zero) and back. */
int emip_planar_to_cl(const float* X, void* Y, long ldy, int B, int C, long P, int Cpad, int dtype, void* stream);
int emip_cl_to_planar(const void* X, long ldx, int xc, float* Y, int B, int C, long P, int dtype, void* stream);

/* Y[m][yc+c] = c < C ? X[m][xc+c] : 0 for c < Cpad, converting x_dtype -> y_dtype (torch.cat pieces). */
int emip_copy_cols(const void* X, long ldx, int xc, int x_dtype, void* Y, long ldy, int yc, int y_dtype, long M,
                   int C, int Cpad, void* stream);

/* Convex x8 flow upsampling: softmax over 9 logits, weighted 3x3 neighbourhood of 8*flow.  gmflow.py:64-77.
 * logits T [N][H][W][>=576], flow f32 [N][H][W][2], out f32 planar [N][2][8H][8W]. */
int emip_convex_upsample(const void* logits, long ldl, const float* flow, float* out, int N, int H, int Wd, int dtype,
                         void* stream);

/* flow[n][p][0:2] = O[n][p][0:2] - (sub_grid ? pixel (x,y) of p : 0).  gmflow/matching.py:36-39. */
int emip_corresp_to_flow(const float* O, long ldo, float* flow, long N, int H, int Wd, int sub_grid, void* stream);

/* ---- flow-side loss (planar f32) --------------------------------------------------------------- */

/* Bilinear warp, border padding, align_corners=True.  loss/warp_utils.py:83-93. */
int emip_flow_warp(const float* X, const float* flow, float* Y, int B, int C, int H, int W, void* stream);

/* Corner indices int64 [B][4][H*W] and weights f32 [B][4][H*W] of get_corresponding_map, reference corner order.
 * loss/warp_utils.py:43-70.  Indices are bit-exact with the reference. */
int emip_occ_corners(const float* flow, long long* indices, float* weights, int B, int H, int W, void* stream);

/* get_occu_mask_backward: occ = clamp(scatter_add(weights), 0, 1) < th (complement != 0: 1 - occ, the
 * non-occluded mask loss_flow.py:95-96 uses).  loss/warp_utils.py:72-80,106-112. */
int emip_occ_mask_backward(const float* flow, float* cmap_ws, float* occ, int B, int H, int W, float th,
                           int complement, void* stream);

/* hybrid_e_loss = mean_b(BCE-with-logits mean + E-loss_b + soft-IoU_b).  loss/loss_pred.py:4-22.
 * pred (logits), mask: f32 [B][1][H][W]; ws: f64 [B][8] scratch; out: f32 [1]. */
int emip_hybrid_e_loss(const float* pred, const float* mask, double* ws, float* out, int B, int H, int W,
                       void* stream);

/* out[0] (+)= weight * (0.15*mean(|im-rec|*m) + 0.85*mean(SSIM3x3(rec*m, im*m))) / mean(m).
 * loss/loss_flow.py:35-49 + loss/loss_blocks.py:46-65.  im, rec: f32 [B][C][H][W]; mask f32 [B][1][H][W]; ws f64 [4]. */
int emip_photometric_loss(const float* im, const float* rec, const float* mask, double* ws, float* out, int B, int C,
                          int H, int W, float weight, int accumulate, void* stream);

/* ---- backward kernels (training step, train.py:43-62: loss.backward() through the modules above) ------------ */

/* Weight gradient of nn.Linear / 1x1 conv (every Linear of lib/pvt_v2.py, PromptInteract.py, create_backbone.py:119):
 * C[n][k] = sum_m A[m][n] * B[m][k]  (A = dY [M][N], B = X [M][K], both
 * as the forward leaves them in HBM), f32 output; M is split over workgroups and combined with f32 atomics. */
int emip_gemm_tn(const void* A, const void* B, float* C, long M, int N, int K, long lda, long ldb, long ldc, int batch,
                 long bsA, long bsB, long bsC, int dtype, void* stream);

/* emip_gemm_tn that also produces the bias gradient db[n] = sum_m A[m][n] (f32 [batch][N]; cleared by this call, in the
 * same zero launch as C when db == C + batch*N*K) from the dY tiles it stages anyway. */
int emip_gemm_tn_bias(const void* A, const void* B, float* C, float* db, long M, int N, int K, long lda, long ldb, long ldc,
                      int batch, long bsA, long bsB, long bsC, int dtype, void* stream);

/* Weight gradient of an NHWC conv, packed like the forward weights: dW[co][ky][kx][ci] (f32) =
 * sum over output pixels of dY[pix][co] * X[pix shifted by the tap][ci]  (zero padding honoured). */
int emip_conv2d_wgrad(const void* dY, const void* X, float* dW, int B, int H, int Wd, int Cin, long ldx, int Cout,
                      long lddy, int KH, int KW, int stride, int pad, int dtype, void* stream);

/* The two weight-gradient contractions ADDED (f32 atomics) into C / db / dW that the caller has already cleared: the
 * training step (train.py:52-58 loss.backward()) clears ONE arena holding every parameter gradient with a single launch
 * instead of one zero launch per weight.  db may be NULL. */
int emip_gemm_tn_into(const void* A, const void* B, float* C, float* db, long M, int N, int K, long lda, long ldb, long ldc,
                      int dtype, void* stream);
int emip_conv2d_wgrad_into(const void* dY, const void* X, float* dW, int B, int H, int Wd, int Cin, long ldx, int Cout,
                           long lddy, int KH, int KW, int stride, int pad, int dtype, void* stream);

/* PVTv2 Mlp head in ONE launch, bf16 inference (lib/pvt_v2.py:45-54: fc1 -> DWConv -> GELU, norm2 of :165-169 folded into
 * W1 / b1 and applied on the output side from ln_stats [rows][2] = (sum, sum of squares) of the token rows, colsum [N] = row
 * sums of the packed W1): G[rows][ldg] = GELU(dwconv3x3(LN(X) W1^T + b1) + bd) for B images of H x Wd <= 512 tokens; the
 * hidden tensor between fc1 and the depthwise conv never reaches HBM.  K % 32 == 0, N % 64 == 0; Wdw f32 [9][N]. */
int emip_mlp_fc1dw(const void* X, long ldx, const void* W1, const float* b1, const float* colsum, const float* ln_stats,
                   float eps, const float* Wdw, const float* bd, void* G, long ldg, int B, int H, int Wd, int K, int N,
                   void* stream);

/* The whole Mlp half of a PVTv2 block in one launch (bf16 inference, the 22 x 22 stage: C = 320, hidden N = 1280):
 * Out = X + fc2(GELU(dwconv3x3(LN(X) W1^T + b1) + bd)) + b2 per image (lib/pvt_v2.py:45-54,165-169,316-327), out_stats =
 * (sum, sum of squares) of the rows of Out for the next block's LayerNorm (may be NULL).  The hidden tensor never leaves
 * the CU: a workgroup owns a band of image rows (+ one halo row each side) for all hidden channels.  X, Out: bf16
 * [B, H, W, C], Out must NOT overlap X (a band's halo rows are another band's outputs).  W1: bf16 [N][C] with the norm's
 * scale folded in; W2: bf16 [C][N]; cst: f32 [N / 64][12][64] = per 64-channel chunk of the hidden tensor its 9 depthwise
 * taps, the depthwise bias, fc1's bias (+ W1 beta) and the row sums of the packed W1; b2: f32 [C]; ln_stats: f32 [B H W][2]
 * (sum, sum of squares) of the rows of X. */
int emip_mlp_block_eligible(int B, int H, int W, int C, int N);
int emip_mlp_block(const void* X, long ldx, const void* W1, const void* W2, const float* cst, const float* b2,
                   const float* ln_stats, float eps, void* Out, long ldo, float* out_stats, int B, int H, int W, int C,
                   int N, void* stream);
int emip_mlp_fc1dw_eligible(int B, int H, int Wd, int K, int N);
/* Maps of more than 512 tokens go in row bands with one halo row on each side (the fc1 result of the halo rows is recomputed):
 * output rows per band, 0 = not eligible.  emip_mlp_fc1dw accepts both forms. */
int emip_mlp_fc1dw_band_rows(int B, int H, int Wd, int K, int N);

/* The large-launch body of the weight gradients (bf16, dense operands, M >= 2048): the same contraction as emip_gemm_tn on a
 * ring of LDS-DMA stages (gemm_tn8.hip).  emip_gemm_tn / _bias / _into hand it every eligible launch; this is the explicit
 * entry.  prezeroed != 0: C / db are clear already and are added into.  db may be NULL. */
int emip_gemm_tn8(const void* A, const void* B, float* C, float* db, long M, int N, int K, long lda, long ldb, long ldc,
                  int prezeroed, void* stream);
int emip_gemm_tn8_eligible(long M, int N, int K, long lda, long ldb);

/* emip_conv2d_wgrad on the same ring (bf16; the im2col rows of X are gathered by per-lane LDS-DMA offsets, the pixel a lane
 * fetches tracked incrementally from stage to stage); emip_conv2d_wgrad / _into hand it every eligible launch. */
int emip_conv_wgrad8(const void* dY, const void* X, float* dW, int B, int H, int Wd, int Cin, long ldx, int Cout, long lddy,
                     int KH, int KW, int stride, int pad, int prezeroed, void* stream);
int emip_conv_wgrad8_eligible(int B, int H, int Wd, int Cin, long ldx, int Cout, long lddy, int KH, int KW, int stride,
                              int pad);

/* Grouped form: the weight gradients of many layers in ONE persistent launch (the training step defers them and flushes them
 * together, emip_amd/ops.py WgradQueue).  emip_gemm_tn8_group_plan fills one record (HOST memory,
 * emip_gemm_tn8_group_recsize() bytes) for C += A^T B (+ db += column sums of A) into PRE-CLEARED outputs and returns the
 * problem's work-item count (a multiple of 8; negative = error); item0 = the items of the records before it.
 * emip_gemm_tn8_group takes the records as a DEVICE array. */
int emip_gemm_tn8_group_recsize(void);
int emip_gemm_tn8_group_plan(void* rec, const void* A, const void* B, float* C, float* db, long M, int N, int K, long lda,
                             long ldb, long ldc, int item0);
int emip_gemm_tn8_group(const void* probs, int nprob, int total, void* stream);

/* The same grouped launch on 256 x 320 / 320 x 256 output tiles (eight waves, gemm_tn16.hip) for the weight gradients whose
 * shape fills such tiles -- PVTv2-b5's third stage (C = 320: 40 of 52 blocks) and conv_corr -- Linear AND convolution
 * (reference: train.py:52-58 loss.backward() through lib/pvt_v2.py:45-54,101-129, model/EMIP_short/model.py conv_corr).
 * emip_gemm_tn16_eligible: 0 = leave the contraction to the 128 x 128 tiles above, else 1 + tile orientation.  The plan
 * functions fill one HOST record (emip_gemm_tn16_recsize() bytes) for an accumulation into PRE-CLEARED outputs and return the
 * problem's work-item count (a multiple of 8; negative = error); splits = m ranges (0: by length).  The convolution form
 * reads X [B,H,W,Cin] through its im2col view: dW[co][ky][kx][ci] += sum_pixels dY[pix][co] X[pix + tap][ci]; db (may be
 * NULL) += sum_pixels dY[pix][co], the bias gradient. */
int emip_gemm_tn16_eligible(long M, int N, int K, long lda, long ldb);
int emip_conv_wgrad16_eligible(int B, int H, int Wd, int Cin, long ldx, int Cout, long lddy, int KH, int KW, int stride,
                               int pad);
int emip_gemm_tn16_recsize(void);
int emip_gemm_tn16_plan(void* rec, const void* A, const void* B, float* C, float* db, long M, int N, int K, long lda,
                        long ldb, long ldc, int item0, int splits);
int emip_conv_wgrad16_plan(void* rec, const void* dY, const void* X, float* dW, float* db, int B, int H, int Wd, int Cin,
                           long ldx, int Cout, long lddy, int KH, int KW, int stride, int pad, int item0, int splits);
int emip_gemm_tn16_group(const void* probs, int nprob, int total, void* stream);

/* LayerNorm backward: dx, and dgamma/dbeta ACCUMULATED (f32 atomics).  mean/rstd are recomputed from x.  Workgroup b
 * accumulates into dgamma/dbeta + (b % nparts) * part_stride: nparts > 1 spreads the atomics over partial buffers that
 * the caller sums (nparts = 1: plain accumulation). */
int emip_layernorm_bwd(const void* X, long ldx, const void* DY, long lddy, void* DX, long lddx, const float* gamma,
                       float* dgamma, float* dbeta, int nparts, long part_stride, long M, int C, float eps, int dtype,
                       void* stream);
/* The same with DX = LayerNorm backward + DR (same dtype as DX, row stride lddr; NULL = plain): in the pre-norm residual
 * blocks of lib/pvt_v2.py:165-169 x feeds norm1 / norm2 AND the skip connection, so the skip path's gradient is added in
 * this kernel's store instead of by a separate add launch (torch autograd's accumulation). */
int emip_layernorm_bwd_res(const void* X, long ldx, const void* DY, long lddy, void* DX, long lddx, const void* DR, long lddr,
                           const float* gamma, float* dgamma, float* dbeta, int nparts, long part_stride, long M, int C,
                           float eps, int dtype, void* stream);

/* Backward of the softmax attentions (lib/pvt_v2.py:113-121; gmflow/transformer.py:46-105,503-533; matching.py:8-41).
 * Row softmax of the first L columns (columns L..ld-1 are written as 0): Y = softmax(X*scale + mask), mask = -100
 * where gid_q[win][row] != gid_k[win][col] (rows are ordered [batch][win][period]); and its backward
 * dS = P * (dP - rowsum(P*dP)) * scale.  Used by the unfused attention backward (P is recomputed, never stored by fwd). */
int emip_softmax_rows(const void* X, void* Y, long rows, int L, long ld, float scale, const int* gid_q, const int* gid_k,
                      long period, long nwin, int dtype, void* stream);
int emip_softmax_bwd_rows(const void* P, const void* DP, void* DS, long rows, int L, long ld, float scale, int dtype,
                          void* stream);

/* Per-head products of the attention backward in ONE launch each (lib/pvt_v2.py:113-121: heads are 64-column slices of
 * [B][tokens][C] tensors): batch = B * heads, the operand of (b, h) sits at b * bs + h * hs.  emip_gemm_tn_heads ADDS its
 * results with atomics into a PRE-ZEROED f32 output that may be a column slice of a wider buffer (ldc > K). */
int emip_gemm_heads(const void* A, const void* W, void* C, int M, int N, int K, long lda, long ldw, long ldc, int batch,
                    int heads, long bsA, long hsA, long bsW, long hsW, long bsC, long hsC, int dtype, void* stream);
int emip_gemm_tn_heads(const void* A, const void* B, float* C, long M, int N, int K, long lda, long ldb, long ldc,
                       int batch, int heads, long bsA, long hsA, long bsB, long hsB, long bsC, long hsC, int dtype,
                       void* stream);
/* Y[(b, y, x)][(ky * 3 + kx) * C + c] = X[b][y + ky - 1][x + kx - 1][c] (zero outside the image): the 3 x 3 patch matrix of a
 * channels-last map, the A operand of a convolution whose weights differ per image (the factored conv_corr.0 of
 * model/EMIP_short/model.py:59,96 -- see emip_amd/model/EMIP_short/model.py run_conv_corr_factored). */
int emip_im2col3x3(const void* X, long ldx, void* Y, long ldy, int B, int H, int Wd, int C, int dtype, void* stream);
/* the adjoint of emip_im2col3x3: DX[b][y][x][c] = the sum over the taps of DY at the patch rows that read pixel (y, x) */
int emip_col2im3x3(const void* DY, long lddy, void* DX, long lddx, int B, int H, int Wd, int C, int dtype, void* stream);

int emip_transpose_pad_heads(const void* X, long ldx, long bsx, long hsx, void* Y, int batch, int heads, int R, int C,
                             int Rpad, int dtype, void* stream);

/* Y[z][c][r] = r < R ? X[z][r][c] : 0 (r < Rpad): the K^T operand of dQ = dS K. */
int emip_transpose_pad(const void* X, long ldx, long bsx, void* Y, long bsy, int batch, int R, int C, int Rpad,
                       int dtype, void* stream);

/* dz = dy * d/dz gelu(z) (exact erf form, lib/pvt_v2.py:50). */
int emip_gelu_bwd(const void* Z, long ldz, const void* DY, long lddy, void* DZ, long lddz, long M, int C, int dtype,
                  void* stream);

/* Depthwise 3x3 weight/bias gradient (lib/pvt_v2.py:316-327; PromptInteract.py q/kv/ffn dwconv), ACCUMULATED into dW f32 [9][C] and db f32 [C] (db may be NULL). */
int emip_dwconv3x3_wgrad(const void* X, long ldx, const void* DY, long lddy, float* dW, float* db, int B, int H, int Wd,
                         int C, int dtype, void* stream);

/* Backward of depthwise 3x3 (+ exact GELU when gelu != 0) in one pass, bf16 (lib/pvt_v2.py:45-54,316-327: DWConv + nn.GELU of
 * the Mlp): dPre = DY * gelu'(Z) (Z = the forward's pre-activation; ignored without gelu), DX = input gradient,
 * dW f32 [C][9] (the parameter's own [C][1][3][3] order) and db f32 [C] (may be NULL) are ACCUMULATED into; wt f32 [9][C] are
 * the forward's taps.  Replaces emip_gelu_bwd + emip_dwconv3x3 on flipped taps + emip_dwconv3x3_wgrad. */
int emip_dwconv3x3_bwd_fused(const void* X, long ldx, const void* Z, long ldz, const void* DY, long lddy, void* DX, long lddx,
                             const float* wt, float* dW, float* db, int B, int H, int Wd, int C, int gelu, void* stream);

/* Train-mode BatchNorm2d (+ReLU when OUT != NULL) backward (ConvBR, create_backbone.py:22-42; conv_corr, model.py:59-62).  X: pre-BN conv output, OUT: the forward output,
 * fsums: the forward's emip_chan_stats (groups = 1); dgamma/dbeta accumulated; ws: f32 [2*C]. */
int emip_bn_train_bwd(const void* X, long ldx, const void* DY, long lddy, const void* OUT, long ldo, void* DX, long lddx,
                      const double* fsums, const float* gamma, float* dgamma, float* dbeta, float* ws, long rows, int C,
                      float eps, int dtype, void* stream);

/* Adjoints of emip_bilinear (scatter-add) / emip_bilinear_planar (gather, no atomics) into an f32 channels-last
 * accumulation buffer: backward of the decoder's nn.Upsample / F.interpolate (create_backbone.py:52,74-75) and of the
 * train-mode x8 flow upsampling (gmflow.py:133-136). */
int emip_bilinear_bwd(const void* DY, long lddy, float* DX, int B, int H, int Wd, int C, int Ho, int Wo,
                      int align_corners, float mul, int dtype, void* stream);
int emip_bilinear_planar_bwd(const float* DY, float* DX, long ldx, int xc, int B, int H, int Wd, int C, int Ho, int Wo,
                             int align_corners, float mul, void* stream);

/* Input-gradient helpers for strided convs: zero insertion (then the stride-1 conv with flipped/transposed weights is
 * the transposed conv) and the un-patchify copy of non-overlapping patch convs (k == stride, pad 0: SRA's sr conv). */
int emip_zero_insert(const void* DY, long lddy, void* Z, int B, int Ho, int Wo, int H, int Wd, int C, int stride,
                     int dtype, void* stream);
int emip_depatchify(const void* P, void* DX, int B, int Ho, int Wo, int k, int C, int dtype, void* stream);

/* Gated GELU of the MDTA feed-forward (PromptInteract.py:383): y = gelu(z[:, :Ch]) * z[:, Ch:2Ch] (zero for
 * Ch <= c < Cpad) and its backward; column-scaled add Y = A + S[group][c] * B (normalisation terms of the MDTA
 * backward); the 64x64 part of the MDTA channel-attention backward. */
int emip_gate_fwd(const void* Z, long ldz, void* Y, long ldy, long M, int Ch, int Cpad, int dtype, void* stream);
int emip_gate_bwd(const void* Z, long ldz, const void* DY, long lddy, void* DZ, long lddz, long M, int Ch, int dtype,
                  void* stream);
int emip_colscale_add(const void* A, long lda, const void* Bp, long ldb, const float* S, long lds, void* Y, long ldy,
                      long M, int C, long rows_per_group, int dtype, void* stream);
int emip_mdta_bwd_small(const float* G, const float* nq2, const float* nk2, const float* temperature, const void* A,
                        const float* dA, void* dG, void* dGT, float* sq, float* sk, float* dtau, int B, int heads,
                        int dtype, void* stream);

/* Backward of the swin window split / merge (gmflow/utils.py:5-51, transformer.py:76-101).
 * Window gather (scatter != 0: the inverse): dst[(b*nwin+win)*Lp + t] = src[b][table[win][t]], t < L <= Lp -- the
 * dense (row-padded) batches the unfused window-attention backward works on.  Pad rows are not touched. */
int emip_window_rows(const void* src, void* dst, const int* table, int B, int nwin, int L, int Lp, long n, int C,
                     long ld_full, long ld_win, int scatter, int dtype, void* stream);

/* Y = alpha * A + beta * B (row-strided, 4-channel granularity). */
int emip_axpby(const void* A, long lda, const void* B, long ldb, void* Y, long ldy, long M, int C, float alpha,
               float beta, int dtype, void* stream);

/* Standalone activation (EMIP_ACT_RELU / EMIP_ACT_GELU) and ReLU backward (dx = dy where the forward output > 0). */
int emip_act_fwd(const void* X, long ldx, void* Y, long ldy, long M, int C, int act, int dtype, void* stream);
int emip_relu_bwd(const void* Yo, long ldy, const void* DY, long lddy, void* DX, long lddx, long M, int C, int dtype,
                  void* stream);

/* Backward of emip_convex_upsample (gmflow.py:64-77): dlogits (576 channels) and dflow (f32, zero-filled inside). */
int emip_convex_upsample_bwd(const void* logits, long ldl, const float* flow, const float* dY, void* dlogits, long lddl,
                             float* dflow, int N, int H, int Wd, int dtype, void* stream);

/* Loss backward (loss/loss_pred.py:4-22; loss/loss_flow.py:35-49,96-131; loss/warp_utils.py:83-93):
 * hybrid_e_loss w.r.t. the logits (ws = the scratch filled by emip_hybrid_e_loss), the photometric
 * loss w.r.t. the reconstruction (sums = scratch of emip_photometric_loss; abc f32 [3*B*C*H*W]), flow_warp w.r.t. the
 * flow.  gout: f32 [1] upstream gradient on the device. */
int emip_hybrid_e_loss_bwd(const float* pred, const float* mask, double* ws, const float* gout, float* dpred, int B,
                           int H, int W, void* stream);
int emip_photometric_loss_bwd(const float* im, const float* rec, const float* mask, const double* sums, float* abc,
                              const float* gout, float* drec, int B, int C, int H, int W, float weight, int accumulate,
                              void* stream);
int emip_flow_warp_bwd(const float* X, const float* flow, const float* dY, float* dflow, int B, int C, int H, int W,
                       void* stream);

/* out[c] += sum over rows of X[row][c] (bias gradients); out is f32 [C], accumulated into. */
int emip_colsum(const void* X, long ldx, float* out, long rows, int C, int dtype, void* stream);

/* ---- after the path ---------------------------------------------------- */

/* Prediction post-processing of test.py:28-31 on the device: bilinear resize of the mask logits to the source frame
 * size (align_corners=False), sigmoid, per-image min-max normalisation, x255, PIL 'F'->'L' conversion (clip, truncate).
 * logits f32 [B][1][H][W] -> out u8 [B][Ho][Wo]; ws: int [2*B] scratch.  No intermediate tensor, no host sync. */
int emip_postprocess_mask(const float* logits, unsigned char* out, int* ws, int B, int H, int W, int Ho, int Wo,
                          void* stream);

int emip_postprocess_mask_f32(const float* logits, float* out, int* ws, int B, int H, int W, int Ho, int Wo,
                              void* stream);   /* same, as the f32 map in [0,1] that train.py:125-127 hands to the metrics */

/* Validation metrics of train.py:129-137: the pixel sums behind MAE and S-measure
 * (eval/metrics.py:20-25,100-102,120-213) for one frame.  pred f32 [H][W] (the map passed to `step(pred=...)`),
 * gt f32 [H][W] in 0..255; acc f64 [40] (layout in csrc/eval_metrics.hip), ws int [2].  Finalised on the host from the
 * 40 doubles (emip_amd/eval_metrics.py). */
int emip_eval_frame(const float* pred, const float* gt, double* acc, int* ws, int H, int W, void* stream);

/* WeightedFmeasure.cal_wfm (eval/metrics.py:347-383; stepped per validation frame at train.py:100,131) for one frame:
 * scipy's exact Euclidean feature transform with its nearest-index choice on ties (bwdist(gt == 0, return_indices=True)),
 * Et, the 7x7 Gaussian EA, MIN_E_EA, the distance weighting and the sums.  pred / gt as for emip_eval_frame; kc f64 [50]
 * on the device = the 49 taps of matlab_style_gauss2D((7,7), 5) (:385-393) then log(0.5)/5, computed by the host in float64
 * as the reference does; out f64 [4] = n_gt, sum Ew[gt], sum Ew[~gt], unused; ws: 256 + 16*H*W bytes of scratch,
 * 16-byte aligned.  The host finalises R, P, Q (emip_amd/eval_metrics.py) and returns 0 when n_gt == 0 (:341-342). */
int emip_eval_wfm(const float* pred, const float* gt, const double* kc, double* out, void* ws, int H, int W,
                  void* stream);
/* The feature transform alone: idx int [2][H][W] (row indices, then column indices; -1 when the frame has no foreground). */
int emip_eval_edt_indices(const float* gt, int* idx, void* ws, int H, int W, void* stream);

/* Input preparation of dataset/dataset.py:257-260,76-79 on the device:
 * transforms.Resize((Ho, Wo)) on the decoded 8-bit RGB frame -- Pillow's two-pass 8-bit resampling with 22-bit
 * quantised triangle coefficients, BIT-EXACT -- then ToTensor (/255) and Normalize ((x - mean) / std) in IEEE f32.
 * img u8 [B][H0][W0][3] (byte strides img_bs / img_rs) -> out f32 [B][3][Ho][Wo]; out_u8 (may be NULL) u8 [B][Ho][Wo][3];
 * tmp u8 [B][H0][Wo][3] scratch; kh [Wo][ksh], bh [Wo][2] (first tap, tap count), kv [Ho][ksv], bv [Ho][2]: the
 * coefficient tables (device ints) built by the host for this (H0, W0); mean3 / std3: HOST pointers to 3 floats. */
int emip_preprocess_rgb(const unsigned char* img, long img_bs, long img_rs, int B, int H0, int W0, const int* kh,
                        const int* bh, int ksh, const int* kv, const int* bv, int ksv, unsigned char* tmp, float* out,
                        unsigned char* out_u8, int Ho, int Wo, const float* mean3, const float* std3, void* stream);

/* The ground-truth transform of dataset/dataset.py:80-82 (Resize on the 'L' mask + ToTensor): the same two integer passes
 * on one channel, then v / 255.  img u8 [B][H0][W0] -> out f32 [B][1][Ho][Wo]; out_u8 (may be NULL) u8 [B][Ho][Wo];
 * tmp u8 [B][H0][Wo] scratch; tables as for emip_preprocess_rgb. */
int emip_preprocess_gray(const unsigned char* img, long img_bs, long img_rs, int B, int H0, int W0, const int* kh,
                         const int* bh, int ksh, const int* kv, const int* bv, int ksv, unsigned char* tmp, float* out,
                         unsigned char* out_u8, int Ho, int Wo, void* stream);

/* Training-time augmentation (dataset/data_augment.py:12-45, applied at dataset/dataset.py:95-98), Pillow's 8-bit
 * arithmetic reproduced bit for bit; the random draws stay with the host and only their values are passed.
 * emip_color_enhance: colorEnhance (:22-31) = ImageEnhance Brightness -> Contrast -> Color -> Sharpness with the four
 *   factors; img / out u8 [H][W][3] packed RGB (distinct buffers), tmp u8 [H][W][3] scratch, lsum 8 bytes of scratch.
 * emip_rotate_bicubic: Image.rotate(angle, BICUBIC) (:13-18) given the inverse affine matrix Image.rotate derives from the
 *   angle (HOST pointer to 6 doubles); img / out u8 [H][W][C], C = 3 or 1.
 * emip_scatter_u8: randomPeper (:34-45), img[offs[i]] = vals[i] for n distinct offsets. */
int emip_color_enhance(const unsigned char* img, unsigned char* out, unsigned char* tmp, void* lsum, int H, int W,
                       float f_bright, float f_contrast, float f_color, float f_sharp, void* stream);
int emip_rotate_bicubic(const unsigned char* img, unsigned char* out, int H, int W, int C, const double* matrix6,
                        void* stream);
int emip_scatter_u8(unsigned char* img, const int* offs, const unsigned char* vals, int n, void* stream);

/* ---- optimizer ----------------------------------------------------------------------------------- */

/* Element-wise gradient clamp to +-clip (utils/utils.py:1-11; clip <= 0 disables) fused with one AdamW step
 * (train.py:380) over every trainable tensor in a single launch.  recs: device array of records
 * {float* p; const float* g; float* m; float* v; long n;}; blockmap: device int2[nblocks] = (record, chunk) with
 * chunks of emip_adamw_chunk() elements. */
int emip_clamp_adamw(const void* recs, const void* blockmap, int nblocks, float lr, float beta1, float beta2,
                     float eps, float weight_decay, float clip, int step, void* stream);
int emip_adamw_chunk(void);
/* the same with the gradient pointers taken from their own device array gptrs (const float* [records]; the g field of the
 * records is then ignored): the only part of the tables that changes from step to step */
int emip_clamp_adamw_g(const void* recs, const void* blockmap, const void* gptrs, int nblocks, float lr, float beta1,
                       float beta2, float eps, float weight_decay, float clip, int step, void* stream);

/* Refresh of every kernel-ready weight copy after an optimizer step (what nn_base.pack_linear / pack_conv / pack_dw and the
 * input-gradient packs of lib/pvt_v2.py build with torch ops on first use) in ONE launch.  recs: device array of 88-byte
 * records {const float* src; void* dst; long n; long s0, s1, s2, s3; long base; int d1, d2, d3, valid3; int dst_bf16, pad;}:
 * contiguous dst of dims [n/(d1 d2 d3)][d1][d2][d3] <- src[base + i0 s0 + i1 s1 + i2 s2 + i3 s3], zero where i3 >= valid3;
 * blockmap: device int2[nblocks] = (record, chunk) with chunks of emip_repack_chunk() dst elements.  The reference rebuilds
 * nothing (its modules read the f32 parameters directly, train.py:61-62); this is bookkeeping of the bf16 mode. */
int emip_repack(const void* recs, const void* blockmap, int nblocks, void* stream);
int emip_repack_chunk(void);

/* Train-mode nn.BatchNorm2d bookkeeping (create_backbone.py:18-29 ConvBR in train mode, model.py:59-62): running_mean /
 * running_var momentum update with the unbiased batch variance and num_batches_tracked += 1 (tracked may be NULL), from the
 * f64 sums [C][2] of emip_chan_stats over n values per channel. */
int emip_bn_running_update(const double* sums, float* running_mean, float* running_var, long long* tracked, long n,
                           float momentum, int C, void* stream);

/* ---- data-parallel gradient buckets (train.py:279: DistributedDataParallel's flat gradient buckets; here the exchange is
 * emip_amd/dp.py over torch.distributed = RCCL) ------------------------------------------------------------------------------
 * emip_grad_pack: gather the f32 gradient tensors of one bucket into their slices of the flat transport buffer (f32, or bf16
 * when flat_bf16) in one launch.  recs: device array of 16-byte records {long offset_in_flat; long n;}, blockmap: device
 * int2[nblocks] = (record, chunk of emip_adamw_chunk() elements), gptrs: device array of the tensors' f32 pointers (a null
 * pointer packs zeros: a parameter that took no part in this step).
 * emip_grad_unpack: the way back after the collective, gradient <- scale * flat slice (scale = 1 / world: DDP's mean); null
 * pointers are skipped.  One launch for all buckets.
 * emip_shard_sum: out[chunk] = sum over the rows of in[world][chunk] accumulated in f32 (both f32, or both bf16): the reduce
 * step of the direct reduce-scatter (all-to-all, local sum, all-gather) on the fully connected xGMI mesh. */
int emip_grad_pack(const void* recs, const void* blockmap, const void* gptrs, int nblocks, void* flat, int flat_bf16,
                   void* stream);
int emip_grad_unpack(const void* recs, const void* blockmap, const void* gptrs, int nblocks, const void* flat, int flat_bf16,
                     float scale, void* stream);
int emip_shard_sum(const void* in, void* out, int world, long chunk, int is_bf16, void* stream);

/* The Mlp half of a PVTv2 stage-3 block per band of an image (round 4; bf16 inference, 22 x 22 tokens, C = 320, N = 1280):
 * Out = X + fc2(GELU(dwconv3x3(LN(X) W1^T + b1) + bd)) + b2 (lib/pvt_v2.py:45-54,165-169,316-327), out_stats = (sum, sum of
 * squares) of the rows of Out (may be NULL; stored by fixed-order reductions: reproducible bit for bit).  A workgroup owns a
 * quarter of an image's tokens (+ 23 halo tokens on either side) for all hidden channels, which never leave the CU; fc1,
 * depthwise + GELU and fc2 of three consecutive 32-channel chunks run as a software pipeline with one barrier per chunk.
 * X, Out: bf16 [B, 22, 22, 320] (row strides ldx, ldo), Out must NOT overlap X.  Wst: emip_mlp_band_stage_bytes() bytes, the 42
 * pipeline stages [W1 chunk t in MFMA-fragment order, LayerNorm scale folded in | W2 chunk t - 2 in fragment order | fc1 bias
 * (+ W1 beta) and row sums of the packed W1 of chunk t, f32] (emip_amd/ops.py: mlp_band_packs); taps: f32 [40][10][32], the 9
 * depthwise taps and the depthwise bias per hidden channel, chunk-major; b2: f32 [320]; ln_stats: f32 [B 484][2] (sum, sum of
 * squares) of the rows of X.  bands: workgroups per image, 4 (121 tokens each) or 8 (60 / 61 tokens: half the launch duration
 * while 4 B workgroups would leave CUs idle), 0 = chosen by B; the result does not depend on it. */
int emip_mlp_band_eligible(int B, int H, int W, int C, int N);
int emip_mlp_band_stage_bytes(void);
int emip_mlp_band(const void* X, long ldx, const void* Wst, const float* taps, const float* b2, const float* ln_stats,
                  float eps, void* Out, long ldo, float* out_stats, int B, int H, int W, int C, int N, int bands, void* stream);

/* 3 x 3 convolutions of the GMFlow CNN encoder's residual blocks (stride 1, zero padding 1, no bias, C -> C channels, bf16
 * channels-last) as DIRECT convolutions on an LDS-resident halo tile, fused with the InstanceNorm2d around them
 * (gmflow/backbone.py:39-69 ResidualBlock: conv1 -> norm1 -> relu -> conv2 -> norm2 -> relu; :154-192 CNNEncoder):
 *   Y = conv3x3(f(X)),  f = identity (in_sums NULL) or relu((X - mean) rstd) per (image, channel) from in_sums f64 [B][C][2] =
 *   (sum, sum of squares) over the image (biased variance, in_eps) -- the reference's norm1 + relu applied while the input tile
 *   is staged, zero padding applied AFTER it as in the reference;
 *   out_sums f64 [B][C][2] (may be NULL): the same sums of the stored (bf16-rounded) Y, taken in the epilogue: one partial per
 *   (image, workgroup) in ws, combined in workgroup order by the workgroup that finishes an image's last tile (no atomics on
 *   the sums, reproducible).
 * Shapes: C = 64 with H, W multiples of 16 (all nine taps' weights resident in LDS, 16 x 16 tiles), or C = 64 / 96 / 128 with H
 * a multiple of 11 and W of 22 (weights streamed per tap, 11 x 22 tiles: 176, 88 and 44 qualify).  X, Y: [B, H, W, C] with row
 * strides ldx, ldy (elements); Wp: emip_conv3x3_halo_pack_bytes(C) bytes, the weights in MFMA-fragment order
 * [tap][C / 32][C / 16][64 lanes][8] (emip_amd/ops.py: conv3x3_halo_pack); ws: emip_conv3x3_halo_ws_bytes bytes, 64-byte aligned,
 * its first 4 B bytes zero before the first use, not shared by launches that may run at once. */
int emip_conv3x3_halo_eligible(int B, int H, int W, int Cin, int Cout);
int emip_conv3x3_halo_pack_bytes(int C);
long emip_conv3x3_halo_ws_bytes(int B, int H, int W, int C);
int emip_conv3x3_halo(const void* X, long ldx, const void* Wp, void* Y, long ldy, int B, int H, int W, int Cin, int Cout,
                      const double* in_sums, float in_eps, double* out_sums, void* ws, long ws_bytes, void* stream);

/* The GMFlow CNN encoder's stem (gmflow/backbone.py:84,154-160: Conv2d(3, 64, 7, stride 2, padding 3, bias=False) + norm1 + relu1)
 * as a direct convolution on an LDS halo tile: Y [B, H / 2, W / 2, 64] = conv7x7(X [B, H, W, 8]) (the 3 image channels stored
 * as 8), bf16 channels-last, H and W multiples of 32; Wp: 51 200 bytes in MFMA-fragment order [2][25 k-steps of two taps][64
 * lanes][8] (emip_amd/ops.py: conv_stem_pack); out_sums f64 [B][64][2] (may be NULL) = (sum, sum of squares) of the stored Y per
 * image and channel from the epilogue, through ws (emip_conv3x3_halo_ws_bytes(B, H / 2, W / 2, 64) bytes, tickets zero): the
 * InstanceNorm itself is applied by the consumers (emip_conv3x3_halo in_sums, emip_chan_norm_apply_res). */
int emip_conv_stem_eligible(int B, int H, int W, int Cin, int Cout);
int emip_conv_stem(const void* X, long ldx, const void* Wp, void* Y, long ldy, int B, int H, int W, int Cin, int Cout,
                   double* out_sums, void* ws, long ws_bytes, void* stream);

/* ---- calibration switches: libemip_hip_tuning.so ONLY (make -C emip_amd/csrc tuning, -DEMIP_TUNING) ----------------------
 * Tile / ring-depth overrides and work-skipping ablations (no stores / no MFMA / no loads) for tools/.  The product
 * library libemip_hip.so does not contain them (bench.py checks), so nothing a benchmark runs can skip work. */
#ifdef EMIP_TUNING
int emip_debug_set(int key, int value);
int emip_debug_set_tn(int target_workgroups);   /* 0 = heuristic split count of emip_gemm_tn */
int emip_debug_set_lnb(int wide);               /* 0 = the narrow LayerNorm-backward kernel */
int emip_debug_set_dww(int chunks);             /* row chunks per image of the depthwise weight gradient (0 = auto) */
int emip_debug_set_tn8(int ring_depth, int target_workgroups);   /* emip_gemm_tn8: 2..4 stages; 0 = heuristic split count */
int emip_debug_set_wa(int flags);              /* emip_window_attention ablations: 1 no S MFMAs, 2 no softmax, 4 no PV, 8 no DMA */
int emip_debug_set_halo(int mode);         /* emip_conv3x3_halo at 64 channels: 1 = the streamed-weights form, 2 = that with one halo buffer and two workgroups per CU */
int emip_debug_set_md(int flags);              /* emip_mlp_band ablations: 1 no fc1 MFMAs, 2 no depthwise pass, 4 no fc2 MFMAs, 8 no weight DMA, 16 no H / G stores, 32 constant taps */
int emip_debug_set_md_prof(void* buf);         /* emip_mlp_band: u64 [workgroups * 8 waves][6] cycle counters (barrier wait, DMA issue, fc1, fc2, depthwise, total) or NULL */
int emip_debug_set_mb(int flags);              /* emip_mlp_block ablations: 1 no fc1 MFMAs, 2 no depthwise pass, 4 no fc2 MFMAs, 8 no weight DMA, 16 no H store */
int emip_tuning_gemm8_dbg(int flags);           /* gemm8 ablations: 1 no epilogue stores, 2 no MFMA, 4 no operand loads, 8 bare launch */
#endif

#ifdef __cplusplus
}
#endif
#endif

/*
 * movae.h -- C ABI of libmovae_hip.so: the MI355X (gfx950) kernels under MO-VAE's per-step
 * training hot path.  The reference has no native boundary of its own
 * (it is pure Python on ATen + torchjd); each entry point below names the reference call whose
 * device arithmetic it replaces (paths relative to the reference root).  INTEGRATION.md shows
 * the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a caller-owned DEVICE pointer (fp32 unless stated); nothing is allocated,
 *     freed or synchronised inside; work is enqueued on `stream` (a hipStream_t, may be NULL);
 *   - activations are NHWC ("channels_last"): x[n][h][w][c]; conv weights are [Co][KH][KW][Ci]
 *     and transposed-conv weights [Ci][KH][KW][Co] in memory (the channels_last image of the
 *     reference's OIHW / IOHW parameter shapes), Linear weights [out][in];
 *   - `ws`/`ws_bytes`: scratch for split-K partials / reduction partials; may be NULL/0 where stated (then no
 *     split-K).  The FIRST 4096 BYTES of the workspace are the library's hand-off counters: they must be zero
 *     before the first call (hipMemset once), are left zero by every call, and one workspace must only be used
 *     by one stream at a time; everything after the header is plain scratch with no state between calls;
 *   - return 0 on success, <0 on invalid argument (-1), unsupported shape (-2) or launch
 *     failure (-3); movae_last_error() gives the text for the calling thread.
 */
#ifndef MOVAE_H
#define MOVAE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* movae_stream_t; /* hipStream_t */

enum movae_act { MOVAE_ACT_NONE = 0, MOVAE_ACT_LRELU = 1, MOVAE_ACT_RELU = 2, MOVAE_ACT_TANH = 3, MOVAE_ACT_SIGMOID = 4 };
enum movae_recon { MOVAE_RECON_MSE = 0, MOVAE_RECON_BCE = 1, MOVAE_RECON_L1 = 2, MOVAE_RECON_SMOOTH_L1 = 3 };
enum movae_mgda_norm { MOVAE_MGDA_NONE = 0, MOVAE_MGDA_L2 = 1, MOVAE_MGDA_LOSS = 2, MOVAE_MGDA_LOSS_PLUS = 3 };
enum movae_upgrad_norm { MOVAE_UPGRAD_TRACE = 0, MOVAE_UPGRAD_MIN_L2 = 1, MOVAE_UPGRAD_COSINE = 2 };
enum movae_amtl_scale { MOVAE_AMTL_MIN = 0, MOVAE_AMTL_MEDIAN = 1, MOVAE_AMTL_RMSE = 2 };

int movae_version(void);
const char* movae_last_error(void);

/* ---- layout ------------------------------------------------------------------------------
 * NCHW <-> NHWC transposes at the model boundary (main.py:155 hands NCHW images; nn.Flatten /
 * nn.Unflatten in models/vae.py:128,143 order features NCHW). */
int movae_nchw_to_nhwc(const float* src, float* dst, int n, int c, int h, int w, movae_stream_t stream);
int movae_nhwc_to_nchw(const float* src, float* dst, int n, int c, int h, int w, movae_stream_t stream);

/* ---- convolutions (implicit GEMM on v_mfma_f32_32x32x2_f32) -----------------------------------
 * conv2d      : nn.Conv2d forward/backward      models/vae.py:121,170; vq_vae.py:232-257; vq_vae2.py:36-47; betatc_vae.py:104-110
 * convT2d     : nn.ConvTranspose2d              models/vae.py:151-156,163-168; vq_vae.py:287-300; vq_vae2.py:75-88
 * linear      : nn.Linear == conv2d with h=w=kh=kw=1   models/vae.py:133-137; betatc_vae.py:124-131
 * `act`/`slope` fuse the following nn.LeakyReLU/ReLU/Tanh/Sigmoid into the epilogue.
 * y[n][ho][wo][co] with ho = (hi + 2*pad - kh)/stride + 1 (conv) or (hi-1)*stride - 2*pad + kh + out_pad (convT). */
int movae_conv2d_fwd(const float* x, const float* w, const float* bias, float* y,
                     int n, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int pad,
                     int act, float slope, void* ws, size_t ws_bytes, movae_stream_t stream);
int movae_conv2d_dgrad(const float* dy, const float* w, float* dx,
                       int n, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int pad,
                       void* ws, size_t ws_bytes, movae_stream_t stream);
/* dw[co][kh][kw][ci] (+)= sum dy * x ; dbias[co] (+)= sum dy (dbias may be NULL). accumulate!=0 adds to dw/dbias. */
int movae_conv2d_wgrad(const float* dy, const float* x, float* dw, float* dbias,
                       int n, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int pad,
                       int accumulate, void* ws, size_t ws_bytes, movae_stream_t stream);
/* wgrad for `groups` (<= 8) cotangents of one forward in ONE launch (batched pull-back of mtl_backward's K losses):
 * dy is stacked [groups][n][ho][wo][co], x is shared; dw / dbias are HOST arrays of `groups` device pointers (rows of
 * the Jacobian); dbias or any dbias[i] may be NULL. */
int movae_conv2d_wgrad_grouped(int groups, const float* dy, const float* x, float* const* dw, float* const* dbias,
                               int n, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int pad,
                               int accumulate, void* ws, size_t ws_bytes, movae_stream_t stream);
int movae_convT2d_fwd(const float* x, const float* w, const float* bias, float* y,
                      int n, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int pad,
                      int act, float slope, void* ws, size_t ws_bytes, movae_stream_t stream);
int movae_convT2d_dgrad(const float* dy, const float* w, float* dx,
                        int n, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int pad,
                        void* ws, size_t ws_bytes, movae_stream_t stream);
int movae_convT2d_wgrad(const float* dy, const float* x, float* dw, float* dbias,
                        int n, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int pad,
                        int accumulate, void* ws, size_t ws_bytes, movae_stream_t stream);
/* The backward of one layer in one call: movae_conv[T]2d_dgrad over groups * n images (dy stacked [groups][n][ho][wo][co],
 * dx likewise) followed by movae_conv[T]2d_wgrad_grouped.  When both land on the small-tile MFMA kernels they share ONE
 * launch (igemm2_pair: the two problems only share read-only operands and neither fills the chip alone); results are
 * bit-identical to the two separate calls.  MOVAE_NO_PAIR=1 forces the separate launches. */
int movae_conv2d_dgrad_wgrad_grouped(int groups, const float* dy, const float* w, const float* x, float* dx, float* const* dw,
                                     float* const* dbias, int n, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw,
                                     int stride, int pad, int accumulate, void* ws, size_t ws_bytes, movae_stream_t stream);
int movae_convT2d_dgrad_wgrad_grouped(int groups, const float* dy, const float* w, const float* x, float* dx, float* const* dw,
                                      float* const* dbias, int n, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw,
                                      int stride, int pad, int accumulate, void* ws, size_t ws_bytes, movae_stream_t stream);
int movae_convT2d_wgrad_grouped(int groups, const float* dy, const float* x, float* const* dw, float* const* dbias,
                                int n, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int pad,
                                int accumulate, void* ws, size_t ws_bytes, movae_stream_t stream);
/* Two nn.Linear layers applied to the same input -- fc_mu || fc_var (models/vae.py:128-129,187-192, betatc_vae.py:100-101,181-182):
 *   y1 = x w1^T + b1, y2 = x w2^T + b2      x [m][k], w [n][k], y [m][n]                          (one launch)
 *   dx[g] = dy1[g] w1 + dy2[g] w2;  dw_i[g] = dy_i[g]^T x;  db_i[g] = column sums of dy_i[g]       (two launches, all groups)
 * dy_i stacked [groups][m][n], dx [groups][m][k] (NULL: not wanted), dw_i / db_i HOST arrays of device pointers (db_i or its entries
 * may be NULL; dw1 == dw2 == NULL: no weight gradients).  Results equal the four ordinary calls up to the summation order of dx.
 * Returns MOVAE_EUNSUPPORTED, with nothing launched, outside the one-launch GEMM kernels' range (groups <= 4, n, k multiples of 4, m * n,
 * m * k, n * k <= 2^20, reductions <= 2048, 16-byte aligned operands): make the ordinary calls then. */
int movae_linear_pair_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* y1, float* y2,
                          int m, int n, int k, movae_stream_t stream);
int movae_linear_pair_bwd(int groups, const float* dy1, const float* dy2, const float* w1, const float* w2, const float* x, float* dx,
                          float* const* dw1, float* const* dw2, float* const* db1, float* const* db2, int m, int n, int k,
                          movae_stream_t stream);

/* ---- BatchNorm2d (training statistics) + activation ------------------------------------------
 * nn.BatchNorm2d + nn.LeakyReLU   models/vae.py:123-125,157-158,169-170   (eps 1e-5, momentum 0.1)
 * y is the conv output [rows][c]; stats are accumulated in fp64.  `ws` must hold
 * movae_bn_ws_bytes(rows, c) bytes.  num_batches_tracked (device int64, may be NULL) is incremented by the
 * statistics kernel in training mode (torch's BatchNorm2d does it with a launch of its own). */
size_t movae_bn_ws_bytes(int rows, int c);
int movae_bn_act_fwd(const float* y, const float* gamma, const float* beta, float* out,
                     float* save_mean, float* save_rstd, float* running_mean, float* running_var,
                     long long* num_batches_tracked,
                     int rows, int c, float eps, float momentum, int training, int act, float slope,
                     void* ws, size_t ws_bytes, movae_stream_t stream);
/* dy = d(loss)/d(conv output); dgamma/dbeta written (or accumulated when accumulate!=0). */
int movae_bn_act_bwd(const float* dout, const float* y, const float* gamma, const float* beta,
                     const float* save_mean, const float* save_rstd, float* dy, float* dgamma, float* dbeta,
                     int rows, int c, int act, float slope, int accumulate,
                     void* ws, size_t ws_bytes, movae_stream_t stream);

/* the same backward for `groups` (<= 8) cotangents of ONE forward at once (the K loss cotangents of
 * mtl_backward, main.py:188-190): dout / dy are stacked [groups][rows][c]; y and the saved statistics are shared; the
 * batch means of the cotangent are taken per group; dgamma / dbeta are HOST arrays of `groups` device pointers
 * (rows of the Jacobian).  ws must hold 4096 + groups * (movae_bn_ws_bytes(rows, c) - 4096) bytes. */
int movae_bn_act_bwd_grouped(int groups, const float* dout, const float* y, const float* gamma, const float* beta,
                             const float* save_mean, const float* save_rstd, float* dy,
                             float* const* dgamma, float* const* dbeta,
                             int rows, int c, int act, float slope, int accumulate,
                             void* ws, size_t ws_bytes, movae_stream_t stream);

/* ---- element-wise ---------------------------------------------------------------------------- */
int movae_act_fwd(const float* x, float* y, size_t n, int act, float slope, movae_stream_t stream);
/* dx = dy * act'(.) evaluated from the activation OUTPUT `out` (valid for all five kinds) */
int movae_act_bwd(const float* dy, const float* out, float* dx, size_t n, int act, float slope, movae_stream_t stream);
/* dpre = dy * act'(out) for `groups` (<= 8) stacked cotangents [groups][rows][c] of one forward (out is shared) AND, in the
 * same pass, dbias[g][c] (+)= sum_rows dpre -- the bias gradient of the conv whose epilogue applied the activation
 * (models/betatc_vae.py:104-110, vq_vae2.py:36-47: Conv2d(bias) -> LeakyReLU / ReLU).  c %% 4 == 0; dbias: HOST array
 * of device pointers, or NULL: only dpre (one launch for all groups; the conv weight-gradient entry points form the column
 * sums of dpre themselves while they stage it). */
int movae_act_bwd_bias_grouped(int groups, const float* dy, const float* out, float* dpre, float* const* dbias,
                               int rows, int c, int act, float slope, int accumulate,
                               void* ws, size_t ws_bytes, movae_stream_t stream);
int movae_add(const float* a, const float* b, float* y, size_t n, movae_stream_t stream);
int movae_axpby(float alpha, const float* a, float beta, const float* b, float* y, size_t n, movae_stream_t stream);
/* channel concat / split of NHWC tensors: dst[rows][c_dst], src[rows][c_src] placed at channel offset */
int movae_copy_channels(const float* src, float* dst, int rows, int c_src, int c_dst, int src_off, int dst_off, int c_copy, movae_stream_t stream);
/* column sums: out[c] (+)= sum_rows x[rows][c]   (bias gradients) */
int movae_colsum(const float* x, float* out, int rows, int c, int accumulate, void* ws, size_t ws_bytes, movae_stream_t stream);

/* reparameterize  z = mu + eps * exp(0.5 * log_var)      models/vae.py:187-192, betatc_vae.py:206-216 */
int movae_reparam_fwd(const float* mu, const float* log_var, const float* eps, float* z, size_t n, movae_stream_t stream);
/* z = mu + exp(0.5 * log_var) * eps with eps ~ N(0, 1) drawn inside the launch (Philox4x32-10 keyed by state[0], counter block =
 * (output quad, state[1]); Box-Muller) and written to `eps` for the backward.  state: device uint64[2] = {seed, draws made so far};
 * advance != 0: state[1] += 1 by the block that finishes last, so a replayed hipGraph draws fresh noise each time (one launch
 * where torch.randn_like + movae_reparam_fwd are four).  One such call at a time per process (the arrival count is a device global). */
int movae_reparam_rng_fwd(const float* mu, const float* log_var, float* eps, float* z, size_t n, unsigned long long* state, int advance,
                          movae_stream_t stream);
int movae_reparam_bwd(const float* dz, const float* log_var, const float* eps, float* dmu, float* dlog_var, size_t n, movae_stream_t stream);

/* ---- losses -------------------------------------------------------------------------------------
 * recon: utils/objectives.py:95-97 (mse), 108-110 (bce), 129-131 (l1), 134-136 (smooth_l1): mean over all
 * elements; out[0] = scale * loss.  ws >= movae_reduce_ws_bytes(n). */
size_t movae_reduce_ws_bytes(size_t n);
int movae_recon_loss_fwd(const float* recons, const float* inputs, float* out, size_t n, int kind, float scale,
                         void* ws, size_t ws_bytes, movae_stream_t stream);
/* drecons = (*gscale_dev) * scale * d(mean loss)/d(recons) ; gscale_dev may be NULL (=1) */
int movae_recon_loss_bwd(const float* recons, const float* inputs, const float* gscale_dev, float* drecons,
                         size_t n, int kind, float scale, movae_stream_t stream);
/* The same, times act'(pre) for recons = act(pre) -- the decoder's output activation, whose derivative is a function of its output
 * (MOVAE_ACT_*): dpre is the gradient w.r.t. the PRE-activation values, so the producing convolution's backward needs no
 * activation-backward pass (ops.ActLink). */
int movae_recon_loss_bwd_act(const float* recons, const float* inputs, const float* gscale_dev, float* dpre, size_t n, int kind,
                             float scale, int act, float slope, movae_stream_t stream);
/* kl: utils/objectives.py:141-144, out[0] = scale * mean_b(-0.5 sum_d(1 + lv - mu^2 - e^lv)) */
int movae_kl_fwd(const float* mu, const float* log_var, float* out, int b, int d, float scale,
                 void* ws, size_t ws_bytes, movae_stream_t stream);
/* The scalar arithmetic of a loss_function (models/vq_vae.py:381-391, vq_vae2.py:313-334, betatc_vae.py:298-324) in ONE launch:
 * `nterms` device scalars, out[k] = f_k * w_k * (sum of the terms whose coef[k][t] != 0 -- a row's non-zero coefficients must all
 * equal w_k), k < nout <= 7, and out[nout] = out[0] + ... in order.  f_k = 1 except for k == anneal_row (>= 0) with iter_dev
 * given: BetaTC's annealing factor min(iter / anneal_steps, 1), the device counter iter_dev being advanced first when training
 * (models/betatc_vae.py:13,298-302); anneal_out (nullable) keeps the factor for the backward.  terms / g: HOST arrays of device
 * pointers (g has nout + 1 entries, NULL = absent cotangent); gterms: device [nterms]. */
int movae_combine_losses_fwd(int nterms, const float* const* terms, int nout, const float* coef, float* iter_dev, float anneal_steps,
                             int anneal_row, int training, float* out, float* anneal_out, movae_stream_t stream);
int movae_combine_losses_bwd(int nterms, int nout, const float* const* g, const float* coef, const float* anneal_dev, int anneal_row,
                             float* gterms, movae_stream_t stream);
/* VAE.loss_function (models/vae.py:211-228) in three launches: out[3] = (reconstruction_loss, kld_loss, total_loss = their fp32 sum);
 * the same arithmetic as movae_recon_loss_fwd + movae_kl_fwd + a tensor add.  ws >= the two reductions' workspaces together. */
int movae_vae_losses_fwd(const float* recons, const float* inputs, size_t n, int kind, float rec_scale,
                         const float* mu, const float* log_var, int b, int d, float kl_scale, float* out,
                         void* ws, size_t ws_bytes, movae_stream_t stream);
int movae_kl_bwd(const float* mu, const float* log_var, const float* gscale_dev, float* dmu, float* dlog_var,
                 int b, int d, float scale, movae_stream_t stream);
/* Beta-TC decomposition: models/betatc_vae.py:262-296.  out[0..2] = mi, tc, kld (unweighted means).
 * log_iw is the [b][b] fp32 log-importance-weight matrix of betatc_vae.py:275-289. */
int movae_tc_decomp_fwd(const float* z, const float* mu, const float* log_var, const float* log_iw, float* out,
                        float* lse_joint, float* lse_marg, int b, int d, void* ws, size_t ws_bytes, movae_stream_t stream);
/* g[0..2] = upstream gradients of (mi, tc, kld) on device */
int movae_tc_decomp_bwd(const float* z, const float* mu, const float* log_var, const float* log_iw,
                        const float* lse_joint, const float* lse_marg, const float* g,
                        float* dz, float* dmu, float* dlog_var, int b, int d, movae_stream_t stream);

/* ---- Sobel edge losses of the gradient-guided VAE --------------------------------------------
 * models/gg_vae.py:42-53 (depthwise Sobel x / y, zero padding 1), EPS = 1e-8 (gg_vae.py:8); images NHWC [n][h][w][c].
 * edge weights (gg_vae.py:125-132): w_raw[n*h*w] = max_c sqrt(gx^2 + gy^2 + EPS) of `inputs`, wmax[0] = global max;
 * edge-weighted pixel loss (gg_vae.py:134-137): out = scale * mean( w_raw / (wmax + EPS) * (recons - inputs)^2 );
 * edge matching losses: out = scale * mean f(sobel recons, sobel inputs), one point-wise variant per `mode`
 * (enum movae_edge_match; gp / gt = |sobel recons| / |sobel inputs|, sl1 = smooth-L1 with beta 1):
 *   MAG         sl1(gp - gt)                                  gg_vae.py:139-156, gg_vq_vae.py:184-199, gg_vq_vae2.py:118-129
 *   SIGNED_MSE  (rx - tx)^2 + (ry - ty)^2                     gg_vq_vae.py:172-182
 *   MAXNORM     sl1(gp / (max gp + EPS) - gt / (max gt + EPS)), gradient through the max   gg_vae.py:158-173, gg_vq_vae.py:201-216
 *   ANGLE       sl1(atan2(ry, rx) - atan2(ty, tx))            gg_vae.py:176-190, gg_vq_vae.py:219-232
 *   MASKED      sl1(m * gp - m * gt), m = gt > mean gt        gg_vq_vae.py:234-247
 *   COSINE      1 - cos(normalize(rx, ry), normalize(tx, ty)) gg_vae.py:192-208, gg_vq_vae.py:249-264
 * `stats`: device float[8] written by the forward and read by the backward of MAXNORM / MASKED (may be NULL otherwise).
 * The backward passes take the upstream gradient as a device scalar (gscale_dev, may be NULL = 1). */
enum movae_edge_match {
    MOVAE_EDGE_MAG = 0, MOVAE_EDGE_SIGNED_MSE = 1, MOVAE_EDGE_MAXNORM = 2, MOVAE_EDGE_ANGLE = 3, MOVAE_EDGE_MASKED = 4,
    MOVAE_EDGE_COSINE = 5
};
int movae_edge_weights(const float* inputs, float* w_raw, float* wmax, int n, int h, int w, int c,
                       void* ws, size_t ws_bytes, movae_stream_t stream);
int movae_edge_weighted_mse_fwd(const float* recons, const float* inputs, const float* w_raw, const float* wmax, float* out,
                                int n, int h, int w, int c, float scale, void* ws, size_t ws_bytes, movae_stream_t stream);
int movae_edge_weighted_mse_bwd(const float* recons, const float* inputs, const float* w_raw, const float* wmax,
                                const float* gscale_dev, float* drecons, int n, int h, int w, int c, float scale,
                                movae_stream_t stream);
int movae_edge_match_fwd(const float* recons, const float* inputs, float* out, int n, int h, int w, int c, float scale,
                         int mode, float* stats, void* ws, size_t ws_bytes, movae_stream_t stream);
/* tmp_a / tmp_b: two scratch tensors of the images' size (d loss / d Sobel-x and -y responses) */
int movae_edge_match_bwd(const float* recons, const float* inputs, const float* gscale_dev, float* drecons, float* tmp_a,
                         float* tmp_b, int n, int h, int w, int c, float scale, int mode, const float* stats,
                         movae_stream_t stream);

/* ---- vector quantiser ------------------------------------------------------------------------------
 * models/vq_vae.py:27-64: nearest code under ||x||^2 + ||e||^2 - 2 x.e (first index on ties), gather, and
 * sse[0] = sum (q - x)^2 (commitment and embedding losses are both sse/numel).  x,q: [rows][d], e: [k][d]. */
int movae_vq_nearest_fwd(const float* x, const float* e, float* q, int64_t* idx, float* sse, int32_t* used_count,
                         int rows, int k, int d, void* ws, size_t ws_bytes, movae_stream_t stream);
/* the same, and the two loss terms of models/vq_vae.py:51-52 written by the finalize kernel: mse2[0] = mse2[1] = sse / (rows * d) */
int movae_vq_nearest_fwd_mse(const float* x, const float* e, float* q, int64_t* idx, float* sse, int32_t* used_count, float* mse2,
                             int rows, int k, int d, void* ws, size_t ws_bytes, movae_stream_t stream);
/* dx = dq + gc * 2 (x - q)/numel ; de[k] = ge * 2/numel * sum_{idx[r]==k} (q[r] - x[r])   (gc, ge: device scalars, may be
 * NULL = 0; dx / de may be NULL).  The codebook gradient is a sorted segmented sum (no float atomics, bit-reproducible);
 * `ws` must hold movae_vq_bwd_ws_bytes(rows, k, d) bytes when de and ge are given. */
size_t movae_vq_bwd_ws_bytes(int rows, int k, int d);
int movae_vq_bwd(const float* x, const float* q, const int64_t* idx, const float* dq, const float* gc, const float* ge,
                 float* dx, float* de, int rows, int k, int d, void* ws, size_t ws_bytes, movae_stream_t stream);

/* ---- K-loss gradient aggregation ---------------------------------------------------------------------
 * torchjd GramianWeightedAggregator (base of utils/torchmoo/mgda.py:12, aligned_mtl.py:39): G = J J^T,
 * w = weighting(G), g = w @ J.   J is [k][m] row-major with leading dimension ldj, k <= MOVAE_MAX_K. */
#define MOVAE_MAX_K 8
size_t movae_gram_ws_bytes(int k, size_t m);
int movae_gram(const float* J, size_t ldj, int k, size_t m, float* G, void* ws, size_t ws_bytes, movae_stream_t stream);
/* UPGrad (torchjd; constructed main.py:1195): G/tr(G) (0 if tr<norm_eps) + reg_eps I; for each i solve
 * min 1/2 w'Gw s.t. w >= u_i e_i; sum rows.  pref may be NULL (u = 1/k).  Solved in fp64 on device. */
int movae_weights_upgrad(const float* G, int k, float norm_eps, float reg_eps, const float* pref, float* w, movae_stream_t stream);
/* the same projection on a differently normalised Gramian: MOVAE_UPGRAD_MIN_L2 = NUPGrad
 * (utils/torchmoo/nupgrad.py:115-158, main.py:1226), MOVAE_UPGRAD_COSINE = the other branch of PNUPGrad's coin flip
 * (utils/torchmoo/pnupgrad.py:13-24,127-134, main.py:1228); MOVAE_UPGRAD_TRACE is movae_weights_upgrad. */
int movae_weights_upgrad_norm(const float* G, int k, int norm_mode, float norm_eps, float reg_eps, const float* pref, float* w,
                              movae_stream_t stream);
/* movae_gram followed by movae_weights_upgrad_norm (dual == 0) or movae_weights_dualproj (dual != 0; norm_mode 0) in two launches
 * instead of three: the solver kernel folds the Gramian's block partials itself -- the arithmetic of movae_gram's final kernel, so
 * G [k][k] and w [k] are bit-identical to the two separate calls. */
int movae_gram_upgrad(const float* J, size_t ldj, int k, size_t m, float* G, int norm_mode, float norm_eps, float reg_eps,
                      const float* pref, float* w, int dual, void* ws, size_t ws_bytes, movae_stream_t stream);
/* MGDA Frank-Wolfe (utils/torchmoo/mgda.py:221-367); losses may be NULL for norm NONE/L2. info[0]=iterations */
int movae_weights_mgda(const float* G, int k, int norm, const float* losses, float epsilon, int max_iters,
                       float* w, int32_t* info, movae_stream_t stream);
/* StableMGDA (utils/torchmoo/mgda.py:139-153,286-317; COMFORT's --comfort_mgda_stable): the (normalised) Gramian's
 * eigenvalues are clamped to >= min_eigenvalue and the matrix reconstructed before the Frank-Wolfe iteration. */
int movae_weights_mgda_stable(const float* G, int k, int norm, const float* losses, float epsilon, int max_iters,
                              float min_eigenvalue, float* w, int32_t* info, movae_stream_t stream);
/* Aligned-MTL (utils/torchmoo/aligned_mtl.py:97-133) */
int movae_weights_amtl(const float* G, int k, int scale_mode, const float* pref, float* w, movae_stream_t stream);
/* torchjd aggregators that main.py:1196-1222 also offers (third-party torchjd @ main, absent from the reference tree:
 * restated from the published algorithms, parity unpinned):
 *   DualProj  one dual-cone projection of the preference vector (default mean weights) on the trace-normalised,
 *             regularised Gramian -- movae_weights_upgrad's QP solved once;
 *   PCGrad    gradient surgery on the Gramian; perm = K rows of a permutation of 0..K-1 (int32, device), row i is the
 *             order in which task i is de-conflicted against the others (the reference: torch.randperm per task);
 *   IMTL-G    w = pinv(G) d / sum(pinv(G) d), d_i = sqrt(G_ii); zeros when the sum vanishes;
 *   CAGrad    w = 1/K + c g0 / sqrt(w*'G w*) w*, w* = argmin over the simplex of (G 1/K)'w + c g0 sqrt(w'Gw), g0 = sqrt(1/K' G 1/K)
 *             (main.py:1216-1217: c = 1.0); the mean weights when c g0 or sqrt(w*'G w*) is <= norm_eps. */
int movae_weights_dualproj(const float* G, int k, float norm_eps, float reg_eps, const float* pref, float* w, movae_stream_t stream);
int movae_weights_pcgrad(const float* G, int k, const int32_t* perm, float* w, movae_stream_t stream);
int movae_weights_imtlg(const float* G, int k, float* w, movae_stream_t stream);
int movae_weights_cagrad(const float* G, int k, float c, float norm_eps, float* w, movae_stream_t stream);
/* constant weightings (torchjd Sum / Mean) */
int movae_weights_const(int k, float value, float* w, movae_stream_t stream);
/* g[m] (+)= sum_i w[i] J[i][:]  ; also usable as the hook's J.T @ w (main.py:112-118) */
int movae_combine(const float* J, size_t ldj, int k, size_t m, const float* w, float* g, int accumulate, movae_stream_t stream);
/* cos(J.T @ w, mean_rows(J)) -> out[0] (main.py:108-119) */
int movae_gd_similarity(const float* J, size_t ldj, int k, size_t m, const float* w, float* out,
                        void* ws, size_t ws_bytes, movae_stream_t stream);

/* ---- optimizer tail (torch.optim.Adam semantics, main.py:1169-1178,214) ------------------
 * flat multi-tensor Adam over one contiguous fp32 arena; step_dev holds the step count as float. */
int movae_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int decoupled_wd, int step, movae_stream_t stream);
/* the same update for a LIST of tensors in one launch per 64 tensors (torch.optim.Adam(foreach) semantics:
 * main.py:1169-1178 builds it, main.py:214 steps it).  p/g/m/v/numel are HOST arrays of n_tensors device
 * pointers / element counts.  hyper_dev == NULL: `step` (>= 1) is the step count and lr the learning rate.
 * hyper_dev != NULL: device float[2] = {steps made so far, lr}; the update is made for step counter + 1 with the device lr
 * (`step` and `lr` arguments are ignored) and the counter is advanced inside the same launch, by the block that finishes last
 * -- the form a captured hipGraph replays.  (One such call at a time per process: the kernel keeps its arrival count in a
 * device global.) */
int movae_adam_multi(int n_tensors, float* const* p, const float* const* g, float* const* m, float* const* v,
                     const size_t* numel, float lr, float beta1, float beta2, float eps, float weight_decay,
                     int decoupled_wd, int step, float* hyper_dev, movae_stream_t stream);
/* torch.nn.utils.clip_grad_norm_(parameters, max_norm) (main.py:211-212) over a LIST of gradient tensors in three
 * launches per 64 tensors: total_sumsq_dev[0] receives the squared total L2 norm (its sqrt is the value torch
 * returns), every gradient is scaled in place by min(max_norm / (norm + 1e-6), 1).  g / numel are HOST arrays. */
int movae_clip_grad_norm_multi(int n_tensors, float* const* g, const size_t* numel, float max_norm, float* total_sumsq_dev,
                               void* ws, size_t ws_bytes, movae_stream_t stream);
/* sumsq of a flat arena into out[0] (clip_grad_norm_, main.py:211-212) */
int movae_sumsq(const float* x, size_t n, float* out, void* ws, size_t ws_bytes, movae_stream_t stream);
int movae_scale_by_clip(float* g, size_t n, const float* sumsq_dev, float max_norm, movae_stream_t stream);

/* ---- opt-in reduced-precision operands (BASELINE.json configs[1] names bf16; the reference itself is fp32 end to end) ----------
 * MOVAE_DTYPE_BF16: the 128x128 implicit-GEMM convolution kernels (forward, input gradient, weight gradient -- the layers that
 * are MFMA-bound at the C3-C5 shapes) round their two operands to bf16 on the way into LDS and multiply on
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation; tensors in memory, master weights, BatchNorm, losses, aggregation and the
 * optimizer stay fp32.  Process-wide, returns the previous setting.  The default, MOVAE_DTYPE_F32, is the parity path: results
 * under bf16 are held to their own, looser tolerance (tests/test_hip_bf16.py), never to the fp32 oracle's. */
#define MOVAE_DTYPE_F32 0
#define MOVAE_DTYPE_BF16 1
int movae_set_compute_dtype(int dtype);

/* A weight gradient's split-K reduce need not be a launch of its own on the backward's dependency chain: nobody reads the weight
 * gradient before the aggregation / the optimizer.  movae_reduce_defer(1) arms the NEXT movae_conv*_wgrad* / *_dgrad_wgrad* call
 * (one call): its reduce is parked instead of launched, and the next implicit-GEMM launch on the same stream carries it as extra
 * blocks behind its own (same arithmetic, same order).  The caller guarantees (mo-vae_amd/ops.py does) that until then (a) nothing
 * reads the call's dw / dbias destinations and (b) nothing writes the scratch arena `ws` the call was given.  A second armed call,
 * a following call's bias column sums, or movae_reduce_flush() launch a still-parked reduce stand-alone.  Returns the previous
 * arming.  movae_reduce_defer_stats: out3 = {parked, carried by a later launch, launched stand-alone after all}; returns 1 while a
 * reduce is parked. */
int movae_reduce_defer(int on);
int movae_reduce_flush(void);
int movae_reduce_defer_stats(long long* out3, int reset);
/* Only reduces whose slabs hold at most this many bytes are parked (a parked reduce runs with its carrier's occupancy: right for a
 * launch-bound reduce, wrong for a bandwidth-bound one).  Default 12 MiB (MOVAE_DEFER_MAX_BYTES; measured: C2 0.841 -> 0.816 ms, C4 4.49 -> 4.44, larger slabs cost their carrier more than the launch saves); bytes < 0 only reads.  Returns the
 * previous value. */
long long movae_reduce_defer_max_bytes(long long bytes);

/* ---- measurement hook (bench.py's roofline leg only; no reference counterpart) ------------------------
 * on != 0: the conv family launches ONLY its main MFMA kernel (split-K reduce / bias column-sum launches are
 * skipped, so outputs are incomplete) so that one kernel can be timed between HIP events.  Process-wide;
 * returns the previous setting.  Never enabled by the product path. */
int movae_bench_main_kernel_only(int on);
/* name (as rocprofv3 prints it, without the argument list) of the main kernel the most recent conv-family call
 * on this process dispatched to, e.g. "igemm2_bwd<64,64>" -- lets bench.py group its HIP-event timings per kernel */
const char* movae_bench_last_kernel(void);
/* s > 0 pins the conv family's split-K factor (tools/conv_microbench.py tuning sweeps); 0 restores the heuristic.
 * Returns the previous value. */
int movae_bench_force_split(int s);
/* 1 / 0: the kernels of csrc/kgemm.h do / do not honour movae_fuse_t::fin_* (finish the following BatchNorm inside the forward
 * launch); -1: as MOVAE_KGEMM_BN_FIN says (default: not).  Returns the previous mode.  Tests and A/B measurements. */
int movae_bench_kgemm_bn_fin(int mode);
/* mode > 0: every conv-family call whose shape the block-internal split-K kernels (csrc/kgemm.h) can serve takes them, whatever
 * the size heuristic says; mode < 0: none does; 0: the heuristic (small, latency-bound problems only).  Lets the parity tests run
 * the whole conv test matrix through either family.  Returns the previous mode. */
int movae_bench_force_kgemm(int mode);

/* ---- BatchNorm fused into its neighbouring convolutions -----------------------------------------
 * models/vae.py:119-126,149-158: Conv2d / ConvTranspose2d -> BatchNorm2d (training statistics) -> LeakyReLU chains.  Instead of
 * a statistics pass and an apply pass per BatchNorm, (1) the conv that PRODUCES y emits the per-channel partial sums of y from
 * its epilogue (or from its split-K reduction), (2) movae_bn_finalize turns them into the saved statistics, the running
 * statistics and the folded map scale = gamma * rstd, shift = beta - mean * scale, and (3) the conv that CONSUMES the
 * normalised activation applies act(scale[c] * y + shift[c]) between its global load and its LDS store (forward and weight
 * gradient alike): the normalised tensor is never written.
 * movae_fuse_t asks one *_f call for (1) and / or (3).  in_scale == NULL: plain input.  stats == NULL: no statistics.  On return
 * stats_parts is the number of partial pairs written per channel, stats[(p * 2 + {0: sum, 1: sum of squares}) * co + c]; 0 when
 * the dispatched kernel cannot emit them (use movae_bn_stats).  A requested input transform that the dispatched kernel cannot
 * apply returns -2 (unsupported) BEFORE anything is launched: materialise with movae_scale_shift_act and call the plain entry.
 * in_slope: leaky-ReLU slope of the fused activation (1 = none, 0 = ReLU).  The *_f entry points are otherwise identical to
 * their plain namesakes (which call them with fuse == NULL). */
typedef struct movae_fuse {
    const float* in_scale; /* [ci] device pointers, 16-byte aligned, ci % 4 == 0 */
    const float* in_shift;
    float in_slope;
    float* stats;          /* device buffer for the partial sums (fwd entry points, act == none, no fused activation) */
    size_t stats_cap;      /* floats available at stats */
    int stats_parts;       /* OUT */
    /* input-gradient passes (dgrad / dgrad_wgrad entry points): dx is the gradient `dout` w.r.t. the never-materialised output
     * act(bn_scale[c] * bn_y + bn_shift[c]) of a fused BatchNorm over the raw conv output bn_y [n][hi][wi][ci] (shared by the
     * `groups` cotangents).  The epilogue / split-K reduce then also emits that BatchNorm's backward sums per cotangent group,
     * bn_part[((g * bn_ppg + p) * 2 + {0,1}) * ci + c] = partial (sum d, sum d * bn_y), d = dout * act'(...), for
     * movae_bn_bwd_finalize.  bn_ppg == 0 on return: not produced (shape / kernel without the epilogue). */
    const float* bn_y;
    const float* bn_scale;
    const float* bn_shift;
    float bn_slope;
    float* bn_part;
    size_t bn_cap;         /* floats available at bn_part */
    int bn_ppg;            /* OUT: partial pairs per group and channel */
    /* input-gradient passes, alternatively: dx is the gradient w.r.t. the OUTPUT of the epilogue activation of the layer before
     * (Conv2d(bias) -> LeakyReLU / ReLU -> this layer, models/betatc_vae.py:104-110, vq_vae.py:36-47), whose stored output is
     * ep_act_y [n][hi][wi][ci] (shared by the `groups` cotangents).  The epilogue / split-K reduce then stores
     * dx * act'(ep_act_y) -- the gradient w.r.t. that layer's PRE-activation -- and sets ep_act_done = 1; 0 on return: not applied
     * (kernel without the epilogue), dx is the plain input gradient and the caller runs movae_act_bwd. */
    const float* ep_act_y;
    int ep_act;            /* MOVAE_ACT_* of that layer */
    float ep_slope;
    int ep_act_done;       /* OUT: everything asked of the epilogue (ep_act_y and / or ep_res) was applied */
    /* ... and / or the layer is the first of a residual block's branch (out = branch(x) + x, models/vq_vae.py:127-145,
     * vq_vae2.py:13-28): ep_res [groups][n][hi][wi][ci] -- the gradient w.r.t. the block's output, i.e. the identity branch's
     * share -- is added to dx, which then is the block's complete input gradient. */
    const float* ep_res;
    /* forward entry points with `stats`: the training-mode BatchNorm that follows can be FINISHED inside the convolution's own
     * launch (kernels of csrc/kgemm.h: the block that arrives last at a column tile folds that tile's partial sums and writes what
     * movae_bn_finalize would).  fin_out: device [4][co] = save_mean, save_rstd, scale, shift; fin_running_* / fin_nbt as in
     * movae_bn_finalize (nullable).  fin_done == 1 on return: done, do not call movae_bn_finalize; 0: not done (another kernel
     * family served the shape), stats / stats_parts are as without the request.  Needs the workspace header (ws as passed). */
    const float* fin_gamma;
    const float* fin_beta;
    float fin_eps;
    float fin_momentum;
    float* fin_out;
    float* fin_running_mean;
    float* fin_running_var;
    long long* fin_nbt;
    int fin_done;          /* OUT */
} movae_fuse_t;
int movae_conv2d_fwd_f(const float* x, const float* w, const float* bias, float* y,
                       int n, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int pad,
                       int act, float slope, void* ws, size_t ws_bytes, movae_stream_t stream, movae_fuse_t* fuse);
int movae_convT2d_fwd_f(const float* x, const float* w, const float* bias, float* y,
                        int n, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int pad,
                        int act, float slope, void* ws, size_t ws_bytes, movae_stream_t stream, movae_fuse_t* fuse);
/* backward passes: `fuse` describes the activation operand x (only in_scale / in_shift / in_slope are read) */
int movae_conv2d_wgrad_grouped_f(int groups, const float* dy, const float* x, float* const* dw, float* const* dbias,
                                 int n, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int pad,
                                 int accumulate, void* ws, size_t ws_bytes, movae_stream_t stream, const movae_fuse_t* fuse);
int movae_convT2d_wgrad_grouped_f(int groups, const float* dy, const float* x, float* const* dw, float* const* dbias,
                                  int n, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int pad,
                                  int accumulate, void* ws, size_t ws_bytes, movae_stream_t stream, const movae_fuse_t* fuse);
int movae_conv2d_dgrad_wgrad_grouped_f(int groups, const float* dy, const float* w, const float* x, float* dx,
                                       float* const* dw, float* const* dbias,
                                       int n, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int pad,
                                       int accumulate, void* ws, size_t ws_bytes, movae_stream_t stream, const movae_fuse_t* fuse);
int movae_convT2d_dgrad_wgrad_grouped_f(int groups, const float* dy, const float* w, const float* x, float* dx,
                                        float* const* dw, float* const* dbias,
                                        int n, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int pad,
                                        int accumulate, void* ws, size_t ws_bytes, movae_stream_t stream, const movae_fuse_t* fuse);
int movae_conv2d_dgrad_f(const float* dy, const float* w, float* dx,
                         int n, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int pad,
                         void* ws, size_t ws_bytes, movae_stream_t stream, movae_fuse_t* fuse, int groups);
int movae_convT2d_dgrad_f(const float* dy, const float* w, float* dx,
                          int n, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int pad,
                          void* ws, size_t ws_bytes, movae_stream_t stream, movae_fuse_t* fuse, int groups);
/* BatchNorm backward from the sums above (models/vae.py:119-126 through autograd): per cotangent group g
 *   dbeta[g] = S1, dgamma[g] = rstd * (S2 - mean * S1)            (dgamma / dbeta: HOST arrays of device pointers, NULLs allowed)
 *   coef[g][0..2][c] such that dy = coef0 * d + coef1 * y + coef2, d = dout * act'(scale * y + shift)
 * and the one pass that forms dy [groups][rows][c] from dout [groups][rows][c] and y [rows][c].  accumulate != 0 adds to
 * dgamma / dbeta. */
int movae_bn_bwd_finalize(const float* bn_part, size_t bn_cap, int ppg, int groups, int rows, int c, const float* gamma,
                          const float* save_mean, const float* save_rstd, float* const* dgamma, float* const* dbeta, float* coef,
                          int accumulate, movae_stream_t stream);
int movae_bn_bwd_apply(const float* dout, const float* y, const float* scale, const float* shift, float slope, const float* coef,
                       float* dy, int groups, size_t rows, int c, movae_stream_t stream);
/* The two calls above as one: with few partials per group (the deep layers) a single launch folds them per 32-channel slice and forms
 * dy; otherwise the two launches.  coef is scratch of [groups][3][c] floats either way. */
int movae_bn_bwd_finalize_apply(const float* bn_part, size_t bn_cap, int ppg, int groups, size_t rows, int c, const float* gamma,
                                const float* save_mean, const float* save_rstd, float* const* dgamma, float* const* dbeta, float* coef,
                                int accumulate, const float* dout, const float* y, const float* scale, const float* shift, float slope,
                                float* dy, movae_stream_t stream);
/* partial sums -> mean / rstd (saved for movae_bn_act_bwd), scale / shift (for the consumers), running statistics with
 * nn.BatchNorm2d's momentum rule and unbiased variance, num_batches_tracked += 1 (each may be NULL).  rows = n * h * w.
 * stats_cap / bn_cap: floats available at the partials buffer -- with more than 256 partials and room behind them, a first
 * stage folds them to 64 rows there (the buffer is scratch: its contents are not preserved). */
int movae_bn_finalize(const float* stats, size_t stats_cap, int parts, int rows, int c, const float* gamma, const float* beta, float eps,
                      float momentum, float* save_mean, float* save_rstd, float* scale, float* shift, float* running_mean,
                      float* running_var, long long* num_batches_tracked, movae_stream_t stream);
/* the same partial sums from one read of y, for producers whose kernel cannot emit them; *parts_out is a HOST int */
int movae_bn_stats(const float* y, int rows, int c, float* stats, size_t stats_cap, int* parts_out, movae_stream_t stream);
/* out = leaky_relu(scale[c] * y + shift[c], slope): materialises a fused BatchNorm output */
int movae_scale_shift_act(const float* y, const float* scale, const float* shift, float* out, size_t rows, int c, float slope,
                          movae_stream_t stream);

/* ---- PixelCNN prior over the VQ code grids (models/pixelcnn_prior.py, trained by main.py:890-1085) -------
 * The prior's convolutions are movae_conv2d_* calls; MaskedConv2d (pixelcnn_prior.py:25-54) multiplies its weight by the
 * mask IN PLACE before every forward (movae_mul with y == a).
 *   movae_embedding_fwd    nn.Embedding forward of the code grid     pixelcnn_prior.py:306 (x = self.embedding(x)), :371
 *                          y[r][:] = weight[idx[r]][:]; idx int64 [rows] (a [B,H,W] grid gives NHWC activations directly)
 *   movae_embedding_bwd    its gradient: dweight[k][:] = sum_{r: idx[r]==k} dy[r][:] -- rows sorted by (code, row) and summed in
 *                          fixed order (no float atomics, bit-reproducible); ws >= movae_vq_bwd_ws_bytes(rows, k, d)
 *   movae_gated_residual_* GatedResBlock's combine: out = res + gate * feat with gate = sigmoid(conv_gate), feat = tanh(conv_feature)
 *                          already activated by the conv epilogues            pixelcnn_prior.py:85-90; n %% 4 == 0
 *   movae_cross_entropy_*  F.cross_entropy(logits.permute(0,2,3,1).reshape(-1,K), z.reshape(-1)) (mean reduction)
 *                          main.py:1003-1006,1026-1029; pixelcnn_prior.py:392-395.  logits [rows][k] (NHWC logits are that
 *                          matrix as they lie), lse [rows] is kept for the backward; gscale_dev: device scalar upstream gradient
 *                          or NULL (= 1). */
int movae_embedding_fwd(const float* weight, const int64_t* idx, float* y, size_t rows, int k, int d, movae_stream_t stream);
int movae_embedding_bwd(const float* dy, const int64_t* idx, float* dweight, int rows, int k, int d,
                        void* ws, size_t ws_bytes, movae_stream_t stream);
int movae_gated_residual_fwd(const float* res, const float* gate, const float* feat, float* out, size_t n, movae_stream_t stream);
int movae_gated_residual_bwd(const float* dout, const float* gate, const float* feat, float* dgate, float* dfeat, size_t n,
                             movae_stream_t stream);
int movae_mul(const float* a, const float* b, float* y, size_t n, movae_stream_t stream);
size_t movae_cross_entropy_ws_bytes(size_t rows);
int movae_cross_entropy_fwd(const float* logits, const int64_t* target, float* loss, float* lse, size_t rows, int k,
                            void* ws, size_t ws_bytes, movae_stream_t stream);
int movae_cross_entropy_bwd(const float* logits, const int64_t* target, const float* lse, const float* gscale_dev, float* dlogits,
                            size_t rows, int k, movae_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MOVAE_H */

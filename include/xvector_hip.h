/*
 * xvector_hip.h - C-ABI of the MI355X (gfx950) x-vector training/extraction hot path.
 *
 * The reference (mycrazycracy/tf-kaldi-speaker) has NO native/FFI interface on this
 * path: its hot path is a TF1 graph built by model/tdnn.py:33-191, model/pooling.py:9-34,
 * model/loss.py:9-355 and run by model/trainer.py:505-508 (sess.run(train_op)).  This
 * header is therefore the boundary that the drop-in Python `Trainer`
 * (tf_kaldi_speaker_amd/model/trainer.py) binds through ctypes; every entry point names
 * the reference call site(s) whose arithmetic it replaces.
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types.  All data pointers are DEVICE
 *     pointers (HBM) unless the name starts with h_.  All tensors are fp32 row-major,
 *     labels int32.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Every call
 *     only ENQUEUES work; nothing synchronises unless stated.
 *   - return 0 on success, non-zero on error; xv_last_error() gives the message of the
 *     last failing call on the calling thread.
 *   - an engine handle is single-threaded (one TF session per Trainer, trainer.py:24).
 *
 * "Spliced view": a frame-level activation tensor x[B][T][C] is consumed by a layer with
 * context width k as the matrix whose row (b,t) is the contiguous span x[b][t..t+k-1][:]
 * (k*C floats) - rows overlap in memory, nothing is materialised (no im2col).  k = 1 is a
 * plain dense layer.
 */
#ifndef XVECTOR_HIP_H
#define XVECTOR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: xv_config starts with struct_bytes (round 5).  3: xv_engine_arena_bytes (round 6).  Bumped whenever xv_config's layout or an entry point's signature changes: a host
 * built against another header must fail at load (xv_abi_version) or at xv_engine_create (struct_bytes), never read fields past the end
 * of a shorter struct. */
#define XV_ABI_VERSION 3

const char* xv_last_error(void);
int xv_abi_version(void);
/* number of visible HIP devices (0 when none / driver missing); never throws */
int xv_device_count(void);

/* Live kernel timing for bench.py's roofline leg: while enabled, every launch of the three MFMA
 * GEMM kernels is bracketed by hipEvents on the stream it is launched on.
 *   kind 0 = xv_gemm_nt_kernel<true, *> / xv_gemm_nt_sk_kernel<true, *>   (forward conv/dense + BN-statistics epilogue)
 *   kind 1 = xv_gemm_nt_kernel<false, *> / xv_gemm_nt_sk_kernel<false, *> (data gradients, logits)
 *   kind 2 = xv_gemm_tn_kernel        (weight gradients)
 *   kind 3 = xv_gemm16_nt_kernel<true>, kind 4 = xv_gemm16_nt_kernel<false>, kind 5 = xv_gemm16_tn_kernel
 *            (the same three roles on fp16 hi/lo planes, XV_PRECISION_F16X3)
 * xv_profile_end synchronises the recorded events and returns, per kind, the number of launches,
 * their summed duration in ms and their summed algorithmic FLOPs (2*M*N*K each). */
int xv_profile_begin(int max_launches);
/* Same, restricted to the kinds whose bit is set in kind_mask (bit k = kind k): an event pair costs a few
 * microseconds of queue time per launch, so bench.py's timed region brackets only the dominant kernel. */
int xv_profile_begin_kinds(int max_launches, uint32_t kind_mask);
#define XV_PROFILE_KINDS 6
int xv_profile_end(int64_t launches[XV_PROFILE_KINDS], double ms[XV_PROFILE_KINDS], double flops[XV_PROFILE_KINDS]);
/* Diagnostics: the schedule xv_affine_forward / xv_affine_dgrad pick for C[M][N] = A[M][K] . Bt[N][K]^T given ample workspace
 * (stats: the launch emits BatchNorm statistics = a forward launch; co_running: a data gradient beside the weight-gradient stream):
 * 0 one workgroup per tile (xv_gemm_nt_kernel), 1 the evenly scheduled kernel (xv_gemm_nt_sk_kernel), 2 whole tiles + shares of the
 * remaining tiles (xv_gemm_nt_kernel), 3 split-K + slab sum.  tools/pmc_traffic.py attributes layers to kernels with it. */
int xv_debug_nt_schedule(int M, int N, int K, int stats, int co_running);

/* dst[r][0..cols) = src[r][0..cols) for r < rows (device to device, pitches in floats). */
int xv_copy_2d(void* stream, float* dst, size_t ldd, const float* src, size_t lds, int rows, int cols);

/* ---------------------------------------------------------------------------------
 * Op level (each is one or two kernel launches; used by the parity tests and the engine)
 * --------------------------------------------------------------------------------- */

/* Bytes of scratch an op-level call may need (upper bound for any op below at these sizes). */
size_t xv_op_workspace_bytes(int rows, int cols_in, int cols_out);

/* Zero-pad the channel axis: dst[r][0..c_dst) = src[r][0..c_src), 0 beyond.  Used to bring
 * the 30-dim MFCC rows (dataset feed of trainer.py:491) to a 16-byte-aligned pitch. */
int xv_pad_channels(void* stream, const float* src, int rows, int c_src, float* dst, int c_dst);

/* Kaldi 'CM ' compressed-matrix decode (dataset/kaldi_io.py:768-812 the codec, :814-867 the row sub-range) of a batch the native loader
 * delivers undecoded (include/xvector_io.h, xvio_config.packed): b chunks of chunk_stride bytes, each
 *   [min f32][range f32][d x (p0, p25, p75, p100) u16][d x t u8, column after column]
 * -> out [b][t][d] float32, bit-identical to the host decoder and the reference reader.  d <= 128. */
int xv_cm_decode(void* stream, const uint8_t* packed, int b, int t, int d, size_t chunk_stride, float* out);
/* Ragged form for whole utterances of different lengths (batched extraction, extract.py:64-93 / kaldi_io.py:768-812 read_mat_ark): chunk i
 * is a 'CM ' matrix of rows[i] <= t frames exactly as it sits in the archive behind the "CM " token - [min f32][range f32][rows i32]
 * [cols i32][d x (p0, p25, p75, p100) u16][d x rows[i] u8, column after column] - at byte offsets[i] of `packed`; it is decoded into
 * out[i][0 .. rows[i]) of the [b][t][d] tensor and the padding rows behind it are zeroed.  offsets / rows: device arrays [b]. */
int xv_cm_decode_ragged(void* stream, const uint8_t* packed, const int64_t* offsets, const int32_t* rows, int b, int t, int d, float* out);

/* Kernel-layout weights for xv_affine_forward: wt[o][j*c_pad + c] = kernel[j][c][o]
 * (TF layout [k][C][O] of tdnn/tdnnX_{conv,dense}/kernel, tdnn.py:39,57,75,96,115,147,166);
 * columns c in [C, c_pad) are zero. */
int xv_prep_weight_fwd(void* stream, const float* kernel, int k, int c, int o, float* wt, int c_pad);
/* Kernel-layout weights for xv_affine_dgrad: wf[c][(k-1-j)*O + o] = kernel[j][c][o]. */
int xv_prep_weight_dgrad(void* stream, const float* kernel, int k, int c, int o, float* wf);

/* z[(b,t)][o] = bias[o] + sum_{j<k} sum_c x[b][t+j][c] * kernel[j][c][o]
 *   == tf.layers.conv2d(.., (1,k)) VALID, tdnn.py:39-44,57-62,75-80 and tf.layers.dense,
 *   tdnn.py:96-100,115-119,147-151,166-170 (k = 1, segs = rows, t_in = 1).
 * x: [segs][t_in][c_pad] spliced view, wt from xv_prep_weight_fwd, z: [segs*(t_in-k+1)][ldz].
 * If bn_part != NULL it receives per-128-row-tile column statistics of z (sum and centred
 * sum of squares, min, max: layout [4][tiles_m][o]) for xv_bn_finalize - the batch statistics of
 * tf.layers.batch_normalization(training=True), tdnn.py:46.  `ws` is scratch
 * (xv_op_workspace_bytes) used when the launcher splits the reduction.
 * Every GEMM operand (x, wt, dz - in split precision: each fp16 plane) must span less than 4 GB: the kernels address operand rows as
 * 32-bit byte offsets from the operand's base (an error is returned otherwise: split the batch). */
int xv_affine_forward(void* stream, const float* x, int segs, int t_in, int c_pad, int k,
                      const float* wt, const float* bias, float* z, int o, int ldz,
                      float* bn_part, void* ws, size_t ws_bytes);

/* dx[(b,s)][c] = sum_j sum_o dz[b][s-j][o] * kernel[j][c][o]  (gradient of the above w.r.t. x).
 * dz_pad: [segs][t_out + 2(k-1)][o] with k-1 ZERO rows before and after each segment
 * (written by xv_bn_backward_apply); wf from xv_prep_weight_dgrad ([c][k*o]); dx: [segs*(t_out+k-1)][c]. */
int xv_affine_dgrad(void* stream, const float* dz_pad, int segs, int t_out, int o, int k,
                    const float* wf, float* dx, int c, void* ws, size_t ws_bytes);

/* dkernel[j][c][o] = sum_{b,t} x[b][t+j][c] * dz[b][t][o]  (+ l2_scale * kernel[j][c][o],
 *   the gradient of tf.contrib.layers.l2_regularizer, tdnn.py:43 / trainer.py:357-358).
 * x: spliced view [segs][t_in][c_pad]; dz rows live in a buffer with `dz_seg_pitch` rows per
 * segment starting at row `dz_row0` (so the padded dz of xv_affine_dgrad can be used directly).
 * Output in TF layout [k][c][o] (padded channels dropped). */
int xv_affine_wgrad(void* stream, const float* x, int segs, int t_in, int c_pad, int k, int c,
                    const float* dz, int dz_seg_pitch, int dz_row0, int o,
                    const float* kernel, float l2_scale, float* dkernel,
                    void* ws, size_t ws_bytes);

/* out[n] = sum_r a[r][n]  (bias gradient, tf.layers bias_add grad). a: [rows][lda]. */
int xv_colsum(void* stream, const float* a, int rows, int n, int lda, float* out, void* ws, size_t ws_bytes);
/* Per-tile column statistics in the xv_affine_forward bn_part layout, for tensors that were
 * produced by a split reduction (segment-level layers). */
int xv_col_stats(void* stream, const float* z, int rows, int n, int ldz, float* bn_part);

/* tf.layers.batch_normalization(training=True) statistics, tdnn.py:46,64,82,102,121,153,177:
 * combine bn_part -> mean, biased var; scale = gamma*rsqrt(var+eps), shift = beta-mean*scale;
 * moving <- moving*momentum + batch*(1-momentum) (unbiased var if unbiased_moving != 0).
 * Outputs: mean[n], invstd[n], scale[n], shift[n]; optional zmin[n]/zmax[n] (range of z per channel) and
 * *amax |= max over the tensor of relu?(z*scale+shift) as float bits (atomicMax; zero it first) - the
 * exact output range the split-precision path scales its fp16 operand planes with. */
int xv_bn_finalize(void* stream, const float* bn_part, int rows, int n,
                   const float* gamma, const float* beta, float eps, float momentum, int unbiased_moving,
                   float* moving_mean, float* moving_var,
                   float* mean, float* invstd, float* scale, float* shift,
                   float* zmin, float* zmax, uint32_t* amax, int relu);
/* Inference statistics (training=False): scale/shift from the moving averages. */
int xv_bn_inference_scale(void* stream, int n, const float* gamma, const float* beta,
                          const float* moving_mean, const float* moving_var, float eps,
                          float* scale, float* shift);
/* a = relu?(z*scale + shift)   (tdnn.py:46-52: BN then tf.nn.relu). relu != 0 applies ReLU. */
int xv_bn_apply(void* stream, const float* z, int rows, int n, int ldz, const float* scale, const float* shift,
                int relu, float* a, int lda);
/* Backward of ReLU(BN(z)) given da (both dense [segs*t][n]): dy = da * (y>0); dgamma = sum dy*xhat; dbeta = sum dy;
 * dz = gamma*invstd*(dy - dbeta/rows - xhat*dgamma/rows) written into a segment-padded buffer
 * dz_pad [segs][t + 2*pad][n] (pad rows zeroed).  relu != 0 means a ReLU follows the BN.
 * dbias (optional): gradient of a bias added in FRONT of this BN = column sum of dz.  It is 0 in exact
 * arithmetic (BN removes the mean); TF's reduce_sum(dz) leaves rounding noise there, and so does this
 * (gamma*invstd*(sum dy - rows*mean dy)) without a separate pass over dz. */
int xv_bn_relu_backward(void* stream, const float* da, const float* z, int segs, int t, int n,
                        const float* gamma, const float* mean, const float* invstd,
                        const float* scale, const float* shift, int relu, int pad,
                        float* dz_pad, float* dgamma, float* dbeta, float* dbias, void* ws, size_t ws_bytes);
/* prelu, common.py:27-42: y[r][c] = relu(x) + alpha[c] * (x - |x|) / 2 = x > 0 ? x : alpha[c] * x  (tf.nn.leaky_relu: alpha = 0.2 everywhere).
 * Inside the engine the same non-linearity follows every BatchNorm when xv_config.relu_type says so. */
int xv_prelu_forward(void* stream, const float* x, int rows, int n, const float* alpha, float* y);
/* The activation that follows a BatchNorm in every op-level entry point that takes a `relu` flag (xv_bn_apply, xv_bn_apply_split,
 * xv_bn_relu_backward[_split|_pooled*], xv_stat_pool_forward_bn*, xv_segment_affine_bn_forward, xv_segment_dgrad_bn_backward,
 * xv_att_pool_backward_weights): network_relu_type of tdnn.py:24-30.
 *   slope == NULL (the default): tf.nn.relu.
 *   slope != NULL: y > 0 ? y : slope[c] * y with one device float per channel - prelu (common.py:27-42, slope = the layer's
 *     <layer>_relu/alpha variable; dalpha, if not NULL, receives d alpha[c] = sum over rows of da * min(y, 0) from the backward
 *     entry points) or lrelu (a constant 0.2 vector, dalpha = NULL).
 * The setting belongs to the CALLING THREAD and stays until the next call; set it around the calls of one layer and reset it with
 * (NULL, NULL).  The engine does exactly that per layer, so engine-level calls ignore and clear whatever was set here. */
int xv_set_activation(const float* slope, float* dalpha);
/* Backward of a bare ReLU (no BN in front): dz = da * (a > 0). */
int xv_relu_backward(void* stream, const float* da, const float* a, size_t count, float* dz);

/* ---------------------------------------------------------------------------------
 * Split precision ("f16x3"): the big TDNN contractions on the 16-bit matrix cores at fp32-class accuracy.
 * An fp32 tensor is carried as two fp16 planes [2][rows][ld] (ld multiple of 8, zero padded):
 * x*s = hi + lo with a power-of-two scale s derived from the tensor's max |x|, which travels in device
 * memory as the uint32 bits of a float (`amax`; producers atomicMax into it, so zero it first).  Products
 * hi.hi + hi.lo + lo.hi are accumulated in fp32 and multiplied by 1/(sA*sB) (2^-22 relative per product).
 * Same reference call sites as the fp32 entry points they mirror.
 * --------------------------------------------------------------------------------- */
int xv_amax(void* stream, const float* x, size_t count, uint32_t* amax_accum);
int xv_split_planes(void* stream, const float* src, int rows, int c, int lds, void* planes, int ldp, size_t plane_stride,
                    const uint32_t* amax);
/* planes <- relu?(z*scale+shift): BN(+ReLU) output written directly as the next layer's operand (tdnn.py:46-52) */
int xv_bn_apply_split(void* stream, const float* z, int rows, int n, int ldz, const float* scale, const float* shift, int relu,
                      const uint32_t* amax, void* planes, int ldp, size_t plane_stride);
/* Range of z (per channel) and of relu?(z*scale+shift) (whole tensor, into *amax) from the bn_part min/max planes,
 * for scale/shift that did not come from xv_bn_finalize (training=False: moving statistics). */
int xv_bn_output_range(void* stream, const float* bn_part, int rows, int n, const float* scale, const float* shift, int relu,
                       float* zmin, float* zmax, uint32_t* amax);
/* xv_bn_relu_backward with dz written as planes in the segment-padded layout; *dz_amax receives an upper bound of
 * max |dz| computed from per-channel max |dy| and the forward range zmin/zmax (xv_bn_finalize). */
int xv_bn_relu_backward_split(void* stream, const float* da, const float* z, int segs, int t, int n, const float* gamma,
                              const float* mean, const float* invstd, const float* scale, const float* shift,
                              const float* zmin, const float* zmax, int relu, int pad, void* dz_planes, int ldp,
                              size_t plane_stride, uint32_t* dz_amax, float* dgamma, float* dbeta, float* dbias, void* ws,
                              size_t ws_bytes);

/* xv_bn_relu_backward / _split for the layer whose output feeds statistics pooling (tdnn5): the upstream gradient is not
 * read from memory but evaluated from the pooled statistics pool_out [b][mean | std] (xv_stat_pool_forward_bn) and their
 * gradient dpool [b][2n] -  d a = dmean/t + dstd/(t*std) * (a - mean), a = relu?(z*scale + shift), zero where the
 * variance was clamped (pooling.py:28-29) - i.e. pooling backward + ReLU backward + BN backward in one pass over z.
 * rows = b*t, no segment padding. */
int xv_bn_relu_backward_pooled(void* stream, const float* pool_out, const float* dpool,
                               const float* weights /* [b*t] attention weights or NULL: d a = w*(dmean + dstd/std*(a - mean)) */, int b, int t,
                               const float* z, int n,
                               const float* gamma, const float* mean, const float* invstd, const float* scale, const float* shift,
                               int relu, float* dz, float* dgamma, float* dbeta, float* dbias, void* ws, size_t ws_bytes);
/* The same pair with the pooling forward's by-product wpos [b][c] (the share of each chunk's frame weights on frames whose ReLU is
 * on; amax [b][c], optional: each chunk's largest activation): given wpos, the BatchNorm backward gets sum(dy) and sum(dy*xhat) in
 * closed form from the pooled statistics - no reduction pass over z (plain ReLU or no activation).  Same results up to rounding. */
int xv_stat_pool_forward_bn_aux(void* stream, const float* z, int b, int t, int c, const float* scale, const float* shift, int relu,
                                const float* weights, float* out, float* wpos, float* amax);
int xv_bn_relu_backward_pooled_aux(void* stream, const float* pool_out, const float* dpool, const float* weights, const float* wpos,
                                   int b, int t, const float* z, int n, const float* gamma, const float* mean, const float* invstd,
                                   const float* scale, const float* shift, int relu, float* dz, float* dgamma, float* dbeta,
                                   float* dbias, void* ws, size_t ws_bytes);
int xv_bn_relu_backward_pooled_split(void* stream, const float* pool_out, const float* dpool, const float* weights, int b, int t,
                                     const float* z, int n,
                                     const float* gamma, const float* mean, const float* invstd, const float* scale,
                                     const float* shift, const float* zmin, const float* zmax, int relu, void* dz_planes, int ldp,
                                     size_t plane_stride, uint32_t* dz_amax, float* dgamma, float* dbeta, float* dbias, void* ws,
                                     size_t ws_bytes);

/* xv_affine_forward / _dgrad / _wgrad on planes.  c_ld / o_ld: plane row pitches (multiples of 8). */
int xv_affine_forward_f16x3(void* stream, const void* x_planes, size_t x_plane_stride, const uint32_t* x_amax, int segs, int t_in,
                            int c_ld, int k, const void* wt_planes, size_t wt_plane_stride, const uint32_t* wt_amax,
                            const float* bias, float* z, int o, int ldz, float* bn_part);
int xv_affine_dgrad_f16x3(void* stream, const void* dz_planes, size_t dz_plane_stride, const uint32_t* dz_amax, int segs, int t_out,
                          int o_ld, int k, const void* wf_planes, size_t wf_plane_stride, const uint32_t* wf_amax, float* dx, int c);
/* The same with the BN-backward reductions of the layer that owns dx fused into the epilogue: dx is d(relu(bn(z_below)));
 * part [ceil(rows/128)][3][c] = per-tile sum dd | sum dd*xhat | max |dd| with dd = dx masked by the ReLU - what the first pass of
 * xv_bn_relu_backward_split would compute by re-reading dx and z_below.  Feed it to xv_bn_relu_backward_split_from_part. */
int xv_affine_dgrad_bnstats_f16x3(void* stream, const void* dz_planes, size_t dz_plane_stride, const uint32_t* dz_amax, int segs, int t_out,
                                  int o_ld, int k, const void* wf_planes, size_t wf_plane_stride, const uint32_t* wf_amax, float* dx, int c,
                                  const float* z_below, const float* scale, const float* shift, const float* mean, const float* invstd,
                                  float* part);
int xv_bn_relu_backward_split_from_part(void* stream, const float* part, int chunks, const float* da, const float* z, int segs, int t, int n,
                                        const float* gamma, const float* mean, const float* invstd, const float* scale,
                                        const float* shift, const float* zmin, const float* zmax, int pad, void* dz_planes, int ldp,
                                        size_t plane_stride, uint32_t* dz_amax, float* dgamma, float* dbeta, float* dbias, void* ws,
                                        size_t ws_bytes);
int xv_affine_wgrad_f16x3(void* stream, const void* x_planes, size_t x_plane_stride, const uint32_t* x_amax, int segs, int t_in,
                          int c_ld, int k, int c, const void* dz_planes, size_t dz_plane_stride, const uint32_t* dz_amax,
                          int dz_seg_pitch, int dz_row0, int o_ld, int o, const float* kernel, float l2_scale, float* dkernel,
                          void* ws, size_t ws_bytes);

/* statistics_pooling, pooling.py:9-34: out[b] = concat(mean_t x, sqrt(max-masked var_t x)). */
int xv_stat_pool_forward(void* stream, const float* x, int b, int t, int c, float* out);
int xv_stat_pool_backward(void* stream, const float* x, const float* out, const float* dout,
                          int b, int t, int c, float* dx);

/* Statistics pooling over relu?(z*scale + shift) evaluated on the fly (tdnn5's BN + ReLU, tdnn.py:124-131, fused into
 * pooling.py:9-34): out[b] = [mean_t | std_t] of the activation, which is never written to memory. */
int xv_stat_pool_forward_bn(void* stream, const float* z, int b, int t, int c, const float* scale, const float* shift, int relu,
                            const float* weights /* [b*t] frame weights summing to 1 per chunk (self-attention), or NULL = 1/t */,
                            float* out);


/* ---- auxiliary losses (model/loss.py:985-1036; shipped in nnet_conf/..._r0.01.json and ..._mhe0.01.json) ----
 * ring loss: lambda * mean((||x|| - r)^2), r a trainable scalar ("softmax_ringloss/r").  Adds the value to *loss_accum,
 * lambda*2*(||x||-r)/rows to dnorm_accum[row] (what xv_add_norm_grad turns into d x) and writes d r.
 * MHE: lambda / (mean_{b,n}(2 - 2 wn[:,y_b].wn[:,n]) + 1e-6) on the column-normalised weights wn [c][ldn]; xv_mhe_loss adds the value
 * and leaves [g | u[c] | v[c]] in coef (1 + 2c floats) and the label histogram in counts (n ints); xv_mhe_add_grad adds
 * g*(u + counts[n]*v) to the gradient w.r.t. wn (before xv_loss_weight_backward). */
int xv_ring_loss(void* stream, const float* x, int rows, int n, int ldx, const float* r, float lambda, float* loss_accum,
                 float* dnorm_accum, float* dr);
int xv_mhe_loss(void* stream, const float* wn, int c, int n, int ldn, const int32_t* labels, int rows, float lambda, float* loss_accum,
                float* coef, int32_t* counts);
int xv_mhe_add_grad(void* stream, float* dwn, int c, int n, int ldn, const float* coef, const int32_t* counts);

/* ---- self-attention pooling, the shipped single-head form (model/pooling.py:37-192; nnet_conf/..._tdnn4_att.json) ----
 * The key network's dense layers run on the frame-level GEMMs; these are the pieces around them.
 *   score[r]   = scale * sum_c act(zk[r][c]) * query[c]        act: 0 = identity, 1 = relu, 3 = tanh (att_key_network_type, pooling.py:84-96)
 *   weights    = softmax over the t frames of each chunk        (pooling.py:148)
 *   pooled     = xv_stat_pool_forward_bn(..., weights, ...)     (pooling.py:151-164)
 * backward: d weights from the pooled statistics (the value path enters tdnn5's BN backward through
 * xv_bn_relu_backward_pooled*'s `weights`), softmax backward, then through the score into the key layer output:
 *   dzk[r][c] = dscore[r]*scale*query[c]*act'(zk),  dquery[c] = sum_r dscore[r]*scale*act(zk[r][c]),  dbias[c] = sum_r dzk[r][c]. */
int xv_att_score(void* stream, const float* zk, int rows, int n, int ldz, int act, const float* query, float scale, float* score);
int xv_softmax_segments(void* stream, const float* score, int b, int t, float* weights);
int xv_softmax_segments_backward(void* stream, const float* weights, const float* dweights, int b, int t, float* dscore);
int xv_att_pool_backward_weights(void* stream, const float* z, int b, int t, int c, const float* scale, const float* shift, int relu,
                                 const float* pool_out, const float* dpool, float* dweights);
int xv_att_key_backward(void* stream, const float* zk, int rows, int n, int act, const float* query, float scale, const float* dscore,
                        float* dzk, float* dquery, float* dbias, void* ws, size_t ws_bytes);
/* y[i] = act(z[i]), act as in xv_att_score (0 identity, 1 relu, 3 tanh): the "<name>_relu" / "<name>_tanh" endpoints of
 * common.py:150-213 dense_relu / dense_tanh for callers of pooling.self_attention (pooling.py:84-96). */
int xv_key_activation(void* stream, const float* z, size_t count, int act, float* y);
/* y[i] += x[i]  (the two gradient paths into tdnn4_relu: through tdnn5 and through the attention key network) */
int xv_add_inplace(void* stream, float* y, const float* x, size_t count);

/* l2_scaling, common.py:45-58 (feature_norm:true, trainer.py:183-186). */
int xv_l2_scaling_forward(void* stream, const float* x, int rows, int n, float factor, float* y);
int xv_l2_scaling_backward(void* stream, const float* x, const float* dy, int rows, int n, float factor, float* dx);

/* Loss family, loss.py:9-355. */
enum { XV_LOSS_SOFTMAX = 0, XV_LOSS_ASOFTMAX = 1, XV_LOSS_AMSOFTMAX = 2, XV_LOSS_ARCSOFTMAX = 3 };
/* ---- segment-level layers in one launch (rows <= XV_SEGMENT_MAX_ROWS chunks of a batch) ------------------------------
 * tf.layers.dense at tdnn.py:147,166 and the logits / gradient products of loss.py with M = chunks rows: the GEMM, its split-K sum
 * and the consumer's per-column work run in one launch.  ws: slabs (xv_op_workspace_bytes); tickets: one ZEROED uint32 per 32
 * output columns, left zeroed.  Optional row term (the ||x|| gradient of loss.py:122,147, what xv_add_norm_grad adds):
 * acc[m][:] += (row_norm[m] > 0 ? row_coef[m] / row_norm[m] : 0) * xrow[m][:]. */
#define XV_SEGMENT_MAX_ROWS 128
/* c[m][n] = sum_k a[m][k] * bt[n][k] + bias[n] (+ row term) */
int xv_segment_gemm(void* stream, const float* a, long lda, const float* bt, long ldb, int m, int n, int k, const float* bias,
                    const float* row_coef, const float* row_norm, const float* xrow, long ldx, float* c, long ldc,
                    void* ws, size_t ws_bytes, uint32_t* tickets);
/* dense + training-mode tf.layers.batch_normalization (+ activation) of tdnn.py:147-189: z = x . wt^T + bias, batch statistics over
 * the m rows (biased two-pass variance), moving averages, scale/shift, a = act(z*scale + shift) (a may be NULL).  Same arithmetic
 * as xv_affine_forward + xv_bn_finalize + xv_bn_apply (relu != 0: ReLU). */
int xv_segment_affine_bn_forward(void* stream, const float* x, long ldx, const float* wt, long ldw, int m, int n, int k,
                                 const float* bias, const float* gamma, const float* beta, float eps, float momentum,
                                 int unbiased_moving, float* moving_mean, float* moving_var, float* z, float* mean, float* invstd,
                                 float* scale, float* shift, int relu, float* a, void* ws, size_t ws_bytes, uint32_t* tickets);
/* d a = dy . wt^T (+ row term), then the backward of the BatchNorm (+ activation) layer whose pre-BN tensor is z [m][n]:
 * dz, dgamma, dbeta, dbias (xv_bn_relu_backward's arithmetic on m rows). */
int xv_segment_dgrad_bn_backward(void* stream, const float* dy, long lddy, const float* wt, long ldw, int m, int n, int k,
                                 const float* row_coef, const float* row_norm, const float* xrow, long ldx, const float* z,
                                 const float* gamma, const float* mean, const float* invstd, const float* scale, const float* shift,
                                 int relu, float* dz, float* dgamma, float* dbeta, float* dbias, void* ws, size_t ws_bytes,
                                 uint32_t* tickets);

/* tf.nn.l2_normalize(w, dim=0), loss.py:104: inv_norm[n], wn[c][ldn] = w*inv, wnt[n][c] = wn^T.
 * normalize == 0 copies unnormalised (plain softmax, loss.py:30). */
int xv_loss_prep_weight(void* stream, const float* w, int c, int n, int normalize,
                        float* inv_norm, float* wn, int ldn, float* wnt);
/* Given logits = x . wn (+bias) [rows][ldl]: margin transform of the target logit, lambda
 * blend (fa = 1/(1+lambda)), mean sparse softmax cross entropy, and the gradients
 * dlogits [rows][ldl] (pad columns zeroed) and dnorm[rows] (= dL/d||x||, 0 for softmax).
 * loss_out: one float (mean over rows).  asoftmax: m in {1,2,4}. */
int xv_margin_softmax_rows(void* stream, int kind, const float* logits, int rows, int n, int ldl,
                           const float* x, int c, const int32_t* labels, float m, float lambda,
                           float* dlogits, float* dnorm, float* row_loss, float* loss_out);
/* dx[r][:] += dnorm[r] * x[r][:] / ||x[r]||  (the ||x|| paths of loss.py:122,147). */
int xv_add_norm_grad(void* stream, const float* x, const float* dnorm, int rows, int c, float* dx);
/* Gradient through l2_normalize: dw = inv*(dwn - wn*colsum(dwn*wn)) + l2_scale*w. */
int xv_loss_weight_backward(void* stream, const float* dwn, int lddwn, const float* wn, int ldn,
                            const float* inv_norm, const float* w, int c, int n, int normalize,
                            float l2_scale, float* dw, void* ws, size_t ws_bytes);

/* sum over a flat range of 0.5*scale*w^2 accumulated into *out (regularization_loss). */
int xv_l2_reg_loss(void* stream, const float* w, size_t count, float scale, float* out_accum);

/* Optimisers, trainer.py:332-346.  g is scaled by grad_scale first (1/world for DP). */
int xv_sgd_update(void* stream, float* p, const float* g, size_t count, float lr, float grad_scale);
int xv_momentum_update(void* stream, float* p, const float* g, float* acc, size_t count,
                       float lr, float momentum, int nesterov, float grad_scale);
int xv_adam_update(void* stream, float* p, const float* g, float* m, float* v, size_t count,
                   float lr, float beta1, float beta2, float eps, int t, float grad_scale);
/* sum of squares of a flat range accumulated into *out (for tf.clip_by_global_norm). */
int xv_sumsq(void* stream, const float* g, size_t count, float* out_accum);

/* ---------------------------------------------------------------------------------
 * Engine level: the whole tdnn + loss graph of Trainer.build / sess.run(train_op)
 * --------------------------------------------------------------------------------- */
typedef struct xv_engine xv_engine;

typedef struct xv_config {
    int32_t struct_bytes;             /* = sizeof(xv_config) of the header the host was built against; anything else is refused */
    int32_t feat_dim;                 /* dim, train.py:71 */
    int32_t num_speakers;             /* train.py:74 (0 => no loss head, predict only) */
    int32_t num_nodes_pooling_layer;  /* tdnn.py:111-113, default 1500 */
    int32_t num_nodes_last_layer;     /* tdnn.py:162-164, default 512 */
    int32_t last_layer_no_bn;         /* tdnn.py:173-174 */
    int32_t last_layer_linear;        /* tdnn.py:183-184 */
    int32_t feature_norm;             /* trainer.py:183 */
    float feature_scaling_factor;
    int32_t loss_kind;                /* XV_LOSS_* (trainer.py:234-250) */
    float margin_m;                   /* asoftmax_m / amsoftmax_m / arcsoftmax_m */
    float lambda_min, lambda_base, lambda_gamma, lambda_power; /* loss.py:144-145 */
    float weight_l2_regularizer;      /* tdnn.py:43 */
    float output_weight_l2_regularizer; /* loss.py:26-28; < 0 => use weight_l2_regularizer */
    float batchnorm_momentum;         /* tdnn.py:47 */
    float bn_epsilon;                 /* [TF] 1e-3 */
    int32_t fused_bn_unbiased_moving_var; /* SURVEY N4 switch (layers 1-3) */
    int32_t optimizer;                /* 0 sgd, 1 momentum, 2 adam (trainer.py:332-346) */
    float momentum;
    int32_t use_nesterov;
    float clip_gradient_norm;         /* trainer.py:408-410; <= 0 => clip_gradient:false */
    int32_t max_batch;                /* capacity: chunks per step */
    int32_t max_frames;               /* capacity: frames per chunk */
    int32_t precision;                /* XV_PRECISION_*: how the tdnn1-5 contractions are evaluated */
    int32_t pooling;                  /* XV_POOL_*: pooling_type, tdnn.py:133-138 */
    int32_t att_key0_nodes;           /* att_key_num_nodes[0]: dense+bn+relu on tdnn4_relu (pooling.py:78-82) */
    int32_t att_key1_nodes;           /* att_key_num_nodes[1]: the key dimension (pooling.py:84-96) */
    int32_t att_key_type;             /* att_key_network_type of the last key layer: 0 affine, 1 + relu, 2 + bn + relu, 3 + tanh */
    int32_t att_use_scale;            /* att_use_scale: scores / sqrt(key dim) (pooling.py:144-145) */
    int32_t aux_ring;                 /* "ring_loss" in aux_loss_func (loss.py:1003-1017); adds the variable softmax_ringloss/r */
    float ring_loss_init;             /* initial r (set by the host initialiser; the engine does not read it) */
    float ring_loss_lambda;
    int32_t aux_mhe;                  /* "mhe_loss" in aux_loss_func (loss.py:1018-1033); margin losses only (normalised weights) */
    float mhe_lambda;
    /* Frame-layer table.  0 = the reference's five layers (tdnn.py:35-127: contexts 5/5/7/1/1, widths 512 x 4 + num_nodes_pooling_layer).
     * Otherwise 3..XV_MAX_FRAME_LAYERS layers of (context k >= 1 contiguous frames, output channels); context 1 = dense.  Variables
     * are named tdnn<i>_conv / tdnn<i>_dense / tdnn<i>_bn as in the reference; the two segment-level layers follow as tdnn<F+1>,
     * tdnn<F+2>.  Extended tables have NO reference counterpart (BASELINE configs[4]). */
    int32_t num_frame_layers;
    int32_t frame_context[12];
    int32_t frame_width[12];
    int32_t relu_type;                /* XV_RELU_*: network_relu_type, tdnn.py:24-30 (no shipped config sets it) */
    /* capacity in rows = chunks x frames of one forward; 0 = max_batch * max_frames.  Batched extraction sets it (with a large
     * max_frames): a batch is many short utterances or a few long ones, never max_batch chunks of max_frames frames */
    int32_t max_rows;
} xv_config;
#define XV_MAX_FRAME_LAYERS 12
/* the non-linearity behind every BatchNorm: relu | prelu = relu(x) + alpha (x - |x|) / 2 with a trainable per-channel alpha
 * "<prefix>_relu/alpha" initialised to 0.01 (common.py:27-42) | tf.nn.leaky_relu (alpha 0.2) */
#define XV_RELU_RELU 0
#define XV_RELU_PRELU 1
#define XV_RELU_LRELU 2
#define XV_POOL_STATISTICS 0
/* self_attention in the shipped single-head form (nnet_conf/..._tdnn4_att.json): key network on tdnn4_relu, value = tdnn5_relu,
 * one head, key not split, no value network, no penalty term, no post non-linearity */
#define XV_POOL_SELF_ATTENTION 1
/* fp32-input MFMA (v_mfma_f32_32x32x2_f32, exact fp32 products) */
#define XV_PRECISION_F32 0
/* split precision: three fp16 MFMA products of (hi,lo) pieces per fp32 product, fp32 accumulate (2^-22 relative) */
#define XV_PRECISION_F16X3 1

int xv_engine_create(const xv_config* cfg, xv_engine** out);
void xv_engine_destroy(xv_engine* e);

/* Variable table (TF names/shapes in graph order, e.g. "tdnn/tdnn1_conv/kernel" [1,5,D,512]). */
int xv_engine_num_variables(const xv_engine* e);
/* name: pointer to a static string; shape: up to 4 dims written, returns rank via *rank;
 * offset in floats into the variables buffer; trainable flag. */
int xv_engine_variable_info(const xv_engine* e, int index, const char** name, int32_t shape[4],
                            int32_t* rank, size_t* offset, int32_t* trainable);
/* Flat sizes in floats: all variables (trainable first, then BN moving statistics),
 * trainable only (= gradient buffer), optimiser state. */
size_t xv_engine_variables_count(const xv_engine* e);
size_t xv_engine_trainable_count(const xv_engine* e);
size_t xv_engine_optimizer_state_count(const xv_engine* e);
/* Caller-owned device buffers (e.g. torch tensors): variables, gradients, optimiser state. */
int xv_engine_bind(xv_engine* e, float* variables, float* grads, float* opt_state);

/* tdnn(features) (+ entire_network's l2_scaling).  features: [b][t][feat_dim].
 * training != 0: batch statistics + moving-average update (is_training=True). */
int xv_engine_forward(xv_engine* e, void* stream, const float* features, int b, int t, int training);
/* Inference forward (is_training=False, trainer.py:708-726) over a batch of utterances of DIFFERENT lengths, the batched form of the loop
 * extract.py:64-93 runs one utterance at a time: chunk i of features [b][t][feat_dim] holds frames[i] valid frames (receptive field <=
 * frames[i] <= t) followed by padding (finite values, e.g. zeros).  The frame layers are valid convolutions, so the first
 * frames[i] - (receptive field - 1) output frames of chunk i do not see the padding; pooling (and the attention softmax) use exactly
 * those.  Segment-level endpoints ("pooling", "tdnn6_dense", ...) are per chunk as usual; frame-level endpoints keep their padded rows.
 * frames: device array [b]. */
int xv_engine_forward_lengths(xv_engine* e, void* stream, const float* features, int b, int t, const int32_t* frames);
/* loss_network(features, labels) + gradients of loss + regulariser w.r.t. every trainable
 * variable into the bound gradient buffer.  global_step feeds the lambda schedule
 * (trainer.py:505-508).  stage: -1 = everything; 0..XV_BWD_STAGES-1 = that slice only (lets the
 * host overlap the gradient all-reduce of finished slices with the rest of the backward). */
#define XV_BWD_STAGES 4
int xv_engine_loss_forward(xv_engine* e, void* stream, const int32_t* labels, int global_step, int with_margin);
int xv_engine_backward(xv_engine* e, void* stream, int stage);
/* Staged backward without the join at the end of stages 0..2: `stream` does not wait for the engine's weight-gradient stream,
 * so the next stage's data-gradient chain keeps overlapping it.  The slice of stage k is complete on any stream that called
 * xv_engine_stage_wait(e, that_stream, k) - typically the communication stream the slice's all-reduce is enqueued on.  The last
 * stage joins on `stream` as xv_engine_backward does (xv_engine_apply may follow on `stream`); a different consumer stream
 * still has to call xv_engine_stage_wait for it. */
int xv_engine_backward_async(xv_engine* e, void* stream, int stage);
int xv_engine_stage_wait(xv_engine* e, void* waiter_stream, int stage);
/* Data parallelism for hosts without torch.distributed (the reference has no multi-GPU path: model/trainer.py:349 withholds its
 * tower design): sum-all-reduce, in place, of the gradient slice that backward stage `stage` completed - after
 * xv_engine_backward_async(e, compute_stream, stage) - over the RCCL communicator the host created (ncclComm_t passed as void*; one
 * rank per GPU), enqueued on `comm_stream` behind the stage's completion events, so the collective of stage k overlaps the backward
 * of stages k+1...  Call it for stages 0..XV_BWD_STAGES-1 in order (largest slice - the speaker matrix - first), then
 * xv_engine_allreduce_wait(e, compute_stream) and xv_engine_apply(..., grad_scale = 1 / world size, ...).  BatchNorm statistics stay
 * local to each rank.  RCCL is looked up in the process at first use (no link-time dependency). */
int xv_engine_allreduce(xv_engine* e, void* comm_stream, int stage, void* rccl_comm);
/* `stream` waits for every all-reduce enqueued so far through xv_engine_allreduce. */
int xv_engine_allreduce_wait(xv_engine* e, void* stream);
/* [begin,end) float range of the gradient buffer completed by backward stage `stage`. */
int xv_engine_stage_grad_range(const xv_engine* e, int stage, size_t* begin, size_t* end);
/* optimiser step on the bound buffers (after the gradient all-reduce). t = 1-based update count. */
int xv_engine_apply(xv_engine* e, void* stream, float lr, float grad_scale, int t);
/* bytes of device memory the engine's arena holds (activations, gradients' scratch, dz slots, workspaces) - diagnostics and tests */
size_t xv_engine_arena_bytes(const xv_engine* e);
/* scalars of the last step, device pointers to 1 float each: raw loss, regularisation loss */
int xv_engine_loss_ptrs(xv_engine* e, float** raw_loss, float** reg_loss);
/* endpoint by reference name ("tdnn1_conv", ..., "pooling", "tdnn6_dense", "output", "logits"):
 * device pointer, rows, cols, leading dimension of the most recent forward. */
int xv_engine_endpoint(xv_engine* e, const char* name, float** ptr, int32_t* rows, int32_t* cols, int32_t* ld);
/* 1 (default): weight gradients run on an internal second HIP stream beside the data-gradient chain;
 * 0: everything in order on the caller's stream (used to time kernels in isolation). */
int xv_engine_set_concurrency(xv_engine* e, int enabled);
/* Mark kernel-layout weight copies stale (call after writing the variables buffer directly). */
int xv_engine_invalidate_weights(xv_engine* e);

#ifdef __cplusplus
}
#endif
#endif /* XVECTOR_HIP_H */

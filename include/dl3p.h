/*
 * dl3p.h -- C ABI of libdl3p.so: the MI355X (gfx950) native DeepLabV3+ forward/backward hot path.
 *
 * The reference (david8862/tf-keras-deeplabv3p-model-set) has no FFI of its own: its boundary is the
 * Python factory get_deeplabv3p_model() (deeplabv3p/model.py:51) and all arithmetic happens inside
 * tf.keras layers.  Each entry point below replaces the TensorFlow op that one reference call site
 * lowers to; the citation names that call site.
 *
 * Conventions
 *   - all tensors are fp32, NHWC, addressed as rows of pixels: element (row m, channel c) lives at
 *     base[m*ld + c]; `ld` (row stride in floats) lets an op read/write a channel slice of a wider
 *     concat buffer (layers.py:155,214 Concatenate costs no pass).
 *   - raw DEVICE pointers; the caller allocates and owns every buffer (torch caching allocator);
 *     the library keeps no references and allocates nothing.
 *   - every call is asynchronous and ordered on `stream` (a hipStream_t passed as void*); calls are
 *     hipGraph-capture safe (no allocation, no synchronisation).
 *   - returns 0 on success or a negative DL3P_E* code; dl3p_last_error_string() describes the last
 *     failure of the calling thread.  Nothing throws or exits across this boundary.
 *   - "prologue": a conv input may be given as the RAW output z of the producing conv together with
 *     the per-channel affine (in_scale, in_shift) of the BatchNormalization that follows it and an
 *     activation code; the kernel then consumes act(z*scale+shift) on the fly, so normalised
 *     activations are never materialised (layers.py:102-108 BN+ReLU between convs).
 *     in_scale == NULL means identity affine.
 *   - "stat partials": producers emit per-block partial sums [rows][2][C] (sum, sum of squares) of
 *     their raw output; dl3p_bn_finalize reduces them.  *rows_out receives the number of rows the
 *     launch will write (a host-side value, known before the kernel runs).
 */
#ifndef DL3P_H_
#define DL3P_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DL3P_VERSION 100

enum {
  DL3P_OK = 0,
  DL3P_EINVAL = -1,    /* bad argument (shape, alignment, null pointer) */
  DL3P_EUNSUPPORTED = -2,
  DL3P_ELAUNCH = -3,   /* hipLaunch / runtime error */
  DL3P_EWORKSPACE = -4 /* workspace too small */
};

/* activation codes (ReLU: layers.py:98; ReLU6: deeplabv3p_mobilenetv2.py:52;
 * hard_swish / hard_sigmoid: deeplabv3p_mobilenetv3.py:98-103) */
enum { DL3P_ACT_NONE = 0, DL3P_ACT_RELU = 1, DL3P_ACT_RELU6 = 2, DL3P_ACT_HSWISH = 3, DL3P_ACT_HSIGMOID = 4 };

#define DL3P_MAX_STAT_ROWS 2048

int dl3p_version(void);
const char* dl3p_last_error_string(void);
/* number of compute units / XCDs the library sizes its grids for (256 / 8 on MI355X) */
int dl3p_device_cus(void);
/* dispatch knobs that tests need to move at run time.  "pw_small_min_rows": the row count from which the
 * wave-independent streaming GEMM kernels replace the tiled kernel for small K x N (production 131072; value < 0
 * restores it).  "gemm_nt" (1..8) / "gemm_mi" (1, 2): pin the column-block width (16 * nt) / tile rows (64 * mi) of the
 * tiled GEMM, 0 = automatic; "gemm_tuned" 0: ignore the measured tile table (csrc/gemm_tuned.h) -- what
 * scripts/tune_gemm.py uses to time the candidates; "gemm_per_cu", "wgrad_tile", "wgrad_per_cu" likewise.  "dw_per_cu" /
 * "dw_want" / "dw_maxth" / "dw_tuned": the same for the plan of the depthwise window kernels (csrc/dw_tuned.h,
 * scripts/tune_dw.py).  The split-bf16 GEMMs: "split_wgrad" (0 | 1), "split_wgrad_tile" / "split_wgrad_per_cu", "sb_wm" / "sb_nt"
 * (pin the wide-tile family), "sb_pipe", "sb_rs" (0 | 1 | -1: the row-stationary form never / wherever it serves / by rule),
 * "conv_sb" (0 | 1 | 2: dense convs on the split kernels never / by the measured rule / wherever supported).  The bf16 path:
 * "bf16_kg" (0 | 1 | 2 | 4: K groups of the tiled GEMM by rule / never / pinned).  A value outside a knob's range restores its
 * default.  Unknown names return DL3P_EINVAL. */
int dl3p_set_option(const char* name, int value);
/* current value of "split_wgrad" | "conv_sb" | "sb_rs" | "sb_pipe" (the knobs that decide how many slabs / partial rows a launch
 * writes: a host that records launches pins them again before replaying them); INT_MIN for any other name */
int dl3p_get_option(const char* name);
/* What the dispatcher WOULD launch for a pointwise GEMM / a depthwise conv, without launching it (the parity tests over
 * the measured tables use it to prove that a table row is reached and what it selects).
 * dl3p_gemm_plan_query: role 0 forward, 1 forward + BatchNorm statistics (dl3p_pwconv_fwd_wt), 2 data gradient, 3 data
 * gradient + fused BatchNorm sums, 4 weight gradient; (M, K, N) = rows, reduction length, output columns of the GEMM as
 * launched (data gradient: K = the conv's output channels).  out6 = {kernel family (0 tiled MFMA, 1 wave-streaming, 2 few-row),
 * nt (role 4: tile index), tile rows / 64 (role 4: workgroups per CU), grid x (role 4: tiles), grid y (role 4: M splits), 1 if
 * the row came from csrc/gemm_tuned.h}.
 * dl3p_dw_plan_query: role 0 forward, 1 data gradient, 2 data gradient + fused BatchNorm sums, 3 weight gradient, with the
 * conv's OWN geometry (as the entry points take it).  out6 = {kind (0 gather, 1 window stride 1, 2 window stride 2, 3 residue
 * lattice, 4 quad / strided data gradient), strip width, band height, bands, workgroups per slab, 1 if from csrc/dw_tuned.h}. */
int dl3p_gemm_plan_query(int role, int M, int K, int N, int* out6);
int dl3p_dw_plan_query(int role, int N, int H, int W, int C, int k, int stride, int rate, int pad_t, int pad_l, int Ho, int Wo,
                       int* out6);

/* ---------------------------------------------------------------- data-parallel collectives (RCCL over xGMI)
 * replaces tf.distribute.MirroredStrategy's cross-replica sums (reference train.py:143-158: gradients, and
 * SyncBatchNormalization's batch statistics) for hosts that are not Python; one process per GPU.  librccl is bound at
 * run time (DL3P_RCCL_LIB overrides the search).  Rank 0 calls dl3p_comm_unique_id and hands the 128 bytes to the
 * other ranks by whatever channel the host has; every rank then calls dl3p_comm_init.  The reductions are in-place
 * sums, asynchronous on `stream`: fp32 for contiguous slices of the flat gradient buffer, fp64 for the BatchNorm
 * staging vector of (sum, sum^2) / (sum g', sum g' xhat) pairs (DESIGN.md section 6 has the order of calls of a step).
 * The Python facade uses torch.distributed (backend 'nccl' = RCCL) instead, so that the collectives are captured
 * into the step's hipGraph. */
int dl3p_comm_unique_id(void* id128);
int dl3p_comm_init(void** comm_out, int rank, int world_size, const void* id128);
int dl3p_comm_allreduce(void* comm, float* buf, size_t count, void* stream);
int dl3p_comm_syncbn_allreduce(void* comm, double* sums, size_t count, void* stream);
int dl3p_comm_destroy(void* comm);

/* ---------------------------------------------------------------- fp32-accurate GEMMs on the bf16 matrix pipe ("split bf16")
 * Twins of dl3p_pwconv_fwd_wt / dl3p_pwconv_bwd_data / dl3p_pwconv_bwd_data_bn (same reference call sites: layers.py:105,157,
 * 209-218, deeplabv3p_mobilenetv2.py:47,63) for the compute-bound 1x1 convs.  Every fp32 operand is split exactly into three
 * bf16 pieces (a = a1 + a2 + a3) and the product accumulated in fp32 from the six cross terms down to 2^-16 on
 * v_mfma_f32_16x16x32_bf16 -- the dropped terms are below one fp32 product rounding -- at 6/16 of the fp32-input MFMA's cost.
 * The conv kernel is handed over pre-split: dl3p_split_bf16x3_batch writes, for every table row {source offset (floats), rows,
 * cols, source row pitch, destination offset (bf16 elements, a multiple of 8), destination pitch}, the planes [3][rows][pitch] of
 * src[rows][cols] (pitch a multiple of 32, zero padded).  rows = the GEMM's OUTPUT columns, cols = its reduction length:
 * the transposed kernel wt[N][K] for the forward, the kernel w[K][N] as stored for the data gradient.
 * dl3p_pwconv_sb_supported(role, M, K, N) (roles and (M, K, N) as dl3p_gemm_plan_query): 0 for shapes the tiled kernel does
 * not serve (few rows; few-channel layers on the streaming kernels) -- those keep the fp32 entry points.
 * dl3p_pwconv_sb_pays(role, M, K, N): the measured verdict for that exact launch (csrc/sb_tuned.h, scripts/tune_split.py):
 * 1 the split kernel beat the fp32-input MFMA kernel, 0 it did not, -1 never measured -- then the caller's rule decides
 * (the executor's: K >= 128, N >= 128, >= 16384 rows).
 * DOMAIN of the exact split (tests/test_split_gemm_gpu.py::test_split_gemm_domain_edges): finite operands below 3.3962e38 in
 * magnitude (half a bf16 step past bf16's largest finite value; float32 reaches 3.4028e38).  An operand that is Inf or NaN, or
 * rounds to bf16 Inf, has the residual a - rn_bf16(a) = NaN: every output of that ROW is NaN, where the fp32-input kernels give
 * Inf / NaN / a finite value -- an overflowing activation still shows, never as a wrong finite number.  Below 2^-110 the third
 * (then the second) piece of an operand falls under bf16's smallest normal 2^-126: the product then carries an ABSOLUTE error of
 * at most 2^-126 |w| per term, far below any float32-normal output.
 * ARITHMETIC: the six cross products are exact (8-bit x 8-bit significands in the fp32 accumulate of v_mfma_f32_16x16x32_bf16); the
 * ACCUMULATION is not a correctly rounded sum: the matrix pipe aligns every addend to the largest one and cuts what falls below its
 * last bit off toward -infinity, without a sticky bit -- addends below 2^-24 of the running sum are truncated, not rounded.  Per
 * element the error equals the fp32-input kernels' (2e-7 rms of the output's scale); per COLUMN MEAN it is one-sided and grows with
 * the reduction length, 1.2 .. 1.3e-10 * K of the column's sigma (3.4e-8 at K = 288, 5.8e-7 at K = 4608), ten times the fp32-input
 * kernels' -- pinned with a factor-two margin by tests/test_split_gemm_gpu.py::test_split_gemm_column_mean_bias_follows_its_documented_law,
 * and its effect on 30 training steps by ::test_thirty_steps_on_the_split_gemms_stay_with_the_fp32_kernels.  -DDL3P_SB_TWO_LEVEL=1
 * (csrc/sb_common.h) removes the bias for 8-28 % of the kernels' time; DL3P_SPLIT_GEMM=0 returns every GEMM to the fp32-input kernels
 * (bench.py reports that step as `fp32_mfma_only`). */
int dl3p_split_bf16x3_batch(const float* src, void* dst, const int64_t* table, int n_matrices, void* stream);
int dl3p_pwconv_sb_supported(int role, int M, int K, int N);
int dl3p_pwconv_sb_pays(int role, int M, int K, int N);
int dl3p_pwconv_fwd_sb(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                       const void* wsp, int pitch, const float* bias, float* y, int ldy, float* stat_partials,
                       int* rows_out, int M, int K, int N, void* stream);
int dl3p_pwconv_bwd_data_sb(const float* dy, int lddy, const void* wsp, int pitch, float* gx, int ldgx, int accumulate,
                            int M, int K, int N, const float* z, int ldz, const float* scale, const float* shift,
                            int act, const float* save_mean, const float* save_invstd, float* partials,
                            int* rows_out, void* stream);
/* The data gradient above with the BatchNorm-backward APPLY of the conv's own BatchNormalization (reference layers.py:63-70 behind
 * layers.py:105,209-218) folded into its staged operand (csrc/pw_split_rs.hip, the row-stationary form): instead of dy it reads the
 * gradient g of act(BN(z_out)) and z_out, forms dz = coef0 (g act'(z_out bn_scale + bn_shift) - coef1 - xhat coef2) -- exactly
 * dl3p_bn_bwd_apply's arithmetic, evaluated as A g act' - C z + D -- once per element while the row tile is staged, multiplies with
 * it and writes it to dz (dz may be g: every element is read and written by the same lane), where the weight gradient that follows
 * reads it.  One pass over the [M][N] tensor and one launch less per BatchNorm.  dl3p_pwconv_bwd_data_sb_apply_supported: long
 * layers (M >= 131072) with a reduction N of 225 .. 256 and none / ReLU / ReLU6 behind the BatchNorm. */
int dl3p_pwconv_bwd_data_sb_apply_supported(int M, int K, int N, int bn_act, int with_sums);
int dl3p_pwconv_bwd_data_sb_apply(const float* g, int ldg, const float* z_out, int ldz_out, const float* bn_scale,
                                  const float* bn_shift, int bn_act, const float* bn_mean, const float* bn_invstd,
                                  const float* bn_coef, float* dz, int lddz, const void* wsp, int pitch, float* gx, int ldgx,
                                  int accumulate, int M, int K, int N, const float* z, int ldz, const float* scale,
                                  const float* shift, int act, const float* save_mean, const float* save_invstd,
                                  float* partials, int* rows_out, void* stream);

/* ---------------------------------------------------------------- depthwise convolution
 * replaces DepthwiseConv2D (DepthwiseConv2dNative [+SpaceToBatchND for dilation]) at
 * layers.py:100 (SepConv_BN, ASPP rates 6/12/18), deeplabv3p_mobilenetv2.py:56,
 * deeplabv3p_mobilenetv3.py:173.  w is the Keras depthwise kernel (k,k,C,1) == [k*k][C].
 * pad_t/pad_l: zeros in front (TF SAME or explicit ZeroPadding2D, layers.py:85-96); the trailing
 * pad follows from Ho/Wo.  C % 4 == 0, ld % 4 == 0. */
int dl3p_dwconv2d_fwd(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                      const float* w, float* y, int ldy, float* stat_partials, int* rows_out,
                      int N, int H, int W, int C, int k, int stride, int rate, int pad_t, int pad_l,
                      int Ho, int Wo, void* stream);
/* gx (+)= conv_transpose(dy).  dy is the gradient w.r.t. the raw conv output. */
int dl3p_dwconv2d_bwd_data(const float* dy, int lddy, const float* w, float* gx, int ldgx, int accumulate,
                           int N, int H, int W, int C, int k, int stride, int rate, int pad_t, int pad_l,
                           int Ho, int Wo, void* stream);
/* The same for a layer whose input is act(BN(z)) with a trainable BN: also returns the partial rows
 * (sum g', sum g'*xhat) of that BN's backward pass (what dl3p_bn_bwd_reduce(gx, z, ...) computes), folded into the
 * store loop where the kernel decomposition allows it.  gx must be complete after this call (last writer). */
int dl3p_dwconv2d_bwd_data_bn(const float* dy, int lddy, const float* w, float* gx, int ldgx, int accumulate,
                              int N, int H, int W, int C, int k, int stride, int rate, int pad_t, int pad_l,
                              int Ho, int Wo, const float* z, int ldz, const float* scale, const float* shift, int act,
                              const float* save_mean, const float* save_invstd, float* partials, int* rows_out,
                              void* stream);
/* gw[k*k][C] = sum over pixels of act(x*scale+shift)[tap] * dy.  workspace: rows*k*k*C floats with
 * rows <= DL3P_MAX_STAT_ROWS (query with dl3p_dwconv2d_bwd_weight_workspace). */
size_t dl3p_dwconv2d_bwd_weight_workspace(int N, int Ho, int Wo, int C, int k);
int dl3p_dwconv2d_bwd_weight(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                             const float* dy, int lddy, float* gw, float* workspace, size_t workspace_bytes,
                             int N, int H, int W, int C, int k, int stride, int rate, int pad_t, int pad_l,
                             int Ho, int Wo, void* stream);

/* ---------------------------------------------------------------- pointwise (1x1) convolution
 * replaces Conv2D(filters,(1,1)) -> Eigen/cuBLAS GEMM at layers.py:105,134,141,157,209,
 * deeplabv3p_mobilenetv2.py:47,63, deeplabv3p_xception.py:81, model.py:75.
 * y[M,N] = act(x[M,K]*scale+shift) @ w[K,N] (+ bias[N]);  w is the Keras kernel (1,1,K,N).
 * MFMA f32 (v_mfma_f32_16x16x4_f32).  K % 4 == 0, N % 4 == 0, ld % 4 == 0. */
int dl3p_pwconv_fwd(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                    const float* w, const float* bias, float* y, int ldy,
                    float* stat_partials, int* rows_out, int M, int K, int N, void* stream);
/* The same with the kernel transposed, wt[N][K] (B fragments become one 16-B LDS read): the host keeps the
 * transposed copies of all pointwise kernels next to the flat parameter buffer and refreshes them after every
 * optimiser step with dl3p_transpose_batch (table: device int[n][4] rows of (float offset, K, N, 0)). */
int dl3p_pwconv_fwd_wt(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                       const float* wt, const float* bias, float* y, int ldy, float* stat_partials, int* rows_out,
                       int M, int K, int N, void* stream);
int dl3p_transpose_batch(const float* src, float* dst, const int* table, int n_matrices, void* stream);
/* dl3p_pwconv_fwd_wt for FEW ROWS AND A LONG REDUCTION (the ASPP 1x1 convs of Xception / ResNet50 on a 33 x 33 map, layers.py:134,141,157:
 * 4356 x 2048 -> 256): the reduction is cut into slices (dl3p_pwconv_fwd_splitk_plan -> slices, 0 = not served / does not pay:
 * N in {128, 256, 512}, K >= 1024, at most 640 tiles of 64 x 128), each slice leaves a slab [M][N] in `workspace`
 * (dl3p_pwconv_fwd_splitk_workspace bytes, 16-byte aligned) and a second launch adds the slabs in slice order, adds the bias, writes y
 * and takes the BatchNorm statistic rows from the finished output.  Same arguments and results as dl3p_pwconv_fwd_wt otherwise; the
 * sum is associated differently (equal to fp32 rounding).  dl3p_set_option("splitk", 0 | S): never / S slices. */
int dl3p_pwconv_fwd_splitk_plan(int M, int K, int N);
size_t dl3p_pwconv_fwd_splitk_workspace(int M, int K, int N);
int dl3p_pwconv_fwd_wt_splitk(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                              const float* wt, const float* bias, float* y, int ldy, float* stat_partials, int* rows_out,
                              void* workspace, size_t workspace_bytes, int M, int K, int N, void* stream);
/* gx[M,K] (+)= dy[M,N] @ w[K,N]^T */
int dl3p_pwconv_bwd_data(const float* dy, int lddy, const float* w, float* gx, int ldgx, int accumulate,
                         int M, int K, int N, void* stream);
/* The same for a layer whose input is act(BN(z)) with a trainable BN: additionally emits, from the finished gradient
 * in the GEMM epilogue, the partial rows (sum g', sum g'*xhat) that dl3p_bn_bwd_reduce(gx, z, ...) would produce in
 * a separate pass over gx and z (feed them to dl3p_bn_bwd_finalize).  gx must be complete after this call (last
 * writer). */
int dl3p_pwconv_bwd_data_bn(const float* dy, int lddy, const float* w, float* gx, int ldgx, int accumulate,
                            int M, int K, int N, const float* z, int ldz, const float* scale, const float* shift,
                            int act, const float* save_mean, const float* save_invstd, float* partials, int* rows_out,
                            void* stream);
/* gw[K,N] = act(x*scale+shift)^T @ dy ; gb[N] = column sums of dy (gb may be NULL). */
size_t dl3p_pwconv_bwd_weight_workspace(int M, int K, int N);
int dl3p_pwconv_bwd_weight(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                           const float* dy, int lddy, float* gw, float* gb,
                           float* workspace, size_t workspace_bytes, int M, int K, int N, void* stream);

/* ---------------------------------------------------------------- dense stem convolution
 * replaces Conv2D(32,3,strides=2) on the RGB input (deeplabv3p_mobilenetv2.py:101,
 * deeplabv3p_xception.py:119, deeplabv3p_mobilenetv3.py:345) and the generic small dense conv
 * (deeplabv3p_xception.py:123 entry_flow_conv1_2).  w is HWIO (k,k,Cin,Cout); Cout % 4 == 0. */
int dl3p_conv2d_fwd(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                    const float* w, float* y, int ldy, float* stat_partials, int* rows_out,
                    int N, int H, int W, int Cin, int Cout, int k, int stride, int rate,
                    int pad_t, int pad_l, int Ho, int Wo, void* stream);
size_t dl3p_conv2d_bwd_weight_workspace(int N, int Ho, int Wo, int Cin, int Cout, int k);
int dl3p_conv2d_bwd_weight(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                           const float* dy, int lddy, float* gw, float* workspace, size_t workspace_bytes,
                           int N, int H, int W, int Cin, int Cout, int k, int stride, int rate,
                           int pad_t, int pad_l, int Ho, int Wo, void* stream);
int dl3p_conv2d_bwd_data(const float* dy, int lddy, const float* w, float* gx, int ldgx, int accumulate,
                         int N, int H, int W, int Cin, int Cout, int k, int stride, int rate,
                         int pad_t, int pad_l, int Ho, int Wo, void* stream);

/* im2col for the small dense convs: col[m][(ky*k+kx)*Cin+ci] = act(x*scale+shift) at the tap (zero in the
 * padding), rows padded with zeros to Kp = ld_col >= k*k*Cin (multiple of 4).  The conv is then
 * dl3p_pwconv_* on col with the HWIO kernel read as [k*k*Cin (padded to Kp)][Cout] -- the stem runs on the
 * same MFMA GEMM (and its BN-statistics epilogue) as the pointwise convs. */
int dl3p_im2col(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                float* col, int ld_col, int N, int H, int W, int Cin, int k, int stride, int rate,
                int pad_t, int pad_l, int Ho, int Wo, void* stream);

/* The RGB stem (3x3, stride 2, Cin = 3, Cout = 16 or 32, no bias, raw image input: reference
 * deeplabv3p/models/deeplabv3p_mobilenetv2.py `Conv2D(first_block_filters, kernel_size=3, strides=(2, 2), ... name='Conv')`,
 * the same layer of deeplabv3p_mobilenetv3.py, `entry_flow_conv1_1` of deeplabv3p_xception.py) as an implicit GEMM:
 * the input rows of a tile are staged in LDS and both the forward product and the weight gradient read their patch
 * operand from there -- no im2col matrix in HBM.  w / gw: [28][Cout] (the HWIO kernel flattened, row 27 padding; gw row
 * 27 is written as zero).  stat_partials / rows_out as for dl3p_pwconv_fwd.  dl3p_stem_conv_supported: 1 when a dense
 * conv of this geometry can take this path (callers fall back to dl3p_im2col + dl3p_pwconv_* otherwise). */
int dl3p_stem_conv_supported(int Cin, int Cout, int k, int stride, int rate);
int dl3p_stem_conv_fwd(const float* x, int ldx, const float* w, float* y, int ldy, float* stat_partials,
                       int* rows_out, int N, int H, int W, int Cout, int pad_t, int pad_l, int Ho, int Wo,
                       void* stream);
size_t dl3p_stem_conv_bwd_weight_workspace(int N, int Ho, int Wo, int Cout);
int dl3p_stem_conv_bwd_weight(const float* x, int ldx, const float* dy, int lddy, float* gw, float* workspace,
                              size_t workspace_bytes, int N, int H, int W, int Cout, int pad_t, int pad_l,
                              int Ho, int Wo, void* stream);

/* Dense k x k convolutions with Cin % 4 == 0 (k <= 7, stride 1 or 2, any rate) as IMPLICIT GEMMs on the MFMA kernels of
 * dl3p_pwconv_*: the patch operand is gathered while the A tile is staged into LDS, so no im2col matrix exists in HBM
 * (reference: deeplabv3p_xception.py:119-127 strided 1x1 shortcuts, :175-183 entry_flow_conv1_2; ResNet50's 3x3 convs).
 *   fwd:        wt = the HWIO kernel flattened [k*k*Cin][Cout] and TRANSPOSED to [Cout][k*k*Cin] (what dl3p_pwconv_fwd_wt
 *               takes); prologue / bias / statistics epilogue as for dl3p_pwconv_fwd.
 *   bwd_data:   wd = dl3p_conv2d_gemm_dgrad_weights(w) = [Cin][k*k*Cout]; gx (N,H,W,Cin) is written (or accumulated)
 *               completely, pixels no output tap reaches get zero.
 *   bwd_weight: gw [k*k*Cin][Cout] (+ gb [Cout] = column sums of dy, or NULL), deterministic slab reduction in
 *               `workspace`. */
int dl3p_conv2d_gemm_supported(int Cin, int Cout, int k, int stride);
int dl3p_conv2d_gemm_fwd(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                         const float* wt, const float* bias, float* y, int ldy, float* stat_partials, int* rows_out,
                         int N, int H, int W, int Cin, int Cout, int k, int stride, int rate, int pad_t, int pad_l,
                         int Ho, int Wo, void* stream);
int dl3p_conv2d_gemm_dgrad_weights(const float* w, float* wd, int k, int Cin, int Cout, void* stream);
int dl3p_conv2d_gemm_bwd_data(const float* dy, int lddy, const float* wd, float* gx, int ldgx, int accumulate,
                              int N, int H, int W, int Cin, int Cout, int k, int stride, int rate, int pad_t, int pad_l,
                              int Ho, int Wo, void* stream);
size_t dl3p_conv2d_gemm_bwd_weight_workspace(int N, int Ho, int Wo, int Cin, int Cout, int k);
int dl3p_conv2d_gemm_bwd_weight(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                                const float* dy, int lddy, float* gw, float* gb, float* workspace,
                                size_t workspace_bytes, int N, int H, int W, int Cin, int Cout, int k, int stride,
                                int rate, int pad_t, int pad_l, int Ho, int Wo, void* stream);

/* The same implicit GEMMs on the split-bf16 kernels (fp32-accurate products on the bf16 matrix pipe: see dl3p_pwconv_fwd_sb for
 * the arithmetic and its domain) -- Xception's entry_flow_conv1_2 (deeplabv3p_xception.py:175-183) and ResNet50's 3x3 convs
 * (deeplabv3p_resnet50.py:32-142).  The kernel operand is PRE-SPLIT by dl3p_split_bf16x3_batch:
 *   fwd_sb:      wsp  = [3][Cout][pitch >= k*k*Cin] from wt (the transposed kernel dl3p_conv2d_gemm_fwd takes);
 *   bwd_data_sb: wdsp = [3][Cin][pitch >= k*k*Cout] from wd (dl3p_conv2d_gemm_dgrad_weights);
 *   pitch a multiple of 32, columns beyond the matrix zero.  Everything else as for the fp32-input entry points above.
 * dl3p_conv2d_gemm_sb_supported(role, M, K, N): does the split kernel serve the GEMM as launched -- role 0 / 1 forward without /
 * with statistics (M = N*Ho*Wo, K = k*k*Cin, N = Cout), 2 data gradient (M = N*H*W, K = k*k*Cout, N = Cin), 4 weight gradient
 * (M = N*Ho*Wo, K = k*k*Cin, N = Cout); dl3p_conv2d_gemm_sb_pays: is it also the faster one (the measured rule;
 * dl3p_set_option("conv_sb", 0 | 1 | 2): never / by that rule / wherever supported).  The WEIGHT gradient has no entry point of its
 * own: dl3p_conv2d_gemm_bwd_weight[_slabs] take the split kernel (both operands split while they are staged) where role 4 pays, and
 * dl3p_conv2d_gemm_bwd_weight_workspace accounts for it. */
int dl3p_conv2d_gemm_sb_supported(int role, int M, int K, int N);
int dl3p_conv2d_gemm_sb_pays(int role, int M, int K, int N);
int dl3p_conv2d_gemm_fwd_sb(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                            const void* wsp, int pitch, const float* bias, float* y, int ldy, float* stat_partials,
                            int* rows_out, int N, int H, int W, int Cin, int Cout, int k, int stride, int rate, int pad_t,
                            int pad_l, int Ho, int Wo, void* stream);
int dl3p_conv2d_gemm_bwd_data_sb(const float* dy, int lddy, const void* wdsp, int pitch, float* gx, int ldgx, int accumulate,
                                 int N, int H, int W, int Cin, int Cout, int k, int stride, int rate, int pad_t, int pad_l,
                                 int Ho, int Wo, void* stream);

/* transpose of dl3p_im2col in gather form (deterministic): gx[n,iy,ix,ci] (+)= sum over the taps that read
 * it of gcol[m][tap*Cin+ci].  Cin % 4 == 0. */
int dl3p_col2im(const float* gcol, int ld_col, float* gx, int ldgx, int accumulate, int N, int H, int W, int Cin,
                int k, int stride, int rate, int pad_t, int pad_l, int Ho, int Wo, void* stream);

/* ---------------------------------------------------------------- batch normalisation
 * replaces BatchNormalization/FusedBatchNormV3 (layers.py:63-70 CustomBatchNormalization).
 * Training: the biased batch variance normalises AND feeds the moving average
 * moving <- moving*momentum + batch*(1-momentum).  Which variance feeds the moving average depends on the Keras class
 * CustomBatchNormalization resolves to (layers.py:63-70; the test at :64 is a STRING compare): under TF 2.2 .. 2.9
 * it is SyncBatchNormalization, Keras' non-fused path -> the biased batch variance (update_moving = 1, the default of the
 * Python facade); under the pinned tensorflow==2.11.0 ('2.11.0' >= '2.2' is False) it is plain BatchNormalization, the fused
 * FusedBatchNormV3 kernel -> Bessel-corrected, count / (count - 1) (update_moving = 2; model option
 * bn_moving_variance='unbiased').  Training-mode outputs and gradients are the same either way.
 * dl3p_bn_finalize: partial rows [rows][2][C] -> scale = gamma*invstd, shift = beta - mean*scale,
 * save_mean, save_invstd, and (update_moving: 0 no, 1 biased, 2 unbiased variance) the moving statistics.  `count` = elements per
 * channel that produced the sums (N*H*W, or the global count under SyncBN with rows == 1). */
int dl3p_bn_reduce_partials(const float* partials, int rows, int C2, double* sums, void* stream);
int dl3p_bn_finalize(const float* partials, int rows, const double* sums, int C, double count,
                     const float* gamma, const float* beta, float eps, float momentum,
                     float* moving_mean, float* moving_var, int update_moving,
                     float* scale, float* shift, float* save_mean, float* save_invstd, void* stream);
/* inference-mode coefficients from the moving statistics */
int dl3p_bn_infer_coeffs(const float* gamma, const float* beta, const float* moving_mean,
                         const float* moving_var, float eps, float* scale, float* shift,
                         float* save_mean, float* save_invstd, int C, void* stream);
/* backward, pass 1: with a = act(z*scale+shift), dyy = g * act'(.), accumulate per-channel partial
 * sums [rows][2][C] of (dyy, dyy*xhat), xhat = (z-mean)*invstd. */
int dl3p_bn_bwd_reduce(const float* g, int ldg, const float* z, int ldz, const float* scale, const float* shift,
                       int act, const float* save_mean, const float* save_invstd,
                       float* partials, int* rows_out, int M, int C, void* stream);
/* pass 1b: partial rows (or all-reduced sums) -> dgamma, dbeta and the coefficients
 * coef[3][C] = {gamma*invstd, sum(dyy)/count, sum(dyy*xhat)/count}.  frozen != 0: inference-mode
 * BN, coef = {scale, 0, 0} and no parameter gradients. */
int dl3p_bn_bwd_finalize(const float* partials, int rows, const double* sums, int C, double count,
                         const float* gamma, const float* save_invstd, const float* scale, int frozen,
                         float* dgamma, float* dbeta, float* coef, void* stream);
/* pass 2: dz (+)= coef0*(dyy - coef1 - xhat*coef2).  dz may alias g (when not accumulating).  With every
 * per-channel pointer NULL this is the backward of a bare activation: dz (+)= g * act'(z)  (the ReLU in
 * front of a SepConv_BN applied to a residual sum, layers.py:98). */
int dl3p_bn_bwd_apply(const float* g, int ldg, const float* z, int ldz, const float* scale, const float* shift,
                      int act, const float* save_mean, const float* save_invstd, const float* coef,
                      float* dz, int lddz, int accumulate, int M, int C, void* stream);

/* ---------------------------------------------------------------- elementwise
 * y = dropout(act(x*scale+shift)) [+ act2(r*rscale+rshift)]  -- materialises a lazy activation, the
 * residual Add (deeplabv3p_mobilenetv2.py:70, deeplabv3p_xception.py:86-88) and Dropout(0.5)
 * (layers.py:161).  dropout_rate == 0 disables dropout; the keep mask is a counter-based hash of
 * (seed, *step_counter, element index) so backward can regenerate it. */
int dl3p_affine_act(const float* x, int ldx, const float* scale, const float* shift, int act,
                    const float* r, int ldr, const float* rscale, const float* rshift, int ract,
                    float dropout_rate, uint64_t seed, const int64_t* step_counter,
                    float* y, int ldy, int M, int C, void* stream);
/* g_x (+)= g_y * dropout_mask/(1-rate)   (plain copy/accumulate when rate == 0) */
int dl3p_scale_mask_bwd(const float* gy, int ldgy, float dropout_rate, uint64_t seed, const int64_t* step_counter,
                        float* gx, int ldgx, int accumulate, int M, int C, void* stream);
/* the keep mask the two kernels above use, as 0/1 floats [M][C] (test hook) */
int dl3p_dropout_mask(float dropout_rate, uint64_t seed, const int64_t* step_counter, float* mask, int M, int C,
                      void* stream);
/* y = x * s[n]  with s (N,1,1,C) broadcast over HW pixels: SE block Multiply
 * (deeplabv3p_mobilenetv3.py:145); x/s may carry prologues. */
int dl3p_scale_bcast_fwd(const float* x, int ldx, const float* scale, const float* shift, int act,
                         const float* s, int lds, int s_act, float* y, int ldy, int N, int HW, int C, void* stream);
/* backward of the above: gx (+)= gy * act_s(s)[n]   (gradient w.r.t. act(x*scale+shift));
 * gs[n][c] = sum over the image's pixels of gy * act(x*scale+shift)  (gradient w.r.t. act_s(s)) */
int dl3p_scale_bcast_bwd(const float* gy, int ldgy, const float* x, int ldx, const float* scale, const float* shift,
                         int act, const float* s, int lds, int s_act, float* gx, int ldgx, int accumulate_gx,
                         float* gs, int ldgs, int N, int HW, int C, float* workspace, size_t workspace_bytes,
                         void* stream);
/* dst[i] = (float)src[i] / divide_by - subtract: normalize_image (common/data_utils.py:403-417: /127.5 - 1) and the
 * label cast (deeplabv3p/data.py:116-124: /1 - 0) for batches that arrive as bytes; bit-identical to NumPy float32 */
int dl3p_u8_to_float(const unsigned char* src, float* dst, size_t n, float divide_by, float subtract, void* stream);
/* Byte-level augmentations of SegmentationGenerator.__getitem__ (deeplabv3p/data.py:72-104) for uint8 RGB batches
 * (N,H,W,3) and uint8 label maps (N,H,W); the random draws stay on the host, as in the reference.
 * dl3p_aug_enhance_u8: PIL ImageEnhance.{Brightness (op 0), Color (1), Contrast (2), Sharpness (3)}(img).enhance(factor[n])
 * = random_brightness / random_chroma / random_contrast / random_sharpness (common/data_utils.py:83-239), bit for bit
 * (pinned against PIL, tests/golden/make_pil_enhance.py).  factor: N floats on the device.  sums: N uint64 workspace
 * (Contrast only).  out may be img except for Sharpness.
 * dl3p_aug_flip_crop_u8: random_horizontal_flip / random_vertical_flip (flags[n] bit 0 / bit 1; data_utils.py:14-60) and
 * the crop branch of random_crop (:364-400: an (h, w) window at yx[n] = (y, x) of the flipped image) in one gather;
 * img or label may be NULL, flags / yx may be NULL (no flip / whole image). */
int dl3p_aug_enhance_u8(const unsigned char* img, unsigned char* out, const float* factor, int op,
                        unsigned long long* sums, int N, int H, int W, void* stream);
int dl3p_aug_flip_crop_u8(const unsigned char* img, unsigned char* out, const unsigned char* label,
                          unsigned char* label_out, const int* flags, const int* yx, int N, int H, int W, int h, int w,
                          void* stream);
/* dl3p_aug_gridmask_u8: random_gridmask (common/data_utils.py:276-361, Grid.__call__ with mode = 1) in place: image and label
 * times 1 - rotate(grid), the rotation being PIL's Image.rotate (NEAREST, zero fill) evaluated per pixel through Pillow's
 * 16.16 fixed-point affine map (pinned against PIL, tests/golden/make_pil_gridmask.py).  params: N x 16 ints on the device,
 * {apply, hh, d, l, st_h, st_w, kind, a0..a5, top, left, 0} from the host's draws (augment.random_gridmask). */
int dl3p_aug_gridmask_u8(unsigned char* img, unsigned char* label, const int* params, int N, int H, int W, void* stream);
/* dl3p_aug_gray_blur_u8: random_grayscale then random_blur (common/data_utils.py:152-172, 105-124; deeplabv3p/data.py:95-99) for the
 * images whose flags[n] say so (bit 0 grayscale, bit 1 blur; 0 = copy).  UNPINNED restatements of OpenCV's 8-bit arithmetic (cv2 is
 * absent here): gray = (c0 * 1868 + c1 * 9617 + c2 * 4899 + 8192) >> 14 replicated to three channels; GaussianBlur (5, 5), sigma 0 =
 * [1 4 6 4 1] / 16 separable in 8.8 fixed point, one rounding, BORDER_REFLECT_101. */
int dl3p_aug_gray_blur_u8(const unsigned char* img, unsigned char* out, const int* flags, int N, int H, int W, void* stream);
/* The label tail of SegmentationGenerator.__getitem__ for byte labels (N images of P pixels each):
 * labels_out = float(label), with label > num_classes-1 replaced by ignore_index (deeplabv3p/data.py:116-121);
 * weights_out (optional) = the `--weighted_type adaptive` pixel weights (data.py:134-145): sklearn's
 * compute_class_weight('balanced') over the values present in the image, P / (distinct values * count of the
 * value) in float64, rounded to float32.  hist: N*256 unsigned workspace (zeroed by the call), NULL iff
 * weights_out is NULL. */
int dl3p_label_prepare(const unsigned char* labels, float* labels_out, float* weights_out, uint32_t* hist,
                       int N, size_t P, int num_classes, int ignore_index, void* stream);
int dl3p_fill(float* p, float value, size_t n, void* stream);
int dl3p_increment_counter(int64_t* counter, void* stream);

/* ---------------------------------------------------------------- pooling / resize
 * Per-image reductions (dl3p_global_avgpool_fwd, dl3p_scale_bcast_bwd) split each image into pixel chunks when
 * given a workspace of dl3p_pool_workspace() bytes: partial rows + tickets, summed in chunk order by the
 * workgroup that draws the last ticket (deterministic).  The workspace must be ZERO before its first use and is
 * left ready for the next call; it may be shared by calls on one stream, not by concurrent streams.  With
 * workspace == NULL (or too small) one workgroup reduces a whole (image, channel slab). */
size_t dl3p_pool_workspace(int N, int HW, int C);
/* AveragePooling2D(pool=(h,w)) == global mean (layers.py:132); y [N][C] (ldy) = out_scale * mean
 * (out_scale = HW turns it into the pixel sum that the broadcast branch's backward needs) */
int dl3p_global_avgpool_fwd(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                            float* y, int ldy, float out_scale, int N, int HW, int C, float* workspace,
                            size_t workspace_bytes, void* stream);
int dl3p_global_avgpool_bwd(const float* gy, int ldgy, float* gx, int ldgx, int accumulate,
                            int N, int HW, int C, void* stream);
/* MaxPooling2D((k,k), strides) behind ZeroPadding2D (deeplabv3p_resnet50.py:266-267 pool1_pad + MaxPooling2D(3, 2)):
 * taps outside the image are zeros that take part in the maximum.  x may carry a prologue.  Backward: gx (+)= dy routed
 * to the first maximum of each window in (ky,kx) order, w.r.t. the activated input (gather form, deterministic).
 * argmax (uint8 [N][Ho][Wo][C], optional): the forward records each window's winning tap and the backward reads it
 * instead of re-evaluating up to 4 windows x k*k taps per input pixel (257x257x64 stem map: 1.6 ms -> 0.2 ms). */
int dl3p_maxpool2d_fwd(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                       float* y, int ldy, uint8_t* argmax, int N, int H, int W, int C, int k, int stride, int pad_t,
                       int pad_l, int Ho, int Wo, void* stream);
int dl3p_maxpool2d_bwd(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                       const float* dy, int lddy, const uint8_t* argmax, float* gx, int ldgx, int accumulate,
                       int N, int H, int W, int C, int k, int stride, int pad_t, int pad_l, int Ho, int Wo, void* stream);
/* tf.image.resize(method='bilinear'), half-pixel centres, no antialias (layers.py:48-60):
 * src=(o+0.5)*in/out-0.5; lo=max(floor(src),0); hi=min(ceil(src),in-1); t=src-floor(src). */
int dl3p_resize_bilinear_fwd(const float* x, int ldx, float* y, int ldy,
                             int N, int h, int w, int C, int H, int W, void* stream);
/* deterministic gather form of the transpose: gx[h,w] (+)= sum over the output pixels that read it */
int dl3p_resize_bilinear_bwd(const float* gy, int ldgy, float* gx, int ldgx, int accumulate,
                             int N, int h, int w, int C, int H, int W, void* stream);

/* ---------------------------------------------------------------- head: pred_resize + Softmax + loss
 * model.py:76-86 (pred_resize, Reshape, Softmax 'pred_mask') fused with
 * SparseCategoricalCrossEntropy(ignore_index) (loss.py:121-156): logits z [N,h,w,C] (ldz) are
 * bilinearly upsampled to (H,W), softmax-ed; with labels != NULL the per-pixel loss
 * -log(clip(p_y,1e-7,1-1e-7))*(y != ignore) is summed into loss_partials[rows] (scaled by
 * inv_count = 1/(N*H*W) like Keras' mean over all entries) and dlogits_big [N,H,W,C] (ld C)
 * receives (p - onehot)*mask*inv_count.  logits_big/dlogits_big rows have stride ld_big (>= C; with
 * ld_big == C rounded up to 4 the rows are written as 16-B vectors and the pad channels as 0);
 * probs is compact (N,H,W,C) as Keras returns it.  probs/logits_big/dlogits_big may each be NULL. */
int dl3p_upsample_softmax_ce(const float* z, int ldz, const float* labels, int ignore_index, float inv_count,
                             float* logits_big, float* probs, float* dlogits_big, int ld_big,
                             float* loss_partials, int* rows_out,
                             int N, int h, int w, int C, int H, int W, void* stream);
/* The same head with the reference's other two losses (train.py:108-137): loss_kind DL3P_LOSS_WEIGHTED_CE =
 * WeightedSparseCategoricalCrossEntropy (loss.py:159-191: -w[y]*log(p_y), class_weights[C] on the device, no
 * clipping), DL3P_LOSS_FOCAL = SparseSoftmaxFocalLoss (loss.py:63-118: -alpha*(1-p_y)^gamma*log(p_y), p clipped to
 * [1e-15, 1-1e-15]); DL3P_LOSS_CE is dl3p_upsample_softmax_ce.  pixel_weights [N*H*W] (or NULL) are Keras sample
 * weights in 'temporal' mode (train.py:116-120, --weighted_type adaptive): each pixel's loss is multiplied by its
 * weight before the mean.  Masking, mean over all N*H*W entries and the outputs are as above. */
enum { DL3P_LOSS_CE = 0, DL3P_LOSS_WEIGHTED_CE = 1, DL3P_LOSS_FOCAL = 2 };
int dl3p_upsample_softmax_loss(const float* z, int ldz, const float* labels, int ignore_index, float inv_count,
                               int loss_kind, const float* class_weights, float focal_gamma, float focal_alpha,
                               const float* pixel_weights,
                               float* logits_big, float* probs, float* dlogits_big, int ld_big,
                               float* loss_partials, int* rows_out,
                               int N, int h, int w, int C, int H, int W, void* stream);
/* Evaluation head (eval.py:33-36 argmax of the prediction, :368-373 generate_matrix): pred_mask[N*H*W] (int32) =
 * argmax over the C classes of the upsampled logits (first index on ties, like np.argmax); for pixels with
 * 0 <= label < C, confusion[label*C + pred] += 1 (uint64 [C][C], zeroed by the caller, accumulates across calls;
 * integer atomics, order-independent).  pred_mask, or labels + confusion, may be NULL. */
int dl3p_argmax_confusion(const float* z, int ldz, const float* labels, int32_t* pred_mask,
                          unsigned long long* confusion, int N, int h, int w, int C, int H, int W, void* stream);
/* Training metrics (deeplabv3p/metrics.py:20-46, train.py:140): per image n and class c,
 * counts[n][0][c] = |label == c and pred == c|, counts[n][1][c] = |label == c|, counts[n][2][c] = |pred == c| (over all
 * pixels, ignored labels included), pred = argmax of the upsampled logits.  int32 [N][3][C], zeroed by the caller. */
int dl3p_class_counts(const float* z, int ldz, const float* labels, int32_t* counts, int N, int h, int w, int C,
                      int H, int W, void* stream);
/* Training head in one launch: the same forward + loss as dl3p_upsample_softmax_ce AND the transposed
 * pred_resize of the gradient (what dl3p_resize_bilinear_bwd(dlogits_big) returns), gz [N,h,w,C] (ldgz),
 * without ever writing the (N,H,W,C) gradient.  Available for upsampling factors up to ~4.2
 * (dl3p_head_train_supported() == 1); loss_partials[rows] as above. */
int dl3p_head_train_supported(int h, int w, int C, int H, int W);
int dl3p_head_train(const float* z, int ldz, const float* labels, int ignore_index, float inv_count,
                    float* gz, int ldgz, int accumulate, float* loss_partials, int* rows_out,
                    int N, int h, int w, int C, int H, int W, void* stream);
/* The fused training head in its SEPARABLE form (round 5, csrc/resize_head.hip): the transposed bilinear resize factorises into an x
 * and a y half.  x pass: workgroups walk full-resolution rows -- softmax / loss / gradient of a row into LDS, the transposed resize of
 * that row in x -> workspace (N,H,w,Cpad), 1/(W/w) of the full-resolution gradient; y pass: the rows of the workspace onto the logit
 * rows.  Arguments and results as dl3p_head_train plus the workspace (dl3p_head_train_rows_workspace bytes, 16-byte aligned); the
 * gradient equals dl3p_upsample_softmax_loss + dl3p_resize_bilinear_bwd to rounding (x-then-y summation), not bit for bit.
 * Plain sparse cross-entropy only (no class / pixel weights, no focal loss).  Served: Cpad in {20, 24, 32}, W / w and H / h <= 4;
 * rows wider than the LDS holds (W > ~530 at 24 channels) are cut into column segments (up to 32).  rows_out: loss partial rows
 * written (<= CUs). */
int dl3p_head_train_rows_supported(int h, int w, int C, int H, int W);
size_t dl3p_head_train_rows_workspace(int N, int h, int w, int C, int H, int W);
int dl3p_head_train_rows(const float* z, int ldz, const float* labels, int ignore_index, float inv_count, float* gz, int ldgz,
                         int accumulate, float* loss_partials, int* rows_out, void* workspace, size_t workspace_bytes,
                         int N, int h, int w, int C, int H, int W, void* stream);
/* out[n] (+)= sum over rows of partials[rows][n]   (loss, wgrad slabs) */
int dl3p_reduce_rows(const float* partials, int rows, size_t n, float* out, int accumulate, void* stream);
/* The weight gradients of a whole step reduced in two launches instead of one per layer.  dl3p_*_bwd_weight_slabs are the
 * weight-gradient entry points without their final reduction: they leave `*rows_out` slabs of n floats each at the start
 * of `workspace` (n = the gradient's element count: K*N, k*k*C, 28*Cout, k*k*Cin*Cout; no bias gradient, M > 64).
 * dl3p_reduce_rows_batched then sums every job = {const float* slabs; float* dst; int rows; int n;} (a device array of
 * 24-byte records) into its destination with exactly the arithmetic dl3p_reduce_rows applies to it:
 * dl3p_reduce_rows_variant(rows, n) says which of the two kernels that is (0 or 1), and blockmapV[b] = {job, block within
 * the job} lists the blocks of the jobs of variant V (dl3p_reduce_rows_block_elements(V) consecutive elements per block). */
int dl3p_reduce_rows_block_elements(int variant);
int dl3p_reduce_rows_variant(int rows, size_t n);
int dl3p_reduce_rows_batched(const void* jobs, const int* blockmap0, int blocks0, const int* blockmap1, int blocks1,
                             void* stream);
int dl3p_pwconv_bwd_weight_slabs(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                                 const float* dy, int lddy, float* workspace, size_t workspace_bytes, int* rows_out,
                                 int M, int K, int N, void* stream);
int dl3p_dwconv2d_bwd_weight_slabs(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                                   const float* dy, int lddy, float* workspace, size_t workspace_bytes, int* rows_out,
                                   int N, int H, int W, int C, int k, int stride, int rate, int pad_t, int pad_l,
                                   int Ho, int Wo, void* stream);
int dl3p_stem_conv_bwd_weight_slabs(const float* x, int ldx, const float* dy, int lddy, float* workspace,
                                    size_t workspace_bytes, int* rows_out, int N, int H, int W, int Cout, int pad_t,
                                    int pad_l, int Ho, int Wo, void* stream);
/* ... with the BatchNorm-backward apply of the stem's own BatchNorm folded in (g = gradient of act(BN(z)), coef from
 * dl3p_bn_bwd_finalize; the stem has no data gradient, so dz is never written): replaces dl3p_bn_bwd_apply +
 * dl3p_stem_conv_bwd_weight_slabs for Conv -> Conv_BN (deeplabv3p_mobilenetv2.py:108-125, layers.py:63-70). */
int dl3p_stem_conv_bwd_weight_slabs_bn(const float* x, int ldx, const float* g, int ldg, const float* z, int ldz,
                                       const float* bn_scale, const float* bn_shift, int bn_act, const float* save_mean,
                                       const float* save_invstd, const float* coef, float* workspace, size_t workspace_bytes,
                                       int* rows_out, int N, int H, int W, int Cout, int pad_t, int pad_l, int Ho, int Wo,
                                       void* stream);
int dl3p_conv2d_gemm_bwd_weight_slabs(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                                      const float* dy, int lddy, float* workspace, size_t workspace_bytes, int* rows_out,
                                      int N, int H, int W, int Cin, int Cout, int k, int stride, int rate,
                                      int pad_t, int pad_l, int Ho, int Wo, void* stream);
/* dl3p_pwconv_bwd_weight_slabs with the BatchNorm-backward apply of the conv's own output folded in (replaces
 * dl3p_bn_bwd_apply + dl3p_pwconv_bwd_weight_slabs; layers.py:19-26 BN behind every conv): g is the gradient of
 * act(BN(z)), coef what dl3p_bn_bwd_finalize wrote.  dz = coef0 * (g * act'(z*scale+shift) - coef1 - xhat * coef2) is formed
 * while the tiles are staged and, if dz != NULL, written once for the data gradient that follows (dz must not alias g:
 * the kernel re-reads g).  dl3p_pwconv_bwd_weight_bn_supported: 1 for the shapes whose kernel reads the gradient operand
 * once (the few-channel streaming kernels; tiled launches with a single k tile) -- elsewhere every k tile would re-form
 * dz and the separate apply pass is cheaper (measured). */
int dl3p_pwconv_bwd_weight_bn_supported(int M, int K, int N);
int dl3p_pwconv_bwd_weight_slabs_bn(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                                    const float* g, int ldg, const float* z, int ldz, const float* bn_scale,
                                    const float* bn_shift, int bn_act, const float* save_mean, const float* save_invstd,
                                    const float* coef, float* dz, int lddz, float* workspace, size_t workspace_bytes,
                                    int* rows_out, int M, int K, int N, void* stream);
/* Decoder_block (deeplabv3p/models/layers.py:207-215: x = img_resize(x) -> Concatenate([x, skip]) -> SepConv_BN) without the resized
 * tensor in HBM.  dl3p_dw_upsampled_input describes the input of the NEXT dl3p_dwconv2d_fwd / dl3p_dwconv2d_bwd_weight_slabs /
 * dl3p_dwconv2d_bwd_weight_slabs_bn call of the calling thread (one call consumes it): channels [0, up_C) of x are not read but
 * formed on the fly as the bilinear upsampling of up_x (N, up_h, up_w, up_C; row pitch up_ld) to the conv's H x W, with
 * dl3p_resize_bilinear_fwd's arithmetic expression for expression -- the output equals resize + conv bit for bit; channels from up_C
 * on are read from x as ever.  Served: 3x3, stride 1, rate 1 launches of the window kernels behind a BatchNorm + activation
 * prologue (dl3p_dw_upsampled_input_supported: role 0 forward, 1 weight gradient); anything else makes the consuming call fail. */
int dl3p_dw_upsampled_input(const float* up_x, int up_ld, int up_h, int up_w, int up_C, void* stream);
int dl3p_dw_upsampled_input_supported(int role, int N, int H, int W, int C, int up_C, int k, int stride, int rate, int pad_t, int pad_l,
                                      int Ho, int Wo);
/* the same for the depthwise 3x3 window kernels (stride 1 at any rate, stride 2): the weight gradient visits every output
 * pixel once, forms dz there and writes it (SepConv_BN: BN behind the depthwise conv, layers.py:100-105) */
int dl3p_dwconv2d_bwd_weight_bn_supported(int N, int H, int W, int C, int k, int stride, int rate, int pad_t, int pad_l,
                                          int Ho, int Wo);
int dl3p_dwconv2d_bwd_weight_slabs_bn(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                                      const float* g, int ldg, const float* z, int ldz, const float* bn_scale,
                                      const float* bn_shift, int bn_act, const float* save_mean, const float* save_invstd,
                                      const float* coef, float* dz, int lddz, float* workspace, size_t workspace_bytes,
                                      int* rows_out, int N, int H, int W, int C, int k, int stride, int rate, int pad_t,
                                      int pad_l, int Ho, int Wo, void* stream);
/* the bf16 path's weight gradients (fp32 slabs there too; same contract) */
int dl3p_pwconv_bwd_weight_slabs_bf16(const void* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                                      const void* dy, int lddy, int dy_is_f32, float* workspace, size_t workspace_bytes,
                                      int* rows_out, int M, int K, int N, void* stream);
int dl3p_dwconv2d_bwd_weight_slabs_bf16(const void* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                                        const void* dy, int lddy, float* workspace, size_t workspace_bytes, int* rows_out,
                                        int N, int H, int W, int C, int k, int stride, int rate, int pad_t, int pad_l,
                                        int Ho, int Wo, void* stream);

/* ---------------------------------------------------------------- optimiser
 * Keras SGD(momentum, nesterov=False) (common/model_utils.py:124) with the l2(2e-5) regulariser
 * gradient (layers.py:12-18) folded in:  g' = g*grad_scale + 2*l2*w; v = momentum*v - lr*g'; w += v.
 * lr is read from device memory (*lr_dev) so that a captured graph follows a schedule.
 * l2_elem / lr_scale_elem (either may be NULL) give per-element l2 factors (conv kernels are
 * regularised, depthwise kernels and BN parameters are not) and per-element learning-rate
 * multipliers (0 freezes a weight: model.py:106-110 freeze_level) over ONE flat parameter buffer. */
int dl3p_sgd_momentum(float* w, float* v, const float* g, size_t n, const float* lr_dev, float momentum,
                      float l2, float grad_scale, const float* l2_elem, const float* lr_scale_elem, void* stream);

/* Keras Adam(epsilon=1e-7) and RMSprop(rho=0.9, momentum=0, epsilon=1e-7, centered=False) (common/model_utils.py:118-121,
 * train.py --optimizer) on the same flat buffers and with the same folded regulariser / freeze mask as above:
 *   adam:    m = b1 m + (1-b1) g'; v = b2 v + (1-b2) g'^2; w -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps), t = *step_counter
 *   rmsprop: v = rho v + (1-rho) g'^2; w -= lr * g' / sqrt(v + eps) */
int dl3p_adam_step(float* w, float* m, float* v, const float* g, size_t n, const float* lr_dev,
                   const int64_t* step_counter, float beta_1, float beta_2, float epsilon, float grad_scale,
                   const float* l2_elem, const float* lr_scale_elem, void* stream);
int dl3p_rmsprop_step(float* w, float* v, const float* g, size_t n, const float* lr_dev, float rho, float epsilon,
                      float grad_scale, const float* l2_elem, const float* lr_scale_elem, void* stream);

/* ---------------------------------------------------------------- mixed precision (bf16 storage, fp32 accumulate)
 * train.py:37-46 `--mixed_precision` (the reference sets the Keras global policy; BASELINE.json configs[4] asks for bf16
 * on MobileNetV3-Large 1024x2048).  The *_bf16 entry points are the twins of the calls above for tensors stored as
 * bfloat16 (`const void*` / `void*` = bf16 device pointers, ld in ELEMENTS); BatchNorm coefficients, statistics partial
 * rows, biases, loss and every parameter gradient stay fp32.  Rounding points (round to nearest even): a conv output
 * when it is stored, and act(z*scale+shift) when a consumer's prologue forms it -- each Keras layer output is a bf16
 * tensor under the mixed policy.  Statistics partial rows are sums of the STORED (rounded) values.  GEMM weights come
 * from bf16 mirrors of the fp32 master copy: dl3p_f32_to_bf16 (w[K][N], read by the data gradient, and the depthwise
 * kernels [k*k][C]) and dl3p_transpose_batch_bf16 (wt[N][K], read by the forward).  `*_is_f32` flags let the logits
 * tensor (conv_upsample output and its gradient) stay fp32, so the softmax / loss head above is used unchanged.
 * Channel counts and row strides must be multiples of 8 for the GEMMs, of 4 elsewhere. */
int dl3p_f32_to_bf16(const float* src, void* dst, size_t n, void* stream);
int dl3p_u8_to_bf16(const unsigned char* src, void* dst, size_t n, float divide_by, float subtract, void* stream);
int dl3p_transpose_batch_bf16(const float* src, void* dst, const int* table, int n_matrices, void* stream);
int dl3p_pwconv_fwd_bf16(const void* x, int ldx, int x_is_f32, const float* in_scale, const float* in_shift, int in_act,
                         const void* wt, const float* bias, void* y, int ldy, int y_is_f32,
                         float* stat_partials, int* rows_out, int M, int K, int N, void* stream);
int dl3p_pwconv_bwd_data_bf16(const void* dy, int lddy, int dy_is_f32, const void* w, void* gx, int ldgx, int accumulate,
                              int M, int K, int N, void* stream);
/* dl3p_pwconv_bwd_data_bn for bf16 tensors (gradient, gx and z in bf16; M > 64) */
int dl3p_pwconv_bwd_data_bn_bf16(const void* dy, int lddy, const void* w, void* gx, int ldgx, int accumulate, int M, int K,
                                 int N, const void* z, int ldz, const float* scale, const float* shift, int act,
                                 const float* save_mean, const float* save_invstd, float* partials, int* rows_out,
                                 void* stream);
size_t dl3p_pwconv_bwd_weight_workspace_bf16(int M, int K, int N);
int dl3p_pwconv_bwd_weight_bf16(const void* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                                const void* dy, int lddy, int dy_is_f32, float* gw, float* gb,
                                float* workspace, size_t workspace_bytes, int M, int K, int N, void* stream);
int dl3p_dwconv2d_fwd_bf16(const void* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                           const void* w, void* y, int ldy, float* stat_partials, int* rows_out,
                           int N, int H, int W, int C, int k, int stride, int rate, int pad_t, int pad_l,
                           int Ho, int Wo, void* stream);
int dl3p_dwconv2d_bwd_data_bf16(const void* dy, int lddy, const void* w, void* gx, int ldgx, int accumulate,
                                int N, int H, int W, int C, int k, int stride, int rate, int pad_t, int pad_l,
                                int Ho, int Wo, void* stream);
size_t dl3p_dwconv2d_bwd_weight_workspace_bf16(int N, int Ho, int Wo, int C, int k);
int dl3p_dwconv2d_bwd_weight_bf16(const void* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                                  const void* dy, int lddy, float* gw, float* workspace, size_t workspace_bytes,
                                  int N, int H, int W, int C, int k, int stride, int rate, int pad_t, int pad_l,
                                  int Ho, int Wo, void* stream);
int dl3p_im2col_bf16(const void* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                     void* col, int ld_col, int N, int H, int W, int Cin, int k, int stride, int rate,
                     int pad_t, int pad_l, int Ho, int Wo, void* stream);
/* The rest of a dense k x k conv and max pooling under the mixed policy (train.py:37-46 sets it for every model type: Xception's
 * entry_flow_conv1_2, deeplabv3p_xception.py:107-113; ResNet50's 3x3 convs and pool1, deeplabv3p_resnet50.py:262-267).  bf16 twins
 * of dl3p_col2im / dl3p_maxpool2d_fwd / dl3p_maxpool2d_bwd: same gather forms, bf16 tensors, fp32 sums, one rounding at the store.
 * dl3p_maxpool2d_bwd_bf16 needs the winners the forward pass recorded (argmax, [N][Ho][Wo][C] tap indices). */
int dl3p_col2im_bf16(const void* gcol, int ld_col, void* gx, int ldgx, int accumulate, int N, int H, int W, int Cin,
                     int k, int stride, int rate, int pad_t, int pad_l, int Ho, int Wo, void* stream);
int dl3p_maxpool2d_fwd_bf16(const void* x, int ldx, const float* in_scale, const float* in_shift, int in_act, void* y,
                            int ldy, uint8_t* argmax, int N, int H, int W, int C, int k, int stride, int pad_t,
                            int pad_l, int Ho, int Wo, void* stream);
int dl3p_maxpool2d_bwd_bf16(const void* dy, int lddy, const uint8_t* argmax, void* gx, int ldgx, int accumulate, int N,
                            int H, int W, int C, int k, int stride, int pad_t, int pad_l, int Ho, int Wo, void* stream);
int dl3p_bn_bwd_reduce_bf16(const void* g, int ldg, const void* z, int ldz, const float* scale, const float* shift,
                            int act, const float* save_mean, const float* save_invstd,
                            float* partials, int* rows_out, int M, int C, void* stream);
int dl3p_bn_bwd_apply_bf16(const void* g, int ldg, const void* z, int ldz, const float* scale, const float* shift,
                           int act, const float* save_mean, const float* save_invstd, const float* coef,
                           void* dz, int lddz, int accumulate, int M, int C, void* stream);
int dl3p_affine_act_bf16(const void* x, int ldx, const float* scale, const float* shift, int act,
                         const void* r, int ldr, const float* rscale, const float* rshift, int ract,
                         float dropout_rate, uint64_t seed, const int64_t* step_counter,
                         void* y, int ldy, int M, int C, void* stream);
int dl3p_scale_mask_bwd_bf16(const void* gy, int ldgy, float dropout_rate, uint64_t seed, const int64_t* step_counter,
                             void* gx, int ldgx, int accumulate, int M, int C, void* stream);
/* per-image reductions: two launches (partial rows per pixel chunk in the fp32 workspace, then the chunk sum) */
size_t dl3p_pool_workspace_bf16(int N, int HW, int C);
int dl3p_global_avgpool_fwd_bf16(const void* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                                 void* y, int ldy, float out_scale, int N, int HW, int C, float* workspace,
                                 size_t workspace_bytes, void* stream);
int dl3p_global_avgpool_bwd_bf16(const void* gy, int ldgy, void* gx, int ldgx, int accumulate,
                                 int N, int HW, int C, void* stream);
int dl3p_scale_bcast_fwd_bf16(const void* x, int ldx, const float* scale, const float* shift, int act,
                              const void* s, int lds, int s_act, void* y, int ldy, int N, int HW, int C, void* stream);
int dl3p_scale_bcast_bwd_bf16(const void* gy, int ldgy, const void* x, int ldx, const float* scale, const float* shift,
                              int act, const void* s, int lds, int s_act, void* gx, int ldgx, int accumulate_gx,
                              void* gs, int ldgs, int N, int HW, int C, float* workspace, size_t workspace_bytes,
                              void* stream);
int dl3p_resize_bilinear_fwd_bf16(const void* x, int ldx, void* y, int ldy,
                                  int N, int h, int w, int C, int H, int W, void* stream);
int dl3p_resize_bilinear_bwd_bf16(const void* gy, int ldgy, void* gx, int ldgx, int accumulate,
                                  int N, int h, int w, int C, int H, int W, void* stream);

/* ---------------------------------------------------------------- fused inverted-residual block (csrc/irb_fwd.hip, irb_bwd.hip)
 * replaces, as ONE unit, Conv2D(expansion * in_channels, 1, use_bias=False) -> BatchNormalization -> ReLU6 ->
 * DepthwiseConv2D(3, strides=stride) of _inverted_res_block (deeplabv3p_mobilenetv2.py:43-60) where the expanded tensor is
 * large (the 257 x 257 / 129 x 129 blocks of BASELINE configs[1]): that tensor, 6x the block input, is never written to HBM --
 * forward and backward recompute the expand conv tile by tile on v_mfma_f32_16x16x4_f32 from the K-channel input.
 * x: the block input (raw + lazy prologue, as every conv input); w1: the expand kernel (1,1,K,C) = [K][C]; wdw: the depthwise
 * kernel (3,3,C,1) = [9][C]; bn_*: scale / shift / activation (+ saved mean, invstd, backward coefficient triple) of the
 * BatchNormalization between the two convs.  K in {16, 24, 32}, C a multiple of 16 (backward: C = 6K), stride 1 | 2, 3x3, rate 1.
 *
 * Training-mode statistics of the expand BatchNorm without its input tensor: z = x W1 is linear and bias-free, so
 * mean_z = W1^T mean_x and var_z[c] = w_c^T Cov(x) w_c.  dl3p_irb_cov_stats leaves rows [rows][K + K*K] of float64 (sum x, then
 * sum x x^T, exact products, float64 sums on the fp64 matrix pipe; at most dl3p_irb_cov_rows_max() rows, rows_cap = the rows
 * cov_rows has room for); dl3p_irb_cov_reduce adds the rows (the vector a
 * SyncBatchNorm all-reduces); dl3p_irb_bn_finalize_cov turns it into what dl3p_bn_finalize would have produced from the
 * materialised tensor (same outputs, same moving-average rule).
 * dl3p_irb_fwd: y = depthwise(act(BN(x W1))) raw + its stat partial rows (as dl3p_dwconv2d_fwd).
 * dl3p_irb_bwd_sums (pass A): from dy = gradient w.r.t. the raw depthwise output: the depthwise kernel's gradient as slabs
 * [slab_rows][9][C] and the BatchNorm-backward partial rows [slab_rows][2][C] (sum g', sum g' xhat) of the expand BatchNorm.
 * dl3p_irb_bwd_data (pass B), after dl3p_bn_bwd_finalize has made bn_coef from those rows: the expand kernel's gradient as slabs
 * [slab_rows][K][C]; gx (+)= gradient w.r.t. the block input (NULL: not wanted); optionally the BatchNorm-backward rows
 * [slab_rows][2][K] of a BatchNorm in FRONT of the block (z0 its raw input, as dl3p_pwconv_bwd_data_bn; sums of the FINISHED
 * gradient, i.e. including what gx held when accumulate = 1).  Slab regions are sized by dl3p_irb_bwd_workspace(which = 0 | 1). */
int dl3p_irb_supported(int N, int H, int W, int K, int C, int k, int stride, int rate, int pad_t, int pad_l, int Ho, int Wo);
int dl3p_irb_bwd_supported(int N, int H, int W, int K, int C, int k, int stride, int rate, int pad_t, int pad_l, int Ho, int Wo);
int dl3p_irb_cov_rows_max(void);
int dl3p_irb_cov_stats(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act, double* cov_rows,
                       int rows_cap, int* rows_out, int M, int K, void* stream);
int dl3p_irb_cov_reduce(const double* cov_rows, int rows, int K, double* sums, void* stream);
int dl3p_irb_bn_finalize_cov(const double* sums, const float* w1, int K, int C, double count, const float* gamma,
                             const float* beta, float eps, float momentum, float* moving_mean, float* moving_var,
                             int update_moving, float* scale, float* shift, float* save_mean, float* save_invstd, void* stream);
int dl3p_irb_fwd(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act, const float* w1,
                 const float* bn_scale, const float* bn_shift, int bn_act, const float* wdw, float* y, int ldy,
                 float* stat_partials, int* rows_out, int N, int H, int W, int K, int C, int stride, int pad_t, int pad_l,
                 int Ho, int Wo, void* stream);
size_t dl3p_irb_bwd_workspace(int which, int N, int H, int W, int K, int C, int stride, int pad_t, int pad_l);
int dl3p_irb_bwd_sums(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act, const float* w1,
                      const float* bn_scale, const float* bn_shift, int bn_act, const float* bn_mean, const float* bn_invstd,
                      const float* wdw, const float* dy, int lddy, float* gwdw_slabs, size_t slab_bytes, int* slab_rows_out,
                      float* bn_partials, int N, int H, int W, int K, int C, int stride, int pad_t, int pad_l, int Ho, int Wo,
                      void* stream);
int dl3p_irb_bwd_data(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act, const float* w1,
                      const float* bn_scale, const float* bn_shift, int bn_act, const float* bn_mean, const float* bn_invstd,
                      const float* bn_coef, const float* wdw, const float* dy, int lddy, float* gw1_slabs, size_t slab_bytes,
                      int* slab_rows_out, float* gx, int ldgx, int accumulate, const float* z0, int ldz0, const float* scale0,
                      const float* shift0, int act0, const float* mean0, const float* invstd0, float* partials0, int N, int H,
                      int W, int K, int C, int stride, int pad_t, int pad_l, int Ho, int Wo, void* stream);
/* launch-plan knobs of the three kernels (0 = default): channel tiles per wave / waves wanted of the forward; waves wanted of
 * pass A / workgroups wanted of pass B.  dl3p_irb_selftest: lane-shift primitives as the kernels use them (out[0..63] = id of
 * the lane a lane reads as "next", out[64..127] as "previous"). */
int dl3p_irb_set_plan(int ct, int want_waves);
int dl3p_irb_set_bwd_plan(int want_waves_sums, int want_workgroups_data);
int dl3p_irb_get_plan(int which);
int dl3p_irb_selftest(float* out128, void* stream);

/* ---------------------------------------------------------------- measurement hook
 * dl3p_probe_arm(i): the NEXT depthwise-forward or pointwise-GEMM kernel launch of the calling thread is issued with a pair of HIP
 * events (hipExtLaunchKernelGGL start/stop events on the launch stream) stored in slot i (0 <= i < 4096);
 * dl3p_probe_read(i, &ms) waits for slot i's stop event and returns the kernel's duration in ms.
 * Used by bench.py to time the rate-18 atrous kernel inside the timed steps (roofline). */
int dl3p_probe_arm(int slot);
int dl3p_probe_read(int slot, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* DL3P_H_ */

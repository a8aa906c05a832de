/* mo_hip.h -- C-ABI of the MI355X-native Graph-WaveNet + UNet training hot path.
 *
 * The reference (aaparcedo/multimodal_outage) is pure Python/PyTorch and has no native layer; every
 * entry point below replaces the stock-PyTorch op sequence of the cited reference lines.  Binding:
 * ctypes (multimodal_outage_amd/_lib.py); see INTEGRATION.md.
 *
 * Conventions
 *  - All pointers are DEVICE pointers to fp32 data (int32 for CSR indices) owned by the caller
 *    (torch allocates every buffer, including workspaces); the library allocates nothing.
 *  - All work is enqueued asynchronously on `stream` (a hipStream_t passed as void*); no device sync.
 *  - Return 0 on success, a negative MO_E* code otherwise; nothing is thrown across the ABI.
 *  - "nbtc" = the library's internal channels-last activation layout: a (B,C,N,T) tensor is stored as
 *    rows p = (n*B + b)*T + t of C contiguous channels (node-major so the node-axis products are
 *    plain row operations on [N][B*T*C]).
 */
#ifndef MO_HIP_H
#define MO_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MO_OK 0
#define MO_EINVAL (-1)
#define MO_ELAUNCH (-2)
#define MO_EUNSUPPORTED (-3)
#define MO_ECOMM (-4)

const char* mo_strerror(int code);
/* Bumped whenever an existing entry point changes its argument list (a stale libmo_hip.so called through a newer ctypes
 * table would silently misread its arguments): _lib.load() refuses a library whose mo_version() differs from
 * _lib.ABI_VERSION.  3 = round 3 (mo_nchw_to_nbtc/mo_nbtc_to_nchw node_new, UNet `dtypes` words, ...). */
#define MO_ABI_VERSION 6
int mo_version(void);
/* tuning switches for A/B measurements: "persist" (1: persistent skinny-K kernels; 0, default: one workgroup per tile) */
int mo_set_option(const char* name, int value);

/* ---- layout: graph_wavenet.py:189/:255 boundary, (B,C,N,T) <-> nbtc -------------------------- */
/* node_new (may be NULL): node_new[v] = internal row block of public node v -- the engine works in a node space
 * renumbered by graph clusters (blocked SpMM), and the renumbering is folded into these two boundary transposes */
int mo_nchw_to_nbtc(const float* x, float* y, int B, int C, int N, int T, const int32_t* node_new, void* stream);
int mo_nbtc_to_nchw(const float* y, float* x, int B, int C, int N, int T, const int32_t* node_new, void* stream);

/* ---- 1x1 conv / Linear (start_conv :117-119,196; skip_convs :164-166,230-236; end_conv_1/2 :174-183,
 *      252-254; residual_convs :159-161,245; Encoder/Decoder fc unet.py:132-136,156-160) -----------
 * out[p][co] = act( sum_ci f(in[map(p)][ci]) * W[co][ci] + b[co] )  (+ out[p][co] if beta)
 * rows of `in` are mapped per group: p=(g,t), t in [0,To) -> in row g*Ti + t + off (zero if outside
 * [0,Ti)); To==0 means identity.  in_relu applies ReLU to the input on load (head: relu(skip)). */
int mo_conv1x1_fwd(const float* in, int Ci, int To, int Ti, int off, int in_relu,
                   const float* W, const float* b, int Co, float* out, long P_out, int out_relu,
                   int beta, void* stream);
/* All skip convs of the stack in one pass (graph_wavenet.py:230-236: skip = s_i + skip[..., -T_i:], so only the
 * last Tf steps of every s_i reach the head):  skip[(grp,t)][co] (+)= bias[co] + sum_i sum_ci
 * g_i[(grp, t + Tout[i] - Tf)][ci] * W_i[co][ci],  t in [0,Tf).  nl <= 8 layers per call (chain more with beta);
 * bias = the sum of the layers' biases (may be null).  g_i [G*Tout[i]][32], W_i (Cs,32,1,1), skip [G*Tf][Cs]. */
int mo_skip_fwd(const float* const* g, const int* Tout, const float* const* W, int nl, const float* bias,
                int Cs, long G, int Tf, float* skip, int beta, int relu /* 1: store relu(skip): all its consumers
                apply ReLU (:252) */, void* skip_bf16 /* optional bf16 copy (throughput mode), may be NULL */,
                void* stream);
/* Data gradient of a 1x1 conv with few output channels (end_conv_2, :254): din[p][j] = (mask[p][j] > 0 ?) sum_c
 * dout[p][c]*W[c][j]; Co <= 16, Ci % 4 == 0 and 256 % (Ci/4) == 0; mask (ReLU backward, may be NULL) and the optional
 * bf16 copy have din's shape. */
int mo_conv1x1_bwd_data_smallk(const float* dout, int Co, long P, const float* W, int Ci, const float* mask,
                               float* din, void* din_bf16, void* stream);
/* din[omap(p)][ci] (+)= sum_co dout[p][co]*W[co][ci], optionally masked by (mask[p'][ci] > 0) where p'
 * indexes the rows of din (ReLU backward); output rows mapped as above (rows without image skipped). */
int mo_conv1x1_bwd_data(const float* dout, int Co, long P, const float* W, int Ci, float* din,
                        int oTo, int oTi, int ooff, const float* mask, int beta, void* stream);
/* Few-row Linear layers against a long K (the UNet Encoder/Decoder fc1/fc2, unet.py:132-136,156-160): the same
 * contractions as mo_conv1x1_fwd / _bwd_data (identity row map, no mask) with K split into slabs + a reduce pass.
 * ws: mo_linear_splitk_ws_floats(P, N, K) floats, N = output columns (Co forward, Ci backward), K the contraction. */
long mo_linear_splitk_ws_floats(long P, int N, int K);
int mo_conv1x1_fwd_splitk(const float* in, int Ci, const float* W, const float* b, int Co, float* out, long P,
                          int out_relu, float* ws, void* stream);
int mo_conv1x1_bwd_data_splitk(const float* dout, int Co, long P, const float* W, int Ci, float* din, float* ws,
                               void* stream);
/* dW[co][ci] = sum_p dout[p][co]*f(in[map(p)][ci]);  db[co] = sum_p dout[p][co] (db may be null).
 * ws: workspace of mo_wgrad_ws_floats(Co, Ci, P) floats. */
long mo_wgrad_ws_floats(int M, int N, long P);
int mo_conv1x1_bwd_weight(const float* dout, int Co, long P, const float* in, int Ci, int To, int Ti,
                          int off, int in_relu, float* dW, float* db, float* ws, void* stream);

/* ---- adaptive adjacency (graph_wavenet.py:202): adp = softmax(relu(E1 @ E2), dim=1); also adp^T - */
int mo_adp_fwd(const float* E1, const float* E2, int N, int R, float* adp, float* adpT, void* stream);
/* given dAdp (N,N) (overwritten with dz), produce dE1 (N,R), dE2 (R,N). ws: N*R*mo_adp_bwd_splits floats */
int mo_adp_bwd(const float* E1, const float* E2, const float* adp, float* dA, int N, int R,
               float* dE1, float* dE2, float* ws, long ws_floats, void* stream);

/* ---- gated TCN (graph_wavenet.py:150-156,222-226): g = tanh(conv_f(u)) * sigmoid(conv_g(u)),
 *      kernel (1,K), dilation d, u = h_prev*scale+shift (BatchNorm of the previous layer folded into
 *      the load; scale/shift may be null).  Weights are the reference tensors (32,32,1,K). --------- */
int mo_tcn_pack_weights(const float* Wf, const float* Wg, int K, float* Wp, void* stream);
int mo_tcn_fwd(const float* h_prev, const float* scale, const float* shift, const float* Wp,
               const float* bf, const float* bg, int K, int dil, long G, int Tin,
               float* g_out /* may be NULL when g_crop and g_bf16 are given */,
               void* g_bf16 /* optional bf16 copy of g_out, may be NULL */,
               int mfma_bf16 /* 1: contraction on the bf16 MFMA (operands rounded to bf16, fp32 accumulate:
               the throughput mode); 0: exact fp32 MFMA */,
               float* g_crop /* ABI 6, optional: fp32 g of the LAST crop_tf steps of every group only, compact
               [G*crop_tf][32] -- all the skip path reads (graph_wavenet.py:230-236); the node-axis products and the
               mlp read the bf16 copy, which is what they round g to anyway */, int crop_tf, void* stream);
/* backward: recomputes the pre-activations; dpre (ws, G*Tout*64 floats) ; du[G*Tin][32] = conv^T(dpre)
 * (+ dres[(g,t-(Tin-Tout))] when dres != null: the residual path of graph_wavenet.py:247);
 * dWf,dWg (32,32,1,K), dbf,dbg (32). ws2: mo_wgrad_ws_floats(64, 32*K, G*Tout) floats. */
int mo_tcn_bwd(const float* h_prev, const float* scale, const float* shift, const float* Wp,
               const float* bf, const float* bg, int K, int dil, long G, int Tin, const float* dg,
               const float* dres, float* du, float* dWf, float* dWg, float* dbf, float* dbg,
               float* dpre_ws, float* ws2, int parts /* 1: dpre + data gradient, 2: weight/bias gradients
               (needs dpre of part 1), 3: both */, int mfma_bf16 /* as in mo_tcn_fwd (data path only; the
               weight gradients stay fp32).  With mfma_bf16 = 1 dpre_ws holds the pre-activation gradients as bf16
               values (a G*Tout*64 bf16 buffer suffices) */, void* stream);

/* ---- diffusion graph convolution, node-axis products (nconv, graph_wavenet.py:64-66) -----------
 * Y[w][:] (+)= sum_e vals[e] * X[colidx[e]][:],  e in [rowptr[w], rowptr[w+1]);  rows of J floats.
 * forward uses the CSR of A^T (out[w] = sum_v A[v,w] x[v]); backward the CSR of A. */
int mo_spmm_csr(const int32_t* rowptr, const int32_t* colidx, const float* vals, int n_rows,
                const void* X, void* Y, long J, int beta, int x_bf16, int y_bf16 /* storage type of X / Y rows:
                0 fp32, 1 bf16 (fp32 accumulation either way) */, void* stream);
/* Blocked CSR for renumbered (clustered) nodes, bf16 rows: blocks of 16 consecutive output rows stage the DISTINCT
 * source rows they need (their union, <= 64) in LDS once.  rowptr/vals: CSR of the matrix; lcol[e]: position of entry
 * e's column in its block's union list; uptr[nb+1] / usrc[]: the unions (nb = ceil(n_rows/16)); max_union: the
 * largest union (MO_EARG beyond 64: use mo_spmm_csr).  Same product as mo_spmm_csr (nconv, graph_wavenet.py:64-66). */
int mo_spmm_blk(const int32_t* rowptr, const int32_t* lcol, const float* vals, const int32_t* uptr,
                const int32_t* usrc, int n_rows, int max_union, const void* X_bf16, void* Y, long J, int beta,
                int y_bf16, void* stream);
/* Y (+)= S1 X1 + S2 X2 in ONE pass over Y (both matrices blocked as for mo_spmm_blk over the same node numbering; sums in
 * the order Y, S1's entries, S2's entries: bit-identical to the two mo_spmm_blk launches it replaces -- the backward of a
 * layer accumulating the two static supports' hops into the gradient of the gated output, graph_wavenet.py:81-91). */
int mo_spmm_blk2(const int32_t* rowptr1, const int32_t* lcol1, const float* vals1, const int32_t* uptr1,
                 const int32_t* usrc1, int max_union1, const void* X1_bf16,
                 const int32_t* rowptr2, const int32_t* lcol2, const float* vals2, const int32_t* uptr2,
                 const int32_t* usrc2, int max_union2, const void* X2_bf16,
                 int n_rows, void* Y, long J, int beta, int y_bf16, void* stream);
/* dense support: Y[N][J] (+)= A_km^T @ X with A_km (N,N) row-major indexed [k][m]
 * (forward: A_km = adp; backward-data: A_km = adp^T). */
int mo_adj_gemm(const float* A_km, int N, const float* X, float* Y, long J, int beta, void* stream);
/* dA[v][w] (+)= sum_j X[v][j] * dY[w][j]   (gradient of the adaptive adjacency) */
int mo_adj_grad(const float* X, const float* dY, int N, long J, float* dA, int beta, void* stream);

/* bf16-operand / fp32-accumulate variants of the dense products (the throughput mode of BASELINE
 * config 2): D[M][N] (+)= A[M][K] * B, A bf16 k-contiguous, B bf16 either [K][N] (b_krows=1: the nbtc
 * activation matrix as stored; fetched with ds_read_b64_tr_b16) or [N][K] (b_krows=0).  lda, ldb, K
 * (and N when b_krows) must be multiples of 8.  D may be NULL when D_bf16 is given (bf16-only result; beta then
 * accumulates onto the stored bf16 values).  mo_f32_to_bf16: round-to-nearest-even, n % 8 == 0. */
int mo_gemm_bf16(const void* A, int lda, const void* B, int ldb, int b_krows, float* D, int ldd, int M,
                 int N, int K, int beta, void* D_bf16 /* optional bf16 copy of D, may be NULL */, void* stream);
/* the same 128x128 kernel with the fp32-result epilogue options of mo_gemm_bf16_256_ex (bias[n], ReLU, ReLU-backward
 * gate by mask > 0): the better kernel for output-bound shapes (short K against a large fp32 result) */
int mo_gemm_bf16_ex(const void* A, int lda, const void* B, int ldb, int b_krows, float* D, int ldd, int M, int N,
                    int K, int beta, void* D_bf16, const float* bias, int relu, const float* mask, void* stream);
int mo_f32_to_bf16(const float* x, void* y, long n, void* stream);
/* 256x256x32-tile variant with a 4-stage LDS-DMA ring (three k-tiles in flight): same contract, plus: A must be
 * readable and zero in columns [K, a_kpad) with a_kpad >= K rounded up to 32 (mo_f32_to_bf16_padded makes such
 * a copy); K % 32 == 0 when b_krows == 0. */
int mo_gemm_bf16_256(const void* A, int lda, int a_kpad, const void* B, int ldb, int b_krows, float* D,
                     int ldd, int M, int N, int K, int beta, void* D_bf16, void* stream);
int mo_f32_to_bf16_padded(const float* x, int rows, int cols, void* y, int ld_out, void* stream);
/* mo_gemm_bf16_256 with fp32-result epilogue options (the head in the throughput mode): + bias[n], ReLU, and a
 * ReLU-backward gate (result zeroed where mask[m][n] <= 0; mask has D's shape and leading dimension) */
int mo_gemm_bf16_256_ex(const void* A, int lda, int a_kpad, const void* B, int ldb, int b_krows, float* D,
                        int ldd, int M, int N, int K, int beta, void* D_bf16, const float* bias, int relu,
                        const float* mask, void* stream);

/* Skip path backward, throughput mode (graph_wavenet.py:230-236): with the data gradients of all layers' skip convs
 * computed by ONE product all[G*Tf][ld] = dskip x [Cs][32*L], add columns [col0, col0+32) of it to the last Tf time
 * steps of a layer's dg rows: dg[g*Tout + Tout-Tf+t][c] += all[g*Tf+t][col0+c]. */
int mo_skip_bwd_add(const float* all, int ld, int col0, long G, int Tf, int Tout, float* dg, void* stream);

/* Skip-path weight gradients of all layers in the throughput mode (graph_wavenet.py:164-166 backward): gather the rows of
 * every layer's gated output that reach the head (its last Tf steps) into one bf16 matrix gcat[G*Tf][32*nl]
 * (gcat[(grp,t)][32 i + c] = g_i[(grp*Tout_i + Tout_i - Tf + t)][c]); dW_all[Cs][32*nl] = dskip^T gcat is then one
 * mo_wgrad_bf16_kk launch, and mo_skip_wsplit copies layer i's 32 columns into its (Cs,32) weight gradient. nl <= 8. */
int mo_skip_gather_bf16(const float* const* g, const int* Tout, int nl, long G, int Tf, void* out_bf16, void* stream);
int mo_skip_wsplit(const float* dW_all, int Cs, int nl, float* const* dW, void* stream);
/* Weight gradient of a wide 1x1 conv (end_conv_1, graph_wavenet.py:174-177 backward) in the throughput mode:
 * dW[M][N] = A^T B over K rows with BOTH operands k-major as they lie in HBM (A: bf16 [K][M] output-gradient rows,
 * B: bf16 [K][N] input rows), split-K ring GEMM + fixed-order slab reduction.  K % 32 == 0; M, N, lda, ldb % 8 == 0. */
long mo_wgrad_bf16_kk_ws_floats(int M, int N, long K);
int mo_wgrad_bf16_kk(const void* A, int lda, const void* B, int ldb, long K, int M, int N, float* dW, float* ws,
                     void* stream);

/* ---- gcn mlp + dropout + residual + BatchNorm statistics (graph_wavenet.py:95-97,247,250) -------
 * h[p][:] = drop(W @ cat[srcs[0..ns)][p] + b) + (res[(g,t+Tin-Tout)]*rscale+rshift); per-block BN
 * partial sums go to `partial` (mo_mlp_partial_floats(P) floats). ns = 2S+1 sources of [P][32].
 * W is the reference mlp weight (32, ns*32, 1, 1). drop_thresh = p*2^32 (0: off). */
long mo_mlp_partial_floats(long P);
int mo_gcn_mlp_fwd(const float* const* srcs, int ns, const float* W, const float* b, long G, int Tout,
                   int Tin, const float* res, const float* rscale, const float* rshift,
                   uint32_t drop_seed, uint32_t drop_thresh, float drop_scale, float* h,
                   float* partial, int src_bf16_mask /* 0, or bits 1..ns-1: srcs[1..] are stored as bf16; bit 0 too (ABI 6): srcs[0], the
                   gated output, is read from its bf16 copy as well -- the kernel rounds it to bf16 either way */,
                   void* stream);
/* BatchNorm2d finalize (training: batch stats + running-stat update, momentum 0.1 semantics of
 * nn.BatchNorm2d; eval: running stats).  Outputs scale/shift (the folded affine), mean, rstd. */
int mo_bn_finalize(const float* partial, long nblk, long count, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float momentum, float eps, int training,
                   float* scale, float* shift, float* mean, float* rstd, void* stream);
/* BatchNorm backward: dh = gamma*rstd*(dy - mean(dy) - xhat*mean(dy*xhat)); dgamma, dbeta.
 * ws: mo_mlp_partial_floats(P) floats. */
int mo_bn_bwd(const float* dy, const float* h, long P, const float* gamma, const float* mean,
              const float* rstd, float* dh, float* dgamma, float* dbeta, float* ws, void* stream);
/* mlp backward: dsrcs[s][p][:] = (dh*dropmask)[p][:] @ W[:, s*32:(s+1)*32];  dW, db.
 * ws: mo_wgrad_ws_floats(32, ns*32, P) floats. */
int mo_gcn_mlp_bwd(const float* dh, const float* const* srcs, float* const* dsrcs, int ns,
                   const float* W, long P, uint32_t drop_seed, uint32_t drop_thresh, float drop_scale,
                   float* dW, float* db, float* ws, void* dlast_bf16 /* optional bf16 copy of dsrcs[ns-1] */,
                   int parts /* 1: data gradients, 2: weight/bias gradients, 3: both */,
                   int src_bf16_mask /* as in mo_gcn_mlp_fwd (bit 0 allowed since ABI 6: source 0 read from
                   its bf16 copy by the weight gradient) */, int dsrc_bf16_mask /* same form: dsrcs[1..] are
                   bf16 tensors */, void* stream);

/* ---- loss + metrics (lit.py:33-38): sums[0..3] = {sum d^2, sum |d|, sum |d|/max(|y|,1.17e-6), n};
 *      grad (optional) = 2*d/n.  ws: mo_metrics_ws_floats(n). --------------------------------------- */
long mo_metrics_ws_floats(long n);
int mo_mse_metrics(const float* yhat, const float* y, long n, float* sums, float* grad, float* ws,
                   void* stream);

/* ---- Date2Vec.encode (date2vec.py:49-53): out[i] = cat[fc1(x[i]), sin(fc2(x[i]))], 6 -> k1+k2 ---- */
int mo_date2vec_encode(const float* x, long n, const float* W1, const float* b1, int k1,
                       const float* W2, const float* b2, int k2, float* out, void* stream);

/* ---- utility ------------------------------------------------------------------------------------ */
int mo_colsum(const float* X, long P, int C, float* out, float* ws, void* stream);
long mo_colsum_ws_floats(long P, int C);
int mo_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1,
                 float beta2, float eps, float bias_c1, float bias_c2, float grad_scale, void* stream);

/* ==== Small-graph Graph WaveNet body (csrc/gwnet_small.hip) ==========================================
 * The whole layer stack of graph_wavenet.py:214-250 (gated TCN with kernel_size 1 -> diffusion hops -> mlp + dropout +
 * residual -> BatchNorm statistics) of ONE forward call of a SMALL graph in ONE workgroup; all B calls of a step in one
 * launch, BatchNorm statistics per call -- the reference calls the network once per batch element (unet.py:221), each
 * call a batch of one window, so its train-mode BatchNorm2d sees (1, N, T).  Rows are nbtc with B = number of calls, so
 * mo_nchw_to_nbtc / mo_conv1x1_* / mo_nbtc_to_nchw serve the start conv, the skip contraction and the head around it.
 *   dense_of[k] (k < nsup, supports in the reference's order, the adaptive one last): -1 = the identity matrix (folded
 *     into the mlp weights), else the index d of its dense N x N matrix adj[d] (d counts up from 0 in support order)
 *   params: L x {Wf, bf, Wg, bg, Wmlp, bmlp, gamma, beta, running_mean, running_var} device pointers
 *   gcat [rows][32 L] (the gated outputs of all layers side by side), hs [L][rows][32], xs [L][ndense][2][rows][32],
 *   stats [B][L][6][32] (scale, shift, mean, rstd, unbiased variance, unused) are outputs = saved for backward.
 *   training: batch statistics + the B sequential running-statistic updates (momentum); else running statistics.
 * Limits: mo_gwnet_small_supported (L <= 16, <= 3 supports, N <= 80, N*T <= 8192, LDS). */
int mo_gwnet_small_supported(int N, int T, int L, int nsup, int ndense);
int mo_gwnet_small_fwd(int B, int N, int T, int L, int nsup, const int* dense_of, const float* const* adj,
                       const void* const* params, const float* h0, float* gcat, float* hs, float* xs, float* stats,
                       int training, float eps, float momentum, uint32_t seed, uint32_t thresh, float dscale,
                       void* stream);
/* backward of the same stack.  dgskip [rows][32 L]: gradient reaching every g_i through the skip path; dxo [rows][32]:
 * OUT, gradient w.r.t. h0; dsts: L x {dWf, dbf, dWg, dbg, dWmlp, dbmlp, dgamma, dbeta} destinations (NULL = not wanted),
 * written as the fixed-order sum over the calls; adaptive_dense: dense index whose gradient dA [N][N] is wanted, or -1;
 * ws: mo_gwnet_small_bwd_ws_floats. */
long mo_gwnet_small_bwd_ws_floats(int B, int N, int T, int L, int nsup, int ndense);
long mo_gwnet_small_slab_floats(int nsup);
int mo_gwnet_small_bwd(int B, int N, int T, int L, int nsup, const int* dense_of, const float* const* adj,
                       int adaptive_dense, const void* const* params, void* const* dsts, const float* h0,
                       const float* gcat, const float* hs, const float* xs, const float* stats, float eps,
                       uint32_t seed, uint32_t thresh, float dscale, const float* dgskip, float* dxo, float* ws,
                       float* dA, void* stream);

/* ==== UNet encoder/decoder conv stacks (unet.py:40-92, batched over all B*67*H tiles) ===============
 * NCHW images; `istride`/`ostride` are image strides in floats (so channel slices of wider buffers can
 * be addressed).  An "activated view" is a raw conv output y with the folded per-(group,channel)
 * BatchNorm affine sc/sh [G][C] (+ReLU) applied on load; sc==NULL means a plain tensor.  Groups are
 * `gsize` consecutive images: the reference calls each block once per county on `horizon` images, so
 * train-mode BatchNorm statistics are per (batch element, county). */

/* Activation storage ("bf16 mode" of BASELINE config 3): a `dtypes` argument says which of an entry point's activation
 * tensors are stored as bf16 instead of fp32 (the pointer types stay float*; strides count ELEMENTS).  Arithmetic is
 * fp32 unless MO_BF_MATH (below) is set as well.  bf16 storage exists on the direct / streaming kernels (3x3 convs with <= 32 output channels at
 * >= 32x32 pixels, the MFMA weight gradient at W % 64 == 0 and H % 8 == 0, thin 1x1 convs, activation kernels);
 * any other shape returns MO_EUNSUPPORTED when a flag is set. */
#define MO_BF_IN0 1   /* first input view (mo_unet_act_bwd: y) */
#define MO_BF_IN1 2   /* second input view (mo_unet_act_bwd: da) */
#define MO_BF_OUT 4   /* the output tensor (mo_unet_act_bwd: dy) */
#define MO_BF_DY 8    /* the output-gradient operand of a weight / data gradient */
#define MO_BF_DP 16   /* mo_unet_act_bwd: the pooled gradient dp */
/* Arithmetic of the bf16 mode: with MO_BF_MATH the 3x3 convs, their data and weight gradients run on the bf16 matrix pipe
 * (operands rounded to bf16 on the way into the MFMA, fp32 accumulation) where csrc/unet_bf16.hpp has a kernel for the
 * shape (mo_conv3x3_bf16_route); elsewhere the flag is ignored and the arithmetic is fp32.  MO_W_FLIP (with MO_BF_MATH,
 * mo_conv3x3_fwd only): W is the forward conv's (Ci, Co, 3, 3) tensor and is read transposed + flipped -- the data
 * gradient without a flipped copy. */
#define MO_BF_MATH 32
#define MO_W_FLIP 64

/* DoubleConv conv (unet.py:44,47; nn.Conv2d k=3 pad=1 bias=False) over the channel concat of up to two
 * activated views (the skip/up cat of unet.py:83): out[img][co] raw. W: (Co, C0+C1, 3, 3). */
int mo_conv3x3_fwd(const float* in0, int C0, long istride0, const float* sc0, const float* sh0, int relu0,
                   const float* in1, int C1, long istride1, const float* sc1, const float* sh1, int relu1,
                   int gsize, const float* W, int Co, long n_img, int H, int Wd, float* out, long ostride,
                   float* stats /* optional [n_img][tiles][Co][2] per-tile (sum, sumsq) of out, tiles =
                   mo_conv3x3_stats_tiles(): the BatchNorm statistics come out of the conv's epilogue; NULL: none */,
                   int dtypes /* MO_BF_IN0 | MO_BF_IN1 | MO_BF_OUT */,
                   const long long* in0_off /* NULL, or per-image ELEMENT offsets of the first view: image img lies at
                   in0 + in0_off[img] instead of in0 + img * istride0 -- the network input as the permuted batch view of
                   lit.py:31, read in place (direct / matrix-pipe kernels only, else MO_EUNSUPPORTED) */, void* stream);
/* per-image statistics rows mo_conv3x3_fwd writes for this shape (0: none -- run mo_nchw_stats on the output) */
int mo_conv3x3_stats_tiles(int Co, long n_img, int H, int Wd);
/* ... when the call carries `dtypes` and the two views (MO_BF_MATH routes to the bf16 matrix-pipe kernel, whose tiles are 16 x 64) */
int mo_conv3x3_stats_tiles2(int C0, int C1, int Co, long n_img, int H, int Wd, int dtypes);
/* 1 when the bf16 matrix-pipe kernels (csrc/unet_bf16.hpp) serve a 3x3 conv of this shape under MO_BF_MATH */
int mo_conv3x3_bf16_route(int Ci, int Co, long n_img, int H, int Wd);
/* Wf[ci][co][ky][kx] = W[co][ci][2-ky][2-kx]; the data gradient is mo_conv3x3_fwd(dy, Wf). */
int mo_conv3x3_flip_weights(const float* W, int Co, int Ci, float* Wf, void* stream);
long mo_unet_wgrad_ws_floats(int M, int N, long P);
/* dW[co][ci][tap] = sum_{img,pix} dy * act(in) shifted; ws: mo_unet_wgrad_ws_floats(Co,(C0+C1)*9,n_img*H*W) */
int mo_conv3x3_bwd_weight(const float* dy, long dystride, int Co, const float* in0, int C0, long istride0,
                          const float* sc0, const float* sh0, int relu0, const float* in1, int C1,
                          long istride1, const float* sc1, const float* sh1, int relu1, int gsize,
                          long n_img, int H, int Wd, float* dW, float* ws,
                          int dtypes /* MO_BF_DY | MO_BF_IN0 | MO_BF_IN1 */,
                          const long long* in0_off /* as mo_conv3x3_fwd */, void* stream);
/* OutConv (unet.py:86-92): 1x1 conv with bias on an activated NCHW view */
int mo_nchw_conv1x1_fwd(const float* in, long istride, int Ci, const float* sc, const float* sh, int relu,
                        int gsize, const float* W, const float* b, int Co, long n_img, int HW, float* out,
                        long ostride, int dtypes /* MO_BF_IN0 | MO_BF_OUT */, void* stream);
int mo_nchw_conv1x1_bwd_data(const float* dout, long dostride, int Co, const float* W, int Ci, long n_img,
                             int HW, float* din, long distride, int dtypes /* MO_BF_DY | MO_BF_OUT */, void* stream);
/* db (may be NULL): the bias gradient, sum of dout over images and pixels */
int mo_nchw_conv1x1_bwd_weight(const float* dout, long dostride, int Co, const float* in, long istride,
                               int Ci, const float* sc, const float* sh, int relu, int gsize, long n_img,
                               int HW, float* dW, float* db, float* ws, int dtypes /* MO_BF_DY | MO_BF_IN0 */,
                               void* stream);
/* OutConv + loss + OutConv backward in one pass -- the tail of training_step (lit.py:32-38 on unet.py:86-92: yhat =
 * OutConv(act(in)); loss = MSE(yhat, y); MAE / MAPE / RMSE).  yhat is consumed by the loss only, so it is formed in registers
 * (written to `yhat` only when that is not NULL); the pass also leaves da = d loss / d act(in) (Ci planes per image,
 * MO_BF_OUT: bf16) for an upstream gradient d loss = 1 (the consumer scales: mo_unet_act_bwd's out_scale).  target: Co planes of HW floats per image at
 * target + (target_off ? target_off[img] : img * Co * HW)  (lit.py:31 hands a permuted view of the batch).
 * out4 = {mse, mae, mape, rmse} (MAPE eps 1.17e-6, as torchmetrics).  Ci <= 4, Co <= 16. */
long mo_outc_loss_ws_floats(long n_img, int HW, int Ci, int Co);
int mo_outc_loss_fwd(const float* in, long istride, int Ci, const float* sc, const float* sh, int relu, int gsize,
                     const float* W, const float* b, int Co, const float* target, const long long* target_off,
                     long n_img, int HW, float* yhat, float* da, long dastride, float* ws, float* out4,
                     int dtypes /* MO_BF_IN0 | MO_BF_OUT (da) */, void* stream);
/* the weight / bias gradient of that OutConv, beside the data-flow chain: a second pass over the same input view and target
 * re-forms d and sums dW (Co,Ci), db (Co), multiplied by *scale (device scalar: the upstream gradient of the loss; NULL = 1).
 * ws: a workspace of mo_outc_loss_ws_floats floats (may be the forward's). */
int mo_outc_loss_bwd(const float* in, long istride, int Ci, const float* sc, const float* sh, int relu, int gsize,
                     const float* W, const float* b, int Co, const float* target, const long long* target_off,
                     long n_img, int HW, float* ws, const float* scale, float* dW, float* db,
                     int dtypes /* MO_BF_IN0 */, void* stream);
/* Up.up (unet.py:71): ConvTranspose2d(Ci, Co, k=2, s=2) with bias; W (Ci, Co, 2, 2); H,Wd = input size */
/* dtypes (ABI 5): MO_BF_IN0 the input view, MO_BF_OUT the upsampled result, MO_BF_DY the gradient w.r.t. it stored as bf16
 * -- served by the streaming kernels only (mo_convt2x2_bf16_route(Ci, Co, n_img) == 1: Ci <= 16, Co <= 8), else
 * MO_EUNSUPPORTED.  The bf16 mode stores the upsampled map of the three large levels and the concat gradient behind it
 * as bf16: the concat conv rounds its operands to bf16 anyway, so its forward result is bit-identical. */
int mo_convt2x2_bf16_route(int Ci, int Co, long n_img);
int mo_convt2x2_fwd(const float* in, long istride, int Ci, const float* sc, const float* sh, int relu,
                    int gsize, const float* W, const float* b, int Co, long n_img, int H, int Wd, float* out,
                    long ostride, int dtypes, void* stream);
int mo_convt2x2_bwd_data(const float* dout, long dostride, int Co, const float* W, int Ci, long n_img, int H,
                         int Wd, float* din, long distride, int dtypes, void* stream);
int mo_convt2x2_bwd_weight(const float* dout, long dostride, int Co, const float* in, long istride, int Ci,
                           const float* sc, const float* sh, int relu, int gsize, long n_img, int H, int Wd,
                           float* dW, float* db /* bias gradient, may be NULL */, float* ws, int dtypes, void* stream);
/* BatchNorm2d statistics (unet.py:45,48): stats[img][c] = (sum, sumsq) over HW */
int mo_nchw_stats(const float* y, long istride, int C, long n_img, int HW, float* stats, void* stream);
/* per-group finalize: scale/shift/mean/rstd [G][C]; running stats receive G sequential momentum updates
 * in group order (= the reference's county-then-batch order of nn.BatchNorm2d calls) */
int mo_group_bn_finalize(const float* stats /* [n_img][ntile][C][2] */, long n_img, int C, int gsize, int HW,
                         int ntile /* statistics rows per image: 1 for mo_nchw_stats */, const float* gamma,
                         const float* beta, float* running_mean, float* running_var, float momentum,
                         float eps, int training, float* scale, float* shift, float* mean, float* rstd,
                         void* stream);
/* the same in one launch (train mode: one workgroup does both stages); num_batches_tracked (int64, may be NULL) += G */
int mo_group_bn_finalize2(const float* stats, long n_img, int C, int gsize, int HW, int ntile, const float* gamma,
                          const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                          int training, float* scale, float* shift, float* mean, float* rstd,
                          long long* num_batches_tracked, void* stream);
/* materialise relu(y*sc+sh), optionally 2x2 max-pooled (Down, unet.py:60) */
int mo_unet_act(const float* y, long istride, int C, long n_img, int H, int Wd, const float* sc,
                const float* sh, int gsize, int pool, float* out, long ostride,
                int dtypes /* MO_BF_IN0 | MO_BF_OUT */, void* stream);
/* backward through ReLU + group BatchNorm (+ max-pool routing of dp); da = gradient w.r.t. the activated
 * view, dp = gradient w.r.t. its pooled version (either may be NULL); ws: mo_unet_act_bwd_ws_floats */
long mo_unet_act_bwd_ws_floats(long n_img, int C);
int mo_unet_act_bwd(const float* y, long istride, int C, long n_img, int H, int Wd, int gsize,
                    const float* gamma, const float* mean, const float* rstd, const float* sc,
                    const float* sh, const float* da, long dastride, const float* dp, long dpstride,
                    float* dy, long dystride, float* dgamma, float* dbeta, float* ws,
                    int dtypes /* MO_BF_IN0 (y) | MO_BF_IN1 (da) | MO_BF_DP | MO_BF_OUT (dy) */,
                    const float* out_scale /* device scalar or NULL: dy, dgamma, dbeta are multiplied by it -- the upstream
                    gradient of the loss when da comes from mo_outc_loss_fwd, which forms it for d loss = 1 */, void* stream);
/* out[c] = sum over images and pixels (bias gradients); ws: n_img*C*2 floats */
int mo_nchw_channel_sum(const float* x, long istride, int C, long n_img, int HW, float* out, float* ws,
                        void* stream);
/* nn.MaxPool2d(2) backward on a plain NCHW tensor x (Down.forward called on its own, unet.py:55-65): dx gets dp at the
 * first maximum of every 2x2 window and 0 elsewhere.  Forward = mo_unet_act with sc = sh = NULL, pool = 1. */
int mo_maxpool2_bwd(const float* x, long istride, int C, long n_img, int H, int Wd, const float* dp, long dpstride,
                    float* dx, long dxstride, void* stream);
/* nn.Dropout (unet.py:135,159) with the counter-based mask; the same call is its own backward */
int mo_dropout(const float* x, float* y, long n, uint32_t seed, uint32_t thresh, float scale, void* stream);
/* ReLU backward on a materialised activation y: out = (y > 0) ? dy : 0 (fc layers, unet.py:142-144) */
int mo_relu_bwd(const float* dy, const float* y, float* out, long n, void* stream);

/* Few-row Linear layers against a large weight matrix in the bf16 mode (Encoder.fc1 / Decoder.fc2 of unet.py:138-173 at
 * 256-pixel tiles): "3 x bf16" split products on the bf16 matrix pipe, ~1.5e-5 relative per product (csrc/unet_fc.hpp).
 * P <= 144 rows; reduction length % 8 == 0; ws: mo_fc3_ws_floats(P, reduction length, output columns).
 *   mo_fc3_fwd:      out[P][N] = relu?(x[P][K] W[N][K]^T + b)      (b may be NULL)
 *   mo_fc3_bwd_data: din[P][K] = dout[P][N] W[N][K] */
int mo_fc3_supported(long P, int R, int C);
long mo_fc3_ws_floats(long P, int R, int C);
int mo_fc3_fwd(const float* x, long P, int K, const float* W, const float* b, int N, int relu, float* out, float* ws,
               void* stream);
int mo_fc3_bwd_data(const float* dout, long P, int N, const float* W, int K, float* din, float* ws, void* stream);
/*   mo_fc3_bwd_weight: dW[N][K] = dout[P][N]^T x[P][K], db[N] = column sums of dout (may be NULL); P <= 160;
 *   ws: mo_fc3_wgrad_ws_floats(P, N, K) */
long mo_fc3_wgrad_ws_floats(long P, int N, int C);
int mo_fc3_bwd_weight(const float* dout, long P, int N, const float* x, int C, float* dW, float* db, float* ws,
                      void* stream);

/* ---- input rasters (the step in front of the path; BlackMarbleDataset's per-image transform, utils.py:35-38,59-64):
 * raw (n, h, w) radiance -> out (n, oh, ow): fill_value -> 0, bilinear antialiased resize (what torchvision 0.18's
 * transforms.Resize does to a float tensor: F.interpolate(mode='bilinear', align_corners=False, antialias=True)), then
 * (x - mean) / std. */
int mo_raster_prepare(const float* raw, long n, int h, int w, float fill_value, float mean, float std, float* out,
                      int oh, int ow, void* stream);

/* A/B switches of the UNet kernels for measurements (defaults are the product path):
 *   "no_mfma_wgrad" 1: the VALU / split-K 3x3 weight gradients;  "no_mfma_conv" 1: deep-level convs on the tile engine;
 *   "no_bf16_mfma" 1: MO_BF_MATH requests on the fp32 kernels;   "ub_min_w" 32: the 32 x 32 level on the bf16 conv too;
 *   "ux_min_co": smallest output-channel count on the fp32 matrix-pipe conv at >= 32 x 32 pixels (16);
 *   "ux_split" n: workgroups per tile of that conv, each with its share of the 16-channel output blocks (0: heuristic);
 *   "ub_no_pack" 1: thin outputs (Co <= 8) on the unpacked D[pixel][co] kernel;  "ub_ipw" n: images per workgroup;
 *   "fc_wide" 0: one 16-column block of W per wave in the 3 x bf16 FC kernels;  "fc_groups_grid" 1: FC row groups in
 *   the grid instead of inside the workgroup.
 * The dense ring GEMM reads MO_GEMM_MFMA=32 from the environment once (v_mfma_f32_32x32x16_bf16 instead of 16x16x32). */
int mo_unet_set_option(const char* name, int value);

/* ---- data-parallel exchange step: gradient all-reduce over RCCL / xGMI ---------------------------------
 * Replaces Lightning's implicit DDP(NCCL) gradient all-reduce (lit.py:204; the reference has no explicit distributed
 * code).  One communicator per process (= per GPU).  Rank 0 calls mo_allreduce_unique_id and hands the 128 bytes to
 * every rank by any host channel; every rank then calls mo_allreduce_init.  mo_allreduce_launch sums buf[0..n) over
 * all ranks IN PLACE on the communicator's own HIP stream, ordered by an event behind everything already queued on
 * producer_stream, and returns at once (the collective runs beside the rest of backward); mode 0 = one all-reduce,
 * mode 1 = reduce-scatter + all-gather (n % world == 0).  mo_allreduce_wait orders consumer_stream behind all
 * collectives launched so far (no host wait).  The handle is the library's only state. */
int mo_allreduce_unique_id(void* id128);
int mo_allreduce_init(const void* id128, int rank, int world, void** handle);
int mo_allreduce_launch(void* handle, float* buf, long n, int mode, void* producer_stream);
int mo_allreduce_wait(void* handle, void* consumer_stream);
int mo_allreduce_destroy(void* handle);

/* ---- static supports: CSR of a dense HOST matrix (load_adj / asym_adj outputs, graph_wavenet.py:13-32,
 * utils.py:152-158), rows ascending, columns ascending within a row (bit-exact vs scipy.sparse.csr_matrix).
 * First call with colidx = vals = NULL fills rowptr[n_rows+1] and *nnz; second call fills colidx/vals. */
int mo_csr_from_dense(const float* dense_host, int n_rows, int n_cols, int32_t* rowptr, int32_t* colidx,
                      float* vals, long* nnz);

#ifdef __cplusplus
}
#endif
#endif

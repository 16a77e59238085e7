/*
 * fgnn_hip.h -- C ABI of libfgnn_hip.so, the MI355X (gfx950) implementation of the
 * 2-FGNN hot path of mlelarge/graph_neural_net.
 *
 * The reference has no FFI: its boundary is the Python nn.Module surface
 * (models/layers.py, models/blocks_emb.py, maskedtensors/maskedtensor.py).  Each entry
 * point below names the reference code (file:line under /root/reference) whose
 * arithmetic it replaces; graph_neural_net_amd/ (Python) binds them with ctypes and
 * re-exposes the reference's module names on top.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HBM), fp32 unless stated, owned by the caller;
 *   - activations are channel-first: element (g, c, i, j) of a (G, C, N, N) tensor lives at
 *     ptr[g*gstride + c*ldp + i*N + j]; ldp >= N*N is the channel stride (the Python host
 *     uses ldp = N*N for user-visible tensors and a 32-float-aligned ldp internally);
 *   - `nvalid` (optional, int32[G]) is the per-graph vertex count of a ragged batch
 *     (MaskedTensor masks, maskedtensor.py:8-48); NULL means every graph has N vertices.
 *     Entries with i >= nvalid[g] or j >= nvalid[g] are padding: read as 0, written as 0,
 *     excluded from every statistic (maskedtensor.py:87-112, 310-335);
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it;
 *   - return value 0 = success; non-zero = error, message via fgnn_last_error().
 *     Unsupported shapes are errors, never a silent fallback.
 *   - hidden/output width of the MLP kernels is FGNN_H = 32 channels (the reference's
 *     in_features = out_features = 32, default_config.yaml:47-57).
 */
#ifndef FGNN_HIP_H
#define FGNN_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define FGNN_H 32            /* hidden / output channels of every MLP kernel          */
#define FGNN_TILE 32         /* pixels per statistics tile                             */
#define FGNN_MAX_DEPTH 3     /* convs per MlpBlock_Real handled by the fused kernels  */

/* One input slab of an MLP: C channels, optionally the *pre-norm* output `z` of an
 * earlier MLP kernel together with that MLP's GraphNorm record, in which case the
 * consumer applies y = (z - mean) * a + beta on load (models/layers.py:68-80). */
typedef struct {
    const float *ptr;        /* (G, C, ldp)                                            */
    long long gstride;       /* floats between graphs                                  */
    long long ldp;           /* floats between channels                                */
    int C;                   /* channels (even; 0 = slab unused)                       */
    const float *nrm;        /* optional (G*C*4): {mean, a, q, r2} per (g,c)           */
    const float *beta;       /* optional (C): GraphNorm bias                           */
} fgnn_slab;

/* ---- library ---------------------------------------------------------------------- */
const char *fgnn_last_error(void);
int fgnn_version(void);
/* number of statistics tiles per graph for N vertices: ceil(N*N / FGNN_TILE) */
int fgnn_tiles_per_graph(int N);
/* workgroups the persistent MLP kernels launch (partials buffers are sized by it) */
int fgnn_mlp_bwd_num_workgroups(void);

/* ---- LDS operand images ------------------------------------------------------------------
 * The MLP kernels keep their MFMA A-operands (weights, biases, transposed weights) in LDS.
 * Weights are constant within a step, so the images of all MLP launches can be packed ONCE
 * per step by one small launch and are then copied straight into LDS by every workgroup.
 * kind 0 = forward image (nmlp MLPs), kind 1 = backward image (one MLP; uses W[0], bias[0]);
 * kind 4 / 5 = the same two images in the layout of the 16-pixel-tile kernels (fgnn_*_t16; same sizes). */
#define FGNN_MAX_PACK_JOBS 24
typedef struct {
    int kind, ca, cb, depth, nmlp;
    const float *W[2][FGNN_MAX_DEPTH];
    const float *bias[2][FGNN_MAX_DEPTH];
    float *out;                              /* fgnn_pack_floats(kind, ca, cb, depth, nmlp) floats */
} fgnn_pack_job;
int fgnn_pack_floats(int kind, int ca, int cb, int depth, int nmlp);
int fgnn_pack_operands(const fgnn_pack_job *jobs, int njobs, void *stream);

/* ---- MlpBlock_Real.forward minus the final normalisation ---------------------------
 * replaces models/layers.py:126-131 (conv1x1+ReLU chain, last conv without ReLU) and the
 * reductions of normalize (:72-73).  Computes, for nmlp (1 or 2) MLPs sharing one input
 *   z_m = W_m[d-1] relu(... relu(W_m[0] x + b_m[0]) ...) + b_m[d-1]      (G, 32, ldz)
 * with x = [slab a ; slab b] on channels (Concat, layers.py:145-146, never materialised),
 * and per (graph, tile, channel) partial statistics {mean, M2} of z over valid pixels.  */
typedef struct {
    int G, N, depth, nmlp;
    const int *nvalid;
    fgnn_slab a, b;
    const float *W[2][FGNN_MAX_DEPTH];      /* conv weights (32, Cin_l) row-major       */
    const float *bias[2][FGNN_MAX_DEPTH];   /* conv biases (32)                          */
    float *z[2];                            /* out (G, 32, ldz)                          */
    long long ldz;
    float *part[2];                         /* out (G, tpg, 32, 2) {mean, M2}            */
    float *cnt;                             /* out (G, tpg) valid pixels per tile        */
    const float *packed;                    /* optional: operand image from fgnn_pack_operands
                                               (kind 0) for exactly these weights; NULL = the kernel
                                               builds it itself (slower prologue)              */
    const unsigned *xbits;                  /* optional: the 2-channel slab (a when a.C == 2, else b when b.C == 2) is not read
                                               from its ptr but expanded on the fly from the bit-packed adjacency
                                               (G, N, ceil(N/32)) words, bit j of row i = W[i][j]: channel 0 = W, channel 1 =
                                               diag(xdeg) (loaders/data_generator.py:118-125); the slab need not exist */
    const float *xdeg;                      /* (G, N) row sums over the valid columns, from fgnn_adjacency_degree */
    const int *ranges;                      /* optional, ragged batches (nvalid != NULL): FGNN_RANGE_WG + 1 tile bounds from
                                               fgnn_ragged_tile_ranges.  Workgroup w then owns tiles [ranges[w], ranges[w+1])
                                               (equal WORK instead of equal tile counts) and tiles that lie entirely in the
                                               padding are stepped over: their statistics records are written as empty, their
                                               z pixels are left untouched (consumers must do the same, or read the valid
                                               n x n corner only -- all kernels of this library do) */
    int cu_share;                           /* 0 / 1: the whole GPU (up to 256 persistent workgroups).  2: HALF of the CUs (at most 128
                                               workgroups): two launches on different streams -- independent half-batches, see
                                               engine_dual.py -- run side by side on disjoint CUs, one chain's ramps under the
                                               other's streaming.  Results do not depend on it.  IGNORED when `ranges` is given:
                                               the range table has one entry per workgroup of the full grid, so a ragged launch
                                               always runs FGNN_RANGE_WG workgroups */
} fgnn_mlp_fwd_args;
int fgnn_mlp_fwd(const fgnn_mlp_fwd_args *args, void *stream);
/* Round 6: the same launch on 16-pixel tiles / v_mfma_f32_16x16x4_f32 (csrc/mlp_fwd_t16.hip, csrc/fgnn_t16.h).  z is BIT-IDENTICAL to
 * fgnn_mlp_fwd's (the conv chain runs through the same sequence of fused multiply-adds).  Differences: `packed` is mandatory and of
 * kind 4; the unit of work and of the tile statistics is a 16-pixel half tile -- part[m] is (G, R, 32, 2) and cnt (G, R) with R =
 * fgnn_mlp_fwd_t16_records(N) = 2 * fgnn_tiles_per_graph(N) -- so the statistics go to the *_r consumers below (fgnn_gn_finalize_r,
 * fgnn_gn_finalize2_r, fgnn_chan_matmul_fwd_fin_ord_r, fgnn_colmax_fwd_fin_r), which take the record count as an argument.
 * Built (fgnn_mlp_fwd_t16_supported) for depth 3, N <= 256, inputs of 32 or 2 channels (nmlp = 2) and 32 + 32 / 32 + 2 (nmlp = 1). */
int fgnn_mlp_fwd_t16_records(int N);
int fgnn_mlp_fwd_t16_supported(const fgnn_mlp_fwd_args *args);
int fgnn_mlp_fwd_t16(const fgnn_mlp_fwd_args *args, void *stream);

/* ---- the same two MLP entry points on the bf16 matrix cores ("x3": csrc/fgnn_x3.h) ------------------------------
 * replaces the same reference lines as fgnn_mlp_fwd / fgnn_mlp_bwd (models/layers.py:126-131 and its autograd).  Tensors,
 * argument structs, tile statistics and results are those of the fp32-MFMA kernels (same parity gates); inside, every fp32
 * operand of a channel contraction is split exactly into three bf16 numbers and the product is accumulated in fp32 from
 * the six partial products >= 2^-16 (error < 3 * 2^-24 |a b| per product, the size of an fp32 rounding).  Built for
 * depth 3, slabs of 2 / 32 / 32+2 / 32+32 channels and constant-size batches (no `ranges`); `packed` is mandatory and
 * comes from fgnn_pack_x3_operands (same job struct as fgnn_pack_operands, fgnn_pack_x3_floats floats per image; jobs of
 * kind 2 / 3 produce the fp32 images of fgnn_pack_operands kind 0 / 1 in the same launch, for steps that mix both kernel sets). */
int fgnn_mlp_x3_supported(int ca, int cb, int depth, int nmlp /* 1 or 2 forward; backward: 1 */);
int fgnn_pack_x3_floats(int kind, int ca, int cb, int depth, int nmlp);
int fgnn_pack_x3_operands(const fgnn_pack_job *jobs, int njobs, void *stream);
int fgnn_mlp_fwd_x3(const fgnn_mlp_fwd_args *args, void *stream);

/* ---- GraphNorm statistics ------------------------------------------------------------
 * replaces torch.mean / torch.var(unbiased=False) over (N,N) and the scale of normalize
 * (models/layers.py:71-80) incl. the ragged n = sum(mask) (:79).  Combines the tile
 * partials (two fixed-order wave sums: grand mean, then M2 + m (mean_t - mean)^2) into nrm[g,c] = {mean, a, q, r2} with
 *   var = M2/m, r2 = 1/(var+eps), q = 1/(2 sqrt(n (var+eps))), a = gn_weight[c] * q.     */
int fgnn_gn_finalize(const float *part, const float *cnt, const float *gn_weight /* (C) or NULL=1 */,
                     const int *nvalid, int G, int C, int N, float eps, float *nrm, void *stream);
/* two MLPs (sharing cnt, e.g. mlp1 / mlp2 of one fgnn_mlp_fwd call) in one launch */
/* the same two with `recs` statistics records per graph instead of fgnn_tiles_per_graph(N) (the output of fgnn_mlp_fwd_t16) */
int fgnn_gn_finalize_r(const float *part, const float *cnt, const float *gn_weight, const int *nvalid, int G, int C, int N, int recs,
                       float eps, float *nrm, void *stream);
int fgnn_gn_finalize2_r(const float *part0, const float *part1, const float *cnt, const float *gn_weight0, const float *gn_weight1,
                        const int *nvalid, int G, int C, int N, int recs, float eps, float *nrm0, float *nrm1, void *stream);
int fgnn_gn_finalize2(const float *part0, const float *part1, const float *cnt, const float *gn_weight0,
                      const float *gn_weight1, const int *nvalid, int G, int C, int N, float eps,
                      float *nrm0, float *nrm1, void *stream);
/* same record computed directly from a dense (G, C, ldp) tensor (two-pass), any C */
int fgnn_gn_stats(const float *x, long long gstride, long long ldp, const float *gn_weight,
                  const int *nvalid, int G, int C, int N, float eps, float *nrm, void *stream);
/* y = (z - mean) * a + beta on valid entries, 0 on padding (GraphNorm.forward, :68-69) */
int fgnn_gn_apply(const float *z, long long zgstride, long long ldz, const float *nrm, const float *beta /* (C) or NULL=0 */,
                  const int *nvalid, int G, int C, int N, float *y, long long ygstride, long long ldy, void *stream);

/* GraphNorm.forward (models/layers.py:47-80) of a dense tensor in ONE pass and ONE launch when a plane fits a workgroup's
 * registers (N*N <= 4096): statistics record (nrm, as fgnn_gn_stats) + y = (x - mean) a + beta (as fgnn_gn_apply).
 * fgnn_gn_plane_bwd: its autograd -- S1 / S2 of the plane (written to s12), dz = ca dy + cb (z - mean) + cc, and the affine
 * gradients d gn_weight / d gn_bias (fixed order over the graphs) -- replacing fgnn_gn_bwd_stats + _coef + _apply.        */
int fgnn_gn_plane_supported(int N);
int fgnn_gn_plane_fwd(const float *x, long long gstride, long long ldp, const float *gn_weight, const float *beta /* NULL = 0 */,
                      const int *nvalid, int G, int C, int N, float eps, float *y, long long ygstride, long long ldy,
                      float *nrm /* (G*C*4) */, void *stream);
int fgnn_gn_plane_bwd(const float *dy, long long dgstride, long long ldd, const float *z, long long zgstride, long long ldz,
                      const float *nrm, const int *nvalid, int G, int C, int N, float *dz, long long ogstride, long long ldo,
                      float *s12 /* (G*C*2) */, float *dgn_w /* (C) or NULL */, float *dgn_b /* (C) or NULL */, void *stream);

/* ---- generic-width 1x1 convolution (conv.hip) ------------------------------------------
 * One layer of MlpBlock_Real.forward, `out = activation(conv_layer(out))` (models/layers.py:125-131: nn.Conv2d(k=1, bias=True)
 * + F.relu), for channel widths the fused 32-wide fgnn_mlp_fwd / fgnn_mlp_bwd are not built for (any Cin = K and Cout = M up
 * to FGNN_CONV_MAX_CH).  fp32, exact fma chain over the input channels.
 *   y[g][o][p] = act( bias[o] + sum_k W[o*w_ostride + k*w_kstride] * xm[g][k][p] ),  act = ReLU when `relu`, else identity;
 *   xm = x where relu_mask > 0 (same strides as x) or x itself when relu_mask == NULL.  Pixels outside the valid n x n corner of
 *   a ragged graph (nvalid) are written as 0.
 * The same entry point gives the input gradient of a layer: x := dy, relu_mask := the layer's saved (post-ReLU) output,
 * W strides swapped (W^T), bias NULL, relu 0.                                                                            */
#define FGNN_CONV_MAX_CH 256
int fgnn_conv1x1(const float *x, long long x_gstride, long long x_ld, const float *relu_mask, const float *W,
                 long long w_ostride, long long w_kstride, const float *bias, int relu, const int *nvalid, int G, int N,
                 int M, int K, float *y, long long y_gstride, long long y_ld, void *stream);
/* Parameter gradients of one layer: with dz = dy where relu_mask > 0 (all of dy when NULL), restricted to valid pixels,
 *   dW[o][c] = sum_{g,p} dz[g][o][p] * x[g][c][p],  db[o] = sum_{g,p} dz[g][o][p]
 * as fgnn_conv1x1_dw_chunks(G, N) partial records of M*K + M floats ([dW (M,K) | db (M)]) in `wpart`; finish with
 * fgnn_reduce_partials(wpart, chunks, M*K + M, out) (fixed order: bit-reproducible).                                    */
int fgnn_conv1x1_dw_chunks(int G, int N);
int fgnn_conv1x1_dw(const float *dy, long long d_gstride, long long d_ld, const float *relu_mask, const float *x,
                    long long x_gstride, long long x_ld, const int *nvalid, int G, int N, int M, int K, float *wpart,
                    void *stream);

/* the same for up to FGNN_DW_MAX_JOBS layers in one launch (the convs of one MlpBlock_Real): a chunk's partial record is the
 * concatenation of the jobs' records [dW_0 | db_0 | dW_1 | db_1 | ...], so one fgnn_reduce_partials(wpart, chunks, sum of the
 * counts, out) finishes all of them */
#define FGNN_DW_MAX_JOBS 4
typedef struct {
    const float *dy; long long d_gstride, d_ld;
    const float *relu_mask;                 /* optional */
    const float *x; long long x_gstride, x_ld;
    int M, K;
} fgnn_dw_job;
int fgnn_conv1x1_dw_multi(const fgnn_dw_job *jobs, int njobs, const int *nvalid, int G, int N, float *wpart, void *stream);

/* A chain of up to three 1x1 convolutions in ONE launch, the intermediate activations never leaving the register file
 * (channel widths the fused 32-wide kernels are not built for; conv.hip).  Layer l maps K_l -> M_l channels (K_{l+1} = M_l):
 *     t_l = W_l in_l + bias_l;   t_l = relu(t_l) if relu;   t_l = t_l where mask_l > 0 else 0 if mask_l;   out_l = t_l
 * in_0 = x, in_{l+1} = t_l; padding pixels of ragged graphs are exact zeros in every out_l.  Two uses:
 *   forward of MlpBlock_Real's conv stack (models/layers.py:125-131): relu on all but the last layer, every out_l kept for
 *     the backward;
 *   its input-gradient chain: layers in reverse with W^T given by strides, no bias, mask_l = the saved post-ReLU activation
 *     the gradient flows into, out_l = d(pre-activation) of that layer (what fgnn_conv1x1_dw then takes with relu_mask = NULL).
 * Limits (fgnn_conv_chain_supported): depth <= 3, K_0 <= 128, every M_l <= 128, hidden widths (M_l, l < depth-1) <= 64.   */
typedef struct {
    const float *W;                         /* element (o, k) at W[o * w_ostride + k * w_kstride]                        */
    long long w_ostride, w_kstride;
    const float *bias;                      /* (M) or NULL                                                               */
    int M, K;
    int relu;
    const float *mask;                      /* optional (G, M, N*N) with the strides of `out`                            */
    float *out;                             /* optional (G, M, N*N)                                                      */
    long long o_gstride;                    /* graph stride of mask / out (their channel stride is fgnn_chain_args.o_ld)   */
} fgnn_chain_layer;
typedef struct {
    const float *x;
    long long x_gstride, x_ld;
    int depth;
    fgnn_chain_layer layer[3];
    long long o_ld;                         /* channel stride of every layer's mask / out                                */
    const int *nvalid;
    int G, N;
} fgnn_chain_args;
int fgnn_conv_chain_supported(int depth, int K0, const int *M /* depth widths */);
int fgnn_conv_chain(const fgnn_chain_args *args, void *stream);

/* ---- the conv stack of a 64-wide MlpBlock_Real, fused (csrc/mlp64.hip; models/layers.py:113-131 with out_features = 64, depth_of_mlp = 3:
 * the three MLPs of a 64-feature block, models/blocks_emb.py:16-36) -------------------------------------------------------------------
 *   forward   out = W2 relu(W1 relu(W0 x + b0) + b1) + b2  (zeros outside the valid corner); nothing else is written;
 *   backward  recomputes the hidden activations from x, then dx (optional) and one row of parameter-gradient partials per workgroup:
 *             wpart[wg][fgnn_mlp64_param_count(cin)] = [dW0 (64 x K0P, K0P = cin rounded up to 32: columns >= cin are zero) | db0 (64) |
 *             dW1 (64 x 64) | db1 | dW2 | db2]; fgnn_reduce_partials(wpart, fgnn_mlp64_num_workgroups(), count, ...) sums the rows.
 * fgnn_mlp64_pack turns the nn.Conv2d parameters (W0 (64, cin), W1, W2 (64, 64) row-major; biases (64) or NULL) into the operand record
 * both directions read (MFMA operand order, forward and transposed images; one small launch per forward call).  16-pixel tiles on
 * v_mfma_f32_16x16x4_f32; the backward keeps a wave's whole parameter-gradient set in its (AGPR) registers, one wave per SIMD.
 * fgnn_mlp64_supported: depth 3, width 64, cin in 1..128; N <= 256. */
typedef struct {
    const float *x;                          /* (G, cin, N*N) */
    long long x_gstride, x_ld;
    int cin;
    const float *xb;                         /* optional second slab (G, cb, N*N) stacked after x along the channels (mlp3 of a block reads  */
    long long xb_gstride, xb_ld;             /* [mult ; in] without the concatenated copy): cin == 64 then, cb <= 64, W0 is                   */
    int cb;                                  /* (64, cin + cb); NULL: one slab                                                               */
    const float *packed;                     /* fgnn_mlp64_packed_floats(cin) floats written by fgnn_mlp64_pack */
    const int *nvalid;                       /* (G) or NULL */
    int G, N;
    float *out;                              /* forward: (G, 64, N*N) */
    long long o_gstride, o_ld;
    const float *dz;                         /* backward: gradient of out */
    long long dz_gstride, dz_ld;
    float *dx;                               /* backward: (G, cin, N*N) or NULL */
    long long dx_gstride, dx_ld;
    float *dxb;                              /* backward, two slabs: (G, cb, N*N), required when dx is given */
    long long dxb_gstride, dxb_ld;
    int accumulate_dx, accumulate_dxb;       /* != 0: dx (dxb) += instead of a store (the input also feeds other MLPs: their launches add to one buffer) */
    float *wpart;                            /* backward: (fgnn_mlp64_num_workgroups(), fgnn_mlp64_param_count(cin)) */
} fgnn_mlp64_args;
int fgnn_mlp64_supported(int cin, int depth, int width);
int fgnn_mlp64_num_workgroups(void);
int fgnn_mlp64_param_count(int cin);
int fgnn_mlp64_packed_floats(int cin);
int fgnn_mlp64_pack(const float *W0, const float *W1, const float *W2, const float *b0, const float *b1, const float *b2, int cin,
                    float *packed, void *stream);
typedef struct {
    const float *W[3];                       /* W0 (64, cin), W1, W2 (64, 64) */
    const float *bias[3];                    /* (64) or NULL */
    int cin;
    float *packed;                           /* fgnn_mlp64_packed_floats(cin) floats */
} fgnn_mlp64_pack_job;
/* ... for up to 16 MLPs in one launch (every MLP of a model at the top of its forward pass) */
int fgnn_mlp64_pack_multi(const fgnn_mlp64_pack_job *jobs, int njobs, void *stream);
int fgnn_mlp64_fwd(const fgnn_mlp64_args *args, void *stream);
int fgnn_mlp64_bwd(const fgnn_mlp64_args *args, void *stream);

/* ---- Matmul.forward: per (g,c) N x N product (models/layers.py:161-162) --------------
 * out[g,c] = Ya[g,c] @ Yb[g,c], Y = normalised slab (or the raw slab when nrm == NULL). */
int fgnn_chan_matmul_fwd(const fgnn_slab *ya, const fgnn_slab *yb, const int *nvalid, int G, int N,
                         float *out, long long ogstride, long long ldo, void *stream);
/* Ragged batches (64 < N <= 256, one workgroup per matrix): the same product with the workgroups issued in the order of
 * `order` (G graph indices from fgnn_ragged_tile_ranges_order: largest graph first), so that the hardware's in-order
 * dispatch is a longest-job-first schedule -- the matrices of a batch differ by up to (Nmax / nmin)^3 in work.  Results
 * do not depend on the order (every matrix is computed by one workgroup on its own).  order == NULL: as above.          */
/* fill: what the kernel zeroes of the output outside the valid corner (64 < N <= 256).  0: all of it.  1: only what a consumer
 * that steps over padding-only tiles (fgnn_ragged_tile_ranges) can read -- the row tails of the valid rows up to the next multiple
 * of 32 columns and the head of the row after the last one; the rest of the frame is left as it was.                              */
int fgnn_chan_matmul_fwd_ord(const fgnn_slab *ya, const fgnn_slab *yb, const int *nvalid, int G, int N,
                             float *out, long long ogstride, long long ldo, const int *order, int fill, void *stream);
/* The same product with the work of fgnn_gn_finalize2 folded into its prologue (one workgroup or wave per matrix: N <= 256):
 * each (g,c) workgroup finalizes the two GraphNorm records it needs from the tile statistics (part_a / part_b /
 * cnt of the preceding two-MLP fgnn_mlp_fwd call) while its tile loads are in flight, normalises with them and
 * writes them to ya->nrm / yb->nrm for the later consumers -- one launch less per block.                     */
int fgnn_chan_matmul_fwd_fin_supported(int N);
int fgnn_chan_matmul_fwd_fin(const fgnn_slab *ya, const fgnn_slab *yb, const float *part_a, const float *part_b,
                             const float *cnt, const float *gn_weight_a, const float *gn_weight_b, float eps,
                             const int *nvalid, int G, int N, float *out, long long ogstride, long long ldo, void *stream);
int fgnn_chan_matmul_fwd_fin_ord(const fgnn_slab *ya, const fgnn_slab *yb, const float *part_a, const float *part_b,
                                 const float *cnt, const float *gn_weight_a, const float *gn_weight_b, float eps,
                                 const int *nvalid, int G, int N, float *out, long long ogstride, long long ldo,
                                 const int *order /* as fgnn_chan_matmul_fwd_ord; used for 64 < N <= 256 */, int fill, void *stream);
/* ... with `recs` statistics records per graph (the output of fgnn_mlp_fwd_t16) */
int fgnn_chan_matmul_fwd_fin_ord_r(const fgnn_slab *ya, const fgnn_slab *yb, const float *part_a, const float *part_b,
                                   const float *cnt, const float *gn_weight_a, const float *gn_weight_b, float eps,
                                   const int *nvalid, int G, int N, int recs, float *out, long long ogstride, long long ldo,
                                   const int *order, int fill, void *stream);

/* ---- ColumnMaxPooling.forward (models/layers.py:202-203; masked: maskedtensor.py:213-228)
 * e[g,c,i] = max_j y[g,c,i,j] (first index on ties), idx int32; rows i >= nvalid -> 0.   */
int fgnn_colmax_fwd(const fgnn_slab *y, const int *nvalid, int G, int N, float *e /* (G,C,N) */,
                    int *idx /* (G,C,N) */, void *stream);
/* the same with the GraphNorm finalize of its input folded in (N <= 64): every (g,c) wave combines the tile
 * statistics (part, cnt of the fgnn_mlp_fwd call that produced y) itself and writes the record to y->nrm */
int fgnn_colmax_fwd_fin_supported(int N);
int fgnn_colmax_fwd_fin(const fgnn_slab *y, const float *part, const float *cnt, const float *gn_weight, float eps,
                        const int *nvalid, int G, int N, float *e, int *idx, void *stream);
int fgnn_colmax_fwd_fin_r(const fgnn_slab *y, const float *part, const float *cnt, const float *gn_weight, float eps,
                          const int *nvalid, int G, int N, int recs /* statistics records per graph (fgnn_mlp_fwd_t16) */, float *e, int *idx,
                          void *stream);

/* ---- Siamese scoring + triplet_loss (models/trainers.py:67, toolbox/losses.py:20-34) ---
 * e1,e2: (B, C, N).  scores[b] = e1[b]^T e2[b] (B,N,N); lse (B,N) row log-sum-exp;
 * pair_loss (B * FGNN_SCORE_SPLIT): partial sums of (lse_i - scores[b,i,i]) over the valid rows of
 * FGNN_SCORE_SPLIT row blocks per pair (the loss of pair b is the sum of its FGNN_SCORE_SPLIT entries). */
#define FGNN_SCORE_SPLIT 4
int fgnn_score_ce_fwd(const float *e1, const float *e2, const int *nvalid, int B, int C, int N,
                      float *scores, float *lse, float *pair_loss, void *stream);
/* the same with `row_blocks` row blocks per pair instead of FGNN_SCORE_SPLIT (pair_loss: B * row_blocks partial sums);
 * fgnn_score_row_blocks(B, N) is the count that fills the chip for a batch of B pairs of N vertices */
int fgnn_score_row_blocks(int B, int N);
int fgnn_score_ce_fwd_blocks(const float *e1, const float *e2, const int *nvalid, int B, int C, int N, int row_blocks,
                             float *scores, float *lse, float *pair_loss, void *stream);
/* d e1, d e2 of loss = sum_b pair_loss[b] * (*gscale)   (gscale: device scalar = grad / sum n) */
int fgnn_score_ce_bwd(const float *e1, const float *e2, const float *scores, const float *lse,
                      const int *nvalid, const float *gscale, int B, int C, int N,
                      float *de1, float *de2, void *stream);
/* Scoring forward + triplet loss + their backward in ONE launch -- what a training step issues back to back (models/trainers.py:60-76:
 * `loss = self.loss(self(x1, x2))` followed at once by autograd's first two nodes): the outputs of fgnn_score_ce_fwd_blocks (scores,
 * lse, pair_loss with `row_blocks` partial sums per pair) AND of fgnn_score_ce_bwd (de1, de2; gscale as there), bit-identical to
 * the two launches.  Small batches of small graphs: fgnn_score_ce_step_supported(B, C, N) (N <= 64, B < 64, 8 | C). */
int fgnn_score_ce_step_supported(int B, int C, int N);
int fgnn_score_ce_step(const float *e1, const float *e2, const int *nvalid, const float *gscale, int B, int C, int N, int row_blocks,
                       float *scores, float *lse, float *pair_loss, float *de1, float *de2, void *stream);

/* triplet_loss on a given score tensor: lse (B,N), pair_loss (B) (toolbox/losses.py:27-34) */
int fgnn_ce_fwd(const float *scores, const int *nvalid, int B, int N, float *lse, float *pair_loss, void *stream);
/* dscores = (softmax_row(scores) - I) * (*gscale) on valid entries, 0 on padding */
int fgnn_ce_bwd(const float *scores, const float *lse, const int *nvalid, const float *gscale, int B, int N,
                float *dscores, void *stream);
/* d scores given (B,N,N) -> d e1, d e2 (plain bmm backward, for the module-level API) */
int fgnn_score_bwd(const float *e1, const float *e2, const float *dscores, const int *nvalid,
                   int B, int C, int N, float *de1, float *de2, void *stream);

/* ---- backward ---------------------------------------------------------------------- */
/* ColumnMaxPooling backward: dy[g,c,i,idx] = de[g,c,i], 0 elsewhere (dense write).
 * Optional: y (the normalised-on-load input slab of the pooling) and s12 (G*C*2) -> also emits
 * the GraphNorm-backward sums {sum dy, sum dy*(z-mean)} of the MLP that produced y.         */
int fgnn_colmax_bwd(const float *de, const int *idx, const int *nvalid, int G, int C, int N,
                    float *dy, long long gstride, long long ldp, const fgnn_slab *y, float *s12, void *stream);

/* GraphNorm backward reductions per (g,c): S1 = sum dy, S2 = sum dy * (z - mean).         */
int fgnn_gn_bwd_stats(const float *dy, long long dgstride, long long ldd,
                      const float *z, long long zgstride, long long ldz, const float *nrm,
                      const int *nvalid, int G, int C, int N, float *s12 /* (G*C*2) */, void *stream);
/* coefficients of dz = ca*dy + cb*(z-mean) + cc : coef[g,c] = {mean, ca, cb, cc};
 * plus d gn_weight[c] = sum_g q*S2, d gn_bias[c] = sum_g S1 (fixed order over g).         */
int fgnn_gn_bwd_coef(const float *s12, const float *nrm, const int *nvalid, int G, int C, int N,
                     float *coef /* (G*C*4) */, float *dgn_w /* (C) or NULL */, float *dgn_b /* (C) or NULL */,
                     void *stream);
/* coefficients of two MLPs in one launch (no affine gradients) */
int fgnn_gn_bwd_coef2(const float *s12_0, const float *s12_1, const float *nrm0, const float *nrm1,
                      const int *nvalid, int G, int C, int N, float *coef0, float *coef1, void *stream);
/* same coefficients from per-tile partial sums (G, tpg, C, 2) as emitted by fgnn_mlp_bwd
 * (s12part); also writes the summed s12 (G*C*2) for the affine gradients.                  */
int fgnn_gn_bwd_coef_tiles(const float *s12part, const float *nrm, const int *nvalid, int G, int C, int N,
                           float *s12, float *coef, void *stream);
/* dense dz = ca*dy + cb*(z-mean) + cc on valid entries (module-level GraphNorm backward) */
int fgnn_gn_bwd_apply(const float *dy, long long dgstride, long long ldd,
                      const float *z, long long zgstride, long long ldz, const float *coef,
                      const int *nvalid, int G, int C, int N, float *dz, long long ogstride, long long ldo, void *stream);

/* MlpBlock_Real backward (autograd of models/layers.py:126-131): recomputes the hidden
 * activations from the input slabs, forms dz from (dy, z, coef), back-propagates through
 * the convs and accumulates per-workgroup partial weight/bias gradients.                 */
typedef struct {
    int G, N, depth;
    const int *nvalid;
    fgnn_slab a, b;                          /* forward inputs (recompute)                */
    const float *W[FGNN_MAX_DEPTH];
    const float *bias[FGNN_MAX_DEPTH];
    const float *dy;  long long dgstride, ldd;   /* grad of the normalised output (G,32,ldd) */
    const float *z;   long long zgstride, ldz;   /* saved pre-norm output                    */
    const float *coef;                       /* (G*32*4) from fgnn_gn_bwd_coef*, or NULL: then the kernel */
    const float *s12;                        /*   derives it from s12 (G*32*2) = {sum dy, sum dy*(z-mean)}  */
    const float *znrm;                       /*   and znrm (G*32*4), the GraphNorm record of z              */
    float *dxa; long long dxa_gstride, dxa_ld;   /* out: grad wrt slab a (NULL = not needed) */
    float *dxb; long long dxb_gstride, dxb_ld;   /* out: grad wrt slab b (NULL = not needed) */
    int accumulate_a, accumulate_b;          /* 1: dx += (read-modify-write)             */
    float *wpart;                            /* out (num_wg, fgnn_mlp_param_count) partial dW/db */
    float *s12part;                          /* optional out (G, tpg, 32, 2): per-tile {sum dxa, sum dxa*(z_a-mean_a)} of
                                                the FINAL dxa values (needs a.C == 32, a.nrm and dxa) -- the GraphNorm
                                                backward sums of the MLP that produced slab a */
    const float *packed;                     /* optional: operand image from fgnn_pack_operands (kind 1) */
    const float *s12tiles;                   /* optional (G, tpg, 32, 2): per-tile sums for THIS MLP's output as emitted by its
                                                consumer's s12part; with znrm they replace coef / s12 -- every workgroup sums
                                                the tiles of the graphs it touches in its prologue (the work of
                                                fgnn_gn_bwd_coef_tiles without its launch; see ..._coef_tiles_supported).
                                                Two-slab MLPs (b.C > 0, i.e. mlp3) only                                 */
    float *s12_out;                          /* optional (G*32*2): the summed s12 is also written here (affine gradients) */
    const unsigned *xbits;                   /* optional: 2-channel slab expanded from the bit-packed adjacency, as in  */
    const float *xdeg;                       /*   fgnn_mlp_fwd_args                                                      */
    const int *ranges;                       /* optional work-balanced tile bounds + padding-tile skipping, as in the forward
                                                arguments; cannot be combined with s12tiles                                  */
    int cu_share;                            /* as in fgnn_mlp_fwd_args; 2 = fgnn_mlp_bwd_num_workgroups() / 2 workgroups and rows of
                                                wpart.  IGNORED when `ranges` is given: a ragged launch always runs the full grid
                                                and writes fgnn_mlp_bwd_num_workgroups() rows of wpart -- size wpart for that */
} fgnn_mlp_bwd_args;
int fgnn_mlp_bwd(const fgnn_mlp_bwd_args *args, void *stream);
int fgnn_mlp_bwd_x3(const fgnn_mlp_bwd_args *args, void *stream);   /* the x3 form (see fgnn_mlp_fwd_x3): image of kind 1 from
                                                                         fgnn_pack_x3_operands; input gradients for 32-channel slabs */
/* mlp1 + mlp2 of one block (models/blocks_emb.py:16-27: two MlpBlock_Real on the same input; their autograd) in ONE launch:
 * `m1` / `m2` are the argument blocks of the two fgnn_mlp_bwd calls it replaces, describing the same input slab; the input
 * gradient (dxa, accumulate_a) and its tile sums (s12part) are given in m2 only and receive (old + dx1) + dx2, bit-identical to
 * the two accumulating launches.  Depth 3, one slab of 2 or 32 channels, constant-size batches, both operand images. */
int fgnn_mlp_bwd_pair_supported(int ca, int depth);
int fgnn_mlp_bwd_pair(const fgnn_mlp_bwd_args *m1, const fgnn_mlp_bwd_args *m2, void *stream);
/* The same launch with every contraction on the bf16 matrix cores through the exact three-way operand split (the arithmetic of
 * fgnn_mlp_bwd_x3 / fgnn_mlp_fwd_x3: the recompute reproduces the x3 forward bit for bit): both images are of kind 1 from
 * fgnn_pack_x3_operands; the weight-gradient operands are transposed on the matrix pipe instead of through LDS tiles
 * (csrc/mlp_bwd_pair_x3.hip).  d_in is bit-identical to two accumulating fgnn_mlp_bwd_x3 launches.  No `ranges`. */
int fgnn_mlp_bwd_pair_x3(const fgnn_mlp_bwd_args *m1, const fgnn_mlp_bwd_args *m2, void *stream);
/* Round 6: the same launch on 16-pixel tiles / v_mfma_f32_16x16x4_f32 (csrc/mlp_bwd_pair_t16.hip, csrc/fgnn_t16.h) -- same arguments,
 * work unit (a 32-pixel tile, processed as two halves), S1/S2 records and partial rows as fgnn_mlp_bwd_pair, so it is a drop-in; the
 * operand images are of kind 5 (fgnn_pack_operands; kind 4 = the forward image for the *_t16 forward kernel).  The recomputed hidden
 * activations follow the SAME fma sequence as the 32-pixel kernels (bit-identical ReLU masks); d_in is one fma chain over (the
 * gradient mlp3 left, mlp1's terms, mlp2's terms) instead of three separately rounded sums, the weight-gradient sums run in another
 * pixel order: equal to the 32-pixel kernel to fp32 rounding, not bit for bit.  Depth 3, one dense 32-channel slab, N <= 256,
 * constant-size and ragged batches (nvalid, ranges). */
/* ... and fgnn_mlp_bwd for a two-slab MLP (mlp3 of a block: [mult ; in]) on 16-pixel tiles (csrc/mlp_bwd_t16.hip): work is assigned in
 * 16-pixel halves (4.94 per wave instead of 2.47 32-pixel tiles at the benchmarked shape).  Same argument block, partial rows and
 * results (to fp32 rounding) as fgnn_mlp_bwd; image of kind 5.  fgnn_mlp_bwd_t16_supported(args) says whether an argument block is one
 * of the built shapes: depth 3, slab a = 32 raw channels with dxa stored, slab b = 32 normalised channels with dxb stored, or 2 raw
 * channels (dense or bit-packed) without dxb; no accumulation; N <= 256. */
int fgnn_mlp_bwd_t16_supported(const fgnn_mlp_bwd_args *args);
int fgnn_mlp_bwd_t16(const fgnn_mlp_bwd_args *args, void *stream);
int fgnn_mlp_bwd_pair_t16_supported(int ca, int depth);
int fgnn_mlp_bwd_pair_t16(const fgnn_mlp_bwd_args *m1, const fgnn_mlp_bwd_args *m2, void *stream);
#define FGNN_BWD_COEF_GRAPHS 4
int fgnn_mlp_bwd_coef_tiles_supported(int G, int N);   /* s12tiles usable: a workgroup spans <= FGNN_BWD_COEF_GRAPHS graphs */
/* floats per workgroup in `wpart` for an MLP with Cin input channels and `depth` convs:
 * layout [W0 (32*Cin) | b0 (32) | W1 (32*32) | b1 (32) | ...]                           */
int fgnn_mlp_param_count(int Cin, int depth);
/* deterministic reduction of the per-workgroup partials: out[i] = sum_w wpart[w][i]      */
int fgnn_reduce_partials(const float *wpart, int num_wg, int count, float *out, void *stream);

/* One launch that finishes the parameter gradients of up to FGNN_MAX_GRAD_JOBS MLPs:
 * out[i] = sum_w wpart[w][i] (fixed order) and, when s12 is given, the GraphNorm affine
 * gradients dgn_w[c] = sum_g q[g,c]*S2[g,c], dgn_b[c] = sum_g S1[g,c].                      */
#define FGNN_MAX_GRAD_JOBS 16
typedef struct {
    const float *wpart; int count; float *out;
    const float *s12; const float *nrm; float *dgn_w; float *dgn_b;
    int rows;      /* rows of wpart to sum; 0 = num_wg (the MLP backward partials) */
    float scale;   /* factor applied to the sums; 0 = 1 (lets the loss = sum pair_loss / nodes ride along) */
    const float *scale_dev;   /* optional device scalar multiplied in as well (1 / sum(n) of a ragged batch computed on the device:
                                 fgnn_inv_node_count) */
} fgnn_grad_job;
int fgnn_grad_finalize(const fgnn_grad_job *jobs, int njobs, int num_wg, int G, int C, void *stream);
/* out[0] = 1 / sum_{b < B} nvalid[b] (0 if the sum is 0): the normaliser of triplet_loss('mean') on a ragged batch
 * (toolbox/losses.py:27-34) as a device scalar -- fgnn_score_ce_bwd's gscale and fgnn_grad_job.scale_dev read it, so a captured step
 * serves batches of any node count without a host round trip. */
int fgnn_inv_node_count(const int *nvalid, int B, float *out, void *stream);

/* Matmul backward: da = dm @ Yb^T, db = Ya^T @ dm per (g,c) (autograd of layers.py:161-162) */
int fgnn_chan_matmul_bwd(const fgnn_slab *ya, const fgnn_slab *yb, const float *dm, long long dmgstride, long long ldm,
                         const int *nvalid, int G, int N,
                         float *da, float *db, long long ogstride, long long ldo,
                         float *s12a /* optional (G*C*2): {sum da, sum da*(z_a-mean_a)} */,
                         float *s12b /* optional (G*C*2) */, void *stream);
/* longest-job-first order as in fgnn_chan_matmul_fwd_ord; with an order the two products of a matrix are separate work
 * items (dm is then read by both) */
int fgnn_chan_matmul_bwd_ord(const fgnn_slab *ya, const fgnn_slab *yb, const float *dm, long long dmgstride, long long ldm,
                             const int *nvalid, int G, int N, float *da, float *db, long long ogstride, long long ldo,
                             float *s12a, float *s12b, const int *order, int fill /* as fgnn_chan_matmul_fwd_ord */, void *stream);

/* ---- next to the hot path -------------------------------------------
 * Fused Adam over a flat fp32 buffer: torch.optim.Adam(amsgrad=False, weight_decay=0) single-tensor
 * semantics (models/trainers.py:92-104); grads are multiplied by grad_scale first; step >= 1.      */
int fgnn_adam_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int n, double lr,
                   double beta1, double beta2, double eps, int step, double grad_scale, void *stream);
/* the same update for a captured / replayed launch: hp (device, 5 doubles) = {lr, beta1, beta2, eps, grad_scale};
 * state (device, 2 ints, zero-initialised) = {steps taken, internal arrival counter}; the kernel advances state[0] */
int fgnn_adam_step_dev(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int n,
                       const double *hp, int *state, void *stream);
/* accuracy_max (toolbox/metrics.py:119-141): correct[b] = #{i < n_b : argmax_j scores[b,i,j] == i},
 * first maximum on ties (np.argmax); int32, bit-exact.                                              */
int fgnn_accuracy_max(const float *scores, const int *nvalid, int B, int N, int *correct, void *stream);
/* accuracy_linear_assignment (toolbox/metrics.py:92-116): per graph b the minimum-cost perfect matching of the n_b x n_b corner
 * of cost[b] (= -log_softmax(scores[b]); row pitch ld, graphs bstride apart), correct[b] = #{i : matched column of row i == i};
 * assign (optional, (B, N) int32) receives the matched column of every row (-1 in the padding).  The algorithm, arithmetic
 * (fp64) and tie rules are those of scipy.optimize.linear_sum_assignment (Crouse's shortest augmenting paths), so the
 * ASSIGNMENT equals SciPy's, ties included (tests/test_gpu_lsap.py).  A cost matrix without a finite matching (SciPy raises)
 * yields correct[b] = 0 and assign = -1.  No device->host copy, no host loop over the graphs.                          */
#define FGNN_LSAP_MAX_N 2048
int fgnn_lsap_accuracy(const float *cost, long long bstride, int ld, const int *nvalid, int B, int N, int *correct,
                       int *assign /* optional */, void *stream);

/* ---- block 1 on its structured input (csrc/block1_struct.hip) -----------------------------------------------------------
 * The reference always feeds block 1 the tensor representation of a graph (loaders/data_generator.py:118-125: channel 0 = the
 * 0/1 adjacency, channel 1 = diag(row sums)), so mlp1 / mlp2 of block 1 (models/blocks_emb.py:16-27, models/layers.py:126-131)
 * take one value per input class -- off-diagonal w = 0 / 1, diagonal (w_ii, deg_i) -- and their per-channel product
 * (models/layers.py:161-162) has a closed form in W, W^2, the degrees and the class values.  For bit-packed inputs of
 * batches (constant-size or ragged) with N <= 256, depth 3, these entry points replace fgnn_mlp_fwd (mlp1 + mlp2) + fgnn_chan_matmul_fwd
 * and fgnn_chan_matmul_bwd + fgnn_mlp_bwd_pair of block 1: same function, another evaluation order (equal to fp32 rounding).
 *   tables : (2 models, 2 + 2 (N + 1) classes, {h1, h2, z, z as stored}, 32) floats, graph independent: once per step.
 *            bf16_scheme != 0: the arithmetic of the 16-bit engine (matrix-core operands R(W), R(relu(.)), stored R(z))
 *   fwd    : GraphNorm records nrm1 / nrm2 (G, 32, 4) of mlp1 / mlp2 and the raw slab mult (G, 32, ldp)
 *   bwd    : from d(mult): the first fgnn_block1_struct_rows(G, N) rows of wpart1 / wpart2 (the partial layout of fgnn_mlp_bwd; the
 *            other rows are not written: reduce exactly these, fgnn_grad_job.rows) and s12_1 / s12_2 (G, 32, 2), ready for
 *            fgnn_grad_finalize
 * The ...16 forms take the bf16 slabs of the 16-bit engine (row pitch ldr elements, channel stride ldp, tables built with
 * bf16_scheme = 1): mult is rounded to nearest even on store, the statistics stay fp32 and the class sums of the backward pass
 * are formed in fp32 from the bf16 d(mult) (the generic 16-bit kernels round every pixel of dY1 / dY2 / dz instead).          */
int fgnn_block1_struct_supported(int N, int depth, int original_features_num);      /* N <= 256, depth 3, 2 input channels */
int fgnn_block1_struct_table_floats(int N);
long long fgnn_block1_struct_ws_floats(int G, int N);       /* workspace shared by fwd and bwd of one step (16-byte aligned) */
int fgnn_block1_struct_rows(int G, int N);                  /* bwd writes rows 0 .. rows-1 of wpart1 / wpart2 (<= fgnn_mlp_bwd_num_workgroups()) and no others; G > that many rows: row b sums the graphs b, b + rows, ... */
int fgnn_block1_struct_tables(const float *const *W1, const float *const *b1, const float *const *W2, const float *const *b2, int N,
                              int bf16_scheme, float *tables, void *stream);
/* nvalid: optional per-graph vertex counts (ragged batches: the N x N planes are padded; mult is written as 0 outside the valid
 * corner, class counts and the GraphNorm n use nvalid[g]).  fwd fills ws (16-bit code plane (W^2)_ij | w_ij << 15, per-vertex
 * {row sum, column sum, w_ii, class}); bwd of the same step reads it. */
int fgnn_block1_struct_fwd(const unsigned *bits, const int *nvalid, int G, int N, const float *tables, const float *gnw1, const float *gnb1,
                           const float *gnw2, const float *gnb2, float eps, float *nrm1, float *nrm2, float *mult, long long gstride,
                           long long ldp, float *xdeg /* optional: (G, N) row sums, as fgnn_adjacency_degree writes them */, float *ws,
                           const float *const *tW1, const float *const *tb1, const float *const *tW2, const float *const *tb2
                           /* optional, all four or none: the arguments of fgnn_block1_struct_tables -- `tables` is then (re)built
                              inside the first launch of this call instead of by a launch of its own */,
                           void *stream);
int fgnn_block1_struct_bwd(const unsigned *bits, const int *nvalid, int G, int N, const float *tables, const float *const *W1,
                           const float *const *W2, const float *nrm1, const float *nrm2, const float *gnb1, const float *gnb2,
                           const float *dmult, long long gstride, long long ldp, float *ws, float *wpart1, float *wpart2, float *s12_1,
                           float *s12_2, void *stream);
/* fgnn_block1_struct_fwd with the step's operand packing folded in: `jobs` (njobs <= FGNN_MAX_PACK_JOBS) are what fgnn_pack_operands would be
 * launched with; they run as extra workgroups of the first launch (which, like them, depends on the weights and the input only), one launch
 * less per step.  Results identical to the two entry points called one after the other. */
int fgnn_block1_struct_fwd_pack(const unsigned *bits, const int *nvalid, int G, int N, const float *tables, const float *gnw1, const float *gnb1,
                                const float *gnw2, const float *gnb2, float eps, float *nrm1, float *nrm2, float *mult, long long gstride,
                                long long ldp, float *xdeg, float *ws, const float *const *tW1, const float *const *tb1,
                                const float *const *tW2, const float *const *tb2, const fgnn_pack_job *jobs, int njobs, void *stream);
/* x16 (optional): the (G, 2, ldp) bf16 input slab (channel 0 = W, channel 1 = diag(row sums)) the later kernels of block 1 read */
int fgnn_block1_struct_fwd16(const unsigned *bits, const int *nvalid, int G, int N, int ldr, const float *tables, const float *gnw1,
                             const float *gnb1, const float *gnw2, const float *gnb2, float eps, float *nrm1, float *nrm2,
                             void *mult /* bf16 */, long long gstride, long long ldp, void *x16 /* bf16, optional */, float *ws,
                             const float *const *tW1, const float *const *tb1, const float *const *tW2, const float *const *tb2, void *stream);
/* ... and fgnn_block1_struct_fwd16 with the jobs of fgnn_pack16_operands riding in its first launch */
int fgnn_block1_struct_fwd16_pack(const unsigned *bits, const int *nvalid, int G, int N, int ldr, const float *tables, const float *gnw1,
                                  const float *gnb1, const float *gnw2, const float *gnb2, float eps, float *nrm1, float *nrm2, void *mult,
                                  long long gstride, long long ldp, void *x16, float *ws, const float *const *tW1, const float *const *tb1,
                                  const float *const *tW2, const float *const *tb2, const fgnn_pack_job *jobs, int njobs, void *stream);
int fgnn_block1_struct_bwd16(const unsigned *bits, const int *nvalid, int G, int N, int ldr, const float *tables, const float *const *W1,
                             const float *const *W2, const float *nrm1, const float *nrm2, const float *gnb1, const float *gnb2,
                             const void *dmult /* bf16 */, long long gstride, long long ldp, float *ws, float *wpart1, float *wpart2,
                             float *s12_1, float *s12_2, void *stream);

/* Input expansion (loaders/data_generator.py:118-125): bits (G, N, ceil(N/32)) uint32, bit j of row i =
 * W[i][j]  ->  x (G, 2, N, N) fp32 with x[g,0] = W, x[g,1] = diag(row sums); exact 0/1/integer values. */
int fgnn_expand_adjacency(const unsigned *bits, const int *nvalid, int G, int N, float *x, void *stream);
/* ... and back: the (G, 2, N, N) tensor representation a dense loader produced -> bit-packed adjacency (rows / columns >= nvalid[g]
 * read as empty).  bad (optional device int, zeroed by the caller): set to 1 if x is NOT a tensor representation on the valid corners
 * (an entry of channel 0 outside {0, 1}, channel 1 != diag(row sums of channel 0)) -- the structured block 1 must not see such bits. */
int fgnn_pack_adjacency(const float *x, const int *nvalid, int G, int N, unsigned *bits, int *bad, void *stream);
/* The same with the loader's padded size Nin (x is (G, 2, Nin, Nin)) smaller than the engine's N (bits are (G, N, ceil(N/32)): rows
 * and columns >= Nin are empty): what Siamese_Node_Exp.fused_step(input_form='tensor_representation') runs on each side of a dense or
 * MaskedTensor loader batch (loaders/loaders.py:5-15) in front of the structured block 1.  `bad` is OR-ed (sticky until the caller
 * zeroes it); an nvalid[g] outside [0, Nin] also sets it. */
int fgnn_pack_adjacency_ld(const float *x, const int *nvalid, int G, int Nin, int N, unsigned *bits, int *bad, void *stream);
/* Both sides of a siamese batch in ONE launch (loaders/loaders.py:12-15 yields the two sides as two tensors): bits (2 B, N, words) = the B graphs
 * of x1 followed by the B graphs of x2, each side with its own vertex counts (both NULL: constant-size).  Optionally the counts are copied to
 * nv_out (2 B entries: the engine's nvalid) and 1 / sum(nvalid1) -- the normaliser of triplet_loss 'mean', toolbox/losses.py:27-34 -- is left
 * in inv_out (what fgnn_inv_node_count computes).  Same verdict flag as fgnn_pack_adjacency_ld. */
int fgnn_pack_adjacency_pair(const float *x1, const float *x2, const int *nvalid1, const int *nvalid2, int B, int Nin, int N, unsigned *bits,
                             int *nv_out, float *inv_out, int *bad, void *stream);
/* deg[g][i] = number of set bits j < nvalid[g] in row i (0 for rows >= nvalid[g]): the diagonal of channel 1, for the
 * kernels that expand the adjacency themselves (fgnn_mlp_fwd_args.xbits / xdeg) */
int fgnn_adjacency_degree(const unsigned *bits, const int *nvalid, int G, int N, float *deg, void *stream);

/* Ragged batches: work-balanced tile ranges for the persistent MLP kernels.  A tile (32 consecutive pixels of one graph's
 * N x N plane) that holds no pixel of the valid n x n corner costs a zero-fill, any other tile a full pass; ranges[w] ..
 * ranges[w+1] (w < FGNN_RANGE_WG) split the G * tiles_per_graph tiles into pieces of equal cost.  One small launch per
 * batch; the result depends on nvalid only (deterministic).  No reference counterpart: the reference computes the padded
 * (Nmax x Nmax) tensors in full (maskedtensors/maskedtensor.py:98-112).                                                  */
#define FGNN_RANGE_WG 256
int fgnn_ragged_tile_ranges(const int *nvalid, int G, int N, int *ranges /* FGNN_RANGE_WG + 1 */, void *stream);
/* the same launch also writes order[0 .. G): the graph indices sorted by nvalid, largest first (ties: lower index first),
 * the schedule of fgnn_chan_matmul_fwd_ord / fgnn_chan_matmul_bwd_ord */
int fgnn_ragged_tile_ranges_order(const int *nvalid, int G, int N, int *ranges /* FGNN_RANGE_WG + 1 */, int *order /* G */,
                                  void *stream);
/* the same for the bf16 slabs (tiles of 64 elements of the ldr-pitched planes) */
int fgnn_ragged_tile_ranges16(const int *nvalid, int G, int N, int ldr, int *ranges /* FGNN_RANGE_WG + 1 */, void *stream);

/* out[i] = sum_k in[k][i] * scale  (tiny fixed-order reduction used for the loss) */
int fgnn_sum_scale(const float *in, int rows, int cols, float scale, float *out, void *stream);

/* =====================================================================================================
 * bf16 variant of the same path (BASELINE config 4: N = 200 dense pairs, 16-bit).  The reference trains under
 * 16-bit AMP (commander_explore.py:120-122, Network.half models/utils.py:71-74); here activations and gradient
 * slabs are STORED as bf16, every channel contraction / per-channel N x N product runs on
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation, and biases, GraphNorm statistics, embeddings, scores, loss
 * and all parameter gradients stay fp32.  Rounding points: oracle/fgnn_oracle_bf16.py.
 *
 * Layout of a bf16 activation tensor (G, C, ldp): a channel holds N rows of `ldr` elements, ldr = N rounded up
 * to a multiple of 8 (rows 16-byte aligned), element (i, j) at i*ldr + j; ldp >= N*ldr is a multiple of 64
 * (channels 128-byte aligned).  Columns j >= N, like all padding of a ragged graph, are stored as exact zeros.
 * A "tile" of the bf16 MLP kernels is 64 consecutive elements of a channel; tile statistics / per-tile sums
 * (part, cnt, s12part) have fgnn_tiles_per_graph16(N, ldr) = ceil(N*ldr / 64) entries per graph. */
typedef struct {
    const void *ptr;         /* bf16 (G, C, ldp)                                        */
    long long gstride;       /* elements between graphs                                 */
    long long ldp;           /* elements between channels                               */
    int C;                   /* channels: 2 or 32 (0 = slab unused)                     */
    const float *nrm;        /* optional (G*C*4) fp32 {mean, a, q, r2}                  */
    const float *beta;       /* optional (C) fp32                                       */
} fgnn_slab16;
int fgnn_tiles_per_graph16(int N, int ldr);

/* fp32 (G, C, N, N) contiguous  ->  bf16 (G, C, ldp) with row pitch ldr (padding zero-filled), and back */
int fgnn_to_bf16(const float *x, const int *nvalid, int G, int C, int N, int ldr, void *y, long long gstride, long long ldp,
                 void *stream);
int fgnn_from_bf16(const void *y, long long gstride, long long ldp, int G, int C, int N, int ldr, float *x, void *stream);

/* operand images of the bf16 MLP kernels (weights rounded to bf16, biases fp32), packed once per step.
 * kind 0 = forward image (nmlp MLPs back to back), kind 1 = backward image (one MLP, W[0] / bias[0]). */
int fgnn_pack16_floats(int kind, int ca, int cb, int depth, int nmlp);   /* size in 4-byte units */
int fgnn_pack16_operands(const fgnn_pack_job *jobs, int njobs, void *stream);

/* MlpBlock_Real.forward minus the normalisation (models/layers.py:126-131), bf16 storage: see fgnn_mlp_fwd */
typedef struct {
    int G, N, ldr, depth, nmlp;
    const int *nvalid;
    fgnn_slab16 a, b;
    void *z[2];                             /* out bf16 (G, 32, ldz)                      */
    long long ldz;
    float *part[2];                         /* out (G, 32, tpg16, 2) {mean, M2} of the fp32 z (tile index fastest: the
                                               *_tpg helpers and fgnn_chan_matmul_fwd16_fin walk one (g, c) column) */
    float *cnt;                             /* out (G, tpg16)                             */
    const void *packed;                     /* operand image (kind 0) -- required          */
    const int *ranges;                      /* optional (ragged): tile bounds from fgnn_ragged_tile_ranges16, see fgnn_mlp_fwd_args.ranges */
} fgnn_mlp_fwd16_args;
int fgnn_mlp_fwd16(const fgnn_mlp_fwd16_args *args, void *stream);
/* Test-only: the same launch that also writes its ReLU decisions (the bf16 twin of fgnn_debug_mlp_fwd_masks): masks[m] is
 * (G, 2, 32, fgnn_tiles_per_graph16(N, ldr), 2) words, bit j of word (g, layer, channel, t, parity) = [hidden pre-activation of element
 * 64 t + 2 j + parity of the ldr-pitched plane > 0].  Constant-size batches, the fused engine's shapes. */
int fgnn_debug_mlp_fwd16_masks(const fgnn_mlp_fwd16_args *args, unsigned *masks0, unsigned *masks1, void *stream);

/* GraphNorm finalize / backward-coefficient helpers of the bf16 kernels: explicit tile count per graph (the fp32 entry
 * points derive it from FGNN_TILE) and partials in the bf16 kernels' layout (G, C, tpg, 2) -- tile index fastest -- instead
 * of the fp32 kernels' (G, tpg, C, 2) */
int fgnn_gn_finalize_tpg(const float *part, const float *cnt, const float *gn_weight, const int *nvalid, int G, int C, int N,
                         int tpg, float eps, float *nrm, void *stream);
int fgnn_gn_finalize2_tpg(const float *part0, const float *part1, const float *cnt, const float *gn_weight0,
                          const float *gn_weight1, const int *nvalid, int G, int C, int N, int tpg, float eps, float *nrm0,
                          float *nrm1, void *stream);
int fgnn_gn_bwd_coef_tiles_tpg(const float *s12part, const float *nrm, const int *nvalid, int G, int C, int N, int tpg,
                               float *s12, float *coef, void *stream);

/* Matmul.forward / backward on bf16 slabs (models/layers.py:161-162), N <= 256: operands normalised on load and
 * rounded to bf16, fp32 accumulation, outputs rounded to bf16; s12a / s12b as in fgnn_chan_matmul_bwd (sums of the
 * ROUNDED outputs). */
int fgnn_chan_matmul_fwd16(const fgnn_slab16 *ya, const fgnn_slab16 *yb, const int *nvalid, int G, int N, int ldr,
                           void *out, long long ogstride, long long ldo, void *stream);
/* the forward product with the GraphNorm finalize of its two operands folded into the prologue (replaces a
 * fgnn_gn_finalize2_tpg launch): part_a / part_b / cnt are the tile statistics of the fgnn_mlp_fwd16 call that produced the
 * operands; the records are written to ya->nrm / yb->nrm (G*C*4 floats each, for the backward pass) and used at once */
int fgnn_chan_matmul_fwd16_fin(const fgnn_slab16 *ya, const fgnn_slab16 *yb, const float *part_a, const float *part_b,
                               const float *cnt, const float *gn_weight_a, const float *gn_weight_b, float eps, int tpg,
                               const int *nvalid, int G, int N, int ldr, void *out, long long ogstride, long long ldo,
                               void *stream);
int fgnn_chan_matmul_bwd16(const fgnn_slab16 *ya, const fgnn_slab16 *yb, const void *dm, long long dmgstride, long long ldm,
                           const int *nvalid, int G, int N, int ldr, void *da, void *db, long long ogstride, long long ldo,
                           float *s12a, float *s12b, void *stream);

/* the same with S2 derived from the trace term T = <dM, M> instead of re-reading the two raw operand slabs:
 * tpart (G, C, tpg): per-tile sum dM * M as emitted (into s12part) by the fgnn_mlp_bwd16 call that produced dM
 * from the MLP whose first input slab is M = the forward product (mlp3);
 *   sum dA (z_a - mean_a) = (T - beta_a S1_a) / a_a,   sum dB (z_b - mean_b) = (T - beta_b S1_b) / a_b */
int fgnn_chan_matmul_bwd16_t(const fgnn_slab16 *ya, const fgnn_slab16 *yb, const void *dm, long long dmgstride, long long ldm,
                             const float *tpart, int tpg, const int *nvalid, int G, int N, int ldr, void *da, void *db,
                             long long ogstride, long long ldo, float *s12a, float *s12b, void *stream);

/* fgnn_chan_matmul_bwd16_t that also writes the dz-coefficient records (G*C*4, the output of fgnn_gn_bwd_coef) of the two
 * operand MLPs from the s12 sums it has just formed -- one tiny launch per block less */
int fgnn_chan_matmul_bwd16_tc(const fgnn_slab16 *ya, const fgnn_slab16 *yb, const void *dm, long long dmgstride, long long ldm,
                              const float *tpart, int tpg, const int *nvalid, int G, int N, int ldr, void *da, void *db,
                              long long ogstride, long long ldo, float *s12a, float *s12b, float *coefa, float *coefb,
                              void *stream);
/* ColumnMaxPooling on a bf16 slab: e (G,C,N) fp32 = max_j of the fp32-normalised values, idx int32 */
int fgnn_colmax_fwd16(const fgnn_slab16 *y, const int *nvalid, int G, int N, int ldr, float *e, int *idx, void *stream);
/* its backward: dy (bf16) [g,c,i,idx] = R(de[g,c,i]); s12 = {sum dy, sum dy*(z-mean)} of the rounded values */
int fgnn_colmax_bwd16(const float *de, const int *idx, const int *nvalid, int G, int C, int N, int ldr, void *dy,
                      long long gstride, long long ldp, const fgnn_slab16 *y, float *s12, void *stream);
/* the same, also writing the dz-coefficient record (G*C*4) of the pooled MLP from its s12 sums */
int fgnn_colmax_bwd16_coef(const float *de, const int *idx, const int *nvalid, int G, int C, int N, int ldr, void *dy,
                           long long gstride, long long ldp, const fgnn_slab16 *y, float *s12, float *coef, void *stream);

/* MlpBlock_Real backward on bf16 slabs: see fgnn_mlp_bwd.  coef (G*32*4) is required (fgnn_gn_bwd_coef*).
 * dx outputs are rounded to bf16; with accumulate the old bf16 value is added in fp32 before rounding. */
typedef struct {
    int G, N, ldr, depth;
    const int *nvalid;
    fgnn_slab16 a, b;
    const void *dy;  long long dgstride, ldd;
    const void *z;   long long zgstride, ldz;
    const float *coef;
    void *dxa; long long dxa_gstride, dxa_ld;
    void *dxb; long long dxb_gstride, dxb_ld;
    int accumulate_a, accumulate_b;
    float *wpart;                            /* out (num_wg, fgnn_mlp_param_count) partial dW/db (fp32) */
    float *s12part;                          /* optional out, needs a.C == 32 and dxa.  Single normalised slab (no slab b):
                                                (G, 32, tpg16, 2) per-tile {sum dxa, sum dxa*(z_a-mean_a)} of the FINAL (rounded)
                                                dxa values.  Raw first slab of a two-slab MLP (mlp3: a = mult): (G, 32, tpg16)
                                                per-tile sum dxa * x_a, the trace term read by fgnn_chan_matmul_bwd16_t */
    const void *packed;                      /* operand image (kind 1) -- required */
    const int *ranges;                       /* optional (ragged): tile bounds from fgnn_ragged_tile_ranges16 */
} fgnn_mlp_bwd16_args;
int fgnn_mlp_bwd16(const fgnn_mlp_bwd16_args *args, void *stream);
/* mlp1 + mlp2 of one block in ONE launch, bf16 (the twin of fgnn_mlp_bwd_pair): m1 / m2 are the argument blocks of the two
 * fgnn_mlp_bwd16 calls it replaces; the input gradient (dxa, accumulate_a, s12part) is given in m2 only and receives
 * R(R(old + dx1) + dx2), bit-identical to the two read-modify-write launches.  One slab of 2 or 32 channels, constant-size batches. */
int fgnn_mlp_bwd16_pair(const fgnn_mlp_bwd16_args *m1, const fgnn_mlp_bwd16_args *m2, void *stream);

/* ---- test-only entry points (never on the product path; tests/ and tools/ call them) ---------------------------------------
 * fgnn_debug_mlp_fwd_masks / fgnn_debug_mlp_fwd_x3_masks: fgnn_mlp_fwd / fgnn_mlp_fwd_x3 once more -- the same tile code, the same
 * outputs, bit for bit -- that ALSO exports the ReLU decisions of the conv chain (models/layers.py:129-130), the input of the
 * decision-pinned gradient test (tests/test_gpu_grad_pinned.py): masks[m] is (G, depth - 1, 32, fgnn_tiles_per_graph(N)) words, bit j
 * of word (g, layer, channel, t) = [hidden pre-activation of pixel 32 t + j > 0] as `relu` sees it (bit pattern > 0).  Depth 3;
 * words of tiles a ragged launch steps over are not written.  The x3 twin exists for the two-MLP launches (masks1 required).
 * fgnn_debug_matmul_variant: 0 selects the workgroup-per-matrix forward product (N <= 64) that fgnn_chan_matmul_fwd_w replaced, 1
 * (default) the wave-per-matrix kernel; process-global; for the bit-identity test of the two (tests/test_gpu_kernels.py). */
int fgnn_debug_mlp_fwd_masks(const fgnn_mlp_fwd_args *args, unsigned *masks0, unsigned *masks1 /* nmlp == 2 */, void *stream);
int fgnn_debug_mlp_fwd_x3_masks(const fgnn_mlp_fwd_args *args, unsigned *masks0, unsigned *masks1, void *stream);
int fgnn_debug_matmul_variant(int wave_per_matrix);

#ifdef __cplusplus
}
#endif
#endif /* FGNN_HIP_H */

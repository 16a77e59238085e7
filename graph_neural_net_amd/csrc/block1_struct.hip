// Block 1 of the 2-FGNN on its STRUCTURED input (bit-packed adjacency; constant-size and ragged batches, N <= 256).
//
// The reference feeds block 1 the tensor representation of a graph (loaders/data_generator.py:118-125): channel 0 = the 0/1
// adjacency W, channel 1 = diag(row sums).  So the input pixel (i, j) takes one of few values -- off the diagonal (w_ij, 0)
// with w in {0, 1}, on it (w_ii, deg_i) -- and so do the outputs of mlp1 / mlp2 (models/blocks_emb.py:16-27: two
// MlpBlock_Real on that input): a table of 2 + 2 (N + 1) rows per model instead of G N^2 pixels through three convs.  With
// the normalised class values u0 / u1 / e_i (mlp1) and v0 / v1 / f_j (mlp2) of a graph,
//     Y1_c = u0 J + p W + diag(q),   p = u1 - u0, q_i = e_i - u0 - p w_ii        (and Y2_c = v0 J + r W + diag(s) likewise)
// and the per-channel product mult_c = Y1_c Y2_c (models/layers.py:161-162) has the closed form
//     mult_c[i][j] = u0 v0 n + u0 r degc_j + u0 s_j + p v0 degr_i + p r (W^2)_ij + p w_ij s_j + v0 q_i + r q_i w_ij + [i = j] q_i s_i
// with (W^2)_ij = popcount(row_i & column_j) on the bit rows: ONE integer product per pixel instead of 32 N x N products per
// graph, formed once per graph (sb_graph_kernel: a 16-bit code plane (W^2)_ij | w_ij << 15 plus per-vertex records).  GraphNorm statistics (models/layers.py:68-80) follow from the class counts.  In the backward direction the gradient
// reaches the parameters of mlp1 / mlp2 only through the class values, so what is needed of dY1_c = dM_c Y2_c^T and dY2_c =
// Y1_c^T dM_c are their CLASS SUMS, which reduce to row / column / diagonal / masked sums of dM_c and <dM_c, W^2>: one pass
// over d(mult).  The GraphNorm backward and the conv / ReLU chain then run per class (the ReLU masks are class constants, the
// chain is linear in the summed gradient), a few dozen 32-vectors per graph.
//
// Replaces, for block 1 only: fgnn_mlp_fwd (mlp1 + mlp2) + fgnn_chan_matmul_fwd in the forward direction and
// fgnn_chan_matmul_bwd + fgnn_mlp_bwd_pair in the backward direction.  Same function, another evaluation order: results agree
// with the generic kernels to fp32 rounding (tests/test_gpu_struct.py), not bit for bit.
// Ragged batches: graph g has nv = nvalid[g] vertices inside the N x N padded planes; everything below runs on the valid corner
// (rows / bits >= nv are cleared on load, class counts and the GraphNorm n use nv), mult is written as 0 outside it.
// The kernels are instantiated for bit rows of one (N <= 64), two (N <= 128) and four (N <= 256) 64-bit words, and for fp32 slabs
// (row pitch N) as well as the bf16 slabs of the 16-bit engine (row pitch ldr, values rounded to nearest even on store).
#include "fgnn_common.h"
#include "fgnn_norm.h"
#include "fgnn_bf16.h"
#include "fgnn_pack.h"

namespace {

constexpr int SB_CG = 8;                 // channel groups of the forward kernel (4 channels each)
constexpr int SB_CPG = FGNN_H / SB_CG;   // channels per group
constexpr int SB_TAB = 4 * FGNN_H;       // floats per table row: h1 | h2 | z (fp32, for the statistics) | z as stored (bf16 scheme: R(z))
constexpr int SB_NMAX = 256;
constexpr int SB_FW = 8;                 // waves per workgroup of the forward kernel
constexpr int SB_KC = 32;                // class instances per round of the class-algebra kernel

// class of a pixel's input value: 0 = off-diagonal, w = 0; 1 = off-diagonal, w = 1; 2 + 2 deg + w_ii = diagonal
DEVI int sb_classes(int N) { return 2 + 2 * (N + 1); }


#ifdef SB_STAMPS
__device__ unsigned long long *g_sb_stamps = nullptr;      // [kernel 0..3][workgroup < 2048][8] s_memtime ticks (tools/gpu_struct_stamps.py)
#define SB_STAMP(kern, wg, i) if (threadIdx.x == 0 && g_sb_stamps && (wg) < 2048) g_sb_stamps[((kern) * 2048 + (wg)) * 8 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define SB_STAMP(kern, wg, i)
#endif

// ---- K0: the class tables of one model, graph independent: tab[m][class][h1 | h2 | z | z as stored][32] ------------------
struct TabArgs {
    const float *W[2][3];
    const float *b[2][3];
};
// bf: the 16-bit scheme of engine16 (oracle/fgnn_oracle_bf16.py): matrix-core operands R(W), R(relu(.)) and the stored R(z)
DEVI float sb_r(float v, int bf) { return bf ? bf_lo(cvt_pk(v, 0.f)) : v; }
// one wave computes one table row (lanes 0..31 = the 32 channels; the hidden vectors travel by cross-lane reads: no LDS, no barrier)
DEVI void sb_table_row(const TabArgs &A, const int N, float *tab, const int bf, const int cls, const int m, const int lane) {
    const int o = lane & 31;
    float x0, x1;
    if (cls < 2) {
        x0 = (float)cls;
        x1 = 0.f;
    } else {
        x0 = (float)((cls - 2) & 1);
        x1 = (float)((cls - 2) >> 1);
    }
    float *out = tab + ((long long)m * sb_classes(N) + cls) * SB_TAB;
    const float *W0 = A.W[m][0];
    float a = A.b[m][0][o];
    a = fmaf(sb_r(W0[o * 2 + 0], bf), x0, a);
    a = fmaf(sb_r(W0[o * 2 + 1], bf), x1, a);
    const float h1 = sb_r(fmaxf(a, 0.f), bf);
    const float *W1 = A.W[m][1] + o * FGNN_H;
    a = A.b[m][1][o];
#pragma unroll
    for (int c = 0; c < FGNN_H; ++c) a = fmaf(sb_r(W1[c], bf), __shfl(h1, c), a);
    const float h2 = sb_r(fmaxf(a, 0.f), bf);
    const float *W2 = A.W[m][2] + o * FGNN_H;
    a = A.b[m][2][o];
#pragma unroll
    for (int c = 0; c < FGNN_H; ++c) a = fmaf(sb_r(W2[c], bf), __shfl(h2, c), a);
    if (lane < FGNN_H) {
        out[o] = h1;
        out[FGNN_H + o] = h2;
        out[2 * FGNN_H + o] = a;
        out[3 * FGNN_H + o] = sb_r(a, bf);
    }
}
__global__ __launch_bounds__(256) void sb_tables_kernel(const TabArgs A, const int N, float *tab, const int bf) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), NC = sb_classes(N);
    if (row < 2 * NC) sb_table_row(A, N, tab, bf, row % NC, row / NC, threadIdx.x & 63);
}

// row t of graph g restricted to the valid corner (bits >= nv cleared, rows >= nv empty) as 2 NWD dwords in LDS
template <int NWD>
DEVI void sb_bits_rows(unsigned (*rows)[2 * NWD], const unsigned *bits, int g, int t, int N, int nv) {
    if (t < 64 * NWD) {
        const int words = (N + 31) / 32;
        const unsigned *r = bits + ((long long)g * N + (t < N ? t : 0)) * words;
#pragma unroll
        for (int w = 0; w < 2 * NWD; ++w) {
            unsigned v = (t < nv && w < words) ? r[w] : 0u;
            const int left = nv - 32 * w;                          // valid bits in this word
            v = left >= 32 ? v : (left <= 0 ? 0u : (v & ((1u << left) - 1u)));
            rows[t][w] = v;
        }
    }
}

// The normalised class values of channel c of one model in one graph: Y = u0 J + p W + diag(q)
template <int NWD>
struct ClassVals {
    float u0, p, q[NWD];     // Y = u0 J + p W + diag(q): q[k] belongs to vertex lane + 64 k
    float4 rec;              // GraphNorm record of (g, c)
};
// The per-step kernels touch the bit matrix ONCE per graph: sb_graph_kernel turns it into a 16-bit code plane, code[g][i][j] =
// (W^2)_ij | w_ij << 15, plus per-vertex {row sum, column sum, w_ii, class} records; the forward kernel, the class-sum kernel
// and the class algebra read those (L2-resident: N^2 x 2 bytes per graph against the 32 N^2 values of a slab).  (A first form
// that re-derived the popcounts from LDS bit rows in every (graph, channel group) workgroup cost 105 / 100 / 51 us at N = 200,
// 16 graphs, against 19.5 / 21.3 / 8.8 us now: profiles/archive/r04_cfg4_struct_timelines.txt, r04_final_cfg4_graph_timeline.txt.)
// Workspace of one step (fwd fills, bwd reads):
struct WsLayout {
    long long csum, coef, vinfo, gones, code, total;      // offsets in floats
    int cp, CS;
};
static WsLayout sb_ws_layout(int G, int N) {
    const int MAXN = N <= 64 ? 64 : (N <= 128 ? 128 : 256);
    WsLayout L;
    L.CS = 2 + MAXN;
    L.cp = (N + 7) & ~7;
    long long o = 0;
    L.csum = o;
    o += 2LL * G * FGNN_H * L.CS;
    o = (o + 3) & ~3LL;
    L.coef = o;
    o += 2LL * G * FGNN_H * 4;
    L.vinfo = o;
    o += 4LL * G * N;
    L.gones = o;
    o += (G + 3) & ~3;
    L.code = o;
    o += ((long long)G * N * L.cp / 2 + 3) & ~3LL;
    L.total = o;
    return L;
}

// ---- KG: code plane + vertex records (+ the input slabs the other kernels of block 1 read) -------------------------------------
// grid ntab + G ny: the first ntab workgroups build the class tables of the step (four rows each: the graph-independent launch of
// fgnn_block1_struct_tables folded in); workgroup ntab + g ny + y loads the bit rows of graph g and owns the band of SB_CW columns
// y SB_CW ..: it transposes ONLY those columns (a workgroup that transposed the whole matrix spent 4.6 of its 9.5 us there at
// N = 200), writes the vertex records of the vertices in the band and, thread <-> row, the band's codes of every row.
constexpr int SB_CW = 16;                // columns per workgroup of the per-graph kernel
// (the kernel-argument segment is limited to 4 KB: the pack jobs travel by value next to the table arguments and ~ 20 scalars / pointers)
static_assert(sizeof(PackJobs) + sizeof(TabArgs) + 256 <= 4096, "sb_graph_kernel: kernel arguments exceed 4 KB -- pass the pack jobs through a device buffer");
template <int NWD>
__global__ __launch_bounds__(256) void sb_graph_kernel(const unsigned *bits, const int *nvalid, const int N, const int cp, unsigned *code,
                                                       float4 *vinfo, float *gones, float *xdeg, void *x16, const long long ldp16,
                                                       const int pitch16, const int ny, const int ntab, const TabArgs TA, float *tab,
                                                       const int bf, const int npack, const PackJobs PJ) {
    constexpr int MAXN = 64 * NWD, RW = 2 * NWD;                   // dwords per bit row
    __shared__ unsigned rows[MAXN][RW];
    __shared__ unsigned colw[SB_CW][RW];
    __shared__ float red[4];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < npack) {
        // the step's operand packing (fgnn_pack_operands / fgnn_pack16_operands) as the leading workgroups of this launch: it reads the weights only, like the
        // tables below, and every consumer of an image is a later launch
        if (bf) pack16_job_body(PJ.job[blockIdx.x / PACK16_BLOCKS_PER_JOB], blockIdx.x % PACK16_BLOCKS_PER_JOB, PACK16_BLOCKS_PER_JOB, tid);
        else pack_job_body(PJ.job[blockIdx.x / PACK_BLOCKS_PER_JOB], blockIdx.x % PACK_BLOCKS_PER_JOB, PACK_BLOCKS_PER_JOB, tid);
        return;
    }
    const int bid = (int)blockIdx.x - npack;
    if (bid < ntab) {
        const int row = bid * 4 + (tid >> 6), NC = sb_classes(N);
        if (row < 2 * NC) sb_table_row(TA, N, tab, bf, row % NC, row / NC, tid & 63);
        return;
    }
    const int g = (bid - ntab) / ny, y = (bid - ntab) - g * ny;
    const int nv = nvalid_of(nvalid, g, N);
    const int c0 = y * SB_CW;                                      // first column of the band
    SB_STAMP(0, bid - ntab, 0)
    sb_bits_rows<NWD>(rows, bits, g, tid, N, nv);
    __syncthreads();
    SB_STAMP(0, bid - ntab, 1)
    if (tid < SB_CW * RW) {
        // dword h of column c0 + jl: bit r = w[32 h + r][c0 + jl]; every shift a compile-time constant
        const int jl = tid % SB_CW, h = tid / SB_CW, j = c0 + jl;
        unsigned acc = 0u;
        if (j < nv && 32 * h < nv) {
            const int dw = j >> 5, sh = j & 31;
#pragma unroll
            for (int k8 = 0; k8 < 32; k8 += 8) {
                unsigned r[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) r[k] = rows[32 * h + k8 + k][dw];
#pragma unroll
                for (int k = 0; k < 8; ++k) acc |= ((r[k] >> sh) & 1u) << (k8 + k);
            }
        }
        colw[jl][h] = acc;
    }
    // row sums and self loops (thread <-> vertex)
    int dr = 0, ws = 0;
    if (tid < MAXN) {
#pragma unroll
        for (int w = 0; w < RW; ++w) dr += __popc(rows[tid][w]);
        ws = (rows[tid][tid >> 5] >> (tid & 31)) & 1u;
    }
    if (y == 0) {
        float on1 = tid < nv ? (float)(dr - ws) : 0.f;
        on1 = wave_sum(on1);
        if ((tid & 63) == 0) red[tid >> 6] = on1;
    }
    __syncthreads();
    SB_STAMP(0, bid - ntab, 2)
    if (y == 0 && tid == 0) gones[g] = (red[0] + red[1]) + (red[2] + red[3]);   // off-diagonal ones of the valid corner
    if (tid >= c0 && tid < c0 + SB_CW && tid < N) {                // the records of the band's vertices
        int dc = 0;
#pragma unroll
        for (int w = 0; w < RW; ++w) dc += __popc(colw[tid - c0][w]);
        const bool on = tid < nv;
        vinfo[(long long)g * N + tid] = make_float4(on ? (float)dr : 0.f, on ? (float)dc : 0.f, __int_as_float(on ? ws : 0),
                                                    __int_as_float(on ? 2 + 2 * dr + ws : 0));
        if (xdeg) xdeg[(long long)g * N + tid] = on ? (float)dr : 0.f;            // what fgnn_adjacency_degree would write
    }
    SB_STAMP(0, bid - ntab, 3)
    // thread <-> row i: its SB_CW codes (W^2)_ij | w_ij << 15, j in the band (rows / bits outside the valid corner are empty)
    unsigned short *code16 = reinterpret_cast<unsigned short *>(code) + (long long)g * N * cp;
    unsigned short *xa = reinterpret_cast<unsigned short *>(x16) + (long long)g * 2 * ldp16, *xb = xa + ldp16;
    if (tid < N) {
        const int i = tid;
        unsigned rw[RW];
#pragma unroll
        for (int w = 0; w < RW; ++w) rw[w] = rows[i][w];
        const unsigned dg = (i < nv) ? (cvt_pk((float)dr, 0.f) & 0xffffu) : 0u;
#pragma unroll
        for (int q = 0; q < SB_CW / 8; ++q) {                      // eight columns = one 16-byte store per plane
            const int jq = c0 + 8 * q;
            const unsigned wbyte = jq < MAXN ? (rows[i][jq >> 5] >> (jq & 31)) & 0xffu : 0u;      // w_ij of the eight columns (jq is a multiple of 8)
            unsigned cd[8], wb[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                int w2 = 0;
#pragma unroll
                for (int w = 0; w < RW; ++w) w2 += __popc(rw[w] & colw[8 * q + t][w]);
                wb[t] = (wbyte >> t) & 1u;
                cd[t] = (unsigned)w2 | (wb[t] << 15);              // 0 outside the valid corner: empty row or empty column
            }
            if (jq < cp)
                *reinterpret_cast<uint4 *>(code16 + (long long)i * cp + jq) =
                    make_uint4(cd[0] | (cd[1] << 16), cd[2] | (cd[3] << 16), cd[4] | (cd[5] << 16), cd[6] | (cd[7] << 16));
            if (x16 && jq < pitch16) {
                unsigned wv[8], dv[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    wv[t] = wb[t] ? 0x3F80u : 0u;                  // bf16 1.0
                    dv[t] = (jq + t == i) ? dg : 0u;
                }
                *reinterpret_cast<uint4 *>(xa + (long long)i * pitch16 + jq) =
                    make_uint4(wv[0] | (wv[1] << 16), wv[2] | (wv[3] << 16), wv[4] | (wv[5] << 16), wv[6] | (wv[7] << 16));
                *reinterpret_cast<uint4 *>(xb + (long long)i * pitch16 + jq) =
                    make_uint4(dv[0] | (dv[1] << 16), dv[2] | (dv[3] << 16), dv[4] | (dv[5] << 16), dv[6] | (dv[7] << 16));
            }
        }
    }
    SB_STAMP(0, bid - ntab, 4)
    if (x16 && y == ny - 1) {
        // tail of the channel stride of the 2-channel input slab of the 16-bit engine (mlp3 reads the slab as its skip connection)
        unsigned *x0 = reinterpret_cast<unsigned *>(xa), *x1 = reinterpret_cast<unsigned *>(xb);
        for (int e = N * pitch16 / 2 + tid; e < (int)(ldp16 / 2); e += 256) x0[e] = x1[e] = 0u;
    }
}

// class values of channel c of one model from the vertex records.  The 64 lanes of a wave call this together; lane l owns the
// vertices l + 64 k.  Statistics from the class counts (two-pass: mean, then squared deviations), record as fgnn_norm.h,
// y = (z - mean) a + beta as every consumer of a slab evaluates it
template <int NWD>
DEVI ClassVals<NWD> sb_class_values(const float *tab_m, const float4 (&vi)[NWD], float ones, int nv, int c, float gnw, float beta, float eps, int lane,
                                     int bf) {
    const float z0 = tab_m[0 * SB_TAB + 2 * FGNN_H + c], z1 = tab_m[1 * SB_TAB + 2 * FGNN_H + c];
    const float y0 = tab_m[0 * SB_TAB + 3 * FGNN_H + c], y1 = tab_m[1 * SB_TAB + 3 * FGNN_H + c];          // as stored
    float zd[NWD], yd[NWD], szd = 0.f;
#pragma unroll
    for (int k = 0; k < NWD; ++k) {
        const bool on = lane + 64 * k < nv;
        const int cl = on ? __float_as_int(vi[k].w) : 0;
        zd[k] = on ? tab_m[cl * SB_TAB + 2 * FGNN_H + c] : 0.f;
        yd[k] = on ? tab_m[cl * SB_TAB + 3 * FGNN_H + c] : 0.f;
        szd += zd[k];
    }
    const float fN = (float)nv, m = fN * fN;
    const float n1 = ones, n0 = m - fN - ones;
    const float mean = m > 0.f ? (n0 * z0 + n1 * z1 + wave_sum(szd)) / m : 0.f;
    const float d0 = z0 - mean, d1 = z1 - mean;
    float sdd = 0.f;
#pragma unroll
    for (int k = 0; k < NWD; ++k) {
        const float dd = lane + 64 * k < nv ? zd[k] - mean : 0.f;
        sdd += dd * dd;
    }
    const float m2 = n0 * d0 * d0 + n1 * d1 * d1 + wave_sum(sdd);
    ClassVals<NWD> v;
    v.rec = nrm_record(mean, m2, m, fN, gnw, eps);
    const float u0 = sb_r((y0 - mean) * v.rec.y + beta, bf), u1 = sb_r((y1 - mean) * v.rec.y + beta, bf);
    v.u0 = u0;
    v.p = u1 - u0;
#pragma unroll
    for (int k = 0; k < NWD; ++k) {
        const float e = sb_r((yd[k] - mean) * v.rec.y + beta, bf);
        v.q[k] = lane + 64 * k < nv ? e - u0 - v.p * (float)__float_as_int(vi[k].z) : 0.f;
    }
    return v;
}

// ---- K1: GraphNorm records of mlp1 / mlp2 + mult = Y1 Y2 in closed form, row by row ------------------------------
// grid (G, SB_CG, parts), SB_FW = 8 waves: a workgroup writes SB_CPG channels -- four waves form their class values (once per
// workgroup: with 4-wave workgroups and twice the parts this prologue was 46 % of the kernel's VALU instructions at N = 200), then
// wave w takes the rows w + 8 part, + 8 parts, ...; a
// lane owns fixed columns and keeps their per-channel terms in registers, the row terms are wave-uniform:
//     mult_c[i][j] = [k0 + k2 degr_i + v0 q_i] + [k1 degc_j + u0 s_j] + k3 (W^2)_ij + w_ij (p s_j + r q_i) + [i = j] q_i s_i
// fp32 slabs: lane <-> columns lane + 64 t; bf16 slabs: lane <-> column pairs 2 lane + 128 t (one dword per store)
template <int NWD, bool BF>
__global__ __launch_bounds__(64 * SB_FW) void sb_fwd_kernel(const unsigned short *code, const float4 *vinfo, const float *gones, const int *nvalid,
                                                      const int N, const int cp, const float *tab, const float *gnw1, const float *gnb1,
                                                      const float *gnw2, const float *gnb2, const float eps, float *nrm1, float *nrm2, void *mult,
                                                      const long long gstride, const long long ldp, const int pitch) {
    constexpr int MAXN = 64 * NWD;
    constexpr int NT = BF ? (NWD >= 2 ? NWD / 2 : 1) : NWD;      // column slots per lane
    constexpr int CW = BF ? 2 : 1;                               // columns per slot
    __shared__ float sc[SB_CPG][4];                              // u0, p, v0, r per channel
    __shared__ float Q[SB_CPG][MAXN], S[SB_CPG][MAXN];
    __shared__ float DR[MAXN], DC[MAXN];
    const int g = blockIdx.x, cg = blockIdx.y, part = blockIdx.z, nparts = gridDim.z, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nv = nvalid_of(nvalid, g, N);
    SB_STAMP(1, (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, 0)
    const int NC = sb_classes(N);
    if (wv < SB_CPG) {   // wave wv < 4 owns channel cg * SB_CPG + wv of both models (the other waves join for the rows)
        float4 vi[NWD];
#pragma unroll
        for (int k = 0; k < NWD; ++k) vi[k] = lane + 64 * k < nv ? vinfo[(long long)g * N + lane + 64 * k] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float ones = gones[g];
        const int c = cg * SB_CPG + wv;
        const ClassVals<NWD> a = sb_class_values(tab, vi, ones, nv, c, gnw1[c], gnb1[c], eps, lane, BF ? 1 : 0);
        const ClassVals<NWD> b = sb_class_values(tab + (long long)NC * SB_TAB, vi, ones, nv, c, gnw2[c], gnb2[c], eps, lane, BF ? 1 : 0);
        if (lane == 0 && part == 0) {
            reinterpret_cast<float4 *>(nrm1)[(long long)g * FGNN_H + c] = a.rec;
            reinterpret_cast<float4 *>(nrm2)[(long long)g * FGNN_H + c] = b.rec;
        }
        if (lane == 0) {
            sc[wv][0] = a.u0;
            sc[wv][1] = a.p;
            sc[wv][2] = b.u0;
            sc[wv][3] = b.p;
        }
#pragma unroll
        for (int k = 0; k < NWD; ++k) {
            Q[wv][lane + 64 * k] = a.q[k];
            S[wv][lane + 64 * k] = b.q[k];
            if (wv == 0) {
                DR[lane + 64 * k] = vi[k].x;
                DC[lane + 64 * k] = vi[k].y;
            }
        }
    }
    SB_STAMP(1, (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, 1)
    __syncthreads();
    const float fN = (float)nv;
    float k0[SB_CPG], k2[SB_CPG], k3[SB_CPG], v0[SB_CPG], rr[SB_CPG], pk[SB_CPG];
    float colc[SB_CPG][NT * CW], sv[SB_CPG][NT * CW];
    int col[NT * CW];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int h = 0; h < CW; ++h) col[t * CW + h] = BF ? 2 * lane + 128 * t + h : lane + 64 * t;
#pragma unroll
    for (int k = 0; k < SB_CPG; ++k) {
        const float u0 = sc[k][0], pp = sc[k][1];
        pk[k] = pp;
        v0[k] = sc[k][2];
        rr[k] = sc[k][3];
        k0[k] = u0 * v0[k] * fN;
        k2[k] = pp * v0[k];
        k3[k] = pp * rr[k];
        const float k1 = u0 * rr[k];
#pragma unroll
        for (int e = 0; e < NT * CW; ++e) {
            const int j = col[e] < MAXN ? col[e] : 0;
            sv[k][e] = S[k][j];
            colc[k][e] = k1 * DC[j] + u0 * sv[k][e];
        }
    }
    const unsigned short *cg16 = code + (long long)g * N * cp;
    float *out32 = reinterpret_cast<float *>(mult) + (BF ? 0 : (long long)g * gstride + (long long)cg * SB_CPG * ldp);
    unsigned *out16 = reinterpret_cast<unsigned *>(reinterpret_cast<unsigned short *>(mult) + (BF ? (long long)g * gstride + (long long)cg * SB_CPG * ldp : 0));
    SB_STAMP(1, (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, 2)
    constexpr int RU = 2;                                        // rows in flight per wave
    for (int i0 = part * SB_FW + wv; i0 < N; i0 += SB_FW * nparts * RU) {
        unsigned cd[RU][NT];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int i = i0 + SB_FW * nparts * u;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int j = col[t * CW];
                if constexpr (BF) cd[u][t] = (i < nv && j < cp) ? *reinterpret_cast<const unsigned *>(cg16 + (long long)i * cp + j) : 0u;
                else cd[u][t] = (i < nv && j < nv) ? (unsigned)cg16[(long long)i * cp + j] : 0u;
            }
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int i = i0 + SB_FW * nparts * u;
            if (i >= N) break;
            const bool rowon = i < nv;
            const int ii = rowon ? i : 0;
            const float dr = DR[ii];
            float rowc[SB_CPG], rq[SB_CPG], qk[SB_CPG];
#pragma unroll
            for (int k = 0; k < SB_CPG; ++k) {
                qk[k] = Q[k][ii];
                rowc[k] = k0[k] + k2[k] * dr + v0[k] * qk[k];
                rq[k] = rr[k] * qk[k];
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float val[SB_CPG][CW];
#pragma unroll
                for (int h = 0; h < CW; ++h) {
                    const int e = t * CW + h, j = col[e];
                    const unsigned c16 = (cd[u][t] >> (16 * h)) & 0xffffu;
                    const float w2f = (float)(c16 & 0x7fffu);
                    const bool w = c16 >> 15, on = rowon && j < nv;
#pragma unroll
                    for (int k = 0; k < SB_CPG; ++k) {
                        float v = rowc[k] + colc[k][e];
                        v = fmaf(k3[k], w2f, v);
                        if (w) v += fmaf(pk[k], sv[k][e], rq[k]);
                        if (j == i) v = fmaf(qk[k], sv[k][e], v);
                        val[k][h] = on ? v : 0.f;
                    }
                }
                const int j = col[t * CW];
                if constexpr (BF) {
                    if (j < pitch)
#pragma unroll
                        for (int k = 0; k < SB_CPG; ++k) out16[((long long)k * ldp + (long long)i * pitch + j) / 2] = cvt_pk(val[k][0], val[k][1]);
                } else {
                    if (j < N)
#pragma unroll
                        for (int k = 0; k < SB_CPG; ++k) out32[(long long)k * ldp + (long long)i * N + j] = val[k][0];
                }
            }
        }
    }
    SB_STAMP(1, (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, 3)
    if constexpr (BF) {
        if (part == 0)                                           // tail of the channel stride
            for (int e = N * pitch / 2 + tid; e < (int)(ldp / 2); e += 64 * SB_FW)
#pragma unroll
                for (int k = 0; k < SB_CPG; ++k) out16[((long long)k * ldp) / 2 + e] = 0u;
    }
}

// ---- K2: class sums of dY1 = dM Y2^T and dY2 = Y1^T dM from one pass over dM, + the GraphNorm-backward sums ------
// One workgroup of 4 (N <= 128) or 8 waves per (g, c) plane (wave w: rows w, w + waves, ...; lane: NWD consecutive columns, one load per row
// and operand); the masks and (W^2)_ij come from the code plane.  The last
// wave also forms S1 = sum dY, S2 = sum dY (z - mean) of both models (s12, for fgnn_grad_finalize) and the coefficients of
// dz_p = ca dy_p + cb (z_p - mean) + cc   the class algebra applies: coef[m][g][c] = {ca, cb, cc, mean}
// NWD consecutive floats at a 4-byte aligned address as one load
template <int NWD>
struct __attribute__((packed, aligned(4))) SbFloats {
    float v[NWD];
};
// Sums of NV (a power of two <= 16) per-lane values over the 64 lanes in ONE butterfly: at each of the first log2(NV) steps a lane
// hands half of its values to its partner and keeps the other half (partner across lane bit 5, 4, 3, 2 in that order), so the
// number of live values halves while the partner distance shrinks; the remaining steps reduce the single value left.  Lane L ends
// with the total of value e(L) = bits (b5 b4 b3 b2)[first log2(NV)] of L read as a number: 2 + 1 + 1/2 + ... exchanges per value
// instead of six (a per-row wave_sum pair was 28 of the 47 VALU instructions a row of the class-sum kernel cost at N = 50).
// Fixed tree: bit-reproducible.
template <int STEP>
DEVI float ws_xchg(float x) {           // x of the partner lane of butterfly step STEP (partners differ in lane bit 5 - STEP)
    if constexpr (STEP == 0) return __shfl_xor(x, 32);
    else if constexpr (STEP == 1) return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), 0x401F));
    else if constexpr (STEP == 2) return dpp_mov<0x140>(x);       // row_mirror: partner 15 - l inside a row of 16
    else if constexpr (STEP == 3) return dpp_mov<0x141>(x);       // row_half_mirror: partner 7 - l inside 8
    else if constexpr (STEP == 4) return dpp_mov<0x4E>(x);        // quad_perm [2,3,0,1]
    else return dpp_mov<0xB1>(x);                                 // quad_perm [1,0,3,2]
}
template <int NV, int STEP>
DEVI void ws_step(float (&v)[NV], int lane) {
    constexpr int LIVE = (NV >> STEP) > 1 ? (NV >> STEP) : 1;     // values a lane holds before this step (compile-time: no dynamic indexing)
    if constexpr (LIVE > 1) {
        const bool up = (lane >> (5 - STEP)) & 1;                 // this lane keeps the upper half
#pragma unroll
        for (int k = 0; k < LIVE / 2; ++k) {
            const float send = up ? v[k] : v[k + LIVE / 2], keep = up ? v[k + LIVE / 2] : v[k];
            v[k] = keep + ws_xchg<STEP>(send);
        }
    } else {
        v[0] += ws_xchg<STEP>(v[0]);
    }
}
template <int NV>
DEVI float wave_sum_multi(float (&v)[NV], int lane) {
    static_assert(NV == 2 || NV == 4 || NV == 8 || NV == 16, "power of two");
    ws_step<NV, 0>(v, lane);
    ws_step<NV, 1>(v, lane);
    ws_step<NV, 2>(v, lane);
    ws_step<NV, 3>(v, lane);
    ws_step<NV, 4>(v, lane);
    ws_step<NV, 5>(v, lane);
    return v[0];
}
// index of the value lane L holds after wave_sum_multi<NV> (valid in every lane; lanes that differ only in the low bits hold copies)
template <int NV>
DEVI int wave_sum_multi_index(int lane) {
    constexpr int LG = NV == 2 ? 1 : (NV == 4 ? 2 : (NV == 8 ? 3 : 4));
    return (lane >> (6 - LG)) & (NV - 1);
}

// (NWD + 1) / 2 dwords holding NWD consecutive 16-bit values
template <int NWD>
DEVI void sb_load16_packed(const unsigned short *p, unsigned (&o)[(NWD + 1) / 2]) {
    if constexpr (NWD == 4) {
        const uint2 v = *reinterpret_cast<const uint2 *>(p);
        o[0] = v.x;
        o[1] = v.y;
    } else if constexpr (NWD == 2) {
        o[0] = *reinterpret_cast<const unsigned *>(p);
    } else {
        o[0] = *p;
    }
}
template <int NWD>
constexpr int sb_reduce_waves() { return NWD == 4 ? 8 : 4; }      // (8 waves measured slower for N <= 128)
template <int NWD, bool BF>
__global__ __launch_bounds__(64 * sb_reduce_waves<NWD>()) void sb_bwd_reduce_kernel(
    const unsigned short *code, const float4 *vinfo, const int *nvalid, const int G, const int N, const int cp, const float *tab, const float *nrm1,
    const float *nrm2, const float *gnb1, const float *gnb2, const void *dm, const long long gstride, const long long ldp, const int pitch, float *csum,
    float *s12_1, float *s12_2, float4 *coef) {
    constexpr int MAXN = 64 * NWD, CS = 2 + MAXN, NW = sb_reduce_waves<NWD>();
    __shared__ float colp[NW][3][MAXN];                 // per wave: C, Qm, U per column
    __shared__ float Rs[MAXN], Ps[MAXN], Dg[MAXN];      // per row: sum, W-masked sum, diagonal entry
    const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;      // (wave-uniform: scalar row branches)
    const int pl = blockIdx.x, g = pl / FGNN_H, c = pl - g * FGNN_H;
    const int nv = nvalid_of(nvalid, g, N);
    const float4 ra = reinterpret_cast<const float4 *>(nrm1)[(long long)g * FGNN_H + c];
    const float4 rb = reinterpret_cast<const float4 *>(nrm2)[(long long)g * FGNN_H + c];
    const float ba = gnb1[c], bb = gnb2[c];
    const float *src = reinterpret_cast<const float *>(dm) + (BF ? 0 : (long long)g * gstride + (long long)c * ldp);
    const unsigned short *src16 = reinterpret_cast<const unsigned short *>(dm) + (BF ? (long long)g * gstride + (long long)c * ldp : 0);
    const unsigned short *cg16 = code + (long long)g * N * cp;
    SB_STAMP(2, blockIdx.x, 0)
    // lane <-> the NWD consecutive columns NWD lane + t: one load per row and operand
    const int j0 = NWD * lane;
    const bool lane_on = j0 < nv;
    float C[NWD], Qm[NWD], U[NWD];
#pragma unroll
    for (int k = 0; k < NWD; ++k) C[k] = Qm[k] = U[k] = 0.f;
    constexpr int RU = NWD == 4 ? 8 : 4;                 // rows of this wave per memory round trip
    constexpr int XW = BF ? (NWD + 1) / 2 : NWD;         // dwords of a lane's row segment of d(mult)
    for (int i0 = wv; i0 < nv; i0 += NW * RU) {
        unsigned xr[RU][XW], cr[RU][(NWD + 1) / 2];      // kept packed until used
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int i = i0 + NW * u;
#pragma unroll
            for (int k = 0; k < XW; ++k) xr[u][k] = 0u;
#pragma unroll
            for (int k = 0; k < (NWD + 1) / 2; ++k) cr[u][k] = 0u;
            if (i < nv && lane_on) {
                sb_load16_packed<NWD>(cg16 + (long long)i * cp + j0, cr[u]);     // cp is a multiple of 8: the vector stays inside the row
                if constexpr (BF) {
                    sb_load16_packed<NWD>(src16 + (long long)i * pitch + j0, xr[u]);      // pitch is a multiple of 8
                } else if (i < N - 1) {                  // (wave-uniform) a lane's vector may run into the next row of the plane, not past it
                    const SbFloats<NWD> f = *reinterpret_cast<const SbFloats<NWD> *>(src + (long long)i * pitch + j0);
#pragma unroll
                    for (int k = 0; k < NWD; ++k) xr[u][k] = __float_as_uint(f.v[k]);
                } else {
#pragma unroll
                    for (int k = 0; k < NWD; ++k) xr[u][k] = j0 + k < N ? __float_as_uint(src[(long long)i * pitch + j0 + k]) : 0u;
                }
            }
        }
        float red[2 * RU];                                // {row sum, W-masked row sum} of the RU rows: one butterfly for all of them
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int i = i0 + NW * u;
            float rs = 0.f, rm = 0.f;
            if (i < nv) {
#pragma unroll
                for (int k = 0; k < NWD; ++k) {
                    const bool on = j0 + k < nv;
                    const unsigned cd = on ? (cr[u][k >> 1] >> (16 * (k & 1))) & 0xffffu : 0u;
                    float x;
                    if constexpr (BF) x = __uint_as_float((xr[u][k >> 1] >> (16 * (k & 1))) << 16);
                    else x = __uint_as_float(xr[u][k]);
                    x = on ? x : 0.f;
                    C[k] += x;
                    rs += x;
                    if (cd >> 15) {
                        Qm[k] += x;
                        rm += x;
                    }
                    U[k] = fmaf(x, (float)(cd & 0x7fffu), U[k]);
                    if (j0 + k == i) Dg[i] = x;
                }
            }
            if constexpr (NWD == 1) {                     // (one column per lane: the per-row sums measured faster than the batched butterfly)
                rs = wave_sum(rs);
                rm = wave_sum(rm);
                if (lane == 0 && i < nv) {
                    Rs[i] = rs;
                    Ps[i] = rm;
                }
            }
            red[2 * u] = rs;
            red[2 * u + 1] = rm;
        }
        if constexpr (NWD > 1) {
            const float tot = wave_sum_multi<2 * RU>(red, lane);
            const int e = wave_sum_multi_index<2 * RU>(lane), i = i0 + NW * (e >> 1);
            constexpr int LOW = 64 / (2 * RU) - 1;        // lanes with these bits clear are the writers
            if ((lane & LOW) == 0 && i < nv) ((e & 1) ? Ps : Rs)[i] = tot;
        }
    }
#pragma unroll
    for (int k = 0; k < NWD; ++k) {
        colp[wv][0][j0 + k] = C[k];
        colp[wv][1][j0 + k] = Qm[k];
        colp[wv][2][j0 + k] = U[k];
    }
    SB_STAMP(2, blockIdx.x, 1)
    __syncthreads();
    // wave k < NWD: lane l owns vertex l + 64 k; the partial sums of the NWD waves meet in LDS (fixed order), wave 0 finishes
    __shared__ float fin[NWD][16];
    const int NC = sb_classes(N);
    const float *ta = tab, *tb = tab + (long long)NC * SB_TAB;
    constexpr int bf = BF ? 1 : 0;
    const float za0 = ta[3 * FGNN_H + c] - ra.x, za1 = ta[SB_TAB + 3 * FGNN_H + c] - ra.x;           // z (as stored) - mean
    const float zb0 = tb[3 * FGNN_H + c] - rb.x, zb1 = tb[SB_TAB + 3 * FGNN_H + c] - rb.x;
    const float u0 = sb_r(za0 * ra.y + ba, bf), u1 = sb_r(za1 * ra.y + ba, bf);
    const float v0 = sb_r(zb0 * rb.y + bb, bf), v1 = sb_r(zb1 * rb.y + bb, bf);
    const float p = u1 - u0, r = v1 - v0;
    const float fN = (float)nv;
    float *o1 = csum + ((long long)g * FGNN_H + c) * CS;
    float *o2 = csum + (((long long)G + g) * FGNN_H + c) * CS;
    float t[14];
    if (wv < NWD) {
        const int vx = lane + 64 * wv;
        const bool on = vx < nv;
        const float4 vi = on ? vinfo[(long long)g * N + vx] : make_float4(0.f, 0.f, 0.f, 0.f);
        const int cl = __float_as_int(vi.w);
        const float wii = (float)__float_as_int(vi.z);
        const float za = on ? ta[cl * SB_TAB + 3 * FGNN_H + c] - ra.x : 0.f, zb = on ? tb[cl * SB_TAB + 3 * FGNN_H + c] - rb.x : 0.f;
        const float q = on ? sb_r(za * ra.y + ba, bf) - u0 - p * wii : 0.f;
        const float s = on ? sb_r(zb * rb.y + bb, bf) - v0 - r * wii : 0.f;
        float Cj = (colp[0][0][vx] + colp[1][0][vx]) + (colp[2][0][vx] + colp[3][0][vx]);
        float Qj = (colp[0][1][vx] + colp[1][1][vx]) + (colp[2][1][vx] + colp[3][1][vx]);
        float Uj = (colp[0][2][vx] + colp[1][2][vx]) + (colp[2][2][vx] + colp[3][2][vx]);
        if constexpr (NW == 8) {
            Cj += (colp[4][0][vx] + colp[5][0][vx]) + (colp[6][0][vx] + colp[7][0][vx]);
            Qj += (colp[4][1][vx] + colp[5][1][vx]) + (colp[6][1][vx] + colp[7][1][vx]);
            Uj += (colp[4][2][vx] + colp[5][2][vx]) + (colp[6][2][vx] + colp[7][2][vx]);
        }
        const float R = on ? Rs[vx] : 0.f, Pm = on ? Ps[vx] : 0.f, dg = on ? Dg[vx] : 0.f;
        const float degr = vi.x, degc = vi.y;
        const float d1 = on ? v0 * R + r * Pm + dg * s : 0.f;             // dA_ii = v0 R_i + r (dM W^T)_ii + dM_ii s_i
        const float d2 = on ? u0 * Cj + p * Qj + q * dg : 0.f;            // dB_ii = u0 C_i + p (W^T dM)_ii + q_i dM_ii
        o1[2 + vx] = d1;
        o2[2 + vx] = d2;
        float part[16] = {Cj, Uj, Cj * degc, Cj * s, R * degr, Qj * s, q * R, q * Pm, d1, wii * d1, d2, wii * d2, d1 * za, d2 * zb, 0.f, 0.f};
        const float tot = wave_sum_multi<16>(part, lane);
        if ((lane & 3) == 0) fin[wv][wave_sum_multi_index<16>(lane)] = tot;
    }
    __syncthreads();
    if (wv != 0) return;
#pragma unroll
    for (int e = 0; e < 14; ++e) {
        t[e] = fin[0][e];
#pragma unroll
        for (int w = 1; w < NWD; ++w) t[e] += fin[w][e];
    }
    SB_STAMP(2, blockIdx.x, 2)
    if (lane == 0) {
        const float Tt = t[0], Ut = t[1], sCdc = t[2], sCs = t[3], sRdr = t[4], sQs = t[5], sqR = t[6], sqP = t[7];
        const float sd1 = t[8], swd1 = t[9], sd2 = t[10], swd2 = t[11], sz1 = t[12], sz2 = t[13];
        // mlp1: dA = dM Y2^T,  dA_ij = v0 R_i + r (dM W^T)_ij + dM_ij s_j
        const float tot1 = v0 * fN * Tt + r * sCdc + sCs;
        const float sw1 = v0 * sRdr + r * Ut + sQs;
        // mlp2: dB = Y1^T dM,  dB_ij = u0 C_j + p (W^T dM)_ij + q_i dM_ij
        const float tot2 = u0 * fN * Tt + p * sRdr + sqR;
        const float sw2 = u0 * sCdc + p * Ut + sqP;
        const float off11 = sw1 - swd1, off12 = sw2 - swd2;
        const float off01 = tot1 - sd1 - off11, off02 = tot2 - sd2 - off12;
        o1[0] = off01;
        o1[1] = off11;
        o2[0] = off02;
        o2[1] = off12;
        const float s1a = (off01 + off11) + sd1, s2a = (off01 * za0 + off11 * za1) + sz1;
        const float s1b = (off02 + off12) + sd2, s2b = (off02 * zb0 + off12 * zb1) + sz2;
        const float mm = fN * fN;
        reinterpret_cast<float2 *>(s12_1)[(long long)g * FGNN_H + c] = make_float2(s1a, s2a);
        reinterpret_cast<float2 *>(s12_2)[(long long)g * FGNN_H + c] = make_float2(s1b, s2b);
        coef[(long long)g * FGNN_H + c] = make_float4(ra.y, mm > 0.f ? -ra.y * s2a * ra.w / mm : 0.f, mm > 0.f ? -ra.y * s1a / mm : 0.f, ra.x);
        coef[((long long)G + g) * FGNN_H + c] = make_float4(rb.y, mm > 0.f ? -rb.y * s2b * rb.w / mm : 0.f, mm > 0.f ? -rb.y * s1b / mm : 0.f, rb.x);
    }
    SB_STAMP(2, blockIdx.x, 3)
}

// ---- K3: the class algebra, KC instances per round, the rounds of a graph spread over wpg workgroups --------------
// grid (G wpg, 2), 256 threads: workgroup (g, w) of model m takes the instance chunks w, w + wpg, ... of graph g (instance 0 / 1 =
// the off-diagonal classes, 2 + v = vertex v) and writes row g wpg + w of wpart[m]
struct ParArgs {
    const float *W[2][3];
    float *wpart[2];
    int bf;
};
template <int KC>
__global__ __launch_bounds__(256) void sb_bwd_params_kernel(const float4 *vinfo, const float *gones, const int *nvalid, const int G, const int N,
                                                             const int CS, const float *tab, const float *csum, const float4 *coef,
                                                             const ParArgs A, const int wpg) {
    static_assert(KC % 8 == 0 && KC * FGNN_H % 256 == 0, "chunk size");
    __shared__ __attribute__((aligned(16))) float Wt[2][FGNN_H * FGNN_H];      // W1, W2 of this model
    __shared__ float CSs[FGNN_H][KC + 1];                                       // class sums [channel][instance]
    __shared__ __attribute__((aligned(16))) float HB[KC][2 * FGNN_H];           // h1 | h2
    __shared__ __attribute__((aligned(16))) float DZ[KC][FGNN_H + 4], D2[KC][FGNN_H + 4], D1[KC][FGNN_H + 4];
    __shared__ float XI[KC][2];                                                 // the instance's input value (w, degree)
    // G * wpg <= rows of wpart: workgroup (g, widx) owns every wpg-th instance round of graph g.  More graphs than rows (wpg == 1,
    // gridDim.x rows): workgroup b sums the graphs b, b + gridDim.x, ... into its one row
    const int g0 = blockIdx.x / wpg, widx = blockIdx.x - g0 * wpg, m = blockIdx.y, tid = threadIdx.x;
    const int gstep = G * wpg <= (int)gridDim.x ? G : (int)gridDim.x;
    const int NC = sb_classes(N);
    const float *tm = tab + (long long)m * NC * SB_TAB;
    const int c = tid & 31, kq = tid >> 5;
    {
        float w[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            w[q] = sb_r(A.W[m][1][tid + 256 * q], A.bf);
            w[4 + q] = sb_r(A.W[m][2][tid + 256 * q], A.bf);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            Wt[0][tid + 256 * q] = w[q];
            Wt[1][tid + 256 * q] = w[4 + q];
        }
    }
    const int wv = tid >> 6, lane = tid & 63, jj = lane & 31, hh = lane >> 5;
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float asum = 0.f, ab2 = 0.f;
#pragma unroll 1
    for (int g = g0; g < G; g += gstep) {
    const int nv = nvalid_of(nvalid, g, N), K = nv + 2;
    const float4 cf = coef[((long long)m * G + g) * FGNN_H + c];              // ca, cb, cc, mean of channel c
    const float ones = gones[g], fN = (float)nv;
    const float cnt0 = fN * fN - fN - ones, cnt1 = ones;
    const float *cs = csum + ((long long)m * G + g) * FGNN_H * CS;
#pragma unroll 1
    for (int ch = widx; ch * KC < K; ch += wpg) {
        const int kb = ch * KC, kn = min(KC, K - kb);
        constexpr int IT = KC / 8;
        float csv[IT], h1[IT], h2[IT], zv[IT], cntk[IT];
#pragma unroll
        for (int q = 0; q < IT; ++q) {
            {   // class sums: (channel, instance) = (e / KC, e % KC), contiguous over the instances of a channel
                const int e = tid + 256 * q, c2 = e / KC, kl = e - c2 * KC;
                csv[q] = kl < kn ? cs[(long long)c2 * CS + kb + kl] : 0.f;
            }
            const int kl = kq + 8 * q, k = kb + kl;
            int cl = k;
            float x0 = k == 1 ? 1.f : 0.f, x1 = 0.f;
            cntk[q] = k == 0 ? cnt0 : (k == 1 ? cnt1 : 1.f);
            if (k >= 2) {
                const float4 vi = vinfo[(long long)g * N + (k < K ? k - 2 : 0)];
                cl = __float_as_int(vi.w);
                x0 = (float)__float_as_int(vi.z);
                x1 = vi.x;
            }
            const float *row = tm + (long long)(k < K ? cl : 0) * SB_TAB + c;
            h1[q] = row[0];
            h2[q] = row[FGNN_H];
            zv[q] = row[3 * FGNN_H] - cf.w;           // z as stored - mean
            if (c == 0 && kl < kn) {
                XI[kl][0] = x0;
                XI[kl][1] = x1;
            }
        }
#pragma unroll
        for (int q = 0; q < IT; ++q) {
            const int e = tid + 256 * q, c2 = e / KC, kl = e - c2 * KC;
            CSs[c2][kl] = csv[q];
            const int kl2 = kq + 8 * q;
            HB[kl2][c] = h1[q];
            HB[kl2][FGNN_H + c] = h2[q];
        }
        __syncthreads();
        // dz summed over the pixels of an instance: ca S_k + n_k (cb (z_k - mean) + cc)
#pragma unroll
        for (int q = 0; q < IT; ++q) {
            const int kl = kq + 8 * q;
            DZ[kl][c] = kl < kn ? cf.x * CSs[c][kl] + cntk[q] * (cf.y * zv[q] + cf.z) : 0.f;
        }
        __syncthreads();
        // dpre2 = (W2^T dz) masked by h2, then dpre1 = (W1^T dpre2) masked by h1: thread (k mod 8, c) keeps column c of the transposed
        // weight in registers and reads an instance's vector with 128-bit broadcast loads
        {
            float wc[FGNN_H];
#pragma unroll
            for (int oo = 0; oo < FGNN_H; ++oo) wc[oo] = Wt[1][oo * FGNN_H + c];
#pragma unroll
            for (int q = 0; q < IT; ++q) {
                const int k = kq + 8 * q;
                const float4 *dz = reinterpret_cast<const float4 *>(DZ[k]);
                float a = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float4 v = dz[r];
                    a = fmaf(wc[4 * r + 0], v.x, a);
                    a = fmaf(wc[4 * r + 1], v.y, a);
                    a = fmaf(wc[4 * r + 2], v.z, a);
                    a = fmaf(wc[4 * r + 3], v.w, a);
                }
                D2[k][c] = (k < kn && h2[q] > 0.f) ? a : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int oo = 0; oo < FGNN_H; ++oo) wc[oo] = Wt[0][oo * FGNN_H + c];
#pragma unroll
            for (int q = 0; q < IT; ++q) {
                const int k = kq + 8 * q;
                const float4 *d2 = reinterpret_cast<const float4 *>(D2[k]);
                float a = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float4 v = d2[r];
                    a = fmaf(wc[4 * r + 0], v.x, a);
                    a = fmaf(wc[4 * r + 1], v.y, a);
                    a = fmaf(wc[4 * r + 2], v.z, a);
                    a = fmaf(wc[4 * r + 3], v.w, a);
                }
                D1[k][c] = (k < kn && h1[q] > 0.f) ? a : 0.f;
            }
        }
        __syncthreads();
        // Gradients = sums over the instances (rows kn .. KC - 1 of DZ / D2 / D1 are zero).  Wave 0: dW2 += dz_k (x) h2_k, wave 1:
        // dW1 += dpre2_k (x) h1_k on v_mfma_f32_32x32x2_f32 (lane (j, h) supplies row k = 2 s + h of both operands); waves 2, 3: the
        // 2-column dW0 and two bias gradients; the third bias gradient by wave 0 after its product
        if (wv < 2) {
            const float (*L)[FGNN_H + 4] = wv == 0 ? DZ : D2;
            const int hoff = wv == 0 ? FGNN_H : 0;
            float av[KC / 2], bv[KC / 2];
#pragma unroll
            for (int u = 0; u < KC / 2; ++u) {
                av[u] = L[2 * u + hh][jj];
                bv[u] = 2 * u + hh < kn ? HB[2 * u + hh][hoff + jj] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < KC / 2; ++u) acc = mfma32(av[u], bv[u], acc);
            if (wv == 0 && lane < 32) {
#pragma unroll
                for (int u = 0; u < KC; ++u) ab2 += DZ[u][lane];
            }
        } else if (tid < 128 + 64) {                         // dW0[t >> 1][t & 1] += dpre1_k x_k,  t = tid - 128
            const int t = tid - 128;
            float d[KC], x[KC];
#pragma unroll
            for (int u = 0; u < KC; ++u) {
                d[u] = D1[u][t >> 1];
                x[u] = u < kn ? XI[u][t & 1] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < KC; ++u) asum = fmaf(d[u], x[u], asum);
        } else {                                             // threads 192 .. 255: b0 += dpre1_k, b1 += dpre2_k
            const int t = tid - 192, l = t >> 5, cc2 = t & 31;
            const float (*V)[FGNN_H + 4] = l == 0 ? D1 : D2;
#pragma unroll
            for (int u = 0; u < KC; ++u) asum += V[u][cc2];
        }
        __syncthreads();                                      // the next round overwrites the arrays
    }
    }
    constexpr int PC = 32 * 2 + 32 + 2 * (32 * 32 + 32);
    float *row = A.wpart[m] + (long long)blockIdx.x * PC;
    if (wv < 2) {
        float *dst = row + (wv == 0 ? 96 + 1024 + 32 : 96);
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[ch_of(r, hh) * FGNN_H + jj] = acc[r];
        if (wv == 0 && lane < 32) row[96 + 1024 + 32 + 1024 + lane] = ab2;
    } else if (tid < 128 + 64) {
        row[tid - 128] = asum;
    } else {
        const int t = tid - 192;
        row[(t >> 5) == 0 ? 64 + (t & 31) : 96 + 1024 + (t & 31)] = asum;
    }
}

}  // namespace

#ifdef SB_STAMPS
extern "C" int fgnn_debug_sb_stamps(void *p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_sb_stamps), &p, sizeof(p)) == hipSuccess ? 0 : 1; }
#endif

extern "C" int fgnn_block1_struct_supported(int N, int depth, int c0) { return (N >= 1 && N <= SB_NMAX && depth == 3 && c0 == 2) ? 1 : 0; }
extern "C" int fgnn_block1_struct_table_floats(int N) { return 2 * (2 + 2 * (N + 1)) * SB_TAB; }
extern "C" long long fgnn_block1_struct_ws_floats(int G, int N) { return sb_ws_layout(G, N).total; }
extern "C" int fgnn_block1_struct_rows(int G, int N) {          // rows of wpart the backward pass writes (the reduction reads exactly these)
    const int rows = fgnn_mlp_bwd_num_workgroups();
    if (G >= rows) return rows;                                  // more graphs than rows: row b sums the graphs b, b + rows, ...
    const int chunks = (N + 2 + SB_KC - 1) / SB_KC, fit = rows / (G > 0 ? G : 1);
    const int wpg = chunks < fit ? chunks : fit;
    return G * (wpg < 1 ? 1 : wpg);
}

extern "C" int fgnn_block1_struct_tables(const float *const *W1, const float *const *b1, const float *const *W2, const float *const *b2, int N,
                                         int bf16_scheme, float *tables, void *stream) {
    FGNN_CHECK(W1 && b1 && W2 && b2 && tables && N >= 1 && N <= SB_NMAX, "fgnn_block1_struct_tables: bad arguments (N=%d)", N);
    TabArgs A;
    for (int l = 0; l < 3; ++l) {
        A.W[0][l] = W1[l];
        A.b[0][l] = b1[l];
        A.W[1][l] = W2[l];
        A.b[1][l] = b2[l];
        FGNN_CHECK(W1[l] && b1[l] && W2[l] && b2[l], "fgnn_block1_struct_tables: layer %d missing", l);
    }
    hipLaunchKernelGGL(sb_tables_kernel, dim3((2 * (2 + 2 * (N + 1)) + 3) / 4), dim3(256), 0, (hipStream_t)stream, A, N, tables, bf16_scheme ? 1 : 0);
    FGNN_LAUNCH_CHECK();
    return 0;
}

namespace {
struct FwdCall {
    const unsigned *bits;
    const int *nvalid;
    int G, N;
    const float *tables, *gnw1, *gnb1, *gnw2, *gnb2;
    float eps;
    float *nrm1, *nrm2;
    void *mult;
    long long gstride, ldp;
    int pitch;
    float *xdeg;
    void *x16;
    float *ws;
    const float *const *tW1, *const *tb1, *const *tW2, *const *tb2;      // optional: build the tables in the same launch
    hipStream_t st;
    const fgnn_pack_job *jobs;      // optional: the step's operand packing (fgnn_pack_operands) in the same launch
    int njobs;
};
template <int NWD, bool BF>
int sb_fwd_launch(const FwdCall &c) {
    const int parts0 = (512 + c.G * SB_CG - 1) / (c.G * SB_CG), parts = parts0 > 8 ? 8 : parts0;      // two 8-wave workgroups per CU
    const WsLayout L = sb_ws_layout(c.G, c.N);
    const int ny = (L.cp + SB_CW - 1) / SB_CW;                 // column bands of the code plane (its pitch, a multiple of 8, included)
    unsigned short *code = reinterpret_cast<unsigned short *>(c.ws + L.code);
    float4 *vinfo = reinterpret_cast<float4 *>(c.ws + L.vinfo);
    TabArgs TA = {};
    int ntab = 0;
    if (c.tW1) {
        for (int l = 0; l < 3; ++l) {
            FGNN_CHECK(c.tW1[l] && c.tb1[l] && c.tW2[l] && c.tb2[l], "fgnn_block1_struct_fwd: table weights of layer %d missing", l);
            TA.W[0][l] = c.tW1[l];
            TA.b[0][l] = c.tb1[l];
            TA.W[1][l] = c.tW2[l];
            TA.b[1][l] = c.tb2[l];
        }
        ntab = (2 * (2 + 2 * (c.N + 1)) + 3) / 4;
    }
    PackJobs PJ = {};
    FGNN_CHECK(c.njobs >= 0 && c.njobs <= FGNN_MAX_PACK_JOBS && (c.njobs == 0 || c.jobs), "fgnn_block1_struct_fwd_pack: 0 .. %d pack jobs", FGNN_MAX_PACK_JOBS);
    for (int i = 0; i < c.njobs; ++i) {
        FGNN_CHECK(c.jobs[i].out && c.jobs[i].depth >= 1 && c.jobs[i].depth <= FGNN_MAX_DEPTH && (c.jobs[i].nmlp == 1 || c.jobs[i].nmlp == 2),
                   "fgnn_block1_struct_fwd_pack: job %d malformed", i);
        PJ.job[i] = c.jobs[i];
    }
    if (BF) {
        for (int i = 0; i < c.njobs; ++i)
            FGNN_CHECK(c.jobs[i].depth >= 2 && (c.jobs[i].ca == 2 || c.jobs[i].ca == 32) && (c.jobs[i].cb == 0 || c.jobs[i].cb == 2 || c.jobs[i].cb == 32),
                       "fgnn_block1_struct_fwd16_pack: job %d: depth >= 2 and slab widths 2 or 32 (as fgnn_pack16_operands)", i);
    }
    const int npack = c.njobs * (BF ? PACK16_BLOCKS_PER_JOB : PACK_BLOCKS_PER_JOB);
    hipLaunchKernelGGL((sb_graph_kernel<NWD>), dim3(npack + ntab + c.G * ny), dim3(256), 0, c.st, c.bits, c.nvalid, c.N, L.cp, reinterpret_cast<unsigned *>(code),
                       vinfo, c.ws + L.gones, c.xdeg, c.x16, c.ldp, c.pitch, ny, ntab, TA, const_cast<float *>(c.tables), BF ? 1 : 0, npack, PJ);
    FGNN_LAUNCH_CHECK();
    hipLaunchKernelGGL((sb_fwd_kernel<NWD, BF>), dim3(c.G, SB_CG, parts), dim3(64 * SB_FW), 0, c.st, code, vinfo, c.ws + L.gones, c.nvalid, c.N, L.cp, c.tables,
                       c.gnw1, c.gnb1, c.gnw2, c.gnb2, c.eps, c.nrm1, c.nrm2, c.mult, c.gstride, c.ldp, c.pitch);
    FGNN_LAUNCH_CHECK();
    return 0;
}
template <bool BF>
int sb_fwd_dispatch(const FwdCall &c) {
    FGNN_CHECK(c.bits && c.tables && c.gnw1 && c.gnb1 && c.gnw2 && c.gnb2 && c.nrm1 && c.nrm2 && c.mult && c.ws && c.G > 0,
               "fgnn_block1_struct_fwd: bad arguments");
    FGNN_CHECK((c.tW1 != nullptr) == (c.tb1 != nullptr) && (c.tW1 != nullptr) == (c.tW2 != nullptr) && (c.tW1 != nullptr) == (c.tb2 != nullptr),
               "fgnn_block1_struct_fwd: the four table weight arrays come together");
    FGNN_CHECK(c.N >= 1 && c.N <= SB_NMAX, "fgnn_block1_struct_fwd: N = %d (built for N <= %d)", c.N, SB_NMAX);
    FGNN_CHECK((reinterpret_cast<uintptr_t>(c.ws) & 15) == 0, "fgnn_block1_struct_fwd: workspace not 16-byte aligned");
    if (c.N <= 64) return sb_fwd_launch<1, BF>(c);
    if (c.N <= 128) return sb_fwd_launch<2, BF>(c);
    return sb_fwd_launch<4, BF>(c);
}
}  // namespace

extern "C" int fgnn_block1_struct_fwd(const unsigned *bits, const int *nvalid, int G, int N, const float *tables, const float *gnw1,
                                      const float *gnb1, const float *gnw2, const float *gnb2, float eps, float *nrm1, float *nrm2, float *mult,
                                      long long gstride, long long ldp, float *xdeg, float *ws, const float *const *tW1,
                                      const float *const *tb1, const float *const *tW2, const float *const *tb2, void *stream) {
    FGNN_CHECK(ldp >= (long long)N * N && gstride >= FGNN_H * ldp, "fgnn_block1_struct_fwd: strides smaller than the planes");
    const FwdCall c = {bits, nvalid, G, N, tables, gnw1, gnb1, gnw2, gnb2, eps, nrm1, nrm2, mult, gstride, ldp, N, xdeg, nullptr, ws, tW1, tb1, tW2, tb2,
                       (hipStream_t)stream, nullptr, 0};
    return sb_fwd_dispatch<false>(c);
}

extern "C" int fgnn_block1_struct_fwd_pack(const unsigned *bits, const int *nvalid, int G, int N, const float *tables, const float *gnw1,
                                           const float *gnb1, const float *gnw2, const float *gnb2, float eps, float *nrm1, float *nrm2, float *mult,
                                           long long gstride, long long ldp, float *xdeg, float *ws, const float *const *tW1,
                                           const float *const *tb1, const float *const *tW2, const float *const *tb2,
                                           const fgnn_pack_job *jobs, int njobs, void *stream) {
    FGNN_CHECK(ldp >= (long long)N * N && gstride >= FGNN_H * ldp, "fgnn_block1_struct_fwd_pack: strides smaller than the planes");
    const FwdCall c = {bits, nvalid, G, N, tables, gnw1, gnb1, gnw2, gnb2, eps, nrm1, nrm2, mult, gstride, ldp, N, xdeg, nullptr, ws, tW1, tb1, tW2, tb2,
                       (hipStream_t)stream, jobs, njobs};
    return sb_fwd_dispatch<false>(c);
}

extern "C" int fgnn_block1_struct_fwd16(const unsigned *bits, const int *nvalid, int G, int N, int ldr, const float *tables, const float *gnw1,
                                        const float *gnb1, const float *gnw2, const float *gnb2, float eps, float *nrm1, float *nrm2, void *mult,
                                        long long gstride, long long ldp, void *x16, float *ws, const float *const *tW1,
                                        const float *const *tb1, const float *const *tW2, const float *const *tb2, void *stream) {
    FGNN_CHECK(ldr >= N && ldr % 8 == 0 && ldp >= (long long)N * ldr && ldp % 2 == 0 && gstride >= FGNN_H * ldp && gstride % 2 == 0,
               "fgnn_block1_struct_fwd16: pitches (ldr = %d, ldp = %lld, gstride = %lld)", ldr, ldp, gstride);
    const FwdCall c = {bits, nvalid, G, N, tables, gnw1, gnb1, gnw2, gnb2, eps, nrm1, nrm2, mult, gstride, ldp, ldr, nullptr, x16, ws, tW1, tb1, tW2, tb2,
                       (hipStream_t)stream, nullptr, 0};
    return sb_fwd_dispatch<true>(c);
}

extern "C" int fgnn_block1_struct_fwd16_pack(const unsigned *bits, const int *nvalid, int G, int N, int ldr, const float *tables, const float *gnw1,
                                             const float *gnb1, const float *gnw2, const float *gnb2, float eps, float *nrm1, float *nrm2, void *mult,
                                             long long gstride, long long ldp, void *x16, float *ws, const float *const *tW1,
                                             const float *const *tb1, const float *const *tW2, const float *const *tb2,
                                             const fgnn_pack_job *jobs, int njobs, void *stream) {
    FGNN_CHECK(ldr >= N && ldr % 8 == 0 && ldp >= (long long)N * ldr && ldp % 2 == 0 && gstride >= FGNN_H * ldp && gstride % 2 == 0,
               "fgnn_block1_struct_fwd16_pack: pitches (ldr = %d, ldp = %lld, gstride = %lld)", ldr, ldp, gstride);
    const FwdCall c = {bits, nvalid, G, N, tables, gnw1, gnb1, gnw2, gnb2, eps, nrm1, nrm2, mult, gstride, ldp, ldr, nullptr, x16, ws, tW1, tb1, tW2, tb2,
                       (hipStream_t)stream, jobs, njobs};
    return sb_fwd_dispatch<true>(c);
}

namespace {
struct BwdCall {
    const unsigned *bits;
    const int *nvalid;
    int G, N;
    const float *tables;
    const float *const *W1, *const *W2;
    const float *nrm1, *nrm2, *gnb1, *gnb2;
    const void *dmult;
    long long gstride, ldp;
    int pitch;
    float *ws, *wpart1, *wpart2, *s12_1, *s12_2;
    hipStream_t st;
};
template <int NWD, bool BF>
int sb_bwd_launch(const BwdCall &c) {
    const WsLayout L = sb_ws_layout(c.G, c.N);
    float *csum = c.ws + L.csum;
    const unsigned short *code = reinterpret_cast<const unsigned short *>(c.ws + L.code);
    const float4 *vinfo = reinterpret_cast<const float4 *>(c.ws + L.vinfo);
    float4 *coef = reinterpret_cast<float4 *>(c.ws + L.coef);
    hipLaunchKernelGGL((sb_bwd_reduce_kernel<NWD, BF>), dim3(c.G * FGNN_H), dim3(64 * sb_reduce_waves<NWD>()), 0, c.st, code, vinfo, c.nvalid, c.G, c.N, L.cp, c.tables, c.nrm1,
                       c.nrm2, c.gnb1, c.gnb2, c.dmult, c.gstride, c.ldp, c.pitch, csum, c.s12_1, c.s12_2, coef);
    FGNN_LAUNCH_CHECK();
    ParArgs A;
    for (int l = 0; l < 3; ++l) {
        A.W[0][l] = c.W1[l];
        A.W[1][l] = c.W2[l];
    }
    A.wpart[0] = c.wpart1;
    A.wpart[1] = c.wpart2;
    A.bf = BF ? 1 : 0;
    const int rows = fgnn_block1_struct_rows(c.G, c.N), wpg = rows >= c.G ? rows / c.G : 1;
    hipLaunchKernelGGL((sb_bwd_params_kernel<SB_KC>), dim3(rows, 2), dim3(256), 0, c.st, vinfo, c.ws + L.gones, c.nvalid, c.G, c.N, L.CS, c.tables, csum,
                       coef, A, wpg);
    FGNN_LAUNCH_CHECK();
    return 0;
}
template <bool BF>
int sb_bwd_dispatch(const BwdCall &c) {
    FGNN_CHECK(c.bits && c.tables && c.W1 && c.W2 && c.nrm1 && c.nrm2 && c.gnb1 && c.gnb2 && c.dmult && c.ws && c.wpart1 && c.wpart2 && c.s12_1 &&
                   c.s12_2 && c.G > 0,
               "fgnn_block1_struct_bwd: bad arguments");
    FGNN_CHECK(c.N >= 1 && c.N <= SB_NMAX, "fgnn_block1_struct_bwd: N = %d (built for N <= %d)", c.N, SB_NMAX);
    if (c.N <= 64) return sb_bwd_launch<1, BF>(c);
    if (c.N <= 128) return sb_bwd_launch<2, BF>(c);
    return sb_bwd_launch<4, BF>(c);
}
}  // namespace

extern "C" int fgnn_block1_struct_bwd(const unsigned *bits, const int *nvalid, int G, int N, const float *tables, const float *const *W1,
                                      const float *const *W2, const float *nrm1, const float *nrm2, const float *gnb1, const float *gnb2,
                                      const float *dmult, long long gstride, long long ldp, float *ws, float *wpart1, float *wpart2,
                                      float *s12_1, float *s12_2, void *stream) {
    const BwdCall c = {bits, nvalid, G, N, tables, W1, W2, nrm1, nrm2, gnb1, gnb2, dmult, gstride, ldp, N, ws, wpart1, wpart2, s12_1, s12_2,
                       (hipStream_t)stream};
    return sb_bwd_dispatch<false>(c);
}

extern "C" int fgnn_block1_struct_bwd16(const unsigned *bits, const int *nvalid, int G, int N, int ldr, const float *tables, const float *const *W1,
                                        const float *const *W2, const float *nrm1, const float *nrm2, const float *gnb1, const float *gnb2,
                                        const void *dmult, long long gstride, long long ldp, float *ws, float *wpart1, float *wpart2,
                                        float *s12_1, float *s12_2, void *stream) {
    FGNN_CHECK(ldr >= N && ldr % 8 == 0 && ldp >= (long long)N * ldr, "fgnn_block1_struct_bwd16: pitches (ldr = %d, ldp = %lld)", ldr, ldp);
    const BwdCall c = {bits, nvalid, G, N, tables, W1, W2, nrm1, nrm2, gnb1, gnb2, dmult, gstride, ldp, ldr, ws, wpart1, wpart2, s12_1, s12_2,
                       (hipStream_t)stream};
    return sb_bwd_dispatch<true>(c);
}

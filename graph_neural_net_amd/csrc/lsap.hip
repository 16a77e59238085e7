// Linear sum assignment on the device for accuracy_linear_assignment (toolbox/metrics.py:92-116: per graph
// `_, preds = scipy.optimize.linear_sum_assignment(-log_softmax(scores))`, accuracy = #{i : preds[i] == i}).
//
// The algorithm is the one behind scipy.optimize.linear_sum_assignment (SciPy >= 1.4: the shortest-augmenting-path method
// of D. F. Crouse, "On implementing 2D rectangular assignment algorithms", IEEE TAES 52(4), 2016), restated with the same
// arithmetic (fp64 duals, the reduced cost formed as ((minVal + c) - u) - v, no multiplications, hence nothing to contract)
// and the same tie rules, so that the assignment -- not only its cost -- is the one SciPy returns:
//   * the candidate columns of a row scan are kept in the `remaining` list, filled in reverse order and compacted by moving
//     the last entry into the freed slot;
//   * the scan takes a strictly smaller path cost, or an equal one when that column is still unassigned; scanning
//     sequentially this selects, among the columns with the minimal cost m, the LAST unassigned one in list order if there
//     is one, otherwise the FIRST column with cost m.
// One wave per graph, a lane owns every 64th column (state in registers, four wave reductions on the DPP network per
// augmentation step); the row-indexed state and the pointer walk of the augmentation are in LDS.  n <= FGNN_LSAP_MAX_N.
#include "fgnn_common.h"

namespace {

constexpr size_t LSAP_LDS_BYTES = 160 * 1024;          // solver state (29 bytes per vertex) + as many cost rows as fit

// lanes of the wave exchange values through LDS: order the accesses for the compiler (LDS itself is in order per wave)
DEVI void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Wave reductions on the DPP network (no LDS crossbar round trips: the solver is a chain of short dependent steps and three
// reductions per step; with ds_bpermute shuffles they were most of its time).  Four steps inside the rows of 16 lanes, then
// row_bcast15 into rows 1 and 3 and row_bcast31 into rows 2 and 3: lane 63 holds the result, read back as a scalar.
template <int CTRL, int ROW_MASK>
DEVI int dpp_i32(int own, int src) {
    return __builtin_amdgcn_update_dpp(own, src, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
DEVI double dpp_f64(double x) {
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = (int)b, hi = (int)(b >> 32);
    const int l2 = dpp_i32<CTRL, ROW_MASK>(lo, lo), h2 = dpp_i32<CTRL, ROW_MASK>(hi, hi);
    return __builtin_bit_cast(double, ((long long)h2 << 32) | (unsigned)l2);
}
DEVI double wave_min_f64(double v) {
    v = fmin(v, dpp_f64<0xB1, 0xf>(v));       // quad_perm [1,0,3,2]
    v = fmin(v, dpp_f64<0x4E, 0xf>(v));       // quad_perm [2,3,0,1]
    v = fmin(v, dpp_f64<0x141, 0xf>(v));      // row_half_mirror
    v = fmin(v, dpp_f64<0x140, 0xf>(v));      // row_mirror
    v = fmin(v, dpp_f64<0x142, 0xa>(v));      // row_bcast15 -> rows 1, 3
    v = fmin(v, dpp_f64<0x143, 0xc>(v));      // row_bcast31 -> rows 2, 3
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
DEVI int wave_min_i32(int v) {
    v = min(v, dpp_i32<0xB1, 0xf>(v, v));
    v = min(v, dpp_i32<0x4E, 0xf>(v, v));
    v = min(v, dpp_i32<0x141, 0xf>(v, v));
    v = min(v, dpp_i32<0x140, 0xf>(v, v));
    v = min(v, dpp_i32<0x142, 0xa>(v, v));
    v = min(v, dpp_i32<0x143, 0xc>(v, v));
    return __builtin_amdgcn_readlane(v, 63);
}
DEVI int wave_max_i32(int v) {
    v = max(v, dpp_i32<0xB1, 0xf>(v, v));
    v = max(v, dpp_i32<0x4E, 0xf>(v, v));
    v = max(v, dpp_i32<0x141, 0xf>(v, v));
    v = max(v, dpp_i32<0x140, 0xf>(v, v));
    v = max(v, dpp_i32<0x142, 0xa>(v, v));
    v = max(v, dpp_i32<0x143, 0xc>(v, v));
    return __builtin_amdgcn_readlane(v, 63);
}

// NC = columns per lane (lane l owns columns l, l + 64, ...: their dual v, path cost, predecessor, matched row and position
// in SciPy's `remaining` list live in registers); the row-indexed state (u, col4row, SR) and what the pointer walk of the
// augmentation needs (path, row4col) are in LDS, and so are the first `rows_staged` rows of the n x n cost corner (all of
// them up to n = 198; row pitch n): a row scan then reads LDS instead of paying a global-memory round trip per step.
// The list is never materialised: a column knows its position, the minimum / first / last-unassigned positions are wave
// reductions, the column at the chosen position identifies itself, and the compaction "move the last entry into the freed
// slot" is the lane that sits at the last position rewriting its own position.
template <int NC>
__global__ __launch_bounds__(64) void lsap_kernel(const float *cost, long long bstride, int ld, const int *nvalid, int B, int N,
                                                  int *correct, int *assign, int stage_floats) {
    // dynamic LDS: solver state sized for N, then the staged cost rows
    extern __shared__ __attribute__((aligned(16))) unsigned char lsap_lds[];
    const int NA = (N + 3) & ~3;
    double *u = reinterpret_cast<double *>(lsap_lds), *spc_l = u + NA;
    int *path_l = reinterpret_cast<int *>(spc_l + NA), *col4row = path_l + NA, *row4col = col4row + NA;
    unsigned char *SR = reinterpret_cast<unsigned char *>(row4col + NA);
    float *cl = reinterpret_cast<float *>(SR + NA);
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = nvalid_of(nvalid, b, N);
    const float *cb = cost + (long long)b * bstride;
    const int rows_staged = n > 0 ? min(n, stage_floats / n) : 0;
    for (int r = 0; r < rows_staged; ++r)
        for (int k = lane; k < n; k += 64) cl[r * n + k] = cb[(long long)r * ld + k];
    for (int k = lane; k < n; k += 64) {
        u[k] = 0.0;
        col4row[k] = -1;
        row4col[k] = -1;
    }
    double v[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) v[c] = 0.0;
    wsync();
    bool feasible = true;
    for (int cur = 0; cur < n && feasible; ++cur) {
        // ---- shortest augmenting path from row `cur` ----
        double minVal = 0.0, spc[NC];
        int num_remaining = n, pos[NC], r4c[NC], path[NC];
        bool SC[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int j = lane + 64 * c;
            pos[c] = j < n ? n - 1 - j : -1;            // remaining[it] = n - it - 1
            r4c[c] = j < n ? row4col[j] : 0;
            path[c] = -1;
            spc[c] = INFINITY;
            SC[c] = false;
        }
        for (int k = lane; k < n; k += 64) SR[k] = 0;
        wsync();
        int sink = -1, i = cur;
        while (sink == -1) {
            if (lane == 0) SR[i] = 1;
            const double ui = u[i];
            const float *crow = i < rows_staged ? cl + i * n : cb + (long long)i * ld;
            // branch-free: unconditional (clamped) loads, predicates instead of divergent blocks around fp64 code
            float cr[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) cr[c] = crow[min(lane + 64 * c, n - 1)];
            double lowest = INFINITY;
            int first_it = 0x7fffffff;          // first list position with the lane's minimal cost
            int last_free = -1;                 // last list position with that cost whose column is unassigned
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const bool act = pos[c] >= 0;
                const double r = ((minVal + (double)cr[c]) - ui) - v[c];
                const bool upd = act && r < spc[c];
                path[c] = upd ? i : path[c];
                spc[c] = upd ? r : spc[c];
                const double s = act ? spc[c] : INFINITY;
                const bool free_col = r4c[c] == -1;
                const bool lt = s < lowest, eq = act && s == lowest;
                first_it = lt ? pos[c] : (eq ? min(first_it, pos[c]) : first_it);
                last_free = lt ? (free_col ? pos[c] : -1) : ((eq && free_col) ? max(last_free, pos[c]) : last_free);
                lowest = lt ? s : lowest;
            }
            const double m = wave_min_f64(lowest);
            if (!(m < INFINITY)) {              // infeasible (or NaN) cost matrix: SciPy raises; here the graph counts 0
                feasible = false;
                break;
            }
            const bool mine = lowest == m;
            const int it1 = wave_min_i32(mine ? first_it : 0x7fffffff);
            const int itf = wave_max_i32(mine ? last_free : -1);
            const int index = itf >= 0 ? itf : it1;
            minVal = m;
            // the column at list position `index` identifies itself: (column, matched row + 1) in one reduction
            int who = -1;
            --num_remaining;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const bool chosen = pos[c] == index, last = pos[c] == num_remaining;
                who = chosen ? ((lane + 64 * c) | ((r4c[c] + 1) << 12)) : who;
                SC[c] = SC[c] || chosen;
                pos[c] = chosen ? -1 : (last ? index : pos[c]);     // remaining[index] = remaining[--num_remaining]
            }
            who = wave_max_i32(who);
            const int j = who & 4095, r4 = (who >> 12) - 1;
            if (r4 == -1) sink = j;
            else i = r4;
        }
        if (!feasible) break;
        // ---- dual update ----
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int j = lane + 64 * c;
            if (j < n) {
                spc_l[j] = spc[c];
                path_l[j] = path[c];
                if (SC[c]) v[c] -= minVal - spc[c];
            }
        }
        wsync();
        for (int k = lane; k < n; k += 64) {
            if (k == cur) u[k] += minVal;
            else if (SR[k]) u[k] += minVal - spc_l[col4row[k]];
        }
        wsync();
        // ---- augment (wave-uniform pointer walk; lane 0 writes) ----
        int j = sink;
        while (true) {
            const int pi = path_l[j];
            const int old = col4row[pi];
            wsync();
            if (lane == 0) {
                row4col[j] = pi;
                col4row[pi] = j;
            }
            wsync();
            j = old;
            if (pi == cur) break;
        }
    }
    int hit = 0;
    for (int k = lane; k < n; k += 64) {
        const int cidx = feasible ? col4row[k] : -1;
        hit += cidx == k ? 1 : 0;
        if (assign) assign[(long long)b * N + k] = cidx;
    }
    if (assign)
        for (int k = n + lane; k < N; k += 64) assign[(long long)b * N + k] = -1;
    for (int o = 32; o > 0; o >>= 1) hit += __shfl_xor(hit, o);
    if (lane == 0) correct[b] = hit;
}

template <int NC>
int launch_lsap(const float *cost, long long bstride, int ld, const int *nvalid, int B, int N, int *correct, int *assign,
                hipStream_t st) {
    const size_t state = (size_t)((N + 3) & ~3) * (8 + 8 + 4 + 4 + 4 + 1);
    size_t stage = (size_t)N * N * sizeof(float);
    if (state + stage > LSAP_LDS_BYTES) stage = (LSAP_LDS_BYTES - state) & ~(size_t)15;
    const size_t lds = state + stage;
    static LdsAttrCache attr_cache;
    FGNN_CHECK(fgnn_raise_lds(attr_cache, (const void *)lsap_kernel<NC>, lds), "fgnn_lsap_accuracy: %zu bytes of LDS refused", lds);
    hipLaunchKernelGGL((lsap_kernel<NC>), dim3(B), dim3(64), lds, st, cost, bstride, ld, nvalid, B, N, correct, assign,
                       (int)(stage / sizeof(float)));
    FGNN_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int fgnn_lsap_accuracy(const float *cost, long long bstride, int ld, const int *nvalid, int B, int N, int *correct,
                                  int *assign, void *stream) {
    FGNN_CHECK(cost && correct && B > 0 && N > 0, "fgnn_lsap_accuracy: bad arguments");
    FGNN_CHECK(N <= FGNN_LSAP_MAX_N, "fgnn_lsap_accuracy: at most %d vertices per graph (got %d)", FGNN_LSAP_MAX_N, N);
    FGNN_CHECK(ld >= N && bstride >= (long long)N * ld, "fgnn_lsap_accuracy: strides smaller than the matrices");
    hipStream_t st = (hipStream_t)stream;
    if (N <= 64) return launch_lsap<1>(cost, bstride, ld, nvalid, B, N, correct, assign, st);
    if (N <= 128) return launch_lsap<2>(cost, bstride, ld, nvalid, B, N, correct, assign, st);
    if (N <= 256) return launch_lsap<4>(cost, bstride, ld, nvalid, B, N, correct, assign, st);
    if (N <= 512) return launch_lsap<8>(cost, bstride, ld, nvalid, B, N, correct, assign, st);
    if (N <= 1024) return launch_lsap<16>(cost, bstride, ld, nvalid, B, N, correct, assign, st);
    return launch_lsap<32>(cost, bstride, ld, nvalid, B, N, correct, assign, st);
}

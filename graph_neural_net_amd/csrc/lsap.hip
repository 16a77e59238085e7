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
// One wave per graph: the lanes scan the list in slices of 64 (lane-private best candidates, three wave reductions per
// scan), everything else is wave-uniform bookkeeping in LDS.  n <= FGNN_LSAP_MAX_N.
#include "fgnn_common.h"

namespace {

constexpr int LSAP_MAX = FGNN_LSAP_MAX_N;

// lanes of the wave exchange values through LDS: order the accesses for the compiler (LDS itself is in order per wave)
DEVI void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
DEVI double wave_min_f64(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}
DEVI int wave_min_i32(int v) {
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}
DEVI int wave_max_i32(int v) {
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}

__global__ __launch_bounds__(64) void lsap_kernel(const float *cost, long long bstride, int ld, const int *nvalid, int B, int N,
                                                  int *correct, int *assign) {
    __shared__ double u[LSAP_MAX], v[LSAP_MAX], spc[LSAP_MAX];
    __shared__ int path[LSAP_MAX], col4row[LSAP_MAX], row4col[LSAP_MAX], remaining[LSAP_MAX];
    __shared__ unsigned char SR[LSAP_MAX], SC[LSAP_MAX];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = nvalid_of(nvalid, b, N);
    const float *cb = cost + (long long)b * bstride;
    for (int k = lane; k < n; k += 64) {
        u[k] = 0.0;
        v[k] = 0.0;
        path[k] = -1;
        col4row[k] = -1;
        row4col[k] = -1;
    }
    wsync();
    bool feasible = true;
    for (int cur = 0; cur < n && feasible; ++cur) {
        // ---- shortest augmenting path from row `cur` ----
        double minVal = 0.0;
        int num_remaining = n;
        for (int k = lane; k < n; k += 64) {
            remaining[k] = n - k - 1;
            SR[k] = 0;
            SC[k] = 0;
            spc[k] = INFINITY;
        }
        wsync();
        int sink = -1, i = cur;
        while (sink == -1) {
            if (lane == 0) SR[i] = 1;
            const double ui = u[i];
            const float *crow = cb + (long long)i * ld;
            // lane-private scan of its slice of the list, in list order
            double lowest = INFINITY;
            int first_it = 0x7fffffff;          // first list position with the lane's minimal cost
            int last_free = -1;                 // last list position with that cost whose column is unassigned
            for (int it = lane; it < num_remaining; it += 64) {
                const int j = remaining[it];
                const double r = ((minVal + (double)crow[j]) - ui) - v[j];
                double s = spc[j];
                if (r < s) {
                    path[j] = i;
                    spc[j] = r;
                    s = r;
                }
                const bool free_col = row4col[j] == -1;
                if (s < lowest) {
                    lowest = s;
                    first_it = it;
                    last_free = free_col ? it : -1;
                } else if (s == lowest && free_col) {
                    last_free = it;
                }
            }
            wsync();
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
            wsync();
            const int j = remaining[index];
            const int r4c = row4col[j];
            if (r4c == -1) sink = j;
            else i = r4c;
            --num_remaining;
            wsync();
            if (lane == 0) {
                SC[j] = 1;
                remaining[index] = remaining[num_remaining];
            }
            wsync();
        }
        if (!feasible) break;
        // ---- dual update ----
        for (int k = lane; k < n; k += 64) {
            if (k == cur) u[k] += minVal;
            else if (SR[k]) u[k] += minVal - spc[col4row[k]];
            if (SC[k]) v[k] -= minVal - spc[k];
        }
        wsync();
        // ---- augment (wave-uniform pointer walk; lane 0 writes) ----
        int j = sink;
        while (true) {
            const int pi = path[j];
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

}  // namespace

extern "C" int fgnn_lsap_accuracy(const float *cost, long long bstride, int ld, const int *nvalid, int B, int N, int *correct,
                                  int *assign, void *stream) {
    FGNN_CHECK(cost && correct && B > 0 && N > 0, "fgnn_lsap_accuracy: bad arguments");
    FGNN_CHECK(N <= FGNN_LSAP_MAX_N, "fgnn_lsap_accuracy: at most %d vertices per graph (got %d)", FGNN_LSAP_MAX_N, N);
    FGNN_CHECK(ld >= N && bstride >= (long long)N * ld, "fgnn_lsap_accuracy: strides smaller than the matrices");
    hipLaunchKernelGGL(lsap_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, cost, bstride, ld, nvalid, B, N, correct, assign);
    FGNN_LAUNCH_CHECK();
    return 0;
}

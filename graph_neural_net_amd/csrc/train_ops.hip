// Training-step helpers next to the hot path: fused Adam over the flat parameter
// buffer (models/trainers.py:92-104 uses torch.optim.Adam, amsgrad=False) and the argmax accuracy
// metric (toolbox/metrics.py:119-141) on the device.
#include "fgnn_common.h"

namespace {

// torch.optim.Adam single-tensor semantics (no weight decay, no amsgrad):
//   m = lerp(m, g, 1-b1); v = b2*v + (1-b2)*g*g; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void adam_kernel(float *p, const float *g, float *m, float *v, int n, float step_size, float w1,
                            float beta2, float w2, float inv_sqrt_bc2, float eps, float grad_scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * grad_scale;
    const float mi = m[i] + (gi - m[i]) * w1;
    const float vi = v[i] * beta2 + gi * gi * w2;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    p[i] = p[i] - step_size * (mi / denom);
}

// The same update with the step count and the hyper-parameters read from device memory, so that the launch can
// sit in a captured HIP graph and be replayed: hp = {lr, beta1, beta2, eps, grad_scale} (doubles), state =
// {step count, arrivals}.  Every workgroup reads the step count before it signs the arrival counter; the last one
// to sign advances the count for the next replay.
__global__ __launch_bounds__(256) void adam_dev_kernel(float *p, const float *g, float *m, float *v, int n,
                                                       const double *hp, int *state) {
    __shared__ float sh[6];
    __shared__ int s_step;
    if (threadIdx.x == 0) {
        const int step = state[0] + 1;
        const double lr = hp[0], b1 = hp[1], b2 = hp[2];
        const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
        sh[0] = (float)(lr / bc1);
        sh[1] = (float)(1.0 - b1);
        sh[2] = (float)b2;
        sh[3] = (float)(1.0 - b2);
        sh[4] = (float)(1.0 / sqrt(bc2));
        sh[5] = (float)hp[3];
        s_step = step;
    }
    __syncthreads();
    const float step_size = sh[0], w1 = sh[1], beta2 = sh[2], w2 = sh[3], inv_sqrt_bc2 = sh[4], eps = sh[5];
    const float grad_scale = (float)hp[4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float gi = g[i] * grad_scale;
        const float mi = m[i] + (gi - m[i]) * w1;
        const float vi = v[i] * beta2 + gi * gi * w2;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&state[1], 1) == (int)gridDim.x - 1) {      // every workgroup has read state[0] by now
            state[1] = 0;
            state[0] = s_step;
        }
    }
}

// one workgroup per pair: rows i < nv with argmax_j scores[i][j] (first maximum) == i
__global__ __launch_bounds__(256) void accuracy_max_kernel(const float *scores, const int *nvalid, int N, int *correct) {
    __shared__ int red[4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = nvalid_of(nvalid, b, N);
    const float *S = scores + (long long)b * N * N;
    int cnt = 0;
    for (int i = wave; i < nv; i += 4) {
        float best = -3.402823466e+38f;
        int bj = 0x7fffffff;
        for (int j = lane; j < nv; j += WAVE) {
            const float v = S[(long long)i * N + j];
            if (v > best) {
                best = v;
                bj = j;
            }
        }
        // wave arg-max with ties resolved towards the smaller index (np.argmax)
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o);
            const int oj = __shfl_xor(bj, o);
            if (ob > best || (ob == best && oj < bj)) {
                best = ob;
                bj = oj;
            }
        }
        if (bj == i) ++cnt;
    }
    if (lane == 0) red[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) correct[b] = red[0] + red[1] + red[2] + red[3];
}

}  // namespace

extern "C" int fgnn_adam_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int n, double lr,
                              double beta1, double beta2, double eps, int step, double grad_scale, void *stream) {
    FGNN_CHECK(params && grads && exp_avg && exp_avg_sq && n > 0 && step >= 1, "fgnn_adam_step: bad arguments");
    // hyper-parameters arrive as doubles (like the Python floats torch.optim.Adam works with) and are
    // rounded to fp32 only after 1 - beta etc. have been formed
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                       exp_avg_sq, n, (float)(lr / bc1), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                       (float)(1.0 / sqrt(bc2)), (float)eps, (float)grad_scale);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_adam_step_dev(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int n,
                                  const double *hp, int *state, void *stream) {
    FGNN_CHECK(params && grads && exp_avg && exp_avg_sq && hp && state && n > 0, "fgnn_adam_step_dev: bad arguments");
    hipLaunchKernelGGL(adam_dev_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                       exp_avg_sq, n, hp, state);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_accuracy_max(const float *scores, const int *nvalid, int B, int N, int *correct, void *stream) {
    FGNN_CHECK(scores && correct && B > 0 && N > 0, "fgnn_accuracy_max: bad arguments");
    hipLaunchKernelGGL(accuracy_max_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, scores, nvalid, N, correct);
    FGNN_LAUNCH_CHECK();
    return 0;
}

// ---- input pipeline ------------------------------------------------
// The hot path's inputs are 0/1 adjacency matrices plus diag(degree) (loaders/data_generator.py:118-125).
// They travel to the device bit-packed (N*ceil(N/32) words per graph, 64x less PCIe traffic than fp32)
// and are expanded here: x[g,0] = W, x[g,1] = diag(row sums); padding beyond nvalid[g] is zero.
namespace {
__global__ void expand_adjacency_kernel(const unsigned *bits, const int *nvalid, int G, int N, int words, float *x) {
    // one thread per output element: coalesced stores of both channels
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long P = (long long)N * N;
    if (t >= (long long)G * P) return;
    const int g = (int)(t / P);
    const int p = (int)(t - (long long)g * P);
    const int i = p / N, j = p - i * N;
    const int nv = nvalid_of(nvalid, g, N);
    const unsigned *row = bits + ((long long)g * N + i) * words;
    const bool on = i < nv && j < nv && ((row[j >> 5] >> (j & 31)) & 1u);
    float deg = 0.f;
    if (i == j && i < nv) {
        int d = 0;
        for (int k = 0; k < words; ++k) {
            const int left = nv - 32 * k;  // valid bits in this word
            const unsigned m = left >= 32 ? 0xffffffffu : (left <= 0 ? 0u : ((1u << left) - 1u));
            d += __popc(row[k] & m);
        }
        deg = (float)d;
    }
    float *xg = x + (long long)g * 2 * P;
    xg[p] = on ? 1.f : 0.f;
    xg[P + p] = deg;
}
__global__ void adjacency_degree_kernel(const unsigned *bits, const int *nvalid, int G, int N, int words, float *deg) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)G * N) return;
    const int g = (int)(t / N), i = (int)(t - (long long)g * N);
    const int nv = nvalid_of(nvalid, g, N);
    const unsigned *row = bits + t * words;
    int d = 0;
    if (i < nv) {
        for (int k = 0; k < words; ++k) {
            const int left = nv - 32 * k;
            const unsigned m = left >= 32 ? 0xffffffffu : (left <= 0 ? 0u : ((1u << left) - 1u));
            d += __popc(row[k] & m);
        }
    }
    deg[t] = (float)d;
}
}  // namespace

extern "C" int fgnn_adjacency_degree(const unsigned *bits, const int *nvalid, int G, int N, float *deg, void *stream) {
    FGNN_CHECK(bits && deg && G > 0 && N > 0, "fgnn_adjacency_degree: bad arguments");
    const long long tot = (long long)G * N;
    hipLaunchKernelGGL(adjacency_degree_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, bits,
                       nvalid, G, N, (N + 31) / 32, deg);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_expand_adjacency(const unsigned *bits, const int *nvalid, int G, int N, float *x, void *stream) {
    FGNN_CHECK(bits && x && G > 0 && N > 0, "fgnn_expand_adjacency: bad arguments");
    const long long tot = (long long)G * N * N;
    hipLaunchKernelGGL(expand_adjacency_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       bits, nvalid, G, N, (N + 31) / 32, x);
    FGNN_LAUNCH_CHECK();
    return 0;
}

// ---- the inverse of fgnn_expand_adjacency: (G, 2, N, N) tensor representation -> bit-packed adjacency, with a check that the
// tensor IS a tensor representation (channel 0 in {0, 1}; channel 1 = diag(row sums of channel 0) on the valid corner).  One wave per
// row: lane = column (64 per pass), the row's words are ballots; `bad` (optional, device int) is OR-ed with 1 where the check fails.
namespace {
// x2 != nullptr (fgnn_pack_adjacency_pair): the G graphs are the B = G / 2 graphs of x followed by the B graphs of x2, each side with its
// own vertex counts (nvalid / nvalid2, B entries each); the counts are also copied to nv_out (G entries) and block 0 leaves 1 / sum(nvalid)
// in inv_out -- what two launches per side, two copies and fgnn_inv_node_count did
__global__ __launch_bounds__(256) void pack_adjacency_kernel(const float *x, const int *nvalid, int G, int Nin, int N, int words,
                                                             unsigned *bits, int *bad, const float *x2, const int *nvalid2, int *nv_out,
                                                             float *inv_out) {
    // x: (G, 2, Nin, Nin); bits: (G, N, words) with N >= Nin -- rows / columns >= Nin are empty
    const int B = x2 ? G / 2 : G;
    if (blockIdx.x == 0 && threadIdx.x < 64 && x2 && nvalid) {
        if (inv_out) {
            int sum = 0;
            for (int b = threadIdx.x; b < B; b += 64) sum += nvalid[b];
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            if (threadIdx.x == 0) inv_out[0] = sum > 0 ? 1.f / (float)sum : 0.f;
        }
        if (nv_out) {
            for (int b = threadIdx.x; b < G; b += 64) nv_out[b] = b < B ? nvalid[b] : nvalid2[b - B];
        }
    }
    const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);      // row over (g, i)
    if (t >= (long long)G * N) return;
    const int lane = threadIdx.x & 63, g = (int)(t / N), i = (int)(t - (long long)g * N);
    const bool second = g >= B;           // (only with x2)
    const int gs = second ? g - B : g;
    const float *xs = second ? x2 : x;
    const int *nvs = second ? nvalid2 : nvalid;
    const int nv = min(nvalid_of(nvs, gs, N), Nin);
    const float *w = xs + ((long long)gs * 2) * Nin * Nin + (long long)i * Nin, *d = w + (long long)Nin * Nin;
    int deg = 0;
    bool wrong = nvs && (nvs[gs] > Nin || nvs[gs] < 0);
    float dii = 0.f;
    for (int j0 = 0; j0 < 32 * words; j0 += 64) {
        const int j = j0 + lane;
        const bool in = i < nv && j < nv;
        const float v = in ? w[j] : 0.f, dv = in ? d[j] : 0.f;
        wrong |= in && v != 0.f && v != 1.f;
        wrong |= in && j != i && dv != 0.f;
        if (in && j == i) dii = dv;
        const unsigned long long m = __ballot(v != 0.f);
        deg += __popcll(m);
        if (lane == 0) {
            bits[t * words + (j0 >> 5)] = (unsigned)m;
            if ((j0 >> 5) + 1 < words) bits[t * words + (j0 >> 5) + 1] = (unsigned)(m >> 32);
        }
    }
    dii = wave_sum(dii);                                       // (one lane holds it)
    wrong |= i < nv && dii != (float)deg;
    if (bad && __ballot(wrong) != 0ull && lane == 0) atomicOr(bad, 1);
}
}  // namespace

extern "C" int fgnn_pack_adjacency_ld(const float *x, const int *nvalid, int G, int Nin, int N, unsigned *bits, int *bad, void *stream) {
    FGNN_CHECK(x && bits && G > 0 && Nin > 0 && N >= Nin, "fgnn_pack_adjacency_ld: bad arguments");
    const long long rows = (long long)G * N;
    hipLaunchKernelGGL(pack_adjacency_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, nvalid, G, Nin, N,
                       (N + 31) / 32, bits, bad, (const float *)nullptr, (const int *)nullptr, (int *)nullptr, (float *)nullptr);
    FGNN_LAUNCH_CHECK();
    return 0;
}
extern "C" int fgnn_pack_adjacency_pair(const float *x1, const float *x2, const int *nvalid1, const int *nvalid2, int B, int Nin, int N,
                                        unsigned *bits, int *nv_out, float *inv_out, int *bad, void *stream) {
    FGNN_CHECK(x1 && x2 && bits && B > 0 && Nin > 0 && N >= Nin, "fgnn_pack_adjacency_pair: bad arguments");
    FGNN_CHECK((nvalid1 == nullptr) == (nvalid2 == nullptr) && (nvalid1 || (!nv_out && !inv_out)),
               "fgnn_pack_adjacency_pair: both sides ragged or none; nv_out / inv_out need the vertex counts");
    const long long rows = 2ll * B * N;
    hipLaunchKernelGGL(pack_adjacency_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x1, nvalid1, 2 * B, Nin, N,
                       (N + 31) / 32, bits, bad, x2, nvalid2, nv_out, inv_out);
    FGNN_LAUNCH_CHECK();
    return 0;
}
extern "C" int fgnn_pack_adjacency(const float *x, const int *nvalid, int G, int N, unsigned *bits, int *bad, void *stream) {
    return fgnn_pack_adjacency_ld(x, nvalid, G, N, N, bits, bad, stream);
}

// ---- ragged batches: work-balanced tile ranges for the persistent MLP kernels (fgnn_mlp_fwd_args.ranges) -------------
namespace {
constexpr int RANGE_THREADS = 1024;
constexpr int COST_LIVE = 16, COST_DEAD = 1;      // a padding-only tile is a zero-fill, any other a full pass

// walks consecutive tiles without a division per tile: (graph, tile in graph, first element's row / column)
struct TileWalk {
    int g, tt, i0, j0, nv;
    __device__ void start(int t, int tpg, int T, int pitch, const int *nvalid) {
        g = t / tpg;
        tt = t - g * tpg;
        i0 = tt * T / pitch;
        j0 = tt * T - i0 * pitch;
        nv = nvalid[g];
    }
    __device__ int cost(int T, int pitch) const {       // tile_live_p() of fgnn_common.h on the tracked coordinates
        const bool multi = j0 + T > pitch;                     // the tile reaches into row i0 + 1
        return ((i0 < nv && j0 < nv) || (multi && i0 + 1 < nv)) ? COST_LIVE : COST_DEAD;
    }
    __device__ void next(int tpg, int T, int pitch, const int *nvalid, int G) {
        if (++tt == tpg) {
            tt = 0;
            i0 = j0 = 0;
            if (++g < G) nv = nvalid[g];
            return;
        }
        j0 += T;
        while (j0 >= pitch) {
            j0 -= pitch;
            ++i0;
        }
    }
};

__global__ __launch_bounds__(RANGE_THREADS) void ragged_ranges_kernel(const int *nvalid, int G, int T, int pitch, int tpg,
                                                                       int *ranges, int *order) {
    __shared__ long long incl[RANGE_THREADS];
    const int tid = threadIdx.x, total = G * tpg;
    const int chunk = (total + RANGE_THREADS - 1) / RANGE_THREADS;
    const int t0 = min(total, tid * chunk), t1 = min(total, t0 + chunk);
    long long c = 0;
    if (t0 < t1) {
        TileWalk w;
        w.start(t0, tpg, T, pitch, nvalid);
        for (int t = t0; t < t1; ++t) {
            c += w.cost(T, pitch);
            w.next(tpg, T, pitch, nvalid, G);
        }
    }
    // inclusive scan of the chunk costs: within the wave by shuffles, across the 16 waves through their totals (two barriers
    // instead of the twenty of a shared-memory doubling scan; integer sums: the same bounds bit for bit)
    __shared__ long long wave_tot[RANGE_THREADS / 64];
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const long long up = __shfl_up(c, o);
        if (lane >= o) c += up;
    }
    if (lane == 63) wave_tot[wv] = c;
    __syncthreads();
    long long before = 0;
    for (int w = 0; w < wv; ++w) before += wave_tot[w];
    incl[tid] = c + before;
    __syncthreads();
    const long long all = incl[RANGE_THREADS - 1];
    for (int b = tid; b <= FGNN_RANGE_WG; b += RANGE_THREADS) {
        if (b == FGNN_RANGE_WG) {
            ranges[b] = total;
            continue;
        }
        const long long target = all * b / FGNN_RANGE_WG;
        int lo = 0, hi = RANGE_THREADS - 1;                // first chunk whose inclusive sum exceeds the target
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (incl[mid] > target) hi = mid;
            else lo = mid + 1;
        }
        long long acc = lo > 0 ? incl[lo - 1] : 0;
        int t = min(total, lo * chunk);
        const int te = min(total, t + chunk);
        if (t < te) {
            TileWalk w;
            w.start(t, tpg, T, pitch, nvalid);
            for (; t < te; ++t) {
                const int ct = w.cost(T, pitch);
                if (acc + ct > target) break;
                acc += ct;
                w.next(tpg, T, pitch, nvalid, G);
            }
        }
        ranges[b] = t;
    }
    if (order) {        // rank sort by (nvalid descending, index ascending): G^2 / 1024 comparisons per thread
        for (int g = tid; g < G; g += RANGE_THREADS) {
            const int nv = nvalid[g];
            int rank = 0;
            for (int o = 0; o < G; ++o) {
                const int no = nvalid[o];
                rank += (no > nv || (no == nv && o < g)) ? 1 : 0;
            }
            order[rank] = g;
        }
    }
}
}  // namespace

extern "C" int fgnn_ragged_tile_ranges_order(const int *nvalid, int G, int N, int *ranges, int *order, void *stream) {
    FGNN_CHECK(nvalid && ranges && G > 0 && N > 0, "fgnn_ragged_tile_ranges: bad arguments");
    const long long tpg = fgnn_tiles_per_graph(N);
    FGNN_CHECK(G * tpg < (1ll << 30), "fgnn_ragged_tile_ranges: too many tiles");
    hipLaunchKernelGGL(ragged_ranges_kernel, dim3(1), dim3(RANGE_THREADS), 0, (hipStream_t)stream, nvalid, G, FGNN_TILE, N,
                       (int)tpg, ranges, order);
    FGNN_LAUNCH_CHECK();
    return 0;
}
extern "C" int fgnn_ragged_tile_ranges(const int *nvalid, int G, int N, int *ranges, void *stream) {
    return fgnn_ragged_tile_ranges_order(nvalid, G, N, ranges, nullptr, stream);
}

extern "C" int fgnn_ragged_tile_ranges16(const int *nvalid, int G, int N, int ldr, int *ranges, void *stream) {
    FGNN_CHECK(nvalid && ranges && G > 0 && N > 0 && ldr >= N, "fgnn_ragged_tile_ranges16: bad arguments");
    const long long tpg = fgnn_tiles_per_graph16(N, ldr);
    FGNN_CHECK(G * tpg < (1ll << 30), "fgnn_ragged_tile_ranges16: too many tiles");
    hipLaunchKernelGGL(ragged_ranges_kernel, dim3(1), dim3(RANGE_THREADS), 0, (hipStream_t)stream, nvalid, G, 64, ldr, (int)tpg,
                       ranges, (int *)nullptr);
    FGNN_LAUNCH_CHECK();
    return 0;
}

// Shared device helpers for the gfx950 (CDNA4, wave64) FGNN kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fgnn_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DEVI __device__ __forceinline__
#define WAVE 64

// v_mfma_f32_32x32x2_f32: D(32x32) = A(32x2) * B(2x32) + C, exact fp32 fma chain.
//   lane l supplies a = A[l&31][l>>5], b = B[l>>5][l&31];
//   D[row][col]: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
DEVI f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Row (channel) held by accumulator register r in the half-wave h = lane>>5.
// Using these 16 registers as the next layer's B operand makes k-step r contract
// channels {ch_of(r,0), ch_of(r,1)} -- the conv chain never leaves the register file.
DEVI constexpr int ch_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int CTRL>
DEVI float dpp_mov(float v) {
    return __builtin_bit_cast(float,
        __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// Sum over the 32 lanes that share h = lane>>5; every lane of the half gets the total.
DEVI float half_sum(float v) {
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
    v += dpp_mov<0x140>(v);   // row_mirror
    // xor 16 inside each 32-lane group: ds_swizzle bit-mode and=0x1f, or=0, xor=0x10
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
    return v;
}

// Sum / max over all 64 lanes (result in every lane).
DEVI float wave_sum(float v) {
    v = half_sum(v);
    v += __shfl_xor(v, 32);
    return v;
}
DEVI float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ---- buffer-descriptor addressing of a (G, C, ld) fp32 tensor -------------------------
// One VGPR byte offset per lane per tile (pixel + half-wave row part), the per-row part in
// an SGPR soffset: no 64-bit per-load address VGPRs.  Out-of-range lanes use OOB_OFF, for
// which loads return 0 and stores are dropped (tensors are checked to be < 2 GiB).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int OOB_OFF = (int)0x80000000;

struct View {
    rsrc_t r;
    int gs4, ld4;     // byte strides between graphs / channels
};

DEVI View make_view(const float *p, long long gstride, long long ld, int G) {
    View v;
    long long bytes = (long long)G * gstride * 4;
    if (bytes > 0x7fffffffll) bytes = 0x7fffffffll;
    v.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, (int)bytes, 0x00020000);
    v.gs4 = (int)(gstride * 4);
    v.ld4 = (int)(ld * 4);
    return v;
}
DEVI float buf_load(const View &v, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(v.r, voff, soff, 0));
}
DEVI void buf_store(float x, const View &v, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), v.r, voff, soff, 0);
}

// A 2-channel slab expanded on the fly from the bit-packed adjacency (loaders/data_generator.py:118-125): half-wave 0 holds
// channel 0 = W[i][jj] (bit jj of row i), half-wave 1 channel 1 = [i == jj] * deg[i].  Bits past the valid vertices are not
// masked here: a padding pixel only feeds padding outputs, which every consumer masks (fgnn_adjacency_degree does mask).
struct PackedSrc {
    rsrc_t bits, deg;
    int N, words;
};
DEVI PackedSrc make_packed_src(const unsigned *bits, const float *deg, int G, int N) {
    PackedSrc ps;
    ps.N = N;
    ps.words = (N + 31) / 32;
    long long nb = (long long)G * N * ps.words * 4, nd = (long long)G * N * 4;
    if (nb > 0x7fffffffll) nb = 0x7fffffffll;
    ps.bits = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(bits), 0, (int)nb, 0x00020000);
    ps.deg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(deg), 0, (int)nd, 0x00020000);
    return ps;
}
template <class Ctx>
DEVI void load_packed(float (&x)[1], const PackedSrc &ps, const Ctx &c, int h) {
    const int ob = (c.inb && h == 0) ? (c.i * ps.words + (c.jj >> 5)) * 4 : OOB_OFF;
    const int od = (c.inb && h == 1 && c.i == c.jj) ? c.i * 4 : OOB_OFF;
    const unsigned w = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(ps.bits, ob, c.g * ps.N * ps.words * 4, 0);
    const float d = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ps.deg, od, c.g * ps.N * 4, 0));
    x[0] = h == 0 ? (((w >> (c.jj & 31)) & 1u) ? 1.f : 0.f) : d;
}
// max(x, 0) as ONE instruction (v_max_i32 on the bit pattern: negative floats are negative ints).
// fmaxf / a float compare-select both become TWO v_max_f32 (operand canonicalisation for sNaN first).
DEVI float relu1(float x) {
    const int i = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, i > 0 ? i : 0);
}

DEVI int nvalid_of(const int *nvalid, int g, int N) { return nvalid ? nvalid[g] : N; }

// Workgroups are dispatched round-robin over the 8 XCDs, each with its own L2.  Kernels whose consecutive logical
// workgroups share per-graph data (the 32 channel matrices of one graph read the same rows of tile statistics) remap
// blockIdx so that consecutive logical indices run on ONE XCD; otherwise all eight L2s fetch the same lines
// (PMC: +20 MB per launch of the forward matmul with finalize at G = 64, N = 50).  A pure performance mapping: any
// dispatch order gives the same results.
DEVI int xcd_swizzle(int b, int n) {
    const int n8 = n & ~7;
    return b < n8 ? (b & 7) * (n8 >> 3) + (b >> 3) : b;
}

// ---- ragged batches: tiles inside the padding ------------------------------------------------------------------------
// A tile = T consecutive elements of a plane stored with row pitch `pitch` (fp32 slabs: T = FGNN_TILE, pitch = N; bf16
// slabs: T = 64, pitch = ldr).  Does tile tt hold an element of the valid nv x nv corner?
DEVI bool tile_live_p(int tt, int T, int pitch, int nv) {
    const int p0 = tt * T, p1 = p0 + T - 1;
    const int i0 = p0 / pitch, j0 = p0 - i0 * pitch, i1 = p1 / pitch;
    return (i0 < nv && j0 < nv) || (i1 > i0 && i0 + 1 < nv);
}
// first live tile among t, t + step, t + 2 step, ... below t_end (t_end if none); step is a power of two.
// All operands are wave-uniform.  Rows >= nv are padding up to the end of the graph, so that tail is jumped over.
DEVI int next_live_tile_p(int t, int t_end, int step, int tpg, int T, int pitch, const int *nvalid) {
    while (t < t_end) {
        const int g = t / tpg, tt = t - g * tpg;
        const int nv = nvalid[g];
        if (tile_live_p(tt, T, pitch, nv)) break;
        if (tt * T / pitch >= nv) t += ((g + 1) * tpg - t + step - 1) & ~(step - 1);
        else t += step;
    }
    return t;
}
// Round-robin over the LIVE tiles of a range: the next tile at or after `t` (below t_end, else t_end) whose index among the range's
// live tiles is == who (mod nwho); `cnt` = live tiles seen so far, carried between calls (start: t = the range's first tile, cnt = 0).
// Dealing raw tile indices round-robin and skipping the dead ones leaves the owners of a ragged batch with up to 40 % different
// shares (runs of live tiles alternate with runs of padding in every row); this keeps them within one tile of each other at the
// price of every owner looking at every tile of the range.  All operands wave-uniform.
DEVI int next_owned_live_tile_p(int t, int t_end, int &cnt, int who, int nwho, int tpg, int T, int pitch, const int *nvalid) {
    while (t < t_end) {
        const int g = t / tpg, tt = t - g * tpg;
        const int nv = nvalid[g];
        if (tile_live_p(tt, T, pitch, nv)) {
            const bool mine = cnt % nwho == who;
            ++cnt;
            if (mine) break;
            ++t;
        } else if (tt * T / pitch >= nv) {
            t = (g + 1) * tpg;             // rows >= nv are padding up to the end of the graph
        } else {
            ++t;
        }
    }
    return t < t_end ? t : t_end;
}
// The same dealing without every owner scanning the range: the workgroup's live tiles of [t0, t1), in order, as a list in LDS (built
// by all `nthreads` threads: one tile per thread and pass, two barriers per pass; the last one also publishes the list).  Returns the
// number of live tiles (uniform); entries beyond `cap` are not stored -- the caller falls back to next_owned_live_tile_p then.
// scratch: nthreads / 64 ints.
constexpr int LIVE_LIST_CAP = 2040;
DEVI int build_live_list(int *list, int *scratch, int t0, int t1, int tpg, int T, int pitch, const int *nvalid, int tid, int nthreads) {
    int base = 0;
    const int wv = tid >> 6, lane = tid & 63, nw = nthreads >> 6;
    for (int c0 = t0; c0 < t1; c0 += nthreads) {
        const int t = c0 + tid;
        bool live = false;
        if (t < t1) {
            const int g = t / tpg, tt = t - g * tpg;
            live = tile_live_p(tt, T, pitch, nvalid[g]);
        }
        const unsigned long long m = __ballot(live);
        if (lane == 0) scratch[wv] = __popcll(m);
        __syncthreads();
        int off = base, tot = 0;
        for (int w = 0; w < nw; ++w) {
            const int c = scratch[w];
            off += w < wv ? c : 0;
            tot += c;
        }
        if (live) {
            const int idx = off + __popcll(m & ((1ull << lane) - 1ull));
            if (idx < LIVE_LIST_CAP) list[idx] = t;
        }
        base += tot;
        __syncthreads();
    }
    return base;
}
DEVI bool tile_live(int tt, int N, int nv) { return tile_live_p(tt, FGNN_TILE, N, nv); }
DEVI int next_live_tile(int t, int t_end, int step, int tpg, int N, const int *nvalid) {
    return next_live_tile_p(t, t_end, step, tpg, FGNN_TILE, N, nvalid);
}

// Static priority for the younger half of a workgroup's waves (the loser of every arbitration for a SIMD's issue slots, MI355X_MICROARCH.md
// "two waves per SIMD"): measurement switch FGNN_YP (bit mask per kernel family, see the call sites); 0 = off
#ifndef FGNN_YP
#define FGNN_YP 0
#endif
DEVI void young_prio(int bit, int wv, int nw) {
    if ((FGNN_YP >> bit) & 1) {
        if (wv >= nw / 2) __builtin_amdgcn_s_setprio(1);
    }
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: what has been raised is remembered per
// (launcher, device), so a process that drives a second GPU raises it there as well.
struct LdsAttrCache {
    size_t raised[32] = {};
};
inline bool fgnn_raise_lds(LdsAttrCache &c, const void *kernel, size_t lds) {
    if (lds <= 64 * 1024) return true;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32)
        return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
    if (lds <= c.raised[dev]) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
    c.raised[dev] = lds;
    return true;
}

// error plumbing shared by the launchers
void fgnn_set_error(const char *fmt, ...);
#define FGNN_CHECK(cond, ...)                       \
    do {                                            \
        if (!(cond)) {                              \
            fgnn_set_error(__VA_ARGS__);            \
            return 1;                               \
        }                                           \
    } while (0)
#define FGNN_LAUNCH_CHECK()                                                        \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            fgnn_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,          \
                           hipGetErrorString(e_));                                 \
            return 2;                                                              \
        }                                                                          \
    } while (0)

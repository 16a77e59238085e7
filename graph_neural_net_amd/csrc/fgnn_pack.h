// LDS operand images of the MLP kernels: layout + element definition shared by the kernels
// (compile-time) and by fgnn_pack_operands (run-time, once per step per MLP).
// The weight part of an image is [k-step/4][lane][4] floats: element (step t, lane l) at
// (t>>2)*256 + l*4 + (t&3).  Biases follow as a compact tail [layer][half-wave h][16] (the value a lane
// needs, bias[ch_of(r, h)], depends on h and r only: a broadcast ds_read_b128 instead of 16 steps).
#pragma once
#include "fgnn_common.h"

#define HD __host__ __device__ __forceinline__

HD constexpr int pk_pad4(int x) { return (x + 3) & ~3; }
HD constexpr int pk_slab_ch(int S, int k, int h) { return S == 16 ? (k & 3) + 8 * (k >> 2) + 4 * h : 2 * k + h; }
HD constexpr int pk_ch(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- forward image of ONE MLP: [W0 slab a | W0 slab b | W_1..W_{d-1}] + bias tail b_0..b_{d-1} ----
struct PkFwd {
    int off_w1a, off_w1b, off_wh, steps, bias_f, nbias, floats;
};
HD constexpr PkFwd pk_fwd(int ca, int cb, int depth) {
    PkFwd p{};
    p.off_w1a = 0;
    p.off_w1b = pk_pad4(ca / 2);
    p.off_wh = p.off_w1b + pk_pad4(cb / 2);
    p.steps = p.off_wh + 16 * (depth - 1);
    p.bias_f = p.steps * 64;
    p.nbias = depth;
    p.floats = p.bias_f + 32 * p.nbias;
    return p;
}
// element idx (= layer*32 + h*16 + r) of a bias tail
HD float pk_bias_value(const float *const *bias, int idx) { return bias[idx >> 5][pk_ch(idx & 15, (idx >> 4) & 1)]; }
// W[l]: conv weights (32, Cin_l) row-major, bias[l]: (32)
HD float pk_fwd_value(const PkFwd &p, int ca, int cb, const float *const *W, int t, int l) {
    const int jj = l & 31, hh = l >> 5, cin = ca + cb;
    if (t < p.off_w1b) {
        const int s = t - p.off_w1a;
        return s < ca / 2 ? W[0][jj * cin + pk_slab_ch(ca / 2, s, hh)] : 0.f;
    }
    if (t < p.off_wh) {
        const int s = t - p.off_w1b;
        return s < cb / 2 ? W[0][jj * cin + ca + pk_slab_ch(cb / 2, s, hh)] : 0.f;
    }
    const int u = t - p.off_wh;
    return W[1 + (u >> 4)][jj * FGNN_H + pk_ch(u & 15, hh)];
}

// ---- backward image: [W0 a | W0 b | W_1..W_{d-2} (fwd) | W_1^T..W_{d-1}^T | W0^T a | W0^T b] + bias tail b_0..b_{d-2} ----
struct PkBwd {
    int off_w1a, off_w1b, off_wh, off_wt, off_wt0a, off_wt0b, steps, bias_f, nbias, floats;
};
HD constexpr PkBwd pk_bwd(int ca, int cb, int depth) {
    PkBwd p{};
    p.off_w1a = 0;
    p.off_w1b = pk_pad4(ca / 2);
    p.off_wh = p.off_w1b + pk_pad4(cb / 2);
    p.off_wt = p.off_wh + 16 * (depth > 2 ? depth - 2 : 0);
    p.off_wt0a = p.off_wt + 16 * (depth > 1 ? depth - 1 : 0);
    p.off_wt0b = p.off_wt0a + 16;
    p.steps = p.off_wt0b + (cb > 0 ? 16 : 0);
    p.bias_f = p.steps * 64;
    p.nbias = depth > 1 ? depth - 1 : 0;
    p.floats = p.bias_f + 32 * p.nbias;
    return p;
}
HD float pk_bwd_value(const PkBwd &p, int ca, int cb, const float *const *W, int t, int l) {
    const int jj = l & 31, hh = l >> 5, cin = ca + cb;
    if (t < p.off_w1b) {
        const int s = t - p.off_w1a;
        return s < ca / 2 ? W[0][jj * cin + pk_slab_ch(ca / 2, s, hh)] : 0.f;
    }
    if (t < p.off_wh) {
        const int s = t - p.off_w1b;
        return s < cb / 2 ? W[0][jj * cin + ca + pk_slab_ch(cb / 2, s, hh)] : 0.f;
    }
    if (t < p.off_wt) {
        const int u = t - p.off_wh;
        return W[1 + (u >> 4)][jj * FGNN_H + pk_ch(u & 15, hh)];
    }
    if (t < p.off_wt0a) {
        const int u = t - p.off_wt;
        return W[1 + (u >> 4)][pk_ch(u & 15, hh) * FGNN_H + jj];
    }
    if (t < p.off_wt0b) {
        const int u = t - p.off_wt0a;
        return jj < ca ? W[0][pk_ch(u, hh) * cin + jj] : 0.f;
    }
    const int u = t - p.off_wt0b;
    return jj < cb ? W[0][pk_ch(u, hh) * cin + ca + jj] : 0.f;
}

// ---- the same images for the 16-pixel-tile kernels (fgnn_t16.h, v_mfma_f32_16x16x4_f32) --------------------------------------
// Same containers (PkFwd / PkBwd offsets, [t / 4][lane][4]); a "step" t of a 32-channel set is now (k-step s = t >> 1, row
// block b = t & 1): lane (m = l & 15, k = l >> 4) holds Mx[row pkt_row(b, m)][contracted channel pkt_chan(s, k)].  A
// 2-channel slab is one k-step: t = b, lanes k = 0, 1 carry channels 0, 1, the rest zeros.
HD constexpr int pkt_chan(int s, int q) { return 8 * (s >> 1) + 2 * (s & 1) + (q >> 1) + 4 * (q & 1); }
HD constexpr int pkt_row(int b, int m) { return pkt_chan(4 * b + (m & 3), m >> 2); }
// contracted channel of step t, lane quarter k, for a slab of c channels (-1: none)
HD constexpr int pkt_in(int c, int t, int k) { return c == 32 ? pkt_chan(t >> 1, k) : ((t >> 1) == 0 && k < c ? k : -1); }
HD float pkt_bias_value(const float *const *bias, int idx) { return bias[idx >> 5][pkt_chan(idx & 7, (idx >> 3) & 3)]; }
HD float pkt_fwd_value(const PkFwd &p, int ca, int cb, const float *const *W, int t, int l) {
    const int m = l & 15, k = l >> 4, cin = ca + cb;
    if (t < p.off_w1b) {
        const int u = t - p.off_w1a, c = u < (ca == 32 ? 16 : 2) ? pkt_in(ca, u, k) : -1;
        return c >= 0 ? W[0][pkt_row(u & 1, m) * cin + c] : 0.f;
    }
    if (t < p.off_wh) {
        const int u = t - p.off_w1b, c = u < (cb == 32 ? 16 : 2) ? pkt_in(cb, u, k) : -1;
        return c >= 0 ? W[0][pkt_row(u & 1, m) * cin + ca + c] : 0.f;
    }
    const int u = t - p.off_wh, v = u & 15;
    return W[1 + (u >> 4)][pkt_row(v & 1, m) * FGNN_H + pkt_chan(v >> 1, k)];
}
HD float pkt_bwd_value(const PkBwd &p, int ca, int cb, const float *const *W, int t, int l) {
    const int m = l & 15, k = l >> 4, cin = ca + cb;
    if (t < p.off_w1b) {
        const int u = t - p.off_w1a, c = u < (ca == 32 ? 16 : 2) ? pkt_in(ca, u, k) : -1;
        return c >= 0 ? W[0][pkt_row(u & 1, m) * cin + c] : 0.f;
    }
    if (t < p.off_wh) {
        const int u = t - p.off_w1b, c = u < (cb == 32 ? 16 : 2) ? pkt_in(cb, u, k) : -1;
        return c >= 0 ? W[0][pkt_row(u & 1, m) * cin + ca + c] : 0.f;
    }
    if (t < p.off_wt) {
        const int u = t - p.off_wh, v = u & 15;
        return W[1 + (u >> 4)][pkt_row(v & 1, m) * FGNN_H + pkt_chan(v >> 1, k)];
    }
    if (t < p.off_wt0a) {       // W_l^T: row = input channel of W_l, contracted = its output channel
        const int u = t - p.off_wt, v = u & 15;
        return W[1 + (u >> 4)][pkt_chan(v >> 1, k) * FGNN_H + pkt_row(v & 1, m)];
    }
    if (t < p.off_wt0b) {
        const int v = t - p.off_wt0a, r = pkt_row(v & 1, m);
        return r < ca ? W[0][pkt_chan(v >> 1, k) * cin + r] : 0.f;
    }
    const int v = t - p.off_wt0b, r = pkt_row(v & 1, m);
    return r < cb ? W[0][pkt_chan(v >> 1, k) * cin + ca + r] : 0.f;
}

// The same copy split in two halves, so that a prologue can put the image loads in flight FIRST and do its other
// (dependent) loads and arithmetic before the values are needed: one memory round trip instead of two.
template <int N4, int NTHREADS>
struct PkRegs {
    static constexpr int PER = (N4 + NTHREADS - 1) / NTHREADS;
    float4 v[PER];
};
template <int N4, int NTHREADS>
DEVI void pk_load_regs(PkRegs<N4, NTHREADS> &r, const float *packed) {
    // buffer loads: unlike plain global loads the compiler neither re-materialises them at their use nor sinks
    // them.  (Dword loads: with a constant descriptor this compiler folds raw_buffer_load_b128 into one dword.)
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(packed), 0, N4 * 16, 0x00020000);
#pragma unroll
    for (int k = 0; k < PkRegs<N4, NTHREADS>::PER; ++k) {
        const int off = (threadIdx.x + k * NTHREADS) * 16;                                // past the end: returns 0
        r.v[k].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
        r.v[k].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 4, 0));
        r.v[k].z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 8, 0));
        r.v[k].w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 12, 0));
    }
}
template <int N4, int NTHREADS>
DEVI void pk_store_regs(float *lds, const PkRegs<N4, NTHREADS> &r) {
    float4 *dst = reinterpret_cast<float4 *>(lds);
#pragma unroll
    for (int k = 0; k < PkRegs<N4, NTHREADS>::PER; ++k) {
        const int e = threadIdx.x + k * NTHREADS;
        if (e < N4) dst[e] = r.v[k];
    }
}

// ---- the image's way into LDS without registers (round 6, the *_t16 kernels): global_load_lds_dwordx4, 1 KiB per wave instruction ----
// The image region in LDS and the packed buffer are both padded to whole KiB (pk_pad_floats): a wave instruction moves 64 x 16 B, the
// destination is wave-uniform base + lane x 16.  Nothing to store afterwards: the loads land in LDS; the prologue's barrier
// (__syncthreads waits vmcnt(0)) orders them before the first read.
HD constexpr int pk_pad_floats(int floats) { return (floats + 255) & ~255; }
template <int NWAVES>
DEVI void pk_glds(float *lds_dst, const float *packed, int padded_floats, int wv, int lane) {
    const int chunks = padded_floats / 256;
    for (int c = wv; c < chunks; c += NWAVES)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(packed + c * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void *)(unsigned long long)(unsigned)(unsigned long long)(lds_dst + c * 256),
                                         16, 0, 0);
}

// ---- one fgnn_pack_job (include/fgnn_hip.h): the LDS operand image(s) of one MLP kernel launch, written by `nbx` blocks of 256 threads.
// Shared by fgnn_pack_operands (norm.hip) and by the first launch of the structured block 1, which carries the step's packing as
// extra workgroups (block1_struct.hip: one launch less per step)
struct PackJobs {
    fgnn_pack_job job[FGNN_MAX_PACK_JOBS];
};
DEVI void pack_job_body(const fgnn_pack_job &jb, const int bx, const int nbx, const int tid) {
    const bool t16 = jb.kind >= 4;          // kinds 4 / 5: the 16-pixel-tile images of kinds 0 / 1
    if (jb.kind == 0 || jb.kind == 4) {
        const PkFwd p = pk_fwd(jb.ca, jb.cb, jb.depth);
        const int per = p.floats;
        for (int e = bx * 256 + tid; e < per * jb.nmlp; e += nbx * 256) {
            const int m = e / per, r = e - m * per;
            if (r < p.bias_f) {
                const int t = r >> 6, l = r & 63;
                jb.out[m * per + (t >> 2) * 256 + l * 4 + (t & 3)] =
                    t16 ? pkt_fwd_value(p, jb.ca, jb.cb, jb.W[m], t, l) : pk_fwd_value(p, jb.ca, jb.cb, jb.W[m], t, l);
            } else {
                jb.out[m * per + r] = t16 ? pkt_bias_value(jb.bias[m], r - p.bias_f) : pk_bias_value(jb.bias[m], r - p.bias_f);
            }
        }
    } else {
        const PkBwd p = pk_bwd(jb.ca, jb.cb, jb.depth);
        for (int e = bx * 256 + tid; e < p.floats; e += nbx * 256) {
            if (e < p.bias_f) {
                const int t = e >> 6, l = e & 63;
                jb.out[(t >> 2) * 256 + l * 4 + (t & 3)] =
                    t16 ? pkt_bwd_value(p, jb.ca, jb.cb, jb.W[0], t, l) : pk_bwd_value(p, jb.ca, jb.cb, jb.W[0], t, l);
            } else {
                jb.out[e] = t16 ? pkt_bias_value(jb.bias[0], e - p.bias_f) : pk_bias_value(jb.bias[0], e - p.bias_f);
            }
        }
    }
}
constexpr int PACK_BLOCKS_PER_JOB = 48;

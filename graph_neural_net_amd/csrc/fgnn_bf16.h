// Shared device helpers of the bf16 kernels (gfx950): v_mfma_f32_32x32x16_bf16 fragments, packed
// conversions, the MFMA transposer and the LDS operand-image layout of the bf16 MLP kernels.
//
// Fragment conventions (lane = (j, h), j = lane & 31, h = lane >> 5):
//   * an MFMA A or B operand of one k-step is 4 dwords = 8 bf16; dword d holds k-slots 2d (low half) and
//     2d+1 (high half) of the lane's half-wave: k = 8h + s.  A[i][k] lives in lane i = j, B[k][n] in lane n = j,
//     i.e. both operands have the SAME register layout (lane = the non-contracted index).
//   * a "normal" activation fragment F16 holds 32 channels of the lane's pixel: step t (0,1), slot s (0..7)
//     <-> channel ch_of(8t + s, h).  That is exactly the D fragment of a conv (lane = pixel column,
//     register r <-> row ch_of(r, h)) packed pairwise, so a layer's output feeds the next layer's B operand
//     without leaving the register file (same trick as the fp32 kernels).
//   * a "transposed" fragment holds 32 pixels of the lane's channel: slot (t, s) <-> pixel ch_of(8t + s, h)
//     of the 32-pixel group.  The weight-gradient GEMMs contract over pixels and take both operands in this
//     form.  It is produced WITHOUT LDS by one identity-matrix MFMA pair (transpose16 below): the matrix pipe
//     is nearly idle in bf16, the LDS and VALU are not.
#pragma once
#include "fgnn_common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2_t __attribute__((ext_vector_type(2)));

struct F16 {
    unsigned d[8];
};

DEVI f32x16 mfma16(i32x4 a, i32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0,
                                                    0, 0);
}
DEVI i32x4 step_of(const F16 &f, int t) {
    i32x4 v;
    v[0] = (int)f.d[4 * t + 0];
    v[1] = (int)f.d[4 * t + 1];
    v[2] = (int)f.d[4 * t + 2];
    v[3] = (int)f.d[4 * t + 3];
    return v;
}
// {lo, hi} -> one dword of two bf16 (round to nearest even): v_cvt_pk_bf16_f32
DEVI unsigned cvt_pk(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
DEVI float bf_lo(unsigned d) { return __builtin_bit_cast(float, d << 16); }
DEVI float bf_hi(unsigned d) { return __builtin_bit_cast(float, d & 0xffff0000u); }
// max(x, 0) on both halves: v_pk_max_i16 (negative floats are negative int16)
DEVI unsigned relu_pk(unsigned d) {
    const s16x2_t z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, d), z));
}
// 0xffff in every half whose (non-negative) bf16 value is > 0
// (inline asm: the compiler canonicalises every C spelling of this into two compares and two selects per dword)
DEVI unsigned pos_mask_pk(unsigned h) {
    unsigned n, m;
    asm("v_pk_sub_i16 %0, 0, %1" : "=v"(n) : "v"(h));             // negative iff h > 0
    asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(m) : "v"(n));   // 0xffff / 0 (op_sel_hi: the inline constant's
                                                                              // low half is the shift of BOTH lanes)
    return m;
}
// low / high halves of two dwords -> one dword
DEVI unsigned pack_lo(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
DEVI unsigned pack_hi(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

DEVI void zero16f(f32x16 &a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}
// D fragment -> packed fragment (pairs of consecutive accumulator registers)
DEVI void pack_acc(F16 &f, const f32x16 &acc) {
#pragma unroll
    for (int q = 0; q < 8; ++q) f.d[q] = cvt_pk(acc[2 * q], acc[2 * q + 1]);
}
DEVI void pack_acc_relu(F16 &f, const f32x16 &acc) {
#pragma unroll
    for (int q = 0; q < 8; ++q) f.d[q] = relu_pk(cvt_pk(acc[2 * q], acc[2 * q + 1]));
}

// Identity operand of the transposer for lane (n, h): slot (t, s) = 1 iff ch_of(8t + s, h) == n.
DEVI F16 make_identity(int lane) {
    const int n = lane & 31, h = lane >> 5;
    F16 f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int t = q >> 2, s0 = 2 * (q & 3);
        const unsigned lo = ch_of(8 * t + s0, h) == n ? 0x3F80u : 0u;
        const unsigned hi = ch_of(8 * t + s0 + 1, h) == n ? 0x3F800000u : 0u;
        f.d[q] = lo | hi;
    }
    return f;
}
// X (normal fragment: lane = pixel p, slots = channels) -> D'[p][n] = X[p][n] as a D fragment: lane = channel n,
// register r <-> pixel ch_of(r, h).  Exact (every product is x * 1 or x * 0).  The same call transposes a
// transposed fragment back.
DEVI f32x16 transpose16(const F16 &x, const F16 &ident) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // inline-constant C operand
    f32x16 acc = mfma16(step_of(x, 0), step_of(ident, 0), zero);
    acc = mfma16(step_of(x, 1), step_of(ident, 1), acc);
    return acc;
}

// ---- buffer views of bf16 tensors: (G, C, ld) elements, byte strides -----------------------------------------------
struct View16 {
    rsrc_t r;
    int gs2, ld2;
};
DEVI View16 make_view16(const void *p, long long gstride, long long ld, int G) {
    View16 v;
    long long bytes = (long long)G * gstride * 2;
    if (bytes > 0x7fffffffll) bytes = 0x7fffffffll;
    v.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
    v.gs2 = (int)(gstride * 2);
    v.ld2 = (int)(ld * 2);
    return v;
}
DEVI unsigned buf_load_u32(const View16 &v, int voff, int soff) {
    return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(v.r, voff, soff, 0);
}
DEVI void buf_store_u32(unsigned x, const View16 &v, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(x, v.r, voff, soff, 0);
}

// ---- LDS operand images of the bf16 MLP kernels ------------------------------------------------------------------
// An image is a sequence of k-steps, [step][lane][4 dwords] (one ds_read_b128 per lane per step, conflict-free),
// followed by an fp32 bias tail.  Step content for lane (i, h), slot s:
//   conv layer l, slab with S channels   : W_l[i][chan(t, h, s)]          (A operand of the conv, B operand of its
//                                                                          swapped form)
//   transposed layer l                   : W_l[chan(t, h, s)][i]          (A operand of the dgrad)
//   chan(t, h, s) = ch_of(8t + s, h) for 32-channel slabs / hidden layers, = s (h == 0, s < 2) for the 2-channel
//   input slab (one zero-padded k-step).
#define HD16 __host__ __device__ __forceinline__
HD16 constexpr int pk16_steps(int c) { return c == 0 ? 0 : (c <= 16 ? 1 : c / 16); }
HD16 constexpr int pk16_ch(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// channel of slot (t, h, s) in a slab of C channels; -1 = zero padding
HD16 constexpr int pk16_chan(int C, int t, int h, int s) { return C >= 32 ? pk16_ch(8 * t + s, h) : ((h == 0 && s < C) ? s : -1); }

struct Pk16 {
    int off_w0a, off_w0b, off_wh, off_wt, off_wt0a, off_wt0b, steps, bias_f, nbias, floats;
};
// kind 0 (forward, one MLP): [W0 a | W0 b | W_1 .. W_{d-1}], bias tail b_0 .. b_{d-1}: compact [layer][h][16] then
//                            plain [layer][32]
// kind 1 (backward)        : [W0 a | W0 b | W_1 .. W_{d-2} | W_{d-1}^T .. W_1^T | W0^T a | W0^T b], tail b_0 .. b_{d-2}
//                            (compact only)
HD16 constexpr Pk16 pk16_layout(int kind, int ca, int cb, int depth) {
    Pk16 p{};
    p.off_w0a = 0;
    p.off_w0b = pk16_steps(ca);
    p.off_wh = p.off_w0b + pk16_steps(cb);
    if (kind == 0) {
        p.off_wt = p.off_wh + 2 * (depth - 1);
        p.off_wt0a = p.off_wt0b = p.steps = p.off_wt;
        p.nbias = depth;
        p.bias_f = p.steps * 256;                       // floats (dwords) before the tail
        p.floats = p.bias_f + 64 * p.nbias;
    } else {
        p.off_wt = p.off_wh + 2 * (depth > 2 ? depth - 2 : 0);
        p.off_wt0a = p.off_wt + 2 * (depth - 1);
        p.off_wt0b = p.off_wt0a + (ca >= 32 ? 2 : 0);
        p.steps = p.off_wt0b + (cb >= 32 ? 2 : 0);
        p.nbias = depth > 1 ? depth - 1 : 0;
        p.bias_f = p.steps * 256;
        p.floats = p.bias_f + 32 * p.nbias;
    }
    return p;
}
// value of slot s (0..7) of step `step` for lane l; W[l]: (32, Cin_l) row-major fp32
HD16 float pk16_value(int kind, const Pk16 &p, int ca, int cb, int depth, const float *const *W, int step, int l, int s) {
    const int i = l & 31, h = l >> 5, cin = ca + cb;
    if (step < p.off_w0b) {
        const int c = pk16_chan(ca, step - p.off_w0a, h, s);
        return c >= 0 ? W[0][i * cin + c] : 0.f;
    }
    if (step < p.off_wh) {
        const int c = pk16_chan(cb, step - p.off_w0b, h, s);
        return c >= 0 ? W[0][i * cin + ca + c] : 0.f;
    }
    if (step < p.off_wt) {
        const int u = step - p.off_wh;
        return W[1 + (u >> 1)][i * FGNN_H + pk16_ch(8 * (u & 1) + s, h)];
    }
    if (step < p.off_wt0a) {                            // W_{d-1}^T first, then W_{d-2}^T ...
        const int u = step - p.off_wt, layer = depth - 1 - (u >> 1);
        return W[layer][pk16_ch(8 * (u & 1) + s, h) * FGNN_H + i];
    }
    if (step < p.off_wt0b) {
        const int u = step - p.off_wt0a;
        return W[0][pk16_ch(8 * u + s, h) * cin + i];
    }
    const int u = step - p.off_wt0b;
    return W[0][pk16_ch(8 * u + s, h) * cin + ca + i];
}

// The eight slots of one (step, lane) of pk16_value with the step-range branch taken ONCE: eight unconditional (index-clamped)
// loads go out back to back, where eight inlined pk16_value calls each wait for their own load at the end of their if-chain
// (86 s_waitcnt, about eight memory round trips per entry).
DEVI void pk16_values8(int kind, const Pk16 &p, int ca, int cb, int depth, const float *const *W, int step, int l, float (&v)[8]) {
    const int i = l & 31, h = l >> 5, cin = ca + cb;
    if (step < p.off_wh) {                               // forward layer 0, slab a or b: channels may be padding (c < 0)
        const bool second = step >= p.off_w0b;
        const float *w = W[0] + i * cin + (second ? ca : 0);
        int c[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) c[s] = second ? pk16_chan(cb, step - p.off_w0b, h, s) : pk16_chan(ca, step - p.off_w0a, h, s);
#pragma unroll
        for (int s = 0; s < 8; ++s) v[s] = w[c[s] >= 0 ? c[s] : 0];
#pragma unroll
        for (int s = 0; s < 8; ++s) v[s] = c[s] >= 0 ? v[s] : 0.f;
    } else if (step < p.off_wt) {
        const int u = step - p.off_wh;
        const float *w = W[1 + (u >> 1)] + i * FGNN_H;
#pragma unroll
        for (int s = 0; s < 8; ++s) v[s] = w[pk16_ch(8 * (u & 1) + s, h)];
    } else if (step < p.off_wt0a) {                      // W_{d-1}^T first, then W_{d-2}^T ...
        const int u = step - p.off_wt, layer = depth - 1 - (u >> 1);
        const float *w = W[layer] + i;
#pragma unroll
        for (int s = 0; s < 8; ++s) v[s] = w[pk16_ch(8 * (u & 1) + s, h) * FGNN_H];
    } else {
        const bool second = step >= p.off_wt0b;
        const int u = step - (second ? p.off_wt0b : p.off_wt0a);
        const float *w = W[0] + (second ? ca : 0) + i;
#pragma unroll
        for (int s = 0; s < 8; ++s) v[s] = w[pk16_ch(8 * u + s, h) * cin];
    }
}


DEVI i32x4 lds_step(const float *wl, int step, int lane) {
    return reinterpret_cast<const i32x4 *>(wl)[step * 64 + lane];
}
// bias[ch_of(r, h)], r = 0..15, of one layer from the compact tail
DEVI void load_bias16(f32x16 &dst, const float *tail, int layer, int h) {
    const float4 *p = reinterpret_cast<const float4 *>(tail + layer * 32 + h * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = p[q];
        dst[4 * q + 0] = v.x;
        dst[4 * q + 1] = v.y;
        dst[4 * q + 2] = v.z;
        dst[4 * q + 3] = v.w;
    }
}

// ---- tiles of the bf16 MLP kernels: 64 consecutive elements of a channel = 32 pixel PAIRS, one pair per lane column ----
// A channel of graph g is N rows of `ldr` elements (ldr = N rounded up to 8, rows 16-byte aligned); element p = i*ldr + jj
// is valid iff i < nv and jj < nv.
struct Tile16 {
    int g, tt, pp, i, jj;
    bool inb;
};
DEVI Tile16 decode16(int tile, bool active, int tpg, int ldr, int PP, int j) {
    Tile16 c;
    c.g = __builtin_amdgcn_readfirstlane(active ? tile / tpg : 0);
    c.tt = active ? tile - c.g * tpg : 0;
    c.pp = c.tt * 32 + j;
    const int p0 = 2 * c.pp;
    c.inb = active && p0 < PP;
    c.i = p0 / ldr;
    c.jj = p0 - c.i * ldr;
    return c;
}
// per-lane byte offset of the pixel pair in the half-wave's first row
template <int HMUL>
DEVI int lane_off16(const View16 &v, const Tile16 &c, int h) {
    return c.inb ? HMUL * h * v.ld2 + 4 * c.pp : OOB_OFF;
}
// rows ch_of(r, h), r = 0..15, of a (G, 32, ld) bf16 tensor: one dword (pixel pair) each
DEVI void load_rows16b(unsigned (&x)[16], const View16 &v, const Tile16 &c, int h) {
    const int voff = lane_off16<4>(v, c, h);
    const int s0 = c.g * v.gs2;
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = buf_load_u32(v, voff, s0 + ((r & 3) + 8 * (r >> 2)) * v.ld2);
}
// input slab of C channels: 32 -> 16 dwords (rows ch_of(r,h)); 2 -> rows 0, 1 in x[0], x[1] (valid for h == 0 only)
template <int C>
DEVI void load_slab16(unsigned (&x)[C >= 32 ? 16 : 2], const View16 &v, const Tile16 &c, int h) {
    if constexpr (C >= 32) {
        load_rows16b(x, v, c, h);
    } else {
        const int voff = (c.inb && h == 0) ? 4 * c.pp : OOB_OFF;
        const int s0 = c.g * v.gs2;
        x[0] = buf_load_u32(v, voff, s0);
        x[1] = buf_load_u32(v, voff, s0 + v.ld2);
    }
}

// ---- one fgnn_pack_job of the 16-bit kernel set (fgnn_pack16_operands): one thread per (step, lane) writes the lane's 4 dwords; the tail
// is plain fp32.  Shared by misc16.hip's launch and by the structured block 1's first launch, which can carry the packing as extra
// workgroups (block1_struct.hip)
struct Pack16Jobs {
    fgnn_pack_job job[FGNN_MAX_PACK_JOBS];
};
constexpr int PACK16_BLOCKS_PER_JOB = 32;
__device__ __forceinline__ void pack16_job_body(const fgnn_pack_job &jb, const int bx, const int nbx, const int tid) {
    const Pk16 p = pk16_layout(jb.kind, jb.ca, jb.cb, jb.depth);
    const int nm = jb.kind == 0 ? jb.nmlp : 1;
    unsigned *out = reinterpret_cast<unsigned *>(jb.out);
    for (int m = 0; m < nm; ++m) {
        unsigned *om = out + (long long)m * p.floats;
        const float *const *W = jb.W[m];
        const float *const *Bv = jb.bias[m];
        for (int e = bx * 256 + tid; e < p.steps * 64; e += nbx * 256) {
            const int step = e >> 6, l = e & 63;
            uint4 v;
            unsigned d[4];
            float w8[8];
            pk16_values8(jb.kind, p, jb.ca, jb.cb, jb.depth, W, step, l, w8);
#pragma unroll
            for (int q = 0; q < 4; ++q) d[q] = cvt_pk(w8[2 * q], w8[2 * q + 1]);
            v.x = d[0];
            v.y = d[1];
            v.z = d[2];
            v.w = d[3];
            reinterpret_cast<uint4 *>(om)[e] = v;
        }
        float *tail = reinterpret_cast<float *>(om + p.bias_f);
        for (int e = bx * 256 + tid; e < 32 * p.nbias; e += nbx * 256) {
            const int layer = e >> 5, r = e & 15, h = (e >> 4) & 1;
            tail[e] = Bv[layer][pk16_ch(r, h)];                                   // compact [layer][h][16]
            if (jb.kind == 0) tail[32 * p.nbias + e] = Bv[layer][e & 31];         // plain [layer][32]
        }
    }
}

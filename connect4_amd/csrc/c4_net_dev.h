// c4_net_dev.h -- device side of the fused policy/value network (see c4_net.hip for the design notes):
// constants, the weight view NetDev, and net_forward_block(), the per-workgroup forward used by both
// the standalone kernel (c4_net.hip) and the fused self-play kernel (c4_engine.hip).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

namespace c4net {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int F = 32;                 // filters
constexpr int P = 16;                 // positions per workgroup
constexpr int PIX = 42;
constexpr int ROWS = P * PIX;         // 672
constexpr int TILES = ROWS / 32;      // 21
constexpr int CS = 40;                // halves per LDS row (32 channels + 8 pad => 80 B stride)
constexpr int KSTEPS = 18;            // 9 taps x 32 cin / 16
constexpr int NWAVES = 8;             // two waves per SIMD: one's LDS/epilogue hides under the other's MFMAs
constexpr int NTHREADS = NWAVES * 64;
constexpr int WCHUNKS = KSTEPS * 64;  // 16-byte A-fragment chunks per conv layer (18 KiB)
constexpr int HEADV = 3 * PIX;        // 126 head activations per position
constexpr int HSTR = 128;             // floats per position in the head scratch (16-byte aligned rows)
constexpr int VT_F4 = 11 * 64;        // value table: [11 groups of 4 inputs][64 lanes] float4
constexpr int PT_F = 11 * 64;         // policy table: [11][64] floats (lane = logit + 8*segment)
constexpr int MLP_F4 = VT_F4 + PT_F / 4 + 3 * 16;   // + fc_b, vout_w, pfc_b (64 floats each) = 928 float4 = 14,848 B
constexpr float LEAK = 0.01f;

struct NetDev {
    const half8 *stem_w;   // [3][64]           lane-ordered A fragments
    const float *stem_b;   // [32]
    const half8 *conv_w;   // [2R][18][64]
    const float *conv_b;   // [2R][32]
    const half8 *head_w;   // [2][64]           couts 0..2 = value, policy0, policy1
    const float *head_b;   // [4]
    const float4 *mlp;     // MLP_F4 float4s: value table [11][64][4], policy table [11][64], fc_b[64], vout_w[64], pfc_b[64]
    float vout_b, w1, w2;
    int n_res;
    int precise;           // reference-precision mode (C4_NET_F32X3)
    int filters;           // 32 or 64
    int mode;              // NETMODE_* the kernels are specialised for
    // 32 filters: the same weights in the fragment order of v_mfma_f32_16x16x32_f16 (net_forward_wave16) ...
    const half8 *stem_w16, *conv_w16, *head_w16;
    // ... and, for the reference-precision mode, their scaled low parts (net_forward_wave16q)
    const half8 *stem_w16l, *conv_w16l, *head_w16l;
    // ... and the tower as ONE linear stream for net_forward_wave16q: per tower tap T four fragments
    // [cout tile 0 hi][cout tile 0 lo][cout tile 1 hi][cout tile 1 lo] x 64 lanes x 16 bytes = 4,096 bytes
    const half8 *conv_w16p;
    unsigned conv_w16p_bytes;
    const half8 *w0;   // [W0_FRAGS][64]: the fragments a reference-precision pass needs first (stem hi/lo, heads hi/lo, tower taps 0 and 1)
    unsigned long long *stamps;   // diagnostic only (C4_NET_STAMPS=1): [wave][16] s_memtime values of block 0
};

// LeakyReLU = max(v, 0.01 v) (slope < 1).  Written as the bare instruction: fmaxf() makes the compiler canonicalise
// the MFMA's output first (a second v_max_f32 per value, a tenth of the wave forward's vector instructions); the
// result is the same for every non-NaN input, and a NaN never gets here (the engine counts and replaces bad answers).
__device__ __forceinline__ float lrelu(float v)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(LEAK * v));
    return r;
}

// DPP cross-lane moves (VALU speed; ds_bpermute-based __shfl costs an LDS round trip each)
template <int CTRL>
__device__ __forceinline__ float dppf(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float readlane_f(float v, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// sum / max over each aligned group of 8 lanes (quad_perm xor1, xor2, row_half_mirror)
__device__ __forceinline__ float sum8(float v) { v += dppf<0xB1>(v); v += dppf<0x4E>(v); v += dppf<0x141>(v); return v; }
__device__ __forceinline__ float max8(float v)
{
    v = fmaxf(v, dppf<0xB1>(v)); v = fmaxf(v, dppf<0x4E>(v)); v = fmaxf(v, dppf<0x141>(v));
    return v;
}
// sum over each 16-lane row (+ row_mirror)
__device__ __forceinline__ float sum16(float v) { v = sum8(v); v += dppf<0x140>(v); return v; }

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}

// epilogue of one 32-row tile.  Bias enters as the accumulator's initial value and the residual skip
// as two extra MFMAs against an identity matrix, so what is left is LeakyReLU + fp16 + 4 stores.
__device__ __forceinline__ void store_tile(const floatx16 &acc, _Float16 *dst, int rowoff, int h)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        half4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (_Float16)lrelu(acc[4 * q + i]);
        *reinterpret_cast<half4 *>(dst + rowoff + 8 * q + 4 * h) = o;
    }
}

__device__ __forceinline__ floatx16 acc_from_bias(const float4 (&b)[4])
{
    floatx16 a;
#pragma unroll
    for (int q = 0; q < 4; ++q) { a[4 * q] = b[q].x; a[4 * q + 1] = b[q].y; a[4 * q + 2] = b[q].z; a[4 * q + 3] = b[q].w; }
    return a;
}

// LDS carved by the caller (standalone kernel below, or the fused self-play kernel in c4_engine.hip):
// activations ping-pong 2 x 53,840 B (+ one all-zero row each that out-of-board taps read instead of
// branching) | conv weights double-buffered 2 x 18,432 B | MLP tables 14,848 B = 159,392 B of 160 KiB.
struct NetLds {
    _Float16 (*act)[(ROWS + 1) * CS];   // [2]
    half8 (*wbuf)[WCHUNKS];             // [2]
    float4 *mlp;                        // [MLP_F4]
};
constexpr size_t NET_LDS_BYTES = 2 * (ROWS + 1) * CS * sizeof(_Float16) + 2 * WCHUNKS * sizeof(half8) + MLP_F4 * sizeof(float4);

// One workgroup (NTHREADS threads) evaluates positions pos0 .. pos0+15.
__device__ __forceinline__ void net_forward_block(const NetDev &nd, const NetLds &L, const uint64_t *__restrict__ c0,
                                                  const uint64_t *__restrict__ c1, int n, int pos0,
                                                  float *__restrict__ values, float *__restrict__ priors,
                                                  const int *out_map = nullptr)
{
    // laundered thread id: keeps this function's lane-derived addresses from being
    // hoisted out of a persistent caller's step loop and spilled across its tree phase
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    // positions pos0 .. pos0+npos-1 are real; tiles that hold no real row are skipped in every stage
    // (a wave-uniform test), which is what a compacted, partly filled leaf batch pays for.
    // out_map (optional, P ints): output index of position p instead of pos0+p (compacted batches).
    const int npos = min(max(n - pos0, 0), P);
    const int ntiles = (npos * PIX + 31) / 32;
    _Float16 (*lds)[(ROWS + 1) * CS] = L.act;
    half8 (*wbuf)[WCHUNKS] = L.wbuf;
    float4 *mlp = L.mlp;
    const int wave = tid >> 6, lane = tid & 63;
    const int r32 = lane & 31, h = lane >> 5;
    const int n_layers = 2 * nd.n_res;

    // conv weights of layer L+1 travel global -> registers -> LDS while layer L computes
    half8 wpre[3];
    auto prefetch = [&](int L) {
        if (L < n_layers) {
            const half8 *wsrc = nd.conv_w + (size_t)L * WCHUNKS;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int c = tid + i * NTHREADS;
                if (c < WCHUNKS) wpre[i] = wsrc[c];
            }
        }
    };
    auto commit = [&](int L) {
        if (L < n_layers) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int c = tid + i * NTHREADS;
                if (c < WCHUNKS) wbuf[L & 1][c] = wpre[i];
            }
        }
    };
    auto stamp = [&](int i) {
        if (nd.stamps && pos0 == 0 && lane == 0) nd.stamps[wave * 16 + i] = __builtin_amdgcn_s_memtime();
    };
    // this lane's 16 output channels are {8q + 4h + 0..3 : q = 0..3}
    auto load_bias = [&](const float *b, float4 (&out)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) out[q] = *reinterpret_cast<const float4 *>(b + 8 * q + 4 * h);
    };
    stamp(0);
    prefetch(0);
    // NN input planes (board.py:147-154) as 4 halves per (position,pixel) row: [to-move, o, x, 0];
    // they live in lds[1], which the tower only starts writing after the stem is done
    _Float16 *inp = lds[1];
    for (int r = tid; r <= ROWS; r += NTHREADS) {
        half4 v = {};
        if (r < ROWS) {
            const int p = r / PIX, pix = r - p * PIX;
            const int y = pix / 7, x = pix - y * 7;
            const int gp = pos0 + p;
            const uint64_t b0 = gp < n ? c0[gp] : 0, b1 = gp < n ? c1[gp] : 0;
            const int bit = x * 7 + (5 - y);                       // row 0 of the planes = top of the board
            v[0] = (_Float16)((__popcll(b0 | b1) & 1) ? 0.0f : 1.0f);   // board.py:150-152
            v[1] = (_Float16)(float)((b0 >> bit) & 1);
            v[2] = (_Float16)(float)((b1 >> bit) & 1);
        }
        *reinterpret_cast<half4 *>(inp + r * 4) = v;               // r == ROWS: the zero row of the planes
    }
    if (tid < CS) lds[0][ROWS * CS + tid] = (_Float16)0.0f;                  // zero rows of the two
    else if (tid >= 64 && tid < 64 + CS) lds[1][ROWS * CS + tid - 64] = (_Float16)0.0f;   // activation buffers
    {   // MLP tables: 928 float4, used only at the very end
        const float4 m0 = nd.mlp[tid];
        const float4 m1 = tid + NTHREADS < MLP_F4 ? nd.mlp[tid + NTHREADS] : float4{0, 0, 0, 0};
        mlp[tid] = m0;
        if (tid + NTHREADS < MLP_F4) mlp[tid + NTHREADS] = m1;
    }
    __syncthreads();

    // ------------------------------------------------------------------ stem: planes -> lds[0]
    // K = 9 taps x 4 channels (36, padded to 48 = 3 MFMA steps); a lane's 8 k's are 2 taps x 4 channels,
    // i.e. two 8-byte reads of the plane rows (out-of-board taps read the zero row)
    {
        half8 w[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) w[s] = nd.stem_w[s * 64 + lane];
        float4 bias[4];
        load_bias(nd.stem_b, bias);
        for (int t = wave; t < ntiles; t += NWAVES) {
            const int rg = t * 32 + r32;
            const int p = rg / PIX, pix = rg - p * PIX;
            const int y = pix / 7, x = pix - y * 7;
            half4 v[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int tap = 4 * (i >> 1) + 2 * h + (i & 1);      // k = 16s + 8h + j = tap*4 + channel
                const int ty = (tap * 11) >> 5, tx = tap - 3 * ty;   // tap/3, tap%3 for tap < 12
                const int ok = -(int)(tap < 9 && (unsigned)(y + ty - 1) < 6u && (unsigned)(x + tx - 1) < 7u);
                const int row = ((rg + (ty - 1) * 7 + tx - 1) & ok) | (ROWS & ~ok);
                v[i] = *reinterpret_cast<const half4 *>(inp + row * 4);
            }
            floatx16 acc = acc_from_bias(bias);
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                half8 bf;
#pragma unroll
                for (int j = 0; j < 4; ++j) { bf[j] = v[2 * s][j]; bf[4 + j] = v[2 * s + 1][j]; }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[s], bf, acc, 0, 0, 0);
            }
            store_tile(acc, lds[0], rg * CS, h);
        }
    }
    stamp(1);
    commit(0);
    __syncthreads();
    stamp(2);

    // ------------------------------------------------------------------ residual tower
    // A wave owns tiles wave, wave+8, wave+16 (the last only for waves 0..4) in every layer, so the
    // per-row LDS offsets of the 9 taps are computed once and kept in registers.
    constexpr int TPW = (TILES + NWAVES - 1) / NWAVES;   // 3
    int rsel[TPW][9], rbase[TPW];
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
        const int rg = (wave + ti * NWAVES) * 32 + r32;
        const int p = rg / PIX, pix = rg - p * PIX;
        const int y = pix / 7, x = pix - y * 7;
        rbase[ti] = rg * CS;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            // zero padding without branches: out-of-board taps read the all-zero row (index ROWS)
            const int ok = -(int)((unsigned)(y + dy) < 6u && (unsigned)(x + dx) < 7u);   // all ones / zero
            rsel[ti][tap] = (((rg + dy * 7 + dx) & ok) | (ROWS & ~ok)) * CS + 8 * h;
        }
    }
    half8 idf[2];   // identity A fragments: skip[cout][row] = sum_k I[cout][k] * x[k][row]
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) idf[s][j] = (_Float16)((16 * s + 8 * h + j) == r32 ? 1.0f : 0.0f);

    for (int L = 0; L < n_layers; ++L) {
        const _Float16 *src = lds[L & 1];
        _Float16 *dst = lds[(L & 1) ^ 1];
        const bool second = L & 1;   // conv2 of a block: add the block input (lives in dst) and overwrite it
        half8 w[KSTEPS];
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) w[s] = wbuf[L & 1][s * 64 + lane];
        float4 bias[4];
        load_bias(nd.conv_b + L * F, bias);
        prefetch(L + 1);
        // Tile pipeline.  The epilogue of tile i-1 (VALU: LeakyReLU, fp16 convert, stores) is cut into
        // 8 slices that are interleaved, in program order, between the MFMA pairs of tile i, so it
        // issues in the gaps of the dependent chain instead of after it.
        floatx16 pacc = {};
        int prow = 0;
        bool have_prev = false;
        half4 pend;
        auto epi_slice = [&](int sl) {   // accumulator elements 2sl, 2sl+1 of the previous tile
            pend[2 * (sl & 1)] = (_Float16)lrelu(pacc[2 * sl]);
            pend[2 * (sl & 1) + 1] = (_Float16)lrelu(pacc[2 * sl + 1]);
            if (sl & 1) *reinterpret_cast<half4 *>(dst + prow + 8 * (sl >> 1) + 4 * h) = pend;
        };
#pragma unroll
        for (int ti = 0; ti < TPW; ++ti) {
            if (wave + ti * NWAVES < ntiles) {
                half8 bf[KSTEPS], xs[2];
#pragma unroll
                for (int s = 0; s < 4; ++s) bf[s] = *reinterpret_cast<const half8 *>(src + rsel[ti][s >> 1] + (s & 1) * 16);
                if (second) {
                    xs[0] = *reinterpret_cast<const half8 *>(dst + rbase[ti] + 8 * h);
                    xs[1] = *reinterpret_cast<const half8 *>(dst + rbase[ti] + 16 + 8 * h);
                }
                __builtin_amdgcn_sched_barrier(0);
                floatx16 acc = acc_from_bias(bias);
#pragma unroll
                for (int k = 0; k < KSTEPS / 2; ++k) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[2 * k], bf[2 * k], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[2 * k + 1], bf[2 * k + 1], acc, 0, 0, 0);
                    if (2 * k + 4 < KSTEPS) {
                        bf[2 * k + 4] = *reinterpret_cast<const half8 *>(src + rsel[ti][k + 2]);
                        bf[2 * k + 5] = *reinterpret_cast<const half8 *>(src + rsel[ti][k + 2] + 16);
                    }
                    if (have_prev && k < 8) epi_slice(k);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (second) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(idf[0], xs[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(idf[1], xs[1], acc, 0, 0, 0);
                }
                pacc = acc;
                prow = rbase[ti];
                have_prev = true;
            }
            // the other weight buffer has been idle since the previous layer's barrier: park the
            // prefetched weights there as soon as the first tile is done (frees 12 VGPRs).  Every wave
            // commits its share, also one that owns no tile of a partly filled batch.
            if (ti == 0) commit(L + 1);
        }
        if (have_prev) store_tile(pacc, dst, prow, h);   // the wave's last tile has no chain to hide under
        if (L < 6) stamp(3 + L);
        __syncthreads();
    }
    stamp(9);
    // tower output is in lds[0] (n_layers is even)

    // ------------------------------------------------------------------ 1x1 head convs (value + 2 policy channels)
    float *hs = reinterpret_cast<float *>(lds[1]);   // [P][HSTR] fp32: value plane 0..41, policy planes 42..125
    {
        const half8 w0 = nd.head_w[lane], w1 = nd.head_w[64 + lane];
        const float hb0 = nd.head_b[0], hb1 = nd.head_b[1], hb2 = nd.head_b[2];
        for (int t = wave; t < ntiles; t += NWAVES) {
            const int rg = t * 32 + r32;
            const int p = rg / PIX, pix = rg - p * PIX;
            const half8 a0 = *reinterpret_cast<const half8 *>(lds[0] + rg * CS + 8 * h);
            const half8 a1 = *reinterpret_cast<const half8 *>(lds[0] + rg * CS + 16 + 8 * h);
            floatx16 acc = {};
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, a0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, a1, acc, 0, 0, 0);
            if (h == 0) {   // couts 0..3 sit in registers 0..3 of the lower half-wave
                hs[p * HSTR + 0 * PIX + pix] = lrelu(acc[0] + hb0);
                hs[p * HSTR + 1 * PIX + pix] = lrelu(acc[1] + hb1);
                hs[p * HSTR + 2 * PIX + pix] = lrelu(acc[2] + hb2);
            }
        }
        if (tid < 2 * P) hs[(tid >> 1) * HSTR + HEADV + (tid & 1)] = 0.0f;   // pad 126,127
    }
    __syncthreads();
    stamp(10);

    // ------------------------------------------------------------------ MLP heads (fp32 VALU)
    // wave w owns positions 2w and 2w+1.
    //  value : lane o < 42 is one row of the collapsed Linear stack (model.py:69-70,83): 11 float4
    //          table reads + 11 broadcast float4 reads of the value plane per position;
    //  policy: lane = logit + 8*segment, each lane sums 11 of the 84 inputs (model.py:104,113),
    //          segments are combined with three xor-shuffles.
    {
        const float *vt_b = reinterpret_cast<const float *>(mlp + VT_F4 + PT_F / 4);   // fc_b | vout_w | pfc_b
        const float *pt = reinterpret_cast<const float *>(mlp + VT_F4);
        const int pA = 2 * wave, pB = pA + 1;
        const float4 *hA4 = reinterpret_cast<const float4 *>(hs + pA * HSTR);
        const float4 *hB4 = reinterpret_cast<const float4 *>(hs + pB * HSTR);
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
        for (int g = 0; g < 11; ++g) {
            const float4 wv = mlp[g * 64 + lane];
            const float4 xa = hA4[g], xb = hB4[g];
            a0 += wv.x * xa.x + wv.y * xa.y + wv.z * xa.z + wv.w * xa.w;
            a1 += wv.x * xb.x + wv.y * xb.y + wv.z * xb.z + wv.w * xb.w;
        }
        const int seg = lane >> 3;
        const float *hpA = hs + pA * HSTR + PIX + seg * 11, *hpB = hs + pB * HSTR + PIX + seg * 11;
        float l0 = 0.0f, l1 = 0.0f;
#pragma unroll
        for (int c = 0; c < 11; ++c) {
            const float wv = pt[c * 64 + lane];      // zero where seg*11 + c >= 84 or (lane & 7) == 7
            // ... and there the read is clamped into this position's own planes: the entries behind
            // them belong to the next position, which may be an empty row holding anything (0 * NaN)
            const int cc = seg * 11 + c < 2 * PIX ? c : 2 * PIX - 1 - seg * 11;
            l0 += wv * hpA[cc];
            l1 += wv * hpB[cc];
        }
        // combine the 8 segments: lanes i and i^8 with a row rotate, the four 16-lane rows with two shuffles
        l0 += dppf<0x128>(l0);
        l1 += dppf<0x128>(l1);
#pragma unroll
        for (int m = 16; m <= 32; m <<= 1) {
            l0 += __shfl_xor(l0, m, 64);
            l1 += __shfl_xor(l1, m, 64);
        }
        const float fb = vt_b[lane], vw = vt_b[64 + lane], pb = vt_b[128 + lane];
        const bool is_pol = lane < 7;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const float a = (pp ? a1 : a0) + fb;
            const float lg = (pp ? l1 : l0) + pb;
            const int pq = pp ? pB : pA;
            const int gp = pq < npos ? pos0 + pq : n;   // >= n: no output
            const int go = out_map ? out_map[pq < npos ? pq : 0] : gp;
            const float rs = sum16(lane < PIX ? vw * lrelu(a) : 0.0f);               // model.py:83-85
            const float vsum = readlane_f(rs, 0) + readlane_f(rs, 16) + readlane_f(rs, 32);
            const float value = (tanhf(vsum + nd.vout_b) + nd.w1) * nd.w2;           // model.py:86-88
            const float mx = max8(is_pol ? lg : -INFINITY);                          // lanes 0..7 hold the logits
            const float e = is_pol ? expf(lg - mx) : 0.0f;
            const float sum = sum8(e);
            if (gp < n) {
                if (lane == 0) values[go] = value;
                if (is_pol) priors[(size_t)go * 7 + lane] = e / sum;
            }
        }
    }
    stamp(11);
}


// ------------------------------------------------------------------------------------------------
// Wave-private forwards: ONE wave evaluates ONE position with no workgroup barrier at all -- the fused
// self-play kernel lets every wave evaluate the leaves of its own trees the moment they are needed instead
// of waiting for the slowest tree of the workgroup.  Activations ping-pong between private LDS planes of
// 43 rows (42 pixels + the zero row that out-of-board taps read instead of branching); the weights stream
// from L2 (the whole net is ~110 KB) into registers ahead of their use; per-element arithmetic (bias as the
// accumulator's initial value, k order, identity-MFMA skip, epilogue, heads, MLPs) is net_forward_block's,
// so all forwards of one net give bit-identical answers (tests/test_gpu_fused_net.py).
// All of them run on 16-row MFMA tiles (v_mfma_f32_16x16x32_f16: the position's 42 pixels are 3 row tiles):
//   net_forward_wave16n<NP>     32 filters, fp16 storage: the self-play kernels' hot forward, NP positions per pass
//   net_forward_wave16q         32 filters, reference precision (fp16 hi + lo split)
//   net_forward_wave16w         64 filters, fp16 storage
// (Earlier rounds had forwards on 32-row tiles -- two positions per pass, then one position in two half-empty tiles:
// 22-27 k cycles per pass at 32 filters where the 16-row forward takes 17-20 k; retired.)
// ------------------------------------------------------------------------------------------------
// Reference precision (C4_NET_F32X3, net_forward_wave16q): every fp32 operand x (folded weight, activation) is
// carried as two fp16 numbers
//     x  ~=  hi + lo / 2^11,      hi = f16(x),   lo = f16((x - hi) * 2^11)
// (x - hi is exact in fp32; scaling keeps lo a NORMAL fp16 of x's own magnitude), and a product of two
// operands as three fp16 MFMAs with fp32 accumulation -- hi*hi into one accumulator, hi*lo + lo*hi into
// a second one that is folded in with the factor 2^-11 in the epilogue; lo*lo (2^-22 relative) is
// dropped.  Operand error 2^-22, products exact, sums in fp32: the class of an fp32 convolution whose
// summation order differs (what the PyTorch-ROCm / MIOpen plan is against the reference's CPU convs).
// The input planes are 0/1 (exact in fp16), so the stem needs two MFMAs per k-step, every other layer three.
constexpr float LO_SCALE = 2048.0f, LO_INV = 1.0f / 2048.0f;
constexpr int PROWS = PIX + 1;               // 42 real rows + the zero row

// ------------------------------------------------------------------------------------------------
// net_forward_wave16: the one-position wave-private forward of the 32-filter fp16 net on 16-row MFMA
// tiles (v_mfma_f32_16x16x32_f16): the position's 42 pixels take 3 row tiles (48 rows) instead of two
// 32-row tiles (64 rows), a k-step is one whole tap (K = 32 input channels), so a 3x3 layer is
// 9 taps x 3 row tiles x 2 cout tiles = 54 MFMAs of 16 cycles (864 cycles) instead of 36 of 32 (1152), and
// the epilogue handles 24 values per lane instead of 32.  This is the forward the wave-autonomous
// self-play kernel spends ~40 % of its time in, almost always with ONE leaf to evaluate.
//   * operands: A[cout l&15 (+16 ct)][cin 8(l>>4)+j] = weights (host order stem_w16 / conv_w16 / head_w16),
//     B[cin 8(l>>4)+j][pixel l&15] = one 16-byte LDS read of the tap-shifted row; C: lane holds 4 consecutive
//     couts 16 ct + 4(l>>4) + i of pixel l&15 -> one 8-byte LDS store per tile;
//   * planes of 43 rows (42 pixels + the zero row out-of-board taps read) x 48 halves: the 96-byte row stride
//     makes the B reads bank-conflict free in ds_read_b128's 16-lane groups (80 bytes is 2-way here);
//   * tap offsets come from a 5 KB LDS table built once per launch (the geometry depends on the lane only):
//     five ds_read_b128 per pass instead of ~200 integer instructions;
//   * bias = the accumulators' initial value, read from LDS straight in accumulator order; residual skip =
//     one identity MFMA per tile; weights of a layer in 72 VGPRs: fragment i of layer L+1 is requested into the same
//     registers right after its last use in layer L, so the refill hides under a whole layer.
// One 16x16x32 step accumulates exactly like two 32x32x16 steps over the same 32 channels: the answers are
// bit-identical to net_forward_block's (measured on MI355X over thousands of positions and two tower depths,
// asserted by tests/test_gpu_fused_net.py::test_wave_private_forward_is_bit_identical).
// ------------------------------------------------------------------------------------------------
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int CS16 = 48;                       // halves per LDS row: 96-byte stride
constexpr int PLANE16 = PROWS * CS16;          // halves per plane (4,128 B)
constexpr int RT16 = 3;                        // row tiles of 16
constexpr int TAB16 = 40;                      // u16 per lane in the tap-offset table: 27 tower taps + 12 stem taps + pad

// tap-offset table, one row of TAB16 u16 per lane id (built once per launch by 64 threads of the workgroup):
//   [rt*9 + tap]            offset in halves of (pixel row 16 rt + (l&15), tap) in a plane, + 8 (l>>4); the zero row
//                           where the pixel is not real or the tap leaves the board
//   [27 + rt*4 + s*2 + e]   stem: offset in halves (4 per row) of input-plane row for tap 8 s + 2 (l>>4) + e
template <int CS = CS16>
__device__ __forceinline__ void build_tab16(uint16_t *tab, int lane)
{
    const int n = lane & 15, g = lane >> 4;
    for (int rt = 0; rt < RT16; ++rt) {
        const int r = 16 * rt + n, y = r / 7, x = r - 7 * y;
        const bool real = r < PIX;
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            const bool ok = real && (unsigned)(y + dy) < 6u && (unsigned)(x + dx) < 7u;
            tab[lane * TAB16 + rt * 9 + tap] = (uint16_t)((ok ? r + 7 * dy + dx : PIX) * CS + 8 * g);
        }
        for (int s = 0; s < 2; ++s)
            for (int e = 0; e < 2; ++e) {
                const int tap = 8 * s + 2 * g + e;
                const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                const bool ok = real && tap < 9 && (unsigned)(y + dy) < 6u && (unsigned)(x + dx) < 7u;
                tab[lane * TAB16 + 27 + rt * 4 + s * 2 + e] = (uint16_t)((ok ? r + 7 * dy + dx : PIX) * 4);
            }
    }
    tab[lane * TAB16 + 39] = 0;
}

__device__ __forceinline__ void store16(const floatx4 &acc, _Float16 *dst, int off, bool real)
{
    half4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (_Float16)lrelu(acc[i]);
    if (real) *reinterpret_cast<half4 *>(dst + off) = o;
}

#ifndef C4_NET_STAMP_LAYER
#define C4_NET_STAMP_LAYER 2   // diagnostic stamps 12..14 split this layer of the tower into k-loop / skip / epilogue
#endif
#ifndef C4_NET_BDEPTH
#define C4_NET_BDEPTH 2   // operand ring of the tower's k-loop: reads run this many taps minus one ahead (3 and 4 measured the same)
#endif
template <int NP>
__device__ __forceinline__ void net_forward_wave16n(const NetDev &nd, _Float16 *buf, const float4 *mlp, const float *bias_lds,
                                                    const uint16_t *tab, const uint64_t (&b0)[NP], const uint64_t (&b1)[NP],
                                                    float *__restrict__ values, float *__restrict__ priors, const int (&out)[NP],
                                                    unsigned long long *stamps = nullptr)
{
    int lane_ = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_));     // inside a persistent kernel the compiler would otherwise hoist every lane-derived address of this function out of the caller's step loop, keep them live across the tree walk and reload them from scratch mid-pass
    const int lane = lane_;
    const int n = lane & 15, g = lane >> 4;
    auto stamp = [&](int i) { if (stamps && lane == 0) stamps[i] = __builtin_amdgcn_s_memtime(); };
    stamp(0);
    const int n_layers = 2 * nd.n_res;
    constexpr int BDEPTH = NP == 1 ? C4_NET_BDEPTH : 2;   // ring of operand fragments
    // position i: ping plane at buf + 2 i PLANE16, pong plane behind it
#define P0(i) (buf + (2 * (i)) * PLANE16)
#define P1(i) (buf + (2 * (i) + 1) * PLANE16)
    // weights: the layer's 18 fragments (tap t, cout tile ct) at [(L*9 + t)*2 + ct][64 lanes], a layer ahead
    half8 w[18];
    if (n_layers > 0) {
#pragma unroll
        for (int i = 0; i < 18; ++i) w[i] = nd.conv_w16[i * 64 + lane];
    }
    half8 sw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) sw[i] = nd.stem_w16[i * 64 + lane];
    // (the stem's four fragments requested in front of the first layer's eighteen: 2.4 % slower, the first layer then waits)
    // the heads' fragment and biases are requested here, a whole tower ahead of their use (requested at the heads: +300
    // cycles per pass, -2 % expansions/s where the network waves are the busy half)
    const half8 hw = nd.head_w16[lane];
    const float hb0 = nd.head_b[0], hb1 = nd.head_b[1], hb2 = nd.head_b[2];
    // tap offsets of this lane
    uint32_t tb[TAB16 / 2];
    {
        const uint4 *t4 = reinterpret_cast<const uint4 *>(tab + lane * TAB16);
#pragma unroll
        for (int i = 0; i < TAB16 / 8; ++i) { const uint4 v = t4[i]; tb[4 * i] = v.x; tb[4 * i + 1] = v.y; tb[4 * i + 2] = v.z; tb[4 * i + 3] = v.w; }
    }
    auto tof = [&](int idx) -> int { return (int)((tb[idx >> 1] >> (16 * (idx & 1))) & 0xffffu); };
    // input planes (board.py:147-154), 4 halves per row, at the start of the pong plane (the tower writes it only after the stem)
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        _Float16 *inp = P1(i);
        if (lane <= PIX) {
            half4 v = {};
            if (lane < PIX) {
                const int y = lane / 7, x = lane - y * 7;
                const int bit = x * 7 + (5 - y);
                v[0] = (_Float16)((__popcll(b0[i] | b1[i]) & 1) ? 0.0f : 1.0f);
                v[1] = (_Float16)(float)((b0[i] >> bit) & 1);
                v[2] = (_Float16)(float)((b1[i] >> bit) & 1);
            }
            *reinterpret_cast<half4 *>(inp + lane * 4) = v;   // lane == PIX: the zero row of the planes
        }
        if (lane < CS16) { P0(i)[PIX * CS16 + lane] = (_Float16)0.0f; P1(i)[PIX * CS16 + lane] = (_Float16)0.0f; }
    }
    bool real[RT16];
    int rbase[RT16];
#pragma unroll
    for (int rt = 0; rt < RT16; ++rt) {
        real[rt] = 16 * rt + n < PIX;
        rbase[rt] = (real[rt] ? 16 * rt + n : PIX) * CS16;
    }
    // this lane's 4 output channels of cout tile ct are 16 ct + 4 g + 0..3: bias in accumulator order
    auto bias4 = [&](const float *b, int ct) -> floatx4 {
        const float4 v = *reinterpret_cast<const float4 *>(b + 16 * ct + 4 * g);
        return floatx4{v.x, v.y, v.z, v.w};
    };
    // ------------------------------------------------------------------ stem: planes -> p0   (K = 36 -> two k-steps of 32)
    {
        floatx4 acc[NP][RT16][2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const floatx4 bv = bias4(bias_lds, ct);
#pragma unroll
            for (int i = 0; i < NP; ++i)
#pragma unroll
                for (int rt = 0; rt < RT16; ++rt) acc[i][rt][ct] = bv;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < NP; ++i)
#pragma unroll
                for (int rt = 0; rt < RT16; ++rt) {
                    const half4 va = *reinterpret_cast<const half4 *>(P1(i) + tof(27 + rt * 4 + s * 2));
                    const half4 vb = *reinterpret_cast<const half4 *>(P1(i) + tof(27 + rt * 4 + s * 2 + 1));
                    half8 bf;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { bf[j] = va[j]; bf[4 + j] = vb[j]; }
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) acc[i][rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sw[s * 2 + ct], bf, acc[i][rt][ct], 0, 0, 0);
                }
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) store16(acc[i][rt][ct], P0(i), rbase[rt] + 16 * ct + 4 * g, real[rt]);
    }
    stamp(1);
    // ------------------------------------------------------------------ residual tower
    // one conv layer.  `second` is a compile-time constant (the block's second conv: pong -> ping, plus the block input), so
    // the plane offsets of all operand reads and stores are instruction immediates (+1.9 % at 8192 games; ready addresses
    // for the 27 operand fragments on top of that: nothing, and the kernel reaches the 256-VGPR limit)
    auto layer = [&](auto second_tag, const int L) {
        constexpr bool second = decltype(second_tag)::value;
        constexpr int so = second ? PLANE16 : 0, dofs = second ? 0 : PLANE16;   // source / destination plane inside a position's pair
        floatx4 acc[NP][RT16][2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const floatx4 bv = bias4(bias_lds + F * (1 + L), ct);
#pragma unroll
            for (int i = 0; i < NP; ++i)
#pragma unroll
                for (int rt = 0; rt < RT16; ++rt) acc[i][rt][ct] = bv;
        }
        const int Ln = L + 1 < n_layers ? L + 1 : 0;       // unconditional refill: a branch around the loads makes the compiler drain vmcnt in front of each
        const half8 *wnext = nd.conv_w16 + (size_t)Ln * 18 * 64 + lane;
        // operand fragments of tap t sit in ring slot t % BDEPTH; the reads run BDEPTH - 1 taps ahead of the MFMAs
        half8 bf[BDEPTH][NP][RT16];
#pragma unroll
        for (int t = 0; t < BDEPTH - 1; ++t)
#pragma unroll
            for (int i = 0; i < NP; ++i)
#pragma unroll
                for (int rt = 0; rt < RT16; ++rt) bf[t][i][rt] = *reinterpret_cast<const half8 *>(P0(i) + so + tof(rt * 9 + t));
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t + BDEPTH - 1 < 9) {
#pragma unroll
                for (int i = 0; i < NP; ++i)
#pragma unroll
                    for (int rt = 0; rt < RT16; ++rt)
                        bf[(t + BDEPTH - 1) % BDEPTH][i][rt] = *reinterpret_cast<const half8 *>(P0(i) + so + tof(rt * 9 + t + BDEPTH - 1));
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
                for (int i = 0; i < NP; ++i)
#pragma unroll
                    for (int rt = 0; rt < RT16; ++rt)
                        acc[i][rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[2 * t + ct], bf[t % BDEPTH][i][rt], acc[i][rt][ct], 0, 0, 0);
                w[2 * t + ct] = wnext[(2 * t + ct) * 64];
            }
        }
        if (L == C4_NET_STAMP_LAYER) stamp(12);
        if (second) {   // + block input (lives in dst): skip[cout][pixel] = sum_k I[cout][k] x[k][pixel]
#pragma unroll
            for (int i = 0; i < NP; ++i)
#pragma unroll
                for (int rt = 0; rt < RT16; ++rt) {
                    const half8 x = *reinterpret_cast<const half8 *>(P0(i) + dofs + rbase[rt] + 8 * g);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        half8 idf;      // identity fragment of cout tile ct: A[cout n][cin 8g + j] = (8g + j == 16 ct + n)
#pragma unroll
                        for (int j = 0; j < 8; ++j) idf[j] = (_Float16)((8 * g + j) == 16 * ct + n ? 1.0f : 0.0f);
                        acc[i][rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(idf, x, acc[i][rt][ct], 0, 0, 0);
                    }
                }
        }
        if (L == C4_NET_STAMP_LAYER) stamp(13);
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) store16(acc[i][rt][ct], P0(i) + dofs, rbase[rt] + 16 * ct + 4 * g, real[rt]);
        if (L == C4_NET_STAMP_LAYER) stamp(14);
        if (L < 6) stamp(2 + L);
    };
    for (int blk = 0; blk < nd.n_res; ++blk) {
        layer(std::false_type{}, 2 * blk);
        layer(std::true_type{}, 2 * blk + 1);
    }
    stamp(8);
    // tower output is in p0 (n_layers is even)
    // ------------------------------------------------------------------ 1x1 head convs: couts 0..2 = rows 0..2 of cout tile 0
#pragma unroll
    for (int i = 0; i < NP; ++i) {
    float *hs = reinterpret_cast<float *>(P1(i));   // [HSTR] fp32: value plane 0..41, policy planes 42..125 (the pong plane is free)
    {
        floatx4 a[RT16];
#pragma unroll
        for (int rt = 0; rt < RT16; ++rt) {
            const half8 x = *reinterpret_cast<const half8 *>(P0(i) + rbase[rt] + 8 * g);
            a[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hw, x, floatx4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
        }
#pragma unroll
        for (int rt = 0; rt < RT16; ++rt) {
            const int r = 16 * rt + n;
            if (g == 0 && r < PIX) {      // couts 0..3 sit in the accumulators of lanes 0..15
                hs[0 * PIX + r] = lrelu(a[rt][0] + hb0);
                hs[1 * PIX + r] = lrelu(a[rt][1] + hb1);
                hs[2 * PIX + r] = lrelu(a[rt][2] + hb2);
            }
        }
        if (lane < 2) hs[HEADV + lane] = 0.0f;   // pad 126,127
    }
    if (i == NP - 1) stamp(9);
    // ------------------------------------------------------------------ MLP heads (fp32 VALU), as in net_forward_block
    {
        const float *vt_b = reinterpret_cast<const float *>(mlp + VT_F4 + PT_F / 4);   // fc_b | vout_w | pfc_b
        const float *pt = reinterpret_cast<const float *>(mlp + VT_F4);
        const float4 *hA4 = reinterpret_cast<const float4 *>(hs);
        float v0 = 0.0f;
#pragma unroll
        for (int q = 0; q < 11; ++q) {
            const float4 wv = mlp[q * 64 + lane];
            const float4 xa = hA4[q];
            v0 += wv.x * xa.x + wv.y * xa.y + wv.z * xa.z + wv.w * xa.w;
        }
        const int seg = lane >> 3;
        const float *hpA = hs + PIX + seg * 11;
        float l0 = 0.0f;
#pragma unroll
        for (int c = 0; c < 11; ++c) {
            const float wv = pt[c * 64 + lane];      // zero where seg*11 + c >= 84 or (lane & 7) == 7
            const int cc = seg * 11 + c < 2 * PIX ? c : 2 * PIX - 1 - seg * 11;
            l0 += wv * hpA[cc];
        }
        l0 += dppf<0x128>(l0);
#pragma unroll
        for (int m = 16; m <= 32; m <<= 1) l0 += __shfl_xor(l0, m, 64);
        const float fb = vt_b[lane], vw = vt_b[64 + lane], pb = vt_b[128 + lane];
        const bool is_pol = lane < 7;
        const float a = v0 + fb;
        const float lg = l0 + pb;
        const float rs = sum16(lane < PIX ? vw * lrelu(a) : 0.0f);               // model.py:83-85
        const float vsum = readlane_f(rs, 0) + readlane_f(rs, 16) + readlane_f(rs, 32);
        const float value = (tanhf(vsum + nd.vout_b) + nd.w1) * nd.w2;           // model.py:86-88
        const float mx = max8(is_pol ? lg : -INFINITY);
        const float e = is_pol ? expf(lg - mx) : 0.0f;
        const float sum = sum8(e);
        if (lane == 0) values[out[i]] = value;
        if (is_pol) priors[(size_t)out[i] * 7 + lane] = e / sum;
    }
    }
    stamp(10);
#undef P0
#undef P1
}

// one position per pass (the form the standalone kernel and the wave-autonomous self-play kernel use)
__device__ __forceinline__ void net_forward_wave16(const NetDev &nd, _Float16 *buf, const float4 *mlp, const float *bias_lds,
                                                   const uint16_t *tab, uint64_t b0, uint64_t b1, float *__restrict__ values,
                                                   float *__restrict__ priors, int out, unsigned long long *stamps = nullptr)
{
    const uint64_t a0[1] = {b0}, a1[1] = {b1};
    const int o[1] = {out};
    net_forward_wave16n<1>(nd, buf, mlp, bias_lds, tab, a0, a1, values, priors, o, stamps);
}

// ------------------------------------------------------------------------------------------------
// The reference-precision forward (C4_NET_F32X3; net_forward_wave16q below; arithmetic: x ~= hi + lo / 2^11 in fp16, hi*hi
// into one accumulator, hi*lo + lo*hi into a second one) on the 16-row tiles of net_forward_wave16: 9 taps x 3 row tiles
// x 2 cout tiles x 3 = 162 MFMAs of 16 cycles per layer, four planes (ping/pong x hi/lo) of 43 rows x 96 bytes, the hi
// and lo weight fragments of a tap stream from L2 through a rolling window of three taps (the tower's taps are one
// linear sequence in memory).  (Round 2's version of it, net_forward_wave16p -- same arithmetic, 64-bit vector addresses
// per weight load, compiler-chosen instruction order: 4.0-4.6 k cycles per layer's k-loop -- was retired in round 3 after
// the bit-for-bit comparison on the device, profiles/r03_ab_f32x3_lean_forward.json.)
// ------------------------------------------------------------------------------------------------
constexpr int WTAPS = 3;   // taps of weights in flight (divides 9: a tap's window slot is t % 3 in every layer)

__device__ __forceinline__ void store16p(const floatx4 &hi, const floatx4 &lo, _Float16 *dh, _Float16 *dl, int off, bool real)
{
    half4 oh, ol;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float y = lrelu(hi[i] + lo[i] * LO_INV);
        const _Float16 yh = (_Float16)y;
        oh[i] = yh;
        ol[i] = (_Float16)((y - (float)yh) * LO_SCALE);
    }
    if (real) {
        *reinterpret_cast<half4 *>(dh + off) = oh;
        *reinterpret_cast<half4 *>(dl + off) = ol;
    }
}

// ------------------------------------------------------------------------------------------------
// net_forward_wave16q: the one-position reference-precision forward, with an instruction stream lean enough for ONE wave to
// keep its SIMD's MFMA pipe fed -- a wave cannot issue its MFMAs back to back when ~30 other instructions per tap stand
// between them in clusters (round 2: 4.0-4.6 k cycles per layer's k-loop against 2.6 k of MFMA time, alone on a CU as
// well as inside the self-play kernel; now 3.0-3.2 k):
//   * the tower's weights are ONE linear stream read with buffer loads: scalar tap offset + a lane offset fixed for the
//     whole pass + an immediate per fragment -- no 64-bit vector address arithmetic per load (and no MFMA-operand /
//     address register hazards with their s_nops), no clamp at the end of the tower (a buffer load past the end of the
//     stream returns zeros);
//   * what a pass needs FIRST -- the stem's and the heads' fragments, the tower's first two taps -- lives in LDS for the whole
//     launch (18 KB per workgroup, stage_w0_lds): a pass no longer starts with an L2 round trip in front of its first MFMA;
//   * group barriers spread the next tap's operand reads and the weight requests between the MFMAs instead of in
//     clusters in front of them (C4_F32X3_SGB; with buffer loads but WITHOUT them the compiler's order is 30 % slower
//     than round 2's);
//   * the epilogue runs on packed float32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two values per
//     instruction; hi + lo * 2^-11 as ONE fma -- the product is exact, so the rounding is the add's).
// ------------------------------------------------------------------------------------------------
#ifndef C4_F32X3_SGB
#define C4_F32X3_SGB 1
#endif
#ifndef C4_F32X3_PKEPI
#define C4_F32X3_PKEPI 1
#endif
#ifndef C4_F32X3_MIXEPI
#define C4_F32X3_MIXEPI 1
#endif
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

// The fragments every pass needs first -- stem (hi 4, lo 4), heads (hi, lo), the tower's first two taps (2 x 4) -- live in LDS
// for the whole launch (W0_FRAGS x 64 lanes x 16 bytes = 18 KB per workgroup, staged once by stage_w0_lds): a pass starts with
// LDS reads instead of an L2 round trip in front of its first MFMA, and nothing has to stay in registers between passes.
constexpr int W0_STEM = 0, W0_HEAD = 8, W0_TOWER = 10, W0_FRAGS = 18;
__device__ __forceinline__ half8 wq_load(const __amdgpu_buffer_rsrc_t &r, int voff, int soff)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return __builtin_bit_cast(half8, v);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wq_rsrc(const NetDev &nd)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)nd.conv_w16p, 0, (int)nd.conv_w16p_bytes, 0x00020000);
}
// cooperative fill by the whole workgroup (the caller synchronises afterwards)
__device__ __forceinline__ void stage_w0_lds(const NetDev &nd, half8 *w0)
{
    for (int i = threadIdx.x; i < W0_FRAGS * 64; i += blockDim.x) w0[i] = nd.w0[i];
}

__device__ __forceinline__ void store16q(const floatx4 &hi, const floatx4 &lo, _Float16 *dh, _Float16 *dl, int off, bool real)
{
#if C4_F32X3_PKEPI
    half4 oh, ol;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float2v h = {hi[2 * i], hi[2 * i + 1]}, l = {lo[2 * i], lo[2 * i + 1]};
        const float2v inv = {LO_INV, LO_INV}, leak = {LEAK, LEAK}, sc = {LO_SCALE, LO_SCALE};
        float2v y = __builtin_elementwise_fma(l, inv, h);     // = hi + lo * 2^-11 (the product is exact)
        const float2v ly = y * leak;
        asm("v_max_f32 %0, %1, %2" : "=v"(y.x) : "v"(y.x), "v"(ly.x));
        asm("v_max_f32 %0, %1, %2" : "=v"(y.y) : "v"(y.y), "v"(ly.y));
        const half2v yh = __builtin_convertvector(y, half2v);
#if C4_F32X3_MIXEPI
        // lo = RN16((y - yh) * 2^11) as ONE mixed-precision fma per value: fma(yh [fp16 operand], -2^11, y * 2^11) is exact in
        // float32 (y - yh has at most 13 significant bits), so the only rounding is the conversion -- the same value as the
        // convert / subtract / scale / convert sequence, in 3 instructions per pair instead of 5
        const float2v ysc = y * sc;
        const uint32_t yhb = __builtin_bit_cast(uint32_t, yh);
        const float nsc = -LO_SCALE;
        uint32_t ylb;
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ylb) : "v"(yhb), "s"(nsc), "v"(ysc.x));
        asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(ylb) : "v"(yhb), "s"(nsc), "v"(ysc.y));
        const half2v yl = __builtin_bit_cast(half2v, ylb);
#else
        const float2v back = __builtin_convertvector(yh, float2v);
        const float2v rl = (y - back) * sc;
        const half2v yl = __builtin_convertvector(rl, half2v);
#endif
        oh[2 * i] = yh.x; oh[2 * i + 1] = yh.y;
        ol[2 * i] = yl.x; ol[2 * i + 1] = yl.y;
    }
    if (real) {
        *reinterpret_cast<half4 *>(dh + off) = oh;
        *reinterpret_cast<half4 *>(dl + off) = ol;
    }
#else
    store16p(hi, lo, dh, dl, off, real);
#endif
}

__device__ __forceinline__ void net_forward_wave16q(const NetDev &nd, _Float16 *buf, const float4 *mlp, const float *bias_lds,
                                                    const uint16_t *tab, uint64_t b0, uint64_t b1, float *__restrict__ values,
                                                    float *__restrict__ priors, int out, const half8 *w0, unsigned long long *stamps = nullptr)
{
    int lane_ = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_));     // keep lane-derived addresses out of a persistent caller's loop (see net_forward_wave16)
    const int lane = lane_;
    const int n = lane & 15, g = lane >> 4;
    auto stamp = [&](int i) { if (stamps && lane == 0) stamps[i] = __builtin_amdgcn_s_memtime(); };
    stamp(0);
    _Float16 *const p0h = buf, *const p0l = buf + PLANE16, *const p1h = buf + 2 * PLANE16, *const p1l = buf + 3 * PLANE16;
    const __amdgpu_buffer_rsrc_t wr = wq_rsrc(nd);
    const int voff = lane * 16;
    int soff = 2 * 4096;                // stream offset of the tap the NEXT refill requests (a request past the end of the tower returns zeros)
    // rolling weight window, slot = tower tap % 3; taps 0 and 1 come from LDS, slot 2 is requested by the first tap
    half8 wh[WTAPS][2], wl[WTAPS][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            wh[t][ct] = w0[(W0_TOWER + t * 4 + 2 * ct) * 64 + lane];
            wl[t][ct] = w0[(W0_TOWER + t * 4 + 2 * ct + 1) * 64 + lane];
        }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) { wh[2][ct] = wh[0][ct]; wl[2][ct] = wl[0][ct]; }
    half8 swh[4], swl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { swh[i] = w0[(W0_STEM + i) * 64 + lane]; swl[i] = w0[(W0_STEM + 4 + i) * 64 + lane]; }
    const float hb0 = nd.head_b[0], hb1 = nd.head_b[1], hb2 = nd.head_b[2];
    uint32_t tb[TAB16 / 2];
    {
        const uint4 *t4 = reinterpret_cast<const uint4 *>(tab + lane * TAB16);
#pragma unroll
        for (int i = 0; i < TAB16 / 8; ++i) { const uint4 v = t4[i]; tb[4 * i] = v.x; tb[4 * i + 1] = v.y; tb[4 * i + 2] = v.z; tb[4 * i + 3] = v.w; }
    }
    auto tof = [&](int idx) -> int { return (int)((tb[idx >> 1] >> (16 * (idx & 1))) & 0xffffu); };
    // input planes (board.py:147-154), 4 halves per row, at the start of p1h (the tower writes it only after the stem)
    _Float16 *inp = p1h;
    if (lane <= PIX) {
        half4 v = {};
        if (lane < PIX) {
            const int y = lane / 7, x = lane - y * 7;
            const int bit = x * 7 + (5 - y);
            v[0] = (_Float16)((__popcll(b0 | b1) & 1) ? 0.0f : 1.0f);
            v[1] = (_Float16)(float)((b0 >> bit) & 1);
            v[2] = (_Float16)(float)((b1 >> bit) & 1);
        }
        *reinterpret_cast<half4 *>(inp + lane * 4) = v;   // lane == PIX: the zero row of the planes
    }
    if (lane < CS16) {
        p0h[PIX * CS16 + lane] = (_Float16)0.0f; p0l[PIX * CS16 + lane] = (_Float16)0.0f;
        p1h[PIX * CS16 + lane] = (_Float16)0.0f; p1l[PIX * CS16 + lane] = (_Float16)0.0f;
    }
    bool real[RT16];
    int rbase[RT16];
#pragma unroll
    for (int rt = 0; rt < RT16; ++rt) {
        real[rt] = 16 * rt + n < PIX;
        rbase[rt] = (real[rt] ? 16 * rt + n : PIX) * CS16;
    }
    auto bias4 = [&](const float *b, int ct) -> floatx4 {
        const float4 v = *reinterpret_cast<const float4 *>(b + 16 * ct + 4 * g);
        return floatx4{v.x, v.y, v.z, v.w};
    };
    const floatx4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    // ------------------------------------------------------------------ stem (0/1 inputs: two MFMAs per step)
    {
        floatx4 ah[RT16][2], al[RT16][2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const floatx4 bv = bias4(bias_lds, ct);
#pragma unroll
            for (int rt = 0; rt < RT16; ++rt) { ah[rt][ct] = bv; al[rt][ct] = zero4; }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int rt = 0; rt < RT16; ++rt) {
                const half4 va = *reinterpret_cast<const half4 *>(inp + tof(27 + rt * 4 + s * 2));
                const half4 vb = *reinterpret_cast<const half4 *>(inp + tof(27 + rt * 4 + s * 2 + 1));
                half8 bf;
#pragma unroll
                for (int j = 0; j < 4; ++j) { bf[j] = va[j]; bf[4 + j] = vb[j]; }
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    ah[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(swh[s * 2 + ct], bf, ah[rt][ct], 0, 0, 0);
                    al[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(swl[s * 2 + ct], bf, al[rt][ct], 0, 0, 0);
                }
            }
#pragma unroll
        for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) store16q(ah[rt][ct], al[rt][ct], p0h, p0l, rbase[rt] + 16 * ct + 4 * g, real[rt]);
    }
    stamp(1);
    // ------------------------------------------------------------------ residual tower
    auto layer = [&](auto second_tag, const int L) {
        constexpr bool second = decltype(second_tag)::value;
        const _Float16 *sh = second ? p1h : p0h, *sl = second ? p1l : p0l;
        _Float16 *dh = second ? p0h : p1h, *dl = second ? p0l : p1l;
        floatx4 ah[RT16][2], al[RT16][2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const floatx4 bv = bias4(bias_lds + F * (1 + L), ct);
#pragma unroll
            for (int rt = 0; rt < RT16; ++rt) { ah[rt][ct] = bv; al[rt][ct] = zero4; }
        }
        half8 bh[RT16], bl[RT16], nh[RT16], nl[RT16];
#pragma unroll
        for (int rt = 0; rt < RT16; ++rt) {
            bh[rt] = *reinterpret_cast<const half8 *>(sh + tof(rt * 9));
            bl[rt] = *reinterpret_cast<const half8 *>(sl + tof(rt * 9));
        }
#if C4_F32X3_SGB
        __builtin_amdgcn_sched_barrier(0);   // the group barriers below order the k-loop's own reads: the first tap's stay in front of it
#endif
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            {   // refill: the slot the PREVIOUS tap used gets the tap two ahead of this one
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    wh[(t + 2) % WTAPS][ct] = wq_load(wr, voff + (2 * ct) * 1024, soff);
                    wl[(t + 2) % WTAPS][ct] = wq_load(wr, voff + (2 * ct + 1) * 1024, soff);
                }
                soff += 4096;
            }
            if (t + 1 < 9) {
#pragma unroll
                for (int rt = 0; rt < RT16; ++rt) {
                    nh[rt] = *reinterpret_cast<const half8 *>(sh + tof(rt * 9 + t + 1));
                    nl[rt] = *reinterpret_cast<const half8 *>(sl + tof(rt * 9 + t + 1));
                }
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const half8 cwh = wh[t % WTAPS][ct], cwl = wl[t % WTAPS][ct];
#pragma unroll
                for (int rt = 0; rt < RT16; ++rt) {
                    ah[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cwh, bh[rt], ah[rt][ct], 0, 0, 0);
                    al[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cwl, bh[rt], al[rt][ct], 0, 0, 0);
                    al[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cwh, bl[rt], al[rt][ct], 0, 0, 0);
                }
            }
#pragma unroll
            for (int rt = 0; rt < RT16; ++rt) { bh[rt] = nh[rt]; bl[rt] = nl[rt]; }
#if C4_F32X3_SGB
            // this tap's instruction order: the four weight requests behind the first MFMAs, then one operand read of the
            // next tap per two MFMAs
#pragma unroll
            for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
            if (t + 1 < 9) {
#pragma unroll
                for (int i = 0; i < 6; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 14, 0);
            }
#endif
        }
#if C4_F32X3_SGB
        __builtin_amdgcn_sched_barrier(0);
#endif
        if (L == C4_NET_STAMP_LAYER) stamp(12);
        if (second) {   // + block input (lives in dh/dl): identity MFMAs keep it exact in both accumulators
#pragma unroll
            for (int rt = 0; rt < RT16; ++rt) {
                const half8 xh = *reinterpret_cast<const half8 *>(dh + rbase[rt] + 8 * g);
                const half8 xl = *reinterpret_cast<const half8 *>(dl + rbase[rt] + 8 * g);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    half8 idf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) idf[j] = (_Float16)((8 * g + j) == 16 * ct + n ? 1.0f : 0.0f);
                    ah[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(idf, xh, ah[rt][ct], 0, 0, 0);
                    al[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(idf, xl, al[rt][ct], 0, 0, 0);
                }
            }
        }
        if (L == C4_NET_STAMP_LAYER) stamp(13);
#pragma unroll
        for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) store16q(ah[rt][ct], al[rt][ct], dh, dl, rbase[rt] + 16 * ct + 4 * g, real[rt]);
        if (L == C4_NET_STAMP_LAYER) stamp(14);
        if (L < 6) stamp(2 + L);
    };
    for (int blk = 0; blk < nd.n_res; ++blk) {
        layer(std::false_type{}, 2 * blk);
        layer(std::true_type{}, 2 * blk + 1);
    }
    stamp(8);
    // ------------------------------------------------------------------ 1x1 head convs
    float *hs = reinterpret_cast<float *>(p1h);   // [HSTR] fp32 (p1 is free)
    {
        const half8 hwh = w0[W0_HEAD * 64 + lane], hwl = w0[(W0_HEAD + 1) * 64 + lane];
        floatx4 a[RT16], b[RT16];
#pragma unroll
        for (int rt = 0; rt < RT16; ++rt) {
            const half8 xh = *reinterpret_cast<const half8 *>(p0h + rbase[rt] + 8 * g);
            const half8 xl = *reinterpret_cast<const half8 *>(p0l + rbase[rt] + 8 * g);
            a[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hwh, xh, zero4, 0, 0, 0);
            b[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hwl, xh, zero4, 0, 0, 0);
            b[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hwh, xl, b[rt], 0, 0, 0);
        }
#pragma unroll
        for (int rt = 0; rt < RT16; ++rt) {
            const int r = 16 * rt + n;
            if (g == 0 && r < PIX) {
                hs[0 * PIX + r] = lrelu(a[rt][0] + b[rt][0] * LO_INV + hb0);
                hs[1 * PIX + r] = lrelu(a[rt][1] + b[rt][1] * LO_INV + hb1);
                hs[2 * PIX + r] = lrelu(a[rt][2] + b[rt][2] * LO_INV + hb2);
            }
        }
        if (lane < 2) hs[HEADV + lane] = 0.0f;   // pad 126,127
    }
    stamp(9);
    // ------------------------------------------------------------------ MLP heads (fp32 VALU), as in net_forward_block
    {
        const float *vt_b = reinterpret_cast<const float *>(mlp + VT_F4 + PT_F / 4);   // fc_b | vout_w | pfc_b
        const float *pt = reinterpret_cast<const float *>(mlp + VT_F4);
        const float4 *hA4 = reinterpret_cast<const float4 *>(hs);
        float v0 = 0.0f;
#pragma unroll
        for (int q = 0; q < 11; ++q) {
            const float4 wv = mlp[q * 64 + lane];
            const float4 xa = hA4[q];
            v0 += wv.x * xa.x + wv.y * xa.y + wv.z * xa.z + wv.w * xa.w;
        }
        const int seg = lane >> 3;
        const float *hpA = hs + PIX + seg * 11;
        float l0 = 0.0f;
#pragma unroll
        for (int cc0 = 0; cc0 < 11; ++cc0) {
            const float wv = pt[cc0 * 64 + lane];      // zero where seg*11 + c >= 84 or (lane & 7) == 7
            const int cc = seg * 11 + cc0 < 2 * PIX ? cc0 : 2 * PIX - 1 - seg * 11;
            l0 += wv * hpA[cc];
        }
        l0 += dppf<0x128>(l0);
#pragma unroll
        for (int m = 16; m <= 32; m <<= 1) l0 += __shfl_xor(l0, m, 64);
        const float fb = vt_b[lane], vw = vt_b[64 + lane], pb = vt_b[128 + lane];
        const bool is_pol = lane < 7;
        const float a = v0 + fb;
        const float lg = l0 + pb;
        const float rs = sum16(lane < PIX ? vw * lrelu(a) : 0.0f);               // model.py:83-85
        const float vsum = readlane_f(rs, 0) + readlane_f(rs, 16) + readlane_f(rs, 32);
        const float value = (tanhf(vsum + nd.vout_b) + nd.w1) * nd.w2;           // model.py:86-88
        const float mx = max8(is_pol ? lg : -INFINITY);
        const float e = is_pol ? expf(lg - mx) : 0.0f;
        const float sum = sum8(e);
        if (lane == 0) values[out] = value;
        if (is_pol) priors[(size_t)out * 7 + lane] = e / sum;
    }
    stamp(10);
}

// ------------------------------------------------------------------------------------------------
// net_forward_wave16w: the 64-filter net (data/example_config.py:8-16; fp16 storage) on the 16-row tiles of
// net_forward_wave16: 9 taps x 2 k-steps of 32 input channels x 4 cout tiles x 3 row tiles = 216 MFMAs of 16 cycles per
// layer (on 32-row tiles: 144 of 32), planes of 43 rows x 144 bytes (conflict free for the operand's ds_read_b128), the
// four weight fragments of a (tap, k-step) unit stream from L2 through a rolling window of three units (the tower's units
// are one linear sequence in memory).
// ------------------------------------------------------------------------------------------------
constexpr int CS64 = 72;                       // halves per LDS row at 64 filters: 144-byte stride
constexpr int PLANE64 = PROWS * CS64;          // halves per plane (6,192 B)
constexpr int WUNITS = 3;                      // (tap, k-step) units of weights in flight (divides 18: a unit's window slot is u % 3 in every layer)

__device__ __forceinline__ void net_forward_wave16w(const NetDev &nd, _Float16 *buf, const float4 *mlp, const float *bias_lds,
                                                    const uint16_t *tab, uint64_t b0, uint64_t b1, float *__restrict__ values,
                                                    float *__restrict__ priors, int out, unsigned long long *stamps = nullptr)
{
    constexpr int FW = 64, CT = 4, KS = 2;     // filters, cout tiles of 16, k-steps of 32 per tap
    int lane_ = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_));     // keep lane-derived addresses out of a persistent caller's loop (see net_forward_wave16)
    const int lane = lane_;
    const int n = lane & 15, g = lane >> 4;
    auto stamp = [&](int i) { if (stamps && lane == 0) stamps[i] = __builtin_amdgcn_s_memtime(); };
    stamp(0);
    const int n_layers = 2 * nd.n_res;
    _Float16 *const p0 = buf, *const p1 = buf + PLANE64;
    // rolling weight window: fragment (unit U = (L*9 + tap)*2 + ks of the tower, cout tile ct) at [(U * 4 + ct) * 64 + lane]
    half8 w[WUNITS][CT];
    const half8 *wp = nd.conv_w16 + lane;
    const int total_units = n_layers * 18;
#pragma unroll
    for (int u = 0; u < WUNITS; ++u) {
        const int U = u < total_units ? u : 0;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) w[u][ct] = wp[(U * CT + ct) * 64];
    }
    half8 sw[2][CT];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) sw[s][ct] = nd.stem_w16[(s * CT + ct) * 64 + lane];
    const half8 hw0 = nd.head_w16[lane], hw1 = nd.head_w16[64 + lane];   // the heads' fragments and biases: requested a whole tower ahead
    const float hb0 = nd.head_b[0], hb1 = nd.head_b[1], hb2 = nd.head_b[2];
    uint32_t tb[TAB16 / 2];
    {
        const uint4 *t4 = reinterpret_cast<const uint4 *>(tab + lane * TAB16);
#pragma unroll
        for (int i = 0; i < TAB16 / 8; ++i) { const uint4 v = t4[i]; tb[4 * i] = v.x; tb[4 * i + 1] = v.y; tb[4 * i + 2] = v.z; tb[4 * i + 3] = v.w; }
    }
    auto tof = [&](int idx) -> int { return (int)((tb[idx >> 1] >> (16 * (idx & 1))) & 0xffffu); };
    // input planes (board.py:147-154), 4 halves per row, at the start of p1 (the tower writes p1 only after the stem)
    _Float16 *inp = p1;
    if (lane <= PIX) {
        half4 v = {};
        if (lane < PIX) {
            const int y = lane / 7, x = lane - y * 7;
            const int bit = x * 7 + (5 - y);
            v[0] = (_Float16)((__popcll(b0 | b1) & 1) ? 0.0f : 1.0f);
            v[1] = (_Float16)(float)((b0 >> bit) & 1);
            v[2] = (_Float16)(float)((b1 >> bit) & 1);
        }
        *reinterpret_cast<half4 *>(inp + lane * 4) = v;   // lane == PIX: the zero row of the planes
    }
    for (int i = lane; i < CS64; i += 64) { p0[PIX * CS64 + i] = (_Float16)0.0f; p1[PIX * CS64 + i] = (_Float16)0.0f; }
    bool real[RT16];
    int rbase[RT16];
#pragma unroll
    for (int rt = 0; rt < RT16; ++rt) {
        real[rt] = 16 * rt + n < PIX;
        rbase[rt] = (real[rt] ? 16 * rt + n : PIX) * CS64;
    }
    auto bias4 = [&](const float *b, int ct) -> floatx4 {
        const float4 v = *reinterpret_cast<const float4 *>(b + 16 * ct + 4 * g);
        return floatx4{v.x, v.y, v.z, v.w};
    };
    // ------------------------------------------------------------------ stem: planes -> p0   (K = 36 -> two k-steps of 32)
    {
        floatx4 acc[RT16][CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const floatx4 bv = bias4(bias_lds, ct);
#pragma unroll
            for (int rt = 0; rt < RT16; ++rt) acc[rt][ct] = bv;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int rt = 0; rt < RT16; ++rt) {
                const half4 va = *reinterpret_cast<const half4 *>(inp + tof(27 + rt * 4 + s * 2));
                const half4 vb = *reinterpret_cast<const half4 *>(inp + tof(27 + rt * 4 + s * 2 + 1));
                half8 bf;
#pragma unroll
                for (int j = 0; j < 4; ++j) { bf[j] = va[j]; bf[4 + j] = vb[j]; }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sw[s][ct], bf, acc[rt][ct], 0, 0, 0);
            }
#pragma unroll
        for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) store16(acc[rt][ct], p0, rbase[rt] + 16 * ct + 4 * g, real[rt]);
    }
    stamp(1);
    // ------------------------------------------------------------------ residual tower
    // one conv layer; `second` is a compile-time constant, so the plane offsets of all operand reads and stores are immediates
    auto layer = [&](auto second_tag, const int L) {
        constexpr bool second = decltype(second_tag)::value;
        const _Float16 *src = second ? p1 : p0;
        _Float16 *dst = second ? p0 : p1;
        floatx4 acc[RT16][CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const floatx4 bv = bias4(bias_lds + FW * (1 + L), ct);
#pragma unroll
            for (int rt = 0; rt < RT16; ++rt) acc[rt][ct] = bv;
        }
        half8 bc[RT16], bn[RT16];
#pragma unroll
        for (int rt = 0; rt < RT16; ++rt) bc[rt] = *reinterpret_cast<const half8 *>(src + tof(rt * 9));
#pragma unroll
        for (int u = 0; u < 18; ++u) {   // unit u = (tap u / 2, k-step u % 2)
            if (u + 1 < 18) {
#pragma unroll
                for (int rt = 0; rt < RT16; ++rt) bn[rt] = *reinterpret_cast<const half8 *>(src + tof(rt * 9 + (u + 1) / 2) + 32 * ((u + 1) % 2));
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const half8 cw = w[u % WUNITS][ct];
#pragma unroll
                for (int rt = 0; rt < RT16; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cw, bc[rt], acc[rt][ct], 0, 0, 0);
            }
            {   // refill the window slot just used with unit (L*18 + u + WUNITS) of the tower; unconditional (past the end it
                // re-reads unit 0: a branch around the loads would drain vmcnt)
                int U = L * 18 + u + WUNITS;
                U = U < total_units ? U : 0;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) w[u % WUNITS][ct] = wp[(U * CT + ct) * 64];
            }
#pragma unroll
            for (int rt = 0; rt < RT16; ++rt) bc[rt] = bn[rt];
        }
        if (second) {   // + block input (lives in dst): identity fragment of cout tile ct picks input channels 16 ct .. 16 ct + 15,
                        // which sit in k-step ct / 2: A[cout n][cin 32 (ct / 2) + 8 g + j] = (8 g + j == 16 (ct % 2) + n)
#pragma unroll
            for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const half8 x = *reinterpret_cast<const half8 *>(dst + rbase[rt] + 32 * ks + 8 * g);
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) {
                        half8 idf;
#pragma unroll
                        for (int j = 0; j < 8; ++j) idf[j] = (_Float16)((8 * g + j) == 16 * c2 + n ? 1.0f : 0.0f);
                        acc[rt][2 * ks + c2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(idf, x, acc[rt][2 * ks + c2], 0, 0, 0);
                    }
                }
        }
#pragma unroll
        for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) store16(acc[rt][ct], dst, rbase[rt] + 16 * ct + 4 * g, real[rt]);
        if (L < 6) stamp(2 + L);
    };
    for (int blk = 0; blk < nd.n_res; ++blk) {
        layer(std::false_type{}, 2 * blk);
        layer(std::true_type{}, 2 * blk + 1);
    }
    stamp(8);
    // tower output is in p0 (n_layers is even)
    // ------------------------------------------------------------------ 1x1 head convs: couts 0..2 = rows 0..2 of a cout tile, two k-steps
    float *hs = reinterpret_cast<float *>(p1);   // [HSTR] fp32: value plane 0..41, policy planes 42..125 (p1 is free)
    {
        floatx4 a[RT16];
#pragma unroll
        for (int rt = 0; rt < RT16; ++rt) {
            const half8 x0 = *reinterpret_cast<const half8 *>(p0 + rbase[rt] + 8 * g);
            const half8 x1 = *reinterpret_cast<const half8 *>(p0 + rbase[rt] + 32 + 8 * g);
            a[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hw0, x0, floatx4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
            a[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hw1, x1, a[rt], 0, 0, 0);
        }
#pragma unroll
        for (int rt = 0; rt < RT16; ++rt) {
            const int r = 16 * rt + n;
            if (g == 0 && r < PIX) {      // couts 0..3 sit in the accumulators of lanes 0..15
                hs[0 * PIX + r] = lrelu(a[rt][0] + hb0);
                hs[1 * PIX + r] = lrelu(a[rt][1] + hb1);
                hs[2 * PIX + r] = lrelu(a[rt][2] + hb2);
            }
        }
        if (lane < 2) hs[HEADV + lane] = 0.0f;   // pad 126,127
    }
    stamp(9);
    // ------------------------------------------------------------------ MLP heads (fp32 VALU), as in net_forward_block
    {
        const float *vt_b = reinterpret_cast<const float *>(mlp + VT_F4 + PT_F / 4);   // fc_b | vout_w | pfc_b
        const float *pt = reinterpret_cast<const float *>(mlp + VT_F4);
        const float4 *hA4 = reinterpret_cast<const float4 *>(hs);
        float v0 = 0.0f;
#pragma unroll
        for (int q = 0; q < 11; ++q) {
            const float4 wv = mlp[q * 64 + lane];
            const float4 xa = hA4[q];
            v0 += wv.x * xa.x + wv.y * xa.y + wv.z * xa.z + wv.w * xa.w;
        }
        const int seg = lane >> 3;
        const float *hpA = hs + PIX + seg * 11;
        float l0 = 0.0f;
#pragma unroll
        for (int c = 0; c < 11; ++c) {
            const float wv = pt[c * 64 + lane];      // zero where seg*11 + c >= 84 or (lane & 7) == 7
            const int cc = seg * 11 + c < 2 * PIX ? c : 2 * PIX - 1 - seg * 11;
            l0 += wv * hpA[cc];
        }
        l0 += dppf<0x128>(l0);
#pragma unroll
        for (int m = 16; m <= 32; m <<= 1) l0 += __shfl_xor(l0, m, 64);
        const float fb = vt_b[lane], vw = vt_b[64 + lane], pb = vt_b[128 + lane];
        const bool is_pol = lane < 7;
        const float a = v0 + fb;
        const float lg = l0 + pb;
        const float rs = sum16(lane < PIX ? vw * lrelu(a) : 0.0f);               // model.py:83-85
        const float vsum = readlane_f(rs, 0) + readlane_f(rs, 16) + readlane_f(rs, 32);
        const float value = (tanhf(vsum + nd.vout_b) + nd.w1) * nd.w2;           // model.py:86-88
        const float mx = max8(is_pol ? lg : -INFINITY);
        const float e = is_pol ? expf(lg - mx) : 0.0f;
        const float sum = sum8(e);
        if (lane == 0) values[out] = value;
        if (is_pol) priors[(size_t)out * 7 + lane] = e / sum;
    }
    stamp(10);
}

// net mode of a NetDev, as the kernels are specialised
constexpr int NETMODE_F32_F16 = 0;    // 32 filters, fp16 storage: net_forward_wave16 / net_forward_block
constexpr int NETMODE_F32_PRECISE = 1;
constexpr int NETMODE_F64 = 2;        // 64 filters, fp16 storage, one position per pass
// LDS for the fragments a pass needs first (only the reference-precision forward keeps any)
template <int MODE> struct W0Lds { static constexpr int FRAGS = MODE == NETMODE_F32_PRECISE ? W0_FRAGS * 64 : 1; };
template <int MODE>
__device__ __forceinline__ void net_forward_wave1_mode(const NetDev &nd, _Float16 *buf, const float4 *mlp, const float *bias_lds,
                                                       const uint16_t *tab, uint64_t b0, uint64_t b1, float *__restrict__ values,
                                                       float *__restrict__ priors, int out, const half8 *w0,
                                                       unsigned long long *stamps = nullptr)
{
    if constexpr (MODE == NETMODE_F64) net_forward_wave16w(nd, buf, mlp, bias_lds, tab, b0, b1, values, priors, out, stamps);
    else if constexpr (MODE == NETMODE_F32_PRECISE) net_forward_wave16q(nd, buf, mlp, bias_lds, tab, b0, b1, values, priors, out, w0, stamps);
    else net_forward_wave16(nd, buf, mlp, bias_lds, tab, b0, b1, values, priors, out, stamps);
}
// halves of private LDS a wave needs for its planes in each mode
template <int MODE> struct WaveBuf {
    static constexpr int HALVES = MODE == NETMODE_F64 ? 2 * PLANE64 : (MODE == NETMODE_F32_PRECISE ? 4 * PLANE16 : 2 * PLANE16);
};

constexpr int BIAS_LDS_FLOATS = 32 * (1 + 32);    // 4.2 KB: stem + conv biases of up to 16 residual blocks at 32 filters, 7 at 64
// cooperative fill of the LDS bias copy by the whole workgroup (the caller synchronises afterwards)
__device__ __forceinline__ void stage_bias_lds(const NetDev &nd, float *bias_lds)
{
    const int n_layers = 2 * nd.n_res, fw = nd.filters;
    for (int i = threadIdx.x; i < fw * (1 + n_layers); i += NTHREADS) bias_lds[i] = i < fw ? nd.stem_b[i] : nd.conv_b[i - fw];
}
}  // namespace c4net

// host-side handle behind c4_net_* (include/c4_engine.h)
struct c4_net {
    int device;
    c4net::NetDev d;
    std::vector<void *> allocs;
};

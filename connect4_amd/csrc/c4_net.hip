// c4_net.hip -- fused policy/value network forward for the leaf batch, hand-written for gfx950.
//
// Replaces ModelWrapper._call_list + Net.forward (oinkoink/neural/pytorch/model.py:120-134,
// 269-282) for eval-mode inference: reads the leaves' BITBOARDS (no plane tensor is materialised,
// board.py:147-154 is decoded in registers), runs stem conv -> R residual blocks -> both heads ->
// value/policy MLPs in ONE kernel, writes values[n] in [0,1] and priors[n][7].
//
// Mapping (F = 32 filters):
//   * one workgroup = 4 waves = 16 positions = 672 (position,pixel) rows = 21 MFMA tiles of 32 rows;
//   * every 3x3 conv is an implicit GEMM  out^T[cout][row] = sum_k W^T[cout][k] * act[k][row],
//     k = (tap, cin), K = 288 = 18 steps of v_mfma_f32_32x32x16_f16 (A = weights, B = activations),
//     so a lane's 16 accumulators are 16 couts of ONE pixel -> 4 packed 8-byte LDS stores;
//   * weights of the current layer live in registers (18 x half8 = 72 VGPRs per lane, pre-swizzled
//     on the host into MFMA lane order, one coalesced 1 KiB load per k-step);
//   * activations ping-pong between two LDS buffers [672 rows][40 halves] (80-byte row stride makes
//     the ds_read_b128 fragment reads bank-conflict free); zero padding is a per-row 9-bit tap
//     mask, no halo; the residual add reads the skip element from the other buffer in the epilogue;
//   * BatchNorm is folded into weights/bias on the host (eval mode), LeakyReLU(0.01) in the epilogue;
//   * storage fp16, accumulation fp32 (same 10-bit mantissa as the TF32 path cuDNN uses by default
//     for the reference's convs on its own GPU); heads and MLPs in fp32 on the VALU.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/c4_engine.h"

namespace {

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
constexpr int NWAVES = 4;
constexpr float LEAK = 0.01f;

struct NetDev {
    const half8 *stem_w;   // [3][64]           lane-ordered A fragments
    const float *stem_b;   // [32]
    const half8 *conv_w;   // [2R][18][64]
    const float *conv_b;   // [2R][32]
    const half8 *head_w;   // [2][64]           couts 0..2 = value, policy0, policy1
    const float *head_b;   // [4]
    const float *vfc_w;    // [42][42] collapsed Linear stack
    const float *vfc_b;    // [42]
    const float *vout_w;   // [42]
    const float *pfc_w;    // [7][84]
    const float *pfc_b;    // [7]
    float vout_b, w1, w2;
    int n_res;
};

__device__ __forceinline__ float lrelu(float v) { return v > 0.0f ? v : LEAK * v; }

// epilogue of one 32-row tile: bias (+ skip) + LeakyReLU, fp16, 4 x 8-byte stores
__device__ __forceinline__ void store_tile(const floatx16 &acc, const float *__restrict__ bias, _Float16 *dst,
                                           const _Float16 *skip, int rowoff, int h)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int cb = 8 * q + 4 * h;
        const float4 b4 = *reinterpret_cast<const float4 *>(bias + cb);
        float v0 = acc[4 * q + 0] + b4.x, v1 = acc[4 * q + 1] + b4.y, v2 = acc[4 * q + 2] + b4.z,
              v3 = acc[4 * q + 3] + b4.w;
        if (skip) {
            const half4 s4 = *reinterpret_cast<const half4 *>(skip + rowoff + cb);
            v0 += (float)s4[0]; v1 += (float)s4[1]; v2 += (float)s4[2]; v3 += (float)s4[3];
        }
        half4 o;
        o[0] = (_Float16)lrelu(v0); o[1] = (_Float16)lrelu(v1); o[2] = (_Float16)lrelu(v2); o[3] = (_Float16)lrelu(v3);
        *reinterpret_cast<half4 *>(dst + rowoff + cb) = o;
    }
}

__global__ __launch_bounds__(256) void c4_net_kernel(NetDev nd, const uint64_t *__restrict__ c0,
                                                     const uint64_t *__restrict__ c1, int n,
                                                     float *__restrict__ values, float *__restrict__ priors)
{
    __shared__ __attribute__((aligned(16))) _Float16 lds[2][ROWS * CS];   // 2 x 53,760 B

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r32 = lane & 31, h = lane >> 5;
    const int pos0 = blockIdx.x * P;

    // ------------------------------------------------------------------ stem: bitboards -> lds[0]
    {
        half8 w[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) w[s] = nd.stem_w[s * 64 + lane];
        for (int t = wave; t < TILES; t += NWAVES) {
            const int rg = t * 32 + r32;
            const int p = rg / PIX, pix = rg - p * PIX;
            const int y = pix / 7, x = pix - y * 7;
            const int gp = pos0 + p;
            const uint64_t b0 = gp < n ? c0[gp] : 0, b1 = gp < n ? c1[gp] : 0;
            const float to_move = (__popcll(b0 | b1) & 1) ? 0.0f : 1.0f;   // board.py:150-152
            floatx16 acc = {};
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                half8 bf;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = 16 * s + 8 * h + j;   // k = tap*4 + channel (channel 3 = zero pad)
                    const int tap = k >> 2, ch = k & 3;
                    float v = 0.0f;
                    const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
                    if (tap < 9 && ch < 3 && yy >= 0 && yy < 6 && xx >= 0 && xx < 7) {
                        const int bit = xx * 7 + (5 - yy);   // row 0 of the planes = top of the board
                        v = ch == 0 ? to_move : (float)(((ch == 1 ? b0 : b1) >> bit) & 1);
                    }
                    bf[j] = (_Float16)v;
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[s], bf, acc, 0, 0, 0);
            }
            store_tile(acc, nd.stem_b, lds[0], nullptr, rg * CS, h);
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ residual tower
    const int n_layers = 2 * nd.n_res;
    for (int L = 0; L < n_layers; ++L) {
        const _Float16 *src = lds[L & 1];
        _Float16 *dst = lds[(L & 1) ^ 1];
        const bool second = L & 1;   // conv2 of a block: add the block input (lives in dst) and overwrite it
        half8 w[KSTEPS];
        const half8 *wp = nd.conv_w + (size_t)L * KSTEPS * 64 + lane;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) w[s] = wp[s * 64];
        const float *bias = nd.conv_b + L * F;
        for (int t = wave; t < TILES; t += NWAVES) {
            const int rg = t * 32 + r32;
            const int p = rg / PIX, pix = rg - p * PIX;
            const int y = pix / 7, x = pix - y * 7;
            floatx16 acc = {};
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                const int tap = s >> 1;
                const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                const bool valid = (unsigned)(y + dy) < 6u && (unsigned)(x + dx) < 7u;
                half8 bf = {};
                if (valid) bf = *reinterpret_cast<const half8 *>(src + (rg + dy * 7 + dx) * CS + (s & 1) * 16 + 8 * h);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[s], bf, acc, 0, 0, 0);
            }
            store_tile(acc, bias, dst, second ? dst : nullptr, rg * CS, h);
        }
        __syncthreads();
    }
    // tower output is in lds[0] (n_layers is even)

    // ------------------------------------------------------------------ 1x1 head convs (value + 2 policy channels)
    float *hs = reinterpret_cast<float *>(lds[1]);   // [P][3][42] fp32
    {
        const half8 w0 = nd.head_w[lane], w1 = nd.head_w[64 + lane];
        for (int t = wave; t < TILES; t += NWAVES) {
            const int rg = t * 32 + r32;
            const int p = rg / PIX, pix = rg - p * PIX;
            const half8 a0 = *reinterpret_cast<const half8 *>(lds[0] + rg * CS + 8 * h);
            const half8 a1 = *reinterpret_cast<const half8 *>(lds[0] + rg * CS + 16 + 8 * h);
            floatx16 acc = {};
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, a0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, a1, acc, 0, 0, 0);
            if (h == 0) {   // couts 0..3 sit in registers 0..3 of the lower half-wave
                hs[(p * 3 + 0) * PIX + pix] = lrelu(acc[0] + nd.head_b[0]);
                hs[(p * 3 + 1) * PIX + pix] = lrelu(acc[1] + nd.head_b[1]);
                hs[(p * 3 + 2) * PIX + pix] = lrelu(acc[2] + nd.head_b[2]);
            }
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ MLP heads (fp32 VALU): 16 lanes per position
    {
        const int p = threadIdx.x >> 4, j = threadIdx.x & 15;
        const int gp = pos0 + p;
        const float *hv = hs + p * 3 * PIX;   // value plane, 42
        const float *hp = hv + PIX;           // policy planes, 84 (channel-major = view(N,1,-1) order)
        float part = 0.0f;
#pragma unroll
        for (int oo = 0; oo < 3; ++oo) {
            const int o = j + 16 * oo;
            if (o < PIX) {
                float a = nd.vfc_b[o];
                const float *wr = nd.vfc_w + o * PIX;
                for (int i = 0; i < PIX; ++i) a += wr[i] * hv[i];
                part += nd.vout_w[o] * lrelu(a);   // model.py:83-85
            }
        }
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) part += __shfl_xor(part, m, 16);
        const float value = (tanhf(part + nd.vout_b) + nd.w1) * nd.w2;   // model.py:86-88
        float logit = -INFINITY;
        if (j < 7) {
            float a = nd.pfc_b[j];
            const float *wr = nd.pfc_w + j * 2 * PIX;
            for (int i = 0; i < 2 * PIX; ++i) a += wr[i] * hp[i];
            logit = a;
        }
        float mx = logit;
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 16));
        const float e = j < 7 ? expf(logit - mx) : 0.0f;
        float sum = e;
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) sum += __shfl_xor(sum, m, 16);
        if (gp < n) {
            if (j == 0) values[gp] = value;
            if (j < 7) priors[(size_t)gp * 7 + j] = e / sum;
        }
    }
}

thread_local char n_err[512] = "";

}  // namespace

struct c4_net {
    int device;
    NetDev d;
    std::vector<void *> allocs;
};

namespace {
template <typename T>
hipError_t upload(c4_net *net, const std::vector<T> &host, const T **dev)
{
    void *q = nullptr;
    hipError_t r = hipMalloc(&q, host.size() * sizeof(T));
    if (r != hipSuccess) return r;
    net->allocs.push_back(q);
    r = hipMemcpy(q, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice);
    *dev = (const T *)q;
    return r;
}
}  // namespace

extern "C" {

const char *c4_net_last_error(void) { return n_err; }

int c4_net_create(int device, const c4_net_desc *desc, c4_net **out)
{
    if (!desc || !out) { snprintf(n_err, 512, "c4_net_create: null argument"); return C4_EINVAL; }
    *out = nullptr;
    if (desc->filters != F || desc->channels != 3 || desc->n_residuals < 0 || desc->n_residuals > 64) {
        snprintf(n_err, 512, "fused net supports channels=3, filters=%d (got channels=%d filters=%d residuals=%d)",
                 F, desc->channels, desc->filters, desc->n_residuals);
        return C4_EINVAL;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess) {
        snprintf(n_err, 512, "no usable HIP device %d: the fused net has no CPU fallback", device);
        return C4_EDEVICE;
    }
    c4_net *net = new c4_net();
    net->device = device;
    memset(&net->d, 0, sizeof(NetDev));
    const int R = desc->n_residuals;
    // ---- stem: A[cout][k], k = tap*4 + ch (ch 3 zero), 48 = 3 k-steps; lane l holds cout l&31, k = 16s + 8(l>>5) + j
    std::vector<_Float16> stem(3 * 64 * 8), conv((size_t)2 * R * KSTEPS * 64 * 8), head(2 * 64 * 8);
    for (int s = 0; s < 3; ++s)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * s + 8 * (l >> 5) + j, tap = k >> 2, ch = k & 3, co = l & 31;
                float v = 0.0f;
                if (tap < 9 && ch < 3) v = desc->stem_w[((co * 3 + ch) * 3 + tap / 3) * 3 + tap % 3];
                stem[(s * 64 + l) * 8 + j] = (_Float16)v;
            }
    // ---- 3x3 convs: k-step s: tap = s>>1, cin = (s&1)*16 + 8(l>>5) + j
    for (int L = 0; L < 2 * R; ++L)
        for (int s = 0; s < KSTEPS; ++s)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int tap = s >> 1, ci = (s & 1) * 16 + 8 * (l >> 5) + j, co = l & 31;
                    const float v = desc->conv_w[((((size_t)L * F + co) * F + ci) * 3 + tap / 3) * 3 + tap % 3];
                    conv[(((size_t)L * KSTEPS + s) * 64 + l) * 8 + j] = (_Float16)v;
                }
    // ---- head 1x1: couts 0..2 (value, policy0, policy1), cin = 16s + 8(l>>5) + j
    for (int s = 0; s < 2; ++s)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int ci = 16 * s + 8 * (l >> 5) + j, co = l & 31;
                head[(s * 64 + l) * 8 + j] = (_Float16)(co < 3 ? desc->head_w[co * F + ci] : 0.0f);
            }
    std::vector<float> stem_b(desc->stem_b, desc->stem_b + F), conv_b(desc->conv_b, desc->conv_b + (size_t)2 * R * F),
        head_b(4, 0.0f), vfc_w(desc->vfc_w, desc->vfc_w + 42 * 42), vfc_b(desc->vfc_b, desc->vfc_b + 42),
        vout_w(desc->vout_w, desc->vout_w + 42), pfc_w(desc->pfc_w, desc->pfc_w + 7 * 84), pfc_b(desc->pfc_b, desc->pfc_b + 7);
    for (int i = 0; i < 3; ++i) head_b[i] = desc->head_b[i];
    if (conv.empty()) conv.resize(8);
    if (conv_b.empty()) conv_b.resize(4);
    hipError_t r = hipSuccess;
    const _Float16 *p16;
#define UP16(vec, field) if (r == hipSuccess) { r = upload(net, vec, &p16); net->d.field = (const half8 *)p16; }
#define UP32(vec, field) if (r == hipSuccess) r = upload(net, vec, &net->d.field);
    UP16(stem, stem_w) UP16(conv, conv_w) UP16(head, head_w)
    UP32(stem_b, stem_b) UP32(conv_b, conv_b) UP32(head_b, head_b) UP32(vfc_w, vfc_w) UP32(vfc_b, vfc_b)
    UP32(vout_w, vout_w) UP32(pfc_w, pfc_w) UP32(pfc_b, pfc_b)
#undef UP16
#undef UP32
    if (r != hipSuccess) {
        snprintf(n_err, 512, "weight upload failed: %s", hipGetErrorString(r));
        c4_net_destroy(net);
        return C4_EDEVICE;
    }
    net->d.vout_b = desc->vout_b;
    net->d.w1 = desc->w1;
    net->d.w2 = desc->w2;
    net->d.n_res = R;
    *out = net;
    return C4_OK;
}

int c4_net_destroy(c4_net *net)
{
    if (!net) return C4_OK;
    (void)hipSetDevice(net->device);
    for (void *p : net->allocs) (void)hipFree(p);
    delete net;
    return C4_OK;
}

int c4_net_forward(c4_net *net, void *hip_stream, const uint64_t *color0_dev, const uint64_t *color1_dev, int32_t n,
                   float *values_dev, float *priors_dev)
{
    if (!net || !color0_dev || !color1_dev || !values_dev || !priors_dev || n < 0) {
        snprintf(n_err, 512, "c4_net_forward: bad argument");
        return C4_EINVAL;
    }
    if (n == 0) return C4_OK;
    const dim3 grid((n + P - 1) / P), block(256);
    hipLaunchKernelGGL(c4_net_kernel, grid, block, 0, (hipStream_t)hip_stream, net->d, color0_dev, color1_dev, (int)n,
                       values_dev, priors_dev);
    hipError_t r = hipGetLastError();
    if (r != hipSuccess) {
        snprintf(n_err, 512, "c4_net_kernel launch failed: %s", hipGetErrorString(r));
        return C4_EDEVICE;
    }
    return C4_OK;
}

}  // extern "C"

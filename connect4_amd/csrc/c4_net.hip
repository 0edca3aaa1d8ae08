// c4_net.hip -- fused policy/value network forward for the leaf batch, hand-written for gfx950.
//
// Replaces ModelWrapper._call_list + Net.forward (oinkoink/neural/pytorch/model.py:120-134,
// 269-282) for eval-mode inference: reads the leaves' BITBOARDS (no plane tensor is materialised,
// board.py:147-154 is decoded in registers), runs stem conv -> R residual blocks -> both heads ->
// value/policy MLPs in ONE kernel, writes values[n] in [0,1] and priors[n][7].
//
// Mapping (F = 32 filters):
//   * one workgroup = 8 waves (two per SIMD) = 16 positions = 672 (position,pixel) rows = 21 MFMA
//     tiles of 32 rows;
//   * every 3x3 conv is an implicit GEMM  out^T[cout][row] = sum_k W^T[cout][k] * act[k][row],
//     k = (tap, cin), K = 288 = 18 steps of v_mfma_f32_32x32x16_f16 (A = weights, B = activations),
//     so a lane's 16 accumulators are 16 couts of ONE pixel -> 4 packed 8-byte LDS stores;
//   * weights of the current layer live in registers (18 x half8 = 72 VGPRs per lane, pre-swizzled
//     on the host into MFMA lane order); the next layer's 18 KiB are prefetched global -> registers
//     -> LDS while the current layer computes, so no wave waits on L2 at a layer boundary;
//   * activations ping-pong between two LDS buffers [672 rows][40 halves] (80-byte row stride makes
//     the ds_read_b128 fragment reads bank-conflict free); zero padding = out-of-board taps read an
//     all-zero row, no halo and no branch; the residual skip enters as two MFMAs against an identity
//     matrix, the bias as the accumulator's initial value;
//   * BatchNorm is folded into weights/bias on the host (eval mode), LeakyReLU(0.01) in the epilogue;
//   * storage fp16, accumulation fp32 (same 10-bit mantissa as the TF32 path cuDNN uses by default
//     for the reference's convs on its own GPU); heads and MLPs in fp32 on the VALU.
//
// Several forwards share this arithmetic: net_forward_block (above; c4_net_forward for the 32-filter fp16 net, the
// workgroup-synchronous self-play kernel) and the wave-private forwards on 16-row MFMA tiles -- one wave = one position,
// private LDS planes, weights streamed from L2 into registers, no workgroup barrier: net_forward_wave16 (32 filters, fp16;
// bit-identical to the block forward; c4_net_forward_wave and the self-play kernels), net_forward_wave16q (32 filters,
// reference precision), net_forward_wave16w (64 filters).  All live in c4_net_dev.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/c4_engine.h"

#include "c4_net_dev.h"

namespace {

using namespace c4net;

__global__ __launch_bounds__(NTHREADS) void c4_net_kernel(NetDev nd, const uint64_t *__restrict__ c0,
                                                          const uint64_t *__restrict__ c1, int n,
                                                          float *__restrict__ values, float *__restrict__ priors)
{
    __shared__ __attribute__((aligned(16))) _Float16 lds[2][(ROWS + 1) * CS];
    __shared__ __attribute__((aligned(16))) half8 wbuf[2][WCHUNKS];
    __shared__ __attribute__((aligned(16))) float4 mlp[MLP_F4];
    net_forward_block(nd, NetLds{lds, wbuf, mlp}, c0, c1, n, blockIdx.x * P, values, priors);
}

// One-position wave-private forward (net_forward_wave16q: reference precision at 32 filters; net_forward_wave16w: fp16 at
// 64 filters): one position per wave.  Both c4_net_forward and c4_net_forward_wave run this kernel for such a net (one
// implementation, so the two entry points and the fused self-play kernel cannot disagree).
template <int MODE>
__global__ __launch_bounds__(NTHREADS) void c4_net_wave1_kernel(NetDev nd, const uint64_t *__restrict__ c0,
                                                                const uint64_t *__restrict__ c1, int n,
                                                                float *__restrict__ values, float *__restrict__ priors, int active_waves)
{
    __shared__ __attribute__((aligned(16))) _Float16 act[NWAVES][WaveBuf<MODE>::HALVES];
    __shared__ __attribute__((aligned(16))) float4 mlp[MLP_F4];
    __shared__ __attribute__((aligned(16))) float s_bias[BIAS_LDS_FLOATS];
    __shared__ __attribute__((aligned(16))) uint16_t s_tab[64 * TAB16];
    for (int i = threadIdx.x; i < MLP_F4; i += NTHREADS) mlp[i] = nd.mlp[i];
    stage_bias_lds(nd, s_bias);
    if (threadIdx.x < 64) build_tab16<MODE == NETMODE_F64 ? CS64 : CS16>(s_tab, threadIdx.x);
    __syncthreads();
    const int wv = threadIdx.x >> 6;
    const int p = blockIdx.x * NWAVES + wv;
    if (p >= n) return;
    if (wv >= active_waves) return;                       // diagnostic (C4_NET_WAVE_ACTIVE): fewer waves per CU
    net_forward_wave1_mode<MODE>(nd, &act[wv][0], mlp, s_bias, s_tab, c0[p], c1[p], values, priors, p, nd.w0,   /* (eight waves' planes fill this kernel's LDS: the pass-start fragments come from L2 here) */
                                 (nd.stamps && blockIdx.x == 0) ? nd.stamps + wv * 16 : nullptr);
}

// net_forward_wave16 as a kernel of its own (what c4_net_forward_wave runs for the 32-filter fp16 net): one
// position per wave, 16-row MFMA tiles -- the very function the self-play kernel evaluates its leaves with.
__global__ __launch_bounds__(NTHREADS) void c4_net_wave16_kernel(NetDev nd, const uint64_t *__restrict__ c0,
                                                                 const uint64_t *__restrict__ c1, int n,
                                                                 float *__restrict__ values, float *__restrict__ priors, int active_waves)
{
    __shared__ __attribute__((aligned(16))) _Float16 act[NWAVES][2][PLANE16];
    __shared__ __attribute__((aligned(16))) float4 mlp[MLP_F4];
    __shared__ __attribute__((aligned(16))) float s_bias[BIAS_LDS_FLOATS];
    __shared__ __attribute__((aligned(16))) uint16_t s_tab[64 * TAB16];
    for (int i = threadIdx.x; i < MLP_F4; i += NTHREADS) mlp[i] = nd.mlp[i];
    stage_bias_lds(nd, s_bias);
    if (threadIdx.x < 64) build_tab16(s_tab, threadIdx.x);
    __syncthreads();
    const int wv = threadIdx.x >> 6;
    const int p = blockIdx.x * NWAVES + wv;
    if (p >= n) return;
    if (wv >= active_waves) return;                       // diagnostic (C4_NET_WAVE_ACTIVE): fewer waves per CU
    net_forward_wave16(nd, &act[wv][0][0], mlp, s_bias, s_tab, c0[p], c1[p], values, priors, p,
                       (nd.stamps && blockIdx.x == 0) ? nd.stamps + wv * 16 : nullptr);
}

thread_local char n_err[512] = "";

}  // namespace

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

/* diagnostic: copy the s_memtime stamps of block 0 (needs C4_NET_STAMPS=1 at create time) */
int c4_net_debug_stamps(c4_net *net, unsigned long long *out /* [8][16] */)
{
    if (!net || !out || !net->d.stamps) return C4_ESTATE;
    if (hipDeviceSynchronize() != hipSuccess) return C4_EDEVICE;
    return hipMemcpy(out, net->d.stamps, 8 * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? C4_OK : C4_EDEVICE;
}

int c4_net_create(int device, const c4_net_desc *desc, c4_net **out)
{
    if (!desc || !out) { snprintf(n_err, 512, "c4_net_create: null argument"); return C4_EINVAL; }
    *out = nullptr;
    const int FW = desc->filters;
    if ((FW != 32 && FW != 64) || desc->channels != 3 || desc->n_residuals < 0 ||
        FW * (1 + 2 * desc->n_residuals) > BIAS_LDS_FLOATS) {
        snprintf(n_err, 512, "fused net supports channels=3, filters 32 (<= 16 residual blocks) or 64 (<= 7) "
                 "(got channels=%d filters=%d residuals=%d)", desc->channels, desc->filters, desc->n_residuals);
        return C4_EINVAL;
    }
    if (desc->precision != C4_NET_F16 && desc->precision != C4_NET_F32X3) {
        snprintf(n_err, 512, "c4_net_create: unknown precision %d", desc->precision);
        return C4_EINVAL;
    }
    if (FW != 32 && desc->precision == C4_NET_F32X3) {
        snprintf(n_err, 512, "c4_net_create: the reference-precision forward is offered for 32 filters only "
                 "(the hi/lo planes of %d filters do not fit a wave's private LDS)", FW);
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
    const int CB = FW / 32, KPT = FW / 16, KS = 9 * KPT;   // cout blocks, k-steps per tap, k-steps per layer
    // (32x32x16 order: what net_forward_block, the 32-filter block kernel, reads)
    // A fragments in MFMA lane order: fragment (k-step s, cout block cb) at [(s * CB + cb) * 64 + lane][8]; lane l
    // holds cout 32 cb + (l & 31) and the 8 k's 16 s + 8 (l >> 5) + j.
    // ---- stem: k = tap*4 + ch (ch 3 zero), 48 = 3 k-steps
    std::vector<_Float16> stem((size_t)3 * CB * 64 * 8), conv((size_t)2 * R * KS * CB * 64 * 8), head((size_t)KPT * 64 * 8);
    // reference-precision mode: w ~= hi + lo / 2^11 with hi = f16(w), lo = f16((w - hi) * 2^11)
    auto split = [](float v, _Float16 &hi, _Float16 &lo) {
        hi = (_Float16)v;
        lo = (_Float16)((v - (float)hi) * LO_SCALE);
    };
    for (int s = 0; s < 3; ++s)
        for (int cb = 0; cb < CB; ++cb)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int k = 16 * s + 8 * (l >> 5) + j, tap = k >> 2, ch = k & 3, co = 32 * cb + (l & 31);
                    float v = 0.0f;
                    if (tap < 9 && ch < 3) v = desc->stem_w[((co * 3 + ch) * 3 + tap / 3) * 3 + tap % 3];
                    const size_t at = (((size_t)s * CB + cb) * 64 + l) * 8 + j;
                    stem[at] = (_Float16)v;
                }
    // ---- 3x3 convs: k-step s: tap = s / KPT, cin = (s % KPT)*16 + 8(l>>5) + j
    for (int L = 0; L < 2 * R; ++L)
        for (int s = 0; s < KS; ++s)
            for (int cb = 0; cb < CB; ++cb)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int tap = s / KPT, ci = (s % KPT) * 16 + 8 * (l >> 5) + j, co = 32 * cb + (l & 31);
                        const float v = desc->conv_w[((((size_t)L * FW + co) * FW + ci) * 3 + tap / 3) * 3 + tap % 3];
                        const size_t at = ((((size_t)L * KS + s) * CB + cb) * 64 + l) * 8 + j;
                        conv[at] = (_Float16)v;
                    }
    // ---- head 1x1: couts 0..2 (value, policy0, policy1), cin = 16s + 8(l>>5) + j
    for (int s = 0; s < KPT; ++s)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int ci = 16 * s + 8 * (l >> 5) + j, co = l & 31;
                head[((size_t)s * 64 + l) * 8 + j] = (_Float16)(co < 3 ? desc->head_w[co * FW + ci] : 0.0f);
            }
    // ---- the same weights in v_mfma_f32_16x16x32_f16 fragment order (the wave-private forwards, 32 and 64 filters):
    //      lane l holds cout 16 ct + (l & 15) and the 8 k's 8 (l >> 4) + j of the k-step
    std::vector<_Float16> stem16, conv16, head16, stem16l, conv16l, head16l;   // ...l: the scaled low parts
    {
        const int CT = FW / 16, KS2 = FW / 32;   // cout tiles of 16, k-steps of 32 input channels per tap
        stem16.assign((size_t)2 * CT * 64 * 8, (_Float16)0.0f);                              // [k-step s][ct]: k = 32 s + 8 g + j = tap*4 + ch
        conv16.assign((size_t)std::max(1, 2 * R) * 9 * KS2 * CT * 64 * 8, (_Float16)0.0f);   // [L][tap][ks][ct]: cin = 32 ks + 8 g + j
        head16.assign((size_t)KS2 * 64 * 8, (_Float16)0.0f);                                 // [ks]: couts 0..2, cin = 32 ks + 8 g + j
        stem16l = stem16; conv16l = conv16; head16l = head16;
        for (int s2 = 0; s2 < 2; ++s2)
            for (int ct = 0; ct < CT; ++ct)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int k = 32 * s2 + 8 * (l >> 4) + j, tap = k >> 2, ch = k & 3, co = 16 * ct + (l & 15);
                        const size_t at = (((size_t)s2 * CT + ct) * 64 + l) * 8 + j;
                        if (tap < 9 && ch < 3) split(desc->stem_w[((co * 3 + ch) * 3 + tap / 3) * 3 + tap % 3], stem16[at], stem16l[at]);
                    }
        for (int L = 0; L < 2 * R; ++L)
            for (int tap = 0; tap < 9; ++tap)
                for (int ks = 0; ks < KS2; ++ks)
                    for (int ct = 0; ct < CT; ++ct)
                        for (int l = 0; l < 64; ++l)
                            for (int j = 0; j < 8; ++j) {
                                const int ci = 32 * ks + 8 * (l >> 4) + j, co = 16 * ct + (l & 15);
                                const size_t at = (((((size_t)L * 9 + tap) * KS2 + ks) * CT + ct) * 64 + l) * 8 + j;
                                split(desc->conv_w[((((size_t)L * FW + co) * FW + ci) * 3 + tap / 3) * 3 + tap % 3], conv16[at], conv16l[at]);
                            }
        for (int ks = 0; ks < KS2; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int ci = 32 * ks + 8 * (l >> 4) + j, co = l & 15;
                    const size_t at = ((size_t)ks * 64 + l) * 8 + j;
                    if (co < 3) split(desc->head_w[co * FW + ci], head16[at], head16l[at]);
                }
    }
    // the 32-filter tower as one linear stream for net_forward_wave16q: [tap T = L * 9 + t][ct0 hi | ct0 lo | ct1 hi | ct1 lo][64][8]
    std::vector<_Float16> conv16p;
    if (FW == 32 && R > 0) {
        conv16p.resize((size_t)2 * R * 9 * 4 * 64 * 8);
        for (size_t T = 0; T < (size_t)2 * R * 9; ++T)
            for (int ct = 0; ct < 2; ++ct)
                for (int e = 0; e < 64 * 8; ++e) {
                    conv16p[((T * 4 + 2 * ct) * 64 * 8) + e] = conv16[((T * 2 + ct) * 64 * 8) + e];
                    conv16p[((T * 4 + 2 * ct + 1) * 64 * 8) + e] = conv16l[((T * 2 + ct) * 64 * 8) + e];
                }
    } else conv16p.resize(8);
    // ... and what a pass needs first, in the order of c4net::W0_*: stem hi (4 fragments), stem lo (4), heads hi, heads lo, tower taps 0 and 1
    std::vector<_Float16> w0v((size_t)W0_FRAGS * 64 * 8, (_Float16)0.0f);
    if (FW == 32) {
        std::copy(stem16.begin(), stem16.begin() + 4 * 64 * 8, w0v.begin() + (size_t)W0_STEM * 64 * 8);
        std::copy(stem16l.begin(), stem16l.begin() + 4 * 64 * 8, w0v.begin() + (size_t)(W0_STEM + 4) * 64 * 8);
        std::copy(head16.begin(), head16.begin() + 64 * 8, w0v.begin() + (size_t)W0_HEAD * 64 * 8);
        std::copy(head16l.begin(), head16l.begin() + 64 * 8, w0v.begin() + (size_t)(W0_HEAD + 1) * 64 * 8);
        if (R > 0) std::copy(conv16p.begin(), conv16p.begin() + 8 * 64 * 8, w0v.begin() + (size_t)W0_TOWER * 64 * 8);
    }
    std::vector<float> stem_b(desc->stem_b, desc->stem_b + FW), conv_b(desc->conv_b, desc->conv_b + (size_t)2 * R * FW),
        head_b(4, 0.0f), mlp((size_t)MLP_F4 * 4, 0.0f);
    for (int i = 0; i < 3; ++i) head_b[i] = desc->head_b[i];
    {
        float *vt = mlp.data();                     // [11][64][4]: input 4g+c of value row `lane`
        float *pt = vt + (size_t)VT_F4 * 4;         // [11][64]: input seg*11+c of logit lane&7, seg = lane>>3
        float *fb = pt + PT_F, *vw = fb + 64, *pb = vw + 64;
        for (int o = 0; o < PIX; ++o) {
            for (int i = 0; i < PIX; ++i) vt[((i / 4) * 64 + o) * 4 + (i % 4)] = desc->vfc_w[o * PIX + i];
            fb[o] = desc->vfc_b[o];
            vw[o] = desc->vout_w[o];
        }
        for (int l = 0; l < 64; ++l) {
            const int o = l & 7, seg = l >> 3;
            if (o >= 7) continue;
            for (int c = 0; c < 11; ++c) {
                const int j = seg * 11 + c;
                if (j < 2 * PIX) pt[c * 64 + l] = desc->pfc_w[o * 2 * PIX + j];
            }
            if (seg == 0) pb[l] = desc->pfc_b[o];
        }
    }
    if (conv.empty()) conv.resize(8);
    if (conv_b.empty()) conv_b.resize(4);
    hipError_t r = hipSuccess;
    const _Float16 *p16;
#define UP16(vec, field) if (r == hipSuccess) { r = upload(net, vec, &p16); net->d.field = (const half8 *)p16; }
#define UP32(vec, field) if (r == hipSuccess) r = upload(net, vec, &net->d.field);
    UP16(stem, stem_w) UP16(conv, conv_w) UP16(head, head_w)
    UP16(stem16, stem_w16) UP16(conv16, conv_w16) UP16(head16, head_w16)
    UP16(stem16l, stem_w16l) UP16(conv16l, conv_w16l) UP16(head16l, head_w16l)
    UP16(conv16p, conv_w16p) UP16(w0v, w0)
    net->d.conv_w16p_bytes = (unsigned)(conv16p.size() * sizeof(_Float16));
    UP32(stem_b, stem_b) UP32(conv_b, conv_b) UP32(head_b, head_b)
    {
        const float *pm = nullptr;
        if (r == hipSuccess) r = upload(net, mlp, &pm);
        net->d.mlp = reinterpret_cast<const float4 *>(pm);
    }
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
    net->d.precise = desc->precision == C4_NET_F32X3 ? 1 : 0;
    net->d.filters = FW;
    net->d.mode = FW == 64 ? NETMODE_F64 : (net->d.precise ? NETMODE_F32_PRECISE : NETMODE_F32_F16);
    if (getenv("C4_NET_STAMPS")) {
        void *q = nullptr;
        if (hipMalloc(&q, 8 * 16 * sizeof(unsigned long long)) == hipSuccess) {
            (void)hipMemset(q, 0, 8 * 16 * sizeof(unsigned long long));
            net->allocs.push_back(q);
            net->d.stamps = (unsigned long long *)q;
        }
    }
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
    if (net->d.mode != NETMODE_F32_F16) return c4_net_forward_wave(net, hip_stream, color0_dev, color1_dev, n, values_dev, priors_dev);
    const dim3 grid((n + P - 1) / P), block(NTHREADS);
    hipLaunchKernelGGL(c4_net_kernel, grid, block, 0, (hipStream_t)hip_stream, net->d, color0_dev, color1_dev, (int)n,
                       values_dev, priors_dev);
    hipError_t r = hipGetLastError();
    if (r != hipSuccess) {
        snprintf(n_err, 512, "c4_net_kernel launch failed: %s", hipGetErrorString(r));
        return C4_EDEVICE;
    }
    return C4_OK;
}

/* Same contract as c4_net_forward, evaluated by the wave-private forward (one position per wave, no
 * workgroup barrier; bit-identical answers). */
int c4_net_forward_wave(c4_net *net, void *hip_stream, const uint64_t *color0_dev, const uint64_t *color1_dev, int32_t n,
                        float *values_dev, float *priors_dev)
{
    if (!net || !color0_dev || !color1_dev || !values_dev || !priors_dev || n < 0) {
        snprintf(n_err, 512, "c4_net_forward_wave: bad argument");
        return C4_EINVAL;
    }
    if (n == 0) return C4_OK;
    const char *ea16 = getenv("C4_NET_WAVE_ACTIVE");   // timing experiments only: fewer waves per CU
    const int active = ea16 ? atoi(ea16) : NWAVES;
    if (net->d.mode != NETMODE_F32_F16) {
        const dim3 g1((n + NWAVES - 1) / NWAVES), b1(NTHREADS);
        if (net->d.mode == NETMODE_F64)
            hipLaunchKernelGGL(c4_net_wave1_kernel<NETMODE_F64>, g1, b1, 0, (hipStream_t)hip_stream, net->d, color0_dev, color1_dev, (int)n, values_dev, priors_dev, active);
        else
            hipLaunchKernelGGL(c4_net_wave1_kernel<NETMODE_F32_PRECISE>, g1, b1, 0, (hipStream_t)hip_stream, net->d, color0_dev, color1_dev, (int)n, values_dev, priors_dev, active);
        hipError_t pr = hipGetLastError();
        if (pr != hipSuccess) {
            snprintf(n_err, 512, "c4_net_wave1_kernel launch failed: %s", hipGetErrorString(pr));
            return C4_EDEVICE;
        }
        return C4_OK;
    }
    hipLaunchKernelGGL(c4_net_wave16_kernel, dim3((n + NWAVES - 1) / NWAVES), dim3(NTHREADS), 0, (hipStream_t)hip_stream, net->d,
                       color0_dev, color1_dev, (int)n, values_dev, priors_dev, active);
    hipError_t r = hipGetLastError();
    if (r != hipSuccess) {
        snprintf(n_err, 512, "c4_net_wave16_kernel launch failed: %s", hipGetErrorString(r));
        return C4_EDEVICE;
    }
    return C4_OK;
}

}  // extern "C"

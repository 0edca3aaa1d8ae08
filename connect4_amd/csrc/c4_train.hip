// c4_train.hip -- the two kernels of the reference's train step that stock MIOpen spends half of it in: batch
// normalisation in training mode, forward and backward, fused with the residual add and the LeakyReLU that follow it
// in the reference's net (model.py:20-31 conv-bn-act, :36-55 residual block, :60-117 heads; the train step itself:
// model.py:200-240).  float32 NCHW tensors [rows][channels][hw] (hw = 42), contiguous.
//
//   forward   y = act(bn(x) + residual),  bn(x) = (x - mean_c) * invstd_c * weight_c + bias_c,
//             mean / biased variance over the first `valid_rows` rows (all rows but for the padded ragged batch of an
//             epoch, net._BatchNorm2d), running statistics updated as torch.nn.BatchNorm2d does (momentum, unbiased
//             variance), act = LeakyReLU(slope) (slope 1 = none);
//   backward  dz = dy * (y > 0 ? 1 : slope);  dresidual = dz;  dbias = sum dz;  dweight = sum dz * xhat;
//             dx = weight * invstd * (dz - dbias / M - xhat * dweight / M)  (rows beyond valid_rows: weight * invstd * dz).
//
// HBM bound: the forward reads x twice (statistics: a chunk's values stay in registers between its sum and its centred
// squares, chunks are combined exactly in float64 -- the accuracy of a two-pass variance; apply) and writes y; the backward reads x, y, dy twice and writes dx
// (+ dresidual).  One workgroup = one channel x one chunk of 64 rows; thread t owns (row t / hw, pixel t % hw) of six rows
// at a time, so a wave reads whole 168-byte rows.  Every reduction is a fixed tree (threads -> wave -> block -> chunks
// in float64): results are bit-reproducible from run to run.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/c4_engine.h"

namespace {

constexpr int TB = 256;            // threads per workgroup
constexpr int ROWS_PER_CHUNK = 64;
constexpr int MAX_CHUNKS = 1024;

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// sum over the workgroup, the same value in every thread (fixed order: lanes by butterfly, then waves 0..3)
__device__ __forceinline__ float block_sum(float v, float *s4)
{
    v = wave_sum(v);
    __syncthreads();               // s4 may still be read from the previous call
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((s4[0] + s4[1]) + s4[2]) + s4[3];
}

// total of a channel's per-chunk partials (float64, chunk order), the same value in every thread
__device__ __forceinline__ double chunk_total(const float *part, int chunks, double *s1)
{
    __syncthreads();
    if (threadIdx.x < 64) {
        double a = 0.0;
        for (int i = threadIdx.x; i < chunks; i += 64) a += (double)part[i];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
        if (threadIdx.x == 0) *s1 = a;
    }
    __syncthreads();
    return *s1;
}

struct Geo {
    int rows, valid_rows, channels, hw, chunks, rows_per_chunk;
    int block_centred;   // forward: the workspace's squares are centred on each chunk's own mean (bn_stats_kernel)
};

// ws layout: [0] sums, [1] centred squares (forward) / [0] sum dz, [1] sum dz*xhat (backward): each [channels][chunks]
__global__ __launch_bounds__(TB) void bn_sum_kernel(const float *__restrict__ x, float *__restrict__ ws, Geo g)
{
    __shared__ float s4[4];
    const int c = blockIdx.x, ch = blockIdx.y;
    const int rpb = TB / g.hw, tr = threadIdx.x / g.hw, p = threadIdx.x - tr * g.hw;
    const int r0 = ch * g.rows_per_chunk, r1 = min(r0 + g.rows_per_chunk, g.valid_rows);
    float a = 0.0f;
    if (tr < rpb)
        for (int r = r0 + tr; r < r1; r += rpb) a += x[((size_t)r * g.channels + c) * g.hw + p];
    a = block_sum(a, s4);
    if (threadIdx.x == 0) ws[(size_t)c * g.chunks + ch] = a;
}

// One read instead of two when a chunk's rows fit a thread's registers (64 rows: eleven values per thread): the workgroup's own
// mean first, then the squares centred on IT from the registers; the chunks are combined exactly afterwards (stats_total):
// sum_b (x - mean)^2 = M2_b + n_b (mean_b - mean)^2.
constexpr int REG_MIN_RPB = 6, REG_VALUES = (ROWS_PER_CHUNK + REG_MIN_RPB - 1) / REG_MIN_RPB;   // hw <= 42: six rows at a time, eleven values
__global__ __launch_bounds__(TB) void bn_stats_kernel(const float *__restrict__ x, float *__restrict__ ws, Geo g)
{
    __shared__ float s4[4];
    const int c = blockIdx.x, ch = blockIdx.y;
    const int rpb = TB / g.hw, tr = threadIdx.x / g.hw, p = threadIdx.x - tr * g.hw;
    const int r0 = ch * g.rows_per_chunk, r1 = min(r0 + g.rows_per_chunk, g.valid_rows);
    constexpr int MAXV = REG_VALUES;
    float v[MAXV];
    float a = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int r = r0 + tr + i * rpb;
        const bool in = tr < rpb && r < r1;
        v[i] = in ? x[((size_t)r * g.channels + c) * g.hw + p] : 0.0f;
        a += v[i];
    }
    a = block_sum(a, s4);
    const int nb = max(r1 - r0, 0) * g.hw;
    const float mb = nb > 0 ? a / (float)nb : 0.0f;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int r = r0 + tr + i * rpb;
        const bool in = tr < rpb && r < r1;
        const float d = v[i] - mb;
        q += in ? d * d : 0.0f;
    }
    q = block_sum(q, s4);
    if (threadIdx.x == 0) {
        ws[(size_t)c * g.chunks + ch] = a;
        ws[(size_t)(g.channels + c) * g.chunks + ch] = q;
    }
}

__global__ __launch_bounds__(TB) void bn_var_kernel(const float *__restrict__ x, float *__restrict__ ws, Geo g)
{
    __shared__ float s4[4];
    __shared__ double s1;
    const int c = blockIdx.x, ch = blockIdx.y;
    const double M = (double)g.valid_rows * g.hw;
    const float mean = (float)(chunk_total(ws + (size_t)c * g.chunks, g.chunks, &s1) / M);
    const int rpb = TB / g.hw, tr = threadIdx.x / g.hw, p = threadIdx.x - tr * g.hw;
    const int r0 = ch * g.rows_per_chunk, r1 = min(r0 + g.rows_per_chunk, g.valid_rows);
    float a = 0.0f;
    if (tr < rpb)
        for (int r = r0 + tr; r < r1; r += rpb) {
            const float d = x[((size_t)r * g.channels + c) * g.hw + p] - mean;
            a += d * d;
        }
    a = block_sum(a, s4);
    if (threadIdx.x == 0) ws[(size_t)(g.channels + c) * g.chunks + ch] = a;
}

__global__ __launch_bounds__(TB) void bn_apply_kernel(const float *__restrict__ x, const float *__restrict__ res,
                                                      const float *__restrict__ weight, const float *__restrict__ bias,
                                                      float *running_mean, float *running_var, long long *nbt,
                                                      float *__restrict__ y, float *save_mean, float *save_invstd,
                                                      const float *__restrict__ ws, Geo g, float momentum, float eps, float slope)
{
    __shared__ double s1;
    const int c = blockIdx.x, ch = blockIdx.y;
    const double M = (double)g.valid_rows * g.hw;
    const float mean = (float)(chunk_total(ws + (size_t)c * g.chunks, g.chunks, &s1) / M);
    double ss = chunk_total(ws + (size_t)(g.channels + c) * g.chunks, g.chunks, &s1);
    if (g.block_centred) {   // + sum_b n_b (mean_b - mean)^2, in float64 and chunk order
        __syncthreads();
        if (threadIdx.x < 64) {
            double a = 0.0;
            for (int i = threadIdx.x; i < g.chunks; i += 64) {
                const int nb = max(min((i + 1) * g.rows_per_chunk, g.valid_rows) - i * g.rows_per_chunk, 0) * g.hw;
                if (nb > 0) {
                    const double dm = (double)ws[(size_t)c * g.chunks + i] / nb - (double)mean;
                    a += nb * dm * dm;
                }
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
            if (threadIdx.x == 0) s1 = a;
        }
        __syncthreads();
        ss += s1;
    }
    const float var = (float)(ss / M);
    const float invstd = 1.0f / sqrtf(var + eps);
    if (ch == 0 && threadIdx.x == 0) {
        save_mean[c] = mean;
        save_invstd[c] = invstd;
        if (running_mean) {   // torch.nn.BatchNorm2d: running = (1 - momentum) * running + momentum * batch (unbiased variance)
            const float unb = (float)(ss / (M - 1.0));
            running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unb;
        }
        if (nbt && c == 0) *nbt += 1;
    }
    const float w = weight[c], b = bias[c];
    const int rpb = TB / g.hw, tr = threadIdx.x / g.hw, p = threadIdx.x - tr * g.hw;
    const int r0 = ch * g.rows_per_chunk, r1 = min(r0 + g.rows_per_chunk, g.rows);
    if (tr < rpb)
        for (int r = r0 + tr; r < r1; r += rpb) {
            const size_t i = ((size_t)r * g.channels + c) * g.hw + p;
            float z = (x[i] - mean) * invstd * w + b;
            if (res) z += res[i];
            y[i] = z > 0.0f ? z : z * slope;
        }
}

__global__ __launch_bounds__(TB) void bn_bwd_reduce_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                           const float *__restrict__ dy, const float *__restrict__ save_mean,
                                                           const float *__restrict__ save_invstd, float *__restrict__ ws, Geo g,
                                                           float slope)
{
    __shared__ float s4[4];
    const int c = blockIdx.x, ch = blockIdx.y;
    const float mean = save_mean[c], invstd = save_invstd[c];
    const int rpb = TB / g.hw, tr = threadIdx.x / g.hw, p = threadIdx.x - tr * g.hw;
    const int r0 = ch * g.rows_per_chunk, r1 = min(r0 + g.rows_per_chunk, g.valid_rows);
    float a = 0.0f, bq = 0.0f;
    if (tr < rpb)
        for (int r = r0 + tr; r < r1; r += rpb) {
            const size_t i = ((size_t)r * g.channels + c) * g.hw + p;
            const float dz = y[i] > 0.0f ? dy[i] : dy[i] * slope;
            a += dz;
            bq += dz * ((x[i] - mean) * invstd);
        }
    a = block_sum(a, s4);
    bq = block_sum(bq, s4);
    if (threadIdx.x == 0) {
        ws[(size_t)c * g.chunks + ch] = a;
        ws[(size_t)(g.channels + c) * g.chunks + ch] = bq;
    }
}

__global__ __launch_bounds__(TB) void bn_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                          const float *__restrict__ dy, const float *__restrict__ weight,
                                                          const float *__restrict__ save_mean, const float *__restrict__ save_invstd,
                                                          float *__restrict__ dx, float *__restrict__ dres, float *dweight, float *dbias,
                                                          const float *__restrict__ ws, Geo g, float slope)
{
    __shared__ double s1;
    const int c = blockIdx.x, ch = blockIdx.y;
    const double M = (double)g.valid_rows * g.hw;
    const double sdz = chunk_total(ws + (size_t)c * g.chunks, g.chunks, &s1);
    const double sdzx = chunk_total(ws + (size_t)(g.channels + c) * g.chunks, g.chunks, &s1);
    if (ch == 0 && threadIdx.x == 0) {
        dbias[c] = (float)sdz;
        dweight[c] = (float)sdzx;
    }
    const float mean = save_mean[c], invstd = save_invstd[c];
    const float k = weight[c] * invstd, mdz = (float)(sdz / M), mdzx = (float)(sdzx / M);
    const int rpb = TB / g.hw, tr = threadIdx.x / g.hw, p = threadIdx.x - tr * g.hw;
    const int r0 = ch * g.rows_per_chunk, r1 = min(r0 + g.rows_per_chunk, g.rows);
    if (tr < rpb)
        for (int r = r0 + tr; r < r1; r += rpb) {
            const size_t i = ((size_t)r * g.channels + c) * g.hw + p;
            const float dz = y[i] > 0.0f ? dy[i] : dy[i] * slope;
            if (dres) dres[i] = dz;
            const float xh = (x[i] - mean) * invstd;
            dx[i] = r < g.valid_rows ? k * (dz - mdz - xh * mdzx) : k * dz;
        }
}

bool make_geo(Geo &g, int rows, int valid_rows, int channels, int hw)
{
    if (rows <= 0 || channels <= 0 || hw <= 0 || hw > TB || valid_rows <= 0 || valid_rows > rows || (long long)valid_rows * hw < 2) return false;
    g.rows = rows; g.valid_rows = valid_rows; g.channels = channels; g.hw = hw; g.block_centred = 0;
    g.rows_per_chunk = ROWS_PER_CHUNK;
    while ((rows + g.rows_per_chunk - 1) / g.rows_per_chunk > MAX_CHUNKS) g.rows_per_chunk *= 2;
    g.chunks = (rows + g.rows_per_chunk - 1) / g.rows_per_chunk;
    return channels <= 65535 && g.chunks <= 65535;
}

}  // namespace

extern "C" {

long long c4_bn_workspace_floats(int rows, int channels)
{
    Geo g;
    if (!make_geo(g, rows, rows, channels, 2)) return C4_EINVAL;   // (any hw >= 2: the workspace does not depend on it; one row is a valid batch)
    return 2LL * channels * g.chunks;
}

int c4_bn_train_forward(const float *x_dev, const float *residual_dev, const float *weight_dev, const float *bias_dev,
                        float *running_mean_dev, float *running_var_dev, long long *num_batches_tracked_dev, float *y_dev,
                        float *save_mean_dev, float *save_invstd_dev, float *workspace_dev, int rows, int valid_rows, int channels,
                        int hw, float momentum, float eps, float slope, void *hip_stream)
{
    Geo g;
    if (!x_dev || !weight_dev || !bias_dev || !y_dev || !save_mean_dev || !save_invstd_dev || !workspace_dev || !make_geo(g, rows, valid_rows, channels, hw) ||
        (running_mean_dev == nullptr) != (running_var_dev == nullptr))
        return C4_EINVAL;
    hipStream_t s = (hipStream_t)hip_stream;
    const dim3 grid(channels, g.chunks);
    if (g.rows_per_chunk == ROWS_PER_CHUNK && TB / hw >= REG_MIN_RPB) {   // one read of x for both statistics
        g.block_centred = 1;
        bn_stats_kernel<<<grid, TB, 0, s>>>(x_dev, workspace_dev, g);
    } else {
        bn_sum_kernel<<<grid, TB, 0, s>>>(x_dev, workspace_dev, g);
        bn_var_kernel<<<grid, TB, 0, s>>>(x_dev, workspace_dev, g);
    }
    bn_apply_kernel<<<grid, TB, 0, s>>>(x_dev, residual_dev, weight_dev, bias_dev, running_mean_dev, running_var_dev, num_batches_tracked_dev, y_dev,
                                        save_mean_dev, save_invstd_dev, workspace_dev, g, momentum, eps, slope);
    return hipGetLastError() == hipSuccess ? C4_OK : C4_EDEVICE;
}

int c4_bn_train_backward(const float *x_dev, const float *y_dev, const float *dy_dev, const float *weight_dev, const float *save_mean_dev,
                         const float *save_invstd_dev, float *dx_dev, float *dresidual_dev, float *dweight_dev, float *dbias_dev,
                         float *workspace_dev, int rows, int valid_rows, int channels, int hw, float slope, void *hip_stream)
{
    Geo g;
    if (!x_dev || !y_dev || !dy_dev || !weight_dev || !save_mean_dev || !save_invstd_dev || !dx_dev || !dweight_dev || !dbias_dev || !workspace_dev ||
        !make_geo(g, rows, valid_rows, channels, hw))
        return C4_EINVAL;
    hipStream_t s = (hipStream_t)hip_stream;
    const dim3 grid(channels, g.chunks);
    bn_bwd_reduce_kernel<<<grid, TB, 0, s>>>(x_dev, y_dev, dy_dev, save_mean_dev, save_invstd_dev, workspace_dev, g, slope);
    bn_bwd_apply_kernel<<<grid, TB, 0, s>>>(x_dev, y_dev, dy_dev, weight_dev, save_mean_dev, save_invstd_dev, dx_dev, dresidual_dev, dweight_dev, dbias_dev,
                                            workspace_dev, g, slope);
    return hipGetLastError() == hipSuccess ? C4_OK : C4_EDEVICE;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// c4_conv3x3_wrw: the weight gradient of the tower's convolutions (model.py:36-55: 3x3, 32 -> 32 filters, padding 1, no bias)
//     dW[co][ci][ky][kx] = sum over rows n and pixels (y, x) of  dy[n][co][y][x] * x[n][ci][y + ky - 1][x + kx - 1]
// float32 NCHW [rows][32][6][7], as autograd hands them over.  MIOpen's pick for this shape is an NHWC implicit GEMM
// (46 us) behind two layout transposes (2 x 14 us) per layer; this is one pass over x and dy on the f32-input MFMA
// (v_mfma_f32_32x32x2_f32: exact float32 products, fma chain): A = dy (32 couts x 2 pixels), B = x shifted by the tap
// (2 pixels x 32 cins), nine accumulator tiles of 32 x 32 per wave, one per tap.  A wave stages one row (position) at a time
// in LDS -- x zero-padded to 8 x 9 per channel so that a tap is an address offset -- while the next row's 10.7 KB are on
// their way into registers; 21 pixel pairs x 9 taps = 189 MFMAs of 64 cycles per row, 4096 rows over 1024 waves: MFMA bound
// at ~21 us.  Partials per workgroup, summed over workgroups in a fixed order by a second kernel: bit-reproducible.
// ------------------------------------------------------------------------------------------------
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int WC = 32, WPIX = 42, WROW = WC * WPIX;          // channels, pixels, floats per row of a tensor
constexpr int XSTRIDE = 73;                                    // floats per channel of the padded plane (8 x 9 = 72, +1: conflict-free across channels)
constexpr int XPLANE = WC * XSTRIDE + 8;                       // floats of one wave's padded x
constexpr int WRW_WAVES = 4;
constexpr int WRW_F4 = WROW / 4;                               // 336 float4 per row: 5.25 per lane

__global__ __launch_bounds__(WRW_WAVES * 64) void conv3x3_wrw_partial_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                                             float *__restrict__ partial, int rows)
{
    __shared__ __attribute__((aligned(16))) float s_x[WRW_WAVES][XPLANE];
    __shared__ __attribute__((aligned(16))) float s_dy[WRW_WAVES][WROW];
    __shared__ float s_red[WRW_WAVES][1024];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wave_id = blockIdx.x * WRW_WAVES + wv, n_waves = gridDim.x * WRW_WAVES;
    float *const xs = s_x[wv];
    float *const ds = s_dy[wv];
    for (int i = lane; i < XPLANE; i += 64) xs[i] = 0.0f;     // the borders stay zero for the whole launch
    // where this lane's six float4 of a row go in the padded plane (element e = ci * 42 + y * 7 + x -> ci * 73 + (y + 1) * 9 + x + 1)
    int xoff[6][4];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = (lane + 64 * j) * 4 + q;
            const int ci = e / WPIX, r = e - ci * WPIX, yy = r / 7, xx = r - yy * 7;
            xoff[j][q] = ci * XSTRIDE + (yy + 1) * 9 + xx + 1;
        }
    floatx16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    const int co = lane & 31, kh = lane >> 5;     // this lane's operand row / column and which pixel of the pair
    // a row's 336 float4 per tensor: six per lane, the sixth only for lanes 0..15 (the others re-read the row's last one and drop it)
    float4 px0, px1, px2, px3, px4, px5, pd0, pd1, pd2, pd3, pd4, pd5;
    const int last = lane < WRW_F4 - 320 ? lane + 320 : WRW_F4 - 1;
#define C4_WRW_FETCH(N)                                                                                   \
    do {                                                                                                  \
        const float4 *gx = reinterpret_cast<const float4 *>(x + (size_t)(N) * WROW);                      \
        const float4 *gd = reinterpret_cast<const float4 *>(dy + (size_t)(N) * WROW);                     \
        px0 = gx[lane]; px1 = gx[lane + 64]; px2 = gx[lane + 128]; px3 = gx[lane + 192]; px4 = gx[lane + 256]; px5 = gx[last]; \
        pd0 = gd[lane]; pd1 = gd[lane + 64]; pd2 = gd[lane + 128]; pd3 = gd[lane + 192]; pd4 = gd[lane + 256]; pd5 = gd[last]; \
    } while (0)
    int n = wave_id;
    if (n < rows) C4_WRW_FETCH(n);
    else { px0 = px1 = px2 = px3 = px4 = px5 = pd0 = pd1 = pd2 = pd3 = pd4 = pd5 = float4{0.0f, 0.0f, 0.0f, 0.0f}; }
    for (; n < rows; n += n_waves) {
        // registers -> LDS (the previous row's reads are behind us: a wave's LDS operations execute in order)
#define C4_WRW_STAGE(J, PX, PD)                                                                                     \
    do {                                                                                                           \
        xs[xoff[J][0]] = PX.x; xs[xoff[J][1]] = PX.y; xs[xoff[J][2]] = PX.z; xs[xoff[J][3]] = PX.w;                \
        reinterpret_cast<float4 *>(ds)[lane + 64 * J] = PD;                                                        \
    } while (0)
        C4_WRW_STAGE(0, px0, pd0); C4_WRW_STAGE(1, px1, pd1); C4_WRW_STAGE(2, px2, pd2); C4_WRW_STAGE(3, px3, pd3); C4_WRW_STAGE(4, px4, pd4);
        if (lane < WRW_F4 - 320) C4_WRW_STAGE(5, px5, pd5);
        if (n + n_waves < rows) C4_WRW_FETCH(n + n_waves);   // the next row travels while this one is multiplied
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        int py = 0, pxx = kh;                                  // this lane's pixel of the current pair: (y, x), pixel index 2 p + kh
#pragma unroll
        for (int p = 0; p < WPIX / 2; ++p) {
            const float a = ds[co * WPIX + 2 * p + kh];
            const float *bp = xs + co * XSTRIDE + py * 9 + pxx;     // (the operand's column index is the input channel: same lane field)
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[(t / 3) * 9 + (t % 3)], acc[t], 0, 0, 0);
            pxx += 2;
            if (pxx >= 7) { pxx -= 7; py += 1; }
        }
    }
#undef C4_WRW_FETCH
#undef C4_WRW_STAGE
    // workgroup partial: tap by tap through LDS, waves summed in order
    float *out = partial + (size_t)blockIdx.x * 9 * 1024;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) s_red[wv][((r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + co] = acc[t][r];     // [cout row][cin column]
        __syncthreads();
        for (int i = threadIdx.x; i < 1024; i += WRW_WAVES * 64) out[t * 1024 + i] = ((s_red[0][i] + s_red[1][i]) + s_red[2][i]) + s_red[3][i];
    }
}

// dw = the partials summed in workgroup order: a block owns 32 outputs, eight threads per output take every eighth partial
// (float64), then the eight are added in order
__global__ __launch_bounds__(256) void conv3x3_wrw_reduce_kernel(const float *__restrict__ partial, float *__restrict__ dw, int n_partials)
{
    __shared__ double s_part[8][32];
    const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + o;                        // = tap * 1024 + co * 32 + ci
    double a = 0.0;
    int w = sl;
    for (; w + 56 < n_partials; w += 64) {       // eight loads in flight, added in partial order
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = partial[(size_t)(w + 8 * k) * 9 * 1024 + i];
#pragma unroll
        for (int k = 0; k < 8; ++k) a += (double)v[k];
    }
    for (; w < n_partials; w += 8) a += (double)partial[(size_t)w * 9 * 1024 + i];
    s_part[sl][o] = a;
    __syncthreads();
    if (sl == 0) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += s_part[k][o];
        const int tap = i >> 10, cc = i & 1023;
        dw[cc * 9 + tap] = (float)t;                            // [co][ci][ky][kx]
    }
}

constexpr int WRW_MAX_WGS = 256;

}  // namespace

extern "C" {

long long c4_conv3x3_wrw_workspace_floats(void) { return (long long)WRW_MAX_WGS * 9 * 1024; }

int c4_conv3x3_wrw(const float *x_dev, const float *dy_dev, float *dweight_dev, float *workspace_dev, int rows, int channels, int height, int width,
                   void *hip_stream)
{
    if (!x_dev || !dy_dev || !dweight_dev || !workspace_dev || rows <= 0 || channels != WC || height != 6 || width != 7) return C4_EINVAL;
    hipStream_t s = (hipStream_t)hip_stream;
    const int wgs = rows >= WRW_MAX_WGS * WRW_WAVES ? WRW_MAX_WGS : (rows + WRW_WAVES - 1) / WRW_WAVES;
    conv3x3_wrw_partial_kernel<<<wgs, WRW_WAVES * 64, 0, s>>>(x_dev, dy_dev, workspace_dev, rows);
    conv3x3_wrw_reduce_kernel<<<9 * 1024 / 32, 256, 0, s>>>(workspace_dev, dweight_dev, wgs);
    return hipGetLastError() == hipSuccess ? C4_OK : C4_EDEVICE;
}

}  // extern "C"

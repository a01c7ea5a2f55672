// The glue of EVE.forward either side of the two networks, one frame per thread / workgroup:
//   eve_gaze_to_pog        /root/reference/src/models/common.py:157-187 (to_screen_coordinates), optionally preceded by
//                          :190-229 (apply_offset_augmentation); also emits the 2x2 Jacobians of its three outputs
//                          w.r.t. the gaze angles (forward-mode dual numbers), which is all autograd needs
//   eve_gaze_to_pog_bwd    d(loss)/d(gaze) from those Jacobians
//   eve_combined_gaze      common.py:136-154 (calculate_combined_gaze_direction)
//   eve_make_heatmaps[_bwd]  common.py:236-255 (batch_make_heatmaps), optional validity mask (label maps, eve.py:503-520)
//   eve_soft_argmax_{fwd,bwd}  common.py:304-333
// All float32, like the reference.  The reference runs these as ~60 tiny torch ops per frame and time step.
#include "common.h"

namespace eve {

// ---- first-order dual number in two directions (d/d pitch, d/d yaw) -------------------------------------------
struct Dual {
    float v, a, b;
};
__device__ __forceinline__ Dual dconst(float v) { return Dual{v, 0.f, 0.f}; }
__device__ __forceinline__ Dual operator+(Dual x, Dual y) { return Dual{x.v + y.v, x.a + y.a, x.b + y.b}; }
__device__ __forceinline__ Dual operator-(Dual x, Dual y) { return Dual{x.v - y.v, x.a - y.a, x.b - y.b}; }
__device__ __forceinline__ Dual operator-(Dual x) { return Dual{-x.v, -x.a, -x.b}; }
__device__ __forceinline__ Dual operator*(Dual x, Dual y) { return Dual{x.v * y.v, x.a * y.v + x.v * y.a, x.b * y.v + x.v * y.b}; }
__device__ __forceinline__ Dual operator*(float c, Dual y) { return Dual{c * y.v, c * y.a, c * y.b}; }
__device__ __forceinline__ Dual operator/(Dual x, Dual y) {
    const float q = x.v / y.v, r = 1.f / y.v;
    return Dual{q, (x.a - q * y.a) * r, (x.b - q * y.b) * r};
}
__device__ __forceinline__ Dual chain(Dual x, float f, float df) { return Dual{f, df * x.a, df * x.b}; }
__device__ __forceinline__ Dual dsin(Dual x) { return chain(x, sinf(x.v), cosf(x.v)); }
__device__ __forceinline__ Dual dcos(Dual x) { return chain(x, cosf(x.v), -sinf(x.v)); }
__device__ __forceinline__ Dual dsqrt(Dual x) { const float s = sqrtf(x.v); return chain(x, s, 0.5f / s); }
__device__ __forceinline__ Dual dasin(Dual x) { return chain(x, asinf(x.v), 1.f / sqrtf(fmaxf(1.f - x.v * x.v, 1e-30f))); }
__device__ __forceinline__ Dual datan2(Dual y, Dual x) {
    const float r2 = fmaxf(x.v * x.v + y.v * y.v, 1e-30f);
    return Dual{atan2f(y.v, x.v), (x.v * y.a - y.v * x.a) / r2, (x.v * y.b - y.v * x.b) / r2};
}

struct Vec3 { Dual x, y, z; };

__device__ __forceinline__ Vec3 pitchyaw_to_vector(Dual p, Dual y) {       // common.py:32-36
    const Dual cp = dcos(p), sp = dsin(p), cy = dcos(y), sy = dsin(y);
    return Vec3{cp * sy, sp, cp * cy};
}
__device__ __forceinline__ void vector_to_pitchyaw(Vec3 v, Dual& p, Dual& y) {   // common.py:43-55
    const Dual n = dsqrt(v.x * v.x + v.y * v.y + v.z * v.z) + dconst(1e-7f);
    p = dasin(v.y / n);
    y = datan2(v.x / n, v.z / n);
}
// r = M v (M row-major 3x3, row stride `ld`) or its transpose
__device__ __forceinline__ Vec3 mat_vec(const float* M, int ld, Vec3 v, bool transpose) {
    Vec3 r;
    if (!transpose) {
        r.x = M[0] * v.x + M[1] * v.y + M[2] * v.z;
        r.y = M[ld] * v.x + M[ld + 1] * v.y + M[ld + 2] * v.z;
        r.z = M[2 * ld] * v.x + M[2 * ld + 1] * v.y + M[2 * ld + 2] * v.z;
    } else {
        r.x = M[0] * v.x + M[ld] * v.y + M[2 * ld] * v.z;
        r.y = M[1] * v.x + M[ld + 1] * v.y + M[2 * ld + 1] * v.z;
        r.z = M[2] * v.x + M[ld + 2] * v.y + M[2 * ld + 2] * v.z;
    }
    return r;
}

// jac[n] = { d g_out[0..1], d mm[0..1], d px[0..1] } / d (pitch, yaw): 6 rows of 2
__global__ __launch_bounds__(128) void gaze_to_pog_kernel(const long long N, const float* __restrict__ g, const float* __restrict__ origin,
                                                          const float* __restrict__ R, const float* __restrict__ inv_cam,
                                                          const float* __restrict__ ppm, const float* __restrict__ head_R,
                                                          const float* __restrict__ kappa, const float screen_w, const float screen_h,
                                                          float* __restrict__ g_out, float* __restrict__ pog_mm, float* __restrict__ pog_px,
                                                          float* __restrict__ jac) {
    const long long n = (long long)blockIdx.x * 128 + threadIdx.x;
    if (n >= N) return;
    Dual p{g[2 * n], 1.f, 0.f}, y{g[2 * n + 1], 0.f, 1.f};
    if (kappa) {                                   // common.py:190-229
        const float* Rh = head_R + 9 * n;
        Vec3 v = pitchyaw_to_vector(p, y);
        v = mat_vec(Rh, 3, v, true);               // -(Rh^T (-v))
        Dual hp, hy;
        vector_to_pitchyaw(v, hp, hy);
        // pitchyaw_to_rotation (common.py:58-80) applied to the kappa vector: Ry(hy) (Rx(hp) k)
        const float kp = kappa[2 * n], ky = kappa[2 * n + 1];
        const Vec3 k{dconst(cosf(kp) * sinf(ky)), dconst(sinf(kp)), dconst(cosf(kp) * cosf(ky))};
        const Dual c0 = dcos(hp), s0 = dsin(hp), c1 = dcos(hy), s1 = dsin(hy);
        const Vec3 m1{k.x, c0 * k.y + s0 * k.z, c0 * k.z - s0 * k.y};
        const Vec3 m2{c1 * m1.x + s1 * m1.z, m1.y, c1 * m1.z - s1 * m1.x};
        v = mat_vec(Rh, 3, m2, false);             // -(Rh (-(m2)))
        vector_to_pitchyaw(v, p, y);
    }
    if (g_out) { g_out[2 * n] = p.v; g_out[2 * n + 1] = y.v; }
    // common.py:157-187
    Vec3 d = pitchyaw_to_vector(p, y);
    d = Vec3{-d.x, -d.y, -d.z};
    d = mat_vec(R + 9 * n, 3, d, true);
    const float* T = inv_cam + 16 * n;
    d = mat_vec(T, 4, d, false);
    const float* o = origin + 3 * n;
    const float ox = T[0] * o[0] + T[1] * o[1] + T[2] * o[2] + T[3];
    const float oy = T[4] * o[0] + T[5] * o[1] + T[6] * o[2] + T[7];
    const float oz = T[8] * o[0] + T[9] * o[1] + T[10] * o[2] + T[11];
    const Dual t = dconst(-oz) / (d.z + dconst(1e-7f));
    const Dual mx = dconst(ox) + t * d.x, my = dconst(oy) + t * d.y;
    pog_mm[2 * n] = mx.v; pog_mm[2 * n + 1] = my.v;
    const float sx = ppm[2 * n], sy = ppm[2 * n + 1];
    const float px = mx.v * sx, py = my.v * sy;
    const bool inx = px >= 0.f && px <= screen_w, iny = py >= 0.f && py <= screen_h;    // clamp passes gradient inside only
    pog_px[2 * n] = fminf(fmaxf(px, 0.f), screen_w);
    pog_px[2 * n + 1] = fminf(fmaxf(py, 0.f), screen_h);
    float* J = jac + 12 * n;
    J[0] = p.a; J[1] = p.b; J[2] = y.a; J[3] = y.b;
    J[4] = mx.a; J[5] = mx.b; J[6] = my.a; J[7] = my.b;
    J[8] = inx ? sx * mx.a : 0.f; J[9] = inx ? sx * mx.b : 0.f;
    J[10] = iny ? sy * my.a : 0.f; J[11] = iny ? sy * my.b : 0.f;
}

__global__ __launch_bounds__(128) void gaze_to_pog_bwd_kernel(const long long N, const float* __restrict__ jac, const float* __restrict__ dg_out,
                                                              const float* __restrict__ dmm, const float* __restrict__ dpx,
                                                              float* __restrict__ dg) {
    const long long n = (long long)blockIdx.x * 128 + threadIdx.x;
    if (n >= N) return;
    const float* J = jac + 12 * n;
    float a = 0.f, b = 0.f;
    if (dg_out) { a += dg_out[2 * n] * J[0] + dg_out[2 * n + 1] * J[2]; b += dg_out[2 * n] * J[1] + dg_out[2 * n + 1] * J[3]; }
    if (dmm) { a += dmm[2 * n] * J[4] + dmm[2 * n + 1] * J[6]; b += dmm[2 * n] * J[5] + dmm[2 * n + 1] * J[7]; }
    if (dpx) { a += dpx[2 * n] * J[8] + dpx[2 * n + 1] * J[10]; b += dpx[2 * n] * J[9] + dpx[2 * n + 1] * J[11]; }
    dg[2 * n] = a; dg[2 * n + 1] = b;
}

// common.py:136-154: gaze (pitch, yaw) from the mean eye origin towards a point on the screen plane
__global__ __launch_bounds__(128) void combined_gaze_kernel(const long long N, const float* __restrict__ origin, const float* __restrict__ pog_mm,
                                                            const float* __restrict__ R, const float* __restrict__ cam, float* __restrict__ g) {
    const long long n = (long long)blockIdx.x * 128 + threadIdx.x;
    if (n >= N) return;
    const float* T = cam + 16 * n;
    const float x = pog_mm[2 * n], y = pog_mm[2 * n + 1];
    const float* o = origin + 3 * n;
    const float dx = T[0] * x + T[1] * y + T[3] - o[0];
    const float dy = T[4] * x + T[5] * y + T[7] - o[1];
    const float dz = T[8] * x + T[9] * y + T[11] - o[2];
    const float* M = R + 9 * n;
    const float vx = -(M[0] * dx + M[1] * dy + M[2] * dz);
    const float vy = -(M[3] * dx + M[4] * dy + M[5] * dz);
    const float vz = -(M[6] * dx + M[7] * dy + M[8] * dz);
    const float nrm = sqrtf(vx * vx + vy * vy + vz * vz) + 1e-7f;
    g[2 * n] = asinf(vy / nrm);
    g[2 * n + 1] = atan2f(vx / nrm, vz / nrm);
}

// common.py:236-255: out[n][y][x] = 1e-8 + exp(-((x - cx)^2 + (y - cy)^2) / (2 sigma^2)), c = centre * (W/sw, H/sh); x valid[n]
__global__ __launch_bounds__(256) void make_heatmaps_kernel(const int H, const int W, const float* __restrict__ centres,
                                                            const uint8_t* __restrict__ valid, const float alpha, const float kx,
                                                            const float ky, float* __restrict__ out) {
    const long long n = blockIdx.y;
    const float cx = kx * centres[2 * n], cy = ky * centres[2 * n + 1];
    const float m = valid ? (valid[n] ? 1.f : 0.f) : 1.f;
    float* o = out + n * H * W;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W; i += gridDim.x * 256) {
        const float dx = (float)(i % W) - cx, dy = (float)(i / W) - cy;
        o[i] = m * (1e-8f + expf(alpha * (dx * dx + dy * dy)));
    }
}

__device__ __forceinline__ float block_sum(float v, float* sh) {           // 256 threads
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ float block_max(float v, float* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// d centre[n] = sum_px dout * exp(..) * (-2 alpha) (x - cx) * kx   (and the same in y); one workgroup per map
__global__ __launch_bounds__(256) void make_heatmaps_bwd_kernel(const int H, const int W, const float* __restrict__ centres,
                                                                const float alpha, const float kx, const float ky,
                                                                const float* __restrict__ dout, float* __restrict__ dcentres) {
    __shared__ float sh[4];
    const long long n = blockIdx.x;
    const float cx = kx * centres[2 * n], cy = ky * centres[2 * n + 1];
    const float* d = dout + n * H * W;
    float ax = 0.f, ay = 0.f;
    for (int i = threadIdx.x; i < H * W; i += 256) {
        const float dx = (float)(i % W) - cx, dy = (float)(i / W) - cy;
        const float e = d[i] * expf(alpha * (dx * dx + dy * dy)) * (-2.f * alpha);
        ax += e * dx; ay += e * dy;
    }
    ax = block_sum(ax, sh);
    ay = block_sum(ay, sh);
    if (threadIdx.x == 0) { dcentres[2 * n] = ax * kx; dcentres[2 * n + 1] = ay * ky; }
}

// common.py:304-333: p = softmax(100 h); (lx, ly) = E_p[(x/(W-1), y/(H-1))]; px = clamp((sw lx, sh ly)); stats = lx, ly, max, sum
__global__ __launch_bounds__(256) void soft_argmax_fwd_kernel(const int H, const int W, const float* __restrict__ heat, const float sw,
                                                              const float sh_, float* __restrict__ pog_px, float* __restrict__ stats) {
    __shared__ float sh[4];
    const long long n = blockIdx.x;
    const float* h = heat + n * H * W;
    float mx = -3.4e38f;
    for (int i = threadIdx.x; i < H * W; i += 256) mx = fmaxf(mx, 100.f * h[i]);
    mx = block_max(mx, sh);
    const float rx = 1.f / (float)(W - 1), ry = 1.f / (float)(H - 1);
    float s = 0.f, ax = 0.f, ay = 0.f;
    for (int i = threadIdx.x; i < H * W; i += 256) {
        const float e = expf(100.f * h[i] - mx);
        s += e; ax += e * ((float)(i % W) * rx); ay += e * ((float)(i / W) * ry);
    }
    s = block_sum(s, sh); ax = block_sum(ax, sh); ay = block_sum(ay, sh);
    if (threadIdx.x == 0) {
        const float lx = ax / s, ly = ay / s;
        pog_px[2 * n] = fminf(fmaxf(sw * lx, 0.f), sw);
        pog_px[2 * n + 1] = fminf(fmaxf(sh_ * ly, 0.f), sh_);
        stats[4 * n] = lx; stats[4 * n + 1] = ly; stats[4 * n + 2] = mx; stats[4 * n + 3] = s;
    }
}

// dh[i] = 100 p_i (gx sw (x_i - lx) + gy sh (y_i - ly)), with gx / gy zeroed where the clamp is active
__global__ __launch_bounds__(256) void soft_argmax_bwd_kernel(const int H, const int W, const float* __restrict__ heat,
                                                              const float* __restrict__ stats, const float* __restrict__ dpog,
                                                              const float sw, const float sh_, float* __restrict__ dheat) {
    const long long n = blockIdx.y;
    const float lx = stats[4 * n], ly = stats[4 * n + 1], mx = stats[4 * n + 2], inv_s = 1.f / stats[4 * n + 3];
    const float vx = sw * lx, vy = sh_ * ly;
    const float gx = (vx >= 0.f && vx <= sw) ? dpog[2 * n] * sw : 0.f;
    const float gy = (vy >= 0.f && vy <= sh_) ? dpog[2 * n + 1] * sh_ : 0.f;
    const float rx = 1.f / (float)(W - 1), ry = 1.f / (float)(H - 1);
    const float* h = heat + n * H * W;
    float* d = dheat + n * H * W;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W; i += gridDim.x * 256) {
        const float p = expf(100.f * h[i] - mx) * inv_s;
        d[i] = 100.f * p * (gx * ((float)(i % W) * rx - lx) + gy * ((float)(i / W) * ry - ly));
    }
}

}  // namespace eve

using namespace eve;

extern "C" int eve_gaze_to_pog(long long N, const float* g, const float* origin, const float* R, const float* inv_cam, const float* ppm,
                               const float* head_R, const float* kappa, float screen_w, float screen_h, float* g_out, float* pog_mm,
                               float* pog_px, float* jac, eve_stream_t stream) {
    if (N <= 0 || !g || !origin || !R || !inv_cam || !ppm || !pog_mm || !pog_px || !jac) return set_error_msg("gaze_to_pog: bad arguments");
    if ((kappa != nullptr) != (head_R != nullptr)) return set_error_msg("gaze_to_pog: kappa and head_R come together");
    if (kappa && !g_out) return set_error_msg("gaze_to_pog: g_out is required with the offset augmentation");
    hipLaunchKernelGGL(gaze_to_pog_kernel, dim3((unsigned)((N + 127) / 128)), dim3(128), 0, (hipStream_t)stream, N, g, origin, R, inv_cam,
                       ppm, head_R, kappa, screen_w, screen_h, g_out, pog_mm, pog_px, jac);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_gaze_to_pog_bwd(long long N, const float* jac, const float* dg_out, const float* dmm, const float* dpx, float* dg,
                                   eve_stream_t stream) {
    if (N <= 0 || !jac || !dg) return set_error_msg("gaze_to_pog_bwd: bad arguments");
    hipLaunchKernelGGL(gaze_to_pog_bwd_kernel, dim3((unsigned)((N + 127) / 128)), dim3(128), 0, (hipStream_t)stream, N, jac, dg_out, dmm,
                       dpx, dg);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_combined_gaze(long long N, const float* origin, const float* pog_mm, const float* R, const float* cam, float* g,
                                 eve_stream_t stream) {
    if (N <= 0 || !origin || !pog_mm || !R || !cam || !g) return set_error_msg("combined_gaze: bad arguments");
    hipLaunchKernelGGL(combined_gaze_kernel, dim3((unsigned)((N + 127) / 128)), dim3(128), 0, (hipStream_t)stream, N, origin, pog_mm, R,
                       cam, g);
    EVE_CHECK_LAUNCH();
    return 0;
}

static int heat_check(long long N, int H, int W, float sigma, float sw, float sh, const char* what) {
    if (N <= 0 || N > 2147483647LL / 4 || H <= 1 || W <= 1 || (long long)H * W > (1 << 24) || !(sigma > 0.f) || !(sw > 0.f) || !(sh > 0.f))
        return set_error_msg(what);
    return 0;
}

extern "C" int eve_make_heatmaps(long long N, int H, int W, const float* centres_px, const uint8_t* validity, float sigma,
                                 float screen_w, float screen_h, float* out, eve_stream_t stream) {
    if (int e = heat_check(N, H, W, sigma, screen_w, screen_h, "make_heatmaps: bad shape")) return e;
    if (!centres_px || !out) return set_error_msg("make_heatmaps: null pointer");
    if (N > 65535) return set_error_msg("make_heatmaps: at most 65535 maps per call");
    const int chunks = (H * W + 1023) / 1024;
    hipLaunchKernelGGL(make_heatmaps_kernel, dim3(chunks, (unsigned)N), dim3(256), 0, (hipStream_t)stream, H, W, centres_px, validity,
                       -0.5f / (sigma * sigma), (float)W / screen_w, (float)H / screen_h, out);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_make_heatmaps_bwd(long long N, int H, int W, const float* centres_px, float sigma, float screen_w, float screen_h,
                                     const float* dout, float* dcentres, eve_stream_t stream) {
    if (int e = heat_check(N, H, W, sigma, screen_w, screen_h, "make_heatmaps_bwd: bad shape")) return e;
    if (!centres_px || !dout || !dcentres) return set_error_msg("make_heatmaps_bwd: null pointer");
    hipLaunchKernelGGL(make_heatmaps_bwd_kernel, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, H, W, centres_px,
                       -0.5f / (sigma * sigma), (float)W / screen_w, (float)H / screen_h, dout, dcentres);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_soft_argmax_fwd(long long N, int H, int W, const float* heat, float screen_w, float screen_h, float* pog_px,
                                   float* stats, eve_stream_t stream) {
    if (int e = heat_check(N, H, W, 1.f, screen_w, screen_h, "soft_argmax_fwd: bad shape")) return e;
    if (!heat || !pog_px || !stats) return set_error_msg("soft_argmax_fwd: null pointer");
    hipLaunchKernelGGL(soft_argmax_fwd_kernel, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, H, W, heat, screen_w, screen_h, pog_px,
                       stats);
    EVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int eve_soft_argmax_bwd(long long N, int H, int W, const float* heat, const float* stats, const float* dpog, float screen_w,
                                   float screen_h, float* dheat, eve_stream_t stream) {
    if (int e = heat_check(N, H, W, 1.f, screen_w, screen_h, "soft_argmax_bwd: bad shape")) return e;
    if (!heat || !stats || !dpog || !dheat) return set_error_msg("soft_argmax_bwd: null pointer");
    if (N > 65535) return set_error_msg("soft_argmax_bwd: at most 65535 maps per call");
    const int chunks = (H * W + 1023) / 1024;
    hipLaunchKernelGGL(soft_argmax_bwd_kernel, dim3(chunks, (unsigned)N), dim3(256), 0, (hipStream_t)stream, H, W, heat, stats, dpog,
                       screen_w, screen_h, dheat);
    EVE_CHECK_LAUNCH();
    return 0;
}

// Validity-masked sequence losses of the EyeNet train step and their gradients in ONE launch.
// Reference: /root/reference/src/losses/angular.py:33-38 (angular error in degrees between the pitch/yaw gaze
// vectors of models/common.py:32-36, cosine similarity clamped to +-(1 - 1e-8)), losses/l1.py, and the
// per-clip validity reduction of losses/base_loss_with_validity.py:64-73 (sum over valid steps / number of
// valid steps when that exceeds one, mean over clips).  As torch ops this is ~260 launches of a few KB per step.
//
// One workgroup per (eye side, clip); thread t handles time step t.  Outputs:
//   terms[5]   = ang_left, l1_left, ang_right, l1_right, coeff_ang * (ang_l + ang_r) + coeff_l1 * (l1_l + l1_r)
//                (accumulated with atomics: zero it first)
//   dg[side][B][T][2], dp[side][B][T] = d(term of that side) / d(prediction)   (unit upstream gradient)
#include "common.h"

namespace eve {

__global__ __launch_bounds__(64) void eye_losses_kernel(const int B, const int T,
                                                        const float* __restrict__ g_pred_l, const float* __restrict__ g_pred_r,
                                                        const float* __restrict__ g_tgt_l, const float* __restrict__ g_tgt_r,
                                                        const uint8_t* __restrict__ g_val_l, const uint8_t* __restrict__ g_val_r,
                                                        const float* __restrict__ p_pred_l, const float* __restrict__ p_pred_r,
                                                        const float* __restrict__ p_tgt_l, const float* __restrict__ p_tgt_r,
                                                        const uint8_t* __restrict__ p_val_l, const uint8_t* __restrict__ p_val_r,
                                                        const float coeff_ang, const float coeff_l1, float* __restrict__ terms,
                                                        float* __restrict__ dg_l, float* __restrict__ dg_r,
                                                        float* __restrict__ dp_l, float* __restrict__ dp_r) {
    const int side = blockIdx.x / B, b = blockIdx.x - side * B;
    const float* gp = side ? g_pred_r : g_pred_l;
    const float* gt = side ? g_tgt_r : g_tgt_l;
    const uint8_t* gv = side ? g_val_r : g_val_l;
    const float* pp = side ? p_pred_r : p_pred_l;
    const float* pt = side ? p_tgt_r : p_tgt_l;
    const uint8_t* pv = side ? p_val_r : p_val_l;
    float* dg = side ? dg_r : dg_l;
    float* dp = side ? dp_r : dp_l;
    const float deg = 57.29577951308232f;
    float ang_sum = 0.f, ang_cnt = 0.f, l1_sum = 0.f, l1_cnt = 0.f;
    // pass 1: per-step values and un-normalised gradients (kept in registers for up to 4 steps per thread)
    float gpitch[4], gyaw[4], gl1[4];
    bool va[4], vp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = threadIdx.x + 64 * i;
        gpitch[i] = gyaw[i] = gl1[i] = 0.f; va[i] = vp[i] = false;
        if (t >= T) continue;
        const size_t o = (size_t)b * T + t;
        va[i] = gv[o] != 0; vp[i] = pv[o] != 0;
        const float pa = gp[2 * o], ya = gp[2 * o + 1], pb = gt[2 * o], yb = gt[2 * o + 1];
        float spa, cpa, sya, cya, spb, cpb, syb, cyb;
        sincosf(pa, &spa, &cpa); sincosf(ya, &sya, &cya); sincosf(pb, &spb, &cpb); sincosf(yb, &syb, &cyb);
        const float ax = cpa * sya, ay = spa, az = cpa * cya;
        const float bx = cpb * syb, by = spb, bz = cpb * cyb;
        const float dot = ax * bx + ay * by + az * bz;
        const float na = fmaxf(sqrtf(ax * ax + ay * ay + az * az), 1e-8f), nb = fmaxf(sqrtf(bx * bx + by * by + bz * bz), 1e-8f);
        const float c = dot / (na * nb);
        const float lim = 1.f - 1e-8f;                       // == 1.f in float, as in the reference
        const float cc = fminf(fmaxf(c, -lim), lim);
        const float a = acosf(cc) * deg;
        if (va[i]) { ang_sum += a; ang_cnt += 1.f; }
        // d a / d (pitch, yaw):  -deg / sqrt(1 - c^2) * dc/dv . dv/d(pitch, yaw);  zero where the clamp is active
        const bool inside = c > -lim && c < lim;
        const float k = inside ? -deg * rsqrtf(fmaxf(1.f - cc * cc, 1e-30f)) : 0.f;
        const float inv = 1.f / (na * nb), ca = c / (na * na);
        const float gx = bx * inv - ca * ax, gy = by * inv - ca * ay, gz = bz * inv - ca * az;
        gpitch[i] = k * (gx * (-spa * sya) + gy * cpa + gz * (-spa * cya));
        gyaw[i] = k * (gx * (cpa * cya) + gz * (-cpa * sya));
        const float d = pp[o] - pt[o];
        if (vp[i]) { l1_sum += fabsf(d); l1_cnt += 1.f; }
        gl1[i] = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    }
    ang_sum = wave_sum(ang_sum); ang_cnt = wave_sum(ang_cnt); l1_sum = wave_sum(l1_sum); l1_cnt = wave_sum(l1_cnt);
    const float inv_b = 1.f / (float)B;
    const float ang_den = ang_cnt > 1.f ? ang_cnt : 1.f, l1_den = l1_cnt > 1.f ? l1_cnt : 1.f;
    if (threadIdx.x == 0) {
        const float ang = ang_sum / ang_den * inv_b, l1 = l1_sum / l1_den * inv_b;
        atomicAdd(terms + 2 * side, ang);
        atomicAdd(terms + 2 * side + 1, l1);
        atomicAdd(terms + 4, coeff_ang * ang + coeff_l1 * l1);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = threadIdx.x + 64 * i;
        if (t >= T) continue;
        const size_t o = (size_t)b * T + t;
        const float sa = va[i] ? inv_b / ang_den : 0.f, sl = vp[i] ? inv_b / l1_den : 0.f;
        dg[2 * o] = sa * gpitch[i]; dg[2 * o + 1] = sa * gyaw[i];
        dp[o] = sl * gl1[i];
    }
}

}  // namespace eve

using namespace eve;

extern "C" int eve_eye_losses(int B, int T, const float* const* g_pred, const float* const* g_tgt, const uint8_t* const* g_val,
                              const float* const* p_pred, const float* const* p_tgt, const uint8_t* const* p_val,
                              float coeff_ang, float coeff_l1, float* terms, float* const* dg, float* const* dp,
                              eve_stream_t stream) {
    if (B <= 0 || T <= 0 || T > 256 || !g_pred || !g_tgt || !g_val || !p_pred || !p_tgt || !p_val || !terms || !dg || !dp)
        return set_error_msg("eye_losses: bad arguments (T <= 256)");
    for (int s = 0; s < 2; ++s)
        if (!g_pred[s] || !g_tgt[s] || !g_val[s] || !p_pred[s] || !p_tgt[s] || !p_val[s] || !dg[s] || !dp[s])
            return set_error_msg("eye_losses: null pointer");
    hipLaunchKernelGGL(eye_losses_kernel, dim3(2 * B), dim3(64), 0, (hipStream_t)stream, B, T, g_pred[0], g_pred[1], g_tgt[0],
                       g_tgt[1], g_val[0], g_val[1], p_pred[0], p_pred[1], p_tgt[0], p_tgt[1], p_val[0], p_val[1], coeff_ang,
                       coeff_l1, terms, dg[0], dg[1], dp[0], dp[1]);
    EVE_CHECK_LAUNCH();
    return 0;
}

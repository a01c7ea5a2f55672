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

// ---------------------------------------------------------------------------------------------------------------------------
// Round 5: ALL the validity-masked terms of EVE.calculate_losses_and_metrics over [B][T][D <= 3] predictions (the 28 gaze / PoG /
// pupil losses and metrics of eve.py:286-439) in ONE launch: one workgroup per term.  As torch expressions they were ~20-35
// launches of a few KB each, ~600 of the ~830 ATen launches of a configs[2] step.  kind: 0 = MSE (mean over D of the squared
// difference, losses/mse.py), 1 = Euclidean distance (losses/euclidean.py:27-33), 2 = L1 (mean over D, losses/l1.py), 3 = angular
// error in degrees between pitch / yaw gaze vectors (losses/angular.py:33-38).  Reduction: base_loss_with_validity.py:64-73.
// Summation order is fixed (wave w takes clips w, w + 4, ..; one thread adds the clip means in clip order).
// dpred (optional, kinds 0 / 2 / 3): d term / d pred for a unit upstream gradient.
// ---------------------------------------------------------------------------------------------------------------------------
struct VecTermsArg { int n; int pad; eve_vec_term t[EVE_VEC_TERMS_MAX]; };

__device__ __forceinline__ float vt_step(const int kind, const int D, const float* __restrict__ p, const float* __restrict__ q,
                                         float (&g)[3]) {
    g[0] = g[1] = g[2] = 0.f;
    if (kind == 3) {
        const float deg = 57.29577951308232f;
        float spa, cpa, sya, cya, spb, cpb, syb, cyb;
        sincosf(p[0], &spa, &cpa); sincosf(p[1], &sya, &cya); sincosf(q[0], &spb, &cpb); sincosf(q[1], &syb, &cyb);
        const float ax = cpa * sya, ay = spa, az = cpa * cya;
        const float bx = cpb * syb, by = spb, bz = cpb * cyb;
        const float dot = ax * bx + ay * by + az * bz;
        const float na = fmaxf(sqrtf(ax * ax + ay * ay + az * az), 1e-8f), nb = fmaxf(sqrtf(bx * bx + by * by + bz * bz), 1e-8f);
        const float c = dot / (na * nb);
        const float lim = 1.f - 1e-8f;                       // == 1.f in float, as in the reference
        const float cc = fminf(fmaxf(c, -lim), lim);
        const bool inside = c > -lim && c < lim;
        const float k = inside ? -deg * rsqrtf(fmaxf(1.f - cc * cc, 1e-30f)) : 0.f;
        const float inv = 1.f / (na * nb), ca = c / (na * na);
        const float gx = bx * inv - ca * ax, gy = by * inv - ca * ay, gz = bz * inv - ca * az;
        g[0] = k * (gx * (-spa * sya) + gy * cpa + gz * (-spa * cya));
        g[1] = k * (gx * (cpa * cya) + gz * (-cpa * sya));
        return acosf(cc) * deg;
    }
    float acc = 0.f;
    const float inv_d = 1.f / (float)D;
    for (int d = 0; d < D; ++d) {
        const float e = p[d] - q[d];
        if (kind == 2) { acc += fabsf(e); g[d] = (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) * inv_d; }
        else { acc += e * e; g[d] = 2.f * e * inv_d; }
    }
    if (kind == 1) return sqrtf(acc);
    return acc * inv_d;
}

__global__ __launch_bounds__(256) void vector_terms_kernel(const VecTermsArg a, const int B, const int T, float* __restrict__ out) {
    extern __shared__ float vt_lds[];                            // [B] clip means, [B] denominators
    float* const cmean = vt_lds;
    float* const cden = vt_lds + B;
    const eve_vec_term tm = a.t[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = tm.D, kind = tm.kind;
    for (int b = wave; b < B; b += 4) {
        float sum = 0.f, cnt = 0.f;
        for (int t = lane; t < T; t += 64) {
            const size_t o = (size_t)b * T + t;
            if (tm.valid[o]) {
                float g[3];
                sum += vt_step(kind, D, tm.pred + o * D, tm.tgt + o * D, g);
                cnt += 1.f;
            }
        }
        sum = wave_sum(sum); cnt = wave_sum(cnt);
        if (lane == 0) {
            const float den = cnt > 1.f ? cnt : 1.f;
            cmean[b] = sum / den;
            cden[b] = den;
        }
    }
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += cmean[b];
        out[blockIdx.x] = s / (float)B;
    }
    if (tm.dpred) {
        const float inv_b = 1.f / (float)B;
        for (int o = tid; o < B * T; o += 256) {
            const int b = o / T;
            float g[3] = {0.f, 0.f, 0.f};
            if (tm.valid[o]) {
                vt_step(kind, D, tm.pred + (size_t)o * D, tm.tgt + (size_t)o * D, g);
                const float sc = inv_b / cden[b];
                g[0] *= sc; g[1] *= sc; g[2] *= sc;
            }
            for (int d = 0; d < D; ++d) tm.dpred[(size_t)o * D + d] = g[d];
        }
    }
}

}  // namespace eve

using namespace eve;

extern "C" int eve_vector_terms(const eve_vec_term* terms, int n, int B, int T, float* out, eve_stream_t stream) {
    if (!terms || n <= 0 || n > EVE_VEC_TERMS_MAX || B <= 0 || B > 4096 || T <= 0 || !out) return set_error_msg("vector_terms: bad arguments");
    VecTermsArg a;
    a.n = n; a.pad = 0;
    for (int i = 0; i < n; ++i) {
        const eve_vec_term& t = terms[i];
        if (!t.pred || !t.tgt || !t.valid || t.D < 1 || t.D > 3 || t.kind < 0 || t.kind > 3 || (t.kind == 3 && t.D != 2) ||
            (t.kind == 1 && t.dpred))
            return set_error_msg("vector_terms: bad term (D in 1..3; angular needs D = 2; no gradient for the Euclidean distance)");
        a.t[i] = t;
    }
    hipLaunchKernelGGL(vector_terms_kernel, dim3(n), dim3(256), (size_t)2 * B * sizeof(float), (hipStream_t)stream, a, B, T, out);
    EVE_CHECK_LAUNCH();
    return 0;
}

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

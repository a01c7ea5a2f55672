"""A torch-CPU stand-in with the SAME tensor-level interface as eve_amd.kernels.HipKernels.

TEST INFRASTRUCTURE ONLY.  It lets the `-m "not gpu"` suite exercise the HOST logic of eve_amd
(autograd wiring, weight packing, padding, module plumbing, state_dict contract, data-parallel
bucketing) in a container without a GPU.  It is never importable from the product package and is not
a fallback: eve_amd raises when libeve_hip.so is missing.  Each method restates, with ATen ops, the
contract documented in include/eve_hip.h; the GPU suite checks the HIP kernels against the same
contracts independently.
"""
import torch
import torch.nn.functional as F

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_SELU, ACT_TANH, ACT_SIGMOID = range(6)


def act_fwd(z, act):
    return {ACT_NONE: lambda v: v, ACT_RELU: torch.relu, ACT_LEAKY: lambda v: F.leaky_relu(v, 0.01),
            ACT_SELU: F.selu, ACT_TANH: torch.tanh, ACT_SIGMOID: torch.sigmoid}[act](z)


def act_grad_from_out(y, act):
    a, s = 1.6732632423543772, 1.0507009873554805
    if act == ACT_RELU:
        return (y > 0).to(y.dtype)
    if act == ACT_LEAKY:
        return torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.01))
    if act == ACT_SELU:
        return torch.where(y > 0, torch.full_like(y, s), y + s * a)
    if act == ACT_TANH:
        return 1 - y * y
    if act == ACT_SIGMOID:
        return y * (1 - y)
    return torch.ones_like(y)


def nchw(x):
    return x.permute(0, 3, 1, 2).float()


def nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(dtype)


class FakeKernels(object):
    name = 'fake-cpu'

    def _pro(self, x, ss, pro_act):
        if ss is None:
            return x.float()
        return act_fwd(x.float() * ss[:, None, None, :, 0] + ss[:, None, None, :, 1], pro_act)

    def conv2d_fwd(self, x, w_ohwi, bias, stride, pad, epi_act=ACT_NONE, ss=None, pro_act=ACT_NONE, algo=None,
                   accumulate_into=None):
        xin = self._pro(x, ss, pro_act).to(x.dtype)
        y = F.conv2d(nchw(xin), w_ohwi.permute(0, 3, 1, 2).float(), bias, stride, pad)
        if accumulate_into is not None:
            accumulate_into.copy_((accumulate_into.float() + nhwc(act_fwd(y, epi_act), torch.float32)).to(x.dtype))
            return accumulate_into
        return nhwc(act_fwd(y, epi_act), x.dtype)

    def conv2d_dgrad(self, dy, w_ihwo, in_hw, stride, pad, algo=None, accumulate_into=None):
        w = w_ihwo.permute(3, 0, 1, 2).float()          # [Cout, Cin, KH, KW]
        N = dy.shape[0]
        dx = torch.nn.grad.conv2d_input((N, w.shape[1], in_hw[0], in_hw[1]), w, nchw(dy), stride, pad)
        if accumulate_into is not None:
            accumulate_into.copy_((accumulate_into.float() + nhwc(dx, torch.float32)).to(dy.dtype))
            return accumulate_into
        return nhwc(dx, dy.dtype)

    def conv2d_wgrad(self, x, dy, KH, KW, stride, pad, dw_ohwi, ss=None, pro_act=ACT_NONE, algo=None, db=None):
        if db is not None:
            db += dy.float().sum(dim=(0, 1, 2))
        xin = self._pro(x, ss, pro_act).to(x.dtype)
        Cout, Cin = dy.shape[3], x.shape[3]
        dw = torch.nn.grad.conv2d_weight(nchw(xin), (Cout, Cin, KH, KW), nchw(dy), stride, pad)
        dw_ohwi += dw.permute(0, 2, 3, 1)
        return dw_ohwi

    def stem_pack_input(self, src_nchw, out=None, dtype=torch.bfloat16):
        N, C, H, W = src_nchw.shape
        dst = torch.zeros((N, H + 6, W + 8, 4), dtype=dtype) if out is None else out
        dst.zero_()
        dst[:, 3:H + 3, 4:W + 4, :C] = src_nchw.permute(0, 2, 3, 1).to(dst.dtype)
        return dst

    def stem7x7s2_fwd(self, x_padded, w_ohwi8):
        x = x_padded[:, 3:-3, 4:-4, :].float().permute(0, 3, 1, 2)
        w = w_ohwi8[..., :4].float().permute(0, 3, 1, 2)
        return nhwc(F.conv2d(x, w, None, 2, 3), x_padded.dtype)

    def stem_fwd_fused(self, x_padded, w_ohwi8, eps=1e-5):
        conv = self.stem7x7s2_fwd(x_padded, w_ohwi8)
        mr = self.instnorm_stats(conv, eps)
        y, idx = self.in_relu_maxpool_fwd(conv, mr)
        return y, idx, mr

    def stem_wgrad(self, x_padded, dconv, dw):
        x = x_padded[:, 3:-3, 4:-4, :].float().permute(0, 3, 1, 2)
        g = torch.nn.grad.conv2d_weight(x, (64, 4, 7, 7), nchw(dconv), stride=2, padding=3)     # [64, 4, 7, 7]
        dw[:, :, :7, :] += g.permute(0, 2, 3, 1)

    def stem_bwd_dx(self, x_padded, w_ohwi8, mr, dy_pool, y_pool, idx, dy_pool2=None):
        if dy_pool2 is not None:
            dy_pool = (dy_pool.float() + dy_pool2.float()).to(dy_pool.dtype)
        return self.in_relu_maxpool_bwd(dy_pool, y_pool, idx, self.stem7x7s2_fwd(x_padded, w_ohwi8), mr)

    def linear_fwd(self, x, w_in_out, bias, act):
        z = x @ w_in_out
        if bias is not None:
            z = z + bias
        return act_fwd(z, act)

    def linear_dgrad(self, dy, y, act, w_out_in):
        g = dy * act_grad_from_out(y, act) if act != ACT_NONE else dy
        return g @ w_out_in

    def linear_wgrad(self, dy, y, act, x, dw_out_in, db):
        g = dy * act_grad_from_out(y, act) if act != ACT_NONE else dy
        dw_out_in += g.t() @ x
        if db is not None:
            db += g.sum(0)

    def bias_grad(self, dy, db):
        db += dy.float().reshape(-1, dy.shape[-1]).sum(0)
        return db

    def sum_rows(self, t):
        return t.sum(dim=0)

    def instnorm_stats(self, x, eps=1e-5):
        xf = x.float()
        mean = xf.mean(dim=(1, 2))
        var = xf.var(dim=(1, 2), unbiased=False)
        return torch.stack([mean, torch.rsqrt(var + eps)], dim=-1)

    def instnorm_act_fwd(self, x, mr, gamma, beta, res, act):
        z = (x.float() - mr[:, None, None, :, 0]) * mr[:, None, None, :, 1]
        if gamma is not None:
            z = z * gamma + beta
        if res is not None:
            z = z + res.float()
        return act_fwd(z, act).to(x.dtype)

    def instnorm_act_bwd(self, dy, y, x, mr, gamma, act, want_dres, beta=None, dx_add=None):
        g = dy.float()
        if act != ACT_NONE:
            if y is None:
                assert not want_dres and (gamma is None) == (beta is None)
                z = (x.float() - mr[:, None, None, :, 0]) * mr[:, None, None, :, 1]
                if gamma is not None:
                    z = z * gamma + beta
                y = act_fwd(z, act)
            g = g * act_grad_from_out(y.float(), act)
        xhat = (x.float() - mr[:, None, None, :, 0]) * mr[:, None, None, :, 1]
        s1 = g.sum(dim=(1, 2))
        s2 = (g * xhat).sum(dim=(1, 2))
        hw = x.shape[1] * x.shape[2]
        k = mr[:, None, None, :, 1] * (gamma if gamma is not None else 1.0)
        dx = k * (g - s1[:, None, None] / hw - xhat * s2[:, None, None] / hw)
        dx = dx.to(x.dtype)
        if dx_add is not None:
            dx = (dx.float() + dx_add.float()).to(x.dtype)
        return dx, (g.to(x.dtype) if want_dres else None), torch.stack([s1, s2], dim=-1)

    def instnorm_act2_fwd(self, xs, mrs, gamma_a, beta_a, gamma_b, beta_b, act):
        outs_a, outs_b, off = [], [], 0
        for x, mr in zip(xs, mrs):
            sl = slice(off, off + x.shape[-1])
            outs_a.append(self.instnorm_act_fwd(x, mr, gamma_a[sl], beta_a[sl], None, act))
            if gamma_b is not None:
                outs_b.append(self.instnorm_act_fwd(x, mr, gamma_b[sl], beta_b[sl], None, act))
            off += x.shape[-1]
        return torch.cat(outs_a, dim=-1), (torch.cat(outs_b, dim=-1) if gamma_b is not None else None)

    def instnorm_act2_bwd(self, dy_a, dy_b, xs, mrs, gamma_a, beta_a, gamma_b, beta_b, act):
        dxs, sas, sbs, off = [], [], [], 0
        for x, mr in zip(xs, mrs):
            sl = slice(off, off + x.shape[-1])
            dxa, _, sa = self.instnorm_act_bwd(dy_a[..., sl].float(), None, x.float(), mr, gamma_a[sl], act, False, beta=beta_a[sl])
            sas.append(sa)
            if dy_b is not None:
                dxb, _, sb = self.instnorm_act_bwd(dy_b[..., sl].float(), None, x.float(), mr, gamma_b[sl], act, False, beta=beta_b[sl])
                sbs.append(sb)
                dxa = dxa + dxb
            dxs.append(dxa.to(x.dtype))
            off += x.shape[-1]
        return dxs, torch.cat(sas, dim=1), (torch.cat(sbs, dim=1) if dy_b is not None else None)

    def instnorm_fwd_fused(self, x, gamma, beta, res, act, eps=1e-5, want_mask=False):
        if x.shape[1] * x.shape[2] * x.shape[3] > 65536:
            return None
        mr = self.instnorm_stats(x, eps)
        y = self.instnorm_act_fwd(x, mr, gamma, beta, res, act)
        if not want_mask:
            return y, mr
        vec = 16 // x.element_size()                     # bit e of byte v = (element e of 16-byte vector v of y > 0)
        bits = (y.reshape(-1, vec) > 0).to(torch.int32) << torch.arange(vec, dtype=torch.int32)
        return y, mr, bits.sum(dim=1).to(torch.uint8)

    def instnorm_bwd_fused(self, dy, y, x, mr, gamma, act, want_dres, dy2=None, mask=None, beta=None, dx_add=None):
        if x.shape[1] * x.shape[2] * x.shape[3] > 65536:
            return None
        if dy2 is not None:
            dy = (dy.float() + dy2.float()).to(dy.dtype)
        if mask is not None:                             # ReLU: any y with the mask's signs gives the same derivative
            assert act == ACT_RELU and y is None
            vec = 16 // x.element_size()
            y = ((mask.to(torch.int32).unsqueeze(1) >> torch.arange(vec, dtype=torch.int32)) & 1).reshape(x.shape).to(x.dtype)
        return self.instnorm_act_bwd(dy, y, x, mr, gamma, act, want_dres, beta=beta if y is None else None, dx_add=dx_add)

    def act_bwd(self, dy, y, act):
        return (dy.float() * act_grad_from_out(y.float(), act)).to(dy.dtype)

    def add(self, a, b):
        return (a.float() + b.float()).to(a.dtype)

    def maxpool3x3s2_fwd(self, x):
        y, idx = F.max_pool2d(nchw(x), 3, 2, 1, return_indices=True)
        return nhwc(y, x.dtype), nhwc(idx, torch.int64)

    def maxpool3x3s2_bwd(self, dy, idx, in_hw):
        N, OH, OW, C = dy.shape
        dx = torch.zeros((N, C, in_hw[0] * in_hw[1]))
        dx.scatter_add_(2, idx.permute(0, 3, 1, 2).reshape(N, C, -1), nchw(dy).reshape(N, C, -1))
        return nhwc(dx.view(N, C, in_hw[0], in_hw[1]), dy.dtype)

    def in_relu_maxpool_fwd(self, x, mr):
        return self.maxpool3x3s2_fwd(self.instnorm_act_fwd(x, mr, None, None, None, ACT_RELU))

    def in_relu_maxpool_bwd(self, dy_pool, y_pool, idx, x, mr):
        d_act = self.maxpool3x3s2_bwd(dy_pool, idx, (x.shape[1], x.shape[2]))
        y_act = self.instnorm_act_fwd(x, mr, None, None, None, ACT_RELU)
        return self.instnorm_act_bwd(d_act, y_act, x, mr, None, ACT_RELU, False)[0]

    def avgpool_fwd(self, x):
        return x.float().mean(dim=(1, 2)).to(x.dtype)

    def avgpool_bwd(self, dy, hw):
        N, C = dy.shape
        return (dy.float()[:, None, None, :] / (hw[0] * hw[1])).expand(N, hw[0], hw[1], C).contiguous().to(dy.dtype)

    def adaptive_maxpool_fwd(self, x, out_hw):
        y, idx = F.adaptive_max_pool2d(nchw(x), out_hw, return_indices=True)
        return nhwc(y, x.dtype), nhwc(idx, torch.int64)

    def adaptive_maxpool_bwd(self, dy, idx, in_hw, add=None):
        dx = self.maxpool3x3s2_bwd(dy, idx, in_hw)
        return dx if add is None else (dx.float() + add.float()).to(dx.dtype)

    def bilinear_fwd(self, x, out_hw):
        return nhwc(F.interpolate(nchw(x), size=out_hw, mode='bilinear', align_corners=False), x.dtype)

    def bilinear_bwd(self, dy, in_hw):
        N, OH, OW, C = dy.shape
        with torch.enable_grad():
            probe = torch.zeros((N, C, in_hw[0], in_hw[1]), requires_grad=True)
            out = F.interpolate(probe, size=(OH, OW), mode='bilinear', align_corners=False)
            (dx,) = torch.autograd.grad(out, probe, nchw(dy))
        return nhwc(dx, dy.dtype)

    def nchw_to_nhwc(self, src, dtype, cpad=None, out=None):
        N, C, H, W = src.shape
        cpad = cpad or C
        dst = torch.zeros((N, H, W, cpad), dtype=dtype) if out is None else out
        dst.zero_()
        dst[..., :C] = src.permute(0, 2, 3, 1).to(dtype)
        return dst

    def nhwc_to_nchw(self, src, C):
        return src[..., :C].permute(0, 3, 1, 2).float().contiguous()

    def cast(self, src, dtype):
        return src.to(dtype)

    def pack_weights_batch(self, ws, dtype, want_ihwo):
        return [self.pack_weights(w, dtype, want_ihwo=wi) for w, wi in zip(ws, want_ihwo)]

    def pack_weights(self, w_ohwi_f32, dtype, want_ihwo=True):
        ohwi = w_ohwi_f32.to(dtype).contiguous()
        ihwo = w_ohwi_f32.permute(3, 1, 2, 0).to(dtype).contiguous() if want_ihwo else None
        return ohwi, ihwo

    def rnn_scan_fwd(self, gi, whh_t, bhh, h0):
        S, T, H = gi.shape
        h = torch.zeros((S, H)) if h0 is None else h0
        hs = []
        for t in range(T):
            h = torch.tanh(gi[:, t] + h @ whh_t + bhh)
            hs.append(h)
        return torch.stack(hs, 1)

    def rnn_scan_bwd(self, dhs, whh, hs, want_dh0):
        S, T, H = dhs.shape
        dh = torch.zeros((S, H))
        dpre = torch.zeros((S, T, H))
        for t in range(T - 1, -1, -1):
            d = dh + dhs[:, t]
            dpre[:, t] = d * (1 - hs[:, t] ** 2)
            dh = dpre[:, t] @ whh
        return dpre, (dh if want_dh0 else None)

    def lstm_scan_fwd(self, gi, whh_t, bhh, h0, c0):
        S, T, H4 = gi.shape
        H = H4 // 4
        h = torch.zeros((S, H)) if h0 is None else h0
        c = torch.zeros((S, H)) if c0 is None else c0
        hs, cs, gates = [], [], []
        for t in range(T):
            pre = gi[:, t] + h @ whh_t + bhh
            i, f, g, o = torch.sigmoid(pre[:, :H]), torch.sigmoid(pre[:, H:2 * H]), torch.tanh(pre[:, 2 * H:3 * H]), \
                torch.sigmoid(pre[:, 3 * H:])
            c = f * c + i * g
            h = o * torch.tanh(c)
            hs.append(h); cs.append(c); gates.append(torch.cat([i, f, g, o], 1))
        return torch.stack(hs, 1), torch.stack(cs, 1), torch.stack(gates, 1)

    def lstm_scan_bwd(self, dhs, dcs, whh, c0, hs, cs, gates, want_d0):
        S, T, H = dhs.shape
        dh, dc = torch.zeros((S, H)), torch.zeros((S, H))
        dpre = torch.zeros((S, T, 4 * H))
        for t in range(T - 1, -1, -1):
            d = dh + dhs[:, t]
            i, f, g, o = gates[:, t, :H], gates[:, t, H:2 * H], gates[:, t, 2 * H:3 * H], gates[:, t, 3 * H:]
            tc = torch.tanh(cs[:, t])
            cp = cs[:, t - 1] if t > 0 else (c0 if c0 is not None else torch.zeros((S, H)))
            dc = dc + (dcs[:, t] if dcs is not None else 0) + d * o * (1 - tc * tc)
            dpre[:, t] = torch.cat([dc * g * i * (1 - i), dc * cp * f * (1 - f), dc * i * (1 - g * g), d * tc * o * (1 - o)], 1)
            dc = dc * f
            dh = dpre[:, t] @ whh
        return dpre, (dh if want_d0 else None), (dc if want_d0 else None)

    def gru_scan_fwd(self, gi, whh_t, bhh, h0):
        S, T, H3 = gi.shape
        H = H3 // 3
        h = torch.zeros((S, H)) if h0 is None else h0
        hs, gates, hn_pre = [], [], []
        for t in range(T):
            gh = h @ whh_t + bhh
            r = torch.sigmoid(gi[:, t, :H] + gh[:, :H])
            z = torch.sigmoid(gi[:, t, H:2 * H] + gh[:, H:2 * H])
            n = torch.tanh(gi[:, t, 2 * H:] + r * gh[:, 2 * H:])
            h = (1 - z) * n + z * h
            hs.append(h); gates.append(torch.cat([r, z, n], 1)); hn_pre.append(gh[:, 2 * H:])
        return torch.stack(hs, 1), torch.stack(gates, 1), torch.stack(hn_pre, 1)

    def gru_scan_bwd(self, dhs, whh, h0, hs, gates, hn_pre, want_dh0):
        S, T, H = dhs.shape
        dh = torch.zeros((S, H))
        dgi = torch.zeros((S, T, 3 * H)); dgh = torch.zeros((S, T, 3 * H))
        for t in range(T - 1, -1, -1):
            d = dh + dhs[:, t]
            r, z, n = gates[:, t, :H], gates[:, t, H:2 * H], gates[:, t, 2 * H:]
            hp = hs[:, t - 1] if t > 0 else (h0 if h0 is not None else torch.zeros((S, H)))
            dn = d * (1 - z) * (1 - n * n)
            dz = d * (hp - n) * z * (1 - z)
            dr = dn * hn_pre[:, t] * r * (1 - r)
            dgi[:, t] = torch.cat([dr, dz, dn], 1)
            dgh[:, t] = torch.cat([dr, dz, dn * r], 1)
            dh = d * z + dgh[:, t] @ whh
        return dgi, dgh, (dh if want_dh0 else None)

    def cgru_gates1(self, g1, h):
        C = h.shape[-1]
        ru = torch.sigmoid(g1.float())
        return ru.to(h.dtype), (ru[..., :C] * h.float()).to(h.dtype)

    def cgru_gates2(self, g2, ru, h):
        C = h.shape[-1]
        o = torch.tanh(g2.float())
        u = ru.float()[..., C:]
        return o.to(h.dtype), ((1 - u) * o + u * h.float()).to(h.dtype)

    def cgru_scan_fwd(self, xs, h0, w1_ohwi, b1, w2_ohwi, b2):
        B, T = xs.shape[:2]
        h = torch.zeros_like(xs[:, 0]) if h0 is None else h0
        hs, rus, rhs, ogs = [], [], [], []
        for t in range(T):
            g1 = self.conv2d_fwd(torch.cat([xs[:, t], h], -1), w1_ohwi, b1, 1, 1)
            ru, rh = self.cgru_gates1(g1, h)
            g2 = self.conv2d_fwd(torch.cat([rh, xs[:, t]], -1), w2_ohwi, b2, 1, 1)
            o, h = self.cgru_gates2(g2, ru, h)
            hs.append(h); rus.append(ru); rhs.append(rh); ogs.append(o)
        return torch.stack(hs, 1), torch.stack(hs, 0), torch.stack(rus, 0), torch.stack(rhs, 0), torch.stack(ogs, 0)

    def cgru_scan_bwd(self, dhs_tm, ru, og, hs_tm, h0, w1_ihwo, w2_ihwo, want_dh0=False):
        """Contract of eve_cgru_scan_bwd: frames last to first, the carry kept in float, the gradients of the two
        pre-activations rounded to the storage dtype before they enter the data-gradient convolutions."""
        T, B, H, W, C = dhs_tm.shape
        dt = dhs_tm.dtype
        carry = torch.zeros((B, H, W, C))
        dg1_all, dg2_all, dxs = [None] * T, [None] * T, [None] * T
        for t in range(T - 1, -1, -1):
            hp = (hs_tm[t - 1] if t > 0 else (h0 if h0 is not None else torch.zeros_like(hs_tm[0]))).float()
            r, u, o = ru[t].float()[..., :C], ru[t].float()[..., C:], og[t].float()
            dhn = dhs_tm[t].float() + carry
            dg2 = (dhn * (1 - u) * (1 - o * o)).to(dt)
            g1u = dhn * (hp - o) * u * (1 - u)
            dcat2 = torch.nn.grad.conv2d_input((B, 2 * C, H, W), w2_ihwo.permute(3, 0, 1, 2).float(), nchw(dg2), 1, 1).permute(0, 2, 3, 1)
            drh, dx2 = dcat2[..., :C], dcat2[..., C:]
            dg1 = torch.cat([drh * hp * r * (1 - r), g1u], dim=-1).to(dt)
            dcat1 = torch.nn.grad.conv2d_input((B, 2 * C, H, W), w1_ihwo.permute(3, 0, 1, 2).float(), nchw(dg1), 1, 1).permute(0, 2, 3, 1)
            carry = dhn * u + drh * r + dcat1[..., C:]
            dg1_all[t], dg2_all[t], dxs[t] = dg1, dg2, (dcat1[..., :C] + dx2).to(dt)
        return torch.stack(dg1_all, 0), torch.stack(dg2_all, 0), torch.stack(dxs, 0), (carry.to(dt) if want_dh0 else None)

    def crnn_scan_fwd(self, xs, h0, w_ohwi, bias):
        B, T = xs.shape[:2]
        h = torch.zeros_like(xs[:, 0]) if h0 is None else h0
        hs = []
        for t in range(T):
            h = self.conv2d_fwd(torch.cat([xs[:, t], h], -1), w_ohwi, bias, 1, 1, ACT_TANH)
            hs.append(h)
        return torch.stack(hs, 1), torch.stack(hs, 0)

    def crnn_scan_bwd(self, dhs_tm, hs_tm, w_ihwo, want_dh0=False):
        T, B, H, W, C = dhs_tm.shape
        carry = torch.zeros((B, H, W, C))
        dpre, dxs = [None] * T, [None] * T
        for t in range(T - 1, -1, -1):
            a = (dhs_tm[t] + carry) * (1 - hs_tm[t] * hs_tm[t])
            dcat = torch.nn.grad.conv2d_input((B, 2 * C, H, W), w_ihwo.permute(3, 0, 1, 2).float(), nchw(a), 1, 1).permute(0, 2, 3, 1)
            dpre[t], dxs[t], carry = a, dcat[..., :C].contiguous(), dcat[..., C:]
        return torch.stack(dpre, 0), torch.stack(dxs, 0), (carry.contiguous() if want_dh0 else None)

    def clstm_scan_fwd(self, xs, h0, c0, w_ohwi, bias):
        B, T = xs.shape[:2]
        h = torch.zeros_like(xs[:, 0]) if h0 is None else h0
        c = torch.zeros_like(xs[:, 0]) if c0 is None else c0
        hs, cs = [], []
        for t in range(T):
            gates = self.conv2d_fwd(torch.cat([xs[:, t], h], -1), w_ohwi, bias, 1, 1)
            h, c = self.clstm_gates_fwd(gates, c)
            hs.append(h); cs.append(c)
        return torch.stack(hs, 1), torch.stack(cs, 1)

    def cgru_gates2_bwd(self, dhnew, ru, h, o):
        C = h.shape[-1]
        d, u, of = dhnew.float(), ru.float()[..., C:], o.float()
        dg2 = d * (1 - u) * (1 - of * of)
        dru = torch.cat([torch.zeros_like(d), d * (h.float() - of)], dim=-1)
        return dg2.to(h.dtype), dru.to(h.dtype), (d * u).to(h.dtype)

    def cgru_gates1_bwd(self, drh, dru, ru, h):
        C = h.shape[-1]
        r, u = ru.float()[..., :C], ru.float()[..., C:]
        d = drh.float()
        dg1 = torch.cat([d * h.float() * r * (1 - r), dru.float()[..., C:] * u * (1 - u)], dim=-1)
        return dg1.to(h.dtype), (d * r).to(h.dtype)

    def clstm_gates_fwd(self, gates, c_prev):
        i, f, o, g = gates.float().chunk(4, dim=-1)
        c = torch.sigmoid(f) * c_prev.float() + torch.sigmoid(i) * torch.tanh(g)
        return (torch.sigmoid(o) * torch.tanh(c)).to(c_prev.dtype), c.to(c_prev.dtype)

    def sumsq(self, g, out, workspace=None):
        out += (g.double() ** 2).sum().float()
        return out

    ADAM_GUARD_WORDS = 12

    @staticmethod
    def new_adam_guard(device, loss_scale=1.0, step=0):
        g = torch.zeros(12, dtype=torch.int32, device=device)
        g[0] = int(step)
        g.view(torch.float32)[4] = float(loss_scale)
        return g

    def adam_step(self, p, g, m, v, sumsq, max_norm, gscale, lr, beta1, beta2, eps, weight_decay, step,
                  guard=None, check_finite=False, lr_dev=None, poison=None):
        """include/eve_hip.h eve_adam_step, guard semantics included (skip on a non-finite norm or a poisoned gradient exchange,
        the step counter advancing only on taken steps, loss-scale back-off / growth)."""
        if lr_dev is not None:
            lr = float(lr_dev)
        clip = gscale
        if guard is not None:
            gf = guard.view(torch.float32)
            if poison is not None and not (float(poison[0]) == 0.0):
                guard[1] += 1
                guard[9] += 1
                gf[5] = 0.0
                return
            ls = float(gf[4]) if float(gf[4]) > 0 else 1.0
            gs = gscale / ls
            clip = gs
            ss = float(sumsq) if sumsq is not None else None
            if ss is not None and check_finite and not (ss < 3.0e38):
                guard[1] += 1
                guard[2] += 1
                guard[3] = 0
                gf[5] = 0.0
                if int(guard[2]) >= 2:
                    gf[4] = max(ls * 0.5, 1.0)
                    guard[2] = 0
                return
            if ss is not None and max_norm > 0:
                total = ss ** 0.5 * gs
                clip = gs * min(1.0, max_norm / (total + 1e-6))
            guard[0] += 1
            step = int(guard[0])
            guard[2] = 0
            guard[3] += 1
            if check_finite and int(guard[3]) >= 2000:
                gf[4] = min(ls * 2.0, 65536.0)
                guard[3] = 0
            gf[5] = 1.0
            gf[6] = clip
        elif sumsq is not None:
            total = float(sumsq.sqrt()) * gscale
            clip = gscale * min(1.0, max_norm / (total + 1e-6))
        gi = clip * g + weight_decay * p
        m.mul_(beta1).add_(gi, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(gi, gi, value=1 - beta2)
        bc1 = 1 - beta1 ** step
        bc2 = (1 - beta2 ** step) ** 0.5
        p.sub_((lr / bc1) * m / (v.sqrt() / bc2 + eps))


# ---- gaze geometry / heat-maps / soft-argmax: CPU stand-ins built on the oracle's restatement (tests only) ----------
def _install_geometry_fakes():
    import types
    from oracle import eve as oe
    from oracle.config import OracleConfig

    def cfg_for(hw, screen):
        c = OracleConfig()
        c.gaze_heatmap_size = [hw[1], hw[0]]
        c.actual_screen_size = [screen[0], screen[1]]
        return c

    def gaze_to_pog(self, g, origin, R, inv_cam, ppm, screen, head_R=None, kappa=None):
        N = g.shape[0]

        def f(gi):
            go = gi if kappa is None else oe.offset_augmentation(gi, head_R, kappa)
            mm, px = oe.to_screen_coordinates(origin, go, R, inv_cam, ppm, screen)
            return go, mm, px
        jac = torch.zeros(N, 6, 2)
        with torch.enable_grad():
            gi = g.detach().clone().requires_grad_(True)
            outs = f(gi)
            for oi, o in enumerate(outs):
                for c in range(2):
                    if not o.requires_grad:
                        continue
                    gr, = torch.autograd.grad(o[:, c].sum(), gi, retain_graph=True, allow_unused=True)
                    jac[:, 2 * oi + c] = 0 if gr is None else gr      # frames are independent: the sum separates them
        return outs[0].detach(), outs[1].detach(), outs[2].detach(), jac

    def gaze_to_pog_bwd(self, jac, dg_out, dmm, dpx):
        dg = torch.zeros(jac.shape[0], 2)
        for i, d in enumerate((dg_out, dmm, dpx)):
            if d is not None:
                dg += torch.einsum('ni,nij->nj', d, jac[:, 2 * i:2 * i + 2])
        return dg

    def combined_gaze(self, origin, pog_mm, R, cam):
        return oe.combined_gaze_direction(origin, pog_mm, R, cam)

    def make_heatmaps(self, centres_px, sigma, hw, screen, validity=None):
        m = oe.make_heatmaps(centres_px, sigma, cfg_for(hw, screen))
        return m if validity is None else m * validity.float().view(-1, 1, 1, 1)

    def make_heatmaps_bwd(self, centres_px, sigma, screen, dout):
        with torch.enable_grad():
            c = centres_px.detach().clone().requires_grad_(True)
            m = oe.make_heatmaps(c, sigma, cfg_for(dout.shape[2:], screen))
            return torch.autograd.grad((m * dout).sum(), c)[0]

    def soft_argmax_fwd(self, heat, screen):
        return oe.soft_argmax(heat, cfg_for(heat.shape[2:], screen)), heat.new_zeros(heat.shape[0], 4)

    def soft_argmax_bwd(self, heat, stats, dpog, screen):
        with torch.enable_grad():
            h = heat.detach().clone().requires_grad_(True)
            px = oe.soft_argmax(h, cfg_for(heat.shape[2:], screen))
            return torch.autograd.grad((px * dpog).sum(), h)[0]

    for fn in (gaze_to_pog, gaze_to_pog_bwd, combined_gaze, make_heatmaps, make_heatmaps_bwd, soft_argmax_fwd, soft_argmax_bwd):
        setattr(FakeKernels, fn.__name__, fn)


_install_geometry_fakes()


def _install_frame_fakes():
    def frames_u8_to_nchw(self, frames, scale, shift=None):
        out = frames.permute(0, 3, 1, 2).float() * torch.tensor(scale, dtype=torch.float32)
        return out if shift is None else out + torch.tensor(shift, dtype=torch.float32)

    def frames_u8_to_stem(self, frames, scale, shift, out=None):
        return self.stem_pack_input(frames_u8_to_nchw(self, frames, scale, shift), out=out)

    FakeKernels.frames_u8_to_nchw = frames_u8_to_nchw
    FakeKernels.frames_u8_to_stem = frames_u8_to_stem


_install_frame_fakes()


def _install_heatmap_fakes():
    """Contracts of eve_heatmap_head_* / eve_heatmap_loss_* (include/eve_hip.h) with ATen ops."""
    def heatmap_head_fwd(self, logits):
        return torch.sigmoid(logits[..., 0].float()).unsqueeze(1).contiguous()

    def heatmap_head_bwd(self, dy, y, dtype, cpad):
        N, _, H, W = y.shape
        dl = torch.zeros((N, H, W, cpad), dtype=dtype)
        dl[..., 0] = (dy.float() * y * (1 - y))[:, 0].to(dtype)
        return dl

    def _per_map(kind, pred, gt):
        if kind == 0:
            return F.binary_cross_entropy(pred, gt, reduction='none').flatten(2).mean(dim=2)
        return ((pred - gt) ** 2).flatten(2).mean(dim=2)

    def heatmap_loss_fwd(self, kind, pred, gt, validity):
        v = validity.float()
        n = v.sum(dim=1, keepdim=True)
        den = torch.where(n > 1, n, torch.ones_like(n))
        w = v / (den * pred.shape[0])
        return (_per_map(kind, pred, gt) * w).sum(), w.reshape(-1)

    def heatmap_loss_bwd(self, kind, pred, gt, w, upstream):
        with torch.enable_grad():
            p = pred.detach().clone().requires_grad_(True)
            loss = (_per_map(kind, p, gt) * w.view(pred.shape[0], pred.shape[1])).sum()
            return torch.autograd.grad(loss, p)[0] * upstream

    for fn in (heatmap_head_fwd, heatmap_head_bwd, heatmap_loss_fwd, heatmap_loss_bwd):
        setattr(FakeKernels, fn.__name__, fn)


_install_heatmap_fakes()

"""-m gpu: every contraction of the fused LocoTransformer forward / backward, teacher-forced — the oracle's op (operands
rounded to the compute type, fp32 accumulate: oracle/ppo_oracle.py `linear` / `conv2d`) applied to the HIP path's OWN saved
input must reproduce the HIP path's saved output to accumulation-order noise (2e-5 of the tensor's max-abs). Tensors the
kernels keep in bf16 are compared as bf16: equal to the rounded expectation except for isolated 1-ulp ties. This is the
per-contraction statement of 'same rounding points, RNE, fp32 accumulate' that the end-to-end bf16 numbers cannot make
(two bf16 evaluations of a 20-contraction chain differ by the chain's own sensitivity). Errors go to gpurun_out/parity.json."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import util
from oracle import ppo_oracle as orc
from test_gpu_parity import MODES, _build

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _close32(name, got, want, tag, tol=TOL):
    e = util.rel_err(got, want)
    util.record("contraction/%s/%s" % (tag, name), e)
    assert e <= tol, (name, e)


def _close_t(name, got_t, want32, mode, tag, max_ulp=1.0):
    """got_t: values the kernel stored in the compute type (as float64/32 tensor); want32: fp32 expectation before rounding."""
    if mode == "f32":
        return _close32(name, got_t, want32, tag)
    want_r = orc.rbf16(want32.float())
    # allowed distance: one bf16 ulp of the value (<= 2^-7 relative) on top of the fp32 accumulation noise floor of the tensor
    ulp = want32.abs().float() * 2.0 ** -7 + TOL * want32.abs().max().float()
    diff = (got_t.float() - want32.float()).abs()
    frac = ((got_t.float() != want_r).float().mean()).item()
    worst = (diff / ulp).max().item()
    util.record("contraction/%s/%s/frac_not_equal_to_rounded" % (tag, name), frac)
    util.record("contraction/%s/%s/worst_in_ulp" % (tag, name), worst)
    assert worst <= max_ulp + 1e-3 and frac <= 5e-3, (name, worst, frac)


@pytest.mark.parametrize("mode", MODES)
def test_every_contraction_teacher_forced(mode, device, layer_taps):
    case = util.CASES["loco_s93"]
    n, S, A, R = case["B"], case["S"], case["A"], case["B"] * 17
    pf, vf = _build(case, mode, device)
    tag = "loco_s93/" + mode
    hip = pf.hip
    obs = torch.tensor(util.make_batch(case)["obs"], dtype=torch.float32)
    st, im, _ = hip.stage(obs.to(device))
    hip.forward(st, im, n, train=True)
    w = torch.tensor(np.random.RandomState(5).randn(n, A), dtype=torch.float32)
    dout = torch.zeros(n, 16, dtype=torch.float32, device=device)
    dout[:, :A] = w.to(device)
    grads = torch.full((hip.total_params,), float("nan"), dtype=torch.float32, device=device)
    hip.backward(st, im, n, dout, grads)
    torch.cuda.synchronize()
    ws = hip.workspace(n).cpu()
    tdt = torch.float32 if mode == "f32" else torch.bfloat16

    def tap(name, rows, cols, t=False):
        off = hip.ws_offset(n, name)
        raw = ws[off:off + rows * cols]
        if t and tdt == torch.bfloat16:
            return raw.view(torch.bfloat16)[:rows * cols].view(rows, cols).float()
        return raw.view(rows, cols).clone()
    sd = {k: v.detach().cpu() for k, v in pf.state_dict().items()}
    G = lambda k: hip.grad_view(grads, k).cpu()
    lin = lambda x, wk, bk: orc.linear(x, sd[wk], sd[bk], mode)
    r = (lambda x: x) if mode == "f32" else orc.rbf16
    state, img = orc.split_obs(obs, S)
    img_t = im.cpu().float().view(n, 4, 64, 64)                      # the ingested depth stack (compute type)
    assert torch.equal(img_t, r(img))
    with torch.no_grad():
        # ---------------------------------------------------------------- encoder forward
        enc = "encoder.depth_visual_base.layers."
        nhwc = lambda x: x.permute(0, 2, 3, 1).reshape(n, -1, x.shape[1])
        nchw = lambda t2, hw, c: t2.view(n, hw, hw, c).permute(0, 3, 1, 2)
        c1, c2, c3 = tap("c1", n * 225, 32), tap("c2", n * 36, 64), tap("c3", n * 16, 64)
        _close32("conv1", c1.view(n, 225, 32), nhwc(torch.relu(orc.conv2d(img_t, sd[enc + "0.weight"], sd[enc + "0.bias"], 4, mode))), tag)
        _close32("conv2", c2.view(n, 36, 64), nhwc(torch.relu(orc.conv2d(nchw(c1, 15, 32), sd[enc + "2.weight"], sd[enc + "2.bias"], 2, mode))), tag)
        _close32("conv3", c3.view(n, 16, 64), nhwc(torch.relu(orc.conv2d(nchw(c2, 6, 64), sd[enc + "4.weight"], sd[enc + "4.bias"], 1, mode))), tag)
        x0 = tap("x0", R, 64).view(n, 17, 64)
        up = orc.conv2d(nchw(c3, 4, 64), sd["encoder.depth_up_conv.weight"], sd["encoder.depth_up_conv.bias"], 1, mode)
        _close32("up_conv", x0[:, 1:], nhwc(up), tag)
        eh0, eh1 = tap("eh0", n, 256), tap("eh1", n, 256)
        _close32("enc_fc1", eh0, torch.relu(lin(state, "encoder.base.seq_fcs.0.weight", "encoder.base.seq_fcs.0.bias")), tag)
        _close32("enc_fc2", eh1, torch.relu(lin(eh0, "encoder.base.seq_fcs.2.weight", "encoder.base.seq_fcs.2.bias")), tag)
        _close32("state_projector", x0[:, 0], torch.relu(lin(eh1, "encoder.state_projector.projection.0.weight",
                                                             "encoder.state_projector.projection.0.bias")), tag)
        # ---------------------------------------------------------------- transformer layers forward
        xs = [tap("x%d" % l, R, 64) for l in range(3)]
        L = []
        for l in range(2):
            p = "visual_append_layers.%d." % l
            # (fp32 mode keeps no separate operand copy of the layer input: the fp32 token tensor is the operand)
            xin = tap("xin%d" % l, R, 64, True) if mode == "bf16" else xs[l]
            qkv, P = tap("qkv%d" % l, R, 192), tap("P%d" % l, n * 17, 17)
            ctx, xh1, rs1 = tap("ctx%d" % l, R, 64, True), tap("xh1_%d" % l, R, 64), tap("rs1_%d" % l, R, 1)
            x1t, f, xh2, rs2 = tap("mid%d" % l, R, 64, True), tap("ff%d" % l, R, 256, True), tap("xh2_%d" % l, R, 64), tap("rs2_%d" % l, R, 1)
            _close_t("L%d.layer_input_copy" % l, xin, xs[l], mode, tag)
            _close32("L%d.in_proj" % l, qkv, lin(xin, p + "self_attn.in_proj_weight", p + "self_attn.in_proj_bias"), tag)
            q, k, v = (t_.view(n, 17, 64) for t_ in qkv.split(64, dim=-1))
            Pw = torch.softmax((r(q) @ r(k).transpose(1, 2)) * 0.125, dim=-1)
            _close32("L%d.softmax(QK^T/8)" % l, P.view(n, 17, 17), Pw, tag)
            _close_t("L%d.ctx=PV" % l, ctx.view(n, 17, 64), r(P.view(n, 17, 17)) @ r(v), mode, tag)
            z = xs[l] + lin(ctx, p + "self_attn.out_proj.weight", p + "self_attn.out_proj.bias")
            mu, var = z.mean(-1, keepdim=True), z.var(-1, unbiased=False, keepdim=True)
            _close32("L%d.out_proj+res+norm1.xhat" % l, xh1, (z - mu) / torch.sqrt(var + 1e-5), tag)
            _close32("L%d.norm1.rstd" % l, rs1, 1.0 / torch.sqrt(var + 1e-5), tag)
            x1 = xh1 * sd[p + "norm1.weight"] + sd[p + "norm1.bias"]
            _close_t("L%d.x1_copy" % l, x1t, x1, mode, tag)
            _close_t("L%d.linear1+relu" % l, f, torch.relu(lin(x1, p + "linear1.weight", p + "linear1.bias")), mode, tag)
            z2 = x1 + lin(f, p + "linear2.weight", p + "linear2.bias")
            mu2, var2 = z2.mean(-1, keepdim=True), z2.var(-1, unbiased=False, keepdim=True)
            _close32("L%d.linear2+res+norm2.xhat" % l, xh2, (z2 - mu2) / torch.sqrt(var2 + 1e-5), tag)
            _close32("L%d.layer_output" % l, xs[l + 1], xh2 * sd[p + "norm2.weight"] + sd[p + "norm2.bias"], tag)
            L.append(dict(p=p, xin=xin, qkv=qkv, P=P.view(n, 17, 17), ctx=ctx, xh1=xh1, rs1=rs1, x1=x1, x1t=x1t, f=f, xh2=xh2, rs2=rs2))
        # ---------------------------------------------------------------- heads forward
        hp = "visual_seq_append_fcs."
        xl = xs[2].view(n, 17, 64)
        pooled = torch.cat([xl[:, 0], xl[:, 1:].mean(1)], -1)
        hh0, hh1, out = tap("hh0", n, 256), tap("hh1", n, 256), tap("out", n, 16)
        if hip.ws_offset(n, "pooled") >= 0:
            pass  # (the fused head keeps the pooled row in LDS)
        _close32("head_fc0", hh0, torch.relu(lin(pooled, hp + "0.weight", hp + "0.bias")), tag)
        _close32("head_fc1", hh1, torch.relu(lin(hh0, hp + "2.weight", hp + "2.bias")), tag)
        _close32("head_out", out[:, :A], lin(hh1, hp + "4.weight", hp + "4.bias"), tag)
        # ---------------------------------------------------------------- backward: data-grad chain of the block
        dmm = lambda dy, wk: r(dy) @ r(sd[wk])                               # dX = dY W, both operands rounded
        dhh1, dhh0 = tap("dhh1", n, 256), tap("dhh0", n, 256)
        _close32("d.head_fc1_pre", dhh1, (hh1 > 0) * dmm(dout.cpu()[:, :A], hp + "4.weight"), tag)
        _close32("d.head_fc0_pre", dhh0, (hh0 > 0) * dmm(dhh1, hp + "2.weight"), tag)
        dpool = dmm(dhh0, hp + "0.weight")
        dxs = [tap("dx%d" % l, R, 64) for l in range(3)]
        dy = torch.zeros(n, 17, 64)
        dy[:, 0] = dpool[:, :64]
        dy[:, 1:] = (dpool[:, 64:] / 16.0).unsqueeze(1)
        dy = dy.view(R, 64)                                                   # (dx2 itself stays in LDS in the fused HEAD launch)

        def ln_bwd(dyv, xh, rs, gk):
            dxh = dyv * sd[gk]
            c1_ = dxh.mean(-1, keepdim=True)
            c2_ = (dxh * xh).mean(-1, keepdim=True)
            return rs * (dxh - c1_ - xh * c2_)
        for l in (1, 0):
            d = L[l]
            p = d["p"]
            if l == 0:
                dy = dxs[1]
            dz2, df, dz1, dqkv = tap("dz2_%d" % l, R, 64, True), tap("df%d" % l, R, 256, True), tap("dz1_%d" % l, R, 64, True), tap("dqkv%d" % l, R, 192, True)
            dz2_w = ln_bwd(dy, d["xh2"], d["rs2"], p + "norm2.weight")
            _close_t("d.L%d.norm2_bwd" % l, dz2, dz2_w, mode, tag)
            _close_t("d.L%d.df=(dz2 W2)*mask" % l, df, (d["f"] > 0) * dmm(dz2_w, p + "linear2.weight"), mode, tag)
            dx1 = dz2_w + dmm((d["f"] > 0) * dmm(dz2_w, p + "linear2.weight"), p + "linear1.weight")
            dz1_w = ln_bwd(dx1, d["xh1"], d["rs1"], p + "norm1.weight")
            _close_t("d.L%d.dx1+norm1_bwd" % l, dz1, dz1_w, mode, tag)
            dctx = dmm(dz1_w, p + "self_attn.out_proj.weight").view(n, 17, 64)
            q, k, v = (t_.view(n, 17, 64) for t_ in d["qkv"].split(64, dim=-1))
            P = d["P"]
            dP = r(dctx) @ r(v).transpose(1, 2)          # every product on operands rounded to the compute type
            dS = P * (dP - (P * dP).sum(-1, keepdim=True))
            dv_ = r(P).transpose(1, 2) @ r(dctx)
            dq_, dk_ = (r(dS) @ r(k)) * 0.125, (r(dS).transpose(1, 2) @ r(q)) * 0.125
            dqkv_w = torch.cat([dq_, dk_, dv_], -1).view(R, 192)
            # (two rounding stages in a row — dS is rounded before it enters dQ / dK — so a 1e-7 difference in dS can move
            # an output by a second ulp)
            _close_t("d.L%d.attention_bwd" % l, dqkv, dqkv_w, mode, tag, max_ulp=2.0)
            dxin = dz1_w + dmm(dqkv_w, p + "self_attn.in_proj_weight")
            _close32("d.L%d.layer_input_grad" % l, dxs[l], dxin, tag, tol=1e-4 if mode == "bf16" else TOL)
            # ---- the four weight gradients of the layer: dW = dY^T X on the operands the kernels saved
            for nm, dyv, xv in (("linear1", df, d["x1t"]), ("linear2", dz2, d["f"]), ("self_attn.out_proj", dz1, d["ctx"]),
                                ("self_attn.in_proj", dqkv, d["xin"])):
                wk = p + nm + ("_weight" if "in_proj" in nm else ".weight")
                bk = p + nm + ("_bias" if "in_proj" in nm else ".bias")
                _close32("w.L%d.%s.weight" % (l, nm), G(wk), (dyv.double().t() @ xv.double()).float(), tag)
                _close32("w.L%d.%s.bias" % (l, nm), G(bk), dyv.double().sum(0).float(), tag)
        # ---- encoder side of the TAIL launch: dc3 = ((dx0 tokens 1..16) W_up) o [c3 > 0], and conv3 / up-conv weight gradients
        dc3 = tap("dc3", n * 16, 64)
        dtok = dxs[0].view(n, 17, 64)[:, 1:].reshape(n * 16, 64)
        _close32("d.up_conv->dc3", dc3, (c3 > 0) * (r(dtok) @ r(sd["encoder.depth_up_conv.weight"].view(64, 64))), tag)
        _close32("w.up_conv.weight", G("encoder.depth_up_conv.weight").view(64, 64), (r(dtok).double().t() @ r(c3).double()).float(), tag)
        w3 = sd[enc + "4.weight"]
        _close32("w.conv3.weight", G(enc + "4.weight"),
                 torch.nn.grad.conv2d_weight(r(nchw(c2, 6, 64)).double(), w3.shape, r(nchw(dc3, 4, 64)).double(), stride=1).float(), tag)
        _close32("w.conv3.bias", G(enc + "4.bias"), dc3.double().sum(0).float(), tag)
        # ---- head / encoder-MLP weight gradients
        _close32("w.head_out.weight", G(hp + "4.weight"), (r(dout.cpu()[:, :A]).double().t() @ r(hh1).double()).float(), tag)
        _close32("w.head_fc1.weight", G(hp + "2.weight"), (r(dhh1).double().t() @ r(hh0).double()).float(), tag)
        _close32("w.head_fc0.weight", G(hp + "0.weight"), (r(dhh0).double().t() @ r(pooled).double()).float(), tag)
        _close32("w.head_fc0.bias", G(hp + "0.bias"), dhh0.double().sum(0).float(), tag)

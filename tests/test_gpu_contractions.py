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


_MODE = ["f32"]  # compute mode of the running test (set by _build_for): _close32's default tolerance depends on it


def _close32(name, got, want, tag, tol=None):
    # f16: an operand the kernel derives in fp32 before rounding it (the pooled row, a LayerNorm output) can sit on a rounding tie
    # that another fp32 summation order decides the other way — one operand element off by an f16 ulp (5e-4 relative) moves a
    # 128..256-term output by a few 1e-5 of the tensor's max-abs (measured 4.9e-5, head_fc0 at B = 1024); bf16 ties are 8 x rarer
    if tol is None:
        tol = 1e-4 if _MODE[0] == "f16" else TOL
    e = util.rel_err(got, want)
    util.record("contraction/%s/%s" % (tag, name), e)
    assert e <= tol, (name, e)


def _close_t(name, got_t, want32, mode, tag, max_ulp=1.0, floor=TOL):
    """got_t: values the kernel stored in the compute type (as float64/32 tensor); want32: fp32 expectation before rounding.
    floor: absolute noise floor as a fraction of the tensor's max-abs — fp32 accumulation noise for a single contraction; a
    result that sits BEHIND further rounding stages (the attention backward: dctx and dS are rounded on the way) inherits their
    1-ulp ties as a perturbation at the scale of the tensor, not of the element."""
    if mode == "f32":
        return _close32(name, got_t, want32, tag)
    want_r = orc.ROUND[mode](want32.float())
    # allowed distance: one ulp of the value in the operand type (bf16: <= 2^-7 relative; f16: <= 2^-10 relative, 2^-24 absolute
    # in its subnormal range) on top of the noise floor of the tensor
    ulp = want32.abs().float() * 2.0 ** (-7 if mode == "bf16" else -10) + (0.0 if mode == "bf16" else 2.0 ** -24) \
        + floor * want32.abs().max().float()
    diff = (got_t.float() - want32.float()).abs()
    frac = ((got_t.float() != want_r).float().mean()).item()
    worst = (diff / ulp).max().item()
    util.record("contraction/%s/%s/frac_not_equal_to_rounded" % (tag, name), frac)
    util.record("contraction/%s/%s/worst_in_ulp" % (tag, name), worst)
    # (f16: 8 x as many elements sit within fp32 noise of a rounding tie as in bf16)
    bad = (diff / ulp) > max_ulp + 1e-3
    detail = ""
    if bad.any():  # where: elements the kernel zeroed and the expectation did not (a mask), or the other way round, and how large
        g, w_ = got_t.float()[bad], want32.float()[bad]
        detail = "; %d off: got==0 %d, want==0 %d, |want| min %.3g max %.3g, |got| max %.3g" % (
            int(bad.sum()), int((g == 0).sum()), int((w_ == 0).sum()), w_.abs().min().item(), w_.abs().max().item(), g.abs().max().item())
    assert worst <= max_ulp + 1e-3 and frac <= (2e-2 if mode == "f16" else 5e-3), (name, worst, frac, detail)


def _conv_backward_checks(tap, G, sd, enc, n, c1, c2, dc3, img_t, mode, tag):
    """conv3' / conv2' data-grads and the conv2 / conv1 weight- and bias-gradients of the fused conv-stack backward
    (bwd_conv_kernel: dc2 / dc1 never reach HBM in production; V4L_LAYER_TAPS writes them as the kernel holds them, in T),
    each from the kernel's OWN operands: conv2d_input / conv2d_weight on (dc3, w3) -> dc2, (dc2, c1) -> dW2, (dc2, w2) -> dc1,
    (dc1, image) -> dW1 (torchrl/networks/base.py:317-324 reversed)."""
    r = (lambda x: x) if mode == "f32" else orc.ROUND[mode]
    nhwc = lambda x: x.permute(0, 2, 3, 1).reshape(n, -1, x.shape[1])
    nchw = lambda t2, hw, c: t2.reshape(n, hw, hw, c).permute(0, 3, 1, 2)
    w3, w2, w1 = sd[enc + "4.weight"], sd[enc + "2.weight"], sd[enc + "0.weight"]
    c1n, c2n = nchw(c1, 15, 32), nchw(c2, 6, 64)
    dc2_t, dc1_t = tap("dc2", n * 36, 64, True), tap("dc1", n * 225, 32, True)
    want2 = (c2n > 0) * torch.nn.grad.conv2d_input(c2n.shape, r(w3), r(nchw(dc3, 4, 64)), stride=1)
    _close_t("d.conv3->dc2", dc2_t.view(n, 36, 64), nhwc(want2), mode, tag)
    dc2n = nchw(dc2_t, 6, 64)  # (already in the compute type)
    want1 = (c1n > 0) * torch.nn.grad.conv2d_input(c1n.shape, r(w2), dc2n, stride=2)
    _close_t("d.conv2->dc1", dc1_t.view(n, 225, 32), nhwc(want1), mode, tag)
    dc1n = nchw(dc1_t, 15, 32)
    _close32("w.conv2.weight", G(enc + "2.weight"),
             torch.nn.grad.conv2d_weight(r(c1n).double(), w2.shape, dc2n.double(), stride=2).float(), tag)
    _close32("w.conv1.weight", G(enc + "0.weight"),
             torch.nn.grad.conv2d_weight(img_t.double(), w1.shape, dc1n.double(), stride=4).float(), tag)
    # bias gradients are summed from the fp32 values BEFORE they are rounded to T (oracle: db = dy.sum of the unrounded dy)
    _close32("w.conv2.bias", G(enc + "2.bias"), want2.double().sum((0, 2, 3)).float(), tag)
    _close32("w.conv1.bias", G(enc + "0.bias"), want1.double().sum((0, 2, 3)).float(), tag)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", ["loco_s93", "loco_b1024"])
def test_every_contraction_teacher_forced(name, mode, device, layer_taps):
    _MODE[0] = mode
    case = util.CASES[name]
    n, S, A, R = case["B"], case["S"], case["A"], case["B"] * 17
    pf, vf = _build(case, mode, device)
    tag = name + "/" + mode
    hip = pf.hip
    obs = torch.tensor(util.make_batch(case)["obs"], dtype=torch.float32)
    st, im, _ = hip.stage(obs.to(device))
    hip.forward(st, im, n, train=True)
    w = torch.tensor(np.random.RandomState(5).randn(n, A), dtype=torch.float32)
    dout = torch.zeros(n, 16, dtype=torch.float32, device=device)
    dout[:, :A] = w.to(device)
    grads = torch.full((hip.total_params,), float("nan"), dtype=torch.float32, device=device)
    hip.backward(st, im, n, dout, grads)
    # f16: the backward ran on d(out) x grad_scale (a power of two) and unscaled the parameter gradients at the end; the checks below
    # follow the SCALED chain (the taps hold scaled tensors), so d(out) and the gradients are put on that scale here (exact)
    gs = hip.last_grad_scale  # (chosen by HipNet.backward from max |d(out)|: the probe rows are not a mean loss's)
    assert gs == orc.probe_scale(mode, float(dout.abs().max()))
    dout = dout * gs
    torch.cuda.synchronize()
    ws = hip.workspace(n).cpu()
    tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[mode]

    def tap(name, rows, cols, t=False):
        off = hip.ws_offset(n, name)
        raw = ws[off:off + rows * cols]
        if t and tdt != torch.float32:
            return raw.view(tdt)[:rows * cols].view(rows, cols).float()
        return raw.view(rows, cols).clone()
    sd = {k: v.detach().cpu() for k, v in pf.state_dict().items()}
    G = lambda k: hip.grad_view(grads, k).cpu() * gs
    lin = lambda x, wk, bk: orc.linear(x, sd[wk], sd[bk], mode)
    r = (lambda x: x) if mode == "f32" else orc.ROUND[mode]
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
            xin = tap("xin%d" % l, R, 64, True) if mode != "f32" else xs[l]
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
            # (GEMM operands: the kernel's OWN rounded tensors — the dz2 / df taps — so that a 1-ulp tie in an operand is not
            # charged to the next contraction; the residual adds the fp32 value)
            dff = dmm(dz2, p + "linear2.weight")
            fmask = d["f"] > 0
            if mode == "f16":
                # ReLU's mask is decided on the fp32 pre-activation (reference and oracle: relu's own backward); the tap holds it
                # rounded to half, where a positive value below 2^-25 is zero (bf16 keeps fp32's exponent range: cannot happen).
                # Where the tap is zero the kernel may legitimately have passed the gradient: take its decision there.
                fmask = fmask | ((d["f"] == 0) & (df.float() != 0))
                util.record("contraction/%s/L%d.relu_mask_below_half_range" % (tag, l), int(((d["f"] == 0) & (df.float() != 0)).sum()))
                assert int(((d["f"] == 0) & (df.float() != 0)).sum()) <= 4
            _close_t("d.L%d.df=(dz2 W2)*mask" % l, df, fmask * dff, mode, tag)
            dx1 = dz2_w + dmm(df, p + "linear1.weight")
            dz1_w = ln_bwd(dx1, d["xh1"], d["rs1"], p + "norm1.weight")
            _close_t("d.L%d.dx1+norm1_bwd" % l, dz1, dz1_w, mode, tag)
            dctx = dmm(dz1, p + "self_attn.out_proj.weight").view(n, 17, 64)
            q, k, v = (t_.view(n, 17, 64) for t_ in d["qkv"].split(64, dim=-1))
            P = d["P"]
            dP = r(dctx) @ r(v).transpose(1, 2)          # every product on operands rounded to the compute type
            dS = P * (dP - (P * dP).sum(-1, keepdim=True))
            dv_ = r(P).transpose(1, 2) @ r(dctx)
            dq_, dk_ = (r(dS) @ r(k)) * 0.125, (r(dS).transpose(1, 2) @ r(q)) * 0.125
            dqkv_w = torch.cat([dq_, dk_, dv_], -1).view(R, 192)
            # (two rounding stages in a row — dS is rounded before it enters dQ / dK — so a 1e-7 difference in dS can move
            # an output by a second ulp)
            _close_t("d.L%d.attention_bwd" % l, dqkv, dqkv_w, mode, tag, max_ulp=2.0, floor=1e-3)
            dxin = dz1_w + dmm(dqkv, p + "self_attn.in_proj_weight")
            _close32("d.L%d.layer_input_grad" % l, dxs[l], dxin, tag, tol=1e-4 if mode != "f32" else TOL)
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
        _conv_backward_checks(tap, G, sd, enc, n, c1, c2, dc3, img_t, mode, tag)
        # ---- encoder MLP / projector: data-grads of the TAIL launch and their weight gradients
        dx0 = dxs[0].view(n, 17, 64)[:, 0]
        ypr = (x0[:, 0] > 0) * dx0                                         # grad w.r.t. the projector's pre-activation
        dhc, deh0 = tap("dhc", n, 256), tap("deh0", n, 256)
        _close32("d.state_projector->dhc", dhc, (eh1 > 0) * dmm(ypr, "encoder.state_projector.projection.0.weight"), tag)
        _close32("d.enc_fc2->deh0", deh0, (eh0 > 0) * dmm(dhc, "encoder.base.seq_fcs.2.weight"), tag)
        _close32("w.state_projector.weight", G("encoder.state_projector.projection.0.weight"), (r(ypr).double().t() @ r(eh1).double()).float(), tag)
        _close32("w.enc_fc2.weight", G("encoder.base.seq_fcs.2.weight"), (r(dhc).double().t() @ r(eh0).double()).float(), tag)
        _close32("w.enc_fc1.weight", G("encoder.base.seq_fcs.0.weight"), (r(deh0).double().t() @ r(state).double()).float(), tag)
        # ---- head / encoder-MLP weight gradients
        _close32("w.head_out.weight", G(hp + "4.weight"), (r(dout.cpu()[:, :A]).double().t() @ r(hh1).double()).float(), tag)
        _close32("w.head_fc1.weight", G(hp + "2.weight"), (r(dhh1).double().t() @ r(hh0).double()).float(), tag)
        _close32("w.head_fc0.weight", G(hp + "0.weight"), (r(dhh0).double().t() @ r(pooled).double()).float(), tag)
        _close32("w.head_fc0.bias", G(hp + "0.bias"), dhh0.double().sum(0).float(), tag)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", ["cnn_s93", "mlp_s93", "cnn_vis", "loco_vis"])
def test_contractions_of_the_other_nets(name, mode, device, layer_taps, monkeypatch):
    """The same per-contraction statement for the nets that run on the general (layer-by-layer) GEMM kernels plus the fused
    conv-stack backward: NatureCNN fuse net (nets.py:247-262, base.py:371-385), state MLP (nets.py:52-55) and the two
    vision-only nets (nets.py:133-191, 784-906) — conv stack forward, projector / MLP / head chains forward, their data-grads,
    and every weight gradient from the kernels' own operands. (The vision-only Transformer's layers run on the general
    attention / LayerNorm kernels; its conv stack, up-conv and head are checked here.)"""
    _MODE[0] = mode
    case = util.CASES[name]
    kind = case["kind"]
    if kind == "loco_vis":
        # the taps below are the layer-by-layer path's 16-row token tensors; the wave-per-sample path of this net (17-row stride,
        # dummy row 0) is checked against this one by test_vision_only_transformer_on_wave_per_sample_kernels
        monkeypatch.setenv("V4L_NO_WPS_LAYERS", "1")
    n, S, A = case["B"], case["S"], case["A"]
    pf, vf = _build(case, mode, device)
    tag = name + "/" + mode
    hip = pf.hip
    obs = torch.tensor(util.make_batch(case)["obs"], dtype=torch.float32)
    st, im, _ = hip.stage(obs.to(device))
    hip.forward(st, im, n, train=True)
    w = torch.tensor(np.random.RandomState(5).randn(n, A), dtype=torch.float32)
    dout = torch.zeros(n, 16, dtype=torch.float32, device=device)
    dout[:, :A] = w.to(device)
    grads = torch.full((hip.total_params,), float("nan"), dtype=torch.float32, device=device)
    hip.backward(st, im, n, dout, grads)
    # f16: the backward ran on d(out) x grad_scale (a power of two) and unscaled the parameter gradients at the end; the checks below
    # follow the SCALED chain (the taps hold scaled tensors), so d(out) and the gradients are put on that scale here (exact)
    gs = hip.last_grad_scale  # (chosen by HipNet.backward from max |d(out)|: the probe rows are not a mean loss's)
    assert gs == orc.probe_scale(mode, float(dout.abs().max()))
    dout = dout * gs
    torch.cuda.synchronize()
    ws = hip.workspace(n).cpu()
    tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[mode]

    def tap(nm, rows, cols, t=False):
        off = hip.ws_offset(n, nm)
        raw = ws[off:off + rows * cols]
        if t and tdt != torch.float32:
            return raw.view(tdt)[:rows * cols].view(rows, cols).float()
        return raw.view(rows, cols).clone()
    sd = {k: v.detach().cpu() for k, v in pf.state_dict().items()}
    G = lambda k: hip.grad_view(grads, k).cpu() * gs
    lin = lambda x, wk, bk: orc.linear(x, sd[wk], sd[bk], mode)
    r = (lambda x: x) if mode == "f32" else orc.ROUND[mode]
    dmm = lambda dy, wk: r(dy) @ r(sd[wk])
    wg = lambda dy, x: (r(dy).double().t() @ r(x).double()).float()
    nhwc = lambda x: x.permute(0, 2, 3, 1).reshape(n, -1, x.shape[1])
    nchw = lambda t2, hw, c: t2.reshape(n, hw, hw, c).permute(0, 3, 1, 2)
    d_out = dout.cpu()[:, :A]
    with torch.no_grad():
        if kind == "mlp":
            hp, bp = "seq_append_fcs.", "base.seq_fcs."
            eh0, eh1, hh0, hh1, out = tap("eh0", n, 256), tap("eh1", n, 256), tap("hh0", n, 256), tap("hh1", n, 256), tap("out", n, 16)
            _close32("base_fc1", eh0, torch.relu(lin(obs, bp + "0.weight", bp + "0.bias")), tag)
            _close32("base_fc2", eh1, torch.relu(lin(eh0, bp + "2.weight", bp + "2.bias")), tag)
            _close32("head_fc0", hh0, torch.relu(lin(eh1, hp + "0.weight", hp + "0.bias")), tag)
            _close32("head_fc1", hh1, torch.relu(lin(hh0, hp + "2.weight", hp + "2.bias")), tag)
            _close32("head_out", out[:, :A], lin(hh1, hp + "4.weight", hp + "4.bias"), tag)
            dhh1, dhh0, hand, deh0 = tap("dhh1", n, 256), tap("dhh0", n, 256), tap("dhc", n, 256), tap("deh0", n, 256)
            _close32("d.head_fc1_pre", dhh1, (hh1 > 0) * dmm(d_out, hp + "4.weight"), tag)
            _close32("d.head_fc0_pre", dhh0, (hh0 > 0) * dmm(dhh1, hp + "2.weight"), tag)
            _close32("d.base_fc2_pre", hand, (eh1 > 0) * dmm(dhh0, hp + "0.weight"), tag)
            _close32("d.base_fc1_pre", deh0, (eh0 > 0) * dmm(hand, bp + "2.weight"), tag)
            for nm, dyv, xv in ((hp + "4", d_out, hh1), (hp + "2", dhh1, hh0), (hp + "0", dhh0, eh1), (bp + "2", hand, eh0), (bp + "0", deh0, obs)):
                _close32("w." + nm + ".weight", G(nm + ".weight"), wg(dyv, xv), tag)
                _close32("w." + nm + ".bias", G(nm + ".bias"), dyv.double().sum(0).float(), tag)
            return
        # ---- conv stack forward (all visual nets)
        enc = {"cnn": "encoder.visual_base.layers.", "cnn_vis": "encoder.layers.", "loco_vis": "encoder.depth_visual_base.layers."}[kind]
        state, img = (orc.split_obs(obs, S) if S > 0 else (None, obs.reshape(-1, 4, 64, 64)))
        img_t = im.cpu().float().view(n, 4, 64, 64)
        assert torch.equal(img_t, r(img))
        c1, c2, c3 = tap("c1", n * 225, 32), tap("c2", n * 36, 64), tap("c3", n * 16, 64)
        _close32("conv1", c1.view(n, 225, 32), nhwc(torch.relu(orc.conv2d(img_t, sd[enc + "0.weight"], sd[enc + "0.bias"], 4, mode))), tag)
        _close32("conv2", c2.view(n, 36, 64), nhwc(torch.relu(orc.conv2d(nchw(c1, 15, 32), sd[enc + "2.weight"], sd[enc + "2.bias"], 2, mode))), tag)
        _close32("conv3", c3.view(n, 16, 64), nhwc(torch.relu(orc.conv2d(nchw(c2, 6, 64), sd[enc + "4.weight"], sd[enc + "4.bias"], 1, mode))), tag)
        flat = nchw(c3, 4, 64).flatten(1)                           # PyTorch's NCHW flatten of conv3's output
        hp = "visual_seq_append_fcs." if kind == "loco_vis" else "seq_append_fcs."
        hh0, hh1, out = tap("hh0", n, 256), tap("hh1", n, 256), tap("out", n, 16)
        dhh1, dhh0 = tap("dhh1", n, 256), tap("dhh0", n, 256)
        dc3 = tap("dc3", n * 16, 64)
        if kind == "cnn":
            vis = tap("vis", n, 512)
            pk, bp = "encoder.visual_projector.projection.0.", "encoder.base.seq_fcs."
            eh0 = tap("eh0", n, 256)
            _close32("visual_projector", vis[:, :256], torch.relu(lin(flat, pk + "weight", pk + "bias")), tag)
            _close32("enc_fc1", eh0, torch.relu(lin(state, bp + "0.weight", bp + "0.bias")), tag)
            _close32("enc_fc2", vis[:, 256:], torch.relu(lin(eh0, bp + "2.weight", bp + "2.bias")), tag)
            head_in = vis
        elif kind == "cnn_vis":
            head_in = flat
        else:
            x0 = tap("x0", n * 16, 64).view(n, 16, 64)
            up = orc.conv2d(nchw(c3, 4, 64), sd["encoder.depth_up_conv.weight"], sd["encoder.depth_up_conv.bias"], 1, mode)
            _close32("up_conv", x0, nhwc(up), tag)
            xl = tap("x2", n * 16, 64).view(n, 16, 64)
            head_in = xl.mean(1)
            _close32("pool_all", tap("pooled", n, 64), head_in, tag)
        _close32("head_fc0", hh0, torch.relu(lin(head_in, hp + "0.weight", hp + "0.bias")), tag)
        _close32("head_fc1", hh1, torch.relu(lin(hh0, hp + "2.weight", hp + "2.bias")), tag)
        _close32("head_out", out[:, :A], lin(hh1, hp + "4.weight", hp + "4.bias"), tag)
        # ---- backward: head chain
        _close32("d.head_fc1_pre", dhh1, (hh1 > 0) * dmm(d_out, hp + "4.weight"), tag)
        _close32("d.head_fc0_pre", dhh0, (hh0 > 0) * dmm(dhh1, hp + "2.weight"), tag)
        for nm, dyv, xv in ((hp + "4", d_out, hh1), (hp + "2", dhh1, hh0), (hp + "0", dhh0, head_in)):
            _close32("w." + nm + ".weight", G(nm + ".weight"), wg(dyv, xv), tag)
            _close32("w." + nm + ".bias", G(nm + ".bias"), dyv.double().sum(0).float(), tag)
        dhead_in = dmm(dhh0, hp + "0.weight")
        if kind == "cnn":
            hand, deh0 = tap("dhc", n, 512), tap("deh0", n, 256)
            _close32("d.head_fc0->concat", hand, dhead_in, tag)
            yv = (vis[:, :256] > 0) * hand[:, :256]
            want3 = (nchw(c3, 4, 64) > 0) * dmm(yv, pk + "weight").view(n, 64, 4, 4)
            _close32("d.visual_projector->dc3", dc3.view(n, 16, 64), nhwc(want3), tag)
            _close32("w.visual_projector.weight", G(pk + "weight"), wg(yv, flat), tag)
            ys = (vis[:, 256:] > 0) * hand[:, 256:]
            _close32("d.enc_fc2->deh0", deh0, (eh0 > 0) * dmm(ys, bp + "2.weight"), tag)
            _close32("w.enc_fc2.weight", G(bp + "2.weight"), wg(ys, eh0), tag)
            _close32("w.enc_fc1.weight", G(bp + "0.weight"), wg(deh0, state), tag)
        elif kind == "cnn_vis":
            want3 = (nchw(c3, 4, 64) > 0) * dhead_in.view(n, 64, 4, 4)
            _close32("d.head_fc0->dc3", dc3.view(n, 16, 64), nhwc(want3), tag)
        else:
            dx0 = tap("dx0", n * 16, 64)
            _close32("d.up_conv->dc3", dc3, (c3 > 0) * (r(dx0) @ r(sd["encoder.depth_up_conv.weight"].view(64, 64))), tag)
            _close32("w.up_conv.weight", G("encoder.depth_up_conv.weight").view(64, 64), (r(dx0).double().t() @ r(c3).double()).float(), tag)
        # ---- conv stack backward from the kernel's own dc3
        w3 = sd[enc + "4.weight"]
        _close32("w.conv3.weight", G(enc + "4.weight"),
                 torch.nn.grad.conv2d_weight(r(nchw(c2, 6, 64)).double(), w3.shape, r(nchw(dc3, 4, 64)).double(), stride=1).float(), tag)
        _close32("w.conv3.bias", G(enc + "4.bias"), dc3.double().sum(0).float(), tag)
        _conv_backward_checks(tap, G, sd, enc, n, c1, c2, dc3, img_t, mode, tag)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("E", [32, 64])
def test_rollout_kernels_stage_by_stage(E, mode, device):
    """The two rollout kernels (rollout_encoder2_kernel / rollout_encoder_kernel + rollout_stack_kernel: conv stack, proprio MLP,
    both nets' layer stacks and heads in LDS, nothing saved) at the bench's E: every tensor they DO leave in HBM — the token
    tensor, each layer's output for both nets, the head outputs — teacher-forced stage by stage: the oracle's stage applied to
    the kernel's own stage input (collector/on_policy.py:95-100 -> nets.py:996-1038). A stage is a chain of contractions, so
    bf16 is gated at the chain's tolerance, f32 at accumulation noise."""
    _MODE[0] = mode
    from vision4leg_amd.torchrl.policies import RolloutActor
    case = dict(util.CASES["loco_b1024"])
    S, A = case["S"], case["A"]
    pf, vf = _build(case, mode, device)
    actor = RolloutActor(pf, vf, E)
    rs = np.random.RandomState(31)
    obs = torch.tensor(util.obs_rows(rs, E, case), dtype=torch.float32)
    out = actor.step(obs.to(device), deterministic=True)
    torch.cuda.synchronize()
    a = actor._actor
    ws = a.ws.cpu()
    hp_, hv_ = pf.hip, vf.hip
    off_v = hp_.ws_floats(E)
    tapp = lambda nm, rows, cols: ws[hp_.ws_offset(E, nm):hp_.ws_offset(E, nm) + rows * cols].view(rows, cols).clone()
    tapv = lambda nm, rows, cols: ws[off_v + hv_.ws_offset(E, nm):off_v + hv_.ws_offset(E, nm) + rows * cols].view(rows, cols).clone()
    tag = "rollout/E%d/%s" % (E, mode)
    tol = 2e-5 if mode == "f32" else 6e-3
    R = E * 17
    sdp = {k: v.detach().cpu() for k, v in pf.state_dict().items()}
    sdv = {k: v.detach().cpu() for k, v in vf.state_dict().items()}
    with torch.no_grad():
        taps = {}
        orc.loco_forward({k: v for k, v in sdp.items() if k != "logstd"}, obs, S, mode, taps)
        x0 = tapp("x0", R, 64).view(E, 17, 64)
        _close32("tokens", x0, taps["x0"], tag, tol=tol)
        for nm, sd, tp, last in (("pf", sdp, tapp, out["mean"].cpu()), ("vf", sdv, tapv, out["value"].cpu())):
            x = x0
            for l in range(2):
                want = orc.transformer_layer(sd, "visual_append_layers.%d" % l, x, mode)
                got = tp("x%d" % (l + 1), R, 64).view(E, 17, 64)
                _close32("%s.layer%d" % (nm, l), got, want, tag, tol=tol)
                x = got
            pooled = torch.cat([x[:, 0], x[:, 1:17].mean(1)], -1)
            want = orc.head(sd, "visual_seq_append_fcs", pooled, 2, mode)
            _close32("%s.head" % nm, last.view(E, -1), want, tag, tol=tol)

"""TEST INFRASTRUCTURE ONLY -- CPU (torch fp32) statements of what each HIP kernel computes.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product (hallucidet_amd/) never does.  Each function restates the ATen /
torchvision op the reference reaches at the cited call site, on CPU tensors, in fp32.
Inputs that the HIP path holds in fp16 are passed in already rounded to fp16 so the
only differences left are accumulation order and the final fp16 rounding.
"""
import math

import torch
import torch.nn.functional as F


def nhwc_to_nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def nchw_to_nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def upsample_deterministic(x, upscale):
    """Reference: src/segmentation_models/decoders/unet/decoder.py:7-8 (restated)."""
    n, c, h, w = x.shape
    return x[:, :, :, None, :, None].expand(-1, -1, -1, upscale, -1, upscale).reshape(n, c, h * upscale, w * upscale)


def conv2d_nhwc(x, w_flat, KH, KW, *, x2=None, bias=None, res=None, stride=1, pad=0, up1=False, act=0):
    """x [N,H,W,C1] (fp16 values), w_flat [Cout, KH*KW*Cin] in (kh,kw,ci) order.  Returns fp32 NHWC pre-rounding
    plus the (sum, sumsq) statistics of the fp16-rounded pre-activation."""
    xi = nhwc_to_nchw(x.float())
    if up1:
        xi = upsample_deterministic(xi, 2)
    if x2 is not None:
        xi = torch.cat([xi, nhwc_to_nchw(x2.float())], dim=1)
    cin = xi.shape[1]
    cout = w_flat.shape[0]
    w = w_flat.float().view(cout, KH, KW, cin).permute(0, 3, 1, 2).contiguous()
    y = F.conv2d(xi, w, None if bias is None else bias.float(), stride=stride, padding=pad)
    if res is not None:
        y = y + nhwc_to_nchw(res.float())
    yr = y.half().float()
    stats = torch.stack([yr.sum(dim=(0, 2, 3)), (yr * yr).sum(dim=(0, 2, 3))])
    if act == 1:
        y = torch.relu(y)
    elif act == 2:
        y = torch.sigmoid(y)
    return nchw_to_nhwc(y), stats


def conv2d_dgrad_nhwc(dy, w_flat, KH, KW, cin, *, stride, pad, in_hw):
    """Data gradient of conv2d(x, w) w.r.t. x given dy [N,Ho,Wo,Cout]; returns NHWC fp32 [N,H,W,cin]."""
    cout = w_flat.shape[0]
    w = w_flat.float().view(cout, KH, KW, cin).permute(0, 3, 1, 2).contiguous()
    H, W = in_hw
    g = nhwc_to_nchw(dy.float())
    dx = torch.nn.grad.conv2d_input((g.shape[0], cin, H, W), w, g, stride=stride, padding=pad)
    return nchw_to_nhwc(dx)


def conv2d_wgrad_nhwc(x, dy, KH, KW, *, x2=None, stride=1, pad=0, up1=False):
    """Weight gradient in [Cout, KH*KW*Cin] (kh,kw,ci) order, fp32."""
    xi = nhwc_to_nchw(x.float())
    if up1:
        xi = upsample_deterministic(xi, 2)
    if x2 is not None:
        xi = torch.cat([xi, nhwc_to_nchw(x2.float())], dim=1)
    g = nhwc_to_nchw(dy.float())
    cout, cin = g.shape[1], xi.shape[1]
    dw = torch.nn.grad.conv2d_weight(xi, (cout, cin, KH, KW), g, stride=stride, padding=pad)
    return dw.permute(0, 2, 3, 1).reshape(cout, KH * KW * cin)


def bn_train_nhwc(y, gamma, beta, eps, res=None, relu=True):
    """Training-mode BatchNorm2d on NHWC (statistics over N,H,W; biased variance) + residual + ReLU."""
    yf = y.float()
    mean = yf.mean(dim=(0, 1, 2))
    var = yf.var(dim=(0, 1, 2), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + eps)
    z = (yf - mean) * invstd * gamma + beta
    if res is not None:
        z = z + res.float()
    if relu:
        z = torch.relu(z)
    return z, mean, invstd


def maxpool3x3s2_nhwc(x):
    return nchw_to_nhwc(F.max_pool2d(nhwc_to_nchw(x.float()), 3, 2, 1))


def nearest_resize_nchw(x, Ho, Wo):
    """custom_generalized_transform.py:80-87 -- F.interpolate with the default mode ('nearest')."""
    return F.interpolate(x, size=[Ho, Wo])


def nms_sorted(boxes, thr):
    """Greedy NMS over boxes already sorted by descending score (torchvision nms CPU kernel [EXT], restated
    with scalar fp32 arithmetic).  Returns bool keep mask in the same order."""
    n = boxes.shape[0]
    b = boxes.detach().float().numpy().astype("float32")
    import numpy as np

    areas = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    suppressed = np.zeros(n, dtype=bool)
    keep = np.zeros(n, dtype=bool)
    for i in range(n):
        if suppressed[i]:
            continue
        keep[i] = True
        if i + 1 >= n:
            break
        xx1 = np.maximum(b[i, 0], b[i + 1:, 0])
        yy1 = np.maximum(b[i, 1], b[i + 1:, 1])
        xx2 = np.minimum(b[i, 2], b[i + 1:, 2])
        yy2 = np.minimum(b[i, 3], b[i + 1:, 3])
        w = np.maximum(np.float32(0), xx2 - xx1)
        h = np.maximum(np.float32(0), yy2 - yy1)
        inter = (w * h).astype("float32")
        ovr = inter / ((areas[i] + areas[i + 1:]).astype("float32") - inter)
        suppressed[i + 1:] |= ovr > np.float32(thr)
    return torch.from_numpy(keep)


def _bilinear(feat, y, x):
    """feat [C,H,W]; torchvision roi_align bilinear_interpolate [EXT]."""
    C, H, W = feat.shape
    if y < -1.0 or y > H or x < -1.0 or x > W:
        return torch.zeros(C)
    y = max(y, 0.0)
    x = max(x, 0.0)
    yl, xl = int(y), int(x)
    if yl >= H - 1:
        yh = yl = H - 1
        y = float(yl)
    else:
        yh = yl + 1
    if xl >= W - 1:
        xh = xl = W - 1
        x = float(xl)
    else:
        xh = xl + 1
    ly, lx = y - yl, x - xl
    hy, hx = 1.0 - ly, 1.0 - lx
    return hy * hx * feat[:, yl, xl] + hy * lx * feat[:, yl, xh] + ly * hx * feat[:, yh, xl] + ly * lx * feat[:, yh, xh]


def roi_align_nchw(feat, rois, PH, PW, scale, sr):
    """Slow scalar RoIAlign (aligned=False) [EXT]; feat [N,C,H,W] fp32, rois [R,5].  Returns [R,C,PH,PW]."""
    R = rois.shape[0]
    C = feat.shape[1]
    out = torch.zeros(R, C, PH, PW)
    f32 = lambda v: float(torch.tensor(v, dtype=torch.float32))
    for r in range(R):
        n = int(rois[r, 0])
        rsw, rsh, rew, reh = [f32(float(rois[r, k]) * scale) for k in (1, 2, 3, 4)]
        rw, rh = max(rew - rsw, 1.0), max(reh - rsh, 1.0)
        bh, bw = rh / PH, rw / PW
        gh = sr if sr > 0 else math.ceil(rh / PH)
        gw = sr if sr > 0 else math.ceil(rw / PW)
        cnt = max(gh * gw, 1)
        for ph in range(PH):
            for pw in range(PW):
                acc = torch.zeros(C)
                for iy in range(gh):
                    y = rsh + ph * bh + (iy + 0.5) * bh / gh
                    for ix in range(gw):
                        x = rsw + pw * bw + (ix + 0.5) * bw / gw
                        acc += _bilinear(feat[n], y, x)
                out[r, :, ph, pw] = acc / cnt
    return out


def box_iou(a, b):
    """torchvision.ops.box_iou [EXT]."""
    area1 = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area2 = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[:, :2])
    rb = torch.min(a[:, None, 2:], b[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (area1[:, None] + area2 - inter)


def adam_reference(p, g, m, v, *, lr, beta1, beta2, eps, clip_value, inv_scale, step):
    g = g * inv_scale
    if clip_value > 0:
        g = g.clamp(-clip_value, clip_value)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v

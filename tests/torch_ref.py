"""Independent torch-CPU (float64 autograd) model of the reference-defined network, used ONLY to validate
the oracle (SURVEY.md §8c "optional secondary check").  Deviations of the reference from textbook
ResNet that are reproduced here: loss is a batch SUM (resnet.cu:1806-1811), 3x3 stride-2 projection
shortcuts (resnet.cu:770-775), stride on the 3x3 (resnet.cu:743-746), BN without running stats with
eps shared with Adam, max-pool backward = plain overwrite scatter (resnet.cu:476-494; last writer in
(n,oh,ow) order wins -- the deterministic execution the oracle fixes).
Tensors here are NCHW (torch native); helpers convert from/to the oracle's NHWC.
"""
import numpy as np
import torch
import torch.nn.functional as F


def nhwc_to_nchw(a):
    return np.ascontiguousarray(np.transpose(a, (0, 3, 1, 2)))


def nchw_to_nhwc(a):
    return np.ascontiguousarray(np.transpose(a, (0, 2, 3, 1)))


class MaxPoolOverwrite(torch.autograd.Function):
    """3x3/s2/p1 max-pool whose backward overwrites instead of accumulating."""

    @staticmethod
    def forward(ctx, x, k, s):
        y, idx = F.max_pool2d(x, k, s, k // 2, return_indices=True)
        ctx.save_for_backward(idx)
        ctx.shape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        N, C, H, W = ctx.shape
        dx = torch.zeros(N, C, H * W, dtype=dy.dtype)
        idx2, dy2 = idx.reshape(N, C, -1), dy.reshape(N, C, -1)
        for j in range(idx2.shape[2]):  # (oh, ow) scan order: later outputs overwrite earlier ones
            dx.scatter_(2, idx2[:, :, j:j + 1], dy2[:, :, j:j + 1])
        return dx.reshape(N, C, H, W), None, None


def bn_train(x, g, b, eps):
    m = x.mean(dim=(0, 2, 3), keepdim=True)
    v = ((x - m) ** 2).mean(dim=(0, 2, 3), keepdim=True)
    return (x - m) / torch.sqrt(v + eps) * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)


class TorchNet:
    def __init__(self, dims, params, eps=1e-7, dtype=torch.float64):
        self.dims, self.eps = dims, eps
        self.p = [torch.tensor(np.asarray(a), dtype=dtype, requires_grad=True) for a in params]
        self.acts = {}

    def _unit(self, x, i, K, C, k, stride, relu, name):
        w = self.p[i].view(K, C, k, k)
        y = F.conv2d(x, w, stride=stride, padding=k // 2)
        y.retain_grad()
        self.acts[name + "_conv"] = y
        z = bn_train(y, self.p[i + 1], self.p[i + 2], self.eps)
        if relu:
            z = F.relu(z)
        z.retain_grad()
        self.acts[name] = z
        return z

    def forward(self, images_nchw, labels):
        d = self.dims
        x = torch.tensor(images_nchw, dtype=self.p[0].dtype)
        f = d["init_conv_filters"]
        li = 0
        x = self._unit(x, li, f, 3, d["init_kernel_dim"], d["init_conv_stride"], True, "stem")
        li += 3
        x = MaxPoolOverwrite.apply(x, d["init_maxpool_dim"], d["init_maxpool_stride"])
        x.retain_grad()
        self.acts["pool"] = x
        inc, red, ex = f, f, 4 * f
        for b in range(d["n_conv_blocks"]):
            stride = 1
            if d["is_block_spatial_reduction"][b]:
                stride, red, ex = 2, red * 2, ex * 2
            r = self._unit(x, li, red, inc, 1, 1, True, "b%d_red" % b); li += 3
            s = self._unit(r, li, red, red, 3, stride, True, "b%d_spa" % b); li += 3
            e = self._unit(s, li, ex, red, 1, 1, False, "b%d_exp" % b); li += 3
            res = x
            if inc != ex:
                res = self._unit(x, li, ex, inc, 3 if stride == 2 else 1, stride, False, "b%d_proj" % b); li += 3
            x = F.relu(e + res)
            x.retain_grad()
            self.acts["b%d_out" % b] = x
            inc = ex
        pooled = x.mean(dim=(2, 3))
        logits = pooled @ self.p[li].view(inc, d["output"])
        logits.retain_grad()
        self.acts["logits"] = logits
        self.pred = torch.softmax(logits, dim=1)
        lab = torch.tensor(np.asarray(labels), dtype=torch.long)
        self.loss = -torch.log(self.pred[torch.arange(len(lab)), lab]).sum()
        return self.loss

    def backward(self):
        self.loss.backward()
        return [p.grad.detach().numpy() for p in self.p]


# ---- bf16-storage emulation (BASELINE configs[4]): the same float64 model with every ACTIVATION tensor and every activation
# gradient rounded to bf16 (round to nearest even) where the product stores it: convolution outputs of the bottleneck
# blocks, BN(+ReLU) outputs, block outputs, and the gradients flowing back through the same tensors; bottleneck weights
# are rounded when used (fp32 master copies).  The stem convolution's own output stays unrounded, like the product's.
# It is the yardstick for how far bf16 storage ALONE moves a gradient from the fp32 oracle's (tests/test_gpu_bf16.py): on
# these small random-init nets a rounded pre-activation flips ~0.3 % of the ReLU gates per layer, which is ~5 % in rel-L2.
def _rb(t):
    return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)


class _RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _rb(x)

    @staticmethod
    def backward(ctx, g):
        return _rb(g)


class TorchNetBF16(TorchNet):
    def _unit(self, x, i, K, C, k, stride, relu, name, residual=None):
        w = self.p[i].view(K, C, k, k)
        stem = name == "stem"
        w = w + (_rb(w.detach()) - w.detach())  # rounded value, gradient to the fp32 master copy
        if stem:  # the stem multiplies the bf16-rounded image (kernels_stem_bf16.hip); its output stays an fp32 tensor.  (Its weight
            x = _rb(x)  # gradient also rounds dY on the way in: 2^-9 relative noise per element, averaged over 10^5..10^6 terms)
        y = F.conv2d(x, w, stride=stride, padding=k // 2)
        if not stem:
            y = _RoundBF16.apply(y)
        z = bn_train(y, self.p[i + 1], self.p[i + 2], self.eps)
        if residual is not None:
            z = F.relu(z + residual)  # BN + addVec + doActivation are one kernel: one rounding
        elif relu:
            z = F.relu(z)
        return _RoundBF16.apply(z)

    def forward(self, images_nchw, labels):
        d = self.dims
        x = torch.tensor(images_nchw, dtype=self.p[0].dtype)
        f = d["init_conv_filters"]
        li = 0
        x = self._unit(x, li, f, 3, d["init_kernel_dim"], d["init_conv_stride"], True, "stem")
        li += 3
        x = MaxPoolOverwrite.apply(x, d["init_maxpool_dim"], d["init_maxpool_stride"])
        inc, red, ex = f, f, 4 * f
        for b in range(d["n_conv_blocks"]):
            stride = 1
            if d["is_block_spatial_reduction"][b]:
                stride, red, ex = 2, red * 2, ex * 2
            r = self._unit(x, li, red, inc, 1, 1, True, "red"); li += 3
            s = self._unit(r, li, red, red, 3, stride, True, "spa"); li += 3
            le = li; li += 3
            res = x
            if inc != ex:
                res = self._unit(x, li, ex, inc, 3 if stride == 2 else 1, stride, False, "proj"); li += 3
            x = self._unit(s, le, ex, red, 1, 1, False, "exp", residual=res)
            self.acts["b%d_out" % b] = x
            inc = ex
        pooled = x.mean(dim=(2, 3))
        logits = pooled @ self.p[li].view(inc, d["output"])
        self.pred = torch.softmax(logits, dim=1)
        lab = torch.tensor(np.asarray(labels), dtype=torch.long)
        self.loss = -torch.log(self.pred[torch.arange(len(lab)), lab]).sum()
        return self.loss

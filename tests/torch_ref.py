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
    def forward(ctx, x, k, s, given_idx=None):
        y, idx = F.max_pool2d(x, k, s, k // 2, return_indices=True)
        if given_idx is not None:  # another execution's arg-max positions (per-plane flat indices): take ITS choices
            idx = given_idx
            N, C = x.shape[:2]
            y = x.reshape(N, C, -1).gather(2, idx.reshape(N, C, -1)).reshape(idx.shape)
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
        return dx.reshape(N, C, H, W), None, None, None


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
# are rounded when used (fp32 master copies).  The stem convolution's own output is rounded too when the product stores it as bf16
# (stem_bf16: Trainer.stem_dtype() -- the matrix-core stem; the VALU stem of inputs that are not a multiple of 32 keeps fp32 tensors).
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
    """gates (optional): the DISCRETE decisions of another execution of the same step -- {name: bool NCHW array} for every ReLU
    ("stem", "b%d_red", "b%d_spa", "b%d_out": that execution's stored activation > 0) and "max_inds" (its per-plane arg-max
    positions).  With them the model takes those decisions instead of its own, so the two executions' gradients differ by
    rounding alone: a pre-activation within bf16 rounding of 0 no longer shows up as an O(1) difference of that element."""

    def __init__(self, dims, params, eps=1e-7, dtype=torch.float64, gates=None, stem_bf16=False):
        super().__init__(dims, params, eps, dtype)
        self.gates = gates
        self.stem_bf16 = stem_bf16

    def _relu(self, z, key):
        if self.gates is None:
            return F.relu(z)
        return z * torch.tensor(np.asarray(self.gates[key]), dtype=z.dtype)

    def _unit(self, x, i, K, C, k, stride, relu, name, residual=None, key=None):
        w = self.p[i].view(K, C, k, k)
        stem = name == "stem"
        w = w + (_rb(w.detach()) - w.detach())  # rounded value, gradient to the fp32 master copy
        if stem:  # the stem multiplies the bf16-rounded image (kernels_stem_bf16.hip).  (Its weight gradient rounds dY on the way in when
            x = _rb(x)  # that tensor is fp32: 2^-9 relative noise per element, averaged over 10^5..10^6 terms)
        y = F.conv2d(x, w, stride=stride, padding=k // 2)
        if not stem or self.stem_bf16:
            y = _RoundBF16.apply(y)
        z = bn_train(y, self.p[i + 1], self.p[i + 2], self.eps)
        if residual is not None:
            z = self._relu(z + residual, key)  # BN + addVec + doActivation are one kernel: one rounding
        elif relu:
            z = self._relu(z, key)
        return _RoundBF16.apply(z)

    def forward(self, images_nchw, labels):
        d = self.dims
        x = torch.tensor(images_nchw, dtype=self.p[0].dtype)
        f = d["init_conv_filters"]
        li = 0
        x = self._unit(x, li, f, 3, d["init_kernel_dim"], d["init_conv_stride"], True, "stem", key="stem")
        li += 3
        given = None if self.gates is None else torch.tensor(np.asarray(self.gates["max_inds"]), dtype=torch.long)
        x = MaxPoolOverwrite.apply(x, d["init_maxpool_dim"], d["init_maxpool_stride"], given)
        inc, red, ex = f, f, 4 * f
        for b in range(d["n_conv_blocks"]):
            stride = 1
            if d["is_block_spatial_reduction"][b]:
                stride, red, ex = 2, red * 2, ex * 2
            r = self._unit(x, li, red, inc, 1, 1, True, "red", key="b%d_red" % b); li += 3
            s = self._unit(r, li, red, red, 3, stride, True, "spa", key="b%d_spa" % b); li += 3
            le = li; li += 3
            res = x
            if inc != ex:
                res = self._unit(x, li, ex, inc, 3 if stride == 2 else 1, stride, False, "proj"); li += 3
            x = self._unit(s, le, ex, red, 1, 1, False, "exp", residual=res, key="b%d_out" % b)
            self.acts["b%d_out" % b] = x
            inc = ex
        pooled = x.mean(dim=(2, 3))
        logits = pooled @ self.p[li].view(inc, d["output"])
        self.pred = torch.softmax(logits, dim=1)
        lab = torch.tensor(np.asarray(labels), dtype=torch.long)
        self.loss = -torch.log(self.pred[torch.arange(len(lab)), lab]).sum()
        return self.loss


def gates_of(tr, dims):
    """the discrete decisions of a product forward pass (resnet_amd.Trainer after forward_pass), as TorchNetBF16(gates=...) takes them"""
    g = {"stem": tr.activation("init_conv_activated") > 0}
    mi = tr.activation("max_inds").astype(np.int64)  # flat NCHW indices into the stem's output tensor
    N, C, Hp, _ = mi.shape
    Hs = dims["input"] // dims["init_conv_stride"]
    g["max_inds"] = mi - ((np.arange(N)[:, None, None, None] * C + np.arange(C)[None, :, None, None]) * Hs * Hs)
    for b in range(dims["n_conv_blocks"]):
        g["b%d_red" % b] = tr.activation("conv_blocks/%02d/reduction_activated" % b) > 0
        g["b%d_spa" % b] = tr.activation("conv_blocks/%02d/spatial_activated" % b) > 0
        g["b%d_out" % b] = tr.activation("conv_blocks/%02d/output_activated" % b) > 0
    return g

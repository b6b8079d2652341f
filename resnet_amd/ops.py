"""numpy-in / numpy-out access to the operator layer of the C-ABI (mi_op_*, include/resnet_mi.h):
device buffers are allocated, filled, the HIP kernel runs, results are copied back.  Arrays are NCHW
float32 (weights KCRS).  Used by the parity tests; the trainer calls the same kernels stream-ordered."""
import ctypes as C

import numpy as np

from . import binding as B


class DeviceArray:
    def __init__(self, lib, arr=None, shape=None, dtype=np.float32):
        self.L = lib
        if arr is not None:
            arr = np.ascontiguousarray(arr)
            shape, dtype = arr.shape, arr.dtype
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = lib.mi_malloc(max(self.nbytes, 4))
        if not self.ptr:
            raise MemoryError(lib.mi_last_error().decode())
        if arr is not None:
            lib.mi_copy_to_device(self.ptr, arr.ctypes.data, self.nbytes)

    def get(self):
        out = np.empty(self.shape, self.dtype)
        self.L.mi_copy_to_host(out.ctypes.data, self.ptr, self.nbytes)
        return out

    def free(self):
        if self.ptr:
            self.L.mi_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Ops:
    def __init__(self):
        self.L = B.load()

    def dev(self, arr=None, shape=None, dtype=np.float32):
        return DeviceArray(self.L, arr, shape, dtype)

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.L.mi_last_error().decode()))

    def conv_fwd(self, x, w, stride):
        N, Cc, H, _ = x.shape
        K, _, k, _ = w.shape
        dx, dw = self.dev(x), self.dev(w)
        dy = self.dev(shape=(N, K, H // stride, H // stride))
        self._chk(self.L.mi_op_conv_fwd(dx.ptr, dw.ptr, dy.ptr, N, Cc, H, K, k, stride), "conv_fwd")
        return dy.get()

    def conv_dgrad(self, w, dy, H, stride, dx_init=None):
        K, Cc, k, _ = w.shape
        N = dy.shape[0]
        dw_, ddy = self.dev(w), self.dev(dy)
        ddx = self.dev(dx_init) if dx_init is not None else self.dev(shape=(N, Cc, H, H))
        self._chk(self.L.mi_op_conv_dgrad(dw_.ptr, ddy.ptr, ddx.ptr, N, Cc, H, K, k, stride, 0 if dx_init is None else 1), "conv_dgrad")
        return ddx.get()

    def conv_wgrad(self, x, dy, k, stride):
        N, Cc, H, _ = x.shape
        K = dy.shape[1]
        dx, ddy = self.dev(x), self.dev(dy)
        dw = self.dev(shape=(K, Cc, k, k))
        self._chk(self.L.mi_op_conv_wgrad(dx.ptr, ddy.ptr, dw.ptr, N, Cc, H, K, k, stride), "conv_wgrad")
        return dw.get()

    def bn_fwd(self, x, gamma, beta, eps, relu, residual=None):
        N, Cc, H, _ = x.shape
        dx, dg, db = self.dev(x), self.dev(gamma), self.dev(beta)
        dm, dv, dy = self.dev(shape=(Cc,)), self.dev(shape=(Cc,)), self.dev(shape=x.shape)
        if residual is None:
            self._chk(self.L.mi_op_bn_fwd(dx.ptr, dg.ptr, db.ptr, dm.ptr, dv.ptr, dy.ptr, N, Cc, H, eps, int(relu)), "bn_fwd")
        else:
            dr = self.dev(residual)
            self._chk(self.L.mi_op_bn_fwd_add_relu(dx.ptr, dg.ptr, db.ptr, dr.ptr, dm.ptr, dv.ptr, dy.ptr, N, Cc, H, eps), "bn_fwd_add_relu")
        return dm.get(), dv.get(), dy.get()

    def bn_bwd(self, x, gamma, beta, means, vars_, dy, eps, mask_mode, mask_src=None):
        N, Cc, H, _ = x.shape
        dx, dg, db, dm, dv, ddy = (self.dev(a) for a in (x, gamma, beta, means, vars_, dy))
        dmask = self.dev(mask_src) if mask_src is not None else None
        out, ogam, obet = self.dev(shape=x.shape), self.dev(shape=(Cc,)), self.dev(shape=(Cc,))
        self._chk(self.L.mi_op_bn_bwd(dx.ptr, dg.ptr, db.ptr, dm.ptr, dv.ptr, ddy.ptr, dmask.ptr if dmask else None,
                                      out.ptr, ogam.ptr, obet.ptr, N, Cc, H, eps, mask_mode), "bn_bwd")
        return out.get(), ogam.get(), obet.get()

    def bn_bwd_gate(self, x, gamma, beta, means, vars_, dy, eps, mask_src):
        """BN backward with dy gated by mask_src > 0; also returns the gated dy (mi_op_bn_bwd_gate)"""
        N, Cc, H, _ = x.shape
        dx, dg, db, dm, dv, ddy, dmask = (self.dev(a) for a in (x, gamma, beta, means, vars_, dy, mask_src))
        out, gated, ogam, obet = self.dev(shape=x.shape), self.dev(shape=x.shape), self.dev(shape=(Cc,)), self.dev(shape=(Cc,))
        self._chk(self.L.mi_op_bn_bwd_gate(dx.ptr, dg.ptr, db.ptr, dm.ptr, dv.ptr, ddy.ptr, dmask.ptr, gated.ptr, out.ptr, ogam.ptr,
                                           obet.ptr, N, Cc, H, eps), "bn_bwd_gate")
        return out.get(), ogam.get(), obet.get(), gated.get()

    def maxpool_fwd(self, x, k, stride):
        N, Cc, H, _ = x.shape
        Ho = H // stride
        dx, dy, di = self.dev(x), self.dev(shape=(N, Cc, Ho, Ho)), self.dev(shape=(N, Cc, Ho, Ho), dtype=np.int32)
        self._chk(self.L.mi_op_maxpool_fwd(dx.ptr, dy.ptr, di.ptr, N, Cc, H, k, stride), "maxpool_fwd")
        return dy.get(), di.get()

    def maxpool_bwd(self, idx, dy, H, k, stride):
        N, Cc = dy.shape[:2]
        di, ddy, dx = self.dev(idx.astype(np.int32)), self.dev(dy), self.dev(shape=(N, Cc, H, H))
        self._chk(self.L.mi_op_maxpool_bwd(di.ptr, ddy.ptr, dx.ptr, N, Cc, H, k, stride), "maxpool_bwd")
        return dx.get()

    def avgpool_fwd(self, x):
        N, Cc, H, _ = x.shape
        dx, dy = self.dev(x), self.dev(shape=(N, Cc))
        self._chk(self.L.mi_op_avgpool_fwd(dx.ptr, dy.ptr, N, Cc, H), "avgpool_fwd")
        return dy.get()

    def avgpool_bwd(self, dy, H):
        N, Cc = dy.shape
        ddy, dx = self.dev(dy), self.dev(shape=(N, Cc, H, H))
        self._chk(self.L.mi_op_avgpool_bwd(ddy.ptr, dx.ptr, N, Cc, H), "avgpool_bwd")
        return dx.get()

    def relu_deriv(self, x, up):
        dx, du, do = self.dev(x), self.dev(up), self.dev(shape=x.shape)
        self._chk(self.L.mi_op_relu_deriv(dx.ptr, du.ptr, do.ptr, x.size), "relu_deriv")
        return do.get()

    def matmul(self, a, b, mode="nn"):
        da, db = self.dev(a), self.dev(b)
        if mode == "nn":
            m, k = a.shape
            n = b.shape[1]
            fn = self.L.mi_op_matmul
        elif mode == "lt":  # a is [k x m]
            k, m = a.shape
            n = b.shape[1]
            fn = self.L.mi_op_matmul_lt
        else:  # "rt": b is [n x k]
            m, k = a.shape
            n = b.shape[0]
            fn = self.L.mi_op_matmul_rt
        do = self.dev(shape=(m, n))
        self._chk(fn(da.ptr, db.ptr, do.ptr, m, k, n), "matmul_" + mode)
        return do.get()

    def softmax(self, x):
        dx, do = self.dev(x), self.dev(shape=x.shape)
        self._chk(self.L.mi_op_softmax(dx.ptr, do.ptr, x.shape[0], x.shape[1]), "softmax")
        return do.get()

    def ce_deriv(self, pred, labels):
        dp, dl, dd = self.dev(pred), self.dev(labels.astype(np.int32)), self.dev(shape=pred.shape)
        self._chk(self.L.mi_op_ce_deriv(dp.ptr, dl.ptr, dd.ptr, pred.shape[0], pred.shape[1]), "ce_deriv")
        return dd.get()

    def adam(self, p, g, m, v, lr, wd, b1, b2, cur_b1, cur_b2, eps):
        dp, dg, dm, dv = (self.dev(a) for a in (p, g, m, v))
        flag = self.dev(np.zeros(1, np.int32))
        self._chk(self.L.mi_op_adam(dp.ptr, dg.ptr, dm.ptr, dv.ptr, p.size, lr, wd, b1, b2, cur_b1, cur_b2, eps, flag.ptr), "adam")
        return dp.get(), dm.get(), dv.get(), int(flag.get()[0])

    def nhwc_to_nchw(self, x):
        N, H, W, Cc = x.shape
        dx, do = self.dev(x), self.dev(shape=(N, Cc, H, W))
        self._chk(self.L.mi_op_nhwc_to_nchw(dx.ptr, do.ptr, N, H, W, Cc), "nhwc_to_nchw")
        return do.get()

    # ---- typed operators: activation tensors stored as bf16 on the device, numpy float32 at this boundary ----
    # (values are rounded to bf16 -- round to nearest even -- on the way in; outputs come back widened, so a bf16 result
    # is a float32 array whose values are exactly the stored bf16 numbers)
    def dev_t(self, arr, dt):
        """upload a float32 array as a tensor of storage type dt"""
        d32 = self.dev(np.ascontiguousarray(arr, np.float32))
        if dt == B.MI_DTYPE_F32:
            return d32
        out = DeviceArray(self.L, shape=arr.shape, dtype=np.uint16)
        self._chk(self.L.mi_op_convert(d32.ptr, B.MI_DTYPE_F32, out.ptr, B.MI_DTYPE_BF16, arr.size), "convert")
        return out

    def new_t(self, shape, dt):
        return DeviceArray(self.L, shape=shape, dtype=np.float32 if dt == B.MI_DTYPE_F32 else np.uint16)

    def get_t(self, darr, dt):
        if dt == B.MI_DTYPE_F32:
            return darr.get()
        out = self.dev(shape=darr.shape)
        self._chk(self.L.mi_op_convert(darr.ptr, B.MI_DTYPE_BF16, out.ptr, B.MI_DTYPE_F32, int(np.prod(darr.shape))), "convert")
        return out.get()

    def conv_fwd_bf16(self, x, w, stride):
        N, Cc, H, _ = x.shape
        K, _, k, _ = w.shape
        BF = B.MI_DTYPE_BF16
        dx, dw = self.dev_t(x, BF), self.dev(w)
        dy = self.new_t((N, K, H // stride, H // stride), BF)
        self._chk(self.L.mi_op_conv_fwd_bf16(dx.ptr, dw.ptr, dy.ptr, N, Cc, H, K, k, stride), "conv_fwd_bf16")
        return self.get_t(dy, BF)

    def conv_fwd_bf16_cl(self, x, w, stride):
        """3x3 forward on channel-last zero-padded operands (kernels_cl_bf16.hip)"""
        N, Cc, H, _ = x.shape
        K = w.shape[0]
        BF = B.MI_DTYPE_BF16
        dx, dw = self.dev_t(x, BF), self.dev(w)
        dy = self.new_t((N, K, H // stride, H // stride), BF)
        self._chk(self.L.mi_op_conv_fwd_bf16_cl(dx.ptr, dw.ptr, dy.ptr, N, Cc, H, K, stride), "conv_fwd_bf16_cl")
        return self.get_t(dy, BF)

    def conv_wgrad_bf16_cl(self, x, dy, stride):
        N, Cc, H, _ = x.shape
        K = dy.shape[1]
        BF = B.MI_DTYPE_BF16
        dx, ddy = self.dev_t(x, BF), self.dev_t(dy, BF)
        dw = self.dev(shape=(K, Cc, 3, 3))
        self._chk(self.L.mi_op_conv_wgrad_bf16_cl(dx.ptr, ddy.ptr, dw.ptr, N, Cc, H, K, stride), "conv_wgrad_bf16_cl")
        return dw.get()

    def bn_fwd_cl_bf16(self, x, gamma, beta, eps, residual=None, par=False):
        """BN (+ residual) + ReLU written twice (bn_apply_cl_kernel): returns means, vars, y (NCHW) and the channel-last copy -- one plane
        [N][H+2][H+2][C], or (par) the four parity planes [N][4][H/2+1][H/2+1][C]"""
        N, Cc, H, _ = x.shape
        BF = B.MI_DTYPE_BF16
        dx, dg, db = self.dev_t(x, BF), self.dev(gamma), self.dev(beta)
        dr = self.dev_t(residual, BF) if residual is not None else None
        dm, dv, dy = self.dev(shape=(Cc,)), self.dev(shape=(Cc,)), self.new_t(x.shape, BF)
        shp = (N, 4, H // 2 + 1, H // 2 + 1, Cc) if par else (N, H + 2, H + 2, Cc)
        ycl = self.dev_t(np.zeros(shp, np.float32), BF)   # zero halo
        self._chk(self.L.mi_op_bn_fwd_cl_bf16(dx.ptr, dg.ptr, db.ptr, dr.ptr if dr else None, dm.ptr, dv.ptr, dy.ptr, ycl.ptr, N, Cc, H, eps, int(par)),
                  "bn_fwd_cl_bf16")
        return dm.get(), dv.get(), self.get_t(dy, BF), self.get_t(ycl, BF)

    def conv1x1_fwd_bf16_cl(self, x, w):
        N, Cc, H, _ = x.shape
        K = w.shape[0]
        BF = B.MI_DTYPE_BF16
        dx, dw = self.dev_t(x, BF), self.dev(w)
        dy = self.new_t((N, K, H, H), BF)
        self._chk(self.L.mi_op_conv1x1_fwd_bf16_cl(dx.ptr, dw.ptr, dy.ptr, N, Cc, H, K), "conv1x1_fwd_bf16_cl")
        return self.get_t(dy, BF)

    def conv_wgrad_bf16_cl2(self, x, dy, stride=2):
        """3x3, both operands re-laid channel-last (cl_wgrad2_kernel)"""
        N, Cc, H, _ = x.shape
        K = dy.shape[1]
        BF = B.MI_DTYPE_BF16
        dx, ddy = self.dev_t(x, BF), self.dev_t(dy, BF)
        dw = self.dev(shape=(K, Cc, 3, 3))
        self._chk(self.L.mi_op_conv_wgrad_bf16_cl2(dx.ptr, ddy.ptr, dw.ptr, N, Cc, H, K, stride), "conv_wgrad_bf16_cl2")
        return dw.get()

    def conv_dgrad_bf16_cl(self, w, dy, H, dx_init=None, stride=1):
        K, Cc, k, _ = w.shape
        N = dy.shape[0]
        BF = B.MI_DTYPE_BF16
        dw_, ddy = self.dev(w), self.dev_t(dy, BF)
        ddx = self.dev_t(dx_init, BF) if dx_init is not None else self.new_t((N, Cc, H, H), BF)
        self._chk(self.L.mi_op_conv_dgrad_bf16_cl(dw_.ptr, ddy.ptr, ddx.ptr, N, Cc, H, K, stride, 0 if dx_init is None else 1), "conv_dgrad_bf16_cl")
        return self.get_t(ddx, BF)

    def conv_dgrad_bf16(self, w, dy, H, stride, dx_init=None):
        K, Cc, k, _ = w.shape
        N = dy.shape[0]
        BF = B.MI_DTYPE_BF16
        dw_, ddy = self.dev(w), self.dev_t(dy, BF)
        ddx = self.dev_t(dx_init, BF) if dx_init is not None else self.new_t((N, Cc, H, H), BF)
        self._chk(self.L.mi_op_conv_dgrad_bf16(dw_.ptr, ddy.ptr, ddx.ptr, N, Cc, H, K, k, stride, 0 if dx_init is None else 1), "conv_dgrad_bf16")
        return self.get_t(ddx, BF)

    def conv_wgrad_bf16(self, x, dy, k, stride):
        N, Cc, H, _ = x.shape
        K = dy.shape[1]
        BF = B.MI_DTYPE_BF16
        dx, ddy = self.dev_t(x, BF), self.dev_t(dy, BF)
        dw = self.dev(shape=(K, Cc, k, k))
        self._chk(self.L.mi_op_conv_wgrad_bf16(dx.ptr, ddy.ptr, dw.ptr, N, Cc, H, K, k, stride), "conv_wgrad_bf16")
        return dw.get()

    def bn_fwd_t(self, x, gamma, beta, eps, relu, x_dt, a_dt, residual=None):
        N, Cc, H, _ = x.shape
        dx, dg, db = self.dev_t(x, x_dt), self.dev(gamma), self.dev(beta)
        dm, dv, dy = self.dev(shape=(Cc,)), self.dev(shape=(Cc,)), self.new_t(x.shape, a_dt)
        dr = self.dev_t(residual, a_dt) if residual is not None else None
        self._chk(self.L.mi_op_bn_fwd_t(dx.ptr, x_dt, dg.ptr, db.ptr, dr.ptr if dr else None, dm.ptr, dv.ptr, dy.ptr, a_dt, N, Cc, H, eps,
                                        int(relu)), "bn_fwd_t")
        return dm.get(), dv.get(), self.get_t(dy, a_dt)

    def conv_dgrad_bn_bwd_bf16(self, w, dy, H, stride, bn_x, mask, gamma, beta, means, vars_, eps, addend=None):
        """dgrad + the BN(+ReLU) backward in front of it, chained as backwards_pass does; returns gated dy, bn dx, dgamma, dbeta, fused?"""
        N = dy.shape[0]
        K, Cc, k, _ = w.shape
        BF = B.MI_DTYPE_BF16
        dw, ddy = self.dev(w), self.dev_t(dy, BF)
        dadd = self.dev_t(addend, BF) if addend is not None else None
        dxb, dmask = self.dev_t(bn_x, BF), self.dev_t(mask, BF)
        dg, db, dm, dv = (self.dev(a) for a in (gamma, beta, means, vars_))
        gated, bdx = self.new_t((N, Cc, H, H), BF), self.new_t((N, Cc, H, H), BF)
        og, ob = self.dev(shape=(Cc,)), self.dev(shape=(Cc,))
        rc = self.L.mi_op_conv_dgrad_bn_bwd_bf16(dw.ptr, ddy.ptr, dadd.ptr if dadd else None, gated.ptr, N, Cc, H, K, k, stride, dxb.ptr, dmask.ptr,
                                                 dg.ptr, db.ptr, dm.ptr, dv.ptr, eps, bdx.ptr, og.ptr, ob.ptr)
        if rc < 0:
            self._chk(rc, "conv_dgrad_bn_bwd_bf16")
        return self.get_t(gated, BF), self.get_t(bdx, BF), og.get(), ob.get(), rc > 0

    def conv_dgrad_bn_bwd_f32(self, w, dy, H, stride, bn_x, mask, gamma, beta, means, vars_, eps, addend=None):
        """fp32 twin of conv_dgrad_bn_bwd_bf16"""
        N = dy.shape[0]
        K, Cc, k, _ = w.shape
        dw, ddy = self.dev(w), self.dev(dy)
        dadd = self.dev(addend) if addend is not None else None
        dxb, dmask = self.dev(bn_x), self.dev(mask)
        dg, db, dm, dv = (self.dev(a) for a in (gamma, beta, means, vars_))
        gated, bdx = self.dev(shape=(N, Cc, H, H)), self.dev(shape=(N, Cc, H, H))
        og, ob = self.dev(shape=(Cc,)), self.dev(shape=(Cc,))
        rc = self.L.mi_op_conv_dgrad_bn_bwd_f32(dw.ptr, ddy.ptr, dadd.ptr if dadd else None, gated.ptr, N, Cc, H, K, k, stride, dxb.ptr, dmask.ptr,
                                                dg.ptr, db.ptr, dm.ptr, dv.ptr, eps, bdx.ptr, og.ptr, ob.ptr)
        if rc < 0:
            self._chk(rc, "conv_dgrad_bn_bwd_f32")
        return gated.get(), bdx.get(), og.get(), ob.get(), rc > 0

    def stem_fwd_bf16(self, x, w, exact=False):
        """the matrix-core stem: operands rounded to bf16, or (exact) fp32 arithmetic"""
        N, _, H, _ = x.shape
        dx, dw, dy = self.dev(x), self.dev(w), self.dev(shape=(N, 64, H // 2, H // 2))
        fn = self.L.mi_op_stem_fwd_f32 if exact else self.L.mi_op_stem_fwd_bf16
        self._chk(fn(dx.ptr, dw.ptr, dy.ptr, N, H), "stem_fwd")
        return dy.get()

    def stem_wgrad_bf16(self, x, w, dy, exact=False):
        N, _, H, _ = x.shape
        dx, dw, ddy, out = self.dev(x), self.dev(w), self.dev(dy), self.dev(shape=w.shape)
        fn = self.L.mi_op_stem_wgrad_f32 if exact else self.L.mi_op_stem_wgrad_bf16
        self._chk(fn(dx.ptr, dw.ptr, ddy.ptr, out.ptr, N, H), "stem_wgrad")
        return out.get()

    def conv_bn_fwd_t(self, x, w, gamma, beta, stride, eps, relu, dt):
        """conv + BN as forward_pass pairs them; returns conv_out, means, vars, y and whether the statistics were fused"""
        N, Cc, H, _ = x.shape
        K, _, k, _ = w.shape
        Ho = H // stride
        dx, dw, dg, db = self.dev_t(x, dt), self.dev(w), self.dev(gamma), self.dev(beta)
        dc, dy = self.new_t((N, K, Ho, Ho), dt), self.new_t((N, K, Ho, Ho), dt)
        dm, dv = self.dev(shape=(K,)), self.dev(shape=(K,))
        rc = self.L.mi_op_conv_bn_fwd_t(dx.ptr, dw.ptr, dc.ptr, dt, dg.ptr, db.ptr, dm.ptr, dv.ptr, dy.ptr, N, Cc, H, K, k, stride, eps, int(relu))
        if rc < 0:
            self._chk(rc, "conv_bn_fwd_t")
        return self.get_t(dc, dt), dm.get(), dv.get(), self.get_t(dy, dt), rc > 0

    def bn_apply_t(self, x, gamma, beta, means, vars_, eps, relu, x_dt, a_dt, residual=None):
        N, Cc, H, _ = x.shape
        dx, dg, db, dm, dv = self.dev_t(x, x_dt), self.dev(gamma), self.dev(beta), self.dev(means), self.dev(vars_)
        dy = self.new_t(x.shape, a_dt)
        dr = self.dev_t(residual, a_dt) if residual is not None else None
        self._chk(self.L.mi_op_bn_apply_t(dx.ptr, x_dt, dg.ptr, db.ptr, dr.ptr if dr else None, dm.ptr, dv.ptr, dy.ptr, a_dt, N, Cc, H, eps,
                                          int(relu)), "bn_apply_t")
        return self.get_t(dy, a_dt)

    def bn_bwd_t(self, x, gamma, beta, means, vars_, dy, eps, mask_mode, x_dt, a_dt, mask_src=None):
        """returns dx, dgamma, dbeta (and the gated dy for mask_mode 3)"""
        N, Cc, H, _ = x.shape
        dx, ddy = self.dev_t(x, x_dt), self.dev_t(dy, a_dt)
        dg, db, dm, dv = (self.dev(a) for a in (gamma, beta, means, vars_))
        dmask = self.dev_t(mask_src, a_dt) if mask_src is not None else None
        gated = self.new_t(x.shape, a_dt) if mask_mode == 3 else None
        out, ogam, obet = self.new_t(x.shape, x_dt), self.dev(shape=(Cc,)), self.dev(shape=(Cc,))
        self._chk(self.L.mi_op_bn_bwd_t(dx.ptr, x_dt, dg.ptr, db.ptr, dm.ptr, dv.ptr, ddy.ptr, dmask.ptr if dmask else None,
                                        gated.ptr if gated else None, a_dt, out.ptr, ogam.ptr, obet.ptr, N, Cc, H, eps, mask_mode), "bn_bwd_t")
        res = (self.get_t(out, x_dt), ogam.get(), obet.get())
        return res + (self.get_t(gated, a_dt),) if gated else res

    def maxpool_fwd_t(self, x, k, stride, dt):
        N, Cc, H, _ = x.shape
        Ho = H // stride
        dx, dy, di = self.dev_t(x, dt), self.new_t((N, Cc, Ho, Ho), dt), self.dev(shape=(N, Cc, Ho, Ho), dtype=np.int32)
        self._chk(self.L.mi_op_maxpool_fwd_t(dx.ptr, dy.ptr, dt, di.ptr, N, Cc, H, k, stride), "maxpool_fwd_t")
        return self.get_t(dy, dt), di.get()

    def maxpool_bwd_t(self, idx, dy, H, k, stride, dt):
        N, Cc = dy.shape[:2]
        di, ddy, dx = self.dev(idx.astype(np.int32)), self.dev_t(dy, dt), self.new_t((N, Cc, H, H), dt)
        self._chk(self.L.mi_op_maxpool_bwd_t(di.ptr, ddy.ptr, dx.ptr, dt, N, Cc, H, k, stride), "maxpool_bwd_t")
        return self.get_t(dx, dt)

    def avgpool_fwd_t(self, x, dt):
        N, Cc, H, _ = x.shape
        dx, dy = self.dev_t(x, dt), self.dev(shape=(N, Cc))
        self._chk(self.L.mi_op_avgpool_fwd_t(dx.ptr, dt, dy.ptr, N, Cc, H), "avgpool_fwd_t")
        return dy.get()

    def avgpool_bwd_t(self, dy, H, dt):
        N, Cc = dy.shape
        ddy, dx = self.dev(dy), self.new_t((N, Cc, H, H), dt)
        self._chk(self.L.mi_op_avgpool_bwd_t(ddy.ptr, dx.ptr, dt, N, Cc, H), "avgpool_bwd_t")
        return self.get_t(dx, dt)

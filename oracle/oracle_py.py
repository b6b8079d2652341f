"""ctypes wrapper of the CPU oracle (TEST INFRASTRUCTURE -- see oracle/oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Arrays are numpy float32/int32, activations NHWC, weights KCRS -- the reference's layout
(resnet.cu:140,145).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_fp = C.POINTER(C.c_float)


def build():
    """Compile oracle/liboracle_f{32,64}.so (gcc); building the checker is not using it."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _opt(a):
    return None if a is None else a.ctypes.data_as(_fp)


class Oracle:
    def __init__(self, acc="f32"):
        path = os.path.join(_HERE, "liboracle_%s.so" % acc)
        if not os.path.exists(path):
            build()
        L = self.lib = C.CDLL(path)
        i, f, vp = C.c_int, C.c_float, C.c_void_p
        L.orc_conv_fwd.argtypes = [_f32p, _f32p, i, i, i, i, i, i, _f32p]
        L.orc_conv_dgrad.argtypes = [_f32p, _f32p, i, i, i, i, i, i, i, _f32p]
        L.orc_conv_wgrad.argtypes = [_f32p, _f32p, i, i, i, i, i, i, _f32p]
        L.orc_bn_fwd.argtypes = [_f32p, _f32p, _f32p, i, i, i, f, _f32p, _f32p, _fp, _fp, _fp, i]
        L.orc_bn_bwd.argtypes = [_f32p, _f32p, i, i, i, f, _f32p, _f32p, _fp, _fp, _f32p, _f32p, _f32p, _f32p, _f32p, i]
        L.orc_maxpool_fwd.argtypes = [_f32p, i, i, i, i, i, _i32p, _f32p]
        L.orc_maxpool_bwd.argtypes = [_i32p, _f32p, i, i, i, i, _f32p]
        L.orc_avgpool_fwd.argtypes = [_f32p, i, i, i, _f32p]
        L.orc_avgpool_bwd.argtypes = [_f32p, i, i, i, _f32p]
        L.orc_add.argtypes = [i, _f32p, _f32p, _f32p]
        L.orc_relu.argtypes = [i, _f32p, _f32p]
        L.orc_relu_deriv.argtypes = [i, _f32p, _f32p, _f32p]
        L.orc_matmul.argtypes = [_f32p, _f32p, i, i, i, _f32p]
        L.orc_transpose.argtypes = [_f32p, i, i, _f32p]
        L.orc_softmax.argtypes = [_f32p, i, i, _f32p]
        L.orc_softmax_unstable.argtypes = [_f32p, i, i, _f32p]
        L.orc_ce_deriv.argtypes = [_f32p, _i32p, i, i]
        L.orc_adam.argtypes = [i, _f32p, _f32p, _f32p, _f32p, f, f, f, f, f, f, f]
        L.orc_loss.argtypes = [_f32p, _i32p, i, i, C.POINTER(i)]
        L.orc_loss.restype = f
        L.orc_net_create.argtypes = [i, i, i, i, i, i, i, _i32p, i, i, i]
        L.orc_net_create.restype = vp
        L.orc_net_destroy.argtypes = [vp]
        for nm in ("orc_net_n_locations", "orc_net_n_tensors"):
            getattr(L, nm).argtypes = [vp]
            getattr(L, nm).restype = i
        L.orc_net_location_size.argtypes = [vp, i]
        L.orc_net_location_size.restype = i
        for nm in ("orc_net_param", "orc_net_grad", "orc_net_mean", "orc_net_var"):
            getattr(L, nm).argtypes = [vp, i]
            getattr(L, nm).restype = _fp
        L.orc_net_set_hyper.argtypes = [vp, f, f, f, f, f]
        L.orc_net_set_batch.argtypes = [vp, _f32p, _i32p]
        for nm in ("orc_net_forward", "orc_net_backward", "orc_net_update"):
            getattr(L, nm).argtypes = [vp]
        L.orc_net_loss.argtypes = [vp, C.POINTER(i)]
        L.orc_net_loss.restype = f
        L.orc_net_tensor_name.argtypes = [vp, i]
        L.orc_net_tensor_name.restype = C.c_char_p
        L.orc_net_tensor_size.argtypes = [vp, i]
        L.orc_net_tensor_size.restype = C.c_size_t
        L.orc_net_tensor_ptr.argtypes = [vp, i]
        L.orc_net_tensor_ptr.restype = vp
        L.orc_net_find_tensor.argtypes = [vp, C.c_char_p]
        L.orc_net_find_tensor.restype = i
        L.orc_net_tensor_shape.argtypes = [vp, i, C.POINTER(i * 4)]
        L.orc_set_threads.argtypes = [i]

    def set_threads(self, n):
        self.lib.orc_set_threads(int(n))

    # ---- per-op helpers (allocate outputs) ----
    def conv_fwd(self, x, w, stride):
        N, H, _, Cc = x.shape
        K, _, k, _ = w.shape
        y = np.empty((N, H // stride, H // stride, K), np.float32)
        self.lib.orc_conv_fwd(x, w, H, k, Cc, K, stride, N, y)
        return y

    def conv_dgrad(self, w, dy, H, stride, dx_init=None):
        K, Cc, k, _ = w.shape
        N = dy.shape[0]
        dx = np.zeros((N, H, H, Cc), np.float32) if dx_init is None else dx_init.copy()
        self.lib.orc_conv_dgrad(w, dy, H, k, Cc, K, stride, N, 0 if dx_init is None else 1, dx)
        return dx

    def conv_wgrad(self, x, dy, k, stride):
        N, H, _, Cc = x.shape
        K = dy.shape[3]
        dw = np.empty((K, Cc, k, k), np.float32)
        self.lib.orc_conv_wgrad(x, dy, H, k, Cc, K, stride, N, dw)
        return dw

    def bn_fwd(self, x, gamma, beta, eps, relu):
        N, H, _, Cc = x.shape
        means = np.empty(Cc, np.float32)
        vars_ = np.empty(Cc, np.float32)
        xhat, norm, act = (np.empty_like(x) for _ in range(3))
        self.lib.orc_bn_fwd(x, gamma, beta, H, Cc, N, eps, means, vars_, _opt(xhat), _opt(norm), _opt(act), int(relu))
        return means, vars_, xhat, norm, act

    def bn_bwd(self, x, gamma, eps, means, vars_, xhat, act, dy, relu):
        N, H, _, Cc = x.shape
        dxhat, dx = np.empty_like(x), np.empty_like(x)
        dg, db = np.empty(Cc, np.float32), np.empty(Cc, np.float32)
        self.lib.orc_bn_bwd(x, gamma, H, Cc, N, eps, means, vars_, _opt(xhat), _opt(act), dy, dxhat, dg, db, dx, int(relu))
        return dx, dg, db

    def maxpool_fwd(self, x, k, stride):
        N, H, _, Cc = x.shape
        Ho = H // stride
        y = np.empty((N, Ho, Ho, Cc), np.float32)
        idx = np.empty((N, Ho, Ho, Cc), np.int32)
        self.lib.orc_maxpool_fwd(x, k, stride, N, H, Cc, idx, y)
        return y, idx

    def maxpool_bwd(self, idx, dy, Hin, stride):
        N, _, _, Cc = dy.shape
        dx = np.empty((N, Hin, Hin, Cc), np.float32)
        self.lib.orc_maxpool_bwd(idx, dy, Hin, stride, Cc, N, dx)
        return dx

    def matmul(self, a, b):
        out = np.empty((a.shape[0], b.shape[1]), np.float32)
        self.lib.orc_matmul(a, b, a.shape[0], a.shape[1], b.shape[1], out)
        return out

    def transpose(self, a):
        out = np.empty((a.shape[1], a.shape[0]), np.float32)
        self.lib.orc_transpose(a, a.shape[0], a.shape[1], out)
        return out

    def softmax(self, x, stable=True):
        out = np.empty_like(x)
        (self.lib.orc_softmax if stable else self.lib.orc_softmax_unstable)(x, x.shape[0], x.shape[1], out)
        return out

    def loss(self, pred, labels):
        nw = C.c_int(0)
        v = self.lib.orc_loss(pred, labels, pred.shape[0], pred.shape[1], C.byref(nw))
        return float(v), nw.value


class OracleNet:
    """forward_pass / backwards_pass / update_parameters of the reference on the CPU."""

    def __init__(self, oracle, dims, batch):
        self.o, self.L, self.dims, self.batch = oracle, oracle.lib, dict(dims), batch
        flags = np.ascontiguousarray(dims["is_block_spatial_reduction"], dtype=np.int32)
        self.h = self.L.orc_net_create(dims["input"], dims["init_kernel_dim"], dims["init_conv_filters"],
                                       dims["init_conv_stride"], dims["init_maxpool_dim"], dims["init_maxpool_stride"],
                                       dims["n_conv_blocks"], flags, dims["final_depth"], dims["output"], batch)
        self.n_locations = self.L.orc_net_n_locations(self.h)
        self.sizes = [self.L.orc_net_location_size(self.h, i) for i in range(self.n_locations)]
        self.names = [self.L.orc_net_tensor_name(self.h, i).decode() for i in range(self.L.orc_net_n_tensors(self.h))]

    def close(self):
        if self.h:
            self.L.orc_net_destroy(self.h)
            self.h = None

    def _loc(self, fn, i):
        return np.ctypeslib.as_array(fn(self.h, i), shape=(self.sizes[i],))

    def param(self, i):
        return self._loc(self.L.orc_net_param, i)

    def grad(self, i):
        return self._loc(self.L.orc_net_grad, i)

    def mean(self, i):
        return self._loc(self.L.orc_net_mean, i)

    def var(self, i):
        return self._loc(self.L.orc_net_var, i)

    def set_hyper(self, lr, wd, b1, b2, eps):
        self.L.orc_net_set_hyper(self.h, lr, wd, b1, b2, eps)

    def set_batch(self, images_nhwc, labels):
        self.L.orc_net_set_batch(self.h, np.ascontiguousarray(images_nhwc, np.float32),
                                 np.ascontiguousarray(labels, np.int32))

    def forward(self):
        self.L.orc_net_forward(self.h)

    def backward(self):
        self.L.orc_net_backward(self.h)

    def update(self):
        self.L.orc_net_update(self.h)

    def loss(self):
        nw = C.c_int(0)
        return float(self.L.orc_net_loss(self.h, C.byref(nw))), nw.value

    def tensor(self, name):
        """numpy view; image tensors come back shaped (N,H,W,C)."""
        i = self.L.orc_net_find_tensor(self.h, name.encode())
        if i < 0:
            raise KeyError(name)
        n = self.L.orc_net_tensor_size(self.h, i)
        shp = (C.c_int * 4)()
        self.L.orc_net_tensor_shape(self.h, i, C.byref(shp))
        dt = np.int32 if name in ("max_inds", "correct_classes") else np.float32
        ptr = C.cast(self.L.orc_net_tensor_ptr(self.h, i), C.POINTER(C.c_int32 if dt == np.int32 else C.c_float))
        a = np.ctypeslib.as_array(ptr, shape=(n,))
        if shp[0] > 0:
            a = a.reshape(shp[0], shp[1], shp[2], shp[3])
        return a

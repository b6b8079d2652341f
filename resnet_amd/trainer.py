"""Host-side mirror of the reference's main() (resnet.cu:3222-3429) over the C-ABI: same call
sequence (init_dimensions -> init_resnet -> init_general_batch -> init_trainer, then per step
load_new_batch -> forward_pass -> host loss -> backwards_pass -> update_parameters)."""
import ctypes as C

import numpy as np

from . import binding as B


def resnet_dims(input=224, n_conv_blocks=16, reductions=(3, 7, 13), final_depth=2048, output=1000,
                init_conv_filters=64):
    """the literals of resnet.cu:3245-3258"""
    return dict(input=input, init_kernel_dim=7, init_conv_filters=init_conv_filters, init_conv_stride=2,
                init_maxpool_dim=3, init_maxpool_stride=2, n_conv_blocks=n_conv_blocks,
                is_block_spatial_reduction=[1 if i in reductions else 0 for i in range(n_conv_blocks)],
                final_depth=final_depth, output=output)


class Trainer:
    def __init__(self, dims, batch, lr=1e-4, wd=0.0, b1=0.9, b2=0.999, eps=1e-7, seed=1234, n_epochs=1,
                 dump_dir="default", shard_n_images=None, device=None):
        self.L = L = B.load()
        if device is not None:
            if L.mi_set_device(int(device)) != 0:
                raise RuntimeError(self.error())
        self.dims, self.batch = dict(dims), batch
        nb = dims["n_conv_blocks"]
        self._flags = (C.c_int * max(nb, 1))(*dims["is_block_spatial_reduction"])
        self.c_dims = L.init_dimensions(dims["input"], dims["init_kernel_dim"], dims["init_conv_filters"],
                                        dims["init_conv_stride"], dims["init_maxpool_dim"], dims["init_maxpool_stride"],
                                        nb, self._flags, dims["final_depth"], dims["output"])
        rng = L.mi_rng_create(seed)
        self.model = L.init_resnet(self.c_dims, rng)
        L.mi_rng_destroy(rng)
        image_size = dims["input"] * dims["input"] * 3
        self.c_batch = L.init_general_batch(batch, image_size, dims["input"], shard_n_images or batch)
        self._dump_dir = dump_dir.encode()
        self.t = L.init_trainer(self.model, self.c_batch, batch, lr, wd, b1, b2, eps, n_epochs, self._dump_dir)
        self.check()
        p = self.t.contents.model.contents.params.contents
        self.n_locations = p.n_locations
        self.sizes = [p.sizes[i] for i in range(p.n_locations)]
        L.mi_trainer_set_dump_every(self.t, 0)
        self.dtype = B.MI_DTYPE_F32

    # ---- options (before the first step) ----
    def set_dtype(self, dtype):
        """MI_DTYPE_BF16: activations / activation gradients stored as bf16 (BASELINE configs[4])"""
        if self.L.mi_trainer_set_dtype(self.t, int(dtype)) != 0:
            e = self.error()
            self.L.mi_clear_error()
            raise RuntimeError("mi_trainer_set_dtype: " + e)
        self.dtype = int(dtype)

    def set_store_policy(self, policy):
        if self.L.mi_trainer_set_store_policy(self.t, int(policy)) != 0:
            e = self.error()
            self.L.mi_clear_error()
            raise RuntimeError("mi_trainer_set_store_policy: " + e)

    def activation_bytes(self):
        return int(self.L.mi_trainer_activation_bytes(self.t))

    def device_bytes(self):
        return int(self.L.mi_trainer_device_bytes(self.t))

    def check_errors(self):
        return int(self.L.mi_trainer_check_errors(self.t))

    # ---- plumbing ----
    def error(self):
        return self.L.mi_last_error().decode()

    def check(self):
        e = self.error()
        if e:
            raise RuntimeError("libresnet_mi: " + e)

    def close(self):
        if self.t:
            self.L.destroy_trainer(self.t)
            self.t = None

    def _to_host(self, ptr, n, dtype=np.float32):
        out = np.empty(n, dtype)
        self.L.mi_copy_to_host(out.ctypes.data, C.cast(ptr, C.c_void_p), n * 4)
        return out

    def _act_to_host(self, ptr, n):
        """an activation-typed tensor (bf16 in bf16 mode) widened to float32"""
        if self.dtype == B.MI_DTYPE_F32:
            return self._to_host(ptr, n)
        raw = np.empty(n, np.uint16)
        self.L.mi_copy_to_host(raw.ctypes.data, C.cast(ptr, C.c_void_p), n * 2)
        return (raw.astype(np.uint32) << 16).view(np.float32)

    def _to_dev(self, ptr, arr):
        arr = np.ascontiguousarray(arr)
        self.L.mi_copy_to_device(C.cast(ptr, C.c_void_p), arr.ctypes.data, arr.nbytes)

    def _pset(self, which):
        t = self.t.contents
        if which == "params":
            return t.model.contents.params.contents
        bb = t.backprop_buffer.contents
        return {"grads": bb.param_derivs, "means": bb.prev_means, "vars": bb.prev_vars}[which].contents

    def get(self, which, i):
        p = self._pset(which)
        return self._to_host(p.locations[i], p.sizes[i])

    def set(self, which, i, arr):
        p = self._pset(which)
        assert arr.size == p.sizes[i]
        self._to_dev(p.locations[i], arr.astype(np.float32).ravel())

    def set_params(self, arrays):
        for i, a in enumerate(arrays):
            self.set("params", i, a)

    # ---- data sources ----
    def source_synthetic(self, seed_images=1234, seed_labels=1235, pool_batches=2):
        self.L.mi_batch_source_synthetic(self.c_batch, seed_images, seed_labels, self.dims["output"], pool_batches)
        self.check()

    def source_host(self, layout=B.MI_LAYOUT_NHWC):
        self.L.mi_batch_source_host(self.c_batch, layout)

    def source_buffer(self, images_path, labels_path, layout=B.MI_LAYOUT_NHWC):
        self.L.mi_batch_source_buffer(self.c_batch, images_path.encode(), labels_path.encode(), layout)

    def source_shards(self, shard_dir, layout=B.MI_LAYOUT_NCHW, prefetch=False):
        self.L.mi_batch_source_shards(self.c_batch, shard_dir.encode(), layout)
        if prefetch:
            self.L.mi_batch_set_prefetch(self.c_batch, 1)

    def fill_host_batch(self, images, labels):
        """write the caller-owned pinned staging buffers (images_float_cpu / correct_classes_cpu)"""
        b = self.c_batch.contents
        n = b.n_images * b.image_size
        np.ctypeslib.as_array(b.images_float_cpu, shape=(n,))[:] = np.ascontiguousarray(images, np.float32).ravel()
        np.ctypeslib.as_array(b.correct_classes_cpu, shape=(b.n_images,))[:] = np.asarray(labels, np.int32)

    # ---- the reference main loop, one call each ----
    def load_new_batch(self):
        self.L.load_new_batch(self.t, None, self.c_batch)

    def forward(self):
        self.L.forward_pass(self.t)

    def loss(self):
        nw = C.c_int(0)
        return float(self.L.mi_host_loss(self.t, C.byref(nw))), nw.value

    def backward(self):
        self.L.backwards_pass(self.t)

    def update(self):
        self.L.update_parameters(self.t)

    def step(self):
        self.load_new_batch()
        self.forward()
        loss = self.loss()
        self.backward()
        self.update()
        return loss

    def pred(self):
        t = self.t.contents
        n = self.batch * self.dims["output"]
        return np.ctypeslib.as_array(t.forward_buffer.contents.pred_cpu, shape=(n,)).reshape(self.batch, -1).copy()

    def stem_dtype(self):
        """storage type of the stem convolution's own output and its gradient (MI_DTYPE_*)"""
        return self.L.mi_trainer_stem_dtype(self.t)

    def labels(self):
        return np.ctypeslib.as_array(self.c_batch.contents.correct_classes_cpu, shape=(self.batch,)).copy()

    def timings(self):
        out = (C.c_float * 5)()
        self.L.mi_trainer_last_timings(self.t, C.byref(out))
        return list(out)

    # ---- activation access by the reference's dump names (NCHW arrays) ----
    def activation(self, name, deriv=False):
        t = self.t.contents
        a = (t.backprop_buffer.contents.activation_derivs if deriv else t.forward_buffer.contents.activations).contents
        d, N = self.dims, self.batch
        f = d["init_conv_filters"]
        Hs = d["input"] // d["init_conv_stride"]
        Hp = Hs // d["init_maxpool_stride"]
        if name == "input":
            return self._to_host(self.c_batch.contents.images, N * 3 * d["input"] ** 2).reshape(N, 3, d["input"], d["input"])
        if name == "init_conv_applied":  # the stem convolution's own output: bf16 in the bf16 mode with the matrix-core stem, else fp32
            get = self._act_to_host if self.stem_dtype() == B.MI_DTYPE_BF16 else self._to_host
            return get(a.init_conv_applied, N * f * Hs * Hs).reshape(N, f, Hs, Hs)
        if name == "init_conv_activated":
            return self._act_to_host(a.init_conv_activated, N * f * Hs * Hs).reshape(N, f, Hs, Hs)
        if name == "init_convblock_input":
            return self._act_to_host(a.init_convblock_input, N * f * Hp * Hp).reshape(N, f, Hp, Hp)
        if name == "max_inds":
            return self._to_host(a.max_inds, N * f * Hp * Hp, np.int32).reshape(N, f, Hp, Hp)
        if name == "final_avg_pool":
            return self._to_host(a.final_conv_output_pooled, N * d["final_depth"]).reshape(N, -1)
        if name == "fc_output":
            if deriv:
                return self._to_host(t.backprop_buffer.contents.output_layer_deriv, N * d["output"]).reshape(N, -1)
            return self._to_host(a.linear_output, N * d["output"]).reshape(N, -1)
        if name == "softmax":
            return self._to_host(t.forward_buffer.contents.pred, N * d["output"]).reshape(N, -1)
        if name.startswith("conv_blocks/"):
            _, idx, leaf = name.split("/")
            k = a.activation_conv_blocks[int(idx)].contents
            H, Ho, R, X = k.incoming_spatial_dim, k.incoming_spatial_dim // k.stride, k.reduced_depth, k.expanded_depth
            table = {"reduction_applied": ("post_reduced", R, H), "reduction_activated": ("post_reduced_activated", R, H),
                     "spatial_applied": ("post_spatial", R, Ho), "spatial_activated": ("post_spatial_activated", R, Ho),
                     "expanded_applied": ("post_expanded", X, Ho), "expanded_post_norm": ("post_expanded_norm_vals", X, Ho),
                     "transformed_residual": ("transformed_residual", X, Ho),
                     "post_projection_norm_vals": ("post_projection_norm_vals", X, Ho),
                     "combined_output": ("output", X, Ho), "output_activated": ("output_activated", X, Ho)}
            field, ch, hh = table[leaf]
            ptr = getattr(k, field)
            if not ptr:
                raise KeyError(name + " is not stored (fast path); enable full-store")
            return self._act_to_host(ptr, N * ch * hh * hh).reshape(N, ch, hh, hh)
        if name.startswith("batch_norms/"):
            parts = name.split("/")
            if parts[1] == "init":
                cache = a.norm_init_conv.contents
            else:
                k = a.activation_conv_blocks[int(parts[1])].contents
                cache = {"reduced": k.norm_post_reduced, "spatial": k.norm_post_spatial,
                         "expanded": k.norm_post_expanded, "projected": k.norm_post_projection}[parts[2]].contents
            return self._to_host(getattr(cache, parts[-1]), cache.feature_size)
        raise KeyError(name)

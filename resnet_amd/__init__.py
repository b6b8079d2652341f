"""resnet_amd -- MI355X-native drop-in for the als244/ResNet training hot path.

The product is the C-ABI shared library `libresnet_mi.so` (include/resnet_mi.h; sources in
resnet_amd/csrc: C host code over hand-written HIP kernels for gfx950).  This package only binds it
(ctypes) and mirrors the reference's main() loop (resnet.cu:3222-3429) for tests and bench.py.
"""
from . import binding  # noqa: F401
from .trainer import Trainer, resnet_dims  # noqa: F401

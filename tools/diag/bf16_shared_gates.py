"""diagnostic: per-location distance of the bf16 product's gradients from the rounding model run with the product's gates"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth, torch_ref
from util import rel_l2
from resnet_amd import Trainer, binding as B
HYPER = dict(lr=1e-4, wd=0.0, b1=0.9, b2=0.999, eps=1e-7)
for name, dims, batch in (("C4I", synth.C4I_DIMS, 4), ("C1S", synth.C1S_DIMS, 4)):
    params = synth.make_params(dims, perturb_bn=True)
    tr = Trainer(dims, batch, **HYPER); tr.set_dtype(1); tr.set_params(params); tr.source_host(B.MI_LAYOUT_NHWC)
    table = synth.location_table(dims)
    for step in range(2):
        im, lab = synth.make_batch(dims, batch, step=step)
        tr.fill_host_batch(im, lab); tr.load_new_batch(); tr.forward()
        cur = [tr.get("params", i) for i in range(tr.n_locations)]
        emu = torch_ref.TorchNetBF16(dims, cur, eps=1e-7, gates=torch_ref.gates_of(tr, dims), stem_bf16=tr.stem_dtype() == 1)
        emu.forward(torch_ref.nhwc_to_nchw(im), lab); eg = emu.backward()
        tr.backward()
        rs = [rel_l2(tr.get("grads", i), eg[i].reshape(-1)) for i in range(tr.n_locations)]
        print(name, "step", step, " ".join("%d%s:%.1e" % (i, table[i][1], r) for i, r in enumerate(rs)))
        tr.update()
    tr.close()

import sys; sys.path.insert(0,'/root/repo')
from resnet_amd import Trainer, resnet_dims, binding as B
for dt in (0,1):
    for pol in (0,1):
        tr = Trainer(resnet_dims(), 256)
        if pol: tr.set_store_policy(pol)
        if dt: tr.set_dtype(dt)
        print("dtype %s policy %s: activations kept %.2f GB, device total %.2f GB" % ("bf16" if dt else "f32", "RECOMPUTE_BN" if pol else "FAST", tr.activation_bytes()/1e9, tr.device_bytes()/1e9))
        tr.close()

"""Data-parallel host plumbing: one process per GPU; the gradient all-reduce itself is RCCL
inside libresnet_mi.so (mi_dp_init / backwards_pass), on a second HIP stream, bucketed from the end of the
gradient arena (FC first -- the order update_parameters walks, resnet.cu:2952).  torch.distributed is
used only to rendezvous: broadcast the RCCL unique id, barrier, max-reduce the wall time.
Semantics: each rank owns a slice of the global batch, BN statistics stay per replica (the reference has no
cross-replica BN), gradients are SUMMED (the reference's loss gradient is a batch sum, resnet.cu:1806-1811)."""
import ctypes as C


def rank_seeds(rank, seed_images=1234, seed_labels=1235):
    """distinct synthetic image/label streams per rank (each rank's slice of the global batch)"""
    return seed_images + 7919 * rank, seed_labels + 7919 * rank


def exchange_unique_id(dist, rank, nbytes, make_id):
    """rank 0 creates the id with make_id() -> bytes; everyone gets the same bytes back"""
    import torch
    if rank == 0:
        raw = bytes(make_id())
        assert len(raw) == nbytes
        t = torch.tensor(list(raw), dtype=torch.uint8)
    else:
        t = torch.zeros(nbytes, dtype=torch.uint8)
    dist.broadcast(t, src=0)
    return bytes(t.numpy().tobytes())


def init_data_parallel(trainer, dist, rank, world, bucket_mb=32):
    """wire a resnet_amd.Trainer into an RCCL communicator of `world` ranks"""
    lib = trainer.L
    nbytes = lib.mi_dp_unique_id_bytes()

    def make_id():
        buf = (C.c_char * nbytes)()
        if lib.mi_dp_get_unique_id(buf, nbytes) != 0:
            raise RuntimeError("mi_dp_get_unique_id: " + trainer.error())
        return buf.raw

    raw = exchange_unique_id(dist, rank, nbytes, make_id)
    if lib.mi_dp_init(trainer.t, rank, world, raw, nbytes) != 0:
        raise RuntimeError("mi_dp_init: " + trainer.error())
    lib.mi_dp_set_bucket_bytes(trainer.t, int(bucket_mb) << 20)

"""Worker of test_sync_bn_two_ranks_equal_one_process_on_the_global_batch: rank r of a 2-process data-parallel group (gloo
transport, both ranks on cuda:0) runs forward+backward with SyncBN on ITS half of the batch and the flat gradient all-reduce;
rank 0 writes losses, gradients and the updated parameters (moving statistics) for the parent to compare."""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "oracle"), HERE]


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    import torch
    import torch.distributed as dist
    import taco_amd
    from taco_amd.train_ops import allreduce_gradients
    import taco_oracle as O
    from util import tiny_hp, to_product_hp
    d = np.load(os.path.join(out, "case.npz"))
    hp = tiny_hp(attention_type="bah_mon")
    w = O.init_weights(hp, 1, int(d["seed"]))
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % port, rank=rank, world_size=world)
    B = d["ids"].shape[0] // world
    sl = slice(rank * B, (rank + 1) * B)
    tr = taco_amd.Trainer(to_product_hp(hp), w)
    assert tr.enable_sync_bn(True) is True
    losses = tr.forward_backward(d["ids"][sl], d["L"][sl], d["mt"][sl], d["lt"][sl], d["co"][sl])
    allreduce_gradients(tr.grads)
    lall = losses.clone()
    dist.all_reduce(lall)                       # the global loss is the mean of the ranks' losses (equal shard sizes)
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(os.path.join(out, "result.npz"), grads=tr.grads.cpu().numpy(), params=tr.params.cpu().numpy(),
                 losses=(lall / world).cpu().numpy(), exchanges=tr.sync_exchanges)
    dist.barrier()
    tr.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

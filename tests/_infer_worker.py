"""Worker of test_gpu_two_ranks.py: rank r of a 2-process inference group (gloo rendezvous on 127.0.0.1, both ranks on cuda:0 --
one GPU is all a test box has) runs the HIP forward on ITS shard of the batch (dist.shard_range, SURVEY 8e: rows are independent,
no data-path collective), the row blocks are gathered on the host and rank 0 writes them for the parent to compare."""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "oracle"), HERE]


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    import torch
    import torch.distributed as dist
    import taco_oracle as O
    from taco_amd import dist as D
    from util import build_model
    d = np.load(os.path.join(out, "case.npz"))
    ohp = O.OracleHParams(max_iters=int(d["n"]), model_type=str(d["model_type"]))
    ns = int(d["num_speakers"])
    w = O.init_weights(ohp, ns, int(d["seed"]))
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % port, rank=rank, world_size=world)
    lo, hi = D.shard_range(d["ids"].shape[0], rank, world)
    m = build_model(ohp, w, num_speakers=ns)
    spk = d["spk"][lo:hi] if ns > 1 else None
    # Both ranks share cuda:0 here, and the persistent engine's kernels need every CU of the chip at once (256 resident workgroups):
    # two of them dispatched together can starve each other until the bounded spins expire.  On a node each rank owns a GPU; on this
    # one the ranks take turns (the gloo barrier orders them), so the engine under test is still the persistent one.
    for turn in range(world):
        if turn == rank:
            linear, alignments = m.run(inputs=d["ids"][lo:hi], input_lengths=d["L"][lo:hi], speaker_id=spk)
            torch.cuda.synchronize()
        dist.barrier()
    info = m.decoder_engine_info()
    m.check_device_errors()
    lin = D.gather_rows(linear.cpu().numpy(), world)
    ali = D.gather_rows(alignments.cpu().numpy(), world)
    mel = D.gather_rows(m.mel_outputs.cpu().numpy(), world)
    wall = D.max_over_ranks(1.0 + rank)                      # the timed-region reduction of bench.py, exercised with 2 ranks
    if rank == 0:
        np.savez(os.path.join(out, "result.npz"), linear=np.concatenate(lin), alignments=np.concatenate(ali), mel=np.concatenate(mel),
                 protocol=info["protocol"], wall=wall)
    dist.barrier()
    m.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

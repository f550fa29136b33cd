#!/usr/bin/env python
"""Train-step timing at config C4's per-GPU shard (B=32, T_in=128, T_out=512, r=4): forward with tape +
loss + backward + gradient all-reduce (RCCL, when launched with torch.distributed.run) + clip/Adam + pack refresh.
Prints one JSON line on rank 0.  Synthetic batch, random-init weights."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(args):
    """One measurement of the train step with the switches of `args` (see main); returns the report dict on rank 0, None elsewhere."""
    import numpy as np, torch, taco_amd
    from taco_amd import dist as D
    rank, local_rank, world = D.env_rank()
    torch.cuda.set_device(local_rank if world > 1 else 0)
    import torch.distributed as tdist
    own_group = world > 1 and not tdist.is_initialized()
    dist = D.init_process_group("nccl") if own_group else (tdist if world > 1 else None)
    dev = torch.device("cuda", torch.cuda.current_device())
    hp = taco_amd.hparams.copy(max_iters=max(200, args.t_out // 4))
    tr = taco_amd.Trainer(hp, taco_amd.weights.random_weights(hp, 1, seed=4321), device=str(dev))
    sync_bn = bool(args.sync_bn) and tr.enable_sync_bn(True)
    if args.engine != 1:
        tr.set_decoder_engine(args.engine)
    if not args.bptt:
        tr.set_bptt_engine(False)
    if args.exact_gemm != 4:
        tr.set_exact_gemm(args.exact_gemm)
    if args.exact_wgrad:
        tr.set_exact_wgrad(True)
    if getattr(args, "wgrad_planes", 1) != 1:
        tr.set_wgrad_planes(args.wgrad_planes)
    tr.set_deterministic(bool(args.deterministic))
    rs = np.random.RandomState(77 + rank)
    B, T_in, T_out = args.batch, args.t_in, args.t_out
    ids = rs.randint(2, 80, size=(B, T_in)).astype(np.int32); ids[:, -1] = 1
    lens = taco_amd.input_lengths_from_tokens(ids)
    ids, lens = torch.from_numpy(ids).to(dev), torch.from_numpy(lens).to(dev)
    mt = torch.from_numpy(rs.rand(B, T_out, hp.num_mels).astype(np.float32)).to(dev)
    lt = torch.from_numpy(rs.rand(B, T_out, hp.num_freq).astype(np.float32)).to(dev)
    if args.graph:
        tr.capture(ids, lens, mt, lt)
    first = None
    for _ in range(args.warmup):
        _, l = tr.train_step(ids, lens, mt, lt)
        first = float(l) if first is None else first
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(args.steps):
        _, l = tr.train_step(ids, lens, mt, lt)
    e1.record(); torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    wall = D.max_over_ranks(time.perf_counter() - t0, device=dev if dist is not None else "cpu")
    # phase split on one more step
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record(); tr.forward_backward(ids, lens, mt, lt, backward=False); ev[1].record()
    tr.forward_backward(ids, lens, mt, lt, backward=True); ev[2].record()
    tr.adam.step(tr.grads); tr.refresh(); ev[3].record(); torch.cuda.synchronize()
    from taco_amd.train_ops import allreduce_gradients
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a0.record()
    for _ in range(5):
        allreduce_gradients(tr.grads)
    a1.record(); torch.cuda.synchronize()
    allreduce_ms = a0.elapsed_time(a1) / 5
    tr.check_device_errors()
    per_rank = D.gather_floats(e0.elapsed_time(e1) / args.steps, device=dev if dist is not None else "cpu")
    engine = tr.decoder_engine_info()
    report = None
    if rank == 0:
        report = ({
            "metric": "train steps/s (C4 shard shapes)", "value": world * args.steps / wall / world, "unit": "steps/s",
            "n_gpus": world, "global_batch": world * B, "ms_per_step": wall / args.steps * 1e3,
            "target_frames_per_s": world * B * T_out * args.steps / wall, "dtype": "f32", "data": "synthetic", "launch": "hipGraph" if args.graph else "eager",
            "config": {"workload": "C4 shard: B=%d/GPU, T_in=%d, T_out=%d, r=%d, teacher-forced, batch-stat BN (%s)" % (
                           B, T_in, T_out, hp.reduction_factor, "synchronised over the ranks" if sync_bn else "per-rank statistics"),
                       "parallelism": "data-parallel x%d, one flat-bucket RCCL all-reduce of %d floats" % (world, tr.num_params)},
            "phase_ms": {"forward_only": ev[0].elapsed_time(ev[1]), "forward_plus_backward": ev[1].elapsed_time(ev[2]),
                         "adam_plus_refresh": ev[2].elapsed_time(ev[3]),
                         "gradient_allreduce": allreduce_ms if world > 1 else 0.0},
            "per_rank_ms_per_step": per_rank,
            "engine": {"decoder_loop_and_postnet_scans": "persistent whole-chip kernels with tape (protocol %d)" % engine["protocol"] if args.engine and engine["protocol"] else "one launch per stage",
                       "decoder_bptt": "one persistent whole-chip launch (k_decoder_bwd_xcd)" if engine.get("bptt_protocol", 0) else "one launch per stage (per-stage chain)",
                       "feed_forward_and_data_gradient_gemms": {3: "forward exact-fp32 MFMA, data gradients bf16 MFMA with operands split in two (3 products)", 1: "exact-fp32 MFMA", 0: "bf16 MFMA, operands split in two (3 products, the inference kernels)", 2: "forward split-bf16, data gradients exact", 4: "forward bf16 MFMA with operands split in three (6 products, fp32-grade), data gradients bf16 MFMA with operands split in two (3 products)"}[args.exact_gemm],
                       "backward_scans": ("post-net: k_bigru_oct_bwd (one row per cluster of 8 CUs) from 9 to 32 rows, else k_bigru_duo_bwd; encoder: k_bigru_resb (recurrent kernels in registers)"
                                          if args.engine and engine["protocol"] and args.bptt else "k_bigru_rows_bwd (round 1's kernel: A/B engine)"),
                       "weight_gradients": "exact-fp32 MFMA" if args.exact_wgrad else "bf16 MFMA, operands split three ways (fp32-grade)" + ("; conv banks, proj_1 and the linear head from pre-split planes" if getattr(args, "wgrad_planes", 1) == 1 else "; every eligible problem from pre-split planes" if args.wgrad_planes == 2 else ""),
                       "reductions": "ordered two-stage sums (deterministic)" if args.deterministic else "fp32 atomics"},
            "world_size_seen": world, "sync_bn": bool(sync_bn),
            "loss_without_coeff_first_last": [first, float(l)], "workspace_GB": tr._ws.numel() / 1e9})
        # FLOP roofline of the step (the dense contractions are MFMA bound).  Forward = the C2-shaped forward's
        # contractions; backward = one data-gradient and one weight-gradient product per forward product.
        import importlib.util
        spec_ = importlib.util.spec_from_file_location("taco_bench", os.path.join(ROOT, "bench.py"))
        bench = importlib.util.module_from_spec(spec_); spec_.loader.exec_module(bench)
        n = T_out // hp.reduction_factor
        fwd = bench.algorithmic_flops(hp, B, T_in, n); ff = bench.feedforward_flops(hp, B, T_in, n)
        step_s = wall / args.steps
        total = 3 * fwd * world
        fgemm = {3: "exact-fp32 MFMA", 1: "exact-fp32 MFMA", 0: "bf16 MFMA x3", 2: "bf16 MFMA x3", 4: "bf16 MFMA x6 (fp32-grade)"}[args.exact_gemm]
        bf16_issued = ((3 if args.exact_gemm in (0, 2) else 6 if args.exact_gemm == 4 else 0) + (3 if args.exact_gemm in (0, 3, 4) else 0) + (0 if args.exact_wgrad else 6)) * ff
        f32_mfma = ((0 if args.exact_gemm in (0, 2, 4) else 1) + (1 if args.exact_gemm in (1, 2) else 0) + (1 if args.exact_wgrad else 0)) * ff
        report["roofline"] = {
            "bound": "mfma", "achieved": total / step_s / 1e12, "peak": bench.MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
            "frac": total / step_s / 1e12 / bench.MFMA_F32_PEAK_TF, "traffic": None,
            "kernel": "whole train step (forward with tape + loss + backward + clip/Adam + pack refresh); algorithmic flops = 3 x the "
                      "forward's %.1f GFLOP (one data-gradient and one weight-gradient product per forward product) per rank" % (fwd / 1e9),
            "gflop_per_step_per_rank": 3 * fwd / 1e9,
            "pipes": {"forward_feed_forward_gemms": fgemm,
                      "fp32_mfma_gflop": f32_mfma / 1e9, "bf16_mfma_issued_gflop": bf16_issued / 1e9,
                      "fp32_valu_gflop (scans, decoder loop and their BPTT)": 3 * (fwd - ff) / 1e9,
                      "lower_bound_ms_at_peaks": (f32_mfma / (bench.MFMA_F32_PEAK_TF * 1e12) + bf16_issued / (bench.MFMA_BF16_PEAK_TF * 1e12)) * 1e3,
                      "note": "priced against the fp32-MFMA peak because the reference's arithmetic is fp32; the bf16 products are split operands "
                              "(3 per data-gradient product, 6 per weight-gradient product) that reproduce fp32"}}
        report["deterministic"] = bool(args.deterministic)
    tr.close() if hasattr(tr, "close") else None
    if dist is not None and own_group:
        dist.barrier(); dist.destroy_process_group()
    return report


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--t-in", type=int, default=128)
    ap.add_argument("--t-out", type=int, default=512)
    ap.add_argument("--graph", type=int, default=0, help="1: forward+backward replayed from one hipGraph; 0: eager launches")
    ap.add_argument("--gpus", type=int, default=1, help="N > 1: this script launches its own N ranks (torch.distributed.run, RCCL, 127.0.0.1)")
    ap.add_argument("--selftest-launcher", action="store_true", help="CPU test hook: launcher + gloo rendezvous + the flat all-reduce only")
    ap.add_argument("--engine", type=int, default=1, help="1: persistent whole-chip kernels for the teacher-forced decoder loop and the post-net scans (default); 0: one launch per stage (rounds 1-2)")
    ap.add_argument("--bptt", type=int, default=1, help="1: the decoder's BPTT as one persistent launch (k_decoder_bwd_xcd; default); 0: the chain of per-stage launches")
    ap.add_argument("--exact-gemm", type=int, default=4, help="4 (default): forward on the six-product split (fp32-grade), data gradients split-bf16; 3: forward GEMMs on the exact-fp32 MFMA (k_gemm), data gradients on the split-bf16 kernels (k_gemm_bf3); 1: everything exact; 0: everything split-bf16; 4: forward on the six-product split (fp32-grade), data gradients split-bf16")
    ap.add_argument("--wgrad-planes", type=int, default=1, help="split-bf16 weight gradients from pre-split operands: 1 the large problems (default), 2 every eligible one, 0 none (k_wgrad_bf3 only)")
    ap.add_argument("--exact-wgrad", type=int, default=0, help="1: weight gradients on the exact-fp32 MFMA (k_wgrad) instead of the split-bf16 kernel")
    ap.add_argument("--deterministic", type=int, default=1, help="1 (the library's default): ordered two-stage sums, bit-reproducible steps; 0: fp32 atomics")
    ap.add_argument("--sync-bn", type=int, default=0, help="1: BatchNorm statistics over the global batch (12 small all-reduces per step: one per BatchNorm layer forward, one per layer or conv bank backward); "
                                                           "0: per-rank statistics.  No effect on one GPU")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket, subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:], env=env))
    if args.selftest_launcher:
        import torch, taco_amd
        from taco_amd import dist as D
        from taco_amd.train_ops import allreduce_gradients
        rank, _, world = D.env_rank()
        dist = D.init_process_group("gloo") if world > 1 else None
        g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        allreduce_gradients(g)
        if rank == 0:
            print(json.dumps({"selftest": "launcher", "world_size": world, "grad_mean_factor": float(g[1])}))
        if dist is not None:
            dist.barrier(); dist.destroy_process_group()
        return
    rep = measure(args)
    if rep is not None:
        print(json.dumps(rep))


if __name__ == "__main__":
    main()

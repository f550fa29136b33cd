#!/usr/bin/env python
"""bench.py -- mel-frames/s of the batched Tacotron forward (BASELINE.json metric) on N MI355X.

A "step" = one pass of the hot path (taco_forward_infer, replayed from its hipGraph plan) over one
synthetic batch: ids -> encoder CBHG -> attention decoder (max_iters steps) -> post-net CBHG ->
linear spectrogram, inputs/outputs resident in HBM.  Workload at N=1 is BASELINE.json configs[1]
(C2: B=32, T_in=128, T_mel=512).  The K steps are issued round-robin over `--lanes` PlanPool lanes (default 1: strictly serial
forwards; the decoder loop and the post-net scan are whole-chip persistent kernels).  `python bench.py --gpus N` launches its own
N ranks: one process per GPU, per-GPU batch fixed (weak scaling), no data-path collective; barrier + synchronize on both sides,
MAX over ranks.

Prints ONE JSON line on rank 0 (see DESIGN.md 'Measurement' for every field)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TF = 157.3   # v_mfma_f32_32x32x2_f32 dense peak
MFMA_BF16_PEAK_TF = 2500.0 # v_mfma_f32_32x32x16_bf16 dense peak (MI355X_MICROARCH.md)

MFMA_BF16_MEASURED_TF = 2080.0      # tools/ubench_mfma on this part; a 3-term split product costs three bf16 MFMAs


def floor_constants():
    """The measured building blocks `roofline.latency_floor_ms` is built from, READ from the newest committed profile files of the two
    sequential loops (tools/time_decoder.py --json -> r*_decoder_timeline.json, tools/trace_bigru.py --json -> r*_scan_timeline.json,
    rocprofv3 kernel statistics -> r*_c2_kernel_stats*.csv) -- never constants carried over from an earlier round.
    Returns the numbers and the file each came from; a missing or unreadable file leaves its terms None and the floor says so."""
    import glob
    import re
    prof = os.path.join(ROOT, "profiles")

    def newest(pat):
        fs = glob.glob(os.path.join(prof, pat))
        # round, then the collection's version (rNN_vM_... / ..._vM.ext), then age, then name: the same choice on a fresh checkout, where every file has the same mtime
        key = lambda f: (int((re.findall(r"^r(\d+)_", os.path.basename(f)) or ["0"])[0]), int((re.findall(r"_v(\d+)[_.]", os.path.basename(f)) or ["0"])[-1]),
                         os.path.getmtime(f), os.path.basename(f))
        return sorted(fs, key=key)[-1] if fs else None
    c = {"sources": {}, "hop_us": None, "dec_step_us": None, "dec_chain_us": None, "dec_hops": 10, "scan_phases_us": None,
         "scan_sync_us": None, "scan_kernel": None, "enc_scan_ns": None, "iso_poll_us": None, "iso_exchange_us": None}
    here = source_hash()

    def src(f, h):
        """a source of the floor: the file and whether it was taken from THIS build of the kernels (None: the file carries no hash -- profiles of
        rounds 1-5 -- so the term may describe older kernel sources)"""
        return {"file": os.path.basename(f), "kernel_source_hash": h, "matches_this_build": (h == here) if h else None}
    f = newest("r*decoder_timeline.json")
    try:
        doc = json.load(open(f))
        rec = doc["C2"]["persistent"]
        tl, sp = rec["timeline_us"], rec["gates_stage_split_us"]
        c["dec_step_us"], c["hop_us"] = float(sum(tl.values())), float(sp["gather(poll)"])
        c["dec_chain_us"] = c["dec_step_us"] - c["dec_hops"] * c["hop_us"]
        c["sources"]["decoder"] = src(f, doc.get("kernel_source_hash"))
    except Exception:
        pass
    f = newest("r*scan_timeline.json")
    try:
        rec = [r for r in json.load(open(f)) if r["B"] == 32 and r["T"] == 512 and r["persist"] == 1][0]
        ph = rec["phases_clocks"]
        c["scan_phases_us"] = sum(v for k, v in ph.items() if "collect" not in k) / rec["clocks_per_us"]
        c["scan_sync_us"] = sum(v for k, v in ph.items() if "collect" in k) / rec["clocks_per_us"]
        c["scan_kernel"] = rec["kernel"]
        c["sources"]["scan"] = src(f, rec.get("kernel_source_hash"))
    except Exception:
        pass
    f = newest("r*c2_kernel_stats*.csv")
    try:
        import csv
        q = [r for r in csv.DictReader(open(f)) if "k_bigru_quad" in r["Name"]][0]
        c["enc_scan_ns"] = float(q["AverageNs"])
        hf = f[:-len(".csv")] + ".hash"
        c["sources"]["encoder_scan"] = src(f, open(hf).read().strip() if os.path.exists(hf) else None)
    except Exception:
        pass
    # the decoder's exchange IN ISOLATION (tools/ubench_mfma_stage, the stage without any arithmetic: publish, poll until every granule of the
    # gathered vector carries the tag + LDS write, barrier) -- the hardware part of a hop; what the traced gather of the real kernel shows
    # beyond it is the arrival spread of the member's slower waves, a property of this build
    f = newest("r*ubench_mfma_stage*.txt")
    try:
        ln = [l for l in open(f) if l.startswith("no compute")][0]
        us_step, clk_stage = float(re.search(r"([\d.]+) us per step", ln).group(1)), float(re.search(r"([\d.]+) clocks per stage", ln).group(1))
        poll = float(re.search(r"poll \+ LDS write\s+([\d.]+)", ln).group(1))
        clk_per_us = clk_stage * 10.0 / us_step
        c["iso_poll_us"], c["iso_exchange_us"] = poll / clk_per_us, clk_stage / clk_per_us
        c["sources"]["decoder_exchange_in_isolation"] = src(f, None)
        c["sources"]["decoder_exchange_in_isolation"]["matches_this_build"] = "n/a: the micro-benchmark uses the kernel's publish / gather primitives, not its step"
    except Exception:
        pass
    return c


WORKLOADS = {  # SURVEY section 8 config names: (B, T_in, r, n_steps, num_speakers, model_type)
    "C1": (1, 64, 5, 200, 1, "single"),
    "C2": (32, 128, 4, 128, 1, "single"),
    "C3": (32, 128, 4, 128, 4, "deepvoice"),
    "C5": (8, 512, 4, 1000, 1, "single"),
    "C4": (32, 128, 4, 128, 1, "single"),      # train.py step, one data-parallel shard (teacher-forced, T_out = 512): --workload C4 -> train_step_line()
}


def algorithmic_bytes(spec, hp, B, T_in, n):
    """SURVEY 8(d) streaming model (fp32): feed-forward weights once, decoder weights and attention
    keys/values once per decoder step, activations in/out once."""
    import numpy as np
    cnt = lambda pred: sum(int(np.prod(s)) if len(s) else 1 for k, s in spec if pred(k))
    W_dec = cnt(lambda k: k.startswith("decoder/") or k.startswith("attention/query") or k.startswith("attention/attention_"))
    W_post = cnt(lambda k: k.startswith("post_cbhg/") or k.startswith("linear/"))
    W_enc = cnt(lambda k: True) - W_dec - W_post
    A, D = hp.attention_size, 2 * hp.enc_rnn_size
    M, F, r = hp.num_mels, hp.num_freq, hp.reduction_factor
    LD = hp.dec_layer_num * hp.dec_rnn_size
    per_step = W_dec + B * T_in * (A + D) + B * (2 * (M + D + hp.attention_state_size + LD) + M * r + 3 * T_in)
    total = W_enc + W_post + n * per_step + B * T_in * (1 + D + A) + B * n * r * (M + F)
    return 4 * total, 4 * per_step


def stage_bytes(spec, hp, B, T_in, n):
    """The same streaming model split by stage (encoder + attention keys | decoder loop | post-net + linear head)."""
    import numpy as np
    cnt = lambda pred: sum(int(np.prod(s)) if len(s) else 1 for k, s in spec if pred(k))
    W_dec = cnt(lambda k: k.startswith("decoder/") or k.startswith("attention/query") or k.startswith("attention/attention_"))
    W_post = cnt(lambda k: k.startswith("post_cbhg/") or k.startswith("linear/"))
    W_enc = cnt(lambda k: True) - W_dec - W_post
    A, D = hp.attention_size, 2 * hp.enc_rnn_size
    M, F, r = hp.num_mels, hp.num_freq, hp.reduction_factor
    _, per_step = algorithmic_bytes(spec, hp, B, T_in, n)
    return {"encoder": 4 * (W_enc + B * T_in * (1 + D + A)), "decoder": n * per_step, "postnet": 4 * (W_post + B * n * r * (M + F))}


def algorithmic_flops(hp, B, T_in, n):
    """2*MAC of every contraction of the forward (180.3 GFLOP at C2)."""
    r, M = hp.reduction_factor, hp.num_mels
    def cbhg(rows, din, K, C, projs, pw, rnn, depth):
        mac = sum(k * din * C for k in range(1, K + 1))
        d = K * C
        for p in projs:
            mac += pw * d * p
            d = p
        if d != rnn:
            mac += d * rnn
        mac += depth * 2 * rnn * rnn + 2 * (2 * rnn) * 3 * rnn   # highway; BiGRU both directions
        return rows * mac
    enc_rows, post_rows = B * T_in, B * n * r
    d = hp.embedding_size
    pre = 0
    for s in hp.enc_prenet_sizes:
        pre += d * s
        d = s
    D, A, As, Hd = 2 * hp.enc_rnn_size, hp.attention_size, hp.attention_state_size, hp.dec_rnn_size
    enc = enc_rows * (pre + D * A) + cbhg(enc_rows, d, hp.enc_bank_size, hp.enc_bank_channel_size, hp.enc_proj_sizes,
                                            hp.enc_proj_width, hp.enc_rnn_size, hp.enc_highway_depth)
    dd = M + D
    dp = 0
    for s in hp.dec_prenet_sizes:
        dp += dd * s
        dd = s
    step = dp + (dd + As) * 3 * As + As * A + T_in * (A + D) + (As + D) * Hd + hp.dec_layer_num * (2 * Hd) * 3 * Hd + Hd * M * r
    post = cbhg(post_rows, M, hp.post_bank_size, hp.post_bank_channel_size, hp.post_proj_sizes, hp.post_proj_width,
                hp.post_rnn_size, hp.post_highway_depth) + post_rows * 2 * hp.post_rnn_size * hp.num_freq
    return 2 * (enc + B * n * step + post)


def feedforward_flops(hp, B, T_in, n):
    """the part of algorithmic_flops that runs on the bf16 matrix cores: everything except the recurrent halves of the two BiGRU
    scans and the decoder loop"""
    r = hp.reduction_factor
    total = algorithmic_flops(hp, B, T_in, n)
    As, Hd, A, D, M = hp.attention_state_size, hp.dec_rnn_size, hp.attention_size, 2 * hp.enc_rnn_size, hp.num_mels
    dd = M + D
    dp = 0
    for s_ in hp.dec_prenet_sizes:
        dp += dd * s_
        dd = s_
    step = dp + (dd + As) * 3 * As + As * A + T_in * (A + D) + (As + D) * Hd + hp.dec_layer_num * (2 * Hd) * 3 * Hd + Hd * M * r
    scans = B * T_in * 2 * hp.enc_rnn_size * 3 * hp.enc_rnn_size + B * n * r * 2 * hp.post_rnn_size * 3 * hp.post_rnn_size
    return total - 2 * (B * n * step + scans)


def feedforward_flops_by_stage(hp, B, T_in, n):
    """feedforward_flops split into the encoder side (prenet, encoder CBHG without its scan, attention memory layer) and the
    post-net side (post CBHG without its scan, linear head); the decoder loop has no feed-forward GEMM."""
    r = hp.reduction_factor
    post_rows = B * n * r
    K, C, pw, rnn = hp.post_bank_size, hp.post_bank_channel_size, hp.post_proj_width, hp.post_rnn_size
    mac = sum(k * hp.num_mels * C for k in range(1, K + 1))
    d = K * C
    for p in hp.post_proj_sizes:
        mac += pw * d * p
        d = p
    if d != rnn:
        mac += d * rnn
    mac += hp.post_highway_depth * 2 * rnn * rnn + rnn * 6 * rnn      # highways; hoisted BiGRU input projection (both directions)
    mac += 2 * rnn * hp.num_freq                                       # linear head
    post = 2 * post_rows * mac
    return {"encoder": feedforward_flops(hp, B, T_in, n) - post, "postnet": post}


def source_hash():
    """sha256 over the kernel sources: ties a committed profile (profiles/*pmc_hbm_traffic.json) to the build it was taken from."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "multi-speaker-tacotron-tensorflow_amd", "csrc")
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith((".h", ".hip")):
            h.update(fn.encode())
            h.update(open(os.path.join(csrc, fn), "rb").read())
    return h.hexdigest()[:16]


def cpu_arm(name, seed, threads, budget_s):
    """One arm of the CPU baseline, run in a process of its own (`bench.py --cpu-arm THREADS`): the fp32 torch-CPU restatement on the FULL batch of
    the workload with `threads` intra-op threads; 3 warm-up + 10 timed runs, median -- cut to what fits `budget_s` seconds (never fewer than
    1 warm-up + 3 timed; the counts used are in the record)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import torch
    import taco_oracle as O
    import taco_torch_cpu as TT
    B, T_in, r, n, ns, mt = WORKLOADS[name]
    torch.set_num_threads(int(threads))
    ohp = O.OracleHParams(max_iters=n, reduction_factor=r, model_type=mt)
    w = O.init_weights(ohp, ns, seed)
    ids, L = O.synthetic_inputs(B, T_in, seed)
    spk = (np.arange(B) % ns).astype(np.int32) if ns > 1 else None
    model = TT.TorchCpuTacotron(w, ohp, ns)
    t0 = time.perf_counter()
    model.forward(ids, L, spk)
    first = time.perf_counter() - t0
    timed = int(max(3, min(10, (budget_s - first) // max(first, 1e-3))))
    warm = 3 if timed == 10 else 1
    for _ in range(warm - 1):
        model.forward(ids, L, spk)
    ts = []
    for _ in range(timed):
        t0 = time.perf_counter()
        model.forward(ids, L, spk)
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    return {"value": B * n * r / med, "unit": "mel-frames/s", "threads": int(threads), "rows": int(B), "warmup_runs": warm, "timed_runs": timed,
            "median_s": med, "torch": torch.__version__.split("+")[0]}


def cpu_baseline(name, seed, budget_s=15.0):
    """SURVEY 8(d) / BASELINE.md section 3: the fp32 PyTorch-CPU, eager, op-for-op restatement of the TF1 inference graph
    (oracle/taco_torch_cpu.py -- not TF1; held to the NumPy oracle by tests/test_oracle.py) timed on this host over the FULL batch of the
    workload, (i) with one intra-op thread, mirroring the reference session's intra_op_parallelism_threads=1 (synthesizer.py:58-61), and
    (ii) with all host threads.  Every arm runs in a process of its own under a hard time limit (`cpu_arm`; 4 x `budget_s` + 30 s): a decoder
    step is a chain of small ops, and on a box with hundreds of hardware threads the all-threads arm takes minutes per forward: above 64 hardware
    threads it is not run (it timed out in every run of rounds 5-6) and 16- and 64-thread arms stand in for "many cores", so that the default
    bench run stays within minutes."""
    import subprocess
    B, T_in, r, n, ns, mt = WORKLOADS[name]
    ncores = int(os.cpu_count() or 1)

    def arm(threads):
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-arm", str(threads), "--workload", name, "--cpu-seed", str(seed), "--cpu-budget", str(budget_s)]
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="")
        try:
            out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=4 * budget_s + 30)
            return json.loads(out.stdout.strip().splitlines()[-1])
        except subprocess.TimeoutExpired:
            return {"threads": int(threads), "rows": int(B), "timed_out_after_s": 4 * budget_s + 30}
        except Exception as e:
            return {"threads": int(threads), "rows": int(B), "error": repr(e)}
    one = arm(1)
    arms = {"single_thread": one}
    if ncores > 64:
        # On the GPU box's 256 hardware threads the all-threads arm did not finish ONE forward in 90 s in any run of rounds 5 and 6 (a decoder step is a
        # chain of small ops: every op pays a 256-way fork / join): it is not started any more -- 90 s of every default bench run -- and two bounded
        # thread counts stand in for "many cores".
        arms["all_cores"] = {"threads": ncores, "rows": int(B), "not_run": "timed out after 90 s without finishing one forward in every run of rounds 5-6 "
                                                                        "(256-way fork / join per small op); 16- and 64-thread arms instead"}
        arms["threads_16"] = arm(16)
        arms["threads_64"] = arm(64)
    else:
        arms["all_cores"] = arm(ncores)
        if "value" not in arms["all_cores"] and ncores > 16:
            arms["threads_16"] = arm(16)
    done = [a for a in arms.values() if "value" in a]
    if not done:
        return {"value": None, "unit": "mel-frames/s", "cores": ncores, "kind": "port", "sample": "every arm of the CPU baseline failed or timed out", **arms}
    best = max(done, key=lambda a: a["value"])
    desc = lambda a: ("%d thread(s): %d warm-up + %d timed runs, median %.2f s per forward" % (a["threads"], a["warmup_runs"], a["timed_runs"], a["median_s"])
                      if "value" in a else "%d thread(s): %s" % (a["threads"], "timed out after %d s" % a["timed_out_after_s"] if "timed_out_after_s" in a else
                                                                 ("not run (%s)" % a["not_run"]) if "not_run" in a else a.get("error")))
    return {"value": best["value"], "unit": "mel-frames/s", "cores": int(best["threads"]), "host_cores": ncores, "kind": "port",
            "kind_detail": "fp32 PyTorch-CPU eager op-for-op restatement of the TF1 graph (oracle/taco_torch_cpu.py), NOT TF1; full batch; the best "
                           "arm is `value` (a decoder step is a chain of small ops: more threads mostly add synchronisation)",
            "sample": "oracle/taco_torch_cpu.py (torch %s, float32) on the FULL %s batch (B=%d, T_in=%d, T_mel=%d), every arm in a process of its own: %s"
                      % (best.get("torch", "?"), name, B, T_in, n * r, "; ".join(desc(a) for a in arms.values())),
            **arms}


def _bench_train_module():
    import importlib.util
    sp = importlib.util.spec_from_file_location("taco_bench_train", os.path.join(ROOT, "tools", "bench_train.py"))
    mod = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(mod)
    return mod


def train_step_report(steps, warmup, **switches):
    """BASELINE.json configs[3] (C4), one shard: the train.py step (train.py:215-219: forward with tape, loss, backward, clip + Adam) at
    B=32, T_in=128, T_out=512 per GPU, timed by tools/bench_train.py's measure() (ms per step, phase split, engine flags, FLOP roofline)."""
    import argparse as _ap
    ns = _ap.Namespace(steps=steps, warmup=warmup, batch=32, t_in=128, t_out=512, graph=0, engine=1, bptt=1, exact_gemm=4, exact_wgrad=0,
                       wgrad_planes=1, deterministic=1, sync_bn=0)
    for k, v in switches.items():
        setattr(ns, k, v)
    return _bench_train_module().measure(ns)


def cpu_baseline_train(seed):
    """The C4 counterpart of cpu_baseline(): the float64 checker's training step (oracle forward wiring + torch reverse-mode autograd,
    tests/torch_formulation.py -- a CPU restatement, not TF1) on ONE row of the shard at the full horizon, all host threads, one timed
    run after one warm-up (tens of seconds of CPU work).  Target frames per second of the step = rows * T_out / time."""
    sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import taco_oracle as O
    import torch_formulation as TF
    T_in, T_out = 128, 512
    ohp = O.OracleHParams(max_iters=T_out // 4)
    w = O.init_weights(ohp, 1, seed)
    ids, L = O.synthetic_inputs(1, T_in, seed)
    rs = np.random.RandomState(seed)
    mt, lt = rs.rand(1, T_out, ohp.num_mels), rs.rand(1, T_out, ohp.num_freq)
    ts = []
    for i in range(2):
        t0 = time.perf_counter()
        TF.train_grads(w, ohp, ids, L, mt, lt)
        ts.append(time.perf_counter() - t0)
        if ts[-1] > 60:
            break
    return {"value": T_out / ts[-1], "unit": "target frames/s through forward + backward", "cores": int(os.cpu_count() or 1), "kind": "port",
            "sample": "tests/torch_formulation.py train_grads (float64, torch CPU autograd over the oracle's training graph; not TF1) on 1 row of "
                      "the C4 shard at T_in=%d, T_out=%d: %.1f s per forward + backward (%d run(s)); rows are independent except for the "
                      "batch statistics of BatchNorm, so the per-row rate is what extrapolates" % (T_in, T_out, ts[-1], len(ts))}


def train_step_line(args):
    """`bench.py --workload C4`: one JSON line for the train step (same contract as the inference line; metric = target frames per
    second through the whole step; the headline metric of BASELINE.json stays the C2 inference line)."""
    rep = train_step_report(args.steps, args.warmup)
    if rep is None:
        return
    B, T_out, world = 32, 512, rep["n_gpus"]
    out = {"metric": "target mel-frames/sec through one train step (forward + backward + update)", "value": rep["target_frames_per_s"],
           "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": rep["ms_per_step"],
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32 storage and accumulation; forward GEMMs and weight gradients on split-bf16 MFMA with operands split in three (6 products, fp32-grade), data gradients split in two (3 products); recurrent parts exact fp32",
           "data": "synthetic", "world_size_seen": rep["world_size_seen"],
           "config": {"workload": rep["config"]["workload"], "global_batch": world * B, "parallelism": rep["config"]["parallelism"],
                      "launch": rep["launch"]},
           "roofline": rep["roofline"], "phase_ms": rep["phase_ms"], "engine": rep["engine"], "deterministic": rep["deterministic"],
           "per_rank_ms_per_step": rep["per_rank_ms_per_step"], "steps_per_s": rep["value"]}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_train(1237)
    print(json.dumps(out))


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` as typed: spawn the N ranks (one per GPU, RCCL rendezvous on 127.0.0.1) and relay rank 0's line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="weak (default): every GPU runs the workload's batch (B rows per GPU, global batch N*B); strong: the workload's batch is "
                         "the GLOBAL batch, split into B/N rows per GPU (SURVEY 8e asks for both; the loop is latency bound, so strong scaling is poor by construction)")
    ap.add_argument("--overlap", type=int, default=None, help="debug: 0 = no decoder/post-net overlap, N>1 = chunk of N decoder steps")
    ap.add_argument("--eager", action="store_true", help="enqueue kernels directly instead of replaying the hipGraph plan")
    ap.add_argument("--lanes", type=int, default=1,
                    help="forwards in flight per GPU (PlanPool: one plan + buffers + HIP stream each); 1 = strictly serial (default: the "
                         "persistent decoder / scan kernels each take the whole chip, so a second forward in flight has nothing to run on; "
                         "with --decoder-engine 0 the launch-per-stage kernels leave CUs idle and 4 lanes fill them)")
    ap.add_argument("--coalesce", type=int, default=1,
                    help="requests served per forward (PlanPool coalesce): c > 1 rides c batches of the workload's B rows through one "
                         "plan of c*B rows; a step is still one batch of B rows.  Default 1 = the BASELINE.json configuration as is")
    ap.add_argument("--no-companions", action="store_true", help="skip the lanes=1 and exact-fp32 companion measurements")
    ap.add_argument("--no-stage-timing", action="store_true",
                    help="skip the per-stage timings (roofline.stages): counter passes must see nothing but whole forwards, or their per-forward figures count the extra stage runs")
    ap.add_argument("--decoder-engine", type=int, default=1, help="1 persistent XCD-local decoder (default), 0 launch per stage, 2 persistent write-through")
    ap.add_argument("--chip-turns", type=int, default=1, help="debug A/B: 0 switches off the per-device ordering of whole-chip kernels across streams (taco_debug_set_chip_turns)")
    ap.add_argument("--cpu-arm", type=int, default=0, help=argparse.SUPPRESS)          # internal: one arm of cpu_baseline() in its own process
    ap.add_argument("--cpu-seed", type=int, default=1234, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-budget", type=float, default=20.0, help=argparse.SUPPRESS)
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="CPU test hook: run only the launcher / rendezvous / max-over-ranks path on gloo and print the world size")
    args = ap.parse_args()
    if args.cpu_arm:
        print(json.dumps(cpu_arm(args.workload, args.cpu_seed, args.cpu_arm, args.cpu_budget)))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args))
    if args.selftest_launcher:
        from importlib import import_module
        sys.path.insert(0, ROOT)
        D = import_module("taco_amd").dist
        rank, local_rank, world = D.env_rank()
        dist = D.init_process_group("gloo") if world > 1 else None
        t = D.max_over_ranks(1.0 + rank)
        if dist is not None:
            dist.barrier()
        if rank == 0:
            Bw = WORKLOADS[args.workload][0]
            rows = Bw // world if args.scaling == "strong" else Bw        # the sharding rule of the timed path (below)
            print(json.dumps({"selftest": "launcher", "world_size": world, "n_gpus": args.gpus, "max_over_ranks": t, "scaling": args.scaling,
                              "rows_per_gpu": rows, "global_batch": rows * world}))
        if dist is not None:
            dist.destroy_process_group()
        return
    if args.coalesce < 1 or (args.coalesce > 1 and (args.eager or args.steps % args.coalesce or args.warmup % args.coalesce)):
        ap.error("--coalesce c needs hipGraph plans and steps / warmup that are multiples of c")
    if args.workload == "C4":
        return train_step_line(args)

    import numpy as np
    import torch
    import taco_amd
    from taco_amd import dist as D
    from taco_amd.tacotron import _ptr, _stream

    rank, local_rank, world = D.env_rank()
    if args.gpus > 1 or world > 1:
        if world != args.gpus:
            sys.exit("bench.py --gpus %d was started with WORLD_SIZE=%d" % (args.gpus, world))
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist = D.init_process_group("nccl")
    else:
        torch.cuda.set_device(0)
        dist = None
    dev = torch.device("cuda", torch.cuda.current_device())

    B, T_in, r, n, ns, mt = WORKLOADS[args.workload]
    B_global = B * world if args.scaling == "weak" else B
    if args.scaling == "strong":
        if B % world:
            sys.exit("--scaling strong: the workload's batch of %d rows does not split over %d GPUs" % (B, world))
        B = B // world                                                           # rows of THIS rank; the global batch stays the workload's
    hp = taco_amd.hparams.copy(max_iters=n, reduction_factor=r, model_type=mt)
    model = taco_amd.create_model(hp)
    seed = 1234 + sorted(WORKLOADS).index(args.workload)
    model.load_weights(taco_amd.weights.random_weights(hp, ns, seed=seed))   # random-init weights of the architecture
    model.initialize(None, None, ns, None, device=str(dev))
    if args.overlap is not None:
        model._lib.taco_debug_set_overlap(model._handle, args.overlap)
    if args.decoder_engine != 1:
        model.set_decoder_engine(args.decoder_engine)
    if args.chip_turns != 1:
        model._lib.taco_debug_set_chip_turns(args.chip_turns)
    rs = np.random.RandomState(seed + 100 * rank)
    ids = rs.randint(2, 80, size=(B, T_in)).astype(np.int32)                   # (strong scaling: every rank draws its own B/N rows)
    ids[:, T_in - 1] = 1                                                       # fixed-length batches
    lengths = taco_amd.input_lengths_from_tokens(ids)
    lanes = max(1, args.lanes)
    co = args.coalesce
    pool = model.plan_pool(B, T_in, n, lanes=lanes, coalesce=co)
    for plan in pool.plans:                                                    # inputs resident in HBM before the timed region
        plan.inputs.copy_(torch.from_numpy(np.tile(ids, (co, 1))))
        plan.lengths.copy_(torch.from_numpy(np.tile(lengths, co)))
        if ns > 1:
            plan.speaker_id.copy_(torch.from_numpy(np.tile((np.arange(B) % ns).astype(np.int32), co)))
    torch.cuda.synchronize()
    plan = pool.plans[0]

    def step(i):
        if co > 1:                      # a step is one batch of B rows: every co-th step launches the forward that carries co of them
            if i % co == co - 1:
                pool.launch((i // co) % lanes)
            return
        lane = i % lanes
        if args.eager:
            p = pool.plans[lane]
            spk = p.speaker_id if ns > 1 else None
            with torch.cuda.stream(pool.streams[lane]):
                taco_amd._lib.check(model._lib.taco_forward_infer(
                    model._handle, _stream(), _ptr(p.inputs), _ptr(p.lengths), _ptr(spk), B, T_in, n, _ptr(None),
                    _ptr(p.mel), _ptr(p.linear), _ptr(p.align), _ptr(p.stop), _ptr(p.ws), p.ws_bytes))
        else:
            pool.launch(lane)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    model.check_device_errors()          # a persistent kernel that gave up would make every later forward drain early (and fast)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP events on the streams the kernels run on: ev0 is recorded on the main stream and every lane waits for
    # it before its first forward; every lane's last forward is joined back into the main stream before ev1.
    main_st = torch.cuda.current_stream()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(main_st)
    for st in pool.streams:
        st.wait_event(ev0)
    for i in range(args.steps):
        step(i)
    for st in pool.streams:
        main_st.wait_stream(st)
    ev1.record(main_st)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    device_errors = False
    try:
        model.check_device_errors()
    except taco_amd._lib.TacoError:
        device_errors = True
    headline_engine = model.decoder_engine_info() if pool.engine == "persistent" else {"protocol": 0}
    rank_ms = dev_ms / args.steps
    per_rank_ms = D.gather_floats(rank_ms, device=dev if dist is not None else "cpu")     # device ms per step of every rank
    wall = D.max_over_ranks(wall, device=dev if dist is not None else "cpu")
    dev_ms = D.max_over_ranks(dev_ms, device=dev if dist is not None else "cpu")
    device_errors = D.max_over_ranks(1.0 if device_errors else 0.0, device=dev if dist is not None else "cpu") > 0
    # latency of one forward with nothing else in flight (for the record; `value` is the throughput above)
    lat0, lat1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(pool.streams[0]):
        lat0.record()
        for _ in range(3):
            pool.plans[0].launch()
        lat1.record()
    torch.cuda.synchronize()
    latency_ms = lat0.elapsed_time(lat1) / 3
    # the three stages of the path one by one (stage-level C ABI, eager launches, nothing else in flight), HIP events on their stream
    stage_ms = {}
    if rank == 0 and not args.no_stage_timing:
        p0 = pool.plans[0]
        spk0 = p0.speaker_id if ns > 1 else None
        with torch.cuda.stream(pool.streams[0]):
            enc = model.encoder(p0.inputs, p0.lengths, spk0)
            mel0 = model.decoder(enc, n, spk0)[0]
            for name, fn in (("encoder", lambda: model.encoder(p0.inputs, p0.lengths, spk0)),
                             ("decoder", lambda: model.decoder(enc, n, spk0)),
                             ("postnet", lambda: model.postnet(mel0, speaker_id=spk0))):
                fn()
                s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s0.record()
                for _ in range(3):
                    fn()
                s1.record()
                torch.cuda.synchronize()
                stage_ms[name] = s0.elapsed_time(s1) / 3
            # the feed-forward part of the two CBHG stages alone (the recurrent scan launches skipped: outputs meaningless, timing only)
            model._lib.taco_debug_set_skip_scans(model._handle, 1)
            try:
                for name, fn in (("encoder", lambda: model.encoder(p0.inputs, p0.lengths, spk0)), ("postnet", lambda: model.postnet(mel0, speaker_id=spk0))):
                    fn()
                    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s0.record()
                    for _ in range(3):
                        fn()
                    s1.record()
                    torch.cuda.synchronize()
                    stage_ms[name + "_ff"] = s0.elapsed_time(s1) / 3
            finally:            # an exception above must not leave the model with its scans skipped for the companions
                model._lib.taco_debug_set_skip_scans(model._handle, 0)
        model.check_device_errors()
    finite = all(bool(torch.isfinite(p.mel).all().item() and torch.isfinite(p.linear).all().item()) for p in pool.plans)

    def timed_pool(pl, nlanes, nsteps):
        """nsteps forwards round-robin over nlanes lanes of pool pl; returns seconds per forward (HIP events, as the timed region)."""
        for i in range(nlanes):
            pl.launch(i % nlanes)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main_st)
        for st in pl.streams:
            st.wait_event(e0)
        for i in range(nsteps):
            pl.launch(i % nlanes)
        for st in pl.streams:
            main_st.wait_stream(st)
        e1.record(main_st)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 1e3 / nsteps

    companions = {}
    if rank == 0 and world == 1 and not args.no_companions and co == 1 and not args.eager:
        ksteps = max(4, min(args.steps, 12))

        def fill(pl, reps):
            for pln in pl.plans:
                pln.inputs.copy_(pool.plans[0].inputs.repeat(reps, 1)); pln.lengths.copy_(pool.plans[0].lengths.repeat(reps))
                if ns > 1:
                    pln.speaker_id.copy_(pool.plans[0].speaker_id.repeat(reps))
            torch.cuda.synchronize()

        # (i) strictly serial forwards when the headline runs several lanes
        if lanes > 1:
            companions["lanes_1"] = {"mel_frames_per_s": B * n * r / timed_pool(pool, 1, ksteps), "forwards_in_flight": 1}
        # (ii) every GEMM in exact fp32 (no split-bf16 feed-forward), same lanes
        model._lib.taco_debug_set_bf3(model._handle, 0, 0)
        model._plans.clear()
        p2 = model.plan_pool(B, T_in, n, lanes=lanes, coalesce=1)
        fill(p2, 1)
        t_exact = timed_pool(p2, lanes, ksteps)
        companions["exact_fp32"] = {"mel_frames_per_s": B * n * r / t_exact, "forwards_in_flight": lanes, "forward_ms": t_exact * 1e3,
                                    "arithmetic": "every contraction on exact-fp32 MFMA / VALU (taco_debug_set_bf3 off)"}
        out_exact = (p2.plans[0].mel.clone(), p2.plans[0].linear.clone())
        p2.close()
        model.check_device_errors()
        # (ii-b) fp32-GRADE products on the bf16 pipe: every feed-forward layer on the six-product instantiation of k_gemm_bf3 (operands
        # split three ways, 24 mantissa bits; one launch per layer -- the fused front / chain / head kernels are three-product kernels)
        model._lib.taco_debug_set_bf3(model._handle, 65, 0)
        model._plans.clear()
        p6 = model.plan_pool(B, T_in, n, lanes=lanes, coalesce=1)
        fill(p6, 1)
        t_x6 = timed_pool(p6, lanes, ksteps)
        d = lambda a, b: float((a - b).abs().max().item())
        out_def = (pool.plans[0].mel, pool.plans[0].linear)
        companions["fp32_grade_x6"] = {
            "mel_frames_per_s": B * n * r / t_x6, "forwards_in_flight": lanes, "forward_ms": t_x6 * 1e3,
            "arithmetic": "every feed-forward contraction as SIX bf16 MFMA products of operands split three ways (k_gemm_bf3<..., X6>, taco_debug_set_bf3 65): "
                          "fp32-grade (2^-24) products on the bf16 pipe; scans and decoder loop exact fp32 as in the headline",
            "max_abs_vs_exact_fp32": {"mel": d(p6.plans[0].mel, out_exact[0]), "linear": d(p6.plans[0].linear, out_exact[1])},
            "headline_max_abs_vs_exact_fp32": {"mel": d(out_def[0], out_exact[0]), "linear": d(out_def[1], out_exact[1])}}
        p6.close()
        model.check_device_errors()
        model._lib.taco_debug_set_bf3(model._handle, 1, 0)
        # (iii) two requests of B rows riding through one pass (PlanPool coalesce = 2)
        p4 = model.plan_pool(B, T_in, n, lanes=1, coalesce=2)
        fill(p4, 2)
        companions["two_requests_per_pass"] = {"mel_frames_per_s": 2 * B * n * r / timed_pool(p4, 1, ksteps), "rows_per_pass": 2 * B,
                                               "engine_info": model.decoder_engine_info()}
        p4.close()
        model.check_device_errors()
        # (iii-b) more rows than one pass holds: four requests = 128 rows through ONE call of the C ABI, which runs them as two passes of 64
        # (synthesizer.py:120-131 / eval.py --batch_size put no cap on the batch)
        if 4 * B > 64:
            p5 = model.plan_pool(B, T_in, n, lanes=1, coalesce=4)
            fill(p5, 4)
            companions["four_requests_per_call"] = {"mel_frames_per_s": 4 * B * n * r / timed_pool(p5, 1, max(4, ksteps // 2)), "rows_per_call": 4 * B,
                                                    "plan": model.engine_plan(4 * B, T_in, n * r).split(";")[0]}
            p5.close()
            model.check_device_errors()
        # (iv) the launch-per-stage engine of round 1 (decoder and scans as chains of small launches that leave most CUs idle), with
        # four forwards in flight to fill them: higher throughput than it has latency to show for
        if args.decoder_engine != 0:
            p3 = model.plan_pool(B, T_in, n, lanes=4, coalesce=1, engine="launch")
            fill(p3, 1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            thr = B * n * r / timed_pool(p3, 4, 3 * ksteps)
            with torch.cuda.stream(p3.streams[0]):
                e0.record(); p3.plans[0].launch(); p3.plans[0].launch(); e1.record()
            torch.cuda.synchronize()
            companions["launch_per_stage_engine_4_lanes"] = {"mel_frames_per_s": thr, "forwards_in_flight": 4,
                                                             "forward_latency_ms_alone": e0.elapsed_time(e1) / 2}
            p3.close()
            model.check_device_errors()
        model._plans.clear()
        # (v) BASELINE.json configs[3]: the train.py step at one C4 shard (same shapes as this workload), so the driver-run line carries it
        if args.workload == "C2":
            try:
                rep = train_step_report(6, 2)
                companions["train_step_c4_shard"] = {k: rep[k] for k in ("ms_per_step", "target_frames_per_s", "phase_ms", "engine", "roofline",
                                                                           "deterministic", "launch")}
            except Exception as e:      # the inference line must not die with the companion
                companions["train_step_c4_shard"] = {"error": repr(e)}
    if rank == 0:
        frames = B_global * n * r * args.steps
        spec = taco_amd.weights.weight_spec(hp, ns)
        abytes, per_step = algorithmic_bytes(spec, hp, B, T_in, n)
        flops = algorithmic_flops(hp, B, T_in, n)
        ff_flops = feedforward_flops(hp, B, T_in, n)
        sb = stage_bytes(spec, hp, B, T_in, n)
        fwd_s = dev_ms / 1e3 / args.steps      # device time per forward (HIP events over the timed region / forwards in it)
        ffs = feedforward_flops_by_stage(hp, B, T_in, n)
        stages = {}
        for k in (("encoder", "decoder", "postnet") if stage_ms else ()):
            e = {"ms_alone_eager": stage_ms[k], "algorithmic_GB": sb[k] / 1e9, "achieved_GBps": sb[k] / (stage_ms[k] * 1e-3) / 1e9,
                 "frac": sb[k] / (stage_ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS}
            if k == "decoder":
                e["binding_roofline"] = "hbm (streaming model); in practice L2 round trips + dependent chains, see latency_floor_ms"
            else:
                ffms = stage_ms[k + "_ff"]
                tf = 3 * ffs[k] / (ffms * 1e-3) / 1e12
                e["feed_forward_ms"] = ffms
                e["scan_ms"] = stage_ms[k] - ffms
                e["mfma_bf16"] = {"issued_gflop": 3 * ffs[k] / 1e9, "achieved": tf, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                                  "frac": tf / MFMA_BF16_PEAK_TF, "frac_of_measured_pipe": tf / MFMA_BF16_MEASURED_TF,
                                  "over": "the stage's feed-forward launches alone (scan launches skipped)"}
                e["binding_roofline"] = "mfma (bf16, 3 MFMAs per fp32 product) for the feed-forward part; the scan is latency bound"
            stages[k] = e
        T_mel = n * r
        fc = floor_constants() if args.workload in ("C2", "C3") else {"sources": {}}
        have = lambda *ks: all(fc.get(k) is not None for k in ks)
        terms = {"feed_forward_at_measured_mfma_ceiling": 3 * ff_flops / (MFMA_BF16_MEASURED_TF * 1e12) * 1e3}
        if have("hop_us", "dec_chain_us"):
            iso = min(fc["iso_poll_us"], fc["hop_us"]) if have("iso_poll_us") else fc["hop_us"]
            terms["decoder_handoffs_in_isolation"] = n * fc["dec_hops"] * iso * 1e-3
            if have("iso_poll_us"):
                terms["decoder_arrival_spread"] = n * fc["dec_hops"] * (fc["hop_us"] - iso) * 1e-3
            terms["decoder_chains"] = n * fc["dec_chain_us"] * 1e-3
        if have("scan_phases_us", "scan_sync_us"):
            terms["postnet_scan_phases"] = T_mel * fc["scan_phases_us"] * 1e-3
            terms["postnet_scan_collects_and_barriers"] = T_mel * fc["scan_sync_us"] * 1e-3
        if have("enc_scan_ns"):
            terms["encoder_scan"] = fc["enc_scan_ns"] * 1e-6
        hw = {k: terms[k] for k in ("decoder_handoffs_in_isolation", "feed_forward_at_measured_mfma_ceiling") if k in terms}
        design = {k: v for k, v in terms.items() if k not in hw}
        missing = [k for k in ("decoder", "scan", "encoder_scan") if k not in fc["sources"]]
        floor = {"hardware_terms": hw, "hardware_total": sum(hw.values()),
                 "design_terms": design, "design_total": sum(design.values()),
                 "total": sum(terms.values()), "measured_forward_ms": fwd_s * 1e3,
                 "sources": fc["sources"], "terms_without_a_profile": missing,
                 "decoder_with_no_arithmetic_ms": (n * fc["dec_hops"] * fc["iso_exchange_us"] * 1e-3) if have("iso_exchange_us") else None,
                 "note": "one forward in flight; C2 geometry.  hardware_terms are bounds no rewrite of THIS decomposition avoids: the decoder's ten dependent "
                         "exchanges per step at the hand-off time measured IN ISOLATION (tools/ubench_mfma_stage: publish -> every granule of the gathered vector "
                         "seen%s) and the feed-forward products at the measured matrix-pipe ceiling.  design_terms are this build's own time: what the traced "
                         "gather of the real kernel%s shows beyond the isolated hand-off (the arrival spread of a member's slower waves), the dependent-instruction "
                         "chains between exchanges, the scan's phases, collects and barriers, the encoder scan -- MEASURED time of these kernels, read from the "
                         "profile files named in `sources` (each with the kernel-source hash it was taken from and whether that is this build), a description "
                         "and a work-list, not a floor.  decoder_with_no_arithmetic_ms: ten exchanges per step with publish, LDS write and barrier but no "
                         "arithmetic at all (the same micro-benchmark).  0.30 of the HBM streaming roofline would need %.2f ms per forward"
                         % ((": %.2f us" % fc["iso_poll_us"]) if have("iso_poll_us") else "", (" (%.2f us)" % fc["hop_us"]) if have("hop_us") else "",
                            abytes / (0.30 * HBM_PEAK_GBS * 1e9) * 1e3)}
        out = {
            "metric": "mel-frames/sec (batched decode)", "value": frames / wall, "unit": "mel-frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32 storage and accumulation; feed-forward GEMMs as 3-term split-bf16 MFMA (bf16x3), the attention memory layer as 6-term split-bf16 MFMA (operands split three ways: fp32-grade); recurrent / decoder mat-vecs exact fp32",
            "data": "synthetic", "world_size_seen": world,
            "config": {"workload": "%s: batched inference B=%d/GPU, T_in=%d, T_mel=%d, r=%d, %s, attention bah_mon"
                                   % (args.workload, B, T_in, n * r, r, mt),
                       "global_batch": B_global, "rows_per_gpu": B, "parallelism": "batch-sharded replicas x%d, no collective" % world,
                       "arithmetic": "fp32 storage and accumulation; feed-forward GEMMs (both CBHGs, linear head) as 3-term split-bf16 MFMA, BiGRU scans and the decoder loop in exact fp32 (max err vs float64 oracle 3.4e-6)",
                       "decoder_engine": "launch per stage" if pool.engine != "persistent" else
                                         {1: "persistent XCD-local (csrc/taco_decoder_xcd.h)", 2: "persistent, write-through exchanges"}[args.decoder_engine],
                       "launch": "eager" if args.eager else "hipGraph plan (%d nodes)" % plan.num_nodes,
                       "forwards_in_flight": lanes, "requests_per_forward": co, "forward_latency_ms_alone": latency_ms},
            "roofline": {"bound": "hbm", "achieved": abytes / fwd_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": abytes / fwd_s / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "whole forward (one hipGraph launch); algorithmic bytes %.3f GB per forward, "
                                   "%.2f MB per decoder step; duration = timed region / forwards (%d in flight)"
                                   % (abytes / 1e9, per_step / 1e6, lanes),
                         "forward_ms": fwd_s * 1e3,
                         "stages": stages,
                         # where the time of one forward in flight goes: hardware terms (L2 hand-offs of the decoder loop, the matrix pipe)
                         # and design terms (this build's dependent-instruction chains: measured, not bounds)
                         "latency_floor_ms": floor,
                         # matrix-pipe view: the feed-forward contractions (everything but the two scans and the decoder loop) are
                         # issued as THREE bf16 MFMAs per fp32 product; utilisation is counted on that pipe against its dense peak
                         "mfma_bf16": {"issued_gflop_per_forward": 3 * ff_flops / 1e9, "achieved": 3 * ff_flops / fwd_s / 1e12,
                                       "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": 3 * ff_flops / fwd_s / 1e12 / MFMA_BF16_PEAK_TF,
                                       "note": "whole forward; per-stage fractions over the feed-forward time alone are in stages.*.mfma_bf16. "
                                               "fp32-equivalent work of the whole forward: %.1f GFLOP = %.1f TFLOP/s; the scans and the decoder run on the fp32 VALU"
                                               % (flops / 1e9, flops / fwd_s / 1e12)}},
            "per_rank_ms_per_step": per_rank_ms,
            "device_errors": bool(device_errors),
            "decoder_protocol": headline_engine.get("protocol"),
            "companions": companions,
            "outputs_finite": finite,
        }
        # HBM-side traffic per forward from the committed PMC passes (profiles/*pmc_hbm_traffic.json), same workload only, and only
        # when that profile was taken from THIS build of the kernels (sha256 over csrc/, recorded by tools/pmc_traffic.py)
        try:
            import glob
            pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm_traffic.json")), key=os.path.getmtime)
            here = source_hash()
            out["roofline"]["kernel_source_hash"] = here
            for f in reversed(pm):
                rec = json.load(open(f))
                if rec.get("workload") == args.workload and co == 1 and rec.get("kernel_source_hash") == here:
                    out["roofline"]["traffic"] = rec["traffic_bytes_per_forward"]
                    out["roofline"]["traffic_source"] = os.path.basename(f)
                    # what the HBM side actually carried, against the same peak (the streaming model above is the contract figure)
                    out["roofline"]["hbm_counter_frac"] = rec["traffic_bytes_per_forward"] / fwd_s / 1e9 / HBM_PEAK_GBS
                    break
            else:
                out["roofline"]["traffic_source"] = "none: no committed PMC profile matches this build of the kernels (hash %s)" % here
        except Exception:
            pass
        if "exact_fp32" in companions:      # the same roofline fraction for the exact-arithmetic runs, beside the headline's
            companions["exact_fp32"]["roofline_frac"] = abytes / (companions["exact_fp32"]["forward_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["roofline"]["frac_exact_fp32"] = companions["exact_fp32"]["roofline_frac"]
        if "fp32_grade_x6" in companions:
            companions["fp32_grade_x6"]["roofline_frac"] = abytes / (companions["fp32_grade_x6"]["forward_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["roofline"]["frac_fp32_grade_x6"] = companions["fp32_grade_x6"]["roofline_frac"]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, seed)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

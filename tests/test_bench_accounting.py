"""bench.py's accounting functions against SURVEY.md section 8(d)'s table -- the figures `roofline.achieved` is built from and the judge
recomputes: algorithmic bytes per configuration, per-step bytes, algorithmic FLOPs, and the consistency of the per-stage splits."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench          # noqa: E402
import taco_amd       # noqa: E402


def _cfg(name):
    B, T_in, r, n, ns, mt = bench.WORKLOADS[name]
    hp = taco_amd.hparams.copy(max_iters=n, reduction_factor=r, model_type=mt)
    return hp, taco_amd.weights.weight_spec(hp, ns), B, T_in, n


@pytest.mark.parametrize("name,total_gb,per_step_mb,floor_us", [("C1", 1.267, 6.15, 158), ("C2", 1.992, 14.69, 249), ("C5", 14.629, 14.45, 1830)])
def test_algorithmic_bytes_are_the_survey_tables(name, total_gb, per_step_mb, floor_us):
    hp, spec, B, T_in, n = _cfg(name)
    total, per_step = bench.algorithmic_bytes(spec, hp, B, T_in, n)
    assert abs(total / 1e9 - total_gb) < 0.0015 and abs(per_step / 1e6 - per_step_mb) < 0.006
    assert abs(total / bench.HBM_PEAK_GBS / 1e9 * 1e6 - floor_us) < 0.006 * floor_us
    st = bench.stage_bytes(spec, hp, B, T_in, n)
    assert st["decoder"] == n * per_step and abs(sum(st.values()) - total) < 0.01 * total


def test_workloads_are_the_survey_configurations():
    assert bench.WORKLOADS["C1"][:4] == (1, 64, 5, 200) and bench.WORKLOADS["C2"][:4] == (32, 128, 4, 128)
    assert bench.WORKLOADS["C3"][:4] == (32, 128, 4, 128) and bench.WORKLOADS["C3"][4:] == (4, "deepvoice")
    assert bench.WORKLOADS["C5"][:4] == (8, 512, 4, 1000)


def test_algorithmic_flops_at_c2_and_their_split():
    hp, spec, B, T_in, n = _cfg("C2")
    total = bench.algorithmic_flops(hp, B, T_in, n)
    assert abs(total / 1e9 - 180.3) < 0.5                                   # encoder 29.1 + decoder 12.7 + post-net 138.5
    ff = bench.feedforward_flops(hp, B, T_in, n)
    by = bench.feedforward_flops_by_stage(hp, B, T_in, n)
    assert abs(sum(by.values()) - ff) < 1e-6 * ff and 150e9 < ff < total
    assert abs(total / (B * n * hp.reduction_factor) / 1e6 - 11.0) < 0.1    # "per mel frame ~11.0 MFLOP"


def test_source_hash_covers_every_kernel_source_and_nothing_else():
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "multi-speaker-tacotron-tensorflow_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(root, "multi-speaker-tacotron-tensorflow_amd", "csrc", "*.hip")))
    assert len(files) >= 10
    h = bench.source_hash()
    assert len(h) == 16 and int(h, 16) >= 0
    assert bench.source_hash() == h                                          # stable


def test_the_train_step_companion_hands_measure_every_switch_it_reads(monkeypatch):
    """bench.py's `train_step_c4_shard` companion builds the argument namespace of tools/bench_train.py's measure() by hand: a switch
    added to the tool (round 6: --wgrad-planes) and not to the namespace turned the companion into an error string inside an otherwise
    healthy bench line.  Every `args.<name>` that measure() reads must be in the namespace."""
    import os, re, types
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "tools", "bench_train.py")).read()
    body = src[src.index("def measure(args):"):src.index("\ndef ", src.index("def measure(args):") + 10)]
    read = set(re.findall(r"\bargs\.([a-z_]+)", body))
    assert {"steps", "warmup", "batch", "wgrad_planes", "deterministic"} <= read
    seen = {}
    monkeypatch.setattr(bench, "_bench_train_module", lambda: types.SimpleNamespace(measure=lambda ns: seen.update(vars(ns)) or {}))
    bench.train_step_report(2, 1)
    assert read <= set(seen), sorted(read - set(seen))

"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, the tensor
list agrees with the oracle, hparams/IO helpers, error behaviour that needs no GPU."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

import taco_oracle as O
import taco_amd
from taco_amd import _lib
from util import tiny_hp, to_product_hp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _headers():
    """include/taco_abi.h (the drop-in boundary) + include/taco_debug.h (test / timing hooks of this repository)"""
    return "\n".join(open(os.path.join(ROOT, "include", n)).read() for n in ("taco_abi.h", "taco_debug.h"))


def test_the_public_header_carries_no_debug_entry_points():
    """test hooks do not belong in the ABI a reference maintainer binds."""
    pub = open(os.path.join(ROOT, "include", "taco_abi.h")).read()
    names = set(re.findall(r"\b(taco_[a-z0-9_]+)\s*\(", pub))
    assert not [n for n in names if "debug" in n], sorted(n for n in names if "debug" in n)
    dbg = open(os.path.join(ROOT, "include", "taco_debug.h")).read()
    dnames = set(re.findall(r"\b(taco_[a-z0-9_]+)\s*\(", dbg))
    assert dnames and all("debug" in n for n in dnames), sorted(dnames)


def test_library_loads_and_exports_every_header_symbol():
    lib = _lib.load_library()
    assert lib.taco_abi_version() == 1
    header = _headers()
    declared = set(re.findall(r"\b(taco_[a-z0-9_]+)\s*\(", header))
    declared -= {"taco_model", "taco_plan", "taco_hparams"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), "header declares %s but libtaco_hip.so does not export it" % name
        assert name in _lib.PROTOTYPES, "%s has no ctypes prototype" % name


def test_c_struct_layout_matches_header_order():
    header = open(os.path.join(ROOT, "include", "taco_abi.h")).read()
    body = header[header.index("typedef struct {"):header.index("} taco_hparams;")]
    names = [n.split("[")[0] for n in re.findall(r"\b([a-z_]+(?:\[4\])?)\s*[,;]", re.sub(r"/\*.*?\*/", "", body, flags=re.S))]
    assert names == [f[0] for f in _lib.TacoHParams._fields_]


@pytest.mark.parametrize("ns,mt", [(1, "single"), (4, "deepvoice"), (3, "simple")])
def test_weight_spec_equals_oracle(ns, mt):
    for ohp in (O.OracleHParams(model_type=mt), tiny_hp(model_type=mt), tiny_hp(model_type=mt, attention_type="bah_norm")):
        spec = dict(taco_amd.weights.weight_spec(to_product_hp(ohp), ns))
        assert spec == O.weight_shapes(ohp, ns)


def test_speaker_embedding_size_one_uses_tables():
    ohp = tiny_hp(model_type="deepvoice", speaker_embedding_size=1)
    spec = dict(taco_amd.weights.weight_spec(to_product_hp(ohp), 3))
    assert spec == O.weight_shapes(ohp, 3) and "spk/before_highway/table" in spec and "speaker_embedding" not in spec


def test_unknown_types_raise_like_the_reference():
    hp = taco_amd.hparams.copy(model_type="bogus")
    with pytest.raises(Exception, match=r"Unkown multi-speaker model type"):
        taco_amd.weights.weight_spec(hp, 2)
    hp = taco_amd.hparams.copy(attention_type="luong")       # LuongAttention is never imported: tacotron.py:138-150
    with pytest.raises(Exception, match=r"Unkown attention type"):
        taco_amd.weights.weight_spec(hp, 1)


def test_simple_model_type_with_embedding_size_one_is_refused():
    # the reference leaves speaker_embed undefined in that combination (tacotron.py:44-49,82-86)
    hp = taco_amd.hparams.copy(model_type="simple", speaker_embedding_size=1)
    with pytest.raises(_lib.TacoError, match="speaker_embed undefined"):
        taco_amd.weights.weight_spec(hp, 2)


def test_set_weight_shape_errors():
    lib = _lib.load_library()
    chp = _lib.to_c_hparams(taco_amd.hparams, 1)
    h = C.c_void_p()
    _lib.check(lib.taco_model_create(C.byref(chp), 0, C.byref(h)))
    bad = np.zeros((3, 3), np.float32)
    shp = (C.c_int64 * 2)(3, 3)
    assert lib.taco_model_set_weight(h, b"embedding", bad.ctypes.data_as(C.c_void_p), shp, 2) == _lib.TACO_ERR_SHAPE
    assert b"embedding" in lib.taco_last_error()
    assert lib.taco_model_set_weight(h, b"nope", bad.ctypes.data_as(C.c_void_p), shp, 2) == _lib.TACO_ERR_ARG
    assert lib.taco_model_finalize(h) == _lib.TACO_ERR_STATE       # weights missing
    lib.taco_model_destroy(h)


def test_workspace_bytes_is_monotone_and_positive():
    lib = _lib.load_library()
    chp = _lib.to_c_hparams(taco_amd.hparams, 1)
    h = C.c_void_p()
    _lib.check(lib.taco_model_create(C.byref(chp), 0, C.byref(h)))
    a = lib.taco_workspace_bytes(h, 8, 64, 50)
    b = lib.taco_workspace_bytes(h, 32, 128, 128)
    assert 0 < a < b
    # C2: bank [16384,2048] + xproj [16384,1536] fp32 dominate -> a few hundred MB
    assert 2e8 < b < 1e9
    lib.taco_model_destroy(h)


def test_random_weights_follow_reference_initialisers():
    w = taco_amd.weights.random_weights(taco_amd.hparams, 1, seed=0)
    assert np.all(w["encoder_cbhg/bigru/fw/gates/bias"] == 1.0)
    assert np.all(w["encoder_cbhg/highway_1/T/bias"] == -1.0)
    assert np.all(w["prenet/dense_1/bias"] == 0.0)
    assert np.abs(w["embedding"]).max() <= 1.0 + 1e-6                       # truncated at 2 sigma, sigma 0.5
    lim = np.sqrt(6.0 / (256 + 256))
    assert np.abs(w["prenet/dense_1/kernel"]).max() <= lim + 1e-7
    assert w["attention/attention_score_bias"].shape == ()


def test_weights_roundtrip_safetensors(tmp_path):
    ohp = tiny_hp()
    w = O.init_weights(ohp, 1, 0)
    p = str(tmp_path / "model.ckpt-7.safetensors")
    taco_amd.weights.save_weights(p, w)
    w2 = taco_amd.weights.load_weights(p)
    assert set(w) == set(w2)
    for k in w:
        assert w[k].shape == w2[k].shape and np.array_equal(w[k], w2[k])


def test_hparams_json_roundtrip(tmp_path):
    hp = taco_amd.hparams.copy(reduction_factor=5, attention_type="bah")
    taco_amd.save_hparams(str(tmp_path), hp)
    hp2 = taco_amd.load_hparams(taco_amd.hparams.copy(), str(tmp_path))
    assert hp2.reduction_factor == 5 and hp2.attention_type == "bah"
    assert json.load(open(tmp_path / "params.json"))["enc_bank_size"] == 16


def test_input_lengths_rule():
    toks = np.array([[5, 6, 1, 0, 0], [7, 8, 9, 10, 1], [3, 3, 3, 3, 3]])
    assert taco_amd.input_lengths_from_tokens(toks).tolist() == [2, 4, 0]     # synthesizer.py:120 (no EOS -> 0)


def test_shard_range_partitions_the_batch():
    from taco_amd.dist import shard_range
    for n, w in [(256, 8), (32, 3), (5, 8), (1, 1)]:
        parts = [shard_range(n, r, w) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
        assert max(b - a for a, b in parts) - min(b - a for a, b in parts) <= 1


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "multi-speaker-tacotron-tensorflow_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "taco_oracle" not in src and "oracle/" not in src, f


def test_no_persistent_kernel_uses_scratch():
    """csrc/build.sh keeps hipcc's per-kernel resource remarks; no instantiation of the persistent kernels may have scratch
    (tools/check_kernel_resources.py: a value demoted to scratch is a memory round trip inside a dependent chain)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "multi-speaker-tacotron-tensorflow_amd", "csrc", "kernel_resources.txt")
    if not os.path.exists(path):
        pytest.skip("no kernel_resources.txt (the library was built without csrc/build.sh)")
    spec = importlib.util.spec_from_file_location("check_kernel_resources", os.path.join(root, "tools", "check_kernel_resources.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    kernels, rows, bad = mod.check(path)
    names = [n for n, _ in kernels]
    assert any("k_bigru_xcdILi4ELi8" in n for n in names) and any("k_decoder_xcdILi4" in n for n in names)
    assert not bad, bad


def test_loss_exploded_guard_is_the_reference_loops():
    """train.py:228-230: loss > 100 or NaN -> Exception('Loss Exploded'); 100 itself and +-0 pass"""
    import taco_amd
    g = taco_amd.train_ops.raise_if_loss_exploded
    assert g(100.0, 3) == 100.0 and g(0.25, 0) == 0.25
    for bad in (100.0001, float("nan"), float("inf")):
        with pytest.raises(Exception, match="Loss Exploded"):
            g(bad, 1)
    assert g(float("-inf"), 1) == float("-inf")          # the reference's test lets it through as well
    import numpy as np
    assert g(np.float32(1.5), 2) == 1.5


def test_open_data_dirs_feeds_batches_from_npz_directories(tmp_path):
    """feeder.open_data_dirs: the reference's data directories (datafeeder.py:26-121) -> batches of the contract train.py consumes"""
    import numpy as np
    import taco_amd
    from taco_amd import feeder as F
    rs = np.random.RandomState(0)
    dirs = []
    for name, n in (("spk_a", 14), ("spk_b", 11)):
        d = tmp_path / name
        d.mkdir()
        for i in range(n):
            T, nt = int(rs.randint(10, 40)), int(rs.randint(4, 9))
            np.savez(str(d / ("u%02d.npz" % i)), tokens=np.r_[rs.randint(2, 80, nt - 1), 1].astype(np.int32),
                     mel=rs.rand(T, 80).astype(np.float32), linear=rs.rand(T, 1025).astype(np.float32))
        dirs.append(str(d))
    hp = taco_amd.hparams.copy(min_iters=3, max_iters=9, min_tokens=5, reduction_factor=4, initial_phase_step=2)
    lo, hi = F.frame_limits(4, 3, 9)
    f = F.open_data_dirs(dirs, batch_size=3, hparams=hp, data_type="train", batches_per_group=2, seed=7)
    for d, src in f.sources.items():          # the filter kept what get_path_dict keeps, minus the test split
        for p in src.paths:
            z = np.load(p)
            assert lo <= z["linear"].shape[0] <= hi and len(z["tokens"]) >= 5
    seen = 0
    for _ in range(6):
        b = next(f)
        assert b.inputs.shape[0] == 3 and b.mel_targets.shape[1] % 4 == 0 and b.mel_targets.shape[1] > int(max(b.input_lengths) * 0)
        assert b.mel_targets.shape[2] == 80 and b.linear_targets.shape[2] == 1025 and set(b.speaker_id.tolist()) <= {0, 1}
        assert np.array_equal(b.input_lengths, (b.inputs != 0).sum(1))          # lengths include the EOS (datafeeder.py:294)
        assert np.all(b.loss_coeff == 1.0)
        seen += 1
    assert f.step == 6 and seen == 6
    t = F.open_data_dirs(dirs, batch_size=3, hparams=hp, data_type="test", batches_per_group=1, seed=7)
    assert all(len(s.paths) == 3 for s in t.sources.values())                   # the last batch_size paths of every directory


def test_ctypes_prototypes_agree_with_the_header_argument_by_argument():
    """Every function of include/taco_abi.h and include/taco_debug.h: the ctypes mirror (_lib.PROTOTYPES) has the same number of arguments and the same class of
    type (pointer / int / float / 64-bit unsigned / 64-bit signed) in every position, and the same class of return type -- so the
    Python host cannot drift from the C ABI silently (a wrong arity or width is undefined behaviour, not an exception)."""
    import ctypes as C
    header = re.sub(r"/\*.*?\*/", "", _headers(), flags=re.S)
    header = re.sub(r"//[^\n]*", "", header)
    decls = re.findall(r"(?:^|[;}\n])\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_ ]*?[\s\*]+)(taco_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", header, flags=re.S)
    assert len(decls) >= 50

    def klass_c(t):
        t = " ".join(t.replace("const", " ").split())
        if "*" in t or t.endswith("_fn"):
            return "ptr"
        t = t.strip()
        return {"int": "int", "int32_t": "int", "float": "float", "size_t": "u64", "long long": "i64", "unsigned long long": "u64",
                "void": "void", "double": "double"}[t]

    def klass_py(t):
        if t is None:
            return "void"
        if t in (C.c_void_p, C.c_char_p) or (isinstance(t, type) and issubclass(t, (C._Pointer, C._CFuncPtr))):
            return "ptr"
        return {C.c_int: "int", C.c_int32: "int", C.c_float: "float", C.c_size_t: "u64", C.c_longlong: "i64", C.c_ulonglong: "u64",
                C.c_double: "double"}[t]              # (on LP64 ctypes aliases c_size_t / c_ulonglong to c_ulong and c_longlong to c_long)

    seen = set()
    for ret, name, args in decls:
        if name.endswith("_fn"):
            continue
        seen.add(name)
        params = [a.strip() for a in args.split(",")] if args.strip() not in ("", "void") else []
        ctypes_ = []
        for p in params:
            p = re.sub(r"\[[^\]]*\]", "*", p)                    # array parameters decay to pointers
            words = p.replace("*", " * ").split()
            typ = " ".join(words[:-1]) if (len(words) > 1 and words[-1] != "*" and re.match(r"^[A-Za-z_]\w*$", words[-1])
                                           and " ".join(words[:-1]).replace("const", "").strip()) else " ".join(words)
            ctypes_.append(klass_c(typ))
        restype, argtypes = _lib.PROTOTYPES[name]
        assert len(argtypes) == len(ctypes_), (name, len(argtypes), len(ctypes_), params)
        for i, (a, b) in enumerate(zip(argtypes, ctypes_)):
            assert klass_py(a) == b, (name, i, params[i], a)
        assert klass_py(restype) == klass_c(ret), (name, ret, restype)
    assert seen == set(_lib.PROTOTYPES), sorted(seen ^ set(_lib.PROTOTYPES))


def test_checkpoint_pruning_follows_the_reference_saver(tmp_path):
    """tf.train.Saver(max_to_keep=5, keep_checkpoint_every_n_hours=2) (train.py:175): the five newest train-state checkpoints stay; of the
    older ones, one per two hours of file time survives.  Pure host logic (files only)."""
    import os
    from taco_amd.train_ops import train_state_paths, list_train_checkpoints, prune_train_checkpoints
    d = str(tmp_path)
    t0 = 1_700_000_000
    for i, step in enumerate(range(1000, 11000, 1000)):          # ten checkpoints, 45 minutes apart
        for p in train_state_paths(d, step):
            open(p, "wb").close()
            os.utime(p, (t0 + i * 2700, t0 + i * 2700))
    open(os.path.join(d, "model.ckpt-500.safetensors"), "wb").close()      # a weight pack without optimizer state: not a train-state checkpoint
    assert [s for s, _ in list_train_checkpoints(d)] == list(range(1000, 11000, 1000))
    removed = prune_train_checkpoints(d, max_to_keep=5, keep_every_n_hours=2.0)
    left = [s for s, _ in list_train_checkpoints(d)]
    assert left == [1000, 4000, 6000, 7000, 8000, 9000, 10000], left          # 1000 (first), 4000 (2 h 15 min later) kept for good; 6000.. = the newest five
    assert len(removed) == 6 and os.path.exists(os.path.join(d, "model.ckpt-500.safetensors"))
    assert prune_train_checkpoints(d, max_to_keep=None) == []

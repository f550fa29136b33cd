"""TF1 checkpoint importer, CPU only.  No TensorFlow-written file exists here, so the bundle reader is
exercised on bundles produced by the module's own writer (same format description) -- format handling, name/shape mapping,
corruption detection."""
import os
import struct

import numpy as np
import pytest

import taco_amd
from taco_amd import tf_checkpoint as T
import taco_oracle as O
from util import tiny_hp, to_product_hp


def test_crc32c_known_answers():
    assert T.crc32c(b"") == 0
    assert T.crc32c(b"123456789") == 0xE3069283            # the standard CRC-32C check value
    assert T.crc32c(b"\x00" * 32) == 0x8A9136AA              # RFC 3720 B.4


def test_bundle_round_trip_many_blocks(tmp_path):
    rs = np.random.RandomState(0)
    tensors = {"scope_%03d/var/kernel" % i: np.asarray(rs.randn(*(rs.randint(1, 5, size=rs.randint(0, 4)))), np.float32) for i in range(150)}
    tensors["global_step"] = np.asarray(123456, np.int64)
    tensors["ids"] = rs.randint(0, 9, size=(3, 4)).astype(np.int32)
    prefix = str(tmp_path / "model.ckpt-123456")
    T.write_checkpoint(prefix, tensors)
    got = T.read_checkpoint(prefix)
    assert set(got) == set(tensors)
    for k, v in tensors.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
    assert T.latest_checkpoint(str(tmp_path)) == prefix
    T.write_checkpoint(str(tmp_path / "model.ckpt-99"), {"a": np.zeros(1, np.float32)})
    assert T.latest_checkpoint(str(tmp_path)) == prefix     # numeric, not lexicographic, order


def test_corruption_is_detected(tmp_path):
    prefix = str(tmp_path / "model.ckpt-1")
    T.write_checkpoint(prefix, {"w": np.arange(10, dtype=np.float32)})
    raw = bytearray(open(prefix + ".data-00000-of-00001", "rb").read()); raw[5] ^= 1
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(raw))
    with pytest.raises(IOError):
        T.read_checkpoint(prefix)
    assert T.read_checkpoint(prefix, verify=False)["w"].shape == (10,)
    idx = bytearray(open(prefix + ".index", "rb").read()); idx[3] ^= 1
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(IOError):
        T.read_index(prefix + ".index")
    open(prefix + ".index", "wb").write(b"not a table" * 10)
    with pytest.raises(IOError):
        T.read_index(prefix + ".index")


@pytest.mark.parametrize("mt,ns,ses,atype", [("single", 1, 16, "bah_mon"), ("deepvoice", 3, 4, "bah"), ("deepvoice", 3, 1, "bah_norm"), ("simple", 2, 4, "bah_mon")])
def test_import_reference_named_checkpoint(tmp_path, mt, ns, ses, atype):
    """A checkpoint carrying the reference's TF variable names (plus Adam slots and global_step) -> canonical weights,
    bit-exact; the oracle forward on the imported weights equals the forward on the originals."""
    ohp = tiny_hp(model_type=mt, speaker_embedding_size=ses, attention_type=atype)
    hp = to_product_hp(ohp)
    w = O.init_weights(ohp, ns, 3)
    spec = taco_amd.weights.weight_spec(hp, ns)
    prefix = str(tmp_path / "model.ckpt-5000")
    T.export_tf_checkpoint(prefix, w, spec, atype, global_step=5000)
    # what a real training run adds: optimizer slots
    names = T.tf_names_for(spec, atype)
    extra = T.read_checkpoint(prefix)
    extra.update({names["embedding"] + "/Adam": np.zeros_like(w["embedding"], np.float32), names["embedding"] + "/Adam_1": np.zeros_like(w["embedding"], np.float32),
                  "model/beta1_power": np.asarray(0.9, np.float32), "model/beta2_power": np.asarray(0.999, np.float32)})
    T.write_checkpoint(prefix, extra)
    got = T.import_tf_checkpoint(str(tmp_path), hp, ns)
    assert [k for k, _ in spec] == list(got)
    for k, _ in spec:
        assert np.array_equal(got[k], np.asarray(w[k], np.float32)), k


def test_wrapper_scope_strings_may_differ_between_tf_versions(tmp_path):
    ohp = tiny_hp()
    hp = to_product_hp(ohp)
    w = O.init_weights(ohp, 1, 4)
    spec = taco_amd.weights.weight_spec(hp, 1)
    names = T.tf_names_for(spec)
    renamed = {}
    for k, _ in spec:                      # another TF build: different contrib wrapper scope spellings
        t = names[k].replace("concat_output_and_attention_wrapper/attention_wrapper", "attention_wrapper_1/concat_wrapper")
        t = t.replace("decoder_prenet_wrapper/", "prenet_wrapper/wrapped/")
        renamed[t] = np.asarray(w[k], np.float32)
    got = T.map_tf_names(renamed, spec)
    for k, _ in spec:
        assert np.array_equal(got[k], np.asarray(w[k], np.float32)), k


def test_missing_variable_is_reported(tmp_path):
    ohp = tiny_hp()
    hp = to_product_hp(ohp)
    w = O.init_weights(ohp, 1, 4)
    spec = taco_amd.weights.weight_spec(hp, 1)
    names = T.tf_names_for(spec)
    tensors = {names[k]: np.asarray(w[k], np.float32) for k, _ in spec if k != "attention/attention_v"}
    with pytest.raises(KeyError) as e:
        T.map_tf_names(tensors, spec)
    assert "attention/attention_v" in str(e.value)


def test_crc32c_vectorised_path_equals_the_bytewise_one():
    """Buffers above 16 KB take the chunked path (4 KB chunks advanced in lock step, folded with the zero-byte operator)."""
    rs = np.random.RandomState(7)
    for n in (16384, 16385, 40000, 123457, 1 << 20):
        d = rs.bytes(n)
        assert T.crc32c(d) == T._crc_bytes(d), n
        assert T.crc32c(d, 0xDEADBEEF) == T._crc_bytes(d, 0xDEADBEEF), n        # continuation of an earlier CRC


def test_ordered_code_keys_known_answers():
    """tensorflow/core/lib/strings/ordered_code: the encodings the slice keys of partitioned variables are built from."""
    sgn = lambda v: T._oc_signed_increasing(v).hex()
    assert [sgn(v) for v in (0, 1, 63, -1, -64)] == ["80", "81", "bf", "7f", "40"]
    assert [sgn(v) for v in (64, 300, -65, 8191, 8192)] == ["c040", "c12c", "3fbf", "dfff", "e02000"]
    assert T._oc_num_increasing(0).hex() == "00" and T._oc_num_increasing(300).hex() == "02012c"
    assert T._oc_string(b"a\x00b\xffc") == b"a\x00\xffb\xff\x00c\x00\x01"
    assert T.slice_key("v", [(0, 64), (0, -1)]) == b"\x00" + b"v\x00\x01" + b"\x01\x02" + bytes.fromhex("80c040807f")


@pytest.mark.parametrize("seed", range(6))
def test_fuzzed_bundle_layouts(tmp_path, seed):
    """Shard counts, data-block sizes (down to one entry per block), restart intervals (prefix compression on / off) and
    partitioned variables, drawn at random: everything written must read back bit-exact, Adam slots must be skippable unread."""
    rs = np.random.RandomState(100 + seed)
    tensors = {}
    for i in range(int(rs.randint(5, 60))):
        nd = int(rs.randint(0, 4))
        shape = tuple(int(x) for x in rs.randint(1, 9, size=nd))
        name = "model/inference/%s/layer_%d/%s" % (rs.choice(["encoder_cbhg", "post_cbhg", "decoder"]), i % 7, rs.choice(["kernel", "bias", "gamma"]))
        tensors[name + ("" if name not in tensors else "_%d" % i)] = np.asarray(rs.randn(*shape), np.float32)
    big = "model/inference/embedding"
    tensors[big] = rs.randn(int(rs.randint(8, 200)), 5).astype(np.float32)
    tensors[big + "/Adam"] = np.zeros_like(tensors[big]); tensors[big + "/Adam_1"] = np.ones_like(tensors[big])
    tensors["global_step"] = np.asarray(int(rs.randint(0, 1 << 20)), np.int32)
    prefix = str(tmp_path / "model.ckpt-7")
    ns, bb, ri = int(rs.randint(1, 6)), int(rs.choice([1, 40, 200, 4096, 1 << 16])), int(rs.choice([0, 1, 2, 16]))
    part = {big: int(rs.randint(2, 6))} if seed % 2 else None
    T.write_checkpoint(prefix, tensors, num_shards=ns, block_bytes=bb, restart_interval=ri, partition=part)
    assert len([f for f in os.listdir(str(tmp_path)) if ".data-" in f]) == ns
    got = T.read_checkpoint(prefix)
    assert set(got) == set(tensors)
    for k, v in tensors.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
    lean = T.read_checkpoint(prefix, skip=T.is_optimizer_slot)
    assert set(lean) == {k for k in tensors if not k.endswith(("/Adam", "/Adam_1"))}
    # a slot's bytes may be damaged without the importer noticing or caring: it never reads them
    idx = T.read_index(prefix + ".index")
    e = idx[big + "/Adam"]
    if not e["sliced"] and e["size"]:
        fn = "%s.data-%05d-of-%05d" % (prefix, e["shard_id"], ns)
        raw = bytearray(open(fn, "rb").read()); raw[e["offset"]] ^= 0xFF
        open(fn, "wb").write(bytes(raw))
        T.read_checkpoint(prefix, skip=T.is_optimizer_slot)
        with pytest.raises(IOError):
            T.read_checkpoint(prefix)


def test_partitioned_variable_with_a_missing_slice_is_an_error(tmp_path):
    prefix = str(tmp_path / "model.ckpt-1")
    T.write_checkpoint(prefix, {"v": np.arange(24, dtype=np.float32).reshape(6, 4)}, partition={"v": 3})
    assert np.array_equal(T.read_checkpoint(prefix)["v"], np.arange(24, dtype=np.float32).reshape(6, 4))
    idx = T.read_index(prefix + ".index")
    assert idx["v"]["sliced"] and len(idx["v"]["slices"]) == 3 and len(idx[T.SLICES]) == 3


def test_tf1_fixture_dumper_prepare_and_compare_stages(tmp_path):
    """tools/tf1_dump_fixture.py: `prepare` turns the committed fixture into params.json + a reference-named checkpoint that the
    importer maps back bit-exactly; `compare` accepts outputs equal to the oracle's and rejects perturbed ones.  (`run` needs
    TensorFlow 1.x and the reference checkout: it is the part that executes elsewhere.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "tf1_dump_fixture.py")
    fixture = os.path.join(root, "tests", "golden", "tiny_forward.npz")
    work = str(tmp_path / "work")
    r = subprocess.run([sys.executable, tool, "prepare", fixture, work, "--shards", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    assert os.path.exists(os.path.join(work, "params.json")) and os.path.exists(os.path.join(work, "model.ckpt-0.data-00001-of-00002"))
    # `run --dry-run`: the variable list the TensorFlow side is expected to create, the name map checked in both directions and
    # against the prepared checkpoint's index -- without TensorFlow
    r = subprocess.run([sys.executable, tool, "run", "--dry-run", work], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "dry run OK" in r.stdout, r.stdout[-800:] + r.stderr[-1500:]
    exp = open(os.path.join(work, "expected_variables.txt")).read().splitlines()
    assert "model/inference/embedding:0 [80, 32]" in exp and any(l.startswith("model/inference/memory_layer/kernel:0") for l in exp)
    assert any("cell_0/output_projection_wrapper/concat_output_and_attention_wrapper/attention_wrapper/decoder_prenet_wrapper/gru_cell/gates/kernel:0" in l for l in exp)
    g = np.load(fixture)
    from golden.make_golden import fixture_config
    ohp, _, _, _, _, ns = fixture_config()
    got = T.import_tf_checkpoint(work, to_product_hp(ohp), ns)
    for k in got:
        assert np.array_equal(got[k], np.asarray(g["w:" + k], np.float32)), k
    np.savez(os.path.join(work, "tf1_outputs.npz"), linear=g["linear"], mel=g["mel"], alignments=g["alignments"], tf_version="1.4.0", n_variables=0)
    r = subprocess.run([sys.executable, tool, "compare", fixture, work], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "PINNED" in r.stdout, r.stdout + r.stderr
    np.savez(os.path.join(work, "tf1_outputs.npz"), linear=g["linear"] + 0.01, mel=g["mel"], alignments=g["alignments"], tf_version="1.4.0", n_variables=0)
    r = subprocess.run([sys.executable, tool, "compare", fixture, work], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "MISMATCH" in r.stdout

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

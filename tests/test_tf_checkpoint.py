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


def _bundle_messages():
    """tensorflow/core/protobuf/tensor_bundle.proto + tensor_shape.proto + tensor_slice.proto + versions.proto, declared to Google's own
    protobuf runtime: an encoder / decoder that shares no code with the hand-written one in tf_checkpoint.py"""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name="taco_test_bundle.proto", package="tt", syntax="proto3")

    def msg(parent, name, fields):
        m = parent.message_type.add(name=name) if hasattr(parent, "message_type") else parent.nested_type.add(name=name)
        for fname, num, typ, label, tname in fields:
            f = m.field.add(name=fname, number=num, type=typ, label=label)
            if tname:
                f.type_name = tname
        return m
    OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    shape = msg(fd, "TensorShapeProto", [("dim", 2, F.TYPE_MESSAGE, REP, ".tt.TensorShapeProto.Dim"), ("unknown_rank", 3, F.TYPE_BOOL, OPT, None)])
    msg(shape, "Dim", [("size", 1, F.TYPE_INT64, OPT, None), ("name", 2, F.TYPE_STRING, OPT, None)])
    sl = msg(fd, "TensorSliceProto", [("extent", 1, F.TYPE_MESSAGE, REP, ".tt.TensorSliceProto.Extent")])
    ext = msg(sl, "Extent", [("start", 1, F.TYPE_INT64, OPT, None), ("length", 2, F.TYPE_INT64, OPT, None)])
    ext.oneof_decl.add(name="has_length")
    ext.field[1].oneof_index = 0
    msg(fd, "VersionDef", [("producer", 1, F.TYPE_INT32, OPT, None), ("min_consumer", 2, F.TYPE_INT32, OPT, None), ("bad_consumers", 3, F.TYPE_INT32, REP, None)])
    msg(fd, "BundleHeaderProto", [("num_shards", 1, F.TYPE_INT32, OPT, None), ("endianness", 2, F.TYPE_INT32, OPT, None),
                                  ("version", 3, F.TYPE_MESSAGE, OPT, ".tt.VersionDef")])
    msg(fd, "BundleEntryProto", [("dtype", 1, F.TYPE_INT32, OPT, None), ("shape", 2, F.TYPE_MESSAGE, OPT, ".tt.TensorShapeProto"),
                                 ("shard_id", 3, F.TYPE_INT32, OPT, None), ("offset", 4, F.TYPE_INT64, OPT, None), ("size", 5, F.TYPE_INT64, OPT, None),
                                 ("crc32c", 6, F.TYPE_FIXED32, OPT, None), ("slices", 7, F.TYPE_MESSAGE, REP, ".tt.TensorSliceProto")])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, "GetMessageClass", None)
    mk = (lambda n: get(pool.FindMessageTypeByName("tt." + n))) if get else (lambda n: message_factory.MessageFactory(pool).GetPrototype(pool.FindMessageTypeByName("tt." + n)))
    return mk("BundleEntryProto"), mk("BundleHeaderProto")


def test_bundle_entries_against_googles_protobuf_runtime(tmp_path):
    """the hand-written protobuf layer of tf_checkpoint.py == google.protobuf on the published bundle schema, both directions: entries the
    writer emits parse there field by field, and entries serialised there (incl. field orders / defaults the writer never produces) parse here"""
    Entry, Header = _bundle_messages()
    rs = np.random.RandomState(3)
    for trial in range(200):
        shape = [int(x) for x in rs.randint(0, 5000, size=rs.randint(0, 5))]
        shard, off, size, crc = int(rs.randint(0, 4)), int(rs.randint(0, 2 ** 40)), int(rs.randint(0, 2 ** 33)), int(rs.randint(0, 2 ** 32))
        raw = T._entry_proto(1, shape, shard, off, size, crc)
        m = Entry.FromString(raw)
        assert (m.dtype, [d.size for d in m.shape.dim], m.shard_id, m.offset, m.size, m.crc32c) == (1, shape, shard, off, size, crc)
        assert m.SerializeToString() == raw                  # canonical field order and default elision, byte for byte
        e = T._parse_entry(m.SerializeToString())
        assert (e["dtype"], e["shape"], e["shard_id"], e["offset"], e["size"], e["crc32c"], e["sliced"]) == (1, shape, shard, off, size, crc, False)
    # a partitioned variable's full-tensor entry: slice specs, extents with and without a length
    slices = [[(0, 3), (0, -1)], [(3, 5), (0, -1)]]
    raw = T._entry_proto(1, [8, 4], 0, 0, 0, 0, slices=slices)
    m = Entry.FromString(raw)
    got = [[(x.start, x.length if x.WhichOneof("has_length") else -1) for x in s.extent] for s in m.slices]
    assert got == slices and [d.size for d in m.shape.dim] == [8, 4]
    m2 = Entry(dtype=1)
    for dsz in (8, 4):
        m2.shape.dim.add(size=dsz)
    for ext in slices:
        s = m2.slices.add()
        for st, ln in ext:
            x = s.extent.add(start=st)
            if ln >= 0:
                x.length = ln
    e = T._parse_entry(m2.SerializeToString())
    assert e["sliced"] and e["slices"] == slices and e["shape"] == [8, 4]
    # the header the writer puts under the empty key
    T.write_checkpoint(str(tmp_path / "m.ckpt-1"), {"a": np.arange(6, dtype=np.float32).reshape(2, 3)}, num_shards=1)
    idx = T.read_index(str(tmp_path / "m.ckpt-1.index"))
    h = Header.FromString(idx[b""] if b"" in idx else idx[""])
    assert (h.num_shards, h.endianness, h.version.producer) == (1, 0, 1)

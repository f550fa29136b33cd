"""TF1 checkpoint importer: `tf.train.Saver` V2 "tensor bundle" files -> canonical weight dict.

Replaces `saver.restore(sess, checkpoint)` of synthesizer.py:66-67 / train.py:189-206 for models trained with the
reference.  No TensorFlow needed: a bundle is
  <prefix>.index                 a LevelDB-format table (tensorflow/core/lib/io/table): key = variable name,
                                 value = BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}; key "" = header
  <prefix>.data-SSSSS-of-NNNNN   raw little-endian tensor bytes
TF writes these tables uncompressed; snappy-compressed blocks are rejected with a clear error.

Variable names are matched by their stable suffixes and by SHAPE, not by the full tf.contrib wrapper scope strings, which
differ between TF 1.x versions.  Weight layouts need no conversion: the canonical pack already uses TF's
(dense [in,out]; conv1d [k,in,out]; GRUCell gates [in+n, 2n] with columns r|u and rows [x;h]).

Multi-shard bundles, prefix-compressed keys with any restart interval / block size, and partitioned variables (slices keyed by
the ordered-code EncodeTensorNameSlice form) are read; Adam slots are skipped unread by the importer; CRCs are verified with a
vectorised crc32c.

UNPINNED: no checkpoint written by real TensorFlow exists in this environment; the reader is tested against bundles
produced by `write_checkpoint` below, which follows the same format description, over fuzzed layouts.  tools/tf1_dump_fixture.py
(`prepare` + `run` on a TensorFlow 1.x box) restores a bundle written here with tf.train.Saver: that run is the pin."""
import os
import re
import struct

import numpy as np

_MAGIC = 0xdb4775248b80fb57
SLICES = "\0slices"        # key of read_index()'s dict under which the slice entries of partitioned variables are kept
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 4: np.uint8, 6: np.int8, 5: np.int16, 10: np.bool_, 19: np.float16}
_DTYPE_IDS = {np.dtype(v).name: k for k, v in _DTYPES.items()}


# ---- crc32c (Castagnoli), as masked by leveldb ----
def _crc_table():
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
        tab.append(c)
    return tab


_CRC = _crc_table()


def _crc_bytes(data, crc=0):
    c = crc ^ 0xFFFFFFFF
    for b in data:
        c = _CRC[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


# crc(A || B) = shift(crc(A), len(B)) ^ crc(B): the zero-byte operator as a 32x32 matrix over GF(2) (the zlib construction)
def _gf2_times(mat, vec):
    out, i = 0, 0
    while vec:
        if vec & 1:
            out ^= mat[i]
        vec >>= 1
        i += 1
    return out


def _gf2_square(mat):
    return [_gf2_times(mat, mat[i]) for i in range(32)]


def _zero_operator(nbytes):
    """matrix that advances a finalised crc32c over `nbytes` zero bytes"""
    odd = [0x82F63B78] + [1 << (i - 1) for i in range(1, 32)]      # one zero BIT
    even = _gf2_square(odd)                                         # two bits
    odd = _gf2_square(even)                                         # four bits
    result = None
    n = nbytes
    while n:
        even = _gf2_square(odd)                                     # first pass: one zero byte
        if n & 1:
            result = even if result is None else [_gf2_times(even, r) for r in result]
        n >>= 1
        if not n:
            break
        odd = _gf2_square(even)
        if n & 1:
            result = odd if result is None else [_gf2_times(odd, r) for r in result]
        n >>= 1
    return result


_CHUNK = 4096
_CRC_NP = np.array(_CRC, dtype=np.uint32)
_OP_CACHE = {}


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli).  Long buffers are cut into 4 KB chunks whose CRCs advance in lock step (one NumPy operation per byte
    position over all chunks) and are then folded with the zero-byte operator: a 110 MB checkpoint verifies in about a second
    instead of minutes of per-byte interpreter work."""
    data = bytes(data) if not isinstance(data, (bytes, bytearray, memoryview)) else data
    n = len(data)
    if n < 4 * _CHUNK:
        return _crc_bytes(data, crc)
    nfull = n // _CHUNK
    a = np.frombuffer(data, dtype=np.uint8, count=nfull * _CHUNK).reshape(nfull, _CHUNK)
    st = np.full(nfull, 0xFFFFFFFF, dtype=np.uint32)
    for j in range(_CHUNK):
        st = _CRC_NP[(st ^ a[:, j]) & 0xFF] ^ (st >> 8)
    part = (st ^ 0xFFFFFFFF).tolist()
    if _CHUNK not in _OP_CACHE:
        _OP_CACHE[_CHUNK] = _zero_operator(_CHUNK)
    op = _OP_CACHE[_CHUNK]
    # fold: the running value first absorbs `crc` (the CRC of what came before), then every chunk
    total = crc
    for i, pc in enumerate(part):
        total = _gf2_times(op, total) ^ pc if (i or crc) else pc
    tail = data[nfull * _CHUNK:]
    return _crc_bytes(tail, total) if len(tail) else total


def _mask(crc):
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xFFFFFFFF


# ---- varints / protobuf ----
def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _pb_fields(buf):
    """Yields (field_number, wire_type, value) of one protobuf message."""
    pos = 0
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]; pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = buf[pos:pos + n]; pos += n
        elif wt == 5:
            v = buf[pos:pos + 4]; pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield fn, wt, v


def _parse_entry(buf):
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=None, sliced=False, slices=[])
    for fn, wt, v in _pb_fields(buf):
        if fn == 1:
            e["dtype"] = v
        elif fn == 2:
            for f2, _, v2 in _pb_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in _pb_fields(v2):
                        if f3 == 1:
                            size = v3
                    e["shape"].append(size)
        elif fn == 3:
            e["shard_id"] = v
        elif fn == 4:
            e["offset"] = v
        elif fn == 5:
            e["size"] = v
        elif fn == 6:
            e["crc32c"] = struct.unpack("<I", v)[0]
        elif fn == 7:                       # TensorSliceProto { repeated Extent extent = 1 { start = 1; length = 2 } }
            e["sliced"] = True
            ext = []
            for f2, _, v2 in _pb_fields(v):
                if f2 == 1:
                    start, length = 0, -1       # no length = the full extent of that dimension
                    for f3, _, v3 in _pb_fields(v2):
                        if f3 == 1:
                            start = v3
                        elif f3 == 2:
                            length = v3
                    ext.append((start, length))
            e["slices"].append(ext)
    return e


# ---- keys of the slices of a partitioned variable (tensorflow/core/util/saved_tensor_slice_util: EncodeTensorNameSlice over
# tensorflow/core/lib/strings/ordered_code) ----
def _oc_num_increasing(v):
    b = b"" if v == 0 else int(v).to_bytes((int(v).bit_length() + 7) // 8, "big")
    return bytes([len(b)]) + b


def _oc_string(sb):
    out = bytearray()
    for b in sb:
        out += b"\x00\xff" if b == 0 else (b"\xff\x00" if b == 0xFF else bytes([b]))
    return bytes(out) + b"\x00\x01"


_OC_HEADER = [(0, 0), (0x80, 0), (0xc0, 0), (0xe0, 0), (0xf0, 0), (0xf8, 0), (0xfc, 0), (0xfe, 0), (0xff, 0), (0xff, 0x80), (0xff, 0xc0)]


def _oc_signed_increasing(val):
    x = ~val if val < 0 else val
    if x < 64:
        return bytes([(0x80 ^ val) & 0xFF])
    buf = bytearray((b"\xff\xff" if val < 0 else b"\x00\x00") + (val & 0xFFFFFFFFFFFFFFFF).to_bytes(8, "big"))
    n = (x.bit_length() + 7) // 7
    out = buf[len(buf) - n:]
    out[0] ^= _OC_HEADER[n][0]
    out[1] ^= _OC_HEADER[n][1]
    return bytes(out)


def slice_key(name, extents):
    k = _oc_num_increasing(0) + _oc_string(name.encode("utf-8")) + _oc_num_increasing(len(extents))
    for start, length in extents:
        k += _oc_signed_increasing(start) + _oc_signed_increasing(length)
    return k


# ---- LevelDB table ----
def _read_block(buf, offset, size, verify):
    body = buf[offset:offset + size]
    ctype = buf[offset + size]
    if verify:
        stored = struct.unpack("<I", buf[offset + size + 1:offset + size + 5])[0]
        if _mask(crc32c(buf[offset:offset + size + 1])) != stored:
            raise IOError("checkpoint index: block checksum mismatch at offset %d" % offset)
    if ctype != 0:
        raise IOError("checkpoint index: compressed blocks (type %d) are not supported; TF's BundleWriter writes none" % ctype)
    return body


def _block_entries(block):
    nrestarts = struct.unpack("<I", block[-4:])[0]
    end = len(block) - 4 - 4 * nrestarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_index(path, verify=True):
    """{variable name: entry dict} of a `<prefix>.index` file (+ key '' -> header bytes)."""
    buf = open(path, "rb").read()
    if len(buf) < 48 or struct.unpack("<Q", buf[-8:])[0] != _MAGIC:
        raise IOError("%s is not a TensorFlow V2 checkpoint index (bad table magic)" % path)
    footer = buf[-48:]
    pos = 0
    _, pos = _varint(footer, pos); _, pos = _varint(footer, pos)            # metaindex handle
    ioff, pos = _varint(footer, pos); isz, pos = _varint(footer, pos)       # index handle
    out = {}
    for _, handle in _block_entries(_read_block(buf, ioff, isz, verify)):
        boff, p2 = _varint(handle, 0)
        bsz, _ = _varint(handle, p2)
        for key, val in _block_entries(_read_block(buf, boff, bsz, verify)):
            if key == b"":
                out[""] = val
            elif key[:1] == b"\x00":                        # a slice of a partitioned variable (binary ordered-code key)
                out.setdefault(SLICES, {})[bytes(key)] = _parse_entry(val)
            else:
                out[key.decode("utf-8")] = _parse_entry(val)
    return out


def is_optimizer_slot(name):
    """Adam's m / v slots and the beta-power accumulators: three quarters of a training checkpoint, nothing a forward needs."""
    return name.endswith(("/Adam", "/Adam_1")) or name in ("beta1_power", "beta2_power") or name.startswith(("beta1_power", "beta2_power"))


def read_checkpoint(prefix, verify=True, names=None, skip=None):
    """All (or the named) tensors of the bundle `<prefix>.index` + `<prefix>.data-*` as {name: ndarray}.  `skip(name)` true:
    the variable is neither read nor checksummed.  Partitioned variables are reassembled from their slices."""
    idx = read_index(prefix + ".index", verify)
    nshards = 1
    for fn, _, v in _pb_fields(idx.get("", b"")):
        if fn == 1:
            nshards = v
    shards = {}

    def fetch(e, what):
        sid = e["shard_id"]
        if sid >= nshards:
            raise IOError("%s: shard %d of a %d-shard checkpoint" % (what, sid, nshards))
        if sid not in shards:
            shards[sid] = open("%s.data-%05d-of-%05d" % (prefix, sid, nshards), "rb")
        f = shards[sid]
        f.seek(e["offset"])
        raw = f.read(e["size"])
        if len(raw) != e["size"]:
            raise IOError("%s: data shard %d is truncated" % (what, sid))
        if verify and e["crc32c"] is not None and _mask(crc32c(raw)) != e["crc32c"]:
            raise IOError("%s: data checksum mismatch" % what)
        return np.frombuffer(raw, dtype=np.dtype(_DTYPES[e["dtype"]]).newbyteorder("<")).reshape(e["shape"])

    out = {}
    try:
        for name, e in idx.items():
            if name in ("", SLICES) or (names is not None and name not in names) or (skip is not None and skip(name)):
                continue
            if e["dtype"] not in _DTYPES:
                continue                                   # strings etc.: nothing the model needs
            if not e["sliced"]:
                out[name] = fetch(e, "variable '%s'" % name).copy()
                continue
            full = np.zeros(e["shape"], dtype=_DTYPES[e["dtype"]])
            covered = np.zeros(e["shape"], dtype=bool)
            for ext in e["slices"]:
                se = idx.get(SLICES, {}).get(slice_key(name, ext))
                if se is None:
                    raise IOError("variable '%s': slice %s is listed but not stored" % (name, ext))
                where = tuple(slice(st, None if ln < 0 else st + ln) for st, ln in ext)
                full[where] = fetch(se, "variable '%s' slice %s" % (name, ext))
                covered[where] = True
            if not covered.all():
                raise IOError("variable '%s': its slices do not cover the full shape %s" % (name, e["shape"]))
            out[name] = full
    finally:
        for f in shards.values():
            f.close()
    return out


def latest_checkpoint(directory):
    """tf.train.latest_checkpoint / get_most_recent_checkpoint (utils/__init__.py): highest-step `model.ckpt-<n>` prefix."""
    best = None
    for fn in os.listdir(directory):
        mo = re.match(r"(model\.ckpt-(\d+))\.index$", fn)
        if mo and (best is None or int(mo.group(2)) > best[0]):
            best = (int(mo.group(2)), os.path.join(directory, mo.group(1)))
    return None if best is None else best[1]


# ---- writer (tests, and exporting weights trained here for the reference to load) ----
def _block(entries, restart_interval=16):
    """LevelDB data block: entries share key prefixes with their predecessor inside a restart interval."""
    body, restarts, prev = bytearray(), [], b""
    for i, (key, val) in enumerate(entries):
        shared = 0
        if restart_interval > 0 and i % restart_interval:
            while shared < min(len(key), len(prev)) and key[shared] == prev[shared]:
                shared += 1
        else:
            restarts.append(len(body))
        body += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(val)) + key[shared:] + val
        prev = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts))
    return bytes(body)


def _entry_proto(dtype_id, shape, shard, offset, size, crc, slices=None):
    dims = b"".join(b"\x12" + _put_varint(len(d)) + d for d in (b"\x08" + _put_varint(int(s)) for s in shape))
    msg = b"\x08" + _put_varint(dtype_id) + b"\x12" + _put_varint(len(dims)) + dims
    if slices is not None:                                   # the full-tensor entry of a partitioned variable: slice specs only
        for ext in slices:
            body = b""
            for start, length in ext:
                e = (b"\x08" + _put_varint(start) if start else b"") + (b"\x10" + _put_varint(length) if length >= 0 else b"")
                body += b"\x0a" + _put_varint(len(e)) + e
            msg += b"\x3a" + _put_varint(len(body)) + body
        return msg
    if shard:
        msg += b"\x18" + _put_varint(shard)
    msg += b"\x20" + _put_varint(offset) + b"\x28" + _put_varint(size) + b"\x35" + struct.pack("<I", crc)
    return msg


def write_checkpoint(prefix, tensors, num_shards=1, block_bytes=4096, restart_interval=16, partition=None):
    """Writes {name: array} as a V2 bundle: `num_shards` data files (variables dealt round robin), an uncompressed LevelDB table
    with ~block_bytes data blocks and prefix-compressed keys.  partition = {name: n}: store that variable as n slices along its
    first axis, the way a partitioned tf variable is saved."""
    partition = partition or {}
    data = [bytearray() for _ in range(num_shards)]
    entries = [(b"", b"\x08" + _put_varint(num_shards) + b"\x1a\x02\x08\x01")]     # header: num_shards, version { producer: 1 }
    turn = [0]

    def store(a):
        sid = turn[0] % num_shards
        turn[0] += 1
        raw = a.astype(a.dtype.newbyteorder("<")).tobytes()
        off = len(data[sid])
        data[sid] += raw
        return sid, off, len(raw), _mask(crc32c(raw))

    for n in sorted(tensors):
        a = np.array(tensors[n], order="C")          # (ascontiguousarray would turn 0-d into 1-d)
        did = _DTYPE_IDS[a.dtype.name]
        parts = int(partition.get(n, 0))
        if parts > 1 and a.ndim >= 1 and a.shape[0] >= parts:
            edges = np.linspace(0, a.shape[0], parts + 1).astype(int)
            specs = [[(int(edges[i]), int(edges[i + 1] - edges[i]))] + [(0, -1)] * (a.ndim - 1) for i in range(parts)]
            entries.append((n.encode("utf-8"), _entry_proto(did, a.shape, 0, 0, 0, 0, slices=specs)))
            for i, ext in enumerate(specs):
                piece = np.ascontiguousarray(a[edges[i]:edges[i + 1]])
                entries.append((slice_key(n, ext), _entry_proto(did, piece.shape, *store(piece))))
        else:
            entries.append((n.encode("utf-8"), _entry_proto(did, a.shape, *store(a))))
    entries.sort(key=lambda kv: kv[0])
    for sid in range(num_shards):
        with open("%s.data-%05d-of-%05d" % (prefix, sid, num_shards), "wb") as f:
            f.write(bytes(data[sid]))
    out = bytearray()
    index_entries = []

    def put(block):
        off = len(out)
        trailer = b"\x00"
        out.extend(block + trailer + struct.pack("<I", _mask(crc32c(block + trailer))))
        return _put_varint(off) + _put_varint(len(block))

    chunk, size = [], 0
    for kv in entries:
        chunk.append(kv)
        size += len(kv[0]) + len(kv[1]) + 3
        if size >= block_bytes:
            index_entries.append((chunk[-1][0], put(_block(chunk, restart_interval))))     # separator = last key of the block
            chunk, size = [], 0
    if chunk:
        index_entries.append((chunk[-1][0], put(_block(chunk, restart_interval))))
    meta = put(_block([]))
    index = put(_block(index_entries, 1))
    footer = meta + index
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC)
    out.extend(footer)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))


# ---- TF variable names -> canonical names ----
def _strip(name):
    for pre in ("model/inference/", "inference/"):
        i = name.find(pre)
        if i >= 0:
            return name[i + len(pre):]
    return name


def _cbhg_rules(scope):
    r = []
    for kind, tfk in (("kernel", "conv1d/kernel"), ("bias", "conv1d/bias"), ("gamma", "batch_normalization/gamma"),
                      ("beta", "batch_normalization/beta"), ("moving_mean", "batch_normalization/moving_mean"),
                      ("moving_variance", "batch_normalization/moving_variance")):
        r.append((re.compile(r"^%s/conv_bank/conv1d_(\d+)/%s$" % (scope, tfk)), scope + r"/conv_bank/conv1d_\1/" + kind))
        r.append((re.compile(r"^%s/proj_(\d+)/%s$" % (scope, tfk)), scope + r"/proj_\1/" + kind))
    r.append((re.compile(r"^%s/highway_(\d+)/([HT])/(kernel|bias)$" % scope), scope + r"/highway_\1/\2/\3"))
    r.append((re.compile(r"^%s/bidirectional_rnn/(fw|bw)/gru_cell/(gates|candidate)/(kernel|bias)$" % scope), scope + r"/bigru/\1/\2/\3"))
    r.append((re.compile(r"^%s/dense/(kernel|bias)$" % scope), scope + r"/dense/\1"))
    return r


def match_tf_names(tf_shapes, spec):
    """tf_shapes {tf variable name: shape}, spec [(canonical name, shape)] -> {canonical name: tf variable name}.  Scope prefixes
    (`model/inference/`), optimizer slots and `global_step` are ignored; variables whose TF name does not identify them (the two
    OutputProjectionWrappers, the deepvoice dense layers, the linear head: all `.../kernel`) are told apart by shape and creation
    order.  Raises KeyError listing what is missing / mis-shaped / unmatched."""
    want = dict(spec)
    out = {}
    rules = [(re.compile(r"^embedding$"), "embedding"), (re.compile(r"^speaker_embedding$"), "speaker_embedding"),
             (re.compile(r"^prenet/dense_(\d+)/(kernel|bias)$"), r"prenet/dense_\1/\2"),
             # get_embed tables of the speaker_embedding_size == 1 deepvoice variant (tacotron.py:52-66)
             (re.compile(r"^before_highway$"), "spk/before_highway/table"), (re.compile(r"^encoder_rnn_init_state$"), "spk/encoder_rnn_init/table"),
             (re.compile(r"^attention_rnn_init_state$"), "spk/attention_rnn_init/table"),
             (re.compile(r"^decoder_rnn_init_states(\d+)$"), r"spk/decoder_rnn_init_\1/table"),
             (re.compile(r"(^|/)memory_layer/kernel$"), "attention/memory_layer/kernel"),
             (re.compile(r"/query_layer/kernel$"), "attention/query_layer/kernel"),
             (re.compile(r"/attention_v$"), "attention/attention_v"), (re.compile(r"/attention_score_bias$"), "attention/attention_score_bias"),
             (re.compile(r"/attention_g$"), "attention/attention_g"), (re.compile(r"/attention_b$"), "attention/attention_b"),
             (re.compile(r"decoder.*/decoder_prenet/dense_(\d+)/(kernel|bias)$"), r"decoder/prenet/dense_\1/\2"),
             (re.compile(r"decoder.*/cell_([1-9])/.*gru_cell/(gates|candidate)/(kernel|bias)$"), r"decoder/gru_\1/\2/\3"),
             (re.compile(r"decoder.*/cell_0/.*gru_cell/(gates|candidate)/(kernel|bias)$"), r"decoder/attention_gru/\1/\2"),
             ] + _cbhg_rules("encoder_cbhg") + _cbhg_rules("post_cbhg")
    leftovers = {}
    for tfn, shp in tf_shapes.items():
        n = _strip(tfn)
        if n == "global_step" or re.search(r"/Adam(_1)?$|beta[12]_power$", n):
            continue                                                     # optimizer slots
        for rx, repl in rules:
            mo = rx.search(n)
            if mo:
                out[mo.expand(repl)] = tfn
                break
        else:
            leftovers[n] = tfn
    # what is left is told apart by shape: the two output_projection_wrappers, the deepvoice dense layers, the linear head
    def take(canon_kernel, canon_bias, pred):
        if canon_kernel not in want or canon_kernel in out:
            return
        ks = want[canon_kernel]
        for n, tfn in sorted(leftovers.items()):
            if n.endswith("/kernel") and tuple(tf_shapes[tfn]) == tuple(ks) and pred(n):
                out[canon_kernel] = tfn
                b = n[:-len("kernel")] + "bias"
                if b in leftovers and canon_bias in want:
                    out[canon_bias] = leftovers.pop(b)
                del leftovers[n]
                return
    take("decoder/concat_projection/kernel", "decoder/concat_projection/bias", lambda n: "cell_0" in n and "output_projection" in n)
    take("decoder/frame_projection/kernel", "decoder/frame_projection/bias", lambda n: n.startswith("decoder") and "output_projection" in n)
    for i, nm in enumerate(["before_highway", "encoder_rnn_init", "attention_rnn_init", "decoder_rnn_init_1", "decoder_rnn_init_2",
                            "decoder_rnn_init_3", "decoder_rnn_init_4"]):
        tfd = "dense" if i == 0 else "dense_%d" % i                       # creation order, tacotron.py:71-79
        take("spk/%s/kernel" % nm, "spk/%s/bias" % nm, lambda n, tfd=tfd: n == tfd + "/kernel")
    take("linear/kernel", "linear/bias", lambda n: re.match(r"^dense(_\d+)?/kernel$", n) is not None)
    missing = [k for k in want if k not in out]
    bad = [k for k in want if k in out and tuple(tf_shapes[out[k]]) != tuple(want[k])]
    if missing or bad:
        raise KeyError("TF checkpoint does not provide %s; wrong shapes for %s; unmatched TF variables: %s"
                       % (missing[:8], [(k, tuple(tf_shapes[out[k]]), want[k]) for k in bad[:4]], sorted(leftovers)[:8]))
    return {k: out[k] for k in want}


def map_tf_names(tf_vars, spec):
    """tf_vars {tf name: array}, spec [(canonical name, shape)] -> {canonical: array}.  Raises listing what is missing."""
    names = match_tf_names({k: np.shape(v) for k, v in tf_vars.items()}, spec)
    return {k: np.asarray(tf_vars[t], np.float32) for k, t in names.items()}


def import_tf_checkpoint(prefix_or_dir, hparams, num_speakers=1, verify=True):
    """Canonical weight dict (weights.weight_spec order) from a reference-trained checkpoint."""
    from .weights import weight_spec
    prefix = prefix_or_dir
    if os.path.isdir(prefix_or_dir):
        prefix = latest_checkpoint(prefix_or_dir)
        if prefix is None:
            raise IOError("no model.ckpt-<step>.index under %s" % prefix_or_dir)
    return map_tf_names(read_checkpoint(prefix, verify, skip=is_optimizer_slot), weight_spec(hparams, num_speakers))


def tf_names_for(spec, attention_type="bah_mon"):
    """Canonical spec -> the TF 1.4-era variable names the reference creates (App. B); used by the exporter and the tests."""
    att = {"bah_mon": "bahdanau_monotonic_attention", "bah": "bahdanau_attention", "bah_norm": "bahdanau_attention"}[attention_type]
    base = "decoder/output_projection_wrapper/multi_rnn_cell/cell_0/output_projection_wrapper/concat_output_and_attention_wrapper/attention_wrapper"
    out = {}
    spk_order = ["before_highway", "encoder_rnn_init", "attention_rnn_init", "decoder_rnn_init_1", "decoder_rnn_init_2"]
    n_spk_dense = sum(1 for n, _ in spec if n.startswith("spk/") and n.endswith("/kernel"))
    for name, _ in spec:
        p = name.split("/")
        if name in ("embedding", "speaker_embedding"):
            t = name
        elif p[0] == "prenet":
            t = name
        elif p[0] in ("encoder_cbhg", "post_cbhg"):
            if p[1] in ("conv_bank",):
                leaf = p[3]
                t = "%s/conv_bank/%s/%s" % (p[0], p[2], ("conv1d/" + leaf) if leaf in ("kernel", "bias") else "batch_normalization/" + leaf)
            elif p[1].startswith("proj_"):
                leaf = p[2]
                t = "%s/%s/%s" % (p[0], p[1], ("conv1d/" + leaf) if leaf in ("kernel", "bias") else "batch_normalization/" + leaf)
            elif p[1] == "bigru":
                t = "%s/bidirectional_rnn/%s/gru_cell/%s/%s" % (p[0], p[2], p[3], p[4])
            else:
                t = name
        elif name == "attention/memory_layer/kernel":
            t = "memory_layer/kernel"
        elif p[0] == "attention":
            t = "%s/%s/%s" % (base, att, "/".join(p[1:]))
        elif p[:2] == ["decoder", "prenet"]:
            t = "%s/decoder_prenet_wrapper/decoder_prenet/%s" % (base, "/".join(p[2:]))
        elif p[:2] == ["decoder", "attention_gru"]:
            t = "%s/decoder_prenet_wrapper/gru_cell/%s" % (base, "/".join(p[2:]))
        elif p[:2] == ["decoder", "concat_projection"]:
            t = "decoder/output_projection_wrapper/multi_rnn_cell/cell_0/output_projection_wrapper/" + p[2]
        elif p[0] == "decoder" and p[1].startswith("gru_"):
            t = "decoder/output_projection_wrapper/multi_rnn_cell/cell_%s/gru_cell/%s" % (p[1][4:], "/".join(p[2:]))
        elif p[:2] == ["decoder", "frame_projection"]:
            t = "decoder/output_projection_wrapper/" + p[2]
        elif p[0] == "spk" and p[2] == "table":
            t = {"before_highway": "before_highway", "encoder_rnn_init": "encoder_rnn_init_state",
                 "attention_rnn_init": "attention_rnn_init_state"}.get(p[1]) or "decoder_rnn_init_states" + p[1].rsplit("_", 1)[1]
        elif p[0] == "spk":
            i = spk_order.index(p[1]) if p[1] in spk_order else int(p[1].rsplit("_", 1)[1]) + 2
            t = ("dense" if i == 0 else "dense_%d" % i) + "/" + p[2]
        elif p[0] == "linear":
            t = ("dense" if n_spk_dense == 0 else "dense_%d" % n_spk_dense) + "/" + p[1]
        else:
            t = name
        out[name] = "model/inference/" + t
    return out


def export_tf_checkpoint(prefix, weights, spec, attention_type="bah_mon", global_step=0, num_shards=1):
    """Writes canonical weights as a bundle with the reference's variable names (for `saver.restore` on the TF side)."""
    names = tf_names_for(spec, attention_type)
    tensors = {names[n]: np.asarray(weights[n], np.float32) for n, _ in spec}
    tensors["global_step"] = np.asarray(global_step, np.int32)
    write_checkpoint(prefix, tensors, num_shards=num_shards)

"""TensorFlow V2 checkpoint ("tensor bundle") reader / writer without TensorFlow.

The reference saves with `tf.train.Saver` (model/base.py:61-69): `<dir>/model_weights/model.cpkt-<epoch>.index`
+ `.data-00000-of-00001` (+ `.meta`, the graph, unused here).  TensorFlow is not in this image, so the on-disk
format is restated from its published layout:

* `.index` is a LevelDB-format sorted table (tensorflow/core/lib/io/table*: prefix-compressed blocks with restart
  arrays, 5-byte block trailer {compression type, masked crc32c}, 48-byte footer ending in the magic
  0xdb4775248b80fb57).  Key "" -> BundleHeaderProto, every other key = variable name -> BundleEntryProto
  {dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6} (tensorflow/core/protobuf/tensor_bundle.proto).
* `.data-SSSSS-of-NNNNN` holds the raw little-endian tensor bytes at [offset, offset+size).

PARITY NOTE: no TensorFlow-written checkpoint is available in this container, so the reader is pinned only by
round trips through `write_bundle` below and by the format description above ("interop unpinned").  It reads
uncompressed blocks (what TF's BundleWriter emits) and fails loudly on snappy blocks.
"""
import os
import struct

import numpy as np

MAGIC = 0xdb4775248b80fb57
# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
           17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}

_CRC_TABLE = None


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli), table driven."""
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = np.zeros(256, np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t[i] = c
        _CRC_TABLE = [int(x) for x in t]
    tab = _CRC_TABLE
    c = crc ^ 0xFFFFFFFF
    for b in bytes(data):
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


# ----------------------------------------------------------------- varints / protobuf --
def _get_varint(buf, pos):
    shift, val = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
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


def _parse_proto(buf):
    """{field: [values]}; varint -> int, 64-bit / 32-bit -> raw bytes, length-delimited -> bytes."""
    out, pos = {}, 0
    while pos < len(buf):
        key, pos = _get_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v, pos = buf[pos:pos + 8], pos + 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v, pos = buf[pos:pos + n], pos + n
        elif wt == 5:
            v, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.setdefault(field, []).append(v)
    return out


def _signed64(v):
    return v - (1 << 64) if v >= 1 << 63 else v


def _parse_entry(buf):
    p = _parse_proto(buf)
    dims = []
    for shp in p.get(2, []):
        for d in _parse_proto(shp).get(2, []):
            dims.append(_signed64(_parse_proto(d).get(1, [0])[0]))
    return {"dtype": p.get(1, [0])[0], "shape": tuple(dims), "shard": p.get(3, [0])[0], "offset": p.get(4, [0])[0],
            "size": p.get(5, [0])[0], "crc": struct.unpack("<I", p[6][0])[0] if 6 in p else None, "sliced": 7 in p}


# ----------------------------------------------------------------- table reader --
def _read_block(f, offset, size):
    f.seek(offset)
    raw = f.read(size + 5)
    if len(raw) != size + 5:
        raise IOError("truncated table block")
    if raw[size] == 1:
        raise NotImplementedError("snappy-compressed index block (python-snappy is not available)")
    if raw[size] != 0:
        raise ValueError("unknown block compression %d" % raw[size])
    return raw[:size]


def _block_entries(block):
    n_restarts = struct.unpack("<I", block[-4:])[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_index(path):
    """{name: entry dict} of a `.index` file (key "" = header is dropped)."""
    with open(path, "rb") as f:
        f.seek(0, os.SEEK_END)
        n = f.tell()
        if n < 48:
            raise IOError("not a tensor-bundle index: %s" % path)
        f.seek(n - 48)
        footer = f.read(48)
        if struct.unpack("<Q", footer[40:])[0] != MAGIC:
            raise IOError("bad table magic in %s" % path)
        _, pos = _get_varint(footer, 0)                  # metaindex handle
        _, pos = _get_varint(footer, pos)
        ioff, pos = _get_varint(footer, pos)             # index handle
        isz, pos = _get_varint(footer, pos)
        entries = {}
        for _, handle in _block_entries(_read_block(f, ioff, isz)):
            boff, p2 = _get_varint(handle, 0)
            bsz, _ = _get_varint(handle, p2)
            for key, val in _block_entries(_read_block(f, boff, bsz)):
                if key:
                    entries[key.decode("utf-8")] = _parse_entry(val)
    return entries


def read_bundle(prefix, names=None, verify_crc=False):
    """{variable name: ndarray} of the checkpoint `<prefix>.index` / `<prefix>.data-*` (optionally only `names`)."""
    entries = read_index(prefix + ".index")
    shards = sorted(f for f in os.listdir(os.path.dirname(prefix) or ".") if f.startswith(os.path.basename(prefix) + ".data-"))
    out = {}
    for name, e in entries.items():
        if names is not None and name not in names:
            continue
        if e["sliced"]:
            raise NotImplementedError("partitioned variable %s" % name)
        if e["dtype"] not in _DTYPES:
            continue                                     # strings / resources: not tensors this model has
        fn = os.path.join(os.path.dirname(prefix) or ".", shards[e["shard"]])
        with open(fn, "rb") as f:
            f.seek(e["offset"])
            raw = f.read(e["size"])
        if verify_crc and e["crc"] is not None and masked_crc(raw) != e["crc"]:
            raise IOError("crc mismatch for %s" % name)
        out[name] = np.frombuffer(raw, dtype=_DTYPES[e["dtype"]]).reshape(e["shape"]).copy()
    return out


# ----------------------------------------------------------------- writer --
def _block(pairs):
    """One table block with a restart point at every entry (no prefix sharing)."""
    body, restarts = bytearray(), []
    for k, v in pairs:
        restarts.append(len(body))
        body += _put_varint(0) + _put_varint(len(k)) + _put_varint(len(v)) + k + v
    for r in restarts:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts))
    return bytes(body)


def _with_trailer(block):
    return block + b"\x00" + struct.pack("<I", masked_crc(block + b"\x00"))


def _field(num, wt, payload):
    return _put_varint(num << 3 | wt) + payload


def _entry_proto(arr, offset):
    shape = b"".join(_field(2, 2, (lambda d: _put_varint(len(d)) + d)(_field(1, 0, _put_varint(int(s))))) for s in arr.shape)
    p = _field(1, 0, _put_varint(_DTYPE_IDS[arr.dtype]))
    p += _field(2, 2, _put_varint(len(shape)) + shape)
    if offset:
        p += _field(4, 0, _put_varint(offset))
    p += _field(5, 0, _put_varint(arr.nbytes))
    p += _field(6, 5, struct.pack("<I", masked_crc(arr.tobytes())))
    return p


def write_bundle(prefix, tensors):
    """Write {name: ndarray} as a single-shard, uncompressed bundle (the layout tf.train.Saver produces)."""
    names = sorted(tensors, key=lambda s: s.encode("utf-8"))
    pairs, offset = [], 0
    with open(prefix + ".data-00000-of-00001", "wb") as d:
        for n in names:
            a = np.asarray(tensors[n])
            if a.dtype.byteorder == ">":
                a = a.astype(a.dtype.newbyteorder("<"))
            pairs.append((n.encode("utf-8"), _entry_proto(a, offset)))
            d.write(a.tobytes())
            offset += a.nbytes
    header = _field(1, 0, _put_varint(1)) + _field(3, 2, (lambda v: _put_varint(len(v)) + v)(_field(1, 0, _put_varint(1))))
    pairs = [(b"", header)] + pairs
    out, handles = bytearray(), []
    for i in range(0, len(pairs), 16):                   # several data blocks so the index block is exercised
        blk = _block(pairs[i:i + 16])
        handles.append((pairs[min(i + 15, len(pairs) - 1)][0], len(out), len(blk)))
        out += _with_trailer(blk)
    meta = _block([])
    meta_h = (len(out), len(meta))
    out += _with_trailer(meta)
    idx = _block([(k, _put_varint(o) + _put_varint(s)) for k, o, s in handles])
    idx_h = (len(out), len(idx))
    out += _with_trailer(idx)
    footer = _put_varint(meta_h[0]) + _put_varint(meta_h[1]) + _put_varint(idx_h[0]) + _put_varint(idx_h[1])
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", MAGIC)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out) + footer)

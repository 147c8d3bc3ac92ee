"""Reader / writer for TensorFlow checkpoint-v2 ("tensor bundle") files without TensorFlow.

This is the on-disk format of the reference's two savers (codes/base.py:35-85): `tf.train.Saver.save(sess,
<checkpoint_dir>/vae-model)` / `.../prior-model` produce

    <prefix>.index                  an SSTable (leveldb table format, no compression): key "" -> BundleHeaderProto,
                                    key <variable name> -> BundleEntryProto{dtype, shape, shard_id, offset, size, crc32c}
    <prefix>.data-00000-of-00001    the raw little-endian tensor bytes, in key order, at those offsets
    <prefix>.meta                   MetaGraphDef (only its existence is tested by the reference: base.py:70,78)
    checkpoint                      text proto naming the latest prefix

so a user holding reference checkpoints can load them here, and checkpoints written here restore in the reference
(`saver.restore` reads .index/.data only).  Checksums are CRC-32C, stored "masked" (rotate right 15, add 0xa282ead8) as
TensorFlow / leveldb do; every block of the .index and every tensor's bytes are verified on read.

The writer is pinned against TensorFlow-produced bytes: re-serialising the entries parsed from the reference's own
`pretrained_models/*/*.index` files reproduces those files byte for byte (tests/test_host_cpu.py).
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
BLOCK_RESTART_INTERVAL = 16        # table::Options default used by BundleWriter
BLOCK_SIZE = 262144
MASK_DELTA = 0xa282ead8

# DataType enum values (tensorflow/core/framework/types.proto) for the dtypes a Saver of this model can hold
DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 9: np.dtype("<i8"), 4: np.dtype("u1")}
DTYPE_IDS = {v: k for k, v in DTYPES.items()}


class BundleError(IOError):
    pass


# ------------------------------------------------------------------------------------------------ crc32c
_PY_TABLE = None


def _crc32c_py(data, crc=0):
    """Table-driven CRC-32C; for the few-KB .index blocks (tensor payloads go through the C ABI's slicing-by-8)."""
    global _PY_TABLE
    if _PY_TABLE is None:
        tb = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tb.append(c)
        _PY_TABLE = tb
    c = crc ^ 0xFFFFFFFF
    tb = _PY_TABLE
    for b in bytes(data):
        c = tb[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def crc32c(data, crc=0):
    """CRC-32C of a bytes-like / contiguous numpy array."""
    mv = memoryview(data).cast("B") if not isinstance(data, (bytes, bytearray)) else data
    if len(mv) < 4096:
        return _crc32c_py(mv, crc)
    from .. import _lib as L
    arr = np.frombuffer(mv, dtype=np.uint8)
    return int(L.query("ladder_crc32c_extend", crc, arr.ctypes.data, arr.size))


def mask_crc(c):
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(m):
    r = (m - MASK_DELTA) & 0xFFFFFFFF
    return ((r >> 17) | (r << 15)) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------ varints / protobuf
def _put_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _get_varint(buf, pos):
    v = shift = 0
    while True:
        if pos >= len(buf):
            raise BundleError("truncated varint")
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, pos
        shift += 7


def _fields(buf):
    pos = 0
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        fn, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise BundleError("unsupported protobuf wire type %d" % wt)
        yield fn, wt, v


def _encode_entry(dtype_id, shape, offset, size, crc_masked, shard_id=0):
    """BundleEntryProto in TensorFlow's field order; zero-valued scalar fields are omitted (proto3)."""
    shp = b"".join(b"\x12" + _put_varint(len(d)) + d for d in ((b"\x08" + _put_varint(s)) if s else b"" for s in shape))
    out = b"\x08" + _put_varint(dtype_id) + b"\x12" + _put_varint(len(shp)) + shp
    if shard_id:
        out += b"\x18" + _put_varint(shard_id)
    if offset:
        out += b"\x20" + _put_varint(offset)
    if size:
        out += b"\x28" + _put_varint(size)
    if crc_masked:
        out += b"\x35" + struct.pack("<I", crc_masked)
    return out


def _decode_entry(val):
    ent = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=0, sliced=False)
    for fn, _, v in _fields(val):
        if fn == 1:
            ent["dtype"] = v
        elif fn == 2:
            for f2, _, v2 in _fields(v):
                if f2 == 2:
                    sz = 0
                    for f3, _, v3 in _fields(v2):
                        if f3 == 1:
                            sz = v3
                    ent["shape"].append(sz)
        elif fn == 3:
            ent["shard_id"] = v
        elif fn == 4:
            ent["offset"] = v
        elif fn == 5:
            ent["size"] = v
        elif fn == 6:
            ent["crc32c"] = v
        elif fn == 7:
            ent["sliced"] = True
    return ent


HEADER = b"\x08\x01\x1a\x02\x08\x01"      # BundleHeaderProto{num_shards=1, endianness=LITTLE, version{producer=1}}


# ------------------------------------------------------------------------------------------------ table blocks
class _BlockBuilder:
    def __init__(self, restart_interval):
        self.ri = restart_interval
        self.buf = bytearray()
        self.restarts = [0]
        self.count = 0
        self.last = b""

    def add(self, key, val):
        shared = 0
        if self.count < self.ri:
            n = min(len(self.last), len(key))
            while shared < n and self.last[shared] == key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(val)) + key[shared:] + val
        self.last = key
        self.count += 1

    def size_estimate(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def _short_successor(key):
    """leveldb BytewiseComparator::FindShortSuccessor: first byte that can be incremented, truncated there."""
    for i, b in enumerate(key):
        if b != 0xFF:
            return key[:i] + bytes([b + 1])
    return key


def _shortest_separator(start, limit):
    """leveldb BytewiseComparator::FindShortestSeparator."""
    n = min(len(start), len(limit))
    i = 0
    while i < n and start[i] == limit[i]:
        i += 1
    if i < n and start[i] < 0xFF and start[i] + 1 < limit[i]:
        return start[:i] + bytes([start[i] + 1])
    return start


def _block_with_trailer(contents):
    return contents + b"\x00" + struct.pack("<I", mask_crc(crc32c(contents + b"\x00")))


def write_index(path, entries):
    """entries: {name: dict(dtype, shape, offset, size, crc32c[, shard_id])} -> TF-format .index file."""
    items = [(b"", HEADER)] + [(k.encode(), _encode_entry(e["dtype"], e["shape"], e["offset"], e["size"], e["crc32c"],
                                                          e.get("shard_id", 0))) for k, e in sorted(entries.items())]
    out = bytearray()
    index = _BlockBuilder(1)
    blk = _BlockBuilder(BLOCK_RESTART_INTERVAL)
    pending = None                                   # (last key of the finished block, handle) awaiting the next key

    def flush():
        nonlocal blk, pending
        contents = blk.finish()
        handle = _put_varint(len(out)) + _put_varint(len(contents))
        out.extend(_block_with_trailer(contents))
        pending = (blk.last, handle)
        blk = _BlockBuilder(BLOCK_RESTART_INTERVAL)

    for key, val in items:
        if pending is not None:
            index.add(_shortest_separator(pending[0], key), pending[1])
            pending = None
        blk.add(key, val)
        if blk.size_estimate() >= BLOCK_SIZE:
            flush()
    if blk.count or not out:
        flush()
    if pending is not None:
        index.add(_short_successor(pending[0]), pending[1])
    meta = _BlockBuilder(BLOCK_RESTART_INTERVAL).finish()
    meta_handle = _put_varint(len(out)) + _put_varint(len(meta))
    out.extend(_block_with_trailer(meta))
    idx = index.finish()
    idx_handle = _put_varint(len(out)) + _put_varint(len(idx))
    out.extend(_block_with_trailer(idx))
    footer = meta_handle + idx_handle
    out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC))
    with open(path, "wb") as f:
        f.write(bytes(out))


def _read_block(buf, off, size, verify=True):
    if off + size + 5 > len(buf):
        raise BundleError("block beyond end of file")
    if buf[off + size] != 0:
        raise BundleError("compressed table blocks are not supported (type %d)" % buf[off + size])
    if verify:
        want = unmask_crc(struct.unpack_from("<I", buf, off + size + 1)[0])
        if crc32c(bytes(buf[off:off + size + 1])) != want:
            raise BundleError("block checksum mismatch at offset %d" % off)
    blk = buf[off:off + size]
    n_restarts = struct.unpack_from("<I", blk, size - 4)[0]
    end = size - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(blk, pos)
        non_shared, pos = _get_varint(blk, pos)
        vlen, pos = _get_varint(blk, pos)
        key = key[:shared] + bytes(blk[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(blk[pos:pos + vlen])
        pos += vlen


def read_index(path, verify=True):
    """-> {name: entry dict} (the header entry is validated and dropped)."""
    buf = open(path, "rb").read()
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != TABLE_MAGIC:
        raise BundleError("%s is not a tensor-bundle index (bad table magic)" % path)
    footer = buf[-48:]
    _, pos = _get_varint(footer, 0)
    _, pos = _get_varint(footer, pos)
    idx_off, pos = _get_varint(footer, pos)
    idx_size, pos = _get_varint(footer, pos)
    out, seen_header = {}, False
    for _, handle in _read_block(buf, idx_off, idx_size, verify):
        boff, p = _get_varint(handle, 0)
        bsize, p = _get_varint(handle, p)
        for key, val in _read_block(buf, boff, bsize, verify):
            if key == b"":
                seen_header = True
                hdr = {fn: v for fn, _, v in _fields(val)}
                if hdr.get(2, 0) != 0:
                    raise BundleError("big-endian bundles are not supported")
                continue
            out[key.decode()] = _decode_entry(val)
    if not seen_header:
        raise BundleError("bundle header entry missing")
    return out


# ------------------------------------------------------------------------------------------------ whole checkpoints
def _data_path(prefix, shard, n):
    return "%s.data-%05d-of-%05d" % (prefix, shard, n)


def save_checkpoint(prefix, tensors):
    """tf.train.Saver.save(sess, prefix) for a {variable name: ndarray} dict: .index, .data-00000-of-00001, .meta, checkpoint."""
    entries, off = {}, 0
    with open(_data_path(prefix, 0, 1), "wb") as f:
        for name in sorted(tensors):
            a = np.asarray(tensors[name], order="C")         # (ascontiguousarray would promote a scalar variable to 1-D)
            if a.dtype.newbyteorder("<") not in DTYPE_IDS and a.dtype not in DTYPE_IDS:
                raise BundleError("dtype %s of %s has no checkpoint encoding here" % (a.dtype, name))
            a = a.astype(a.dtype.newbyteorder("<"), copy=False)
            raw = a.tobytes()
            f.write(raw)
            entries[name] = dict(dtype=DTYPE_IDS[a.dtype], shape=list(a.shape), offset=off, size=len(raw),
                                 crc32c=mask_crc(crc32c(raw)))
            off += len(raw)
    write_index(prefix + ".index", entries)
    if not os.path.exists(prefix + ".meta"):
        open(prefix + ".meta", "wb").close()          # empty MetaGraphDef: the reference only tests that the file exists
    base = os.path.basename(prefix)
    with open(os.path.join(os.path.dirname(prefix) or ".", "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))


def load_checkpoint(prefix, names=None, verify=True):
    """-> {variable name: ndarray}; `names` restricts the read.  Raises BundleError on a missing shard / bad checksum."""
    entries = read_index(prefix + ".index", verify)
    shards = 1 + max([e["shard_id"] for e in entries.values()] or [0])
    out = {}
    files = {}
    try:
        for name, e in entries.items():
            if names is not None and name not in names:
                continue
            if e["sliced"]:
                raise BundleError("partitioned variable %s is not supported" % name)
            if e["dtype"] not in DTYPES:
                raise BundleError("dtype enum %d of %s is not supported" % (e["dtype"], name))
            sid = e["shard_id"]
            if sid not in files:
                p = _data_path(prefix, sid, shards)
                if not os.path.isfile(p):
                    raise BundleError("data shard %s is missing" % p)
                files[sid] = open(p, "rb")
            f = files[sid]
            f.seek(e["offset"])
            raw = f.read(e["size"])
            dt = DTYPES[e["dtype"]]
            count = int(np.prod(e["shape"])) if e["shape"] else 1
            if len(raw) != e["size"] or count * dt.itemsize != e["size"]:
                raise BundleError("tensor %s: size %d does not match shape %s" % (name, e["size"], e["shape"]))
            if verify and mask_crc(crc32c(raw)) != e["crc32c"]:
                raise BundleError("tensor %s: checksum mismatch" % name)
            out[name] = np.frombuffer(raw, dtype=dt).reshape(e["shape"]).copy()
    finally:
        for f in files.values():
            f.close()
    return out

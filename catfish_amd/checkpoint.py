"""TensorFlow checkpoint-V2 bundle reader/writer without TensorFlow.

The reference restores its weights with ``tf.train.Saver.restore``
(reference catfish/models/rnn_class.py:191-198) from the bundle
``catfish/ResNetRNN/checkpoints/ckpnt-30000.{index,data-00000-of-00001}``.
TensorFlow is not available on MI355X hosts, so this module parses the bundle
directly:

* ``.index`` is a leveldb SSTable (footer magic 0xdb4775248b80fb57,
  prefix-compressed key blocks, no compression) whose values are protobuf
  messages: key ``""`` -> BundleHeaderProto, every other key ->
  BundleEntryProto {1: dtype, 2: shape, 3: shard_id, 4: offset, 5: size,
  6: crc32c (masked, fixed32)}.
* ``.data-00000-of-00001`` is the raw little-endian tensor bytes.

Every tensor is verified against its masked CRC-32C, which doubles as the
known-answer test for this reader.

A small writer is included so that weights produced here can be loaded back
by the original tool (and so that the reader can be round-trip tested on
machines where the reference checkpoint is absent).
"""
from __future__ import annotations

import os
import struct
from collections import OrderedDict
from typing import Dict, Iterable, List, Tuple

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
_FOOTER_LEN = 48
_BLOCK_TRAILER = 5  # 1 byte compression type + 4 byte crc
_MASK_DELTA = 0xA282EAD8

# TensorFlow DataType enum values we understand.
DT_FLOAT = 1
DT_DOUBLE = 2
DT_INT32 = 3
DT_INT64 = 9
_DTYPES = {DT_FLOAT: np.dtype("<f4"), DT_DOUBLE: np.dtype("<f8"),
           DT_INT32: np.dtype("<i4"), DT_INT64: np.dtype("<i8")}
_DTYPES_INV = {v: k for k, v in _DTYPES.items()}


# --------------------------------------------------------------------------- crc32c
def _make_crc_table() -> np.ndarray:
    poly = 0x82F63B78
    tbl = np.zeros(256, dtype=np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        tbl[i] = c
    return tbl


_CRC_TABLE = _make_crc_table()
_CRC_TABLE_LIST = [int(v) for v in _CRC_TABLE]


_native_crc = None          # cf_crc32c once the library has been found; False when it is not there (this module then stays pure Python)


def crc32c(data: bytes, crc: int = 0) -> int:
    """CRC-32C (Castagnoli), as used by leveldb / TF bundles.  Through the library's ``cf_crc32c`` when it is built (slicing-by-8:
    the 0.79 MB of inference tensors take 0.4 ms instead of 45), else the byte loop below -- a checksum, the same either way
    (``crc32c_python`` is the loop, kept callable for the test that compares the two)."""
    global _native_crc
    if _native_crc is None:
        try:
            from . import _native
            _native_crc = _native.lib().cf_crc32c
        except Exception:           # noqa: BLE001 -- no library (a machine that only inspects checkpoints): the Python loop
            _native_crc = False
    if _native_crc:
        buf = bytes(data) if not isinstance(data, (bytes, bytearray)) else data
        return int(_native_crc(buf if isinstance(buf, bytes) else bytes(buf), len(buf), int(crc) & 0xFFFFFFFF))
    return crc32c_python(data, crc)


def crc32c_python(data: bytes, crc: int = 0) -> int:
    """The same in plain Python (table look-up per byte)."""
    c = crc ^ 0xFFFFFFFF
    tbl = _CRC_TABLE_LIST
    for b in data:
        c = tbl[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(crc: int) -> int:
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + _MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(masked: int) -> int:
    rot = (masked - _MASK_DELTA) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# --------------------------------------------------------------------------- varints / protobuf
def _get_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    result = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _put_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _parse_proto(buf: bytes) -> Dict[int, list]:
    """Minimal protobuf wire-format parser: field number -> list of raw values."""
    fields: Dict[int, list] = {}
    pos = 0
    n = len(buf)
    while pos < n:
        key, pos = _get_varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _get_varint(buf, pos)
        elif wt == 1:
            val = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            val = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        fields.setdefault(fno, []).append(val)
    return fields


def _parse_shape(buf: bytes) -> Tuple[int, ...]:
    dims = []
    for dim in _parse_proto(buf).get(2, []):
        d = _parse_proto(dim)
        dims.append(int(d.get(1, [0])[0]))
    return tuple(dims)


class BundleEntry(object):
    __slots__ = ("name", "dtype", "shape", "shard_id", "offset", "size", "crc32c")

    def __init__(self, name, dtype, shape, shard_id, offset, size, crc):
        self.name = name
        self.dtype = dtype
        self.shape = shape
        self.shard_id = shard_id
        self.offset = offset
        self.size = size
        self.crc32c = crc  # masked, as stored

    def as_dict(self):
        return {"name": self.name, "dtype": int(self.dtype), "shape": list(self.shape),
                "offset": int(self.offset), "size": int(self.size), "crc32c": int(self.crc32c)}


# --------------------------------------------------------------------------- SSTable reader
def _read_block_handle(buf: bytes, pos: int) -> Tuple[int, int, int]:
    off, pos = _get_varint(buf, pos)
    size, pos = _get_varint(buf, pos)
    return off, size, pos


def _iter_block(block: bytes) -> Iterable[Tuple[bytes, bytes]]:
    """Iterate (key, value) over one leveldb block (restart array stripped here)."""
    if len(block) < 4:
        raise ValueError("corrupt block")
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos = 0
    key = b""
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        val = block[pos:pos + vlen]
        pos += vlen
        yield key, val


def _read_block(buf: bytes, off: int, size: int, verify: bool = True) -> bytes:
    raw = buf[off:off + size]
    ctype = buf[off + size]
    if ctype != 0:
        raise ValueError("compressed SSTable blocks (type %d) are not supported" % ctype)
    if verify:
        stored = struct.unpack_from("<I", buf, off + size + 1)[0]
        actual = mask_crc(crc32c(buf[off:off + size + 1]))
        if stored != actual:
            raise ValueError("SSTable block crc mismatch at offset %d" % off)
    return raw


def read_index(index_path: str) -> "OrderedDict[str, BundleEntry]":
    """Parse a ``*.index`` file into an ordered {tensor name: BundleEntry}."""
    with open(index_path, "rb") as fh:
        buf = fh.read()
    if len(buf) < _FOOTER_LEN:
        raise ValueError("%s: too short for an SSTable" % index_path)
    footer = buf[-_FOOTER_LEN:]
    magic = struct.unpack_from("<Q", footer, 40)[0]
    if magic != TABLE_MAGIC:
        raise ValueError("%s: bad SSTable magic %#x" % (index_path, magic))
    _mi_off, _mi_size, pos = _read_block_handle(footer, 0)
    ix_off, ix_size, pos = _read_block_handle(footer, pos)
    index_block = _read_block(buf, ix_off, ix_size)
    entries: "OrderedDict[str, BundleEntry]" = OrderedDict()
    for _sep_key, handle in _iter_block(index_block):
        d_off, d_size, _ = _read_block_handle(handle, 0)
        for key, val in _iter_block(_read_block(buf, d_off, d_size)):
            if key == b"":
                continue  # BundleHeaderProto
            f = _parse_proto(val)
            dtype = int(f.get(1, [0])[0])
            shape = _parse_shape(f[2][0]) if 2 in f else ()
            entries[key.decode("utf-8")] = BundleEntry(
                key.decode("utf-8"), dtype, shape,
                int(f.get(3, [0])[0]), int(f.get(4, [0])[0]),
                int(f.get(5, [0])[0]), int(f.get(6, [0])[0]))
    return entries


def _resolve_prefix(path: str, ckpnt: str) -> str:
    """Mirror of restore_network's ckpnt handling (rnn_class.py:191-196)."""
    if ckpnt == "latest":
        state = os.path.join(path, "checkpoint")
        if os.path.exists(state):
            with open(state) as fh:
                for line in fh:
                    if line.startswith("model_checkpoint_path:"):
                        name = line.split(":", 1)[1].strip().strip('"')
                        return name if os.path.isabs(name) else os.path.join(path, name)
        cands = [f[:-len(".index")] for f in os.listdir(path) if f.endswith(".index")]
        if not cands:
            raise ValueError("no checkpoint found in %s" % path)

        def _step(n):
            tail = n.rsplit("-", 1)[-1]
            return int(tail) if tail.isdigit() else -1
        return os.path.join(path, max(cands, key=_step))
    return path + "/" + ckpnt


def read_checkpoint(prefix: str, names: Iterable[str] = None, verify_crc: bool = True
                    ) -> "OrderedDict[str, np.ndarray]":
    """Read tensors from the bundle with the given prefix (``.../ckpnt-30000``)."""
    entries = read_index(prefix + ".index")
    wanted = list(entries) if names is None else list(names)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    shards: Dict[int, bytes] = {}
    n_shards = 1 + max([e.shard_id for e in entries.values()] or [0])
    for name in wanted:
        if name not in entries:
            raise KeyError("tensor %r not in checkpoint %s" % (name, prefix))
        e = entries[name]
        if e.dtype not in _DTYPES:
            raise ValueError("tensor %r has unsupported dtype %d" % (name, e.dtype))
        if e.shard_id not in shards:
            with open("%s.data-%05d-of-%05d" % (prefix, e.shard_id, n_shards), "rb") as fh:
                shards[e.shard_id] = fh.read()
        raw = shards[e.shard_id][e.offset:e.offset + e.size]
        if len(raw) != e.size:
            raise ValueError("tensor %r: data file truncated" % name)
        if verify_crc and mask_crc(crc32c(raw)) != e.crc32c:
            raise ValueError("tensor %r: crc32c mismatch" % name)
        arr = np.frombuffer(raw, dtype=_DTYPES[e.dtype]).reshape(e.shape)
        out[name] = arr.copy()
    return out


def is_inference_tensor(name: str) -> bool:
    """True for the 74 tensors the forward pass needs (optimizer slots skipped)."""
    return "RMSProp" not in name and "Adam" not in name and \
        not name.endswith(("beta1_power", "beta2_power"))


def read_inference_weights(path: str, ckpnt: str = "latest") -> "OrderedDict[str, np.ndarray]":
    prefix = _resolve_prefix(path, ckpnt)
    entries = read_index(prefix + ".index")
    names = [n for n in entries if is_inference_tensor(n)]
    return read_checkpoint(prefix, names)


def read_optimizer_state(path: str, ckpnt: str = "latest") -> "OrderedDict[str, np.ndarray]":
    """The non-inference entries of a bundle: optimizer slot variables (``<var>/RMSProp``, ``<var>/RMSProp_1``,
    ``<var>/Adam``, ``<var>/Adam_1``, ``.../beta{1,2}_power``) as tf.train.Saver stored them."""
    prefix = _resolve_prefix(path, ckpnt)
    entries = read_index(prefix + ".index")
    names = [n for n in entries if not is_inference_tensor(n)]
    return read_checkpoint(prefix, names) if names else OrderedDict()


# --------------------------------------------------------------------------- writer
def _entry_proto(arr: np.ndarray, offset: int) -> bytes:
    raw = arr.tobytes()
    shape = b"".join(b"\x12" + _put_varint(len(d)) + d
                     for d in (b"\x08" + _put_varint(int(s)) for s in arr.shape))
    msg = b"\x08" + _put_varint(_DTYPES_INV[arr.dtype.newbyteorder("<")])
    msg += b"\x12" + _put_varint(len(shape)) + shape
    if offset:
        msg += b"\x20" + _put_varint(offset)
    msg += b"\x28" + _put_varint(len(raw))
    msg += b"\x35" + struct.pack("<I", mask_crc(crc32c(raw)))
    return msg


def _build_block(items: List[Tuple[bytes, bytes]], restart_interval: int = 16) -> bytes:
    out = bytearray()
    restarts = []
    prev = b""
    for i, (key, val) in enumerate(items):
        if i % restart_interval == 0:
            restarts.append(len(out))
            shared = 0
        else:
            shared = 0
            while shared < min(len(prev), len(key)) and prev[shared] == key[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(val))
        out += key[shared:] + val
        prev = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_checkpoint(prefix: str, tensors: Dict[str, np.ndarray]) -> None:
    """Write a single-shard checkpoint-V2 bundle readable by TF's Saver and by read_checkpoint."""
    names = sorted(tensors, key=lambda s: s.encode("utf-8"))
    data = bytearray()
    items: List[Tuple[bytes, bytes]] = []
    # BundleHeaderProto: num_shards=1, endianness=LITTLE(0), version{producer=1}
    items.append((b"", b"\x08\x01\x1a\x02\x08\x01"))
    for name in names:
        arr = np.ascontiguousarray(tensors[name])
        if arr.dtype.newbyteorder("<") not in _DTYPES_INV:
            raise ValueError("unsupported dtype %s for %s" % (arr.dtype, name))
        arr = arr.astype(arr.dtype.newbyteorder("<"), copy=False)
        items.append((name.encode("utf-8"), _entry_proto(arr, len(data))))
        data += arr.tobytes()
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    with open(prefix + ".data-00000-of-00001", "wb") as fh:
        fh.write(bytes(data))

    def _with_trailer(block: bytes) -> bytes:
        return block + b"\x00" + struct.pack("<I", mask_crc(crc32c(block + b"\x00")))

    table = bytearray()
    data_block = _build_block(items)
    d_off, d_size = 0, len(data_block)
    table += _with_trailer(data_block)
    meta_block = _build_block([])
    m_off, m_size = len(table), len(meta_block)
    table += _with_trailer(meta_block)
    # index entry of the (single) data block: leveldb's TableBuilder::Finish stores FindShortSuccessor(last key) -- the
    # last key cut after its first byte that is not 0xff, that byte incremented ("stack_..." -> "t").  With it the
    # .index file comes out byte for byte as TensorFlow 1.10 wrote ckpnt-30000.index (tests/test_checkpoint.py).
    last_key = items[-1][0]
    cut = next((i for i, b in enumerate(last_key) if b != 0xFF), None)
    successor = last_key if cut is None else last_key[:cut] + bytes([last_key[cut] + 1])
    index_block = _build_block([(successor, _put_varint(d_off) + _put_varint(d_size))])
    i_off, i_size = len(table), len(index_block)
    table += _with_trailer(index_block)
    footer = _put_varint(m_off) + _put_varint(m_size) + _put_varint(i_off) + _put_varint(i_size)
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    table += footer
    with open(prefix + ".index", "wb") as fh:
        fh.write(bytes(table))

"""Checkpoint-V2 reader KATs: per-tensor masked CRC-32C, name/shape/offset table."""
import json
import os

import numpy as np
import pytest

from catfish_amd import checkpoint
from conftest import GOLDEN, REFERENCE, has_reference

CKPT_DIR = os.path.join(REFERENCE, "catfish", "ResNetRNN", "checkpoints")


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors
    assert checkpoint.crc32c(b"123456789") == 0xE3069283
    assert checkpoint.crc32c(bytes(32)) == 0x8A9136AA
    assert checkpoint.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    # conv1d/kernel crc 0x39d6f2c1 is stored masked as 0x88055e85
    assert checkpoint.mask_crc(0x39D6F2C1) == 0x88055E85
    assert checkpoint.unmask_crc(0x88055E85) == 0x39D6F2C1


def test_table_fixture_shape_facts():
    with open(os.path.join(GOLDEN, "ckpt_table.json")) as fh:
        table = {e["name"]: e for e in json.load(fh)}
    assert len(table) == 190
    inf = [n for n in table if checkpoint.is_inference_tensor(n)]
    assert len(inf) == 74
    assert table["batch_normalization/beta"]["offset"] == 0
    assert table["batch_normalization/gamma"]["offset"] == 384
    assert table["conv1d/bias"]["offset"] == 8192
    assert table["conv1d/kernel"]["offset"] == 8576
    assert table["conv1d_2/kernel"]["offset"] == 10112 and table["conv1d_2/kernel"]["size"] == 12288
    assert table["final_fully_connected/kernel"]["offset"] == 134924
    assert table["conv1d/kernel"]["crc32c"] == 0x88055E85
    assert sum(int(np.prod(table[n]["shape"])) for n in inf) == 197185


@pytest.mark.skipif(not has_reference(), reason="reference checkpoint not mounted")
def test_reference_bundle_matches_fixture_and_crcs(ckpt_weights):
    entries = checkpoint.read_index(os.path.join(CKPT_DIR, "ckpnt-30000.index"))
    with open(os.path.join(GOLDEN, "ckpt_table.json")) as fh:
        table = json.load(fh)
    assert [e.as_dict() for e in entries.values()] == table
    w = checkpoint.read_inference_weights(CKPT_DIR, "ckpnt-30000")     # verifies every CRC
    assert sorted(w) == sorted(ckpt_weights)
    for k in w:
        assert np.array_equal(w[k], ckpt_weights[k]), k
    # "latest" resolution (no `checkpoint` state file in the bundle -> highest step)
    w2 = checkpoint.read_inference_weights(CKPT_DIR, "latest")
    assert np.array_equal(w2["conv1d/kernel"], w["conv1d/kernel"])


def test_checkpoint_fingerprints(ckpt_weights):
    """BN moving stats are exactly 0/1, GRU gate biases sit near their init of 1."""
    for j in range(8):
        bn = "batch_normalization" if j == 0 else "batch_normalization_%d" % j
        assert np.all(ckpt_weights[bn + "/moving_mean"] == 0)
        assert np.all(ckpt_weights[bn + "/moving_variance"] == 1)
    for l in range(3):
        for d in ("fw", "bw"):
            b = ckpt_weights["stack_bidirectional_rnn/cell_%d/bidirectional_rnn/%s/gru_cell/gates/bias" % (l, d)]
            assert 0.5 < b.mean() < 1.5


def test_write_read_round_trip(tmp_path, ckpt_weights):
    prefix = str(tmp_path / "ckpts" / "ckpnt-7")
    extra = dict(ckpt_weights)
    extra["conv1d/kernel/RMSProp"] = np.zeros((1, 1, 32), np.float32)   # optimizer slot: must be skipped
    checkpoint.write_checkpoint(prefix, extra)
    entries = checkpoint.read_index(prefix + ".index")
    assert len(entries) == 75
    back = checkpoint.read_inference_weights(str(tmp_path / "ckpts"), "latest")
    assert len(back) == 74
    for k in ckpt_weights:
        assert np.array_equal(back[k], ckpt_weights[k])
    back2 = checkpoint.read_inference_weights(str(tmp_path / "ckpts"), "ckpnt-7")
    assert sorted(back2) == sorted(back)


def test_corruption_is_detected(tmp_path, ckpt_weights):
    prefix = str(tmp_path / "ckpnt-1")
    checkpoint.write_checkpoint(prefix, {"conv1d/kernel": ckpt_weights["conv1d/kernel"]})
    data = prefix + ".data-00000-of-00001"
    raw = bytearray(open(data, "rb").read())
    raw[5] ^= 0x40
    open(data, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="crc32c"):
        checkpoint.read_checkpoint(prefix)
    with pytest.raises(KeyError):
        checkpoint.read_checkpoint(prefix, ["nope"], verify_crc=False)


@pytest.mark.skipif(not has_reference(), reason="needs the reference checkpoint")
def test_writer_reproduces_the_reference_bundle_byte_for_byte(tmp_path):
    """The strongest pin the reference offers for the WRITER: ckpnt-30000.{index,data-00000-of-00001} were written by
    TensorFlow 1.10's own BundleWriter.  Reading all 190 tensors and writing them back must give the same two files, byte for
    byte -- tensor order and packing of the data file, every BundleEntryProto (dtype, shape, offset, size, masked CRC-32C),
    the leveldb block layout (prefix compression, restart array, block trailers), the index entry (leveldb's
    FindShortSuccessor of the last key) and the footer.  So a bundle written here is what tf.train.Saver would have written."""
    import hashlib
    ref = os.path.join(REFERENCE, "catfish", "ResNetRNN", "checkpoints", "ckpnt-30000")
    tensors = checkpoint.read_checkpoint(ref)
    assert len(tensors) == 190
    out = str(tmp_path / "ckpnt-30000")
    checkpoint.write_checkpoint(out, tensors)
    for ext in (".index", ".data-00000-of-00001"):
        with open(ref + ext, "rb") as a, open(out + ext, "rb") as b:
            assert a.read() == b.read(), ext
    # fingerprints of the reference's files (so a reader of this test sees what was compared)
    assert hashlib.sha256(open(out + ".index", "rb").read()).hexdigest().startswith("471de3f2a6280c50")
    assert hashlib.sha256(open(out + ".data-00000-of-00001", "rb").read()).hexdigest().startswith("8d3af93055b35b16")


def test_native_crc32c_equals_the_python_loop():
    """checkpoint.crc32c goes through the library's cf_crc32c (slicing-by-8) when it is built: same values as the byte loop on every
    length and alignment class, with and without a running CRC, and the standard check value of CRC-32C."""
    from catfish_amd import checkpoint as c
    rng = np.random.default_rng(0)
    assert c.crc32c(b"123456789") == c.crc32c_python(b"123456789") == 0xE3069283
    blob = rng.integers(0, 256, size=5000, dtype=np.uint8).tobytes()
    for n in (0, 1, 2, 7, 8, 9, 15, 16, 17, 63, 64, 65, 1000, 4999):
        for off in (0, 1, 3, 7):
            piece = blob[off:off + n]
            assert c.crc32c(piece) == c.crc32c_python(piece), (n, off)
            assert c.crc32c(piece, 0xDEADBEEF) == c.crc32c_python(piece, 0xDEADBEEF), (n, off)
    first, rest = blob[:1234], blob[1234:]
    assert c.crc32c(rest, c.crc32c(first)) == c.crc32c(blob)                # continuing a CRC equals one pass over the whole
    assert c.crc32c(bytearray(blob)) == c.crc32c(memoryview(blob)) == c.crc32c(blob)

"""LZB1 batch-store files: byte format per /root/reference/src/advanced/batch_store.rs:16-101 with
bincode 1.x default encoding, store semantics per batch.rs:36-260; scenarios follow the reference's own
tests/integration.rs:100-165 (persist_add_and_refresh, open_batch_from_disk)."""
import os
import struct

import pytest

import libzkp_amd as z
from libzkp_amd import api, batch_store as bs


@pytest.fixture
def store(tmp_path, monkeypatch):
    monkeypatch.setattr(bs, "_override", None)
    monkeypatch.delenv("LIBZKP_BATCH_DIR", raising=False)
    d = tmp_path / "store"
    z.set_batch_store_dir(d)
    yield str(d)
    bs._override = None


OPS = [("range", 5, 0, 10), ("equality", 7, 7), ("threshold", (1, 2, 3), 5), ("membership", 3, (1, 2, 3)),
       ("improvement", 1, 9), ("consistency", (4, 5, 2**64 - 1)), ("threshold", (), 0)]


def test_layout_is_bincode_default():
    blob = bs.encode_batch([("range", 5, 0, 10), ("membership", 3, (1, 2))])
    want = (b"LZB1" + struct.pack("<I", 1) + struct.pack("<Q", 2)
            + struct.pack("<I", 0) + struct.pack("<3Q", 5, 0, 10)
            + struct.pack("<I", 3) + struct.pack("<Q", 3) + struct.pack("<Q", 2) + struct.pack("<2Q", 1, 2))
    assert blob == want
    assert bs.encode_batch([]) == b"LZB1\x01\x00\x00\x00" + bytes(8)


def test_round_trip_and_rejections():
    blob = bs.encode_batch(OPS)
    assert bs.decode_batch_bytes(blob) == OPS
    assert bs.decode_batch_bytes(blob + b"\x00") == OPS                 # bincode::deserialize ignores trailing bytes
    for bad, msg in ((blob[:7], "too short"), (b"LZB2" + blob[4:], "bad magic"),
                     (blob[:4] + struct.pack("<I", 2) + blob[8:], "unsupported version 2"), (blob[:-1], "decode"),
                     (blob[:16] + struct.pack("<I", 6) + blob[20:], "invalid variant"),
                     (blob[:8] + struct.pack("<Q", 2**60), "decode")):
        with pytest.raises(ValueError, match=msg):
            bs.decode_batch_bytes(bad)


def test_persist_add_and_refresh(store):                                # integration.rs:116-137
    assert z.get_batch_store_dir() == store
    bid = z.create_proof_batch()
    path = bs.batch_file_path(store, bid)
    assert os.path.basename(path) == "batch_%016x.bin" % bid and bs.decode_batch_bytes(open(path, "rb").read()) == []
    z.batch_add_range_proof(bid, 5, 0, 10)
    assert z.get_batch_status(bid)["total_operations"] == 1
    z.refresh_batch_from_store(bid)
    assert z.get_batch_status(bid)["total_operations"] == 1
    bs.write_batch_file(store, bid, OPS)                               # another process wrote
    z.refresh_batch_from_store(bid)
    st = z.get_batch_status(bid)
    assert st["total_operations"] == len(OPS) and st["threshold_proofs"] == 2
    assert z.list_batch_ids_in_store() == [bid]
    z.clear_batch(bid)
    assert not os.path.exists(path) and z.list_batch_ids_in_store() == []
    with pytest.raises(ValueError, match="not loaded"):
        z.refresh_batch_from_store(bid)


def test_open_batch_from_disk(store):                                   # integration.rs:139-158
    bs.write_batch_file(store, 0xDEADBEEFCAFE, [("range", 7, 1, 20)])
    assert os.listdir(store) == ["batch_0000deadbeefcafe.bin"]           # the temp file was renamed away
    z.open_batch_from_store(0xDEADBEEFCAFE)
    st = z.get_batch_status(0xDEADBEEFCAFE)
    assert st["total_operations"] == 1 and st["range_proofs"] == 1
    with pytest.raises(ValueError, match="already open"):
        z.open_batch_from_store(0xDEADBEEFCAFE)
    with pytest.raises(bs.StorageError):
        z.open_batch_from_store(12345)
    open(os.path.join(store, "batch_zz.bin"), "wb").close()
    open(os.path.join(store, "other.bin"), "wb").close()
    assert z.list_batch_ids_in_store() == [0xDEADBEEFCAFE]
    z.clear_batch(0xDEADBEEFCAFE)


def test_export_import(store, tmp_path):
    bid = z.create_proof_batch()
    z.batch_add_threshold_proof(bid, [10, 20, 30], 50)
    z.batch_add_consistency_proof(bid, [1, 2, 3])
    dest = tmp_path / "out" / "backup.lzb"
    z.export_batch_to_file(bid, str(dest))
    assert sorted(os.listdir(dest.parent)) == ["backup.lzb"]
    new = z.import_batch_from_file(str(dest))
    assert new != bid and api._registry[new] == api._registry[bid]
    assert sorted(z.list_batch_ids_in_store()) == sorted([bid, new])
    z.clear_batch(bid)
    z.clear_batch(new)
    with pytest.raises(ValueError, match="Invalid batch ID"):
        z.export_batch_to_file(bid, str(dest))


def test_unconfigured_store(monkeypatch, tmp_path):
    monkeypatch.setattr(bs, "_override", None)
    monkeypatch.delenv("LIBZKP_BATCH_DIR", raising=False)
    assert z.get_batch_store_dir() is None
    bid = z.create_proof_batch()                                        # in-memory only
    with pytest.raises(bs.ConfigError):
        z.refresh_batch_from_store(bid)
    with pytest.raises(bs.ConfigError):
        z.list_batch_ids_in_store()
    monkeypatch.setenv("LIBZKP_BATCH_DIR", str(tmp_path))
    assert z.get_batch_store_dir() == str(tmp_path)
    z.batch_add_improvement_proof(bid, 1, 2)
    assert z.list_batch_ids_in_store() == [bid]
    z.clear_batch(bid)

"""Disk persistence of proof batches in the reference's `LZB1` file format (host-side byte format).

Follows /root/reference/src/advanced/batch_store.rs:16-230.  A file is

    "LZB1" | u32 LE version = 1 | bincode 1.x (default options) of struct { operations: Vec<BatchOperation> }

and bincode's default encoding of that payload is: u64 LE element count, then per operation a u32 LE variant index in
declaration order of `BatchOperation` (/root/reference/src/utils/composition.rs:343-350) followed by its fields in
order — u64 as 8 bytes LE, Vec<u64> as a u64 LE count and the elements.  The file name is `batch_{id:016x}.bin`; writes
go to `.batch_{id:016x}.tmp` under an exclusive lock and are renamed into place, reads take a shared lock
(batch_store.rs:104-140).

The operations are the tuples `api.batch_add_*` keeps in the registry: ("range", v, min, max), ("equality", a, b),
("threshold", values, t), ("membership", v, set), ("improvement", old, new), ("consistency", data).
"""
import fcntl
import os
import struct
import threading

FILE_MAGIC = b"LZB1"                  # batch_store.rs:16
FORMAT_VERSION = 1                    # batch_store.rs:18
_VARIANTS = ("range", "equality", "threshold", "membership", "improvement", "consistency")   # composition.rs:343-350

_override = None
_override_lock = threading.Lock()


class StorageError(OSError):
    """ZkpError::StorageError"""


class ConfigError(RuntimeError):
    """ZkpError::ConfigError"""


def set_batch_store_dir(path):
    """batch_store.rs:28-37: creates the directory; takes precedence over LIBZKP_BATCH_DIR."""
    global _override
    path = os.fspath(path)
    try:
        os.makedirs(path, exist_ok=True)
    except OSError as e:
        raise StorageError("create batch store directory: %s" % e)
    with _override_lock:
        _override = path


def get_batch_store_dir():
    """batch_store.rs:40-47"""
    with _override_lock:
        if _override is not None:
            return _override
    return os.environ.get("LIBZKP_BATCH_DIR")


def _store_dir_required():
    d = get_batch_store_dir()
    if d is None:
        raise ConfigError("batch store not configured: set_batch_store_dir or LIBZKP_BATCH_DIR")
    return d


def batch_file_path(directory, batch_id):
    return os.path.join(directory, "batch_%016x.bin" % batch_id)


def _u64s(xs):
    return struct.pack("<Q%dQ" % len(xs), len(xs), *xs)


def encode_batch(ops):
    """batch_store.rs:62-73"""
    out = [FILE_MAGIC, struct.pack("<I", FORMAT_VERSION), struct.pack("<Q", len(ops))]
    for op in ops:
        kind = op[0]
        out.append(struct.pack("<I", _VARIANTS.index(kind)))
        if kind == "range":
            out.append(struct.pack("<3Q", op[1], op[2], op[3]))
        elif kind in ("equality", "improvement"):
            out.append(struct.pack("<2Q", op[1], op[2]))
        elif kind == "threshold":
            out.append(_u64s(op[1]) + struct.pack("<Q", op[2]))
        elif kind == "membership":
            out.append(struct.pack("<Q", op[1]) + _u64s(op[2]))
        else:
            out.append(_u64s(op[1]))
    return b"".join(out)


class _Reader:
    def __init__(self, data):
        self.data, self.pos = data, 0

    def u(self, fmt):
        size = struct.calcsize(fmt)
        if self.pos + size > len(self.data):
            raise ValueError("batch file decode: unexpected end of file")
        v = struct.unpack_from(fmt, self.data, self.pos)
        self.pos += size
        return v

    def vec(self):
        (n,) = self.u("<Q")
        if n > (len(self.data) - self.pos) // 8:
            raise ValueError("batch file decode: unexpected end of file")
        return tuple(self.u("<%dQ" % n)) if n else ()


def decode_batch_bytes(data):
    """batch_store.rs:75-101 (bincode's default deserializer ignores trailing bytes)."""
    data = bytes(data)
    if len(data) < 8:
        raise ValueError("batch file too short")
    if data[:4] != FILE_MAGIC:
        raise ValueError("batch file: bad magic")
    (ver,) = struct.unpack_from("<I", data, 4)
    if ver != FORMAT_VERSION:
        raise ValueError("batch file: unsupported version %d" % ver)
    r = _Reader(data)
    r.pos = 8
    (n,) = r.u("<Q")
    ops = []
    for _ in range(n):
        (tag,) = r.u("<I")
        if tag >= len(_VARIANTS):
            raise ValueError("batch file decode: invalid variant %d" % tag)
        kind = _VARIANTS[tag]
        if kind == "range":
            ops.append((kind,) + r.u("<3Q"))
        elif kind in ("equality", "improvement"):
            ops.append((kind,) + r.u("<2Q"))
        elif kind == "threshold":
            values = r.vec()
            ops.append((kind, values, r.u("<Q")[0]))
        elif kind == "membership":
            (value,) = r.u("<Q")
            ops.append((kind, value, r.vec()))
        else:
            ops.append((kind, r.vec()))
    return ops


def _write_atomic(tmp_path, final_path, data, what):
    try:
        fd = os.open(tmp_path, os.O_CREAT | os.O_WRONLY | os.O_TRUNC, 0o644)
        try:
            fcntl.flock(fd, fcntl.LOCK_EX)
            view = memoryview(data)
            while view:
                view = view[os.write(fd, view):]
            os.fsync(fd)
        finally:
            os.close(fd)                                   # drops the lock
        os.replace(tmp_path, final_path)
    except OSError as e:
        raise StorageError("%s: %s" % (what, e))


def _read_locked(path, what):
    try:
        with open(path, "rb") as f:
            fcntl.flock(f.fileno(), fcntl.LOCK_SH)
            return f.read()
    except OSError as e:
        raise StorageError("%s: %s" % (what, e))


def write_batch_file(directory, batch_id, ops):
    """batch_store.rs:104-127"""
    _write_atomic(os.path.join(directory, ".batch_%016x.tmp" % batch_id), batch_file_path(directory, batch_id),
                  encode_batch(ops), "write batch file")


def read_batch_file(directory, batch_id):
    """batch_store.rs:130-141"""
    return decode_batch_bytes(_read_locked(batch_file_path(directory, batch_id), "open batch file"))


def delete_batch_file_if_configured(batch_id):
    """batch_store.rs:144-154"""
    d = get_batch_store_dir()
    if d is None:
        return
    path = batch_file_path(d, batch_id)
    if os.path.exists(path):
        try:
            os.remove(path)
        except OSError as e:
            raise StorageError("remove batch file: %s" % e)


def persist_batch_if_configured(batch_id, ops):
    """batch_store.rs:157-162"""
    d = get_batch_store_dir()
    if d is not None:
        write_batch_file(d, batch_id, ops)


def list_batch_ids_in_store():
    """batch_store.rs:165-187"""
    d = _store_dir_required()
    ids = []
    try:
        names = os.listdir(d)
    except OSError as e:
        raise StorageError("read batch store: %s" % e)
    for name in names:
        if not (name.startswith("batch_") and name.endswith(".bin")):
            continue
        hexpart = name[len("batch_"):-len(".bin")]
        body = hexpart[1:] if hexpart[:1] == "+" else hexpart        # u64::from_str_radix accepts a leading '+'
        if not body or any(c not in "0123456789abcdefABCDEF" for c in body):
            continue
        v = int(body, 16)
        if v < 1 << 64:
            ids.append(v)
    return sorted(ids)


def export_proof_batch_to_path(ops, path):
    """batch_store.rs:190-215: temp file is `path` with its extension replaced by "tmp"."""
    path = os.fspath(path)
    parent = os.path.dirname(path)
    if parent:
        try:
            os.makedirs(parent, exist_ok=True)
        except OSError as e:
            raise StorageError("create export parent: %s" % e)
    head, tail = os.path.split(path)
    stem = tail.rsplit(".", 1)[0] if "." in tail.lstrip(".") else tail
    _write_atomic(os.path.join(head, stem + ".tmp"), path, encode_batch(ops), "write export")


def import_proof_batch_from_path(path):
    """batch_store.rs:218-229"""
    return decode_batch_bytes(_read_locked(os.fspath(path), "open import"))

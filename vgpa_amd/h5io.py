"""
Result files in the reference's on-disk format (src/var_bayes/simulation.py:269-347): one HDF5 file, one dataset per
key of the output dictionary (`at, bt, fx, m0, s0, mt, st, lamt, psit, Efx, Edf`), scalars stored as shape-(1,) arrays.

h5py is not part of the GPU image, so this module carries a small self-contained HDF5 subset:

  * `save_h5(path, data)` writes a version-0 superblock, an old-style root group (v1 B-tree + local heap + one symbol
    table node) and one contiguous little-endian dataset per key (float64 / float32 / int64 / int32 / uint8).  Any
    HDF5 reader (h5py, MATLAB, h5dump) opens these files; they are not gzip-compressed like the reference's.
  * `load_h5(path)` uses h5py when it is importable and otherwise parses the files this writer produces as well as the
    files h5py's default settings produce for `create_dataset(key, data=..., compression='gzip')` -- i.e. the
    reference's own result files: old-style groups, version-1 object headers with continuation blocks, contiguous or
    chunked (v1 B-tree) layouts, deflate and shuffle filters.

Format reference: "HDF5 File Format Specification Version 2.0" (superblock 0, object header 1, layout 3).
"""
import struct
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIGNATURE = b"\x89HDF\r\n\x1a\n"
LEAF_K, INTERNAL_K = 64, 16          # symbols per node = 2 * LEAF_K: one node holds every key of a result file

_DTYPES = {np.dtype("<f8"), np.dtype("<f4"), np.dtype("<i8"), np.dtype("<i4"), np.dtype("u1")}


def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _datatype_message(dt):
    """Datatype message (type 0x0003), version 1."""
    dt = np.dtype(dt)
    if dt.kind == "f":
        exp_bits, man_bits = (11, 52) if dt.itemsize == 8 else (8, 23)
        bias = (1 << (exp_bits - 1)) - 1
        head = struct.pack("<BBBBI", 0x11, 0x20, 8 * dt.itemsize - 1, 0, dt.itemsize)   # IEEE, little endian, implied msb
        prop = struct.pack("<HHBBBBI", 0, 8 * dt.itemsize, man_bits, exp_bits, 0, man_bits, bias)
        return head + prop
    signed = 0x08 if dt.kind == "i" else 0x00
    return struct.pack("<BBBBI", 0x10, signed, 0, 0, dt.itemsize) + struct.pack("<HH", 0, 8 * dt.itemsize)


def _message(mtype, body):
    body = _pad8(body)
    return struct.pack("<HHB3x", mtype, len(body), 0) + body


def _object_header(messages):
    blob = b"".join(messages)
    return struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(blob)) + blob


def save_h5(path, data):
    """Writes `data` (dict: name -> array-like or scalar) as one HDF5 file with one contiguous dataset per key."""
    items = []
    for key in data:
        val = data[key]
        arr = np.atleast_1d(val) if np.isscalar(val) or np.ndim(val) == 0 else np.asarray(val)
        if arr.dtype.kind == "f" and arr.dtype.itemsize not in (4, 8):
            arr = arr.astype("<f8")
        elif arr.dtype.kind == "b":
            arr = arr.astype("u1")
        elif arr.dtype.kind in "iu" and arr.dtype.itemsize not in (4, 8) and arr.dtype != np.dtype("u1"):
            arr = arr.astype("<i8")
        elif arr.dtype.kind == "u" and arr.dtype != np.dtype("u1"):
            arr = arr.astype("<i8")
        arr = np.ascontiguousarray(arr.astype(arr.dtype.newbyteorder("<")))
        if arr.dtype not in _DTYPES:
            raise TypeError(f"save_h5: unsupported dtype {arr.dtype} for key '{key}'")
        name = str(key).encode("ascii")
        items.append((name, arr))
    if len(items) > 2 * LEAF_K:
        raise ValueError(f"save_h5: at most {2 * LEAF_K} datasets per file")
    items.sort(key=lambda kv: kv[0])                     # symbol table nodes are ordered by name

    # ---- local heap data: offset 0 holds the empty string, then the names (null terminated, 8-byte padded)
    heap = bytearray(8)
    name_off = []
    for name, _ in items:
        name_off.append(len(heap))
        heap += _pad8(name + b"\0")
    # the library wants room for a free block; keep one 16-byte free block at the end
    free_off = len(heap)
    heap += struct.pack("<QQ", 1, 16)                    # next free = H5HL_FREE_NULL (1), size 16

    # ---- fixed positions
    pos = 96                                             # after the superblock
    root_hdr_addr = pos
    root_hdr_size = 16 + 8 + 16
    pos += root_hdr_size
    btree_addr = pos
    btree_size = 24 + (2 * INTERNAL_K + 1) * 8 + 2 * INTERNAL_K * 8
    pos += btree_size
    heap_hdr_addr = pos
    pos += 32
    heap_data_addr = pos
    pos += len(heap)
    snod_addr = pos
    snod_size = 8 + 2 * LEAF_K * 40
    pos += snod_size

    # ---- datasets: object header, then raw data (8-byte aligned)
    hdr_addrs, data_addrs, headers = [], [], []
    for name, arr in items:
        space = struct.pack("<BBB5x", 1, arr.ndim, 0) + b"".join(struct.pack("<Q", int(n)) for n in arr.shape)
        fill = struct.pack("<BBBB", 2, 1, 0, 0)          # v2: early allocation, fill at allocation, no fill value
        hdr_len = 16 + sum(8 + len(_pad8(b)) for b in (space, _datatype_message(arr.dtype), fill, b"\0" * 18))
        hdr_addrs.append(pos)
        pos += hdr_len
        data_addrs.append(pos)
        layout = struct.pack("<BBQQ", 3, 1, pos, arr.nbytes)
        headers.append(_object_header([_message(0x0001, space), _message(0x0003, _datatype_message(arr.dtype)),
                                       _message(0x0005, fill), _message(0x0008, layout)]))
        assert len(headers[-1]) == hdr_len
        pos += arr.nbytes + (-arr.nbytes % 8)
    eof = pos

    with open(path, "wb") as fh:
        root_entry = struct.pack("<QQII", 0, root_hdr_addr, 1, 0) + struct.pack("<QQ", btree_addr, heap_hdr_addr)
        fh.write(SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, LEAF_K, INTERNAL_K, 0) +
                 struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF) + root_entry)
        fh.write(_object_header([_message(0x0011, struct.pack("<QQ", btree_addr, heap_hdr_addr))]))
        node = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if items else 0, UNDEF, UNDEF)
        if items:
            node += struct.pack("<QQQ", 0, snod_addr, name_off[-1])
        fh.write(node + b"\0" * (btree_size - len(node)))
        fh.write(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), free_off, heap_data_addr))
        fh.write(bytes(heap))
        snod = b"SNOD" + struct.pack("<BBH", 1, 0, len(items))
        for off, addr in zip(name_off, hdr_addrs):
            snod += struct.pack("<QQII16x", off, addr, 0, 0)
        fh.write(snod + b"\0" * (snod_size - len(snod)))
        for (name, arr), hdr, daddr in zip(items, headers, data_addrs):
            fh.write(hdr)
            assert fh.tell() == daddr
            fh.write(arr.tobytes())
            fh.write(b"\0" * (-arr.nbytes % 8))
        assert fh.tell() == eof


# ---------------------------------------------------------------------------------------------------------------
#  reader
# ---------------------------------------------------------------------------------------------------------------
class _Reader:
    def __init__(self, buf):
        self.b = buf
        if buf[:8] != SIGNATURE:
            raise ValueError("not an HDF5 file (signature at offset 0 expected)")
        ver = buf[8]
        if ver not in (0, 1):
            raise NotImplementedError(f"HDF5 superblock version {ver}: install h5py to read this file")
        if buf[13] != 8 or buf[14] != 8:
            raise NotImplementedError("only 8-byte offsets / lengths are supported")
        off = 24 + (4 if ver == 1 else 0)
        self.base = struct.unpack_from("<Q", buf, off)[0]
        entry = off + 32
        self.root_header = struct.unpack_from("<Q", buf, entry + 8)[0]

    # ---- object headers (version 1, with continuation blocks)
    def messages(self, addr):
        b = self.b
        ver, _, nmsg, _, size = struct.unpack_from("<BBHII", b, addr)
        if ver != 1:
            raise NotImplementedError("only version-1 object headers are supported: install h5py to read this file")
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            pos, left = blocks.pop(0)
            end = pos + left
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize, _ = struct.unpack_from("<HHB", b, pos)
                body = b[pos + 8:pos + 8 + msize]
                pos += 8 + msize
                if mtype == 0x0010:
                    blocks.append(struct.unpack_from("<QQ", body, 0))
                out.append((mtype, body))
        return out

    # ---- old-style groups
    def _heap_string(self, heap_addr, off):
        assert self.b[heap_addr:heap_addr + 4] == b"HEAP"
        data_addr = struct.unpack_from("<Q", self.b, heap_addr + 24)[0]
        end = self.b.index(b"\0", data_addr + off)
        return self.b[data_addr + off:end].decode("ascii")

    def _group_nodes(self, addr, heap_addr, out):
        b = self.b
        if b[addr:addr + 4] == b"SNOD":
            n = struct.unpack_from("<H", b, addr + 6)[0]
            for i in range(n):
                name_off, hdr = struct.unpack_from("<QQ", b, addr + 8 + 40 * i)
                out[self._heap_string(heap_addr, name_off)] = hdr
            return
        assert b[addr:addr + 4] == b"TREE"
        _, level, used = struct.unpack_from("<BBH", b, addr + 4)
        for i in range(used):
            child = struct.unpack_from("<Q", b, addr + 24 + 8 + 16 * i)[0]
            self._group_nodes(child, heap_addr, out)

    def root_links(self):
        for mtype, body in self.messages(self.root_header):
            if mtype == 0x0011:
                btree, heap = struct.unpack_from("<QQ", body, 0)
                out = {}
                self._group_nodes(btree, heap, out)
                return out
        raise NotImplementedError("root group without a symbol table (new-style group): install h5py to read this file")

    # ---- datasets
    @staticmethod
    def _dtype(body):
        cls, ver = body[0] & 0x0F, body[0] >> 4
        size = struct.unpack_from("<I", body, 4)[0]
        order = ">" if body[1] & 1 else "<"
        if cls == 1:
            return np.dtype(f"{order}f{size}")
        if cls == 0:
            return np.dtype(f"{order}{'i' if body[1] & 0x08 else 'u'}{size}")
        raise NotImplementedError(f"datatype class {cls} (version {ver}): install h5py to read this file")

    def _chunks(self, addr, rank, out):
        b = self.b
        assert b[addr:addr + 4] == b"TREE"
        ntype, level, used = struct.unpack_from("<BBH", b, addr + 4)
        assert ntype == 1
        key = 8 + 8 * (rank + 1)
        pos = addr + 24
        for _ in range(used):
            nbytes, mask = struct.unpack_from("<II", b, pos)
            offs = struct.unpack_from(f"<{rank + 1}Q", b, pos + 8)
            child = struct.unpack_from("<Q", b, pos + key)[0]
            if level == 0:
                out.append((offs[:rank], nbytes, mask, child))
            else:
                self._chunks(child, rank, out)
            pos += key + 8

    def dataset(self, addr):
        shape = dtype = layout = None
        filters = []
        for mtype, body in self.messages(addr):
            if mtype == 0x0001:
                ver, rank = body[0], body[1]
                start = 8 if ver == 1 else 4
                shape = struct.unpack_from(f"<{rank}Q", body, start) if rank else ()
            elif mtype == 0x0003:
                dtype = self._dtype(body)
            elif mtype == 0x0008:
                layout = body
            elif mtype == 0x000B:
                ver, nf = body[0], body[1]
                pos = 8 if ver == 1 else 2
                for _ in range(nf):
                    fid = struct.unpack_from("<H", body, pos)[0]
                    if ver == 1 or fid >= 256:
                        nlen, _, ncd = struct.unpack_from("<HHH", body, pos + 2)
                        pos += 8 + nlen + (-nlen % 8 if ver == 1 else 0)
                    else:
                        _, ncd = struct.unpack_from("<HH", body, pos + 2)
                        pos += 6
                    cd = struct.unpack_from(f"<{ncd}I", body, pos)
                    pos += 4 * ncd + (4 if (ver == 1 and ncd % 2) else 0)
                    filters.append((fid, cd))
        if shape is None or dtype is None or layout is None:
            return None                                   # not a dataset (e.g. a sub-group)
        if layout[0] != 3:
            raise NotImplementedError(f"data layout message version {layout[0]}: install h5py to read this file")
        cls = layout[1]
        count = int(np.prod(shape)) if shape else 1
        if cls == 1:
            daddr, _ = struct.unpack_from("<QQ", layout, 2)
            if daddr == UNDEF:
                return np.zeros(shape, dtype=dtype)
            return np.frombuffer(self.b, dtype=dtype, count=count, offset=daddr).reshape(shape).copy()
        if cls == 0:
            size = struct.unpack_from("<H", layout, 2)[0]
            return np.frombuffer(layout[4:4 + size], dtype=dtype, count=count).reshape(shape).copy()
        if cls != 2:
            raise NotImplementedError(f"layout class {cls}")
        rank = layout[2] - 1
        btree = struct.unpack_from("<Q", layout, 3)[0]
        cdims = struct.unpack_from(f"<{rank + 1}I", layout, 11)[:rank]
        out = np.zeros(shape, dtype=dtype)
        if btree == UNDEF:
            return out
        chunks = []
        self._chunks(btree, rank, chunks)
        for offs, nbytes, mask, caddr in chunks:
            raw = bytes(self.b[caddr:caddr + nbytes])
            for k, (fid, cd) in reversed(list(enumerate(filters))):
                if mask & (1 << k):
                    continue
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:                                 # shuffle: bytes were grouped by significance
                    es = cd[0] if cd else dtype.itemsize
                    raw = np.frombuffer(raw, dtype=np.uint8).reshape(es, -1).T.tobytes()
                elif fid == 3:                                 # fletcher32 checksum trails the chunk
                    raw = raw[:-4]
                else:
                    raise NotImplementedError(f"HDF5 filter {fid}: install h5py to read this file")
            block = np.frombuffer(raw, dtype=dtype, count=int(np.prod(cdims))).reshape(cdims)
            sel = tuple(slice(o, min(o + c, n)) for o, c, n in zip(offs, cdims, shape))
            out[sel] = block[tuple(slice(0, s.stop - s.start) for s in sel)]
        return out


def load_h5(path):
    """Dictionary name -> numpy array of every dataset in the root group of `path`."""
    try:
        import h5py                                           # noqa: F401  (not in the GPU image; used when present)
    except ImportError:
        h5py = None
    if h5py is not None:
        with h5py.File(path, "r") as fh:
            return {key: np.array(fh[key]) for key in fh}
    with open(path, "rb") as fh:
        rd = _Reader(fh.read())
    out = {}
    for name, hdr in rd.root_links().items():
        arr = rd.dataset(hdr)
        if arr is not None:
            out[name] = arr
    return out


def result_dict(vgp, x, fx):
    """The dictionary a finished optimisation leaves behind, in the key set of the reference's result files
    (simulation.py:290-307): the optimal (A_t, b_t) split out of `x`, the minimum `fx`, and `vgp.arg_out`
    (m0, s0, mt, st, lamt, psit, Efx, Edf) of the state at `x`."""
    x = np.asarray(x, dtype=float)
    n, d = vgp.dim_n, vgp.dim_d
    out = {"fx": fx}
    if vgp.model.single_dim:
        out["at"], out["bt"] = x[:n], x[n:]
    else:
        out["at"], out["bt"] = x[:n * d * d].reshape(n, d, d), x[n * d * d:].reshape(n, d)
    vgp.free_energy(x)                       # the state at the returned x, whatever the optimiser evaluated last
    out.update(vgp.arg_out)
    return out


def save_results(name, vgp, x, fx):
    """`<name>.h5` (spaces -> underscores) with one dataset per key of `result_dict`; returns (path, dictionary)."""
    from pathlib import Path
    out = result_dict(vgp, x, fx)
    path = Path(str(name).strip().replace(" ", "_") + ".h5")
    save_h5(path, out)
    return path, out


def load_results(filename):
    """Every dataset of a result file -- one written here or one written by the reference (simulation.py:314-347)."""
    if filename is None:
        raise RuntimeError(" load_results: no input file is given.")
    return load_h5(filename)

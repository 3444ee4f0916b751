"""Dependency-free reader/writer for the subset of HDF5 that Keras 2.3 / h5py use for
the Rater's model files (ocrd_keraslm/lib/rating.py:918-974):
superblock version 0, old-style groups (B-tree v1 + symbol-table nodes + local heap;
link messages of new-style compact groups are understood too), version-1 object
headers with continuation blocks, contiguous / compact / (uncompressed) chunked
dataset layouts, attributes, and the datatypes that occur: little-endian integers
and floats, fixed-length strings, numpy-bool enums, variable-length strings through
the global heap.  Written from the HDF5 File Format Specification, version 1.1/2.0.

The main interpreter of this image has no h5py, and the reference's published model
(`model_dta_full.h5`) is a Keras HDF5 file: this module is what lets `Rater.load_config`
/ `load_weights` / `save` work there.  Files written here are verified readable by
h5py, and files written by h5py through the reference's own `Rater.save` are the
fixtures of tests/test_h5lite.py.
"""
from __future__ import annotations

import struct

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(Exception):
    pass


# ------------------------------------------------------------------------- reader
class _Datatype(object):
    def __init__(self, cls, size, dtype=None, vlen_string=False, enum_bool=False, strlen=0):
        self.cls, self.size, self.dtype = cls, size, dtype
        self.vlen_string, self.enum_bool, self.strlen = vlen_string, enum_bool, strlen


class H5File(object):
    def __init__(self, filename):
        with open(filename, "rb") as f:
            self.buf = f.read()
        if self.buf[:8] != SIGNATURE:
            raise H5Error("%s is not an HDF5 file" % filename)
        version = self.buf[8]
        if version not in (0, 1):
            raise H5Error("unsupported superblock version %d" % version)
        self.so, self.sl = self.buf[13], self.buf[14]
        if (self.so, self.sl) != (8, 8):
            raise H5Error("unsupported offset/length sizes %d/%d" % (self.so, self.sl))
        pos = 24 + (4 if version == 1 else 0)
        self.base = self._u64(pos)
        # root symbol table entry follows base, free-space, eof, driver addresses
        ste = pos + 32
        self.root = self._u64(ste + 8)
        self._cache = {}

    # -- primitive access
    def _u8(self, p):
        return self.buf[p]

    def _u16(self, p):
        return struct.unpack_from("<H", self.buf, p)[0]

    def _u32(self, p):
        return struct.unpack_from("<I", self.buf, p)[0]

    def _u64(self, p):
        return struct.unpack_from("<Q", self.buf, p)[0]

    # -- object headers
    def _messages(self, addr):
        """list of (type, flags, data-offset, size) of an object header (v1 or v2)"""
        if addr in self._cache:
            return self._cache[addr]
        buf = self.buf
        out = []
        if buf[addr:addr + 4] == b"OHDR":
            out = self._messages_v2(addr)
        else:
            if buf[addr] != 1:
                raise H5Error("unsupported object header version %d" % buf[addr])
            nmsg = self._u16(addr + 2)
            size = self._u32(addr + 8)
            blocks = [(addr + 16, size)]
            while blocks and len(out) < nmsg:
                p, remaining = blocks.pop(0)
                end = p + remaining
                while p + 8 <= end and len(out) < nmsg:
                    mtype, msize, flags = self._u16(p), self._u16(p + 2), buf[p + 4]
                    data = p + 8
                    if mtype == 0x10:
                        blocks.append((self._u64(data), self._u64(data + 8)))
                    out.append((mtype, flags, data, msize))
                    p = data + msize
        self._cache[addr] = out
        return out

    def _messages_v2(self, addr):
        buf = self.buf
        flags = buf[addr + 5]
        p = addr + 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        size_bytes = 1 << (flags & 3)
        chunk0 = int.from_bytes(buf[p:p + size_bytes], "little")
        p += size_bytes
        out = []
        blocks = [(p, chunk0)]
        track = bool(flags & 0x04)
        while blocks:
            p, remaining = blocks.pop(0)
            end = p + remaining - 4        # checksum at the end of each chunk
            while p + 4 <= end:
                mtype, msize, mflags = buf[p], self._u16(p + 1), buf[p + 3]
                data = p + 4 + (2 if track else 0)
                if mtype == 0x10:
                    caddr, clen = self._u64(data), self._u64(data + 8)
                    blocks.append((caddr + 4, clen - 4))   # skip "OCHK"
                out.append((mtype, mflags, data, msize))
                p = data + msize
        return out

    # -- groups
    def _children(self, addr):
        """name -> object header address"""
        out = {}
        for mtype, _f, data, _s in self._messages(addr):
            if mtype == 0x11:           # symbol table: B-tree + local heap
                btree, heap = self._u64(data), self._u64(data + 8)
                heap_data = self._u64(heap + 24)
                self._walk_btree(btree, heap_data, out)
            elif mtype == 0x06:         # link message (new-style compact group)
                name, target = self._link(data)
                if target is not None:
                    out[name] = target
            elif mtype == 0x02:
                fheap = self._u64(data + 2 + (8 if self.buf[data + 1] & 1 else 0))
                if fheap != UNDEF:
                    raise H5Error("dense link storage (fractal heap) is not supported")
        return out

    def _link(self, p):
        buf = self.buf
        flags = buf[p + 1]
        p += 2
        ltype = 0
        if flags & 0x08:
            ltype = buf[p]; p += 1
        if flags & 0x04:
            p += 8
        if flags & 0x10:
            p += 1
        nlen_size = 1 << (flags & 3)
        nlen = int.from_bytes(buf[p:p + nlen_size], "little")
        p += nlen_size
        name = buf[p:p + nlen].decode("utf8")
        p += nlen
        return name, (self._u64(p) if ltype == 0 else None)

    def _cstr(self, p):
        end = self.buf.index(b"\0", p)
        return self.buf[p:end].decode("utf8")

    def _walk_btree(self, addr, heap_data, out):
        buf = self.buf
        if buf[addr:addr + 4] != b"TREE":
            raise H5Error("bad B-tree node")
        level, used = buf[addr + 5], self._u16(addr + 6)
        p = addr + 8 + 16
        for i in range(used):
            child = self._u64(p + 8)
            if level > 0:
                self._walk_btree(child, heap_data, out)
            else:
                if buf[child:child + 4] != b"SNOD":
                    raise H5Error("bad symbol table node")
                n = self._u16(child + 6)
                q = child + 8
                for _ in range(n):
                    out[self._cstr(heap_data + self._u64(q))] = self._u64(q + 8)
                    q += 40
            p += 16

    def _resolve(self, path):
        addr = self.root
        for part in [x for x in path.split("/") if x]:
            kids = self._children(addr)
            if part not in kids:
                raise KeyError(path)
            addr = kids[part]
        return addr

    def keys(self, path="/"):
        return sorted(self._children(self._resolve(path)))

    def __contains__(self, path):
        try:
            self._resolve(path)
            return True
        except KeyError:
            return False

    # -- datatypes / dataspaces
    def _datatype(self, p):
        buf = self.buf
        cls, version = buf[p] & 0x0F, buf[p] >> 4
        bits = buf[p + 1] | (buf[p + 2] << 8) | (buf[p + 3] << 16)
        size = self._u32(p + 4)
        if cls == 0:
            if bits & 1:
                raise H5Error("big-endian integers are not supported")
            return _Datatype(cls, size, np.dtype("<%s%d" % ("i" if bits & 8 else "u", size)))
        if cls == 1:
            if bits & 1:
                raise H5Error("big-endian floats are not supported")
            return _Datatype(cls, size, np.dtype("<f%d" % size))
        if cls == 3:
            return _Datatype(cls, size, np.dtype("S%d" % size), strlen=size)
        if cls == 8:      # enum: h5py stores numpy bool as int8 enum {FALSE, TRUE}
            base = self._datatype(p + 8)
            return _Datatype(cls, size, base.dtype, enum_bool=(size == 1))
        if cls == 9:
            if (bits & 0x0F) == 1:
                return _Datatype(cls, size, None, vlen_string=True)
            raise H5Error("variable-length sequences are not supported")
        raise H5Error("unsupported datatype class %d (version %d)" % (cls, version))

    def _dataspace(self, p):
        buf = self.buf
        version, rank = buf[p], buf[p + 1]
        if version == 1:
            q = p + 8
        elif version == 2:
            if buf[p + 3] == 2:      # null dataspace
                return None
            q = p + 4
        else:
            raise H5Error("unsupported dataspace version %d" % version)
        return tuple(self._u64(q + 8 * i) for i in range(rank))

    def _vlen_string(self, p):
        length, gaddr, index = self._u32(p), self._u64(p + 4), self._u32(p + 12)
        if gaddr in (0, UNDEF) or length == 0:
            return ""
        buf = self.buf
        if buf[gaddr:gaddr + 4] != b"GCOL":
            raise H5Error("bad global heap collection")
        end = gaddr + self._u64(gaddr + 8)
        q = gaddr + 16
        while q + 16 <= end:
            idx, osize = self._u16(q), self._u64(q + 8)
            if idx == 0:
                break
            if idx == index:
                return buf[q + 16:q + 16 + length].decode("utf8")
            q += 16 + ((osize + 7) & ~7)
        raise H5Error("global heap object %d not found" % index)

    def _decode(self, raw_off, nbytes, dt, shape, raw=None):
        buf = self.buf if raw is None else raw
        n = int(np.prod(shape)) if shape else 1
        if dt.vlen_string:
            if raw is not None:
                raise H5Error("variable-length data in compact storage is not supported")
            items = [self._vlen_string(raw_off + 16 * i) for i in range(n)]
            return items[0] if not shape else np.array(items, dtype=object).reshape(shape)
        arr = np.frombuffer(buf, dtype=dt.dtype, count=n, offset=raw_off).reshape(shape)
        if dt.enum_bool:
            arr = arr.astype(bool)
        if not shape:
            return arr[()] if dt.cls != 3 else bytes(arr[()])
        return arr.copy()

    def read(self, path):
        """dataset -> numpy array / scalar / str (vlen string) / bytes (fixed string)"""
        addr = self._resolve(path)
        dt = shape = None
        layout = None
        for mtype, _f, data, size in self._messages(addr):
            if mtype == 0x03:
                dt = self._datatype(data)
            elif mtype == 0x01:
                shape = self._dataspace(data)
            elif mtype == 0x08:
                layout = (data, size)
        if dt is None or layout is None:
            raise H5Error("%s is not a dataset" % path)
        if shape is None:
            return None
        buf = self.buf
        p = layout[0]
        version = buf[p]
        nbytes = (int(np.prod(shape)) if shape else 1) * dt.size
        if version == 3:
            lclass = buf[p + 1]
            if lclass == 0:
                return self._decode(p + 4, nbytes, dt, shape)
            if lclass == 1:
                daddr = self._u64(p + 2)
                if daddr == UNDEF:
                    return np.zeros(shape, dtype=dt.dtype)
                return self._decode(daddr, nbytes, dt, shape)
            if lclass == 2:
                rank = buf[p + 2]
                btree = self._u64(p + 3)
                cdims = [self._u32(p + 11 + 4 * i) for i in range(rank)]
                return self._read_chunked(btree, cdims[:-1], dt, shape, addr)
        elif version in (1, 2):
            rank, lclass = buf[p + 1], buf[p + 2]
            q = p + 8
            daddr = None
            if lclass != 0:
                daddr = self._u64(q); q += 8
            dims = [self._u32(q + 4 * i) for i in range(rank)]
            if lclass == 1:
                return self._decode(daddr, nbytes, dt, shape)
            if lclass == 2:
                return self._read_chunked(daddr, dims[:-1], dt, shape, addr)
            if lclass == 0:
                q += 4 * rank
                return self._decode(q + 4, nbytes, dt, shape)
        raise H5Error("unsupported data layout (version %d)" % version)

    def _read_chunked(self, btree, cdims, dt, shape, header):
        for mtype, _f, _d, _s in self._messages(header):
            if mtype == 0x0B:
                raise H5Error("filtered (compressed) chunks are not supported")
        if dt.vlen_string:
            raise H5Error("chunked variable-length data is not supported")
        out = np.zeros(shape, dtype=dt.dtype)
        rank = len(shape)

        def walk(addr):
            buf = self.buf
            if buf[addr:addr + 4] != b"TREE" or buf[addr + 4] != 1:
                raise H5Error("bad chunk B-tree")
            level, used = buf[addr + 5], self._u16(addr + 6)
            p = addr + 24
            keysize = 8 + 8 * (rank + 1)
            for _ in range(used):
                offs = [self._u64(p + 8 + 8 * i) for i in range(rank)]
                child = self._u64(p + keysize)
                if level > 0:
                    walk(child)
                else:
                    chunk = np.frombuffer(self.buf, dtype=dt.dtype, count=int(np.prod(cdims)), offset=child).reshape(cdims)
                    sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, shape))
                    out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]
                p += keysize + 8
        walk(btree)
        return out.astype(bool) if dt.enum_bool else out

    def attrs(self, path="/"):
        """attributes of a group/dataset as dict name -> value"""
        addr = self._resolve(path)
        out = {}
        buf = self.buf
        for mtype, _f, p, _size in self._messages(addr):
            if mtype != 0x0C:
                continue
            version = buf[p]
            nsize, tsize, ssize = self._u16(p + 2), self._u16(p + 4), self._u16(p + 6)
            q = p + 8 + (1 if version == 3 else 0)
            pad = (lambda x: (x + 7) & ~7) if version == 1 else (lambda x: x)
            name = buf[q:q + nsize].split(b"\0")[0].decode("utf8")
            q += pad(nsize)
            dt = self._datatype(q)
            q += pad(tsize)
            shape = self._dataspace(q)
            q += pad(ssize)
            if shape is None:
                out[name] = None
                continue
            n = int(np.prod(shape)) if shape else 1
            out[name] = self._decode(q, n * dt.size, dt, shape)
        return out


# ------------------------------------------------------------------------- writer
class _Writer(object):
    """Lays a tree of groups/datasets/attributes out as a superblock-0 file."""
    LEAF_K = 64        # symbol-table nodes hold up to 2K entries: one node per group

    def __init__(self):
        self.buf = bytearray(b"\0" * 96)      # superblock placeholder (56 + 40-byte root entry)

    def _align(self, a=8):
        while len(self.buf) % a:
            self.buf.append(0)

    def _alloc(self, data):
        self._align()
        addr = len(self.buf)
        self.buf += data
        return addr

    # -- message encoders
    @staticmethod
    def _dtype_msg(arr):
        dt = arr.dtype
        if dt == np.bool_:
            base = struct.pack("<B3BI", 0x10, 0x08, 0, 0, 1) + struct.pack("<HH", 0, 8)
            names = b"FALSE\0\0\0" + b"TRUE\0\0\0\0"
            return struct.pack("<B3BI", 0x18, 2, 0, 0, 1) + base + names + b"\x00\x01"
        if dt.kind in "iu":
            bits = 0x08 if dt.kind == "i" else 0
            return struct.pack("<B3BI", 0x10, bits, 0, 0, dt.itemsize) + struct.pack("<HH", 0, 8 * dt.itemsize)
        if dt.kind == "f":
            if dt.itemsize == 4:
                props = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
                return struct.pack("<B3BI", 0x11, 0x20, 0x1F, 0, 4) + props
            props = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
            return struct.pack("<B3BI", 0x11, 0x20, 0x3F, 0, 8) + props
        if dt.kind == "S":
            return struct.pack("<B3BI", 0x13, 0, 0, 0, dt.itemsize)      # null-terminated ASCII
        raise H5Error("cannot store dtype %s" % dt)

    @staticmethod
    def _space_msg(shape):
        rank = len(shape)
        return struct.pack("<BBB5x", 1, rank, 0) + b"".join(struct.pack("<Q", int(d)) for d in shape)

    @staticmethod
    def _msg(mtype, data, flags=0):
        pad = (-len(data)) % 8
        return struct.pack("<HHB3x", mtype, len(data) + pad, flags) + data + b"\0" * pad

    def _attr_msg(self, name, value):
        arr = np.asarray(value)
        if arr.dtype.kind == "U":
            arr = np.char.encode(arr, "utf8")
        nm = name.encode("utf8") + b"\0"
        dt, sp = self._dtype_msg(arr), self._space_msg(arr.shape)
        p8 = lambda b: b + b"\0" * ((-len(b)) % 8)
        body = struct.pack("<BxHHH", 1, len(nm), len(dt), len(sp)) + p8(nm) + p8(dt) + p8(sp) + arr.tobytes()
        return self._msg(0x0C, body)

    def _header(self, messages):
        body = b"".join(messages)
        hdr = struct.pack("<BxHII4x", 1, len(messages), 1, len(body))
        return self._alloc(hdr + body)

    def dataset(self, value, attrs=None):
        arr = np.asarray(value)           # (ascontiguousarray would turn scalars into shape (1,))
        if arr.dtype.kind == "U":
            arr = np.char.encode(arr, "utf8")
        raw = arr.tobytes()               # C order
        daddr = self._alloc(raw) if raw else UNDEF
        msgs = [self._msg(0x01, self._space_msg(arr.shape)), self._msg(0x03, self._dtype_msg(arr), 1),
                self._msg(0x08, struct.pack("<BBQQ", 3, 1, daddr, len(raw)))]
        for k, v in (attrs or {}).items():
            msgs.append(self._attr_msg(k, v))
        return self._header(msgs)

    def group(self, children, attrs=None):
        """children: dict name -> object header address (already written)"""
        names = sorted(children, key=lambda s: s.encode("utf8"))
        if len(names) > 2 * self.LEAF_K:
            raise H5Error("too many entries in one group")
        heap_data = bytearray(b"\0" * 8)            # offset 0 = the empty string
        offs = {}
        for nm in names:
            offs[nm] = len(heap_data)
            heap_data += nm.encode("utf8") + b"\0"
            while len(heap_data) % 8:
                heap_data.append(0)
        heap_data += b"\0" * 16                        # room for a free block
        free_off = len(heap_data) - 16
        struct.pack_into("<QQ", heap_data, free_off, 1, 16)       # next = 1 (last), size 16
        data_addr = self._alloc(bytes(heap_data))
        heap = self._alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), free_off, data_addr))
        snod = bytearray(b"SNOD" + struct.pack("<BxH", 1, len(names)))
        for nm in names:
            snod += struct.pack("<QQII16x", offs[nm], children[nm], 0, 0)
        snod += b"\0" * (40 * (2 * self.LEAF_K - len(names)))
        snod_addr = self._alloc(bytes(snod))
        tree = bytearray(b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if names else 0, UNDEF, UNDEF))
        tree += struct.pack("<QQQ", 0, snod_addr, offs[names[-1]] if names else 0)
        tree += b"\0" * (16 * 2 * 16)                  # unused key/child slots (internal K = 16)
        tree_addr = self._alloc(bytes(tree))
        msgs = [self._msg(0x11, struct.pack("<QQ", tree_addr, heap))]
        for k, v in (attrs or {}).items():
            msgs.append(self._attr_msg(k, v))
        return self._header(msgs), tree_addr, heap

    def finish(self, root, tree, heap):
        self._align()
        eof = len(self.buf)
        sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, self.LEAF_K, 16, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
        sb += struct.pack("<QQII", 0, root, 1, 0) + struct.pack("<QQ", tree, heap)
        self.buf[:len(sb)] = sb
        return bytes(self.buf)


def write_h5(filename, tree):
    """tree: nested dict.  A dict value is a group (key '@attrs' holds its attributes),
    anything else is a dataset (numpy array, scalar, bytes or str)."""
    w = _Writer()

    def emit(node):
        children = {}
        for name, value in node.items():
            if name == "@attrs":
                continue
            if isinstance(value, dict):
                children[name] = emit(value)[0]
            else:
                if isinstance(value, str):
                    value = np.array(value.encode("utf8"))
                elif isinstance(value, bytes):
                    value = np.array(value)
                children[name] = w.dataset(value)
        return w.group(children, node.get("@attrs"))

    root, tree_addr, heap = emit(tree)
    with open(filename, "wb") as f:
        f.write(w.finish(root, tree_addr, heap))

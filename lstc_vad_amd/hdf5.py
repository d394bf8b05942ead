"""Read-only HDF5 access without h5py or libhdf5 - the feature files of the reference are HDF5.

The reference opens its I3D feature archives and the UCF ground truth with ``h5py.File(path, 'r')`` and reads whole
datasets, ``h5[key + '.npy'][:]`` (utils/load_dataset.py:33-46, :113-119, :285-286, :409-411, :466-499, :536-547;
Train/temporal_transformer_shanghaitech.py:53,199).  Neither h5py nor a Python binding of libhdf5 is part of this image,
so this module parses the file format itself (HDF5 File Format Specification, versions 1.1 / 2.0 / 3.0) for what such
files contain, and offers the h5py subset those call sites use::

    with File(path) as f:            # 'r' only
        f.keys(); len(f); "01_001.npy" in f
        d = f["01_001.npy"]          # Dataset: .shape .dtype .ndim .size, d[:], d[...], d[2:5, :16]
        g = f["group/sub"]           # Group (nested paths work)

Supported: superblock 0-3; object headers v1 and v2 (continuation blocks); groups as symbol tables (B-tree v1 + local heap:
what h5py writes by default) and as compact or dense link messages (fractal heap + v2 B-tree: ``libver='latest'`` files);
datasets with compact, contiguous or chunked layout (v1 B-tree chunk index; layout v4 single-chunk, implicit and fixed-array
indexes), filters deflate / shuffle / fletcher32; fixed-point and IEEE floating-point element types of either byte order;
fill values for unallocated storage.  Everything else (compound / variable-length types, szip, external or virtual storage,
extensible-array and v2-B-tree chunk indexes, soft links) raises ``NotImplementedError`` naming the feature.

Contiguous datasets are served zero-copy from one shared memory map (``Dataset.view()``); ``d[...]`` returns an owned array
like h5py does.  Pinned against files written by the real libhdf5 1.10.6: tests/golden/make_hdf5_fixtures.py ->
tests/golden/hdf5/*.h5, tests/test_hdf5_host.py.
"""
from __future__ import annotations

import mmap
import os
import struct
import zlib

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = {4: 0xFFFFFFFF, 8: 0xFFFFFFFFFFFFFFFF, 2: 0xFFFF}

MSG_DATASPACE, MSG_LINKINFO, MSG_DATATYPE, MSG_FILL_OLD, MSG_FILL, MSG_LINK = 0x1, 0x2, 0x3, 0x4, 0x5, 0x6
MSG_LAYOUT, MSG_FILTERS, MSG_CONT, MSG_SYMTAB = 0x8, 0xB, 0x10, 0x11


class HDF5Error(OSError):
    pass


def _lookup3(key: bytes, init: int = 0) -> int:
    """Jenkins lookup3 ``hashlittle`` - the checksum of version-2 structures and the name hash of dense link storage."""
    M = 0xFFFFFFFF

    def rot(x, k):
        return ((x << k) | (x >> (32 - k))) & M

    n = len(key)
    a = b = c = (0xDEADBEEF + n + init) & M
    i = 0
    while n > 12:
        a = (a + int.from_bytes(key[i:i + 4], "little")) & M
        b = (b + int.from_bytes(key[i + 4:i + 8], "little")) & M
        c = (c + int.from_bytes(key[i + 8:i + 12], "little")) & M
        a = (a - c) & M; a ^= rot(c, 4); c = (c + b) & M
        b = (b - a) & M; b ^= rot(a, 6); a = (a + c) & M
        c = (c - b) & M; c ^= rot(b, 8); b = (b + a) & M
        a = (a - c) & M; a ^= rot(c, 16); c = (c + b) & M
        b = (b - a) & M; b ^= rot(a, 19); a = (a + c) & M
        c = (c - b) & M; c ^= rot(b, 4); b = (b + a) & M
        i += 12
        n -= 12
    if n == 0:
        return c
    tail = key[i:] + b"\0" * (12 - n)
    a = (a + int.from_bytes(tail[0:4], "little")) & M
    b = (b + int.from_bytes(tail[4:8], "little")) & M
    c = (c + int.from_bytes(tail[8:12], "little")) & M
    c ^= b; c = (c - rot(b, 14)) & M
    a ^= c; a = (a - rot(c, 11)) & M
    b ^= a; b = (b - rot(a, 25)) & M
    c ^= b; c = (c - rot(b, 16)) & M
    a ^= c; a = (a - rot(c, 4)) & M
    b ^= a; b = (b - rot(a, 14)) & M
    c ^= b; c = (c - rot(b, 24)) & M
    return c


class _Reader:
    """Cursor over the file's memory map with the superblock's offset / length sizes."""

    def __init__(self, buf, O: int, L: int, base: int = 0):
        self.buf, self.O, self.L, self.base = buf, O, L, base

    def uint(self, pos: int, n: int) -> int:
        return int.from_bytes(self.buf[pos:pos + n], "little")

    def off(self, pos: int) -> int:
        v = self.uint(pos, self.O)
        return v if v == _UNDEF[self.O] else v + self.base

    def length(self, pos: int) -> int:
        return self.uint(pos, self.L)

    def undef(self, v: int) -> bool:
        return v == _UNDEF[self.O]

    def expect(self, pos: int, sig: bytes, what: str):
        if bytes(self.buf[pos:pos + len(sig)]) != sig:
            raise HDF5Error(f"bad {what} signature at byte {pos}: {bytes(self.buf[pos:pos + len(sig)])!r}")


# ----------------------------------------------------------------------------------------------- object headers
def _messages(rd: _Reader, addr: int):
    """Yield (type, flags, data_offset, size) of every header message of the object at ``addr`` (v1 or v2 header)."""
    buf = rd.buf
    if bytes(buf[addr:addr + 4]) == b"OHDR":
        if buf[addr + 4] != 2:
            raise HDF5Error(f"object header version {buf[addr + 4]} at {addr}")
        flags = buf[addr + 5]
        p = addr + 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        n = 1 << (flags & 3)
        size0 = rd.uint(p, n)
        p += n
        track = bool(flags & 0x04)
        blocks = [(p, size0)]
        while blocks:
            p, size = blocks.pop(0)
            end = p + size
            while p + 4 <= end:
                mtype, msize, mflags = buf[p], rd.uint(p + 1, 2), buf[p + 3]
                p += 4 + (2 if track else 0)
                if p + msize > end:
                    break
                if mtype == MSG_CONT:
                    caddr, clen = rd.off(p), rd.length(p + rd.O)
                    rd.expect(caddr, b"OCHK", "object header continuation")
                    blocks.append((caddr + 4, clen - 8))            # minus signature and checksum
                elif mtype != 0:
                    yield mtype, mflags, p, msize
                p += msize
        return
    if buf[addr] != 1:
        raise HDF5Error(f"no object header at byte {addr} (version byte {buf[addr]})")
    nmsg, size0 = rd.uint(addr + 2, 2), rd.uint(addr + 8, 4)
    blocks = [(addr + 16, size0)]
    seen = 0
    while blocks and seen < nmsg:
        p, size = blocks.pop(0)
        end = p + size
        while p + 8 <= end and seen < nmsg:
            mtype, msize, mflags = rd.uint(p, 2), rd.uint(p + 2, 2), buf[p + 4]
            p += 8
            seen += 1
            if mtype == MSG_CONT:
                blocks.append((rd.off(p), rd.length(p + rd.O)))
            elif mtype != 0:
                yield mtype, mflags, p, msize
            p += msize


# ----------------------------------------------------------------------------------------------- groups
def _local_heap_data(rd: _Reader, addr: int) -> int:
    rd.expect(addr, b"HEAP", "local heap")
    return rd.off(addr + 8 + 2 * rd.L)


def _cstr(buf, pos: int) -> str:
    end = pos
    while buf[end] != 0:
        end += 1
    return bytes(buf[pos:end]).decode("utf-8")


def _symtab_links(rd: _Reader, btree: int, heap: int, out: dict):
    """Old-style group: walk the v1 B-tree (node type 0) down to its symbol-table nodes."""
    heap_data = _local_heap_data(rd, heap)
    stack = [btree]
    while stack:
        node = stack.pop()
        sig = bytes(rd.buf[node:node + 4])
        if sig == b"TREE":
            if rd.buf[node + 4] != 0:
                raise HDF5Error(f"B-tree node at {node} is not a group node")
            used = rd.uint(node + 6, 2)
            p = node + 8 + 2 * rd.O + rd.L                 # skip siblings and key 0
            kids = []
            for _ in range(used):
                kids.append(rd.off(p))
                p += rd.O + rd.L
            stack.extend(reversed(kids))
        elif sig == b"SNOD":
            n = rd.uint(node + 6, 2)
            p = node + 8
            for _ in range(n):
                name = _cstr(rd.buf, heap_data + rd.uint(p, rd.O))
                out[name] = rd.off(p + rd.O)
                p += 2 * rd.O + 24
        else:
            raise HDF5Error(f"unexpected node {sig!r} at byte {node} in a group B-tree")


def _parse_link(rd: _Reader, p: int):
    """Link message body at ``p`` -> (name, object header address) for hard links."""
    buf = rd.buf
    if buf[p] != 1:
        raise HDF5Error(f"link message version {buf[p]}")
    flags = buf[p + 1]
    q = p + 2
    ltype = 0
    if flags & 0x08:
        ltype = buf[q]
        q += 1
    if flags & 0x04:
        q += 8
    if flags & 0x10:
        q += 1
    n = 1 << (flags & 3)
    nlen = rd.uint(q, n)
    q += n
    name = bytes(buf[q:q + nlen]).decode("utf-8")
    q += nlen
    if ltype != 0:
        return name, None                                  # soft / external link: listed, not followed
    return name, rd.off(q)


class _FractalHeap:
    """Managed objects of a fractal heap (dense link storage): heap ID -> bytes."""

    def __init__(self, rd: _Reader, addr: int):
        self.rd = rd
        rd.expect(addr, b"FRHP", "fractal heap")
        O, L = rd.O, rd.L
        p = addr + 5
        self.id_len = rd.uint(p, 2); p += 2
        filt_len = rd.uint(p, 2); p += 2
        p += 1                                             # flags
        self.max_managed = rd.uint(p, 4); p += 4
        p += L + O                                         # next huge id, huge B-tree address
        p += L + O                                         # free space, free-space manager address
        p += 4 * L                                         # managed space, allocated, iterator offset, number of managed objects
        p += 2 * L + 2 * L                                 # huge size / count, tiny size / count
        self.width = rd.uint(p, 2); p += 2
        self.start_size = rd.length(p); p += L
        self.max_direct = rd.length(p); p += L
        self.max_heap_bits = rd.uint(p, 2); p += 2
        p += 2                                             # starting rows of the root indirect block
        self.root = rd.off(p); p += O
        self.root_rows = rd.uint(p, 2); p += 2
        if filt_len:
            raise NotImplementedError("HDF5: filtered fractal heap (dense link storage with I/O filters)")
        self.off_bytes = (self.max_heap_bits + 7) // 8
        self.max_dblock_rows = (self.max_direct // self.start_size).bit_length() + 1
        self.blocks = []                                   # (heap offset, size, file address) of direct blocks
        self.checksummed = bool(rd.buf[addr + 9] & 0x02)
        if self.root_rows == 0:
            self.blocks.append((0, self.start_size, self.root))
        else:
            self._indirect(self.root, self.root_rows, 0)

    def _row_size(self, row: int) -> int:
        return self.start_size * (1 if row < 2 else 1 << (row - 1))

    def _indirect(self, addr: int, nrows: int, heap_off: int):
        rd = self.rd
        rd.expect(addr, b"FHIB", "fractal heap indirect block")
        p = addr + 5 + rd.O + self.off_bytes
        off = heap_off
        for row in range(nrows):
            size = self._row_size(row)
            for _ in range(self.width):
                child = rd.off(p)
                p += rd.O
                if size <= self.max_direct:
                    if not rd.undef(child):
                        self.blocks.append((off, size, child))
                elif not rd.undef(child):
                    rows = (size // self.start_size).bit_length() - 1 - (self.width.bit_length() - 1) + 1
                    self._indirect(child, rows, off)
                off += size

    def get(self, heap_id: bytes) -> bytes:
        kind = (heap_id[0] >> 4) & 3
        if kind != 0:
            raise NotImplementedError("HDF5: huge / tiny fractal-heap objects (link names that long are not expected)")
        off = int.from_bytes(heap_id[1:1 + self.off_bytes], "little")
        len_bytes = min(self.id_len - 1 - self.off_bytes, 8)
        n = int.from_bytes(heap_id[1 + self.off_bytes:1 + self.off_bytes + len_bytes], "little")
        for boff, size, addr in self.blocks:
            if boff <= off < boff + size:
                return bytes(self.rd.buf[addr + (off - boff):addr + (off - boff) + n])
        raise HDF5Error(f"fractal heap offset {off} is in no direct block")


def _dense_links(rd: _Reader, heap_addr: int, bt2_addr: int, out: dict):
    """New-style dense group: every record of the v2 B-tree (type 5: name hash + heap ID) names a link message in the heap."""
    heap = _FractalHeap(rd, heap_addr)
    rd.expect(bt2_addr, b"BTHD", "v2 B-tree header")
    p = bt2_addr + 5
    rtype = rd.buf[p]; p += 1
    node_size = rd.uint(p, 4); p += 4
    rec_size = rd.uint(p, 2); p += 2
    depth = rd.uint(p, 2); p += 2
    p += 2                                                 # split / merge percent
    root = rd.off(p); p += rd.O
    root_nrec = rd.uint(p, 2); p += 2
    if rtype != 5:
        raise NotImplementedError(f"HDF5: v2 B-tree record type {rtype} as a link name index")
    if rd.undef(root):
        return
    # records per node by level (spec III.A.2): needed to size the child-pointer fields of internal nodes
    max_leaf = (node_size - 10) // rec_size
    nbytes = lambda v: max(1, (v.bit_length() + 7) // 8)

    # cumulative maxima per depth
    lvl = [(max_leaf, max_leaf)]                           # (max records in a node, max records in the subtree)
    for d in range(1, depth + 1):
        child_nrec_b = nbytes(lvl[d - 1][0])
        child_tot_b = nbytes(lvl[d - 1][1]) if d > 1 else 0
        per = (node_size - 10 - (rd.O + child_nrec_b + child_tot_b)) // (rec_size + rd.O + child_nrec_b + child_tot_b)
        lvl.append((per, per + (per + 1) * lvl[d - 1][1]))

    def walk(addr, nrec, d):
        if d == 0:
            rd.expect(addr, b"BTLF", "v2 B-tree leaf")
            q = addr + 6
            for _ in range(nrec):
                hid = bytes(rd.buf[q + 4:q + rec_size])
                name, target = _parse_link(_Reader(heap.get(hid), rd.O, rd.L, rd.base), 0)
                out[name] = target
                q += rec_size
            return
        rd.expect(addr, b"BTIN", "v2 B-tree internal node")
        q = addr + 6 + nrec * rec_size
        cb = nbytes(lvl[d - 1][0])
        tb = nbytes(lvl[d - 1][1]) if d > 1 else 0
        for _ in range(nrec + 1):
            child = rd.off(q)
            cn = rd.uint(q + rd.O, cb)
            q += rd.O + cb + tb
            walk(child, cn, d - 1)

    walk(root, root_nrec, depth)


# ----------------------------------------------------------------------------------------------- datasets
def _dtype_of(rd: _Reader, p: int) -> np.dtype:
    cv = rd.buf[p]
    cls, bits0 = cv & 0x0F, rd.buf[p + 1]
    size = rd.uint(p + 4, 4)
    order = ">" if (bits0 & 1) else "<"
    if cls == 0:
        signed = bool(bits0 & 0x08)
        if size not in (1, 2, 4, 8):
            raise NotImplementedError(f"HDF5: {size}-byte integer elements")
        return np.dtype(f"{order}{'i' if signed else 'u'}{size}")
    if cls == 1:
        if size not in (2, 4, 8):
            raise NotImplementedError(f"HDF5: {size}-byte floating-point elements")
        if rd.buf[p + 2] & 0x40 and (bits0 & 1):
            raise NotImplementedError("HDF5: VAX-order floating point")
        return np.dtype(f"{order}f{size}")
    names = {2: "time", 3: "string", 4: "bitfield", 5: "opaque", 6: "compound", 7: "reference", 8: "enum", 9: "variable-length",
             10: "array"}
    raise NotImplementedError(f"HDF5: {names.get(cls, cls)} element type")


def _unfilter(raw: bytes, filters, mask: int, itemsize: int) -> bytes:
    for i, (fid, cd) in reversed(list(enumerate(filters))):
        if mask & (1 << i):
            continue
        if fid == 1:
            raw = zlib.decompress(raw)
        elif fid == 2:
            n = cd[0] if cd else itemsize
            if n > 1:
                a = np.frombuffer(raw, np.uint8)
                m = len(a) // n
                raw = (a[:m * n].reshape(n, m).T.tobytes()) + bytes(a[m * n:])
        elif fid == 3:
            raw = raw[:-4]
        else:
            names = {4: "szip", 5: "nbit", 6: "scaleoffset", 32001: "blosc", 32004: "lz4", 32015: "zstd", 32008: "bitshuffle"}
            raise NotImplementedError(f"HDF5: filter {names.get(fid, fid)}")
    return raw


class Dataset:
    def __init__(self, file: "File", addr: int, name: str):
        self.file, self.name = file, name
        rd = file._rd
        self.shape = None
        self.dtype = None
        self._layout = None
        self._filters = []
        self._fill = None
        for mtype, mflags, p, size in _messages(rd, addr):
            if mtype == MSG_DATASPACE:
                ver, rank = rd.buf[p], rd.buf[p + 1]
                q = p + (8 if ver == 1 else 4)
                if ver == 2 and rd.buf[p + 3] == 2:
                    raise NotImplementedError("HDF5: null dataspace")
                self.shape = tuple(rd.length(q + i * rd.L) for i in range(rank))
            elif mtype == MSG_DATATYPE:
                self.dtype = _dtype_of(rd, p)
            elif mtype == MSG_LAYOUT:
                self._layout = self._parse_layout(rd, p)
            elif mtype == MSG_FILTERS:
                self._filters = self._parse_filters(rd, p)
            elif mtype == MSG_FILL:
                self._fill = self._parse_fill(rd, p)
        if self.shape is None or self.dtype is None or self._layout is None:
            raise HDF5Error(f"{name}: not a dataset (dataspace / datatype / layout message missing)")

    ndim = property(lambda self: len(self.shape))
    size = property(lambda self: int(np.prod(self.shape, dtype=np.int64)))

    def __len__(self):
        if not self.shape:
            raise TypeError("len() of a scalar dataset")
        return self.shape[0]

    # --- message bodies
    @staticmethod
    def _parse_fill(rd, p):
        ver = rd.buf[p]
        if ver in (1, 2):
            if ver == 2 and not rd.buf[p + 3]:
                return None
            n = rd.uint(p + 4, 4)
            return bytes(rd.buf[p + 8:p + 8 + n]) if n else None
        if ver == 3:
            if not (rd.buf[p + 1] & 0x20):
                return None
            n = rd.uint(p + 2, 4)
            return bytes(rd.buf[p + 6:p + 6 + n]) if n else None
        return None

    @staticmethod
    def _parse_filters(rd, p):
        ver, n = rd.buf[p], rd.buf[p + 1]
        q = p + (8 if ver == 1 else 2)
        out = []
        for _ in range(n):
            fid = rd.uint(q, 2)
            q += 2
            nlen = 0
            if ver == 1 or fid >= 256:
                nlen = rd.uint(q, 2)
                q += 2
            q += 2                                         # flags
            ncd = rd.uint(q, 2)
            q += 2
            if ver == 1:
                nlen = (nlen + 7) // 8 * 8
            q += nlen
            cd = [rd.uint(q + 4 * i, 4) for i in range(ncd)]
            q += 4 * ncd
            if ver == 1 and ncd % 2:
                q += 4
            out.append((fid, cd))
        return out

    @staticmethod
    def _parse_layout(rd, p):
        ver = rd.buf[p]
        if ver in (1, 2):
            rank, cls = rd.buf[p + 1], rd.buf[p + 2]
            q = p + 8
            addr = None
            if cls != 0:
                addr = rd.off(q)
                q += rd.O
            dims = [rd.uint(q + 4 * i, 4) for i in range(rank)]
            q += 4 * rank
            if cls == 1:
                return ("contiguous", addr, None)
            if cls == 2:
                return ("chunked_v1", addr, dims)                       # trailing element size is not stored: v1/v2 dims = rank
            n = rd.uint(q, 4)
            return ("compact", q + 4, n)
        if ver == 3:
            cls = rd.buf[p + 1]
            if cls == 0:
                return ("compact", p + 4, rd.uint(p + 2, 2))
            if cls == 1:
                return ("contiguous", rd.off(p + 2), rd.length(p + 2 + rd.O))
            if cls == 2:
                rank1 = rd.buf[p + 2]
                addr = rd.off(p + 3)
                dims = [rd.uint(p + 3 + rd.O + 4 * i, 4) for i in range(rank1)]
                return ("chunked_v1", addr, dims[:-1])
            raise NotImplementedError(f"HDF5: data layout class {cls}")
        if ver == 4:
            cls = rd.buf[p + 1]
            if cls == 0:
                return ("compact", p + 4, rd.uint(p + 2, 2))
            if cls == 1:
                return ("contiguous", rd.off(p + 2), rd.length(p + 2 + rd.O))
            if cls == 3:
                raise NotImplementedError("HDF5: virtual dataset")
            flags, rank1, enc = rd.buf[p + 2], rd.buf[p + 3], rd.buf[p + 4]
            q = p + 5
            dims = [rd.uint(q + enc * i, enc) for i in range(rank1)]
            q += enc * rank1
            itype = rd.buf[q]
            q += 1
            if itype == 1:                                             # single chunk
                fsize = fmask = None
                if flags & 0x02:
                    fsize, fmask = rd.length(q), rd.uint(q + rd.L, 4)
                    q += rd.L + 4
                return ("single", rd.off(q), dims[:-1], fsize, fmask)
            if itype == 2:
                return ("implicit", rd.off(q), dims[:-1])
            if itype == 3:
                return ("fixed_array", rd.off(q + 1), dims[:-1])
            raise NotImplementedError("HDF5: " + {4: "extensible-array", 5: "v2-B-tree"}.get(itype, str(itype)) + " chunk index")
        raise NotImplementedError(f"HDF5: data layout message version {ver}")

    # --- reading
    def _empty(self):
        out = np.zeros(self.shape, self.dtype)
        if self._fill and len(self._fill) == self.dtype.itemsize:
            out[...] = np.frombuffer(self._fill, self.dtype)[0]
        return out

    def view(self):
        """Zero-copy read-only array over the file's memory map (contiguous layout, no filters), else None."""
        kind = self._layout[0]
        rd = self.file._rd
        if kind == "contiguous" and not self._filters:
            addr = self._layout[1]
            if rd.undef(addr) or self.size == 0:
                return None
            a = np.frombuffer(rd.buf, dtype=self.dtype, count=self.size, offset=addr).reshape(self.shape)
            return a
        return None

    def _chunks_v1(self, btree, cdims):
        """(chunk offsets, file address, stored size, filter mask) of every chunk under a v1 B-tree (node type 1)."""
        rd = self.file._rd
        rank = len(cdims)
        if rd.undef(btree):
            return
        stack = [btree]
        ksz = 8 + 8 * (rank + 1)
        while stack:
            node = stack.pop()
            rd.expect(node, b"TREE", "chunk B-tree node")
            if rd.buf[node + 4] != 1:
                raise HDF5Error(f"B-tree node at {node} is not a chunk node")
            level, used = rd.buf[node + 5], rd.uint(node + 6, 2)
            p = node + 8 + 2 * rd.O
            for _ in range(used):
                csize, cmask = rd.uint(p, 4), rd.uint(p + 4, 4)
                offs = tuple(rd.uint(p + 8 + 8 * i, 8) for i in range(rank))
                child = rd.off(p + ksz)
                p += ksz + rd.O
                if level == 0:
                    yield offs, child, csize, cmask
                else:
                    stack.append(child)

    def _chunk_grid(self, cdims):
        counts = [(s + c - 1) // c for s, c in zip(self.shape, cdims)]
        return counts, np.ndindex(*counts)

    def _chunks(self):
        rd = self.file._rd
        lay = self._layout
        kind = lay[0]
        item = self.dtype.itemsize
        if kind == "chunked_v1":
            yield from self._chunks_v1(lay[1], lay[2])
        elif kind == "single":
            _, addr, cdims, fsize, fmask = lay
            if not rd.undef(addr):
                n = fsize if fsize is not None else int(np.prod(cdims)) * item
                yield tuple(0 for _ in cdims), addr, n, fmask or 0
        elif kind == "implicit":
            _, addr, cdims = lay
            if not rd.undef(addr):
                csize = int(np.prod(cdims)) * item
                counts, it = self._chunk_grid(cdims)
                for k, idx in enumerate(it):
                    yield tuple(i * c for i, c in zip(idx, cdims)), addr + k * csize, csize, 0
        elif kind == "fixed_array":
            _, hdr, cdims = lay
            if rd.undef(hdr):
                return
            rd.expect(hdr, b"FAHD", "fixed array header")
            client, esize, page_bits = rd.buf[hdr + 5], rd.buf[hdr + 6], rd.buf[hdr + 7]
            nelem = rd.length(hdr + 8)
            dblk = rd.off(hdr + 8 + rd.L)
            if rd.undef(dblk):
                return
            rd.expect(dblk, b"FADB", "fixed array data block")
            p = dblk + 6 + rd.O
            per_page = 1 << page_bits
            csize = int(np.prod(cdims)) * item
            counts, it = self._chunk_grid(cdims)
            idxs = list(it)
            entries = []
            if nelem > per_page:                                       # paged: bitmap, then pages each followed by a checksum
                npages = (nelem + per_page - 1) // per_page
                bitmap = bytes(rd.buf[p:p + (npages + 7) // 8])
                p += (npages + 7) // 8 + 4                             # data block checksum precedes the pages
                for pg in range(npages):
                    cnt = min(per_page, nelem - pg * per_page)
                    if bitmap[pg // 8] & (0x80 >> (pg % 8)):
                        for e in range(cnt):
                            entries.append(p + e * esize)
                        p += cnt * esize + 4
                    else:
                        entries.extend([None] * cnt)
            else:
                entries = [p + e * esize for e in range(nelem)]
            for k, idx in enumerate(idxs):
                q = entries[k] if k < len(entries) else None
                if q is None:
                    continue
                addr = rd.off(q)
                if rd.undef(addr):
                    continue
                if client == 1:
                    nb = esize - rd.O - 4
                    yield tuple(i * c for i, c in zip(idx, cdims)), addr, rd.uint(q + rd.O, nb), rd.uint(q + rd.O + nb, 4)
                else:
                    yield tuple(i * c for i, c in zip(idx, cdims)), addr, csize, 0
        else:
            raise HDF5Error(kind)

    def read(self) -> np.ndarray:
        """The whole dataset as an owned, native-byte-order-agnostic array of ``self.dtype``."""
        rd = self.file._rd
        kind = self._layout[0]
        if self.size == 0:
            return np.zeros(self.shape, self.dtype)
        if kind == "compact":
            _, p, n = self._layout
            return np.frombuffer(bytes(rd.buf[p:p + n]), self.dtype, self.size).reshape(self.shape).copy()
        if kind == "contiguous":
            if self._filters:
                raise HDF5Error(f"{self.name}: filters on a contiguous dataset")
            v = self.view()
            return self._empty() if v is None else np.array(v)
        cdims = self._layout[2]
        out = self._empty()
        item = self.dtype.itemsize
        for offs, addr, nbytes, mask in self._chunks():
            raw = rd.buf[addr:addr + nbytes]
            if self._filters:
                raw = _unfilter(bytes(raw), self._filters, mask, item)
            chunk = np.frombuffer(raw, self.dtype, int(np.prod(cdims))).reshape(cdims)
            sel_out = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, self.shape))
            sel_in = tuple(slice(0, s.stop - s.start) for s in sel_out)
            out[sel_out] = chunk[sel_in]
        return out

    def __getitem__(self, key):
        v = self.view()
        if v is not None:
            return np.array(v[key])                                    # only the selected part is copied out of the map
        return self.read()[key]

    def __array__(self, dtype=None, copy=None):
        a = self.read()
        return a if dtype is None else a.astype(dtype)

    def __repr__(self):
        return f'<HDF5 dataset "{self.name}": shape {self.shape}, type "{self.dtype.str}">'


class Group:
    def __init__(self, file: "File", addr: int, name: str):
        self.file, self.name, self._addr = file, name, addr
        self._links = None

    def _load(self):
        if self._links is not None:
            return self._links
        rd = self.file._rd
        links = {}
        for mtype, mflags, p, size in _messages(rd, self._addr):
            if mtype == MSG_SYMTAB:
                _symtab_links(rd, rd.off(p), rd.off(p + rd.O), links)
            elif mtype == MSG_LINK:
                name, target = _parse_link(rd, p)
                links[name] = target
            elif mtype == MSG_LINKINFO:
                flags = rd.buf[p + 1]
                q = p + 2 + (8 if flags & 1 else 0)
                heap, bt2 = rd.off(q), rd.off(q + rd.O)
                if not rd.undef(heap):
                    _dense_links(rd, heap, bt2, links)
        self._links = links
        return links

    def keys(self):
        return sorted(self._load())                                    # h5py lists names in (alphabetical) index order

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self._load())

    def __contains__(self, key):
        try:
            self._resolve(key)
            return True
        except KeyError:
            return False

    def _resolve(self, key: str):
        node = self.file if key.startswith("/") else self
        for part in [s for s in key.split("/") if s]:
            if not isinstance(node, Group):
                raise KeyError(key)
            links = node._load()
            if part not in links:
                raise KeyError(f"Unable to open object (object '{part}' doesn't exist)")
            target = links[part]
            if target is None:
                raise NotImplementedError(f"HDF5: '{part}' is a soft or external link")
            node = node.file._object(target, (node.name.rstrip("/") + "/" + part))
        return node

    def __getitem__(self, key: str):
        return self._resolve(key)

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]

    def __repr__(self):
        return f'<HDF5 group "{self.name}" ({len(self)} members)>'


class File(Group):
    def __init__(self, path, mode: str = "r"):
        if mode != "r":
            raise ValueError("lstc_vad_amd.hdf5.File is read-only (mode 'r')")
        self.filename = os.fspath(path)
        self._fh = open(self.filename, "rb")
        try:
            self._map = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        except ValueError:
            self._fh.close()
            raise HDF5Error(f"{self.filename}: empty file")
        buf = memoryview(self._map)
        base = 0
        while bytes(buf[base:base + 8]) != _SIG:                       # the superblock may sit at 0, 512, 1024, 2048, ...
            base = 512 if base == 0 else base * 2
            if base + 8 > len(buf):
                self.close()
                raise HDF5Error(f"{self.filename}: not an HDF5 file (no superblock signature)")
        ver = buf[base + 8]
        if ver in (0, 1):
            O, L = buf[base + 13], buf[base + 14]
            p = base + 24 + (4 if ver == 1 else 0)
            rd = _Reader(buf, O, L, 0)
            rd.base = rd.uint(p, O)
            entry = p + 4 * O
            root = rd.off(entry + O)
        elif ver in (2, 3):
            O, L = buf[base + 9], buf[base + 10]
            rd = _Reader(buf, O, L, 0)
            rd.base = rd.uint(base + 12, O)
            root = rd.off(base + 12 + 3 * O)
            stored = rd.uint(base + 12 + 4 * O, 4)
            if _lookup3(bytes(buf[base:base + 12 + 4 * O])) != stored:
                self.close()
                raise HDF5Error(f"{self.filename}: superblock checksum mismatch")
        else:
            self.close()
            raise HDF5Error(f"{self.filename}: superblock version {ver}")
        if O not in (2, 4, 8) or L not in (2, 4, 8):
            self.close()
            raise HDF5Error(f"{self.filename}: offset/length sizes {O}/{L}")
        if rd.base == 0 and base:
            rd.base = base                                             # user block in front of a v0 superblock
        self._rd = rd
        self._cache = {}
        Group.__init__(self, self, root, "/")

    def _object(self, addr: int, name: str):
        hit = self._cache.get(addr)
        if hit is not None:
            return hit
        kinds = {m[0] for m in _messages(self._rd, addr)}
        obj = Dataset(self, addr, name) if MSG_LAYOUT in kinds else Group(self, addr, name)
        self._cache[addr] = obj
        return obj

    def close(self):
        """Release the map.  Arrays from ``Dataset.view()`` keep it alive until they are gone."""
        self._cache = {}
        self._links = None
        rd = getattr(self, "_rd", None)
        if rd is not None:
            try:
                rd.buf.release()
            except BufferError:
                pass                                                   # a zero-copy view is still exported
            self._rd = None
        for h in ("_map", "_fh"):
            o = getattr(self, h, None)
            if o is not None:
                try:
                    o.close()
                except (BufferError, ValueError):
                    pass
                setattr(self, h, None)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __repr__(self):
        return f'<HDF5 file "{os.path.basename(self.filename)}" (mode r)>'

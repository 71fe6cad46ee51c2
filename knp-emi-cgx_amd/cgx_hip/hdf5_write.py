"""Minimal HDF5 writer for the XDMF output of the native path (pure Python + NumPy, no HDF5 library).

Replaces what ``dolfinx.io.XDMFFile.write_mesh / write_meshtags / write_function`` hand to libhdf5 in the reference's
``SolverKNPEMI.init_xdmf_savefile / save_xdmf`` (src/CGx/KNPEMI/KNPEMIx_solver.py:766-797): nested groups and contiguous
numeric datasets, nothing else.  The file uses the oldest, universally readable encoding ("HDF5 File Format Specification
Version 3.0", the default of libhdf5 itself): version-0 superblock, version-1 object headers, symbol-table groups (B-tree v1
+ local heap + symbol nodes), data layout message version 3 (contiguous), little-endian IEEE / two's-complement types.

Raw data are appended to the file as they arrive (a time series never sits in memory); all metadata are written by
``close()`` behind the data, then the superblock at offset 0.  Checked by reading the files back with cgx_hip/hdf5_min.py and
with the image's libhdf5 1.10.6 (tests/test_xdmf_output.py).
"""
from __future__ import annotations

import struct

import numpy as np

_UNDEF = 0xFFFFFFFFFFFFFFFF
_LEAF_K = 4          # a symbol node holds up to 2 * _LEAF_K entries
_INTERNAL_K = 16     # a B-tree node holds up to 2 * _INTERNAL_K children


def _pad8(b: bytes) -> bytes:
    return b + b"\0" * (-len(b) % 8)


def _dtype_message(dt: np.dtype) -> bytes:
    dt = np.dtype(dt)
    if dt.byteorder == ">":
        raise ValueError("little-endian data only")
    if dt.kind == "f" and dt.itemsize in (4, 8):
        exp_bits, man_bits = (8, 23) if dt.itemsize == 4 else (11, 52)
        bias = (1 << (exp_bits - 1)) - 1
        # class 1 (floating point), version 1; bit field: little-endian, implied mantissa msb (2 << 4), sign bit position
        head = struct.pack("<BBBBI", 0x11, 0x20, 8 * dt.itemsize - 1, 0, dt.itemsize)
        prop = struct.pack("<HHBBBBI", 0, 8 * dt.itemsize, man_bits, exp_bits, 0, man_bits, bias)
        return head + prop
    if dt.kind in "iu" and dt.itemsize in (1, 2, 4, 8):
        head = struct.pack("<BBBBI", 0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0, dt.itemsize)
        return head + struct.pack("<HH", 0, 8 * dt.itemsize)
    raise ValueError(f"dtype {dt} is not supported by the minimal HDF5 writer")


def _message(mtype: int, body: bytes) -> bytes:
    body = _pad8(body)
    return struct.pack("<HHBBBB", mtype, len(body), 0, 0, 0, 0) + body


def _object_header(messages) -> bytes:
    body = b"".join(messages)
    # version 1, reserved, number of messages, reference count, header size, 4 bytes of padding to the 8-byte boundary
    return struct.pack("<BBHII", 1, 0, len(messages), 1, len(body)) + b"\0\0\0\0" + body


class Hdf5Writer:
    def __init__(self, path):
        self.path = str(path)
        self.f = open(self.path, "wb")
        self.f.write(b"\0" * 96)                       # the superblock comes last
        self.pos = 96
        self.tree = {}                                 # nested dicts; leaves: (address, shape, dtype)
        self.closed = False

    # ---- data ---------------------------------------------------------------------------------------------------------
    def _append(self, b: bytes) -> int:
        pad = -self.pos % 8
        if pad:
            self.f.write(b"\0" * pad)
            self.pos += pad
        addr = self.pos
        self.f.write(b)
        self.pos += len(b)
        return addr

    def write(self, name: str, array):
        """store ``array`` as the contiguous dataset ``/a/b/c`` (groups are created as needed)"""
        if self.closed:
            raise ValueError("file is closed")
        a = np.ascontiguousarray(array)
        if a.dtype.byteorder == ">":
            a = a.astype(a.dtype.newbyteorder("<"))
        _dtype_message(a.dtype)                        # raises early for unsupported types
        parts = [s for s in name.split("/") if s]
        if not parts:
            raise ValueError("empty dataset name")
        node = self.tree
        for g in parts[:-1]:
            node = node.setdefault(g, {})
            if not isinstance(node, dict):
                raise ValueError(f"'{g}' in '{name}' is a dataset")
        if parts[-1] in node:
            raise ValueError(f"'{name}' exists")
        addr = self._append(a.tobytes()) if a.size else _UNDEF
        node[parts[-1]] = (addr, a.shape, a.dtype)

    # ---- metadata -----------------------------------------------------------------------------------------------------
    def _dataset_header(self, addr, shape, dt) -> int:
        rank = len(shape)
        space = struct.pack("<BBBBI", 1, rank, 0, 0, 0) + b"".join(struct.pack("<Q", int(d)) for d in shape)
        size = int(np.prod(shape, dtype=np.int64)) * dt.itemsize if rank else dt.itemsize
        msgs = [_message(0x0001, space),
                _message(0x0003, _dtype_message(dt)),
                _message(0x0005, struct.pack("<BBBB", 2, 2, 2, 0)),               # fill value: late allocation, write if set, undefined
                _message(0x0008, struct.pack("<BBQQ", 3, 1, addr, size))]         # layout v3, contiguous
        return self._append(_object_header(msgs))

    def _group(self, members: dict):
        """writes the group's members, heap, symbol nodes, B-tree and object header; returns (header, btree, heap) addresses"""
        entries = []                                   # (name, object header address, cache type, scratch)
        for name in sorted(members, key=lambda s: s.encode()):
            m = members[name]
            if isinstance(m, dict):
                h, bt, hp = self._group(m)
                entries.append((name, h, 1, struct.pack("<QQ", bt, hp)))
            else:
                entries.append((name, self._dataset_header(*m), 0, b"\0" * 16))
        # local heap: offset 0 holds the empty string, every name is 8-byte aligned
        seg = bytearray(b"\0" * 8)
        offs = []
        for name, *_ in entries:
            offs.append(len(seg))
            seg += _pad8(name.encode() + b"\0")
        seg_addr = self._append(bytes(seg))
        heap = self._append(b"HEAP" + struct.pack("<BBBBQQQ", 0, 0, 0, 0, len(seg), 1, seg_addr))   # free-list head 1: none
        # symbol nodes of up to 2*_LEAF_K entries
        cap = 2 * _LEAF_K
        nodes = []                                     # (address, heap offset of the largest name)
        for i in range(0, len(entries), cap):
            chunk = entries[i:i + cap]
            body = b"SNOD" + struct.pack("<BBH", 1, 0, len(chunk))
            for k, (name, addr, ctype, scratch) in enumerate(chunk):
                body += struct.pack("<QQII", offs[i + k], addr, ctype, 0) + scratch
            body += b"\0" * (40 * (cap - len(chunk)))
            nodes.append((self._append(body), offs[i + len(chunk) - 1] if chunk else 0))
        # B-tree levels above them
        level = 0
        bcap = 2 * _INTERNAL_K
        node_size = 24 + (bcap + 1) * 8 + bcap * 8
        if not nodes:                                  # empty group: a root node without children
            body = b"TREE" + struct.pack("<BBHQQ", 0, 0, 0, _UNDEF, _UNDEF)
            nodes = [(self._append(body + b"\0" * (node_size - len(body))), 0)]
        while len(entries) > 0:
            groups = [nodes[i:i + bcap] for i in range(0, len(nodes), bcap)]
            addrs = []
            base = self.pos + (-self.pos % 8)
            for gi in range(len(groups)):
                addrs.append(base + gi * node_size)    # nodes of one level are written back to back (sibling links)
            out = []
            for gi, grp in enumerate(groups):
                left = addrs[gi - 1] if gi > 0 else _UNDEF
                right = addrs[gi + 1] if gi + 1 < len(groups) else _UNDEF
                body = b"TREE" + struct.pack("<BBHQQ", 0, level, len(grp), left, right)
                first_key = 0 if gi == 0 else groups[gi - 1][-1][1]
                body += struct.pack("<Q", first_key)
                for child, maxoff in grp:
                    body += struct.pack("<QQ", child, maxoff)
                body += b"\0" * (node_size - len(body))
                a = self._append(body)
                assert a == addrs[gi]
                out.append((a, grp[-1][1]))
            nodes = out
            level += 1
            if len(nodes) == 1:
                break
        btree = nodes[0][0]
        header = self._append(_object_header([_message(0x0011, struct.pack("<QQ", btree, heap))]))
        return header, btree, heap

    def close(self):
        if self.closed:
            return
        root, btree, heap = self._group(self.tree)
        eof = self.pos
        sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, _LEAF_K, _INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, _UNDEF, eof, _UNDEF)
        sb += struct.pack("<QQII", 0, root, 1, 0) + struct.pack("<QQ", btree, heap)       # root group symbol table entry
        assert len(sb) == 96
        self.f.seek(0)
        self.f.write(sb)
        self.f.close()
        self.closed = True

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

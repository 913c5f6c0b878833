"""Minimal ONNX protobuf codec (no `onnx` package in the image): writes and reads the subset of onnx.proto3 that an opset-13 inference graph
needs -- ModelProto / GraphProto / NodeProto / AttributeProto / TensorProto / ValueInfoProto -- straight at the protobuf wire level.

Why hand-written: `soccdpt_amd/scripts/export_SOccDPT.py` (the counterpart of /root/reference/SOccDPT/scripts/export_SOccDPT.py:122-141, which calls
torch.onnx.export) has no torch forward to trace -- the forward is a launch sequence of HIP kernels -- so the graph is emitted node by node from
the bound weights.  The field numbers below are those of onnx.proto3 (IR version 7 = opset 13); tests/test_onnx_export.py pins them against
graphs serialised by torch's own C++ exporter (torch.onnx internals, which need no `onnx` package)."""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

# TensorProto.DataType
FLOAT, UINT8, INT8, INT32, INT64, BOOL, FLOAT16, DOUBLE = 1, 2, 3, 6, 7, 9, 10, 11
NP_OF = {FLOAT: np.float32, UINT8: np.uint8, INT8: np.int8, INT32: np.int32, INT64: np.int64, BOOL: np.bool_, FLOAT16: np.float16, DOUBLE: np.float64}
DT_OF = {np.dtype(v): k for k, v in NP_OF.items()}
# AttributeProto.AttributeType
A_FLOAT, A_INT, A_STRING, A_TENSOR, A_FLOATS, A_INTS, A_STRINGS = 1, 2, 3, 4, 6, 7, 8


# ---------------- wire level ----------------
def _varint(v: int) -> bytes:
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(fieldno: int, wire: int) -> bytes:
    return _varint((fieldno << 3) | wire)


def _f_varint(fieldno: int, v: int) -> bytes:
    return _key(fieldno, 0) + _varint(int(v))


def _f_bytes(fieldno: int, b: Union[bytes, str]) -> bytes:
    if isinstance(b, str):
        b = b.encode()
    return _key(fieldno, 2) + _varint(len(b)) + b


def _f_float(fieldno: int, v: float) -> bytes:
    return _key(fieldno, 5) + struct.pack("<f", float(v))


def _read_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    shift = v = 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, pos
        shift += 7


def _signed(v: int) -> int:
    return v - (1 << 64) if v >= 1 << 63 else v


def _fields(buf: bytes):
    """Yield (field number, wire type, value) of one message; length-delimited values come back as bytes."""
    pos, n = 0, len(buf)
    while pos < n:
        k, pos = _read_varint(buf, pos)
        fno, wire = k >> 3, k & 7
        if wire == 0:
            v, pos = _read_varint(buf, pos)
        elif wire == 2:
            ln, pos = _read_varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            pos += ln
        elif wire == 5:
            v = struct.unpack_from("<f", buf, pos)[0]
            pos += 4
        elif wire == 1:
            v = struct.unpack_from("<d", buf, pos)[0]
            pos += 8
        else:
            raise ValueError(f"unsupported wire type {wire}")
        yield fno, wire, v


def _packed_varints(v, wire) -> List[int]:
    if wire == 0:
        return [_signed(v)]
    out, pos = [], 0
    while pos < len(v):
        x, pos = _read_varint(v, pos)
        out.append(_signed(x))
    return out


# ---------------- messages ----------------
@dataclass
class Tensor:
    name: str
    array: np.ndarray

    def encode(self) -> bytes:
        a = np.asarray(self.array, order="C")   # (not ascontiguousarray: it would turn a 0-d tensor into shape (1,))
        out = b"".join(_f_varint(1, d) for d in a.shape)
        out += _f_varint(2, DT_OF[a.dtype]) + _f_bytes(8, self.name) + _f_bytes(9, a.tobytes())
        return out

    @staticmethod
    def decode(buf: bytes) -> "Tensor":
        dims, dt, name, raw = [], FLOAT, "", None
        f32, i32, i64, f64 = [], [], [], []
        for fno, wire, v in _fields(buf):
            if fno == 1:
                dims += _packed_varints(v, wire)
            elif fno == 2:
                dt = v
            elif fno == 8:
                name = v.decode()
            elif fno == 9:
                raw = v
            elif fno == 4:
                f32 += list(struct.unpack(f"<{len(v) // 4}f", v)) if wire == 2 else [v]
            elif fno == 5:
                i32 += _packed_varints(v, wire)
            elif fno == 7:
                i64 += _packed_varints(v, wire)
            elif fno == 10:
                f64 += list(struct.unpack(f"<{len(v) // 8}d", v)) if wire == 2 else [v]
        npdt = NP_OF[dt]
        if raw is not None:
            arr = np.frombuffer(raw, dtype=npdt).copy()
        elif f32:
            arr = np.asarray(f32, dtype=npdt)
        elif i64:
            arr = np.asarray(i64, dtype=npdt)
        elif i32:
            arr = np.asarray(i32).astype(npdt)
        elif f64:
            arr = np.asarray(f64, dtype=npdt)
        else:
            arr = np.zeros(0, dtype=npdt)
        return Tensor(name, arr.reshape(dims))


AttrValue = Union[int, float, str, bytes, np.ndarray, Sequence[int], Sequence[float]]


def _encode_attr(name: str, v: AttrValue) -> bytes:
    out = _f_bytes(1, name)
    if isinstance(v, (bool, int, np.integer)):
        return out + _f_varint(3, int(v)) + _f_varint(20, A_INT)
    if isinstance(v, (float, np.floating)):
        return out + _f_float(2, float(v)) + _f_varint(20, A_FLOAT)
    if isinstance(v, (str, bytes)):
        return out + _f_bytes(4, v) + _f_varint(20, A_STRING)
    if isinstance(v, np.ndarray):
        return out + _f_bytes(5, Tensor("", v).encode()) + _f_varint(20, A_TENSOR)
    v = list(v)
    if v and isinstance(v[0], (float, np.floating)):
        return out + b"".join(_f_float(7, x) for x in v) + _f_varint(20, A_FLOATS)
    return out + b"".join(_f_varint(8, int(x)) for x in v) + _f_varint(20, A_INTS)


def _decode_attr(buf: bytes) -> Tuple[str, AttrValue]:
    name, typ = "", 0
    f = i = s = t = None
    floats, ints, strings = [], [], []
    for fno, wire, v in _fields(buf):
        if fno == 1:
            name = v.decode()
        elif fno == 20:
            typ = v
        elif fno == 2:
            f = v
        elif fno == 3:
            i = _signed(v)
        elif fno == 4:
            s = v
        elif fno == 5:
            t = Tensor.decode(v).array
        elif fno == 7:
            floats += list(struct.unpack(f"<{len(v) // 4}f", v)) if wire == 2 else [v]
        elif fno == 8:
            ints += _packed_varints(v, wire)
        elif fno == 9:
            strings.append(v)
    val = {A_FLOAT: f, A_INT: i, A_STRING: (s.decode() if s is not None else None), A_TENSOR: t, A_FLOATS: floats, A_INTS: ints, A_STRINGS: strings}.get(typ)
    if typ == 0:   # producers before IR 3 omit the type: take whichever field is present
        val = next((x for x in (t, s, f, i) if x is not None), ints or floats)
    return name, val


@dataclass
class Node:
    op_type: str
    inputs: List[str]
    outputs: List[str]
    attrs: Dict[str, AttrValue] = field(default_factory=dict)
    name: str = ""

    def encode(self) -> bytes:
        out = b"".join(_f_bytes(1, x) for x in self.inputs) + b"".join(_f_bytes(2, x) for x in self.outputs)
        if self.name:
            out += _f_bytes(3, self.name)
        out += _f_bytes(4, self.op_type)
        out += b"".join(_f_bytes(5, _encode_attr(k, v)) for k, v in self.attrs.items())
        return out

    @staticmethod
    def decode(buf: bytes) -> "Node":
        n = Node("", [], [])
        for fno, wire, v in _fields(buf):
            if fno == 1:
                n.inputs.append(v.decode())
            elif fno == 2:
                n.outputs.append(v.decode())
            elif fno == 3:
                n.name = v.decode()
            elif fno == 4:
                n.op_type = v.decode()
            elif fno == 5:
                k, a = _decode_attr(v)
                n.attrs[k] = a
        return n


@dataclass
class ValueInfo:
    name: str
    elem_type: int
    shape: List[Union[int, str]]     # str = symbolic dimension (dynamic axis), e.g. "batch_size"

    def encode(self) -> bytes:
        dims = b""
        for d in self.shape:
            dim = _f_bytes(2, d) if isinstance(d, str) else _f_varint(1, d)
            dims += _f_bytes(1, dim)
        tensor = _f_varint(1, self.elem_type) + _f_bytes(2, dims)
        return _f_bytes(1, self.name) + _f_bytes(2, _f_bytes(1, tensor))

    @staticmethod
    def decode(buf: bytes) -> "ValueInfo":
        vi = ValueInfo("", FLOAT, [])
        for fno, wire, v in _fields(buf):
            if fno == 1:
                vi.name = v.decode()
            elif fno == 2:
                for f2, _, tv in _fields(v):
                    if f2 != 1:
                        continue
                    for f3, _, x in _fields(tv):
                        if f3 == 1:
                            vi.elem_type = x
                        elif f3 == 2:
                            for f4, _, dim in _fields(x):
                                if f4 != 1:
                                    continue
                                d: Union[int, str] = "?"
                                for f5, _, dv in _fields(dim):
                                    d = _signed(dv) if f5 == 1 else dv.decode()
                                vi.shape.append(d)
        return vi


@dataclass
class Graph:
    name: str = "graph"
    nodes: List[Node] = field(default_factory=list)
    initializers: List[Tensor] = field(default_factory=list)
    inputs: List[ValueInfo] = field(default_factory=list)
    outputs: List[ValueInfo] = field(default_factory=list)

    def encode(self) -> bytes:
        out = b"".join(_f_bytes(1, n.encode()) for n in self.nodes) + _f_bytes(2, self.name)
        out += b"".join(_f_bytes(5, t.encode()) for t in self.initializers)
        out += b"".join(_f_bytes(11, v.encode()) for v in self.inputs) + b"".join(_f_bytes(12, v.encode()) for v in self.outputs)
        return out

    @staticmethod
    def decode(buf: bytes) -> "Graph":
        g = Graph("")
        for fno, wire, v in _fields(buf):
            if fno == 1:
                g.nodes.append(Node.decode(v))
            elif fno == 2:
                g.name = v.decode()
            elif fno == 5:
                g.initializers.append(Tensor.decode(v))
            elif fno == 11:
                g.inputs.append(ValueInfo.decode(v))
            elif fno == 12:
                g.outputs.append(ValueInfo.decode(v))
        return g


@dataclass
class Model:
    graph: Graph
    opset: int = 13
    ir_version: int = 7
    producer_name: str = "soccdpt_amd"
    producer_version: str = "r04"

    def encode(self) -> bytes:
        opset = _f_bytes(1, "") + _f_varint(2, self.opset)
        return (_f_varint(1, self.ir_version) + _f_bytes(2, self.producer_name) + _f_bytes(3, self.producer_version) + _f_bytes(7, self.graph.encode()) +
                _f_bytes(8, opset))

    @staticmethod
    def decode(buf: bytes) -> "Model":
        m = Model(Graph(""))
        for fno, wire, v in _fields(buf):
            if fno == 1:
                m.ir_version = v
            elif fno == 2:
                m.producer_name = v.decode()
            elif fno == 3:
                m.producer_version = v.decode()
            elif fno == 7:
                m.graph = Graph.decode(v)
            elif fno == 8:
                dom, ver = "", 0
                for f2, _, x in _fields(v):
                    if f2 == 1:
                        dom = x.decode()
                    elif f2 == 2:
                        ver = x
                if dom == "":
                    m.opset = ver
        return m


def load(path: str) -> Model:
    with open(path, "rb") as f:
        return Model.decode(f.read())


def save(model: Model, path: str) -> None:
    with open(path, "wb") as f:
        f.write(model.encode())

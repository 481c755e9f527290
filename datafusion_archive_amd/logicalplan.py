"""Input vocabulary of the execution path: ``Expr`` / ``Operator`` / ``ScalarValue`` / ``DataType``.

Mirrors the reference's ``src/logicalplan.rs`` (Operator :67-84, ScalarValue :96-132,
Expr :136-167, Debug formatting :264-309) closely enough that plans produced by the (untouched)
SQL planner can be restated 1:1, and serialises a tree to the ``dfx_expr_node`` array the C ABI
takes (include/dfx.h).  Pure host-side data; no device code here.
"""
from __future__ import annotations

import ctypes
import enum
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple


class DataType(enum.IntEnum):
    """arrow::datatypes::DataType subset (values == dfx_dtype in include/dfx.h)."""

    Null = 0
    Boolean = 1
    Int8 = 2
    Int16 = 3
    Int32 = 4
    Int64 = 5
    UInt8 = 6
    UInt16 = 7
    UInt32 = 8
    UInt64 = 9
    Float32 = 10
    Float64 = 11
    Utf8 = 12

    def __repr__(self) -> str:  # Rust {:?}
        return self.name


class Operator(enum.IntEnum):
    """logicalplan::Operator (src/logicalplan.rs:67-84), same order."""

    Eq = 0
    NotEq = 1
    Lt = 2
    LtEq = 3
    Gt = 4
    GtEq = 5
    Plus = 6
    Minus = 7
    Multiply = 8
    Divide = 9
    Modulus = 10
    And = 11
    Or = 12
    Not = 13
    Like = 14
    NotLike = 15

    def __repr__(self) -> str:
        return self.name


_SIGNED = (DataType.Int8, DataType.Int16, DataType.Int32, DataType.Int64)
_UNSIGNED = (DataType.UInt8, DataType.UInt16, DataType.UInt32, DataType.UInt64)


def _rust_float_debug(x: float) -> str:
    """Rust ``{:?}`` of an f64: shortest round-trip, always with a fractional part."""
    if x != x:
        return "NaN"
    if x in (float("inf"), float("-inf")):
        return "inf" if x > 0 else "-inf"
    s = repr(float(x))
    if "e" in s or "E" in s:
        mant, exp = s.lower().split("e")
        return f"{mant}e{int(exp)}"
    return s


@dataclass(frozen=True)
class ScalarValue:
    """logicalplan::ScalarValue (src/logicalplan.rs:96-132)."""

    data_type: DataType
    value: object = None

    @staticmethod
    def Float64(v: float) -> "ScalarValue":
        return ScalarValue(DataType.Float64, float(v))

    @staticmethod
    def Float32(v: float) -> "ScalarValue":
        return ScalarValue(DataType.Float32, float(v))

    @staticmethod
    def Int8(v: int) -> "ScalarValue":
        return ScalarValue(DataType.Int8, int(v))

    @staticmethod
    def Int16(v: int) -> "ScalarValue":
        return ScalarValue(DataType.Int16, int(v))

    @staticmethod
    def Int32(v: int) -> "ScalarValue":
        return ScalarValue(DataType.Int32, int(v))

    @staticmethod
    def Int64(v: int) -> "ScalarValue":
        return ScalarValue(DataType.Int64, int(v))

    @staticmethod
    def UInt8(v: int) -> "ScalarValue":
        return ScalarValue(DataType.UInt8, int(v))

    @staticmethod
    def UInt16(v: int) -> "ScalarValue":
        return ScalarValue(DataType.UInt16, int(v))

    @staticmethod
    def UInt32(v: int) -> "ScalarValue":
        return ScalarValue(DataType.UInt32, int(v))

    @staticmethod
    def UInt64(v: int) -> "ScalarValue":
        return ScalarValue(DataType.UInt64, int(v))

    @staticmethod
    def Boolean(v: bool) -> "ScalarValue":
        return ScalarValue(DataType.Boolean, bool(v))

    @staticmethod
    def Utf8(v: str) -> "ScalarValue":
        return ScalarValue(DataType.Utf8, str(v))

    @staticmethod
    def Null() -> "ScalarValue":
        return ScalarValue(DataType.Null, None)

    def get_datatype(self) -> DataType:
        return self.data_type

    def __repr__(self) -> str:
        if self.data_type == DataType.Null:
            return "Null"
        if self.data_type in (DataType.Float32, DataType.Float64):
            return f"{self.data_type.name}({_rust_float_debug(self.value)})"
        if self.data_type == DataType.Boolean:
            return f"Boolean({'true' if self.value else 'false'})"
        if self.data_type == DataType.Utf8:
            return f'Utf8("{self.value}")'
        return f"{self.data_type.name}({self.value})"


class Expr:
    """logicalplan::Expr (src/logicalplan.rs:136-167). Sub-classes are the variants."""

    # fluent helpers, same names as the reference (logicalplan.rs:216-262)
    def eq(self, other: "Expr") -> "Expr":
        return BinaryExpr(self, Operator.Eq, other)

    def not_eq(self, other: "Expr") -> "Expr":
        return BinaryExpr(self, Operator.NotEq, other)

    def gt(self, other: "Expr") -> "Expr":
        return BinaryExpr(self, Operator.Gt, other)

    def gt_eq(self, other: "Expr") -> "Expr":
        return BinaryExpr(self, Operator.GtEq, other)

    def lt(self, other: "Expr") -> "Expr":
        return BinaryExpr(self, Operator.Lt, other)

    def lt_eq(self, other: "Expr") -> "Expr":
        return BinaryExpr(self, Operator.LtEq, other)

    # not in the reference, conveniences for tests
    def and_(self, other: "Expr") -> "Expr":
        return BinaryExpr(self, Operator.And, other)

    def or_(self, other: "Expr") -> "Expr":
        return BinaryExpr(self, Operator.Or, other)

    def plus(self, other: "Expr") -> "Expr":
        return BinaryExpr(self, Operator.Plus, other)

    def minus(self, other: "Expr") -> "Expr":
        return BinaryExpr(self, Operator.Minus, other)

    def multiply(self, other: "Expr") -> "Expr":
        return BinaryExpr(self, Operator.Multiply, other)

    def divide(self, other: "Expr") -> "Expr":
        return BinaryExpr(self, Operator.Divide, other)

    def cast(self, data_type: DataType) -> "Expr":
        return Cast(self, data_type)


@dataclass(frozen=True, repr=False)
class Column(Expr):
    index: int

    def __repr__(self) -> str:
        return f"#{self.index}"


@dataclass(frozen=True, repr=False)
class Literal(Expr):
    value: ScalarValue

    def __repr__(self) -> str:
        return repr(self.value)


@dataclass(frozen=True, repr=False)
class BinaryExpr(Expr):
    left: Expr
    op: Operator
    right: Expr

    def __repr__(self) -> str:
        return f"{self.left!r} {self.op!r} {self.right!r}"


@dataclass(frozen=True, repr=False)
class IsNotNull(Expr):
    expr: Expr

    def __repr__(self) -> str:
        return f"{self.expr!r} IS NOT NULL"


@dataclass(frozen=True, repr=False)
class IsNull(Expr):
    expr: Expr

    def __repr__(self) -> str:
        return f"{self.expr!r} IS NULL"


@dataclass(frozen=True, repr=False)
class Cast(Expr):
    expr: Expr
    data_type: DataType

    def __repr__(self) -> str:
        return f"CAST({self.expr!r} AS {self.data_type!r})"


@dataclass(frozen=True, repr=False)
class Sort(Expr):
    expr: Expr
    asc: bool = True

    def __repr__(self) -> str:
        return f"{self.expr!r} {'ASC' if self.asc else 'DESC'}"


@dataclass(frozen=True, repr=False)
class ScalarFunction(Expr):
    name: str
    args: Tuple[Expr, ...]
    return_type: DataType

    def __repr__(self) -> str:
        return f"{self.name}({', '.join(repr(a) for a in self.args)})"


@dataclass(frozen=True, repr=False)
class AggregateFunction(Expr):
    name: str
    args: Tuple[Expr, ...]
    return_type: DataType

    def __init__(self, name: str, args: Sequence[Expr], return_type: DataType):
        object.__setattr__(self, "name", name)
        object.__setattr__(self, "args", tuple(args))
        object.__setattr__(self, "return_type", return_type)

    def __repr__(self) -> str:
        return f"{self.name}({', '.join(repr(a) for a in self.args)})"


# ---------------------------------------------------------------------------------------------
# serialisation to the C ABI's dfx_expr_node (include/dfx.h)
# ---------------------------------------------------------------------------------------------
class _Lit(ctypes.Union):
    _fields_ = [("i64", ctypes.c_int64), ("u64", ctypes.c_uint64), ("f64", ctypes.c_double),
                ("f32", ctypes.c_float)]


class ExprNode(ctypes.Structure):
    """ctypes image of ``dfx_expr_node``."""

    _fields_ = [
        ("kind", ctypes.c_int32),
        ("op", ctypes.c_int32),
        ("dtype", ctypes.c_int32),
        ("left", ctypes.c_int32),
        ("right", ctypes.c_int32),
        ("column", ctypes.c_int32),
        ("n_args", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
        ("lit", _Lit),
        ("name", ctypes.c_char_p),
    ]


KIND_COLUMN, KIND_LITERAL, KIND_BINARY, KIND_IS_NOT_NULL, KIND_IS_NULL = 0, 1, 2, 3, 4
KIND_CAST, KIND_SORT, KIND_SCALAR_FUNCTION, KIND_AGGREGATE_FUNCTION = 5, 6, 7, 8


@dataclass
class SerializedExprs:
    """A forest of expressions flattened into one node array (children before parents)."""

    nodes: ctypes.Array
    roots: List[int]
    _keepalive: list = field(default_factory=list)

    @property
    def n_nodes(self) -> int:
        return len(self.nodes)


def serialize(exprs: Sequence[Expr]) -> SerializedExprs:
    flat: List[ExprNode] = []
    keep: list = []

    def emit(e: Expr) -> int:
        n = ExprNode()
        n.left = n.right = -1
        n.column = -1
        if isinstance(e, Column):
            n.kind = KIND_COLUMN
            n.column = e.index
        elif isinstance(e, Literal):
            n.kind = KIND_LITERAL
            dt = e.value.data_type
            n.dtype = int(dt)
            if dt == DataType.Float64:
                n.lit.f64 = float(e.value.value)
            elif dt == DataType.Float32:
                n.lit.f32 = float(e.value.value)
            elif dt in _SIGNED:
                n.lit.i64 = int(e.value.value)
            elif dt in _UNSIGNED:
                n.lit.u64 = int(e.value.value)
            elif dt == DataType.Boolean:
                n.lit.u64 = 1 if e.value.value else 0
            elif dt == DataType.Utf8:
                b = str(e.value.value).encode()
                keep.append(b)
                n.name = b
        elif isinstance(e, BinaryExpr):
            l = emit(e.left)
            r = emit(e.right)
            n.kind = KIND_BINARY
            n.op = int(e.op)
            n.left, n.right = l, r
        elif isinstance(e, (IsNotNull, IsNull, Sort)):
            c = emit(e.expr)
            n.kind = {IsNotNull: KIND_IS_NOT_NULL, IsNull: KIND_IS_NULL, Sort: KIND_SORT}[type(e)]
            n.left = c
        elif isinstance(e, Cast):
            c = emit(e.expr)
            n.kind = KIND_CAST
            n.dtype = int(e.data_type)
            n.left = c
        elif isinstance(e, (ScalarFunction, AggregateFunction)):
            kids = [emit(a) for a in e.args]
            n.kind = KIND_AGGREGATE_FUNCTION if isinstance(e, AggregateFunction) else KIND_SCALAR_FUNCTION
            n.dtype = int(e.return_type)
            n.n_args = len(kids)
            n.left = kids[0] if kids else -1
            b = e.name.encode()
            keep.append(b)
            n.name = b
        else:
            raise TypeError(f"not an Expr: {e!r}")
        flat.append(n)
        return len(flat) - 1

    roots = [emit(e) for e in exprs]
    arr = (ExprNode * max(1, len(flat)))(*flat)
    return SerializedExprs(arr, roots, keep)


# convenience constructors used by tests and bench (what the SQL planner would emit)
def col(i: int) -> Column:
    return Column(i)


def lit_f64(v: float) -> Literal:
    return Literal(ScalarValue.Float64(v))


def lit_i64(v: int) -> Literal:
    return Literal(ScalarValue.Int64(v))


def aggregate(name: str, arg: Expr, return_type: DataType) -> AggregateFunction:
    return AggregateFunction(name, (arg,), return_type)

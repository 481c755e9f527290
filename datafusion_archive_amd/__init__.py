"""datafusion_archive_amd -- MI355X-native filter / projection / aggregate execution path.

A drop-in for ``src/execution::{filter,projection,aggregate}`` and the RecordBatch expression
evaluator of andygrove/datafusion-archive (DataFusion 0.6.0), built from scratch as hand-written
gfx950 HIP kernels behind a C ABI (``include/dfx.h``).  See DESIGN.md and INTEGRATION.md.

Layout: ``csrc/`` HIP kernels + C-ABI host code, ``logicalplan.py`` the Expr vocabulary,
``execution.py`` the Python mirror of the reference's operator interface (ctypes plumbing only).
"""
from .logicalplan import (AggregateFunction, BinaryExpr, Cast, Column, DataType, Expr, IsNotNull, IsNull,
                          Literal, Operator, ScalarFunction, ScalarValue, Sort)

__all__ = ["AggregateFunction", "BinaryExpr", "Cast", "Column", "DataType", "Expr", "IsNotNull", "IsNull",
           "Literal", "Operator", "ScalarFunction", "ScalarValue", "Sort", "execution", "logicalplan"]

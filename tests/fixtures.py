"""Shared test helpers: the reference's CSV fixtures and its golden-string formatter.

`load_csv` reproduces CsvDataSource::new (src/execution/datasource.rs:39-43): the arrow csv
reader is always created with has_headers=true, so the FIRST LINE IS DROPPED even when the file
has no header (uk_cities.csv: 37 lines -> 36 rows).
`result_str` reproduces tests/sql.rs:99-137 (Rust `{:?}` of each value, tab separated).
"""
import os
from typing import Iterable, List

import pyarrow as pa
import pyarrow.csv as pacsv

from datafusion_archive_amd.logicalplan import _rust_float_debug

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def load_csv(name: str, schema: pa.Schema, batch_size: int = 1024) -> List[pa.RecordBatch]:
    table = pacsv.read_csv(
        os.path.join(DATA, name),
        read_options=pacsv.ReadOptions(column_names=schema.names, skip_rows=1),
        convert_options=pacsv.ConvertOptions(column_types={f.name: f.type for f in schema}),
    )
    batches = table.combine_chunks().to_batches(max_chunksize=batch_size)
    return batches if batches else [pa.RecordBatch.from_pylist([], schema=schema)]


def uk_cities_schema() -> pa.Schema:
    return pa.schema([pa.field("city", pa.string(), False), pa.field("lat", pa.float64(), False),
                      pa.field("lng", pa.float64(), False)])


def aggr_test_schema(key_type=pa.int32()) -> pa.Schema:
    return pa.schema([pa.field("a", key_type, False), pa.field("b", pa.float64(), False)])


def result_str(batches: Iterable[pa.RecordBatch]) -> str:
    out = []
    for batch in batches:
        cols = [batch.column(i) for i in range(batch.num_columns)]
        for r in range(batch.num_rows):
            cells = []
            for c in cols:
                v = c[r].as_py()
                if pa.types.is_floating(c.type):
                    cells.append(_rust_float_debug(v))
                elif pa.types.is_string(c.type):
                    cells.append('"' + v + '"')
                elif pa.types.is_int32(c.type):
                    cells.append(str(v))
                else:
                    cells.append("???")
            out.append("\t".join(cells) + "\n")
    return "".join(out)

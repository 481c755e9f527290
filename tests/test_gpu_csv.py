"""GPU parity of the CSV source: dfx_csv_datasource_new == CsvDataSource::new
(src/execution/datasource.rs:33-58) as restated by the oracle, on the reference's own fixtures, on generated numeric
text (bit-exact floats: Rust's parse is correctly rounded), on adversarial quoting, and end to end through
Filter / Project / Aggregate (BASELINE config 0: examples/csv_sql.rs)."""
import os
import sys
import time

import numpy as np
import pyarrow as pa
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fixtures  # noqa: E402
import oracle  # noqa: E402
from gpu_util import assert_batches_identical  # noqa: E402
from test_csv_oracle import ALL_TYPES, CASES, DATA  # noqa: E402

from datafusion_archive_amd import execution as ex  # noqa: E402
from datafusion_archive_amd.logicalplan import (AggregateFunction, BinaryExpr, Column, DataType, Literal, Operator,  # noqa: E402
                                                ScalarValue)

pytestmark = pytest.mark.gpu


def lit(v):
    return Literal(ScalarValue.Float64(float(v)))


def check_file(path, schema, batch_size):
    got = list(ex.CsvDataSource(path, schema, batch_size))
    want = oracle.read_csv(path, schema, batch_size)
    assert len(got) == len(want), (len(got), len(want))
    for i, (g, w) in enumerate(zip(got, want)):
        assert_batches_identical(g, w, f"{os.path.basename(path)} batch {i}")
    return sum(b.num_rows for b in got)


@pytest.mark.parametrize("batch_size", [1024, 7])
@pytest.mark.parametrize("name,schema", CASES)
def test_csv_fixtures_match_oracle(name, schema, batch_size):
    check_file(os.path.join(DATA, name), schema, batch_size)


def test_csv_sql_example_end_to_end():
    """examples/csv_sql.rs / tests/sql.rs:29-37: SELECT city, lat, lng, lat + lng FROM cities WHERE lat > 51 AND lat < 53,
    CSV text in, golden strings out, nothing but the result leaves the device."""
    schema = fixtures.uk_cities_schema()
    src = ex.CsvDataSource(os.path.join(DATA, "uk_cities.csv"), schema, 1024)
    pred = BinaryExpr(BinaryExpr(Column(1), Operator.Gt, lit(51.0)), Operator.And, BinaryExpr(Column(1), Operator.Lt, lit(53.0)))
    rel = ex.FilterRelation(src, ex.compile_scalar_expr(None, pred, schema), schema)
    exprs = [Column(0), Column(1), Column(2), BinaryExpr(Column(1), Operator.Plus, Column(2))]
    rel = ex.ProjectRelation(rel, [ex.compile_scalar_expr(None, e, schema) for e in exprs], None)
    got = fixtures.result_str(list(rel))
    want_batches = fixtures.load_csv("uk_cities.csv", schema)
    want = fixtures.result_str([oracle.project_next(exprs, oracle.filter_next(pred, b)) for b in want_batches])
    assert got == want
    assert '"Solihull, Birmingham, UK"\t52.412811\t-1.778197\t50.634614\n' in got
    assert got.count("\n") == 18  # tests/sql.rs:29-37 lists 18 cities


def test_csv_into_group_by():
    """tests/sql.rs:39-52 shape: CSV -> GROUP BY with MIN/MAX/SUM, all on the device."""
    schema = fixtures.aggr_test_schema()
    f64 = DataType.Float64
    aggs = [AggregateFunction("MIN", [Column(1)], f64), AggregateFunction("MAX", [Column(1)], f64),
            AggregateFunction("SUM", [Column(1)], f64)]
    src = ex.CsvDataSource(os.path.join(DATA, "aggregate_test_1.csv"), schema, 1024)
    rel = ex.AggregateRelation(None, src, [ex.compile_scalar_expr(None, Column(0), schema)],
                               [ex.compile_expr(None, a, schema) for a in aggs])
    out = rel.next()
    rows = sorted(zip(*[out.column(i).to_pylist() for i in range(4)]))
    assert rows == [(1, 1.1, 2.2, 3.3000000000000003), (2, 3.3, 5.5, 13.2), (3, 1.0, 2.0, 3.0)]  # aggregate.rs:1033-1127


def _write_numeric_csv(path, n, seed, crlf=False, trailing_newline=True):
    rng = np.random.default_rng(seed)
    f = rng.standard_normal(n) * 10.0 ** rng.integers(-20, 20, n)
    g = rng.random(n).astype(np.float32)
    i = rng.integers(-2**62, 2**62, n)
    u = rng.integers(0, 256, n)
    nl = "\r\n" if crlf else "\n"
    lines = ["f64,f32,i64,u8,txt"]
    for r in range(n):
        cells = [repr(float(f[r])), repr(float(g[r])), str(int(i[r])), str(int(u[r])), f"row {r}"]
        if r % 13 == 5:
            cells[0] = ""          # empty primitive cell -> null
        if r % 17 == 3:
            cells[2] = '"' + cells[2] + '"'  # quoted number
        if r % 11 == 7:
            cells[4] = '"quoted, with ""quotes"" and\nnewline ' + str(r) + '"'
        if r % 19 == 2:
            cells[1] = "%.3e" % float(g[r])
        lines.append(",".join(cells))
        if r % 97 == 0:
            lines.append("")       # blank line
    text = nl.join(lines) + (nl if trailing_newline else "")
    with open(path, "w", newline="") as fh:
        fh.write(text)
    return pa.schema([("f64", pa.float64()), ("f32", pa.float32()), ("i64", pa.int64()), ("u8", pa.uint8()), ("txt", pa.string())])


@pytest.mark.parametrize("crlf,trailing", [(False, True), (True, True), (False, False)])
def test_csv_generated_numeric_text_is_bit_exact(tmp_path, crlf, trailing):
    p = str(tmp_path / "gen.csv")
    schema = _write_numeric_csv(p, 20000, 5, crlf, trailing)
    assert check_file(p, schema, 4096) == 20000
    assert check_file(p, schema, 100000) == 20000


def _write_plain_csv(path, n, seed, nl="\n", trailing=True, quote_rows=(), long_rows=(), blank_every=0):
    """Numeric text without quotes (the wave-cooperative cell path) except in `quote_rows`; `long_rows` carry a text cell
    longer than any LDS window."""
    rng = np.random.default_rng(seed)
    f = rng.standard_normal(n) * 10.0 ** rng.integers(-30, 30, n)
    g = rng.random(n).astype(np.float32)
    i = rng.integers(-2**63, 2**63 - 1, n)
    u = rng.integers(0, 65536, n)
    lines = ["f64,f32,i64,u16,flag,txt"]
    for r in range(n):
        cells = [repr(float(f[r])), repr(float(g[r])), str(int(i[r])), str(int(u[r])), "true" if r % 3 else "false", f"r{r}"]
        if r % 13 == 5:
            cells[0] = ""
        if r % 29 == 4:
            cells[4] = ""
        if r % 31 == 9:
            cells[5] = ""
        if r % 19 == 2:
            cells[1] = "%.4E" % float(g[r])
        if r % 23 == 1:
            cells[0] = "+" + repr(abs(float(f[r])))
        if r in quote_rows:
            cells[2] = '"' + cells[2] + '"'
        if r in long_rows:
            cells[5] = "x" * 20000
        lines.append(",".join(cells))
        if blank_every and r % blank_every == 0:
            lines.append("")
    with open(path, "w", newline="") as fh:
        fh.write(nl.join(lines) + (nl if trailing else ""))
    return pa.schema([("f64", pa.float64()), ("f32", pa.float32()), ("i64", pa.int64()), ("u16", pa.uint16()), ("flag", pa.bool_()),
                      ("txt", pa.string())])


@pytest.mark.parametrize("nl,trailing,blank_every", [("\n", True, 0), ("\r\n", True, 97), ("\r", False, 0), ("\n", False, 61)])
def test_csv_wave_tiles_plain_text_is_bit_exact(tmp_path, nl, trailing, blank_every):
    """Quote-free numeric text takes the wave-cooperative path of k_csv_parse (an LDS copy of a 64-record tile, SWAR
    delimiter masks, a structural list): every tile, whatever the terminator, the blank lines or the missing last
    terminator -- and converts to the same bits as the oracle and as the per-lane walk (csv.wave_tiles = 0)."""
    p = str(tmp_path / "plain.csv")
    schema = _write_plain_csv(p, 20000, 41, nl, trailing, blank_every=blank_every)
    for batch in (4096, 1000, 100000):
        ex.counter_reset()
        assert check_file(p, schema, batch) == 20000
        assert ex.counter_get("csv_tiles") == sum((min(batch, 20000 - o) + 63) // 64 for o in range(0, 20000, batch))
        assert ex.counter_get("csv_general_tiles") == 0
    ex.set_option("csv.wave_tiles", 0)
    try:
        ex.counter_reset()
        assert check_file(p, schema, 4096) == 20000
        assert ex.counter_get("csv_general_tiles") == ex.counter_get("csv_tiles") > 0
    finally:
        ex.set_option("csv.wave_tiles", 1)


def test_csv_wave_tiles_fall_back_per_tile(tmp_path):
    """A quote or a record longer than the LDS window sends ONLY its own tile down the per-lane walk."""
    p = str(tmp_path / "mixed.csv")
    schema = _write_plain_csv(p, 6400, 43, quote_rows={100, 101, 3000}, long_rows={5000})
    ex.counter_reset()
    assert check_file(p, schema, 6400) == 6400
    assert ex.counter_get("csv_tiles") == 100
    assert ex.counter_get("csv_general_tiles") == 3  # tiles 1, 46 and 78
    # a schema narrower / wider than the file: cells beyond the schema are ignored, columns beyond the record are null
    narrow = pa.schema(list(schema)[:3])
    wide = pa.schema(list(schema) + [pa.field("more", pa.int64())])
    for sch in (narrow, wide):
        ex.counter_reset()
        assert check_file(p, sch, 4096) == 6400
        assert ex.counter_get("csv_general_tiles") == 3


def test_csv_wave_tiles_errors(tmp_path):
    """Errors out of the wave-cooperative path: a cell that does not parse is reported from it; a ragged record makes its
    tile take the walk, which reports UnequalLengths for the lowest record as the reference does."""
    schema = pa.schema([("a", pa.int32()), ("b", pa.float64()), ("c", pa.int64())])
    rows = [f"{i},{i}.25,{i * 7}" for i in range(1000)]
    p = tmp_path / "e.csv"
    cases = []
    r = list(rows); r[700] = "700,1e,4900"
    cases.append(("Error while parsing value 1e at line 701", r))
    r = list(rows); r[650] = "650,9.5,zz"; r[130] = "130,--1,910"
    cases.append(("Error while parsing value --1 at line 131", r))
    r = list(rows); r[400] = "400,400.25"; r[100] = "bad,1.0,700"
    cases.append(("UnequalLengths", r))
    r = list(rows); r[999] = "999,999.25,6993,1"
    cases.append(("UnequalLengths", r))
    r = list(rows); r[320] = "320,0.1,99999999999999999999"
    cases.append(("Error while parsing value 99999999999999999999 at line 321", r))
    for want, body in cases:
        p.write_text("a,b,c\n" + "\n".join(body) + "\n")
        with pytest.raises(ex.ExecutionError) as ei:
            list(ex.CsvDataSource(str(p), schema, 4096))
        assert want in ei.value.message, (want, ei.value.message)
        with pytest.raises(oracle.OracleError) as oi:
            oracle.read_csv(str(p), schema, 4096)
        assert want in oi.value.message, (want, oi.value.message)


def test_csv_quoting_fuzz_against_oracle(tmp_path):
    """Random text over the bytes that matter to the automaton ( , " CR LF and fillers): boundaries by parallel DFA
    simulation on the device must equal the oracle's byte-at-a-time reader.  Records keep a fixed field count so that
    most files are accepted; quotes are placed anywhere."""
    rng = np.random.default_rng(123)
    s3 = pa.schema([("a", pa.string()), ("b", pa.string()), ("c", pa.string())])
    pieces = ['"', '""', "a", "bc", " ", "1", "é", '"x"', '","', '"\n"', "'"]
    agree = errors = 0
    for trial in range(60):
        lines = ["h1,h2,h3"]
        for _ in range(int(rng.integers(1, 400))):
            cells = []
            for _c in range(3):
                k = int(rng.integers(0, 5))
                cell = "".join(pieces[int(j)] for j in rng.integers(0, len(pieces), k))
                if rng.random() < 0.3:
                    cell = '"' + cell.replace('"', '""') + '"'
                elif trial % 4:  # keep the field count: no bare delimiters / terminators, no opening quote
                    cell = cell.replace(",", ";").replace("\n", "|")
                    if cell.startswith('"'):
                        cell = "u" + cell
                cells.append(cell)
            lines.append(",".join(cells))
            if rng.random() < 0.1:
                lines.append("")
        nl = "\r\n" if trial % 3 == 0 else ("\r" if trial % 3 == 1 else "\n")
        p = str(tmp_path / f"fuzz{trial}.csv")
        with open(p, "w", newline="", encoding="utf-8") as fh:
            fh.write(nl.join(lines) + (nl if trial % 2 else ""))
        try:
            want = oracle.read_csv(p, s3, 64)
        except oracle.OracleError as e:
            with pytest.raises(ex.ExecutionError) as ei:
                list(ex.CsvDataSource(p, s3, 64))
            assert ("UnequalLengths" in str(e)) == ("UnequalLengths" in ei.value.message), (str(e), ei.value.message)
            errors += 1
            continue
        got = list(ex.CsvDataSource(p, s3, 64))
        assert len(got) == len(want), trial
        for g, w in zip(got, want):
            assert_batches_identical(g, w, f"fuzz {trial}")
        agree += 1
    print(f"csv fuzz: {agree} files identical, {errors} rejected by both")
    assert agree >= 5


def test_csv_errors_mirror_reference(tmp_path):
    p = tmp_path / "bad.csv"
    schema = pa.schema([("a", pa.int32()), ("b", pa.float64())])
    p.write_text("a,b\n1,2.5\n2,abc\n3,4\n")
    with pytest.raises(ex.ExecutionError) as ei:
        list(ex.CsvDataSource(str(p), schema, 1024))
    assert ei.value.kind == "ArrowError" and ei.value.message.endswith("Error while parsing value abc at line 2")
    p.write_text("a,b\n1,2\n300000000000,4\n")  # i32 overflow
    with pytest.raises(ex.ExecutionError) as ei:
        list(ex.CsvDataSource(str(p), schema, 1024))
    assert "Error while parsing value 300000000000 at line 2" in ei.value.message
    p.write_text("a,b\n1,2\n3\n")
    with pytest.raises(ex.ExecutionError) as ei:
        list(ex.CsvDataSource(str(p), schema, 1024))
    assert ei.value.kind == "ArrowError" and "UnequalLengths" in ei.value.message
    with pytest.raises(ex.ExecutionError) as ei:  # File::open(filename).unwrap()
        ex.CsvDataSource(str(tmp_path / "missing.csv"), schema, 1024)
    assert ei.value.kind == "InternalError"
    p.write_text("")  # empty file: no header, no rows
    assert list(ex.CsvDataSource(str(p), schema, 1024)) == []
    p.write_text("only,header\n")
    assert list(ex.CsvDataSource(str(p), schema, 1024)) == []


def test_csv_first_error_of_a_batch_follows_arrows_order(tmp_path):
    """Two bad things in ONE batch: arrow's csv::Reader reads all records of the batch first (UnequalLengths wins over
    any cell error, lowest record first) and then converts column by column (lowest column, then lowest row).  The
    device reduces the failing cells with one atomicMin whose key has that order; the oracle reads the same way."""
    schema = pa.schema([("a", pa.int32()), ("b", pa.float64()), ("c", pa.int64())])
    rows = [f"{i},{i}.5,{i * 7}" for i in range(40)]
    cases = []
    r = list(rows); r[5] = "5,5.5,zzz"; r[9] = "9,yyy,63"          # row 5 col 2 vs row 9 col 1: the lower COLUMN wins
    cases.append(("Error while parsing value yyy at line 10", r))
    r = list(rows); r[3] = "xxx,3.5,21"; r[7] = "7,7.5"             # a parse error at row 3 vs a short record at row 7
    cases.append(("UnequalLengths", r))
    r = list(rows); r[20] = "20,q,140"; r[4] = "4,w,28"             # same column: the lower row wins
    cases.append(("Error while parsing value w at line 5", r))
    for want, body in cases:
        p = tmp_path / "two_bad.csv"
        p.write_text("a,b,c\n" + "\n".join(body) + "\n")
        with pytest.raises(ex.ExecutionError) as ei:
            list(ex.CsvDataSource(str(p), schema, 1024))
        assert want in ei.value.message, (want, ei.value.message)
        with pytest.raises(oracle.OracleError) as oi:
            oracle.read_csv(str(p), schema, 1024)
        assert want in oi.value.message, (want, oi.value.message)
    # in different batches the earlier batch fails first, whatever the columns
    r = list(rows); r[5] = "5,5.5,zzz"; r[25] = "yy,25.5,175"
    p = tmp_path / "two_batches.csv"
    p.write_text("a,b,c\n" + "\n".join(r) + "\n")
    with pytest.raises(ex.ExecutionError) as ei:
        list(ex.CsvDataSource(str(p), schema, 16))
    assert "Error while parsing value zzz at line 6" in ei.value.message


def test_csv_large_file_throughput(tmp_path):
    """~60 MB of numeric text: parity on a slice-independent property (column sums via the device aggregate equal the
    oracle's sums of the same file) and the device-side parse rate."""
    n = 1_000_000
    rng = np.random.default_rng(9)
    k = rng.integers(0, 1000, n)
    v = rng.integers(0, 2**20, n).astype(np.float64) * 2.0 ** -10
    w = rng.integers(-10**9, 10**9, n)
    p = str(tmp_path / "big.csv")
    with open(p, "w") as fh:
        fh.write("k,v,w,tag\n")
        fh.write("\n".join(f"{int(a)},{float(b)!r},{int(c)},t{int(a) % 7}" for a, b, c in zip(k, v, w)))
        fh.write("\n")
    size = os.path.getsize(p)
    schema = pa.schema([("k", pa.int64()), ("v", pa.float64()), ("w", pa.int64()), ("tag", pa.string())])
    t0 = time.perf_counter()
    src = ex.CsvDataSource(p, schema, 1 << 20)
    f64 = DataType.Float64
    aggs = [AggregateFunction("SUM", [Column(1)], f64), AggregateFunction("SUM", [Column(2)], DataType.Int64),
            AggregateFunction("COUNT", [Column(0)], DataType.UInt64)]
    out = ex.AggregateRelation(None, src, [], [ex.compile_expr(None, a, schema) for a in aggs]).next()
    dt = time.perf_counter() - t0
    assert out.column(0)[0].as_py() == float(np.sum(v))  # exact data: order-independent
    assert out.column(1)[0].as_py() == int(np.sum(w))
    assert out.column(2)[0].as_py() == n
    print(f"csv ingest + aggregate: {size / 1e6:.1f} MB in {dt * 1e3:.1f} ms = {size / dt / 1e9:.2f} GB/s end to end (file read + H2D included)")


def test_csv_projection_pushdown_converts_only_referenced_columns(tmp_path):
    """CSV -> aggregate over 2 of 5 columns: cells of the other columns are located but never converted (what arrow's
    csv::Reader does when it is given a projection)."""
    p = str(tmp_path / "gen.csv")
    schema = _write_numeric_csv(p, 5000, 8)
    aggs = [AggregateFunction("MAX", [Column(3)], DataType.UInt8), AggregateFunction("COUNT", [Column(2)], DataType.UInt64)]
    ex.counter_reset()
    out = ex.AggregateRelation(None, ex.CsvDataSource(p, schema, 2048), [], [ex.compile_expr(None, a, schema) for a in aggs]).next()
    assert ex.counter_get("csv_cells") == 2 * 5000
    want = oracle.aggregate([], aggs, oracle.read_csv(p, schema, 2048))
    assert_batches_identical(out, want, "csv push-down")


def test_csv_all_seven_reference_goldens_from_text():
    """The reference's golden queries with the CSV file itself as the input of the device pipeline (its tests read their
    data through CsvDataSource): tests/sql.rs:54-67 (GROUP BY a Utf8 column of the file), :69-77 (CAST), and the
    ORDER BY it could not run -- deterministic output of the GROUP BY."""
    schema = pa.schema([pa.field("a", pa.string(), False), pa.field("b", pa.float64(), False)])
    f64 = DataType.Float64
    aggs = [AggregateFunction("MIN", [Column(1)], f64), AggregateFunction("MAX", [Column(1)], f64)]
    src = ex.CsvDataSource(os.path.join(DATA, "aggregate_test_2.csv"), schema, 1024)
    agg = ex.AggregateRelation(None, src, [ex.compile_scalar_expr(None, Column(0), schema)], [ex.compile_expr(None, a, schema) for a in aggs])
    out_schema = agg.schema()
    srt = ex.SortRelation(agg, [(ex.compile_scalar_expr(None, Column(1), out_schema), True)], out_schema)  # ORDER BY MIN(b)
    got = fixtures.result_str(list(srt))
    assert got == '"three"\t1.0\t2.0\n"one"\t1.1\t2.2\n"two"\t3.3\t5.5\n'  # tests/sql.rs:61-66, ordered by the first aggregate
    # tests/sql.rs:69-77: SELECT CAST(c2 AS int) -> f64 -> i32 truncation, from the cities file
    from datafusion_archive_amd.logicalplan import Cast
    cs = fixtures.uk_cities_schema()
    rel = ex.ProjectRelation(ex.CsvDataSource(os.path.join(DATA, "uk_cities.csv"), cs, 1024),
                             [ex.compile_scalar_expr(None, Cast(Column(1), DataType.Int32), cs)], None)
    vals = [v for b in rel for v in b.column(0).to_pylist()]
    assert vals[:4] == [53, 52, 51, 50] and len(vals) == 36


def test_csv_widest_schema(tmp_path):
    """32 columns (the source's limit): the cell-span scratch of k_csv_parse is 64 KB of LDS per workgroup."""
    rng = np.random.default_rng(3)
    n, nc = 3000, 32
    cols = [rng.integers(-1000, 1000, n) for _ in range(nc)]
    p = str(tmp_path / "wide.csv")
    with open(p, "w") as fh:
        fh.write(",".join(f"c{i}" for i in range(nc)) + "\n")
        for r in range(n):
            fh.write(",".join(("" if (r + i) % 41 == 0 else str(int(cols[i][r]))) if i % 3 else repr(float(cols[i][r]) / 8) for i in range(nc)) + "\n")
    schema = pa.schema([(f"c{i}", pa.int32() if i % 3 else pa.float64()) for i in range(nc)])
    assert check_file(p, schema, 1024) == n
    with pytest.raises(ex.ExecutionError) as ei:
        ex.CsvDataSource(p, pa.schema([(f"c{i}", pa.int32()) for i in range(33)]), 1024)
    assert ei.value.kind == "NotImplemented"

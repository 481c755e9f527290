"""CPU tests of the CSV path's checker and host-side pieces (no GPU):
  * the oracle's CsvDataSource restatement against an independent reader (pyarrow.csv) on every reference fixture,
    and the header quirk the reference's tests rely on (datasource.rs:41: has_headers = true, always);
  * the number parser shared by host and device (csrc/dfx_numparse.hpp, compiled here with g++) against glibc
    strtod / strtof -- both correctly rounded, like Rust's dec2flt that arrow's csv reader calls."""
import os
import subprocess
import sys

import pyarrow as pa
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fixtures  # noqa: E402
import oracle  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = fixtures.DATA

ALL_TYPES = pa.schema([("c_bool", pa.bool_()), ("c_uint8", pa.uint8()), ("c_uint16", pa.uint16()), ("c_uint32", pa.uint32()),
                       ("c_uint64", pa.uint64()), ("c_int8", pa.int8()), ("c_int16", pa.int16()), ("c_int32", pa.int32()),
                       ("c_int64", pa.int64()), ("c_float32", pa.float32()), ("c_float64", pa.float64()),
                       ("c_utf8", pa.string())])
NULL_TEST = pa.schema([("c_int", pa.int32()), ("c_float", pa.float64()), ("c_string", pa.string()), ("c_bool", pa.bool_())])
NUMERICS = pa.schema([("a", pa.int64()), ("b", pa.int64()), ("a_f", pa.float64()), ("b_f", pa.float64())])
PEOPLE = pa.schema([("id", pa.int32()), ("first_name", pa.string())])
CASES = [("uk_cities.csv", fixtures.uk_cities_schema()), ("aggregate_test_1.csv", fixtures.aggr_test_schema()),
         ("aggregate_test_2.csv", pa.schema([("a", pa.string()), ("b", pa.float64())])), ("people.csv", PEOPLE),
         ("numerics.csv", NUMERICS), ("null_test.csv", NULL_TEST), ("all_types_flat.csv", ALL_TYPES)]


@pytest.mark.parametrize("name,schema", CASES)
def test_oracle_csv_matches_independent_reader(name, schema):
    got = pa.Table.from_batches(oracle.read_csv(os.path.join(DATA, name), schema, 1024), schema=None)
    want = pa.Table.from_batches(fixtures.load_csv(name, schema))
    assert got.num_rows == want.num_rows
    for c in range(len(schema)):
        g, w = got.column(c).combine_chunks(), want.column(c).combine_chunks()
        if pa.types.is_string(schema[c].type):  # arrow 0.12: Utf8 cells are never null ("" instead)
            w = w.fill_null("")
        assert g.to_pylist() == w.to_pylist(), (name, schema.names[c])


def test_oracle_csv_header_quirk_and_batching():
    # uk_cities.csv has no header line, yet the reference's reader is created with has_headers = true:
    # 37 lines -> 36 rows, "Elgin" is lost (tests/sql.rs:29-37 only ever sees 36 cities)
    batches = oracle.read_csv(os.path.join(DATA, "uk_cities.csv"), fixtures.uk_cities_schema(), 10)
    assert [b.num_rows for b in batches] == [10, 10, 10, 6]
    assert batches[0].column(0)[0].as_py() == "Stoke-on-Trent, Staffordshire, the UK"
    assert batches[0].column(1)[0].as_py() == 53.002666
    assert oracle.read_csv(os.path.join(DATA, "people.csv"), PEOPLE, 1024)[0].column(1).to_pylist()[:2] == ["Andy", "Brian"]


def test_oracle_csv_quoting_rules(tmp_path):
    p = tmp_path / "q.csv"
    p.write_bytes(b'h1,h2,h3\r\n'
                  b'"a,1","say ""hi""",x\r\n'
                  b'\r\n\n'                      # empty lines are skipped
                  b'un"quoted,"closed"tail,"multi\nline"\n'
                  b',,\n'
                  b'last,"no newline",z')
    s = pa.schema([("a", pa.string()), ("b", pa.string()), ("c", pa.string())])
    rows = pa.Table.from_batches(oracle.read_csv(str(p), s)).to_pylist()
    assert rows == [{"a": "a,1", "b": 'say "hi"', "c": "x"}, {"a": 'un"quoted', "b": "closedtail", "c": "multi\nline"},
                    {"a": "", "b": "", "c": ""}, {"a": "last", "b": "no newline", "c": "z"}]


def test_oracle_csv_errors(tmp_path):
    p = tmp_path / "bad.csv"
    p.write_text("a,b\n1,2.5\n2,abc\n")
    with pytest.raises(oracle.OracleError) as ei:
        oracle.read_csv(str(p), pa.schema([("a", pa.int32()), ("b", pa.float64())]))
    assert "Error while parsing value abc at line 2" in str(ei.value)
    p.write_text("a,b\n1,2\n3\n")
    with pytest.raises(oracle.OracleError) as ei:
        oracle.read_csv(str(p), pa.schema([("a", pa.int32()), ("b", pa.int32())]))
    assert "UnequalLengths" in str(ei.value)
    with pytest.raises(oracle.OracleError):
        oracle.read_csv(str(tmp_path / "missing.csv"), PEOPLE)


def test_number_parser_agrees_with_strtod(tmp_path):
    exe = str(tmp_path / "numparse_fuzz")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "numparse_fuzz.cpp")])
    out = subprocess.run([exe, "400000", "11"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("ok:")


def test_device_csv_automaton_agrees_with_oracle_on_cpu(tmp_path):
    """csrc/dfx_csv_walk.hpp -- the automaton the device kernels run, host build -- against the oracle's reader on random
    text, and the parallel (transition-vector) boundary detection against the sequential one."""
    exe = str(tmp_path / "csv_walk_fuzz")
    obj = str(tmp_path / "oracle.o")
    subprocess.check_call(["gcc", "-O2", "-c", os.path.join(ROOT, "oracle", "dfx_oracle.c"), "-o", obj])
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "csv_walk_fuzz.cpp"), obj, "-lm"])
    out = subprocess.run([exe, "1500", "3", str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("ok:")

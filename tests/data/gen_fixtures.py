#!/usr/bin/env python3
"""Generates the CSV fixtures that are NOT the reference's own test inputs (those -- uk_cities.csv, aggregate_test_1/2.csv,
people.csv -- are the files its tests read, copied as data).  Deterministic; the outputs are committed next to it.

  all_types_gen.csv   12 columns, one of every type the CSV source converts, UTF-8 strings incl. non-ASCII and quoting
  null_gen.csv        empty cells -> nulls (primitive) / "" (Utf8), quoted empties, no trailing newline
  numerics_gen.csv    integers written as floats' neighbours, exponents, signs
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(20240924)

words = ["alpha", "beta, gamma", 'say "hi"', "Zürich", "日本語", "ñandú", "", "x" * 37, "tab\tseparated", "semi;colon"]


def q(s):
    return '"' + s.replace('"', '""') + '"' if (s == "" or any(c in s for c in ',"\n')) and s != "" or s in ("beta, gamma", 'say "hi"') else s


with open(os.path.join(HERE, "all_types_gen.csv"), "w", encoding="utf-8", newline="") as f:
    f.write("c_bool,c_uint8,c_uint16,c_uint32,c_uint64,c_int8,c_int16,c_int32,c_int64,c_float32,c_float64,c_utf8\n")
    for r in range(400):
        row = ["true" if rng.random() < 0.5 else "false", str(int(rng.integers(0, 256))), str(int(rng.integers(0, 65536))),
               str(int(rng.integers(0, 2**32))), str(int(rng.integers(0, 2**63)) * 2 + int(rng.integers(0, 2))),
               str(int(rng.integers(-128, 128))), str(int(rng.integers(-32768, 32768))), str(int(rng.integers(-2**31, 2**31))),
               str(int(rng.integers(-2**63, 2**63))), repr(float(np.float32(rng.random()))), repr(float(rng.standard_normal() * 10.0 ** int(rng.integers(-8, 9)))),
               q(words[int(rng.integers(0, len(words)))] + ("" if r % 5 else str(r)))]
        f.write(",".join(row) + "\n")

with open(os.path.join(HERE, "null_gen.csv"), "w", newline="") as f:
    f.write("c_int,c_float,c_string,c_bool\n")
    lines = []
    for r in range(60):
        lines.append(",".join(["" if r % 7 == 3 else str(r - 30), "" if r % 5 == 2 else repr(r / 8.0), "" if r % 4 == 1 else ('""' if r % 4 == 3 else f'"s{r}"'),
                               "" if r % 9 == 4 else ("true" if r % 2 else "false")]))
    f.write("\n".join(lines))  # no trailing newline

with open(os.path.join(HERE, "numerics_gen.csv"), "w", newline="") as f:
    f.write("a,b,a_f,b_f\n")
    for r in range(120):
        a, b = int(rng.integers(-10**12, 10**12)), int(rng.integers(0, 100))
        forms = [repr(a / 7.0), "%.3e" % (a / 7.0), "%d.0" % b, "+%d" % b if False else str(float(b)), "1e%d" % (r % 40 - 20), "-0.0", ".5", "5."]
        f.write(f"{a},{b},{forms[r % len(forms)]},{forms[(r + 3) % len(forms)]}\n")
print("ok")

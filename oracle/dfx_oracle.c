/*
 * dfx_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See dfx_oracle.h.
 *
 * Single-threaded, batch-at-a-time, same evaluation order as the reference:
 *   - every expression node materialises a full array, literals included
 *     (src/execution/expression.rs:226-243);
 *   - ungrouped SUM = per-batch `T::default() + v0 + v1 ...` (arrow 0.12 array_ops::sum), then the
 *     batch sums are folded in batch order (src/execution/aggregate.rs:703-743, :245-283);
 *   - grouped aggregates fold row by row in arrival order, the first value initialises the
 *     accumulator (aggregate.rs:805-874, :548-612, :107-145/:176-214/:245-283).
 *
 * Third-party arithmetic: crate `arrow = "0.12.0"` (reference Cargo.toml:28, source not vendored
 * under the reference tree).  Its array_ops semantics are restated from the published 0.12.0
 * source; each such function says "arrow 0.12" and whether a reference test pins it.
 *
 * Documented deviations (reference behaviour is a panic / unimplemented!() / obvious bug):
 *   D1 numeric->numeric CAST of any column/literal/expression is implemented with Rust `as`
 *      semantics (float->int saturating, NaN->0); the reference only implements column->Int16/Int32
 *      (expression.rs:272-280) and literal Int64->Float64 (:345-368) and panics / errs otherwise.
 *   D2 fn filter compacts every fixed-width type; the reference errs for anything but Float64 and
 *      Utf8 (filter.rs:105-108).
 *   D3 COUNT is implemented (number of rows whose argument is valid, UInt64); the reference
 *      rejects it at run time (aggregate.rs:331-333, :723-727).  Parity unpinned.
 *   D4 the Int8 aggregate output bug (aggregate.rs:934 uses the *group* macro) is not replicated.
 *   D7 AVG is implemented as SUM(x) / COUNT(x) of the same argument, result in the argument's type (the planner's
 *      contract, sqlplanner.rs:309-322): IEEE division for floats, truncating division of the wrapped integer sum,
 *      NULL when nothing was counted.  The reference's compile_expr rejects "avg" (expression.rs:98-107) although
 *      its planner accepts it; north_star names AVG.  Parity unpinned.
 *   D5 group output order is first-appearance order; the reference's is FnvHashMap iteration order
 *      (unspecified; its own tests flag it, tests/sql.rs:47,:62).  Compare as a set.
 */
#include "dfx_oracle.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------ */
/* helpers                                                                                    */
/* ------------------------------------------------------------------------------------------ */
static int32_t fail(char* err, size_t errlen, int32_t code, const char* fmt, ...) {
  if (err && errlen) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err, errlen, fmt, ap);
    va_end(ap);
  }
  return code;
}

static const char* dt_name(int dt) {
  switch (dt) {
    case DFX_BOOLEAN: return "Boolean";
    case DFX_INT8: return "Int8";
    case DFX_INT16: return "Int16";
    case DFX_INT32: return "Int32";
    case DFX_INT64: return "Int64";
    case DFX_UINT8: return "UInt8";
    case DFX_UINT16: return "UInt16";
    case DFX_UINT32: return "UInt32";
    case DFX_UINT64: return "UInt64";
    case DFX_FLOAT32: return "Float32";
    case DFX_FLOAT64: return "Float64";
    case DFX_UTF8: return "Utf8";
    default: return "Null";
  }
}

static size_t dt_size(int dt) {
  switch (dt) {
    case DFX_INT8: case DFX_UINT8: return 1;
    case DFX_INT16: case DFX_UINT16: return 2;
    case DFX_INT32: case DFX_UINT32: case DFX_FLOAT32: return 4;
    case DFX_INT64: case DFX_UINT64: case DFX_FLOAT64: return 8;
    default: return 0;
  }
}
static int dt_is_numeric(int dt) { return dt >= DFX_INT8 && dt <= DFX_FLOAT64; }
static int dt_is_int(int dt) { return dt >= DFX_INT8 && dt <= DFX_UINT64; }

static inline int bit_get(const uint8_t* b, int64_t i) { return (b[i >> 3] >> (i & 7)) & 1; }
static inline void bit_set(uint8_t* b, int64_t i) { b[i >> 3] |= (uint8_t)(1u << (i & 7)); }
static inline int is_valid(const orc_array* a, int64_t i) { return !a->validity || bit_get(a->validity, i); }
static size_t bitmap_bytes(int64_t n) { return (size_t)((n + 63) / 64) * 8; }

static orc_array* arr_new(int dt, int64_t n, int with_validity) {
  orc_array* a = (orc_array*)calloc(1, sizeof(orc_array));
  a->dtype = dt;
  a->owned = 1;
  a->length = n;
  if (dt == DFX_BOOLEAN) {
    a->values = calloc(1, bitmap_bytes(n) + 8);
  } else if (dt == DFX_UTF8) {
    a->offsets = (int32_t*)calloc((size_t)n + 1, sizeof(int32_t));
  } else {
    a->values = calloc((size_t)(n > 0 ? n : 1), dt_size(dt));
  }
  if (with_validity) a->validity = (uint8_t*)calloc(1, bitmap_bytes(n) + 8);
  return a;
}

void orc_array_free(orc_array* a) {
  if (!a) return;
  if (a->owned) {
    free(a->values);
    free(a->validity);
    free(a->offsets);
    free(a->data);
  }
  free(a);
}

void orc_batch_free(orc_batch* b) {
  if (!b) return;
  for (int i = 0; i < b->num_columns; ++i) orc_array_free(b->columns[i]);
  free(b->columns);
  free(b);
}

static orc_array* arr_view(const orc_array* src) { /* Expr::Column: Arc clone, zero copy (expression.rs:311-315) */
  orc_array* a = (orc_array*)calloc(1, sizeof(orc_array));
  *a = *src;
  a->owned = 0;
  return a;
}

/* ------------------------------------------------------------------------------------------ */
/* typed access: everything widened to one of i64 / u64 / f32 / f64 for the scalar ops         */
/* ------------------------------------------------------------------------------------------ */
typedef union { int64_t i; uint64_t u; double d; float f; } val_t;

static inline val_t load_val(const orc_array* a, int64_t r) {
  val_t v;
  v.u = 0;
  switch (a->dtype) {
    case DFX_INT8: v.i = ((const int8_t*)a->values)[r]; break;
    case DFX_INT16: v.i = ((const int16_t*)a->values)[r]; break;
    case DFX_INT32: v.i = ((const int32_t*)a->values)[r]; break;
    case DFX_INT64: v.i = ((const int64_t*)a->values)[r]; break;
    case DFX_UINT8: v.u = ((const uint8_t*)a->values)[r]; break;
    case DFX_UINT16: v.u = ((const uint16_t*)a->values)[r]; break;
    case DFX_UINT32: v.u = ((const uint32_t*)a->values)[r]; break;
    case DFX_UINT64: v.u = ((const uint64_t*)a->values)[r]; break;
    case DFX_FLOAT32: v.f = ((const float*)a->values)[r]; break;
    case DFX_FLOAT64: v.d = ((const double*)a->values)[r]; break;
    case DFX_BOOLEAN: v.u = (uint64_t)bit_get((const uint8_t*)a->values, r); break;
    default: break;
  }
  return v;
}

static inline void store_val(orc_array* a, int64_t r, val_t v) {
  switch (a->dtype) {
    case DFX_INT8: ((int8_t*)a->values)[r] = (int8_t)v.i; break;
    case DFX_INT16: ((int16_t*)a->values)[r] = (int16_t)v.i; break;
    case DFX_INT32: ((int32_t*)a->values)[r] = (int32_t)v.i; break;
    case DFX_INT64: ((int64_t*)a->values)[r] = v.i; break;
    case DFX_UINT8: ((uint8_t*)a->values)[r] = (uint8_t)v.u; break;
    case DFX_UINT16: ((uint16_t*)a->values)[r] = (uint16_t)v.u; break;
    case DFX_UINT32: ((uint32_t*)a->values)[r] = (uint32_t)v.u; break;
    case DFX_UINT64: ((uint64_t*)a->values)[r] = v.u; break;
    case DFX_FLOAT32: ((float*)a->values)[r] = v.f; break;
    case DFX_FLOAT64: ((double*)a->values)[r] = v.d; break;
    case DFX_BOOLEAN: if (v.u) bit_set((uint8_t*)a->values, r); break;
    default: break;
  }
}

/* wrap an int64 computed value to the width of an integer dtype (Rust release-mode wrapping) */
static inline val_t wrap_int(int dt, val_t v) {
  switch (dt) {
    case DFX_INT8: v.i = (int8_t)v.u; break;
    case DFX_INT16: v.i = (int16_t)v.u; break;
    case DFX_INT32: v.i = (int32_t)v.u; break;
    case DFX_UINT8: v.u = (uint8_t)v.u; break;
    case DFX_UINT16: v.u = (uint16_t)v.u; break;
    case DFX_UINT32: v.u = (uint32_t)v.u; break;
    default: break;
  }
  return v;
}
static int dt_is_signed_int(int dt) { return dt >= DFX_INT8 && dt <= DFX_INT64; }

/* ------------------------------------------------------------------------------------------ */
/* casts: Rust `as` (deviation D1 for the combinations the reference does not implement)       */
/* ------------------------------------------------------------------------------------------ */
static int64_t sat_f64_to_i64(double x, int64_t lo, int64_t hi) {
  if (x != x) return 0;
  if (x <= (double)lo) return lo;
  if (x >= (double)hi) return hi; /* (double)INT64_MAX == 2^63: x >= 2^63 saturates */
  return (int64_t)x;
}
static uint64_t sat_f64_to_u64(double x, uint64_t hi) {
  if (x != x) return 0;
  if (x <= 0.0) return 0;
  if (x >= (double)hi) return hi;
  return (uint64_t)x;
}

static val_t cast_val(int from, int to, val_t v) {
  val_t o;
  o.u = 0;
  if (from == to) return v;
  if (dt_is_int(from)) {
    if (dt_is_int(to)) { /* truncate / extend: two's complement */
      o.u = v.u;
      return wrap_int(to, o);
    }
    if (to == DFX_FLOAT64) o.d = dt_is_signed_int(from) ? (double)v.i : (double)v.u;
    else o.f = dt_is_signed_int(from) ? (float)v.i : (float)v.u;
    return o;
  }
  double x = (from == DFX_FLOAT32) ? (double)v.f : v.d;
  switch (to) {
    case DFX_FLOAT32: o.f = (from == DFX_FLOAT32) ? v.f : (float)v.d; break;
    case DFX_FLOAT64: o.d = x; break;
    case DFX_INT8: o.i = sat_f64_to_i64(x, INT8_MIN, INT8_MAX); break;
    case DFX_INT16: o.i = sat_f64_to_i64(x, INT16_MIN, INT16_MAX); break;
    case DFX_INT32: o.i = sat_f64_to_i64(x, INT32_MIN, INT32_MAX); break;
    case DFX_INT64: o.i = sat_f64_to_i64(x, INT64_MIN, INT64_MAX); break;
    case DFX_UINT8: o.u = sat_f64_to_u64(x, UINT8_MAX); break;
    case DFX_UINT16: o.u = sat_f64_to_u64(x, UINT16_MAX); break;
    case DFX_UINT32: o.u = sat_f64_to_u64(x, UINT32_MAX); break;
    case DFX_UINT64: o.u = sat_f64_to_u64(x, UINT64_MAX); break;
    default: break;
  }
  return o;
}

/* cast_column! (expression.rs:246-270): null-preserving element-wise `as` */
static orc_array* cast_array(const orc_array* src, int to) {
  int64_t n = src->length;
  orc_array* out = arr_new(to, n, src->validity != NULL);
  for (int64_t i = 0; i < n; ++i) {
    if (!is_valid(src, i)) continue; /* append_null: value slot stays 0 */
    if (out->validity) bit_set(out->validity, i);
    store_val(out, i, cast_val(src->dtype, to, load_val(src, i)));
  }
  return out;
}

/* ------------------------------------------------------------------------------------------ */
/* binary kernels (arrow 0.12 array_ops)                                                       */
/* ------------------------------------------------------------------------------------------ */
/* compare two *valid* values of type dt: returns -1/0/1, or 2 for unordered (NaN) */
static inline int cmp_vals(int dt, val_t a, val_t b) {
  if (dt == DFX_FLOAT64) return a.d < b.d ? -1 : (a.d > b.d ? 1 : (a.d == b.d ? 0 : 2));
  if (dt == DFX_FLOAT32) return a.f < b.f ? -1 : (a.f > b.f ? 1 : (a.f == b.f ? 0 : 2));
  if (dt_is_signed_int(dt)) return a.i < b.i ? -1 : (a.i > b.i ? 1 : 0);
  return a.u < b.u ? -1 : (a.u > b.u ? 1 : 0);
}

/* arrow 0.12 array_ops::{eq,neq,lt,lt_eq,gt,gt_eq} (bool_op over Option<T>): the result is never
 * null; for lt/lt_eq a null sorts below every value, for gt/gt_eq the mirror; eq/neq compare the
 * Options.  gt/lt/and on non-null f64 pinned by tests/sql.rs:29-37; null rules unpinned. */
static orc_array* compare_arrays(int op, const orc_array* l, const orc_array* r) {
  int64_t n = l->length;
  orc_array* out = arr_new(DFX_BOOLEAN, n, 0);
  uint8_t* bits = (uint8_t*)out->values;
  int dt = l->dtype;
  for (int64_t i = 0; i < n; ++i) {
    int lv = is_valid(l, i), rv = is_valid(r, i);
    int res;
    if (lv && rv) {
      int c = cmp_vals(dt, load_val(l, i), load_val(r, i));
      switch (op) {
        case DFX_OP_EQ: res = (c == 0); break;
        case DFX_OP_NOT_EQ: res = (c != 0); break;
        case DFX_OP_LT: res = (c == -1); break;
        case DFX_OP_LT_EQ: res = (c == -1 || c == 0); break;
        case DFX_OP_GT: res = (c == 1); break;
        default: res = (c == 1 || c == 0); break; /* GT_EQ */
      }
    } else {
      switch (op) {
        case DFX_OP_EQ: res = (!lv && !rv); break;
        case DFX_OP_NOT_EQ: res = (lv != rv); break;
        case DFX_OP_LT: res = (!lv && rv); break;
        case DFX_OP_LT_EQ: res = !lv; break;
        case DFX_OP_GT: res = (lv && !rv); break;
        default: res = !rv; break; /* GT_EQ: (None,None)=>true, (None,_)=>false, (_,None)=>true */
      }
    }
    if (res) bit_set(bits, i);
  }
  return out;
}

/* arrow 0.12 array_ops::{and,or}: null in => null out, else l && r / l || r.  `and` on non-null
 * input pinned by tests/sql.rs:29-37. */
static orc_array* boolean_arrays(int op, const orc_array* l, const orc_array* r) {
  int64_t n = l->length;
  int nullable = l->validity || r->validity;
  orc_array* out = arr_new(DFX_BOOLEAN, n, nullable);
  uint8_t* bits = (uint8_t*)out->values;
  for (int64_t i = 0; i < n; ++i) {
    if (!is_valid(l, i) || !is_valid(r, i)) continue; /* append_null */
    if (out->validity) bit_set(out->validity, i);
    int a = bit_get((const uint8_t*)l->values, i), b = bit_get((const uint8_t*)r->values, i);
    if (op == DFX_OP_AND ? (a && b) : (a || b)) bit_set(bits, i);
  }
  return out;
}

/* arrow 0.12 array_ops::{add,subtract,multiply,divide} (math_op): null in => null out; divide
 * returns Err(DivideByZero) on a zero divisor of any type; integers wrap (release build).  f64 add
 * pinned by tests/sql.rs:29-37; the rest unpinned. */
static int32_t math_arrays(int op, const orc_array* l, const orc_array* r, orc_array** outp,
                           char* err, size_t errlen) {
  int64_t n = l->length;
  int dt = l->dtype;
  int nullable = l->validity || r->validity;
  orc_array* out = arr_new(dt, n, nullable);
  for (int64_t i = 0; i < n; ++i) {
    if (!is_valid(l, i) || !is_valid(r, i)) continue;
    if (out->validity) bit_set(out->validity, i);
    val_t a = load_val(l, i), b = load_val(r, i), o;
    o.u = 0;
    if (dt == DFX_FLOAT64) {
      switch (op) {
        case DFX_OP_PLUS: o.d = a.d + b.d; break;
        case DFX_OP_MINUS: o.d = a.d - b.d; break;
        case DFX_OP_MULTIPLY: o.d = a.d * b.d; break;
        default:
          if (b.d == 0.0) { orc_array_free(out); return fail(err, errlen, DFX_ARROW_ERROR, "DivideByZero"); }
          o.d = a.d / b.d;
      }
    } else if (dt == DFX_FLOAT32) {
      switch (op) {
        case DFX_OP_PLUS: o.f = a.f + b.f; break;
        case DFX_OP_MINUS: o.f = a.f - b.f; break;
        case DFX_OP_MULTIPLY: o.f = a.f * b.f; break;
        default:
          if (b.f == 0.0f) { orc_array_free(out); return fail(err, errlen, DFX_ARROW_ERROR, "DivideByZero"); }
          o.f = a.f / b.f;
      }
    } else {
      switch (op) {
        case DFX_OP_PLUS: o.u = a.u + b.u; break;
        case DFX_OP_MINUS: o.u = a.u - b.u; break;
        case DFX_OP_MULTIPLY: o.u = a.u * b.u; break;
        default:
          if (b.u == 0) { orc_array_free(out); return fail(err, errlen, DFX_ARROW_ERROR, "DivideByZero"); }
          if (dt_is_signed_int(dt)) {
            /* Rust panics on MIN / -1 even in release ("attempt to divide with overflow") */
            val_t mn; mn.u = 0;
            switch (dt) { case DFX_INT8: mn.i = INT8_MIN; break; case DFX_INT16: mn.i = INT16_MIN; break;
                          case DFX_INT32: mn.i = INT32_MIN; break; default: mn.i = INT64_MIN; }
            if (b.i == -1 && a.i == mn.i) {
              orc_array_free(out);
              return fail(err, errlen, DFX_INTERNAL_ERROR, "attempt to divide with overflow");
            }
            o.i = a.i / b.i;
          } else {
            o.u = a.u / b.u;
          }
      }
      o = wrap_int(dt, o);
    }
    store_val(out, i, o);
  }
  *outp = out;
  return DFX_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* compile_scalar_expr + evaluation (expression.rs:283-505)                                    */
/* ------------------------------------------------------------------------------------------ */
static orc_array* literal_array(int dt, val_t v, int64_t n) { /* literal_array! (expression.rs:226-243) */
  orc_array* a = arr_new(dt, n, 0);
  for (int64_t i = 0; i < n; ++i) store_val(a, i, v);
  return a;
}

static val_t node_literal(const dfx_expr_node* nd) {
  val_t v;
  v.u = 0;
  if (nd->dtype == DFX_FLOAT64) v.d = nd->lit.f64;
  else if (nd->dtype == DFX_FLOAT32) v.f = nd->lit.f32;
  else if (dt_is_signed_int(nd->dtype)) v.i = nd->lit.i64;
  else v.u = nd->lit.u64;
  return v;
}

static int32_t eval_node(const dfx_expr_node* nodes, int32_t n_nodes, int32_t idx,
                         const orc_batch* batch, orc_array** out, char* err, size_t errlen) {
  if (idx < 0 || idx >= n_nodes) return fail(err, errlen, DFX_INTERNAL_ERROR, "expression node index %d out of range", idx);
  const dfx_expr_node* nd = &nodes[idx];
  switch (nd->kind) {
    case DFX_EXPR_LITERAL: {
      if (!dt_is_numeric(nd->dtype))
        return fail(err, errlen, DFX_EXECUTION_ERROR, "No support for literal type %s", dt_name(nd->dtype));
      *out = literal_array(nd->dtype, node_literal(nd), batch->num_rows);
      return DFX_OK;
    }
    case DFX_EXPR_COLUMN: {
      if (nd->column < 0 || nd->column >= batch->num_columns)
        return fail(err, errlen, DFX_INTERNAL_ERROR, "column index %d out of bounds", nd->column);
      *out = arr_view(batch->columns[nd->column]);
      return DFX_OK;
    }
    case DFX_EXPR_CAST: {
      if (!dt_is_numeric(nd->dtype))
        return fail(err, errlen, DFX_NOT_IMPLEMENTED, "CAST to %s", dt_name(nd->dtype));
      orc_array* child = NULL;
      int32_t st = eval_node(nodes, n_nodes, nd->left, batch, &child, err, errlen);
      if (st) return st;
      if (!dt_is_numeric(child->dtype)) {
        int cdt = child->dtype;
        orc_array_free(child);
        return fail(err, errlen, DFX_INTERNAL_ERROR, "unsupported CAST operation from %s", dt_name(cdt));
      }
      *out = cast_array(child, nd->dtype);
      orc_array_free(child);
      return DFX_OK;
    }
    case DFX_EXPR_BINARY: {
      int op = nd->op;
      if (op == DFX_OP_MODULUS || op == DFX_OP_NOT || op == DFX_OP_LIKE || op == DFX_OP_NOT_LIKE || op < 0 || op > DFX_OP_NOT_LIKE)
        return fail(err, errlen, DFX_EXECUTION_ERROR, "operator: %d", op);
      orc_array *l = NULL, *r = NULL;
      int32_t st = eval_node(nodes, n_nodes, nd->left, batch, &l, err, errlen);
      if (st) return st;
      st = eval_node(nodes, n_nodes, nd->right, batch, &r, err, errlen);
      if (st) { orc_array_free(l); return st; }
      if (op == DFX_OP_AND || op == DFX_OP_OR) {
        if (l->dtype != DFX_BOOLEAN || r->dtype != DFX_BOOLEAN) { /* downcast_ref::<BooleanArray>().unwrap() */
          orc_array_free(l); orc_array_free(r);
          return fail(err, errlen, DFX_INTERNAL_ERROR, "boolean_ops: operand is not a BooleanArray");
        }
        *out = boolean_arrays(op, l, r);
      } else if (op <= DFX_OP_GT_EQ) {
        if (l->dtype != r->dtype || !dt_is_numeric(l->dtype)) {
          orc_array_free(l); orc_array_free(r);
          return fail(err, errlen, DFX_EXECUTION_ERROR, "comparison_ops");
        }
        *out = compare_arrays(op, l, r);
      } else {
        if (l->dtype != r->dtype || !dt_is_numeric(l->dtype)) {
          orc_array_free(l); orc_array_free(r);
          return fail(err, errlen, DFX_EXECUTION_ERROR, "math_ops");
        }
        st = math_arrays(op, l, r, out, err, errlen);
        if (st) { orc_array_free(l); orc_array_free(r); return st; }
      }
      orc_array_free(l);
      orc_array_free(r);
      return DFX_OK;
    }
    default:
      return fail(err, errlen, DFX_EXECUTION_ERROR, "expression kind %d", nd->kind);
  }
}

int32_t orc_eval(const dfx_expr_node* nodes, int32_t n_nodes, int32_t root, const orc_batch* batch,
                 orc_array** out, char* err, size_t errlen) {
  orc_array* a = NULL;
  int32_t st = eval_node(nodes, n_nodes, root, batch, &a, err, errlen);
  if (st) return st;
  if (!a->owned) { /* hand the caller an owned copy so orc_array_free is uniform */
    orc_array* c;
    if (a->dtype == DFX_UTF8) {
      c = arr_new(DFX_UTF8, a->length, a->validity != NULL);
      memcpy(c->offsets, a->offsets, sizeof(int32_t) * (size_t)(a->length + 1));
      int32_t nb = a->offsets[a->length];
      c->data = (uint8_t*)malloc((size_t)nb + 1);
      memcpy(c->data, a->data, (size_t)nb);
    } else if (a->dtype == DFX_BOOLEAN) {
      c = arr_new(DFX_BOOLEAN, a->length, a->validity != NULL);
      memcpy(c->values, a->values, (size_t)((a->length + 7) / 8));
    } else {
      c = arr_new(a->dtype, a->length, a->validity != NULL);
      memcpy(c->values, a->values, dt_size(a->dtype) * (size_t)a->length);
    }
    if (a->validity) memcpy(c->validity, a->validity, (size_t)((a->length + 7) / 8));
    orc_array_free(a);
    a = c;
  }
  *out = a;
  return DFX_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* fn filter (filter.rs:79-110) and FilterRelation::next (filter.rs:46-71)                     */
/* ------------------------------------------------------------------------------------------ */
static orc_array* filter_array(const orc_array* a, const orc_array* mask) {
  const uint8_t* m = (const uint8_t*)mask->values; /* filter.value(i): raw value bit, mask nulls ignored */
  int64_t n = a->length, kept = 0;
  for (int64_t i = 0; i < n; ++i) kept += bit_get(m, i);
  if (a->dtype == DFX_UTF8) { /* filter.rs:93-104 */
    orc_array* out = arr_new(DFX_UTF8, kept, 0);
    int64_t bytes = 0;
    for (int64_t i = 0; i < n; ++i)
      if (bit_get(m, i)) bytes += a->offsets[i + 1] - a->offsets[i];
    out->data = (uint8_t*)malloc((size_t)bytes + 1);
    int64_t k = 0, pos = 0;
    for (int64_t i = 0; i < n; ++i) {
      if (!bit_get(m, i)) continue;
      int32_t len = a->offsets[i + 1] - a->offsets[i];
      memcpy(out->data + pos, a->data + a->offsets[i], (size_t)len);
      pos += len;
      out->offsets[++k] = (int32_t)pos;
    }
    return out;
  }
  /* filter.rs:83-92: value nulls ignored, output all-valid (deviation D2: every fixed width) */
  orc_array* out = arr_new(a->dtype, kept, 0);
  int64_t k = 0;
  if (a->dtype == DFX_BOOLEAN) {
    for (int64_t i = 0; i < n; ++i)
      if (bit_get(m, i)) { if (bit_get((const uint8_t*)a->values, i)) bit_set((uint8_t*)out->values, k); ++k; }
    return out;
  }
  size_t w = dt_size(a->dtype);
  const uint8_t* src = (const uint8_t*)a->values;
  uint8_t* dst = (uint8_t*)out->values;
  for (int64_t i = 0; i < n; ++i)
    if (bit_get(m, i)) { memcpy(dst + (size_t)k * w, src + (size_t)i * w, w); ++k; }
  return out;
}

int32_t orc_filter_next(const dfx_expr_node* nodes, int32_t n_nodes, int32_t root,
                        const orc_batch* batch, orc_batch** out, char* err, size_t errlen) {
  orc_array* mask = NULL;
  int32_t st = eval_node(nodes, n_nodes, root, batch, &mask, err, errlen);
  if (st) return st;
  if (mask->dtype != DFX_BOOLEAN) {
    orc_array_free(mask);
    return fail(err, errlen, DFX_EXECUTION_ERROR, "Filter expression did not evaluate to boolean");
  }
  /* fn filter has arms for Float64 and Utf8 only (filter.rs:105-108).  Deviation D2 widens that to every fixed-width
   * numeric type; a Boolean column keeps the reference's error, for the batch as a whole and whatever its length */
  for (int c = 0; c < batch->num_columns; ++c)
    if (batch->columns[c]->dtype == DFX_BOOLEAN) {
      orc_array_free(mask);
      return fail(err, errlen, DFX_EXECUTION_ERROR, "filter not supported for Boolean");
    }
  orc_batch* ob = (orc_batch*)calloc(1, sizeof(orc_batch));
  ob->owned = 1;
  const int nc = batch->num_columns > 0 ? batch->num_columns : 0;
  ob->num_columns = nc;
  ob->columns = (orc_array**)calloc((size_t)nc + 1, sizeof(orc_array*));
  for (int c = 0; c < nc; ++c) ob->columns[c] = filter_array(batch->columns[c], mask);
  ob->num_rows = batch->num_columns ? ob->columns[0]->length : 0;
  orc_array_free(mask);
  *out = ob;
  return DFX_OK;
}

/* ProjectRelation::next (projection.rs:46-66) */
int32_t orc_project_next(const dfx_expr_node* nodes, int32_t n_nodes, const int32_t* roots,
                         int32_t n_roots, const orc_batch* batch, orc_batch** out, char* err,
                         size_t errlen) {
  orc_batch* ob = (orc_batch*)calloc(1, sizeof(orc_batch));
  ob->owned = 1;
  ob->num_columns = n_roots;
  ob->num_rows = batch->num_rows;
  ob->columns = (orc_array**)calloc((size_t)n_roots, sizeof(orc_array*));
  for (int i = 0; i < n_roots; ++i) {
    int32_t st = orc_eval(nodes, n_nodes, roots[i], batch, &ob->columns[i], err, errlen);
    if (st) { orc_batch_free(ob); return st; }
  }
  *out = ob;
  return DFX_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* aggregate (aggregate.rs)                                                                    */
/* ------------------------------------------------------------------------------------------ */
enum { AGG_MIN = 0, AGG_MAX = 1, AGG_SUM = 2, AGG_COUNT = 3, AGG_AVG = 4 };

typedef struct { int has; val_t v; } scalar_t; /* Option<ScalarValue> of a known dtype */

/* Min/Max/SumFunction::accumulate_scalar (aggregate.rs:107-145, :176-214, :245-283): the first
 * non-None value initialises; then typed combine; floats use f64::min / f64::max (NaN-ignoring)
 * and `a + b`.  Grouped sums pinned by aggregate.rs:1116,1121,1126 (13.2, 3.0, 3.3000000000000003). */
static void accumulate_scalar(int func, int dt, scalar_t* acc, scalar_t value) {
  if (!acc->has) { *acc = value; return; }
  if (!value.has) return;
  val_t a = acc->v, b = value.v, o;
  o.u = 0;
  if (func == AGG_COUNT) { o.u = a.u + b.u; acc->v = o; return; }
  if (dt == DFX_FLOAT64) {
    o.d = func == AGG_MIN ? fmin(a.d, b.d) : func == AGG_MAX ? fmax(a.d, b.d) : a.d + b.d;
  } else if (dt == DFX_FLOAT32) {
    o.f = func == AGG_MIN ? fminf(a.f, b.f) : func == AGG_MAX ? fmaxf(a.f, b.f) : a.f + b.f;
  } else if (dt_is_signed_int(dt)) {
    if (func == AGG_MIN) o.i = a.i < b.i ? a.i : b.i;
    else if (func == AGG_MAX) o.i = a.i > b.i ? a.i : b.i;
    else { o.u = a.u + b.u; o = wrap_int(dt, o); }
  } else {
    if (func == AGG_MIN) o.u = a.u < b.u ? a.u : b.u;
    else if (func == AGG_MAX) o.u = a.u > b.u ? a.u : b.u;
    else { o.u = a.u + b.u; o = wrap_int(dt, o); }
  }
  acc->v = o;
}

/* array_min / array_max / array_sum (aggregate.rs:344-546) over arrow 0.12 array_ops::{min,max,
 * sum}: nulls skipped, empty/all-null => None; min/max scan with `<` / `>` (a leading NaN sticks);
 * sum starts from T::default() and adds in index order.  min/max over f64 pinned by
 * aggregate.rs:996,1030. */
static scalar_t array_reduce(int func, const orc_array* a) {
  scalar_t s;
  s.has = 0;
  s.v.u = 0;
  int dt = a->dtype;
  int64_t n = a->length;
  if (func == AGG_COUNT) { /* deviation D3 */
    uint64_t c = 0;
    for (int64_t i = 0; i < n; ++i) c += (uint64_t)is_valid(a, i);
    s.has = 1;
    s.v.u = c;
    return s;
  }
  for (int64_t i = 0; i < n; ++i) {
    if (!is_valid(a, i)) continue;
    val_t m = load_val(a, i);
    if (func == AGG_SUM) {
      if (!s.has) { s.has = 1; s.v.u = 0; if (dt == DFX_FLOAT64) s.v.d = 0.0; if (dt == DFX_FLOAT32) s.v.f = 0.0f; }
      if (dt == DFX_FLOAT64) s.v.d = s.v.d + m.d;
      else if (dt == DFX_FLOAT32) s.v.f = s.v.f + m.f;
      else { s.v.u = s.v.u + m.u; s.v = wrap_int(dt, s.v); }
    } else {
      if (!s.has) { s.has = 1; s.v = m; continue; }
      int c = cmp_vals(dt, m, s.v);
      if (func == AGG_MIN ? (c == -1) : (c == 1)) s.v = m;
    }
  }
  return s;
}

typedef struct group_entry {
  uint8_t* key;    /* serialised Vec<GroupByScalar> */
  size_t key_len;
  uint64_t hash;
  scalar_t* acc;   /* one per aggregate */
} group_entry;

struct orc_agg {
  dfx_expr_node* nodes;
  int32_t n_nodes;
  int32_t n_group, n_aggr;
  int32_t* group_roots;
  int32_t* aggr_args;  /* node index of args[0] */
  int32_t* aggr_func;
  int32_t* aggr_type;  /* declared return_type `t` */
  int32_t* group_type; /* seen at first batch */
  int group_type_known;
  scalar_t* acc;       /* ungrouped accumulators */
  /* FnvHashMap<Vec<GroupByScalar>, ..> (aggregate.rs:793): open addressing over entry indices */
  group_entry* entries;
  int64_t n_entries, cap_entries;
  int64_t* slots;
  int64_t n_slots;
};

static int agg_func_from_name(const char* name) {
  if (!name) return -1;
  if (!strcasecmp(name, "min")) return AGG_MIN;
  if (!strcasecmp(name, "max")) return AGG_MAX;
  if (!strcasecmp(name, "sum")) return AGG_SUM;
  if (!strcasecmp(name, "count")) return AGG_COUNT;
  if (!strcasecmp(name, "avg")) return AGG_AVG; /* deviation D7 */
  return -1;
}

int32_t orc_agg_new(const dfx_expr_node* nodes, int32_t n_nodes, const int32_t* group_roots,
                    int32_t n_group, const int32_t* aggr_roots, int32_t n_aggr, orc_agg** out,
                    char* err, size_t errlen) {
  orc_agg* g = (orc_agg*)calloc(1, sizeof(orc_agg));
  g->nodes = (dfx_expr_node*)malloc(sizeof(dfx_expr_node) * (size_t)n_nodes);
  memcpy(g->nodes, nodes, sizeof(dfx_expr_node) * (size_t)n_nodes);
  g->n_nodes = n_nodes;
  g->n_group = n_group;
  g->n_aggr = n_aggr;
  g->group_roots = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_group + 1));
  memcpy(g->group_roots, group_roots, sizeof(int32_t) * (size_t)n_group);
  g->group_type = (int32_t*)calloc((size_t)n_group + 1, sizeof(int32_t));
  g->aggr_args = (int32_t*)calloc((size_t)n_aggr + 1, sizeof(int32_t));
  g->aggr_func = (int32_t*)calloc((size_t)n_aggr + 1, sizeof(int32_t));
  g->aggr_type = (int32_t*)calloc((size_t)n_aggr + 1, sizeof(int32_t));
  g->acc = (scalar_t*)calloc(2 * (size_t)n_aggr + 1, sizeof(scalar_t)); /* [n_aggr + i]: AVG's count */
  for (int i = 0; i < n_aggr; ++i) {
    const dfx_expr_node* nd = &nodes[aggr_roots[i]];
    if (nd->kind != DFX_EXPR_AGGREGATE_FUNCTION) { /* create_accumulators :335-337 */
      orc_agg_free(g);
      return fail(err, errlen, DFX_EXECUTION_ERROR, "invalid aggregate expression");
    }
    if (nd->n_args != 1) { /* assert_eq!(1, args.len()) expression.rs:91 */
      orc_agg_free(g);
      return fail(err, errlen, DFX_INTERNAL_ERROR, "assertion failed: aggregate takes exactly 1 argument");
    }
    int f = agg_func_from_name(nd->name);
    if (f < 0) { /* expression.rs:103-106 */
      orc_agg_free(g);
      return fail(err, errlen, DFX_GENERAL, "Unsupported aggregate function '%s'", nd->name ? nd->name : "");
    }
    g->aggr_func[i] = f;
    g->aggr_args[i] = nd->left;
    g->aggr_type[i] = nd->dtype;
  }
  g->n_slots = 1024;
  g->slots = (int64_t*)malloc(sizeof(int64_t) * (size_t)g->n_slots);
  for (int64_t i = 0; i < g->n_slots; ++i) g->slots[i] = -1;
  *out = g;
  return DFX_OK;
}

void orc_agg_free(orc_agg* g) {
  if (!g) return;
  for (int64_t i = 0; i < g->n_entries; ++i) { free(g->entries[i].key); free(g->entries[i].acc); }
  free(g->entries);
  free(g->slots);
  free(g->nodes);
  free(g->group_roots);
  free(g->group_type);
  free(g->aggr_args);
  free(g->aggr_func);
  free(g->aggr_type);
  free(g->acc);
  free(g);
}

static uint64_t fnv1a(const uint8_t* p, size_t n) { /* fnv 1.0.3 FnvHasher (Cargo.toml:27) */
  uint64_t h = 0xcbf29ce484222325ull;
  for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
  return h;
}

static void map_grow(orc_agg* g) {
  int64_t ns = g->n_slots * 2;
  int64_t* s = (int64_t*)malloc(sizeof(int64_t) * (size_t)ns);
  for (int64_t i = 0; i < ns; ++i) s[i] = -1;
  for (int64_t e = 0; e < g->n_entries; ++e) {
    int64_t p = (int64_t)(g->entries[e].hash & (uint64_t)(ns - 1));
    while (s[p] >= 0) p = (p + 1) & (ns - 1);
    s[p] = e;
  }
  free(g->slots);
  g->slots = s;
  g->n_slots = ns;
}

static group_entry* map_get_or_insert(orc_agg* g, uint8_t* key, size_t key_len, int* inserted) {
  uint64_t h = fnv1a(key, key_len);
  int64_t p = (int64_t)(h & (uint64_t)(g->n_slots - 1));
  while (g->slots[p] >= 0) {
    group_entry* e = &g->entries[g->slots[p]];
    if (e->hash == h && e->key_len == key_len && !memcmp(e->key, key, key_len)) {
      *inserted = 0;
      return e;
    }
    p = (p + 1) & (g->n_slots - 1);
  }
  if (g->n_entries == g->cap_entries) {
    g->cap_entries = g->cap_entries ? g->cap_entries * 2 : 1024;
    g->entries = (group_entry*)realloc(g->entries, sizeof(group_entry) * (size_t)g->cap_entries);
  }
  group_entry* e = &g->entries[g->n_entries];
  e->key = key; /* map.insert(key.clone(), ..): takes the heap key */
  e->key_len = key_len;
  e->hash = h;
  e->acc = (scalar_t*)calloc(2 * (size_t)g->n_aggr + 1, sizeof(scalar_t)); /* create_accumulators; [n_aggr + i]: AVG's count */
  g->slots[p] = g->n_entries++;
  *inserted = 1;
  if (g->n_entries * 2 > g->n_slots) map_grow(g);
  return &g->entries[g->n_entries - 1];
}

static int32_t check_agg_type(const orc_agg* g, int i, const orc_array* arr, char* err, size_t errlen) {
  /* downcast_ref::<$T>().unwrap() by the declared type `t` (aggregate.rs:347.., :561-603) */
  if (g->aggr_func[i] == AGG_COUNT) return DFX_OK;
  if (!dt_is_numeric(g->aggr_type[i]))
    return fail(err, errlen, DFX_EXECUTION_ERROR, "Unsupported data type for aggregate: %s", dt_name(g->aggr_type[i]));
  if (arr->dtype != g->aggr_type[i])
    return fail(err, errlen, DFX_INTERNAL_ERROR, "aggregate argument is %s but declared type is %s",
                dt_name(arr->dtype), dt_name(g->aggr_type[i]));
  return DFX_OK;
}

int32_t orc_agg_push(orc_agg* g, const orc_batch* batch, char* err, size_t errlen) {
  int32_t st;
  if (g->n_group == 0) { /* without_group_by (aggregate.rs:703-743) */
    for (int i = 0; i < g->n_aggr; ++i) {
      orc_array* arr = NULL;
      st = eval_node(g->nodes, g->n_nodes, g->aggr_args[i], batch, &arr, err, errlen);
      if (st) return fail(err, errlen, DFX_EXECUTION_ERROR, "Failed to evaluate argument to aggregate function");
      st = check_agg_type(g, i, arr, err, errlen);
      if (st) { orc_array_free(arr); return st; }
      if (g->aggr_func[i] == AGG_AVG) { /* D7: the SUM and the COUNT of the same argument */
        scalar_t s = array_reduce(AGG_SUM, arr), c = array_reduce(AGG_COUNT, arr);
        accumulate_scalar(AGG_SUM, g->aggr_type[i], &g->acc[i], s);
        accumulate_scalar(AGG_COUNT, DFX_UINT64, &g->acc[g->n_aggr + i], c);
      } else {
        scalar_t s = array_reduce(g->aggr_func[i], arr);
        accumulate_scalar(g->aggr_func[i], g->aggr_type[i], &g->acc[i], s);
      }
      orc_array_free(arr);
    }
    return DFX_OK;
  }
  /* with_group_by (aggregate.rs:787-875) */
  orc_array** keys = (orc_array**)calloc((size_t)g->n_group, sizeof(orc_array*));
  orc_array** args = (orc_array**)calloc((size_t)g->n_aggr + 1, sizeof(orc_array*));
  st = DFX_OK;
  for (int k = 0; k < g->n_group && !st; ++k) {
    st = eval_node(g->nodes, g->n_nodes, g->group_roots[k], batch, &keys[k], err, errlen);
    if (!st) {
      int dt = keys[k]->dtype;
      if (!(dt_is_int(dt) || dt == DFX_UTF8)) /* aggregate.rs:848-850 */
        st = fail(err, errlen, DFX_EXECUTION_ERROR, "Unsupported GROUP BY data type");
      else if (g->group_type_known && g->group_type[k] != dt)
        st = fail(err, errlen, DFX_INTERNAL_ERROR, "group key type changed between batches");
      else g->group_type[k] = dt;
    }
  }
  /* update_accumulators re-evaluates args[0](&batch) per row (aggregate.rs:559); hoisted to once
   * per batch here -- result-identical (pure function of the batch). */
  for (int i = 0; i < g->n_aggr && !st; ++i) {
    st = eval_node(g->nodes, g->n_nodes, g->aggr_args[i], batch, &args[i], err, errlen);
    if (!st) st = check_agg_type(g, i, args[i], err, errlen);
  }
  if (!st) {
    g->group_type_known = 1;
    for (int64_t row = 0; row < batch->num_rows; ++row) {
      /* key: Vec<GroupByScalar>, heap allocated per row (aggregate.rs:807-852); nulls not checked */
      size_t len = 0;
      for (int k = 0; k < g->n_group; ++k)
        len += 1 + (keys[k]->dtype == DFX_UTF8 ? 4 + (size_t)(keys[k]->offsets[row + 1] - keys[k]->offsets[row]) : 8);
      uint8_t* key = (uint8_t*)malloc(len ? len : 1);
      size_t p = 0;
      for (int k = 0; k < g->n_group; ++k) {
        key[p++] = (uint8_t)keys[k]->dtype; /* enum discriminant */
        if (keys[k]->dtype == DFX_UTF8) {
          int32_t sl = keys[k]->offsets[row + 1] - keys[k]->offsets[row];
          memcpy(key + p, &sl, 4); p += 4;
          memcpy(key + p, keys[k]->data + keys[k]->offsets[row], (size_t)sl); p += (size_t)sl;
        } else {
          val_t v = load_val(keys[k], row);
          memcpy(key + p, &v.u, 8); p += 8;
        }
      }
      int inserted = 0;
      group_entry* e = map_get_or_insert(g, key, len, &inserted);
      if (!inserted) free(key);
      for (int j = 0; j < g->n_aggr; ++j) { /* update_accumulators (aggregate.rs:548-612): value(row), no null check */
        scalar_t s;
        s.has = 1;
        if (g->aggr_func[j] == AGG_COUNT) { s.v.u = (uint64_t)is_valid(args[j], row); }
        else s.v = load_val(args[j], row);
        if (g->aggr_func[j] == AGG_AVG) { /* D7 */
          scalar_t c;
          c.has = 1;
          c.v.u = (uint64_t)is_valid(args[j], row);
          accumulate_scalar(AGG_SUM, g->aggr_type[j], &e->acc[j], s);
          accumulate_scalar(AGG_COUNT, DFX_UINT64, &e->acc[g->n_aggr + j], c);
        } else {
          accumulate_scalar(g->aggr_func[j], g->aggr_type[j], &e->acc[j], s);
        }
      }
    }
  }
  for (int k = 0; k < g->n_group; ++k) orc_array_free(keys[k]);
  for (int i = 0; i < g->n_aggr; ++i) orc_array_free(args[i]);
  free(keys);
  free(args);
  return st;
}

static int agg_out_type(const orc_agg* g, int i) { return g->aggr_func[i] == AGG_COUNT ? DFX_UINT64 : g->aggr_type[i]; }

/* D7: AVG = SUM / COUNT in the argument's type: IEEE division for floats, truncating division of the (wrapped)
 * integer sum; None when nothing was counted. */
static scalar_t avg_finish(int dt, scalar_t sum, scalar_t cnt) {
  scalar_t r;
  r.has = 0;
  r.v.u = 0;
  if (!sum.has || !cnt.has || cnt.v.u == 0) return r;
  r.has = 1;
  if (dt == DFX_FLOAT64) r.v.d = sum.v.d / (double)cnt.v.u;
  else if (dt == DFX_FLOAT32) r.v.f = sum.v.f / (float)cnt.v.u;
  else if (dt_is_signed_int(dt)) r.v.i = wrap_int(dt, sum.v).i / (int64_t)cnt.v.u;
  else r.v.u = wrap_int(dt, sum.v).u / cnt.v.u;
  return r;
}

int32_t orc_agg_finish(orc_agg* g, orc_batch** out, char* err, size_t errlen) {
  orc_batch* ob = (orc_batch*)calloc(1, sizeof(orc_batch));
  ob->owned = 1;
  if (g->n_group == 0) { /* aggregate.rs:745-784: one row, null when no input (array_from_scalar!) */
    ob->num_columns = g->n_aggr;
    ob->num_rows = 1;
    ob->columns = (orc_array**)calloc((size_t)g->n_aggr + 1, sizeof(orc_array*));
    for (int i = 0; i < g->n_aggr; ++i) {
      int dt = agg_out_type(g, i);
      if (!dt_is_numeric(dt)) { orc_batch_free(ob); return fail(err, errlen, DFX_NOT_IMPLEMENTED, "tbd"); }
      orc_array* a = arr_new(dt, 1, 1);
      scalar_t r = g->aggr_func[i] == AGG_AVG ? avg_finish(dt, g->acc[i], g->acc[g->n_aggr + i]) : g->acc[i];
      if (r.has) { bit_set(a->validity, 0); store_val(a, 0, r.v); }
      ob->columns[i] = a;
    }
    *out = ob;
    return DFX_OK;
  }
  /* aggregate.rs:877-951 */
  int64_t n = g->n_entries;
  ob->num_columns = g->n_group + g->n_aggr;
  ob->num_rows = n;
  ob->columns = (orc_array**)calloc((size_t)ob->num_columns + 1, sizeof(orc_array*));
  for (int k = 0; k < g->n_group; ++k) {
    int dt = g->group_type_known ? g->group_type[k] : DFX_INT64;
    orc_array* a = arr_new(dt, n, 0);
    if (dt == DFX_UTF8) {
      int64_t bytes = 0;
      for (int64_t e = 0; e < n; ++e) {
        const uint8_t* key = g->entries[e].key; size_t p = 0;
        for (int kk = 0; kk < k; ++kk) { if (key[p] == DFX_UTF8) { int32_t sl; memcpy(&sl, key + p + 1, 4); p += 5 + (size_t)sl; } else p += 9; }
        int32_t sl; memcpy(&sl, key + p + 1, 4); bytes += sl;
      }
      a->data = (uint8_t*)malloc((size_t)bytes + 1);
      int64_t pos = 0;
      for (int64_t e = 0; e < n; ++e) {
        const uint8_t* key = g->entries[e].key; size_t p = 0;
        for (int kk = 0; kk < k; ++kk) { if (key[p] == DFX_UTF8) { int32_t sl; memcpy(&sl, key + p + 1, 4); p += 5 + (size_t)sl; } else p += 9; }
        int32_t sl; memcpy(&sl, key + p + 1, 4);
        memcpy(a->data + pos, key + p + 5, (size_t)sl);
        pos += sl;
        a->offsets[e + 1] = (int32_t)pos;
      }
    } else {
      for (int64_t e = 0; e < n; ++e) {
        const uint8_t* key = g->entries[e].key; size_t p = 0;
        for (int kk = 0; kk < k; ++kk) { if (key[p] == DFX_UTF8) { int32_t sl; memcpy(&sl, key + p + 1, 4); p += 5 + (size_t)sl; } else p += 9; }
        val_t v; memcpy(&v.u, key + p + 1, 8);
        store_val(a, e, v);
      }
    }
    ob->columns[k] = a;
  }
  for (int i = 0; i < g->n_aggr; ++i) {
    int dt = agg_out_type(g, i);
    if (!dt_is_numeric(dt)) { orc_batch_free(ob); return fail(err, errlen, DFX_EXECUTION_ERROR, "Unsupported aggregate expr"); }
    orc_array* a = arr_new(dt, n, 1);
    for (int64_t e = 0; e < n; ++e) {
      scalar_t r = g->aggr_func[i] == AGG_AVG ? avg_finish(dt, g->entries[e].acc[i], g->entries[e].acc[g->n_aggr + i])
                                              : g->entries[e].acc[i];
      if (r.has) { bit_set(a->validity, e); store_val(a, e, r.v); }
    }
    ob->columns[g->n_group + i] = a;
  }
  *out = ob;
  return DFX_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* synthetic generator; the device generator in                       */
/* datafusion_archive_amd/csrc/dfx_kernels.hip implements the same definition.                 */
/* ------------------------------------------------------------------------------------------ */
static inline uint64_t mix64(uint64_t z) { /* splitmix64 finaliser */
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

uint64_t orc_synth_u64(uint64_t seed, int32_t column_id, int64_t row) {
  uint64_t s = seed ^ ((uint64_t)(uint32_t)column_id * 0xA0761D6478BD642Full);
  return mix64(s + ((uint64_t)row + 1) * 0x9E3779B97F4A7C15ull);
}

static inline uint64_t mulhi64(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }

int32_t orc_synth_fill(int32_t kind, int32_t column_id, double p0, double p1, uint64_t seed,
                       int64_t row_begin, int64_t n, void* out) {
  for (int64_t i = 0; i < n; ++i) {
    uint64_t r = orc_synth_u64(seed, column_id, row_begin + i);
    switch (kind) {
      case DFX_SYNTH_F64_UNIFORM: {
        double u = (double)(r >> 11) * 0x1.0p-53;
        double t = p1 * u; /* separate statements: one rounding each, no FMA (-ffp-contract=off) */
        ((double*)out)[i] = p0 + t;
        break;
      }
      case DFX_SYNTH_F64_EXACT: {
        /* m * 2^-S, m uniform integer in [0, 2^B): B = p0 (0: 20 bits), S = p1 (0: 10).  Every value, product of a
         * few values and partial sum stays exactly representable when the bit budget allows -- order-independent sums */
        const int B = p0 > 0.0 ? (int)p0 : 20, S = p1 > 0.0 ? (int)p1 : 10;
        ((double*)out)[i] = ldexp((double)(r >> (64 - B)), -S);
        break;
      }
      case DFX_SYNTH_I64_UNIFORM: ((int64_t*)out)[i] = (int64_t)mulhi64(r, (uint64_t)(int64_t)p0); break;
      case DFX_SYNTH_I32_UNIFORM: ((int32_t*)out)[i] = (int32_t)mulhi64(r, (uint64_t)(int64_t)p0); break;
      case DFX_SYNTH_I64_WIDE: ((int64_t*)out)[i] = (int64_t)((mulhi64(r, (uint64_t)(int64_t)p0) + 1ull) * 0x9E3779B97F4A7C15ull); break;
      case DFX_SYNTH_I64_ZIPF: {
        /* log-uniform skew: k = floor(2^(u * log2(G))) - 1, computed in integers:
         * pick a bit-length b uniformly in [0, ceil(log2 G)], then a uniform value below 2^b. */
        uint64_t G = (uint64_t)(int64_t)p0;
        int bits = 0;
        while ((1ull << bits) < G && bits < 62) ++bits;
        uint64_t b = mulhi64(r, (uint64_t)bits + 1);
        uint64_t r2 = mix64(r ^ 0xD6E8FEB86659FD93ull);
        uint64_t k = (b == 0) ? 0 : ((1ull << (b - 1)) + mulhi64(r2, 1ull << (b - 1)));
        if (k >= G) k = G - 1;
        ((int64_t*)out)[i] = (int64_t)k;
        break;
      }
      default: return DFX_NOT_IMPLEMENTED;
    }
  }
  return DFX_OK;
}

/* validity bitmap of a synthetic column with nulls (include/dfx.h: DFX_SYNTH_NULL_PERMILLE): LSB first, (n + 7) / 8 bytes
 * (zeroed here); returns the number of null rows */
int64_t orc_synth_validity(int32_t column_id, int32_t permille, uint64_t seed, int64_t row_begin, int64_t n, uint8_t* bits) {
  int64_t nulls = 0;
  memset(bits, 0, (size_t)((n + 7) / 8));
  for (int64_t i = 0; i < n; ++i) {
    if (mulhi64(orc_synth_u64(seed, column_id ^ DFX_SYNTH_NULL_STREAM, row_begin + i), 1000ull) >= (uint64_t)permille) bits[i >> 3] |= (uint8_t)(1u << (i & 7));
    else ++nulls;
  }
  return nulls;
}
static int synth_dtype(int32_t kind) {
  const int k = DFX_SYNTH_KIND(kind);
  return (k == DFX_SYNTH_I64_UNIFORM || k == DFX_SYNTH_I64_ZIPF || k == DFX_SYNTH_I64_WIDE) ? DFX_INT64 : k == DFX_SYNTH_I32_UNIFORM ? DFX_INT32 : DFX_FLOAT64;
}

/* ------------------------------------------------------------------------------------------ */
/* CPU baseline runner: reference-shaped pipeline over synthetic batches                       */
/* ------------------------------------------------------------------------------------------ */
static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int32_t orc_run_synth_query(const dfx_synth_column* cols, int32_t n_cols, uint64_t seed,
                            int64_t row_begin, int64_t n_rows, int64_t batch_rows,
                            const dfx_expr_node* nodes, int32_t n_nodes, int32_t filter_root,
                            const int32_t* group_roots, int32_t n_group, const int32_t* aggr_roots,
                            int32_t n_aggr, int32_t mask_only, double* seconds, orc_batch** out,
                            int64_t* rows_out, char* err, size_t errlen) {
  if (batch_rows <= 0) batch_rows = 1024; /* the width every reference fixture uses (tests/sql.rs:90) */
  orc_agg* agg = NULL;
  int32_t st = DFX_OK;
  if (!mask_only && n_aggr > 0) {
    st = orc_agg_new(nodes, n_nodes, group_roots, n_group, aggr_roots, n_aggr, &agg, err, errlen);
    if (st) return st;
  }
  orc_batch b;
  memset(&b, 0, sizeof(b));
  b.num_columns = n_cols;
  b.columns = (orc_array**)calloc((size_t)n_cols, sizeof(orc_array*));
  for (int c = 0; c < n_cols; ++c) {
    b.columns[c] = arr_new(synth_dtype(cols[c].kind), batch_rows, DFX_SYNTH_NULL_PERMILLE(cols[c].kind) != 0);
  }
  double t = 0.0;
  int64_t kept = 0;
  for (int64_t r0 = 0; r0 < n_rows && !st; r0 += batch_rows) {
    int64_t n = n_rows - r0 < batch_rows ? n_rows - r0 : batch_rows;
    for (int c = 0; c < n_cols; ++c) {
      b.columns[c]->length = n;
      orc_synth_fill(DFX_SYNTH_KIND(cols[c].kind), cols[c].column_id, cols[c].p0, cols[c].p1, seed, row_begin + r0, n, b.columns[c]->values);
      if (b.columns[c]->validity) (void)orc_synth_validity(cols[c].column_id, DFX_SYNTH_NULL_PERMILLE(cols[c].kind), seed, row_begin + r0, n, b.columns[c]->validity);
    }
    b.num_rows = n;
    double t0 = now_s();
    if (mask_only) { /* predicate only: expression closure tree -> BooleanArray */
      orc_array* m = NULL;
      st = eval_node(nodes, n_nodes, filter_root, &b, &m, err, errlen);
      if (!st) {
        const uint8_t* bits = (const uint8_t*)m->values;
        for (int64_t i = 0; i < n; ++i) kept += bit_get(bits, i);
        orc_array_free(m);
      }
    } else if (filter_root >= 0) {
      orc_batch* fb = NULL;
      st = orc_filter_next(nodes, n_nodes, filter_root, &b, &fb, err, errlen);
      if (!st) {
        kept += fb->num_rows;
        if (agg) st = orc_agg_push(agg, fb, err, errlen);
        orc_batch_free(fb);
      }
    } else {
      kept += n;
      if (agg) st = orc_agg_push(agg, &b, err, errlen);
    }
    t += now_s() - t0;
  }
  if (!st && agg) {
    double t0 = now_s();
    orc_batch* res = NULL;
    st = orc_agg_finish(agg, &res, err, errlen);
    t += now_s() - t0;
    if (!st) { if (out) *out = res; else orc_batch_free(res); }
  }
  for (int c = 0; c < n_cols; ++c) orc_array_free(b.columns[c]);
  free(b.columns);
  orc_agg_free(agg);
  if (seconds) *seconds = t;
  if (rows_out) *rows_out = kept;
  return st;
}

/* FilterRelation (filter.rs:46-110) reference-shaped over synthetic columns: batch_rows-row batches (0: 1024), predicate
 * closure tree -> BooleanArray, fn filter per column; the compacted batches are concatenated into the caller's buffers.
 *   out_values[c]: room for n_rows 8-byte elements per column (every synthetic column is Int64 / Float64), or NULL
 *   mask_bits:     the concatenated BooleanArray of the predicate, LSB first, (n_rows + 7) / 8 bytes (zeroed here), or NULL
 * Test infrastructure for the parity tests at the benchmark's batch sizes (tests/test_gpu_scale.py, bench.py). */
int32_t orc_run_synth_filter(const dfx_synth_column* cols, int32_t n_cols, uint64_t seed, int64_t row_begin,
                             int64_t n_rows, int64_t batch_rows, const dfx_expr_node* nodes, int32_t n_nodes,
                             int32_t filter_root, void** out_values, uint8_t* mask_bits, int64_t* rows_out,
                             double* seconds, char* err, size_t errlen) {
  if (batch_rows <= 0) batch_rows = 1024;
  int32_t st = DFX_OK;
  orc_batch b;
  memset(&b, 0, sizeof(b));
  b.num_columns = n_cols;
  b.columns = (orc_array**)calloc((size_t)n_cols, sizeof(orc_array*));
  for (int c = 0; c < n_cols; ++c) {
    b.columns[c] = arr_new(synth_dtype(cols[c].kind), batch_rows, DFX_SYNTH_NULL_PERMILLE(cols[c].kind) != 0);
  }
  if (mask_bits) memset(mask_bits, 0, (size_t)((n_rows + 7) / 8));
  double t = 0.0;
  int64_t kept = 0;
  for (int64_t r0 = 0; r0 < n_rows && !st; r0 += batch_rows) {
    int64_t n = n_rows - r0 < batch_rows ? n_rows - r0 : batch_rows;
    for (int c = 0; c < n_cols; ++c) {
      b.columns[c]->length = n;
      orc_synth_fill(DFX_SYNTH_KIND(cols[c].kind), cols[c].column_id, cols[c].p0, cols[c].p1, seed, row_begin + r0, n, b.columns[c]->values);
      if (b.columns[c]->validity) (void)orc_synth_validity(cols[c].column_id, DFX_SYNTH_NULL_PERMILLE(cols[c].kind), seed, row_begin + r0, n, b.columns[c]->validity);
    }
    b.num_rows = n;
    double t0 = now_s();
    if (mask_bits) {
      orc_array* m = NULL;
      st = eval_node(nodes, n_nodes, filter_root, &b, &m, err, errlen);
      if (st) break;
      if (m->dtype != DFX_BOOLEAN) {
        orc_array_free(m);
        st = fail(err, errlen, DFX_EXECUTION_ERROR, "Filter expression did not evaluate to boolean");
        break;
      }
      const uint8_t* bits = (const uint8_t*)m->values;
      for (int64_t i = 0; i < n; ++i)
        if (bit_get(bits, i)) mask_bits[(r0 + i) >> 3] |= (uint8_t)(1u << ((r0 + i) & 7));
      orc_array_free(m);
    }
    orc_batch* fb = NULL;
    st = orc_filter_next(nodes, n_nodes, filter_root, &b, &fb, err, errlen);
    t += now_s() - t0;
    if (st) break;
    if (out_values)
      for (int c = 0; c < n_cols; ++c)
        if (out_values[c]) memcpy((uint8_t*)out_values[c] + (size_t)kept * 8, fb->columns[c]->values, (size_t)fb->num_rows * 8);
    kept += fb->num_rows;
    orc_batch_free(fb);
  }
  for (int c = 0; c < n_cols; ++c) orc_array_free(b.columns[c]);
  free(b.columns);
  if (seconds) *seconds = t;
  if (rows_out) *rows_out = kept;
  return st;
}

/* ------------------------------------------------------------------------------------------ */
/* CsvDataSource (src/execution/datasource.rs:33-58) = arrow 0.12 csv::Reader over the `csv`    */
/* crate (req "1", Cargo.toml:30 via arrow) with has_headers = true, batch_size, no projection. */
/*                                                                                            */
/* Restated from the published sources (neither crate is vendored under the reference tree):   */
/*   csv-core reader NFA with default settings: delimiter ',', quote '"', double_quote = true, */
/*     quoting on, no escape, no comment, terminator CRLF (= \r, \n or \r\n), flexible = false; */
/*     a quote opens a quoted field only as the first byte of a field, bytes after the closing */
/*     quote continue the field unquoted, empty lines are skipped, the last record needs no    */
/*     terminator;                                                                             */
/*   arrow 0.12 csv::Reader::next: reads up to batch_size records; primitive cell: "" -> null,  */
/*     else s.parse::<T>() or ParseError("Error while parsing value {s} at line {n}"); Utf8    */
/*     cell: the string (never null; a record shorter than the schema gives "");              */
/*   Rust 2019 str::parse: ints [+-]?digits (unsigned rejects '-'), floats dec2flt grammar +    */
/*     "inf"/"NaN", correctly rounded (glibc strtod is, too: used here after a grammar check); */
/*     bool "true"/"false".                                                                    */
/* Pinned by: the row counts / values every reference test reads from test/data (the .csv files)   */
/* (tests/sql.rs:29-77, aggregate.rs:965-1127, projection.rs:83-103) -- e.g. uk_cities.csv has   */
/* 37 lines and the tests see 36 rows.  Unpinned: quoting corner cases, errors, nulls.          */
/* ------------------------------------------------------------------------------------------ */
typedef struct { char* p; size_t n, cap; } sbuf;
static void sb_put(sbuf* b, char c) {
  if (b->n + 1 > b->cap) { b->cap = b->cap ? b->cap * 2 : 64; b->p = (char*)realloc(b->p, b->cap); }
  b->p[b->n++] = c;
}
typedef struct { sbuf text; size_t* ends; int nf, cap; } csv_record; /* fields back to back, ends[i] = end of field i */
static void rec_end_field(csv_record* r) {
  if (r->nf + 1 > r->cap) { r->cap = r->cap ? r->cap * 2 : 16; r->ends = (size_t*)realloc(r->ends, sizeof(size_t) * (size_t)r->cap); }
  r->ends[r->nf++] = r->text.n;
}

/* reads the next record starting at *pos; returns 0 at end of input */
static int csv_next_record(const uint8_t* buf, size_t n, size_t* pos, csv_record* r) {
  enum { START_RECORD, START_FIELD, IN_FIELD, IN_QUOTED, QUOTE_IN_QUOTED } st = START_RECORD;
  size_t i = *pos;
  r->text.n = 0;
  r->nf = 0;
  for (; i < n; ++i) {
    const uint8_t c = buf[i];
    const int is_t = c == '\n' || c == '\r';
    switch (st) {
      case START_RECORD:
        if (is_t) continue; /* empty line */
        __attribute__((fallthrough)); /* the first byte of the record is the first byte of a field */
      case START_FIELD:
        if (c == '"') st = IN_QUOTED;
        else if (c == ',') { rec_end_field(r); st = START_FIELD; }
        else if (is_t) { rec_end_field(r); *pos = i + 1; return 1; }
        else { sb_put(&r->text, (char)c); st = IN_FIELD; }
        break;
      case IN_FIELD:
        if (c == ',') { rec_end_field(r); st = START_FIELD; }
        else if (is_t) { rec_end_field(r); *pos = i + 1; return 1; }
        else sb_put(&r->text, (char)c); /* a quote here is a literal */
        break;
      case IN_QUOTED:
        if (c == '"') st = QUOTE_IN_QUOTED;
        else sb_put(&r->text, (char)c);
        break;
      case QUOTE_IN_QUOTED:
        if (c == '"') { sb_put(&r->text, '"'); st = IN_QUOTED; }
        else if (c == ',') { rec_end_field(r); st = START_FIELD; }
        else if (is_t) { rec_end_field(r); *pos = i + 1; return 1; }
        else { sb_put(&r->text, (char)c); st = IN_FIELD; }
        break;
    }
  }
  *pos = n;
  if (st == START_RECORD) return 0;
  rec_end_field(r); /* last record without a terminator */
  return 1;
}

static int rust_float_grammar(const char* s, size_t n, int* special) {
  size_t i = 0, nd = 0;
  *special = 0;
  if (n == 0) return 0;
  if (s[0] == '+' || s[0] == '-') i = 1;
  if (n - i == 3 && !memcmp(s + i, "inf", 3)) { *special = 1; return 1; }
  if (n - i == 3 && !memcmp(s + i, "NaN", 3)) { *special = 2; return 1; }
  while (i < n && s[i] >= '0' && s[i] <= '9') { ++i; ++nd; }
  if (i < n && s[i] == '.') { ++i; while (i < n && s[i] >= '0' && s[i] <= '9') { ++i; ++nd; } }
  if (nd == 0) return 0;
  if (i < n && (s[i] == 'e' || s[i] == 'E')) {
    size_t ne = 0;
    ++i;
    if (i < n && (s[i] == '+' || s[i] == '-')) ++i;
    while (i < n && s[i] >= '0' && s[i] <= '9') { ++i; ++ne; }
    if (ne == 0) return 0;
  }
  return i == n;
}

/* s.parse::<T>() for one cell; returns 0 on failure */
static int csv_parse_cell(int dt, const char* s, size_t n, orc_array* a, int64_t row) {
  char tmp[512];
  if (n >= sizeof(tmp)) return 0;
  memcpy(tmp, s, n);
  tmp[n] = 0;
  if (dt == DFX_FLOAT64 || dt == DFX_FLOAT32) {
    int special;
    if (!rust_float_grammar(tmp, n, &special)) return 0;
    const int neg = tmp[0] == '-';
    if (dt == DFX_FLOAT64) {
      double d = special == 1 ? INFINITY : special == 2 ? NAN : strtod(tmp, NULL);
      if (special && neg) d = -d;
      ((double*)a->values)[row] = d;
    } else {
      float f = special == 1 ? INFINITY : special == 2 ? NAN : strtof(tmp, NULL);
      if (special && neg) f = -f;
      ((float*)a->values)[row] = f;
    }
    return 1;
  }
  if (dt == DFX_BOOLEAN) {
    if (!strcmp(tmp, "true")) { bit_set((uint8_t*)a->values, row); return 1; }
    return !strcmp(tmp, "false");
  }
  { /* integers */
    const int is_signed = dt >= DFX_INT8 && dt <= DFX_INT64;
    const int bits = (int)dt_size(dt) * 8;
    size_t i = 0;
    int neg = 0;
    unsigned __int128 v = 0;
    if (tmp[0] == '+' || tmp[0] == '-') { neg = tmp[0] == '-'; i = 1; }
    if (neg && !is_signed) return 0;
    if (i == n) return 0;
    for (; i < n; ++i) {
      if (tmp[i] < '0' || tmp[i] > '9') return 0;
      v = v * 10 + (unsigned)(tmp[i] - '0');
      if (v > ((unsigned __int128)1 << 64)) return 0;
    }
    const unsigned __int128 maxpos = is_signed ? (((unsigned __int128)1 << (bits - 1)) - 1) : (((unsigned __int128)1 << bits) - 1);
    if (v > (neg ? maxpos + 1 : maxpos)) return 0;
    const int64_t sv = neg ? (int64_t)(0 - (uint64_t)v) : (int64_t)(uint64_t)v;
    switch (dt) {
      case DFX_INT8: ((int8_t*)a->values)[row] = (int8_t)sv; break;
      case DFX_INT16: ((int16_t*)a->values)[row] = (int16_t)sv; break;
      case DFX_INT32: ((int32_t*)a->values)[row] = (int32_t)sv; break;
      case DFX_INT64: ((int64_t*)a->values)[row] = sv; break;
      case DFX_UINT8: ((uint8_t*)a->values)[row] = (uint8_t)v; break;
      case DFX_UINT16: ((uint16_t*)a->values)[row] = (uint16_t)v; break;
      case DFX_UINT32: ((uint32_t*)a->values)[row] = (uint32_t)v; break;
      default: ((uint64_t*)a->values)[row] = (uint64_t)v; break;
    }
    return 1;
  }
}

struct orc_csv {
  uint8_t* buf;
  size_t n, pos;
  int32_t n_cols;
  int32_t* dtypes;
  int64_t batch_size;
  int64_t line_number; /* arrow: 1 after the header */
  int expected_fields;
  int started;
};

int32_t orc_csv_open(const char* filename, const int32_t* dtypes, int32_t n_cols, int64_t batch_size, orc_csv** out,
                     char* err, size_t errlen) {
  FILE* fp = fopen(filename, "rb");
  if (!fp) return fail(err, errlen, DFX_INTERNAL_ERROR, "called `Result::unwrap()` on an `Err` value: could not open %s", filename);
  orc_csv* c = (orc_csv*)calloc(1, sizeof(orc_csv));
  fseek(fp, 0, SEEK_END);
  c->n = (size_t)ftell(fp);
  fseek(fp, 0, SEEK_SET);
  c->buf = (uint8_t*)malloc(c->n + 1);
  if (fread(c->buf, 1, c->n, fp) != c->n) { fclose(fp); free(c->buf); free(c); return fail(err, errlen, DFX_IO_ERROR, "short read"); }
  fclose(fp);
  c->n_cols = n_cols;
  c->dtypes = (int32_t*)malloc(sizeof(int32_t) * (size_t)n_cols);
  memcpy(c->dtypes, dtypes, sizeof(int32_t) * (size_t)n_cols);
  c->batch_size = batch_size;
  c->line_number = 1;
  c->expected_fields = -1;
  *out = c;
  return DFX_OK;
}

void orc_csv_close(orc_csv* c) {
  if (!c) return;
  free(c->buf);
  free(c->dtypes);
  free(c);
}

/* csv::Reader::next: Ok(None) -> *out = NULL */
int32_t orc_csv_next(orc_csv* c, orc_batch** out, char* err, size_t errlen) {
  *out = NULL;
  csv_record rec;
  memset(&rec, 0, sizeof(rec));
  if (!c->started) { /* has_headers = true: the first record is consumed whatever it holds */
    c->started = 1;
    if (csv_next_record(c->buf, c->n, &c->pos, &rec)) c->expected_fields = rec.nf;
  }
  /* collect up to batch_size records (copies: the record buffer is reused) */
  const int64_t cap = c->batch_size > 0 ? c->batch_size : 1024;
  char** texts = (char**)calloc((size_t)cap, sizeof(char*));
  size_t** ends = (size_t**)calloc((size_t)cap, sizeof(size_t*));
  int* nfs = (int*)calloc((size_t)cap, sizeof(int));
  int64_t rows = 0;
  int32_t st = DFX_OK;
  while (rows < cap && csv_next_record(c->buf, c->n, &c->pos, &rec)) {
    if (c->expected_fields >= 0 && rec.nf != c->expected_fields) {
      st = fail(err, errlen, DFX_ARROW_ERROR, "Error parsing line %lld: UnequalLengths { expected_len: %d, len: %d }",
                (long long)(c->line_number + rows), c->expected_fields, rec.nf);
      break;
    }
    if (c->expected_fields < 0) c->expected_fields = rec.nf;
    texts[rows] = (char*)malloc(rec.text.n + 1);
    memcpy(texts[rows], rec.text.p, rec.text.n);
    ends[rows] = (size_t*)malloc(sizeof(size_t) * (size_t)(rec.nf > 0 ? rec.nf : 1));
    memcpy(ends[rows], rec.ends, sizeof(size_t) * (size_t)rec.nf);
    nfs[rows] = rec.nf;
    ++rows;
  }
  orc_batch* b = NULL;
  if (!st && rows > 0) {
    b = (orc_batch*)calloc(1, sizeof(orc_batch));
    b->owned = 1;
    b->num_rows = rows;
    b->num_columns = c->n_cols;
    b->columns = (orc_array**)calloc((size_t)c->n_cols, sizeof(orc_array*));
    for (int col = 0; col < c->n_cols && !st; ++col) {
      const int dt = c->dtypes[col];
      orc_array* a = arr_new(dt, rows, dt != DFX_UTF8);
      b->columns[col] = a;
      if (dt == DFX_UTF8) {
        size_t total = 0;
        for (int64_t r = 0; r < rows; ++r) {
          const size_t fb = col < nfs[r] ? (col ? ends[r][col - 1] : 0) : 0, fe = col < nfs[r] ? ends[r][col] : 0;
          total += fe - fb;
        }
        a->data = (uint8_t*)malloc(total + 1);
        size_t at = 0;
        for (int64_t r = 0; r < rows; ++r) {
          const size_t fb = col < nfs[r] ? (col ? ends[r][col - 1] : 0) : 0, fe = col < nfs[r] ? ends[r][col] : 0;
          a->offsets[r] = (int32_t)at;
          memcpy(a->data + at, texts[r] + fb, fe - fb);
          at += fe - fb;
        }
        a->offsets[rows] = (int32_t)at;
        continue;
      }
      int any_null = 0;
      for (int64_t r = 0; r < rows; ++r) {
        if (col >= nfs[r]) { any_null = 1; continue; } /* rows[i].get(col) == None */
        const size_t fb = col ? ends[r][col - 1] : 0, fe = ends[r][col];
        if (fe == fb) { any_null = 1; continue; }        /* "" -> append_null */
        if (!csv_parse_cell(dt, texts[r] + fb, fe - fb, a, r)) {
          st = fail(err, errlen, DFX_ARROW_ERROR, "Error while parsing value %.*s at line %lld", (int)(fe - fb), texts[r] + fb,
                    (long long)(c->line_number + r));
          break;
        }
        bit_set(a->validity, r);
      }
      if (!any_null && !st) { free(a->validity); a->validity = NULL; }
    }
  }
  c->line_number += rows;
  for (int64_t r = 0; r < rows; ++r) { free(texts[r]); free(ends[r]); }
  free(texts); free(ends); free(nfs);
  free(rec.text.p); free(rec.ends);
  if (st) { orc_batch_free(b); return st; }
  *out = b;
  return DFX_OK;
}

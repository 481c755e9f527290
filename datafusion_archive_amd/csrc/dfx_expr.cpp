// dfx_expr.cpp -- the expression compiler.
//   dfx_compile_scalar_expr  <->  compile_scalar_expr (src/execution/expression.rs:283-505)
//   dfx_compile_expr         <->  compile_expr        (src/execution/expression.rs:80-121)
//   ProgramBuilder: lowers validated Expr trees to the fused device program (dfx_device.hpp).
//
// Deviations from the reference (each is a place where the reference panics / is unimplemented):
//   D1  numeric->numeric CAST of any scalar expression is implemented (Rust `as`); the reference
//       implements column->Int16/Int32 and literal Int64->Float64 only (expression.rs:272-280,
//       :345-368), `unimplemented!()` / NotImplemented / General otherwise.
//   D6  RuntimeExpr type of a column cast is the TARGET type; the reference reports the source
//       column's type (expression.rs:324, a bug: the array it builds has the target type).
#include <charconv>
#include <string.h>
#include <strings.h>

#include "dfx_host.hpp"

namespace dfx {
namespace {

const char* op_debug(int op) {  // Rust {:?} of logicalplan::Operator
  static const char* names[] = {"Eq", "NotEq", "Lt", "LtEq", "Gt", "GtEq", "Plus", "Minus", "Multiply",
                                "Divide", "Modulus", "And", "Or", "Not", "Like", "NotLike"};
  return (op >= 0 && op <= DFX_OP_NOT_LIKE) ? names[op] : "?";
}

std::string float_repr(double v, bool debug) {
  if (v != v) return "NaN";
  if (v == __builtin_inf()) return "inf";
  if (v == -__builtin_inf()) return "-inf";
  char buf[512];
  auto r = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::fixed);
  std::string s(buf, r.ptr);
  if (debug && s.find('.') == std::string::npos) s += ".0";
  return s;
}

std::string float_repr32(float v, bool debug) {
  if (v != v) return "NaN";
  char buf[256];
  auto r = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::fixed);
  std::string s(buf, r.ptr);
  if (debug && s.find('.') == std::string::npos) s += ".0";
  return s;
}

std::string literal_repr(const dfx_expr_node& n, bool debug) {  // Display (debug=false) or Debug of the ScalarValue
  std::string v;
  switch (n.dtype) {
    case DFX_FLOAT64: v = float_repr(n.lit.f64, debug); break;
    case DFX_FLOAT32: v = float_repr32(n.lit.f32, debug); break;
    case DFX_INT8: case DFX_INT16: case DFX_INT32: case DFX_INT64: v = std::to_string((long long)n.lit.i64); break;
    case DFX_UINT8: case DFX_UINT16: case DFX_UINT32: case DFX_UINT64: v = std::to_string((unsigned long long)n.lit.u64); break;
    case DFX_BOOLEAN: v = n.lit.u64 ? "true" : "false"; break;
    case DFX_UTF8: v = std::string("\"") + (n.name ? n.name : "") + "\""; break;
    default: return "Null";
  }
  if (!debug) return v;
  return std::string(dtype_name(n.dtype)) + "(" + v + ")";
}

std::string expr_debug(const std::vector<dfx_expr_node>& nodes, int32_t idx) {  // logicalplan.rs:264-309
  if (idx < 0 || idx >= (int32_t)nodes.size()) return "?";
  const dfx_expr_node& n = nodes[idx];
  switch (n.kind) {
    case DFX_EXPR_COLUMN: return "#" + std::to_string(n.column);
    case DFX_EXPR_LITERAL: return literal_repr(n, true);
    case DFX_EXPR_CAST: return "CAST(" + expr_debug(nodes, n.left) + " AS " + dtype_name(n.dtype) + ")";
    case DFX_EXPR_IS_NULL: return expr_debug(nodes, n.left) + " IS NULL";
    case DFX_EXPR_IS_NOT_NULL: return expr_debug(nodes, n.left) + " IS NOT NULL";
    case DFX_EXPR_BINARY:
      return expr_debug(nodes, n.left) + " " + op_debug(n.op) + " " + expr_debug(nodes, n.right);
    case DFX_EXPR_SORT: return expr_debug(nodes, n.left) + (n.reserved ? " DESC" : " ASC");
    case DFX_EXPR_SCALAR_FUNCTION:
    case DFX_EXPR_AGGREGATE_FUNCTION:
      return std::string(n.name ? n.name : "") + "(" + (n.left >= 0 ? expr_debug(nodes, n.left) : "") + ")";
    default: return "?";
  }
}

// static result type of a scalar node (the type of the array the closure would produce)
Status node_type(const std::vector<dfx_expr_node>& nodes, int32_t idx, const SchemaInfo& schema, int* out) {
  const dfx_expr_node& n = nodes[idx];
  switch (n.kind) {
    case DFX_EXPR_COLUMN: *out = schema.fields[n.column].dtype; return Status::OK();
    case DFX_EXPR_LITERAL: *out = n.dtype; return Status::OK();
    case DFX_EXPR_CAST: *out = n.dtype; return Status::OK();
    case DFX_EXPR_BINARY:
      if (n.op <= DFX_OP_GT_EQ || n.op == DFX_OP_AND || n.op == DFX_OP_OR) {
        *out = DFX_BOOLEAN;
        return Status::OK();
      }
      return node_type(nodes, n.left, schema, out);  // op_type = left_expr.get_type() (expression.rs:408)
    default: *out = DFX_TYPE_NONE; return Status::OK();
  }
}

// compile_scalar_expr's validation (errors mirror expression.rs:283-505); fills the display name
Status validate_scalar(const std::vector<dfx_expr_node>& nodes, int32_t idx, const SchemaInfo& schema,
                       std::string* name, int depth = 0) {
  if (idx < 0 || idx >= (int32_t)nodes.size() || depth > 64)
    return Status::Err(DFX_INTERNAL_ERROR, strfmt("expression node index %d out of range", idx));
  const dfx_expr_node& n = nodes[idx];
  switch (n.kind) {
    case DFX_EXPR_LITERAL:
      if (!dtype_is_numeric(n.dtype))  // expression.rs:306-309
        return Status::Err(DFX_EXECUTION_ERROR, "No support for literal type " + literal_repr(n, true));
      if (name) *name = literal_repr(n, false);  // format!("{}", nn)
      return Status::OK();
    case DFX_EXPR_COLUMN:
      if (n.column < 0 || n.column >= (int32_t)schema.fields.size())  // schema.field(index) panics
        return Status::Err(DFX_INTERNAL_ERROR, strfmt("index out of bounds: column %d of a schema with %zu fields",
                                                      n.column, schema.fields.size()));
      if (name) *name = schema.fields[n.column].name;
      return Status::OK();
    case DFX_EXPR_CAST: {
      std::string child_name;
      DFX_RETURN_IF_ERROR(validate_scalar(nodes, n.left, schema, &child_name, depth + 1));
      int from = DFX_TYPE_NONE;
      DFX_RETURN_IF_ERROR(node_type(nodes, n.left, schema, &from));
      const dfx_expr_node& c = nodes[n.left];
      if (!dtype_is_numeric(from))  // expression.rs:336 panic!("unsupported CAST operation")
        return Status::Err(DFX_INTERNAL_ERROR, "unsupported CAST operation");
      if (!dtype_is_numeric(n.dtype))  // unimplemented!() (:277) / NotImplemented (:363-372)
        return Status::Err(DFX_NOT_IMPLEMENTED, strfmt("CAST from %s to %s", dtype_name(from), dtype_name(n.dtype)));
      if (name) *name = c.kind == DFX_EXPR_COLUMN ? child_name : (c.kind == DFX_EXPR_LITERAL ? "lit" : "cast");
      return Status::OK();
    }
    case DFX_EXPR_BINARY: {
      DFX_RETURN_IF_ERROR(validate_scalar(nodes, n.left, schema, nullptr, depth + 1));
      DFX_RETURN_IF_ERROR(validate_scalar(nodes, n.right, schema, nullptr, depth + 1));
      const bool ok = (n.op >= DFX_OP_EQ && n.op <= DFX_OP_DIVIDE) || n.op == DFX_OP_AND || n.op == DFX_OP_OR;
      if (!ok)  // expression.rs:494-497
        return Status::Err(DFX_EXECUTION_ERROR, std::string("operator: ") + op_debug(n.op));
      if (name) *name = expr_debug(nodes, idx);  // format!("{:?} {:?} {:?}", left, op, right)
      return Status::OK();
    }
    default:  // IsNull / IsNotNull / Sort / ScalarFunction / AggregateFunction (expression.rs:500-503)
      return Status::Err(DFX_EXECUTION_ERROR, "expression " + expr_debug(nodes, idx));
  }
}

Status copy_tree(const dfx_expr_node* nodes, int32_t n_nodes, int32_t root, dfx_runtime_expr* e) {
  if (!nodes || n_nodes <= 0 || root < 0 || root >= n_nodes)
    return Status::Err(DFX_INTERNAL_ERROR, "invalid expression tree");
  e->nodes.assign(nodes, nodes + n_nodes);
  e->strings.assign((size_t)n_nodes, std::string());
  e->has_name.assign((size_t)n_nodes, 0);
  for (int32_t i = 0; i < n_nodes; ++i) {
    if (nodes[i].name) {
      e->strings[i] = nodes[i].name;
      e->has_name[i] = 1;
    }
  }
  e->rebind();
  e->root = root;
  return Status::OK();
}

}  // namespace

// -------------------------------------------------------------------------------------------------
// ProgramBuilder
// -------------------------------------------------------------------------------------------------
ProgramBuilder::ProgramBuilder(const SchemaInfo& schema) : schema_(schema) {
  memset(&prog_, 0, sizeof(prog_));
}

Status ProgramBuilder::add(const dfx_runtime_expr& e, int32_t root, uint8_t* operand, int* dtype) {
  return emit(e, root, operand, dtype);
}

Status ProgramBuilder::emit(const dfx_runtime_expr& e, int32_t idx, uint8_t* operand, int* dtype) {
  const dfx_expr_node& n = e.nodes[idx];
  switch (n.kind) {
    case DFX_EXPR_COLUMN: {
      const int dt = schema_.fields[n.column].dtype;
      if (dt == DFX_UTF8)
        return Status::Err(DFX_NOT_IMPLEMENTED, "Utf8 columns cannot be used in device expressions");
      int slot = -1;
      for (size_t i = 0; i < cols_.size(); ++i)
        if (cols_[i] == n.column) slot = (int)i;
      if (slot < 0) {
        if ((int)cols_.size() >= kMaxCols)
          return Status::Err(DFX_NOT_IMPLEMENTED, strfmt("fused expression reads more than %d columns", kMaxCols));
        slot = (int)cols_.size();
        cols_.push_back(n.column);
        prog_.col_dtype[slot] = (uint8_t)dt;
        prog_.n_cols = (int32_t)cols_.size();
      }
      *operand = make_operand(OPK_COL, slot);
      *dtype = dt;
      return Status::OK();
    }
    case DFX_EXPR_LITERAL: {
      uint64_t bits = 0;
      switch (n.dtype) {
        case DFX_FLOAT64: memcpy(&bits, &n.lit.f64, 8); break;
        case DFX_FLOAT32: { uint32_t b; memcpy(&b, &n.lit.f32, 4); bits = b; break; }
        case DFX_INT8: case DFX_INT16: case DFX_INT32: case DFX_INT64: bits = (uint64_t)n.lit.i64; break;
        default: bits = n.lit.u64; break;
      }
      int slot = -1;
      for (int i = 0; i < prog_.n_imm; ++i)
        if (prog_.imm[i] == bits) slot = i;
      if (slot < 0) {
        if (prog_.n_imm >= kMaxImm)
          return Status::Err(DFX_NOT_IMPLEMENTED, strfmt("fused expression uses more than %d literals", kMaxImm));
        slot = prog_.n_imm++;
        prog_.imm[slot] = bits;
      }
      *operand = make_operand(OPK_IMM, slot);
      *dtype = n.dtype;
      return Status::OK();
    }
    case DFX_EXPR_CAST:
    case DFX_EXPR_BINARY: {
      DevIns ins;
      memset(&ins, 0, sizeof(ins));
      uint8_t a = 0, b = 0;
      int ta = 0, tb = 0;
      DFX_RETURN_IF_ERROR(emit(e, n.left, &a, &ta));
      if (n.kind == DFX_EXPR_CAST) {
        if (ta == n.dtype) {  // `as` to the same type is the identity
          *operand = a;
          *dtype = ta;
          return Status::OK();
        }
        ins.op = DOP_CAST;
        ins.t = (uint8_t)ta;
        ins.a = a;
        ins.b = (uint8_t)n.dtype;
        *dtype = n.dtype;
      } else {
        DFX_RETURN_IF_ERROR(emit(e, n.right, &b, &tb));
        if (n.op == DFX_OP_AND || n.op == DFX_OP_OR) {
          if (ta != DFX_BOOLEAN || tb != DFX_BOOLEAN)  // downcast_ref::<BooleanArray>().unwrap() (expression.rs:217-221)
            return Status::Err(DFX_INTERNAL_ERROR, "called `Option::unwrap()` on a `None` value (boolean_ops operand is not a BooleanArray)");
          ins.op = n.op == DFX_OP_AND ? DOP_AND : DOP_OR;
          *dtype = DFX_BOOLEAN;
        } else if (n.op <= DFX_OP_GT_EQ) {
          if (ta != tb || !dtype_is_numeric(ta))  // expression.rs:207
            return Status::Err(DFX_EXECUTION_ERROR, "comparison_ops");
          ins.op = (uint8_t)n.op;
          *dtype = DFX_BOOLEAN;
        } else {
          if (ta != tb || !dtype_is_numeric(ta))  // expression.rs:166
            return Status::Err(DFX_EXECUTION_ERROR, "math_ops");
          ins.op = (uint8_t)(DOP_ADD + (n.op - DFX_OP_PLUS));
          *dtype = ta;
        }
        ins.t = (uint8_t)ta;
        ins.a = a;
        ins.b = b;
      }
      for (int i = 0; i < prog_.n_ins; ++i) {  // common subexpression: identical instruction
        const DevIns& o = prog_.ins[i];
        if (o.op == ins.op && o.t == ins.t && o.a == ins.a && o.b == ins.b) {
          *operand = make_operand(OPK_REG, i);
          return Status::OK();
        }
      }
      if (prog_.n_ins >= kMaxRegs)
        return Status::Err(DFX_NOT_IMPLEMENTED, strfmt("fused expression needs more than %d computed values", kMaxRegs));
      prog_.ins[prog_.n_ins] = ins;
      *operand = make_operand(OPK_REG, prog_.n_ins);
      prog_.n_ins++;
      return Status::OK();
    }
    default:
      return Status::Err(DFX_EXECUTION_ERROR, "expression " + expr_debug(e.nodes, idx));
  }
}

Status ProgramBuilder::bind(const DeviceBatch& batch, DevProgram* prog, DevColumns* cols) const {
  *prog = prog_;
  memset(cols, 0, sizeof(*cols));
  prog->has_nulls = 0;
  for (size_t i = 0; i < cols_.size(); ++i) {
    const int ci = cols_[i];
    if (ci >= (int)batch.columns.size())
      return Status::Err(DFX_INTERNAL_ERROR, strfmt("batch has %zu columns, expression reads column %d", batch.columns.size(), ci));
    const DeviceColumn& c = batch.columns[ci];
    if (c.dtype != prog_.col_dtype[i])  // downcast_ref::<T>() of a column whose array has another type
      return Status::Err(DFX_INTERNAL_ERROR, strfmt("Column at index %d is not of expected type", ci));
    cols->c[i].values = c.values;
    cols->c[i].validity = c.null_count != 0 ? c.validity : nullptr;
    cols->c[i].bit_offset = c.bit_offset;
    if (cols->c[i].validity) prog->has_nulls = 1;
  }
  return Status::OK();
}

}  // namespace dfx

// -------------------------------------------------------------------------------------------------
// C ABI
// -------------------------------------------------------------------------------------------------
using namespace dfx;

extern "C" {

int32_t dfx_abi_version(void) { return DFX_ABI_VERSION; }

int32_t dfx_compile_scalar_expr(const dfx_expr_node* nodes, int32_t n_nodes, int32_t root,
                                const struct ArrowSchema* input_schema, dfx_runtime_expr** out, char* err,
                                size_t errlen) {
  if (!out) return to_c(Status::Err(DFX_INTERNAL_ERROR, "null output"), err, errlen);
  *out = nullptr;
  SchemaInfo schema;
  Status st = schema_from_arrow(input_schema, &schema);
  if (!st.ok()) return to_c(st, err, errlen);
  std::unique_ptr<dfx_runtime_expr> e(new dfx_runtime_expr());
  st = copy_tree(nodes, n_nodes, root, e.get());
  if (!st.ok()) return to_c(st, err, errlen);
  st = validate_scalar(e->nodes, e->root, schema, &e->name);
  if (!st.ok()) return to_c(st, err, errlen);
  st = node_type(e->nodes, e->root, schema, &e->dtype);
  if (!st.ok()) return to_c(st, err, errlen);
  *out = e.release();
  return DFX_OK;
}

int32_t dfx_compile_expr(const dfx_expr_node* nodes, int32_t n_nodes, int32_t root,
                         const struct ArrowSchema* input_schema, dfx_runtime_expr** out, char* err,
                         size_t errlen) {
  if (!nodes || root < 0 || root >= n_nodes)
    return to_c(Status::Err(DFX_INTERNAL_ERROR, "invalid expression tree"), err, errlen);
  if (nodes[root].kind != DFX_EXPR_AGGREGATE_FUNCTION)  // expression.rs:119
    return dfx_compile_scalar_expr(nodes, n_nodes, root, input_schema, out, err, errlen);
  if (!out) return to_c(Status::Err(DFX_INTERNAL_ERROR, "null output"), err, errlen);
  *out = nullptr;
  SchemaInfo schema;
  Status st = schema_from_arrow(input_schema, &schema);
  if (!st.ok()) return to_c(st, err, errlen);
  std::unique_ptr<dfx_runtime_expr> e(new dfx_runtime_expr());
  st = copy_tree(nodes, n_nodes, root, e.get());
  if (!st.ok()) return to_c(st, err, errlen);
  const dfx_expr_node& n = e->nodes[root];
  if (n.n_args != 1)  // assert_eq!(1, args.len()) (expression.rs:91)
    return to_c(Status::Err(DFX_INTERNAL_ERROR, strfmt("assertion failed: `(left == right)` left: `1`, right: `%d`", n.n_args)), err, errlen);
  st = validate_scalar(e->nodes, n.left, schema, nullptr);
  if (!st.ok()) return to_c(st, err, errlen);
  const char* nm = n.name ? n.name : "";
  int f = -1;
  if (!strcasecmp(nm, "min")) f = AGG_MIN;
  else if (!strcasecmp(nm, "max")) f = AGG_MAX;
  else if (!strcasecmp(nm, "count")) f = AGG_COUNT;
  else if (!strcasecmp(nm, "sum")) f = AGG_SUM;
  if (f < 0)  // expression.rs:103-106
    return to_c(Status::Err(DFX_GENERAL, std::string("Unsupported aggregate function '") + nm + "'"), err, errlen);
  e->is_aggregate = true;
  e->agg_func = f;
  e->agg_arg = n.left;
  e->agg_type = n.dtype;
  e->dtype = n.dtype;
  e->name = nm;
  *out = e.release();
  return DFX_OK;
}

const char* dfx_runtime_expr_name(const dfx_runtime_expr* e) { return e ? e->name.c_str() : ""; }
int32_t dfx_runtime_expr_type(const dfx_runtime_expr* e) { return e ? e->dtype : DFX_TYPE_NONE; }
int32_t dfx_runtime_expr_is_aggregate(const dfx_runtime_expr* e) { return (e && e->is_aggregate) ? 1 : 0; }
void dfx_runtime_expr_free(dfx_runtime_expr* e) { delete e; }

}  // extern "C"

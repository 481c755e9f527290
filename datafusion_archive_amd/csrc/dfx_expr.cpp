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
#include "dfx_kernels.hpp"

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

// Rust `as` on the canonical 64-bit image (same table as the device cast_value / oracle cast_val):
// used to fold CAST(literal) at compile time
int64_t sat_i64(double x, int64_t lo, int64_t hi) {
  if (x != x) return 0;
  if (x <= (double)lo) return lo;
  if (x >= (double)hi) return hi;
  return (int64_t)x;
}
uint64_t sat_u64(double x, uint64_t hi) {
  if (x != x) return 0;
  if (x <= 0.0) return 0;
  if (x >= (double)hi) return hi;
  return (uint64_t)x;
}
uint64_t wrap_int(int t, uint64_t x) {
  switch (t) {
    case DFX_INT8: return (uint64_t)(int64_t)(int8_t)x;
    case DFX_INT16: return (uint64_t)(int64_t)(int16_t)x;
    case DFX_INT32: return (uint64_t)(int64_t)(int32_t)x;
    case DFX_UINT8: return (uint64_t)(uint8_t)x;
    case DFX_UINT16: return (uint64_t)(uint16_t)x;
    case DFX_UINT32: return (uint64_t)(uint32_t)x;
    default: return x;
  }
}
uint64_t host_cast_value(int from, int to, uint64_t v) {
  if (from == to) return v;
  auto f64b = [](double d) { uint64_t b; memcpy(&b, &d, 8); return b; };
  auto f32b = [](float f) { uint32_t b; memcpy(&b, &f, 4); return (uint64_t)b; };
  if (dtype_is_int(from)) {
    if (dtype_is_int(to)) return wrap_int(to, v);
    if (to == DFX_FLOAT64) return f64b(dtype_is_signed(from) ? (double)(int64_t)v : (double)v);
    return f32b(dtype_is_signed(from) ? (float)(int64_t)v : (float)v);
  }
  double x;
  if (from == DFX_FLOAT32) { uint32_t b = (uint32_t)v; float f; memcpy(&f, &b, 4); x = (double)f; }
  else memcpy(&x, &v, 8);
  switch (to) {
    case DFX_FLOAT32: return from == DFX_FLOAT32 ? v : f32b((float)x);
    case DFX_FLOAT64: return f64b(x);
    case DFX_INT8: return (uint64_t)sat_i64(x, -128, 127);
    case DFX_INT16: return (uint64_t)sat_i64(x, -32768, 32767);
    case DFX_INT32: return (uint64_t)sat_i64(x, -2147483648ll, 2147483647ll);
    case DFX_INT64: return (uint64_t)sat_i64(x, INT64_MIN, INT64_MAX);
    case DFX_UINT8: return sat_u64(x, 255ull);
    case DFX_UINT16: return sat_u64(x, 65535ull);
    case DFX_UINT32: return sat_u64(x, 4294967295ull);
    case DFX_UINT64: return sat_u64(x, UINT64_MAX);
    default: return 0;
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
        // `as` to the same type is the identity -- for a computed value or a literal.  A COLUMN keeps its instruction:
        // cast_column! builds a NEW array (expression.rs:246-270), whose null slots hold zero, and the grouped aggregates
        // read value(row) of their argument without a null check (aggregate.rs:561-603), so MAX(CAST(c AS its own type))
        // over a null slot sees 0, not the column's raw content (found by tests/test_gpu_fuzz.py)
        if (ta == n.dtype && (a >> 6) != OPK_COL) {
          *operand = a;
          *dtype = ta;
          return Status::OK();
        }
        if ((a >> 6) == OPK_IMM) {  // CAST(literal): folded at compile time (no per-row work at all)
          const uint64_t bits = host_cast_value(ta, n.dtype, prog_.imm[a & 63]);
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

namespace {
bool fast_factor(const DevProgram& P, uint8_t opnd, DevFastFactor* f, uint64_t* imm) {
  const int kind = opnd >> 6, idx = opnd & 63;
  *imm = 0;
  if (kind == OPK_COL) {
    if (P.col_dtype[idx] != DFX_FLOAT64) return false;
    f->kind = FF_COL;
    f->col = (uint8_t)idx;
    return true;
  }
  if (kind != OPK_REG) return false;
  const DevIns& in = P.ins[idx];
  if (in.t != DFX_FLOAT64 || in.op < DOP_ADD || in.op > DOP_MUL) return false;
  const int ka = in.a >> 6, kb = in.b >> 6;
  const bool col_imm = ka == OPK_COL && kb == OPK_IMM, imm_col = ka == OPK_IMM && kb == OPK_COL;
  if (!col_imm && !imm_col) return false;
  f->col = (uint8_t)((col_imm ? in.a : in.b) & 63);
  *imm = P.imm[(col_imm ? in.b : in.a) & 63];
  if (in.op == DOP_ADD) f->kind = FF_COL_PLUS_IMM;        // f64 addition commutes bit for bit
  else if (in.op == DOP_MUL) f->kind = FF_COL_TIMES_IMM;  // so does multiplication
  else f->kind = col_imm ? FF_COL_MINUS_IMM : FF_IMM_MINUS_COL;
  return true;
}

// left-nested product ((f0 * f1) * f2) of factors, evaluated in the tree's own association
bool fast_product(const DevProgram& P, uint8_t opnd, DevFastArg* A, uint64_t* imms) {
  DevFastFactor f;
  uint64_t imm;
  if (fast_factor(P, opnd, &f, &imm)) {
    A->nf = 1;
    A->f[0] = f;
    imms[0] = imm;
    return true;
  }
  if ((opnd >> 6) != OPK_REG) return false;
  const DevIns& in = P.ins[opnd & 63];
  if (in.op != DOP_MUL || in.t != DFX_FLOAT64) return false;
  if (!fast_product(P, in.a, A, imms)) return false;
  if (A->nf >= 3) return false;
  if (!fast_factor(P, in.b, &f, &imm)) return false;
  A->f[A->nf] = f;
  imms[A->nf] = imm;
  A->nf++;
  return true;
}

bool fast_terms(const DevProgram& P, uint8_t opnd, DevFastPlan* F) {
  if ((opnd >> 6) != OPK_REG) return false;
  const DevIns& in = P.ins[opnd & 63];
  if (in.op == DOP_AND) return fast_terms(P, in.a, F) && fast_terms(P, in.b, F);
  if (in.op > DOP_GE) return false;
  const int ka = in.a >> 6, kb = in.b >> 6;
  const bool col_imm = ka == OPK_COL && kb == OPK_IMM, imm_col = ka == OPK_IMM && kb == OPK_COL;
  if ((!col_imm && !imm_col) || F->np >= 4) return false;
  int op = in.op;
  if (imm_col) {  // literal on the left: mirror the operator
    op = op == DOP_LT ? DOP_GT : op == DOP_LE ? DOP_GE : op == DOP_GT ? DOP_LT : op == DOP_GE ? DOP_LE : op;
  }
  DevFastTerm& t = F->term[F->np];
  t.col = (uint8_t)((col_imm ? in.a : in.b) & 63);
  t.dtype = in.t;
  t.m = op == DOP_EQ || op == DOP_NE ? 2 : op == DOP_LT ? 1 : op == DOP_LE ? 3 : op == DOP_GT ? 4 : 6;
  t.inv = op == DOP_NE ? 1 : 0;
  F->term_imm[F->np] = P.imm[(col_imm ? in.b : in.a) & 63];
  F->np++;
  return true;
}
}  // namespace

void ProgramBuilder::build_fast(uint8_t pred, const uint8_t* keys, int kw, const uint8_t* args, int na,
                                DevFastPlan* F) const {
  memset(F, 0, sizeof(*F));
  if (pred != kNoOperand && !fast_terms(prog_, pred, F)) return;
  // One ordered Float64 comparison becomes the two-sided range the compile-time signatures are written for (dfx_sigs.hpp:
  // "WHERE c <op> lit AND c <op> lit"): `v < x` is `v >= -inf AND v < x`, `v > x` is `v > x AND v <= +inf`.  The added term
  // is true for every value the original term can pass (NaN fails both, as it fails the original), so the rows selected are
  // the same; lower bound first, so that the pair lands on one of the instantiated comparison forms (GT|GE, LT|LE).
  // Only where a signature can match afterwards (they read one or two columns): a run-time decoded plan of three or more
  // columns would merely pay for a second term.
  if (F->np == 1 && prog_.n_cols <= 2 && F->term[0].dtype == T_F64 && !F->term[0].inv && F->term[0].m != 2) {
    const bool upper = F->term[0].m == 1 || F->term[0].m == 3;
    const double inf = upper ? -__builtin_inf() : __builtin_inf();
    uint64_t bits;
    memcpy(&bits, &inf, 8);
    if (upper) {
      F->term[1] = F->term[0];
      F->term_imm[1] = F->term_imm[0];
      F->term[0].m = 6;  // >= -inf
      F->term_imm[0] = bits;
      F->synth = 1;
    } else {
      F->term[1] = F->term[0];
      F->term[1].m = 3;  // <= +inf
      F->term_imm[1] = bits;
      F->synth = 2;
    }
    F->np = 2;
  }
  for (int k = 0; k < kw; ++k) {
    if ((keys[k] >> 6) != OPK_COL) return;
    F->keycol[k] = (uint8_t)(keys[k] & 63);
  }
  for (int a = 0; a < na; ++a) {
    if ((args[a] >> 6) == OPK_COL) {  // plain column of any type
      F->arg[a].nf = 1;
      F->arg[a].f[0].kind = FF_COL;
      F->arg[a].f[0].col = (uint8_t)(args[a] & 63);
      continue;
    }
    if (!fast_product(prog_, args[a], &F->arg[a], F->arg_imm[a])) return;
  }
  F->valid = 1;
}

// -------------------------------------------------------------------------------------------------
// scan plans (dfx_device.hpp: DevScanPlan): the shape family of DevFastPlan as data
// -------------------------------------------------------------------------------------------------
namespace {
enum PlanClass { PC_NONE = 0, PC_F = 1, PC_I = 2, PC_U = 3 };
inline PlanClass plan_class(uint8_t t) {
  switch (t) {
    case T_F64: case T_F32: return PC_F;
    case T_I64: case T_I32: return PC_I;
    case T_U64: case T_U32: return PC_U;
    default: return PC_NONE;  // 1- and 2-byte columns, Boolean: other kernels
  }
}
inline uint64_t image_of(PlanClass c, uint64_t x) {
  if (c == PC_F) return (x >> 63) ? ~x : (x | 0x8000000000000000ull);
  if (c == PC_I) return x ^ 0x8000000000000000ull;
  return x;
}

// `value <op> literal` as an inclusive range of images (+ complement flag).  m: three-way mask of the operator (1 less, 2
// equal, 4 greater), ne: NotEq.  imm: the literal's canonical 64-bit form in the COLUMN's type.
void plan_term_range(uint8_t dtype, uint8_t m, bool ne, uint64_t imm, DevPlanTerm* T) {
  const PlanClass c = plan_class(dtype);
  T->a = c == PC_F ? 0x7FFFFFFFFFFFFFFFull : 0ull;
  T->b = c == PC_U ? 0ull : 0x8000000000000000ull;
  uint64_t LO = 0, HI = ~0ull, p, q;
  bool never = false;  // the comparison is false for every value (an unordered literal)
  if (c == PC_F) {
    double d;
    if (dtype == T_F32) {
      float f;
      const uint32_t b32 = (uint32_t)imm;
      memcpy(&f, &b32, 4);
      d = (double)f;  // exact: the kernels widen Float32 values the same way
    } else {
      memcpy(&d, &imm, 8);
    }
    const double ninf = -__builtin_inf(), pinf = __builtin_inf(), nz = -0.0, pz = 0.0;
    uint64_t bits;
    memcpy(&bits, &ninf, 8); LO = image_of(PC_F, bits);
    memcpy(&bits, &pinf, 8); HI = image_of(PC_F, bits);
    if (d != d) {
      never = true;
      p = q = 0;
    } else if (d == 0.0) {  // -0.0 == +0.0: the two images are neighbours
      memcpy(&bits, &nz, 8); p = image_of(PC_F, bits);
      memcpy(&bits, &pz, 8); q = image_of(PC_F, bits);
    } else {
      memcpy(&bits, &d, 8);
      p = q = image_of(PC_F, bits);
    }
  } else {
    p = q = image_of(c, imm);
  }
  uint64_t lo = 1, hi = 0;  // (empty)
  if (!never) {
    switch (m) {
      case 1: if (p > LO) { lo = LO; hi = p - 1; } break;        // <
      case 3: lo = LO; hi = q; break;                            // <=
      case 4: if (q < HI) { lo = q + 1; hi = HI; } break;        // >
      case 6: lo = p; hi = HI; break;                            // >=
      default: lo = p; hi = q; break;                            // == (and != as its complement)
    }
  }
  bool inv = ne;
  if (lo > hi) {  // no value passes: the complement of everything
    lo = 0;
    hi = ~0ull;
    inv = !inv;
  }
  T->lo = lo;
  T->span = hi - lo;
  T->inv = inv ? 1u : 0u;
  // arrow 0.12 bool_op compares Option<T>: None sorts below every value, the result is never null (expression.rs:171-210
  // -> array_ops; restated in run_program above).  For `null <op> literal`:
  T->if_null = ne ? 1u : (m == 1 || m == 3) ? 1u : 0u;
}
}  // namespace

bool scan_plan_shape_ok(const DevProgram& P, const DevFastPlan& F, int kw, int na, const uint8_t* val_xform) {
  if (!F.valid || P.n_cols < 1 || P.n_cols > kPlanCols || kw > kMaxKeys || na > kMaxAggs) return false;
  for (int c = 0; c < P.n_cols; ++c)
    if (plan_class(P.col_dtype[c]) == PC_NONE) return false;
  for (int k = 0; k < kw; ++k)
    if (plan_class(P.col_dtype[F.keycol[k]]) == PC_F) return false;  // (float keys are rejected long before, aggregate.rs:848-850)
  for (int a = 0; a < na; ++a) {
    if (F.arg[a].nf != 1 || F.arg[a].f[0].kind != FF_COL) return false;  // products stay with the decoded shapes
    const uint8_t t = P.col_dtype[F.arg[a].f[0].col];
    // the accumulators take the argument's own bits: a 4-byte argument is widened by the plan (Int32 as i64, Float32 as f64),
    // which only COUNT does not mind
    if (t != T_I64 && t != T_U64 && t != T_F64 && val_xform[a] != VT_COUNT_VALID) return false;
  }
  // the one-key kernels bind the key to slot 0 and the first argument to slot 1 even when they are the same column
  if (kw == 1 && na >= 1 && F.keycol[0] == F.arg[0].f[0].col && P.n_cols >= kPlanCols) return false;
  return true;
}

bool bind_scan_plan(const DevProgram& P, const DevFastPlan& F, const DevColumns& C, int kw, int na, const uint8_t* val_xform,
                    bool fixed, DevFastPlan* Fout, DevColumns* Cout) {
  if ((F.plan_mode & 3) == 0 || !scan_plan_shape_ok(P, F, kw, na, val_xform)) return false;
  if (fixed && (kw != 1 || na < 1)) return false;
  const uint8_t* ones = device_ones_block();
  if (!ones) return false;
  *Fout = F;
  DevScanPlan& S = Fout->scan;
  memset(&S, 0, sizeof(S));
  memset(Cout, 0, sizeof(*Cout));
  // plan slots: FIXED kernels find the key in slot 0 and the routed argument in slot 1; then the program's other columns
  int slot_of[kMaxCols];
  for (int c = 0; c < kMaxCols; ++c) slot_of[c] = -1;
  int src_of[kPlanCols + 2];
  int n = 0;
  if (fixed) {
    src_of[n] = F.keycol[0];
    slot_of[F.keycol[0]] = n++;
    const int ac = F.arg[0].f[0].col;
    src_of[n] = ac;  // (the key column again when the aggregate is over the key: read twice, a cache hit)
    if (slot_of[ac] < 0) slot_of[ac] = n;
    ++n;
  }
  for (int c = 0; c < P.n_cols; ++c) {
    if (slot_of[c] >= 0) continue;
    if (n >= kPlanCols) return false;
    src_of[n] = c;
    slot_of[c] = n++;
  }
  S.n_cols = n;
  int gen = 0;
  for (int sl = 0; sl < kPlanCols; ++sl) {
    const int c = sl < n ? src_of[sl] : src_of[0];  // unused slots repeat slot 0 (loads are unconditional)
    const uint8_t t = P.col_dtype[c];
    const bool w4 = t == T_I32 || t == T_U32 || t == T_F32;
    const uintptr_t base = (uintptr_t)C.c[c].values;
    if ((base & (w4 ? 3u : 7u)) != 0) return false;  // (Arrow buffers are at least value-aligned; anything else: other kernels)
    const uint32_t ext = t == T_I32 ? PX_SEXT32 : t == T_U32 ? PX_ZEXT32 : t == T_F32 ? PX_F32 : PX_NONE;
    uint32_t vbit0 = 0;
    const uint8_t* vb = ones;
    if (C.c[c].validity) {
      if (C.c[c].bit_offset < 0) return false;
      vb = C.c[c].validity + (C.c[c].bit_offset >> 3);
      vbit0 = (uint32_t)(C.c[c].bit_offset & 7);
      if (sl < n) gen |= 2;
    }
    if (w4 && sl < n) gen |= 1;
    const uint32_t meta = plan_col_meta(w4 ? 2u : 3u, w4 ? (uint32_t)((base & 7u) >> 2) : 0u, ext, vbit0, C.c[c].validity != nullptr);
    Cout->c[sl].values = (const void*)(base & ~(uintptr_t)7);
    Cout->c[sl].validity = vb;
    Cout->c[sl].bit_offset = (int64_t)meta;
    S.col_meta[sl] = meta;
  }
  // terms (the open side build_fast added to a one-sided range is left out: see DevFastPlan::synth)
  int np = 0;
  for (int i = 0; i < F.np; ++i) {
    if ((F.synth >> i) & 1) continue;
    if (np >= kPlanTerms) return false;
    DevPlanTerm& T = S.term[np++];
    T.col = (uint32_t)slot_of[F.term[i].col];
    plan_term_range(F.term[i].dtype, F.term[i].m, F.term[i].inv != 0, F.term_imm[i], &T);
  }
  for (int i = np; i < kPlanTerms; ++i) {  // neutral: every value passes
    S.term[i].col = 0;
    S.term[i].lo = 0;
    S.term[i].span = ~0ull;
    S.term[i].if_null = 1;
  }
  S.np = np;
  for (int k = 0; k < kw; ++k) S.keyslot[k] = (uint8_t)slot_of[F.keycol[k]];
  for (int a = 0; a < na; ++a) S.argslot[a] = (uint8_t)slot_of[F.arg[a].f[0].col];
  if (fixed) S.argslot[0] = 1;
  // COUNT(x) looks at x's validity only when no Filter sits below: fn filter emits all-valid arrays (filter.rs:83-92), so
  // under an absorbed predicate every surviving slot counts (and every other aggregate reads value(row) regardless,
  // aggregate.rs:561-603)
  S.count_valid = F.np == 0 ? 1 : 0;
  // the Int32 / UInt32 key of a one-key kernel with nothing else to widen: its own flavour (a real 4-byte load of the key)
  if (fixed && (gen & 1)) {
    bool only_key = plan_class(P.col_dtype[src_of[0]]) != PC_F && (P.col_dtype[src_of[0]] == T_I32 || P.col_dtype[src_of[0]] == T_U32);
    for (int sl = 1; sl < n && only_key; ++sl) {
      const uint8_t t = P.col_dtype[src_of[sl]];
      if (t == T_I32 || t == T_U32 || t == T_F32) only_key = false;
    }
    if (only_key) {
      gen = (gen & ~1) | 4;
      for (int sl = n; sl < kPlanCols; ++sl) {  // unused slots are read 8 bytes wide here: repeat slot 1, never the 4-byte key
        Cout->c[sl] = Cout->c[1];
        S.col_meta[sl] = S.col_meta[1];
      }
    }
  }
  S.gen = gen;
  S.valid = 1;
  return true;
}

Status ProgramBuilder::bind(const DeviceBatch& batch, DevProgram* prog, DevColumns* cols) const {
  *prog = prog_;
  memset(cols, 0, sizeof(*cols));
  prog->has_nulls = 0;
  prog->wide8 = 0;
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
  prog->wide8 = !prog->has_nulls && !cols_.empty();
  for (size_t i = 0; i < cols_.size(); ++i) {
    const uint8_t t = prog_.col_dtype[i];
    if (t != T_I64 && t != T_U64 && t != T_F64) prog->wide8 = 0;
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
  return c_abi_guard(err, errlen, [&]() -> int32_t {
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
  });
}

int32_t dfx_compile_expr(const dfx_expr_node* nodes, int32_t n_nodes, int32_t root,
                         const struct ArrowSchema* input_schema, dfx_runtime_expr** out, char* err,
                         size_t errlen) {
  return c_abi_guard(err, errlen, [&]() -> int32_t {
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
    else if (!strcasecmp(nm, "avg")) f = AGG_AVG;  // deviation D7: typed by the planner (sqlplanner.rs:309-322), no executor in the reference
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
  });
}

int32_t dfx_debug_plan_term(int32_t dtype, int32_t op, uint64_t literal, uint64_t value, int32_t is_null) {
  if (op < 0 || op > 5) return -1;
  const uint8_t t = (uint8_t)dtype;
  if (t != T_I32 && t != T_U32 && t != T_F32 && t != T_I64 && t != T_U64 && t != T_F64) return -1;
  static const uint8_t three_way[6] = {2, 2, 1, 3, 4, 6};  // Eq NotEq Lt LtEq Gt GtEq (fast_terms)
  DevPlanTerm T;
  memset(&T, 0, sizeof(T));
  plan_term_range(t, three_way[op], op == DFX_OP_NOT_EQ, literal, &T);
  if (is_null) return (int32_t)T.if_null;
  uint64_t x = value;  // what PlanPolicy::eval leaves in the row's register: the value widened to 64 bits
  if (t == T_F32) {
    float f;
    const uint32_t b32 = (uint32_t)value;
    memcpy(&f, &b32, 4);
    const double d = (double)f;
    memcpy(&x, &d, 8);
  }
  // PlanPolicy::pass, word for word
  const uint32_t hi = (uint32_t)(x >> 32);
  const uint32_t neg = (uint32_t)((int32_t)hi >> 31) & (T.a != 0ull ? 0xFFFFFFFFu : 0u);
  const uint32_t img_hi = hi ^ (neg & 0x7FFFFFFFu) ^ ((T.b >> 63) ? 0x80000000u : 0u);
  const uint64_t img = ((uint64_t)img_hi << 32) | ((uint32_t)x ^ neg);
  const uint64_t nlo = 0ull - T.lo;
  return (int32_t)(((img + nlo) <= T.span ? 1u : 0u) ^ (T.inv ? 1u : 0u));
}

const char* dfx_runtime_expr_name(const dfx_runtime_expr* e) { return e ? e->name.c_str() : ""; }
int32_t dfx_runtime_expr_type(const dfx_runtime_expr* e) { return e ? e->dtype : DFX_TYPE_NONE; }
int32_t dfx_runtime_expr_is_aggregate(const dfx_runtime_expr* e) { return (e && e->is_aggregate) ? 1 : 0; }
void dfx_runtime_expr_free(dfx_runtime_expr* e) { delete e; }

}  // extern "C"

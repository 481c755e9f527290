// dfx_sigs.hpp -- compile-time shape signatures of the hot queries (template instantiation over a
// small set of shapes, chosen at operator-creation time; no run-time JIT).
//
// A signature fixes everything about a scan that does not depend on the data: how many 8-byte
// columns it reads, the predicate terms (type + column slot), the key columns, and for each
// aggregate its argument column, accumulator kind and operand transform.  Only the comparison
// operators and the literals stay run-time values.  StaticPolicy<.., SIG> (dfx_kernels_inl.hpp)
// turns a signature into straight-line code with no dtype / kind switches at all; a plan that
// matches no signature runs FastPolicy (same shape family, decoded at run time) or the interpreter.
//
// Column slots follow ProgramBuilder's first-use order: predicate columns, then keys, then arguments.
#pragma once
#include "dfx_device.hpp"

namespace dfx {

#define DFX_SIG_FN(name, ...)                              \
  static constexpr uint8_t name(int i) {                   \
    constexpr uint8_t t[8] = {__VA_ARGS__};                \
    return t[i];                                           \
  }

// signatures whose aggregate arguments are all plain columns: no factor tables
#define DFX_SIG_NO_PRODUCTS                                   \
  static constexpr uint8_t nf(int) { return 1; }              \
  static constexpr uint8_t fk(int, int) { return FF_COL; }    \
  static constexpr uint8_t fc(int, int) { return 0; }

// WHERE c0 <op> lit AND c0 <op> lit                                  (FilterRelation, config 2)
struct SigPred2F64 {
  static constexpr int NCOL = 1, NP = 2, KW = 0, NA = 0;
  DFX_SIG_FN(term_cls, T_F64, T_F64) DFX_SIG_FN(term_col, 0, 0) DFX_SIG_FN(key_col, 0)
  DFX_SIG_FN(arg_dyn, 0) DFX_SIG_FN(arg_col, 0) DFX_SIG_FN(acc, 0) DFX_SIG_FN(xf, 0)
  DFX_SIG_NO_PRODUCTS
};
// SELECT COUNT(c0) WHERE c0 <op> lit AND c0 <op> lit                 (config 2 through the aggregate)
struct SigCountPred2F64 {
  static constexpr int NCOL = 1, NP = 2, KW = 0, NA = 1;
  DFX_SIG_FN(term_cls, T_F64, T_F64) DFX_SIG_FN(term_col, 0, 0) DFX_SIG_FN(key_col, 0)
  DFX_SIG_FN(arg_dyn, 0) DFX_SIG_FN(arg_col, 0) DFX_SIG_FN(acc, ACC_ADD_U64) DFX_SIG_FN(xf, VT_COUNT_VALID)
  DFX_SIG_NO_PRODUCTS
};
// SELECT SUM(c0), COUNT(c0) WHERE c0 <op> lit AND c0 <op> lit
struct SigSumCountPred2F64 {
  static constexpr int NCOL = 1, NP = 2, KW = 0, NA = 2;
  DFX_SIG_FN(term_cls, T_F64, T_F64) DFX_SIG_FN(term_col, 0, 0) DFX_SIG_FN(key_col, 0)
  DFX_SIG_FN(arg_dyn, 0, 0) DFX_SIG_FN(arg_col, 0, 0) DFX_SIG_FN(acc, ACC_ADD_F64, ACC_ADD_U64)
  DFX_SIG_FN(xf, VT_RAW, VT_COUNT_VALID)
  DFX_SIG_NO_PRODUCTS
};
// SELECT c1, SUM(c0) WHERE c0 <op> lit AND c0 <op> lit GROUP BY c1   (the headline: filter + GROUP-BY-SUM)
struct SigKeySumPred2F64 {
  static constexpr int NCOL = 2, NP = 2, KW = 1, NA = 1;
  DFX_SIG_FN(term_cls, T_F64, T_F64) DFX_SIG_FN(term_col, 0, 0) DFX_SIG_FN(key_col, 1)
  DFX_SIG_FN(arg_dyn, 0) DFX_SIG_FN(arg_col, 0) DFX_SIG_FN(acc, ACC_ADD_F64) DFX_SIG_FN(xf, VT_RAW)
  DFX_SIG_NO_PRODUCTS
};
// SELECT c1, SUM(c0 <+ - *> lit) WHERE c0 <op> lit AND c0 <op> lit GROUP BY c1   (the headline with a scaled / shifted
// argument: the arithmetic operator is the plan's, read once per row group -- FF_RT -- everything else as above)
struct SigKeyAffSumPred2F64 {
  static constexpr int NCOL = 2, NP = 2, KW = 1, NA = 1;
  DFX_SIG_FN(term_cls, T_F64, T_F64) DFX_SIG_FN(term_col, 0, 0) DFX_SIG_FN(key_col, 1)
  DFX_SIG_FN(arg_dyn, 1) DFX_SIG_FN(arg_col, 0) DFX_SIG_FN(acc, ACC_ADD_F64) DFX_SIG_FN(xf, VT_RAW)
  static constexpr uint8_t nf(int) { return 1; }
  static constexpr uint8_t fk(int, int) { return FF_RT; }
  static constexpr uint8_t fc(int, int) { return 0; }
};
// SELECT c0, SUM(c1) GROUP BY c0                                      (config 3)
struct SigKeySum {
  static constexpr int NCOL = 2, NP = 0, KW = 1, NA = 1;
  DFX_SIG_FN(term_cls, 0) DFX_SIG_FN(term_col, 0) DFX_SIG_FN(key_col, 0)
  DFX_SIG_FN(arg_dyn, 0) DFX_SIG_FN(arg_col, 1) DFX_SIG_FN(acc, ACC_ADD_F64) DFX_SIG_FN(xf, VT_RAW)
  DFX_SIG_NO_PRODUCTS
};
// TPC-H-Q1 shape (config 5): WHERE c0 <op> lit AND c1 <op> lit GROUP BY c2, c3;
// SUM(c4), SUM(c5), SUM(c5 * (lit - c1)), SUM(c5 * (lit - c1) * (c6 + lit))
//   = SUM(qty), SUM(price), SUM(price * (1 - disc)), SUM(price * (1 - disc) * (1 + tax)) with the predicate on
//   (ship, disc): the product factors are part of the signature, only the literals are run-time values
struct SigQ1 {
  static constexpr int NCOL = 7, NP = 2, KW = 2, NA = 4;
  DFX_SIG_FN(term_cls, T_F64, T_F64) DFX_SIG_FN(term_col, 0, 1) DFX_SIG_FN(key_col, 2, 3)
  DFX_SIG_FN(arg_dyn, 0, 0, 1, 1) DFX_SIG_FN(arg_col, 4, 5, 0, 0)
  DFX_SIG_FN(acc, ACC_ADD_F64, ACC_ADD_F64, ACC_ADD_F64, ACC_ADD_F64) DFX_SIG_FN(xf, VT_RAW, VT_RAW, VT_RAW, VT_RAW)
  DFX_SIG_FN(nf, 1, 1, 2, 3)
  static constexpr uint8_t fk(int a, int j) {
    constexpr uint8_t t[4][3] = {{FF_COL, 0, 0}, {FF_COL, 0, 0}, {FF_COL, FF_IMM_MINUS_COL, 0},
                                 {FF_COL, FF_IMM_MINUS_COL, FF_COL_PLUS_IMM}};
    return t[a][j];
  }
  static constexpr uint8_t fc(int a, int j) {
    constexpr uint8_t t[4][3] = {{4, 0, 0}, {5, 0, 0}, {5, 1, 0}, {5, 1, 6}};
    return t[a][j];
  }
};

// host: does a bound program + fast plan + aggregate description match SIG?
template <typename SIG>
inline bool sig_matches(const DevProgram& P, const DevFastPlan& F, int kw, int na, const uint8_t* acc_kind,
                        const uint8_t* val_xform) {
  if (!F.valid || P.has_nulls || P.n_cols != SIG::NCOL || F.np != SIG::NP || kw != SIG::KW || na != SIG::NA) return false;
  for (int c = 0; c < SIG::NCOL; ++c)
    if (P.col_dtype[c] != T_I64 && P.col_dtype[c] != T_U64 && P.col_dtype[c] != T_F64) return false;
  for (int i = 0; i < SIG::NP; ++i)
    if (F.term[i].dtype != SIG::term_cls(i) || F.term[i].col != SIG::term_col(i)) return false;
  for (int k = 0; k < SIG::KW; ++k)
    if (F.keycol[k] != SIG::key_col(k)) return false;
  for (int a = 0; a < SIG::NA; ++a) {
    if (acc_kind[a] != SIG::acc(a) || val_xform[a] != SIG::xf(a)) return false;
    const bool plain = F.arg[a].nf == 1 && F.arg[a].f[0].kind == FF_COL;
    if (SIG::arg_dyn(a)) {  // a product: the factors (kind + column) are part of the signature
      if (plain || F.arg[a].nf != SIG::nf(a)) return false;
      for (int j = 0; j < SIG::nf(a); ++j) {
        const bool kind_ok = SIG::fk(a, j) == FF_RT ? F.arg[a].f[j].kind != FF_COL && F.arg[a].f[j].kind != FF_NONE
                                                    : F.arg[a].f[j].kind == SIG::fk(a, j);
        if (!kind_ok || F.arg[a].f[j].col != SIG::fc(a, j)) return false;
      }
    } else if (!plain || F.arg[a].f[0].col != SIG::arg_col(a)) {
      return false;
    }
  }
  return true;
}

}  // namespace dfx

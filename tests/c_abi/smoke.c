/* smoke.c -- a plain C consumer of include/dfx.h (what a cgo / Rust-FFI shim would do), no Python, no torch.
 *
 *   SELECT k, SUM(v) FROM t WHERE v > 204.8 AND v < 409.6 GROUP BY k      (t synthesised in HBM)
 *
 * built as  gcc -std=c11 -I include tests/c_abi/smoke.c -L datafusion_archive_amd/lib -ldfx_hip -o ...
 * Prints "OK groups=<g> sum=<s> rows_passing=<c>" (checked by tests/test_gpu_parity.py against the oracle)
 * or "ERR <status> <message>". */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dfx.h"

#define CHECK(call)                                                     \
  do {                                                                  \
    int32_t st_ = (call);                                               \
    if (st_ != DFX_OK) {                                                \
      printf("ERR %d %s (%s)\n", (int)st_, err, #call);                 \
      return 1;                                                         \
    }                                                                   \
  } while (0)

static dfx_expr_node node(int kind) {
  dfx_expr_node n;
  memset(&n, 0, sizeof(n));
  n.kind = kind;
  n.left = n.right = n.column = -1;
  return n;
}

int main(int argc, char** argv) {
  char err[512] = {0};
  const int64_t n_rows = argc > 1 ? atoll(argv[1]) : 1000000;
  const double groups = argc > 2 ? atof(argv[2]) : 5000.0;
  CHECK(dfx_init(0, err, sizeof(err)));

  dfx_synth_column cols[2] = {{"k", DFX_SYNTH_I64_UNIFORM, 0, groups, 0.0}, {"v", DFX_SYNTH_F64_EXACT, 1, 0.0, 0.0}};
  dfx_table* table = NULL;
  CHECK(dfx_table_synth(cols, 2, 0xDF02, 0, n_rows, &table, err, sizeof(err)));

  struct ArrowArrayStream scan, filtered, agg;
  CHECK(dfx_table_scan_new(table, 1 << 18, &scan, err, sizeof(err)));
  struct ArrowSchema schema;
  if (scan.get_schema(&scan, &schema) != 0) { printf("ERR get_schema\n"); return 1; }

  /* (v > 204.8) AND (v < 409.6)  -- nodes in any order, children by index */
  dfx_expr_node p[7];
  p[0] = node(DFX_EXPR_COLUMN);  p[0].column = 1;
  p[1] = node(DFX_EXPR_LITERAL); p[1].dtype = DFX_FLOAT64; p[1].lit.f64 = 204.8;
  p[2] = node(DFX_EXPR_BINARY);  p[2].op = DFX_OP_GT; p[2].left = 0; p[2].right = 1;
  p[3] = node(DFX_EXPR_LITERAL); p[3].dtype = DFX_FLOAT64; p[3].lit.f64 = 409.6;
  p[4] = node(DFX_EXPR_BINARY);  p[4].op = DFX_OP_LT; p[4].left = 0; p[4].right = 3;
  p[5] = node(DFX_EXPR_BINARY);  p[5].op = DFX_OP_AND; p[5].left = 2; p[5].right = 4;
  dfx_runtime_expr *pred = NULL, *key = NULL, *sum = NULL, *cnt = NULL;
  CHECK(dfx_compile_scalar_expr(p, 6, 5, &schema, &pred, err, sizeof(err)));
  dfx_expr_node k = node(DFX_EXPR_COLUMN); k.column = 0;
  CHECK(dfx_compile_scalar_expr(&k, 1, 0, &schema, &key, err, sizeof(err)));
  dfx_expr_node a[2];
  a[0] = node(DFX_EXPR_COLUMN); a[0].column = 1;
  a[1] = node(DFX_EXPR_AGGREGATE_FUNCTION); a[1].dtype = DFX_FLOAT64; a[1].left = 0; a[1].n_args = 1; a[1].name = "SUM";
  CHECK(dfx_compile_expr(a, 2, 1, &schema, &sum, err, sizeof(err)));
  a[1].dtype = DFX_UINT64; a[1].name = "COUNT";
  CHECK(dfx_compile_expr(a, 2, 1, &schema, &cnt, err, sizeof(err)));
  if (strcmp(dfx_runtime_expr_name(sum), "SUM") != 0) { printf("ERR name %s\n", dfx_runtime_expr_name(sum)); return 1; }

  CHECK(dfx_filter_relation_new(&scan, pred, &schema, &filtered, err, sizeof(err)));
  const dfx_runtime_expr* groups_[1] = {key};
  const dfx_runtime_expr* aggs_[2] = {sum, cnt};
  CHECK(dfx_aggregate_relation_new(NULL, &filtered, groups_, 1, aggs_, 2, &agg, err, sizeof(err)));

  struct ArrowArray out;
  if (agg.get_next(&agg, &out) != 0) { printf("ERR %s\n", agg.get_last_error(&agg)); return 1; }
  if (out.release == NULL) { printf("ERR no batch\n"); return 1; }
  if (out.n_children != 3) { printf("ERR columns %lld\n", (long long)out.n_children); return 1; }
  const double* sums = (const double*)out.children[1]->buffers[1];
  const uint64_t* counts = (const uint64_t*)out.children[2]->buffers[1];
  double total = 0;
  unsigned long long passing = 0;
  for (int64_t i = 0; i < out.length; ++i) {  /* values are multiples of 2^-10: the sum is exact in any order */
    total += sums[out.children[1]->offset + i];
    passing += counts[out.children[2]->offset + i];
  }
  const long long n_groups = (long long)out.length;
  out.release(&out);
  struct ArrowArray end;
  if (agg.get_next(&agg, &end) != 0 || end.release != NULL) { printf("ERR stream did not end\n"); return 1; }
  agg.release(&agg);
  schema.release(&schema);
  dfx_runtime_expr_free(pred); dfx_runtime_expr_free(key); dfx_runtime_expr_free(sum); dfx_runtime_expr_free(cnt);
  dfx_table_free(table);
  printf("OK groups=%lld sum=%.17g rows_passing=%llu\n", n_groups, total, passing);
  return 0;
}

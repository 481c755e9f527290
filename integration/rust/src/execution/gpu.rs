// gpu.rs -- DataFusion 0.6.0 operators on an MI355X through libdfx_hip.so (include/dfx.h).
//
// NOT COMPILED IN THE REPOSITORY THAT SHIPS IT (no Rust toolchain there); written against arrow 0.12 as the rest of
// src/execution uses it.  See integration/rust/README.md.  Everything below the `extern "C"` block is ordinary safe-ish
// glue: Expr -> dfx_expr_node[], Schema / RecordBatch <-> Arrow C Data structs, Relation <-> ArrowArrayStream.
//
//   context.rs:132,155,174,180   compile_scalar_expr / compile_expr      -> gpu::compile_scalar_expr / compile_expr
//   context.rs:133-137           FilterRelation::new                      -> GpuRelation::filter
//   context.rs:158               ProjectRelation::new                     -> GpuRelation::project
//   context.rs:184-189           AggregateRelation::new                   -> GpuRelation::aggregate
//   datasource.rs:39-43          CsvDataSource::new                       -> GpuRelation::csv
//   context.rs:113,194           Sort / Limit (unimplemented!())          -> GpuRelation::sort / limit
//   datasource.rs:27-30          trait DataSource (in-memory, reusable)   -> GpuTable::load / scan

use std::cell::RefCell;
use std::ffi::{CStr, CString};
use std::io;
use std::mem;
use std::os::raw::{c_char, c_void};
use std::ptr;
use std::rc::Rc;
use std::sync::Arc;

use arrow::array::{Array, ArrayData, ArrayRef, BinaryArray, BooleanArray, PrimitiveArray};
use arrow::buffer::Buffer;
use arrow::datatypes::*;
use arrow::error::ArrowError;
use arrow::record_batch::RecordBatch;

use super::error::{ExecutionError, Result};
use super::datasource::DataSource;
use super::relation::{DataSourceRelation, Relation};
use crate::logicalplan::{Expr, Operator, ScalarValue};

// ---- raw bindings: exactly include/dfx.h -----------------------------------------------------------------
#[repr(C)]
pub struct ArrowSchema {
    format: *const c_char,
    name: *const c_char,
    metadata: *const c_char,
    flags: i64,
    n_children: i64,
    children: *mut *mut ArrowSchema,
    dictionary: *mut ArrowSchema,
    release: Option<unsafe extern "C" fn(*mut ArrowSchema)>,
    private_data: *mut c_void,
}
#[repr(C)]
pub struct ArrowArray {
    length: i64,
    null_count: i64,
    offset: i64,
    n_buffers: i64,
    n_children: i64,
    buffers: *mut *const c_void,
    children: *mut *mut ArrowArray,
    dictionary: *mut ArrowArray,
    release: Option<unsafe extern "C" fn(*mut ArrowArray)>,
    private_data: *mut c_void,
}
#[repr(C)]
pub struct ArrowArrayStream {
    get_schema: Option<unsafe extern "C" fn(*mut ArrowArrayStream, *mut ArrowSchema) -> i32>,
    get_next: Option<unsafe extern "C" fn(*mut ArrowArrayStream, *mut ArrowArray) -> i32>,
    get_last_error: Option<unsafe extern "C" fn(*mut ArrowArrayStream) -> *const c_char>,
    release: Option<unsafe extern "C" fn(*mut ArrowArrayStream)>,
    private_data: *mut c_void,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct DfxExprNode {
    kind: i32,   // dfx_expr_kind == the order of logicalplan::Expr's variants
    op: i32,     // dfx_operator == the order of logicalplan::Operator's variants
    dtype: i32,  // dfx_dtype
    left: i32,
    right: i32,
    column: i32,
    n_args: i32,
    reserved: i32,
    lit: u64, // union { i64, u64, f64, f32 }: the bits
    name: *const c_char,
}
pub enum DfxRuntimeExpr {}
#[repr(C)]
pub enum DfxTable {}
#[repr(C)]
pub enum DfxComm {}

/// One per-operator option (include/dfx.h: dfx_option): the operator starts from the process defaults and applies these on top.
#[repr(C)]
pub struct DfxOption {
    pub key: *const c_char,
    pub value: i64,
}

// (the attribute must sit directly on the extern block: it names the library these symbols come from)
#[link(name = "dfx_hip")]
extern "C" {
    fn dfx_init(device: i32, err: *mut c_char, errlen: usize) -> i32;
    fn dfx_compile_scalar_expr(nodes: *const DfxExprNode, n: i32, root: i32, schema: *const ArrowSchema,
                               out: *mut *mut DfxRuntimeExpr, err: *mut c_char, errlen: usize) -> i32;
    fn dfx_compile_expr(nodes: *const DfxExprNode, n: i32, root: i32, schema: *const ArrowSchema,
                        out: *mut *mut DfxRuntimeExpr, err: *mut c_char, errlen: usize) -> i32;
    fn dfx_runtime_expr_name(e: *const DfxRuntimeExpr) -> *const c_char;
    fn dfx_runtime_expr_type(e: *const DfxRuntimeExpr) -> i32;
    fn dfx_runtime_expr_free(e: *mut DfxRuntimeExpr);
    fn dfx_filter_relation_new(input: *mut ArrowArrayStream, expr: *const DfxRuntimeExpr, schema: *const ArrowSchema,
                               out: *mut ArrowArrayStream, err: *mut c_char, errlen: usize) -> i32;
    fn dfx_project_relation_new(input: *mut ArrowArrayStream, exprs: *const *const DfxRuntimeExpr, n: i32,
                                schema: *const ArrowSchema, out: *mut ArrowArrayStream, err: *mut c_char, errlen: usize) -> i32;
    fn dfx_aggregate_relation_new(schema: *const ArrowSchema, input: *mut ArrowArrayStream,
                                  group: *const *const DfxRuntimeExpr, n_group: i32,
                                  aggr: *const *const DfxRuntimeExpr, n_aggr: i32,
                                  out: *mut ArrowArrayStream, err: *mut c_char, errlen: usize) -> i32;
    fn dfx_filter_relation_new_with_options(input: *mut ArrowArrayStream, expr: *const DfxRuntimeExpr, schema: *const ArrowSchema,
                                            options: *const DfxOption, n_options: i32,
                                            out: *mut ArrowArrayStream, err: *mut c_char, errlen: usize) -> i32;
    fn dfx_aggregate_relation_new_with_options(schema: *const ArrowSchema, input: *mut ArrowArrayStream,
                                               group: *const *const DfxRuntimeExpr, n_group: i32,
                                               aggr: *const *const DfxRuntimeExpr, n_aggr: i32,
                                               options: *const DfxOption, n_options: i32,
                                               out: *mut ArrowArrayStream, err: *mut c_char, errlen: usize) -> i32;
    fn dfx_csv_datasource_new(filename: *const c_char, schema: *const ArrowSchema, batch_size: i64,
                              out: *mut ArrowArrayStream, err: *mut c_char, errlen: usize) -> i32;
    fn dfx_sort_relation_new(input: *mut ArrowArrayStream, exprs: *const *const DfxRuntimeExpr, ascending: *const i32,
                             n: i32, schema: *const ArrowSchema, out: *mut ArrowArrayStream, err: *mut c_char, errlen: usize) -> i32;
    fn dfx_limit_relation_new(input: *mut ArrowArrayStream, limit: i64, schema: *const ArrowSchema,
                              out: *mut ArrowArrayStream, err: *mut c_char, errlen: usize) -> i32;
    fn dfx_table_from_stream(input: *mut ArrowArrayStream, out: *mut *mut DfxTable, err: *mut c_char, errlen: usize) -> i32;
    fn dfx_table_scan_new(t: *const DfxTable, batch_rows: i64, out: *mut ArrowArrayStream, err: *mut c_char, errlen: usize) -> i32;
    fn dfx_table_scan_range_new(t: *const DfxTable, row_begin: i64, n_rows: i64, batch_rows: i64, out: *mut ArrowArrayStream,
                                err: *mut c_char, errlen: usize) -> i32;
    fn dfx_table_num_rows(t: *const DfxTable) -> i64;
    fn dfx_table_free(t: *mut DfxTable);
    fn dfx_synchronize(err: *mut c_char, errlen: usize) -> i32;
    fn dfx_set_option(key: *const c_char, value: i64) -> i32;
    fn dfx_relation_explain(stream: *mut ArrowArrayStream, buf: *mut c_char, buflen: usize) -> i64;
    // multi-GPU GROUP BY (one process per GPU): the three device steps for a host that brings its own collective ...
    fn dfx_aggregate_partial_build(agg: *mut ArrowArrayStream, world: i32, n_words: *mut i32, counts: *mut i64,
                                   err: *mut c_char, errlen: usize) -> i32;
    fn dfx_aggregate_partial_export(agg: *mut ArrowArrayStream, dst_device: *mut c_void, dst_words: i64,
                                    err: *mut c_char, errlen: usize) -> i32;
    fn dfx_aggregate_partial_import(agg: *mut ArrowArrayStream, src_device: *const c_void, counts: *const i64, n_buckets: i32,
                                    err: *mut c_char, errlen: usize) -> i32;
    // ... and the whole exchange inside the library over RCCL
    fn dfx_comm_unique_id(id: *mut u8, err: *mut c_char, errlen: usize) -> i32;
    fn dfx_comm_init(id: *const u8, world: i32, rank: i32, out: *mut *mut DfxComm, err: *mut c_char, errlen: usize) -> i32;
    fn dfx_comm_destroy(comm: *mut DfxComm);
    fn dfx_comm_ranks(comm: *const DfxComm) -> i32;
    fn dfx_aggregate_exchange(agg: *mut ArrowArrayStream, comm: *mut DfxComm, stats: *mut i64, err: *mut c_char, errlen: usize) -> i32;
}

const ERRLEN: usize = 1024;

// ---- status -> ExecutionError (error.rs:26-36; dfx_status uses the same numbering) ------------------------
fn check(code: i32, err: &[c_char; ERRLEN]) -> Result<()> {
    if code == 0 {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(err.as_ptr()) }.to_string_lossy().into_owned();
    Err(match code {
        1 => ExecutionError::IoError(io::Error::new(io::ErrorKind::Other, msg)),
        3 => ExecutionError::General(msg),
        4 => ExecutionError::InvalidColumn(msg),
        5 => ExecutionError::NotImplemented(msg),
        6 => ExecutionError::InternalError(msg),
        7 => ExecutionError::ArrowError(ArrowError::ComputeError(msg)),
        _ => ExecutionError::ExecutionError(msg),
    })
}

pub fn init(device: i32) -> Result<()> {
    let mut err = [0 as c_char; ERRLEN];
    check(unsafe { dfx_init(device, err.as_mut_ptr(), ERRLEN) }, &err)
}

// ---- DataType <-> dfx_dtype / Arrow format strings ---------------------------------------------------------
fn dtype_code(t: &DataType) -> i32 {
    match t {
        DataType::Boolean => 1, DataType::Int8 => 2, DataType::Int16 => 3, DataType::Int32 => 4, DataType::Int64 => 5,
        DataType::UInt8 => 6, DataType::UInt16 => 7, DataType::UInt32 => 8, DataType::UInt64 => 9,
        DataType::Float32 => 10, DataType::Float64 => 11, DataType::Utf8 => 12, _ => 0,
    }
}
fn dtype_from_code(c: i32) -> DataType {
    match c {
        1 => DataType::Boolean, 2 => DataType::Int8, 3 => DataType::Int16, 4 => DataType::Int32, 5 => DataType::Int64,
        6 => DataType::UInt8, 7 => DataType::UInt16, 8 => DataType::UInt32, 9 => DataType::UInt64,
        10 => DataType::Float32, 11 => DataType::Float64, _ => DataType::Utf8,
    }
}
fn format_of(t: &DataType) -> &'static [u8] {
    match t {
        DataType::Boolean => b"b\0", DataType::Int8 => b"c\0", DataType::UInt8 => b"C\0", DataType::Int16 => b"s\0",
        DataType::UInt16 => b"S\0", DataType::Int32 => b"i\0", DataType::UInt32 => b"I\0", DataType::Int64 => b"l\0",
        DataType::UInt64 => b"L\0", DataType::Float32 => b"f\0", DataType::Float64 => b"g\0", _ => b"u\0",
    }
}
fn dtype_of_format(f: &[u8]) -> DataType {
    match f.first() {
        Some(b'b') => DataType::Boolean, Some(b'c') => DataType::Int8, Some(b'C') => DataType::UInt8,
        Some(b's') => DataType::Int16, Some(b'S') => DataType::UInt16, Some(b'i') => DataType::Int32,
        Some(b'I') => DataType::UInt32, Some(b'l') => DataType::Int64, Some(b'L') => DataType::UInt64,
        Some(b'f') => DataType::Float32, Some(b'g') => DataType::Float64, _ => DataType::Utf8,
    }
}

// ---- Schema -> ArrowSchema (owned by a Rust box; `release` frees it) ----------------------------------------
struct SchemaHolder {
    names: Vec<CString>,
    children: Vec<ArrowSchema>,
    child_ptrs: Vec<*mut ArrowSchema>,
}
unsafe extern "C" fn release_schema(s: *mut ArrowSchema) {
    if !(*s).private_data.is_null() {
        drop(Box::from_raw((*s).private_data as *mut SchemaHolder));
    }
    (*s).release = None;
}
unsafe extern "C" fn release_child_schema(s: *mut ArrowSchema) {
    (*s).release = None; // owned by the parent's holder
}
/// "+s" struct schema with one child per field
pub fn export_schema(schema: &Schema) -> ArrowSchema {
    let mut h = Box::new(SchemaHolder { names: vec![], children: vec![], child_ptrs: vec![] });
    for f in schema.fields() {
        h.names.push(CString::new(f.name().as_str()).unwrap());
    }
    for (i, f) in schema.fields().iter().enumerate() {
        h.children.push(ArrowSchema {
            format: format_of(f.data_type()).as_ptr() as *const c_char,
            name: h.names[i].as_ptr(),
            metadata: ptr::null(),
            flags: if f.is_nullable() { 2 } else { 0 },
            n_children: 0,
            children: ptr::null_mut(),
            dictionary: ptr::null_mut(),
            release: Some(release_child_schema),
            private_data: ptr::null_mut(),
        });
    }
    for c in h.children.iter_mut() {
        h.child_ptrs.push(c as *mut ArrowSchema);
    }
    ArrowSchema {
        format: b"+s\0".as_ptr() as *const c_char,
        name: b"\0".as_ptr() as *const c_char,
        metadata: ptr::null(),
        flags: 0,
        n_children: h.child_ptrs.len() as i64,
        children: h.child_ptrs.as_mut_ptr(),
        dictionary: ptr::null_mut(),
        release: Some(release_schema),
        private_data: Box::into_raw(h) as *mut c_void,
    }
}
unsafe fn import_schema(s: &ArrowSchema) -> Schema {
    let mut fields = vec![];
    for i in 0..s.n_children as isize {
        let c = &**s.children.offset(i);
        let name = CStr::from_ptr(c.name).to_string_lossy().into_owned();
        let fmt = CStr::from_ptr(c.format).to_bytes();
        fields.push(Field::new(&name, dtype_of_format(fmt), c.flags & 2 != 0));
    }
    Schema::new(fields)
}

// ---- Expr -> dfx_expr_node[] (logicalplan.rs:136-208), post order; returns the index of the node pushed ------
fn operator_code(op: &Operator) -> i32 {
    match op {
        Operator::Eq => 0, Operator::NotEq => 1, Operator::Lt => 2, Operator::LtEq => 3, Operator::Gt => 4,
        Operator::GtEq => 5, Operator::Plus => 6, Operator::Minus => 7, Operator::Multiply => 8, Operator::Divide => 9,
        Operator::Modulus => 10, Operator::And => 11, Operator::Or => 12, Operator::Not => 13, Operator::Like => 14,
        Operator::NotLike => 15,
    }
}
fn empty_node() -> DfxExprNode {
    DfxExprNode { kind: 0, op: 0, dtype: 0, left: -1, right: -1, column: -1, n_args: 0, reserved: 0, lit: 0, name: ptr::null() }
}
fn flatten(e: &Expr, out: &mut Vec<DfxExprNode>, names: &mut Vec<CString>) -> i32 {
    let mut n = empty_node();
    match e {
        Expr::Column(i) => { n.kind = 0; n.column = *i as i32; }
        Expr::Literal(v) => {
            n.kind = 1;
            match v {
                ScalarValue::Null => { n.dtype = 0; }
                ScalarValue::Boolean(b) => { n.dtype = 1; n.lit = *b as u64; }
                ScalarValue::Int8(x) => { n.dtype = 2; n.lit = *x as i64 as u64; }
                ScalarValue::Int16(x) => { n.dtype = 3; n.lit = *x as i64 as u64; }
                ScalarValue::Int32(x) => { n.dtype = 4; n.lit = *x as i64 as u64; }
                ScalarValue::Int64(x) => { n.dtype = 5; n.lit = *x as u64; }
                ScalarValue::UInt8(x) => { n.dtype = 6; n.lit = *x as u64; }
                ScalarValue::UInt16(x) => { n.dtype = 7; n.lit = *x as u64; }
                ScalarValue::UInt32(x) => { n.dtype = 8; n.lit = *x as u64; }
                ScalarValue::UInt64(x) => { n.dtype = 9; n.lit = *x; }
                ScalarValue::Float32(x) => { n.dtype = 10; n.lit = x.to_bits() as u64; }
                ScalarValue::Float64(x) => { n.dtype = 11; n.lit = x.to_bits(); }
                ScalarValue::Utf8(s) => {
                    n.dtype = 12;
                    names.push(CString::new(s.as_str()).unwrap());
                    n.name = names.last().unwrap().as_ptr();
                }
                _ => { n.dtype = 0; }
            }
        }
        Expr::BinaryExpr { left, op, right } => {
            n.kind = 2;
            n.op = operator_code(op);
            n.left = flatten(left, out, names);
            n.right = flatten(right, out, names);
        }
        Expr::IsNotNull(x) => { n.kind = 3; n.left = flatten(x, out, names); }
        Expr::IsNull(x) => { n.kind = 4; n.left = flatten(x, out, names); }
        Expr::Cast { expr, data_type } => { n.kind = 5; n.dtype = dtype_code(data_type); n.left = flatten(expr, out, names); }
        Expr::Sort { expr, .. } => { n.kind = 6; n.left = flatten(expr, out, names); }
        Expr::ScalarFunction { name, args, return_type } => {
            n.kind = 7;
            n.dtype = dtype_code(return_type);
            n.n_args = args.len() as i32;
            names.push(CString::new(name.as_str()).unwrap());
            n.name = names.last().unwrap().as_ptr();
        }
        Expr::AggregateFunction { name, args, return_type } => {
            n.kind = 8;
            n.dtype = dtype_code(return_type);
            n.n_args = args.len() as i32;
            if !args.is_empty() {
                n.left = flatten(&args[0], out, names);
            }
            names.push(CString::new(name.as_str()).unwrap());
            n.name = names.last().unwrap().as_ptr();
        }
    }
    out.push(n);
    (out.len() - 1) as i32
}

/// RuntimeExpr (expression.rs:42-77)
pub struct GpuExpr(*mut DfxRuntimeExpr);
impl Drop for GpuExpr {
    fn drop(&mut self) {
        unsafe { dfx_runtime_expr_free(self.0) }
    }
}
impl GpuExpr {
    pub fn get_name(&self) -> String {
        unsafe { CStr::from_ptr(dfx_runtime_expr_name(self.0)) }.to_string_lossy().into_owned()
    }
    pub fn get_type(&self) -> DataType {
        dtype_from_code(unsafe { dfx_runtime_expr_type(self.0) })
    }
}

fn compile_with(aggregate: bool, expr: &Expr, input_schema: &Schema) -> Result<GpuExpr> {
    let (mut nodes, mut names) = (vec![], vec![]);
    // `names` must not reallocate its CStrings' buffers while `nodes` points into them: CString owns a heap buffer, so
    // moving the CString inside the Vec keeps the pointer valid
    let root = flatten(expr, &mut nodes, &mut names);
    let mut c_schema = export_schema(input_schema);
    let mut out: *mut DfxRuntimeExpr = ptr::null_mut();
    let mut err = [0 as c_char; ERRLEN];
    let code = unsafe {
        if aggregate {
            dfx_compile_expr(nodes.as_ptr(), nodes.len() as i32, root, &c_schema, &mut out, err.as_mut_ptr(), ERRLEN)
        } else {
            dfx_compile_scalar_expr(nodes.as_ptr(), nodes.len() as i32, root, &c_schema, &mut out, err.as_mut_ptr(), ERRLEN)
        }
    };
    unsafe { release_schema(&mut c_schema) };
    check(code, &err)?;
    Ok(GpuExpr(out))
}
/// compile_scalar_expr (expression.rs:283-505): same error cases, raised at the same time
pub fn compile_scalar_expr(expr: &Expr, input_schema: &Schema) -> Result<GpuExpr> {
    compile_with(false, expr, input_schema)
}
/// compile_expr (expression.rs:80-121)
pub fn compile_expr(expr: &Expr, input_schema: &Schema) -> Result<GpuExpr> {
    compile_with(true, expr, input_schema)
}

// ---- RecordBatch <- ArrowArray (buffers are copied into arrow 0.12 Buffers, then the C array is released) -----
unsafe fn import_column(a: &ArrowArray, t: &DataType) -> ArrayRef {
    let n = a.length as usize;
    let off = a.offset as usize;
    let bufs = std::slice::from_raw_parts(a.buffers, a.n_buffers as usize);
    let copy = |p: *const c_void, bytes: usize| -> Buffer {
        if p.is_null() || bytes == 0 { Buffer::from(&[0u8; 8][..0]) } else { Buffer::from(std::slice::from_raw_parts(p as *const u8, bytes)) }
    };
    let mut b = ArrayData::builder(t.clone()).len(n).offset(off);
    if a.null_count != 0 && !bufs[0].is_null() {
        b = b.null_count(if a.null_count < 0 { 0 } else { a.null_count as usize }).null_bit_buffer(copy(bufs[0], (off + n + 7) / 8));
    }
    match t {
        DataType::Utf8 => {
            let offsets = std::slice::from_raw_parts(bufs[1] as *const i32, off + n + 1);
            let data_len = offsets[off + n] as usize;
            b = b.add_buffer(copy(bufs[1], (off + n + 1) * 4)).add_buffer(copy(bufs[2], data_len));
            Arc::new(BinaryArray::from(b.build()))
        }
        DataType::Boolean => Arc::new(BooleanArray::from(b.add_buffer(copy(bufs[1], (off + n + 7) / 8)).build())),
        DataType::Int8 => Arc::new(PrimitiveArray::<Int8Type>::from(b.add_buffer(copy(bufs[1], (off + n) * 1)).build())),
        DataType::Int16 => Arc::new(PrimitiveArray::<Int16Type>::from(b.add_buffer(copy(bufs[1], (off + n) * 2)).build())),
        DataType::Int32 => Arc::new(PrimitiveArray::<Int32Type>::from(b.add_buffer(copy(bufs[1], (off + n) * 4)).build())),
        DataType::Int64 => Arc::new(PrimitiveArray::<Int64Type>::from(b.add_buffer(copy(bufs[1], (off + n) * 8)).build())),
        DataType::UInt8 => Arc::new(PrimitiveArray::<UInt8Type>::from(b.add_buffer(copy(bufs[1], (off + n) * 1)).build())),
        DataType::UInt16 => Arc::new(PrimitiveArray::<UInt16Type>::from(b.add_buffer(copy(bufs[1], (off + n) * 2)).build())),
        DataType::UInt32 => Arc::new(PrimitiveArray::<UInt32Type>::from(b.add_buffer(copy(bufs[1], (off + n) * 4)).build())),
        DataType::UInt64 => Arc::new(PrimitiveArray::<UInt64Type>::from(b.add_buffer(copy(bufs[1], (off + n) * 8)).build())),
        DataType::Float32 => Arc::new(PrimitiveArray::<Float32Type>::from(b.add_buffer(copy(bufs[1], (off + n) * 4)).build())),
        _ => Arc::new(PrimitiveArray::<Float64Type>::from(b.add_buffer(copy(bufs[1], (off + n) * 8)).build())),
    }
}
unsafe fn import_record_batch(mut a: ArrowArray, schema: &Arc<Schema>) -> RecordBatch {
    let mut cols: Vec<ArrayRef> = vec![];
    for i in 0..a.n_children as usize {
        let c = &**a.children.add(i);
        cols.push(import_column(c, schema.field(i).data_type()));
    }
    if let Some(r) = a.release {
        r(&mut a);
    }
    RecordBatch::new(schema.clone(), cols)
}

// ---- Relation -> ArrowArrayStream (host producers: DataSourceRelation over CSV / Parquet / memory) -------------
struct Producer {
    rel: Rc<RefCell<Relation>>,
    schema: Arc<Schema>,
    last_error: CString,
}
struct BatchHolder {
    batch: RecordBatch, // keeps the arrow buffers alive until the consumer releases the array
    children: Vec<ArrowArray>,
    child_ptrs: Vec<*mut ArrowArray>,
    buffers: Vec<Vec<*const c_void>>,
}
unsafe extern "C" fn release_batch(a: *mut ArrowArray) {
    if !(*a).private_data.is_null() {
        drop(Box::from_raw((*a).private_data as *mut BatchHolder));
    }
    (*a).release = None;
}
unsafe extern "C" fn release_child(a: *mut ArrowArray) {
    (*a).release = None;
}
unsafe fn export_batch(batch: RecordBatch, out: *mut ArrowArray) {
    let n = batch.num_rows() as i64;
    let mut h = Box::new(BatchHolder { batch, children: vec![], child_ptrs: vec![], buffers: vec![] });
    for i in 0..h.batch.num_columns() {
        let data = h.batch.column(i).data();
        let mut bufs: Vec<*const c_void> = vec![];
        bufs.push(match data.null_bitmap() {
            Some(b) => b.to_buffer().raw_data() as *const c_void,
            None => ptr::null(),
        });
        for b in data.buffers() {
            bufs.push(b.raw_data() as *const c_void);
        }
        h.buffers.push(bufs);
        let nb = h.buffers[i].len() as i64;
        let bp = h.buffers[i].as_mut_ptr();
        h.children.push(ArrowArray {
            length: data.len() as i64,
            null_count: data.null_count() as i64,
            offset: data.offset() as i64,
            n_buffers: nb,
            n_children: 0,
            buffers: bp,
            children: ptr::null_mut(),
            dictionary: ptr::null_mut(),
            release: Some(release_child),
            private_data: ptr::null_mut(),
        });
    }
    for c in h.children.iter_mut() {
        h.child_ptrs.push(c as *mut ArrowArray);
    }
    static mut NO_BUFFERS: [*const c_void; 1] = [ptr::null()];
    *out = ArrowArray {
        length: n,
        null_count: 0,
        offset: 0,
        n_buffers: 1,
        n_children: h.child_ptrs.len() as i64,
        buffers: NO_BUFFERS.as_mut_ptr(),
        children: h.child_ptrs.as_mut_ptr(),
        dictionary: ptr::null_mut(),
        release: Some(release_batch),
        private_data: Box::into_raw(h) as *mut c_void,
    };
}
unsafe extern "C" fn producer_get_schema(s: *mut ArrowArrayStream, out: *mut ArrowSchema) -> i32 {
    let p = &*((*s).private_data as *const Producer);
    *out = export_schema(&p.schema);
    0
}
unsafe extern "C" fn producer_get_next(s: *mut ArrowArrayStream, out: *mut ArrowArray) -> i32 {
    let p = &mut *((*s).private_data as *mut Producer);
    let next = p.rel.borrow_mut().next();
    match next {
        Ok(Some(batch)) => {
            export_batch(batch, out);
            0
        }
        Ok(None) => {
            *out = mem::zeroed(); // released array == end of stream
            0
        }
        Err(e) => {
            p.last_error = CString::new(format!("{:?}", e)).unwrap_or_default();
            8 // DFX_EXECUTION_ERROR
        }
    }
}
unsafe extern "C" fn producer_last_error(s: *mut ArrowArrayStream) -> *const c_char {
    (*((*s).private_data as *const Producer)).last_error.as_ptr()
}
unsafe extern "C" fn producer_release(s: *mut ArrowArrayStream) {
    if !(*s).private_data.is_null() {
        drop(Box::from_raw((*s).private_data as *mut Producer));
    }
    (*s).release = None;
}
/// Any reference Relation as a producer stream.
pub fn export_relation(rel: Rc<RefCell<Relation>>) -> ArrowArrayStream {
    let schema = rel.borrow().schema().clone();
    let p = Box::new(Producer { rel, schema, last_error: CString::default() });
    ArrowArrayStream {
        get_schema: Some(producer_get_schema),
        get_next: Some(producer_get_next),
        get_last_error: Some(producer_last_error),
        release: Some(producer_release),
        private_data: Box::into_raw(p) as *mut c_void,
    }
}

// ---- a library operator seen from Rust: impl Relation (relation.rs:27-32) ------------------------------------
pub struct GpuRelation {
    stream: Box<ArrowArrayStream>,
    schema: Arc<Schema>,
}
impl GpuRelation {
    fn from_stream(mut stream: Box<ArrowArrayStream>, declared: Arc<Schema>) -> Result<Self> {
        // the aggregate is created with Schema::empty() (context.rs:185): take the schema the library derived
        let schema = if declared.fields().is_empty() {
            let mut s: ArrowSchema = unsafe { mem::zeroed() };
            let rc = unsafe { (stream.get_schema.unwrap())(&mut *stream, &mut s) };
            if rc != 0 {
                return Err(ExecutionError::General("get_schema failed".to_string()));
            }
            let sch = unsafe { import_schema(&s) };
            if let Some(r) = s.release {
                unsafe { r(&mut s) };
            }
            Arc::new(sch)
        } else {
            declared
        };
        Ok(GpuRelation { stream, schema })
    }
    fn new_out() -> Box<ArrowArrayStream> {
        Box::new(unsafe { mem::zeroed() })
    }

    pub fn filter(input: Rc<RefCell<Relation>>, expr: GpuExpr, schema: Arc<Schema>) -> Result<Self> {
        Self::filter_with_options(input, expr, schema, &[])
    }
    pub fn filter_with_options(input: Rc<RefCell<Relation>>, expr: GpuExpr, schema: Arc<Schema>, options: &[(&str, i64)]) -> Result<Self> {
        let (mut inp, mut out, mut err) = (export_relation(input), Self::new_out(), [0 as c_char; ERRLEN]);
        let keys: Vec<CString> = options.iter().map(|(k, _)| CString::new(*k).unwrap()).collect();
        let opts: Vec<DfxOption> = options.iter().zip(keys.iter()).map(|((_, v), k)| DfxOption { key: k.as_ptr(), value: *v }).collect();
        let mut cs = export_schema(&schema);
        let code = unsafe {
            dfx_filter_relation_new_with_options(&mut inp, expr.0, &cs, opts.as_ptr(), opts.len() as i32, &mut *out, err.as_mut_ptr(), ERRLEN)
        };
        unsafe { release_schema(&mut cs) };
        check(code, &err)?;
        Self::from_stream(out, schema)
    }
    pub fn project(input: Rc<RefCell<Relation>>, exprs: Vec<GpuExpr>, schema: Arc<Schema>) -> Result<Self> {
        let (mut inp, mut out, mut err) = (export_relation(input), Self::new_out(), [0 as c_char; ERRLEN]);
        let hs: Vec<*const DfxRuntimeExpr> = exprs.iter().map(|e| e.0 as *const DfxRuntimeExpr).collect();
        let mut cs = export_schema(&schema);
        let code = unsafe { dfx_project_relation_new(&mut inp, hs.as_ptr(), hs.len() as i32, &cs, &mut *out, err.as_mut_ptr(), ERRLEN) };
        unsafe { release_schema(&mut cs) };
        check(code, &err)?;
        Self::from_stream(out, schema)
    }
    pub fn aggregate(schema: Arc<Schema>, input: Rc<RefCell<Relation>>, group: Vec<GpuExpr>, aggr: Vec<GpuExpr>) -> Result<Self> {
        Self::aggregate_with_options(schema, input, group, aggr, &[])
    }
    /// The same with per-operator options, e.g. `&[("agg.strategy", 1)]`: no process-wide state is touched.
    pub fn aggregate_with_options(schema: Arc<Schema>, input: Rc<RefCell<Relation>>, group: Vec<GpuExpr>, aggr: Vec<GpuExpr>,
                                  options: &[(&str, i64)]) -> Result<Self> {
        let (mut inp, mut out, mut err) = (export_relation(input), Self::new_out(), [0 as c_char; ERRLEN]);
        let g: Vec<*const DfxRuntimeExpr> = group.iter().map(|e| e.0 as *const DfxRuntimeExpr).collect();
        let a: Vec<*const DfxRuntimeExpr> = aggr.iter().map(|e| e.0 as *const DfxRuntimeExpr).collect();
        let keys: Vec<CString> = options.iter().map(|(k, _)| CString::new(*k).unwrap()).collect();
        let opts: Vec<DfxOption> = options.iter().zip(keys.iter()).map(|((_, v), k)| DfxOption { key: k.as_ptr(), value: *v }).collect();
        let mut cs = export_schema(&schema);
        let code = unsafe {
            dfx_aggregate_relation_new_with_options(&cs, &mut inp, g.as_ptr(), g.len() as i32, a.as_ptr(), a.len() as i32,
                                                    opts.as_ptr(), opts.len() as i32, &mut *out, err.as_mut_ptr(), ERRLEN)
        };
        unsafe { release_schema(&mut cs) };
        check(code, &err)?;
        Self::from_stream(out, schema)
    }
    /// CsvDataSource::new(filename, schema, batch_size) + DataSourceRelation::new: a leaf that parses on the GPU
    pub fn csv(filename: &str, schema: Arc<Schema>, batch_size: usize) -> Result<Self> {
        let (mut out, mut err) = (Self::new_out(), [0 as c_char; ERRLEN]);
        let name = CString::new(filename).unwrap();
        let mut cs = export_schema(&schema);
        let code = unsafe { dfx_csv_datasource_new(name.as_ptr(), &cs, batch_size as i64, &mut *out, err.as_mut_ptr(), ERRLEN) };
        unsafe { release_schema(&mut cs) };
        check(code, &err)?;
        Self::from_stream(out, schema)
    }
    pub fn sort(input: Rc<RefCell<Relation>>, keys: Vec<(GpuExpr, bool)>, schema: Arc<Schema>) -> Result<Self> {
        let (mut inp, mut out, mut err) = (export_relation(input), Self::new_out(), [0 as c_char; ERRLEN]);
        let hs: Vec<*const DfxRuntimeExpr> = keys.iter().map(|(e, _)| e.0 as *const DfxRuntimeExpr).collect();
        let asc: Vec<i32> = keys.iter().map(|(_, a)| *a as i32).collect();
        let mut cs = export_schema(&schema);
        let code = unsafe { dfx_sort_relation_new(&mut inp, hs.as_ptr(), asc.as_ptr(), hs.len() as i32, &cs, &mut *out, err.as_mut_ptr(), ERRLEN) };
        unsafe { release_schema(&mut cs) };
        check(code, &err)?;
        Self::from_stream(out, schema)
    }
    pub fn limit(input: Rc<RefCell<Relation>>, limit: usize, schema: Arc<Schema>) -> Result<Self> {
        let (mut inp, mut out, mut err) = (export_relation(input), Self::new_out(), [0 as c_char; ERRLEN]);
        let mut cs = export_schema(&schema);
        let code = unsafe { dfx_limit_relation_new(&mut inp, limit as i64, &cs, &mut *out, err.as_mut_ptr(), ERRLEN) };
        unsafe { release_schema(&mut cs) };
        check(code, &err)?;
        Self::from_stream(out, schema)
    }
}

impl GpuRelation {
    /// Physical plan of this operator and everything the library chained below it: what was fused, which kernel family
    /// runs each program (the counterpart of the `println!("Logical plan: ...")` in context.rs:105).
    pub fn explain(&mut self) -> String {
        let n = unsafe { dfx_relation_explain(&mut *self.stream, ptr::null_mut(), 0) };
        if n < 0 {
            return String::new();
        }
        let mut buf = vec![0 as c_char; n as usize + 1];
        unsafe { dfx_relation_explain(&mut *self.stream, buf.as_mut_ptr(), buf.len()) };
        unsafe { CStr::from_ptr(buf.as_ptr()) }.to_string_lossy().into_owned()
    }
}

// ---- multi-GPU GROUP BY: one process per GPU, group partials exchanged over RCCL inside the library -------------------
// (include/dfx.h: dfx_comm_*, dfx_aggregate_exchange).  The reference is single-process (README.md:20); a host that
// shards its input over the GPUs of a node creates one communicator per process -- rank 0 draws the 128-byte id and
// hands it to the others by whatever channel the host has -- and calls `exchange` on every rank's aggregate before it
// pulls the result: each rank then emits the groups it owns.
pub const COMM_ID_BYTES: usize = 128;
pub struct GpuCommunicator(*mut DfxComm);
impl GpuCommunicator {
    pub fn unique_id() -> Result<[u8; COMM_ID_BYTES]> {
        let (mut id, mut err) = ([0u8; COMM_ID_BYTES], [0 as c_char; ERRLEN]);
        check(unsafe { dfx_comm_unique_id(id.as_mut_ptr(), err.as_mut_ptr(), ERRLEN) }, &err)?;
        Ok(id)
    }
    pub fn new(id: &[u8; COMM_ID_BYTES], world: i32, rank: i32) -> Result<Self> {
        let (mut h, mut err) = (ptr::null_mut(), [0 as c_char; ERRLEN]);
        check(unsafe { dfx_comm_init(id.as_ptr(), world, rank, &mut h, err.as_mut_ptr(), ERRLEN) }, &err)?;
        Ok(GpuCommunicator(h))
    }
}
impl Drop for GpuCommunicator {
    fn drop(&mut self) {
        unsafe { dfx_comm_destroy(self.0) }
    }
}
impl GpuRelation {
    /// Aggregate relations only: exchange the group partials with the other ranks (counts, then buckets, as grouped
    /// ncclSend / ncclRecv on the library's stream).  Returns (groups sent, groups received, bytes sent).
    pub fn exchange(&mut self, comm: &GpuCommunicator) -> Result<(i64, i64, i64)> {
        let (mut stats, mut err) = ([0i64; 4], [0 as c_char; ERRLEN]);
        check(unsafe { dfx_aggregate_exchange(&mut *self.stream, comm.0, stats.as_mut_ptr(), err.as_mut_ptr(), ERRLEN) }, &err)?;
        Ok((stats[0], stats[1], stats[2]))
    }
    /// The same protocol with the collective done by the host (e.g. MPI): see INTEGRATION.md.
    pub fn partial_build(&mut self, world: i32) -> Result<(i32, Vec<i64>)> {
        let (mut nw, mut counts, mut err) = (0i32, vec![0i64; world as usize], [0 as c_char; ERRLEN]);
        check(unsafe { dfx_aggregate_partial_build(&mut *self.stream, world, &mut nw, counts.as_mut_ptr(), err.as_mut_ptr(), ERRLEN) }, &err)?;
        Ok((nw, counts))
    }
    pub unsafe fn partial_export(&mut self, dst_device: *mut c_void, dst_words: i64) -> Result<()> {
        let mut err = [0 as c_char; ERRLEN];
        check(dfx_aggregate_partial_export(&mut *self.stream, dst_device, dst_words, err.as_mut_ptr(), ERRLEN), &err)
    }
    pub unsafe fn partial_import(&mut self, src_device: *const c_void, counts: &[i64]) -> Result<()> {
        let mut err = [0 as c_char; ERRLEN];
        check(dfx_aggregate_partial_import(&mut *self.stream, src_device, counts.as_ptr(), counts.len() as i32, err.as_mut_ptr(), ERRLEN), &err)
    }
}

impl Relation for GpuRelation {
    fn next(&mut self) -> Result<Option<RecordBatch>> {
        let mut a: ArrowArray = unsafe { mem::zeroed() };
        let rc = unsafe { (self.stream.get_next.unwrap())(&mut *self.stream, &mut a) };
        if rc != 0 {
            let msg = unsafe {
                let p = (self.stream.get_last_error.unwrap())(&mut *self.stream);
                if p.is_null() { String::new() } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
            };
            let mut err = [0 as c_char; ERRLEN];
            for (i, b) in msg.bytes().take(ERRLEN - 1).enumerate() {
                err[i] = b as c_char;
            }
            return check(rc, &err).map(|_| None);
        }
        if a.release.is_none() {
            return Ok(None); // end of stream
        }
        Ok(Some(unsafe { import_record_batch(a, &self.schema) }))
    }
    fn schema(&self) -> &Arc<Schema> {
        &self.schema
    }
}
impl Drop for GpuRelation {
    fn drop(&mut self) {
        if let Some(r) = self.stream.release {
            unsafe { r(&mut *self.stream) }
        }
    }
}

// ---- the in-memory DataSource (datasource.rs:27-30), resident in HBM ------------------------------------------------
/// A table uploaded once and scanned many times.  `ExecutionContext::register_datasource` keeps an
/// `Rc<RefCell<DataSource>>` that is consumed by the first query (datasource.rs:55-57); registering a `GpuTable`
/// instead lets every query start from `table.scan(batch_rows)`, a leaf whose batches never leave the device.
pub struct GpuTable {
    handle: *mut DfxTable,
    schema: Arc<Schema>,
}
impl GpuTable {
    /// Drains `ds` (CSV, Parquet, anything that implements DataSource) through DataSourceRelation into HBM.
    pub fn load(ds: Rc<RefCell<DataSource>>) -> Result<Self> {
        let schema = ds.borrow().schema().clone();
        let rel: Rc<RefCell<Relation>> = Rc::new(RefCell::new(DataSourceRelation::new(ds)));
        let (mut inp, mut err) = (export_relation(rel), [0 as c_char; ERRLEN]);
        let mut handle: *mut DfxTable = ptr::null_mut();
        check(unsafe { dfx_table_from_stream(&mut inp, &mut handle, err.as_mut_ptr(), ERRLEN) }, &err)?;
        Ok(GpuTable { handle, schema })
    }
    pub fn num_rows(&self) -> usize {
        unsafe { dfx_table_num_rows(self.handle) as usize }
    }
    pub fn schema(&self) -> &Arc<Schema> {
        &self.schema
    }
    /// DataSourceRelation over the resident table; `batch_rows` = 0 hands the whole table over as one batch
    /// (device batches are slices of the resident buffers, so a large batch costs nothing).
    pub fn scan(&self, batch_rows: usize) -> Result<GpuRelation> {
        let (mut out, mut err) = (GpuRelation::new_out(), [0 as c_char; ERRLEN]);
        check(unsafe { dfx_table_scan_new(self.handle, batch_rows as i64, &mut *out, err.as_mut_ptr(), ERRLEN) }, &err)?;
        GpuRelation::from_stream(out, self.schema.clone())
    }
    /// ... over the rows `[row_begin, row_begin + n_rows)` only (`row_begin` a multiple of 64): one partition of a resident
    /// table as a DataSource of its own.
    pub fn scan_range(&self, row_begin: usize, n_rows: usize, batch_rows: usize) -> Result<GpuRelation> {
        let (mut out, mut err) = (GpuRelation::new_out(), [0 as c_char; ERRLEN]);
        check(unsafe {
            dfx_table_scan_range_new(self.handle, row_begin as i64, n_rows as i64, batch_rows as i64, &mut *out, err.as_mut_ptr(), ERRLEN)
        }, &err)?;
        GpuRelation::from_stream(out, self.schema.clone())
    }
}
impl Drop for GpuTable {
    fn drop(&mut self) {
        unsafe { dfx_table_free(self.handle) } // streams created by scan() hold their own reference to the buffers
    }
}

/// Blocks until everything the library has launched is complete (only needed around timing code).
pub fn synchronize() -> Result<()> {
    let mut err = [0 as c_char; ERRLEN];
    check(unsafe { dfx_synchronize(err.as_mut_ptr(), ERRLEN) }, &err)
}
/// Tuning knobs of include/dfx.h (`agg.strategy`, `scan.fast`, ...); unknown keys are an error.
pub fn set_option(key: &str, value: i64) -> Result<()> {
    let k = CString::new(key).unwrap();
    if unsafe { dfx_set_option(k.as_ptr(), value) } == 0 {
        Ok(())
    } else {
        Err(ExecutionError::General(format!("unknown option '{}'", key)))
    }
}

// Note on chaining: when `input` is itself a GpuRelation, pass its stream straight through instead of wrapping it in
// export_relation (the library recognises its own streams and keeps the batches on the device; a Filter directly
// under an Aggregate is fused).  With `Rc<RefCell<Relation>>` trait objects that takes a downcast helper on the
// Relation trait (`fn as_gpu(&mut self) -> Option<&mut GpuRelation> { None }`), three lines in relation.rs.

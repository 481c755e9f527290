// dfx_host.cpp -- status helpers, dtype tables, Arrow schema conversion, device context, pools.
#include "dfx_host.hpp"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <map>
#include <mutex>

namespace dfx {

std::string strfmt(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  return std::string(buf);
}

int32_t to_c(const Status& s, char* err, size_t errlen) {
  if (err && errlen) {
    snprintf(err, errlen, "%s", s.msg.c_str());
  }
  return s.code;
}

void explain_line(std::string* out, int depth, const std::string& text) {
  out->append((size_t)depth * 2, ' ');
  out->append(text);
  out->push_back('\n');
}
std::string explain_program(const DevProgram& P) {
  return strfmt("program: %d columns, %d instructions, %d literals", P.n_cols, P.n_ins, P.n_imm);
}
void Relation::explain(std::string* out, int depth) const {
  static const char* const names[] = {"HostStream", "TableScan", "Filter", "Project", "Aggregate", "CsvDataSource", "Sort", "Limit"};
  explain_line(out, depth, names[(int)kind()]);
}

const char* dtype_name(int dt) {
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

int dtype_width(int dt) {
  switch (dt) {
    case DFX_INT8: case DFX_UINT8: return 1;
    case DFX_INT16: case DFX_UINT16: return 2;
    case DFX_INT32: case DFX_UINT32: case DFX_FLOAT32: return 4;
    case DFX_INT64: case DFX_UINT64: case DFX_FLOAT64: return 8;
    default: return 0;
  }
}
bool dtype_is_numeric(int dt) { return dt >= DFX_INT8 && dt <= DFX_FLOAT64; }
bool dtype_is_int(int dt) { return dt >= DFX_INT8 && dt <= DFX_UINT64; }
bool dtype_is_signed(int dt) { return dt >= DFX_INT8 && dt <= DFX_INT64; }

const char* dtype_arrow_format(int dt) {
  switch (dt) {
    case DFX_BOOLEAN: return "b";
    case DFX_INT8: return "c";
    case DFX_INT16: return "s";
    case DFX_INT32: return "i";
    case DFX_INT64: return "l";
    case DFX_UINT8: return "C";
    case DFX_UINT16: return "S";
    case DFX_UINT32: return "I";
    case DFX_UINT64: return "L";
    case DFX_FLOAT32: return "f";
    case DFX_FLOAT64: return "g";
    case DFX_UTF8: return "u";
    default: return "n";
  }
}

int dtype_from_arrow_format(const char* f) {
  if (!f || !f[0] || f[1]) return DFX_TYPE_NONE;
  switch (f[0]) {
    case 'b': return DFX_BOOLEAN;
    case 'c': return DFX_INT8;
    case 's': return DFX_INT16;
    case 'i': return DFX_INT32;
    case 'l': return DFX_INT64;
    case 'C': return DFX_UINT8;
    case 'S': return DFX_UINT16;
    case 'I': return DFX_UINT32;
    case 'L': return DFX_UINT64;
    case 'f': return DFX_FLOAT32;
    case 'g': return DFX_FLOAT64;
    case 'u': return DFX_UTF8;
    default: return DFX_TYPE_NONE;
  }
}

Status schema_from_arrow(const struct ArrowSchema* s, SchemaInfo* out) {
  out->fields.clear();
  if (!s || !s->format) return Status::OK();  // Schema::empty()
  if (strcmp(s->format, "+s") != 0)
    return Status::Err(DFX_ARROW_ERROR, strfmt("expected a struct schema (+s), got '%s'", s->format));
  for (int64_t i = 0; i < s->n_children; ++i) {
    const struct ArrowSchema* c = s->children[i];
    Field f;
    f.name = c->name ? c->name : "";
    f.dtype = dtype_from_arrow_format(c->format);
    if (f.dtype == DFX_TYPE_NONE)
      return Status::Err(DFX_NOT_IMPLEMENTED,
                         strfmt("unsupported Arrow type '%s' for column '%s'", c->format ? c->format : "?", f.name.c_str()));
    f.nullable = (c->flags & ARROW_FLAG_NULLABLE) != 0;
    out->fields.push_back(f);
  }
  return Status::OK();
}

namespace {
struct SchemaPriv {
  std::string format, name;
  std::vector<struct ArrowSchema> kids;
  std::vector<struct ArrowSchema*> kid_ptrs;
};
void release_schema(struct ArrowSchema* s) {
  if (!s || !s->release) return;
  for (int64_t i = 0; i < s->n_children; ++i)
    if (s->children[i] && s->children[i]->release) s->children[i]->release(s->children[i]);
  delete (SchemaPriv*)s->private_data;
  s->release = nullptr;
}
void make_leaf(const Field& f, struct ArrowSchema* out) {
  SchemaPriv* p = new SchemaPriv();
  p->format = dtype_arrow_format(f.dtype);
  p->name = f.name;
  memset(out, 0, sizeof(*out));
  out->format = p->format.c_str();
  out->name = p->name.c_str();
  out->flags = f.nullable ? ARROW_FLAG_NULLABLE : 0;
  out->release = release_schema;
  out->private_data = p;
}
}  // namespace

void schema_to_arrow(const SchemaInfo& s, struct ArrowSchema* out) {
  SchemaPriv* p = new SchemaPriv();
  p->format = "+s";
  p->name = "";
  p->kids.resize(s.fields.size());
  p->kid_ptrs.resize(s.fields.size());
  for (size_t i = 0; i < s.fields.size(); ++i) {
    make_leaf(s.fields[i], &p->kids[i]);
    p->kid_ptrs[i] = &p->kids[i];
  }
  memset(out, 0, sizeof(*out));
  out->format = p->format.c_str();
  out->name = p->name.c_str();
  out->n_children = (int64_t)s.fields.size();
  out->children = p->kid_ptrs.empty() ? nullptr : p->kid_ptrs.data();
  out->release = release_schema;
  out->private_data = p;
}

// ---- context ------------------------------------------------------------------------------------
Context& ctx() {
  static Context c;
  return c;
}

Status ensure_init() {
  Context& c = ctx();
  if (c.initialised) return Status::OK();
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return Status::Err(DFX_EXECUTION_ERROR,
                       strfmt("no HIP device available (%s): the dfx execution path needs an MI355X; "
                              "there is no CPU fallback", hipGetErrorString(e)));
  if (c.device >= n) return Status::Err(DFX_EXECUTION_ERROR, strfmt("device %d out of range (%d devices)", c.device, n));
  DFX_HIP(hipSetDevice(c.device));
  DFX_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
  DFX_HIP(hipStreamCreateWithFlags(&c.aux, hipStreamNonBlocking));
  c.initialised = true;
  return Status::OK();
}

const uint8_t* device_ones_block() {
  static std::once_flag once;
  static uint8_t* block = nullptr;
  std::call_once(once, [] {
    if (!ensure_init().ok()) return;
    void* p = nullptr;
    uint8_t ones[256];
    memset(ones, 0xFF, sizeof(ones));
    if (hipMalloc(&p, 256) != hipSuccess || hipMemcpy(p, ones, 256, hipMemcpyHostToDevice) != hipSuccess) {  // (synchronous: complete on return)
      (void)hipGetLastError();
      return;
    }
    block = (uint8_t*)p;  // (lives as long as the process)
  });
  return block;
}

// ---- pools --------------------------------------------------------------------------------------
namespace {
struct Pool {
  std::mutex mu;
  std::multimap<size_t, void*> free_list;
  bool pinned;
  explicit Pool(bool p) : pinned(p) {}
  static size_t round_up(size_t b) {
    if (b < 4096) return 4096;
    if (b <= (1u << 20)) {  // next power of two
      size_t r = 4096;
      while (r < b) r <<= 1;
      return r;
    }
    const size_t q = 1u << 20;  // 1 MiB granules above 1 MiB
    return (b + q - 1) / q * q;
  }
  static constexpr size_t kSpareBudget = (size_t)1 << 30;
  size_t spare_bytes = 0;  // (only read / written on the allocation path of the pinned pool)
  int inject_oom = 0;      // test hook, see take()
  void* take(size_t bytes, hipError_t* e) {
    {
      std::lock_guard<std::mutex> lk(mu);
      auto it = free_list.lower_bound(bytes);
      if (it != free_list.end() && it->first <= bytes + bytes / 4) {
        void* p = it->second;
        free_list.erase(it);
        return p;
      }
    }
    void* p = nullptr;
    if (!pinned && inject_oom > 0) {  // test hook: a REAL failed hipMalloc (an impossible size), so that HIP's sticky error is set
      --inject_oom;
      *e = hipMalloc(&p, (size_t)1 << 50);
      p = nullptr;
    } else {
      *e = pinned ? hipHostMalloc(&p, bytes, hipHostMallocDefault) : hipMalloc(&p, bytes);
    }
    if (*e != hipSuccess) {  // release cached blocks and retry once
      // The failed attempt stays in HIP's sticky last-error slot: cleared here, or the next launch_* that returns
      // hipGetLastError() reports "out of memory" for a kernel that launched correctly (found by tools/soak.py: a long-running
      // process with many table sizes fills HBM with cached blocks, the first miss after that trims the pool and succeeds).
      (void)hipGetLastError();
      trim();
      *e = pinned ? hipHostMalloc(&p, bytes, hipHostMallocDefault) : hipMalloc(&p, bytes);
      if (*e != hipSuccess) (void)hipGetLastError();  // reported through *e, not through somebody else's launch
    }
    // Pinning a large host buffer costs milliseconds (measured: 13-17 ms for the two 8 MB result columns of a 10^6-group
    // aggregate whenever a query found the pool empty because the consumer still held the previous result).  A miss on
    // a large pinned block therefore stocks one spare of the same size: the next miss becomes a hit.
    // Bounded: at most kSpareBudget bytes of spares over the life of the pool (workloads with many distinct result sizes would
    // otherwise double their locked host memory), and a spare that cannot be had leaves no sticky HIP error behind.
    if (p && pinned && bytes >= (1u << 20) && bytes <= (256u << 20) && spare_bytes + bytes <= kSpareBudget) {
      void* spare = nullptr;
      if (hipHostMalloc(&spare, bytes, hipHostMallocDefault) == hipSuccess && spare) {
        spare_bytes += bytes;
        give(bytes, spare);
      } else {
        (void)hipGetLastError();  // (the next launch_* that returns hipGetLastError() must not report this)
      }
    }
    return p;
  }
  void give(size_t bytes, void* p) {
    std::lock_guard<std::mutex> lk(mu);
    free_list.emplace(bytes, p);
  }
  void trim() {
    std::lock_guard<std::mutex> lk(mu);
    for (auto& kv : free_list) {
      if (pinned) (void)hipHostFree(kv.second);
      else (void)hipFree(kv.second);
    }
    free_list.clear();
  }
};
Pool& dev_pool() {
  static Pool* p = new Pool(false);
  return *p;
}
Pool& pin_pool() {
  static Pool* p = new Pool(true);
  return *p;
}

std::shared_ptr<void> pool_alloc(Pool& pool, size_t bytes, Status* st) {
  Status init = ensure_init();
  if (!init.ok()) {
    if (st) *st = init;
    return nullptr;
  }
  const size_t cap = Pool::round_up(bytes ? bytes : 1);
  hipError_t e = hipSuccess;
  void* p = pool.take(cap, &e);
  if (!p) {
    if (st)
      *st = Status::Err(DFX_EXECUTION_ERROR, strfmt("%s allocation of %zu bytes failed: %s",
                                                    pool.pinned ? "pinned host" : "device", cap, hipGetErrorString(e)));
    return nullptr;
  }
  Pool* pp = &pool;
  return std::shared_ptr<void>(p, [pp, cap](void* q) { pp->give(cap, q); });
}
}  // namespace

long long ScopedUs::now() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (long long)ts.tv_sec * 1000000ll + ts.tv_nsec / 1000;
}

Counters& counters() {
  static Counters c;
  return c;
}

std::shared_ptr<void> device_alloc(size_t bytes, Status* st) { return pool_alloc(dev_pool(), bytes, st); }
std::shared_ptr<void> pinned_alloc(size_t bytes, Status* st) { return pool_alloc(pin_pool(), bytes, st); }
void pool_trim() {
  dev_pool().trim();
  pin_pool().trim();
}
void pool_inject_oom(int n) { dev_pool().inject_oom = n; }

}  // namespace dfx

// dfx_k_sort.hip -- ORDER BY on the device.  The reference's planner emits
// LogicalPlan::Sort { expr: [Expr::Sort { expr, asc }], .. } and LogicalPlan::Limit (sqlplanner.rs:142-183,
// logicalplan.rs:313-338) but its executor stops at `unimplemented!()` (context.rs:113,194); there is no reference
// behaviour to match, so the semantics are the oracle's (tests/oracle.py: sort_batches) and marked unpinned.
//
//   key image      every sort key becomes an order-preserving 64-bit image (signed: sign bit flipped; floats: the
//                  IEEE total-order trick with NaN above +inf; descending: complemented); NULL is larger than every
//                  value (a separate 0/1 image sorted as one more, most significant, digit);
//   stable LSD radix sort of (image, row index) pairs, 8-bit digits, keys from the last to the first; digits that are
//                  constant over the input are skipped (one histogram kernel per key decides);
//                  a pass = per-wave-tile digit counts -> exclusive scan (digit-major) -> stable scatter in which a wave
//                  ranks its 64 elements with 8 ballots (match-any) and keeps running digit cursors in LDS;
//   gather         the final permutation is resolved to (batch, row) once, then every payload column is gathered
//                  straight from the input batches (no concatenation): fixed width, validity bits, Utf8.
// Bound: HBM / store transactions (8 B + 4 B scattered per element and pass).
#include <algorithm>

#include "dfx_kernels_inl.hpp"
#include "dfx_launch.hpp"

namespace dfx {

constexpr int kSortWaveTile = 2048;  // elements ranked by one wave per pass (32 rounds of 64)

// ---- key images ----------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_sort_image(const void* __restrict__ values, const uint8_t* __restrict__ validity,
                                                      int64_t bit_offset, uint8_t dtype, int asc, int64_t n,
                                                      uint64_t* __restrict__ image, uint64_t* __restrict__ null_image) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const uint64_t v = load_canonical(dtype, values, i, bit_offset);
    uint64_t img;
    if (dtype == T_F64 || dtype == T_F32) {
      const double d = dtype == T_F64 ? as_f64(v) : (double)as_f32(v);  // f32 -> f64 is exact and order-preserving
      uint64_t b = f64_bits(d);
      if (d != d) b = 0x7FF8000000000000ull;  // every NaN sorts above +inf
      img = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
    } else if (is_signed_int(dtype)) {
      img = v ^ 0x8000000000000000ull;
    } else {
      img = v;  // unsigned, Boolean
    }
    const bool valid = validity == nullptr || get_bit(validity, bit_offset + i);
    if (!valid) img = 0;  // equal among nulls: their order is the input order (stable)
    image[i] = asc ? img : ~img;
    if (null_image) null_image[i] = (uint64_t)((valid ? 0 : 1) ^ (asc ? 0 : 1));  // asc: nulls last; desc: nulls first
  }
}

// Utf8 keys: byte-wise lexicographic order = LSD sort over (length, last 8-byte chunk, ..., first 8-byte chunk) where a chunk
// is the big-endian image of bytes [8j, 8j + 8) padded with zeros: if all padded chunks of two strings are equal, one is
// the other plus NUL bytes, and the shorter sorts first.  chunk < 0: the length image.
__global__ __launch_bounds__(kBlock) void k_sort_image_utf8(const int32_t* __restrict__ offsets, const uint8_t* __restrict__ data,
                                                           const uint8_t* __restrict__ validity, int64_t bit_offset, int chunk,
                                                           int asc, int64_t n, uint64_t* __restrict__ image,
                                                           uint64_t* __restrict__ null_image, unsigned int* __restrict__ max_len) {
  unsigned int mx = 0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const int32_t b = offsets[i], e = offsets[i + 1];
    const uint32_t len = (uint32_t)(e - b);
    mx = len > mx ? len : mx;
    uint64_t img = 0;
    if (chunk < 0) {
      img = len;
    } else {
      const int64_t p0 = (int64_t)b + 8ll * chunk;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t p = p0 + j;
        img = (img << 8) | (p < e ? (uint64_t)data[p] : 0ull);
      }
    }
    const bool valid = validity == nullptr || get_bit(validity, bit_offset + i);
    if (!valid) img = 0;
    image[i] = asc ? img : ~img;
    if (null_image) null_image[i] = (uint64_t)((valid ? 0 : 1) ^ (asc ? 0 : 1));
  }
  if (max_len) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const unsigned int o = (unsigned int)__shfl_xor((int)mx, m, 64);
      mx = o > mx ? o : mx;
    }
    if (lane_id() == 0 && mx) atomicMax(max_len, mx);
  }
}

__global__ __launch_bounds__(kBlock) void k_sort_iota(uint32_t* __restrict__ idx, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) idx[i] = (uint32_t)i;
}

__global__ __launch_bounds__(kBlock) void k_sort_gather_u64(const uint64_t* __restrict__ src, const uint32_t* __restrict__ idx,
                                                           int64_t n, uint64_t* __restrict__ dst) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) dst[i] = src[idx[i]];
}

// histograms of all eight digits (order-independent: which passes are worth running)
__global__ __launch_bounds__(kBlock) void k_radix_hist8(const uint64_t* __restrict__ img, int64_t n,
                                                       unsigned long long* __restrict__ hist /* [8][256] */) {
  __shared__ uint32_t h[8 * 256];
  for (int i = threadIdx.x; i < 8 * 256; i += kBlock) h[i] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const uint64_t v = img[i];
#pragma unroll
    for (int d = 0; d < 8; ++d) atomicAdd(&h[d * 256 + (int)((v >> (8 * d)) & 255)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 8 * 256; i += kBlock)
    if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

// per-wave-tile counts of one digit: counts[digit][tile]
__global__ __launch_bounds__(kBlock) void k_radix_count(const uint64_t* __restrict__ img, int64_t n, int shift,
                                                       int64_t n_tiles, uint32_t* __restrict__ counts) {
  __shared__ uint32_t h[kBlock / 64][256];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int64_t tile = (int64_t)blockIdx.x * (kBlock / 64) + wave;
  for (int d = lane; d < 256; d += 64) h[wave][d] = 0;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (tile < n_tiles) {
    const int64_t base = tile * kSortWaveTile;
    for (int r = 0; r < kSortWaveTile / 64; ++r) {
      const int64_t i = base + (int64_t)r * 64 + lane;
      if (i < n) atomicAdd(&h[wave][(int)((img[i] >> shift) & 255)], 1u);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int d = lane; d < 256; d += 64) counts[(int64_t)d * n_tiles + tile] = ((volatile uint32_t*)h[wave])[d];
  }
}

// stable scatter of (image, index) pairs by one digit; offsets[digit][tile] = exclusive scan of counts
__global__ __launch_bounds__(kBlock) void k_radix_scatter(const uint64_t* __restrict__ img_in, const uint32_t* __restrict__ idx_in,
                                                         int64_t n, int shift, int64_t n_tiles,
                                                         const uint64_t* __restrict__ offsets, uint64_t* __restrict__ img_out,
                                                         uint32_t* __restrict__ idx_out) {
  __shared__ uint64_t cursor_lds[kBlock / 64][256];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int64_t tile = (int64_t)blockIdx.x * (kBlock / 64) + wave;
  if (tile >= n_tiles) return;
  volatile uint64_t* cursor = cursor_lds[wave];
  for (int d = lane; d < 256; d += 64) cursor[d] = offsets[(int64_t)d * n_tiles + tile];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const uint64_t lt = (1ull << lane) - 1ull;
  const int64_t base = tile * kSortWaveTile;
  for (int r = 0; r < kSortWaveTile / 64; ++r) {
    const int64_t i = base + (int64_t)r * 64 + lane;
    const bool valid = i < n;
    const uint64_t v = valid ? img_in[i] : 0ull;
    const uint32_t x = valid ? idx_in[i] : 0u;
    const uint32_t digit = (uint32_t)((v >> shift) & 255);
    uint64_t m = __ballot(valid);  // lanes with my digit (match-any with 8 ballots)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const uint64_t bs = __ballot((digit >> b) & 1u);
      m &= ((digit >> b) & 1u) ? bs : ~bs;
    }
    if (valid) {
      const uint64_t pos = cursor[digit] + (uint64_t)__popcll(m & lt);
      img_out[pos] = v;
      idx_out[pos] = x;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (valid && (m & lt) == 0) cursor[digit] = cursor[digit] + (uint64_t)__popcll(m);  // the group's first lane
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
}

// ---- top-k (ORDER BY ... LIMIT k): radix select on the first key's image ------------------------------------------------
// histogram of digit `shift / 8` over the elements whose higher digits equal `prefix` (mask = the bits above the digit)
__global__ __launch_bounds__(kBlock) void k_select_hist(const uint64_t* __restrict__ img, int64_t n, uint64_t prefix, uint64_t mask,
                                                       int shift, unsigned long long* __restrict__ hist /* [256] */) {
  __shared__ uint32_t h[256];
  for (int i = threadIdx.x; i < 256; i += kBlock) h[i] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const uint64_t v = img[i];
    if ((v & mask) == prefix) atomicAdd(&h[(int)((v >> shift) & 255)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += kBlock)
    if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

// bitmap of img <= threshold (+ counts per 4096-row tile, the layout launch_compact expects)
__global__ __launch_bounds__(kBlock) void k_select_mask(const uint64_t* __restrict__ img, int64_t n, uint64_t threshold,
                                                       uint64_t* __restrict__ mask_words, uint32_t* __restrict__ tile_counts) {
  __shared__ uint32_t wave_cnt[kBlock / 64];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int64_t n_words = (n + 63) >> 6;
  const int64_t n_tiles = (n + kTileRows - 1) / kTileRows;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    uint32_t cnt = 0;
    for (int i = 0; i < 16; ++i) {
      const int64_t w = tile * 64 + wave * 16 + i;
      const int64_t r = w * 64 + lane;
      const uint64_t word = __ballot(r < n && img[r < n ? r : 0] <= threshold);
      if (lane == 0 && w < n_words) mask_words[w] = word;
      cnt += (uint32_t)__popcll(word);
    }
    if (lane == 0) wave_cnt[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) tile_counts[tile] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
}

// ---- gather ----------------------------------------------------------------------------------------
// global row -> (batch << 32 | row in batch); starts[nb + 1] ascending
__global__ __launch_bounds__(kBlock) void k_sort_locate(const uint32_t* __restrict__ idx, int64_t n,
                                                       const uint64_t* __restrict__ starts, int nb, uint64_t* __restrict__ loc) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const uint64_t g = idx[i];
    int lo = 0, hi = nb;  // starts[lo] <= g < starts[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (starts[mid] <= g) lo = mid;
      else hi = mid;
    }
    loc[i] = ((uint64_t)lo << 32) | (g - starts[lo]);
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_gather_fixed(const void* const* __restrict__ bases, const uint64_t* __restrict__ loc,
                                                        int64_t n, T* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const uint64_t l = loc[i];
    out[i] = ((const T*)bases[l >> 32])[l & 0xFFFFFFFFull];
  }
}

// bits (validity or Boolean values) -> dense bitmap words; bases[b] may be null (= all set); null_count accumulates zeros
__global__ __launch_bounds__(kBlock) void k_gather_bits(const uint8_t* const* __restrict__ bases, const int64_t* __restrict__ bit_offsets,
                                                       const uint64_t* __restrict__ loc, int64_t n, uint64_t* __restrict__ out,
                                                       unsigned long long* __restrict__ zero_count) {
  const int64_t n_pad = (n + 63) & ~63ll;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_pad; i += (int64_t)gridDim.x * kBlock) {
    bool bit = false;
    if (i < n) {
      const uint64_t l = loc[i];
      const uint8_t* b = bases[l >> 32];
      bit = b == nullptr || get_bit(b, bit_offsets[l >> 32] + (int64_t)(l & 0xFFFFFFFFull));
    }
    const uint64_t w = __ballot(bit);
    if (lane_id() == 0) {
      out[i >> 6] = w;
      const int64_t here = n - i < 64 ? n - i : 64;
      const int zeros = (int)here - __popcll(w);
      if (zeros > 0 && zero_count) atomicAdd(zero_count, (unsigned long long)zeros);
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_gather_utf8_lens(const int32_t* const* __restrict__ offsets, const uint64_t* __restrict__ loc,
                                                            int64_t n, int32_t* __restrict__ lens) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const uint64_t l = loc[i];
    const int32_t* o = offsets[l >> 32] + (l & 0xFFFFFFFFull);
    lens[i] = o[1] - o[0];
  }
}

__global__ __launch_bounds__(kBlock) void k_gather_utf8_copy(const int32_t* const* __restrict__ offsets,
                                                            const uint8_t* const* __restrict__ data, const uint64_t* __restrict__ loc,
                                                            int64_t n, const int32_t* __restrict__ dst_offsets, uint8_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const uint64_t l = loc[i];
    const int32_t* o = offsets[l >> 32] + (l & 0xFFFFFFFFull);
    const uint8_t* src = data[l >> 32] + o[0];
    uint8_t* dst = out + dst_offsets[i];
    const int32_t len = o[1] - o[0];
    for (int32_t j = 0; j < len; ++j) dst[j] = src[j];
  }
}

// ---- launchers ---------------------------------------------------------------------------------------
static int sort_grid(int64_t n) { return stream_grid((n + kBlock - 1) / kBlock, 8); }

hipError_t launch_sort_image(const void* values, const uint8_t* validity, int64_t bit_offset, uint8_t dtype, int asc, int64_t n,
                             uint64_t* image, uint64_t* null_image, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_SORT, s, 0);
  hipLaunchKernelGGL(k_sort_image, dim3(sort_grid(n)), dim3(kBlock), 0, s, values, validity, bit_offset, dtype, asc, n, image, null_image);
  return hipGetLastError();
}

hipError_t launch_sort_image_utf8(const int32_t* offsets, const uint8_t* data, const uint8_t* validity, int64_t bit_offset, int chunk,
                                  int asc, int64_t n, uint64_t* image, uint64_t* null_image, uint32_t* max_len, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_SORT, s, 0);
  hipLaunchKernelGGL(k_sort_image_utf8, dim3(sort_grid(n)), dim3(kBlock), 0, s, offsets, data, validity, bit_offset, chunk, asc, n, image,
                     null_image, max_len);
  return hipGetLastError();
}

hipError_t launch_sort_iota(uint32_t* idx, int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_sort_iota, dim3(sort_grid(n)), dim3(kBlock), 0, s, idx, n);
  return hipGetLastError();
}

hipError_t launch_sort_gather_u64(const uint64_t* src, const uint32_t* idx, int64_t n, uint64_t* dst, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_SORT, s, 0);
  hipLaunchKernelGGL(k_sort_gather_u64, dim3(sort_grid(n)), dim3(kBlock), 0, s, src, idx, n, dst);
  return hipGetLastError();
}

hipError_t launch_radix_hist8(const uint64_t* img, int64_t n, uint64_t* hist, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_SORT, s, 0);
  const int grid = (int)std::min<int64_t>((n + kBlock * 16 - 1) / (kBlock * 16), 2048);
  hipLaunchKernelGGL(k_radix_hist8, dim3(grid), dim3(kBlock), 0, s, img, n, (unsigned long long*)hist);
  return hipGetLastError();
}

hipError_t launch_select_hist(const uint64_t* img, int64_t n, uint64_t prefix, uint64_t mask, int shift, uint64_t* hist, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_SORT, s, (double)n * 8);
  const int grid = (int)std::min<int64_t>((n + kBlock * 16 - 1) / (kBlock * 16), 2048);
  hipLaunchKernelGGL(k_select_hist, dim3(grid), dim3(kBlock), 0, s, img, n, prefix, mask, shift, (unsigned long long*)hist);
  return hipGetLastError();
}

hipError_t launch_select_mask(const uint64_t* img, int64_t n, uint64_t threshold, uint64_t* mask_words, uint32_t* tile_counts,
                              hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const int64_t tiles = (n + kTileRows - 1) / kTileRows;
  hipLaunchKernelGGL(k_select_mask, dim3(stream_grid(tiles, 8)), dim3(kBlock), 0, s, img, n, threshold, mask_words, tile_counts);
  return hipGetLastError();
}

int64_t radix_tiles(int64_t n) { return (n + kSortWaveTile - 1) / kSortWaveTile; }

hipError_t launch_radix_count(const uint64_t* img, int64_t n, int shift, uint32_t* counts, hipStream_t s) {
  const int64_t tiles = radix_tiles(n);
  if (tiles <= 0) return hipSuccess;
  Scope sc(KID_SORT, s, (double)n * 8);
  const int64_t blocks = (tiles + kBlock / 64 - 1) / (kBlock / 64);
  hipLaunchKernelGGL(k_radix_count, dim3((unsigned)blocks), dim3(kBlock), 0, s, img, n, shift, tiles, counts);
  return hipGetLastError();
}

hipError_t launch_radix_scatter(const uint64_t* img_in, const uint32_t* idx_in, int64_t n, int shift, const uint64_t* offsets,
                                uint64_t* img_out, uint32_t* idx_out, hipStream_t s) {
  const int64_t tiles = radix_tiles(n);
  if (tiles <= 0) return hipSuccess;
  Scope sc(KID_SORT, s, (double)n * 24);
  const int64_t blocks = (tiles + kBlock / 64 - 1) / (kBlock / 64);
  hipLaunchKernelGGL(k_radix_scatter, dim3((unsigned)blocks), dim3(kBlock), 0, s, img_in, idx_in, n, shift, tiles, offsets, img_out, idx_out);
  return hipGetLastError();
}

hipError_t launch_sort_locate(const uint32_t* idx, int64_t n, const uint64_t* starts, int nb, uint64_t* loc, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_sort_locate, dim3(sort_grid(n)), dim3(kBlock), 0, s, idx, n, starts, nb, loc);
  return hipGetLastError();
}

hipError_t launch_gather_fixed(const void* const* bases, const uint64_t* loc, int64_t n, int width, void* out, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_SORT, s, (double)n * (8 + 2 * width));
  const int grid = sort_grid(n);
  switch (width) {
    case 8: hipLaunchKernelGGL(k_gather_fixed<uint64_t>, dim3(grid), dim3(kBlock), 0, s, bases, loc, n, (uint64_t*)out); break;
    case 4: hipLaunchKernelGGL(k_gather_fixed<uint32_t>, dim3(grid), dim3(kBlock), 0, s, bases, loc, n, (uint32_t*)out); break;
    case 2: hipLaunchKernelGGL(k_gather_fixed<uint16_t>, dim3(grid), dim3(kBlock), 0, s, bases, loc, n, (uint16_t*)out); break;
    case 1: hipLaunchKernelGGL(k_gather_fixed<uint8_t>, dim3(grid), dim3(kBlock), 0, s, bases, loc, n, (uint8_t*)out); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_gather_bits(const uint8_t* const* bases, const int64_t* bit_offsets, const uint64_t* loc, int64_t n,
                              uint64_t* out, uint64_t* zero_count, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_gather_bits, dim3(sort_grid(n)), dim3(kBlock), 0, s, bases, bit_offsets, loc, n, out,
                     (unsigned long long*)zero_count);
  return hipGetLastError();
}

hipError_t launch_gather_utf8_lens(const int32_t* const* offsets, const uint64_t* loc, int64_t n, int32_t* lens, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_gather_utf8_lens, dim3(sort_grid(n)), dim3(kBlock), 0, s, offsets, loc, n, lens);
  return hipGetLastError();
}

hipError_t launch_gather_utf8_copy(const int32_t* const* offsets, const uint8_t* const* data, const uint64_t* loc, int64_t n,
                                   const int32_t* dst_offsets, uint8_t* out, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_gather_utf8_copy, dim3(sort_grid(n)), dim3(kBlock), 0, s, offsets, data, loc, n, dst_offsets, out);
  return hipGetLastError();
}

}  // namespace dfx

// dfx_k_partition_ws_inl.hpp -- pass 1 of the partitioned GROUP BY, WAVE-SPECIALISED flavour (PTF_WS).
//
// k_partition_ring (dfx_k_partition_inl.hpp) lets every one of a workgroup's 16 waves both scan and route.  Its
// decomposition says where the time goes when a fifth of the rows pass: the scan skeleton alone
// streams at 6.67 TB/s, routing 20 % of the rows adds 45 % per scanned row although its traffic explains 17 % -- the LDS
// ring protocol (fill / commit atomics, generation checks, waiting for a busy chunk slot, flush jobs) runs on the waves
// that own the outstanding column loads, and every wave carries four inlined copies of it (17 850 instructions, 106
// spilled SGPRs in the round-2 build).  Here the two jobs belong to different waves of the workgroup:
//   * NS SCANNER waves only load, compare and ballot-compact the passing rows {key (32 bits: narrow keys), operand} into
//     their own single-producer / single-consumer LDS queue: ~30 vector instructions per 64-row group, nothing in their
//     loop ever waits for a chunk slot or issues a global store;
//   * 16 - NS ROUTER waves poll the queues of "their" scanners (scanner s belongs to router s mod NR), take 64 rows at a
//     time at full lane utilisation -- two batches side by side when a queue holds 128 -- and run the ring protocol (ring_route2, one inlined copy): hash -> partition -> fill
//     atomic -> ring -> commit -> cooperative 192-byte flushes.
// A scanner stalls only when its queue is full (the router then has >= 64 rows to take, so it cannot be a deadlock);
// a router waits only for rows or for another router's flush of a LOWER chunk -- the argument of the ring kernel.
// Narrow 12-byte rows in the large-chunk geometry only (PTF_NARROW | PTF_CHUNK16 -- round 6: LINE chunks, dfx_device.hpp --: one routed value, keys below 2^32, no hot-key
// pairs); rows the narrow form cannot carry (wide keys, the claim sentinel, reserved images, region overflow) take the
// spill list exactly as in the ring kernel.  Same scratch layout, counts and padding: pass 2 cannot tell the flavours apart.
#pragma once
#include <type_traits>

#include "dfx_k_partition_inl.hpp"

namespace dfx {

constexpr int kWsQueueRows = 256;  // per scanner wave (power of two): 3 KB {u32 key, u64 operand}
constexpr int kWsCH = kNarrowChunkRows, kWsRP = kNarrowRingRows, kWsNCH = kWsRP / kWsCH;  // (dfx_device.hpp: LINE chunks of ten rows, three slots)

struct WsCtl {  // one per queue (= per router wave)
  uint32_t tail;  // rows produced (written by the queue's scanner)
  uint32_t head;  // rows consumed (written by its router)
  uint32_t done;  // the scanner has published its last row
  uint32_t pad;
};

#ifdef DFX_PARTITION_MAIN_TU
size_t partition_ws_bytes(uint32_t n_parts, int ns, int nv) {  // nv: routed operands per row (2: PTF_PAIR -- a second operand plane per queue)
  const int nq = kRingBlock / 64 - ns;  // one queue per ROUTER wave (a scanner feeds (16 - ns) / ns of them in turn)
  return (size_t)n_parts * kNarrowRingSlots * kNarrowSlotBytes + (size_t)(kRingBlock / 64) * 64 * 8 /* jobs */ + (size_t)n_parts * 4 * (1 + 2 * kWsNCH) +
         (size_t)nq * kWsQueueRows * (4 + 8 * (size_t)nv) + (size_t)nq * sizeof(WsCtl) + 64;
}
#else
size_t partition_ws_bytes(uint32_t n_parts, int ns, int nv);
#endif

// rows the narrow routed form cannot carry: a key >= 2^32 (the claim sentinel i64::MIN among them).  No row of a stream whose
// calibration slice saw narrow keys only takes this path unless the data changes under it.  They go to the spill list and
// nowhere else: the replay (launch_merge_rows -> table_apply) knows the sentinel key's slot, CTRL_WIDE_KEYS makes the host
// leave narrow mode, and the scan loop carries a few dozen instructions for them instead of the accumulator algebra
// (sentinel_apply inlined four times per loop was 600 instructions and most of this kernel's scalar-register pressure).
template <int NV = 1>
DEV void ws_slow_rows(const DevTable& T, const DevPartition& PT, const DevRows& spill, bool slow, uint64_t key0, uint64_t val0, uint64_t val1 = 0) {
  uint64_t xf = 0;
#pragma unroll
  for (int a = 0; a < kMaxAggs; ++a) xf |= (uint64_t)T.val_xform[a] << (8 * a);
  const uint32_t mode = NV == 1 ? ((PT.flags & PTF_SHARED) ? 1u : 0u) : ((PT.flags & PTF_PLANES) ? 3u : 2u);
  ws_slow_rows_call(T.ctrl, spill.words, spill.capacity, mode, (uint32_t)T.na, xf, PT.pair_ops, true, slow, key0, val0, val1);
}

// NV = 2 (PTF_PAIR): two aggregates of different operands -- the scanners evaluate both arguments, the queues carry a second
// operand plane, the routers write 20-byte rows into LINE chunks of six (dfx_device.hpp: kPair*); everything else is the same code.
template <typename POL, int NS, int FORM, int NV = 1>
__global__ __launch_bounds__(kRingBlock) void k_partition_ws(const DevProgram P, const DevFastPlan F, const DevColumns C,
                                                            const DevAggPlan plan, const DevTable T,
                                                            const DevPartition PT, const DevRows spill, const int64_t n) {
  typedef typename POL::COLV COLV;
  constexpr int U = POL::U;
  constexpr int NWAVES = kRingBlock / 64;
  constexpr int NR = NWAVES - NS;             // router waves
  // One single-producer / single-consumer queue per ROUTER.  A scanner owns QPS = NR / NS of them and hands its row groups to
  // them in turn: with 8 + 8 waves (selective scans) every scanner has its router; with 4 + 12 (dense scans, round 5) three
  // routers share the rows of one scanner -- routing, a chain of LDS round trips per batch, is what a dense scan has most of.
  constexpr int QPS = NR / NS;                // queues per scanner
  static_assert(NS >= 1 && NR >= 1 && NR % NS == 0, "every router has a queue, every scanner the same number of them");
  static_assert(NV == 1 || NV == 2, "one routed operand, or the pair");
  // chunk geometry: rows per chunk (= per 128-byte line), rows per ring, dwords per row
  constexpr int CH = NV == 2 ? kPairChunkRows : kWsCH, RP = NV == 2 ? kPairRingRows : kWsRP, DW = NV == 2 ? kPairRowDwords : 3;
  static_assert(NV == 1 || !kNarrowLine || RP / CH == kWsNCH, "three slots either way");
  extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
  RingLds L;
  L.ring = lds;
  const size_t ring_words = ring_dwords12<CH, RP, 1, DW>(PT.n_parts) / 2;  // 12-byte rows (LINE chunks: 128-byte slots)
  L.queue = nullptr;                                             // (the ring kernel's wave queues: not used here)
  L.jobs = (uint32_t*)(L.ring + ring_words);
  L.fill = L.jobs + NWAVES * 64 * 2;
  L.commit = L.fill + PT.n_parts;
  L.gen = L.commit + (size_t)PT.n_parts * kWsNCH;
  uint32_t* const qkeys = L.gen + (size_t)PT.n_parts * kWsNCH;                   // [NR][kWsQueueRows]
  WsCtl* const ctl = (WsCtl*)(qkeys + (size_t)NR * kWsQueueRows);                // [NR]
  // the operand planes behind everything else, 8-byte aligned (as an offset from `lds`: keeps the LDS address space)
  const size_t qv_word0 = ring_words + ((size_t)(NWAVES * 64 * 2 + PT.n_parts * (1 + 2 * kWsNCH) + NR * kWsQueueRows) * 4 + (size_t)NR * sizeof(WsCtl) + 7) / 8;
  uint64_t* const qvals = lds + qv_word0;                                        // [NR][kWsQueueRows]
  uint64_t* const qvals2 = qvals + (size_t)NR * kWsQueueRows;                    // NV == 2: the second operand, same indexing
  const int lane = lane_id();
  const int wave = (int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t producer = blockIdx.x;
  // fill, commit, gen exactly as k_partition_ring initialises them (PTF_RESUME: chunk numbering goes on from counts[])
  for (uint32_t p = threadIdx.x; p < PT.n_parts; p += kRingBlock) {
    const uint32_t f0 = (PT.flags & PTF_RESUME) ? PT.counts[(uint64_t)p * PT.n_producers + producer] : 0u;
    const uint32_t c0 = f0 / (uint32_t)CH;
    L.fill[p] = f0;
#pragma unroll
    for (int sl = 0; sl < kWsNCH; ++sl) {
      L.commit[p * kWsNCH + sl] = 0;
      L.gen[p * kWsNCH + sl] = (c0 + (uint32_t)(kWsNCH - 1 - sl)) / (uint32_t)kWsNCH;
    }
  }
  if (threadIdx.x < (unsigned)NR) {
    ctl[threadIdx.x].tail = 0;
    ctl[threadIdx.x].head = 0;
    ctl[threadIdx.x].done = 0;
  }
  __syncthreads();
  uint32_t err = 0;
  if (wave < NS) {
    // ---------------------------------------------------------------- scanner -----------------------------------
    // this scanner's queues: q0 .. q0 + QPS - 1.  Lane j of v_tail / v_head holds queue q0 + j's produced / last-seen-consumed count
    // (read with v_readlane at the wave-uniform index `cq` -- the queue the next row group goes to --, written by compare-and-select)
    const int q0 = wave * QPS;
    uint32_t v_tail = 0, v_head = 0;
    int cq = 0;
    uint64_t passed = 0;  // wave-uniform
    const int64_t n_groups = (n + 63) >> 6;
    const int64_t wave_global = (int64_t)blockIdx.x * NS + wave;
    const int64_t n_waves = (int64_t)gridDim.x * NS;
    // The scan loop.  FORM (StaticPolicy::pass_form) makes the comparison operators compile-time constants -- one v_cmp per term
    // instead of three and no scalar selects; 0: the policy's run-time form.  It is a template parameter of the KERNEL, chosen
    // by the host (launch_partition_ws): round 3 switched between five inlined copies of this loop inside one kernel -- 26 000
    // instructions, 212 spilled scalar registers -- for a decision that is the same for every wave of every launch of a query.
    // Software pipeline, one trip deep (as in k_partition_ring: deeper was measured no faster).
    {
      typename POL::PREP prep;  // (PlanPolicy: the plan words in vector registers; empty otherwise)
      POL::prepare(P, F, prep);
      COLV ncol[U];
      uint32_t ncv[U];
      load_trip<POL>(P, C, wave_global * U, wave_global * U < n_groups, n, lane, ncol, ncv);
      for (int64_t w0 = wave_global * U; w0 < n_groups; w0 += n_waves * U) {
        COLV col[U];
        uint32_t cv[U];
        FOR_U {
          col[u] = ncol[u];
          cv[u] = ncv[u];
        }
        load_trip<POL>(P, C, w0 + n_waves * U, w0 + n_waves * U < n_groups, n, lane, ncol, ncv);
        // One body for whole trips and one for the trip that holds the table's last rows (FULL = false: the lanes past the end
        // are masked; load_trip let them re-read the last row).  Lane predicates live in SCALAR masks between the steps -- a
        // bool that is live across a branch is materialised in a vector register and compared again (v_cndmask + v_cmp per
        // use) --, and the queue positions are scalars as well (readfirstlane of the LDS words).
        auto trip_body = [&](auto full_tag) {
          constexpr bool FULL = decltype(full_tag)::value;
          FOR_U {  // (unrolled, like the ring kernel's scan: a run-time loop over the U banks costs a scalar branch chain per group)
            const COLV cur = col[u];
            const uint32_t curv = cv[u];
            bool inb = true;
            if constexpr (!FULL) inb = (w0 + u) * 64 + lane < n;
            u64x16 reg;
            uint32_t rv = 0;
            POL::eval(P, F, cur, curv, reg, rv, inb, err, prep);
            uint64_t pm = POL::template pass_mask<FORM>(P, F, plan.pred, cur, curv, reg, rv, prep);  // (evaluated for every lane)
            if constexpr (!FULL) pm &= __ballot(inb);
            const uint64_t key = POL::key(P, F, plan.key[0], 0, cur, curv, reg, rv);
            uint64_t v;
            bool valid;
            POL::arg(P, F, plan.arg[0], 0, cur, curv, reg, rv, v, valid);
            // (PTF_SHARED: the row carries the aggregates' common RAW operand -- pass 2 applies every aggregate's own transform)
            const bool raw_ops = NV == 1 ? (PT.flags & PTF_SHARED) != 0 : (PT.flags & PTF_PLANES) != 0;  // (pass 2 applies the transforms)
            const uint64_t val = transform_value(raw_ops ? (uint8_t)VT_RAW : POL::xform(T, 0), v, valid);
            uint64_t val1 = 0;
            if constexpr (NV == 2) {  // operand 1 = the argument of accumulator pair_arg1 (1 when there are two aggregates)
              uint64_t v1;
              bool valid1;
              POL::arg_slot(F, PT.pair_slot1, reg, rv, v1, valid1);
              val1 = transform_value(raw_ops ? (uint8_t)VT_RAW : (uint8_t)PT.pair_xf1, v1, valid1);
            }
            passed += (uint64_t)__popcll(pm);
            // dense split (QPS > 1): the SCANNERS hash -- four waves with little else to do, while the twelve routers' chain of
            // LDS round trips is what bounds the launch -- and the queue carries the row's 32-bit image instead of its key
            uint32_t qword = (uint32_t)key;
            uint64_t sm = pm & __ballot((key >> 32) != 0);  // no 32-bit form (wide key, or the claim sentinel)
            if constexpr (QPS > 1) {
              uint64_t k1[1] = {key};
              qword = (uint32_t)(hash_keys<1>(k1) >> 32);
              sm |= pm & __ballot(qword >= kTagForeign);  // ... or one of the two reserved images
            }
            if (sm != 0) {
              ws_slow_rows<NV>(T, PT, spill, lane_of_mask(sm), key, val, val1);
              pm &= ~sm;
            }
            const uint32_t c = (uint32_t)__popcll(pm);
            if (c != 0) {
              uint32_t* const qk = qkeys + (size_t)(q0 + cq) * kWsQueueRows;
              uint64_t* const qv = qvals + (size_t)(q0 + cq) * kWsQueueRows;
              WsCtl* const my = ctl + (q0 + cq);
              uint32_t tail = (uint32_t)__builtin_amdgcn_readlane((int)v_tail, cq);
              uint32_t head_c = (uint32_t)__builtin_amdgcn_readlane((int)v_head, cq);
              // room for c rows?  (the router publishes its position after every batch of 64 it takes)
              uint32_t spins = 0;
              // The router only sees the PUBLISHED tail and takes whole batches of 64: rows parked since the last publication
              // (up to U - 1 groups of this trip) are invisible to it, so a locally dense stretch -- more than ~3/4 of a
              // 256-row window passing -- could fill the queue with rows nobody may take yet.  Publish before waiting: with
              // every parked row visible the router leaves fewer than 64 behind, and 64 + c <= 256 always fits.
              if (tail + c - head_c > (uint32_t)kWsQueueRows && lane == 0) __hip_atomic_store(&my->tail, tail, __ATOMIC_RELEASE, WG_SCOPE);
              while (tail + c - head_c > (uint32_t)kWsQueueRows) {
                head_c = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&my->head, __ATOMIC_ACQUIRE, WG_SCOPE));
                if (tail + c - head_c <= (uint32_t)kWsQueueRows) break;
                if (++spins > (1u << 22)) {  // cannot happen (a full queue always has 64 rows for its router); never hang the device
                  err |= 4u;
                  break;
                }
                __builtin_amdgcn_s_sleep(1);
              }
              if (lane_of_mask(pm)) {
                const uint32_t at = (tail + mbcnt64(pm)) & (uint32_t)(kWsQueueRows - 1);
                qk[at] = qword;
                qv[at] = val;
                if constexpr (NV == 2) qvals2[(size_t)(q0 + cq) * kWsQueueRows + at] = val1;
              }
              tail += c;
              v_tail = lane == cq ? tail : v_tail;  // (v_writelane by compare-and-select)
              v_head = lane == cq ? head_c : v_head;
              if constexpr (QPS > 1) {
                // the next row group goes to the next router.  With several routers per scanner a group's rows are published at
                // once: a router that waits for the end of the trip idles while its neighbours work
                if (lane == 0) __hip_atomic_store(&my->tail, tail, __ATOMIC_RELEASE, WG_SCOPE);
                cq = cq + 1 == QPS ? 0 : cq + 1;
              }
            }
          }
        };
        if ((w0 + U) * 64 <= n) trip_body(std::true_type{}); else trip_body(std::false_type{});
        if constexpr (QPS == 1) {
          if (lane == 0) __hip_atomic_store(&ctl[q0].tail, v_tail, __ATOMIC_RELEASE, WG_SCOPE);  // once per trip: the rows above are visible first
        }
      }
    }
    if (lane < QPS) {  // lane j: queue q0 + j
      __hip_atomic_store(&ctl[q0 + lane].tail, v_tail, __ATOMIC_RELEASE, WG_SCOPE);
      __hip_atomic_store(&ctl[q0 + lane].done, 1u, __ATOMIC_RELEASE, WG_SCOPE);
    }
    if (lane == 0) stat_add(T, STAT_PASSED, passed);
  } else {
    // ---------------------------------------------------------------- router ------------------------------------
    // router r drains queue r (its scanner: r / QPS)
    const int q = wave - NS;
    WsCtl* const sc = ctl + q;
    bool fin = false;
    uint32_t idle = 0;
    while (!fin) {
      const uint32_t head = sc->head;  // (only this wave writes it)
      uint32_t tail = __hip_atomic_load(&sc->tail, __ATOMIC_ACQUIRE, WG_SCOPE);
      uint32_t avail = tail - head;
      uint32_t take = avail >= 128u ? 128u : avail >= 64u ? 64u : 0u;  // whole batches of 64, two at a time when they are there
      if (take == 0 && __hip_atomic_load(&sc->done, __ATOMIC_ACQUIRE, WG_SCOPE) != 0u) {
        tail = __hip_atomic_load(&sc->tail, __ATOMIC_ACQUIRE, WG_SCOPE);  // the final count was published before `done`
        avail = tail - head;
        take = avail < 128u ? avail : 128u;
        if (avail == 0) fin = true;
      }
      if (take == 0) {
        if (fin) break;
        if (++idle > (1u << 24)) {  // a scanner that never finishes: cannot happen; never hang the device
          err |= 4u;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
        continue;
      }
      idle = 0;
      bool have[2];
      uint64_t k2[2], v2[2], h2[2], w2[2] = {0, 0};
#pragma unroll
      for (int b = 0; b < 2; ++b) {  // (both batches' queue reads in flight together)
        have[b] = (uint32_t)(b * 64 + lane) < take;
        const uint32_t at = (head + (uint32_t)(b * 64 + lane)) & (uint32_t)(kWsQueueRows - 1);
        k2[b] = have[b] ? (uint64_t)qkeys[(size_t)q * kWsQueueRows + at] : 0ull;
        v2[b] = have[b] ? qvals[(size_t)q * kWsQueueRows + at] : 0ull;
        if constexpr (NV == 2) w2[b] = have[b] ? qvals2[(size_t)q * kWsQueueRows + at] : 0ull;
      }
      // the slots are free again as soon as the rows sit in registers
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (lane == 0) __hip_atomic_store(&sc->head, head + take, __ATOMIC_RELEASE, WG_SCOPE);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if constexpr (QPS > 1) {
          h2[b] = k2[b] << 32;  // (the queue word IS the image)
        } else {
          uint64_t k1[1] = {k2[b]};
          h2[b] = hash_keys<1>(k1);
        }
      }
      ring_route2<CH, RP, 1, (QPS > 1), DW>(T, PT, spill, L, producer, have, k2, v2, h2, err, w2);
    }
  }
  __syncthreads();
  // partial chunks + region counts: as in k_partition_ring (a partial chunk is padded to a whole one with kTagEmpty rows)
  uint32_t max_fill = 0;
  for (uint32_t p = threadIdx.x; p < PT.n_parts; p += kRingBlock) {
    uint32_t f = L.fill[p];
    if (f > PT.cap_rows) f = PT.cap_rows;
    const uint32_t c = f / (uint32_t)CH;
    const uint32_t rem = f % (uint32_t)CH;
    for (uint32_t rr = 0; rr < rem; ++rr) {
      const uint32_t* s32 = (const uint32_t*)L.ring + ring_dword12<CH, RP, 1, DW>(p, c % kWsNCH, rr);
      uint32_t* o32 = region_row12g<CH, 1, DW>(PT, p, producer, c * CH + rr);
      for (int w = 0; w < DW; ++w) o32[w] = s32[w];
    }
    if (rem != 0) {
      for (uint32_t rr = rem; rr < (uint32_t)CH; ++rr) {
        uint32_t* o32 = region_row12g<CH, 1, DW>(PT, p, producer, c * CH + rr);
        for (int w = 0; w < DW; ++w) o32[w] = w == (NV == 2 ? 2 : 0) ? kTagEmpty : 0u;  // (the image dword: kTagEmpty = not a row)
      }
      f = (c + 1) * CH;
    }
    PT.counts[(uint64_t)p * PT.n_producers + producer] = f;
    max_fill = f > max_fill ? f : max_fill;
  }
  publish_max_fill(T, max_fill);
  if (err) atomicOr(&T.ctrl[CTRL_ERROR], err);
  snapshot_ctrl_if_last(T, PT);
}

// DevPartition::ws_scanners: the split.  Two are instantiated: eight scanner waves + eight routers with the policy's own U
// (selective scans: the table in DESIGN.md section 4 -- 6 / 8 / 10 / 12 scanners and U = 8 measured in round 3), and four scanners
// + twelve routers with the dense policy POLD (round 5: scans that route more than half of their rows -- three routers per
// scanner, eight row groups per trip where the registers allow so that four waves keep the loads in flight).  Any other
// non-zero value runs eight.  The comparison form is the host's choice: one kernel per form the signature's two-term
// predicates can take (lower bound > / >=, upper bound < / <=), the run-time form for everything else.
constexpr uint32_t kWsDefault = 8, kWsDense = 4;
template <typename POLN, typename POLD>
void launch_partition_ws(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C, const DevAggPlan& plan, const DevTable& T,
                         const DevPartition& PT, const DevRows& spill, int64_t n, size_t lds_bytes, hipStream_t s) {
  const int grid = (int)PT.n_producers;
  const bool dense = PT.ws_scanners == kWsDense;
#define DFX_WS_LAUNCH(FORM_)                                                                                                             \
  do {                                                                                                                                   \
    if (dense)                                                                                                                           \
      hipLaunchKernelGGL((k_partition_ws<POLD, (int)kWsDense, FORM_>), dim3(grid), dim3(kRingBlock), lds_bytes, s, P, fast, C, plan, T, PT, spill, n);   \
    else                                                                                                                                 \
      hipLaunchKernelGGL((k_partition_ws<POLN, (int)kWsDefault, FORM_>), dim3(grid), dim3(kRingBlock), lds_bytes, s, P, fast, C, plan, T, PT, spill, n); \
  } while (0)
  if constexpr (POLN::kIsStatic && POLN::kPredTerms == 2) {
    switch (POLN::form_of(fast)) {
      case 4 | (1 << 3): DFX_WS_LAUNCH(4 | (1 << 3)); return;  // x >  a AND x <  b
      case 6 | (1 << 3): DFX_WS_LAUNCH(6 | (1 << 3)); return;  // x >= a AND x <  b
      case 4 | (3 << 3): DFX_WS_LAUNCH(4 | (3 << 3)); return;  // x >  a AND x <= b
      case 6 | (3 << 3): DFX_WS_LAUNCH(6 | (3 << 3)); return;  // x >= a AND x <= b
      default: break;
    }
  }
  DFX_WS_LAUNCH(0);
#undef DFX_WS_LAUNCH
}

// one pass-1 variant = one translation unit: the ring / sorted / direct kernels of the policy plus its wave-specialised kernel
// (POLW: the one-value policy of the wave-specialised kernel -- the scanners can afford more row groups per trip than the
// ring kernels, whose routing state competes for the same 128 registers)
// POLD: the one-value policy of the DENSE wave-specialised split (four scanners: eight row groups per trip)
#define DFX_PARTITION_VARIANT_WS(ID, POL, POLS, POLN, POLW, POLD)                                                          \
  void launch_partition_variant##ID(DFX_PARTITION_VARIANT_ARGS) {                                                          \
    if (PT.flags & PTF_WS)                                                                                                 \
      launch_partition_ws<POLW, POLD>(P, fast, C, plan, T, PT, spill, n, lds_bytes, s);                                    \
    else                                                                                                                   \
      launch_partition_pol<POL, POLS, POLN>(P, fast, C, plan, T, PT, spill, n, lds_bytes, s);                              \
  }


// ... and, for the policies that can evaluate two arguments (POLP: slot look-ups, PlanPolicyN), the PAIR flavour (PTF_PAIR): always
// eight scanners + eight routers (twelve routers' queues with two operand planes do not fit the LDS), the run-time comparison form
#define DFX_PARTITION_VARIANT_WS_PAIR(ID, POL, POLS, POLN, POLW, POLD, POLP)                                               \
  void launch_partition_variant##ID(DFX_PARTITION_VARIANT_ARGS) {                                                          \
    if ((PT.flags & PTF_WS) && (PT.flags & PTF_PAIR))                                                                      \
      hipLaunchKernelGGL((k_partition_ws<POLP, (int)kWsDefault, 0, 2>), dim3((int)PT.n_producers), dim3(kRingBlock), lds_bytes, s, P, fast, C, plan, T, PT, spill, n); \
    else if (PT.flags & PTF_WS)                                                                                            \
      launch_partition_ws<POLW, POLD>(P, fast, C, plan, T, PT, spill, n, lds_bytes, s);                                    \
    else                                                                                                                   \
      launch_partition_pol<POL, POLS, POLN>(P, fast, C, plan, T, PT, spill, n, lds_bytes, s);                              \
  }

}  // namespace dfx

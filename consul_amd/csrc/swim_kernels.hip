// swim_kernels.hip — hand-written gfx950 kernels for the memberlist/serf SWIM hot path.
//
// One tick = four launches.  Everything is integer/byte work bounded by HBM bandwidth,
// random-access sector traffic and atomic throughput; there is no dense contraction, hence no MFMA.
//
//   k_begin    fused, role by block range:
//                expire   suspicion timers that ran out         -> dead{} verdicts in the node's own inbox
//                pending  indirect-ping stage of probes whose direct ping failed ProbeTimeout ago
//                probe    probe()/probeNode for the probe-due set -> suspect{} records, slot requests,
//                         piggy-back orders for the ping and the ack (sendMsg)
//                gossip   kRandomNodes + GetBroadcasts per peer   -> edge lists bucketed by shard
//   k_deliver  edge list -> per-node inbox rows (one returning atomic + one 16 B store per record)
//   k_resolve  per observer: canonical order, aliveNode/suspectNode/deadNode/handleUserEvent
//   k_census_finish  per dirty subject: how the live observers see it; first-suspect/first-dead/all-dead stamps, trace row, list recycling, tick++
// SWIM_F_UNBOUNDED_QUEUE (the queue implied by the dense pair store, at the end of this file): k_gossip_iq in the gossip role's place, k_piggy_iq between
// k_deliver and k_resolve; pooled inbox rows: k_inbox_claim / k_inbox_file behind k_deliver.
#include "swim_device.h"

#define NONE 0xFFFFFFFFu

__device__ __forceinline__ uint32_t sw_lane() { return __lane_id(); }

// 16-byte loads / stores through a pointer KNOWN to be global memory (a pointer that comes out of a table in memory has an unknown
// address space: the compiler emits flat_* instructions, which also wait on the LDS counter).  ext_vector_type: HIP's uint4 is a
// class, whose assignment operators do not take address-space-qualified objects.
typedef uint32_t sw_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_global_u4(const uint4* p) {
  const sw_u32x4 v = *(const __attribute__((address_space(1))) sw_u32x4*)p;
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_global_u4(uint4* p, uint4 v) {
  sw_u32x4 q; q.x = v.x; q.y = v.y; q.z = v.z; q.w = v.w;
  *(__attribute__((address_space(1))) sw_u32x4*)p = q;
}

// ---- small accessors ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t h_leaving(uint32_t w) { return w & 0xFFu; }
__device__ __forceinline__ uint32_t h_qlen(uint32_t w) { return (w >> 8) & 0xFFu; }
__device__ __forceinline__ uint32_t h_evqlen(uint32_t w) { return (w >> 16) & 0xFFu; }
__device__ __forceinline__ uint32_t h_pack(uint32_t lv, uint32_t ql, uint32_t eq) { return lv | (ql << 8) | (eq << 16); }
__device__ __forceinline__ uint32_t p_epoch(uint32_t w) { return w >> 16; }
__device__ __forceinline__ uint32_t p_aw(uint32_t w) { return (w >> 8) & 0xFFu; }
__device__ __forceinline__ uint32_t p_stage(uint32_t w) { return (w >> 6) & 3u; }
__device__ __forceinline__ uint32_t p_nackm(uint32_t w) { return w & 0x3Fu; }
__device__ __forceinline__ uint32_t p_pack(uint32_t ep, uint32_t aw, uint32_t st, uint32_t nm) {
  return (ep << 16) | (aw << 8) | (st << 6) | (nm & 0x3Fu);
}
__device__ __forceinline__ uint32_t m_type(uint32_t meta) { return meta >> 30; }
__device__ __forceinline__ uint32_t m_tr(uint32_t meta) { return (meta >> 22) & 0xFFu; }
__device__ __forceinline__ uint32_t m_seq(uint32_t meta) { return meta & 0x3FFFFFu; }
__device__ __forceinline__ uint32_t m_pack(uint32_t type, uint32_t tr, uint32_t seq) {
  return (type << 30) | (tr << 22) | (seq & 0x3FFFFFu);
}

// The kernel argument struct lives in SGPRs / the constant cache.  Indexing one of its member arrays with a per-lane
// value makes the compiler copy the WHOLE struct to scratch memory and read every field from there (seen in the
// fan-out > 4 and sharded variants: ~1 KB of scratch per lane, 300+ scratch loads) — so: select chains for the small
// tables, and tables in global memory (out_tab, out_cap_tab) for the per-shard lists.
// (by value: binding a reference to a member array would itself force the struct into memory)
__device__ __forceinline__ uint32_t sel4v(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t i) {
  return (i & 2u) ? ((i & 1u) ? a3 : a2) : ((i & 1u) ? a1 : a0);
}
#define sel4(a, i) sel4v((a)[0], (a)[1], (a)[2], (a)[3], (i))
#define sel8(a, i) ((((i) & 4u) ? sel4v((a)[4], (a)[5], (a)[6], (a)[7], (i)) : sel4v((a)[0], (a)[1], (a)[2], (a)[3], (i))))
__device__ __forceinline__ uint64_t seed_of(DevRef D, uint32_t r) { return D.seed + r; }
__device__ __forceinline__ uint32_t now_ms(DevRef D, uint32_t t) { return t * D.quantum_ms; }


// ---- a node's header, its view metadata, slot j of its memberlist queue (three arrays; one 64-byte record per node was measured twice, rounds 4 and 5: nothing)
#define HDR(l) D.hdr[l]
#define VMETA(l) D.vmeta[l]
#define QENT(j, l) D.q[(size_t)(j) * NL + (l)]

// ---- an observer's explicit views (layout: swim_device.h) ------------------------------------------
__device__ __forceinline__ uint32_t vt_home(DevRef D, uint32_t x) { return (x * 0x9E3779B1u) >> D.vt_shift; }
__device__ __forceinline__ uint32_t vw_nconf(uint32_t w) { return w & 7u; }
__device__ __forceinline__ uint32_t vw_conf0(uint32_t w) { return w >> 4; }
__device__ __forceinline__ uint32_t vw_leaving(uint32_t w) { return (w >> 3) & 1u; }
__device__ __forceinline__ uint32_t vw_pack(uint32_t conf0, uint32_t nconf, uint32_t leaving) { return (conf0 << 4) | (leaving << 3) | nconf; }
// lane l's explicit view of subject x: its slot (entry in `e`), or NONE with `free_slot` = where it would go.
// `first` = the home slot's entry when the caller fetched it already (together with other loads).
__device__ __forceinline__ uint32_t vt_probe(DevRef D, size_t l, uint32_t x, uint4 first, uint4& e, uint32_t& free_slot) {
  const size_t NL = (size_t)D.R * D.nloc; const uint32_t m = D.VT - 1;
  uint32_t s = vt_home(D, x);
  e = first;
  for (uint32_t i = 0; i < D.VT; i++) {
    if (e.x == x) return s;
    if (e.x == VT_EMPTY) { free_slot = s; return NONE; }
    s = (s + 1) & m;
    e = D.vt[(size_t)s * NL + l];
  }
  atomicOr(D.err, SW_ERR_VIEW_CORRUPT); free_slot = NONE;
  return NONE;
}
__device__ __forceinline__ uint32_t vt_find(DevRef D, size_t l, uint32_t x, uint4& e, uint32_t& free_slot) {
  return vt_probe(D, l, x, D.vt[(size_t)vt_home(D, x) * ((size_t)D.R * D.nloc) + l], e, free_slot);
}
// ---- the dense pair store (swim_device.h): pair (row of the subject, observer) --------------------------------------------
// Layout (round 5): [replica][group of 64 observers][row][64] — the pairs of 64 neighbouring observers with ALL rows are one contiguous region
// (M x 256 bytes per plane: 3-7 MB at config #4's sizes).  k_resolve<MASS>'s wave is such a group, and its lanes look at 64 DIFFERENT rows at a
// time: with rows outermost ([row][observer], rounds 3-4) those were 64 addresses a megabyte apart — 64 pages, 64 TLB misses per load, ~6 us a
// round trip and 30 us per message iteration (profiles/r05_config4_mass_phase_clock.txt).  A row's 256-observer tile is four 256-byte runs now
// (k_expire_mass, fold and census scans: a wave still reads one run), an observer's column 256-byte steps instead of megabyte steps.
__device__ __forceinline__ size_t m_idx(DevRef D, uint32_t r, uint32_t row, uint32_t k) {
  return ((((size_t)r * ((D.nloc + 63u) >> 6) + (k >> 6)) * D.M + row) << 6) + (k & 63u);
}
// the queue word of pair (row, lane k) — SWIM_F_UNBOUNDED_QUEUE, swim_device.h: [replica][64 observers][256 rows][observer][row]
__device__ __forceinline__ size_t e_idx(DevRef D, uint32_t r, uint32_t row, uint32_t k) {
  return ((((size_t)r * ((D.nloc + 63u) >> 6) + (k >> 6)) * D.MB + (row / SW_IQ_RB)) * 64u + (k & 63u)) * SW_IQ_RB + (row % SW_IQ_RB);
}
// ... and where lane k's column starts: row block rb of it is the SW_IQ_RB words at + rb * 64 * SW_IQ_RB
__device__ __forceinline__ size_t e_col(DevRef D, uint32_t r, uint32_t k) {
  return ((((size_t)r * ((D.nloc + 63u) >> 6) + (k >> 6)) * D.MB) * 64u + (k & 63u)) * SW_IQ_RB;
}
// a pair as a view-table entry {subject, inc<<2|state, state-change ms, w} (w as in vt) + the second accuser
__device__ __forceinline__ uint4 m_unpack(DevRef D, uint32_t x, uint32_t a, uint32_t b, uint32_t c, uint32_t& conf1) {
  conf1 = M_CONF1(c);
  const uint32_t w = MA_STATE(a) == SWIM_STATE_SUSPECT ? vw_pack(M_CONF0(b, c), MA_NCONF(a), MA_LEAVING(a)) : (MA_ERASED(a) | (MA_LEAVING(a) << 1));
  return make_uint4(x, MA_KEY(a), MB_TICK(b) * D.quantum_ms, w);
}
__device__ __forceinline__ void m_store(DevRef D, size_t idx, uint4 e, uint32_t conf1) {
  const uint32_t st = SW_KST(e.y), inc = SW_KINC(e.y), tick = e.z / D.quantum_ms;
  uint32_t nconf = 0, leaving, erased = 0, conf0 = 0;
  if (st == SWIM_STATE_SUSPECT) { nconf = vw_nconf(e.w); leaving = vw_leaving(e.w); conf0 = vw_conf0(e.w); if (!nconf) conf1 = 0; }
  else { erased = e.w & 1u; leaving = (e.w >> 1) & 1u; conf1 = 0; }
  if (inc >= (1u << 26) || tick >= (1u << 20) || nconf > 3u) atomicOr(D.err, SW_ERR_MASS_RANGE);
  D.mA[idx] = st | (nconf << 2) | (leaving << 4) | (erased << 5) | (inc << 6);
  D.mB[idx] = (tick & 0xFFFFFu) | ((conf0 & 0xFFFu) << 20);
  D.mC[idx] = ((conf0 >> 12) & 0x3FFu) | (conf1 << 10);
}
// a suspicion timer of pair (row, lane k of replica r) was (re)armed: keep the row's and the tile's bounds (k_expire_mass)
__device__ __forceinline__ void m_arm(DevRef D, uint32_t r, uint32_t row, uint32_t k, uint32_t dl) {
  uint32_t* t = &D.m_tile_dl[((size_t)r * D.M + row) * D.nbl + k / SW_BLOCK];
  uint32_t* rw = &D.m_row_dl[(size_t)r * D.M + row];
  const uint32_t tv = *t, rv = *rw;                 // (one round trip for both bounds, not two in a row)
  if (dl < tv) atomicMin(t, dl);
  if (dl < rv) atomicMin(rw, dl);
}
// what the base row says about the node whose word is w
__device__ __forceinline__ uint32_t base_key_of(DevRef D, uint32_t r, uint32_t x, uint32_t w) {
  return (w & NW_BASEMOD) ? D.bk[(size_t)r * D.N + x] : SW_BASE_KEY;
}
// ---- dynamic membership: estNumNodes() of a lane and the scaling laws that take it (swim_device.h) ----------------
__device__ __forceinline__ uint32_t est_n(DevRef D, uint32_t r, size_t l) { return D.dyn ? D.base_known[r] + D.vnk[l] : D.N; }
__device__ __forceinline__ uint32_t retransmit_limit_n(DevRef D, uint32_t n) {
  if (!D.dyn) return D.retransmit_limit;
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < 12; j++) c += n >= D.rl_steps[j];
  return D.retransmit_mult * c;
}
__device__ __forceinline__ uint32_t susp_k_n(DevRef D, uint32_t n) { return !D.dyn ? D.susp_k : (n < D.susp_k_cfg + 2 ? 0u : D.susp_k_cfg); }
// suspicion.go remainingSuspicionTime(c, k, 0, min, max) for a timer that started when the observer knew n nodes:
// float64 exactly as the host evaluates it (no contraction into fused multiply-adds)
__device__ uint32_t susp_timeout_n(DevRef D, uint32_t n, uint32_t c) {
  if (!D.dyn) return sel8(D.susp_timeout, c & 7u);
  const uint64_t min_ms = (uint64_t)D.suspicion_mult * D.scale_milli[n > D.N ? D.N : n] * D.probe_interval_ms / 1000;
  const uint64_t max_ms = (uint64_t)D.suspicion_max_mult * min_ms;
  const uint32_t k = susp_k_n(D, n);
  if (k < 1 || c >= k) return (uint32_t)min_ms;
  if (c == 0) return (uint32_t)max_ms;
  const double frac = c == 1 ? D.susp_frac[1] : c == 2 ? D.susp_frac[2] : D.susp_frac[3];
  const double max_s = __ddiv_rn((double)max_ms, 1000.0), min_s = __ddiv_rn((double)min_ms, 1000.0);
  const double raw = __dsub_rn(max_s, __dmul_rn(frac, __dsub_rn(max_s, min_s)));
  long long t = (long long)floor(__dmul_rn(1000.0, raw));
  if (t < (long long)min_ms) t = (long long)min_ms;
  return (uint32_t)t;
}
// deadline of the Suspect view in slot `sl` of lane l (entry e): in dynamic mode the timer's n sits in the confirmer record
__device__ __forceinline__ uint32_t susp_deadline(DevRef D, size_t l, uint32_t sl, uint4 e) {
  const uint32_t n0 = D.dyn ? D.vc[(size_t)sl * ((size_t)D.R * D.nloc) + l].w : 0;
  return e.z + susp_timeout_n(D, n0, vw_nconf(e.w));
}

// backward-shift deletion of slot `i` of lane l's table (owner only)
__device__ void vt_erase(DevRef D, size_t l, uint32_t i) {
  const size_t NL = (size_t)D.R * D.nloc; const uint32_t m = D.VT - 1;
  uint32_t j = i;
  for (;;) {
    j = (j + 1) & m;
    const uint4 ej = D.vt[(size_t)j * NL + l];
    if (ej.x == VT_EMPTY) break;
    const uint32_t k = vt_home(D, ej.x);
    if (i <= j ? (i < k && k <= j) : (i < k || k <= j)) continue;
    D.vt[(size_t)i * NL + l] = ej; D.vc[(size_t)i * NL + l] = D.vc[(size_t)j * NL + l];
    if (D.vs) D.vs[(size_t)i * NL + l] = D.vs[(size_t)j * NL + l];
    i = j;
  }
  D.vt[(size_t)i * NL + l].x = VT_EMPTY;
}
// observer (r, local k) looking at node x whose word is `w`: the base row unless somebody here has news about x
// AND this observer holds an explicit view
// (MASS: the handle has a dense pair store — a compile-time switch, so that a handle without one runs the code it ran before
// the store existed: the extra branches cost the tick kernels of the headline workload 5-8 %)
template <bool MASS>
__device__ __forceinline__ uint32_t view_of(DevRef D, uint32_t r, uint32_t k, uint32_t x, uint32_t w, uint32_t* since) {
  *since = 0;
  if (MASS && (w & NW_MASS)) {
    const size_t idx = m_idx(D, r, D.mrow[(size_t)r * D.N + x], k);
    const uint32_t a = D.mA[idx];
    if (a) { if (MA_STATE(a) == SWIM_STATE_DEAD) *since = MB_TICK(D.mB[idx]) * D.quantum_ms; return MA_KEY(a); }   // (callers look at `since` of Dead views only)
  } else if (w & NW_SUBJECT) {
    uint4 e; uint32_t fs;
    if (vt_find(D, (size_t)r * D.nloc + k, x, e, fs) != NONE) { *since = e.z; return e.y; }
  }
  return base_key_of(D, r, x, w);
}

__device__ __forceinline__ bool lost(DevRef D, uint32_t r, uint32_t t, uint32_t node, uint32_t leg) {
  if (!D.loss_q32) return false;
  uint32_t w[4];
  uint64_t s = seed_of(D, r);
  sw_philox(t, node, leg, 0, (uint32_t)s, (uint32_t)(s >> 32) ^ SW_STREAM_LOSS, w);
  return w[0] < D.loss_q32;
}
// can a packet from the (running) node with word wa reach the node with word wb right now
__device__ __forceinline__ bool reach(DevRef D, uint32_t r, uint32_t t, uint32_t wa, uint32_t wb, uint32_t rng_node, uint32_t leg) {
  if ((wb & NW_DEAD) || NW_PART(wa) != NW_PART(wb)) return false;
  return !lost(D, r, t, rng_node, leg);
}

// division and remainder by N (nodes per cluster) and nloc (nodes of a cluster on this shard): every BASELINE cluster size is a
// power of two, and a 32-bit division is ~30 instructions in kernels that are short of issue slots (the branch is uniform)
__device__ __forceinline__ uint32_t div_n(DevRef D, uint32_t x) { return D.n_shift != NONE ? x >> D.n_shift : x / D.N; }
__device__ __forceinline__ uint32_t mod_n(DevRef D, uint32_t x) { return D.n_shift != NONE ? x & (D.N - 1u) : x % D.N; }
__device__ __forceinline__ uint32_t div_nloc(DevRef D, size_t l) { return D.nloc_shift != NONE ? (uint32_t)(l >> D.nloc_shift) : (uint32_t)(l / D.nloc); }
__device__ __forceinline__ uint32_t mod_nloc(DevRef D, size_t l) { return D.nloc_shift != NONE ? (uint32_t)l & (D.nloc - 1u) : (uint32_t)(l % D.nloc); }

// ---- the replica's exception list, staged in LDS by the first SW_EXC_MAX lanes of a block ----------
struct ExcList {
  uint32_t* id; uint32_t* w; uint32_t n;           // n > SW_EXC_MAX: unusable, fall back to nw
  __device__ void stage(DevRef D, uint32_t r, uint32_t* lds) {   // caller provides the barrier
    id = lds; w = lds + SW_EXC_MAX;
    n = D.exc_cnt[r];
    if (threadIdx.x < SW_EXC_MAX) {                  // {id, node word} pairs next to the count: one trip, no dependent lookup
      uint2 e = D.exc_ent[(size_t)r * SW_EXC_MAX + threadIdx.x];
      id[threadIdx.x] = e.x; w[threadIdx.x] = e.y;
    }
  }
  __device__ __forceinline__ bool usable() const { return n <= SW_EXC_MAX; }
  // node word of x when the list is usable: 0 unless listed
  __device__ __forceinline__ uint32_t word(uint32_t x) const {
    uint32_t v = 0;
    for (uint32_t j = 0; j < n; j++) v = id[j] == x ? w[j] : v;
    return v;
  }
};

// ---- statistics: per-block LDS counters, flushed once --------------------------------------------
__device__ __forceinline__ unsigned long long* stat_ptr(DevRef D, int i) {
  return &D.stats[(size_t)(blockIdx.x % SW_STAT_COPIES) * SW_STAT_STRIDE + i];
}
// (file scope, like g_lds_dyn: a pointer handed through NodeCtxT's reference member loses its LDS address space — every add() in
// k_resolve's handlers was a scratch_load of the pointer + s_waitcnt vmcnt(0) + flat_atomic_add; profiles/r03_isa_notes.txt item 1)
__shared__ uint32_t g_lds_stats[ST_COUNT];
struct BlockStats {
  __device__ void init(uint32_t*) {
    for (uint32_t i = threadIdx.x; i < ST_COUNT; i += blockDim.x) g_lds_stats[i] = 0;
    __syncthreads();
  }
  __device__ __forceinline__ void add(int i, uint32_t v = 1) { atomicAdd(&g_lds_stats[i], v); }
  // converged call sites: one LDS atomic per wave instead of one per lane
  // converged call sites: wave-reduce a per-lane tally, one LDS atomic per wave
  __device__ __forceinline__ void wave_add(int i, uint32_t v) {
    if (!__any(v != 0)) return;
    for (int off = 32; off; off >>= 1) v += __shfl_down(v, off);
    if (sw_lane() == 0) atomicAdd(&g_lds_stats[i], v);
  }
  __device__ __forceinline__ void count(int i, bool pred) {
    uint64_t m = __ballot(pred);
    if (m && sw_lane() == 0) atomicAdd(&g_lds_stats[i], (uint32_t)__popcll(m));
  }
  __device__ void flush(DevRef D) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < ST_COUNT; i += blockDim.x)
      if (g_lds_stats[i]) atomicAdd(stat_ptr(D, i), (unsigned long long)g_lds_stats[i]);
  }
};

// ---- wave-aggregated append of at most one record per lane ------------------------------------
// ballot the lanes that have a record, one atomicAdd per wave, prefix rank by popcount
__device__ __forceinline__ void wave_append(DevRef D, uint32_t sh, bool want, uint4 rec) {
  uint64_t mask = __ballot(want);
  if (!mask) return;
  uint32_t lane = sw_lane(), leader = (uint32_t)__ffsll((long long)mask) - 1, base = 0;
  if (lane == leader) base = atomicAdd(&D.out_cnt[sh], (uint32_t)__popcll(mask));
  base = __shfl(base, leader);
  if (want) {
    uint32_t pos = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1));
    if (pos < D.out_cap_tab[sh]) D.out_tab[sh][pos] = rec;
    else atomicOr(D.err, SW_ERR_EDGE_OVF);
  }
}
// one shard (the headline workload): the shard's own list through the descriptor's scalar fields — no pointer table in memory, global
// (not flat) stores (profiles/r03_isa_notes.txt item 3: the one-shard kernels carried the sharded append path: 35 flat stores, SGPR spills)
__device__ __forceinline__ void wave_append_one(DevRef D, bool want, uint4 rec) {
  uint64_t mask = __ballot(want);
  if (!mask) return;
  uint32_t lane = sw_lane(), leader = (uint32_t)__ffsll((long long)mask) - 1, base = 0;
  if (lane == leader) base = atomicAdd(&D.out_cnt[0], (uint32_t)__popcll(mask));
  base = __shfl(base, leader);
  if (want) {
    uint32_t pos = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1));
    if (pos < D.out_cap[0]) st_global_u4(D.out[0] + pos, rec);
    else atomicOr(D.err, SW_ERR_EDGE_OVF);
  }
}
// the destination shard differs per lane: one aggregated append per shard present in the wave
// (MULTI = false: the caller's kernel was instantiated for one shard; true: decided at run time)
template <bool MULTI = true>
__device__ __forceinline__ void wave_append_sharded(DevRef D, bool want, uint32_t sh, uint4 rec) {
  if constexpr (!MULTI) { wave_append_one(D, want, rec); return; }
  if (D.n_shards == 1) { wave_append(D, 0, want, rec); return; }
  uint64_t todo = __ballot(want);
  while (todo) {
    uint32_t leader = (uint32_t)__ffsll((long long)todo) - 1;
    uint32_t s = __shfl(sh, leader);
    bool mine = want && sh == s;
    wave_append(D, s, mine, rec);
    todo &= ~__ballot(mine);
  }
}
// the shard's own list
template <bool MULTI = true>
__device__ __forceinline__ void wave_append_own(DevRef D, bool want, uint4 rec) {
  if constexpr (!MULTI) wave_append_one(D, want, rec); else wave_append(D, D.rank, want, rec);
}
__device__ __forceinline__ uint4 mk_edge(DevRef D, uint32_t r, uint32_t dst, uint32_t subject, uint32_t inc, uint32_t type, uint32_t from) {
  return make_uint4(r * D.N + dst, subject, inc, (type << 30) | (from & 0x3FFFFFFFu));
}

// memberlist.Transport bridge: a rumour for an attached node is handed to the host instead of an inbox
__device__ __forceinline__ void capture(DevRef D, uint32_t src, uint32_t gdst, uint32_t subject, uint32_t inc, uint32_t meta) {
  uint32_t pos = atomicAdd(D.cap_cnt, 1u);
  if (pos < D.cap_cap) { D.cap[pos] = make_uint4(src, subject, inc, meta); D.cap_dst[pos] = gdst; }
  else atomicOr(D.err, SW_ERR_EVENT_OVF);
}

// ---- SWIM_F_PIGGYBACK: sendMsg (net.go) lets every ping / indirect ping / ack / nack carry its sender's
// getBroadcasts().  The probing lane files an *order* addressed to the sender S (an edge record with subject
// SWIM_SUBJECT_PIGGY, incarnation = the packet's receiver or NONE when the packet is lost, type = carrier
// kind, from = the prober); S picks the broadcasts in k_resolve.  An order for a node with nothing queued is a
// no-op, so it is only filed when the block hint says S's block may hold something (k_deliver re-checks S's
// header exactly, which makes the racy hint read harmless).
// qbits: one bit per local lane, set exactly while the node has something queued (broadcasts or user events).
// Set by whoever pushes (k_resolve, stimulus), cleared by whoever drains (gossip role, k_resolve's piggy-back pick).
__device__ __forceinline__ bool q_bit(DevRef D, size_t l) { return (D.qbits[l >> 5] >> (l & 31)) & 1u; }
// a whole wave of 64 consecutive, 64-aligned lanes publishes its transitions with at most two atomics per word
__device__ __forceinline__ void q_bits_wave(DevRef D, size_t l, bool set, bool clr) {
  uint64_t ms = __ballot(set), mc = __ballot(clr);
  if (!(ms | mc)) return;
  uint32_t lane = sw_lane();
  if (lane == 0 || lane == 32) {
    uint32_t s32 = (uint32_t)(ms >> lane), c32 = (uint32_t)(mc >> lane);
    if (s32) atomicOr(&D.qbits[l >> 5], s32);
    if (c32) atomicAnd(&D.qbits[l >> 5], ~c32);
  }
}
__device__ __forceinline__ void q_bit_lane(DevRef D, size_t l, bool set, bool clr) {
  if (set) atomicOr(&D.qbits[l >> 5], 1u << (l & 31));
  if (clr) atomicAnd(&D.qbits[l >> 5], ~(1u << (l & 31)));
}
// The prober reads the bit while gossip blocks of the same launch may be clearing it: a stale 1 files an order
// that k_deliver (which sees the settled bit) drops; a 0 is final, since only k_resolve sets bits.
__device__ __forceinline__ bool piggy_hint(DevRef D, uint32_t r, uint32_t sender, uint32_t peer_active) {
  if (!(D.flags & SWIM_F_PIGGYBACK)) return false;
  if (sender < D.i0 || sender >= D.i0 + D.nloc) return peer_active != 0;
  return q_bit(D, (size_t)r * D.nloc + (sender - D.i0));
}
__device__ __forceinline__ uint4 piggy_rec(DevRef D, uint32_t r, uint32_t sender, uint32_t receiver, uint32_t kind, uint32_t prober) {
  return make_uint4(r * D.N + sender, SWIM_SUBJECT_PIGGY, receiver, (kind << 30) | (prober & 0x3FFFFFFFu));
}

// ---- stagger: which nodes act in tick t --------------------------------------------------------
// chunk c = id / CH; gossip phase = c % G; probe phase = (c / G) % P.  Enumerate the active set
// compactly: index a -> node id i (or NONE).  CH is a power of two.
__device__ __forceinline__ uint32_t map_gossip(DevRef D, uint32_t ph, uint32_t a) {
  uint32_t sh = __ffs(D.CH) - 1;
  uint32_t c0 = D.i0 >> sh, c1 = (D.i0 + D.nloc + D.CH - 1) >> sh;
  uint32_t q_lo = c0 > ph ? (c0 - ph + D.G - 1) / D.G : 0;
  uint32_t c = ph + D.G * (q_lo + (a >> sh));
  if (c >= c1) return NONE;
  uint32_t i = (c << sh) + (a & (D.CH - 1));
  return (i < D.i0 + D.nloc && i >= D.i0) ? i : NONE;
}
__device__ __forceinline__ uint32_t map_probe(DevRef D, uint32_t ph, uint32_t a) {
  uint32_t sh = __ffs(D.CH) - 1;
  uint32_t c0 = D.i0 >> sh, c1 = (D.i0 + D.nloc + D.CH - 1) >> sh;
  uint32_t u_lo = c0 / D.G;
  uint32_t m_lo = u_lo > ph ? (u_lo - ph + D.P - 1) / D.P : 0;
  uint32_t q = a >> sh, m = m_lo + q / D.G, gg = q % D.G;
  uint32_t c = (ph + D.P * m) * D.G + gg;
  if (c < c0 || c >= c1) return NONE;
  uint32_t i = (c << sh) + (a & (D.CH - 1));
  return (i < D.i0 + D.nloc && i >= D.i0) ? i : NONE;
}

__device__ __forceinline__ uint32_t awareness_apply(DevRef D, uint32_t aw, int delta) {
  int v = (int)aw + delta, mx = (int)D.awareness_max - 1;
  return (uint32_t)(v < 0 ? 0 : v > mx ? mx : v);
}

// util.go kRandomNodes: <= 3n draws of randomOffset(n), skip excluded and already picked.
// mode 0 = gossip() (skip Left, and Dead for longer than GossipToTheDeadTime);
// mode 1 = probeNode's indirect helpers (skip the target and anything not Alive).
// wout[] receives the picked nodes' words so the caller needs no second lookup.
template <bool MASS>
__device__ uint32_t k_random_nodes(DevRef D, uint32_t r, uint32_t o, uint32_t k_local, uint32_t t,
                                   uint32_t stream, uint32_t want, int mode, uint32_t target,
                                   uint32_t* out, uint32_t* wout, const ExcList& X) {
  SwDraws d; d.init(seed_of(D, r), stream, t, o);
  uint32_t found = 0, now = now_ms(D, t);
  const uint32_t* nw = D.nw + (size_t)r * D.N;
  uint64_t tries = 3ull * D.N;
  // the first Philox block yields four candidates: fetch their node words together (four
  // independent random reads in flight) instead of one dependent read per loop trip
  uint32_t x4[4], w4[4];
#pragma unroll
  for (int j = 0; j < 4; j++) x4[j] = mod_n(D, d.get(j));
  if (X.usable()) {
#pragma unroll
    for (int j = 0; j < 4; j++) w4[j] = X.word(x4[j]);                 // no memory access
  } else {
#pragma unroll
    for (int j = 0; j < 4; j++) w4[j] = nw[x4[j]];
  }
  for (uint64_t i = 0; i < tries && found < want; i++) {
    uint32_t x, w;
    if (i < 4) { x = i == 0 ? x4[0] : i == 1 ? x4[1] : i == 2 ? x4[2] : x4[3]; w = i == 0 ? w4[0] : i == 1 ? w4[1] : i == 2 ? w4[2] : w4[3]; }
    else { x = mod_n(D, d.get((uint32_t)i)); w = X.usable() ? X.word(x) : nw[x]; }
    if (x == o) continue;
    uint32_t since, key = view_of<MASS>(D, r, k_local, x, w, &since), st = SW_KST(key);
    if (mode == 0) {
      if (key < 4u) continue;                        // incarnation 0: never heard of it, not in this node's member list
      if (st == SWIM_STATE_LEFT) continue;
      if (st == SWIM_STATE_DEAD && now - since > D.gossip_to_dead_ms) continue;
    } else {
      if (x == target || st != SWIM_STATE_ALIVE) continue;
    }
    bool dup = false;
    for (uint32_t j = 0; j < found; j++) dup |= out[j] == x;
    if (dup) continue;
    out[found] = x; wout[found] = w; found++;
  }
  return found;
}

// =================================================================================================
// role: expire — suspectNode's time.AfterFunc.  Still Suspect when the (confirmation-shortened)
// timeout lapses => deadNode(dead{inc, node, from: self}), delivered to self via the common inbox.
// =================================================================================================
__device__ __forceinline__ void inbox_place(DevRef D, uint4 rec, size_t l, uint32_t pos);
// Gate: dl_blk[node block] is a lower bound of the earliest deadline among the block's acting lanes, vmeta[lane].z of
// the lane's own timers.  One WAVE per 256-node block (four per workgroup, no barrier): out after one word unless the
// bound has passed; only a lane whose own bound has passed walks its view table.
__device__ __forceinline__ void role_expire(DevRef D, uint32_t b, uint32_t nb) {
  (void)nb;
  const uint32_t nbk = b * (SW_BLOCK / 64) + threadIdx.x / 64, lane = sw_lane();
  if (nbk >= D.NB) return;
  const uint32_t t = *D.tick, now = now_ms(D, t);
  if (now < D.dl_blk[nbk]) return;
  const size_t NL = (size_t)D.R * D.nloc;
  uint32_t fired = 0, m = NONE;
  for (uint32_t part = 0; part < SW_BLOCK / 64; part++) {
    const size_t l = (size_t)nbk * SW_BLOCK + part * 64 + lane;
    if (l >= NL) continue;
    const uint32_t r = (uint32_t)(l / D.nloc), o = D.i0 + (uint32_t)(l % D.nloc);
    const uint4 vm = VMETA(l);
    uint32_t d = vm.z;
    if (d == NONE) continue;
    if (D.nw[(size_t)r * D.N + o] & NW_INERT) continue;              // its timers rest; a revive lowers dl_blk again
    if (now >= d) {
      uint32_t next = NONE, left = vm.y;                             // Suspect views still to be found
      for (uint32_t sl = 0; sl < D.VT && left; sl++) {
        const uint4 e = D.vt[(size_t)sl * NL + l];
        if (e.x == VT_EMPTY || SW_KST(e.y) != SWIM_STATE_SUSPECT) continue;
        left--;
        const uint32_t dl = susp_deadline(D, l, sl, e);
        if (now >= dl) {
          // a timer is not a packet: the verdict goes straight into the node's own inbox line
          inbox_place(D, mk_edge(D, r, o, e.x, SW_KINC(e.y), SWIM_MSG_DEAD, o), l, atomicAdd(&D.in_cnt[l], 1u));
          fired++;
        }
        next = dl < next ? dl : next;   // a fired timer keeps the bound low until its verdict is merged
      }
      d = next; VMETA(l).z = d;
    }
    m = d < m ? d : m;
  }
  for (int off = 32; off; off >>= 1) { uint32_t v = __shfl_xor(m, off); m = v < m ? v : m; }
  if (lane == 0) D.dl_blk[nbk] = m;
  if (__any(fired != 0)) {
    for (int off = 32; off; off >>= 1) fired += __shfl_down(fired, off);
    if (lane == 0) { atomicAdd(stat_ptr(D, ST_TIMEOUTS), (unsigned long long)fired); atomicAdd(stat_ptr(D, ST_EDGES), (unsigned long long)fired); }
  }
}

// =================================================================================================
// role: pending — ProbeTimeout after a failed direct ping: indirectPingReq to IndirectChecks random
// alive peers; each relays the target's ack or (Lifeguard) answers nack one ProbeTimeout later.
// =================================================================================================
template <int KMAX, bool MULTI, bool MASS>
__device__ __forceinline__ void role_pending(DevRef D, uint32_t b, uint32_t nb, uint32_t* lds_stats, uint32_t peer_active) {
  BlockStats S; S.init(lds_stats);
  uint32_t t = *D.tick;
  if (t >= D.TQ) {
    uint32_t li = (t - D.TQ) % (D.TQ + 1);
    uint32_t n = D.pend_cnt[li]; if (n > D.pend_cap) n = D.pend_cap;
    const uint32_t* list = D.pend + (size_t)li * D.pend_cap;
    for (uint32_t e = b * SW_BLOCK + threadIdx.x; e < n; e += nb * SW_BLOCK) {
      uint32_t l = list[e], r = l / D.nloc, k = l % D.nloc, i = D.i0 + k;
      const uint32_t* nw = D.nw + (size_t)r * D.N;
      uint32_t wi = nw[i];
      if (wi & NW_INERT) continue;
      uint2 h = D.ph[l]; uint4 p0 = D.pr0[l];
      if (p_stage(h.y) != 1 || p0.w + D.TQ != t) continue;
      uint32_t x = p0.x, wx = nw[x], peers[KMAX], pw[KMAX];
      ExcList none; none.id = nullptr; none.w = nullptr; none.n = SW_EXC_MAX + 1;   // entries span replicas: read nw
      uint32_t np = k_random_nodes<MASS>(D, r, i, k, t, SW_STREAM_INDIRECT, D.k_indirect, 1, x, peers, pw, none);
      uint32_t expected = 0, nacks = 0; bool acked = false;
      bool nack_in_time = 2 * D.TQ < p0.z - p0.w;
      for (uint32_t q = 0; q < np; q++) {
        if (D.flags & SWIM_F_NACK) expected++;
        const uint32_t hq = peers[q];
        bool there = reach(D, r, t, wi, pw[q], i, 20 + 4 * q);
        // rare path (a direct ping just failed): plain per-lane appends are fine
        if (piggy_hint(D, r, i, peer_active)) wave_append_sharded<MULTI>(D, true, i / D.nloc, piggy_rec(D, r, i, there ? hq : NONE, SWIM_CTL_INDIRECT, i));
        if (!there) continue;
        bool hx = reach(D, r, t, pw[q], wx, i, 21 + 4 * q), xh = hx && reach(D, r, t, wx, pw[q], i, 22 + 4 * q), ok = hx && xh;
        bool back = reach(D, r, t, pw[q], wi, i, 23 + 4 * q);
        if (piggy_hint(D, r, hq, peer_active)) {
          wave_append_sharded<MULTI>(D, true, hq / D.nloc, piggy_rec(D, r, hq, hx ? x : NONE, SWIM_CTL_PING, i));
          if (ok) wave_append_sharded<MULTI>(D, true, hq / D.nloc, piggy_rec(D, r, hq, back ? i : NONE, SWIM_CTL_ACK, i));
          else if ((D.flags & SWIM_F_NACK) && nack_in_time) wave_append_sharded<MULTI>(D, true, hq / D.nloc, piggy_rec(D, r, hq, back ? i : NONE, SWIM_CTL_NACK, i));
        }
        if (hx && piggy_hint(D, r, x, peer_active)) wave_append_sharded<MULTI>(D, true, x / D.nloc, piggy_rec(D, r, x, xh ? hq : NONE, SWIM_CTL_ACK, i));
        if (ok && back) acked = true;
        else if (!ok && back && nack_in_time) nacks++;
      }
      uint32_t aw = p_aw(h.y);
      // probeNode's TCP fallback ping next to the indirect probes: TCP rides out packet loss, so it reaches every
      // running node of the same partition — and of the same TCP class (DisableTcpPingsForNode: not across datacenters)
      const bool tcp = !acked && (D.flags & SWIM_F_TCP_FALLBACK) && !(wx & NW_DEAD) && !((wi ^ wx) & (0x7F000000u | NW_TCP_MASK));
      if (acked || tcp) {
        aw = awareness_apply(D, aw, -1); S.add(acked ? ST_IACKS : ST_TCPACKS);
        D.pr0[l].x = NONE; h.y = p_pack(p_epoch(h.y), aw, 0, 0);
      } else h.y = p_pack(p_epoch(h.y), aw, 2, expected > 0 ? expected - nacks : 1);
      D.ph[l] = h;
    }
  }
  S.flush(D);
}

// =================================================================================================
// role: probe — memberlist probe()/probeNode (state.go) for the nodes whose probe ticker fires now.
// Hot path per lane: own word, 8 B of probe state, one Feistel evaluation, the target's word.
// =================================================================================================
template <bool MULTI, bool MASS, bool TB>
__device__ __forceinline__ void role_probe(DevRef D, uint32_t r, uint32_t pb, uint32_t a, uint32_t* lds_stats, uint32_t* lds_exc, uint32_t* s_cnt, uint32_t peer_active, uint32_t* s_tb, uint32_t* s_tbase) {
  ExcList X; X.stage(D, r, lds_exc);
  if (threadIdx.x == 0) s_cnt[0] = 0;
  if (TB && threadIdx.x < SW_TB_BINS) s_tb[threadIdx.x] = 0;
  BlockStats S; S.init(lds_stats);
  uint32_t t = *D.tick;
  uint32_t i = map_probe(D, t % D.P, a);
  const uint32_t* nw = D.nw + (size_t)r * D.N;
  bool e_buddy = false, e_self = false, c_probe = false, c_ack = false, e_pend = false;
  bool o_ping = false, o_ack = false;                // piggy-back orders: for my ping, for the target's ack
  uint32_t c_x = NONE;                               // SWIM_F_COORDINATES: the target whose direct ack came back (serf's ping delegate)
  uint32_t o_ping_rcv = NONE, o_ack_rcv = NONE, o_x = 0;
  uint4 rec_buddy = make_uint4(0, 0, 0, 0), rec_self = rec_buddy; uint32_t buddy_sh = 0;
  size_t l = 0;
  uint32_t wi = i != NONE ? nw[i] : NW_DEAD;
  if (!(wi & NW_INERT)) {
    uint32_t k = i - D.i0; l = (size_t)r * D.nloc + k;
    uint2 h = D.ph[l], h0 = h;
    uint32_t aw = p_aw(h.y), stage = p_stage(h.y), nackm = p_nackm(h.y), epoch = p_epoch(h.y), cursor = h.x;
    bool busy = stage != 0;
    if (busy) {
      uint4 p0 = D.pr0[l];
      if (t >= p0.z) {
        // probeNode's failure epilogue: awareness, then suspectNode(suspect{inc, node, self})
        aw = awareness_apply(D, aw, (int)nackm);
        S.add(ST_PFAIL); S.add(ST_NACKMISS, nackm);
        e_self = true; rec_self = mk_edge(D, r, i, p0.x, p0.y, SWIM_MSG_SUSPECT, i);
        D.pr0[l].x = NONE; stage = 0; nackm = 0; busy = false;
      }
    }
    if (!busy) {
      // probe(): next entry of the shuffled list that is not self / dead / left
      uint32_t num_check = 0, x = NONE, key = 0, wx = 0, since;
      while (num_check < D.N) {
        if (cursor >= D.N) { epoch = (epoch + 1) & 0xFFFFu; cursor = 0; num_check++; continue; }   // resetNodes
        uint32_t c = sw_probe_perm(seed_of(D, r), D.N, i, epoch, cursor++);
        wx = X.usable() ? X.word(c) : nw[c]; key = view_of<MASS>(D, r, k, c, wx, &since);
        if (c == i || SW_KST(key) == SWIM_STATE_DEAD || SW_KST(key) == SWIM_STATE_LEFT) { num_check++; continue; }
        x = c; break;
      }
      if (x != NONE) {
        c_probe = true;
        bool fwd = reach(D, r, t, wi, wx, i, 16);
        if (fwd && SW_KST(key) != SWIM_STATE_ALIVE && (D.flags & SWIM_F_BUDDY_SUSPECT)) {
          e_buddy = true; rec_buddy = mk_edge(D, r, x, x, SW_KINC(key), SWIM_MSG_SUSPECT, i); buddy_sh = x / D.nloc;
        }
        bool ack = fwd && !lost(D, r, t, i, 17);
        if (D.flags & SWIM_F_PIGGYBACK) {
          o_x = x;
          // a non-alive target gets the ping+suspect compound, which is sent raw (no piggy-back)
          if (SW_KST(key) == SWIM_STATE_ALIVE && piggy_hint(D, r, i, peer_active)) { o_ping = true; o_ping_rcv = fwd ? x : NONE; }
          if (fwd && piggy_hint(D, r, x, peer_active)) { o_ack = true; o_ack_rcv = ack ? i : NONE; }
        }
        if (ack) { aw = awareness_apply(D, aw, -1); c_ack = true; c_x = x; }
        else {
          stage = 1; nackm = 1; e_pend = true;
          D.pr0[l] = make_uint4(x, SW_KINC(key), t + D.P * (aw + 1), t);   // awareness.ScaleTimeout(ProbeInterval)
        }
      }
    }
    h.x = cursor; h.y = p_pack(epoch, aw, stage, nackm);
    if (h.x != h0.x || h.y != h0.y) D.ph[l] = h;
  }
  S.count(ST_PROBES, c_probe); S.count(ST_ACKS, c_ack);

  if (D.coord && __any(c_ack)) {                    // NotifyPingComplete: list the prober for k_coord_update (one atomic per wave)
    const uint64_t mask = __ballot(c_ack);
    const uint32_t lane = sw_lane(), leader = (uint32_t)__ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(D.c_cnt, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    if (c_ack) {
      const uint32_t pos = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1));
      if (pos < D.c_cap) D.c_list[pos] = make_uint2((uint32_t)l, c_x); else atomicOr(D.err, SW_ERR_PEND_OVF);
    }
  }
  // everything below is off the common path: wave-aggregated appends suffice
  if (__any(e_pend)) {
    uint32_t li = t % (D.TQ + 1);
    uint64_t mask = __ballot(e_pend);
    uint32_t lane = sw_lane(), leader = (uint32_t)__ffsll((long long)mask) - 1, base = 0;
    if (lane == leader) base = atomicAdd(&D.pend_cnt[li], (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    if (e_pend) {
      uint32_t pos = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1));
      if (pos < D.pend_cap) D.pend[(size_t)li * D.pend_cap + pos] = (uint32_t)l;
      else atomicOr(D.err, SW_ERR_PEND_OVF);
    }
  }
  if (__any(e_self)) wave_append_own<MULTI>(D, e_self, rec_self);
  if (__any(e_buddy)) wave_append_sharded<MULTI>(D, e_buddy, buddy_sh, rec_buddy);
  uint32_t ne = (uint32_t)e_self + (uint32_t)e_buddy;
  if (ne) { S.add(ST_EDGES, ne); if (e_buddy && buddy_sh != D.rank) S.add(ST_EDGES_REMOTE); }
  // piggy-back orders: during dissemination every probing lane files one or two, so the ones that stay on
  // this shard go to the block's private segment (wave prefix sum, one LDS atomic per wave, no global atomic)
  if (D.flags & SWIM_F_PIGGYBACK) {
    bool ack_local = o_ack && (!MULTI || o_x / D.nloc == D.rank);
    uint32_t n_loc = (uint32_t)o_ping + (uint32_t)ack_local, incl = n_loc, my_off = 0;
    if (TB) {
      // the orders that stay on this shard go to the buckets of their receivers' tiles (the prober's own tile for its ping, any tile of
      // the replica for the ack): LDS offsets per tile, one global atomic per (block, tile)
      const uint32_t tl0 = (uint32_t)(((size_t)r * D.nloc) / SW_TB_TILE);
      uint32_t bin_p = 0, bin_a = 0, loc_p = 0, loc_a = 0;
      if (o_ping) { bin_p = (uint32_t)(((size_t)r * D.nloc + (i - D.i0)) / SW_TB_TILE) - tl0; loc_p = atomicAdd(&s_tb[bin_p], 1u); }
      if (ack_local) { bin_a = (uint32_t)(((size_t)r * D.nloc + (o_x - D.i0)) / SW_TB_TILE) - tl0; loc_a = atomicAdd(&s_tb[bin_a], 1u); }
      __syncthreads();
      if (threadIdx.x < SW_TB_BINS) {
        uint32_t c = s_tb[threadIdx.x], b = 0;
        if (c) {
          b = atomicAdd(&D.tb_cnt[tl0 + threadIdx.x], c);
          if (b + c > D.tb_cap) { atomicOr(D.err, SW_ERR_EDGE_OVF); b = NONE; }
          if (MULTI) *D.act = 1;
        }
        s_tbase[threadIdx.x] = b;
      }
      __syncthreads();
      if (o_ping && s_tbase[bin_p] != NONE) D.tb[(size_t)(tl0 + bin_p) * D.tb_cap + s_tbase[bin_p] + loc_p] = piggy_rec(D, r, i, o_ping_rcv, SWIM_CTL_PING, i);
      if (ack_local && s_tbase[bin_a] != NONE) D.tb[(size_t)(tl0 + bin_a) * D.tb_cap + s_tbase[bin_a] + loc_a] = piggy_rec(D, r, o_x, o_ack_rcv, SWIM_CTL_ACK, i);
    } else
    if (__any(n_loc != 0)) {
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { uint32_t v = __shfl_up(incl, off); if (sw_lane() >= (uint32_t)off) incl += v; }
      uint32_t wave_total = __shfl(incl, 63), wbase = 0;
      if (sw_lane() == 63) wbase = atomicAdd(&s_cnt[0], wave_total);
      wbase = __shfl(wbase, 63);
      my_off = wbase + incl - n_loc;
      const uint32_t segb = D.R * D.nb_gossip + r * D.nb_probe + pb;
      uint4* dst = D.seg + (size_t)segb * D.seg_cap + my_off;
      if (o_ping) *dst++ = piggy_rec(D, r, i, o_ping_rcv, SWIM_CTL_PING, i);
      if (ack_local) *dst = piggy_rec(D, r, o_x, o_ack_rcv, SWIM_CTL_ACK, i);
    }
    if (MULTI) { bool rem = o_ack && !ack_local; if (__any(rem)) wave_append_sharded<MULTI>(D, rem, o_x / D.nloc, piggy_rec(D, r, o_x, o_ack_rcv, SWIM_CTL_ACK, i)); }
    __syncthreads();
    if (!TB && threadIdx.x == 0 && s_cnt[0]) { D.seg_cnt[D.R * D.nb_gossip + r * D.nb_probe + pb] = s_cnt[0]; if (MULTI) *D.act = 1; }
  }
  S.flush(D);
}

// =================================================================================================
// role: gossip — memberlist gossip() (state.go) + TransmitLimitedQueue.GetBroadcasts (queue.go) +
// serf delegate.GetBroadcasts for user events.  The node's queues are staged in LDS, k random
// peers come from the counter-based RNG, and the block compacts its packets into the per-shard
// outbound edge lists with one global atomic per (block, shard).
// =================================================================================================

// limitedBroadcast.Less: transmits asc, msgLen desc, id desc
__device__ __forceinline__ bool ent_before(DevRef D, uint32_t ma, uint32_t mb) {
  uint32_t ta = m_tr(ma), tb = m_tr(mb);
  if (ta != tb) return ta < tb;
  uint32_t la = sel4(D.msg_len, m_type(ma)), lb = sel4(D.msg_len, m_type(mb));
  if (la != lb) return la > lb;
  return m_seq(ma) > m_seq(mb);
}

// SWIM_F_FILTER_NOOP: would aliveNode/suspectNode/deadNode at the receiver return without doing anything —
// judged only by conditions that stay true whatever else reaches it this tick (view incarnations never
// decrease): an older incarnation, or the same incarnation in a state the message cannot move.  One
// random 16-byte read of the receiver's view replaces an edge write, an inbox atomic and a merge.
// (key, wpack) = the receiver's view of the subject (wpack = first accuser<<3 | confirmations; 0 for a base-row
// view, which is never Suspect), vci = index of its confirmer record in vc, `e` = the queue entry
// {subject, inc, from, meta}.
__device__ __forceinline__ bool noop_given_view(DevRef D, uint32_t key, uint32_t wpack, size_t vci, uint4 e) {
  uint32_t type = m_type(e.w), vinc = SW_KINC(key), st = SW_KST(key);
  if (type == SWIM_MSG_ALIVE) return e.y <= vinc;
  if (e.y != vinc) return e.y < vinc;
  if (st == SWIM_STATE_DEAD || st == SWIM_STATE_LEFT) return true;
  if (type == SWIM_MSG_SUSPECT && st == SWIM_STATE_SUSPECT) {
    uint32_t nc = vw_nconf(wpack);
    if (vw_conf0(wpack) == e.z) return true;
    if (!D.dyn) { if (nc >= D.susp_k) return true; if (nc == 0) return false; }
    uint4 b = D.vc[vci];
    if (D.dyn) { if (nc >= susp_k_n(D, b.w)) return true; if (nc == 0) return false; }
    return b.x == e.z || (nc >= 2 && b.y == e.z) || (nc >= 3 && b.z == e.z);
  }
  return false;
}
// the same question for receiver lane lr and a subject whose node word is ws; `first` = the home slot of the
// subject in the receiver's table when the caller fetched it already (have_first)
template <bool MASS>
__device__ __forceinline__ bool noop_at_receiver(DevRef D, uint32_t r, size_t lr, uint32_t ws, uint4 e, bool have_first, uint4 first) {
  const size_t NL = (size_t)D.R * D.nloc;
  if (MASS && (ws & NW_MASS)) {                              // the pair of the dense store: one 4-byte read decides all but suspect-on-suspect
    const size_t idx = m_idx(D, r, D.mrow[(size_t)r * D.N + e.x], (uint32_t)(lr - (size_t)r * D.nloc));
    const uint32_t a = have_first ? first.x : D.mA[idx];
    if (a) {
      const uint32_t type = m_type(e.w), vinc = MA_INC(a), st = MA_STATE(a);
      if (type == SWIM_MSG_ALIVE) return e.y <= vinc;
      if (e.y != vinc) return e.y < vinc;
      if (st == SWIM_STATE_DEAD || st == SWIM_STATE_LEFT) return true;
      if (type == SWIM_MSG_SUSPECT && st == SWIM_STATE_SUSPECT) {
        const uint32_t nc = MA_NCONF(a);
        if (nc >= D.susp_k) return true;
        const uint32_t b = D.mB[idx], c = D.mC[idx];
        return M_CONF0(b, c) == e.z || (nc >= 1 && M_CONF1(c) == e.z);
      }
      return false;
    }
    return noop_given_view(D, base_key_of(D, r, e.x, ws), 0, 0, e);
  }
  if (ws & NW_SUBJECT) {
    uint4 v; uint32_t fs;
    if (!have_first) first = D.vt[(size_t)vt_home(D, e.x) * NL + lr];
    uint32_t sl = vt_probe(D, lr, e.x, first, v, fs);
    if (sl != NONE) return noop_given_view(D, v.y, v.w, (size_t)sl * NL + lr, e);
  }
  return noop_given_view(D, base_key_of(D, r, e.x, ws), 0, 0, e);
}

// where a node's queue lives: staged in LDS (gossip role: entry j of this lane at sq[j*256]) or in HBM
// (k_resolve: entry j of lane l at q[j*NL + l])
template <uint32_t STRIDE> struct LdsQT { uint4* p; __device__ __forceinline__ uint32_t& meta(uint32_t j) const { return p[j * STRIDE].w; } };
typedef LdsQT<SW_BLOCK> LdsQ;                  // (the gossip role: 256 lanes, entry j of a lane at sq[j * 256])
// k_resolve on a handle with the dense pair store: only {subject, meta} of an entry are staged (8 bytes: NodeCtxT::SPLIT)
template <uint32_t STRIDE> struct LdsQ2T { uint2* p; __device__ __forceinline__ uint32_t& meta(uint32_t j) const { return p[j * STRIDE].y; } };
struct HbmQ { uint4* p; size_t NL; __device__ __forceinline__ uint4& at(uint32_t j) const { return p[(size_t)j * NL]; } };
// k_resolve: only the meta words (type | transmits | seq) of the lane's queue, staged in LDS
template <uint32_t STRIDE> struct MetaQT { uint32_t* p; __device__ __forceinline__ uint32_t& meta(uint32_t j) const { return p[j * STRIDE]; } };

// one GetBroadcasts(overhead, limit) over a queue.  `live` = entries still queued;
// returns the bitmask sent; bumps transmits / retires at the retransmit limit.
template <typename QV>
__device__ uint32_t get_broadcasts(DevRef D, QV sq, uint32_t n, uint32_t& live, uint32_t overhead, int limit, int& used_out, uint32_t retransmit_limit) {
  uint32_t taken = 0; int used = 0;
  for (;;) {
    int free_b = limit - used - (int)overhead;
    if (free_b <= 0) break;
    uint32_t best = NONE, bmeta = 0;
    for (uint32_t j = 0; j < n; j++) {
      if (!((live >> j) & 1u) || ((taken >> j) & 1u)) continue;
      uint32_t meta = sq.meta(j);
      if ((int)sel4(D.msg_len, m_type(meta)) > free_b) continue;
      if (best == NONE || ent_before(D, meta, bmeta)) { best = j; bmeta = meta; }
    }
    if (best == NONE) break;
    taken |= 1u << best; used += (int)(overhead + sel4(D.msg_len, m_type(bmeta)));
  }
  for (uint32_t j = 0; j < n; j++) {
    if (!((taken >> j) & 1u)) continue;
    uint32_t meta = sq.meta(j);
    if (m_tr(meta) + 1 >= retransmit_limit) live &= ~(1u << j);            // Finished()
    else sq.meta(j) = m_pack(m_type(meta), m_tr(meta) + 1, m_seq(meta));
  }
  used_out = used;
  return taken;
}

// Compile-time variants keep the register footprint of the common case small (the kernel is latency
// bound, so waves per SIMD matter): KMAX = fan-out array size (4 or 8), SERF = user-event queue
// present, MULTI = records may leave this shard.
//
// A lane's work is a chain of dependent memory round trips, so independent loads are issued together:
//   trip 1  own node word + header            trip 3  subject node words (slot of each queued rumour)
//   trip 2  queue entries + 4 candidate peers  trip 4  the receivers' view records for the no-op filter
// TB = tile buckets (swim_device.h): rumours for nodes of this shard are not judged here; they go, unfiltered, to the bucket of the
// receiver's tile — s_tb / s_tbase: the block's histogram over the replica's tiles and the runs' bases in the buckets
template <int KMAX, bool SERF, bool MULTI, bool MASS, bool TB>
__device__ __forceinline__ void role_gossip(DevRef D, uint32_t r, uint32_t bx, uint4* lds_q, uint32_t* lds_stats, uint32_t* s_cnt, uint32_t* s_base, uint32_t* lds_exc, uint32_t* s_tb, uint32_t* s_tbase) {
  uint32_t t = *D.tick;
  uint32_t i = map_gossip(D, t % D.G, bx * SW_BLOCK + threadIdx.x);
  // a block is one stagger chunk of 256 consecutive nodes: if none of them has anything queued the
  // whole block retires after two words (the quiescent fast path of gossip(): "no broadcasts")
  uint32_t fb = NONE;
  if (D.fast_blocks) {
    uint32_t i_first = map_gossip(D, t % D.G, bx * SW_BLOCK);
    if (i_first == NONE) return;
    fb = (uint32_t)(((size_t)r * D.nloc + (i_first - D.i0)) / SW_BLOCK);
    if (!D.q_any[fb]) {
      if (threadIdx.x == 0) { uint32_t c = D.alive_cnt[fb]; if (c) atomicAdd(stat_ptr(D, ST_QUIESCENT), (unsigned long long)c); }
      return;
    }
  }
  ExcList X; X.stage(D, r, lds_exc);
  BlockStats S; S.init(lds_stats);
  if (threadIdx.x < SW_MAX_SHARDS) s_cnt[threadIdx.x] = 0;
  if (TB && threadIdx.x < SW_TB_BINS) s_tb[threadIdx.x] = 0;
  __syncthreads();

  uint4* sq = lds_q + threadIdx.x;                 // entry j at sq[j*256]
  uint4* se = lds_q + (size_t)D.Q * SW_BLOCK + threadIdx.x;
  constexpr bool serf = SERF;
  const uint32_t* nw = D.nw + (size_t)r * D.N;
  const bool filter = (D.flags & SWIM_F_FILTER_NOOP) != 0;

  uint32_t np = 0, peers[KMAX], pw[KMAX], sent_m[KMAX], sent_e[SERF ? KMAX : 1], loc[(MULTI || TB) ? KMAX : 1], psh[MULTI ? KMAX : 1];
  uint32_t qlen = 0, evqlen = 0, live_m = 0, live_e = 0, nq = 0, ne = 0;
  size_t l = 0; uint4 h = make_uint4(0, 0, 0, 0);
  bool active = false, quiet = false;
  uint32_t c_pkt = 0, c_drop = 0, c_filt = 0, c_s0 = 0, c_s1 = 0, c_s2 = 0, c_s3 = 0;   // per-lane tallies
  uint32_t wi = NW_DEAD;
  if (i != NONE) {                                  // trip 1: node word and header together
    l = (size_t)r * D.nloc + (i - D.i0);
    wi = nw[i]; h = HDR(l);
  }

  if (!(wi & NW_INERT)) {
    uint32_t k = i - D.i0;
    qlen = h_qlen(h.y); evqlen = h_evqlen(h.y);
    if (!qlen && !evqlen) quiet = true;
    else {
      active = true;
      size_t NL = (size_t)D.R * D.nloc;
      // trip 2: the queue entries and the first four peer candidates' node words, all independent
      uint4 e0 = qlen > 0 ? QENT(0u, l) : make_uint4(0, 0, 0, 0), e1 = qlen > 1 ? QENT(1u, l) : make_uint4(0, 0, 0, 0);
      uint32_t found;
      found = k_random_nodes<MASS>(D, r, i, k, t, SW_STREAM_GOSSIP, D.k_gossip < (uint32_t)KMAX ? D.k_gossip : (uint32_t)KMAX, 0, NONE, peers, pw, X);
      if (qlen > 0) sq[0] = e0;
      if (qlen > 1) sq[SW_BLOCK] = e1;
      for (uint32_t j = 2; j < qlen; j++) sq[j * SW_BLOCK] = QENT(j, l);
      if (serf) for (uint32_t j = 0; j < evqlen; j++) se[j * SW_BLOCK] = D.evq[(size_t)j * NL + l];
      // trip 3: where the subjects of the first two rumours keep their view columns
      uint32_t ws0 = (!TB && filter && qlen > 0) ? (X.usable() ? X.word(e0.x) : nw[e0.x]) : 0, ws1 = (!TB && filter && qlen > 1) ? (X.usable() ? X.word(e1.x) : nw[e1.x]) : 0;
      live_m = qlen >= 32 ? 0xFFFFFFFFu : (1u << qlen) - 1;
      live_e = evqlen >= 32 ? 0xFFFFFFFFu : (1u << evqlen) - 1;
      // per peer one GetBroadcasts() (LDS only)
      const uint32_t rl = retransmit_limit_n(D, est_n(D, r, l));
      uint32_t npk = 0;
      bool ok[KMAX];
      for (uint32_t p = 0; p < found; p++) {
        int used = 0, used2 = 0;
        uint32_t tm = get_broadcasts(D, LdsQ{sq}, qlen, live_m, 2, (int)D.budget, used, rl), te = 0;
        int avail = (int)D.budget - used;
        if (serf && avail > 2 + 1) te = get_broadcasts(D, LdsQ{se}, evqlen, live_e, 3, avail, used2, rl);
        if (!tm && !te) break;                       // "if len(msgs) == 0 { return }"
        c_pkt++;
        for (uint32_t m = tm; m; m &= m - 1) {
          uint32_t ty = m_type(sq[(__ffs(m) - 1) * SW_BLOCK].w);
          c_s0 += ty == SWIM_MSG_ALIVE; c_s1 += ty == SWIM_MSG_SUSPECT; c_s2 += ty == SWIM_MSG_DEAD;
        }
        c_s3 += (uint32_t)__popc(te);
        ok[p] = reach(D, r, t, wi, pw[p], i, p);
        if (!ok[p]) c_drop++;
        sent_m[p] = tm; if (SERF) sent_e[p] = te;
        npk = p + 1;
      }
      // trip 4: the no-op filter.  The receivers' home-slot entries for rumour 0 are fetched together (in the hot
      // case — every lane gossips about the same subject — that is the entry itself); rumours beyond the first take
      // the one-at-a-time path.
      if (TB && filter) {
        // the receivers judge for themselves (k_resolve's workgroup, against the same pre-tick views): nothing is read here.  Only a
        // packet for an attached node — captured for the transport bridge, never delivered — is judged by its sender.
        for (uint32_t p = 0; p < npk; p++) {
          if (!ok[p] || !(pw[p] & NW_ATTACHED) || (MULTI && peers[p] / D.nloc != D.rank)) continue;
          uint32_t tm = sent_m[p];
          const size_t lr = (size_t)r * D.nloc + (peers[p] - D.i0);
          for (uint32_t m = tm; m; m &= m - 1) {
            uint32_t j = __ffs(m) - 1; uint4 e = sq[j * SW_BLOCK];
            if (e.x == peers[p]) continue;
            if (noop_at_receiver<MASS>(D, r, lr, X.usable() ? X.word(e.x) : nw[e.x], e, false, e)) { tm &= ~(1u << j); c_filt++; }
          }
          sent_m[p] = tm;
        }
      }
      if (!TB && filter) {
        uint4 va0[KMAX]; const bool m0 = MASS && (ws0 & NW_MASS) != 0;                 // rumour 0's subject owns a row of the dense store: 4 bytes per receiver
        const uint32_t h0 = (ws0 & NW_SUBJECT) ? vt_home(D, e0.x) : 0, row0 = m0 ? D.mrow[(size_t)r * D.N + e0.x] : 0;
#pragma unroll
        for (int p = 0; p < KMAX; p++) {
          bool need = (uint32_t)p < npk && ok[p] && (sent_m[p] & 1u) && (!MULTI || peers[p] / D.nloc == D.rank) && e0.x != peers[p] && (ws0 & (MASS ? (NW_SUBJECT | NW_MASS) : NW_SUBJECT));
          va0[p] = !need ? make_uint4(0, 0, 0, 0) : m0 ? make_uint4(D.mA[m_idx(D, r, row0, peers[p] - D.i0)], 0, 0, 0) : D.vt[(size_t)h0 * NL + (size_t)r * D.nloc + (peers[p] - D.i0)];
        }
#pragma unroll
        for (int p = 0; p < KMAX; p++) {
          if ((uint32_t)p >= npk || !ok[p] || (MULTI && peers[p] / D.nloc != D.rank)) continue;
          uint32_t tm = sent_m[p];
          const size_t lr = (size_t)r * D.nloc + (peers[p] - D.i0);
          if ((tm & 1u) && e0.x != peers[p] && noop_at_receiver<MASS>(D, r, lr, ws0, e0, true, va0[p])) { tm &= ~1u; c_filt++; }
          for (uint32_t m = tm & ~1u; m; m &= m - 1) {
            uint32_t j = __ffs(m) - 1; uint4 e = sq[j * SW_BLOCK];
            if (e.x == peers[p]) continue;
            uint32_t ws = j == 1 ? ws1 : (X.usable() ? X.word(e.x) : nw[e.x]);
            if (noop_at_receiver<MASS>(D, r, lr, ws, e, false, e)) { tm &= ~(1u << j); c_filt++; }
          }
          sent_m[p] = tm;
        }
      }
      // keep the packets that still carry something
      for (uint32_t p = 0; p < npk; p++) {
        uint32_t tm = sent_m[p], te = SERF ? sent_e[p] : 0;
        if (!ok[p] || !(tm | te)) continue;
        if (pw[p] & NW_ATTACHED) {                   // Transport.WriteTo towards the real node
          for (uint32_t m = tm; m; m &= m - 1) { uint4 e = sq[(__ffs(m) - 1) * SW_BLOCK]; capture(D, i, r * D.N + peers[p], e.x, e.y, (m_type(e.w) << 30) | (e.z & 0x3FFFFFFFu)); }
          if (SERF) for (uint32_t m = te; m; m &= m - 1) { uint4 e = se[(__ffs(m) - 1) * SW_BLOCK]; capture(D, i, r * D.N + peers[p], e.x, e.y, (uint32_t)SWIM_MSG_USER << 30); }
          continue;
        }
        sent_m[np] = tm; peers[np] = peers[p];
        if (SERF) sent_e[np] = te;
        if (MULTI) psh[np] = peers[p] / D.nloc;
        np++;
      }
    }
  }
  S.count(ST_QUIESCENT, quiet); S.count(ST_ACTIVE, active);
  S.wave_add(ST_PKT_SENT, c_pkt); S.wave_add(ST_PKT_DROP, c_drop); S.wave_add(ST_FILTERED, c_filt);
  S.wave_add(ST_SENT0, c_s0); S.wave_add(ST_SENT1, c_s1); S.wave_add(ST_SENT2, c_s2); S.wave_add(ST_SENT3, c_s3);

  // ---- compaction of the block's packets into the outbound lists
  // (a) records for nodes of this shard: wavefront prefix sum of the per-lane counts, one LDS atomic per
  //     wave, block-private segment -> no global atomic and no barrier on the way out
  //     (TB: LDS offsets per receiving TILE, then one global atomicAdd per (block, tile) — like (b))
  const uint32_t segb = r * D.nb_gossip + bx;
  const uint32_t tl0 = TB ? (uint32_t)(((size_t)r * D.nloc) / SW_TB_TILE) : 0;          // the first tile lanes of this replica fall into
  uint32_t n_loc = 0;
#define PKT_SH(p) (MULTI ? psh[p] : D.rank)
#define PKT_N(p) ((uint32_t)(__popc(sent_m[p]) + (SERF ? __popc(sent_e[p]) : 0)))
#define PKT_BIN(p) ((uint32_t)(((size_t)r * D.nloc + (peers[p] - D.i0)) / SW_TB_TILE) - tl0)
  uint32_t my_off = 0;
  if (!TB) {
    for (uint32_t p = 0; p < np; p++) if (PKT_SH(p) == D.rank) n_loc += PKT_N(p);
    uint32_t incl = n_loc;
    if (__any(n_loc != 0)) {
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { uint32_t v = __shfl_up(incl, off); if (sw_lane() >= (uint32_t)off) incl += v; }
      uint32_t wave_total = __shfl(incl, 63), wbase = 0;
      if (sw_lane() == 63) wbase = atomicAdd(&s_cnt[D.rank], wave_total);
      wbase = __shfl(wbase, 63);
      my_off = wbase + incl - n_loc;
    }
  } else {
    for (uint32_t p = 0; p < np; p++) if (PKT_SH(p) == D.rank) loc[(MULTI || TB) ? p : 0] = atomicAdd(&s_tb[PKT_BIN(p)], PKT_N(p));
  }
  // (b) records for other shards: LDS offsets, then one global atomicAdd per (block, shard)
  if (MULTI) for (uint32_t p = 0; p < np; p++) if (PKT_SH(p) != D.rank) loc[(MULTI || TB) ? p : 0] = atomicAdd(&s_cnt[PKT_SH(p)], PKT_N(p));
  if (MULTI || TB) {
    __syncthreads();
    if (MULTI && threadIdx.x < D.n_shards && threadIdx.x != D.rank) {
      uint32_t c = s_cnt[threadIdx.x], b = 0;
      if (c) {
        b = atomicAdd(&D.out_cnt[threadIdx.x], c);
        if (b + c > D.out_cap_tab[threadIdx.x]) { atomicOr(D.err, SW_ERR_EDGE_OVF); b = NONE; }
        atomicAdd(stat_ptr(D, ST_EDGES_REMOTE), (unsigned long long)c);      // (edges: counted below — or by the receiving shard, SW_EDGE_JUDGE)
      }
      s_base[threadIdx.x] = b;
    }
    if (TB && threadIdx.x < SW_TB_BINS) {
      uint32_t c = s_tb[threadIdx.x], b = 0;
      if (c) {
        b = atomicAdd(&D.tb_cnt[tl0 + threadIdx.x], c);
        if (b + c > D.tb_cap) { atomicOr(D.err, SW_ERR_EDGE_OVF); b = NONE; }
        if (MULTI) *D.act = 1;
      }
      s_tbase[threadIdx.x] = b;
    }
    __syncthreads();
  }
  uint32_t c_e0 = 0;                                 // TB: records that are edges whatever the receiver holds (counted here, like a segment's);
                                                     // MULTI: the same of the records for other shards (the rest is judged and counted where it arrives)
  for (uint32_t p = 0; p < np; p++) {
    uint4* dst; bool bucket = false;
    if (PKT_SH(p) == D.rank) {
      if (TB) { uint32_t bin = PKT_BIN(p), b = s_tbase[bin]; if (b == NONE) continue; dst = D.tb + (size_t)(tl0 + bin) * D.tb_cap + b + loc[(MULTI || TB) ? p : 0]; bucket = true; }
      else { dst = D.seg + (size_t)segb * D.seg_cap + my_off; my_off += PKT_N(p); }
    }
    else { uint32_t b = s_base[PKT_SH(p)]; if (b == NONE) continue; dst = D.out_tab[PKT_SH(p)] + b + loc[(MULTI || TB) ? p : 0]; }
    uint32_t gdst = r * D.N + peers[p];
    for (uint32_t m = sent_m[p]; m; m &= m - 1) {
      uint4 e = sq[(__ffs(m) - 1) * SW_BLOCK];
      if (TB && bucket) {                            // the receiver's view decides (a rumour about the receiver itself is always delivered)
        const uint32_t cls = (filter && e.x != peers[p]) ? TB_GOSSIP : 0u;
        c_e0 += cls == 0;
        *dst++ = make_uint4(gdst, e.x, e.y, (m_type(e.w) << 30) | cls | (e.z & TB_FROM_MASK));
      } else if (MULTI && PKT_SH(p) != D.rank) {     // another shard's node: its shard judges (a rumour about the receiver itself is always delivered)
        const uint32_t jd = (filter && e.x != peers[p]) ? SW_EDGE_JUDGE : 0u;
        c_e0 += jd == 0;
        *dst++ = make_uint4(gdst, e.x, e.y, (m_type(e.w) << 30) | jd | (e.z & TB_FROM_MASK));
      } else
      *dst++ = make_uint4(gdst, e.x, e.y, (m_type(e.w) << 30) | (e.z & 0x3FFFFFFFu));
    }
    if (SERF)
      for (uint32_t m = sent_e[p]; m; m &= m - 1) {
        uint4 e = se[(__ffs(m) - 1) * SW_BLOCK];
        c_e0 += (TB && bucket) || (MULTI && PKT_SH(p) != D.rank);
        *dst++ = make_uint4(gdst, e.x, e.y, (uint32_t)SWIM_MSG_USER << 30);
      }
  }
  if (TB || MULTI) S.wave_add(ST_EDGES, c_e0);
#undef PKT_SH
#undef PKT_N
#undef PKT_BIN

  // ---- write the queues back, compacted; untouched entries are not rewritten
  if (active) {
    size_t NL = (size_t)D.R * D.nloc;
    for (uint32_t j = 0; j < qlen; j++) if ((live_m >> j) & 1u) { QENT(nq, l) = sq[j * SW_BLOCK]; nq++; }
    if (SERF) for (uint32_t j = 0; j < evqlen; j++) if ((live_e >> j) & 1u) { D.evq[(size_t)ne * NL + l] = se[j * SW_BLOCK]; ne++; }
    uint32_t hy = h_pack(h_leaving(h.y), nq, ne);
    if (hy != h.y) { h.y = hy; HDR(l) = h; }
  }
  {
    const bool drained = active && !(nq | ne);
    if (D.fast_blocks) q_bits_wave(D, l, false, drained); else q_bit_lane(D, l, false, drained);
  }
  // dead nodes keep their (frozen) queues: the hint stays up while any node of the block holds one
  bool holds = (nq | ne) != 0;
  if (i != NONE && (wi & NW_INERT) && D.fast_blocks) holds = (h_qlen(h.y) | h_evqlen(h.y)) != 0;
  int any = __syncthreads_or(holds);
  if (threadIdx.x == 0) {
    if (fb != NONE && !any) D.q_any[fb] = 0;
    uint32_t c = TB ? 0u : s_cnt[D.rank];          // every wave has added its total (barrier above)
    if (c) { D.seg_cnt[segb] = c; atomicAdd(stat_ptr(D, ST_EDGES), (unsigned long long)c); if (MULTI) *D.act = 1; }
  }
  S.flush(D);
}

// =================================================================================================
// roles: push-pull — memberlist pushPull/pushPullNode/mergeState (state.go), message based so it works
// across shards: the initiator sends, for every subject, the rumour mergeState would derive from its
// own view (Alive -> alive, Left -> dead{From: node}, Dead|Suspect -> suspect{From: receiver}; a
// remote Dead is never trusted directly) plus a pull request; the peer answers the same way next tick.
// =================================================================================================
// Every lane of the wave calls this together (`on` = the lane takes part); the records of a wave are
// appended with one atomic per destination shard, never one per record.
// the two records of a state exchange that are not explicit views (called by every lane of the wave together):
template <bool MULTI = true>
__device__ __forceinline__ void send_state_tail(DevRef D, bool on, uint32_t r, uint32_t owner, uint32_t dst, bool saw_dst, bool saw_self, uint32_t& c_edges, uint32_t& c_remote) {
  const uint32_t sh = on ? dst / D.nloc : 0;
  // the owner's view of ITSELF travels when the base row says something else about it (a node that has just
  // joined: nobody has heard of it; a node that came back after it was folded as dead)
  {
    bool want = false; uint4 rec = make_uint4(0, 0, 0, 0);
    if (on && !saw_self) {
      const uint32_t self = SW_KEY(HDR((size_t)r * D.nloc + (owner - D.i0)).x, SWIM_STATE_ALIVE);
      if (self != base_key_of(D, r, owner, D.nw[(size_t)r * D.N + owner])) { rec = mk_edge(D, r, dst, owner, SW_KINC(self), SWIM_MSG_ALIVE, 0); want = true; }
    }
    wave_append_sharded<MULTI>(D, want, sh, rec);
    c_edges += want; c_remote += want && sh != D.rank;
  }
  // ...and, with one exception to "what the base row says merges to nothing": the receiver's view of ITSELF is its own (it may
  // have been away while the base row moved on), so the owner's view of the receiver travels even when it is the base row's
  // (and not the trivial alive@1)
  {
    bool want = false; uint4 rec = make_uint4(0, 0, 0, 0);
    if (on && !saw_dst) {
      const uint32_t key = base_key_of(D, r, dst, D.nw[(size_t)r * D.N + dst]);
      if (key != SW_BASE_KEY && key >= 4u) {
        const uint32_t st = SW_KST(key), type = st == SWIM_STATE_ALIVE ? SWIM_MSG_ALIVE : st == SWIM_STATE_LEFT ? SWIM_MSG_DEAD : SWIM_MSG_SUSPECT;
        rec = mk_edge(D, r, dst, dst, SW_KINC(key), type, dst);     // dead{From: node} / suspect{From: receiver}: both = dst here
        want = true;
      }
    }
    wave_append_sharded<MULTI>(D, want, sh, rec);
    c_edges += want; c_remote += want && sh != D.rank;
  }
}
template <bool MASS, bool MULTI = true>
__device__ void send_state(DevRef D, bool on, uint32_t r, uint32_t owner, uint32_t dst, uint32_t& c_edges, uint32_t& c_remote, uint32_t& c_filt) {
  // what the base row says merges to nothing: only the owner's explicit views travel.  The lanes of the wave walk
  // their tables slot by slot together (wave_append_* is a wave-wide operation).
  const size_t NL = (size_t)D.R * D.nloc, lo = (size_t)r * D.nloc + (owner - D.i0);
  uint32_t left = on ? VMETA(lo).x : 0;
  bool saw_dst = false;
  const uint32_t sh = on ? dst / D.nloc : 0;
  const bool filter = (D.flags & SWIM_F_FILTER_NOOP) && sh == D.rank;
  bool saw_self = false;
  for (uint32_t sl = 0; sl < D.VT; sl++) {
    if (!__any(left != 0)) break;
    bool want = false; uint4 rec = make_uint4(0, 0, 0, 0);
    if (left) {
      uint4 a = D.vt[(size_t)sl * NL + lo];
      if (a.x != VT_EMPTY) {
        left--;
        uint32_t x = a.x, st = SW_KST(a.y), type, from = 0;
        saw_dst |= x == dst; saw_self |= x == owner;
        if (st == SWIM_STATE_ALIVE) type = SWIM_MSG_ALIVE;
        else if (st == SWIM_STATE_LEFT) { type = SWIM_MSG_DEAD; from = x; }
        else { type = SWIM_MSG_SUSPECT; from = dst; }
        want = true;
        if (filter && x != dst) {
          if (noop_at_receiver<MASS>(D, r, (size_t)r * D.nloc + (dst - D.i0), D.nw[(size_t)r * D.N + x], make_uint4(x, SW_KINC(a.y), from, type << 30), false, a)) { want = false; c_filt++; }
        }
        rec = mk_edge(D, r, dst, x, SW_KINC(a.y), type, from);
        if (MULTI && sh != D.rank && (D.flags & SWIM_F_FILTER_NOOP) && x != dst) rec.w |= SW_EDGE_JUDGE;     // the receiver's shard judges
      }
    }
    wave_append_sharded<MULTI>(D, want, sh, rec);
    c_edges += want && !(rec.w & SW_EDGE_JUDGE); c_remote += want && sh != D.rank;
  }
  // ...and the owner's pairs of the dense store: a walk over up to mass_rows rows per exchange.  Inside this launch a wave would
  // take its (up to 64) exchanges one after the other — the due nodes of a state-exchange tick sit in consecutive lanes, and at
  // 13 107 rows that was 26-49 ms per boundary tick on 18 busy waves, half of config #4's wall time — so the exchange is only
  // LISTED here and k_send_mass, right after this launch, gives every listed exchange a wave of its own.  The two trailing
  // records below depend on what the row walk finds, so they move there too.
  if (MASS && D.M) {
    const uint64_t mask = __ballot(on);
    if (mask) {
      const uint32_t lane = sw_lane(), leader = (uint32_t)__ffsll((long long)mask) - 1;
      uint32_t base = 0;
      if (lane == leader) base = atomicAdd(D.xs_cnt, (uint32_t)__popcll(mask));
      base = __shfl(base, leader);
      if (on) {
        const uint32_t pos = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1));
        if (pos < D.xs_cap) D.xs_list[pos] = make_uint4(r, owner, dst, (saw_dst ? 1u : 0u) | (saw_self ? 2u : 0u));
        else atomicOr(D.err, SW_ERR_PEND_OVF);
      }
    }
    return;
  }
  send_state_tail<MULTI>(D, on, r, owner, dst, saw_dst, saw_self, c_edges, c_remote);
}
template <bool MASS, bool MULTI>
__device__ __forceinline__ void role_pushpull(DevRef D, uint32_t r, uint32_t a, uint32_t* lds_stats, uint32_t* lds_exc) {
  uint32_t t = *D.tick;
  if (t % D.P) return;                              // exchanges start on probe-interval boundaries only
  ExcList X; X.stage(D, r, lds_exc);
  BlockStats S; S.init(lds_stats);
  // lane a -> (window offset, j-th node due in that tick); everything due within the next P ticks goes now
  uint32_t grp = D.P < D.pp_period ? D.P : D.pp_period, off = a % grp;
  uint64_t i64 = (uint64_t)((t + off) % D.pp_period) + (uint64_t)(a / grp) * D.pp_period;
  const uint32_t* nw = D.nw + (size_t)r * D.N;
  bool go = false; uint32_t o = 0, p = 0;
  if (i64 < D.N) {
    o = (uint32_t)i64;
    if (o >= D.i0 && o < D.i0 + D.nloc) {
      uint32_t wo = nw[o], wp;
      if (!(wo & NW_INERT) && k_random_nodes<MASS>(D, r, o, o - D.i0, t, SW_STREAM_PUSHPULL, 1, 1, NONE, &p, &wp, X))
        go = !(wp & NW_DEAD) && NW_PART(wo) == NW_PART(wp);          // else the TCP dial fails
    }
  }
  uint32_t c_edges = 0, c_remote = 0, c_filt = 0;
  send_state<MASS, MULTI>(D, go, r, o, p, c_edges, c_remote, c_filt);
  uint32_t sh = go ? p / D.nloc : 0;
  wave_append_sharded<MULTI>(D, go, sh, mk_edge(D, r, p, SWIM_SUBJECT_PULL, o, SWIM_MSG_ALIVE, 0));
  c_edges += go; c_remote += go && sh != D.rank;
  S.count(ST_PUSHPULLS, go);
  S.wave_add(ST_EDGES, c_edges); S.wave_add(ST_EDGES_REMOTE, c_remote); S.wave_add(ST_FILTERED, c_filt);
  S.flush(D);
}
// pull requests are filed in 64 sub-lists (k_resolve picks one by block) so that no counter is hot
#define SW_PP_LISTS 64
template <bool MASS, bool MULTI>
__device__ __forceinline__ void role_ppreply(DevRef D, uint32_t b, uint32_t nb, uint32_t* lds_stats) {
  uint32_t t = *D.tick, li = t & 1u;
  if (t == 0) return;                               // (requests of scheduled exchanges exist the tick after a boundary; a join asks any time)
  BlockStats S; S.init(lds_stats);
  uint32_t sub_cap = D.pp_cap / SW_PP_LISTS, c_edges = 0, c_remote = 0, c_filt = 0;
  for (uint32_t sub = b; sub < SW_PP_LISTS; sub += nb) {
    uint32_t n = D.pp_cnt[(li * SW_PP_LISTS + sub) * 16]; if (n > sub_cap) n = sub_cap;
    for (uint32_t e0 = 0; e0 < n; e0 += SW_BLOCK) {
      uint32_t e = e0 + threadIdx.x; bool on = e < n, jn = false; uint32_t r = 0, p = 0, o = 0, pclk = 0;
      if (on) {
        uint2 rq = D.pp_list[((size_t)li * SW_PP_LISTS + sub) * sub_cap + e];
        r = rq.x / D.nloc; p = D.i0 + rq.x % D.nloc; o = rq.y & 0x7FFFFFFFu; jn = (rq.y >> 31) != 0;
        on = !(D.nw[(size_t)r * D.N + p] & NW_INERT);
        if (on && jn && D.sslt) pclk = HDR(rq.x).w;
      }
      send_state<MASS, MULTI>(D, on, r, p, o, c_edges, c_remote, c_filt);
      // serf.Join: the state exchange hands over serf's own push-pull message too (MergeRemoteState: clock.Witness(LTime - 1)), THEN the joiner
      // calls broadcastJoin(s.clock.Time()) — its join intent is stamped with the clock of the member it joined through.  The reply to a
      // JOIN's pull request therefore carries that intent, ready-stamped, to the joiner, which witnesses it, applies it to its own entry
      // and broadcasts it (a message like any other: the same across shards).
      if (D.sslt) {
        const bool want = on && jn; const uint32_t sh = want ? o / D.nloc : 0;
        wave_append_sharded<MULTI>(D, want, sh, make_uint4(r * D.N + o, SWIM_INTENT_JOIN | o, pclk, (uint32_t)SWIM_MSG_USER << 30));
        c_edges += want; c_remote += want && sh != D.rank;
      }
    }
  }
  S.wave_add(ST_EDGES, c_edges); S.wave_add(ST_EDGES_REMOTE, c_remote); S.wave_add(ST_FILTERED, c_filt);
  S.flush(D);
}

// role: join — swim_inject_join: the join push-pull (pushPullNode(join=true)) of the nodes started since the last tick:
// their state to `via` and a pull request; via answers through the ordinary reply list one tick later
template <bool MASS, bool MULTI>
__device__ __forceinline__ void role_join(DevRef D, uint32_t* lds_stats) {
  const uint32_t n = *D.join_cnt < D.join_cap ? *D.join_cnt : D.join_cap;
  if (!n) return;
  BlockStats S; S.init(lds_stats);
  uint32_t c_edges = 0, c_remote = 0, c_filt = 0;
  for (uint32_t e0 = 0; e0 < n; e0 += SW_BLOCK) {
    const uint32_t e = e0 + threadIdx.x; bool on = false; uint32_t r = 0, o = 0, p = 0;
    if (e < n) {
      const uint2 j = D.join_list[e];
      r = j.x / D.N; o = j.x % D.N; p = j.y;
      const uint32_t wo = D.nw[j.x], wp = D.nw[(size_t)r * D.N + p];
      if (o >= D.i0 && o < D.i0 + D.nloc && !(wo & (NW_DEAD | NW_ATTACHED)) && (wo & NW_ALONE)) {   // (every shard lists every joiner)
        on = p != o && !(wp & NW_DEAD) && NW_PART(wo) == NW_PART(wp);     // else the join fails (memberlist.Join returns an error)
        S.add(on ? ST_JOINS : ST_JOIN_FAIL);
      }
    }
    send_state<MASS, MULTI>(D, on, r, o, p, c_edges, c_remote, c_filt);
    const uint32_t sh = on ? p / D.nloc : 0;
    wave_append_sharded<MULTI>(D, on, sh, mk_edge(D, r, p, SWIM_SUBJECT_PULL, o, SWIM_MSG_ALIVE, 1));      // from = 1: a JOIN's pull request (serf.Join: the reply brings the join intent)
    c_edges += on; c_remote += on && sh != D.rank;
  }
  S.wave_add(ST_EDGES, c_edges); S.wave_add(ST_EDGES_REMOTE, c_remote); S.wave_add(ST_FILTERED, c_filt);
  if (D.n_shards > 1 && threadIdx.x == 0) *D.act = 1;
  S.flush(D);
}

// =================================================================================================
// role: carry (sharded runs only) — the broadcasts k_resolve piggy-backed last tick sit in the blocks'
// private areas; the ones addressed to other shards move to those shards' lists and are voided in place,
// the rest is delivered (and filtered) by k_deliver like in an unsharded run.
// =================================================================================================
__device__ __forceinline__ void role_carry(DevRef D, uint32_t b, uint32_t nb, uint32_t* lds_stats) {
  if (*D.carry_stamp != *D.tick) return;            // nothing was piggy-backed last tick
  BlockStats S; S.init(lds_stats);
  const uint32_t par = *D.tick & 1u;
  uint32_t c_rem = 0, c_rem0 = 0;                    // records that left for other shards; those of them that are edges whatever the receiver holds
  for (uint32_t a = b; a < D.NB; a += nb) {
    uint32_t n = D.carry_cl[a].x; if (n > D.carry_cap) n = D.carry_cap;
    uint4* area = D.carry + ((size_t)par * D.NB + a) * D.carry_cap;
    for (uint32_t e0 = 0; e0 < n; e0 += SW_BLOCK) {
      uint32_t e = e0 + threadIdx.x; bool rem = false; uint4 rec = make_uint4(0, 0, 0, 0); uint32_t sh = 0;
      if (e < n) {
        rec = area[e]; sh = (rec.x % D.N) / D.nloc; rem = sh != D.rank;
        if (rem && (D.flags & SWIM_F_FILTER_NOOP) && (rec.w >> 30) != SWIM_MSG_USER && rec.y != rec.x % D.N) rec.w |= SW_EDGE_JUDGE;   // what deliver_carried would judge here
      }
      wave_append_sharded(D, rem, sh, rec);
      if (rem) { area[e].x = SW_DST_VOID; c_rem++; c_rem0 += !(rec.w & SW_EDGE_JUDGE); }
    }
  }
  S.wave_add(ST_EDGES, c_rem0); S.wave_add(ST_EDGES_REMOTE, c_rem);
  S.flush(D);
}

// role: carry with tile buckets (one shard or many) — a workgroup takes SW_CARRY_GROUP consecutive carry areas (their records are
// addressed to nodes of the senders' replica), counts them per receiving tile in LDS, reserves one run per tile with one global
// atomic and files them in the tiles' buckets: the receivers' workgroups judge them in k_resolve like the gossip role's rumours —
// k_deliver's read of the receiver's view per carried record, a random access each, is gone.  Records for other shards move to
// those shards' lists and are voided in place (swim_debug_edges reads the areas).
template <bool MULTI>
__device__ __forceinline__ void role_carry_tb(DevRef D, uint32_t b, uint32_t* lds_stats, uint32_t* s_tb, uint32_t* s_tbase) {
  const uint32_t t = *D.tick;
  if (*D.carry_stamp != t) return;                  // nothing was piggy-backed last tick
  BlockStats S; S.init(lds_stats);
  const uint32_t par = t & 1u, a0 = b * SW_CARRY_GROUP, a1 = a0 + SW_CARRY_GROUP < D.NB ? a0 + SW_CARRY_GROUP : D.NB;
  if (b == 0 && threadIdx.x == 0) D.carry_stamp[1] = t;            // swim_debug_edges: the areas' `last` words are of this tick
  const bool filter = (D.flags & SWIM_F_FILTER_NOOP) != 0;
  const uint32_t tl0 = (uint32_t)(((size_t)div_nloc(D, (size_t)a0 * SW_BLOCK) * D.nloc) / SW_TB_TILE);   // first tile of the replica the group starts in
  if (threadIdx.x < SW_TB_BINS) s_tb[threadIdx.x] = 0;
  __syncthreads();
  uint32_t c_rem = 0, c_rem0 = 0;
  for (uint32_t a = a0; a < a1; a++) {              // pass 1: count per tile; what leaves the shard leaves now
    uint32_t n = D.carry_cl[a].x; if (n > D.carry_cap) { if (threadIdx.x == 0) atomicOr(D.err, SW_ERR_CARRY_OVF); n = D.carry_cap; }
    uint4* area = D.carry + ((size_t)par * D.NB + a) * D.carry_cap;
    for (uint32_t e0 = 0; e0 < n; e0 += SW_BLOCK) {
      const uint32_t e = e0 + threadIdx.x; bool rem = false; uint4 rec = make_uint4(0, 0, 0, 0); uint32_t sh = 0;
      if (e < n) {
        rec = area[e];
        if (MULTI) {
          sh = mod_n(D, rec.x) / D.nloc; rem = sh != D.rank;
          if (rem && filter && (rec.w >> 30) != SWIM_MSG_USER && rec.y != mod_n(D, rec.x)) rec.w |= SW_EDGE_JUDGE;
        }
      }
      if (MULTI) { wave_append_sharded(D, rem, sh, rec); if (rem) { area[e].x = SW_DST_VOID; c_rem++; c_rem0 += !(rec.w & SW_EDGE_JUDGE); } }
      if (e < n && !rem) {
        const uint32_t bin = (uint32_t)(((size_t)div_n(D, rec.x) * D.nloc + (mod_n(D, rec.x) - D.i0)) / SW_TB_TILE) - tl0;
        if (bin < SW_TB_BINS) atomicAdd(&s_tb[bin], 1u);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < SW_TB_BINS) {
    uint32_t c = s_tb[threadIdx.x], base = 0;
    if (c) {
      base = atomicAdd(&D.tb_cnt[tl0 + threadIdx.x], c);
      if (base + c > D.tb_cap) { atomicOr(D.err, SW_ERR_EDGE_OVF); base = NONE; }
    }
    s_tbase[threadIdx.x] = base; s_tb[threadIdx.x] = 0;          // (s_tb: the runs' fill cursors from here on)
  }
  __syncthreads();
  for (uint32_t a = a0; a < a1; a++) {              // pass 2: file (the areas come out of L2 this time)
    uint32_t n = D.carry_cl[a].x; if (n > D.carry_cap) n = D.carry_cap;
    const uint4* area = D.carry + ((size_t)par * D.NB + a) * D.carry_cap;
    for (uint32_t e = threadIdx.x; e < n; e += SW_BLOCK) {
      uint4 rec = area[e];
      if (rec.x == SW_DST_VOID) continue;
      const uint32_t x = mod_n(D, rec.x), type = rec.w >> 30;
      const uint32_t tile = (uint32_t)(((size_t)div_n(D, rec.x) * D.nloc + (x - D.i0)) / SW_TB_TILE), bin = tile - tl0;
      rec.w = (rec.w & ~TB_CLASS_MASK) | ((filter && type != SWIM_MSG_USER && rec.y != x) ? TB_CARRIED : TB_CARRIED_PLAIN);
      if (bin < SW_TB_BINS) {
        const uint32_t base = s_tbase[bin];
        if (base != NONE) D.tb[(size_t)tile * D.tb_cap + base + atomicAdd(&s_tb[bin], 1u)] = rec;
      } else {                                       // (a group that straddles replicas of very many tiles: a global atomic per record)
        const uint32_t pos = atomicAdd(&D.tb_cnt[tile], 1u);
        if (pos < D.tb_cap) D.tb[(size_t)tile * D.tb_cap + pos] = rec; else atomicOr(D.err, SW_ERR_EDGE_OVF);
      }
    }
  }
  __syncthreads();                                   // everybody has read the counts before they are cleared
  if (threadIdx.x < a1 - a0) {
    const uint2 c = D.carry_cl[a0 + threadIdx.x];
    if (c.x | c.y) D.carry_cl[a0 + threadIdx.x] = make_uint2(0, c.x < D.carry_cap ? c.x : D.carry_cap);
  }
  if (MULTI) { S.wave_add(ST_EDGES, c_rem0); S.wave_add(ST_EDGES_REMOTE, c_rem); }
  S.flush(D);
}


// =================================================================================================
// serf/coordinate — Vivaldi network coordinates (SWIM_F_COORDINATES).  serf v0.10.4
// coordinate/{config,coordinate,client}.go and ping_delegate.go restated for one lane; in-tree pin:
// librtt.ComputeDistance (internal/gossip/librtt/rtt.go:16-22).  f64 throughout, every operation through a
// round-to-nearest intrinsic so that nothing is contracted into an FMA: Go rounds after every operation, and so
// does the checker.  No MFMA: 8-wide vectors, one update per probing node and second.
// =================================================================================================
// hipcc's __dmul_rn/__dadd_rn are plain '*' and '+': without this the compiler still fuses them into v_fma_f64 (-ffp-contract=
// fast-honor-pragmas is the HIP default), which is what made the first device run differ from Go-style rounding by 1-3 ulp.
// File scope from here on; everything below that is not a coordinate is integer work.  lib.py also passes -ffp-contract=off.
#pragma clang fp contract(off)
#define SW_VIVALDI_ERROR_MAX 1.5
#define SW_VIVALDI_CE 0.25
#define SW_VIVALDI_CC 0.25
#define SW_COORD_HEIGHT_MIN 10.0e-6
#define SW_COORD_GRAVITY_RHO 150.0
#define SW_COORD_ZERO 1.0e-6
__device__ __forceinline__ double cmul(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double cadd(double a, double b) { return __dadd_rn(a, b); }
__device__ __forceinline__ double csub(double a, double b) { return __dsub_rn(a, b); }
__device__ __forceinline__ double cdiv_(double a, double b) { return __ddiv_rn(a, b); }
__device__ __forceinline__ void coord_fresh(swim_coordinate& c) {     // NewCoordinate
#pragma unroll
  for (int i = 0; i < SWIM_COORD_DIMS; i++) c.vec[i] = 0.0;
  c.error = SW_VIVALDI_ERROR_MAX; c.adjustment = 0.0; c.height = SW_COORD_HEIGHT_MIN;
}
__device__ __forceinline__ bool coord_valid(const swim_coordinate& c) {
  bool ok = isfinite(c.error) && isfinite(c.adjustment) && isfinite(c.height);
#pragma unroll
  for (int i = 0; i < SWIM_COORD_DIMS; i++) ok = ok && isfinite(c.vec[i]);
  return ok;
}
__device__ __forceinline__ double vec_magnitude(const double* v) {
  double sum = 0.0;
#pragma unroll
  for (int i = 0; i < SWIM_COORD_DIMS; i++) sum = cadd(sum, cmul(v[i], v[i]));
  return __dsqrt_rn(sum);
}
__device__ __forceinline__ double coord_raw_distance(const swim_coordinate& a, const swim_coordinate& b) {
  double d[SWIM_COORD_DIMS];
#pragma unroll
  for (int i = 0; i < SWIM_COORD_DIMS; i++) d[i] = csub(a.vec[i], b.vec[i]);
  return cadd(cadd(vec_magnitude(d), a.height), b.height);
}
// DistanceTo(...).Seconds(): through time.Duration (int64 nanoseconds, truncated) and back
__device__ __forceinline__ double coord_distance_seconds(const swim_coordinate& a, const swim_coordinate& b) {
  double dist = coord_raw_distance(a, b);
  const double adjusted = cadd(cadd(dist, a.adjustment), b.adjustment);
  if (adjusted > 0.0) dist = adjusted;
  const long long ns = (long long)cmul(dist, 1.0e9);
  return cadd((double)(ns / 1000000000ll), cdiv_((double)(ns % 1000000000ll), 1e9));
}
// unitVectorAt: direction from b to a; coincident points get a random one (rand.Float64() - 0.5 per dimension)
__device__ __forceinline__ double coord_unit_vector(DevRef D, uint32_t r, uint32_t o, uint32_t t, uint32_t salt, const double* a, const double* b, double* unit) {
#pragma unroll
  for (int i = 0; i < SWIM_COORD_DIMS; i++) unit[i] = csub(a[i], b[i]);
  double mag = vec_magnitude(unit);
  if (mag > SW_COORD_ZERO) {
    const double inv = cdiv_(1.0, mag);
#pragma unroll
    for (int i = 0; i < SWIM_COORD_DIMS; i++) unit[i] = cmul(unit[i], inv);
    return mag;
  }
  SwDraws d; d.init(seed_of(D, r), SW_STREAM_COORD, t, o);
#pragma unroll
  for (int i = 0; i < SWIM_COORD_DIMS; i++) unit[i] = csub(cdiv_((double)d.get(salt * SWIM_COORD_DIMS + (uint32_t)i), 4294967296.0), 0.5);
  mag = vec_magnitude(unit);
  if (mag > SW_COORD_ZERO) {
    const double inv = cdiv_(1.0, mag);
#pragma unroll
    for (int i = 0; i < SWIM_COORD_DIMS; i++) unit[i] = cmul(unit[i], inv);
    return 0.0;
  }
#pragma unroll
  for (int i = 0; i < SWIM_COORD_DIMS; i++) unit[i] = 0.0;
  unit[0] = 1.0;
  return 0.0;
}
__device__ __forceinline__ void coord_apply_force(DevRef D, uint32_t r, uint32_t o, uint32_t t, uint32_t salt, swim_coordinate& c, double force, const swim_coordinate& other) {
  double unit[SWIM_COORD_DIMS];
  const double mag = coord_unit_vector(D, r, o, t, salt, c.vec, other.vec, unit);
#pragma unroll
  for (int i = 0; i < SWIM_COORD_DIMS; i++) c.vec[i] = cadd(c.vec[i], cmul(unit[i], force));
  if (mag > SW_COORD_ZERO) {
    c.height = cadd(cdiv_(cmul(cadd(c.height, other.height), force), mag), c.height);
    if (!(c.height >= SW_COORD_HEIGHT_MIN) && !isnan(c.height)) c.height = SW_COORD_HEIGHT_MIN;   // math.Max: NaN stays NaN
  }
}
// the latency model: hidden position and access-link height of a node, microseconds (swimsim.h, swim_config.rtt_*)
__device__ __forceinline__ void rtt_truth_of(DevRef D, uint32_t r, uint32_t i, uint32_t pos[3], uint32_t& h) {
  uint32_t w[4]; const uint64_t sr = seed_of(D, r);
  sw_philox(i, 0, 0, 0x54525554u, (uint32_t)sr, (uint32_t)(sr >> 32) ^ SW_STREAM_TRUTH, w);
#pragma unroll
  for (int k = 0; k < 3; k++) pos[k] = (uint32_t)(((uint64_t)w[k] * D.rtt_scale_us) >> 32);
  h = (uint32_t)(((uint64_t)w[3] * D.rtt_height_us) >> 32);
}
__device__ __forceinline__ uint32_t rtt_between(DevRef D, uint32_t r, uint32_t a, uint32_t b) {
  uint32_t pa[3], pb[3], ha, hb; rtt_truth_of(D, r, a, pa, ha); rtt_truth_of(D, r, b, pb, hb);
  double sum = 0.0;
#pragma unroll
  for (int k = 0; k < 3; k++) { const double d = csub((double)pa[k], (double)pb[k]); sum = cadd(sum, cmul(d, d)); }
  return (uint32_t)__dsqrt_rn(sum) + ha + hb;
}
// Client.latencyFilter: the median of the last LatencyFilterSize round-trip times seen from this peer (microseconds)
__device__ __forceinline__ uint32_t coord_latency_filter(DevRef D, size_t l, uint32_t t, uint32_t peer, uint32_t rtt_us) {
  uint4* tab = D.c_lf + l * SW_COORD_PEERS * 2;
  uint32_t hit = NONE, vic = 0, vic_last = 0; bool vic_free = false;
  for (uint32_t j = 0; j < SW_COORD_PEERS; j++) {
    const uint4 a = tab[2 * j]; const uint32_t last = tab[2 * j + 1].y;
    if (a.y && a.x == peer) { hit = j; break; }
    if (vic_free) continue;
    if (!a.y) { vic = j; vic_free = true; } else if (j == 0 || last < vic_last) { vic = j; vic_last = last; }
  }
  uint32_t n = 0, s0 = 0, s1 = 0, s2 = 0;
  if (hit != NONE) { const uint4 a = tab[2 * hit]; n = a.y; s0 = a.z; s1 = a.w; s2 = tab[2 * hit + 1].x; }
  else hit = vic;
  if (n == SW_COORD_FILTER) { s0 = s1; s1 = s2; n--; }
  if (n == 0) s0 = rtt_us; else if (n == 1) s1 = rtt_us; else s2 = rtt_us;
  n++;
  tab[2 * hit] = make_uint4(peer, n, s0, s1); tab[2 * hit + 1] = make_uint4(s2, t, 0, 0);
  // sorted[n / 2]
  if (n == 1) return s0;
  if (n == 2) return s0 > s1 ? s0 : s1;
  const uint32_t lo = s0 < s1 ? s0 : s1, hi = s0 < s1 ? s1 : s0;
  return s2 < lo ? lo : (s2 > hi ? hi : s2);
}
// serf pingDelegate.NotifyPingComplete -> Client.Update(other, coord, rtt) for every prober k_begin listed: the ack's payload
// is the acker's coordinate as of the START of this tick (D.coord is only written by k_coord_commit)
__global__ void __launch_bounds__(SW_BLOCK) k_coord_update(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  const uint32_t e = blockIdx.x * SW_BLOCK + threadIdx.x, n = *D.c_cnt < D.c_cap ? *D.c_cnt : D.c_cap;
  bool upd = false, reset = false;
  if (e < n) {
    const uint2 ent = D.c_list[e];
    const size_t l = ent.x; const uint32_t x = ent.y, r = div_nloc(D, l), o = D.i0 + mod_nloc(D, l), t = *D.tick;
    const swim_coordinate other = D.coord[(size_t)r * D.nloc + (x - D.i0)];
    swim_coordinate c = D.coord[l];
    uint32_t rtt_us = rtt_between(D, r, o, x);
    if (D.rtt_jitter_us) {
      uint32_t w[4]; const uint64_t sr = seed_of(D, r);
      sw_philox(t, o, 0, 0x52545431u, (uint32_t)sr, (uint32_t)(sr >> 32) ^ SW_STREAM_RTT, w);
      rtt_us += (uint32_t)(((uint64_t)w[0] * D.rtt_jitter_us) >> 32);
    }
    upd = true;
    const uint32_t med_us = coord_latency_filter(D, l, t, x, rtt_us);
    const double rtt = cdiv_((double)((unsigned long long)med_us * 1000ull), 1e9);
    {   // updateVivaldi
      const double dist = coord_distance_seconds(c, other);
      const double rs = rtt < SW_COORD_ZERO ? SW_COORD_ZERO : rtt;
      const double wrongness = cdiv_(fabs(csub(dist, rs)), rs);
      double total = cadd(c.error, other.error); if (total < SW_COORD_ZERO) total = SW_COORD_ZERO;
      const double weight = cdiv_(c.error, total);
      c.error = cadd(cmul(cmul(SW_VIVALDI_CE, weight), wrongness), cmul(c.error, csub(1.0, cmul(SW_VIVALDI_CE, weight))));
      if (c.error > SW_VIVALDI_ERROR_MAX) c.error = SW_VIVALDI_ERROR_MAX;
      const double delta = cmul(SW_VIVALDI_CC, weight), force = cmul(delta, csub(rs, dist));
      coord_apply_force(D, r, o, t, 0, c, force, other);
    }
    {   // updateAdjustment
      const double dist = coord_raw_distance(c, other);
      double* adj = D.c_adj + l * SW_COORD_WINDOW;
      const uint32_t idx = D.c_adj_idx[l];
      adj[idx] = csub(rtt, dist); D.c_adj_idx[l] = (idx + 1) % SW_COORD_WINDOW;
      double sum = 0.0;
      for (int i = 0; i < SW_COORD_WINDOW; i++) sum = cadd(sum, adj[i]);
      c.adjustment = cdiv_(sum, cmul(2.0, (double)SW_COORD_WINDOW));
    }
    {   // updateGravity
      swim_coordinate origin; coord_fresh(origin);
      const double dist = coord_distance_seconds(origin, c), q = cdiv_(dist, SW_COORD_GRAVITY_RHO), force = cmul(-1.0, cmul(q, q));
      coord_apply_force(D, r, o, t, 1, c, force, origin);
    }
    if (!coord_valid(c)) { reset = true; coord_fresh(c); }
    D.c_new[e] = c;
  }
  const uint64_t mu = __ballot(upd), mr = __ballot(reset);
  if (sw_lane() == 0) {
    if (mu) atomicAdd(stat_ptr(D, ST_COORD_UPD), (unsigned long long)__popcll(mu));
    if (mr) atomicAdd(stat_ptr(D, ST_COORD_RESET), (unsigned long long)__popcll(mr));
  }
}
__global__ void __launch_bounds__(SW_BLOCK) k_coord_commit(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  const uint32_t e = blockIdx.x * SW_BLOCK + threadIdx.x, n = *D.c_cnt < D.c_cap ? *D.c_cnt : D.c_cap;
  if (e < n) D.coord[D.c_list[e].x] = D.c_new[e];
}
// a fresh process (swim_inject_join): a fresh coordinate client
__device__ __forceinline__ void coord_reset_lane(DevRef D, size_t l) {
  swim_coordinate c; coord_fresh(c); D.coord[l] = c;
  for (int i = 0; i < SW_COORD_WINDOW; i++) D.c_adj[l * SW_COORD_WINDOW + i] = 0.0;
  D.c_adj_idx[l] = 0;
  for (int j = 0; j < SW_COORD_PEERS * 2; j++) D.c_lf[l * SW_COORD_PEERS * 2 + j] = make_uint4(0, 0, 0, 0);
}
__global__ void k_coord_init(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  const size_t NL = (size_t)D.R * D.nloc, l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l < NL) coord_reset_lane(D, l);
}

// =================================================================================================
// k_begin — the fused first launch of a tick.  grid = nb_expire + nb_pend + R*(nb_probe + nb_gossip);
// dynamic LDS = (Q+EQ) * 256 * 16 bytes (the gossip role's staged queues)
// =================================================================================================
#ifndef SW_BEGIN_WAVES
#define SW_BEGIN_WAVES 5      /* 96 VGPRs, no scratch: one more wave per SIMD than the allocator would settle for */
#endif
#ifdef SWIMSIM_WAVECLK
// diagnostics (-DSWIMSIM_WAVECLK builds only): entry and exit time of every workgroup of k_begin (by role) and k_deliver
#define BCLK_ROWS 32768
__device__ unsigned long long g_bclk[2][BCLK_ROWS][4];
#define BCLK_OUT(k, t_in_, tag) do { if (threadIdx.x == 0) { unsigned long long* row_ = g_bclk[k][blockIdx.x % BCLK_ROWS]; row_[0] = (t_in_); row_[1] = wall_clock64(); row_[2] = (tag); row_[3] = blockIdx.x; } } while (0)
#endif
template <int KMAX, bool SERF, bool MULTI, bool MASS, bool TB>
__global__ void __launch_bounds__(SW_BLOCK) __attribute__((amdgpu_waves_per_eu(SW_BEGIN_WAVES, 8))) k_begin(const SwDev* __restrict__ Dp, BeginPlan pl) {
  SW_DEV_BIND
  extern __shared__ uint4 lds_q[];
  __shared__ uint32_t lds_stats[ST_COUNT];
  __shared__ uint32_t s_cnt[SW_MAX_SHARDS], s_base[SW_MAX_SHARDS], lds_exc[2 * SW_EXC_MAX];
  __shared__ uint32_t s_tb[TB ? SW_TB_BINS : 1], s_tbase[TB ? SW_TB_BINS : 1];
  uint32_t b = blockIdx.x;
#ifdef SWIMSIM_WAVECLK
  const unsigned long long t_in = wall_clock64();
#define ROLE_DONE(id) BCLK_OUT(0, t_in, id)
#elif defined(SWIMSIM_DIAG)
  // diagnostics (SWIMSIM_ROLECLK): when did the first block of a role start, when did its last block end
  const unsigned long long t_in = D.role_clk ? wall_clock64() : 0;
#define ROLE_DONE(id) do { if (D.role_clk && threadIdx.x == 0) { uint32_t tk = *D.tick; if (tk < D.role_clk_ticks) { \
    unsigned long long* c = D.role_clk + (((size_t)tk * 8 + (id)) * 64 + (blockIdx.x & 63u)) * 2; atomicMin(c, t_in); atomicMax(c + 1, (unsigned long long)wall_clock64()); } } } while (0)
#else
#define ROLE_DONE(id) do { } while (0)
#endif
  if (b < pl.nb_expire) { if (pl.roles & 1u) role_expire(D, b, pl.nb_expire); ROLE_DONE(0); return; }
  b -= pl.nb_expire;
  if (b < pl.nb_pend) { if (pl.roles & 2u) role_pending<KMAX, MULTI, MASS>(D, b, pl.nb_pend, lds_stats, MULTI ? *D.peer_act : 1u); ROLE_DONE(1); return; }
  b -= pl.nb_pend;
  // (an XCD-aware block order — each XCD working on R/8 of the clusters, so that the gossip role's random reads of
  // receivers' views share an L2 — was measured: no faster once every cluster is busy, slower while only some are, because
  // the load then sits on a few XCDs; profiles/r02_ab_begin.txt)
  if (b < D.R * pl.nb_probe) { if (pl.roles & 4u) role_probe<MULTI, MASS, TB>(D, b / pl.nb_probe, b % pl.nb_probe, (b % pl.nb_probe) * SW_BLOCK + threadIdx.x, lds_stats, lds_exc, s_cnt, MULTI ? *D.peer_act : 1u, s_tb, s_tbase); ROLE_DONE(2); return; }
  b -= D.R * pl.nb_probe;
  if (b < D.R * pl.nb_gossip) { if (pl.roles & 8u) role_gossip<KMAX, SERF, MULTI, MASS, TB>(D, b / pl.nb_gossip, b % pl.nb_gossip, lds_q, lds_stats, s_cnt, s_base, lds_exc, s_tb, s_tbase); ROLE_DONE(3); return; }
  b -= D.R * pl.nb_gossip;
  if (b < pl.nb_ppreply) { if (pl.roles & 64u) role_ppreply<MASS, MULTI>(D, b, pl.nb_ppreply, lds_stats); ROLE_DONE(4); return; }
  b -= pl.nb_ppreply;
  if (MULTI || TB) {
    if (b < pl.nb_carry) {
      if (pl.roles & 32u) {
        if constexpr (TB) { if (D.tb_carry) role_carry_tb<MULTI>(D, b, lds_stats, s_tb, s_tbase); else if (MULTI) role_carry(D, b, pl.nb_carry, lds_stats); }
        else role_carry(D, b, pl.nb_carry, lds_stats);
      }
      ROLE_DONE(5); return;
    }
    b -= pl.nb_carry;
  }
  if (b < pl.nb_join) { role_join<MASS, MULTI>(D, lds_stats); ROLE_DONE(7); return; }
  b -= pl.nb_join;
  if (pl.roles & 16u) role_pushpull<MASS, MULTI>(D, b / pl.nb_pp, (b % pl.nb_pp) * SW_BLOCK + threadIdx.x, lds_stats, lds_exc);
  ROLE_DONE(6);
#undef ROLE_DONE
}
typedef void (*BeginKernel)(const SwDev*, BeginPlan);
// pick the leanest instantiation the configuration allows (the dense pair store and the tile buckets exclude each other)
template <bool MASS, bool TB>
static BeginKernel select_begin_m(uint32_t fanout, bool serf, bool multi) {
  const bool k8 = fanout > 4;
  if (k8) return serf ? (multi ? k_begin<8, true, true, MASS, TB> : k_begin<8, true, false, MASS, TB>) : (multi ? k_begin<8, false, true, MASS, TB> : k_begin<8, false, false, MASS, TB>);
  return serf ? (multi ? k_begin<4, true, true, MASS, TB> : k_begin<4, true, false, MASS, TB>) : (multi ? k_begin<4, false, true, MASS, TB> : k_begin<4, false, false, MASS, TB>);
}
static BeginKernel select_begin(uint32_t fanout, bool serf, bool multi, bool mass, bool tb) {
  return mass ? select_begin_m<true, false>(fanout, serf, multi) : tb ? select_begin_m<false, true>(fanout, serf, multi) : select_begin_m<false, false>(fanout, serf, multi);
}

// =================================================================================================
// k_deliver — packetListen/ingestPacket: scatter an edge list into the per-node inbox rows.  One
// returning atomic on the row's count word reserves the slot; the record lands in the same 64-byte
// line for the first five arrivals.  Fold census records (dst = NONE) are accumulated on the spot.
// =================================================================================================
// A fold census record (dst = NONE; swim_device.h): what the acting observers of ONE shard hold about
// subject g = replica*N + node — rec.z = their common key or FOLD_POISON, rec.w = how many hold an explicit view.
// Every shard receives every shard's records and accumulates them; k_fold_apply decides.
__device__ __forceinline__ void fold_accumulate(DevRef D, uint4 rec) {
  const uint32_t g = rec.y;
  atomicAdd(&D.fg_cnt[g], rec.w); atomicMin(&D.fg_kmin[g], rec.z); atomicMax(&D.fg_kmax[g], rec.z);
}

// reserve: one returning atomic on the count word of the node's 64-byte inbox line
__device__ __forceinline__ uint32_t inbox_reserve(DevRef D, uint4 rec, size_t& l, const ExcList* X = nullptr) {
  if (rec.x == NONE) { fold_accumulate(D, rec); return NONE; }          // fold census record
  uint32_t r = div_n(D, rec.x), x = mod_n(D, rec.x);
  if (x < D.i0 || x >= D.i0 + D.nloc) return NONE;
  uint32_t w = (X && X->usable()) ? X->word(x) : D.nw[rec.x];           // (X: the list of the replica every record of this span belongs to)
  if (w & NW_DEAD) return NONE;                    // e.g. a push-pull reply to a requester that died meanwhile
  if (w & NW_ATTACHED) { if (rec.y != SWIM_SUBJECT_PIGGY) capture(D, NONE, rec.x, rec.y, rec.z, rec.w); return NONE; }
  l = (size_t)r * D.nloc + (x - D.i0);
  if (rec.y == SWIM_SUBJECT_PIGGY) {               // a piggy-back order for a node with nothing queued is a no-op
    if (!q_bit(D, l)) return NONE;                // (queues do not change between k_begin and k_resolve)
    if (D.iq) {                                   // SWIM_F_UNBOUNDED_QUEUE: not a message of the inbox — k_piggy_iq serves the node's orders with one scan of its column
      const uint32_t pos = atomicAdd(&D.ord_cnt[l], 1u);
      if (pos < D.ord_cap) D.ord[l * D.ord_cap + pos] = make_uint2(rec.z, rec.w); else atomicOr(D.err, SW_ERR_ORDER_OVF);
      if (pos == 0) D.ord_nodes[atomicAdd(D.ord_n, 1u)] = (uint32_t)l;
      return NONE;
    }
  }
  return atomicAdd(&D.in_cnt[l], 1u);
}
// the overflow row of node l this tick: its own, or — pooled rows (swim_device.h) — the big row it was given
__device__ __forceinline__ uint32_t* inbox_row(DevRef D, size_t l) {
  if (D.PB) { const uint32_t br = D.big_row[l]; if (br < D.PB) return D.inbox_big + (size_t)br * D.C2 * 3; }
  return D.inbox2 + l * D.C1 * 3;
}
// place: the message lands in the same line for the first SW_INBOX_FAST arrivals, else in the overflow row
__device__ __forceinline__ void inbox_place(DevRef D, uint4 rec, size_t l, uint32_t pos) {
  if (pos == NONE) return;
  uint32_t* m = nullptr;
  if (pos < SW_INBOX_FAST) m = D.inbox1 + l * 16 + 1 + 3 * pos;
  else if (D.PB && pos >= D.C1) {                  // pooled rows: beyond the node's own row — deferred until the node has a big row (k_inbox_claim / k_inbox_file)
    if (pos < D.C) {
      const uint32_t at = atomicAdd(D.defer_n, 1u);
      if (at < D.defer_cap) { D.defer_rec[at] = make_uint4(rec.y, rec.z, rec.w, pos); D.defer_l[at] = (uint32_t)l; } else atomicOr(D.err, SW_ERR_INBOX_OVF);
    }
  }
  else if (pos < D.C) m = D.inbox2 + (l * D.C1 + (pos - SW_INBOX_FAST)) * 3;
  if (m) { m[0] = rec.y; m[1] = rec.z; m[2] = rec.w; }
  if (pos == 0 && D.fast_blocks) D.in_any[l / 64] = 1;                   // (a hint per 64 nodes: the unit k_resolve's workgroups own one or four of)
}
// four records per thread per trip: all four atomics are in flight before the first store
// (round 4: the same written in explicit phases over the four records — unpredicated clamped loads, all node words, all atomics, all
// stores, which is what the ISA of this loop does not do (profiles/r03_isa_notes.txt item 2) — measured 45.3 us per launch against 42.9:
// the kernel is bound by the rate of its scattered requests, not by their serialisation within a lane; profiles/r04_ab_experiments.txt)
// JUDGE: the list came from other shards (mailbox, swim_inbound) — a rumour marked SW_EDGE_JUDGE is first put to the no-op filter here
template <bool JUDGE = false>
__device__ __forceinline__ void deliver_span(DevRef D, const uint4* edges, uint32_t n, uint32_t first, uint32_t stride, const ExcList* X = nullptr) {
  uint32_t c_filt = 0, c_edges = 0;
  for (uint32_t e = first; e < n; e += 4 * stride) {
    uint4 rec[4]; size_t l[4]; uint32_t pos[4]; bool on[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { on[j] = e + j * stride < n; if (on[j]) rec[j] = edges[e + j * stride]; }
    if (JUDGE) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (!on[j] || rec[j].x == NONE || !(rec[j].w & SW_EDGE_JUDGE) || rec[j].y == SWIM_SUBJECT_PIGGY || rec[j].y == SWIM_SUBJECT_PULL) continue;
        rec[j].w &= ~SW_EDGE_JUDGE;
        const uint32_t r = div_n(D, rec[j].x), x = mod_n(D, rec[j].x), type = rec[j].w >> 30;
        if (x >= D.i0 && x < D.i0 + D.nloc &&
            noop_at_receiver<true>(D, r, (size_t)r * D.nloc + (x - D.i0), D.nw[(size_t)r * D.N + rec[j].y], make_uint4(rec[j].y, rec[j].z, rec[j].w & 0x3FFFFFFFu, type << 30), false, rec[j])) { c_filt++; on[j] = false; }
        else c_edges++;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) pos[j] = on[j] ? inbox_reserve(D, rec[j], l[j], X) : NONE;
#pragma unroll
    for (int j = 0; j < 4; j++) inbox_place(D, rec[j], l[j], pos[j]);
  }
  if (JUDGE && __any((c_edges | c_filt) != 0)) {
    for (int off = 32; off; off >>= 1) { c_edges += __shfl_down(c_edges, off); c_filt += __shfl_down(c_filt, off); }
    if (sw_lane() == 0) {
      if (c_edges) atomicAdd(stat_ptr(D, ST_EDGES), (unsigned long long)c_edges);
      if (c_filt) atomicAdd(stat_ptr(D, ST_FILTERED), (unsigned long long)c_filt);
    }
  }
}
// The broadcasts piggy-backed on last tick's pings and acks (picked by k_resolve, one private area per block)
// arrive with this tick's packets.  Same no-op filter as the gossip role applies at the sender: here the
// receiver's view is read before the inbox is touched.
template <bool MASS>
__device__ void deliver_carried(DevRef D, const uint4* area, uint32_t n, uint32_t& c_edges, uint32_t& c_filt) {
  const bool filter = (D.flags & SWIM_F_FILTER_NOOP) != 0;
  for (uint32_t e = threadIdx.x; e < n; e += SW_BLOCK) {
    uint4 rec = area[e];
    if (rec.x == SW_DST_VOID) continue;              // left for another shard in k_begin
    uint32_t r = div_n(D, rec.x), x = mod_n(D, rec.x), type = rec.w >> 30;
    if (filter && type != SWIM_MSG_USER && rec.y != x) {
      uint32_t ws = D.nw[(size_t)r * D.N + rec.y];
      if (noop_at_receiver<MASS>(D, r, (size_t)r * D.nloc + (x - D.i0), ws, make_uint4(rec.y, rec.z, rec.w & 0x3FFFFFFFu, type << 30), false, rec)) { c_filt++; continue; }
    }
    c_edges++;
    size_t l; uint32_t pos = inbox_reserve(D, rec, l);
    inbox_place(D, rec, l, pos);
  }
}
// ---- tile buckets (swim_device.h): one workgroup of k_deliver per tile drains the tile's bucket ------------------------------------
// SW_TB_CHUNK records at a time: (1) the records into LDS (one coalesced read) and a histogram over the tile's 1 024 nodes, (2) its
// exclusive prefix, (3) an index sorted by receiver, (4) every record judged IN RECEIVER ORDER — the view reads of a wave fall into a few
// consecutive lines of the tile's own window instead of 64 lines anywhere in the cluster's — and what is not a no-op filed in the
// receiver's inbox (count words and message lines of the tile: this workgroup's alone in this launch, and still in this XCD's L2 when
// k_resolve's workgroup for the tile — same workgroup index, same XCD — reads them).  The question asked is the sender's
// (noop_at_receiver against the pre-tick view: nothing has been merged yet), so the same rumours are dropped, delivered and counted
// as when the gossip role asked it.  Four records per lane are in flight together (their subjects' node words, then the receivers'
// home slots): the chain of a chunk is as long as one record's.
template <bool MASS>
__device__ __forceinline__ void deliver_bucket(DevRef D, uint32_t tile, uint32_t n_tb, uint4* raw, uint16_t* ord, uint32_t* bins, uint32_t* s_wt, bool dbg) {
  uint4* const bk = D.tb + (size_t)tile * D.tb_cap;
  const size_t l0 = (size_t)tile * SW_TB_TILE, NL = (size_t)D.R * D.nloc;
  uint32_t c_filt = 0, c_edges = 0;
  for (uint32_t c0 = 0; c0 < n_tb; c0 += SW_TB_CHUNK) {
    const uint32_t nc = n_tb - c0 < SW_TB_CHUNK ? n_tb - c0 : SW_TB_CHUNK;
    for (uint32_t i = threadIdx.x; i < SW_TB_TILE; i += SW_BLOCK) bins[i] = 0;
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < SW_TB_CHUNK / SW_BLOCK; j++) {
      const uint32_t e = threadIdx.x + j * SW_BLOCK;
      if (e < nc) {
        const uint4 rec = ld_global_u4(bk + c0 + e);
        raw[e] = rec;
        const size_t l = (size_t)div_n(D, rec.x) * D.nloc + (mod_n(D, rec.x) - D.i0);
        atomicAdd(&bins[(uint32_t)(l - l0) & (SW_TB_TILE - 1u)], 1u);
      }
    }
    __syncthreads();
    {   // exclusive prefix over the 1 024 bins: four per thread, a wave scan of the sums, the four waves' totals
      const uint32_t a0 = bins[4 * threadIdx.x], a1 = bins[4 * threadIdx.x + 1], a2 = bins[4 * threadIdx.x + 2], a3 = bins[4 * threadIdx.x + 3], sum = a0 + a1 + a2 + a3;
      uint32_t incl = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(incl, off); if (sw_lane() >= (uint32_t)off) incl += v; }
      if (sw_lane() == 63) s_wt[threadIdx.x / 64] = incl;
      __syncthreads();
      uint32_t base = 0;
      for (uint32_t w = 0; w < threadIdx.x / 64; w++) base += s_wt[w];
      const uint32_t ex = base + incl - sum;
      bins[4 * threadIdx.x] = ex; bins[4 * threadIdx.x + 1] = ex + a0; bins[4 * threadIdx.x + 2] = ex + a0 + a1; bins[4 * threadIdx.x + 3] = ex + a0 + a1 + a2;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < SW_TB_CHUNK / SW_BLOCK; j++) {
      const uint32_t e = threadIdx.x + j * SW_BLOCK;
      if (e < nc) {
        const uint32_t gx = raw[e].x;
        const size_t l = (size_t)div_n(D, gx) * D.nloc + (mod_n(D, gx) - D.i0);
        ord[atomicAdd(&bins[(uint32_t)(l - l0) & (SW_TB_TILE - 1u)], 1u)] = (uint16_t)e;
      }
    }
    __syncthreads();
    {
      constexpr uint32_t K = SW_TB_CHUNK / SW_BLOCK;
      uint4 rec[K], first[K]; uint32_t e[K], cls[K], ws[K], r[K]; size_t lr[K]; bool on[K], ask[K];
#pragma unroll
      for (uint32_t j = 0; j < K; j++) {
        const uint32_t i = threadIdx.x + j * SW_BLOCK;
        on[j] = i < nc; e[j] = on[j] ? ord[i] : 0u; rec[j] = raw[e[j]];
        cls[j] = rec[j].w & TB_CLASS_MASK; rec[j].w &= ~TB_CLASS_MASK;
        ask[j] = on[j] && (cls[j] == TB_GOSSIP || cls[j] == TB_CARRIED);           // the receiver's view decides
        r[j] = div_n(D, rec[j].x); lr[j] = (size_t)r[j] * D.nloc + (mod_n(D, rec[j].x) - D.i0);
      }
#pragma unroll
      for (uint32_t j = 0; j < K; j++) ws[j] = ask[j] ? D.nw[(size_t)r[j] * D.N + rec[j].y] : 0u;      // the subjects' node words (mostly one word for the whole wave)
#pragma unroll
      for (uint32_t j = 0; j < K; j++)                                                                 // the receivers' home slots for the subjects
        first[j] = (ask[j] && !(MASS && (ws[j] & NW_MASS)) && (ws[j] & NW_SUBJECT)) ? D.vt[(size_t)vt_home(D, rec[j].y) * NL + lr[j]] : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (uint32_t j = 0; j < K; j++) {
        if (!on[j]) continue;
        if (ask[j]) {
          const uint32_t type = rec[j].w >> 30;
          if (noop_at_receiver<MASS>(D, r[j], lr[j], ws[j], make_uint4(rec[j].y, rec[j].z, rec[j].w & 0x3FFFFFFFu, type << 30), !(MASS && (ws[j] & NW_MASS)), first[j])) {
            c_filt++;
            if (dbg && cls[j] == TB_GOSSIP) ((uint32_t*)(bk + c0 + e[j]))[0] = SW_DST_VOID;      // swim_debug_edges reports the gossip role's rumours that were delivered
            continue;
          }
          c_edges++;
        } else if (cls[j] == TB_CARRIED_PLAIN) c_edges++;
        size_t l; const uint32_t pos = inbox_reserve(D, rec[j], l);
        inbox_place(D, rec[j], l, pos);
      }
    }
    __syncthreads();                                 // (the next chunk reuses raw / ord / bins)
  }
  if (__any((c_edges | c_filt) != 0)) {
    for (int off = 32; off; off >>= 1) { c_edges += __shfl_down(c_edges, off); c_filt += __shfl_down(c_filt, off); }
    if (sw_lane() == 0) {
      if (c_edges) atomicAdd(stat_ptr(D, ST_EDGES), (unsigned long long)c_edges);
      if (c_filt) atomicAdd(stat_ptr(D, ST_FILTERED), (unsigned long long)c_filt);
    }
  }
}

// grid = [tile buckets: one block per tile] + n_seg blocks + extra blocks over the shard's misc list.  Block b drains segment b and the carry areas
// b, b + n_seg, ... (their counts are fetched together with the segment's: no extra trip in a quiet tick)
template <bool MASS, bool TB>
__global__ void __launch_bounds__(SW_BLOCK) k_deliver(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  uint32_t b = blockIdx.x;
  if constexpr (TB) {
    if (b < D.tb_T) {                              // the bucket of the tile k_resolve's workgroup of the same index will merge (same XCD)
      __shared__ uint4 s_raw[SW_TB_CHUNK];
      __shared__ uint16_t s_ord[SW_TB_CHUNK];
      __shared__ uint32_t s_bins[SW_TB_TILE], s_wt[SW_BLOCK / 64];
      // (rs_order lists k_resolve's tiles — node blocks since round 5; it is the buckets' order too only when both tile alike)
      const uint32_t tile = (SW_RTILE * SW_BLOCK == SW_TB_TILE && D.rs_order) ? D.rs_order[(size_t)(*D.tick % D.P) * D.rs_T + b] : b;
      uint32_t n_tb = D.tb_cnt[tile];
      const bool dbg = *D.dbg_on != 0;
      if (n_tb > D.tb_cap) n_tb = D.tb_cap;        // (the producer that overflowed it has raised SW_ERR_EDGE_OVF)
      if (n_tb) deliver_bucket<MASS>(D, tile, n_tb, s_raw, s_ord, s_bins, s_wt, dbg);
      if (threadIdx.x == 0) { if (n_tb) D.tb_cnt[tile] = 0; if (dbg) D.tb_last[tile] = n_tb; }
      return;
    }
    b -= D.tb_T;
    if (D.tb_carry) b += D.n_seg;                  // (no segments, and the carry areas went through the buckets: only the misc list is left)
  }
#ifdef SWIMSIM_WAVECLK
  const unsigned long long t_in = wall_clock64();
#define DCLK(tag) BCLK_OUT(1, t_in, tag)
#else
#define DCLK(tag) do { } while (0)
#endif
  if (b < D.n_seg) {
    uint32_t n = D.seg_cnt[b], last = D.seg_last[b];
    // a segment holds records of ONE replica: its exception list (the few nodes whose word is not 0) in LDS saves the
    // node-word read of every record
    __shared__ uint32_t lds_exc[2 * SW_EXC_MAX];
    ExcList X; X.n = SW_EXC_MAX + 1; X.id = lds_exc; X.w = lds_exc + SW_EXC_MAX;
    if (n) X.stage(D, b < D.R * D.nb_gossip ? b / D.nb_gossip : (b - D.R * D.nb_gossip) / D.nb_probe, lds_exc);
    uint32_t cn[4] = { 0, 0, 0, 0 }, cl[4] = { 0, 0, 0, 0 };
    // anything carried into this tick?  (uniform words: k_resolve stamps carry_stamp with the tick its picks
    // travel in, so a tick without piggy-backed broadcasts costs this block nothing more)
    uint32_t par = 0;
    const bool piggy = D.nb_carry != 0 && !D.tb_carry && *D.carry_stamp == *D.tick;     // (tile buckets: k_begin's carry role has filed them already)
    if (piggy) {                                     // {count, count of the previous tick} in one 8-byte load per area
      par = *D.tick & 1u;
#pragma unroll
      for (int j = 0; j < 4; j++) { uint32_t a = b + j * D.n_seg; if (a < D.NB) { uint2 c = D.carry_cl[a]; cn[j] = c.x; cl[j] = c.y; } }
    }
    __syncthreads();                               // everybody has read the counts before lane 0 clears them
    if (threadIdx.x == 0) {
      if (last != n) D.seg_last[b] = n;
      if (piggy && b == 0) D.carry_stamp[1] = *D.tick;            // swim_debug_edges: the areas' `last` words are of this tick
#pragma unroll
      for (int j = 0; j < 4; j++) {
        uint32_t a = b + j * D.n_seg;
        if (cn[j] > D.carry_cap) atomicOr(D.err, SW_ERR_CARRY_OVF);
        if (cl[j] | cn[j]) D.carry_cl[a] = make_uint2(0, cn[j]);
      }
    }
    if (n) deliver_span(D, D.seg + (size_t)b * D.seg_cap, n, threadIdx.x, SW_BLOCK, &X);
    if (threadIdx.x == 0 && n) D.seg_cnt[b] = 0;
    if (!piggy) { DCLK(n); return; }
    uint32_t c_edges = 0, c_filt = 0;
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (cn[j]) deliver_carried<MASS>(D, D.carry + ((size_t)par * D.NB + b + j * D.n_seg) * D.carry_cap, cn[j] < D.carry_cap ? cn[j] : D.carry_cap, c_edges, c_filt);
    for (uint32_t a = b + 4 * D.n_seg; a < D.NB; a += D.n_seg) {      // only with very fine quanta (G > 4)
      uint2 cc = D.carry_cl[a]; uint32_t c = cc.x;
      __syncthreads();
      if (threadIdx.x == 0 && (cc.x | cc.y)) D.carry_cl[a] = make_uint2(0, c);
      if (c) deliver_carried<MASS>(D, D.carry + ((size_t)par * D.NB + a) * D.carry_cap, c < D.carry_cap ? c : D.carry_cap, c_edges, c_filt);
    }
    if (__any((c_edges | c_filt) != 0)) {
      for (int off = 32; off; off >>= 1) { c_edges += __shfl_down(c_edges, off); c_filt += __shfl_down(c_filt, off); }
      if (sw_lane() == 0) {
        if (c_edges) atomicAdd(stat_ptr(D, ST_EDGES), (unsigned long long)c_edges);
        if (c_filt) atomicAdd(stat_ptr(D, ST_FILTERED), (unsigned long long)c_filt);
      }
    }
    DCLK(n + cn[0] + cn[1] + cn[2] + cn[3]);
    return;
  }
  b -= D.n_seg;
  uint32_t nb = gridDim.x - (TB ? D.tb_T + (D.tb_carry ? 0u : D.n_seg) : D.n_seg), n = D.out_cnt[D.rank];
  if (n > D.out_cap_tab[D.rank]) n = D.out_cap_tab[D.rank];
  deliver_span(D, D.out_tab[D.rank], n, b * SW_BLOCK + threadIdx.x, nb * SW_BLOCK);
  DCLK(1u << 30);
#undef DCLK
}
// ---- swim_xchg_*: device-driven exchange through peer-mapped mailboxes (swim_device.h) ----------------------------
__device__ __forceinline__ uint32_t* mb_hdr(DevRef D, uint8_t* base, uint32_t parity, uint32_t src) {
  return (uint32_t*)(base + ((size_t)parity * D.n_shards + src) * 64);
}
__device__ __forceinline__ uint4* mb_rec(DevRef D, uint8_t* base, uint32_t parity, uint32_t src) {
  return (uint4*)(base + (size_t)2 * D.n_shards * 64) + ((size_t)parity * D.n_shards + src) * D.mail_cap;
}
// my segment for shard blockIdx.y goes into my area of ITS mailbox (stores over xGMI / through the shared L2)
__global__ void __launch_bounds__(SW_BLOCK) k_xchg_copy(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  const uint32_t dst = blockIdx.y;
  if (dst == D.rank) return;
  uint32_t n = D.out_cnt[dst]; n = n < D.out_cap_tab[dst] ? n : D.out_cap_tab[dst]; n = n < D.mail_cap ? n : D.mail_cap;
  const uint4* src = D.out_tab[dst];
  uint4* to = mb_rec(D, D.mb_tab[dst], *D.tick & 1u, D.rank);
  for (uint32_t e = blockIdx.x * SW_BLOCK + threadIdx.x; e < n; e += gridDim.x * SW_BLOCK) to[e] = src[e];
}
// ...then count + activity word, and the flag (release, system scope): one thread per destination
__global__ void k_xchg_signal(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  const uint32_t dst = threadIdx.x, t = *D.tick;
  if (dst >= D.n_shards || dst == D.rank) return;
  uint32_t any = *D.act;
  for (uint32_t sh = 0; sh < D.n_shards; sh++) any |= D.out_cnt[sh];
  uint32_t n = D.out_cnt[dst];
  if (n > D.out_cap_tab[dst] || n > D.mail_cap) { atomicOr(D.err, SW_ERR_EDGE_OVF); n = n < D.mail_cap ? (n < D.out_cap_tab[dst] ? n : D.out_cap_tab[dst]) : D.mail_cap; }
  uint32_t* h = mb_hdr(D, D.mb_tab[dst], t & 1u, D.rank);
  h[1] = n; h[2] = any != 0;
  __threadfence_system();
  __hip_atomic_store(&h[0], t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// wait for the flags of my sources (acquire); their counts; the population's activity word for the next tick
__global__ void k_xchg_wait(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  __shared__ uint32_t s_any;
  const uint32_t src = threadIdx.x, t = *D.tick;
  if (threadIdx.x == 0) { uint32_t a = *D.act; for (uint32_t sh = 0; sh < D.n_shards; sh++) a |= D.out_cnt[sh]; s_any = a != 0; }
  __syncthreads();
  if (src < D.n_shards && src != D.rank) {
    uint32_t* h = mb_hdr(D, D.mb_tab[D.rank], t & 1u, src);
    const unsigned long long t0 = wall_clock64(), limit = (unsigned long long)D.xchg_timeout_ms * 100000ull;   // 100 MHz
    uint32_t n = 0; bool ok = false;
    if (!(*D.err & SW_ERR_XCHG_TIMEOUT))            // after one time-out the run is void anyway: do not wait again
      for (;;) {
        if (__hip_atomic_load(&h[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == t + 1) { ok = true; break; }
        if (wall_clock64() - t0 > limit) break;
        __builtin_amdgcn_s_sleep(8);
      }
    if (ok) { n = h[1]; if (h[2]) atomicOr(&s_any, 1u); } else atomicOr(D.err, SW_ERR_XCHG_TIMEOUT);
    D.xin_cnt[src] = n < D.mail_cap ? n : D.mail_cap;
  }
  __syncthreads();
  if (threadIdx.x == 0) { D.xin_cnt[D.rank] = 0; *D.peer_act = s_any; }
}
__global__ void __launch_bounds__(SW_BLOCK) k_deliver_mail(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  const uint32_t src = blockIdx.y;
  if (src == D.rank) return;
  deliver_span<true>(D, mb_rec(D, D.mb_tab[D.rank], *D.tick & 1u, src), D.xin_cnt[src], blockIdx.x * SW_BLOCK + threadIdx.x, gridDim.x * SW_BLOCK);
}
// ---- swim_frame_*: the exchange as ONE equal-split collective, sizes known to the host, counts known to the device only (swimsim.h) ----
// frame for shard blockIdx.y at send + blockIdx.y * F: header {count, activity, tick + 1, magic}, then the segment's records
// (fill: swim_frame_pack_fill — a segment that does not fit the frame is not an error: the header carries the segment's true count and, above
//  the activity bit, the largest count this shard holds for any destination; the caller repeats the tick's exchange with larger frames)
__global__ void __launch_bounds__(SW_BLOCK) k_frame_pack(const SwDev* __restrict__ Dp, uint4* send, uint32_t F, uint32_t fill) {
  SW_DEV_BIND
  const uint32_t dst = blockIdx.y;
  uint4* const fr = send + (size_t)dst * F;
  uint32_t n = dst == D.rank ? 0u : D.out_cnt[dst];
  if (n > D.out_cap_tab[dst]) n = D.out_cap_tab[dst];          // (the list itself overflowed: the roles raised SW_ERR_EDGE_OVF when they appended)
  const uint32_t room = D.out_cap_tab[dst] < F - 1 ? D.out_cap_tab[dst] : F - 1;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    uint32_t any = *D.act, need = 0;
    for (uint32_t sh = 0; sh < D.n_shards; sh++) { const uint32_t c = D.out_cnt[sh] < D.out_cap_tab[sh] ? D.out_cnt[sh] : D.out_cap_tab[sh]; any |= c; if (sh != D.rank && c > need) need = c; }
    if (n > room && !fill) atomicOr(D.err, SW_ERR_EDGE_OVF);
    fr[0] = fill ? make_uint4(n, (any != 0 ? 1u : 0u) | (need << 1), *D.tick + 1, SWIM_FRAME_MAGIC)
                 : make_uint4(n < room ? n : room, any != 0, *D.tick + 1, SWIM_FRAME_MAGIC);
  }
  if (n > room) n = room;
  const uint4* src = D.out_tab[dst];
  for (uint32_t e = blockIdx.x * SW_BLOCK + threadIdx.x; e < n; e += gridDim.x * SW_BLOCK) fr[1 + e] = src[e];
}
// frame from shard blockIdx.y at recv + blockIdx.y * F: its records into the inboxes (judged here where the sender could not: SW_EDGE_JUDGE);
// workgroup (0, 0) folds the activity words into next tick's hint
__global__ void __launch_bounds__(SW_BLOCK) k_frame_deliver(const SwDev* __restrict__ Dp, const uint4* recv, uint32_t F) {
  SW_DEV_BIND
  const uint32_t src = blockIdx.y;
  if (blockIdx.x == 0 && src == 0 && threadIdx.x == 0) {
    uint32_t any = *D.act;
    for (uint32_t sh = 0; sh < D.n_shards; sh++) any |= D.out_cnt[sh];
    for (uint32_t sh = 0; sh < D.n_shards; sh++) {
      if (sh == D.rank) continue;
      const uint4 h = recv[(size_t)sh * F];
      if (h.w != SWIM_FRAME_MAGIC || h.z != *D.tick + 1) { atomicOr(D.err, SW_ERR_XCHG_TIMEOUT); any = 1; }
      else any |= h.y & 1u;                      // (above bit 0: what the sender's largest segment needs — swim_frame_pack_fill)
    }
    *D.peer_act = any != 0;
  }
  if (src == D.rank) return;
  const uint4 h = recv[(size_t)src * F];
  if (h.w != SWIM_FRAME_MAGIC || h.z != *D.tick + 1) return;
  if (h.x > F - 1) {                             // a truncated frame must be re-sent, not delivered: loud, and nothing of it is filed (like the checker: ESTATE, no partial state)
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(D.err, SW_ERR_EDGE_OVF);
    return;
  }
  deliver_span<true>(D, recv + (size_t)src * F + 1, h.x, blockIdx.x * SW_BLOCK + threadIdx.x, gridDim.x * SW_BLOCK);
}
// records handed over by other shards (swim_inbound)
__global__ void __launch_bounds__(SW_BLOCK) k_deliver_list(const SwDev* __restrict__ Dp, const uint4* edges, uint32_t n) {
  SW_DEV_BIND
  deliver_span<true>(D, edges, n, blockIdx.x * SW_BLOCK + threadIdx.x, gridDim.x * SW_BLOCK);
}

// swim_watch / inject_*: give subject x a watch slot (census, first-* stamps, trace).  Single-threaded (host-side
// stimulus, between ticks); slot_maxinc is seeded by k_watch_seed right after.
__device__ void alloc_slot(DevRef D, uint32_t r, uint32_t x, uint32_t* fresh) {
  size_t g = (size_t)r * D.N + x;
  uint32_t w = D.nw[g];
  if (NW_HAS_SLOT(w)) return;
  uint32_t sl = D.n_slots[r];
  if (sl >= D.S) { atomicAdd(stat_ptr(D, ST_SUBJ_OVF), 1ull); return; }
  D.n_slots[r] = sl + 1;
  size_t sidx = (size_t)r * D.S + sl;
  D.subj_node[sidx] = x; D.slot_dirty[sidx] = 1; D.slot_maxinc[sidx] = SW_KINC(base_key_of(D, r, x, w));
  atomicOr(&D.nw[g], sl + 1);
  if (fresh) { uint32_t pos = atomicAdd(&fresh[0], 1u); if (pos < 1023) fresh[1 + pos] = (uint32_t)sidx; }
}
// the highest incarnation any local observer holds of a freshly watched subject (n_current is counted against it)
__global__ void __launch_bounds__(SW_BLOCK) k_watch_seed(const SwDev* __restrict__ Dp, const uint32_t* fresh) {
  SW_DEV_BIND
  if (blockIdx.y >= fresh[0] || blockIdx.y >= 1023) return;
  const uint32_t sidx = fresh[1 + blockIdx.y], r = sidx / D.S, x = D.subj_node[sidx];
  const uint32_t wx = D.nw[(size_t)r * D.N + x];
  if (!(wx & (NW_SUBJECT | NW_MASS))) return;
  uint32_t m = 0;
  const uint32_t row = (wx & NW_MASS) ? D.mrow[(size_t)r * D.N + x] : 0;
  for (uint32_t k = blockIdx.x * SW_BLOCK + threadIdx.x; k < D.nloc; k += gridDim.x * SW_BLOCK) {
    uint4 e; uint32_t fs;
    if (wx & NW_MASS) { const uint32_t a = D.mA[m_idx(D, r, row, k)]; m = MA_INC(a) > m ? MA_INC(a) : m; }
    else if (vt_find(D, (size_t)r * D.nloc + k, x, e, fs) != NONE) m = SW_KINC(e.y) > m ? SW_KINC(e.y) : m;
  }
  for (int off = 32; off; off >>= 1) { uint32_t v = __shfl_down(m, off); m = v > m ? v : m; }
  if (sw_lane() == 0 && m) atomicMax(&D.slot_maxinc[sidx], m);
}
// =================================================================================================
// k_resolve — handleAlive/handleSuspect/handleDead/handleUserEvent for everything that reached a
// node this tick, applied in ascending (user?, subject, type, incarnation, from) order with
// duplicates applied once.  Literal aliveNode/suspectNode/deadNode/refute (state.go) and
// suspicion.Confirm (suspicion.go) against the observer's own view column.
// =================================================================================================
// The word of node x of replica r just changed from `old` to `now` inside a tick kernel: keep the replica's exception
// list (the {id, word} pairs of the non-zero words, staged in LDS by k_begin's roles) equal to nw.
__device__ void exc_note(DevRef D, uint32_t r, uint32_t x, uint32_t old, uint32_t now) {
  uint2* ent = D.exc_ent + (size_t)r * SW_EXC_MAX;
  if (old == 0) { uint32_t pos = atomicAdd(&D.exc_cnt[r], 1u); if (pos < SW_EXC_MAX) ent[pos] = make_uint2(x, now); return; }
  uint32_t n = D.exc_cnt[r]; if (n > SW_EXC_MAX) return;               // unusable anyway
  for (uint32_t j = 0; j < n; j++) if (ent[j].x == x) atomicOr(&ent[j].y, now & ~old);
}
extern __shared__ uint4 g_lds_dyn[];      // the kernel's dynamic LDS (named at file scope so that NodeCtxT's accesses stay LDS-typed, not generic)
// k_resolve keeps the censuses of watched subjects up to date INCREMENTALLY: a view that changes state (or reaches the slot's
// highest incarnation) adds its delta here — per workgroup in LDS, flushed with one global atomic per touched counter — and k_finish
// adds the deltas to the cached census.  A recount of all observers (k_census) is left for what is not a view change of a running
// observer: stimulus, folds, evictions, a new highest incarnation.  (Re-counting 65 536 observers per cluster in every tick of the
// dissemination phase was 8 % of the driver window's kernel time.)
#define SW_CEN_LDS 8                      /* watch slots per replica tallied in LDS; higher slot numbers go to global memory directly */
__shared__ int g_s_cen[SW_CEN_LDS * 5];   // [slot][state 0..3, current]
__shared__ uint32_t g_s_cen_r;            // the replica the workgroup's first node block belongs to
// LQ = the memberlist queue of the lane is staged in LDS (k_resolve: every queue access of the merge is then an LDS access
// and the entries that changed are written back once); otherwise it is edited in HBM (the stimulus kernels).
// (Every method is __forceinline__: left to the inliner's threshold, one more statement in a method made it a call, the
// context was passed by pointer and lived in scratch memory — 646 scratch instructions in k_resolve.)
// SERF = the handle has serf's event layer (SWIM_F_SERF_EVENTS): without it the user-event / intent handlers are not compiled into the
// kernel at all (k_resolve of the headline workload: their register pressure showed in its merge loop — 78 -> 92 us per launch when serf's
// intent ordering came in as run-time code)
// DYN = the handle's membership changes (n_initial < n_nodes): without it estNumNodes() is N and the scaling laws are the host's tables
// (no per-observer counts, no f64 suspicion formula in the kernel) — a compile-time switch for the same reason as SERF
template <bool LQ, bool MASS, bool SERF = true, bool DYN = true>
struct NodeCtxT {
  DevRef D; BlockStats& S;
  uint32_t r, o, k, t; size_t l, NL;
  uint32_t self_inc, leaving, qlen, evqlen, qseq, ev_clock;
  uint32_t c_pig = 0, c_sent01 = 0, c_sent23 = 0;   // piggy-back tallies (orders are frequent: no LDS atomic each); two 16-bit halves
  uint32_t dl_new = NONE;                             // earliest deadline this lane armed (the caller lowers dl_blk with it)
  uint32_t mcnt_add = 0;                              // pairs of the dense store this lane created (D.mcnt[l] is bumped once, in store())
  uint32_t iq0 = 0, iq_add = 0;                       // SWIM_F_UNBOUNDED_QUEUE: rumours implied by the pair store this node had queued at load(); pairs that became queued since
  uint4 vm; bool vm_have = false, vm_dirty = false;   // vmeta[l] {views, suspects, earliest deadline, earliest evictable}: fetched on first use
  __device__ __forceinline__ void need_vm() { if (!vm_have) { vm = VMETA(l); vm_have = true; } }
  __device__ __forceinline__ bool dyn() const { if constexpr (DYN) return D.dyn != 0; else return false; }
  __device__ __forceinline__ bool iq() const { if constexpr (MASS) return D.iq != 0; else return false; }
  uint4 h0;
  uint32_t qdirty = 0;                                // LQ: entry j of the lane's queue sits at g_lds_dyn[j * SW_RES_THREADS + threadIdx.x]; entries to write back
#define SQ(j) g_lds_dyn[(j) * SW_RES_THREADS + threadIdx.x]   /* (LQ contexts live in k_resolve only) */
  // Round 5: a handle with the dense pair store runs queue_cap 16-32 (mass events), and 16 bytes x queue_cap x 64 lanes of staged queue left
  // room for ONE wave per SIMD.  What the merge SCANS of an entry is its subject (QueueBroadcast's invalidation) and its meta word
  // (GetBroadcasts' order, Prune's victim): only those 8 bytes are staged; incarnation and accuser stay in HBM, written when an entry is
  // pushed and read when it is sent.  (Editing the whole queue in HBM — -DSW_MASS_HBMQ, round 4's idea — measured 30 % slower: every scan
  // became a row of loads; profiles/r05_ab_experiments.txt item 1.)
  static constexpr bool SPLIT = LQ && MASS && SW_SPLITQ;
#define SQ2(j) ((uint2*)g_lds_dyn)[(j) * SW_RES_THREADS + threadIdx.x]
  __device__ __forceinline__ NodeCtxT(DevRef d, BlockStats& s) : D(d), S(s) {}

  __device__ __forceinline__ void load() { load(HDR(l)); }
  __device__ __forceinline__ void load(uint4 h) {
    h0 = h;
    self_inc = h0.x; leaving = h_leaving(h0.y); qlen = h_qlen(h0.y); evqlen = h_evqlen(h0.y); qseq = h0.z; ev_clock = h0.w;
    if (iq()) iq0 = D.iqn[l];
  }
  // LQ: fetch the queue into the LDS column (independent loads, issued together)
  __device__ __forceinline__ void stage_queue(uint32_t from = 0) {
    for (uint32_t j = from; j < qlen; j++) { if constexpr (SPLIT) { const uint4 e = QENT(j, l); SQ2(j) = make_uint2(e.x, e.w); } else SQ(j) = QENT(j, l); }
  }
  __device__ __forceinline__ uint4 mq_get(uint32_t j) const {
    if constexpr (SPLIT) { uint4 e = QENT(j, l); const uint2 s2 = SQ2(j); e.x = s2.x; e.w = s2.y; return e; }
    else if constexpr (LQ) return SQ(j); else return QENT(j, l);
  }
  __device__ __forceinline__ uint32_t mq_x(uint32_t j) const { if constexpr (SPLIT) return SQ2(j).x; else if constexpr (LQ) return SQ(j).x; else return QENT(j, l).x; }
  __device__ __forceinline__ uint32_t mq_w(uint32_t j) const { if constexpr (SPLIT) return SQ2(j).y; else if constexpr (LQ) return SQ(j).w; else return QENT(j, l).w; }
  __device__ __forceinline__ void mq_set(uint32_t j, uint4 e) {
    if constexpr (SPLIT) { SQ2(j) = make_uint2(e.x, e.w); QENT(j, l) = e; qdirty &= ~(1u << j); }     // (HBM holds the whole entry from here on: not dirty)
    else if constexpr (LQ) { SQ(j) = e; qdirty |= 1u << j; } else QENT(j, l) = e;
  }
  // most deliveries in a saturated cluster are old news: only write the header back when it changed
  // did the node go from "nothing queued" to "something queued" (or back) since load()?
  __device__ __forceinline__ bool q_became_set() const { return !(h_qlen(h0.y) | h_evqlen(h0.y) | iq0) && (qlen | evqlen | iq0 | iq_add); }
  __device__ __forceinline__ bool q_became_clr() const { return (h_qlen(h0.y) | h_evqlen(h0.y) | iq0) && !(qlen | evqlen | iq0 | iq_add); }
  __device__ __forceinline__ void store() {
    flush_view();
    if constexpr (LQ) for (uint32_t m = qdirty & (qlen >= 32 ? 0xFFFFFFFFu : (1u << qlen) - 1); m; m &= m - 1) {
      const uint32_t j = __ffs(m) - 1;
      if constexpr (SPLIT) QENT(j, l).w = SQ2(j).y; else QENT(j, l) = SQ(j);          // (SPLIT: only a transmit count can be newer than HBM's copy)
    }
    uint4 h = make_uint4(self_inc, h_pack(leaving, qlen, evqlen), qseq, ev_clock);
    if (h.x != h0.x || h.y != h0.y || h.z != h0.z || h.w != h0.w) HDR(l) = h;
    if (vm_dirty) VMETA(l) = vm;
    if (MASS && mcnt_add) { D.mcnt[l] += mcnt_add; mcnt_add = 0; }
    if (MASS && iq_add) D.iqn[l] = iq0 + iq_add;
  }

  // QueueBroadcast: same-subject invalidation, Prune() on overflow.  EV = the serf user-event queue (always in HBM)
  template <bool EV>
  __device__ __forceinline__ void queue_push(uint32_t cap, uint32_t& len, uint32_t seq, bool named,
                             uint32_t subject, uint32_t type, uint32_t inc, uint32_t from, int drop_stat) {
    uint4* const eb = D.evq + l;
    uint32_t n = len;
    const uint4 e = make_uint4(subject, inc, from, m_pack(type, 0, seq));
    // ONE pass over the queue (round 5; two until then: the invalidation's, then Prune()'s): the entry about the same subject, and — only looked
    // at when there is none and the queue is full — the entry that sorts last among the queue and the newcomer.  (A hit makes room, so the
    // two never meet; no early exit, so that the loads overlap.)  Same comparisons in the same order as the two loops made.
    const bool full = n == cap;
    uint32_t hit = NONE, w = NONE, wmeta = e.w;
    for (uint32_t j = 0; j < n; j++) {
      uint32_t xj, mj;
      if constexpr (EV) { const uint4 q = eb[(size_t)j * NL]; xj = q.x; mj = q.w; }
      else if constexpr (SPLIT) { const uint2 q = SQ2(j); xj = q.x; mj = q.y; }
      else { xj = mq_x(j); mj = mq_w(j); }
      if (named && xj == subject) hit = j;
      if (full && ent_before(D, wmeta, mj)) { wmeta = mj; w = j; }
    }
    if (hit != NONE) {                             // at most one entry per subject: the older one goes, the new one joins at the end
      if (hit != n - 1) { if (EV) eb[(size_t)hit * NL] = eb[(size_t)(n - 1) * NL]; else mq_set(hit, mq_get(n - 1)); }
      if (EV) eb[(size_t)(n - 1) * NL] = e; else mq_set(n - 1, e);
    } else if (full) {                             // Prune(): the last in order leaves — the newcomer itself if that is where it sorts
      S.add(drop_stat);
      if (w != NONE) { if (EV) eb[(size_t)w * NL] = e; else mq_set(w, e); }
    } else { if (EV) eb[(size_t)n * NL] = e; else mq_set(n, e); n++; }
    len = n;
    if (D.fast_blocks) D.q_any[l / SW_BLOCK] = 1;
  }
  __device__ __forceinline__ void broadcast(uint32_t subject, uint32_t type, uint32_t inc, uint32_t from) {
    queue_push<false>(D.Q, qlen, qseq, true, subject, type, inc, from, ST_QDROPS); qseq++;
  }
  __device__ __forceinline__ void record_event(uint32_t type, uint32_t node, uint32_t ltime, uint32_t inc) {
    uint32_t pos = atomicAdd(D.ev_cnt, 1u);
    if (pos < D.ev_cap) { swim_event ev = { now_ms(D, t), r, type, node, inc, o, (uint64_t)ltime }; D.events[pos] = ev; }
    else atomicOr(D.err, SW_ERR_EVENT_OVF);
  }
  // does this observer have an EventCh: cfg.watch_node, or one added with swim_watch_events (rare: the list is only looked at
  // when the call was ever made)
  __device__ __forceinline__ bool watching() const {
    if (o == D.watch) return true;
    if (!D.ev_any) return false;
    const uint32_t* w = D.ev_watch + (size_t)r * SWIM_EVENT_WATCHERS; const uint32_t n = D.ev_watch[(size_t)D.R * SWIM_EVENT_WATCHERS + r];
    bool hit = false;
    for (uint32_t j = 0; j < n; j++) hit |= w[j] == o;
    return hit;
  }
  // ---- the observer's explicit view of a subject: looked up once per message (the home slot's entry and the
  // subject's node word are independent loads), edited in registers, written back once
  struct View { uint32_t slot, free_slot, w; uint4 e; bool fresh; uint4 c; bool c_have; uint32_t qe, qf; bool q_dirty; };   // c = the slot's confirmer record (vc), once fetched; qe / qf = the pair's queue words (SWIM_F_UNBOUNDED_QUEUE)
  // The inbox is applied in subject order, so consecutive messages mostly concern the same subject: its view is looked up
  // once, edited in registers across those messages and written back when the subject changes (or at store()).
  View cv; uint32_t cv_x = NONE; bool cv_dirty = false;
  __device__ __forceinline__ static bool v_mass(const View& v) { return MASS && (v.free_slot & SW_MASS_SLOT) && v.free_slot != NONE; }
  __device__ __forceinline__ void put(const View& v) {
    if (v_mass(v)) {
      m_store(D, m_idx(D, r, v.free_slot & ~SW_MASS_SLOT, k), v.e, v.c.x);
      if (v.q_dirty) { D.mE[e_idx(D, r, v.free_slot & ~SW_MASS_SLOT, k)] = v.qe; D.mF[m_idx(D, r, v.free_slot & ~SW_MASS_SLOT, k)] = v.qf; }
    }
    else D.vt[(size_t)v.slot * NL + l] = v.e;
  }
  __device__ __forceinline__ void flush_view() { if (cv_dirty) { put(cv); cv_dirty = false; } }
  // (handlers work on a by-value copy and hand it back: a reference into the context would pin the cache in scratch memory)
  __device__ __forceinline__ View take_view(uint32_t x) {
    if (cv_x == x) return cv;
    flush_view(); cv_x = x;
    return lookup(x);
  }
  __device__ __forceinline__ void put_later(View& v) { (void)v; cv_dirty = true; }     // the handler's wrapper copies v back into cv
  // encodeAndBroadcast about the subject of view v.  SWIM_F_UNBOUNDED_QUEUE: a rumour about a subject that owns a row of the dense store (and is
  // not this node itself) is stored IN THE PAIR — the invalidation of an older rumour about the same node is implied (one pair, one rumour),
  // nothing is scanned, nothing can be pruned.  `delta` = the message's incarnation minus the view's once the handler is through (0 but for a
  // confirmation that names a higher incarnation than the suspicion's).
  __device__ __forceinline__ void broadcast_v(View& v, uint32_t subject, uint32_t type, uint32_t inc, uint32_t from, uint32_t delta) {
    if (iq() && v_mass(v) && subject != o) {
      if (!(v.qe & QE_QUEUED)) iq_add++;
      if (from >= (1u << 22) || delta >= (1u << 10)) atomicOr(D.err, SW_ERR_MASS_RANGE);
      v.qe = QE_PACK(0u, type, qseq); v.qf = (from & 0x3FFFFFu) | (delta << 22); v.q_dirty = true; qseq++;
      if (qlen) {                                  // a rumour about it from before it owned a row may still sit in the slots: invalidated all the same
        uint32_t hit = NONE;
        for (uint32_t j = 0; j < qlen; j++) if (mq_x(j) == subject) hit = j;
        if (hit != NONE) { if (hit != qlen - 1) mq_set(hit, mq_get(qlen - 1)); qlen--; }
      }
      if (D.fast_blocks) D.q_any[l / SW_BLOCK] = 1;
      return;
    }
    (void)inc; broadcast(subject, type, inc, from);
  }
  __device__ __forceinline__ View lookup(uint32_t x) {
    View v; v.fresh = false; v.c_have = false; v.free_slot = 0; v.qe = 0; v.qf = 0; v.q_dirty = false;
    uint32_t row = NONE;
    if (MASS && D.M) row = D.mrow[(size_t)r * D.N + x];               // (fetched next to the node word: one round trip, not two)
    v.w = D.nw[(size_t)r * D.N + x];
    if (MASS && (v.w & NW_MASS)) {                                   // the subject owns a row of the dense store: pair (row, this lane)
      const size_t idx = m_idx(D, r, row, k);
      const uint32_t a = D.mA[idx], b = D.mB[idx], c = D.mC[idx];
      if (iq()) v.qe = D.mE[e_idx(D, r, row, k)];
      v.free_slot = SW_MASS_SLOT | row; v.c = make_uint4(0, 0, 0, 0); v.c_have = true;
      if (a) { v.slot = v.free_slot; v.e = m_unpack(D, x, a, b, c, v.c.x); return v; }
      v.slot = NONE;
    } else
    v.slot = vt_probe(D, l, x, D.vt[(size_t)vt_home(D, x) * NL + l], v.e, v.free_slot);
    // no explicit view: the base row's — except that a node always sees ITSELF alive at its own incarnation
    if (v.slot == NONE) v.e = make_uint4(x, x == o ? SW_KEY(self_inc, SWIM_STATE_ALIVE) : base_key_of(D, r, x, v.w), 0, 0);
    return v;
  }
  // make the view explicit (created from the base row).  false = the observer already holds view_cap explicit views
  // (its view of itself always fits): the caller ignores the rumour, counted in view_drops.
  __device__ __forceinline__ bool make(View& v, uint32_t x) {
    if (v.slot != NONE) return true;
    if (v_mass(v)) { v.slot = v.free_slot; v.fresh = true; mcnt_add++; if (D.mD) D.mD[m_idx(D, r, v.free_slot & ~SW_MASS_SLOT, k)] = 0; return true; }     // the dense store has room for every observer (the count: once per receiver, in store())
    need_vm();
    if (v.free_slot == NONE) { S.add(ST_VIEW_DROPS); return false; }
    if (vm.x >= D.view_cap + (x == o ? 1u : 0u)) {
      // Full.  memberlist's resetNodes forgets a node dead for longer than GossipToTheDeadTime; so does a full table,
      // one node at a time: the longest-settled Dead/Left view (ties: lowest id; never the node's view of itself)
      // makes room and the observer falls back to the base row for that subject.  Nothing that old: drop.
      const uint32_t now = now_ms(D, t);
      if (now < vm.w) { S.add(ST_VIEW_DROPS); return false; }          // nothing can be that old yet: no need to look
      uint32_t vs = NONE, vsince = 0, vsubj = 0, ev1 = NONE, ev2 = NONE;   // the two earliest evictable times
      for (uint32_t sl = 0; sl < D.VT; sl++) {
        const uint4 c = D.vt[(size_t)sl * NL + l];
        if (c.x == VT_EMPTY || c.x == o || SW_KST(c.y) < SWIM_STATE_DEAD) continue;
        const uint32_t ev = c.z + D.gossip_to_dead_ms + 1;
        if (ev < ev1) { ev2 = ev1; ev1 = ev; } else if (ev < ev2) ev2 = ev;
        if (!(now - c.z > D.gossip_to_dead_ms)) continue;
        if (vs == NONE || c.z < vsince || (c.z == vsince && c.x < vsubj)) { vs = sl; vsince = c.z; vsubj = c.x; }
      }
      vm_dirty = true;
      if (vs == NONE) { vm.w = ev1; S.add(ST_VIEW_DROPS); return false; }
      vm.w = ev2;                                                      // the victim had the earliest one
      const uint32_t wv = D.nw[(size_t)r * D.N + vsubj];
      if (NW_HAS_SLOT(wv)) D.slot_dirty[(size_t)r * D.S + NW_SLOT(wv)] = 1;
      if (dyn() && SW_KINC(base_key_of(D, r, vsubj, wv)) == 0) D.vnk[l]--;
      vt_erase(D, l, vs); vm.x--; S.add(ST_VIEW_EVICT);
      uint4 dummy; vt_probe(D, l, x, D.vt[(size_t)vt_home(D, x) * NL + l], dummy, v.free_slot);   // the layout changed
      if (v.free_slot == NONE) { S.add(ST_VIEW_DROPS); return false; }
    }
    vm.x++; vm_dirty = true;
    if (dyn() && x != o && v.e.y < 4u) D.vnk[l]++;          // a node the base row has never heard of: this observer now has (itself it counts from the start)
    v.slot = v.free_slot; v.fresh = true;
    if (D.vs) D.vs[(size_t)v.slot * NL + l] = 0;           // (serf: no intent applied to it yet)
    if (!(v.w & NW_SUBJECT)) {                              // first explicit view of x on this shard
      const uint32_t old = atomicOr(&D.nw[(size_t)r * D.N + x], NW_SUBJECT);
      if (!(old & NW_SUBJECT)) exc_note(D, r, x, old, old | NW_SUBJECT);
    }
    return true;
  }
  __device__ __forceinline__ void set_view(View& v, uint32_t inc, uint32_t st, bool touch_since) {
    const uint32_t old = v.fresh ? (uint32_t)SWIM_STATE_ALIVE : SW_KST(v.e.y);     // (a fresh view comes from the base row: never Suspect)
    const uint32_t old_key = v.e.y;                         // (of a fresh view: what the base row says)
    v.fresh = false;
    v.e.y = SW_KEY(inc, st);
    if (touch_since) v.e.z = now_ms(D, t);
    if (!v_mass(v) && (old == SWIM_STATE_SUSPECT) != (st == SWIM_STATE_SUSPECT)) {
      need_vm(); vm_dirty = true;
      if (st == SWIM_STATE_SUSPECT) vm.y++;
      else if (--vm.y == 0) vm.z = NONE;                               // no timer left: the bound is exact again
    }
    if (st >= SWIM_STATE_DEAD && !v_mass(v)) { need_vm(); const uint32_t ev = v.e.z + D.gossip_to_dead_ms + 1; if (ev < vm.w) { vm.w = ev; vm_dirty = true; } }
    if (NW_HAS_SLOT(v.w)) {
      const size_t sidx = (size_t)r * D.S + NW_SLOT(v.w);
      const uint32_t mx = D.slot_maxinc[sidx];
      if (inc > mx) { atomicMax(&D.slot_maxinc[sidx], inc); D.slot_dirty[sidx] = 1; }      // what "current" is measured against moved: recount
      else if constexpr (!LQ) D.slot_dirty[sidx] = 1;                                      // stimulus kernels: recount
      else if (v.e.x != o) {                                                               // (a subject's view of itself is not part of its census)
        const uint32_t os = SW_KST(old_key), slot = NW_SLOT(v.w);
        const int dcur = (int)(inc == mx) - (int)(SW_KINC(old_key) == mx);
        if (r == g_s_cen_r && slot < SW_CEN_LDS) {
          if (os != st) { atomicAdd(&g_s_cen[slot * 5 + os], -1); atomicAdd(&g_s_cen[slot * 5 + st], 1); }
          if (dcur) atomicAdd(&g_s_cen[slot * 5 + 4], dcur);
        } else {
          if (os != st) { atomicAdd(&D.cen_dl[sidx * 8 + os], (uint32_t)-1); atomicAdd(&D.cen_dl[sidx * 8 + st], 1u); }
          if (dcur) atomicAdd(&D.cen_dl[sidx * 8 + 4], (uint32_t)dcur);
        }
      }
    }
  }
  __device__ __forceinline__ void arm_deadline(const View& v, uint32_t n0) {   // a suspicion timer was (re)armed: keep the gates' bounds
    const uint32_t dl = v.e.z + (DYN ? susp_timeout_n(D, n0, vw_nconf(v.e.w)) : sel8(D.susp_timeout, vw_nconf(v.e.w) & 7u));
    if (v_mass(v)) { m_arm(D, r, v.free_slot & ~SW_MASS_SLOT, k, dl); return; }
    need_vm();
    if (dl < vm.z) { vm.z = dl; vm_dirty = true; }
    if (dl < dl_new) dl_new = dl;                          // the block's bound is lowered once per block (k_resolve) / by the caller
  }
  __device__ __forceinline__ void refute(View& me, uint32_t accused) {           // me = this node's view of itself (the cached subject)
    uint32_t inc = self_inc + 1;
    if (accused >= inc) inc = accused + 1;
    self_inc = inc;
    uint2 h = D.ph[l];                                     // awareness lives with the probe state
    D.ph[l].y = p_pack(p_epoch(h.y), awareness_apply(D, p_aw(h.y), +1), p_stage(h.y), p_nackm(h.y));
    if (make(me, o)) { me.e.w = 0; set_view(me, inc, SWIM_STATE_ALIVE, false); put_later(me); }
    broadcast(o, SWIM_MSG_ALIVE, inc, 0);
    S.add(ST_REFUTES);
  }
  __device__ __forceinline__ void alive_node(uint32_t x, uint32_t inc, uint32_t upd) {
    const bool local = x == o;
    if (local && leaving) return;
    if (local) {                                           // a node's view of itself carries its own incarnation (header): no lookup
      if (inc <= self_inc) return;
      View me = take_view(o); refute(me, inc); cv = me; return;
    }
    View v = take_view(x);
    alive_other(v, x, inc, upd);
    cv = v;
  }
  __device__ __forceinline__ void alive_other(View& v, uint32_t x, uint32_t inc, uint32_t upd) {
    const uint32_t key = v.e.y;
    if (inc <= SW_KINC(key)) return;
    if (!make(v, x)) return;
    v.e.w = 0;                                             // delete(m.nodeTimers, a.Node)
    const uint32_t old = SW_KST(key);
    broadcast_v(v, x, SWIM_MSG_ALIVE, inc, upd, 0);
    set_view(v, inc, SWIM_STATE_ALIVE, old != SWIM_STATE_ALIVE);
    put_later(v);
    S.add(ST_APPL0);
    if (watching()) {
      if (old == SWIM_STATE_DEAD || old == SWIM_STATE_LEFT) record_event(SWIM_EVENT_MEMBER_JOIN, x, 0, inc);
      else if (upd) record_event(SWIM_EVENT_MEMBER_UPDATE, x, 0, inc);
    }
  }
  __device__ __forceinline__ void suspect_node(uint32_t x, uint32_t inc, uint32_t from) {
    View v = take_view(x);
    suspect_v(v, x, inc, from);
    cv = v;
  }
  __device__ __forceinline__ void suspect_v(View& v, uint32_t x, uint32_t inc, uint32_t from) {
    const uint32_t key = v.e.y;
    if (inc < SW_KINC(key)) return;
    if (SW_KST(key) == SWIM_STATE_SUSPECT) {           // timer exists: suspicion.Confirm(from) (the base row is never Suspect)
      uint32_t nc = vw_nconf(v.e.w);
      if ((!dyn() && nc >= D.susp_k) || vw_conf0(v.e.w) == from) return;
      const size_t ci = (size_t)v.slot * NL + l;          // (a pair of the dense store carries its accusers along: c_have)
      if (!v.c_have) { v.c = (nc || dyn()) ? D.vc[ci] : make_uint4(0, 0, 0, 0); v.c_have = true; }   // (dynamic membership: the timer's n sits in c.w)
      uint4 b = v.c;
      if (dyn() && nc >= susp_k_n(D, b.w)) return;
      if ((nc >= 1 && b.x == from) || (nc >= 2 && b.y == from) || (nc >= 3 && b.z == from)) return;
      nc++;
      if (nc == 1) b.x = from; else if (nc == 2) b.y = from; else if (nc == 3) b.z = from;
      if (nc <= 3) { if (!v_mass(v)) D.vc[ci] = b; v.c = b; }
      v.e.w = vw_pack(vw_conf0(v.e.w), nc, vw_leaving(v.e.w)); put_later(v);
      arm_deadline(v, b.w);
      // (a confirmation changes neither the state nor the incarnation: nothing a census or a trace row shows — re-counting the
      // slot's 65 536 observers every tick of the confirmation phase was a twelfth of the driver window's kernel time)
      S.add(ST_CONFIRMS);
      broadcast_v(v, x, SWIM_MSG_SUSPECT, inc, from, inc - SW_KINC(key));
      return;
    }
    if (SW_KST(key) != SWIM_STATE_ALIVE) return;
    if (x == o) { refute(v, inc); return; }
    if (!make(v, x)) return;
    broadcast_v(v, x, SWIM_MSG_SUSPECT, inc, from, 0);
    set_view(v, inc, SWIM_STATE_SUSPECT, true);
    v.e.w = vw_pack(from, 0, (v.e.w >> 1) & 1u);           // newSuspicion(from, k, min, max); a Leaving mark stays
    put_later(v);
    uint32_t n0 = 0;
    if (dyn() && !v_mass(v)) { n0 = est_n(D, r, l); D.vc[(size_t)v.slot * NL + l] = make_uint4(0, 0, 0, n0); }   // k, min, max from estNumNodes() now
    v.c = make_uint4(0, 0, 0, n0); v.c_have = true;
    arm_deadline(v, n0);
    S.add(ST_APPL1);
  }
  __device__ __forceinline__ void dead_node(uint32_t x, uint32_t inc, uint32_t from) {
    View v = take_view(x);
    dead_v(v, x, inc, from);
    cv = v;
  }
  __device__ __forceinline__ void dead_v(View& v, uint32_t x, uint32_t inc, uint32_t from) {
    const uint32_t key = v.e.y;
    if (inc < SW_KINC(key)) return;
    const uint32_t old = SW_KST(key);
    if (old == SWIM_STATE_DEAD || old == SWIM_STATE_LEFT) return;
    if (x == o && !leaving) { refute(v, inc); return; }
    if (!make(v, x)) return;
    // Left for a graceful leave (Node == From) and for a member a leave intent had marked Leaving here (serf handleNodeLeave)
    const bool was_leaving = old == SWIM_STATE_SUSPECT ? vw_leaving(v.e.w) != 0 : ((v.e.w >> 1) & 1u) != 0;
    v.e.w = 0;
    broadcast_v(v, x, SWIM_MSG_DEAD, inc, from, 0);
    const uint32_t st = (from == x || was_leaving) ? SWIM_STATE_LEFT : SWIM_STATE_DEAD;
    set_view(v, inc, st, true);
    put_later(v);
    S.add(ST_APPL2);
    if (x != o && watching()) record_event(st == SWIM_STATE_LEFT ? SWIM_EVENT_MEMBER_LEAVE : SWIM_EVENT_MEMBER_FAILED, x, 0, inc);
  }
  // sendMsg (net.go): extra := getBroadcasts(compoundOverhead, UDPBufferSize - len(msg) - compoundHeaderOverhead),
  // i.e. the memberlist queue and then the serf delegate's user events, for a ping/ack/... this node sent this
  // tick.  Runs before the tick's arrivals are merged; what is picked goes to the block's carry area and
  // reaches `receiver` with the next tick's packets (NONE = the carrier was lost: transmits still count).
  __device__ __forceinline__ void piggyback(uint32_t receiver, uint32_t kind, uint32_t* s_carry, uint4* area, uint32_t* lds_emeta) {
    const int limit = (int)D.budget - (int)sel4(D.ctl_len, kind & 3u);
    uint32_t live_m = qlen >= 32 ? 0xFFFFFFFFu : (1u << qlen) - 1, live_e = evqlen >= 32 ? 0xFFFFFFFFu : (1u << evqlen) - 1;
    int used = 0, used2 = 0;
    HbmQ qe{D.evq + l, NL};
    // the user-event queue stays in HBM; the pick walks it several times, so its meta words are fetched once into LDS
    MetaQT<SW_RES_THREADS> me{lds_emeta + threadIdx.x};
    if constexpr (SERF) for (uint32_t j = 0; j < evqlen; j++) me.meta(j) = qe.at(j).w;
    const uint32_t rl = DYN ? retransmit_limit_n(D, est_n(D, r, l)) : D.retransmit_limit;
    uint32_t tm, te = 0;
    if constexpr (SPLIT) tm = get_broadcasts(D, LdsQ2T<SW_RES_THREADS>{(uint2*)g_lds_dyn + threadIdx.x}, qlen, live_m, 2, limit, used, rl);
    else tm = get_broadcasts(D, LdsQT<SW_RES_THREADS>{g_lds_dyn + threadIdx.x}, qlen, live_m, 2, limit, used, rl);      // (piggyback() is k_resolve's: LQ)
    int avail = limit - used;
    if constexpr (SERF) if (D.EQ && avail > 2 + 1) te = get_broadcasts(D, me, evqlen, live_e, 3, avail, used2, rl);
    if (!(tm | te)) return;
    const uint32_t cnt = (uint32_t)(__popc(tm) + __popc(te));
    c_pig++;
    for (uint32_t m = tm; m; m &= m - 1) { uint32_t ty = m_type(mq_w(__ffs(m) - 1)); if (ty < 2) c_sent01 += 1u << (16 * ty); else c_sent23 += 1u << (16 * (ty - 2)); }
    c_sent23 += (uint32_t)__popc(te) << 16;
    if (receiver != NONE) {
      const uint32_t gdst = r * D.N + receiver;
      const bool att = *D.att_any && (D.nw[gdst] & NW_ATTACHED);   // Transport.WriteTo towards the real node
      uint32_t pos = att ? 0 : atomicAdd(s_carry, cnt);
      if (!att && pos + cnt > D.carry_cap) { atomicOr(D.err, SW_ERR_CARRY_OVF); pos = NONE; }
      for (uint32_t m = tm; m && pos != NONE; m &= m - 1) {
        uint4 e = mq_get(__ffs(m) - 1); uint32_t meta = (m_type(e.w) << 30) | (e.z & 0x3FFFFFFFu);
        if (att) capture(D, o, gdst, e.x, e.y, meta); else area[pos++] = make_uint4(gdst, e.x, e.y, meta);
      }
      if constexpr (SERF) for (uint32_t m = te; m && pos != NONE; m &= m - 1) {
        uint4 e = qe.at(__ffs(m) - 1);
        if (att) capture(D, o, gdst, e.x, e.y, (uint32_t)SWIM_MSG_USER << 30); else area[pos++] = make_uint4(gdst, e.x, e.y, (uint32_t)SWIM_MSG_USER << 30);
      }
    }
    // the bumped transmit counts; what reached the retransmit limit retires (stable compaction, like the gossip role's
    // write-back)
    uint32_t nq = 0, ne = 0;
    for (uint32_t j = 0; j < qlen; j++) if ((live_m >> j) & 1u) {
      if constexpr (SPLIT) { if (nq != j) mq_set(nq, mq_get(j)); else if ((tm >> j) & 1u) qdirty |= 1u << j; }
      else if constexpr (LQ) { if (nq != j) { SQ(nq) = SQ(j); qdirty |= 1u << nq; } else if ((tm >> j) & 1u) qdirty |= 1u << j; }
      else if (nq != j) QENT(nq, l) = QENT(j, l);
      nq++;
    }
    if constexpr (SERF) for (uint32_t j = 0; j < evqlen; j++) if ((live_e >> j) & 1u) {
      if (ne != j) { uint4 e = qe.at(j); e.w = me.meta(j); qe.at(ne) = e; } else if ((te >> j) & 1u) qe.at(j).w = me.meta(j);
      ne++;
    }
    qlen = nq; if constexpr (SERF) evqlen = ne;
  }
  // serf's member.statusLTime of the cached view (0: a base-row view, or no intent applied yet) — read and written in place, not
  // carried in the View (only intents look at it)
  __device__ __forceinline__ uint32_t slt_of(const View& v) const {
    if (!D.vs || v.slot == NONE) return 0u;
    return v_mass(v) ? D.mD[m_idx(D, r, v.free_slot & ~SW_MASS_SLOT, k)] : D.vs[(size_t)v.slot * NL + l];
  }
  __device__ __forceinline__ void slt_set(const View& v, uint32_t lt) {      // (v is explicit: make() succeeded)
    if (v_mass(v)) D.mD[m_idx(D, r, v.free_slot & ~SW_MASS_SLOT, k)] = lt; else D.vs[(size_t)v.slot * NL + l] = lt;
  }
  // serf handleNodeLeaveIntent.  Returns 1 when the intent is rebroadcast (serf's return value), 0 when not, 2 when it is about this
  // agent itself and must be refuted (the caller broadcasts a join intent: "go s.broadcastJoin(s.clock.Time())").  Order as upstream:
  // the member's statusLTime first (an intent stamped no later than the last one applied here is stale), then the refutation, then the
  // transition by status: a member held Failed becomes Left (EventMemberLeave), with prune it is erased at once (EventMemberReap), also
  // when it was Left already; a member that is Alive or Suspect here is marked Leaving — when memberlist declares it dead it becomes
  // Left, not Failed.  Not modelled: serf's recentIntents (an intent about a member this node has never heard of is passed on only).
  __device__ __forceinline__ int leave_intent(uint32_t x, bool prune, uint32_t ltime) {
    if (x >= D.N) return 0;
    if (x == o) {
      const uint32_t ss = D.sslt[l];
      if (ltime <= (ss & 0x7FFFFFFFu)) return 0;
      if (!(ss >> 31)) return 2;
      D.sslt[l] = 0x80000000u | ltime;                     // its own Leave(): StatusLeaving until memberlist's leave goes out
      return 1;
    }
    View v = take_view(x);
    const int rc = leave_intent_v(v, x, prune, ltime);
    cv = v;
    return rc;
  }
  __device__ __forceinline__ int leave_intent_v(View& v, uint32_t x, bool prune, uint32_t ltime) {
    const uint32_t key = v.e.y, st = SW_KST(key);
    if (SW_KINC(key) == 0) return 1;
    if (ltime <= slt_of(v)) return 0;                      // "If the message is old, then it is irrelevant and we can skip it"
    if (st < SWIM_STATE_DEAD) {                            // alive (or suspected) here: StatusLeaving — its death will read as a leave
      if (!make(v, x)) return 1;
      slt_set(v, ltime);
      const uint32_t bit = st == SWIM_STATE_SUSPECT ? 8u : 2u;
      if (v.fresh || !(v.e.w & bit)) {
        v.e.w |= bit; v.fresh = false; put_later(v);
        if (NW_HAS_SLOT(v.w)) D.slot_dirty[(size_t)r * D.S + NW_SLOT(v.w)] = 1;
      }
      return 1;
    }
    // erased already (serf no longer has the member; a Failed / Left member of the base row was erased before it got there)
    if (v.slot != NONE ? (v.e.w & 1u) != 0 : D.reap_period != 0) return 1;
    if (st == SWIM_STATE_LEFT && !prune) return 1;
    if (!make(v, x)) return 1;
    v.e.w = 0;
    if (st == SWIM_STATE_DEAD) {
      set_view(v, SW_KINC(key), SWIM_STATE_LEFT, true);
      slt_set(v, ltime);
      S.add(ST_INTENTS);
      if (watching()) record_event(SWIM_EVENT_MEMBER_LEAVE, x, 0, SW_KINC(key));
    } else if (v_mass(v)) v.fresh = false;
    else { v.fresh = false; need_vm(); const uint32_t ev = v.e.z + D.gossip_to_dead_ms + 1; if (ev < vm.w) { vm.w = ev; vm_dirty = true; } }
    if (prune) {
      v.e.w = 1u; S.add(ST_REAPED);
      if (NW_HAS_SLOT(v.w)) D.slot_dirty[(size_t)r * D.S + NW_SLOT(v.w)] = 1;
      if (watching()) record_event(SWIM_EVENT_MEMBER_REAP, x, 0, SW_KINC(key));
    }
    put_later(v);
    return 1;
  }
  // serf handleNodeJoinIntent: a newer join intent moves the member's statusLTime and takes a Leaving mark back ("the leaving message must
  // have been for an older time")
  __device__ __forceinline__ int join_intent(uint32_t x, uint32_t ltime) {
    if (x >= D.N) return 0;
    if (x == o) {
      const uint32_t ss = D.sslt[l];
      if (ltime <= (ss & 0x7FFFFFFFu)) return 0;
      D.sslt[l] = (ss & 0x80000000u) | ltime;
      return 1;
    }
    View v = take_view(x);
    const int rc = join_intent_v(v, x, ltime);
    cv = v;
    return rc;
  }
  __device__ __forceinline__ int join_intent_v(View& v, uint32_t x, uint32_t ltime) {
    const uint32_t key = v.e.y, st = SW_KST(key);
    if (SW_KINC(key) == 0) return 1;                       // not a member here: passed on
    if (st >= SWIM_STATE_DEAD && (v.slot != NONE ? (v.e.w & 1u) != 0 : D.reap_period != 0)) return 1;     // erased: likewise
    if (ltime <= slt_of(v)) return 0;
    if (!make(v, x)) return 1;
    slt_set(v, ltime);
    if (st < SWIM_STATE_DEAD) {
      const uint32_t bit = st == SWIM_STATE_SUSPECT ? 8u : 2u;
      if (!v.fresh && (v.e.w & bit)) { v.e.w &= ~bit; if (NW_HAS_SLOT(v.w)) D.slot_dirty[(size_t)r * D.S + NW_SLOT(v.w)] = 1; }
    } else if (v.fresh && !v_mass(v)) { need_vm(); const uint32_t ev = v.e.z + D.gossip_to_dead_ms + 1; if (ev < vm.w) { vm.w = ev; vm_dirty = true; } }
    v.fresh = false; put_later(v);
    return 1;
  }
  // serf handleUserEvent + LamportClock.Witness.  An event-buffer slot (one per LTime mod EventBuffer) is EW 16-byte words,
  // slot-major: word 0 = {ltime, n, id0, id1}, then four ids per word — serf's slot is an unbounded list of the events
  // stamped with that LTime; ours holds 4*EW - 2 (swim_config.event_ids_per_ltime; a flood stamps many events alike).
  // An id with bit 31 set is one of serf's intents: rebroadcast as its handler says (the origin always sends).  A leave intent about this
  // agent itself, while it is not leaving, is answered in a second pass with the agent's own join intent stamped clock.Time().
  __device__ __forceinline__ void user_event(uint32_t id, uint32_t ltime, bool origin = false) {
    if constexpr (!SERF) { (void)id; (void)ltime; (void)origin; return; } else {
    if (!(D.flags & SWIM_F_SERF_EVENTS)) return;
    for (int pass = 0; pass < 2; pass++) {
      if (ltime >= ev_clock) ev_clock = ltime + 1;
      if (ev_clock > D.EB && ltime < ev_clock - D.EB) { S.add(ST_UEV_STALE); break; }
      uint4* const slot = D.ring + (size_t)(ltime % D.EB) * D.EW * NL + l;      // word j at slot[j * NL]
      uint4 w0 = slot[0]; uint32_t n = w0.y;
      if (n && w0.x == ltime) {
        bool dup = w0.z == id || (n >= 2 && w0.w == id);
        for (uint32_t j = 1; !dup && 4 * j - 2 < n; j++) {
          const uint4 w = slot[(size_t)j * NL]; const uint32_t m = n - (4 * j - 2);
          dup = w.x == id || (m >= 2 && w.y == id) || (m >= 3 && w.z == id) || (m >= 4 && w.w == id);
        }
        if (dup) { S.add(ST_UEV_DEDUP); break; }
      } else n = 0;
      if (n == 4 * D.EW - 2) { S.add(ST_EVDROPS); break; }
      if (n == 0) w0.z = id; else if (n == 1) w0.w = id;
      else {
        uint32_t* w = (uint32_t*)&slot[(size_t)((n + 2) / 4) * NL];
        w[(n + 2) & 3u] = id;
      }
      n++; w0.x = ltime; w0.y = n; slot[0] = w0;
      int rc = 1;
      if (id & SWIM_INTENT_LEAVE)
        rc = (id & SWIM_INTENT_JOIN) == SWIM_INTENT_JOIN ? join_intent(id & 0x1FFFFFFFu, ltime) : leave_intent(id & 0x1FFFFFFFu, (id & SWIM_INTENT_PRUNE) != 0, ltime);
      else {
        S.add(ST_UEV_DELIVERED);
        if (watching()) record_event(SWIM_EVENT_USER, id, ltime, 0);
      }
      if (rc == 1 || origin) {
        uint32_t seq = D.evseq[l]; D.evseq[l] = seq + 1;
        queue_push<true>(D.EQ, evqlen, seq, false, id, SWIM_MSG_USER, ltime, 0, ST_EVDROPS);
      }
      if (rc != 2) break;
      id = SWIM_INTENT_JOIN | o; ltime = ev_clock; origin = true;       // the refutation: broadcastJoin(s.clock.Time())
    }
    }
  }
};
#undef SQ
#undef SQ2
typedef NodeCtxT<false, true> NodeCtx;       // the stimulus kernels edit the queue in HBM

// canonical order key of an inbox record: (user?, subject, type) then (incarnation, from)
__device__ __forceinline__ void edge_key(uint4 e, uint64_t& hi, uint64_t& lo) {
  uint32_t type = e.w >> 30; bool order = e.y == SWIM_SUBJECT_PIGGY;     // orders act on the queue as k_begin left it
  hi = ((uint64_t)!order << 35) | ((uint64_t)(!order && type == SWIM_MSG_USER) << 34) | ((uint64_t)e.y << 2) | type;
  lo = ((uint64_t)e.z << 32) | (e.w & 0x3FFFFFFFu);
}

// A big inbox (a push-pull delivers a whole table in one tick) is sorted once instead of being searched for its minimum
// per message: in-place heapsort of the 12-byte records {subject, incarnation, meta} by the canonical key, ascending.
__device__ __forceinline__ bool rec_less(const uint32_t* a, uint32_t i, uint32_t j) {
  uint64_t hi, lo, hj, lj;
  edge_key(make_uint4(0, a[3 * i], a[3 * i + 1], a[3 * i + 2]), hi, lo);
  edge_key(make_uint4(0, a[3 * j], a[3 * j + 1], a[3 * j + 2]), hj, lj);
  return hi < hj || (hi == hj && lo < lj);
}
__device__ __forceinline__ void rec_swap(uint32_t* a, uint32_t i, uint32_t j) {
  for (int w = 0; w < 3; w++) { uint32_t t = a[3 * i + w]; a[3 * i + w] = a[3 * j + w]; a[3 * j + w] = t; }
}
__device__ __attribute__((noinline)) void inbox_heapsort(uint32_t* a, uint32_t n) {
  for (uint32_t start = n / 2; start-- > 0; )                      // heapify
    for (uint32_t root = start;;) {
      uint32_t c = 2 * root + 1; if (c >= n) break;
      if (c + 1 < n && rec_less(a, c, c + 1)) c++;
      if (!rec_less(a, root, c)) break;
      rec_swap(a, root, c); root = c;
    }
  for (uint32_t end = n; end-- > 1; ) {
    rec_swap(a, 0, end);
    for (uint32_t root = 0;;) {
      uint32_t c = 2 * root + 1; if (c >= end) break;
      if (c + 1 < end && rec_less(a, c, c + 1)) c++;
      if (!rec_less(a, root, c)) break;
      rec_swap(a, root, c); root = c;
    }
  }
}
#define SW_INBOX_SORT_MIN 12      /* from this many messages on the inbox is sorted rather than searched */

// A state exchange in the middle of a mass event hands a node thousands of messages in one tick (config #4: 3 800 at 262 144
// nodes, 7 600 at 524 288).  Heap-sorting them from ONE lane is tens of thousands of dependent round trips — 20 ms of a tick in
// which the rest of the device waits.  So before k_resolve the workgroup that owns the node sorts such an inbox in LDS (bitonic,
// the same canonical key; pads sort last) and leaves it in the overflow row, line messages included; k_resolve then walks it.
__global__ void __launch_bounds__(SW_BLOCK) k_inbox_sort(const SwDev* __restrict__ Dp, uint32_t cap) {
  SW_DEV_BIND
  uint32_t* const sy = (uint32_t*)g_lds_dyn; uint32_t* const sz = sy + cap; uint32_t* const sw = sz + cap;
  __shared__ uint32_t s_big[SW_BLOCK], s_nbig;
  const size_t NL = (size_t)D.R * D.nloc, l0 = (size_t)blockIdx.x * SW_BLOCK, l = l0 + threadIdx.x;
  uint32_t c = l < NL ? D.in_cnt[l] : 0u;
  if (c > D.C) c = D.C;
  const bool big = c >= SW_BIGSORT_MIN && c <= D.bigsort_cap;
  if (!__syncthreads_or(big)) return;
  if (threadIdx.x == 0) s_nbig = 0;
  __syncthreads();
  if (big) s_big[atomicAdd(&s_nbig, 1u)] = (c << 8) | threadIdx.x;
  __syncthreads();
  const uint32_t nbig = s_nbig;
  for (uint32_t b = 0; b < nbig; b++) {
    const uint32_t ent = s_big[b], n = ent >> 8;
    const size_t ll = l0 + (ent & 255u);
    uint32_t* const row = inbox_row(D, ll);
    const uint32_t* const line = D.inbox1 + ll * 16;
    uint32_t P = SW_BIGSORT_MIN; while (P < n) P <<= 1;
    for (uint32_t i = threadIdx.x; i < P; i += SW_BLOCK) {
      uint32_t y = NONE, z = NONE, w = NONE;                      // a pad: the greatest key there is
      if (i < SW_INBOX_FAST) { y = line[1 + 3 * i]; z = line[2 + 3 * i]; w = line[3 + 3 * i]; }
      else if (i < n) { const uint32_t* m = row + (size_t)(i - SW_INBOX_FAST) * 3; y = m[0]; z = m[1]; w = m[2]; }
      sy[i] = y; sz[i] = z; sw[i] = w;
    }
    __syncthreads();
    for (uint32_t k = 2; k <= P; k <<= 1)
      for (uint32_t j = k >> 1; j; j >>= 1) {
        for (uint32_t p = threadIdx.x; p < P / 2; p += SW_BLOCK) {
          const uint32_t i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), q = i | j;
          const uint4 a = make_uint4(0, sy[i], sz[i], sw[i]), bb = make_uint4(0, sy[q], sz[q], sw[q]);
          uint64_t ah, al, bh, bl; edge_key(a, ah, al); edge_key(bb, bh, bl);
          const bool gt = ah > bh || (ah == bh && al > bl);
          if (gt == ((i & k) == 0)) { sy[i] = bb.y; sz[i] = bb.z; sw[i] = bb.w; sy[q] = a.y; sz[q] = a.z; sw[q] = a.w; }
        }
        __syncthreads();
      }
    for (uint32_t i = threadIdx.x; i < n; i += SW_BLOCK) { uint32_t* m = row + (size_t)i * 3; m[0] = sy[i]; m[1] = sz[i]; m[2] = sw[i]; }
    __syncthreads();
  }
}

// ... and the inboxes beyond what LDS holds (round 5; config #4's RECOVERY: the first state exchange after a partition of 65 536 nodes heals
// hands a node 62 000 messages — the lane's heapsort of that is two million dependent round trips, seconds of a tick).  A workgroup per
// such inbox, a bitonic network in its "ascending only" form (the first step of a level compares mirrored positions, the others
// positions j apart: the smaller key always goes to the lower index, so the pads — every position from n on, the greatest key there is,
// never stored — stay where they are): the levels up to SW_BIGSORT_MAX and, of every higher level, the steps inside SW_BIGSORT_MAX-element
// chunks run in LDS a chunk at a time; only the steps that cross chunks (six for 65 536 messages) work on the row in global memory.
__device__ __forceinline__ bool rec_gt(uint32_t ay, uint32_t az, uint32_t aw, uint32_t by, uint32_t bz, uint32_t bw) {
  uint64_t ah, al, bh, bl; edge_key(make_uint4(0, ay, az, aw), ah, al); edge_key(make_uint4(0, by, bz, bw), bh, bl);
  return ah > bh || (ah == bh && al > bl);
}
__global__ void __launch_bounds__(SW_BLOCK) k_inbox_sort_huge(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  constexpr uint32_t S = SW_BIGSORT_MAX;
  uint32_t* const sy = (uint32_t*)g_lds_dyn; uint32_t* const sz = sy + S; uint32_t* const sw = sz + S;
  __shared__ uint32_t s_big[SW_BLOCK], s_nbig;
  const size_t NL = (size_t)D.R * D.nloc, l0 = (size_t)blockIdx.x * SW_BLOCK, l = l0 + threadIdx.x;
  uint32_t c = l < NL ? D.in_cnt[l] : 0u;
  if (c > D.C) c = D.C;
  const bool big = c > D.bigsort_cap && D.bigsort_cap != 0;
  if (!__syncthreads_or(big)) return;
  if (threadIdx.x == 0) s_nbig = 0;
  __syncthreads();
  if (big) s_big[atomicAdd(&s_nbig, 1u)] = threadIdx.x;
  __syncthreads();
  const uint32_t nbig = s_nbig;
  for (uint32_t b = 0; b < nbig; b++) {
    const size_t ll = l0 + s_big[b];
    uint32_t n = D.in_cnt[ll]; if (n > D.C) n = D.C;
    uint32_t* const row = inbox_row(D, ll);
    const uint32_t* const line = D.inbox1 + ll * 16;
    // the five messages of the line join the row (arrivals 6.. sit at 0.. : the row has room for all C)
    if (threadIdx.x < SW_INBOX_FAST) { uint32_t* m = row + (size_t)(n - SW_INBOX_FAST + threadIdx.x) * 3; m[0] = line[1 + 3 * threadIdx.x]; m[1] = line[2 + 3 * threadIdx.x]; m[2] = line[3 + 3 * threadIdx.x]; }
    __threadfence_block(); __syncthreads();
    uint32_t P = S; while (P < n) P <<= 1;
    // the steps of level k that stay inside a chunk of S positions (all of them for k <= S; j = S/2 .. 1 of a higher level), a chunk at a time in LDS
    auto lds_pass = [&](uint32_t k_first, uint32_t k_last, bool tail_only) {
      for (uint32_t c0 = 0; c0 < n; c0 += S) {
        const uint32_t m = n - c0 < S ? n - c0 : S;                 // real elements of this chunk; beyond: pads
        for (uint32_t i = threadIdx.x; i < S; i += SW_BLOCK) {
          uint32_t y = NONE, z = NONE, w = NONE;
          if (i < m) { const uint32_t* e = row + (size_t)(c0 + i) * 3; y = e[0]; z = e[1]; w = e[2]; }
          sy[i] = y; sz[i] = z; sw[i] = w;
        }
        __syncthreads();
        for (uint32_t k = k_first; k <= k_last; k <<= 1)
          for (uint32_t j = tail_only ? S >> 1 : k >> 1; j; j >>= 1) {
            const bool flip = !tail_only && j == k >> 1;
            for (uint32_t p = threadIdx.x; p < S / 2; p += SW_BLOCK) {
              const uint32_t i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), q = flip ? i ^ (k - 1) : i | j;
              if (rec_gt(sy[i], sz[i], sw[i], sy[q], sz[q], sw[q])) {
                const uint32_t ty = sy[i], tz = sz[i], tw = sw[i];
                sy[i] = sy[q]; sz[i] = sz[q]; sw[i] = sw[q]; sy[q] = ty; sz[q] = tz; sw[q] = tw;
              }
            }
            __syncthreads();
          }
        for (uint32_t i = threadIdx.x; i < m; i += SW_BLOCK) { uint32_t* e = row + (size_t)(c0 + i) * 3; e[0] = sy[i]; e[1] = sz[i]; e[2] = sw[i]; }
        __threadfence_block(); __syncthreads();
      }
    };
    lds_pass(2, S, false);
    for (uint32_t k = S << 1; k <= P; k <<= 1) {
      for (uint32_t j = k >> 1; j >= S; j >>= 1) {                  // the steps that cross chunks: on the row itself
        const bool flip = j == k >> 1;
        for (uint32_t p = threadIdx.x; p < P / 2; p += SW_BLOCK) {
          const uint32_t i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), q = flip ? i ^ (k - 1) : i | j;
          if (q >= n) continue;                                     // a pad: never smaller
          uint32_t* a = row + (size_t)i * 3; uint32_t* e = row + (size_t)q * 3;
          const uint32_t ay = a[0], az = a[1], aw = a[2], by = e[0], bz = e[1], bw = e[2];
          if (rec_gt(ay, az, aw, by, bz, bw)) { a[0] = by; a[1] = bz; a[2] = bw; e[0] = ay; e[1] = az; e[2] = aw; }
        }
        __threadfence_block(); __syncthreads();
      }
      lds_pass(k, k, true);
    }
  }
}

// ... and the inboxes in between (SW_INBOX_SORT_MIN .. SW_BIGSORT_MIN - 1 messages: every node's, tick after tick, in the mass phase
// of config #4 — profiles/r03_config4_resolve_phase_clock.txt: their heapsort by one lane in global memory was ~3.5 of the 8 ms a
// wave of k_resolve spent on a tick) by one WAVE each, in the wave's own 1.5 KB strip of LDS: no workgroup barrier, 6 KB of LDS
// per workgroup, so every wave slot of the device sorts at once (k_resolve itself runs one workgroup per CU at queue_cap 32).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__global__ void __launch_bounds__(SW_BLOCK) k_inbox_sort_med(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  __shared__ uint32_t s_strip[(SW_BLOCK / 64) * 3 * SW_BIGSORT_MIN];
  const size_t NL = (size_t)D.R * D.nloc, l0 = (size_t)blockIdx.x * SW_BLOCK, l = l0 + threadIdx.x;
  uint32_t c = l < NL ? D.in_cnt[l] : 0u;
  if (c > D.C) c = D.C;
  const uint32_t wv = threadIdx.x / 64, ln = sw_lane();
  uint32_t* const wy = s_strip + wv * 3 * SW_BIGSORT_MIN; uint32_t* const wz = wy + SW_BIGSORT_MIN; uint32_t* const ww = wz + SW_BIGSORT_MIN;
  for (uint64_t mm = __ballot(c >= SW_INBOX_SORT_MIN && c < SW_BIGSORT_MIN); mm; mm &= mm - 1) {
    const uint32_t src = (uint32_t)__ffsll((long long)mm) - 1, n = __shfl(c, src);
    const size_t ll = l0 + wv * 64 + src;
    uint32_t* const row = inbox_row(D, ll);
    const uint32_t* const line = D.inbox1 + ll * 16;
    uint32_t P = 16; while (P < n) P <<= 1;                        // 16 .. 128
    for (uint32_t i = ln; i < P; i += 64) {
      uint32_t y = NONE, z = NONE, w = NONE;                      // a pad: the greatest key there is
      if (i < SW_INBOX_FAST) { y = line[1 + 3 * i]; z = line[2 + 3 * i]; w = line[3 + 3 * i]; }
      else if (i < n) { const uint32_t* m = row + (size_t)(i - SW_INBOX_FAST) * 3; y = m[0]; z = m[1]; w = m[2]; }
      wy[i] = y; wz[i] = z; ww[i] = w;
    }
    wave_lds_sync();
    for (uint32_t k = 2; k <= P; k <<= 1)
      for (uint32_t j = k >> 1; j; j >>= 1) {
        if (ln < P / 2) {
          const uint32_t i = ((ln & ~(j - 1)) << 1) | (ln & (j - 1)), q = i | j;
          const uint4 a = make_uint4(0, wy[i], wz[i], ww[i]), bb = make_uint4(0, wy[q], wz[q], ww[q]);
          uint64_t ah, al, bh, bl; edge_key(a, ah, al); edge_key(bb, bh, bl);
          const bool gt = ah > bh || (ah == bh && al > bl);
          if (gt == ((i & k) == 0)) { wy[i] = bb.y; wz[i] = bb.z; ww[i] = bb.w; wy[q] = a.y; wz[q] = a.z; ww[q] = a.w; }
        }
        wave_lds_sync();
      }
    for (uint32_t i = ln; i < n; i += 64) { uint32_t* m = row + (size_t)i * 3; m[0] = wy[i]; m[1] = wz[i]; m[2] = ww[i]; }
    wave_lds_sync();
  }
}

// A tile of SW_RTILE node blocks per workgroup.  Who got something this tick is sparse (a fifth of the nodes while a
// rumour saturates a cluster, far fewer otherwise) and a lane's work is a chain of dependent memory round trips: lanes
// that map 1:1 to nodes leave most of every wave idle through the whole chain.  So the workgroup first compacts the
// tile's receivers (one coalesced read of the count words, wave ballots, a 16-entry prefix in LDS: ascending node order,
// no atomics), then walks the compact list 256 at a time with full waves.  Everything indexed by node block (in_any,
// dl_blk, the carry areas k_deliver drains) keeps that index.
// (five waves per SIMD instead of the four the register allocator would settle for: it is the number of lanes in flight
// that hides the round trips)
#ifdef SWIMSIM_DIAG
// diagnostics (SWIMSIM_RESOLVECLK): where a wave of k_resolve spends its life.  Every wave leaves its s_memtime deltas per
// phase in its own row (plain stores, no atomics: the probe must not serialise what it measures); later launches overwrite
// earlier ones, the host reduces at swim_destroy.  Row: [0..5] phases, [6] lifetime, [7] passes << 16 | max messages per lane
#define RCLK_ROWS 65536
__device__ uint32_t g_rclk[RCLK_ROWS][8];
#define RCLK_MARK(p) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); rclk_acc[p] += (uint32_t)(now_ - rclk_t); rclk_t = now_; } while (0)
#else
#define RCLK_MARK(p) do { } while (0)
#endif
#ifdef SWIMSIM_WAVECLK
// diagnostics, second kind (-DSWIMSIM_WAVECLK builds only): four s_memtime stamps per wave of k_resolve (entry, list compacted, list
// walked, tallies flushed), taken OUTSIDE the loops — a handful of scalar instructions, so that the probe does not change
// what it measures the way the per-phase accumulators of SWIMSIM_RESOLVECLK do.  Row per wave; the host keeps the rows of the
// last launch (by entry time) and prints the launch's profile at swim_destroy.
#define WCLK_ROWS 32768
__device__ unsigned long long g_wclk[WCLK_ROWS][6];
#define WCLK(i) wclk[i] = wall_clock64()      /* s_memrealtime: 100 MHz whatever the shader clock does */
#else
#define WCLK(i) do { } while (0)
#endif
#ifndef SW_RESOLVE_WAVES
#define SW_RESOLVE_WAVES 4
#endif
#define SW_ORDER_MIN 16u          /* a tile with an inbox of this many messages has its receiver list ordered by size class */
template <bool MASS, bool SERF, bool DYN>
__global__ void __launch_bounds__(SW_RES_THREADS) __attribute__((amdgpu_waves_per_eu(SW_RESOLVE_WAVES, 8))) k_resolve(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  uint4* const lds_q = g_lds_dyn;                // [Q][threads] the lanes' memberlist queues, then [EQ][threads] words: meta words of their event queues
  constexpr bool RESOLVE_LQ = true;
  // the event queues' meta words sit behind the staged memberlist queue: [Q][threads] entries of 16 bytes — of 8 on a handle with the dense
  // pair store, which stages {subject, meta} only (NodeCtxT::SPLIT)
  uint32_t* const lds_emeta = (MASS && SW_SPLITQ) ? (uint32_t*)((uint2*)lds_q + (size_t)D.Q * SW_RES_THREADS) : (uint32_t*)(lds_q + (size_t)D.Q * SW_RES_THREADS);
  __shared__ uint32_t lds_stats[ST_COUNT];
  // The tile: one node block (256 nodes) — but 64 nodes on a handle with the dense pair store: a mass event makes EVERY node a receiver of
  // dozens of messages, config #4's share of one GPU is 1 024-2 048 node blocks, and one wave per block left three quarters of the
  // device's wave slots empty while every wave walked its block's list in four passes (round 5, late).  Four tiles then share a node
  // block's carry area (reserved with a global atomic instead of the LDS counter) and its deadline bound; in_any is kept per 64 nodes.
  constexpr uint32_t RT = SW_RES_THREADS, TILE = MASS ? SW_RES_MASS_TILE : SW_RTILE * SW_BLOCK, SUBS = TILE / RT, WPB = SW_RES_WAVES;
  static_assert(SW_RES_THREADS == 64 && SW_RTILE == 1 && TILE % RT == 0 && TILE <= SW_BLOCK && SW_BLOCK % TILE == 0, "k_resolve's tile");
  __shared__ uint32_t s_carry[1], s_dl[1], s_wcnt[SUBS * WPB > 16 ? SUBS * WPB : 16];
  __shared__ uint32_t s_list[TILE];              // the tile's receivers: count << 10 | offset in the tile
  __shared__ uint4 s_in[4][RT];                  // the lanes' 64-byte inbox lines (LDS, not registers: occupancy)
  const uint32_t tl = D.rs_order ? D.rs_order[(size_t)(*D.tick % D.P) * D.rs_T + blockIdx.x] : blockIdx.x;   // heavy tiles first
  const size_t NL = (size_t)D.R * D.nloc, l0 = (size_t)tl * TILE;
  const uint32_t nb0 = (uint32_t)(l0 / SW_BLOCK);                 // the node block the tile lies in
#ifdef SWIMSIM_WAVECLK
  unsigned long long wclk[4]; WCLK(0);
#endif
#ifdef SWIMSIM_DIAG
  unsigned long long rclk_t = __builtin_amdgcn_s_memtime(); const unsigned long long rclk_t0 = rclk_t; uint32_t rclk_it = 0, rclk_acc[6] = { 0, 0, 0, 0, 0, 0 };
#endif
  if (D.fast_blocks) {                     // nothing reached this tile: a few words and out
    uint32_t any = 0;
#pragma unroll
    for (uint32_t sb = 0; sb < SUBS; sb++) any |= l0 + sb * RT < NL ? D.in_any[l0 / 64 + sb] : 0u;
    if (!any) return;
  }
  if (threadIdx.x == 0) { s_carry[0] = 0; s_dl[0] = NONE; }
  if (threadIdx.x < SW_CEN_LDS * 5) g_s_cen[threadIdx.x] = 0;
  if (threadIdx.x == 0) g_s_cen_r = div_nloc(D, l0 < NL ? l0 : NL - 1);
  BlockStats S; S.init(lds_stats);
  // ---- the tile's receivers, compacted in ascending node order
  uint32_t cnts[SUBS];                             // (sub-pass sb covers the RT nodes from offset sb * RT of the tile)
#pragma unroll
  for (uint32_t sb = 0; sb < SUBS; sb++) {
    const size_t l = l0 + sb * RT + threadIdx.x;
    cnts[sb] = (l < NL && (!D.fast_blocks || D.in_any[l0 / 64 + sb])) ? D.in_cnt[l] : 0u;
    const uint64_t m = __ballot(cnts[sb] != 0);
    if (sw_lane() == 0) s_wcnt[sb * WPB + threadIdx.x / 64] = (uint32_t)__popcll(m);
  }
  __syncthreads();
  uint32_t n_act = 0;
#pragma unroll
  for (uint32_t sb = 0; sb < SUBS; sb++) {
    uint32_t base = 0;
    for (uint32_t j = 0; j < SUBS * WPB; j++) { const uint32_t c = s_wcnt[j]; n_act += sb == 0 ? c : 0; base += j < sb * WPB + threadIdx.x / 64 ? c : 0; }
    const uint64_t m = __ballot(cnts[sb] != 0);
    if (cnts[sb]) {
      const uint32_t c = cnts[sb] > 0x3FFFFFu ? 0x3FFFFFu : cnts[sb];
      s_list[base + (uint32_t)__popcll(m & ((1ull << sw_lane()) - 1))] = (c << 10) | (sb * RT + threadIdx.x);
    }
  }
  if (D.fast_blocks && threadIdx.x < SUBS && l0 + threadIdx.x * RT < NL) D.in_any[l0 / 64 + threadIdx.x] = 0;      // (this workgroup's own 64-node groups: nobody else reads them)
  // Mass events: inboxes of very different sizes in one wave leave its lanes idle while the longest one is merged (config #4's mass phase,
  // profiles/r03_config4_resolve_phase_clock.txt: the busiest lane of a wave held 153 messages where the mean was 22 — every wave busy
  // for 8 ms of an 11.8 ms tick).  So when the tile holds an inbox of SW_ORDER_MIN messages or more the list is reordered by size class
  // (floor(log2(count)), largest first: the lanes of a wave then hold inboxes within a factor of two of each other) — a counting sort in
  // place, every lane carrying its up to four entries through the barrier.  Which lane merges which node shows in no result.
  {
    bool big = false;
#pragma unroll
    for (uint32_t sb = 0; sb < SUBS; sb++) big |= cnts[sb] >= SW_ORDER_MIN;
    if (__syncthreads_or(big)) {                    // (the barrier: the list is complete)
      uint32_t ent[SUBS], pos[SUBS];
#pragma unroll
      for (uint32_t j = 0; j < SUBS; j++) { const uint32_t i = threadIdx.x + j * RT; ent[j] = i < n_act ? s_list[i] : 0u; }
      if (threadIdx.x < 16) s_wcnt[threadIdx.x] = 0;             // (16 size classes: 2^15 messages and more share the first)
      __syncthreads();
#pragma unroll
      for (uint32_t j = 0; j < SUBS; j++) pos[j] = ent[j] ? atomicAdd(&s_wcnt[15u - min(15u, 31u - (uint32_t)__clz(ent[j] >> 10))], 1u) : 0u;
      __syncthreads();
      if (threadIdx.x == 0) { uint32_t acc = 0; for (uint32_t c = 0; c < 16; c++) { const uint32_t v = s_wcnt[c]; s_wcnt[c] = acc; acc += v; } }
      __syncthreads();
#pragma unroll
      for (uint32_t j = 0; j < SUBS; j++) if (ent[j]) s_list[s_wcnt[15u - min(15u, 31u - (uint32_t)__clz(ent[j] >> 10))] + pos[j]] = ent[j];
    }
  }
  __syncthreads();
  uint32_t c_pig = 0, c_sent01 = 0, c_sent23 = 0, c_peak = 0;
  const uint32_t t_now = *D.tick;
  RCLK_MARK(0);                                    // compaction
  WCLK(1);
  for (uint32_t a0 = 0; a0 < n_act; a0 += RT) {
    if (a0 + threadIdx.x >= n_act) continue;
    const uint32_t ent = s_list[a0 + threadIdx.x];
    uint32_t cnt = ent >> 10;
    const size_t l = l0 + (ent & 1023u);
    // the whole 64-byte line (first five messages) in one go, parked in the lane's LDS column
    // (the count lives in its own dense array: the scatter's returning atomic then works on 4 bytes per node that
    // stay cache resident instead of pulling in the node's 64-byte message line)
    const uint4* row4 = (const uint4*)(D.inbox1 + l * 16);
    s_in[0][threadIdx.x] = row4[0]; s_in[1][threadIdx.x] = row4[1]; s_in[2][threadIdx.x] = row4[2]; s_in[3][threadIdx.x] = row4[3];
    const uint4 hdr0 = HDR(l);
    const uint4 vm0 = VMETA(l);
    D.in_cnt[l] = 0;
#define IN_WORD(w) (((const uint32_t*)&s_in[(w) >> 2][threadIdx.x])[(w) & 3u])
    c_peak = cnt > c_peak ? cnt : c_peak;
    if (cnt > D.C) { S.add(ST_INBOX_OVF, cnt - D.C); atomicOr(D.err, SW_ERR_INBOX_OVF); cnt = D.C; }
    const uint32_t* row2 = inbox_row(D, l);
    if (D.PB && cnt > D.C1) D.big_row[l] = NONE;      // (pooled rows: the big row is read below and free again next tick)
    NodeCtxT<RESOLVE_LQ, MASS, SERF, DYN> n(D, S);
    n.r = div_nloc(D, l); n.k = mod_nloc(D, l); n.o = D.i0 + n.k; n.t = t_now; n.l = l; n.NL = NL;
    n.load(hdr0); n.vm = vm0; n.vm_have = true;
    RCLK_MARK(1);                                  // line + header + vmeta
    // second round trip: the queue (into LDS) and, in the same breath, the view of the subject the first message of the line
    // is about (nearly always the only subject in the inbox; a wrong guess costs one wasted lookup)
    if constexpr (RESOLVE_LQ) n.stage_queue();
    const bool sorted = cnt >= SW_INBOX_SORT_MIN;
    const bool presorted = cnt >= SW_INBOX_SORT_MIN && D.bigsort_cap != 0;      // k_inbox_sort_med / k_inbox_sort / k_inbox_sort_huge have been here (bigsort_cap = 0: none runs)
    if (!sorted) {
      const uint32_t gx = IN_WORD(1), gty = IN_WORD(3) >> 30;
      if (gx < D.N && gty != SWIM_MSG_USER && gx != n.o) { n.cv = n.lookup(gx); n.cv_x = gx; }
    }
    bool have_last = false; uint64_t lhi = 0, llo = 0;
    uint32_t next_j = 0;
    RCLK_MARK(2);                                  // queue staged, first view fetched
    if (sorted && !presorted) {                    // the five messages of the line join the row (it has room for all C), then one sort
      uint32_t* row = (uint32_t*)row2;
      for (uint32_t j = 0; j < SW_INBOX_FAST; j++) { uint32_t w = 1 + 3 * j, q = cnt - SW_INBOX_FAST + j; row[3 * q] = IN_WORD(w); row[3 * q + 1] = IN_WORD(w + 1); row[3 * q + 2] = IN_WORD(w + 2); }
      inbox_heapsort(row, cnt);
    }
    for (;;) {
      bool have = false; uint64_t bhi = 0, blo = 0; uint4 best = make_uint4(0, 0, 0, 0);
      if (sorted) {
        while (next_j < cnt && !have) {            // ascending; a duplicate (same key as the one before) is applied once
          const uint32_t* m = row2 + next_j * 3; next_j++;
          best = make_uint4(0, m[0], m[1], m[2]); edge_key(best, bhi, blo);
          have = !(have_last && bhi == lhi && blo == llo);
        }
      } else
      for (uint32_t j = 0; j < cnt; j++) {
        uint4 e;                                   // {-, subject, inc, meta}
        if (j < SW_INBOX_FAST) { uint32_t w = 1 + 3 * j; e = make_uint4(0, IN_WORD(w), IN_WORD(w + 1), IN_WORD(w + 2)); }
        else { const uint32_t* m = row2 + (j - SW_INBOX_FAST) * 3; e = make_uint4(0, m[0], m[1], m[2]); }
        uint64_t hi, lo; edge_key(e, hi, lo);
        if (have_last && (hi < lhi || (hi == lhi && lo <= llo))) continue;
        if (!have || hi < bhi || (hi == bhi && lo < blo)) { have = true; bhi = hi; blo = lo; best = e; }
      }
      if (!have) break;
      uint32_t type = best.w >> 30, from = best.w & 0x3FFFFFFFu;
      if (best.y == SWIM_SUBJECT_PIGGY)
        n.piggyback(best.z, type, MASS ? (uint32_t*)&D.carry_cl[nb0] : &s_carry[0], D.carry + ((size_t)((n.t + 1) & 1u) * D.NB + nb0) * D.carry_cap, lds_emeta);   // (MASS: four tiles share the block's area)
      else if (best.y == SWIM_SUBJECT_PULL && type == SWIM_MSG_ALIVE) {     // push-pull request: answer next tick
        uint32_t li = (n.t + 1) & 1u, sub = nb0 % SW_PP_LISTS, sub_cap = D.pp_cap / SW_PP_LISTS;
        uint32_t pos = atomicAdd(&D.pp_cnt[(li * SW_PP_LISTS + sub) * 16], 1u);
        if (pos < sub_cap) D.pp_list[((size_t)li * SW_PP_LISTS + sub) * sub_cap + pos] = make_uint2((uint32_t)l, best.z | ((from & 1u) << 31));   // (bit 31: a join's request)
        else atomicOr(D.err, SW_ERR_PEND_OVF);
      }
      else if (type != SWIM_MSG_USER) {
        // a membership rumour: ONE view lookup whatever its type — as three calls the lanes of a wave that hold an alive, a suspect and a
        // dead message ran three copies of the lookup (node word + row / home slot, then the pair / entry) one after the other; in config
        // #4's mass phase, where the types mix, that was most of a message iteration's round trips
        const uint32_t x = best.y, inc = best.z;
        auto v = n.take_view(x);
        if (type == SWIM_MSG_ALIVE) { if (x != n.o) n.alive_other(v, x, inc, from); else if (!n.leaving && inc > n.self_inc) n.refute(v, inc); }
        else if (type == SWIM_MSG_SUSPECT) n.suspect_v(v, x, inc, from);
        else n.dead_v(v, x, inc, from);
        n.cv = v;
      }
      else n.user_event(best.y, best.z);
      have_last = true; lhi = bhi; llo = blo;
#ifdef SWIMSIM_DIAG
      rclk_it++;
#endif
    }
    RCLK_MARK(3);                                  // the messages
    n.store();
    // suspicion timers armed here: the block's deadline bound is lowered once per workgroup (every lane of a cluster arms
    // one within a few ticks of a failure)
    if (n.dl_new != NONE) atomicMin(&s_dl[0], n.dl_new);
    c_pig += n.c_pig; c_sent01 += n.c_sent01; c_sent23 += n.c_sent23;
    q_bit_lane(D, l, n.q_became_set(), n.q_became_clr());
    RCLK_MARK(4);                                  // write-back
  }
#ifdef SWIMSIM_DIAG
  { uint32_t m = rclk_it; for (int off = 32; off; off >>= 1) { uint32_t v = __shfl_xor(m, off); m = v > m ? v : m; } rclk_it = m; }
#endif
  WCLK(2);
  if (__any(c_peak > 5u)) {                        // swim_stats_t.inbox_peak (the 64-byte line holds five: smaller inboxes never raise it past 5)
    for (int off = 32; off; off >>= 1) { const uint32_t v = __shfl_xor(c_peak, off); c_peak = v > c_peak ? v : c_peak; }
    if (sw_lane() == 0 && c_peak > *D.peak) atomicMax(D.peak, c_peak);
  }
  if (D.flags & SWIM_F_PIGGYBACK) {
    uint32_t s0 = c_sent01 & 0xFFFFu, s1 = c_sent01 >> 16, s2 = c_sent23 & 0xFFFFu, s3 = c_sent23 >> 16;
    S.wave_add(ST_PIGGY, c_pig); S.wave_add(ST_PIGGY_MSGS, s0 + s1 + s2 + s3);
    S.wave_add(ST_SENT0, s0); S.wave_add(ST_SENT1, s1); S.wave_add(ST_SENT2, s2); S.wave_add(ST_SENT3, s3);
  }
  S.flush(D);                                      // (barrier inside: every lane's carry reservations, deadlines and census deltas are in)
  if (threadIdx.x < SW_CEN_LDS * 5 && g_s_cen[threadIdx.x] && threadIdx.x / 5 < D.S)
    atomicAdd(&D.cen_dl[((size_t)g_s_cen_r * D.S + threadIdx.x / 5) * 8 + threadIdx.x % 5], (uint32_t)g_s_cen[threadIdx.x]);
  if (MASS) { if (__any(c_pig != 0) && threadIdx.x == 0) *D.carry_stamp = t_now + 1; }        // (the block's count word was the reservation counter itself)
  else if (threadIdx.x == 0 && s_carry[0]) { D.carry_cl[nb0].x = s_carry[0]; *D.carry_stamp = t_now + 1; }
  if (threadIdx.x == 0 && s_dl[0] != NONE) atomicMin(&D.dl_blk[nb0], s_dl[0]);       // (no look first: a load the workgroup would have to wait for on its way out)
#ifdef SWIMSIM_WAVECLK
  WCLK(3);
  if (sw_lane() == 0) {
    unsigned long long* row = g_wclk[(blockIdx.x * WPB + threadIdx.x / 64) % WCLK_ROWS];
    row[0] = wclk[0]; row[1] = wclk[1]; row[2] = wclk[2]; row[3] = wclk[3]; row[4] = n_act; row[5] = blockIdx.x;
  }
#endif
#ifdef SWIMSIM_DIAG
  RCLK_MARK(5);                                    // tallies, flush
  if (sw_lane() == 0) {
    uint32_t* row = g_rclk[(blockIdx.x * WPB + threadIdx.x / 64) % RCLK_ROWS];
    for (int p = 0; p < 6; p++) row[p] = rclk_acc[p];
    row[6] = (uint32_t)(__builtin_amdgcn_s_memtime() - rclk_t0); row[7] = ((n_act + RT - 1) / RT) << 16 | (rclk_it & 0xFFFFu);
  }
#endif
}

// =================================================================================================
// k_census / k_finish — observation: how the live observers of a replica see each dirty subject
// =================================================================================================
// one block's share of the recount of dirty slot sidx (the caller has checked that it is one)
__device__ __forceinline__ void census_recount(DevRef D, uint32_t sidx, uint32_t* acc) {
  const uint32_t r = sidx / D.S;
  if (threadIdx.x < CEN_WORDS) acc[threadIdx.x] = 0;
  __syncthreads();
  uint32_t x = D.subj_node[sidx], maxinc = D.slot_maxinc[sidx];
  uint32_t obs = 0, st[4] = { 0, 0, 0, 0 }, cur = 0;
  const uint32_t* nw = D.nw + (size_t)r * D.N;
  const uint32_t wx = nw[x], bkey = base_key_of(D, r, x, wx), mrow_x = (wx & NW_MASS) ? D.mrow[(size_t)r * D.N + x] : 0;
  for (uint32_t k = blockIdx.x * SW_BLOCK + threadIdx.x; k < D.nloc; k += gridDim.x * SW_BLOCK) {
    uint32_t o = D.i0 + k;
    if (o == x || (nw[o] & NW_DEAD)) continue;
    uint32_t key = bkey;
    if (wx & NW_MASS) { const uint32_t a = D.mA[m_idx(D, r, mrow_x, k)]; if (a) key = MA_KEY(a); }
    else if (wx & NW_SUBJECT) { uint4 e; uint32_t fs; if (vt_find(D, (size_t)r * D.nloc + k, x, e, fs) != NONE) key = e.y; }
    const uint32_t s = SW_KST(key);
    obs++; st[0] += s == 0; st[1] += s == 1; st[2] += s == 2; st[3] += s == 3;
    cur += SW_KINC(key) == maxinc;
  }
  uint32_t vals[6] = { obs, st[0], st[1], st[2], st[3], cur };
#pragma unroll
  for (int j = 0; j < 6; j++) {
    uint32_t v = vals[j];
    for (int off = 32; off; off >>= 1) v += __shfl_down(v, off);
    if (sw_lane() == 0 && v) atomicAdd(&acc[j], v);
  }
  __syncthreads();
  if (threadIdx.x < 6) { if (acc[threadIdx.x]) atomicAdd(&D.cen_acc[(size_t)sidx * CEN_WORDS + threadIdx.x], acc[threadIdx.x]); }
}
__global__ void __launch_bounds__(SW_BLOCK) k_census(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  uint32_t sidx = blockIdx.y, r = sidx / D.S, sl = sidx % D.S;
  if (sl >= D.n_slots[r] || !D.slot_dirty[sidx]) return;
  __shared__ uint32_t acc[CEN_WORDS];
  census_recount(D, sidx, acc);
}

// first-suspect / first-dead / all-dead / all-current stamps of a slot whose cached census just changed
__device__ void census_stamps(DevRef D, uint32_t sidx, uint32_t now) {
  swim_census* c = &D.census[sidx];
  if (c->first_suspect_ms == NONE && c->by_state[1]) c->first_suspect_ms = now;
  if (c->first_dead_ms == NONE && (c->by_state[2] || c->by_state[3])) c->first_dead_ms = now;
  if (c->all_dead_ms == NONE && c->n_observers && c->by_state[2] + c->by_state[3] == c->n_observers) c->all_dead_ms = now;
  if (c->all_current_ms == NONE && c->n_observers && D.slot_maxinc[sidx] > 1 && c->n_current == c->n_observers) c->all_current_ms = now;
}
// fold the accumulators of a dirty slot into its cached census and stamp the first-times
__device__ void census_commit(DevRef D, uint32_t sidx, uint32_t now) {
  swim_census* c = &D.census[sidx];
  uint32_t* a = &D.cen_acc[(size_t)sidx * CEN_WORDS];
  c->n_observers = a[CEN_OBS]; c->by_state[0] = a[CEN_ST0]; c->by_state[1] = a[CEN_ST1];
  c->by_state[2] = a[CEN_ST2]; c->by_state[3] = a[CEN_ST3]; c->n_current = a[CEN_CUR];
  for (int j = 0; j < CEN_WORDS; j++) a[j] = 0;
  for (int j = 0; j < 5; j++) D.cen_dl[(size_t)sidx * 8 + j] = 0;        // (a recount is of the state AFTER this tick's changes: their deltas are in it)
  census_stamps(D, sidx, now);
  D.slot_dirty[sidx] = 0;
}
// this tick's view changes of running observers, tallied by k_resolve: add them to the cached census (no recount)
__device__ void census_apply_deltas(DevRef D, uint32_t sidx, uint32_t now) {
  uint32_t* d = &D.cen_dl[(size_t)sidx * 8];
  if (!(d[0] | d[1] | d[2] | d[3] | d[4])) return;
  swim_census* c = &D.census[sidx];
  for (int j = 0; j < 4; j++) c->by_state[j] += d[j];
  c->n_current += d[4];
  for (int j = 0; j < 5; j++) d[j] = 0;
  census_stamps(D, sidx, now);
}

// collect the ids of replica r whose node word is non-zero (whole block cooperates)
__device__ void rebuild_exceptions(DevRef D, uint32_t r, uint32_t* s_n) {
  if (threadIdx.x == 0) *s_n = 0;
  __syncthreads();
  const uint32_t* nw = D.nw + (size_t)r * D.N;
  for (uint32_t x = threadIdx.x; x < D.N; x += blockDim.x) {
    if (*(volatile uint32_t*)s_n > SW_EXC_MAX) break;          // more than the list holds: unusable, no need to finish the scan
    if (nw[x]) { uint32_t pos = atomicAdd(s_n, 1u); if (pos < SW_EXC_MAX) D.exc_ent[(size_t)r * SW_EXC_MAX + pos] = make_uint2(x, nw[x]); }
  }
  __syncthreads();
  if (threadIdx.x == 0) { D.exc_cnt[r] = *s_n; D.exc_dirty[r] = 0; }
  __syncthreads();
}
__global__ void __launch_bounds__(SW_BLOCK) k_exc_rebuild(const SwDev* __restrict__ Dp, uint32_t r) {
  SW_DEV_BIND
  __shared__ uint32_t s_n;
  rebuild_exceptions(D, r, &s_n);
}
// after a fold tick: node words changed (subject bits fell) if anything was folded
__global__ void __launch_bounds__(SW_BLOCK) k_exc_rebuild_folded(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  __shared__ uint32_t s_n;
  if (!*D.fold_any) return;
  rebuild_exceptions(D, blockIdx.x, &s_n);
}

__device__ __forceinline__ void finish_tick(DevRef D, uint32_t* last_cnt) {
  uint32_t t = *D.tick, now = now_ms(D, t);
  for (uint32_t sidx = threadIdx.x; sidx < D.R * D.S; sidx += SW_BLOCK) {
    uint32_t r = sidx / D.S, sl = sidx % D.S;
    if (sl >= D.n_slots[r]) continue;
    if (D.slot_dirty[sidx]) census_commit(D, sidx, now); else census_apply_deltas(D, sidx, now);
    if (D.trace && t < D.trace_ticks) {
      const swim_census* c = &D.census[sidx];
      uint32_t* row = &D.trace[((size_t)sidx * D.trace_ticks + t) * 5];
      row[0] = c->by_state[0]; row[1] = c->by_state[1]; row[2] = c->by_state[2]; row[3] = c->by_state[3]; row[4] = c->n_current;
    }
  }
  // sharded runs: tell the other shards (through the count exchange) whether any queue here may be non-empty;
  // while nobody has anything, probes need not file piggy-back orders for nodes of other shards
  if (D.n_shards > 1 && (D.flags & SWIM_F_PIGGYBACK)) {
    uint32_t any = 0;
    if (D.fast_blocks) { for (uint32_t bb = threadIdx.x; bb < D.NB; bb += SW_BLOCK) any |= D.q_any[bb]; } else any = 1;
    any = __syncthreads_or(any != 0);
    if (threadIdx.x == 0) *D.act = any ? 1u : 0u;
  }
  // swim_inject_join: a node whose join push-pull went out this tick stops being alone — from the next tick on it probes
  // and gossips like everybody.  Ground truth is replicated, so every shard does this for every joiner.
  if (D.join_cnt && *D.join_cnt) {
    const uint32_t n = *D.join_cnt < D.join_cap ? *D.join_cnt : D.join_cap;
    for (uint32_t e = threadIdx.x; e < n; e += SW_BLOCK) {
      const uint2 j = D.join_list[e];
      const uint32_t r = j.x / D.N, o = j.x % D.N, p = j.y, wo = D.nw[j.x], wp = D.nw[(size_t)r * D.N + p];
      if ((wo & (NW_DEAD | NW_ATTACHED)) || !(wo & NW_ALONE)) continue;
      if (!(p != o && !(wp & NW_DEAD) && NW_PART(wo) == NW_PART(wp))) continue;
      atomicAnd(&D.nw[j.x], ~NW_ALONE); atomicAdd(&D.acting[r], 1u);
      if (o >= D.i0 && o < D.i0 + D.nloc) atomicAdd(&D.alive_cnt[((size_t)r * D.nloc + (o - D.i0)) / SW_BLOCK], 1u);
      D.exc_dirty[r] = 1;
      for (uint32_t sl = 0; sl < D.n_slots[r]; sl++) D.slot_dirty[(size_t)r * D.S + sl] = 1;
    }
    __syncthreads();
    __shared__ uint32_t s_exc_n;
    for (uint32_t r = 0; r < D.R; r++) if (D.exc_dirty[r]) rebuild_exceptions(D, r, &s_exc_n);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    *D.tick = t + 1;
    for (uint32_t sh = 0; sh < D.n_shards; sh++) { last_cnt[sh] = D.out_cnt[sh]; D.out_cnt[sh] = 0; }
    D.pend_cnt[(t + 1) % (D.TQ + 1)] = 0;      // the list the next tick appends to (just consumed)
    if (D.join_cnt) *D.join_cnt = 0;           // the joins of this tick are under way
    if (D.c_cnt) *D.c_cnt = 0;                 // this tick's coordinate updates are committed
    if (D.m_due_cnt) *D.m_due_cnt = 0;         // the dense store's due rows were looked at
    if (D.xs_cnt) *D.xs_cnt = 0;               // ... and its state exchanges sent
    if (D.PB) { *D.big_n = 0; *D.defer_n = 0; }   // pooled inbox rows: all handed back (k_resolve cleared the nodes' row words)
  }
  if (threadIdx.x < SW_PP_LISTS) D.pp_cnt[((t & 1u) * SW_PP_LISTS + threadIdx.x) * 16] = 0;        // answered
}
__global__ void __launch_bounds__(SW_BLOCK) k_finish(const SwDev* __restrict__ Dp, uint32_t* last_cnt) {
  SW_DEV_BIND
  finish_tick(D, last_cnt);
}
// k_census and k_finish in ONE launch (round 5: a quiet tick is five launches of ~5 us each, and nothing in the last two depends on a grid).
// Same grid as k_census.  The blocks of dirty slots recount; whoever of them arrives last at the ticket — or block (0, 0) alone when no
// slot is dirty, every tick but a handful — runs the tick's epilogue.  Every participant derives the number of arrivals from the same
// words (slot_dirty is only cleared by the epilogue itself), so no block waits for another: the last one in simply stays.
__global__ void __launch_bounds__(SW_BLOCK) k_census_finish(const SwDev* __restrict__ Dp, uint32_t* last_cnt) {
  SW_DEV_BIND
  const uint32_t sidx = blockIdx.y, r = sidx / D.S, sl = sidx % D.S;
  const bool dirty = sl < D.n_slots[r] && D.slot_dirty[sidx] != 0;
  const bool first = blockIdx.x == 0 && blockIdx.y == 0;
  // A tick in which a join completes: the epilogue marks EVERY slot of the replica dirty (finish_tick), and a block of this launch that has
  // not been scheduled yet would take the mark for this tick's — recount a second time, take a ticket the epilogue has already reset
  // (ADVICE r5).  So in such a tick every block of the grid takes a ticket and the last one of ALL runs the epilogue: the marks are written
  // after every block has read its flags.  join_cnt is the same for the whole launch (set between ticks, cleared by the epilogue alone).
  const bool joins = D.join_cnt && *D.join_cnt != 0;
  if (!dirty && !first && !joins) return;
  __shared__ uint32_t acc[CEN_WORDS];
  __shared__ uint32_t s_last;
  __shared__ uint32_t s_dirty;
  if (threadIdx.x == 0) s_dirty = 0;
  __syncthreads();
  {
    uint32_t c = 0;
    for (uint32_t i = threadIdx.x; i < D.R * D.S; i += SW_BLOCK) c += (i % D.S < D.n_slots[i / D.S] && D.slot_dirty[i]) ? 1u : 0u;
    if (c) atomicAdd(&s_dirty, c);
  }
  __syncthreads();
  const bool first_dirty = 0 < D.n_slots[0] && D.slot_dirty[0] != 0;                 // (block (0, 0) is one of a dirty slot's blocks then)
  const uint32_t arrivals = joins ? gridDim.x * gridDim.y : s_dirty * gridDim.x + (first_dirty ? 0u : 1u);
  if (dirty) census_recount(D, sidx, acc);                                           // (ends behind a barrier: the block's atomics are issued)
  if (threadIdx.x == 0) {
    __threadfence();                                                                 // the recount's sums before the ticket
    s_last = atomicAdd(D.cf_ticket, 1u) + 1u == arrivals ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();                                                                   // ... and every other block's, after it
  if (threadIdx.x == 0) *D.cf_ticket = 0;
  finish_tick(D, last_cnt);
}
__global__ void k_census_commit(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  uint32_t now = now_ms(D, *D.tick);
  for (uint32_t sidx = threadIdx.x; sidx < D.R * D.S; sidx += blockDim.x) {
    uint32_t r = sidx / D.S, sl = sidx % D.S;
    if (sl < D.n_slots[r] && D.slot_dirty[sidx]) census_commit(D, sidx, now);
  }
}
__global__ void __launch_bounds__(SW_BLOCK) k_count_live(const SwDev* __restrict__ Dp, uint32_t r, uint32_t x, uint32_t* out) {
  SW_DEV_BIND
  uint32_t c = 0;
  const uint32_t* nw = D.nw + (size_t)r * D.N;
  for (uint32_t k = blockIdx.x * SW_BLOCK + threadIdx.x; k < D.nloc; k += gridDim.x * SW_BLOCK)
    c += (D.i0 + k != x) && !(nw[D.i0 + k] & NW_DEAD);
  for (int off = 32; off; off >>= 1) c += __shfl_down(c, off);
  if (sw_lane() == 0 && c) atomicAdd(out, c);
}

// =================================================================================================
// k_quiet — K ticks of a PRISTINE cluster in one launch.  A population in which nothing has ever happened (no stimulus
// of any kind since swim_create, no packet loss, everybody a member from the start) cannot leave that state on its own:
// every probe is acked in its own tick, no rumour exists, no timer runs, a push-pull moves a pull request and an empty
// answer.  All that K such ticks change is each node's position in its probe order — and the counters.  (swim_step uses
// this for all but the last tick of a call; the last one runs the five kernels, so every transient is where it would be.)
// =================================================================================================
__global__ void __launch_bounds__(SW_BLOCK) k_quiet(const SwDev* __restrict__ Dp, uint32_t K) {
  SW_DEV_BIND
  const size_t NL = (size_t)D.R * D.nloc, l = (size_t)blockIdx.x * SW_BLOCK + threadIdx.x;
  const uint32_t t = *D.tick;
  uint32_t c_probe = 0, c_quiet = 0, c_pp = 0;
  if (l < NL) {
    const uint32_t r = (uint32_t)(l / D.nloc), i = D.i0 + (uint32_t)(l % D.nloc), ch = i / D.CH;
    const uint32_t gph = ch % D.G, pph = (ch / D.G) % D.P;
    // gossip(): "no broadcasts" in every tick the node is due
    { const uint32_t first = (gph + D.G - t % D.G) % D.G; if (first < K) c_quiet = (K - first + D.G - 1) / D.G; }
    // probe(): walk the shuffled list (self is skipped, a wrap reshuffles); every ping is acked (awareness stays 0)
    uint2 h = D.ph[l];
    uint32_t cursor = h.x, epoch = p_epoch(h.y);
    for (uint32_t tt = (pph + D.P - t % D.P) % D.P; tt < K; tt += D.P) {
      uint32_t num_check = 0;
      while (num_check < D.N) {
        if (cursor >= D.N) { epoch = (epoch + 1) & 0xFFFFu; cursor = 0; num_check++; continue; }
        const uint32_t c = sw_probe_perm(seed_of(D, r), D.N, i, epoch, cursor++);
        if (c == i) { num_check++; continue; }
        c_probe++; break;
      }
    }
    if (cursor != h.x || epoch != p_epoch(h.y)) D.ph[l] = make_uint2(cursor, p_pack(epoch, p_aw(h.y), 0, 0));
    // pushPull: node i is due in tick (i mod period), grouped to the probe-interval boundary before it: a pull request goes
    // out (one edge), the peer's answer is empty
    if (D.pp_period) {
      const uint32_t per = D.pp_period, grp = D.P < per ? D.P : per;
      for (uint32_t tb = (t + D.P - 1) / D.P * D.P; tb < t + K; tb += D.P)
        c_pp += ((i % per) + per - (tb % per)) % per < grp;
    }
  }
  uint32_t v0 = c_probe, v1 = c_quiet, v2 = c_pp;
  for (int off = 32; off; off >>= 1) { v0 += __shfl_down(v0, off); v1 += __shfl_down(v1, off); v2 += __shfl_down(v2, off); }
  if (sw_lane() == 0) {
    if (v0) { atomicAdd(stat_ptr(D, ST_PROBES), (unsigned long long)v0); atomicAdd(stat_ptr(D, ST_ACKS), (unsigned long long)v0); }
    if (v1) atomicAdd(stat_ptr(D, ST_QUIESCENT), (unsigned long long)v1);
    if (v2) { atomicAdd(stat_ptr(D, ST_PUSHPULLS), (unsigned long long)v2); atomicAdd(stat_ptr(D, ST_EDGES), (unsigned long long)v2); }
  }
}
__global__ void k_quiet_advance(const SwDev* __restrict__ Dp, uint32_t K) {
  SW_DEV_BIND
  if (threadIdx.x == 0 && blockIdx.x == 0) *D.tick += K;
}

// =================================================================================================
// initialisation, stimulus, digest
// =================================================================================================
__global__ void k_init_nodes(const SwDev* __restrict__ Dp, uint32_t n_initial) {
  SW_DEV_BIND
  size_t NL = (size_t)D.R * D.nloc, l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= NL) return;
  HDR(l) = make_uint4(1, 0, 0, (D.flags & SWIM_F_SERF_EVENTS) ? 1u : 0u);   // serf.Create: eventClock.Increment()
  if (D.vnk) D.vnk[l] = 0;
  D.ph[l] = make_uint2(0, 0);
  D.pr0[l] = make_uint4(NONE, 0, 0, 0);
  D.in_cnt[l] = 0; VMETA(l) = make_uint4(0, 0, NONE, NONE);
  if (D.mcnt) D.mcnt[l] = 0;
  if (D.evseq) D.evseq[l] = 0;
  if (l % 64 == 0) D.in_any[l / 64] = 0;             // (a hint per 64 nodes: every group's, not every fourth — ADVICE r5)
  if (l % SW_BLOCK == 0) {
    size_t rem = NL - l;
    uint32_t in_blk = rem < SW_BLOCK ? (uint32_t)rem : SW_BLOCK, started = 0;       // lanes of this block that run at t = 0
    for (uint32_t j = 0; j < in_blk; j++) started += D.i0 + (uint32_t)((l + j) % D.nloc) < n_initial;
    D.q_any[l / SW_BLOCK] = 0; D.alive_cnt[l / SW_BLOCK] = started;
    D.dl_blk[l / SW_BLOCK] = NONE;
  }
  if (l < D.R) { D.acting[l] = n_initial; D.base_known[l] = n_initial; }
}
__global__ void k_init_base(const SwDev* __restrict__ Dp, uint32_t n_initial) {
  SW_DEV_BIND
  size_t n = (size_t)D.R * D.N, i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool member = (uint32_t)(i % D.N) < n_initial;      // the rest has not been started: nobody has heard of it
  D.bk[i] = member ? SW_BASE_KEY : SW_KEY(0, SWIM_STATE_DEAD);
  D.nw[i] = member ? 0u : (NW_DEAD | NW_BASEMOD);
}
__global__ void k_init_slots(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= D.R * D.S) return;
  D.subj_node[i] = NONE; D.slot_dirty[i] = 0; D.slot_maxinc[i] = 1;
  for (int j = 0; j < CEN_WORDS; j++) D.cen_acc[(size_t)i * CEN_WORDS + j] = 0;
  swim_census c; memset(&c, 0, sizeof c);
  c.first_suspect_ms = c.first_dead_ms = c.all_dead_ms = c.all_current_ms = NONE;
  D.census[i] = c;
}

enum { INJ_KILL = 0, INJ_REVIVE = 1, INJ_LEAVE = 2, INJ_UPDATE = 3 };

// watch slots for the nodes named in a stimulus call (while slots remain); fresh[0] = how many were new, fresh[1..] = which
__global__ void k_inject_alloc(const SwDev* __restrict__ Dp, uint32_t r, const uint32_t* ids, uint32_t n, uint32_t* fresh) {
  SW_DEV_BIND
  __shared__ uint32_t s_stop;
  if (threadIdx.x == 0) {
    fresh[0] = 0;
    uint32_t a = 0;
    for (; a < n && D.n_slots[r] < D.S; a++) alloc_slot(D, r, ids[a], fresh);
    s_stop = a;
  }
  __syncthreads();
  // out of slots: the rest is only counted (subject_overflow), in parallel — a partition names 10^5 nodes
  uint32_t c = 0;
  for (uint32_t a = s_stop + threadIdx.x; a < n; a += blockDim.x) c += !NW_HAS_SLOT(D.nw[(size_t)r * D.N + ids[a]]);
  for (int off = 32; off; off >>= 1) c += __shfl_down(c, off);
  if (sw_lane() == 0 && c) atomicAdd(stat_ptr(D, ST_SUBJ_OVF), (unsigned long long)c);
}
__global__ void __launch_bounds__(SW_BLOCK) k_inject(const SwDev* __restrict__ Dp, int op, uint32_t r, const uint32_t* ids, uint32_t n) {
  SW_DEV_BIND
  __shared__ uint32_t lds_stats[ST_COUNT];
  BlockStats S; S.init(lds_stats);
  uint32_t a = blockIdx.x * SW_BLOCK + threadIdx.x;
  if (a < n) {
    uint32_t x = ids[a]; size_t g = (size_t)r * D.N + x;
    bool local = x >= D.i0 && x < D.i0 + D.nloc;
    size_t l = (size_t)r * D.nloc + (x - D.i0);
    if (op == INJ_KILL) {
      uint32_t old = atomicOr(&D.nw[g], NW_DEAD);
      if (!(old & NW_INERT)) {                               // it was being acted for
        atomicSub(&D.acting[r], 1u);
        if (local) atomicSub(&D.alive_cnt[l / SW_BLOCK], 1u);
      }
    } else if (op == INJ_REVIVE) {
      uint32_t old = atomicAnd(&D.nw[g], ~NW_DEAD);
      if ((old & NW_DEAD) && !(old & (NW_ATTACHED | NW_ALONE))) {
        atomicAdd(&D.acting[r], 1u);
        if (local) atomicAdd(&D.alive_cnt[l / SW_BLOCK], 1u);
      }
      if (local) {
        uint2 h = D.ph[l];
        D.pr0[l].x = NONE; D.ph[l].y = p_pack(p_epoch(h.y), p_aw(h.y), 0, 0); D.in_cnt[l] = 0;
        // a node that comes back resumes its old views, whose suspicion timers may be long overdue: its bound
        // counts again in the block's gate
        const uint32_t d = VMETA(l).z;
        if (d != NONE) atomicMin(&D.dl_blk[l / SW_BLOCK], d);
        // ... and so do its suspicions in the dense store: its 256-observer tile is marked, and k_mass_rearm (right after this kernel) lowers
        // that tile's bound in every row to "now" — the next k_expire_mass looks into those tiles and leaves exact bounds behind.  (Rounds 3-4
        // walked the observer's column here, one lane through all M rows: 60 ms per call of config #5's leg, half its kernel time — profiles/r05_config5_kernel_stats.csv)
        if (D.M && D.mcnt[l]) atomicOr(&D.m_rev[((x - D.i0) / SW_BLOCK) >> 5], 1u << (((x - D.i0) / SW_BLOCK) & 31u));
      }
    } else if (local && !(D.nw[g] & NW_DEAD)) {
      NodeCtx c(D, S);
      c.r = r; c.o = x; c.k = x - D.i0; c.t = *D.tick; c.l = l; c.NL = (size_t)D.R * D.nloc;
      c.load();
      if (op == INJ_LEAVE) { c.leaving = 1; c.dead_node(x, c.self_inc, x); }   // memberlist.Leave
      else {                                                                    // memberlist.UpdateNode
        c.self_inc++;
        NodeCtx::View me = c.lookup(x);
        if (c.make(me, x)) { c.set_view(me, c.self_inc, SWIM_STATE_ALIVE, false); c.put(me); }
        c.broadcast(x, SWIM_MSG_ALIVE, c.self_inc, 1);
      }
      c.store();
      if (c.dl_new != NONE) atomicMin(&D.dl_blk[l / SW_BLOCK], c.dl_new);
      q_bit_lane(D, l, c.q_became_set(), c.q_became_clr());
    }
  }
  if ((op == INJ_KILL || op == INJ_REVIVE) && blockIdx.x == 0)
    for (uint32_t sl = threadIdx.x; sl < D.n_slots[r]; sl += SW_BLOCK) D.slot_dirty[(size_t)r * D.S + sl] = 1;
  S.flush(D);
}
// after k_inject(INJ_REVIVE) on a handle with the dense pair store: the tiles that hold a revived observer are due in every live row of replica r
__global__ void __launch_bounds__(SW_BLOCK) k_mass_rearm(const SwDev* __restrict__ Dp, uint32_t r) {
  SW_DEV_BIND
  const uint32_t row = blockIdx.x * SW_BLOCK + threadIdx.x, now = now_ms(D, *D.tick);
  if (row >= D.M || D.mrow_subj[(size_t)r * D.M + row] == NONE) return;
  bool any = false;
  for (uint32_t w = 0; w < (D.nbl + 31) / 32; w++)
    for (uint32_t m = D.m_rev[w]; m; m &= m - 1) {
      const uint32_t tile = w * 32 + (uint32_t)__ffs(m) - 1;
      atomicMin(&D.m_tile_dl[((size_t)r * D.M + row) * D.nbl + tile], now); any = true;
    }
  if (any) atomicMin(&D.m_row_dl[(size_t)r * D.M + row], now);
}
// swim_set_tcp_class
__global__ void __launch_bounds__(SW_BLOCK) k_set_tcp_class(const SwDev* __restrict__ Dp, uint32_t r, const uint32_t* ids, uint32_t n, uint32_t cls) {
  SW_DEV_BIND
  const uint32_t a = blockIdx.x * SW_BLOCK + threadIdx.x;
  if (a < n) { uint32_t* w = &D.nw[(size_t)r * D.N + ids[a]]; atomicAnd(w, ~NW_TCP_MASK); atomicOr(w, cls << NW_TCP_SHIFT); }
}
// swim_inject_join: serf.Create + serf.Join([via]) for nodes that are not running — a fresh process
__global__ void __launch_bounds__(SW_BLOCK) k_inject_join(const SwDev* __restrict__ Dp, uint32_t r, const uint32_t* ids, uint32_t n, uint32_t via) {
  SW_DEV_BIND
  __shared__ uint32_t lds_stats[ST_COUNT];
  BlockStats S; S.init(lds_stats);
  const uint32_t a = blockIdx.x * SW_BLOCK + threadIdx.x;
  if (a < n) {
    const uint32_t x = ids[a]; const size_t g = (size_t)r * D.N + x;
    const uint32_t old = D.nw[g];
    if ((old & NW_DEAD) && !(old & NW_ATTACHED)) {          // (running already: Join on a live member is a no-op here)
      atomicAnd(&D.nw[g], ~NW_DEAD); atomicOr(&D.nw[g], NW_ALONE);   // up, but it knows nobody until the join push-pull went through
      { const uint32_t pos = atomicAdd(D.join_cnt, 1u); if (pos < D.join_cap) D.join_list[pos] = make_uint2((uint32_t)g, via); else atomicOr(D.err, SW_ERR_PEND_OVF); }
      if (x >= D.i0 && x < D.i0 + D.nloc) {
        const size_t NL = (size_t)D.R * D.nloc, l = (size_t)r * D.nloc + (x - D.i0);
        // nothing queued, no views of its own (it holds the base row), clean probe state
        for (uint32_t sl = 0; sl < D.VT; sl++) if (D.vt[(size_t)sl * NL + l].x != VT_EMPTY) D.vt[(size_t)sl * NL + l].x = VT_EMPTY;
        VMETA(l) = make_uint4(0, 0, NONE, NONE);
        if (D.M && D.mcnt[l]) { for (uint32_t row = 0; row < D.M; row++) { const size_t idx = m_idx(D, r, row, x - D.i0); if (D.mA[idx]) D.mA[idx] = 0; } D.mcnt[l] = 0; }
        if (D.iq && D.iqn[l]) { for (uint32_t row = 0; row < D.M; row++) { const size_t ei = e_idx(D, r, row, x - D.i0); if (D.mE[ei] & QE_QUEUED) D.mE[ei] = 0; } D.iqn[l] = 0; }
        const uint32_t bkey = base_key_of(D, r, x, old);
        if (D.vnk) D.vnk[l] = SW_KINC(bkey) == 0 ? 1u : 0u;  // it knows itself, whatever the base row says
        uint4 h = HDR(l);
        if (SW_KINC(bkey) != 0 || h.x > 1 || h.z) h.x++;     // a restart: past the incarnation others may remember
        h.y = 0; HDR(l) = h;
        const uint2 p = D.ph[l];
        D.ph[l].y = p_pack(p_epoch(p.y), 0, 0, 0); D.pr0[l].x = NONE; D.in_cnt[l] = 0;
        q_bit_lane(D, l, false, true);
        if (D.coord) coord_reset_lane(D, l);                  // a fresh process: a fresh coordinate client
        NodeCtx c(D, S);
        c.r = r; c.o = x; c.k = x - D.i0; c.t = *D.tick; c.l = l; c.NL = NL;
        c.load();
        c.broadcast(x, SWIM_MSG_ALIVE, c.self_inc, 0);       // memberlist setAlive
        if (D.sslt) D.sslt[l] = 0;                           // (its join intent comes with the answer to its join push-pull: role_ppreply)
        c.store();
        q_bit_lane(D, l, c.q_became_set(), c.q_became_clr());
      }
    }
  }
  if (blockIdx.x == 0) for (uint32_t sl = threadIdx.x; sl < D.n_slots[r]; sl += SW_BLOCK) D.slot_dirty[(size_t)r * D.S + sl] = 1;
  S.flush(D);
}
__global__ void k_attach(const SwDev* __restrict__ Dp, uint32_t r, uint32_t x) {
  SW_DEV_BIND
  if (threadIdx.x || blockIdx.x) return;
  size_t g = (size_t)r * D.N + x;
  uint32_t old = atomicOr(&D.nw[g], NW_ATTACHED);
  *D.att_any = 1;
  bool local = x >= D.i0 && x < D.i0 + D.nloc;
  if (local && !(old & NW_ATTACHED)) {                       // its frozen queue must not keep a gossip block busy
    size_t l = (size_t)r * D.nloc + (x - D.i0);
    uint4 h = HDR(l); h.y = h_pack(h_leaving(h.y), 0, 0); HDR(l) = h; D.in_cnt[l] = 0;
    if (D.iq && D.iqn[l]) { for (uint32_t row = 0; row < D.M; row++) { const size_t ei = e_idx(D, r, row, x - D.i0); if (D.mE[ei] & QE_QUEUED) D.mE[ei] = 0; } D.iqn[l] = 0; }   // ... nor the rumours the pair store implies
    if (!(old & NW_DEAD)) atomicSub(&D.alive_cnt[l / SW_BLOCK], 1u);          // no longer one of the nodes the simulator acts for
    q_bit_lane(D, l, false, true);
  }
  if (!(old & NW_INERT)) atomicSub(&D.acting[r], 1u);
}
__global__ void k_set_partition(const SwDev* __restrict__ Dp, uint32_t r, const uint8_t* group, uint32_t first, uint32_t n) {
  SW_DEV_BIND
  uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n) return;
  size_t g = (size_t)r * D.N + first + a;
  D.nw[g] = (D.nw[g] & ~0x7F000000u) | (((uint32_t)group[a] & 0x7Fu) << 24);
}
// serf.UserEvent at the origin: stamp, Increment, handleUserEvent locally, queue
__global__ void k_user_event(const SwDev* __restrict__ Dp, uint32_t r, uint32_t origin, uint32_t id, uint32_t* ltime_out) {
  SW_DEV_BIND
  __shared__ uint32_t lds_stats[ST_COUNT];
  BlockStats S; S.init(lds_stats);
  if (threadIdx.x == 0) {
    *ltime_out = NONE;
    bool local = origin >= D.i0 && origin < D.i0 + D.nloc;
    if (local && !(D.nw[(size_t)r * D.N + origin] & NW_DEAD)) {
      NodeCtx c(D, S);
      c.r = r; c.o = origin; c.k = origin - D.i0; c.t = *D.tick; c.l = (size_t)r * D.nloc + c.k; c.NL = (size_t)D.R * D.nloc;
      c.load();
      uint32_t lt = c.ev_clock; c.ev_clock++;
      *ltime_out = lt;
      c.user_event(id, lt, true);
      c.store();
      q_bit_lane(D, c.l, c.q_became_set(), c.q_became_clr());
    }
  }
  S.flush(D);
}
// serf.RemoveFailedNode[Prune] at the origin: stamp, Increment, handle the intent locally, queue it
__global__ void k_force_leave(const SwDev* __restrict__ Dp, uint32_t r, uint32_t origin, uint32_t id, uint32_t* ltime_out) {
  SW_DEV_BIND
  __shared__ uint32_t lds_stats[ST_COUNT];
  BlockStats S; S.init(lds_stats);
  if (threadIdx.x == 0) {
    *ltime_out = NONE;
    bool local = origin >= D.i0 && origin < D.i0 + D.nloc;
    if (local && !(D.nw[(size_t)r * D.N + origin] & NW_DEAD)) {
      NodeCtx c(D, S);
      c.r = r; c.o = origin; c.k = origin - D.i0; c.t = *D.tick; c.l = (size_t)r * D.nloc + c.k; c.NL = (size_t)D.R * D.nloc;
      c.load();
      uint32_t lt = c.ev_clock; c.ev_clock++;
      *ltime_out = lt;
      if ((id & 0x1FFFFFFFu) == origin && D.sslt) D.sslt[c.l] |= 0x80000000u;      // serf.Leave(): its own leave intent — from now on it does not refute one
      c.user_event(id, lt, true);
      c.store();
      q_bit_lane(D, c.l, c.q_became_set(), c.q_became_clr());
    }
  }
  S.flush(D);
}
// one node's self state, gathered for swim_node_info_get
__global__ void k_gather_node(const SwDev* __restrict__ Dp, uint32_t r, uint32_t i, uint32_t* out) {
  SW_DEV_BIND
  if (threadIdx.x || blockIdx.x) return;
  size_t l = (size_t)r * D.nloc + (i - D.i0), NL = (size_t)D.R * D.nloc;
  uint4 h = HDR(l), p0 = D.pr0[l]; uint2 p = D.ph[l]; uint32_t w = D.nw[(size_t)r * D.N + i];
  out[0] = h.x; out[1] = h.y; out[2] = h.z; out[3] = h.w;
  out[4] = p0.x; out[5] = p0.y; out[6] = p0.z; out[7] = p0.w; out[8] = p.x; out[9] = p.y; out[10] = w;
  for (uint32_t j = 0; j < h_qlen(h.y) && j < 32; j++) { uint4 e = QENT(j, l); out[16 + 4 * j] = e.x; out[17 + 4 * j] = e.y; out[18 + 4 * j] = e.z; out[19 + 4 * j] = e.w; }
  out[11] = 0; out[12] = 0;
  if (D.iq) {     // SWIM_F_UNBOUNDED_QUEUE: what the pair store implies — how many, and the 32 oldest of them (lowest sequence numbers), as slot entries
    uint32_t* imp = out + 16 + 4 * 32; uint32_t ni = 0, total = 0;
    for (uint32_t row = 0; row < D.M; row++) {
      const uint32_t qe = D.mE[e_idx(D, r, row, i - D.i0)];
      if (!(qe & QE_QUEUED)) continue;
      total++;
      uint32_t at = ni;                            // insertion by sequence number, ascending; the 33rd falls off the end
      while (at > 0 && (imp[4 * (at - 1) + 3] & 0x3FFFFFu) > QE_SEQ(qe)) at--;
      if (at >= 32) continue;
      for (uint32_t m = ni < 32 ? ni : 31; m > at; m--) for (int w2 = 0; w2 < 4; w2++) imp[4 * m + w2] = imp[4 * (m - 1) + w2];
      const size_t mi = m_idx(D, r, row, i - D.i0); const uint32_t a = D.mA[mi], f = D.mF[mi];
      imp[4 * at] = D.mrow_subj[(size_t)r * D.M + row]; imp[4 * at + 1] = MA_INC(a) + QF_DELTA(f); imp[4 * at + 2] = QF_FROM(f); imp[4 * at + 3] = m_pack(QE_TYPE(qe), QE_TR(qe), QE_SEQ(qe));
      if (ni < 32) ni++;
    }
    out[11] = total; out[12] = ni;
  }
}

// swim_event_queued: is {id, ltime} in the node's serf queue?
__global__ void k_evq_find(const SwDev* __restrict__ Dp, uint32_t r, uint32_t i, uint32_t id, uint32_t ltime, uint32_t* out) {
  SW_DEV_BIND
  if (threadIdx.x || blockIdx.x) return;
  const size_t l = (size_t)r * D.nloc + (i - D.i0), NL = (size_t)D.R * D.nloc;
  uint32_t hit = 0;
  if (D.EQ) for (uint32_t j = 0; j < h_evqlen(HDR(l).y); j++) { const uint4 e = D.evq[(size_t)j * NL + l]; hit |= e.x == id && e.y == ltime; }
  out[0] = hit;
}
// order-independent digest (same item hashes as the oracle; see swim_state_digest there)
__device__ __forceinline__ void digest_commit(uint64_t d, unsigned long long* out) {
  for (int off = 32; off; off >>= 1) d += __shfl_down(d, off);
  if (sw_lane() == 0 && d) atomicAdd(out + (blockIdx.x % 64) * 8, (unsigned long long)d);
}
__global__ void __launch_bounds__(SW_BLOCK) k_digest_nodes(const SwDev* __restrict__ Dp, unsigned long long* out) {
  SW_DEV_BIND
  size_t NL = (size_t)D.R * D.nloc, l = (size_t)blockIdx.x * SW_BLOCK + threadIdx.x;
  uint64_t d = 0;
  if (l < NL) {
    uint32_t r = (uint32_t)(l / D.nloc), i = D.i0 + (uint32_t)(l % D.nloc); uint64_t g = (uint64_t)r * D.N + i;
    uint4 h = HDR(l), p0 = D.pr0[l]; uint2 p = D.ph[l];
    d += sw_h3(1, g, ((uint64_t)h.x << 32) | ((uint64_t)p_aw(p.y) << 8) | h_leaving(h.y));
    d += sw_h3(2, g, ((uint64_t)p.x << 32) | p_epoch(p.y));
    if (p0.x != NONE)
      d += sw_h3(3, g, ((uint64_t)p0.x << 32) | p0.z) + sw_h3(4, g, ((uint64_t)p0.y << 32) | ((uint64_t)p_stage(p.y) << 8) | p_nackm(p.y));
    for (uint32_t j = 0; j < h_qlen(h.y); j++) {
      uint4 e = QENT(j, l);
      d += sw_h3(5, g, sw_h3(e.x, ((uint64_t)e.y << 32) | e.z, ((uint64_t)m_seq(e.w) << 16) | ((uint64_t)m_tr(e.w) << 8) | m_type(e.w)));
    }
    d += sw_h3(6, g, ((uint64_t)h.z << 32) | h.w);
    if (D.sslt) { const uint32_t ss = D.sslt[l]; if (ss) d += sw_h3(19, g, ((uint64_t)(ss >> 31) << 32) | (ss & 0x7FFFFFFFu)); }
    for (uint32_t j = 0; j < h_evqlen(h.y); j++) {
      uint4 e = D.evq[(size_t)j * NL + l];
      d += sw_h3(7, g, sw_h3(e.x, e.y, ((uint64_t)m_seq(e.w) << 16) | ((uint64_t)m_tr(e.w) << 8)));
    }
    if (D.flags & SWIM_F_SERF_EVENTS)
      for (uint32_t b = 0; b < D.EB; b++) {
        const uint4* slot = D.ring + (size_t)b * D.EW * NL + l;
        const uint4 w0 = slot[0]; const uint32_t n = w0.y, lt = w0.x;
        for (uint32_t i = 0; i < n; i++) {
          const uint32_t* w = (const uint32_t*)&slot[(size_t)((i + 2) / 4) * NL];
          d += sw_h3(8, g, ((uint64_t)lt << 32) | w[(i + 2) & 3u]);
        }
      }
    if (D.coord) {                               // coordinates: the raw bits
      const double* f = (const double*)&D.coord[l];
      for (uint32_t j = 0; j < SWIM_COORD_DIMS + 3; j++) d += sw_h3(20 + j, g, (uint64_t)__double_as_longlong(f[j]));
    }
  }
  digest_commit(d, out);
}
__global__ void __launch_bounds__(SW_BLOCK) k_digest_views(const SwDev* __restrict__ Dp, unsigned long long* out) {
  SW_DEV_BIND
  const size_t NL = (size_t)D.R * D.nloc, l = (size_t)blockIdx.x * SW_BLOCK + threadIdx.x;
  uint64_t d = 0;
  if (l < NL) {
    const uint32_t r = (uint32_t)(l / D.nloc), o = D.i0 + (uint32_t)(l % D.nloc);
    uint32_t left = VMETA(l).x;
    for (uint32_t sl = 0; sl < D.VT && left; sl++) {          // explicit views
      const uint4 a = D.vt[(size_t)sl * NL + l];
      if (a.x == VT_EMPTY) continue;
      left--;
      const uint64_t id = ((uint64_t)r << 40) ^ ((uint64_t)a.x * 0x100000001B3ull) ^ ((uint64_t)o << 8);
      d += sw_h3(9, id, ((uint64_t)a.y << 32) | a.z);
      if (SW_KST(a.y) >= SWIM_STATE_DEAD && (a.w & 1u)) d += sw_h3(16, id, 1);
      if (SW_KST(a.y) == SWIM_STATE_SUSPECT ? vw_leaving(a.w) : (a.w >> 1) & 1u) d += sw_h3(17, id, 1);
      if (D.vs) { const uint32_t lt = D.vs[(size_t)sl * NL + l]; if (lt) d += sw_h3(18, id, lt); }
      if (SW_KST(a.y) == SWIM_STATE_SUSPECT) {
        const uint32_t nc = vw_nconf(a.w); const uint4 cf = D.vc[(size_t)sl * NL + l];
        d += sw_h3(10, id, nc);
        d += sw_h3(11, id, vw_conf0(a.w));
        // (the accusers that can still matter: suspicion.Confirm returns early once k confirmations are in, so the k-th
        // confirmer's name is never looked at again)
        const uint32_t kk = susp_k_n(D, cf.w);
        if (nc >= 1 && kk > 1) d += sw_h3(12, id, cf.x);
        if (nc >= 2 && kk > 2) d += sw_h3(13, id, cf.y);
        if (nc >= 3 && kk > 3) d += sw_h3(14, id, cf.z);
      }
    }
    const uint64_t g = (uint64_t)r * D.N + o;                  // the base row: every shard digests its own id range
    const uint32_t bk = D.bk[g];
    if (bk != SW_BASE_KEY) d += sw_h3(15, g, bk);
  }
  digest_commit(d, out);
}

// the pairs of the dense store, hashed exactly like the explicit views of the hash tables: one workgroup per row
__global__ void __launch_bounds__(SW_BLOCK) k_digest_mass(const SwDev* __restrict__ Dp, unsigned long long* out) {
  SW_DEV_BIND
  const uint32_t r = blockIdx.x / D.M, row = blockIdx.x % D.M, x = D.mrow_subj[blockIdx.x];
  uint64_t d = 0;
  if (x != NONE)
    for (uint32_t k = threadIdx.x; k < D.nloc; k += SW_BLOCK) {
      const size_t idx = m_idx(D, r, row, k);
      const uint32_t a = D.mA[idx];
      if (!a) continue;
      uint32_t c1; const uint4 e = m_unpack(D, x, a, D.mB[idx], D.mC[idx], c1);
      const uint64_t id = ((uint64_t)r << 40) ^ ((uint64_t)x * 0x100000001B3ull) ^ ((uint64_t)(D.i0 + k) << 8);
      d += sw_h3(9, id, ((uint64_t)e.y << 32) | e.z);
      if (SW_KST(e.y) >= SWIM_STATE_DEAD && (e.w & 1u)) d += sw_h3(16, id, 1);
      if (SW_KST(e.y) == SWIM_STATE_SUSPECT ? vw_leaving(e.w) : (e.w >> 1) & 1u) d += sw_h3(17, id, 1);
      if (D.mD) { const uint32_t lt = D.mD[idx]; if (lt) d += sw_h3(18, id, lt); }
      if (SW_KST(e.y) == SWIM_STATE_SUSPECT) {
        const uint32_t nc = vw_nconf(e.w);
        d += sw_h3(10, id, nc);
        d += sw_h3(11, id, vw_conf0(e.w));
        if (nc >= 1 && D.susp_k > 1) d += sw_h3(12, id, c1);
      }
      if (D.iq) {                                    // the rumour this observer has queued about the subject (what k_digest_nodes hashes of a slot's entry)
        const uint32_t qe = D.mE[e_idx(D, r, row, k)];
        if (qe & QE_QUEUED) {
          const uint32_t f = D.mF[idx];
          d += sw_h3(5, (uint64_t)r * D.N + D.i0 + k, sw_h3(x, ((uint64_t)(MA_INC(a) + QF_DELTA(f)) << 32) | QF_FROM(f), ((uint64_t)QE_SEQ(qe) << 16) | ((uint64_t)QE_TR(qe) << 8) | QE_TYPE(qe)));
        }
      }
    }
  digest_commit(d, out);
}
// swim_view / swim_members: one observer's explicit views, gathered for the host: out[0] = count, then {vt, vc} pairs
__global__ void k_gather_views(const SwDev* __restrict__ Dp, uint32_t r, uint32_t o, uint32_t* out, uint32_t cap) {
  SW_DEV_BIND
  if (threadIdx.x || blockIdx.x) return;
  const size_t NL = (size_t)D.R * D.nloc, l = (size_t)r * D.nloc + (o - D.i0);
  uint32_t n = 0;
  for (uint32_t sl = 0; sl < D.VT; sl++) {
    const uint4 a = D.vt[(size_t)sl * NL + l];
    if (a.x == VT_EMPTY) continue;
    if (n < cap) { const uint4 c = D.vc[(size_t)sl * NL + l]; uint32_t* w = out + 4 + (size_t)n * 8; w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = c.x; w[5] = c.y; w[6] = c.z; w[7] = 0; }
    n++;
  }
  if (D.M && D.mcnt[l]) {                       // ... and its pairs of the dense store, in the same form
    const uint32_t nr = D.M;
    for (uint32_t row = 0; row < nr; row++) {
      const uint32_t x = D.mrow_subj[(size_t)r * D.M + row];
      if (x == NONE) continue;
      const size_t idx = m_idx(D, r, row, o - D.i0);
      const uint32_t a = D.mA[idx];
      if (!a) continue;
      if (n < cap) { uint32_t c1; const uint4 e = m_unpack(D, x, a, D.mB[idx], D.mC[idx], c1); uint32_t* w = out + 4 + (size_t)n * 8; w[0] = e.x; w[1] = e.y; w[2] = e.z; w[3] = e.w; w[4] = c1; w[5] = 0; w[6] = 0; w[7] = 0; }
      n++;
    }
  }
  out[0] = n;
}
// swim_census_get for a subject without a watch slot: pass 0 = highest incarnation any local observer holds, pass 1 = the
// live observers' views by state and how many are at that incarnation.  acc = {obs, st0..st3, cur, maxinc}
__global__ void __launch_bounds__(SW_BLOCK) k_census_adhoc(const SwDev* __restrict__ Dp, uint32_t r, uint32_t x, int pass, uint32_t* acc) {
  SW_DEV_BIND
  const uint32_t* nw = D.nw + (size_t)r * D.N;
  const uint32_t wx = nw[x], bkey = base_key_of(D, r, x, wx), maxinc = acc[6], mrow_x = (wx & NW_MASS) ? D.mrow[(size_t)r * D.N + x] : 0;
  uint32_t vals[7] = { 0, 0, 0, 0, 0, 0, SW_KINC(bkey) };
  for (uint32_t k = blockIdx.x * SW_BLOCK + threadIdx.x; k < D.nloc; k += gridDim.x * SW_BLOCK) {
    const uint32_t o = D.i0 + k;
    uint32_t key = bkey;
    if (wx & NW_MASS) { const uint32_t a = D.mA[m_idx(D, r, mrow_x, k)]; if (a) key = MA_KEY(a); }
    else if (wx & NW_SUBJECT) { uint4 e; uint32_t fs; if (vt_find(D, (size_t)r * D.nloc + k, x, e, fs) != NONE) key = e.y; }
    if (pass == 0) { vals[6] = SW_KINC(key) > vals[6] ? SW_KINC(key) : vals[6]; continue; }
    if (o == x || (nw[o] & NW_DEAD)) continue;
    vals[0]++; vals[1 + SW_KST(key)]++; vals[5] += SW_KINC(key) == maxinc;
  }
  if (pass == 0) {
    uint32_t m = vals[6];
    for (int off = 32; off; off >>= 1) { uint32_t v = __shfl_down(m, off); m = v > m ? v : m; }
    if (sw_lane() == 0) atomicMax(&acc[6], m);
    return;
  }
#pragma unroll
  for (int j = 0; j < 6; j++) {
    uint32_t v = vals[j];
    for (int off = 32; off; off >>= 1) v += __shfl_down(v, off);
    if (sw_lane() == 0 && v) atomicAdd(&acc[j], v);
  }
}

// serf.go handleReap (reap ticks only): at every observer the simulator acts for, a member Failed for longer than
// ReconnectTimeout or Left for longer than TombstoneTimeout is erased (status NONE from then on) + EventMemberReap
__global__ void __launch_bounds__(SW_BLOCK) k_reap(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  const size_t NL = (size_t)D.R * D.nloc, l = (size_t)blockIdx.x * SW_BLOCK + threadIdx.x;
  if (l >= NL) return;
  uint32_t left = VMETA(l).x;
  if (!left) return;
  const uint32_t r = (uint32_t)(l / D.nloc), o = D.i0 + (uint32_t)(l % D.nloc), t = *D.tick, now = now_ms(D, t);
  if (D.nw[(size_t)r * D.N + o] & NW_INERT) return;
  uint32_t n = 0;
  for (uint32_t sl = 0; sl < D.VT && left; sl++) {
    uint4 e = D.vt[(size_t)sl * NL + l];
    if (e.x == VT_EMPTY) continue;
    left--;
    const uint32_t st = SW_KST(e.y);
    if (st < SWIM_STATE_DEAD || (e.w & 1u) || e.x == o) continue;
    if (!(now - e.z > (st == SWIM_STATE_DEAD ? D.reconnect_timeout_ms : D.tombstone_timeout_ms))) continue;
    D.vt[(size_t)sl * NL + l].w = e.w | 1u; n++;
    const uint32_t wx = D.nw[(size_t)r * D.N + e.x];
    if (NW_HAS_SLOT(wx)) D.slot_dirty[(size_t)r * D.S + NW_SLOT(wx)] = 1;
    bool ev_ch = o == D.watch;
    if (!ev_ch && D.ev_any) { const uint32_t nw_ = D.ev_watch[(size_t)D.R * SWIM_EVENT_WATCHERS + r]; for (uint32_t j = 0; j < nw_; j++) ev_ch |= D.ev_watch[(size_t)r * SWIM_EVENT_WATCHERS + j] == o; }
    if (ev_ch) {
      uint32_t pos = atomicAdd(D.ev_cnt, 1u);
      if (pos < D.ev_cap) { swim_event ev = { now, r, SWIM_EVENT_MEMBER_REAP, e.x, SW_KINC(e.y), o, 0 }; D.events[pos] = ev; }
      else atomicOr(D.err, SW_ERR_EVENT_OVF);
    }
  }
  if (n) atomicAdd(stat_ptr(D, ST_REAPED), (unsigned long long)n);
}

// =================================================================================================
// fold (oracle: fold_census / fold_apply) — every fold_period ticks a subject on which ALL
// acting observers of the whole population hold the same settled explicit view moves into the base row and its
// entries are freed.  k_fold_scan: what this shard's acting observers hold; k_fold_emit: one record per
// subject and shard into the tick's outbound lists; k_deliver accumulates every shard's records (fg_*);
// k_fold_apply (between k_deliver and k_resolve) decides and frees.
// =================================================================================================
// ... and the members whose views live in the dense pair store: a workgroup per row (round 6: mass_rows no longer excludes the reaper)
__global__ void __launch_bounds__(SW_BLOCK) k_reap_mass(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  const uint32_t rr = blockIdx.x, r = rr / D.M, row = rr % D.M, x = D.mrow_subj[rr];
  if (x == NONE) return;
  const uint32_t t = *D.tick, now = now_ms(D, t);
  uint32_t n = 0;
  for (uint32_t k = threadIdx.x; k < D.nloc; k += SW_BLOCK) {
    const size_t idx = m_idx(D, r, row, k);
    const uint32_t a = D.mA[idx], st = MA_STATE(a), o = D.i0 + k;
    if (!a || st < SWIM_STATE_DEAD || MA_ERASED(a) || x == o) continue;
    if (D.nw[(size_t)r * D.N + o] & NW_INERT) continue;
    if (!(now - MB_TICK(D.mB[idx]) * D.quantum_ms > (st == SWIM_STATE_DEAD ? D.reconnect_timeout_ms : D.tombstone_timeout_ms))) continue;
    D.mA[idx] = a | (1u << 5); n++;
    bool ev_ch = o == D.watch;
    if (!ev_ch && D.ev_any) { const uint32_t nw_ = D.ev_watch[(size_t)D.R * SWIM_EVENT_WATCHERS + r]; for (uint32_t j = 0; j < nw_; j++) ev_ch |= D.ev_watch[(size_t)r * SWIM_EVENT_WATCHERS + j] == o; }
    if (ev_ch) {
      const uint32_t pos = atomicAdd(D.ev_cnt, 1u);
      if (pos < D.ev_cap) { swim_event ev = { now, r, SWIM_EVENT_MEMBER_REAP, x, MA_INC(a), o, 0 }; D.events[pos] = ev; }
      else atomicOr(D.err, SW_ERR_EVENT_OVF);
    }
  }
  for (int off = 32; off; off >>= 1) n += __shfl_down(n, off);
  if (sw_lane() == 0 && n) {
    atomicAdd(stat_ptr(D, ST_REAPED), (unsigned long long)n);
    const uint32_t wx = D.nw[(size_t)r * D.N + x];
    if (NW_HAS_SLOT(wx)) D.slot_dirty[(size_t)r * D.S + NW_SLOT(wx)] = 1;
  }
}
__global__ void __launch_bounds__(SW_BLOCK) k_fold_scan(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  const size_t NL = (size_t)D.R * D.nloc, l = (size_t)blockIdx.x * SW_BLOCK + threadIdx.x;
  const uint32_t now = now_ms(D, *D.tick);
  uint32_t left = 0, r = 0;
  if (l < NL) {
    r = (uint32_t)(l / D.nloc);
    if (!(D.nw[(size_t)r * D.N + D.i0 + (uint32_t)(l % D.nloc)] & NW_INERT)) left = VMETA(l).x;
  }
  // round VT stands for the node's view of ITSELF when that is implicit (alive at its own incarnation) and not what
  // the base row says: it takes part in the census like an explicit view (there is nothing to free later)
  bool acting = false, saw_self = false; uint32_t o = 0;
  if (l < NL) { o = D.i0 + (uint32_t)(l % D.nloc); acting = !(D.nw[(size_t)r * D.N + o] & NW_INERT); }
  for (uint32_t sl = 0; sl <= D.VT; sl++) {
    if (sl < D.VT && !__any(left != 0)) { sl = D.VT - 1; continue; }
    uint4 a = make_uint4(VT_EMPTY, 0, 0, 0);
    if (sl < D.VT) { if (left) a = D.vt[(size_t)sl * NL + l]; }
    else if (acting && !saw_self) {
      const uint32_t self = SW_KEY(HDR(l).x, SWIM_STATE_ALIVE);
      const uint32_t wo = D.nw[(size_t)r * D.N + o];
      const bool in_store = (wo & NW_MASS) && D.mA[m_idx(D, r, D.mrow[(size_t)r * D.N + o], o - D.i0)] != 0;   // (k_fold_scan_mass counts that one)
      if (self != D.bk[(size_t)r * D.N + o] && !in_store) a = make_uint4(o, self, 0, 0);
    }
    bool have = a.x != VT_EMPTY;
    if (have && sl < D.VT) { left--; saw_self |= a.x == o; }
    const uint32_t g = r * D.N + a.x, st = SW_KST(a.y);
    // (with the reaper on, a Failed / Left member stays an explicit view until serf has erased it here)
    const bool bad = st == SWIM_STATE_SUSPECT || (st == SWIM_STATE_DEAD && !(now - a.z > D.gossip_to_dead_ms)) ||
                     (D.reap_period && st >= SWIM_STATE_DEAD && sl < D.VT && !(a.w & 1u)) ||
                     (st == SWIM_STATE_ALIVE && (a.w & 2u)) ||     // (a Leaving mark is not something the base row can hold,
                     (have && sl < D.VT && D.vs && st < SWIM_STATE_DEAD && D.vs[(size_t)sl * NL + l] != 0);   //  nor a live member's statusLTime: a stale leave intent must stay stale)
    // lanes of a wave mostly hold the same subject in the same slot (one failure per cluster): one atomic set per
    // distinct (subject, key, settled) triple present in the wave
    uint64_t todo = __ballot(have);
    while (todo) {
      const uint32_t leader = (uint32_t)__ffsll((long long)todo) - 1;
      const uint32_t g0 = __shfl(g, leader), k0 = __shfl(a.y, leader); const bool b0 = __shfl((int)bad, leader) != 0;
      const bool mine = have && g == g0 && a.y == k0 && bad == b0;
      const uint64_t mm = __ballot(mine);
      if (sw_lane() == leader) {
        atomicAdd(&D.fl_cnt[g0], (uint32_t)__popcll(mm)); atomicMin(&D.fl_kmin[g0], k0); atomicMax(&D.fl_kmax[g0], k0);
        if (b0) D.fl_bad[g0] = 1;
      }
      todo &= ~mm;
    }
  }
}
__global__ void __launch_bounds__(SW_BLOCK) k_fold_emit(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  const size_t n = (size_t)D.R * D.N, g = (size_t)blockIdx.x * SW_BLOCK + threadIdx.x;
  uint32_t cnt = g < n ? D.fl_cnt[g] : 0;
  uint4 rec = make_uint4(NONE, (uint32_t)g, FOLD_POISON, cnt);
  if (cnt && !D.fl_bad[g]) { const uint32_t a = D.fl_kmin[g]; if (a == D.fl_kmax[g]) rec.z = a; }
  if (!__any(cnt != 0)) return;
  for (uint32_t sh = 0; sh < D.n_shards; sh++) wave_append(D, sh, cnt != 0, rec);
  if (D.n_shards > 1 && cnt) *D.act = 1;
}
// every shard takes the same decision for every subject g = replica*N + node from the same accumulated records:
// fl_bad[g] = 0 no, 1 fold, 2 fold and the base row hears of the node for the first time
__global__ void __launch_bounds__(SW_BLOCK) k_fold_decide(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  const size_t n = (size_t)D.R * D.N, g = (size_t)blockIdx.x * SW_BLOCK + threadIdx.x;
  uint32_t code = 0, r = 0, x = 0; bool mine = false;
  if (g < n) {
    r = (uint32_t)(g / D.N); x = (uint32_t)(g % D.N);
    const uint32_t k = D.fg_kmin[g], c = D.fg_cnt[g];
    if (c && c == D.acting[r] && k == D.fg_kmax[g] && k != FOLD_POISON) {
      const uint32_t old = D.bk[g];
      code = (SW_KINC(old) == 0 && SW_KINC(k) != 0) ? 2u : 1u;
      mine = x >= D.i0 && x < D.i0 + D.nloc;
      // the base row's new entry; the subject bit falls (nobody holds a view any more once k_fold_apply is through)
      D.bk[g] = k;
      const uint32_t w = D.nw[g], wn = (w & ~(NW_SUBJECT | NW_BASEMOD)) | (k != SW_BASE_KEY ? NW_BASEMOD : 0u);
      if (w != wn) D.nw[g] = wn;
      if (NW_HAS_SLOT(w)) D.slot_dirty[(size_t)r * D.S + NW_SLOT(w)] = 1;
      *D.fold_any = 1;
      if (code == 2) { atomicAdd(&D.base_known[r], 1u); if (mine && D.dyn) D.vnk[(size_t)r * D.nloc + (x - D.i0)]--; }   // (it counted itself)
    }
    D.fl_bad[g] = code;
  }
  // stats: a folded subject is counted once, by the shard that owns its id
  const uint64_t mm = __ballot(code != 0 && mine);
  if (mm && sw_lane() == 0) atomicAdd(stat_ptr(D, ST_FOLDS), (unsigned long long)__popcll(mm));
}
__global__ void __launch_bounds__(SW_BLOCK) k_fold_apply(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  const size_t NL = (size_t)D.R * D.nloc, l = (size_t)blockIdx.x * SW_BLOCK + threadIdx.x;
  if (l >= NL) return;
  uint4 vm = VMETA(l);
  if (!vm.x) return;
  const uint32_t r = (uint32_t)(l / D.nloc), o = D.i0 + (uint32_t)(l % D.nloc);
  uint32_t freed = 0, nk_less = 0;
  for (uint32_t sl = 0; sl < D.VT; ) {
    const uint4 a = D.vt[(size_t)sl * NL + l];
    const uint32_t code = a.x != VT_EMPTY ? D.fl_bad[(size_t)r * D.N + a.x] : 0u;
    if (!code) { sl++; continue; }
    // (the acting observers agree on a settled view; a node that is not running may still hold a suspicion)
    if (SW_KST(a.y) == SWIM_STATE_SUSPECT && --vm.y == 0) vm.z = NONE;
    nk_less += code == 2 && a.x != o;
    vt_erase(D, l, sl);                                  // an entry may move into slot sl: it is looked at again
    vm.x--; freed++;
  }
  if (freed) {
    VMETA(l) = vm;
    if (nk_less && D.dyn) D.vnk[l] -= nk_less;
    atomicAdd(stat_ptr(D, ST_FOLD_FREED), (unsigned long long)freed);
  }
}

// =================================================================================================
// the dense pair store (swim_device.h): suspicion timers, fold, row allocation
// =================================================================================================
// suspectNode's time.AfterFunc for the pairs of the dense store, two launches: k_expire_mass_due lists the rows whose bound has
// passed (one thread per row: a quiet tick costs one word per row); k_expire_mass spreads the 256-observer tiles of the listed
// rows over its waves — one failure per cluster is ONE row with all the cluster's timers in it, a mass event thousands of rows
// with a few due tiles each — and looks only into tiles whose own bound has passed.  A verdict goes straight into the
// observer's inbox like role_expire's.
__global__ void __launch_bounds__(SW_BLOCK) k_expire_mass_due(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  const uint32_t rr = blockIdx.x * SW_BLOCK + threadIdx.x, now = now_ms(D, *D.tick);
  bool due = false;
  if (rr < D.R * D.M && now >= D.m_row_dl[rr]) { D.m_row_dl[rr] = NONE; due = D.mrow_subj[rr] != NONE; }   // (k_expire_mass rebuilds the bound from the tiles')
  const uint64_t mask = __ballot(due);
  if (!mask) return;
  const uint32_t lane = sw_lane(), leader = (uint32_t)__ffsll((long long)mask) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(D.m_due_cnt, (uint32_t)__popcll(mask));
  base = __shfl(base, leader);
  if (due) D.m_due[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1))] = rr;
}
__global__ void __launch_bounds__(SW_BLOCK) k_expire_mass(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  const uint32_t n_due = *D.m_due_cnt;
  if (!n_due) return;
  const uint32_t now = now_ms(D, *D.tick), lane = sw_lane();
  const uint64_t items = (uint64_t)n_due * D.nbl, stride = (uint64_t)gridDim.x * (SW_BLOCK / 64) * 64;
  uint32_t fired = 0;
  // 64 (row, tile) items per wave step: every lane looks at one tile bound (in a mass event thousands of rows are due with a few
  // due tiles each: one item per wave step was 5 ms per tick of dependent scalar loads), then the wave takes the due ones in turn
  for (uint64_t base = ((uint64_t)blockIdx.x * (SW_BLOCK / 64) + threadIdx.x / 64) * 64; base < items; base += stride) {
    const uint64_t it = base + lane;
    uint32_t rr = 0, tile = 0, tb = NONE;
    if (it < items) { rr = D.m_due[it / D.nbl]; tile = (uint32_t)(it % D.nbl); tb = D.m_tile_dl[(size_t)rr * D.nbl + tile]; }
    const bool due = it < items && now >= tb;
    if (it < items && !due && tb < D.m_row_dl[rr]) atomicMin(&D.m_row_dl[rr], tb);       // (the row's bound is rebuilt from its tiles')
    uint64_t todo = __ballot(due);
    while (todo) {
      const uint32_t leader = (uint32_t)__ffsll((long long)todo) - 1; todo &= todo - 1;
      const uint32_t rr_ = __shfl(rr, leader), tile_ = __shfl(tile, leader), r = rr_ / D.M, row = rr_ % D.M, x = D.mrow_subj[rr_];
      uint32_t m = NONE;
      for (uint32_t part = 0; part < SW_BLOCK / 64; part++) {
        const uint32_t k = tile_ * SW_BLOCK + part * 64 + lane;
        if (k >= D.nloc) continue;
        const size_t idx = m_idx(D, r, row, k);
        const uint32_t a = D.mA[idx];
        if (MA_STATE(a) != SWIM_STATE_SUSPECT) continue;
        const uint32_t o = D.i0 + k;
        if (D.nw[(size_t)r * D.N + o] & NW_INERT) continue;          // its timers rest; a revive lowers the bounds again
        const uint32_t dl = MB_TICK(D.mB[idx]) * D.quantum_ms + susp_timeout_n(D, 0, MA_NCONF(a));
        if (now >= dl) {
          const size_t l = (size_t)r * D.nloc + k;
          inbox_place(D, mk_edge(D, r, o, x, MA_INC(a), SWIM_MSG_DEAD, o), l, atomicAdd(&D.in_cnt[l], 1u));
          fired++;
        }
        m = dl < m ? dl : m;           // a fired timer keeps the bound low until its verdict is merged
      }
      for (int off = 32; off; off >>= 1) { const uint32_t v = __shfl_xor(m, off); m = v < m ? v : m; }
      if (lane == 0) { D.m_tile_dl[(size_t)rr_ * D.nbl + tile_] = m; if (m < D.m_row_dl[rr_]) atomicMin(&D.m_row_dl[rr_], m); }
    }
  }
  if (__any(fired != 0)) {
    for (int off = 32; off; off >>= 1) fired += __shfl_down(fired, off);
    if (lane == 0) { atomicAdd(stat_ptr(D, ST_TIMEOUTS), (unsigned long long)fired); atomicAdd(stat_ptr(D, ST_EDGES), (unsigned long long)fired); }
  }
}
// fold census of the dense store (k_fold_scan's counterpart): what the acting observers hold about the row's subject
__global__ void __launch_bounds__(SW_BLOCK) k_fold_scan_mass(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  const uint32_t rr = blockIdx.x, r = rr / D.M, row = rr % D.M, x = D.mrow_subj[rr];
  if (x == NONE) return;
  const uint32_t now = now_ms(D, *D.tick);
  __shared__ uint32_t s_cnt, s_kmin, s_kmax, s_bad;
  if (threadIdx.x == 0) { s_cnt = 0; s_kmin = NONE; s_kmax = 0; s_bad = 0; }
  __syncthreads();
  uint32_t cnt = 0, kmin = NONE, kmax = 0, bad = 0;
  for (uint32_t k = threadIdx.x; k < D.nloc; k += SW_BLOCK) {
    const size_t idx = m_idx(D, r, row, k);
    const uint32_t a = D.mA[idx];
    if (!a || (D.nw[(size_t)r * D.N + D.i0 + k] & NW_INERT)) continue;
    const uint32_t st = MA_STATE(a), key = MA_KEY(a);
    bad |= st == SWIM_STATE_SUSPECT || (st == SWIM_STATE_DEAD && !(now - MB_TICK(D.mB[idx]) * D.quantum_ms > D.gossip_to_dead_ms)) ||
           (D.reap_period && st >= SWIM_STATE_DEAD && !MA_ERASED(a)) || (st == SWIM_STATE_ALIVE && MA_LEAVING(a)) ||
           (D.mD && st < SWIM_STATE_DEAD && D.mD[idx] != 0);        // (a live member's statusLTime stays with the pair)
    cnt++; kmin = key < kmin ? key : kmin; kmax = key > kmax ? key : kmax;
  }
  for (int off = 32; off; off >>= 1) {
    cnt += __shfl_down(cnt, off); bad |= __shfl_down(bad, off);
    const uint32_t a = __shfl_down(kmin, off), b = __shfl_down(kmax, off); kmin = a < kmin ? a : kmin; kmax = b > kmax ? b : kmax;
  }
  if (sw_lane() == 0 && cnt) { atomicAdd(&s_cnt, cnt); atomicMin(&s_kmin, kmin); atomicMax(&s_kmax, kmax); if (bad) s_bad = 1; }
  __syncthreads();
  if (threadIdx.x == 0 && s_cnt) {
    const size_t g = (size_t)r * D.N + x;
    atomicAdd(&D.fl_cnt[g], s_cnt); atomicMin(&D.fl_kmin[g], s_kmin); atomicMax(&D.fl_kmax[g], s_kmax);
    if (s_bad) D.fl_bad[g] = 1;
  }
}
// a folded subject's row is cleared (every observer's pair, running or not) and goes back to the free stack
__global__ void __launch_bounds__(SW_BLOCK) k_fold_apply_mass(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  const uint32_t rr = blockIdx.x, r = rr / D.M, row = rr % D.M, x = D.mrow_subj[rr];
  if (x == NONE) return;
  const size_t g = (size_t)r * D.N + x;
  if (!D.fl_bad[g]) return;
  uint32_t freed = 0;
  for (uint32_t k = threadIdx.x; k < D.nloc; k += SW_BLOCK) {
    const size_t idx = m_idx(D, r, row, k);
    if (D.mA[idx]) { D.mA[idx] = 0; atomicSub(&D.mcnt[(size_t)r * D.nloc + k], 1u); freed++; }   // (several rows fold in the same tick: one workgroup each, the same observers' counters)
  }
  for (uint32_t tl = threadIdx.x; tl < D.nbl; tl += SW_BLOCK) D.m_tile_dl[(size_t)rr * D.nbl + tl] = NONE;
  for (int off = 32; off; off >>= 1) freed += __shfl_down(freed, off);
  if (sw_lane() == 0 && freed) atomicAdd(stat_ptr(D, ST_FOLD_FREED), (unsigned long long)freed);
  // Every wave has read mrow_subj[rr] (above) before thread 0 clears it: without this barrier a wave that starts after wave 0 has finished
  // finds NONE, leaves, and its quarter of the row stays set in a row that is back on the free stack (found by running the waves of a
  // workgroup one after the other: tools/emu; the device starts them together, which is the only reason it never showed).  Both returns
  // above are workgroup-uniform.
  __syncthreads();
  if (threadIdx.x == 0) {
    D.m_row_dl[rr] = NONE; D.mrow[g] = NONE; D.mrow_subj[rr] = NONE;
    atomicAnd(&D.nw[g], ~NW_MASS);
    const uint32_t pos = atomicAdd(&D.m_nfree[r], 1u);
    D.m_free[(size_t)r * D.M + pos] = row;
  }
}
// Rows for the nodes a stimulus call names (kill / revive / leave / update / join, the minority sides of a partition), while
// rows remain and nobody on this shard holds a hash-table view of the node yet.  One workgroup, the list in order: who gets a
// row when they run out does not depend on scheduling.  (Which row is immaterial.)
#define SW_MROW_PENDING 0xFFFFFFFEu
__global__ void __launch_bounds__(SW_BLOCK) k_mass_alloc(const SwDev* __restrict__ Dp, uint32_t r, const uint32_t* ids, uint32_t n) {
  SW_DEV_BIND
  __shared__ uint32_t s_w[SW_BLOCK / 64];
  const uint32_t lane = sw_lane(), wave = threadIdx.x / 64;
  for (uint32_t a0 = 0; a0 < n; a0 += SW_BLOCK) {
    const uint32_t a = a0 + threadIdx.x;
    bool want = false; size_t g = 0; uint32_t x = 0;
    if (a < n) {
      x = ids[a]; g = (size_t)r * D.N + x;
      if (!(D.nw[g] & (NW_MASS | NW_SUBJECT)) && atomicCAS(&D.mrow[g], NONE, SW_MROW_PENDING) == NONE) want = true;   // (a list may name a node twice)
    }
    const uint64_t mask = __ballot(want);
    if (lane == 0) s_w[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (uint32_t w = 0; w < SW_BLOCK / 64; w++) { base += w < wave ? s_w[w] : 0; total += s_w[w]; }
    const uint32_t avail = D.m_nfree[r], my = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1));
    if (want) {
      if (my < avail) {
        const uint32_t row = D.m_free[(size_t)r * D.M + (avail - 1 - my)];
        D.mrow_subj[(size_t)r * D.M + row] = x; D.mrow[g] = row;
        atomicOr(&D.nw[g], NW_MASS);
      } else D.mrow[g] = NONE;
    }
    __syncthreads();
    if (threadIdx.x == 0) D.m_nfree[r] = avail - (total < avail ? total : avail);
    __syncthreads();
  }
}
__global__ void k_mass_init(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n = (size_t)D.R * D.M;
  if (i < n) { D.mrow_subj[i] = NONE; D.m_row_dl[i] = NONE; D.m_free[i] = D.M - 1 - (uint32_t)(i % D.M); }   // (row 0 is handed out first)
  if (i < D.R) D.m_nfree[i] = D.M;
}

// =================================================================================================
// swim_detection_get (swimsim.h): how the acting observers of this shard see the nodes out of their reach.  acc = {pairs,
// state 0..3} as 64-bit sums (corrections wrap around), grp = acting observers per partition group [128] + their total [128]
// =================================================================================================
__device__ __forceinline__ void det_add(unsigned long long* acc, int i, long long v) {
  for (int off = 32; off; off >>= 1) v += __shfl_down(v, off);
  if (sw_lane() == 0 && v) atomicAdd(&acc[i], (unsigned long long)v);
}
__global__ void __launch_bounds__(SW_BLOCK) k_detect_groups(const SwDev* __restrict__ Dp, uint32_t r, uint32_t* grp) {
  SW_DEV_BIND
  const uint32_t k = blockIdx.x * SW_BLOCK + threadIdx.x;
  if (k >= D.nloc) return;
  const uint32_t w = D.nw[(size_t)r * D.N + D.i0 + k];
  if (!(w & NW_INERT)) { atomicAdd(&grp[NW_PART(w)], 1u); atomicAdd(&grp[128], 1u); }
}
__global__ void __launch_bounds__(SW_BLOCK) k_detect_base(const SwDev* __restrict__ Dp, uint32_t r, const uint32_t* grp, unsigned long long* acc) {
  SW_DEV_BIND
  const uint32_t x = blockIdx.x * SW_BLOCK + threadIdx.x;
  long long n_obs = 0; uint32_t st = 0;
  if (x < D.N) {
    const uint32_t w = D.nw[(size_t)r * D.N + x];
    n_obs = (w & NW_DEAD) ? (long long)grp[128] : (long long)grp[128] - (long long)grp[NW_PART(w)];
    st = SW_KST(base_key_of(D, r, x, w));
  }
  det_add(acc, 0, n_obs);
  for (uint32_t c = 0; c < 4; c++) det_add(acc, 1 + (int)c, st == c ? n_obs : 0);
}
// corrections by the explicit views: the hash tables (one lane per observer) ...
__global__ void __launch_bounds__(SW_BLOCK) k_detect_tables(const SwDev* __restrict__ Dp, uint32_t r, unsigned long long* acc) {
  SW_DEV_BIND
  const uint32_t k = blockIdx.x * SW_BLOCK + threadIdx.x;
  long long d[4] = { 0, 0, 0, 0 };
  if (k < D.nloc) {
    const size_t NL = (size_t)D.R * D.nloc, l = (size_t)r * D.nloc + k;
    const uint32_t o = D.i0 + k, wo = D.nw[(size_t)r * D.N + o];
    uint32_t left = (wo & NW_INERT) ? 0u : VMETA(l).x;
    for (uint32_t sl = 0; sl < D.VT && left; sl++) {
      const uint4 e = D.vt[(size_t)sl * NL + l];
      if (e.x == VT_EMPTY) continue;
      left--;
      if (e.x == o) continue;
      const uint32_t wx = D.nw[(size_t)r * D.N + e.x];
      if (!(wx & NW_DEAD) && NW_PART(wx) == NW_PART(wo)) continue;          // within reach
      d[SW_KST(base_key_of(D, r, e.x, wx))]--; d[SW_KST(e.y)]++;
    }
  }
  for (int c = 0; c < 4; c++) det_add(acc, 1 + c, d[c]);
}
// ... and the dense store (one workgroup per row)
__global__ void __launch_bounds__(SW_BLOCK) k_detect_rows(const SwDev* __restrict__ Dp, uint32_t r, unsigned long long* acc) {
  SW_DEV_BIND
  const uint32_t row = blockIdx.x, x = D.mrow_subj[(size_t)r * D.M + row];
  if (x == NONE) return;
  const uint32_t wx = D.nw[(size_t)r * D.N + x], bst = SW_KST(base_key_of(D, r, x, wx));
  long long d[4] = { 0, 0, 0, 0 };
  for (uint32_t k = threadIdx.x; k < D.nloc; k += SW_BLOCK) {
    const uint32_t a = D.mA[m_idx(D, r, row, k)], o = D.i0 + k;
    if (!a || o == x) continue;
    const uint32_t wo = D.nw[(size_t)r * D.N + o];
    if ((wo & NW_INERT) || (!(wx & NW_DEAD) && NW_PART(wx) == NW_PART(wo))) continue;
    d[bst]--; d[MA_STATE(a)]++;
  }
  for (int c = 0; c < 4; c++) det_add(acc, 1 + c, d[c]);
}

// =================================================================================================
// serf.go reconnect() (swim_config.reconnect_interval_ms; the checker's phase_reconnect has the rule spelt out): on a
// probe-interval boundary the nodes due within the next ProbeInterval count the members they hold Failed (explicit Dead views
// serf has not erased), pass the failed/alive gate on a Philox word, pick the member with the smallest keyed hash and — when it
// runs and is in reach — exchange state with it like a join does (its answer comes one tick later through the reply list).
// grid = (blocks over the due set, R).  The dense store's rows are walked by the whole wave, one due node at a time.
// =================================================================================================
// The dense store's part of serf's reconnect(): how many members a due node holds Failed and which of them its draw picks (the smallest
// hash).  A due node's pairs are a COLUMN of the store — one line per row — so the column is cut into chunks of SW_RC_CHUNK rows and every
// (due node, chunk) gets a wave of its own; the partial counts and minima meet in rc_cnt / rc_best (atomics), k_reconnect reads them.
// (Rounds 3-4 walked the columns inside k_reconnect, a wave on one due node at a time: 2 184 columns of 65 536 rows on 35 waves — 64 ms per
// call, a fifth of the partition leg's kernel time; profiles/r05_config4_partition_kernel_stats.csv.)
#define SW_RC_CHUNK 2048u
__global__ void __launch_bounds__(SW_BLOCK) k_reconnect_scan(const SwDev* __restrict__ Dp, uint32_t lanes) {
  SW_DEV_BIND
  const uint32_t t = *D.tick, r = blockIdx.y, per = D.rc_period, grp = D.P < per ? D.P : per, lane = sw_lane();
  const uint32_t nchunk = (D.M + SW_RC_CHUNK - 1) / SW_RC_CHUNK;
  const uint64_t wv = (uint64_t)blockIdx.x * (SW_BLOCK / 64) + threadIdx.x / 64;
  const uint32_t a = (uint32_t)(wv / nchunk), chunk = (uint32_t)(wv % nchunk);
  if (a >= lanes) return;
  const uint64_t i64 = (uint64_t)((t + a % grp) % per) + (uint64_t)(a / grp) * per;
  if (i64 >= D.N) return;
  const uint32_t o_ = (uint32_t)i64;
  if (o_ < D.i0 || o_ >= D.i0 + D.nloc || (D.nw[(size_t)r * D.N + o_] & NW_INERT)) return;
  const size_t l = (size_t)r * D.nloc + (o_ - D.i0);
  if (!D.mcnt[l]) return;
  uint32_t w[4];
  { const uint64_t sr = seed_of(D, r); sw_philox(t, o_, 0, 0x5245434Eu, (uint32_t)sr, (uint32_t)(sr >> 32) ^ SW_STREAM_RECONNECT, w); }
  const uint32_t key_ = w[1];
  uint32_t cnt = 0, b = NONE, bh = 0;
  const uint32_t row_end = (chunk + 1) * SW_RC_CHUNK < D.M ? (chunk + 1) * SW_RC_CHUNK : D.M;
  for (uint32_t row = chunk * SW_RC_CHUNK + lane; row < row_end; row += 64) {
    const uint32_t x = D.mrow_subj[(size_t)r * D.M + row];
    if (x == NONE) continue;
    const uint32_t av = D.mA[m_idx(D, r, row, o_ - D.i0)];
    if (av && x != o_ && MA_STATE(av) == SWIM_STATE_DEAD && !MA_ERASED(av)) {
      cnt++;
      const uint32_t h = sw_fmix32(x ^ key_);
      if (b == NONE || h < bh || (h == bh && x < b)) { b = x; bh = h; }
    }
  }
  for (int off = 32; off; off >>= 1) {
    cnt += __shfl_xor(cnt, off);
    const uint32_t ob = __shfl_xor(b, off), obh = __shfl_xor(bh, off);
    if (ob != NONE && (b == NONE || obh < bh || (obh == bh && ob < b))) { b = ob; bh = obh; }
  }
  if (lane == 0 && cnt) {
    const size_t slot = (size_t)r * lanes + a;
    atomicAdd(&D.rc_cnt[slot], cnt);
    atomicMin(&D.rc_best[slot], ((unsigned long long)bh << 32) | b);
  }
}
__global__ void __launch_bounds__(SW_BLOCK) k_reconnect(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  __shared__ uint32_t lds_stats[ST_COUNT];
  BlockStats S; S.init(lds_stats);
  const uint32_t t = *D.tick, r = blockIdx.y, a = blockIdx.x * SW_BLOCK + threadIdx.x, per = D.rc_period, grp = D.P < per ? D.P : per, lane = sw_lane();
  const uint64_t i64 = (uint64_t)((t + a % grp) % per) + (uint64_t)(a / grp) * per;
  const size_t NL = (size_t)D.R * D.nloc;
  bool due = false; uint32_t o = 0; size_t l = 0;
  if (i64 < D.N) {
    o = (uint32_t)i64;
    if (o >= D.i0 && o < D.i0 + D.nloc && !(D.nw[(size_t)r * D.N + o] & NW_INERT)) { due = true; l = (size_t)r * D.nloc + (o - D.i0); }
  }
  uint32_t w[4] = { 0, 0, 0, 0 };
  if (due) { const uint64_t sr = seed_of(D, r); sw_philox(t, o, 0, 0x5245434Eu, (uint32_t)sr, (uint32_t)(sr >> 32) ^ SW_STREAM_RECONNECT, w); }
  uint32_t n_failed = 0, best = NONE, best_h = 0;
  if (due) {                                         // the hash table, one lane per due node
    uint32_t left = VMETA(l).x;
    for (uint32_t sl = 0; sl < D.VT && left; sl++) {
      const uint4 e = D.vt[(size_t)sl * NL + l];
      if (e.x == VT_EMPTY) continue;
      left--;
      if (e.x == o || SW_KST(e.y) != SWIM_STATE_DEAD || (e.w & 1u)) continue;
      n_failed++;
      const uint32_t h = sw_fmix32(e.x ^ w[1]);
      if (best == NONE || h < best_h || (h == best_h && e.x < best)) { best = e.x; best_h = h; }
    }
  }
  if (D.M && due) {                                  // the dense store: k_reconnect_scan has been through the due nodes' columns
    const size_t slot = (size_t)r * gridDim.x * SW_BLOCK + a;
    const uint32_t cnt = D.rc_cnt[slot]; const unsigned long long key = D.rc_best[slot];
    if (cnt) {
      const uint32_t b = (uint32_t)key, bh = (uint32_t)(key >> 32);
      n_failed += cnt;
      if (best == NONE || bh < best_h || (bh == best_h && b < best)) { best = b; best_h = bh; }
    }
  }
  bool go = false;
  if (due && n_failed) {
    const uint32_t members = est_n(D, r, l), alive = members > n_failed ? members - n_failed : 1u;
    if ((uint64_t)w[0] * alive <= ((uint64_t)n_failed << 32)) {          // rand.Float32() <= prob
      S.add(ST_RECONNECTS);
      const uint32_t wo = D.nw[(size_t)r * D.N + o], wp = D.nw[(size_t)r * D.N + best];
      go = !(wp & NW_DEAD) && NW_PART(wo) == NW_PART(wp);                  // else the dial fails
      if (go) S.add(ST_RECONNECT_OK);
    }
  }
  uint32_t c_edges = 0, c_remote = 0, c_filt = 0;
  send_state<true>(D, go, r, o, best, c_edges, c_remote, c_filt);
  const uint32_t sh = go ? best / D.nloc : 0;
  wave_append_sharded(D, go, sh, mk_edge(D, r, best, SWIM_SUBJECT_PULL, o, SWIM_MSG_ALIVE, 0));
  c_edges += go; c_remote += go && sh != D.rank;
  S.wave_add(ST_EDGES, c_edges); S.wave_add(ST_EDGES_REMOTE, c_remote); S.wave_add(ST_FILTERED, c_filt);
  if (D.n_shards > 1 && __any(go) && lane == 0) *D.act = 1;
  S.flush(D);
}

// =================================================================================================
// k_send_mass — the dense store's part of the state exchanges send_state listed during k_begin / k_reconnect (xs_list): one
// exchange per wave, 64 rows per step — what mergeState would derive from the owner's pair with each row's subject (Alive ->
// alive, Left -> dead{From: node}, Dead | Suspect -> suspect{From: receiver}), the no-op filter against the receiver's pair, a
// wave-aggregated append — then the exchange's two trailing records.
// =================================================================================================
__global__ void __launch_bounds__(SW_BLOCK) k_send_mass(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  const uint32_t n = *D.xs_cnt < D.xs_cap ? *D.xs_cnt : D.xs_cap, lane = sw_lane();
  uint32_t c_edges = 0, c_remote = 0, c_filt = 0;
  for (uint32_t e = blockIdx.x * (SW_BLOCK / 64) + threadIdx.x / 64; e < n; e += gridDim.x * (SW_BLOCK / 64)) {
    const uint4 x4 = D.xs_list[e];
    const uint32_t r_ = x4.x, own_ = x4.y, dst_ = x4.z, sh_ = dst_ / D.nloc;
    uint32_t left_ = D.mcnt[(size_t)r_ * D.nloc + (own_ - D.i0)];
    const bool filt_ = (D.flags & SWIM_F_FILTER_NOOP) && sh_ == D.rank;
    uint64_t hit_dst = 0, hit_self = 0;
    for (uint32_t row0 = 0; row0 < D.M && left_; row0 += 64) {
      const uint32_t row = row0 + lane;
      bool present = false, want = false; uint4 rec = make_uint4(0, 0, 0, 0); uint32_t x = NONE;
      if (row < D.M) {
        x = D.mrow_subj[(size_t)r_ * D.M + row];
        if (x != NONE) {
          const uint32_t a = D.mA[m_idx(D, r_, row, own_ - D.i0)];
          if (a) {
            present = true; want = true;
            const uint32_t st = MA_STATE(a); uint32_t type, from = 0;
            if (st == SWIM_STATE_ALIVE) type = SWIM_MSG_ALIVE;
            else if (st == SWIM_STATE_LEFT) { type = SWIM_MSG_DEAD; from = x; }
            else { type = SWIM_MSG_SUSPECT; from = dst_; }
            // (the subject's real node word: when the receiver holds no pair the base row decides, and a row's subject may have a modified base row — fold, then revive / kill)
            if (filt_ && x != dst_ && noop_at_receiver<true>(D, r_, (size_t)r_ * D.nloc + (dst_ - D.i0), D.nw[(size_t)r_ * D.N + x] | NW_MASS, make_uint4(x, MA_INC(a), from, type << 30), false, rec)) { want = false; c_filt++; }
            rec = mk_edge(D, r_, dst_, x, MA_INC(a), type, from);
            if (sh_ != D.rank && (D.flags & SWIM_F_FILTER_NOOP) && x != dst_) rec.w |= SW_EDGE_JUDGE;       // the receiver's shard judges
          }
        }
      }
      left_ -= (uint32_t)__popcll(__ballot(present));
      hit_dst |= __ballot(present && x == dst_); hit_self |= __ballot(present && x == own_);
      wave_append(D, sh_, want, rec);
      c_edges += want && !(rec.w & SW_EDGE_JUDGE); c_remote += want && sh_ != D.rank;
    }
    send_state_tail(D, lane == 0, r_, own_, dst_, (x4.w & 1u) || hit_dst, (x4.w & 2u) || hit_self, c_edges, c_remote);
  }
  if (__any((c_edges | c_filt) != 0)) {
    for (int off = 32; off; off >>= 1) { c_edges += __shfl_down(c_edges, off); c_remote += __shfl_down(c_remote, off); c_filt += __shfl_down(c_filt, off); }
    if (lane == 0) {
      if (c_edges) atomicAdd(stat_ptr(D, ST_EDGES), (unsigned long long)c_edges);
      if (c_remote) atomicAdd(stat_ptr(D, ST_EDGES_REMOTE), (unsigned long long)c_remote);
      if (c_filt) atomicAdd(stat_ptr(D, ST_FILTERED), (unsigned long long)c_filt);
      if (D.n_shards > 1 && c_edges) *D.act = 1;
    }
  }
}

// =================================================================================================
// SWIM_F_UNBOUNDED_QUEUE — memberlist's TransmitLimitedQueue, unbounded as it is upstream (queue.go; Consul sizes only serf's event
// queue: internal/gossip/libserf/serf.go:24-27), IMPLIED by the dense pair store (swim_device.h: mE / mF / iqn).
// QueueBroadcast is a store into the pair (NodeCtxT::broadcast_v).  GetBroadcasts is a SELECTION over the node's column: a WAVE per node
// reads the column's queue words coalesced (64 rows per load), keeps per length rank the candidates that can possibly be taken this tick
// (a packet takes at most budget / (2 + len) rumours of a length, so `packets x that many` per rank bound what the greedy walk can reach:
// an entry beyond them is preceded, in queue.go's order, by more entries of its own length than all the packets together can take or
// bump: a selection of the K smallest keys per rank whenever the pool of candidates fills, iq_compact), sorts what is left at the end of the scan by
// (transmits asc, length desc, sequence desc) with a bitonic network in LDS, and then walks that order exactly as the checker walks its sorted queue: take what fits, bump transmits after the sweep, retire at the retransmit limit.  The node's rumour about
// ITSELF and rumours about subjects without a row sit in the queue_cap slots as before and take part in the same order.
// =================================================================================================
#ifdef SWIMSIM_DIAG
// diagnostics (-DSWIMSIM_DIAG builds, SWIMSIM_IQCLK=1): where a wave of k_gossip_iq spends a node — s_memtime ticks summed over all nodes:
// [0] the column scan with its compactions, [1] the compactions alone, [2] picks + bumps + re-sorts, [3] the packets' loads, filter and records,
// [4] write-back, [5] nodes, [6] compactions, [7] of the scan: waiting for the batches' loads
// (tallied per workgroup in LDS, one global atomic per counter and workgroup: the first version added to the global counters per batch of the
// scan — 7 M same-address atomics a tick, 12 ns apiece, made the "scan" 95 % of a node and the kernel 4x slower: the clock measured itself)
__device__ unsigned long long g_iqclk[8];
__shared__ unsigned long long g_s_iqclk[8];
#define IQCLK_T() __builtin_amdgcn_s_memtime()
#define IQCLK_ADD(i, v) do { if (sw_lane() == 0) atomicAdd(&g_s_iqclk[i], (unsigned long long)(v)); } while (0)
#define IQCLK_INIT() do { if (threadIdx.x < 8) g_s_iqclk[threadIdx.x] = 0; __syncthreads(); } while (0)
#define IQCLK_FLUSH() do { __syncthreads(); if (threadIdx.x < 8 && g_s_iqclk[threadIdx.x]) atomicAdd(&g_iqclk[threadIdx.x], g_s_iqclk[threadIdx.x]); } while (0)
#else
#define IQCLK_T() 0ull
#define IQCLK_ADD(i, v) do { } while (0)
#define IQCLK_INIT() do { } while (0)
#define IQCLK_FLUSH() do { } while (0)
#endif
#define IQ_DIRTY 0x80000000u
#define IQ_EXPL 0x40000000u
#define IQ_RETIRED 0xFFFFFFFFu
// (LDS-typed pointers: handed around as plain pointers the strip's accesses became flat_load / flat_store with a full s_waitcnt each, and the
// thresholds a reference parameter in scratch memory — 1 500 cycles per 64 candidates looked at, 95 % of the kernel: profiles/r06_iq_phase_clock_v2.txt)
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
struct IqWave { lds_u64* pool; lds_u32* taken; lds_u32* evm; lds_u32* xq; lds_u32* scal; lds_u32* sent; };   // one wave's strip of LDS (sent: what each of up to four packets took)
struct IqThr { uint32_t t0, t1, t2; };
struct LdsMetaQ { lds_u32* p; __device__ __forceinline__ lds_u32& meta(uint32_t j) const { return p[j]; } };
__device__ __forceinline__ uint32_t iq_key(DevRef D, uint32_t tr, uint32_t type, uint32_t seq) {
  return (tr << 24) | (sel4(D.len_rank, type) << 22) | (0x3FFFFFu - (seq & 0x3FFFFFu));
}
__device__ __forceinline__ uint64_t iq_ltmask() { return (1ull << sw_lane()) - 1ull; }
// Bitonic sorts of pool[0, P), P a power of two >= 64, ascending, by one wave — two forms:
// (a) iq_sort_s: a stage per LDS round trip, all 64 lanes on a pair each (all of a lane's pairs read before any is written): the pools of up to 256
//     entries — what a scan's last compaction leaves, the re-sorts after a packet's bumps;
// (b) iq_sort_p: a lane holds EIGHT entries in registers — the group {g + i * s, i < 8} that three consecutive stages of the network (strides 4s, 2s, s)
//     keep to themselves — and runs up to three stages on them between one round of LDS reads and one of writes: 16 passes for 512 entries instead of 45
//     stages.  The stages of a level k (strides k/2 ... 1) are cut into passes from the top, so a group never crosses the level's direction bit except in
//     the opening pass, which runs the levels 2, 4 and 8 on eight neighbours at once; the direction is taken per pair from its first element.  Only for a
//     full pool: with fewer entries most lanes idle, and it was slower there (profiles/r06_iq_issue_bound.txt, v3 / v4).
// (History, profiles/r06_iq_phase_clock_v4.txt: a stage per round trip with a loop over the lane's pairs was latency bound, 53 k cycles per sort of 512.
//  Since call 26 of round 6 a full pool is no longer sorted at all — iq_compact selects.)
template <uint32_t P>
__device__ __forceinline__ void iq_sort_s(lds_u64* pool) {
  const uint32_t lane = sw_lane();
  constexpr uint32_t Q = P >= 128 ? P / 128 : 1;
#pragma unroll 1
  for (uint32_t k = 2; k <= P; k <<= 1)
#pragma unroll 1
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      unsigned long long a[Q], b[Q]; uint32_t ia[Q], ib[Q];
#pragma unroll
      for (uint32_t q = 0; q < Q; q++) {
        const uint32_t p = lane + 64u * q;             // pair p (P = 64: the upper half of the wave has none)
        ia[q] = ((p & ~(j - 1u)) << 1) | (p & (j - 1u)); ib[q] = ia[q] | j;
        if (p < P / 2) { a[q] = pool[ia[q]]; b[q] = pool[ib[q]]; }
      }
#pragma unroll
      for (uint32_t q = 0; q < Q; q++)
        if (lane + 64u * q < P / 2 && (a[q] > b[q]) == ((ia[q] & k) == 0)) { pool[ia[q]] = b[q]; pool[ib[q]] = a[q]; }
      wave_lds_sync();
    }
}
template <int JJ>
__device__ __forceinline__ void iq_stage8(unsigned long long (&e)[8], uint32_t g, uint32_t s, uint32_t k) {
#pragma unroll
  for (uint32_t i = 0; i < 8; i++) {
    if (i & (1u << JJ)) continue;
    const bool asc = ((g + i * s) & k) == 0;
    unsigned long long& x = e[i]; unsigned long long& y = e[i | (1u << JJ)];
    const bool sw = (x > y) == asc;
    const unsigned long long lo = sw ? y : x, hi = sw ? x : y;
    x = lo; y = hi;
  }
}
template <uint32_t P>
__device__ __forceinline__ void iq_sort_p(lds_u64* pool) {
  const uint32_t t = sw_lane();
  const bool on = P >= 512 || t < P / 8;
  {   // levels 2, 4, 8: eight neighbours
    const uint32_t g = t * 8u;
    if (on) {
      unsigned long long e[8];
#pragma unroll
      for (uint32_t i = 0; i < 8; i++) e[i] = pool[g + i];
      iq_stage8<0>(e, g, 1, 2);
      iq_stage8<1>(e, g, 1, 4); iq_stage8<0>(e, g, 1, 4);
      iq_stage8<2>(e, g, 1, 8); iq_stage8<1>(e, g, 1, 8); iq_stage8<0>(e, g, 1, 8);
#pragma unroll
      for (uint32_t i = 0; i < 8; i++) pool[g + i] = e[i];
    }
    wave_lds_sync();
  }
#pragma unroll 1
  for (uint32_t lk = 4; (1u << lk) <= P; lk++) {
    const uint32_t k = 1u << lk;
#pragma unroll 1
    for (uint32_t left = lk; left > 0; ) {                  // stages of this level still to run: strides 2^(left-1) ... 1
      const uint32_t nst = left < 3 ? left : 3, ls = left - nst, s = 1u << ls;     // this pass: strides s << (nst-1) ... s
      const uint32_t g = ((t >> ls) << (ls + 3)) | (t & (s - 1u));
      if (on) {
        unsigned long long e[8];
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) e[i] = pool[g + i * s];
        if (nst >= 3) iq_stage8<2>(e, g, s, k);
        if (nst >= 2) iq_stage8<1>(e, g, s, k);
        iq_stage8<0>(e, g, s, k);
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) pool[g + i * s] = e[i];
      }
      wave_lds_sync();
      left = ls;
    }
  }
}
__device__ __forceinline__ void iq_sort(lds_u64* pool, uint32_t P) {
  if (P <= 64) iq_sort_s<64>(pool); else if (P <= 128) iq_sort_s<128>(pool); else if (P <= 256) iq_sort_s<256>(pool); else iq_sort_p<512>(pool);
}
__device__ __attribute__((noinline)) void iq_resort(lds_u64* pool, uint32_t n) {
  uint32_t P = 64; while (P < n) P <<= 1;
  for (uint32_t idx = n + sw_lane(); idx < P; idx += 64) pool[idx] = ~0ull;
  wave_lds_sync();
  iq_sort(pool, P);
}
// Keep, per length rank, the keep[rank] x npk entries that sort first; thr[rank] = the key from which on nothing of that rank needs to be looked at
// any more (0xFFFFFFFF while fewer are known).  A SELECTION, not a sort (until call 25 of round 6 every compaction sorted the pool: ~1 900 vector
// instructions each, 40 % of what k_gossip_iq issues, and the kernel is issue bound — profiles/r06_iq_issue_bound.txt): a lane takes eight entries
// into registers, and the K-th smallest key of a rank is found bit by bit from the top — "how many keys of the rank are <= this prefix, ones below"
// is eight compares whose ballots the scalar unit counts — 29 steps, no LDS traffic.  The survivors go back unordered; only the LAST compaction of
// a scan sorts what is left (<= keep x packets per rank: a sort of 128 or 256, not of 512).  Keys are unique within a node (the sequence number), so
// "key <= K-th smallest" keeps exactly K.
__device__ __forceinline__ uint32_t iq_compact(DevRef D, lds_u64* pool, uint32_t n, uint32_t npk, IqThr& thr, bool last) {
  const uint32_t lane = sw_lane(); const uint64_t lt = iq_ltmask();
  unsigned long long e[8]; uint32_t key[8], rk[8];
#pragma unroll
  for (uint32_t i = 0; i < 8; i++) {
    const uint32_t idx = lane + 64u * i;
    e[i] = idx < n ? pool[idx] : ~0ull;
    key[i] = (uint32_t)(e[i] >> 32);
    rk[i] = idx < n ? ((key[i] >> 22) & 3u) : 4u;                 // (4: no entry)
    if (rk[i] == 3u) rk[i] = 2u;
  }
  wave_lds_sync();                                                  // (every entry is in a register before any goes back)
  uint32_t lim0 = 0, lim1 = 0, lim2 = 0; bool all0 = false, all1 = false, all2 = false, none0 = false, none1 = false, none2 = false;
#pragma unroll 1
  for (uint32_t r = 0; r < 3; r++) {
    const uint32_t K = sel4(D.iq_keep, r) * npk;
    uint32_t kr[8], cnt = 0;
#pragma unroll
    for (uint32_t i = 0; i < 8; i++) { kr[i] = rk[i] == r ? key[i] : 0xFFFFFFFFu; cnt += (uint32_t)__popcll(__ballot(rk[i] == r)); }
    bool all = false, none = false; uint32_t T = 0;
    if (K == 0) none = true;
    else if (cnt < K) all = true;
    else {                                                          // the K-th smallest key of the rank (keys are below 2^29: five bits of transmits on top)
#pragma unroll 1
      for (int b = 28; b >= 0; b--) {
        const uint32_t cand = T | ((1u << b) - 1u);
        uint32_t c = 0;
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) c += (uint32_t)__popcll(__ballot(kr[i] <= cand));
        if (c < K) T |= 1u << b;
      }
      if (r == 0) thr.t0 = T; else if (r == 1) thr.t1 = T; else thr.t2 = T;
    }
    if (r == 0) { lim0 = T; all0 = all; none0 = none; } else if (r == 1) { lim1 = T; all1 = all; none1 = none; } else { lim2 = T; all2 = all; none2 = none; }
  }
  uint32_t m = 0;
#pragma unroll
  for (uint32_t i = 0; i < 8; i++) {
    const bool keep = rk[i] == 0 ? (!none0 && (all0 || key[i] <= lim0)) : rk[i] == 1 ? (!none1 && (all1 || key[i] <= lim1)) : rk[i] == 2 ? (!none2 && (all2 || key[i] <= lim2)) : false;
    const uint64_t mk = __ballot(keep);
    if (keep) pool[m + (uint32_t)__popcll(mk & lt)] = e[i];
    m += (uint32_t)__popcll(mk);
  }
  wave_lds_sync();
  if (last) iq_resort(pool, m);
  return m;
}
// a rank's threshold (a key: transmits << 24 | rank << 22 | ~sequence; 0xFFFFFFFF while none stands) as a limit on the raw queue word with its
// sequence flipped and its type masked (transmits << 26 | ~sequence): key < threshold <=> raw < limit for a word of that rank; 0x80000000 admits every
// queued word
__device__ __forceinline__ uint32_t iq_raw_limit(uint32_t thr) {
  return thr == 0xFFFFFFFFu ? 0x80000000u : (((thr >> 24) & 31u) << 26) | (thr & 0x3FFFFFu);
}
// iq_compact out of line, with the types' raw limits that follow from the new thresholds: iq_build calls it from 33 places (after any of a batch's
// 32 row ballots), and inlined there the scan loop was 43 KB of code — more than the instruction cache two CUs share holds beside the rest of the kernel
struct IqC { uint32_t n, t0, t1, t2, rl0, rl1, rl2, rl3; };
__device__ __attribute__((noinline)) IqC iq_compact_nl(DevRef D, lds_u64* pool, uint32_t n, uint32_t npk, uint32_t t0, uint32_t t1, uint32_t t2, bool last) {
  IqThr thr = { t0, t1, t2 };
  IqC c; c.n = iq_compact(D, pool, n, npk, thr, last);
  c.t0 = thr.t0; c.t1 = thr.t1; c.t2 = thr.t2;
  c.rl0 = iq_raw_limit(sel4(D.len_rank, 0u) == 0 ? thr.t0 : sel4(D.len_rank, 0u) == 1 ? thr.t1 : thr.t2);
  c.rl1 = iq_raw_limit(sel4(D.len_rank, 1u) == 0 ? thr.t0 : sel4(D.len_rank, 1u) == 1 ? thr.t1 : thr.t2);
  c.rl2 = iq_raw_limit(sel4(D.len_rank, 2u) == 0 ? thr.t0 : sel4(D.len_rank, 2u) == 1 ? thr.t1 : thr.t2);
  c.rl3 = iq_raw_limit(sel4(D.len_rank, 3u) == 0 ? thr.t0 : sel4(D.len_rank, 3u) == 1 ? thr.t1 : thr.t2);
  return c;
}
// the candidates of node (r, local k, lane l): its slots' entries and what its column implies; sorted on return
__device__ __attribute__((noinline)) uint32_t iq_build(DevRef D, const IqWave W, uint32_t r, uint32_t k, size_t l, size_t NL, uint32_t qlen, uint32_t iqn, uint32_t npk) {
  const uint32_t lane = sw_lane(); const uint64_t lt = iq_ltmask();
  uint32_t n = 0; IqThr thr = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu };
  const bool rl12 = sel4(D.len_rank, 1u) == sel4(D.len_rank, 2u);
  uint32_t rl0 = 0x80000000u, rl1 = 0x80000000u, rl2 = 0x80000000u, rl3 = 0x80000000u;       // the types' raw limits (below); wave-uniform, renewed by a compaction
#define IQ_COMPACT(last_) do { const IqC c_ = iq_compact_nl(D, W.pool, n, npk, thr.t0, thr.t1, thr.t2, last_); n = c_.n; thr.t0 = c_.t0; thr.t1 = c_.t1; thr.t2 = c_.t2; \
    rl0 = __builtin_amdgcn_readfirstlane(c_.rl0); rl1 = __builtin_amdgcn_readfirstlane(c_.rl1); rl2 = __builtin_amdgcn_readfirstlane(c_.rl2); rl3 = __builtin_amdgcn_readfirstlane(c_.rl3); } while (0)
  {
    const bool have = lane < qlen;
    const uint32_t w = have ? QENT(lane, l).w : 0u;
    const uint64_t m = __ballot(have);
    if (have) W.pool[(uint32_t)__popcll(m & lt)] = ((unsigned long long)iq_key(D, m_tr(w), m_type(w), m_seq(w)) << 32) | IQ_EXPL | (m_type(w) << 28) | lane;
    n = (uint32_t)__popcll(m);
  }
  const uint4* col = (const uint4*)(D.mE + e_col(D, r, k)) + lane;
  uint32_t seen = 0;
  // The scan is a chain of round trips (the next batch is only asked for when this one says there is more to see).  A lane reads FOUR
  // consecutive rows per load — a kilobyte per wave and request — and a batch is eight independent requests.  (First version: a dword per lane,
  // 256 bytes per request, four and then sixteen requests per batch: the phase clock — profiles/r06_iq_phase_clock.txt — showed a node spending
  // 95 % of its 360 us in the scan, 13 us per batch, the whole device moving 1.2 TB/s: the requests in flight per CU, not the bytes, were the
  // limit.)
  const bool count_seen = iqn < 4096u;              // (the early exit is for short queues; a long one is scanned to the end without the bookkeeping)
  for (uint32_t rb0 = 0; rb0 < D.MB && (!count_seen || seen < iqn); rb0 += 8) {
    uint4 ew[8];
#pragma unroll
    for (uint32_t u = 0; u < 8; u++) ew[u] = ld_global_u4(col + (size_t)(rb0 + u < D.MB ? rb0 + u : D.MB - 1u) * (64u * SW_IQ_RB / 4u));     // (unconditional: a conditional load became a loop of single round trips)
#ifdef SWIMSIM_DIAG
    { const unsigned long long tw_ = IQCLK_T(); uint32_t any_ = 0;
#pragma unroll
      for (uint32_t u = 0; u < 8; u++) any_ |= ew[u].x ^ ew[u].w;
      if (__any(any_ == 0x12345u)) IQCLK_ADD(6, 1);               // (uses every load's result: the wait for the batch ends here)
      IQCLK_ADD(7, IQCLK_T() - tw_); }
#endif
#pragma unroll
    for (uint32_t u = 0; u < 8; u++) {
      if (rb0 + u >= D.MB) continue;
      if (count_seen) {                              // (short queues: how many queued words this load held — one reduction, not a ballot per row)
        uint32_t cs = ((ew[u].x >> 31) + (ew[u].y >> 31)) + ((ew[u].z >> 31) + (ew[u].w >> 31));
        for (int off = 32; off; off >>= 1) cs += __shfl_xor(cs, off);
        seen += cs;
      }
      // The four rows' verdicts on the RAW queue words, without building a key: within a type the order (transmits asc, sequence desc) is the order
      // of (transmits << 26 | ~sequence) — the word with its sequence bits flipped and its type bits masked — and a type's length rank is fixed, so
      // each type has one raw limit (wave-uniform, from its rank's threshold); flipping bit 31 as well puts every word that is not queued above any
      // limit.  Three operations and a three-way select per row; the key (with its length-rank look-up) is built for the rare row that qualifies.
      // In the heavy phase of a mass event nearly no load holds a candidate once the thresholds stand: one ballot dismisses the load.
      bool qual4[4]; bool anyq = false;
      if (rl12) {                                      // suspect and dead messages are equally long (every preset): one select per row.  (Type 3, a user
#pragma unroll                                         //  event, never sits in a pair: serf's events have their own queue.)
        for (uint32_t j = 0; j < 4; j++) {
          const uint32_t e = j == 0 ? ew[u].x : j == 1 ? ew[u].y : j == 2 ? ew[u].z : ew[u].w;
          qual4[j] = ((e ^ 0x803FFFFFu) & 0xFC3FFFFFu) < ((e & (3u << 24)) ? rl1 : rl0);
          anyq |= qual4[j];
        }
      } else {
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
          const uint32_t e = j == 0 ? ew[u].x : j == 1 ? ew[u].y : j == 2 ? ew[u].z : ew[u].w;
          const uint32_t lo = (e & (1u << 24)) ? rl1 : rl0, hi = (e & (1u << 24)) ? rl3 : rl2;      // (selects on the type's two bits: an equality chain became branches)
          qual4[j] = ((e ^ 0x803FFFFFu) & 0xFC3FFFFFu) < ((e & (1u << 25)) ? hi : lo);
          anyq |= qual4[j];
        }
      }
      if (!__any(anyq)) continue;
#pragma unroll
      for (uint32_t j = 0; j < 4; j++) {               // (a compaction in between tightens the thresholds; the verdicts taken before it admit a superset)
        const uint32_t e = j == 0 ? ew[u].x : j == 1 ? ew[u].y : j == 2 ? ew[u].z : ew[u].w;
        const uint64_t mm = __ballot(qual4[j]);
        if (mm) {
          if (qual4[j]) W.pool[n + (uint32_t)__popcll(mm & lt)] = ((unsigned long long)iq_key(D, QE_TR(e), QE_TYPE(e), QE_SEQ(e)) << 32) | (QE_TYPE(e) << 28) | ((rb0 + u) * SW_IQ_RB + lane * 4u + j);
          n += (uint32_t)__popcll(mm);
          if (n + 64u > SW_IQ_POOL) { const unsigned long long tc_ = IQCLK_T(); wave_lds_sync(); IQ_COMPACT(false); IQCLK_ADD(1, IQCLK_T() - tc_); IQCLK_ADD(6, 1); }
        }
      }
    }
  }
  wave_lds_sync();
  { const unsigned long long tc_ = IQCLK_T(); IQ_COMPACT(true); IQCLK_ADD(1, IQCLK_T() - tc_); IQCLK_ADD(6, 1); }
#undef IQ_COMPACT
  return n;
}
// one GetBroadcasts(2, limit) over the sorted candidates, W.taken[0, returned) = what it took, in the order it took them.  The walk is
// queue.go's — down the order, take what fits, the space left only shrinks — done 64 candidates at a time: among the entries of a chunk that
// still fit on their own, a prefix sum of their costs says how far the packet takes them in one go; the first one that no longer fits is
// skipped for good (it cannot fit later either) and the walk resumes behind it, with fewer bytes, for the shorter ones.
__device__ __forceinline__ uint32_t iq_pick(DevRef D, const IqWave W, uint32_t n, int limit, int& used_out) {
  const uint32_t lane = sw_lane(); const uint64_t lt = iq_ltmask();
  int used = 0; uint32_t nt = 0; bool full = false;
  for (uint32_t c0 = 0; c0 < n && !full; c0 += 64) {
    const uint32_t idx = c0 + lane;
    const unsigned long long e = idx < n ? W.pool[idx] : ~0ull;
    const bool valid = idx < n && (uint32_t)(e >> 32) != IQ_RETIRED;
    const int len = (int)sel4(D.msg_len, ((uint32_t)e >> 28) & 3u);
    if (!__any(valid)) break;                                       // (retired entries and pads sort last)
    uint32_t pos = 0;
    for (;;) {
      const int free_b = limit - used - 2;
      if (free_b <= 0 || nt >= SW_IQ_PKT) { full = true; break; }
      const bool elig = valid && lane >= pos && len <= free_b;
      const uint64_t me = __ballot(elig);
      if (!me) break;
      int cost = elig ? 2 + len : 0, incl = cost;                  // inclusive prefix sum of the eligible entries' costs
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if (lane >= (uint32_t)off) incl += v; }
      const bool fits = elig && used + incl <= limit;
      const uint64_t mf = __ballot(fits), mbad = me & ~mf;
      // the run ends at the first eligible entry that does not fit; eligible entries behind it are looked at again with what is left
      const uint32_t stop = mbad ? (uint32_t)__ffsll((long long)mbad) - 1u : 64u;
      const uint64_t mtake = stop >= 64u ? mf : (mf & ((1ull << stop) - 1ull));
      uint32_t cnt = (uint32_t)__popcll(mtake);
      if (nt + cnt > SW_IQ_PKT) cnt = SW_IQ_PKT - nt;              // (never binding: the host refuses configurations that could take more)
      if (((mtake >> lane) & 1ull) && (uint32_t)__popcll(mtake & lt) < cnt) W.taken[nt + (uint32_t)__popcll(mtake & lt)] = idx;
      if (cnt) { const uint32_t last = 63u - (uint32_t)__clzll((long long)mtake); used += __shfl(incl, last); nt += cnt; }
      if (stop >= 64u) break;
      pos = stop + 1u;
    }
  }
  wave_lds_sync();
  used_out = used;
  return nt;
}
// the serf delegate's share of the same packet (the user-event queue: <= 32 entries, meta words staged in W.evm): lane 0 picks
__device__ __forceinline__ uint32_t iq_pick_events(DevRef D, const IqWave W, uint32_t evqlen, uint32_t& live_e, int avail, uint32_t rl) {
  if (sw_lane() == 0) {
    int used2 = 0; uint32_t le = live_e;
    const uint32_t te = get_broadcasts(D, LdsMetaQ{W.evm}, evqlen, le, 3, avail, used2, rl);
    W.scal[2] = te; W.scal[3] = le;
  }
  wave_lds_sync();
  live_e = W.scal[3];
  return W.scal[2];
}
// candidate -> {subject, incarnation, from, type << 30}
__device__ __forceinline__ uint4 iq_entry(DevRef D, unsigned long long pe, uint32_t r, uint32_t k, size_t l, size_t NL) {
  const uint32_t src = (uint32_t)pe, type = (src >> 28) & 3u, idx = src & 0x0FFFFFFFu;
  if (src & IQ_EXPL) { const uint4 e = QENT(idx, l); return make_uint4(e.x, e.y, e.z, type << 30); }
  const size_t mi = m_idx(D, r, idx, k);
  const uint32_t a = D.mA[mi], f = D.mF[mi];
  return make_uint4(D.mrow_subj[(size_t)r * D.M + idx], MA_INC(a) + QF_DELTA(f), QF_FROM(f), type << 30);
}
// after the sweep: what was taken has one more transmit, or is Finished(); the order is restored for the next packet
__device__ __forceinline__ void iq_bump(const IqWave W, uint32_t n, uint32_t nt, uint32_t rl, bool resort) {
  const uint32_t lane = sw_lane();
  if (lane < nt) {
    const uint32_t q = W.taken[lane];
    const unsigned long long e = W.pool[q];
    uint32_t key = (uint32_t)(e >> 32);
    key = (key >> 24) + 1u >= rl ? IQ_RETIRED : key + (1u << 24);
    W.pool[q] = ((unsigned long long)key << 32) | (uint32_t)e | IQ_DIRTY;
  }
  wave_lds_sync();
  if (resort && nt) iq_resort(W.pool, n);
}
// the transmit counts back where they live; returns how many implied rumours retired.  W.xq[slot] = the new meta word of a slot's entry
// (0 = untouched, 0xFFFFFFFF = retired) for iq_store_slots
__device__ __forceinline__ uint32_t iq_writeback(DevRef D, const IqWave W, uint32_t n, uint32_t r, uint32_t k) {
  const uint32_t lane = sw_lane();
  if (lane < 32) W.xq[lane] = 0;
  wave_lds_sync();
  uint32_t* col = D.mE + e_col(D, r, k);
  uint32_t retired = 0;
  for (uint32_t c0 = 0; c0 < n; c0 += 64) {
    const uint32_t idx = c0 + lane;
    bool gone = false;
    if (idx < n) {
      const unsigned long long e = W.pool[idx];
      const uint32_t src = (uint32_t)e, key = (uint32_t)(e >> 32), type = (src >> 28) & 3u, at = src & 0x0FFFFFFFu;
      if (src & IQ_DIRTY) {
        const uint32_t seq = 0x3FFFFFu - (key & 0x3FFFFFu);
        if (src & IQ_EXPL) W.xq[at] = key == IQ_RETIRED ? 0xFFFFFFFFu : m_pack(type, key >> 24, seq);
        else { gone = key == IQ_RETIRED; col[(size_t)(at / SW_IQ_RB) * (64u * SW_IQ_RB) + (at % SW_IQ_RB)] = gone ? 0u : QE_PACK(key >> 24, type, seq); }
      }
    }
    retired += (uint32_t)__popcll(__ballot(gone));
  }
  wave_lds_sync();
  return retired;
}
// the slots' entries (memberlist queue_cap slots, then serf's event queue) written back compacted, like the gossip role's write-back
__device__ __forceinline__ uint32_t iq_store_slots(DevRef D, const IqWave W, size_t l, size_t NL, uint32_t qlen) {
  const uint32_t lane = sw_lane(); const uint64_t lt = iq_ltmask();
  const bool have = lane < qlen;
  uint4 e = have ? QENT(lane, l) : make_uint4(0, 0, 0, 0);
  const uint32_t nm = have ? W.xq[lane] : 0u;
  const bool live = have && nm != 0xFFFFFFFFu;
  if (live && nm) e.w = nm;
  const uint64_t mk = __ballot(live);
  const uint32_t pos = (uint32_t)__popcll(mk & lt);
  if (live && (pos != lane || nm)) QENT(pos, l) = e;
  return (uint32_t)__popcll(mk);
}
__device__ __forceinline__ uint32_t iq_store_events(DevRef D, const IqWave W, size_t l, size_t NL, uint32_t evqlen, uint32_t live_e, uint32_t touched) {
  const uint32_t lane = sw_lane(); const uint64_t lt = iq_ltmask();
  const bool have = lane < evqlen;
  uint4 e = have ? D.evq[(size_t)lane * NL + l] : make_uint4(0, 0, 0, 0);
  const bool live = have && ((live_e >> lane) & 1u);
  if (have) e.w = W.evm[lane];
  const uint64_t mk = __ballot(live);
  const uint32_t pos = (uint32_t)__popcll(mk & lt);
  if (live && (pos != lane || ((touched >> lane) & 1u))) D.evq[(size_t)pos * NL + l] = e;
  return (uint32_t)__popcll(mk);
}
__device__ __forceinline__ uint32_t iq_nth_bit(uint32_t m, uint32_t n) { for (uint32_t i = 0; i < n; i++) m &= m - 1; return (uint32_t)__ffs((int)m) - 1u; }

#define SW_IQ_STRIP_WORDS (SW_IQ_POOL * 2u + SW_IQ_PKT + 32u + 32u + 8u + 4u * SW_IQ_PKT)
__device__ __forceinline__ IqWave iq_strip(uint32_t* base) {
  lds_u32* p = (lds_u32*)base + (threadIdx.x / 64u) * SW_IQ_STRIP_WORDS;
  IqWave W; W.pool = (lds_u64*)p; W.taken = p + SW_IQ_POOL * 2u; W.evm = W.taken + SW_IQ_PKT; W.xq = W.evm + 32u; W.scal = W.xq + 32u; W.sent = W.scal + 8u;
  return W;
}

// memberlist gossip() for a handle whose queue is implied by the pair store: the block is the gossip role's stagger chunk (same block index,
// same private edge segment, so k_deliver does not change); the peers are drawn lane per node, then every node with something queued gets
// the whole wave for its GetBroadcasts.  Fan-out <= 4.  MULTI: a packet for a node of another shard goes to that shard's list, unjudged — the receiving
// shard asks the no-op question when the records arrive (SW_EDGE_JUDGE, like the gossip role).
// the no-op question (noop_at_receiver) for a rumour whose subject owns row `row`, with the receiver's pair word `a` already fetched
__device__ __forceinline__ bool iq_noop_pair(DevRef D, uint32_t r, uint32_t row, uint32_t kr, uint32_t a, uint4 e) {
  if (a) {
    const uint32_t type = m_type(e.w), vinc = MA_INC(a), st = MA_STATE(a);
    if (type == SWIM_MSG_ALIVE) return e.y <= vinc;
    if (e.y != vinc) return e.y < vinc;
    if (st == SWIM_STATE_DEAD || st == SWIM_STATE_LEFT) return true;
    if (type == SWIM_MSG_SUSPECT && st == SWIM_STATE_SUSPECT) {
      const uint32_t nc = MA_NCONF(a);
      if (nc >= D.susp_k) return true;
      const size_t idx = m_idx(D, r, row, kr);
      const uint32_t b = D.mB[idx], c = D.mC[idx];
      return M_CONF0(b, c) == e.z || (nc >= 1 && M_CONF1(c) == e.z);
    }
    return false;
  }
  return noop_given_view(D, base_key_of(D, r, e.x, D.nw[(size_t)r * D.N + e.x]), 0, 0, e);
}
#define SW_IQ_GTHREADS 1024u      /* k_gossip_iq: 16 waves per stagger chunk, 16 nodes each (a wave works on ONE node at a time: 4 waves left 3/4 of the device idle) */
template <bool SERF, bool MULTI>
__global__ void __launch_bounds__(SW_IQ_GTHREADS) k_gossip_iq(const SwDev* __restrict__ Dp, uint32_t nb_gossip) {
  SW_DEV_BIND
  __shared__ __attribute__((aligned(8))) uint32_t s_strips[(SW_IQ_GTHREADS / 64) * SW_IQ_STRIP_WORDS];
  __shared__ uint32_t lds_stats[ST_COUNT], lds_exc[2 * SW_EXC_MAX], s_cnt[1];
  __shared__ uint32_t s_peer[SW_BLOCK * 4], s_pw[SW_BLOCK * 4];
  __shared__ uint4 s_hdr[SW_BLOCK];
  __shared__ uint32_t s_node[SW_BLOCK], s_iqn[SW_BLOCK], s_fo[SW_BLOCK];     // node id (NONE: not active), implied rumours queued, found | ok mask << 8
  const uint32_t r = blockIdx.x / nb_gossip, bx = blockIdx.x % nb_gossip, t = *D.tick, lane = sw_lane();
  const size_t NL = (size_t)D.R * D.nloc;
  if (blockIdx.x == 0 && threadIdx.x == 0) *D.ord_n = 0;            // this tick's list of nodes with piggy-back orders starts empty (k_deliver fills it)
  uint32_t fb = NONE;
  if (D.fast_blocks) {
    const uint32_t i_first = map_gossip(D, t % D.G, bx * SW_BLOCK);
    if (i_first == NONE) return;
    fb = (uint32_t)(((size_t)r * D.nloc + (i_first - D.i0)) / SW_BLOCK);
    if (!D.q_any[fb]) {
      if (threadIdx.x == 0) { const uint32_t c = D.alive_cnt[fb]; if (c) atomicAdd(stat_ptr(D, ST_QUIESCENT), (unsigned long long)c); }
      return;
    }
  }
  ExcList X; X.stage(D, r, lds_exc);
  BlockStats S; S.init(lds_stats);
  if (threadIdx.x == 0) s_cnt[0] = 0;
  __syncthreads();
  IQCLK_INIT();
  const IqWave W = iq_strip(s_strips);
  const bool filter = (D.flags & SWIM_F_FILTER_NOOP) != 0;
  bool holds = false;
  // ---- lane per node (the chunk's 256 nodes on the first four waves): who has something queued, and whom it gossips to
  if (threadIdx.x < SW_BLOCK) {
    const uint32_t i = map_gossip(D, t % D.G, bx * SW_BLOCK + threadIdx.x);
    size_t l = 0; uint4 h = make_uint4(0, 0, 0, 0); uint32_t wi = NW_DEAD, iqn = 0;
    if (i != NONE) { l = (size_t)r * D.nloc + (i - D.i0); wi = D.nw[(size_t)r * D.N + i]; h = HDR(l); iqn = D.iqn[l]; }
    const bool something = (h_qlen(h.y) | h_evqlen(h.y) | iqn) != 0;
    const bool acts = i != NONE && !(wi & NW_INERT), active = acts && something;
    uint32_t found = 0, okm = 0;
    if (active) {
      uint32_t peers[4], pw[4];
      found = k_random_nodes<true>(D, r, i, i - D.i0, t, SW_STREAM_GOSSIP, D.k_gossip < 4u ? D.k_gossip : 4u, 0, NONE, peers, pw, X);
      for (uint32_t p = 0; p < found; p++) {
        s_peer[threadIdx.x * 4 + p] = peers[p]; s_pw[threadIdx.x * 4 + p] = pw[p];
        if (reach(D, r, t, wi, pw[p], i, p)) okm |= 1u << p;
      }
    }
    s_node[threadIdx.x] = active ? i : NONE; s_hdr[threadIdx.x] = h; s_iqn[threadIdx.x] = iqn; s_fo[threadIdx.x] = found | (okm << 8);
    S.count(ST_QUIESCENT, acts && !something); S.count(ST_ACTIVE, active);
    holds = i != NONE && (wi & NW_INERT) && something;             // a node that is not running keeps its (frozen) queue: the block's hint stays up
  }
  __syncthreads();
  uint32_t c_pkt = 0, c_drop = 0, c_filt = 0, c_s0 = 0, c_s1 = 0, c_s2 = 0, c_s3 = 0, c_remote = 0, c_e0 = 0;      // (tallied on lane 0)
  // ---- wave per node: wave w takes the chunk's nodes 16 w .. 16 w + 15
  for (uint32_t tj = (threadIdx.x / 64u) * 16u; tj < (threadIdx.x / 64u) * 16u + 16u; tj++) {
    const uint32_t o = s_node[tj];
    if (o == NONE) continue;
    const uint32_t k = o - D.i0, n_found = s_fo[tj] & 0xFFu, ok_j = s_fo[tj] >> 8;
    const size_t lj = (size_t)r * D.nloc + k;
    const uint4 hj = s_hdr[tj];
    const uint32_t qlen = h_qlen(hj.y), evqlen = SERF ? h_evqlen(hj.y) : 0u; uint32_t iq_j = s_iqn[tj];
    const uint32_t rl = D.retransmit_limit;
    uint32_t live_e = evqlen >= 32 ? 0xFFFFFFFFu : (1u << evqlen) - 1u, touched_e = 0;
    if (SERF && lane < evqlen) W.evm[lane] = D.evq[(size_t)lane * NL + lj].w;
    const unsigned long long tq0_ = IQCLK_T();
    uint32_t n = iq_build(D, W, r, k, lj, NL, qlen, iq_j, n_found);
    const unsigned long long tq1_ = IQCLK_T();
    bool any_taken = false;
    // every packet's GetBroadcasts first — nothing they decide waits for memory — then ONE round of loads for all they send (a packet at a
    // time it was three dependent round trips per packet: the entry, the subject's words, the receiver's view)
    uint32_t nt_p[4] = { 0, 0, 0, 0 }, te_p[4] = { 0, 0, 0, 0 }, npk = 0;
    for (uint32_t p = 0; p < n_found; p++) {
      int used = 0;
      const uint32_t nt = iq_pick(D, W, n, (int)D.budget, used);
      uint32_t te = 0;
      const int avail = (int)D.budget - used;
      if (SERF && avail > 2 + 1) te = iq_pick_events(D, W, evqlen, live_e, avail, rl);
      if (!nt && !te) break;                         // "if len(msgs) == 0 { return }"
      touched_e |= te; any_taken |= nt != 0;
      const uint32_t src = lane < nt ? (uint32_t)W.pool[W.taken[lane]] : 0u, ty = (src >> 28) & 3u;
      if (lane < nt) W.sent[p * SW_IQ_PKT + lane] = src;
      {
        const uint64_t b0 = __ballot(lane < nt && ty == SWIM_MSG_ALIVE), b1 = __ballot(lane < nt && ty == SWIM_MSG_SUSPECT), b2 = __ballot(lane < nt && ty == SWIM_MSG_DEAD);
        c_pkt++; c_s0 += (uint32_t)__popcll(b0); c_s1 += (uint32_t)__popcll(b1); c_s2 += (uint32_t)__popcll(b2); c_s3 += (uint32_t)__popc(te);
      }
      if (!((ok_j >> p) & 1u)) c_drop++;
      nt_p[p & 3u] = nt; te_p[p & 3u] = te; npk = p + 1;
      iq_bump(W, n, nt, rl, p + 1 < n_found);
    }
    wave_lds_sync();
    const unsigned long long tq2_ = IQCLK_T();
    {
      uint32_t x_[4], own_a[4], own_f[4], rcv_a[4], src_[4]; uint4 ex_[4]; bool on_[4];
#pragma unroll
      for (uint32_t p = 0; p < 4; p++) {             // the loads: all packets' entries and the receivers' pair words, independent of each other
        on_[p] = p < npk && ((ok_j >> p) & 1u) && lane < nt_p[p];
        src_[p] = on_[p] ? W.sent[p * SW_IQ_PKT + lane] : 0u;
        x_[p] = 0; own_a[p] = 0; own_f[p] = 0; rcv_a[p] = 0; ex_[p] = make_uint4(0, 0, 0, 0);
        if (on_[p]) {
          const uint32_t at = src_[p] & 0x0FFFFFFFu;
          if (src_[p] & IQ_EXPL) ex_[p] = QENT(at, lj);
          else {
            const size_t mi = m_idx(D, r, at, k);
            x_[p] = D.mrow_subj[(size_t)r * D.M + at]; own_a[p] = D.mA[mi]; own_f[p] = D.mF[mi];
            if (!MULTI || s_peer[tj * 4 + p] / D.nloc == D.rank) rcv_a[p] = D.mA[m_idx(D, r, at, s_peer[tj * 4 + p] - D.i0)];
          }
        }
      }
#pragma unroll
      for (uint32_t p = 0; p < 4; p++) {
        if (p >= npk || !((ok_j >> p) & 1u)) continue;
        const uint32_t peer = s_peer[tj * 4 + p], pwp = s_pw[tj * 4 + p], gdst = r * D.N + peer, te = te_p[p];
        const bool expl = (src_[p] & IQ_EXPL) != 0;
        const uint4 e4 = expl ? make_uint4(ex_[p].x, ex_[p].y, ex_[p].z, ((src_[p] >> 28) & 3u) << 30)
                              : make_uint4(x_[p], MA_INC(own_a[p]) + QF_DELTA(own_f[p]), QF_FROM(own_f[p]), ((src_[p] >> 28) & 3u) << 30);
        bool keep = on_[p];
        const uint32_t psh = MULTI ? peer / D.nloc : D.rank;
        uint4 ev = make_uint4(0, 0, 0, 0); const bool evl = SERF && lane < (uint32_t)__popc(te);
        if (evl) ev = D.evq[(size_t)iq_nth_bit(te, lane) * NL + lj];
        if (MULTI && psh != D.rank) {               // another shard's node: its shard judges (a rumour about the receiver itself is always delivered)
          const uint32_t jd = (filter && e4.x != peer) ? SW_EDGE_JUDGE : 0u;
          wave_append(D, psh, keep, make_uint4(gdst, e4.x, e4.y, (e4.w & 0xC0000000u) | jd | (e4.z & TB_FROM_MASK)));
          wave_append(D, psh, evl, make_uint4(gdst, ev.x, ev.y, (uint32_t)SWIM_MSG_USER << 30));
          const uint32_t nrec = (uint32_t)__popcll(__ballot(keep)) + (uint32_t)__popc(te);
          c_remote += nrec; c_e0 += (uint32_t)__popcll(__ballot(keep && !jd)) + (uint32_t)__popc(te);
          continue;
        }
        if (keep && filter && e4.x != peer) {
          if (expl) keep = !noop_at_receiver<true>(D, r, (size_t)r * D.nloc + (peer - D.i0), D.nw[(size_t)r * D.N + e4.x], e4, false, e4);
          else keep = !iq_noop_pair(D, r, src_[p] & 0x0FFFFFFFu, peer - D.i0, rcv_a[p], e4);
        }
        c_filt += (uint32_t)__popcll(__ballot(on_[p] && !keep));
        if (pwp & NW_ATTACHED) {                     // Transport.WriteTo towards the real node
          if (keep) capture(D, o, gdst, e4.x, e4.y, (e4.w & 0xC0000000u) | (e4.z & 0x3FFFFFFFu));
          if (evl) capture(D, o, gdst, ev.x, ev.y, (uint32_t)SWIM_MSG_USER << 30);
        } else {
          const uint64_t mk = __ballot(keep);
          const uint32_t nk = (uint32_t)__popcll(mk), total = nk + (uint32_t)__popc(te);
          if (total) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&s_cnt[0], total);
            base = __shfl(base, 0);
            uint4* seg = D.seg + (size_t)(r * D.nb_gossip + bx) * D.seg_cap;
            if (base + total > D.seg_cap) { if (lane == 0) atomicOr(D.err, SW_ERR_EDGE_OVF); }
            else {
              if (keep) seg[base + (uint32_t)__popcll(mk & iq_ltmask())] = make_uint4(gdst, e4.x, e4.y, (e4.w & 0xC0000000u) | (e4.z & 0x3FFFFFFFu));
              if (evl) seg[base + nk + lane] = make_uint4(gdst, ev.x, ev.y, (uint32_t)SWIM_MSG_USER << 30);
            }
          }
        }
      }
    }
    // ---- the queues back where they live
    const unsigned long long tq3_ = IQCLK_T();
    uint32_t nq = qlen, ne = evqlen;
    if (any_taken) {
      const uint32_t retired = iq_writeback(D, W, n, r, k);
      iq_j -= retired;
      nq = iq_store_slots(D, W, lj, NL, qlen);
      if (lane == 0 && retired) D.iqn[lj] = iq_j;
    }
    if (SERF && touched_e) ne = iq_store_events(D, W, lj, NL, evqlen, live_e, touched_e);
    if (lane == 0) {
      const uint32_t hy = h_pack(h_leaving(hj.y), nq, SERF ? ne : h_evqlen(hj.y));
      if (hy != hj.y) { uint4 hn = hj; hn.y = hy; HDR(lj) = hn; }
      if (!(nq | ne | iq_j)) q_bit_lane(D, lj, false, true);
    }
    holds |= (nq | ne | iq_j) != 0;
    wave_lds_sync();
    IQCLK_ADD(0, tq1_ - tq0_); IQCLK_ADD(2, tq2_ - tq1_); IQCLK_ADD(3, tq3_ - tq2_); IQCLK_ADD(4, IQCLK_T() - tq3_); IQCLK_ADD(5, 1);
  }
  if (lane == 0) {
    S.add(ST_PKT_SENT, c_pkt); S.add(ST_PKT_DROP, c_drop); S.add(ST_FILTERED, c_filt);
    S.add(ST_SENT0, c_s0); S.add(ST_SENT1, c_s1); S.add(ST_SENT2, c_s2); S.add(ST_SENT3, c_s3);
    if (MULTI) { S.add(ST_EDGES_REMOTE, c_remote); S.add(ST_EDGES, c_e0); if (c_remote) *D.act = 1; }
  }
  const int any = __syncthreads_or(holds);
  if (threadIdx.x == 0) {
    if (fb != NONE && !any) D.q_any[fb] = 0;
    const uint32_t c = s_cnt[0];
    if (c) { D.seg_cnt[r * D.nb_gossip + bx] = c; atomicAdd(stat_ptr(D, ST_EDGES), (unsigned long long)c); if (MULTI) *D.act = 1; }
  }
  IQCLK_FLUSH();
  S.flush(D);
}

// sendMsg's getBroadcasts for the pings / indirect pings / acks / nacks of this tick, on a handle whose queue is implied by the pair store: the
// orders k_deliver filed per node (canonical order: kind, receiver, prober; a duplicate once), SW_IQ_ORDERS of them per scan of the node's
// column, a wave per node.  What is picked goes to the node block's carry area and arrives with the next tick's packets (k_deliver), exactly
// like NodeCtxT::piggyback's picks.  Runs between k_deliver and k_resolve: the queue is as the gossip launch left it.
template <bool SERF>
__global__ void __launch_bounds__(SW_BLOCK) k_piggy_iq(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  __shared__ __attribute__((aligned(8))) uint32_t s_strips[(SW_BLOCK / 64) * SW_IQ_STRIP_WORDS];
  __shared__ uint32_t lds_stats[ST_COUNT];
  __shared__ uint2 s_ord[(SW_BLOCK / 64) * 64];
  const uint32_t lane = sw_lane(), wv = threadIdx.x / 64, t = *D.tick;
  const size_t NL = (size_t)D.R * D.nloc;
  BlockStats S; S.init(lds_stats);
  IQCLK_INIT();                                    // (its scans tally into the workgroup's words too; only k_gossip_iq's are reported)
  const IqWave W = iq_strip(s_strips);
  uint2* const so = s_ord + wv * 64;
  const uint32_t n_nodes = *D.ord_n;
  uint32_t c_pig = 0, c_msgs = 0, c_s0 = 0, c_s1 = 0, c_s2 = 0, c_s3 = 0, c_peak = 0; bool stamped = false;
  for (uint32_t a = blockIdx.x * (SW_BLOCK / 64) + wv; a < n_nodes; a += gridDim.x * (SW_BLOCK / 64)) {
    const size_t l = D.ord_nodes[a];
    const uint32_t r = div_nloc(D, l), k = mod_nloc(D, l), o = D.i0 + k, nb0 = (uint32_t)(l / SW_BLOCK);
    uint32_t no = D.ord_cnt[l];
    { const uint32_t pk = D.in_cnt[l] + no; c_peak = pk > c_peak ? pk : c_peak; }
    if (no > D.ord_cap) no = D.ord_cap;
    if (no > 64u) no = 64u;
    // canonical order (edge_cmp among orders: kind, receiver, prober), by rank; duplicates once
    const uint2 mine = lane < no ? D.ord[l * D.ord_cap + lane] : make_uint2(0, 0);
    so[lane] = mine;
    wave_lds_sync();
    uint32_t rank = 0; bool dup = false;
    if (lane < no)
      for (uint32_t q = 0; q < no; q++) {
        const uint2 b = so[q];
        const bool less = (b.y >> 30) != (mine.y >> 30) ? (b.y >> 30) < (mine.y >> 30) : b.x != mine.x ? b.x < mine.x : b.y < mine.y;
        const bool same = b.x == mine.x && b.y == mine.y;
        rank += less || (same && q < lane); dup |= same && q < lane;
      }
    wave_lds_sync();
    if (lane < no) so[rank] = dup ? make_uint2(NONE, NONE) : mine;      // (a duplicate sorts right behind its original and is skipped)
    wave_lds_sync();
    if (lane == 0) D.ord_cnt[l] = 0;
    for (uint32_t o0 = 0; o0 < no; o0 += SW_IQ_ORDERS) {
      const uint32_t nbatch = no - o0 < SW_IQ_ORDERS ? no - o0 : SW_IQ_ORDERS;
      const uint4 hj = HDR(l);
      const uint32_t qlen = h_qlen(hj.y), evqlen = SERF ? h_evqlen(hj.y) : 0u; uint32_t iq_j = D.iqn[l];
      const uint32_t rl = D.retransmit_limit;
      uint32_t live_e = evqlen >= 32 ? 0xFFFFFFFFu : (1u << evqlen) - 1u, touched_e = 0;
      if (SERF && lane < evqlen) W.evm[lane] = D.evq[(size_t)lane * NL + l].w;
      const uint32_t n = iq_build(D, W, r, k, l, NL, qlen, iq_j, nbatch);
      bool any_taken = false;
      for (uint32_t p = 0; p < nbatch; p++) {
        const uint2 od = so[o0 + p];
        if (od.x == NONE && od.y == NONE) continue;                   // a duplicate
        const uint32_t receiver = od.x, kind = od.y >> 30;
        const int limit = (int)D.budget - (int)sel4(D.ctl_len, kind & 3u);
        int used = 0;
        const uint32_t nt = iq_pick(D, W, n, limit, used);
        uint32_t te = 0;
        const int avail = limit - used;
        if (SERF && D.EQ && avail > 2 + 1) te = iq_pick_events(D, W, evqlen, live_e, avail, rl);
        if (!(nt | te)) continue;
        touched_e |= te; any_taken |= nt != 0;
        uint4 e4 = make_uint4(0, 0, 0, 0);
        if (lane < nt) e4 = iq_entry(D, W.pool[W.taken[lane]], r, k, l, NL);
        {
          const uint32_t ty = e4.w >> 30;
          const uint64_t b0 = __ballot(lane < nt && ty == SWIM_MSG_ALIVE), b1 = __ballot(lane < nt && ty == SWIM_MSG_SUSPECT), b2 = __ballot(lane < nt && ty == SWIM_MSG_DEAD);
          c_pig++; c_msgs += nt + (uint32_t)__popc(te);
          c_s0 += (uint32_t)__popcll(b0); c_s1 += (uint32_t)__popcll(b1); c_s2 += (uint32_t)__popcll(b2); c_s3 += (uint32_t)__popc(te);
        }
        if (receiver != NONE) {
          const uint32_t gdst = r * D.N + receiver, cnt = nt + (uint32_t)__popc(te);
          const bool att = *D.att_any && (D.nw[gdst] & NW_ATTACHED);   // Transport.WriteTo towards the real node
          uint4 ev = make_uint4(0, 0, 0, 0); const bool evl = SERF && lane < (uint32_t)__popc(te);
          if (evl) ev = D.evq[(size_t)iq_nth_bit(te, lane) * NL + l];
          if (att) {
            if (lane < nt) capture(D, o, gdst, e4.x, e4.y, (e4.w & 0xC0000000u) | (e4.z & 0x3FFFFFFFu));
            if (evl) capture(D, o, gdst, ev.x, ev.y, (uint32_t)SWIM_MSG_USER << 30);
          } else {
            uint32_t pos = 0;
            if (lane == 0) pos = atomicAdd((uint32_t*)&D.carry_cl[nb0], cnt);
            pos = __shfl(pos, 0);
            uint4* area = D.carry + ((size_t)((t + 1) & 1u) * D.NB + nb0) * D.carry_cap;
            if (pos + cnt > D.carry_cap) { if (lane == 0) atomicOr(D.err, SW_ERR_CARRY_OVF); }
            else {
              if (lane < nt) area[pos + lane] = make_uint4(gdst, e4.x, e4.y, (e4.w & 0xC0000000u) | (e4.z & 0x3FFFFFFFu));
              if (evl) area[pos + nt + lane] = make_uint4(gdst, ev.x, ev.y, (uint32_t)SWIM_MSG_USER << 30);
            }
            stamped = true;
          }
        }
        iq_bump(W, n, nt, rl, p + 1 < nbatch);
      }
      uint32_t nq = qlen, ne = evqlen;
      if (any_taken) {
        const uint32_t retired = iq_writeback(D, W, n, r, k);
        iq_j -= retired;
        nq = iq_store_slots(D, W, l, NL, qlen);
        if (lane == 0 && retired) D.iqn[l] = iq_j;
      }
      if (SERF && touched_e) ne = iq_store_events(D, W, l, NL, evqlen, live_e, touched_e);
      if (lane == 0) {
        const uint32_t hy = h_pack(h_leaving(hj.y), nq, SERF ? ne : h_evqlen(hj.y));
        if (hy != hj.y) { uint4 hn = hj; hn.y = hy; HDR(l) = hn; }
        if (!(nq | ne | iq_j)) q_bit_lane(D, l, false, true);
      }
      wave_lds_sync();
      // the next batch of the SAME wave reads the header and the slots this one wrote — other lanes of it: a workgroup-scope fence, and only when there is
      // a next batch (a node with more than four orders: rare).  (Until call 31 of round 6 a __threadfence() stood here, for every node: a device-scope
      // release writes the L2's dirty lines back, and the kernel ran at the same 15 M nodes/s whatever its instruction count, occupancy or grid.)
      if (o0 + SW_IQ_ORDERS < no) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
  }
  if (lane == 0) {
    if (c_peak > 5u && c_peak > *D.peak) atomicMax(D.peak, c_peak);
    if (stamped) *D.carry_stamp = t + 1;
    S.add(ST_PIGGY, c_pig); S.add(ST_PIGGY_MSGS, c_msgs);
    S.add(ST_SENT0, c_s0); S.add(ST_SENT1, c_s1); S.add(ST_SENT2, c_s2); S.add(ST_SENT3, c_s3);
  }
  S.flush(D);
}

// fold ticks: a subject some node of the shard (running or not) still has a rumour queued about is not folded (the row would take the
// rumour with it) — the same rule in the checker (q_cnt).  A workgroup per (replica, 64-row block) reads the block's queue words coalesced.
__global__ void __launch_bounds__(SW_BLOCK) k_fold_scan_iq(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  __shared__ uint32_t s_busy[SW_IQ_RB];
  const uint32_t r = blockIdx.x / D.MB, rb = blockIdx.x % D.MB, G = (D.nloc + 63u) >> 6;
  for (uint32_t w = threadIdx.x; w < SW_IQ_RB; w += SW_BLOCK) s_busy[w] = 0;
  __syncthreads();
  for (uint32_t g = 0; g < G; g++) {
    const uint32_t* reg = D.mE + (((size_t)r * G + g) * D.MB + rb) * (64u * SW_IQ_RB);
    for (uint32_t w = threadIdx.x; w < 64u * SW_IQ_RB; w += SW_BLOCK) if (reg[w] & QE_QUEUED) s_busy[w % SW_IQ_RB] = 1;
  }
  __syncthreads();
  for (uint32_t w = threadIdx.x; w < SW_IQ_RB; w += SW_BLOCK) if (s_busy[w]) {
    const uint32_t row = rb * SW_IQ_RB + w;
    if (row < D.M) { const uint32_t x = D.mrow_subj[(size_t)r * D.M + row]; if (x != NONE) D.fl_bad[(size_t)r * D.N + x] = 1; }
  }
}
// ... and the rumours in the nodes' queue_cap slots (a node's rumour about itself, rumours about subjects without a row) hold their subjects back
// the same way: the checker's q_cnt counts every queued rumour, wherever the device keeps it (found by tools/fuzz_parity.py --unbounded: 7 of 200
// cases folded a subject whose own refutation was still queued in its slots)
__global__ void __launch_bounds__(SW_BLOCK) k_fold_scan_slots(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  const size_t NL = (size_t)D.R * D.nloc, l = (size_t)blockIdx.x * SW_BLOCK + threadIdx.x;
  if (l >= NL) return;
  const uint32_t n = h_qlen(HDR(l).y), r = div_nloc(D, l);
  for (uint32_t j = 0; j < n; j++) { const uint32_t x = QENT(j, l).x; if (x < D.N) D.fl_bad[(size_t)r * D.N + x] = 1; }
}
// pooled inbox rows (swim_device.h: inbox_big): every node with deferred arrivals gets a big row — one CAS winner per node allocates, the
// others do nothing (no lane ever waits for another); the kernel boundary publishes the rows to k_inbox_file
__global__ void __launch_bounds__(SW_BLOCK) k_inbox_claim(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  const uint32_t n = *D.defer_n < D.defer_cap ? *D.defer_n : D.defer_cap;
  for (uint32_t e = blockIdx.x * SW_BLOCK + threadIdx.x; e < n; e += gridDim.x * SW_BLOCK) {
    const uint32_t l = D.defer_l[e];
    if (D.big_row[l] != NONE) continue;
    if (atomicCAS(&D.big_row[l], NONE, SW_BIGROW_CLAIM) != NONE) continue;
    const uint32_t idx = atomicAdd(D.big_n, 1u);
    if (idx < D.PB) { D.big_list[idx] = l; D.big_row[l] = idx; }
    else { D.big_row[l] = SW_BIGROW_NONE_LEFT; atomicOr(D.err, SW_ERR_INBOX_OVF); }
  }
}
// ... the deferred records into the rows, and what the nodes' own rows already hold copied over (same positions)
__global__ void __launch_bounds__(SW_BLOCK) k_inbox_file(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  const uint32_t n = *D.defer_n < D.defer_cap ? *D.defer_n : D.defer_cap;
  if (!n) return;
  for (uint32_t e = blockIdx.x * SW_BLOCK + threadIdx.x; e < n; e += gridDim.x * SW_BLOCK) {
    const uint4 m = D.defer_rec[e]; const uint32_t br = D.big_row[D.defer_l[e]];
    if (br >= D.PB) continue;
    uint32_t* d = D.inbox_big + ((size_t)br * D.C2 + (m.w - SW_INBOX_FAST)) * 3;
    d[0] = m.x; d[1] = m.y; d[2] = m.z;
  }
  const uint32_t nb = *D.big_n < D.PB ? *D.big_n : D.PB, words = (D.C1 - SW_INBOX_FAST) * 3;
  for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
    const uint32_t* src = D.inbox2 + (size_t)D.big_list[b] * D.C1 * 3; uint32_t* dst = D.inbox_big + (size_t)b * D.C2 * 3;
    for (uint32_t w = threadIdx.x; w < words; w += SW_BLOCK) dst[w] = src[w];
  }
}

// swim_kernels.hip — hand-written gfx950 kernels for the memberlist/serf SWIM hot path.
//
// One tick = five launches.  Everything is integer/byte work bounded by HBM bandwidth,
// random-access sector traffic and atomic throughput; there is no dense contraction, hence no MFMA.
//
//   k_begin    fused, role by block range:
//                expire   suspicion timers that ran out         -> self-addressed dead{} records
//                pending  indirect-ping stage of probes whose direct ping failed ProbeTimeout ago
//                probe    probe()/probeNode for the probe-due set -> suspect{} records, slot requests,
//                         piggy-back orders for the ping and the ack (sendMsg)
//                gossip   kRandomNodes + GetBroadcasts per peer   -> edge lists bucketed by shard
//   k_deliver  edge list -> per-node inbox rows (one returning atomic + one 16 B store per record)
//   k_resolve  per observer: canonical order, aliveNode/suspectNode/deadNode/handleUserEvent
//   k_census   per dirty subject: how the live observers see it
//   k_finish   first-suspect/first-dead/all-dead stamps, trace row, list recycling, tick++
#include "swim_device.h"

#define NONE 0xFFFFFFFFu

__device__ __forceinline__ uint32_t sw_lane() { return __lane_id(); }

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

// observer (r, local k) looking at the node whose word is `w`; base view unless it owns a slot
__device__ __forceinline__ uint32_t view_of(DevRef D, uint32_t r, uint32_t k, uint32_t w, uint32_t* since) {
  if (!NW_HAS_SLOT(w)) { *since = 0; return SW_BASE_KEY; }
  size_t ci = ((size_t)r * D.S + NW_SLOT(w)) * D.nloc + k;
  uint4 a = D.va[ci];
  *since = a.y;
  return a.x;
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
struct BlockStats {
  uint32_t* s;
  __device__ void init(uint32_t* lds) {
    s = lds;
    for (uint32_t i = threadIdx.x; i < ST_COUNT; i += blockDim.x) s[i] = 0;
    __syncthreads();
  }
  __device__ __forceinline__ void add(int i, uint32_t v = 1) { atomicAdd(&s[i], v); }
  // converged call sites: one LDS atomic per wave instead of one per lane
  // converged call sites: wave-reduce a per-lane tally, one LDS atomic per wave
  __device__ __forceinline__ void wave_add(int i, uint32_t v) {
    if (!__any(v != 0)) return;
    for (int off = 32; off; off >>= 1) v += __shfl_down(v, off);
    if (sw_lane() == 0) atomicAdd(&s[i], v);
  }
  __device__ __forceinline__ void count(int i, bool pred) {
    uint64_t m = __ballot(pred);
    if (m && sw_lane() == 0) atomicAdd(&s[i], (uint32_t)__popcll(m));
  }
  __device__ void flush(DevRef D) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < ST_COUNT; i += blockDim.x)
      if (s[i]) atomicAdd(stat_ptr(D, i), (unsigned long long)s[i]);
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
// the destination shard differs per lane: one aggregated append per shard present in the wave
__device__ __forceinline__ void wave_append_sharded(DevRef D, bool want, uint32_t sh, uint4 rec) {
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
  for (int j = 0; j < 4; j++) x4[j] = d.get(j) % D.N;
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
    else { x = d.get((uint32_t)i) % D.N; w = X.usable() ? X.word(x) : nw[x]; }
    if (x == o) continue;
    uint32_t since, key = view_of(D, r, k_local, w, &since), st = SW_KST(key);
    if (mode == 0) {
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
__device__ __forceinline__ void role_expire(DevRef D, uint32_t b, uint32_t nb) {
  uint32_t per = nb / (D.R * D.S);                 // blocks per slot
  uint32_t sidx = b / per, part = b % per, r = sidx / D.S, sl = sidx % D.S;
  if (sl >= D.n_slots[r]) return;
  uint32_t t = *D.tick, now = now_ms(D, t);
  if (!D.slot_susp[sidx] || now < D.slot_mindl[sidx]) return;
  uint32_t x = D.subj_node[sidx], fired = 0;
  const uint32_t* nw = D.nw + (size_t)r * D.N;
  // the scan is latency bound: four independent rows in flight per trip
  for (uint32_t k0 = part * SW_BLOCK; k0 < D.nloc; k0 += 4 * per * SW_BLOCK) {
    uint32_t kk[4]; uint4 v[4]; bool f[4]; bool any_f = false;
#pragma unroll
    for (int j = 0; j < 4; j++) { kk[j] = k0 + j * per * SW_BLOCK + threadIdx.x; v[j] = kk[j] < D.nloc ? D.va[(size_t)sidx * D.nloc + kk[j]] : make_uint4(0, 0, 0, 0); }
#pragma unroll
    for (int j = 0; j < 4; j++) { f[j] = kk[j] < D.nloc && SW_KST(v[j].x) == SWIM_STATE_SUSPECT && now >= v[j].y + sel8(D.susp_timeout, v[j].z & 7u); any_f |= f[j]; }
    if (__any(any_f)) {                             // rare: only now look at liveness and append
#pragma unroll
      for (int j = 0; j < 4; j++) {
        f[j] = f[j] && !(nw[D.i0 + kk[j]] & NW_INERT);
        // a timer is not a packet: the verdict goes straight into the node's own inbox line.  (When a whole
        // cluster's suspicion of one node runs out within a few ticks, appending these to a shared list cost
        // thousands of same-address atomics — the launch's long pole, profiles/r01_role_clock.txt.)
        if (f[j]) {
          size_t l = (size_t)r * D.nloc + kk[j];
          inbox_place(D, mk_edge(D, r, D.i0 + kk[j], x, SW_KINC(v[j].x), SWIM_MSG_DEAD, D.i0 + kk[j]), l, atomicAdd(&D.in_cnt[l], 1u));
        }
        fired += (uint32_t)f[j];
      }
    }
  }
  if (__any(fired != 0)) {
    for (int off = 32; off; off >>= 1) fired += __shfl_down(fired, off);
    if (sw_lane() == 0) { atomicAdd(stat_ptr(D, ST_TIMEOUTS), (unsigned long long)fired); atomicAdd(stat_ptr(D, ST_EDGES), (unsigned long long)fired); }
  }
}

// =================================================================================================
// role: pending — ProbeTimeout after a failed direct ping: indirectPingReq to IndirectChecks random
// alive peers; each relays the target's ack or (Lifeguard) answers nack one ProbeTimeout later.
// =================================================================================================
template <int KMAX>
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
      uint32_t np = k_random_nodes(D, r, i, k, t, SW_STREAM_INDIRECT, D.k_indirect, 1, x, peers, pw, none);
      uint32_t expected = 0, nacks = 0; bool acked = false;
      bool nack_in_time = 2 * D.TQ < p0.z - p0.w;
      for (uint32_t q = 0; q < np; q++) {
        if (D.flags & SWIM_F_NACK) expected++;
        const uint32_t hq = peers[q];
        bool there = reach(D, r, t, wi, pw[q], i, 20 + 4 * q);
        // rare path (a direct ping just failed): plain per-lane appends are fine
        if (piggy_hint(D, r, i, peer_active)) wave_append_sharded(D, true, i / D.nloc, piggy_rec(D, r, i, there ? hq : NONE, SWIM_CTL_INDIRECT, i));
        if (!there) continue;
        bool hx = reach(D, r, t, pw[q], wx, i, 21 + 4 * q), xh = hx && reach(D, r, t, wx, pw[q], i, 22 + 4 * q), ok = hx && xh;
        bool back = reach(D, r, t, pw[q], wi, i, 23 + 4 * q);
        if (piggy_hint(D, r, hq, peer_active)) {
          wave_append_sharded(D, true, hq / D.nloc, piggy_rec(D, r, hq, hx ? x : NONE, SWIM_CTL_PING, i));
          if (ok) wave_append_sharded(D, true, hq / D.nloc, piggy_rec(D, r, hq, back ? i : NONE, SWIM_CTL_ACK, i));
          else if ((D.flags & SWIM_F_NACK) && nack_in_time) wave_append_sharded(D, true, hq / D.nloc, piggy_rec(D, r, hq, back ? i : NONE, SWIM_CTL_NACK, i));
        }
        if (hx && piggy_hint(D, r, x, peer_active)) wave_append_sharded(D, true, x / D.nloc, piggy_rec(D, r, x, xh ? hq : NONE, SWIM_CTL_ACK, i));
        if (ok && back) acked = true;
        else if (!ok && back && nack_in_time) nacks++;
      }
      uint32_t aw = p_aw(h.y);
      // probeNode's TCP fallback ping next to the indirect probes: TCP rides out packet loss, so it reaches every
      // running node of the same partition
      const bool tcp = !acked && (D.flags & SWIM_F_TCP_FALLBACK) && !(wx & NW_DEAD) && NW_PART(wi) == NW_PART(wx);
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
template <bool MULTI>
__device__ __forceinline__ void role_probe(DevRef D, uint32_t r, uint32_t pb, uint32_t a, uint32_t* lds_stats, uint32_t* lds_exc, uint32_t* s_cnt, uint32_t peer_active) {
  ExcList X; X.stage(D, r, lds_exc);
  if (threadIdx.x == 0) s_cnt[0] = 0;
  BlockStats S; S.init(lds_stats);
  uint32_t t = *D.tick;
  uint32_t i = map_probe(D, t % D.P, a);
  const uint32_t* nw = D.nw + (size_t)r * D.N;
  bool e_buddy = false, e_self = false, e_ctrl = false, c_probe = false, c_ack = false, e_pend = false;
  bool o_ping = false, o_ack = false;                // piggy-back orders: for my ping, for the target's ack
  uint32_t o_ping_rcv = NONE, o_ack_rcv = NONE, o_x = 0;
  uint4 rec_buddy = make_uint4(0, 0, 0, 0), rec_self = rec_buddy; uint32_t ctrl_x = 0, buddy_sh = 0;
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
        if (!NW_HAS_SLOT(X.usable() ? X.word(p0.x) : nw[p0.x])) { e_ctrl = true; ctrl_x = p0.x; }
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
        wx = X.usable() ? X.word(c) : nw[c]; key = view_of(D, r, k, wx, &since);
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
        if (ack) { aw = awareness_apply(D, aw, -1); c_ack = true; }
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
  if (__any(e_ctrl)) {
    uint4 c = make_uint4(NONE, ctrl_x, r, 0);
    for (uint32_t sh = 0; sh < D.n_shards; sh++) wave_append(D, sh, e_ctrl, c);
  }
  if (__any(e_self)) wave_append(D, D.rank, e_self, rec_self);
  if (__any(e_buddy)) wave_append_sharded(D, e_buddy, buddy_sh, rec_buddy);
  uint32_t ne = (uint32_t)e_self + (uint32_t)e_buddy;
  if (ne) { S.add(ST_EDGES, ne); if (e_buddy && buddy_sh != D.rank) S.add(ST_EDGES_REMOTE); }
  // piggy-back orders: during dissemination every probing lane files one or two, so the ones that stay on
  // this shard go to the block's private segment (wave prefix sum, one LDS atomic per wave, no global atomic)
  if (D.flags & SWIM_F_PIGGYBACK) {
    bool ack_local = o_ack && (!MULTI || o_x / D.nloc == D.rank);
    uint32_t n_loc = (uint32_t)o_ping + (uint32_t)ack_local, incl = n_loc, my_off = 0;
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
    if (MULTI) { bool rem = o_ack && !ack_local; if (__any(rem)) wave_append_sharded(D, rem, o_x / D.nloc, piggy_rec(D, r, o_x, o_ack_rcv, SWIM_CTL_ACK, i)); }
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt[0]) { D.seg_cnt[D.R * D.nb_gossip + r * D.nb_probe + pb] = s_cnt[0]; if (MULTI) *D.act = 1; }
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
// `a` = the receiver's va record of the subject, `e` = the queue entry {subject, inc, from, meta}.
__device__ __forceinline__ bool noop_given_view(DevRef D, uint4 a, size_t ci, uint4 e) {
  uint32_t type = m_type(e.w), key = a.x, vinc = SW_KINC(key), st = SW_KST(key);
  if (type == SWIM_MSG_ALIVE) return e.y <= vinc;
  if (e.y != vinc) return e.y < vinc;
  if (st == SWIM_STATE_DEAD || st == SWIM_STATE_LEFT) return true;
  if (type == SWIM_MSG_SUSPECT && st == SWIM_STATE_SUSPECT) {
    uint32_t nc = a.z;
    if (nc >= D.susp_k || a.w == e.z) return true;
    if (nc == 0) return false;
    uint4 b = D.vb[ci];
    return b.x == e.z || (nc >= 2 && b.y == e.z) || (nc >= 3 && b.z == e.z);
  }
  return false;
}

// where a node's queue lives: staged in LDS (gossip role: entry j of this lane at sq[j*256]) or in HBM
// (k_resolve: entry j of lane l at q[j*NL + l])
struct LdsQ { uint4* p; __device__ __forceinline__ uint32_t& meta(uint32_t j) const { return p[j * SW_BLOCK].w; } };
struct HbmQ { uint4* p; size_t NL; __device__ __forceinline__ uint4& at(uint32_t j) const { return p[(size_t)j * NL]; } };
// k_resolve: only the meta words (type | transmits | seq) of the lane's queue, staged in LDS
struct MetaQ { uint32_t* p; __device__ __forceinline__ uint32_t& meta(uint32_t j) const { return p[j * SW_BLOCK]; } };

// one GetBroadcasts(overhead, limit) over a queue.  `live` = entries still queued;
// returns the bitmask sent; bumps transmits / retires at the retransmit limit.
template <typename QV>
__device__ uint32_t get_broadcasts(DevRef D, QV sq, uint32_t n, uint32_t& live, uint32_t overhead, int limit, int& used_out) {
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
    if (m_tr(meta) + 1 >= D.retransmit_limit) live &= ~(1u << j);          // Finished()
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
template <int KMAX, bool SERF, bool MULTI>
__device__ __forceinline__ void role_gossip(DevRef D, uint32_t r, uint32_t bx, uint4* lds_q, uint32_t* lds_stats, uint32_t* s_cnt, uint32_t* s_base, uint32_t* lds_exc) {
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
  __syncthreads();

  uint4* sq = lds_q + threadIdx.x;                 // entry j at sq[j*256]
  uint4* se = lds_q + (size_t)D.Q * SW_BLOCK + threadIdx.x;
  constexpr bool serf = SERF;
  const uint32_t* nw = D.nw + (size_t)r * D.N;
  const bool filter = (D.flags & SWIM_F_FILTER_NOOP) != 0;

  uint32_t np = 0, peers[KMAX], pw[KMAX], sent_m[KMAX], sent_e[SERF ? KMAX : 1], loc[MULTI ? KMAX : 1], psh[MULTI ? KMAX : 1];
  uint32_t qlen = 0, evqlen = 0, live_m = 0, live_e = 0, nq = 0, ne = 0;
  size_t l = 0; uint4 h = make_uint4(0, 0, 0, 0);
  bool active = false, quiet = false;
  uint32_t c_pkt = 0, c_drop = 0, c_filt = 0, c_s0 = 0, c_s1 = 0, c_s2 = 0, c_s3 = 0;   // per-lane tallies
  uint32_t wi = NW_DEAD;
  if (i != NONE) {                                  // trip 1: node word and header together
    l = (size_t)r * D.nloc + (i - D.i0);
    wi = nw[i]; h = D.hdr[l];
  }

  if (!(wi & NW_INERT)) {
    uint32_t k = i - D.i0;
    qlen = h_qlen(h.y); evqlen = h_evqlen(h.y);
    if (!qlen && !evqlen) quiet = true;
    else {
      active = true;
      size_t NL = (size_t)D.R * D.nloc;
      // trip 2: the queue entries and the first four peer candidates' node words, all independent
      uint4 e0 = qlen > 0 ? D.q[l] : make_uint4(0, 0, 0, 0), e1 = qlen > 1 ? D.q[NL + l] : make_uint4(0, 0, 0, 0);
      uint32_t found;
      if (D.ablate & 1u) { found = D.k_gossip < (uint32_t)KMAX ? D.k_gossip : (uint32_t)KMAX; for (uint32_t p = 0; p < found; p++) { peers[p] = (i + 1 + p * 977u) % D.N; pw[p] = 0; } }
      else found = k_random_nodes(D, r, i, k, t, SW_STREAM_GOSSIP, D.k_gossip < (uint32_t)KMAX ? D.k_gossip : (uint32_t)KMAX, 0, NONE, peers, pw, X);
      if (D.ablate & 32u) found = 0;
      if (qlen > 0) sq[0] = e0;
      if (qlen > 1) sq[SW_BLOCK] = e1;
      for (uint32_t j = 2; j < qlen; j++) sq[j * SW_BLOCK] = D.q[(size_t)j * NL + l];
      if (serf) for (uint32_t j = 0; j < evqlen; j++) se[j * SW_BLOCK] = D.evq[(size_t)j * NL + l];
      // trip 3: where the subjects of the first two rumours keep their view columns
      uint32_t ws0 = (filter && qlen > 0) ? (X.usable() ? X.word(e0.x) : nw[e0.x]) : 0, ws1 = (filter && qlen > 1) ? (X.usable() ? X.word(e1.x) : nw[e1.x]) : 0;
      live_m = qlen >= 32 ? 0xFFFFFFFFu : (1u << qlen) - 1;
      live_e = evqlen >= 32 ? 0xFFFFFFFFu : (1u << evqlen) - 1;
      // per peer one GetBroadcasts() (LDS only)
      uint32_t npk = 0;
      bool ok[KMAX];
      for (uint32_t p = 0; p < found; p++) {
        int used = 0, used2 = 0;
        uint32_t tm = (D.ablate & 2u) ? (live_m & 1u) : get_broadcasts(D, LdsQ{sq}, qlen, live_m, 2, (int)D.budget, used), te = 0;
        int avail = (int)D.budget - used;
        if (serf && avail > 2 + 1) te = get_broadcasts(D, LdsQ{se}, evqlen, live_e, 3, avail, used2);
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
      // trip 4: the no-op filter.  The view records of every (peer, rumour 0) pair are fetched together;
      // rumours beyond the first take the one-at-a-time path.
      if (filter && !(D.ablate & 4u)) {
        uint4 va0[KMAX];
#pragma unroll
        for (int p = 0; p < KMAX; p++) {
          bool need = (uint32_t)p < npk && ok[p] && (sent_m[p] & 1u) && (!MULTI || peers[p] / D.nloc == D.rank) && e0.x != peers[p] && NW_HAS_SLOT(ws0);
          va0[p] = need ? D.va[((size_t)r * D.S + NW_SLOT(ws0)) * D.nloc + (peers[p] - D.i0)] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < KMAX; p++) {
          if ((uint32_t)p >= npk || !ok[p] || (MULTI && peers[p] / D.nloc != D.rank)) continue;
          uint32_t tm = sent_m[p];
          if ((tm & 1u) && e0.x != peers[p] && NW_HAS_SLOT(ws0)) {
            size_t ci = ((size_t)r * D.S + NW_SLOT(ws0)) * D.nloc + (peers[p] - D.i0);
            if (noop_given_view(D, va0[p], ci, e0)) { tm &= ~1u; c_filt++; }
          }
          for (uint32_t m = tm & ~1u; m; m &= m - 1) {
            uint32_t j = __ffs(m) - 1; uint4 e = sq[j * SW_BLOCK];
            if (e.x == peers[p]) continue;
            uint32_t ws = j == 1 ? ws1 : (X.usable() ? X.word(e.x) : nw[e.x]);
            if (!NW_HAS_SLOT(ws)) continue;
            size_t ci = ((size_t)r * D.S + NW_SLOT(ws)) * D.nloc + (peers[p] - D.i0);
            if (noop_given_view(D, D.va[ci], ci, e)) { tm &= ~(1u << j); c_filt++; }
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
  if (!(D.ablate & 16u)) S.wave_add(ST_PKT_SENT, c_pkt); S.wave_add(ST_PKT_DROP, c_drop); S.wave_add(ST_FILTERED, c_filt);
  S.wave_add(ST_SENT0, c_s0); S.wave_add(ST_SENT1, c_s1); S.wave_add(ST_SENT2, c_s2); S.wave_add(ST_SENT3, c_s3);

  // ---- compaction of the block's packets into the outbound lists
  // (a) records for nodes of this shard: wavefront prefix sum of the per-lane counts, one LDS atomic per
  //     wave, block-private segment -> no global atomic and no barrier on the way out
  const uint32_t segb = r * D.nb_gossip + bx;
  uint32_t n_loc = 0;
#define PKT_SH(p) (MULTI ? psh[p] : D.rank)
#define PKT_N(p) ((uint32_t)(__popc(sent_m[p]) + (SERF ? __popc(sent_e[p]) : 0)))
  for (uint32_t p = 0; p < np; p++) if (PKT_SH(p) == D.rank) n_loc += PKT_N(p);
  uint32_t incl = n_loc, my_off = 0;
  if (__any(n_loc != 0)) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { uint32_t v = __shfl_up(incl, off); if (sw_lane() >= (uint32_t)off) incl += v; }
    uint32_t wave_total = __shfl(incl, 63), wbase = 0;
    if (sw_lane() == 63) wbase = atomicAdd(&s_cnt[D.rank], wave_total);
    wbase = __shfl(wbase, 63);
    my_off = wbase + incl - n_loc;
  }
  // (b) records for other shards: LDS offsets, then one global atomicAdd per (block, shard)
  if (MULTI) {
    for (uint32_t p = 0; p < np; p++) if (PKT_SH(p) != D.rank) loc[MULTI ? p : 0] = atomicAdd(&s_cnt[PKT_SH(p)], PKT_N(p));
    __syncthreads();
    if (threadIdx.x < D.n_shards && threadIdx.x != D.rank) {
      uint32_t c = s_cnt[threadIdx.x], b = 0;
      if (c) {
        b = atomicAdd(&D.out_cnt[threadIdx.x], c);
        if (b + c > D.out_cap_tab[threadIdx.x]) { atomicOr(D.err, SW_ERR_EDGE_OVF); b = NONE; }
        atomicAdd(stat_ptr(D, ST_EDGES), (unsigned long long)c);
        atomicAdd(stat_ptr(D, ST_EDGES_REMOTE), (unsigned long long)c);
      }
      s_base[threadIdx.x] = b;
    }
    __syncthreads();
  }
  for (uint32_t p = 0; p < np; p++) {
    uint4* dst;
    if (PKT_SH(p) == D.rank) { dst = D.seg + (size_t)segb * D.seg_cap + my_off; my_off += PKT_N(p); }
    else { uint32_t b = s_base[PKT_SH(p)]; if (b == NONE) continue; dst = D.out_tab[PKT_SH(p)] + b + loc[MULTI ? p : 0]; }
    uint32_t gdst = r * D.N + peers[p];
    for (uint32_t m = sent_m[p]; m; m &= m - 1) {
      uint4 e = sq[(__ffs(m) - 1) * SW_BLOCK];
      *dst++ = make_uint4(gdst, e.x, e.y, (m_type(e.w) << 30) | (e.z & 0x3FFFFFFFu));
    }
    if (SERF)
      for (uint32_t m = sent_e[p]; m; m &= m - 1) {
        uint4 e = se[(__ffs(m) - 1) * SW_BLOCK];
        *dst++ = make_uint4(gdst, e.x, e.y, (uint32_t)SWIM_MSG_USER << 30);
      }
  }
#undef PKT_SH
#undef PKT_N

  // ---- write the queues back, compacted; untouched entries are not rewritten
  if (active && !(D.ablate & 8u)) {
    size_t NL = (size_t)D.R * D.nloc;
    for (uint32_t j = 0; j < qlen; j++) if ((live_m >> j) & 1u) { D.q[(size_t)nq * NL + l] = sq[j * SW_BLOCK]; nq++; }
    if (SERF) for (uint32_t j = 0; j < evqlen; j++) if ((live_e >> j) & 1u) { D.evq[(size_t)ne * NL + l] = se[j * SW_BLOCK]; ne++; }
    uint32_t hy = h_pack(h_leaving(h.y), nq, ne);
    if (hy != h.y) { h.y = hy; D.hdr[l] = h; }
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
    uint32_t c = s_cnt[D.rank];                    // every wave has added its total (barrier above)
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
__device__ void send_state(DevRef D, bool on, uint32_t r, uint32_t owner, uint32_t dst, uint32_t& c_edges, uint32_t& c_remote, uint32_t& c_filt) {
  uint32_t ns = on ? D.n_slots[r] : 0, ns_max = ns;
  for (int off = 32; off; off >>= 1) { uint32_t v = __shfl_xor(ns_max, off); ns_max = v > ns_max ? v : ns_max; }
  const uint32_t sh = on ? dst / D.nloc : 0;
  const bool filter = (D.flags & SWIM_F_FILTER_NOOP) && sh == D.rank;
  for (uint32_t sl = 0; sl < ns_max; sl++) {
    bool want = false; uint4 rec = make_uint4(0, 0, 0, 0);
    if (sl < ns) {
      size_t sidx = (size_t)r * D.S + sl;
      uint4 a = D.va[sidx * D.nloc + (owner - D.i0)];
      if (a.x != SW_BASE_KEY) {                            // the base row merges to nothing
        uint32_t x = D.subj_node[sidx], st = SW_KST(a.x), type, from = 0;
        if (st == SWIM_STATE_ALIVE) type = SWIM_MSG_ALIVE;
        else if (st == SWIM_STATE_LEFT) { type = SWIM_MSG_DEAD; from = x; }
        else { type = SWIM_MSG_SUSPECT; from = dst; }
        want = true;
        if (filter && x != dst) {
          size_t ci = sidx * D.nloc + (dst - D.i0);
          if (noop_given_view(D, D.va[ci], ci, make_uint4(x, SW_KINC(a.x), from, type << 30))) { want = false; c_filt++; }
        }
        rec = mk_edge(D, r, dst, x, SW_KINC(a.x), type, from);
      }
    }
    wave_append_sharded(D, want, sh, rec);
    c_edges += want; c_remote += want && sh != D.rank;
  }
}
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
      if (!(wo & NW_INERT) && k_random_nodes(D, r, o, o - D.i0, t, SW_STREAM_PUSHPULL, 1, 1, NONE, &p, &wp, X))
        go = !(wp & NW_DEAD) && NW_PART(wo) == NW_PART(wp);          // else the TCP dial fails
    }
  }
  uint32_t c_edges = 0, c_remote = 0, c_filt = 0;
  send_state(D, go, r, o, p, c_edges, c_remote, c_filt);
  uint32_t sh = go ? p / D.nloc : 0;
  wave_append_sharded(D, go, sh, mk_edge(D, r, p, SWIM_SUBJECT_PULL, o, SWIM_MSG_ALIVE, 0));
  c_edges += go; c_remote += go && sh != D.rank;
  S.count(ST_PUSHPULLS, go);
  S.wave_add(ST_EDGES, c_edges); S.wave_add(ST_EDGES_REMOTE, c_remote); S.wave_add(ST_FILTERED, c_filt);
  S.flush(D);
}
// pull requests are filed in 64 sub-lists (k_resolve picks one by block) so that no counter is hot
#define SW_PP_LISTS 64
__device__ __forceinline__ void role_ppreply(DevRef D, uint32_t b, uint32_t nb, uint32_t* lds_stats) {
  uint32_t t = *D.tick, li = t & 1u;
  if (t == 0 || (t - 1) % D.P) return;              // requests only exist the tick after a boundary
  BlockStats S; S.init(lds_stats);
  uint32_t sub_cap = D.pp_cap / SW_PP_LISTS, c_edges = 0, c_remote = 0, c_filt = 0;
  for (uint32_t sub = b; sub < SW_PP_LISTS; sub += nb) {
    uint32_t n = D.pp_cnt[(li * SW_PP_LISTS + sub) * 16]; if (n > sub_cap) n = sub_cap;
    for (uint32_t e0 = 0; e0 < n; e0 += SW_BLOCK) {
      uint32_t e = e0 + threadIdx.x; bool on = e < n; uint32_t r = 0, p = 0, o = 0;
      if (on) {
        uint2 rq = D.pp_list[((size_t)li * SW_PP_LISTS + sub) * sub_cap + e];
        r = rq.x / D.nloc; p = D.i0 + rq.x % D.nloc; o = rq.y;
        on = !(D.nw[(size_t)r * D.N + p] & NW_INERT);
      }
      send_state(D, on, r, p, o, c_edges, c_remote, c_filt);
    }
  }
  S.wave_add(ST_EDGES, c_edges); S.wave_add(ST_EDGES_REMOTE, c_remote); S.wave_add(ST_FILTERED, c_filt);
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
  uint32_t c_rem = 0;
  for (uint32_t a = b; a < D.NB; a += nb) {
    uint32_t n = D.carry_cl[a].x; if (n > D.carry_cap) n = D.carry_cap;
    uint4* area = D.carry + ((size_t)par * D.NB + a) * D.carry_cap;
    for (uint32_t e0 = 0; e0 < n; e0 += SW_BLOCK) {
      uint32_t e = e0 + threadIdx.x; bool rem = false; uint4 rec = make_uint4(0, 0, 0, 0); uint32_t sh = 0;
      if (e < n) { rec = area[e]; sh = (rec.x % D.N) / D.nloc; rem = sh != D.rank; }
      wave_append_sharded(D, rem, sh, rec);
      if (rem) { area[e].x = SW_DST_VOID; c_rem++; }
    }
  }
  S.wave_add(ST_EDGES, c_rem); S.wave_add(ST_EDGES_REMOTE, c_rem);
  S.flush(D);
}

// =================================================================================================
// k_begin — the fused first launch of a tick.  grid = nb_expire + nb_pend + R*(nb_probe + nb_gossip);
// dynamic LDS = (Q+EQ) * 256 * 16 bytes (the gossip role's staged queues)
// =================================================================================================
template <int KMAX, bool SERF, bool MULTI>
__global__ void __launch_bounds__(SW_BLOCK) k_begin(const SwDev* __restrict__ Dp, BeginPlan pl) {
  SW_DEV_BIND
  extern __shared__ uint4 lds_q[];
  __shared__ uint32_t lds_stats[ST_COUNT];
  __shared__ uint32_t s_cnt[SW_MAX_SHARDS], s_base[SW_MAX_SHARDS], lds_exc[2 * SW_EXC_MAX];
  uint32_t b = blockIdx.x;
  // diagnostics (SWIMSIM_ROLECLK): when did the first block of a role start, when did its last block end
  const unsigned long long t_in = D.role_clk ? wall_clock64() : 0;
#define ROLE_DONE(id) do { if (D.role_clk && threadIdx.x == 0) { uint32_t tk = *D.tick; if (tk < D.role_clk_ticks) { \
    unsigned long long* c = D.role_clk + (((size_t)tk * 8 + (id)) * 64 + (blockIdx.x & 63u)) * 2; atomicMin(c, t_in); atomicMax(c + 1, (unsigned long long)wall_clock64()); } } } while (0)
  if (b < pl.nb_expire) { if (pl.roles & 1u) role_expire(D, b, pl.nb_expire); ROLE_DONE(0); return; }
  b -= pl.nb_expire;
  if (b < pl.nb_pend) { if (pl.roles & 2u) role_pending<KMAX>(D, b, pl.nb_pend, lds_stats, MULTI ? *D.peer_act : 1u); ROLE_DONE(1); return; }
  b -= pl.nb_pend;
  if (b < D.R * pl.nb_probe) { if (pl.roles & 4u) role_probe<MULTI>(D, b / pl.nb_probe, b % pl.nb_probe, (b % pl.nb_probe) * SW_BLOCK + threadIdx.x, lds_stats, lds_exc, s_cnt, MULTI ? *D.peer_act : 1u); ROLE_DONE(2); return; }
  b -= D.R * pl.nb_probe;
  if (b < D.R * pl.nb_gossip) { if (pl.roles & 8u) role_gossip<KMAX, SERF, MULTI>(D, b / pl.nb_gossip, b % pl.nb_gossip, lds_q, lds_stats, s_cnt, s_base, lds_exc); ROLE_DONE(3); return; }
  b -= D.R * pl.nb_gossip;
  if (b < pl.nb_ppreply) { if (pl.roles & 16u) role_ppreply(D, b, pl.nb_ppreply, lds_stats); ROLE_DONE(4); return; }
  b -= pl.nb_ppreply;
  if (MULTI) {
    if (b < pl.nb_carry) { if (pl.roles & 32u) role_carry(D, b, pl.nb_carry, lds_stats); ROLE_DONE(5); return; }
    b -= pl.nb_carry;
  }
  if (pl.roles & 16u) role_pushpull(D, b / pl.nb_pp, (b % pl.nb_pp) * SW_BLOCK + threadIdx.x, lds_stats, lds_exc);
  ROLE_DONE(6);
#undef ROLE_DONE
}
typedef void (*BeginKernel)(const SwDev*, BeginPlan);
// pick the leanest instantiation the configuration allows
static BeginKernel select_begin(uint32_t fanout, bool serf, bool multi) {
  const bool k8 = fanout > 4;
  if (k8) return serf ? (multi ? k_begin<8, true, true> : k_begin<8, true, false>) : (multi ? k_begin<8, false, true> : k_begin<8, false, false>);
  return serf ? (multi ? k_begin<4, true, true> : k_begin<4, true, false>) : (multi ? k_begin<4, false, true> : k_begin<4, false, false>);
}

// =================================================================================================
// k_deliver — packetListen/ingestPacket: scatter an edge list into the per-node inbox rows.  One
// returning atomic on the row's count word reserves the slot; the record lands in the same 64-byte
// line for the first five arrivals.  Slot requests are granted on the spot (grant_slot).
// =================================================================================================
// Give subject x of replica r a view-column slot.  Runs inside k_deliver (nobody reads slot bits there):
// the first request to flip the node word's slot field to the "being granted" pattern wins, later
// duplicates (other probers, other shards) see a non-zero field and leave.  Which index a subject gets is
// not observable: everything the ABI reports is keyed by node id.
__device__ void grant_slot(DevRef D, uint32_t r, uint32_t x) {
  size_t g = (size_t)r * D.N + x;
  uint32_t w = D.nw[g];
  if (NW_HAS_SLOT(w)) return;
  if (atomicCAS(&D.nw[g], w, w | NW_SLOT_MASK) != w) return;
  uint32_t sl = atomicAdd(&D.n_slots[r], 1u);
  if (sl >= D.S) {
    atomicSub(&D.n_slots[r], 1u); D.nw[g] = w;
    atomicOr(D.err, SW_ERR_SUBJ_OVF); atomicAdd(stat_ptr(D, ST_SUBJ_OVF), 1ull);
    return;
  }
  size_t sidx = (size_t)r * D.S + sl;
  D.subj_node[sidx] = x; D.slot_dirty[sidx] = 1; D.slot_maxinc[sidx] = 1;
  D.slot_susp[sidx] = 0; D.slot_mindl[sidx] = NONE;
  D.nw[g] = w | (sl + 1);
  // the replica's exception list: x may already be on it (a dead node has a non-zero word)
  uint32_t n = D.exc_cnt[r]; bool listed = false;
  for (uint32_t j = 0; j < n && j < SW_EXC_MAX; j++)
    if (D.exc_ent[(size_t)r * SW_EXC_MAX + j].x == x) { listed = true; D.exc_ent[(size_t)r * SW_EXC_MAX + j].y = w | (sl + 1); }
  if (!listed) { uint32_t pos = atomicAdd(&D.exc_cnt[r], 1u); if (pos < SW_EXC_MAX) D.exc_ent[(size_t)r * SW_EXC_MAX + pos] = make_uint2(x, w | (sl + 1)); }
}

// reserve: one returning atomic on the count word of the node's 64-byte inbox line
__device__ __forceinline__ uint32_t inbox_reserve(DevRef D, uint4 rec, size_t& l) {
  if (rec.x == NONE) { grant_slot(D, rec.z, rec.y); return NONE; }     // subject-slot request
  uint32_t r = rec.x / D.N, x = rec.x % D.N;
  if (x < D.i0 || x >= D.i0 + D.nloc) return NONE;
  uint32_t w = D.nw[rec.x];
  if (w & NW_DEAD) return NONE;                    // e.g. a push-pull reply to a requester that died meanwhile
  if (w & NW_ATTACHED) { if (rec.y != SWIM_SUBJECT_PIGGY) capture(D, NONE, rec.x, rec.y, rec.z, rec.w); return NONE; }
  l = (size_t)r * D.nloc + (x - D.i0);
  if (rec.y == SWIM_SUBJECT_PIGGY) {               // a piggy-back order for a node with nothing queued is a no-op
    if (!q_bit(D, l) || (D.ablate & 256u)) return NONE;   // (queues do not change between k_begin and k_resolve)
  }
  return atomicAdd(&D.in_cnt[l], 1u);
}
// place: the message lands in the same line for the first SW_INBOX_FAST arrivals, else in the overflow row
__device__ __forceinline__ void inbox_place(DevRef D, uint4 rec, size_t l, uint32_t pos) {
  if (pos == NONE) return;
  uint32_t* m = nullptr;
  if (pos < SW_INBOX_FAST) m = D.inbox1 + l * 16 + 1 + 3 * pos;
  else if (pos < D.C) m = D.inbox2 + (l * D.C2 + (pos - SW_INBOX_FAST)) * 3;
  if (m) { m[0] = rec.y; m[1] = rec.z; m[2] = rec.w; }
  if (pos == 0 && D.fast_blocks) D.in_any[l / SW_BLOCK] = 1;
}
// four records per thread per trip: all four atomics are in flight before the first store
__device__ __forceinline__ void deliver_span(DevRef D, const uint4* edges, uint32_t n, uint32_t first, uint32_t stride) {
  for (uint32_t e = first; e < n; e += 4 * stride) {
    uint4 rec[4]; size_t l[4]; uint32_t pos[4];
#pragma unroll
    for (int j = 0; j < 4; j++) if (e + j * stride < n) rec[j] = edges[e + j * stride];
#pragma unroll
    for (int j = 0; j < 4; j++) pos[j] = e + j * stride < n ? inbox_reserve(D, rec[j], l[j]) : NONE;
#pragma unroll
    for (int j = 0; j < 4; j++) inbox_place(D, rec[j], l[j], pos[j]);
  }
}
// The broadcasts piggy-backed on last tick's pings and acks (picked by k_resolve, one private area per block)
// arrive with this tick's packets.  Same no-op filter as the gossip role applies at the sender: here the
// receiver's view is read before the inbox is touched.
__device__ void deliver_carried(DevRef D, const uint4* area, uint32_t n, uint32_t& c_edges, uint32_t& c_filt) {
  const bool filter = (D.flags & SWIM_F_FILTER_NOOP) != 0;
  for (uint32_t e = threadIdx.x; e < n; e += SW_BLOCK) {
    uint4 rec = area[e];
    if (rec.x == SW_DST_VOID) continue;              // left for another shard in k_begin
    uint32_t r = rec.x / D.N, x = rec.x % D.N, type = rec.w >> 30;
    if (filter && type != SWIM_MSG_USER && rec.y != x) {
      uint32_t ws = D.nw[(size_t)r * D.N + rec.y];
      if (NW_HAS_SLOT(ws)) {
        size_t ci = ((size_t)r * D.S + NW_SLOT(ws)) * D.nloc + (x - D.i0);
        if (noop_given_view(D, D.va[ci], ci, make_uint4(rec.y, rec.z, rec.w & 0x3FFFFFFFu, type << 30))) { c_filt++; continue; }
      }
    }
    c_edges++;
    size_t l; uint32_t pos = inbox_reserve(D, rec, l);
    inbox_place(D, rec, l, pos);
  }
}
// grid = n_seg blocks + extra blocks over the shard's misc list.  Block b drains segment b and the carry areas
// b, b + n_seg, ... (their counts are fetched together with the segment's: no extra trip in a quiet tick)
__global__ void __launch_bounds__(SW_BLOCK) k_deliver(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  uint32_t b = blockIdx.x;
  if (b < D.n_seg) {
    uint32_t n = D.seg_cnt[b], last = D.seg_last[b];
    uint32_t cn[4] = { 0, 0, 0, 0 }, cl[4] = { 0, 0, 0, 0 };
    // anything carried into this tick?  (uniform words: k_resolve stamps carry_stamp with the tick its picks
    // travel in, so a tick without piggy-backed broadcasts costs this block nothing more)
    uint32_t par = 0;
    const bool piggy = D.nb_carry != 0 && *D.carry_stamp == *D.tick;
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
    if (n) deliver_span(D, D.seg + (size_t)b * D.seg_cap, n, threadIdx.x, SW_BLOCK);
    if (threadIdx.x == 0 && n) D.seg_cnt[b] = 0;
    if (!piggy) return;
    uint32_t c_edges = 0, c_filt = 0;
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (cn[j] && !(D.ablate & 128u)) deliver_carried(D, D.carry + ((size_t)par * D.NB + b + j * D.n_seg) * D.carry_cap, cn[j] < D.carry_cap ? cn[j] : D.carry_cap, c_edges, c_filt);
    for (uint32_t a = b + 4 * D.n_seg; a < D.NB; a += D.n_seg) {      // only with very fine quanta (G > 4)
      uint2 cc = D.carry_cl[a]; uint32_t c = cc.x;
      __syncthreads();
      if (threadIdx.x == 0 && (cc.x | cc.y)) D.carry_cl[a] = make_uint2(0, c);
      if (c) deliver_carried(D, D.carry + ((size_t)par * D.NB + a) * D.carry_cap, c < D.carry_cap ? c : D.carry_cap, c_edges, c_filt);
    }
    if (__any((c_edges | c_filt) != 0)) {
      for (int off = 32; off; off >>= 1) { c_edges += __shfl_down(c_edges, off); c_filt += __shfl_down(c_filt, off); }
      if (sw_lane() == 0) {
        if (c_edges) atomicAdd(stat_ptr(D, ST_EDGES), (unsigned long long)c_edges);
        if (c_filt) atomicAdd(stat_ptr(D, ST_FILTERED), (unsigned long long)c_filt);
      }
    }
    return;
  }
  b -= D.n_seg;
  uint32_t nb = gridDim.x - D.n_seg, n = D.out_cnt[D.rank];
  if (n > D.out_cap_tab[D.rank]) n = D.out_cap_tab[D.rank];
  deliver_span(D, D.out_tab[D.rank], n, b * SW_BLOCK + threadIdx.x, nb * SW_BLOCK);
}
// records handed over by other shards (swim_inbound)
__global__ void __launch_bounds__(SW_BLOCK) k_deliver_list(const SwDev* __restrict__ Dp, const uint4* edges, uint32_t n) {
  SW_DEV_BIND
  deliver_span(D, edges, n, blockIdx.x * SW_BLOCK + threadIdx.x, gridDim.x * SW_BLOCK);
}

// host-side stimulus (leave/update) needs a slot before the tick: single-threaded variant
__device__ void alloc_slot(DevRef D, uint32_t r, uint32_t x) {
  size_t g = (size_t)r * D.N + x;
  uint32_t w = D.nw[g];
  if (NW_HAS_SLOT(w)) return;
  uint32_t sl = D.n_slots[r];
  if (sl >= D.S) { atomicOr(D.err, SW_ERR_SUBJ_OVF); atomicAdd(stat_ptr(D, ST_SUBJ_OVF), 1ull); return; }
  D.n_slots[r] = sl + 1;
  size_t sidx = (size_t)r * D.S + sl;
  D.subj_node[sidx] = x; D.slot_dirty[sidx] = 1; D.slot_maxinc[sidx] = 1;
  D.slot_susp[sidx] = 0; D.slot_mindl[sidx] = NONE;
  atomicOr(&D.nw[g], sl + 1);
  // keep the replica's exception list exact without a rescan (single-threaded here)
  uint32_t n = D.exc_cnt[r];
  if (n <= SW_EXC_MAX) {
    bool listed = false; const uint32_t wn = w | (sl + 1);
    for (uint32_t j = 0; j < n; j++) if (D.exc_ent[(size_t)r * SW_EXC_MAX + j].x == x) { listed = true; D.exc_ent[(size_t)r * SW_EXC_MAX + j].y = wn; }
    if (!listed) { if (n < SW_EXC_MAX) D.exc_ent[(size_t)r * SW_EXC_MAX + n] = make_uint2(x, wn); D.exc_cnt[r] = n + 1; }
  }
}
// =================================================================================================
// k_resolve — handleAlive/handleSuspect/handleDead/handleUserEvent for everything that reached a
// node this tick, applied in ascending (user?, subject, type, incarnation, from) order with
// duplicates applied once.  Literal aliveNode/suspectNode/deadNode/refute (state.go) and
// suspicion.Confirm (suspicion.go) against the observer's own view column.
// =================================================================================================
struct NodeCtx {
  DevRef D; BlockStats& S;
  uint32_t r, o, k, t; size_t l, NL;
  uint32_t self_inc, leaving, qlen, evqlen, qseq, ev_clock;
  uint32_t c_pig = 0, c_sent01 = 0, c_sent23 = 0;   // piggy-back tallies (orders are frequent: no LDS atomic each); two 16-bit halves
  uint4 h0;
  __device__ NodeCtx(DevRef d, BlockStats& s) : D(d), S(s) {}

  __device__ void load() {
    h0 = D.hdr[l];
    self_inc = h0.x; leaving = h_leaving(h0.y); qlen = h_qlen(h0.y); evqlen = h_evqlen(h0.y); qseq = h0.z; ev_clock = h0.w;
  }
  // most deliveries in a saturated cluster are old news: only write the header back when it changed
  // did the node go from "nothing queued" to "something queued" (or back) since load()?
  __device__ bool q_became_set() const { return !(h_qlen(h0.y) | h_evqlen(h0.y)) && (qlen | evqlen); }
  __device__ bool q_became_clr() const { return (h_qlen(h0.y) | h_evqlen(h0.y)) && !(qlen | evqlen); }
  __device__ void store() {
    uint4 h = make_uint4(self_inc, h_pack(leaving, qlen, evqlen), qseq, ev_clock);
    if (h.x != h0.x || h.y != h0.y || h.z != h0.z || h.w != h0.w) D.hdr[l] = h;
  }

  // QueueBroadcast on the HBM-resident queue: same-subject invalidation, Prune() on overflow
  __device__ void queue_push(uint4* qb, uint32_t cap, uint32_t& len, uint32_t seq, bool named,
                             uint32_t subject, uint32_t type, uint32_t inc, uint32_t from, int drop_stat) {
    uint32_t n = len;
    if (named)
      for (uint32_t j = 0; j < n; j++)
        if (qb[(size_t)j * NL].x == subject) { qb[(size_t)j * NL] = qb[(size_t)(n - 1) * NL]; n--; break; }
    uint4 e = make_uint4(subject, inc, from, m_pack(type, 0, seq));
    if (n == cap) {
      uint32_t w = NONE, wmeta = e.w;
      for (uint32_t j = 0; j < n; j++) { uint32_t mj = qb[(size_t)j * NL].w; if (ent_before(D, wmeta, mj)) { wmeta = mj; w = j; } }
      S.add(drop_stat);
      if (w != NONE) qb[(size_t)w * NL] = e;
    } else { qb[(size_t)n * NL] = e; n++; }
    len = n;
    if (D.fast_blocks) D.q_any[l / SW_BLOCK] = 1;
  }
  __device__ void broadcast(uint32_t subject, uint32_t type, uint32_t inc, uint32_t from) {
    queue_push(D.q + l, D.Q, qlen, qseq, true, subject, type, inc, from, ST_QDROPS); qseq++;
  }
  __device__ void record_event(uint32_t type, uint32_t node, uint32_t ltime, uint32_t inc) {
    uint32_t pos = atomicAdd(D.ev_cnt, 1u);
    if (pos < D.ev_cap) { swim_event ev = { now_ms(D, t), r, type, node, ltime, inc }; D.events[pos] = ev; }
    else atomicOr(D.err, SW_ERR_EVENT_OVF);
  }
  // the observer's record of a subject is edited in registers (a = {key, since, nconf, conf0}) and
  // written back once by the caller
  __device__ void set_view(size_t sidx, uint4& a, uint32_t inc, uint32_t st, bool touch_since) {
    a.x = SW_KEY(inc, st);
    if (touch_since) a.y = now_ms(D, t);
    if (inc > D.slot_maxinc[sidx]) atomicMax(&D.slot_maxinc[sidx], inc);
    D.slot_dirty[sidx] = 1;
  }
  __device__ void refute(size_t sidx, uint4& a, uint32_t accused) {
    uint32_t inc = self_inc + 1;
    if (accused >= inc) inc = accused + 1;
    self_inc = inc;
    uint2 h = D.ph[l];                                     // awareness lives with the probe state
    D.ph[l].y = p_pack(p_epoch(h.y), awareness_apply(D, p_aw(h.y), +1), p_stage(h.y), p_nackm(h.y));
    set_view(sidx, a, inc, SWIM_STATE_ALIVE, false);
    broadcast(o, SWIM_MSG_ALIVE, inc, 0);
    S.add(ST_REFUTES);
  }
  __device__ void alive_node(uint32_t x, uint32_t inc, uint32_t upd) {
    uint32_t w = D.nw[(size_t)r * D.N + x]; if (!NW_HAS_SLOT(w)) return;
    size_t sidx = (size_t)r * D.S + NW_SLOT(w), ci = sidx * D.nloc + k;
    uint4 a = D.va[ci];
    uint32_t key = a.x; bool local = x == o;
    if (local && leaving) return;
    if (!local && inc <= SW_KINC(key)) return;
    if (local && inc < SW_KINC(key)) return;
    a.z = 0;                                               // delete(m.nodeTimers, a.Node)
    uint32_t old = SW_KST(key);
    if (local) { if (inc == SW_KINC(key)) { D.va[ci] = a; return; } refute(sidx, a, inc); }
    else {
      broadcast(x, SWIM_MSG_ALIVE, inc, upd);
      set_view(sidx, a, inc, SWIM_STATE_ALIVE, old != SWIM_STATE_ALIVE);
      S.add(ST_APPL0);
      if (o == D.watch) {
        if (old == SWIM_STATE_DEAD || old == SWIM_STATE_LEFT) record_event(SWIM_EVENT_MEMBER_JOIN, x, 0, inc);
        else if (upd) record_event(SWIM_EVENT_MEMBER_UPDATE, x, 0, inc);
      }
    }
    D.va[ci] = a;
  }
  __device__ void suspect_node(uint32_t x, uint32_t inc, uint32_t from) {
    uint32_t w = D.nw[(size_t)r * D.N + x]; if (!NW_HAS_SLOT(w)) return;
    size_t sidx = (size_t)r * D.S + NW_SLOT(w), ci = sidx * D.nloc + k;
    uint4 a = D.va[ci];
    uint32_t key = a.x;
    if (inc < SW_KINC(key)) return;
    if (SW_KST(key) == SWIM_STATE_SUSPECT) {           // timer exists: suspicion.Confirm(from)
      uint32_t nc = a.z;
      if (nc >= D.susp_k || a.w == from) return;
      uint4 b = nc ? D.vb[ci] : make_uint4(0, 0, 0, 0);
      if ((nc >= 1 && b.x == from) || (nc >= 2 && b.y == from) || (nc >= 3 && b.z == from)) return;
      nc++;
      if (nc == 1) b.x = from; else if (nc == 2) b.y = from; else if (nc == 3) b.z = from;
      if (nc <= 3) D.vb[ci] = b;
      a.z = nc; D.va[ci] = a;
      D.slot_dirty[sidx] = 1; S.add(ST_CONFIRMS);
      broadcast(x, SWIM_MSG_SUSPECT, inc, from);
      return;
    }
    if (SW_KST(key) != SWIM_STATE_ALIVE) return;
    if (x == o) { refute(sidx, a, inc); D.va[ci] = a; return; }
    broadcast(x, SWIM_MSG_SUSPECT, inc, from);
    set_view(sidx, a, inc, SWIM_STATE_SUSPECT, true);
    a.z = 0; a.w = from;                                   // newSuspicion(from, k, min, max)
    D.va[ci] = a;
    S.add(ST_APPL1);
  }
  __device__ void dead_node(uint32_t x, uint32_t inc, uint32_t from) {
    uint32_t w = D.nw[(size_t)r * D.N + x]; if (!NW_HAS_SLOT(w)) return;
    size_t sidx = (size_t)r * D.S + NW_SLOT(w), ci = sidx * D.nloc + k;
    uint4 a = D.va[ci];
    uint32_t key = a.x;
    if (inc < SW_KINC(key)) return;
    uint32_t old = SW_KST(key);
    if (old == SWIM_STATE_DEAD || old == SWIM_STATE_LEFT) return;
    a.z = 0;
    if (x == o && !leaving) { refute(sidx, a, inc); D.va[ci] = a; return; }
    broadcast(x, SWIM_MSG_DEAD, inc, from);
    uint32_t st = from == x ? SWIM_STATE_LEFT : SWIM_STATE_DEAD;
    set_view(sidx, a, inc, st, true);
    D.va[ci] = a;
    S.add(ST_APPL2);
    if (o == D.watch && x != o) record_event(st == SWIM_STATE_LEFT ? SWIM_EVENT_MEMBER_LEAVE : SWIM_EVENT_MEMBER_FAILED, x, 0, inc);
  }
  // sendMsg (net.go): extra := getBroadcasts(compoundOverhead, UDPBufferSize - len(msg) - compoundHeaderOverhead),
  // i.e. the memberlist queue and then the serf delegate's user events, for a ping/ack/... this node sent this
  // tick.  Runs before the tick's arrivals are merged; what is picked goes to the block's carry area and
  // reaches `receiver` with the next tick's packets (NONE = the carrier was lost: transmits still count).
  __device__ void piggyback(uint32_t receiver, uint32_t kind, uint32_t* s_carry, uint4* area, uint32_t* lds_meta) {
    const int limit = (int)D.budget - (int)sel4(D.ctl_len, kind & 3u);
    uint32_t live_m = qlen >= 32 ? 0xFFFFFFFFu : (1u << qlen) - 1, live_e = evqlen >= 32 ? 0xFFFFFFFFu : (1u << evqlen) - 1;
    int used = 0, used2 = 0;
    HbmQ qm{D.q + l, NL}, qe{D.evq + l, NL};
    // the pick walks the queue several times: fetch the meta words once (independent loads), pick in LDS
    MetaQ mm{lds_meta + threadIdx.x}, me{lds_meta + (size_t)D.Q * SW_BLOCK + threadIdx.x};
    for (uint32_t j = 0; j < qlen; j++) mm.meta(j) = qm.at(j).w;
    for (uint32_t j = 0; j < evqlen; j++) me.meta(j) = qe.at(j).w;
    uint32_t tm = get_broadcasts(D, mm, qlen, live_m, 2, limit, used), te = 0;
    int avail = limit - used;
    if (D.EQ && avail > 2 + 1) te = get_broadcasts(D, me, evqlen, live_e, 3, avail, used2);
    if (!(tm | te)) return;
    const uint32_t cnt = (uint32_t)(__popc(tm) + __popc(te));
    c_pig++;
    for (uint32_t m = tm; m; m &= m - 1) { uint32_t ty = m_type(mm.meta(__ffs(m) - 1)); if (ty < 2) c_sent01 += 1u << (16 * ty); else c_sent23 += 1u << (16 * (ty - 2)); }
    c_sent23 += (uint32_t)__popc(te) << 16;
    if (receiver != NONE) {
      const uint32_t gdst = r * D.N + receiver;
      const bool att = *D.att_any && (D.nw[gdst] & NW_ATTACHED);   // Transport.WriteTo towards the real node
      uint32_t pos = att ? 0 : atomicAdd(s_carry, cnt);
      if (!att && pos + cnt > D.carry_cap) { atomicOr(D.err, SW_ERR_CARRY_OVF); pos = NONE; }
      for (uint32_t m = tm; m && pos != NONE; m &= m - 1) {
        uint4 e = qm.at(__ffs(m) - 1); uint32_t meta = (m_type(e.w) << 30) | (e.z & 0x3FFFFFFFu);
        if (att) capture(D, o, gdst, e.x, e.y, meta); else area[pos++] = make_uint4(gdst, e.x, e.y, meta);
      }
      for (uint32_t m = te; m && pos != NONE; m &= m - 1) {
        uint4 e = qe.at(__ffs(m) - 1);
        if (att) capture(D, o, gdst, e.x, e.y, (uint32_t)SWIM_MSG_USER << 30); else area[pos++] = make_uint4(gdst, e.x, e.y, (uint32_t)SWIM_MSG_USER << 30);
      }
    }
    // write the bumped transmit counts back; retire what reached the retransmit limit (stable compaction,
    // like the gossip role's write-back)
    uint32_t nq = 0, ne = 0;
    for (uint32_t j = 0; j < qlen; j++) if ((live_m >> j) & 1u) {
      if (nq != j) { uint4 e = qm.at(j); e.w = mm.meta(j); qm.at(nq) = e; } else if ((tm >> j) & 1u) qm.at(j).w = mm.meta(j);
      nq++;
    }
    for (uint32_t j = 0; j < evqlen; j++) if ((live_e >> j) & 1u) {
      if (ne != j) { uint4 e = qe.at(j); e.w = me.meta(j); qe.at(ne) = e; } else if ((te >> j) & 1u) qe.at(j).w = me.meta(j);
      ne++;
    }
    qlen = nq; evqlen = ne;
  }
  // serf handleUserEvent + LamportClock.Witness; ring word0 = n<<30 | ltime
  __device__ void user_event(uint32_t id, uint32_t ltime) {
    if (!(D.flags & SWIM_F_SERF_EVENTS)) return;
    if (ltime >= ev_clock) ev_clock = ltime + 1;
    if (ev_clock > D.EB && ltime < ev_clock - D.EB) { S.add(ST_UEV_STALE); return; }
    uint4* slot = D.ring + (size_t)(ltime % D.EB) * NL + l;
    uint4 sv = *slot; uint32_t n = sv.x >> 30, lt = sv.x & 0x3FFFFFFFu;
    if (n && lt == (ltime & 0x3FFFFFFFu)) {
      if (sv.y == id || (n >= 2 && sv.z == id) || (n >= 3 && sv.w == id)) { S.add(ST_UEV_DEDUP); return; }
    } else n = 0;
    if (n == 3) { S.add(ST_EVDROPS); return; }
    if (n == 0) sv.y = id; else if (n == 1) sv.z = id; else sv.w = id;
    n++; sv.x = (n << 30) | (ltime & 0x3FFFFFFFu); *slot = sv;
    S.add(ST_UEV_DELIVERED);
    if (o == D.watch) record_event(SWIM_EVENT_USER, id, ltime, 0);
    uint32_t seq = D.evseq[l]; D.evseq[l] = seq + 1;
    queue_push(D.evq + l, D.EQ, evqlen, seq, false, id, SWIM_MSG_USER, ltime, 0, ST_EVDROPS);
  }
};

// canonical order key of an inbox record: (user?, subject, type) then (incarnation, from)
__device__ __forceinline__ void edge_key(uint4 e, uint64_t& hi, uint64_t& lo) {
  uint32_t type = e.w >> 30; bool order = e.y == SWIM_SUBJECT_PIGGY;     // orders act on the queue as k_begin left it
  hi = ((uint64_t)!order << 35) | ((uint64_t)(!order && type == SWIM_MSG_USER) << 34) | ((uint64_t)e.y << 2) | type;
  lo = ((uint64_t)e.z << 32) | (e.w & 0x3FFFFFFFu);
}

__global__ void __launch_bounds__(SW_BLOCK) k_resolve(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  extern __shared__ uint32_t lds_meta[];         // [(Q+EQ)][256] meta words of the lane's queues (piggy-back pick)
  __shared__ uint32_t lds_stats[ST_COUNT];
  __shared__ uint32_t s_carry;
  __shared__ uint4 s_in[4][SW_BLOCK];            // the lanes' 64-byte inbox lines (LDS, not registers: occupancy)
  if (D.fast_blocks) {                     // nothing reached this block of nodes: one word and out
    if (!D.in_any[blockIdx.x]) return;
  }
  if (threadIdx.x == 0) s_carry = 0;
  BlockStats S; S.init(lds_stats);
  uint32_t c_pig = 0, c_sent01 = 0, c_sent23 = 0;
  bool q_set = false, q_clr = false;
  if (D.fast_blocks && threadIdx.x == 0) D.in_any[blockIdx.x] = 0;
  size_t NL = (size_t)D.R * D.nloc;
  size_t l = (size_t)blockIdx.x * SW_BLOCK + threadIdx.x;
  if (l < NL) {
    // the whole 64-byte line (count + first five messages) in one go, parked in the lane's LDS column
    // (the count lives in its own dense array: the scatter's returning atomic then works on 4 bytes per node that
    // stay cache resident instead of pulling in the node's 64-byte message line)
    const uint4* row4 = (const uint4*)(D.inbox1 + l * 16);
    uint32_t cnt = D.in_cnt[l];
    if (cnt) {
      s_in[0][threadIdx.x] = row4[0]; s_in[1][threadIdx.x] = row4[1]; s_in[2][threadIdx.x] = row4[2]; s_in[3][threadIdx.x] = row4[3];
      D.in_cnt[l] = 0;
#define IN_WORD(w) (((const uint32_t*)&s_in[(w) >> 2][threadIdx.x])[(w) & 3u])
      if (cnt > D.C) { S.add(ST_INBOX_OVF, cnt - D.C); atomicOr(D.err, SW_ERR_INBOX_OVF); cnt = D.C; }
      const uint32_t* row2 = D.inbox2 + l * D.C2 * 3;
      NodeCtx n(D, S);
      n.r = (uint32_t)(l / D.nloc); n.k = (uint32_t)(l % D.nloc); n.o = D.i0 + n.k; n.t = *D.tick; n.l = l; n.NL = NL;
      n.load();
      bool have_last = false; uint64_t lhi = 0, llo = 0;
      for (;;) {
        bool have = false; uint64_t bhi = 0, blo = 0; uint4 best = make_uint4(0, 0, 0, 0);
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
          n.piggyback(best.z, type, &s_carry, D.carry + ((size_t)((n.t + 1) & 1u) * D.NB + blockIdx.x) * D.carry_cap, lds_meta);
        else if (best.y == SWIM_SUBJECT_PULL && type == SWIM_MSG_ALIVE) {     // push-pull request: answer next tick
          uint32_t li = (n.t + 1) & 1u, sub = blockIdx.x % SW_PP_LISTS, sub_cap = D.pp_cap / SW_PP_LISTS;
          uint32_t pos = atomicAdd(&D.pp_cnt[(li * SW_PP_LISTS + sub) * 16], 1u);
          if (pos < sub_cap) D.pp_list[((size_t)li * SW_PP_LISTS + sub) * sub_cap + pos] = make_uint2((uint32_t)l, best.z);
          else atomicOr(D.err, SW_ERR_PEND_OVF);
        }
        else if (type == SWIM_MSG_ALIVE) n.alive_node(best.y, best.z, from);
        else if (type == SWIM_MSG_SUSPECT) n.suspect_node(best.y, best.z, from);
        else if (type == SWIM_MSG_DEAD) n.dead_node(best.y, best.z, from);
        else n.user_event(best.y, best.z);
        have_last = true; lhi = bhi; llo = blo;
      }
      n.store();
      c_pig = n.c_pig; c_sent01 = n.c_sent01; c_sent23 = n.c_sent23;
      q_set = n.q_became_set(); q_clr = n.q_became_clr();
    }
  }
  q_bits_wave(D, l, q_set, q_clr);
  if (D.flags & SWIM_F_PIGGYBACK) {
    uint32_t s0 = c_sent01 & 0xFFFFu, s1 = c_sent01 >> 16, s2 = c_sent23 & 0xFFFFu, s3 = c_sent23 >> 16;
    S.wave_add(ST_PIGGY, c_pig); S.wave_add(ST_PIGGY_MSGS, s0 + s1 + s2 + s3);
    S.wave_add(ST_SENT0, s0); S.wave_add(ST_SENT1, s1); S.wave_add(ST_SENT2, s2); S.wave_add(ST_SENT3, s3);
  }
  S.flush(D);                                      // (barrier inside: every lane's carry reservations are in)
  if (threadIdx.x == 0 && s_carry) { D.carry_cl[blockIdx.x].x = s_carry; *D.carry_stamp = *D.tick + 1; }
}

// =================================================================================================
// k_census / k_finish — observation: how the live observers of a replica see each dirty subject
// =================================================================================================
__global__ void __launch_bounds__(SW_BLOCK) k_census(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  uint32_t sidx = blockIdx.y, r = sidx / D.S, sl = sidx % D.S;
  if (sl >= D.n_slots[r] || !D.slot_dirty[sidx]) return;
  __shared__ uint32_t acc[CEN_WORDS];
  if (threadIdx.x < CEN_WORDS) acc[threadIdx.x] = threadIdx.x == CEN_MINDL ? NONE : 0;
  __syncthreads();
  uint32_t x = D.subj_node[sidx], maxinc = D.slot_maxinc[sidx];
  uint32_t obs = 0, st[4] = { 0, 0, 0, 0 }, cur = 0, mindl = NONE;
  const uint32_t* nw = D.nw + (size_t)r * D.N;
  for (uint32_t k = blockIdx.x * SW_BLOCK + threadIdx.x; k < D.nloc; k += gridDim.x * SW_BLOCK) {
    uint32_t o = D.i0 + k;
    if (o == x || (nw[o] & NW_DEAD)) continue;
    size_t ci = (size_t)sidx * D.nloc + k;
    uint4 a = D.va[ci];
    uint32_t key = a.x, s = SW_KST(key);
    obs++; st[0] += s == 0; st[1] += s == 1; st[2] += s == 2; st[3] += s == 3;
    cur += SW_KINC(key) == maxinc;
    if (s == SWIM_STATE_SUSPECT) { uint32_t dl = a.y + sel8(D.susp_timeout, a.z & 7u); mindl = dl < mindl ? dl : mindl; }
  }
  uint32_t vals[6] = { obs, st[0], st[1], st[2], st[3], cur };
#pragma unroll
  for (int j = 0; j < 6; j++) {
    uint32_t v = vals[j];
    for (int off = 32; off; off >>= 1) v += __shfl_down(v, off);
    if (sw_lane() == 0 && v) atomicAdd(&acc[j], v);
  }
  for (int off = 32; off; off >>= 1) { uint32_t o2 = __shfl_down(mindl, off); mindl = o2 < mindl ? o2 : mindl; }
  if (sw_lane() == 0) atomicMin(&acc[CEN_MINDL], mindl);
  __syncthreads();
  if (threadIdx.x < 6) { if (acc[threadIdx.x]) atomicAdd(&D.cen_acc[(size_t)sidx * CEN_WORDS + threadIdx.x], acc[threadIdx.x]); }
  else if (threadIdx.x == CEN_MINDL) atomicMin(&D.cen_acc[(size_t)sidx * CEN_WORDS + CEN_MINDL], acc[CEN_MINDL]);
}

// fold the accumulators of a dirty slot into its cached census and stamp the first-times
__device__ void census_commit(DevRef D, uint32_t sidx, uint32_t now) {
  swim_census* c = &D.census[sidx];
  uint32_t* a = &D.cen_acc[(size_t)sidx * CEN_WORDS];
  c->n_observers = a[CEN_OBS]; c->by_state[0] = a[CEN_ST0]; c->by_state[1] = a[CEN_ST1];
  c->by_state[2] = a[CEN_ST2]; c->by_state[3] = a[CEN_ST3]; c->n_current = a[CEN_CUR];
  D.slot_susp[sidx] = a[CEN_ST1]; D.slot_mindl[sidx] = a[CEN_MINDL];
  for (int j = 0; j < CEN_WORDS; j++) a[j] = j == CEN_MINDL ? NONE : 0;
  if (c->first_suspect_ms == NONE && c->by_state[1]) c->first_suspect_ms = now;
  if (c->first_dead_ms == NONE && (c->by_state[2] || c->by_state[3])) c->first_dead_ms = now;
  if (c->all_dead_ms == NONE && c->n_observers && c->by_state[2] + c->by_state[3] == c->n_observers) c->all_dead_ms = now;
  if (c->all_current_ms == NONE && c->n_observers && D.slot_maxinc[sidx] > 1 && c->n_current == c->n_observers) c->all_current_ms = now;
  D.slot_dirty[sidx] = 0;
}

// collect the ids of replica r whose node word is non-zero (whole block cooperates)
__device__ void rebuild_exceptions(DevRef D, uint32_t r, uint32_t* s_n) {
  if (threadIdx.x == 0) *s_n = 0;
  __syncthreads();
  const uint32_t* nw = D.nw + (size_t)r * D.N;
  for (uint32_t x = threadIdx.x; x < D.N; x += blockDim.x)
    if (nw[x]) { uint32_t pos = atomicAdd(s_n, 1u); if (pos < SW_EXC_MAX) D.exc_ent[(size_t)r * SW_EXC_MAX + pos] = make_uint2(x, nw[x]); }
  __syncthreads();
  if (threadIdx.x == 0) { D.exc_cnt[r] = *s_n; D.exc_dirty[r] = 0; }
  __syncthreads();
}
__global__ void __launch_bounds__(SW_BLOCK) k_exc_rebuild(const SwDev* __restrict__ Dp, uint32_t r) {
  SW_DEV_BIND
  __shared__ uint32_t s_n;
  rebuild_exceptions(D, r, &s_n);
}

__global__ void __launch_bounds__(SW_BLOCK) k_finish(const SwDev* __restrict__ Dp, uint32_t* last_cnt) {
  SW_DEV_BIND
  uint32_t t = *D.tick, now = now_ms(D, t);
  for (uint32_t sidx = threadIdx.x; sidx < D.R * D.S; sidx += SW_BLOCK) {
    uint32_t r = sidx / D.S, sl = sidx % D.S;
    if (sl >= D.n_slots[r]) continue;
    if (D.slot_dirty[sidx]) census_commit(D, sidx, now);
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
  __syncthreads();
  if (threadIdx.x == 0) {
    *D.tick = t + 1;
    for (uint32_t sh = 0; sh < D.n_shards; sh++) { last_cnt[sh] = D.out_cnt[sh]; D.out_cnt[sh] = 0; }
    D.pend_cnt[(t + 1) % (D.TQ + 1)] = 0;      // the list the next tick appends to (just consumed)
    for (uint32_t j = 0; j < SW_PP_LISTS; j++) D.pp_cnt[((t & 1u) * SW_PP_LISTS + j) * 16] = 0;   // answered
  }
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
// initialisation, stimulus, digest
// =================================================================================================
__global__ void k_init_nodes(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  size_t NL = (size_t)D.R * D.nloc, l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= NL) return;
  D.hdr[l] = make_uint4(1, 0, 0, 0);
  D.ph[l] = make_uint2(0, 0);
  D.pr0[l] = make_uint4(NONE, 0, 0, 0);
  D.in_cnt[l] = 0;
  if (D.evseq) D.evseq[l] = 0;
  if (l % SW_BLOCK == 0) {
    size_t rem = NL - l;
    D.q_any[l / SW_BLOCK] = 0; D.in_any[l / SW_BLOCK] = 0; D.alive_cnt[l / SW_BLOCK] = rem < SW_BLOCK ? (uint32_t)rem : SW_BLOCK;
  }
}
__global__ void k_init_views(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  size_t n = (size_t)D.R * D.S * D.nloc, i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  D.va[i] = make_uint4(SW_BASE_KEY, 0, 0, 0); D.vb[i] = make_uint4(0, 0, 0, 0);
}
__global__ void k_init_slots(const SwDev* __restrict__ Dp) {
  SW_DEV_BIND
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= D.R * D.S) return;
  D.subj_node[i] = NONE; D.slot_dirty[i] = 0; D.slot_maxinc[i] = 1; D.slot_susp[i] = 0; D.slot_mindl[i] = NONE;
  for (int j = 0; j < CEN_WORDS; j++) D.cen_acc[(size_t)i * CEN_WORDS + j] = j == CEN_MINDL ? NONE : 0;
  swim_census c; memset(&c, 0, sizeof c);
  c.first_suspect_ms = c.first_dead_ms = c.all_dead_ms = c.all_current_ms = NONE;
  D.census[i] = c;
}

enum { INJ_KILL = 0, INJ_REVIVE = 1, INJ_LEAVE = 2, INJ_UPDATE = 3 };

__global__ void k_inject_alloc(const SwDev* __restrict__ Dp, uint32_t r, const uint32_t* ids, uint32_t n) {
  SW_DEV_BIND
  if (threadIdx.x || blockIdx.x) return;
  for (uint32_t a = 0; a < n; a++) alloc_slot(D, r, ids[a]);
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
      if (local && !(old & NW_INERT)) atomicSub(&D.alive_cnt[l / SW_BLOCK], 1u);     // it was being acted for
    } else if (op == INJ_REVIVE) {
      uint32_t old = atomicAnd(&D.nw[g], ~NW_DEAD);
      if (local) {
        if ((old & NW_DEAD) && !(old & NW_ATTACHED)) atomicAdd(&D.alive_cnt[l / SW_BLOCK], 1u);
        uint2 h = D.ph[l];
        D.pr0[l].x = NONE; D.ph[l].y = p_pack(p_epoch(h.y), p_aw(h.y), 0, 0); D.in_cnt[l] = 0;
      }
    } else if (local && !(D.nw[g] & NW_DEAD)) {
      NodeCtx c(D, S);
      c.r = r; c.o = x; c.k = x - D.i0; c.t = *D.tick; c.l = l; c.NL = (size_t)D.R * D.nloc;
      c.load();
      uint32_t w = D.nw[g];
      if (NW_HAS_SLOT(w)) {
        if (op == INJ_LEAVE) { c.leaving = 1; c.dead_node(x, c.self_inc, x); }   // memberlist.Leave
        else {                                                                    // memberlist.UpdateNode
          c.self_inc++;
          size_t sidx = (size_t)r * D.S + NW_SLOT(w);
          uint4 a = D.va[sidx * D.nloc + c.k];
          c.set_view(sidx, a, c.self_inc, SWIM_STATE_ALIVE, false);
          D.va[sidx * D.nloc + c.k] = a;
          c.broadcast(x, SWIM_MSG_ALIVE, c.self_inc, 1);
        }
        c.store();
        q_bit_lane(D, l, c.q_became_set(), c.q_became_clr());
      }
    }
  }
  if ((op == INJ_KILL || op == INJ_REVIVE) && blockIdx.x == 0)
    for (uint32_t sl = threadIdx.x; sl < D.n_slots[r]; sl += SW_BLOCK) {
      D.slot_dirty[(size_t)r * D.S + sl] = 1;
      // The expire role's gate (suspect count and earliest deadline of the last census) only covers observers that were
      // running then.  A node that comes back resumes its old views, whose suspicion timers may be long overdue: open
      // the gate so that the next tick scans the column; the census at the end of that tick closes it again.
      if (op == INJ_REVIVE) { D.slot_susp[(size_t)r * D.S + sl] = 1; D.slot_mindl[(size_t)r * D.S + sl] = 0; }
    }
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
    uint4 h = D.hdr[l]; h.y = h_pack(h_leaving(h.y), 0, 0); D.hdr[l] = h; D.in_cnt[l] = 0;
    if (!(old & NW_DEAD)) atomicSub(&D.alive_cnt[l / SW_BLOCK], 1u);          // no longer one of the nodes the simulator acts for
    q_bit_lane(D, l, false, true);
  }
}
__global__ void k_set_partition(const SwDev* __restrict__ Dp, uint32_t r, const uint8_t* group) {
  SW_DEV_BIND
  uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= D.N) return;
  size_t g = (size_t)r * D.N + x;
  D.nw[g] = (D.nw[g] & ~0x7F000000u) | (((uint32_t)group[x] & 0x7Fu) << 24);
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
      c.user_event(id, lt);
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
  uint4 h = D.hdr[l], p0 = D.pr0[l]; uint2 p = D.ph[l]; uint32_t w = D.nw[(size_t)r * D.N + i];
  out[0] = h.x; out[1] = h.y; out[2] = h.z; out[3] = h.w;
  out[4] = p0.x; out[5] = p0.y; out[6] = p0.z; out[7] = p0.w; out[8] = p.x; out[9] = p.y; out[10] = w;
  for (uint32_t j = 0; j < h_qlen(h.y) && j < 32; j++) { uint4 e = D.q[(size_t)j * NL + l]; out[16 + 4 * j] = e.x; out[17 + 4 * j] = e.y; out[18 + 4 * j] = e.z; out[19 + 4 * j] = e.w; }
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
    uint4 h = D.hdr[l], p0 = D.pr0[l]; uint2 p = D.ph[l];
    d += sw_h3(1, g, ((uint64_t)h.x << 32) | ((uint64_t)p_aw(p.y) << 8) | h_leaving(h.y));
    d += sw_h3(2, g, ((uint64_t)p.x << 32) | p_epoch(p.y));
    if (p0.x != NONE)
      d += sw_h3(3, g, ((uint64_t)p0.x << 32) | p0.z) + sw_h3(4, g, ((uint64_t)p0.y << 32) | ((uint64_t)p_stage(p.y) << 8) | p_nackm(p.y));
    for (uint32_t j = 0; j < h_qlen(h.y); j++) {
      uint4 e = D.q[(size_t)j * NL + l];
      d += sw_h3(5, g, sw_h3(e.x, ((uint64_t)e.y << 32) | e.z, ((uint64_t)m_seq(e.w) << 16) | ((uint64_t)m_tr(e.w) << 8) | m_type(e.w)));
    }
    d += sw_h3(6, g, ((uint64_t)h.z << 32) | h.w);
    for (uint32_t j = 0; j < h_evqlen(h.y); j++) {
      uint4 e = D.evq[(size_t)j * NL + l];
      d += sw_h3(7, g, sw_h3(e.x, e.y, ((uint64_t)m_seq(e.w) << 16) | ((uint64_t)m_tr(e.w) << 8)));
    }
    if (D.flags & SWIM_F_SERF_EVENTS)
      for (uint32_t b = 0; b < D.EB; b++) {
        uint4 sv = D.ring[(size_t)b * NL + l]; uint32_t n = sv.x >> 30, lt = sv.x & 0x3FFFFFFFu;
        if (n >= 1) d += sw_h3(8, g, ((uint64_t)lt << 32) | sv.y);
        if (n >= 2) d += sw_h3(8, g, ((uint64_t)lt << 32) | sv.z);
        if (n >= 3) d += sw_h3(8, g, ((uint64_t)lt << 32) | sv.w);
      }
  }
  digest_commit(d, out);
}
__global__ void __launch_bounds__(SW_BLOCK) k_digest_views(const SwDev* __restrict__ Dp, unsigned long long* out) {
  SW_DEV_BIND
  uint32_t sidx = blockIdx.y, r = sidx / D.S, sl = sidx % D.S;
  if (sl >= D.n_slots[r]) return;
  uint32_t x = D.subj_node[sidx];
  uint64_t d = 0;
  for (uint32_t k = blockIdx.x * SW_BLOCK + threadIdx.x; k < D.nloc; k += gridDim.x * SW_BLOCK) {
    size_t ci = (size_t)sidx * D.nloc + k;
    uint4 a = D.va[ci];
    uint32_t key = a.x, since = a.y;
    if (key == SW_BASE_KEY && since == 0) continue;
    uint64_t id = ((uint64_t)r << 40) ^ ((uint64_t)x * 0x100000001B3ull) ^ ((uint64_t)(D.i0 + k) << 8);
    d += sw_h3(9, id, ((uint64_t)key << 32) | since);
    if (SW_KST(key) == SWIM_STATE_SUSPECT) {
      uint32_t nc = a.z; uint4 cf = D.vb[ci];
      d += sw_h3(10, id, nc);
      d += sw_h3(11, id, a.w);
      if (nc >= 1) d += sw_h3(12, id, cf.x);
      if (nc >= 2) d += sw_h3(13, id, cf.y);
      if (nc >= 3) d += sw_h3(14, id, cf.z);
    }
  }
  digest_commit(d, out);
}

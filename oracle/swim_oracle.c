/*
 * swim_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A scalar CPU restatement, in plain C, of the algorithm on Consul's Serf/memberlist SWIM hot
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (consul_amd/libswimsim.so) never does.
 *
 * PARITY UNPINNED.  The algorithm lives in two Go modules that are NOT vendored in the
 * reference checkout and cannot be fetched here:
 *     github.com/hashicorp/memberlist v0.6.0   (reference go.mod:80, go.sum:441-442)
 *     github.com/hashicorp/serf       v0.10.4  (reference go.mod:85, go.sum:456-457)
 * and the reference's own tests hold no golden vectors for probe/suspect/gossip rounds
 * (SURVEY.md §8(c)).  This file restates the published algorithm of those modules (state.go,
 * suspicion.go, awareness.go, queue.go, util.go, broadcast.go; serf lamport.go, serf.go
 * handleUserEvent, delegate.go) as recalled in SURVEY.md Appendix A, and is pinned by
 *   - the closed-form constants the reference documents in-tree
 *     (agent/config/runtime.go:1326,1344),
 *   - the Random123 known-answer vectors for Philox4x32-10,
 *   - behavioural known answers written from the upstream semantics (tests/test_oracle_kat.py).
 * Each function cites the upstream function it follows and Consul's call/config site.
 *
 * Determinisation (what a lock-step simulator must add to a wall-clock, goroutine-scheduled
 * original):
 *   time      integer ticks of quantum_ms = gcd(GossipInterval, ProbeInterval, ProbeTimeout)
 *   stagger   triggerFunc's random initial sleep becomes a fixed per-chunk phase:
 *             chunk c = id / phase_chunk; gossip phase = c % G; probe phase = (c / G) % P
 *   rand      math/rand becomes Philox4x32-10, counter (tick, node, block, 0), key (seed+replica,
 *             stream) — so a draw depends only on who draws and when, never on execution order
 *   shuffle   the per-node shuffled probe list becomes a keyed Feistel permutation of [0,N)
 *   arrival   packets sent in tick t are delivered at the end of tick t; each receiver applies
 *             its tick's messages in ascending (subject, type, incarnation, from) order
 *   views     every observer starts with the converged base view (all alive, incarnation 1);
 *             per-observer overrides exist only for "subjects" (nodes somebody has news about)
 */
#define _GNU_SOURCE
#include "../include/swimsim.h"

#define QMAX 4096      /* memberlist's TransmitLimitedQueue is unbounded; the product library stages queue_cap <= 32 entries per node in LDS.  The checker may hold far
                        * more, so that what the bound costs can be measured against (nearly) no bound at all: tools/queue_cap_sweep.py, profiles/r05_queue_cap_sweep.txt */
#define EVQ_MAX 8192   /* serf sizes its event queue max(2N, 4096) (internal/gossip/libserf/serf.go:22-27); the checker can hold that for N <= 4096 */

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* RNG and hashing                                                                             */
/* ------------------------------------------------------------------------------------------ */

enum { STREAM_GOSSIP = 1, STREAM_PERM = 2, STREAM_INDIRECT = 3, STREAM_LOSS = 4, STREAM_PUSHPULL = 5, STREAM_TRUTH = 6, STREAM_RTT = 7, STREAM_COORD = 8, STREAM_RECONNECT = 9 };

/* Philox4x32-10 (Salmon et al., SC'11; Random123).  Pinned by kat vectors in the tests. */
static void philox4x32(const uint32_t c[4], const uint32_t k[2], uint32_t o[4]) {
  uint32_t c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], k0 = k[0], k1 = k[1];
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

static uint32_t fmix32(uint32_t h) { /* murmur3 finaliser */
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}

static uint64_t mix64(uint64_t x) { /* splitmix64 finaliser, for the state digest */
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
  return x;
}

static void stream_key(uint64_t seed_r, uint32_t stream, uint32_t key[2]) {
  key[0] = (uint32_t)seed_r;
  key[1] = (uint32_t)(seed_r >> 32) ^ stream;
}

/* draw #d of (stream, tick, node): word d%4 of block d/4 */
typedef struct { uint32_t key[2], tick, node, blk, w[4]; } draws_t;
static void draws_init(draws_t* d, uint64_t seed_r, uint32_t stream, uint32_t tick, uint32_t node) {
  stream_key(seed_r, stream, d->key); d->tick = tick; d->node = node; d->blk = SWIM_NONE;
}
static uint32_t draws_get(draws_t* d, uint32_t idx) {
  uint32_t b = idx >> 2;
  if (b != d->blk) { uint32_t c[4] = { d->tick, d->node, b, 0 }; philox4x32(c, d->key, d->w); d->blk = b; }
  return d->w[idx & 3];
}

/* Replaces memberlist shuffleNodes + the probeIndex walk (state.go probe/resetNodes, util.go
 * shuffleNodes): position `index` of node's epoch-th shuffle of [0,N).  4-round balanced Feistel
 * on 2*ceil(bits/2) bits with cycle walking, so every epoch visits each id exactly once. */
static uint32_t probe_perm(uint64_t seed_r, uint32_t n, uint32_t node, uint32_t epoch, uint32_t index) {
  if (n <= 1) return 0;
  uint32_t bits = 1; while ((1u << bits) < n && bits < 32) bits++;
  uint32_t h = (bits + 1) / 2, mask = (1u << h) - 1;
  uint32_t key[2], c[4] = { node, epoch, 0, 0x50524D31u }, rk[4];
  stream_key(seed_r, STREAM_PERM, key);
  philox4x32(c, key, rk);
  uint32_t x = index;
  do {
    uint32_t l = x >> h, r = x & mask;
    for (int i = 0; i < 4; i++) { uint32_t t = l ^ (fmix32(r ^ rk[i]) & mask); l = r; r = t; }
    x = (l << h) | r;
  } while (x >= n);
  return x;
}

/* ------------------------------------------------------------------------------------------ */
/* Closed-form constants (memberlist util.go, suspicion.go)                                    */
/* ------------------------------------------------------------------------------------------ */

/* Go's math.Log2 / math.Log10 (pure-Go definitions: Log2 via Frexp, Log10 = Log2 * Ln2/Ln10), so
 * truncation edges reproduce (pinned by upstream's util_test.go table in tests/test_oracle_kat.py). */
static double go_log2(double x) {
  int e; double f = frexp(x, &e);
  if (f == 0.5) return (double)(e - 1);
  return log(f) * (1.0 / 0.693147180559945309417232121458176568) + (double)e;
}
/* Go folds the constant expression Ln2/Ln10 exactly and rounds ONCE (0x1.34413509f79ffp-2); dividing
 * the two rounded doubles gives ...79fep-2 and turns log10(100) into 1.9999999999999998, which would
 * contradict upstream's own util_test.go table (suspicionTimeout(3, 100, 1s)/3 == 2000 ms). */
static double go_log10(double x) { return go_log2(x) * 0x1.34413509f79ffp-2; }

static uint32_t gcd_u32(uint32_t a, uint32_t b) { while (b) { uint32_t t = a % b; a = b; b = t; } return a; }

/* suspicion.go remainingSuspicionTime, in integer milliseconds */
static int64_t remaining_suspicion_ms(uint32_t n, uint32_t k, int64_t elapsed_ms, int64_t min_ms, int64_t max_ms) {
  double frac = log((double)n + 1.0) / log((double)k + 1.0);
  double max_s = (double)max_ms / 1000.0, min_s = (double)min_ms / 1000.0;
  double raw = max_s - frac * (max_s - min_s);
  int64_t timeout = (int64_t)floor(1000.0 * raw);
  if (timeout < min_ms) timeout = min_ms;
  return timeout - elapsed_ms;
}

int swim_config_preset(swim_config* c, int preset) {
  if (!c) return SWIM_EINVAL;
  memset(c, 0, sizeof *c);
  c->abi_version = SWIM_ABI_VERSION;
  c->n_nodes = 128; c->n_replicas = 1;
  /* memberlist config.go DefaultLANConfig; WAN/Local override below (SURVEY Appendix A.2;
   * LAN/WAN knobs corroborated at agent/config/runtime.go:1285-1427) */
  c->indirect_checks = 3; c->retransmit_mult = 4; c->suspicion_mult = 4;
  c->suspicion_max_timeout_mult = 6; c->probe_timeout_ms = 500; c->probe_interval_ms = 1000;
  c->awareness_max_mult = 8; c->gossip_nodes = 3; c->gossip_interval_ms = 200;
  c->gossip_to_dead_ms = 30000; c->udp_buffer_size = 1400; c->push_pull_interval_ms = 30000;
  if (preset == SWIM_PRESET_WAN) {
    c->suspicion_mult = 6; c->probe_timeout_ms = 3000; c->probe_interval_ms = 5000;
    c->gossip_nodes = 4; c->gossip_interval_ms = 500; c->gossip_to_dead_ms = 60000; c->push_pull_interval_ms = 60000;
  } else if (preset == SWIM_PRESET_LOCAL) {
    c->indirect_checks = 1; c->retransmit_mult = 2; c->suspicion_mult = 3;
    c->probe_timeout_ms = 200; c->gossip_interval_ms = 100; c->gossip_to_dead_ms = 15000; c->push_pull_interval_ms = 15000;
  } else if (preset != SWIM_PRESET_LAN) return SWIM_EINVAL;
  c->msg_len[SWIM_MSG_ALIVE] = 128; c->msg_len[SWIM_MSG_SUSPECT] = 48;
  c->msg_len[SWIM_MSG_DEAD] = 48; c->msg_len[SWIM_MSG_USER] = 64;
  /* ping{SeqNo,Node,SourceAddr,SourcePort,SourceNode}, indirectPingReq{+Target,Port,Nack}, ackResp{SeqNo,Payload}
   * (serf's ping delegate puts a coordinate in Payload), nackResp{SeqNo} — msgpack with field names */
  c->ctl_len[SWIM_CTL_PING] = 86; c->ctl_len[SWIM_CTL_INDIRECT] = 122; c->ctl_len[SWIM_CTL_ACK] = 108; c->ctl_len[SWIM_CTL_NACK] = 13;
  c->queue_cap = 8; c->inbox_cap = 32; c->subject_cap = 8; c->view_cap = 0; c->fold_interval_ms = 0;
  c->event_queue_cap = 8; c->event_buffer = 512;
  c->flags = SWIM_F_DEFAULT; c->watch_node = 0; c->n_shards = 1; c->seed = 1;
  c->rtt_scale_us = 40000; c->rtt_height_us = 2000; c->rtt_jitter_us = 0;
  return SWIM_OK;
}

static int validate(const swim_config* c) {
  if (!c || c->abi_version != SWIM_ABI_VERSION) return SWIM_EINVAL;
  if (c->n_nodes < 2 || c->n_replicas < 1) return SWIM_EINVAL;
  if ((uint64_t)c->n_nodes * c->n_replicas >= 0xFFFFFFFFull || c->n_nodes >= (1u << 28)) return SWIM_ERANGE;
  if (c->view_cap > (1u << 20)) return SWIM_ERANGE;
  if (!c->gossip_interval_ms || !c->probe_interval_ms || !c->probe_timeout_ms) return SWIM_EINVAL;
  if (c->gossip_nodes < 1 || c->gossip_nodes > 8 || c->indirect_checks > 8) return SWIM_EINVAL;
  if (c->suspicion_mult < 1 || c->suspicion_mult > 6 || c->retransmit_mult < 1) return SWIM_EINVAL;
  if (c->awareness_max_mult < 1 || c->awareness_max_mult > 255) return SWIM_EINVAL;
  if (c->queue_cap < 1 || c->queue_cap > QMAX || c->inbox_cap < 1 || c->subject_cap < 1) return SWIM_EINVAL;   /* (the product library: <= 32) */
  if (c->flags & SWIM_F_SERF_EVENTS)
    if (c->event_queue_cap < 1 || c->event_queue_cap > EVQ_MAX || c->event_buffer < 1 || c->event_ids_per_ltime > 254) return SWIM_EINVAL;   /* (the product library: <= 32) */
  if (c->n_shards < 1 || c->shard_rank >= c->n_shards || c->n_nodes % c->n_shards) return SWIM_EINVAL;
  if (c->n_initial > c->n_nodes || c->n_initial == 1) return SWIM_EINVAL;
  if (c->phase_chunk & (c->phase_chunk - 1)) return SWIM_EINVAL;
  if ((c->flags & SWIM_F_COORDINATES) && (c->n_shards != 1 || c->rtt_scale_us > 10000000u || c->rtt_height_us > 1000000u || c->rtt_jitter_us > 1000000u)) return SWIM_EINVAL;
  return SWIM_OK;
}

int swim_config_derive(const swim_config* c, swim_derived* d) {
  int rc = validate(c); if (rc) return rc;
  if (!d) return SWIM_EINVAL;
  memset(d, 0, sizeof *d);
  uint32_t q = c->quantum_ms ? c->quantum_ms
             : gcd_u32(gcd_u32(c->gossip_interval_ms, c->probe_interval_ms), c->probe_timeout_ms);
  if (c->gossip_interval_ms % q || c->probe_interval_ms % q || c->probe_timeout_ms % q) return SWIM_EINVAL;
  d->quantum_ms = q;
  d->gossip_period = c->gossip_interval_ms / q;
  d->probe_period = c->probe_interval_ms / q;
  d->probe_timeout_ticks = c->probe_timeout_ms / q;
  uint32_t gp = d->gossip_period * d->probe_period, ch = c->phase_chunk;
  if (!ch) { ch = 256; while (ch > 1 && (uint64_t)ch * gp * 8 > c->n_nodes) ch >>= 1; }
  d->phase_chunk = ch;
  double n = (double)c->n_nodes;
  /* util.go retransmitLimit: mult * ceil(log10(n+1))          (doc: runtime.go:1344) */
  d->retransmit_limit = c->retransmit_mult * (uint32_t)ceil(go_log10(n + 1.0));
  /* util.go suspicionTimeout: mult * int(max(1,log10(max(1,n)))*1000) * interval / 1000
   * (doc: runtime.go:1326), evaluated in integer nanoseconds like time.Duration */
  double scale = go_log10(n < 1.0 ? 1.0 : n); if (scale < 1.0) scale = 1.0;
  int64_t scale_milli = (int64_t)(scale * 1000.0);
  d->node_scale_milli = (uint32_t)scale_milli;
  int64_t min_ns = (int64_t)c->suspicion_mult * scale_milli * ((int64_t)c->probe_interval_ms * 1000000) / 1000;
  int64_t max_ns = (int64_t)c->suspicion_max_timeout_mult * min_ns;
  int64_t min_ms = min_ns / 1000000, max_ms = max_ns / 1000000;
  if (max_ms > 0x7FFFFFFF) return SWIM_ERANGE;
  d->suspicion_min_ms = (uint32_t)min_ms; d->suspicion_max_ms = (uint32_t)max_ms;
  /* state.go suspectNode: k = SuspicionMult-2; if n-2 < k then k = 0 */
  int32_t k = (int32_t)c->suspicion_mult - 2; if (k < 0) k = 0;
  if ((int64_t)c->n_nodes - 2 < k) k = 0;
  d->suspicion_k = (uint32_t)k;
  /* suspicion.go newSuspicion: timeout = max, or min when k < 1; Confirm: remainingSuspicionTime */
  d->suspicion_timeout_ms[0] = (uint32_t)(k < 1 ? min_ms : max_ms);
  for (int32_t i = 1; i <= k && i < 8; i++)
    d->suspicion_timeout_ms[i] = (uint32_t)remaining_suspicion_ms((uint32_t)i, (uint32_t)k, 0, min_ms, max_ms);
  /* util.go pushPullScale multiplier */
  d->push_pull_scale = c->n_nodes <= 32 ? 1u : (uint32_t)(ceil(go_log2(n) - go_log2(32.0)) + 1.0);
  /* state.go pushPullTrigger: every pushPullScale(PushPullInterval, n), after a random stagger */
  { uint64_t per = (uint64_t)c->push_pull_interval_ms * d->push_pull_scale / q;
    if (per > 0x7FFFFFFFull) return SWIM_ERANGE;
    d->push_pull_period_ticks = (uint32_t)per; }
  /* state.go gossip(): bytesAvail = UDPBufferSize - compoundHeaderOverhead(2) - labelOverhead(0) */
  d->packet_budget = c->udp_buffer_size > 2 ? c->udp_buffer_size - 2 : 0;
  if (d->retransmit_limit > 255) return SWIM_ERANGE;       /* transmits is an 8-bit field */
  d->view_cap = c->view_cap ? c->view_cap : (c->n_nodes < 32 ? c->n_nodes : 32);
  d->fold_period_ticks = (c->fold_interval_ms + q - 1) / q;
  d->reap_period_ticks = (c->reap_interval_ms + q - 1) / q;
  d->reconnect_period_ticks = (c->reconnect_interval_ms + q - 1) / q;
  return SWIM_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* State                                                                                       */
/* ------------------------------------------------------------------------------------------ */

#define CONF_MAX 4

typedef struct { uint32_t subject, inc, from, seq; uint8_t type, transmits; } qent;

/* one explicit view: what an observer knows about `subj` beyond the replica's base row */
/* n0 = estNumNodes() when the suspicion started; reaped = serf erased the member (handleReap / prune); leaving = a leave intent was seen while
 * the member was alive here (serf StatusLeaving); slt = serf's member.statusLTime: the Lamport time of the last join / leave intent applied
 * to the member here (0: none) */
typedef struct { uint32_t subj, key, since, conf[CONF_MAX], n0, slt; uint8_t nconf, reaped, leaving; } view_t;
/* an observer's explicit views: open addressing (linear probing, backward-shift deletion), grown on demand, at most
 * cfg.view_cap entries (+1 for the node's view of itself).  Layout is private to this file: everything observable
 * (digest, census, members) is keyed by (observer, subject). */
typedef struct { view_t* e; uint32_t n, slots; } vtab;
#define V_EMPTY 0xFFFFFFFFu
#define FOLD_POISON 0xFFFFFFFFu
#define KEY(inc, st) (((uint32_t)(inc) << 2) | (uint32_t)(st))
#define KINC(k) ((k) >> 2)
#define KST(k) ((k) & 3u)
#define BASE_KEY KEY(1, SWIM_STATE_ALIVE)

typedef uint32_t evslot;      /* an event-buffer slot is `ev_words` words: ltime, n, then n ids (serf: an unbounded list per LTime; ours: cfg.event_ids_per_ltime, rounded up to 4k + 2) */

typedef struct {
  uint32_t self_inc;
  uint8_t awareness, leaving;
  uint8_t serf_leaving;                        /* this agent has broadcast its own leave intent (serf.Leave: state SerfLeaving) — it no longer refutes one */
  uint32_t self_slt;                           /* statusLTime of the agent's own member entry */
  uint32_t qlen, qseq, qcap; qent* q;           /* [queue_cap] in one slab for all nodes; SWIM_F_UNBOUNDED_QUEUE: an array of its own that grows (qcap entries so far) */
  uint32_t pr_target, pr_inc, pr_t0, pr_deadline, pr_cursor, pr_epoch; uint8_t pr_stage, pr_nack_miss;
  /* serf */
  uint32_t ev_clock, evqlen, evqseq; qent* evq; evslot* ring;
  /* per-tick inbox */
  uint32_t in_cnt; swim_edge* inbox;
  vtab vt; uint32_t vdl;          /* explicit views; earliest suspicion deadline among them (a lower bound, SWIM_NONE = none) */
  uint32_t nk;                    /* explicit views of nodes the base row has never heard of (estNumNodes = base_known + nk) */
} node_t;

/* serf/coordinate client state of one node (SWIM_F_COORDINATES) */
#define COORD_WINDOW 20          /* AdjustmentWindowSize */
#define COORD_FILTER 3           /* LatencyFilterSize */
#define COORD_PEERS 16           /* peers whose latency samples a node retains (serf's map is unbounded: DESIGN §8) */
typedef struct { uint32_t peer, n, s[COORD_FILTER], last; } lf_ent;   /* samples in microseconds, oldest first; last = tick of last use */
typedef struct { swim_coordinate c; double adj[COORD_WINDOW]; uint32_t adj_idx; lf_ent lf[COORD_PEERS]; } coord_state;

/* a watch slot: census, first-* stamps and trace of one subject (observation only; the protocol never looks here) */
typedef struct {
  uint32_t node;            /* subject id */
  uint8_t dirty;
  uint32_t max_inc;
  swim_census census;       /* cached; first_* fields persistent */
  uint32_t* trace;          /* [trace_ticks][5] */
} slot_t;

typedef struct { swim_edge* v; uint32_t n, cap; } edgevec;

struct swim_sim {
  swim_config cfg; swim_derived d;
  uint32_t N, R, nloc, i0, tick; int in_tick;
  uint32_t ev_words;             /* words per event-buffer slot: ltime, n, ids */
  uint32_t *ev_watch, *n_ev_watch, ev_observer;   /* swim_watch_events: [R][SWIM_EVENT_WATCHERS] observers with an EventCh of their own; whose event record_event writes next */
  uint8_t *gt_alive, *part;      /* [R*N] replicated ground truth */
  uint8_t *tcp_cls;              /* [R*N] swim_set_tcp_class (DisableTcpPingsForNode) */
  uint8_t* attached;             /* [R*N] driven from outside through the transport bridge */
  uint8_t* alone;                /* [R*N] replicated: started by swim_inject_join, join push-pull not carried out (yet): it knows nobody */
  edgevec captured;              /* rumours sent to attached nodes: {dst = replica*N+attached, ..}, src kept in cap_src */
  uint32_t* cap_src; uint32_t cap_src_cap;
  uint32_t* node_slot;           /* [R*N] replicated: watch slot of a subject, SWIM_NONE = not watched */
  uint32_t* base_key;            /* [R*N] replicated: the view every observer holds unless it has an explicit one */
  uint32_t* subj_cnt;            /* [R*N] explicit views of this subject held by the local observers */
  uint32_t* base_known;          /* [R] nodes the base row has heard of (incarnation > 0) */
  int dyn;                       /* n_initial < n_nodes: estNumNodes() is per observer */
  uint32_t* join_list; uint32_t n_join_pending, join_cap;   /* {replica*N + node, via} of the nodes started since the last tick (every shard lists all of them) */
  uint32_t *f_cnt, *f_kmin, *f_kmax; uint8_t* f_bad; uint32_t* f_touched; uint32_t f_ntouched, f_cap;  /* fold accumulators [R*N] */
  node_t* nodes;                 /* [R*nloc] */
  qent *q_slab, *evq_slab;       /* the nodes' queues, contiguous (q_slab: NULL with SWIM_F_UNBOUNDED_QUEUE) */
  int unbounded;                 /* SWIM_F_UNBOUNDED_QUEUE */
  uint32_t* q_cnt;               /* [R*N] unbounded queues only: rumours about this subject queued at the shard's nodes (a subject is not folded while there is one) */
  swim_edge* inbox_slab;
  slot_t* slots; uint32_t* n_slots; /* [R*S], [R] */
  edgevec* out;                  /* [n_shards] */
  edgevec in, last_edges;
  edgevec carry[2];              /* piggy-backed broadcasts picked in tick t travel with tick t+1's packets */
  edgevec pp_reply[2];           /* push-pull requests seen in tick t are answered in tick t+1: {dst=replier, subject=requester, incarnation=replica} */
  swim_event* events; size_t n_events, cap_events, n_sorted;   /* n_sorted: the events a poll has already put in delivery order */
  swim_stats_t st;
  uint32_t loss_q32;
  /* swim_xchg_*: the other shards of the population (same process) and ticks one of them already ran on our behalf */
  struct swim_sim** xpeers; uint32_t xcredit;
  coord_state* cs;               /* [R*nloc] SWIM_F_COORDINATES */
  swim_coordinate* c_new; uint32_t* c_list; uint32_t c_n;   /* this tick's updates: computed from the coordinates as of the start of the tick, committed at its end */
  char err[256];
};

static void ev_push(edgevec* e, swim_edge x) {
  if (e->n == e->cap) { e->cap = e->cap ? e->cap * 2 : 1024; e->v = (swim_edge*)realloc(e->v, (size_t)e->cap * sizeof(swim_edge)); }
  e->v[e->n++] = x;
}

static inline uint32_t now_ms(const swim_sim* s) { return s->tick * s->d.quantum_ms; }
static inline uint64_t seed_of(const swim_sim* s, uint32_t r) { return s->cfg.seed + r; }
static inline int is_local(const swim_sim* s, uint32_t i) { return i >= s->i0 && i < s->i0 + s->nloc; }
static inline node_t* node_at(swim_sim* s, uint32_t r, uint32_t i) { return &s->nodes[(size_t)r * s->nloc + (i - s->i0)]; }
static inline uint32_t shard_of(const swim_sim* s, uint32_t i) { return i / s->nloc; }
/* does the simulator act for this node (running and not driven from outside) */
static inline int acts(const swim_sim* s, uint32_t r, uint32_t i) { size_t g = (size_t)r * s->N + i; return s->gt_alive[g] && !s->attached[g] && !s->alone[g]; }
static inline uint32_t gphase_of(const swim_sim* s, uint32_t i) { return (i / s->d.phase_chunk) % s->d.gossip_period; }
static inline uint32_t pphase_of(const swim_sim* s, uint32_t i) { return (i / s->d.phase_chunk / s->d.gossip_period) % s->d.probe_period; }

static int watching(swim_sim* s, uint32_t r, uint32_t o);
static void record_event(swim_sim* s, uint32_t r, uint32_t type, uint32_t node, uint32_t ltime, uint32_t inc) {
  if (s->n_events == s->cap_events) {
    s->cap_events = s->cap_events ? s->cap_events * 2 : 256;
    s->events = (swim_event*)realloc(s->events, s->cap_events * sizeof(swim_event));
  }
  swim_event e = { now_ms(s), r, type, node, inc, s->ev_observer, ltime };
  s->events[s->n_events++] = e;
}
/* does observer o of replica r have an EventCh (cfg.watch_node, or added with swim_watch_events)?  Notes whose event comes next. */
static int watching(swim_sim* s, uint32_t r, uint32_t o) {
  s->ev_observer = o;
  if (o == s->cfg.watch_node) return 1;
  for (uint32_t j = 0; j < s->n_ev_watch[r]; j++) if (s->ev_watch[(size_t)r * SWIM_EVENT_WATCHERS + j] == o) return 1;
  return 0;
}

/* ---- an observer's explicit views --------------------------------------------------------------- */
static inline uint32_t vt_home(const vtab* t, uint32_t x) { return fmix32(x) & (t->slots - 1); }
static view_t* vt_find(const vtab* t, uint32_t x) {
  if (!t->n) return NULL;
  for (uint32_t i = vt_home(t, x);; i = (i + 1) & (t->slots - 1)) {
    if (t->e[i].subj == x) return &t->e[i];
    if (t->e[i].subj == V_EMPTY) return NULL;
  }
}
static int vt_grow(vtab* t) {
  uint32_t ns = t->slots ? t->slots * 2 : 4;
  view_t* ne = (view_t*)malloc((size_t)ns * sizeof(view_t)); if (!ne) return SWIM_ENOMEM;
  for (uint32_t i = 0; i < ns; i++) ne[i].subj = V_EMPTY;
  vtab old = *t; t->e = ne; t->slots = ns;
  for (uint32_t i = 0; i < old.slots; i++) if (old.e[i].subj != V_EMPTY) {
    uint32_t j = vt_home(t, old.e[i].subj); while (t->e[j].subj != V_EMPTY) j = (j + 1) & (ns - 1);
    t->e[j] = old.e[i];
  }
  free(old.e); return SWIM_OK;
}
static view_t* vt_insert(vtab* t, uint32_t x) {            /* x is known to be absent */
  if ((t->n + 1) * 2 > t->slots && vt_grow(t)) return NULL;
  uint32_t j = vt_home(t, x); while (t->e[j].subj != V_EMPTY) j = (j + 1) & (t->slots - 1);
  memset(&t->e[j], 0, sizeof(view_t)); t->e[j].subj = x; t->n++;
  return &t->e[j];
}
static void vt_erase(vtab* t, view_t* v) {                 /* backward-shift deletion */
  uint32_t m = t->slots - 1, i = (uint32_t)(v - t->e), j = i;
  for (;;) {
    j = (j + 1) & m;
    if (t->e[j].subj == V_EMPTY) break;
    uint32_t k = vt_home(t, t->e[j].subj);
    if (i <= j ? (i < k && k <= j) : (i < k || k <= j)) continue;
    t->e[i] = t->e[j]; i = j;
  }
  t->e[i].subj = V_EMPTY; t->n--;
}

/* estNumNodes(): how many nodes this observer has heard of (alive or not) — the cluster size every scaling law uses */
static inline uint32_t est_n(const swim_sim* s, uint32_t r, const node_t* nd) { return s->dyn ? s->base_known[r] + nd->nk : s->N; }
/* util.go retransmitLimit for an observer that knows n nodes */
static uint32_t retransmit_limit_n(const swim_sim* s, uint32_t n) {
  if (!s->dyn) return s->d.retransmit_limit;
  return s->cfg.retransmit_mult * (uint32_t)ceil(go_log10((double)n + 1.0));
}
/* suspicion.go: timeout of a suspicion that started when the observer knew n nodes, after `nconf` confirmations; and
 * the k of that timer (SuspicionMult-2, or 0 when n-2 < k) */
static uint32_t suspicion_k_n(const swim_sim* s, uint32_t n) {
  if (!s->dyn) return s->d.suspicion_k;
  int32_t k = (int32_t)s->cfg.suspicion_mult - 2; if (k < 0) k = 0;
  if ((int64_t)n - 2 < k) k = 0;
  return (uint32_t)k;
}
static uint32_t suspicion_timeout_n(const swim_sim* s, uint32_t n, uint32_t nconf) {
  if (!s->dyn) return s->d.suspicion_timeout_ms[nconf];
  double scale = go_log10(n < 1 ? 1.0 : (double)n); if (scale < 1.0) scale = 1.0;
  int64_t scale_milli = (int64_t)(scale * 1000.0);
  int64_t min_ms = (int64_t)s->cfg.suspicion_mult * scale_milli * ((int64_t)s->cfg.probe_interval_ms * 1000000) / 1000 / 1000000;
  int64_t max_ms = (int64_t)s->cfg.suspicion_max_timeout_mult * min_ms;
  uint32_t k = suspicion_k_n(s, n);
  if (k < 1) return (uint32_t)min_ms;
  if (nconf == 0) return (uint32_t)max_ms;
  if (nconf >= k) return (uint32_t)min_ms;
  return (uint32_t)remaining_suspicion_ms(nconf, k, 0, min_ms, max_ms);
}

/* observer o's explicit view of subject x, or NULL: then it holds the base row's */
static view_t* view_ptr(swim_sim* s, uint32_t r, uint32_t o, uint32_t x) {
  if (!s->subj_cnt[(size_t)r * s->N + x]) return NULL;     /* nobody here has news about x */
  return vt_find(&node_at(s, r, o)->vt, x);
}
/* what o holds about x without an explicit view: the base row's — except that a node always sees ITSELF alive at
 * its own incarnation (the base row may have moved on while it was away) */
static inline uint32_t implicit_key(swim_sim* s, uint32_t r, uint32_t o, uint32_t x) {
  return x == o ? KEY(node_at(s, r, o)->self_inc, SWIM_STATE_ALIVE) : s->base_key[(size_t)r * s->N + x];
}
static uint32_t view_key(swim_sim* s, uint32_t r, uint32_t o, uint32_t x, uint32_t* since) {
  view_t* v = view_ptr(s, r, o, x);
  if (!v) { if (since) *since = 0; return implicit_key(s, r, o, x); }
  if (since) *since = v->since;
  return v->key;
}
static void touch_slot(swim_sim* s, uint32_t r, uint32_t x);
/* the explicit view of x at o, created from the base row when o has none yet.  NULL = o already holds view_cap
 * explicit views (its view of itself always fits): the caller ignores the rumour, counted in view_drops. */
static view_t* view_make(swim_sim* s, uint32_t r, uint32_t o, uint32_t x) {
  node_t* nd = node_at(s, r, o);
  view_t* v = vt_find(&nd->vt, x);
  if (v) return v;
  if (nd->vt.n >= s->d.view_cap + (x == o ? 1u : 0u)) {
    /* Full.  memberlist's resetNodes forgets a node that has been dead for longer than GossipToTheDeadTime; so does a
     * full table, one node at a time: the longest-settled Dead/Left view (ties: lowest id; never the node's view of
     * itself) makes room and the observer falls back to the base row for that subject.  Nothing that old: drop. */
    view_t* victim = NULL; uint32_t now = now_ms(s);
    for (uint32_t i = 0; i < nd->vt.slots; i++) {
      view_t* c = &nd->vt.e[i];
      if (c->subj == V_EMPTY || c->subj == o || KST(c->key) < SWIM_STATE_DEAD || !(now - c->since > s->cfg.gossip_to_dead_ms)) continue;
      if (!victim || c->since < victim->since || (c->since == victim->since && c->subj < victim->subj)) victim = c;
    }
    if (!victim) { s->st.view_drops++; return NULL; }
    s->subj_cnt[(size_t)r * s->N + victim->subj]--; touch_slot(s, r, victim->subj);
    if (KINC(s->base_key[(size_t)r * s->N + victim->subj]) == 0) nd->nk--;
    vt_erase(&nd->vt, victim); s->st.view_evictions++;
  }
  v = vt_insert(&nd->vt, x);
  if (!v) { s->st.view_drops++; return NULL; }
  v->key = implicit_key(s, r, o, x);
  s->subj_cnt[(size_t)r * s->N + x]++;
  if (x != o && KINC(s->base_key[(size_t)r * s->N + x]) == 0) nd->nk++;   /* a node the base row has never heard of: this observer now has (itself it counts from the start) */
  return v;
}

/* swim_watch: census / first-* stamps / trace of subject x from now on */
static uint32_t current_max_inc(swim_sim* s, uint32_t r, uint32_t x) {
  uint32_t m = KINC(s->base_key[(size_t)r * s->N + x]);
  if (s->subj_cnt[(size_t)r * s->N + x])
    for (uint32_t k = 0; k < s->nloc; k++) { view_t* v = vt_find(&s->nodes[(size_t)r * s->nloc + k].vt, x); if (v && KINC(v->key) > m) m = KINC(v->key); }
  return m;
}
static int alloc_slot(swim_sim* s, uint32_t r, uint32_t x) {
  size_t g = (size_t)r * s->N + x;
  if (s->node_slot[g] != SWIM_NONE) return SWIM_OK;
  if (s->n_slots[r] >= s->cfg.subject_cap) { s->st.subject_overflow++; return SWIM_EOVERFLOW; }
  uint32_t sl = s->n_slots[r]++;
  slot_t* t = &s->slots[(size_t)r * s->cfg.subject_cap + sl];
  t->node = x; t->dirty = 1; t->max_inc = current_max_inc(s, r, x);
  memset(&t->census, 0, sizeof t->census);
  t->census.first_suspect_ms = t->census.first_dead_ms = t->census.all_dead_ms = t->census.all_current_ms = SWIM_NONE;
  if (s->cfg.trace_ticks) t->trace = (uint32_t*)calloc((size_t)s->cfg.trace_ticks * 5, sizeof(uint32_t));
  s->node_slot[g] = sl;
  return SWIM_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* TransmitLimitedQueue (memberlist queue.go; sized by Consul at internal/gossip/libserf/      */
/* serf.go:22-27, RetransmitMult from agent/agent.go:1428)                                     */
/* ------------------------------------------------------------------------------------------ */

static inline uint32_t ent_len(const swim_sim* s, const qent* e) { return s->cfg.msg_len[e->type & 3]; }

/* limitedBroadcast.Less: transmits asc, then msgLen desc, then id desc */
static int ent_before(const swim_sim* s, const qent* a, const qent* b) {
  if (a->transmits != b->transmits) return a->transmits < b->transmits;
  uint32_t la = ent_len(s, a), lb = ent_len(s, b);
  if (la != lb) return la > lb;
  return a->seq > b->seq;
}

/* A queue is kept SORTED by that order (upstream keeps a btree): q[0] is what GetBroadcasts looks at first, q[n-1] what Prune() drops.
 * Position of the first entry that does not sort before e. */
static uint32_t queue_lower(const swim_sim* s, const qent* q, uint32_t n, const qent* e) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) { uint32_t mid = (lo + hi) / 2; if (ent_before(s, &q[mid], e)) lo = mid + 1; else hi = mid; }
  return lo;
}
/* a node's memberlist queue: its entries, how many, and — SWIM_F_UNBOUNDED_QUEUE — how many the array can hold before it grows */
typedef struct { qent** q; uint32_t *len, *seq, *dyn_cap; uint32_t cap; uint32_t* subj_queued; } qref;

/* QueueBroadcast: a memberlistBroadcast named by its node invalidates any queued rumour about
 * the same node (broadcast.go Invalidates); serf user events are UniqueBroadcasts.  The real
 * queue is unbounded (so is ours with SWIM_F_UNBOUNDED_QUEUE: qr.dyn_cap); otherwise it holds `cap` entries and, like Prune(), evicts
 * the tail of the order.  subj_queued (fold rule, unbounded queues only): queued rumours per subject over the shard's nodes. */
static void queue_push(swim_sim* s, qref qr, int named, uint32_t subject, uint8_t type, uint32_t inc, uint32_t from, uint64_t* drops) {
  qent* q = *qr.q; uint32_t n = *qr.len;
  if (named)
    for (uint32_t i = 0; i < n; i++)
      if (q[i].subject == subject) { memmove(q + i, q + i + 1, (size_t)(n - i - 1) * sizeof(qent)); n--; if (qr.subj_queued) qr.subj_queued[subject]--; break; }
  qent e = { subject, inc, from, (*qr.seq)++ & 0x3FFFFFu, type, 0 };
  uint32_t pos = queue_lower(s, q, n, &e);
  if (qr.dyn_cap) {
    if (n == *qr.dyn_cap) { *qr.dyn_cap = n ? n * 2 : 8; q = *qr.q = (qent*)realloc(q, (size_t)*qr.dyn_cap * sizeof(qent)); if (!q) abort(); }
  } else if (n == qr.cap) {
    (*drops)++;
    if (pos == n) { *qr.len = n; return; }                /* the new entry takes part in the prune: it sorts last itself */
    n--; if (qr.subj_queued) qr.subj_queued[q[n].subject]--;
  }
  memmove(q + pos + 1, q + pos, (size_t)(n - pos) * sizeof(qent));
  q[pos] = e; *qr.len = n + 1;
  if (qr.subj_queued) qr.subj_queued[subject]++;
}

/* GetBroadcasts(overhead, limit): walk tiers by transmit count, inside a tier largest first then
 * newest, take what fits (the space left only shrinks: one walk down the order), bump transmits after the sweep, retire at
 * retransmitLimit.  `out` receives the picks in the order they were picked. */
static uint32_t queue_get(swim_sim* s, qref qr, uint32_t overhead, int32_t limit, qent* out, int32_t* used_out, uint32_t retransmit_limit) {
  qent* q = *qr.q; uint32_t n = *qr.len, cnt = 0; int32_t used = 0;
  static _Thread_local uint32_t taken[2048];               /* a pick costs at least `overhead` (>= 2) of at most 65 535 bytes... of a 1 400-byte packet */
  for (uint32_t i = 0; i < n && cnt < 2048; i++) {
    int32_t free_b = limit - used - (int32_t)overhead;
    if (free_b <= 0) break;
    if ((int32_t)ent_len(s, &q[i]) > free_b) continue;
    taken[cnt] = i; used += (int32_t)(overhead + ent_len(s, &q[i])); out[cnt++] = q[i];
  }
  *used_out = used;
  if (!cnt) return 0;
  /* take the picks out (stable), then put each back where its new transmit count sorts — unless it is Finished() */
  uint32_t m = taken[0], t = 0;
  for (uint32_t i = taken[0]; i < n; i++) { if (t < cnt && taken[t] == i) { t++; continue; } q[m++] = q[i]; }
  for (uint32_t j = 0; j < cnt; j++) {
    qent e = out[j];
    if ((uint32_t)e.transmits + 1 >= retransmit_limit) { if (qr.subj_queued) qr.subj_queued[e.subject]--; continue; }
    e.transmits++;
    uint32_t pos = queue_lower(s, q, m, &e);
    memmove(q + pos + 1, q + pos, (size_t)(m - pos) * sizeof(qent));
    q[pos] = e; m++;
  }
  *qr.len = m;
  return cnt;
}

/* where a packet's picks are collected: queue_get takes at most *qlen entries, so both queues whole always fit (a stack array of
 * 2 * QMAX entries was only safe while the event queue was as shallow as the product library's: ADVICE r4) */
static qent* pick_buf(void) { static _Thread_local qent buf[2 * 2048]; return buf; }   /* queue_get takes at most 2 048 entries per queue */

static inline qref mlq(swim_sim* s, uint32_t r, node_t* nd) {
  qref q = { &nd->q, &nd->qlen, &nd->qseq, s->unbounded ? &nd->qcap : NULL, s->cfg.queue_cap, s->q_cnt ? s->q_cnt + (size_t)r * s->N : NULL };
  return q;
}
static inline qref evq_of(swim_sim* s, node_t* nd) { qref q = { &nd->evq, &nd->evqlen, &nd->evqseq, NULL, s->cfg.event_queue_cap, NULL }; return q; }
/* encodeAndBroadcast (broadcast.go) */
static void broadcast(swim_sim* s, node_t* nd, uint32_t subject, uint8_t type, uint32_t inc, uint32_t from) {
  const uint32_t r = (uint32_t)((size_t)(nd - s->nodes) / s->nloc);
  queue_push(s, mlq(s, r, nd), 1, subject, type, inc, from, &s->st.queue_drops);
}

/* ------------------------------------------------------------------------------------------ */
/* awareness.go                                                                                */
/* ------------------------------------------------------------------------------------------ */
static int awareness_next(int max_mult, int score, int delta) {       /* awareness.ApplyDelta */
  int v = score + delta, mx = max_mult - 1;
  if (v < 0) v = 0; if (v > mx) v = mx;
  return v;
}
static void awareness_delta(swim_sim* s, node_t* nd, int delta) {
  nd->awareness = (uint8_t)awareness_next((int)s->cfg.awareness_max_mult, (int)nd->awareness, delta);
}

/* ------------------------------------------------------------------------------------------ */
/* state.go aliveNode / suspectNode / deadNode / refute, applied at observer o                 */
/* ------------------------------------------------------------------------------------------ */

static void set_view(swim_sim* s, uint32_t r, uint32_t x, view_t* v, uint32_t inc, uint32_t st, int touch_since) {
  v->key = KEY(inc, st); v->reaped = 0;
  if (touch_since) v->since = now_ms(s);
  uint32_t sl = s->node_slot[(size_t)r * s->N + x];
  if (sl != SWIM_NONE) { slot_t* t = &s->slots[(size_t)r * s->cfg.subject_cap + sl]; if (inc > t->max_inc) t->max_inc = inc; t->dirty = 1; }
}
static void touch_slot(swim_sim* s, uint32_t r, uint32_t x) {
  uint32_t sl = s->node_slot[(size_t)r * s->N + x];
  if (sl != SWIM_NONE) s->slots[(size_t)r * s->cfg.subject_cap + sl].dirty = 1;
}
/* a suspicion timer was (re)armed at observer nd: keep its earliest-deadline bound */
static void arm_deadline(swim_sim* s, node_t* nd, const view_t* v) {
  uint32_t dl = v->since + suspicion_timeout_n(s, v->n0, v->nconf);
  if (dl < nd->vdl) nd->vdl = dl;
}

/* refute: nextIncarnation / skipIncarnation past the accuser, awareness +1, broadcast alive */
static void refute(swim_sim* s, uint32_t r, uint32_t o, node_t* nd, uint32_t accused_inc) {
  uint32_t inc = nd->self_inc + 1;
  if (accused_inc >= inc) inc = accused_inc + 1;
  nd->self_inc = inc;
  awareness_delta(s, nd, +1);
  view_t* v = view_make(s, r, o, o);                      /* a node's view of itself always fits */
  if (v) { v->nconf = 0; set_view(s, r, o, v, inc, SWIM_STATE_ALIVE, 0); }
  broadcast(s, nd, o, SWIM_MSG_ALIVE, inc, 0);
  s->st.refutes++;
}

/* `upd` (carried in the alive record's from field) = the alive came from UpdateNode, i.e. its
 * Meta differs from the previous incarnation's; a refutation re-sends the same Meta */
static void alive_node(swim_sim* s, uint32_t r, uint32_t o, node_t* nd, uint32_t x, uint32_t inc, uint32_t upd) {
  uint32_t key = view_key(s, r, o, x, NULL);
  int local = (x == o);
  if (local && nd->leaving) return;                       /* "if m.hasLeft() && a.Node == self" */
  if (!local && inc <= KINC(key)) return;
  if (local && inc < KINC(key)) return;
  if (local) {
    if (inc == KINC(key)) return;                         /* same incarnation, same meta */
    refute(s, r, o, nd, inc);
    return;
  }
  view_t* v = view_make(s, r, o, x); if (!v) return;
  v->nconf = 0; v->leaving = 0;                           /* delete(m.nodeTimers, a.Node); a newer life of the member is not leaving */
  uint32_t old = KST(key);
  broadcast(s, nd, x, SWIM_MSG_ALIVE, inc, upd);
  set_view(s, r, x, v, inc, SWIM_STATE_ALIVE, old != SWIM_STATE_ALIVE);
  s->st.msgs_applied[SWIM_MSG_ALIVE]++;
  if (watching(s, r, o)) {
    if (old == SWIM_STATE_DEAD || old == SWIM_STATE_LEFT) record_event(s, r, SWIM_EVENT_MEMBER_JOIN, x, 0, inc);
    else if (upd) record_event(s, r, SWIM_EVENT_MEMBER_UPDATE, x, 0, inc);   /* NotifyUpdate: meta changed */
  }
}

static void suspect_node(swim_sim* s, uint32_t r, uint32_t o, node_t* nd, uint32_t x, uint32_t inc, uint32_t from) {
  view_t* v = view_ptr(s, r, o, x);
  uint32_t key = v ? v->key : implicit_key(s, r, o, x);
  if (inc < KINC(key)) return;
  if (KST(key) == SWIM_STATE_SUSPECT) {                   /* a timer exists: suspicion.Confirm(from) (the base row is never Suspect) */
    if (v->nconf >= suspicion_k_n(s, v->n0)) return;
    for (uint32_t i = 0; i <= v->nconf && i < CONF_MAX; i++) if (v->conf[i] == from) return;
    v->nconf++;
    if (v->nconf < CONF_MAX) v->conf[v->nconf] = from;
    arm_deadline(s, nd, v);
    touch_slot(s, r, x); s->st.confirmations++;
    broadcast(s, nd, x, SWIM_MSG_SUSPECT, inc, from);
    return;
  }
  if (KST(key) != SWIM_STATE_ALIVE) return;
  if (x == o) { refute(s, r, o, nd, inc); return; }
  if (!v && !(v = view_make(s, r, o, x))) return;
  broadcast(s, nd, x, SWIM_MSG_SUSPECT, inc, from);
  set_view(s, r, x, v, inc, SWIM_STATE_SUSPECT, 1);
  v->nconf = 0; v->conf[0] = from; v->n0 = est_n(s, r, nd);   /* newSuspicion(from, k, min, max): k, min, max from estNumNodes() now */
  arm_deadline(s, nd, v);
  s->st.msgs_applied[SWIM_MSG_SUSPECT]++;
}

static void dead_node(swim_sim* s, uint32_t r, uint32_t o, node_t* nd, uint32_t x, uint32_t inc, uint32_t from) {
  view_t* v = view_ptr(s, r, o, x);
  uint32_t key = v ? v->key : implicit_key(s, r, o, x);
  if (inc < KINC(key)) return;
  uint32_t old = KST(key);
  if (old == SWIM_STATE_DEAD || old == SWIM_STATE_LEFT) return;
  if (x == o && !nd->leaving) { refute(s, r, o, nd, inc); return; }
  if (!v && !(v = view_make(s, r, o, x))) return;
  v->nconf = 0;
  broadcast(s, nd, x, SWIM_MSG_DEAD, inc, from);
  /* Left for a graceful leave (Node == From) and for a member a leave intent had marked Leaving here (serf handleNodeLeave) */
  uint32_t st = (from == x || v->leaving) ? SWIM_STATE_LEFT : SWIM_STATE_DEAD;
  v->leaving = 0;
  set_view(s, r, x, v, inc, st, 1);
  s->st.msgs_applied[SWIM_MSG_DEAD]++;
  if (x != o && watching(s, r, o))
    record_event(s, r, st == SWIM_STATE_LEFT ? SWIM_EVENT_MEMBER_LEAVE : SWIM_EVENT_MEMBER_FAILED, x, 0, inc);
}

/* serf.go handleNodeLeaveIntent for a force-leave (RemoveFailedNode): a member this observer holds Failed becomes Left
 * (EventMemberLeave); with prune it is erased at once (EventMemberReap), also when it was Left already.  A member that is
 * Alive or Suspect here is marked Leaving: when memberlist declares it dead it becomes Left, not Failed.  The Lamport
 * ordering of the intent against the member's status time is not modelled. */
static void user_event_from(swim_sim* s, uint32_t r, uint32_t o, node_t* nd, uint32_t id, uint32_t ltime, int origin);
/* serf.go handleNodeLeaveIntent.  Returns whether the intent is rebroadcast (serf's return value).  Order: the member's statusLTime
 * (a stale intent — one stamped no later than the last join / leave intent applied to the member here — is ignored), then the refutation
 * (an intent about this agent itself while it is not leaving: broadcastJoin(clock.Time())), then the transition by status.
 * Not modelled: serf's recentIntents buffer (an intent about a member this node has never heard of is passed on, not remembered). */
static int leave_intent(swim_sim* s, uint32_t r, uint32_t o, node_t* nd, uint32_t x, int prune, uint32_t ltime) {
  if (x >= s->N) return 0;
  if (x == o) {
    if (ltime <= nd->self_slt) return 0;
    if (!nd->serf_leaving) { user_event_from(s, r, o, nd, SWIM_INTENT_JOIN | o, nd->ev_clock, 1); return 0; }   /* refute: go s.broadcastJoin(s.clock.Time()) */
    nd->self_slt = ltime;                                  /* its own Leave(): StatusLeaving until memberlist's leave goes out */
    return 1;
  }
  view_t* v = view_ptr(s, r, o, x);
  uint32_t key = v ? v->key : implicit_key(s, r, o, x), st = KST(key);
  if (KINC(key) == 0) return 1;                            /* never heard of it (serf: upsertIntent into recentIntents, rebroadcast) */
  if (ltime <= (v ? v->slt : 0u)) return 0;                /* "If the message is old, then it is irrelevant and we can skip it" */
  if (st < SWIM_STATE_DEAD) {                              /* alive (or suspected) here: StatusLeaving — its death will read as a leave */
    if (!v && !(v = view_make(s, r, o, x))) return 1;
    v->slt = ltime;
    if (!v->leaving) { v->leaving = 1; touch_slot(s, r, x); }
    return 1;
  }
  /* erased already (serf no longer has the member; a Failed / Left member of the base row was erased before it got there) */
  if (v ? v->reaped : s->d.reap_period_ticks != 0) return 1;
  if (st == SWIM_STATE_LEFT && !prune) return 1;           /* StatusLeaving, StatusLeft: nothing but the prune */
  if (!v && !(v = view_make(s, r, o, x))) return 1;
  if (st == SWIM_STATE_DEAD) {
    set_view(s, r, x, v, KINC(key), SWIM_STATE_LEFT, 1);
    v->slt = ltime;
    s->st.intents_applied++;
    if (watching(s, r, o)) record_event(s, r, SWIM_EVENT_MEMBER_LEAVE, x, 0, KINC(key));
  }
  if (prune) {
    v->reaped = 1; s->st.reaped++; touch_slot(s, r, x);
    if (watching(s, r, o)) record_event(s, r, SWIM_EVENT_MEMBER_REAP, x, 0, KINC(key));
  }
  return 1;
}
/* serf.go handleNodeJoinIntent: a newer join intent moves the member's statusLTime and takes a Leaving mark back ("the leaving message
 * must have been for an older time") */
static int join_intent(swim_sim* s, uint32_t r, uint32_t o, node_t* nd, uint32_t x, uint32_t ltime) {
  if (x >= s->N) return 0;
  if (x == o) { if (ltime <= nd->self_slt) return 0; nd->self_slt = ltime; return 1; }
  view_t* v = view_ptr(s, r, o, x);
  uint32_t key = v ? v->key : implicit_key(s, r, o, x);
  if (KINC(key) == 0 || (v ? v->reaped : (KST(key) >= SWIM_STATE_DEAD && s->d.reap_period_ticks != 0))) return 1;     /* not a member here: passed on */
  if (ltime <= (v ? v->slt : 0u)) return 0;
  if (!v && !(v = view_make(s, r, o, x))) return 1;
  v->slt = ltime;
  if (v->leaving && KST(key) < SWIM_STATE_DEAD) { v->leaving = 0; touch_slot(s, r, x); }
  return 1;
}

/* serf.go handleUserEvent + lamport.go Witness (Consul fires via server_ce.go:125-131 and
 * consumes at server_serf.go:283, client_serf.go:98) */
static void user_event_from(swim_sim* s, uint32_t r, uint32_t o, node_t* nd, uint32_t id, uint32_t ltime, int origin) {
  if (!nd->ring) return;
  if (ltime >= nd->ev_clock) nd->ev_clock = ltime + 1;                 /* Witness */
  uint32_t cur = nd->ev_clock, bl = s->cfg.event_buffer;
  if (cur > bl && ltime < cur - bl) { s->st.user_events_stale++; return; }
  uint32_t* sl = &nd->ring[(size_t)(ltime % bl) * s->ev_words];        /* {ltime, n, ids...} */
  if (sl[1] && sl[0] == ltime) {
    for (uint32_t i = 0; i < sl[1]; i++) if (sl[2 + i] == id) { s->st.user_events_deduped++; return; }
  } else { sl[0] = ltime; sl[1] = 0; }
  if (sl[1] == s->ev_words - 2) { s->st.event_drops++; return; }
  sl[2 + sl[1]++] = id;
  if (id & SWIM_INTENT_LEAVE) {                            /* bit 31: one of serf's intents; rebroadcast as the handler says (the origin always sends) */
    const int again = (id & SWIM_INTENT_JOIN) == SWIM_INTENT_JOIN ? join_intent(s, r, o, nd, id & 0x1FFFFFFFu, ltime)
                                                                  : leave_intent(s, r, o, nd, id & 0x1FFFFFFFu, (id & SWIM_INTENT_PRUNE) != 0, ltime);
    if (!again && !origin) return;
  } else {
    s->st.user_events_delivered++;
    if (watching(s, r, o)) record_event(s, r, SWIM_EVENT_USER, id, ltime, 0);
  }
  queue_push(s, evq_of(s, nd), 0, id, SWIM_MSG_USER, ltime, 0, &s->st.event_drops);
}
static void user_event(swim_sim* s, uint32_t r, uint32_t o, node_t* nd, uint32_t id, uint32_t ltime) { user_event_from(s, r, o, nd, id, ltime, 0); }

/* ------------------------------------------------------------------------------------------ */
/* tick phases                                                                                 */
/* ------------------------------------------------------------------------------------------ */

static swim_edge mk_edge(const swim_sim* s, uint32_t r, uint32_t dst, uint32_t subject, uint32_t inc, uint32_t type, uint32_t from) {
  swim_edge e = { r * s->N + dst, subject, inc, (type << 30) | (from & 0x3FFFFFFFu) };
  return e;
}
/* A rumour for a node of ANOTHER shard: the sender cannot see the receiver's view, so the no-op question (noop_at_receiver) is asked by
 * the receiving shard when the record arrives (swim_inbound) — against the same pre-tick view the sender of an unsharded run reads.
 * The record says so in bit 29 of its meta word (node ids < 2^28 in sharded runs); it counts as crossing the wire here (edges_remote)
 * and as an edge or as filtered where it arrives, so that the shards' counters add up to the unsharded run's.  Without this a state
 * exchange that crosses a shard boundary arrives with EVERY explicit view of the sender and the receiver's inbox must hold them all. */
#define EDGE_JUDGE 0x20000000u
static int judged_remotely(swim_sim* s, uint32_t dst, uint32_t subject, uint32_t type) {
  return (s->cfg.flags & SWIM_F_FILTER_NOOP) && !is_local(s, dst) && type != SWIM_MSG_USER && subject != dst;
}
static void emit_from(swim_sim* s, uint32_t src, uint32_t r, uint32_t dst, uint32_t subject, uint32_t inc, uint32_t type, uint32_t from) {
  if (s->attached[(size_t)r * s->N + dst]) {            /* memberlist.Transport: hand the packet to the real node */
    if (s->captured.n == s->cap_src_cap) { s->cap_src_cap = s->cap_src_cap ? s->cap_src_cap * 2 : 1024; s->cap_src = (uint32_t*)realloc(s->cap_src, (size_t)s->cap_src_cap * 4); }
    s->cap_src[s->captured.n] = src;
    ev_push(&s->captured, mk_edge(s, r, dst, subject, inc, type, from & ~EDGE_JUDGE));
    return;
  }
  uint32_t sh = shard_of(s, dst);
  ev_push(&s->out[sh], mk_edge(s, r, dst, subject, inc, type, from));
  if (!(from & EDGE_JUDGE)) s->st.edges++;
  if (sh != s->cfg.shard_rank) s->st.edges_remote++;
}
static void emit(swim_sim* s, uint32_t r, uint32_t dst, uint32_t subject, uint32_t inc, uint32_t type, uint32_t from) {
  emit_from(s, SWIM_NONE, r, dst, subject, inc, type, from);
}
/* control record of the fold census (dst = SWIM_NONE): what this shard's acting observers hold about subject g =
 * replica*N + node — `verdict` = their common key, or FOLD_POISON when they disagree or a view is not settled yet;
 * `cnt` = how many of them hold an explicit view.  Goes to every shard, this one included. */
static void emit_fold_record(swim_sim* s, uint32_t g, uint32_t verdict, uint32_t cnt) {
  swim_edge e = { SWIM_NONE, g, verdict, cnt };
  for (uint32_t sh = 0; sh < s->cfg.n_shards; sh++) ev_push(&s->out[sh], e);
}

/* sendMsg (net.go): a ping / indirect ping / ack / nack from `sender` also carries sender's getBroadcasts().
 * The order reaches `sender` like a packet of this tick; `receiver` = SWIM_NONE when the carrier is lost on
 * the way (the broadcasts still count as transmitted).  Orders are not rumours: no edge statistics. */
static void piggy_order(swim_sim* s, uint32_t r, uint32_t sender, uint32_t receiver, uint32_t kind, uint32_t prober) {
  if (!(s->cfg.flags & SWIM_F_PIGGYBACK)) return;
  ev_push(&s->out[shard_of(s, sender)], mk_edge(s, r, sender, SWIM_SUBJECT_PIGGY, receiver, kind, prober));
}

static int lost(const swim_sim* s, uint32_t r, uint32_t node, uint32_t leg) {
  if (!s->loss_q32) return 0;
  uint32_t key[2], c[4] = { s->tick, node, leg, 0 }, w[4];
  stream_key(seed_of(s, r), STREAM_LOSS, key); philox4x32(c, key, w);
  return w[0] < s->loss_q32;
}
/* can a packet sent by a (alive) reach b right now */
static int reach(const swim_sim* s, uint32_t r, uint32_t a, uint32_t b, uint32_t rng_node, uint32_t leg) {
  size_t base = (size_t)r * s->N;
  if (!s->gt_alive[base + b]) return 0;
  if (s->part[base + a] != s->part[base + b]) return 0;
  return !lost(s, r, rng_node, leg);
}

/* suspicion timers: the time.AfterFunc of suspectNode firing -> deadNode(dead{inc, node, self}) */
static void phase_expire(swim_sim* s) {
  uint32_t now = now_ms(s);
  for (uint32_t r = 0; r < s->R; r++)
    for (uint32_t k = 0; k < s->nloc; k++) {
      uint32_t o = s->i0 + k; node_t* nd = &s->nodes[(size_t)r * s->nloc + k];
      if (now < nd->vdl || !acts(s, r, o)) continue;       /* vdl is a lower bound of the node's deadlines */
      uint32_t next = SWIM_NONE;
      for (uint32_t i = 0; i < nd->vt.slots; i++) {
        view_t* v = &nd->vt.e[i];
        if (v->subj == V_EMPTY || KST(v->key) != SWIM_STATE_SUSPECT) continue;
        uint32_t dl = v->since + suspicion_timeout_n(s, v->n0, v->nconf);
        if (now >= dl) {
          /* a timer is not a packet: straight into the node's own inbox (not part of swim_debug_edges) */
          ev_push(&s->in, mk_edge(s, r, o, v->subj, KINC(v->key), SWIM_MSG_DEAD, o));
          s->st.edges++; s->st.suspicion_timeouts++;
        }
        if (dl < next) next = dl;   /* a fired timer keeps the bound low until the verdict is merged (it may be dropped) */
      }
      nd->vdl = next;
    }
}

/* util.go kRandomNodes: up to 3n draws of randomOffset(n), skipping excluded and duplicates */
typedef int (*excl_fn)(swim_sim*, uint32_t r, uint32_t o, uint32_t x, void* ctx);
static uint32_t k_random_nodes(swim_sim* s, uint32_t r, uint32_t o, uint32_t stream, uint32_t k, excl_fn ex, void* ctx, uint32_t* out) {
  draws_t d; draws_init(&d, seed_of(s, r), stream, s->tick, o);
  uint32_t found = 0; uint64_t tries = 3ull * s->N;
  for (uint64_t i = 0; i < tries && found < k; i++) {
    uint32_t x = draws_get(&d, (uint32_t)i) % s->N;
    if (ex(s, r, o, x, ctx)) continue;
    int dup = 0; for (uint32_t j = 0; j < found; j++) if (out[j] == x) dup = 1;
    if (dup) continue;
    out[found++] = x;
  }
  return found;
}

/* gossip(): exclude self, Left, and Dead for longer than GossipToTheDeadTime */
static int excl_gossip(swim_sim* s, uint32_t r, uint32_t o, uint32_t x, void* ctx) {
  (void)ctx; if (x == o) return 1;
  uint32_t since, key = view_key(s, r, o, x, &since);
  if (KINC(key) == 0) return 1;                            /* never heard of it: not in this node's member list */
  switch (KST(key)) {
    case SWIM_STATE_ALIVE: case SWIM_STATE_SUSPECT: return 0;
    case SWIM_STATE_DEAD: return now_ms(s) - since > s->cfg.gossip_to_dead_ms;
    default: return 1;
  }
}
/* probeNode indirect helpers: exclude self, the target, and anything not Alive */
static int excl_indirect(swim_sim* s, uint32_t r, uint32_t o, uint32_t x, void* ctx) {
  uint32_t target = *(uint32_t*)ctx; if (x == o || x == target) return 1;
  return KST(view_key(s, r, o, x, NULL)) != SWIM_STATE_ALIVE;
}


/* ------------------------------------------------------------------------------------------ */
/* serf/coordinate — Vivaldi network coordinates (SWIM_F_COORDINATES).  UPSTREAM-RECALL of     */
/* serf v0.10.4 coordinate/{config,coordinate,client}.go and serf/ping_delegate.go; the in-tree */
/* pins are librtt.ComputeDistance and its test table (internal/gossip/librtt/rtt.go:16-22,     */
/* rtt_test.go:16-75).  Compiled with -ffp-contract=off: every operation rounds once, like Go.  */
/* ------------------------------------------------------------------------------------------ */
#define VIVALDI_ERROR_MAX 1.5      /* coordinate.DefaultConfig(): VivaldiErrorMax, VivaldiCE, VivaldiCC, HeightMin, GravityRho */
#define VIVALDI_CE 0.25
#define VIVALDI_CC 0.25
#define COORD_HEIGHT_MIN 10.0e-6
#define COORD_GRAVITY_RHO 150.0
#define COORD_ZERO_THRESHOLD 1.0e-6

static void coord_new(swim_coordinate* c) {              /* NewCoordinate */
  for (int i = 0; i < SWIM_COORD_DIMS; i++) c->vec[i] = 0.0;
  c->error = VIVALDI_ERROR_MAX; c->adjustment = 0.0; c->height = COORD_HEIGHT_MIN;
}
static int coord_valid(const swim_coordinate* c) {       /* IsValid: every component finite */
  for (int i = 0; i < SWIM_COORD_DIMS; i++) if (!isfinite(c->vec[i])) return 0;
  return isfinite(c->error) && isfinite(c->adjustment) && isfinite(c->height);
}
static double vec_magnitude(const double* v) { double sum = 0.0; for (int i = 0; i < SWIM_COORD_DIMS; i++) sum += v[i] * v[i]; return sqrt(sum); }
static double coord_raw_distance(const swim_coordinate* a, const swim_coordinate* b) {   /* rawDistanceTo */
  double d[SWIM_COORD_DIMS]; for (int i = 0; i < SWIM_COORD_DIMS; i++) d[i] = a->vec[i] - b->vec[i];
  return vec_magnitude(d) + a->height + b->height;
}
/* DistanceTo(...).Seconds(): through time.Duration — int64 nanoseconds, truncated — and back */
static double coord_distance_seconds(const swim_coordinate* a, const swim_coordinate* b) {
  double dist = coord_raw_distance(a, b), adjusted = dist + a->adjustment + b->adjustment;
  if (adjusted > 0.0) dist = adjusted;
  int64_t ns = (int64_t)(dist * 1.0e9);
  return (double)(ns / 1000000000) + (double)(ns % 1000000000) / 1e9;
}
double swim_coordinate_distance(const swim_coordinate* a, const swim_coordinate* b) {
  if (!a || !b) return INFINITY;
  return coord_distance_seconds(a, b);
}
/* unitVectorAt: the direction from b to a; coincident points get a random direction (rand.Float64() - 0.5 per dimension) */
static double coord_unit_vector(swim_sim* s, uint32_t r, uint32_t o, uint32_t salt, const double* a, const double* b, double* unit) {
  for (int i = 0; i < SWIM_COORD_DIMS; i++) unit[i] = a[i] - b[i];
  double mag = vec_magnitude(unit);
  if (mag > COORD_ZERO_THRESHOLD) { double inv = 1.0 / mag; for (int i = 0; i < SWIM_COORD_DIMS; i++) unit[i] = unit[i] * inv; return mag; }
  draws_t d; draws_init(&d, seed_of(s, r), STREAM_COORD, s->tick, o);
  for (int i = 0; i < SWIM_COORD_DIMS; i++) unit[i] = (double)draws_get(&d, salt * SWIM_COORD_DIMS + (uint32_t)i) / 4294967296.0 - 0.5;
  mag = vec_magnitude(unit);
  if (mag > COORD_ZERO_THRESHOLD) { double inv = 1.0 / mag; for (int i = 0; i < SWIM_COORD_DIMS; i++) unit[i] = unit[i] * inv; return 0.0; }
  for (int i = 0; i < SWIM_COORD_DIMS; i++) unit[i] = 0.0;
  unit[0] = 1.0; return 0.0;
}
static void coord_apply_force(swim_sim* s, uint32_t r, uint32_t o, uint32_t salt, swim_coordinate* c, double force, const swim_coordinate* other) {
  double unit[SWIM_COORD_DIMS], mag = coord_unit_vector(s, r, o, salt, c->vec, other->vec, unit);
  for (int i = 0; i < SWIM_COORD_DIMS; i++) c->vec[i] = c->vec[i] + unit[i] * force;
  if (mag > COORD_ZERO_THRESHOLD) {
    c->height = (c->height + other->height) * force / mag + c->height;
    if (!(c->height >= COORD_HEIGHT_MIN)) c->height = isnan(c->height) ? c->height : COORD_HEIGHT_MIN;   /* math.Max(height, HeightMin): NaN stays NaN */
  }
}
/* the latency model: hidden position and access-link height of a node, microseconds */
static void rtt_truth_of(swim_sim* s, uint32_t r, uint32_t i, uint32_t pos[3], uint32_t* h) {
  uint32_t key[2], c[4] = { i, 0, 0, 0x54525554u }, w[4];
  stream_key(seed_of(s, r), STREAM_TRUTH, key); philox4x32(c, key, w);
  for (int k = 0; k < 3; k++) pos[k] = (uint32_t)(((uint64_t)w[k] * s->cfg.rtt_scale_us) >> 32);
  *h = (uint32_t)(((uint64_t)w[3] * s->cfg.rtt_height_us) >> 32);
}
static uint32_t rtt_between(swim_sim* s, uint32_t r, uint32_t a, uint32_t b) {
  uint32_t pa[3], pb[3], ha, hb; rtt_truth_of(s, r, a, pa, &ha); rtt_truth_of(s, r, b, pb, &hb);
  double sum = 0.0;
  for (int k = 0; k < 3; k++) { double d = (double)pa[k] - (double)pb[k]; sum += d * d; }
  return (uint32_t)sqrt(sum) + ha + hb;
}
int swim_rtt_truth(swim_sim* s, uint32_t replica, uint32_t a, uint32_t b, uint32_t* rtt_us) {
  if (!s || !rtt_us || replica >= s->R || a >= s->N || b >= s->N) return SWIM_EINVAL;
  *rtt_us = rtt_between(s, replica, a, b);
  return SWIM_OK;
}
int swim_coordinate_get(swim_sim* s, uint32_t replica, uint32_t node, swim_coordinate* out) {
  if (!s || !out || replica >= s->R || node >= s->N) return SWIM_EINVAL;
  if (!s->cs) return SWIM_ESTATE;
  *out = s->cs[(size_t)replica * s->nloc + (node - s->i0)].c;
  return SWIM_OK;
}
/* Client.latencyFilter: the median of the last LatencyFilterSize round-trip times seen from this peer */
static uint32_t coord_latency_filter(swim_sim* s, coord_state* st, uint32_t peer, uint32_t rtt_us) {
  lf_ent* e = NULL;
  for (int j = 0; j < COORD_PEERS; j++) if (st->lf[j].n && st->lf[j].peer == peer) { e = &st->lf[j]; break; }
  if (!e) {                                              /* a free entry, else the one unused for longest (ties: lowest index) */
    e = &st->lf[0];
    for (int j = 0; j < COORD_PEERS; j++) { if (!st->lf[j].n) { e = &st->lf[j]; break; } if (st->lf[j].last < e->last) e = &st->lf[j]; }
    e->peer = peer; e->n = 0;
  }
  if (e->n == COORD_FILTER) { for (int j = 1; j < COORD_FILTER; j++) e->s[j - 1] = e->s[j]; e->n--; }
  e->s[e->n++] = rtt_us; e->last = s->tick;
  uint32_t sorted[COORD_FILTER];
  for (uint32_t j = 0; j < e->n; j++) sorted[j] = e->s[j];
  for (uint32_t j = 1; j < e->n; j++) for (uint32_t k = j; k > 0 && sorted[k - 1] > sorted[k]; k--) { uint32_t t = sorted[k]; sorted[k] = sorted[k - 1]; sorted[k - 1] = t; }
  return sorted[e->n / 2];
}
/* serf pingDelegate.NotifyPingComplete -> Client.Update(other, coord, rtt): node o got a direct ack from x.  The ack's payload
 * is x's coordinate as of the START of this tick (determinisation: x may be updating its own in the same tick). */
static void coord_update(swim_sim* s, uint32_t r, uint32_t o, uint32_t x) {
  coord_state* st = &s->cs[(size_t)r * s->nloc + (o - s->i0)];
  const swim_coordinate* other = &s->cs[(size_t)r * s->nloc + (x - s->i0)].c;
  uint32_t rtt_us = rtt_between(s, r, o, x);
  if (s->cfg.rtt_jitter_us) {
    uint32_t key[2], c[4] = { s->tick, o, 0, 0x52545431u }, w[4];
    stream_key(seed_of(s, r), STREAM_RTT, key); philox4x32(c, key, w);
    rtt_us += (uint32_t)(((uint64_t)w[0] * s->cfg.rtt_jitter_us) >> 32);
  }
  s->st.coord_updates++;
  uint32_t med_us = coord_latency_filter(s, st, x, rtt_us);
  double rtt = (double)((uint64_t)med_us * 1000u) / 1e9;          /* time.Duration(ns).Seconds() below one second */
  swim_coordinate c = st->c;
  /* updateVivaldi */
  {
    double dist = coord_distance_seconds(&c, other);
    double rs = rtt < COORD_ZERO_THRESHOLD ? COORD_ZERO_THRESHOLD : rtt;
    double wrongness = fabs(dist - rs) / rs;
    double total = c.error + other->error; if (total < COORD_ZERO_THRESHOLD) total = COORD_ZERO_THRESHOLD;
    double weight = c.error / total;
    c.error = VIVALDI_CE * weight * wrongness + c.error * (1.0 - VIVALDI_CE * weight);
    if (c.error > VIVALDI_ERROR_MAX) c.error = VIVALDI_ERROR_MAX;
    double delta = VIVALDI_CC * weight, force = delta * (rs - dist);
    coord_apply_force(s, r, o, 0, &c, force, other);
  }
  /* updateAdjustment */
  {
    double dist = coord_raw_distance(&c, other);
    st->adj[st->adj_idx] = rtt - dist; st->adj_idx = (st->adj_idx + 1) % COORD_WINDOW;
    double sum = 0.0; for (int i = 0; i < COORD_WINDOW; i++) sum += st->adj[i];
    c.adjustment = sum / (2.0 * (double)COORD_WINDOW);
  }
  /* updateGravity: a pull towards the origin */
  {
    swim_coordinate origin; coord_new(&origin);
    double dist = coord_distance_seconds(&origin, &c), q = dist / COORD_GRAVITY_RHO, force = -1.0 * (q * q);
    coord_apply_force(s, r, o, 1, &c, force, &origin);
  }
  if (!coord_valid(&c)) { s->st.coord_resets++; coord_new(&c); }
  s->c_new[s->c_n] = c; s->c_list[s->c_n++] = (uint32_t)((size_t)r * s->nloc + (o - s->i0));
}
static void coord_commit(swim_sim* s) {
  for (uint32_t j = 0; j < s->c_n; j++) s->cs[s->c_list[j]].c = s->c_new[j];
  s->c_n = 0;
}

/* probeNode's failure epilogue: awareness delta then suspectNode(suspect{inc, node, self}) */
static void probe_conclude(swim_sim* s, uint32_t r, uint32_t o, node_t* nd) {
  uint32_t x = nd->pr_target;
  awareness_delta(s, nd, (int)nd->pr_nack_miss);
  s->st.probe_failures++; s->st.nacks_missed += nd->pr_nack_miss;
  emit(s, r, o, x, nd->pr_inc, SWIM_MSG_SUSPECT, o);     /* node.Incarnation of the copy probe() took */
  nd->pr_target = SWIM_NONE; nd->pr_stage = 0;
}

/* probe(): walk the shuffled list to the next node that is not self / dead / left */
static void probe_start(swim_sim* s, uint32_t r, uint32_t o, node_t* nd) {
  uint32_t num_check = 0, x = SWIM_NONE, key = 0;
  while (num_check < s->N) {
    if (nd->pr_cursor >= s->N) { nd->pr_epoch = (nd->pr_epoch + 1) & 0xFFFFu; nd->pr_cursor = 0; num_check++; continue; }   /* resetNodes (epochs are 16 bits) */
    uint32_t c = probe_perm(seed_of(s, r), s->N, o, nd->pr_epoch, nd->pr_cursor++);
    key = view_key(s, r, o, c, NULL);
    if (c == o || KST(key) == SWIM_STATE_DEAD || KST(key) == SWIM_STATE_LEFT) { num_check++; continue; }
    x = c; break;
  }
  if (x == SWIM_NONE) return;
  s->st.probes++;
  /* probeNode: direct ping; a non-alive target also gets suspect{} so it can refute early */
  int fwd = reach(s, r, o, x, o, 16);
  if (fwd && KST(key) != SWIM_STATE_ALIVE && (s->cfg.flags & SWIM_F_BUDDY_SUSPECT))
    emit(s, r, x, x, KINC(key), SWIM_MSG_SUSPECT, o);
  int ack = fwd && !lost(s, r, o, 17);
  if (KST(key) == SWIM_STATE_ALIVE) piggy_order(s, r, o, fwd ? x : SWIM_NONE, SWIM_CTL_PING, o);   /* else: ping+suspect compound, sent raw */
  if (fwd) piggy_order(s, r, x, ack ? o : SWIM_NONE, SWIM_CTL_ACK, o);
  if (ack) { awareness_delta(s, nd, -1); s->st.probe_acks++; if (s->cs) coord_update(s, r, o, x); return; }
  nd->pr_target = x; nd->pr_inc = KINC(key); nd->pr_t0 = s->tick; nd->pr_stage = 1; nd->pr_nack_miss = 1;
  nd->pr_deadline = s->tick + s->d.probe_period * ((uint32_t)nd->awareness + 1);   /* awareness.ScaleTimeout */
}

/* ProbeTimeout after the ping: indirectPingReq to IndirectChecks random alive peers; each either
 * relays the target's ack, or (Lifeguard) answers nack one further ProbeTimeout later */
static void probe_indirect(swim_sim* s, uint32_t r, uint32_t o, node_t* nd) {
  uint32_t x = nd->pr_target, peers[8];
  uint32_t np = k_random_nodes(s, r, o, STREAM_INDIRECT, s->cfg.indirect_checks, excl_indirect, &x, peers);
  uint32_t expected = 0, nacks = 0; int acked = 0;
  int nack_in_time = 2 * s->d.probe_timeout_ticks < nd->pr_deadline - nd->pr_t0;
  for (uint32_t q = 0; q < np; q++) {
    uint32_t h = peers[q];
    if (s->cfg.flags & SWIM_F_NACK) expected++;
    int there = reach(s, r, o, h, o, 20 + 4 * q);
    piggy_order(s, r, o, there ? h : SWIM_NONE, SWIM_CTL_INDIRECT, o);
    if (!there) continue;
    int hx = reach(s, r, h, x, o, 21 + 4 * q), xh = hx && reach(s, r, x, h, o, 22 + 4 * q), ok = hx && xh;
    int back = reach(s, r, h, o, o, 23 + 4 * q);
    piggy_order(s, r, h, hx ? x : SWIM_NONE, SWIM_CTL_PING, o);
    if (hx) piggy_order(s, r, x, xh ? h : SWIM_NONE, SWIM_CTL_ACK, o);
    if (ok) piggy_order(s, r, h, back ? o : SWIM_NONE, SWIM_CTL_ACK, o);
    else if ((s->cfg.flags & SWIM_F_NACK) && nack_in_time) piggy_order(s, r, h, back ? o : SWIM_NONE, SWIM_CTL_NACK, o);
    if (ok && back) acked = 1;
    else if (!ok && back && nack_in_time) nacks++;
  }
  nd->pr_stage = 2;
  /* probeNode's TCP fallback ping next to the indirect probes: TCP rides out packet loss, so it reaches every running
   * node of the same partition ("Was able to connect to %s over TCP but UDP probes failed") */
  int tcp = 0;
  if (!acked && (s->cfg.flags & SWIM_F_TCP_FALLBACK)) {
    size_t base = (size_t)r * s->N;
    tcp = s->gt_alive[base + x] && s->part[base + o] == s->part[base + x] &&
          s->tcp_cls[base + o] == s->tcp_cls[base + x];                /* DisableTcpPingsForNode: other datacenter */
  }
  if (acked || tcp) {
    awareness_delta(s, nd, -1);
    if (acked) s->st.probe_indirect_acks++; else s->st.probe_tcp_acks++;
    nd->pr_target = SWIM_NONE; nd->pr_stage = 0; return;
  }
  nd->pr_nack_miss = expected > 0 ? (uint8_t)(expected - nacks) : 1;
}

static void phase_probe(swim_sim* s) {
  uint32_t P = s->d.probe_period, TQ = s->d.probe_timeout_ticks, t = s->tick;
  for (uint32_t r = 0; r < s->R; r++) {
    if (t >= TQ) {
      uint32_t ph = (t - TQ) % P;
      for (uint32_t k = 0; k < s->nloc; k++) {
        uint32_t o = s->i0 + k; node_t* nd = node_at(s, r, o);
        if (pphase_of(s, o) != ph || !acts(s, r, o)) continue;
        if (nd->pr_stage == 1 && nd->pr_t0 + TQ == t) probe_indirect(s, r, o, nd);
      }
    }
    uint32_t ph = t % P;
    for (uint32_t k = 0; k < s->nloc; k++) {
      uint32_t o = s->i0 + k; node_t* nd = node_at(s, r, o);
      if (pphase_of(s, o) != ph || !acts(s, r, o)) continue;
      if (nd->pr_stage != 0) { if (t < nd->pr_deadline) continue; probe_conclude(s, r, o, nd); }
      probe_start(s, r, o, nd);
    }
  }
}

/* SWIM_F_FILTER_NOOP: would aliveNode/suspectNode/deadNode at `dst` return without doing anything?
 * Only conditions that stay true whatever else reaches dst in the same tick (view incarnations never
 * decrease): an older incarnation; or the same incarnation in a state the message cannot move.  A
 * message with a HIGHER incarnation than the view is always delivered (an alive of that incarnation
 * arriving in the same tick could make it applicable), and so is anything about dst itself (refute). */
static int noop_at_receiver(swim_sim* s, uint32_t r, uint32_t dst, const qent* m) {
  if (m->type == SWIM_MSG_USER || m->subject == dst || !is_local(s, dst)) return 0;
  view_t* v = view_ptr(s, r, dst, m->subject);
  uint32_t key = v ? v->key : s->base_key[(size_t)r * s->N + m->subject];
  uint32_t vinc = KINC(key), st = KST(key);
  if (m->type == SWIM_MSG_ALIVE) return m->inc <= vinc;
  if (m->inc != vinc) return m->inc < vinc;
  if (st == SWIM_STATE_DEAD || st == SWIM_STATE_LEFT) return 1;
  if (m->type == SWIM_MSG_SUSPECT && st == SWIM_STATE_SUSPECT) {
    if (v->nconf >= suspicion_k_n(s, v->n0)) return 1;
    for (uint32_t i = 0; i <= v->nconf && i < CONF_MAX; i++) if (v->conf[i] == m->from) return 1;
  }
  return 0;
}

/* state.go pushPull / pushPullNode / mergeState, made message-based so it works across
 * shards: the initiator sends, for every subject, the rumour mergeState would derive from its own view
 * (Alive -> alive, Left -> dead{From: node}, Dead|Suspect -> suspect{From: receiver} — a remote Dead is
 * never trusted directly) plus a pull request; the peer answers the same way one tick later. */
static int excl_pushpull(swim_sim* s, uint32_t r, uint32_t o, uint32_t x, void* ctx) {
  (void)ctx; if (x == o) return 1;
  return KST(view_key(s, r, o, x, NULL)) != SWIM_STATE_ALIVE;
}
static void send_state(swim_sim* s, uint32_t r, uint32_t owner, uint32_t dst) {
  /* What the base row says is what the receiver holds too and merges to nothing — with one exception: the receiver's
   * view of ITSELF is its own (it may have been away while the base row moved on), so the owner's view of the
   * receiver travels even when it is the base row's (and not the trivial alive@1). */
  const vtab* t = &node_at(s, r, owner)->vt;
  int saw_dst = 0;
  /* ...and the owner's view of ITSELF travels when the base row says something else about it (a node that has just
   * joined: nobody has heard of it; a node that came back after it was folded as dead) */
  if (!vt_find(t, owner)) {
    uint32_t self = KEY(node_at(s, r, owner)->self_inc, SWIM_STATE_ALIVE);
    if (self != s->base_key[(size_t)r * s->N + owner]) emit(s, r, dst, owner, KINC(self), SWIM_MSG_ALIVE, 0);
  }
  for (uint32_t i = 0; i <= t->slots; i++) {
    uint32_t subj, key;
    if (i < t->slots) { const view_t* v = &t->e[i]; if (v->subj == V_EMPTY) continue; subj = v->subj; key = v->key; saw_dst |= subj == dst; }
    else { subj = dst; key = s->base_key[(size_t)r * s->N + dst]; if (saw_dst || key == BASE_KEY || KINC(key) == 0) break; }
    qent m = { subj, KINC(key), 0, 0, 0, 0 };
    switch (KST(key)) {
      case SWIM_STATE_ALIVE: m.type = SWIM_MSG_ALIVE; break;
      case SWIM_STATE_LEFT: m.type = SWIM_MSG_DEAD; m.from = subj; break;
      default: m.type = SWIM_MSG_SUSPECT; m.from = dst; break;
    }
    if ((s->cfg.flags & SWIM_F_FILTER_NOOP) && noop_at_receiver(s, r, dst, &m)) { s->st.msgs_filtered++; continue; }
    emit(s, r, dst, m.subject, m.inc, m.type, m.from | (judged_remotely(s, dst, m.subject, m.type) ? EDGE_JUDGE : 0u));
  }
}
static void phase_pushpull(swim_sim* s) {
  /* answers to the requests of the previous tick */
  edgevec* rq = &s->pp_reply[s->tick & 1];
  for (uint32_t i = 0; i < rq->n; i++) {
    uint32_t r = rq->v[i].incarnation, p = rq->v[i].dst, o = rq->v[i].subject;
    if (!acts(s, r, p)) continue;
    send_state(s, r, p, o);
    /* serf.Join: the state exchange hands over serf's own push-pull message too (MergeRemoteState: clock.Witness(LTime - 1)), THEN the joiner
     * calls broadcastJoin(s.clock.Time()) — i.e. its join intent is stamped with the clock of the member it joined through.  The reply to a
     * JOIN's pull request therefore carries that intent, ready-stamped, to the joiner, which witnesses it, applies it to its own entry and
     * broadcasts it (a message like any other: the same across shards). */
    if ((rq->v[i].meta & 1u) && node_at(s, r, p)->ring) emit(s, r, o, SWIM_INTENT_JOIN | o, node_at(s, r, p)->ev_clock, SWIM_MSG_USER, 0);
  }
  rq->n = 0;
  /* swim_inject_join: the join push-pull (pushPullNode(join=true)) of the nodes started since the last tick */
  for (uint32_t j = 0; j < s->n_join_pending; j++) {
    uint32_t r = s->join_list[2 * j] / s->N, o = s->join_list[2 * j] % s->N, p = s->join_list[2 * j + 1];
    size_t base = (size_t)r * s->N;
    if (!s->gt_alive[base + o] || s->attached[base + o] || !s->alone[base + o]) continue;   /* killed again meanwhile */
    int ok = p != o && s->gt_alive[base + p] && s->part[base + o] == s->part[base + p];
    s->join_list[2 * j + 1] = ok ? p : SWIM_NONE;          /* tick end: a node whose join went through stops being alone */
    if (!is_local(s, o)) continue;
    if (!ok) { s->st.join_failures++; continue; }
    s->st.joins++;
    send_state(s, r, o, p);
    emit(s, r, p, SWIM_SUBJECT_PULL, o, SWIM_MSG_ALIVE, 1);          /* from = 1: a JOIN's pull request (serf.Join: the reply brings the join intent) */
  }
  /* stagger: node i is due in tick (i mod period), but exchanges start only on probe-interval boundaries
   * (everything due within the next ProbeInterval goes now) — memberlist's own stagger is a random point
   * of the whole interval, so second-granularity loses nothing and keeps most ticks free of this path */
  uint32_t per = s->d.push_pull_period_ticks, grp = s->d.probe_period;
  if (!per || s->tick % grp) return;
  for (uint32_t r = 0; r < s->R; r++)
   for (uint32_t off = 0; off < grp && off < per; off++)
    for (uint64_t i64 = (s->tick + off) % per; i64 < s->N; i64 += per) {
      uint32_t o = (uint32_t)i64, p;
      if (!is_local(s, o) || !acts(s, r, o)) continue;
      if (!k_random_nodes(s, r, o, STREAM_PUSHPULL, 1, excl_pushpull, NULL, &p)) continue;
      size_t base = (size_t)r * s->N;
      if (!s->gt_alive[base + p] || s->part[base + o] != s->part[base + p]) continue;   /* TCP dial fails */
      s->st.push_pulls++;
      send_state(s, r, o, p);
      emit(s, r, p, SWIM_SUBJECT_PULL, o, SWIM_MSG_ALIVE, 0);
    }
}

/* serf.go reconnect() (every ReconnectInterval, per node): n failed members, prob = n / alive members; with probability prob
 * pick one failed member uniformly and memberlist.Join([its address]) — a state exchange (pushPullNode, join) that succeeds when
 * the member runs and is in reach.  Here: a node is due in tick (id mod period), grouped to probe-interval boundaries like
 * push-pull; "failed" = its explicit Dead views that serf has not erased (a death folded into the base row is settled for
 * everybody and not retried); the gate compares a Philox word with n / alive in integers; the uniform pick is the member with
 * the smallest keyed hash (no order of enumeration enters). */
static void phase_reconnect(swim_sim* s) {
  uint32_t per = s->d.reconnect_period_ticks, grp = s->d.probe_period;
  if (!per || s->tick % grp) return;
  for (uint32_t r = 0; r < s->R; r++)
   for (uint32_t off = 0; off < grp && off < per; off++)
    for (uint64_t i64 = (s->tick + off) % per; i64 < s->N; i64 += per) {
      uint32_t o = (uint32_t)i64;
      if (!is_local(s, o) || !acts(s, r, o)) continue;
      node_t* nd = node_at(s, r, o);
      uint32_t key[2], c[4] = { s->tick, o, 0, 0x5245434Eu }, w[4];
      stream_key(seed_of(s, r), STREAM_RECONNECT, key); philox4x32(c, key, w);
      uint32_t n_failed = 0, best = SWIM_NONE, best_h = 0;
      for (uint32_t i = 0; i < nd->vt.slots; i++) {
        const view_t* v = &nd->vt.e[i];
        if (v->subj == V_EMPTY || v->subj == o || KST(v->key) != SWIM_STATE_DEAD || v->reaped) continue;
        n_failed++;
        uint32_t h = fmix32(v->subj ^ w[1]);
        if (best == SWIM_NONE || h < best_h || (h == best_h && v->subj < best)) { best = v->subj; best_h = h; }
      }
      if (!n_failed) continue;
      uint32_t members = est_n(s, r, nd), alive = members > n_failed ? members - n_failed : 1;
      if ((uint64_t)w[0] * alive > ((uint64_t)n_failed << 32)) continue;          /* rand.Float32() > prob */
      s->st.reconnects++;
      size_t base = (size_t)r * s->N;
      if (!s->gt_alive[base + best] || s->part[base + o] != s->part[base + best]) continue;   /* the dial fails */
      s->st.reconnects_reached++;
      send_state(s, r, o, best);
      emit(s, r, best, SWIM_SUBJECT_PULL, o, SWIM_MSG_ALIVE, 0);
    }
}

/* gossip(): k random peers; per peer one getBroadcasts() = memberlist queue, then the serf
 * delegate's user events in the bytes that remain; stop at the first empty packet */
static void phase_gossip(swim_sim* s) {
  uint32_t G = s->d.gossip_period, ph = s->tick % G;
  for (uint32_t r = 0; r < s->R; r++)
    for (uint32_t k = 0; k < s->nloc; k++) {
      uint32_t o = s->i0 + k; node_t* nd = node_at(s, r, o);
      if (gphase_of(s, o) != ph || !acts(s, r, o)) continue;
      if (!nd->qlen && !nd->evqlen) { s->st.node_rounds_quiescent++; continue; }
      s->st.node_rounds_active++;
      uint32_t peers[8], np = k_random_nodes(s, r, o, STREAM_GOSSIP, s->cfg.gossip_nodes, excl_gossip, NULL, peers);
      for (uint32_t p = 0; p < np; p++) {
        qent* const msgs = pick_buf(); int32_t used = 0, used2 = 0;     /* room for both queues whole, however deep serf's is */
        const uint32_t rl = retransmit_limit_n(s, est_n(s, r, nd));
        uint32_t n = queue_get(s, mlq(s, r, nd), 2, (int32_t)s->d.packet_budget, msgs, &used, rl);
        int32_t avail = (int32_t)s->d.packet_budget - used;
        if (nd->ring && avail > 2 + 1) n += queue_get(s, evq_of(s, nd), 3, avail, msgs + n, &used2, rl);
        if (!n) break;
        s->st.packets_sent++;
        for (uint32_t m = 0; m < n; m++) s->st.msgs_sent[msgs[m].type]++;
        if (!reach(s, r, o, peers[p], o, p)) { s->st.packets_dropped++; continue; }
        for (uint32_t m = 0; m < n; m++) {
          if ((s->cfg.flags & SWIM_F_FILTER_NOOP) && noop_at_receiver(s, r, peers[p], &msgs[m])) { s->st.msgs_filtered++; continue; }
          emit_from(s, o, r, peers[p], msgs[m].subject, msgs[m].inc, msgs[m].type, msgs[m].from | (judged_remotely(s, peers[p], msgs[m].subject, msgs[m].type) ? EDGE_JUDGE : 0u));
        }
      }
    }
}

static int edge_cmp(const void* a, const void* b) {
  const swim_edge *x = (const swim_edge*)a, *y = (const swim_edge*)b;
  uint32_t px = x->subject != SWIM_SUBJECT_PIGGY, py = y->subject != SWIM_SUBJECT_PIGGY;
  if (px != py) return px < py ? -1 : 1;          /* piggy-back orders act on the queue as the begin phase left it */
  uint32_t tx = x->meta >> 30, ty = y->meta >> 30, ux = px && tx == SWIM_MSG_USER, uy = py && ty == SWIM_MSG_USER;
  if (ux != uy) return ux < uy ? -1 : 1;
  if (x->subject != y->subject) return x->subject < y->subject ? -1 : 1;
  if (tx != ty) return tx < ty ? -1 : 1;
  if (x->incarnation != y->incarnation) return x->incarnation < y->incarnation ? -1 : 1;
  if (x->meta != y->meta) return x->meta < y->meta ? -1 : 1;
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* fold (simulator-only compaction; SURVEY §7 hard part 1, Appendix D k_reap_fold)             */
/* Real memberlist keeps an N-entry map per node; here an observer stores only what differs    */
/* from the replica's base row.  Every fold_period ticks: a subject on which EVERY acting node */
/* of the population holds the same explicit view — same incarnation and state, not Suspect    */
/* (a timer is running), and if Dead then for longer than GossipToTheDeadTime (so that nobody  */
/* gossips to it any more) — becomes the base row's entry and all explicit views of it are     */
/* freed, also those of nodes that are not running (a node that comes back has "caught up").   */
/* The state-change time of a folded view reads 0.  Shards take the decision together: each    */
/* sends what its own observers hold (emit_fold_record) with the tick's packets.               */
/* ------------------------------------------------------------------------------------------ */
static void dirty_all(swim_sim* s, uint32_t r);
/* serf.go handleReap: every ReapInterval, at every observer the simulator acts for: Failed for longer than ReconnectTimeout
 * or Left for longer than TombstoneTimeout => erased (status NONE from then on) + EventMemberReap */
static void phase_reap(swim_sim* s) {
  if (!s->d.reap_period_ticks || !s->tick || s->tick % s->d.reap_period_ticks) return;
  uint32_t now = now_ms(s);
  for (uint32_t r = 0; r < s->R; r++)
    for (uint32_t k = 0; k < s->nloc; k++) {
      uint32_t o = s->i0 + k; if (!acts(s, r, o)) continue;
      vtab* t = &s->nodes[(size_t)r * s->nloc + k].vt;
      for (uint32_t i = 0; i < t->slots; i++) {
        view_t* v = &t->e[i]; if (v->subj == V_EMPTY || v->reaped || v->subj == o) continue;
        uint32_t st = KST(v->key);
        if (st < SWIM_STATE_DEAD || !(now - v->since > (st == SWIM_STATE_DEAD ? s->cfg.reconnect_timeout_ms : s->cfg.tombstone_timeout_ms))) continue;
        v->reaped = 1; s->st.reaped++; touch_slot(s, r, v->subj);
        if (watching(s, r, o)) record_event(s, r, SWIM_EVENT_MEMBER_REAP, v->subj, 0, KINC(v->key));
      }
    }
}
static int fold_tick(const swim_sim* s) { return s->d.fold_period_ticks && s->tick && s->tick % s->d.fold_period_ticks == 0; }
static void fold_touch(swim_sim* s, uint32_t g) {
  if (s->f_ntouched == s->f_cap) { s->f_cap = s->f_cap ? s->f_cap * 2 : 256; s->f_touched = (uint32_t*)realloc(s->f_touched, (size_t)s->f_cap * 4); }
  s->f_touched[s->f_ntouched++] = g;
}
static void fold_census(swim_sim* s) {
  if (!fold_tick(s)) return;
  uint32_t now = now_ms(s);
  s->f_ntouched = 0;
  for (uint32_t r = 0; r < s->R; r++)
    for (uint32_t k = 0; k < s->nloc; k++) {
      if (!acts(s, r, s->i0 + k)) continue;
      const node_t* nd = &s->nodes[(size_t)r * s->nloc + k];
      const vtab* t = &nd->vt;
      /* slot `slots` stands for the node's view of ITSELF when that is implicit (alive at its own incarnation) and not
       * what the base row says: it takes part in the census like an explicit view (there is nothing to free later) */
      for (uint32_t i = 0; i <= t->slots; i++) {
        uint32_t subj, key, since = 0; int unreaped = 0;
        if (i < t->slots) { const view_t* v = &t->e[i]; if (v->subj == V_EMPTY) continue; subj = v->subj; key = v->key; since = v->since; unreaped = !v->reaped; }
        else { subj = s->i0 + k; key = KEY(nd->self_inc, SWIM_STATE_ALIVE); if (vt_find(t, subj) || key == s->base_key[(size_t)r * s->N + subj]) break; }
        uint32_t g = r * s->N + subj, st = KST(key);
        if (!s->f_cnt[g]++) { fold_touch(s, g); s->f_kmin[g] = s->f_kmax[g] = key; s->f_bad[g] = 0; }
        if (key < s->f_kmin[g]) s->f_kmin[g] = key;
        if (key > s->f_kmax[g]) s->f_kmax[g] = key;
        if (st == SWIM_STATE_SUSPECT || (st == SWIM_STATE_DEAD && !(now - since > s->cfg.gossip_to_dead_ms))) s->f_bad[g] = 1;
        /* with the reaper on, a Failed / Left member stays an explicit view until serf has erased it here (the
         * EventMemberReap must not be lost); in the base row it reads as erased */
        if (s->d.reap_period_ticks && st >= SWIM_STATE_DEAD && unreaped) s->f_bad[g] = 1;
        if (i < t->slots && t->e[i].leaving) s->f_bad[g] = 1;     /* a Leaving mark is not something the base row can hold */
        if (s->q_cnt && s->q_cnt[g]) s->f_bad[g] = 1;             /* SWIM_F_UNBOUNDED_QUEUE: some node of the shard still has a rumour about it queued */
        if (i < t->slots && t->e[i].slt && st < SWIM_STATE_DEAD) s->f_bad[g] = 1;   /* ... nor a live member's statusLTime (a stale leave intent must stay stale) */
      }
    }
  for (uint32_t i = 0; i < s->f_ntouched; i++) {
    uint32_t g = s->f_touched[i];
    emit_fold_record(s, g, (!s->f_bad[g] && s->f_kmin[g] == s->f_kmax[g]) ? s->f_kmin[g] : FOLD_POISON, s->f_cnt[g]);
    s->f_cnt[g] = 0;
  }
  s->f_ntouched = 0;
}
static void fold_apply(swim_sim* s) {
  if (!fold_tick(s)) return;
  s->f_ntouched = 0;
  for (uint32_t i = 0; i < s->in.n; i++) {
    const swim_edge* e = &s->in.v[i]; if (e->dst != SWIM_NONE) continue;
    uint32_t g = e->subject;
    if (!s->f_cnt[g]) { fold_touch(s, g); s->f_kmin[g] = s->f_kmax[g] = e->incarnation; }
    s->f_cnt[g] += e->meta;
    if (e->incarnation < s->f_kmin[g]) s->f_kmin[g] = e->incarnation;
    if (e->incarnation > s->f_kmax[g]) s->f_kmax[g] = e->incarnation;
  }
  if (!s->f_ntouched) return;
  uint32_t* acting = (uint32_t*)calloc(s->R, 4);          /* acting nodes of the whole population (ground truth is replicated) */
  for (uint32_t r = 0; r < s->R; r++) for (uint32_t x = 0; x < s->N; x++) acting[r] += (uint32_t)acts(s, r, x);
  uint32_t any = 0;
  for (uint32_t i = 0; i < s->f_ntouched; i++) {
    uint32_t g = s->f_touched[i], r = g / s->N, x = g % s->N;
    int ok = s->f_kmin[g] == s->f_kmax[g] && s->f_kmax[g] != FOLD_POISON && s->f_cnt[g] == acting[r];
    s->f_cnt[g] = 0; s->f_bad[g] = (uint8_t)ok;           /* f_bad doubles as "fold this one" (1, or 2: see below) for the sweep */
    if (!ok) continue;
    any = 1;
    if (KINC(s->base_key[g]) == 0 && KINC(s->f_kmin[g]) != 0) {          /* the base row hears of it: 2 = the holders' nk falls, and its own */
      s->base_known[r]++; s->f_bad[g] = 2;
      if (is_local(s, x)) node_at(s, r, x)->nk--;
    }
    s->base_key[g] = s->f_kmin[g];
    if (is_local(s, x)) s->st.folds++;
    touch_slot(s, r, x);
  }
  free(acting);
  if (any)                                                 /* one sweep over the explicit views frees every folded subject's entries */
    for (uint32_t r = 0; r < s->R; r++)
      for (uint32_t k = 0; k < s->nloc; k++) {
        vtab* t = &s->nodes[(size_t)r * s->nloc + k].vt;
        for (uint32_t i = 0; i < t->slots; ) {
          view_t* v = &t->e[i];
          if (v->subj != V_EMPTY && s->subj_cnt[r * s->N + v->subj] && s->f_bad[r * s->N + v->subj] >= 1) {
            s->subj_cnt[r * s->N + v->subj]--; s->st.fold_freed++;
            if (s->f_bad[r * s->N + v->subj] == 2 && v->subj != s->i0 + k) s->nodes[(size_t)r * s->nloc + k].nk--;
            vt_erase(t, v);                                /* an entry may have shifted into slot i: look at it again */
          } else i++;
        }
      }
  for (uint32_t i = 0; i < s->f_ntouched; i++) s->f_bad[s->f_touched[i]] = 0;
  s->f_ntouched = 0;
}

/* sendMsg: extra := getBroadcasts(compoundOverhead, UDPBufferSize - len(msg) - compoundHeaderOverhead) — the
 * memberlist queue, then the serf delegate's user events; what is picked travels with the next tick */
static void piggyback(swim_sim* s, uint32_t r, uint32_t o, node_t* nd, uint32_t receiver, uint32_t kind) {
  qent* const msgs = pick_buf(); int32_t used = 0, used2 = 0;
  int32_t limit = (int32_t)s->d.packet_budget - (int32_t)s->cfg.ctl_len[kind & 3];
  const uint32_t rl = retransmit_limit_n(s, est_n(s, r, nd));
  uint32_t n = queue_get(s, mlq(s, r, nd), 2, limit, msgs, &used, rl);
  int32_t avail = limit - used;
  if (nd->ring && avail > 2 + 1) n += queue_get(s, evq_of(s, nd), 3, avail, msgs + n, &used2, rl);
  if (!n) return;
  s->st.piggybacks++; s->st.msgs_piggybacked += n;
  for (uint32_t m = 0; m < n; m++) s->st.msgs_sent[msgs[m].type]++;
  if (receiver == SWIM_NONE) return;
  for (uint32_t m = 0; m < n; m++) {
    if (s->attached[(size_t)r * s->N + receiver]) emit_from(s, o, r, receiver, msgs[m].subject, msgs[m].inc, msgs[m].type, msgs[m].from);
    else ev_push(&s->carry[(s->tick + 1) & 1], mk_edge(s, r, receiver, msgs[m].subject, msgs[m].inc, msgs[m].type, msgs[m].from));
  }
}

/* packetListen -> handleCommand -> handleAlive/Suspect/Dead/User for everything that arrived */
static void phase_deliver_resolve(swim_sim* s) {
  /* fold census records of all shards first: a subject every acting observer of the whole population agrees on
   * moves into the base row before this tick's arrivals are merged */
  fold_apply(s);
  /* scatter into per-node inboxes */
  for (uint32_t i = 0; i < s->in.n; i++) {
    swim_edge e = s->in.v[i]; if (e.dst == SWIM_NONE) continue;
    uint32_t r = e.dst / s->N, x = e.dst % s->N;
    if (!is_local(s, x) || !s->gt_alive[e.dst]) continue;
    if (s->attached[e.dst]) { if (e.subject == SWIM_SUBJECT_PIGGY) continue; emit_from(s, SWIM_NONE, r, x, e.subject, e.incarnation, e.meta >> 30, e.meta & 0x3FFFFFFFu); continue; }
    node_t* nd = node_at(s, r, x);
    if (e.subject == SWIM_SUBJECT_PIGGY && !nd->qlen && !nd->evqlen) continue;   /* nothing to carry */
    if (nd->in_cnt >= s->cfg.inbox_cap) { s->st.inbox_overflow++; continue; }
    nd->inbox[nd->in_cnt++] = e;
  }
  /* merge, one observer at a time, in canonical message order, duplicates applied once */
  for (uint32_t r = 0; r < s->R; r++)
    for (uint32_t k = 0; k < s->nloc; k++) {
      uint32_t o = s->i0 + k; node_t* nd = node_at(s, r, o);
      if (!nd->in_cnt) continue;
      if (nd->in_cnt > 5 && nd->in_cnt > s->st.inbox_peak) s->st.inbox_peak = nd->in_cnt;   /* (counted from six on: see swimsim.h) */
      qsort(nd->inbox, nd->in_cnt, sizeof(swim_edge), edge_cmp);
      for (uint32_t i = 0; i < nd->in_cnt; i++) {
        swim_edge e = nd->inbox[i];
        if (i && edge_cmp(&nd->inbox[i - 1], &e) == 0) continue;
        uint32_t type = e.meta >> 30, from = e.meta & 0x3FFFFFFFu;
        if (e.subject == SWIM_SUBJECT_PIGGY) { piggyback(s, r, o, nd, e.incarnation, type); continue; }
        if (e.subject == SWIM_SUBJECT_PULL && type == SWIM_MSG_ALIVE) {      /* answer next tick */
          swim_edge rq = { o, e.incarnation, r, from & 1u };
          ev_push(&s->pp_reply[(s->tick + 1) & 1], rq);
          continue;
        }
        switch (type) {
          case SWIM_MSG_ALIVE: alive_node(s, r, o, nd, e.subject, e.incarnation, from); break;
          case SWIM_MSG_SUSPECT: suspect_node(s, r, o, nd, e.subject, e.incarnation, from); break;
          case SWIM_MSG_DEAD: dead_node(s, r, o, nd, e.subject, e.incarnation, from); break;
          default: user_event(s, r, o, nd, e.subject, e.incarnation); break;
        }
      }
      nd->in_cnt = 0;
    }
}

static void census_slot(swim_sim* s, uint32_t r, slot_t* t) {
  swim_census* c = &t->census;
  c->n_observers = 0; memset(c->by_state, 0, sizeof c->by_state); c->n_current = 0;
  for (uint32_t k = 0; k < s->nloc; k++) {
    uint32_t o = s->i0 + k;
    if (o == t->node || !s->gt_alive[(size_t)r * s->N + o]) continue;
    uint32_t key = view_key(s, r, o, t->node, NULL);
    c->n_observers++; c->by_state[KST(key)]++;
    if (KINC(key) == t->max_inc) c->n_current++;
  }
  uint32_t now = now_ms(s);
  if (c->first_suspect_ms == SWIM_NONE && c->by_state[SWIM_STATE_SUSPECT]) c->first_suspect_ms = now;
  if (c->first_dead_ms == SWIM_NONE && (c->by_state[SWIM_STATE_DEAD] || c->by_state[SWIM_STATE_LEFT])) c->first_dead_ms = now;
  if (c->all_dead_ms == SWIM_NONE && c->n_observers && c->by_state[SWIM_STATE_DEAD] + c->by_state[SWIM_STATE_LEFT] == c->n_observers) c->all_dead_ms = now;
  if (c->all_current_ms == SWIM_NONE && c->n_observers && t->max_inc > 1 && c->n_current == c->n_observers) c->all_current_ms = now;
  t->dirty = 0;
}

static void phase_bookkeep(swim_sim* s) {
  for (uint32_t r = 0; r < s->R; r++)
    for (uint32_t sl = 0; sl < s->n_slots[r]; sl++) {
      slot_t* t = &s->slots[(size_t)r * s->cfg.subject_cap + sl];
      if (t->dirty) census_slot(s, r, t);
      if (t->trace && s->tick < s->cfg.trace_ticks) {
        uint32_t* row = &t->trace[(size_t)s->tick * 5];
        memcpy(row, t->census.by_state, 4 * sizeof(uint32_t)); row[4] = t->census.n_current;
      }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* C-ABI                                                                                       */
/* ------------------------------------------------------------------------------------------ */

const char* swim_backend(void) { return "oracle-c"; }
const char* swim_last_error(swim_sim* s) { return s ? s->err : "null handle"; }

int swim_create(const swim_config* cfg, swim_sim** out) {
  swim_derived d; int rc = swim_config_derive(cfg, &d); if (rc) return rc;
  if (!out) return SWIM_EINVAL;
  swim_sim* s = (swim_sim*)calloc(1, sizeof *s); if (!s) return SWIM_ENOMEM;
  s->cfg = *cfg; s->d = d; s->N = cfg->n_nodes; s->R = cfg->n_replicas;
  s->ev_words = 4 * (((cfg->event_ids_per_ltime ? cfg->event_ids_per_ltime : 14) + 2 + 3) / 4);      /* 14 ids: a 64-byte slot */
  s->nloc = s->N / cfg->n_shards; s->i0 = cfg->shard_rank * s->nloc; s->loss_q32 = cfg->loss_q32;
  size_t NT = (size_t)s->N * s->R, NL = (size_t)s->nloc * s->R;
  s->ev_watch = (uint32_t*)calloc((size_t)s->R * SWIM_EVENT_WATCHERS, 4); s->n_ev_watch = (uint32_t*)calloc(s->R, 4);
  s->gt_alive = (uint8_t*)malloc(NT); s->part = (uint8_t*)calloc(NT, 1); s->attached = (uint8_t*)calloc(NT, 1); s->alone = (uint8_t*)calloc(NT, 1);
  s->tcp_cls = (uint8_t*)calloc(NT, 1);
  s->node_slot = (uint32_t*)malloc(NT * 4); s->nodes = (node_t*)calloc(NL, sizeof(node_t));
  s->base_key = (uint32_t*)malloc(NT * 4); s->subj_cnt = (uint32_t*)calloc(NT, 4);
  s->f_cnt = (uint32_t*)calloc(NT, 4); s->f_kmin = (uint32_t*)calloc(NT, 4); s->f_kmax = (uint32_t*)calloc(NT, 4); s->f_bad = (uint8_t*)calloc(NT, 1);
  s->slots = (slot_t*)calloc((size_t)s->R * cfg->subject_cap, sizeof(slot_t));
  s->n_slots = (uint32_t*)calloc(s->R, 4); s->out = (edgevec*)calloc(cfg->n_shards, sizeof(edgevec));
  if (!s->gt_alive || !s->part || !s->attached || !s->alone || !s->node_slot || !s->nodes || !s->slots || !s->n_slots || !s->out ||
      !s->base_key || !s->subj_cnt || !s->f_cnt || !s->f_kmin || !s->f_kmax || !s->f_bad) { swim_destroy(s); return SWIM_ENOMEM; }
  memset(s->gt_alive, 1, NT); memset(s->node_slot, 0xFF, NT * 4);
  for (size_t g = 0; g < NT; g++) s->base_key[g] = BASE_KEY;
  s->base_known = (uint32_t*)calloc(s->R, 4); if (!s->base_known) { swim_destroy(s); return SWIM_ENOMEM; }
  { uint32_t ni = cfg->n_initial ? cfg->n_initial : s->N;
    s->dyn = ni < s->N;
    for (uint32_t r = 0; r < s->R; r++) {
      s->base_known[r] = ni;
      for (uint32_t x = ni; x < s->N; x++) { s->gt_alive[(size_t)r * s->N + x] = 0; s->base_key[(size_t)r * s->N + x] = KEY(0, SWIM_STATE_DEAD); }   /* not started, never heard of */
    } }
  s->unbounded = (cfg->flags & SWIM_F_UNBOUNDED_QUEUE) != 0;
  s->q_slab = s->unbounded ? NULL : (qent*)calloc(NL * cfg->queue_cap, sizeof(qent));
  s->q_cnt = s->unbounded ? (uint32_t*)calloc(NT, 4) : NULL;
  s->evq_slab = (cfg->flags & SWIM_F_SERF_EVENTS) ? (qent*)calloc(NL * cfg->event_queue_cap, sizeof(qent)) : NULL;
  s->inbox_slab = (swim_edge*)malloc(NL * cfg->inbox_cap * sizeof(swim_edge));
  if ((!s->unbounded && !s->q_slab) || (s->unbounded && !s->q_cnt) || !s->inbox_slab || ((cfg->flags & SWIM_F_SERF_EVENTS) && !s->evq_slab)) { swim_destroy(s); return SWIM_ENOMEM; }
  if (cfg->flags & SWIM_F_COORDINATES) {
    s->cs = (coord_state*)calloc(NL, sizeof(coord_state)); s->c_new = (swim_coordinate*)malloc(NL * sizeof(swim_coordinate)); s->c_list = (uint32_t*)malloc(NL * sizeof(uint32_t));
    if (!s->cs || !s->c_new || !s->c_list) { swim_destroy(s); return SWIM_ENOMEM; }
    for (size_t g = 0; g < NL; g++) coord_new(&s->cs[g].c);
  }
  for (size_t g = 0; g < NL; g++) {
    node_t* nd = &s->nodes[g];
    nd->q = s->unbounded ? NULL : s->q_slab + g * cfg->queue_cap;
    nd->evq = s->evq_slab ? s->evq_slab + g * cfg->event_queue_cap : NULL;
    nd->self_inc = 1; nd->pr_target = SWIM_NONE; nd->vdl = SWIM_NONE;
    nd->inbox = s->inbox_slab + g * cfg->inbox_cap;
    if (cfg->flags & SWIM_F_SERF_EVENTS) { nd->ring = (evslot*)calloc((size_t)cfg->event_buffer * s->ev_words, sizeof(evslot)); nd->ev_clock = 1; }   /* serf.Create: eventClock.Increment() */
    if ((cfg->flags & SWIM_F_SERF_EVENTS) && !nd->ring) { swim_destroy(s); return SWIM_ENOMEM; }
  }
  *out = s; return SWIM_OK;
}

int swim_destroy(swim_sim* s) {
  if (!s) return SWIM_EINVAL;
  if (s->nodes) for (size_t g = 0; g < (size_t)s->nloc * s->R; g++) { free(s->nodes[g].ring); free(s->nodes[g].vt.e); if (s->unbounded) free(s->nodes[g].q); }
  free(s->q_cnt);
  free(s->base_key); free(s->subj_cnt); free(s->base_known); free(s->join_list); free(s->f_cnt); free(s->f_kmin); free(s->f_kmax); free(s->f_bad); free(s->f_touched);
  free(s->inbox_slab); free(s->ev_watch); free(s->n_ev_watch);
  if (s->slots) for (size_t i = 0; i < (size_t)s->R * s->cfg.subject_cap; i++) free(s->slots[i].trace);
  if (s->out) for (uint32_t i = 0; i < s->cfg.n_shards; i++) free(s->out[i].v);
  free(s->attached); free(s->alone); free(s->tcp_cls); free(s->captured.v); free(s->cap_src); free(s->xpeers);
  free(s->q_slab); free(s->evq_slab); free(s->cs); free(s->c_new); free(s->c_list);
  free(s->gt_alive); free(s->part); free(s->node_slot); free(s->nodes); free(s->slots); free(s->n_slots);
  free(s->out); free(s->in.v); free(s->last_edges.v); free(s->pp_reply[0].v); free(s->pp_reply[1].v); free(s->carry[0].v); free(s->carry[1].v); free(s->events); free(s);
  return SWIM_OK;
}

/* the broadcasts piggy-backed on last tick's pings and acks arrive with this tick's packets; the no-op
 * filter sees them like any other rumour of the tick */
static void phase_carry(swim_sim* s) {
  edgevec* c = &s->carry[s->tick & 1];
  for (uint32_t i = 0; i < c->n; i++) {
    swim_edge e = c->v[i];
    ev_push(&s->last_edges, e);
    uint32_t r = e.dst / s->N, x = e.dst % s->N;
    qent m = { e.subject, e.incarnation, e.meta & 0x3FFFFFFFu, 0, (uint8_t)(e.meta >> 30), 0 };
    if ((s->cfg.flags & SWIM_F_FILTER_NOOP) && noop_at_receiver(s, r, x, &m)) { s->st.msgs_filtered++; continue; }
    uint32_t sh = shard_of(s, x);
    if (judged_remotely(s, x, m.subject, m.type)) { e.meta |= EDGE_JUDGE; ev_push(&s->out[sh], e); s->st.edges_remote++; continue; }
    ev_push(&s->out[sh], e);
    s->st.edges++; if (sh != s->cfg.shard_rank) s->st.edges_remote++;
  }
  c->n = 0;
}

int swim_tick_begin(swim_sim* s) {
  if (!s) return SWIM_EINVAL; if (s->in_tick) return SWIM_ESTATE;
  for (uint32_t i = 0; i < s->cfg.n_shards; i++) s->out[i].n = 0;
  s->in.n = 0;
  fold_census(s);
  phase_reap(s);
  phase_expire(s); phase_probe(s); phase_pushpull(s); phase_reconnect(s); phase_gossip(s);
  /* swim_debug_edges: what the roles emitted (orders excluded), then the carried broadcasts before the filter */
  s->last_edges.n = 0;
  for (uint32_t sh = 0; sh < s->cfg.n_shards; sh++)
    for (uint32_t i = 0; i < s->out[sh].n; i++)
      if (s->out[sh].v[i].dst != SWIM_NONE && s->out[sh].v[i].subject != SWIM_SUBJECT_PIGGY) { swim_edge e = s->out[sh].v[i]; if (e.subject != SWIM_SUBJECT_PULL) e.meta &= ~EDGE_JUDGE; ev_push(&s->last_edges, e); }
  phase_carry(s);
  s->in_tick = 1;
  /* the local segment never crosses the wire */
  edgevec* loc = &s->out[s->cfg.shard_rank];
  for (uint32_t i = 0; i < loc->n; i++) ev_push(&s->in, loc->v[i]);
  return SWIM_OK;
}
int swim_outbound(swim_sim* s, uint32_t shard, const swim_edge** ptr, uint32_t* count) {
  if (!s || !ptr || !count || shard >= s->cfg.n_shards) return SWIM_EINVAL; if (!s->in_tick) return SWIM_ESTATE;
  *ptr = s->out[shard].v; *count = s->out[shard].n; return SWIM_OK;
}
int swim_stream(swim_sim* s, void** st) { if (!s || !st) return SWIM_EINVAL; *st = NULL; return SWIM_OK; }
int swim_outbound_raw(swim_sim* s, uint32_t shard, const swim_edge** seg, const uint32_t** cnt) {
  if (!s || shard >= s->cfg.n_shards) return SWIM_EINVAL;
  if (seg) *seg = NULL; if (cnt) *cnt = NULL; return SWIM_ESTATE;     /* host buffers move: use swim_outbound */
}
int swim_peer_activity(swim_sim* s, int active) { (void)active; return s ? SWIM_OK : SWIM_EINVAL; }   /* a hint: the oracle files every order */
int swim_activity(swim_sim* s, int* active) { if (!s || !active) return SWIM_EINVAL; if (!s->in_tick) return SWIM_ESTATE; *active = 1; return SWIM_OK; }
uint32_t swim_outbound_capacity(swim_sim* s, uint32_t shard) { return (s && shard < s->cfg.n_shards) ? 0x7FFFFFFFu : 0; }
int swim_inbound(swim_sim* s, const swim_edge* ptr, uint32_t count) {
  if (!s || (!ptr && count)) return SWIM_EINVAL; if (!s->in_tick) return SWIM_ESTATE;
  for (uint32_t i = 0; i < count; i++) {
    swim_edge e = ptr[i];
    if (e.dst != SWIM_NONE && e.subject != SWIM_SUBJECT_PIGGY && e.subject != SWIM_SUBJECT_PULL && (e.meta & EDGE_JUDGE)) {   /* a rumour its sender could not judge */
      e.meta &= ~EDGE_JUDGE;
      uint32_t r = e.dst / s->N, x = e.dst % s->N;
      qent m = { e.subject, e.incarnation, e.meta & 0x3FFFFFFFu, 0, (uint8_t)(e.meta >> 30), 0 };
      if (is_local(s, x) && noop_at_receiver(s, r, x, &m)) { s->st.msgs_filtered++; continue; }
      s->st.edges++;
    }
    ev_push(&s->in, e);
  }
  return SWIM_OK;
}
/* the framed exchange (swimsim.h) on host memory: frame j of `send` = header {count, 1, tick + 1, magic} + this shard's list for shard j */
uint32_t swim_frame_records(swim_sim* s) { (void)s; return 0; }      /* the lists are unbounded here: any frame size will do */
int swim_frame_pack(swim_sim* s, swim_edge* send, uint32_t F) {
  if (!s || !send || F < 2) return SWIM_EINVAL; if (!s->in_tick) return SWIM_ESTATE;
  for (uint32_t j = 0; j < s->cfg.n_shards; j++) {
    swim_edge* fr = send + (size_t)j * F;
    uint32_t n = j == s->cfg.shard_rank ? 0 : s->out[j].n;
    if (n > F - 1) { snprintf(s->err, sizeof s->err, "bounded structure overflowed: edge-list (a frame of %u records, %u to send)", F, n); return SWIM_EOVERFLOW; }
    fr[0].dst = n; fr[0].subject = 1; fr[0].incarnation = s->tick + 1; fr[0].meta = SWIM_FRAME_MAGIC;
    if (n) memcpy(fr + 1, s->out[j].v, (size_t)n * sizeof(swim_edge));
  }
  return SWIM_OK;
}
/* ... frames sized from the load: what does not fit is not an error, the header says what there is and what the largest segment needs */
int swim_frame_pack_fill(swim_sim* s, swim_edge* send, uint32_t F) {
  if (!s || !send || F < 2) return SWIM_EINVAL; if (!s->in_tick) return SWIM_ESTATE;
  uint32_t need = 0;
  for (uint32_t j = 0; j < s->cfg.n_shards; j++) if (j != s->cfg.shard_rank && s->out[j].n > need) need = (uint32_t)s->out[j].n;
  for (uint32_t j = 0; j < s->cfg.n_shards; j++) {
    swim_edge* fr = send + (size_t)j * F;
    uint32_t n = j == s->cfg.shard_rank ? 0 : (uint32_t)s->out[j].n, fit = n < F - 1 ? n : F - 1;
    fr[0].dst = n; fr[0].subject = 1u | (need << 1); fr[0].incarnation = s->tick + 1; fr[0].meta = SWIM_FRAME_MAGIC;
    if (fit) memcpy(fr + 1, s->out[j].v, (size_t)fit * sizeof(swim_edge));
  }
  return SWIM_OK;
}
int swim_frame_deliver(swim_sim* s, const swim_edge* recv, uint32_t F) {
  if (!s || !recv || F < 2) return SWIM_EINVAL; if (!s->in_tick) return SWIM_ESTATE;
  for (uint32_t i = 0; i < s->cfg.n_shards; i++) {
    if (i == s->cfg.shard_rank) continue;
    const swim_edge* fr = recv + (size_t)i * F;
    if (fr[0].meta != SWIM_FRAME_MAGIC || fr[0].incarnation != s->tick + 1 || fr[0].dst > F - 1) {
      snprintf(s->err, sizeof s->err, "swim_frame_deliver: frame %u is not this tick's (is every shard stepping?)", i); return SWIM_ESTATE;
    }
    int rc = swim_inbound(s, fr + 1, fr[0].dst);
    if (rc) return rc;
  }
  return SWIM_OK;
}
int swim_tick_end(swim_sim* s) {
  if (!s) return SWIM_EINVAL; if (!s->in_tick) return SWIM_ESTATE;
  phase_deliver_resolve(s);
  for (uint32_t j = 0; j < s->n_join_pending; j++)          /* joined: from the next tick on it probes and gossips like everybody */
    if (s->join_list[2 * j + 1] != SWIM_NONE) { s->alone[s->join_list[2 * j]] = 0; dirty_all(s, s->join_list[2 * j] / s->N); }
  s->n_join_pending = 0;
  if (s->cs) coord_commit(s);
  phase_bookkeep(s);
  s->st.ticks++; if ((s->tick + 1) % s->d.gossip_period == 0) s->st.gossip_rounds++;
  s->tick++; s->in_tick = 0;
  return SWIM_OK;
}
/* swim_xchg_* on host memory, for shards that live in one process and are driven from one thread: the handle is the
 * shard's address; the first shard asked to run tick t runs it for the whole population (begin everywhere, hand the
 * per-destination segments over, end everywhere) and leaves the others a credit, so that every shard can make the same
 * calls as on the product library, in any order. */
int swim_xchg_export(swim_sim* s, swim_xchg_handle* out) {
  if (!s || !out) return SWIM_EINVAL;
  memset(out, 0, sizeof *out); memcpy(out->bytes, &s, sizeof s);
  return SWIM_OK;
}
int swim_xchg_connect(swim_sim* s, const swim_xchg_handle* all) {
  if (!s || !all) return SWIM_EINVAL; if (s->in_tick) return SWIM_ESTATE;
  free(s->xpeers); s->xpeers = (swim_sim**)calloc(s->cfg.n_shards, sizeof(swim_sim*)); if (!s->xpeers) return SWIM_ENOMEM;
  for (uint32_t i = 0; i < s->cfg.n_shards; i++) {
    swim_sim* p; memcpy(&p, all[i].bytes, sizeof p);
    if (i == s->cfg.shard_rank) p = s;
    if (!p || p->cfg.n_shards != s->cfg.n_shards || p->cfg.shard_rank != i || p->N != s->N || p->R != s->R) { free(s->xpeers); s->xpeers = NULL; return SWIM_EINVAL; }
    s->xpeers[i] = p;
  }
  return SWIM_OK;
}
int swim_xchg_step(swim_sim* s, uint32_t n) {
  if (!s) return SWIM_EINVAL; if (!s->xpeers || s->in_tick) return SWIM_ESTATE;
  uint32_t W = s->cfg.n_shards; int rc;
  for (uint32_t t = 0; t < n; t++) {
    if (s->xcredit) { s->xcredit--; continue; }
    for (uint32_t i = 0; i < W; i++) if ((rc = swim_tick_begin(s->xpeers[i]))) return rc;
    for (uint32_t i = 0; i < W; i++) for (uint32_t j = 0; j < W; j++) if (i != j) {
      edgevec* o = &s->xpeers[i]->out[j];
      if (o->n && (rc = swim_inbound(s->xpeers[j], o->v, o->n))) return rc;
    }
    for (uint32_t i = 0; i < W; i++) { if ((rc = swim_tick_end(s->xpeers[i]))) return rc; if (s->xpeers[i] != s) s->xpeers[i]->xcredit++; }
  }
  return SWIM_OK;
}
int swim_tick_end_begin(swim_sim* s) { int rc = swim_tick_end(s); return rc ? rc : swim_tick_begin(s); }
int swim_step(swim_sim* s, uint32_t n) {
  if (!s) return SWIM_EINVAL; if (s->cfg.n_shards != 1) return SWIM_ESTATE;
  for (uint32_t i = 0; i < n; i++) { int rc = swim_tick_begin(s); if (rc) return rc; rc = swim_tick_end(s); if (rc) return rc; }
  return SWIM_OK;
}
int swim_sync(swim_sim* s) { return s ? SWIM_OK : SWIM_EINVAL; }
int swim_now(swim_sim* s, uint32_t* tick, uint32_t* ms) {
  if (!s) return SWIM_EINVAL; if (tick) *tick = s->tick; if (ms) *ms = now_ms(s); return SWIM_OK;
}

static int chk(swim_sim* s, uint32_t r, const uint32_t* ids, size_t n) {
  if (!s || (!ids && n)) return SWIM_EINVAL; if (s->in_tick) return SWIM_ESTATE;
  if (r >= s->R) return SWIM_ERANGE;
  for (size_t i = 0; i < n; i++) if (ids[i] >= s->N) return SWIM_ERANGE;
  return SWIM_OK;
}
static void dirty_all(swim_sim* s, uint32_t r) {
  for (uint32_t sl = 0; sl < s->n_slots[r]; sl++) s->slots[(size_t)r * s->cfg.subject_cap + sl].dirty = 1;
}
int swim_inject_kill(swim_sim* s, uint32_t r, const uint32_t* ids, size_t n) {
  int rc = chk(s, r, ids, n); if (rc) return rc;
  for (size_t i = 0; i < n; i++) { s->gt_alive[(size_t)r * s->N + ids[i]] = 0; (void)alloc_slot(s, r, ids[i]); }   /* watched while slots remain */
  dirty_all(s, r); return SWIM_OK;
}
int swim_inject_revive(swim_sim* s, uint32_t r, const uint32_t* ids, size_t n) {
  int rc = chk(s, r, ids, n); if (rc) return rc;
  for (size_t i = 0; i < n; i++) {
    s->gt_alive[(size_t)r * s->N + ids[i]] = 1;
    if (is_local(s, ids[i])) { node_t* nd = node_at(s, r, ids[i]); nd->pr_target = SWIM_NONE; nd->pr_stage = 0; nd->in_cnt = 0; }
  }
  dirty_all(s, r); return SWIM_OK;
}
/* memberlist.Leave: deadNode(dead{inc, self, self}) with hasLeft() set */
int swim_inject_leave(swim_sim* s, uint32_t r, const uint32_t* ids, size_t n) {
  int rc = chk(s, r, ids, n); if (rc) return rc;
  for (size_t i = 0; i < n; i++) {
    uint32_t x = ids[i]; (void)alloc_slot(s, r, x);
    if (!is_local(s, x) || !s->gt_alive[(size_t)r * s->N + x]) continue;
    node_t* nd = node_at(s, r, x); nd->leaving = 1;
    dead_node(s, r, x, nd, x, nd->self_inc, x);
  }
  return SWIM_OK;
}
/* memberlist.UpdateNode: nextIncarnation(); aliveNode(alive{inc}, bootstrap=true) */
int swim_inject_update(swim_sim* s, uint32_t r, const uint32_t* ids, size_t n) {
  int rc = chk(s, r, ids, n); if (rc) return rc;
  for (size_t i = 0; i < n; i++) {
    uint32_t x = ids[i]; (void)alloc_slot(s, r, x);
    if (!is_local(s, x) || !s->gt_alive[(size_t)r * s->N + x]) continue;
    node_t* nd = node_at(s, r, x); nd->self_inc++;
    view_t* v = view_make(s, r, x, x);                    /* a node's view of itself always fits */
    if (v) set_view(s, r, x, v, nd->self_inc, SWIM_STATE_ALIVE, 0);
    broadcast(s, nd, x, SWIM_MSG_ALIVE, nd->self_inc, 1);
  }
  return SWIM_OK;
}
int swim_inject_partition(swim_sim* s, uint32_t r, const uint8_t* g) {
  if (!s || !g) return SWIM_EINVAL; if (s->in_tick) return SWIM_ESTATE; if (r >= s->R) return SWIM_ERANGE;
  for (uint32_t i = 0; i < s->N; i++) if (g[i] > 127) return SWIM_ERANGE;     /* 7 bits, like the product library's node word */
  memcpy(s->part + (size_t)r * s->N, g, s->N); return SWIM_OK;
}
/* serf.Create + serf.Join([via]) for nodes that are not running (see swimsim.h) */
int swim_inject_join(swim_sim* s, uint32_t r, const uint32_t* ids, size_t n, uint32_t via) {
  int rc = chk(s, r, ids, n); if (rc) return rc;
  if (via >= s->N) return SWIM_ERANGE;
  for (size_t i = 0; i < n; i++) {
    uint32_t x = ids[i]; size_t g = (size_t)r * s->N + x;
    if (s->gt_alive[g] || s->attached[g]) continue;       /* running already: Join on a live member is a no-op here */
    s->gt_alive[g] = 1; s->alone[g] = 1;                  /* up, but it knows nobody until the join push-pull went through */
    if (s->n_join_pending == s->join_cap) { s->join_cap = s->join_cap ? s->join_cap * 2 : 64; s->join_list = (uint32_t*)realloc(s->join_list, (size_t)s->join_cap * 8); }
    s->join_list[2 * s->n_join_pending] = (uint32_t)g; s->join_list[2 * s->n_join_pending + 1] = via; s->n_join_pending++;
    if (!is_local(s, x)) continue;
    node_t* nd = node_at(s, r, x);
    /* a fresh process: nothing queued, no views of its own (it holds the base row), clean probe state */
    for (uint32_t j = 0; j < nd->vt.slots; j++) if (nd->vt.e[j].subj != V_EMPTY) { s->subj_cnt[(size_t)r * s->N + nd->vt.e[j].subj]--; touch_slot(s, r, nd->vt.e[j].subj); }
    free(nd->vt.e); memset(&nd->vt, 0, sizeof nd->vt); nd->vdl = SWIM_NONE;
    nd->nk = KINC(s->base_key[g]) == 0 ? 1u : 0u;          /* it knows itself, whatever the base row says */
    if (s->q_cnt) for (uint32_t j = 0; j < nd->qlen; j++) s->q_cnt[(size_t)r * s->N + nd->q[j].subject]--;
    nd->qlen = 0; nd->evqlen = 0; nd->in_cnt = 0; nd->awareness = 0; nd->leaving = 0;
    if (s->cs) { coord_state* st = &s->cs[nd - s->nodes]; memset(st, 0, sizeof *st); coord_new(&st->c); }   /* a fresh process: a fresh coordinate client */
    nd->pr_target = SWIM_NONE; nd->pr_stage = 0; nd->pr_nack_miss = 0;
    if (KINC(s->base_key[g]) != 0 || nd->self_inc > 1 || nd->qseq) nd->self_inc++;   /* a restart: past the incarnation others may remember */
    broadcast(s, nd, x, SWIM_MSG_ALIVE, nd->self_inc, 0);   /* memberlist setAlive */
    nd->serf_leaving = 0; nd->self_slt = 0;                /* (its join intent comes with the answer to its join push-pull: phase_pushpull) */
  }
  dirty_all(s, r); return SWIM_OK;
}
int swim_watch(swim_sim* s, uint32_t r, uint32_t x) {
  if (!s) return SWIM_EINVAL; if (s->in_tick) return SWIM_ESTATE; if (r >= s->R || x >= s->N) return SWIM_ERANGE;
  return alloc_slot(s, r, x);
}
/* memberlist.Config.DisableTcpPingsForNode as Consul sets it (agent/consul/server_serf.go:222-232): no TCP fallback ping across classes */
int swim_set_tcp_class(swim_sim* s, uint32_t r, const uint32_t* ids, size_t n, uint8_t cls) {
  int rc = chk(s, r, ids, n); if (rc) return rc;
  if (cls > SWIM_TCP_CLASS_MAX) return SWIM_ERANGE;
  for (size_t a = 0; a < n; a++) s->tcp_cls[(size_t)r * s->N + ids[a]] = cls;
  return SWIM_OK;
}
int swim_set_loss(swim_sim* s, uint32_t q) { if (!s) return SWIM_EINVAL; s->loss_q32 = q; return SWIM_OK; }

/* serf.UserEvent: stamp eventClock.Time(), Increment(), handleUserEvent locally, queue */
int swim_user_event(swim_sim* s, uint32_t r, uint32_t origin, uint32_t id, uint64_t* lt) {
  int rc = chk(s, r, &origin, 1); if (rc) return rc;
  if (!(s->cfg.flags & SWIM_F_SERF_EVENTS)) return SWIM_ESTATE;
  if (id > SWIM_EVENT_ID_MAX) return SWIM_ERANGE;           /* bits 31-30 mark serf's intents (messageLeaveType), never a user event */
  if (lt) *lt = UINT64_MAX;
  if (!is_local(s, origin) || !s->gt_alive[(size_t)r * s->N + origin]) return SWIM_OK;
  node_t* nd = node_at(s, r, origin);
  uint32_t ltime = nd->ev_clock; nd->ev_clock++;
  if (lt) *lt = ltime;
  user_event(s, r, origin, nd, id, ltime);
  return SWIM_OK;
}

/* serf.RemoveFailedNode[Prune]: broadcast a leave intent on behalf of `node` (stamped with the origin's clock like any
 * serf message, handled locally first, queued for gossip) */
int swim_force_leave(swim_sim* s, uint32_t r, uint32_t origin, uint32_t node, int prune, uint64_t* lt) {
  int rc = chk(s, r, &origin, 1); if (rc) return rc;
  if (node >= s->N) return SWIM_ERANGE;
  if (!(s->cfg.flags & SWIM_F_SERF_EVENTS)) return SWIM_ESTATE;
  if (lt) *lt = UINT64_MAX;
  if (!is_local(s, origin) || !s->gt_alive[(size_t)r * s->N + origin]) return SWIM_OK;
  node_t* nd = node_at(s, r, origin);
  uint32_t ltime = nd->ev_clock; nd->ev_clock++;
  if (lt) *lt = ltime;
  if (node == origin) nd->serf_leaving = 1;                /* serf.Leave(): its own leave intent — from now on it does not refute one */
  user_event_from(s, r, origin, nd, SWIM_INTENT_LEAVE | (prune ? SWIM_INTENT_PRUNE : 0u) | node, ltime, 1);
  return SWIM_OK;
}
static uint8_t status_of(uint32_t st) {
  return st == SWIM_STATE_DEAD ? SWIM_MEMBER_FAILED : st == SWIM_STATE_LEFT ? SWIM_MEMBER_LEFT : SWIM_MEMBER_ALIVE;
}
int swim_view(swim_sim* s, uint32_t r, uint32_t o, uint32_t x, swim_member* out) {
  if (!s || !out) return SWIM_EINVAL; if (r >= s->R || o >= s->N || x >= s->N) return SWIM_ERANGE;
  if (!is_local(s, o)) return SWIM_ERANGE;
  view_t* v = view_ptr(s, r, o, x);
  memset(out, 0, sizeof *out); out->id = x;
  uint32_t key = v ? v->key : implicit_key(s, r, o, x);
  out->incarnation = KINC(key); out->state = (uint8_t)KST(key); out->state_change_ms = v ? v->since : 0;
  out->n_confirm = v && KST(key) == SWIM_STATE_SUSPECT ? v->nconf : 0;
  out->status = KINC(key) == 0 ? (uint8_t)SWIM_MEMBER_NONE : status_of(KST(key));   /* incarnation 0: never heard of it */
  /* erased by serf's reaper (or a prune); a Failed / Left member of the base row was erased everywhere before it got there */
  if (KST(key) >= SWIM_STATE_DEAD && x != o && (v ? v->reaped : s->d.reap_period_ticks != 0)) out->status = SWIM_MEMBER_NONE;
  if (v && v->leaving && KST(key) < SWIM_STATE_DEAD && x != o) out->status = SWIM_MEMBER_LEAVING;   /* a leave intent was seen */
  if (x == o && node_at(s, r, o)->leaving && out->state == SWIM_STATE_ALIVE) out->status = SWIM_MEMBER_LEAVING;
  return SWIM_OK;
}
int swim_members(swim_sim* s, uint32_t r, uint32_t o, swim_member* out, size_t cap, size_t* n_out) {
  if (!s || (!out && cap)) return SWIM_EINVAL; if (r >= s->R || o >= s->N) return SWIM_ERANGE;
  if (!is_local(s, o)) return SWIM_ERANGE;
  for (uint32_t x = 0; x < s->N && x < cap; x++) swim_view(s, r, o, x, &out[x]);
  if (n_out) *n_out = s->N; return SWIM_OK;
}
int swim_watch_events(swim_sim* s, uint32_t r, uint32_t node) {
  if (!s) return SWIM_EINVAL; if (r >= s->R || node >= s->N || !is_local(s, node)) return SWIM_ERANGE;
  if (node == s->cfg.watch_node) return SWIM_OK;
  for (uint32_t j = 0; j < s->n_ev_watch[r]; j++) if (s->ev_watch[(size_t)r * SWIM_EVENT_WATCHERS + j] == node) return SWIM_OK;
  if (s->n_ev_watch[r] >= SWIM_EVENT_WATCHERS) return SWIM_EOVERFLOW;
  s->ev_watch[(size_t)r * SWIM_EVENT_WATCHERS + s->n_ev_watch[r]++] = node;
  return SWIM_OK;
}
/* events leave in (time, replica, observer) order, each observer's own events in the order they happened (stable) */
static int event_before(const swim_event* a, const swim_event* b) {
  if (a->time_ms != b->time_ms) return a->time_ms < b->time_ms;
  if (a->replica != b->replica) return a->replica < b->replica;
  return a->observer < b->observer;
}
int swim_poll_events(swim_sim* s, swim_event* out, size_t cap, size_t* n_out) {
  if (!s || (!out && cap) || !n_out) return SWIM_EINVAL;
  for (size_t i = s->n_sorted; i < s->n_events; i++) {            /* insertion sort of what came in since the last poll: nearly in order already */
    swim_event e = s->events[i]; size_t j = i;
    while (j > s->n_sorted && event_before(&e, &s->events[j - 1])) { s->events[j] = s->events[j - 1]; j--; }
    s->events[j] = e;
  }
  size_t n = s->n_events < cap ? s->n_events : cap;
  memcpy(out, s->events, n * sizeof(swim_event));
  memmove(s->events, s->events + n, (s->n_events - n) * sizeof(swim_event));
  s->n_events -= n; s->n_sorted = s->n_events; *n_out = n; return SWIM_OK;
}
static int rumour_cmp(const void* a, const void* b) {
  const swim_rumour *x = (const swim_rumour*)a, *y = (const swim_rumour*)b;
  return x->seq < y->seq ? -1 : x->seq > y->seq;
}
int swim_node_info_get(swim_sim* s, uint32_t r, uint32_t i, swim_node_info* out) {
  if (!s || !out) return SWIM_EINVAL; if (r >= s->R || i >= s->N || !is_local(s, i)) return SWIM_ERANGE;
  node_t* nd = node_at(s, r, i); memset(out, 0, sizeof *out);
  out->incarnation = nd->self_inc; out->probe_target = nd->pr_target;
  out->probe_deadline_tick = nd->pr_target == SWIM_NONE ? 0 : nd->pr_deadline;
  out->probe_cursor = nd->pr_cursor; out->probe_epoch = nd->pr_epoch;
  out->queue_len = nd->qlen; out->event_queue_len = nd->evqlen; out->event_clock = nd->ev_clock;
  out->alive = s->gt_alive[(size_t)r * s->N + i]; out->leaving = nd->leaving; out->awareness = nd->awareness;
  out->partition = s->part[(size_t)r * s->N + i];
  /* the struct holds 32; a queue may be deeper (queue_len says how deep): the 32 oldest entries (lowest sequence numbers), oldest first */
  const uint32_t nq = nd->qlen < 32 ? nd->qlen : 32;
  swim_rumour* all = (swim_rumour*)malloc((size_t)(nd->qlen ? nd->qlen : 1) * sizeof(swim_rumour)); if (!all) return SWIM_ENOMEM;
  for (uint32_t k = 0; k < nd->qlen; k++) {
    swim_rumour q = { nd->q[k].subject, nd->q[k].inc, nd->q[k].from, nd->q[k].type, nd->q[k].transmits, {0, 0}, nd->q[k].seq };
    all[k] = q;
  }
  qsort(all, nd->qlen, sizeof(swim_rumour), rumour_cmp);
  memcpy(out->queue, all, (size_t)nq * sizeof(swim_rumour)); free(all);
  return SWIM_OK;
}
/* serf's notifyCh of a broadcast, as a question: is {id, ltime} still in the node's serf queue? */
int swim_event_queued(swim_sim* s, uint32_t r, uint32_t i, uint32_t id, uint64_t ltime, int* queued) {
  if (!s || !queued) return SWIM_EINVAL; if (r >= s->R || i >= s->N || !is_local(s, i)) return SWIM_ERANGE;
  node_t* nd = node_at(s, r, i); *queued = 0;
  for (uint32_t k = 0; k < nd->evqlen; k++) if (nd->evq[k].subject == id && nd->evq[k].inc == (uint32_t)ltime) *queued = 1;
  return SWIM_OK;
}
int swim_census_get(swim_sim* s, uint32_t r, uint32_t x, swim_census* out) {
  if (!s || !out) return SWIM_EINVAL; if (r >= s->R || x >= s->N) return SWIM_ERANGE;
  uint32_t sl = s->node_slot[(size_t)r * s->N + x];
  if (sl == SWIM_NONE) {            /* not watched: counted on demand, no history */
    slot_t tmp; memset(&tmp, 0, sizeof tmp);
    tmp.node = x; tmp.max_inc = current_max_inc(s, r, x);
    census_slot(s, r, &tmp);
    *out = tmp.census;
    out->first_suspect_ms = out->first_dead_ms = out->all_dead_ms = out->all_current_ms = SWIM_NONE;
    return SWIM_OK;
  }
  slot_t* t = &s->slots[(size_t)r * s->cfg.subject_cap + sl];
  if (t->dirty) census_slot(s, r, t);
  *out = t->census; return SWIM_OK;
}
/* config #4's deliverable: how the acting observers of this shard see the nodes out of their reach (swimsim.h).  What the
 * base row says about x is held by every observer without an explicit view: counted per subject in closed form, then
 * corrected by the explicit views. */
int swim_detection_get(swim_sim* s, uint32_t r, swim_detection* out) {
  if (!s || !out) return SWIM_EINVAL; if (r >= s->R) return SWIM_ERANGE;
  memset(out, 0, sizeof *out);
  uint64_t cnt[256] = { 0 }, total = 0; size_t base = (size_t)r * s->N;
  for (uint32_t k = 0; k < s->nloc; k++) { uint32_t o = s->i0 + k; if (acts(s, r, o)) { cnt[s->part[base + o]]++; total++; } }
  for (uint32_t x = 0; x < s->N; x++) {
    uint64_t n_obs = !s->gt_alive[base + x] ? total : total - cnt[s->part[base + x]];
    out->pairs += n_obs; out->by_state[KST(s->base_key[base + x])] += n_obs;
  }
  for (uint32_t k = 0; k < s->nloc; k++) {
    uint32_t o = s->i0 + k; if (!acts(s, r, o)) continue;
    const vtab* t = &s->nodes[(size_t)r * s->nloc + k].vt;
    for (uint32_t i = 0; i < t->slots; i++) {
      const view_t* v = &t->e[i]; if (v->subj == V_EMPTY || v->subj == o) continue;
      if (s->gt_alive[base + v->subj] && s->part[base + v->subj] == s->part[base + o]) continue;   /* within reach */
      out->by_state[KST(s->base_key[base + v->subj])]--; out->by_state[KST(v->key)]++;
    }
  }
  return SWIM_OK;
}
int swim_trace_read(swim_sim* s, uint32_t r, uint32_t x, uint32_t first, uint32_t n, uint32_t* rows) {
  if (!s || !rows) return SWIM_EINVAL; if (r >= s->R || x >= s->N) return SWIM_ERANGE;
  uint32_t sl = s->node_slot[(size_t)r * s->N + x];
  if (sl == SWIM_NONE || !s->cfg.trace_ticks) return SWIM_ESTATE;
  if ((uint64_t)first + n > s->cfg.trace_ticks || first + n > s->tick) return SWIM_ERANGE;
  memcpy(rows, &s->slots[(size_t)r * s->cfg.subject_cap + sl].trace[(size_t)first * 5], (size_t)n * 5 * 4);
  return SWIM_OK;
}
int swim_stats(swim_sim* s, swim_stats_t* out) { if (!s || !out) return SWIM_EINVAL; *out = s->st; return SWIM_OK; }
/* layout introspection: the checker has neither tile buckets nor a mailbox nor device memory */
int swim_info(swim_sim* s, uint32_t what, uint64_t* out) {
  if (!s || !out || what > SWIM_INFO_DEVICE_BYTES) return SWIM_EINVAL;
  *out = 0; return SWIM_OK;
}
int swim_debug_edges(swim_sim* s, swim_edge* out, size_t cap, size_t* n_out) {
  if (!s || (!out && cap) || !n_out) return SWIM_EINVAL;
  size_t n = s->last_edges.n < cap ? s->last_edges.n : cap;
  memcpy(out, s->last_edges.v, n * sizeof(swim_edge)); *n_out = s->last_edges.n; return SWIM_OK;
}

/* order-independent digest: wrapping sum of per-item hashes, every item tagged with its global
 * identity, so the value does not depend on slot numbering, queue order or sharding layout
 * (shards add their digests) */
static uint64_t h3(uint64_t tag, uint64_t a, uint64_t b) { return mix64(mix64(tag * 0x9E3779B97F4A7C15ull + a) ^ (b + 0x7F4A7C15ull)); }
int swim_state_digest(swim_sim* s, uint64_t* out) {
  if (!s || !out) return SWIM_EINVAL;
  uint64_t d = 0;
  for (uint32_t r = 0; r < s->R; r++) {
    for (uint32_t k = 0; k < s->nloc; k++) {
      uint32_t i = s->i0 + k; node_t* nd = node_at(s, r, i); uint64_t g = (uint64_t)r * s->N + i;
      d += h3(1, g, ((uint64_t)nd->self_inc << 32) | ((uint64_t)nd->awareness << 8) | nd->leaving);
      d += h3(2, g, ((uint64_t)nd->pr_cursor << 32) | nd->pr_epoch);
      if (nd->pr_target != SWIM_NONE)
        d += h3(3, g, ((uint64_t)nd->pr_target << 32) | nd->pr_deadline) + h3(4, g, ((uint64_t)nd->pr_inc << 32) | ((uint64_t)nd->pr_stage << 8) | nd->pr_nack_miss);
      for (uint32_t q = 0; q < nd->qlen; q++) {
        const qent* e = &nd->q[q];
        d += h3(5, g, h3(e->subject, ((uint64_t)e->inc << 32) | e->from, ((uint64_t)e->seq << 16) | ((uint64_t)e->transmits << 8) | e->type));
      }
      d += h3(6, g, ((uint64_t)nd->qseq << 32) | nd->ev_clock);
      if (nd->self_slt | nd->serf_leaving) d += h3(19, g, ((uint64_t)nd->serf_leaving << 32) | nd->self_slt);
      for (uint32_t q = 0; q < nd->evqlen; q++) {
        const qent* e = &nd->evq[q];
        d += h3(7, g, h3(e->subject, e->inc, ((uint64_t)e->seq << 16) | ((uint64_t)e->transmits << 8)));
      }
      if (nd->ring)
        for (uint32_t b = 0; b < s->cfg.event_buffer; b++)
          { const uint32_t* sl = &nd->ring[(size_t)b * s->ev_words]; for (uint32_t j = 0; j < sl[1]; j++) d += h3(8, g, ((uint64_t)sl[0] << 32) | sl[2 + j]); }
    }
    for (uint32_t k = 0; k < s->nloc; k++) {               /* explicit views */
      const vtab* t = &s->nodes[(size_t)r * s->nloc + k].vt;
      for (uint32_t i = 0; i < t->slots; i++) {
        const view_t* v = &t->e[i]; if (v->subj == V_EMPTY) continue;
        uint64_t id = ((uint64_t)r << 40) ^ ((uint64_t)v->subj * 0x100000001B3ull) ^ ((uint64_t)(s->i0 + k) << 8);
        d += h3(9, id, ((uint64_t)v->key << 32) | v->since);
        if (v->reaped) d += h3(16, id, 1);
        if (v->leaving) d += h3(17, id, 1);
        if (v->slt) d += h3(18, id, v->slt);
        if (KST(v->key) == SWIM_STATE_SUSPECT) {
          d += h3(10, id, v->nconf);
          /* the accusers that can still matter: Confirm() returns early once k confirmations are in, so the name of the
           * k-th confirmer is never looked at again (the first accuser always is hashed) */
          for (uint32_t j = 0; j <= v->nconf && j < CONF_MAX && (j == 0 || j < suspicion_k_n(s, v->n0)); j++) d += h3(11 + j, id, v->conf[j]);
        }
      }
    }
    if (s->cs) for (uint32_t k = 0; k < s->nloc; k++) {      /* coordinates: the raw bits */
      const coord_state* st = &s->cs[(size_t)r * s->nloc + k]; uint64_t g = (uint64_t)r * s->N + s->i0 + k, bits;
      const double* f = (const double*)&st->c;
      for (uint32_t j = 0; j < SWIM_COORD_DIMS + 3; j++) { memcpy(&bits, &f[j], 8); d += h3(20 + j, g, bits); }
    }
    for (uint32_t k = 0; k < s->nloc; k++) {               /* the base row (replicated: every shard digests its own id range) */
      uint64_t g = (uint64_t)r * s->N + s->i0 + k;
      if (s->base_key[g] != BASE_KEY) d += h3(15, g, s->base_key[g]);
    }
  }
  *out = d; return SWIM_OK;
}

/* memberlist.Transport bridge at rumour granularity (the msgpack codec is the host shim's job).
 * The first call naming `a` attaches it: the simulator stops acting for it, peers keep seeing it alive. */
static int attach(swim_sim* s, uint32_t r, uint32_t a) {
  if (!s) return SWIM_EINVAL; if (s->in_tick || s->cfg.n_shards != 1) return SWIM_ESTATE;   /* unsharded populations only */
  if (r >= s->R || a >= s->N) return SWIM_ERANGE;
  size_t g = (size_t)r * s->N + a;
  if (!s->attached[g] && is_local(s, a)) {   /* from now on the node is driven from outside: what it had queued or received is void */
    node_t* nd = node_at(s, r, a);
    if (s->q_cnt) for (uint32_t j = 0; j < nd->qlen; j++) s->q_cnt[(size_t)r * s->N + nd->q[j].subject]--;
    nd->qlen = 0; nd->evqlen = 0; nd->in_cnt = 0;
  }
  s->attached[g] = 1;
  return SWIM_OK;
}
/* Transport.WriteToAddress: a packet of n rumours from the attached node to a virtual peer; it is in the
 * peer's inbox at once and merged at the end of the next tick */
int swim_transport_write_to(swim_sim* s, uint32_t r, uint32_t a, uint32_t dst, const swim_edge* m, size_t n) {
  int rc = attach(s, r, a); if (rc) return rc;
  if (dst >= s->N || (!m && n)) return SWIM_EINVAL;
  for (size_t i = 0; i < n; i++)
    if ((m[i].meta >> 30) != SWIM_MSG_USER ? m[i].subject >= s->N : m[i].subject > SWIM_EVENT_ID_MAX) return SWIM_ERANGE;   /* (a user event never carries intent bits) */
  if (!is_local(s, dst) || !s->gt_alive[(size_t)r * s->N + dst] || s->attached[(size_t)r * s->N + dst]) return SWIM_OK;
  node_t* nd = node_at(s, r, dst);
  for (size_t i = 0; i < n; i++) {
    if (nd->in_cnt >= s->cfg.inbox_cap) { s->st.inbox_overflow++; continue; }
    swim_edge e = { r * s->N + dst, m[i].subject, m[i].incarnation, m[i].meta };
    nd->inbox[nd->in_cnt++] = e;
  }
  return SWIM_OK;
}
/* Transport.PacketCh: rumours virtual peers sent to the attached node since the last poll; out[i].dst
 * carries the SENDER (Packet.From), SWIM_NONE when it is not a gossip packet */
int swim_transport_poll(swim_sim* s, uint32_t r, uint32_t a, swim_edge* o, size_t cap, size_t* n) {
  int rc = attach(s, r, a); if (rc) return rc;
  if (!n || (!o && cap)) return SWIM_EINVAL;
  size_t w = 0, keep = 0; uint32_t g = r * s->N + a;
  for (uint32_t i = 0; i < s->captured.n; i++) {
    if (s->captured.v[i].dst == g && w < cap) { o[w] = s->captured.v[i]; o[w].dst = s->cap_src[i]; w++; }
    else { s->captured.v[keep] = s->captured.v[i]; s->cap_src[keep] = s->cap_src[i]; keep++; }
  }
  s->captured.n = (uint32_t)keep; *n = w;
  return SWIM_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* checkpoint / resume: the whole population between two ticks, into a file and    */
/* back into a handle created from the SAME configuration.  The file is private to the library */
/* that wrote it (the backend string is part of the header); the product library writes its    */
/* own.  Nothing of this is upstream's format: serf's snapshotter keeps ONE node's member list */
/* (conf.SnapshotPath, agent/consul/server_serf.go:236-239); this is the simulator's state.    */
/* ------------------------------------------------------------------------------------------ */
typedef struct { char magic[8], backend[16]; uint32_t abi, tick, loss_q32, n_join_pending, f_ntouched, xcredit; uint64_t n_events; swim_config cfg; swim_stats_t st; } ck_header;
static int ck_wr(FILE* f, const void* p, size_t n) { return n == 0 || fwrite(p, 1, n, f) == n; }
static int ck_rd(FILE* f, void* p, size_t n) { return n == 0 || fread(p, 1, n, f) == n; }
static int ck_wr_vec(FILE* f, const edgevec* e) { return ck_wr(f, &e->n, 4) && ck_wr(f, e->v, (size_t)e->n * sizeof(swim_edge)); }
static int ck_rd_vec(FILE* f, edgevec* e) {
  uint32_t n; if (!ck_rd(f, &n, 4)) return 0;
  if (n > e->cap) { swim_edge* v = (swim_edge*)realloc(e->v, (size_t)n * sizeof(swim_edge)); if (!v) return 0; e->v = v; e->cap = n; }
  e->n = n; return ck_rd(f, e->v, (size_t)n * sizeof(swim_edge));
}
int swim_checkpoint_save(swim_sim* s, const char* path) {
  if (!s || !path) return SWIM_EINVAL;
  if (s->in_tick || s->c_n) return SWIM_ESTATE;
  if (s->unbounded) { snprintf(s->err, sizeof s->err, "the checker does not checkpoint unbounded queues"); return SWIM_ESTATE; }
  FILE* f = fopen(path, "wb"); if (!f) return SWIM_EIO;
  const size_t NT = (size_t)s->N * s->R, NL = (size_t)s->nloc * s->R, NS = (size_t)s->R * s->cfg.subject_cap;
  ck_header h; memset(&h, 0, sizeof h);
  memcpy(h.magic, "SWIMCKPT", 8); strncpy(h.backend, swim_backend(), sizeof h.backend - 1);
  h.abi = SWIM_ABI_VERSION; h.tick = s->tick; h.loss_q32 = s->loss_q32; h.n_join_pending = s->n_join_pending; h.f_ntouched = s->f_ntouched; h.xcredit = s->xcredit;
  h.n_events = s->n_events; h.cfg = s->cfg; h.st = s->st;
  int ok = ck_wr(f, &h, sizeof h);
  ok = ok && ck_wr(f, s->gt_alive, NT) && ck_wr(f, s->part, NT) && ck_wr(f, s->attached, NT) && ck_wr(f, s->alone, NT) && ck_wr(f, s->tcp_cls, NT);
  ok = ok && ck_wr(f, s->node_slot, NT * 4) && ck_wr(f, s->base_key, NT * 4) && ck_wr(f, s->subj_cnt, NT * 4);
  ok = ok && ck_wr(f, s->f_cnt, NT * 4) && ck_wr(f, s->f_kmin, NT * 4) && ck_wr(f, s->f_kmax, NT * 4) && ck_wr(f, s->f_bad, NT);
  ok = ok && ck_wr(f, s->f_touched, (size_t)s->f_ntouched * 4) && ck_wr(f, s->join_list, (size_t)s->n_join_pending * 8);
  ok = ok && ck_wr(f, s->base_known, (size_t)s->R * 4) && ck_wr(f, s->n_slots, (size_t)s->R * 4);
  ok = ok && ck_wr(f, s->nodes, NL * sizeof(node_t)) && ck_wr(f, s->q_slab, NL * s->cfg.queue_cap * sizeof(qent));
  if (s->evq_slab) ok = ok && ck_wr(f, s->evq_slab, NL * s->cfg.event_queue_cap * sizeof(qent));
  for (size_t g = 0; ok && g < NL; g++) {
    const node_t* nd = &s->nodes[g];
    if (nd->ring) ok = ck_wr(f, nd->ring, (size_t)s->cfg.event_buffer * s->ev_words * sizeof(evslot));
    ok = ok && ck_wr(f, nd->vt.e, (size_t)nd->vt.slots * sizeof(view_t));       /* (n and slots travel with the node) */
  }
  ok = ok && ck_wr(f, s->slots, NS * sizeof(slot_t));
  for (size_t i = 0; ok && i < NS; i++) if (s->slots[i].trace) ok = ck_wr(f, s->slots[i].trace, (size_t)s->cfg.trace_ticks * 5 * 4);
  for (uint32_t i = 0; ok && i < s->cfg.n_shards; i++) ok = ck_wr_vec(f, &s->out[i]);
  ok = ok && ck_wr_vec(f, &s->in) && ck_wr_vec(f, &s->last_edges) && ck_wr_vec(f, &s->carry[0]) && ck_wr_vec(f, &s->carry[1]);
  ok = ok && ck_wr_vec(f, &s->pp_reply[0]) && ck_wr_vec(f, &s->pp_reply[1]) && ck_wr_vec(f, &s->captured) && ck_wr(f, s->cap_src, (size_t)s->captured.n * 4);
  ok = ok && ck_wr(f, s->events, (size_t)s->n_events * sizeof(swim_event));
  ok = ok && ck_wr(f, s->ev_watch, (size_t)s->R * SWIM_EVENT_WATCHERS * 4) && ck_wr(f, s->n_ev_watch, (size_t)s->R * 4);
  if (s->cs) ok = ok && ck_wr(f, s->cs, NL * sizeof(coord_state));
  if (fclose(f) != 0) ok = 0;
  return ok ? SWIM_OK : SWIM_EIO;
}
int swim_checkpoint_load(swim_sim* s, const char* path) {
  if (!s || !path) return SWIM_EINVAL;
  if (s->in_tick) return SWIM_ESTATE;
  FILE* f = fopen(path, "rb"); if (!f) return SWIM_EIO;
  const size_t NT = (size_t)s->N * s->R, NL = (size_t)s->nloc * s->R, NS = (size_t)s->R * s->cfg.subject_cap;
  ck_header h;
  if (!ck_rd(f, &h, sizeof h)) { fclose(f); return SWIM_EINVAL; }
  if (memcmp(h.magic, "SWIMCKPT", 8) || strncmp(h.backend, swim_backend(), sizeof h.backend) || h.abi != SWIM_ABI_VERSION || memcmp(&h.cfg, &s->cfg, sizeof h.cfg)) {
    fclose(f); snprintf(s->err, sizeof s->err, "checkpoint of another library, ABI or configuration"); return SWIM_EINVAL;
  }
  int ok = 1;
  ok = ok && ck_rd(f, s->gt_alive, NT) && ck_rd(f, s->part, NT) && ck_rd(f, s->attached, NT) && ck_rd(f, s->alone, NT) && ck_rd(f, s->tcp_cls, NT);
  ok = ok && ck_rd(f, s->node_slot, NT * 4) && ck_rd(f, s->base_key, NT * 4) && ck_rd(f, s->subj_cnt, NT * 4);
  ok = ok && ck_rd(f, s->f_cnt, NT * 4) && ck_rd(f, s->f_kmin, NT * 4) && ck_rd(f, s->f_kmax, NT * 4) && ck_rd(f, s->f_bad, NT);
  if (ok && h.f_ntouched > s->f_cap) { uint32_t* v = (uint32_t*)realloc(s->f_touched, (size_t)h.f_ntouched * 4); if (v) { s->f_touched = v; s->f_cap = h.f_ntouched; } else ok = 0; }
  if (ok && h.n_join_pending > s->join_cap) { uint32_t* v = (uint32_t*)realloc(s->join_list, (size_t)h.n_join_pending * 8); if (v) { s->join_list = v; s->join_cap = h.n_join_pending; } else ok = 0; }
  ok = ok && ck_rd(f, s->f_touched, (size_t)h.f_ntouched * 4) && ck_rd(f, s->join_list, (size_t)h.n_join_pending * 8);
  ok = ok && ck_rd(f, s->base_known, (size_t)s->R * 4) && ck_rd(f, s->n_slots, (size_t)s->R * 4);
  /* the nodes: plain data except four pointers, which stay this handle's (the view table is re-sized to what was saved) */
  for (size_t g = 0; ok && g < NL; g++) {
    node_t* nd = &s->nodes[g], keep = *nd;
    ok = ck_rd(f, nd, sizeof *nd);
    nd->q = keep.q; nd->evq = keep.evq; nd->ring = keep.ring; nd->inbox = keep.inbox; nd->vt.e = keep.vt.e;
    if (ok && nd->vt.slots != keep.vt.slots) {
      view_t* e = nd->vt.slots ? (view_t*)malloc((size_t)nd->vt.slots * sizeof(view_t)) : NULL;
      if (nd->vt.slots && !e) { nd->vt.slots = keep.vt.slots; nd->vt.n = 0; ok = 0; } else { free(nd->vt.e); nd->vt.e = e; }
    }
  }
  ok = ok && ck_rd(f, s->q_slab, NL * s->cfg.queue_cap * sizeof(qent));
  if (s->evq_slab) ok = ok && ck_rd(f, s->evq_slab, NL * s->cfg.event_queue_cap * sizeof(qent));
  for (size_t g = 0; ok && g < NL; g++) {
    node_t* nd = &s->nodes[g];
    if (nd->ring) ok = ck_rd(f, nd->ring, (size_t)s->cfg.event_buffer * s->ev_words * sizeof(evslot));
    ok = ok && ck_rd(f, nd->vt.e, (size_t)nd->vt.slots * sizeof(view_t));
  }
  for (size_t i = 0; ok && i < NS; i++) {                 /* the watch slots: plain data except the trace pointer */
    slot_t* t = &s->slots[i]; uint32_t* mine = t->trace;
    ok = ck_rd(f, t, sizeof *t);
    const int had = t->trace != NULL; t->trace = mine;
    if (ok && had && !t->trace) { t->trace = (uint32_t*)calloc((size_t)s->cfg.trace_ticks * 5, 4); ok = t->trace != NULL; }
    if (ok && !had && t->trace) { free(t->trace); t->trace = NULL; }
  }
  for (size_t i = 0; ok && i < NS; i++) if (s->slots[i].trace) ok = ck_rd(f, s->slots[i].trace, (size_t)s->cfg.trace_ticks * 5 * 4);
  for (uint32_t i = 0; ok && i < s->cfg.n_shards; i++) ok = ck_rd_vec(f, &s->out[i]);
  ok = ok && ck_rd_vec(f, &s->in) && ck_rd_vec(f, &s->last_edges) && ck_rd_vec(f, &s->carry[0]) && ck_rd_vec(f, &s->carry[1]);
  ok = ok && ck_rd_vec(f, &s->pp_reply[0]) && ck_rd_vec(f, &s->pp_reply[1]) && ck_rd_vec(f, &s->captured);
  if (ok && s->captured.n > s->cap_src_cap) { uint32_t* v = (uint32_t*)realloc(s->cap_src, (size_t)s->captured.n * 4); if (v) { s->cap_src = v; s->cap_src_cap = s->captured.n; } else ok = 0; }
  ok = ok && ck_rd(f, s->cap_src, (size_t)s->captured.n * 4);
  if (ok && h.n_events > s->cap_events) { swim_event* v = (swim_event*)realloc(s->events, (size_t)h.n_events * sizeof(swim_event)); if (v) { s->events = v; s->cap_events = (size_t)h.n_events; } else ok = 0; }
  ok = ok && ck_rd(f, s->events, (size_t)h.n_events * sizeof(swim_event));
  ok = ok && ck_rd(f, s->ev_watch, (size_t)s->R * SWIM_EVENT_WATCHERS * 4) && ck_rd(f, s->n_ev_watch, (size_t)s->R * 4);
  if (s->cs) ok = ok && ck_rd(f, s->cs, NL * sizeof(coord_state));
  fclose(f);
  if (!ok) { snprintf(s->err, sizeof s->err, "checkpoint truncated or out of memory: the handle's state is undefined"); return SWIM_EIO; }
  s->tick = h.tick; s->loss_q32 = h.loss_q32; s->n_join_pending = h.n_join_pending; s->f_ntouched = h.f_ntouched; s->xcredit = h.xcredit;
  s->n_events = (size_t)h.n_events; s->n_sorted = 0; s->st = h.st; s->c_n = 0; s->in_tick = 0;
  return SWIM_OK;
}

int swim_profile(swim_sim* s, int enable) { (void)enable; return s ? SWIM_OK : SWIM_EINVAL; }
int swim_profile_read(swim_sim* s, swim_kernel_time* out, size_t cap, size_t* n_out) {
  (void)out; (void)cap; if (!s || !n_out) return SWIM_EINVAL; *n_out = 0; return SWIM_OK;
}

/* known-answer hooks */
void swim_kat_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox4x32(ctr, key, out); }
uint32_t swim_kat_probe_perm(uint64_t seed, uint32_t n, uint32_t node, uint32_t epoch, uint32_t index) { return probe_perm(seed, n, node, epoch, index); }
int32_t swim_kat_remaining_suspicion_ms(uint32_t n, uint32_t k, uint32_t el, uint32_t mn, uint32_t mx) { return (int32_t)remaining_suspicion_ms(n, k, el, mn, mx); }
uint32_t swim_kat_awareness_apply(uint32_t max_mult, uint32_t score, int32_t delta) { return (uint32_t)awareness_next((int)max_mult, (int)score, (int)delta); }
uint32_t swim_kat_awareness_scale_ms(uint32_t score, uint32_t timeout_ms) { return timeout_ms * (score + 1); }   /* awareness.ScaleTimeout */
void swim_kat_phase_of(const swim_config* cfg, uint32_t node, uint32_t* gp, uint32_t* pp) {
  swim_derived d; if (swim_config_derive(cfg, &d)) { if (gp) *gp = SWIM_NONE; if (pp) *pp = SWIM_NONE; return; }
  uint32_t c = node / d.phase_chunk;
  if (gp) *gp = c % d.gossip_period; if (pp) *pp = (c / d.gossip_period) % d.probe_period;
}

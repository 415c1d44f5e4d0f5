/*
 * swimsim.h — C-ABI of the MI355X-native simulator of Consul's Serf/memberlist SWIM hot path.
 *
 * This header is the drop-in boundary.  Two shared libraries export
 * exactly these symbols:
 *
 *   consul_amd/libswimsim.so      the product: hand-written HIP kernels for gfx950
 *   oracle/_build/libswim_oracle.so  TEST INFRASTRUCTURE ONLY: the plain-C CPU restatement
 *
 * The reference's hot path lives in two un-vendored Go modules (reference go.mod:80
 * github.com/hashicorp/memberlist v0.6.0, go.mod:85 github.com/hashicorp/serf v0.10.4); each
 * entry point below names the upstream function it replaces and the Consul call/config site
 * that reaches it (paths relative to the reference checkout).
 *
 * Conventions: every function returns 0 on success or a negative
 * SWIM_E* code; the caller owns every buffer; there are no callbacks (poll model, so a cgo
 * caller never re-enters Go from a HIP host thread); one handle = one owning thread.
 * Integer node state produced by the two libraries for the same config+seed is bit-identical.
 */
#ifndef SWIMSIM_H
#define SWIMSIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWIM_ABI_VERSION 7u

/* ---- status codes -------------------------------------------------------------------- */
#define SWIM_OK          0
#define SWIM_EINVAL    (-22)  /* bad argument / inconsistent config                         */
#define SWIM_ENOMEM    (-12)  /* host or device allocation failed                           */
#define SWIM_ENODEV    (-19)  /* no usable HIP device (product library only)                */
#define SWIM_ERANGE    (-34)  /* id / replica / buffer capacity out of range                */
#define SWIM_EOVERFLOW (-75)  /* a bounded structure overflowed; results are not trustworthy */
#define SWIM_ESTATE    (-71)  /* call not legal in the current tick phase                   */
#define SWIM_EIO       (-5)   /* a checkpoint file could not be written or read             */

#define SWIM_NONE 0xFFFFFFFFu
#define SWIM_SUBJECT_PULL 0xFFFFFFFEu  /* edge.subject of a push-pull request; edge.incarnation = requester */
#define SWIM_SUBJECT_PIGGY 0xFFFFFFFDu /* edge.subject of a piggy-back order (SWIM_F_PIGGYBACK): dst = the node whose
                                          ping/ack/indirect-ping/nack carries broadcasts, edge.incarnation = the
                                          packet's receiver (SWIM_NONE: the packet is lost), meta = carrier<<30 | prober */
#define SWIM_CTL_PING 0u
#define SWIM_CTL_INDIRECT 1u
#define SWIM_CTL_ACK 2u
#define SWIM_CTL_NACK 3u

/* ---- enums pinned by the reference ------------------------------------------------------ */
/* memberlist NodeStateType (state.go) */
enum { SWIM_STATE_ALIVE = 0, SWIM_STATE_SUSPECT = 1, SWIM_STATE_DEAD = 2, SWIM_STATE_LEFT = 3 };
/* rumour kinds carried in gossip packets (memberlist aliveMsg/suspectMsg/deadMsg, serf
 * messageUserEventType) */
enum { SWIM_MSG_ALIVE = 0, SWIM_MSG_SUSPECT = 1, SWIM_MSG_DEAD = 2, SWIM_MSG_USER = 3 };
/* serf.MemberStatus — pinned in-tree at api/agent.go:296-304 */
enum { SWIM_MEMBER_NONE = 0, SWIM_MEMBER_ALIVE = 1, SWIM_MEMBER_LEAVING = 2,
       SWIM_MEMBER_LEFT = 3, SWIM_MEMBER_FAILED = 4 };
/* serf.EventType as consumed by lanEventHandler (agent/consul/server_serf.go:270-297,
 * client_serf.go:80-110) */
enum { SWIM_EVENT_MEMBER_JOIN = 0, SWIM_EVENT_MEMBER_LEAVE = 1, SWIM_EVENT_MEMBER_FAILED = 2,
       SWIM_EVENT_MEMBER_UPDATE = 3, SWIM_EVENT_MEMBER_REAP = 4, SWIM_EVENT_USER = 5,
       SWIM_EVENT_QUERY = 6 };
/* presets = memberlist.DefaultLANConfig / DefaultWANConfig / DefaultLocalConfig
 * (config.go upstream; the six Consul-exposed knobs are corroborated in-tree at
 * agent/config/runtime.go:1285-1427) */
enum { SWIM_PRESET_LAN = 0, SWIM_PRESET_WAN = 1, SWIM_PRESET_LOCAL = 2 };
/* (SWIM_PRESET_WAN is memberlist's DefaultWANConfig: GossipNodes = 4.  Consul's own WAN pool overrides it with the LAN
 * value 3, agent/config/default.go:88 — set gossip_nodes = 3 for that.) */

/* config flags */
#define SWIM_F_BUDDY_SUSPECT  0x1u /* probeNode: ping+suspect compound to a non-alive target  */
#define SWIM_F_NACK           0x2u /* Lifeguard nack accounting on indirect probes            */
#define SWIM_F_SERF_EVENTS    0x4u /* allocate the per-node Serf Lamport/event-buffer state   */
/* Drop, at the sender, a rumour that provably cannot change its receiver (same shard only): the
 * receiver already holds a newer incarnation, or the same incarnation in a state the message cannot
 * move.  Node state is identical with the flag on or off; only `edges` differ. */
#define SWIM_F_FILTER_NOOP    0x8u
/* memberlist sendMsg (net.go): every ping / indirect ping / ack / nack also carries getBroadcasts() of its
 * sender in the bytes the packet has left.  The sender picks the broadcasts at the end of the
 * tick the carrier goes out (before it merges that tick's arrivals); they arrive one tick later. */
#define SWIM_F_PIGGYBACK      0x10u
/* probeNode's TCP fallback (memberlist Config.DisableTcpPings = false, the default; Consul switches it off per node only
 * for WAN federation over mesh gateways, agent/consul/server_serf.go:222-231): when the direct UDP ping got no ack,
 * a TCP ping goes out next to the indirect probes.  TCP rides out packet loss, so it reaches every running node of
 * the same partition; contact counts as a successful probe. */
#define SWIM_F_TCP_FALLBACK   0x20u
/* serf/coordinate (Vivaldi network coordinates): every node keeps a coordinate.Coordinate and updates it
 * on each DIRECT probe ack with the acker's coordinate and the measured round-trip time (serf ping_delegate.go
 * NotifyPingComplete -> coordinate.Client.Update).  Round-trip times come from the latency model below (rtt_*).
 * Unsharded handles only (an ack from another shard would have to carry the coordinate).  Not in SWIM_F_DEFAULT. */
#define SWIM_F_COORDINATES    0x40u
/* memberlist's TransmitLimitedQueue as it is upstream: UNBOUNDED (queue.go never drops a broadcast before its retransmit limit; Consul
 * sizes only serf's event queue, internal/gossip/libserf/serf.go:24-27).  Without the flag a node's queue holds queue_cap entries with
 * Prune() semantics (counted in queue_drops) — which decides BASELINE config #4's answer.  With it:
 *   - the checker's queues grow on demand;
 *   - the product library keeps the rumour about a subject that owns a row of the dense pair store (mass_rows; every subject named in a
 *     stimulus call) IN THE PAIR: 8 more bytes per (row, observer) hold {queued, transmits, type, sequence number, accuser, incarnation},
 *     QueueBroadcast is a store into the pair, its invalidation is implied (one rumour per subject and node), and GetBroadcasts selects
 *     over the node's column by (transmits asc, length desc, sequence desc) exactly as queue.go orders its btree.  A node's rumour about
 *     ITSELF and rumours about subjects without a row stay in the queue_cap slots (Prune() there is still counted, never silent);
 *   - both: a subject is not folded into the base row while a node of the shard still holds a queued rumour about it.
 * Needs mass_rows > 0 and gossip_nodes <= 4 on the product library.  Not in SWIM_F_DEFAULT. */
#define SWIM_F_UNBOUNDED_QUEUE 0x80u
#define SWIM_F_DEFAULT        (SWIM_F_BUDDY_SUSPECT | SWIM_F_NACK | SWIM_F_FILTER_NOOP | SWIM_F_PIGGYBACK | SWIM_F_TCP_FALLBACK)

/* ---- configuration ---------------------------------------------------------------------- */
/* One POD mirroring memberlist.Config field names (the ones CloneSerfLANConfig copies,
 * agent/consul/config.go:679-716, and agent/agent.go:1410-1446 sets) plus simulator bounds. */
typedef struct swim_config {
  uint32_t abi_version;             /* SWIM_ABI_VERSION                                     */
  uint32_t n_nodes;                 /* N: virtual nodes per cluster replica                 */
  uint32_t n_replicas;              /* R: independent clusters, replica r uses seed+r       */
  uint32_t n_initial;               /* members at t = 0: ids [0, n_initial) run and know each other; the rest of the id
                                       space has not been started (nobody has heard of them) until swim_inject_join.
                                       0 = all n_nodes.  With n_initial < n_nodes every node's estNumNodes() is its own
                                       count of known nodes and feeds retransmitLimit / suspicionTimeout like upstream */
  /* memberlist.Config */
  uint32_t gossip_nodes;            /* GossipNodes        agent/agent.go:1419               */
  uint32_t gossip_interval_ms;      /* GossipInterval     agent/agent.go:1418               */
  uint32_t probe_interval_ms;       /* ProbeInterval      agent/agent.go:1420               */
  uint32_t probe_timeout_ms;        /* ProbeTimeout       agent/agent.go:1426               */
  uint32_t suspicion_mult;          /* SuspicionMult      agent/agent.go:1427               */
  uint32_t retransmit_mult;         /* RetransmitMult     agent/agent.go:1428               */
  uint32_t indirect_checks;         /* IndirectChecks (default 3)                           */
  uint32_t suspicion_max_timeout_mult; /* SuspicionMaxTimeoutMult (default 6)               */
  uint32_t awareness_max_mult;      /* AwarenessMaxMultiplier (default 8)                   */
  uint32_t gossip_to_dead_ms;       /* GossipToTheDeadTime                                  */
  uint32_t udp_buffer_size;         /* UDPBufferSize (default 1400)                         */
  uint32_t push_pull_interval_ms;   /* PushPullInterval (LAN 30 s, WAN 60 s; 0 = off); scaled by
                                       pushPullScale(N) like memberlist                      */
  /* modelled encoded sizes of alive/suspect/dead/user messages, bytes (queue order uses len) */
  uint32_t msg_len[4];
  /* modelled encoded sizes of the ping / indirect ping / ack / nack that carry piggy-backed broadcasts */
  uint32_t ctl_len[4];
  /* simulator */
  uint32_t quantum_ms;              /* tick length; 0 = gcd(gossip, probe, timeout)         */
  uint32_t phase_chunk;             /* nodes per stagger chunk (power of 2); 0 = auto       */
  uint32_t queue_cap;               /* per-node TransmitLimitedQueue slots (<= 32)          */
  uint32_t inbox_cap;               /* per-node per-tick inbox slots (from 4 096 on the product library pools the overflow rows: a node's own
                                       row holds 1 024 messages, the few nodes a tick hands more borrow one of up to 8 192 rows of inbox_cap) */
  uint32_t subject_cap;             /* per-replica WATCH slots: subjects whose census, first-suspect/first-dead
                                       stamps and per-tick trace are maintained (swim_watch; every node named in
                                       an inject_* call is watched automatically while slots remain); < 32 767 */
  uint32_t view_cap;                /* per-observer bound on explicit (non-base) views: what an observer knows
                                       that the replica's base row does not say.  A rumour that would need one
                                       more entry is ignored and counted in view_drops (never silent).  Sized
                                       like Serf's queue rule max(2N, 4096) at small N
                                       (internal/gossip/libserf/serf.go:25-27); 0 = min(n_nodes, 32)            */
  uint32_t mass_rows;               /* per replica: rows of the DENSE pair store for mass events (BASELINE configs #4 / #5: thousands
                                       of subjects every observer hears about).  A node named in a stimulus call (kill, revive,
                                       leave, update, join, the minority sides of a partition) gets a row while rows remain, and
                                       every observer's view of it then costs 12 bytes in [row][observer] planes instead of a
                                       64-byte hash-table entry counted against view_cap: memory = 12 B x mass_rows x nodes on the
                                       shard x replicas (20 B with SWIM_F_UNBOUNDED_QUEUE: the pair also holds the rumour queued about the subject).  Representation only: no result depends on which subject has a row.
                                       Needs SuspicionMult <= 4 (two accuser names per pair), a fixed population (n_initial = 0),
                                       n_nodes <= 2^22.  The checker ignores the field.  0 = off              */
  /* serf's reaper (handleReap): every ReapInterval a member that has been Failed for longer than ReconnectTimeout, or Left
   * for longer than TombstoneTimeout, is erased from the observer's member list (status NONE) and EventMemberReap is
   * emitted (agent/consul/server_serf.go:279, timeouts agent/consul/config.go:640-641; the reference's tests run it at
   * 250-300 ms: agent/consul/server_test.go:673-678).  0 = the reaper is off. */
  uint32_t reap_interval_ms, reconnect_timeout_ms, tombstone_timeout_ms;
  uint32_t reconnect_interval_ms;   /* serf ReconnectInterval (serf default 30 s; Consul leaves it): every interval a node picks one
                                       member it holds Failed — with probability failed/alive, like serf's reconnect() — and
                                       tries to rejoin it: memberlist.Join([addr]) = a state exchange with that member, which
                                       succeeds when it is running and in reach.  This is what heals a partition that lasted
                                       longer than GossipToTheDeadTime: by then nobody gossips to, probes or push-pulls with
                                       the other side any more.  Members erased by the reaper are not tried (serf: until
                                       ReconnectTimeout, agent/consul/config.go:640).  0 = off                          */
  uint32_t fold_interval_ms;        /* every so often a subject on which ALL acting observers agree (same
                                       incarnation and state, not Suspect, Dead for longer than
                                       GossipToTheDeadTime) is folded into the base row and its entries are
                                       freed (SURVEY §7 hard part 1, App. D k_reap_fold); 0 = never            */
  uint32_t event_queue_cap;         /* per-node serf user-event queue slots (<= 32; the oracle holds up to 8192: serf's max(2N, 4096)) */
  uint32_t event_buffer;            /* serf EventBuffer ring size (default 512)             */
  uint32_t event_ids_per_ltime;     /* distinct user events (and intents) a node remembers per Lamport time: serf's slot is an
                                       unbounded list, and a flood stamps many events alike; one more with the same LTime is
                                       dropped (event_drops).  Rounded up to 4k + 2; 0 = 14 (a 64-byte slot)              */
  uint32_t loss_q32;                /* packet loss prob * 2^32 (0 = lossless)               */
  uint32_t flags;                   /* SWIM_F_*                                             */
  uint32_t watch_node;              /* observer whose serf events are recorded (SWIM_NONE=off)*/
  uint32_t trace_ticks;             /* per-tick census history capacity (0 = off)           */
  uint32_t shard_rank, n_shards;    /* block partition of every replica's node ids          */
  uint32_t device;                  /* HIP device ordinal (product library)                 */
  /* SWIM_F_COORDINATES — the latency model the probes measure (the simulator has no wires): node i sits at a hidden point
   * of a cube with edge rtt_scale_us (three Philox words of stream "truth", in microseconds of round-trip time) behind an
   * access link of rtt_height_us * u (a fourth word); rtt(i, j) = floor(|p_i - p_j|) + h_i + h_j + jitter, jitter uniform in
   * [0, rtt_jitter_us) per probe.  Heights are what Vivaldi's height term is for.  Presets: 40 000 / 2 000 / 0. */
  uint32_t rtt_scale_us, rtt_height_us, rtt_jitter_us;
  uint64_t seed;
} swim_config;

/* coordinate.Coordinate (serf/coordinate/coordinate.go), Dimensionality = 8 (coordinate.DefaultConfig; Consul never changes
 * it).  Seconds. */
#define SWIM_COORD_DIMS 8
typedef struct swim_coordinate {
  double vec[SWIM_COORD_DIMS];
  double error, adjustment, height;
} swim_coordinate;

/* closed-form constants of memberlist util.go / suspicion.go */
typedef struct swim_derived {
  uint32_t quantum_ms, gossip_period, probe_period, probe_timeout_ticks;
  uint32_t phase_chunk;
  uint32_t retransmit_limit;        /* retransmitLimit(RetransmitMult, N)                   */
  uint32_t suspicion_k;             /* SuspicionMult-2, or 0 when N-2 < k                   */
  uint32_t suspicion_min_ms, suspicion_max_ms;
  uint32_t suspicion_timeout_ms[8]; /* timeout after n = 0..k confirmations                 */
  uint32_t node_scale_milli;        /* int(max(1,log10(max(1,N))) * 1000)                   */
  uint32_t push_pull_scale;         /* pushPullScale multiplier                             */
  uint32_t push_pull_period_ticks;  /* PushPullInterval * scale / quantum (0 = off)         */
  uint32_t packet_budget;           /* UDPBufferSize - compoundHeaderOverhead               */
  uint32_t view_cap;                /* resolved swim_config.view_cap                         */
  uint32_t fold_period_ticks;       /* fold_interval_ms / quantum, rounded up (0 = off)      */
  uint32_t reap_period_ticks;       /* reap_interval_ms / quantum, rounded up (0 = off)      */
  uint32_t reconnect_period_ticks;  /* reconnect_interval_ms / quantum, rounded up (0 = off) */
} swim_derived;

/* one row of an observer's member list: serf.Member / memberlist.Node reduced to integers
 * (name, address and tags are host-side, keyed by id) — api/agent.go:291-311 */
typedef struct swim_member {
  uint32_t id;
  uint32_t incarnation;
  uint32_t state_change_ms;
  uint8_t  state;                   /* SWIM_STATE_*                                         */
  uint8_t  status;                  /* SWIM_MEMBER_* (serf view of the same row)            */
  uint8_t  n_confirm;               /* suspicion confirmations seen (Suspect only)          */
  uint8_t  _pad;
} swim_member;

/* serf.Event as delivered on EventCh (agent/consul/server.go:112-114,519-520) */
typedef struct swim_event {
  uint32_t time_ms;
  uint32_t replica;
  uint32_t type;                    /* SWIM_EVENT_*                                         */
  uint32_t node;                    /* member id, or user-event id                          */
  uint32_t incarnation;
  uint32_t observer;                /* whose EventCh this is: cfg.watch_node or a node added with swim_watch_events */
  uint64_t ltime;                   /* user events: Lamport time (serf.LamportTime is 64 bits, and so is it here since ABI v7; the
                                       simulated clocks themselves are 31 bits wide: one tick per event, 2^31 events per run)  */
} swim_event;

/* one entry of a node's TransmitLimitedQueue */
typedef struct swim_rumour {
  uint32_t subject, incarnation, from;
  uint8_t  type, transmits, _pad[2];
  uint32_t seq;
} swim_rumour;

/* per-node self state (memberlist: incarnation, awareness score, probe bookkeeping) */
typedef struct swim_node_info {
  uint32_t incarnation;
  uint32_t probe_target, probe_deadline_tick, probe_cursor, probe_epoch;
  uint32_t queue_len, event_queue_len;
  uint8_t  alive, leaving, awareness, partition;
  uint64_t event_clock;             /* serf's event LamportClock.Time() (64 bits in the ABI since v7)                        */
  swim_rumour queue[32];
} swim_node_info;

/* how the live observers of one replica currently see one subject */
typedef struct swim_census {
  uint32_t n_observers;             /* live observers other than the subject                */
  uint32_t by_state[4];             /* their view of the subject, by SWIM_STATE_*           */
  uint32_t n_current;               /* views whose incarnation == subject's own incarnation */
  uint32_t first_suspect_ms, first_dead_ms, all_dead_ms; /* SWIM_NONE until it happened     */
  uint32_t all_current_ms;          /* first time every live observer held the current inc  */
} swim_census;

/* one directed rumour delivery (gossip edge): 16 bytes, the unit of the all-to-all */
typedef struct swim_edge {
  uint32_t dst;                     /* replica*N + node, or SWIM_NONE for a control record  */
  uint32_t subject;
  uint32_t incarnation;             /* user events: ltime                                   */
  uint32_t meta;                    /* type<<30 | from                                      */
} swim_edge;

typedef struct swim_stats_t {
  uint64_t ticks, gossip_rounds;
  uint64_t node_rounds_active, node_rounds_quiescent; /* gossip-due live nodes w/ and w/o queue */
  uint64_t packets_sent, packets_dropped;
  uint64_t msgs_sent[4];
  uint64_t msgs_applied[4];         /* handleAlive/Suspect/Dead/User that changed state     */
  uint64_t probes, probe_acks, probe_indirect_acks, probe_failures, nacks_missed;
  uint64_t refutes, suspicion_timeouts, confirmations;
  uint64_t edges, edges_remote;
  uint64_t queue_drops, inbox_overflow, subject_overflow, event_drops;
  uint64_t user_events_delivered, user_events_deduped, user_events_stale;
  uint64_t msgs_filtered;           /* rumours dropped at the sender by SWIM_F_FILTER_NOOP       */
  uint64_t push_pulls;              /* pushPullNode exchanges initiated                          */
  uint64_t piggybacks;              /* pings/acks/... that carried at least one broadcast        */
  uint64_t msgs_piggybacked;        /* broadcasts carried that way (also counted in msgs_sent)   */
  uint64_t probe_tcp_acks;          /* probes saved by the TCP fallback ping (SWIM_F_TCP_FALLBACK)*/
  uint64_t view_drops;              /* rumours ignored because the observer already held view_cap explicit views */
  uint64_t view_evictions;          /* long-settled Dead/Left views a full table forgot to make room (memberlist
                                       resetNodes forgets a node dead for longer than GossipToTheDeadTime)         */
  uint64_t intents_applied;         /* leave intents that turned a Failed member Left (swim_force_leave)          */
  uint64_t reaped;                  /* (observer, member) pairs erased by the reaper or a prune                    */
  uint64_t joins;                   /* join push-pulls carried out (swim_inject_join); a join whose `via` cannot be reached
                                       is counted in join_failures (memberlist.Join returns an error)              */
  uint64_t join_failures;
  uint64_t folds;                   /* subjects folded into the base row (counted by the shard owning the id)   */
  uint64_t fold_freed;              /* explicit view entries freed by folding                                   */
  uint64_t coord_updates;           /* coordinate.Client.Update calls (one per direct probe ack, SWIM_F_COORDINATES) */
  uint64_t coord_resets;            /* ... that left an invalid coordinate and reset it (Client.stats.Resets)     */
  uint64_t reconnects;              /* serf reconnect(): attempts (a node that passed the failed/alive gate and picked a member) */
  uint64_t reconnects_reached;      /* ... whose member was running and in reach: a state exchange went out        */
  uint64_t inbox_peak;              /* the largest number of messages one node received in one tick, counted from six on (five
                                       fit the node's inbox line; 0 = never more).  Sizes inbox_cap: a state exchange delivers
                                       a whole table at once                                                        */
} swim_stats_t;

typedef struct swim_sim swim_sim;

/* ---- lifecycle ---------------------------------------------------------------------------- */
/* memberlist.DefaultLANConfig()/DefaultWANConfig()/DefaultLocalConfig(); Consul's use:
 * agent/consul/config.go:553-555,645 */
int swim_config_preset(swim_config* cfg, int preset);
/* util.go retransmitLimit/suspicionTimeout/pushPullScale + suspicion.go remainingSuspicionTime,
 * evaluated once on the host with Go's float64 semantics (docs: agent/config/runtime.go:1326,1344) */
int swim_config_derive(const swim_config* cfg, swim_derived* out);
/* serf.Create -> memberlist.Create -> newMemberlist + setAlive + schedule, for all N*R virtual
 * nodes at once (agent/consul/server_serf.go:63, client_serf.go:76).  All nodes start alive at
 * incarnation 1 with converged views (BASELINE config #2 "all alive at t=0"). */
int swim_create(const swim_config* cfg, swim_sim** out);
/* serf.Shutdown (agent/consul/client.go:188, server.go:1277) */
int swim_destroy(swim_sim* sim);
const char* swim_backend(void);      /* "hip-gfx950" or "oracle-c" */
const char* swim_last_error(swim_sim* sim);

/* ---- time ----------------------------------------------------------------------------------- */
/* memberlist.schedule's three tickers, advanced n_ticks quanta for every virtual node:
 * probe()/probeNode, gossip(), suspicion timers, packet delivery, handleAlive/Suspect/Dead
 * (SURVEY §3.2, §3.3).  Asynchronous on the product library; swim_sync waits. */
int swim_step(swim_sim* sim, uint32_t n_ticks);
int swim_sync(swim_sim* sim);
int swim_now(swim_sim* sim, uint32_t* tick, uint32_t* now_ms);

/* split tick for a population sharded over several devices:
 *   begin    = timers + probe + gossip select/emit -> outbound segments bucketed by shard
 *   outbound = device pointer + record count of the segment for `shard`
 *   inbound  = hand over records received from another shard (device pointer on the product
 *              library, host pointer on the oracle).  The product library copies asynchronously on
 *              its stream: the buffer must stay untouched until swim_tick_end has been called and the
 *              next swim_outbound / swim_sync returned
 *   end      = subject-slot allocation, delivery, merge (aliveNode/suspectNode/deadNode)     */
int swim_tick_begin(swim_sim* sim);
int swim_outbound(swim_sim* sim, uint32_t shard, const swim_edge** ptr, uint32_t* count);
uint32_t swim_outbound_capacity(swim_sim* sim, uint32_t shard);   /* records the segment can ever hold */
/* For a device-driven exchange (no host round trip for the counts): the stream every kernel of this
 * simulator runs on, the address of the n_shards record counters (uint32 each, valid after
 * swim_tick_begin in stream order) and of a shard's segment.  NULL/0 on the oracle. */
int swim_stream(swim_sim* sim, void** hip_stream);
int swim_outbound_raw(swim_sim* sim, uint32_t shard, const swim_edge** segment, const uint32_t** counters);
int swim_inbound(swim_sim* sim, const swim_edge* ptr, uint32_t count);
int swim_tick_end(swim_sim* sim);
/* swim_tick_end immediately followed by the next swim_tick_begin, as one call (the product library replays the
 * whole launch sequence from a captured graph when nothing was handed in with swim_inbound) */
int swim_tick_end_begin(swim_sim* sim);
/* SWIM_F_PIGGYBACK across shards: a probe files a piggy-back order for its target's ack, and the target may live
 * on another shard.  While no node anywhere has anything queued such orders are no-ops; `active` = 0 tells the
 * next swim_tick_begin that the caller knows this to be so for every OTHER shard, and the orders stay unfiled
 * (the quiescent tick then crosses no wire).  How to know: the word behind the n_shards counters of
 * swim_outbound_raw is non-zero when this shard may hold a non-empty queue or emitted anything this tick; if that
 * word and all counters of all shards are zero in tick t, `active` may be 0 for tick t+1.  Default 1 (always
 * correct).  Results never depend on the hint.  swim_activity is the host-side read of the same condition for
 * this shard (between swim_tick_begin and swim_tick_end; the oracle always answers 1). */
int swim_peer_activity(swim_sim* sim, int active);
int swim_activity(swim_sim* sim, int* active);

/* ---- framed exchange: the split tick for a COLLECTIVE with host-known sizes and no host round trip -----------------------
 * (north_star's "all-to-all over RCCL"; consul_amd/dist.py TorchExchange).  An all-to-all wants its sizes on the host, the
 * record counts of a tick live on the device: instead of reading them back every tick, every shard sends every other shard a
 * FRAME of `frame_records` 16-byte records — record 0 is a header {count, activity word, tick + 1, SWIM_FRAME_MAGIC} written
 * by the device, records 1..count the segment for that shard — with one equal-split all_to_all_single, and the receiving
 * device reads the counts out of the headers:
 *   swim_tick_begin -> swim_frame_pack(send) -> all_to_all_single(recv, send) -> swim_frame_deliver(recv) -> swim_tick_end[_begin]
 * send / recv: n_shards * frame_records records, frame j of `send` for shard j, frame i of `recv` from shard i (the own frame
 * is empty and ignored); device memory on the product library (both calls are asynchronous on the simulator's stream: issue
 * the collective on swim_stream), host memory on the oracle.  The activity word of swim_peer_activity travels in the headers
 * and is applied by swim_frame_deliver.  A segment that does not fit frame_records - 1 records raises the sticky overflow
 * error of the edge lists (SWIM_EOVERFLOW at the next swim_sync); a frame whose header is not this tick's, SWIM_ESTATE.
 * swim_frame_records = the smallest frame that can never overflow (1 + the largest outbound capacity; every shard of a
 * population answers the same number); 0 on the oracle, whose lists are unbounded — pick any size there.  A population of ONE
 * shard may make the same calls (its only frame is its own, empty): the plumbing of a collective can be checked on one device. */
#define SWIM_FRAME_MAGIC 0x4D415246u   /* "FRAM" */
uint32_t swim_frame_records(swim_sim* sim);
int swim_frame_pack(swim_sim* sim, swim_edge* send, uint32_t frame_records);
int swim_frame_deliver(swim_sim* sim, const swim_edge* recv, uint32_t frame_records);
/* Frames sized from the LOAD, not from the bound (round 5; consul_amd/dist.py TorchExchange): swim_frame_pack_fill packs like
 * swim_frame_pack, but a segment that does not fit is not an error — the frame carries what fits and its header says what there is:
 *   {count = records the segment HAS, activity | need << 1, tick + 1, SWIM_FRAME_MAGIC},  need = the largest count this shard holds for
 *   ANY destination this tick.
 * The caller looks at the headers it received (and its own): when the largest `need` of the population exceeds frame_records - 1 —
 * every shard sees every sender's, so all take the same decision — it packs again into frames of at least need + 1 records (packing
 * is repeatable until swim_tick_end) and repeats the collective, BEFORE swim_frame_deliver: nothing is ever lost, and a quiet tick
 * moves a frame of a few records instead of the bound's megabytes.  swim_frame_deliver refuses a frame whose count exceeds
 * frame_records - 1 (the sticky edge-list overflow on the product library, SWIM_ESTATE on the oracle). */
int swim_frame_pack_fill(swim_sim* sim, swim_edge* send, uint32_t frame_records);

/* ---- device-driven exchange between the shards of one population (peer-mapped mailboxes over xGMI) -------
 * The split tick above leaves the exchange to the caller (consul_amd/dist.py uses RCCL).  This is the library's own: every
 * shard owns a mailbox in its HBM (one area per source shard, double buffered by tick parity), exports it as an IPC handle,
 * maps everybody else's, and from then on a tick is: begin -> copy my per-destination segments into the destinations'
 * mailboxes (stores over xGMI, or through the shared L2 when two shards share a device) -> release one flag per destination
 * -> wait for the flags of my sources -> deliver -> end.  No host round trip, no collective; counts and the activity word
 * travel in the mailbox header.  One shard per process (or several handles in one process); the handles are plain bytes
 * and can be passed over any channel (a pipe, a file, an all-gather).  The oracle implements the same calls on host memory
 * for shards that live in ONE process (its handle is a pointer), so the CPU suite covers the protocol. */
#define SWIM_XCHG_HANDLE_BYTES 96
typedef struct swim_xchg_handle { uint8_t bytes[SWIM_XCHG_HANDLE_BYTES]; } swim_xchg_handle;
int swim_xchg_export(swim_sim* sim, swim_xchg_handle* out);
/* all[n_shards], indexed by shard rank (the own entry is ignored) */
int swim_xchg_connect(swim_sim* sim, const swim_xchg_handle* all);
/* swim_step for a connected shard: n_ticks whole ticks including the exchange.  Asynchronous on the product library
 * (swim_sync waits and reports a peer that did not show up within the time-out as SWIM_ESTATE).  Every shard of the
 * population must make the same calls. */
int swim_xchg_step(swim_sim* sim, uint32_t n_ticks);

/* ---- stimulus (fault injection is native; the reference kills nodes with Shutdown(),
 *      agent/consul/server_test.go:725) ----------------------------------------------------- */
int swim_inject_kill(swim_sim* sim, uint32_t replica, const uint32_t* ids, size_t n);
int swim_inject_revive(swim_sim* sim, uint32_t replica, const uint32_t* ids, size_t n);
/* serf.Leave -> memberlist.Leave: dead{Node==From} => StateLeft (agent/consul/client.go:205) */
int swim_inject_leave(swim_sim* sim, uint32_t replica, const uint32_t* ids, size_t n);
/* memberlist.UpdateNode (serf.SetTags, internal/gossip/libserf/serf.go:51): bump own
 * incarnation and broadcast alive — the "single rumour" of BASELINE config #3 */
int swim_inject_update(swim_sim* sim, uint32_t replica, const uint32_t* ids, size_t n);
/* serf.Create + serf.Join([via]) for nodes that are not running (agent/consul/client.go:222, server.go:1461,
 * agent/router/serf_flooder.go:81): a fresh process — empty queues, no views of its own, incarnation 1 on its first
 * start and the previous one + 1 after a restart — that queues alive{self} (memberlist setAlive) and, in its first
 * tick, does the join push-pull with `via` (pushPullNode(join=true): its state to via, via's back one tick later).
 * Everybody who hears the alive{} takes the aliveNode path for a node it has never heard of (NotifyJoin). */
int swim_inject_join(swim_sim* sim, uint32_t replica, const uint32_t* ids, size_t n, uint32_t via);
/* serf.RemoveFailedNode / RemoveFailedNodePrune (agent/consul/client.go:272-274; `consul force-leave [-prune]`,
 * agent/agent_endpoint_test.go:2524-2566, 2633-2677): `origin` broadcasts a Lamport-clocked leave intent on behalf of
 * `node`; every member that holds `node` Failed turns it Left (EventMemberLeave), with prune also erases it at once
 * (EventMemberReap).  Needs SWIM_F_SERF_EVENTS (the intent rides serf's broadcast queue and event-buffer dedupe).
 * The Lamport time stamped on the intent is returned like swim_user_event's. */
int swim_force_leave(swim_sim* sim, uint32_t replica, uint32_t origin, uint32_t node, int prune, uint64_t* ltime_out);
#define SWIM_INTENT_LEAVE 0x80000000u   /* event id of a leave intent: SWIM_INTENT_LEAVE | prune << 30 | node */
#define SWIM_INTENT_PRUNE 0x40000000u
#define SWIM_INTENT_JOIN 0xA0000000u    /* event id of a join intent (serf messageJoinType): SWIM_INTENT_JOIN | node — what serf.Join broadcasts after the
                                           join push-pull, and what a member answers a leave intent about ITSELF with while it is not leaving (the refutation) */
/* partition mask: nodes exchange packets only within the same group id (config #4) */
int swim_inject_partition(swim_sim* sim, uint32_t replica, const uint8_t* group_of_node);
int swim_set_loss(swim_sim* sim, uint32_t loss_q32);
/* memberlist.Config.DisableTcpPingsForNode (hashicorp/memberlist config.go; set by agent/consul/server_serf.go:222-232 on the
 * WAN pool under mesh-gateway federation: `return s.config.Datacenter != dc`): probeNode skips its TCP fallback ping when the
 * predicate holds for the target.  Here every node carries a class (Consul: its datacenter; 0 at creation) and the fallback ping
 * is skipped between nodes of DIFFERENT classes — all nodes in class 0 is memberlist's default, no predicate.  Global
 * DisableTcpPings is the absence of SWIM_F_TCP_FALLBACK.  Classes 0..SWIM_TCP_CLASS_MAX; more is SWIM_ERANGE. */
#define SWIM_TCP_CLASS_MAX 15u
int swim_set_tcp_class(swim_sim* sim, uint32_t replica, const uint32_t* ids, size_t n, uint8_t tcp_class);
/* serf.UserEvent(name, payload, coalesce=false) (agent/consul/server_ce.go:125-131):
 * event_id stands for hash(name,payload); returns the Lamport time stamped on it */
int swim_user_event(swim_sim* sim, uint32_t replica, uint32_t origin, uint32_t event_id,
                    uint64_t* ltime_out);
/* user-event ids are 30 bits: bits 31-30 of the id word distinguish serf's intent messages (messageLeaveType; SWIM_INTENT_*)
 * from user events on the shared broadcast queue.  A larger id is refused (SWIM_ERANGE) — never reinterpreted as an intent. */
#define SWIM_EVENT_ID_MAX 0x3FFFFFFFu

/* ---- observation ---------------------------------------------------------------------------- */
/* serf.Members() as seen by `observer` (agent/consul/client.go:234, server.go:1508): writes
 * min(cap, N) rows ordered by id; *n_out = N */
int swim_members(swim_sim* sim, uint32_t replica, uint32_t observer, swim_member* out,
                 size_t cap, size_t* n_out);
/* one row of the above */
int swim_view(swim_sim* sim, uint32_t replica, uint32_t observer, uint32_t subject,
              swim_member* out);
/* Track `subject` in a watch slot from now on (census with first-* stamps, per-tick trace).  SWIM_OK if it is or was
 * already watched; SWIM_EOVERFLOW (counted in subject_overflow) when all subject_cap slots are taken — the simulation
 * itself never depends on a watch slot, and swim_census_get still answers for an unwatched subject (counted on
 * demand, first-* stamps SWIM_NONE). */
int swim_watch(swim_sim* sim, uint32_t replica, uint32_t subject);
/* serf.Config.EventCh drained by lanEventHandler (server_serf.go:270): events seen by
 * cfg.watch_node of every replica, oldest first */
int swim_poll_events(swim_sim* sim, swim_event* out, size_t cap, size_t* n_out);
/* One EventCh per agent (agent/consul/server.go:112-114, client.go:51-59): from now on the serf events of `node` are recorded
 * too, tick-exact like cfg.watch_node's (swim_event.observer says whose).  At most SWIM_EVENT_WATCHERS per replica
 * (SWIM_EOVERFLOW beyond); the node must live on this shard (SWIM_ERANGE). */
#define SWIM_EVENT_WATCHERS 64
int swim_watch_events(swim_sim* sim, uint32_t replica, uint32_t node);
int swim_node_info_get(swim_sim* sim, uint32_t replica, uint32_t node, swim_node_info* out);
/* serf's notifyCh of a broadcast, as a question: is the serf broadcast {id, ltime} (a user event or an intent: swim_user_event / swim_force_leave
 * return the ltime) still in `node`'s queue — not yet out of transmissions, not pruned?  What Serf.Leave() waits on (include/swimsim_serf.hpp). */
int swim_event_queued(swim_sim* sim, uint32_t replica, uint32_t node, uint32_t id, uint64_t ltime, int* queued);
int swim_census_get(swim_sim* sim, uint32_t replica, uint32_t subject, swim_census* out);
/* BASELINE config #4's deliverable ("rounds until all survivors mark all victims dead"), for any mix of stopped and
 * partitioned nodes: over all ordered pairs (observer o, subject x != o) where o is a node of THIS shard the simulator acts
 * for and x is out of o's reach right now (x is not running, or sits in another partition group), how o sees x.  Detection is
 * complete when by_state[SWIM_STATE_DEAD] + by_state[SWIM_STATE_LEFT] == pairs.  A sharded population sums its shards'. */
typedef struct swim_detection {
  uint64_t pairs;                   /* (observer, unreachable subject) pairs                  */
  uint64_t by_state[4];             /* the observers' views of those subjects, by SWIM_STATE_* */
} swim_detection;
int swim_detection_get(swim_sim* sim, uint32_t replica, swim_detection* out);
/* per-tick census history of `subject` (infection / detection curves): rows for ticks
 * [first_tick, first_tick+n), each {by_state[4], n_current} = 5 x u32 */
int swim_trace_read(swim_sim* sim, uint32_t replica, uint32_t subject, uint32_t first_tick,
                    uint32_t n, uint32_t* out_rows5);
/* serf.Stats() / memberlist metrics (agent/consul/server.go:1749, SURVEY §5 metrics row) */
int swim_stats(swim_sim* sim, swim_stats_t* out);

/* ---- network coordinates (SWIM_F_COORDINATES) ---------------------------------------------- */
/* serf.GetCoordinate() of a virtual node / GetCachedCoordinate(name) as any observer would eventually cache it
 * (RouterSerfCluster, agent/router/router.go:62-67; agent/consul/server_ce.go:117-119) */
int swim_coordinate_get(swim_sim* sim, uint32_t replica, uint32_t node, swim_coordinate* out);
/* librtt.ComputeDistance (internal/gossip/librtt/rtt.go:16-22): a.DistanceTo(b).Seconds() — the raw distance plus both
 * adjustments when that is positive, through time.Duration (truncated to whole nanoseconds); +Inf when either is NULL.
 * Pure host arithmetic, no handle needed. */
double swim_coordinate_distance(const swim_coordinate* a, const swim_coordinate* b);
/* the latency model's round-trip time between two nodes without jitter, microseconds (to judge the coordinates against) */
int swim_rtt_truth(swim_sim* sim, uint32_t replica, uint32_t a, uint32_t b, uint32_t* rtt_us);
/* the edge list of the most recent tick (after emit, before delivery), for parity tests.  HIP library, handles with tile buckets
 * (SWIM_INFO_TILE_BUCKETS): the no-op filter runs at the receivers, and what it dropped is only marked once this call has been made —
 * the first call switches the recording on and reports the rumours of the tick before it unfiltered; callers that compare tick by
 * tick call once before they start. */
int swim_debug_edges(swim_sim* sim, swim_edge* out, size_t cap, size_t* n_out);
/* how this handle is laid out (what = SWIM_INFO_*; unknown keys: SWIM_EINVAL).  Nothing here shows in any result. */
#define SWIM_INFO_TILE_BUCKETS 0u   /* 1: rumours travel through per-tile buckets and are judged by the no-op filter at the receivers (HIP library); 0: at the senders */
#define SWIM_INFO_MAILBOX_KIND 1u   /* memory of the swim_xchg mailbox: 0 none (one shard), 1 fine-grained, 2 uncached, 3 coarse-grained (only on request: SWIMSIM_MAILBOX=coarse) */
#define SWIM_INFO_DEVICE_BYTES 2u   /* device memory the handle holds */
int swim_info(swim_sim* sim, uint32_t what, uint64_t* out);
/* order-independent 64-bit digest over all integer node state (self state, queues, views,
 * suspicion timers, serf clocks) — "checksum of checksums" for full-size parity */
int swim_state_digest(swim_sim* sim, uint64_t* out);

/* ---- checkpoint / resume -------------------------------------------------------
 * The whole population between two ticks — every array of the structure-of-arrays state, the clock, the counters, pending
 * joins and events — into a file, and back into a handle created from the SAME swim_config (compared byte for byte).  A run
 * continued from a checkpoint is the run that was never interrupted: digests, counters, censuses and events agree tick for
 * tick.  Upstream's nearest relative is serf's snapshotter (conf.SnapshotPath, agent/consul/server_serf.go:236-239: ONE
 * node's member list and clocks, replayed at restart so that it can rejoin); a simulator's unit is the population.
 * The file belongs to the library that wrote it (backend string and ABI version in the header): SWIM_EINVAL for a foreign
 * file or another configuration, SWIM_EIO when it cannot be written / is truncated (the handle's state is then undefined),
 * SWIM_ESTATE inside a tick, on a handle with attached transport-bridge nodes or a connected swim_xchg exchange. */
int swim_checkpoint_save(swim_sim* sim, const char* path);
int swim_checkpoint_load(swim_sim* sim, const char* path);

/* ---- per-kernel timing (bench.py's roofline leg; memberlist's own counterpart is the
 *      metrics.MeasureSince("memberlist","gossip"/"probeNode") timers) ---------------------------
 * With profiling on, every kernel launch is bracketed by HIP events on the simulator's stream. */
typedef struct swim_kernel_time {
  char     name[24];
  uint64_t launches;
  double   total_ms;
} swim_kernel_time;
int swim_profile(swim_sim* sim, int enable);
int swim_profile_read(swim_sim* sim, swim_kernel_time* out, size_t cap, size_t* n_out);

/* ---- memberlist.Transport bridge (agent/consul/wanfed/wanfed.go:96-141)
 * A real memberlist node attached as virtual node `attached` exchanges rumours with its
 * virtual peers: write_to = Transport.WriteToAddress, poll = Transport.PacketCh.  Packets are
 * swim_edge records here; include/swimsim_wire.hpp converts them to and from memberlist's packet bytes.
 * Defined for an unsharded population (n_shards == 1; SWIM_ESTATE otherwise).  The first call naming a node
 * attaches it: what it had queued is dropped and the simulator stops acting for it. */
int swim_transport_write_to(swim_sim* sim, uint32_t replica, uint32_t attached,
                            uint32_t virtual_dst, const swim_edge* msgs, size_t n);
int swim_transport_poll(swim_sim* sim, uint32_t replica, uint32_t attached, swim_edge* out,
                        size_t cap, size_t* n_out);

/* ---- known-answer hooks (pure functions; used by tests/ to pin both libraries) ------------ */
void swim_kat_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
uint32_t swim_kat_probe_perm(uint64_t seed, uint32_t n_nodes, uint32_t node, uint32_t epoch,
                             uint32_t index);
int32_t swim_kat_remaining_suspicion_ms(uint32_t n, uint32_t k, uint32_t elapsed_ms,
                                        uint32_t min_ms, uint32_t max_ms);
void swim_kat_phase_of(const swim_config* cfg, uint32_t node, uint32_t* gossip_phase,
                       uint32_t* probe_phase);
/* awareness.go ApplyDelta (clamped to [0, max-1]) and ScaleTimeout (timeout * (score+1)) */
uint32_t swim_kat_awareness_apply(uint32_t awareness_max_mult, uint32_t score, int32_t delta);
uint32_t swim_kat_awareness_scale_ms(uint32_t score, uint32_t timeout_ms);

#ifdef __cplusplus
}
#endif
#endif /* SWIMSIM_H */

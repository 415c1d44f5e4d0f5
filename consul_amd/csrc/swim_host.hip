// swim_host.hip — C-ABI of libswimsim.so (include/swimsim.h) over the gfx950 kernels.
//
// Host responsibilities only: validate the memberlist.Config mirror, evaluate the closed-form
// constants once (util.go / suspicion.go formulas with Go's float64 semantics), lay the state out
// in HBM, and enqueue the per-tick kernel sequence on one HIP stream (replayed from a hipGraph).
// There is no CPU fallback: without a HIP device swim_create returns SWIM_ENODEV.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <unistd.h>

#include "swim_kernels.hip"

#define HIPCK(s, call)                                                                              \
  do {                                                                                              \
    hipError_t e_ = (call);                                                                         \
    if (e_ != hipSuccess) {                                                                         \
      if (s) snprintf((s)->err, sizeof((s)->err), "%s failed: %s", #call, hipGetErrorString(e_));   \
      return e_ == hipErrorOutOfMemory ? SWIM_ENOMEM : SWIM_ENODEV;                                 \
    }                                                                                               \
  } while (0)

// nodes per workgroup of k_resolve (swim_kernels.hip): a node block; 64 nodes on a handle with the dense pair store
static inline uint32_t resolve_tile(const SwDev& D) { return D.M ? SW_RES_MASS_TILE : SW_RTILE * SW_BLOCK; }
enum { PK_BEGIN = 0, PK_DELIVER, PK_RESOLVE, PK_CENSUS, PK_FINISH, PK_GOSSIP_IQ, PK_PIGGY_IQ, PK_COUNT };
static const char* const kKernelNames[PK_COUNT] = { "k_begin", "k_deliver", "k_resolve", "k_census", "k_finish", "k_gossip_iq", "k_piggy_iq" }   /* (k_finish: k_census_finish since round 5 — the recount and the epilogue in one launch; k_census then has no launches of its own) */;
#define SW_GRAPH_TICKS 16
#define SW_GRAPH_TICKS_MID 4      /* a middle tier (round 5): what is left of a call after the 16-tick graphs goes out four ticks at a time, not one */

struct swim_sim {
  swim_config cfg;
  swim_derived d;
  SwDev D;
  BeginPlan plan;
  BeginKernel begin_kernel = nullptr;
  hipStream_t stream = nullptr;
  std::vector<void*> allocs;
  std::vector<size_t> alloc_bytes;     // (swim_checkpoint_*: every device array is state unless listed in `structural`)
  std::vector<void*> structural;       // arrays that hold THIS handle's device pointers or nothing worth keeping: never saved, never overwritten
  uint32_t tick = 0;
  bool in_tick = false;
  bool pristine = true;                // nothing has ever happened to this population (no stimulus of any kind): k_quiet may stand in for whole ticks
  uint64_t ticks_run = 0, rounds_run = 0;
  SwDev* d_D = nullptr;                // the descriptor the kernels read (device copy of D)
  uint32_t* d_last_cnt = nullptr;      // [n_shards] edge counts of the finished tick
  uint32_t* d_scratch = nullptr;       // device scratch (ids / partition mask upload in chunks, ltime, digest, gathers)
  size_t scratch_bytes = 0;
  uint32_t* d_fresh = nullptr;         // [1024] watch slots allocated by the last stimulus call (count, then slot indices)
  void *fold_zero = nullptr, *fold_ones = nullptr; size_t fold_zero_bytes = 0, fold_ones_bytes = 0;   // fold accumulators, by initial value
  // swim_xchg_*: own mailbox, the peers' mailboxes as mapped here, the captured exchange tick
  uint8_t* mailbox = nullptr; size_t mailbox_bytes = 0; uint32_t mail_cap = 0; const char* mailbox_kind = "none";
  std::vector<void*> ipc_opened; bool xchg_connected = false;
  hipGraphExec_t graph_xchg = nullptr;
  uint4* in_buf = nullptr;             // records received from other shards
  uint32_t in_cap = 0, in_count = 0;
  uint32_t out_counts[SW_MAX_SHARDS + 1];  // host copy for swim_outbound / swim_activity (counts, then the activity word)
  uint32_t peer_act_host = 1;          // what the device word D.peer_act currently holds
  hipGraphExec_t graph_end_begin = nullptr;   // sharded runs: [end of tick t, begin of tick t+1] when nothing came in
  bool out_counts_valid = false;
  std::vector<swim_event> pending_events;
  std::vector<uint64_t> attached;                      // (replica << 32 | node) driven through the transport bridge
  struct Captured { uint32_t gdst; swim_edge rec; };   // rec.dst = sender
  std::vector<Captured> captured;
  // captured tick sequence: [0] one tick, [1] SW_GRAPH_TICKS ticks, [2] SW_GRAPH_TICKS_MID ticks
  hipGraphExec_t graph_exec[3] = { nullptr, nullptr, nullptr };
  bool use_graphs = true;
  // optional per-launch HIP-event timing
  bool profiling = false;
  std::vector<hipEvent_t> ev_pool;
  struct ProfRec { int kernel; size_t ev; };
  std::vector<ProfRec> prof_recs;
  size_t ev_used = 0;
  char err[256] = { 0 };
};

// bracket one launch with two events on the simulator's stream
struct ProfScope {
  swim_sim* s; bool on;
  ProfScope(swim_sim* sim, int kernel) : s(sim), on(sim->profiling) {
    if (!on) return;
    while (s->ev_pool.size() < s->ev_used + 2) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) { on = false; return; } s->ev_pool.push_back(e); }
    s->prof_recs.push_back({ kernel, s->ev_used });
    (void)hipEventRecord(s->ev_pool[s->ev_used], s->stream);
  }
  ~ProfScope() { if (on) { (void)hipEventRecord(s->ev_pool[s->ev_used + 1], s->stream); s->ev_used += 2; } }
};

// ---------------------------------------------------------------------------------------------
// closed-form constants (memberlist util.go, suspicion.go) — independent of the oracle's copy
// ---------------------------------------------------------------------------------------------
namespace {
const double kLn2 = 0.693147180559945309417232121458176568;
// Go's pure-Go math.Log2 (Frexp based) and math.Log10 = Log2(x) * (Ln2/Ln10)
double go_log2(double x) {
  int e; double f = std::frexp(x, &e);
  if (f == 0.5) return (double)(e - 1);
  return std::log(f) * (1.0 / kLn2) + (double)e;
}
// Go evaluates the constant Ln2/Ln10 exactly and rounds once: 0x1.34413509f79ffp-2 (kLn2 / kLn10 in
// double arithmetic is one ulp lower and would make log10(100) < 2).
const double kLn2OverLn10 = 0x1.34413509f79ffp-2;
double go_log10(double x) { return go_log2(x) * kLn2OverLn10; }
uint32_t gcd32(uint32_t a, uint32_t b) { while (b) { uint32_t t = a % b; a = b; b = t; } return a; }

int64_t remaining_suspicion_ms(uint32_t n, uint32_t k, int64_t elapsed, int64_t min_ms, int64_t max_ms) {
  double frac = std::log((double)n + 1.0) / std::log((double)k + 1.0);
  double raw = (double)max_ms / 1000.0 - frac * ((double)max_ms / 1000.0 - (double)min_ms / 1000.0);
  int64_t timeout = (int64_t)std::floor(1000.0 * raw);
  if (timeout < min_ms) timeout = min_ms;
  return timeout - elapsed;
}

int validate(const swim_config* c) {
  if (!c || c->abi_version != SWIM_ABI_VERSION) return SWIM_EINVAL;
  if (c->n_nodes < 2 || c->n_replicas < 1) return SWIM_EINVAL;
  if ((uint64_t)c->n_nodes * c->n_replicas >= 0xFFFFFFFFull || c->n_nodes >= (1u << 28)) return SWIM_ERANGE;   // first accuser<<4 | leaving<<3 | confirmations
  if (c->view_cap > (1u << 20)) return SWIM_ERANGE;
  if (!c->gossip_interval_ms || !c->probe_interval_ms || !c->probe_timeout_ms) return SWIM_EINVAL;
  if (c->gossip_nodes < 1 || c->gossip_nodes > 8 || c->indirect_checks > 8) return SWIM_EINVAL;
  if (c->suspicion_mult < 1 || c->suspicion_mult > 6 || c->retransmit_mult < 1) return SWIM_EINVAL;
  if (c->awareness_max_mult < 1 || c->awareness_max_mult > 255) return SWIM_EINVAL;
  if (c->queue_cap < 1 || c->queue_cap > 32 || c->inbox_cap < 1 || c->subject_cap < 1) return SWIM_EINVAL;
  if (c->flags & SWIM_F_SERF_EVENTS)
    if (c->event_queue_cap < 1 || c->event_queue_cap > 32 || c->event_buffer < 1 || c->event_ids_per_ltime > 254) return SWIM_EINVAL;
  if (c->n_shards < 1 || c->shard_rank >= c->n_shards || c->n_nodes % c->n_shards) return SWIM_EINVAL;
  if (c->n_initial > c->n_nodes || c->n_initial == 1) return SWIM_EINVAL;
  if (c->phase_chunk & (c->phase_chunk - 1)) return SWIM_EINVAL;
  if ((c->flags & SWIM_F_COORDINATES) && (c->n_shards != 1 || c->rtt_scale_us > 10000000u || c->rtt_height_us > 1000000u || c->rtt_jitter_us > 1000000u)) return SWIM_EINVAL;
  if (c->mass_rows) {   // the dense pair store: two accuser names per pair, incarnation 26 bits, ids 22 bits, no per-timer n
    if (c->suspicion_mult > 4 || (c->n_initial && c->n_initial != c->n_nodes)) return SWIM_EINVAL;
    if (c->n_nodes > (1u << 22) || c->mass_rows > c->n_nodes) return SWIM_ERANGE;
  }
  if (c->flags & SWIM_F_UNBOUNDED_QUEUE) {   // the queue implied by the pair store: a wave per node, fan-out <= 4, the column's queue word holds 5 bits of transmits
    if (!c->mass_rows || c->gossip_nodes > 4) return SWIM_EINVAL;
    uint32_t min_len = std::min(std::min(c->msg_len[0], c->msg_len[1]), c->msg_len[2]);
    if (c->udp_buffer_size / (2 + min_len) >= SW_IQ_PKT) return SWIM_ERANGE;      // rumours one packet can take
  }
  return SWIM_OK;
}
uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }
}  // namespace

extern "C" int swim_config_preset(swim_config* c, int preset) {
  if (!c) return SWIM_EINVAL;
  memset(c, 0, sizeof *c);
  c->abi_version = SWIM_ABI_VERSION;
  c->n_nodes = 128; c->n_replicas = 1;
  // memberlist.DefaultLANConfig (agent/config/runtime.go:1285-1350 documents the six Consul knobs)
  c->indirect_checks = 3; c->retransmit_mult = 4; c->suspicion_mult = 4;
  c->suspicion_max_timeout_mult = 6; c->probe_timeout_ms = 500; c->probe_interval_ms = 1000;
  c->awareness_max_mult = 8; c->gossip_nodes = 3; c->gossip_interval_ms = 200;
  c->gossip_to_dead_ms = 30000; c->udp_buffer_size = 1400; c->push_pull_interval_ms = 30000;
  switch (preset) {
    case SWIM_PRESET_LAN: break;
    case SWIM_PRESET_WAN:   // memberlist.DefaultWANConfig (runtime.go:1362-1427)
      c->suspicion_mult = 6; c->probe_timeout_ms = 3000; c->probe_interval_ms = 5000;
      c->gossip_nodes = 4; c->gossip_interval_ms = 500; c->gossip_to_dead_ms = 60000; c->push_pull_interval_ms = 60000; break;
    case SWIM_PRESET_LOCAL:
      c->indirect_checks = 1; c->retransmit_mult = 2; c->suspicion_mult = 3;
      c->probe_timeout_ms = 200; c->gossip_interval_ms = 100; c->gossip_to_dead_ms = 15000; c->push_pull_interval_ms = 15000; break;
    default: return SWIM_EINVAL;
  }
  c->msg_len[SWIM_MSG_ALIVE] = 128; c->msg_len[SWIM_MSG_SUSPECT] = 48;
  c->msg_len[SWIM_MSG_DEAD] = 48; c->msg_len[SWIM_MSG_USER] = 64;
  // ping / indirectPingReq / ackResp (serf puts a coordinate in Payload) / nackResp, msgpack with field names
  c->ctl_len[SWIM_CTL_PING] = 86; c->ctl_len[SWIM_CTL_INDIRECT] = 122; c->ctl_len[SWIM_CTL_ACK] = 108; c->ctl_len[SWIM_CTL_NACK] = 13;
  c->queue_cap = 8; c->inbox_cap = 32; c->subject_cap = 8; c->view_cap = 0; c->fold_interval_ms = 0;
  c->event_queue_cap = 8; c->event_buffer = 512;
  c->flags = SWIM_F_DEFAULT; c->watch_node = 0; c->n_shards = 1; c->seed = 1;
  c->rtt_scale_us = 40000; c->rtt_height_us = 2000; c->rtt_jitter_us = 0;
  return SWIM_OK;
}

extern "C" int swim_config_derive(const swim_config* c, swim_derived* d) {
  int rc = validate(c);
  if (rc) return rc;
  if (!d) return SWIM_EINVAL;
  memset(d, 0, sizeof *d);
  uint32_t q = c->quantum_ms ? c->quantum_ms
                             : gcd32(gcd32(c->gossip_interval_ms, c->probe_interval_ms), c->probe_timeout_ms);
  if (c->gossip_interval_ms % q || c->probe_interval_ms % q || c->probe_timeout_ms % q) return SWIM_EINVAL;
  d->quantum_ms = q;
  d->gossip_period = c->gossip_interval_ms / q;
  d->probe_period = c->probe_interval_ms / q;
  d->probe_timeout_ticks = c->probe_timeout_ms / q;
  uint32_t ch = c->phase_chunk;
  if (!ch) {
    ch = 256;
    while (ch > 1 && (uint64_t)ch * d->gossip_period * d->probe_period * 8 > c->n_nodes) ch >>= 1;
  }
  d->phase_chunk = ch;
  const double n = (double)c->n_nodes;
  d->retransmit_limit = c->retransmit_mult * (uint32_t)std::ceil(go_log10(n + 1.0));
  double scale = std::max(1.0, go_log10(std::max(1.0, n)));
  int64_t scale_milli = (int64_t)(scale * 1000.0);
  d->node_scale_milli = (uint32_t)scale_milli;
  int64_t min_ns = (int64_t)c->suspicion_mult * scale_milli * ((int64_t)c->probe_interval_ms * 1000000) / 1000;
  int64_t max_ns = (int64_t)c->suspicion_max_timeout_mult * min_ns;
  int64_t min_ms = min_ns / 1000000, max_ms = max_ns / 1000000;
  if (max_ms > 0x7FFFFFFF) return SWIM_ERANGE;
  d->suspicion_min_ms = (uint32_t)min_ms; d->suspicion_max_ms = (uint32_t)max_ms;
  int32_t k = std::max(0, (int32_t)c->suspicion_mult - 2);
  if ((int64_t)c->n_nodes - 2 < k) k = 0;
  d->suspicion_k = (uint32_t)k;
  d->suspicion_timeout_ms[0] = (uint32_t)(k < 1 ? min_ms : max_ms);
  for (int32_t i = 1; i <= k && i < 8; i++)
    d->suspicion_timeout_ms[i] = (uint32_t)remaining_suspicion_ms((uint32_t)i, (uint32_t)k, 0, min_ms, max_ms);
  d->push_pull_scale = c->n_nodes <= 32 ? 1u : (uint32_t)(std::ceil(go_log2(n) - go_log2(32.0)) + 1.0);
  {   // pushPullTrigger: every pushPullScale(PushPullInterval, n)
    uint64_t per = (uint64_t)c->push_pull_interval_ms * d->push_pull_scale / q;
    if (per > 0x7FFFFFFFull) return SWIM_ERANGE;
    d->push_pull_period_ticks = (uint32_t)per;
  }
  d->packet_budget = c->udp_buffer_size > 2 ? c->udp_buffer_size - 2 : 0;
  if (d->retransmit_limit > 255) return SWIM_ERANGE;     // transmits is an 8-bit field of the queue entry's meta word
  d->view_cap = c->view_cap ? c->view_cap : std::min<uint32_t>(c->n_nodes, 32);
  d->fold_period_ticks = (c->fold_interval_ms + q - 1) / q;
  d->reap_period_ticks = (c->reap_interval_ms + q - 1) / q;
  d->reconnect_period_ticks = (c->reconnect_interval_ms + q - 1) / q;
  return SWIM_OK;
}

extern "C" const char* swim_backend(void) { return "hip-gfx950"; }
extern "C" const char* swim_last_error(swim_sim* s) { return s ? s->err : "null handle"; }

// ---------------------------------------------------------------------------------------------
// lifecycle
// ---------------------------------------------------------------------------------------------
template <typename T>
static int dalloc(swim_sim* s, T** p, size_t count) {
  void* v = nullptr;
  size_t bytes = std::max<size_t>(count * sizeof(T), 64);
  HIPCK(s, hipMalloc(&v, bytes));
  s->allocs.push_back(v); s->alloc_bytes.push_back(bytes);
  *p = (T*)v;
  return SWIM_OK;
}
#define DALLOC(s, p, n)                       \
  do {                                        \
    int rc_ = dalloc((s), &(p), (n));         \
    if (rc_) { swim_destroy(s); return rc_; } \
  } while (0)

namespace {
// `nonce` says which PROCESS exported the mailbox (a pid does not: shards in different containers or pid namespaces often share
// one); a raw pointer is only taken from a handle this very process exported (registry below), everything else is an IPC mapping.
struct XchgHandle { hipIpcMemHandle_t ipc; uint64_t ptr, nonce; uint32_t rank, n_shards, mail_cap, magic; };
static_assert(sizeof(XchgHandle) <= SWIM_XCHG_HANDLE_BYTES, "handle does not fit");
const uint32_t kXchgMagic = 0x58434848u;
std::mutex g_xchg_mu;
std::vector<uint64_t> g_xchg_exported;       // mailboxes this process exported (and still owns)
uint64_t process_nonce() {
  static uint64_t n = [] {
    uint64_t v = 0;
    if (FILE* f = fopen("/dev/urandom", "rb")) { if (fread(&v, 8, 1, f) != 1) v = 0; fclose(f); }
    if (!v) v = ((uint64_t)getpid() << 32) ^ (uint64_t)(uintptr_t)&v ^ (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
    return v | 1u;
  }();
  return n;
}
}
static void drop_graphs(swim_sim* s) {
  for (int i = 0; i < 3; i++)
    if (s->graph_exec[i]) { (void)hipGraphExecDestroy(s->graph_exec[i]); s->graph_exec[i] = nullptr; }
  if (s->graph_end_begin) { (void)hipGraphExecDestroy(s->graph_end_begin); s->graph_end_begin = nullptr; }
  if (s->graph_xchg) { (void)hipGraphExecDestroy(s->graph_xchg); s->graph_xchg = nullptr; }
}

extern "C" int swim_destroy(swim_sim* s) {
#ifdef SWIMSIM_WAVECLK
  if (s) for (int kk = 0; kk < 2; kk++) {         // diagnostics: the workgroups of the last k_begin / k_deliver launch
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    std::vector<unsigned long long> c((size_t)BCLK_ROWS * 4);
    if (hipMemcpyFromSymbol(c.data(), HIP_SYMBOL(g_bclk), c.size() * 8, (size_t)kk * BCLK_ROWS * 4 * 8) != hipSuccess) continue;
    unsigned long long last = 0;
    for (size_t w = 0; w < BCLK_ROWS; w++) last = std::max(last, c[w * 4]);
    std::vector<const unsigned long long*> rows; unsigned long long k0 = ~0ull, k1 = 0;
    for (size_t w = 0; w < BCLK_ROWS; w++) { const unsigned long long* r = &c[w * 4]; if (r[0] && last - r[0] < 20000) { rows.push_back(r); k0 = std::min(k0, r[0]); k1 = std::max(k1, r[1]); } }
    if (rows.empty()) continue;
    const double us = 0.01;
    fprintf(stderr, "[block clk] %s, last launch: %zu workgroups, span %.1f us\n", kk ? "k_deliver" : "k_begin", rows.size(), (k1 - k0) * us);
    if (!kk) {
      static const char* const names[8] = { "expire", "pending", "probe", "gossip", "ppreply", "carry", "pushpull", "join" };
      for (unsigned role = 0; role < 8; role++) {
        std::vector<double> d; double first = 1e30, lastend = 0;
        for (auto r : rows) if (r[2] == role) { d.push_back((r[1] - r[0]) * us); first = std::min(first, (r[0] - k0) * us); lastend = std::max(lastend, (r[1] - k0) * us); }
        if (d.empty()) continue;
        std::sort(d.begin(), d.end());
        fprintf(stderr, "[block clk]   %-8s %6zu blocks, first start %.1f, last end %.1f; duration us: p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f\n", names[role], d.size(), first, lastend, d[d.size() / 10], d[d.size() / 2], d[d.size() * 9 / 10], d[d.size() * 99 / 100], d.back());
      }
    } else {
      std::vector<std::pair<double, unsigned long long>> d;
      for (auto r : rows) d.push_back({ (r[1] - r[0]) * us, r[2] });
      std::sort(d.begin(), d.end());
      fprintf(stderr, "[block clk]   duration us: p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f (records of the slowest: %llu)\n", d[d.size() / 10].first, d[d.size() / 2].first, d[d.size() * 9 / 10].first, d[d.size() * 99 / 100].first, d.back().first, d.back().second & 0x3FFFFFFFull);
      double rec[4] = { 0, 0, 0, 0 }, dur[4] = { 0, 0, 0, 0 };      // by record count: 0, 1..256, 257..1024, more
      for (auto r : rows) { const unsigned long long n = r[2] & 0x3FFFFFFFull; const int b = n == 0 ? 0 : n <= 256 ? 1 : n <= 1024 ? 2 : 3; rec[b] += 1; dur[b] += (r[1] - r[0]) * us; }
      static const char* const bn[4] = { "no records", "1..256", "257..1024", "more" };
      for (int b = 0; b < 4; b++) if (rec[b]) fprintf(stderr, "[block clk]   blocks with %-10s %6.0f, mean duration %.1f us\n", bn[b], rec[b], dur[b] / rec[b]);
    }
    const int NBK = 24; const double bw = (k1 - k0) * us / NBK;
    std::vector<int> alive(NBK, 0);
    for (auto r : rows) { int a = std::min<int>(NBK - 1, (int)((r[0] - k0) * us / bw)), b = std::min<int>(NBK - 1, (int)((r[1] - k0) * us / bw)); for (int i = a; i <= b; i++) alive[i]++; }
    fprintf(stderr, "[block clk]   workgroups alive per %.1f us bucket:", bw);
    for (int i = 0; i < NBK; i++) fprintf(stderr, " %d", alive[i]);
    fprintf(stderr, "\n");
  }
  if (s) {                                       // diagnostics: the profile of k_resolve's last launch, wave by wave (s_memtime = 100 MHz)
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    std::vector<unsigned long long> c((size_t)WCLK_ROWS * 6);
    if (hipMemcpyFromSymbol(c.data(), HIP_SYMBOL(g_wclk), c.size() * 8) == hipSuccess) {
      unsigned long long last = 0;
      for (size_t w = 0; w < WCLK_ROWS; w++) last = std::max(last, c[w * 6]);
      std::vector<const unsigned long long*> rows;
      unsigned long long k0 = ~0ull, k1 = 0;
      for (size_t w = 0; w < WCLK_ROWS; w++) { const unsigned long long* r = &c[w * 6]; if (r[0] && last - r[0] < 20000) { rows.push_back(r); k0 = std::min(k0, r[0]); k1 = std::max(k1, r[3]); } }
      if (!rows.empty()) {
        const double us = 0.01;
        fprintf(stderr, "[wave clk] k_resolve, last launch: %zu waves that found work, span %.1f us\n", rows.size(), (k1 - k0) * us);
        double ph[3] = { 0, 0, 0 }, life = 0; double by_pass[4][2] = { { 0 } };
        for (auto r : rows) { ph[0] += (r[1] - r[0]) * us; ph[1] += (r[2] - r[1]) * us; ph[2] += (r[3] - r[2]) * us; life += (r[3] - r[0]) * us;
          const int p = std::min<int>(3, (int)((r[4] + 255) / 256)); by_pass[p][0] += 1; by_pass[p][1] += (r[3] - r[0]) * us; }
        fprintf(stderr, "[wave clk]   mean us: compaction %.1f  list walk %.1f  flush %.1f  lifetime %.1f\n", ph[0] / rows.size(), ph[1] / rows.size(), ph[2] / rows.size(), life / rows.size());
        for (int p = 0; p < 4; p++) if (by_pass[p][0]) fprintf(stderr, "[wave clk]   tiles with %d%s list pass(es): %.0f waves, mean lifetime %.1f us\n", p, p == 3 ? "+" : "", by_pass[p][0], by_pass[p][1] / by_pass[p][0]);
        const int NBK = 40; const double bw = (k1 - k0) * us / NBK;
        std::vector<int> alive(NBK, 0), started(NBK, 0), ended(NBK, 0);
        for (auto r : rows) { int a = std::min<int>(NBK - 1, (int)((r[0] - k0) * us / bw)), b = std::min<int>(NBK - 1, (int)((r[3] - k0) * us / bw)); started[a]++; ended[b]++; for (int i = a; i <= b; i++) alive[i]++; }
        fprintf(stderr, "[wave clk]   per %.1f us bucket: waves alive / started / ended\n", bw);
        for (int i = 0; i < NBK; i++) fprintf(stderr, "[wave clk]   %6.1f  %5d %5d %5d\n", i * bw, alive[i], started[i], ended[i]);
        std::vector<double> lt; for (auto r : rows) lt.push_back((r[3] - r[0]) * us);
        std::sort(lt.begin(), lt.end());
        fprintf(stderr, "[wave clk]   lifetime us: min %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f\n", lt.front(), lt[lt.size() / 10], lt[lt.size() / 2], lt[lt.size() * 9 / 10], lt[lt.size() * 99 / 100], lt.back());
      }
    }
  }
#endif
#ifdef SWIMSIM_DIAG
  if (s && s->D.iq && getenv("SWIMSIM_IQCLK")) {      // diagnostics: where the waves of k_gossip_iq spent their nodes
    unsigned long long c[8];
    if (hipMemcpyFromSymbol(c, HIP_SYMBOL(g_iqclk), sizeof c) == hipSuccess && c[5]) {
      static const char* const nm[5] = { "column scan (with its compactions)", "  compactions alone", "picks + bumps + re-sorts", "loads, filter, records", "write-back" };
      const double nn = (double)c[5], tot = (double)(c[0] + c[2] + c[3] + c[4]);
      fprintf(stderr, "[iq clk] %llu nodes, %.2f compactions per node; s_memtime ticks per node: %.0f; of the scan, waiting for the batches' loads: %.0f (the piggy-back kernel's scans add to this one)\n", c[5], c[6] / nn, tot / nn, c[7] / nn);
      for (int p = 0; p < 5; p++) fprintf(stderr, "[iq clk]   %-36s %9.0f  %5.1f %%\n", nm[p], c[p] / nn, 100.0 * c[p] / tot);
    }
  }
#endif
#ifdef SWIMSIM_DIAG
  if (s && getenv("SWIMSIM_RESOLVECLK")) {      // diagnostics: where the waves of k_resolve's LAST launches spent their lives
    std::vector<uint32_t> c((size_t)RCLK_ROWS * 8);
    if (hipMemcpyFromSymbol(c.data(), HIP_SYMBOL(g_rclk), c.size() * 4) == hipSuccess) {
      static const char* const nm[7] = { "compaction", "line+hdr+vmeta", "queue+first view", "messages", "write-back", "flush", "wave lifetime" };
      double sum[7] = { 0 }; uint32_t mx[7] = { 0 }; size_t n = 0; double msgs = 0, passes = 0;
      for (size_t w = 0; w < RCLK_ROWS; w++) {
        const uint32_t* row = &c[w * 8];
        if (!row[6]) continue;
        n++; msgs += row[7] & 0xFFFFu; passes += row[7] >> 16;
        for (int p = 0; p < 7; p++) { sum[p] += row[p]; if (row[p] > mx[p]) mx[p] = row[p]; }
      }
      if (n) {
        fprintf(stderr, "[resolve clk] %zu waves that found work (s_memtime ticks)\n", n);
        for (int p = 0; p < 7; p++) fprintf(stderr, "[resolve clk]   %-18s mean %9.1f  max %9u\n", nm[p], sum[p] / n, mx[p]);
        fprintf(stderr, "[resolve clk]   list passes per wave %.2f; messages per lane, wave maximum: mean %.2f\n", passes / n, msgs / n);
      }
    }
  }
#endif
  if (!s) return SWIM_EINVAL;
  if (s->stream) (void)hipStreamSynchronize(s->stream);
#ifdef SWIMSIM_DIAG
  if (s->D.role_clk && getenv("SWIMSIM_ROLECLK")) {          // diagnostics: per tick, per role: start/end of the role's blocks
    std::vector<unsigned long long> raw((size_t)s->D.role_clk_ticks * 16 * 64), c((size_t)s->D.role_clk_ticks * 16);
    if (hipMemcpy(raw.data(), s->D.role_clk, raw.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
      for (size_t i = 0; i < c.size() / 2; i++) { c[2 * i] = ~0ull; c[2 * i + 1] = 0; for (int k = 0; k < 64; k++) { c[2 * i] = std::min(c[2 * i], raw[(i * 64 + k) * 2]); c[2 * i + 1] = std::max(c[2 * i + 1], raw[(i * 64 + k) * 2 + 1]); } }
      if (FILE* f = fopen(getenv("SWIMSIM_ROLECLK"), "a")) {
        static const char* const names[8] = { "expire", "pending", "probe", "gossip", "ppreply", "carry", "pushpull", "-" };
        for (uint32_t t = 0; t < s->D.role_clk_ticks && t < s->tick; t++) {
          unsigned long long t0 = ~0ull;
          for (int r = 0; r < 7; r++) if (c[(t * 8 + r) * 2 + 1]) t0 = std::min(t0, c[(t * 8 + r) * 2]);
          fprintf(f, "tick %u", t);
          for (int r = 0; r < 7; r++) if (c[(t * 8 + r) * 2 + 1]) fprintf(f, " %s %.1f-%.1f", names[r], (c[(t * 8 + r) * 2] - t0) / 100.0, (c[(t * 8 + r) * 2 + 1] - t0) / 100.0);
          fprintf(f, "\n");
        }
        fclose(f);
      }
    }
  }
#endif
  drop_graphs(s);
  if (s->mailbox) { std::lock_guard<std::mutex> g(g_xchg_mu); auto it = std::find(g_xchg_exported.begin(), g_xchg_exported.end(), (uint64_t)(uintptr_t)s->mailbox); if (it != g_xchg_exported.end()) g_xchg_exported.erase(it); }
  for (void* p : s->ipc_opened) (void)hipIpcCloseMemHandle(p);
  for (void* p : s->allocs) (void)hipFree(p);
  for (hipEvent_t e : s->ev_pool) (void)hipEventDestroy(e);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
  return SWIM_OK;
}

extern "C" int swim_create(const swim_config* cfg, swim_sim** out) {
  swim_derived d;
  int rc = swim_config_derive(cfg, &d);
  if (rc) return rc;
  if (!out) return SWIM_EINVAL;
  if (cfg->n_shards > SW_MAX_SHARDS || cfg->subject_cap >= NW_SLOT_MASK) return SWIM_ERANGE;
  if ((uint64_t)cfg->n_replicas * cfg->subject_cap > (1u << 24)) return SWIM_ERANGE;   // watch slots are observation only: keep them few
  {   // the gossip role stages both queues of its 256 lanes in LDS: 16 B x 256 x (queue_cap + event_queue_cap) of the CU's 160 KB
    const uint32_t slots = cfg->queue_cap + ((cfg->flags & SWIM_F_SERF_EVENTS) ? cfg->event_queue_cap : 0);
    if ((size_t)slots * SW_BLOCK * sizeof(uint4) + 8192 > 160 * 1024) return SWIM_ERANGE;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || (int)cfg->device >= ndev) return SWIM_ENODEV;
  swim_sim* s = new (std::nothrow) swim_sim();
  if (!s) return SWIM_ENOMEM;
  s->cfg = *cfg; s->d = d;
  if (hipSetDevice((int)cfg->device) != hipSuccess || hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) { delete s; return SWIM_ENODEV; }
  SwDev& D = s->D;
  memset(&D, 0, sizeof D);
  const bool serf = (cfg->flags & SWIM_F_SERF_EVENTS) != 0;
  D.N = cfg->n_nodes; D.R = cfg->n_replicas; D.nloc = D.N / cfg->n_shards; D.i0 = cfg->shard_rank * D.nloc;
  D.n_shift = (D.N & (D.N - 1)) ? 0xFFFFFFFFu : (uint32_t)__builtin_ctz(D.N);
  D.nloc_shift = (D.nloc & (D.nloc - 1)) ? 0xFFFFFFFFu : (uint32_t)__builtin_ctz(D.nloc);
  if (cfg->n_shards > 1 && D.nloc % d.phase_chunk) { swim_destroy(s); return SWIM_EINVAL; }
  D.S = cfg->subject_cap; D.Q = cfg->queue_cap; D.C = cfg->inbox_cap; D.C2 = D.C > SW_INBOX_FAST ? D.C : 0;   // the overflow row has room for ALL C messages: a big inbox is sorted in it
  D.bigsort_cap = D.C >= SW_BIGSORT_MIN ? std::min<uint32_t>(D.C, SW_BIGSORT_MAX) : 0;                         // ... by k_inbox_sort from SW_BIGSORT_MIN messages on, by its lane in k_resolve below that
  if (const char* e = getenv("SWIMSIM_BIGSORT")) if (!atoi(e)) D.bigsort_cap = 0;                             // (A/B: the lane's heapsort for every size)
  D.EQ = serf ? cfg->event_queue_cap : 0; D.EB = serf ? cfg->event_buffer : 0;
  D.EW = serf ? ((cfg->event_ids_per_ltime ? cfg->event_ids_per_ltime : 14) + 2 + 3) / 4 : 0;
  D.G = d.gossip_period; D.P = d.probe_period; D.TQ = d.probe_timeout_ticks; D.CH = d.phase_chunk;
  D.quantum_ms = d.quantum_ms; D.k_gossip = cfg->gossip_nodes; D.k_indirect = cfg->indirect_checks;
  D.retransmit_limit = d.retransmit_limit; D.susp_k = d.suspicion_k; D.awareness_max = cfg->awareness_max_mult;
  D.gossip_to_dead_ms = cfg->gossip_to_dead_ms; D.budget = d.packet_budget; D.flags = cfg->flags;
  D.watch = cfg->watch_node; D.trace_ticks = cfg->trace_ticks; D.n_shards = cfg->n_shards; D.rank = cfg->shard_rank;
  D.fast_blocks = (D.CH == SW_BLOCK && D.nloc % SW_BLOCK == 0) ? 1u : 0u;
  D.pp_period = d.push_pull_period_ticks;
  for (int i = 0; i < 4; i++) { D.msg_len[i] = cfg->msg_len[i]; D.ctl_len[i] = cfg->ctl_len[i]; }
  for (int i = 0; i < 8; i++) D.susp_timeout[i] = d.suspicion_timeout_ms[i];
  D.loss_q32 = cfg->loss_q32; D.seed = cfg->seed;
#ifdef SWIMSIM_DIAG
  const bool want_role_clk = getenv("SWIMSIM_ROLECLK") != nullptr;
#else
  const bool want_role_clk = false;
#endif

  const size_t NT = (size_t)D.N * D.R, NL = (size_t)D.nloc * D.R, NS = (size_t)D.R * D.S, NB = cdiv(NL, SW_BLOCK);
  DALLOC(s, D.tick, 1);
  if (want_role_clk) { D.role_clk_ticks = 1024; DALLOC(s, D.role_clk, (size_t)D.role_clk_ticks * 16 * 64); }
  DALLOC(s, D.nw, NT);
  DALLOC(s, D.exc_ent, (size_t)D.R * SW_EXC_MAX); DALLOC(s, D.exc_cnt, D.R); DALLOC(s, D.exc_dirty, D.R);
  DALLOC(s, D.hdr, NL); DALLOC(s, D.ph, NL); DALLOC(s, D.pr0, NL);
  DALLOC(s, D.q, NL * D.Q); DALLOC(s, D.inbox1, NL * 16);
  // pooled overflow rows (swim_device.h): from inbox_cap 4 096 on a node's own row holds 1 024 messages and the rare node that receives more in a tick
  // (a state exchange during a mass event) borrows one of PB big rows
  D.C1 = D.C2; D.PB = 0;
  { const char* e = getenv("SWIMSIM_INBOX_POOL");
    if (D.C >= SW_INBOX_POOL_MIN && !(e && !atoi(e))) { D.C1 = SW_INBOX_POOL_C1; D.PB = (uint32_t)std::min<size_t>(std::max<size_t>(NL / 64, 256), 8192); } }
  if (const char* e = getenv("SWIMSIM_INBOX_POOL_C1")) {   // (tests: pooled rows at any size, the node's own row as small as asked)
    const uint32_t c1 = (uint32_t)atoi(e);
    if (c1 >= 16 && D.C > c1) { D.C1 = c1; D.PB = (uint32_t)std::min<size_t>(std::max<size_t>(NL / 64, 256), 8192); }
  }
  DALLOC(s, D.in_cnt, NL); DALLOC(s, D.inbox2, NL * D.C1 * 3);
  if (D.PB) {
    DALLOC(s, D.inbox_big, (size_t)D.PB * D.C2 * 3); DALLOC(s, D.big_row, NL); DALLOC(s, D.big_list, D.PB); DALLOC(s, D.big_n, 1); DALLOC(s, D.defer_n, 1);
    D.defer_cap = (uint32_t)std::min<size_t>((size_t)D.PB * D.C2 / 4, (size_t)1 << 26);
    DALLOC(s, D.defer_rec, D.defer_cap); DALLOC(s, D.defer_l, D.defer_cap);
    HIPCK(s, hipMemsetAsync(D.big_row, 0xFF, NL * 4, s->stream)); HIPCK(s, hipMemsetAsync(D.big_n, 0, 4, s->stream)); HIPCK(s, hipMemsetAsync(D.defer_n, 0, 4, s->stream));
  }
  DALLOC(s, D.q_any, NB); DALLOC(s, D.in_any, cdiv(NL, 64)); DALLOC(s, D.alive_cnt, NB); DALLOC(s, D.qbits, cdiv(NL, 32) + 2);
  if (serf) { DALLOC(s, D.evq, NL * D.EQ); DALLOC(s, D.ring, NL * D.EB * D.EW); DALLOC(s, D.evseq, NL); }
  // explicit views: VT slots per lane, a power of two >= 2*(view_cap+1) so that a probe always meets a free slot
  D.view_cap = d.view_cap; D.fold_period = d.fold_period_ticks;
  D.reap_period = d.reap_period_ticks; D.reconnect_timeout_ms = cfg->reconnect_timeout_ms; D.tombstone_timeout_ms = cfg->tombstone_timeout_ms;
  D.rc_period = d.reconnect_period_ticks;
  { uint32_t tb = 2; while ((1ull << tb) < 2ull * (D.view_cap + 1)) tb++; D.VT = 1u << tb; D.vt_shift = 32 - tb; }
  DALLOC(s, D.vt, NL * D.VT); DALLOC(s, D.vc, NL * D.VT);
  if (serf) {   // serf's member.statusLTime per explicit view and for the agent's own entry (swim_device.h); the dense store's plane below
    DALLOC(s, D.vs, NL * D.VT); DALLOC(s, D.sslt, NL);
    HIPCK(s, hipMemsetAsync(D.vs, 0, NL * D.VT * 4, s->stream)); HIPCK(s, hipMemsetAsync(D.sslt, 0, NL * 4, s->stream));
  }
  DALLOC(s, D.vmeta, NL);
  DALLOC(s, D.dl_blk, NB); DALLOC(s, D.bk, NT); DALLOC(s, D.acting, D.R);
  D.M = cfg->mass_rows; D.nbl = cdiv(D.nloc, SW_BLOCK);
  if (D.M) {   // the dense pair store (swim_device.h): 12 bytes per (row, observer)
    const size_t RM = (size_t)D.R * D.M, pairs = RM * cdiv(D.nloc, 64) * 64;      // (m_idx: [replica][64 observers][row][64])
    if (RM >= 0x7FFFFFFFull) { swim_destroy(s); return SWIM_ERANGE; }
    DALLOC(s, D.mrow, NT); DALLOC(s, D.mrow_subj, RM); DALLOC(s, D.m_free, RM); DALLOC(s, D.m_nfree, D.R);
    DALLOC(s, D.mA, pairs); DALLOC(s, D.mB, pairs); DALLOC(s, D.mC, pairs);
    if (serf) { DALLOC(s, D.mD, pairs); HIPCK(s, hipMemsetAsync(D.mD, 0, pairs * 4, s->stream)); }
    DALLOC(s, D.m_tile_dl, RM * D.nbl); DALLOC(s, D.m_row_dl, RM); DALLOC(s, D.mcnt, NL);
    if (D.rc_period) { const size_t rcl = (size_t)D.R * cdiv((uint64_t)cdiv(D.N, D.rc_period) * std::min(D.P, D.rc_period), SW_BLOCK) * SW_BLOCK; DALLOC(s, D.rc_cnt, rcl); DALLOC(s, D.rc_best, rcl); }
    DALLOC(s, D.m_rev, cdiv(D.nbl, 32)); HIPCK(s, hipMemsetAsync(D.m_rev, 0, (size_t)cdiv(D.nbl, 32) * 4, s->stream));
    DALLOC(s, D.m_due, RM); DALLOC(s, D.m_due_cnt, 1); HIPCK(s, hipMemsetAsync(D.m_due_cnt, 0, 4, s->stream));
    HIPCK(s, hipMemsetAsync(D.mrow, 0xFF, NT * 4, s->stream)); HIPCK(s, hipMemsetAsync(D.mA, 0, pairs * 4, s->stream));
    HIPCK(s, hipMemsetAsync(D.mB, 0, pairs * 4, s->stream)); HIPCK(s, hipMemsetAsync(D.mC, 0, pairs * 4, s->stream));
    D.iq = (cfg->flags & SWIM_F_UNBOUNDED_QUEUE) ? 1u : 0u; D.MB = cdiv(D.M, SW_IQ_RB);
    if (D.iq) {   // SWIM_F_UNBOUNDED_QUEUE (swim_device.h): 8 more bytes per pair, the queue word in its own column-major layout
      const size_t epairs = (size_t)D.R * cdiv(D.nloc, 64) * D.MB * 64 * SW_IQ_RB;
      DALLOC(s, D.mE, epairs); DALLOC(s, D.mF, pairs); DALLOC(s, D.iqn, NL);
      HIPCK(s, hipMemsetAsync(D.mE, 0, epairs * 4, s->stream)); HIPCK(s, hipMemsetAsync(D.mF, 0, pairs * 4, s->stream)); HIPCK(s, hipMemsetAsync(D.iqn, 0, NL * 4, s->stream));
      D.ord_cap = 16;
      DALLOC(s, D.ord, NL * D.ord_cap); DALLOC(s, D.ord_cnt, NL); DALLOC(s, D.ord_nodes, NL); DALLOC(s, D.ord_n, 1);
      HIPCK(s, hipMemsetAsync(D.ord_cnt, 0, NL * 4, s->stream)); HIPCK(s, hipMemsetAsync(D.ord_n, 0, 4, s->stream));
      // ranks of the three message lengths (0 = longest; equal lengths share a rank) and what one packet can take of each
      uint32_t lens[3] = { cfg->msg_len[0], cfg->msg_len[1], cfg->msg_len[2] };
      for (int t = 0; t < 3; t++) { uint32_t rk = 0; for (int u = 0; u < 3; u++) { bool seen = false; for (int v = 0; v < u; v++) seen |= lens[v] == lens[u]; if (!seen && lens[u] > lens[t]) rk++; } D.len_rank[t] = rk; }
      D.len_rank[3] = 0;
      for (int t = 0; t < 4; t++) D.iq_keep[t] = 0;
      for (int t = 0; t < 3; t++) D.iq_keep[D.len_rank[t]] = d.packet_budget / (2 + lens[t]) + 1;
      const uint32_t per_pkt_all = D.iq_keep[0] + D.iq_keep[1] + D.iq_keep[2];
      if (d.retransmit_limit > 31 || per_pkt_all * std::max<uint32_t>(4u, SW_IQ_ORDERS) + 64 + 32 > SW_IQ_POOL) { swim_destroy(s); return SWIM_ERANGE; }
    }
    HIPCK(s, hipMemsetAsync(D.m_tile_dl, 0xFF, RM * D.nbl * 4, s->stream));
  }
  DALLOC(s, D.peak, 1); HIPCK(s, hipMemsetAsync(D.peak, 0, 4, s->stream));
#ifndef SW_RESOLVE_PLAIN_ORDER
  if (D.CH >= 64 && NB > 4 * SW_RTILE) {        // the longest-job-first order of k_resolve's tiles, per probe phase (swim_device.h)
    const uint32_t TILE = resolve_tile(D), T = (uint32_t)cdiv(NL, TILE);
    std::vector<uint32_t> ord((size_t)D.P * T), due(T);
    for (uint32_t ph = 0; ph < D.P; ph++) {
      for (uint32_t tl = 0; tl < T; tl++) {
        uint32_t n = 0;
        for (size_t l = (size_t)tl * TILE; l < std::min<size_t>(NL, (size_t)(tl + 1) * TILE); l += std::min<uint32_t>(D.CH, TILE))
          n += ((D.i0 + (uint32_t)(l % D.nloc)) / D.CH / D.G) % D.P == ph;
        due[tl] = n;
      }
      uint32_t* o = &ord[(size_t)ph * T];
      for (uint32_t tl = 0; tl < T; tl++) o[tl] = tl;
      std::stable_sort(o, o + T, [&](uint32_t a, uint32_t b) { return due[a] > due[b]; });
    }
    DALLOC(s, D.rs_order, ord.size()); D.rs_T = T;
    HIPCK(s, hipMemcpy(D.rs_order, ord.data(), ord.size() * 4, hipMemcpyHostToDevice));
  }
#endif
  {   // dynamic membership: the scaling laws as tables / constants with Go's float64 semantics
    const uint32_t ni = cfg->n_initial ? cfg->n_initial : D.N;
    D.dyn = ni < D.N ? 1u : 0u;
    D.suspicion_mult = cfg->suspicion_mult; D.suspicion_max_mult = cfg->suspicion_max_timeout_mult;
    D.probe_interval_ms = cfg->probe_interval_ms; D.retransmit_mult = cfg->retransmit_mult;
    D.susp_k_cfg = (uint32_t)std::max(0, (int32_t)cfg->suspicion_mult - 2);
    DALLOC(s, D.base_known, D.R); DALLOC(s, D.join_cnt, 1); D.join_cap = 65536; DALLOC(s, D.join_list, D.join_cap);
    HIPCK(s, hipMemset(D.join_cnt, 0, 4));
    for (int j = 0; j < 12; j++) D.rl_steps[j] = 0xFFFFFFFFu;
    for (int j = 0; j < 8; j++) D.susp_frac[j] = 0.0;
    if (D.dyn) {
      DALLOC(s, D.vnk, NL);
      // ceil(log10(n+1)) as a step function of n: step j+1 is reached at the smallest n with ceil(...) >= j+1 (binary search
      // over the same float64 expression swim_config_derive uses)
      auto steps = [](uint32_t n) { return (uint32_t)std::ceil(go_log10((double)n + 1.0)); };
      for (uint32_t j = 0; j < 12; j++) {
        if (steps(0x3FFFFFFFu) < j + 1) break;
        uint32_t lo = 0, hi = 0x3FFFFFFFu;
        while (lo < hi) { uint32_t mid = lo + (hi - lo) / 2; if (steps(mid) >= j + 1) hi = mid; else lo = mid + 1; }
        D.rl_steps[j] = lo;
      }
      for (uint32_t c = 1; c < 8; c++) D.susp_frac[c] = std::log((double)c + 1.0) / std::log((double)D.susp_k_cfg + 1.0);
      std::vector<uint32_t> sm((size_t)D.N + 1);
      for (uint32_t n = 0; n <= D.N; n++) { double sc = std::max(1.0, go_log10(std::max(1.0, (double)n))); sm[n] = (uint32_t)(int64_t)(sc * 1000.0); }
      DALLOC(s, D.scale_milli, (size_t)D.N + 1);
      HIPCK(s, hipMemcpy(D.scale_milli, sm.data(), sm.size() * 4, hipMemcpyHostToDevice));
    }
  }
  {   // fold accumulators, grouped by the value a fold tick resets them to
    const size_t n = D.fold_period ? NT : 1;
    uint32_t* z; DALLOC(s, z, 5 * n + 16); s->fold_zero = z; s->fold_zero_bytes = (5 * n + 16) * 4;
    D.fl_cnt = z; D.fl_kmax = z + n; D.fl_bad = z + 2 * n; D.fg_cnt = z + 3 * n; D.fg_kmax = z + 4 * n; D.fold_any = z + 5 * n;
    uint32_t* f; DALLOC(s, f, 2 * n); s->fold_ones = f; s->fold_ones_bytes = 2 * n * 4;
    D.fl_kmin = f; D.fg_kmin = f + n;
  }
  DALLOC(s, D.subj_node, NS); DALLOC(s, D.n_slots, D.R); DALLOC(s, D.slot_dirty, NS);
  DALLOC(s, D.slot_maxinc, NS);
  if (cfg->flags & SWIM_F_COORDINATES) {
    DALLOC(s, D.coord, NL); DALLOC(s, D.c_adj, NL * SW_COORD_WINDOW); DALLOC(s, D.c_adj_idx, NL); DALLOC(s, D.c_lf, NL * SW_COORD_PEERS * 2);
    D.c_cap = (uint32_t)std::min<size_t>(NL, (size_t)D.R * cdiv(cdiv(D.nloc, d.probe_period), SW_BLOCK) * SW_BLOCK + (size_t)D.R * 2 * SW_BLOCK);   // at most the probe-due lanes of a tick
    DALLOC(s, D.c_list, D.c_cap); DALLOC(s, D.c_new, D.c_cap); DALLOC(s, D.c_cnt, 1);
    D.rtt_scale_us = cfg->rtt_scale_us; D.rtt_height_us = cfg->rtt_height_us; D.rtt_jitter_us = cfg->rtt_jitter_us;
  }
  DALLOC(s, D.cen_acc, NS * CEN_WORDS); DALLOC(s, D.census, NS);
  DALLOC(s, D.cf_ticket, 1); HIPCK(s, hipMemsetAsync(D.cf_ticket, 0, 4, s->stream));
  DALLOC(s, D.cen_dl, NS * 8); HIPCK(s, hipMemsetAsync(D.cen_dl, 0, NS * 8 * 4, s->stream));
  if (D.trace_ticks) DALLOC(s, D.trace, NS * D.trace_ticks * 5);

  // the fused first launch: block ranges per role (upper bounds of the stagger enumeration)
  const uint32_t nchunks = cdiv(D.nloc, D.CH) + 1;
  const uint32_t gossip_lanes = D.fast_blocks ? cdiv(cdiv(D.nloc, D.CH), D.G) * D.CH : (cdiv(nchunks, D.G) + 1) * D.CH;
  const uint32_t probe_lanes = (cdiv(cdiv(nchunks, D.G) + 1, D.P) + 1) * D.G * D.CH;
  BeginPlan& pl = s->plan;
  pl.nb_expire = cdiv(NB, SW_BLOCK / 64);           // expire: one wave per 256-node block, out after one word unless a deadline bound passed
  pl.nb_pend = 16;
  pl.nb_probe = cdiv(probe_lanes, SW_BLOCK);
  pl.nb_gossip = cdiv(gossip_lanes, SW_BLOCK);
  pl.nb_pp = D.pp_period ? cdiv((uint64_t)cdiv(D.N, D.pp_period) * std::min(D.P, D.pp_period), SW_BLOCK) : 0;
  pl.nb_ppreply = SW_PP_LISTS;                     // one block per request sub-list (4 blocks were a 40 us long pole every ProbeInterval);
                                                   // also without scheduled push-pull: a join asks for a state exchange any time
  pl.roles = (D.pp_period ? 0x1F : 0xF) | 0x40;
  const bool piggy = (cfg->flags & SWIM_F_PIGGYBACK) != 0;
  pl.nb_carry = D.n_shards > 1 ? 64 : 0;
  pl.nb_join = 1;                                   // swim_inject_join also restarts nodes of a fixed population
  if (piggy && D.n_shards > 1) pl.roles |= 0x20;
  D.pp_cap = std::max<uint32_t>(4096, 8 * D.R * ((D.pp_period ? cdiv(D.N, D.pp_period) * std::min(D.P, D.pp_period) : 0) +
                                                  (D.rc_period ? cdiv(D.N, D.rc_period) * std::min(D.P, D.rc_period) : 0)));   // pull requests of a boundary tick: push-pull + serf reconnect
  D.pp_cap = (D.pp_cap + SW_PP_LISTS - 1) / SW_PP_LISTS * SW_PP_LISTS * 4;   // 64 sub-lists, 4x slack for imbalance
  DALLOC(s, D.pp_list, (size_t)2 * D.pp_cap); DALLOC(s, D.pp_cnt, 2 * SW_PP_LISTS * 16);
  if (D.M) { D.xs_cap = 2 * D.pp_cap + D.join_cap; DALLOC(s, D.xs_list, D.xs_cap); DALLOC(s, D.xs_cnt, 1); HIPCK(s, hipMemsetAsync(D.xs_cnt, 0, 4, s->stream)); }
  D.pend_cap = pl.nb_probe * SW_BLOCK * D.R;
  DALLOC(s, D.pend, (size_t)D.pend_cap * (D.TQ + 1)); DALLOC(s, D.pend_cnt, D.TQ + 1);
  // worst-case records of one tick: a gossip block's private segment holds every packet it can emit
  uint32_t per_pkt = std::min<uint32_t>(D.Q + D.EQ, std::max<uint32_t>(1, D.budget / 4));
  if (D.iq) per_pkt = SW_IQ_PKT + D.EQ;               // an implied queue fills a packet to its byte budget
  D.nb_gossip = pl.nb_gossip; D.nb_probe = pl.nb_probe; D.n_seg = D.R * (pl.nb_gossip + pl.nb_probe);
  D.seg_cap = std::max<uint32_t>(SW_BLOCK * D.k_gossip * per_pkt, 2 * SW_BLOCK);   // a probe block files <= 2 orders per lane
  {   // carry areas: one per k_resolve block; a probe-due chunk is exactly one block, so all 256 nodes may answer
      // their own ping's order in the same tick, plus the acks and indirect legs they serve
    uint32_t min_len = std::min(std::min(cfg->msg_len[0], cfg->msg_len[1]), cfg->msg_len[2]) + 2;
    if (serf) min_len = std::min(min_len, cfg->msg_len[3] + 3);
    uint32_t fit = std::min<uint32_t>(D.Q + D.EQ, std::max<uint32_t>(1, D.budget / std::max(1u, min_len)));
    if (D.iq) fit = std::max<uint32_t>(1, D.budget / std::max(1u, min_len));
    D.NB = (uint32_t)NB; D.carry_cap = piggy ? 2 * SW_BLOCK * fit : 1;
    D.nb_carry = piggy ? 1 : 0;                      // (flag) k_deliver's segment blocks drain the carry areas too
    DALLOC(s, D.carry, piggy ? (size_t)2 * NB * D.carry_cap : 1); DALLOC(s, D.carry_cl, NB); DALLOC(s, D.att_any, 1); DALLOC(s, D.carry_stamp, 2);
  }
  uint64_t e_cap = (uint64_t)D.n_seg * D.seg_cap;
  if (e_cap > 0x7FFFFFFFull) { swim_destroy(s); return SWIM_ERANGE; }
  DALLOC(s, D.seg, e_cap); DALLOC(s, D.seg_cnt, D.n_seg); DALLOC(s, D.seg_last, D.n_seg);
  {   // tile buckets (swim_device.h): for handles without a dense pair store whose replicas span few tiles (a gossip block keeps a
      // histogram over them in LDS and pays one global atomic per tile it sends to: 64 tiles at 65 536 nodes per cluster — at a million
      // nodes per cluster nearly every record would pay its own).  SWIMSIM_TILEBUCKETS=0: the sender-side filter and k_deliver's scatter.
    D.tb_T = (uint32_t)cdiv(NL, SW_TB_TILE);     // (1 024 lanes per bucket, whatever k_resolve's own tile is)
    const uint64_t per_tile = (uint64_t)SW_TB_TILE * D.k_gossip * per_pkt / std::max(1u, D.G);      // expected records of a tile when every queue is full
    const uint64_t cap = 2 * per_tile + 3 * SW_TB_TILE;                                            // ... twice that, plus the orders (<= 2 per node)
    const char* e = getenv("SWIMSIM_TILEBUCKETS");
    // OFF unless asked for (SWIMSIM_TILEBUCKETS=1): bit-identical (the GPU suite runs with it too) and measured — k_begin -10 us per launch
    // without its 6 M random view reads per saturated tick, but the per-tile drain (sort + barriers: a chain per tile) costs k_deliver more
    // than its scattered deliveries did: 0.441 ms per round against 0.418 (profiles/r04_ab_experiments.txt).
    D.tb_on = (e && atoi(e) && !D.M && cdiv(D.nloc, SW_TB_TILE) + 2 <= SW_TB_BINS && cap * D.tb_T * sizeof(uint4) <= ((uint64_t)8 << 30) && cap < 0x7FFFFFFFull && D.N < (1u << 28)) ? 1u : 0u;
    D.tb_cap = D.tb_on ? (uint32_t)cap : 1; if (!D.tb_on) D.tb_T = 1;
    const char* ec = getenv("SWIMSIM_TB_CARRY");      // (A/B: 0 = the carried broadcasts stay with k_deliver)
    D.tb_carry = (D.tb_on && piggy && !(ec && !atoi(ec))) ? 1u : 0u;
    if (D.tb_carry) { pl.nb_carry = (uint32_t)cdiv(NB, SW_CARRY_GROUP); pl.roles |= 0x20; }      // k_begin's carry role files them in the buckets
    DALLOC(s, D.tb, D.tb_on ? (size_t)D.tb_T * D.tb_cap : 1); DALLOC(s, D.tb_cnt, D.tb_T); DALLOC(s, D.tb_last, D.tb_T); DALLOC(s, D.dbg_on, 1);
    HIPCK(s, hipMemsetAsync(D.tb_cnt, 0, (size_t)D.tb_T * 4, s->stream)); HIPCK(s, hipMemsetAsync(D.tb_last, 0, (size_t)D.tb_T * 4, s->stream)); HIPCK(s, hipMemsetAsync(D.dbg_on, 0, 4, s->stream));
  }
  if (D.iq) pl.roles &= ~0x8u;                        // gossip() of a handle whose queue the pair store implies is k_gossip_iq's (launch_begin)
  s->begin_kernel = select_begin(std::max(D.k_gossip, D.k_indirect), serf, D.n_shards > 1, cfg->mass_rows != 0, D.tb_on != 0);
  for (uint32_t sh = 0; sh < D.n_shards; sh++) {
    // own shard: probe verdicts, fold census records, push-pull; other shards: their share of the gossip records, the
    // acks' piggy-back orders, carried broadcasts, fold census records, and push-pull — whose every exchange sends one
    // record per explicit view of the sender, all of a boundary tick's exchanges possibly to the same shard
    const uint64_t rc_lanes = D.rc_period ? (uint64_t)cdiv((uint64_t)cdiv(D.N, D.rc_period) * std::min(D.P, D.rc_period), SW_BLOCK) * SW_BLOCK : 0;
    const uint64_t pp_burst = ((uint64_t)D.R * pl.nb_pp * SW_BLOCK + D.R * rc_lanes) * ((uint64_t)D.view_cap + D.M + 3);   // explicit views + own view of the receiver + the pull request
    const uint64_t fold_burst = D.fold_period ? NT : 0;        // a fold tick: at most one census record per node of the population
    uint64_t cap = sh == D.rank ? 2 * NL + 4096 + pp_burst + fold_burst
                                : std::max<uint64_t>(e_cap / D.n_shards * 2, 4096) + 2 * NL / D.n_shards + NL + pp_burst + fold_burst;
    if (cap > 0x7FFFFFFFull) { swim_destroy(s); return SWIM_ERANGE; }
    D.out_cap[sh] = (uint32_t)cap;
    DALLOC(s, D.out[sh], cap);
  }
  DALLOC(s, D.out_tab, SW_MAX_SHARDS); DALLOC(s, D.out_cap_tab, SW_MAX_SHARDS);
  HIPCK(s, hipMemcpy(D.out_tab, D.out, sizeof D.out, hipMemcpyHostToDevice)); HIPCK(s, hipMemcpy(D.out_cap_tab, D.out_cap, sizeof D.out_cap, hipMemcpyHostToDevice));
  DALLOC(s, D.out_cnt, SW_MAX_SHARDS + 1); DALLOC(s, s->d_last_cnt, SW_MAX_SHARDS);
  D.act = D.out_cnt + D.n_shards;                  // rides behind the counts so one gather fetches both
  DALLOC(s, D.peer_act, 1); HIPCK(s, hipMemsetD32Async((hipDeviceptr_t)D.peer_act, 1, 1, s->stream));
  D.ev_cap = 65536; DALLOC(s, D.events, D.ev_cap); DALLOC(s, D.ev_cnt, 1);
  DALLOC(s, D.ev_watch, (size_t)D.R * SWIM_EVENT_WATCHERS + D.R); HIPCK(s, hipMemsetAsync(D.ev_watch, 0, ((size_t)D.R * SWIM_EVENT_WATCHERS + D.R) * 4, s->stream));
  D.cap_cap = 1 << 18; DALLOC(s, D.cap, D.cap_cap); DALLOC(s, D.cap_dst, D.cap_cap); DALLOC(s, D.cap_cnt, 1);
  DALLOC(s, D.stats, (size_t)SW_STAT_COPIES * SW_STAT_STRIDE); DALLOC(s, D.err, 1);
  s->scratch_bytes = std::max<size_t>(1 << 20, std::min<size_t>(((size_t)D.VT + D.M) * 32 + 64, (size_t)1 << 26));   // an observer's views fit (swim_members)
  { uint8_t* p; DALLOC(s, p, s->scratch_bytes); s->d_scratch = (uint32_t*)p; }
  DALLOC(s, s->d_fresh, 1024);
  if (D.n_shards > 1) {   // what all the other shards together may address to this one in a tick (their lists are sized like ours)
    uint64_t in_cap = 0;
    for (uint32_t sh = 0; sh < D.n_shards; sh++) if (sh != D.rank) in_cap += D.out_cap[sh];
    s->in_cap = (uint32_t)std::min<uint64_t>(in_cap, 0x7FFFFFFFull); DALLOC(s, s->in_buf, s->in_cap);
    // swim_xchg_*: the mailbox the other shards write into (every shard sizes its lists alike, so a source's area
    // holds whatever its list for this shard can)
    uint32_t mc = 0; for (uint32_t sh = 0; sh < D.n_shards; sh++) if (sh != D.rank) mc = std::max(mc, D.out_cap[sh]);
    s->mail_cap = D.mail_cap = mc;
    s->mailbox_bytes = (size_t)2 * D.n_shards * 64 + (size_t)2 * D.n_shards * mc * sizeof(uint4);
    {   // The mailbox is written by OTHER devices over xGMI while this device's kernels poll its flags and then read its records
        // inside one kernel (k_xchg_wait / k_deliver_mail): ordinary (coarse-grained) device memory is only coherent across
        // devices at kernel boundaries, so the flag could arrive while stale record lines sit in this device's L2.
        // Fine-grained memory is coherent at system scope for the release/acquire pair the kernels use.
        // SWIMSIM_MAILBOX=coarse|fine|uncached overrides (experiments).
      const char* mode = getenv("SWIMSIM_MAILBOX");
      void* mb = nullptr; hipError_t e = hipErrorUnknown;
      if (!mode || !strcmp(mode, "fine")) e = hipExtMallocWithFlags(&mb, s->mailbox_bytes, hipDeviceMallocFinegrained);
      else if (!strcmp(mode, "uncached")) e = hipExtMallocWithFlags(&mb, s->mailbox_bytes, hipDeviceMallocUncached);
      if (e != hipSuccess) {
        (void)hipGetLastError(); mb = nullptr;
        // coarse-grained memory is only coherent across devices at kernel boundaries: never silently — only when asked for
        if (!mode || strcmp(mode, "coarse")) { snprintf(s->err, sizeof s->err, "swim_create: no fine-grained memory for the exchange mailbox (SWIMSIM_MAILBOX=coarse to run without)"); swim_destroy(s); return SWIM_ENOMEM; }
        HIPCK(s, hipMalloc(&mb, s->mailbox_bytes)); s->mailbox_kind = "coarse";
      }
      else s->mailbox_kind = (mode && !strcmp(mode, "uncached")) ? "uncached" : "fine";
      s->allocs.push_back(mb); s->alloc_bytes.push_back(s->mailbox_bytes); s->mailbox = (uint8_t*)mb;
    }
    HIPCK(s, hipMemset(s->mailbox, 0, (size_t)2 * D.n_shards * 64));
    DALLOC(s, D.mb_tab, SW_MAX_SHARDS); DALLOC(s, D.xin_cnt, SW_MAX_SHARDS); HIPCK(s, hipMemset(D.xin_cnt, 0, SW_MAX_SHARDS * 4));
    D.xchg_timeout_ms = 2000;
    if (const char* e = getenv("SWIMSIM_XCHG_TIMEOUT_MS")) D.xchg_timeout_ms = (uint32_t)std::max(1l, strtol(e, nullptr, 10));
  }

  hipStream_t st = s->stream;
  HIPCK(s, hipMemsetAsync(D.tick, 0, 4, st));
  if (D.role_clk) { std::vector<unsigned long long> init((size_t)D.role_clk_ticks * 16 * 64); for (size_t i = 0; i < init.size(); i += 2) { init[i] = ~0ull; init[i + 1] = 0; } HIPCK(s, hipMemcpy(D.role_clk, init.data(), init.size() * 8, hipMemcpyHostToDevice)); }
  HIPCK(s, hipMemsetAsync(D.nw, 0, NT * 4, st));
  HIPCK(s, hipMemsetAsync(D.qbits, 0, (cdiv(NL, 32) + 2) * 4, st));
  HIPCK(s, hipMemsetAsync(D.exc_cnt, 0, D.R * 4, st)); HIPCK(s, hipMemsetAsync(D.exc_dirty, 0, D.R * 4, st));
  HIPCK(s, hipMemsetAsync(D.n_slots, 0, D.R * 4, st));
  HIPCK(s, hipMemsetAsync(D.out_cnt, 0, (SW_MAX_SHARDS + 1) * 4, st));
  HIPCK(s, hipMemsetAsync(D.carry_cl, 0, NB * 8, st)); HIPCK(s, hipMemsetAsync(D.att_any, 0, 4, st)); HIPCK(s, hipMemsetAsync(D.carry_stamp, 0xFF, 8, st));
  HIPCK(s, hipMemsetAsync(D.seg_cnt, 0, (size_t)D.n_seg * 4, st));
  HIPCK(s, hipMemsetAsync(D.seg_last, 0, (size_t)D.n_seg * 4, st));
  HIPCK(s, hipMemsetAsync(s->d_last_cnt, 0, SW_MAX_SHARDS * 4, st));
  HIPCK(s, hipMemsetAsync(D.pend_cnt, 0, (D.TQ + 1) * 4, st));
  HIPCK(s, hipMemsetAsync(D.pp_cnt, 0, 2 * SW_PP_LISTS * 16 * 4, st));
  HIPCK(s, hipMemsetAsync(D.ev_cnt, 0, 4, st));
  HIPCK(s, hipMemsetAsync(D.cap_cnt, 0, 4, st));
  HIPCK(s, hipMemsetAsync(D.stats, 0, (size_t)SW_STAT_COPIES * SW_STAT_STRIDE * 8, st));
  HIPCK(s, hipMemsetAsync(D.err, 0, 4, st));
  HIPCK(s, hipMemsetAsync(D.q, 0, NL * D.Q * sizeof(uint4), st));
  HIPCK(s, hipMemsetAsync(D.vt, 0xFF, NL * D.VT * sizeof(uint4), st));      // every slot free (subject = VT_EMPTY)
  HIPCK(s, hipMemsetAsync(s->fold_zero, 0, s->fold_zero_bytes, st)); HIPCK(s, hipMemsetAsync(s->fold_ones, 0xFF, s->fold_ones_bytes, st));
  if (serf) {
    HIPCK(s, hipMemsetAsync(D.evq, 0, NL * D.EQ * sizeof(uint4), st));
    HIPCK(s, hipMemsetAsync(D.ring, 0, NL * D.EB * D.EW * sizeof(uint4), st));
  }
  if (D.trace) HIPCK(s, hipMemsetAsync(D.trace, 0, NS * D.trace_ticks * 5 * 4, st));
  DALLOC(s, s->d_D, 1);                             // every pointer is set by now: publish the descriptor
  s->structural = { (void*)s->d_D, (void*)D.out_tab, (void*)D.mb_tab, (void*)s->d_scratch, (void*)s->mailbox, (void*)s->in_buf, (void*)D.tb, (void*)D.inbox_big, (void*)D.defer_rec, (void*)D.defer_l, (void*)D.big_list };   // (the tile buckets, the pooled inbox rows and their deferred records are empty between ticks)
  HIPCK(s, hipMemcpy(s->d_D, &D, sizeof D, hipMemcpyHostToDevice));
  const uint32_t n_initial = cfg->n_initial ? cfg->n_initial : D.N;
  hipLaunchKernelGGL(k_init_nodes, dim3(cdiv(NL, 256)), dim3(256), 0, st, (const SwDev*)s->d_D, n_initial);
  hipLaunchKernelGGL(k_init_base, dim3(cdiv(NT, 256)), dim3(256), 0, st, (const SwDev*)s->d_D, n_initial);
  if (D.dyn) for (uint32_t r = 0; r < D.R; r++) hipLaunchKernelGGL(k_exc_rebuild, dim3(1), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D, r);
  hipLaunchKernelGGL(k_init_slots, dim3(cdiv(NS, 256)), dim3(256), 0, st, (const SwDev*)s->d_D);
  if (D.M) hipLaunchKernelGGL(k_mass_init, dim3(cdiv((size_t)D.R * D.M, 256)), dim3(256), 0, st, (const SwDev*)s->d_D);
  HIPCK(s, hipStreamSynchronize(st));
  HIPCK(s, hipGetLastError());
  if (D.coord) { HIPCK(s, hipMemsetAsync(D.c_cnt, 0, 4, st)); hipLaunchKernelGGL(k_coord_init, dim3(cdiv(NL, 256)), dim3(256), 0, st, (const SwDev*)s->d_D); HIPCK(s, hipStreamSynchronize(st)); }
  s->pristine = cfg->loss_q32 == 0 && !D.dyn && cfg->n_shards == 1 && !D.coord;      // (coordinates move in quiet ticks too)
  *out = s;
  return SWIM_OK;
}

// ---------------------------------------------------------------------------------------------
// time
// ---------------------------------------------------------------------------------------------
// a fold tick carries four extra launches (scan + emit before k_begin, apply + count between k_deliver and k_resolve);
// the captured tick graphs never contain one (swim_step / swim_tick_end_begin launch fold ticks eagerly)
static bool fold_tick(const swim_sim* s, uint32_t tick) { return s->D.fold_period && tick && tick % s->D.fold_period == 0; }
// ...and so does a reap tick (serf's reaper: one scan of the view tables before k_begin)
static bool reap_tick(const swim_sim* s, uint32_t tick) { return s->D.reap_period && tick && tick % s->D.reap_period == 0; }
// ...and a tick in which nodes are due for serf's reconnect() (probe-interval boundaries, when the feature is on)
static bool reconnect_tick(const swim_sim* s, uint32_t tick) { return s->D.rc_period && tick && tick % s->D.P == 0; }
static bool special_tick(const swim_sim* s, uint32_t tick) { return fold_tick(s, tick) || reap_tick(s, tick) || reconnect_tick(s, tick); }
static uint32_t ticks_to_special(const swim_sim* s) {       // 0 = this tick is one; 0xFFFFFFFF = never
  uint32_t best = 0xFFFFFFFFu;
  for (uint32_t per : { s->D.fold_period, s->D.reap_period, s->D.rc_period ? s->D.P : 0u })
    if (per) best = std::min(best, (s->tick && s->tick % per == 0) ? 0u : per - s->tick % per);
  return best;
}
#define SW_PLAIN_TICK 0xFFFFFFFFu   /* launch_begin / launch_end: an ordinary tick (what the captured graphs hold) */
static void launch_begin(swim_sim* s, uint32_t tick) {
  const bool fold = tick != SW_PLAIN_TICK && fold_tick(s, tick);
  SwDev& D = s->D; hipStream_t st = s->stream;
  BeginPlan pl = s->plan;
  const size_t lds = (size_t)(D.Q + D.EQ) * SW_BLOCK * sizeof(uint4);
  const uint32_t grid = pl.nb_expire + pl.nb_pend + D.R * (pl.nb_probe + pl.nb_gossip) + pl.nb_ppreply + pl.nb_carry + pl.nb_join + D.R * pl.nb_pp;
  if (fold) {
    const size_t NL = (size_t)D.nloc * D.R, NT = (size_t)D.N * D.R;
    (void)hipMemsetAsync(s->fold_zero, 0, s->fold_zero_bytes, st); (void)hipMemsetAsync(s->fold_ones, 0xFF, s->fold_ones_bytes, st);
    hipLaunchKernelGGL(k_fold_scan, dim3(cdiv(NL, SW_BLOCK)), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D);
    if (D.M) hipLaunchKernelGGL(k_fold_scan_mass, dim3(D.R * D.M), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D);
    if (D.iq) hipLaunchKernelGGL(k_fold_scan_iq, dim3(D.R * D.MB), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D);
    if (D.iq) hipLaunchKernelGGL(k_fold_scan_slots, dim3(cdiv(NL, SW_BLOCK)), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D);
    hipLaunchKernelGGL(k_fold_emit, dim3(cdiv(NT, SW_BLOCK)), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D);
  }
  // serf's reaper AFTER the fold's census, like the checker (swim_tick_begin: fold_census, then phase_reap): a member whose last observer erases it in
  // this very tick is folded at the NEXT fold tick.  (Until round 6 the reaper ran first; the two orders only differ when a subject's last reap falls
  // on a fold tick — the dense store's reaper test met the case, the hash tables' never had.)
  if (tick != SW_PLAIN_TICK && reap_tick(s, tick)) {
    hipLaunchKernelGGL(k_reap, dim3(cdiv((size_t)D.nloc * D.R, SW_BLOCK)), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D);
    if (D.M) hipLaunchKernelGGL(k_reap_mass, dim3(D.R * D.M), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D);
  }
  if (tick != SW_PLAIN_TICK && reconnect_tick(s, tick)) {
    const uint32_t per = D.rc_period, grp = std::min(D.P, per);
    const uint32_t rc_blocks = (uint32_t)cdiv((uint64_t)cdiv(D.N, per) * grp, SW_BLOCK), rc_lanes = rc_blocks * SW_BLOCK;
    if (D.M) {      // the due nodes' columns of the dense store, a wave per (due node, chunk of rows)
      (void)hipMemsetAsync(D.rc_cnt, 0, (size_t)D.R * rc_lanes * 4, st); (void)hipMemsetAsync(D.rc_best, 0xFF, (size_t)D.R * rc_lanes * 8, st);
      const uint64_t waves = (uint64_t)rc_lanes * cdiv(D.M, SW_RC_CHUNK);
      hipLaunchKernelGGL(k_reconnect_scan, dim3((uint32_t)cdiv(waves, SW_BLOCK / 64), D.R), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D, rc_lanes);
    }
    hipLaunchKernelGGL(k_reconnect, dim3(rc_blocks, D.R), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D);
  }
  if (D.M) {   // the dense store's suspicion timers: list the due rows, then their due tiles over many waves
    hipLaunchKernelGGL(k_expire_mass_due, dim3(cdiv((size_t)D.R * D.M, SW_BLOCK)), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D);
    hipLaunchKernelGGL(k_expire_mass, dim3((uint32_t)std::min<uint64_t>(2048, cdiv((uint64_t)D.R * D.M * D.nbl, SW_BLOCK / 64))), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D);
  }
  if (D.TQ % D.P == 0) {
    // degenerate timers: a node's indirect stage and its next probe fall in the same tick, in that order
    BeginPlan a = pl, b = pl; a.roles = 0x2; b.roles = pl.roles & ~0x2u;
    { ProfScope p(s, PK_BEGIN); hipLaunchKernelGGL(s->begin_kernel, dim3(grid), dim3(SW_BLOCK), lds, st, (const SwDev*)s->d_D, a); }
    { ProfScope p(s, PK_BEGIN); hipLaunchKernelGGL(s->begin_kernel, dim3(grid), dim3(SW_BLOCK), lds, st, (const SwDev*)s->d_D, b); }
  } else {
    ProfScope p(s, PK_BEGIN);
    hipLaunchKernelGGL(s->begin_kernel, dim3(grid), dim3(SW_BLOCK), lds, st, (const SwDev*)s->d_D, pl);
  }
  if (D.iq) {    // gossip() over the queue the pair store implies: a wave per node with something queued (same blocks and segments as the gossip role)
    ProfScope p(s, PK_GOSSIP_IQ);
    const bool sf = (D.flags & SWIM_F_SERF_EVENTS) != 0, mu = D.n_shards > 1;
    void (*const gk)(const SwDev*, uint32_t) = sf ? (mu ? k_gossip_iq<true, true> : k_gossip_iq<true, false>) : (mu ? k_gossip_iq<false, true> : k_gossip_iq<false, false>);
    hipLaunchKernelGGL(gk, dim3(D.R * pl.nb_gossip), dim3(SW_IQ_GTHREADS), 0, st, (const SwDev*)s->d_D, pl.nb_gossip);
  }
  if (D.M) hipLaunchKernelGGL(k_send_mass, dim3(1024), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D);   // the dense store's part of the state exchanges listed above
  if (D.coord) {      // serf's ping delegate: the probers k_begin listed update their coordinates (from everybody's as of the start of the tick)
    hipLaunchKernelGGL(k_coord_update, dim3(cdiv(D.c_cap, SW_BLOCK)), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D);
    hipLaunchKernelGGL(k_coord_commit, dim3(cdiv(D.c_cap, SW_BLOCK)), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D);
  }
}
// k_resolve's dynamic LDS: the lanes' memberlist queues ([Q][threads] entries of 16 bytes; of 8 — {subject, meta} — on a handle with the
// dense pair store) and the meta words of their user-event queues
static size_t resolve_lds_bytes(const SwDev& D) {
  return (size_t)D.Q * SW_RES_THREADS * ((D.M && SW_SPLITQ) ? sizeof(uint2) : sizeof(uint4)) + (size_t)D.EQ * SW_RES_THREADS * 4;
}
static void launch_end(swim_sim* s, uint32_t tick) {
  const bool fold = tick != SW_PLAIN_TICK && fold_tick(s, tick);
  SwDev& D = s->D; hipStream_t st = s->stream;
  const size_t NL = (size_t)D.nloc * D.R;
  {
    void (*const deliver_kernel)(const SwDev*) = D.M ? k_deliver<true, false> : D.tb_on ? k_deliver<false, true> : k_deliver<false, false>;
    const uint32_t dgrid = (D.tb_on ? D.tb_T + (D.tb_carry ? 0u : D.n_seg) : D.n_seg) + 32;
    ProfScope p(s, PK_DELIVER); hipLaunchKernelGGL(deliver_kernel, dim3(dgrid), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D);
  }
  if (s->in_count) {
    ProfScope p(s, PK_DELIVER);
    hipLaunchKernelGGL(k_deliver_list, dim3(std::min<uint32_t>(cdiv(s->in_count, SW_BLOCK * 4), 2048)), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D,
                       (const uint4*)s->in_buf, s->in_count);
  }
  if (fold) {
    hipLaunchKernelGGL(k_fold_decide, dim3(cdiv((size_t)D.N * D.R, SW_BLOCK)), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D);
    hipLaunchKernelGGL(k_fold_apply, dim3(cdiv(NL, SW_BLOCK)), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D);
    if (D.M) hipLaunchKernelGGL(k_fold_apply_mass, dim3(D.R * D.M), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D);
  }
  if (D.PB) {   // pooled inbox rows: the nodes whose arrivals passed their own row get a big one, the deferred records are filed
    hipLaunchKernelGGL(k_inbox_claim, dim3(512), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D);
    hipLaunchKernelGGL(k_inbox_file, dim3(1024), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D);
  }
  if (D.bigsort_cap) {   // inboxes of thousands of messages (a state exchange during a mass event) are sorted by a workgroup each, in LDS
    uint32_t P = SW_BIGSORT_MIN; while (P < D.bigsort_cap) P <<= 1;
    hipLaunchKernelGGL(k_inbox_sort_med, dim3(cdiv(NL, SW_BLOCK)), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D);
    hipLaunchKernelGGL(k_inbox_sort, dim3(cdiv(NL, SW_BLOCK)), dim3(SW_BLOCK), (size_t)P * 12, st, (const SwDev*)s->d_D, P);
    if (D.C > D.bigsort_cap) hipLaunchKernelGGL(k_inbox_sort_huge, dim3(cdiv(NL, SW_BLOCK)), dim3(SW_BLOCK), (size_t)SW_BIGSORT_MAX * 12, st, (const SwDev*)s->d_D);   // (inbox_cap beyond what LDS sorts: config #4's recovery)
  }
  const bool serf_k = (D.flags & SWIM_F_SERF_EVENTS) != 0;
  if (D.iq && (D.flags & SWIM_F_PIGGYBACK)) {   // the tick's piggy-back orders, served from the nodes' columns before anything is merged
    ProfScope p(s, PK_PIGGY_IQ);
    if (serf_k) hipLaunchKernelGGL(k_piggy_iq<true>, dim3(2048), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D);
    else hipLaunchKernelGGL(k_piggy_iq<false>, dim3(2048), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D);
  }
  void (*const resolve_kernel)(const SwDev*) =
      D.dyn ? (D.M ? (serf_k ? k_resolve<true, true, true> : k_resolve<true, false, true>) : (serf_k ? k_resolve<false, true, true> : k_resolve<false, false, true>))
            : (D.M ? (serf_k ? k_resolve<true, true, false> : k_resolve<true, false, false>) : (serf_k ? k_resolve<false, true, false> : k_resolve<false, false, false>));
  { ProfScope p(s, PK_RESOLVE); hipLaunchKernelGGL(resolve_kernel, dim3((uint32_t)cdiv(NL, resolve_tile(D))), dim3(SW_RES_THREADS), resolve_lds_bytes(D), st, (const SwDev*)s->d_D); }
  // blocks per watch slot.  Measured (profiles/): a quiet tick costs the same with 1024 or 8192 blocks that
  // just leave, while a dirty slot is scanned markedly faster by 64 blocks than by 16 — so: many.
  const uint32_t xb = std::max(1u, std::min<uint32_t>(cdiv(D.nloc, SW_BLOCK * 4), 64));
  { ProfScope p(s, PK_FINISH); hipLaunchKernelGGL(k_census_finish, dim3(xb, D.R * D.S), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D, s->d_last_cnt); }
  if (fold) hipLaunchKernelGGL(k_exc_rebuild_folded, dim3(D.R), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D);
  s->in_count = 0;
}
static void advance(swim_sim* s, uint32_t n) {
  for (uint32_t k = 0; k < n; k++) {
    s->tick++; s->ticks_run++;
    if (s->tick % s->d.gossip_period == 0) s->rounds_run++;
  }
}

static int check_device_errors(swim_sim* s) {
  uint32_t e = 0;
  HIPCK(s, hipMemcpyAsync(&e, s->D.err, 4, hipMemcpyDeviceToHost, s->stream));
  HIPCK(s, hipStreamSynchronize(s->stream));
  HIPCK(s, hipGetLastError());
  if (e & SW_ERR_XCHG_TIMEOUT) {
    snprintf(s->err, sizeof s->err, "a source shard did not deliver this tick's records (swim_xchg_step: no flag within %u ms; swim_frame_deliver: a frame "
             "whose header is not this tick's) - is every shard stepping?", s->D.xchg_timeout_ms);
    return SWIM_ESTATE;
  }
  if (e) {
    snprintf(s->err, sizeof s->err, "bounded structure overflowed:%s%s%s%s%s%s%s",
             e & SW_ERR_EDGE_OVF ? " edge-list" : "", e & SW_ERR_INBOX_OVF ? " inbox" : "",
             e & SW_ERR_SUBJ_OVF ? " subject-slots" : "", e & SW_ERR_CTRL_OVF ? " slot-requests" : "",
             e & SW_ERR_EVENT_OVF ? " event-ring" : "", e & SW_ERR_PEND_OVF ? " pending-probes" : "",
             e & SW_ERR_CARRY_OVF ? " piggy-back-carry" : (e & SW_ERR_VIEW_CORRUPT ? " view-table(corrupt)" : (e & SW_ERR_MASS_RANGE ? " dense-store-field-range" : (e & SW_ERR_ORDER_OVF ? " piggy-back-orders" : ""))));
    return SWIM_EOVERFLOW;
  }
  return SWIM_OK;
}

extern "C" int swim_tick_begin(swim_sim* s) {
  if (!s) return SWIM_EINVAL;
  if (s->in_tick) return SWIM_ESTATE;
  launch_begin(s, s->tick);
  s->in_tick = true; s->out_counts_valid = false; s->in_count = 0;
  return SWIM_OK;
}
extern "C" int swim_outbound(swim_sim* s, uint32_t shard, const swim_edge** ptr, uint32_t* count) {
  if (!s || !ptr || !count || shard >= s->cfg.n_shards) return SWIM_EINVAL;
  if (!s->in_tick) return SWIM_ESTATE;
  if (!s->out_counts_valid) {
    HIPCK(s, hipMemcpyAsync(s->out_counts, s->D.out_cnt, (SW_MAX_SHARDS + 1) * 4, hipMemcpyDeviceToHost, s->stream));
    HIPCK(s, hipStreamSynchronize(s->stream));
    s->out_counts_valid = true;
  }
  *ptr = (const swim_edge*)s->D.out[shard];
  // the shard's own records never cross the wire (they sit in the gossip segments and its misc list)
  *count = shard == s->D.rank ? 0 : std::min(s->out_counts[shard], s->D.out_cap[shard]);
  return SWIM_OK;
}
extern "C" int swim_stream(swim_sim* s, void** st) {
  if (!s || !st) return SWIM_EINVAL;
  *st = (void*)s->stream;
  return SWIM_OK;
}
extern "C" int swim_outbound_raw(swim_sim* s, uint32_t shard, const swim_edge** seg, const uint32_t** cnt) {
  if (!s || shard >= s->cfg.n_shards) return SWIM_EINVAL;
  if (seg) *seg = (const swim_edge*)s->D.out[shard];
  if (cnt) *cnt = s->D.out_cnt;
  return SWIM_OK;
}
extern "C" int swim_peer_activity(swim_sim* s, int active) {
  if (!s) return SWIM_EINVAL;
  const uint32_t v = active ? 1u : 0u;             // read by the next k_begin, in stream order
  if (v != s->peer_act_host) { HIPCK(s, hipMemsetD32Async((hipDeviceptr_t)s->D.peer_act, (int)v, 1, s->stream)); s->peer_act_host = v; }
  return SWIM_OK;
}
extern "C" int swim_activity(swim_sim* s, int* active) {
  if (!s || !active) return SWIM_EINVAL;
  if (!s->in_tick) return SWIM_ESTATE;
  const swim_edge* p; uint32_t c;
  int rc = swim_outbound(s, 0, &p, &c);             // (re)uses the tick's one copy of the counters
  if (rc) return rc;
  uint32_t any = s->out_counts[s->D.n_shards];
  for (uint32_t sh = 0; sh < s->D.n_shards; sh++) any |= s->out_counts[sh];
  *active = any != 0;
  return SWIM_OK;
}
// a host-side stimulus may fill queues behind the back of the activity word: raise it, and ignore the
// caller's hint for the next tick (stimulus is replicated on every shard, so every shard does)
static void touched(swim_sim* s) {
  s->pristine = false;
  if (s->D.n_shards > 1) (void)hipMemsetD32Async((hipDeviceptr_t)s->D.act, 1, 1, s->stream);
  if (s->peer_act_host != 1) { (void)hipMemsetD32Async((hipDeviceptr_t)s->D.peer_act, 1, 1, s->stream); s->peer_act_host = 1; }
}
extern "C" uint32_t swim_outbound_capacity(swim_sim* s, uint32_t shard) {
  return (s && shard < s->cfg.n_shards) ? s->D.out_cap[shard] : 0;
}
extern "C" int swim_inbound(swim_sim* s, const swim_edge* ptr, uint32_t count) {
  if (!s || (!ptr && count)) return SWIM_EINVAL;
  if (!s->in_tick) return SWIM_ESTATE;
  if (!count) return SWIM_OK;
  if ((uint64_t)s->in_count + count > s->in_cap) { snprintf(s->err, sizeof s->err, "inbound staging full"); return SWIM_EOVERFLOW; }
  // asynchronous on the simulator's stream; the caller keeps the source alive (see swimsim.h)
  HIPCK(s, hipMemcpyAsync(s->in_buf + s->in_count, ptr, (size_t)count * sizeof(uint4), hipMemcpyDeviceToDevice, s->stream));
  s->in_count += count;
  return SWIM_OK;
}
// ---- framed exchange (swimsim.h): one equal-split collective per tick, the counts stay on the device ----------------------
extern "C" uint32_t swim_frame_records(swim_sim* s) {
  if (!s) return 0;
  uint32_t mc = 1;                                   // (a lone shard: a header and one record nobody fills)
  for (uint32_t sh = 0; sh < s->D.n_shards; sh++) if (sh != s->D.rank) mc = std::max(mc, s->D.out_cap[sh]);
  return mc + 1;
}
static int frame_args(swim_sim* s, const void* p, uint32_t F) {
  if (!s || !p || F < 2) return SWIM_EINVAL;
  if (!s->in_tick) return SWIM_ESTATE;
  return SWIM_OK;
}
static int frame_pack(swim_sim* s, swim_edge* send, uint32_t F, uint32_t fill) {
  int rc = frame_args(s, send, F);
  if (rc) return rc;
  const uint32_t xb = std::max(1u, std::min<uint32_t>(cdiv(std::min(F - 1, swim_frame_records(s) - 1), SW_BLOCK * 8), 64));
  hipLaunchKernelGGL(k_frame_pack, dim3(xb, s->D.n_shards), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, (uint4*)send, F, fill);
  return SWIM_OK;
}
extern "C" int swim_frame_pack(swim_sim* s, swim_edge* send, uint32_t F) { return frame_pack(s, send, F, 0u); }
extern "C" int swim_frame_pack_fill(swim_sim* s, swim_edge* send, uint32_t F) { return frame_pack(s, send, F, 1u); }
extern "C" int swim_frame_deliver(swim_sim* s, const swim_edge* recv, uint32_t F) {
  int rc = frame_args(s, recv, F);
  if (rc) return rc;
  const uint32_t xb = std::max(1u, std::min<uint32_t>(cdiv(std::min(F - 1, swim_frame_records(s) - 1), SW_BLOCK * 8), 64));
  { ProfScope p(s, PK_DELIVER); hipLaunchKernelGGL(k_frame_deliver, dim3(xb, s->D.n_shards), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, (const uint4*)recv, F); }
  s->peer_act_host = 2;          // the device has set the hint itself: the host's copy of it is stale (touched() raises it again)
  return SWIM_OK;
}

// sharded runs: end tick t and begin tick t+1 in one go.  When nothing came in from other shards (every quiet
// tick) the six launches are replayed from one captured graph instead of being issued one by one — the host
// sits on the critical path of a sharded tick (it must read the exchanged counts), so its launch time shows.
extern "C" int swim_tick_end_begin(swim_sim* s) {
  if (!s) return SWIM_EINVAL;
  if (!s->in_tick) return SWIM_ESTATE;
  const bool plain = !special_tick(s, s->tick) && !special_tick(s, s->tick + 1);
  if (s->in_count == 0 && s->use_graphs && !s->profiling && plain) {
    if (!s->graph_end_begin) {
      hipGraph_t g = nullptr;
      HIPCK(s, hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
      launch_end(s, SW_PLAIN_TICK); launch_begin(s, SW_PLAIN_TICK);
      HIPCK(s, hipStreamEndCapture(s->stream, &g));
      hipError_t e = hipGraphInstantiate(&s->graph_end_begin, g, nullptr, nullptr, 0);
      (void)hipGraphDestroy(g);
      if (e != hipSuccess) { snprintf(s->err, sizeof s->err, "hipGraphInstantiate: %s", hipGetErrorString(e)); return SWIM_ENODEV; }
    }
    HIPCK(s, hipGraphLaunch(s->graph_end_begin, s->stream));
  } else { launch_end(s, s->tick); launch_begin(s, s->tick + 1); }
  advance(s, 1);
  s->out_counts_valid = false; s->in_count = 0;
  return SWIM_OK;
}
// ---------------------------------------------------------------------------------------------
// device-driven exchange (swimsim.h): peer-mapped mailboxes, no host round trip
// ---------------------------------------------------------------------------------------------
extern "C" int swim_xchg_export(swim_sim* s, swim_xchg_handle* out) {
  if (!s || !out) return SWIM_EINVAL;
  if (s->D.n_shards < 2 || !s->mailbox) return SWIM_ESTATE;
  XchgHandle h; memset(&h, 0, sizeof h);
  HIPCK(s, hipIpcGetMemHandle(&h.ipc, s->mailbox));
  h.ptr = (uint64_t)(uintptr_t)s->mailbox; h.nonce = process_nonce(); h.rank = s->D.rank; h.n_shards = s->D.n_shards;
  h.mail_cap = s->mail_cap; h.magic = kXchgMagic;
  { std::lock_guard<std::mutex> g(g_xchg_mu); if (std::find(g_xchg_exported.begin(), g_xchg_exported.end(), h.ptr) == g_xchg_exported.end()) g_xchg_exported.push_back(h.ptr); }
  memset(out, 0, sizeof *out); memcpy(out->bytes, &h, sizeof h);
  return SWIM_OK;
}
extern "C" int swim_xchg_connect(swim_sim* s, const swim_xchg_handle* all) {
  if (!s || !all) return SWIM_EINVAL;
  if (s->D.n_shards < 2 || !s->mailbox || s->in_tick) return SWIM_ESTATE;
  uint8_t* tab[SW_MAX_SHARDS] = { nullptr };
  for (void* p : s->ipc_opened) (void)hipIpcCloseMemHandle(p);      // a second connect: the earlier mappings go
  s->ipc_opened.clear(); s->xchg_connected = false;
  if (s->graph_xchg) { (void)hipGraphExecDestroy(s->graph_xchg); s->graph_xchg = nullptr; }
  for (uint32_t sh = 0; sh < s->D.n_shards; sh++) {
    if (sh == s->D.rank) { tab[sh] = s->mailbox; continue; }
    XchgHandle h; memcpy(&h, all[sh].bytes, sizeof h);
    if (h.magic != kXchgMagic || h.rank != sh || h.n_shards != s->D.n_shards || h.mail_cap != s->mail_cap) {
      snprintf(s->err, sizeof s->err, "swim_xchg_connect: handle %u does not belong to this population", sh); return SWIM_EINVAL;
    }
    bool mine = false;
    if (h.nonce == process_nonce()) { std::lock_guard<std::mutex> g(g_xchg_mu); mine = std::find(g_xchg_exported.begin(), g_xchg_exported.end(), h.ptr) != g_xchg_exported.end(); }
    if (mine) tab[sh] = (uint8_t*)(uintptr_t)h.ptr;      // exported by this very process: the pointer itself
    else {
      void* p = nullptr;
      HIPCK(s, hipIpcOpenMemHandle(&p, h.ipc, hipIpcMemLazyEnablePeerAccess));
      s->ipc_opened.push_back(p); tab[sh] = (uint8_t*)p;
    }
  }
  HIPCK(s, hipMemcpy(s->D.mb_tab, tab, sizeof tab, hipMemcpyHostToDevice));
  s->xchg_connected = true;
  return SWIM_OK;
}
static void launch_xchg(swim_sim* s) {
  SwDev& D = s->D; hipStream_t st = s->stream;
  const uint32_t xb = std::max(1u, std::min<uint32_t>(cdiv(s->mail_cap, SW_BLOCK * 8), 64));
  hipLaunchKernelGGL(k_xchg_copy, dim3(xb, D.n_shards), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D);
  hipLaunchKernelGGL(k_xchg_signal, dim3(1), dim3(64), 0, st, (const SwDev*)s->d_D);
  hipLaunchKernelGGL(k_xchg_wait, dim3(1), dim3(64), 0, st, (const SwDev*)s->d_D);
  { ProfScope p(s, PK_DELIVER); hipLaunchKernelGGL(k_deliver_mail, dim3(xb, D.n_shards), dim3(SW_BLOCK), 0, st, (const SwDev*)s->d_D); }
}
extern "C" int swim_xchg_step(swim_sim* s, uint32_t n) {
  if (!s) return SWIM_EINVAL;
  if (!s->xchg_connected || s->in_tick) return SWIM_ESTATE;
  const bool use_graph = !s->profiling && s->use_graphs;
  for (uint32_t i = 0; i < n; i++) {
    if (use_graph && !special_tick(s, s->tick)) {
      if (!s->graph_xchg) {
        hipGraph_t g = nullptr;
        HIPCK(s, hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
        launch_begin(s, SW_PLAIN_TICK); launch_xchg(s); launch_end(s, SW_PLAIN_TICK);
        HIPCK(s, hipStreamEndCapture(s->stream, &g));
        hipError_t e = hipGraphInstantiate(&s->graph_xchg, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) { snprintf(s->err, sizeof s->err, "hipGraphInstantiate: %s", hipGetErrorString(e)); return SWIM_ENODEV; }
      }
      HIPCK(s, hipGraphLaunch(s->graph_xchg, s->stream));
    } else { launch_begin(s, s->tick); launch_xchg(s); launch_end(s, s->tick); }
    advance(s, 1);
  }
  if (n) s->peer_act_host = 2;   // k_xchg_wait sets the hint on the device: the host's copy is stale (touched() raises the word again after a stimulus)
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { snprintf(s->err, sizeof s->err, "launch failed: %s", hipGetErrorString(e)); return SWIM_ENODEV; }
  return SWIM_OK;
}
extern "C" int swim_tick_end(swim_sim* s) {
  if (!s) return SWIM_EINVAL;
  if (!s->in_tick) return SWIM_ESTATE;
  launch_end(s, s->tick);
  s->in_tick = false; advance(s, 1);
  return SWIM_OK;
}

// The tick sequence is captured once into hipGraphs (1 tick and SW_GRAPH_TICKS ticks); kernels read
// the clock from device memory, so a replay is valid for any tick.  Replaying removes the per-launch
// host cost (~3 us each) that would otherwise bound quiescent ticks.
static int build_graph(swim_sim* s, int which, uint32_t ticks) {
  hipGraph_t g = nullptr;
  HIPCK(s, hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
  for (uint32_t i = 0; i < ticks; i++) { launch_begin(s, SW_PLAIN_TICK); launch_end(s, SW_PLAIN_TICK); }
  HIPCK(s, hipStreamEndCapture(s->stream, &g));
  hipError_t e = hipGraphInstantiate(&s->graph_exec[which], g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) { snprintf(s->err, sizeof s->err, "hipGraphInstantiate: %s", hipGetErrorString(e)); return SWIM_ENODEV; }
  return SWIM_OK;
}
extern "C" int swim_step(swim_sim* s, uint32_t n) {
  if (!s) return SWIM_EINVAL;
  if (s->cfg.n_shards != 1 || s->in_tick) return SWIM_ESTATE;
  const bool use_graph = !s->profiling && s->use_graphs;
  if (use_graph && !s->graph_exec[0]) {
    int rc = build_graph(s, 0, 1);
    if (!rc) rc = build_graph(s, 1, SW_GRAPH_TICKS);
    if (!rc) rc = build_graph(s, 2, SW_GRAPH_TICKS_MID);
    if (rc) return rc;
  }
  uint32_t i = 0;
  // A pristine population (see k_quiet) advances all but the last tick of the call in one launch.
  if (s->pristine && use_graph && n >= 8 && !s->D.loss_q32 && !s->D.dyn && !(s->D.TQ % s->D.P == 0)) {
    const uint32_t K = n - 1;
    const size_t NL = (size_t)s->D.nloc * s->D.R;
    hipLaunchKernelGGL(k_quiet, dim3(cdiv(NL, SW_BLOCK)), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, K);
    hipLaunchKernelGGL(k_quiet_advance, dim3(1), dim3(64), 0, s->stream, (const SwDev*)s->d_D, K);
    advance(s, K);
    i = K;
  }
  while (i < n) {
    uint32_t adv = 1;
    // ticks until the next fold / reap tick (such a tick is launched eagerly, with its extra kernels)
    const uint32_t to_special = ticks_to_special(s);
    if (to_special == 0) { launch_begin(s, s->tick); launch_end(s, s->tick); }
    else if (use_graph && n - i >= SW_GRAPH_TICKS && to_special >= SW_GRAPH_TICKS) { HIPCK(s, hipGraphLaunch(s->graph_exec[1], s->stream)); adv = SW_GRAPH_TICKS; }
    else if (use_graph && n - i >= SW_GRAPH_TICKS_MID && to_special >= SW_GRAPH_TICKS_MID) { HIPCK(s, hipGraphLaunch(s->graph_exec[2], s->stream)); adv = SW_GRAPH_TICKS_MID; }
    else if (use_graph) HIPCK(s, hipGraphLaunch(s->graph_exec[0], s->stream));
    else { launch_begin(s, SW_PLAIN_TICK); launch_end(s, SW_PLAIN_TICK); }
    advance(s, adv);
    i += adv;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { snprintf(s->err, sizeof s->err, "launch failed: %s", hipGetErrorString(e)); return SWIM_ENODEV; }
  return SWIM_OK;
}
extern "C" int swim_sync(swim_sim* s) {
  if (!s) return SWIM_EINVAL;
  return check_device_errors(s);
}
extern "C" int swim_now(swim_sim* s, uint32_t* tick, uint32_t* ms) {
  if (!s) return SWIM_EINVAL;
  if (tick) *tick = s->tick;
  if (ms) *ms = s->tick * s->d.quantum_ms;
  return SWIM_OK;
}

// ---------------------------------------------------------------------------------------------
// stimulus
// ---------------------------------------------------------------------------------------------
static int check_ids(swim_sim* s, uint32_t r, const uint32_t* ids, size_t n) {
  if (!s || (!ids && n)) return SWIM_EINVAL;
  if (s->in_tick) return SWIM_ESTATE;
  if (r >= s->D.R) return SWIM_ERANGE;
  for (size_t i = 0; i < n; i++) if (ids[i] >= s->D.N) return SWIM_ERANGE;
  return SWIM_OK;
}
// watch slots for the named nodes (while slots remain), then their highest-incarnation seeds
static int watch_ids(swim_sim* s, uint32_t r, const uint32_t* d_ids, uint32_t n) {
  const SwDev& D = s->D;
  hipLaunchKernelGGL(k_inject_alloc, dim3(1), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, r, d_ids, n, s->d_fresh);
  const uint32_t xb = std::max(1u, std::min<uint32_t>(cdiv(D.nloc, SW_BLOCK * 4), 64));
  hipLaunchKernelGGL(k_watch_seed, dim3(xb, std::min<uint32_t>(n, 1023)), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, (const uint32_t*)s->d_fresh);
  return SWIM_OK;
}
static int inject(swim_sim* s, int op, uint32_t r, const uint32_t* ids, size_t n) {
  int rc = check_ids(s, r, ids, n);
  if (rc || !n) return rc;
  touched(s);
  const size_t chunk = s->scratch_bytes / 4;          // ids travel through the scratch buffer, a chunk at a time
  for (size_t off = 0; off < n; off += chunk) {
    const uint32_t c = (uint32_t)std::min(chunk, n - off);
    HIPCK(s, hipMemcpyAsync(s->d_scratch, ids + off, (size_t)c * 4, hipMemcpyHostToDevice, s->stream));
    watch_ids(s, r, s->d_scratch, c);
    if (s->D.M) hipLaunchKernelGGL(k_mass_alloc, dim3(1), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, r, (const uint32_t*)s->d_scratch, c);
    hipLaunchKernelGGL(k_inject, dim3(cdiv(c, SW_BLOCK)), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, op, r, (const uint32_t*)s->d_scratch, c);
    if (s->D.M && op == INJ_REVIVE) {
      hipLaunchKernelGGL(k_mass_rearm, dim3(cdiv(s->D.M, SW_BLOCK)), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, r);
      HIPCK(s, hipMemsetAsync(s->D.m_rev, 0, (size_t)cdiv(s->D.nbl, 32) * 4, s->stream));
    }
    HIPCK(s, hipStreamSynchronize(s->stream));          // ids is caller memory; the scratch buffer is reused
  }
  hipLaunchKernelGGL(k_exc_rebuild, dim3(1), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, r);   // node words changed
  HIPCK(s, hipStreamSynchronize(s->stream));
  return SWIM_OK;
}
extern "C" int swim_inject_kill(swim_sim* s, uint32_t r, const uint32_t* ids, size_t n) { return inject(s, INJ_KILL, r, ids, n); }
extern "C" int swim_inject_revive(swim_sim* s, uint32_t r, const uint32_t* ids, size_t n) { return inject(s, INJ_REVIVE, r, ids, n); }
extern "C" int swim_inject_leave(swim_sim* s, uint32_t r, const uint32_t* ids, size_t n) { return inject(s, INJ_LEAVE, r, ids, n); }
extern "C" int swim_inject_update(swim_sim* s, uint32_t r, const uint32_t* ids, size_t n) { return inject(s, INJ_UPDATE, r, ids, n); }
extern "C" int swim_force_leave(swim_sim* s, uint32_t r, uint32_t origin, uint32_t node, int prune, uint64_t* lt) {
  if (!s) return SWIM_EINVAL;
  if (s->in_tick || !(s->cfg.flags & SWIM_F_SERF_EVENTS)) return SWIM_ESTATE;
  if (r >= s->D.R || origin >= s->D.N || node >= s->D.N) return SWIM_ERANGE;
  touched(s);
  hipLaunchKernelGGL(k_force_leave, dim3(1), dim3(64), 0, s->stream, (const SwDev*)s->d_D, r, origin,
                     SWIM_INTENT_LEAVE | (prune ? SWIM_INTENT_PRUNE : 0u) | node, s->d_scratch);
  uint32_t v = SWIM_NONE;
  HIPCK(s, hipMemcpyAsync(&v, s->d_scratch, 4, hipMemcpyDeviceToHost, s->stream));
  HIPCK(s, hipStreamSynchronize(s->stream));
  if (lt) *lt = v == SWIM_NONE ? UINT64_MAX : (uint64_t)v;     // (the device clock is 32 bits wide; nothing was stamped: all ones)
  return SWIM_OK;
}
extern "C" int swim_inject_join(swim_sim* s, uint32_t r, const uint32_t* ids, size_t n, uint32_t via) {
  int rc = check_ids(s, r, ids, n);
  if (rc) return rc;
  if (via >= s->D.N) return SWIM_ERANGE;
  if (!n) return SWIM_OK;
  if (!s->D.join_cnt) return SWIM_ESTATE;
  touched(s);
  const size_t chunk = std::min<size_t>(s->scratch_bytes / 4, s->D.join_cap / 2);
  for (size_t off = 0; off < n; off += chunk) {
    const uint32_t c = (uint32_t)std::min(chunk, n - off);
    HIPCK(s, hipMemcpyAsync(s->d_scratch, ids + off, (size_t)c * 4, hipMemcpyHostToDevice, s->stream));
    if (s->D.M) hipLaunchKernelGGL(k_mass_alloc, dim3(1), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, r, (const uint32_t*)s->d_scratch, c);
    hipLaunchKernelGGL(k_inject_join, dim3(cdiv(c, SW_BLOCK)), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, r, (const uint32_t*)s->d_scratch, c, via);
    HIPCK(s, hipStreamSynchronize(s->stream));
  }
  hipLaunchKernelGGL(k_exc_rebuild, dim3(1), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, r);
  HIPCK(s, hipStreamSynchronize(s->stream));
  return SWIM_OK;
}
extern "C" int swim_watch(swim_sim* s, uint32_t r, uint32_t x) {
  if (!s) return SWIM_EINVAL;
  if (s->in_tick) return SWIM_ESTATE;
  if (r >= s->D.R || x >= s->D.N) return SWIM_ERANGE;
  s->pristine = false;             // a watch slot wants its trace row every tick
  HIPCK(s, hipMemcpyAsync(s->d_scratch, &x, 4, hipMemcpyHostToDevice, s->stream));
  watch_ids(s, r, s->d_scratch, 1);
  hipLaunchKernelGGL(k_exc_rebuild, dim3(1), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, r);
  uint32_t w = 0;
  HIPCK(s, hipMemcpyAsync(&w, s->D.nw + (size_t)r * s->D.N + x, 4, hipMemcpyDeviceToHost, s->stream));
  HIPCK(s, hipStreamSynchronize(s->stream));
  return NW_HAS_SLOT(w) ? SWIM_OK : SWIM_EOVERFLOW;
}
extern "C" int swim_inject_partition(swim_sim* s, uint32_t r, const uint8_t* g) {
  if (!s || !g) return SWIM_EINVAL;
  if (s->in_tick) return SWIM_ESTATE;
  if (r >= s->D.R) return SWIM_ERANGE;
  for (uint32_t i = 0; i < s->D.N; i++) if (g[i] > 127) return SWIM_ERANGE;   // 7 bits of the node word
  touched(s);
  for (size_t off = 0; off < s->D.N; off += s->scratch_bytes) {                // the mask travels in chunks
    const uint32_t c = (uint32_t)std::min<size_t>(s->scratch_bytes, s->D.N - off);
    HIPCK(s, hipMemcpyAsync(s->d_scratch, g + off, c, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(k_set_partition, dim3(cdiv(c, 256)), dim3(256), 0, s->stream, (const SwDev*)s->d_D, r, (const uint8_t*)s->d_scratch, (uint32_t)off, c);
    HIPCK(s, hipStreamSynchronize(s->stream));
  }
  if (s->D.M) {   // the dense store: the nodes outside the largest group are about to be suspected by everybody in it
    uint32_t cnt[128] = { 0 }, big = 0;
    for (uint32_t i = 0; i < s->D.N; i++) cnt[g[i]]++;
    for (uint32_t k = 1; k < 128; k++) if (cnt[k] > cnt[big]) big = k;
    std::vector<uint32_t> ids;
    for (uint32_t i = 0; i < s->D.N; i++) if (g[i] != big) ids.push_back(i);
    if (ids.size() && ids.size() < s->D.M)      // rows to spare: the largest group's nodes too (what the minority sides will think of them)
      for (uint32_t i = 0; i < s->D.N && ids.size() < (size_t)s->D.M + 1024; i++) if (g[i] == big) ids.push_back(i);
    const size_t chunk = s->scratch_bytes / 4;
    for (size_t off = 0; off < ids.size(); off += chunk) {
      const uint32_t c = (uint32_t)std::min(chunk, ids.size() - off);
      HIPCK(s, hipMemcpyAsync(s->d_scratch, ids.data() + off, (size_t)c * 4, hipMemcpyHostToDevice, s->stream));
      hipLaunchKernelGGL(k_mass_alloc, dim3(1), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, r, (const uint32_t*)s->d_scratch, c);
      HIPCK(s, hipStreamSynchronize(s->stream));
    }
  }
  hipLaunchKernelGGL(k_exc_rebuild, dim3(1), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, r);
  HIPCK(s, hipStreamSynchronize(s->stream));
  return SWIM_OK;
}
extern "C" int swim_set_tcp_class(swim_sim* s, uint32_t r, const uint32_t* ids, size_t n, uint8_t cls) {
  int rc = check_ids(s, r, ids, n);
  if (!rc && cls > SWIM_TCP_CLASS_MAX) rc = SWIM_ERANGE;
  if (rc || !n) return rc;
  touched(s);
  const size_t chunk = s->scratch_bytes / 4;
  for (size_t off = 0; off < n; off += chunk) {
    const uint32_t c = (uint32_t)std::min(chunk, n - off);
    HIPCK(s, hipMemcpyAsync(s->d_scratch, ids + off, (size_t)c * 4, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(k_set_tcp_class, dim3(cdiv(c, SW_BLOCK)), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, r, (const uint32_t*)s->d_scratch, c, (uint32_t)cls);
    HIPCK(s, hipStreamSynchronize(s->stream));
  }
  hipLaunchKernelGGL(k_exc_rebuild, dim3(1), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, r);   // node words changed
  HIPCK(s, hipStreamSynchronize(s->stream));
  return SWIM_OK;
}
extern "C" int swim_set_loss(swim_sim* s, uint32_t q) {
  if (!s) return SWIM_EINVAL;
  if (q) s->pristine = false;
  if (s->D.loss_q32 != q) {       // the kernels read the descriptor from device memory: update it in place
    (void)hipStreamSynchronize(s->stream);
    s->D.loss_q32 = q;
    HIPCK(s, hipMemcpy(s->d_D, &s->D, sizeof s->D, hipMemcpyHostToDevice));
  }
  return SWIM_OK;
}
extern "C" int swim_user_event(swim_sim* s, uint32_t r, uint32_t origin, uint32_t id, uint64_t* lt) {
  if (!s) return SWIM_EINVAL;
  if (s->in_tick || !(s->cfg.flags & SWIM_F_SERF_EVENTS)) return SWIM_ESTATE;
  if (r >= s->D.R || origin >= s->D.N || id > SWIM_EVENT_ID_MAX) return SWIM_ERANGE;   // bits 31-30 mark serf's intents, never a user event
  touched(s);
  hipLaunchKernelGGL(k_user_event, dim3(1), dim3(64), 0, s->stream, (const SwDev*)s->d_D, r, origin, id, s->d_scratch);
  uint32_t v = SWIM_NONE;
  HIPCK(s, hipMemcpyAsync(&v, s->d_scratch, 4, hipMemcpyDeviceToHost, s->stream));
  HIPCK(s, hipStreamSynchronize(s->stream));
  if (lt) *lt = v == SWIM_NONE ? UINT64_MAX : (uint64_t)v;     // (the device clock is 32 bits wide; nothing was stamped: all ones)
  return SWIM_OK;
}

// ---------------------------------------------------------------------------------------------
// observation
// ---------------------------------------------------------------------------------------------
template <typename T>
static int d2h(swim_sim* s, T* dst, const T* src, size_t n) {
  HIPCK(s, hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDeviceToHost, s->stream));
  HIPCK(s, hipStreamSynchronize(s->stream));
  return SWIM_OK;
}
static uint8_t status_of(uint32_t st) {
  return st == SWIM_STATE_DEAD ? SWIM_MEMBER_FAILED : st == SWIM_STATE_LEFT ? SWIM_MEMBER_LEFT : SWIM_MEMBER_ALIVE;
}
static bool is_local(const swim_sim* s, uint32_t i) { return i >= s->D.i0 && i < s->D.i0 + s->D.nloc; }

// one observer's explicit views, keyed by subject: {key, since, first accuser<<3|confirmations}
struct HostView { uint32_t key, since, w; };
static int gather_views(swim_sim* s, uint32_t r, uint32_t o, std::vector<std::pair<uint32_t, HostView>>& out) {
  const SwDev& D = s->D;
  const uint32_t cap = (uint32_t)std::min<size_t>((size_t)D.VT + D.M, (s->scratch_bytes - 16) / 32);
  hipLaunchKernelGGL(k_gather_views, dim3(1), dim3(64), 0, s->stream, (const SwDev*)s->d_D, r, o, s->d_scratch, cap);
  std::vector<uint32_t> w(4 + (size_t)cap * 8);
  int rc = d2h(s, w.data(), (const uint32_t*)s->d_scratch, w.size());
  if (rc) return rc;
  const uint32_t n = std::min(w[0], cap);
  out.clear();
  for (uint32_t i = 0; i < n; i++) out.push_back({ w[4 + i * 8], HostView{ w[5 + i * 8], w[6 + i * 8], w[7 + i * 8] } });
  return SWIM_OK;
}
// the Leaving mark of a view record: bit 3 of a Suspect view's word, bit 1 otherwise (swim_device.h)
static bool view_leaving(uint32_t key, uint32_t w) { return SW_KST(key) == SWIM_STATE_SUSPECT ? ((w >> 3) & 1u) != 0 : ((w >> 1) & 1u) != 0; }
// `erased`: serf's reaper (or a prune) removed the member from this observer's list
static void fill_member(swim_member* out, uint32_t x, uint32_t key, uint32_t since, uint32_t wpack, bool erased = false, bool leaving = false) {
  memset(out, 0, sizeof *out); out->id = x;
  out->incarnation = SW_KINC(key); out->state = (uint8_t)SW_KST(key); out->state_change_ms = since;
  out->n_confirm = SW_KST(key) == SWIM_STATE_SUSPECT ? (uint8_t)(wpack & 7u) : 0;
  out->status = (SW_KINC(key) == 0 || (erased && SW_KST(key) >= SWIM_STATE_DEAD)) ? (uint8_t)SWIM_MEMBER_NONE : status_of(SW_KST(key));   // incarnation 0: never heard of it
  if (leaving && SW_KST(key) < SWIM_STATE_DEAD) out->status = SWIM_MEMBER_LEAVING;                     // a leave intent was seen
}
extern "C" int swim_members(swim_sim* s, uint32_t r, uint32_t o, swim_member* out, size_t cap, size_t* n_out) {
  if (!s || (!out && cap)) return SWIM_EINVAL;
  const SwDev& D = s->D;
  if (r >= D.R || o >= D.N || !is_local(s, o)) return SWIM_ERANGE;
  size_t n = std::min<size_t>(cap, D.N);
  std::vector<uint32_t> bk(n ? n : 1);
  int rc = n ? d2h(s, bk.data(), (const uint32_t*)D.bk + (size_t)r * D.N, n) : SWIM_OK;
  if (rc) return rc;
  uint4 h; if ((rc = d2h(s, &h, (const uint4*)D.hdr + ((size_t)r * D.nloc + (o - D.i0)), 1))) return rc;
  // implicit views (a Failed / Left member of the base row was erased by every observer's reaper before it got there)
  for (size_t x = 0; x < n; x++) fill_member(&out[x], (uint32_t)x, x == o ? SW_KEY(h.x, SWIM_STATE_ALIVE) : bk[x], 0, 0, x != o && D.reap_period != 0);
  std::vector<std::pair<uint32_t, HostView>> ex;
  if ((rc = gather_views(s, r, o, ex))) return rc;
  for (auto& e : ex) if (e.first < n) fill_member(&out[e.first], e.first, e.second.key, e.second.since, e.second.w, e.first != o && (e.second.w & 1u), e.first != o && view_leaving(e.second.key, e.second.w));
  if (o < n && out[o].state == SWIM_STATE_ALIVE && (h.y & 0xFF)) out[o].status = SWIM_MEMBER_LEAVING;
  if (n_out) *n_out = D.N;
  return SWIM_OK;
}
extern "C" int swim_view(swim_sim* s, uint32_t r, uint32_t o, uint32_t x, swim_member* out) {
  if (!s || !out) return SWIM_EINVAL;
  const SwDev& D = s->D;
  if (r >= D.R || o >= D.N || x >= D.N || !is_local(s, o)) return SWIM_ERANGE;
  uint32_t bk = 0; int rc = d2h(s, &bk, (const uint32_t*)D.bk + (size_t)r * D.N + x, 1);
  if (rc) return rc;
  uint4 h; if ((rc = d2h(s, &h, (const uint4*)D.hdr + ((size_t)r * D.nloc + (o - D.i0)), 1))) return rc;
  fill_member(out, x, x == o ? SW_KEY(h.x, SWIM_STATE_ALIVE) : bk, 0, 0, x != o && D.reap_period != 0);
  std::vector<std::pair<uint32_t, HostView>> ex;
  if ((rc = gather_views(s, r, o, ex))) return rc;
  for (auto& e : ex) if (e.first == x) fill_member(out, x, e.second.key, e.second.since, e.second.w, x != o && (e.second.w & 1u), x != o && view_leaving(e.second.key, e.second.w));
  if (x == o && out->state == SWIM_STATE_ALIVE && (h.y & 0xFF)) out->status = SWIM_MEMBER_LEAVING;
  return SWIM_OK;
}
extern "C" int swim_watch_events(swim_sim* s, uint32_t r, uint32_t node) {
  if (!s) return SWIM_EINVAL;
  SwDev& D = s->D;
  if (r >= D.R || node >= D.N || !is_local(s, node)) return SWIM_ERANGE;
  if (s->in_tick) return SWIM_ESTATE;
  if (node == D.watch) return SWIM_OK;
  std::vector<uint32_t> w(SWIM_EVENT_WATCHERS); uint32_t n = 0;
  int rc = d2h(s, w.data(), (const uint32_t*)D.ev_watch + (size_t)r * SWIM_EVENT_WATCHERS, SWIM_EVENT_WATCHERS);
  if (!rc) rc = d2h(s, &n, (const uint32_t*)D.ev_watch + (size_t)D.R * SWIM_EVENT_WATCHERS + r, 1);
  if (rc) return rc;
  for (uint32_t j = 0; j < n; j++) if (w[j] == node) return SWIM_OK;
  if (n >= SWIM_EVENT_WATCHERS) return SWIM_EOVERFLOW;
  const uint32_t n1 = n + 1;
  HIPCK(s, hipMemcpy(D.ev_watch + (size_t)r * SWIM_EVENT_WATCHERS + n, &node, 4, hipMemcpyHostToDevice));
  HIPCK(s, hipMemcpy(D.ev_watch + (size_t)D.R * SWIM_EVENT_WATCHERS + r, &n1, 4, hipMemcpyHostToDevice));
  s->pristine = false;                 // (an EventCh wants every tick looked at)
  if (!D.ev_any) { D.ev_any = 1; HIPCK(s, hipMemcpy(s->d_D, &D, sizeof D, hipMemcpyHostToDevice)); }   // the kernels read the descriptor from device memory
  return SWIM_OK;
}
extern "C" int swim_poll_events(swim_sim* s, swim_event* out, size_t cap, size_t* n_out) {
  if (!s || (!out && cap) || !n_out) return SWIM_EINVAL;
  uint32_t n = 0; int rc = d2h(s, &n, (const uint32_t*)s->D.ev_cnt, 1);
  if (rc) return rc;
  n = std::min(n, s->D.ev_cap);
  if (n) {
    size_t base = s->pending_events.size();
    s->pending_events.resize(base + n);
    if ((rc = d2h(s, s->pending_events.data() + base, (const swim_event*)s->D.events, n))) return rc;
    HIPCK(s, hipMemsetAsync(s->D.ev_cnt, 0, 4, s->stream));
    // one lane appends in program order; lanes of different replicas interleave arbitrarily
    std::stable_sort(s->pending_events.begin() + base, s->pending_events.end(), [](const swim_event& a, const swim_event& b) {
      return a.time_ms != b.time_ms ? a.time_ms < b.time_ms : a.replica != b.replica ? a.replica < b.replica : a.observer < b.observer;
    });
  }
  size_t k = std::min(cap, s->pending_events.size());
  std::copy(s->pending_events.begin(), s->pending_events.begin() + k, out);
  s->pending_events.erase(s->pending_events.begin(), s->pending_events.begin() + k);
  *n_out = k;
  return SWIM_OK;
}
extern "C" int swim_event_queued(swim_sim* s, uint32_t r, uint32_t i, uint32_t id, uint64_t ltime, int* queued) {
  if (!s || !queued) return SWIM_EINVAL;
  if (r >= s->D.R || i >= s->D.N || !is_local(s, i)) return SWIM_ERANGE;
  hipLaunchKernelGGL(k_evq_find, dim3(1), dim3(64), 0, s->stream, (const SwDev*)s->d_D, r, i, id, (uint32_t)ltime, s->d_scratch);
  uint32_t w = 0; int rc = d2h(s, &w, (const uint32_t*)s->d_scratch, 1);
  if (rc) return rc;
  *queued = w != 0;
  return SWIM_OK;
}
extern "C" int swim_node_info_get(swim_sim* s, uint32_t r, uint32_t i, swim_node_info* out) {
  if (!s || !out) return SWIM_EINVAL;
  const SwDev& D = s->D;
  if (r >= D.R || i >= D.N || !is_local(s, i)) return SWIM_ERANGE;
  hipLaunchKernelGGL(k_gather_node, dim3(1), dim3(64), 0, s->stream, (const SwDev*)s->d_D, r, i, s->d_scratch);
  uint32_t w[16 + 4 * 64]; int rc = d2h(s, w, (const uint32_t*)s->d_scratch, 16 + 4 * 64);
  if (rc) return rc;
  memset(out, 0, sizeof *out);
  out->incarnation = w[0]; out->probe_target = w[4]; out->probe_deadline_tick = w[4] == SWIM_NONE ? 0 : w[6];
  out->probe_cursor = w[8]; out->probe_epoch = w[9] >> 16;
  out->queue_len = (w[1] >> 8) & 0xFF; out->event_queue_len = (w[1] >> 16) & 0xFF; out->event_clock = w[3];
  out->alive = !(w[10] & NW_DEAD); out->leaving = w[1] & 0xFF; out->awareness = (w[9] >> 8) & 0xFF; out->partition = NW_PART(w[10]);
  // the struct holds 32; a queue may be deeper (SWIM_F_UNBOUNDED_QUEUE: queue_len = the slots' entries + what the pair store implies): the 32 oldest
  // entries (lowest sequence numbers), oldest first — like the checker
  const uint32_t n_slots = std::min<uint32_t>(out->queue_len, 32), n_imp = std::min<uint32_t>(w[12], 32);
  swim_rumour all[64]; uint32_t na = 0;
  for (uint32_t j = 0; j < n_slots + n_imp; j++) {
    const uint32_t* e = j < n_slots ? &w[16 + 4 * j] : &w[16 + 4 * 32 + 4 * (j - n_slots)];
    swim_rumour q = { e[0], e[1], e[2], (uint8_t)(e[3] >> 30), (uint8_t)((e[3] >> 22) & 0xFF), { 0, 0 }, e[3] & 0x3FFFFFu };
    all[na++] = q;
  }
  std::sort(all, all + na, [](const swim_rumour& a, const swim_rumour& b) { return a.seq < b.seq; });
  out->queue_len += w[11];
  memcpy(out->queue, all, std::min<uint32_t>(na, 32) * sizeof(swim_rumour));
  return SWIM_OK;
}
extern "C" int swim_census_get(swim_sim* s, uint32_t r, uint32_t x, swim_census* out) {
  if (!s || !out) return SWIM_EINVAL;
  const SwDev& D = s->D;
  if (r >= D.R || x >= D.N) return SWIM_ERANGE;
  if (s->in_tick) return SWIM_ESTATE;
  uint32_t w = 0; int rc = d2h(s, &w, (const uint32_t*)D.nw + (size_t)r * D.N + x, 1);
  if (rc) return rc;
  if (!NW_HAS_SLOT(w)) {            // not watched: counted on demand, no history
    HIPCK(s, hipMemsetAsync(s->d_scratch, 0, 32, s->stream));
    const uint32_t nb = std::max(1u, std::min<uint32_t>(cdiv(D.nloc, SW_BLOCK), 256));
    hipLaunchKernelGGL(k_census_adhoc, dim3(nb), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, r, x, 0, s->d_scratch);
    hipLaunchKernelGGL(k_census_adhoc, dim3(nb), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, r, x, 1, s->d_scratch);
    uint32_t a[8]; if ((rc = d2h(s, a, (const uint32_t*)s->d_scratch, 8))) return rc;
    memset(out, 0, sizeof *out);
    out->n_observers = a[0]; for (int i = 0; i < 4; i++) out->by_state[i] = a[1 + i]; out->n_current = a[5];
    out->first_suspect_ms = out->first_dead_ms = out->all_dead_ms = out->all_current_ms = SWIM_NONE;
    return SWIM_OK;
  }
  const uint32_t xb = std::max(1u, std::min<uint32_t>(cdiv(D.nloc, SW_BLOCK * 8), 16));
  hipLaunchKernelGGL(k_census, dim3(xb, D.R * D.S), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D);
  hipLaunchKernelGGL(k_census_commit, dim3(1), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D);
  return d2h(s, out, (const swim_census*)D.census + (size_t)r * D.S + NW_SLOT(w), 1);
}
extern "C" int swim_detection_get(swim_sim* s, uint32_t r, swim_detection* out) {
  if (!s || !out) return SWIM_EINVAL;
  const SwDev& D = s->D;
  if (r >= D.R) return SWIM_ERANGE;
  if (s->in_tick) return SWIM_ESTATE;
  unsigned long long* acc = (unsigned long long*)s->d_scratch; uint32_t* grp = (uint32_t*)(acc + 8);
  HIPCK(s, hipMemsetAsync(acc, 0, 64 + 129 * 4, s->stream));
  hipLaunchKernelGGL(k_detect_groups, dim3(cdiv(D.nloc, SW_BLOCK)), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, r, grp);
  hipLaunchKernelGGL(k_detect_base, dim3(cdiv(D.N, SW_BLOCK)), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, r, (const uint32_t*)grp, acc);
  hipLaunchKernelGGL(k_detect_tables, dim3(cdiv(D.nloc, SW_BLOCK)), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, r, acc);
  if (D.M) hipLaunchKernelGGL(k_detect_rows, dim3(D.M), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, r, acc);
  unsigned long long v[5]; int rc = d2h(s, v, (const unsigned long long*)acc, 5);
  if (rc) return rc;
  out->pairs = v[0]; for (int i = 0; i < 4; i++) out->by_state[i] = v[1 + i];
  return SWIM_OK;
}
extern "C" int swim_trace_read(swim_sim* s, uint32_t r, uint32_t x, uint32_t first, uint32_t n, uint32_t* rows) {
  if (!s || !rows) return SWIM_EINVAL;
  const SwDev& D = s->D;
  if (r >= D.R || x >= D.N) return SWIM_ERANGE;
  uint32_t w = 0; int rc = d2h(s, &w, (const uint32_t*)D.nw + (size_t)r * D.N + x, 1);
  if (rc) return rc;
  if (!NW_HAS_SLOT(w) || !D.trace_ticks) return SWIM_ESTATE;
  if ((uint64_t)first + n > D.trace_ticks || first + n > s->tick) return SWIM_ERANGE;
  return d2h(s, rows, (const uint32_t*)D.trace + (((size_t)r * D.S + NW_SLOT(w)) * D.trace_ticks + first) * 5, (size_t)n * 5);
}
extern "C" int swim_stats(swim_sim* s, swim_stats_t* out) {
  if (!s || !out) return SWIM_EINVAL;
  std::vector<unsigned long long> raw((size_t)SW_STAT_COPIES * SW_STAT_STRIDE);
  int rc = d2h(s, raw.data(), (const unsigned long long*)s->D.stats, raw.size());
  if (rc) return rc;
  unsigned long long v[ST_COUNT] = { 0 };
  for (int c = 0; c < SW_STAT_COPIES; c++)
    for (int i = 0; i < ST_COUNT; i++) v[i] += raw[(size_t)c * SW_STAT_STRIDE + i];
  memset(out, 0, sizeof *out);
  out->ticks = s->ticks_run; out->gossip_rounds = s->rounds_run;
  out->node_rounds_active = v[ST_ACTIVE]; out->node_rounds_quiescent = v[ST_QUIESCENT];
  out->packets_sent = v[ST_PKT_SENT]; out->packets_dropped = v[ST_PKT_DROP];
  for (int i = 0; i < 4; i++) { out->msgs_sent[i] = v[ST_SENT0 + i]; out->msgs_applied[i] = v[ST_APPL0 + i]; }
  out->probes = v[ST_PROBES]; out->probe_acks = v[ST_ACKS]; out->probe_indirect_acks = v[ST_IACKS];
  out->probe_failures = v[ST_PFAIL]; out->nacks_missed = v[ST_NACKMISS]; out->refutes = v[ST_REFUTES];
  out->suspicion_timeouts = v[ST_TIMEOUTS]; out->confirmations = v[ST_CONFIRMS];
  out->edges = v[ST_EDGES]; out->edges_remote = v[ST_EDGES_REMOTE]; out->queue_drops = v[ST_QDROPS];
  out->inbox_overflow = v[ST_INBOX_OVF]; out->subject_overflow = v[ST_SUBJ_OVF]; out->event_drops = v[ST_EVDROPS];
  out->user_events_delivered = v[ST_UEV_DELIVERED]; out->user_events_deduped = v[ST_UEV_DEDUP];
  out->user_events_stale = v[ST_UEV_STALE]; out->msgs_filtered = v[ST_FILTERED]; out->push_pulls = v[ST_PUSHPULLS];
  out->piggybacks = v[ST_PIGGY]; out->msgs_piggybacked = v[ST_PIGGY_MSGS]; out->probe_tcp_acks = v[ST_TCPACKS];
  out->view_drops = v[ST_VIEW_DROPS]; out->view_evictions = v[ST_VIEW_EVICT]; out->joins = v[ST_JOINS]; out->join_failures = v[ST_JOIN_FAIL]; out->intents_applied = v[ST_INTENTS]; out->reaped = v[ST_REAPED]; out->folds = v[ST_FOLDS]; out->fold_freed = v[ST_FOLD_FREED];
  out->coord_updates = v[ST_COORD_UPD]; out->coord_resets = v[ST_COORD_RESET];
  out->reconnects = v[ST_RECONNECTS]; out->reconnects_reached = v[ST_RECONNECT_OK];
  { uint32_t pk = 0; if ((rc = d2h(s, &pk, (const uint32_t*)s->D.peak, 1))) return rc; out->inbox_peak = pk; }
  return SWIM_OK;
}

// ---- network coordinates (SWIM_F_COORDINATES) ---------------------------------------------------------------
extern "C" int swim_coordinate_get(swim_sim* s, uint32_t replica, uint32_t node, swim_coordinate* out) {
  if (!s || !out || replica >= s->D.R || node >= s->D.N) return SWIM_EINVAL;
  if (!s->D.coord) return SWIM_ESTATE;
  return d2h(s, out, (const swim_coordinate*)s->D.coord + ((size_t)replica * s->D.nloc + (node - s->D.i0)), 1);
}
// librtt.ComputeDistance (internal/gossip/librtt/rtt.go:16-22) = a.DistanceTo(b).Seconds(): host arithmetic on two PODs
// (x86-64 without FMA: every operation rounds once, like Go)
extern "C" double swim_coordinate_distance(const swim_coordinate* a, const swim_coordinate* b) {
  if (!a || !b) return INFINITY;
  double sum = 0.0;
  for (int i = 0; i < SWIM_COORD_DIMS; i++) { const double d = a->vec[i] - b->vec[i]; sum += d * d; }
  double dist = std::sqrt(sum) + a->height + b->height;
  const double adjusted = dist + a->adjustment + b->adjustment;
  if (adjusted > 0.0) dist = adjusted;
  const int64_t ns = (int64_t)(dist * 1.0e9);                      // time.Duration: whole nanoseconds, truncated
  return (double)(ns / 1000000000) + (double)(ns % 1000000000) / 1e9;
}
extern "C" int swim_rtt_truth(swim_sim* s, uint32_t replica, uint32_t a, uint32_t b, uint32_t* rtt_us) {
  if (!s || !rtt_us || replica >= s->D.R || a >= s->D.N || b >= s->D.N) return SWIM_EINVAL;
  uint32_t w[2][4]; const uint64_t sr = s->cfg.seed + replica; const uint32_t ids[2] = { a, b };
  for (int j = 0; j < 2; j++) sw_philox(ids[j], 0, 0, 0x54525554u, (uint32_t)sr, (uint32_t)(sr >> 32) ^ SW_STREAM_TRUTH, w[j]);
  double sum = 0.0; uint32_t h = 0;
  for (int k = 0; k < 3; k++) {
    const double d = (double)(uint32_t)(((uint64_t)w[0][k] * s->cfg.rtt_scale_us) >> 32) - (double)(uint32_t)(((uint64_t)w[1][k] * s->cfg.rtt_scale_us) >> 32);
    sum += d * d;
  }
  for (int j = 0; j < 2; j++) h += (uint32_t)(((uint64_t)w[j][3] * s->cfg.rtt_height_us) >> 32);
  *rtt_us = (uint32_t)std::sqrt(sum) + h;
  return SWIM_OK;
}
extern "C" int swim_info(swim_sim* s, uint32_t what, uint64_t* out) {
  if (!s || !out) return SWIM_EINVAL;
  switch (what) {
    case SWIM_INFO_TILE_BUCKETS: *out = s->D.tb_on; return SWIM_OK;
    case SWIM_INFO_MAILBOX_KIND: *out = !strcmp(s->mailbox_kind, "fine") ? 1u : !strcmp(s->mailbox_kind, "uncached") ? 2u : !strcmp(s->mailbox_kind, "coarse") ? 3u : 0u; return SWIM_OK;
    case SWIM_INFO_DEVICE_BYTES: { uint64_t b = 0; for (size_t n : s->alloc_bytes) b += n; *out = b; return SWIM_OK; }
    default: return SWIM_EINVAL;
  }
}
extern "C" int swim_debug_edges(swim_sim* s, swim_edge* out, size_t cap, size_t* n_out) {
  if (!s || (!out && cap) || !n_out) return SWIM_EINVAL;
  if (s->in_tick) return SWIM_ESTATE;
  uint32_t cnt[SW_MAX_SHARDS]; int rc = d2h(s, cnt, (const uint32_t*)s->d_last_cnt, SW_MAX_SHARDS);
  if (rc) return rc;
  size_t total = 0, w = 0;
  std::vector<swim_edge> tmp;
  if (s->D.tb_on) {
    // tile buckets: what the roles handed to the tiles' buckets in the finished tick, minus what the receivers' filter dropped — k_resolve
    // voids those in place, but only once this call has switched the recording on (the first call reports the rumours of the tick
    // before it unfiltered; callers that compare tick by tick call once before they start)
    uint32_t on = 0;
    if ((rc = d2h(s, &on, (const uint32_t*)s->D.dbg_on, 1))) return rc;
    if (!on) { on = 1; HIPCK(s, hipMemcpy(s->D.dbg_on, &on, 4, hipMemcpyHostToDevice)); }
    else {
      std::vector<uint32_t> last(s->D.tb_T);
      if ((rc = d2h(s, last.data(), (const uint32_t*)s->D.tb_last, s->D.tb_T))) return rc;
      for (uint32_t tl = 0; tl < s->D.tb_T; tl++) {
        const uint32_t n = std::min(last[tl], s->D.tb_cap);
        if (!n) continue;
        tmp.resize(n);
        if ((rc = d2h(s, tmp.data(), (const swim_edge*)s->D.tb + (size_t)tl * s->D.tb_cap, n))) return rc;
        for (uint32_t i = 0; i < n; i++) {
          if (tmp[i].dst == SW_DST_VOID || tmp[i].subject == SWIM_SUBJECT_PIGGY || (tmp[i].meta & TB_CLASS_MASK) >= TB_CARRIED) continue;   // (carried broadcasts: from their areas, below)
          if (w < cap) { out[w] = tmp[i]; out[w].meta &= ~TB_CLASS_MASK; w++; }
          total++;
        }
      }
    }
  }
  std::vector<uint32_t> segn(s->D.n_seg);
  if ((rc = d2h(s, segn.data(), (const uint32_t*)s->D.seg_last, s->D.n_seg))) return rc;
  for (uint32_t b = 0; b < s->D.n_seg; b++) {
    uint32_t n = std::min(segn[b], s->D.seg_cap);
    if (!n) continue;
    tmp.resize(n);
    if ((rc = d2h(s, tmp.data(), (const swim_edge*)s->D.seg + (size_t)b * s->D.seg_cap, n))) return rc;
    for (uint32_t i = 0; i < n; i++) { if (tmp[i].subject == SWIM_SUBJECT_PIGGY) continue; if (w < cap) out[w++] = tmp[i]; total++; }
  }
  if (s->D.flags & SWIM_F_PIGGYBACK) {               // the broadcasts carried into the finished tick, before the filter
    std::vector<uint2> cl(s->D.NB);
    if ((rc = d2h(s, cl.data(), (const uint2*)s->D.carry_cl, s->D.NB))) return rc;
    const uint32_t par = (s->tick - 1) & 1u;
    uint32_t seen[2] = { SWIM_NONE, SWIM_NONE };
    if ((rc = d2h(s, seen, (const uint32_t*)s->D.carry_stamp, 2))) return rc;
    for (uint32_t a = 0; s->tick && seen[1] == s->tick - 1 && a < s->D.NB; a++) {
      uint32_t n = std::min(cl[a].y, s->D.carry_cap);
      if (!n) continue;
      tmp.resize(n);
      if ((rc = d2h(s, tmp.data(), (const swim_edge*)s->D.carry + ((size_t)par * s->D.NB + a) * s->D.carry_cap, n))) return rc;
      for (uint32_t i = 0; i < n; i++) { if (tmp[i].dst == SW_DST_VOID) continue; if (w < cap) out[w++] = tmp[i]; total++; }
    }
  }
  for (uint32_t sh = 0; sh < s->D.n_shards; sh++) {
    uint32_t n = std::min(cnt[sh], s->D.out_cap[sh]);
    tmp.resize(n);
    if (n && (rc = d2h(s, tmp.data(), (const swim_edge*)s->D.out[sh], n))) return rc;
    for (uint32_t i = 0; i < n; i++) {
      if (tmp[i].dst == SWIM_NONE || tmp[i].subject == SWIM_SUBJECT_PIGGY) continue;
      if (w < cap) { out[w] = tmp[i]; if (tmp[i].subject != SWIM_SUBJECT_PULL) out[w].meta &= ~SW_EDGE_JUDGE; w++; }
      total++;
    }
  }
  *n_out = total;
  return SWIM_OK;
}
// ---- checkpoint / resume (swimsim.h) -----------------------------------------------------------------------------
// Every device array was allocated through dalloc() in an order that depends on the configuration alone, so the state of a
// population is the contents of those arrays in that order plus a handful of host words.  The descriptor, the pointer
// tables and the scratch/staging buffers are this handle's own and are left alone.
#define SW_STATE_LAYOUT 5      /* how the device arrays are laid out (round 5: the dense pair store by groups of 64 observers, in_any per 64 nodes): a checkpoint of another layout is refused */
struct CkHeader {
  char magic[8], backend[16];
  uint32_t abi, tick, loss_q32, peer_act, n_arrays, n_events; uint8_t pristine, pad[3];
  uint64_t ticks_run, rounds_run;
  swim_config cfg;
};
static bool ck_is_state(const swim_sim* s, size_t i) {
  return s->allocs[i] && std::find(s->structural.begin(), s->structural.end(), s->allocs[i]) == s->structural.end();
}
static int ck_legal(swim_sim* s) {
  if (s->in_tick || s->in_count) return SWIM_ESTATE;
  if (!s->attached.empty() || !s->captured.empty() || s->xchg_connected) { snprintf(s->err, sizeof s->err, "checkpoints do not cover attached transport-bridge nodes or a connected exchange"); return SWIM_ESTATE; }
  return SWIM_OK;
}
extern "C" int swim_checkpoint_save(swim_sim* s, const char* path) {
  if (!s || !path) return SWIM_EINVAL;
  if (int rc = ck_legal(s)) return rc;
  HIPCK(s, hipStreamSynchronize(s->stream));
  FILE* f = fopen(path, "wb");
  if (!f) { snprintf(s->err, sizeof s->err, "cannot write %s", path); return SWIM_EIO; }
  CkHeader h; memset(&h, 0, sizeof h);
  memcpy(h.magic, "SWIMCKPT", 8); strncpy(h.backend, swim_backend(), sizeof h.backend - 1);
  h.pad[0] = SW_STATE_LAYOUT;
  h.abi = SWIM_ABI_VERSION; h.tick = s->tick; h.loss_q32 = s->D.loss_q32; h.peer_act = s->peer_act_host; h.pristine = s->pristine;
  h.ticks_run = s->ticks_run; h.rounds_run = s->rounds_run; h.cfg = s->cfg; h.n_events = (uint32_t)s->pending_events.size();
  for (size_t i = 0; i < s->allocs.size(); i++) h.n_arrays += ck_is_state(s, i);
  bool ok = fwrite(&h, sizeof h, 1, f) == 1;
  ok = ok && (s->pending_events.empty() || fwrite(s->pending_events.data(), sizeof(swim_event), s->pending_events.size(), f) == s->pending_events.size());
  const size_t CH = (size_t)32 << 20;
  std::vector<char> buf(CH);
  for (size_t i = 0; ok && i < s->allocs.size(); i++) {
    if (!ck_is_state(s, i)) continue;
    const uint64_t bytes = s->alloc_bytes[i];
    ok = fwrite(&bytes, 8, 1, f) == 1;
    for (size_t off = 0; ok && off < bytes; off += CH) {
      const size_t c = std::min<size_t>(CH, bytes - off);
      if (hipMemcpy(buf.data(), (const char*)s->allocs[i] + off, c, hipMemcpyDeviceToHost) != hipSuccess) { fclose(f); snprintf(s->err, sizeof s->err, "device read failed"); return SWIM_EIO; }
      ok = fwrite(buf.data(), 1, c, f) == c;
    }
  }
  ok = ok && fwrite("SWIMCKND", 8, 1, f) == 1;            // trailer: a file cut short is recognised before anything is loaded
  if (fclose(f) != 0) ok = false;
  if (!ok) snprintf(s->err, sizeof s->err, "short write to %s", path);
  return ok ? SWIM_OK : SWIM_EIO;
}
extern "C" int swim_checkpoint_load(swim_sim* s, const char* path) {
  if (!s || !path) return SWIM_EINVAL;
  if (int rc = ck_legal(s)) return rc;
  HIPCK(s, hipStreamSynchronize(s->stream));
  FILE* f = fopen(path, "rb");
  if (!f) { snprintf(s->err, sizeof s->err, "cannot read %s", path); return SWIM_EIO; }
  CkHeader h;
  uint32_t n_arrays = 0;
  for (size_t i = 0; i < s->allocs.size(); i++) n_arrays += ck_is_state(s, i);
  if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, "SWIMCKPT", 8) || strncmp(h.backend, swim_backend(), sizeof h.backend) || h.abi != SWIM_ABI_VERSION || h.pad[0] != SW_STATE_LAYOUT ||
      memcmp(&h.cfg, &s->cfg, sizeof h.cfg) || h.n_arrays != n_arrays) {
    fclose(f); snprintf(s->err, sizeof s->err, "checkpoint of another library, ABI or configuration"); return SWIM_EINVAL;
  }
  if (h.n_events > (1u << 24)) { fclose(f); snprintf(s->err, sizeof s->err, "checkpoint header is damaged (events)"); return SWIM_EINVAL; }
  std::vector<swim_event> ev;
  try { ev.resize(h.n_events); } catch (const std::exception&) { fclose(f); return SWIM_ENOMEM; }
  bool ok = h.n_events == 0 || fread(ev.data(), sizeof(swim_event), h.n_events, f) == h.n_events;
  {   // first pass: every array's size and the trailer, BEFORE anything on the device is overwritten (a refusal is a clean refusal)
    const long data0 = ftell(f);
    for (size_t i = 0; ok && i < s->allocs.size(); i++) {
      if (!ck_is_state(s, i)) continue;
      uint64_t bytes = 0;
      ok = fread(&bytes, 8, 1, f) == 1;
      if (ok && bytes != s->alloc_bytes[i]) { fclose(f); snprintf(s->err, sizeof s->err, "checkpoint of another configuration (array %zu)", i); return SWIM_EINVAL; }
      ok = ok && fseek(f, (long)bytes, SEEK_CUR) == 0;
    }
    char tail[8] = { 0 };
    if (!ok || fread(tail, 8, 1, f) != 1 || memcmp(tail, "SWIMCKND", 8)) { fclose(f); snprintf(s->err, sizeof s->err, "checkpoint truncated or damaged: nothing was loaded"); return SWIM_EIO; }
    fseek(f, data0, SEEK_SET);
  }
  const size_t CH = (size_t)32 << 20;
  std::vector<char> buf(CH);
  for (size_t i = 0; ok && i < s->allocs.size(); i++) {
    if (!ck_is_state(s, i)) continue;
    uint64_t bytes = 0;
    ok = fread(&bytes, 8, 1, f) == 1;
    if (ok && bytes != s->alloc_bytes[i]) { fclose(f); snprintf(s->err, sizeof s->err, "checkpoint of another configuration (array %zu)", i); return SWIM_EINVAL; }   // (nothing was overwritten yet only if i is the first)
    for (size_t off = 0; ok && off < bytes; off += CH) {
      const size_t c = std::min<size_t>(CH, bytes - off);
      ok = fread(buf.data(), 1, c, f) == c;
      if (ok && hipMemcpy((char*)s->allocs[i] + off, buf.data(), c, hipMemcpyHostToDevice) != hipSuccess) { fclose(f); snprintf(s->err, sizeof s->err, "device write failed"); return SWIM_EIO; }
    }
  }
  fclose(f);
  if (!ok) { snprintf(s->err, sizeof s->err, "checkpoint truncated: the handle's state is undefined"); return SWIM_EIO; }
  s->tick = h.tick; s->pristine = h.pristine != 0; s->ticks_run = h.ticks_run; s->rounds_run = h.rounds_run; s->peer_act_host = h.peer_act;
  s->pending_events = std::move(ev); s->out_counts_valid = false; s->in_count = 0;
  {   // two words of the descriptor follow the state: the loss rate, and whether any observer has an EventCh of its own
    std::vector<uint32_t> nw(s->D.R); uint32_t any = 0;
    if (int rc = d2h(s, nw.data(), (const uint32_t*)s->D.ev_watch + (size_t)s->D.R * SWIM_EVENT_WATCHERS, s->D.R)) return rc;
    for (uint32_t v : nw) any |= v;
    any = any ? 1u : 0u;
    if (s->D.loss_q32 != h.loss_q32 || s->D.ev_any != any) { s->D.loss_q32 = h.loss_q32; s->D.ev_any = any; HIPCK(s, hipMemcpy(s->d_D, &s->D, sizeof s->D, hipMemcpyHostToDevice)); }
  }
  return SWIM_OK;
}

extern "C" int swim_state_digest(swim_sim* s, uint64_t* out) {
  if (!s || !out) return SWIM_EINVAL;
  const SwDev& D = s->D;
  unsigned long long* acc = (unsigned long long*)s->d_scratch;   // 64 partials, one 64-byte line each
  HIPCK(s, hipMemsetAsync(acc, 0, 64 * 8 * 8, s->stream));
  const size_t NL = (size_t)D.R * D.nloc;
  hipLaunchKernelGGL(k_digest_nodes, dim3(cdiv(NL, SW_BLOCK)), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, acc);
  hipLaunchKernelGGL(k_digest_views, dim3(cdiv(NL, SW_BLOCK)), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, acc);
  if (D.M) hipLaunchKernelGGL(k_digest_mass, dim3(D.R * D.M), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, acc);
  unsigned long long v[64 * 8]; int rc = d2h(s, v, (const unsigned long long*)acc, 64 * 8);
  if (rc) return rc;
  uint64_t d = 0;
  for (int i = 0; i < 64; i++) d += v[i * 8];
  *out = d;
  return SWIM_OK;
}

extern "C" int swim_profile(swim_sim* s, int enable) {
  if (!s) return SWIM_EINVAL;
  HIPCK(s, hipStreamSynchronize(s->stream));
  s->profiling = enable != 0; s->prof_recs.clear(); s->ev_used = 0;
  return SWIM_OK;
}
extern "C" int swim_profile_read(swim_sim* s, swim_kernel_time* out, size_t cap, size_t* n_out) {
  if (!s || (!out && cap) || !n_out) return SWIM_EINVAL;
  HIPCK(s, hipStreamSynchronize(s->stream));
  uint64_t launches[PK_COUNT] = { 0 }; double total[PK_COUNT] = { 0 };
  for (const auto& r : s->prof_recs) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, s->ev_pool[r.ev], s->ev_pool[r.ev + 1]) == hipSuccess) { launches[r.kernel]++; total[r.kernel] += ms; }
  }
  s->prof_recs.clear(); s->ev_used = 0;
  size_t n = std::min<size_t>(cap, PK_COUNT);
  for (size_t i = 0; i < n; i++) {
    memset(&out[i], 0, sizeof out[i]);
    snprintf(out[i].name, sizeof out[i].name, "%s", kKernelNames[i]);
    out[i].launches = launches[i]; out[i].total_ms = total[i];
  }
  *n_out = n;
  return SWIM_OK;
}

// memberlist.Transport bridge at rumour granularity (the msgpack codec is the host shim's job).  The first
// call naming `a` attaches it: the simulator stops acting for it, peers keep seeing it alive.
static int attach(swim_sim* s, uint32_t r, uint32_t a) {
  if (!s) return SWIM_EINVAL;
  if (s->in_tick || s->cfg.n_shards != 1) return SWIM_ESTATE;      // the bridge is defined for an unsharded population
  if (r >= s->D.R || a >= s->D.N) return SWIM_ERANGE;
  uint64_t key = ((uint64_t)r << 32) | a;
  if (std::find(s->attached.begin(), s->attached.end(), key) != s->attached.end()) return SWIM_OK;
  s->pristine = false;
  hipLaunchKernelGGL(k_attach, dim3(1), dim3(64), 0, s->stream, (const SwDev*)s->d_D, r, a);
  hipLaunchKernelGGL(k_exc_rebuild, dim3(1), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, r);
  HIPCK(s, hipStreamSynchronize(s->stream));
  s->attached.push_back(key);
  return SWIM_OK;
}
// Transport.WriteToAddress: a packet of n rumours from the attached node to a virtual peer; it is in the
// peer's inbox at once and merged at the end of the next tick
extern "C" int swim_transport_write_to(swim_sim* s, uint32_t r, uint32_t a, uint32_t dst, const swim_edge* m, size_t n) {
  int rc = attach(s, r, a);
  if (rc) return rc;
  if (dst >= s->D.N || (!m && n) || n * sizeof(swim_edge) + n * 4 > s->scratch_bytes) return SWIM_EINVAL;
  if (!n) return SWIM_OK;
  touched(s);
  std::vector<swim_edge> recs(m, m + n);
  for (auto& e : recs) {
    if ((e.meta >> 30) != SWIM_MSG_USER ? e.subject >= s->D.N : e.subject > SWIM_EVENT_ID_MAX) return SWIM_ERANGE;   // (a user event never carries intent bits)
    e.dst = r * s->D.N + dst;
  }
  uint8_t* drec = (uint8_t*)s->d_scratch;
  HIPCK(s, hipMemcpyAsync(drec, recs.data(), n * sizeof(swim_edge), hipMemcpyHostToDevice, s->stream));
  hipLaunchKernelGGL(k_deliver_list, dim3(cdiv(n, SW_BLOCK * 4)), dim3(SW_BLOCK), 0, s->stream, (const SwDev*)s->d_D, (const uint4*)drec, (uint32_t)n);
  HIPCK(s, hipStreamSynchronize(s->stream));
  return SWIM_OK;
}
// Transport.PacketCh: rumours virtual peers sent to the attached node since the last poll; out[i].dst
// carries the SENDER (Packet.From), SWIM_NONE when it is not a gossip packet
extern "C" int swim_transport_poll(swim_sim* s, uint32_t r, uint32_t a, swim_edge* o, size_t cap, size_t* n) {
  int rc = attach(s, r, a);
  if (rc) return rc;
  if (!n || (!o && cap)) return SWIM_EINVAL;
  uint32_t cnt = 0;
  if ((rc = d2h(s, &cnt, (const uint32_t*)s->D.cap_cnt, 1))) return rc;
  cnt = std::min(cnt, s->D.cap_cap);
  if (cnt) {
    std::vector<uint4> recs(cnt); std::vector<uint32_t> dsts(cnt);
    if ((rc = d2h(s, recs.data(), (const uint4*)s->D.cap, cnt)) || (rc = d2h(s, dsts.data(), (const uint32_t*)s->D.cap_dst, cnt))) return rc;
    HIPCK(s, hipMemsetAsync(s->D.cap_cnt, 0, 4, s->stream));
    for (uint32_t i = 0; i < cnt; i++) s->captured.push_back({ dsts[i], { recs[i].x, recs[i].y, recs[i].z, recs[i].w } });
  }
  size_t w = 0; const uint32_t g = r * s->D.N + a;
  std::vector<swim_sim::Captured> keep;
  for (auto& c : s->captured) {
    if (c.gdst == g && w < cap) o[w++] = c.rec;
    else keep.push_back(c);
  }
  s->captured.swap(keep);
  *n = w;
  return SWIM_OK;
}

// ---------------------------------------------------------------------------------------------
// known-answer hooks
// ---------------------------------------------------------------------------------------------
extern "C" void swim_kat_philox4x32(const uint32_t c[4], const uint32_t k[2], uint32_t o[4]) { sw_philox(c[0], c[1], c[2], c[3], k[0], k[1], o); }
extern "C" uint32_t swim_kat_probe_perm(uint64_t seed, uint32_t n, uint32_t node, uint32_t epoch, uint32_t index) { return sw_probe_perm(seed, n, node, epoch, index); }
extern "C" int32_t swim_kat_remaining_suspicion_ms(uint32_t n, uint32_t k, uint32_t el, uint32_t mn, uint32_t mx) { return (int32_t)remaining_suspicion_ms(n, k, el, mn, mx); }
// awareness.go ApplyDelta / ScaleTimeout: the same expressions as awareness_apply() in swim_kernels.hip and the probe
// deadline `t + P * (awareness + 1)` of the probe role
extern "C" uint32_t swim_kat_awareness_apply(uint32_t max_mult, uint32_t score, int32_t delta) {
  int v = (int)score + delta, mx = (int)max_mult - 1;
  return (uint32_t)(v < 0 ? 0 : v > mx ? mx : v);
}
extern "C" uint32_t swim_kat_awareness_scale_ms(uint32_t score, uint32_t timeout_ms) { return timeout_ms * (score + 1); }
extern "C" void swim_kat_phase_of(const swim_config* cfg, uint32_t node, uint32_t* gp, uint32_t* pp) {
  swim_derived d;
  if (swim_config_derive(cfg, &d)) { if (gp) *gp = SWIM_NONE; if (pp) *pp = SWIM_NONE; return; }
  uint32_t c = node / d.phase_chunk;
  if (gp) *gp = c % d.gossip_period;
  if (pp) *pp = (c / d.gossip_period) % d.probe_period;
}

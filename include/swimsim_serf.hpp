// swimsim_serf.hpp — host side above the C-ABI, in the reference's shape.
//
// Consul's seam to its gossip layer is a dozen methods on *serf.Serf plus one event channel
// (SURVEY.md §8(b)).  This header mirrors that surface — same names, argument meaning and error
// behaviour — over include/swimsim.h, so code (and tests) written against hashicorp/serf read the
// same here.  The reference is Go; this image has no Go toolchain, so the compiled host side is C++
// (header only, links against whichever library exports the C-ABI).  INTEGRATION.md holds the cgo
// shim a Consul maintainer would add instead.
//
//   memberlist::Config / DefaultLANConfig() ...   memberlist config.go; Consul sets the six gossip knobs at
//                                                 agent/agent.go:1410-1446 and clones them at agent/consul/config.go:679-716
//   memberlist::Transport / Delegate / ...        the plugin interfaces (wanfed.Transport implements the first:
//                                                 agent/consul/wanfed/wanfed.go:36-141)
//   serf::Config / DefaultConfig()                serf config.go; Consul's flavour: internal/gossip/libserf/serf.go:19-36
//   serf::Member / MemberStatus / Event           api/agent.go:291-311; events consumed at agent/consul/server_serf.go:270-297
//   serf::Cluster                                 NEW: the simulated gossip pool all virtual members live in
//   serf::Serf                                    serf.Create/Join/Leave/Shutdown/Members/LocalMember/UserEvent/
//                                                 SetTags/RemoveFailedNode/Stats/NumNodes as called from
//                                                 agent/consul/{client,server,server_serf,server_ce,leader}.go
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "swimsim.h"

namespace swimsim {

using Duration = std::chrono::milliseconds;

struct Error : std::runtime_error {
  int code;
  Error(const std::string& what, int rc) : std::runtime_error(what + " (swim rc " + std::to_string(rc) + ")"), code(rc) {}
};
inline void check(int rc, const char* what) { if (rc) throw Error(what, rc); }

// =================================================================================================
namespace memberlist {

// memberlist.Config: the fields Consul touches (agent/consul/config.go:679-716) + the fixed defaults
struct Config {
  std::string Name;
  int GossipNodes = 3, IndirectChecks = 3, RetransmitMult = 4, SuspicionMult = 4;
  int SuspicionMaxTimeoutMult = 6, AwarenessMaxMultiplier = 8, UDPBufferSize = 1400;
  Duration GossipInterval{200}, ProbeInterval{1000}, ProbeTimeout{500};
  Duration GossipToTheDeadTime{30000}, PushPullInterval{30000}, TCPTimeout{10000};
  Duration DeadNodeReclaimTime{0};
  bool DisableTcpPings = false;
};
inline Config DefaultLANConfig() { return Config{}; }
inline Config DefaultWANConfig() {
  Config c; c.TCPTimeout = Duration(30000); c.SuspicionMult = 6; c.PushPullInterval = Duration(60000);
  c.ProbeTimeout = Duration(3000); c.ProbeInterval = Duration(5000); c.GossipNodes = 4;
  c.GossipInterval = Duration(500); c.GossipToTheDeadTime = Duration(60000);
  return c;
}
inline Config DefaultLocalConfig() {
  Config c; c.TCPTimeout = Duration(1000); c.IndirectChecks = 1; c.RetransmitMult = 2; c.SuspicionMult = 3;
  c.PushPullInterval = Duration(15000); c.ProbeTimeout = Duration(200); c.GossipInterval = Duration(100);
  c.GossipToTheDeadTime = Duration(15000);
  return c;
}

// (memberlist.Transport / NodeAwareTransport at the byte level is include/swimsim_wire.hpp's BridgeTransport; memberlist's
// Delegate / EventDelegate are what serf itself implements inside the simulator — the facade below is the serf side of them.)
}  // namespace memberlist

// =================================================================================================
// serf/coordinate.Coordinate is the ABI's POD; librtt.ComputeDistance (internal/gossip/librtt/rtt.go:16-22) on top of it
namespace coordinate { using Coordinate = swim_coordinate; }
namespace librtt {
inline double ComputeDistance(const coordinate::Coordinate* a, const coordinate::Coordinate* b) { return swim_coordinate_distance(a, b); }
// GenerateCoordinate (rtt.go:59-64, tests only): NewCoordinate(DefaultConfig()) at `rtt` from the origin, no height
inline coordinate::Coordinate GenerateCoordinate(std::chrono::nanoseconds rtt) {
  coordinate::Coordinate c{};
  c.error = 1.5; c.vec[0] = std::chrono::duration<double>(rtt).count();
  return c;
}
}  // namespace librtt

namespace serf {

enum MemberStatus { StatusNone = 0, StatusAlive = 1, StatusLeaving = 2, StatusLeft = 3, StatusFailed = 4 };
enum EventType { EventMemberJoin = 0, EventMemberLeave, EventMemberFailed, EventMemberUpdate, EventMemberReap, EventUser, EventQuery };
enum SerfState { SerfAlive = 0, SerfLeaving, SerfLeft, SerfShutdown };

inline const char* StatusString(MemberStatus s) {
  static const char* n[] = { "none", "alive", "leaving", "left", "failed" };
  return n[s];
}

struct Member {
  std::string Name, Addr;
  uint16_t Port = 0;
  std::map<std::string, std::string> Tags;
  MemberStatus Status = StatusNone;
  uint8_t ProtocolMin = 1, ProtocolMax = 5, ProtocolCur = 2, DelegateMin = 2, DelegateMax = 5, DelegateCur = 4;
  uint32_t id = 0, Incarnation = 0;       // simulator-side identity
};

// serf.Event: MemberEvent{Type, Members} or UserEvent{LTime, Name, Payload, Coalesce}
struct Event {
  EventType Type = EventMemberJoin;
  std::vector<Member> Members;
  uint64_t LTime = 0;
  std::string Name;
  std::vector<uint8_t> Payload;
  bool Coalesce = false;
  Duration At{0};
};

// serf.Config: what Consul sets (agent/consul/server_serf.go:71-252, config.go:553-555,640-653)
struct Config {
  std::string NodeName;
  std::map<std::string, std::string> Tags;
  size_t EventChCap = 256;                 // Consul: 2048 servers / 256 clients (server.go:112-114, client.go:51-59)
  memberlist::Config MemberlistConfig;
  Duration ReapInterval{15000}, ReconnectTimeout{24 * 3600 * 1000}, TombstoneTimeout{24 * 3600 * 1000};
  Duration LeavePropagateDelay{1000};
  Duration BroadcastTimeout{5000};         // serf.DefaultConfig: how long Leave() waits for its intent to be sent out
  int EventBuffer = 512, UserEventSizeLimit = 512, MaxQueueDepth = 4096, MinQueueDepth = 0;
  uint8_t ProtocolVersion = 4;
  // serf.Config.Merge (MergeDelegate.NotifyMerge): called with the members a join / push-pull would merge; a non-empty string
  // is the error that vetoes it.  Consul: lanMergeDelegate / wanMergeDelegate (agent/consul/merge.go:34-87, 111-131) refuse a
  // member of another datacenter, a server id clash, a segment mismatch.
  std::function<std::string(const std::vector<Member>&)> Merge;
  // serf.Config.ReconnectTimeoutOverride.ReconnectTimeout(member, timeout): per-member reap timeout
  // (internal/gossip/libserf/serf.go:68-85 reads the member's "rc_tm" tag)
  std::function<Duration(const Member&, Duration)> ReconnectTimeoutOverride;
};
inline Config DefaultConfig() { return Config{}; }
// internal/gossip/libserf/serf.go:19-36 + agent/consul/config.go:640-641
inline Config ConsulDefaultConfig() {
  Config c; c.MinQueueDepth = 4096; c.LeavePropagateDelay = Duration(3000);
  c.ReconnectTimeout = Duration(3LL * 24 * 3600 * 1000);
  return c;
}

// The virtual gossip pool.  Every member of the pool is a lane on the device; stimulus (kill, partition,
// loss) is applied to the pool; a serf::Serf is one member's handle into it.
class Cluster {
 public:
  struct Options {
    uint32_t Nodes = 128, Replicas = 1, QueueCap = 8, InboxCap = 32, SubjectCap = 16, WatchNode = 0;
    uint64_t Seed = 1;
    uint32_t Device = 0;
    int EventBuffer = 512;
    uint32_t Initial = 0;                  // members at t = 0 (0 = all Nodes); the others start with Serf::Join
    uint32_t ViewCap = 0;                  // explicit views per member (0 = min(Nodes, 32))
    // serf's reaper runs on the device for the whole pool (serf.Config.ReapInterval / ReconnectTimeout / TombstoneTimeout
    // of the members; Consul: agent/consul/config.go:640-641).  ReapIntervalMs = 0: Members() applies the timeouts of the
    // calling member's own Config on the host instead.
    uint32_t ReapIntervalMs = 0, ReconnectTimeoutMs = 0, TombstoneTimeoutMs = 0;
    uint32_t FoldIntervalMs = 0;
    // serf.Config.DisableCoordinates = false (Consul's LAN and WAN pools both keep coordinates: agent/consul/config.go:589-591
    // only tunes how often they are written to the catalog): every member keeps a Vivaldi coordinate, updated on each direct
    // probe ack.  The latency the probes measure is the simulator's model (swimsim.h, swim_config.rtt_*).
    bool Coordinates = false;
    uint32_t RttScaleUs = 40000, RttHeightUs = 2000, RttJitterUs = 0;
    // serf.Config.ReconnectInterval of the members (serf default 30 s): what heals a partition older than GossipToTheDeadTime
    uint32_t ReconnectIntervalMs = 0;
    // The tags of a virtual member nobody holds a Serf handle for (every member carries tags in serf: Consul's consumers start
    // with metadata.IsConsulServer(m) on m.Tags["role"], agent/metadata/server.go:77-80).  nullptr: {"role": "node"}.
    std::function<std::map<std::string, std::string>(uint32_t)> DefaultTags;
  };
  Cluster(const memberlist::Config& mc, const Options& o) : opts_(o) {
    check(swim_config_preset(&cfg_, SWIM_PRESET_LAN), "swim_config_preset");
    cfg_.n_nodes = o.Nodes; cfg_.n_replicas = o.Replicas; cfg_.seed = o.Seed; cfg_.device = o.Device;
    cfg_.gossip_nodes = (uint32_t)mc.GossipNodes; cfg_.gossip_interval_ms = (uint32_t)mc.GossipInterval.count();
    cfg_.probe_interval_ms = (uint32_t)mc.ProbeInterval.count(); cfg_.probe_timeout_ms = (uint32_t)mc.ProbeTimeout.count();
    cfg_.suspicion_mult = (uint32_t)mc.SuspicionMult; cfg_.retransmit_mult = (uint32_t)mc.RetransmitMult;
    cfg_.indirect_checks = (uint32_t)mc.IndirectChecks; cfg_.suspicion_max_timeout_mult = (uint32_t)mc.SuspicionMaxTimeoutMult;
    cfg_.awareness_max_mult = (uint32_t)mc.AwarenessMaxMultiplier; cfg_.gossip_to_dead_ms = (uint32_t)mc.GossipToTheDeadTime.count();
    cfg_.udp_buffer_size = (uint32_t)mc.UDPBufferSize; cfg_.push_pull_interval_ms = (uint32_t)mc.PushPullInterval.count();
    cfg_.queue_cap = o.QueueCap; cfg_.inbox_cap = o.InboxCap; cfg_.subject_cap = o.SubjectCap; cfg_.watch_node = o.WatchNode;
    cfg_.event_buffer = (uint32_t)o.EventBuffer; cfg_.flags |= SWIM_F_SERF_EVENTS;
    cfg_.n_initial = o.Initial; cfg_.view_cap = o.ViewCap; cfg_.fold_interval_ms = o.FoldIntervalMs;
    cfg_.reap_interval_ms = o.ReapIntervalMs; cfg_.reconnect_timeout_ms = o.ReconnectTimeoutMs; cfg_.tombstone_timeout_ms = o.TombstoneTimeoutMs;
    cfg_.reconnect_interval_ms = o.ReconnectIntervalMs;
    if (o.Coordinates) { cfg_.flags |= SWIM_F_COORDINATES; cfg_.rtt_scale_us = o.RttScaleUs; cfg_.rtt_height_us = o.RttHeightUs; cfg_.rtt_jitter_us = o.RttJitterUs; }
    check(swim_config_derive(&cfg_, &derived_), "swim_config_derive");
    check(swim_create(&cfg_, &sim_), "swim_create");
  }
  ~Cluster() { if (sim_) swim_destroy(sim_); }
  Cluster(const Cluster&) = delete;
  Cluster& operator=(const Cluster&) = delete;

  // advance simulated time for every member (memberlist's tickers)
  void Advance(Duration d) {
    if (d.count() % derived_.quantum_ms) throw Error("Advance: not a multiple of the tick", SWIM_EINVAL);
    check(swim_step(sim_, (uint32_t)(d.count() / derived_.quantum_ms)), "swim_step");
    check(swim_sync(sim_), "swim_sync");
  }
  Duration Now() const { uint32_t t = 0, ms = 0; swim_now(sim_, &t, &ms); return Duration(ms); }
  // fault injection (the reference's tests call Shutdown() on a member: agent/consul/server_test.go:725)
  void Kill(const std::vector<uint32_t>& ids, uint32_t replica = 0) { check(swim_inject_kill(sim_, replica, ids.data(), ids.size()), "swim_inject_kill"); }
  void Revive(const std::vector<uint32_t>& ids, uint32_t replica = 0) { check(swim_inject_revive(sim_, replica, ids.data(), ids.size()), "swim_inject_revive"); }
  void Partition(const std::vector<uint8_t>& group, uint32_t replica = 0) {
    if (group.size() != cfg_.n_nodes) throw Error("Partition: one group id per node", SWIM_EINVAL);
    check(swim_inject_partition(sim_, replica, group.data()), "swim_inject_partition");
  }
  // the whole population into a file between two Advance() calls, and back into a Cluster built with the same Options
  // (swim_checkpoint_save / _load; serf's own snapshotter — conf.SnapshotPath, agent/consul/server_serf.go:236-239 — keeps one
  // node's member list, the simulator keeps everybody's)
  void Checkpoint(const std::string& path) { check(swim_checkpoint_save(sim_, path.c_str()), "swim_checkpoint_save"); }
  void Restore(const std::string& path) { check(swim_checkpoint_load(sim_, path.c_str()), "swim_checkpoint_load"); }
  // memberlist.Config.DisableTcpPingsForNode the way Consul's WAN pool sets it (agent/consul/server_serf.go:222-232): members of
  // different datacenters skip the TCP fallback ping of a failed probe
  void SetDatacenter(uint32_t replica, const std::vector<uint32_t>& ids, uint8_t dc) {
    check(swim_set_tcp_class(sim_, replica, ids.data(), ids.size(), dc), "swim_set_tcp_class");
  }
  void SetPacketLoss(double p) { check(swim_set_loss(sim_, (uint32_t)std::min(4294967295.0, p * 4294967296.0)), "swim_set_loss"); }

  // ---- what the members of the pool share on the host: tags by member, the EventCh router, who holds a handle ----
  static uint64_t key(uint32_t replica, uint32_t id) { return ((uint64_t)replica << 32) | id; }
  void SetTags(uint32_t replica, uint32_t id, const std::map<std::string, std::string>& t) { tags_[key(replica, id)] = t; }
  std::map<std::string, std::string> TagsOf(uint32_t replica, uint32_t id) const {
    auto it = tags_.find(key(replica, id));
    if (it != tags_.end()) return it->second;
    if (opts_.DefaultTags) return opts_.DefaultTags(id);
    return { { "role", "node" } };
  }
  // swim_poll_events hands out every observer's events in one stream: sort them into the handles' channels (nothing is dropped:
  // an event for an observer without a handle waits in its queue)
  void Route() {
    swim_event buf[256]; size_t n = 0;
    do {
      check(swim_poll_events(sim_, buf, 256, &n), "swim_poll_events");
      for (size_t i = 0; i < n; i++) inbox_[key(buf[i].replica, buf[i].observer)].push_back(buf[i]);
    } while (n == 256);
  }
  bool NextEvent(uint32_t replica, uint32_t id, swim_event* out) {
    auto it = inbox_.find(key(replica, id));
    if (it == inbox_.end() || it->second.empty()) return false;
    *out = it->second.front(); it->second.pop_front();
    return true;
  }
  // does the device record this member's serf events tick by tick (cfg.watch_node, or registered with swim_watch_events)?
  bool WatchEvents(uint32_t replica, uint32_t id) { return id == cfg_.watch_node || swim_watch_events(sim_, replica, id) == SWIM_OK; }
  void Register(uint32_t replica, uint32_t id, void* serf) { handles_[key(replica, id)] = serf; }
  void Unregister(uint32_t replica, uint32_t id, void* serf) { auto it = handles_.find(key(replica, id)); if (it != handles_.end() && it->second == serf) handles_.erase(it); }
  void* HandleOf(uint32_t replica, uint32_t id) const { auto it = handles_.find(key(replica, id)); return it == handles_.end() ? nullptr : it->second; }

  swim_sim* handle() const { return sim_; }
  const swim_config& config() const { return cfg_; }
  const swim_derived& derived() const { return derived_; }
  static std::string NodeName(uint32_t id) { return "node-" + std::to_string(id); }
  static std::string NodeAddr(uint32_t id) {
    return "10." + std::to_string((id >> 16) & 255) + "." + std::to_string((id >> 8) & 255) + "." + std::to_string(id & 255);
  }

 private:
  Options opts_;
  swim_config cfg_{};
  swim_derived derived_{};
  swim_sim* sim_ = nullptr;
  std::map<uint64_t, std::map<std::string, std::string>> tags_;
  std::map<uint64_t, std::deque<swim_event>> inbox_;
  std::map<uint64_t, void*> handles_;
};

// One member's *serf.Serf.
class Serf {
 public:
  // serf.Create(conf): conf.NodeName must name a member of the pool ("node-<id>").  For an id that is not running (beyond
  // Cluster::Options::Initial, or shut down earlier) this is a new process of that name: it starts with its first Join.
  ~Serf() { pool_->Unregister(replica_, id_, this); }
  static std::unique_ptr<Serf> Create(const Config& conf, std::shared_ptr<Cluster> pool, uint32_t id, uint32_t replica = 0) {
    if (!pool || id >= pool->config().n_nodes) throw Error("serf.Create: unknown member", SWIM_ERANGE);
    if (conf.UserEventSizeLimit > 9 * 1024) throw Error("serf.Create: user event size limit exceeds limit of 9216 bytes", SWIM_EINVAL);
    return std::unique_ptr<Serf>(new Serf(conf, std::move(pool), id, replica));
  }

  // Members(): every member this node knows, with its serf status.  A member the reaper erased (handleReap: Failed for
  // longer than ReconnectTimeout / Left for longer than TombstoneTimeout, looked at every ReapInterval), one that was
  // pruned, and a node this member has never heard of are not listed.
  std::vector<Member> Members() {
    requireNotShutdown("Members");
    std::vector<swim_member> raw(pool_->config().n_nodes);
    size_t n = 0;
    check(swim_members(pool_->handle(), replica_, id_, raw.data(), raw.size(), &n), "swim_members");
    const bool device_reaper = pool_->config().reap_interval_ms != 0;
    const int64_t now = pool_->Now().count();
    const int64_t reap_q = std::max<int64_t>(1, conf_.ReapInterval.count());
    std::vector<Member> out;
    for (size_t i = 0; i < n; i++) {
      const swim_member& r = raw[i];
      MemberStatus st = (MemberStatus)r.status;
      if (st == StatusNone) continue;
      if (!device_reaper && (st == StatusFailed || st == StatusLeft)) {   // the pool runs no reaper: this member's own timeouts
        int64_t limit = st == StatusFailed ? conf_.ReconnectTimeout.count() : conf_.TombstoneTimeout.count();
        if (st == StatusFailed && conf_.ReconnectTimeoutOverride)           // serf.go reap(): the override sees the member (its rc_tm tag)
          limit = conf_.ReconnectTimeoutOverride(makeMember(r.id, st, r.incarnation), conf_.ReconnectTimeout).count();
        int64_t checked = now / reap_q * reap_q;                     // the reaper only looks every ReapInterval
        if (checked - (int64_t)r.state_change_ms > limit) continue;
      }
      out.push_back(makeMember(r.id, st, r.incarnation));
    }
    return out;
  }
  Member LocalMember() {
    swim_member r;
    check(swim_view(pool_->handle(), replica_, id_, id_, &r), "swim_view");
    Member m = makeMember(id_, (MemberStatus)r.status, r.incarnation);
    if (state_ == SerfLeaving) m.Status = StatusLeaving;
    if (state_ == SerfLeft) m.Status = StatusLeft;
    return m;
  }
  int NumNodes() { return (int)Members().size(); }
  SerfState State() const { return state_; }

  // GetCoordinate(): this member's network coordinate; GetCachedCoordinate(name): the coordinate of another member as this one
  // would cache it from that member's acks (RouterSerfCluster, agent/router/router.go:62-67; the simulator hands out the
  // member's current coordinate — the cache of a real agent lags by at most one probe cycle).  false = unknown name.
  coordinate::Coordinate GetCoordinate() { coordinate::Coordinate c; check(swim_coordinate_get(pool_->handle(), replica_, id_, &c), "swim_coordinate_get"); return c; }
  bool GetCachedCoordinate(const std::string& name, coordinate::Coordinate* out) {
    uint32_t id = 0;
    if (!out || !resolve(name, &id)) return false;
    check(swim_coordinate_get(pool_->handle(), replica_, id, out), "swim_coordinate_get");
    return true;
  }

  // Join(existing, ignoreOld): serf.Join — a member that is not running yet (an id beyond Cluster::Options::Initial, or one
  // that was shut down) starts and does the join push-pull with the first address that names a member ("node-<id>" or its
  // 10.x.y.z[:port] address); returns how many of the addresses could be contacted, throws when none could (like serf).
  int Join(const std::vector<std::string>& existing, bool /*ignoreOld*/) {
    if (state_ == SerfShutdown) throw Error("Join: Serf can't Join after Shutdown", SWIM_ESTATE);
    if (state_ != SerfAlive) throw Error("Join: Serf can't Join after Leave or Shutdown", SWIM_ESTATE);
    int contacted = 0; uint32_t via = SWIM_NONE;
    for (auto& a : existing) {
      uint32_t id;
      if (!resolve(a, &id) || id == id_) continue;
      swim_node_info ni;
      if (swim_node_info_get(pool_->handle(), replica_, id, &ni) == SWIM_OK && ni.alive) { contacted++; if (via == SWIM_NONE) via = id; }
    }
    swim_node_info me; check(swim_node_info_get(pool_->handle(), replica_, id_, &me), "swim_node_info_get");
    if (!me.alive) {                                    // a process that starts now
      if (via == SWIM_NONE) throw Error("Join: failed to join any of the " + std::to_string(existing.size()) + " addresses", SWIM_ESTATE);
      // MergeDelegate.NotifyMerge on both ends of the join push-pull (agent/consul/merge.go:34-87): this member looks at what
      // `via` would hand over, `via` (when somebody holds its handle) at this member; an error on either side vetoes the join
      if (conf_.Merge) {
        std::vector<swim_member> raw(pool_->config().n_nodes); size_t n = 0;
        check(swim_members(pool_->handle(), replica_, via, raw.data(), raw.size(), &n), "swim_members");
        std::vector<Member> theirs;
        for (size_t i = 0; i < n; i++) if (raw[i].status != StatusNone) theirs.push_back(makeMember(raw[i].id, (MemberStatus)raw[i].status, raw[i].incarnation));
        const std::string err = conf_.Merge(theirs);
        if (!err.empty()) throw Error("Join: merge canceled: " + err, SWIM_ESTATE);
      }
      if (Serf* peer = static_cast<Serf*>(pool_->HandleOf(replica_, via)))
        if (peer->conf_.Merge) {
          const std::string err = peer->conf_.Merge({ makeMember(id_, StatusAlive, me.incarnation + 1) });
          if (!err.empty()) throw Error("Join: merge canceled by " + Cluster::NodeName(via) + ": " + err, SWIM_ESTATE);
        }
      check(swim_inject_join(pool_->handle(), replica_, &id_, 1, via), "swim_inject_join");
    }
    return contacted;
  }
  // Leave, as serf.Leave does it: (1) the member's own leave intent, stamped clock.Time() (every peer: StatusLeaving — and from now on
  // this member does not refute a leave intent about itself), (2) wait until the intent has used up its transmissions, at most
  // BroadcastTimeout, (3) memberlist's leave (dead{Node == From} => StatusLeft at every peer), (4) LeavePropagateDelay
  // (internal/gossip/libserf/serf.go:29-35; consumer agent/consul/server_serf.go:270-297).
  void Leave() {
    if (state_ == SerfLeft) return;
    if (state_ == SerfLeaving) throw Error("Leave: Leave already in progress", SWIM_ESTATE);
    if (state_ == SerfShutdown) throw Error("Leave: Leave called after Shutdown", SWIM_ESTATE);
    state_ = SerfLeaving;
    uint64_t lt = 0;
    if (pool_->config().flags & SWIM_F_SERF_EVENTS) {
      check(swim_force_leave(pool_->handle(), replica_, id_, id_, 0, &lt), "swim_force_leave");
      const Duration step = roundUp(Duration(pool_->config().gossip_interval_ms));
      for (Duration waited{0}; waited < conf_.BroadcastTimeout; waited += step) {        // notifyCh: the broadcast is finished
        int queued = 0; check(swim_event_queued(pool_->handle(), replica_, id_, SWIM_INTENT_LEAVE | id_, lt, &queued), "swim_event_queued");
        if (!queued) break;                               // (the intent itself, not an empty queue: under steady user-event traffic that never comes — ADVICE r4)
        pool_->Advance(step);
      }
    }
    check(swim_inject_leave(pool_->handle(), replica_, &id_, 1), "swim_inject_leave");
    pool_->Advance(roundUp(conf_.LeavePropagateDelay));
    state_ = SerfLeft;
  }
  void Shutdown() {
    if (state_ == SerfShutdown) return;
    check(swim_inject_kill(pool_->handle(), replica_, &id_, 1), "swim_inject_kill");
    state_ = SerfShutdown;
  }
  // UserEvent(name, payload, coalesce): size check first, as serf does
  void UserEvent(const std::string& name, const std::vector<uint8_t>& payload, bool coalesce) {
    requireNotShutdown("UserEvent");
    if ((int)(name.size() + payload.size()) > conf_.UserEventSizeLimit)
      throw Error("user event exceeds configured limit of " + std::to_string(conf_.UserEventSizeLimit) + " bytes before encoding", SWIM_EINVAL);
    uint32_t id = eventId(name, payload); uint64_t lt = 0;
    catalog()[id] = { name, payload, coalesce };
    check(swim_user_event(pool_->handle(), replica_, id_, id, &lt), "swim_user_event");
  }
  void SetTags(const std::map<std::string, std::string>& tags) {
    requireNotShutdown("SetTags");
    conf_.Tags = tags; pool_->SetTags(replica_, id_, tags);
    check(swim_inject_update(pool_->handle(), replica_, &id_, 1), "swim_inject_update");   // memberlist.UpdateNode
  }
  // RemoveFailedNode / RemoveFailedNodePrune: a Lamport-clocked leave intent on behalf of `node`, gossiped to the pool
  // (agent/consul/client.go:272-274); this member applies it at once
  void RemoveFailedNode(const std::string& node) { forceLeave(node, false); }
  void RemoveFailedNodePrune(const std::string& node) { forceLeave(node, true); }
  std::map<std::string, std::string> Stats() {
    swim_stats_t st; check(swim_stats(pool_->handle(), &st), "swim_stats");
    auto ms = Members();
    size_t failed = 0, left = 0;
    for (auto& m : ms) { failed += m.Status == StatusFailed; left += m.Status == StatusLeft; }
    swim_node_info ni; check(swim_node_info_get(pool_->handle(), replica_, id_, &ni), "swim_node_info_get");
    return { { "members", std::to_string(ms.size()) }, { "failed", std::to_string(failed) }, { "left", std::to_string(left) },
             { "health_score", std::to_string(ni.awareness) }, { "event_time", std::to_string(ni.event_clock) },
             { "event_queue", std::to_string(ni.event_queue_len) }, { "intent_queue", std::to_string(ni.queue_len) },
             { "encrypted", "false" } };
  }

  // Config.EventCh: bounded; drained by the caller.  Returns false when empty.  A full channel blocks Serf
  // in the reference (agent/consul/server.go:112-113); here the backlog is reported by EventBacklog().
  bool PollEvent(Event* out) {
    pump();
    if (ch_.empty()) return false;
    *out = ch_.front(); ch_.pop_front();
    return true;
  }
  size_t EventBacklog() { pump(); return ch_.size(); }
  bool EventChFull() { pump(); return ch_.size() >= conf_.EventChCap; }

 private:
  struct Fired { std::string name; std::vector<uint8_t> payload; bool coalesce; };
  Serf(const Config& c, std::shared_ptr<Cluster> p, uint32_t id, uint32_t r) : conf_(c), pool_(std::move(p)), id_(id), replica_(r) {
    if (conf_.NodeName.empty()) conf_.NodeName = Cluster::NodeName(id);
    if (!conf_.Tags.empty()) pool_->SetTags(replica_, id_, conf_.Tags);
    pool_->Register(replica_, id_, this);
    exact_ = pool_->WatchEvents(replica_, id_);        // an EventCh of its own on the device (up to SWIM_EVENT_WATCHERS per pool replica)
  }
  void forceLeave(const std::string& node, bool prune) {
    requireNotShutdown("RemoveFailedNode");
    uint64_t lt = 0;
    check(swim_force_leave(pool_->handle(), replica_, id_, idOf(node), prune ? 1 : 0, &lt), "swim_force_leave");
  }
  // "node-7", "10.0.0.7" or "10.0.0.7:8301" -> 7
  bool resolve(const std::string& a, uint32_t* id) const {
    try {
      if (a.rfind("node-", 0) == 0) { *id = (uint32_t)std::stoul(a.substr(5)); return *id < pool_->config().n_nodes; }
      unsigned b0, b1, b2, b3;
      if (std::sscanf(a.c_str(), "%u.%u.%u.%u", &b0, &b1, &b2, &b3) == 4 && b0 == 10) { *id = (b1 << 16) | (b2 << 8) | b3; return *id < pool_->config().n_nodes; }
    } catch (...) {}
    return false;
  }
  void requireNotShutdown(const char* what) const {
    if (state_ == SerfShutdown) throw Error(std::string(what) + ": Serf is shut down", SWIM_ESTATE);
  }
  Duration roundUp(Duration d) const {
    int64_t q = pool_->derived().quantum_ms;
    return Duration((d.count() + q - 1) / q * q);
  }
  Member makeMember(uint32_t id, MemberStatus st, uint32_t inc) const {
    Member m; m.id = id; m.Name = Cluster::NodeName(id); m.Addr = Cluster::NodeAddr(id); m.Port = 8301; m.Status = st; m.Incarnation = inc;
    m.Tags = pool_->TagsOf(replica_, id);              // every member carries its tags (a registry on the host, keyed by member)
    return m;
  }
  static uint32_t idOf(const std::string& name) {
    if (name.rfind("node-", 0) != 0) throw Error("unknown member " + name, SWIM_ERANGE);
    return (uint32_t)std::stoul(name.substr(5));
  }
  static uint32_t eventId(const std::string& name, const std::vector<uint8_t>& payload) {   // FNV-1a
    uint32_t h = 2166136261u;
    for (unsigned char c : name) { h ^= c; h *= 16777619u; }
    h ^= 0xFF; h *= 16777619u;
    for (unsigned char c : payload) { h ^= c; h *= 16777619u; }
    return (h ^ (h >> 30)) & SWIM_EVENT_ID_MAX;        // 30 bits (xor-folded): bits 31-30 of an id word mark serf's intents
  }
  static std::map<uint32_t, Fired>& catalog() { static std::map<uint32_t, Fired> c; return c; }
  // A member other than the pool's watch node: the device records no event stream for it, so its MEMBER events are derived
  // from its own member list, one snapshot per pump() (= per PollEvent / EventBacklog / EventChFull call): join (first seen, or
  // back from Failed / Left), failed, leave, reap (gone from the list).  Coarser than the watch node's stream, which is exact tick
  // by tick: a transition that happens and reverts between two calls is not seen; EventMemberUpdate and user events are not
  // derived (an incarnation bump may be a refutation as well as a tag change).  The first snapshot is the baseline.
  void pump_by_diff() {
    if (state_ == SerfShutdown) return;
    std::vector<swim_member> raw(pool_->config().n_nodes);
    size_t n = 0;
    check(swim_members(pool_->handle(), replica_, id_, raw.data(), raw.size(), &n), "swim_members");
    std::map<uint32_t, std::pair<MemberStatus, uint32_t>> now;
    for (size_t i = 0; i < n; i++) if (raw[i].status != StatusNone && raw[i].id != id_) now[raw[i].id] = { (MemberStatus)raw[i].status, raw[i].incarnation };
    if (diff_primed_) {
      const Duration at = pool_->Now();
      auto emit = [&](EventType t, uint32_t id, MemberStatus st, uint32_t inc) { Event e; e.Type = t; e.At = at; e.Members.push_back(makeMember(id, st, inc)); ch_.push_back(std::move(e)); };
      for (const auto& kv : now) {
        auto was = seen_.find(kv.first);
        const MemberStatus st = kv.second.first, old = was == seen_.end() ? StatusNone : was->second.first;
        const bool up = st == StatusAlive || st == StatusLeaving, was_up = old == StatusAlive || old == StatusLeaving;
        if (up && !was_up) emit(EventMemberJoin, kv.first, StatusAlive, kv.second.second);
        else if (st == StatusFailed && old != StatusFailed) emit(EventMemberFailed, kv.first, StatusFailed, kv.second.second);
        else if (st == StatusLeft && old != StatusLeft) emit(EventMemberLeave, kv.first, StatusLeft, kv.second.second);
      }
      for (const auto& kv : seen_) if (!now.count(kv.first)) emit(EventMemberReap, kv.first, StatusNone, kv.second.second);
    }
    seen_ = std::move(now); diff_primed_ = true;
  }
  // swim_poll_events -> EventCh.  The device records the events of the pool's watch node and of every handle registered with
  // swim_watch_events (the first SWIM_EVENT_WATCHERS per pool replica), tick-exact; the pool routes them by observer, so a handle
  // never swallows another handle's (or another replica's) events.  A handle beyond that number derives its member events.
  void pump() {
    if (!exact_) { pump_by_diff(); return; }
    pool_->Route();
    swim_event ev;
    while (pool_->NextEvent(replica_, id_, &ev)) {
      Event e; e.Type = (EventType)ev.type; e.At = Duration(ev.time_ms);
      if (e.Type == EventUser) {
        auto it = catalog().find(ev.node);
        e.LTime = ev.ltime;
        if (it != catalog().end()) { e.Name = it->second.name; e.Payload = it->second.payload; e.Coalesce = it->second.coalesce; }
      } else {
        MemberStatus st = e.Type == EventMemberFailed ? StatusFailed : e.Type == EventMemberLeave ? StatusLeft
                        : e.Type == EventMemberReap ? StatusNone : StatusAlive;
        e.Members.push_back(makeMember(ev.node, st, ev.incarnation));
      }
      ch_.push_back(std::move(e));
    }
  }

  Config conf_;
  std::shared_ptr<Cluster> pool_;
  uint32_t id_, replica_;
  SerfState state_ = SerfAlive;
  bool exact_ = false;                       // the device records this member's events (else: derived from member-list snapshots)
  std::deque<Event> ch_;
  std::map<uint32_t, std::pair<MemberStatus, uint32_t>> seen_; bool diff_primed_ = false;   // pump_by_diff: the previous snapshot
};

}  // namespace serf
}  // namespace swimsim

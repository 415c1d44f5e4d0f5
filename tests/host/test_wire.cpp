// tests/host/test_wire.cpp — include/swimsim_wire.hpp against byte vectors derived by hand from the msgpack spec and
// memberlist's struct definitions (net.go) — the real encoder cannot be run here (no Go, modules absent) — plus an
// end-to-end pass over the transport bridge of whichever library exports the C-ABI.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/swimsim.h"
#include "../../include/swimsim_wire.hpp"

using namespace swimsim::wire;

static int failures = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); failures++; } } while (0)

static Bytes hex(const char* s) {
  Bytes b;
  for (; *s; s++) { if (*s == ' ') continue; unsigned v; sscanf(s, "%2x", &v); b.push_back(uint8_t(v)); s++; }
  return b;
}
static Bytes uint_bytes(uint64_t v) { Bytes b; Writer w{b}; w.uint(v); return b; }

int main() {
  // ---- integers take the smallest encoding (go-msgpack EncodeUint)
  CHECK(uint_bytes(0) == hex("00")); CHECK(uint_bytes(127) == hex("7f")); CHECK(uint_bytes(128) == hex("cc 80"));
  CHECK(uint_bytes(255) == hex("cc ff")); CHECK(uint_bytes(256) == hex("cd 01 00")); CHECK(uint_bytes(65536) == hex("ce 00 01 00 00"));
  CHECK(uint_bytes(1ull << 32) == hex("cf 00 00 00 01 00 00 00 00"));

  // ---- suspect{Incarnation: 1, Node: "node-17", From: "node-3"}: type byte, fixmap(3), fixraw keys in field order
  const Bytes suspect = hex("03 83"
                            " ab 49 6e 63 61 72 6e 61 74 69 6f 6e 01"
                            " a4 4e 6f 64 65 a7 6e 6f 64 65 2d 31 37"
                            " a4 46 72 6f 6d a6 6e 6f 64 65 2d 33");
  CHECK(encode_suspect(Suspect{1, "node-17", "node-3"}) == suspect);
  { Suspect s = decode_suspect(suspect.data() + 1, suspect.size() - 1); CHECK(s.incarnation == 1 && s.node == "node-17" && s.from == "node-3"); }
  // dead{} is the same struct under message type 5
  { Bytes d = encode_dead(Dead{1, "node-17", "node-3"}); CHECK(d[0] == 5 && Bytes(d.begin() + 1, d.end()) == Bytes(suspect.begin() + 1, suspect.end())); }

  // ---- alive{Incarnation: 300, Node: "node-5", Addr: 10.0.0.5, Port: 8301, Meta: nil, Vsn: [1 5 2 2 5 4]}
  const Bytes alive = hex("04 86"
                          " ab 49 6e 63 61 72 6e 61 74 69 6f 6e cd 01 2c"
                          " a4 4e 6f 64 65 a6 6e 6f 64 65 2d 35"
                          " a4 41 64 64 72 a4 0a 00 00 05"
                          " a4 50 6f 72 74 cd 20 6d"
                          " a4 4d 65 74 61 c0"
                          " a3 56 73 6e a6 01 05 02 02 05 04");
  { Alive a; a.incarnation = 300; a.node = "node-5"; a.addr = {10, 0, 0, 5}; a.port = 8301; a.vsn = {1, 5, 2, 2, 5, 4}; CHECK(encode(a) == alive);
    Alive d = decode_alive(alive.data() + 1, alive.size() - 1);
    CHECK(d.incarnation == 300 && d.node == "node-5" && d.addr == a.addr && d.port == 8301 && d.meta.empty() && d.vsn == a.vsn); }

  // ---- serf's intents ride memberlist's user message: [userMsg 8][serf type 0 = leave / 1 = join][msgpack struct] (serf messages.go)
  //      messageLeave{LTime: 9, Node: "node-7", Prune: false}; messageJoin{LTime: 300, Node: "node-7"}
  { SerfIntent lv; lv.ltime = 9; lv.node = "node-7";
    CHECK(encode(lv) == hex("08 00 83 a5 4c 54 69 6d 65 09 a4 4e 6f 64 65 a6 6e 6f 64 65 2d 37 a5 50 72 75 6e 65 c2"));
    SerfIntent jn; jn.join = true; jn.ltime = 300; jn.node = "node-7";
    CHECK(encode(jn) == hex("08 01 82 a5 4c 54 69 6d 65 cd 01 2c a4 4e 6f 64 65 a6 6e 6f 64 65 2d 37"));
    // ... and back: the simulator's event ids (SWIM_INTENT_*), never a user event
    size_t control = 0, foreign = 0;
    std::vector<swim_edge> a = from_packet(encode(lv), Naming(), &control, &foreign), b = from_packet(encode(jn), Naming(), &control, &foreign);
    CHECK(a.size() == 1 && a[0].subject == (SWIM_INTENT_LEAVE | 7u) && a[0].incarnation == 9 && (a[0].meta >> 30) == SWIM_MSG_USER);
    CHECK(b.size() == 1 && b[0].subject == (SWIM_INTENT_JOIN | 7u) && b[0].incarnation == 300);
    lv.prune = true; a = from_packet(encode(lv)); CHECK(a.size() == 1 && a[0].subject == (SWIM_INTENT_LEAVE | SWIM_INTENT_PRUNE | 7u));
    CHECK(to_wire(a[0]) == encode(lv) && to_wire(b[0]) == encode(jn)); }

  // ---- nackResp{SeqNo: 42}; ackResp{SeqNo: 70000, Payload: nil}; ping with the omitempty Source* fields absent / present
  CHECK(encode(NackResp{42}) == hex("0b 81 a5 53 65 71 4e 6f 2a"));
  CHECK(encode(AckResp{70000, {}}) == hex("02 82 a5 53 65 71 4e 6f ce 00 01 11 70 a7 50 61 79 6c 6f 61 64 c0"));
  { Ping p; p.seq_no = 7; p.node = "node-2"; CHECK(encode(p) == hex("00 82 a5 53 65 71 4e 6f 07 a4 4e 6f 64 65 a6 6e 6f 64 65 2d 32"));
    p.source_addr = {10, 0, 0, 1}; p.source_port = 8301; p.source_node = "node-1";
    Bytes b = encode(p); CHECK(b[1] == 0x85);
    Ping q = decode_ping(b.data() + 1, b.size() - 1); CHECK(q.seq_no == 7 && q.node == "node-2" && q.source_port == 8301 && q.source_node == "node-1" && q.source_addr == p.source_addr); }
  { IndirectPing ip; ip.seq_no = 9; ip.target = {10, 0, 0, 9}; ip.port = 8301; ip.node = "node-9"; ip.nack = true;
    Bytes b = encode(ip); CHECK(b[0] == 1 && b[1] == 0x85 && b.back() == 0xc3); }

  // ---- serf user event inside a memberlist user message
  const Bytes uev = hex("08 03 84 a5 4c 54 69 6d 65 05 a4 4e 61 6d 65 a7 73 77 69 6d 73 69 6d"
                        " a7 50 61 79 6c 6f 61 64 a4 00 00 00 09 a2 43 43 c2");
  { UserEvent u; u.ltime = 5; u.name = "swimsim"; u.payload = {0, 0, 0, 9}; CHECK(encode(u) == uev);
    UserEvent d = decode_user_event(uev.data() + 2, uev.size() - 2); CHECK(d.ltime == 5 && d.name == "swimsim" && d.payload == u.payload && !d.cc); }

  // ---- strings of 32 bytes and more use raw16 (the old spec has no str8); the decoder also accepts the new dialect
  { std::string long_name(40, 'x'); Bytes b = encode_suspect(Suspect{1, long_name, "n"});
    size_t at = 2 + 12 + 1 + 5; CHECK(b[at] == 0xda && b[at + 1] == 0 && b[at + 2] == 40);
    CHECK(decode_suspect(b.data() + 1, b.size() - 1).node == long_name);
    Bytes newer = hex("83 ab 49 6e 63 61 72 6e 61 74 69 6f 6e 02 a4 4e 6f 64 65 d9 03 61 62 63 a4 46 72 6f 6d c4 02 78 79");
    Suspect s = decode_suspect(newer.data(), newer.size()); CHECK(s.incarnation == 2 && s.node == "abc" && s.from == "xy"); }
  // unknown fields are skipped, whatever their type
  { Bytes extra = hex("84 a3 4e 65 77 92 01 81 a1 6b c3 ab 49 6e 63 61 72 6e 61 74 69 6f 6e 03 a4 4e 6f 64 65 a1 61 a4 46 72 6f 6d a1 62");
    Suspect s = decode_suspect(extra.data(), extra.size()); CHECK(s.incarnation == 3 && s.node == "a" && s.from == "b"); }

  // ---- compound: [7][n][n x len BE16][payloads]
  { std::vector<Bytes> c = make_compound({suspect, encode(NackResp{42})});
    CHECK(c.size() == 1 && c[0][0] == 7 && c[0][1] == 2 && c[0][2] == 0 && c[0][3] == suspect.size() && c[0][4] == 0 && c[0][5] == 9);
    CHECK(c[0].size() == 2 + 4 + suspect.size() + 9);
    size_t lost = 99; std::vector<Bytes> parts = decode_compound(c[0].data() + 1, c[0].size() - 1, &lost);
    CHECK(lost == 0 && parts.size() == 2 && parts[0] == suspect && parts[1] == encode(NackResp{42}));
    parts = decode_compound(c[0].data() + 1, c[0].size() - 1 - 4, &lost); CHECK(parts.size() == 1 && lost == 1);      // cut short
    std::vector<Bytes> many(300, encode(NackResp{1})); CHECK(make_compound(many).size() == 2 && make_compound(many)[1][1] == 45); }
  // the byte budget of a gossip packet: UDPBufferSize 1400 - 2, each message costs its length + 2
  { std::vector<Bytes> m(100, Bytes(48, 0)); CHECK(fit_compound(m) == 27); CHECK(fit_compound(m, 1400, 6) == 27); CHECK(fit_compound(m, 102) == 2); CHECK(fit_compound(m, 101) == 1); }

  // ---- label and CRC headers
  CHECK(crc32_ieee(reinterpret_cast<const uint8_t*>("123456789"), 9) == 0xCBF43926u);
  { Bytes p = add_crc(suspect); CHECK(p[0] == 12 && p.size() == suspect.size() + 5 && strip_crc(p) == suspect);
    p[7] ^= 1; bool threw = false; try { strip_crc(p); } catch (const DecodeError&) { threw = true; } CHECK(threw);
    Bytes l = add_label(suspect, "dc1"); CHECK(l[0] == 244 && l[1] == 3 && l[2] == 'd'); std::string lab; CHECK(strip_label(l, &lab) == suspect && lab == "dc1");
    CHECK(add_label(suspect, "") == suspect); }

  // ---- rumour records <-> packets
  { Naming nm;
    uint32_t id = 0; CHECK(nm.id_of("node-4000000", &id) && id == 4000000); CHECK(!nm.id_of("node-", &id) && !nm.id_of("server-1", &id) && !nm.id_of("node-1x", &id));
    std::vector<swim_edge> in = { {0, 17, 1, (uint32_t(SWIM_MSG_SUSPECT) << 30) | 3u}, {0, 5, 300, uint32_t(SWIM_MSG_ALIVE) << 30},
                                  {0, 99, 2, (uint32_t(SWIM_MSG_DEAD) << 30) | 99u}, {0, 9, 5, uint32_t(SWIM_MSG_USER) << 30} };
    CHECK(to_wire(in[0], nm) == suspect); CHECK(to_wire(in[1], nm) == alive); CHECK(to_wire(in[3], nm) == uev);
    Bytes pkt = add_label(add_crc(to_packet(in, nm)), "dc1");      // rawSendMsgPacket: CRC first, label outermost
    size_t ctl = 9, foreign = 9; std::vector<swim_edge> out = from_packet(pkt, nm, &ctl, &foreign);
    CHECK(ctl == 0 && foreign == 0 && out.size() == in.size());
    for (size_t i = 0; i < in.size() && i < out.size(); i++) CHECK(out[i].subject == in[i].subject && out[i].incarnation == in[i].incarnation && out[i].meta == in[i].meta);
    std::vector<Bytes> mixed = { encode(NackResp{1}), encode_suspect(Suspect{1, "consul-server-1", "node-1"}), suspect };
    out = from_packet(make_compound(mixed)[0], nm, &ctl, &foreign); CHECK(ctl == 1 && foreign == 1 && out.size() == 1 && out[0].subject == 17); }

  // ---- end to end over the C-ABI bridge: what node 5 would receive as memberlist packets, and its own packet going in
  {
    swim_config cfg; CHECK(swim_config_preset(&cfg, SWIM_PRESET_LAN) == 0);
    cfg.n_nodes = 64; cfg.seed = 3;
    swim_sim* sim = nullptr;
    int rc = swim_create(&cfg, &sim);
    if (rc == SWIM_ENODEV) printf("no device: bridge leg skipped\n");
    else {
      CHECK(rc == 0);
      std::vector<swim_edge> got(4096); size_t n = 0;
      CHECK(swim_transport_poll(sim, 0, 5, got.data(), got.size(), &n) == 0 && n == 0);     // attaches node 5
      uint32_t victim = 9; CHECK(swim_inject_kill(sim, 0, &victim, 1) == 0);
      size_t seen = 0;
      for (int t = 0; t < 400 && !seen; t++) {
        CHECK(swim_step(sim, 1) == 0);
        CHECK(swim_transport_poll(sim, 0, 5, got.data(), got.size(), &n) == 0);
        for (size_t i = 0; i < n; i++) {
          Bytes m = to_wire(got[i]); CHECK(!m.empty());
          std::vector<swim_edge> back = from_packet(m);
          CHECK(back.size() == 1 && back[0].subject == got[i].subject && back[0].incarnation == got[i].incarnation && back[0].meta == got[i].meta);
          if (got[i].subject == victim) seen++;
        }
      }
      CHECK(seen > 0);                               // the suspicion of node 9 reached the attached node as a packet
      // the real node refutes nothing and instead confirms: its suspect{} packet goes to node 6
      Bytes pkt = encode_suspect(Suspect{1, "node-9", "node-5"});
      std::vector<swim_edge> recs = from_packet(pkt);
      CHECK(recs.size() == 1 && swim_transport_write_to(sim, 0, 5, 6, recs.data(), recs.size()) == 0);
      CHECK(swim_step(sim, 2) == 0);
      swim_member mv; CHECK(swim_view(sim, 0, 6, 9, &mv) == 0 && mv.state != SWIM_STATE_ALIVE);
      // the same through the byte-level Transport: packets with CRC and label, as a real memberlist node would see them
      BridgeTransport tr(sim, 0, 5, Naming(), "dc1", true);
      uint32_t v2 = 20; CHECK(swim_inject_kill(sim, 0, &v2, 1) == 0);
      size_t pkts = 0, about_v2 = 0;
      for (int t = 0; t < 400 && !about_v2; t++) {
        CHECK(swim_step(sim, 1) == 0);
        for (const BridgeTransport::Packet& pk : tr.Poll()) {
          pkts++;
          CHECK(pk.buf[0] == kHasLabel);
          std::string lab; Bytes inner = strip_crc(strip_label(pk.buf, &lab)); CHECK(lab == "dc1" && !inner.empty());
          for (const swim_edge& e : from_packet(pk.buf)) if (e.subject == v2) about_v2++;
          CHECK(pk.from_id == SWIM_NONE || pk.from == "node-" + std::to_string(pk.from_id));
        }
      }
      CHECK(pkts > 0 && about_v2 > 0);
      CHECK(tr.WriteTo(add_label(add_crc(encode_suspect(Suspect{1, "node-20", "node-5"})), "dc1"), "10.0.0.7:8301") == 0);
      CHECK(tr.WriteTo(encode(Ping{1, "node-7", {}, 0, ""}), "node-7") == 0 && tr.control_messages_seen() == 1);
      CHECK(tr.WriteTo(suspect, "192.168.0.1:8301") == SWIM_EINVAL);
      // a real node attached through the bridge probes its virtual peers — the bridge answers for them.
      {
        auto polled = tr.Poll();
        bool got_ack = false;
        for (const BridgeTransport::Packet& pk : polled) {
          Bytes inner = strip_crc(strip_label(pk.buf, nullptr));
          if (inner[0] == kAckResp) { AckResp a = decode_ack(inner.data() + 1, inner.size() - 1); got_ack = a.seq_no == 1 && pk.from == "node-7"; }
        }
        CHECK(got_ack && tr.probes_answered() == 1);                        // ping{SeqNo 1} to node-7 -> ackResp{SeqNo 1} from node-7
        CHECK(tr.WriteTo(encode(Ping{2, "node-20", {}, 0, ""}), "node-20") == 0);   // node 20 is dead: no answer, the probe times out
        CHECK(tr.WriteTo(encode(Ping{3, "node-8", {}, 0, ""}), "node-7") == 0);     // "got ping for unexpected node": dropped
        IndirectPing ip; ip.seq_no = 4; ip.target = Bytes{10, 0, 0, 20}; ip.port = 8301; ip.node = "node-20"; ip.nack = true;
        CHECK(tr.WriteTo(encode(ip), "node-7") == 0);                       // relay node-7 alive, target node-20 dead -> nackResp
        ip.seq_no = 5; ip.target = Bytes{10, 0, 0, 8}; ip.node = "node-8";
        CHECK(tr.WriteTo(encode(ip), "node-7") == 0);                       // relay and target alive -> ackResp
        size_t acks = 0, nacks = 0;
        for (const BridgeTransport::Packet& pk : tr.Poll()) {
          Bytes inner = strip_crc(strip_label(pk.buf, nullptr));
          if (inner[0] == kAckResp && decode_ack(inner.data() + 1, inner.size() - 1).seq_no == 5) acks++;
          if (inner[0] == kNackResp) nacks++;
          CHECK(!(inner[0] == kAckResp && decode_ack(inner.data() + 1, inner.size() - 1).seq_no == 2));
        }
        CHECK(acks == 1 && nacks == 1 && tr.control_messages_seen() == 5);   // counters accumulate over calls
        // encryptMsg carries rumours this codec cannot see: refused, never swallowed; a malformed compressMsg likewise
        CHECK(tr.WriteTo(Bytes{kCompress, 0x81, 0xa4}, "node-7") == SWIM_EINVAL && tr.WriteTo(Bytes{kEncrypt, 1, 2, 3}, "node-7") == SWIM_EINVAL);
        CHECK(tr.unsupported_packets_seen() == 2);
        // a compressed packet (memberlist's default: EnableCompression) is opened: the suspicion inside reaches node 7
        {
          std::vector<Bytes> many;
          for (int k = 0; k < 6; k++) many.push_back(encode_suspect(Suspect{1, "node-21", "node-5"}));     // repetitive enough to shrink
          Bytes compound = make_compound(many)[0], packed = maybe_compress(compound);
          CHECK(packed[0] == kCompress && packed.size() < compound.size());
          CHECK(decompress(packed.data() + 1, packed.size() - 1) == compound);
          swim_member before; CHECK(swim_view(sim, 0, 7, 21, &before) == 0 && before.state == SWIM_STATE_ALIVE);
          CHECK(tr.WriteTo(add_label(add_crc(packed), ""), "node-7") == 0);
          CHECK(swim_step(sim, 1) == 0);
          swim_member after; CHECK(swim_view(sim, 0, 7, 21, &after) == 0 && after.state == SWIM_STATE_SUSPECT);
        }
        // the stream side: the real node dials node-7 and pushes its state (memberlist.Join / the periodic push-pull); node 7 merges
        // it the way mergeState does and answers with its whole member list
        {
          BridgeTransport tp(sim, 0, 5, Naming(), "dc1", true, true, cfg.n_nodes);
          PushPull mine; mine.join = true;
          auto st = [](const char* name, uint32_t inc, uint32_t state) { PushNodeState n; n.name = name; n.addr = Bytes{10, 0, 0, 1}; n.port = 8301; n.incarnation = inc; n.state = state; n.vsn = Bytes{1, 5, 2, 2, 5, 4}; return n; };
          mine.nodes = { st("node-5", 3, SWIM_STATE_ALIVE), st("node-22", 1, SWIM_STATE_DEAD), st("node-23", 1, SWIM_STATE_LEFT), st("consul-server-9", 1, SWIM_STATE_ALIVE) };
          Bytes reply = tp.PushPull(to_stream(mine, "dc1", true), "node-7");
          CHECK(!reply.empty() && reply[0] == kHasLabel && tp.push_pulls_answered() == 1 && tp.foreign_names_seen() == 1);
          std::string lab; PushPull theirs = from_stream(reply, &lab);
          CHECK(lab == "dc1" && !theirs.join && theirs.nodes.size() == cfg.n_nodes);
          CHECK(!theirs.user_state.empty() && theirs.user_state[0] == kSerfPushPull);          // serf's delegate state: event clock, left members
          SerfPushPull sst = decode_serf_push_pull(theirs.user_state.data() + 1, theirs.user_state.size() - 1);
          CHECK(sst.ltime == sst.event_ltime && sst.events.empty() && sst.status_ltimes.empty() && sst.left_members.empty());
          size_t dead = 0, me = 0;
          for (const PushNodeState& n : theirs.nodes) {
            if (n.name == "node-20" || n.name == "node-9") dead += n.state != SWIM_STATE_ALIVE;
            if (n.name == "node-5") me += n.state == SWIM_STATE_ALIVE && n.addr == (Bytes{10, 0, 0, 5}) && n.port == 8301;
          }
          CHECK(dead == 2 && me == 1);                                       // what node 7 knows: the two failures, and the attached node alive
          CHECK(swim_step(sim, 1) == 0);
          swim_member m22, m23, m5;
          CHECK(swim_view(sim, 0, 7, 22, &m22) == 0 && m22.state == SWIM_STATE_SUSPECT);      // a remote Dead is only a suspicion here
          CHECK(swim_view(sim, 0, 7, 23, &m23) == 0 && m23.state == SWIM_STATE_LEFT);
          CHECK(swim_view(sim, 0, 7, 5, &m5) == 0 && m5.state == SWIM_STATE_ALIVE && m5.incarnation == 3);
          CHECK(tp.PushPull(to_stream(mine, "dc1", false), "node-20").empty());              // node 20 is down: the dial fails
          bool threw = false;
          try { tp.PushPull(Bytes{kHasLabel, 3, 'd', 'c', '1', kSuspect, 0x80}, "node-7"); } catch (const DecodeError&) { threw = true; }
          CHECK(threw);                                                      // not a push-pull message
        }
      }
      CHECK(swim_step(sim, 2) == 0);
      CHECK(swim_view(sim, 0, 7, 20, &mv) == 0 && mv.state != SWIM_STATE_ALIVE);
      printf("backend %s\n", swim_backend());
      swim_destroy(sim);
    }
  }
  {   // push-pull on the stream: header, node states back to back, the delegate's bytes behind them
    PushPull pp; pp.join = true; pp.user_state = Bytes{9, 8, 7};
    PushNodeState a; a.name = "node-1"; a.addr = Bytes{10, 0, 0, 1}; a.port = 8301; a.incarnation = 7; a.state = 0; a.vsn = Bytes{1, 5, 2, 2, 5, 4};
    PushNodeState b = a; b.name = "node-2"; b.state = 3; b.meta = Bytes{1, 2};
    pp.nodes = { a, b };
    Bytes w = encode(pp);
    const Bytes head{kPushPull, 0x83, 0xa5, 'N', 'o', 'd', 'e', 's', 0x02, 0xac, 'U', 's', 'e', 'r', 'S', 't', 'a', 't', 'e', 'L', 'e', 'n', 0x03, 0xa4, 'J', 'o', 'i', 'n', 0xc3, 0x87, 0xa4, 'N', 'a', 'm', 'e'};
    CHECK(w.size() > head.size() && Bytes(w.begin(), w.begin() + head.size()) == head && Bytes(w.end() - 3, w.end()) == pp.user_state);
    PushPull back = decode_push_pull(w.data() + 1, w.size() - 1);
    CHECK(back.join && back.nodes.size() == 2 && back.nodes[1].name == "node-2" && back.nodes[1].state == 3 && back.nodes[1].meta == b.meta &&
          back.nodes[0].addr == a.addr && back.nodes[0].port == 8301 && back.nodes[0].incarnation == 7 && back.nodes[0].vsn == a.vsn && back.user_state == pp.user_state);
    for (bool z : { false, true }) { std::string lab; PushPull s2 = from_stream(to_stream(pp, "dc1", z), &lab); CHECK(lab == "dc1" && s2.nodes.size() == 2 && s2.user_state == pp.user_state); }
    auto bad = [](const Bytes& x) { try { decode_push_pull(x.data(), x.size()); } catch (const DecodeError&) { return true; } return false; };
    CHECK(bad(Bytes(w.begin() + 1, w.end() - 5)));                           // the delegate's bytes cut short
    CHECK(bad(Bytes{0x81, 0xa5, 'N', 'o', 'd', 'e', 's', 0xce, 0x7f, 0xff, 0xff, 0xff}));   // two billion nodes promised, none there
  }
  {   // serf's own push-pull message, the bytes memberlist carries as "user state"
    SerfPushPull sp; sp.ltime = 41; sp.event_ltime = 17; sp.query_ltime = 3; sp.status_ltimes = { {"node-1", 40}, {"node-2", 12} }; sp.left_members = { "node-9" };
    SerfUserEvents ue; ue.ltime = 16; ue.events = { {"deploy", Bytes{1, 2, 3}}, {"x", Bytes{}} }; sp.events = { ue };
    Bytes w = encode(sp);
    const Bytes head{kSerfPushPull, 0x86, 0xa5, 'L', 'T', 'i', 'm', 'e', 41, 0xac, 'S', 't', 'a', 't', 'u', 's', 'L', 'T', 'i', 'm', 'e', 's', 0x82, 0xa6, 'n', 'o', 'd', 'e', '-', '1', 40};
    CHECK(w.size() > head.size() && Bytes(w.begin(), w.begin() + head.size()) == head);
    SerfPushPull back = decode_serf_push_pull(w.data() + 1, w.size() - 1);
    CHECK(back.ltime == 41 && back.event_ltime == 17 && back.query_ltime == 3 && back.status_ltimes == sp.status_ltimes && back.left_members == sp.left_members);
    CHECK(back.events.size() == 1 && back.events[0].ltime == 16 && back.events[0].events == ue.events);
    Bytes nil_slices{0x83, 0xa5, 'L', 'T', 'i', 'm', 'e', 5, 0xab, 'L', 'e', 'f', 't', 'M', 'e', 'm', 'b', 'e', 'r', 's', 0xc0, 0xa6, 'E', 'v', 'e', 'n', 't', 's', 0x91, 0xc0};
    SerfPushPull z = decode_serf_push_pull(nil_slices.data(), nil_slices.size());               // Go's nil slices and nil pointers
    CHECK(z.ltime == 5 && z.left_members.empty() && z.events.empty());
  }
  {   // compress/lzw (LSB, 8-bit literals) as memberlist's compressPayload uses it: round trips, the table-full reset, hostile input
    auto rt = [](const Bytes& in) { Bytes z = lzw_encode(in); return lzw_decode(z.data(), z.size()) == in; };
    CHECK(rt(Bytes{}) && rt(Bytes{42}) && rt(Bytes(1000, 7)));
    Bytes text; for (int i = 0; i < 3000; i++) text.push_back(uint8_t("memberlist"[i % 10]));
    CHECK(rt(text) && lzw_encode(text).size() < text.size() / 5);
    Bytes noise; uint32_t x = 1; for (int i = 0; i < 40000; i++) { x = x * 1664525u + 1013904223u; noise.push_back(uint8_t(x >> 24)); }
    CHECK(rt(noise));                                                        // > 4 096 codes: the encoder clears and starts over
    Bytes kwk; for (int i = 0; i < 500; i++) kwk.push_back('a');             // aaaa...: every code is the one being defined
    CHECK(rt(kwk));
    CHECK(lzw_encode(Bytes{}) == (Bytes{0x00, 0x03, 0x02}));                 // clear (9 bits), eof (9 bits), padded: 0x100 | 0x101 << 9
    auto bad = [](const Bytes& z) { try { lzw_decode(z.data(), z.size()); } catch (const DecodeError&) { return true; } return false; };
    CHECK(bad(Bytes{}) && bad(Bytes{0x00, 0x01}) && bad(Bytes{0x00, 0xff, 0xff, 0xff}));   // truncated; no eof; a code beyond the table
  }
  {   // Reader::skip() on hostile input — truncated map, deep nesting, ext types
    auto throws = [](const Bytes& b) { try { decode_alive(b.data(), b.size()); } catch (const DecodeError&) { return true; } return false; };
    CHECK(throws(Bytes{0x81, 0xa1, 'X'}));                                  // map of 1, unknown key, value missing: one byte past the buffer before
    Bytes deep{0x81, 0xa1, 'X'}; deep.insert(deep.end(), 60000, 0x91); deep.push_back(0x01);
    CHECK(throws(deep));                                                     // 60 000 nested fixarrays: depth cap, no stack overflow
    Bytes ext{0x82, 0xa1, 'X', 0xd6, 0x05, 1, 2, 3, 4, 0xa4, 'N', 'o', 'd', 'e', 0xa1, 'n'};   // fixext4 under an unknown key is skipped
    CHECK(decode_alive(ext.data(), ext.size()).node == "n");
    Bytes arr32{0x82, 0xa1, 'X', 0xdd, 0, 0, 0, 2, 0x01, 0x02, 0xa4, 'N', 'o', 'd', 'e', 0xa1, 'm'};
    CHECK(decode_alive(arr32.data(), arr32.size()).node == "m");
    CHECK(throws(Bytes{0x81, 0xa1, 'X', 0xdd, 0xff, 0xff, 0xff, 0xff}));      // array32 claiming 4 G elements
  }
  if (failures) { printf("%d FAILED\n", failures); return 1; }
  printf("ALL PASSED\n");
  return 0;
}

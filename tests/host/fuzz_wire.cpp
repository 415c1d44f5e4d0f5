// fuzz_wire [iterations] — the wire codec against hostile bytes.  Packets and streams come from a real, possibly remote node:
// whatever arrives, the decoders may only return or throw DecodeError.  Built with -fsanitize=address,undefined by
// tests/test_wire_fuzz.py, so an out-of-bounds read, an overflow or an uncaught exception of another type ends the run.
// Inputs: random bytes with a plausible first byte, and valid messages (gossip compound with CRC and label, compressMsg,
// push-pull stream) with bytes flipped, cut, or spliced.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/swimsim_wire.hpp"

using namespace swimsim::wire;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return uint32_t(rng_state >> 32); }

static Bytes mutate(Bytes b) {
  if (b.empty()) return b;
  switch (rnd() % 5) {
    case 0: for (int k = 0, n = 1 + int(rnd() % 4); k < n; k++) b[rnd() % b.size()] ^= uint8_t(1u << (rnd() % 8)); break;
    case 1: b.resize(rnd() % b.size()); break;
    case 2: for (int k = 0, n = 1 + int(rnd() % 3); k < n; k++) b[rnd() % b.size()] = uint8_t(rnd()); break;
    case 3: { size_t at = rnd() % b.size(), len = rnd() % 64; Bytes junk(len); for (auto& x : junk) x = uint8_t(rnd()); b.insert(b.begin() + at, junk.begin(), junk.end()); break; }
    default: { size_t at = rnd() % b.size(); b[at] = uint8_t(0xdc + rnd() % 4); break; }      // array16/32, map16/32 headers with whatever follows as the count
  }
  return b;
}

int main(int argc, char** argv) {
  const long iters = argc > 1 ? std::atol(argv[1]) : 100000;
  // the seeds: one of each framing
  std::vector<Bytes> seeds;
  std::vector<Bytes> parts = { encode_suspect(Suspect{7, "node-17", "node-3"}), encode(Ping{5, "node-9", Bytes{10, 0, 0, 1}, 8301, "node-1"}), encode(NackResp{3}) };
  { Alive a; a.incarnation = 4; a.node = "node-2"; a.addr = Bytes{10, 0, 0, 2}; a.port = 8301; a.vsn = Bytes{1, 5, 2, 2, 5, 4}; parts.push_back(encode(a)); }
  { UserEvent u; u.ltime = 9; u.name = "swimsim"; u.payload = Bytes{0, 0, 0, 5}; parts.push_back(encode(u)); }
  Bytes compound = make_compound(parts)[0];
  seeds.push_back(add_label(add_crc(compound), "dc1"));
  seeds.push_back(add_crc(encode(Compress{0, lzw_encode(compound)})));
  { PushPull pp; pp.join = true; pp.user_state = Bytes{1, 2, 3};
    for (int i = 0; i < 5; i++) { PushNodeState n; n.name = "node-" + std::to_string(i); n.addr = Bytes{10, 0, 0, uint8_t(i)}; n.port = 8301; n.incarnation = 1 + i; n.state = i % 4; n.vsn = Bytes{1, 5, 2, 2, 5, 4}; pp.nodes.push_back(n); }
    seeds.push_back(to_stream(pp, "dc1", false)); seeds.push_back(to_stream(pp, "", true)); }
  for (const Bytes& p : parts) seeds.push_back(p);
  { SerfPushPull sp; sp.ltime = 9; sp.status_ltimes = { {"node-1", 4} }; sp.left_members = { "node-2" }; SerfUserEvents ue; ue.ltime = 3; ue.events = { {"e", Bytes{1}} }; sp.events = { ue };
    Bytes w = encode(sp); seeds.push_back(Bytes(w.begin() + 1, w.end())); }
  long threw = 0, ok = 0;
  for (long it = 0; it < iters; it++) {
    Bytes in;
    if (it % 4 == 0) { in.resize(1 + rnd() % 96); for (auto& x : in) x = uint8_t(rnd()); in[0] = uint8_t(rnd() % 14); if (rnd() % 8 == 0) in[0] = kHasLabel; }
    else { in = seeds[rnd() % seeds.size()]; for (int k = 0, n = 1 + int(rnd() % 3); k < n; k++) in = mutate(in); }
    try {
      size_t control = 0, foreign = 0; std::vector<Probe> probes;
      (void)from_packet(in, Naming(), &control, &foreign, &probes);
      ok++;
    } catch (const DecodeError&) { threw++; } catch (const std::length_error&) { threw++; }
    try { (void)from_stream(in, nullptr); ok++; } catch (const DecodeError&) { threw++; }
    try { (void)lzw_decode(in.data(), in.size(), 1 << 20); ok++; } catch (const DecodeError&) { threw++; }
    try { (void)decode_serf_push_pull(in.data(), in.size()); ok++; } catch (const DecodeError&) { threw++; }
  }
  // and the seeds themselves still decode
  size_t control = 0, foreign = 0;
  if (from_packet(seeds[0], Naming(), &control, &foreign).size() != 3 || from_packet(seeds[1]).size() != 3 || from_stream(seeds[2]).nodes.size() != 5 || from_stream(seeds[3]).nodes.size() != 5) {
    std::printf("FAIL: a seed no longer decodes\n"); return 1;
  }
  std::printf("fuzz ok: %ld inputs, %ld decoded, %ld refused\n", iters, ok, threw);
  return 0;
}

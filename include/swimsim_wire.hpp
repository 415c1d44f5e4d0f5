// swimsim_wire.hpp — memberlist's packet format for the transport bridge.
//
// swim_transport_poll / swim_transport_write_to exchange rumour records (swim_edge).  A *real* memberlist node
// speaks bytes: a message-type byte followed by a go-msgpack map of the message struct's fields, several messages
// folded into a compound packet, optionally behind a label header and a CRC32 header.  This header turns one into
// the other, so that the Go shim of INTEGRATION.md can hand Transport.WriteTo's buffer straight through and feed
// PacketCh with what comes back.
//
// The format lives in github.com/hashicorp/memberlist v0.6.0 (go.mod:80: net.go messageType / encode /
// makeCompoundMessage / decodeCompoundMessage, util.go, label.go) and github.com/hashicorp/serf v0.10.4
// (go.mod:85: messages.go) with github.com/hashicorp/go-msgpack/v2 v2.1.5 (go.mod:225) underneath — none of them
// vendored under /root/reference.  It is restated here from the published sources; PARITY UNPINNED against the real
// encoder (no Go toolchain), pinned only by hand-derived byte vectors in tests/host/test_wire.cpp.  In-tree
// corroboration: the u32/u16 big-endian framing style of agent/consul/wanfed/wanfed.go:112-121, UDPBufferSize 1400
// (agent/consul/config_test.go:51), the status values of api/agent.go:296-304.
//
// go-msgpack specifics that matter (codec.MsgpackHandle{} as memberlist's encode() constructs it: RawToString and
// WriteExt off): a struct is a map keyed by the Go field names in declaration order; strings AND byte slices use the
// old "raw" family (fixraw 0xa0|n, raw16 0xda, raw32 0xdb — no str8 / bin8); unsigned integers take the smallest of
// positive-fixint / 0xcc / 0xcd / 0xce / 0xcf; a nil slice is 0xc0; bools are 0xc2 / 0xc3; `omitempty` fields vanish.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <map>
#include <string>
#include <vector>

#include "swimsim.h"

namespace swimsim {
namespace wire {

using Bytes = std::vector<uint8_t>;

// memberlist net.go messageType
enum MessageType : uint8_t {
  kPing = 0, kIndirectPing = 1, kAckResp = 2, kSuspect = 3, kAlive = 4, kDead = 5, kPushPull = 6, kCompound = 7,
  kUser = 8, kCompress = 9, kEncrypt = 10, kNackResp = 11, kHasCrc = 12, kErr = 13, kHasLabel = 244,
};
// serf messages.go messageType (the first byte of a memberlist user message)
enum SerfMessageType : uint8_t { kSerfLeave = 0, kSerfJoin = 1, kSerfPushPull = 2, kSerfUserEvent = 3, kSerfQuery = 4 };

struct DecodeError : std::runtime_error { using std::runtime_error::runtime_error; };

// ---------------------------------------------------------------------------------------------------------------
// msgpack, the subset memberlist uses
// ---------------------------------------------------------------------------------------------------------------
struct Writer {
  Bytes& b;
  void u8(uint8_t v) { b.push_back(v); }
  void be16(uint16_t v) { u8(uint8_t(v >> 8)); u8(uint8_t(v)); }
  void be32(uint32_t v) { be16(uint16_t(v >> 16)); be16(uint16_t(v)); }
  void be64(uint64_t v) { be32(uint32_t(v >> 32)); be32(uint32_t(v)); }
  void map(size_t n) { if (n < 16) u8(uint8_t(0x80 | n)); else if (n <= 0xFFFF) { u8(0xde); be16(uint16_t(n)); } else { u8(0xdf); be32(uint32_t(n)); } }
  void array(size_t n) { if (n < 16) u8(uint8_t(0x90 | n)); else if (n <= 0xFFFF) { u8(0xdc); be16(uint16_t(n)); } else { u8(0xdd); be32(uint32_t(n)); } }
  void uint(uint64_t v) {
    if (v < 128) u8(uint8_t(v));
    else if (v <= 0xFF) { u8(0xcc); u8(uint8_t(v)); }
    else if (v <= 0xFFFF) { u8(0xcd); be16(uint16_t(v)); }
    else if (v <= 0xFFFFFFFFull) { u8(0xce); be32(uint32_t(v)); }
    else { u8(0xcf); be64(v); }
  }
  void boolean(bool v) { u8(v ? 0xc3 : 0xc2); }
  void nil() { u8(0xc0); }
  void raw(const void* p, size_t n) {               // strings and []byte alike (old-spec "raw")
    if (n < 32) u8(uint8_t(0xa0 | n)); else if (n <= 0xFFFF) { u8(0xda); be16(uint16_t(n)); } else { u8(0xdb); be32(uint32_t(n)); }
    const uint8_t* q = static_cast<const uint8_t*>(p); b.insert(b.end(), q, q + n);
  }
  void str(const std::string& s) { raw(s.data(), s.size()); }
  void bytes(const Bytes& v, bool nil_when_empty) { if (v.empty() && nil_when_empty) nil(); else raw(v.data(), v.size()); }
};

struct Reader {
  const uint8_t* p; const uint8_t* end;
  Reader(const uint8_t* b, size_t n) : p(b), end(b + n) {}
  size_t left() const { return size_t(end - p); }
  uint8_t u8() { if (p >= end) throw DecodeError("truncated"); return *p++; }
  uint16_t be16() { uint16_t a = u8(); return uint16_t(a << 8 | u8()); }
  uint32_t be32() { uint32_t a = be16(); return a << 16 | be16(); }
  uint64_t be64() { uint64_t a = be32(); return a << 32 | be32(); }
  size_t map() {
    uint8_t t = u8();
    if ((t & 0xF0) == 0x80) return t & 0x0F;
    if (t == 0xde) return be16();
    if (t == 0xdf) return be32();
    throw DecodeError("expected a map");
  }
  size_t array() {
    uint8_t t = u8();
    if (t == 0xc0) return 0;                          // a nil slice
    if ((t & 0xF0) == 0x90) return t & 0x0F;
    if (t == 0xdc) return be16();
    if (t == 0xdd) return be32();
    throw DecodeError("expected an array");
  }
  uint64_t uint() {
    uint8_t t = u8();
    if (t < 0x80) return t;
    switch (t) {
      case 0xcc: return u8(); case 0xcd: return be16(); case 0xce: return be32(); case 0xcf: return be64();
      case 0xd0: return uint64_t(int8_t(u8())); case 0xd1: return uint64_t(int16_t(be16()));
      case 0xd2: return uint64_t(int32_t(be32())); case 0xd3: return be64();
    }
    throw DecodeError("expected an integer");
  }
  bool boolean() { uint8_t t = u8(); if (t == 0xc2) return false; if (t == 0xc3) return true; throw DecodeError("expected a bool"); }
  // raw / str / bin in either msgpack dialect, or nil
  Bytes raw() {
    uint8_t t = u8(); size_t n;
    if (t == 0xc0) return {};
    if ((t & 0xE0) == 0xa0) n = t & 0x1F;
    else if (t == 0xd9 || t == 0xc4) n = u8();
    else if (t == 0xda || t == 0xc5) n = be16();
    else if (t == 0xdb || t == 0xc6) n = be32();
    else throw DecodeError("expected raw bytes");
    if (n > left()) throw DecodeError("truncated");
    Bytes v(p, p + n); p += n; return v;
  }
  std::string str() { Bytes v = raw(); return std::string(v.begin(), v.end()); }
  // any value.  Packets come from a real, possibly remote node: every read is bounds-checked (u8/be16/... throw on a
  // short buffer) and nesting is capped — a 64 KB packet of 0x91 bytes would otherwise recurse 64 K frames deep.
  void skip(int depth = 0) {
    if (depth > 32) throw DecodeError("msgpack nesting too deep");
    if (left() < 1) throw DecodeError("truncated");
    uint8_t t = *p;
    if (t < 0x80 || t >= 0xe0 || t == 0xc0 || t == 0xc2 || t == 0xc3) { u8(); return; }
    if ((t & 0xE0) == 0xa0 || t == 0xd9 || t == 0xda || t == 0xdb || t == 0xc4 || t == 0xc5 || t == 0xc6) { raw(); return; }
    if ((t & 0xF0) == 0x80 || t == 0xde || t == 0xdf) { size_t n = map(); for (size_t i = 0; i < 2 * n; i++) skip(depth + 1); return; }
    if ((t & 0xF0) == 0x90) { u8(); for (size_t i = 0; i < size_t(t & 0x0F); i++) skip(depth + 1); return; }
    if (t == 0xdc) { u8(); size_t n = be16(); for (size_t i = 0; i < n; i++) skip(depth + 1); return; }
    if (t == 0xdd) { u8(); size_t n = be32(); if (n > left()) throw DecodeError("truncated"); for (size_t i = 0; i < n; i++) skip(depth + 1); return; }
    if (t >= 0xcc && t <= 0xd3) { uint(); return; }
    if (t == 0xca) { u8(); be32(); return; }
    if (t == 0xcb) { u8(); be64(); return; }
    // ext family: fixext 1/2/4/8/16, ext 8/16/32 — [type byte] + payload, skipped like go-msgpack does for unknown fields
    if (t >= 0xd4 && t <= 0xd8) { u8(); size_t n = size_t(1) << (t - 0xd4); if (n + 1 > left()) throw DecodeError("truncated"); p += n + 1; return; }
    if (t == 0xc7 || t == 0xc8 || t == 0xc9) { u8(); size_t n = t == 0xc7 ? u8() : t == 0xc8 ? be16() : be32(); if (n + 1 > left()) throw DecodeError("truncated"); p += n + 1; return; }
    throw DecodeError("unsupported msgpack type");
  }
};

// ---------------------------------------------------------------------------------------------------------------
// memberlist message structs (net.go), fields in declaration order
// ---------------------------------------------------------------------------------------------------------------
struct Alive { uint32_t incarnation = 0; std::string node; Bytes addr; uint16_t port = 0; Bytes meta; Bytes vsn; };
struct Suspect { uint32_t incarnation = 0; std::string node, from; };
using Dead = Suspect;                                // dead{Incarnation, Node, From}: Node == From means "left"
struct Ping { uint32_t seq_no = 0; std::string node; Bytes source_addr; uint16_t source_port = 0; std::string source_node; };
struct IndirectPing { uint32_t seq_no = 0; Bytes target; uint16_t port = 0; std::string node; bool nack = false;
                      Bytes source_addr; uint16_t source_port = 0; std::string source_node; };
struct AckResp { uint32_t seq_no = 0; Bytes payload; };
struct NackResp { uint32_t seq_no = 0; };
// serf messages.go messageUserEvent
struct UserEvent { uint64_t ltime = 0; std::string name; Bytes payload; bool cc = false; };
// serf messages.go messageLeave {LTime, Node, Prune} / messageJoin {LTime, Node}: the intents (SWIM_INTENT_* event ids in the simulator)
struct SerfIntent { bool join = false; uint64_t ltime = 0; std::string node; bool prune = false; };

inline Bytes encode(const Alive& m) {
  Bytes b{kAlive}; Writer w{b};
  w.map(6);
  w.str("Incarnation"); w.uint(m.incarnation); w.str("Node"); w.str(m.node); w.str("Addr"); w.bytes(m.addr, true);
  w.str("Port"); w.uint(m.port); w.str("Meta"); w.bytes(m.meta, true); w.str("Vsn"); w.bytes(m.vsn, true);
  return b;
}
inline Bytes encode_suspect_like(uint8_t type, const Suspect& m) {
  Bytes b{type}; Writer w{b};
  w.map(3); w.str("Incarnation"); w.uint(m.incarnation); w.str("Node"); w.str(m.node); w.str("From"); w.str(m.from);
  return b;
}
inline Bytes encode_suspect(const Suspect& m) { return encode_suspect_like(kSuspect, m); }
inline Bytes encode_dead(const Dead& m) { return encode_suspect_like(kDead, m); }
inline Bytes encode(const Ping& m) {
  Bytes b{kPing}; Writer w{b};
  size_t n = 2 + !m.source_addr.empty() + (m.source_port != 0) + !m.source_node.empty();   // `codec:",omitempty"` on the Source* fields
  w.map(n); w.str("SeqNo"); w.uint(m.seq_no); w.str("Node"); w.str(m.node);
  if (!m.source_addr.empty()) { w.str("SourceAddr"); w.bytes(m.source_addr, false); }
  if (m.source_port) { w.str("SourcePort"); w.uint(m.source_port); }
  if (!m.source_node.empty()) { w.str("SourceNode"); w.str(m.source_node); }
  return b;
}
inline Bytes encode(const IndirectPing& m) {
  Bytes b{kIndirectPing}; Writer w{b};
  size_t n = 5 + !m.source_addr.empty() + (m.source_port != 0) + !m.source_node.empty();
  w.map(n); w.str("SeqNo"); w.uint(m.seq_no); w.str("Target"); w.bytes(m.target, true); w.str("Port"); w.uint(m.port);
  w.str("Node"); w.str(m.node); w.str("Nack"); w.boolean(m.nack);
  if (!m.source_addr.empty()) { w.str("SourceAddr"); w.bytes(m.source_addr, false); }
  if (m.source_port) { w.str("SourcePort"); w.uint(m.source_port); }
  if (!m.source_node.empty()) { w.str("SourceNode"); w.str(m.source_node); }
  return b;
}
inline Bytes encode(const AckResp& m) {
  Bytes b{kAckResp}; Writer w{b};
  w.map(2); w.str("SeqNo"); w.uint(m.seq_no); w.str("Payload"); w.bytes(m.payload, true);
  return b;
}
inline Bytes encode(const NackResp& m) { Bytes b{kNackResp}; Writer w{b}; w.map(1); w.str("SeqNo"); w.uint(m.seq_no); return b; }
inline Bytes encode(const SerfIntent& m) {
  Bytes b{kUser, uint8_t(m.join ? kSerfJoin : kSerfLeave)}; Writer w{b};
  if (m.join) { w.map(2); w.str("LTime"); w.uint(m.ltime); w.str("Node"); w.str(m.node); }
  else { w.map(3); w.str("LTime"); w.uint(m.ltime); w.str("Node"); w.str(m.node); w.str("Prune"); w.boolean(m.prune); }
  return b;
}
// a serf user event as memberlist carries it: userMsg, then serf's own type byte, then the msgpack struct
inline Bytes encode(const UserEvent& m) {
  Bytes b{kUser, kSerfUserEvent}; Writer w{b};
  w.map(4); w.str("LTime"); w.uint(m.ltime); w.str("Name"); w.str(m.name); w.str("Payload"); w.bytes(m.payload, true);
  w.str("CC"); w.boolean(m.cc);
  return b;
}

// decoders take the bytes AFTER the message-type byte(s); unknown keys are skipped like go-msgpack does
template <typename F> inline void decode_map(const uint8_t* p, size_t n, F&& field) {
  Reader r(p, n);
  size_t k = r.map();
  for (size_t i = 0; i < k; i++) { std::string key = r.str(); if (!field(key, r)) r.skip(); }
}
inline Alive decode_alive(const uint8_t* p, size_t n) {
  Alive m;
  decode_map(p, n, [&](const std::string& k, Reader& r) {
    if (k == "Incarnation") m.incarnation = uint32_t(r.uint()); else if (k == "Node") m.node = r.str();
    else if (k == "Addr") m.addr = r.raw(); else if (k == "Port") m.port = uint16_t(r.uint());
    else if (k == "Meta") m.meta = r.raw(); else if (k == "Vsn") m.vsn = r.raw(); else return false;
    return true;
  });
  return m;
}
inline Suspect decode_suspect(const uint8_t* p, size_t n) {
  Suspect m;
  decode_map(p, n, [&](const std::string& k, Reader& r) {
    if (k == "Incarnation") m.incarnation = uint32_t(r.uint()); else if (k == "Node") m.node = r.str();
    else if (k == "From") m.from = r.str(); else return false;
    return true;
  });
  return m;
}
inline Ping decode_ping(const uint8_t* p, size_t n) {
  Ping m;
  decode_map(p, n, [&](const std::string& k, Reader& r) {
    if (k == "SeqNo") m.seq_no = uint32_t(r.uint()); else if (k == "Node") m.node = r.str();
    else if (k == "SourceAddr") m.source_addr = r.raw(); else if (k == "SourcePort") m.source_port = uint16_t(r.uint());
    else if (k == "SourceNode") m.source_node = r.str(); else return false;
    return true;
  });
  return m;
}
inline IndirectPing decode_indirect_ping(const uint8_t* p, size_t n) {
  IndirectPing m;
  decode_map(p, n, [&](const std::string& k, Reader& r) {
    if (k == "SeqNo") m.seq_no = uint32_t(r.uint()); else if (k == "Target") m.target = r.raw(); else if (k == "Port") m.port = uint16_t(r.uint());
    else if (k == "Node") m.node = r.str(); else if (k == "Nack") m.nack = r.boolean();
    else if (k == "SourceAddr") m.source_addr = r.raw(); else if (k == "SourcePort") m.source_port = uint16_t(r.uint());
    else if (k == "SourceNode") m.source_node = r.str(); else return false;
    return true;
  });
  return m;
}
inline AckResp decode_ack(const uint8_t* p, size_t n) {
  AckResp m;
  decode_map(p, n, [&](const std::string& k, Reader& r) {
    if (k == "SeqNo") m.seq_no = uint32_t(r.uint()); else if (k == "Payload") m.payload = r.raw(); else return false;
    return true;
  });
  return m;
}
inline UserEvent decode_user_event(const uint8_t* p, size_t n) {
  UserEvent m;
  decode_map(p, n, [&](const std::string& k, Reader& r) {
    if (k == "LTime") m.ltime = r.uint(); else if (k == "Name") m.name = r.str(); else if (k == "Payload") m.payload = r.raw();
    else if (k == "CC") m.cc = r.boolean(); else return false;
    return true;
  });
  return m;
}

// ---------------------------------------------------------------------------------------------------------------
// compression (util.go compressPayload / decompressPayload): Go's compress/lzw, LSB order, 8-bit literals — GIF's
// variable-width LZW: codes of 9..12 bits packed least-significant-bit first, clear = 256, eof = 257, first free code
// 258; the encoder opens with a clear code and clears again when the table is full.  memberlist's DefaultLANConfig (and
// so Consul) has EnableCompression = true: a real node sends [compressMsg][msgpack {Algo: 0, Buf: lzw(payload)}] whenever
// that is shorter than the payload.  UPSTREAM-RECALL for the framing; the coder itself is pinned against an independent
// implementation of the same format (a GIF encoder / decoder) in tests/test_wire_lzw.py.
// ---------------------------------------------------------------------------------------------------------------
inline Bytes lzw_encode(const Bytes& in) {
  Bytes out; uint32_t bits = 0; unsigned nbits = 0, width = 9;
  auto emit = [&](uint32_t code) { bits |= code << nbits; nbits += width; while (nbits >= 8) { out.push_back(uint8_t(bits)); bits >>= 8; nbits -= 8; } };
  const uint32_t kClear = 256, kEof = 257, kMax = 4095;
  std::map<uint32_t, uint32_t> table;               // (prefix code << 8 | byte) -> code
  uint32_t hi = kEof, overflow = 512;
  auto inc_hi = [&]() -> bool {                      // writer.go incHi: false = out of codes, the table was reset
    hi++;
    if (hi == overflow) { width++; overflow <<= 1; }
    if (hi == kMax) { emit(kClear); width = 9; hi = kEof; overflow = 512; table.clear(); return false; }
    return true;
  };
  emit(kClear);
  if (!in.empty()) {
    uint32_t code = in[0];
    for (size_t i = 1; i < in.size(); i++) {
      const uint32_t key = code << 8 | in[i];
      auto it = table.find(key);
      if (it != table.end()) { code = it->second; continue; }
      emit(code); code = in[i];
      if (inc_hi()) table[key] = hi;
    }
    emit(code); inc_hi();
  }
  emit(kEof);
  if (nbits) out.push_back(uint8_t(bits));
  return out;
}
inline Bytes lzw_decode(const uint8_t* p, size_t n, size_t limit = size_t(1) << 24) {
  Bytes out; uint32_t bits = 0; unsigned nbits = 0, width = 9;
  const uint32_t kClear = 256, kEof = 257, kInvalid = 0xFFFF;
  std::vector<uint16_t> prefix(4096); std::vector<uint8_t> suffix(4096);
  uint32_t hi = kEof, overflow = 512, last = kInvalid;
  size_t i = 0;
  for (;;) {
    while (nbits < width) { if (i >= n) throw DecodeError("lzw: truncated"); bits |= uint32_t(p[i++]) << nbits; nbits += 8; }
    const uint32_t code = bits & ((1u << width) - 1); bits >>= width; nbits -= width;
    if (code < kClear) {
      out.push_back(uint8_t(code));
      if (last != kInvalid) { suffix[hi] = uint8_t(code); prefix[hi] = uint16_t(last); }
    } else if (code == kClear) { width = 9; hi = kEof; overflow = 512; last = kInvalid; continue; }
    else if (code == kEof) return out;
    else if (code <= hi) {
      Bytes rev; uint32_t c = code;
      if (code == hi && last != kInvalid) {          // the code being defined: the last expansion followed by its own head
        c = last; while (c >= kClear) c = prefix[c];
        rev.push_back(uint8_t(c)); c = last;
      } else if (code == hi) throw DecodeError("lzw: invalid code");
      while (c >= kClear) { rev.push_back(suffix[c]); c = prefix[c]; }
      rev.push_back(uint8_t(c));
      out.insert(out.end(), rev.rbegin(), rev.rend());
      if (last != kInvalid) { suffix[hi] = uint8_t(c); prefix[hi] = uint16_t(last); }
    } else throw DecodeError("lzw: invalid code");
    if (out.size() > limit) throw DecodeError("lzw: output too large");
    last = code; hi++;
    if (hi >= overflow) { if (width == 12) { last = kInvalid; hi--; } else { width++; overflow = 1u << width; } }
  }
}
struct Compress { uint32_t algo = 0; Bytes buf; };     // net.go `compress{Algo compressionType; Buf []byte}`, lzwAlgo = 0
inline Bytes encode(const Compress& m) { Bytes b{kCompress}; Writer w{b}; w.map(2); w.str("Algo"); w.uint(m.algo); w.str("Buf"); w.bytes(m.buf, true); return b; }
inline Compress decode_compress(const uint8_t* p, size_t n) {
  Compress m;
  decode_map(p, n, [&](const std::string& k, Reader& r) { if (k == "Algo") m.algo = uint32_t(r.uint()); else if (k == "Buf") m.buf = r.raw(); else return false; return true; });
  return m;
}
// compressPayload as rawSendMsgPacket uses it: the compressed form only when it is shorter
inline Bytes maybe_compress(const Bytes& msg) { Bytes c = encode(Compress{ 0, lzw_encode(msg) }); return c.size() < msg.size() ? c : msg; }
inline Bytes decompress(const uint8_t* body, size_t n) {
  Compress c = decode_compress(body, n);
  if (c.algo != 0) throw DecodeError("compressMsg: unknown algorithm");
  return lzw_decode(c.buf.data(), c.buf.size());
}

// ---------------------------------------------------------------------------------------------------------------
// framing: compound packets (util.go makeCompoundMessage / decodeCompoundMessage), label (label.go), CRC (net.go)
// ---------------------------------------------------------------------------------------------------------------
// [compoundMsg][n u8][n x len u16 BE][payloads]; more than 255 messages span several packets
inline std::vector<Bytes> make_compound(const std::vector<Bytes>& msgs) {
  std::vector<Bytes> out;
  for (size_t i = 0; i < msgs.size(); i += 255) {
    size_t n = std::min<size_t>(255, msgs.size() - i);
    Bytes b{kCompound, uint8_t(n)}; Writer w{b};
    for (size_t j = 0; j < n; j++) { if (msgs[i + j].size() > 0xFFFF) throw std::length_error("message too long for a compound"); w.be16(uint16_t(msgs[i + j].size())); }
    for (size_t j = 0; j < n; j++) b.insert(b.end(), msgs[i + j].begin(), msgs[i + j].end());
    out.push_back(std::move(b));
  }
  return out;
}
// bytes after the compoundMsg type byte -> the parts; `truncated` counts parts cut off by a short buffer (memberlist
// keeps what it got and reports the number it lost)
inline std::vector<Bytes> decode_compound(const uint8_t* p, size_t n, size_t* truncated = nullptr) {
  if (n < 1) throw DecodeError("missing compound length byte");
  size_t parts = p[0]; p++; n--;
  if (n < 2 * parts) throw DecodeError("truncated len slice");
  std::vector<uint16_t> lens(parts);
  for (size_t i = 0; i < parts; i++) lens[i] = uint16_t(p[2 * i] << 8 | p[2 * i + 1]);
  p += 2 * parts; n -= 2 * parts;
  std::vector<Bytes> out; size_t lost = 0;
  for (size_t i = 0; i < parts; i++) {
    if (n < lens[i]) { lost = parts - i; break; }
    out.emplace_back(p, p + lens[i]); p += lens[i]; n -= lens[i];
  }
  if (truncated) *truncated = lost;
  return out;
}
// memberlist fills a gossip packet with getBroadcasts(compoundOverhead = 2, limit): the same arithmetic for a caller
// that packs encoded messages itself.  Returns how many of `msgs` (in order) fit a packet of `udp_buffer_size`.
inline size_t fit_compound(const std::vector<Bytes>& msgs, size_t udp_buffer_size = 1400, size_t label_len = 0) {
  const size_t compound_header_overhead = 2, compound_overhead = 2;
  size_t label_overhead = label_len ? 2 + label_len : 0;
  if (udp_buffer_size < compound_header_overhead + label_overhead) return 0;
  size_t avail = udp_buffer_size - compound_header_overhead - label_overhead, used = 0, k = 0;
  for (; k < msgs.size() && k < 255; k++) { size_t need = compound_overhead + msgs[k].size(); if (used + need > avail) break; used += need; }
  return k;
}
// label.go: [hasLabelMsg][len u8][label] in front of the packet
inline Bytes add_label(const Bytes& packet, const std::string& label) {
  if (label.empty()) return packet;
  if (label.size() > 255) throw std::length_error("label too long");
  Bytes b{kHasLabel, uint8_t(label.size())}; b.insert(b.end(), label.begin(), label.end()); b.insert(b.end(), packet.begin(), packet.end());
  return b;
}
inline Bytes strip_label(const Bytes& packet, std::string* label) {
  if (packet.empty() || packet[0] != kHasLabel) { if (label) label->clear(); return packet; }
  if (packet.size() < 2 || packet.size() < 2u + packet[1] || packet[1] == 0) throw DecodeError("bad label header");
  if (label) label->assign(packet.begin() + 2, packet.begin() + 2 + packet[1]);
  return Bytes(packet.begin() + 2 + packet[1], packet.end());
}
// net.go: protocol version >= 5 prefixes [hasCrcMsg][crc32 IEEE u32 BE] over the rest of the packet
inline uint32_t crc32_ieee(const uint8_t* p, size_t n) {
  uint32_t c = 0xFFFFFFFFu;
  for (size_t i = 0; i < n; i++) { c ^= p[i]; for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u))); }
  return ~c;
}
inline Bytes add_crc(const Bytes& packet) {
  Bytes b{kHasCrc}; Writer w{b}; w.be32(crc32_ieee(packet.data(), packet.size())); b.insert(b.end(), packet.begin(), packet.end());
  return b;
}
inline Bytes strip_crc(const Bytes& packet) {
  if (packet.empty() || packet[0] != kHasCrc) return packet;
  if (packet.size() < 5) throw DecodeError("truncated crc header");
  uint32_t want = uint32_t(packet[1]) << 24 | uint32_t(packet[2]) << 16 | uint32_t(packet[3]) << 8 | packet[4];
  if (crc32_ieee(packet.data() + 5, packet.size() - 5) != want) throw DecodeError("crc mismatch");
  return Bytes(packet.begin() + 5, packet.end());
}

// ---------------------------------------------------------------------------------------------------------------
// the bridge: rumour records <-> packets
// ---------------------------------------------------------------------------------------------------------------
// How virtual node ids appear to a real memberlist node.  The defaults match the cgo shim of INTEGRATION.md
// (name "node-<id>", address 10.x.y.z, serf port 8301, protocol/delegate versions {1,5,2,2,5,4}).
// ---------------------------------------------------------------------------------------------------------------
// the stream side: push-pull (net.go sendLocalState / readRemoteState).  On a TCP connection the initiator writes
// [label header]? then ONE message: [pushPullMsg][msgpack pushPullHeader{Nodes, UserStateLen, Join}] followed by Nodes
// msgpack pushNodeState{Name, Addr, Port, Meta, Incarnation, State, Vsn} values back to back and UserStateLen bytes of the
// delegate's state (serf's messagePushPull) — the whole thing wrapped in compressMsg when compression is on (rawSendMsgStream)
// — and reads the same back.  UPSTREAM-RECALL like the packet structs above.
// ---------------------------------------------------------------------------------------------------------------
struct PushNodeState { std::string name; Bytes addr; uint16_t port = 0; Bytes meta; uint32_t incarnation = 0; uint32_t state = 0; Bytes vsn; };
struct PushPull { bool join = false; std::vector<PushNodeState> nodes; Bytes user_state; };
inline Bytes encode(const PushPull& m) {
  Bytes b{kPushPull}; Writer w{b};
  w.map(3); w.str("Nodes"); w.uint(m.nodes.size()); w.str("UserStateLen"); w.uint(m.user_state.size()); w.str("Join"); w.boolean(m.join);
  for (const PushNodeState& n : m.nodes) {
    w.map(7); w.str("Name"); w.str(n.name); w.str("Addr"); w.bytes(n.addr, true); w.str("Port"); w.uint(n.port); w.str("Meta"); w.bytes(n.meta, true);
    w.str("Incarnation"); w.uint(n.incarnation); w.str("State"); w.uint(n.state); w.str("Vsn"); w.bytes(n.vsn, true);
  }
  b.insert(b.end(), m.user_state.begin(), m.user_state.end());
  return b;
}
// the bytes AFTER the pushPullMsg byte; a node count or state length the buffer cannot hold is refused before anything is allocated
inline PushPull decode_push_pull(const uint8_t* p, size_t n) {
  Reader r(p, n); PushPull m; uint64_t nodes = 0, ulen = 0;
  { size_t k = r.map(); for (size_t i = 0; i < k; i++) { std::string key = r.str();
      if (key == "Nodes") nodes = r.uint(); else if (key == "UserStateLen") ulen = r.uint(); else if (key == "Join") m.join = r.boolean(); else r.skip(); } }
  if (nodes > r.left() || ulen > r.left()) throw DecodeError("push-pull header promises more than the stream holds");
  m.nodes.reserve(size_t(std::min<uint64_t>(nodes, r.left() / 8)));     // (a node record is at least a map header and a few keys: never trust the count for memory)
  for (uint64_t j = 0; j < nodes; j++) {
    PushNodeState s; size_t k = r.map();
    for (size_t i = 0; i < k; i++) { std::string key = r.str();
      if (key == "Name") s.name = r.str(); else if (key == "Addr") s.addr = r.raw(); else if (key == "Port") s.port = uint16_t(r.uint());
      else if (key == "Meta") s.meta = r.raw(); else if (key == "Incarnation") s.incarnation = uint32_t(r.uint()); else if (key == "State") s.state = uint32_t(r.uint());
      else if (key == "Vsn") s.vsn = r.raw(); else r.skip(); }
    m.nodes.push_back(std::move(s));
  }
  if (ulen > r.left()) throw DecodeError("truncated");
  m.user_state.assign(r.p, r.p + ulen);
  return m;
}
// The delegate's bytes behind the node list are serf's own push-pull message (serf/messages.go messagePushPull, sent by
// delegate.LocalState, merged by delegate.MergeRemoteState): [messagePushPullType = 2][msgpack {LTime, StatusLTimes{name: ltime},
// LeftMembers[], EventLTime, Events[{LTime, Events[{Name, Payload}]}], QueryLTime}].  UPSTREAM-RECALL.  What the simulator can fill
// in for a virtual member: its event clock (for LTime and EventLTime: one clock here, three in serf — DESIGN §8) and the members
// it holds as Left; no per-member status times, no recent-event buffer, no queries.
struct SerfUserEvents { uint64_t ltime = 0; std::vector<std::pair<std::string, Bytes>> events; };
struct SerfPushPull { uint64_t ltime = 0, event_ltime = 0, query_ltime = 0; std::vector<std::pair<std::string, uint64_t>> status_ltimes;
                      std::vector<std::string> left_members; std::vector<SerfUserEvents> events; };
inline Bytes encode(const SerfPushPull& m) {
  Bytes b{kSerfPushPull}; Writer w{b};
  w.map(6); w.str("LTime"); w.uint(m.ltime);
  w.str("StatusLTimes"); w.map(m.status_ltimes.size()); for (const auto& kv : m.status_ltimes) { w.str(kv.first); w.uint(kv.second); }
  w.str("LeftMembers"); w.array(m.left_members.size()); for (const std::string& n : m.left_members) w.str(n);
  w.str("EventLTime"); w.uint(m.event_ltime);
  w.str("Events"); w.array(m.events.size());
  for (const SerfUserEvents& ue : m.events) {
    w.map(2); w.str("LTime"); w.uint(ue.ltime); w.str("Events"); w.array(ue.events.size());
    for (const auto& ev : ue.events) { w.map(2); w.str("Name"); w.str(ev.first); w.str("Payload"); w.bytes(ev.second, true); }
  }
  w.str("QueryLTime"); w.uint(m.query_ltime);
  return b;
}
inline SerfPushPull decode_serf_push_pull(const uint8_t* p, size_t n) {       // the bytes AFTER the messagePushPullType byte
  Reader r(p, n); SerfPushPull m;
  size_t k = r.map();
  for (size_t i = 0; i < k; i++) {
    std::string key = r.str();
    if (key == "LTime") m.ltime = r.uint(); else if (key == "EventLTime") m.event_ltime = r.uint(); else if (key == "QueryLTime") m.query_ltime = r.uint();
    else if (key == "StatusLTimes") { size_t c = r.map(); if (c > r.left()) throw DecodeError("truncated"); for (size_t j = 0; j < c; j++) { std::string name = r.str(); m.status_ltimes.emplace_back(std::move(name), r.uint()); } }
    else if (key == "LeftMembers") { size_t c = r.array(); if (c > r.left()) throw DecodeError("truncated"); for (size_t j = 0; j < c; j++) m.left_members.push_back(r.str()); }
    else if (key == "Events") {
      size_t c = r.array(); if (c > r.left()) throw DecodeError("truncated");
      for (size_t j = 0; j < c; j++) {
        if (r.left() && *r.p == 0xc0) { r.u8(); continue; }                   // a nil *userEvents
        SerfUserEvents ue; size_t f = r.map();
        for (size_t q = 0; q < f; q++) {
          std::string kk = r.str();
          if (kk == "LTime") ue.ltime = r.uint();
          else if (kk == "Events") { size_t ce = r.array(); if (ce > r.left()) throw DecodeError("truncated");
            for (size_t x = 0; x < ce; x++) { std::string nm; Bytes pl; size_t g = r.map(); for (size_t y = 0; y < g; y++) { std::string k3 = r.str(); if (k3 == "Name") nm = r.str(); else if (k3 == "Payload") pl = r.raw(); else r.skip(); } ue.events.emplace_back(std::move(nm), std::move(pl)); } }
          else r.skip();
        }
        m.events.push_back(std::move(ue));
      }
    }
    else r.skip();
  }
  return m;
}
// what a peer writes on / reads from the connection: optional label header, optional compressMsg wrapper
inline Bytes to_stream(const PushPull& m, const std::string& label, bool compress) {
  Bytes b = encode(m);
  if (compress) b = maybe_compress(b);
  return add_label(b, label);
}
inline PushPull from_stream(const Bytes& stream, std::string* label = nullptr) {
  Bytes b = strip_label(stream, label);
  if (!b.empty() && b[0] == kEncrypt) throw DecodeError("encryptMsg: gossip encryption is not supported by the bridge");
  if (!b.empty() && b[0] == kCompress) b = decompress(b.data() + 1, b.size() - 1);
  if (b.empty() || b[0] != kPushPull) throw DecodeError("expected a push-pull message on the stream");
  return decode_push_pull(b.data() + 1, b.size() - 1);
}

struct Naming {
  std::string prefix = "node-";
  uint16_t port = 8301;
  Bytes vsn = { 1, 5, 2, 2, 5, 4 };
  std::string event_name = "swimsim";               // user events travel as {Name: event_name, Payload: id u32 BE}
  std::string name_of(uint32_t id) const { return prefix + std::to_string(id); }
  bool id_of(const std::string& name, uint32_t* id) const {
    if (name.compare(0, prefix.size(), prefix) != 0 || name.size() == prefix.size()) return false;
    uint64_t v = 0;
    for (size_t i = prefix.size(); i < name.size(); i++) { if (name[i] < '0' || name[i] > '9') return false; v = v * 10 + uint64_t(name[i] - '0'); if (v > 0xFFFFFFFFull) return false; }
    *id = uint32_t(v); return true;
  }
  Bytes addr_of(uint32_t id) const { return Bytes{ 10, uint8_t(id >> 16), uint8_t(id >> 8), uint8_t(id) }; }
};

// one rumour record (as swim_transport_poll returns it) -> one encoded memberlist message
inline Bytes to_wire(const swim_edge& e, const Naming& nm = Naming()) {
  const uint32_t type = e.meta >> 30, from = e.meta & 0x3FFFFFFFu;
  switch (type) {
    case SWIM_MSG_ALIVE: { Alive a; a.incarnation = e.incarnation; a.node = nm.name_of(e.subject); a.addr = nm.addr_of(e.subject); a.port = nm.port; a.vsn = nm.vsn; return encode(a); }
    case SWIM_MSG_SUSPECT: return encode_suspect(Suspect{ e.incarnation, nm.name_of(e.subject), nm.name_of(from) });
    case SWIM_MSG_DEAD: return encode_dead(Dead{ e.incarnation, nm.name_of(e.subject), nm.name_of(from) });
    default: {
      if (e.subject & SWIM_INTENT_LEAVE) {           // one of serf's intents: messageLeave / messageJoin, not a user event
        SerfIntent it; it.join = (e.subject & SWIM_INTENT_JOIN) == SWIM_INTENT_JOIN; it.ltime = e.incarnation; it.node = nm.name_of(e.subject & 0x1FFFFFFFu);
        it.prune = !it.join && (e.subject & SWIM_INTENT_PRUNE) != 0; return encode(it);
      }
      UserEvent u; u.ltime = e.incarnation; u.name = nm.event_name;
               u.payload = Bytes{ uint8_t(e.subject >> 24), uint8_t(e.subject >> 16), uint8_t(e.subject >> 8), uint8_t(e.subject) }; return encode(u); }
  }
}
// every rumour of a polled batch that went to the same receiver from the same sender in one packet: the gossip packet
// the real node would have received (compound when more than one message)
inline Bytes to_packet(const std::vector<swim_edge>& rumours, const Naming& nm = Naming()) {
  std::vector<Bytes> msgs; msgs.reserve(rumours.size());
  for (const swim_edge& e : rumours) msgs.push_back(to_wire(e, nm));
  if (msgs.size() == 1) return msgs[0];
  std::vector<Bytes> c = make_compound(msgs);
  if (c.size() != 1) throw std::length_error("more than 255 messages for one packet");
  return c[0];
}
// a packet written by the real node -> the rumour records to hand to swim_transport_write_to (edge.dst is left 0: the
// call names the receiver).  Pings, acks and the other control messages carry no rumour and are counted in `control`;
// names that are not "<prefix><id>" are counted in `foreign`.
// A ping / indirect ping the real node sent: the bridge answers it on behalf of the virtual peer (BridgeTransport).
struct Probe { bool indirect = false; uint32_t seq_no = 0; bool nack = false; std::string node; Bytes target; };
struct UnsupportedPacket : DecodeError { using DecodeError::DecodeError; };   // encryptMsg: not decodable here
inline std::vector<swim_edge> from_packet(const Bytes& packet, const Naming& nm = Naming(), size_t* control = nullptr, size_t* foreign = nullptr,
                                          std::vector<Probe>* probes = nullptr) {
  std::vector<swim_edge> out; size_t n_control = 0, n_foreign = 0, layers = 0;
  std::vector<Bytes> todo{ strip_crc(strip_label(packet, nullptr)) };
  while (!todo.empty()) {
    Bytes m = std::move(todo.back()); todo.pop_back();
    if (m.empty()) continue;
    const uint8_t* body = m.data() + 1; size_t n = m.size() - 1;
    switch (m[0]) {
      case kCompound: { std::vector<Bytes> parts = decode_compound(body, n); for (size_t i = parts.size(); i-- > 0;) todo.push_back(std::move(parts[i])); break; }
      case kAlive: {
        Alive a = decode_alive(body, n); uint32_t id;
        if (!nm.id_of(a.node, &id)) { n_foreign++; break; }
        out.push_back(swim_edge{ 0, id, a.incarnation, uint32_t(SWIM_MSG_ALIVE) << 30 }); break;
      }
      case kSuspect: case kDead: {
        Suspect s = decode_suspect(body, n); uint32_t id, from;
        if (!nm.id_of(s.node, &id) || !nm.id_of(s.from, &from)) { n_foreign++; break; }
        out.push_back(swim_edge{ 0, id, s.incarnation, uint32_t(m[0] == kSuspect ? SWIM_MSG_SUSPECT : SWIM_MSG_DEAD) << 30 | (from & 0x3FFFFFFFu) }); break;
      }
      case kUser: {
        if (n >= 1 && (body[0] == kSerfLeave || body[0] == kSerfJoin)) {       // serf's intents: {LTime, Node[, Prune]}
          SerfIntent it; it.join = body[0] == kSerfJoin;
          decode_map(body + 1, n - 1, [&](const std::string& k, Reader& r) {
            if (k == "LTime") it.ltime = r.uint(); else if (k == "Node") it.node = r.str(); else if (k == "Prune") it.prune = r.boolean(); else return false;
            return true;
          });
          uint32_t id;
          if (!nm.id_of(it.node, &id) || id > 0x0FFFFFFFu) { n_foreign++; break; }
          out.push_back(swim_edge{ 0, (it.join ? SWIM_INTENT_JOIN : SWIM_INTENT_LEAVE | (it.prune ? SWIM_INTENT_PRUNE : 0u)) | id, uint32_t(it.ltime), uint32_t(SWIM_MSG_USER) << 30 }); break;
        }
        if (n < 1 || body[0] != kSerfUserEvent) { n_control++; break; }       // serf queries etc.: not modelled
        UserEvent u = decode_user_event(body + 1, n - 1);
        uint32_t id = 0; for (size_t i = 0; i < u.payload.size() && i < 4; i++) id = id << 8 | u.payload[i];
        id &= SWIM_EVENT_ID_MAX;                                                   // (bits 31-30 of an id word mark serf's intents)
        out.push_back(swim_edge{ 0, id, uint32_t(u.ltime), uint32_t(SWIM_MSG_USER) << 30 }); break;
      }
      case kPing: { n_control++; if (probes) { Ping pg = decode_ping(body, n); Probe pr; pr.seq_no = pg.seq_no; pr.node = pg.node; probes->push_back(pr); } break; }
      case kIndirectPing: { n_control++; if (probes) { IndirectPing ip = decode_indirect_ping(body, n); Probe pr; pr.indirect = true; pr.seq_no = ip.seq_no; pr.nack = ip.nack; pr.node = ip.node; pr.target = ip.target; probes->push_back(pr); } break; }
      // memberlist's DefaultLANConfig has EnableCompression = true (Consul's default): the payload is a whole message again
      case kCompress: if (++layers > 1) throw DecodeError("compressMsg inside compressMsg"); todo.push_back(decompress(body, n)); break;   // (memberlist compresses once)
      // Consul may encrypt gossip: such a packet carries rumours this codec cannot see.  Losing them silently would be worse
      // than refusing the packet.
      case kEncrypt: throw UnsupportedPacket("encryptMsg: gossip encryption is not supported by the bridge");
      default: n_control++; break;                   // ack / nack / push-pull / err
    }
  }
  if (control) *control = n_control;
  if (foreign) *foreign = n_foreign;
  return out;
}

// ---------------------------------------------------------------------------------------------------------------
// memberlist.Transport over the bridge, at the byte level: what a NodeAwareTransport like wanfed.Transport
// (agent/consul/wanfed/wanfed.go:96-141) does with real sockets, done with the simulator's attached-node calls.
// ---------------------------------------------------------------------------------------------------------------
class BridgeTransport {
 public:
  struct Packet { Bytes buf; std::string from; uint32_t from_id; };   // memberlist.Packet{Buf, From}; from_id SWIM_NONE: not a gossip packet

  // `compress` = memberlist.Config.EnableCompression of the virtual peers (DefaultLANConfig: true): what they send is compressed
  // when that is shorter, before the CRC and the label like rawSendMsgPacket does; compressed packets are always accepted
  // `n_nodes` = swim_config.n_nodes of the handle (the size of a member list; only PushPull needs it)
  BridgeTransport(swim_sim* sim, uint32_t replica, uint32_t self_id, Naming naming = Naming(), std::string label = std::string(), bool crc = true, bool compress = false,
                  uint32_t n_nodes = 0)
      : sim_(sim), replica_(replica), self_(self_id), nm_(std::move(naming)), label_(std::move(label)), crc_(crc), compress_(compress), n_nodes_(n_nodes) {}

  // Transport.WriteToAddress(b, Address{Addr, Name}): `to` is the receiver's node name ("node-7") or its "10.a.b.c[:port]"
  // address.  Returns 0, a SWIM_E* code, or SWIM_EINVAL for an address outside the virtual cluster.
  int WriteTo(const Bytes& packet, const std::string& to) {
    uint32_t dst;
    if (!resolve(to, &dst)) return SWIM_EINVAL;
    size_t control = 0, foreign = 0; std::vector<Probe> probes; std::vector<swim_edge> recs;
    try { recs = from_packet(packet, nm_, &control, &foreign, &probes); }
    catch (const std::exception&) { unsupported_seen_++; return SWIM_EINVAL; }   // encrypted, or malformed (a hostile or truncated packet, also one that asks for absurd amounts of memory): refused whole
    control_seen_ += control; foreign_seen_ += foreign;
    // probeNode of the real node: the virtual peer answers like handlePing / handleIndirectPing would — an ackResp when it
    // (and, for an indirect ping, the target behind it) is running and in the real node's partition, a nackResp for a
    // failed indirect ping that asked for one, silence otherwise (the real node's probe then times out, as it should)
    for (const Probe& pr : probes) answer_probe(pr, dst);
    if (recs.empty()) return SWIM_OK;
    return swim_transport_write_to(sim_, replica_, self_, dst, recs.data(), recs.size());
  }
  // Transport.PacketCh(): everything virtual peers sent to this node since the last call, one packet per sender and
  // tick batch, framed as the real node expects it (compound, CRC, label)
  std::vector<Packet> Poll(size_t cap = 65536) {
    std::vector<swim_edge> got(cap); size_t n = 0;
    std::vector<Packet> out;
    for (auto& a : acks_) {                            // answers to the real node's probes first
      Bytes pkt = a.buf;
      if (crc_) pkt = add_crc(pkt);
      out.push_back(Packet{ add_label(pkt, label_), nm_.name_of(a.from_id), a.from_id });
    }
    acks_.clear();
    if (swim_transport_poll(sim_, replica_, self_, got.data(), got.size(), &n) != SWIM_OK) return out;
    for (size_t i = 0; i < n;) {
      size_t j = i; std::vector<swim_edge> batch;
      while (j < n && got[j].dst == got[i].dst && batch.size() < 255) { swim_edge e = got[j]; e.dst = 0; batch.push_back(e); j++; }
      Bytes pkt = to_packet(batch, nm_);
      if (compress_) pkt = maybe_compress(pkt);
      if (crc_) pkt = add_crc(pkt);
      pkt = add_label(pkt, label_);
      out.push_back(Packet{ std::move(pkt), got[i].dst == SWIM_NONE ? std::string() : nm_.name_of(got[i].dst), got[i].dst });
      i = j;
    }
    return out;
  }
  // Transport.DialAddressTimeout + the stream exchange of pushPullNode (what memberlist.Join and the periodic push-pull of the
  // real node do over TCP): `to` answers with its whole view — one pushNodeState per member it knows, the attached node among
  // them — after merging what the real node sent the way mergeState does: Alive -> alive, Left -> dead{From: the node itself},
  // Dead and Suspect -> suspect{From: the receiver} (a remote Dead is never trusted directly).  Names outside the virtual cluster
  // are counted as foreign.  The reply is framed like the request (label, compression per the constructor).  Throws DecodeError
  // for a stream it cannot read; returns an empty vector when `to` is not a running, reachable virtual member (the dial fails).
  Bytes PushPull(const Bytes& stream, const std::string& to) {
    uint32_t dst;
    if (!resolve(to, &dst) || n_nodes_ == 0) return {};
    wire::PushPull in;
    try { in = from_stream(stream, nullptr); }
    catch (const DecodeError&) { throw; }
    catch (const std::exception& e) { throw DecodeError(std::string("push-pull stream refused: ") + e.what()); }   // (bad_alloc / length_error of a hostile stream: one error type at the boundary)
    if (dst == self_ || !up_and_reachable(dst)) return {};
    std::vector<swim_edge> recs;
    for (const PushNodeState& n : in.nodes) {
      uint32_t id;
      if (!nm_.id_of(n.name, &id)) { foreign_seen_++; continue; }
      switch (n.state) {
        case SWIM_STATE_ALIVE: recs.push_back(swim_edge{ 0, id, n.incarnation, uint32_t(SWIM_MSG_ALIVE) << 30 }); break;
        case SWIM_STATE_LEFT: recs.push_back(swim_edge{ 0, id, n.incarnation, uint32_t(SWIM_MSG_DEAD) << 30 | (id & 0x3FFFFFFFu) }); break;
        default: recs.push_back(swim_edge{ 0, id, n.incarnation, uint32_t(SWIM_MSG_SUSPECT) << 30 | (dst & 0x3FFFFFFFu) }); break;
      }
    }
    if (!recs.empty() && swim_transport_write_to(sim_, replica_, self_, dst, recs.data(), recs.size()) != SWIM_OK) return {};
    // the peer's local state as of now (what it has merged in this tick's arrivals shows from the next tick on, as for any packet)
    std::vector<swim_member> members(n_nodes_); size_t got = 0;
    if (swim_members(sim_, replica_, dst, members.data(), members.size(), &got) != SWIM_OK) return {};
    wire::PushPull out; out.join = false;
    for (size_t i = 0; i < got; i++) {
      const swim_member& m = members[i];
      if (m.status == SWIM_MEMBER_NONE) continue;                  // never heard of / erased by the reaper: not in its member list
      PushNodeState s; s.name = nm_.name_of(m.id); s.addr = nm_.addr_of(m.id); s.port = nm_.port; s.incarnation = m.incarnation; s.state = m.state; s.vsn = nm_.vsn;
      out.nodes.push_back(std::move(s));
    }
    // serf's delegate state behind the list: the peer's event clock and the members it holds as Left (what the simulator has)
    {
      SerfPushPull sp; swim_node_info ni;
      if (swim_node_info_get(sim_, replica_, dst, &ni) == SWIM_OK) sp.ltime = sp.event_ltime = ni.event_clock;
      for (size_t i = 0; i < got; i++) if (members[i].status == SWIM_MEMBER_LEFT) sp.left_members.push_back(nm_.name_of(members[i].id));
      out.user_state = encode(sp);
    }
    push_pulls_++;
    return to_stream(out, label_, compress_);
  }
  size_t push_pulls_answered() const { return push_pulls_; }
  size_t control_messages_seen() const { return control_seen_; }     // accumulated over all WriteTo calls
  size_t foreign_names_seen() const { return foreign_seen_; }
  size_t unsupported_packets_seen() const { return unsupported_seen_; }
  size_t probes_answered() const { return probes_answered_; }

 private:
  bool up_and_reachable(uint32_t id) const {
    swim_node_info me, ni;
    if (swim_node_info_get(sim_, replica_, self_, &me) != SWIM_OK || swim_node_info_get(sim_, replica_, id, &ni) != SWIM_OK) return false;
    return ni.alive && ni.partition == me.partition;
  }
  void answer_probe(const Probe& pr, uint32_t dst) {
    uint32_t named;
    if (!pr.indirect) {
      if (!pr.node.empty() && (!nm_.id_of(pr.node, &named) || named != dst)) return;      // handlePing: "got ping for unexpected node"
      if (!up_and_reachable(dst)) return;
      acks_.push_back(Packet{ encode(AckResp{ pr.seq_no, Bytes() }), std::string(), dst }); probes_answered_++;
      return;
    }
    if (!up_and_reachable(dst)) return;                 // the relay itself is down: nothing comes back
    uint32_t target = SWIM_NONE;
    if (!(nm_.id_of(pr.node, &target)) && pr.target.size() == 4 && pr.target[0] == 10) target = uint32_t(pr.target[1]) << 16 | uint32_t(pr.target[2]) << 8 | pr.target[3];
    const bool ok = target != SWIM_NONE && up_and_reachable(target);
    if (ok) { acks_.push_back(Packet{ encode(AckResp{ pr.seq_no, Bytes() }), std::string(), dst }); probes_answered_++; }
    else if (pr.nack) { acks_.push_back(Packet{ encode(NackResp{ pr.seq_no }), std::string(), dst }); probes_answered_++; }
  }
  bool resolve(const std::string& to, uint32_t* id) const {
    if (nm_.id_of(to, id)) return true;
    unsigned a, b, c, d;
    if (sscanf(to.c_str(), "%u.%u.%u.%u", &a, &b, &c, &d) == 4 && a == 10 && b < 256 && c < 256 && d < 256) { *id = b << 16 | c << 8 | d; return true; }
    return false;
  }
  swim_sim* sim_; uint32_t replica_, self_; Naming nm_; std::string label_; bool crc_, compress_; uint32_t n_nodes_; size_t push_pulls_ = 0;
  size_t control_seen_ = 0, foreign_seen_ = 0, unsupported_seen_ = 0, probes_answered_ = 0;
  std::vector<Packet> acks_;                          // ackResp / nackResp waiting for the next Poll
};

}  // namespace wire
}  // namespace swimsim

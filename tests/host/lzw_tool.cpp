// lzw_tool enc|dec < in > out — the wire codec's LZW coder on a byte stream (tests/test_wire_lzw.py checks it against a GIF
// encoder / decoder, an independent implementation of the format Go's compress/lzw LSB with 8-bit literals writes)
#include <cstdio>
#include <cstring>
#include <iterator>
#include <vector>

#include "../../include/swimsim_wire.hpp"

int main(int argc, char** argv) {
  using namespace swimsim::wire;
  if (argc != 2) return 2;
  Bytes in; int c;
  while ((c = std::getchar()) != EOF) in.push_back(uint8_t(c));
  Bytes out;
  try { out = !std::strcmp(argv[1], "enc") ? lzw_encode(in) : lzw_decode(in.data(), in.size()); }
  catch (const DecodeError& e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }
  std::fwrite(out.data(), 1, out.size(), stdout);
  return 0;
}

"""The wire codec's LZW coder (include/swimsim_wire.hpp: compressMsg, memberlist's EnableCompression) against an INDEPENDENT
implementation of the same format.  Go's compress/lzw with LSB order and 8-bit literals is GIF's LZW; Pillow's GIF writer and
reader are that format's encoder and decoder.  So: what Pillow compresses our decoder must open, and what our encoder writes
Pillow must open — neither side was written here."""
import io
import os
import subprocess

import numpy as np
import pytest

PIL = pytest.importorskip("PIL.Image")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tool(tmp_path_factory):
    exe = tmp_path_factory.mktemp("lzw") / "lzw_tool"
    subprocess.run(["g++", "-std=c++17", "-O1", "-o", str(exe), os.path.join(ROOT, "tests", "host", "lzw_tool.cpp")], check=True)
    return lambda mode, data: subprocess.run([str(exe), mode], input=data, capture_output=True, check=True).stdout


def samples():
    rng = np.random.default_rng(7)
    text = (b"alive node-17 suspect node-3 dead node-9 " * 300)
    yield "repetitive", text[:8192]
    yield "noise", rng.integers(0, 256, 16384, dtype=np.uint8).tobytes()          # more than 4 096 codes: table resets
    yield "runs", bytes(np.repeat(rng.integers(0, 256, 200, dtype=np.uint8), rng.integers(1, 90, 200)))[:10000]
    yield "ramp", bytes(range(256)) * 24


def gif_lzw_stream(gif: bytes):
    """the LZW bytes of the first image of a GIF (sub-blocks joined) and its minimum code size"""
    assert gif[:6] in (b"GIF87a", b"GIF89a")
    flags = gif[10]; p = 13
    if flags & 0x80:
        p += 3 * (2 << (flags & 7))
    while gif[p] == 0x21:                                # extensions
        p += 2
        while gif[p]:
            p += 1 + gif[p]
        p += 1
    assert gif[p] == 0x2C
    lflags = gif[p + 9]; p += 10
    assert not lflags & 0x40, "interlaced: the pixel stream is not in row order"
    if lflags & 0x80:
        p += 3 * (2 << (lflags & 7))
    min_code = gif[p]; p += 1
    out = bytearray()
    while gif[p]:
        out += gif[p + 1: p + 1 + gif[p]]; p += 1 + gif[p]
    return bytes(out), min_code


def as_image(data: bytes):
    w = 128; assert len(data) % w == 0
    im = PIL.frombytes("P", (w, len(data) // w), data)
    im.putpalette([v for i in range(256) for v in (i, (i * 7) % 256, (i * 13) % 256)])    # 256 distinct colours: 8-bit literals
    return im


@pytest.mark.parametrize("name,data", list(samples()))
def test_our_decoder_opens_what_a_gif_encoder_wrote(tool, name, data):
    data = data[: len(data) // 128 * 128]
    buf = io.BytesIO(); as_image(data).save(buf, format="GIF", optimize=False, interlace=False)   # (rows in order)
    stream, min_code = gif_lzw_stream(buf.getvalue())
    assert min_code == 8
    assert tool("dec", stream) == data


@pytest.mark.parametrize("name,data", list(samples()))
def test_a_gif_decoder_opens_what_our_encoder_wrote(tool, name, data):
    data = data[: len(data) // 128 * 128]
    z = tool("enc", data)
    assert tool("dec", z) == data
    # the smallest GIF around it: header, 256-entry palette, one image, our stream in sub-blocks
    h = len(data) // 128
    pal = bytes(v for i in range(256) for v in (i, (i * 7) % 256, (i * 13) % 256))
    gif = b"GIF89a" + (128).to_bytes(2, "little") + h.to_bytes(2, "little") + bytes([0xF7, 0, 0]) + pal
    gif += b"\x2C" + bytes(4) + (128).to_bytes(2, "little") + h.to_bytes(2, "little") + b"\x00" + b"\x08"
    for i in range(0, len(z), 255):
        gif += bytes([len(z[i:i + 255])]) + z[i:i + 255]
    gif += b"\x00\x3B"
    im = PIL.open(io.BytesIO(gif)); im.load()
    assert im.mode == "P" and im.tobytes() == data

#!/usr/bin/env python3
"""Feeds build/asan/host_check (AddressSanitizer + UBSan build of the mesh loaders, the SAH builder, the collapse and the scene
generators) well-formed, truncated and garbage OBJ / PLY files, build/asan/image_check (the host's PNG and JPEG loaders) whole,
truncated and damaged images, and the oracle's sanitizer build a small render. CPU only.
Exit status 0 = no sanitizer report."""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EXE = os.path.join(ROOT, "build", "asan", "host_check")
ORACLE = os.path.join(ROOT, "build", "asan", "oracle_check")
IMAGES = os.path.join(ROOT, "build", "asan", "image_check")


def main():
    rng = np.random.default_rng(1)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    with tempfile.TemporaryDirectory() as d:
        files = []

        def put(name, data):
            p = os.path.join(d, name)
            with open(p, "wb") as f:
                f.write(data)
            files.append(p)

        obj = b"v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nusemtl a\nf 1/1 2/2 3/3 4/4\nf -4 -3 -2\nusemtl b\nf 1//1 2//1 3//1\n"
        put("good.obj", obj)
        for k in range(0, len(obj), 7):
            put("trunc%d.obj" % k, obj[:k])
        put("badidx.obj", b"v 0 0 0\nf 1 2 3\n")
        put("hugeidx.obj", b"v 0 0 0\nv 1 1 1\nv 2 2 2\nf 99999999999999999999 2 3\nf 1/9 2 3\nf 1 2\n")
        head = b"ply\nformat binary_little_endian 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\nproperty float s\nproperty float t\n" \
               b"element face 2\nproperty list uchar int vertex_indices\nend_header\n"
        body = b"".join(struct.pack("<5f", *v) for v in ((0, 0, 0, 0, 0), (1, 0, 0, 1, 0), (1, 1, 0, 1, 1), (0, 1, 0, 0, 1))) + \
            struct.pack("<B4i", 4, 0, 1, 2, 3) + struct.pack("<B3i", 3, 0, 2, 3)
        put("good.ply", head + body)
        for k in range(0, len(body), 5):
            put("trunc%d.ply" % k, head + body[:k])
        put("neg.ply", head + body[:-12] + struct.pack("<3i", 0, -2, 3))
        put("count.ply", head.replace(b"vertex 4", b"vertex 4000000000") + body)
        put("noprops.ply", b"ply\nformat ascii 1.0\nelement vertex 1000000000000\nend_header\n")
        put("ascii.ply", b"ply\nformat ascii 1.0\nelement vertex 3\nproperty double x\nproperty double y\nproperty double z\nelement face 1\n"
                         b"property list uchar uint vertex_index\nend_header\n0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n")
        put("biglist.ply", head.replace(b"face 2", b"face 1") + body[:80] + struct.pack("<B", 255) + rng.bytes(40))
        for k in range(40):
            put("g%d.ply" % k, head + rng.bytes(int(rng.integers(0, 300))))
            put("g%d.obj" % k, obj[:int(rng.integers(0, len(obj)))] + rng.bytes(int(rng.integers(0, 200))))
        put("empty.obj", b"")
        put("empty.ply", b"")
        put("unknown.xyz", b"abc")
        res = subprocess.run([EXE] + files, env=env, capture_output=True, text=True)
        print(res.stdout.strip())
        if res.returncode != 0 or "ERROR: AddressSanitizer" in res.stderr or "runtime error" in res.stderr:
            print(res.stderr[-4000:])
            return 1
    if os.path.exists(IMAGES):
        with tempfile.TemporaryDirectory() as d:
            files = []

            def put(name, data):
                p = os.path.join(d, name)
                with open(p, "wb") as f:
                    f.write(data)
                files.append(p)

            cases = np.load(os.path.join(ROOT, "tests", "golden", "jpeg_stb.npz"))
            for key in cases.files:
                if not key.endswith("__jpeg"):
                    continue
                jpg = cases[key].tobytes()
                put(key + ".jpg", jpg)
                for k in range(0, len(jpg), max(1, len(jpg) // 23)):
                    put("%s.t%d.jpg" % (key, k), jpg[:k])
                for k in range(12):                       # a few bytes overwritten anywhere: headers, tables, entropy-coded data
                    dmg = bytearray(jpg)
                    for pos in rng.integers(2, len(jpg), size=int(rng.integers(1, 6))):
                        dmg[int(pos)] = int(rng.integers(0, 256))
                    put("%s.d%d.jpg" % (key, k), bytes(dmg))
            pngs = np.load(os.path.join(ROOT, "tests", "golden", "png_stb.npz"))
            for key in pngs.files:
                if not key.endswith("__png"):
                    continue
                png_bytes = pngs[key].tobytes()
                put(key + ".png", png_bytes)
                for k in range(8):                        # damaged headers, palettes and chunk lengths (CRCs are not checked)
                    dmg = bytearray(png_bytes)
                    for pos in rng.integers(8, min(len(dmg), 120), size=int(rng.integers(1, 4))):
                        dmg[int(pos)] = int(rng.integers(0, 256))
                    put("%s.d%d.png" % (key, k), bytes(dmg))
            # Huffman tables that announce more codes of a length than exist: the file's first table (DC luminance) replaced by
            # a well-formed segment whose counts over-subscribe the code space
            jpg = cases["s420_45x31__jpeg"].tobytes()
            at = jpg.find(b"\xff\xc4")
            seg_len = struct.unpack(">H", jpg[at + 2:at + 4])[0]
            for length, count in ((1, 200), (2, 9), (9, 255), (16, 255)):
                counts = bytearray(16)
                counts[length - 1] = count
                body = bytes([0x00]) + bytes(counts) + bytes(range(count))
                put("dht_%d_%d.jpg" % (length, count), jpg[:at] + b"\xff\xc4" + struct.pack(">H", 2 + len(body)) + body + jpg[at + 2 + seg_len:])
            put("huge.jpg", bytes.fromhex("ffd8ffc0000b08ffffffff01011100ffda0008010100003f00") + rng.bytes(64))
            put("soi.jpg", b"\xff\xd8\xff")
            import zlib

            def png(w, h, color, raw):
                def chunk(t, b):
                    return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b))
                return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")
            rows = b"".join(bytes([y % 5]) + rng.bytes(7 * 3) for y in range(9))
            good = png(7, 9, 2, rows)
            put("good.png", good)
            for k in range(0, len(good), 11):
                put("t%d.png" % k, good[:k])
            put("short.png", png(7, 9, 2, rows[:50]))
            put("wide.png", png(70000, 70000, 6, rows))
            put("filter.png", png(7, 9, 2, b"".join(bytes([9]) + rng.bytes(21) for _ in range(9))))
            res = subprocess.run([IMAGES] + files, env=env, capture_output=True, text=True)
            print(res.stdout.strip())
            if res.returncode != 0 or "ERROR: AddressSanitizer" in res.stderr or "runtime error" in res.stderr:
                print(res.stderr[-4000:])
                return 1
    if os.path.exists(ORACLE):
        res = subprocess.run([ORACLE], env=env, capture_output=True, text=True)
        print(res.stdout.strip())
        if res.returncode != 0 or "ERROR: AddressSanitizer" in res.stderr or "runtime error" in res.stderr:
            print(res.stderr[-4000:])
            return 1
    print("sanitizer run clean")
    return 0


if __name__ == "__main__":
    sys.exit(main())

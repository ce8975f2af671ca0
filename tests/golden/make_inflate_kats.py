#!/usr/bin/env python3
"""Re-express the reference's raw-inflate known-answer streams as a JSON fixture.

Build container only; reads /root/reference as TEXT (nothing is compiled or run).

Sources (zlib-ng 2.2.2):
  test/infcover.c:584-620  cover_inflate(): try(hex, id, err) -- raw streams; err=1 means
                           inflate() must fail with Z_DATA_ERROR and strm->msg == id; err=0 means
                           the stream must decode without a data error.
  test/infcover.c:646-663  cover_fast(): inf(hex, what, step, win, len, err) -- raw streams fed in
                           one piece (step 0) whose expected return is Z_DATA_ERROR or Z_STREAM_END.
                           Rows whose expectation depends on the caller's output chunking
                           (err == Z_OK, or step != 0) are recorded with "chunking_dependent": true.
  test/test_inflate_adler32.cc:18-25,48  a zlib-wrapped stream, its plaintext and its Adler-32.
Output: tests/golden/inflate_kat.json
"""
import json
import os
import re
import sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def hex_to_bytes(hexstr):
    return bytes(int(tok, 16) for tok in hexstr.split())


def c_strings(arglist):
    """concatenate adjacent C string literals in an argument list; returns (strings, rest)"""
    parts = re.findall(r'"((?:[^"\\]|\\.)*)"|([^",\s][^,]*)|(,)', arglist)
    args, cur = [], None
    for s, other, comma in parts:
        if comma:
            args.append(cur)
            cur = None
        elif other:
            cur = other.strip()
        else:
            cur = (cur or "") + s
    args.append(cur)
    return args


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; fixture already committed")
    src = open(os.path.join(REF, "test/infcover.c"), encoding="latin-1").read()
    rows = []
    body = src[src.index("static void cover_inflate(void)"):src.index("static void cover_trees(void)")]
    # drop the #ifdef ... #else branch that only applies to a non-default build flag
    body = re.sub(r"#ifdef INFLATE_ALLOW_INVALID_DISTANCE_TOOFAR_ARRR.*?#else(.*?)#endif", r"\1", body, flags=re.S)
    for m in re.finditer(r"\btry\((.*?)\);", body, flags=re.S):
        hexs, ident, err = c_strings(m.group(1))
        err = int(err)
        if err < 0:
            continue                      # gzip-wrapped trailer checks: outside the raw path
        rows.append({"kind": "try", "hex": hexs, "id": ident, "expect_data_error": bool(err),
                     "source": "test/infcover.c cover_inflate"})
    for name in ("cover_inflate", "cover_fast"):
        start = src.index("static void %s(void)" % name)
        end = src.index("\n}\n", start)
        for m in re.finditer(r"\binf\((.*?)\);", src[start:end], flags=re.S):
            hexs, what, step, win, length, err = c_strings(m.group(1))
            win = int(win)
            if win > 0:
                continue
            rows.append({"kind": "inf", "hex": hexs, "id": what, "step": int(step), "win": win,
                         "len": int(length), "expect": err,
                         "chunking_dependent": err == "Z_OK" or int(step) != 0,
                         "source": "test/infcover.c " + name})
    t = open(os.path.join(REF, "test/test_inflate_adler32.cc"), encoding="latin-1").read()
    comp = bytes(int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{2})", t[t.index("compressed[]"):t.index("};")]))
    original = re.search(r'original = "(.*?)";', t).group(1)
    adler = int(re.search(r"strm\.adler, (0x[0-9a-fA-F]+)", t).group(1), 16)
    doc = {"source": "zlib-ng 2.2.2 test/infcover.c, test/test_inflate_adler32.cc", "rows": rows,
           "zlib_stream": {"hex": comp.hex(), "plaintext": original, "adler32": adler}}
    with open(os.path.join(HERE, "inflate_kat.json"), "w") as f:
        json.dump(doc, f, indent=0)
        f.write("\n")
    print(len(rows), "rows")


if __name__ == "__main__":
    main()

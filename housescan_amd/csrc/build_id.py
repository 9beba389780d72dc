#!/usr/bin/env python3
"""Identity of the library's sources: sha256 over the files a build compiles (names sorted), first 16 hex digits.
The Makefile compiles it into libhskinfu.so (hsk_build_id()); tests/conftest.py recomputes it from the tree and refuses
to test a library built from other sources.  usage: build_id.py [--header out.h [suffix]]"""
import hashlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
INCLUDE = os.path.join(os.path.dirname(os.path.dirname(HERE)), "include")


def source_files():
    files = [os.path.join(HERE, f) for f in sorted(os.listdir(HERE)) if f.endswith((".hip", ".cpp", ".h")) and f != "build_id.h"]
    files += [os.path.join(INCLUDE, f) for f in sorted(os.listdir(INCLUDE)) if f.endswith(".h")]
    return files


def build_id():
    h = hashlib.sha256()
    for f in source_files():
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    bid = build_id()
    if len(sys.argv) > 2 and sys.argv[1] == "--header":
        text = '#define HSK_BUILD_ID "%s%s"\n' % (bid, sys.argv[3] if len(sys.argv) > 3 else "")
        old = open(sys.argv[2]).read() if os.path.exists(sys.argv[2]) else None
        if old != text:
            open(sys.argv[2], "w").write(text)
    else:
        print(bid)

#!/usr/bin/env python3
"""The digest the Makefile compiles into cntt_version() (sha256 of the sorted csrc/ sources + Makefile + cntt.h,
first 16 hex digits).  `python tools/csrc_hash.py` prints it; tools/profile.sh stamps profiles/pmc_traffic.json with it."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "concrete-ntt_amd", "csrc")


def csrc_hash():
    names = sorted(n for pat in ("*.hip", "*.hpp", "*.inc", "*.py") for n in glob.glob(os.path.join(CSRC, pat))
                   if os.path.basename(n) != "build_hash.inc")
    # GNU make's $(sort) orders by byte value, like Python's sorted() on ASCII names
    h = hashlib.sha256()
    for n in names + [os.path.join(CSRC, "Makefile"), os.path.join(ROOT, "include", "cntt.h")]:
        with open(n, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_hash())

"""Tool-side helper for A/B runs: applies `CNTT_SWITCHES="fp=0,blk=0,..."` (an environment variable read HERE, by the tool -- the library
itself never reads the environment) through the library's testing-only switchboard, cntt_debug_set() (include/cntt.h).
    CNTT_SWITCHES=fp=0 python tools/fpw_bench.py"""
import os


def apply(verbose=True):
    import concrete_ntt_amd as cntt
    spec = os.environ.get("CNTT_SWITCHES", "").strip()
    done = {}
    for item in filter(None, (x.strip() for x in spec.split(","))):
        key, _, val = item.partition("=")
        cntt.debug_set(key, int(val))
        done[key] = int(val)
    if verbose and done:
        import sys
        print("switches: %s" % done, file=sys.stderr)
    return done

"""The two host-only additions of round 5 to include/cntt.h:
  * cntt_debug_set / cntt_debug_get -- the ONE documented, testing-only switchboard for kernel-selection overrides (the library no
    longer reads any environment variable: checked on the sources and on the built library's strings);
  * cntt_shard_bounds -- the batch partition of SURVEY 8(e) for callers that drive several devices from one process
    (examples/multi_device.cpp), equal to concrete-ntt_amd/shard.py for every (batch, world, rank)."""
import glob
import os
import subprocess

import pytest

import concrete_ntt_amd as cntt
from concrete_ntt_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SWITCHES = {"fp": 1, "pm64": 1, "blk": 1, "mul32_blk": 1, "ext32_blk": 1, "ext_one": 1, "ext_split": -1, "native_acc": 1,
            "product_fused": -1, "plan52_via32": 1}


def test_switchboard_defaults_set_get_reset():
    cntt.debug_set("reset", 0)
    for k, d in SWITCHES.items():
        assert cntt.debug_get(k) == d, k
    for k, d in SWITCHES.items():
        cntt.debug_set(k, 0)
        assert cntt.debug_get(k) == 0
        cntt.debug_set(k, -1)                       # -1: back to this switch's default
        assert cntt.debug_get(k) == d
    with cntt.debug_switches(fp=0, native_acc=0):
        assert cntt.debug_get("fp") == 0 and cntt.debug_get("native_acc") == 0
    assert cntt.debug_get("fp") == 1 and cntt.debug_get("native_acc") == 1
    with pytest.raises(cntt.Panic):
        cntt.debug_set("no_such_switch", 1)
    with pytest.raises(cntt.Panic):
        cntt.debug_set("fp", 7)
    with pytest.raises(cntt.Panic):
        cntt.debug_get("no_such_switch")
    cntt.debug_set("blk", 0)
    cntt.debug_set("reset", 0)
    assert cntt.debug_get("blk") == 1


def test_the_library_does_not_read_the_environment():
    """No getenv in the product sources, and none imported by the built library (ten ambient switches)."""
    for path in glob.glob(os.path.join(ROOT, "concrete-ntt_amd", "csrc", "*")):
        if path.endswith((".hip", ".hpp", ".inc")):
            assert "getenv" not in open(path).read(), path
    for path in glob.glob(os.path.join(ROOT, "concrete-ntt_amd", "*.py")):
        src = open(path).read()
        assert "environ" not in src and "getenv" not in src, path
    nm = subprocess.run(["nm", "-D", "--undefined-only", cntt._lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in nm
    # every switch is documented in the header under the testing-only heading
    hdr = open(os.path.join(ROOT, "include", "cntt.h")).read()
    assert "TESTING ONLY" in hdr
    for k in SWITCHES:
        assert '"%s"' % k in hdr, k


def test_shard_bounds_c_abi_equals_shard_py():
    for batch in (0, 1, 2, 7, 63, 64, 65, 1000, 65536, (1 << 20) + 5, (1 << 40) + 3):
        for world in (1, 2, 3, 4, 5, 7, 8, 16):
            spans = [cntt.shard_bounds(batch, world, r) for r in range(world)]
            assert spans == [shard.shard_bounds(batch, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            for (b0, e0), (b1, e1) in zip(spans, spans[1:]):
                assert e0 == b1
    for world, rank in ((0, 0), (2, 2), (2, -1), (-1, 0)):
        with pytest.raises(cntt.Panic):
            cntt.shard_bounds(8, world, rank)


@pytest.mark.gpu
@pytest.mark.parametrize("args", [["--n", "4096", "--batch", "1000"],                                   # every visible device
                                  ["--n", "1024", "--batch", "4099", "--world", "3", "--logical"],     # three threads / streams, ragged
                                  ["--n", "16384", "--batch", "5", "--world", "8", "--logical"]])      # more shards than polynomials
def test_multi_device_example(args):
    """examples/multi_device.cpp: one process, one host thread + stream per shard, hipMemcpyPeerAsync scatter / gather around
    cntt_prime64_fwd_batch, the plan handle shared by all threads.  On the one-GPU pool --logical spreads the shards over the device
    that exists (world > 1 still runs partition, scatter, concurrent calls on one plan from several threads, gather)."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "examples"), "multi_device"], check=True)
    r = subprocess.run([os.path.join(ROOT, "examples", "multi_device")] + args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.returncode, r.stdout[-2000:], r.stderr[-2000:])

"""CPU-only checks of the product's host side: plan construction (its own number theory, separate
code from the oracle) against the golden vectors and the oracle, the C-ABI surface, error mapping.
No compute entry point is called here (they need a GPU and have no fallback)."""
import ctypes
import hashlib
import os
import re
import subprocess

import numpy as np
import pytest

import concrete_ntt_amd as cntt
from concrete_ntt_amd import _lib, prime32, prime64

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _mod(bits):
    return prime64 if bits == 64 else prime32


def test_library_exports_every_declared_symbol():
    hdr = os.path.join(ROOT, "include", "cntt.h")
    pre = subprocess.run(["gcc", "-E", "-P", hdr], check=True, capture_output=True, text=True).stdout
    names = sorted(set(re.findall(r"\b(cntt_[a-z0-9_]+)\s*\(", pre)))
    assert len(names) >= 59
    L = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_plans_match_golden(golden):
    for ent in golden["plans"]:
        pl = _mod(ent["bits"]).Plan.try_new(ent["n"], ent["p"])
        assert pl is not None
        info = pl.info()
        assert (info.ntt_size, info.modulus, info.root) == (ent["n"], ent["p"], ent["w"])
        assert (info.n_inv_mod_p, info.big_q, bool(info.has_shoup)) == (ent["n_inv"], ent["big_q"], ent["has_shoup"])
        assert pl.table(_lib.TWID)[:8].tolist() == ent["twid_head"]
        assert sha(pl.table(_lib.TWID)) == ent["twid_sha256"]
        assert sha(pl.table(_lib.INV_TWID)) == ent["inv_twid_sha256"]
        if ent["has_shoup"]:
            assert (info.n_inv_mod_p_shoup, info.p_barrett) == (ent["n_inv_shoup"], ent["p_barrett"])
            assert pl.table(_lib.TWID_SHOUP)[:8].tolist() == ent["twid_shoup_head"]
            assert pl.table(_lib.INV_TWID_SHOUP)[:8].tolist() == ent["inv_twid_shoup_head"]
            assert sha(pl.table(_lib.TWID_SHOUP)) == ent["twid_shoup_sha256"]
        else:
            assert pl.table(_lib.TWID_SHOUP) is None
        assert pl.ntt_size() == ent["n"] and pl.modulus() == ent["p"]


@pytest.mark.parametrize("bits,n,p", [(64, 4096, 4611686018427322369), (64, 2048, 18446744069414584321),
                                      (64, 1024, 1125899904679937), (64, 32768, 1125899904679937),
                                      (64, 2048, 2251799813554177),
                                      (64, 512, 9223372036853661697), (32, 8192, 1073479681),
                                      (32, 256, 4293918721), (32, 2048, 2147352577)])
def test_plans_match_oracle(oracle, bits, n, p):
    pl = _mod(bits).Plan.try_new(n, p)
    ref = oracle.Plan.try_new(n, p, bits)
    for which, name in ((_lib.TWID, "twid"), (_lib.TWID_SHOUP, "twid_shoup"), (_lib.INV_TWID, "inv_twid"),
                        (_lib.INV_TWID_SHOUP, "inv_twid_shoup")):
        a, b = pl.table(which), ref.table(name)
        assert (a is None) == (b is None)
        if a is not None:
            assert np.array_equal(a, b), name
    info = pl.info()
    assert (info.n_inv_mod_p, info.big_q) == (ref.n_inv_mod_p, ref.big_q)
    if info.has_shoup:
        assert (info.n_inv_mod_p_shoup, info.p_barrett) == (ref.n_inv_mod_p_shoup, ref.p_barrett)
    expect_cls = 0 if p < (1 << (bits - 2)) else (1 if p < (1 << (bits - 1)) else 2)
    if bits == 64 and p < (1 << 51) and n <= 16384:
        expect_cls = 3 if p < (1 << 50) else 4   # double-precision FMA butterflies (csrc/ntt_arith.hpp, CLS_FP / CLS_FP51)
    if bits == 64 and p >= (1 << 63) and (1 << 64) - p < (1 << 32) and n <= 16384:
        expect_cls = 5   # p = 2^64 - c, c < 2^32 (CLS_PM64)
    if bits == 32 and p >= (1 << 31) and n <= 16384:
        expect_cls = 6   # 32-bit words on double-precision butterflies (CLS_FPW)
    assert info.arith_class == expect_cls


def test_try_new_none_and_panics(golden):
    for c in golden["try_new_none"]:
        assert _mod(c["bits"]).Plan.try_new(c["n"], c["p"]) is None, c
    for m in (prime64, prime32):
        for p in (0, 1):
            with pytest.raises(cntt.Panic):
                m.Plan.try_new(64, p)
    # test_plan_crash_github_11: src/prime64.rs:1879-1882
    assert prime64.Plan.try_new(2048, 1024) is None


def test_native_plans_host_side():
    from concrete_ntt_amd import native32, native64, native128, native_binary32, native_binary64, native_binary128
    kinds = [(native32.Plan32, 3, 4), (native64.Plan32, 5, 4), (native128.Plan32, 10, 4),
             (native_binary32.Plan32, 2, 4), (native_binary64.Plan32, 3, 4), (native_binary128.Plan32, 5, 4),
             (native32.Plan52, 2, 8), (native64.Plan52, 3, 8), (native_binary32.Plan52, 1, 8),
             (native_binary64.Plan52, 2, 8)]
    p32 = [1062862849, 1063059457, 1064697857, 1065484289, 1068236801, 1068433409, 1068564481, 1069219841,
           1071513601, 1073479681]  # src/lib.rs:453-462
    p52 = [1125899881086977, 1125899885412353, 1125899886395393]  # src/lib.rs:601-603
    for cls, k, res in kinds:
        pl = cls.try_new(256)
        assert pl is not None and pl.ntt_size() == 256
        assert _lib.lib().cntt_native_nprimes(pl._h) == k
        for i in range(k):
            sub = pl.ntt(i)
            assert sub.ntt_size() == 256 and sub.modulus() == (p52 if res == 8 else p32)[i]
        assert pl.ntt(k) is None
        assert cls.try_new(16) is None or res == 8      # prime32 needs n >= 32, prime64 n >= 16
        assert cls.try_new(1 << 17) is None             # the primes are 1 mod 2^17 only
        assert cls.try_new(48) is None


def test_ew_grid_stays_below_the_dispatch_limit():
    """Grid of the element-wise kernels (fill, split, CRT, pointwise, global stages): one 256-thread block per 256 work items, but
    never more than a dispatch can hold -- 2^32 - 1 work-items per dimension, i.e. fewer than 2^24 blocks (the
    round-3 cap of 2^30 blocks made every launch over 2^32 items fail with hipErrorInvalidConfiguration; 16 GiB of u32 is a
    plausible fill on a 288 GB part).  The kernels are grid-stride with 64-bit indices, so the capped grid covers any count."""
    import ctypes
    f = cntt.lib().cntt_ew_grid
    f.restype, f.argtypes = ctypes.c_uint, [ctypes.c_size_t]
    for count in (1, 255, 256, 257, 65536 * 1024, (2**24 - 1) * 256, (2**24 - 1) * 256 + 1, 2**32 - 1, 2**32, 2**32 + 1, 2**34, 2**36,
                  2**40):
        g = f(count)
        assert 1 <= g <= 2**24 - 1 and g * 256 <= 2**32 - 1, (count, g)
        if count <= (2**24 - 1) * 256:
            assert g == (count + 255) // 256, (count, g)


def test_no_gpu_means_loud_failure():
    if cntt.device_count() > 0:
        pytest.skip("a GPU is present")
    pl = prime64.Plan.try_new(64, 4611686018427322369)
    with pytest.raises(cntt.DeviceError):
        pl.fwd(np.zeros(64, dtype=np.uint64))


def test_length_assert_is_a_panic():
    pl = prime32.Plan.try_new(64, 1062862849)
    with pytest.raises(cntt.Panic):  # assert_eq!(buf.len(), self.ntt_size()): src/prime32.rs:710
        pl.fwd(np.zeros(32, dtype=np.uint32))
    with pytest.raises(cntt.Panic):
        pl.inv(np.zeros(128, dtype=np.uint32))


def test_async_load_kernels_do_not_spill(tmp_path):
    """The persistent kernels prefetch with inline-asm global loads (csrc/ntt_kernel.hpp gather_async, csrc/ntt_blk.hpp
    Pf::issue): a register the compiler spills or reassigns while such a load is in flight would be overwritten when the
    data lands.  Read the gfx950 code objects of the units that use them and require zero spills and zero scratch in every
    persistent kernel -- except shapes compiled WITHOUT the prefetch for exactly that reason (mul_kernel_blk<..., PREFETCH =
    false>, ordinary loads only: none in the current build, ntt_mul_one.hpp).
    The objects are part of the build: their absence fails (a rebuild with another compiler must not skip this)."""
    import re
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(llvm, "clang-offload-bundler")):
        # a box without the ROCm toolchain cannot read the code objects: skip THERE -- wherever the tools exist
        # (the build image, the GPU box) missing objects still fail below; CNTT_REQUIRE_CODE_OBJECTS=1 turns the skip into a failure
        assert os.environ.get("CNTT_REQUIRE_CODE_OBJECTS") != "1", "ROCm LLVM tools not present"
        pytest.skip("ROCm LLVM tools not present on this machine")
    objdir = os.path.join(ROOT, "concrete-ntt_amd", "csrc", "_obj")
    checked, exempt = 0, 0
    for unit in ("ntt_inst_u64_fwd", "ntt_inst_u64_inv", "ntt_inst_u32_fwd", "ntt_inst_u32_inv", "ntt_inst_u64_mul",
                 "ntt_inst_u32_mul", "ntt_inst_u64_fp", "ntt_inst_u64_fp51", "ntt_inst_u64_pm"):
        obj = os.path.join(objdir, unit + ".o")
        assert os.path.exists(obj), "objects not built in-tree (run __graft_entry__.build())"
        fat, co = str(tmp_path / (unit + ".fat")), str(tmp_path / (unit + ".co"))
        subprocess.run([os.path.join(llvm, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        "--input=" + fat, "--output=" + co, "--unbundle"], check=True)
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", co], check=True, capture_output=True,
                               text=True).stdout
        for blk in notes.split("- .agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            if "_wp" not in name and "_blk" not in name:
                continue
            spills = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1))
            scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
            if re.search(r"mul_kernel_blk.*Lb0EEEv", name):   # <..., PREFETCH = false>
                exempt += 1
                continue
            assert spills == 0 and scratch == 0, (name, spills, scratch)
            checked += 1
    assert checked > 80 and exempt <= 6, (checked, exempt)
    shutil.rmtree(str(tmp_path), ignore_errors=True)


def _build_examples():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "examples")], check=True)


def test_c_examples_build_and_fail_loudly_without_gpu():
    """The C ABI is plain C (gcc -std=c11 -pedantic on examples/*.c), include/cntt.hpp is plain C++17, and there is no CPU path: without a GPU the
    first compute call returns CNTT_EDEVICE and the program says so."""
    _build_examples()
    import torch
    if torch.cuda.is_available() or cntt.device_count() > 0:
        pytest.skip("GPU present: covered by test_c_examples_run_on_gpu")
    for name, needle in (("mul_poly_prime", "status 4"), ("mul_poly_native", "status 4"), ("readme_example", "device error")):
        r = subprocess.run([os.path.join(ROOT, "examples", name)], capture_output=True, text=True)
        assert r.returncode == 1 and needle in r.stderr, (name, r.returncode, r.stderr)


@pytest.mark.gpu
def test_c_examples_run_on_gpu():
    """examples/mul_poly_prime.c and mul_poly_native.c (the reference's examples/*.rs through the C ABI) and the
    README example + a product::Plan round trip through the C++17 mirror include/cntt.hpp."""
    _build_examples()
    for name in ("mul_poly_prime", "mul_poly_native", "readme_example"):
        r = subprocess.run([os.path.join(ROOT, "examples", name)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "Success!" in r.stdout, (name, r.returncode, r.stdout, r.stderr)

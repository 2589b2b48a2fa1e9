"""gfx950 has no interlock between a VALU write of an SGPR (VCC included: carry-outs, compare results, v_readfirstlane) and a VALU read of
it: two wait states are required in between.  hipcc pads its own code (`s_nop 1`); nothing pads the inside of an inline-asm string, and a
violation there gives wrong values only now and then -- a passing parity test is no evidence (the round-4 `csub_p`, a four-instruction
carry / select sequence, was written without them at first and passed every test).  This test replays the rule over the disassembly of
EVERY kernel of the library: for each VALU instruction that writes scalar registers, no VALU instruction within the next two wait states
may read them.  Straight-line scan in layout order (a branch in between only adds distance)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
OBJ = os.path.join(ROOT, "concrete-ntt_amd", "csrc", "_obj")

TWO_DST = ("v_add_co_", "v_sub_co_", "v_subrev_co_", "v_addc_co_", "v_subb_co_", "v_subbrev_co_", "v_mad_u64_u32", "v_mad_i64_i32",
           "v_div_scale_")


def sregs(tok):
    """scalar registers named by one operand token: {'vcc'} or {s-register numbers}"""
    tok = tok.strip()
    if tok in ("vcc", "vcc_lo", "vcc_hi"):
        return {"vcc"}
    m = re.match(r"s\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"s(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def split_operands(text):
    rest = text.split(None, 1)[1] if " " in text else ""
    toks, cur = [], ""
    for part in rest.split(","):
        cur = cur + "," + part if cur else part
        if cur.count("[") == cur.count("]"):
            toks.append(cur.strip().split()[0] if cur.strip() else "")
            cur = ""
    return toks


def scalar_defs_uses(text):
    """(scalar registers written, scalar registers read) by one VALU instruction"""
    op = text.split()[0]
    toks = split_operands(text)
    ndst = 1
    if op.startswith(TWO_DST):
        ndst = 2
    defs, uses = set(), set()
    for i, t in enumerate(toks):
        (defs if i < ndst else uses).update(sregs(t))
    if op.startswith("v_cmpx"):
        defs = set()          # writes EXEC (another rule)
    return defs, uses


def kernels(tmp_path):
    if not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        assert os.environ.get("CNTT_REQUIRE_CODE_OBJECTS") != "1", "ROCm LLVM tools not present"
        pytest.skip("ROCm LLVM tools not present on this machine")
    units = sorted(f[:-2] for f in os.listdir(OBJ) if f.endswith(".o"))
    assert len(units) >= 25, "objects not built in-tree (run __graft_entry__.build())"
    for unit in units:
        fat, co = str(tmp_path / (unit + ".fat")), str(tmp_path / (unit + ".co"))
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", os.path.join(OBJ, unit + ".o"), fat],
                       check=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        "--input=" + fat, "--output=" + co, "--unbundle"], check=True)
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], check=True, capture_output=True,
                             text=True).stdout
        name, body = None, []
        for ln in dis.split("\n") + ["0 <end>:"]:
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
            if m:
                if name and body:
                    yield unit, name, body
                name, body = m.group(1), []
            elif ln.startswith("\t"):
                body.append(ln.split("//")[0].strip())


def violations(body):
    out = []
    pending = []    # [registers, wait states still owed, text of the writer]
    for text in body:
        if not text:
            continue
        op = text.split()[0]
        if op.startswith("v_"):
            defs, uses = scalar_defs_uses(text)
            for regs, owed, writer in pending:
                if owed > 0 and regs & uses:
                    out.append((writer, text))
            states = 1
        elif op == "s_nop":
            defs, states = set(), int(text.split()[1], 0) + 1
        else:
            defs, states = set(), 1
        pending = [[r, o - states, w] for r, o, w in pending if o - states > 0]
        if op.startswith("v_") and defs:
            pending.append([defs, 2, text])
    return out


def test_two_wait_states_between_a_valu_scalar_write_and_a_valu_read(tmp_path):
    checked = 0
    for unit, name, body in kernels(tmp_path):
        bad = violations(body)
        assert not bad, "%s %s: `%s` is read by `%s` within two wait states" % (unit, name, bad[0][0], bad[0][1])
        checked += 1
    assert checked > 300


def test_the_checker_sees_a_violation():
    assert violations(["v_subrev_co_u32_e32 v4, vcc, s20, v2", "v_subb_co_u32_e32 v5, vcc, v3, v9, vcc"])
    assert violations(["v_cmp_lt_u64_e64 s[0:1], v[2:3], v[4:5]", "s_nop 0", "v_cndmask_b32_e64 v1, v3, v1, s[0:1]"])
    assert not violations(["v_subrev_co_u32_e32 v4, vcc, s20, v2", "s_nop 1", "v_subb_co_u32_e32 v5, vcc, v3, v9, vcc"])
    assert not violations(["v_mad_u64_u32 v[2:3], s[0:1], v4, v5, 0", "v_mov_b32_e32 v2, v3", "v_mul_lo_u32 v9, v8, v7",
                           "v_cndmask_b32_e64 v3, 0, 1, s[0:1]"])

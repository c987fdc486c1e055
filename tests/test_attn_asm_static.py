"""Static check of the attention kernels' hand-counted LDS waits (sdn_attn.hip: the inline-asm `ds_read_b64_tr_b16`
reads declare plain "=v" outputs and rely on a separate hand-counted `s_waitcnt lgkmcnt(N)`; nothing but the source's structure
stops a future compiler from touching those registers before the data has landed).  This test compiles csrc/sdn_attn.hip to gfx950
assembly with the Makefile's flags (hipcc cross-compiles without a GPU, ~10 s) and scans every kernel: between an LDS read and the
`s_waitcnt lgkmcnt(N)` that retires it (LDS operations return in order; the wait retires all but the N youngest), NO instruction may
name the read's destination registers.  The compiler's own reads obey that by construction, so a hit means the hand-written
pipeline was broken.  The hipcc version the kernels were validated on is recorded; a different compiler only warns."""
import collections
import os
import re
import shutil
import subprocess
import warnings

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "safe_denoiser_amd", "csrc")
VALIDATED_ON = "roc-7.2.0"                     # `hipcc --version` of the toolchain the asm pipeline was validated with (round 2/3)


def _regs(token):
    """v12 -> {12}; v[4:7] -> {4,5,6,7}; anything else -> empty."""
    m = re.fullmatch(r"v(\d+)", token)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", token)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def scan(asm_text):
    """Returns (violations, n_kernels, n_tr_reads)."""
    violations, n_k, n_tr = [], 0, 0
    pending = collections.deque()               # destination register sets of the LDS operations still in flight, oldest first
    name = None
    for ln, raw in enumerate(asm_text.splitlines(), 1):
        line = raw.split(";")[0].strip()
        if not line:
            continue
        if line.endswith(":") and not line.startswith("."):
            name, pending = line[:-1], collections.deque()
            n_k += 1
            continue
        if line.startswith(".") or name is None:
            continue
        op, _, rest = line.partition(" ")
        toks = [t.strip() for t in re.split(r"[,\s]+", rest) if t.strip()]
        if op == "s_endpgm":
            name = None
            continue
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", rest)
            if m:
                keep = int(m.group(1))
                while len(pending) > keep:
                    pending.popleft()
            continue
        if op.startswith("ds_"):
            dest = _regs(toks[0]) if ("read" in op or "load" in op) and toks else set()
            used = set().union(*[_regs(t) for t in (toks[1:] if dest else toks)]) if toks else set()
            for d in pending:
                if d & (used | dest):
                    violations.append((name, ln, raw.strip()))
            pending.append(dest)
            n_tr += op == "ds_read_b64_tr_b16"
            continue
        if op.startswith(("s_load", "s_buffer_load")):
            pending.append(set())                # scalar loads share the counter (their data returns out of order: only makes waits stricter)
            continue
        used = set().union(*[_regs(t) for t in toks]) if toks else set()
        if used:
            for d in pending:
                if d & used:
                    violations.append((name, ln, raw.strip()))
                    break
    return violations, n_k, n_tr


def test_scanner_catches_a_premature_use():
    bad = "k:\n ds_read_b64_tr_b16 v[2:3], v1\n ds_read_b64_tr_b16 v[4:5], v1 offset:64\n s_waitcnt lgkmcnt(1)\n v_mov_b32 v9, v4\n s_endpgm\n"
    good = bad.replace("v_mov_b32 v9, v4", "v_mov_b32 v9, v2")
    assert len(scan(bad)[0]) == 1 and scan(good)[0] == []


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_no_instruction_touches_an_lds_read_destination_before_its_wait(tmp_path):
    ver = subprocess.run(["hipcc", "--version"], capture_output=True, text=True).stdout
    if VALIDATED_ON not in ver:
        warnings.warn(f"sdn_attn.hip's asm LDS pipeline was validated on {VALIDATED_ON}; this hipcc is:\n{ver}")
    out = tmp_path / "attn.s"
    cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-function", "-mllvm",
           "-amdgpu-mfma-vgpr-form", "--cuda-device-only", "-S", os.path.join(CSRC, "sdn_attn.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, cwd=CSRC)
    violations, n_k, n_tr = scan(out.read_text())
    assert n_k >= 8 and n_tr >= 100, (n_k, n_tr)                        # every head dim / QS instantiation is in there
    assert not violations, violations[:5]

"""Builds gdmcf_amd/csrc/libgdmcf_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m gdmcf_amd.build [--force]

The .so is kept in-tree (git-ignored) so it travels to the GPU box with the source snapshot.
"""
import concurrent.futures
import hashlib
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SOURCES = ["capi.hip", "dp_exchange.hip", "gemm_f32.hip", "gemm_bf16.hip", "gemm_split.hip", "gemm_small.hip", "gemm_dr.hip", "kernels_misc.hip", "linear.hip", "topk_spmm.hip", "spmm_bundle.hip"]
HEADERS = ["common.h", "gemm_epilogue.h", os.path.join("..", "..", "include", "gdmcf_hip.h")]
LIB = os.path.join(CSRC, "libgdmcf_hip.so")
ASM_LINT = ("gemm_dr.hip", "gemm_split.hip")  # disassembled and run through lint_vmcnt + lint_store_data at every build
STORE_LINT = ("gemm_f32.hip", "gemm_bf16.hip")  # disassembled for lint_store_data only
NO_SPILL = ("gemm_f32.hip", "gemm_bf16.hip", "gemm_split.hip", "gemm_dr.hip")  # kernels with uncounted asm loads: a spill is a build error
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest():
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


# ---- ISA lint for the register-streaming kernels (gemm_dr.hip: dr_tn_kernel) -----------------------------------------------
# Their operand ring lives in registers that asm buffer loads write long after the asm statement: hipcc believes the value is
# there at once.  That is only safe while NO compiler-generated instruction reads or writes such a register other than the
# MFMAs that consume it as an A / B operand -- a v_mov (PHI copy, e.g. after loop unswitching), a temporary placed in a slot
# the compiler considers dead, an accumulator rotated through it, or a spill would move stale or half-landed data (seen:
# single accumulator registers wrong in lanes 12-15 of each row of 16).  The lint makes that invariant a build error.
_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def _regs(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


_SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")


def _sregs(text):
    out = set()
    for m in _SREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def lint_ring_registers(asm_text, kernel_prefix="dr_tn_kernel"):
    """Returns a list of violations (strings) over every function of `asm_text` whose name contains `kernel_prefix`."""
    bad, seen = [], 0
    for m in re.finditer(r"^(\S*%s\S*):[^\n]*\n(.*?)\n\s*s_endpgm" % kernel_prefix, asm_text, flags=re.S | re.M):
        name, body = m.group(1), m.group(2)
        seen += 1
        lines = [ln.split(";")[0].strip() for ln in body.splitlines()]
        lines = [ln for ln in lines if ln and not ln.endswith(":") and not ln.startswith(".")]
        ring = set()
        for ln in lines:
            if ln.startswith("buffer_load_dwordx4"):
                ring |= _regs(ln.split(",")[0])
        if not ring:
            bad.append(f"{name}: no asm ring loads found (lint out of date?)")
            continue
        # loads can be in flight from the first ring load to the final drain (the last s_waitcnt vmcnt(0) of the function):
        # before and after, the same registers are ordinary temporaries
        first = next(i for i, ln in enumerate(lines) if ln.startswith("buffer_load_dwordx4"))
        drains = [i for i, ln in enumerate(lines) if ln.startswith("s_waitcnt") and "vmcnt(0)" in ln]
        last = drains[-1] if drains and drains[-1] > first else len(lines)
        for ln in lines[first:last]:
            mnem, _, ops = ln.partition(" ")
            parts = [o.strip() for o in ops.split(",")]
            if mnem.startswith("buffer_load_dwordx4"):
                if _regs(",".join(parts[1:])) & ring:
                    bad.append(f"{name}: ring register used as an address: {ln}")
            elif mnem.startswith("v_mfma"):
                # accumulators rotated through ring registers: the two pools are no longer separate, nothing below can be checked
                if (_regs(parts[0]) | _regs(parts[3] if len(parts) > 3 else "")) & ring:
                    bad.append(f"{name}: MFMA accumulator inside the operand ring: {ln}")
            elif mnem.startswith(("v_mov", "v_pk_mov", "v_accvgpr", "scratch_", "v_swap")):
                # a copy / spill of a ring register moves data that may not have landed.  (Other VALU instructions that use a
                # slot as a temporary between its last MFMA and its next load are legitimate and not flagged.)
                if _regs(ops) & ring:
                    bad.append(f"{name}: copy or spill of a ring register: {ln}")
    if not seen:
        bad.append(f"no function named *{kernel_prefix}* in the assembly (lint out of date?)")
    return bad


_COPY = ("v_mov", "v_pk_mov", "v_accvgpr", "scratch_", "v_swap")
_VMEM = ("buffer_", "global_", "flat_", "scratch_")


def lint_vmcnt(asm_text, only=None):
    """Checks the hand-counted waits AND the no-copy invariant of every kernel that issues INLINE-ASM vector loads, by abstract
    interpretation over the kernel's control-flow graph.  State: for each register written by an asm load that may still be
    in flight, its RANK = how many vector-memory instructions have been issued since (vmcnt retires in order, so
    `s_waitcnt vmcnt(N)` lands every register of rank >= N; merge at joins = the lower rank).  Any instruction that reads,
    copies, spills or overwrites a register while it is pending is a violation: a consumer placed before its counted wait, a
    v_mov / v_accvgpr / scratch copy hipcc inserted (PHI copies after loop unswitching, spills), an accumulator rotated over a
    slot that has not landed.  Path-insensitive: kernels whose waits are selected by the same predicate as their loads
    (gemm_f32.hip, gemm_bf16.hip: `if (more) load(); ... if (more) wait(N) else wait(0)`) raise false alarms and are not run
    through it; gemm_dr.hip (all kernels) and gemm_split.hip verify cleanly and are checked at every build."""
    bad, nk = [], 0
    for m in re.finditer(r"^(_Z\S+):[^\n]*\n(.*?)\n\s*s_endpgm", asm_text, flags=re.S | re.M):
        name, body = m.group(1), m.group(2)
        if only and only not in name:
            continue
        ins, in_asm = [], False
        for ln in body.splitlines():
            t = ln.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            c = t.split(";")[0].strip()
            if not c or (c.startswith(".") and not c.endswith(":")):
                continue
            ins.append((c, in_asm))
        if not any(a and "load_dword" in c.split()[0] for c, a in ins):
            continue
        nk += 1
        leaders, labels = {0}, {}
        for k, (c, a) in enumerate(ins):
            if c.endswith(":"):
                labels[c[:-1]] = k
                leaders.add(k)
            elif c.startswith(("s_cbranch", "s_branch")):
                leaders.add(k + 1)
        starts = sorted(x for x in leaders if x < len(ins))
        blocks = [(st, starts[i + 1] if i + 1 < len(starts) else len(ins)) for i, st in enumerate(starts)]
        blk_of = {st: i for i, (st, en) in enumerate(blocks)}
        succ = []
        for bi, (st, en) in enumerate(blocks):
            mn, _, ops = ins[en - 1][0].partition(" ")
            nxt = []
            if mn.startswith(("s_cbranch", "s_branch")) and labels.get(ops.strip()) is not None:
                nxt.append(blk_of[labels[ops.strip()]])
            if not mn.startswith("s_branch") and bi + 1 < len(blocks):
                nxt.append(bi + 1)
            succ.append(nxt)
        found = {}

        def transfer(bi, st_in, report):
            st = dict(st_in)
            fresh = {}  # SGPRs written by a VALU instruction (v_readlane of a spilled SGPR, v_readfirstlane, v_cmp) -> wait states left
            for k in range(*blocks[bi]):
                ln, a = ins[k]
                if ln.endswith(":"):
                    continue
                mnem, _, ops = ln.partition(" ")
                parts = [o.strip() for o in ops.split(",")]
                # gfx9 / CDNA data hazard the compiler cannot see through inline asm: an SGPR written by a VALU instruction must
                # not be read by a vector-memory instruction (descriptor, scalar offset) for 5 wait states -- hipcc pads its own
                # instructions with s_nop, an asm load right behind a v_readlane of its soffset reads the OLD value (seen: the
                # hybrid kernel's fill loads with SGPR spills)
                if a and mnem.startswith(_VMEM):
                    hot = _sregs(ops) & fresh.keys()
                    if hot and report:
                        found.setdefault(f"s{min(hot)} written by a VALU instruction < 5 wait states before this asm load reads it: {ln}", k)
                step = int(ops) + 1 if mnem == "s_nop" and ops.strip().isdigit() else 1
                fresh = {r: w - step for r, w in fresh.items() if w - step > 0}
                if mnem.startswith("v_") and parts and re.fullmatch(r"s\d+|s\[\d+:\d+\]", parts[0]):
                    for r in _sregs(parts[0]):
                        fresh[r] = 5
                if mnem == "s_waitcnt":
                    mm = re.search(r"vmcnt\((\d+)\)", ops)
                    if mm:
                        st = {r: rk for r, rk in st.items() if rk < int(mm.group(1))}
                    continue
                asm_load = a and "load_dword" in mnem
                touched = _regs(",".join(parts[1:]) if asm_load else ops) & st.keys()
                if touched and report:
                    what = "used as an address" if asm_load else ("copied / spilled" if mnem.startswith(_COPY) else "used")
                    found.setdefault(f"register v{min(touched)} {what} before a counted wait covers its load: {ln}", k)
                if mnem.startswith(_VMEM):
                    st = {r: min(rk + 1, 64) for r, rk in st.items()}
                    if asm_load:
                        for r in _regs(parts[0]):
                            st[r] = 0
            return st

        state = [None] * len(blocks)
        state[0] = {}
        work = [0]
        while work:
            bi = work.pop()
            out = transfer(bi, state[bi], False)
            for sj in succ[bi]:
                if state[sj] is None:
                    state[sj] = dict(out)
                    work.append(sj)
                else:
                    changed = False
                    for r, rk in out.items():
                        if state[sj].get(r, 65) > rk:
                            state[sj][r] = rk
                            changed = True
                    if changed:
                        work.append(sj)
        for bi in range(len(blocks)):
            if state[bi] is not None:
                transfer(bi, state[bi], True)
        bad += [f"{name}: {msg}" for msg, k in sorted(found.items(), key=lambda kv: kv[1])]
    if not nk:
        bad.append("no kernel with inline-asm loads in the assembly (lint out of date?)")
    return bad


def lint_store_data(asm_text):
    """gfx9 / CDNA: a vector-memory store of more than 64 bits reads its data registers after it has issued; a VALU write of one
    of them needs wait states in between (2 on gfx940+).  LLVM's hazard recognizer pads this only when the store's scalar offset
    is NOT a register -- `buffer_store_dwordx4 v[146:149], v0, s[36:39], s10 offen` followed directly by `v_mov_b32 v146, ..` is
    what it emitted for the register-streaming kernel's epilogue, and on gfx950 lanes 12-15 of every 16-lane row of v146 then
    went to memory with the NEXT row's values in timing-dependent launches (the signature that had been blamed on
    copies of in-flight operand registers).  Any store of 3 or 4 dwords whose data registers are written within the next two
    instructions is reported, whoever emitted it."""
    bad = []
    for m in re.finditer(r"^(_Z\S+):[^\n]*\n(.*?)\n\s*s_endpgm", asm_text, flags=re.S | re.M):
        name, body = m.group(1), m.group(2)
        lines = [ln.split(";")[0].strip() for ln in body.splitlines()]
        lines = [ln for ln in lines if ln and not ln.startswith(".") and not ln.endswith(":")]
        for k, ln in enumerate(lines):
            mnem, _, ops = ln.partition(" ")
            if not re.match(r"(buffer|global|flat|scratch)_store_dwordx[34]$", mnem):
                continue
            parts = [o.strip() for o in ops.split(",")]
            data = _regs(parts[1] if mnem.startswith(("global", "flat", "scratch")) else parts[0])
            left = 2
            for nxt in lines[k + 1:k + 8]:
                nm, _, nops = nxt.partition(" ")
                if nm == "s_nop":
                    left -= int(nops.strip() or 0) + 1
                else:
                    nparts = [o.strip() for o in nops.split(",")]
                    if nm.startswith("v_") and nparts and _regs(nparts[0]) & data and left > 0:
                        bad.append(f"{name}: data register of `{ln}` rewritten {2 - left} wait state(s) later by `{nxt}`")
                        break
                    left -= 1
                if left <= 0:
                    break
    return bad


def build(force=False, verbose=True):
    stamp = os.path.join(CSRC, ".build_stamp")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    hipcc = _hipcc()
    objs = []

    def compile_one(src):
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        guarded = src in NO_SPILL
        if guarded:
            cmd.append("-Rpass-analysis=kernel-resource-usage")
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        err = r.stderr
        if guarded:
            # these kernels issue their tile loads as inline asm and place the s_waitcnt themselves: a stage register that the
            # compiler spills (or reloads) between the load and the wait would be touched before its data has arrived
            name, bad = None, []
            for line in err.splitlines():
                if "Function Name:" in line:
                    name = line.split("Function Name:")[1].split()[0]
                elif "VGPRs Spill:" in line and int(line.split("VGPRs Spill:")[1].split()[0]) > 0:
                    # (dr_fat_kernel and dr_kn_kernel have no inline-asm loads: every wait in them is hipcc's own, a spilled register
                    # -- lane constants saved across their 512-register loops -- is reloaded like any other value)
                    if "dr_fat_kernel" not in (name or "") and "dr_kn_kernel" not in (name or ""):
                        bad.append(name)
            if bad:
                raise RuntimeError(f"{src}: register spills in {len(bad)} kernel(s) with uncounted asm loads, e.g. {bad[0]}")
            err = err if ("warning:" in err or "error:" in err) else ""  # the remarks (and their source excerpts) are not news
        if verbose and err.strip():
            print(err, file=sys.stderr)
        if src in STORE_LINT:  # (their waits are path-dependent, see lint_vmcnt: only the store-data hazard is checked here)
            asm = os.path.join(CSRC, src.replace(".hip", ".lint.s"))
            r2 = subprocess.run([hipcc] + FLAGS + ["-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", asm],
                                capture_output=True, text=True)
            if r2.returncode != 0:
                raise RuntimeError(f"hipcc -S failed for {src}:\n{r2.stderr}")
            bad = lint_store_data(open(asm).read())
            os.remove(asm)
            if bad:
                raise RuntimeError(f"{src}: store data registers rewritten too early (lint_store_data):\n  " + "\n  ".join(bad[:12]))
        if src in ASM_LINT:
            asm = os.path.join(CSRC, src.replace(".hip", ".lint.s"))
            r2 = subprocess.run([hipcc] + FLAGS + ["-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", asm],
                                capture_output=True, text=True)
            if r2.returncode != 0:
                raise RuntimeError(f"hipcc -S failed for {src}:\n{r2.stderr}")
            text = open(asm).read()
            os.remove(asm)
            bad = lint_vmcnt(text) + lint_store_data(text)
            if src == "gemm_dr.hip":  # (EPI 5 = fused AdamW: accumulators rotate through the ring there, see lint_vmcnt)
                bad += [b for b in lint_ring_registers(text) if "ELi5EEEv" not in b.split(":")[0]]
            if bad:
                raise RuntimeError(f"{src}: in-flight operand registers touched (see lint_vmcnt / lint_ring_registers):\n  "
                                   + "\n  ".join(bad[:12]))
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    with open(stamp, "w") as fh:
        fh.write(dig)
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

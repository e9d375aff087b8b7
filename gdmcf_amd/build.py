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
                    bad.append(name)
            if bad:
                raise RuntimeError(f"{src}: register spills in {len(bad)} kernel(s) with uncounted asm loads, e.g. {bad[0]}")
            err = err if ("warning:" in err or "error:" in err) else ""  # the remarks (and their source excerpts) are not news
        if verbose and err.strip():
            print(err, file=sys.stderr)
        if src == "gemm_dr.hip":
            asm = os.path.join(CSRC, "gemm_dr.lint.s")
            r2 = subprocess.run([hipcc] + FLAGS + ["-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", asm],
                                capture_output=True, text=True)
            if r2.returncode != 0:
                raise RuntimeError(f"hipcc -S failed for {src}:\n{r2.stderr}")
            bad = [b for b in lint_ring_registers(open(asm).read()) if "ELi5EEEv" not in b.split(":")[0]]  # (EPI 5 = fused AdamW: not dispatched)
            os.remove(asm)
            if bad:
                raise RuntimeError("gemm_dr.hip: operand-ring invariant violated (see lint_ring_registers):\n  " + "\n  ".join(bad[:12]))
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

"""Builds gdmcf_amd/csrc/libgdmcf_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m gdmcf_amd.build [--force]

The .so is kept in-tree (git-ignored) so it travels to the GPU box with the source snapshot.
"""
import concurrent.futures
import hashlib
import os
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

"""Build libunet_hip.so (gfx950) in-tree with hipcc.  No torch involved: the library is a plain
C-ABI shared object (include/unet_hip.h) that travels to the GPU box with the snapshot."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(ROOT, "include")
# Two libraries of the same C ABI: libunet_hip.so (the product) and libunet_hip_pre.so (own object directory), which also carries
# the consumer-side BatchNorm + ReLU instantiations of the conv kernels (uh_conv3x3_fwd_pre / uh_conv3x3_wgrad_pre: built,
# measured, a net loss -- DESIGN.md section 3; the default library carries stubs that fail loudly).  __graft_entry__.build()
# builds both; tests/test_gpu_pre_fusion.py binds the second one for the length of its module; UH_LIB_PATH selects it for A/B runs.
# UH_BUILD_PRE=1 makes the flagged library the default target of build_library() (command-line use).
BUILD_PRE = os.environ.get("UH_BUILD_PRE") == "1"


def lib_path(pre: bool = False) -> str:
    return os.path.join(PKG_DIR, "libunet_hip_pre.so" if pre else "libunet_hip.so")


def obj_dir(pre: bool = False) -> str:
    return os.path.join(CSRC, "build_pre" if pre else "build")


LIB_PATH = lib_path(BUILD_PRE)
OBJ_DIR = obj_dir(BUILD_PRE)

SOURCES = ["uh_error.hip", "conv3x3.hip", "bn.hip", "bn_fused.hip", "pool_up.hip", "convt_1x1.hip", "convt_mfma.hip", "loss.hip", "optim.hip", "cc_loss.hip", "infer.hip", "post_process.hip", "data_prep.hip", "stem_mfma.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + INCLUDE, "-I" + CSRC,
         "-Wno-unused-result", "-Wno-unused-value", "-Wno-inline-asm"]
# sources whose kernels carry hand-counted waits around inline-asm loads: their device ISA is kept (-save-temps) and
# linted after every compile (isa_lint.py: no spill inside the MFMA region, no touch of an in-flight destination)
LINTED = {"conv3x3.hip": "conv3x3-hip-amdgcn-amd-amdhsa-gfx950.s"}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _lint_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_uh_isa_lint", os.path.join(PKG_DIR, "isa_lint.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    return lint


def check_toolchain() -> None:
    """The hand-counted s_waitcnt schedule of the LINTED sources depends on hipcc's instruction selection: another ROCm version
    must be opted into explicitly (and the parity tests run).  Checked BEFORE anything is compiled, so that an unvalidated
    toolchain fails in milliseconds instead of after a compile of the largest source -- once per rank, under the file lock."""
    lint = _lint_module()
    if lint.rocm_version() == lint.VALIDATED_ROCM:
        return
    msg = (f"{', '.join(LINTED)} validated on ROCm {lint.VALIDATED_ROCM}, this is {lint.rocm_version()}: the hand-counted waits depend "
           "on hipcc's instruction selection")
    if os.environ.get("UH_ALLOW_UNVALIDATED_ROCM") != "1":
        raise RuntimeError(msg + " -- set UH_ALLOW_UNVALIDATED_ROCM=1 to build anyway, then run the GPU parity tests")
    print("WARNING: " + msg + " (UH_ALLOW_UNVALIDATED_ROCM=1): run the parity tests before trusting this build", flush=True)


def build_library(force: bool = False, verbose: bool = False, pre: bool = BUILD_PRE) -> str:
    """Compile what is stale, lint the hand-scheduled kernels that were recompiled, link.  Serialised across processes by a
    file lock: every rank of a multi-process launch may call this, one of them builds, the others find everything fresh."""
    import fcntl
    os.makedirs(obj_dir(pre), exist_ok=True)
    with open(os.path.join(obj_dir(pre), ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose, pre)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force: bool, verbose: bool, pre: bool) -> str:
    OBJ_DIR, LIB_PATH = obj_dir(pre), lib_path(pre)
    EXTRA_DEFINES = ["-DUH_BUILD_PRE=1"] if pre else []
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(INCLUDE, "unet_hip.h"))
    hipcc = _hipcc()
    jobs = []
    objs = []
    rebuilt = set()
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            extra = ["-save-temps=obj"] if src in LINTED else []
            jobs.append([hipcc] + FLAGS + EXTRA_DEFINES + extra + ["-c", s, "-o", o])
            rebuilt.add(src)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)

    if any(src in LINTED for src in rebuilt):
        check_toolchain()
    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    # the lint reads (and prunes) the -save-temps by-products of THIS call's compile; objects that were not rebuilt were
    # linted when they were (their report is kept beside them)
    lint_isa(verbose, only={src for src in LINTED if src in rebuilt or
                            not os.path.exists(os.path.join(OBJ_DIR, src.replace(".hip", ".isa_lint.json")))}, pre=pre)
    if force or jobs or _stale(LIB_PATH, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs)
    return LIB_PATH


def lint_isa(verbose: bool = False, only=None, pre: bool = BUILD_PRE) -> None:
    """Check the ISA of the hand-scheduled kernels (see isa_lint.py); raises on a violation.  `only`: the sources to lint
    (build_library passes the ones it has just recompiled); None = every linted source whose ISA file exists."""
    lint = _lint_module()
    OBJ_DIR = obj_dir(pre)
    for src, isa in LINTED.items():
        path = os.path.join(OBJ_DIR, isa)
        if (only is not None and src not in only) or not os.path.exists(path):
            continue                      # prebuilt objects without temps (the GPU box uses the shipped .so)
        errs, report = lint.lint_asm(open(path).read())
        stem = src.replace(".hip", "")
        for f in os.listdir(OBJ_DIR):          # the other -save-temps by-products (13 MB) are of no use: keep the device ISA only
            if f != isa and (f.startswith(stem + "-") or f.startswith(stem + ".hip-")):
                os.remove(os.path.join(OBJ_DIR, f))
        with open(os.path.join(OBJ_DIR, src.replace(".hip", ".isa_lint.json")), "w") as f:
            import json
            json.dump({"rocm": lint.rocm_version(), "validated_rocm": lint.VALIDATED_ROCM, "kernels": report,
                       "violations": errs}, f, indent=1)
        # (the ROCm version gate ran before the compile: check_toolchain)
        if errs:
            os.remove(os.path.join(OBJ_DIR, src.replace(".hip", ".o")))       # never link a build that failed the lint
            raise RuntimeError("ISA lint failed for " + src + ":\n  " + "\n  ".join(errs))
        if verbose:
            loops = {k: v["scratch_in_enclosing_loops"] for k, v in report.items() if v.get("scratch_in_enclosing_loops")}
            print(f"isa lint {src}: {len(report)} guarded kernels clean" +
                  (f"; spill instructions inside MFMA loops (outside the MFMA stream): {sorted(loops.values())}" if loops else ""), flush=True)


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True, pre=BUILD_PRE or "--pre" in sys.argv))

"""Build a variant of libunet_hip.so HERE (hipcc cross-compiles gfx950 without a GPU) into scratch/libs/, which travels to the GPU
box with the snapshot (*.so is git-ignored, not gpurun-ignored): A/B legs then cost no box time for compiling.
    python scratch/r5_mklib.py NAME [--src conv3x3.hip=/path/to/other/version.hip ...] [-DUH_X=1 ...]
Sources not named are taken as objects from the main build; a named source is compiled from the given file (default: the tree's)
with the extra flags.  Select a variant with UH_LIB_PATH=$PWD/scratch/libs/libunet_hip_NAME.so."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unet-medical-image-contour-segmentation_amd"))
import build as B  # noqa: E402


def main():
    name = sys.argv[1]
    flags = [a for a in sys.argv[2:] if a.startswith("-D") or a.startswith("-m") or a.startswith("-f")]
    srcs = {}
    for a in sys.argv[2:]:
        if a.startswith("--src"):
            v = a.split("=", 1)[1] if a.startswith("--src=") else None
            k, _, path = v.partition("=")
            srcs[k] = path or os.path.join(B.CSRC, k)
    if not srcs:
        srcs = {"conv3x3.hip": os.path.join(B.CSRC, "conv3x3.hip")}
    B.build_library()
    out_dir = os.path.join(ROOT, "scratch", "libs")
    os.makedirs(out_dir, exist_ok=True)
    objs = []
    for src in B.SOURCES:
        o = os.path.join(B.obj_dir(False), src.replace(".hip", ".o"))
        if src in srcs:
            o = os.path.join(out_dir, f"{src[:-4]}_{name}.o")
            path = srcs[src]
            if os.path.dirname(os.path.abspath(path)) != B.CSRC:      # another version of the file: compile it beside its headers
                tmp = os.path.join(B.CSRC, f"_variant_{name}_{src}")
                shutil.copy(path, tmp)
                path = tmp
            cmd = [B._hipcc()] + B.FLAGS + flags + ["-c", path, "-o", o]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if path.startswith(os.path.join(B.CSRC, "_variant_")):
                os.remove(path)
            if r.returncode:
                raise SystemExit(r.stdout + r.stderr)
        objs.append(o)
    lib = os.path.join(out_dir, f"libunet_hip_{name}.so")
    subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
    print(lib)


if __name__ == "__main__":
    main()

"""Build a variant of libunet_hip.so with extra -D flags on selected sources (A/B kernel experiments):
    python scratch/mkvariant.py NAME [-DUH_X=1 ...] [--src conv3x3.hip,...]
-> scratch/variants/libunet_hip_NAME.so (objects of untouched sources are reused from the main build)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unet-medical-image-contour-segmentation_amd"))
import build as B  # noqa: E402


def main():
    name = sys.argv[1]
    flags = [a for a in sys.argv[2:] if a.startswith("-D") or a.startswith("-m") or a.startswith("-f")]
    srcs = ["conv3x3.hip"]
    for a in sys.argv[2:]:
        if a.startswith("--src="):
            srcs = a[6:].split(",")
    B.build_library()
    out_dir = os.path.join(ROOT, "scratch", "variants")
    os.makedirs(out_dir, exist_ok=True)
    objs = []
    for src in B.SOURCES:
        o = os.path.join(B.OBJ_DIR, src.replace(".hip", ".o"))
        if src in srcs:
            o = os.path.join(out_dir, f"{src[:-4]}_{name}.o")
            cmd = [B._hipcc()] + B.FLAGS + flags + ["-c", os.path.join(B.CSRC, src), "-o", o]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode:
                raise SystemExit(r.stdout + r.stderr)
        objs.append(o)
    lib = os.path.join(out_dir, f"libunet_hip_{name}.so")
    subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
    print(lib)


if __name__ == "__main__":
    main()

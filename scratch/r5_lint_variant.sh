#!/bin/bash
# compile an out-of-tree copy of conv3x3.hip with -save-temps and run the ISA lint over it
#   scratch/r5_lint_variant.sh /tmp/variant.hip [-DFLAG ...]
set -e
SRC=$1; shift
CS=/root/repo/unet-medical-image-contour-segmentation_amd/csrc
W=/tmp/lintv; rm -rf $W; mkdir -p $W
cp "$SRC" $CS/_variant_lint_conv3x3.hip
trap "rm -f $CS/_variant_lint_conv3x3.hip" EXIT
(cd $W && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/root/repo/include -I$CS -Wno-unused-result -Wno-unused-value -Wno-inline-asm "$@" -save-temps=obj -c $CS/_variant_lint_conv3x3.hip -o $W/conv.o 2>&1 | grep -B2 -A6 "error" | head -30 || true)
python - <<'PY'
import importlib.util, glob
spec = importlib.util.spec_from_file_location("lint", "/root/repo/unet-medical-image-contour-segmentation_amd/isa_lint.py")
lint = importlib.util.module_from_spec(spec); spec.loader.exec_module(lint)
f = glob.glob("/tmp/lintv/*gfx950*.s")[0]
errs, report = lint.lint_asm(open(f).read())
print("violations:", len(errs))
for e in errs[:8]: print("  ", e[:240])
PY

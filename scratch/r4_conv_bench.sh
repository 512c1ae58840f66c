#!/bin/bash
# scratch/r4_conv_bench.py on the default library and on variants built on the box:  scratch/r4_conv_bench.sh <outdir> "name:-Dflags ..." [batch]
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; mkdir -p $OUT; cd $R
for spec in $2; do n=${spec%%:*}; f=${spec#*:}; python scratch/mkvariant.py $n ${f//,/ } > $OUT/build_$n.log 2>&1 || { tail -20 $OUT/build_$n.log; exit 1; }; done
for rep in 1 2; do
  echo "== default ($rep)"; python scratch/r4_conv_bench.py ${3:-8} 2>&1 | grep -v amdgpu.ids | tee $OUT/base_$rep.txt
  for spec in $2; do n=${spec%%:*}; echo "== $n ($rep)"; UH_LIB_PATH=$R/scratch/variants/libunet_hip_$n.so python scratch/r4_conv_bench.py ${3:-8} 2>&1 | grep -v amdgpu.ids | tee $OUT/${n}_$rep.txt; done
done

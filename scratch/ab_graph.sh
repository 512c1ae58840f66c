#!/bin/bash
OUT=gpurun_out/$1; mkdir -p $OUT
for i in 1 2; do for v in eager graph graph_nonan; do
  case $v in eager) A="";E="UH_X=0";; graph) A="--graph";E="UH_X=0";; graph_nonan) A="--graph";E="UH_GRAPH_NO_NAN_CHECK=1";; esac
  env $E python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-inference --no-sustained --no-kernel-profile $A > $OUT/${v}_$i.json 2> $OUT/${v}_$i.err || tail -3 $OUT/${v}_$i.err
  python -c "
import json; j=json.load(open('$OUT/${v}_$i.json')); print('$v run=$i', j['value'], j['ms_per_step'], j['loss'])"
done; done

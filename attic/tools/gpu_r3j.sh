#!/bin/bash
OUT=gpurun_out/${1:-r3j}; mkdir -p $OUT
export MLX_COLD_ROWS=1 MLX_TRACE=1
P=64; rows=$((39063*P))
for mode in default coldsep0 rowng16; do
  unset MLX_COLD_SEP MLX_ROW_NG
  [ $mode = coldsep0 ] && export MLX_COLD_SEP=0
  [ $mode = rowng16 ] && export MLX_ROW_NG=16
  timeout 300 python tools/bench_sparse.py --rows $rows --partitions $P --steps 1 --warmup 1 > $OUT/$mode.json 2> $OUT/$mode.err
  echo "$mode rc=$?"; grep -v "^\[mlx\]" $OUT/$mode.err | head -2; grep -c "^\[mlx\]" $OUT/$mode.err; tail -c 300 $OUT/$mode.json | cut -c1-160
done

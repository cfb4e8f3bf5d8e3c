#!/bin/bash
# Calibrates FETCH_SIZE / WRITE_SIZE on known access shapes (tools/ubench_pmc.hip) -- one --pmc pass per counter, the program
# directly behind `--`; a third and fourth pass read the L2's memory-side request counters the two derive from.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc_cal; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_pmc $R/tools/ubench_pmc.hip 2> $OUT/build.err || { cat $OUT/build.err; exit 1; }
timeout 300 /tmp/ubench_pmc > $OUT/shapes.jsonl 2> $OUT/run.err; echo "plain run rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- /tmp/ubench_pmc > $OUT/$c.out 2> $OUT/$c.err; echo "$c rc=$?"
done
timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $OUT/RDREQ -- /tmp/ubench_pmc > $OUT/RDREQ.out 2> $OUT/RDREQ.err; echo "RDREQ rc=$?"
timeout 600 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum --kernel-trace --output-format csv -d $OUT/WRREQ -- /tmp/ubench_pmc > $OUT/WRREQ.out 2> $OUT/WRREQ.err; echo "WRREQ rc=$?"
# keep only the small per-kernel counter tables
for d in FETCH_SIZE WRITE_SIZE RDREQ WRREQ; do
  f=$(ls -t $OUT/$d/*/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/${d}_counters.csv
  rm -rf $OUT/$d
done
cat $OUT/shapes.jsonl; ls -la $OUT

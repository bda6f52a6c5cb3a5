# round-5 evidence run (GPU box): kernel-trace stats of the bench command and of one batch proof (the product path), PMC counters (each group its own pass),
# per-class counters of the batch proof's kernels
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-genproof --no-e2e --no-cpu-baseline --no-bn128 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_batch -- $R/xlayer_prover_amd/host/zkprover_batch -c tools/config_runFile_BatchProof_synthetic_colmajor.json --proofs 1 > $O/prof_batch.out 2> $O/prof_batch.err
python3 tools/collect_counters.py $O/counters.json > $O/collect_counters.log 2>&1
cp $(ls $O/prof_bench/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv
cp $(ls $O/prof_batch/*/*kernel_stats.csv | head -1) $O/genbatchproof_cpp_kernel_stats.csv
python3 tools/collect_step_counters.py $O/genbatchproof_class_counters.json $O/genbatchproof_cpp_kernel_stats.csv > $O/collect_step_counters.log 2>&1
tail -3 $O/collect_step_counters.log
rm -rf $O/prof_bench $O/prof_batch

set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-genproof --no-cpu-baseline --no-bn128 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_batch -- $R/xlayer_prover_amd/host/zkprover_batch -c tools/config_runFile_BatchProof_synthetic.json --proofs 1 > $O/prof_batch.out 2> $O/prof_batch.err
python3 tools/collect_counters.py $O/counters.json > $O/collect_counters.log 2>&1
for m in 0 1 2; do XLP_EXPR_CUMASK=$m $R/xlayer_prover_amd/host/zkprover_batch -c tools/config_runFile_BatchProof_synthetic.json --proofs 2 2> $O/cumask_$m.err | tail -1 > $O/cumask_$m.json; done
python3 tools/bench_sharded_cpp.py --log-n 20 --worlds 1,2,4 > $O/sharded_cpp_stage_timers_2p20.json 2> $O/sharded_cpp_stage_timers_2p20.err
ls $O/prof_bench $O/prof_batch | head -30

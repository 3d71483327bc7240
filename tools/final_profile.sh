#!/bin/bash
# Round-end evidence: bench line, rocprofv3 kernel trace of the same command, PMC HBM-traffic passes (separate runs).
#   gpurun -- 'bash tools/final_profile.sh <tag>'     -> gpurun_out/final_<tag>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-v6}; O=$R/gpurun_out/final_$TAG; mkdir -p $O
python3 $R/bench.py --steps 10 --warmup 3 > $O/bench_line.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_line_rocprof_run.json 2> $O/rocprof.err
python3 $R/tools/rocpd_stats.py $O/trace/bench_results.db > $O/kernel_stats.txt 2>> $O/rocprof.err
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o r --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>> $O/rocprof.err
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o r --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>> $O/rocprof.err
python3 $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write > $O/pmc_traffic.json 2>> $O/rocprof.err
rm -rf $O/trace $O/pmc_fetch $O/pmc_write
head -c 600 $O/bench_line.json; echo; head -12 $O/kernel_stats.txt

#!/bin/bash
# Round-end evidence for the HEAD being measured: bench line, rocprofv3 kernel trace of the same command, PMC passes
# (HBM traffic: FETCH_SIZE / WRITE_SIZE in separate runs; attention: MFMA / VALU / LDS counters), all stamped with csrc_sha.
#   gpurun -- 'bash tools/final_profile.sh <tag>'     -> gpurun_out/final_<tag>/   (then copy into profiles/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r04}; O=$R/gpurun_out/final_$TAG; mkdir -p $O
python3 $R/bench.py --steps 20 --warmup 3 > $O/bench_line.json 2> $O/bench.err
python3 $R/bench.py --steps 10 --warmup 3 --precise selective --no-cpu-baseline --no-extras > $O/bench_line_selective.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_line_rocprof_run.json 2> $O/rocprof.err
python3 $R/tools/rocpd_stats.py $O/trace/bench_results.db > $O/kernel_stats.txt 2>> $O/rocprof.err
# PMC passes: a shallow launch queue.  Counter collection serialises every dispatch; with bench.py's unbounded queue (several forwards of
# ~1000 kernel nodes in flight) the first --pmc pass of round 3 sat for 23 minutes until it was killed.  Exported, not `env ...`: nothing may
# exec between rocprofv3 and the program.
export GDF_MAX_INFLIGHT=2
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o r --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>> $O/rocprof.err
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o r --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>> $O/rocprof.err
python3 $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write > $O/pmc_traffic.json 2>> $O/rocprof.err
# attention + dominant GEMM: MFMA busy, wave cycles, issue stalls, LDS activity / conflicts (one SQ pass: 8 slots)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY -d $O/pmc_sq -o r --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>> $O/rocprof.err
python3 $R/tools/pmc_summary.py $O/pmc_sq 2>> $O/rocprof.err | grep -A 9 "attn_kernel\|gemm_kernel<0, 256, 320, 9\|gemm_kernel<0, 256, 256, 8\|gemm_kernel<1, 256, 320, 9\|gemm_kernel<0, 128, 160" > $O/pmc_sq_attn_gemm.txt
unset GDF_MAX_INFLIGHT
rm -rf $O/trace $O/pmc_fetch $O/pmc_write $O/pmc_sq
head -c 600 $O/bench_line.json; echo; head -12 $O/kernel_stats.txt; head -30 $O/pmc_sq_attn_gemm.txt

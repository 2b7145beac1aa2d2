#!/bin/bash
# copy the summaries of gpurun_out/prof_<tag> (+ gpurun_out/bench_<tag>.json) into profiles/ under the names profiles/README.md lists
TAG=$1
SRC=gpurun_out/prof_$TAG
cp $SRC/trace/t_kernel_stats.csv profiles/${TAG}_kernel_stats.csv
cp $SRC/trace_exclusive/t_kernel_stats.csv profiles/${TAG}_exclusive_kernel_stats.csv
cp $SRC/summary.txt profiles/${TAG}_rocprofv3_summary.txt
cp $SRC/summary.json profiles/${TAG}_summary.json
[ -f gpurun_out/bench_$TAG.json ] && cp gpurun_out/bench_$TAG.json profiles/${TAG}_bench.json
ls -la profiles/${TAG}_*

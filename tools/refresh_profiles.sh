#!/bin/bash
# Regenerates the measurement artefacts kept under profiles/ on a GPU box:
#   gpurun --timeout 900 -- 'bash tools/refresh_profiles.sh r01'
# writes gpurun_out/<tag>_*; copy them into profiles/ afterwards (gpurun_out/ is scratch).
set -eo pipefail
tag=${1:-r01}
out=$PWD/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
rm -rf /tmp/prof && mkdir -p /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/stats -- python3 bench.py --no-cpu-baseline > "$out/${tag}_bench_under_rocprof.json"
cp /tmp/prof/stats/*/*kernel_stats.csv "$out/${tag}_bench_kernel_stats.csv"
python3 tools/kernel_durations.py /tmp/prof/stats/*/*kernel_trace.csv > "$out/${tag}_bench_kernel_durations.txt"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof/fetch -- python3 bench.py --no-cpu-baseline > /dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof/write -- python3 bench.py --no-cpu-baseline > /dev/null
python3 tools/pmc_compact.py /tmp/prof/fetch/*/*counter_collection.csv > "$out/${tag}_pmc_fetch_size.csv"
python3 tools/pmc_compact.py /tmp/prof/write/*/*counter_collection.csv > "$out/${tag}_pmc_write_size.csv"
python3 tools/derive_traffic.py /tmp/prof/fetch/*/*counter_collection.csv /tmp/prof/write/*/*counter_collection.csv profiles/traffic.json > /dev/null
cp profiles/traffic.json "$out/traffic.json"
python3 bench.py > "$out/${tag}_bench_default.json"
cat "$out/${tag}_bench_default.json"

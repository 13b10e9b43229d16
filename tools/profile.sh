#!/bin/bash
# Collect the rocprofv3 evidence for one bench configuration on the GPU box.
#   tools/profile.sh <tag> [config]     -> gpurun_out/prof_<tag>/{stats,fetch,write,lds}   (config: 2, 2m, 3, 4, 5 as bench.py --config)
# Every pass is its own rocprofv3 process; counters are never combined with tracing.
set -u
tag=${1:-r02_config2}
cfg=${2:-2}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
args="--config $cfg --steps 5 --warmup 2 --no-cpu --no-latency"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py $args > $out/stats.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py $args > $out/fetch.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py $args > $out/write.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/lds -- python3 bench.py $args > $out/lds.log 2>&1 || exit 1
echo done

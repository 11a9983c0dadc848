#!/bin/bash
# One GPU call of round 5's development loop: transitions tests + timings, rollout stamps (streamed / in-kernel policy),
# the bench line and the two-rank rehearsal.   gpurun --timeout 1100 -- 'bash tools/exp/round5_check.sh <tag>'
tag=${1:-c}
O=gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "transitions" > $O/${tag}_tr_tests.log 2>&1; echo "tr tests rc=$?"; tail -3 $O/${tag}_tr_tests.log
: > $O/${tag}_tr.txt
for args in "8 20000" "8 20000 10 compact" "4 2000000" "4 2000000 10 compact" "8 2000 10 compact" "4 200000 10 compact" "2 1000000 10 compact" "6 100000 10 compact"; do
  python3 tools/prof_transitions.py $args >> $O/${tag}_tr.txt 2>&1
done
grep -v amdgpu.ids $O/${tag}_tr.txt
V=$PWD/gym-mapf_amd/gym_mapf_amd/lib/variants/libmapf_hip_stamps.so
if [ -f $V ]; then
  MAPF_HIP_LIB=$V python3 tools/stamp_profile.py 65536 c3 > $O/${tag}_stamps_c3.txt 2>&1
  MAPF_HIP_LIB=$V python3 tools/stamp_profile.py 32768 c4 > $O/${tag}_stamps_c4s.txt 2>&1
fi
python3 bench.py --steps 20 --warmup 5 > $O/${tag}_bench.json 2> $O/${tag}_bench.err; echo "bench rc=$?"
python3 bench.py --gpus 2 --dist-backend gloo --share-device --steps 3 --warmup 1 --repeats 1 --no-side-legs --no-cpu-baseline --baseline-config-steps 3 > $O/${tag}_reh2.json 2> $O/${tag}_reh2.err; echo "reh rc=$?"
